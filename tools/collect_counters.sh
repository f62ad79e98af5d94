#!/bin/bash
# Per-kernel PMC counters, one rocprofv3 --pmc pass per counter group (kernel-trace only, each pass bounded by
# `timeout`), for the bench's posterior pipeline, the Viterbi pass (config 4) and the large-q path (config 5 shape).
# Run on the GPU box from the repo root:
#   bash tools/collect_counters.sh [tag]   ->  gpurun_out/counters_<tag>.json
# SQ counters: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves,
# SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles (MI355X_MICROARCH.md, "rocprofv3 PMC slots").
set -u
R=$PWD
TAG=${1:-latest}
OUT=$R/gpurun_out/pmc_counters_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_groups() {   # run_groups <name> <python args...>
  local name=$1; shift
  local i=0
  while read -r GROUP; do
    [ -z "$GROUP" ] && continue
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$OUT/${name}_g$i" -- \
        python3 "$@" > "$OUT/${name}_g$i.log" 2>&1
    echo "$name group $i ($GROUP) rc=$?"
  done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
GRBM_GUI_ACTIVE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
GROUPS
}
run_groups bench "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-accuracy
run_groups viterbi "$R/tools/experiments/vit_prof.py"
run_groups largeq "$R/tools/experiments/largeq_time.py"
python3 - "$OUT" "$R/gpurun_out/counters_$TAG.json" <<'PY'
import collections, csv, glob, json, sys
src, dst = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.Counter())
n = collections.defaultdict(set)
for f in glob.glob(src + "/*_g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not k.startswith("k_"):
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])].add((f, r["Dispatch_Id"]))
out = {k: dict({c: v / max(1, len(n[(k, c)])) for c, v in cs.items()}, launches_seen=max(len(n[(k, c)]) for c in cs))
       for k, cs in tot.items()}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
for k in sorted(out):
    c = out[k]
    if c.get("SQ_BUSY_CYCLES", 0) < 1e5 and c.get("SQ_WAVE_CYCLES", 0) < 1e7:
        continue
    print(k[:60], {x: "%.3g" % c[x] for x in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                            "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU") if x in c})
PY
