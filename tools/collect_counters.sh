#!/bin/bash
# Per-kernel PMC counters of the bench's posterior pipeline, one rocprofv3 --pmc pass per group
# (kernel-trace only, each pass bounded by `timeout`).  Run on the GPU box from the repo root:
#   bash tools/collect_counters.sh [tag]   ->  gpurun_out/counters_<tag>.json
set -u
R=$PWD
TAG=${1:-latest}
OUT=$R/gpurun_out/pmc_counters_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$OUT/g$i" -- \
      python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-variants > "$OUT/g$i.log" 2>&1
  echo "group $i ($GROUP) rc=$?"
done <<'GROUPS'
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVES
GROUPS
python3 - "$OUT" "$R/gpurun_out/counters_$TAG.json" <<'PY'
import collections, csv, glob, json, sys
src, dst = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: collections.Counter())
n = collections.defaultdict(set)
for f in glob.glob(src + "/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
out = {k: {c: v / max(1, len(n[(k, c)])) for c, v in cs.items()} for k, cs in tot.items() if k.startswith("k_")}
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
for k in ("k_reduce_sparse", "k_forward", "k_backward"):
    print(k, json.dumps(out.get(k, {}), sort_keys=True))
PY
