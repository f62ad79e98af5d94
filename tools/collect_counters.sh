#!/bin/bash
# Per-kernel PMC counters, one rocprofv3 --pmc pass per counter group (kernel-trace only, each pass bounded by
# `timeout`), for the bench's posterior pipeline, the Viterbi pass (config 4) and the large-q path (config 5 shape).
# Run on the GPU box from the repo root:
#   bash tools/collect_counters.sh [tag] ["bench viterbi largeq"]   ->  gpurun_out/counters_<tag>.json
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
WHAT=${2:-bench viterbi largeq}
for W in $WHAT; do
  case $W in
    bench) run_groups bench "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-accuracy ;;
    viterbi) run_groups viterbi "$R/tools/experiments/vit_prof.py" ;;
    largeq) run_groups largeq "$R/tools/experiments/largeq_time.py" ;;
  esac
done
python3 "$R/tools/pmc_aggregate.py" "$OUT" "$R/gpurun_out/counters_$TAG.json"
