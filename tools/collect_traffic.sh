#!/bin/bash
# HBM traffic per kernel launch from PMC counters, as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one pass), each
# with --kernel-trace only; every rocprofv3 call is bounded by `timeout`.
# Run on the GPU box from the repo root:  bash tools/collect_traffic.sh  [tag]
set -u
R=$PWD
TAG=${1:-latest}
OUT=$R/gpurun_out/pmc_traffic_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -- \
      python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-variants > "$OUT/$C.log" 2>&1
  echo "$C pass rc=$?"
done
python3 "$R/tools/pmc_to_traffic.py" "$OUT" "$R/gpurun_out/traffic_$TAG.json"
