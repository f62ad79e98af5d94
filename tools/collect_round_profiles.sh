#!/bin/bash
# Everything profiles/ holds for a round, from one GPU-box session (run from the repo root):
#   bash tools/collect_round_profiles.sh r2
# bench line (with variants and the CPU baseline), rocprofv3 kernel stats of the bench's timed pipeline, PMC
# traffic (tools/collect_traffic.sh), and kernel stats of the Viterbi, large-q and posterior-gradient paths.
# Every step is bounded by `timeout`; the steps are chained with && so nothing runs after a failure.
set -u
R=$PWD
TAG=${1:-latest}
OUT=$R/gpurun_out/round_$TAG
mkdir -p "$OUT"
stats() {   # stats <name> <program args...>: rocprofv3 --kernel-trace --stats of `python3 <args>`
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv \
      -d "$OUT/prof_$name" -o "$name" -- python3 "$@" > "$OUT/$name.log" 2>&1 ) || return 1
  local f
  f=$(find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${TAG}_${name}_kernel_stats.csv"
}
timeout -k 10 900 python3 "$R/bench.py" > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err" && echo "bench ok" &&
stats bench "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-variants && echo "bench stats ok" &&
grep '^{' "$OUT/bench.log" | tail -1 > "$OUT/${TAG}_bench_under_rocprof.json" &&
bash "$R/tools/collect_traffic.sh" "$TAG" && cp "$R/gpurun_out/traffic_$TAG.json" "$OUT/${TAG}_traffic.json" && echo "traffic ok" &&
stats viterbi "$R/tools/experiments/vit_prof.py" && echo "viterbi stats ok" &&
stats largeq "$R/tools/experiments/largeq_time.py" && echo "largeq stats ok" &&
stats postgrad "$R/tools/experiments/postgrad_time.py" && echo "postgrad stats ok" &&
stats grad "$R/tools/experiments/grad_time.py" && echo "grad stats ok"
