#!/bin/bash
# rocprofv3 --kernel-trace --stats of one python program, summary printed and kept:
#   bash tools/prof_one.sh NAME script.py [args...]     -> gpurun_out/prof_NAME/NAME_kernel_stats.csv
set -u
R=$PWD
name=$1; shift
OUT=$R/gpurun_out/prof_$name
mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv \
    -d "$OUT" -o "$name" -- python3 "$R/$1" "${@:2}" > "$OUT/$name.log" 2>&1 ) || { tail -5 "$OUT/$name.log"; exit 1; }
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel stats"; exit 1; }
cp "$f" "$OUT/${name}_kernel_stats.csv"
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-46s calls %5s avg %10.1f us  %6s %%" % (r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
grep -v "^W2\|^E2\|amdgpu.ids" "$OUT/$name.log" | tail -3
