#!/bin/bash
# Register / LDS / spill usage per kernel of the engine (compiles device code only, no GPU needed).
#   bash tools/kernel_resources.sh [regex]
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
cd "$T" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans \
  -fno-slp-vectorize -I"$R/include" $DEFS --cuda-device-only -S "$R/hmm_layer_amd/csrc/hmm_engine.hip" -o eng.s 2>/dev/null
python3 - "$T/eng.s" "${1:-.}" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    import subprocess
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn).replace("void ", "")
    if not pat.search(dn):
        continue
    g = lambda k: (re.search(r"\.amdhsa_%s (\d+)" % k, body) or [0, "?"])[1]
    sp = re.search(r"; ScratchSize: (\d+)", txt[m.end():m.end() + 3000])
    vg = re.search(r"; NumVgprs: (\d+)", txt[m.end():m.end() + 3000])
    occ = re.search(r"; Occupancy: (\d+)", txt[m.end():m.end() + 3000])
    print("%-60s vgpr %4s  lds %6s  scratch %4s  occ %s" % (dn[:60], vg.group(1) if vg else g("next_free_vgpr"),
          g("group_segment_fixed_size"), sp.group(1) if sp else "?", occ.group(1) if occ else "?"))
PY
rm -rf "$T"
