"""Builds the in-tree native libraries with hipcc for gfx950.

  hmm_layer_amd/libhmm_engine.so   the HIP engine behind include/hmm_engine.h

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box
with the repo snapshot.  Run:  python -m hmm_layer_amd.build
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = os.path.join(PKG, "csrc", "hmm_engine.hip")
INC = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libhmm_engine.so")
ARCH = "gfx950"


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP engine cannot be built")
    return exe


def sources():
    """Every file the library is compiled from: the .hip translation unit, the .inc kernel files it
    includes, and the public header."""
    import glob
    csrc = os.path.join(PKG, "csrc")
    return sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.inc"))
                  + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(INC, "*.h")))


def source_hash():
    import hashlib
    h = hashlib.sha256()
    for f in sources():
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


STAMP = LIB + ".srchash"


def needs_build():
    """True unless the library exists and was built from exactly the current sources (content hash
    of csrc/*.hip, csrc/*.inc, include/*.h stored next to it — mtimes do not survive a snapshot)."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=False, out=None, defines=()):
    """Compile the engine.  `out` / `defines` build a tuning variant next to the product library
    (scratch experiments); the product is always hmm_layer_amd/libhmm_engine.so with defaults."""
    if out is None and not force and not needs_build():
        return LIB
    # -amdgpu-mfma-vgpr-form: MFMA results stay in VGPRs (gfx950's register file is unified);
    # without it hipcc parks them in AGPRs and pays a v_accvgpr_read + s_nop per result register
    # in every step of the recurrence.
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
           "-mllvm", "-amdgpu-mfma-vgpr-form",
           "-fno-honor-nans",        # inputs are finite: fmaxf needs no NaN-quieting extra v_max
           # measured on MI355X (tools/experiments/valu_rate.hip): v_fma/v_mul/v_add_f32 issue every ~2.5 cycles,
           # v_pk_fma_f32 every ~8 (slower per flop), v_pk_mul_f32 ~4.5 (neutral): keep the SLP
           # vectoriser from packing the recurrence's fma chains
           "-fno-slp-vectorize",
           "-I" + INC, SRC, "-o", out or LIB] + ["-D" + d for d in defines]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    if out is None and not defines:
        with open(STAMP, "w") as fh:
            fh.write(source_hash() + "\n")
    return out or LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
