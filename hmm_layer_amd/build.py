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


def needs_build():
    if not os.path.exists(LIB):
        return True
    deps = [SRC, os.path.join(INC, "hmm_engine.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + INC, SRC, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
