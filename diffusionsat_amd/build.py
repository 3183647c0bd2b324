"""Build the HIP extension ``libdsat.so`` in-tree for gfx950.

``hipcc`` cross-compiles without a GPU, so this runs in the build container; the
resulting ``.so`` travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdsat.so")
ARCH = "gfx950"


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
        os.path.join(HERE, "..", "include", "dsat.h")]


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps() if os.path.exists(p))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every .hip under csrc/ into one shared library; returns its path."""
    if not force and not is_stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build diffusionsat_amd/libdsat.so")
    tmp = LIB + ".tmp.%d" % os.getpid()
    # -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs.  With AGPR accumulators hipcc
    # (ROCm 7.2) shuffles them through v_accvgpr_read/write every loop iteration of the
    # weight-streaming GEMM (236 copies per k-block ring turn), costing ~10 % of its time.
    cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
           "-mllvm", "-amdgpu-mfma-vgpr-form=1",
           "-Wall", "-Wno-unused-function", "-o", tmp] + _sources()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
