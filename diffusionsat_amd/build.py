"""Build the HIP extension ``libdsat.so`` in-tree for gfx950.

``hipcc`` cross-compiles without a GPU, so this runs in the build container; the
resulting ``.so`` travels with the repository snapshot to the GPU box.

Every ``csrc/*.hip`` is compiled to its own object (in parallel, only when it or a header
changed) and the objects are linked into one shared library.
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libdsat.so")
ARCH = "gfx950"

# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs.  With AGPR accumulators hipcc
# (ROCm 7.2) shuffles them through v_accvgpr_read/write every loop iteration of the
# weight-streaming GEMM (236 copies per k-block ring turn), costing ~10 % of its time.
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "dsat.h")]


def _deps():
    return _sources() + _headers()


def _extra_flags():
    """Experiment switches: DSAT_CXXFLAGS="-DDSAT_PD1=2 ..." -- part of the staleness key: the flags a library
    was built with are recorded next to it (``<lib>.flags``), and a library built with other flags is stale."""
    return os.environ.get("DSAT_CXXFLAGS", "").split()


def _built_flags(lib: str):
    try:
        with open(lib + ".flags") as f:
            return f.read().split()
    except OSError:
        return []          # libraries from before the sidecar existed were built without extra flags


def is_stale(out: str = LIB, extra=None) -> bool:
    if not os.path.exists(out):
        return True
    if _built_flags(out) != (list(extra) if extra is not None else _extra_flags()):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(p) > t for p in _deps() if os.path.exists(p))


def _hipcc() -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build diffusionsat_amd/libdsat.so")
    return hipcc


def build(force: bool = False, verbose: bool = False, out: str = LIB) -> str:
    """Compile every .hip under csrc/ into one shared library; returns its path."""
    extra = _extra_flags()
    if not force and not is_stale(out, extra):
        return out
    hipcc = _hipcc()
    tag = "" if not extra else "_" + hashlib.sha1(" ".join(extra).encode()).hexdigest()[:8]
    objdir = OBJDIR + tag
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers() if os.path.exists(h))

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            return obj
        cmd = [hipcc] + FLAGS + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, res.stdout, res.stderr))
        if verbose and res.stderr.strip():
            print(res.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    tmp = out + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, out)
    with open(out + ".flags", "w") as f:
        f.write(" ".join(extra))
    return out


def source_hash() -> str:
    """sha1 over csrc/ and include/dsat.h: identifies the kernels a measurement (e.g. the PMC traffic file bench.py
    reads) was taken on."""
    h = hashlib.sha1()
    for p in _deps():
        if os.path.exists(p):
            h.update(os.path.basename(p).encode())
            with open(p, "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
