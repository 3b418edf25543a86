"""Builds csrc/libnnc_hip.so (gfx950) in-tree with hipcc.

``python -m neural_network_compression_amd.build`` or ``build_native()``.  hipcc
cross-compiles without a GPU, so this also runs in the CPU-only build container; the
resulting .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
DIAG = os.environ.get("NNC_DIAG", "0") not in ("", "0")   # diagnostics build (phase traces, ablated kernels): tools/ only
LIB = os.path.join(CSRC, "libnnc_hip_diag.so" if DIAG else "libnnc_hip.so")
SOURCES = [os.path.join(CSRC, f) for f in ("nnc_hip.hip", "nnc_core.hip", "nnc_reduce.hip", "nnc_reffit.hip", "nnc_huffman.hip", "nnc_comm.hip", "nnc_sort.hip", "nnc_pp.hip",
                                            "nnc_codec.hip", "nnc_layer.hip")]   # (the longest unit first: the units compile side by side)
EXTRA_LIBS: list = []


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = SOURCES + _headers()
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    return [os.path.join(INCLUDE, "nnc.h")] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))


def _flags():
    return (["--offload-arch=gfx950", "-O3", "-std=c++17",
             "-ffp-contract=off",  # the reference arithmetic is unfused float32; never contract
             # the first sixteen dwords of a kernel's arguments arrive in SGPRs with the wave (gfx950 preloads them) instead of
             # through a scalar load at its head: one dependent memory round trip less in front of every launch's first useful
             # load -- the Lloyd loop is a chain of ~130 K-sized launches (measured: 2.73 -> 2.63 ms per bench step)
             "-mllvm", "-amdgpu-kernarg-preload-count=16",
             "-fPIC", "-I", INCLUDE]
            + (["-DNNC_DIAG"] if DIAG else []) + (["-DNNC_NO_HELP"] if os.environ.get("NNC_NO_HELP") else [])
            + os.environ.get("NNC_EXTRA_CXXFLAGS", "").split())   # (tuning experiments on the GPU box, e.g. -DOS_THREADS=256)


def build_native(force: bool = False, verbose: bool = False) -> str:
    """One object per translation unit under csrc/_obj/ (compiled side by side, re-made only when the unit or a header
    changed), then one link.  The library is built aside and renamed: another process never maps a half-written file."""
    if not force and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    hipcc = hipcc_path()
    objdir = os.path.join(CSRC, "_obj_diag" if DIAG else "_obj")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    log = []

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            return obj
        tmp = f"{obj}.{os.getpid()}.tmp"
        cmd = [hipcc] + _flags() + ["-c", src, "-o", tmp]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + proc.stdout + proc.stderr)
        os.replace(tmp, obj)
        log.append(proc.stdout + proc.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = f"{LIB}.{os.getpid()}.tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + EXTRA_LIBS
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc (link) failed:\n" + " ".join(cmd) + "\n" + proc.stdout + proc.stderr)
    os.replace(tmp, LIB)
    if verbose:
        print("".join(log) + proc.stdout + proc.stderr)
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
