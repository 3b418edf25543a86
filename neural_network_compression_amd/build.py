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
SOURCES = [os.path.join(CSRC, "nnc_hip.hip"), os.path.join(CSRC, "nnc_sort.hip"), os.path.join(CSRC, "nnc_pp.hip"), os.path.join(CSRC, "nnc_codec.hip"), os.path.join(CSRC, "nnc_layer.hip")]
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
    deps = SOURCES + [os.path.join(INCLUDE, "nnc.h"), os.path.join(CSRC, "nnc_lloyd.hpp")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    tmp = f"{LIB}.{os.getpid()}.tmp"   # built aside and renamed: another process never maps a half-written library
    cmd = [
        hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17",
        "-ffp-contract=off",  # the reference arithmetic is unfused float32; never contract
        "-fPIC", "-shared", "-I", INCLUDE, "-o", tmp,
    ] + (["-DNNC_DIAG"] if DIAG else []) + (["-DNNC_NO_HELP"] if os.environ.get("NNC_NO_HELP") else []) + SOURCES + EXTRA_LIBS
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + proc.stdout + proc.stderr)
    os.replace(tmp, LIB)
    if verbose:
        print(proc.stdout + proc.stderr)
    return LIB


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
