"""Builds libsnappyhash.so (gfx950 HIP kernels + C-ABI host code) in-tree with hipcc.

Run as `python -m snappy_amd.build`.  hipcc cross-compiles for gfx950 without a
GPU present; the shared library lands next to this file so that it travels
with the source tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# SNAPPY_BUILD_TAG=<tag> builds a variant beside the product (objects in _obj_<tag>, library
# libsnappyhash_<tag>.so), e.g. a measurement build with SNAPPY_EXTRA_CXXFLAGS; tools load it
# through SNAPPY_LIB_PATH.
TAG = os.environ.get("SNAPPY_BUILD_TAG", "")
LIB = os.path.join(HERE, f"libsnappyhash_{TAG}.so" if TAG else "libsnappyhash.so")

SOURCES = [
    "sha1_kernels.hip",
    "sha1_plan.hip",
    "sha1_noalloc.hip",
    "lz4_kernels.hip", "lz4_encode.hip",
    "runtime.cpp", "transfer.cpp",
    "sha1_api.cpp",
    "lz4net_host.cpp",
    "mcdf_host.cpp",
    "mcdf_api.cpp",
    "host_mirror.cpp",
]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libsnappyhash needs the ROCm toolchain")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "include", "snappyhash.h"))
    objdir = os.path.join(HERE, f"_obj_{TAG}" if TAG else "_obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra",
              "-Wno-unused-parameter"]
    # Measurement builds: e.g. SNAPPY_EXTRA_CXXFLAGS=-DSNAPPY_LZ4_PHASE_SWITCHES with --force.
    common += os.environ.get("SNAPPY_EXTRA_CXXFLAGS", "").split()
    objs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + common + ["-c", src, "-o", obj]
            if src.endswith(".cpp"):
                cmd[1:1] = ["-x", "hip"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
