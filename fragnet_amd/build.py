"""Builds libfragnet_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = [os.path.join(HERE, "csrc", "fragnet_hip.hip")]
HEADERS = [os.path.join(ROOT, "include", "fragnet_hip.h")]
OUT = os.path.join(HERE, "lib", "libfragnet_hip.so")


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected on PATH or at /opt/rocm/bin/hipcc)")


def stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-I", os.path.join(ROOT, "include"), *SOURCES, "-o", OUT]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    return OUT


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
