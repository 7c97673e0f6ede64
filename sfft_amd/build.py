"""Build recipe for the HIP library (gfx950 only).  `python -m sfft_amd.build` or __graft_entry__.build()."""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG_DIR, "csrc", "sfft_amd.hip")
LIB = os.path.join(PKG_DIR, "libsfft_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    newest = max(os.path.getmtime(os.path.join(dp, f)) for dp, _, fs in os.walk(os.path.join(PKG_DIR, "csrc")) for f in fs)
    hdr = os.path.join(os.path.dirname(PKG_DIR), "include", "sfft_amd.h")
    if os.path.exists(hdr):
        newest = max(newest, os.path.getmtime(hdr))
    return newest > os.path.getmtime(LIB)


def build_library(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + ["-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
