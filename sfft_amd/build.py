"""Build recipe for the HIP library (gfx950 only).  `python -m sfft_amd.build` or __graft_entry__.build().
Two translation units -- csrc/sfft_amd.hip (everything but the pivoted LU) and csrc/lu.hip (the LU kernels: one panel kernel
instantiation per register-tile shape, minutes of compile time) -- compiled side by side into build/*.o and linked into one
libsfft_amd.so; an object is rebuilt only when one of its own sources is newer."""
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
SRC = os.path.join(CSRC, "sfft_amd.hip")
LIB = os.path.join(PKG_DIR, "libsfft_amd.so")
def _obj_dir():
    """<checkout>/build/obj in a source tree; for an installed package (the parent is site-packages: shared, maybe read-only) a per-user cache."""
    parent = os.path.dirname(PKG_DIR)
    if os.path.isdir(os.path.join(parent, ".git")) or os.path.exists(os.path.join(parent, "bench.py")):
        return os.path.join(parent, "build", "obj")
    return os.path.join(os.environ.get("XDG_CACHE_HOME", os.path.join(os.path.expanduser("~"), ".cache")), "sfft_amd", "obj")


OBJ_DIR = _obj_dir()
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# csrc/sfft_amd.hip only: matrix instructions take their accumulators in ordinary registers.  chol_dataflow needs more than 256 registers, and the
# compiler's default for such a kernel is the accumulation-register form of EVERY matrix instruction -- 16 v_accvgpr moves and a full-latency
# wait around each rank-4 update of chol_factor_diag (981 moves in the kernel, 177 with the flag): block step 19.7 -> 18.3 us (docs/LOG.md).
# Kernels that fit 256 registers (the Omega launches, vconv_tensor) are in this form already and compile to the same code.
MAIN_FLAGS = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
HEADER = os.path.join(os.path.dirname(PKG_DIR), "include", "sfft_amd.h")
LU_ONLY = ("lu.hip", "lu.hpp")                      # sources only csrc/lu.hip sees; lu_api.hpp is seen by both units


def _units():
    every = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    main_deps = [f for f in every if os.path.basename(f) not in LU_ONLY] + ([HEADER] if os.path.exists(HEADER) else []) + [os.path.abspath(__file__)]      # (this file: MAIN_FLAGS)
    lu_deps = [os.path.join(CSRC, f) for f in ("lu.hip", "lu.hpp", "lu_api.hpp")]
    return [(SRC, os.path.join(OBJ_DIR, "sfft_amd.o"), main_deps, MAIN_FLAGS), (os.path.join(CSRC, "lu.hip"), os.path.join(OBJ_DIR, "lu.o"), lu_deps, [])]


def _stale(target, deps):
    return (not os.path.exists(target)) or max(os.path.getmtime(d) for d in deps) > os.path.getmtime(target)


def needs_build():
    units = _units()
    every_source = sorted(set(d for _, _, deps, _ in units for d in deps))
    if os.path.exists(LIB) and not _stale(LIB, every_source):
        return False        # a library newer than every source is current even when only the .so was shipped (no objects beside it)
    return any(_stale(o, d) for _, o, d, _ in units) or _stale(LIB, [o for _, o, _, _ in units if os.path.exists(o)] or [SRC])


def build_library(force=False, verbose=True):
    units = _units()
    os.makedirs(OBJ_DIR, exist_ok=True)
    procs = []
    for src, obj, deps, extra in units:
        if force or _stale(obj, deps):
            cmd = [HIPCC] + CFLAGS + extra + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            plain = [c for c in cmd if c not in MAIN_FLAGS]
            if plain != cmd:        # a toolchain that does not know the -mllvm option: the sources are the same without it, only slower
                print("sfft_amd.build: retrying without %s" % " ".join(MAIN_FLAGS), flush=True)
                subprocess.run(plain, check=True)
                continue
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    objs = [o for _, o, _, _ in units]
    if procs or force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
