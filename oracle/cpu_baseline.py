"""Loader and timing protocol of the C++/OpenMP CPU restatement (oracle/csrc/sfft_cpu.cpp) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__ and bench.py's cpu_baseline leg import this module; nothing under sfft_amd/ does.

`build()` compiles oracle/csrc/sfft_cpu.cpp with g++ into oracle/_build/libsfft_cpu.so (git-ignored; it travels to the GPU box
with the snapshot, and is rebuilt there if missing -- the image has g++).  `measure()` is what bench.py reports as `cpu_baseline`:
BASELINE.md section 3's protocol -- the restatement of the reference's Numpy path at the benchmark's own size (4096 x 4096,
KerHW 8, orders 2/2), on the same seeded synthetic pair as the GPU run's pair 0, at 8 threads (the reference default
NUM_CPU_THREADS_4SUBTRACT=8, sfft/CustomizedPacket.py:16) and at all cores.
"""
import ctypes
import os
import subprocess
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "sfft_cpu.cpp")
CXXFLAGS = ["-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", "-std=c++17"]


def _host_tag():
    """-march=native binds the library to the build host's instruction set; the file name carries a hash of the CPU model, its
    feature flags and the compiler flags, so a library built on one host is never loaded (SIGILL) on another -- the GPU box
    rebuilds its own on first use."""
    import hashlib
    model = flags = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("flags") and not flags:
                flags = line.split(":", 1)[1].strip()
            if model and flags:
                break
    except OSError:
        pass
    return hashlib.sha1((model + "|" + flags + "|" + " ".join(CXXFLAGS)).encode()).hexdigest()[:12]


LIB = os.path.join(HERE, "_build", "libsfft_cpu.%s.so" % _host_tag())

_lib = None


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["g++"] + CXXFLAGS + ["-o", LIB, SRC])
    return LIB


def physical_cores():
    """Physical cores this process may run on (one hardware thread per core: the restatement is barrier-heavy and memory-bound,
    SMT siblings only add contention)."""
    try:
        allowed = os.sched_getaffinity(0)
    except Exception:
        allowed = set(range(os.cpu_count() or 1))
    seen = set()
    try:
        for cpu in allowed:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % cpu
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        return max(1, len(seen))
    except Exception:
        return max(1, len(allowed))


def lib():
    global _lib
    if _lib is None:
        # read by libgomp when the library's OpenMP runtime starts: spread the team over the cores, one thread per core
        os.environ.setdefault("OMP_PROC_BIND", "spread")
        os.environ.setdefault("OMP_PLACES", "cores")
        L = ctypes.CDLL(build())
        i, d = ctypes.c_int, ctypes.c_void_p
        L.sfftcpu_solve.argtypes = [i, i, i, i, i, i, d, d, d, d, d, i, d]
        L.sfftcpu_apply.argtypes = [i, i, i, i, i, i, d, d, d, d, i, d]
        L.sfftcpu_gss.argtypes = [i, i, i, i, i, i, d, d, d, d, d, d, i, d]
        L.sfftcpu_fft2.argtypes = [i, i, d, i, i]
        for f in (L.sfftcpu_solve, L.sfftcpu_apply, L.sfftcpu_gss, L.sfftcpu_fft2, L.sfftcpu_max_threads):
            f.restype = i
        _lib = L
    return _lib


def _f8(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert not np.isnan(a).any()
    return a


def _check(rc):
    if rc == -4:
        raise np.linalg.LinAlgError("Singular matrix")
    if rc:
        raise RuntimeError("sfft_cpu: error %d" % rc)


def solve(I, J, w, DK, DB, cpr=True, nthreads=0, want_system=False):
    I, J = _f8(I), _f8(J)
    N0, N1 = I.shape
    Fab = (2 * w + 1) ** 2
    NEQ = (DK + 1) * (DK + 2) // 2 * Fab + (DB + 1) * (DB + 2) // 2
    sol = np.empty(NEQ)
    LH = np.empty((NEQ, NEQ)) if want_system else None
    rhs = np.empty(NEQ) if want_system else None
    st = np.zeros(7)
    _check(lib().sfftcpu_solve(N0, N1, w, DK, DB, int(cpr), I.ctypes.data, J.ctypes.data, sol.ctypes.data,
                               LH.ctypes.data if want_system else None, rhs.ctypes.data if want_system else None,
                               nthreads, st.ctypes.data))
    return (sol, LH, rhs, st) if want_system else (sol, st)


def apply(I, J, solution, w, DK, DB, cpr=True, nthreads=0):
    I, J, solution = _f8(I), _f8(J), _f8(solution)
    N0, N1 = I.shape
    diff = np.empty((N0, N1))
    st = np.zeros(4)
    _check(lib().sfftcpu_apply(N0, N1, w, DK, DB, int(cpr), I.ctypes.data, J.ctypes.data, solution.ctypes.data,
                               diff.ctypes.data, nthreads, st.ctypes.data))
    return diff, st


def gss(I, J, mI, mJ, w, DK, DB, cpr=True, nthreads=0):
    I, J, mI, mJ = _f8(I), _f8(J), _f8(mI), _f8(mJ)
    N0, N1 = I.shape
    Fab = (2 * w + 1) ** 2
    NEQ = (DK + 1) * (DK + 2) // 2 * Fab + (DB + 1) * (DB + 2) // 2
    sol, diff, st = np.empty(NEQ), np.empty((N0, N1)), np.zeros(11)
    _check(lib().sfftcpu_gss(N0, N1, w, DK, DB, int(cpr), I.ctypes.data, J.ctypes.data, mI.ctypes.data, mJ.ctypes.data,
                             sol.ctypes.data, diff.ctypes.data, nthreads, st.ctypes.data))
    return sol, diff, st


def fft2(a, inverse=False, nthreads=0):
    a = np.ascontiguousarray(a, dtype=np.complex128).copy()
    rc = lib().sfftcpu_fft2(a.shape[0], a.shape[1], a.ctypes.data, 1 if inverse else -1, nthreads)
    assert rc == 0
    return a


STAGES = ["prelim", "OMG", "GAM", "PSI", "PHI", "THE+DEL", "solve", "prelim(apply)", "twiddle tables", "Construct_FDIFF", "inverse DFT"]


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def measure(N0, N1, w, DK, DB, quick=True):
    """cpu_baseline object of bench.py.  quick: one warm-up (at all cores) and the median of `n_all` runs at all cores + one
    run at 8 threads, so that the default bench finishes in a few minutes; full: 3 warm-ups and the median of 10 at both counts
    (bench.py --cpu-full; its output is kept under profiles/)."""
    from sfft_amd.utils.synthetic import make_pair
    pair = make_pair(N0, N1, seed=1234, mask=True, sky=0.0, bkg_scale=0.05)      # the GPU run's pair 0
    args = (pair["REF"], pair["SCI"], pair["mREF"], pair["mSCI"], w, DK, DB, True)
    ncores = physical_cores()

    def run(nthreads, warm, n, budget_s):
        ts, last, t_begin = [], None, time.perf_counter()
        for k in range(warm + n):
            t0 = time.perf_counter()
            sol, diff, st = gss(*args, nthreads=nthreads)
            dt = time.perf_counter() - t0
            if k >= warm or (time.perf_counter() - t_begin) > budget_s:
                ts.append(dt)
                last = st
            if (time.perf_counter() - t_begin) > budget_s and ts:        # bounded: never more than the budget + one run
                break
        return float(np.median(ts)), ts, last, diff

    if quick:
        t_all, ts_all, st_all, diff = run(ncores, 1, 3, 40.0)
        t_8, ts_8, st_8, _ = run(8, 0, 1, 40.0)
    else:
        t_all, ts_all, st_all, diff = run(ncores, 3, 10, 300.0)
        t_8, ts_8, st_8, _ = run(8, 3, 10, 600.0)
    assert np.isfinite(diff).all()
    # headline = the faster of the two thread counts (the restatement's strided column passes and full-size table sweeps stop
    # scaling well before 128 threads on a two-socket host; both measurements are reported)
    best_t, best_n = (t_all, ncores) if t_all <= t_8 else (t_8, 8)
    fmt = lambda ts, full: ("3 warm-ups + median of %d" % len(ts)) if full else ("%d run(s)" % len(ts))
    return {"value": 1.0 / best_t, "unit": "image-pairs/s", "mpix_per_s": N0 * N1 / 1e6 / best_t, "cores": best_n, "kind": "port",
            "restatement": "C++/OpenMP restatement of the reference's Numpy path (oracle/csrc/sfft_cpu.cpp: same 17 functions, c2c fp64 "
                           "transforms of the same planes, full-size twiddle planes, per-pixel Construct_FDIFF, LU solve), pinned by the "
                           "reference-made fixtures (tests/test_cpu_restatement.py); own mixed-radix Stockham FFT; g++ %s" % " ".join(CXXFLAGS[:3]),
            "cpu_model": _cpu_model(), "physical_cores": ncores, "seconds_per_pair": best_t,
            "all_cores": {"value": 1.0 / t_all, "seconds_per_pair": t_all, "runs": ts_all, "cores": ncores,
                          "stage_s": dict(zip(STAGES, [float(v) for v in st_all]))},
            "threads_8": {"value": 1.0 / t_8, "seconds_per_pair": t_8, "runs": ts_8, "cores": 8,
                          "stage_s": dict(zip(STAGES, [float(v) for v in st_8]))},
            "sample": "one full GSS (solve on the masked pair + apply) of the %dx%d pair with seed 1234 (pair 0 of the GPU batch), KerHW %d, "
                      "orders %d/%d, no size scaling: %s at %d threads (all physical cores, one thread per core) and %s at 8 threads (the "
                      "reference's default NUM_CPU_THREADS_4SUBTRACT); `value` is the faster of the two (%d threads)"
                      % (N0, N1, w, DK, DB, ("1 warm-up + " + fmt(ts_all, False)) if quick else fmt(ts_all, True), ncores,
                         fmt(ts_8, not quick), best_n)}
