"""ctypes binding of libsfft_amd.so (the C ABI declared in include/sfft_amd.h).

There is no CPU fallback: if the HIP library is missing or fails to load, importing this
module raises, and so does every operator of the package.
"""
import ctypes
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFFT_AMD_LIB") or os.path.join(PKG_DIR, "libsfft_amd.so")     # (override: A/B builds of the same sources)

# error codes / query fields / stage ids: keep in sync with include/sfft_amd.h
SFFT_OK = 0
SFFT_ERR_INVALID_ARG = -1
SFFT_ERR_UNSUPPORTED_SIZE = -2
SFFT_ERR_HIP = -3
SFFT_ERR_SINGULAR = -4
SFFT_ERR_NOMEM = -5
SFFT_ERR_STALL = -6

QUERY_FIELDS = ["N0", "N1", "w0", "w1", "DK", "DB", "ConstPhotRatio", "L0", "L1", "Fab", "Fij", "Fpq", "NEQ", "Fijab",
                "NEQ_FSfree", "FOMG", "FGAM", "FTHE", "FPSI", "FPHI", "FDEL", "WORKSPACE_BYTES", "LAST_SOLVER",
                "NUM_GREEK_PAIRS", "ScaFij", "SOLVE_GRAPH", "THETA_FUSED", "OMG_OFFDIAG", "OMG_DIAG", "G1_DECIMATED", "G1_CHUNKS", "G1_MFMA",
                "CHOL_DATAFLOW", "SOLVER_N", "OMG_SPARSE", "CHOL_STATUS", "SOLVES", "LU_FALLBACKS", "CHOL_STALLS"]
STAGES = ["prelim_solve", "greek_g1", "greek_g2", "fill", "solve", "prelim_apply", "construct", "inverse", "greek_g1b",
          "fwd_rows", "fwd_cols"]

EXPORTS = ["sfft_plan_create", "sfft_plan_create_basis", "sfft_plan_create_varscale", "sfft_plan_set_regularization", "sfft_plan_destroy", "sfft_plan_query", "sfft_solve", "sfft_apply", "sfft_subtract",
           "sfft_get_system", "sfft_get_solver_system", "sfft_dbg_solve_dense", "sfft_dbg_forward_spectrum", "sfft_fft_plan_create", "sfft_fft2_r2c", "sfft_ifft2_c2r",
           "sfft_grid_convolve", "sfft_spec_abs2_accumulate", "sfft_real_rsqrt", "sfft_spec_multiply", "sfft_half_to_full_real", "sfft_set_timing", "sfft_stage_ms", "sfft_stage_kernels", "sfft_set_force_lu",
           "sfft_last_error", "sfft_version"]


class SfftLibraryMissing(RuntimeError):
    pass


class SfftError(Exception):
    """A non-zero status of the C ABI, in the reference's convention (an Exception whose text starts 'MeLOn ERROR:');
    `.code` keeps the ABI's return code (include/sfft_amd.h) for callers that record it, e.g. sharding.run_shard."""

    def __init__(self, code, msg):
        super().__init__("MeLOn ERROR: %s" % msg)
        self.code = int(code)


def _load():
    if not os.path.exists(LIB_PATH):
        raise SfftLibraryMissing(
            "sfft_amd: HIP library %s not found. Build it with `python -m sfft_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, dp, ip = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int
    lib.sfft_plan_create.argtypes = [ctypes.POINTER(vp), ip, ip, ip, ip, ip, ip, ip]
    lib.sfft_plan_create_basis.argtypes = [ctypes.POINTER(vp), ip, ip, ip, ip, ip, dp, dp, ip, dp, ip, ip, dp, dp, ip, dp, ip, ip]
    lib.sfft_plan_create_varscale.argtypes = [ctypes.POINTER(vp), ip, ip, ip, ip, ip, dp, dp, ip, dp, ip, ip, dp, dp, ip, dp,
                                              ip, ip, dp, dp, ip, dp, ip]
    lib.sfft_plan_set_regularization.argtypes = [vp, ctypes.c_double, dp, dp, dp, dp]
    lib.sfft_plan_destroy.argtypes = [vp]
    lib.sfft_plan_query.argtypes = [vp, ip, ctypes.POINTER(ctypes.c_longlong)]
    lib.sfft_solve.argtypes = [vp, dp, dp, dp, vp]
    lib.sfft_apply.argtypes = [vp, dp, dp, dp, dp, vp]
    lib.sfft_subtract.argtypes = [vp, dp, dp, dp, dp, dp, dp, vp]
    lib.sfft_get_system.argtypes = [vp, dp, dp, vp]
    lib.sfft_get_solver_system.argtypes = [vp, dp, dp, vp]
    lib.sfft_dbg_solve_dense.argtypes = [vp, dp, ip, dp, vp]
    lib.sfft_dbg_forward_spectrum.argtypes = [vp, dp, ip, ip, dp, vp]
    dbl, ll = ctypes.c_double, ctypes.c_longlong
    lib.sfft_fft_plan_create.argtypes = [ctypes.POINTER(vp), ip, ip, ip]
    lib.sfft_fft2_r2c.argtypes = [vp, dp, dp, dbl, vp]
    lib.sfft_ifft2_c2r.argtypes = [vp, dp, dp, dbl, vp]
    lib.sfft_spec_abs2_accumulate.argtypes = [dp, dp, dbl, dp, ll, vp]
    lib.sfft_real_rsqrt.argtypes = [dp, dp, ll, vp]
    lib.sfft_spec_multiply.argtypes = [dp, dp, ip, dp, ll, vp]
    lib.sfft_half_to_full_real.argtypes = [dp, dp, ip, ip, vp]
    lib.sfft_grid_convolve.argtypes = [dp, dp, dp, ip, ip, ip, ip, ip, dp, ip, vp]
    lib.sfft_set_timing.argtypes = [vp, ip]
    lib.sfft_stage_ms.argtypes = [vp, ip, ctypes.POINTER(ctypes.c_float)]
    lib.sfft_stage_kernels.argtypes = [vp, ip, ctypes.c_char_p, ip]
    lib.sfft_set_force_lu.argtypes = [vp, ip]
    for name in EXPORTS:
        getattr(lib, name).restype = ctypes.c_int
    lib.sfft_last_error.restype = ctypes.c_char_p
    lib.sfft_last_error.argtypes = []
    lib.sfft_version.restype = ctypes.c_char_p
    lib.sfft_version.argtypes = []
    return lib


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _load()
    return _LIB


def last_error():
    return lib().sfft_last_error().decode("utf-8", "replace")


def check(rc):
    """Turn a non-zero status into the reference's exception convention (Exception('MeLOn ERROR: ...'))."""
    if rc == SFFT_OK:
        return
    msg = last_error()
    if rc == SFFT_ERR_SINGULAR:
        import numpy as np
        raise np.linalg.LinAlgError(msg)      # what numpy.linalg.solve raises in the reference's Numpy backend
    raise SfftError(rc, msg)
