"""Plan handle: Python owner of one `sfft_plan` (the cacheable replacement of SingleSFFTConfigure's JIT step)."""
import collections
import ctypes
import functools
import threading

import torch

from . import _lib


def _locked(fn):
    """A plan owns one set of workspaces, a second stream and one pinned status word: one call at a time.  Host threads that
    share a Plan object (same geometry through the plan cache) are serialised here; threads that want pairs in flight
    concurrently take plans of their own (`get_plan(..., slot=k)`)."""
    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        with self.lock:
            return fn(self, *a, **k)
    return wrapper


class Plan:
    def __init__(self, N0, N1, KerHW, DK=None, DB=None, ConstPhotRatio=True, device=0, basis=None):
        """Polynomial plan (DK, DB, ConstPhotRatio) or, with `basis`, a plan over tabulated separable bases:
        basis = dict(kbx=[nkx,N0], kby=[nky,N1], ker_pairs=[Fij,2], tbx=[nbx,N0], tby=[nby,N1], bkg_pairs=[Fpq,2],
                     scaling_mode=0|1|2)   (see sfft_plan_create_basis in include/sfft_amd.h);
        with the extra keys sbx=[nsx,N0], sby=[nsy,N1], sca_pairs=[ScaFij,2] the flux scaling varies on its own basis
        (sfft_plan_create_varscale; scaling_mode is ignored)."""
        self._h = ctypes.c_void_p()
        self.device = int(device)
        self.lock = threading.RLock()
        self._reg_token = None          # which SFFTConfig's regularisation the plan currently carries (BSplineSFFT.SSC)
        if basis is None:
            rc = _lib.lib().sfft_plan_create(ctypes.byref(self._h), int(N0), int(N1), int(KerHW), int(DK), int(DB),
                                             1 if ConstPhotRatio else 0, int(device))
        else:
            import numpy as np
            f8 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
            i4 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
            kbx, kby, tbx, tby = f8(basis["kbx"]), f8(basis["kby"]), f8(basis["tbx"]), f8(basis["tby"])
            kp, bp = i4(basis["ker_pairs"]), i4(basis["bkg_pairs"])
            assert kbx.shape[1] == N0 and tbx.shape[1] == N0 and kby.shape[1] == N1 and tby.shape[1] == N1
            assert kp.ndim == 2 and kp.shape[1] == 2 and bp.ndim == 2 and bp.shape[1] == 2
            self._keep = (kbx, kby, tbx, tby, kp, bp)
            if "sca_pairs" in basis:
                sbx, sby, sp = f8(basis["sbx"]), f8(basis["sby"]), i4(basis["sca_pairs"])
                assert sbx.shape[1] == N0 and sby.shape[1] == N1 and sp.ndim == 2 and sp.shape[1] == 2
                rc = _lib.lib().sfft_plan_create_varscale(
                    ctypes.byref(self._h), int(N0), int(N1), int(KerHW),
                    kbx.shape[0], kby.shape[0], kbx.ctypes.data, kby.ctypes.data, kp.shape[0], kp.ctypes.data,
                    sbx.shape[0], sby.shape[0], sbx.ctypes.data, sby.ctypes.data, sp.shape[0], sp.ctypes.data,
                    tbx.shape[0], tby.shape[0], tbx.ctypes.data, tby.ctypes.data, bp.shape[0], bp.ctypes.data, int(device))
            else:
                rc = _lib.lib().sfft_plan_create_basis(
                    ctypes.byref(self._h), int(N0), int(N1), int(KerHW),
                    kbx.shape[0], kby.shape[0], kbx.ctypes.data, kby.ctypes.data, kp.shape[0], kp.ctypes.data,
                    tbx.shape[0], tby.shape[0], tbx.ctypes.data, tby.ctypes.data, bp.shape[0], bp.ctypes.data,
                    int(basis.get("scaling_mode", 0)), int(device))
        _lib.check(rc)
        self.N0, self.N1 = int(N0), int(N1)
        self.NEQ = self.query("NEQ")
        self.Fijab = self.query("Fijab")
        self.Fpq = self.query("Fpq")

    def query(self, name):
        v = ctypes.c_longlong()
        _lib.check(_lib.lib().sfft_plan_query(self._h, _lib.QUERY_FIELDS.index(name), ctypes.byref(v)))
        return int(v.value)

    # -- helpers -----------------------------------------------------------------------------
    def _dev(self):
        return torch.device("cuda", self.device)

    @staticmethod
    def _stream_ptr(device):
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    def _check_img(self, t, name):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
                and tuple(t.shape) == (self.N0, self.N1) and t.device.index == self.device):
            raise Exception("MeLOn ERROR: %s must be a contiguous float64 tensor of shape [%d, %d] on cuda:%d"
                            % (name, self.N0, self.N1, self.device))

    # -- C ABI calls ---------------------------------------------------------------------------
    @_locked
    def solve(self, I, J):
        self._check_img(I, "PixA_I"); self._check_img(J, "PixA_J")
        sol = torch.empty(self.NEQ, dtype=torch.float64, device=self._dev())
        _lib.check(_lib.lib().sfft_solve(self._h, I.data_ptr(), J.data_ptr(), sol.data_ptr(), self._stream_ptr(self._dev())))
        return sol

    @_locked
    def apply(self, I, J, solution):
        self._check_img(I, "PixA_I"); self._check_img(J, "PixA_J")
        sol = solution.to(device=self._dev(), dtype=torch.float64).contiguous()
        if sol.numel() != self.NEQ:
            raise Exception("MeLOn ERROR: SFFTSolution must have %d entries" % self.NEQ)
        diff = torch.empty((self.N0, self.N1), dtype=torch.float64, device=self._dev())
        _lib.check(_lib.lib().sfft_apply(self._h, I.data_ptr(), J.data_ptr(), sol.data_ptr(), diff.data_ptr(),
                                         self._stream_ptr(self._dev())))
        return diff

    @_locked
    def subtract(self, I, J, mI, mJ, out_solution=None, out_diff=None):
        for t, n in ((I, "PixA_I"), (J, "PixA_J"), (mI, "PixA_mI"), (mJ, "PixA_mJ")):
            self._check_img(t, n)
        sol = out_solution if out_solution is not None else torch.empty(self.NEQ, dtype=torch.float64, device=self._dev())
        diff = out_diff if out_diff is not None else torch.empty((self.N0, self.N1), dtype=torch.float64, device=self._dev())
        _lib.check(_lib.lib().sfft_subtract(self._h, I.data_ptr(), J.data_ptr(), mI.data_ptr(), mJ.data_ptr(),
                                            sol.data_ptr(), diff.data_ptr(), self._stream_ptr(self._dev())))
        return sol, diff

    @_locked
    def get_system(self):
        LH = torch.empty((self.NEQ, self.NEQ), dtype=torch.float64, device=self._dev())
        rhs = torch.empty(self.NEQ, dtype=torch.float64, device=self._dev())
        _lib.check(_lib.lib().sfft_get_system(self._h, LH.data_ptr(), rhs.data_ptr(), self._stream_ptr(self._dev())))
        return LH, rhs

    @_locked
    def get_solver_system(self):
        """(A [n][n], rhs [n], index [n]) of the reduced system the factorisation received (sfft_get_solver_system)."""
        n = self.query("SOLVER_N")
        B = torch.empty((n + 1, n + 1), dtype=torch.float64, device=self._dev())
        idx = torch.empty(n, dtype=torch.int32, device=self._dev())
        _lib.check(_lib.lib().sfft_get_solver_system(self._h, B.data_ptr(), idx.data_ptr(), self._stream_ptr(self._dev())))
        return B[:n, :n], B[n, :n], idx

    @_locked
    def solve_dense(self, A, b, use_lu=True):
        """The dense solver alone (sfft_dbg_solve_dense): x with A x = b for a caller's [n][n] matrix, n = SOLVER_N.  use_lu: LU with
        partial pivoting (any nonsingular A), else the Cholesky path (symmetric positive definite A).  Raises LinAlgError when singular."""
        n = self.query("SOLVER_N")
        assert tuple(A.shape) == (n, n) and tuple(b.shape) == (n,), (A.shape, b.shape, n)
        B = torch.zeros((n + 1, n + 1), dtype=torch.float64, device=self._dev())
        B[:n, :n] = A
        B[:n, n] = b
        B[n, :n] = b
        x = torch.empty(n, dtype=torch.float64, device=self._dev())
        _lib.check(_lib.lib().sfft_dbg_solve_dense(self._h, B.data_ptr(), 1 if use_lu else 0, x.data_ptr(), self._stream_ptr(self._dev())))
        return x

    @_locked
    def forward_spectrum(self, I, i, j):
        self._check_img(I, "PixA_I")
        out = torch.empty((self.N0, self.N1 // 2 + 1), dtype=torch.complex128, device=self._dev())
        _lib.check(_lib.lib().sfft_dbg_forward_spectrum(self._h, I.data_ptr(), int(i), int(j), out.data_ptr(),
                                                        self._stream_ptr(self._dev())))
        return out

    @_locked
    def set_regularization(self, lam, ireg=None, sst=None, csst=None, dsst=None):
        """LHMAT += lam * SCALE^2 * S (x) ireg on every later solve (sfft_plan_set_regularization); lam = 0 switches it off."""
        import numpy as np
        f8 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        ireg, sst, csst, dsst = f8(ireg), f8(sst), f8(csst), f8(dsst)
        ptr = lambda a: None if a is None else a.ctypes.data
        self._reg_token = None          # a direct call: whatever config's penalty the plan carried is gone (see _bound_plan)
        _lib.check(_lib.lib().sfft_plan_set_regularization(self._h, float(lam), ptr(ireg), ptr(sst), ptr(csst), ptr(dsst)))

    def set_timing(self, enable=True):
        _lib.check(_lib.lib().sfft_set_timing(self._h, 1 if enable else 0))

    def set_force_lu(self, enable=True):
        _lib.check(_lib.lib().sfft_set_force_lu(self._h, 1 if enable else 0))

    @_locked
    def stage_ms(self):
        out = {}
        for k, name in enumerate(_lib.STAGES):
            v = ctypes.c_float()
            _lib.check(_lib.lib().sfft_stage_ms(self._h, k, ctypes.byref(v)))
            out[name] = float(v.value)
        return out

    @_locked
    def stage_kernels(self):
        """{stage: [kernel names]} -- the HIP kernels each stage has launched since set_timing(True) (sfft_stage_kernels)."""
        out = {}
        buf = ctypes.create_string_buffer(4096)
        for k, name in enumerate(_lib.STAGES):
            _lib.check(_lib.lib().sfft_stage_kernels(self._h, k, buf, len(buf)))
            out[name] = [n for n in buf.value.decode().split(";") if n]
        return out

    def close(self):
        with self.lock:
            if self._h:
                _lib.lib().sfft_plan_destroy(self._h)
                self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_CACHE = collections.OrderedDict()
_CACHE_MAX = 6
_CACHE_LOCK = threading.Lock()


def get_plan(N0, N1, KerHW, DK, DB, ConstPhotRatio, device=0, slot=0):
    """Plans are cached per (device, shape, KerHW, DK, DB, ConstPhotRatio): the reference re-JITs on every
    SSC call (sfft/PureCupyCustomizedPacket.py:139-141, 7.6 s); here repeated packets reuse tables and workspaces.
    `slot` distinguishes independent plans of the same geometry (one per host thread / stream when several pairs
    are pipelined on one GPU)."""
    key = (int(device), int(N0), int(N1), int(KerHW), int(DK), int(DB), bool(ConstPhotRatio), int(slot))
    with _CACHE_LOCK:
        p = _CACHE.get(key)
        if p is not None:
            _CACHE.move_to_end(key)
            return p
        p = Plan(N0, N1, KerHW, DK, DB, ConstPhotRatio, device)
        _CACHE[key] = p
        while len(_CACHE) > _CACHE_MAX:
            _CACHE.popitem(last=False)      # freed when the last SFFTConfig holding it goes away
        return p


def clear_plan_cache():
    """Drop the cache's references.  A plan is destroyed (Plan.__del__) once the last SFFTConfig that holds it is gone, so
    configs made before the call stay usable."""
    with _CACHE_LOCK:
        _CACHE.clear()
