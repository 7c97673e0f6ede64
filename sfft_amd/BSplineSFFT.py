"""B-spline form of SFFT on the HIP backend -- mirror of the core classes of sfft/BSplineSFFT.py
(SingleSFFTConfigure.SSC :2536-2607, ElementalSFFTSubtract.ESS :3865-3877, GeneralSFFTSubtract.GSS :3880-3965,
BSpline_Packet.BSP :3967-4356).

The kernel's and the background's spatial variation are separable per term (a function of the row times a function of
the column), for polynomials and B-splines alike; this module tabulates those 1-D factors on the host exactly like the
reference (scipy.interpolate.BSpline, BSplineSFFT.py:2624-2645) and hands them to `sfft_plan_create_basis`.

All three scaling modes of the reference (BSplineSFFT.py:47-60) are served: 'ENTANGLED' (SEPARATE_SCALING=False),
'SEPARATE-CONSTANT' (SEPARATE_SCALING=True, ScaSpDegree=0) and 'SEPARATE-VARYING' (ScaSpDegree > 0, the scaling has its
own polynomial / B-spline basis -> `sfft_plan_create_varscale`), each with optional REGULARIZE_KERNEL: the small
matrices of the Laplacian penalty (iREGMAT, SSTMAT, CSSTMAT, DSSTMAT, BSplineSFFT.py:3570-3686) are built here on the
host like the reference does and handed to `sfft_plan_set_regularization`; the device adds them while filling LHMAT.
"""
import os.path as pa
import time

import threading

import numpy as np
import torch

from ._packet_common import union_nan_mask, assign_roles, finish_diff
from .plan import Plan
from .sfftcore.SFFTSubtract import (ElementalSFFTSubtract, ElementalSFFTSubtract_PureCupy, GeneralSFFTSubtract,
                                    GeneralSFFTSubtract_PureCupy)
from .utils import minifits

__all__ = ["SingleSFFTConfigure", "ElementalSFFTSubtract", "GeneralSFFTSubtract", "GeneralSFFTSubtract_PureCupy",
           "BSpline_Packet", "Create_BSplineBasis", "Create_BSplineBasis_Req", "Read_SFFTSolution", "BSpline_MatchingKernel",
           "ConvKernel_Convertion", "BSpline_DeCorrelation", "BSpline_GridConvolve"]

try:  # pragma: no cover - astropy is absent from the target image
    from astropy.io import fits as _afits
except Exception:
    _afits = None


def Create_BSplineBasis(N, IntKnot, BSplineDegree):
    """Basis functions on the scaled pixel-centre coordinates (1..N)/N, boundary knots at 0.5/N and (N+0.5)/N
    (BSplineSFFT.py:2624-2645).  Returns [number of control points][N] float64."""
    from scipy.interpolate import BSpline
    PixCoord = (1.0 + np.arange(N)) / N
    Knot = np.concatenate(([0.5] * (BSplineDegree + 1), list(IntKnot), [N + 0.5] * (BSplineDegree + 1))) / N
    Nc = len(IntKnot) + BSplineDegree + 1
    out = []
    for idx in range(Nc):
        Coeff = (np.arange(Nc) == idx).astype(float)
        out.append(BSpline(t=Knot, c=Coeff, k=BSplineDegree, extrapolate=False)(PixCoord))
    return np.array(out).astype(np.float64)


def Create_BSplineBasis_Req(N, IntKnot, BSplineDegree, ReqCoord):
    """The same basis functions at requested scaled coordinates (BSplineSFFT.py:2636-2646)."""
    from scipy.interpolate import BSpline
    Knot = np.concatenate(([0.5] * (BSplineDegree + 1), list(IntKnot), [N + 0.5] * (BSplineDegree + 1))) / N
    Nc = len(IntKnot) + BSplineDegree + 1
    out = []
    for idx in range(Nc):
        Coeff = (np.arange(Nc) == idx).astype(float)
        out.append(BSpline(t=Knot, c=Coeff, k=BSplineDegree, extrapolate=False)(ReqCoord))
    return np.array(out)


def _spatial_at(N0, N1, SpType, Degree, IntKnotX, IntKnotY, CX, CY):
    """[number of terms][len(CX)]: every spatial term of a basis evaluated at scaled coordinates (CX, CY)."""
    if SpType == 'Polynomial':
        return np.array([CX ** i * CY ** j for i in range(Degree + 1) for j in range(Degree + 1 - i)])
    BX = Create_BSplineBasis_Req(N0, IntKnotX, Degree, CX)
    BY = Create_BSplineBasis_Req(N1, IntKnotY, Degree, CY)
    return np.array([BX[i] * BY[j] for i in range(BX.shape[0]) for j in range(BY.shape[0])])


def _laplacian_iregmat(w0, w1, IGNORE_LAPLACIAN_KERCENT):
    """iREGMAT [Fab][Fab]: the squared 5-point Laplacian of the kernel stamp expressed in the modified-delta basis
    (LAPMAT / LTLMAT / fill_iregmat, BSplineSFFT.py:3640-3686 and :2009-2087), vectorised."""
    L0, L1 = 2 * w0 + 1, 2 * w1 + 1
    Fab = L0 * L1
    rr, cc = np.divmod(np.arange(Fab), L1)
    dist = np.abs(rr[:, None] - rr[None, :]) + np.abs(cc[:, None] - cc[None, :])
    LAP = -(dist == 1).astype(np.int64)
    LAP[np.arange(Fab), np.arange(Fab)] = (dist == 1).sum(axis=1)        # number of 4-neighbours inside the stamp
    if IGNORE_LAPLACIAN_KERCENT:
        for r in ((w0 - 1) * L1 + w1, w0 * L1 + w1 - 1, w0 * L1 + w1, w0 * L1 + w1 + 1, (w0 + 1) * L1 + w1):
            LAP[r, :] = 0
    LTL = LAP.T @ LAP
    c0 = w0 * L1 + w1
    sym = LTL + LTL.T
    colc, rowc = LTL[:, c0], LTL[c0, :]
    ireg = sym - rowc[:, None] - rowc[None, :] - colc[:, None] - colc[None, :] + 2 * LTL[c0, c0]
    edge = colc + rowc - 2 * LTL[c0, c0]
    ireg[:, c0] = edge
    ireg[c0, :] = edge
    ireg[c0, c0] = 2 * LTL[c0, c0]
    return ireg.astype(np.float64)


def _axis_tables(N0, N1, SpType, Degree, IntKnotX, IntKnotY):
    if SpType == 'Polynomial':
        cx = (np.arange(N0, dtype=np.float64) + 1.0) / N0
        cy = (np.arange(N1, dtype=np.float64) + 1.0) / N1
        bx = np.stack([np.ones(N0) if e == 0 else cx ** e for e in range(Degree + 1)])
        by = np.stack([np.ones(N1) if e == 0 else cy ** e for e in range(Degree + 1)])
        # integer powers by repeated multiplication, like the polynomial plan (bit-identical factors)
        for e in range(2, Degree + 1):
            bx[e] = bx[e - 1] * cx
            by[e] = by[e - 1] * cy
        pairs = [(i, j) for i in range(Degree + 1) for j in range(Degree + 1 - i)]
    else:
        bx = Create_BSplineBasis(N0, IntKnotX, Degree)
        by = Create_BSplineBasis(N1, IntKnotY, Degree)
        pairs = [(i, j) for i in range(bx.shape[0]) for j in range(by.shape[0])]
    return bx, by, np.array(pairs, dtype=np.int32)


_PLANS = {}
_PLANS_LOCK = threading.Lock()


class SingleSFFTConfigure:
    @staticmethod
    def SSC(NX, NY, KerHW=8, KerSpType='Polynomial', KerSpDegree=2, KerIntKnotX=[], KerIntKnotY=[],
            SEPARATE_SCALING=True, ScaSpType='Polynomial', ScaSpDegree=0, ScaIntKnotX=[], ScaIntKnotY=[],
            BkgSpType='Polynomial', BkgSpDegree=2, BkgIntKnotX=[], BkgIntKnotY=[],
            REGULARIZE_KERNEL=False, IGNORE_LAPLACIAN_KERCENT=True, XY_REGULARIZE=None, WEIGHT_REGULARIZE=None,
            LAMBDA_REGULARIZE=1e-6, BACKEND_4SUBTRACT='Cupy', MAX_THREADS_PER_BLOCK=8,
            MINIMIZE_GPU_MEMORY_USAGE=False, NUM_CPU_THREADS_4SUBTRACT=8, VERBOSE_LEVEL=2, CUDA_DEVICE_4SUBTRACT=None):
        """Same arguments as the reference (MAX_THREADS_PER_BLOCK, MINIMIZE_GPU_MEMORY_USAGE, NUM_CPU_THREADS_4SUBTRACT are
        accepted and ignored: Greek planes are always streamed here).  Returns SFFTConfig = (SFFTParam_dict, SFFTModule_dict)."""
        N0, N1, w0 = int(NX), int(NY), int(KerHW)
        DK, DB = int(KerSpDegree), int(BkgSpDegree)
        if BACKEND_4SUBTRACT not in ('Cupy', 'HIP'):
            raise Exception("MeLOn ERROR: sfft_amd only provides the GPU backend (BACKEND_4SUBTRACT='Cupy'); there is no CPU path")
        assert DK >= 0 and DB >= 0
        assert KerSpType in ['Polynomial', 'B-Spline']
        assert BkgSpType in ['Polynomial', 'B-Spline']
        if KerSpType == 'B-Spline' and DK == 0:
            assert len(KerIntKnotX) == 0 and len(KerIntKnotY) == 0   # otherwise, discontinuity
        if BkgSpType == 'B-Spline' and DB == 0:
            assert len(BkgIntKnotX) == 0 and len(BkgIntKnotY) == 0   # otherwise, discontinuity
        # SCALING_MODE (BSplineSFFT.py:47-60)
        if not SEPARATE_SCALING:
            SCALING_MODE = 'ENTANGLED'
        elif int(ScaSpDegree) == 0:
            SCALING_MODE = 'SEPARATE-CONSTANT'
        else:
            SCALING_MODE = 'SEPARATE-VARYING'
        if SCALING_MODE == 'SEPARATE-CONSTANT':
            assert DK != 0   # otherwise, reduced to ENTANGLED
        DS = None
        if SEPARATE_SCALING:
            DS = int(ScaSpDegree)
            assert DS >= 0
            assert ScaSpType in ['Polynomial', 'B-Spline']
            if ScaSpType == 'B-Spline' and DS == 0:
                assert len(ScaIntKnotX) == 0 and len(ScaIntKnotY) == 0   # otherwise, discontinuity
        if SCALING_MODE == 'SEPARATE-VARYING':
            # (the reference also insists on MINIMIZE_GPU_MEMORY_USAGE here, :68; Greek planes are always streamed in this build)
            if KerSpType == 'Polynomial' and ScaSpType == 'Polynomial':
                assert DS != DK   # otherwise, reduced to ENTANGLED
            if KerSpType == 'B-Spline' and ScaSpType == 'B-Spline':
                if np.array_equal(np.asarray(KerIntKnotX), np.asarray(ScaIntKnotX)) and \
                        np.array_equal(np.asarray(KerIntKnotY), np.asarray(ScaIntKnotY)):
                    assert DS != DK   # otherwise, reduced to ENTANGLED
        if REGULARIZE_KERNEL:
            assert XY_REGULARIZE is not None
            XY_REGULARIZE = np.asarray(XY_REGULARIZE, dtype=np.float64)
            assert len(XY_REGULARIZE.shape) == 2 and XY_REGULARIZE.shape[1] == 2

        kbx, kby, kpairs = _axis_tables(N0, N1, KerSpType, DK, KerIntKnotX, KerIntKnotY)
        tbx, tby, bpairs = _axis_tables(N0, N1, BkgSpType, DB, BkgIntKnotX, BkgIntKnotY)
        sbx = sby = spairs = None
        if SCALING_MODE == 'ENTANGLED':
            mode = 0
        elif SCALING_MODE == 'SEPARATE-CONSTANT':
            mode = 1 if KerSpType == 'Polynomial' else 2    # TweakLS: delete vs. sum the ij00 rows/columns (:2171-2272)
        else:
            mode = 3
            sbx, sby, spairs = _axis_tables(N0, N1, ScaSpType, DS, ScaIntKnotX, ScaIntKnotY)
            assert len(spairs) <= len(kpairs)               # ScaFij <= Fij (:190)
        if CUDA_DEVICE_4SUBTRACT is None:
            device = torch.cuda.current_device()
        else:
            device = int(CUDA_DEVICE_4SUBTRACT)
        key = (device, N0, N1, w0, KerSpType, DK, tuple(KerIntKnotX), tuple(KerIntKnotY), BkgSpType, DB,
               tuple(BkgIntKnotX), tuple(BkgIntKnotY), mode,
               (ScaSpType, DS, tuple(ScaIntKnotX), tuple(ScaIntKnotY)) if mode == 3 else None)
        with _PLANS_LOCK:
            plan = _PLANS.get(key)
            if plan is None:
                if VERBOSE_LEVEL in [1, 2]:
                    print('\n --//--//--//--//-- TRIGGER SFFT COMPILATION [HIP] --//--//--//--//-- ')
                bdict = dict(kbx=kbx, kby=kby, ker_pairs=kpairs, tbx=tbx, tby=tby, bkg_pairs=bpairs, scaling_mode=mode)
                if mode == 3:
                    bdict.update(sbx=sbx, sby=sby, sca_pairs=spairs)
                plan = Plan(N0, N1, w0, device=device, basis=bdict)
                if len(_PLANS) >= 3:
                    _PLANS.pop(next(iter(_PLANS)))
                _PLANS[key] = plan

        L0 = L1 = 2 * w0 + 1
        Fab = L0 * L1
        Fij, Fpq = len(kpairs), len(bpairs)

        # kernel regularisation belongs to THIS config, not to the plan (plans are shared between configs of one geometry):
        # the small matrices are kept in the module dictionary and put on the plan right before every solve made through
        # this config (sfftcore.SFFTSubtract._bound_plan), as the reference derives them per config at ESS time
        if REGULARIZE_KERNEL:
            NREG = XY_REGULARIZE.shape[0]
            CX_REG, CY_REG = XY_REGULARIZE[:, 0] / N0, XY_REGULARIZE[:, 1] / N1
            SPMAT = _spatial_at(N0, N1, KerSpType, DK, KerIntKnotX, KerIntKnotY, CX_REG, CY_REG)
            if WEIGHT_REGULARIZE is None:
                WS = np.full(NREG, 1.0 / NREG)
            else:
                WS = np.asarray(WEIGHT_REGULARIZE, dtype=np.float64) / np.sum(WEIGHT_REGULARIZE)   # unit sum
            SST = (SPMAT * WS) @ SPMAT.T
            CSST = DSST = None
            if mode == 3:
                ScaSPMAT = _spatial_at(N0, N1, ScaSpType, DS, ScaIntKnotX, ScaIntKnotY, CX_REG, CY_REG)
                if ScaSPMAT.shape[0] < Fij:     # zero place-holders
                    ScaSPMAT = np.concatenate((ScaSPMAT, np.zeros((Fij - ScaSPMAT.shape[0], NREG))), axis=0)
                CSST = (SPMAT * WS) @ ScaSPMAT.T
                DSST = (ScaSPMAT * WS) @ ScaSPMAT.T
            regularization = (float(LAMBDA_REGULARIZE), _laplacian_iregmat(w0, w0, IGNORE_LAPLACIAN_KERCENT), SST, CSST, DSST)
        else:
            regularization = (0.0,)
        P = {}
        P['KerHW'], P['KerSpType'], P['KerSpDegree'] = KerHW, KerSpType, KerSpDegree
        P['KerIntKnotX'], P['KerIntKnotY'] = KerIntKnotX, KerIntKnotY
        P['SEPARATE_SCALING'], P['SCALING_MODE'] = SEPARATE_SCALING, SCALING_MODE
        if SEPARATE_SCALING:
            P['ScaSpType'], P['ScaSpDegree'], P['ScaIntKnotX'], P['ScaIntKnotY'], P['DS'] = ScaSpType, ScaSpDegree, ScaIntKnotX, ScaIntKnotY, DS
        P['REGULARIZE_KERNEL'], P['IGNORE_LAPLACIAN_KERCENT'] = REGULARIZE_KERNEL, IGNORE_LAPLACIAN_KERCENT
        P['XY_REGULARIZE'], P['WEIGHT_REGULARIZE'], P['LAMBDA_REGULARIZE'] = XY_REGULARIZE, WEIGHT_REGULARIZE, LAMBDA_REGULARIZE
        P['BkgSpType'], P['BkgSpDegree'], P['BkgIntKnotX'], P['BkgIntKnotY'] = BkgSpType, BkgSpDegree, BkgIntKnotX, BkgIntKnotY
        P['N0'], P['N1'], P['w0'], P['w1'], P['DK'], P['DB'] = N0, N1, w0, w0, DK, DB
        P['SCALE'], P['SCALE_L'] = np.float64(1 / (N0 * N1)), np.float64(N0 * N1)
        P['L0'], P['L1'], P['Fab'] = L0, L1, Fab
        P['Fi'] = kbx.shape[0] if KerSpType == 'B-Spline' else -1
        P['Fj'] = kby.shape[0] if KerSpType == 'B-Spline' else -1
        P['Fp'] = tbx.shape[0] if BkgSpType == 'B-Spline' else -1
        P['Fq'] = tby.shape[0] if BkgSpType == 'B-Spline' else -1
        P['Fij'], P['Fpq'], P['Fijab'] = Fij, Fpq, Fij * Fab
        P['FOMG'], P['FGAM'], P['FTHE'] = Fij ** 2, Fij * Fpq, Fij
        P['FPSI'], P['FPHI'], P['FDEL'] = Fpq * Fij, Fpq ** 2, Fpq
        P['NEQ'] = Fij * Fab + Fpq
        P['NEQt'] = P['NEQ']
        if mode in (1, 2):
            P['NEQt'] = P['NEQ'] - Fij + 1
        if mode == 3:
            P['ScaFi'] = sbx.shape[0] if ScaSpType == 'B-Spline' else -1
            P['ScaFj'] = sby.shape[0] if ScaSpType == 'B-Spline' else -1
            P['ScaFij'] = len(spairs)
            P['NEQt'] = P['NEQ'] - (Fij - P['ScaFij'])
        P['ConstPhotRatio'] = mode in (1, 2)
        return (P, {'plan': plan, 'backend': 'HIP', 'regularization': regularization})


class BSpline_Packet:
    @staticmethod
    def BSP(FITS_REF, FITS_SCI, FITS_mREF, FITS_mSCI, FITS_DIFF=None, FITS_Solution=None,
            ForceConv='REF', GKerHW=8, KerSpType='Polynomial', KerSpDegree=2, KerIntKnotX=[], KerIntKnotY=[],
            SEPARATE_SCALING=True, ScaSpType='Polynomial', ScaSpDegree=0, ScaIntKnotX=[], ScaIntKnotY=[],
            BkgSpType='Polynomial', BkgSpDegree=2, BkgIntKnotX=[], BkgIntKnotY=[],
            REGULARIZE_KERNEL=False, IGNORE_LAPLACIAN_KERCENT=True, XY_REGULARIZE=None,
            WEIGHT_REGULARIZE=None, LAMBDA_REGULARIZE=1e-6, BACKEND_4SUBTRACT='Cupy',
            CUDA_DEVICE_4SUBTRACT='0', MAX_THREADS_PER_BLOCK=8, MINIMIZE_GPU_MEMORY_USAGE=False,
            NUM_CPU_THREADS_4SUBTRACT=8, VERBOSE_LEVEL=2):
        """FITS paths in, (Solution, PixA_DIFF) out; same parameters and conventions as the reference
        (BSplineSFFT.py:3969-4356).  ForceConv='REF' -> DIFF = SCI - Conv(REF); 'SCI' -> DIFF = Conv(SCI) - REF."""
        def read_T(path):
            a = (_afits.getdata(path, ext=0) if _afits is not None else minifits.getdata(path)[0]).T
            return np.ascontiguousarray(a, dtype=np.float64)
        PixA_REF, PixA_SCI = read_T(FITS_REF), read_T(FITS_SCI)
        PixA_mREF, PixA_mSCI = read_T(FITS_mREF), read_T(FITS_mSCI)
        NaNmask_U = union_nan_mask(np, PixA_REF, PixA_SCI)
        assert np.sum(np.isnan(PixA_mREF)) == 0
        assert np.sum(np.isnan(PixA_mSCI)) == 0
        ConvdSide, KerHW = ForceConv, GKerHW
        torch.cuda.set_device(int(CUDA_DEVICE_4SUBTRACT))
        if VERBOSE_LEVEL in [0, 1, 2]:
            print('MeLOn CheckPoint: TRIGGER Function Compilations of SFFT-SUBTRACTION!')
        Tcomp_start = time.time()
        SFFTConfig = SingleSFFTConfigure.SSC(
            NX=PixA_REF.shape[0], NY=PixA_REF.shape[1], KerHW=KerHW, KerSpType=KerSpType, KerSpDegree=KerSpDegree,
            KerIntKnotX=KerIntKnotX, KerIntKnotY=KerIntKnotY, SEPARATE_SCALING=SEPARATE_SCALING, ScaSpType=ScaSpType,
            ScaSpDegree=ScaSpDegree, ScaIntKnotX=ScaIntKnotX, ScaIntKnotY=ScaIntKnotY, BkgSpType=BkgSpType,
            BkgSpDegree=BkgSpDegree, BkgIntKnotX=BkgIntKnotX, BkgIntKnotY=BkgIntKnotY, REGULARIZE_KERNEL=REGULARIZE_KERNEL,
            IGNORE_LAPLACIAN_KERCENT=IGNORE_LAPLACIAN_KERCENT, XY_REGULARIZE=XY_REGULARIZE, WEIGHT_REGULARIZE=WEIGHT_REGULARIZE,
            LAMBDA_REGULARIZE=LAMBDA_REGULARIZE, BACKEND_4SUBTRACT=BACKEND_4SUBTRACT, VERBOSE_LEVEL=VERBOSE_LEVEL,
            CUDA_DEVICE_4SUBTRACT=int(CUDA_DEVICE_4SUBTRACT))
        if VERBOSE_LEVEL in [1, 2]:
            print('\nMeLOn Report: FUNCTION COMPILATIONS OF SFFT-SUBTRACTION TAKES [%.3f s] \n' % (time.time() - Tcomp_start))
        PixA_I, PixA_J, PixA_mI, PixA_mJ = assign_roles(np, PixA_REF, PixA_SCI, PixA_mREF, PixA_mSCI, ConvdSide, NaNmask_U)
        Tsub_start = time.time()
        Solution, PixA_DIFF = GeneralSFFTSubtract.GSS(PixA_I=PixA_I, PixA_J=PixA_J, PixA_mI=PixA_mI, PixA_mJ=PixA_mJ,
                                                     SFFTConfig=SFFTConfig, ContamMask_I=None, VERBOSE_LEVEL=VERBOSE_LEVEL)[:2]
        if VERBOSE_LEVEL in [1, 2]:
            print('\nMeLOn Report: SFFT-SUBTRACTION TAKES [%.3f s] \n' % (time.time() - Tsub_start))
        PixA_DIFF = finish_diff(PixA_DIFF, ConvdSide, NaNmask_U)

        kw = [('NAME_REF', pa.basename(FITS_REF)), ('NAME_SCI', pa.basename(FITS_SCI)), ('BEND4SUB', BACKEND_4SUBTRACT),
              ('CONVD', ConvdSide), ('KERHW', KerHW), ('KSPTYPE', str(KerSpType)), ('KSPDEG', KerSpDegree),
              ('NKIKX', len(KerIntKnotX))] + [('KIKX%d' % i, k) for i, k in enumerate(KerIntKnotX)] + \
             [('NKIKY', len(KerIntKnotY))] + [('KIKY%d' % i, k) for i, k in enumerate(KerIntKnotY)] + \
             [('SEPSCA', str(SEPARATE_SCALING))]
        if SEPARATE_SCALING:
            kw += [('SSPTYPE', str(ScaSpType)), ('SSPDEG', ScaSpDegree), ('NSIKX', len(ScaIntKnotX))] + \
                  [('SIKX%d' % i, k) for i, k in enumerate(ScaIntKnotX)] + [('NSIKY', len(ScaIntKnotY))] + \
                  [('SIKY%d' % i, k) for i, k in enumerate(ScaIntKnotY)]
        kw += [('BSPTYPE', str(BkgSpType)), ('BSPDEG', BkgSpDegree), ('NBIKX', len(BkgIntKnotX))] + \
              [('BIKX%d' % i, k) for i, k in enumerate(BkgIntKnotX)] + [('NBIKY', len(BkgIntKnotY))] + \
              [('BIKY%d' % i, k) for i, k in enumerate(BkgIntKnotY)] + \
              [('REGKER', str(REGULARIZE_KERNEL)), ('ILKC', str(IGNORE_LAPLACIAN_KERCENT)),
               ('NREG', -1 if XY_REGULARIZE is None else int(np.asarray(XY_REGULARIZE).shape[0])),
               ('REGW', 'UNIFORM' if WEIGHT_REGULARIZE is None else 'SPECIFIED'), ('REGLAMB', LAMBDA_REGULARIZE)]
        if FITS_DIFF is not None:
            if _afits is not None:
                with _afits.open(FITS_SCI) as hdl:
                    hdl[0].data[:, :] = PixA_DIFF.T
                    for k, v in kw:
                        hdl[0].header[k] = (v, 'SFFT')
                    hdl.writeto(FITS_DIFF, overwrite=True)
            else:
                data0, cards = minifits.getdata(FITS_SCI)
                for k, v in kw:
                    minifits.set_card(cards, k, v, 'SFFT')
                out = PixA_DIFF.T
                if data0.dtype.kind == 'f':
                    out = out.astype(data0.dtype.newbyteorder('='))
                minifits.writeto(FITS_DIFF, np.ascontiguousarray(out), cards)
        if FITS_Solution is not None:
            P = SFFTConfig[0]
            # key set and order of BSplineSFFT.py:4282-4351
            skw = kw[:2] + kw[2:] + [('N0', P['N0']), ('N1', P['N1']), ('W0', P['w0']), ('W1', P['w1']), ('DK', P['DK']), ('DB', P['DB'])]
            if SEPARATE_SCALING:
                skw += [('DS', P['DS'])]
            skw += [('L0', P['L0']), ('L1', P['L1']), ('FAB', P['Fab']), ('FI', P['Fi']), ('FJ', P['Fj']), ('FIJ', P['Fij']),
                    ('FP', P['Fp']), ('FQ', P['Fq']), ('FPQ', P['Fpq'])]
            if SEPARATE_SCALING and ScaSpDegree > 0:
                skw += [('SCAFI', P['ScaFi']), ('SCAFJ', P['ScaFj']), ('SCAFIJ', P['ScaFij'])]
            skw += [('FIJAB', P['Fijab']), ('NEQ', P['NEQ']), ('NEQT', P['NEQt'])]
            if _afits is not None:
                phdu = _afits.PrimaryHDU()
                for k, v in skw:
                    phdu.header[k] = (v, 'SFFT')
                phdu.data = Solution.reshape((-1, 1)).T
                _afits.HDUList([phdu]).writeto(FITS_Solution, overwrite=True)
            else:
                cards = []
                for k, v in skw:
                    minifits.set_card(cards, k, v, 'SFFT')
                minifits.writeto(FITS_Solution, np.ascontiguousarray(Solution.reshape((-1, 1)).T), cards)
        return Solution, PixA_DIFF


# ------------------------------------------------------------------------------------------------
# Post-subtraction classes of sfft/BSplineSFFT.py.  The solution decoding and the kernel realisation are
# host numpy in the reference too; the decorrelation kernel and the grid convolution run on the GPU.
# ------------------------------------------------------------------------------------------------
def _scaling_mode(SEPARATE_SCALING, DS):
    if not SEPARATE_SCALING:
        return 'ENTANGLED'
    return 'SEPARATE-CONSTANT' if DS == 0 else 'SEPARATE-VARYING'


def _term_list(SpType, Degree, Fi, Fj):
    """REF_ij of a basis: (i, j) with i + j <= Degree, or the full Fi x Fj tensor grid."""
    if SpType == 'Polynomial':
        return [(i, j) for i in range(Degree + 1) for j in range(Degree + 1 - i)]
    return [(i, j) for i in range(Fi) for j in range(Fj)]


class Read_SFFTSolution:
    """Solution vector -> dictionaries of the SFFT-format coefficients ac_ijab = a_ijab / (N0 N1) (BSplineSFFT.py:4358-4553):
    SfftKerDict[(i, j)] is the [L0][L1] stamp of spatial term (i, j); with separately varying scaling its centre entry is NaN
    and SfftScaDict[(i, j)] holds the coefficient of scaling term (i, j) (otherwise SfftScaDict is None)."""

    def FromArray(self, Solution, KerSpType, N0, N1, DK, L0, L1, Fi, Fj, Fpq, SEPARATE_SCALING, ScaSpType, DS, ScaFi, ScaFj):
        mode = _scaling_mode(SEPARATE_SCALING, DS)
        w0, w1 = (L0 - 1) // 2, (L1 - 1) // 2
        kterms = _term_list(KerSpType, DK, Fi, Fj)
        ac = (np.asarray(Solution, dtype=np.float64)[:-Fpq] / (N0 * N1)).reshape(len(kterms), L0, L1)
        SfftKerDict = {t: ac[k].copy() for k, t in enumerate(kterms)}
        if mode != 'SEPARATE-VARYING':
            return SfftKerDict, None
        sterms = _term_list(ScaSpType, DS, ScaFi, ScaFj)       # the first ScaFij kernel slots carry the scaling terms
        SfftScaDict = {t: float(ac[k, w0, w1]) for k, t in enumerate(sterms)}
        for t in kterms:
            SfftKerDict[t][w0, w1] = np.nan
        return SfftKerDict, SfftScaDict

    def FromFITS(self, FITS_Solution):
        Solution, phdr = _read_solution_fits(FITS_Solution)
        a = _solution_header_args(phdr)
        return self.FromArray(Solution=Solution, KerSpType=a['KerSpType'], N0=a['N0'], N1=a['N1'], DK=a['DK'], L0=a['L0'], L1=a['L1'],
                              Fi=a['Fi'], Fj=a['Fj'], Fpq=a['Fpq'], SEPARATE_SCALING=a['SEPARATE_SCALING'], ScaSpType=a['ScaSpType'],
                              DS=a['DS'], ScaFi=a['ScaFi'], ScaFj=a['ScaFj'])


def _read_solution_fits(FITS_Solution):
    if _afits is not None:
        return _afits.getdata(FITS_Solution, ext=0)[0], _afits.getheader(FITS_Solution, ext=0)
    data, cards = minifits.getdata(FITS_Solution)
    return np.asarray(data, dtype=np.float64)[0], minifits.header_dict(cards)


def _solution_header_args(phdr):
    """The keywords BSP writes next to the solution (BSplineSFFT.py:4282-4351) back into FromArray's arguments."""
    a = dict(KerHW=phdr['KERHW'], KerSpType=phdr['KSPTYPE'], N0=int(phdr['N0']), N1=int(phdr['N1']), DK=int(phdr['DK']),
             L0=int(phdr['L0']), L1=int(phdr['L1']), Fi=int(phdr['FI']), Fj=int(phdr['FJ']), Fpq=int(phdr['FPQ']),
             KerIntKnotX=[phdr['KIKX%d' % i] for i in range(int(phdr['NKIKX']))],
             KerIntKnotY=[phdr['KIKY%d' % i] for i in range(int(phdr['NKIKY']))],
             SEPARATE_SCALING=str(phdr['SEPSCA']) == 'True', ScaSpType=None, DS=None, ScaIntKnotX=None, ScaIntKnotY=None,
             ScaFi=None, ScaFj=None)
    if a['SEPARATE_SCALING']:
        a['ScaSpType'], a['DS'] = phdr['SSPTYPE'], int(phdr['SSPDEG'])
        a['ScaIntKnotX'] = [phdr['SIKX%d' % i] for i in range(int(phdr['NSIKX']))]
        a['ScaIntKnotY'] = [phdr['SIKY%d' % i] for i in range(int(phdr['NSIKY']))]
        if a['DS'] > 0:
            a['ScaFi'], a['ScaFj'] = int(phdr['SCAFI']), int(phdr['SCAFJ'])
    return a


class BSpline_MatchingKernel:
    """Matching-kernel stamps [NPOINT][L0][L1] (standard delta basis) realised at the coordinates XY_q (FortranCoor)
    (BSplineSFFT.py:4555-4723)."""

    def __init__(self, XY_q, VERBOSE_LEVEL=2):
        self.XY_q = XY_q
        self.VERBOSE_LEVEL = VERBOSE_LEVEL

    def FromArray(self, Solution, KerSpType, KerIntKnotX, KerIntKnotY, N0, N1, DK, L0, L1, Fi, Fj, Fpq,
                  SEPARATE_SCALING, ScaSpType, ScaIntKnotX, ScaIntKnotY, DS, ScaFi, ScaFj):
        sXY = np.array(self.XY_q, dtype=np.float64)
        CX, CY = sXY[:, 0] / N0, sXY[:, 1] / N1                                 # ScaledFortranCoor
        SfftKerDict, SfftScaDict = Read_SFFTSolution().FromArray(
            Solution=Solution, KerSpType=KerSpType, N0=N0, N1=N1, DK=DK, L0=L0, L1=L1, Fi=Fi, Fj=Fj, Fpq=Fpq,
            SEPARATE_SCALING=SEPARATE_SCALING, ScaSpType=ScaSpType, DS=DS, ScaFi=ScaFi, ScaFj=ScaFj)
        w0, w1 = (L0 - 1) // 2, (L1 - 1) // 2
        KerBASE = _spatial_at(N0, N1, KerSpType, DK, KerIntKnotX, KerIntKnotY, CX, CY)            # [Fij][NPOINT]
        KerCOEFF = np.array([SfftKerDict[t] for t in _term_list(KerSpType, DK, Fi, Fj)])          # [Fij][L0][L1]
        KerStack = np.tensordot(KerBASE, KerCOEFF, (0, 0))                                        # [NPOINT][L0][L1]
        # modified delta basis -> pixels: the centre pixel is (scaling) - (sum of the off-centre pixels)
        if SfftScaDict is None:
            centre = KerStack[:, w0, w1].copy()
            KerStack[:, w0, w1] = centre - (np.sum(KerStack, axis=(1, 2)) - centre)
        else:
            ScaBASE = _spatial_at(N0, N1, ScaSpType, DS, ScaIntKnotX, ScaIntKnotY, CX, CY)        # [ScaFij][NPOINT]
            ScaCOEFF = np.array([SfftScaDict[t] for t in _term_list(ScaSpType, DS, ScaFi, ScaFj)])
            KerStack[:, w0, w1] = ScaCOEFF @ ScaBASE - np.nansum(KerStack, axis=(1, 2))            # centre entries are NaN here
        return KerStack

    def FromFITS(self, FITS_Solution):
        Solution, phdr = _read_solution_fits(FITS_Solution)
        a = _solution_header_args(phdr)
        if self.VERBOSE_LEVEL in [1, 2]:
            print('\n --//--//--//--//-- SFFT CONFIGURATION --//--//--//--//-- ')
            print('\n ---//--- %s Kernel | KerSpDegree %d | KerHW %d ---//---' % (a['KerSpType'], a['DK'], a['KerHW']))
        return self.FromArray(Solution=Solution, KerSpType=a['KerSpType'], KerIntKnotX=a['KerIntKnotX'], KerIntKnotY=a['KerIntKnotY'],
                              N0=a['N0'], N1=a['N1'], DK=a['DK'], L0=a['L0'], L1=a['L1'], Fi=a['Fi'], Fj=a['Fj'], Fpq=a['Fpq'],
                              SEPARATE_SCALING=a['SEPARATE_SCALING'], ScaSpType=a['ScaSpType'], ScaIntKnotX=a['ScaIntKnotX'],
                              ScaIntKnotY=a['ScaIntKnotY'], DS=a['DS'], ScaFi=a['ScaFi'], ScaFj=a['ScaFj'])


class ConvKernel_Convertion:
    """CSZ / iCSZ as in BSplineSFFT.py:4725-4753 (iCSZ returns the lost weight as well, unlike sfft/utils/ConvKernelConvertion.py)."""

    def CSZ(ConvKernel, N0, N1):
        L0, L1 = ConvKernel.shape
        out = np.zeros((N0, N1), dtype=np.result_type(ConvKernel, np.float64))
        out[:L0, :L1] = ConvKernel
        return np.roll(out, (-((L0 - 1) // 2), -((L1 - 1) // 2)), axis=(0, 1))

    def iCSZ(KIMG, L0, L1):
        back = np.roll(KIMG, ((L0 - 1) // 2, (L1 - 1) // 2), axis=(0, 1))
        ConvKernel = back[:L0, :L1]
        return ConvKernel, 1.0 - np.sum(np.abs(ConvKernel)) / np.sum(np.abs(back))


class BSpline_DeCorrelation:
    @staticmethod
    def BDC(MK_JLst, SkySig_JLst, MK_ILst=[], SkySig_ILst=[], MK_Fin=None, KERatio=2.0, DENO_CLIP_RATIO=100000.0, VERBOSE_LEVEL=2,
            CUDA_DEVICE=None):
        """Noise-decorrelation kernel from realised matching kernels (BSplineSFFT.py:4755-4868): as DeCorrelation_Calculator.DCC
        plus a floor of max / DENO_CLIP_RATIO on the Fourier-space denominator.  Host arrays in, host array out; the
        transforms run on the GPU (sfft_fft2_r2c / sfft_ifft2_c2r)."""
        import math
        from .utils.PureCupyDeCorrelationCalculator import PureCupy_DeCorrelation_Calculator
        NumI, NumJ = len(MK_ILst), len(MK_JLst)
        if NumI == 0:
            if NumJ < 2:
                raise Exception('MeLOn ERROR: %s' % 'Image-Stacking Mode requires at least 2 J-images!')
            if np.sum([MKj is not None for MKj in MK_JLst]) == 0:
                raise Exception('MeLOn ERROR: %s' % 'Image-Stacking Mode requires at least 1 not-None J-kernel!')
            MK_Queue = list(MK_JLst)
        else:
            if NumJ == 0:
                raise Exception('MeLOn ERROR: %s' % 'Image-Subtraction Mode requires at least 1 I-image & 1 J-image!')
            if np.sum([MK is not None for MK in list(MK_JLst) + list(MK_ILst) + [MK_Fin]]) == 0:
                raise Exception('MeLOn ERROR: %s' % 'Image-Subtraction Mode requires at least 1 not-None J/I/Fin-kernel!')
            MK_Queue = list(MK_JLst) + [MK_Fin] + list(MK_ILst)
        shapes = np.array([MK.shape for MK in MK_Queue if MK is not None])
        L_KDeCo = [int(round(KERatio * shapes[:, ax].max())) for ax in (0, 1)]
        L_KDeCo = [L + 1 if L % 2 == 0 else L for L in L_KDeCo]
        if VERBOSE_LEVEL in [1, 2]:
            print('MeLOn CheckPoint: %s' % ('DeCorrelation Kernel with size [%d, %d]' % tuple(L_KDeCo)))
        N0, N1 = [2 ** (math.ceil(np.log2(shapes[:, ax].max())) + 1) for ax in (0, 1)]   # "trivial image size" (:4805-4806)
        dev = torch.device("cuda", torch.cuda.current_device() if CUDA_DEVICE is None else int(CUDA_DEVICE))
        tg = lambda K: None if K is None else torch.from_numpy(np.ascontiguousarray(K, dtype=np.float64)).to(dev)
        KDeCo = PureCupy_DeCorrelation_Calculator.PCDC(
            NX_IMG=N0, NY_IMG=N1, KERNEL_GPU_JQueue=[tg(K) for K in MK_JLst], BKGSIG_JQueue=list(SkySig_JLst),
            KERNEL_GPU_IQueue=[tg(K) for K in MK_ILst], BKGSIG_IQueue=list(SkySig_ILst), MATCH_KERNEL_GPU=tg(MK_Fin),
            REAL_OUTPUT=True, REAL_OUTPUT_SIZE=tuple(L_KDeCo), NORMALIZE_OUTPUT=True, VERBOSE_LEVEL=VERBOSE_LEVEL,
            CUDA_DEVICE=dev.index, DENO_CLIP_RATIO=DENO_CLIP_RATIO)
        return KDeCo.cpu().numpy()


class BSpline_GridConvolve:
    """Grid-wise space-varying convolution (BSplineSFFT.py:4870-5008): AllocatedL labels compact boxes of PixA_obj, KerStack[label]
    is the kernel of each box.  Only the GPU variant exists here (`sfft_grid_convolve`: direct sums in fp64, which is what both
    branches of the reference's GSVC_GPU compute -- `use_fft` only chooses how)."""

    def __init__(self, PixA_obj, AllocatedL, KerStack, nan_fill_value=0.0, use_fft=False, normalize_kernel=True):
        PixA_in = np.array(PixA_obj, dtype=np.float64)
        PixA_in[np.isnan(PixA_in)] = nan_fill_value
        self.PixA_in = PixA_in
        self.AllocatedL = AllocatedL
        self.KerStack = KerStack
        self.use_fft = use_fft
        self.normalize_kernel = normalize_kernel

    def GSVC_CPU(self, nproc=32):
        raise Exception("MeLOn ERROR: sfft_amd only provides the GPU variant (GSVC_GPU); there is no CPU path")

    def GSVC_GPU(self, CUDA_DEVICE='0', CLEAN_GPU_MEMORY=False, nproc=32):
        import ctypes
        from . import _lib
        dev = torch.device("cuda", int(CUDA_DEVICE))
        N0, N1 = self.PixA_in.shape
        KerStack = np.asarray(self.KerStack, dtype=np.float64)
        Nseg, L0, L1 = KerStack.shape
        if self.normalize_kernel:
            KerStack = KerStack / np.sum(KerStack, axis=(1, 2))[:, np.newaxis, np.newaxis]
        lab = np.ascontiguousarray(self.AllocatedL, dtype=np.int32)
        assert lab.shape == (N0, N1) and lab.min() >= 0 and lab.max() < Nseg
        d_in = torch.from_numpy(np.ascontiguousarray(self.PixA_in)).to(dev)
        d_lab = torch.from_numpy(lab).to(dev)
        d_ker = torch.from_numpy(np.ascontiguousarray(KerStack)).to(dev)
        d_out = torch.empty((N0, N1), dtype=torch.float64, device=dev)
        _lib.check(_lib.lib().sfft_grid_convolve(d_in.data_ptr(), d_lab.data_ptr(), d_ker.data_ptr(), N0, N1, Nseg, L0, L1,
                                                 d_out.data_ptr(), dev.index,
                                                 ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return d_out.cpu().numpy()
