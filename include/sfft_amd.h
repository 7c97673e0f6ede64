/*
 * sfft_amd.h -- C ABI of the MI355X-native SFFT subtraction core (libsfft_amd.so).
 *
 * This is the drop-in boundary for the hot path of thomasvrussell/sfft behind
 * sfft.Customized_Packet.CP / sfft.PureCupy_Customized_Packet.PCCP -> sfft.sfftcore.
 * The reference has no native FFI of its own on this path (its 17 CUDA kernels are C
 * strings JIT-compiled through cupy.RawModule, sfft/sfftcore/SFFTConfigure.py:106-808),
 * so each entry point below cites the reference *Python* interface it replaces.
 *
 * Conventions
 *   - plain C types only: pointers, sizes, ints; no torch / hip types in signatures.
 *   - image pointers are DEVICE pointers to float64, C order, shape [N0][N1], axis 0 = FITS
 *     NAXIS1 (the packets transpose on read, sfft/CustomizedPacket.py:93); NaN-free.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - the caller allocates every output; the plan owns all workspaces.
 *   - every function returns SFFT_OK (0) or a negative SFFT_ERR_* code; a human readable
 *     message for the last failure on the calling thread is returned by sfft_last_error().
 *   - one plan per (device, N0, N1, KerHW, DK, DB, ConstPhotRatio); a plan may be used by
 *     one host thread at a time; no global state.
 */
#ifndef SFFT_AMD_H
#define SFFT_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sfft_plan sfft_plan;

enum {
    SFFT_OK = 0,
    SFFT_ERR_INVALID_ARG = -1,      /* bad DK/DB/size: Python shim raises the reference's 'MeLOn ERROR' texts */
    SFFT_ERR_UNSUPPORTED_SIZE = -2, /* an image side above 16384 with no factorisation into on-chip transforms, or more than 24 576 unknowns */
    SFFT_ERR_HIP = -3,              /* a HIP runtime call failed (message has the HIP error string) */
    SFFT_ERR_SINGULAR = -4,         /* linear system could not be solved (the pivoted LU found no nonzero pivot in a column) */
    SFFT_ERR_NOMEM = -5,
    SFFT_ERR_STALL = -6             /* the pivoted LU's workgroups did not get through a hand-off twice in a row (bounded polls ran out: the GPU was too
                                       busy to keep them co-resident); nothing is known about the system -- call again.  Never raised for a singular system */
};

/* fields for sfft_plan_query(); the dict keys callers read from SFFTConfig[0]
 * (sfft/sfftcore/SFFTConfigure.py:50-75, read at sfft/CustomizedPacket.py:207-217) */
enum {
    SFFT_Q_N0 = 0, SFFT_Q_N1, SFFT_Q_W0, SFFT_Q_W1, SFFT_Q_DK, SFFT_Q_DB, SFFT_Q_CONSTPHOTRATIO,
    SFFT_Q_L0, SFFT_Q_L1, SFFT_Q_FAB, SFFT_Q_FIJ, SFFT_Q_FPQ, SFFT_Q_NEQ, SFFT_Q_FIJAB, SFFT_Q_NEQ_FSFREE,
    SFFT_Q_FOMG, SFFT_Q_FGAM, SFFT_Q_FTHE, SFFT_Q_FPSI, SFFT_Q_FPHI, SFFT_Q_FDEL,
    SFFT_Q_WORKSPACE_BYTES,         /* device memory owned by the plan */
    SFFT_Q_LAST_SOLVER,             /* 1 = Cholesky, 2 = LU fallback, for the most recent solve */
    SFFT_Q_NUM_GREEK_PAIRS,         /* spectral products actually transformed per solve */
    SFFT_Q_SCAFIJ,                  /* scaling terms of a plan made by sfft_plan_create_varscale (0 otherwise) */
    SFFT_Q_SOLVE_GRAPH,             /* 1 once the factorisation chain of this plan has been captured and replays as a hipGraph */
    SFFT_Q_THETA_FUSED,             /* 1: the Theta passes ride in the Omega launch of this plan (stage GREEK_G1 then carries them) */
    SFFT_Q_OMG_OFFDIAG,             /* Omega products I_a x conj(I_b), a < b, that are transformed (all Fij (Fij - 1) / 2 of them) */
    SFFT_Q_OMG_DIAG,                /* the same for a = b */
    SFFT_Q_G1_DECIMATED,            /* 1: the Omega launch of this plan takes a radix-2 decimation step along the rows (half the matrix instructions) */
    SFFT_Q_G1_CHUNKS,               /* row chunks of the Greek stage-1 launches: each writes one partial lag sum per pass, lag and spectrum column */
    SFFT_Q_G1_MFMA,                 /* 1: the Omega passes of this plan run on the matrix cores (lag half-width 9 .. 16), 0: on the vector kernel */
    SFFT_Q_CHOL_DATAFLOW,           /* 1: the Cholesky factorisation of this plan is the single-launch dataflow kernel, 0: the launch chain */
    SFFT_Q_SOLVER_N,                /* unknowns of the system the factorisation receives (NEQ when nothing is removed or tied; NEQ_FSFREE is the
                                       reference's dictionary entry, defined whether or not the stripes are removed) */
    SFFT_Q_OMG_SPARSE,              /* Omega products of this plan that are summed in real space (tabulated bases: terms with disjoint supports
                                       along one axis), not through the transforms; of FOMG / Fab^2 = Fij (Fij + 1) / 2 products */
    SFFT_Q_CHOL_STATUS,             /* status bits the most recent Cholesky attempt left (0 = factorised): 1 | 2 = a pivot was not positive (the
                                       reference's LU, SFFTSubtract.py:15-23, then takes the system), 4 = a dataflow hand-off poll ran out (a
                                       scheduling stall, not a property of the system), 8 = launched on fewer than two workgroups */
    SFFT_Q_SOLVES,                  /* dense solves this plan has completed since it was created (cumulative; the three counters let a caller that
                                       keeps several pairs in flight say what its timed region ran on: difference before / after) */
    SFFT_Q_LU_FALLBACKS,            /* ... of which the Cholesky attempt failed and the pivoted LU redid the system (forced LU runs are not counted) */
    SFFT_Q_CHOL_STALLS,             /* ... Cholesky attempts that ended with status bit 4 (a hand-off poll ran out: a scheduling stall) */
    SFFT_Q_COUNT
};

/* stage ids for sfft_stage_ms(); letters follow the reference's VERBOSE_LEVEL=2 printout
 * (sfft/sfftcore/SFFTSubtract.py:594-597, 763-769, 814-816) */
enum {
    SFFT_ST_PRELIM_SOLVE = 0,  /* b+c: spatial polynomial + forward DFTs of the masked pair */
    SFFT_ST_GREEK_G1,          /* d: Omega passes -- Hadamard products + pruned column transform (kernel greek_g1, 2w lags) */
    SFFT_ST_GREEK_G2,          /* d..h: pruned row transform -> lag patches */
    SFFT_ST_FILL,              /* FillLS_* + Remove_LSFStripes */
    SFFT_ST_SOLVE,             /* i: dense solve + Extend_Solution */
    SFFT_ST_PRELIM_APPLY,      /* b+c on the full pair */
    SFFT_ST_CONSTRUCT,         /* j+k: kernel transfer function + Construct_FDIFF */
    SFFT_ST_INVERSE,           /* k: inverse DFT + DIFF epilogue */
    SFFT_ST_GREEK_G1B,         /* e..h: Theta and Gamma passes (kernel greek_g1, w lags) */
    SFFT_ST_FWD_ROWS,          /* inside b+c of the solve: the row pass of the forward DFTs (kernel rows_r2c*) */
    SFFT_ST_FWD_COLS,          /* inside b+c of the solve: the column pass of the forward DFTs (kernel cols_c2c*) */
    SFFT_ST_COUNT
};

/* SingleSFFTConfigure.SSC (sfft/sfftcore/SFFTConfigure.py:1369-1397): validates arguments, derives the
 * parameter dictionary (:34-75), builds twiddle / index / polynomial tables and all workspaces on `device`.
 * Replaces the per-call nvcc / numba JIT of the reference with a cacheable handle. */
int sfft_plan_create(sfft_plan** plan, int N0, int N1, int KerHW, int KerPolyOrder, int BGPolyOrder,
                     int ConstPhotRatio, int device);

/* The same "compile" step for general separable spatial bases -- the B-spline form of SFFT
 * (sfft/BSplineSFFT.py: SingleSFFTConfigure.SSC :2536-2607, bases tabulated on the host like :2624-2645).
 * Kernel term ij is  kbx[ker_pairs[2 ij]][row] * kby[ker_pairs[2 ij + 1]][col]; background term pq is
 * tbx[bkg_pairs[2 pq]][row] * tby[bkg_pairs[2 pq + 1]][col].  Tables are HOST pointers, row-major
 * [n factors][axis length]; they are copied.  scaling_mode: 0 = scaling entangled with the kernel (no constraint),
 * 1 = unknowns ij00[1:] removed (polynomial ConstPhotRatio, Remove_LSFStripes), 2 = unknowns ij00 tied to one
 * value (B-spline constant scaling: rows/columns summed, BSplineSFFT.py:2201-2272).
 * Limits: Fij <= 64, Fpq <= 64, <= 16 factors per axis. */
int sfft_plan_create_basis(sfft_plan** plan, int N0, int N1, int KerHW,
                           int nkx, int nky, const double* kbx, const double* kby, int Fij, const int* ker_pairs,
                           int nbx, int nby, const double* tbx, const double* tby, int Fpq, const int* bkg_pairs,
                           int scaling_mode, int device);

/* The B-spline form with SEPARATELY VARYING flux scaling (sfft/BSplineSFFT.py SCALING_MODE 'SEPARATE-VARYING':
 * parameters :173-201, scaling planes :334-397, linear system :1348-2005, TweakLS :2293-2342, Construct_FDIFF :2429-2527).
 * As sfft_plan_create_basis, plus the spatial basis of the scaling: term s (s < ScaFij <= Fij) is
 * sbx[sca_pairs[2 s]][row] * sby[sca_pairs[2 s + 1]][col].  The unknown (ij, ab = kernel centre) is the coefficient of
 * scaling term ij for ij < ScaFij; for ij >= ScaFij it leaves the system and is returned as 0 (the reference's zero
 * place-holder terms, ScaREF_ij == (-1, -1)). */
int sfft_plan_create_varscale(sfft_plan** plan, int N0, int N1, int KerHW,
                              int nkx, int nky, const double* kbx, const double* kby, int Fij, const int* ker_pairs,
                              int nsx, int nsy, const double* sbx, const double* sby, int ScaFij, const int* sca_pairs,
                              int nbx, int nby, const double* tbx, const double* tby, int Fpq, const int* bkg_pairs,
                              int device);

/* Kernel regularisation (sfft/BSplineSFFT.py REGULARIZE_KERNEL: matrices :3570-3686, fill_regmat :2090-2166, update :3700):
 * every later solve on this plan adds  lambda * SCALE^2 * S[k][k8] * ireg[c][c8]  to LHMAT[(k, c), (k8, c8)] before the
 * scaling constraint is applied.  ireg [Fab][Fab] = the reference's iREGMAT (integer valued), sst [Fij][Fij] = SSTMAT;
 * csst / dsst [Fij][Fij] = CSSTMAT / DSSTMAT, required only by plans from sfft_plan_create_varscale (S = csst when exactly
 * one of c, c8 is the kernel centre -- indexed [kernel term][scaling term] -- and dsst when both are).  HOST pointers,
 * copied.  lambda == 0 or ireg == NULL switches regularisation off. */
int sfft_plan_set_regularization(sfft_plan* plan, double lambda, const double* ireg, const double* sst,
                                 const double* csst, const double* dsst);

int sfft_plan_destroy(sfft_plan* plan);

/* read one SFFT_Q_* field */
int sfft_plan_query(const sfft_plan* plan, int field, long long* value);

/* ElementalSFFTSubtract.ESS(PixA_I, PixA_J, SFFTConfig, SFFTSolution=None, Subtract=False)[0]
 * (sfft/sfftcore/SFFTSubtract.py:823-837; body :8-412 Cupy, :477-755 Numpy):
 * establish the normal equations from the (masked) pair and solve them.
 * d_solution: [NEQ] float64, ordered [a_ijab (ij-major, ab row-major), b_pq]. Synchronises `stream`. */
int sfft_solve(sfft_plan* plan, const double* d_I, const double* d_J, double* d_solution, void* stream);

/* ElementalSFFTSubtract.ESS(PixA_I, PixA_J, SFFTConfig, SFFTSolution=sol, Subtract=True)[1]
 * (sfft/sfftcore/SFFTSubtract.py:428-461 Cupy, :773-807 Numpy): DIFF = J - I (*) K - B for a given solution.
 * d_diff: [N0][N1] float64. Stream-ordered, does not synchronise. */
int sfft_apply(sfft_plan* plan, const double* d_I, const double* d_J, const double* d_solution,
               double* d_diff, void* stream);

/* GeneralSFFTSubtract.GSS / GeneralSFFTSubtract_PureCupy.GSS with ContamMask_I=None
 * (sfft/sfftcore/SFFTSubtract.py:839-904, 1371-1430): solve on (mI, mJ), apply to (I, J). Synchronises. */
int sfft_subtract(sfft_plan* plan, const double* d_I, const double* d_J, const double* d_mI, const double* d_mJ,
                  double* d_solution, double* d_diff, void* stream);

/* Parity aid: the linear system of the most recent sfft_solve()/sfft_subtract() on this plan, before
 * stripe removal, exactly as ESS holds it (LHMAT[NEQ][NEQ], RHb[NEQ]; SFFTSubtract.py:616-617).
 * Either pointer may be NULL. Synchronises. */
int sfft_get_system(sfft_plan* plan, double* d_LHMAT, double* d_RHb, void* stream);

/* Parity aid: the system the factorisation actually receives for the most recent solve -- after Remove_LSFStripes
 * (SFFTConfigure.py:693-711) / TweakLS (BSplineSFFT.py:2170-2338), with the regularisation term added -- written by the same
 * kernel launch (`fill_system`) that fills the solver's workspace.  n = SFFT_Q_SOLVER_N unknowns.
 *   d_bordered [n+1][n+1] float64: rows / columns 0 .. n-1 the matrix, row n (and column n) the right-hand side, [n][n] = 0;
 *   d_index    [n] int32, may be NULL: position of reduced unknown k in the full Solution vector (identity when nothing was removed).
 * Synchronises. */
int sfft_get_solver_system(sfft_plan* plan, double* d_bordered, int* d_index, void* stream);

/* Parity aid: the dense solver alone on a caller's system of the plan's size.  d_bordered [n+1][n+1] float64 in the layout
 * sfft_get_solver_system writes (rows / columns 0 .. n-1 the matrix, column n AND row n the right-hand side), n = SFFT_Q_SOLVER_N;
 * use_lu = 1: LU with partial pivoting, the reference's solver (np.linalg.solve / cupy.linalg.solve, SFFTSubtract.py:15-23), for
 * any nonsingular matrix; use_lu = 0: the Cholesky path (symmetric positive definite input, lower triangle + border row read).
 * d_x [n] float64 receives the solution in the solver's own ordering (no Extend_Solution).  Returns SFFT_ERR_SINGULAR like
 * sfft_solve.  Synchronises. */
int sfft_dbg_solve_dense(sfft_plan* plan, const double* d_bordered, int use_lu, double* d_x, void* stream);

/* Parity aid: SCALE * DFT2(I * kbx[i][row] * kby[j][col]) (= I * cx^i * cy^j for polynomial plans with i, j <= DK)
 * in the plan's half-spectrum layout, d_spec: [N0][N1/2+1] complex128 (interleaved re,im) -- items 3+4 of SURVEY.md 8(a). */
int sfft_dbg_forward_spectrum(sfft_plan* plan, const double* d_I, int i, int j, double* d_spec, void* stream);

/* ---- FFT utilities behind the post-subtraction helpers: sfft/utils/PureCupyFFTKits.py
 * (KERNEL_CSZ, FFT_CONVOLVE :37-105) and sfft/utils/PureCupyDeCorrelationCalculator.py (PCDC :46-126),
 * CPU twin sfft/utils/DeCorrelationCalculator.py (DCC :11-104).  cupy.fft.fft2 / ifft2 of REAL images become: ---- */

/* plan that only serves the FFT entry points below (any supported shape) */
int sfft_fft_plan_create(sfft_plan** plan, int N0, int N1, int device);

/* d_spec [N0][N1/2+1] complex128 (dense, interleaved) = scale * DFT2(d_real [N0][N1]); numpy.fft.rfft2 layout */
int sfft_fft2_r2c(sfft_plan* plan, const double* d_real, double* d_spec, double scale, void* stream);

/* d_real [N0][N1] = scale * sum_k d_spec[k] exp(+2 pi i k.x / N), d_spec the half spectrum of a real image
 * (scale = 1/(N0*N1) reproduces numpy.fft.irfft2) */
int sfft_ifft2_c2r(sfft_plan* plan, const double* d_spec, double* d_real, double scale, void* stream);

/* d_acc[i] += coeff * |d_a[i]|^2 * |d_b[i]|^2   (d_b may be NULL); a, b complex128, acc float64 */
int sfft_spec_abs2_accumulate(const double* d_a, const double* d_b, double coeff, double* d_acc, long long n, void* stream);

/* d_out[i] = 1 / sqrt(d_acc[i]) */
int sfft_real_rsqrt(const double* d_acc, double* d_out, long long n, void* stream);

/* d_out[i] = d_a[i] * d_b[i]; a, out complex128; b complex128 (b_is_real = 0) or float64 (b_is_real = 1) */
int sfft_spec_multiply(const double* d_a, const double* d_b, int b_is_real, double* d_out, long long n, void* stream);

/* d_full [N0][N1] float64 of a real conjugate-symmetric spectrum quantity from its half d_half [N0][N1/2+1] */
int sfft_half_to_full_real(const double* d_half, double* d_full, int N0, int N1, void* stream);

/* BSpline_GridConvolve.GSVC_GPU (sfft/BSplineSFFT.py:4951-5006): grid-wise space-varying convolution.
 * d_labels [N0][N1] int32 assigns every pixel to one of Nseg box segments (the reference's AllocatedL), d_kerstack
 * [Nseg][L0][L1] float64 holds one kernel per segment (normalise on the host if wanted).
 *   d_out[x][y] = sum_ab kerstack[label][a][b] * d_in[x + (L0-1)/2 - a][y + (L1-1)/2 - b],  zeros beyond the image,
 * i.e. convolve2d(mode='same', boundary='fill', fillvalue=0) of each segment with its own kernel.  Stream-ordered. */
int sfft_grid_convolve(const double* d_in, const int* d_labels, const double* d_kerstack, int N0, int N1, int Nseg,
                       int L0, int L1, double* d_out, int device, void* stream);

/* enable (1) / disable (0) hipEvent timing of the stages of subsequent calls */
int sfft_set_timing(sfft_plan* plan, int enable);

/* milliseconds spent in stage `stage` (SFFT_ST_*) during the most recent timed call; synchronises */
int sfft_stage_ms(sfft_plan* plan, int stage, float* ms);

/* Names of the HIP kernels stage `stage` has launched since timing was last enabled (sfft_set_timing(plan, 1) clears the lists), as
 * written at their launch sites, each once, ';'-separated, NUL-terminated, truncated to `cap` bytes -- so that a benchmark line can name the
 * kernels its stage times belong to (the reference prints stage letters only, sfft/sfftcore/SFFTSubtract.py:594-597). */
int sfft_stage_kernels(sfft_plan* plan, int stage, char* buf, int cap);

/* force the LU fallback for every solve (1) or let the plan choose (0, default: Cholesky first) */
int sfft_set_force_lu(sfft_plan* plan, int enable);

const char* sfft_last_error(void);

/* library version string, e.g. "sfft_amd 0.1 (gfx950)" */
const char* sfft_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SFFT_AMD_H */
