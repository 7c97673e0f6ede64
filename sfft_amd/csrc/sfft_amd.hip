// sfft_amd.hip -- MI355X (gfx950) native SFFT subtraction core: kernels + C ABI (include/sfft_amd.h).
//
// Path covered (reference = thomasvrussell/sfft v1.7.3):
//   SpatialCoor/SpatialPoly   sfft/sfftcore/SFFTConfigure.py:84-145     fused into rows_r2c (no coordinate planes)
//   preliminary DFTs          sfft/sfftcore/SFFTSubtract.py:146-168     rows_r2c + cols_c2c (half spectrum, fp64)
//   HadProd_* + Greek DFTs    SFFTConfigure.py:150-662, SFFTSubtract.py:226-383   greek_g1 + greek_g2 (pruned to the lags FillLS reads)
//   FillLS_* + stripes        SFFTConfigure.py:198-711                  fill_system
//   LSSolver                  SFFTSubtract.py:15-23, 398-403            blocked Cholesky (LU with partial pivoting as fallback)
//   Extend_Solution           SFFTConfigure.py:716-732                  scatter_solution / lu_backsolve
//   twiddles + Construct_FDIFF SFFTSubtract.py:433-447, SFFTConfigure.py:737-809   kernel_rtab + construct_fd
//   inverse DFT               SFFTSubtract.py:460-461                   cols_c2c(inverse) + rows_c2r_diff
//
// Layout in HBM: images [N0][N1] f64 row-major; spectra [plane][N0][Nhp] complex128 with Nh = N1/2+1
// columns kept (the inputs are real, so the other half is the conjugate mirror) and Nhp >= Nh the padded
// row stride.  All arithmetic is IEEE fp64.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "../../include/sfft_amd.h"

typedef double2 cplx;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
static int set_err(int code, const std::string& msg) { g_last_error = msg; return code; }
#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            char _b[512];                                                                         \
            snprintf(_b, sizeof(_b), "%s failed at %s:%d: %s", #call, __FILE__, __LINE__,        \
                     hipGetErrorString(_e));                                                      \
            return set_err(SFFT_ERR_HIP, _b);                                                     \
        }                                                                                         \
    } while (0)

// Every entry point runs on the plan's device and leaves the calling thread's current device as it found it (a host thread that
// drives several GPUs, or torch code that reads the current device afterwards, must not see it change behind its back).
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == dev) prev = -1;
        else ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define ON_DEVICE(dev)                                                                            \
    DeviceGuard device_guard_(dev);                                                               \
    if (!device_guard_.ok) return set_err(SFFT_ERR_HIP, "hipSetDevice failed")
// the stateless helpers take no plan: they run on the device their stream belongs to (the current device for the null stream)
static int stream_device(hipStream_t s)
{
    int dev = 0;
    if (s) { hipDevice_t d; if (hipStreamGetDevice(s, &d) == hipSuccess) return (int)d; }
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return dev;
}

extern "C" const char* sfft_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* sfft_version(void) { return "sfft_amd 0.1 (gfx950)"; }

// ---- which kernels a timed stage launched (sfft_stage_kernels) -----------------------------------------------------------------
// Every launch of this file goes through SFFT_LAUNCH.  While a StageTimer of a plan with timing enabled is alive, the name of each
// launched kernel (the macro argument as written, template arguments included) is appended once to that stage's list; the
// factorisation chain, which replays as a hipGraph after its first call, keeps the list of the capture and notes it on every replay.
static thread_local std::string* tl_klog = nullptr;
static void note_kernel(const char* name)
{
    std::string n;
    for (const char* c = name; *c; ++c) if (*c != ' ' && !(c == name && *c == '(')) n.push_back(*c);
    if (!n.empty() && name[0] == '(' && n.back() == ')') n.pop_back();
    std::string& L = *tl_klog;
    size_t pos = 0;
    while (pos <= L.size()) {           // entries are separated by ';'
        const size_t e = L.find(';', pos);
        const size_t len = (e == std::string::npos ? L.size() : e) - pos;
        if (len == n.size() && L.compare(pos, len, n) == 0) return;
        if (e == std::string::npos) break;
        pos = e + 1;
    }
    if (!L.empty()) L.push_back(';');
    L += n;
}
static void note_kernels(const std::string& list)
{
    size_t pos = 0;
    while (pos < list.size()) {
        const size_t e = list.find(';', pos);
        const std::string n = list.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
        if (!n.empty()) note_kernel(n.c_str());
        if (e == std::string::npos) break;
        pos = e + 1;
    }
}
struct KLogScope {      // route the names of the launches in this scope to `log` (nullptr: leave as is)
    std::string* prev; bool on;
    explicit KLogScope(std::string* log) : prev(tl_klog), on(log != nullptr) { if (on) tl_klog = log; }
    ~KLogScope() { if (on) tl_klog = prev; }
};
#define SFFT_LAUNCH(kernel, ...) do { if (tl_klog) note_kernel(#kernel); hipLaunchKernelGGL(kernel, __VA_ARGS__); } while (0)

#include "device_common.hpp"
#include "fft_generic.hpp"
#include "fft_fourstep.hpp"
#include "fft_r16_4096.hpp"
#include "fft_r24.hpp"
#include "greek.hpp"
#include "fill.hpp"
#include "solver.hpp"
#include "lu_api.hpp"
#include "construct.hpp"

// ================================================================================================
// host side
// ================================================================================================
struct AxisHost {
    int r16 = 0;                                    // power-of-two M >= 16: radix-16 stages (M + M/16 LDS elements per sequence)
    int N = 0, M = 0, logM = 0, blue = 0, n3 = 0;   // n3 > 0: M = N = 2^logM * 3^n3 (mixed-radix on-chip transform)
    cplx *tw = nullptr, *chirp = nullptr, *bf = nullptr, *root = nullptr;
    bool root_is_tw = false;
    int rader = 0;                                  // N = 577 as a four-step sub-transform: Rader's algorithm on M = 576 points (lds_rader577)
    int *rin = nullptr, *rout = nullptr;            // [N] each: LDS slot of input / output element n in Rader order
    // four-step decomposition for lengths that do not fit one on-chip transform: N = A * B
    bool big = false;
    int A = 0, B = 0;
    AxisHost* subA = nullptr;
    AxisHost* subB = nullptr;
    // ... and for lengths with no on-chip factorisation either (a prime factor > 4608): Bluestein through a four-step transform of
    // BM = 2^k >= 2 N - 1 points (subA = the BM-point axis; chirp [N], bf [BM]); big is set too, big_axis_transform dispatches
    bool bigblue = false;
    int BM = 0;
};

struct sfft_plan {
    int dev = 0;
    int N0 = 0, N1 = 0, w = 0, DK = 0, DB = 0, cpr = 0;
    int L = 0, Fab = 0, Fij = 0, Fpq = 0, Fijab = 0, NEQ = 0, NEQfs = 0;
    int Nh = 0, Nhp = 0;
    double scale = 0.0;
    AxisHost ax0, ax1;
    SpecLayout lay;                     // layout of every half-spectrum plane of this plan (row-major, or panels on the 4096^2 fast path)
    int TC = 1, MS = 0;                 // column pass tiling
    int nt_rows = 64, nt_cols = 64;
    size_t lds_rows = 0, lds_cols = 0;
    // separable spatial bases (polynomial powers or B-spline basis functions), tabulated per axis
    int nkx = 0, nky = 0, nbx = 0, nby = 0;
    std::vector<int> kpair, bpair;      // [Fij][2] / [Fpq][2]: (x-factor, y-factor) of kernel term ij / background term pq
    double *d_kbx = nullptr, *d_kby = nullptr, *d_tbx = nullptr, *d_tby = nullptr;   // [nkx][N0], [nky][N1], [nbx][N0], [nby][N1]
    int mode = 0;                       // 0: free scaling, 1: unknowns ij00[1:] removed, 2: unknowns ij00 tied together,
                                        // 3: centre unknowns carry their own (scaling) spatial basis, ij00[nsca:] removed
    // scaling basis of mode 3 (BSplineSFFT.py 'SEPARATE-VARYING'): scaling plane s = I * sbx[spair[2s]] (x) sby[spair[2s+1]]
    int nsx = 0, nsy = 0, nsca = 0;
    std::vector<int> spair;
    double *d_sbx = nullptr, *d_sby = nullptr;
    ScaArgs sa;
    bool has_idx = false;               // a row/column selection (d_idx) precedes the solve
    // kernel regularisation tables (device copies; see FillArgs)
    double *d_ireg = nullptr, *d_sst = nullptr, *d_csst = nullptr, *d_dsst = nullptr;
    BkgArgs bk;
    // device tables
    int* d_idx = nullptr;               // [NEQfs] (only when mode != 0)
    double* d_phi = nullptr;            // [Fpq*Fpq]
    cplx* d_Xp = nullptr;               // [nbx][N0]   DFT of the background x-factors
    cplx* d_Yq = nullptr;               // [nby][Nhp]  DFT of the background y-factors (half spectrum)
    cplx* d_w0tab = nullptr; int hm = 1;   // [N0][hm] twiddle rows of the pruned column transform
    G1Pass* d_passes = nullptr;
    PatchJob* d_jobs = nullptr;
    std::vector<G1Pass> passes;         // order: Omega (i'j' <= ij), Theta (i'j'), Gamma dense (i'j', p >= 1), Gamma p = 0
    std::vector<PatchJob> jobs;         // order: Omega, Gamma (i'j', pq), Theta  (= patch layout read by fill_system)
    int n_omg = 0, n_gam = 0, n_the = 0, n_gamp = 0, n_gam0 = 0;
    int n_omg_rec = 0;                  // Omega pass RECORDS (launched + dual partners; products done by omega_sparse have none): the short passes follow them
    int n_omg_off = 0, n_omg_diag = 0;  // Omega products that are transformed (off-diagonal / diagonal)
    int n_omg_launch = 0;               // Omega pass records that are launched (the rest are partners of dual diagonal passes)
    // polynomial plans: the Gamma block straight from row moments of I (gamma_patches) instead of column-factor passes
    int gamma_analytic = 0; double* d_cyp = nullptr; double* d_rowmomI = nullptr; double* d_gamR = nullptr; GammaArgs ga;
    int gam_tab = 0, gam_nmu = 0, gam_db = 0;   // tabulated kernel column factors; moments per row; background degree
    int n_dense_w = 0, n_row0 = 0;      // passes of half width w through greek_g1 / through greek_g1_row0

    int S = 1, rows_per_chunk = 0;
    FillArgs fa;
    // workspaces
    cplx* d_spec = nullptr;             // [Fij+1][N0][Nhp]   (plane Fij: J in solve, FD in apply)
    cplx *d_big1 = nullptr, *d_big2 = nullptr, *d_colscr = nullptr;   // work arrays of the four-step path
    cplx *d_bb1 = nullptr, *d_bb2 = nullptr; int bb_lines = 0;        // work arrays [bb_lines][BM] of the Bluestein-through-four-step axes
    size_t bb_elems = 0;                // complex elements of each of them: min(BIGBLUE_WORK_ELEMS, the most lines a pass of this plan transforms x BM)
    double* d_ones = nullptr;           // [max(N0, N1)] of 1.0: the weight table of unweighted planes on the fast row pass
    double *d_zero = nullptr, *d_zsol = nullptr;   // zero image / zero solution for the stand-alone inverse FFT (lazy)
    // mixed-domain apply (polynomial kernels on the staged fast path, KerHW <= 8): no column transforms in the apply pass
    int vt_na = 0; int* d_ibase = nullptr;     // vconv_tensor: at most vt_na consecutive row factors are nonzero on a row, the first listed per row (0: all factors)
    int vtensor = 0, vncf = 0;          // vtensor = n: n x n tensor basis through vconv_tensor; vncf = stage planes (column factors) of the mixed-domain apply
    int use_vconv = 0, vw = 8;          // vw = compile-time half width the tables are padded to (4, 8 or 12)
    bool staged_solve = false;          // the solve pass leaves the stage planes of I first in d_stage (4096^2 fast path)
    cplx* d_stage_a = nullptr;          // [DK+1] stage planes of the full image (apply pass)
    cplx* d_ctabm = nullptr;            // [Fij][2 vw + 1][Nhp]
    cplx* d_stage = nullptr;            // fast path: row-pass output, one plane per distinct (image, column factor) (lazy)
    int n_stage_alloc = 0;
    cplx* d_spec2 = nullptr;            // [Fij][N0][Nhp] spectra of the full pair, filled on stream s2 during the solve (lazy)
    hipStream_t s2 = nullptr; hipEvent_t ev_in = nullptr, ev_pre = nullptr, ev_mom = nullptr, ev_gam = nullptr; 
    const double* overlap_I = nullptr;  // set by sfft_subtract for the duration of its sfft_solve call
    cplx* d_gp = nullptr;
    double* d_patches = nullptr; size_t n_patches = 0;
    double* d_A = nullptr; int ld = 0;
    double* d_dbuf = nullptr;           // [2][CB][CB] diagonal blocks handed from chol_update to chol_panel
    double* d_xv = nullptr;             // [NEQfs] solution in stripe-free ordering
    double* d_rd = nullptr;             // [NEQfs] reciprocal diagonal of the Cholesky factor
    double* d_partial = nullptr;        // [BACK_SLICES][CB] strip-product partials of the back substitution
    unsigned int* d_counter = nullptr;
    double* d_winv = nullptr;           // [nblk][CB][CB] inverses of the diagonal blocks of the Cholesky factor
    unsigned int* d_bflags = nullptr;   // [nblk] "x_b published" flags of the back substitution, stamped with the solve's epoch
    unsigned int* d_epoch = nullptr;    // solve counter behind the flag stamps (device resident: the launch chain has constant arguments)
    unsigned int* d_tflags = nullptr;   // [nbc][nbc + 1] "tile factored" flags of chol_dataflow, [nbc * (nbc + 1)] = its task counter
    double* d_w16 = nullptr;            // [nbc][4][16][16] inverses of the 16 x 16 diagonal sub-blocks of the factor (chol_dataflow's MFMA triangular solves)
    unsigned long long* d_trace = nullptr;   // env SFFT_DF_TRACE=1: [nbc][16] wall-clock stamps of the critical path of chol_dataflow (development aid)
    int dataflow = 1;                   // 1: n < chol_outer_min factors in the single-launch dataflow kernel (fixed; the launch-per-step chain remains for the outer-blocked tail)
    int df_groups = 64;                 // persistent workgroups of chol_dataflow (64: swept 24 .. 96 in rounds 4 and 5)
    hipGraphExec_t lu_exec = nullptr;   // the pivoted-LU chain (lu.hpp), captured the same way on its first use
    std::string lu_graph_kernels;
    LuPerm* d_luperm = nullptr;         // [panels] row permutation lists of the LU panels
    double* d_luxchg = nullptr;         // hand-off slots of the multi-workgroup LU panel (systems taller than 2048 rows)
    hipGraphExec_t chol_exec = nullptr; // the factorisation + back substitution chain, captured once (env SFFT_NO_GRAPH=1: plain launches)
    int use_graph = 1;
    int* h_status = nullptr;            // pinned: status word of the most recent attempt
    int test_fail_chol = 0;             // env SFFT_TEST_FAIL_CHOL=1 (tests): report the Cholesky attempt as failed, to exercise the LU redo path
    bool attempt_lu = false;            // the most recent attempt used LU
    int fused_step = 1;                 // fused update + panel launches (fixed)
    int n_bflags = 0;
    int back_variant = 1;               // back substitution as one launch (fixed)
    double* d_sol = nullptr;            // [NEQ] internal solution copy
    cplx* d_rtab = nullptr; int wpad = 4;   // [Fij][N0][1 + 2 wpad] per-row kernel transfer table of the apply pass
    double* d_rowmom = nullptr; double* d_delta = nullptr;
    int* d_status = nullptr;
    size_t ws_bytes = 0;
    int last_solver = 0, force_lu = 0, chol_status = 0;
    int vconv_rp = 2;                   // mixed-domain apply: 2 = two source rows per LDS table read (vconv_mixed2); env SFFT_VCONV_RP=1: one row (vconv_mixed)
    int num_cu = 256;
    int vconv_direct_launch = 0;        // env SFFT_VCONV_DIRECT=1: the leftover columns of the mixed-domain apply in a launch of their own (vconv_direct)
    int g1_dit = 1;                     // grouped Omega launch with the radix-2 decimation step along the rows wherever the shape allows (fixed)
    int vconv_r = -1;                   // env SFFT_VCONV_R: output rows per stream of vconv_mixed2 (-1: balanced against the CU count, 0: KS * L - 2 W as before)
    long long n_solves = 0, n_lu_fallback = 0, n_chol_stall = 0;     // SFFT_Q_SOLVES / _LU_FALLBACKS / _CHOL_STALLS
    int colq = 1;                       // four columns per workgroup on 4-column panels (cols_fwd_weighted_4096_q); other panel widths take the two-column kernel
    int colz = 1;                       // solve pass of the 4096^2 path: cols_fwd_weighted_4096_z (two workgroups per CU; pair-major stage lines in, 2-column
                                        // panels out).  env SFFT_COLZ=0: cols_fwd_weighted_4096_q.  Set to 0 by the plan when its conditions do not hold.
    SpecLayout lay_spec;                // layout of the solve pass's spectra in d_spec (= lay unless colz / spec2)
    int spec2 = 1;                      // 6144-point column axis on the paired kernel (cols_fwd_weighted_r24<16, 2>): spectra in 2-column panels.  env SFFT_SPEC2=0: the plan's layout
    int chol_outer_min = 3000;          // env SFFT_CHOL_OUTER_MIN: systems at least this large factor in 256-column outer blocks
    int g1_mfma = 3;                    // Omega passes on the matrix cores: 3 = greek_g1_mfma4g (pass groups that share plane loads, v_mfma_f64_4x4x4_4b_f64),
                                        // (fixed 3: the grouped launch; 2 = greek_g1_mfma4, one pass per wave, where no groups are built)
    unsigned long long* d_g1trace = nullptr;   // env SFFT_G1_TRACE=file: per-wave start / end stamps of the grouped Omega launch (development aid)
    G1Group* d_groups = nullptr;        // pass groups of the Omega launch
    int n_groups = 0;
    int panel4 = 1, ncu = 0;            // the four panel steps of an outer block as one launch (chol_panel4; fixed)
    unsigned int* d_pq = nullptr;       // [PANEL4_MAX_OUTER] role counters of chol_panel4 + [16] its hand-off flags
    int rowmom_fused = 0;               // 1: the row moments of the masked pair come out of rows_r2c_4096 (else separate row_moments launches)
    int n_the_fused = 0;                // leading Theta passes that ride in the groups (all Fij of them when Fij is even)
    int theta_in_groups = 0;            // 1: the Fij Theta passes ride in the edge groups of the Omega launch (else a vector launch of their own)
    int theta_slots = 0;                // 1: half width 9 .. 16 (KerHW 9 .. 16): the Theta passes as ordinary slots in groups of their own behind the Omega groups
    int n_groups_omg = 0;               //    (first launch only: n_groups counts them, n_groups_omg does not; env SFFT_THETA_SLOTS=0: a launch of their own)
    int rows_r24 = 0;                   // 16 / 24: 6144- / 9216-point row axis on the register-resident kernels of fft_r24.hpp (env SFFT_NO_ROWS_R24=1: the generic pass, A/B)
    bool cols_r24_pair = true;          // 6144-point columns: two panel neighbours per workgroup in neighbouring lanes (env SFFT_COLS_R24_PAIR=0: one column)
    int cols_r24 = 0;                   // the same for the column axis (env SFFT_NO_COLS_R24=1)
    std::vector<int> kbx_lo, kbx_hi;    // [nkx] first row / one past the last row where the kernel row factor is nonzero
    int no_staged = 0;                  // env SFFT_NO_STAGED=1: one row transform per plane instead of one per column factor (A/B testing)
    int no_fast_fft = 0;                // (fixed 0; SFFT_NO_R16 / SFFT_NO_MIXED_RADIX select the generic transforms for tests)
    // A/B switches of the launch paths, read ONCE at plan creation (never getenv on a hot path)
    bool want_mom_event = false, mom_event_recorded = false;   // solve pass: ev_mom right behind the row pass (the Gamma block then runs beside the COLUMN pass)
    int colscr_planes = 1;              // scratch planes of the four-step column path
    int no_dft16_regs = 0, no_rader_r24 = 0, vconv2_w12 = 1, inv_r24 = -1;
    // Omega products of basis terms with (nearly) disjoint supports: computed in real space by omega_sparse, no transform pass
    std::vector<SparseProd> sprods; SparseProd* d_sprods = nullptr; SparseLine* d_slines = nullptr; int* d_scols = nullptr;
    double* d_strip = nullptr; int n_scols = 0; int2* d_sitems = nullptr; int n_sitems = 0;      // (items: eight runs of n_sitems / 8, one per XCD)
    int launch_error = 0;               // a launch path met a case it does not serve (set by launch_pass; reported by the entry points)
    int timing = 0;
    std::string stage_kernels[SFFT_ST_COUNT];      // kernels each stage launched in the most recent timed call (sfft_stage_kernels)
    std::string graph_kernels;                     // kernels captured into the solver graph
    hipEvent_t ev[SFFT_ST_COUNT][2];
    bool ev_valid[SFFT_ST_COUNT];
    bool have_system = false;
};

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static SpecLayout rowmajor_layout(int ld) { SpecLayout L; L.shift = 31; L.mask = 0x7fffffff; L.rstride = ld; L.pstride = 0; return L; }
static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// Debug allocator (env SFFT_GUARD=1 | 2): every plan buffer is mapped on its own with unmapped address space on both sides
// (HIP virtual-memory API), its END (1) or START (2) flush with the mapping, so that a kernel reading or writing out of
// bounds faults instead of silently touching a neighbour.  Such buffers are never freed (debug runs only).
static int guard_mode() { static int v = [] { const char* e = getenv("SFFT_GUARD"); return e ? atoi(e) : 0; }(); return v; }
// switches that several places of the plan builder consult (read at every plan creation, never cached: tests flip them between plans)
static bool env_no_r16() { return getenv("SFFT_NO_R16") != nullptr; }
static bool env_no_mixed_radix() { return getenv("SFFT_NO_MIXED_RADIX") != nullptr; }
static bool env_no_vconv() { return getenv("SFFT_NO_VCONV") != nullptr; }
static int env_theta_slots() { const char* e = getenv("SFFT_THETA_SLOTS"); return e ? atoi(e) : 1; }
static hipError_t guarded_malloc(void** out, size_t bytes, int dev)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return e;
    const size_t mapped = (bytes + gran - 1) / gran * gran;
    void* va = nullptr;
    if ((e = hipMemAddressReserve(&va, mapped + 2 * gran, gran, nullptr, 0)) != hipSuccess) return e;
    hipMemGenericAllocationHandle_t h;
    if ((e = hipMemCreate(&h, mapped, &prop, 0)) != hipSuccess) return e;
    char* base = (char*)va + gran;
    if ((e = hipMemMap(base, mapped, 0, h, 0)) != hipSuccess) return e;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(base, mapped, &acc, 1)) != hipSuccess) return e;
    const size_t slack = (mapped - bytes) & ~(size_t)15;             // keep 16-byte alignment
    *out = guard_mode() == 1 ? base + slack : base;
    return hipSuccess;
}
static void dev_free(void* q) { if (q && !guard_mode()) hipFree(q); }

template <typename T>
static int dev_alloc(sfft_plan* p, T** ptr, size_t count)
{
    void* q = nullptr;
    hipError_t e = guard_mode() ? guarded_malloc(&q, count * sizeof(T) > 0 ? count * sizeof(T) : 16, p->dev)
                                : hipMalloc(&q, count * sizeof(T) > 0 ? count * sizeof(T) : 16);
    if (e != hipSuccess) return set_err(SFFT_ERR_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e));
    *ptr = reinterpret_cast<T*>(q);
    p->ws_bytes += count * sizeof(T);
    return SFFT_OK;
}

static void host_fft_pow2(std::vector<long double>& re, std::vector<long double>& im)
{
    const int n = (int)re.size();
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    const long double PI = acosl(-1.0L);
    for (int len = 2; len <= n; len <<= 1) {
        for (int i = 0; i < n; i += len) {
            for (int k = 0; k < len / 2; ++k) {
                const long double ang = -2.0L * PI * k / len;
                const long double wr = cosl(ang), wi = sinl(ang);
                const int a = i + k, b = i + k + len / 2;
                const long double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
        }
    }
}

static const size_t LDS_MAX_ELEMS = 8192;   // longest on-chip transform: 128 KiB of complex128 (160 KiB LDS per CU on gfx950)
static const size_t LDS_MIXED_ELEMS = 9216; // longest 2^a*3^b transform: 144 KiB (+16 B of column padding)
static size_t lds_col_elems() { return 9216; }
#define LDS_COL_ELEMS lds_col_elems()      // column tile budget (144 KiB): two padded 4096-point columns fit

// N = 2^a * 3^b with b >= 1: writes a, b
static bool is_2a3b(int N, int* a, int* b)
{
    int e2 = 0, e3 = 0;
    while (N % 2 == 0) { N /= 2; ++e2; }
    while (N % 3 == 0) { N /= 3; ++e3; }
    if (N != 1 || e3 == 0) return false;
    *a = e2; *b = e3;
    return true;
}

// forward DFT of a length 2^a 3^b sequence on the host (long double; plan creation only)
static void host_fft_23(std::vector<long double>& re, std::vector<long double>& im)
{
    const int n = (int)re.size();
    if (n == 1) return;
    const int f = (n % 2 == 0) ? 2 : 3, m = n / f;
    std::vector<long double> sr[3], si[3];
    for (int r = 0; r < f; ++r) {
        sr[r].resize(m); si[r].resize(m);
        for (int j = 0; j < m; ++j) { sr[r][j] = re[r + f * j]; si[r][j] = im[r + f * j]; }
        host_fft_23(sr[r], si[r]);
    }
    const long double PI = acosl(-1.0L);
    for (int k = 0; k < m; ++k)
        for (int q = 0; q < f; ++q) {
            const int o = k + m * q;
            long double xr = 0.0L, xi = 0.0L;
            for (int r = 0; r < f; ++r) {
                const long double ang = -2.0L * PI * (long double)((long long)r * o % n) / n;
                const long double wr = cosl(ang), wi = sinl(ang);
                xr += sr[r][k] * wr - si[r][k] * wi;
                xi += sr[r][k] * wi + si[r][k] * wr;
            }
            re[o] = xr; im[o] = xi;
        }
}

static bool is_2a3b(int N, int* a, int* b);
// Bluestein transform length for an N-point axis: the power of two >= 2 N - 1 when it fits on chip; beyond that (N in
// 4097 .. 4608) the 9216-point mixed-radix transform.  (Shorter 2^a 3^b lengths were measured and lose to the next power of
// two -- 577 points on 1296 instead of 2048: 19.8 -> 23.8 ms for config 5's columns -- the radix-3 stages and their general
// index arithmetic cost more than the shorter length saves.)  0 if nothing fits.
static int bluestein_len(int N)
{
    const int need = 2 * N - 1;
    int p2 = 1; while (p2 < need) p2 <<= 1;
    if ((size_t)p2 <= LDS_MAX_ELEMS) return p2;
    if (need <= 9216 && !env_no_mixed_radix() && !env_no_r16()) return 9216;
    return 0;
}

static bool fits_on_chip(int N)
{
    if (is_pow2(N)) return (size_t)N <= LDS_MAX_ELEMS;
    int a, b;
    if (is_2a3b(N, &a, &b) && (size_t)N <= LDS_MIXED_ELEMS && !env_no_mixed_radix()) return true;
    return bluestein_len(N) != 0;
}

// rader: this axis may use Rader sub-transforms (the COLUMN axis only: strided_rader577 is the lines-fastest pass of a column transform)
static int build_axis(sfft_plan* p, AxisHost& ax, int N, bool sub_axis = false, bool rader = false);
static bool rader_ok(int N) { return N == RADER_M + 1 && !getenv("SFFT_NO_RADER") && !env_no_r16(); }

#define BIGBLUE_MAX_N 16384
#define BIGBLUE_WORK_ELEMS ((size_t)1 << 24)         // complex elements per work array (256 MB): lines are taken in batches of this many / BM
static int build_big_axis(sfft_plan* p, AxisHost& ax, int N, bool rader);
// exp(-i pi k^2 / N) and the BM-point transform of its conjugate's wrap-around extension (the Bluestein filter), BM = 2^k >= 2 N - 1 > 8192
static int build_bigblue_axis(sfft_plan* p, AxisHost& ax, int N)
{
    const long double PI = acosl(-1.0L);
    if (N > BIGBLUE_MAX_N) return set_err(SFFT_ERR_UNSUPPORTED_SIZE, "image side not supported by this build: sides above 16384 must factor as A * B "
                                          "with both factors on chip (power of two <= 8192, 2^a 3^b <= 9216, any length <= 4608)");
    int M = 1; while (M < 2 * N - 1) M <<= 1;
    ax.N = N; ax.big = true; ax.bigblue = true; ax.BM = M; ax.A = 0; ax.B = 0; ax.M = 0; ax.logM = 0; ax.blue = 0;
    ax.subA = new AxisHost();
    int rc;
    if ((rc = build_big_axis(p, *ax.subA, M, false))) return rc;          // M = 16384 / 32768: 4096 x 4 / 4096 x 8
    std::vector<cplx> c(N);
    std::vector<long double> fr(M, 0.0L), fi(M, 0.0L);
    for (int k = 0; k < N; ++k) {
        const long long q = ((long long)k * k) % (2LL * N);
        const long double a2 = PI * q / N;
        c[k] = make_double2((double)cosl(a2), (double)-sinl(a2));
        fr[k] = cosl(a2); fi[k] = sinl(a2);
        if (k > 0) { fr[M - k] = fr[k]; fi[M - k] = fi[k]; }
    }
    host_fft_pow2(fr, fi);
    std::vector<cplx> bf(M);
    for (int k = 0; k < M; ++k) bf[k] = make_double2((double)(fr[k] / M), (double)(fi[k] / M));
    std::vector<cplx> r(N);                             // the N-th roots of unity every big axis carries (untangle_rows, the mixed-domain apply ...)
    for (int k = 0; k < N; ++k) {
        const long double ang = -2.0L * PI * k / N;
        r[k] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    if ((rc = dev_alloc(p, &ax.chirp, N)) || (rc = dev_alloc(p, &ax.bf, M)) || (rc = dev_alloc(p, &ax.root, N))) return rc;
    HIPCHK(hipMemcpy(ax.root, r.data(), N * sizeof(cplx), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ax.chirp, c.data(), N * sizeof(cplx), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ax.bf, bf.data(), M * sizeof(cplx), hipMemcpyHostToDevice));
    // work arrays: a batch of lines x M elements, never more than the plan's longest pass needs (an FFT-only plan of a small prime-sided image
    // used to hold 2 x 256 MB here); the two axes of a plan share them
    const size_t max_lines = (size_t)std::max(std::max(p->N0, p->N1), 1);
    const size_t need = std::min<size_t>(BIGBLUE_WORK_ELEMS, max_lines * (size_t)M);
    if (need > p->bb_elems) {
        if (p->d_bb1) { dev_free(p->d_bb1); p->d_bb1 = nullptr; }
        if (p->d_bb2) { dev_free(p->d_bb2); p->d_bb2 = nullptr; }
        if ((rc = dev_alloc(p, &p->d_bb1, need)) || (rc = dev_alloc(p, &p->d_bb2, need))) return rc;
        p->bb_elems = need;
    }
    return SFFT_OK;
}

// N = A * B with A the largest power-of-two factor (<= 4096) such that B fits on chip too
static int build_big_axis(sfft_plan* p, AxisHost& ax, int N, bool rader)
{
    const long double PI = acosl(-1.0L);
    // any factorisation N = A * B with both factors on chip; cost per element ~ passes over LDS of the two sub-transforms
    // (1 for a direct power-of-two / 2^a 3^b length, 4 M / len for Bluestein: two transforms of M >= 2 len - 1 plus products)
    auto cost = [](int len, bool may_rader) {      // (only factor B is ever built with Rader: subA carries the four-step twiddles)
        int e2, e3;
        if (is_pow2(len) || is_2a3b(len, &e2, &e3)) return 1.0;
        if (may_rader && rader_ok(len)) return 2.5;      // two transforms of len - 1 points
        return 4.0 * bluestein_len(len) / len;
    };
    int A = 0, B = 0;
    double best = 1e300;
    for (int a = 2; (long long)a * a <= N; ++a) {
        if (N % a) continue;
        const int b = N / a;
        if (!fits_on_chip(a) || !fits_on_chip(b)) continue;
        auto padded = [](int len) { int e2, e3; if (is_pow2(len) || is_2a3b(len, &e2, &e3)) return len; return bluestein_len(len); };
        const double c = cost(a, false) + cost(b, rader) + 1e-6 * (padded(a) + padded(b));      // (ties: the smaller on-chip transforms)
        if (c < best) { best = c; A = a; B = b; }
    }
    if (!A) return build_bigblue_axis(p, ax, N);        // no factorisation with both factors on chip: Bluestein through a four-step transform
    ax.N = N; ax.big = true; ax.A = A; ax.B = B; ax.M = 0; ax.logM = 0; ax.blue = 0;
    ax.subA = new AxisHost(); ax.subB = new AxisHost();
    int rc;
    if ((rc = build_axis(p, *ax.subA, A, true, false))) return rc;       // (the first pass carries the four-step twiddles: strided_dft)
    if ((rc = build_axis(p, *ax.subB, B, true, rader))) return rc;
    std::vector<cplx> r(N);
    for (int k = 0; k < N; ++k) {
        const long double ang = -2.0L * PI * k / N;
        r[k] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    if ((rc = dev_alloc(p, &ax.root, N))) return rc;
    HIPCHK(hipMemcpy(ax.root, r.data(), N * sizeof(cplx), hipMemcpyHostToDevice));
    return SFFT_OK;
}

// Rader tables of a prime N whose N - 1 = RADER_M (only strided_dft, the four-step kernel, sizes its tiles for M < N)
static int build_rader_axis(sfft_plan* p, AxisHost& ax, int N)
{
    const long double PI = acosl(-1.0L);
    const int M = N - 1;
    ax.N = N; ax.M = M; ax.blue = 0; ax.rader = 1; ax.n3 = 0; ax.logM = 0; ax.r16 = 1;       // (r16: padded LDS layout, M + M/16 elements)
    auto powmod = [&](long long b, long long e) { long long r = 1; b %= N; while (e) { if (e & 1) r = r * b % N; b = b * b % N; e >>= 1; } return r; };
    int g = 0;
    for (int c = 2; c < N && !g; ++c) {             // smallest primitive root: c^((N-1)/q) != 1 for every prime q | N - 1
        bool ok = true;
        int m = M;
        for (int q = 2; q <= m && ok; ++q)
            if (m % q == 0) { if (powmod(c, M / q) == 1) ok = false; while (m % q == 0) m /= q; }
        if (ok) g = c;
    }
    if (!g) return set_err(SFFT_ERR_INVALID_ARG, "Rader: length is not prime");
    const long long ginv = powmod(g, N - 2);
    std::vector<int> rin(2 * N, 0), rout(2 * N, 0); // [0, N): LDS slots of input / output element n (strided_rader577); [N, 2N - 1): the inverse
                                                    // maps, element of slot r: g^r, g^-q (strided_rader577_r24)
    std::vector<long double> br(M), bi(M);
    for (int r = 0; r < M; ++r) { rin[(int)powmod(g, r)] = r; rin[N + r] = (int)powmod(g, r); }
    rin[0] = RADER_XS; rout[0] = RADER_XS;
    for (int q = 0; q < M; ++q) {
        const int n = (int)powmod(ginv, q);
        rout[n] = q; rout[N + q] = n;
        const long double ang = -2.0L * PI * n / N;   // b[q] = W_N^(g^-q)
        br[q] = cosl(ang); bi[q] = sinl(ang);
    }
    host_fft_23(br, bi);
    std::vector<cplx> bf(M), tw(M), root(N);
    for (int k = 0; k < M; ++k) {
        bf[k] = make_double2((double)(br[k] / M), (double)(bi[k] / M));
        const long double ang = -2.0L * PI * k / M;
        tw[k] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    bf[0] = make_double2(-1.0 / M, 0.0);            // sum of the non-trivial N-th roots of unity: lds_rader577 reads A[0] back through it
    for (int k = 0; k < N; ++k) { const long double ang = -2.0L * PI * k / N; root[k] = make_double2((double)cosl(ang), (double)sinl(ang)); }
    int rc;
    if ((rc = dev_alloc(p, &ax.tw, M)) || (rc = dev_alloc(p, &ax.bf, M)) || (rc = dev_alloc(p, &ax.root, N)) ||
        (rc = dev_alloc(p, &ax.rin, 2 * N)) || (rc = dev_alloc(p, &ax.rout, 2 * N))) return rc;
    HIPCHK(hipMemcpy(ax.tw, tw.data(), M * sizeof(cplx), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ax.bf, bf.data(), M * sizeof(cplx), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ax.root, root.data(), N * sizeof(cplx), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ax.rin, rin.data(), 2 * N * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ax.rout, rout.data(), 2 * N * sizeof(int), hipMemcpyHostToDevice));
    return SFFT_OK;
}

static int build_axis(sfft_plan* p, AxisHost& ax, int N, bool sub_axis, bool rader)
{
    const long double PI = acosl(-1.0L);
    if (!fits_on_chip(N)) return build_big_axis(p, ax, N, rader);
    if (sub_axis && rader && rader_ok(N)) return build_rader_axis(p, ax, N);
    ax.N = N;
    int e2 = 0, e3 = 0;
    if (is_pow2(N)) { ax.M = N; ax.blue = 0; }
    else if (is_2a3b(N, &e2, &e3) && !env_no_mixed_radix()) { ax.M = N; ax.blue = 0; ax.n3 = e3; }
    else {
        ax.M = bluestein_len(N); ax.blue = 1;
        if (!is_pow2(ax.M)) { is_2a3b(ax.M, &e2, &e3); ax.n3 = e3; }
    }
    ax.logM = ax.n3 ? e2 : ilog2(ax.M);
    ax.r16 = ((ax.n3 || ax.M >= 16) && !env_no_r16()) ? 1 : 0;
    int rc;
    std::vector<cplx> h(ax.M);
    for (int k = 0; k < ax.M; ++k) {
        const long double ang = -2.0L * PI * k / ax.M;
        h[k] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    if ((rc = dev_alloc(p, &ax.tw, ax.M))) return rc;
    HIPCHK(hipMemcpy(ax.tw, h.data(), ax.M * sizeof(cplx), hipMemcpyHostToDevice));
    if (!ax.blue) { ax.root = ax.tw; ax.root_is_tw = true; return SFFT_OK; }
    std::vector<cplx> r(N), c(N);
    for (int k = 0; k < N; ++k) {
        const long double ang = -2.0L * PI * k / N;
        r[k] = make_double2((double)cosl(ang), (double)sinl(ang));
        const long long q = ((long long)k * k) % (2LL * N);
        const long double a2 = -PI * q / N;
        c[k] = make_double2((double)cosl(a2), (double)sinl(a2));
    }
    if ((rc = dev_alloc(p, &ax.root, N))) return rc;
    HIPCHK(hipMemcpy(ax.root, r.data(), N * sizeof(cplx), hipMemcpyHostToDevice));
    if ((rc = dev_alloc(p, &ax.chirp, N))) return rc;
    HIPCHK(hipMemcpy(ax.chirp, c.data(), N * sizeof(cplx), hipMemcpyHostToDevice));
    std::vector<long double> fr(ax.M, 0.0L), fi(ax.M, 0.0L);
    for (int k = 0; k < N; ++k) {
        const long long q = ((long long)k * k) % (2LL * N);
        const long double a2 = PI * q / N;      // conj(chirp)
        fr[k] = cosl(a2); fi[k] = sinl(a2);
        if (k > 0) { fr[ax.M - k] = fr[k]; fi[ax.M - k] = fi[k]; }
    }
    if (is_pow2(ax.M)) host_fft_pow2(fr, fi); else host_fft_23(fr, fi);
    std::vector<cplx> bf(ax.M);
    for (int k = 0; k < ax.M; ++k) bf[k] = make_double2((double)(fr[k] / ax.M), (double)(fi[k] / ax.M));
    if ((rc = dev_alloc(p, &ax.bf, ax.M))) return rc;
    HIPCHK(hipMemcpy(ax.bf, bf.data(), ax.M * sizeof(cplx), hipMemcpyHostToDevice));
    return SFFT_OK;
}

// LDS elements one sequence of this axis needs
static int axis_lds_len(const AxisHost& a) { return a.rader ? RADER_XS + 1 : a.r16 ? a.M + a.M / 16 : a.M; }

// threads of an on-chip FFT workgroup over `elems` LDS elements: one radix-16 butterfly each, whole waves
static int fft_threads(int elems) { return std::min(SFFT_FFT_MAX_THREADS, ((elems + 15) / 16 + 63) / 64 * 64); }

// Column tile: TC sequences side by side in LDS, sequence stride MS.  The load / store phases of the column kernels walk
// (element, sequence) with the sequence index fastest, so MS = 16 / TC (mod 16) elements puts the 16 lanes of a quarter wave
// on 16 different 16-byte bank groups.
static void pick_col_tile(const AxisHost& a, int* TC, int* MS, size_t budget = 0)
{
    if (!budget) budget = LDS_COL_ELEMS;
    const int len = (axis_lds_len(a) + 15) / 16 * 16;
    for (int tc = 16; tc >= 1; tc >>= 1) {
        const int ms = len + (tc == 1 ? 0 : 16 / tc);
        if (tc == 1 || ((size_t)tc * ms <= budget && (tc * a.M + 15) / 16 <= SFFT_FFT_MAX_THREADS)) { *TC = tc; *MS = ms; return; }
    }
}

static AxisDev axis_dev(const AxisHost& a)
{
    AxisDev d; d.N = a.N; d.M = a.M; d.logM = a.logM; d.blue = a.blue; d.n3 = a.n3; d.r16 = a.r16; d.tw = a.tw; d.chirp = a.chirp; d.bf = a.bf; d.root = a.root;
    d.rader = a.rader; d.rin = a.rin; d.rout = a.rout;
    return d;
}

static double ipow_host(double x, int e) { double r = 1.0; for (int t = 0; t < e; ++t) r *= x; return r; }

struct BasisSpec {
    int nkx = 0, nky = 0, nbx = 0, nby = 0, Fij = 0, Fpq = 0, mode = 0;
    std::vector<double> kbx, kby, tbx, tby;   // [nkx][N0], [nky][N1], [nbx][N0], [nby][N1]
    std::vector<int> kpair, bpair;            // [Fij][2], [Fpq][2]
    // mode 3 only: scaling factors and the (x-factor, y-factor) pair of each of the ScaFij <= Fij scaling terms
    int nsx = 0, nsy = 0, ScaFij = 0;
    std::vector<double> sbx, sby;             // [nsx][N0], [nsy][N1]
    std::vector<int> spair;                   // [ScaFij][2]
};

// DFT of a tabulated 1-D factor, direct O(N^2) in extended precision; an all-ones factor gives exactly N * delta
static void table_axis_dft(const double* v, int N, int nout, std::vector<cplx>& out, bool* is_const)
{
    const long double PI = acosl(-1.0L);
    bool ones = true;
    for (int x = 0; x < N; ++x) ones &= (v[x] == 1.0);
    *is_const = ones;
    out.assign(nout, make_double2(0.0, 0.0));
    if (ones) { out[0] = make_double2((double)N, 0.0); return; }
    std::vector<long double> cr(N), ci(N);
    for (int x = 0; x < N; ++x) { const long double ang = -2.0L * PI * x / N; cr[x] = cosl(ang); ci[x] = sinl(ang); }
    for (int k = 0; k < nout; ++k) {
        long double sr = 0.0L, si = 0.0L;
        long long q = 0;
        for (int x = 0; x < N; ++x) {
            sr += (long double)v[x] * cr[q]; si += (long double)v[x] * ci[q];
            q += k; if (q >= N) q -= N;
        }
        out[k] = make_double2((double)sr, (double)si);
    }
}

static int g1_padded(int h);

static int plan_create_impl(sfft_plan** out, int N0, int N1, int KerHW, const BasisSpec& BS, int DK, int DB, int device)
{
    if (!out) return set_err(SFFT_ERR_INVALID_ARG, "plan pointer is NULL");
    *out = nullptr;
    if (N0 < 8 || N1 < 8) return set_err(SFFT_ERR_INVALID_ARG, "Input Image has dramatically small size!");
    if (KerHW < 0 || KerHW > 32) return set_err(SFFT_ERR_INVALID_ARG, "KerHW must be in [0, 32]");
    if (BS.Fij < 1 || BS.Fij > 64 || BS.Fpq < 1 || BS.Fpq > SFFT_MAX_PQ || BS.nby > SFFT_MAX_BQ || BS.nbx > 16 || BS.nkx > 16 || BS.nky > 16)
        return set_err(SFFT_ERR_INVALID_ARG, "spatial basis too large: at most 64 kernel terms, 64 background terms, 16 factors per axis");
    ON_DEVICE(device);
    sfft_plan* p = new sfft_plan();
    p->dev = device;
    if (const char* ev = getenv("SFFT_NO_STAGED")) p->no_staged = atoi(ev);
    if (const char* ev = getenv("SFFT_CHOL_OUTER_MIN")) p->chol_outer_min = atoi(ev);
    if (const char* ev = getenv("SFFT_COLZ")) p->colz = atoi(ev);
    if (getenv("SFFT_NO_GRAPH")) p->use_graph = 0;
    if (getenv("SFFT_TEST_FAIL_CHOL")) p->test_fail_chol = 1;
    if (const char* ev = getenv("SFFT_VCONV_RP")) p->vconv_rp = atoi(ev);
    if (const char* ev = getenv("SFFT_VCONV_R")) p->vconv_r = atoi(ev);
    if (const char* ev = getenv("SFFT_VCONV_DIRECT")) p->vconv_direct_launch = atoi(ev);
    if (getenv("SFFT_NO_DFT16_REGS")) p->no_dft16_regs = 1;
    if (getenv("SFFT_NO_RADER_R24")) p->no_rader_r24 = 1;
    if (const char* ev = getenv("SFFT_VCONV2_W12")) p->vconv2_w12 = atoi(ev);
    if (const char* ev = getenv("SFFT_INV_R24")) p->inv_r24 = atoi(ev);
    p->N0 = N0; p->N1 = N1; p->w = KerHW; p->DK = DK; p->DB = DB; p->mode = BS.mode; p->cpr = (BS.mode == 1 || BS.mode == 2);
    if (BS.mode == 3) {
        if (BS.ScaFij < 1 || BS.ScaFij > BS.Fij || BS.nsx < 1 || BS.nsx > 16 || BS.nsy < 1 || BS.nsy > 16) {
            delete p;
            return set_err(SFFT_ERR_INVALID_ARG, "scaling basis: 1 <= ScaFij <= Fij and at most 16 factors per axis");
        }
        p->nsca = BS.ScaFij; p->nsx = BS.nsx; p->nsy = BS.nsy; p->spair = BS.spair;
    }
    p->has_idx = p->cpr || (BS.mode == 3 && p->nsca < BS.Fij);
    p->L = 2 * KerHW + 1; p->Fab = p->L * p->L;
    p->Fij = BS.Fij; p->Fpq = BS.Fpq;
    p->nkx = BS.nkx; p->nky = BS.nky; p->nbx = BS.nbx; p->nby = BS.nby;
    p->kpair = BS.kpair; p->bpair = BS.bpair;
    p->Fijab = p->Fij * p->Fab; p->NEQ = p->Fijab + p->Fpq;
    p->NEQfs = p->cpr ? p->NEQ - (p->Fij - 1) : (p->mode == 3 ? p->NEQ - (p->Fij - p->nsca) : p->NEQ);
    p->scale = 1.0 / ((double)N0 * (double)N1);
    p->Nh = N1 / 2 + 1;
    p->Nhp = (p->Nh + 3) & ~3;
    if (((p->Nhp / 4) & 1) == 0) p->Nhp += 4;     // row stride an odd multiple of 64 B: no power-of-two column stride
    for (int s = 0; s < SFFT_ST_COUNT; ++s) p->ev_valid[s] = false;
    int rc;
#define PLAN_TRY(x) do { rc = (x); if (rc) { sfft_plan_destroy(p); return rc; } } while (0)
#define PLAN_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { set_err(SFFT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); sfft_plan_destroy(p); return SFFT_ERR_HIP; } } while (0)
    for (int s = 0; s < SFFT_ST_COUNT; ++s) { PLAN_HIP(hipEventCreate(&p->ev[s][0])); PLAN_HIP(hipEventCreate(&p->ev[s][1])); }
    {
        int lo = 0, hi = 0;
        PLAN_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));     // lo = least urgent
        // the second stream runs the apply pass's forward transforms beside the dense solve: keep it off a
        // subset of the CUs when SFFT_S2_CUMASK is set (bit pattern per 32 CUs) so that the solver's small,
        // latency-bound launches always find free CUs; fall back to a low-priority stream if masking fails
        uint32_t pat = 0u;      // CU masking measured slower than a plain low-priority stream on MI355X; off by default
        hipDeviceProp_t prop;
        PLAN_HIP(hipGetDeviceProperties(&prop, device));
        p->num_cu = prop.multiProcessorCount;
        const int nwords = (prop.multiProcessorCount + 31) / 32;
        std::vector<uint32_t> mask(nwords, pat);
        if (pat == 0 || hipExtStreamCreateWithCUMask(&p->s2, (uint32_t)nwords, mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            PLAN_HIP(hipStreamCreateWithPriority(&p->s2, hipStreamNonBlocking, lo));
        }
        PLAN_HIP(hipEventCreateWithFlags(&p->ev_in, hipEventDisableTiming));
        PLAN_HIP(hipEventCreateWithFlags(&p->ev_pre, hipEventDisableTiming));
        PLAN_HIP(hipEventCreateWithFlags(&p->ev_mom, hipEventDisableTiming));
        PLAN_HIP(hipEventCreateWithFlags(&p->ev_gam, hipEventDisableTiming));
    }
    {
        std::vector<double> ones((size_t)std::max(N0, N1), 1.0);
        PLAN_TRY(dev_alloc(p, &p->d_ones, ones.size()));
        PLAN_HIP(hipMemcpy(p->d_ones, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    {   // basis tables on the device
        PLAN_TRY(dev_alloc(p, &p->d_kbx, (size_t)BS.nkx * N0));
        PLAN_TRY(dev_alloc(p, &p->d_kby, (size_t)BS.nky * N1));
        PLAN_TRY(dev_alloc(p, &p->d_tbx, (size_t)BS.nbx * N0));
        PLAN_TRY(dev_alloc(p, &p->d_tby, (size_t)BS.nby * N1));
        PLAN_HIP(hipMemcpy(p->d_kbx, BS.kbx.data(), (size_t)BS.nkx * N0 * sizeof(double), hipMemcpyHostToDevice));
        // support of every kernel row factor (B-spline factors vanish outside a few knot spans): the weighted column pass skips the rest
        p->kbx_lo.assign(BS.nkx, 0); p->kbx_hi.assign(BS.nkx, N0);
        for (int i = 0; i < BS.nkx; ++i) {
            const double* t = BS.kbx.data() + (size_t)i * N0;
            int lo = 0, hi = N0;
            while (lo < N0 && t[lo] == 0.0) ++lo;
            while (hi > lo && t[hi - 1] == 0.0) --hi;
            p->kbx_lo[i] = lo; p->kbx_hi[i] = hi;
        }
        PLAN_HIP(hipMemcpy(p->d_kby, BS.kby.data(), (size_t)BS.nky * N1 * sizeof(double), hipMemcpyHostToDevice));
        PLAN_HIP(hipMemcpy(p->d_tbx, BS.tbx.data(), (size_t)BS.nbx * N0 * sizeof(double), hipMemcpyHostToDevice));
        PLAN_HIP(hipMemcpy(p->d_tby, BS.tby.data(), (size_t)BS.nby * N1 * sizeof(double), hipMemcpyHostToDevice));
        memset(&p->sa, 0, sizeof(p->sa));
        if (p->mode == 3) {
            PLAN_TRY(dev_alloc(p, &p->d_sbx, (size_t)BS.nsx * N0));
            PLAN_TRY(dev_alloc(p, &p->d_sby, (size_t)BS.nsy * N1));
            PLAN_HIP(hipMemcpy(p->d_sbx, BS.sbx.data(), (size_t)BS.nsx * N0 * sizeof(double), hipMemcpyHostToDevice));
            PLAN_HIP(hipMemcpy(p->d_sby, BS.sby.data(), (size_t)BS.nsy * N1 * sizeof(double), hipMemcpyHostToDevice));
            p->sa.nsca = p->nsca; p->sa.Fab = p->Fab; p->sa.cen = KerHW * p->L + KerHW; p->sa.sbx = p->d_sbx; p->sa.sby = p->d_sby;
            for (int t = 0; t < p->nsca; ++t) { p->sa.sp[t] = BS.spair[2 * t]; p->sa.sq[t] = BS.spair[2 * t + 1]; }
        }
        memset(&p->bk, 0, sizeof(p->bk));
        p->bk.npq = p->Fpq; p->bk.nq = BS.nby; p->bk.tbx = p->d_tbx; p->bk.tby = p->d_tby;
        for (int t = 0; t < p->Fpq; ++t) { p->bk.p[t] = BS.bpair[2 * t]; p->bk.q[t] = BS.bpair[2 * t + 1]; }
    }
    PLAN_TRY(build_axis(p, p->ax0, N0, false, true));
    PLAN_TRY(build_axis(p, p->ax1, N1));
    p->lay = rowmajor_layout(p->Nhp);
    {   // panel layout: whenever both passes run on chip (the four-step kernels are row-major)
        int pw = 4;     // measured at 4096^2: 2 is best for the column pass alone (0.60 -> 0.40 ms) but slows every row-wise
                        // consumer (32 lines per wave load); 4 keeps them at speed and still gives 0.60 -> 0.47 ms; 8 gains nothing
        if (const char* ev = getenv("SFFT_PANEL")) pw = atoi(ev);
        const bool both_fast = !p->no_fast_fft && !p->ax0.big && !p->ax0.blue && p->ax0.M == 4096 && !p->ax1.big && !p->ax1.blue && p->ax1.M == 4096;
        const bool both_onchip = !p->ax0.big && !p->ax1.big;
        if ((both_fast || both_onchip) && pw > 1 && is_pow2(pw) && p->Nhp % pw == 0) {
            p->lay.shift = ilog2(pw); p->lay.mask = pw - 1; p->lay.rstride = pw; p->lay.pstride = (long long)N0 * pw;
        }
        // polynomial plans (DK >= 0: made by sfft_plan_create, REF_ij term order) with a stamp of at most 25 x 25 take the
        // mixed-domain apply, whatever the image shape: it needs the row pass only, and no transform along axis 0 at all
        p->staged_solve = both_fast && !p->no_staged;
        // (the pair-per-workgroup column pass: the staged 4096^2 solve pass on 4-column panels; its 2-column spectra are read by the Greek launches only --
        //  plans whose apply pass reads d_spec keep the four-column kernel, see below)
        if (!(both_fast && !p->no_staged && p->colq && p->lay.mask == 3 && p->lay.rstride == 4)) p->colz = 0;
        if (DK >= 0 && DK <= 3 && p->mode != 3 && KerHW >= 1 && KerHW <= 12 && !env_no_vconv()) {
            p->use_vconv = 1;
            p->vncf = DK + 1;
            p->vw = KerHW <= 4 ? 4 : KerHW <= 8 ? 8 : 12;
            PLAN_TRY(dev_alloc(p, &p->d_stage_a, (size_t)(DK + 1) * N0 * p->Nhp));
            PLAN_TRY(dev_alloc(p, &p->d_ctabm, (size_t)p->Fij * (2 * p->vw + 1) * p->Nhp + 256));   // + 256: trash slots of vconv_mixed
        }
        // basis plans whose terms are the full tensor product of nkx row factors and nky column factors in (ii, jj) order -- the B-spline
        // kernels of BSplineSFFT -- take the same route through vconv_tensor (4 x 4 .. 6 x 6 terms, KerHW <= 8)
        if (DK < 0 && p->mode != 3 && KerHW >= 1 && KerHW <= 8 && BS.nkx == BS.nky && BS.nkx >= 4 && BS.nkx <= 6 && p->Fij == BS.nkx * BS.nky &&
            !env_no_vconv()) {
            bool tensor = true;
            for (int t = 0; t < p->Fij; ++t) tensor = tensor && BS.kpair[2 * t] == t / BS.nky && BS.kpair[2 * t + 1] == t % BS.nky;
            if (tensor) {
                p->use_vconv = 1; p->vtensor = BS.nkx; p->vncf = BS.nky;
                p->vw = KerHW <= 4 ? 4 : 8;
                PLAN_TRY(dev_alloc(p, &p->d_stage_a, (size_t)p->vncf * N0 * p->Nhp));
                PLAN_TRY(dev_alloc(p, &p->d_ctabm, (size_t)p->Fij * (2 * p->vw + 1) * p->Nhp + 256));
                // quadratic B-splines: three consecutive row factors per row (vconv_tensor's NA); anything wider sums over all of them
                std::vector<int> ibase((size_t)N0, 0);
                int span = 0;
                for (int l = 0; l < N0; ++l) {
                    int lo = BS.nkx, hi = -1;
                    for (int i = 0; i < BS.nkx; ++i) if (BS.kbx[(size_t)i * N0 + l] != 0.0) { lo = std::min(lo, i); hi = std::max(hi, i); }
                    if (hi < 0) { lo = 0; hi = 0; }
                    span = std::max(span, hi - lo + 1);
                    ibase[l] = std::min(lo, BS.nkx - 3);
                }
                if (span <= 3 && BS.nkx > 3) {
                    p->vt_na = 3;
                    PLAN_TRY(dev_alloc(p, &p->d_ibase, ibase.size()));
                    PLAN_HIP(hipMemcpy(p->d_ibase, ibase.data(), ibase.size() * sizeof(int), hipMemcpyHostToDevice));
                }
            }
        }
    }
    // the solve pass's spectra: 2-column panels behind cols_fwd_weighted_4096_z, the plan's layout otherwise.  Without the mixed-domain apply
    // the apply pass transforms into (and Construct_FDIFF reads) d_spec in the plan's layout, and a call with I as its own mask reuses the solve pass's spectra.
    if (!p->use_vconv) p->colz = 0;
    p->lay_spec = p->lay;
    if (p->colz) { p->lay_spec.shift = 1; p->lay_spec.mask = 1; p->lay_spec.rstride = 2; p->lay_spec.pstride = (long long)N0 * 2; }
    // launch geometry of the on-chip FFT kernels (axes that need the four-step path use strided_dft instead)
    if (!p->ax1.big) {
        p->nt_rows = fft_threads(p->ax1.M);
        p->lds_rows = (size_t)axis_lds_len(p->ax1) * sizeof(cplx);
    }
    auto r24_q = [](const AxisHost& ax) { return (ax.big || ax.blue) ? 0 : ax.M == 6144 ? 16 : ax.M == 9216 ? 24 : 0; };
    p->rows_r24 = (p->no_fast_fft || getenv("SFFT_NO_ROWS_R24")) ? 0 : r24_q(p->ax1);
    p->cols_r24 = (p->no_fast_fft || getenv("SFFT_NO_COLS_R24")) ? 0 : r24_q(p->ax0);
    if (const char* e = getenv("SFFT_COLS_R24_PAIR")) p->cols_r24_pair = atoi(e) != 0;
    if (p->lay.mask < 1) p->cols_r24_pair = false;          // (row-major planes: neighbouring columns are not neighbours in memory)
    if (const char* ev = getenv("SFFT_SPEC2")) p->spec2 = atoi(ev);
    if (!(p->use_vconv && !p->no_staged && p->cols_r24 == 16 && p->cols_r24_pair && p->lay.mask == 3 && p->lay.rstride == 4)) p->spec2 = 0;
    if (p->spec2) { p->lay_spec.shift = 1; p->lay_spec.mask = 1; p->lay_spec.rstride = 2; p->lay_spec.pstride = (long long)N0 * 2; }
    PLAN_HIP(hipFuncSetAttribute((const void*)rows_r2c_r24<24>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)rows_c2r_diff_r24<24, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)cols_fwd_weighted_r24<24>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)cols_fwd_weighted_r24<16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (!p->ax0.big) {
        pick_col_tile(p->ax0, &p->TC, &p->MS);
        p->nt_cols = fft_threads(p->TC * p->ax0.M);
        p->lds_cols = (size_t)p->TC * p->MS * sizeof(cplx);
    }
    PLAN_HIP(hipFuncSetAttribute((const void*)strided_dft, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)strided_rader577, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (p->ax1.big) {          // two packed-row work arrays [ceil(N0/2)][N1]
        PLAN_TRY(dev_alloc(p, &p->d_big1, (size_t)((N0 + 1) / 2) * N1));
        PLAN_TRY(dev_alloc(p, &p->d_big2, (size_t)((N0 + 1) / 2) * N1));
    }
    if (p->ax0.big) {
        // 16 x B column axes: the outputs of a stage plane share the first pass, each through a scratch plane of its own (up to DFT16_MAX_OUT)
        const bool multi = !p->ax0.bigblue && p->ax0.A == 16 && p->ax0.subA && p->ax0.subA->M == 16 && !p->ax0.subA->blue && !p->no_dft16_regs;
        p->colscr_planes = multi ? std::min(DFT16_MAX_OUT, std::max(1, p->nkx)) : 1;
        PLAN_TRY(dev_alloc(p, &p->d_colscr, (size_t)p->colscr_planes * N0 * p->Nhp));
    }
    PLAN_HIP(hipFuncSetAttribute((const void*)rows_r2c, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)rows_r2c_4096, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)cols_c2c_4096, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)cols_fwd_weighted_4096_q, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)cols_fwd_weighted_4096_z, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)cols_fwd_weighted_4096, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)rows_c2r_diff_4096<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)rows_c2r_diff_4096<SFFT_MAX_BQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)cols_c2c, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)cols_fwd_weighted, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)rows_c2r_diff<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)rows_c2r_diff<SFFT_MAX_BQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (p->NEQfs > LU_MAX_ROWS) {       // (the pivoted-LU panel runs on at most 16 workgroups of 1536 rows)
        sfft_plan_destroy(p);
        return set_err(SFFT_ERR_UNSUPPORTED_SIZE, "linear system too large for the pivoted-LU panel of this build");
    }

    // index map of Remove_LSFStripes (SFFTSubtract.py:83-90); with tied scaling (mode 2) the kept entry ij00[0]
    // stands for the whole tied group
    if (p->has_idx) {
        std::vector<int> idx;
        std::vector<char> forb(p->NEQ, 0);
        const int ij00_first = KerHW * p->L + KerHW;
        // mode 3: the place-holder scaling terms ij00[ScaFij:] leave the system (BSplineSFFT.py:3732-3750)
        for (int ij = (p->mode == 3 ? p->nsca : 1); ij < p->Fij; ++ij) forb[ij00_first + ij * p->Fab] = 1;
        for (int r = 0; r < p->NEQ; ++r) if (!forb[r]) idx.push_back(r);
        PLAN_TRY(dev_alloc(p, &p->d_idx, idx.size()));
        PLAN_HIP(hipMemcpy(p->d_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    // PHI block: PrePHI[p'q',pq][0][0] = SCALE * sum_x T_p'q' T_pq  (SFFTSubtract.py:680-694); T is separable, so
    // the sum factors into one sum per axis
    {
        std::vector<long double> Gx((size_t)BS.nbx * BS.nbx), Gy((size_t)BS.nby * BS.nby);
        for (int a = 0; a < BS.nbx; ++a) for (int b = 0; b < BS.nbx; ++b) {
            long double acc = 0.0L;
            for (int x = 0; x < N0; ++x) acc += (long double)BS.tbx[(size_t)a * N0 + x] * (long double)BS.tbx[(size_t)b * N0 + x];
            Gx[(size_t)a * BS.nbx + b] = acc;
        }
        for (int a = 0; a < BS.nby; ++a) for (int b = 0; b < BS.nby; ++b) {
            long double acc = 0.0L;
            for (int y = 0; y < N1; ++y) acc += (long double)BS.tby[(size_t)a * N1 + y] * (long double)BS.tby[(size_t)b * N1 + y];
            Gy[(size_t)a * BS.nby + b] = acc;
        }
        std::vector<double> phi((size_t)p->Fpq * p->Fpq);
        for (int a = 0; a < p->Fpq; ++a) for (int b = 0; b < p->Fpq; ++b)
            phi[(size_t)a * p->Fpq + b] = (double)((long double)p->scale * Gx[(size_t)BS.bpair[2 * a] * BS.nbx + BS.bpair[2 * b]]
                                                   * Gy[(size_t)BS.bpair[2 * a + 1] * BS.nby + BS.bpair[2 * b + 1]]);
        PLAN_TRY(dev_alloc(p, &p->d_phi, phi.size()));
        PLAN_HIP(hipMemcpy(p->d_phi, phi.data(), phi.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    // rank-1 spectra of T_pq: FT_pq[l][m] = SCALE * Xp[p][l] * Yq[q][m]
    // Gamma block in real space (gamma_rows): needs a polynomial background (its factors' shift identity); the kernel basis may be
    // polynomial (moments of combined degree) or tabulated (one moment per (column factor, background degree))
    int bkg_deg = DB;
    if (bkg_deg < 0 && BS.nby <= 4 && BS.nbx <= 4) {          // tabulated plan: is the background the power basis cx^p cy^q?
        bool poly = true;
        for (int q = 0; q < BS.nby && poly; ++q) for (int x = 0; x < N1; ++x) if (fabs(BS.tby[(size_t)q * N1 + x] - ipow_host((x + 1.0) / N1, q)) > 1e-14) { poly = false; break; }
        for (int e = 0; e < BS.nbx && poly; ++e) for (int x = 0; x < N0; ++x) if (fabs(BS.tbx[(size_t)e * N0 + x] - ipow_host((x + 1.0) / N0, e)) > 1e-14) { poly = false; break; }
        if (poly) bkg_deg = BS.nby - 1;
    }
    const int gam_nmu = (DK >= 0) ? DK + bkg_deg + 1 : BS.nky * (bkg_deg + 1);
    p->gamma_analytic = (bkg_deg >= 0 && p->mode != 3 && KerHW <= GAMMA_MAXW && gam_nmu <= (DK >= 0 ? 7 : GAMMA_ND) && p->Fij <= 64) ? 1 : 0;
    p->gam_tab = (DK >= 0) ? 0 : 1; p->gam_nmu = gam_nmu; p->gam_db = bkg_deg;
    std::vector<char> const_x(BS.nbx, 0);
    {
        PLAN_TRY(dev_alloc(p, &p->d_Xp, (size_t)BS.nbx * N0));
        PLAN_TRY(dev_alloc(p, &p->d_Yq, (size_t)BS.nby * p->Nhp));
        PLAN_HIP(hipMemset(p->d_Yq, 0, (size_t)BS.nby * p->Nhp * sizeof(cplx)));
        std::vector<cplx> v;
        bool cst;
        for (int e = 0; e < BS.nbx; ++e) {
            table_axis_dft(BS.tbx.data() + (size_t)e * N0, N0, N0, v, &cst);
            const_x[e] = cst;
            PLAN_HIP(hipMemcpy(p->d_Xp + (size_t)e * N0, v.data(), (size_t)N0 * sizeof(cplx), hipMemcpyHostToDevice));
        }
        for (int e = 0; e < BS.nby; ++e) {
            table_axis_dft(BS.tby.data() + (size_t)e * N1, N1, p->Nh, v, &cst);
            PLAN_HIP(hipMemcpy(p->d_Yq + (size_t)e * p->Nhp, v.data(), (size_t)p->Nh * sizeof(cplx), hipMemcpyHostToDevice));
        }
    }
    // Greek work lists.  G1 passes: Omega (i'j' <= ij), Theta (i'j'), Gamma column-factor passes (i'j', p).
    // Patch jobs (one lag patch each): Omega, Gamma (i'j', pq), Theta.
    {
        const int hO = 2 * KerHW, hG = KerHW;
        const int PHo = 2 * hO + 1, PHg = 2 * hG + 1;
        int S = 1;
        const int colblocks = (p->Nh + 63) / 64;
        const int nsca = p->nsca;
        const int npass_est = p->Fij * (p->Fij + 1) / 2 + p->Fij * BS.nbx + p->Fij
                              + nsca * p->Fij + nsca * (nsca + 1) / 2 + nsca * (BS.nbx + 1);
        const int JP = p->Fij;                                  // plane of J
        auto SP = [&](int s) { return p->Fij + 1 + s; };        // plane of scaling term s
        // (with the decimation step of the stage-1 kernels a wave covers its rows in half the steps: half the chunks -- and half the partial
        //  sums -- fill the chip as well; measured at 4096^2: 4 chunks 675 pairs/s, 8 chunks 665, 2 chunks 663)
        const long long fill_target = (p->g1_dit && N0 % 2 == 0) ? 3072 : 6144;
        while (S < 16 && (long long)colblocks * S * npass_est < fill_target && N0 / (2 * S) >= 64) S *= 2;
        if (const char* ev = getenv("SFFT_G1_S")) { const int v = atoi(ev); if (v >= 1 && v <= 16 && N0 / v >= 64) S = v; }      // A/B: row chunks of the Omega launch
        // The grouped matrix-core launch shares a tile's planes between its sibling waves through the XCD's L2, which works while the
        // siblings stay within a few hundred rows of each other: row chunks of at most ~1200 rows (4096 / 4 = 1024 at the headline size;
        // 9232 rows in one chunk: 11.2 ms for config 5's two launches, 8 chunks: see DESIGN).  With the decimation step a chunk is
        // rows_per_chunk / 2 rows x' and their partners, a whole number of 8-row steps: rows_per_chunk is a multiple of 16 and the last
        // chunk takes what is left (N0 a multiple of 16).
        if (p->g1_mfma >= 3 && hO >= 9 && hO <= 32 && p->g1_dit && N0 % 16 == 0 && !getenv("SFFT_G1_S")) {
            // ... as long as the per-chunk partial sums (pass x chunk x lag x column) stay a fraction of the planes the launch reads
            const double cap = 0.5 * (double)(p->Fij + 1) * N0 / ((double)(p->Fij * (p->Fij + 1) / 2) * PHo);
            while (S < 16 && N0 / S > 1200 && 2 * S <= cap) S *= 2;
        }
        p->rows_per_chunk = (N0 + S - 1) / S;
        if (p->g1_dit && N0 % 16 == 0 && p->rows_per_chunk % 16 != 0) {
            p->rows_per_chunk = (p->rows_per_chunk + 15) / 16 * 16;
            S = (N0 + p->rows_per_chunk - 1) / p->rows_per_chunk;
        }
        p->S = S;
        long long goff = 0;
        auto add_pass = [&](int a, int b, int bp, int h) {
            G1Pass d; d.a_plane = a; d.b_plane = b; d.bp = bp; d.h = h; d.gp_off = goff; d.gp_off2 = 0; d.dual = 0;
            p->passes.push_back(d); goff += (long long)S * (2 * h + 1) * p->Nhp;
            return (int)p->passes.size() - 1;
        };
        std::vector<int> omg_pass, the_pass, gam_pass((size_t)p->Fij * BS.nbx);
        std::vector<int> sk_pass, ss_pass, st_pass, sg_pass((size_t)nsca * BS.nbx);
        // Omega passes.  On the matrix-core path the diagonal passes (a, a) go in pairs: one "dual" pass carries |A_a|^2 and |A_b|^2
        // (greek_g1_mfma, DG) and the partner's record only owns its partial buffer; partner records sit behind the launched ones.
        const bool dual_diag = p->g1_mfma && hO >= 9 && (hO <= 16 || (hO <= 32 && p->g1_mfma >= 3)) && p->Fij >= 2;
        omg_pass.assign((size_t)p->Fij * (p->Fij + 1) / 2, -1);
        auto okey = [&](int a, int b) { return a * p->Fij - (a * (a - 1)) / 2 + (b - a); };      // (a <= b) -> k, the job / patch order
        std::vector<std::pair<int, int>> partners;     // (leader pass, partner plane)
        // Products of terms whose row (or column) factors have (nearly) disjoint supports -- B-spline bases -- are not transformed: their
        // patches are a few 1-D correlations in real space (omega_sparse, greek.hpp).  A product qualifies when at most 6 h positions of
        // the sparse axis can carry a nonzero weight for some lag |u| <= h.  Tabulated bases without scaling planes, h <= 16.
        std::vector<char> omg_sparse(omg_pass.size(), 0);
        std::vector<int> slists;              // per (product, lag): the positions on the sparse axis whose weight is nonzero
        if (DK < 0 && p->mode != 3 && hO >= 1 && hO <= OSP_HP && N0 % 4 == 0 && N1 % 4 == 0 && N0 >= 64 && N1 >= 64 && !getenv("SFFT_NO_OMG_SPARSE")) {
            auto hull = [](const double* t, int N, int* lo, int* hi) { *lo = 0; *hi = N; while (*lo < N && t[*lo] == 0.0) ++*lo; while (*hi > *lo && t[*hi - 1] == 0.0) --*hi; };
            auto candidates = [&](const double* ta, const double* tb, int N, std::vector<int>& out) {
                int la, ha, lb, hb;
                hull(ta, N, &la, &ha); hull(tb, N, &lb, &hb);
                out.clear();
                if (ha - la > N / 2 + 2 * hO && hb - lb > N / 2 + 2 * hO) return false;        // (both wide: dense for sure)
                for (int c = la; c < ha; ++c) {
                    bool any = false;
                    for (int u = -hO; u <= hO && !any; ++u) { int cb = c + u; cb = cb < 0 ? cb + N : (cb >= N ? cb - N : cb); any = cb >= lb && cb < hb; }
                    if (any) { out.push_back(c); if ((int)out.size() > 6 * hO) return false; }
                }
                return true;
            };
            std::vector<int> cand;
            for (int a = 0; a < p->Fij; ++a) for (int b = a + 1; b < p->Fij; ++b) {
                if (dual_diag && a % 2 == 0 && b == a + 1) continue;      // the edge of a dual-diagonal group: its wave carries two Theta passes, keep it
                const int ia = BS.kpair[2 * a], ja = BS.kpair[2 * a + 1], ib = BS.kpair[2 * b], jb = BS.kpair[2 * b + 1];
                SparseProd sp; memset(&sp, 0, sizeof(sp));
                bool ok = candidates(BS.kbx.data() + (size_t)ia * N0, BS.kbx.data() + (size_t)ib * N0, N0, cand);
                if (ok) { sp.ymode = 0; sp.fa_cross = ia; sp.fb_cross = ib; sp.fa_line = ja; sp.fb_line = jb; }
                else {
                    ok = candidates(BS.kby.data() + (size_t)ja * N1, BS.kby.data() + (size_t)jb * N1, N1, cand);
                    if (ok) { sp.ymode = 1; sp.fa_cross = ja; sp.fb_cross = jb; sp.fa_line = ia; sp.fb_line = ib; }
                }
                if (!ok) continue;
                {   // per lag u: the candidates whose weight is nonzero
                    const int NSx = sp.ymode ? N1 : N0;
                    const double* ta = (sp.ymode ? BS.kby.data() : BS.kbx.data()) + (size_t)sp.fa_cross * NSx;
                    const double* tb = (sp.ymode ? BS.kby.data() : BS.kbx.data()) + (size_t)sp.fb_cross * NSx;
                    for (int u = -hO; u <= hO; ++u) {
                        sp.ustart[u + hO] = (int)slists.size();
                        for (int c : cand) { int cb = c + u; cb = cb < 0 ? cb + NSx : (cb >= NSx ? cb - NSx : cb); if (ta[c] != 0.0 && tb[cb] != 0.0) slists.push_back(c); }
                    }
                    sp.ustart[2 * hO + 1] = (int)slists.size();
                    // the span of a line worth visiting: the support of fA, cut by the support of fB widened by h when that does not wrap
                    const int NLx = sp.ymode ? N0 : N1;
                    const double* fa = (sp.ymode ? BS.kbx.data() : BS.kby.data()) + (size_t)sp.fa_line * NLx;
                    const double* fb = (sp.ymode ? BS.kbx.data() : BS.kby.data()) + (size_t)sp.fb_line * NLx;
                    int la, ha, lb, hb;
                    hull(fa, NLx, &la, &ha); hull(fb, NLx, &lb, &hb);
                    sp.t0 = la & ~3; sp.t1 = std::min(NLx, (ha + 3) & ~3);
                    if (hb <= lb) sp.t1 = sp.t0;
                    else if (lb - hO >= 0 && hb + hO <= NLx) { sp.t0 = std::max(sp.t0, (lb - hO) & ~3); sp.t1 = std::min(sp.t1, (hb + hO + 3) & ~3); }
                    if (sp.t1 < sp.t0) sp.t1 = sp.t0;
                }
                sp.patch_off = okey(a, b);                  // (product index for now: turned into the patch offset once the jobs exist)
                p->sprods.push_back(sp);
                omg_sparse[okey(a, b)] = 1;
            }
        }
        for (int a = 0; a < p->Fij; ++a) for (int b = a; b < p->Fij; ++b) {
            if (omg_sparse[okey(a, b)]) continue;        // no pass, no partial buffer, no stage-2 job
            if (a != b || !dual_diag) { omg_pass[okey(a, b)] = add_pass(a, b, 0, hO); continue; }
            if (a % 2 == 0 && a + 1 < p->Fij) {          // leader of the pair (a, a + 1)
                const int lead = add_pass(a, a + 1, 0, hO);
                p->passes[lead].dual = 1;
                omg_pass[okey(a, a)] = lead;
                partners.push_back({lead, a + 1});
            } else if (a % 2 == 0) omg_pass[okey(a, a)] = add_pass(a, a, 0, hO);      // odd Fij: the last diagonal stays an ordinary pass
        }
        p->n_omg_launch = (int)p->passes.size();
        for (auto& pr : partners) {
            const int rec = add_pass(pr.second, pr.second, 0, hO);
            p->passes[pr.first].gp_off2 = p->passes[rec].gp_off;
            omg_pass[okey(pr.second, pr.second)] = rec;
        }
        p->n_omg = (int)omg_pass.size();
        for (int a = 0; a < p->Fij; ++a) for (int b = a; b < p->Fij; ++b) if (omg_pass[okey(a, b)] >= 0) ++(a == b ? p->n_omg_diag : p->n_omg_off);
        // passes of half width w through greek_g1: Theta, dense Gamma column factors, then the scaling planes' passes
        const int dense0 = (int)p->passes.size();
        p->n_omg_rec = dense0;
        for (int a = 0; a < p->Fij; ++a) the_pass.push_back(add_pass(a, JP, 0, hG));
        p->n_the = p->Fij;
        p->n_gamp = 0; p->n_gam0 = 0;
        if (!p->gamma_analytic)
        for (int a = 0; a < p->Fij; ++a) for (int e = 0; e < BS.nbx; ++e) if (!const_x[e]) { gam_pass[(size_t)a * BS.nbx + e] = add_pass(a, -1, e, hG); ++p->n_gamp; }
        for (int sI = 0; sI < nsca; ++sI) for (int b = 0; b < p->Fij; ++b) sk_pass.push_back(add_pass(SP(sI), b, 0, hG));
        for (int sI = 0; sI < nsca; ++sI) for (int t = sI; t < nsca; ++t) ss_pass.push_back(add_pass(SP(sI), SP(t), 0, hG));
        for (int sI = 0; sI < nsca; ++sI) st_pass.push_back(add_pass(SP(sI), JP, 0, hG));
        for (int sI = 0; sI < nsca; ++sI) for (int e = 0; e < BS.nbx; ++e) if (!const_x[e]) sg_pass[(size_t)sI * BS.nbx + e] = add_pass(SP(sI), -1, e, hG);
        p->n_dense_w = (int)p->passes.size() - dense0;
        // column factors that are the constant 1 (Xp = N0 * delta): only spectrum row 0 contributes (greek_g1_row0)
        const int row0 = (int)p->passes.size();
        if (!p->gamma_analytic)
        for (int a = 0; a < p->Fij; ++a) for (int e = 0; e < BS.nbx; ++e) if (const_x[e]) { gam_pass[(size_t)a * BS.nbx + e] = add_pass(a, -1, e, hG); ++p->n_gam0; }
        for (int sI = 0; sI < nsca; ++sI) for (int e = 0; e < BS.nbx; ++e) if (const_x[e]) sg_pass[(size_t)sI * BS.nbx + e] = add_pass(SP(sI), -1, e, hG);
        p->n_row0 = (int)p->passes.size() - row0;
        int poff = 0;
        auto add_job = [&](int pass, int yq, int h, double scale) {
            PatchJob j; j.pass = pass; j.yq = yq; j.h = h; j.patch_off = poff; j.scale = scale;
            p->jobs.push_back(j); poff += (2 * h + 1) * (2 * h + 1);
        };
        p->fa.omg_off = poff;
        for (int k = 0; k < p->n_omg; ++k) add_job(omg_pass[k], -1, hO, p->scale * p->scale);   // PreOMG = SCALE*Re[SCALE*DFT] (SFFTSubtract.py:233-240)
        p->fa.gam_off = poff;
        for (int a = 0; a < p->Fij; ++a) for (int q = 0; q < p->Fpq; ++q) {
            if (p->gamma_analytic) poff += (2 * hG + 1) * (2 * hG + 1);       // same patch slot, filled by gamma_patches
            else add_job(gam_pass[(size_t)a * BS.nbx + BS.bpair[2 * q]], BS.bpair[2 * q + 1], hG, p->scale);   // PreGAM = Re[SCALE*DFT] (:262-268)
        }
        p->n_gam = p->Fij * p->Fpq;
        p->fa.the_off = poff;
        for (int a = 0; a < p->Fij; ++a) add_job(the_pass[a], -1, hG, p->scale);                  // PreTHE = Re[SCALE*DFT] (:353-362)
        // scaling planes (BSplineSFFT.py:3293-3565): OMG01/10 (scaling x kernel), OMG00, GAM0/PSI0, THE0
        p->fa.sk_off = poff;
        for (size_t k = 0; k < sk_pass.size(); ++k) add_job(sk_pass[k], -1, hG, p->scale * p->scale);
        p->fa.ss_off = poff;
        for (size_t k = 0; k < ss_pass.size(); ++k) add_job(ss_pass[k], -1, hG, p->scale * p->scale);
        p->fa.sg_off = poff;
        for (int sI = 0; sI < nsca; ++sI) for (int q = 0; q < p->Fpq; ++q)
            add_job(sg_pass[(size_t)sI * BS.nbx + BS.bpair[2 * q]], BS.bpair[2 * q + 1], hG, p->scale);
        p->fa.st_off = poff;
        for (int sI = 0; sI < nsca; ++sI) add_job(st_pass[sI], -1, hG, p->scale);
        p->fa.sv = (p->mode == 3) ? 1 : 0; p->fa.nsca = nsca;
        p->fa.reg_coef = 0.0; p->fa.ireg = nullptr; p->fa.sst = p->fa.csst = p->fa.dsst = nullptr;
        p->n_patches = poff;
        if (!p->sprods.empty()) {
            // y-mode products read image COLUMNS: the few columns involved (candidates and their lag partners) go through a transposed strip
            std::vector<int> colidx((size_t)N1, -1), cols;
            for (SparseProd& sp : p->sprods) {
                sp.patch_off = p->jobs[sp.patch_off].patch_off;
                if (!sp.ymode) continue;
                for (int u = -hO; u <= hO; ++u)
                    for (int k = sp.ustart[u + hO]; k < sp.ustart[u + hO + 1]; ++k)
                        for (int c0 : {slists[k], slists[k] + u}) {
                            const int c = c0 < 0 ? c0 + N1 : (c0 >= N1 ? c0 - N1 : c0);
                            if (colidx[c] < 0) { colidx[c] = (int)cols.size(); cols.push_back(c); }
                        }
            }
            p->n_scols = (int)cols.size();
            PLAN_TRY(dev_alloc(p, &p->d_sprods, p->sprods.size()));
            PLAN_HIP(hipMemcpy(p->d_sprods, p->sprods.data(), p->sprods.size() * sizeof(SparseProd), hipMemcpyHostToDevice));
            std::vector<SparseLine> slines(slists.size());
            for (const SparseProd& sp : p->sprods) {
                const int NSx = sp.ymode ? N1 : N0;
                const double* ta = (sp.ymode ? BS.kby.data() : BS.kbx.data()) + (size_t)sp.fa_cross * NSx;
                const double* tb = (sp.ymode ? BS.kby.data() : BS.kbx.data()) + (size_t)sp.fb_cross * NSx;
                for (int u = -hO; u <= hO; ++u)
                    for (int k = sp.ustart[u + hO]; k < sp.ustart[u + hO + 1]; ++k) {
                        const int c = slists[k], c1 = c + u, cb = c1 < 0 ? c1 + NSx : (c1 >= NSx ? c1 - NSx : c1);
                        slines[k].w = ta[c] * tb[cb];
                        slines[k].a_off = sp.ymode ? (long long)colidx[c] * N0 : (long long)c * N1;
                        slines[k].b_off = sp.ymode ? (long long)colidx[cb] * N0 : (long long)cb * N1;
                    }
            }
            PLAN_TRY(dev_alloc(p, &p->d_slines, slines.size()));
            PLAN_HIP(hipMemcpy(p->d_slines, slines.data(), slines.size() * sizeof(SparseLine), hipMemcpyHostToDevice));
            if (p->n_scols) {
                PLAN_TRY(dev_alloc(p, &p->d_scols, cols.size()));
                PLAN_HIP(hipMemcpy(p->d_scols, cols.data(), cols.size() * sizeof(int), hipMemcpyHostToDevice));
                PLAN_TRY(dev_alloc(p, &p->d_strip, (size_t)p->n_scols * N0));
            }
            // The items (product, lag), ordered so that those sharing image rows -- same mode and cross factors -- are neighbours, dealt to the
            // eight XCDs in contiguous runs of equal cost (lines x span); within a run the costly items first.  Items without lines stay
            // in (they write the zeros of their patch row).
            struct Item { int prod, u; long long cost; };
            std::vector<Item> its;
            for (size_t i = 0; i < p->sprods.size(); ++i) {
                const SparseProd& sp = p->sprods[i];
                for (int u = -hO; u <= hO; ++u) its.push_back({(int)i, u, (long long)(sp.ustart[u + hO + 1] - sp.ustart[u + hO]) * (sp.t1 - sp.t0)});
            }
            std::stable_sort(its.begin(), its.end(), [&](const Item& x, const Item& y) {
                const SparseProd& a = p->sprods[x.prod]; const SparseProd& b = p->sprods[y.prod];
                return std::make_tuple(a.ymode, a.fa_cross, a.fb_cross) < std::make_tuple(b.ymode, b.fa_cross, b.fb_cross);
            });
            long long total = 0;
            for (const Item& it : its) total += it.cost + 1;
            std::vector<std::vector<Item>> runs(8);
            { long long run = 0; for (const Item& it : its) { runs[std::min<long long>(7, run * 8 / total)].push_back(it); run += it.cost + 1; } }
            size_t per = 0;
            for (auto& r : runs) { std::stable_sort(r.begin(), r.end(), [](const Item& x, const Item& y) { return x.cost > y.cost; }); per = std::max(per, r.size()); }
            std::vector<int2> items(8 * per, make_int2(-1, 0));
            for (int x = 0; x < 8; ++x) for (size_t i = 0; i < runs[x].size(); ++i) items[x * per + i] = make_int2(runs[x][i].prod, runs[x][i].u);
            p->n_sitems = (int)items.size();
            PLAN_TRY(dev_alloc(p, &p->d_sitems, items.size()));
            PLAN_HIP(hipMemcpy(p->d_sitems, items.data(), items.size() * sizeof(int2), hipMemcpyHostToDevice));
        }
        (void)PHo; (void)PHg;
        if (p->g1_mfma >= 3 && hO >= 9 && hO <= 32) {
            // Pass groups of the Omega launch (greek_g1_mfma4g): an edge (x, y) with the dual-diagonal pass of the same two planes;
            // then triangles (x,y), (y,z), (x,z) among the remaining ordinary passes, greedily; then pairs of passes that share a
            // plane; then single passes.
            const int nl = p->n_omg_launch;
            std::vector<char> used((size_t)nl, 0);
            std::vector<G1Group> groups;
            // slots have fixed operand pairs: 0 = (v0, v1), 1 = (v1, v2), 2 = (v0, v2); passes are oriented a_plane <= b_plane
            auto find_edge = [&](int x, int y) {      // the unused ordinary pass (x, y), x < y
                for (int k = 0; k < nl; ++k) {
                    const G1Pass& q = p->passes[k];
                    if (!used[k] && !q.dual && q.a_plane == x && q.b_plane == y) return k;
                }
                return -1;
            };
            auto blank = [&](int v0, int v1, int v2) {
                G1Group g; g.plane[0] = v0; g.plane[1] = v1; g.plane[2] = v2; g.mask = 0;
                g.pass[0] = g.pass[1] = g.pass[2] = 0;
                g.tpass[0] = g.tpass[1] = -1; g.ht = 0;
                return g;
            };
            auto put = [&](G1Group& g, int sl, int k) { g.mask |= 1 << sl; g.pass[sl] = k; used[k] = 1; };
            for (int k = 0; k < nl; ++k) if (p->passes[k].dual && !used[k]) {          // the two diagonals of (x, y) [slot 0] + the edge (x, y) [slot 2]
                const int x = p->passes[k].a_plane, y = p->passes[k].b_plane;
                G1Group g = blank(x, y, x);          // (third plane = first: "two planes"; slot 2 then reads (v0, v1) as well)
                put(g, 0, k);
                const int e = find_edge(std::min(x, y), std::max(x, y));
                if (e >= 0 && x < y) put(g, 2, e);
                groups.push_back(g);
            }
            for (int k = 0; k < nl; ++k) if (!used[k] && p->passes[k].a_plane < p->passes[k].b_plane) {      // triangles x < y < z
                const int u = p->passes[k].a_plane, v = p->passes[k].b_plane;
                for (int w3 = 0; w3 < p->Fij && !used[k]; ++w3) {
                    if (w3 == u || w3 == v) continue;
                    int t[3] = {u, v, w3};
                    std::sort(t, t + 3);
                    used[k] = 1;                        // (so that find_edge skips it)
                    int e[3] = {find_edge(t[0], t[1]), find_edge(t[1], t[2]), find_edge(t[0], t[2])};
                    used[k] = 0;
                    for (int q = 0; q < 3; ++q) { const int ea = q == 1 ? t[1] : t[0], eb = q == 0 ? t[1] : t[2]; if (ea == u && eb == v) e[q] = k; }
                    if (e[0] >= 0 && e[1] >= 0 && e[2] >= 0) {
                        G1Group g = blank(t[0], t[1], t[2]);
                        put(g, 0, e[0]); put(g, 1, e[1]); put(g, 2, e[2]);
                        groups.push_back(g);
                    }
                }
            }
            for (int k = 0; k < nl; ++k) if (!used[k]) {                               // what is left: two passes that share a plane, single passes
                const int x = p->passes[k].a_plane, y = p->passes[k].b_plane;
                bool done = false;
                for (int k2 = k + 1; k2 < nl && !done && x < y; ++k2) if (!used[k2] && !p->passes[k2].dual && p->passes[k2].a_plane < p->passes[k2].b_plane) {
                    const int u = p->passes[k2].a_plane, v = p->passes[k2].b_plane;
                    int t[3], nt = 0;
                    for (int w3 : {x, y, u, v}) { bool in = false; for (int q = 0; q < nt; ++q) in = in || t[q] == w3; if (!in && nt < 3) t[nt++] = w3; else if (!in) nt = 4; }
                    if (nt != 3) continue;
                    std::sort(t, t + 3);
                    auto slot_for = [&](int a2, int b2) { return (a2 == t[0] && b2 == t[1]) ? 0 : (a2 == t[1] && b2 == t[2]) ? 1 : 2; };
                    G1Group g = blank(t[0], t[1], t[2]);
                    put(g, slot_for(x, y), k); put(g, slot_for(u, v), k2);
                    groups.push_back(g);
                    done = true;
                }
                if (!done) { G1Group g = blank(x, y, x); put(g, 0, k); groups.push_back(g); }
            }
            // The Theta passes (x, J), half width w <= 8, ride in the edge groups (x, y) + dual(x, y): the group then loads J as its third
            // plane, and the separate vector launch (which re-reads all Fij + 1 planes) disappears -- when every kernel plane has such a
            // group (Fij even) and the short-pass launch holds nothing but the Theta passes
            {
                bool ok = p->gamma_analytic && p->n_dense_w == p->n_the && hG <= 8;
                std::vector<int> owner((size_t)p->Fij, -1);
                if (ok) for (size_t gi = 0; gi < groups.size(); ++gi) {
                    const G1Group& g = groups[gi];
                    if ((g.mask & 1) && p->passes[g.pass[0]].dual && (g.mask & 4) && g.plane[2] == g.plane[0] && g.plane[0] < p->Fij && g.plane[1] < p->Fij) {
                        owner[g.plane[0]] = (int)gi; owner[g.plane[1]] = (int)gi;
                    }
                }
                // (odd Fij: the last plane has no dual partner, hence no such group; its Theta pass stays in a vector launch of its own)
                int nf = 0;
                while (nf < p->Fij && owner[nf] >= 0) ++nf;
                for (int a = nf; a < p->Fij && ok; ++a) ok = owner[a] < 0;
                ok = ok && nf > 0;
                if (ok) {
                    p->n_the_fused = nf;
                    for (int a = 0; a < nf; ++a) {
                        G1Group& g = groups[owner[a]];
                        g.tpass[g.plane[0] == a ? 0 : 1] = the_pass[a];
                        g.plane[2] = JP; g.ht = hG;
                    }
                    p->theta_in_groups = 1;
                    // odd Fij: the last plane's Theta pass (x, J) as an ordinary slot beside its lone diagonal pass (x, x): planes (x, x, J),
                    // slot 0 = (v0, v1) = the diagonal, slot 2 = (v0, v2) = the Theta pass with its own lag half width -- instead of a
                    // vector launch of its own that re-reads two planes (config 3: 0.32 ms)
                    if (nf == p->Fij - 1 && (env_theta_slots() != 0))
                        for (G1Group& g : groups) {
                            const G1Pass& q0 = p->passes[g.pass[0]];
                            if (g.mask == 1 && !q0.dual && q0.a_plane == nf && q0.b_plane == nf && g.tpass[0] < 0) {
                                g.plane[0] = nf; g.plane[1] = nf; g.plane[2] = JP;
                                g.mask |= 4; g.pass[2] = the_pass[nf];
                                p->n_the_fused = p->Fij;
                                break;
                            }
                        }
                }
            }
            // Half widths 9 .. 16 (KerHW 9 .. 16: config 5): too wide to ride as the edge groups' half slots, and their own launch
            // (greek_g1_mfma4<2>, one pass per wave) re-reads all Fij + 1 planes: 1.7 ms of a config-5 pair.  They become ordinary slots
            // of the FIRST launch (lags 1 .. 16) in groups of their own -- two passes (a, J), (a + 1, J) on the planes (a, a + 1, J) --
            // behind the Omega groups, which is where the later launches (lags 17 ..) stop.
            p->n_groups_omg = (int)groups.size();
            if (!p->theta_in_groups && p->gamma_analytic && p->n_dense_w == p->n_the && hG >= 9 && hG <= 16 && hO <= 32 &&
                (env_theta_slots() != 0)) {
                for (int a = 0; a + 1 < p->Fij; a += 2) {
                    G1Group g = blank(a, a + 1, JP);
                    g.mask = 6; g.pass[1] = the_pass[a + 1]; g.pass[2] = the_pass[a];          // slot 1 = (v1, v2), slot 2 = (v0, v2)
                    groups.push_back(g);
                }
                if (p->Fij % 2) {
                    G1Group g = blank(p->Fij - 1, JP, p->Fij - 1);
                    g.mask = 1; g.pass[0] = the_pass[p->Fij - 1];
                    groups.push_back(g);
                }
                p->n_the_fused = p->Fij;
                p->theta_slots = 1;
            }
            if (getenv("SFFT_G1_TRACE")) { PLAN_TRY(dev_alloc(p, &p->d_g1trace, (size_t)3 * 65536)); PLAN_HIP(hipMemset(p->d_g1trace, 0, (size_t)3 * 65536 * 8)); }
            p->n_groups = (int)groups.size();
            PLAN_TRY(dev_alloc(p, &p->d_groups, groups.size()));
            PLAN_HIP(hipMemcpy(p->d_groups, groups.data(), groups.size() * sizeof(G1Group), hipMemcpyHostToDevice));
        }
        PLAN_TRY(dev_alloc(p, &p->d_passes, p->passes.size()));
        PLAN_HIP(hipMemcpy(p->d_passes, p->passes.data(), p->passes.size() * sizeof(G1Pass), hipMemcpyHostToDevice));
        PLAN_TRY(dev_alloc(p, &p->d_jobs, p->jobs.size()));
        PLAN_HIP(hipMemcpy(p->d_jobs, p->jobs.data(), p->jobs.size() * sizeof(PatchJob), hipMemcpyHostToDevice));
        PLAN_TRY(dev_alloc(p, &p->d_gp, (size_t)goff));
        p->hm = std::max(g1_padded(hO), g1_padded(hG)) + 1;     // padded: a launch may compute (and drop) lags beyond h
        PLAN_TRY(dev_alloc(p, &p->d_w0tab, (size_t)N0 * p->hm));
        SFFT_LAUNCH(build_w0tab, dim3((N0 * p->hm + 255) / 256), dim3(256), 0, 0, p->ax0.root, p->d_w0tab, N0, p->hm);
        PLAN_TRY(dev_alloc(p, &p->d_patches, (size_t)poff));
        if (p->gamma_analytic) {
            // moment weights: cy^d (polynomial kernel: d = combined degree) or kby[j] cy^d (tabulated kernel: index j (DB + 1) + d)
            const int nd = p->gam_nmu, NQB = p->gam_db + 1;
            std::vector<double> cyp((size_t)nd * N1);
            for (int d = 0; d < nd; ++d) for (int x1 = 0; x1 < N1; ++x1)
                cyp[(size_t)d * N1 + x1] = p->gam_tab ? BS.kby[(size_t)(d / NQB) * N1 + x1] * ipow_host((x1 + 1.0) / N1, d % NQB) : ipow_host((x1 + 1.0) / N1, d);
            // the 4096^2 fast path with a polynomial kernel basis: the moments come out of the row pass of the forward transforms
            p->rowmom_fused = (!p->gam_tab && !p->no_fast_fft && !p->no_staged && N0 == 4096 && N1 == 4096 && nd <= ROWMOM_FUSED_MAX && p->nby <= ROWMOM_FUSED_MAX) ? 1 : 0;
            PLAN_TRY(dev_alloc(p, &p->d_cyp, cyp.size()));
            PLAN_HIP(hipMemcpy(p->d_cyp, cyp.data(), cyp.size() * sizeof(double), hipMemcpyHostToDevice));
            PLAN_TRY(dev_alloc(p, &p->d_rowmomI, (size_t)N0 * GAMMA_ND));
            PLAN_TRY(dev_alloc(p, &p->d_gamR, (size_t)(p->gam_tab ? BS.nky : DK + 1) * NQB * (2 * KerHW + 1) * N0));
            memset(&p->ga, 0, sizeof(p->ga));
            p->ga.Fij = p->Fij; p->ga.Fpq = p->Fpq; p->ga.w = KerHW;
            for (int k = 0; k < p->Fij; ++k) { p->ga.ki[k] = p->kpair[2 * k]; p->ga.kj[k] = p->kpair[2 * k + 1]; }
            for (int q = 0; q < p->Fpq; ++q) { p->ga.bp[q] = BS.bpair[2 * q]; p->ga.bq[q] = BS.bpair[2 * q + 1]; }
        }
        p->fa.Fij = p->Fij; p->fa.Fpq = p->Fpq; p->fa.Fab = p->Fab; p->fa.Fijab = p->Fijab; p->fa.L1 = p->L;
        p->fa.w0 = KerHW; p->fa.w1 = KerHW; p->fa.h_omg = hO; p->fa.h_gam = hG;
        p->fa.tie_first = KerHW * p->L + KerHW; p->fa.tie_stride = p->Fab; p->fa.tie_cnt = (p->mode == 2) ? p->Fij : 0;
    }
    PLAN_TRY(dev_alloc(p, &p->d_spec, (size_t)(p->Fij + 1 + p->nsca) * N0 * p->Nhp));
    p->ld = (p->NEQfs + 1 + 15) & ~15;          // rows are whole 128-byte lines (chol_dataflow hands tiles between workgroups)
    PLAN_TRY(dev_alloc(p, &p->d_A, (size_t)(p->NEQfs + 1) * p->ld));
    PLAN_TRY(dev_alloc(p, &p->d_dbuf, (size_t)3 * CB * CB));      // two hand-over blocks of the two-kernel path + the raw block of chol_step
    PLAN_TRY(dev_alloc(p, &p->d_xv, (size_t)p->NEQfs));
    PLAN_TRY(dev_alloc(p, &p->d_rd, (size_t)p->NEQfs));
    PLAN_TRY(dev_alloc(p, &p->d_partial, (size_t)BACK_SLICES * CB));
    {
        const int nblk_b = (p->NEQfs + CB - 1) / CB;
        PLAN_TRY(dev_alloc(p, &p->d_winv, (size_t)nblk_b * CB * CB));
        p->n_bflags = nblk_b + 1;                   // + 1: the hand-off flag of chol_step
        PLAN_TRY(dev_alloc(p, &p->d_bflags, (size_t)p->n_bflags));
        PLAN_HIP(hipMemset(p->d_bflags, 0, (size_t)p->n_bflags * sizeof(unsigned int)));
        PLAN_HIP(hipHostMalloc((void**)&p->h_status, sizeof(int), hipHostMallocDefault));
        *p->h_status = 0;
        PLAN_TRY(dev_alloc(p, &p->d_epoch, (size_t)1));
        PLAN_HIP(hipMemset(p->d_epoch, 0, sizeof(unsigned int)));
        PLAN_TRY(dev_alloc(p, &p->d_tflags, (size_t)nblk_b * (nblk_b + 1) + 1));
        PLAN_HIP(hipMemset(p->d_tflags, 0, ((size_t)nblk_b * (nblk_b + 1) + 1) * sizeof(unsigned int)));
        PLAN_TRY(dev_alloc(p, &p->d_w16, (size_t)nblk_b * 1024));
        PLAN_TRY(dev_alloc(p, &p->d_luperm, (size_t)LU_PERMS_PER_PANEL * nblk_b));
        if (p->NEQfs > 1024) {
            PLAN_TRY(dev_alloc(p, &p->d_luxchg, (lu_xchg_bytes() + 7) / 8));
            PLAN_HIP(hipMemset(p->d_luxchg, 0, lu_xchg_bytes()));
        }
        if (getenv("SFFT_DF_TRACE")) { PLAN_TRY(dev_alloc(p, &p->d_trace, (size_t)nblk_b * 16)); PLAN_HIP(hipMemset(p->d_trace, 0, (size_t)nblk_b * 16 * 8)); }
        PLAN_TRY(dev_alloc(p, &p->d_pq, (size_t)PANEL4_MAX_OUTER + 16));
        PLAN_HIP(hipMemset(p->d_pq, 0, ((size_t)PANEL4_MAX_OUTER + 16) * sizeof(unsigned int)));
        int ncu = 0;
        PLAN_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        p->ncu = ncu;
        if (nblk_b > ncu) p->back_variant = 0;      // the one-launch version needs every block's workgroup resident at once
    }
    PLAN_TRY(dev_alloc(p, &p->d_counter, (size_t)1));
    PLAN_HIP(hipMemset(p->d_counter, 0, sizeof(unsigned int)));
    PLAN_TRY(dev_alloc(p, &p->d_sol, (size_t)p->NEQ));
    p->wpad = KerHW <= 4 ? 4 : KerHW <= 8 ? 8 : KerHW <= 12 ? 12 : KerHW <= 16 ? 16 : KerHW <= 24 ? 24 : 32;
    PLAN_TRY(dev_alloc(p, &p->d_rtab, (size_t)p->Fij * N0 * (1 + 2 * p->wpad)));
    PLAN_HIP(hipMemset(p->d_rtab, 0, (size_t)p->Fij * N0 * (1 + 2 * p->wpad) * sizeof(cplx)));     // entries beyond w stay zero
    PLAN_TRY(dev_alloc(p, &p->d_rowmom, (size_t)N0 * SFFT_MAX_BQ));
    PLAN_TRY(dev_alloc(p, &p->d_delta, (size_t)p->Fpq));
    PLAN_TRY(dev_alloc(p, &p->d_status, (size_t)1));
    PLAN_HIP(hipMemset(p->d_status, 0, sizeof(int)));
    PLAN_HIP(hipDeviceSynchronize());
#undef PLAN_TRY
#undef PLAN_HIP
    *out = p;
    return SFFT_OK;
}

// SingleSFFTConfigure.SSC with polynomial spatial variation: kernel terms cx^i cy^j, i + j <= DK, background terms
// cx^p cy^q, p + q <= DB (REF_ij / REF_pq order, SFFTSubtract.py:62-64), constant scaling by stripe removal.
extern "C" int sfft_plan_create(sfft_plan** out, int N0, int N1, int KerHW, int DK, int DB, int cpr, int device)
{
    if (DK < 0 || DK > 3) return set_err(SFFT_ERR_INVALID_ARG, "Input KerPolyOrder should be 0/1/2/3!");
    if (DB < 0 || DB > 3) return set_err(SFFT_ERR_INVALID_ARG, "Input BGPolyOrder should be 0/1/2/3!");
    if (N0 < 8 || N1 < 8) return set_err(SFFT_ERR_INVALID_ARG, "Input Image has dramatically small size!");
    BasisSpec B;
    B.nkx = B.nky = DK + 1; B.nbx = B.nby = DB + 1; B.mode = cpr ? 1 : 0;
    auto powers = [](int n, int N, std::vector<double>& t) {
        t.resize((size_t)n * N);
        for (int e = 0; e < n; ++e) for (int x = 0; x < N; ++x) t[(size_t)e * N + x] = ipow_host((double(x) + 1.0) / N, e);
    };
    powers(B.nkx, N0, B.kbx); powers(B.nky, N1, B.kby); powers(B.nbx, N0, B.tbx); powers(B.nby, N1, B.tby);
    for (int i = 0; i <= DK; ++i) for (int j = 0; j <= DK - i; ++j) { B.kpair.push_back(i); B.kpair.push_back(j); }
    for (int a = 0; a <= DB; ++a) for (int b = 0; b <= DB - a; ++b) { B.bpair.push_back(a); B.bpair.push_back(b); }
    B.Fij = (int)B.kpair.size() / 2; B.Fpq = (int)B.bpair.size() / 2;
    return plan_create_impl(out, N0, N1, KerHW, B, DK, DB, device);
}

// General separable spatial bases (B-spline SFFT, sfft/BSplineSFFT.py:2536-2607): the caller tabulates the 1-D basis
// functions per axis (host pointers) and lists which (x-factor, y-factor) pair makes each kernel / background term.
static int create_from_tables(sfft_plan** out, int N0, int N1, int KerHW,
                              int nkx, int nky, const double* kbx, const double* kby, int Fij, const int* ker_pairs,
                              int nsx, int nsy, const double* sbx, const double* sby, int ScaFij, const int* sca_pairs,
                              int nbx, int nby, const double* tbx, const double* tby, int Fpq, const int* bkg_pairs,
                              int scaling_mode, int device);

extern "C" int sfft_plan_create_basis(sfft_plan** out, int N0, int N1, int KerHW,
                                      int nkx, int nky, const double* kbx, const double* kby, int Fij, const int* ker_pairs,
                                      int nbx, int nby, const double* tbx, const double* tby, int Fpq, const int* bkg_pairs,
                                      int scaling_mode, int device)
{
    if (scaling_mode < 0 || scaling_mode > 2) return set_err(SFFT_ERR_INVALID_ARG, "scaling_mode must be 0, 1 or 2");
    return create_from_tables(out, N0, N1, KerHW, nkx, nky, kbx, kby, Fij, ker_pairs, 0, 0, nullptr, nullptr, 0, nullptr,
                              nbx, nby, tbx, tby, Fpq, bkg_pairs, scaling_mode, device);
}

// Separately varying scaling (BSplineSFFT.py SCALING_MODE 'SEPARATE-VARYING', :173-201, 334-397): as
// sfft_plan_create_basis, plus the spatial basis of the flux scaling -- ScaFij <= Fij terms
// sbx[sca_pairs[2 s]][row] * sby[sca_pairs[2 s + 1]][col].  Unknown (ij, ab = centre) is the coefficient of scaling term
// ij for ij < ScaFij and leaves the system for ij >= ScaFij (the reference's zero place-holders).
extern "C" int sfft_plan_create_varscale(sfft_plan** out, int N0, int N1, int KerHW,
                                         int nkx, int nky, const double* kbx, const double* kby, int Fij, const int* ker_pairs,
                                         int nsx, int nsy, const double* sbx, const double* sby, int ScaFij, const int* sca_pairs,
                                         int nbx, int nby, const double* tbx, const double* tby, int Fpq, const int* bkg_pairs,
                                         int device)
{
    if (!sbx || !sby || !sca_pairs) return set_err(SFFT_ERR_INVALID_ARG, "NULL scaling basis table");
    if (nsx < 1 || nsy < 1 || ScaFij < 1) return set_err(SFFT_ERR_INVALID_ARG, "empty scaling basis");
    if (ScaFij > Fij) return set_err(SFFT_ERR_INVALID_ARG, "ScaFij must not exceed Fij");
    for (int k = 0; k < ScaFij; ++k) if (sca_pairs[2 * k] < 0 || sca_pairs[2 * k] >= nsx || sca_pairs[2 * k + 1] < 0 || sca_pairs[2 * k + 1] >= nsy)
        return set_err(SFFT_ERR_INVALID_ARG, "scaling term refers to a basis factor that does not exist");
    return create_from_tables(out, N0, N1, KerHW, nkx, nky, kbx, kby, Fij, ker_pairs, nsx, nsy, sbx, sby, ScaFij, sca_pairs,
                              nbx, nby, tbx, tby, Fpq, bkg_pairs, 3, device);
}

static int create_from_tables(sfft_plan** out, int N0, int N1, int KerHW,
                              int nkx, int nky, const double* kbx, const double* kby, int Fij, const int* ker_pairs,
                              int nsx, int nsy, const double* sbx, const double* sby, int ScaFij, const int* sca_pairs,
                              int nbx, int nby, const double* tbx, const double* tby, int Fpq, const int* bkg_pairs,
                              int scaling_mode, int device)
{
    if (!kbx || !kby || !tbx || !tby || !ker_pairs || !bkg_pairs) return set_err(SFFT_ERR_INVALID_ARG, "NULL basis table");
    if (N0 < 8 || N1 < 8) return set_err(SFFT_ERR_INVALID_ARG, "Input Image has dramatically small size!");
    if (nkx < 1 || nky < 1 || nbx < 1 || nby < 1 || Fij < 1 || Fpq < 1) return set_err(SFFT_ERR_INVALID_ARG, "empty basis");
    BasisSpec B;
    B.nkx = nkx; B.nky = nky; B.nbx = nbx; B.nby = nby; B.Fij = Fij; B.Fpq = Fpq; B.mode = scaling_mode;
    B.kbx.assign(kbx, kbx + (size_t)nkx * N0); B.kby.assign(kby, kby + (size_t)nky * N1);
    B.tbx.assign(tbx, tbx + (size_t)nbx * N0); B.tby.assign(tby, tby + (size_t)nby * N1);
    B.kpair.assign(ker_pairs, ker_pairs + 2 * (size_t)Fij); B.bpair.assign(bkg_pairs, bkg_pairs + 2 * (size_t)Fpq);
    if (scaling_mode == 3) {
        B.nsx = nsx; B.nsy = nsy; B.ScaFij = ScaFij;
        B.sbx.assign(sbx, sbx + (size_t)nsx * N0); B.sby.assign(sby, sby + (size_t)nsy * N1);
        B.spair.assign(sca_pairs, sca_pairs + 2 * (size_t)ScaFij);
    }
    for (int k = 0; k < Fij; ++k) if (B.kpair[2 * k] < 0 || B.kpair[2 * k] >= nkx || B.kpair[2 * k + 1] < 0 || B.kpair[2 * k + 1] >= nky)
        return set_err(SFFT_ERR_INVALID_ARG, "kernel term refers to a basis factor that does not exist");
    for (int k = 0; k < Fpq; ++k) if (B.bpair[2 * k] < 0 || B.bpair[2 * k] >= nbx || B.bpair[2 * k + 1] < 0 || B.bpair[2 * k + 1] >= nby)
        return set_err(SFFT_ERR_INVALID_ARG, "background term refers to a basis factor that does not exist");
    return plan_create_impl(out, N0, N1, KerHW, B, -1, -1, device);
}

// Kernel regularisation (BSplineSFFT.py REGULARIZE_KERNEL, :2007-2168, 3570-3700):
//     LHMAT[(k, c), (k8, c8)] += lambda * SCALE^2 * S[k][k8] * ireg[c][c8]
// for every later solve on this plan.  ireg [Fab][Fab] is the reference's iREGMAT (integers, passed as doubles), sst
// [Fij][Fij] its SSTMAT; plans with separately varying scaling also take CSSTMAT (kernel x scaling) and DSSTMAT
// (scaling x scaling), used when one / both of c, c8 are the kernel centre (:2121-2166).  HOST pointers, copied.
// lambda == 0 (or ireg == NULL) switches regularisation off.
extern "C" int sfft_plan_set_regularization(sfft_plan* p, double lambda, const double* ireg, const double* sst,
                                            const double* csst, const double* dsst)
{
    if (!p) return set_err(SFFT_ERR_INVALID_ARG, "NULL plan");
    ON_DEVICE(p->dev);
    if (lambda == 0.0 || !ireg) { p->fa.reg_coef = 0.0; return SFFT_OK; }
    if (!sst) return set_err(SFFT_ERR_INVALID_ARG, "NULL SSTMAT");
    if (p->mode == 3 && (!csst || !dsst)) return set_err(SFFT_ERR_INVALID_ARG, "separately varying scaling needs CSSTMAT and DSSTMAT");
    const size_t nab = (size_t)p->Fab * p->Fab, nij = (size_t)p->Fij * p->Fij;
    int rc;
    if (!p->d_ireg) {
        if ((rc = dev_alloc(p, &p->d_ireg, nab))) return rc;
        if ((rc = dev_alloc(p, &p->d_sst, nij))) return rc;
        if ((rc = dev_alloc(p, &p->d_csst, nij))) return rc;
        if ((rc = dev_alloc(p, &p->d_dsst, nij))) return rc;
    }
    HIPCHK(hipMemcpy(p->d_ireg, ireg, nab * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->d_sst, sst, nij * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->d_csst, p->mode == 3 ? csst : sst, nij * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->d_dsst, p->mode == 3 ? dsst : sst, nij * sizeof(double), hipMemcpyHostToDevice));
    p->fa.ireg = p->d_ireg; p->fa.sst = p->d_sst; p->fa.csst = p->d_csst; p->fa.dsst = p->d_dsst;
    p->fa.reg_coef = lambda * p->scale * p->scale;
    return SFFT_OK;
}

static void free_axis(AxisHost& a)
{
    if (a.subA) { free_axis(*a.subA); delete a.subA; a.subA = nullptr; }
    if (a.subB) { free_axis(*a.subB); delete a.subB; a.subB = nullptr; }
    dev_free(a.tw);
    if (!a.root_is_tw) dev_free(a.root);
    dev_free(a.chirp);
    dev_free(a.bf);
    dev_free(a.rin); dev_free(a.rout);
    a.tw = a.root = a.chirp = a.bf = nullptr;
    a.rin = a.rout = nullptr;
}

extern "C" int sfft_plan_destroy(sfft_plan* p)
{
    if (!p) return SFFT_OK;
    DeviceGuard device_guard_(p->dev);
    free_axis(p->ax0); free_axis(p->ax1);
    void* ptrs[] = {p->d_idx, p->d_phi, p->d_Xp, p->d_Yq, p->d_passes, p->d_jobs, p->d_spec, p->d_gp, p->d_patches, p->d_A, p->d_sol,
                    p->d_rtab, p->d_rowmom, p->d_delta, p->d_status, p->d_dbuf, p->d_xv, p->d_partial, p->d_counter, p->d_w0tab, p->d_rd, p->d_spec2, p->d_big1, p->d_big2, p->d_colscr, p->d_bb1, p->d_bb2, p->d_kbx, p->d_kby, p->d_tbx, p->d_tby, p->d_zero, p->d_zsol,
                    p->d_sbx, p->d_sby, p->d_ireg, p->d_sst, p->d_csst, p->d_dsst, p->d_ones, p->d_stage, p->d_stage_a, p->d_ctabm, p->d_winv, p->d_luperm, p->d_luxchg, p->d_bflags, p->d_epoch, p->d_tflags, p->d_trace, p->d_w16, p->d_groups, p->d_g1trace, p->d_sprods, p->d_slines, p->d_scols, p->d_strip, p->d_sitems, p->d_ibase, p->d_cyp, p->d_rowmomI, p->d_gamR, p->d_pq};
    if (p->chol_exec) hipGraphExecDestroy(p->chol_exec);
    if (p->lu_exec) hipGraphExecDestroy(p->lu_exec);

    if (p->h_status) hipHostFree(p->h_status);
    for (void* q : ptrs) dev_free(q);
    for (int s = 0; s < SFFT_ST_COUNT; ++s) { if (p->ev[s][0]) hipEventDestroy(p->ev[s][0]); if (p->ev[s][1]) hipEventDestroy(p->ev[s][1]); }
    if (p->s2) { hipStreamSynchronize(p->s2); hipStreamDestroy(p->s2); }
    if (p->ev_in) hipEventDestroy(p->ev_in);
    if (p->ev_pre) hipEventDestroy(p->ev_pre);
    if (p->ev_mom) hipEventDestroy(p->ev_mom);
    if (p->ev_gam) hipEventDestroy(p->ev_gam);
    delete p;
    return SFFT_OK;
}

static bool g1_decimated(const sfft_plan* p);
extern "C" int sfft_plan_query(const sfft_plan* p, int field, long long* v)
{
    if (!p || !v) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    switch (field) {
        case SFFT_Q_N0: *v = p->N0; break;
        case SFFT_Q_N1: *v = p->N1; break;
        case SFFT_Q_W0: case SFFT_Q_W1: *v = p->w; break;
        case SFFT_Q_DK: *v = p->DK; break;
        case SFFT_Q_DB: *v = p->DB; break;
        case SFFT_Q_CONSTPHOTRATIO: *v = p->cpr; break;
        case SFFT_Q_L0: case SFFT_Q_L1: *v = p->L; break;
        case SFFT_Q_FAB: *v = p->Fab; break;
        case SFFT_Q_FIJ: *v = p->Fij; break;
        case SFFT_Q_FPQ: *v = p->Fpq; break;
        case SFFT_Q_NEQ: *v = p->NEQ; break;
        case SFFT_Q_FIJAB: *v = p->Fijab; break;
        case SFFT_Q_NEQ_FSFREE: *v = (p->mode == 3) ? p->NEQfs : p->NEQ - (p->Fij - 1); break;
        case SFFT_Q_FOMG: *v = p->Fij * p->Fij; break;
        case SFFT_Q_FGAM: case SFFT_Q_FPSI: *v = p->Fij * p->Fpq; break;
        case SFFT_Q_FTHE: *v = p->Fij; break;
        case SFFT_Q_FPHI: *v = p->Fpq * p->Fpq; break;
        case SFFT_Q_FDEL: *v = p->Fpq; break;
        case SFFT_Q_WORKSPACE_BYTES: *v = (long long)p->ws_bytes; break;
        case SFFT_Q_LAST_SOLVER: *v = p->last_solver; break;
        case SFFT_Q_CHOL_STATUS: *v = p->chol_status; break;
        case SFFT_Q_SOLVES: *v = p->n_solves; break;
        case SFFT_Q_LU_FALLBACKS: *v = p->n_lu_fallback; break;
        case SFFT_Q_CHOL_STALLS: *v = p->n_chol_stall; break;
        case SFFT_Q_NUM_GREEK_PAIRS: *v = (long long)(p->n_omg + p->n_dense_w); break;
        case SFFT_Q_SCAFIJ: *v = p->nsca; break;
        case SFFT_Q_SOLVE_GRAPH: *v = p->chol_exec ? 1 : 0; break;
        case SFFT_Q_THETA_FUSED: *v = ((p->theta_in_groups || p->theta_slots) && p->g1_mfma >= 3) ? 1 : 0; break;
        case SFFT_Q_OMG_OFFDIAG: *v = p->n_omg_off; break;
        case SFFT_Q_G1_CHUNKS: *v = p->S; break;
        case SFFT_Q_G1_DECIMATED: *v = (g1_decimated(p) && 2 * p->w >= 9 && 2 * p->w <= 32) ? 1 : 0; break;
        case SFFT_Q_OMG_DIAG: *v = p->n_omg_diag; break;
        case SFFT_Q_G1_MFMA: *v = (p->g1_mfma && 2 * p->w >= 9 && (2 * p->w <= 16 || (2 * p->w <= 32 && p->g1_mfma >= 3 && p->d_groups))) ? 1 : 0; break;
        case SFFT_Q_CHOL_DATAFLOW: *v = (p->dataflow && p->NEQfs < p->chol_outer_min) ? 1 : 0; break;
        case SFFT_Q_SOLVER_N: *v = p->NEQfs; break;
        case SFFT_Q_OMG_SPARSE: *v = (long long)p->sprods.size(); break;
        default: return set_err(SFFT_ERR_INVALID_ARG, "unknown query field");
    }
    return SFFT_OK;
}

extern "C" int sfft_set_timing(sfft_plan* p, int enable)
{
    if (!p) return set_err(SFFT_ERR_INVALID_ARG, "NULL plan");
    p->timing = enable;
    if (enable) for (int s = 0; s < SFFT_ST_COUNT; ++s) p->stage_kernels[s].clear();
    return SFFT_OK;
}
extern "C" int sfft_set_force_lu(sfft_plan* p, int enable) { if (!p) return set_err(SFFT_ERR_INVALID_ARG, "NULL plan"); p->force_lu = enable; return SFFT_OK; }

extern "C" int sfft_stage_ms(sfft_plan* p, int stage, float* ms)
{
    if (!p || !ms || stage < 0 || stage >= SFFT_ST_COUNT) return set_err(SFFT_ERR_INVALID_ARG, "bad argument");
    *ms = 0.0f;
    if (!p->ev_valid[stage]) return SFFT_OK;
    HIPCHK(hipEventSynchronize(p->ev[stage][1]));
    HIPCHK(hipEventElapsedTime(ms, p->ev[stage][0], p->ev[stage][1]));
    return SFFT_OK;
}

extern "C" int sfft_stage_kernels(sfft_plan* p, int stage, char* buf, int cap)
{
    if (!p || !buf || cap < 1 || stage < 0 || stage >= SFFT_ST_COUNT) return set_err(SFFT_ERR_INVALID_ARG, "bad argument");
    const std::string& L = p->stage_kernels[stage];
    const size_t n = std::min(L.size(), (size_t)cap - 1);
    memcpy(buf, L.data(), n);
    buf[n] = 0;
    return SFFT_OK;
}

struct StageTimer {
    sfft_plan* p; int st; hipStream_t s; KLogScope k;
    StageTimer(sfft_plan* p_, int st_, hipStream_t s_) : p(p_), st(st_), s(s_), k(p_->timing ? &p_->stage_kernels[st_] : nullptr) { if (p->timing) { hipEventRecord(p->ev[st][0], s); } }
    ~StageTimer() { if (p->timing) { hipEventRecord(p->ev[st][1], s); p->ev_valid[st] = true; } }
};

#define LAUNCH_CHECK() HIPCHK(hipGetLastError())

// forward transforms of `nplanes` polynomial-weighted planes into spec planes [0, nplanes)
static bool fast_axis(const AxisHost& a) { return !a.big && !a.blue && a.M == 4096; }

// one pass of batched strided sub-transforms
static void launch_pass(sfft_plan* p, const cplx* in, cplx* out, PassDesc d, const AxisHost& sub, const cplx* rootN, hipStream_t s)
{
    if (d.mode == 2 && d.len == 16 && sub.M == 16 && !sub.blue && !p->no_dft16_regs) {        // 16-point first pass of a column transform: registers only
        SFFT_LAUNCH(strided_dft16_cols, dim3((d.nlines + 63) / 64, (d.J + 3) / 4), dim3(256), 0, s, in, out, d, rootN);
        return;
    }
    int TC, MS;
    // Bluestein sub-transforms run two passes over the tile with barriers throughout: two workgroups per CU (half the LDS each)
    // hide more than a wider tile gains (9232-point columns: 22.4 -> 20.1 ms for 11 planes)
    pick_col_tile(sub, &TC, &MS, sub.blue ? std::min((size_t)4800, (size_t)LDS_COL_ELEMS) : 0);
    if (sub.rader) {                                // RADER_TC sequences of RADER_XS + 1 elements (fft_fourstep.hpp)
        TC = RADER_TC; MS = (RADER_XS + 1 + 15) / 16 * 16 + 16 / TC;
        if (d.mode == 2 && !d.twiddle && !d.w && sub.N == 577 && !p->no_rader_r24) {       // ... in registers, 24 x 24
            SFFT_LAUNCH(strided_rader577_r24, dim3((d.nlines + RDR_SEQ - 1) / RDR_SEQ, d.J), dim3(RDR_NT), 0, s, in, out, d, axis_dev(sub));
            return;
        }
        if (d.mode == 2 && !d.twiddle) {            // (the second pass of a column transform: its own kernel)
            SFFT_LAUNCH(strided_rader577, dim3((d.nlines + TC - 1) / TC, d.J), dim3(RADER_NT), (size_t)TC * MS * sizeof(cplx), s, in, out, d,
                               axis_dev(sub), MS);
            return;
        }
        // strided_dft / lds_dft have no Rader branch: a Rader sub-axis anywhere but the second pass of a column transform would be
        // transformed as a 576-point sequence.  No caller does this today (only ax0's factor B is built with Rader); refuse rather than compute
        set_err(SFFT_ERR_UNSUPPORTED_SIZE, "internal: Rader sub-transform requested outside the second pass of a column transform");
        p->launch_error = SFFT_ERR_UNSUPPORTED_SIZE;
        return;
    }
    const int nt = fft_threads(TC * (sub.rader ? sub.N : sub.M));
    const int ngroups = (d.mode == 2) ? (d.nlines + TC - 1) / TC : (d.J + TC - 1) / TC;
    const int gy = (d.mode == 2) ? d.J : d.nlines;
    SFFT_LAUNCH(strided_dft, dim3(ngroups, gy), dim3(nt), (size_t)TC * MS * sizeof(cplx), s, in, out, d, axis_dev(sub), rootN, TC, ilog2(TC), MS);
}

// four-step transform of `nlines` lines of length ax.N; element stride st, line stride lst (complex elements).
// Result lands in `data` again (scr is a same-shaped scratch).  inverse: e^{+i} (conjugation on the way in and out).
// src / wrow (optional): read the input from another plane of the same shape and multiply input element (row) by wrow[row] -- the
// weighted forward column pass of the staged transforms (columns only: the element index is the row).
static void four_step_passes(const AxisHost& ax, long long st, long long lst, int nlines, bool lines_fastest, int inverse, const double* wrow,
                             PassDesc* p1, PassDesc* p2)
{
    PassDesc d1; memset(&d1, 0, sizeof(d1));
    d1.w = wrow; d1.w_js = 1; d1.w_es = ax.B;
    d1.len = ax.A; d1.J = ax.B; d1.nlines = nlines; d1.mode = lines_fastest ? 2 : 1;
    d1.js_in = st; d1.es_in = (long long)ax.B * st; d1.lst_in = lst;
    d1.js_out = d1.js_in; d1.es_out = d1.es_in; d1.lst_out = lst;
    d1.twiddle = 1; d1.N = ax.N; d1.conj_in = inverse; d1.conj_out = 0; d1.scale = 1.0;
    PassDesc d2; memset(&d2, 0, sizeof(d2));
    d2.len = ax.B; d2.J = ax.A; d2.nlines = nlines; d2.mode = lines_fastest ? 2 : 0;
    d2.js_in = (long long)ax.B * st; d2.es_in = st; d2.lst_in = lst;
    d2.js_out = st; d2.es_out = (long long)ax.A * st; d2.lst_out = lst;
    d2.twiddle = 0; d2.N = ax.N; d2.conj_in = 0; d2.conj_out = inverse; d2.scale = 1.0;
    *p1 = d1; *p2 = d2;
}

static void big_axis_transform(sfft_plan* p, const AxisHost& ax, cplx* data, cplx* scr, long long st, long long lst, int nlines,
                               bool lines_fastest, int inverse, hipStream_t s, const cplx* src = nullptr, const double* wrow = nullptr)
{
    if (ax.bigblue) {       // Bluestein through the BM-point four-step transform, a batch of lines at a time through the compact work arrays
        const int M = ax.BM, per = (int)std::max<size_t>(1, std::min<size_t>(p->bb_elems / (size_t)M, 16384));
        for (int l0 = 0; l0 < nlines; l0 += per) {
            const int nl = std::min(per, nlines - l0);
            BlueDesc bd; bd.N = ax.N; bd.M = M; bd.nlines = nl; bd.line0 = l0; bd.conj = inverse; bd.transposed = (lst < st) ? 1 : 0;
            bd.st = st; bd.lst = lst; bd.w = wrow;
            SFFT_LAUNCH(bigblue_move, dim3((M + 15) / 16, (nl + 15) / 16), dim3(256), 0, s, src ? src : (const cplx*)data, p->d_bb1, bd, (const cplx*)ax.chirp, 0);
            big_axis_transform(p, *ax.subA, p->d_bb1, p->d_bb2, 1, M, nl, false, 0, s);
            SFFT_LAUNCH(bigblue_filter, dim3((M + 255) / 256, nl), dim3(256), 0, s, p->d_bb1, (const cplx*)ax.bf, M);
            big_axis_transform(p, *ax.subA, p->d_bb1, p->d_bb2, 1, M, nl, false, 1, s);
            bd.w = nullptr;
            SFFT_LAUNCH(bigblue_move, dim3((ax.N + 15) / 16, (nl + 15) / 16), dim3(256), 0, s, (const cplx*)p->d_bb1, data, bd, (const cplx*)ax.chirp, 1);
        }
        return;
    }
    PassDesc d1, d2;
    four_step_passes(ax, st, lst, nlines, lines_fastest, inverse, wrow, &d1, &d2);
    launch_pass(p, src ? src : data, scr, d1, *ax.subA, ax.root, s);
    launch_pass(p, scr, data, d2, *ax.subB, ax.root, s);
}

// The staged forward column pass on a four-step axis whose first factor is 16 (9232 = 16 x 577): the outputs of one stage plane, at most
// DFT16_MAX_OUT at a time, share ONE read of the stage plane in the first pass (strided_dft16_cols_multi); every output then takes its own
// second pass from its scratch plane.  false: this axis / build does not take that route (the caller transforms output by output).
static bool staged_cols_shared_first_pass(sfft_plan* p, const AxisHost& ax, const cplx* stage, int nout, cplx* const* dsts, const double* const* wx,
                                          long long st, long long lst, int nlines, hipStream_t s)
{
    if (ax.bigblue || !ax.subA || ax.A != 16 || ax.subA->M != 16 || ax.subA->blue || p->no_dft16_regs || p->colscr_planes < 2) return false;
    PassDesc d1, d2;
    four_step_passes(ax, st, lst, nlines, true, 0, p->d_ones, &d1, &d2);        // (d1.w only says "weighted": the kernel takes the weights per output)
    const size_t plane_sz = (size_t)p->N0 * p->Nhp;
    for (int o0 = 0; o0 < nout; o0 += std::min(DFT16_MAX_OUT, p->colscr_planes)) {
        const int no = std::min(std::min(DFT16_MAX_OUT, p->colscr_planes), nout - o0);
        Dft16Outs oo; memset(&oo, 0, sizeof(oo));
        oo.nout = no;
        for (int q = 0; q < no; ++q) { oo.w[q] = wx[o0 + q]; oo.out[q] = p->d_colscr + (size_t)q * plane_sz; }
        SFFT_LAUNCH(strided_dft16_cols_multi, dim3((d1.nlines + 63) / 64, (d1.J + 3) / 4), dim3(256), 0, s, stage, oo, d1, (const cplx*)ax.root);
        for (int q = 0; q < no; ++q) launch_pass(p, p->d_colscr + (size_t)q * plane_sz, dsts[o0 + q], d2, *ax.subB, ax.root, s);
    }
    return true;
}

static void launch_cols(sfft_plan* p, cplx* data, int nplanes, int inverse, hipStream_t s)
{
    if (p->ax0.big) {
        for (int k = 0; k < nplanes; ++k)
            big_axis_transform(p, p->ax0, data + (size_t)k * p->N0 * p->Nhp, p->d_colscr, p->Nhp, 1, p->Nh, true, inverse, s);
        return;
    }
    if (fast_axis(p->ax0) && !p->no_fast_fft) {
        const int npairs = (p->Nh + 1) / 2;
        const int per = (npairs + 7) / 8;
        SFFT_LAUNCH(cols_c2c_4096, dim3(8 * per, nplanes), dim3(512), (2 * F4K_LDS + 8) * sizeof(cplx), s, data, p->Nh, p->Nhp,
                           p->lay, p->ax0.tw, inverse, 1.0, per);
    } else {
        const int G8 = 8 * (p->TC >= 8 ? 1 : 8 / p->TC);
        dim3 g2(((p->Nh + p->TC - 1) / p->TC + G8 - 1) / G8 * G8, nplanes);
        SFFT_LAUNCH(cols_c2c, g2, dim3(p->nt_cols), p->lds_cols, s, data, p->N0, p->Nh, p->Nhp, p->TC, ilog2(p->TC), p->MS,
                           p->lay, axis_dev(p->ax0), inverse, 1.0);
    }
}

// st_rows / st_cols: stage ids that time the row pass / the column pass of this call (-1: not timed)
// rows_only: stop after the row pass (the planes are then "stage" planes: row-DFT only)
static int forward_planes(sfft_plan* p, const RowsArgs& ra, int nplanes, cplx* dst, hipStream_t s, int st_rows = -1, int st_cols = -1,
                          bool rows_only = false)
{
    std::string* const klog_outer = tl_klog;      // (the stage events below also redirect the kernel-name log)
    dim3 g1((p->N0 + 1) / 2, nplanes);
    if (p->timing && st_rows >= 0) { hipEventRecord(p->ev[st_rows][0], s); tl_klog = &p->stage_kernels[st_rows]; }
    if (p->ax1.big) {
        const int npr = (p->N0 + 1) / 2;
        for (int k = 0; k < nplanes; ++k) {
            SFFT_LAUNCH(pack_rows, dim3((p->N1 + 255) / 256, npr), dim3(256), 0, s, ra.src[k], ra.wx[k], ra.wy[k], p->d_big1, p->N0, p->N1);
            big_axis_transform(p, p->ax1, p->d_big1, p->d_big2, 1, p->N1, npr, false, 0, s);
            SFFT_LAUNCH(untangle_rows, dim3((p->Nh + 255) / 256, npr), dim3(256), 0, s, p->d_big1, dst + (size_t)k * p->N0 * p->Nhp,
                               p->N0, p->N1, p->Nh, p->Nhp, p->scale);
        }
    } else if (fast_axis(p->ax1) && !p->no_fast_fft) {
        RowsArgs rw = ra;                       // unweighted planes (J, plain FFTs) get the table of ones: branch-free kernel
        for (int k = 0; k < nplanes; ++k) { if (!rw.wx[k]) rw.wx[k] = p->d_ones; if (!rw.wy[k]) rw.wy[k] = p->d_ones; }
        RowGroups grp; grp.ngroups = 0;
        for (int u = 0; u < SFFT_MAX_PLANES; ++u) { grp.mom_out[u] = nullptr; grp.mom_nq[u] = 0; }
        for (int k = 0; k < nplanes; ++k) {
            if (k > 0 && ra.src[k] == ra.src[k - 1]) ++grp.count[grp.ngroups - 1];
            else { grp.first[grp.ngroups] = k; grp.count[grp.ngroups] = 1; ++grp.ngroups; }
        }
        const int rp_per = ((p->N0 + 1) / 2 + 7) / 8;
        SFFT_LAUNCH(rows_r2c_4096, dim3(8 * rp_per, grp.ngroups), dim3(256), F4K_LDS * sizeof(cplx), s, rw, grp, dst,
                           p->N0, p->Nhp, p->lay, p->ax1.tw, p->scale, rp_per, 0);
    }
    else if (p->rows_r24) {
        RowsArgs rw = ra;
        for (int k = 0; k < nplanes; ++k) { if (!rw.wx[k]) rw.wx[k] = p->d_ones; if (!rw.wy[k]) rw.wy[k] = p->d_ones; }
        const int rp_per = ((p->N0 + 1) / 2 + 7) / 8;
        if (p->rows_r24 == 16)
            SFFT_LAUNCH(rows_r2c_r24<16>, dim3(8 * rp_per * nplanes), dim3(R24<16>::NT), R24<16>::LDS * sizeof(double), s, rw, nplanes, dst,
                               p->N0, p->Nhp, p->lay, p->ax1.tw, p->scale, rp_per);
        else
            SFFT_LAUNCH(rows_r2c_r24<24>, dim3(8 * rp_per * nplanes), dim3(R24<24>::NT), R24<24>::LDS * sizeof(double), s, rw, nplanes, dst,
                               p->N0, p->Nhp, p->lay, p->ax1.tw, p->scale, rp_per);
    }
    else
        SFFT_LAUNCH(rows_r2c, g1, dim3(p->nt_rows), p->lds_rows, s, ra, dst, p->N0, p->N1, p->Nh, p->Nhp,
                           p->lay, axis_dev(p->ax1), p->scale);
    LAUNCH_CHECK();
    if (p->timing && st_rows >= 0) { hipEventRecord(p->ev[st_rows][1], s); p->ev_valid[st_rows] = true; tl_klog = klog_outer; }
    if (rows_only) return SFFT_OK;
    if (p->timing && st_cols >= 0) { hipEventRecord(p->ev[st_cols][0], s); tl_klog = &p->stage_kernels[st_cols]; }
    launch_cols(p, dst, nplanes, 0, s);
    if (p->timing && st_cols >= 0) { hipEventRecord(p->ev[st_cols][1], s); p->ev_valid[st_cols] = true; tl_klog = klog_outer; }
    LAUNCH_CHECK();
    return SFFT_OK;
}

// forward spectra of the Fij kernel-basis planes of image d_I (and, when d_J is given, of d_J itself as plane Fij)
// and, with_sca, of the scaling planes (planes Fij + 1 ...)
// Fast path (both axes 4096): the row pass runs once per distinct (image, column factor) into the stage buffer, the weighted
// column pass transforms stage plane x row factor for every output plane (see cols_fwd_weighted_4096).
static int forward_basis_planes_staged(sfft_plan* p, const double* d_I, const double* d_J, cplx* dst, hipStream_t s, bool with_sca,
                                       int st_rows, int st_cols)
{
    std::string* const klog_outer = tl_klog;      // (the stage events below also redirect the kernel-name log)
    struct Out { int plane; const double* wx; };
    struct Stage { const double* src; const double* wy; std::vector<Out> outs; };
    std::vector<Stage> stages;
    auto add = [&](const double* src, const double* wy, int plane, const double* wx) {
        for (auto& st : stages) if (st.src == src && st.wy == wy) { st.outs.push_back({plane, wx}); return; }
        stages.push_back({src, wy, {{plane, wx}}});
    };
    for (int k = 0; k < p->Fij; ++k)
        add(d_I, p->d_kby + (size_t)p->kpair[2 * k + 1] * p->N1, k, p->d_kbx + (size_t)p->kpair[2 * k] * p->N0);
    if (d_J) add(d_J, p->d_ones, p->Fij, p->d_ones);
    if (with_sca && d_J)
        for (int sI = 0; sI < p->nsca; ++sI)
            add(d_I, p->d_sby + (size_t)p->spair[2 * sI + 1] * p->N1, p->Fij + 1 + sI, p->d_sbx + (size_t)p->spair[2 * sI] * p->N0);
    // stages that read the same image must be consecutive for the row kernel's one-read-per-image grouping
    std::stable_sort(stages.begin(), stages.end(), [&](const Stage& a, const Stage& b) { return (a.src == d_I) > (b.src == d_I); });
    const int nst = (int)stages.size();
    if (nst > p->n_stage_alloc) {
        if (p->d_stage) { HIPCHK(hipStreamSynchronize(s)); dev_free(p->d_stage); p->d_stage = nullptr; }
        int rc = dev_alloc(p, &p->d_stage, (size_t)nst * p->N0 * p->Nhp);
        if (rc) return rc;
        p->n_stage_alloc = nst;
    }
    const size_t plane_sz = (size_t)p->N0 * p->Nhp;
    const bool fast = !p->no_fast_fft && fast_axis(p->ax0) && fast_axis(p->ax1);
    if (!fast) {
        // generic shapes: the ordinary row pass (any axis-1 length) per stage plane, then the weighted on-chip column pass
        for (int k0 = 0; k0 < nst; k0 += SFFT_MAX_PLANES) {
            const int n = std::min(SFFT_MAX_PLANES, nst - k0);
            RowsArgs ra;
            for (int u = 0; u < SFFT_MAX_PLANES; ++u) { ra.src[u] = nullptr; ra.wx[u] = nullptr; ra.wy[u] = nullptr; }
            for (int u = 0; u < n; ++u) { ra.src[u] = stages[k0 + u].src; ra.wy[u] = stages[k0 + u].wy; }
            int rc = forward_planes(p, ra, n, p->d_stage + (size_t)k0 * plane_sz, s, st_rows, -1, true);
            if (rc) return rc;
        }
        if (p->timing && st_cols >= 0) { hipEventRecord(p->ev[st_cols][0], s); tl_klog = &p->stage_kernels[st_cols]; }
        if (p->ax0.big) {
            // four-step column axis: every output plane is the transform of its stage plane times the row factor, applied as the first
            // pass reads the stage plane (same traffic as transforming a finished plane; the row pass ran once per column factor)
            for (int k = 0; k < nst; ++k) {
                std::vector<cplx*> dsts; std::vector<const double*> wxs;
                for (const Out& o : stages[k].outs) { dsts.push_back(dst + (size_t)o.plane * plane_sz); wxs.push_back(o.wx == p->d_ones ? nullptr : o.wx); }
                if (staged_cols_shared_first_pass(p, p->ax0, p->d_stage + (size_t)k * plane_sz, (int)dsts.size(), dsts.data(), wxs.data(), p->Nhp, 1, p->Nh, s)) continue;
                for (const Out& o : stages[k].outs)
                    big_axis_transform(p, p->ax0, dst + (size_t)o.plane * plane_sz, p->d_colscr, p->Nhp, 1, p->Nh, true, 0, s,
                                       p->d_stage + (size_t)k * plane_sz, o.wx == p->d_ones ? nullptr : o.wx);
            }
            LAUNCH_CHECK();
            if (p->timing && st_cols >= 0) { hipEventRecord(p->ev[st_cols][1], s); p->ev_valid[st_cols] = true; tl_klog = klog_outer; }
            return SFFT_OK;
        }
        // the solve pass's spectra of a plan with the paired 6144-point column kernel go out in 2-column panels (lay_spec; the Greek launches read them)
        const SpecLayout lay_cols_out = (p->spec2 && d_J != nullptr && dst == p->d_spec) ? p->lay_spec : p->lay;
        const int G = p->TC >= 8 ? 1 : 8 / p->TC;
        const int ntiles = (p->Nh + p->TC - 1) / p->TC;
        const int ntg = (ntiles + 8 * G - 1) / (8 * G);           // tile groups per XCD
        int k = 0;
        while (k < nst) {
            ColOuts g; memset(&g, 0, sizeof(g));
            while (k < nst && g.nout + (int)stages[k].outs.size() <= COLG_MAX_OUT) {
                for (const Out& o : stages[k].outs) {
                    g.stage_plane[g.nout] = k; g.out_plane[g.nout] = o.plane; g.wx[g.nout] = o.wx;
                    g.lo[g.nout] = 0; g.hi[g.nout] = p->N0;
                    const ptrdiff_t off = o.wx - p->d_kbx;
                    if (off >= 0 && off < (ptrdiff_t)p->kbx_lo.size() * p->N0) {
                        g.lo[g.nout] = p->kbx_lo[off / p->N0]; g.hi[g.nout] = p->kbx_hi[off / p->N0];
                    }
                    ++g.nout;
                }
                ++k;
            }
            if (g.nout == 0) return set_err(SFFT_ERR_INVALID_ARG, "too many planes share one column factor for the weighted column pass");
            if (p->cols_r24) {     // register-resident 6144- / 9216-point columns, one per workgroup, 64 columns per (XCD-interleaved) column group
                const dim3 grid(64 * g.nout * ((p->Nh + 63) / 64));
                if (p->cols_r24 == 16 && p->cols_r24_pair)        // two panel neighbours per workgroup, in neighbouring lanes
                    SFFT_LAUNCH((cols_fwd_weighted_r24<16, 2>), dim3(grid.x / 2), dim3(2 * R24<16>::NT), (2 * R24<16>::LDS + 16) * sizeof(double), s,
                                       p->d_stage, dst, g, p->Nh, p->Nhp, p->lay, lay_cols_out, p->ax0.tw);
                else if (p->cols_r24 == 16)
                    SFFT_LAUNCH(cols_fwd_weighted_r24<16>, grid, dim3(R24<16>::NT), R24<16>::LDS * sizeof(double), s,
                                       p->d_stage, dst, g, p->Nh, p->Nhp, p->lay, lay_cols_out, p->ax0.tw);
                else
                    SFFT_LAUNCH(cols_fwd_weighted_r24<24>, grid, dim3(R24<24>::NT), R24<24>::LDS * sizeof(double), s,
                                       p->d_stage, dst, g, p->Nh, p->Nhp, p->lay, lay_cols_out, p->ax0.tw);
                continue;
            }
            SFFT_LAUNCH(cols_fwd_weighted, dim3(8 * G * g.nout * ntg), dim3(p->nt_cols), p->lds_cols, s, p->d_stage, dst, g, p->N0, p->Nh,
                               p->Nhp, p->TC, ilog2(p->TC), p->MS, p->lay, axis_dev(p->ax0));
        }
        LAUNCH_CHECK();
        if (p->timing && st_cols >= 0) { hipEventRecord(p->ev[st_cols][1], s); p->ev_valid[st_cols] = true; tl_klog = klog_outer; }
        return SFFT_OK;
    }
    const int rp_per = ((p->N0 + 1) / 2 + 7) / 8;
    // the solve pass (its spectra go to the Greek launches, which read d_spec in lay_spec): one column pair per workgroup, pair-major stage lines
    const bool zpath = p->colz && d_J != nullptr && dst == p->d_spec;
    if (p->timing && st_rows >= 0) { hipEventRecord(p->ev[st_rows][0], s); tl_klog = &p->stage_kernels[st_rows]; }
    for (int k0 = 0; k0 < nst; k0 += SFFT_MAX_PLANES) {
        const int n = std::min(SFFT_MAX_PLANES, nst - k0);
        RowsArgs ra;
        for (int u = 0; u < SFFT_MAX_PLANES; ++u) { ra.src[u] = nullptr; ra.wx[u] = p->d_ones; ra.wy[u] = p->d_ones; }
        RowGroups grp; grp.ngroups = 0;
        for (int u = 0; u < SFFT_MAX_PLANES; ++u) { grp.mom_out[u] = nullptr; grp.mom_nq[u] = 0; }
        for (int u = 0; u < n; ++u) {
            ra.src[u] = stages[k0 + u].src; ra.wy[u] = stages[k0 + u].wy;
            if (u > 0 && ra.src[u] == ra.src[u - 1]) ++grp.count[grp.ngroups - 1];
            else { grp.first[grp.ngroups] = u; grp.count[grp.ngroups] = 1; ++grp.ngroups; }
        }
        if (p->rowmom_fused && d_J && k0 == 0)        // solve pass: the moments of the masked pair from the rows this launch reads anyway
            for (int gI = 0; gI < grp.ngroups; ++gI) {
                const double* src = ra.src[grp.first[gI]];
                if (src == d_J) { grp.mom_out[gI] = p->d_rowmom; grp.mom_nq[gI] = p->nby; }
                else if (src == d_I) { grp.mom_out[gI] = p->d_rowmomI; grp.mom_nq[gI] = p->gam_nmu; }
            }
        SFFT_LAUNCH(rows_r2c_4096, dim3(8 * rp_per, grp.ngroups), dim3(256), F4K_LDS * sizeof(cplx), s, ra, grp,
                           p->d_stage + (size_t)k0 * plane_sz, p->N0, p->Nhp, p->lay, p->ax1.tw, p->scale, rp_per, zpath ? 1 : 0);
    }
    LAUNCH_CHECK();
    if (p->want_mom_event && p->rowmom_fused && d_J) { HIPCHK(hipEventRecord(p->ev_mom, s)); p->mom_event_recorded = true; }     // the row moments exist from here on
    if (p->timing && st_rows >= 0) { hipEventRecord(p->ev[st_rows][1], s); p->ev_valid[st_rows] = true; tl_klog = klog_outer; }
    if (p->timing && st_cols >= 0) { hipEventRecord(p->ev[st_cols][0], s); tl_klog = &p->stage_kernels[st_cols]; }
    const int npairs = (p->Nh + 1) / 2;
    int k = 0;
    while (k < nst) {          // whole stages per launch, so that a stage tile's readers sit next to each other in the grid
        ColOuts g; memset(&g, 0, sizeof(g));
        while (k < nst && g.nout + (int)stages[k].outs.size() <= COLG_MAX_OUT) {
            for (const Out& o : stages[k].outs) { g.stage_plane[g.nout] = k; g.out_plane[g.nout] = o.plane; g.wx[g.nout] = o.wx; ++g.nout; }
            ++k;
        }
        if (g.nout == 0) return set_err(SFFT_ERR_INVALID_ARG, "too many planes share one column factor for the weighted column pass");
        if (zpath) {
            const int total = npairs * g.nout;
            SFFT_LAUNCH(cols_fwd_weighted_4096_z, dim3(8 * ((total + 7) / 8)), dim3(512), 2 * Z4K_LDS * sizeof(double), s, p->d_stage, dst, g,
                               p->Nhp, p->lay.pstride, p->ax0.tw, npairs);
            continue;
        }
        if (p->colq && p->lay.rstride == 4 && p->lay.mask == 3) {      // four columns per workgroup: whole 64-byte pieces per lane quad
            const int nquads = (p->Nh + 3) / 4;
            const int total = nquads * g.nout;
            SFFT_LAUNCH(cols_fwd_weighted_4096_q, dim3(8 * ((total + 7) / 8)), dim3(512), (2 * F4K_LDS + 8) * sizeof(cplx), s, p->d_stage, dst, g,
                               p->Nhp, p->lay, p->ax0.tw, nquads);
            continue;
        }
        const int total = npairs * g.nout;
        SFFT_LAUNCH(cols_fwd_weighted_4096, dim3(8 * ((total + 7) / 8)), dim3(512), (2 * F4K_LDS + 8) * sizeof(cplx), s, p->d_stage, dst, g,
                           p->Nh, p->Nhp, p->lay, p->ax0.tw, npairs);
    }
    LAUNCH_CHECK();
    if (p->timing && st_cols >= 0) { hipEventRecord(p->ev[st_cols][1], s); p->ev_valid[st_cols] = true; tl_klog = klog_outer; }
    return SFFT_OK;
}

static int forward_basis_planes(sfft_plan* p, const double* d_I, const double* d_J, cplx* dst, hipStream_t s, bool with_sca = false,
                                int st_rows = -1, int st_cols = -1)
{
    // staged: one row transform per distinct column factor (the column pass applies the row factor); pays when planes share factors
    if (!p->no_staged)
        return forward_basis_planes_staged(p, d_I, d_J, dst, s, with_sca, st_rows, st_cols);
    const int total = p->Fij + (d_J ? 1 : 0) + ((with_sca && d_J) ? p->nsca : 0);
    const size_t plane_sz = (size_t)p->N0 * p->Nhp;
    for (int k0 = 0; k0 < total; k0 += SFFT_MAX_PLANES) {
        const int n = std::min(SFFT_MAX_PLANES, total - k0);
        RowsArgs ra;
        for (int u = 0; u < SFFT_MAX_PLANES; ++u) { ra.src[u] = nullptr; ra.wx[u] = nullptr; ra.wy[u] = nullptr; }
        for (int u = 0; u < n; ++u) {
            const int k = k0 + u;
            if (k < p->Fij) {
                ra.src[u] = d_I;
                ra.wx[u] = p->d_kbx + (size_t)p->kpair[2 * k] * p->N0;
                ra.wy[u] = p->d_kby + (size_t)p->kpair[2 * k + 1] * p->N1;
            } else if (k == p->Fij) ra.src[u] = d_J;
            else {
                const int sI = k - p->Fij - 1;
                ra.src[u] = d_I;
                ra.wx[u] = p->d_sbx + (size_t)p->spair[2 * sI] * p->N0;
                ra.wy[u] = p->d_sby + (size_t)p->spair[2 * sI + 1] * p->N1;
            }
        }
        int rc = forward_planes(p, ra, n, dst + (size_t)k0 * plane_sz, s, st_rows, st_cols);     // (more than one chunk: the last one is timed)
        if (rc) return rc;
    }
    return SFFT_OK;
}

template <int HBW, int U>
static void launch_g1(sfft_plan* p, int pass0, int npass, int h, hipStream_t s)
{
    const int ncb = (p->Nh + 63) / 64;
    const int total = ncb * p->S * npass;
    dim3 g(8 * ((total + 7) / 8));
    // even N0: one radix-2 decimation step along the rows (half the lag arithmetic; the chunks split the first half of the rows)
    const bool dit = p->g1_dit && (p->N0 % 2) == 0 && (HBW % 2) == 0 && p->N0 / (2 * p->S) >= 8;
    const int rpc2 = (p->N0 / 2 + p->S - 1) / p->S;
    for (int rb = 0; rb < h || rb == 0; rb += HBW) {
        if (dit) SFFT_LAUNCH((greek_g1<HBW, U, true>), g, dim3(64), 0, s, p->d_spec, p->d_passes, pass0, p->d_gp, p->N0, p->Nh,
                                    p->Nhp, p->lay_spec, rpc2, rb, p->d_w0tab, p->hm, p->d_Xp, ncb, p->S, npass);
        else SFFT_LAUNCH((greek_g1<HBW, U>), g, dim3(64), 0, s, p->d_spec, p->d_passes, pass0, p->d_gp, p->N0, p->Nh,
                                p->Nhp, p->lay_spec, p->rows_per_chunk, rb, p->d_w0tab, p->hm, p->d_Xp, ncb, p->S, npass);
    }
}

// lags per launch: the whole band in one launch up to 16 lags (64 accumulators per lane), else the split with the
// least padding and fewest launches
static int g1_band(int h)
{
    if (h <= 4) return 4;
    if (h <= 8) return 8;
    if (h <= 12) return 12;
    if (h <= 16) return 16;
    int best = 16, best_cost = 1 << 30;
    for (int b = 16; b >= 8; b -= 4) {
        const int launches = (h + b - 1) / b;
        const int cost = launches * b * 4 + launches;      // computed lags dominate; ties go to fewer launches
        if (cost < best_cost) { best_cost = cost; best = b; }
    }
    return best;
}
static int g1_padded(int h) { const int b = g1_band(h); return ((std::max(h, 1) + b - 1) / b) * b; }

// the grouped Omega launch runs with the radix-2 decimation step (greek_g1_mfma4g<false, true>): whole chunks of an even number of 8-row steps
static bool g1_decimated(const sfft_plan* p)
{
    // whole 8-row steps of x' in every chunk, the (possibly shorter) last one included
    return p->g1_mfma >= 3 && p->d_groups && p->g1_dit && (p->N0 % 16) == 0 && (p->rows_per_chunk % 16) == 0;
}

static int greek_g1_group(sfft_plan* p, int pass0, int npass, int h, hipStream_t s, bool planes_only = false)
{
    if (npass <= 0) return SFFT_OK;
    (void)planes_only;
    // 9 .. 16 lags (the Omega passes at KerHW 5 .. 8): on the matrix cores.  For <= 8 lags the pass is not FMA bound and the
    // vector kernel is as fast or faster (measured 0.33 vs 0.34 - 0.38 ms with the 8-lag packing <NT, true>).
    const bool grouped = p->g1_mfma >= 3 && p->d_groups && pass0 == 0 && npass == p->n_omg_launch;
    if (h >= 9 && (h <= 16 || (h <= 32 && grouped)) && p->g1_mfma) {
        const int ncb = (p->Nh + 31) / 32;
        const int total = ncb * p->S * npass;
        if (grouped) {
            const int ncb16 = (p->Nh + 15) / 16;
            const int totg = ncb16 * p->S * p->n_groups;
            const bool whole = (p->rows_per_chunk % 8) == 0 && (p->N0 % p->rows_per_chunk) == 0;      // no step runs past its chunk
            const bool dit = g1_decimated(p);
            // lag half-widths beyond 16 (KerHW 9 .. 16): the 16 lags lag0 + 1 .. lag0 + 16 per launch (the planes are read once per launch)
            for (int lag0 = 0; lag0 < h; lag0 += 16) {
            // (Theta passes as slots of their own groups: in the first launch only, behind the Omega groups)
            const int ngl = (lag0 == 0 || !p->theta_slots) ? p->n_groups : p->n_groups_omg;
            const int totl = ncb16 * p->S * ngl;
            if (dit && lag0 > 0 && h - lag0 <= 8)
                SFFT_LAUNCH((greek_g1_mfma4g<false, true, true>), dim3(8 * ((totl + 7) / 8)), dim3(64), 0, s, p->d_spec, p->d_passes, p->d_groups, ngl, p->d_gp,
                                   p->N0, p->Nh, p->Nhp, p->lay_spec, p->rows_per_chunk, p->d_w0tab, p->hm, ncb16, p->S, p->d_g1trace, lag0);
            else if (dit)
                SFFT_LAUNCH((greek_g1_mfma4g<false, true>), dim3(8 * ((totl + 7) / 8)), dim3(64), 0, s, p->d_spec, p->d_passes, p->d_groups, ngl, p->d_gp,
                                   p->N0, p->Nh, p->Nhp, p->lay_spec, p->rows_per_chunk, p->d_w0tab, p->hm, ncb16, p->S, p->d_g1trace, lag0);
            else if (whole)
                SFFT_LAUNCH(greek_g1_mfma4g<false>, dim3(8 * ((totl + 7) / 8)), dim3(64), 0, s, p->d_spec, p->d_passes, p->d_groups, ngl, p->d_gp,
                                   p->N0, p->Nh, p->Nhp, p->lay_spec, p->rows_per_chunk, p->d_w0tab, p->hm, ncb16, p->S, p->d_g1trace, lag0);
            else
                SFFT_LAUNCH(greek_g1_mfma4g<true>, dim3(8 * ((totl + 7) / 8)), dim3(64), 0, s, p->d_spec, p->d_passes, p->d_groups, ngl, p->d_gp,
                                   p->N0, p->Nh, p->Nhp, p->lay_spec, p->rows_per_chunk, p->d_w0tab, p->hm, ncb16, p->S, p->d_g1trace, lag0);
            }
            if (p->d_g1trace) {     // development aid (SFFT_G1_TRACE=file): dump the wave stamps of this launch
                hipStreamSynchronize(s);
                std::vector<unsigned long long> h((size_t)3 * std::min(65536, 8 * ((totg + 7) / 8)));      // (the trace buffer holds 65536 waves: larger launches are truncated)
                if (hipMemcpy(h.data(), p->d_g1trace, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess)
                    if (FILE* f = fopen(getenv("SFFT_G1_TRACE"), "w")) { for (size_t k = 0; k + 2 < h.size(); k += 3) fprintf(f, "%llu %llu %llu\n", h[k], h[k + 1], h[k + 2]); fclose(f); }
            }
        } else
            SFFT_LAUNCH((greek_g1_mfma4<2>), dim3(8 * ((total + 7) / 8)), dim3(64), 0, s, p->d_spec, p->d_passes, pass0, p->d_gp, p->N0,
                               p->Nh, p->Nhp, p->lay_spec, p->rows_per_chunk, p->d_w0tab, p->hm, p->d_Xp, ncb, p->S, npass);
        LAUNCH_CHECK();
        return SFFT_OK;
    }
    switch (g1_band(h)) {
        case 4: launch_g1<4, 4>(p, pass0, npass, h, s); break;
        case 8: launch_g1<8, 2>(p, pass0, npass, h, s); break;
        case 12: launch_g1<12, 2>(p, pass0, npass, h, s); break;
        default: launch_g1<16, 2>(p, pass0, npass, h, s); break;
    }
    LAUNCH_CHECK();
    return SFFT_OK;
}

static int run_fill(sfft_plan* p, hipStream_t s, bool lower_only)
{
    const int n = p->NEQfs;
    dim3 g((n + 1 + 15) / 16, (n + 1 + 15) / 16);
    SFFT_LAUNCH(fill_system, g, dim3(256), 0, s, p->d_patches, p->d_phi, p->d_delta, p->fa, p->d_idx, n, p->NEQ,
                       p->d_A, p->ld, (double*)nullptr, lower_only ? 1 : 0);
    LAUNCH_CHECK();
    return SFFT_OK;
}

__global__ void set_i32(int* __restrict__ v, int value) { *v = value; }

__global__ void zero_f64(double* __restrict__ v, size_t n)
{
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) v[k] = 0.0;
}

// L^T x = y on the factor's layout (lower triangle, border row n = y, rd = reciprocal diagonal), then Extend_Solution.  The pivoted LU
// (lu.hpp) hands its U over in the same layout (lu_transpose_upper) and ends here too.
static int run_back_substitution(sfft_plan* p, double* d_solution, hipStream_t s, bool have_winv)
{
    const int n = p->NEQfs;
    // Extend_Solution's zeros (removed unknowns stay exactly 0).  A kernel, not hipMemsetAsync: captured into the plan's hipGraph a
    // memset node was seen to leave these entries unwritten now and then when several plans replay their graphs from different
    // host threads at once (bench.py --pairs: 5 forbidden entries of a pair's Solution holding stale bytes); SFFT_SOL_MEMSET=1 restores it
    SFFT_LAUNCH(zero_f64, dim3((p->NEQ + 255) / 256), dim3(256), 0, s, d_solution, (size_t)p->NEQ);
    const int nblk = (n + CB - 1) / CB;
    if (p->back_variant == 1) {
        if (!have_winv)     // (chol_dataflow leaves the inverses of the diagonal blocks behind itself)
            SFFT_LAUNCH(chol_inv_diag, dim3(nblk), dim3(256), 0, s, p->d_A, p->ld, n, p->d_rd, p->d_winv);
        SFFT_LAUNCH(chol_back_all, dim3(nblk), dim3(256), 0, s, p->d_A, p->ld, n, p->d_winv, p->d_xv, p->d_bflags, p->d_epoch, p->d_status);
    } else
    for (int b = nblk - 1; b >= 0; --b) {
        const int kb = b * CB, nb = std::min(CB, n - kb);
        const int rows_below = n - (kb + nb);
        const int nslice = rows_below > 0 ? std::min(BACK_SLICES, (rows_below + CB - 1) / CB) : 1;
        SFFT_LAUNCH(chol_back_step, dim3(nslice), dim3(256), 0, s, p->d_A, p->ld, n, kb, p->d_xv, p->d_partial, p->d_counter, p->d_rd);
    }
    SFFT_LAUNCH(scatter_solution, dim3((n + 255) / 256), dim3(256), 0, s, p->d_xv, n, p->d_idx, d_solution, p->NEQ,
                       p->fa.tie_first, p->fa.tie_cnt, p->fa.tie_stride);
    LAUNCH_CHECK();
    return SFFT_OK;
}

static int run_cholesky_launches(sfft_plan* p, double* d_solution, hipStream_t s)
{
    const int n = p->NEQfs;
    const int nbc = (n + CB - 1) / CB;
    unsigned int* d_queue = p->d_tflags + (size_t)nbc * (nbc + 1);
    SFFT_LAUNCH(chol_begin, dim3(1), dim3(PANEL4_MAX_OUTER), 0, s, p->d_epoch, d_queue, p->d_pq);
    const bool dataflow = p->dataflow && n < p->chol_outer_min;
    if (dataflow) {
        // the whole factorisation as one launch of persistent workgroups (see chol_dataflow)
        const int ntask = 2 + (nbc - 1) * (nbc + 2) / 2;
        SFFT_LAUNCH(chol_dataflow, dim3(std::max(2, std::min(p->df_groups, ntask))), dim3(256), 0, s, p->d_A, p->ld, n, p->d_tflags, d_queue,
                           p->d_epoch, p->d_status, p->d_rd, p->d_w16, p->d_winv, p->d_trace);
    } else
    SFFT_LAUNCH(chol_copy_diag, dim3(1), dim3(256), 0, s, p->d_A, p->ld, std::min(CB, n), p->d_dbuf);
    int step = 0;
    int kb = dataflow ? n : 0;  // first column not yet factored
    if (p->fused_step && n >= p->chol_outer_min) {
        // outer blocks of 256 columns (see chol_syrk): the inner steps stay inside the block, one rank-256 update per block
        const int OB = 4 * CB;
        int outer = 0;
        while (n - kb >= OB + CB) {
            const int ntile4 = (n + 1 - kb + CB - 1) / CB;       // 64-row tiles of this block column (border row included)
            if (p->panel4 && p->d_pq && ntile4 <= p->ncu && outer < PANEL4_MAX_OUTER) {
                // the four panel steps as one launch of co-resident workgroups (chol_panel4)
                SFFT_LAUNCH(chol_panel4, dim3(ntile4), dim3(256), 0, s, p->d_A, p->ld, n, kb, p->d_pq + PANEL4_MAX_OUTER, p->d_pq, p->d_epoch,
                                   (unsigned)outer, p->d_status, p->d_rd, p->d_w16, nbc);
            } else {
            SFFT_LAUNCH(chol_panel, dim3(1 + (n + 1 - kb - CB + CB - 1) / CB), dim3(256), 0, s, p->d_A, p->ld, n, kb, p->d_dbuf, p->d_status, p->d_rd);
            for (int st = 1; st < 4; ++st) {
                const int k = kb + st * CB;
                const int ntile = (n + 1 - k + CB - 1) / CB;
                SFFT_LAUNCH(chol_step, dim3(4 - st, ntile), dim3(256), 0, s, p->d_A, p->ld, n, k - CB, p->d_dbuf + (size_t)2 * CB * CB,
                                   p->d_bflags + p->n_bflags - 1, p->d_epoch, (unsigned)((k / CB) % 255), p->d_status, p->d_rd);
            }
            }
            ++outer;
            const int r0 = kb + OB;
            const int nt = (n + 1 - r0 + 63) / 64;
            SFFT_LAUNCH(chol_syrk<64>, dim3(nt, nt), dim3(256), 0, s, p->d_A, p->ld, n, kb, OB, r0, 0);
            kb = r0;
            // chol_panel takes its diagonal block from the hand-over buffer, chol_panel4 from the matrix itself: the copy is only needed
            // when the next panel is a chol_panel launch (the last outer block's successor, or a block column too tall for chol_panel4)
            const bool next_p4 = (n - kb >= OB + CB) && p->panel4 && p->d_pq && (n + 1 - kb + CB - 1) / CB <= p->ncu && outer < PANEL4_MAX_OUTER;
            if (!next_p4)
                SFFT_LAUNCH(chol_copy_diag, dim3(1), dim3(256), 0, s, p->d_A + (size_t)kb * p->ld + kb, p->ld, std::min(CB, n - kb), p->d_dbuf);
        }
    }
    int k_start = kb;
    if (p->fused_step && n - kb >= 2 * CB) {
        // first panel only; steps with a full block: one fused launch each (update with the previous panel + this panel)
        {
            const int rows_below = n + 1 - kb - CB;
            SFFT_LAUNCH(chol_panel, dim3(1 + (rows_below + CB - 1) / CB), dim3(256), 0, s, p->d_A, p->ld, n, kb, p->d_dbuf, p->d_status, p->d_rd);
        }
        int k = kb + CB;
        for (; n - k >= CB; k += CB) {
            const int ntile = (n + 1 - k + CB - 1) / CB;
            SFFT_LAUNCH(chol_step, dim3(ntile, ntile), dim3(256), 0, s, p->d_A, p->ld, n, k - CB, p->d_dbuf + (size_t)2 * CB * CB,
                               p->d_bflags + p->n_bflags - 1, p->d_epoch, (unsigned)((k / CB) % 255), p->d_status, p->d_rd);
        }
        // what is left: the update with the last full panel (it also hands over the raw diagonal block) and the partial block
        if (k < n) {
            const int rows_below = n + 1 - k;
            const int ntile = (rows_below + CB - 1) / CB;
            SFFT_LAUNCH(chol_update, dim3(ntile, ntile), dim3(256), 0, s, p->d_A, p->ld, n, k - CB, p->d_dbuf);
        }
        k_start = k;
        step = 0;
    }
    for (int k = k_start; k < n; k += CB, ++step) {
        const int nb = std::min(CB, n - k);
        const int rows_below = n + 1 - (k + nb);
        const int nblk = 1 + (rows_below + CB - 1) / CB;
        double* Dcur = p->d_dbuf + (size_t)(step & 1) * CB * CB;
        double* Dnxt = p->d_dbuf + (size_t)((step + 1) & 1) * CB * CB;
        SFFT_LAUNCH(chol_panel, dim3(nblk), dim3(256), 0, s, p->d_A, p->ld, n, k, Dcur, p->d_status, p->d_rd);
        const int ntile = (rows_below + CB - 1) / CB;
        if (ntile > 0 && k + nb < n)
            SFFT_LAUNCH(chol_update, dim3(ntile, ntile), dim3(256), 0, s, p->d_A, p->ld, n, k, Dnxt);
    }
    LAUNCH_CHECK();
    LAUNCH_CHECK();
    return run_back_substitution(p, d_solution, s, dataflow);
}

// The factorisation and back substitution are ~35 dependent launches with constant arguments (the flag stamps come from a
// device counter): captured once per plan and replayed as one hipGraph.  Streams that cannot be captured (the legacy default
// stream) and any capture failure fall back to plain launches.
typedef int (*solver_chain_fn)(sfft_plan*, double*, hipStream_t);
static int run_chain_graph(sfft_plan* p, double* d_solution, hipStream_t s, solver_chain_fn chain, hipGraphExec_t* exec, std::string* kernels)
{
    if (!p->use_graph || s == nullptr || d_solution != p->d_sol) return chain(p, d_solution, s);
    if (!*exec) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            p->use_graph = 0;
            return chain(p, d_solution, s);
        }
        int rc;
        { KLogScope cap(kernels); kernels->clear(); rc = chain(p, d_solution, s); }
        const hipError_t e = hipStreamEndCapture(s, &graph);
        hipError_t e2 = hipSuccess;
        if (rc == SFFT_OK && e == hipSuccess && graph) e2 = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
        if (graph) hipGraphDestroy(graph);
        if (rc != SFFT_OK || e != hipSuccess || e2 != hipSuccess || !*exec) {
            (void)hipGetLastError();
            *exec = nullptr;
            p->use_graph = 0;
            if (rc != SFFT_OK) return rc;          // the chain itself refused (e.g. a system beyond the LU panel's reach): not a capture problem
            return chain(p, d_solution, s);
        }
    }
    if (tl_klog) note_kernels(*kernels);
    HIPCHK(hipGraphLaunch(*exec, s));
    return SFFT_OK;
}

static int run_cholesky(sfft_plan* p, double* d_solution, hipStream_t s)
{
    return run_chain_graph(p, d_solution, s, run_cholesky_launches, &p->chol_exec, &p->graph_kernels);
}

// ---- pivoted LU (lu.hpp): the reference's solver (SFFTSubtract.py:15-23) ---------------------------------------------------------
static void lu_note(const char* name) { if (tl_klog) note_kernel(name); }
static int run_lu_launches(sfft_plan* p, double* d_solution, hipStream_t s)
{
    const int n = p->NEQfs;
    if (n > LU_MAX_ROWS) return set_err(SFFT_ERR_UNSUPPORTED_SIZE, "linear system too large for the pivoted-LU panel of this build");
    const int npan = (n + LU_NB - 1) / LU_NB;
    unsigned int* d_queue = p->d_tflags + (size_t)npan * (npan + 1);
    SFFT_LAUNCH(chol_begin, dim3(1), dim3(PANEL4_MAX_OUTER), 0, s, p->d_epoch, d_queue, p->d_pq);       // a new stamp for chol_back_all's flags
    lu_factor_launches(p->d_A, p->ld, n, p->d_luperm, p->d_status, p->d_rd, p->d_luxchg, p->d_epoch, s, lu_note);
    LAUNCH_CHECK();
    return run_back_substitution(p, d_solution, s, false);
}

static int run_lu(sfft_plan* p, double* d_solution, hipStream_t s)
{
    return run_chain_graph(p, d_solution, s, run_lu_launches, &p->lu_exec, &p->lu_graph_kernels);
}

static int apply_prelim(sfft_plan* p, const double* d_I, cplx* dst, hipStream_t s);

// One attempt at the dense system, enqueued without a host sync: fill, factorisation (or pivoted LU), the status word read back
// into pinned host memory, the solution copied out.
static int solve_attempt(sfft_plan* p, bool use_lu, double* d_solution, hipStream_t s)
{
    int rc;
    {
        StageTimer t(p, SFFT_ST_FILL, s);
        SFFT_LAUNCH(set_i32, dim3(1), dim3(1), 0, s, p->d_status, 0);     // (a kernel like zero_f64, not a runtime memset)
        if ((rc = run_fill(p, s, !use_lu))) return rc;
    }
    {
        StageTimer t(p, SFFT_ST_SOLVE, s);
        if (use_lu) { if ((rc = run_lu(p, p->d_sol, s))) return rc; }
        else { if ((rc = run_cholesky(p, p->d_sol, s))) return rc; }
    }
    if (!use_lu && p->test_fail_chol) SFFT_LAUNCH(set_i32, dim3(1), dim3(1), 0, s, p->d_status, 1);      // test hook: pretend a pivot failed
    HIPCHK(hipMemcpyAsync(p->h_status, p->d_status, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(d_solution, p->d_sol, (size_t)p->NEQ * sizeof(double), hipMemcpyDeviceToDevice, s));
    p->attempt_lu = use_lu;
    return SFFT_OK;
}

// After the stream has been synchronised: if the Cholesky attempt met a non-positive pivot, redo the system with pivoted LU
// like the reference's gesv (and wait for it); *redone tells the caller that d_solution has changed.  A system that LU
// finds singular too is the reference's LinAlgError.
static int solve_check(sfft_plan* p, double* d_solution, hipStream_t s, bool* redone)
{
    if (redone) *redone = false;
    p->last_solver = p->attempt_lu ? 2 : 1;
    if (p->d_trace && !p->attempt_lu) {      // development aid (SFFT_DF_TRACE=1): critical-path stamps of chol_dataflow, in 10 ns ticks
        const int nbc = (p->NEQfs + CB - 1) / CB;
        std::vector<unsigned long long> h((size_t)nbc * 16);
        if (hipMemcpy(h.data(), p->d_trace, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            const unsigned long long t0 = h[8];
            for (int j = 0; j < nbc; ++j) {
                fprintf(stderr, "df_trace j=%d", j);
                for (int k = 0; k < 13; ++k) fprintf(stderr, " %lld", h[(size_t)j * 16 + k] ? (long long)(h[(size_t)j * 16 + k] - t0) : -1LL);
                fprintf(stderr, "\n");
            }
        }
    }
    if (!p->attempt_lu) p->chol_status = *p->h_status;
    ++p->n_solves;
    if (!p->attempt_lu && (*p->h_status & 4)) ++p->n_chol_stall;
    if (*p->h_status == 0) return SFFT_OK;
    int rc;
    // What a nonzero status means after the pivoted LU: bit 2 = no nonzero pivot in some column (the reference's LinAlgError: "Singular matrix");
    // bit 4 alone = a bounded hand-off poll of lu_panel_mw ran out -- its workgroups were not co-resident beside other streams' kernels --
    // which says nothing about the system: the attempt is repeated once, then refused as a stall, never reported as singular.
    auto lu_outcome = [&]() -> int {
        if (*p->h_status == 0) return SFFT_OK;
        if (*p->h_status & 2) return set_err(SFFT_ERR_SINGULAR, "Singular matrix");
        fprintf(stderr, "sfft_amd: the pivoted LU reported status %d (4: a hand-off poll between a panel's workgroups timed out); repeating the attempt\n", *p->h_status);
        int rc2;
        if ((rc2 = solve_attempt(p, true, d_solution, s))) return rc2;
        HIPCHK(hipStreamSynchronize(s));
        if (*p->h_status == 0) { if (redone) *redone = true; return SFFT_OK; }
        if (*p->h_status & 2) return set_err(SFFT_ERR_SINGULAR, "Singular matrix");
        return set_err(SFFT_ERR_STALL, "pivoted LU: a hand-off poll between the workgroups of a panel timed out twice (a scheduling stall on a busy GPU, not a singular system); call again");
    };
    if (p->attempt_lu) return lu_outcome();
    if (*p->h_status & 12)      // not a property of the system: say so before the pivoted LU takes over
        fprintf(stderr, "sfft_amd: chol_dataflow reported status %d (4: hand-off poll timed out, 8: grid < 2 workgroups); solving by LU\n", *p->h_status);
    ++p->n_lu_fallback;
    if ((rc = solve_attempt(p, true, d_solution, s))) return rc;
    HIPCHK(hipStreamSynchronize(s));
    p->last_solver = 2;
    if ((rc = lu_outcome())) return rc;
    if (redone) *redone = true;
    return SFFT_OK;
}

// SingleSFFTConfigure + ESS(Subtract=False) on device images.  defer_check: return with everything enqueued and leave the
// status check (solve_check, after a sync) to the caller -- sfft_subtract syncs once, at the end of the pair.
static int solve_impl(sfft_plan* p, const double* d_I, const double* d_J, double* d_solution, hipStream_t s, bool defer_check)
{
    ON_DEVICE(p->dev);
    int rc;
    p->launch_error = 0;     // per call: one refused launch must not poison every later call on the plan
    // The fused row moments give each row-pass group (one per distinct source image) ONE moment output.  solve(I, I) has a single
    // group, so only one of the two moment sets would be written: such a call takes the separate row_moments launches.
    struct ScopedInt { int& r; int old; ScopedInt(int& ref, int v) : r(ref), old(ref) { r = v; } ~ScopedInt() { r = old; } }
        rowmom_scope(p->rowmom_fused, d_I == d_J ? 0 : p->rowmom_fused);
    const bool gamma_aside = p->gamma_analytic && p->rowmom_fused && p->s2 && s != nullptr;
    // Without the fused row moments (every shape but 4096^2, and solve(I, I)) the Gamma block needs nothing but the image itself: its three
    // kernels (row moments of I, gamma_rows, gamma_patches: 0.42 ms at config 3, 0.60 ms at config 5, a few workgroups each) go to the second
    // stream right at the start of the solve, beside the forward transforms, instead of sitting on the main stream behind the Omega launch
    const bool gamma_early = p->gamma_analytic && !p->rowmom_fused && p->s2 && s != nullptr;
    if (gamma_early) {
        const int nd = p->gam_nmu, NQB = p->gam_db + 1, NJ = p->gam_tab ? p->nky : p->DK + 1;
        HIPCHK(hipEventRecord(p->ev_mom, s));                      // (behind the previous pair's fill_system, which reads the patches)
        HIPCHK(hipStreamWaitEvent(p->s2, p->ev_mom, 0));
#define ROWMOM_I(NQT) SFFT_LAUNCH(row_moments<NQT>, dim3((p->N0 + ROWMOM_R - 1) / ROWMOM_R), dim3(256), 0, p->s2, d_I, p->d_rowmomI, p->N0, p->N1, p->d_cyp, nd)
        switch (nd) {
            case 1: ROWMOM_I(1); break; case 2: ROWMOM_I(2); break; case 3: ROWMOM_I(3); break; case 4: ROWMOM_I(4); break;
            case 5: ROWMOM_I(5); break; case 6: ROWMOM_I(6); break; case 7: ROWMOM_I(7); break; default: ROWMOM_I(SFFT_MAX_BQ); break;
        }
#undef ROWMOM_I
        SFFT_LAUNCH(gamma_rows, dim3((p->N0 + 255) / 256, NJ * NQB), dim3(256), 0, p->s2, d_I, p->d_rowmomI, p->d_tby,
                           p->gam_tab ? p->d_kby : (const double*)nullptr, nd, p->gam_db, p->w, p->N0, p->N1, p->d_gamR);
        SFFT_LAUNCH(gamma_patches, dim3(p->Fij * p->Fpq, 2 * p->w + 1), dim3(256), 0, p->s2, p->d_gamR, p->d_kbx, p->d_tbx, p->ga, NQB,
                           p->N0, p->d_patches + p->fa.gam_off, p->scale * p->scale);
        LAUNCH_CHECK();
        HIPCHK(hipEventRecord(p->ev_gam, p->s2));
    }
    {
        StageTimer t(p, SFFT_ST_PRELIM_SOLVE, s);
        p->want_mom_event = gamma_aside; p->mom_event_recorded = false;
        rc = forward_basis_planes(p, d_I, d_J, p->d_spec, s, true, SFFT_ST_FWD_ROWS, SFFT_ST_FWD_COLS);
        p->want_mom_event = false;
        if (rc) return rc;
#define ROWMOM_J(NQT) SFFT_LAUNCH(row_moments<NQT>, dim3((p->N0 + ROWMOM_R - 1) / ROWMOM_R), dim3(256), 0, s, d_J, p->d_rowmom, p->N0, p->N1, p->d_tby, p->nby)
        if (p->rowmom_fused) { /* written by rows_r2c_4096 */ }
        else if (p->nby == 1) ROWMOM_J(1); else if (p->nby == 2) ROWMOM_J(2); else if (p->nby == 3) ROWMOM_J(3); else if (p->nby == 4) ROWMOM_J(4);
#undef ROWMOM_J
        else SFFT_LAUNCH(row_moments<SFFT_MAX_BQ>, dim3((p->N0 + ROWMOM_R - 1) / ROWMOM_R), dim3(256), 0, s, d_J, p->d_rowmom, p->N0, p->N1, p->d_tby, p->nby);
        SFFT_LAUNCH(delta_finish, dim3(p->Fpq), dim3(256), 0, s, p->d_rowmom, p->d_delta, p->N0, p->bk, p->scale);
        LAUNCH_CHECK();
    }
    // The real-space Gamma block (two small kernels on the row moments) does not depend on the spectra: when the moments came out of
    // the row pass it runs on the plan's second stream and joins before the system is filled.  On the 4096^2 fast path its start event
    // sits right behind the ROW pass, so it runs beside the column pass and is long done when fill_system is due; beside the Omega
    // launch (rounds 2 - 4) its one-thread-per-row kernel stretched from 27 to 342 us and fill_system waited ~28 us for it
    // (profiles/r05_timeline_cfg2_one_pair.txt)
    if (gamma_aside) {
        const int nd = p->gam_nmu, NQB = p->gam_db + 1, NJ = p->gam_tab ? p->nky : p->DK + 1;
        if (!p->mom_event_recorded) HIPCHK(hipEventRecord(p->ev_mom, s));
        HIPCHK(hipStreamWaitEvent(p->s2, p->ev_mom, 0));
        SFFT_LAUNCH(gamma_rows, dim3((p->N0 + 255) / 256, NJ * NQB), dim3(256), 0, p->s2, d_I, p->d_rowmomI, p->d_tby,
                           p->gam_tab ? p->d_kby : (const double*)nullptr, nd, p->gam_db, p->w, p->N0, p->N1, p->d_gamR);
        SFFT_LAUNCH(gamma_patches, dim3(p->Fij * p->Fpq, 2 * p->w + 1), dim3(256), 0, p->s2, p->d_gamR, p->d_kbx, p->d_tbx, p->ga, NQB,
                           p->N0, p->d_patches + p->fa.gam_off, p->scale * p->scale);
        LAUNCH_CHECK();
        HIPCHK(hipEventRecord(p->ev_gam, p->s2));
    }
    {
        StageTimer t(p, SFFT_ST_GREEK_G1, s);
        if ((rc = greek_g1_group(p, 0, p->n_omg_launch, 2 * p->w, s, true))) return rc;
    }
    {
        StageTimer t(p, SFFT_ST_GREEK_G1B, s);
        if (!p->sprods.empty()) {      // Omega products of (nearly) disjoint basis terms: real-space correlations of a few image rows / columns
            const int hO = 2 * p->w;
            if (p->n_scols) SFFT_LAUNCH(gather_cols, dim3((p->n_scols + 63) / 64, (p->N0 + 3) / 4), dim3(256), 0, s, d_I, p->d_scols, p->n_scols, p->d_strip, p->N0, p->N1);
            SFFT_LAUNCH(omega_sparse, dim3((unsigned)p->n_sitems), dim3(OSP_NT), 0, s, d_I, p->d_strip, p->d_sprods, p->d_slines, p->d_sitems,
                        p->d_kbx, p->d_kby, p->N0, p->N1, hO, p->scale * p->scale * p->scale, p->d_patches);
            LAUNCH_CHECK();
        }
        {
            const int fused = ((p->theta_in_groups || p->theta_slots) && p->g1_mfma >= 3) ? p->n_the_fused : 0;      // (the rest: a vector launch of their own)
            if (fused < p->n_dense_w && (rc = greek_g1_group(p, p->n_omg_rec + fused, p->n_dense_w - fused, p->w, s))) return rc;
        }
        if (p->gamma_analytic && !gamma_aside && !gamma_early) {     // Gamma block: row moments of I, then the patches (no spectra involved)
            const int nd = p->gam_nmu, NQB = p->gam_db + 1, NJ = p->gam_tab ? p->nky : p->DK + 1;
#define ROWMOM_I(NQT) SFFT_LAUNCH(row_moments<NQT>, dim3((p->N0 + ROWMOM_R - 1) / ROWMOM_R), dim3(256), 0, s, d_I, p->d_rowmomI, p->N0, p->N1, p->d_cyp, nd)
            if (!p->rowmom_fused)
            switch (nd) {           // exactly nd table values per column (a larger template bound re-reads clamped copies)
                case 1: ROWMOM_I(1); break; case 2: ROWMOM_I(2); break; case 3: ROWMOM_I(3); break; case 4: ROWMOM_I(4); break;
                case 5: ROWMOM_I(5); break; case 6: ROWMOM_I(6); break; case 7: ROWMOM_I(7); break; default: ROWMOM_I(SFFT_MAX_BQ); break;
            }
#undef ROWMOM_I
            SFFT_LAUNCH(gamma_rows, dim3((p->N0 + 255) / 256, NJ * NQB), dim3(256), 0, s, d_I, p->d_rowmomI, p->d_tby,
                               p->gam_tab ? p->d_kby : (const double*)nullptr, nd, p->gam_db, p->w, p->N0, p->N1, p->d_gamR);
            SFFT_LAUNCH(gamma_patches, dim3(p->Fij * p->Fpq, 2 * p->w + 1), dim3(256), 0, s, p->d_gamR, p->d_kbx, p->d_tbx, p->ga, NQB,
                               p->N0, p->d_patches + p->fa.gam_off, p->scale * p->scale);
            LAUNCH_CHECK();
        }
        if (p->n_row0 > 0) {
            SFFT_LAUNCH(greek_g1_row0, dim3((p->Nh + 255) / 256, p->n_row0), dim3(256), 0, s, p->d_spec, p->d_passes,
                               p->n_omg_rec + p->n_dense_w, p->d_gp, p->N0, p->Nh, p->Nhp, p->lay_spec, p->S);
            LAUNCH_CHECK();
        }
    }
    {
        StageTimer t(p, SFFT_ST_GREEK_G2, s);
        // one launch for all patch jobs: the grid is as tall as the widest patch (4 w + 1 rows), the workgroups past a narrower
        // job's 2 w + 1 rows return at once (two launches ran one after the other, each far from filling the chip)
        if (p->S <= 4)
            SFFT_LAUNCH(greek_g2<4>, dim3(4 * p->w + 1, (int)p->jobs.size()), dim3(256), 0, s, p->d_gp, p->d_passes, p->d_jobs, 0, p->d_patches,
                               p->Nh, p->Nhp, p->N1, p->S, p->ax1.root, p->d_Yq, p->scale);
        else
            SFFT_LAUNCH(greek_g2<8>, dim3(4 * p->w + 1, (int)p->jobs.size()), dim3(256), 0, s, p->d_gp, p->d_passes, p->d_jobs, 0, p->d_patches,
                               p->Nh, p->Nhp, p->N1, p->S, p->ax1.root, p->d_Yq, p->scale);
        LAUNCH_CHECK();
    }
    if (gamma_aside || gamma_early) HIPCHK(hipStreamWaitEvent(s, p->ev_gam, 0));
    p->have_system = true;
    if (p->overlap_I) {      // sfft_subtract: start the full pair's forward transforms now, beside the dense solve
        const double* dI = p->overlap_I;
        p->overlap_I = nullptr;
        HIPCHK(hipEventRecord(p->ev_in, s));
        HIPCHK(hipStreamWaitEvent(p->s2, p->ev_in, 0));
        if ((rc = apply_prelim(p, dI, p->d_spec2, p->s2))) return rc;
        HIPCHK(hipEventRecord(p->ev_pre, p->s2));
    }
    if ((rc = solve_attempt(p, p->force_lu != 0, d_solution, s))) return rc;
    if (p->launch_error) return p->launch_error;
    if (defer_check) return SFFT_OK;
    HIPCHK(hipStreamSynchronize(s));
    return solve_check(p, d_solution, s, nullptr);
}

extern "C" int sfft_solve(sfft_plan* p, const double* d_I, const double* d_J, double* d_solution, void* stream)
{
    if (!p || !d_I || !d_J || !d_solution) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    return solve_impl(p, d_I, d_J, d_solution, (hipStream_t)stream, false);
}

// forward spectra of the polynomial-weighted planes of I for the apply pass, into `dst` ([Fij][N0][Nhp])
static int apply_prelim(sfft_plan* p, const double* d_I, cplx* dst, hipStream_t s)
{
    StageTimer t(p, SFFT_ST_PRELIM_APPLY, s);
    if (p->use_vconv) {      // only the row pass: DK + 1 stage planes I * cy^j, j = 0 .. DK
        RowsArgs ra;
        for (int u = 0; u < SFFT_MAX_PLANES; ++u) { ra.src[u] = nullptr; ra.wx[u] = nullptr; ra.wy[u] = nullptr; }
        for (int jj = 0; jj < p->vncf; ++jj) { ra.src[jj] = d_I; ra.wy[jj] = p->d_kby + (size_t)jj * p->N1; }
        return forward_planes(p, ra, p->vncf, p->d_stage_a, s, -1, -1, true);
    }
    return forward_basis_planes(p, d_I, nullptr, dst, s);
}

// Construct_FDIFF + inverse transform + DIFF epilogue from the spectra FI; FD is a scratch plane
static int apply_finish(sfft_plan* p, const cplx* FI, cplx* FD, const double* d_I, const double* d_J, const double* d_solution,
                        double* d_diff, hipStream_t s)
{
    if (p->use_vconv) {
        // `FI` is then the stage buffer holding S_j = row-DFT(I cy^j), j = 0 .. DK, in its first DK + 1 planes
        StageTimer t(p, SFFT_ST_CONSTRUCT, s);
        const int LT = 2 * p->vw + 1;
        SFFT_LAUNCH(kernel_ctab_mixed, dim3((p->Nhp + 255) / 256, p->Fij * LT), dim3(256), 0, s, d_solution, p->d_ctabm, p->L, p->L, p->w,
                           p->w, p->vw, p->Nh, p->Nhp, p->N1, p->ax1.root, (double)p->N0 * p->scale);
        if (p->vtensor) {
            // tensor basis: 8-column tiles, 8 streams per wave; stream length from the same cost model (workgroups per CU x steps per stream)
            constexpr int CT = 8;
            const int ntile_t = (p->Nh + CT - 1) / CT;
            int Rt = 4 * LT - 2 * p->vw; double best = 1e30;
            for (int y = 1; y <= (p->N0 + 511) / 512; ++y) {
                const int Rc = (p->N0 + 32 * y - 1) / (32 * y);
                if (Rc < 16 && y > 1) break;
                const int wgs = ntile_t * y, rounds = (wgs + 2 * p->num_cu - 1) / (2 * p->num_cu);
                const int k = rounds > 1 ? rounds * 2 : (wgs + p->num_cu - 1) / p->num_cu;
                const double cost = (double)k * ((Rc + 2 * p->vw + 1) / 2) * (k == 1 ? 1.6 : 1.0);
                if (cost < best) { best = cost; Rt = Rc; }
            }
            if (p->vconv_r > 0) Rt = p->vconv_r;
            const int nstr = (p->N0 + Rt - 1) / Rt;
            dim3 gt(ntile_t, (nstr + 31) / 32);
            const size_t ldst = (size_t)p->Fij * LT * CT * sizeof(cplx);
            cplx* trash = p->d_ctabm + (size_t)p->Fij * LT * p->Nhp;
#define VT_LAUNCH(NN, WT, NA) do { \
            HIPCHK(hipFuncSetAttribute((const void*)vconv_tensor<NN, NN, WT, CT, NA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            SFFT_LAUNCH((vconv_tensor<NN, NN, WT, CT, NA>), gt, dim3(256), ldst, s, FI, FD, p->d_ctabm, p->d_kbx, p->d_ibase, p->N0, p->Nh, p->Nhp, p->lay, trash, Rt); } while (0)
#define VT_W(NN) do { if (p->vt_na == 3) { if (p->vw == 4) VT_LAUNCH(NN, 4, 3); else VT_LAUNCH(NN, 8, 3); } \
                      else { if (p->vw == 4) VT_LAUNCH(NN, 4, NN); else VT_LAUNCH(NN, 8, NN); } } while (0)
            switch (p->vtensor) { case 4: VT_W(4); break; case 5: VT_W(5); break; default: VT_W(6); break; }
#undef VT_W
#undef VT_LAUNCH
            LAUNCH_CHECK();
        } else {
        constexpr int KS = 4;       // source rows per stream = KS * L (3..10 measured: 4 is best at KerHW 8)
        int R = KS * LT - 2 * p->vw, Rrt = 0, ntile = (p->Nh + 15) / 16, m_direct = p->Nh;
        // the two-row walk (vconv_mixed2) for KerHW 9 .. 12 as well (two workgroups per CU there: 64 KB of weights for 10 terms x 25 taps):
        // config 5 construct 3.2 -> 1.8 ms; SFFT_VCONV2_W12=0 restores the one-row walk (vconv_mixed) for those widths
        const int vw2max = p->vconv2_w12 ? 12 : 8;
        if (p->vconv_rp == 2 && p->vw <= vw2max && p->vconv_r != 0) {
            // the walk is fp64-VALU bound and a workgroup puts one wave on each SIMD of its CU, so the launch takes
            // (workgroups per CU, rounded up) x (steps per stream): pick the stream length that minimises it.  A last tile of one or
            // two columns (the Nyquist column of an even N1) goes to vconv_direct, so that it does not cost a round of its own.
            const int rem = p->Nh % 16;
            if (rem >= 1 && rem <= 2 && ntile > 1) { m_direct = p->Nh - rem; --ntile; }
            if (p->vconv_r > 0) { R = Rrt = p->vconv_r; }
            else {
                double best = 1e30;
                for (int y = 1; y <= (p->N0 + 255) / 256; ++y) {
                    const int Rc = (p->N0 + 16 * y - 1) / (16 * y);
                    if (Rc < 16 && y > 1) break;
                    const int occ = (p->DK <= 2 || p->vw > 8) ? 2 : 3;          // workgroups resident per CU (launch bounds of vconv_mixed2)
                    const int wgs = ntile * y, rounds = (wgs + occ * p->num_cu - 1) / (occ * p->num_cu);
                    const int k = rounds > 1 ? rounds * occ : (wgs + p->num_cu - 1) / p->num_cu;
                    const double cost = (double)k * ((Rc + 2 * p->vw + 1) / 2) * (k == 1 ? 1.6 : 1.0);      // (a lone wave issues fp64 FMAs at 0.6 of the rate)
                    if (cost < best) { best = cost; R = Rrt = Rc; }
                }
            }
        }
        const int nstreams = (p->N0 + R - 1) / R;
        dim3 g(ntile, (nstreams + 15) / 16);
        const size_t lds = (size_t)p->Fij * LT * 16 * sizeof(cplx);
#define VCONV_LAUNCH(DKT, WT) do { \
        if (p->vconv_rp >= 2 && WT <= vw2max) { \
        HIPCHK(hipFuncSetAttribute((const void*)vconv_mixed2<DKT, WT, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        if (m_direct < p->Nh && p->vconv_direct_launch) SFFT_LAUNCH((vconv_direct<DKT, WT>), dim3((p->N0 + 255) / 256, p->Nh - m_direct), \
                                                 dim3(256), 0, s, FI, FD, p->d_ctabm, p->d_kbx, p->N0, p->Nh, p->Nhp, p->lay, m_direct); \
        SFFT_LAUNCH((vconv_mixed2<DKT, WT, KS>), g, dim3(256), lds, s, FI, FD, p->d_ctabm, p->d_kbx, p->N0, p->Nh, \
                           p->Nhp, p->lay, p->d_ctabm + (size_t)p->Fij * LT * p->Nhp, Rrt, p->vconv_direct_launch ? p->Nh : m_direct); } else { \
        HIPCHK(hipFuncSetAttribute((const void*)vconv_mixed<DKT, WT, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        SFFT_LAUNCH((vconv_mixed<DKT, WT, KS>), g, dim3(256), lds, s, FI, FD, p->d_ctabm, p->d_kbx, p->N0, p->Nh, \
                           p->Nhp, p->lay, p->d_ctabm + (size_t)p->Fij * LT * p->Nhp); } } while (0)
#define VCONV_DK(WT) switch (p->DK) { case 0: VCONV_LAUNCH(0, WT); break; case 1: VCONV_LAUNCH(1, WT); break; case 2: VCONV_LAUNCH(2, WT); break; default: VCONV_LAUNCH(3, WT); }
        if (p->vw == 4) { VCONV_DK(4) } else if (p->vw == 8) { VCONV_DK(8) } else { VCONV_DK(12) }
#undef VCONV_DK
#undef VCONV_LAUNCH
        LAUNCH_CHECK();
        }
    } else {
        StageTimer t(p, SFFT_ST_CONSTRUCT, s);
        SFFT_LAUNCH(kernel_rtab, dim3((p->N0 + 255) / 256, p->w + 1, p->Fij), dim3(256), 0, s, d_solution, p->d_rtab, p->Fij,
                           p->L, p->L, p->w, p->w, p->N0, 1 + 2 * p->wpad, p->ax0.root, p->mode == 3 ? 1 : 0);
        const int rpw = 32;             // rows per wave: 33 x 128 waves at 4096^2; fewer, longer waves measured slower
        dim3 g((p->Nh + 63) / 64, (p->N0 + rpw - 1) / rpw);
#define CONSTRUCT_LAUNCH(W, U, G) SFFT_LAUNCH((construct_fd<W, U, G>), g, dim3(64), 0, s, FI, FD, p->d_rtab, p->ax1.root, \
                                                     p->N0, p->N1, p->Nh, p->Nhp, p->lay, p->Fij, rpw, p->scale)
        switch (p->wpad) {
            case 4: CONSTRUCT_LAUNCH(4, 2, 3); break;
            case 8: CONSTRUCT_LAUNCH(8, 2, 3); break;
            case 12: CONSTRUCT_LAUNCH(12, 2, 3); break;
            case 16: CONSTRUCT_LAUNCH(16, 2, 3); break;
            case 24: CONSTRUCT_LAUNCH(24, 2, 2); break;
            default: CONSTRUCT_LAUNCH(32, 2, 2); break;
        }
#undef CONSTRUCT_LAUNCH
        LAUNCH_CHECK();
    }
    {
        StageTimer t(p, SFFT_ST_INVERSE, s);
        if (!p->use_vconv) launch_cols(p, FD, 1, 1, s);         // (the mixed-domain kernel already left the column-inverse in FD)
        if (p->ax1.big) {
            const int npr = (p->N0 + 1) / 2;
            SFFT_LAUNCH(retangle_rows, dim3((p->N1 + 255) / 256, npr), dim3(256), 0, s, FD, p->d_big1, p->N0, p->N1, p->Nh, p->Nhp);
            big_axis_transform(p, p->ax1, p->d_big1, p->d_big2, 1, p->N1, npr, false, 0, s);
            SFFT_LAUNCH(finish_diff, dim3((p->N1 + 255) / 256, npr), dim3(256), 0, s, p->d_big1, d_J, d_solution + p->Fijab, p->bk,
                               d_diff, p->N0, p->N1);
        } else if (fast_axis(p->ax1) && !p->no_fast_fft)
        {
            if (p->nby <= 4)
                SFFT_LAUNCH(rows_c2r_diff_4096<4>, dim3((p->N0 + 1) / 2), dim3(256), F4K_LDS * sizeof(cplx), s, FD, d_J,
                                   d_solution + p->Fijab, p->bk, d_diff, p->N0, p->lay, p->ax1.tw);
            else
                SFFT_LAUNCH(rows_c2r_diff_4096<SFFT_MAX_BQ>, dim3((p->N0 + 1) / 2), dim3(256), F4K_LDS * sizeof(cplx), s, FD, d_J,
                                   d_solution + p->Fijab, p->bk, d_diff, p->N0, p->lay, p->ax1.tw);
        }
        else if (p->rows_r24 && p->nby <= 4 && (p->inv_r24 == 1 || (p->inv_r24 < 0 && p->rows_r24 == 16))) {
            // 6144- / 9216-point rows: the register-resident inverse pass (fft_r24.hpp).  Alone it is faster (config 3 inverse 0.49 -> 0.30 ms,
            // config 5 1.17 -> 1.01 ms).  Pipelined, round 3 measured a loss (two pairs in flight: its 246 registers x 384 threads fill a
            // CU's register file where the generic pass leaves room for the other pair's Omega waves); with round 4's kernels and three
            // pairs in flight 6144-point rows gain 1 - 2 % (60.8 -> 61.4 pairs/s over three runs), 9216-point rows (84 spilled registers)
            // are flat.  Default: on for 6144, off for 9216; SFFT_INV_R24=0 / 1 forces it.
            if (p->rows_r24 == 16)
                SFFT_LAUNCH((rows_c2r_diff_r24<16, 4>), dim3((p->N0 + 1) / 2), dim3(R24<16>::NT), R24<16>::LDS * sizeof(double), s, FD, d_J,
                                   d_solution + p->Fijab, p->bk, d_diff, p->N0, p->lay, p->ax1.tw);
            else
                SFFT_LAUNCH((rows_c2r_diff_r24<24, 4>), dim3((p->N0 + 1) / 2), dim3(R24<24>::NT), R24<24>::LDS * sizeof(double), s, FD, d_J,
                                   d_solution + p->Fijab, p->bk, d_diff, p->N0, p->lay, p->ax1.tw);
        }
        else if (p->nby <= 4)
            SFFT_LAUNCH(rows_c2r_diff<4>, dim3((p->N0 + 1) / 2), dim3(p->nt_rows), p->lds_rows, s, FD, d_J,
                               d_solution + p->Fijab, p->bk, d_diff, p->N0, p->N1, p->Nh, p->Nhp, p->lay, axis_dev(p->ax1));
        else
            SFFT_LAUNCH(rows_c2r_diff<SFFT_MAX_BQ>, dim3((p->N0 + 1) / 2), dim3(p->nt_rows), p->lds_rows, s, FD, d_J,
                               d_solution + p->Fijab, p->bk, d_diff, p->N0, p->N1, p->Nh, p->Nhp, p->lay, axis_dev(p->ax1));
        if (p->mode == 3)
            SFFT_LAUNCH(scaling_term, dim3((p->N1 + 255) / 256, p->N0), dim3(256), 0, s, d_I, d_solution, p->sa, d_diff,
                               p->N0, p->N1, p->scale);
        LAUNCH_CHECK();
    }
    return SFFT_OK;
}

extern "C" int sfft_apply(sfft_plan* p, const double* d_I, const double* d_J, const double* d_solution, double* d_diff,
                          void* stream)
{
    if (!p || !d_I || !d_J || !d_solution || !d_diff) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    ON_DEVICE(p->dev);
    int rc;
    p->launch_error = 0;
    if ((rc = apply_prelim(p, d_I, p->d_spec, s))) return rc;
    if (p->launch_error) return p->launch_error;     // a pass was refused: do not run the rest of the pipeline on untransformed planes
    if ((rc = apply_finish(p, p->use_vconv ? p->d_stage_a : p->d_spec, p->d_spec + (size_t)p->Fij * p->N0 * p->Nhp, d_I, d_J, d_solution, d_diff, s))) return rc;
    return p->launch_error;
}

// GSS: the forward transforms of the full pair do not depend on the solution, and the dense solve leaves most
// CUs idle, so they run on the plan's second (low priority) stream while the caller's stream establishes and
// solves the system; the streams join before Construct_FDIFF.
extern "C" int sfft_subtract(sfft_plan* p, const double* d_I, const double* d_J, const double* d_mI, const double* d_mJ,
                             double* d_solution, double* d_diff, void* stream)
{
    if (!p || !d_I || !d_J || !d_mI || !d_mJ || !d_solution || !d_diff) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    ON_DEVICE(p->dev);
    int rc;
    // One host sync per pair: the solve is enqueued without waiting for its status word, the apply pass follows on the stream, and
    // the status is looked at after the final sync.  Only if the Cholesky attempt failed (LU fallback) is the apply pass redone.
    bool redone = false;
    if (!p->use_vconv && !p->d_spec2) {      // (the mixed-domain apply keeps its DK + 1 stage planes in d_stage_a instead)
        if ((rc = dev_alloc(p, &p->d_spec2, (size_t)p->Fij * p->N0 * p->Nhp))) return rc;
    }
    if (d_I == d_mI) {
        // the caller passed the full image as its own mask ("'same' means it is identical with I",
        // SFFTSubtract.py:849): the spectra of the solve pass are the spectra of the apply pass
        if ((rc = solve_impl(p, d_mI, d_mJ, d_solution, s, true))) return rc;
        // (mixed-domain apply: a staged solve pass left the stage planes of I first in d_stage; otherwise they are made now)
        // (colz: those stage planes have pair-major lines, which only the column pass reads)
        const bool reuse_stage = p->staged_solve && !p->colz;
        if (p->use_vconv && !reuse_stage && (rc = apply_prelim(p, d_I, p->d_spec, s))) return rc;
        const cplx* FIa = p->use_vconv ? (reuse_stage ? p->d_stage : p->d_stage_a) : p->d_spec;
        if ((rc = apply_finish(p, FIa, p->d_spec + (size_t)p->Fij * p->N0 * p->Nhp, d_I, d_J, d_solution, d_diff, s))) return rc;
        HIPCHK(hipStreamSynchronize(s));
        if ((rc = solve_check(p, d_solution, s, &redone))) return rc;
        if (redone) {
            if ((rc = apply_finish(p, FIa, p->d_spec + (size_t)p->Fij * p->N0 * p->Nhp, d_I, d_J, d_solution, d_diff, s))) return rc;
            HIPCHK(hipStreamSynchronize(s));
        }
        return SFFT_OK;
    }
    p->overlap_I = d_I;
    rc = solve_impl(p, d_mI, d_mJ, d_solution, s, true);
    p->overlap_I = nullptr;
    if (rc) { hipStreamSynchronize(p->s2); return rc; }
    HIPCHK(hipStreamWaitEvent(s, p->ev_pre, 0));
    const cplx* FIb = p->use_vconv ? p->d_stage_a : p->d_spec2;
    if ((rc = apply_finish(p, FIb, p->d_spec + (size_t)p->Fij * p->N0 * p->Nhp, d_I, d_J, d_solution, d_diff, s))) return rc;
    HIPCHK(hipStreamSynchronize(s));
    if ((rc = solve_check(p, d_solution, s, &redone))) return rc;
    if (redone) {
        if ((rc = apply_finish(p, FIb, p->d_spec + (size_t)p->Fij * p->N0 * p->Nhp, d_I, d_J, d_solution, d_diff, s))) return rc;
        HIPCHK(hipStreamSynchronize(s));
    }
    return SFFT_OK;
}

extern "C" int sfft_get_system(sfft_plan* p, double* d_LHMAT, double* d_RHb, void* stream)
{
    if (!p) return set_err(SFFT_ERR_INVALID_ARG, "NULL plan");
    if (!p->have_system) return set_err(SFFT_ERR_INVALID_ARG, "no linear system yet: call sfft_solve first");
    hipStream_t s = (hipStream_t)stream;
    ON_DEVICE(p->dev);
    dim3 g((p->NEQ + 15) / 16, (p->NEQ + 15) / 16);
    SFFT_LAUNCH(fill_plain, g, dim3(256), 0, s, p->d_patches, p->d_phi, p->d_delta, p->fa, p->NEQ, d_LHMAT, d_RHb);
    LAUNCH_CHECK();
    HIPCHK(hipStreamSynchronize(s));
    return SFFT_OK;
}

__global__ void iota_or_copy(const int* __restrict__ idx, int n, int* __restrict__ out)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[k] = idx ? idx[k] : k;
}

extern "C" int sfft_get_solver_system(sfft_plan* p, double* d_bordered, int* d_index, void* stream)
{
    if (!p || !d_bordered) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    if (!p->have_system) return set_err(SFFT_ERR_INVALID_ARG, "no linear system yet: call sfft_solve first");
    hipStream_t s = (hipStream_t)stream;
    ON_DEVICE(p->dev);
    const int n = p->NEQfs;
    dim3 g((n + 1 + 15) / 16, (n + 1 + 15) / 16);
    // the launch of run_fill with the caller's buffer (leading dimension n + 1) in place of the solver's workspace, both triangles
    SFFT_LAUNCH(fill_system, g, dim3(256), 0, s, p->d_patches, p->d_phi, p->d_delta, p->fa, p->d_idx, n, p->NEQ,
                       d_bordered, n + 1, (double*)nullptr, 0);
    if (d_index) SFFT_LAUNCH(iota_or_copy, dim3((n + 255) / 256), dim3(256), 0, s, p->d_idx, n, d_index);
    LAUNCH_CHECK();
    HIPCHK(hipStreamSynchronize(s));
    return SFFT_OK;
}

__global__ void __launch_bounds__(256) dbg_load_bordered(const double* __restrict__ src, int n, double* __restrict__ A, int ld)
{
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c <= n) A[(size_t)r * ld + c] = src[(size_t)r * (n + 1) + c];
}

extern "C" int sfft_dbg_solve_dense(sfft_plan* p, const double* d_bordered, int use_lu, double* d_x, void* stream)
{
    if (!p || !d_bordered || !d_x) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    ON_DEVICE(p->dev);
    const int n = p->NEQfs;
    int rc;
    SFFT_LAUNCH(set_i32, dim3(1), dim3(1), 0, s, p->d_status, 0);
    SFFT_LAUNCH(dbg_load_bordered, dim3((n + 1 + 255) / 256, n + 1), dim3(256), 0, s, d_bordered, n, p->d_A, p->ld);
    LAUNCH_CHECK();
    {
        StageTimer t(p, SFFT_ST_SOLVE, s);
        if (use_lu) { if ((rc = run_lu(p, p->d_sol, s))) return rc; }
        else { if ((rc = run_cholesky(p, p->d_sol, s))) return rc; }
    }
    HIPCHK(hipMemcpyAsync(p->h_status, p->d_status, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(d_x, p->d_xv, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    p->last_solver = use_lu ? 2 : 1;
    if (!use_lu) p->chol_status = *p->h_status;
    if (*p->h_status != 0 && !(*p->h_status & 3))       // bits 4 / 8 alone: a scheduling stall, not a property of the matrix
        return set_err(SFFT_ERR_STALL, "dense solve: a hand-off poll between workgroups timed out (status " + std::to_string(*p->h_status) + "); not a singular system, call again");
    if (*p->h_status != 0) return set_err(SFFT_ERR_SINGULAR, "Singular matrix");
    return SFFT_OK;
}

// ---- general real 2-D FFT entry points (used by the decorrelation / FFT-convolution utilities) ---------------------
extern "C" int sfft_fft_plan_create(sfft_plan** out, int N0, int N1, int device)
{
    return sfft_plan_create(out, N0, N1, 0, 0, 0, 0, device);     // a plan with the smallest SFFT geometry: the FFT machinery only
}

// d_spec[N0][N1/2+1] (dense complex128) = scale * DFT2(d_real[N0][N1]), numpy.fft.rfft2 convention
extern "C" int sfft_fft2_r2c(sfft_plan* p, const double* d_real, double* d_spec, double scale, void* stream)
{
    if (!p || !d_real || !d_spec) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    ON_DEVICE(p->dev);
    RowsArgs ra;
    for (int k = 0; k < SFFT_MAX_PLANES; ++k) { ra.src[k] = nullptr; ra.wx[k] = nullptr; ra.wy[k] = nullptr; }
    ra.src[0] = d_real;
    int rc = forward_planes(p, ra, 1, p->d_spec, s);
    if (rc) return rc;
    SFFT_LAUNCH(copy_spectrum_scaled, dim3((p->Nh + 255) / 256, p->N0), dim3(256), 0, s, p->d_spec, (cplx*)d_spec, p->N0, p->Nh,
                       p->lay, rowmajor_layout(p->Nh), scale / p->scale);
    LAUNCH_CHECK();
    return SFFT_OK;
}

// d_real[N0][N1] = scale * sum_k spec[k] e^{+2 pi i k x / N} from the half spectrum of a real image (numpy.fft.irfft2 * N0*N1
// when scale = 1; pass scale = 1/(N0*N1) for numpy's normalisation)
extern "C" int sfft_ifft2_c2r(sfft_plan* p, const double* d_spec, double* d_real, double scale, void* stream)
{
    if (!p || !d_real || !d_spec) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    ON_DEVICE(p->dev);
    int rc;
    if (!p->d_zero) {
        if ((rc = dev_alloc(p, &p->d_zero, (size_t)p->N0 * p->N1))) return rc;
        SFFT_LAUNCH(zero_f64, dim3((unsigned)(((size_t)p->N0 * p->N1 + 255) / 256)), dim3(256), 0, s, p->d_zero, (size_t)p->N0 * p->N1);
        if ((rc = dev_alloc(p, &p->d_zsol, (size_t)p->NEQ))) return rc;
        SFFT_LAUNCH(zero_f64, dim3((p->NEQ + 255) / 256), dim3(256), 0, s, p->d_zsol, (size_t)p->NEQ);
    }
    cplx* FD = p->d_spec + (size_t)p->Fij * p->N0 * p->Nhp;
    // reuse the inverse path of the subtraction: with J = 0 and b = 0 it returns -IDFT2(FD); the caller's factor (and that sign) rides on
    // the copy into the panel layout -- the transform is linear -- instead of a pass of its own over the image afterwards
    SFFT_LAUNCH(copy_spectrum_scaled, dim3((p->Nh + 255) / 256, p->N0), dim3(256), 0, s, (const cplx*)d_spec, FD, p->N0, p->Nh,
                       rowmajor_layout(p->Nh), p->lay, -scale);
    LAUNCH_CHECK();
    launch_cols(p, FD, 1, 1, s);
    if (p->ax1.big) {
        const int npr = (p->N0 + 1) / 2;
        SFFT_LAUNCH(retangle_rows, dim3((p->N1 + 255) / 256, npr), dim3(256), 0, s, FD, p->d_big1, p->N0, p->N1, p->Nh, p->Nhp);
        big_axis_transform(p, p->ax1, p->d_big1, p->d_big2, 1, p->N1, npr, false, 0, s);
        SFFT_LAUNCH(finish_diff, dim3((p->N1 + 255) / 256, npr), dim3(256), 0, s, p->d_big1, p->d_zero, p->d_zsol + p->Fijab, p->bk,
                           d_real, p->N0, p->N1);
    } else if (fast_axis(p->ax1) && !p->no_fast_fft)
        SFFT_LAUNCH(rows_c2r_diff_4096<4>, dim3((p->N0 + 1) / 2), dim3(256), F4K_LDS * sizeof(cplx), s, FD, p->d_zero,
                           p->d_zsol + p->Fijab, p->bk, d_real, p->N0, p->lay, p->ax1.tw);
    else if (p->rows_r24 == 16 && p->inv_r24 != 0)         // (nothing else is in flight here: the register-resident pass wins alone at both sizes)
        SFFT_LAUNCH((rows_c2r_diff_r24<16, 4>), dim3((p->N0 + 1) / 2), dim3(R24<16>::NT), R24<16>::LDS * sizeof(double), s, FD, p->d_zero,
                           p->d_zsol + p->Fijab, p->bk, d_real, p->N0, p->lay, p->ax1.tw);
    else if (p->rows_r24 == 24 && p->inv_r24 != 0)
        SFFT_LAUNCH((rows_c2r_diff_r24<24, 4>), dim3((p->N0 + 1) / 2), dim3(R24<24>::NT), R24<24>::LDS * sizeof(double), s, FD, p->d_zero,
                           p->d_zsol + p->Fijab, p->bk, d_real, p->N0, p->lay, p->ax1.tw);
    else
        SFFT_LAUNCH(rows_c2r_diff<4>, dim3((p->N0 + 1) / 2), dim3(p->nt_rows), p->lds_rows, s, FD, p->d_zero,
                           p->d_zsol + p->Fijab, p->bk, d_real, p->N0, p->N1, p->Nh, p->Nhp, p->lay, axis_dev(p->ax1));
    LAUNCH_CHECK();
    return SFFT_OK;
}

// acc[i] += coeff * |a[i]|^2 * |b[i]|^2 (b may be NULL); a, b complex128, acc float64, n elements
extern "C" int sfft_spec_abs2_accumulate(const double* d_a, const double* d_b, double coeff, double* d_acc, long long n, void* stream)
{
    if (!d_a || !d_acc || n < 0) return set_err(SFFT_ERR_INVALID_ARG, "bad argument");
    ON_DEVICE(stream_device((hipStream_t)stream));
    SFFT_LAUNCH(spec_abs2_acc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const cplx*)d_a, (const cplx*)d_b,
                       coeff, d_acc, (size_t)n);
    LAUNCH_CHECK();
    return SFFT_OK;
}

// out[i] = 1 / sqrt(acc[i])
extern "C" int sfft_real_rsqrt(const double* d_acc, double* d_out, long long n, void* stream)
{
    if (!d_acc || !d_out || n < 0) return set_err(SFFT_ERR_INVALID_ARG, "bad argument");
    ON_DEVICE(stream_device((hipStream_t)stream));
    SFFT_LAUNCH(real_rsqrt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_acc, d_out, (size_t)n);
    LAUNCH_CHECK();
    return SFFT_OK;
}

// out[i] = a[i] * b[i] with b complex (b_is_real = 0) or real (b_is_real = 1); a, out complex128
extern "C" int sfft_spec_multiply(const double* d_a, const double* d_b, int b_is_real, double* d_out, long long n, void* stream)
{
    if (!d_a || !d_b || !d_out || n < 0) return set_err(SFFT_ERR_INVALID_ARG, "bad argument");
    ON_DEVICE(stream_device((hipStream_t)stream));
    SFFT_LAUNCH(spec_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const cplx*)d_a,
                       b_is_real ? (const cplx*)nullptr : (const cplx*)d_b, b_is_real ? d_b : (const double*)nullptr, (cplx*)d_out, (size_t)n);
    LAUNCH_CHECK();
    return SFFT_OK;
}

// full [N0][N1] of a real, conjugate-symmetric spectrum quantity from its half [N0][N1/2+1]
extern "C" int sfft_half_to_full_real(const double* d_half, double* d_full, int N0, int N1, void* stream)
{
    if (!d_half || !d_full) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    ON_DEVICE(stream_device((hipStream_t)stream));
    SFFT_LAUNCH(half_to_full_real, dim3((N1 + 255) / 256, N0), dim3(256), 0, (hipStream_t)stream, d_half, d_full, N0, N1, N1 / 2 + 1);
    LAUNCH_CHECK();
    return SFFT_OK;
}

// BSpline_GridConvolve.GSVC_GPU (sfft/BSplineSFFT.py:4951-5006): every pixel is convolved with the kernel of its segment
extern "C" int sfft_grid_convolve(const double* d_in, const int* d_labels, const double* d_kerstack, int N0, int N1, int Nseg,
                                  int L0, int L1, double* d_out, int device, void* stream)
{
    if (!d_in || !d_labels || !d_kerstack || !d_out) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    if (N0 < 1 || N1 < 1 || Nseg < 1 || L0 < 1 || L1 < 1) return set_err(SFFT_ERR_INVALID_ARG, "bad size");
    // stamp rows per LDS band: the whole stamp when its halo fits 64 KB (two workgroups per CU), else as many rows as do
    const size_t row_bytes = (size_t)(16 + L1 - 1) * sizeof(double);
    if (16 * row_bytes > 150 * 1024) return set_err(SFFT_ERR_UNSUPPORTED_SIZE, "kernel stamp too wide for the on-chip band of this build");
    long long A = (long long)(64 * 1024 / row_bytes) - 15;
    if (A < 1) A = 1;
    if (A > L0) A = L0;
    const size_t lds = (size_t)(16 + A - 1) * row_bytes;
    ON_DEVICE(device);
    HIPCHK(hipFuncSetAttribute((const void*)grid_convolve, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SFFT_LAUNCH(grid_convolve, dim3((N1 + 15) / 16, (N0 + 15) / 16), dim3(256), lds, (hipStream_t)stream, d_in, d_labels, d_kerstack,
                       N0, N1, Nseg, L0, L1, (int)A, d_out);
    LAUNCH_CHECK();
    return SFFT_OK;
}

extern "C" int sfft_dbg_forward_spectrum(sfft_plan* p, const double* d_I, int i, int j, double* d_spec_out, void* stream)
{
    if (!p || !d_I || !d_spec_out) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    if (i < 0 || j < 0 || i >= p->nkx || j >= p->nky) return set_err(SFFT_ERR_INVALID_ARG, "basis factor index out of range for this plan");
    hipStream_t s = (hipStream_t)stream;
    ON_DEVICE(p->dev);
    RowsArgs ra;
    for (int k = 0; k < SFFT_MAX_PLANES; ++k) { ra.src[k] = nullptr; ra.wx[k] = nullptr; ra.wy[k] = nullptr; }
    ra.src[0] = d_I; ra.wx[0] = p->d_kbx + (size_t)i * p->N0; ra.wy[0] = p->d_kby + (size_t)j * p->N1;
    int rc = forward_planes(p, ra, 1, p->d_spec, s);
    if (rc) return rc;
    SFFT_LAUNCH(copy_spectrum, dim3((p->Nh + 255) / 256, p->N0), dim3(256), 0, s, p->d_spec, (cplx*)d_spec_out, p->N0, p->Nh, p->lay);
    LAUNCH_CHECK();
    HIPCHK(hipStreamSynchronize(s));
    return SFFT_OK;
}
