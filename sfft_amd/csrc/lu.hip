// lu.hip -- translation unit of the pivoted-LU kernels (lu.hpp) and their launchers (lu_api.hpp).  gfx950 only.
#include <hip/hip_runtime.h>
#include <type_traits>

typedef double d4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ d4s mfma16(double av, double bv, d4s acc) { return __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0); }

#include "lu.hpp"

template <int W, int R>
static const char* panel(const char* name, double* A, int ld, int n, int k0, int nb, LuPerm* perm, int* status, hipStream_t s)
{
    hipLaunchKernelGGL((lu_panel<W, R>), dim3(1), dim3(LU_NT), 0, s, A, ld, n, k0, nb, perm, status);
    return name;
}

// One workgroup, 16-column sub-panels, R rows per thread (256 threads): panels of up to LU_ONE_KERNEL_ROWS rows
static const char* launch_panel(double* A, int ld, int n, int k0, int nb, LuPerm* perm, int* status, hipStream_t s)
{
    const int rpt = (n - k0 + LU_NT - 1) / LU_NT;           // rows per thread
    if (rpt <= 2) return panel<16, 2>("lu_panel<16,2>", A, ld, n, k0, nb, perm, status, s);
    return panel<16, 4>("lu_panel<16,4>", A, ld, n, k0, nb, perm, status, s);
}

static inline void say(lu_note_fn note, const char* name) { if (note && name) note(name); }

// U12 = L11^-1 A12 (after the rows of `perm`, if any, went into place) and A22 -= L21 U12 on the columns [k0 + nb, cend)
static void update(double* A, int ld, int n, int k0, int nb, const LuPerm* perm, int cend, hipStream_t s, lu_note_fn note)
{
    const int right = cend - (k0 + nb), below = n - (k0 + nb);
    if (right <= 0) return;
    hipLaunchKernelGGL(lu_swap_trsm, dim3((right + LU_NB - 1) / LU_NB), dim3(256), 0, s, A, ld, n, k0, nb, perm, cend);
    say(note, "lu_swap_trsm");
    if (below > 0) {
        hipLaunchKernelGGL(lu_gemm, dim3((right + LU_NB - 1) / LU_NB, (below + LU_NB - 1) / LU_NB), dim3(256), 0, s, A, ld, n, k0, nb, cend);
        say(note, "lu_gemm");
    }
}

static void apply_perm(double* A, int ld, int n, const LuPerm* perm, int cbeg, int cend, hipStream_t s, lu_note_fn note)
{
    if (cend <= cbeg) return;
    hipLaunchKernelGGL(lu_apply_perm, dim3((cend - cbeg + LU_NB - 1) / LU_NB), dim3(256), 0, s, A, ld, n, perm, cbeg, cend);
    say(note, "lu_apply_perm");
}

// Panels of 64 columns.  Up to LU_ONE_KERNEL_ROWS rows a panel is ONE workgroup (16-column sub-panels inside the kernel, left-looking);
// taller panels run on G <= 16 workgroups (lu_panel_mw<R>, R rows per thread: as few as 16 workgroups allow -- the pivot chain's cost per
// column grows with them; below 1024 rows one workgroup is faster: 5.5 vs 5.1 ms at n = 1735 with the threshold at 512).
#define LU_ONE_KERNEL_ROWS (4 * LU_NT)
size_t lu_xchg_bytes() { return sizeof(LuXchg); }

void lu_factor_launches(double* A, int ld, int n, LuPerm* perms, int* status, double* rd, void* xchg, const unsigned int* epoch_ctr, hipStream_t s, lu_note_fn note)
{
    const int npan = (n + LU_NB - 1) / LU_NB;
    for (int pn = 0; pn < npan; ++pn) {
        const int K0 = pn * LU_NB, NBo = n - K0 < LU_NB ? n - K0 : LU_NB;
        LuPerm* pl = perms + (size_t)LU_PERMS_PER_PANEL * pn;
        if (n - K0 <= LU_ONE_KERNEL_ROWS) {
            say(note, launch_panel(A, ld, n, K0, NBo, pl, status, s));
            update(A, ld, n, K0, NBo, pl, n + 1, s, note);
            continue;
        }
        {
            const int m = n - K0;           // (m <= LU_MAX_ROWS = 6 x 256 x 16: the caller refuses larger systems)
            const int Rw = m <= 1 * LU_NT * LU_MW_MAXG ? 1 : m <= 2 * LU_NT * LU_MW_MAXG ? 2 : m <= 3 * LU_NT * LU_MW_MAXG ? 3 : m <= 4 * LU_NT * LU_MW_MAXG ? 4 : 6;
            const int G = (m + Rw * LU_NT - 1) / (Rw * LU_NT);
            hipLaunchKernelGGL(lu_perm_reset, dim3(1), dim3(64), 0, s, pl);
            // (8 G workgroups of which every eighth works: see lu_panel_mw)
#define LU_MW(RR) hipLaunchKernelGGL(lu_panel_mw<RR>, dim3(8 * G), dim3(LU_NT), 0, s, A, ld, n, K0, NBo, pl, status, (LuXchg*)xchg, epoch_ctr, pn)
            if (Rw == 1) LU_MW(1); else if (Rw == 2) LU_MW(2); else if (Rw == 3) LU_MW(3); else if (Rw == 4) LU_MW(4); else LU_MW(6);
#undef LU_MW
            say(note, "lu_panel_mw");
            apply_perm(A, ld, n, pl, K0, K0 + NBo, s, note);                // the panel's own columns: rows into place
            update(A, ld, n, K0, NBo, pl, n + 1, s, note);
        }
    }
    hipLaunchKernelGGL(lu_transpose_upper, dim3((n + 1 + LU_NB - 1) / LU_NB, npan), dim3(256), 0, s, A, ld, n, rd, status);
    say(note, "lu_transpose_upper");
}
