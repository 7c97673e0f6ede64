// sfft_amd.hip -- MI355X (gfx950) native SFFT subtraction core: kernels + C ABI (include/sfft_amd.h).
//
// Path covered (reference = thomasvrussell/sfft v1.7.3):
//   SpatialCoor/SpatialPoly   sfft/sfftcore/SFFTConfigure.py:84-145     fused into rows_r2c (no coordinate planes)
//   preliminary DFTs          sfft/sfftcore/SFFTSubtract.py:146-168     rows_r2c + cols_c2c (half spectrum, fp64)
//   HadProd_* + Greek DFTs    SFFTConfigure.py:150-662, SFFTSubtract.py:226-383   greek_g1 + greek_g2 (pruned to the lags FillLS reads)
//   FillLS_* + stripes        SFFTConfigure.py:198-711                  fill_system
//   LSSolver                  SFFTSubtract.py:15-23, 398-403            blocked Cholesky (LU with partial pivoting as fallback)
//   Extend_Solution           SFFTConfigure.py:716-732                  scatter_solution / lu_backsolve
//   twiddles + Construct_FDIFF SFFTSubtract.py:433-447, SFFTConfigure.py:737-809   kernel_ctab + construct_fd
//   inverse DFT               SFFTSubtract.py:460-461                   cols_c2c(inverse) + rows_c2r_diff
//
// Layout in HBM: images [N0][N1] f64 row-major; spectra [plane][N0][Nhp] complex128 with Nh = N1/2+1
// columns kept (the inputs are real, so the other half is the conjugate mirror) and Nhp >= Nh the padded
// row stride.  All arithmetic is IEEE fp64.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

extern "C" const char* sfft_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* sfft_version(void) { return "sfft_amd 0.1 (gfx950)"; }

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
__device__ __forceinline__ cplx cconj(cplx a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double ipow(double x, int e) { double r = 1.0; for (int t = 0; t < e; ++t) r *= x; return r; }

// In-place Stockham autosort FFT (forward, e^{-i}) of `nb` transforms of length M = 2^logM held in LDS at
// s + f*stride.  Radix-4 stages (one leading radix-2 stage when logM is odd).  Every thread of the block
// must call; requires nb*M <= 16*blockDim.x so that a thread owns at most 4 radix-4 butterflies per stage.
// tw[k] = exp(-2*pi*i*k/M), k < M (global memory, cached).
__device__ __forceinline__ void lds_fft(cplx* s, int M, int logM, int nb, int stride, const cplx* __restrict__ tw)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    int p = 1, logp = 0;
    if (logM & 1) {
        const int T = M >> 1, logT = logM - 1, total = nb * T;
        cplx u[8][2];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int g = tid + it * nt;
            if (g < total) {
                const int f = g >> logT, i = g & (T - 1);
                const cplx* b = s + f * stride;
                u[it][0] = b[i];
                u[it][1] = b[i + T];
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int g = tid + it * nt;
            if (g < total) {
                const int f = g >> logT, i = g & (T - 1);
                cplx* b = s + f * stride;
                b[2 * i] = cadd(u[it][0], u[it][1]);
                b[2 * i + 1] = csub(u[it][0], u[it][1]);
            }
        }
        __syncthreads();
        p = 2; logp = 1;
    }
    const int T = M >> 2, logT = logM - 2, total = nb * T;
    for (; p < M; p <<= 2, logp += 2) {
        cplx y[4][4];
        const int tshift = logM - logp - 2;   // twiddle step M/(4p)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int g = tid + it * nt;
            if (g < total) {
                const int f = g >> logT, i = g & (T - 1);
                const int k = i & (p - 1);
                const cplx* b = s + f * stride;
                cplx u0 = b[i], u1 = b[i + T], u2 = b[i + 2 * T], u3 = b[i + 3 * T];
                if (p > 1) {
                    const int q = k << tshift;
                    u1 = cmul(u1, tw[q]);
                    u2 = cmul(u2, tw[2 * q]);
                    u3 = cmul(u3, tw[3 * q]);
                }
                const cplx a02 = cadd(u0, u2), s02 = csub(u0, u2);
                const cplx a13 = cadd(u1, u3), s13 = csub(u1, u3);
                y[it][0] = cadd(a02, a13);
                y[it][2] = csub(a02, a13);
                // -i*(u1-u3) = (s13.y, -s13.x)
                y[it][1] = make_double2(s02.x + s13.y, s02.y - s13.x);
                y[it][3] = make_double2(s02.x - s13.y, s02.y + s13.x);
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int g = tid + it * nt;
            if (g < total) {
                const int f = g >> logT, i = g & (T - 1);
                const int k = i & (p - 1);
                cplx* b = s + f * stride + (((i - k) << 2) + k);
                b[0] = y[it][0];
                b[p] = y[it][1];
                b[2 * p] = y[it][2];
                b[3 * p] = y[it][3];
            }
        }
        __syncthreads();
    }
}

// One 1-D axis: length N transformed either directly (N = M power of two) or by Bluestein's chirp-z
// (M = power of two >= 2N-1).  All tables live in device memory.
struct AxisDev {
    int N, M, logM, blue;
    const cplx* tw;     // [M]   exp(-2 pi i k / M)
    const cplx* chirp;  // [N]   exp(-i pi n^2 / N)            (Bluestein only)
    const cplx* bf;     // [M]   FFT_M(conj-chirp filter) / M  (Bluestein only)
    const cplx* root;   // [N]   exp(-2 pi i k / N)
};

// forward length-N DFT of nb sequences already resident in LDS (entries n >= N must be zero when blue).
// On return entries [0, N) of each sequence hold the DFT.
__device__ __forceinline__ void lds_dft(cplx* s, const AxisDev& ax, int nb, int stride)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    if (!ax.blue) { lds_fft(s, ax.M, ax.logM, nb, stride, ax.tw); return; }
    const int M = ax.M;
    for (int e = tid; e < nb * M; e += nt) {           // a[n] = x[n] * chirp[n]
        const int f = e / M, n = e - f * M;
        if (n < ax.N) s[f * stride + n] = cmul(s[f * stride + n], ax.chirp[n]);
    }
    __syncthreads();
    lds_fft(s, M, ax.logM, nb, stride, ax.tw);
    for (int e = tid; e < nb * M; e += nt) {           // conj(A * Bf): second forward FFT then acts as inverse
        const int f = e / M, k = e - f * M;
        s[f * stride + k] = cconj(cmul(s[f * stride + k], ax.bf[k]));
    }
    __syncthreads();
    lds_fft(s, M, ax.logM, nb, stride, ax.tw);
    for (int e = tid; e < nb * M; e += nt) {
        const int f = e / M, k = e - f * M;
        if (k < ax.N) s[f * stride + k] = cmul(ax.chirp[k], cconj(s[f * stride + k]));
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// forward pass 1: rows, real -> half complex, two image rows per complex transform, SpatialPoly fused
// ------------------------------------------------------------------------------------------------
#define SFFT_MAX_PLANES 12
struct RowsArgs {                           // plane k = src[k] * wx[k][row] * wy[k][col]   (null weight = 1)
    const double* src[SFFT_MAX_PLANES];
    const double* wx[SFFT_MAX_PLANES];      // [N0] factor of the spatial basis along axis 0 (cx^i or a B-spline basis function)
    const double* wy[SFFT_MAX_PLANES];      // [N1] factor along axis 1
};

__global__ void __launch_bounds__(1024) rows_r2c(RowsArgs a, cplx* __restrict__ out, int N0, int N1, int Nh, int Nhp,
                                                  AxisDev ax, double scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* s = reinterpret_cast<cplx*>(smem_raw);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int plane = blockIdx.y;
    const int l0 = 2 * blockIdx.x, l1 = l0 + 1;
    const double* __restrict__ src = a.src[plane];
    const double* __restrict__ wx = a.wx[plane];
    const double* __restrict__ wy = a.wy[plane];
    const bool has1 = l1 < N0;
    const double cx0 = wx ? wx[l0] : 1.0;
    const double cx1 = (wx && has1) ? wx[l1] : 1.0;
    for (int n = tid; n < ax.M; n += nt) {
        cplx z = make_double2(0.0, 0.0);
        if (n < N1) {
            const double cyp = wy ? wy[n] : 1.0;
            z.x = src[(size_t)l0 * N1 + n] * (cx0 * cyp);
            if (has1) z.y = src[(size_t)l1 * N1 + n] * (cx1 * cyp);
        }
        s[n] = z;
    }
    __syncthreads();
    lds_dft(s, ax, 1, ax.M);
    cplx* o0 = out + ((size_t)plane * N0 + l0) * Nhp;
    cplx* o1 = out + ((size_t)plane * N0 + l1) * Nhp;
    for (int m = tid; m < Nh; m += nt) {
        const cplx z = s[m];
        const cplx zc = cconj(s[m == 0 ? 0 : N1 - m]);
        o0[m] = make_double2(0.5 * scale * (z.x + zc.x), 0.5 * scale * (z.y + zc.y));
        if (has1) {
            const double dx = z.x - zc.x, dy = z.y - zc.y;   // (Z - Zc) / (2i) = (dy, -dx)/2
            o1[m] = make_double2(0.5 * scale * dy, -0.5 * scale * dx);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pass 2: columns, complex -> complex in place, TC adjacent columns per workgroup
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) cols_c2c(cplx* __restrict__ data, int N0, int ncols, int Nhp, int TC, int MS,
                                                  AxisDev ax, int inverse, double scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* s = reinterpret_cast<cplx*>(smem_raw);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int c0 = blockIdx.x * TC;
    cplx* __restrict__ base = data + (size_t)blockIdx.y * N0 * Nhp;
    for (int e = tid; e < TC * ax.M; e += nt) {
        const int l = e / TC, c = e - l * TC;
        cplx z = make_double2(0.0, 0.0);
        if (l < N0 && c0 + c < ncols) {
            z = base[(size_t)l * Nhp + c0 + c];
            if (inverse) z.y = -z.y;
        }
        s[c * MS + l] = z;
    }
    __syncthreads();
    lds_dft(s, ax, TC, MS);
    for (int e = tid; e < TC * N0; e += nt) {
        const int l = e / TC, c = e - l * TC;
        if (c0 + c < ncols) {
            cplx z = s[c * MS + l];
            if (inverse) z.y = -z.y;
            base[(size_t)l * Nhp + c0 + c] = make_double2(z.x * scale, z.y * scale);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// inverse pass 2: rows, half complex -> real, two rows per transform, DIFF epilogue fused:
//   DIFF = J - sum_pq b_pq cx^p cy^q - conv          (SFFTSubtract.py:452-461 with the J and T terms kept in real space)
// ------------------------------------------------------------------------------------------------
#define SFFT_MAX_PQ 64
#define SFFT_MAX_BQ 16
// differential background B(row, col) = sum_t b[t] * tbx[p[t]][row] * tby[q[t]][col]  (tables of the 1-D basis factors)
struct BkgArgs {
    int npq, nq;                    // terms, distinct column factors
    const double* tbx;              // [nbx][N0]
    const double* tby;              // [nby][N1]
    int p[SFFT_MAX_PQ], q[SFFT_MAX_PQ];
};

// per-row coefficients of the column factors: c[q] = sum_{t: q[t] = q} b[t] * tbx[p[t]][row]
// (NQ = compile-time bound on the number of column factors: 4 covers polynomial backgrounds, 16 the general case)
template <int NQ>
__device__ __forceinline__ void bkg_row_coeffs(const BkgArgs& bk, const double* __restrict__ bpq, int row, int N0, double (&c)[NQ])
{
#pragma unroll
    for (int q = 0; q < NQ; ++q) c[q] = 0.0;
    for (int t = 0; t < bk.npq; ++t) {
        const double v = bpq[t] * bk.tbx[(size_t)bk.p[t] * N0 + row];
#pragma unroll
        for (int q = 0; q < NQ; ++q) c[q] += (bk.q[t] == q) ? v : 0.0;
    }
}
template <int NQ>
__device__ __forceinline__ double bkg_eval(const BkgArgs& bk, const double (&c)[NQ], int col, int N1)
{
    double B = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const double t = bk.tby[(size_t)min(q, bk.nq - 1) * N1 + col];      // clamped: always a valid, branch-free load
        B = fma((q < bk.nq) ? c[q] : 0.0, t, B);
    }
    return B;
}

__global__ void __launch_bounds__(1024) rows_c2r_diff(const cplx* __restrict__ FD, const double* __restrict__ J,
                                                       const double* __restrict__ bpq, BkgArgs bk, double* __restrict__ DIFF,
                                                       int N0, int N1, int Nh, int Nhp, AxisDev ax)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* s = reinterpret_cast<cplx*>(smem_raw);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int l0 = 2 * blockIdx.x, l1 = l0 + 1;
    const bool has1 = l1 < N0;
    const cplx* f0 = FD + (size_t)l0 * Nhp;
    const cplx* f1 = FD + (size_t)(has1 ? l1 : l0) * Nhp;
    const bool even = (N1 & 1) == 0;
    for (int m = tid; m < ax.M; m += nt) {
        cplx z = make_double2(0.0, 0.0);
        if (m < N1) {
            const bool mir = m >= Nh;
            const int mm = mir ? N1 - m : m;
            cplx x0 = f0[mm];
            cplx x1 = has1 ? f1[mm] : make_double2(0.0, 0.0);
            if (mm == 0 || (even && mm == N1 / 2)) { x0.y = 0.0; x1.y = 0.0; }
            if (mir) { x0.y = -x0.y; x1.y = -x1.y; }
            // Z = X0 + i X1, conjugated on input so that the forward transform acts as the inverse
            z = make_double2(x0.x - x1.y, -(x0.y + x1.x));
        }
        s[m] = z;
    }
    __syncthreads();
    lds_dft(s, ax, 1, ax.M);
    double c0[SFFT_MAX_BQ], c1[SFFT_MAX_BQ];
    bkg_row_coeffs<SFFT_MAX_BQ>(bk, bpq, l0, N0, c0);
    bkg_row_coeffs<SFFT_MAX_BQ>(bk, bpq, has1 ? l1 : l0, N0, c1);
    for (int n = tid; n < N1; n += nt) {
        const cplx z = s[n];                 // conj(result): row0 = z.x, row1 = -z.y
        DIFF[(size_t)l0 * N1 + n] = J[(size_t)l0 * N1 + n] - bkg_eval<SFFT_MAX_BQ>(bk, c0, n, N1) - z.x;
        if (has1) DIFF[(size_t)l1 * N1 + n] = J[(size_t)l1 * N1 + n] - bkg_eval<SFFT_MAX_BQ>(bk, c1, n, N1) + z.y;
    }
}

// ================================================================================================
// Axes too long for one on-chip transform (e.g. 6144, 9216, 9232): four-step decomposition N = A * B,
//     X[ka + A kb] = sum_nb W_N^(nb ka) [ sum_na x[B na + nb] W_A^(na ka) ] W_B^(nb kb),
// as two passes of batched strided sub-transforms (lengths A and B, each power of two or Bluestein on chip)
// through global memory.  Correctness path for the large BASELINE configs; not tuned.
// ================================================================================================
struct PassDesc {
    int len, J, nlines, mode;              // mode: which index runs fastest over threads (0: element, 1: j, 2: line)
    long long js_in, es_in, lst_in;        // strides in complex elements: sequence j, element e, line
    long long js_out, es_out, lst_out;
    int twiddle, N;                        // multiply output k of sequence j by rootN[(j k) mod N]
    int conj_in, conj_out;
    double scale;
};

__global__ void __launch_bounds__(1024) strided_dft(const cplx* __restrict__ in, cplx* __restrict__ out, PassDesc d, AxisDev ax,
                                                     const cplx* __restrict__ rootN, int TC, int MS)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* s = reinterpret_cast<cplx*>(smem_raw);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int M = ax.M;
    for (int x = tid; x < TC * M; x += nt) {
        int c, e;
        if (d.mode == 0) { c = x / M; e = x - c * M; } else { e = x / TC; c = x - e * TC; }
        const int j = (d.mode == 2) ? (int)blockIdx.y : (int)blockIdx.x * TC + c;
        const int line = (d.mode == 2) ? (int)blockIdx.x * TC + c : (int)blockIdx.y;
        cplx z = make_double2(0.0, 0.0);
        if (e < d.len && j < d.J && line < d.nlines) {
            z = in[(long long)line * d.lst_in + (long long)j * d.js_in + (long long)e * d.es_in];
            if (d.conj_in) z.y = -z.y;
        }
        s[c * MS + e] = z;
    }
    __syncthreads();
    lds_dft(s, ax, TC, MS);
    for (int x = tid; x < TC * M; x += nt) {
        int c, e;
        if (d.mode == 0) { c = x / M; e = x - c * M; } else { e = x / TC; c = x - e * TC; }
        const int j = (d.mode == 2) ? (int)blockIdx.y : (int)blockIdx.x * TC + c;
        const int line = (d.mode == 2) ? (int)blockIdx.x * TC + c : (int)blockIdx.y;
        if (e < d.len && j < d.J && line < d.nlines) {
            cplx z = s[c * MS + e];
            if (d.twiddle) z = cmul(z, rootN[(int)(((long long)j * e) % d.N)]);
            if (d.conj_out) z.y = -z.y;
            out[(long long)line * d.lst_out + (long long)j * d.js_out + (long long)e * d.es_out] = make_double2(z.x * d.scale, z.y * d.scale);
        }
    }
}

// Z[pair][n] = (I[2 pair][n] w, I[2 pair + 1][n] w'): two real rows per complex sequence, SpatialPoly fused
__global__ void __launch_bounds__(256) pack_rows(const double* __restrict__ src, const double* __restrict__ wx,
                                                 const double* __restrict__ wy, cplx* __restrict__ Z, int N0, int N1)
{
    const int n = blockIdx.x * 256 + threadIdx.x, pr = blockIdx.y;
    if (n >= N1) return;
    const int l0 = 2 * pr, l1 = l0 + 1;
    const double cyp = wy ? wy[n] : 1.0;
    const double v0 = src[(size_t)l0 * N1 + n] * ((wx ? wx[l0] : 1.0) * cyp);
    const double v1 = (l1 < N0) ? src[(size_t)l1 * N1 + n] * ((wx ? wx[l1] : 1.0) * cyp) : 0.0;
    Z[(size_t)pr * N1 + n] = make_double2(v0, v1);
}

// half spectra of the two real rows from the transform of their packed sequence (same algebra as rows_r2c)
__global__ void __launch_bounds__(256) untangle_rows(const cplx* __restrict__ Zf, cplx* __restrict__ out, int N0, int N1, int Nh, int Nhp,
                                                     double scale)
{
    const int m = blockIdx.x * 256 + threadIdx.x, pr = blockIdx.y;
    if (m >= Nh) return;
    const int l0 = 2 * pr, l1 = l0 + 1;
    const cplx z = Zf[(size_t)pr * N1 + m];
    const cplx zp = Zf[(size_t)pr * N1 + (m == 0 ? 0 : N1 - m)];
    const cplx zc = make_double2(zp.x, -zp.y);
    const double hs = 0.5 * scale;
    out[(size_t)l0 * Nhp + m] = make_double2(hs * (z.x + zc.x), hs * (z.y + zc.y));
    if (l1 < N0) out[(size_t)l1 * Nhp + m] = make_double2(hs * (z.y - zc.y), -hs * (z.x - zc.x));
}

// conj(X0 + i X1) on the full length from the half spectra of two rows (input of the inverse row transform)
__global__ void __launch_bounds__(256) retangle_rows(const cplx* __restrict__ FD, cplx* __restrict__ Z, int N0, int N1, int Nh, int Nhp)
{
    const int m = blockIdx.x * 256 + threadIdx.x, pr = blockIdx.y;
    if (m >= N1) return;
    const int l0 = 2 * pr, l1 = l0 + 1;
    const bool has1 = l1 < N0, even = (N1 & 1) == 0;
    const bool mir = m >= Nh;
    const int mm = mir ? N1 - m : m;
    cplx x0 = FD[(size_t)l0 * Nhp + mm];
    cplx x1 = has1 ? FD[(size_t)l1 * Nhp + mm] : make_double2(0.0, 0.0);
    if (mm == 0 || (even && mm == N1 / 2)) { x0.y = 0.0; x1.y = 0.0; }
    if (mir) { x0.y = -x0.y; x1.y = -x1.y; }
    Z[(size_t)pr * N1 + m] = make_double2(x0.x - x1.y, -(x0.y + x1.x));
}

// DIFF = J - sum_pq b_pq cx^p cy^q - conv from the transformed packed rows (see rows_c2r_diff)
__global__ void __launch_bounds__(256) finish_diff(const cplx* __restrict__ Zf, const double* __restrict__ J, const double* __restrict__ bpq,
                                                   BkgArgs bk, double* __restrict__ DIFF, int N0, int N1)
{
    const int n = blockIdx.x * 256 + threadIdx.x, pr = blockIdx.y;
    if (n >= N1) return;
    const int l0 = 2 * pr, l1 = l0 + 1;
    const cplx z = Zf[(size_t)pr * N1 + n];
    double c0[SFFT_MAX_BQ], c1[SFFT_MAX_BQ];
    bkg_row_coeffs<SFFT_MAX_BQ>(bk, bpq, l0, N0, c0);
    bkg_row_coeffs<SFFT_MAX_BQ>(bk, bpq, (l1 < N0) ? l1 : l0, N0, c1);
    DIFF[(size_t)l0 * N1 + n] = J[(size_t)l0 * N1 + n] - bkg_eval<SFFT_MAX_BQ>(bk, c0, n, N1) - z.x;
    if (l1 < N0) DIFF[(size_t)l1 * N1 + n] = J[(size_t)l1 * N1 + n] - bkg_eval<SFFT_MAX_BQ>(bk, c1, n, N1) + z.y;
}

// ================================================================================================
// Fast path for 4096-point axes: register-resident radix-16 FFT.  256 threads own 16 points each through
// three radix-16 stages (4096 = 16^3); LDS is used only for the two inter-stage exchanges (padded by one
// element per 16 so that the stride-16 writes of stage 1 are conflict free), not as the working array.
// ================================================================================================
#define R16_OUT(s) (4 * ((s) & 3) + ((s) >> 2))      // register holding output s of dft16()
#define F4K_LDS 4352                                 // 4096 + 4096/16 complex per transform

__device__ __forceinline__ void dft4(cplx& a, cplx& b, cplx& c, cplx& d)
{
    const cplx s02 = cadd(a, c), d02 = csub(a, c), s13 = cadd(b, d), d13 = csub(b, d);
    a = cadd(s02, s13);
    c = csub(s02, s13);
    b = make_double2(d02.x + d13.y, d02.y - d13.x);   // d02 - i d13
    d = make_double2(d02.x - d13.y, d02.y + d13.x);   // d02 + i d13
}

// forward 16-point DFT in registers; output s ends in u[R16_OUT(s)]
__device__ __forceinline__ void dft16(cplx (&u)[16])
{
    const double c1 = 0.92387953251128673848, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
#pragma unroll
    for (int b = 0; b < 4; ++b) dft4(u[b], u[4 + b], u[8 + b], u[12 + b]);
    // u[4c + b] *= W16^(b c)
    u[5] = cmul(u[5], make_double2(c1, -s1));          // W^1
    u[6] = cmul(u[6], make_double2(h, -h));            // W^2
    u[7] = cmul(u[7], make_double2(s1, -c1));          // W^3
    u[9] = cmul(u[9], make_double2(h, -h));            // W^2
    u[10] = make_double2(u[10].y, -u[10].x);           // W^4 = -i
    u[11] = cmul(u[11], make_double2(-h, -h));         // W^6
    u[13] = cmul(u[13], make_double2(s1, -c1));        // W^3
    u[14] = cmul(u[14], make_double2(-h, -h));         // W^6
    u[15] = cmul(u[15], make_double2(-c1, s1));        // W^9
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4(u[4 * c], u[4 * c + 1], u[4 * c + 2], u[4 * c + 3]);
}

// u[r] *= tw[r q], r = 1..15, from four table entries (products of at most three factors)
__device__ __forceinline__ void twiddle16(cplx (&u)[16], const cplx* __restrict__ tw, int q)
{
    const cplx w1 = tw[q], w2 = tw[2 * q], w4 = tw[4 * q], w8 = tw[8 * q];
    const cplx w3 = cmul(w1, w2), w5 = cmul(w4, w1), w6 = cmul(w4, w2);
    const cplx w7 = cmul(w4, w3);
    u[1] = cmul(u[1], w1); u[2] = cmul(u[2], w2); u[3] = cmul(u[3], w3); u[4] = cmul(u[4], w4);
    u[5] = cmul(u[5], w5); u[6] = cmul(u[6], w6); u[7] = cmul(u[7], w7); u[8] = cmul(u[8], w8);
    u[9] = cmul(u[9], cmul(w8, w1)); u[10] = cmul(u[10], cmul(w8, w2)); u[11] = cmul(u[11], cmul(w8, w3));
    u[12] = cmul(u[12], cmul(w8, w4)); u[13] = cmul(u[13], cmul(w8, w5)); u[14] = cmul(u[14], cmul(w8, w6));
    u[15] = cmul(u[15], cmul(w8, w7));
}

__device__ __forceinline__ int pad16(int i) { return i + (i >> 4); }

// 4096-point forward FFT.  In: u[r] = x[j + 256 r].  Out: u[R16_OUT(s)] = X[j + 256 s].  j in [0, 256).
// `lds` = this transform's F4K_LDS-element scratch.  Every thread of the block must call (barriers inside).
__device__ __forceinline__ void fft4096_core(cplx (&u)[16], int j, cplx* lds, const cplx* __restrict__ tw)
{
    dft16(u);
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) lds[17 * j + sx] = u[R16_OUT(sx)];            // pad16(16 j + s)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r] = lds[pad16(j + 256 * r)];
    __syncthreads();
    const int k = j & 15;
    twiddle16(u, tw, 16 * k);
    dft16(u);
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) lds[pad16((j - k) * 16 + k + 16 * sx)] = u[R16_OUT(sx)];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) u[r] = lds[pad16(j + 256 * r)];
    twiddle16(u, tw, j);
    dft16(u);
}

// rows, real -> half complex (N1 = 4096), two image rows per transform, spatial factors fused.  Planes [first, first +
// count) of a launch group share their source image: the workgroup reads its two rows once and produces every plane.
struct RowGroups { int ngroups; int first[SFFT_MAX_PLANES]; int count[SFFT_MAX_PLANES]; };

__global__ void __launch_bounds__(256) rows_r2c_4096(RowsArgs a, RowGroups grp, cplx* __restrict__ out, int N0, int Nhp,
                                                     const cplx* __restrict__ tw, double scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    const int N1 = 4096;
    const int j = threadIdx.x;
    const int pfirst = grp.first[blockIdx.y], pcount = grp.count[blockIdx.y];
    const int l0 = 2 * blockIdx.x, l1 = l0 + 1;
    const bool has1 = l1 < N0;
    const double* __restrict__ src = a.src[pfirst];
    const double* r0p = src + (size_t)l0 * N1;
    const double* r1p = src + (size_t)(has1 ? l1 : l0) * N1;
    double x0[16], x1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = j + 256 * r;
        x0[r] = r0p[n];
        x1[r] = has1 ? r1p[n] : 0.0;
    }
    const double hs = 0.5 * scale;
    for (int pp = 0; pp < pcount; ++pp) {
        const int plane = pfirst + pp;
        const double* __restrict__ wx = a.wx[plane];
        const double* __restrict__ wy = a.wy[plane];
        const double cx0 = wx ? wx[l0] : 1.0;
        const double cx1 = (wx && has1) ? wx[l1] : 1.0;
        cplx u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double cyp = wy ? wy[j + 256 * r] : 1.0;
            u[r] = make_double2(x0[r] * (cx0 * cyp), x1[r] * (cx1 * cyp));
        }
        if (pp > 0) __syncthreads();            // the previous plane's partner reads are done
        fft4096_core(u, j, lds, tw);
        __syncthreads();
#pragma unroll
        for (int sx = 0; sx < 16; ++sx) lds[j + 256 * sx] = u[R16_OUT(sx)];
        __syncthreads();
        cplx* o0 = out + ((size_t)plane * N0 + l0) * Nhp;
        cplx* o1 = out + ((size_t)plane * N0 + l1) * Nhp;
#pragma unroll
        for (int sx = 0; sx <= 8; ++sx) {
            const int m = j + 256 * sx;
            if (sx < 8 || j == 0) {
                const cplx z = u[R16_OUT(sx)];
                const cplx zp = lds[(N1 - m) & (N1 - 1)];
                const cplx zc = make_double2(zp.x, -zp.y);
                o0[m] = make_double2(hs * (z.x + zc.x), hs * (z.y + zc.y));
                if (has1) o1[m] = make_double2(hs * (z.y - zc.y), -hs * (z.x - zc.x));
            }
        }
    }
}

// columns, complex -> complex in place (N0 = 4096), two adjacent columns per workgroup (512 threads).
// Blocks that share 128-byte lines are mapped to the same XCD (block b runs on XCD b % 8) so its L2 merges them.
__global__ void __launch_bounds__(512) cols_c2c_4096(cplx* __restrict__ data, int ncols, int Nhp, const cplx* __restrict__ tw,
                                                     int inverse, double scale, int pairs_per_xcd)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    const int N0 = 4096;
    const int c = threadIdx.x & 1, j = threadIdx.x >> 1;
    const int cp = (blockIdx.x & 7) * pairs_per_xcd + (blockIdx.x >> 3);
    const int col = 2 * cp + c;
    const bool ok = (blockIdx.x >> 3) < pairs_per_xcd && col < ncols;
    cplx* __restrict__ base = data + (size_t)blockIdx.y * N0 * Nhp + (ok ? col : 0);
    cplx u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        cplx z = ok ? base[(size_t)(j + 256 * r) * Nhp] : make_double2(0.0, 0.0);
        if (inverse) z.y = -z.y;
        u[r] = z;
    }
    fft4096_core(u, j, lds + c * (F4K_LDS + 4), tw);     // +4: the two columns' regions sit half a bank row apart
    if (!ok) return;
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) {
        cplx z = u[R16_OUT(sx)];
        if (inverse) z.y = -z.y;
        base[(size_t)(j + 256 * sx) * Nhp] = make_double2(z.x * scale, z.y * scale);
    }
}

// rows, half complex -> real (N1 = 4096), two rows per transform, DIFF epilogue (see rows_c2r_diff)
template <int NQ>
__global__ void __launch_bounds__(256) rows_c2r_diff_4096(const cplx* __restrict__ FD, const double* __restrict__ J,
                                                          const double* __restrict__ bpq, BkgArgs bk, double* __restrict__ DIFF,
                                                          int N0, int Nhp, const cplx* __restrict__ tw)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* lds = reinterpret_cast<cplx*>(smem_raw);
    const int N1 = 4096;
    const int j = threadIdx.x;
    const int l0 = 2 * blockIdx.x, l1 = l0 + 1;
    const bool has1 = l1 < N0;
    const cplx* f0 = FD + (size_t)l0 * Nhp;
    const cplx* f1 = FD + (size_t)(has1 ? l1 : l0) * Nhp;
    cplx u[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = j + 256 * r;
        const bool mir = m > N1 / 2;
        const int mm = mir ? N1 - m : m;
        cplx x0 = f0[mm];
        cplx x1 = has1 ? f1[mm] : make_double2(0.0, 0.0);
        if (mm == 0 || mm == N1 / 2) { x0.y = 0.0; x1.y = 0.0; }
        if (mir) { x0.y = -x0.y; x1.y = -x1.y; }
        u[r] = make_double2(x0.x - x1.y, -(x0.y + x1.x));      // conj(X0 + i X1)
    }
    fft4096_core(u, j, lds, tw);
    double c0[NQ], c1[NQ];
    bkg_row_coeffs<NQ>(bk, bpq, l0, N0, c0);
    bkg_row_coeffs<NQ>(bk, bpq, has1 ? l1 : l0, N0, c1);
    const double* j0 = J + (size_t)l0 * N1;
    const double* j1 = J + (size_t)(has1 ? l1 : l0) * N1;
    double* d0 = DIFF + (size_t)l0 * N1;
    double* d1 = DIFF + (size_t)(has1 ? l1 : l0) * N1;
#pragma unroll
    for (int sx = 0; sx < 16; ++sx) {
        const int n = j + 256 * sx;
        const cplx z = u[R16_OUT(sx)];
        d0[n] = j0[n] - bkg_eval<NQ>(bk, c0, n, N1) - z.x;
        if (has1) d1[n] = j1[n] - bkg_eval<NQ>(bk, c1, n, N1) + z.y;
    }
}

// ------------------------------------------------------------------------------------------------
// Greek stage 1: for every listed pass (A, B) and column m of the half spectrum
//      G[r][m] = sum_l A[l][m] * conj(B[l][m]) * W0^(l r),   |r| <= h          (pruned column DFT of the
// Hadamard product; only these lags are ever read by FillLS_*, SFFTConfigure.py:251-269, 364-371, 621-628).
// r and -r share their four real products.  B is a stored plane (Omega, Theta) or, for Gamma, the column
// factor Xp[l] = DFT(cx^p)[l] of the rank-1 spectrum FT_pq = SCALE * Xp (x) Yq -- the row factor Yq[m] does not
// depend on l and is applied in stage 2, so one pass serves every q.
// One wave per 64 columns; RS waves of a workgroup split the lags; rows are loaded U at a time ahead of use.
// ------------------------------------------------------------------------------------------------
struct G1Pass {
    int a_plane;      // plane index into spec
    int b_plane;      // plane index, or -1: B[l][m] = Xp[bp][l]
    int bp;
    int h;            // lag half width
    long long gp_off; // offset (cplx) of this pass's [S][2h+1][Nhp] partial buffer
};

struct PatchJob {
    int pass;         // G1 pass that produced G
    int yq;           // -1, or q: G[r][m] is multiplied by conj(tscale * Yq[q][m])
    int h;
    int patch_off;    // offset (doubles) of this job's [(2h+1)][(2h+1)] patch
    double scale;
};

template <int HBW, int RS>
__global__ void __launch_bounds__(64 * RS) greek_g1(const cplx* __restrict__ spec, const G1Pass* __restrict__ passes, int pass0,
                                                    cplx* __restrict__ Gp, int N0, int Nh, int Nhp, int rows_per_chunk,
                                                    int r_base, const cplx* __restrict__ W0tab, int HM, const cplx* __restrict__ Xp)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x * 64 + lane;
    const int chunk = blockIdx.y;
    const G1Pass pr = passes[pass0 + blockIdx.z];
    const int h = pr.h;
    const int PH = 2 * h + 1;
    const int lb = chunk * rows_per_chunk;
    const int le = min(N0, lb + rows_per_chunk);
    const bool active = m < Nh;
    const int mc = active ? m : 0;
    const size_t plane_sz = (size_t)N0 * Nhp;
    const cplx* __restrict__ A = spec + (size_t)pr.a_plane * plane_sz + mc;
    const bool colfac = pr.b_plane < 0;
    const cplx* __restrict__ B = colfac ? A : spec + (size_t)pr.b_plane * plane_sz + mc;
    const cplx* __restrict__ xp = Xp + (size_t)pr.bp * N0;
    const int rfirst = r_base + 1 + wv * HBW;       // this wave's lags: rfirst .. rfirst + HBW - 1 (wave-uniform)
    int nact = h - (rfirst - 1);
    if (nact > HBW) nact = HBW;
    if (nact < 0) nact = 0;
    double S1[HBW], S2[HBW], S3[HBW], S4[HBW];
#pragma unroll
    for (int t = 0; t < HBW; ++t) { S1[t] = S2[t] = S3[t] = S4[t] = 0.0; }
    double g0x = 0.0, g0y = 0.0;
    const bool do_g0 = (r_base == 0 && wv == 0);
    // W0tab[l][r] = W0^(l r), r = 0..HM-1: one contiguous, wave-uniform row of twiddles per image row (scalar loads)
    const cplx* __restrict__ trow = W0tab + (size_t)lb * HM + rfirst;
    for (int l = lb; l < le; ++l, trow += HM) {
        const cplx av = A[(size_t)l * Nhp];
        const cplx bv = colfac ? xp[l] : B[(size_t)l * Nhp];
        const cplx H = cmulc(av, bv);
        if (do_g0) { g0x += H.x; g0y += H.y; }
#pragma unroll
        for (int t = 0; t < HBW; ++t) {
            if (t < nact) {
                const cplx w = trow[t];
                S1[t] = fma(H.x, w.x, S1[t]);
                S2[t] = fma(H.y, w.y, S2[t]);
                S3[t] = fma(H.x, w.y, S3[t]);
                S4[t] = fma(H.y, w.x, S4[t]);
            }
        }
    }
    if (!active) return;
    cplx* g = Gp + pr.gp_off + (size_t)chunk * PH * Nhp + m;
    if (do_g0) g[(size_t)h * Nhp] = make_double2(g0x, g0y);
#pragma unroll
    for (int t = 0; t < HBW; ++t) {
        if (t < nact) {
            const int r = rfirst + t;
            g[(size_t)(h + r) * Nhp] = make_double2(S1[t] - S2[t], S3[t] + S4[t]);
            g[(size_t)(h - r) * Nhp] = make_double2(S1[t] + S2[t], S4[t] - S3[t]);
        }
    }
}

// W0tab[l][r] = root0[(l r) mod N0], r = 0 .. HM-1 (column 0 is the constant 1: lag 0)
__global__ void __launch_bounds__(256) build_w0tab(const cplx* __restrict__ root0, cplx* __restrict__ W0tab, int N0, int HM)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= N0 * HM) return;
    const int l = e / HM, r = e - l * HM;
    W0tab[e] = root0[(int)(((long long)l * r) % N0)];
}

// Gamma passes with p = 0: Xp = N0 * delta[l], so G[r][m] = N0 * A[0][m] for every lag (chunk 0; other chunks zero)
__global__ void __launch_bounds__(256) greek_g1_row0(const cplx* __restrict__ spec, const G1Pass* __restrict__ passes, int pass0,
                                                     cplx* __restrict__ Gp, int N0, int Nh, int Nhp, int S)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= Nh) return;
    const G1Pass pr = passes[pass0 + blockIdx.y];
    const int PH = 2 * pr.h + 1;
    const cplx a0 = spec[(size_t)pr.a_plane * N0 * Nhp + m];
    const cplx v = make_double2(a0.x * (double)N0, a0.y * (double)N0);
    cplx* g = Gp + pr.gp_off + m;
    for (int c = 0; c < S; ++c)
        for (int r = 0; r < PH; ++r) g[((size_t)c * PH + r) * Nhp] = (c == 0) ? v : make_double2(0.0, 0.0);
}

// Greek stage 2: patch[r][e] = scale * sum_{m < Nh} wgt[m] * Re( W1^(m e) * y[m] * sum_chunks G[r][m] ),  |e| <= h.
// wgt = 1 for the self-conjugate columns (m = 0, and m = N1/2 when N1 is even), 2 otherwise;
// y[m] = conj(tscale * Yq[q][m]) for Gamma jobs, 1 otherwise.
__global__ void __launch_bounds__(256) greek_g2(const cplx* __restrict__ Gp, const G1Pass* __restrict__ passes,
                                                const PatchJob* __restrict__ jobs, int job0,
                                                double* __restrict__ patches, int Nh, int Nhp, int N1, int S,
                                                const cplx* __restrict__ root1, const cplx* __restrict__ Yq, double tscale)
{
    const PatchJob jb = jobs[job0 + blockIdx.y];
    const int h = jb.h, PH = 2 * h + 1;
    const int r = blockIdx.x;
    if (r >= PH) return;
    const int tid = threadIdx.x;
    __shared__ double red[2][4][17];
    const cplx* g = Gp + passes[jb.pass].gp_off + (size_t)r * Nhp;
    const cplx* yq = jb.yq >= 0 ? Yq + (size_t)jb.yq * Nhp : nullptr;
    const bool even = (N1 & 1) == 0;
    double* out = patches + jb.patch_off + (size_t)r * PH + h;
    for (int e0 = 0; e0 == 0 || e0 < h; e0 += 16) {
        double U[17], V[17];
#pragma unroll
        for (int t = 0; t < 17; ++t) { U[t] = 0.0; V[t] = 0.0; }
        const int ne = min(16, h - e0);   // lags e0+1 .. e0+ne, plus lag 0 when e0 == 0
        for (int m = tid; m < Nh; m += 256) {
            double gx = 0.0, gy = 0.0;
            for (int c = 0; c < S; ++c) {
                const cplx v = g[(size_t)c * PH * Nhp + m];
                gx += v.x; gy += v.y;
            }
            if (yq) {
                const cplx y = yq[m];
                const cplx t = cmulc(make_double2(gx, gy), make_double2(y.x * tscale, y.y * tscale));
                gx = t.x; gy = t.y;
            }
            const double wgt = (m == 0 || (even && m == N1 / 2)) ? 1.0 : 2.0;
            gx *= wgt; gy *= wgt;
            if (e0 == 0) U[0] += gx;
            int idx = (int)(((long long)m * e0) % N1);
#pragma unroll
            for (int t = 1; t <= 16; ++t) {
                if (t <= ne) {
                    idx += m; if (idx >= N1) idx -= N1;
                    const cplx w = root1[idx];
                    U[t] = fma(gx, w.x, U[t]);
                    V[t] = fma(gy, w.y, V[t]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 17; ++t) {
            double u = U[t], v = V[t];
            for (int off = 32; off > 0; off >>= 1) { u += __shfl_down(u, off); v += __shfl_down(v, off); }
            if ((tid & 63) == 0) { red[0][tid >> 6][t] = u; red[1][tid >> 6][t] = v; }
        }
        __syncthreads();
        if (tid < 17) {
            const double u = red[0][0][tid] + red[0][1][tid] + red[0][2][tid] + red[0][3][tid];
            const double v = red[1][0][tid] + red[1][1][tid] + red[1][2][tid] + red[1][3][tid];
            if (tid == 0) { if (e0 == 0) out[0] = jb.scale * u; }
            else if (tid <= ne) {
                out[e0 + tid] = jb.scale * (u - v);
                out[-(e0 + tid)] = jb.scale * (u + v);
            }
        }
        __syncthreads();
    }
}

// Delta: rowmom[l][q] = sum_n J[l][n] tby[q][n], then delta[t] = SCALE * sum_l tbx[p[t]][l] rowmom[l][q[t]]
// (= PreDEL[pq][0][0], SFFTSubtract.py:706-729, evaluated in real space: only element [0][0] is ever read).
template <int NQ>
__global__ void __launch_bounds__(256) row_moments(const double* __restrict__ J, double* __restrict__ rowmom, int N0, int N1,
                                                   const double* __restrict__ tby, int nq)
{
    const int l = blockIdx.x, tid = threadIdx.x;
    double acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
    for (int n = tid; n < N1; n += 256) {
        const double v = J[(size_t)l * N1 + n];
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = fma(v, tby[(size_t)min(q, nq - 1) * N1 + n], acc[q]);   // q >= nq: unused copies
    }
    __shared__ double red[4][NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double u = acc[q];
        for (int off = 32; off > 0; off >>= 1) u += __shfl_down(u, off);
        if ((tid & 63) == 0) red[tid >> 6][q] = u;
    }
    __syncthreads();
    if (tid < NQ) rowmom[(size_t)l * SFFT_MAX_BQ + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

__global__ void __launch_bounds__(256) delta_finish(const double* __restrict__ rowmom, double* __restrict__ delta, int N0,
                                                    BkgArgs bk, double scale)
{
    const int pq = blockIdx.x, tid = threadIdx.x;
    const int pi = bk.p[pq], q = bk.q[pq];
    double acc = 0.0;
    for (int l = tid; l < N0; l += 256) acc = fma(bk.tbx[(size_t)pi * N0 + l], rowmom[(size_t)l * SFFT_MAX_BQ + q], acc);
    __shared__ double red[4];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) delta[pq] = scale * (red[0] + red[1] + red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------
// FillLS_{OMG,GAM,PSI,PHI,THE,DEL} + Remove_LSFStripes, one thread per matrix element
// (SFFTConfigure.py:957-1293).  PSI is filled from GAM through Psi[p'q',ij](-rho) == Gam[ij,p'q'](rho), and
// Omega pairs with i'j' > ij from Omega[ij,i'j'](-rho); both identities are exact.
// out is [(n+1)][ld]: rows/cols < n hold LHMAT (after the optional index map), row n and column n hold RHb.
// ------------------------------------------------------------------------------------------------
struct FillArgs {
    int Fij, Fpq, Fab, Fijab, L1, w0, w1;
    int h_omg;            // 2w
    int h_gam;            // w
    int omg_off;          // patches offset of Omega pair 0; pair (i'<=i) index = i'*Fij - i'(i'-1)/2 + (i - i')
    int gam_off;          // Gam pair (ij, pq) at gam_off + (ij*Fpq+pq)*PHg*PHg
    int the_off;          // Theta pair ij at the_off + ij*PHg*PHg
    // tied scaling (B-spline constant photometric ratio, BSplineSFFT.py:2201-2272): the unknowns tie_first + k*tie_stride,
    // k < tie_cnt, are one unknown; its row / column is the SUM of theirs.  tie_cnt = 0: no tie.
    int tie_first, tie_cnt, tie_stride;
};

__device__ __forceinline__ double omg_at(const double* P, const FillArgs& f, int i8, int ij, int r0, int r1)
{
    const int PH = 2 * f.h_omg + 1;
    int lo = i8, hi = ij;
    if (i8 > ij) { lo = ij; hi = i8; r0 = -r0; r1 = -r1; }
    const int pidx = lo * f.Fij - (lo * (lo - 1)) / 2 + (hi - lo);
    return P[f.omg_off + (size_t)pidx * PH * PH + (size_t)(r0 + f.h_omg) * PH + (r1 + f.h_omg)];
}

__device__ double sys_element(const double* P, const double* phi, const double* delta, const FillArgs& f, int R, int C, int NEQ)
{
    const int PHg = 2 * f.h_gam + 1;
    if (C == NEQ) {   // right hand side
        if (R < f.Fijab) {
            const int i8 = R / f.Fab, ab8 = R - i8 * f.Fab;
            const int a8 = ab8 / f.L1 - f.w0, b8 = ab8 % f.L1 - f.w1;
            const double* T = P + f.the_off + (size_t)i8 * PHg * PHg;
            const double t0 = T[(size_t)f.h_gam * PHg + f.h_gam];
            if (a8 == 0 && b8 == 0) return t0;
            return T[(size_t)(a8 + f.h_gam) * PHg + (b8 + f.h_gam)] - t0;
        }
        return delta[R - f.Fijab];
    }
    if (R < f.Fijab && C < f.Fijab) {
        const int i8 = R / f.Fab, ab8 = R - i8 * f.Fab;
        const int ij = C / f.Fab, ab = C - ij * f.Fab;
        const int a8 = ab8 / f.L1 - f.w0, b8 = ab8 % f.L1 - f.w1;
        const int a = ab / f.L1 - f.w0, b = ab % f.L1 - f.w1;
        const bool c8 = (a8 == 0 && b8 == 0), c = (a == 0 && b == 0);
        const double o00 = omg_at(P, f, i8, ij, 0, 0);
        if (c8 && c) return o00;
        if (c8) return omg_at(P, f, i8, ij, -a, -b) - o00;
        if (c) return omg_at(P, f, i8, ij, a8, b8) - o00;
        return -omg_at(P, f, i8, ij, a8, b8) - omg_at(P, f, i8, ij, -a, -b) + omg_at(P, f, i8, ij, a8 - a, b8 - b) + o00;
    }
    if (R < f.Fijab) {          // GAM block
        const int pq = C - f.Fijab;
        const int i8 = R / f.Fab, ab8 = R - i8 * f.Fab;
        const int a8 = ab8 / f.L1 - f.w0, b8 = ab8 % f.L1 - f.w1;
        const double* G = P + f.gam_off + (size_t)(i8 * f.Fpq + pq) * PHg * PHg;
        const double g0 = G[(size_t)f.h_gam * PHg + f.h_gam];
        if (a8 == 0 && b8 == 0) return g0;
        return G[(size_t)(a8 + f.h_gam) * PHg + (b8 + f.h_gam)] - g0;
    }
    if (C < f.Fijab) {          // PSI block = GAM transposed
        const int pq = R - f.Fijab;
        const int ij = C / f.Fab, ab = C - ij * f.Fab;
        const int a = ab / f.L1 - f.w0, b = ab % f.L1 - f.w1;
        const double* G = P + f.gam_off + (size_t)(ij * f.Fpq + pq) * PHg * PHg;
        const double g0 = G[(size_t)f.h_gam * PHg + f.h_gam];
        if (a == 0 && b == 0) return g0;
        return G[(size_t)(a + f.h_gam) * PHg + (b + f.h_gam)] - g0;
    }
    return phi[(R - f.Fijab) * f.Fpq + (C - f.Fijab)];
}

// element of the (possibly tied) system: sum over the members of the row group and of the column group
__device__ double sys_group_element(const double* P, const double* phi, const double* delta, const FillArgs& f, int R, int C, int NEQ)
{
    const int nr = (f.tie_cnt && R == f.tie_first) ? f.tie_cnt : 1;
    const int nc = (f.tie_cnt && C == f.tie_first) ? f.tie_cnt : 1;
    double acc = 0.0;
    for (int a = 0; a < nr; ++a)
        for (int b = 0; b < nc; ++b) acc += sys_element(P, phi, delta, f, R + a * f.tie_stride, C + b * f.tie_stride, NEQ);
    return acc;
}

__global__ void __launch_bounds__(256) fill_system(const double* __restrict__ P, const double* __restrict__ phi,
                                                   const double* __restrict__ delta, FillArgs f, const int* __restrict__ idx,
                                                   int n, int NEQ, double* __restrict__ out, int ld,
                                                   double* __restrict__ rhs_vec)
{
    const int Cp = blockIdx.x * 16 + (threadIdx.x & 15);
    const int Rp = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (Rp > n || Cp > n) return;
    if (Rp == n && Cp == n) { if (out) out[(size_t)n * ld + n] = 0.0; return; }
    if (Rp == n) {   // rhs row (and optional separate vector)
        const int C = idx ? idx[Cp] : Cp;
        const double v = sys_group_element(P, phi, delta, f, C, NEQ, NEQ);
        if (out) out[(size_t)n * ld + Cp] = v;
        if (rhs_vec) rhs_vec[Cp] = v;
        return;
    }
    if (!out) return;
    const int R = idx ? idx[Rp] : Rp;
    if (Cp == n) { out[(size_t)Rp * ld + n] = sys_group_element(P, phi, delta, f, R, NEQ, NEQ); return; }
    const int C = idx ? idx[Cp] : Cp;
    out[(size_t)Rp * ld + Cp] = sys_group_element(P, phi, delta, f, R, C, NEQ);
}

// plain LHMAT export for sfft_get_system (no border)
__global__ void __launch_bounds__(256) fill_plain(const double* __restrict__ P, const double* __restrict__ phi,
                                                  const double* __restrict__ delta, FillArgs f, int NEQ,
                                                  double* __restrict__ LH, double* __restrict__ rhs)
{
    const int C = blockIdx.x * 16 + (threadIdx.x & 15);
    const int R = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (R >= NEQ || C >= NEQ) return;
    if (LH) LH[(size_t)R * NEQ + C] = sys_element(P, phi, delta, f, R, C, NEQ);
    if (rhs && C == 0) rhs[R] = sys_element(P, phi, delta, f, R, NEQ, NEQ);
}

// ------------------------------------------------------------------------------------------------
// Dense solve.  A is the bordered system [(n+1)][ld] (row n = right hand side), SPD in exact arithmetic
// (it is a Gram matrix).  Right-looking blocked Cholesky on the lower triangle; the
// border row rides along so that the forward substitution L y = b is a by-product (y = row n of L).
// ------------------------------------------------------------------------------------------------
#define CB 64
#define BACK_SLICES 64
// The diagonal block is read from Dsrc ([CB][CB], written by the previous step's trailing update) rather than
// from A, because workgroup 0 overwrites A's diagonal block with the factor while the others may still start.
__global__ void __launch_bounds__(256) chol_copy_diag(const double* __restrict__ A, int ld, int nb, double* __restrict__ Dst)
{
    for (int e = threadIdx.x; e < nb * nb; e += 256) {
        const int i = e / nb, j = e - i * nb;
        Dst[i * CB + j] = A[(size_t)i * ld + j];
    }
}

// 1/sqrt(d) to double precision: hardware estimate + two Newton steps (no IEEE division / sqrt sequences on the
// critical path of the factorisation)
__device__ __forceinline__ double rsqrt_nr(double d)
{
    double r = __builtin_amdgcn_rsq(d);
    r = r * fma(-0.5 * d, r * r, 1.5);
    r = r * fma(-0.5 * d, r * r, 1.5);
    return r;
}

// One panel step.  Every workgroup factors the 64x64 diagonal block (right-looking; thread (i, cg) owns elements
// (i, cg + 4q), q < 16, in registers; four columns per pair of barriers; fully unrolled so that all register
// indices are compile-time and only the triangular part is touched), workgroup 0 stores it, workgroups
// b >= 1 then solve X L^T = A_panel for 64 rows below it (border row n included) without barriers (see below).
__global__ void __launch_bounds__(256) chol_panel(double* __restrict__ A, int ld, int n, int k, const double* __restrict__ Dsrc,
                                                  int* __restrict__ status, double* __restrict__ rd)
{
    __shared__ double Dl[CB][CB + 1];     // factor of the diagonal block
    __shared__ double rdiag[CB];          // 1 / L[j][j]
    __shared__ double Rw[CB][4];          // raw column block published in step 1
    __shared__ double Fw[CB][4];          // final column block published in step 3
    const int tid = threadIdx.x;
    const int nb = min(CB, n - k);
    const int i = tid >> 2, cg = tid & 3;
    double a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = cg + 4 * q;
        a[q] = (i < nb && c <= i) ? Dsrc[i * CB + c] : ((i == c) ? 1.0 : 0.0);   // identity padding beyond nb
    }
    // Four columns per step (16 steps, two barriers each).  Step jq eliminates columns 4 jq .. 4 jq + 3:
    //   1. every thread publishes its raw element of that column block (the quad of a row holds the four of them);
    //   2. all threads factor the 4x4 pivot block T redundantly (four reciprocal square roots in sequence);
    //   3. thread (i, cg) forward-substitutes its row through T up to column cg -> final L[i][4 jq + cg], published;
    //   4. rank-4 update of the columns to the right from the published finals.
#pragma unroll
    for (int jq = 0; jq < 16; ++jq) {
        Rw[i][cg] = a[jq];
        __syncthreads();
        const int j0 = 4 * jq;
        const double r00 = Rw[j0][0];
        const double r10 = Rw[j0 + 1][0], r11 = Rw[j0 + 1][1];
        const double r20 = Rw[j0 + 2][0], r21 = Rw[j0 + 2][1], r22 = Rw[j0 + 2][2];
        const double r30 = Rw[j0 + 3][0], r31 = Rw[j0 + 3][1], r32 = Rw[j0 + 3][2], r33 = Rw[j0 + 3][3];
        const double x0 = Rw[i][0], x1 = Rw[i][1], x2 = Rw[i][2], x3 = Rw[i][3];
        const double rs0 = rsqrt_nr(r00);
        const double t10 = r10 * rs0, t20 = r20 * rs0, t30 = r30 * rs0;
        const double d1 = fma(-t10, t10, r11);
        const double rs1 = rsqrt_nr(d1);
        const double t21 = fma(-t20, t10, r21) * rs1, t31 = fma(-t30, t10, r31) * rs1;
        const double d2 = fma(-t21, t21, fma(-t20, t20, r22));
        const double rs2 = rsqrt_nr(d2);
        const double t32 = fma(-t31, t21, fma(-t30, t20, r32)) * rs2;
        const double d3 = fma(-t32, t32, fma(-t31, t31, fma(-t30, t30, r33)));
        const double rs3 = rsqrt_nr(d3);
        if (tid == 0 && blockIdx.x == 0 && j0 < nb && !(r00 > 0.0 && d1 > 0.0 && d2 > 0.0 && d3 > 0.0)) atomicOr(status, 1);
        const double l0 = x0 * rs0;
        const double l1 = fma(-l0, t10, x1) * rs1;
        const double l2 = fma(-l1, t21, fma(-l0, t20, x2)) * rs2;
        const double l3 = fma(-l2, t32, fma(-l1, t31, fma(-l0, t30, x3))) * rs3;
        const double lf = (cg == 0) ? l0 : (cg == 1) ? l1 : (cg == 2) ? l2 : l3;
        a[jq] = lf;                                      // final L[i][4 jq + cg] (entries above the diagonal: unused garbage)
        Fw[i][cg] = lf;
        if (tid < 4) rdiag[j0 + tid] = (tid == 0) ? rs0 : (tid == 1) ? rs1 : (tid == 2) ? rs2 : rs3;
        __syncthreads();
        if (jq < 15) {
            const double f0 = Fw[i][0], f1 = Fw[i][1], f2 = Fw[i][2], f3 = Fw[i][3];
#pragma unroll
            for (int q = jq + 1; q < 16; ++q) {
                const int c = cg + 4 * q;
                a[q] = fma(-f3, Fw[c][3], fma(-f2, Fw[c][2], fma(-f1, Fw[c][1], fma(-f0, Fw[c][0], a[q]))));
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = cg + 4 * q;
        Dl[i][c] = (c <= i) ? a[q] : 0.0;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int e = tid; e < nb * nb; e += 256) {
            const int r = e / nb, c = e - r * nb;
            if (c <= r) A[(size_t)(k + r) * ld + k + c] = Dl[r][c];
        }
        if (tid < nb) rd[k + tid] = rdiag[tid];
        return;
    }
    const int r0 = k + nb + (blockIdx.x - 1) * CB;
    const int nr = min(CB, n + 1 - r0);
    if (nr <= 0) return;
    double p[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = cg + 4 * q;
        p[q] = (i < nr && c < nb) ? A[(size_t)(r0 + i) * ld + k + c] : 0.0;
    }
    // X L^T = A_panel, row by row: x_j = (a_j - sum_{t<j} x_t L[j][t]) / L[j][j].  The four lanes of a row each hold
    // the x_t with t = cg (mod 4); they form partial sums over their own t and combine them with two quad
    // shuffles, so this phase needs no barrier at all (rows are independent).
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        const int jq = j >> 2, jr = j & 3;
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int q = 0; q < jq; ++q) {
            const double lv = Dl[j][cg + 4 * q];
            if (q & 1) acc1 = fma(p[q], lv, acc1); else acc0 = fma(p[q], lv, acc0);
        }
        {   // columns 4 jq + cg < j only
            const double lv = (cg < jr) ? Dl[j][cg + 4 * jq] : 0.0;
            acc0 = fma(p[jq], lv, acc0);
        }
        double tot = acc0 + acc1;
        tot += __shfl_xor(tot, 1);
        tot += __shfl_xor(tot, 2);
        const double xj = (p[jq] - tot) * rdiag[j];
        p[jq] = (cg == jr) ? xj : p[jq];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = cg + 4 * q;
        if (i < nr && c < nb) A[(size_t)(r0 + i) * ld + k + c] = p[q];
    }
}

// trailing update A[i][j] -= sum_t L[i][k+t] L[j][k+t] for i >= j >= k+nb (j < n), 64x64 tiles, 4x4 per thread
__global__ void __launch_bounds__(256) chol_update(double* __restrict__ A, int ld, int n, int k, double* __restrict__ Dnext)
{
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    __shared__ double Li[CB][CB + 1];
    __shared__ double Lj[CB][CB + 1];
    const int tid = threadIdx.x;
    const int nb = min(CB, n - k);
    const int r0 = k + nb;
    const int i0 = r0 + ti * CB, j0 = r0 + tj * CB;
    const int ni = min(CB, n + 1 - i0), nj = min(CB, n - j0);
    if (ni <= 0 || nj <= 0) return;
#pragma unroll
    for (int it = 0; it < 16; ++it) {                 // 32 independent loads in flight per thread
        const int e = tid + 256 * it;
        const int i = e >> 6, t = e & 63;
        Li[i][t] = (i < ni && t < nb) ? A[(size_t)(i0 + i) * ld + k + t] : 0.0;
        Lj[i][t] = (i < nj && t < nb) ? A[(size_t)(j0 + i) * ld + k + t] : 0.0;
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    double c[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) c[r][q] = 0.0;
#pragma unroll 8
    for (int t = 0; t < CB; ++t) {                    // columns t >= nb are zero padded
        double av[4], bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = Li[ty + 16 * r][t];
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = Lj[tx + 16 * q][t];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) c[r][q] = fma(av[r], bv[q], c[r][q]);
    }
    // epilogue: all 16 loads first (independent), then the stores
    double old[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = ty + 16 * r, j = tx + 16 * q;
            const bool ok = (i < ni) && (j < nj) && (j0 + j <= i0 + i);
            old[r][q] = ok ? A[(size_t)(i0 + i) * ld + j0 + j] : 0.0;
        }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = ty + 16 * r, j = tx + 16 * q;
            const bool ok = (i < ni) && (j < nj) && (j0 + j <= i0 + i);   // lower triangle only
            if (ok) {
                const double v = old[r][q] - c[r][q];
                A[(size_t)(i0 + i) * ld + j0 + j] = v;
                if (ti == 0 && tj == 0) Dnext[i * CB + j] = v;          // next step's diagonal block
            }
        }
}

// Back substitution L^T x = y (y = border row n of the factor), one launch per 64-row block, last block first.
// x_b = L_bb^-T ( y_b - sum_{rows below} L[row][b]^T x[row] ).  The strip product is spread over gridDim.x
// workgroups (64 rows each); the last one to arrive (device-scope counter) reduces the partials and solves the
// 64x64 triangle.  xv is [n] (stripe-free ordering).
__global__ void __launch_bounds__(256) chol_back_step(const double* __restrict__ A, int ld, int n, int kb, double* __restrict__ xv,
                                                      double* __restrict__ partial, unsigned int* __restrict__ counter,
                                                      const double* __restrict__ rd)
{
    __shared__ double red[4][CB];
    __shared__ double D[CB][CB + 1];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int nb = min(CB, n - kb);
    const int c = tid & 63, rg = tid >> 6;
    const int rows_below = n - (kb + nb);
    const int nslice = gridDim.x;
    if (rows_below > 0) {
        const int per = (rows_below + nslice - 1) / nslice;
        const int rb = kb + nb + blockIdx.x * per;
        const int re = min(n, rb + per);
        double acc = 0.0;
        if (c < nb) {
#pragma unroll 8
            for (int row = rb + rg; row < re; row += 4) acc = fma(A[(size_t)row * ld + kb + c], xv[row], acc);
        }
        red[rg][c] = acc;
        __syncthreads();
        if (tid < CB) partial[(size_t)blockIdx.x * CB + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        __syncthreads();
        if (tid == 0) {                                   // one lane releases for the workgroup
            __threadfence();
            const unsigned int prev = atomicAdd(counter, 1u);
            is_last = (prev == (unsigned int)(nslice - 1));
            if (is_last) __threadfence();                 // ... and acquires for the last arriver
        }
        __syncthreads();
        if (!is_last) return;
    } else if (blockIdx.x != 0) return;
    // last arriver: y_b - strip product, then the triangle
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it;
        const int r = e >> 6, q = e & 63;
        D[r][q] = (r < nb && q < nb) ? A[(size_t)(kb + r) * ld + kb + q] : 0.0;
    }
    double ps = 0.0;
    if (rows_below > 0 && c < nb)
        for (int g = rg; g < nslice; g += 4) ps += __hip_atomic_load(&partial[(size_t)g * CB + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[rg][c] = ps;
    __syncthreads();
    if (tid < 64) {
        double yt = 0.0, rdj = 1.0;
        if (tid < nb) {
            yt = A[(size_t)n * ld + kb + tid] - (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
            rdj = rd[kb + tid];
        }
        for (int j = nb - 1; j >= 0; --j) {
            const double xj = __shfl(yt, j) * __shfl(rdj, j);
            if (tid == j) yt = xj;
            else if (tid < j) yt = fma(-D[j][tid], xj, yt);
        }
        if (tid < nb) xv[kb + tid] = yt;
        if (tid == 0) *counter = 0u;
    }
}

// Extend_Solution / Restore_Solution scatter (SFFTConfigure.py:1299-1311; BSplineSFFT.py:2274-2338):
// solution[idx[i]] = x[i]; removed entries stay zero, tied entries all receive the value of their representative
__global__ void __launch_bounds__(256) scatter_solution(const double* __restrict__ xv, int n, const int* __restrict__ idx,
                                                        double* __restrict__ solution, int NEQ, int tie_first, int tie_cnt, int tie_stride)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n) solution[idx ? idx[t] : t] = xv[t];
    if (t >= 1 && t < tie_cnt) solution[tie_first + t * tie_stride] = xv[tie_first];   // position of tie_first in x equals its value
}

// ---- LU with partial pivoting (fallback; matches the reference's getrf/gesv semantics) --------------------
// A is [(n+1)][ld]; rows < n, columns <= n (column n = rhs).  Unblocked right-looking elimination.
__global__ void __launch_bounds__(1024) lu_pivot(double* __restrict__ A, int ld, int n, int k, int* __restrict__ status)
{
    __shared__ double bestv[16];
    __shared__ int besti[16];
    __shared__ int piv;
    const int tid = threadIdx.x;
    double bv = -1.0; int bi = k;
    for (int i = k + tid; i < n; i += 1024) {
        const double v = fabs(A[(size_t)i * ld + k]);
        if (v > bv) { bv = v; bi = i; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(bv, off); const int oi = __shfl_down(bi, off);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { bestv[tid >> 6] = bv; besti[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) if (bestv[w] > bv || (bestv[w] == bv && besti[w] < bi)) { bv = bestv[w]; bi = besti[w]; }
        piv = bi;
        if (!(bv > 0.0)) atomicOr(status, 2);
    }
    __syncthreads();
    const int p = piv;
    if (p != k) {
        for (int c = tid; c <= n; c += 1024) {
            const double t = A[(size_t)k * ld + c];
            A[(size_t)k * ld + c] = A[(size_t)p * ld + c];
            A[(size_t)p * ld + c] = t;
        }
    }
    __syncthreads();
    const double d = A[(size_t)k * ld + k];
    for (int i = k + 1 + tid; i < n; i += 1024) A[(size_t)i * ld + k] /= d;
}

__global__ void __launch_bounds__(256) lu_rank1(double* __restrict__ A, int ld, int n, int k)
{
    const int j = k + 1 + blockIdx.x * 64 + (threadIdx.x & 63);
    const int ib = k + 1 + blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
    if (j > n) return;
    const double u = A[(size_t)k * ld + j];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = ib + r;
        if (i < n) A[(size_t)i * ld + j] = fma(-A[(size_t)i * ld + k], u, A[(size_t)i * ld + j]);
    }
}

__global__ void __launch_bounds__(1024) lu_backsolve(const double* __restrict__ A, int ld, int n, double* __restrict__ xv)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* yv = reinterpret_cast<double*>(smem_raw);
    const int tid = threadIdx.x;
    for (int c = tid; c < n; c += 1024) yv[c] = A[(size_t)c * ld + n];
    __syncthreads();
    for (int i = n - 1; i >= 0; --i) {
        if (tid == 0) yv[i] = yv[i] / A[(size_t)i * ld + i];
        __syncthreads();
        const double xi = yv[i];
        for (int c = tid; c < i; c += 1024) yv[c] = fma(-A[(size_t)c * ld + i], xi, yv[c]);
        __syncthreads();
    }
    for (int i = tid; i < n; i += 1024) xv[i] = yv[i];
}

// ------------------------------------------------------------------------------------------------
// Subtraction: kernel transfer function tables + Construct_FDIFF (SFFTConfigure.py:737-809)
//   Ctab[ij][a][m] = sum_b a_ijab W1^(m b);   Soff[ij] = sum_{ab != centre} a_ijab
//   FD[l][m] = sum_ij FI_ij[l][m] * SCALE * ( sum_a W0^(l a) Ctab[ij][a][m] - Soff[ij] )
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) kernel_ctab(const double* __restrict__ sol, cplx* __restrict__ Ctab, double* __restrict__ Soff,
                                                   int Fij, int L0, int L1, int w1, int Nh, int Nhp, int N1,
                                                   const cplx* __restrict__ root1)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    const int ija = blockIdx.y;                   // ij*L0 + a
    const int Fab = L0 * L1;
    if (m == 0 && (ija % L0) == 0) {              // one thread per ij
        const int ij = ija / L0;
        double sacc = 0.0;
        const int cen = (L0 / 2) * L1 + w1;
        for (int ab = 0; ab < Fab; ++ab) if (ab != cen) sacc += sol[ij * Fab + ab];
        Soff[ij] = sacc;
    }
    if (m >= Nh) return;
    const double* arow = sol + (size_t)ija * L1;  // ij*Fab + a*L1
    double cxr = 0.0, cyi = 0.0;
    for (int bb = 0; bb < L1; ++bb) {
        const int b = bb - w1;
        long long q = ((long long)m * b) % N1; if (q < 0) q += N1;
        const cplx w = root1[q];
        cxr = fma(arow[bb], w.x, cxr);
        cyi = fma(arow[bb], w.y, cyi);
    }
    Ctab[(size_t)ija * Nhp + m] = make_double2(cxr, cyi);
}

#define CRL 8
__global__ void __launch_bounds__(256) construct_fd(const cplx* __restrict__ FI, cplx* __restrict__ FD, const cplx* __restrict__ Ctab,
                                                    const double* __restrict__ Soff, const cplx* __restrict__ root0,
                                                    int N0, int Nh, int Nhp, int Fij, int L0, int w0, double scale)
{
    __shared__ cplx wl[CRL][72];
    const int tid = threadIdx.x;
    const int m = blockIdx.x * 256 + tid;
    const int lbase = blockIdx.y * CRL;
    for (int e = tid; e < CRL * L0; e += 256) {
        const int r = e / L0, aa = e - r * L0;
        const int l = lbase + r;
        long long q = ((long long)l * (aa - w0)) % N0; if (q < 0) q += N0;
        wl[r][aa] = root0[q];
    }
    __syncthreads();
    if (m >= Nh) return;
    cplx acc[CRL];
#pragma unroll
    for (int r = 0; r < CRL; ++r) acc[r] = make_double2(0.0, 0.0);
    const size_t plane_sz = (size_t)N0 * Nhp;
    for (int ij = 0; ij < Fij; ++ij) {
        cplx kk[CRL];
#pragma unroll
        for (int r = 0; r < CRL; ++r) kk[r] = make_double2(0.0, 0.0);
        for (int aa = 0; aa < L0; ++aa) {
            const cplx c = Ctab[((size_t)ij * L0 + aa) * Nhp + m];
#pragma unroll
            for (int r = 0; r < CRL; ++r) {
                const cplx w = wl[r][aa];
                kk[r].x = fma(w.x, c.x, fma(-w.y, c.y, kk[r].x));
                kk[r].y = fma(w.x, c.y, fma(w.y, c.x, kk[r].y));
            }
        }
        const double so = Soff[ij];
#pragma unroll
        for (int r = 0; r < CRL; ++r) {
            const int l = lbase + r;
            if (l < N0) {
                const cplx fi = FI[(size_t)ij * plane_sz + (size_t)l * Nhp + m];
                const cplx kf = make_double2(scale * (kk[r].x - so), scale * kk[r].y);
                acc[r] = cadd(acc[r], cmul(fi, kf));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < CRL; ++r) {
        const int l = lbase + r;
        if (l < N0) FD[(size_t)l * Nhp + m] = acc[r];
    }
}

// ------------------------------------------------------------------------------------------------
// Small spectrum-arithmetic kernels behind the FFT utilities (noise decorrelation, FFT convolution:
// sfft/utils/PureCupyFFTKits.py, PureCupyDeCorrelationCalculator.py)
// ------------------------------------------------------------------------------------------------
__global__ void copy_spectrum_scaled(const cplx* __restrict__ src, cplx* __restrict__ dst, int N0, int Nh, int src_ld, int dst_ld, double f)
{
    const int m = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
    if (m < Nh) { const cplx v = src[(size_t)l * src_ld + m]; dst[(size_t)l * dst_ld + m] = make_double2(v.x * f, v.y * f); }
}
__global__ void scale_real(double* __restrict__ a, double f, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] *= f;
}
// acc[i] += coeff * |a[i]|^2 * (b ? |b[i]|^2 : 1)
__global__ void spec_abs2_acc(const cplx* __restrict__ a, const cplx* __restrict__ b, double coeff, double* __restrict__ acc, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const cplx u = a[i];
    double v = coeff * (u.x * u.x + u.y * u.y);
    if (b) { const cplx w = b[i]; v *= (w.x * w.x + w.y * w.y); }
    acc[i] += v;
}
__global__ void real_rsqrt(const double* __restrict__ acc, double* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = 1.0 / sqrt(acc[i]);
}
// out[i] = a[i] * (b is complex ? b[i] : breal[i])
__global__ void spec_mul(const cplx* __restrict__ a, const cplx* __restrict__ b, const double* __restrict__ breal, cplx* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const cplx u = a[i];
    if (b) out[i] = cmul(u, b[i]);
    else { const double r = breal[i]; out[i] = make_double2(u.x * r, u.y * r); }
}
// full[l][m] of a real, conjugate-symmetric spectrum quantity from its half [N0][Nh]: full[l][N1-m] = half[(N0-l)%N0][m]
__global__ void half_to_full_real(const double* __restrict__ half, double* __restrict__ full, int N0, int N1, int Nh)
{
    const int m = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
    if (m >= N1) return;
    full[(size_t)l * N1 + m] = (m < Nh) ? half[(size_t)l * Nh + m] : half[(size_t)((N0 - l) % N0) * Nh + (N1 - m)];
}

// debug: copy a padded half-spectrum plane to a dense [N0][Nh] array
__global__ void copy_spectrum(const cplx* __restrict__ src, cplx* __restrict__ dst, int N0, int Nh, int Nhp)
{
    const int m = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
    if (m < Nh) dst[(size_t)l * Nh + m] = src[(size_t)l * Nhp + m];
}

// ================================================================================================
// host side
// ================================================================================================
struct AxisHost {
    int N = 0, M = 0, logM = 0, blue = 0;
    cplx *tw = nullptr, *chirp = nullptr, *bf = nullptr, *root = nullptr;
    bool root_is_tw = false;
    // four-step decomposition for lengths that do not fit one on-chip transform: N = A * B
    bool big = false;
    int A = 0, B = 0;
    AxisHost* subA = nullptr;
    AxisHost* subB = nullptr;
};

struct sfft_plan {
    int dev = 0;
    int N0 = 0, N1 = 0, w = 0, DK = 0, DB = 0, cpr = 0;
    int L = 0, Fab = 0, Fij = 0, Fpq = 0, Fijab = 0, NEQ = 0, NEQfs = 0;
    int Nh = 0, Nhp = 0;
    double scale = 0.0;
    AxisHost ax0, ax1;
    int TC = 1, MS = 0;                 // column pass tiling
    int nt_rows = 64, nt_cols = 64;
    size_t lds_rows = 0, lds_cols = 0;
    // separable spatial bases (polynomial powers or B-spline basis functions), tabulated per axis
    int nkx = 0, nky = 0, nbx = 0, nby = 0;
    std::vector<int> kpair, bpair;      // [Fij][2] / [Fpq][2]: (x-factor, y-factor) of kernel term ij / background term pq
    double *d_kbx = nullptr, *d_kby = nullptr, *d_tbx = nullptr, *d_tby = nullptr;   // [nkx][N0], [nky][N1], [nbx][N0], [nby][N1]
    int mode = 0;                       // 0: free scaling, 1: unknowns ij00[1:] removed, 2: unknowns ij00 tied together
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

    int S = 1, rows_per_chunk = 0;
    FillArgs fa;
    // workspaces
    cplx* d_spec = nullptr;             // [Fij+1][N0][Nhp]   (plane Fij: J in solve, FD in apply)
    cplx *d_big1 = nullptr, *d_big2 = nullptr, *d_colscr = nullptr;   // work arrays of the four-step path
    double *d_zero = nullptr, *d_zsol = nullptr;   // zero image / zero solution for the stand-alone inverse FFT (lazy)
    cplx* d_spec2 = nullptr;            // [Fij][N0][Nhp] spectra of the full pair, filled on stream s2 during the solve (lazy)
    hipStream_t s2 = nullptr; hipEvent_t ev_in = nullptr, ev_pre = nullptr; int no_overlap = 0;
    const double* overlap_I = nullptr;  // set by sfft_subtract for the duration of its sfft_solve call
    cplx* d_gp = nullptr;
    double* d_patches = nullptr; size_t n_patches = 0;
    double* d_A = nullptr; int ld = 0;
    double* d_dbuf = nullptr;           // [2][CB][CB] diagonal blocks handed from chol_update to chol_panel
    double* d_xv = nullptr;             // [NEQfs] solution in stripe-free ordering
    double* d_rd = nullptr;             // [NEQfs] reciprocal diagonal of the Cholesky factor
    double* d_partial = nullptr;        // [BACK_SLICES][CB] strip-product partials of the back substitution
    unsigned int* d_counter = nullptr;
    double* d_sol = nullptr;            // [NEQ] internal solution copy
    cplx* d_ctab = nullptr; double* d_soff = nullptr;
    double* d_rowmom = nullptr; double* d_delta = nullptr;
    int* d_status = nullptr;
    size_t ws_bytes = 0;
    int last_solver = 0, force_lu = 0;
    int no_fast_fft = 0;                // env SFFT_NO_FAST_FFT=1: use the generic LDS FFT for 4096-point axes too (A/B testing)
    int g1_variant = 2;                 // tuning knob (env SFFT_G1_VARIANT): how the lags of a Greek pass are split over waves
    int timing = 0;
    hipEvent_t ev[SFFT_ST_COUNT][2];
    bool ev_valid[SFFT_ST_COUNT];
    bool have_system = false;
};

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

template <typename T>
static int dev_alloc(sfft_plan* p, T** ptr, size_t count)
{
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T) > 0 ? count * sizeof(T) : 16);
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
static const size_t LDS_COL_ELEMS = 9216;   // column tile budget (144 KiB): two padded 4096-point columns fit

static bool fits_on_chip(int N)
{
    if (is_pow2(N)) return (size_t)N <= LDS_MAX_ELEMS;
    int M = 1; while (M < 2 * N - 1) M <<= 1;
    return (size_t)M <= LDS_MAX_ELEMS;
}

static int build_axis(sfft_plan* p, AxisHost& ax, int N);

// N = A * B with A the largest power-of-two factor (<= 4096) such that B fits on chip too
static int build_big_axis(sfft_plan* p, AxisHost& ax, int N)
{
    const long double PI = acosl(-1.0L);
    int a = 1;
    while (N % (a * 2) == 0 && a * 2 <= 4096) a *= 2;
    int A = 0, B = 0;
    for (; a >= 2; a /= 2) { if (fits_on_chip(N / a)) { A = a; B = N / a; break; } }
    if (!A) return set_err(SFFT_ERR_UNSUPPORTED_SIZE,
                           "image side not supported by this build: it must fit one on-chip transform (power of two <= 8192, "
                           "any length <= 4096) or factor as 2^k * B with both factors on chip");
    ax.N = N; ax.big = true; ax.A = A; ax.B = B; ax.M = 0; ax.logM = 0; ax.blue = 0;
    ax.subA = new AxisHost(); ax.subB = new AxisHost();
    int rc;
    if ((rc = build_axis(p, *ax.subA, A))) return rc;
    if ((rc = build_axis(p, *ax.subB, B))) return rc;
    std::vector<cplx> r(N);
    for (int k = 0; k < N; ++k) {
        const long double ang = -2.0L * PI * k / N;
        r[k] = make_double2((double)cosl(ang), (double)sinl(ang));
    }
    if ((rc = dev_alloc(p, &ax.root, N))) return rc;
    HIPCHK(hipMemcpy(ax.root, r.data(), N * sizeof(cplx), hipMemcpyHostToDevice));
    return SFFT_OK;
}

static int build_axis(sfft_plan* p, AxisHost& ax, int N)
{
    const long double PI = acosl(-1.0L);
    if (!fits_on_chip(N)) return build_big_axis(p, ax, N);
    ax.N = N;
    if (is_pow2(N)) { ax.M = N; ax.blue = 0; }
    else { int M = 1; while (M < 2 * N - 1) M <<= 1; ax.M = M; ax.blue = 1; }
    ax.logM = ilog2(ax.M);
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
    host_fft_pow2(fr, fi);
    std::vector<cplx> bf(ax.M);
    for (int k = 0; k < ax.M; ++k) bf[k] = make_double2((double)(fr[k] / ax.M), (double)(fi[k] / ax.M));
    if ((rc = dev_alloc(p, &ax.bf, ax.M))) return rc;
    HIPCHK(hipMemcpy(ax.bf, bf.data(), ax.M * sizeof(cplx), hipMemcpyHostToDevice));
    return SFFT_OK;
}

static AxisDev axis_dev(const AxisHost& a)
{
    AxisDev d; d.N = a.N; d.M = a.M; d.logM = a.logM; d.blue = a.blue; d.tw = a.tw; d.chirp = a.chirp; d.bf = a.bf; d.root = a.root;
    return d;
}

static double ipow_host(double x, int e) { double r = 1.0; for (int t = 0; t < e; ++t) r *= x; return r; }

struct BasisSpec {
    int nkx = 0, nky = 0, nbx = 0, nby = 0, Fij = 0, Fpq = 0, mode = 0;
    std::vector<double> kbx, kby, tbx, tby;   // [nkx][N0], [nky][N1], [nbx][N0], [nby][N1]
    std::vector<int> kpair, bpair;            // [Fij][2], [Fpq][2]
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

static int plan_create_impl(sfft_plan** out, int N0, int N1, int KerHW, const BasisSpec& BS, int DK, int DB, int device)
{
    if (!out) return set_err(SFFT_ERR_INVALID_ARG, "plan pointer is NULL");
    *out = nullptr;
    if (N0 < 8 || N1 < 8) return set_err(SFFT_ERR_INVALID_ARG, "Input Image has dramatically small size!");
    if (KerHW < 0 || KerHW > 32) return set_err(SFFT_ERR_INVALID_ARG, "KerHW must be in [0, 32]");
    if (BS.Fij < 1 || BS.Fij > 64 || BS.Fpq < 1 || BS.Fpq > SFFT_MAX_PQ || BS.nby > SFFT_MAX_BQ || BS.nbx > 16 || BS.nkx > 16 || BS.nky > 16)
        return set_err(SFFT_ERR_INVALID_ARG, "spatial basis too large: at most 64 kernel terms, 64 background terms, 16 factors per axis");
    HIPCHK(hipSetDevice(device));
    sfft_plan* p = new sfft_plan();
    p->dev = device;
    if (const char* ev = getenv("SFFT_G1_VARIANT")) p->g1_variant = atoi(ev);
    if (const char* ev = getenv("SFFT_NO_FAST_FFT")) p->no_fast_fft = atoi(ev);
    if (const char* ev = getenv("SFFT_NO_OVERLAP")) p->no_overlap = atoi(ev);
    p->N0 = N0; p->N1 = N1; p->w = KerHW; p->DK = DK; p->DB = DB; p->mode = BS.mode; p->cpr = BS.mode != 0;
    p->L = 2 * KerHW + 1; p->Fab = p->L * p->L;
    p->Fij = BS.Fij; p->Fpq = BS.Fpq;
    p->nkx = BS.nkx; p->nky = BS.nky; p->nbx = BS.nbx; p->nby = BS.nby;
    p->kpair = BS.kpair; p->bpair = BS.bpair;
    p->Fijab = p->Fij * p->Fab; p->NEQ = p->Fijab + p->Fpq;
    p->NEQfs = p->cpr ? p->NEQ - (p->Fij - 1) : p->NEQ;
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
        if (const char* ev = getenv("SFFT_S2_CUMASK")) pat = (uint32_t)strtoul(ev, nullptr, 0);
        hipDeviceProp_t prop;
        PLAN_HIP(hipGetDeviceProperties(&prop, device));
        const int nwords = (prop.multiProcessorCount + 31) / 32;
        std::vector<uint32_t> mask(nwords, pat);
        if (pat == 0 || hipExtStreamCreateWithCUMask(&p->s2, (uint32_t)nwords, mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            PLAN_HIP(hipStreamCreateWithPriority(&p->s2, hipStreamNonBlocking, lo));
        }
        PLAN_HIP(hipEventCreateWithFlags(&p->ev_in, hipEventDisableTiming));
        PLAN_HIP(hipEventCreateWithFlags(&p->ev_pre, hipEventDisableTiming));
    }
    {   // basis tables on the device
        PLAN_TRY(dev_alloc(p, &p->d_kbx, (size_t)BS.nkx * N0));
        PLAN_TRY(dev_alloc(p, &p->d_kby, (size_t)BS.nky * N1));
        PLAN_TRY(dev_alloc(p, &p->d_tbx, (size_t)BS.nbx * N0));
        PLAN_TRY(dev_alloc(p, &p->d_tby, (size_t)BS.nby * N1));
        PLAN_HIP(hipMemcpy(p->d_kbx, BS.kbx.data(), (size_t)BS.nkx * N0 * sizeof(double), hipMemcpyHostToDevice));
        PLAN_HIP(hipMemcpy(p->d_kby, BS.kby.data(), (size_t)BS.nky * N1 * sizeof(double), hipMemcpyHostToDevice));
        PLAN_HIP(hipMemcpy(p->d_tbx, BS.tbx.data(), (size_t)BS.nbx * N0 * sizeof(double), hipMemcpyHostToDevice));
        PLAN_HIP(hipMemcpy(p->d_tby, BS.tby.data(), (size_t)BS.nby * N1 * sizeof(double), hipMemcpyHostToDevice));
        memset(&p->bk, 0, sizeof(p->bk));
        p->bk.npq = p->Fpq; p->bk.nq = BS.nby; p->bk.tbx = p->d_tbx; p->bk.tby = p->d_tby;
        for (int t = 0; t < p->Fpq; ++t) { p->bk.p[t] = BS.bpair[2 * t]; p->bk.q[t] = BS.bpair[2 * t + 1]; }
    }
    PLAN_TRY(build_axis(p, p->ax0, N0));
    PLAN_TRY(build_axis(p, p->ax1, N1));
    // launch geometry of the on-chip FFT kernels (axes that need the four-step path use strided_dft instead)
    if (!p->ax1.big) {
        p->nt_rows = std::min(1024, std::max(64, p->ax1.M / 16));
        p->lds_rows = (size_t)p->ax1.M * sizeof(cplx);
    }
    if (!p->ax0.big) {
        p->MS = p->ax0.M + 1;
        p->TC = 1;
        while (p->TC < 16 && (size_t)(2 * p->TC) * p->MS <= LDS_COL_ELEMS) p->TC *= 2;
        p->nt_cols = std::min(1024, std::max(64, p->TC * p->ax0.M / 16));
        p->lds_cols = (size_t)p->TC * p->MS * sizeof(cplx);
    }
    PLAN_HIP(hipFuncSetAttribute((const void*)strided_dft, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (p->ax1.big) {          // two packed-row work arrays [ceil(N0/2)][N1]
        PLAN_TRY(dev_alloc(p, &p->d_big1, (size_t)((N0 + 1) / 2) * N1));
        PLAN_TRY(dev_alloc(p, &p->d_big2, (size_t)((N0 + 1) / 2) * N1));
    }
    if (p->ax0.big) PLAN_TRY(dev_alloc(p, &p->d_colscr, (size_t)N0 * p->Nhp));
    PLAN_HIP(hipFuncSetAttribute((const void*)rows_r2c, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)rows_r2c_4096, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)cols_c2c_4096, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)rows_c2r_diff_4096<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)rows_c2r_diff_4096<SFFT_MAX_BQ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)cols_c2c, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)rows_c2r_diff, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PLAN_HIP(hipFuncSetAttribute((const void*)lu_backsolve, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if ((size_t)(p->NEQfs + 2) * 8 + (size_t)CB * (CB + 1) * 8 > 150 * 1024) {
        sfft_plan_destroy(p);
        return set_err(SFFT_ERR_UNSUPPORTED_SIZE, "linear system too large for the on-chip back substitution of this build");
    }

    // index map of Remove_LSFStripes (SFFTSubtract.py:83-90); with tied scaling (mode 2) the kept entry ij00[0]
    // stands for the whole tied group
    if (p->cpr) {
        std::vector<int> idx;
        std::vector<char> forb(p->NEQ, 0);
        const int ij00_first = KerHW * p->L + KerHW;
        for (int ij = 1; ij < p->Fij; ++ij) forb[ij00_first + ij * p->Fab] = 1;
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
        const int npass_est = p->Fij * (p->Fij + 1) / 2 + p->Fij * BS.nbx + p->Fij;
        while (S < 16 && (long long)colblocks * S * npass_est < 6144 && N0 / (2 * S) >= 64) S *= 2;
        p->S = S;
        p->rows_per_chunk = (N0 + S - 1) / S;
        long long goff = 0;
        auto add_pass = [&](int a, int b, int bp, int h) {
            G1Pass d; d.a_plane = a; d.b_plane = b; d.bp = bp; d.h = h; d.gp_off = goff;
            p->passes.push_back(d); goff += (long long)S * (2 * h + 1) * p->Nhp;
            return (int)p->passes.size() - 1;
        };
        std::vector<int> omg_pass, the_pass, gam_pass((size_t)p->Fij * BS.nbx);
        for (int a = 0; a < p->Fij; ++a) for (int b = a; b < p->Fij; ++b) omg_pass.push_back(add_pass(a, b, 0, hO));
        p->n_omg = (int)omg_pass.size();
        for (int a = 0; a < p->Fij; ++a) the_pass.push_back(add_pass(a, p->Fij, 0, hG));
        p->n_the = p->Fij;
        // Gamma column-factor passes: dense ones first, then those whose x-factor is the constant 1 (Xp = N0 * delta)
        p->n_gamp = 0; p->n_gam0 = 0;
        for (int a = 0; a < p->Fij; ++a) for (int e = 0; e < BS.nbx; ++e) if (!const_x[e]) { gam_pass[(size_t)a * BS.nbx + e] = add_pass(a, -1, e, hG); ++p->n_gamp; }
        for (int a = 0; a < p->Fij; ++a) for (int e = 0; e < BS.nbx; ++e) if (const_x[e]) { gam_pass[(size_t)a * BS.nbx + e] = add_pass(a, -1, e, hG); ++p->n_gam0; }
        int poff = 0;
        auto add_job = [&](int pass, int yq, int h, double scale) {
            PatchJob j; j.pass = pass; j.yq = yq; j.h = h; j.patch_off = poff; j.scale = scale;
            p->jobs.push_back(j); poff += (2 * h + 1) * (2 * h + 1);
        };
        p->fa.omg_off = poff;
        for (int k = 0; k < p->n_omg; ++k) add_job(omg_pass[k], -1, hO, p->scale * p->scale);   // PreOMG = SCALE*Re[SCALE*DFT] (SFFTSubtract.py:233-240)
        p->fa.gam_off = poff;
        for (int a = 0; a < p->Fij; ++a) for (int q = 0; q < p->Fpq; ++q)
            add_job(gam_pass[(size_t)a * BS.nbx + BS.bpair[2 * q]], BS.bpair[2 * q + 1], hG, p->scale);   // PreGAM = Re[SCALE*DFT] (:262-268)
        p->n_gam = p->Fij * p->Fpq;
        p->fa.the_off = poff;
        for (int a = 0; a < p->Fij; ++a) add_job(the_pass[a], -1, hG, p->scale);                  // PreTHE = Re[SCALE*DFT] (:353-362)
        p->n_patches = poff;
        (void)PHo; (void)PHg;
        PLAN_TRY(dev_alloc(p, &p->d_passes, p->passes.size()));
        PLAN_HIP(hipMemcpy(p->d_passes, p->passes.data(), p->passes.size() * sizeof(G1Pass), hipMemcpyHostToDevice));
        PLAN_TRY(dev_alloc(p, &p->d_jobs, p->jobs.size()));
        PLAN_HIP(hipMemcpy(p->d_jobs, p->jobs.data(), p->jobs.size() * sizeof(PatchJob), hipMemcpyHostToDevice));
        PLAN_TRY(dev_alloc(p, &p->d_gp, (size_t)goff));
        p->hm = hO + 1;
        PLAN_TRY(dev_alloc(p, &p->d_w0tab, (size_t)N0 * p->hm));
        hipLaunchKernelGGL(build_w0tab, dim3((N0 * p->hm + 255) / 256), dim3(256), 0, 0, p->ax0.root, p->d_w0tab, N0, p->hm);
        PLAN_TRY(dev_alloc(p, &p->d_patches, (size_t)poff));
        p->fa.Fij = p->Fij; p->fa.Fpq = p->Fpq; p->fa.Fab = p->Fab; p->fa.Fijab = p->Fijab; p->fa.L1 = p->L;
        p->fa.w0 = KerHW; p->fa.w1 = KerHW; p->fa.h_omg = hO; p->fa.h_gam = hG;
        p->fa.tie_first = KerHW * p->L + KerHW; p->fa.tie_stride = p->Fab; p->fa.tie_cnt = (p->mode == 2) ? p->Fij : 0;
    }
    PLAN_TRY(dev_alloc(p, &p->d_spec, (size_t)(p->Fij + 1) * N0 * p->Nhp));
    p->ld = (p->NEQfs + 1 + 3) & ~3;
    PLAN_TRY(dev_alloc(p, &p->d_A, (size_t)(p->NEQfs + 1) * p->ld));
    PLAN_TRY(dev_alloc(p, &p->d_dbuf, (size_t)2 * CB * CB));
    PLAN_TRY(dev_alloc(p, &p->d_xv, (size_t)p->NEQfs));
    PLAN_TRY(dev_alloc(p, &p->d_rd, (size_t)p->NEQfs));
    PLAN_TRY(dev_alloc(p, &p->d_partial, (size_t)BACK_SLICES * CB));
    PLAN_TRY(dev_alloc(p, &p->d_counter, (size_t)1));
    PLAN_HIP(hipMemset(p->d_counter, 0, sizeof(unsigned int)));
    PLAN_TRY(dev_alloc(p, &p->d_sol, (size_t)p->NEQ));
    PLAN_TRY(dev_alloc(p, &p->d_ctab, (size_t)p->Fij * p->L * p->Nhp));
    PLAN_TRY(dev_alloc(p, &p->d_soff, (size_t)p->Fij));
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
extern "C" int sfft_plan_create_basis(sfft_plan** out, int N0, int N1, int KerHW,
                                      int nkx, int nky, const double* kbx, const double* kby, int Fij, const int* ker_pairs,
                                      int nbx, int nby, const double* tbx, const double* tby, int Fpq, const int* bkg_pairs,
                                      int scaling_mode, int device)
{
    if (!kbx || !kby || !tbx || !tby || !ker_pairs || !bkg_pairs) return set_err(SFFT_ERR_INVALID_ARG, "NULL basis table");
    if (scaling_mode < 0 || scaling_mode > 2) return set_err(SFFT_ERR_INVALID_ARG, "scaling_mode must be 0, 1 or 2");
    if (N0 < 8 || N1 < 8) return set_err(SFFT_ERR_INVALID_ARG, "Input Image has dramatically small size!");
    if (nkx < 1 || nky < 1 || nbx < 1 || nby < 1 || Fij < 1 || Fpq < 1) return set_err(SFFT_ERR_INVALID_ARG, "empty basis");
    BasisSpec B;
    B.nkx = nkx; B.nky = nky; B.nbx = nbx; B.nby = nby; B.Fij = Fij; B.Fpq = Fpq; B.mode = scaling_mode;
    B.kbx.assign(kbx, kbx + (size_t)nkx * N0); B.kby.assign(kby, kby + (size_t)nky * N1);
    B.tbx.assign(tbx, tbx + (size_t)nbx * N0); B.tby.assign(tby, tby + (size_t)nby * N1);
    B.kpair.assign(ker_pairs, ker_pairs + 2 * (size_t)Fij); B.bpair.assign(bkg_pairs, bkg_pairs + 2 * (size_t)Fpq);
    for (int k = 0; k < Fij; ++k) if (B.kpair[2 * k] < 0 || B.kpair[2 * k] >= nkx || B.kpair[2 * k + 1] < 0 || B.kpair[2 * k + 1] >= nky)
        return set_err(SFFT_ERR_INVALID_ARG, "kernel term refers to a basis factor that does not exist");
    for (int k = 0; k < Fpq; ++k) if (B.bpair[2 * k] < 0 || B.bpair[2 * k] >= nbx || B.bpair[2 * k + 1] < 0 || B.bpair[2 * k + 1] >= nby)
        return set_err(SFFT_ERR_INVALID_ARG, "background term refers to a basis factor that does not exist");
    return plan_create_impl(out, N0, N1, KerHW, B, -1, -1, device);
}

static void free_axis(AxisHost& a)
{
    if (a.subA) { free_axis(*a.subA); delete a.subA; a.subA = nullptr; }
    if (a.subB) { free_axis(*a.subB); delete a.subB; a.subB = nullptr; }
    if (a.tw) hipFree(a.tw);
    if (a.root && !a.root_is_tw) hipFree(a.root);
    if (a.chirp) hipFree(a.chirp);
    if (a.bf) hipFree(a.bf);
    a.tw = a.root = a.chirp = a.bf = nullptr;
}

extern "C" int sfft_plan_destroy(sfft_plan* p)
{
    if (!p) return SFFT_OK;
    hipSetDevice(p->dev);
    free_axis(p->ax0); free_axis(p->ax1);
    void* ptrs[] = {p->d_idx, p->d_phi, p->d_Xp, p->d_Yq, p->d_passes, p->d_jobs, p->d_spec, p->d_gp, p->d_patches, p->d_A, p->d_sol,
                    p->d_ctab, p->d_soff, p->d_rowmom, p->d_delta, p->d_status, p->d_dbuf, p->d_xv, p->d_partial, p->d_counter, p->d_w0tab, p->d_rd, p->d_spec2, p->d_big1, p->d_big2, p->d_colscr, p->d_kbx, p->d_kby, p->d_tbx, p->d_tby, p->d_zero, p->d_zsol};
    for (void* q : ptrs) if (q) hipFree(q);
    for (int s = 0; s < SFFT_ST_COUNT; ++s) { if (p->ev[s][0]) hipEventDestroy(p->ev[s][0]); if (p->ev[s][1]) hipEventDestroy(p->ev[s][1]); }
    if (p->s2) { hipStreamSynchronize(p->s2); hipStreamDestroy(p->s2); }
    if (p->ev_in) hipEventDestroy(p->ev_in);
    if (p->ev_pre) hipEventDestroy(p->ev_pre);
    delete p;
    return SFFT_OK;
}

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
        case SFFT_Q_NEQ_FSFREE: *v = p->NEQ - (p->Fij - 1); break;
        case SFFT_Q_FOMG: *v = p->Fij * p->Fij; break;
        case SFFT_Q_FGAM: case SFFT_Q_FPSI: *v = p->Fij * p->Fpq; break;
        case SFFT_Q_FTHE: *v = p->Fij; break;
        case SFFT_Q_FPHI: *v = p->Fpq * p->Fpq; break;
        case SFFT_Q_FDEL: *v = p->Fpq; break;
        case SFFT_Q_WORKSPACE_BYTES: *v = (long long)p->ws_bytes; break;
        case SFFT_Q_LAST_SOLVER: *v = p->last_solver; break;
        case SFFT_Q_NUM_GREEK_PAIRS: *v = (long long)(p->n_omg + p->n_the + p->n_gamp); break;
        default: return set_err(SFFT_ERR_INVALID_ARG, "unknown query field");
    }
    return SFFT_OK;
}

extern "C" int sfft_set_timing(sfft_plan* p, int enable) { if (!p) return set_err(SFFT_ERR_INVALID_ARG, "NULL plan"); p->timing = enable; return SFFT_OK; }
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

struct StageTimer {
    sfft_plan* p; int st; hipStream_t s;
    StageTimer(sfft_plan* p_, int st_, hipStream_t s_) : p(p_), st(st_), s(s_) { if (p->timing) { hipEventRecord(p->ev[st][0], s); } }
    ~StageTimer() { if (p->timing) { hipEventRecord(p->ev[st][1], s); p->ev_valid[st] = true; } }
};

#define LAUNCH_CHECK() HIPCHK(hipGetLastError())

// forward transforms of `nplanes` polynomial-weighted planes into spec planes [0, nplanes)
static bool fast_axis(const AxisHost& a) { return !a.big && !a.blue && a.M == 4096; }

// one pass of batched strided sub-transforms
static void launch_pass(sfft_plan* p, const cplx* in, cplx* out, PassDesc d, const AxisHost& sub, const cplx* rootN, hipStream_t s)
{
    const int MS = sub.M + 1;
    int TC = 1;
    while (TC < 16 && (size_t)(2 * TC) * MS <= LDS_COL_ELEMS) TC *= 2;
    const int nt = std::min(1024, std::max(64, TC * sub.M / 16));
    const int ngroups = (d.mode == 2) ? (d.nlines + TC - 1) / TC : (d.J + TC - 1) / TC;
    const int gy = (d.mode == 2) ? d.J : d.nlines;
    hipLaunchKernelGGL(strided_dft, dim3(ngroups, gy), dim3(nt), (size_t)TC * MS * sizeof(cplx), s, in, out, d, axis_dev(sub), rootN, TC, MS);
}

// four-step transform of `nlines` lines of length ax.N; element stride st, line stride lst (complex elements).
// Result lands in `data` again (scr is a same-shaped scratch).  inverse: e^{+i} (conjugation on the way in and out).
static void big_axis_transform(sfft_plan* p, const AxisHost& ax, cplx* data, cplx* scr, long long st, long long lst, int nlines,
                               bool lines_fastest, int inverse, hipStream_t s)
{
    PassDesc d1; memset(&d1, 0, sizeof(d1));
    d1.len = ax.A; d1.J = ax.B; d1.nlines = nlines; d1.mode = lines_fastest ? 2 : 1;
    d1.js_in = st; d1.es_in = (long long)ax.B * st; d1.lst_in = lst;
    d1.js_out = d1.js_in; d1.es_out = d1.es_in; d1.lst_out = lst;
    d1.twiddle = 1; d1.N = ax.N; d1.conj_in = inverse; d1.conj_out = 0; d1.scale = 1.0;
    launch_pass(p, data, scr, d1, *ax.subA, ax.root, s);
    PassDesc d2; memset(&d2, 0, sizeof(d2));
    d2.len = ax.B; d2.J = ax.A; d2.nlines = nlines; d2.mode = lines_fastest ? 2 : 0;
    d2.js_in = (long long)ax.B * st; d2.es_in = st; d2.lst_in = lst;
    d2.js_out = st; d2.es_out = (long long)ax.A * st; d2.lst_out = lst;
    d2.twiddle = 0; d2.N = ax.N; d2.conj_in = 0; d2.conj_out = inverse; d2.scale = 1.0;
    launch_pass(p, scr, data, d2, *ax.subB, ax.root, s);
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
        hipLaunchKernelGGL(cols_c2c_4096, dim3(8 * per, nplanes), dim3(512), (2 * F4K_LDS + 8) * sizeof(cplx), s, data, p->Nh, p->Nhp,
                           p->ax0.tw, inverse, 1.0, per);
    } else {
        dim3 g2((p->Nh + p->TC - 1) / p->TC, nplanes);
        hipLaunchKernelGGL(cols_c2c, g2, dim3(p->nt_cols), p->lds_cols, s, data, p->N0, p->Nh, p->Nhp, p->TC, p->MS,
                           axis_dev(p->ax0), inverse, 1.0);
    }
}

static int forward_planes(sfft_plan* p, const RowsArgs& ra, int nplanes, cplx* dst, hipStream_t s)
{
    dim3 g1((p->N0 + 1) / 2, nplanes);
    if (p->ax1.big) {
        const int npr = (p->N0 + 1) / 2;
        for (int k = 0; k < nplanes; ++k) {
            hipLaunchKernelGGL(pack_rows, dim3((p->N1 + 255) / 256, npr), dim3(256), 0, s, ra.src[k], ra.wx[k], ra.wy[k], p->d_big1, p->N0, p->N1);
            big_axis_transform(p, p->ax1, p->d_big1, p->d_big2, 1, p->N1, npr, false, 0, s);
            hipLaunchKernelGGL(untangle_rows, dim3((p->Nh + 255) / 256, npr), dim3(256), 0, s, p->d_big1, dst + (size_t)k * p->N0 * p->Nhp,
                               p->N0, p->N1, p->Nh, p->Nhp, p->scale);
        }
    } else if (fast_axis(p->ax1) && !p->no_fast_fft) {
        RowGroups grp; grp.ngroups = 0;
        for (int k = 0; k < nplanes; ++k) {
            if (k > 0 && ra.src[k] == ra.src[k - 1]) ++grp.count[grp.ngroups - 1];
            else { grp.first[grp.ngroups] = k; grp.count[grp.ngroups] = 1; ++grp.ngroups; }
        }
        hipLaunchKernelGGL(rows_r2c_4096, dim3((p->N0 + 1) / 2, grp.ngroups), dim3(256), F4K_LDS * sizeof(cplx), s, ra, grp, dst,
                           p->N0, p->Nhp, p->ax1.tw, p->scale);
    }
    else
        hipLaunchKernelGGL(rows_r2c, g1, dim3(p->nt_rows), p->lds_rows, s, ra, dst, p->N0, p->N1, p->Nh, p->Nhp,
                           axis_dev(p->ax1), p->scale);
    LAUNCH_CHECK();
    launch_cols(p, dst, nplanes, 0, s);
    LAUNCH_CHECK();
    return SFFT_OK;
}

// forward spectra of the Fij kernel-basis planes of image d_I (and, when d_J is given, of d_J itself as plane Fij)
static int forward_basis_planes(sfft_plan* p, const double* d_I, const double* d_J, cplx* dst, hipStream_t s)
{
    const int total = p->Fij + (d_J ? 1 : 0);
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
            } else ra.src[u] = d_J;
        }
        int rc = forward_planes(p, ra, n, dst + (size_t)k0 * plane_sz, s);
        if (rc) return rc;
    }
    return SFFT_OK;
}

template <int HBW, int RS>
static void launch_g1(sfft_plan* p, int pass0, int npass, int h, hipStream_t s)
{
    dim3 g((p->Nh + 63) / 64, p->S, npass);
    const int per_launch = HBW * RS;
    for (int rb = 0; rb < h || rb == 0; rb += per_launch) {
        hipLaunchKernelGGL((greek_g1<HBW, RS>), g, dim3(64 * RS), 0, s, p->d_spec, p->d_passes, pass0, p->d_gp, p->N0, p->Nh,
                               p->Nhp, p->rows_per_chunk, rb, p->d_w0tab, p->hm, p->d_Xp);
        if (h == 0) break;
    }
}

static int greek_g1_group(sfft_plan* p, int pass0, int npass, int h, hipStream_t s)
{
    if (npass <= 0) return SFFT_OK;
    if (h <= 4) launch_g1<4, 1>(p, pass0, npass, h, s);
    else if (h <= 8) launch_g1<8, 1>(p, pass0, npass, h, s);
    else if (h <= 16) {
        if (p->g1_variant == 1) launch_g1<8, 2>(p, pass0, npass, h, s);
        else if (p->g1_variant == 2) launch_g1<4, 4>(p, pass0, npass, h, s);
        else launch_g1<16, 1>(p, pass0, npass, h, s);
    } else {
        if (p->g1_variant == 2) launch_g1<4, 4>(p, pass0, npass, h, s);
        else launch_g1<8, 2>(p, pass0, npass, h, s);
    }
    LAUNCH_CHECK();
    return SFFT_OK;
}

static int run_fill(sfft_plan* p, hipStream_t s)
{
    const int n = p->NEQfs;
    dim3 g((n + 1 + 15) / 16, (n + 1 + 15) / 16);
    hipLaunchKernelGGL(fill_system, g, dim3(256), 0, s, p->d_patches, p->d_phi, p->d_delta, p->fa, p->d_idx, n, p->NEQ,
                       p->d_A, p->ld, (double*)nullptr);
    LAUNCH_CHECK();
    return SFFT_OK;
}

static int run_cholesky(sfft_plan* p, double* d_solution, hipStream_t s)
{
    const int n = p->NEQfs;
    hipLaunchKernelGGL(chol_copy_diag, dim3(1), dim3(256), 0, s, p->d_A, p->ld, std::min(CB, n), p->d_dbuf);
    int step = 0;
    for (int k = 0; k < n; k += CB, ++step) {
        const int nb = std::min(CB, n - k);
        const int rows_below = n + 1 - (k + nb);
        const int nblk = 1 + (rows_below + CB - 1) / CB;
        double* Dcur = p->d_dbuf + (size_t)(step & 1) * CB * CB;
        double* Dnxt = p->d_dbuf + (size_t)((step + 1) & 1) * CB * CB;
        hipLaunchKernelGGL(chol_panel, dim3(nblk), dim3(256), 0, s, p->d_A, p->ld, n, k, Dcur, p->d_status, p->d_rd);
        const int ntile = (rows_below + CB - 1) / CB;
        if (ntile > 0 && k + nb < n)
            hipLaunchKernelGGL(chol_update, dim3(ntile, ntile), dim3(256), 0, s, p->d_A, p->ld, n, k, Dnxt);
    }
    LAUNCH_CHECK();
    HIPCHK(hipMemsetAsync(d_solution, 0, (size_t)p->NEQ * sizeof(double), s));
    const int nblk = (n + CB - 1) / CB;
    for (int b = nblk - 1; b >= 0; --b) {
        const int kb = b * CB, nb = std::min(CB, n - kb);
        const int rows_below = n - (kb + nb);
        const int nslice = rows_below > 0 ? std::min(BACK_SLICES, (rows_below + CB - 1) / CB) : 1;
        hipLaunchKernelGGL(chol_back_step, dim3(nslice), dim3(256), 0, s, p->d_A, p->ld, n, kb, p->d_xv, p->d_partial, p->d_counter, p->d_rd);
    }
    hipLaunchKernelGGL(scatter_solution, dim3((n + 255) / 256), dim3(256), 0, s, p->d_xv, n, p->d_idx, d_solution, p->NEQ,
                       p->fa.tie_first, p->fa.tie_cnt, p->fa.tie_stride);
    LAUNCH_CHECK();
    return SFFT_OK;
}

static int run_lu(sfft_plan* p, double* d_solution, hipStream_t s)
{
    const int n = p->NEQfs;
    for (int k = 0; k < n; ++k) {
        hipLaunchKernelGGL(lu_pivot, dim3(1), dim3(1024), 0, s, p->d_A, p->ld, n, k, p->d_status);
        const int rem = n - k - 1;
        if (rem > 0)
            hipLaunchKernelGGL(lu_rank1, dim3((rem + 1 + 63) / 64, (rem + 15) / 16), dim3(256), 0, s, p->d_A, p->ld, n, k);
    }
    LAUNCH_CHECK();
    const size_t lds = (size_t)(n + 2) * 8;
    hipLaunchKernelGGL(lu_backsolve, dim3(1), dim3(1024), lds, s, p->d_A, p->ld, n, p->d_xv);
    HIPCHK(hipMemsetAsync(d_solution, 0, (size_t)p->NEQ * sizeof(double), s));
    hipLaunchKernelGGL(scatter_solution, dim3((n + 255) / 256), dim3(256), 0, s, p->d_xv, n, p->d_idx, d_solution, p->NEQ,
                       p->fa.tie_first, p->fa.tie_cnt, p->fa.tie_stride);
    LAUNCH_CHECK();
    return SFFT_OK;
}

static int apply_prelim(sfft_plan* p, const double* d_I, cplx* dst, hipStream_t s);

extern "C" int sfft_solve(sfft_plan* p, const double* d_I, const double* d_J, double* d_solution, void* stream)
{
    if (!p || !d_I || !d_J || !d_solution) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(p->dev));
    int rc;
    {
        StageTimer t(p, SFFT_ST_PRELIM_SOLVE, s);
        if ((rc = forward_basis_planes(p, d_I, d_J, p->d_spec, s))) return rc;
        if (p->nby <= 4) hipLaunchKernelGGL(row_moments<4>, dim3(p->N0), dim3(256), 0, s, d_J, p->d_rowmom, p->N0, p->N1, p->d_tby, p->nby);
        else hipLaunchKernelGGL(row_moments<SFFT_MAX_BQ>, dim3(p->N0), dim3(256), 0, s, d_J, p->d_rowmom, p->N0, p->N1, p->d_tby, p->nby);
        hipLaunchKernelGGL(delta_finish, dim3(p->Fpq), dim3(256), 0, s, p->d_rowmom, p->d_delta, p->N0, p->bk, p->scale);
        LAUNCH_CHECK();
    }
    {
        StageTimer t(p, SFFT_ST_GREEK_G1, s);
        if ((rc = greek_g1_group(p, 0, p->n_omg, 2 * p->w, s))) return rc;
    }
    {
        StageTimer t(p, SFFT_ST_GREEK_G1B, s);
        if ((rc = greek_g1_group(p, p->n_omg, p->n_the + p->n_gamp, p->w, s))) return rc;
        if (p->n_gam0 > 0) {
            hipLaunchKernelGGL(greek_g1_row0, dim3((p->Nh + 255) / 256, p->n_gam0), dim3(256), 0, s, p->d_spec, p->d_passes,
                               p->n_omg + p->n_the + p->n_gamp, p->d_gp, p->N0, p->Nh, p->Nhp, p->S);
            LAUNCH_CHECK();
        }
    }
    {
        StageTimer t(p, SFFT_ST_GREEK_G2, s);
        hipLaunchKernelGGL(greek_g2, dim3(4 * p->w + 1, p->n_omg), dim3(256), 0, s, p->d_gp, p->d_passes, p->d_jobs, 0, p->d_patches,
                           p->Nh, p->Nhp, p->N1, p->S, p->ax1.root, p->d_Yq, p->scale);
        hipLaunchKernelGGL(greek_g2, dim3(2 * p->w + 1, p->n_gam + p->n_the), dim3(256), 0, s, p->d_gp, p->d_passes, p->d_jobs,
                           p->n_omg, p->d_patches, p->Nh, p->Nhp, p->N1, p->S, p->ax1.root, p->d_Yq, p->scale);
        LAUNCH_CHECK();
    }
    p->have_system = true;
    if (p->overlap_I) {      // sfft_subtract: start the full pair's forward transforms now, beside the dense solve
        const double* dI = p->overlap_I;
        p->overlap_I = nullptr;
        HIPCHK(hipEventRecord(p->ev_in, s));
        HIPCHK(hipStreamWaitEvent(p->s2, p->ev_in, 0));
        if ((rc = apply_prelim(p, dI, p->d_spec2, p->s2))) return rc;
        HIPCHK(hipEventRecord(p->ev_pre, p->s2));
    }
    int status = 0;
    bool use_lu = p->force_lu != 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        {
            StageTimer t(p, SFFT_ST_FILL, s);
            HIPCHK(hipMemsetAsync(p->d_status, 0, sizeof(int), s));
            if ((rc = run_fill(p, s))) return rc;
        }
        {
            StageTimer t(p, SFFT_ST_SOLVE, s);
            if (use_lu) { if ((rc = run_lu(p, p->d_sol, s))) return rc; }
            else { if ((rc = run_cholesky(p, p->d_sol, s))) return rc; }
        }
        HIPCHK(hipMemcpyAsync(&status, p->d_status, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        p->last_solver = use_lu ? 2 : 1;
        if (status == 0) break;
        if (use_lu) return set_err(SFFT_ERR_SINGULAR, "Singular matrix");
        use_lu = true;   // Cholesky met a non-positive pivot: redo with pivoted LU like the reference's gesv
    }
    HIPCHK(hipMemcpyAsync(d_solution, p->d_sol, (size_t)p->NEQ * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return SFFT_OK;
}

// forward spectra of the polynomial-weighted planes of I for the apply pass, into `dst` ([Fij][N0][Nhp])
static int apply_prelim(sfft_plan* p, const double* d_I, cplx* dst, hipStream_t s)
{
    StageTimer t(p, SFFT_ST_PRELIM_APPLY, s);
    return forward_basis_planes(p, d_I, nullptr, dst, s);
}

// Construct_FDIFF + inverse transform + DIFF epilogue from the spectra FI; FD is a scratch plane
static int apply_finish(sfft_plan* p, const cplx* FI, cplx* FD, const double* d_J, const double* d_solution, double* d_diff,
                        hipStream_t s)
{
    {
        StageTimer t(p, SFFT_ST_CONSTRUCT, s);
        hipLaunchKernelGGL(kernel_ctab, dim3((p->Nh + 255) / 256, p->Fij * p->L), dim3(256), 0, s, d_solution, p->d_ctab, p->d_soff,
                           p->Fij, p->L, p->L, p->w, p->Nh, p->Nhp, p->N1, p->ax1.root);
        hipLaunchKernelGGL(construct_fd, dim3((p->Nh + 255) / 256, (p->N0 + CRL - 1) / CRL), dim3(256), 0, s, FI, FD, p->d_ctab,
                           p->d_soff, p->ax0.root, p->N0, p->Nh, p->Nhp, p->Fij, p->L, p->w, p->scale);
        LAUNCH_CHECK();
    }
    {
        StageTimer t(p, SFFT_ST_INVERSE, s);
        launch_cols(p, FD, 1, 1, s);
        if (p->ax1.big) {
            const int npr = (p->N0 + 1) / 2;
            hipLaunchKernelGGL(retangle_rows, dim3((p->N1 + 255) / 256, npr), dim3(256), 0, s, FD, p->d_big1, p->N0, p->N1, p->Nh, p->Nhp);
            big_axis_transform(p, p->ax1, p->d_big1, p->d_big2, 1, p->N1, npr, false, 0, s);
            hipLaunchKernelGGL(finish_diff, dim3((p->N1 + 255) / 256, npr), dim3(256), 0, s, p->d_big1, d_J, d_solution + p->Fijab, p->bk,
                               d_diff, p->N0, p->N1);
        } else if (fast_axis(p->ax1) && !p->no_fast_fft)
        {
            if (p->nby <= 4)
                hipLaunchKernelGGL(rows_c2r_diff_4096<4>, dim3((p->N0 + 1) / 2), dim3(256), F4K_LDS * sizeof(cplx), s, FD, d_J,
                                   d_solution + p->Fijab, p->bk, d_diff, p->N0, p->Nhp, p->ax1.tw);
            else
                hipLaunchKernelGGL(rows_c2r_diff_4096<SFFT_MAX_BQ>, dim3((p->N0 + 1) / 2), dim3(256), F4K_LDS * sizeof(cplx), s, FD, d_J,
                                   d_solution + p->Fijab, p->bk, d_diff, p->N0, p->Nhp, p->ax1.tw);
        }
        else
            hipLaunchKernelGGL(rows_c2r_diff, dim3((p->N0 + 1) / 2), dim3(p->nt_rows), p->lds_rows, s, FD, d_J,
                               d_solution + p->Fijab, p->bk, d_diff, p->N0, p->N1, p->Nh, p->Nhp, axis_dev(p->ax1));
        LAUNCH_CHECK();
    }
    return SFFT_OK;
}

extern "C" int sfft_apply(sfft_plan* p, const double* d_I, const double* d_J, const double* d_solution, double* d_diff,
                          void* stream)
{
    if (!p || !d_I || !d_J || !d_solution || !d_diff) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(p->dev));
    int rc;
    if ((rc = apply_prelim(p, d_I, p->d_spec, s))) return rc;
    return apply_finish(p, p->d_spec, p->d_spec + (size_t)p->Fij * p->N0 * p->Nhp, d_J, d_solution, d_diff, s);
}

// GSS: the forward transforms of the full pair do not depend on the solution, and the dense solve leaves most
// CUs idle, so they run on the plan's second (low priority) stream while the caller's stream establishes and
// solves the system; the streams join before Construct_FDIFF.
extern "C" int sfft_subtract(sfft_plan* p, const double* d_I, const double* d_J, const double* d_mI, const double* d_mJ,
                             double* d_solution, double* d_diff, void* stream)
{
    if (!p || !d_I || !d_J || !d_mI || !d_mJ || !d_solution || !d_diff) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(p->dev));
    int rc;
    if (p->no_overlap) {
        if ((rc = sfft_solve(p, d_mI, d_mJ, d_solution, stream))) return rc;
        if ((rc = sfft_apply(p, d_I, d_J, d_solution, d_diff, stream))) return rc;
        HIPCHK(hipStreamSynchronize(s));
        return SFFT_OK;
    }
    if (!p->d_spec2) {
        if ((rc = dev_alloc(p, &p->d_spec2, (size_t)p->Fij * p->N0 * p->Nhp))) return rc;
    }
    if (d_I == d_mI) {
        // the caller passed the full image as its own mask ("'same' means it is identical with I",
        // SFFTSubtract.py:849): the spectra of the solve pass are the spectra of the apply pass
        if ((rc = sfft_solve(p, d_mI, d_mJ, d_solution, stream))) return rc;
        if ((rc = apply_finish(p, p->d_spec, p->d_spec + (size_t)p->Fij * p->N0 * p->Nhp, d_J, d_solution, d_diff, s))) return rc;
        HIPCHK(hipStreamSynchronize(s));
        return SFFT_OK;
    }
    p->overlap_I = d_I;
    rc = sfft_solve(p, d_mI, d_mJ, d_solution, stream);
    p->overlap_I = nullptr;
    if (rc) { hipStreamSynchronize(p->s2); return rc; }
    HIPCHK(hipStreamWaitEvent(s, p->ev_pre, 0));
    if ((rc = apply_finish(p, p->d_spec2, p->d_spec + (size_t)p->Fij * p->N0 * p->Nhp, d_J, d_solution, d_diff, s))) return rc;
    HIPCHK(hipStreamSynchronize(s));
    return SFFT_OK;
}

extern "C" int sfft_get_system(sfft_plan* p, double* d_LHMAT, double* d_RHb, void* stream)
{
    if (!p) return set_err(SFFT_ERR_INVALID_ARG, "NULL plan");
    if (!p->have_system) return set_err(SFFT_ERR_INVALID_ARG, "no linear system yet: call sfft_solve first");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(p->dev));
    dim3 g((p->NEQ + 15) / 16, (p->NEQ + 15) / 16);
    hipLaunchKernelGGL(fill_plain, g, dim3(256), 0, s, p->d_patches, p->d_phi, p->d_delta, p->fa, p->NEQ, d_LHMAT, d_RHb);
    LAUNCH_CHECK();
    HIPCHK(hipStreamSynchronize(s));
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
    HIPCHK(hipSetDevice(p->dev));
    RowsArgs ra;
    for (int k = 0; k < SFFT_MAX_PLANES; ++k) { ra.src[k] = nullptr; ra.wx[k] = nullptr; ra.wy[k] = nullptr; }
    ra.src[0] = d_real;
    int rc = forward_planes(p, ra, 1, p->d_spec, s);
    if (rc) return rc;
    hipLaunchKernelGGL(copy_spectrum_scaled, dim3((p->Nh + 255) / 256, p->N0), dim3(256), 0, s, p->d_spec, (cplx*)d_spec, p->N0, p->Nh,
                       p->Nhp, p->Nh, scale / p->scale);
    LAUNCH_CHECK();
    return SFFT_OK;
}

// d_real[N0][N1] = scale * sum_k spec[k] e^{+2 pi i k x / N} from the half spectrum of a real image (numpy.fft.irfft2 * N0*N1
// when scale = 1; pass scale = 1/(N0*N1) for numpy's normalisation)
extern "C" int sfft_ifft2_c2r(sfft_plan* p, const double* d_spec, double* d_real, double scale, void* stream)
{
    if (!p || !d_real || !d_spec) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(p->dev));
    int rc;
    if (!p->d_zero) {
        if ((rc = dev_alloc(p, &p->d_zero, (size_t)p->N0 * p->N1))) return rc;
        HIPCHK(hipMemsetAsync(p->d_zero, 0, (size_t)p->N0 * p->N1 * sizeof(double), s));
        if ((rc = dev_alloc(p, &p->d_zsol, (size_t)p->NEQ))) return rc;
        HIPCHK(hipMemsetAsync(p->d_zsol, 0, (size_t)p->NEQ * sizeof(double), s));
    }
    cplx* FD = p->d_spec + (size_t)p->Fij * p->N0 * p->Nhp;
    hipLaunchKernelGGL(copy_spectrum_scaled, dim3((p->Nh + 255) / 256, p->N0), dim3(256), 0, s, (const cplx*)d_spec, FD, p->N0, p->Nh,
                       p->Nh, p->Nhp, 1.0);
    LAUNCH_CHECK();
    // reuse the inverse path of the subtraction: with J = 0 and b = 0 it returns -IDFT2(FD)
    launch_cols(p, FD, 1, 1, s);
    if (p->ax1.big) {
        const int npr = (p->N0 + 1) / 2;
        hipLaunchKernelGGL(retangle_rows, dim3((p->N1 + 255) / 256, npr), dim3(256), 0, s, FD, p->d_big1, p->N0, p->N1, p->Nh, p->Nhp);
        big_axis_transform(p, p->ax1, p->d_big1, p->d_big2, 1, p->N1, npr, false, 0, s);
        hipLaunchKernelGGL(finish_diff, dim3((p->N1 + 255) / 256, npr), dim3(256), 0, s, p->d_big1, p->d_zero, p->d_zsol + p->Fijab, p->bk,
                           d_real, p->N0, p->N1);
    } else if (fast_axis(p->ax1) && !p->no_fast_fft)
        hipLaunchKernelGGL(rows_c2r_diff_4096<4>, dim3((p->N0 + 1) / 2), dim3(256), F4K_LDS * sizeof(cplx), s, FD, p->d_zero,
                           p->d_zsol + p->Fijab, p->bk, d_real, p->N0, p->Nhp, p->ax1.tw);
    else
        hipLaunchKernelGGL(rows_c2r_diff, dim3((p->N0 + 1) / 2), dim3(p->nt_rows), p->lds_rows, s, FD, p->d_zero,
                           p->d_zsol + p->Fijab, p->bk, d_real, p->N0, p->N1, p->Nh, p->Nhp, axis_dev(p->ax1));
    const size_t n = (size_t)p->N0 * p->N1;
    hipLaunchKernelGGL(scale_real, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_real, -scale, n);
    LAUNCH_CHECK();
    return SFFT_OK;
}

// acc[i] += coeff * |a[i]|^2 * |b[i]|^2 (b may be NULL); a, b complex128, acc float64, n elements
extern "C" int sfft_spec_abs2_accumulate(const double* d_a, const double* d_b, double coeff, double* d_acc, long long n, void* stream)
{
    if (!d_a || !d_acc || n < 0) return set_err(SFFT_ERR_INVALID_ARG, "bad argument");
    hipLaunchKernelGGL(spec_abs2_acc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const cplx*)d_a, (const cplx*)d_b,
                       coeff, d_acc, (size_t)n);
    LAUNCH_CHECK();
    return SFFT_OK;
}

// out[i] = 1 / sqrt(acc[i])
extern "C" int sfft_real_rsqrt(const double* d_acc, double* d_out, long long n, void* stream)
{
    if (!d_acc || !d_out || n < 0) return set_err(SFFT_ERR_INVALID_ARG, "bad argument");
    hipLaunchKernelGGL(real_rsqrt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_acc, d_out, (size_t)n);
    LAUNCH_CHECK();
    return SFFT_OK;
}

// out[i] = a[i] * b[i] with b complex (b_is_real = 0) or real (b_is_real = 1); a, out complex128
extern "C" int sfft_spec_multiply(const double* d_a, const double* d_b, int b_is_real, double* d_out, long long n, void* stream)
{
    if (!d_a || !d_b || !d_out || n < 0) return set_err(SFFT_ERR_INVALID_ARG, "bad argument");
    hipLaunchKernelGGL(spec_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const cplx*)d_a,
                       b_is_real ? (const cplx*)nullptr : (const cplx*)d_b, b_is_real ? d_b : (const double*)nullptr, (cplx*)d_out, (size_t)n);
    LAUNCH_CHECK();
    return SFFT_OK;
}

// full [N0][N1] of a real, conjugate-symmetric spectrum quantity from its half [N0][N1/2+1]
extern "C" int sfft_half_to_full_real(const double* d_half, double* d_full, int N0, int N1, void* stream)
{
    if (!d_half || !d_full) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    hipLaunchKernelGGL(half_to_full_real, dim3((N1 + 255) / 256, N0), dim3(256), 0, (hipStream_t)stream, d_half, d_full, N0, N1, N1 / 2 + 1);
    LAUNCH_CHECK();
    return SFFT_OK;
}

extern "C" int sfft_dbg_forward_spectrum(sfft_plan* p, const double* d_I, int i, int j, double* d_spec_out, void* stream)
{
    if (!p || !d_I || !d_spec_out) return set_err(SFFT_ERR_INVALID_ARG, "NULL argument");
    if (i < 0 || j < 0 || i >= p->nkx || j >= p->nky) return set_err(SFFT_ERR_INVALID_ARG, "basis factor index out of range for this plan");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(p->dev));
    RowsArgs ra;
    for (int k = 0; k < SFFT_MAX_PLANES; ++k) { ra.src[k] = nullptr; ra.wx[k] = nullptr; ra.wy[k] = nullptr; }
    ra.src[0] = d_I; ra.wx[0] = p->d_kbx + (size_t)i * p->N0; ra.wy[0] = p->d_kby + (size_t)j * p->N1;
    int rc = forward_planes(p, ra, 1, p->d_spec, s);
    if (rc) return rc;
    hipLaunchKernelGGL(copy_spectrum, dim3((p->Nh + 255) / 256, p->N0), dim3(256), 0, s, p->d_spec, (cplx*)d_spec_out, p->N0, p->Nh, p->Nhp);
    LAUNCH_CHECK();
    HIPCHK(hipStreamSynchronize(s));
    return SFFT_OK;
}
