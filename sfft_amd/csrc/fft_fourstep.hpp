// fft_fourstep.hpp -- four-step transforms for axes that do not fit one on-chip FFT.
// Part of libsfft_amd (MI355X / gfx950); included by sfft_amd.hip only.
#ifndef SFFT_AMD_FFT_FOURSTEP_HPP
#define SFFT_AMD_FFT_FOURSTEP_HPP

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
    const double* w;                       // optional real weight of input element (j, e): w[j * w_js + e * w_es] (the row factor of a staged
    int w_js, w_es;                        //   forward column pass: the transform reads a stage plane and applies kbx[i][row] on the way in)
};

__global__ void __launch_bounds__(SFFT_FFT_MAX_THREADS) strided_dft(const cplx* __restrict__ in, cplx* __restrict__ out, PassDesc d, AxisDev ax,
                                                     const cplx* __restrict__ rootN, int TC, int LT, int MS)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* s = reinterpret_cast<cplx*>(smem_raw);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int M = ax.rader ? ax.N : ax.M;            // tile extent per sequence (Rader transforms N points on M = N - 1)
    {   // at most 16 elements per thread (the block has >= TC * M / 16 threads): every load is issued before the first use
        cplx zz[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int x = tid + it * nt;
            int c, e;
            if (d.mode == 0) { c = x / M; e = x - c * M; } else { e = x >> LT; c = x & (TC - 1); }
            const int j = (d.mode == 2) ? (int)blockIdx.y : (int)blockIdx.x * TC + c;
            const int line = (d.mode == 2) ? (int)blockIdx.x * TC + c : (int)blockIdx.y;
            const bool ok = x < TC * M && e < d.len && j < d.J && line < d.nlines;
            zz[it] = ok ? in[(long long)line * d.lst_in + (long long)j * d.js_in + (long long)e * d.es_in] : make_double2(0.0, 0.0);
            if (d.w) { const double f = ok ? d.w[j * d.w_js + e * d.w_es] : 0.0; zz[it].x *= f; zz[it].y *= f; }
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int x = tid + it * nt;
            int c, e;
            if (d.mode == 0) { c = x / M; e = x - c * M; } else { e = x >> LT; c = x & (TC - 1); }
            if (x < TC * M) s[c * MS + e] = d.conj_in ? cconj(zz[it]) : zz[it];
        }
    }
    __syncthreads();
    lds_dft(s, ax, TC, MS);
    for (int x = tid; x < TC * M; x += nt) {
        int c, e;
        if (d.mode == 0) { c = x / M; e = x - c * M; } else { e = x >> LT; c = x & (TC - 1); }
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

// First pass of a four-step COLUMN transform whose first factor is 16 (9232 = 16 x 577): a 16-point transform per (column, j) entirely in
// registers -- lanes are 64 consecutive columns (1 KB contiguous per row and instruction), the 16 inputs of a thread are the rows
// j + B e, the four-step twiddle rootN[(j k) mod N] is wave-uniform (scalar loads).  No LDS, no barriers; replaces strided_dft's
// LDS round trip for this pass (0.55 -> see DESIGN ms per 9232 x 4609 plane).  d as for strided_dft (mode 2, len 16, twiddle 1).
__global__ void __launch_bounds__(256) strided_dft16_cols(const cplx* __restrict__ in, cplx* __restrict__ out, PassDesc d, const cplx* __restrict__ rootN)
{
    const int lane = threadIdx.x & 63;
    const int j = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6);          // wave-uniform
    const int line = (int)blockIdx.x * 64 + lane;
    if (j >= d.J) return;
    const bool ok = line < d.nlines;
    const long long li = (long long)(ok ? line : d.nlines - 1) * d.lst_in + (long long)j * d.js_in;
    cplx u[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) u[e] = in[li + (long long)e * d.es_in];
    if (d.w) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { const double f = d.w[j * d.w_js + e * d.w_es]; u[e].x *= f; u[e].y *= f; }
    }
    if (d.conj_in) {
#pragma unroll
        for (int e = 0; e < 16; ++e) u[e].y = -u[e].y;
    }
    dft16(u);
    if (!ok) return;
    const long long lo = (long long)line * d.lst_out + (long long)j * d.js_out;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        cplx z = u[R16_OUT(k)];
        if (d.twiddle) z = cmul(z, rootN[(int)(((long long)j * k) % d.N)]);
        if (d.conj_out) z.y = -z.y;
        out[lo + (long long)k * d.es_out] = make_double2(z.x * d.scale, z.y * d.scale);
    }
}

// ... and for every output plane that shares the stage plane (the staged forward column pass: outputs = stage plane x row factor
// kbx[i], round 5): the 16 strided points are read ONCE, each output applies its own row factor, transforms and writes to its own
// scratch plane.  Config 5 (orders 3 / 3): 5 stage planes feed 11 outputs -- 5 plane reads instead of 11 in this pass.
#define DFT16_MAX_OUT 4
struct Dft16Outs { int nout; const double* w[DFT16_MAX_OUT]; cplx* out[DFT16_MAX_OUT]; };
__global__ void __launch_bounds__(256) strided_dft16_cols_multi(const cplx* __restrict__ in, Dft16Outs o, PassDesc d, const cplx* __restrict__ rootN)
{
    const int lane = threadIdx.x & 63;
    const int j = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6);          // wave-uniform
    const int line = (int)blockIdx.x * 64 + lane;
    if (j >= d.J) return;
    const bool ok = line < d.nlines;
    const long long li = (long long)(ok ? line : d.nlines - 1) * d.lst_in + (long long)j * d.js_in;
    cplx u[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) u[e] = in[li + (long long)e * d.es_in];
    const long long lo = (long long)line * d.lst_out + (long long)j * d.js_out;
#pragma unroll
    for (int q = 0; q < DFT16_MAX_OUT; ++q) {
        if (q >= o.nout) break;                                            // (uniform)
        cplx v[16];
        const double* __restrict__ w = o.w[q];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const double f = w ? w[j * d.w_js + e * d.w_es] : 1.0;
            v[e] = make_double2(u[e].x * f, u[e].y * f);
        }
        dft16(v);
        if (ok) {
            cplx* __restrict__ out = o.out[q];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                cplx z = v[R16_OUT(k)];
                z = cmul(z, rootN[(int)(((long long)j * k) % d.N)]);
                out[lo + (long long)k * d.es_out] = z;
            }
        }
    }
}

// The same pass for a Rader sub-axis (N = 577, config 5's 9232 = 16 x 577 column axis), lines fastest (mode 2): a kernel of its own so that
// the compiler sees one transform, not every path of lds_dft (strided_dft is 52 k instructions and sits at its register cap).
// RADER_TC lines (columns) x 577 elements per workgroup: each row of the tile is RADER_TC x 16 contiguous bytes.
// Measured per 9232 x 4609 plane: 8 lines x 320 threads (80 KB, two workgroups per CU) 737 us; 4 lines x 192 threads (40 KB, four per
// CU) 455 us; 4 x 256 threads 480 us.  (The same pass without its two transforms: 318 us; Bluestein on 2048 points: 1378 us.)
#ifndef RADER_TC
#define RADER_TC 4
#define RADER_NT 192
#endif
#define RADER_LTC (RADER_TC == 8 ? 3 : RADER_TC == 4 ? 2 : 4)
static_assert(RADER_TC * (RADER_M + 1) <= 15 * RADER_NT && RADER_TC * RADER_M <= 16 * RADER_NT, "Rader tile: at most 15 elements per thread");
__global__ void __launch_bounds__(RADER_NT) strided_rader577(const cplx* __restrict__ in, cplx* __restrict__ out, PassDesc d, AxisDev ax, int MS)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    cplx* s = reinterpret_cast<cplx*>(smem_raw);
    constexpr int N = RADER_M + 1, TC = RADER_TC, TOT = TC * N;
    const int tid = threadIdx.x;
    const int j = (int)blockIdx.y;
    const long long base_in = (long long)j * d.js_in, base_out = (long long)j * d.js_out;
    {
        cplx zz[15];                                 // RADER_TC x 577 elements over RADER_NT threads: at most 15 each
#pragma unroll
        for (int it = 0; it < 15; ++it) {
            const int x = min(tid + it * RADER_NT, TOT - 1);
            const int e = x >> RADER_LTC, c = x & (TC - 1);
            const int line = min((int)blockIdx.x * TC + c, d.nlines - 1);
            zz[it] = in[(long long)line * d.lst_in + base_in + (long long)e * d.es_in];
        }
#pragma unroll
        for (int it = 0; it < 15; ++it) {
            const int x = tid + it * RADER_NT;
            const int e = x >> RADER_LTC, c = x & (TC - 1);
            if (x < TOT) s[c * MS + ax.rin[e]] = d.conj_in ? cconj(zz[it]) : zz[it];
        }
    }
    __syncthreads();
    lds_rader577(s, TC, MS, ax.tw, ax.bf);
#pragma unroll
    for (int it = 0; it < 15; ++it) {
        const int x = tid + it * RADER_NT;
        const int e = x >> RADER_LTC, c = x & (TC - 1);
        const int line = (int)blockIdx.x * TC + c;
        if (x < TOT && line < d.nlines) {
            cplx z = s[c * MS + ax.rout[e]];
            if (d.conj_out) z.y = -z.y;
            out[(long long)line * d.lst_out + base_out + (long long)e * d.es_out] = make_double2(z.x * d.scale, z.y * d.scale);
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
// Axes with no on-chip factorisation at all (a prime factor above the on-chip Bluestein limit: 4621, 10006 = 2 x 5003, 10007 ...):
// Bluestein's algorithm THROUGH the four-step transform -- x_n c_n zero-padded to M = 2^k >= 2 N - 1 points, an M-point four-step
// transform, the product with the transformed chirp filter, the inverse M-point transform, the product with c_k (c_k = exp(-i pi k^2 / N)).
// The three element-wise kernels move 16 x 16 tiles of (line, element) through LDS so that both the strided image side (column
// transforms: the LINE index is contiguous) and the compact work array (the element index is contiguous) see whole 256-byte pieces.
// mode 0: in -> W (chirp, optional row weight, optional conjugation, zero padding); mode 1: W -> out (chirp, optional conjugation).
// ================================================================================================
struct BlueDesc { int N, M, nlines, line0, conj, transposed; long long st, lst; const double* w; };

__global__ void __launch_bounds__(256) bigblue_move(const cplx* __restrict__ src, cplx* __restrict__ dst, BlueDesc d, const cplx* __restrict__ chirp, int mode)
{
    __shared__ cplx tile[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int k0 = blockIdx.x * 16, l0 = blockIdx.y * 16;
    const int kend = mode == 0 ? d.M : d.N;
    // image side coordinates of this thread: transposed (lines contiguous) -> tx runs over lines; else tx runs over elements
    const int li = d.transposed ? l0 + tx : l0 + ty, ki = d.transposed ? k0 + ty : k0 + tx;
    // work-array side: tx runs over elements
    const int lw = l0 + ty, kw = k0 + tx;
    if (mode == 0) {
        cplx z = make_double2(0.0, 0.0);
        if (li < d.nlines && ki < d.N) {
            z = src[(size_t)(d.line0 + li) * d.lst + (size_t)ki * d.st];
            if (d.conj) z.y = -z.y;
            if (d.w) { const double f = d.w[ki]; z.x *= f; z.y *= f; }
            const cplx c = chirp[ki];
            z = make_double2(z.x * c.x - z.y * c.y, z.x * c.y + z.y * c.x);
        }
        if (d.transposed) { tile[ty][tx] = z; __syncthreads(); z = tile[tx][ty]; }
        if (lw < d.nlines && kw < kend) dst[(size_t)lw * d.M + kw] = z;
    } else {
        cplx z = make_double2(0.0, 0.0);
        if (lw < d.nlines && kw < kend) {
            z = src[(size_t)lw * d.M + kw];
            const cplx c = chirp[kw];
            z = make_double2(z.x * c.x - z.y * c.y, z.x * c.y + z.y * c.x);
            if (d.conj) z.y = -z.y;
        }
        if (d.transposed) { tile[ty][tx] = z; __syncthreads(); z = tile[tx][ty]; }
        if (li < d.nlines && ki < kend) dst[(size_t)(d.line0 + li) * d.lst + (size_t)ki * d.st] = z;
    }
}

// W[l][k] *= bf[k]  (bf: the M-point transform of the chirp filter, divided by M)
__global__ void __launch_bounds__(256) bigblue_filter(cplx* __restrict__ W, const cplx* __restrict__ bf, int M)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= M) return;
    cplx* q = W + (size_t)blockIdx.y * M + k;
    const cplx z = *q, b = bf[k];
    *q = make_double2(z.x * b.x - z.y * b.y, z.x * b.y + z.y * b.x);
}

#endif
