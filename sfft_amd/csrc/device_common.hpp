// device_common.hpp -- complex arithmetic helpers and the on-chip (LDS) FFT core: Stockham radix-4 + Bluestein.
// Part of libsfft_amd (MI355X / gfx950); included by sfft_amd.hip only.
#ifndef SFFT_AMD_DEVICE_COMMON_HPP
#define SFFT_AMD_DEVICE_COMMON_HPP

// ------------------------------------------------------------------------------------------------
// Layout of one half-spectrum plane ([N0] rows x [Nhp] padded columns of complex128, N0 * Nhp elements either way):
//   row-major   element (l, m) at l * Nhp + m                                   (shift = 31: every column is "panel 0")
//   panels      [Nhp / PW][N0][PW]: (m / PW) * N0 * PW + l * PW + (m % PW), PW a power of two.
// The column pass of the 4096-point fast path walks one column pair down all rows: in row-major order every 32-byte
// piece it touches lies in a different 128-byte line (measured: that, not arithmetic or DRAM locality, bounds the pass);
// with PW = 2 its tile is one contiguous 128 KiB block.  Row-wise consumers see PW * 16 contiguous bytes per row.
// ------------------------------------------------------------------------------------------------
struct SpecLayout {
    int shift, mask;          // panel of column m = m >> shift, position inside it = m & mask
    int rstride;              // elements between consecutive rows of one column
    long long pstride;        // elements between consecutive panels
    __host__ __device__ __forceinline__ size_t col(int m) const { return (size_t)(m >> shift) * (size_t)pstride + (size_t)(m & mask); }
    __host__ __device__ __forceinline__ size_t at(int l, int m) const { return col(m) + (size_t)l * (size_t)rstride; }
};

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
// Streaming accesses of the transform passes (planes far larger than the L2, written once / read once): SFFT_NT bit 0 = non-temporal
// loads, bit 1 = non-temporal stores (a plain 16-byte copy gains 5 - 9 % from both on this device, scripts/micro/hbm_stream2.hip).
#ifndef SFFT_NT
#define SFFT_NT 0
#endif
// SFFT_CACHE_RESIDENT (build flag of scripts/cache_resident.sh, never of the product library): every workgroup of the 4096-point passes
// touches the memory of the same n rows / column tiles -- the same instructions, LDS exchanges and L1 requests with next to no HBM traffic;
// results are wrong by construction.  What remains of a pass is its on-chip time.
#ifdef SFFT_CACHE_RESIDENT
#define SFFT_CR(i, n) ((i) & ((n) - 1))
#else
#define SFFT_CR(i, n) (i)
#endif
typedef double v2d_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_stream(cplx* p, cplx z)
{
#if SFFT_NT & 2
    v2d_t v = {z.x, z.y};
    __builtin_nontemporal_store(v, reinterpret_cast<v2d_t*>(p));
#else
    *p = z;
#endif
}
__device__ __forceinline__ void st_stream(double* p, double z)
{
#if SFFT_NT & 2
    __builtin_nontemporal_store(z, p);
#else
    *p = z;
#endif
}
// Stores that are ALWAYS non-temporal: the spectra of cols_fwd_weighted_4096_z and the DIFF rows of rows_c2r_diff_4096 (measured per kernel, round 6:
// with every st_stream non-temporal the column pass, the Omega launch that reads its spectra and the inverse pass gain 0.014 / 0.03 / 0.006 ms,
// the row passes lose 0.007 - 0.014 ms -- so the row passes keep plain stores; non-temporal LOADS cost the column pass 0.08 ms, its two
// workgroups per panel share every line through the L2).  SFFT_NO_NT_STORES (build flag) restores plain stores for an A/B run.
__device__ __forceinline__ void st_nt(cplx* p, cplx z)
{
#ifdef SFFT_NO_NT_STORES
    *p = z;
#else
    v2d_t v = {z.x, z.y};
    __builtin_nontemporal_store(v, reinterpret_cast<v2d_t*>(p));
#endif
}
__device__ __forceinline__ void st_nt(double* p, double z)
{
#ifdef SFFT_NO_NT_STORES
    *p = z;
#else
    __builtin_nontemporal_store(z, p);
#endif
}
__device__ __forceinline__ cplx ld_stream(const cplx* p)
{
#if SFFT_NT & 1
    const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t*>(p));
    return make_double2(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ double ld_stream(const double* p)
{
#if SFFT_NT & 1
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
__device__ __forceinline__ cplx cconj(cplx a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double ipow(double x, int e) { double r = 1.0; for (int t = 0; t < e; ++t) r *= x; return r; }

#define R16_OUT(s) (4 * ((s) & 3) + ((s) >> 2))      // register holding output s of dft16()

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

// The on-chip FFT kernels run nb * M / 16 threads (one radix-16 butterfly each); with at most 9216 elements per workgroup
// that is <= 576 threads, and a 640-thread bound leaves the compiler 168 registers per lane.
#define SFFT_FFT_MAX_THREADS 640

// forward 8-point DFT in registers, outputs in natural order
__device__ __forceinline__ void dft8(cplx (&u)[8])
{
    const double h = 0.70710678118654752440;
    dft4(u[0], u[2], u[4], u[6]);          // even samples -> E[0..3] in u[0], u[2], u[4], u[6]
    dft4(u[1], u[3], u[5], u[7]);          // odd samples  -> O[0..3] in u[1], u[3], u[5], u[7]
    const cplx o1 = cmul(u[3], make_double2(h, -h));
    const cplx o2 = make_double2(u[5].y, -u[5].x);
    const cplx o3 = cmul(u[7], make_double2(-h, -h));
    const cplx e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6], o0 = u[1];
    u[0] = cadd(e0, o0); u[4] = csub(e0, o0);
    u[1] = cadd(e1, o1); u[5] = csub(e1, o1);
    u[2] = cadd(e2, o2); u[6] = csub(e2, o2);
    u[3] = cadd(e3, o3); u[7] = csub(e3, o3);
}

// Radix-16 Stockham FFT for M = 2^logM >= 16: one leading radix-2/4/8 stage when logM is not a multiple of 4 (no twiddles),
// then radix-16 stages: half the passes over LDS and half the barriers of the radix-4 version below, and the 15 twiddles of
// a butterfly come from four table entries.  Same contract (nb * M <= 16 * blockDim.x, every thread calls).  Between the
// stages the sequence lives in a padded layout (element i at i + i / 16, so that the stride-16 writes of the first radix-16
// stage spread over the banks): each sequence needs M + M / 16 elements of LDS; input and result are in natural order.
// PRE: input element n is multiplied by pre[n] on the way in (n < npre; the rest of the sequence is zero) -- Bluestein's chirp
template <int R, bool PRE>
__device__ __forceinline__ void lds_stage_small(cplx* s, int M, int nb, int stride, const cplx* __restrict__ pre, int npre)
{
    constexpr int IT = 16 / R;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int T = M / R, total = nb * T;
    cplx y[IT][R];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int g = tid + it * nt;
        if (g < total) {
            const int f = g / T, i = g - f * T;
            const cplx* b = s + f * stride;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                y[it][r] = b[i + r * T];
                if (PRE) y[it][r] = cmul(y[it][r], pre[min(i + r * T, npre - 1)]);      // (entries >= npre are zero)
            }
            if constexpr (R == 2) { const cplx a = y[it][0], c = y[it][1]; y[it][0] = cadd(a, c); y[it][1] = csub(a, c); }
            else if constexpr (R == 4) dft4(y[it][0], y[it][1], y[it][2], y[it][3]);
            else dft8(y[it]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int g = tid + it * nt;
        if (g < total) {
            const int f = g / T, i = g - f * T;
            cplx* b = s + f * stride;
#pragma unroll
            for (int r = 0; r < R; ++r) b[pad16(i * R + r)] = y[it][r];
        }
    }
    __syncthreads();
}

// one radix-3 stage on the padded layout (natural on the way out when `last`); p = product of the earlier radices
template <int POST>
__device__ __forceinline__ void lds_stage3_padded(cplx* s, int M, int p, int nb, int stride, const cplx* __restrict__ tw, bool padded_in, bool last,
                                                  const cplx* __restrict__ post, int npost)
{
    constexpr int IT = 6;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int T = M / 3, total = nb * T, step = T / p;
    const double S3 = 0.86602540378443864676;      // sin(pi/3)
    cplx y[IT][3];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int g = tid + it * nt;
        if (g < total) {
            const int f = g / T, i = g - f * T;
            const int k = i % p;
            const cplx* b = s + f * stride;
            cplx u0, u1, u2;
            if (padded_in) { u0 = b[pad16(i)]; u1 = b[pad16(i + T)]; u2 = b[pad16(i + 2 * T)]; }
            else { u0 = b[i]; u1 = b[i + T]; u2 = b[i + 2 * T]; }
            if (p > 1) {
                const int q = k * step;
                u1 = cmul(u1, tw[q]);
                u2 = cmul(u2, tw[2 * q]);
            }
            const cplx t1 = cadd(u1, u2), dd = csub(u1, u2);
            const cplx t2 = make_double2(u0.x - 0.5 * t1.x, u0.y - 0.5 * t1.y);
            y[it][0] = cadd(u0, t1);
            y[it][1] = make_double2(t2.x + S3 * dd.y, t2.y - S3 * dd.x);
            y[it][2] = make_double2(t2.x - S3 * dd.y, t2.y + S3 * dd.x);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int g = tid + it * nt;
        if (g < total) {
            const int f = g / T, i = g - f * T;
            const int k = i % p;
            cplx* b = s + f * stride;
            const int o = (i - k) * 3 + k;
            if (last) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    cplx v = y[it][r];
                    const int kk = o + r * p;
                    if (POST == 1) v = cconj(cmul(v, post[kk]));
                    if (POST == 2) v = cmul(post[min(kk, npost - 1)], cconj(v));
                    b[kk] = v;
                }
            }
            else { b[pad16(o)] = y[it][0]; b[pad16(o + p)] = y[it][1]; b[pad16(o + 2 * p)] = y[it][2]; }
        }
    }
    __syncthreads();
}

// M = 2^log2p * 3^n3 (log2p >= 4 or n3 >= 1): leading radix-2/4/8 stage, radix-16 stages, then the radix-3 stages.
// POST (applied to output k of the last stage): 1: conj(X[k] * post[k]) -- the Bluestein filter; the next
// forward transform then acts as the inverse one;  2: post[k] * conj(X[k]) for k < npost -- the closing chirp
template <bool PRE, int POST>
__device__ __forceinline__ void lds_fft_r16(cplx* s, int M, int log2p, int n3, int nb, int stride, const cplx* __restrict__ tw,
                                            const cplx* __restrict__ pre, int npre, const cplx* __restrict__ post, int npost)
{
    const int tid = threadIdx.x;
    const int rem = log2p & 3;
    if (rem == 1) lds_stage_small<2, PRE>(s, M, nb, stride, pre, npre);
    else if (rem == 2) lds_stage_small<4, PRE>(s, M, nb, stride, pre, npre);
    else if (rem == 3) lds_stage_small<8, PRE>(s, M, nb, stride, pre, npre);
    bool padded = rem != 0;
    const int T = M >> 4, total = nb * T;
    const bool mine = tid < total;
    const int f = mine ? tid / T : 0, i = tid - f * T;
    cplx* b = s + f * stride;
    const int tstep0 = M >> 4;                       // twiddle step of a radix-16 stage is M / (16 p)
    for (int logp = rem; logp + 4 <= log2p; logp += 4) {
        const int p = 1 << logp;
        const int k = i & (p - 1);
        const bool last = (logp + 8 > log2p) && n3 == 0;
        cplx u[16];
        if (mine) {
            if (padded) {
#pragma unroll
                for (int r = 0; r < 16; ++r) u[r] = b[pad16(i + r * T)];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    u[r] = b[i + r * T];
                    if (PRE) u[r] = cmul(u[r], pre[min(i + r * T, npre - 1)]);
                }
            }
            if (logp > 0) twiddle16(u, tw, k * (tstep0 >> logp));
            dft16(u);
        }
        __syncthreads();
        if (mine) {
            const int o = ((i - k) << 4) + k;
            if (last) {
#pragma unroll
                for (int sx = 0; sx < 16; ++sx) {
                    cplx v = u[R16_OUT(sx)];
                    const int kk = o + sx * p;
                    if (POST == 1) v = cconj(cmul(v, post[kk]));
                    if (POST == 2) v = cmul(post[min(kk, npost - 1)], cconj(v));
                    b[kk] = v;
                }
            } else {
#pragma unroll
                for (int sx = 0; sx < 16; ++sx) b[pad16(o + sx * p)] = u[R16_OUT(sx)];
            }
        }
        __syncthreads();
        padded = true;
    }
    int p3 = 1 << log2p;
    for (int l = 0; l < n3; ++l, p3 *= 3) {
        lds_stage3_padded<POST>(s, M, p3, nb, stride, tw, padded, l == n3 - 1, post, npost);
        padded = true;
    }
}

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

// Mixed-radix variant for M = 2^log2p * 3^n3 (image axes such as 6144 or 9216): the same two-phase Stockham stages
// with general index arithmetic, radix-2/4 stages first and the radix-3 stages last.  A thread owns at most
// ceil(16/R) butterflies of a radix-R stage (same nb*M <= 16*blockDim.x contract as lds_fft).
template <int R>
__device__ __forceinline__ void lds_stage_mixed(cplx* s, int M, int p, int nb, int stride, const cplx* __restrict__ tw)
{
    constexpr int IT = (R == 2) ? 8 : (R == 4) ? 4 : 6;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int T = M / R, total = nb * T, step = T / p;       // twiddle step M/(R p)
    cplx y[IT][R];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int g = tid + it * nt;
        if (g < total) {
            const int f = g / T, i = g - f * T;
            const int k = i % p;
            const cplx* b = s + f * stride;
            cplx u[R];
#pragma unroll
            for (int r = 0; r < R; ++r) u[r] = b[i + r * T];
            if (p > 1) {
                const int q = k * step;
#pragma unroll
                for (int r = 1; r < R; ++r) u[r] = cmul(u[r], tw[q * r]);
            }
            if constexpr (R == 2) {
                y[it][0] = cadd(u[0], u[1]);
                y[it][1] = csub(u[0], u[1]);
            } else if constexpr (R == 3) {
                const double S3 = 0.86602540378443864676;      // sin(pi/3)
                const cplx t1 = cadd(u[1], u[2]), d = csub(u[1], u[2]);
                const cplx t2 = make_double2(u[0].x - 0.5 * t1.x, u[0].y - 0.5 * t1.y);
                y[it][0] = cadd(u[0], t1);
                // -i*S3*d = (S3*d.y, -S3*d.x)
                y[it][1] = make_double2(t2.x + S3 * d.y, t2.y - S3 * d.x);
                y[it][2] = make_double2(t2.x - S3 * d.y, t2.y + S3 * d.x);
            } else {
                const cplx a02 = cadd(u[0], u[2]), s02 = csub(u[0], u[2]);
                const cplx a13 = cadd(u[1], u[3]), s13 = csub(u[1], u[3]);
                y[it][0] = cadd(a02, a13);
                y[it][2] = csub(a02, a13);
                y[it][1] = make_double2(s02.x + s13.y, s02.y - s13.x);
                y[it][3] = make_double2(s02.x - s13.y, s02.y + s13.x);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int g = tid + it * nt;
        if (g < total) {
            const int f = g / T, i = g - f * T;
            const int k = i % p;
            cplx* b = s + f * stride + ((i - k) * R + k);
#pragma unroll
            for (int r = 0; r < R; ++r) b[r * p] = y[it][r];
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void lds_fft_mixed(cplx* s, int M, int log2p, int n3, int nb, int stride, const cplx* __restrict__ tw)
{
    int p = 1;
    if (log2p & 1) { lds_stage_mixed<2>(s, M, p, nb, stride, tw); p = 2; }
    for (int l = log2p & 1; l < log2p; l += 2) { lds_stage_mixed<4>(s, M, p, nb, stride, tw); p *= 4; }
    for (int l = 0; l < n3; ++l) { lds_stage_mixed<3>(s, M, p, nb, stride, tw); p *= 3; }
}

// ------------------------------------------------------------------------------------------------------------------------
// Prime-length sub-transform of a four-step axis by Rader's algorithm (config 5's 9232 = 16 x 577 column axis): with g a primitive
// root mod N, a[r] = x[g^r] and b[t] = W_N^(g^-t), X[g^-q] = x[0] + (a (*) b)[q] -- a cyclic convolution of length N - 1 = 576 =
// 16 * 4 * 9 -- and X[0] = x[0] + sum a.  Two 576-point transforms instead of Bluestein's two 2048-point ones, a third of the LDS
// per sequence (eight sequences per workgroup instead of two: 128 contiguous bytes per row instead of 32).
// The 576-point transform is a Stockham autosort with COMPILE-TIME radices (16, 4, 9), so that the index arithmetic of every stage
// divides by constants; intermediate stages use the padded layout of lds_fft_r16, first read and last write are in natural order.
// POST on the last stage's output k: 1: conj(X[k] * post[k]) (the next forward transform then inverts), 3: conj(X[k]).
// ------------------------------------------------------------------------------------------------------------------------
// forward 9-point DFT, outputs in natural order
__device__ __forceinline__ void dft9(const cplx (&x)[9], cplx (&X)[9])
{
    const double S3 = 0.86602540378443864676;
    const cplx w1 = make_double2(0.76604444311897803520, -0.64278760968653932632);      // W9^1
    const cplx w2 = make_double2(0.17364817766693034885, -0.98480775301220805937);      // W9^2
    const cplx w4 = make_double2(-0.93969262078590838405, -0.34202014332566873304);     // W9^4
    cplx t[3][3];                                        // t[n2][k1] = DFT3 over n1 of x[3 n1 + n2]
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) {
        const cplx u0 = x[n2], u1 = x[3 + n2], u2 = x[6 + n2];
        const cplx a = cadd(u1, u2), d = csub(u1, u2);
        const cplx m = make_double2(u0.x - 0.5 * a.x, u0.y - 0.5 * a.y);
        t[n2][0] = cadd(u0, a);
        t[n2][1] = make_double2(m.x + S3 * d.y, m.y - S3 * d.x);
        t[n2][2] = make_double2(m.x - S3 * d.y, m.y + S3 * d.x);
    }
    t[1][1] = cmul(t[1][1], w1); t[1][2] = cmul(t[1][2], w2);
    t[2][1] = cmul(t[2][1], w2); t[2][2] = cmul(t[2][2], w4);
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {                     // X[k1 + 3 k2] = DFT3 over n2 of t[n2][k1]
        const cplx u0 = t[0][k1], u1 = t[1][k1], u2 = t[2][k1];
        const cplx a = cadd(u1, u2), d = csub(u1, u2);
        const cplx m = make_double2(u0.x - 0.5 * a.x, u0.y - 0.5 * a.y);
        X[k1] = cadd(u0, a);
        X[k1 + 3] = make_double2(m.x + S3 * d.y, m.y - S3 * d.x);
        X[k1 + 6] = make_double2(m.x - S3 * d.y, m.y + S3 * d.x);
    }
}

// one Stockham stage of radix R on nb sequences of length M; P = product of the earlier radices.  nb * M <= 16 * blockDim.x.
template <int M, int R, int P, bool FIRST, bool LAST, int POST>
__device__ __forceinline__ void ct_stage(cplx* s, int nb, int stride, const cplx* __restrict__ tw, const cplx* __restrict__ post)
{
    constexpr int T = M / R, IT = (16 + R - 1) / R, STEP = T / P;
    const int tid = threadIdx.x, nt = blockDim.x, total = nb * T;
    // nb is a power of two and the SEQUENCE index runs fastest over the threads: the nb lanes that work on the same butterfly of
    // different sequences read the same twiddle / filter entries (one cache line per nb lanes instead of one per lane)
    const int lnb = __builtin_ctz(nb);
    cplx y[IT][R];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int g = tid + it * nt;
        if (g < total) {
            const int f = g & (nb - 1), i = g >> lnb, k = i % P;
            const cplx* b = s + f * stride;
            cplx u[R];
#pragma unroll
            for (int r = 0; r < R; ++r) u[r] = FIRST ? b[i + r * T] : b[pad16(i + r * T)];
            if (P > 1) {
                const int q = k * STEP;
#pragma unroll
                for (int r = 1; r < R; ++r) u[r] = cmul(u[r], tw[q * r]);
            }
            if constexpr (R == 16) {
                cplx v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = u[r];
                dft16(v);
#pragma unroll
                for (int r = 0; r < 16; ++r) y[it][r] = v[R16_OUT(r)];
            } else if constexpr (R == 4) {
                dft4(u[0], u[1], u[2], u[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) y[it][r] = u[r];
            } else {
                dft9(u, y[it]);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int g = tid + it * nt;
        if (g < total) {
            const int f = g & (nb - 1), i = g >> lnb, k = i % P;
            cplx* b = s + f * stride;
            const int o = (i - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                cplx v = y[it][r];
                const int kk = o + r * P;
                if (LAST) {
                    if (POST == 1) v = cconj(cmul(v, post[kk]));
                    if (POST == 3) v = cconj(v);
                    b[kk] = v;
                } else b[pad16(kk)] = v;
            }
        }
    }
    __syncthreads();
}

template <int POST>
__device__ __forceinline__ void ct_fft576(cplx* s, int nb, int stride, const cplx* __restrict__ tw, const cplx* __restrict__ post)
{
    ct_stage<576, 16, 1, true, false, 0>(s, nb, stride, tw, nullptr);
    ct_stage<576, 4, 16, false, false, 0>(s, nb, stride, tw, nullptr);
    ct_stage<576, 9, 64, false, true, POST>(s, nb, stride, tw, post);
}

#define RADER_M 576
#define RADER_XS (RADER_M + RADER_M / 16)            // slot of x[0] / X[0] behind the padded transform region: a sequence needs RADER_XS + 1 elements
// 577-point DFT of nb sequences (nb a power of two) by Rader's algorithm, in LDS.  The caller's load phase has already put the
// sequence in Rader order -- a[r] = x[g^r] at s[r], r < 576, and x[0] at s[RADER_XS] (it stores element n at s[rin[n]]) -- and its store
// phase reads output n from s[rout[n]]: c[q] = X[g^-q] at s[q], X[0] at s[RADER_XS].  No permutation passes over LDS.
// bf = FFT_576(b) / 576 with b[t] = W_N^(g^-t) and bf[0] = -1 / 576 exactly.
__device__ __forceinline__ void lds_rader577(cplx* s, int nb, int stride, const cplx* __restrict__ tw, const cplx* __restrict__ bf)
{
    constexpr int M = RADER_M;
    const int tid = threadIdx.x;
    ct_fft576<1>(s, nb, stride, tw, bf);             // s[k] = conj(A[k] bf[k])
    if (tid < nb) {
        const cplx x0 = s[tid * stride + RADER_XS];
        const cplx t = s[tid * stride];              // conj(A[0] bf[0]) = -conj(A[0]) / M
        s[tid * stride + RADER_XS] = make_double2(x0.x - (double)M * t.x, x0.y + (double)M * t.y);       // X[0] = x[0] + A[0]
        s[tid * stride] = make_double2(t.x + x0.x, t.y - x0.y);                                          // + x[0] on every c[q]
    }
    __syncthreads();
    ct_fft576<3>(s, nb, stride, tw, nullptr);        // s[q] = x[0] + (a (*) b)[q]
}

// One 1-D axis: length N transformed either directly (N = M power of two) or by Bluestein's chirp-z
// (M = power of two >= 2N-1), or directly with N = M = 2^logM * 3^n3.  All tables live in device memory.
struct AxisDev {
    int N, M, logM, blue, n3, r16;     // r16: power-of-two M >= 16 on the radix-16 stages (sequence needs M + M/16 LDS elements)
    int rader;                         // 1: N = 577 by Rader's algorithm (M = 576; bf, rin, rout below)
    const int* rin;     // [N]   LDS slot of input element n: r with g^r = n (mod N); RADER_XS for n = 0
    const int* rout;    // [N]   LDS slot of output element n: q with g^-q = n (mod N); RADER_XS for n = 0
    const cplx* tw;     // [M]   exp(-2 pi i k / M)
    const cplx* chirp;  // [N]   exp(-i pi n^2 / N)            (Bluestein only)
    const cplx* bf;     // [M]   FFT_M(conj-chirp filter) / M  (Bluestein only)
    const cplx* root;   // [N]   exp(-2 pi i k / N)
};

__device__ __forceinline__ void lds_fft_pow2(cplx* s, const AxisDev& ax, int nb, int stride)
{
    if (ax.r16) lds_fft_r16<false, 0>(s, ax.M, ax.logM, 0, nb, stride, ax.tw, nullptr, 0, nullptr, 0);
    else lds_fft(s, ax.M, ax.logM, nb, stride, ax.tw);
}

// forward length-N DFT of nb sequences already resident in LDS (entries n >= N must be zero when blue).
// On return entries [0, N) of each sequence hold the DFT.
__device__ __forceinline__ void lds_dft(cplx* s, const AxisDev& ax, int nb, int stride)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    // (Rader sub-axes never come through here: strided_rader577 loads and stores in Rader order itself)
    if (ax.n3 && !ax.blue) {
        if (ax.r16) lds_fft_r16<false, 0>(s, ax.M, ax.logM, ax.n3, nb, stride, ax.tw, nullptr, 0, nullptr, 0);
        else lds_fft_mixed(s, ax.M, ax.logM, ax.n3, nb, stride, ax.tw);
        return;
    }
    if (!ax.blue) { lds_fft_pow2(s, ax, nb, stride); return; }
    if (ax.r16) {
        // Bluestein with the three pointwise products folded into the transforms' first read / last write:
        //   a = x * chirp;  A = FFT(a);  c = conj(A * Bf);  C = FFT(c);  X[k] = chirp[k] * conj(C[k])
        lds_fft_r16<true, 1>(s, ax.M, ax.logM, ax.n3, nb, stride, ax.tw, ax.chirp, ax.N, ax.bf, ax.M);
        lds_fft_r16<false, 2>(s, ax.M, ax.logM, ax.n3, nb, stride, ax.tw, nullptr, 0, ax.chirp, ax.N);
        return;
    }
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

#endif
