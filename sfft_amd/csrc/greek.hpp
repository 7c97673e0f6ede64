// greek.hpp -- Greek stage: pruned column/row transforms of the Hadamard products, Delta moments.
// Part of libsfft_amd (MI355X / gfx950); included by sfft_amd.hip only.
#ifndef SFFT_AMD_GREEK_HPP
#define SFFT_AMD_GREEK_HPP

// ------------------------------------------------------------------------------------------------
// Greek stage 1: for every listed pass (A, B) and column m of the half spectrum
//      G[r][m] = sum_l A[l][m] * conj(B[l][m]) * W0^(l r),   |r| <= h          (pruned column DFT of the
// Hadamard product; only these lags are ever read by FillLS_*, SFFTConfigure.py:251-269, 364-371, 621-628).
// r and -r share their four real products.  B is a stored plane (Omega, Theta) or, for Gamma, the column
// factor Xp[l] = DFT(cx^p)[l] of the rank-1 spectrum FT_pq = SCALE * Xp (x) Yq -- the row factor Yq[m] does not
// depend on l and is applied in stage 2, so one pass serves every q.
// ------------------------------------------------------------------------------------------------
struct G1Pass {
    int a_plane;      // plane index into spec
    int b_plane;      // plane index, or -1: B[l][m] = Xp[bp][l]
    int bp;
    int h;            // lag half width
    long long gp_off; // offset (cplx) of this pass's [S][2h+1][Nhp] partial buffer
    long long gp_off2;// dual diagonal pass (b_plane >= 0, dual = 1): partial buffer of the second plane's self-product
    int dual;         // 1: this pass is |A|^2 and |B|^2 side by side (two diagonal Omega passes in one wave)
};

struct PatchJob {
    int pass;         // G1 pass that produced G
    int yq;           // -1, or q: G[r][m] is multiplied by conj(tscale * Yq[q][m])
    int h;
    int patch_off;    // offset (doubles) of this job's [(2h+1)][(2h+1)] patch
    double scale;
};

// One wave per workgroup owns 64 columns x one row chunk x ALL lags of the launch (HBW of them, 4 real FMAs per lag and
// row), so no two waves load the same element.  Every wave issues the loads of U rows before it consumes any of them
// (the first version of this kernel waited on two loads per row and was bound by ~1 us of loaded memory latency per
// iteration, not by bandwidth); the body is branch free -- lags beyond h are computed on padded twiddle columns and
// simply not stored -- and the HBW twiddles of a row are one scalar load.
template <int HBW>
__device__ __forceinline__ void g1_row(const cplx av, const cplx bv, const cplx* __restrict__ trow, double (&S1)[HBW],
                                       double (&S2)[HBW], double (&S3)[HBW], double (&S4)[HBW], double& g0x, double& g0y)
{
    const cplx H = cmulc(av, bv);
    g0x += H.x; g0y += H.y;
#pragma unroll
    for (int t = 0; t < HBW; ++t) {
        const cplx w = trow[t];
        S1[t] = fma(H.x, w.x, S1[t]);
        S2[t] = fma(H.y, w.y, S2[t]);
        S3[t] = fma(H.x, w.y, S3[t]);
        S4[t] = fma(H.y, w.x, S4[t]);
    }
}

// The same for a row pair (x', x' + N0 / 2) after one radix-2 decimation step: the twiddle of lag r at the partner row is (-1)^r times
// the one at row x', so even lags take the sum of the two products and odd lags their difference, and the lag sums run over half
// the rows (see greek_g1_mfma4g<.., true>).  The band starts at an odd lag (r_base is even): t even -> odd lag.
template <int HBW>
__device__ __forceinline__ void g1_row_dit(const cplx av, const cplx bv, const cplx av2, const cplx bv2, const cplx* __restrict__ trow,
                                           double (&S1)[HBW], double (&S2)[HBW], double (&S3)[HBW], double (&S4)[HBW], double& g0x, double& g0y)
{
    const cplx H = cmulc(av, bv), H2 = cmulc(av2, bv2);
    const cplx Ye = make_double2(H.x + H2.x, H.y + H2.y), Yo = make_double2(H.x - H2.x, H.y - H2.y);
    g0x += Ye.x; g0y += Ye.y;
#pragma unroll
    for (int t = 0; t < HBW; ++t) {
        const cplx w = trow[t];
        const cplx Y = (t & 1) ? Ye : Yo;
        S1[t] = fma(Y.x, w.x, S1[t]);
        S2[t] = fma(Y.y, w.y, S2[t]);
        S3[t] = fma(Y.x, w.y, S3[t]);
        S4[t] = fma(Y.y, w.x, S4[t]);
    }
}

template <int HBW, int U, bool DIT = false>
__global__ void __launch_bounds__(64) greek_g1(const cplx* __restrict__ spec, const G1Pass* __restrict__ passes, int pass0,
                                               cplx* __restrict__ Gp, int N0, int Nh, int Nhp, SpecLayout lay, int rows_per_chunk,
                                               int r_base, const cplx* __restrict__ W0tab, int HM, const cplx* __restrict__ Xp,
                                               int ncb, int S, int npass)
{
    const int lane = threadIdx.x;
    // 1-D grid over (tile = column block x row chunk, pass).  Workgroup ids go round-robin over the 8 XCDs, so XCD x is
    // given the contiguous range [x * per, (x + 1) * per) of the logical order "pass fastest": all passes of one tile run
    // on one XCD at about the same time and share that tile's slice of every plane through its L2.
    const int total = ncb * S * npass;
    const int per = (total + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (logical >= total) return;
    const int tile = logical / npass;
    const int chunk = tile / ncb;
    const int m = (tile - chunk * ncb) * 64 + lane;
    const G1Pass pr = passes[pass0 + (logical - tile * npass)];
    const int h = pr.h;
    const int PH = 2 * h + 1;
    const int lb = chunk * rows_per_chunk;              // (DIT: rows_per_chunk counts rows x' of the first half; their partners are x' + N0 / 2)
    const int le = min(DIT ? N0 / 2 : N0, lb + rows_per_chunk);
    const bool active = m < Nh;
    const int mc = active ? m : 0;
    const size_t plane_sz = (size_t)N0 * Nhp;
    const size_t mo = lay.col(mc);
    const size_t rs = (size_t)lay.rstride;
    const cplx* __restrict__ A = spec + (size_t)pr.a_plane * plane_sz + mo;
    const bool colfac = pr.b_plane < 0;
    const int rfirst = r_base + 1;                  // lags rfirst .. rfirst + HBW - 1 (those beyond h are not stored)
    double S1[HBW], S2[HBW], S3[HBW], S4[HBW];
#pragma unroll
    for (int t = 0; t < HBW; ++t) { S1[t] = S2[t] = S3[t] = S4[t] = 0.0; }
    double g0x = 0.0, g0y = 0.0;
    // W0tab[l][r] = W0^(l r): one contiguous, wave-uniform row of twiddles per image row
    const cplx* __restrict__ trow = W0tab + (size_t)lb * HM + rfirst;
    int l = lb;
    if (DIT) {
        const size_t half = (size_t)(N0 / 2);
        const cplx* __restrict__ A2 = A + half * rs;
        if (!colfac) {
            const cplx* __restrict__ B = spec + (size_t)pr.b_plane * plane_sz + mo;
            const cplx* __restrict__ B2 = B + half * rs;
            for (; l + U <= le; l += U, trow += (size_t)U * HM) {
                cplx av[U], bv[U], av2[U], bv2[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { av[u] = A[(size_t)(l + u) * rs]; av2[u] = A2[(size_t)(l + u) * rs]; }
#pragma unroll
                for (int u = 0; u < U; ++u) { bv[u] = B[(size_t)(l + u) * rs]; bv2[u] = B2[(size_t)(l + u) * rs]; }
#pragma unroll
                for (int u = 0; u < U; ++u) g1_row_dit<HBW>(av[u], bv[u], av2[u], bv2[u], trow + (size_t)u * HM, S1, S2, S3, S4, g0x, g0y);
            }
            for (; l < le; ++l, trow += HM)
                g1_row_dit<HBW>(A[(size_t)l * rs], B[(size_t)l * rs], A2[(size_t)l * rs], B2[(size_t)l * rs], trow, S1, S2, S3, S4, g0x, g0y);
        } else {
            const cplx* __restrict__ xp = Xp + (size_t)pr.bp * N0;
            for (; l + U <= le; l += U, trow += (size_t)U * HM) {
                cplx av[U], av2[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { av[u] = A[(size_t)(l + u) * rs]; av2[u] = A2[(size_t)(l + u) * rs]; }
#pragma unroll
                for (int u = 0; u < U; ++u) g1_row_dit<HBW>(av[u], xp[l + u], av2[u], xp[l + u + half], trow + (size_t)u * HM, S1, S2, S3, S4, g0x, g0y);
            }
            for (; l < le; ++l, trow += HM)
                g1_row_dit<HBW>(A[(size_t)l * rs], xp[l], A2[(size_t)l * rs], xp[l + half], trow, S1, S2, S3, S4, g0x, g0y);
        }
    } else
    if (!colfac) {
        const cplx* __restrict__ B = spec + (size_t)pr.b_plane * plane_sz + mo;
        for (; l + U <= le; l += U, trow += (size_t)U * HM) {
            cplx av[U], bv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) av[u] = A[(size_t)(l + u) * rs];
#pragma unroll
            for (int u = 0; u < U; ++u) bv[u] = B[(size_t)(l + u) * rs];
#pragma unroll
            for (int u = 0; u < U; ++u) g1_row<HBW>(av[u], bv[u], trow + (size_t)u * HM, S1, S2, S3, S4, g0x, g0y);
        }
        for (; l < le; ++l, trow += HM) g1_row<HBW>(A[(size_t)l * rs], B[(size_t)l * rs], trow, S1, S2, S3, S4, g0x, g0y);
    } else {
        const cplx* __restrict__ xp = Xp + (size_t)pr.bp * N0;     // wave-uniform column factor
        for (; l + U <= le; l += U, trow += (size_t)U * HM) {
            cplx av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) av[u] = A[(size_t)(l + u) * rs];
#pragma unroll
            for (int u = 0; u < U; ++u) g1_row<HBW>(av[u], xp[l + u], trow + (size_t)u * HM, S1, S2, S3, S4, g0x, g0y);
        }
        for (; l < le; ++l, trow += HM) g1_row<HBW>(A[(size_t)l * rs], xp[l], trow, S1, S2, S3, S4, g0x, g0y);
    }
    if (!active) return;
    cplx* g = Gp + pr.gp_off + (size_t)chunk * PH * Nhp + m;
    if (r_base == 0) g[(size_t)h * Nhp] = make_double2(g0x, g0y);
#pragma unroll
    for (int t = 0; t < HBW; ++t) {
        const int r = rfirst + t;
        if (r <= h) {
            g[(size_t)(h + r) * Nhp] = make_double2(S1[t] - S2[t], S3[t] + S4[t]);
            g[(size_t)(h - r) * Nhp] = make_double2(S1[t] + S2[t], S4[t] - S3[t]);
        }
    }
}

// The same pass on the matrix cores, for plane x plane passes with at most 16 lags (the Omega passes at KerHW <= 8).
// Per step a wave takes 4 image rows x 64 columns as four 4 x 16 tiles.  v_mfma_f64_16x16x4_f64 computes D[16 x 16] += A[16 x 4]
// B[4 x 16] with one double per lane for A and B (lane = 16 k + i holds A[i][k]; lane = 16 k + j holds B[k][j]) and four per lane
// for D (column j = lane & 15, rows (lane >> 4) + 4 q).  With A = the twiddles (i = lag - 1, k = row) and B = the Hadamard
// product H (k = row, j = column), the four real products of the lag pair +-r are four MFMAs per tile:
//   S1 = wx Hx, S2 = wy Hy, S3 = wy Hx, S4 = wx Hy.
// The loads put H directly in operand layout (lane = 16 * row + column), the twiddles are ONE vector load per step (no
// scalar-register bottleneck), and the 64 accumulators per lane of the vector version become 16 four-vectors.
typedef double d4v __attribute__((ext_vector_type(4)));

// The same passes on the OTHER fp64 matrix instruction.  A loop of independent v_mfma_f64_16x16x4_f64 sustains 47 TFLOP/s on MI355X,
// one of v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 blocks per instruction) 71 - 74.5 (scripts/micro/mfma_f64_peak.hip,
// profiles/r02_mfma_f64_peak.txt).  Its layout (scripts/micro/mfma_f64_4x4_layout.hip): A_blk[i][k] in lane 16 k + 4 blk + i,
// B_blk[k][j] in lane 16 k + 4 blk + j, D_blk[i][j] in lane 16 i + 4 blk + j.  With k = image row of the step and the four blocks =
// the four 4-column groups of a 16-column tile, B is the Hadamard product in exactly the lane layout the loads already produce
// (lane = 16 row + column), A is the twiddle of lag 4 g + i (the same in all four blocks: gathered from the one twiddle per lane
// the loads bring, by ds_swizzle within the 16-lane row), and the lane (i, column) of D holds lag 4 g + i: four instructions (lag groups) per real product where the
// 16 x 16 x 4 form has one, at a third of the cycles each.
template <int NT>
__global__ void __launch_bounds__(64, 3) greek_g1_mfma4(const cplx* __restrict__ spec, const G1Pass* __restrict__ passes, int pass0,
                                                                       cplx* __restrict__ Gp, int N0, int Nh, int Nhp, SpecLayout lay,
                                                                       int rows_per_chunk, const cplx* __restrict__ W0tab, int HM,
                                                                       const cplx* __restrict__ Xp, int ncb, int S, int npass)
{
    const int lane = threadIdx.x, n = lane & 15, kq = lane >> 4;
    const int total = ncb * S * npass;
    const int per = (total + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (logical >= total) return;
    const int tile = logical / npass;
    const int chunk = tile / ncb;
    const int m0 = (tile - chunk * ncb) * 16 * NT;
    const G1Pass pr = passes[pass0 + (logical - tile * npass)];
    const int h = pr.h, PH = 2 * h + 1;
    const int lb = chunk * rows_per_chunk;
    const int le = min(N0, lb + rows_per_chunk);
    const size_t plane_sz = (size_t)N0 * Nhp, rs = (size_t)lay.rstride;
    const bool colfac = pr.b_plane < 0;
    const cplx* __restrict__ A = spec + (size_t)pr.a_plane * plane_sz;
    const cplx* __restrict__ B = colfac ? A : spec + (size_t)pr.b_plane * plane_sz;
    const cplx* __restrict__ xp = Xp + (size_t)pr.bp * N0;
    const int tcol = min(1 + n, HM - 1);   // the lane LOADS the twiddle of lag n + 1 (lags beyond the table: unused rows of D); the A operands of
                                           // the four lag groups are gathered from the lanes of its 16-lane row (see compute)
    size_t co[NT];
    bool act[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int m = m0 + 16 * t + n;
        act[t] = m < Nh;
        co[t] = lay.col(act[t] ? m : Nh - 1);
    }
    constexpr int NS = 4;                                  // accumulator sets per column tile (S1 .. S4), four lag groups each
    d4v Sx[NT][NS];
    double g0x[NT], g0y[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int q = 0; q < NS; ++q) Sx[t][q] = (d4v){0.0, 0.0, 0.0, 0.0};
        g0x[t] = g0y[t] = 0.0;
    }
    // Software pipeline, written out as two register sets that alternate (the compiler sinks a plain "load next, use
    // current" formulation back to load-then-wait within one step, which left every step exposed to the L2 latency): the
    // loads of step s + 1 are issued, a scheduling barrier pins them there, then the MFMAs of step s run.  Addresses are
    // 32-bit byte offsets from wave-uniform plane bases (scalar base + vector offset loads), advanced by one add and one
    // clamp per step; rows past the chunk are masked by vf, so the clamp only has to keep the reads inside the plane.
    struct LoadSet { cplx tw, a[NT], b[NT]; };
    const char* __restrict__ Ab = reinterpret_cast<const char*>(A);
    const char* __restrict__ Bb = reinterpret_cast<const char*>(B);
    const char* __restrict__ Xb = reinterpret_cast<const char*>(xp);
    const char* __restrict__ Wb = reinterpret_cast<const char*>(W0tab) + (size_t)tcol * sizeof(cplx);
    unsigned cob[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) cob[t] = (unsigned)(co[t] * sizeof(cplx));
    const unsigned rsb = (unsigned)(rs * sizeof(cplx)), hmb = (unsigned)(HM * sizeof(cplx));
    const unsigned rlast = (unsigned)(N0 - 1);
    const unsigned r0 = min((unsigned)(lb + kq), rlast);
    unsigned rowb = r0 * rsb, twb = r0 * hmb, xb = r0 * (unsigned)sizeof(cplx);
    const unsigned rowb_max = rlast * rsb, twb_max = rlast * hmb, xb_max = rlast * (unsigned)sizeof(cplx);
    // DG: two diagonal Omega passes side by side (plane A with itself, plane B with itself).  Their products |A|^2, |B|^2 are real,
    // so each needs only two of the four sums: one wave does both with the loads and MFMAs of one ordinary pass -- and walks the
    // rows at the pace of the ordinary passes, which the L2 sharing of a tile depends on (single diagonal passes at half the
    // MFMAs ran ahead of the others and doubled the kernel's HBM traffic)
    auto run = [&](auto CF, auto DGt) {
        constexpr bool cf = decltype(CF)::value;
        constexpr bool DG = decltype(DGt)::value;
        auto issue = [&](LoadSet& L) {
            L.tw = *reinterpret_cast<const cplx*>(Wb + twb);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                L.a[t] = *reinterpret_cast<const cplx*>(Ab + (cob[t] + rowb));
                L.b[t] = cf ? *reinterpret_cast<const cplx*>(Xb + xb) : *reinterpret_cast<const cplx*>(Bb + (cob[t] + rowb));
            }
            rowb = min(rowb + 4u * rsb, rowb_max);
            twb = min(twb + 4u * hmb, twb_max);
            if (cf) xb = min(xb + 4u * (unsigned)sizeof(cplx), xb_max);
        };
        auto compute = [&](const LoadSet& L, double vf) {
            // A operand of lag group g: lane (row, n) needs the twiddle of lag 4 g + (n & 3), which lane (row, 4 g + (n & 3)) of the same
            // 16-lane row loaded: ds_swizzle in bit-mask mode, source lane = (lane & 0x13) | (4 g) within each half wave
            const double wx0 = L.tw.x * vf, wy0 = L.tw.y * vf;
            double wx[4], wy[4];
#define SFFT_SWZ(v, G) __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x13 | ((4 * (G)) << 5)), \
                                        __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x13 | ((4 * (G)) << 5)))
            wx[0] = SFFT_SWZ(wx0, 0); wx[1] = SFFT_SWZ(wx0, 1); wx[2] = SFFT_SWZ(wx0, 2); wx[3] = SFFT_SWZ(wx0, 3);
            wy[0] = SFFT_SWZ(wy0, 0); wy[1] = SFFT_SWZ(wy0, 1); wy[2] = SFFT_SWZ(wy0, 2); wy[3] = SFFT_SWZ(wy0, 3);
#undef SFFT_SWZ
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                // DG: H.x = |a|^2, H.y = |b|^2 (two real products; g0x / g0y are their lag-0 sums)
                const cplx H = DG ? make_double2(fma(L.a[t].x, L.a[t].x, L.a[t].y * L.a[t].y), fma(L.b[t].x, L.b[t].x, L.b[t].y * L.b[t].y))
                                  : cmulc(L.a[t], L.b[t]);
                g0x[t] = fma(H.x, vf, g0x[t]);
                g0y[t] = fma(H.y, vf, g0y[t]);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    Sx[t][0][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wx[gq], H.x, Sx[t][0][gq], 0, 0, 0);      // S1            (DG: S1 of |A|^2)
                    Sx[t][1][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wy[gq], H.y, Sx[t][1][gq], 0, 0, 0);      // S2            (DG: S3 of |B|^2)
                    Sx[t][2][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wy[gq], H.x, Sx[t][2][gq], 0, 0, 0);      // S3            (DG: S3 of |A|^2)
                    Sx[t][3][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wx[gq], H.y, Sx[t][3][gq], 0, 0, 0);      // S4            (DG: S1 of |B|^2)
                }
            }
        };
        LoadSet L0, L1;
        issue(L0);
        for (int l = lb; l < le; l += 8) {
            issue(L1);
            __builtin_amdgcn_sched_barrier(0);
            compute(L0, (l + kq < le) ? 1.0 : 0.0);
            issue(L0);
            __builtin_amdgcn_sched_barrier(0);
            compute(L1, (l + 4 + kq < le) ? 1.0 : 0.0);      // (a step past the chunk runs on zero weights)
        }
    };
    const bool diag = !colfac && pr.dual;
    if (colfac) run(std::true_type{}, std::false_type{});
    else if (diag) run(std::false_type{}, std::true_type{});
    else run(std::false_type{}, std::false_type{});
    cplx* g = Gp + pr.gp_off + (size_t)chunk * PH * Nhp;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        double sx = g0x[t], sy = g0y[t];
        sx += __shfl_xor(sx, 16); sy += __shfl_xor(sy, 16);
        sx += __shfl_xor(sx, 32); sy += __shfl_xor(sy, 32);
        const int m = m0 + 16 * t + n;
        if (!act[t]) continue;
        if (diag) {     // two real, even sequences: G(+-r) = S1 +- i S3 for |A|^2 (Sx[0], Sx[2]) and for |B|^2 (Sx[3], Sx[1])
            cplx* g2 = Gp + pr.gp_off2 + (size_t)chunk * PH * Nhp;
            if (kq == 0) { g[(size_t)h * Nhp + m] = make_double2(sx, 0.0); g2[(size_t)h * Nhp + m] = make_double2(sy, 0.0); }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * q + kq + 1;               // D lane 16 i + ... of lag group q holds lag index 4 q + i
                if (r <= h) {
                    g[(size_t)(h + r) * Nhp + m] = make_double2(Sx[t][0][q], Sx[t][2][q]);
                    g[(size_t)(h - r) * Nhp + m] = make_double2(Sx[t][0][q], -Sx[t][2][q]);
                    g2[(size_t)(h + r) * Nhp + m] = make_double2(Sx[t][3][q], Sx[t][1][q]);
                    g2[(size_t)(h - r) * Nhp + m] = make_double2(Sx[t][3][q], -Sx[t][1][q]);
                }
            }
            continue;
        }
        if (kq == 0) g[(size_t)h * Nhp + m] = make_double2(sx, sy);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * q + kq + 1;               // lag group q, row i = lane >> 4 of its 4 x 4 result blocks
            if (r <= h) {
                const double s1 = Sx[t][0][q], s2 = Sx[t][1][q], s3 = Sx[t][2][q], s4 = Sx[t][3][q];
                g[(size_t)(h + r) * Nhp + m] = make_double2(s1 - s2, s3 + s4);
                g[(size_t)(h - r) * Nhp + m] = make_double2(s1 + s2, s4 - s3);
            }
        }
    }
}

// ---- pass GROUPS: what bounds the kernels above is not the matrix pipe but the L1 / texture-addresser path -- every pass of a tile
// re-reads its two planes through the CU's L1 (42 plane reads per tile for 6 distinct planes at Fij = 6; the 64-byte pieces of the
// panel layout cost a full line each).  Here a wave takes up to three passes that share planes and loads each plane once per step:
//   a triangle (x,y), (y,z), (x,z): 3 planes for 3 passes;  an edge with the two diagonals of its planes, (x,y) + dual(|x|^2, |y|^2):
//   2 planes for the work of 2 passes.  The slots of a group have FIXED operand pairs (v0,v1), (v1,v2), (v0,v2) -- the host orders the
//   planes of a group to fit -- so that nothing in the loop is indexed at run time.  At Fij = 6 the 15 + 3 launched passes become 4 triangles + 3 edge groups: 7 waves per tile
//   and 18 plane loads per step instead of 36 (+ 7 twiddle loads instead of 9).
// One 16-column tile per wave (48 accumulator registers per slot set); the 4 x 4 x 4 matrix instruction as in greek_g1_mfma4.
// Measured at 4096^2, KerHW 8 (profiles/r02_*): 0.455 ms for the 7 groups against 0.435 - 0.45 for greek_g1_mfma and 0.458 for
// greek_g1_mfma4 -- the launch does not move.  Experiments on this kernel: without the matrix instructions (loads, Hadamard products
// and twiddles only) it takes 0.306 ms = 1.25 GB of HBM traffic at 4.1 TB/s; re-reading the same rows every step (no new memory
// traffic) 0.409 ms against the 0.31 ms its 4 x 4 x 4 instructions need at the 74 TFLOP/s the instruction sustains alone; three
// waves per SIMD (168 registers, 36 bytes of scratch) 0.53 ms; twiddles by recurrence instead of load + swizzle 0.453 ms; bursts
// of 2 or 3 steps 0.453 / 0.458 ms.  The memory side and the compute side each need most of the launch and overlap poorly at two
// waves per SIMD; the memory floor (0.8 GB of planes read once, no partial sums written) would be 0.19 ms.
// HBM fetch of the launch with the Theta passes aboard: 1.55 GB for 0.94 GB of planes (PMC, profiles/r02_b_*): sibling groups start
// whenever a slot frees up and drift apart by more rows than the 4 MB L2 holds.  Putting the groups of a tile into ONE workgroup (7 or
// 4 waves, dispatched together) was measured and dropped: 0.65 / 0.57 ms against 0.51 (the 8-slot CU leaves a slot idle).
// The stamps of SFFT_G1_TRACE (scripts/g1_trace.py) show the siblings of a tile starting within 0.8 us of each other (median) and
// ending 25 us apart (of 126 us): they drift by ~100 of their 512 rows.  Pace keeping (every wave publishes its row every 32 rows
// and naps when more than 48 .. 256 rows ahead of its slowest sibling) does bring the fetch down to the ideal -- 0.96 GB, end spread
// 7 us -- and costs far more than it saves: 0.64 - 0.80 ms (the waves differ in speed, and a napping wave leaves the matrix pipe
// idle).  Dropped; the re-reads are the price of letting every wave run at its own pace.
// Where the compute side's cycles go (scripts/micro/mfma4_pattern.hip, profiles/r02_mfma4_pattern.txt): 48 matrix instructions on 48
// accumulators with 8 different A and 6 different B operands -- this kernel's pattern -- sustain the same 74 TFLOP/s as one operand
// pair, but 26 fp64 vector instructions per 48 matrix instructions (three complex products and the operand scaling) bring it down to
// 61 at two waves per SIMD: vector fp64 work is not hidden beside the matrix pipe, it takes ~6 cycles of it per instruction.  A step
// here carries ~24 fp64 vector instructions, 16 ds_swizzle, ~10 integer instructions and 4 loads beside its 48 matrix instructions;
// a wave alone on its SIMD needs 1345 cycles per step (720 of them matrix instructions), and two waves sharing a SIMD twice that.
struct G1Group {
    int plane[3];     // planes v0, v1, v2 loaded per step (an unused v2 repeats v0)
    int mask;         // bit s set: slot s is in use.  Slot 0 = (v0, v1), slot 1 = (v1, v2), slot 2 = (v0, v2): fixed operand pairs
    int pass[3];      // slot -> G1Pass record (gp_off, gp_off2, dual; only slot 0 may be a dual pass)
    int tpass[2];     // >= 0: Theta passes (v0, J) and (v1, J) of half width ht <= 8 ride along; J is then plane[2] and slot 2 reads (v0, v1)
    int ht;
};

// MASK = false: every row chunk is a whole number of 8-row iterations (4096 / 8 chunks: 512 rows), so no step ever runs past its chunk
// and the per-step row mask (a compare, two selects, two multiplies and the masked lag-0 sums) drops out of the loop -- vector
// instructions are not free beside the matrix pipe here.
// DIT = true (whole chunks only): one radix-2 decimation step along the rows before the matrix instructions.  The twiddle of lag r at
// row x' + N0 / 2 is (-1)^r times the one at row x', so with Ye = H(x') + H(x' + N0/2) and Yo = H(x') - H(x' + N0/2) the even lags
// sum W_r(x') Ye over HALF the rows and the odd lags W_r(x') Yo: a step takes two rows of every plane (x' and x' + N0/2), forms both
// products and their sum and difference (4 more vector additions per slot), and issues the same 16 matrix instructions per slot as
// before -- for twice the rows.  Lag groups are then {2,4,6,8}, {10,..,16} on Ye and {1,3,5,7}, {9,..,15} on Yo; lane n loads the
// twiddle of the lag its 4-lane group needs, so the ds_swizzle gather of the A operands is unchanged.  A chunk is rows_per_chunk / 2
// rows x' and their partners.
// lag0 / HALF (lag half-widths 17 .. 32, e.g. KerHW 12: h = 24): a launch covers the 16 lags lag0 + 1 .. lag0 + 16 (lag0 = 0 or 16; the lag-0
// sums belong to the first launch); HALF = true (DIT only) issues just the lag groups lag0 + {2,4,6,8} and lag0 + {1,3,5,7} -- the second
// launch of h <= 24 needs no others -- i.e. half the matrix instructions of a step.
template <bool MASK, bool DIT = false, bool HALF = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) greek_g1_mfma4g(const cplx* __restrict__ spec, const G1Pass* __restrict__ passes,
                                                        const G1Group* __restrict__ groups, int ngroup,
                                                        cplx* __restrict__ Gp, int N0, int Nh, int Nhp, SpecLayout lay,
                                                        int rows_per_chunk, const cplx* __restrict__ W0tab, int HM, int ncb, int S,
                                                        unsigned long long* __restrict__ trace, int lag0)
{
    const unsigned long long t_start = trace ? wall_clock64() : 0ULL;      // (SFFT_G1_TRACE: start / end stamp and XCD of every wave)
    const int lane = (int)threadIdx.x, n = lane & 15, kq = lane >> 4;
    const int total = ncb * S * ngroup;
    const int per = (total + 7) >> 3;
    // the groups of a tile run back to back on one XCD
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (logical >= total) return;
    const int tile = logical / ngroup;
    const int chunk = tile / ncb;
    const int m0 = (tile - chunk * ncb) * 16;
    const G1Group gr = groups[logical - tile * ngroup];
    const bool use0 = (gr.mask & 1) != 0, use1 = (gr.mask & 2) != 0, use2 = (gr.mask & 4) != 0;
    const int k0 = use0 ? gr.pass[0] : (use1 ? gr.pass[1] : gr.pass[2]), k1 = use1 ? gr.pass[1] : k0, k2 = use2 ? gr.pass[2] : k0;
    const long long go0 = passes[k0].gp_off, go1 = passes[k1].gp_off, go2 = passes[k2].gp_off;
    const long long gd0 = passes[k0].gp_off2;
    const bool d0 = use0 && passes[k0].dual != 0;
    const int h = passes[k0].h, PH = 2 * h + 1;
    const int lb = DIT ? chunk * (rows_per_chunk / 2) : chunk * rows_per_chunk;
    const int le = DIT ? min(N0 / 2, lb + rows_per_chunk / 2) : min(N0, lb + rows_per_chunk);      // (DIT: the last chunk may be shorter, always whole 8-row steps)
    const size_t plane_sz = (size_t)N0 * Nhp, rs = (size_t)lay.rstride;
    const int m = m0 + n;
    const bool act = m < Nh;
    const unsigned cob = (unsigned)(lay.col(act ? m : Nh - 1) * sizeof(cplx));
    const char* __restrict__ P0 = reinterpret_cast<const char*>(spec + (size_t)gr.plane[0] * plane_sz);
    const char* __restrict__ P1 = reinterpret_cast<const char*>(spec + (size_t)gr.plane[1] * plane_sz);
    const char* __restrict__ P2 = reinterpret_cast<const char*>(spec + (size_t)gr.plane[2] * plane_sz);
    // lag whose twiddle this lane loads: plain = n + 1 (group g = lags 4 g + 1 .. 4 g + 4); DIT = groups {2,4,6,8}, {10,..,16}, {1,3,5,7}, {9,..,15}
    const int dlag = (n < 8) ? 8 * (n >> 2) + 2 * ((n & 3) + 1) : 8 * ((n >> 2) - 2) + 2 * (n & 3) + 1;
    const int tcol = min(lag0 + (DIT ? dlag : 1 + n), HM - 1);
    const char* __restrict__ Wb = reinterpret_cast<const char*>(W0tab) + (size_t)tcol * sizeof(cplx);
    const unsigned rsb = (unsigned)(rs * sizeof(cplx)), hmb = (unsigned)(HM * sizeof(cplx));
    const unsigned rlast = (unsigned)(N0 - 1);
    const unsigned r0 = min((unsigned)(lb + kq), rlast);
    unsigned rowb = r0 * rsb, twb = r0 * hmb;
    const unsigned rowb_max = (DIT ? (unsigned)(N0 / 2 - 1) : rlast) * rsb, twb_max = rlast * hmb;      // (DIT: the partner row of the clamp is the last row)
    const bool theta = gr.tpass[0] >= 0;                    // (wave uniform) edge + dual group with the Theta passes of its two planes
    const bool three = gr.plane[2] != gr.plane[0];          // (wave uniform) a third plane is in use
    const long long gt0 = theta ? passes[gr.tpass[0]].gp_off : 0, gt1 = theta ? passes[gr.tpass[1]].gp_off : 0;
    const int ht = gr.ht, PHt = 2 * ht + 1;
    d4v Sx[3][4];
    double g0x[3], g0y[3];
    double St[2][4][2], t0x[2] = {0.0, 0.0}, t0y[2] = {0.0, 0.0};       // Theta slots: four sums x two lag groups (lags 1 .. 8)
#pragma unroll
    for (int ts = 0; ts < 2; ++ts)
#pragma unroll
        for (int q = 0; q < 4; ++q) St[ts][q][0] = St[ts][q][1] = 0.0;
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Sx[sl][q] = (d4v){0.0, 0.0, 0.0, 0.0};
        g0x[sl] = g0y[sl] = 0.0;
    }
    struct LoadSet { cplx tw, v[3], u[DIT ? 3 : 1]; };        // u: the partner rows x' + N0 / 2 (DIT)
    const unsigned halfb = (unsigned)(N0 / 2) * (unsigned)(rs * sizeof(cplx));
    // (two instantiations of the loop, with and without the third plane: a run-time `three ? load : v0` makes the compiler select
    //  between ADDRESSES, which puts the whole load set in scratch)
    auto run = [&](auto MODE) {
    constexpr int mode = decltype(MODE)::value;      // 0: two planes, slot 0 only; 1: three planes; 2: two planes, slot 0 = the dual pass and slot 2 = the edge of the SAME two planes;
                                                     // 3: as 2, plus the Theta passes (v0, J), (v1, J) with J as the third plane
    constexpr bool three_c = mode == 1 || mode == 3;
    auto issue = [&](LoadSet& L) {
        L.tw = *reinterpret_cast<const cplx*>(Wb + twb);
        L.v[0] = *reinterpret_cast<const cplx*>(P0 + (cob + rowb));
        L.v[1] = *reinterpret_cast<const cplx*>(P1 + (cob + rowb));
        if (three_c) L.v[2] = *reinterpret_cast<const cplx*>(P2 + (cob + rowb)); else L.v[2] = make_double2(0.0, 0.0);
        if (DIT) {
            L.u[0] = *reinterpret_cast<const cplx*>(P0 + (cob + rowb + halfb));
            L.u[1] = *reinterpret_cast<const cplx*>(P1 + (cob + rowb + halfb));
            if (three_c) L.u[DIT ? 2 : 0] = *reinterpret_cast<const cplx*>(P2 + (cob + rowb + halfb)); else L.u[DIT ? 2 : 0] = make_double2(0.0, 0.0);
        }
        rowb = min(rowb + 4u * rsb, rowb_max);
        twb = min(twb + 4u * hmb, twb_max);
    };
    auto compute = [&](const LoadSet& L, double vf) {
        double wx[4], wy[4];
        const double wx0 = MASK ? L.tw.x * vf : L.tw.x, wy0 = MASK ? L.tw.y * vf : L.tw.y;
#define SFFT_SWZ(v, G) __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x13 | ((4 * (G)) << 5)), \
                                        __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x13 | ((4 * (G)) << 5)))
        wx[0] = SFFT_SWZ(wx0, 0); wx[1] = SFFT_SWZ(wx0, 1); wx[2] = SFFT_SWZ(wx0, 2); wx[3] = SFFT_SWZ(wx0, 3);
        wy[0] = SFFT_SWZ(wy0, 0); wy[1] = SFFT_SWZ(wy0, 1); wy[2] = SFFT_SWZ(wy0, 2); wy[3] = SFFT_SWZ(wy0, 3);
#undef SFFT_SWZ
        // DU: 0 = an ordinary pass, 1 = a dual pass, 2 = decided at run time (d0).  Only a group of one pass on two planes (mode 0) needs
        // the run-time form; written with a run-time flag everywhere, every slot-0 step computed BOTH products and selected between them
        // (8 more fp64 instructions and 8 v_cndmask per step: this launch is bound by vector / matrix instruction issue, see DESIGN).
        auto slot = [&](auto SL, auto DU) {
            constexpr int sl = decltype(SL)::value;
            constexpr int du = decltype(DU)::value;
            const bool dual = du == 2 ? d0 : (du == 1);
            const cplx va = L.v[sl == 1 ? 1 : 0], vb = L.v[(sl == 0 || mode >= 2) ? 1 : 2];
            // dual: H.x = |a|^2, H.y = |b|^2 (two real products side by side)
            const cplx H = dual ? make_double2(fma(va.x, va.x, va.y * va.y), fma(vb.x, vb.x, vb.y * vb.y)) : cmulc(va, vb);
            if (DIT) {
                const cplx ua = L.u[DIT ? (sl == 1 ? 1 : 0) : 0], ub = L.u[DIT ? ((sl == 0 || mode >= 2) ? 1 : 2) : 0];
                const cplx Hh = dual ? make_double2(fma(ua.x, ua.x, ua.y * ua.y), fma(ub.x, ub.x, ub.y * ub.y)) : cmulc(ua, ub);
                const cplx Ye = make_double2(H.x + Hh.x, H.y + Hh.y), Yo = make_double2(H.x - Hh.x, H.y - Hh.y);
                g0x[sl] += Ye.x;
                g0y[sl] += Ye.y;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    if (HALF && (gq & 1)) continue;
                    const double bx = gq < 2 ? Ye.x : Yo.x, by = gq < 2 ? Ye.y : Yo.y;
                    Sx[sl][0][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wx[gq], bx, Sx[sl][0][gq], 0, 0, 0);
                    Sx[sl][1][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wy[gq], by, Sx[sl][1][gq], 0, 0, 0);
                    Sx[sl][2][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wy[gq], bx, Sx[sl][2][gq], 0, 0, 0);
                    Sx[sl][3][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wx[gq], by, Sx[sl][3][gq], 0, 0, 0);
                }
                return;
            }
            g0x[sl] = MASK ? fma(H.x, vf, g0x[sl]) : g0x[sl] + H.x;
            g0y[sl] = MASK ? fma(H.y, vf, g0y[sl]) : g0y[sl] + H.y;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                Sx[sl][0][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wx[gq], H.x, Sx[sl][0][gq], 0, 0, 0);      // S1   (dual: S1 of |A|^2)
                Sx[sl][1][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wy[gq], H.y, Sx[sl][1][gq], 0, 0, 0);      // S2   (dual: S3 of |B|^2)
                Sx[sl][2][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wy[gq], H.x, Sx[sl][2][gq], 0, 0, 0);      // S3   (dual: S3 of |A|^2)
                Sx[sl][3][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wx[gq], H.y, Sx[sl][3][gq], 0, 0, 0);      // S4   (dual: S1 of |B|^2)
            }
        };
        using DU0 = std::integral_constant<int, 0>; using DU1 = std::integral_constant<int, 1>; using DU2 = std::integral_constant<int, 2>;
        if (mode == 0) { if (use0) slot(std::integral_constant<int, 0>{}, DU2{}); }
        if (mode == 1) {
            if (use0) slot(std::integral_constant<int, 0>{}, DU0{});
            if (use1) slot(std::integral_constant<int, 1>{}, DU0{});
            if (use2) slot(std::integral_constant<int, 2>{}, DU0{});
        }
        if (mode >= 2) { slot(std::integral_constant<int, 0>{}, DU1{}); slot(std::integral_constant<int, 2>{}, DU0{}); }
        if (mode == 3) {
#pragma unroll
            for (int ts = 0; ts < 2; ++ts) {
                const cplx H = cmulc(L.v[ts], L.v[2]);           // FI_x conj(FJ)
                if (DIT) {      // lags 2, 4, 6, 8 on Ye (twiddle group 0) and 1, 3, 5, 7 on Yo (twiddle group 2)
                    const cplx Hh = cmulc(L.u[DIT ? ts : 0], L.u[DIT ? 2 : 0]);
                    const cplx Ye = make_double2(H.x + Hh.x, H.y + Hh.y), Yo = make_double2(H.x - Hh.x, H.y - Hh.y);
                    t0x[ts] += Ye.x;
                    t0y[ts] += Ye.y;
#pragma unroll
                    for (int gq = 0; gq < 2; ++gq) {
                        const double bx = gq == 0 ? Ye.x : Yo.x, by = gq == 0 ? Ye.y : Yo.y;
                        St[ts][0][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wx[2 * gq], bx, St[ts][0][gq], 0, 0, 0);
                        St[ts][1][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wy[2 * gq], by, St[ts][1][gq], 0, 0, 0);
                        St[ts][2][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wy[2 * gq], bx, St[ts][2][gq], 0, 0, 0);
                        St[ts][3][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wx[2 * gq], by, St[ts][3][gq], 0, 0, 0);
                    }
                    continue;
                }
                t0x[ts] = MASK ? fma(H.x, vf, t0x[ts]) : t0x[ts] + H.x;
                t0y[ts] = MASK ? fma(H.y, vf, t0y[ts]) : t0y[ts] + H.y;
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
                    St[ts][0][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wx[gq], H.x, St[ts][0][gq], 0, 0, 0);
                    St[ts][1][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wy[gq], H.y, St[ts][1][gq], 0, 0, 0);
                    St[ts][2][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wy[gq], H.x, St[ts][2][gq], 0, 0, 0);
                    St[ts][3][gq] = __builtin_amdgcn_mfma_f64_4x4x4f64(wx[gq], H.y, St[ts][3][gq], 0, 0, 0);
                }
            }
        }
    };
    // two load sets alternate: the loads of step k + 1 are in flight while step k's matrix instructions issue (bursts of 2 or 3 steps per
    // set were measured and changed nothing)
    LoadSet LA, LB;
    issue(LA);
    for (int l = lb; l < le; l += 8) {
        issue(LB);
        __builtin_amdgcn_sched_barrier(0);
        compute(LA, (!MASK || l + kq < le) ? 1.0 : 0.0);
        issue(LA);
        __builtin_amdgcn_sched_barrier(0);
        compute(LB, (!MASK || l + 4 + kq < le) ? 1.0 : 0.0);      // (a step past the chunk runs on zero weights)
    }
    };
    if (theta) run(std::integral_constant<int, 3>{});
    else if (three) run(std::integral_constant<int, 1>{});
    else if (use2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 0>{});
    if (!act) return;
    auto emit = [&](auto SL, long long gp_off, long long gp_off2, bool dual, int h) {      // h: the lag half width of THIS slot's pass
        constexpr int sl = decltype(SL)::value;
        const int PH = 2 * h + 1;
        double sx = g0x[sl], sy = g0y[sl];
        sx += __shfl_xor(sx, 16); sy += __shfl_xor(sy, 16);
        sx += __shfl_xor(sx, 32); sy += __shfl_xor(sy, 32);
        cplx* g = Gp + gp_off + (size_t)chunk * PH * Nhp;
        if (dual) {     // two real, even sequences: G(+-r) = S1 +- i S3 for |A|^2 (Sx[0], Sx[2]) and for |B|^2 (Sx[3], Sx[1])
            cplx* g2 = Gp + gp_off2 + (size_t)chunk * PH * Nhp;
            if (kq == 0 && lag0 == 0) { g[(size_t)h * Nhp + m] = make_double2(sx, 0.0); g2[(size_t)h * Nhp + m] = make_double2(sy, 0.0); }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (HALF && (q & 1)) continue;
                const int r = lag0 + (DIT ? (q < 2 ? 8 * q + 2 * (kq + 1) : 8 * (q - 2) + 2 * kq + 1) : 4 * q + kq + 1);
                if (r <= h) {
                    g[(size_t)(h + r) * Nhp + m] = make_double2(Sx[sl][0][q], Sx[sl][2][q]);
                    g[(size_t)(h - r) * Nhp + m] = make_double2(Sx[sl][0][q], -Sx[sl][2][q]);
                    g2[(size_t)(h + r) * Nhp + m] = make_double2(Sx[sl][3][q], Sx[sl][1][q]);
                    g2[(size_t)(h - r) * Nhp + m] = make_double2(Sx[sl][3][q], -Sx[sl][1][q]);
                }
            }
            return;
        }
        if (kq == 0 && lag0 == 0) g[(size_t)h * Nhp + m] = make_double2(sx, sy);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (HALF && (q & 1)) continue;
            const int r = lag0 + (DIT ? (q < 2 ? 8 * q + 2 * (kq + 1) : 8 * (q - 2) + 2 * kq + 1) : 4 * q + kq + 1);
            if (r <= h) {
                const double s1 = Sx[sl][0][q], s2 = Sx[sl][1][q], s3 = Sx[sl][2][q], s4 = Sx[sl][3][q];
                g[(size_t)(h + r) * Nhp + m] = make_double2(s1 - s2, s3 + s4);
                g[(size_t)(h - r) * Nhp + m] = make_double2(s1 + s2, s4 - s3);
            }
        }
    };
    if (use0) emit(std::integral_constant<int, 0>{}, go0, gd0, d0, passes[k0].h);
    if (use1) emit(std::integral_constant<int, 1>{}, go1, 0LL, false, passes[k1].h);
    if (use2) emit(std::integral_constant<int, 2>{}, go2, 0LL, false, passes[k2].h);
    if (theta) {
#pragma unroll
        for (int ts = 0; ts < 2; ++ts) {
            double sx = t0x[ts], sy = t0y[ts];
            sx += __shfl_xor(sx, 16); sy += __shfl_xor(sy, 16);
            sx += __shfl_xor(sx, 32); sy += __shfl_xor(sy, 32);
            cplx* g = Gp + (ts == 0 ? gt0 : gt1) + (size_t)chunk * PHt * Nhp;
            if (kq == 0) g[(size_t)ht * Nhp + m] = make_double2(sx, sy);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = DIT ? (q == 0 ? 2 * (kq + 1) : 2 * kq + 1) : 4 * q + kq + 1;
                if (r <= ht) {
                    const double s1 = St[ts][0][q], s2 = St[ts][1][q], s3 = St[ts][2][q], s4 = St[ts][3][q];
                    g[(size_t)(ht + r) * Nhp + m] = make_double2(s1 - s2, s3 + s4);
                    g[(size_t)(ht - r) * Nhp + m] = make_double2(s1 + s2, s4 - s3);
                }
            }
        }
    }
    if (trace && lane == 0 && blockIdx.x < 65536u) { trace[3 * (size_t)blockIdx.x] = t_start; trace[3 * (size_t)blockIdx.x + 1] = wall_clock64(); trace[3 * (size_t)blockIdx.x + 2] = (unsigned long long)logical; }
}

// ---- Omega products of basis terms with (nearly) DISJOINT supports, in real space (round 4) ---------------------------------------
// PreOMG[(a),(b)](r, e) = SCALE^3 sum_x I_a(x) I_b(x + (r, e))  (circular), I_a = I * kbx[ia][x0] * kby[ja][x1].
// For a B-spline basis most pairs of terms have supports that overlap in neither axis or in one only: of the 325 products of a 5 x 5
// basis 132 have row (or column) factors whose supports are disjoint, so that kbx[ia][x0] kbx[ib][x0 + r] is nonzero only for the few
// rows within |r| <= h of a shared knot or of the image edge (the circular wrap) -- ~270 (row, lag) pairs instead of 6144 x 33.  Such a
// product does not go through the transforms at all (no Omega pass, no partial sums, no stage-2 job): its patch is a handful of 1-D
// correlations between image rows (x mode) or image columns (y mode: from a transposed copy of the few columns involved).
// One workgroup per ITEM (product, lag u along the sparse axis): it walks the lines c of the product whose weight wA[c] wB[c + u] does
// not vanish (listed per lag by the host: scanning a candidate list on the device cost two dependent global loads per candidate,
// 3.3 ms at config 3) and, of each line, only the span [t0, t1) on which the in-line factors can meet (the support of fA, cut by the
// support of fB widened by h when that does not wrap).  A line is cut into STEPS of OSP_CH positions; a step keeps its own positions
// (times in-line factor and weight) in registers, the partner line's window in LDS with HP positions of circular halo, and accumulates
// the 2 h + 1 lags along the line.  The loop is software-pipelined: the global loads of step s + 1 are in flight while step s is
// computed, the partner windows alternate between two LDS buffers (one barrier per step).  One block reduction at the end writes a row
// (x mode) or a column (y mode) of the patch.  Deterministic: no atomics.
// The host orders the items by (mode, cross factors) -- items that read the same few image rows -- and deals contiguous runs of that
// order to the eight XCDs (workgroup b runs on XCD b & 7), so that an XCD's L2 holds the rows its items share (a run's rows are a few
// MB; dealt round-robin every XCD pulled every row through the fabric: 1.56 ms at config 3 for 0.13 ms of multiply-adds).
struct SparseProd {
    int fa_cross, fb_cross;   // factor tables (rows of `cross`) whose product along the SPARSE axis gives the weight
    int fa_line, fb_line;     // factor tables (rows of `inl`) along the line
    int patch_off;            // doubles, into patches
    int t0, t1;               // span of line positions to visit (multiples of 4; empty: the patch is zero)
    int ustart[2 * 16 + 2];   // lines[ustart[u + h] .. ustart[u + h + 1]): the line pairs of lag u
    int ymode;                // 0: lines are image rows (sparse axis = rows); 1: lines are image columns (from the transposed strip)
};
// one line pair of an item: position c on the sparse axis with weight w = wA[c] wB[c + u] != 0 (the tables are plan constants: the host
// multiplies them); the offsets of line c and of its partner c + u (circular) in the image (x mode) or in the strip (y mode)
struct SparseLine { long long a_off, b_off; double w; };

// strip[c][x0] = I[x0][cols[c]]: the columns the y-mode products touch, transposed (coalesced along c)
__global__ void __launch_bounds__(256) gather_cols(const double* __restrict__ I, const int* __restrict__ cols, int ncols, double* __restrict__ strip, int N0, int N1)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), x0 = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c < ncols && x0 < N0) strip[(size_t)c * N0 + x0] = I[(size_t)x0 * N1 + cols[c]];
}

#define OSP_NT 256                      // threads per workgroup
#define OSP_SW 2                        // sweeps per step: a thread owns four consecutive positions per sweep
#define OSP_HP 16                       // lag half width, padded
#define OSP_NG (OSP_NT * OSP_SW)        // groups of four window positions per step = one buffer's groups (two per thread)
#define OSP_CH (4 * OSP_NG)                        // line positions per step (2048: spans are mostly multiples of it -- with 2008 a third of the steps carried 40 .. 120 positions)
#define OSP_NT_TAIL (2 * OSP_HP + 8)               // window positions behind the OSP_NG groups (the far halo and the reach of the last windows): one per thread tid < 40
#define OSP_PL (OSP_NG + OSP_NT_TAIL / 4)          // slots per plane
struct __attribute__((aligned(8))) OspD4 { double v[4]; };     // (image rows of a caller's array: 8-byte alignment is all that is promised)

__global__ void __launch_bounds__(OSP_NT, 2) omega_sparse(const double* __restrict__ I, const double* __restrict__ strip,
                                                    const SparseProd* __restrict__ prods, const SparseLine* __restrict__ lines, const int2* __restrict__ items,
                                                    const double* __restrict__ kbx, const double* __restrict__ kby,
                                                    int N0, int N1, int h, double scale3, double* __restrict__ patches)
{
    constexpr int HP = OSP_HP;
    // the partner window in FOUR interleaved planes -- window position p sits at plane p & 3, slot p >> 2 -- so that the lanes of a wave,
    // whose windows start four positions apart, read consecutive slots of one plane (with the window stored contiguously every read was
    // an eight-way bank conflict)
    __shared__ double Bl[2][4][OSP_PL + 1];
    __shared__ double red[OSP_NT / 64][2 * HP + 1];
    const int2 item = items[(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)];      // (product, lag); XCD b & 7 walks its own run
    if (item.x < 0) return;
    const SparseProd& P = prods[item.x];                        // (read in place: a local copy of its per-lag table would sit in scratch)
    const int u = item.y;                                       // lag along the sparse axis
    const int tid = threadIdx.x;
    const int NL = P.ymode ? N0 : N1;                           // line length (>= 2 HP + 8, a multiple of 4: the host's conditions)
    const double* __restrict__ src = P.ymode ? strip : I;
    const double* __restrict__ inl = P.ymode ? kbx : kby;       // factors along the line
    const double* __restrict__ fA = inl + (size_t)P.fa_line * NL;
    const double* __restrict__ fB = inl + (size_t)P.fb_line * NL;
    const int t0 = P.t0, t1 = P.t1;
    const int nch = (t1 - t0 + OSP_CH - 1) / OSP_CH;            // steps per line
    const int k0 = P.ustart[u + h], k1 = P.ustart[u + h + 1];
    const int nsteps = t1 > t0 ? (k1 - k0) * nch : 0;
    double acc[2 * HP + 1];
#pragma unroll
    for (int e = 0; e <= 2 * HP; ++e) acc[e] = 0.0;
    // the raw operands of one step: own positions (line a, factor fA, weight) and the partner window (line b, factor fB)
    OspD4 rA[OSP_SW], rF[OSP_SW], rB[OSP_SW], rG[OSP_SW];
    double rw[OSP_SW], rtb = 0.0, rtg = 0.0;       // (rtb, rtg: this thread's position of the window's tail, threads tid < OSP_NT_TAIL)
    bool rv[OSP_SW], rtv = false;
    auto fetch = [&](int kk, int chn) {            // line kk of the item's list, step chn of the line
        const SparseLine L = lines[k0 + kk];
        const double w = L.w;
        const double* __restrict__ la = src + L.a_off;
        const double* __restrict__ lb = src + L.b_off;
        const int base = t0 + chn * OSP_CH, len = min(OSP_CH, t1 - base);
#pragma unroll
        for (int ch = 0; ch < OSP_SW; ++ch) {       // (clamped, loaded unconditionally, masked: `cond ? ptr[i] : 0` is a branch with a full wait per load)
            const int g = OSP_NT * ch + tid;
            const int t = base + min(4 * g, len - 4);
            rA[ch] = *reinterpret_cast<const OspD4*>(la + t);
            rF[ch] = *reinterpret_cast<const OspD4*>(fA + t);
            rw[ch] = 4 * g < len ? w : 0.0;
            int q = base - HP + min(4 * g, len + 2 * HP - 4);      // window group g: line positions q .. q + 3 (NL and every offset here are multiples of 4: a group never straddles the wrap)
            q += q < 0 ? NL : 0; q -= q >= NL ? NL : 0;
            rB[ch] = *reinterpret_cast<const OspD4*>(lb + q);
            rG[ch] = *reinterpret_cast<const OspD4*>(fB + q);
            rv[ch] = 4 * g < len + 2 * HP;          // (zeros behind the halo: the last windows reach past it)
        }
        {
            const int pt = OSP_CH + min(tid, OSP_NT_TAIL - 1);
            int q = base - HP + min(pt, len + 2 * HP - 1);
            q += q < 0 ? NL : 0; q -= q >= NL ? NL : 0;
            rtb = lb[q]; rtg = fB[q];
            rtv = pt < len + 2 * HP;
        }
    };
    double areg[OSP_SW][4];
    auto settle = [&](int buf) {                   // raw operands -> this step's registers and window
#pragma unroll
        for (int ch = 0; ch < OSP_SW; ++ch)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                areg[ch][v] = rA[ch].v[v] * rF[ch].v[v] * rw[ch];
                const double b = rB[ch].v[v] * rG[ch].v[v];
                Bl[buf][v][OSP_NT * ch + tid] = rv[ch] ? b : 0.0;
            }
        // (branch-free: the other threads write a slot nobody reads -- behind a branch the compiler sinks the multiply-adds again)
        Bl[buf][tid & 3][tid < OSP_NT_TAIL ? OSP_NG + (tid >> 2) : OSP_PL] = rtv ? rtb * rtg : 0.0;
    };
    if (nsteps > 0) { fetch(0, 0); settle(0); }
    __syncthreads();
    int fk = 0, fc = 0;                             // the step being fetched
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps && ++fc == nch) { fc = 0; ++fk; }
        fetch(fk, fc);                              // in flight during the multiply-adds below (the last step's is a harmless repeat)
        const int buf = s & 1;
        // a thread keeps the partners of its four positions in registers, the lags in two halves (a window of HP + 4 values at a time):
        // 20 8-byte LDS reads per 68 / 64 multiply-adds (one read per multiply-add made the kernel LDS-bound)
#pragma unroll
        for (int ch = 0; ch < OSP_SW; ++ch) {
            const int g = OSP_NT * ch + tid;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                __builtin_amdgcn_sched_barrier(0);  // (one window live at a time)
                constexpr int NE0 = HP + 1;         // lags [0, HP] then [HP + 1, 2 HP]
                const int e0 = half ? NE0 : 0, ne = half ? HP : NE0;
                double bl[HP + 4];
#pragma unroll
                for (int j = 0; j < HP + 4; ++j) bl[j] = (j < ne + 3) ? Bl[buf][(e0 + j) & 3][g + ((e0 + j) >> 2)] : 0.0;
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int e = 0; e < NE0; ++e) if (e < ne) acc[e0 + e] = fma(areg[ch][v], bl[v + e], acc[e0 + e]);       // lag e0 + e - HP
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        settle(buf ^ 1);                            // (unconditional -- after the last step a repeat nobody reads: behind a branch the compiler sank every multiply-add of the step below it, past the sched_barriers)
        __syncthreads();
    }
    // block reduction of the 2 HP + 1 sums
#pragma unroll
    for (int e = 0; e <= 2 * HP; ++e) {
        double v = acc[e];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((tid & 63) == 0) red[tid >> 6][e] = v;
    }
    __syncthreads();
    const int PH = 2 * h + 1;
    if (tid < PH) {
        const int e = tid - h + HP;                              // line lag tid - h
        double v = 0.0;
#pragma unroll
        for (int wv = 0; wv < OSP_NT / 64; ++wv) v += red[wv][e];
        v *= scale3;
        // x mode: u = row lag r, the line lag = column lag e;  y mode: u = column lag e, the line lag = row lag r
        double* out = patches + P.patch_off;
        if (P.ymode) out[(size_t)tid * PH + (u + h)] = v; else out[(size_t)(u + h) * PH + tid] = v;
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
                                                     cplx* __restrict__ Gp, int N0, int Nh, int Nhp, SpecLayout lay, int S)
{
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= Nh) return;
    const G1Pass pr = passes[pass0 + blockIdx.y];
    const int PH = 2 * pr.h + 1;
    const cplx a0 = spec[(size_t)pr.a_plane * N0 * Nhp + lay.col(m)];
    const cplx v = make_double2(a0.x * (double)N0, a0.y * (double)N0);
    cplx* g = Gp + pr.gp_off + m;
    for (int c = 0; c < S; ++c)
        for (int r = 0; r < PH; ++r) g[((size_t)c * PH + r) * Nhp] = (c == 0) ? v : make_double2(0.0, 0.0);
}

// Greek stage 2: patch[r][e] = scale * sum_{m < Nh} wgt[m] * Re( W1^(m e) * y[m] * sum_chunks G[r][m] ),  |e| <= h.
// wgt = 1 for the self-conjugate columns (m = 0, and m = N1/2 when N1 is even), 2 otherwise;
// y[m] = conj(tscale * Yq[q][m]) for Gamma jobs, 1 otherwise.
template <int CB8>      // chunk partials per batch of loads (4 when the launch has at most four chunks, else 8)
__global__ void __launch_bounds__(256) greek_g2(const cplx* __restrict__ Gp, const G1Pass* __restrict__ passes,
                                                const PatchJob* __restrict__ jobs, int job0,
                                                double* __restrict__ patches, int Nh, int Nhp, int N1, int S,
                                                const cplx* __restrict__ root1, const cplx* __restrict__ Yq, double tscale)
{
    const PatchJob jb = jobs[job0 + blockIdx.y];
    if (jb.pass < 0) return;                 // a derived Omega patch (omega_derive)
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
            for (int c0 = 0; c0 < S; c0 += CB8) {       // a batch of chunk partials: the loads are independent and issue together
                cplx v[CB8];
#pragma unroll
                for (int c = 0; c < CB8; ++c) v[c] = g[(size_t)min(c0 + c, S - 1) * PH * Nhp + m];
#pragma unroll
                for (int c = 0; c < CB8; ++c) {
                    const double f = (c0 + c < S) ? 1.0 : 0.0;
                    gx = fma(v[c].x, f, gx); gy = fma(v[c].y, f, gy);
                }
            }
            if (yq) {
                const cplx y = yq[m];
                const cplx t = cmulc(make_double2(gx, gy), make_double2(y.x * tscale, y.y * tscale));
                gx = t.x; gy = t.y;
            }
            const double wgt = (m == 0 || (even && m == N1 / 2)) ? 1.0 : 2.0;
            gx *= wgt; gy *= wgt;
            if (e0 == 0) U[0] += gx;
            // W1^(m (e0 + t)), t = 1 .. 16, from two table entries: products of at most five factors (16 scattered table reads
            // per column were the cost of this kernel)
            const cplx wb = root1[(int)(((long long)m * e0) % N1)];
            cplx wq[5];
            wq[0] = make_double2(1.0, 0.0);
            wq[1] = root1[m];
            wq[2] = cmul(wq[1], wq[1]);
            wq[3] = cmul(wq[2], wq[1]);
            wq[4] = cmul(wq[2], wq[2]);
            cplx w4k = wb;                          // W1^(m (e0 + 4 k))
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
#pragma unroll
                for (int rr = 1; rr <= 4; ++rr) {
                    const int t = 4 * k4 + rr;
                    const cplx w = cmul(w4k, wq[rr]);
                    if (t <= ne) {
                        U[t] = fma(gx, w.x, U[t]);
                        V[t] = fma(gy, w.y, V[t]);
                    }
                    if (rr == 4) w4k = w;
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
// Four image rows per workgroup: a thread reads the nq table values of its column once for the four rows (with one row per
// workgroup the table re-reads out of L2 were four times the image traffic).
#define ROWMOM_R 4
template <int NQ>
__global__ void __launch_bounds__(256) row_moments(const double* __restrict__ J, double* __restrict__ rowmom, int N0, int N1,
                                                   const double* __restrict__ tby, int nq)
{
    const int l0 = blockIdx.x * ROWMOM_R, tid = threadIdx.x;
    double acc[ROWMOM_R][NQ];
#pragma unroll
    for (int rr = 0; rr < ROWMOM_R; ++rr)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[rr][q] = 0.0;
    const double* __restrict__ jr[ROWMOM_R];
#pragma unroll
    for (int rr = 0; rr < ROWMOM_R; ++rr) jr[rr] = J + (size_t)min(l0 + rr, N0 - 1) * N1;      // (rows past the image: recomputed, not stored)
    if ((N1 & 1) == 0 && ((reinterpret_cast<unsigned long long>(J) | reinterpret_cast<unsigned long long>(tby)) & 15ULL) == 0) {      // 16-byte loads: two columns per lane and instruction
#pragma unroll 2
        for (int n = 2 * tid; n < N1; n += 512) {
            double2 t[NQ], v[ROWMOM_R];
#pragma unroll
            for (int rr = 0; rr < ROWMOM_R; ++rr) v[rr] = *reinterpret_cast<const double2*>(jr[rr] + n);
#pragma unroll
            for (int q = 0; q < NQ; ++q) t[q] = *reinterpret_cast<const double2*>(tby + (size_t)min(q, nq - 1) * N1 + n);
#pragma unroll
            for (int rr = 0; rr < ROWMOM_R; ++rr)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[rr][q] = fma(v[rr].y, t[q].y, fma(v[rr].x, t[q].x, acc[rr][q]));
        }
    } else {
#pragma unroll 4
        for (int n = tid; n < N1; n += 256) {
            double t[NQ], v[ROWMOM_R];
#pragma unroll
            for (int rr = 0; rr < ROWMOM_R; ++rr) v[rr] = jr[rr][n];
#pragma unroll
            for (int q = 0; q < NQ; ++q) t[q] = tby[(size_t)min(q, nq - 1) * N1 + n];               // q >= nq: unused copies
#pragma unroll
            for (int rr = 0; rr < ROWMOM_R; ++rr)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[rr][q] = fma(v[rr], t[q], acc[rr][q]);
        }
    }
    __shared__ double red[4][ROWMOM_R][NQ];
#pragma unroll
    for (int rr = 0; rr < ROWMOM_R; ++rr)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double u = acc[rr][q];
            for (int off = 32; off > 0; off >>= 1) u += __shfl_down(u, off);
            if ((tid & 63) == 0) red[tid >> 6][rr][q] = u;
        }
    __syncthreads();
    if (tid < ROWMOM_R * NQ) {
        const int rr = tid / NQ, q = tid - rr * NQ;
        if (l0 + rr < N0) rowmom[(size_t)(l0 + rr) * SFFT_MAX_BQ + q] = red[0][rr][q] + red[1][rr][q] + red[2][rr][q] + red[3][rr][q];
    }
}

// ------------------------------------------------------------------------------------------------
// Gamma block for polynomial bases, without any transform.  PreGAM[ij, pq](r, e) = SCALE^2 * sum_x I_ij(x) T_pq(x + (r, e))
// (circular) is a sum over pixels of I times polynomial weights:
//     sum_x0 cx(x0)^i cx((x0 + r) mod N0)^p  R_{j,q,e}[x0],     R_{j,q,e}[x0] = sum_x1 I(x0, x1) cy(x1)^j cy((x1 + e) mod N1)^q.
// Away from the wrap, cy(x1 + e) = cy(x1) + e / N1, so R is a binomial combination of the row moments mu_d[x0] = sum_x1 I cy^d
// (d <= DK + DB, one streaming read of I: row_moments) plus a correction from the |e| columns that wrap.  This replaces the
// Fij * (DB + 1) column-factor passes of greek_g1 (12 of the 18 short passes at orders 2/2) and their stage-2 jobs.
// Two small kernels: gamma_rows (one thread per image row: every R_{j,q,e} of the row) and gamma_patches (one workgroup per
// (ij, pq, e): the sum over rows for the 2 w + 1 row lags).
// ------------------------------------------------------------------------------------------------
#define GAMMA_MAXW 12
#define GAMMA_ND SFFT_MAX_BQ             // row stride of the moment table (that of row_moments)
struct GammaArgs {
    int Fij, Fpq, w;
    int ki[64], kj[64];                     // kernel term ij = (row factor ki) x (column factor kj): exponents, or table rows
    int bp[SFFT_MAX_PQ], bq[SFFT_MAX_PQ];   // background term pq = cx^bp cy^bq
};

// step 1: R_{j,q,e}[x0] for every row, every (j, q) with j <= DK, q <= DB and every column lag e: Rtab[(j * NQB + q)][e + w][x0]
// kby != nullptr: tabulated column factors of the kernel basis (B-spline kernels): the moments are mu_{j,d} = sum_x1 I kby[j] cy^d at
// index j * (DB + 1) + d, and kby[j][x1] replaces cy(x1)^j in the wrap correction (the background must still be polynomial)
__global__ void __launch_bounds__(256) gamma_rows(const double* __restrict__ I, const double* __restrict__ mu, const double* __restrict__ tby,
                                                  const double* __restrict__ kby, int nmu, int DB, int w, int N0, int N1,
                                                  double* __restrict__ Rtab)
{
    const int x0 = blockIdx.x * 256 + threadIdx.x;
    if (x0 >= N0) return;
    const int PH = 2 * w + 1, NQB = DB + 1;
    double m[GAMMA_ND];
#pragma unroll
    for (int d = 0; d < GAMMA_ND; ++d) m[d] = (d < nmu) ? mu[(size_t)x0 * GAMMA_ND + d] : 0.0;
    // the w first and w last pixels of the row: the only ones whose shifted partner can wrap
    double lo[GAMMA_MAXW], hi[GAMMA_MAXW];
    const double* __restrict__ row = I + (size_t)x0 * N1;
#pragma unroll
    for (int t = 0; t < GAMMA_MAXW; ++t) {
        lo[t] = (t < w) ? row[min(t, N1 - 1)] : 0.0;
        hi[t] = (t < w) ? row[max(N1 - 1 - t, 0)] : 0.0;           // hi[t] = I(x0, N1 - 1 - t)
    }
    {
            const int j = (int)blockIdx.y / NQB, q = (int)blockIdx.y - j * NQB;          // grid.y = (DK + 1) (DB + 1)
            double* __restrict__ out = Rtab + (size_t)(j * NQB + q) * PH * N0 + x0;
            for (int e = -w; e <= w; ++e) {
                const double de = (double)e / (double)N1;
                double R = 0.0, bin = 1.0, dk = 1.0;
                for (int k = 0; k <= q; ++k) {              // sum_k C(q, k) de^k mu_{j + q - k}
                    const int mi = kby ? j * NQB + (q - k) : j + q - k;
                    double mv = 0.0;
#pragma unroll
                    for (int d = 0; d < GAMMA_ND; ++d) mv = (d == mi) ? m[d] : mv;
                    R = fma(bin * dk, mv, R);
                    bin = bin * (double)(q - k) / (double)(k + 1);
                    dk *= de;
                }
                const double* __restrict__ tq = tby + (size_t)q * N1;
                const int ne = e < 0 ? -e : e;
#pragma unroll
                for (int t = 0; t < GAMMA_MAXW; ++t) {
                    if (t < ne) {
                        // e > 0: the last e columns, x1 = N1 - 1 - t, wrap to x1 + e - N1;  e < 0: the first |e| columns, x1 = t, wrap to x1 + e + N1
                        const int x1 = e > 0 ? N1 - 1 - t : t;
                        const int xw = e > 0 ? x1 + e - N1 : x1 + e + N1;
                        const double cy = (double)(x1 + 1) / (double)N1;
                        const double pix = e > 0 ? hi[t] : lo[t];
                        const double fj = kby ? kby[(size_t)j * N1 + x1] : ipow(cy, j);
                        R = fma(pix * fj, tq[xw] - ipow(cy + de, q), R);
                    }
                }
                out[(size_t)(e + w) * N0] = R;
            }
        }
}

// step 2: patch[ij, pq](r, e) = scale * sum_x0 cx(x0)^i cx((x0 + r) mod N0)^p R_{j,q,e}[x0]; one workgroup per (ij, pq, e)
__global__ void __launch_bounds__(256) gamma_patches(const double* __restrict__ Rtab, const double* __restrict__ kbx, const double* __restrict__ tbx,
                                                     GammaArgs ga, int NQB, int N0, double* __restrict__ patches, double scale)
{
    const int ij = blockIdx.x / ga.Fpq, pq = blockIdx.x - ij * ga.Fpq;
    const int w = ga.w, PH = 2 * w + 1;
    const int eI = blockIdx.y;
    const int i = ga.ki[ij], j = ga.kj[ij], pp = ga.bp[pq], q = ga.bq[pq];
    const int tid = threadIdx.x;
    double acc[2 * GAMMA_MAXW + 1];
#pragma unroll
    for (int t = 0; t <= 2 * GAMMA_MAXW; ++t) acc[t] = 0.0;
    const double* __restrict__ R = Rtab + ((size_t)(j * NQB + q) * PH + eI) * N0;
    const double* __restrict__ tp = tbx + (size_t)pp * N0;
    const double* __restrict__ ki = kbx + (size_t)i * N0;
    for (int x0 = tid; x0 < N0; x0 += 256) {
        const double wr = ki[x0] * R[x0];
#pragma unroll
        for (int t = 0; t <= 2 * GAMMA_MAXW; ++t) {             // row lag r = t - w (compile-time register indices)
            int xr = x0 + t - w; if (xr < 0) xr += N0; else if (xr >= N0) xr -= N0;
            const double tv = (t < PH) ? tp[xr] : 0.0;
            acc[t] = fma(wr, tv, acc[t]);
        }
    }
    __shared__ double red[4][2 * GAMMA_MAXW + 1];
#pragma unroll
    for (int t = 0; t <= 2 * GAMMA_MAXW; ++t) {
        double u = acc[t];
        for (int off = 32; off > 0; off >>= 1) u += __shfl_down(u, off);
        if ((tid & 63) == 0) red[tid >> 6][t] = u;
    }
    __syncthreads();
    if (tid < PH) patches[(size_t)(ij * ga.Fpq + pq) * PH * PH + (size_t)tid * PH + eI] = scale * (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
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

#endif
