// sfft_cpu.cpp -- C++/OpenMP CPU restatement of the reference's Numpy backend.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// What it restates (thomasvrussell/sfft v1.7.3, paths relative to the reference checkout):
//   sfft/sfftcore/SFFTConfigure.py:817-1367   SingleSFFTConfigure_Numpy.SSCN: the 17 numba functions
//       SpatialCoor :889-902, SpatialPoly :905-937, HadProd_OMG :941-954, FillLS_OMG :957-1018, HadProd_GAM :1025-1038,
//       FillLS_GAM :1041-1077, HadProd_PSI :1084-1097, FillLS_PSI :1100-1136, HadProd_PHI :1143-1156, FillLS_PHI :1159-1180,
//       HadProd_THE :1187-1198, FillLS_THE :1201-1234, HadProd_DEL :1241-1252, FillLS_DEL :1255-1272,
//       Remove_LSFStripes :1279-1293, Extend_Solution :1299-1311, Construct_FDIFF :1316-1359
//   sfft/sfftcore/SFFTSubtract.py:477-821     ElementalSFFTSubtract_Numpy.ESSN (sequencing, scalings, twiddle tables :778-792)
//   sfft/sfftcore/SFFTSubtract.py:839-923     GeneralSFFTSubtract.GSS (solve on the masked pair, apply to the full pair)
// Third-party arithmetic under the reference's path, not in its checkout: pyfftw.interfaces.numpy_fft.fft2 / ifft2 (FFTW,
// pyfftw >= 0.12.0, setup.py:29) -> the mixed-radix Stockham transform below; numpy.linalg.solve (LAPACK gesv) -> the blocked
// LU with partial pivoting below.  DFT and LU are mathematically fixed; parity is within the fp64 tolerances of SURVEY 8(c).
//
// It is the SAME ALGORITHM as the reference's CPU path: complex-to-complex transforms of every plane (1 + Fij + Fpq
// preliminary planes per ESS, FOMG + FGAM + FPSI + FPHI + FTHE + FDEL Greek planes), the full-size twiddle planes
// Kab_Wla / Kab_Wmb, the per-pixel Construct_FDIFF triple loop.  The one liberty: Greek planes are produced, transformed
// and gathered one at a time instead of as one batched array (SFFTSubtract.py:628 batches FOMG planes), which does not
// change a value and keeps memory at ~50 planes instead of ~190.
//
// Uses: (1) tests/test_cpu_restatement.py pins it against the reference-made fixtures (LHMAT / RHb / Solution / DIFF);
//       (2) bench.py's cpu_baseline leg times it on the GPU box's host cores (oracle/cpu_baseline.py).
// Only tests/, __graft_entry__ and bench.py's cpu_baseline leg load this library.
//
// Build: g++ -O3 -march=native -fopenmp -shared -fPIC -o oracle/_build/libsfft_cpu.so oracle/csrc/sfft_cpu.cpp
#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

typedef std::complex<double> cd;

namespace {

// big plane buffers: not value-initialised (a std::vector would zero 3.5 GB on one thread and place every page on its NUMA node);
// pages are first touched by the OpenMP loops that fill them
template <typename T> struct Buf {
    T* p = nullptr;
    size_t n = 0;
    Buf() {}
    explicit Buf(size_t n_) { resize(n_); }
    ~Buf() { std::free(p); }
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;
    void resize(size_t n_) { if (n_ <= n && p) return; std::free(p); n = n_; p = (T*)std::malloc(sizeof(T) * (n_ ? n_ : 1)); }      // grow only
    T* data() { return p; }
    const T* data() const { return p; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

// The big buffers live as long as the process and only ever grow: a 13 GB working set that is unmapped after every call is
// 3.3 million page faults per call, taken under the process's address-space lock -- at 128 threads that lock, not the arithmetic,
// sets the time.  (Calls are not re-entrant: one GSS at a time per process, which is how the tests and the bench use it.)
static Buf<cd> g_planes, g_work, g_Wla, g_Wmb, g_FD;

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------------------------------
// 1-D complex DFT of any length: Stockham autosort, radices 4 / 2 / 3 / 5 and a generic O(r^2) butterfly for other primes.
// sign = -1: forward (numpy.fft.fft), +1: backward without the 1/N (numpy.fft.ifft divides by N; done by the caller).
// ------------------------------------------------------------------------------------------------------------------
struct Fft1d {
    int N = 0;
    std::vector<int> radix;
    std::vector<cd> w;          // w[t] = exp(-2 pi i t / N)
    bool ok = true;             // false: a prime factor above 64 (the generic butterfly's buffer); the entry points return -2
    void init(int n)
    {
        N = n;
        radix.clear();
        int m = n;
        while (m % 4 == 0) { radix.push_back(4); m /= 4; }
        while (m % 2 == 0) { radix.push_back(2); m /= 2; }
        while (m % 3 == 0) { radix.push_back(3); m /= 3; }
        while (m % 5 == 0) { radix.push_back(5); m /= 5; }
        for (int f = 7; m > 1; f += 2)
            while (m % f == 0) { radix.push_back(f); m /= f; if (f > 64) ok = false; }
        w.resize(n);
        for (int t = 0; t < n; ++t) {
            // octant-exact arguments keep the table accurate to an ulp for every t
            const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)t / (long double)n;
            w[t] = cd((double)cosl(a), (double)sinl(a));
        }
    }
    // x: input (destroyed), y: scratch; returns the pointer (x or y) that holds the result in natural order
    cd* run(cd* x, cd* y, int sign) const
    {
        int n = N, s = 1;
        for (size_t st = 0; st < radix.size(); ++st) {
            const int r = radix[st], m = n / r, tw = N / n;
            for (int p = 0; p < m; ++p) {
                cd wp[8];
                cd wbig[64];
                cd* wj = r <= 8 ? wp : wbig;
                for (int j = 0; j < r && j < 64; ++j) {
                    cd v = w[(size_t)((long long)p * j * tw % N)];
                    wj[j] = sign < 0 ? v : std::conj(v);
                }
                if (r == 4) {
                    for (int q = 0; q < s; ++q) {
                        const cd a0 = x[q + s * (p)], a1 = x[q + s * (p + m)], a2 = x[q + s * (p + 2 * m)], a3 = x[q + s * (p + 3 * m)];
                        const cd t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3;
                        cd t3 = a1 - a3;                       // times -i (forward) or +i (backward)
                        t3 = sign < 0 ? cd(t3.imag(), -t3.real()) : cd(-t3.imag(), t3.real());
                        y[q + s * (4 * p)] = t0 + t2;
                        y[q + s * (4 * p + 1)] = (t1 + t3) * wj[1];
                        y[q + s * (4 * p + 2)] = (t0 - t2) * wj[2];
                        y[q + s * (4 * p + 3)] = (t1 - t3) * wj[3];
                    }
                } else if (r == 2) {
                    for (int q = 0; q < s; ++q) {
                        const cd a0 = x[q + s * p], a1 = x[q + s * (p + m)];
                        y[q + s * (2 * p)] = a0 + a1;
                        y[q + s * (2 * p + 1)] = (a0 - a1) * wj[1];
                    }
                } else {
                    // generic radix: b_j = sum_k a_k W_r^(jk)
                    const int rt = N / r;
                    for (int q = 0; q < s; ++q) {
                        cd a[64];
                        for (int k = 0; k < r; ++k) a[k] = x[q + s * (p + m * k)];
                        for (int j = 0; j < r; ++j) {
                            cd acc = a[0];
                            for (int k = 1; k < r; ++k) {
                                cd v = w[(size_t)((long long)j * k % r) * rt];
                                acc += a[k] * (sign < 0 ? v : std::conj(v));
                            }
                            y[q + s * (r * p + j)] = acc * wj[j];
                        }
                    }
                }
            }
            std::swap(x, y);
            n = m;
            s *= r;
        }
        return x;
    }
};

// per-thread scratch that lives as long as the thread (OpenMP pool threads persist): a fresh std::vector per parallel region is a
// 0.5 MB mmap / munmap per thread per transform, which serialises on the process's address-space lock at high thread counts
struct Scratch {
    cd* p = nullptr;
    size_t n = 0;
    ~Scratch() { std::free(p); }
    cd* get(size_t need) { if (need > n) { std::free(p); p = (cd*)std::malloc(sizeof(cd) * need); n = need; } return p; }
};
thread_local Scratch tl_scratch;

struct Fft2d {
    int N0, N1;
    Fft1d f0, f1;
    Fft2d(int n0, int n1) : N0(n0), N1(n1) { f0.init(n0); f1.init(n1); }
    bool ok() const { return f0.ok && f1.ok; }
    // in-place 2-D transform of a row-major [N0][N1] complex plane, times `scale`
    void run(cd* a, int sign, double scale, int nthreads) const
    {
        const int CB = 16;  // columns gathered per block for the axis-0 pass (256 contiguous bytes per row)
#pragma omp parallel num_threads(nthreads)
        {
            const size_t M = (size_t)std::max(N0, N1);
            cd* base = tl_scratch.get(2 * M + (size_t)CB * N0);
            cd* bx = base;
            cd* by = base + M;
            cd* blk = base + 2 * M;
#pragma omp for schedule(static)
            for (int r = 0; r < N0; ++r) {
                cd* row = a + (size_t)r * N1;
                std::memcpy(bx, row, sizeof(cd) * N1);
                const cd* res = f1.run(bx, by, sign);
                std::memcpy(row, res, sizeof(cd) * N1);
            }
#pragma omp for schedule(static)
            for (int c0 = 0; c0 < N1; c0 += CB) {
                const int nc = std::min(CB, N1 - c0);
                for (int r = 0; r < N0; ++r)
                    for (int c = 0; c < nc; ++c) blk[(size_t)c * N0 + r] = a[(size_t)r * N1 + c0 + c];
                for (int c = 0; c < nc; ++c) {
                    std::memcpy(bx, &blk[(size_t)c * N0], sizeof(cd) * N0);
                    const cd* res = f0.run(bx, by, sign);
                    for (int r = 0; r < N0; ++r) blk[(size_t)c * N0 + r] = res[r] * scale;
                }
                for (int r = 0; r < N0; ++r)
                    for (int c = 0; c < nc; ++c) a[(size_t)r * N1 + c0 + c] = blk[(size_t)c * N0 + r];
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------------------------
// numpy.linalg.solve: LU with partial pivoting (gesv), blocked right-looking, row-major n x n, one right-hand side.
// Returns 0, or 1 if an exactly zero pivot is met (numpy raises LinAlgError("Singular matrix")).
// ------------------------------------------------------------------------------------------------------------------
int lu_solve(double* A, double* b, int n, int nthreads)
{
    const int NB = 48;
    std::vector<int> piv(n);
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int kb = std::min(NB, n - k0);
        // panel factorisation (columns k0 .. k0+kb), whole rows are swapped
        for (int k = k0; k < k0 + kb; ++k) {
            int p = k;
            double best = std::fabs(A[(size_t)k * n + k]);
            for (int i = k + 1; i < n; ++i) {
                const double v = std::fabs(A[(size_t)i * n + k]);
                if (v > best) { best = v; p = i; }
            }
            if (best == 0.0) return 1;
            piv[k] = p;
            if (p != k) {
                for (int j = 0; j < n; ++j) std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]);
                std::swap(b[k], b[p]);
            }
            const double inv = 1.0 / A[(size_t)k * n + k];
            for (int i = k + 1; i < n; ++i) {          // (the panel is 48 columns wide: one thread; the parallel work is the trailing update)
                const double l = A[(size_t)i * n + k] * inv;
                A[(size_t)i * n + k] = l;
                double* ri = A + (size_t)i * n;
                const double* rk = A + (size_t)k * n;
                for (int j = k + 1; j < k0 + kb; ++j) ri[j] -= l * rk[j];
            }
        }
        const int k1 = k0 + kb;
        if (k1 >= n) break;
        // U12 = L11^-1 A12 (unit lower triangular solve on the panel rows)
        for (int k = k0; k < k1; ++k)
            for (int i = k + 1; i < k1; ++i) {
                const double l = A[(size_t)i * n + k];
                double* ri = A + (size_t)i * n;
                const double* rk = A + (size_t)k * n;
                for (int j = k1; j < n; ++j) ri[j] -= l * rk[j];
            }
        // A22 -= L21 U12
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int i = k1; i < n; ++i) {
            double* ri = A + (size_t)i * n;
            for (int k = k0; k < k1; ++k) {
                const double l = ri[k];
                const double* rk = A + (size_t)k * n;
                for (int j = k1; j < n; ++j) ri[j] -= l * rk[j];
            }
        }
    }
    for (int i = 0; i < n; ++i) {            // forward substitution (L has a unit diagonal; b is already permuted)
        double acc = b[i];
        const double* ri = A + (size_t)i * n;
        for (int j = 0; j < i; ++j) acc -= ri[j] * b[j];
        b[i] = acc;
    }
    for (int i = n - 1; i >= 0; --i) {
        double acc = b[i];
        const double* ri = A + (size_t)i * n;
        for (int j = i + 1; j < n; ++j) acc -= ri[j] * b[j];
        b[i] = acc / ri[i];
    }
    return 0;
}

struct Params {
    int N0, N1, w0, w1, DK, DB, cpr;
    int L0, L1, Fab, Fij, Fpq, Fijab, NEQ, NEQ_FSfree;
    double SCALE, SCALE_L;
    std::vector<int> REF_ij, REF_pq, REF_ab;     // pairs, flattened
};

Params make_params(int N0, int N1, int w, int DK, int DB, int cpr)
{
    Params p;                                          // SFFTConfigure.py:825-883
    p.N0 = N0; p.N1 = N1; p.w0 = w; p.w1 = w; p.DK = DK; p.DB = DB; p.cpr = cpr;
    p.L0 = 2 * w + 1; p.L1 = 2 * w + 1; p.Fab = p.L0 * p.L1;
    p.Fij = (DK + 1) * (DK + 2) / 2; p.Fpq = (DB + 1) * (DB + 2) / 2;
    p.Fijab = p.Fij * p.Fab; p.NEQ = p.Fijab + p.Fpq; p.NEQ_FSfree = p.NEQ - (p.Fij - 1);
    p.SCALE = 1.0 / ((double)N0 * (double)N1);
    p.SCALE_L = 1.0 / p.SCALE;
    for (int i = 0; i <= DK; ++i) for (int j = 0; j <= DK - i; ++j) { p.REF_ij.push_back(i); p.REF_ij.push_back(j); }   // SFFTSubtract.py:514-517
    for (int i = 0; i <= DB; ++i) for (int j = 0; j <= DB - i; ++j) { p.REF_pq.push_back(i); p.REF_pq.push_back(j); }
    for (int a = 0; a < p.L0; ++a) for (int b = 0; b < p.L1; ++b) { p.REF_ab.push_back(a - w); p.REF_ab.push_back(b - w); }
    return p;
}

inline int modn(int v, int N)        // fmod then + N if negative (SFFTConfigure.py:978-1000)
{
    int t = v % N;
    return t < 0 ? t + N : t;
}

// SpatialCoor + SpatialPoly (:889-937) followed by the preliminary DFTs (SFFTSubtract.py:572-587):
// planes[0] = SCALE * DFT2(J), planes[1 .. Fij] = SCALE * DFT2(I * cx^i * cy^j), planes[Fij+1 ..] = SCALE * DFT2(cx^p * cy^q)
void preliminary(const Params& p, const Fft2d& fft, const double* I, const double* J, Buf<cd>& planes, int nthreads)
{
    const int N0 = p.N0, N1 = p.N1;
    const size_t P = (size_t)N0 * N1;
    planes.resize((size_t)(1 + p.Fij + p.Fpq) * P);
    std::vector<double> cx(N0), cy(N1);
    for (int r = 0; r < N0; ++r) cx[r] = ((double)r + 1.0) / (double)N0;     // ScaledFortranCoor
    for (int c = 0; c < N1; ++c) cy[c] = ((double)c + 1.0) / (double)N1;
    for (int k = 0; k < 1 + p.Fij + p.Fpq; ++k) {
        cd* pl = planes.data() + (size_t)k * P;
        int ex = 0, ey = 0, kind = 0;      // kind 0: J, 1: I * poly, 2: poly
        if (k >= 1 && k <= p.Fij) { kind = 1; ex = p.REF_ij[2 * (k - 1)]; ey = p.REF_ij[2 * (k - 1) + 1]; }
        if (k > p.Fij) { kind = 2; ex = p.REF_pq[2 * (k - 1 - p.Fij)]; ey = p.REF_pq[2 * (k - 1 - p.Fij) + 1]; }
        std::vector<double> pyv(N1);
        for (int c = 0; c < N1; ++c) pyv[c] = std::pow(cy[c], ey);
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int r = 0; r < N0; ++r) {
            const double px = std::pow(cx[r], ex);
            for (int c = 0; c < N1; ++c) {
                const double poly = px * pyv[c];
                const double v = kind == 0 ? J[(size_t)r * N1 + c] : (kind == 1 ? I[(size_t)r * N1 + c] * poly : poly);
                pl[(size_t)r * N1 + c] = cd(v, 0.0);
            }
        }
        fft.run(pl, -1, p.SCALE, nthreads);
    }
}

// one Greek plane: work = left * conj(right) (HadProd_*), forward DFT2 times SCALE, real part times `post`
void greek_plane(const Params& p, const Fft2d& fft, const cd* left, const cd* right_conj_of, cd* work, double post, int nthreads)
{
    const size_t P = (size_t)p.N0 * p.N1;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long long t = 0; t < (long long)P; ++t) work[t] = left[t] * std::conj(right_conj_of[t]);
    fft.run(work, -1, p.SCALE * post, nthreads);
}

}  // namespace

extern "C" {

// ESS(PixA_I, PixA_J, SFFTConfig, SFFTSolution=None): LHMAT [NEQ][NEQ] and RHb [NEQ] as the reference hands them to the stripe
// removal (may be NULL), Solution [NEQ].  stage_s (may be NULL): seconds of [prelim, OMG, GAM, PSI, PHI, THE+DEL, solve].
int sfftcpu_solve(int N0, int N1, int w, int DK, int DB, int cpr, const double* I, const double* J, double* solution,
                  double* LHMAT_out, double* RHb_out, int nthreads, double* stage_s)
{
    if (nthreads < 1) nthreads = omp_get_max_threads();
    const Params p = make_params(N0, N1, w, DK, DB, cpr);
    const Fft2d fft(N0, N1);
    if (!fft.ok()) return -2;
    const size_t P = (size_t)N0 * N1;
    const int Fij = p.Fij, Fpq = p.Fpq, Fab = p.Fab, Fijab = p.Fijab, NEQ = p.NEQ;
    double t0 = now_s(), st[7] = {0, 0, 0, 0, 0, 0, 0};
    Buf<cd>& planes = g_planes;
    preliminary(p, fft, I, J, planes, nthreads);
    const cd* FJ = planes.data();
    const cd* FI = planes.data() + P;
    const cd* FT = planes.data() + (size_t)(1 + Fij) * P;
    st[0] = now_s() - t0;
    Buf<cd>& work = g_work;
    work.resize(P);
    std::vector<double> LH((size_t)NEQ * NEQ), RHb(NEQ);
    const int* ab = p.REF_ab.data();

    // OMEGA: PreOMG = SCALE * Re[SCALE * DFT2(FI[i8j8] * conj(FI[ij]))]  (SFFTSubtract.py:622-636), FillLS_OMG :957-1018
    t0 = now_s();
    for (int i8j8 = 0; i8j8 < Fij; ++i8j8)
        for (int ij = 0; ij < Fij; ++ij) {
            greek_plane(p, fft, FI + (size_t)i8j8 * P, FI + (size_t)ij * P, work.data(), p.SCALE, nthreads);
            const cd* Pre = work.data();
            const double P00 = Pre[0].real();
#pragma omp parallel for num_threads(nthreads) schedule(static)
            for (int a8b8 = 0; a8b8 < Fab; ++a8b8) {
                const int a8 = ab[2 * a8b8], b8 = ab[2 * a8b8 + 1];
                const int ROW = i8j8 * Fab + a8b8;
                for (int abi = 0; abi < Fab; ++abi) {
                    const int a = ab[2 * abi], b = ab[2 * abi + 1];
                    const int COL = ij * Fab + abi;
                    const double vrow = Pre[(size_t)modn(a8, N0) * N1 + modn(b8, N1)].real();
                    const double vcol = Pre[(size_t)modn(-a, N0) * N1 + modn(-b, N1)].real();
                    const double vdif = Pre[(size_t)modn(a8 - a, N0) * N1 + modn(b8 - b, N1)].real();
                    double v;
                    const bool rc = (a8 == 0 && b8 == 0), cc = (a == 0 && b == 0);
                    if (!rc && !cc) v = -vrow - vcol + vdif + P00;
                    else if (rc && !cc) v = vcol - P00;
                    else if (!rc && cc) v = vrow - P00;
                    else v = P00;
                    LH[(size_t)ROW * NEQ + COL] = v;
                }
            }
        }
    st[1] = now_s() - t0;
    // GAMMA: PreGAM = Re[SCALE * DFT2(FI[i8j8] * conj(FT[pq]))]  (:642-655), FillLS_GAM :1041-1077
    t0 = now_s();
    for (int i8j8 = 0; i8j8 < Fij; ++i8j8)
        for (int pq = 0; pq < Fpq; ++pq) {
            greek_plane(p, fft, FI + (size_t)i8j8 * P, FT + (size_t)pq * P, work.data(), 1.0, nthreads);
            const cd* Pre = work.data();
            for (int a8b8 = 0; a8b8 < Fab; ++a8b8) {
                const int a8 = ab[2 * a8b8], b8 = ab[2 * a8b8 + 1];
                const double v = (a8 == 0 && b8 == 0) ? Pre[0].real() : Pre[(size_t)modn(a8, N0) * N1 + modn(b8, N1)].real() - Pre[0].real();
                LH[(size_t)(i8j8 * Fab + a8b8) * NEQ + Fijab + pq] = v;
            }
        }
    st[2] = now_s() - t0;
    // PSI: PrePSI = Re[SCALE * DFT2(conj(FI[ij]) * FT[p8q8])]  (:661-674), FillLS_PSI :1100-1136
    t0 = now_s();
    for (int p8q8 = 0; p8q8 < Fpq; ++p8q8)
        for (int ij = 0; ij < Fij; ++ij) {
            greek_plane(p, fft, FT + (size_t)p8q8 * P, FI + (size_t)ij * P, work.data(), 1.0, nthreads);
            const cd* Pre = work.data();
            for (int abi = 0; abi < Fab; ++abi) {
                const int a = ab[2 * abi], b = ab[2 * abi + 1];
                const double v = (a == 0 && b == 0) ? Pre[0].real() : Pre[(size_t)modn(-a, N0) * N1 + modn(-b, N1)].real() - Pre[0].real();
                LH[(size_t)(Fijab + p8q8) * NEQ + ij * Fab + abi] = v;
            }
        }
    st[3] = now_s() - t0;
    // PHI: PrePHI = SCALE_L * Re[SCALE * DFT2(FT[p8q8] * conj(FT[pq]))]  (:680-694), FillLS_PHI :1159-1180
    t0 = now_s();
    for (int p8q8 = 0; p8q8 < Fpq; ++p8q8)
        for (int pq = 0; pq < Fpq; ++pq) {
            greek_plane(p, fft, FT + (size_t)p8q8 * P, FT + (size_t)pq * P, work.data(), p.SCALE_L, nthreads);
            LH[(size_t)(Fijab + p8q8) * NEQ + Fijab + pq] = work.data()[0].real();
        }
    st[4] = now_s() - t0;
    // THETA, DELTA (:701-729): PreTHE = Re[SCALE * DFT2(conj(FJ) * FI[i8j8])], PreDEL = SCALE_L * Re[SCALE * DFT2(conj(FJ) * FT[p8q8])]
    t0 = now_s();
    for (int i8j8 = 0; i8j8 < Fij; ++i8j8) {
        greek_plane(p, fft, FI + (size_t)i8j8 * P, FJ, work.data(), 1.0, nthreads);
        const cd* Pre = work.data();
        for (int a8b8 = 0; a8b8 < Fab; ++a8b8) {
            const int a8 = ab[2 * a8b8], b8 = ab[2 * a8b8 + 1];
            RHb[i8j8 * Fab + a8b8] = (a8 == 0 && b8 == 0) ? Pre[0].real() : Pre[(size_t)modn(a8, N0) * N1 + modn(b8, N1)].real() - Pre[0].real();
        }
    }
    for (int p8q8 = 0; p8q8 < Fpq; ++p8q8) {
        greek_plane(p, fft, FT + (size_t)p8q8 * P, FJ, work.data(), p.SCALE_L, nthreads);
        RHb[Fijab + p8q8] = work.data()[0].real();
    }
    st[5] = now_s() - t0;
    if (LHMAT_out) std::memcpy(LHMAT_out, LH.data(), sizeof(double) * LH.size());
    if (RHb_out) std::memcpy(RHb_out, RHb.data(), sizeof(double) * NEQ);

    // Remove_LSFStripes (:1279-1293), numpy.linalg.solve, Extend_Solution (:1299-1311)  (SFFTSubtract.py:734-755)
    t0 = now_s();
    std::vector<int> idx;
    for (int k = 0; k < NEQ; ++k) {
        bool forbidden = false;
        if (cpr) for (int ij = 1; ij < Fij; ++ij) if (k == ij * Fab + p.w0 * p.L1 + p.w1) forbidden = true;
        if (!forbidden) idx.push_back(k);
    }
    const int n = (int)idx.size();
    std::vector<double> A((size_t)n * n), bb(n);
    for (int i = 0; i < n; ++i) {
        bb[i] = RHb[idx[i]];
        for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = LH[(size_t)idx[i] * NEQ + idx[j]];
    }
    const int sing = lu_solve(A.data(), bb.data(), n, nthreads);
    for (int k = 0; k < NEQ; ++k) solution[k] = 0.0;
    for (int i = 0; i < n; ++i) solution[idx[i]] = bb[i];
    st[6] = now_s() - t0;
    if (stage_s) for (int k = 0; k < 7; ++k) stage_s[k] = st[k];
    return sing ? -4 : 0;
}

// ESS(PixA_I, PixA_J, SFFTConfig, SFFTSolution=solution, Subtract=True): DIFF [N0][N1].
// stage_s (may be NULL): seconds of [prelim, twiddle tables, Construct_FDIFF, inverse DFT].
int sfftcpu_apply(int N0, int N1, int w, int DK, int DB, int cpr, const double* I, const double* J, const double* solution,
                  double* diff, int nthreads, double* stage_s)
{
    if (nthreads < 1) nthreads = omp_get_max_threads();
    const Params p = make_params(N0, N1, w, DK, DB, cpr);
    const Fft2d fft(N0, N1);
    if (!fft.ok()) return -2;
    const size_t P = (size_t)N0 * N1;
    const int Fij = p.Fij, Fpq = p.Fpq, Fab = p.Fab, Fijab = p.Fijab, L0 = p.L0, L1 = p.L1;
    double t0 = now_s(), st[4] = {0, 0, 0, 0};
    Buf<cd>& planes = g_planes;
    preliminary(p, fft, I, J, planes, nthreads);
    const cd* FJ = planes.data();
    const cd* FI = planes.data() + P;
    const cd* FT = planes.data() + (size_t)(1 + Fij) * P;
    st[0] = now_s() - t0;

    // Kab_Wla[a + w0] = Wl ** a, Kab_Wmb[b + w1] = Wm ** b as FULL-SIZE planes (SFFTSubtract.py:778-792), Wl = exp(-2 pi i row / N0)
    t0 = now_s();
    Buf<cd>& Wla = g_Wla;
    Buf<cd>& Wmb = g_Wmb;
    Wla.resize((size_t)L0 * P);
    Wmb.resize((size_t)L1 * P);
    for (int a = -p.w0; a <= p.w0; ++a) {
        cd* pl = Wla.data() + (size_t)(a + p.w0) * P;
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int r = 0; r < N0; ++r) {
            const double ang = -2.0 * M_PI * (double)modn(r * a, N0) / (double)N0;
            const cd v(std::cos(ang), std::sin(ang));
            for (int c = 0; c < N1; ++c) pl[(size_t)r * N1 + c] = v;
        }
    }
    for (int b = -p.w1; b <= p.w1; ++b) {
        cd* pl = Wmb.data() + (size_t)(b + p.w1) * P;
        std::vector<cd> rowv(N1);               // Wm ** b depends on the column only: one row of values, broadcast to the plane
        for (int c = 0; c < N1; ++c) {
            const double ang = -2.0 * M_PI * (double)modn(c * b, N1) / (double)N1;
            rowv[c] = cd(std::cos(ang), std::sin(ang));
        }
#pragma omp parallel for num_threads(nthreads) schedule(static)
        for (int r = 0; r < N0; ++r) std::memcpy(pl + (size_t)r * N1, rowv.data(), sizeof(cd) * N1);
    }
    st[1] = now_s() - t0;

    // Construct_FDIFF (SFFTConfigure.py:1316-1359): per pixel, over ab then ij
    t0 = now_s();
    Buf<cd>& FD = g_FD;
    FD.resize(P);
    const int* ab = p.REF_ab.data();
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int ROW = 0; ROW < N0; ++ROW)
        for (int COL = 0; COL < N1; ++COL) {
            const size_t t = (size_t)ROW * N1 + COL;
            cd PVAL(0.0, 0.0);
            for (int abi = 0; abi < Fab; ++abi) {
                const int a = ab[2 * abi], b = ab[2 * abi + 1];
                cd FKab;
                if (a == 0 && b == 0) FKab = cd(p.SCALE, 0.0);
                else FKab = cd(p.SCALE, 0.0) * (Wla[(size_t)(p.w0 + a) * P + t] * Wmb[(size_t)(p.w1 + b) * P + t] - cd(1.0, 0.0));
                for (int ij = 0; ij < Fij; ++ij) PVAL += (cd(solution[ij * Fab + abi], 0.0) * FI[(size_t)ij * P + t]) * FKab;
            }
            for (int pq = 0; pq < Fpq; ++pq) PVAL += cd(solution[Fijab + pq], 0.0) * FT[(size_t)pq * P + t];
            FD[t] = FJ[t] - PVAL;
        }
    st[2] = now_s() - t0;
    // DIFF = Re[SCALE_L * ifft2(FDIFF)]; numpy's ifft2 carries 1 / (N0 N1)  (SFFTSubtract.py:804-807)
    t0 = now_s();
    fft.run(FD.data(), +1, p.SCALE_L * p.SCALE, nthreads);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (long long t = 0; t < (long long)P; ++t) diff[t] = FD[t].real();
    st[3] = now_s() - t0;
    if (stage_s) for (int k = 0; k < 4; ++k) stage_s[k] = st[k];
    return 0;
}

// GeneralSFFTSubtract.GSS (SFFTSubtract.py:839-905): solve on (mI, mJ), apply to (I, J).  stage_s: 7 + 4 entries.
int sfftcpu_gss(int N0, int N1, int w, int DK, int DB, int cpr, const double* I, const double* J, const double* mI, const double* mJ,
                double* solution, double* diff, int nthreads, double* stage_s)
{
    int rc = sfftcpu_solve(N0, N1, w, DK, DB, cpr, mI, mJ, solution, nullptr, nullptr, nthreads, stage_s);
    if (rc) return rc;
    return sfftcpu_apply(N0, N1, w, DK, DB, cpr, I, J, solution, diff, nthreads, stage_s ? stage_s + 7 : nullptr);
}

// numpy.fft.fft2 / ifft2 of a [N0][N1] complex plane in place (sign -1 / +1; ifft2 includes 1/(N0 N1)) -- for the unit test
int sfftcpu_fft2(int N0, int N1, double* plane, int sign, int nthreads)
{
    if (nthreads < 1) nthreads = omp_get_max_threads();
    const Fft2d fft(N0, N1);
    if (!fft.ok()) return -2;
    fft.run((cd*)plane, sign, sign < 0 ? 1.0 : 1.0 / ((double)N0 * (double)N1), nthreads);
    return 0;
}

int sfftcpu_max_threads(void) { return omp_get_max_threads(); }

}  // extern "C"
