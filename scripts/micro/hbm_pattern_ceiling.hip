// What can a pure data mover reach with the ACCESS PATTERNS of the transform passes (no arithmetic, no LDS, no phases)?
// Planes of 4096 x 2052 complex (134 MB) in the 4-column panel layout [Nhp/4][N0][4] of the library.
//   copy      : contiguous 16 bytes per lane, plane -> plane                                   (the device's copy ceiling)
//   rows_like : reads real rows contiguously (8 bytes per lane), writes 16 bytes per lane into three planes of the panel layout -- 64 consecutive
//               columns of a row = 16 pieces of 64 bytes, 256 KB apart; the two rows of a pair by two instructions (row-pass stores)
//   cols_like : one workgroup per 4-column panel piece walk: lane quad = one 64-byte piece, 16 rows per wave instruction (1 KB
//               contiguous), 16 instructions 256 rows apart in flight, plane -> plane                (column-pass loads and stores)
//   omega_like: one wave per 16-column tile: an instruction takes 4 rows x 16 columns (4 pieces of 256 B), 3 planes, read only
// Reported: GB/s of bytes moved (read + written).   hipcc --offload-arch=gfx950 -O3 -o /tmp/hpc scripts/micro/hbm_pattern_ceiling.hip && /tmp/hpc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double2 cplx;
constexpr int N0 = 4096, Nhp = 2052, NP = Nhp / 4;
constexpr size_t PLANE = (size_t)N0 * Nhp;

__global__ void __launch_bounds__(256) copy_k(const cplx* __restrict__ a, cplx* __restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
// one workgroup per row pair: 256 threads x 16 columns each
__global__ void __launch_bounds__(256) rows_like(const double* __restrict__ img, cplx* __restrict__ out)
{
    const int l0 = 2 * blockIdx.x, j = threadIdx.x;
    double a0[16], a1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = img[(size_t)l0 * N0 + j + 256 * r]; a1[r] = img[(size_t)(l0 + 1) * N0 + j + 256 * r]; }
    for (int p = 0; p < 3; ++p)            // three stage planes per image, as the solve pass writes them
#pragma unroll
    for (int r = 0; r < 8; ++r) {          // half spectrum: 2048 columns
        const int m = j + 256 * r;
        cplx* o = out + p * PLANE + (size_t)(m >> 2) * N0 * 4 + (size_t)l0 * 4 + (m & 3);
        o[0] = make_double2(a0[r], a0[r + 8] + p);
        o[4] = make_double2(a1[r], a1[r + 8] + p);
    }
}
// one workgroup (512 threads) per panel: lane quad = 64-byte piece, thread owns rows (tid >> 2) + 128 r... 32 rows per thread
__global__ void __launch_bounds__(512) cols_like(const cplx* __restrict__ a, cplx* __restrict__ b)
{
    const size_t base = (size_t)blockIdx.x * N0 * 4;
    const int q = threadIdx.x & 3, row = threadIdx.x >> 2;
    cplx v[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) v[r] = a[base + (size_t)(row + 128 * r) * 4 + q];
#pragma unroll
    for (int r = 0; r < 32; ++r) b[base + (size_t)(row + 128 * r) * 4 + q] = v[r];
}
// one wave per (16-column tile, 512-row chunk): reads 3 planes
__global__ void __launch_bounds__(64) omega_like(const cplx* __restrict__ a, cplx* __restrict__ sink)
{
    const int tile = blockIdx.x >> 3, chunk = blockIdx.x & 7, lane = threadIdx.x;
    const int col = 16 * tile + (lane & 15), rr = lane >> 4;
    if (col >= Nhp) return;
    const size_t co = (size_t)(col >> 2) * N0 * 4 + (col & 3);
    cplx acc = make_double2(0.0, 0.0);
    for (int x = 512 * chunk; x < 512 * (chunk + 1); x += 8) {
        cplx v[6];
#pragma unroll
        for (int p = 0; p < 3; ++p) { v[2 * p] = a[p * PLANE + co + (size_t)(x + rr) * 4]; v[2 * p + 1] = a[p * PLANE + co + (size_t)(x + 4 + rr) * 4]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) { acc.x += v[k].x; acc.y += v[k].y; }
    }
    if (acc.x == 1.2345e-300) sink[blockIdx.x] = acc;
}
int main()
{
    cplx *a, *b; double* img;
    hipMalloc(&a, 3 * PLANE * sizeof(cplx)); hipMalloc(&b, 3 * PLANE * sizeof(cplx)); hipMalloc(&img, (size_t)N0 * N0 * 8);
    hipMemset(a, 0, 3 * PLANE * sizeof(cplx)); hipMemset(img, 0, (size_t)N0 * N0 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, double bytes, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s %8.3f ms per launch  %7.1f GB/s  [%s]\n", name, ms / 10, bytes / (ms / 10) * 1e-6, hipGetErrorString(hipGetLastError()));
    };
    timeit("copy", 2.0 * 3 * PLANE * 16, [&] { hipLaunchKernelGGL(copy_k, dim3(256 * 16), dim3(256), 0, 0, a, b, 3 * PLANE); });
    timeit("rows_like", (double)N0 * N0 * 8 + 3.0 * N0 * 2048 * 16, [&] { hipLaunchKernelGGL(rows_like, dim3(N0 / 2), dim3(256), 0, 0, img, b); });
    timeit("cols_like", 2.0 * PLANE * 16, [&] { hipLaunchKernelGGL(cols_like, dim3(NP), dim3(512), 0, 0, a, b); });
    timeit("cols_x3", 2.0 * 3 * PLANE * 16, [&] { for (int p = 0; p < 3; ++p) hipLaunchKernelGGL(cols_like, dim3(NP), dim3(512), 0, 0, a + p * PLANE, b + p * PLANE); });
    timeit("omega_like", 3.0 * PLANE * 16, [&] { hipLaunchKernelGGL(omega_like, dim3(((Nhp + 15) / 16) * 8), dim3(64), 0, 0, a, b); });
    return 0;
}
