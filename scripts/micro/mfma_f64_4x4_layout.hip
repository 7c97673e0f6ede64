// Probe of the operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950: one wave, for every pair (L, M) of lanes the
// A operand is 1 in lane L (else 0) and the B operand is 1 in lane M; the lanes where D != 0 tell which (block, i, k) lane L
// holds and which (block, k, j) lane M holds.   hipcc --offload-arch=gfx950 -O3 -o /tmp/lay scripts/micro/mfma_f64_4x4_layout.hip && /tmp/lay
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) probe(unsigned long long* out)
{
    const int lane = threadIdx.x;
    for (int L = 0; L < 64; ++L)
        for (int M = 0; M < 64; ++M) {
            const double a = lane == L ? 1.0 : 0.0, b = lane == M ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) out[L * 64 + M] = m;
        }
}
int main()
{
    unsigned long long* d; (void)hipMalloc(&d, 4096 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    std::vector<unsigned long long> h(4096);
    (void)hipMemcpy(h.data(), d, 4096 * 8, hipMemcpyDeviceToHost);
    // for each A lane L: the set of B lanes it pairs with, and the output lane of each pairing
    for (int L = 0; L < 64; ++L) {
        printf("A lane %2d pairs with B lanes -> D lane:", L);
        for (int M = 0; M < 64; ++M) {
            if (!h[L * 64 + M]) continue;
            int dl = __builtin_ctzll(h[L * 64 + M]);
            printf(" %d->%d%s", M, dl, __builtin_popcountll(h[L * 64 + M]) > 1 ? "(+)" : "");
        }
        printf("\n");
    }
    return 0;
}
