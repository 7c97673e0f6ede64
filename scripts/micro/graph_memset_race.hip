// Reproducer attempt for the stale fill seen in round 3 (docs/LOG.md, "a runtime bug found by the bitwise post-check"): a hipMemsetAsync(0)
// node captured into a hipGraph, several such graphs (one per host thread, own stream, own buffers) replayed at the same time, left 8-byte
// entries holding a stale pointer-like pattern instead of 0 "now and then" (ROCm 7.2.0, MI355X, HIP runtime of PyTorch 2.10.0+rocm7.0).
// The library no longer depends on it (zero_f64 / set_i32 kernel nodes; SFFT_SOL_MEMSET=1 restores the memset); this file is the minimal
// form of the pattern, to be run against a new ROCm before that workaround is removed:
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -pthread -o /tmp/graph_memset_race scripts/micro/graph_memset_race.hip && /tmp/graph_memset_race [threads] [replays]
// Each thread: stream, buffer X [n] doubles, capture { poison<<<>>>(X, tagged NaN-free garbage); hipMemsetAsync(X, 0, bytes); touch<<<>>>(X + hole) }
// once, then replay the graph `replays` times; after every replay a checker kernel counts the entries of X outside the touched range that are
// not exactly 0.  Prints the number of bad replays per thread; exit status 1 if any.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <atomic>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void poison(double* x, int n, unsigned long long tag)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = __longlong_as_double((long long)(0x00007f00deadbe00ull + tag * 4096ull + (unsigned)i));      // a pointer-like, denormal-sized pattern
}
__global__ void touch(double* x, int lo, int hi)        // what the solver does after the memset: writes part of the vector
{
    const int i = lo + blockIdx.x * 256 + threadIdx.x;
    if (i < hi) x[i] = 1.0 + i;
}
__global__ void check(const double* x, int n, int lo, int hi, unsigned int* bad)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && (i < lo || i >= hi) && __double_as_longlong(x[i]) != 0) atomicAdd(bad, 1u);
}

int main(int argc, char** argv)
{
    const int nthreads = argc > 1 ? atoi(argv[1]) : 4, replays = argc > 2 ? atoi(argv[2]) : 2000;
    const int n = 1740, lo = 5, hi = 1735;          // the Solution vector of config 2: 5 removed unknowns stay 0
    std::atomic<int> total_bad(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t]() {
            CK(hipSetDevice(0));
            hipStream_t s; CK(hipStreamCreate(&s));
            double* x; unsigned int* bad; unsigned int* hbad;
            CK(hipMalloc(&x, n * sizeof(double))); CK(hipMalloc(&bad, 4)); CK(hipHostMalloc(&hbad, 4));
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            poison<<<(n + 255) / 256, 256, 0, s>>>(x, n, (unsigned long long)t);
            CK(hipMemsetAsync(x, 0, n * sizeof(double), s));
            touch<<<(hi - lo + 255) / 256, 256, 0, s>>>(x, lo, hi);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            int bad_replays = 0;
            for (int r = 0; r < replays; ++r) {
                CK(hipMemsetAsync(bad, 0, 4, s));
                CK(hipGraphLaunch(ge, s));
                check<<<(n + 255) / 256, 256, 0, s>>>(x, n, lo, hi, bad);
                CK(hipMemcpyAsync(hbad, bad, 4, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
                if (*hbad) ++bad_replays;
            }
            printf("thread %d: %d of %d replays left non-zero entries behind the memset node\n", t, bad_replays, replays);
            total_bad += bad_replays;
            hipGraphExecDestroy(ge); hipGraphDestroy(g); hipFree(x); hipFree(bad); hipHostFree(hbad); hipStreamDestroy(s);
        });
    for (auto& q : th) q.join();
    int rv = 0; CK(hipRuntimeGetVersion(&rv));
    printf("HIP runtime %d: %s\n", rv, total_bad ? "STALE FILL REPRODUCED" : "not reproduced in this form");
    return total_bad ? 1 : 0;
}
