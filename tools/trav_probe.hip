// Latency probe for single-workgroup level-synchronous loops (tuning aid, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ __launch_bounds__(1024) void probe(int mode, int iters, const int* __restrict__ chase, int* __restrict__ sink,
                                              long long* __restrict__ out) {
    __shared__ int s_cnt[3];
    __shared__ int s_q[2048];
    const int tid = threadIdx.x;
    if (tid < 3) s_cnt[tid] = 0;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int p = tid;
    for (int it = 0; it < iters; ++it) {
        if (mode >= 1) __syncthreads();
        if (mode >= 2) p = chase[p];                       // dependent global load
        if (mode >= 3) { int i = atomicAdd(&s_cnt[it % 3], 1); s_q[i & 2047] = p; }
        if (mode >= 4) sink[p] = it;                       // fire-and-forget store
        if (mode >= 5) { if (tid == 0) s_cnt[(it + 2) % 3] = 0; int c = s_cnt[(it + 1) % 3]; if (c < 0) break; }
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (p == -7) sink[0] = p;
}
int main() {
    const int n = 400000, iters = 1000;
    std::vector<int> perm(n); std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 g(1); std::shuffle(perm.begin(), perm.end(), g);
    int *d_chase, *d_sink; long long* d_out;
    hipMalloc(&d_chase, n * 4); hipMalloc(&d_sink, n * 4); hipMalloc(&d_out, 16);
    hipMemcpy(d_chase, perm.data(), n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode <= 5; ++mode) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(1024), 0, 0, mode, iters, d_chase, d_sink, d_out);
            long long h[2]; hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
            printf("rep %d mode %d: %.1f cycles/iter, %.3f us/iter, clock %.2f GHz\n", rep, mode, (double)h[0] / iters,
                   (double)h[1] / 100.0 / iters, (double)h[0] / ((double)h[1] / 100.0) / 1000.0);
        }
    return 0;
}
