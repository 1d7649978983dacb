// Timing aid: f32 matrix-core rate of ONE wave per SIMD with 1, 2 and 4 independent accumulators
// (how much a chain of dependent v_mfma_f32_32x32x2_f32 costs), and the clock it runs at.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* clk) {
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[u % NACC], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) s += c[u][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
template <int NACC> void run(float* out, long long* clk, int blocks) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f; long long h[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, clk); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double fl = (double)blocks * 4 * iters * 8 * 4096.0;
    printf("blocks %d accs %d: %.1f TFLOP/s, %.1f cycles per product, clock %.2f GHz\n", blocks, NACC, fl / best / 1e9,
           (double)h[0] / (iters * 8.0), (double)h[0] / (h[1] * 10.0));
}
int main() {
    float* out; long long* clk; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 16);
    for (int blocks : {256, 16}) { run<1>(out, clk, blocks); run<2>(out, clk, blocks); run<4>(out, clk, blocks); }
    return 0;
}
