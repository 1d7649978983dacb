// Timing aid: what slows a chain of v_mfma_f32_32x32x2_f32 below 64 cycles per product with one wave per SIMD?
// V1: weights from 64 distinct registers.  V2: + tile fragments from LDS (ds_read_b128 a group of 4 products
// ahead).  V4: V2 + 8 independent v_add_f32 of the same wave after every product (do they hide behind it?).
// V2 and V4 again with two waves per SIMD (do they hide behind the OTHER wave's products?).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int H = 128;
template <int V>
__global__ __launch_bounds__(512) void k(const float* w, float* out, int iters, long long* clk) {
    __shared__ __attribute__((aligned(16))) float tile[64 * H];
    const int tid = threadIdx.x, lane = tid & 63, frow = lane & 31, fh = lane >> 5;
    for (int i = tid; i < 64 * H; i += blockDim.x) tile[i] = i * 1e-4f;
    float4 wr[3][16];
    for (int l = 0; l < 3; ++l)
        for (int g = 0; g < 16; ++g) wr[l][g] = *reinterpret_cast<const float4*>(w + (l * 128 + tid % 128) * H + 8 * g + 4 * fh);
    __syncthreads();
    f32x16 c = {0};
    float junk[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        const float* in = tile + (i & 1) * 32 * H;
        float4 a = *reinterpret_cast<const float4*>(in + frow * H + 4 * (fh ^ (frow & 15)));
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            float4 n = a;
            if (V >= 2) n = *reinterpret_cast<const float4*>(in + frow * H + 4 * ((2 * ((g + 1) & 15) + fh) ^ (frow & 15)));
            float4 b = wr[0][g];
            if (V == 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(u == 0 ? b.x : u == 1 ? b.y : u == 2 ? b.z : b.w,
                                                             u == 0 ? a.x : u == 1 ? a.y : u == 2 ? a.z : a.w, c, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) asm volatile("v_add_f32 %0, %0, %1" : "+v"(junk[e]) : "v"(b.x));
                }
            } else {
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, c, 0, 0, 0);
            }
            a = n;
            if (V >= 2 && V != 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (V != 4) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c[r];
    for (int e = 0; e < 8; ++e) s += junk[e];
    for (int l = 1; l < 3; ++l) for (int g = 0; g < 16; ++g) s += wr[l][g].x;
    out[blockIdx.x * 512 + tid] = s;
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
template <int V> void run(const float* w, float* out, long long* clk, int threads = 256) {
    const int iters = 600, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f; long long h[2];
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, w, out, iters, clk); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double fl = (double)blocks * (threads / 64) * iters * 64 * 4096.0;
    printf("variant %d, %d waves per SIMD: %.1f TFLOP/s, %.1f cycles per product and wave, clock %.2f GHz\n", V, threads / 256, fl / best / 1e9,
           (double)h[0] / (iters * 64.0), (double)h[0] / (h[1] * 10.0));
}
int main() {
    float *w, *out; long long* clk; hipMalloc(&w, 3 * 128 * H * 4); hipMemset(w, 0, 3 * 128 * H * 4);
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 16);
    run<1>(w, out, clk); run<2>(w, out, clk); run<4>(w, out, clk);
    run<2>(w, out, clk, 512); run<4>(w, out, clk, 512);
    return 0;
}
