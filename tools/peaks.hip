// Achievable peaks of the GPU this runs on (tuning aid, not part of the library): a float4 stream copy and a
// loop of independent f32 32x32x2 MFMA instructions.  Prints one JSON object.
//   hipcc --offload-arch=gfx950 -O3 -o tools/peaks tools/peaks.hip && ./tools/peaks
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float* __restrict__ sink, size_t n) {
    float s = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {   // four loads in flight per thread
        const float4 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        s += v0.x + v0.y + v0.z + v0.w + v1.x + v1.y + v1.z + v1.w + v2.x + v2.y + v2.z + v2.w + v3.x + v3.y + v3.z + v3.w;
    }
    for (; i < n; i += stride) s += a[i].x;
    if (s == 123.456f) sink[0] = s;
}
__global__ __launch_bounds__(256) void k_mfma(float* __restrict__ out, int iters) {
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const float a = (float)threadIdx.x, b = 1.0f;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += c0[k] + c1[k] + c2[k] + c3[k];
    if (s == 123.456f) out[0] = s;
}

int main() {
    const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
    float4 *a, *b;
    float* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 256);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms, best_copy = 1e9f, best_read = 1e9f, best_mfma = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_copy, dim3(256 * 32), dim3(256), 0, 0, a, b, n4);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best_copy) best_copy = ms;
        for (int g = 4; g <= 64; g *= 2) {   // best over a few grid sizes
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_read, dim3(256 * g), dim3(256), 0, 0, a, sink, n4);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best_read) best_read = ms;
        }
        hipEventRecord(e0);
        hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best_copy) best_copy = ms;
    }
    const int iters = 20000, blocks = 256 * 8;   // 8 blocks of 4 waves per CU
    float sust = 0.f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best_mfma) best_mfma = ms;
        if (rep == 5) sust = ms;
    }
    const double flops = (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * (32.0 * 32 * 2 * 2);
    printf("{\"hbm_copy_GBs\": %.0f, \"hbm_read_GBs\": %.0f, \"mfma_f32_TFs_best\": %.1f, \"mfma_f32_TFs_last\": %.1f, "
           "\"note\": \"1 GiB float4 copy (read + write counted) and read; %d x 4 independent v_mfma_f32_32x32x2f32 per wave, %d waves\"}\n",
           2.0 * bytes / best_copy / 1e6, (double)bytes / best_read / 1e6, flops / best_mfma / 1e9, flops / sust / 1e9, iters,
           blocks * 4);
    return 0;
}
