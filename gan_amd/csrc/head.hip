// Critic head: the 256 -> 128 -> 64 -> 1 perceptron applied to the pooled graph vector
// (models/discriminator.py:36-40,98-100: self.mlp = Linear, ReLU, Linear, ReLU; self.out = Linear).
//
// One row through three small layers is ~41k multiply-adds - nothing - but as library calls it is
// 3 GEMV launches forward and a dozen launches backward (grad-input, grad-weight, bias and ReLU
// kernels), ten times per training step.  Here the forward pass, the backward pass and the two
// passes the gradient penalty needs (the input gradient ds/dr and the weight gradient of <ds/dr, u>)
// are one single-workgroup launch each.  Sums run in a fixed order: results are reproducible.
#include "common.h"

namespace aig {
namespace {

constexpr int HD_IN = 256, HD_H1 = 128, HD_H2 = 64;
constexpr int HD_THREADS = 256;
constexpr int HD_FWD_THREADS = 1024;  // forward pass: 16 waves, so that every weight row is in flight at once

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// s_a1 / s_a2 <- activations of the two hidden layers for input s_r (all in LDS); wave per output row,
// four rows at a time, any number of waves
__device__ __forceinline__ void head_hidden(const float* __restrict__ W1, const float* __restrict__ b1,
                                            const float* __restrict__ W2, const float* __restrict__ b2,
                                            const float* s_r, float* s_a1, float* s_a2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, step = 4 * (blockDim.x >> 6);
    const float4 rv = reinterpret_cast<const float4*>(s_r)[lane];
    for (int row = wave * 4; row < HD_H1; row += step) {  // four rows per wave at a time
        float p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 w = reinterpret_cast<const float4*>(W1 + (size_t)(row + u) * HD_IN)[lane];
            p[u] = w.x * rv.x + w.y * rv.y + w.z * rv.z + w.w * rv.w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float s = wave_sum(p[u]);
            if (lane == 0) s_a1[row + u] = fmaxf(s + b1[row + u], 0.f);
        }
    }
    __syncthreads();
    const float2 av = reinterpret_cast<const float2*>(s_a1)[lane];
    for (int row = wave * 4; row < HD_H2; row += step) {
        float p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float2 w = reinterpret_cast<const float2*>(W2 + (size_t)(row + u) * HD_H1)[lane];
            p[u] = w.x * av.x + w.y * av.y;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float s = wave_sum(p[u]);
            if (lane == 0) s_a2[row + u] = fmaxf(s + b2[row + u], 0.f);
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(HD_FWD_THREADS) void k_head_fwd(const float* __restrict__ r, const float* __restrict__ W1,
                                                         const float* __restrict__ b1, const float* __restrict__ W2,
                                                         const float* __restrict__ b2, const float* __restrict__ W3,
                                                         const float* __restrict__ b3, float* __restrict__ score,
                                                         float* __restrict__ act /*[128 + 64]*/) {
    __shared__ __attribute__((aligned(16))) float s_r[HD_IN];
    __shared__ __attribute__((aligned(16))) float s_a1[HD_H1];
    __shared__ float s_a2[HD_H2];
    const int tid = threadIdx.x;
    if (tid < HD_IN) s_r[tid] = r[tid];
    __syncthreads();
    head_hidden(W1, b1, W2, b2, s_r, s_a1, s_a2);
    if (tid < 64) {
        const float s = wave_sum(W3[tid] * s_a2[tid]);
        if (tid == 0) score[0] = s + b3[0];
    }
    if (tid < HD_H1) act[tid] = s_a1[tid];
    else if (tid < HD_H1 + HD_H2) act[tid] = s_a2[tid - HD_H1];
}

// Backward of the forward pass for d(loss)/d(score) = ds[0].  Any output pointer may be null.
__global__ __launch_bounds__(HD_THREADS) void k_head_bwd(const float* __restrict__ ds, const float* __restrict__ r,
                                                         const float* __restrict__ W1, const float* __restrict__ W2,
                                                         const float* __restrict__ W3, const float* __restrict__ act,
                                                         float* __restrict__ d_r, float* __restrict__ dW1,
                                                         float* __restrict__ db1, float* __restrict__ dW2,
                                                         float* __restrict__ db2, float* __restrict__ dW3,
                                                         float* __restrict__ db3) {
    __shared__ float s_r[HD_IN], s_a1[HD_H1], s_a2[HD_H2], s_d1[HD_H1], s_d2[HD_H2];
    const int tid = threadIdx.x;
    const float g = ds[0];
    s_r[tid] = r[tid];
    if (tid < HD_H1) s_a1[tid] = act[tid];
    else if (tid < HD_H1 + HD_H2) s_a2[tid - HD_H1] = act[tid];
    __syncthreads();
    if (tid < HD_H2) {
        const float a2 = s_a2[tid];
        const float d2 = a2 > 0.f ? g * W3[tid] : 0.f;
        s_d2[tid] = d2;
        if (dW3) dW3[tid] = g * a2;
        if (db2) db2[tid] = d2;
    }
    if (tid == 0 && db3) db3[0] = g;
    __syncthreads();
    if (tid < HD_H1) {  // d_a1[k] = relu'(a1[k]) * sum_j W2[j][k] d_a2[j]
        float s = 0.f;
        for (int j = 0; j < HD_H2; ++j) s += W2[(size_t)j * HD_H1 + tid] * s_d2[j];
        const float d1 = s_a1[tid] > 0.f ? s : 0.f;
        s_d1[tid] = d1;
        if (db1) db1[tid] = d1;
    }
    if (dW2)
        for (int e = tid; e < HD_H2 * HD_H1; e += HD_THREADS) dW2[e] = s_d2[e / HD_H1] * s_a1[e % HD_H1];
    __syncthreads();
    if (dW1) {
        const float rc = s_r[tid];
        for (int k = 0; k < HD_H1; ++k) dW1[(size_t)k * HD_IN + tid] = s_d1[k] * rc;
    }
    if (d_r) {
        float s = 0.f;
        for (int k = 0; k < HD_H1; ++k) s += W1[(size_t)k * HD_IN + tid] * s_d1[k];
        d_r[tid] = s;
    }
}

// With g_r = ds/dr = W1^T (m1 * W2^T (m2 * W3^T)) (m1, m2: the ReLU masks of the forward pass), the gradient
// of <g_r, u> w.r.t. the three weight matrices (the biases only move the masks: no gradient).
__global__ __launch_bounds__(HD_THREADS) void k_head_gr_wgrad(const float* __restrict__ u, const float* __restrict__ W1,
                                                              const float* __restrict__ W2, const float* __restrict__ W3,
                                                              const float* __restrict__ act, float* __restrict__ dW1,
                                                              float* __restrict__ dW2, float* __restrict__ dW3) {
    __shared__ __attribute__((aligned(16))) float s_u[HD_IN];
    __shared__ __attribute__((aligned(16))) float s_t1[HD_H1];   // m1 * (W1 u)
    __shared__ float s_p1[HD_H1], s_p2[HD_H2], s_m1[HD_H1], s_m2[HD_H2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s_u[tid] = u[tid];
    if (tid < HD_H1) s_m1[tid] = act[tid] > 0.f ? 1.f : 0.f;
    else if (tid < HD_H1 + HD_H2) {
        const int j = tid - HD_H1;
        const float m2 = act[tid] > 0.f ? 1.f : 0.f;
        s_m2[j] = m2;
        s_p2[j] = m2 * W3[j];
    }
    __syncthreads();
    if (tid < HD_H1) {  // p1[k] = m1[k] * sum_j W2[j][k] p2[j]
        float s = 0.f;
        for (int j = 0; j < HD_H2; ++j) s += W2[(size_t)j * HD_H1 + tid] * s_p2[j];
        s_p1[tid] = s_m1[tid] * s;
    }
    const float4 uv = reinterpret_cast<const float4*>(s_u)[lane];
    for (int row = wave * 4; row < HD_H1; row += 16) {  // t1[k] = m1[k] * sum_c W1[k][c] u[c]
        float p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w = reinterpret_cast<const float4*>(W1 + (size_t)(row + q) * HD_IN)[lane];
            p[q] = w.x * uv.x + w.y * uv.y + w.z * uv.z + w.w * uv.w;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float s = wave_sum(p[q]);
            if (lane == 0) s_t1[row + q] = s_m1[row + q] * s;
        }
    }
    __syncthreads();
    {
        const float uc = s_u[tid];
        for (int k = 0; k < HD_H1; ++k) dW1[(size_t)k * HD_IN + tid] = s_p1[k] * uc;
    }
    for (int e = tid; e < HD_H2 * HD_H1; e += HD_THREADS) dW2[e] = s_p2[e / HD_H1] * s_t1[e % HD_H1];
    const float2 tv = reinterpret_cast<const float2*>(s_t1)[lane];
    for (int row = wave * 4; row < HD_H2; row += 16) {  // dW3[j] = m2[j] * sum_k W2[j][k] t1[k]
        float p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float2 w = reinterpret_cast<const float2*>(W2 + (size_t)(row + q) * HD_H1)[lane];
            p[q] = w.x * tv.x + w.y * tv.y;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float s = wave_sum(p[q]);
            if (lane == 0) dW3[row + q] = s_m2[row + q] * s;
        }
    }
}

}  // namespace
}  // namespace aig

using namespace aig;

extern "C" int aig_head_fwd(const float* r, const float* w1, const float* b1, const float* w2, const float* b2,
                            const float* w3, const float* b3, float* score, float* act, void* stream) {
    AIG_REQUIRE(r && w1 && b1 && w2 && b2 && w3 && b3 && score && act, "aig_head_fwd: null pointer");
    hipLaunchKernelGGL(k_head_fwd, dim3(1), dim3(HD_FWD_THREADS), 0, static_cast<hipStream_t>(stream), r, w1, b1, w2, b2,
                       w3, b3, score, act);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

extern "C" int aig_head_bwd(const float* d_score, const float* r, const float* w1, const float* w2, const float* w3,
                            const float* act, float* d_r, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                            float* d_w3, float* d_b3, void* stream) {
    AIG_REQUIRE(d_score && r && w1 && w2 && w3 && act, "aig_head_bwd: null pointer");
    hipLaunchKernelGGL(k_head_bwd, dim3(1), dim3(HD_THREADS), 0, static_cast<hipStream_t>(stream), d_score, r, w1, w2, w3,
                       act, d_r, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

extern "C" int aig_head_input_grad_wgrad(const float* u, const float* w1, const float* w2, const float* w3,
                                         const float* act, float* d_w1, float* d_w2, float* d_w3, void* stream) {
    AIG_REQUIRE(u && w1 && w2 && w3 && act && d_w1 && d_w2 && d_w3, "aig_head_input_grad_wgrad: null pointer");
    hipLaunchKernelGGL(k_head_gr_wgrad, dim3(1), dim3(HD_THREADS), 0, static_cast<hipStream_t>(stream), u, w1, w2, w3, act,
                       d_w1, d_w2, d_w3);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}
