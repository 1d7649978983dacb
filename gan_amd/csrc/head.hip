// Critic head: the 256 -> 128 -> 64 -> 1 perceptron applied to the pooled graph vector
// (models/discriminator.py:36-40,98-100: self.mlp = Linear, ReLU, Linear, ReLU; self.out = Linear).
//
// One row through three small layers is ~41k multiply-adds - nothing - but as library calls it is
// 3 GEMV launches forward and a dozen launches backward (grad-input, grad-weight, bias and ReLU
// kernels), ten times per training step.  Here the forward pass, the backward pass and the two
// passes the gradient penalty needs (the input gradient ds/dr and the weight gradient of <ds/dr, u>)
// are one single-workgroup launch each.  Sums run in a fixed order: results are reproducible.
#include <cstdint>

#include "common.h"

namespace aig {
namespace {

constexpr int HD_IN = 256, HD_H1 = 128, HD_H2 = 64;
constexpr int HD_THREADS = 256;
constexpr int HD_FWD_THREADS = 1024;  // forward pass: 16 waves, so that every weight row is in flight at once

__device__ __forceinline__ float wave_sum(float v) { return wave_sum_butterfly(v); }

// Sixteen waves: wave w owns rows 4w .. 4w+3 and 64 + 4w .. + 3 of the first layer and rows 4w .. 4w+3 of the second.
// Everything a wave needs from memory - its ten weight rows, their biases, the input - is requested before anything
// is used: one memory round trip per launch (requested layer by layer, bias behind the sum, it was five, and the
// kernel took 9 us of which the arithmetic is nothing).
__global__ __launch_bounds__(HD_FWD_THREADS) void k_head_fwd(const float* __restrict__ r, const float* __restrict__ W1,
                                                         const float* __restrict__ b1, const float* __restrict__ W2,
                                                         const float* __restrict__ b2, const float* __restrict__ W3,
                                                         const float* __restrict__ b3, float* __restrict__ score,
                                                         float* __restrict__ act /*[128 + 64]*/) {
    static_assert(HD_FWD_THREADS == 1024 && HD_H1 == 128 && HD_H2 == 64, "rows are dealt to sixteen waves");
    __shared__ __attribute__((aligned(16))) float s_a1[HD_H1];
    __shared__ float s_a2[HD_H2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 rv = reinterpret_cast<const float4*>(r)[lane];
    float4 w1v[2][4];
    float b1v[2][4], b2v[4];
    float2 w2v[4];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 64 * it + 4 * wave + u;
            w1v[it][u] = reinterpret_cast<const float4*>(W1 + (size_t)row * HD_IN)[lane];
            b1v[it][u] = b1[row];
        }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        w2v[u] = reinterpret_cast<const float2*>(W2 + (size_t)(4 * wave + u) * HD_H1)[lane];
        b2v[u] = b2[4 * wave + u];
    }
    const float w3v = W3[lane], b3v = b3[0];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 w = w1v[it][u];
            const float s = wave_sum(w.x * rv.x + w.y * rv.y + w.z * rv.z + w.w * rv.w);
            if (lane == 0) s_a1[64 * it + 4 * wave + u] = fmaxf(s + b1v[it][u], 0.f);
        }
    __syncthreads();
    const float2 av = reinterpret_cast<const float2*>(s_a1)[lane];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float s = wave_sum(w2v[u].x * av.x + w2v[u].y * av.y);
        if (lane == 0) s_a2[4 * wave + u] = fmaxf(s + b2v[u], 0.f);
    }
    __syncthreads();
    if (tid < 64) {
        const float s = wave_sum(w3v * s_a2[tid]);
        if (tid == 0) score[0] = s + b3v;
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
// (eight waves - the rows a wave multiplies and a column of W2 per thread are requested before anything is used, as in
// the forward pass, and eight waves have the registers to hold them)
constexpr int HD_GR_THREADS = 512;
__global__ __launch_bounds__(HD_GR_THREADS) void k_head_gr_wgrad(const float* __restrict__ u, const float* __restrict__ W1,
                                                                 const float* __restrict__ W2, const float* __restrict__ W3,
                                                                 const float* __restrict__ act, float* __restrict__ dW1,
                                                                 float* __restrict__ dW2, float* __restrict__ dW3,
                                                                 float out_scale /* every output times this, rounded once more */) {
    static_assert(HD_H1 == 4 * 4 * (HD_GR_THREADS / 64) && HD_H2 == 2 * 4 * (HD_GR_THREADS / 64), "rows are dealt to eight waves");
    __shared__ __attribute__((aligned(16))) float s_u[HD_IN];
    __shared__ __attribute__((aligned(16))) float s_t1[HD_H1];   // m1 * (W1 u)
    __shared__ float s_p1[HD_H1], s_p2[HD_H2], s_m1[HD_H1], s_m2[HD_H2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 uv = reinterpret_cast<const float4*>(u)[lane];
    float4 w1v[4][4];       // rows 32 it + 4 wave + q of W1
    float2 w2v[2][4];       // rows 32 it + 4 wave + q of W2
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int q = 0; q < 4; ++q) w1v[it][q] = reinterpret_cast<const float4*>(W1 + (size_t)(32 * it + 4 * wave + q) * HD_IN)[lane];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int q = 0; q < 4; ++q) w2v[it][q] = reinterpret_cast<const float2*>(W2 + (size_t)(32 * it + 4 * wave + q) * HD_H1)[lane];
    float w2col[HD_H2];          // column tid % 128 of W2, for p1 below (used by the first two waves)
#pragma unroll
    for (int j = 0; j < HD_H2; ++j) w2col[j] = W2[(size_t)j * HD_H1 + (tid & 127)];
    if (tid < HD_IN) s_u[tid] = u[tid];
    if (tid < HD_H1) s_m1[tid] = act[tid] > 0.f ? 1.f : 0.f;
    else if (tid < HD_H1 + HD_H2) {
        const int j = tid - HD_H1;
        const float m2 = act[tid] > 0.f ? 1.f : 0.f;
        s_m2[j] = m2;
        s_p2[j] = m2 * W3[j];
    }
    __syncthreads();
    if (tid < HD_H1) {  // p1[k] = m1[k] * sum_j W2[j][k] p2[j], j ascending
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < HD_H2; ++j) s += w2col[j] * s_p2[j];
        s_p1[tid] = s_m1[tid] * s;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it)      // t1[k] = m1[k] * sum_c W1[k][c] u[c]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 w = w1v[it][q];
            const float s = wave_sum(w.x * uv.x + w.y * uv.y + w.z * uv.z + w.w * uv.w);
            const int row = 32 * it + 4 * wave + q;
            if (lane == 0) s_t1[row] = s_m1[row] * s;
        }
    __syncthreads();
    for (int e = tid; e < HD_H1 * HD_IN; e += HD_GR_THREADS) dW1[e] = (s_p1[e / HD_IN] * s_u[e % HD_IN]) * out_scale;
    for (int e = tid; e < HD_H2 * HD_H1; e += HD_GR_THREADS) dW2[e] = (s_p2[e / HD_H1] * s_t1[e % HD_H1]) * out_scale;
    const float2 tv = reinterpret_cast<const float2*>(s_t1)[lane];
#pragma unroll
    for (int it = 0; it < 2; ++it)      // dW3[j] = m2[j] * sum_k W2[j][k] t1[k]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float s = wave_sum(w2v[it][q].x * tv.x + w2v[it][q].y * tv.y);
            const int row = 32 * it + 4 * wave + q;
            if (lane == 0) dW3[row] = (s_m2[row] * s) * out_scale;
        }
}

}  // namespace
}  // namespace aig

using namespace aig;

extern "C" int aig_head_fwd(const float* r, const float* w1, const float* b1, const float* w2, const float* b2,
                            const float* w3, const float* b3, float* score, float* act, void* stream) {
    AIG_REQUIRE(r && w1 && b1 && w2 && b2 && w3 && b3 && score && act, "aig_head_fwd: null pointer");
    AIG_REQUIRE(((reinterpret_cast<uintptr_t>(r) | reinterpret_cast<uintptr_t>(w1)) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(w2) & 7) == 0,
                "aig_head_fwd: r and w1 must be 16-byte aligned, w2 8-byte aligned (rows are read as vectors)");
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

extern "C" int aig_head_input_grad_wgrad_scaled(const float* u, const float* w1, const float* w2, const float* w3,
                                                const float* act, float* d_w1, float* d_w2, float* d_w3, float out_scale,
                                                void* stream) {
    AIG_REQUIRE(u && w1 && w2 && w3 && act && d_w1 && d_w2 && d_w3, "aig_head_input_grad_wgrad: null pointer");
    AIG_REQUIRE(((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(w1)) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(w2) & 7) == 0,
                "aig_head_input_grad_wgrad: u and w1 must be 16-byte aligned, w2 8-byte aligned (rows are read as vectors)");
    hipLaunchKernelGGL(k_head_gr_wgrad, dim3(1), dim3(HD_GR_THREADS), 0, static_cast<hipStream_t>(stream), u, w1, w2, w3, act,
                       d_w1, d_w2, d_w3, out_scale);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

extern "C" int aig_head_input_grad_wgrad(const float* u, const float* w1, const float* w2, const float* w3,
                                         const float* act, float* d_w1, float* d_w2, float* d_w3, void* stream) {
    return aig_head_input_grad_wgrad_scaled(u, w1, w2, w3, act, d_w1, d_w2, d_w3, 1.0f, stream);
}
