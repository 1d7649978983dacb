// Weight gradient of the critic's dense projections: dW[j][f] = sum_n g[n][j] * x[n][f]
// (g = gradient w.r.t. x W^T, [n,128]; x [n,F]; dW [128,F] in PyTorch's [out,in] layout).
// The reduction runs over the NODES (n ~ 1e5..1e6) while the output is tiny, a shape library GEMMs
// serve with a handful of tiles; here the node range is split over the whole chip, every block
// accumulates a full [128,F] partial on the f32 matrix cores straight from global memory (both
// operands are read as 128-byte row segments, no staging), and the partials are added in block
// order, so the result is reproducible.
#include <climits>

#include "common.h"

namespace aig {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int H = AIG_HIDDEN;
constexpr int WG_BLOCKS_MAX = 256;  // one partial per CU

// F == 128: wave w of a block owns rows 32w..32w+31 of dW and all four 32-column tiles.
// MFMA 32x32x2: A[i][k] = g[n0+k][32w+i], B[k][c] = x[n0+k][32t+c], k = lane>>5, i/c = lane&31.
__global__ __launch_bounds__(256) void k_wgrad128(int n, const float* __restrict__ g, const float* __restrict__ x,
                                                  float* __restrict__ partial /*[grid][128][128]*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = lane & 31, kh = lane >> 5;
    const int chunk = ((n + gridDim.x - 1) / gridDim.x + 1) & ~1;  // even: MFMA steps take node pairs
    const int n0 = blockIdx.x * chunk, n1 = min(n, n0 + chunk);
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    // four node pairs per trip: all twenty loads are issued before the first MFMA needs one
    for (int nn = n0; nn < n1; nn += 8) {
        float a[4], b[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = nn + 2 * u + kh;
            const bool ok = row < n1;
            const size_t off = (size_t)(ok ? row : n0) * H;
            a[u] = ok ? g[off + 32 * w + col] : 0.f;
            const float* xr = x + off + col;
            b[u][0] = ok ? xr[0] : 0.f;
            b[u][1] = ok ? xr[32] : 0.f;
            b[u][2] = ok ? xr[64] : 0.f;
            b[u][3] = ok ? xr[96] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][1], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][2], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][3], acc[3], 0, 0, 0);
        }
    }
    float* out = partial + (size_t)blockIdx.x * H * H;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;  // accumulator register -> tile row
            out[(size_t)(32 * w + row) * H + 32 * t + col] = acc[t][r];
        }
}

// small F (the first layer has F = 2): thread (j, f) walks the block's node chunk
template <int F>
__global__ __launch_bounds__(128 * F) void k_wgrad_small(int n, const float* __restrict__ g,
                                                         const float* __restrict__ x,
                                                         float* __restrict__ partial /*[grid][128][F]*/) {
    const int j = threadIdx.x % H, f = threadIdx.x / H;
    const int chunk = (n + gridDim.x - 1) / gridDim.x;
    const int n0 = blockIdx.x * chunk, n1 = min(n, n0 + chunk);
    float acc = 0.f;
#pragma unroll 8
    for (int nn = n0; nn < n1; ++nn) acc = fmaf(g[(size_t)nn * H + j], x[(size_t)nn * F + f], acc);
    partial[(size_t)blockIdx.x * H * F + j * F + f] = acc;
}

// out[e] = sum_b partial[b][e], b ascending per thread stripe then a fixed tree
__global__ __launch_bounds__(64) void k_wgrad_reduce(int nb, int elems, const float* __restrict__ partial,
                                                      float* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= elems) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains keep loads in flight
    int b = 0;
    for (; b + 4 <= nb; b += 4) {
        s0 += partial[(size_t)b * elems + e];
        s1 += partial[(size_t)(b + 1) * elems + e];
        s2 += partial[(size_t)(b + 2) * elems + e];
        s3 += partial[(size_t)(b + 3) * elems + e];
    }
    for (; b < nb; ++b) s0 += partial[(size_t)b * elems + e];
    out[e] = (s0 + s1) + (s2 + s3);
}

int wg_blocks(int64_t n, int per_block) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > WG_BLOCKS_MAX) b = WG_BLOCKS_MAX;
    return (int)b;
}

}  // namespace
}  // namespace aig

using namespace aig;

extern "C" size_t aig_wgrad_ws_bytes(int64_t n, int64_t F) {
    (void)n;
    return align_up((size_t)WG_BLOCKS_MAX * H * (size_t)(F > 0 ? F : 1) * 4) + 256;
}

extern "C" int aig_wgrad(int64_t n, int64_t F, const float* g, const float* x, float* dW, void* ws,
                         size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX, "aig_wgrad: n=%lld out of range", (long long)n);
    AIG_REQUIRE(F == 128 || F == 1 || F == 2 || F == 4, "aig_wgrad: input width %lld not built (1, 2, 4 or 128)",
                (long long)F);
    AIG_REQUIRE(dW, "aig_wgrad: null output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) {
        AIG_HIP_CHECK(hipMemsetAsync(dW, 0, (size_t)H * F * 4, s));
        return AIG_OK;
    }
    AIG_REQUIRE(g && x && ws, "aig_wgrad: null pointer");
    if (ws_bytes < aig_wgrad_ws_bytes(n, F)) return fail(AIG_EWORKSPACE, "aig_wgrad: workspace too small");
    float* partial = static_cast<float*>(ws);
    int nb;
    if (F == 128) {
        nb = wg_blocks(n, 256);
        hipLaunchKernelGGL(k_wgrad128, dim3(nb), dim3(256), 0, s, (int)n, g, x, partial);
    } else if (F == 1) {
        nb = wg_blocks(n, 1024);
        hipLaunchKernelGGL(k_wgrad_small<1>, dim3(nb), dim3(128), 0, s, (int)n, g, x, partial);
    } else if (F == 2) {
        nb = wg_blocks(n, 1024);
        hipLaunchKernelGGL(k_wgrad_small<2>, dim3(nb), dim3(256), 0, s, (int)n, g, x, partial);
    } else {
        nb = wg_blocks(n, 1024);
        hipLaunchKernelGGL(k_wgrad_small<4>, dim3(nb), dim3(512), 0, s, (int)n, g, x, partial);
    }
    AIG_LAUNCH_CHECK();
    const int elems = (int)(H * F);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((elems + 63) / 64), dim3(64), 0, s, nb, elems, partial, dW);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}
