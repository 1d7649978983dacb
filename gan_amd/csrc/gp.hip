// Gradient penalty of the critic on an edge-free interpolate, in closed form, and the backward of
// the level readout.
//
// On an empty edge list a GAT layer is elu(x W^T + b) (only the self loop remains, alpha = 1), so
//   s(x^) = head( P elu( elu(x^ W1^T + b1) W2^T + b2 ) ),   P = level readout (mean || max)
// and the penalty (||ds/dx^||_2 - 1)^2 and its gradient w.r.t. W1, b1, W2, b2 and dL/d(ds/dr) have
// closed forms made of four [n,128]x[128,128] products, three node-dimension weight gradients and
// row-wise passes.  Autograd needs ~200 launches of elementwise kernels for the same numbers
// (double backward); here every row-wise pass is one kernel, half-wave (32 lanes x float4) per row.
// The tiny head (256 -> 128 -> 64 -> 1) stays with the caller.
//
// elu'(z) and elu''(z) are taken from the OUTPUT h = elu(z):  h > 0 ? 1 : h + 1  and  h > 0 ? 0 : h + 1.
#include <climits>

#include "common.h"
#include "gat_common.h"

namespace aig {
namespace {

constexpr int H = AIG_HIDDEN;
constexpr int THREADS = 256;
constexpr int HW_PER_BLOCK = THREADS / 32;
constexpr int GP_BLOCKS_MAX = 2048;

__device__ __forceinline__ float d_elu(float h) { return h > 0.f ? 1.f : h + 1.f; }
__device__ __forceinline__ float dd_elu(float h) { return h > 0.f ? 0.f : h + 1.f; }
__device__ __forceinline__ float elu(float z) { return z > 0.f ? z : expm1f(z); }

__device__ __forceinline__ float halfwave_sum(float v) {
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) v += __shfl_xor(v, off, 32);
    return v;
}

struct Rows {
    int begin, end;
};
__device__ __forceinline__ Rows block_rows(int n, const int* __restrict__ row_limit = nullptr) {
    n = limited_rows(n, row_limit);
    const int chunk = (n + gridDim.x - 1) / gridDim.x;
    const int b = blockIdx.x * chunk;
    return {b, min(n, b + chunk)};
}

// d_h[i,c] = w[i] g[c] + [argmax[level(i),c] == i] g[128+c] / (L+1)
__global__ __launch_bounds__(THREADS) void k_pool_bwd(int n, int L, const int* __restrict__ meta, const float* __restrict__ g,
                                                      const int* __restrict__ node_level, const float* __restrict__ w,
                                                      const int* __restrict__ argmax, float* __restrict__ d_h) {
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const float4 g_mean = reinterpret_cast<const float4*>(g)[l], g_max = reinterpret_cast<const float4*>(g + H)[l];
    const float inv_levels = 1.f / (float)(level_count(L, meta));
    const int4* argmax4 = reinterpret_cast<const int4*>(argmax);
    const Rows r = block_rows(n);
    for (int i = r.begin + hw; i < r.end; i += HW_PER_BLOCK)
        reinterpret_cast<float4*>(d_h + (size_t)i * H)[l] =
            readout_pull(i, node_level[i], w[i], inv_levels, g_mean, g_max, argmax4, l);
}

// block partial of column sums: red[hw][width] summed over the half-waves in fixed order
template <int WIDTH>
__device__ __forceinline__ void block_colsum(float (*red)[WIDTH], float* __restrict__ partial) {
    __syncthreads();
    for (int c = threadIdx.x; c < WIDTH; c += THREADS) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < HW_PER_BLOCK; ++k) s += red[k][c];
        partial[(size_t)blockIdx.x * WIDTH + c] = s;
    }
}

// adjoint of k_pool_bwd: out[c] = sum_i w[i] gg[i,c], out[128+c] = sum_i [argmax == i] gg[i,c] / (L+1)
__global__ __launch_bounds__(THREADS) void k_pool_bwd_adj(int n, int L, const int* __restrict__ meta,
                                                          const float* __restrict__ gg,
                                                          const int* __restrict__ node_level,
                                                          const float* __restrict__ w, const int* __restrict__ argmax,
                                                          float* __restrict__ partial /*[grid][256]*/) {
    __shared__ float red[HW_PER_BLOCK][2 * H];
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const float inv_levels = 1.f / (float)(level_count(L, meta));
    const int4* argmax4 = reinterpret_cast<const int4*>(argmax);
    float4 sm = make_float4(0.f, 0.f, 0.f, 0.f), sx = sm;
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const Rows r = block_rows(n);
    for (int i = r.begin + hw; i < r.end; i += HW_PER_BLOCK) {
        const float4 v = reinterpret_cast<const float4*>(gg + (size_t)i * H)[l];
        const float wi = w[i];
        const float4 pick = readout_pull(i, node_level[i], 0.f, inv_levels, zero, one, argmax4, l);
        sm.x += wi * v.x; sm.y += wi * v.y; sm.z += wi * v.z; sm.w += wi * v.w;
        sx.x += pick.x * v.x; sx.y += pick.y * v.y; sx.z += pick.z * v.z; sx.w += pick.w * v.w;
    }
    reinterpret_cast<float4*>(&red[hw][0])[l] = sm;
    reinterpret_cast<float4*>(&red[hw][H])[l] = sx;
    block_colsum<2 * H>(red, partial);
}

// out[c] = sum_b partial[b][c]: one block per column, strided partial sums then a fixed-shape tree.  Columns from
// `split` on go to out2[c - split] (two results of one pass over the rows, summed by one launch).
__global__ __launch_bounds__(256) void k_sum_cols(int nb, int width, const float* __restrict__ partial,
                                                  float* __restrict__ out, int split = INT_MAX,
                                                  float* __restrict__ out2 = nullptr) {
    __shared__ float red[256];
    const int c = blockIdx.x, t = threadIdx.x;
    float s = 0.f;
    for (int b = t; b < nb; b += 256) s += partial[(size_t)b * width + c];
    red[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) {
        if (c < split) out[c] = red[0];
        else out2[c - split] = red[0];
    }
}

// x^ = alpha xr + (1 - alpha) xf,  h1 = elu(x^ W1^T + b1)
__global__ __launch_bounds__(THREADS) void k_gp_enc(int n, const int* __restrict__ row_limit, const float* __restrict__ xr, const float* __restrict__ xf,
                                                    const float* __restrict__ alpha, const float* __restrict__ W1,
                                                    const float* __restrict__ b1, float* __restrict__ xhat,
                                                    float* __restrict__ h1) {
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const float a = alpha[0];
    float w0[4], w1[4], bb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        w0[u] = W1[2 * (4 * l + u)];
        w1[u] = W1[2 * (4 * l + u) + 1];
        bb[u] = b1[4 * l + u];
    }
    const Rows r = block_rows(n, row_limit);
    for (int i = r.begin + hw; i < r.end; i += HW_PER_BLOCK) {
        const float2 pr = reinterpret_cast<const float2*>(xr)[i], pf = reinterpret_cast<const float2*>(xf)[i];
        const float x0 = a * pr.x + (1.f - a) * pf.x, x1 = a * pr.y + (1.f - a) * pf.y;
        if (l == 0) reinterpret_cast<float2*>(xhat)[i] = make_float2(x0, x1);
        float4 o;
        o.x = elu(x0 * w0[0] + x1 * w1[0] + bb[0]);
        o.y = elu(x0 * w0[1] + x1 * w1[1] + bb[1]);
        o.z = elu(x0 * w0[2] + x1 * w1[2] + bb[2]);
        o.w = elu(x0 * w0[3] + x1 * w1[3] + bb[3]);
        reinterpret_cast<float4*>(h1 + (size_t)i * H)[l] = o;
    }
}

// g_z2 = (P^T g_r) * elu'(z2)
__global__ __launch_bounds__(THREADS) void k_gp_gz2(int n, const int* __restrict__ row_limit, int L, const int* __restrict__ meta, const float* __restrict__ g_r,
                                                    const int* __restrict__ node_level, const float* __restrict__ w,
                                                    const int* __restrict__ argmax, const float* __restrict__ h2,
                                                    float* __restrict__ g_z2) {
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const float4 g_mean = reinterpret_cast<const float4*>(g_r)[l], g_max = reinterpret_cast<const float4*>(g_r + H)[l];
    const float inv_levels = 1.f / (float)(level_count(L, meta));
    const int4* argmax4 = reinterpret_cast<const int4*>(argmax);
    const Rows r = block_rows(n, row_limit);
    for (int i = r.begin + hw; i < r.end; i += HW_PER_BLOCK) {
        const float4 hv = reinterpret_cast<const float4*>(h2 + (size_t)i * H)[l];
        float4 gh = readout_pull(i, node_level[i], w[i], inv_levels, g_mean, g_max, argmax4, l);
        gh.x *= d_elu(hv.x); gh.y *= d_elu(hv.y); gh.z *= d_elu(hv.z); gh.w *= d_elu(hv.w);
        reinterpret_cast<float4*>(g_z2 + (size_t)i * H)[l] = gh;
    }
}

// g_z1 = g_h1 * elu'(z1) (in place over g_h1's copy `g_z1`),  g_x = g_z1 W1,  partial sums of |g_x|^2
__global__ __launch_bounds__(THREADS) void k_gp_gx(int n, const int* __restrict__ row_limit, const float* __restrict__ g_h1, const float* __restrict__ h1,
                                                   const float* __restrict__ W1, float* __restrict__ g_z1,
                                                   float* __restrict__ g_x, float* __restrict__ partial /*[grid]*/) {
    __shared__ float red[HW_PER_BLOCK];
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    float w0[4], w1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        w0[u] = W1[2 * (4 * l + u)];
        w1[u] = W1[2 * (4 * l + u) + 1];
    }
    float acc = 0.f;
    const Rows r = block_rows(n, row_limit);
    for (int i = r.begin + hw; i < r.end; i += HW_PER_BLOCK) {
        const float4 gv = reinterpret_cast<const float4*>(g_h1 + (size_t)i * H)[l];
        const float4 hv = reinterpret_cast<const float4*>(h1 + (size_t)i * H)[l];
        const float4 z = make_float4(gv.x * d_elu(hv.x), gv.y * d_elu(hv.y), gv.z * d_elu(hv.z), gv.w * d_elu(hv.w));
        reinterpret_cast<float4*>(g_z1 + (size_t)i * H)[l] = z;
        const float s0 = halfwave_sum(z.x * w0[0] + z.y * w0[1] + z.z * w0[2] + z.w * w0[3]);
        const float s1 = halfwave_sum(z.x * w1[0] + z.y * w1[1] + z.z * w1[2] + z.w * w1[3]);
        if (l == 0) reinterpret_cast<float2*>(g_x)[i] = make_float2(s0, s1);
        acc += s0 * s0 + s1 * s1;
    }
    if (l == 0) red[hw] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < HW_PER_BLOCK; ++k) s += red[k];
        partial[blockIdx.x] = s;
    }
}

// u = coef g_x;  ubar_gz1 = u W1^T;  ubar_gh1 = ubar_gz1 * elu'(z1);  dz1a = ubar_gz1 * g_h1 * elu''(z1)
__global__ __launch_bounds__(THREADS) void k_gp_back1(int n, const int* __restrict__ row_limit, const float* __restrict__ coef,
                                                      const float* __restrict__ g_x, const float* __restrict__ W1,
                                                      const float* __restrict__ h1, const float* __restrict__ g_h1,
                                                      float* __restrict__ u, float* __restrict__ ubar_gh1,
                                                      float* __restrict__ dz1a) {
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const float c = coef[0];
    float w0[4], w1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        w0[k] = W1[2 * (4 * l + k)];
        w1[k] = W1[2 * (4 * l + k) + 1];
    }
    const Rows r = block_rows(n, row_limit);
    for (int i = r.begin + hw; i < r.end; i += HW_PER_BLOCK) {
        const float2 gx = reinterpret_cast<const float2*>(g_x)[i];
        const float u0 = c * gx.x, u1 = c * gx.y;
        if (l == 0) reinterpret_cast<float2*>(u)[i] = make_float2(u0, u1);
        const float4 hv = reinterpret_cast<const float4*>(h1 + (size_t)i * H)[l];
        const float4 gv = reinterpret_cast<const float4*>(g_h1 + (size_t)i * H)[l];
        const float4 z = make_float4(u0 * w0[0] + u1 * w1[0], u0 * w0[1] + u1 * w1[1], u0 * w0[2] + u1 * w1[2],
                                     u0 * w0[3] + u1 * w1[3]);
        reinterpret_cast<float4*>(ubar_gh1 + (size_t)i * H)[l] =
            make_float4(z.x * d_elu(hv.x), z.y * d_elu(hv.y), z.z * d_elu(hv.z), z.w * d_elu(hv.w));
        reinterpret_cast<float4*>(dz1a + (size_t)i * H)[l] =
            make_float4(z.x * gv.x * dd_elu(hv.x), z.y * gv.y * dd_elu(hv.y), z.z * gv.z * dd_elu(hv.z),
                        z.w * gv.w * dd_elu(hv.w));
    }
}

// ubar_gh2 = ubar_gz2 * elu'(z2);  dz2 = ubar_gz2 * g_h2 * elu''(z2)  (g_h2 = P^T g_r, recomputed);
// block partials of [ colsum(dz2) | sum_i w_i ubar_gh2 | sum_i [argmax == i] ubar_gh2 / (L+1) ]
__global__ __launch_bounds__(THREADS) void k_gp_back2(int n, const int* __restrict__ row_limit, int L, const int* __restrict__ meta,
                                                      const float* __restrict__ ubar_gz2,
                                                      const float* __restrict__ h2, const float* __restrict__ g_r,
                                                      const int* __restrict__ node_level, const float* __restrict__ w,
                                                      const int* __restrict__ argmax, float* __restrict__ dz2,
                                                      float* __restrict__ partial /*[grid][384]*/) {
    __shared__ float red[HW_PER_BLOCK][3 * H];
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const float4 g_mean = reinterpret_cast<const float4*>(g_r)[l], g_max = reinterpret_cast<const float4*>(g_r + H)[l];
    const float inv_levels = 1.f / (float)(level_count(L, meta));
    const int4* argmax4 = reinterpret_cast<const int4*>(argmax);
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sb = zero, sm = zero, sx = zero;
    const Rows r = block_rows(n, row_limit);
    for (int i = r.begin + hw; i < r.end; i += HW_PER_BLOCK) {
        const float4 zv = reinterpret_cast<const float4*>(ubar_gz2 + (size_t)i * H)[l];
        const float4 hv = reinterpret_cast<const float4*>(h2 + (size_t)i * H)[l];
        const int lvl = node_level[i];
        const float wi = w[i];
        const float4 pick = readout_pull(i, lvl, 0.f, inv_levels, zero, one, argmax4, l);   // [argmax == i] / (L+1)
        const float4 gh = make_float4(wi * g_mean.x + pick.x * g_max.x, wi * g_mean.y + pick.y * g_max.y,
                                      wi * g_mean.z + pick.z * g_max.z, wi * g_mean.w + pick.w * g_max.w);
        const float4 ug = make_float4(zv.x * d_elu(hv.x), zv.y * d_elu(hv.y), zv.z * d_elu(hv.z), zv.w * d_elu(hv.w));
        const float4 dz = make_float4(zv.x * gh.x * dd_elu(hv.x), zv.y * gh.y * dd_elu(hv.y), zv.z * gh.z * dd_elu(hv.z),
                                      zv.w * gh.w * dd_elu(hv.w));
        reinterpret_cast<float4*>(dz2 + (size_t)i * H)[l] = dz;
        sb.x += dz.x; sb.y += dz.y; sb.z += dz.z; sb.w += dz.w;
        sm.x += wi * ug.x; sm.y += wi * ug.y; sm.z += wi * ug.z; sm.w += wi * ug.w;
        sx.x += pick.x * ug.x; sx.y += pick.y * ug.y; sx.z += pick.z * ug.z; sx.w += pick.w * ug.w;
    }
    reinterpret_cast<float4*>(&red[hw][0])[l] = sb;
    reinterpret_cast<float4*>(&red[hw][H])[l] = sm;
    reinterpret_cast<float4*>(&red[hw][2 * H])[l] = sx;
    block_colsum<3 * H>(red, partial);
}

// dz1 = dz1a + dh1 * elu'(z1) (in place over dz1a); block partials of colsum(dz1)
__global__ __launch_bounds__(THREADS) void k_gp_back3(int n, const int* __restrict__ row_limit, const float* __restrict__ dh1, const float* __restrict__ h1,
                                                      float* __restrict__ dz1, float* __restrict__ partial /*[grid][128]*/) {
    __shared__ float red[HW_PER_BLOCK][H];
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    float4 sb = make_float4(0.f, 0.f, 0.f, 0.f);
    const Rows r = block_rows(n, row_limit);
    for (int i = r.begin + hw; i < r.end; i += HW_PER_BLOCK) {
        const float4 dv = reinterpret_cast<const float4*>(dh1 + (size_t)i * H)[l];
        const float4 hv = reinterpret_cast<const float4*>(h1 + (size_t)i * H)[l];
        float4 z = reinterpret_cast<const float4*>(dz1 + (size_t)i * H)[l];
        z.x += dv.x * d_elu(hv.x); z.y += dv.y * d_elu(hv.y); z.z += dv.z * d_elu(hv.z); z.w += dv.w * d_elu(hv.w);
        reinterpret_cast<float4*>(dz1 + (size_t)i * H)[l] = z;
        sb.x += z.x; sb.y += z.y; sb.z += z.z; sb.w += z.w;
    }
    reinterpret_cast<float4*>(&red[hw][0])[l] = sb;
    block_colsum<H>(red, partial);
}

// The tile kernels (gp_tile.hip) win where the chain is bound by its launches: over the distinct rows of a penalty
// (a device-side row count, a few hundred rows as a rule) and over small inputs.  Over every node of a large graph the
// chain's six-product kernels are faster (4.47 against 4.91 ms per step at 100k nodes): it keeps those calls.
constexpr int64_t GP_TILES_MAX_PLAIN_ROWS = 16384;
bool use_tiles(int64_t n, const int32_t* n_rows_dev) {
    return gp_tiles_enabled() && (n_rows_dev != nullptr || n <= GP_TILES_MAX_PLAIN_ROWS);
}

int gp_blocks(int64_t n) {
    int64_t g = (n + 63) / 64;
    if (g > GP_BLOCKS_MAX) g = GP_BLOCKS_MAX;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace
}  // namespace aig

using namespace aig;

// --------------------------------------------------------------------------------------------- readout backward
extern "C" int aig_level_pool_bwd(int64_t n, int64_t L, const int32_t* lvl_meta, const float* g, const int32_t* node_level,
                                  const float* w, const int32_t* argmax, float* d_h, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX && L >= 0 && L < INT_MAX, "aig_level_pool_bwd: bad sizes");
    if (n == 0) return AIG_OK;
    AIG_REQUIRE(g && node_level && w && argmax && d_h, "aig_level_pool_bwd: null pointer");
    hipLaunchKernelGGL(k_pool_bwd, dim3(gp_blocks(n)), dim3(THREADS), 0, static_cast<hipStream_t>(stream), (int)n, (int)L,
                       lvl_meta, g, node_level, w, argmax, d_h);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

extern "C" size_t aig_level_pool_bwd_adj_ws_bytes(int64_t n) {
    (void)n;
    return align_up((size_t)GP_BLOCKS_MAX * 2 * H * 4) + 256;
}

extern "C" int aig_level_pool_bwd_adj(int64_t n, int64_t L, const int32_t* lvl_meta, const float* gg,
                                      const int32_t* node_level, const float* w, const int32_t* argmax, float* out,
                                      void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX && L >= 0 && L < INT_MAX, "aig_level_pool_bwd_adj: bad sizes");
    AIG_REQUIRE(out, "aig_level_pool_bwd_adj: null output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) {
        AIG_HIP_CHECK(fill_words(out, 0u, 2 * H * 4, s));
        return AIG_OK;
    }
    AIG_REQUIRE(gg && node_level && w && argmax && ws, "aig_level_pool_bwd_adj: null pointer");
    if (ws_bytes < aig_level_pool_bwd_adj_ws_bytes(n)) return fail(AIG_EWORKSPACE, "aig_level_pool_bwd_adj: workspace too small");
    float* partial = static_cast<float*>(ws);
    const int g = gp_blocks(n);
    hipLaunchKernelGGL(k_pool_bwd_adj, dim3(g), dim3(THREADS), 0, s, (int)n, (int)L, lvl_meta, gg, node_level, w, argmax,
                       partial);
    hipLaunchKernelGGL(k_sum_cols, dim3(2 * H), dim3(256), 0, s, g, 2 * H, partial, out);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

// --------------------------------------------------------------------------------------------- gradient penalty
extern "C" int aig_gp_forward_clearing(int64_t n, const int32_t* n_rows_dev, const float* x_real, const float* x_fake,
                                       const float* alpha, const float* W1, const float* b1, const float* W2, const float* b2,
                                       float* xhat, float* h1, float* h2, void* clear, size_t clear_bytes, void* stream) {
    AIG_REQUIRE(n > 0 && n < INT_MAX, "aig_gp_forward: n=%lld out of range", (long long)n);
    AIG_REQUIRE(x_real && x_fake && alpha && W1 && b1 && W2 && b2 && xhat && h1 && h2, "aig_gp_forward: null pointer");
    AIG_REQUIRE(clear_bytes == 0 || (clear && (reinterpret_cast<uintptr_t>(clear) & 7) == 0 && clear_bytes % 8 == 0 &&
                                     clear_bytes / 8 < (size_t)INT_MAX),
                "aig_gp_forward_clearing: the buffer to clear must be 8-byte aligned and a multiple of 8 bytes long");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (use_tiles(n, n_rows_dev))
        return launch_gp_tiles_fwd(s, (int)n, n_rows_dev, x_real, x_fake, alpha, W1, b1, W2, b2, xhat, h1, h2, clear, clear_bytes);
    if (clear_bytes) AIG_HIP_CHECK(fill_words(clear, 0u, clear_bytes, s));
    hipLaunchKernelGGL(k_gp_enc, dim3(gp_blocks(n)), dim3(THREADS), 0, s, (int)n, n_rows_dev, x_real, x_fake, alpha, W1, b1, xhat, h1);
    AIG_LAUNCH_CHECK();
    RowLimitScope limit(n_rows_dev);
    return aig_linear128(n, h1, W2, 0, b2, 2, h2, stream);
}

extern "C" int aig_gp_forward(int64_t n, const int32_t* n_rows_dev, const float* x_real, const float* x_fake, const float* alpha, const float* W1,
                              const float* b1, const float* W2, const float* b2, float* xhat, float* h1, float* h2,
                              void* stream) {
    return aig_gp_forward_clearing(n, n_rows_dev, x_real, x_fake, alpha, W1, b1, W2, b2, xhat, h1, h2, nullptr, 0, stream);
}

__global__ void k_gp_scalars(const float* __restrict__ norm_sq, float scale, float* __restrict__ out3) {
    const float norm = sqrtf(norm_sq[0]);
    const float d = norm - 1.0f;
    out3[0] = norm;
    out3[1] = d * d;
    out3[2] = norm > 0.f ? scale * (2.0f * d / norm) : 0.f;
}

// norm_sq = sum_b partial[b] (k_sum_cols' order: strided partial sums, then a fixed-shape tree) and the three scalars
// of k_gp_scalars behind it, in one launch
__global__ __launch_bounds__(256) void k_gp_norm_scalars(int nb, const float* __restrict__ partial, float scale,
                                                         float* __restrict__ norm_sq, float* __restrict__ out3) {
    __shared__ float red[256];
    const int t = threadIdx.x;
    float s = 0.f;
    for (int b = t; b < nb; b += 256) s += partial[b];
    red[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) {
        const float nsq = red[0];
        norm_sq[0] = nsq;
        const float norm = sqrtf(nsq);
        const float d = norm - 1.0f;
        out3[0] = norm;
        out3[1] = d * d;
        out3[2] = norm > 0.f ? scale * (2.0f * d / norm) : 0.f;
    }
}

// out[i] *= scale over four arrays (the chain's parameter gradients, scaled after the fact)
__global__ void k_gp_scale4(float* __restrict__ a, int na, float* __restrict__ b, int nb, float* __restrict__ c, int nc,
                            float* __restrict__ d, int nd, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) a[i] *= scale;
    else if (i < na + nb) b[i - na] *= scale;
    else if (i < na + nb + nc) c[i - na - nb] *= scale;
    else if (i < na + nb + nc + nd) d[i - na - nb - nc] *= scale;
}

extern "C" int aig_gp_scalars(const float* norm_sq, float scale, float* out3, void* stream) {
    AIG_REQUIRE(norm_sq && out3, "aig_gp_scalars: null pointer");
    hipLaunchKernelGGL(k_gp_scalars, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), norm_sq, scale, out3);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

extern "C" size_t aig_gp_input_grad_ws_bytes(int64_t n) {
    (void)n;
    return align_up((size_t)GP_BLOCKS_MAX * 4) + 256;
}

namespace {
int gp_input_grad(int64_t n, const int32_t* n_rows_dev, int64_t L, const int32_t* lvl_meta, const float* g_r,
                  const int32_t* node_level, const float* w, const int32_t* argmax, const float* W1, const float* W2, const float* h1,
                  const float* h2, float* g_z2, float* g_h1, float* g_z1, float* g_x, float* norm_sq, const float* grad_scale,
                  float* out3, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n > 0 && n < INT_MAX && L >= 0 && L < INT_MAX, "aig_gp_input_grad: bad sizes");
    AIG_REQUIRE(g_r && node_level && w && argmax && W1 && W2 && h1 && h2 && g_z2 && g_h1 && g_z1 && g_x && norm_sq && ws,
                "aig_gp_input_grad: null pointer");
    if (ws_bytes < aig_gp_input_grad_ws_bytes(n)) return fail(AIG_EWORKSPACE, "aig_gp_input_grad: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* partial = static_cast<float*>(ws);
    int blocks = 0;
    if (use_tiles(n, n_rows_dev)) {
        int rc = launch_gp_tiles_input_grad(s, (int)n, n_rows_dev, (int)L, lvl_meta, g_r, node_level, w, argmax, W1, W2, h1, h2,
                                            g_z2, g_h1, g_z1, g_x, partial, &blocks);
        if (rc) return rc;
    } else {
        RowLimitScope limit(n_rows_dev);
        blocks = gp_blocks(n);
        hipLaunchKernelGGL(k_gp_gz2, dim3(blocks), dim3(THREADS), 0, s, (int)n, n_rows_dev, (int)L, lvl_meta, g_r, node_level, w,
                           argmax, h2, g_z2);
        AIG_LAUNCH_CHECK();
        int rc = aig_linear128(n, g_z2, W2, 1, nullptr, 0, g_h1, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(k_gp_gx, dim3(blocks), dim3(THREADS), 0, s, (int)n, n_rows_dev, g_h1, h1, W1, g_z1, g_x, partial);
    }
    if (out3)
        hipLaunchKernelGGL(k_gp_norm_scalars, dim3(1), dim3(256), 0, s, blocks, partial, *grad_scale, norm_sq, out3);
    else
        hipLaunchKernelGGL(k_sum_cols, dim3(1), dim3(256), 0, s, blocks, 1, partial, norm_sq);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}
}  // namespace

extern "C" int aig_gp_input_grad(int64_t n, const int32_t* n_rows_dev, int64_t L, const int32_t* lvl_meta, const float* g_r,
                                 const int32_t* node_level, const float* w, const int32_t* argmax, const float* W1, const float* W2, const float* h1,
                                 const float* h2, float* g_z2, float* g_h1, float* g_z1, float* g_x, float* norm_sq,
                                 void* ws, size_t ws_bytes, void* stream) {
    return gp_input_grad(n, n_rows_dev, L, lvl_meta, g_r, node_level, w, argmax, W1, W2, h1, h2, g_z2, g_h1, g_z1, g_x, norm_sq,
                         nullptr, nullptr, ws, ws_bytes, stream);
}

extern "C" int aig_gp_input_grad_scalars(int64_t n, const int32_t* n_rows_dev, int64_t L, const int32_t* lvl_meta, const float* g_r,
                                         const int32_t* node_level, const float* w, const int32_t* argmax, const float* W1,
                                         const float* W2, const float* h1, const float* h2, float* g_z2, float* g_h1, float* g_z1,
                                         float* g_x, float* norm_sq, float grad_scale, float* out3, void* ws, size_t ws_bytes,
                                         void* stream) {
    AIG_REQUIRE(out3, "aig_gp_input_grad_scalars: null output");
    return gp_input_grad(n, n_rows_dev, L, lvl_meta, g_r, node_level, w, argmax, W1, W2, h1, h2, g_z2, g_h1, g_z1, g_x, norm_sq,
                         &grad_scale, out3, ws, ws_bytes, stream);
}

extern "C" size_t aig_gp_param_grad_ws_bytes(int64_t n) {
    const size_t N = (size_t)(n > 0 ? n : 1);
    size_t wg = 2 * aig_wgrad_ws_bytes(n, H), wg2 = 2 * aig_wgrad_ws_bytes(n, 2);      // (two products per weight)
    const size_t chain = 5 * align_up(N * H * 4) + align_up(N * 2 * 4) + align_up((size_t)GP_BLOCKS_MAX * 3 * H * 4) +
                         align_up(wg > wg2 ? wg : wg2) + 1024;
    const size_t tiles = gp_tiles_partial_bytes() + 256;
    if (!gp_tiles_enabled()) return chain;
    return n <= GP_TILES_MAX_PLAIN_ROWS ? tiles : (chain > tiles ? chain : tiles);      // (large n: either may run)
}

extern "C" int aig_gp_param_grad_scaled(int64_t n, const int32_t* n_rows_dev, int64_t L, const int32_t* lvl_meta, const float* coef,
                                        const float* g_r, const int32_t* node_level, const float* w, const int32_t* argmax,
                                        const float* W1, const float* W2, const float* xhat, const float* h1, const float* h2,
                                        const float* g_z2, const float* g_h1, const float* g_z1, const float* g_x, float* dW1,
                                        float* db1, float* dW2, float* db2, float* ubar_gr, float out_scale, void* ws,
                                        size_t ws_bytes, void* stream);

extern "C" int aig_gp_param_grad(int64_t n, const int32_t* n_rows_dev, int64_t L, const int32_t* lvl_meta, const float* coef,
                                 const float* g_r,
                                 const int32_t* node_level, const float* w, const int32_t* argmax, const float* W1, const float* W2,
                                 const float* xhat, const float* h1, const float* h2, const float* g_z2,
                                 const float* g_h1, const float* g_z1, const float* g_x, float* dW1, float* db1,
                                 float* dW2, float* db2, float* ubar_gr, void* ws, size_t ws_bytes, void* stream) {
    return aig_gp_param_grad_scaled(n, n_rows_dev, L, lvl_meta, coef, g_r, node_level, w, argmax, W1, W2, xhat, h1, h2, g_z2, g_h1,
                                    g_z1, g_x, dW1, db1, dW2, db2, ubar_gr, 1.0f, ws, ws_bytes, stream);
}

extern "C" int aig_gp_param_grad_scaled(int64_t n, const int32_t* n_rows_dev, int64_t L, const int32_t* lvl_meta, const float* coef,
                                        const float* g_r, const int32_t* node_level, const float* w, const int32_t* argmax,
                                        const float* W1, const float* W2, const float* xhat, const float* h1, const float* h2,
                                        const float* g_z2, const float* g_h1, const float* g_z1, const float* g_x, float* dW1,
                                        float* db1, float* dW2, float* db2, float* ubar_gr, float out_scale, void* ws,
                                        size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n > 0 && n < INT_MAX && L >= 0 && L < INT_MAX, "aig_gp_param_grad: bad sizes");
    AIG_REQUIRE(coef && g_r && node_level && w && argmax && W1 && W2 && xhat && h1 && h2 && g_z2 && g_h1 && g_z1 && g_x &&
                    dW1 && db1 && dW2 && db2 && ubar_gr && ws,
                "aig_gp_param_grad: null pointer");
    if (ws_bytes < aig_gp_param_grad_ws_bytes(n)) return fail(AIG_EWORKSPACE, "aig_gp_param_grad: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (use_tiles(n, n_rows_dev))
        return launch_gp_tiles_param_grad(s, (int)n, n_rows_dev, (int)L, lvl_meta, coef, g_r, node_level, w, argmax, W1, W2, xhat,
                                          h1, h2, g_z2, g_h1, g_z1, g_x, dW1, db1, dW2, db2, ubar_gr, static_cast<float*>(ws), out_scale);
    const size_t NH = (size_t)n * H;
    Carver c(ws);
    float* ubar_gh1 = c.take<float>(NH);
    float* dz1 = c.take<float>(NH);      // dz1a first, completed by k_gp_back3
    float* ubar_gz2 = c.take<float>(NH);
    float* dz2 = c.take<float>(NH);
    float* dh1 = c.take<float>(NH);
    float* u = c.take<float>((size_t)n * 2);
    float* partial = c.take<float>((size_t)GP_BLOCKS_MAX * 3 * H);
    const size_t wg_bytes = ws_bytes - c.off;
    void* wg = c.base + c.off;
    const int g = gp_blocks(n);
    int rc;
    RowLimitScope limit(n_rows_dev);

    hipLaunchKernelGGL(k_gp_back1, dim3(g), dim3(THREADS), 0, s, (int)n, n_rows_dev, coef, g_x, W1, h1, g_h1, u, ubar_gh1,
                       dz1);
    AIG_LAUNCH_CHECK();
    if ((rc = aig_linear128(n, ubar_gh1, W2, 0, nullptr, 0, ubar_gz2, stream))) return rc;
    hipLaunchKernelGGL(k_gp_back2, dim3(g), dim3(THREADS), 0, s, (int)n, n_rows_dev, (int)L, lvl_meta, ubar_gz2, h2, g_r,
                       node_level, w, argmax, dz2, partial);
    // columns 0..127 -> db2, 128..383 -> ubar_gr
    hipLaunchKernelGGL(k_sum_cols, dim3(3 * H), dim3(256), 0, s, g, 3 * H, partial, db2, H, ubar_gr);
    AIG_LAUNCH_CHECK();
    // dW2 = g_z2^T ubar_gh1 + dz2^T h1 (the partial sums of both products added up by one launch)
    if ((rc = wgrad_sum2(n, H, g_z2, ubar_gh1, dz2, h1, dW2, wg, wg_bytes, s))) return rc;
    // dz1 = dz1a + (dz2 W2) * elu'(z1)
    if ((rc = aig_linear128(n, dz2, W2, 1, nullptr, 0, dh1, stream))) return rc;
    hipLaunchKernelGGL(k_gp_back3, dim3(g), dim3(THREADS), 0, s, (int)n, n_rows_dev, dh1, h1, dz1, partial);
    hipLaunchKernelGGL(k_sum_cols, dim3(H), dim3(256), 0, s, g, H, partial, db1);
    AIG_LAUNCH_CHECK();
    // dW1 = g_z1^T u + dz1^T x^
    if ((rc = wgrad_sum2(n, 2, g_z1, u, dz1, xhat, dW1, wg, wg_bytes, s))) return rc;
    if (out_scale != 1.0f) {
        hipLaunchKernelGGL(k_gp_scale4, dim3((H * H + 2 * H + 2 * H + 255) / 256), dim3(256), 0, s, dW2, H * H, dW1, 2 * H, db1, H,
                           db2, H, out_scale);
        AIG_LAUNCH_CHECK();
    }
    return AIG_OK;
}
