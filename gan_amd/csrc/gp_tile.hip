// The closed-form gradient penalty (gp.hip) in three launches over tiles of 32 rows.
//
// The penalty's chain is row-local between its global sums: h1 -> h2 (forward), readout pull -> g_z2 -> g_h1 ->
// g_z1 -> g_x (input gradient), and everything behind the norm (second-order pass, the two weight gradients' partial
// sums).  Over the few hundred distinct rows the penalty usually runs on, a kernel per elementwise pass and per
// product is a chain of ~25 launches that each wait for the one before; here a workgroup keeps a tile of 32 rows in
// LDS next to the second layer's 128x128 weight and walks the chain on it: products on the f32 matrix cores
// (32x32x2, one wave per 32 output columns), the elementwise steps in the accumulators' own layout.  A block's sums
// (weight gradients, bias gradients, the readout's adjoint) leave as one row of partial sums, added up over the
// blocks in a fixed order by one more launch.
#include <climits>
#include <cstdlib>

#include "common.h"
#include "gat_common.h"

namespace aig {
namespace {

constexpr int H = AIG_HIDDEN;
constexpr int TR = 32;                 // rows of a tile
constexpr int LD = H + 1;              // LDS row pitch: rows of a tile / of the weight fall on different banks
constexpr int THREADS = 256;           // four waves: wave v owns output columns 32v .. 32v+31
constexpr int GPT_BLOCKS_MAX = 256;
constexpr int SMALL = 6 * TR;          // per-row scalars: u[2], xhat[2], w, level
constexpr size_t LDS_FWD = (size_t)(H * LD + TR * LD + SMALL) * 4;
constexpr size_t LDS_IN = (size_t)(H * LD + 2 * TR * LD + SMALL + 2 * H) * 4;
constexpr size_t LDS_PAR = (size_t)(H * LD + 4 * TR * LD + SMALL) * 4;
// one block's partial sums: dW2 | db2 | sum_i w_i ubar_gh2 | sum_i [argmax == i] ubar_gh2 / levels | db1 | dW1
constexpr int P_DB2 = H * H, P_SM = P_DB2 + H, P_SX = P_SM + H, P_DB1 = P_SX + H, P_DW1 = P_DB1 + H, P_ROW = P_DW1 + 2 * H;

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float d_elu(float h) { return h > 0.f ? 1.f : h + 1.f; }
__device__ __forceinline__ float dd_elu(float h) { return h > 0.f ? 0.f : h + 1.f; }
__device__ __forceinline__ float elu(float z) { return z > 0.f ? z : expm1f(z); }

// accumulator register i of the half kh (lanes 32..63: kh = 1) -> row of the 32x32 block
__device__ __forceinline__ int acc_row(int i, int kh) { return (i & 3) + 8 * (i >> 2) + 4 * kh; }

__device__ __forceinline__ void stage_w2(float* __restrict__ W2s, const float* __restrict__ W2) {
    constexpr int N = H * H / 4 / THREADS;      // 16-byte loads per thread, all in flight before the first write
    float4 v[N];
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = reinterpret_cast<const float4*>(W2)[threadIdx.x + u * THREADS];
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int q = threadIdx.x + u * THREADS;
        float* d = W2s + (q >> 5) * LD + 4 * (q & 31);
        d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
    }
}

// acc[row][c] = sum_k A[row][k] * (WT ? W2[c][k] : W2[k][c]) for the wave's 32 columns (c = this lane's column): two
// chains of 32 products each
template <bool WT>
__device__ __forceinline__ f32x16 tile_times_w2(const float* __restrict__ A, const float* __restrict__ W2s, int ln, int kh,
                                                int c) {
    f32x16 acc0 = {0}, acc1 = {0};
    const float* ap = A + ln * LD + kh;
    const float* bp = WT ? W2s + c * LD + kh : W2s + kh * LD + c;
#pragma unroll 8
    for (int kk = 0; kk < H / 2; kk += 2) {
        const float a0 = ap[2 * kk], a1 = ap[2 * kk + 2];
        const float b0 = WT ? bp[2 * kk] : bp[2 * kk * LD];
        const float b1 = WT ? bp[2 * kk + 2] : bp[(2 * kk + 2) * LD];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}

// x^ = alpha xr + (1 - alpha) xf,  h1 = elu(x^ W1^T + b1),  h2 = elu(h1 W2^T + b2)
__global__ __launch_bounds__(THREADS) void k_gpt_fwd(int n, const int* __restrict__ row_limit, const float* __restrict__ xr,
                                                     const float* __restrict__ xf, const float* __restrict__ alpha,
                                                     const float* __restrict__ W1, const float* __restrict__ b1,
                                                     const float* __restrict__ W2, const float* __restrict__ b2,
                                                     float* __restrict__ xhat, float* __restrict__ h1,
                                                     float* __restrict__ h2, uint2* __restrict__ clear, int clear_count) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* W2s = lds;
    float* TA = W2s + H * LD;
    float* xs = TA + TR * LD;
    // (what the readout behind this kernel wants zeroed: by every block of the grid, with or without a tile)
    for (int q = blockIdx.x * THREADS + threadIdx.x; q < clear_count; q += gridDim.x * THREADS) clear[q] = make_uint2(0u, 0u);
    n = limited_rows(n, row_limit);
    const int ntiles = (n + TR - 1) / TR;
    if ((int)blockIdx.x >= ntiles) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, ln = lane & 31, kh = lane >> 5, c = 32 * wv + ln;
    stage_w2(W2s, W2);
    const float a = alpha[0];
    const float w0 = W1[2 * c], w1 = W1[2 * c + 1], bb1 = b1[c], bb2 = b2[c];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * TR;
        __syncthreads();      // the tile of the trip before is no longer read (first trip: nothing to wait for)
        if (t < TR) {
            const int row = row0 + t;
            float2 x = make_float2(0.f, 0.f);
            if (row < n) {
                const float2 pr = reinterpret_cast<const float2*>(xr)[row], pf = reinterpret_cast<const float2*>(xf)[row];
                x = make_float2(a * pr.x + (1.f - a) * pf.x, a * pr.y + (1.f - a) * pf.y);
                reinterpret_cast<float2*>(xhat)[row] = x;
            }
            xs[2 * t] = x.x;
            xs[2 * t + 1] = x.y;
        }
        __syncthreads();      // (also: the weight is staged)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = acc_row(i, kh), row = row0 + r;
            const float v = elu(xs[2 * r] * w0 + xs[2 * r + 1] * w1 + bb1);
            if (row < n) h1[(size_t)row * H + c] = v;
            TA[r * LD + c] = row < n ? v : 0.f;
        }
        __syncthreads();
        const f32x16 acc = tile_times_w2<true>(TA, W2s, ln, kh, c);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = row0 + acc_row(i, kh);
            if (row < n) h2[(size_t)row * H + c] = elu(acc[i] + bb2);
        }
    }
}

// g_z2 = (P^T g_r) * elu'(z2),  g_h1 = g_z2 W2,  g_z1 = g_h1 * elu'(z1),  g_x = g_z1 W1,  the block's share of |g_x|^2
__global__ __launch_bounds__(THREADS) void k_gpt_input_grad(int n, const int* __restrict__ row_limit, int L,
                                                            const int* __restrict__ meta, const float* __restrict__ g_r,
                                                            const int* __restrict__ node_level, const float* __restrict__ w,
                                                            const int* __restrict__ argmax, const float* __restrict__ W1,
                                                            const float* __restrict__ W2, const float* __restrict__ h1,
                                                            const float* __restrict__ h2, float* __restrict__ g_z2,
                                                            float* __restrict__ g_h1, float* __restrict__ g_z1,
                                                            float* __restrict__ g_x, float* __restrict__ partial /*[grid]*/) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* W2s = lds;
    float* TA = W2s + H * LD;
    float* TB = TA + TR * LD;
    float* wr = TB + TR * LD;                       // [TR] readout weight of the row
    int* lv = reinterpret_cast<int*>(wr + TR);      // [TR] level of the row
    float* red = wr + 2 * TR;                       // [TR] (inside the per-row scalars' space)
    float* W1s = wr + SMALL;                        // [2H]
    n = limited_rows(n, row_limit);
    const int ntiles = (n + TR - 1) / TR;
    const int t = threadIdx.x;
    if ((int)blockIdx.x >= ntiles) {
        if (t == 0) partial[blockIdx.x] = 0.f;
        return;
    }
    const int lane = t & 63, wv = t >> 6, ln = lane & 31, kh = lane >> 5, c = 32 * wv + ln;
    stage_w2(W2s, W2);
    W1s[t] = W1[t];
    const float g_mean = g_r[c], g_max = g_r[H + c];
    const float inv_levels = 1.f / (float)(level_count(L, meta));
    float norm_acc = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * TR;
        __syncthreads();
        if (t < TR) {
            const int row = row0 + t;
            wr[t] = row < n ? w[row] : 0.f;
            lv[t] = row < n ? node_level[row] : -1;
        }
        __syncthreads();
        // (loads of rows past the end go to the last row instead of standing under a condition: sixteen loads in
        //  flight, not sixteen round trips)
        float hv2[16], hv1[16];
        int am[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = acc_row(i, kh), rr = min(row0 + r, n - 1);
            hv2[i] = h2[(size_t)rr * H + c];
            am[i] = argmax[(size_t)max(lv[r], 0) * H + c];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) hv1[i] = h1[(size_t)min(row0 + acc_row(i, kh), n - 1) * H + c];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = acc_row(i, kh), row = row0 + r;
            const bool hit = lv[r] >= 0 && am[i] == row;
            const float pull = wr[r] * g_mean + (hit ? g_max * inv_levels : 0.f);
            const float gz = row < n ? pull * d_elu(hv2[i]) : 0.f;
            if (row < n) g_z2[(size_t)row * H + c] = gz;
            TA[r * LD + c] = gz;
        }
        __syncthreads();
        const f32x16 acc = tile_times_w2<false>(TA, W2s, ln, kh, c);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = acc_row(i, kh), row = row0 + r;
            const float z = row < n ? acc[i] * d_elu(hv1[i]) : 0.f;
            if (row < n) {
                g_h1[(size_t)row * H + c] = acc[i];
                g_z1[(size_t)row * H + c] = z;
            }
            TB[r * LD + c] = z;
        }
        __syncthreads();
        {   // g_x of row t / 8: eight lanes take sixteen columns each
            const int r = t >> 3, p = t & 7, row = row0 + r;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int col = p + 8 * k;
                const float z = TB[r * LD + col];
                s0 = fmaf(z, W1s[2 * col], s0);
                s1 = fmaf(z, W1s[2 * col + 1], s1);
            }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                s0 += __shfl_xor(s0, o, 8);
                s1 += __shfl_xor(s1, o, 8);
            }
            if (p == 0 && row < n) {
                reinterpret_cast<float2*>(g_x)[row] = make_float2(s0, s1);
                norm_acc += s0 * s0 + s1 * s1;
            }
        }
    }
    __syncthreads();
    if ((t & 7) == 0) red[t >> 3] = norm_acc;
    __syncthreads();
    if (t == 0) {
        float s = 0.f;
        for (int k = 0; k < TR; ++k) s += red[k];
        partial[blockIdx.x] = s;
    }
}

// Everything behind the norm, per tile:
//   u = coef g_x;  ubar_gz1 = u W1^T;  ubar_gh1 = ubar_gz1 * elu'(z1);  dz1a = ubar_gz1 * g_h1 * elu''(z1)
//   ubar_gz2 = ubar_gh1 W2^T;  ubar_gh2 = ubar_gz2 * elu'(z2);  dz2 = ubar_gz2 * (P^T g_r) * elu''(z2)
//   dz1 = dz1a + (dz2 W2) * elu'(z1)
// and the block's sums  dW2 += g_z2^T ubar_gh1 + dz2^T h1,  db2 += dz2,  the readout's adjoint of ubar_gh2,
// db1 += dz1,  dW1 += g_z1^T u + dz1^T x^.
__global__ __launch_bounds__(THREADS) void k_gpt_param_grad(
    int n, const int* __restrict__ row_limit, int L, const int* __restrict__ meta, const float* __restrict__ coef,
    const float* __restrict__ g_r, const int* __restrict__ node_level, const float* __restrict__ w,
    const int* __restrict__ argmax, const float* __restrict__ W1, const float* __restrict__ W2,
    const float* __restrict__ xhat, const float* __restrict__ h1, const float* __restrict__ h2,
    const float* __restrict__ g_z2, const float* __restrict__ g_h1, const float* __restrict__ g_z1,
    const float* __restrict__ g_x, float* __restrict__ partial /*[grid][P_ROW]*/) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* W2s = lds;
    float* T_ug1 = W2s + H * LD;
    float* T_h1 = T_ug1 + TR * LD;
    float* T_gz2 = T_h1 + TR * LD;
    float* T_dz2 = T_gz2 + TR * LD;
    float* us = T_dz2 + TR * LD;                    // [TR][2] u
    float* xh = us + 2 * TR;                        // [TR][2] x^
    float* wr = xh + 2 * TR;                        // [TR]
    int* lv = reinterpret_cast<int*>(wr + TR);      // [TR]
    n = limited_rows(n, row_limit);
    const int ntiles = (n + TR - 1) / TR;
    if ((int)blockIdx.x >= ntiles) return;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, ln = lane & 31, kh = lane >> 5, c = 32 * wv + ln;
    stage_w2(W2s, W2);
    const float cf = coef[0];
    const float w0 = W1[2 * c], w1 = W1[2 * c + 1];
    const float g_mean = g_r[c], g_max = g_r[H + c];
    const float inv_levels = 1.f / (float)(level_count(L, meta));
    f32x16 accW[4] = {{0}, {0}, {0}, {0}};
    float sb = 0.f, sm = 0.f, sx = 0.f, sb1 = 0.f, d10 = 0.f, d11 = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * TR;
        __syncthreads();
        if (t < TR) {
            const int row = row0 + t;
            float2 gx = make_float2(0.f, 0.f), xv = gx;
            if (row < n) {
                gx = reinterpret_cast<const float2*>(g_x)[row];
                xv = reinterpret_cast<const float2*>(xhat)[row];
            }
            us[2 * t] = cf * gx.x;
            us[2 * t + 1] = cf * gx.y;
            xh[2 * t] = xv.x;
            xh[2 * t + 1] = xv.y;
            wr[t] = row < n ? w[row] : 0.f;
            lv[t] = row < n ? node_level[row] : -1;
        }
        __syncthreads();
        // (loads of rows past the end go to the last row instead of standing under a condition: all of a phase's
        //  loads in flight at once)
        float dz1a[16], hv2[16], gz1[16];
        int am[16];
        {
            float hv[16], gz[16], gv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const size_t at = (size_t)min(row0 + acc_row(i, kh), n - 1) * H + c;
                hv[i] = h1[at];
                gz[i] = g_z2[at];
                gv[i] = g_h1[at];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = acc_row(i, kh);
                const size_t at = (size_t)min(row0 + r, n - 1) * H + c;
                hv2[i] = h2[at];
                gz1[i] = g_z1[at];
                am[i] = argmax[(size_t)max(lv[r], 0) * H + c];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = acc_row(i, kh);
                const bool valid = row0 + r < n;
                const float z = us[2 * r] * w0 + us[2 * r + 1] * w1;
                dz1a[i] = valid ? z * gv[i] * dd_elu(hv[i]) : 0.f;
                T_ug1[r * LD + c] = valid ? z * d_elu(hv[i]) : 0.f;
                T_h1[r * LD + c] = valid ? hv[i] : 0.f;
                T_gz2[r * LD + c] = valid ? gz[i] : 0.f;
            }
        }
        __syncthreads();
        const f32x16 zv2 = tile_times_w2<true>(T_ug1, W2s, ln, kh, c);       // ubar_gz2
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = acc_row(i, kh), row = row0 + r;
            const bool valid = row < n;
            const float wi = wr[r];
            const float pick = (valid && lv[r] >= 0 && am[i] == row) ? inv_levels : 0.f;
            const float gh = wi * g_mean + pick * g_max;
            const float ug2 = valid ? zv2[i] * d_elu(hv2[i]) : 0.f;
            const float dz = valid ? zv2[i] * gh * dd_elu(hv2[i]) : 0.f;
            sb += dz;
            sm += wi * ug2;
            sx += pick * ug2;
            T_dz2[r * LD + c] = dz;
        }
        __syncthreads();
        const f32x16 dh1 = tile_times_w2<false>(T_dz2, W2s, ln, kh, c);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = acc_row(i, kh);
            if (row0 + r < n) {
                const float dz1 = dz1a[i] + dh1[i] * d_elu(T_h1[r * LD + c]);
                sb1 += dz1;
                d10 += gz1[i] * us[2 * r] + dz1 * xh[2 * r];
                d11 += gz1[i] * us[2 * r + 1] + dz1 * xh[2 * r + 1];
            }
        }
        // dW2[m][c] += sum_rows g_z2[row][m] ubar_gh1[row][c] + dz2[row][m] h1[row][c]
#pragma unroll 4
        for (int kk = 0; kk < TR / 2; ++kk) {
            const int k = 2 * kk + kh;
            const float b1v = T_ug1[k * LD + c], b2v = T_h1[k * LD + c];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float a1 = T_gz2[k * LD + 32 * mb + ln], a2 = T_dz2[k * LD + 32 * mb + ln];
                accW[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1v, accW[mb], 0, 0, 0);
                accW[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2v, accW[mb], 0, 0, 0);
            }
        }
    }
    float* P = partial + (size_t)blockIdx.x * P_ROW;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int i = 0; i < 16; ++i) P[(size_t)(32 * mb + acc_row(i, kh)) * H + c] = accW[mb][i];
    sb += __shfl_xor(sb, 32);
    sm += __shfl_xor(sm, 32);
    sx += __shfl_xor(sx, 32);
    sb1 += __shfl_xor(sb1, 32);
    d10 += __shfl_xor(d10, 32);
    d11 += __shfl_xor(d11, 32);
    if (kh == 0) {
        P[P_DB2 + c] = sb;
        P[P_SM + c] = sm;
        P[P_SX + c] = sx;
        P[P_DB1 + c] = sb1;
        P[P_DW1 + 2 * c] = d10;
        P[P_DW1 + 2 * c + 1] = d11;
    }
}

// the partial rows of the blocks that had a tile, added up in ascending order (four chains, combined in a fixed way)
__global__ __launch_bounds__(256) void k_gpt_reduce(int n, const int* __restrict__ row_limit, int grid,
                                                    const float* __restrict__ partial, float* __restrict__ dW2,
                                                    float* __restrict__ db2, float* __restrict__ ubar_gr,
                                                    float* __restrict__ db1, float* __restrict__ dW1, float out_scale) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= P_ROW) return;
    n = limited_rows(n, row_limit);
    const int nb = min(grid, (n + TR - 1) / TR);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = 0;
    for (; b + 3 < nb; b += 4) {
        s0 += partial[(size_t)b * P_ROW + e];
        s1 += partial[(size_t)(b + 1) * P_ROW + e];
        s2 += partial[(size_t)(b + 2) * P_ROW + e];
        s3 += partial[(size_t)(b + 3) * P_ROW + e];
    }
    for (; b < nb; ++b) s0 += partial[(size_t)b * P_ROW + e];
    const float s = (s0 + s1) + (s2 + s3);
    // (the parameter gradients leave multiplied by out_scale - one rounding each, as a multiplication after the fact
    //  would give; the readout's adjoint is an intermediate of the head's share and stays as it is)
    if (e < P_DB2) dW2[e] = s * out_scale;
    else if (e < P_SM) db2[e - P_DB2] = s * out_scale;
    else if (e < P_DB1) ubar_gr[e - P_SM] = s;
    else if (e < P_DW1) db1[e - P_DB1] = s * out_scale;
    else dW1[e - P_DW1] = s * out_scale;
}

int tiles_grid(int64_t n) {
    int64_t g = (n + TR - 1) / TR;
    return (int)(g < 1 ? 1 : g > GPT_BLOCKS_MAX ? GPT_BLOCKS_MAX : g);
}

}  // namespace

bool gp_tiles_enabled() {
    static const bool on = [] { const char* e = getenv("AIG_GP_TILES"); return !(e && e[0] == '0'); }();
    return on;
}

size_t gp_tiles_partial_bytes() { return align_up((size_t)GPT_BLOCKS_MAX * P_ROW * 4); }

int launch_gp_tiles_fwd(hipStream_t s, int n, const int* row_limit, const float* xr, const float* xf, const float* alpha,
                        const float* W1, const float* b1, const float* W2, const float* b2, float* xhat, float* h1, float* h2,
                        void* clear, size_t clear_bytes) {
    AIG_REQUIRE(((reinterpret_cast<uintptr_t>(xr) | reinterpret_cast<uintptr_t>(xf) | reinterpret_cast<uintptr_t>(xhat)) & 7) == 0 &&
                    (reinterpret_cast<uintptr_t>(W2) & 15) == 0,
                "gradient penalty: feature rows are read as 8-byte pairs, the weight as 16-byte vectors");
    AIG_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_gpt_fwd), LDS_FWD));
    hipLaunchKernelGGL(k_gpt_fwd, dim3(tiles_grid(n)), dim3(THREADS), LDS_FWD, s, n, row_limit, xr, xf, alpha, W1, b1, W2, b2,
                       xhat, h1, h2, static_cast<uint2*>(clear), (int)(clear_bytes / 8));
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

int launch_gp_tiles_input_grad(hipStream_t s, int n, const int* row_limit, int L, const int* meta, const float* g_r,
                               const int* node_level, const float* w, const int* argmax, const float* W1, const float* W2,
                               const float* h1, const float* h2, float* g_z2, float* g_h1, float* g_z1, float* g_x,
                               float* partial, int* blocks) {
    AIG_REQUIRE((reinterpret_cast<uintptr_t>(g_x) & 7) == 0 && (reinterpret_cast<uintptr_t>(W2) & 15) == 0,
                "gradient penalty: g_x is written as 8-byte pairs, the weight read as 16-byte vectors");
    AIG_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_gpt_input_grad), LDS_IN));
    const int g = tiles_grid(n);
    hipLaunchKernelGGL(k_gpt_input_grad, dim3(g), dim3(THREADS), LDS_IN, s, n, row_limit, L, meta, g_r, node_level, w, argmax,
                       W1, W2, h1, h2, g_z2, g_h1, g_z1, g_x, partial);
    AIG_LAUNCH_CHECK();
    *blocks = g;
    return AIG_OK;
}

int launch_gp_tiles_param_grad(hipStream_t s, int n, const int* row_limit, int L, const int* meta, const float* coef,
                               const float* g_r, const int* node_level, const float* w, const int* argmax, const float* W1,
                               const float* W2, const float* xhat, const float* h1, const float* h2, const float* g_z2,
                               const float* g_h1, const float* g_z1, const float* g_x, float* dW1, float* db1, float* dW2,
                               float* db2, float* ubar_gr, float* partial, float out_scale) {
    AIG_REQUIRE(((reinterpret_cast<uintptr_t>(g_x) | reinterpret_cast<uintptr_t>(xhat)) & 7) == 0 &&
                    (reinterpret_cast<uintptr_t>(W2) & 15) == 0,
                "gradient penalty: g_x and x^ are read as 8-byte pairs, the weight as 16-byte vectors");
    AIG_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_gpt_param_grad), LDS_PAR));
    const int g = tiles_grid(n);
    hipLaunchKernelGGL(k_gpt_param_grad, dim3(g), dim3(THREADS), LDS_PAR, s, n, row_limit, L, meta, coef, g_r, node_level, w,
                       argmax, W1, W2, xhat, h1, h2, g_z2, g_h1, g_z1, g_x, partial);
    hipLaunchKernelGGL(k_gpt_reduce, dim3((P_ROW + 255) / 256), dim3(256), 0, s, n, row_limit, g, partial, dW2, db2, ubar_gr,
                       db1, dW1, out_scale);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

}  // namespace aig
