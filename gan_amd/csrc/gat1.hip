// First attention layer of the critic, GATConv(2, 32, heads=4, edge_dim=1) + ELU (models/discriminator.py:30,60),
// without its projected rows.  The layer's input has TWO features per node (node type, inverter count), so
//   xw[j]     = W x[j]                        two multiply-adds per channel away from 8 bytes,
//   a_s[j,h]  = <xw[j,h,:], att_src[h,:]>    = x[j,0] cs[h][0] + x[j,1] cs[h][1]      (cs, cd: 8 numbers each)
//   out[i]    = elu(sum_e alpha_e xw[src_e] + b) = elu(W xa[i] + b),   xa[i,h,:] = sum_e alpha_{e,h} x[src_e]
// and nothing [n,128] has to be gathered: the forward kernel reads 8 bytes per list entry instead of a 512-byte
// row, the backward pass is ONE pass by target with no gathered rows and no pass by source at all:
//   dalpha_{e,h} = <g[i,h,:], xw[src_e,h,:]> = x[src_e,0] gw0[i,h] + x[src_e,1] gw1[i,h],  gw_f[i,h] = <g[i,h,:], W[h,:,f]>
//   dW[c,f]      = sum_i g[i,c] xa[i,h(c),f] + att_src[c] S_s[h,f] + att_dst[c] S_d[h,f]
//   S_s[h,f]     = sum_entries d_raw_{e,h} x[src_e,f],   S_d[h,f] = sum_entries d_raw_{e,h} x[dst_e,f]
//   d att_src[c] = W[c,0] S_s[h,0] + W[c,1] S_s[h,1],   d att_dst likewise from S_d
// (the sums over a node's out-edges that the generic layer makes by source are sums over list entries here, and
// every list entry is seen by its target).  Same half-wave-per-node shape as gat.hip, no floating-point atomics,
// block partials added in fixed order.
#include <climits>

#include "common.h"
#include "gat_common.h"

namespace aig {
namespace {

constexpr int H = GAT_H, HEADS = GAT_HEADS, THREADS = GAT_THREADS, HW_PER_BLOCK = GAT_HW_PER_BLOCK;
// coef[0..7] = cs[f][h] (f major), coef[8..15] = cd[f][h], coef[16..19] = ce[h]
constexpr int COEF = 20;
// columns of a block's partial row in the backward pass
constexpr int P_BIAS = 0, P_DW0 = H, P_DW1 = 2 * H, P_CE = 3 * H, P_SS = 3 * H + 4, P_SD = 3 * H + 12, P_WIDTH = 3 * H + 20;

__global__ void k_gat1_coef(const float* __restrict__ W, const float* __restrict__ att_src,
                            const float* __restrict__ att_dst, const float* __restrict__ lin_edge_w,
                            const float* __restrict__ att_edge, float* __restrict__ coef) {
    const int t = threadIdx.x;
    if (t < 16) {
        const int which = t >> 3, f = (t >> 2) & 1, h = t & 3;
        const float* a = which ? att_dst : att_src;
        float s = 0.f;
        for (int c = 0; c < H / HEADS; ++c) s += W[(h * (H / HEADS) + c) * 2 + f] * a[h * (H / HEADS) + c];
        coef[t] = s;
    } else if (t < COEF) {
        const int h = t - 16;
        float s = 0.f;
        for (int c = 0; c < H / HEADS; ++c) s += lin_edge_w[h * (H / HEADS) + c] * att_edge[h * (H / HEADS) + c];
        coef[t] = s;
    }
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 mul4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 mul44(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ void acc4(float4& a, float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
// x0 * c0 + x1 * c1 per head
__device__ __forceinline__ float4 lin2(float2 x, float4 c0, float4 c1) {
    return make_float4(x.x * c0.x + x.y * c1.x, x.x * c0.y + x.y * c1.y, x.x * c0.z + x.y * c1.z, x.x * c0.w + x.y * c1.w);
}
__device__ __forceinline__ float pick4(float4 v, int h) { return h == 0 ? v.x : h == 1 ? v.y : h == 2 ? v.z : v.w; }

struct Lists {
    const int* __restrict__ row_ptr;
    const int* __restrict__ src;
    const float* __restrict__ ea;
    const float* __restrict__ loop_ea;
};

// The two weight columns of the four channels lane l owns: W is [128][2] row-major, channels 4l .. 4l+3.
struct LaneW {
    float4 w0, w1;
};
__device__ __forceinline__ LaneW lane_weights(const float* __restrict__ W, int l) {
    const float4 a = ld4(W + 8 * l), b = ld4(W + 8 * l + 4);
    return {make_float4(a.x, a.z, b.x, b.z), make_float4(a.y, a.w, b.y, b.w)};
}

__global__ __launch_bounds__(THREADS) void k_gat1_fwd(int n, const float2* __restrict__ x, const float* __restrict__ W,
                                                      const float* __restrict__ coef, const float* __restrict__ bias,
                                                      Lists g, float* __restrict__ out, float* __restrict__ alpha_e,
                                                      float* __restrict__ alpha_loop) {
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5, h = l >> 3;
    const float4 cs0 = ld4(coef), cs1 = ld4(coef + 4), cd0 = ld4(coef + 8), cd1 = ld4(coef + 12), ce4 = ld4(coef + 16);
    const LaneW w = lane_weights(W, l);
    const float4 b4 = ld4(bias + 4 * l);
    float4* alpha_e4 = reinterpret_cast<float4*>(alpha_e);
    float4* alpha_loop4 = reinterpret_cast<float4*>(alpha_loop);
    const bool has_edges = g.row_ptr[n] > 0;   // clamped list reads need one valid element
    const int* src_safe = has_edges ? g.src : g.row_ptr;
    const float* ea_safe = has_edges ? g.ea : g.loop_ea;
    const float NEG_INF = -__builtin_huge_valf();
    NodeRange r = block_nodes(n);
    for (int i = r.begin + hw; i < r.end; i += HW_PER_BLOCK) {
        const int beg = g.row_ptr[i], deg = g.row_ptr[i + 1] - beg;
        const float2 xi = x[i];
        const float lea = g.loop_ea[i];
        const float4 ad4 = lin2(xi, cd0, cd1);
        // entry idx of the list: edges 0..deg-1, then the self loop
        auto entry = [&](int idx, float2& xs, float4& e) {
            const bool is_edge = idx < deg;
            const int p = is_edge ? beg + idx : 0;
            const int sv = src_safe[p];
            const float eav = ea_safe[p];
            const int s = is_edge ? sv : i;
            const float ea = is_edge ? eav : lea;
            xs = x[s];
            const float4 as4 = lin2(xs, cs0, cs1);
            e.x = lrelu(as4.x + ad4.x + ea * ce4.x); e.y = lrelu(as4.y + ad4.y + ea * ce4.y);
            e.z = lrelu(as4.z + ad4.z + ea * ce4.z); e.w = lrelu(as4.w + ad4.w + ea * ce4.w);
            if (idx > deg) e = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
        };
        auto put_alpha = [&](int idx, float4 al) {
            if (idx < deg) alpha_e4[beg + idx] = al;
            if (idx == deg) alpha_loop4[i] = al;
        };
        float4 xa0, xa1;   // xa[h][f]: sum over the list of alpha_h * x_f, all four heads in every lane
        if (deg < 32) {
            float2 xs;
            float4 e;
            entry(l, xs, e);
            const bool wide = deg > 3;
            const float4 m = halfwave_reduce(e, wide, MaxOp());
            float4 a = make_float4(fast_exp(e.x - m.x), fast_exp(e.y - m.y), fast_exp(e.z - m.z), fast_exp(e.w - m.w));
            const float4 den = halfwave_reduce(a, wide, SumOp());
            a.x *= 1.f / (den.x + SOFTMAX_EPS); a.y *= 1.f / (den.y + SOFTMAX_EPS);
            a.z *= 1.f / (den.z + SOFTMAX_EPS); a.w *= 1.f / (den.w + SOFTMAX_EPS);
            put_alpha(l, a);
            xa0 = halfwave_reduce(mul4(a, xs.x), wide, SumOp());
            xa1 = halfwave_reduce(mul4(a, xs.y), wide, SumOp());
        } else {
            // long list: chunks of 32 entries, three sweeps (max, denominator, alpha and its sums)
            const int cnt = deg + 1;
            float4 m = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
            for (int c0 = 0; c0 < cnt; c0 += 32) {
                float2 xs;
                float4 e;
                entry(c0 + l, xs, e);
                m.x = fmaxf(m.x, e.x); m.y = fmaxf(m.y, e.y); m.z = fmaxf(m.z, e.z); m.w = fmaxf(m.w, e.w);
            }
            m = halfwave_reduce(m, true, MaxOp());
            float4 den = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int c0 = 0; c0 < cnt; c0 += 32) {
                float2 xs;
                float4 e;
                entry(c0 + l, xs, e);
                den.x += fast_exp(e.x - m.x); den.y += fast_exp(e.y - m.y);
                den.z += fast_exp(e.z - m.z); den.w += fast_exp(e.w - m.w);
            }
            den = halfwave_reduce(den, true, SumOp());
            const float4 inv = make_float4(1.f / (den.x + SOFTMAX_EPS), 1.f / (den.y + SOFTMAX_EPS),
                                           1.f / (den.z + SOFTMAX_EPS), 1.f / (den.w + SOFTMAX_EPS));
            xa0 = make_float4(0.f, 0.f, 0.f, 0.f);
            xa1 = xa0;
            for (int c0 = 0; c0 < cnt; c0 += 32) {
                float2 xs;
                float4 e;
                entry(c0 + l, xs, e);
                const float4 a = make_float4(fast_exp(e.x - m.x) * inv.x, fast_exp(e.y - m.y) * inv.y,
                                             fast_exp(e.z - m.z) * inv.z, fast_exp(e.w - m.w) * inv.w);
                put_alpha(c0 + l, a);
                acc4(xa0, mul4(a, xs.x));
                acc4(xa1, mul4(a, xs.y));
            }
            xa0 = halfwave_reduce(xa0, true, SumOp());
            xa1 = halfwave_reduce(xa1, true, SumOp());
        }
        const float a0 = pick4(xa0, h), a1 = pick4(xa1, h);
        const float4 o = make_float4(w.w0.x * a0 + w.w1.x * a1 + b4.x, w.w0.y * a0 + w.w1.y * a1 + b4.y,
                                     w.w0.z * a0 + w.w1.z * a1 + b4.z, w.w0.w * a0 + w.w1.w * a1 + b4.w);
        reinterpret_cast<float4*>(out + (size_t)i * H)[l] = elu4(o);
    }
}

// The whole backward pass of the layer, by target (see the head of the file).  g_out is either read ([n,128]) or -
// POOLED, for a critic of one attention layer feeding the readout directly - not used here; the second layer's
// variant lives in gat.hip.
__global__ __launch_bounds__(THREADS) void k_gat1_bwd(int n, const float2* __restrict__ x, const float* __restrict__ W,
                                                      const float* __restrict__ coef, const float* __restrict__ g_out,
                                                      const float* __restrict__ out, Lists g,
                                                      const float* __restrict__ alpha_e,
                                                      const float* __restrict__ alpha_loop,
                                                      float* __restrict__ partial /*[grid][P_WIDTH]*/) {
    __shared__ float red[HW_PER_BLOCK][P_WIDTH];
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5, h = l >> 3;
    const float4 cs0 = ld4(coef), cs1 = ld4(coef + 4), cd0 = ld4(coef + 8), cd1 = ld4(coef + 12), ce4 = ld4(coef + 16);
    const LaneW w = lane_weights(W, l);
    const float4* al_loop4 = reinterpret_cast<const float4*>(alpha_loop);
    const bool has_edges = g.row_ptr[n] > 0;
    const int* src_safe = has_edges ? g.src : g.row_ptr;
    const float* ea_safe = has_edges ? g.ea : g.loop_ea;
    const float4* al_safe = reinterpret_cast<const float4*>(has_edges ? alpha_e : alpha_loop);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gsum = zero4, dw0 = zero4, dw1 = zero4;             // per channel (this lane's four)
    float4 dce = zero4, ss0 = zero4, ss1 = zero4, sd0 = zero4, sd1 = zero4;   // per head, this lane's entries
    NodeRange r = block_nodes(n);
    for (int i = r.begin + hw; i < r.end; i += HW_PER_BLOCK) {
        const float4 o = ld4(out + (size_t)i * H + 4 * l);
        float4 gg = ld4(g_out + (size_t)i * H + 4 * l);
        const int beg = g.row_ptr[i], deg = g.row_ptr[i + 1] - beg;
        const float2 xi = x[i];
        const float lea = g.loop_ea[i];
        const float4 all4 = al_loop4[i];
        gg.x *= o.x > 0.f ? 1.f : o.x + 1.f; gg.y *= o.y > 0.f ? 1.f : o.y + 1.f;
        gg.z *= o.z > 0.f ? 1.f : o.z + 1.f; gg.w *= o.w > 0.f ? 1.f : o.w + 1.f;
        acc4(gsum, gg);
        // gw_f[h] = <g[h,:], W[h,:,f]>: summed over the eight lanes of a head, then all four heads to every lane
        const float v0 = head_sum(dot4(gg, w.w0)), v1 = head_sum(dot4(gg, w.w1));
        const float4 gw0 = make_float4(__shfl(v0, 0, 32), __shfl(v0, 8, 32), __shfl(v0, 16, 32), __shfl(v0, 24, 32));
        const float4 gw1 = make_float4(__shfl(v1, 0, 32), __shfl(v1, 8, 32), __shfl(v1, 16, 32), __shfl(v1, 24, 32));
        const float4 ad4 = lin2(xi, cd0, cd1);
        // entry idx: edges 0..deg-1, then the self loop; lanes past the list get alpha 0
        auto entry = [&](int idx, float2& xs, float& ea, float4& al, float4& slope) {
            const bool is_edge = idx < deg;
            const int p = is_edge ? beg + idx : 0;
            const int sv = src_safe[p];
            const float eav = ea_safe[p];
            const float4 alv = al_safe[p];
            const int s = is_edge ? sv : i;
            ea = is_edge ? eav : lea;
            al = is_edge ? alv : (idx == deg ? all4 : zero4);
            xs = x[s];
            const float4 as4 = lin2(xs, cs0, cs1);
            slope.x = as4.x + ad4.x + ea * ce4.x > 0.f ? 1.f : NEG_SLOPE;
            slope.y = as4.y + ad4.y + ea * ce4.y > 0.f ? 1.f : NEG_SLOPE;
            slope.z = as4.z + ad4.z + ea * ce4.z > 0.f ? 1.f : NEG_SLOPE;
            slope.w = as4.w + ad4.w + ea * ce4.w > 0.f ? 1.f : NEG_SLOPE;
        };
        auto tally = [&](float4 dr, float2 xs, float ea) {
            acc4(dce, mul4(dr, ea));
            acc4(ss0, mul4(dr, xs.x));
            acc4(ss1, mul4(dr, xs.y));
            acc4(sd0, mul4(dr, xi.x));
            acc4(sd1, mul4(dr, xi.y));
        };
        float4 xa0, xa1;
        if (deg < 32) {
            float2 xs;
            float ea;
            float4 al, slope;
            entry(l, xs, ea, al, slope);
            const bool wide = deg > 3;
            const float4 da = lin2(xs, gw0, gw1);
            const float4 sdot = halfwave_reduce(mul44(al, da), wide, SumOp());
            const float4 dr = make_float4(al.x * (da.x - sdot.x) * slope.x, al.y * (da.y - sdot.y) * slope.y,
                                          al.z * (da.z - sdot.z) * slope.z, al.w * (da.w - sdot.w) * slope.w);
            tally(dr, xs, ea);
            xa0 = halfwave_reduce(mul4(al, xs.x), wide, SumOp());
            xa1 = halfwave_reduce(mul4(al, xs.y), wide, SumOp());
        } else {
            const int cnt = deg + 1;
            float4 sdot = zero4;
            xa0 = zero4;
            xa1 = zero4;
            for (int c0 = 0; c0 < cnt; c0 += 32) {
                float2 xs;
                float ea;
                float4 al, slope;
                entry(c0 + l, xs, ea, al, slope);
                acc4(sdot, mul44(al, lin2(xs, gw0, gw1)));
                acc4(xa0, mul4(al, xs.x));
                acc4(xa1, mul4(al, xs.y));
            }
            sdot = halfwave_reduce(sdot, true, SumOp());
            xa0 = halfwave_reduce(xa0, true, SumOp());
            xa1 = halfwave_reduce(xa1, true, SumOp());
            for (int c0 = 0; c0 < cnt; c0 += 32) {
                float2 xs;
                float ea;
                float4 al, slope;
                entry(c0 + l, xs, ea, al, slope);
                const float4 da = lin2(xs, gw0, gw1);
                const float4 dr = make_float4(al.x * (da.x - sdot.x) * slope.x, al.y * (da.y - sdot.y) * slope.y,
                                              al.z * (da.z - sdot.z) * slope.z, al.w * (da.w - sdot.w) * slope.w);
                tally(dr, xs, ea);
            }
        }
        acc4(dw0, mul4(gg, pick4(xa0, h)));
        acc4(dw1, mul4(gg, pick4(xa1, h)));
    }
    // block partials: channels as they are, the per-head tallies summed over the lanes first; half-waves in fixed order
    dce = halfwave_reduce(dce, true, SumOp());
    ss0 = halfwave_reduce(ss0, true, SumOp());
    ss1 = halfwave_reduce(ss1, true, SumOp());
    sd0 = halfwave_reduce(sd0, true, SumOp());
    sd1 = halfwave_reduce(sd1, true, SumOp());
    *reinterpret_cast<float4*>(&red[hw][P_BIAS + 4 * l]) = gsum;
    *reinterpret_cast<float4*>(&red[hw][P_DW0 + 4 * l]) = dw0;
    *reinterpret_cast<float4*>(&red[hw][P_DW1 + 4 * l]) = dw1;
    if (l == 0) {
        *reinterpret_cast<float4*>(&red[hw][P_CE]) = dce;
        *reinterpret_cast<float4*>(&red[hw][P_SS]) = ss0;
        *reinterpret_cast<float4*>(&red[hw][P_SS + 4]) = ss1;
        *reinterpret_cast<float4*>(&red[hw][P_SD]) = sd0;
        *reinterpret_cast<float4*>(&red[hw][P_SD + 4]) = sd1;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < P_WIDTH; c += THREADS) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < HW_PER_BLOCK; ++k) s += red[k][c];
        partial[(size_t)blockIdx.x * P_WIDTH + c] = s;
    }
}

// sums[P_WIDTH] (the partial rows added up) -> the parameter gradients
__global__ void k_gat1_finish(const float* __restrict__ sums, const float* __restrict__ W,
                              const float* __restrict__ att_src, const float* __restrict__ att_dst,
                              const float* __restrict__ lin_edge_w, const float* __restrict__ att_edge,
                              float* __restrict__ d_W, float* __restrict__ d_att_src, float* __restrict__ d_att_dst,
                              float* __restrict__ d_ce, float* __restrict__ d_bias, float* __restrict__ d_lin_edge_w,
                              float* __restrict__ d_att_edge) {
    const int c = threadIdx.x;
    if (c >= H) return;
    const int h = c / (H / HEADS);
    const float ss0 = sums[P_SS + h], ss1 = sums[P_SS + 4 + h], sd0 = sums[P_SD + h], sd1 = sums[P_SD + 4 + h];
    d_bias[c] = sums[P_BIAS + c];
    d_W[2 * c] = sums[P_DW0 + c] + att_src[c] * ss0 + att_dst[c] * sd0;
    d_W[2 * c + 1] = sums[P_DW1 + c] + att_src[c] * ss1 + att_dst[c] * sd1;
    d_att_src[c] = W[2 * c] * ss0 + W[2 * c + 1] * ss1;
    d_att_dst[c] = W[2 * c] * sd0 + W[2 * c + 1] * sd1;
    const float dce = sums[P_CE + h];
    if (c < HEADS) d_ce[c] = sums[P_CE + c];
    if (d_lin_edge_w) {
        d_lin_edge_w[c] = dce * att_edge[c];
        d_att_edge[c] = dce * lin_edge_w[c];
    }
}

constexpr int BWD_BLOCKS = 2048;

}  // namespace
}  // namespace aig

using namespace aig;

extern "C" int aig_gat1_fwd(int64_t n, const float* x, const float* W, const float* att_src, const float* att_dst,
                            const float* lin_edge_w, const float* att_edge, const float* bias, const int32_t* row_ptr,
                            const int32_t* csr_src, const float* csr_ea, const float* loop_ea, float* coef, float* out,
                            float* alpha_e, float* alpha_loop, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX, "aig_gat1_fwd: n=%lld out of range", (long long)n);
    AIG_REQUIRE(W && att_src && att_dst && lin_edge_w && att_edge && coef, "aig_gat1_fwd: null parameter");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_gat1_coef, dim3(1), dim3(64), 0, s, W, att_src, att_dst, lin_edge_w, att_edge, coef);
    AIG_LAUNCH_CHECK();
    if (n == 0) return AIG_OK;
    AIG_REQUIRE(x && bias && row_ptr && loop_ea && out && alpha_loop, "aig_gat1_fwd: null pointer");
    const Lists g{row_ptr, csr_src, csr_ea, loop_ea};
    AIG_LAUNCH_TIMED(AIG_KT_GAT1_FWD, k_gat1_fwd, dim3(grid_blocks(n)), dim3(THREADS), 0, s, (int)n,
                     reinterpret_cast<const float2*>(x), W, coef, bias, g, out, alpha_e, alpha_loop);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

extern "C" size_t aig_gat1_bwd_ws_bytes(int64_t n) {
    (void)n;
    return align_up((size_t)BWD_BLOCKS * P_WIDTH * 4) + align_up((size_t)P_WIDTH * 4) + 256;
}

extern "C" int aig_gat1_bwd(int64_t n, const float* x, const float* W, const float* att_src, const float* att_dst,
                            const float* lin_edge_w, const float* att_edge, const float* coef, const float* g_out,
                            const float* out, const int32_t* row_ptr, const int32_t* csr_src, const float* csr_ea,
                            const float* loop_ea, const float* alpha_e, const float* alpha_loop, float* d_W,
                            float* d_att_src, float* d_att_dst, float* d_ce, float* d_bias, float* d_lin_edge_w,
                            float* d_att_edge, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX, "aig_gat1_bwd: n=%lld out of range", (long long)n);
    AIG_REQUIRE(W && att_src && att_dst && lin_edge_w && att_edge && coef && d_W && d_att_src && d_att_dst && d_ce &&
                    d_bias && ws,
                "aig_gat1_bwd: null pointer");
    AIG_REQUIRE((d_lin_edge_w == nullptr) == (d_att_edge == nullptr), "aig_gat1_bwd: d_lin_edge_w and d_att_edge come together");
    if (ws_bytes < aig_gat1_bwd_ws_bytes(n))
        return fail(AIG_EWORKSPACE, "aig_gat1_bwd: workspace %zu < %zu", ws_bytes, aig_gat1_bwd_ws_bytes(n));
    hipStream_t s = static_cast<hipStream_t>(stream);
    Carver c(ws);
    float* partial = c.take<float>((size_t)BWD_BLOCKS * P_WIDTH);
    float* sums = c.take<float>(P_WIDTH);
    int nb = 0;
    if (n > 0) {
        AIG_REQUIRE(x && g_out && out && row_ptr && loop_ea && alpha_loop, "aig_gat1_bwd: null pointer");
        nb = grid_blocks(n, BWD_BLOCKS);
        const Lists g{row_ptr, csr_src, csr_ea, loop_ea};
        AIG_LAUNCH_TIMED(AIG_KT_GAT1_BWD, k_gat1_bwd, dim3(nb), dim3(THREADS), 0, s, (int)n,
                         reinterpret_cast<const float2*>(x), W, coef, g_out, out, g, alpha_e, alpha_loop, partial);
        AIG_LAUNCH_CHECK();
    }
    ColSums sums_job{};
    sums_job.j[0] = ColSum{partial, sums, nb, P_WIDTH, P_WIDTH};     // (no partial rows: every sum is zero)
    hipLaunchKernelGGL(k_reduce_cols, dim3(P_WIDTH), dim3(256), 0, s, sums_job);
    hipLaunchKernelGGL(k_gat1_finish, dim3(1), dim3(H), 0, s, (const float*)sums, W, att_src, att_dst, lin_edge_w, att_edge,
                       d_W, d_att_src, d_att_dst, d_ce, d_bias, d_lin_edge_w, d_att_edge);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}
