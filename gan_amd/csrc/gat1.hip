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
// every list entry is seen by its target).  No floating-point atomics, block partials added in fixed order.
#include <climits>

#include "common.h"
#include "gat_common.h"

namespace aig {
namespace {

constexpr int H = GAT_H, HEADS = GAT_HEADS, THREADS = GAT_THREADS;
// coef[0..7] = cs[f][h] (f major), coef[8..15] = cd[f][h], coef[16..19] = ce[h]
constexpr int COEF = 20;
// columns of a block's partial row in the backward pass
constexpr int P_BIAS = 0, P_DW0 = H, P_DW1 = 2 * H, P_CE = 3 * H, P_SS = 3 * H + 4, P_SD = 3 * H + 12, P_WIDTH = 3 * H + 20;

// coefficient t (< COEF) of the layer: a 32-term dot product, terms in ascending order
__device__ __forceinline__ float coef_of(int t, const float* __restrict__ W, const float* __restrict__ att_src,
                                         const float* __restrict__ att_dst, const float* __restrict__ lin_edge_w,
                                         const float* __restrict__ att_edge) {
    constexpr int C = H / HEADS;
    float s = 0.f;
    if (t < 16) {
        const int which = t >> 3, f = (t >> 2) & 1, h = t & 3;
        const float* a = which ? att_dst : att_src;
        float wv[C], av[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            wv[c] = W[(h * C + c) * 2 + f];
            av[c] = a[h * C + c];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) s += wv[c] * av[c];
    } else {
        const int h = t - 16;
        float wv[C], av[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            wv[c] = lin_edge_w[h * C + c];
            av[c] = att_edge[h * C + c];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) s += wv[c] * av[c];
    }
    return s;
}

// (the forward kernel computes the coefficients itself, per block; this launch serves an empty graph)
__global__ void k_gat1_coef(const float* __restrict__ W, const float* __restrict__ att_src,
                            const float* __restrict__ att_dst, const float* __restrict__ lin_edge_w,
                            const float* __restrict__ att_edge, float* __restrict__ coef) {
    const int t = threadIdx.x;
    if (t < COEF) coef[t] = coef_of(t, W, att_src, att_dst, lin_edge_w, att_edge);
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

// Lane layout.  What a node costs here is arithmetic on its list entries - a handful of numbers each - and a vector
// instruction occupies its SIMD for four cycles whatever its lanes hold: with a half-wave per node (the layout of
// gat.hip, whose rows need 32 lanes) a target with three list entries kept 29 lanes idle through every score,
// exponential and reduction, and the kernels ran at the instruction rate, not the memory's.  So the ENTRY phases take
// eight nodes per wave - group k = lane / 8 owns node k of the wave's current eight, lane j = lane % 8 its list
// entry j (edges, then the self loop; a list of more than eight entries is set aside for the whole wave) - with
// reductions over a group as three DPP steps; the ROW phases (512 bytes per node) keep the half-wave per node and
// take the eight nodes two at a time; the two phases hand each other eight floats per node through LDS or a lane read.
constexpr int GROUP = 8;                       // lanes per node in the entry phases = most list entries taken there
constexpr int NODES_PER_WAVE = 64 / GROUP;
constexpr int WAVES = THREADS / 64;
constexpr int NODES_PER_BLOCK_TURN = WAVES * NODES_PER_WAVE;

template <class Op>
__device__ __forceinline__ float4 group_reduce(float4 v, Op op) {   // over the 8 lanes of a group, result in all 8
    v.x = op(v.x, dpp<DPP_XOR1>(v.x)); v.y = op(v.y, dpp<DPP_XOR1>(v.y));
    v.z = op(v.z, dpp<DPP_XOR1>(v.z)); v.w = op(v.w, dpp<DPP_XOR1>(v.w));
    v.x = op(v.x, dpp<DPP_XOR2>(v.x)); v.y = op(v.y, dpp<DPP_XOR2>(v.y));
    v.z = op(v.z, dpp<DPP_XOR2>(v.z)); v.w = op(v.w, dpp<DPP_XOR2>(v.w));
    v.x = op(v.x, dpp<DPP_HALF_MIRROR>(v.x)); v.y = op(v.y, dpp<DPP_HALF_MIRROR>(v.y));
    v.z = op(v.z, dpp<DPP_HALF_MIRROR>(v.z)); v.w = op(v.w, dpp<DPP_HALF_MIRROR>(v.w));
    return v;
}
template <class Op>
__device__ __forceinline__ float4 wave_reduce(float4 v, Op op) {    // over all 64 lanes, result in all of them
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        v.x = op(v.x, __shfl_xor(v.x, off, 64)); v.y = op(v.y, __shfl_xor(v.y, off, 64));
        v.z = op(v.z, __shfl_xor(v.z, off, 64)); v.w = op(v.w, __shfl_xor(v.w, off, 64));
    }
    return v;
}

// What the entry phases read of a node's list.  Every load is unconditional, on a clamped index (a load under a
// condition compiles to a branch with a wait behind it).
struct ListReader {
    const float2* __restrict__ x;
    Lists g;
    const int* __restrict__ src_safe;
    const float* __restrict__ ea_safe;
    int n;
    __device__ __forceinline__ ListReader(const float2* x_, const Lists& g_, int n_) : x(x_), g(g_), n(n_) {
        const bool has_edges = g.row_ptr[n] > 0;   // clamped list reads need one valid element
        src_safe = has_edges ? g.src : g.row_ptr;
        ea_safe = has_edges ? g.ea : g.loop_ea;
    }
    // The loads behind a node depend on each other (list bounds -> list entry -> the source's features), and a wave
    // takes its nodes a turn (eight of them) at a time: taken one turn at a time, a turn costs three memory round
    // trips and the kernel runs at latency.  So the stages run ahead of the work: in a turn the wave requests the
    // bounds of the turn after next, the entries of the next turn (their bounds arrived a turn ago) and the
    // features behind this turn's entries, then works.
    struct Bounds {      // stage 1, per node
        int ic, beg, deg;
        float2 xi;
        float lea;
    };
    struct Entry {       // stage 2, per list entry
        int p, s;        // place in the by-target arrays (0 past the edges), source (the node itself past the edges)
        float ea;
    };
    __device__ __forceinline__ Bounds bounds(int i) const {
        const int ic = min(i, n - 1);
        const int b = g.row_ptr[ic], e = g.row_ptr[ic + 1];
        return {ic, b, e - b, x[ic], g.loop_ea[ic]};
    }
    __device__ __forceinline__ Entry entry_of(const Bounds& b, int idx) const {
        const bool is_edge = idx < b.deg;
        const int p = is_edge ? b.beg + idx : 0;
        const int sv = src_safe[p];
        const float eav = ea_safe[p];
        return {p, is_edge ? sv : b.ic, is_edge ? eav : b.lea};
    }
    // entry idx of node i (list at beg, deg edges, then the self loop with attribute lea): its place in the
    // by-target arrays (edges only), the source's features and the edge attribute
    __device__ __forceinline__ void entry(int i, int beg, int deg, float lea, int idx, int& p, float2& xs, float& ea) const {
        const bool is_edge = idx < deg;
        p = is_edge ? beg + idx : 0;
        const int sv = src_safe[p];
        const float eav = ea_safe[p];
        xs = x[is_edge ? sv : i];
        ea = is_edge ? eav : lea;
    }
};

struct Coef {
    float4 cs0, cs1, cd0, cd1, ce;
    __device__ __forceinline__ explicit Coef(const float* __restrict__ c)
        : cs0(ld4(c)), cs1(ld4(c + 4)), cd0(ld4(c + 8)), cd1(ld4(c + 12)), ce(ld4(c + 16)) {}
    // pre-activation attention scores of an entry for the four heads (ad4: the target's own term)
    __device__ __forceinline__ float4 raw(float2 xs, float4 ad4, float ea) const {
        const float4 as4 = lin2(xs, cs0, cs1);
        return make_float4(as4.x + ad4.x + ea * ce.x, as4.y + ad4.y + ea * ce.y, as4.z + ad4.z + ea * ce.z,
                           as4.w + ad4.w + ea * ce.w);
    }
};
__device__ __forceinline__ float4 lrelu4(float4 r) { return make_float4(lrelu(r.x), lrelu(r.y), lrelu(r.z), lrelu(r.w)); }
__device__ __forceinline__ float4 exp_sub4(float4 e, float4 m) {
    return make_float4(fast_exp(e.x - m.x), fast_exp(e.y - m.y), fast_exp(e.z - m.z), fast_exp(e.w - m.w));
}
__device__ __forceinline__ float4 inv_den4(float4 den) {
    return make_float4(1.f / (den.x + SOFTMAX_EPS), 1.f / (den.y + SOFTMAX_EPS), 1.f / (den.z + SOFTMAX_EPS),
                       1.f / (den.w + SOFTMAX_EPS));
}

// The layer's twenty coefficients (products of weights only) are computed here, by every block for itself, instead
// of by a launch in front of this one; block 0 leaves them in `coef` for the backward pass.
__global__ __launch_bounds__(THREADS) void k_gat1_fwd(int n, const float2* __restrict__ x, const float* __restrict__ W,
                                                      const float* __restrict__ att_src, const float* __restrict__ att_dst,
                                                      const float* __restrict__ lin_edge_w, const float* __restrict__ att_edge,
                                                      float* __restrict__ coef, const float* __restrict__ bias,
                                                      Lists g, float* __restrict__ out, float* __restrict__ alpha_e,
                                                      float* __restrict__ alpha_loop) {
    __shared__ float s_xa[WAVES][NODES_PER_WAVE][8];      // per node: xa[f][h] = sum over its list of alpha_h * x_f
    __shared__ __attribute__((aligned(16))) float s_coef[COEF];
    if (threadIdx.x < COEF) {
        const float v = coef_of(threadIdx.x, W, att_src, att_dst, lin_edge_w, att_edge);
        s_coef[threadIdx.x] = v;
        if (blockIdx.x == 0) coef[threadIdx.x] = v;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k = lane >> 3, j = lane & 7;       // entry phase: node of the wave's eight, list entry
    const int l = lane & 31, hv = lane >> 5, h = l >> 3;   // row phase: channel quad, which of the two nodes, head
    const LaneW w = lane_weights(W, l);
    const float4 b4 = ld4(bias + 4 * l);
    float4* alpha_e4 = reinterpret_cast<float4*>(alpha_e);
    float4* alpha_loop4 = reinterpret_cast<float4*>(alpha_loop);
    const ListReader rd(x, g, n);
    const float NEG_INF = -__builtin_huge_valf();
    const float4 ninf4 = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF), zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    NodeRange r = block_nodes(n);
    const int first = r.begin + wv * NODES_PER_WAVE;
    ListReader::Bounds bd = rd.bounds(first + k), bd1 = rd.bounds(first + NODES_PER_BLOCK_TURN + k);
    ListReader::Entry en = rd.entry_of(bd, j);
    __syncthreads();
    const Coef cf(s_coef);
    for (int base = first; base < r.end; base += NODES_PER_BLOCK_TURN) {
        // ---- entry phase: node base + k, entry j (requests of the turns ahead first)
        const float2 xs = x[en.s];
        const ListReader::Entry en1 = rd.entry_of(bd1, j);
        const ListReader::Bounds bd2 = rd.bounds(base + 2 * NODES_PER_BLOCK_TURN + k);
        const int i = base + k;
        const bool live = i < r.end;
        const int ic = bd.ic, beg = bd.beg, deg = bd.deg;
        const float2 xi = bd.xi;
        const float lea = bd.lea;
        const bool is_long = deg + 1 > GROUP;
        const int p = en.p;
        const float ea = en.ea;
        const float4 ad4 = lin2(xi, cf.cd0, cf.cd1);
        const float4 e = j <= deg ? lrelu4(cf.raw(xs, ad4, ea)) : ninf4;
        const float4 m = group_reduce(e, MaxOp());
        float4 a = exp_sub4(e, m);
        const float4 inv = inv_den4(group_reduce(a, SumOp()));
        a = mul44(a, inv);
        if (live && !is_long) {
            if (j < deg) alpha_e4[p] = a;
            if (j == deg) alpha_loop4[i] = a;
        }
        const float4 xa0 = group_reduce(mul4(a, xs.x), SumOp()), xa1 = group_reduce(mul4(a, xs.y), SumOp());
        if (j == 0) {
            *reinterpret_cast<float4*>(&s_xa[wv][k][0]) = xa0;
            *reinterpret_cast<float4*>(&s_xa[wv][k][4]) = xa1;
        }
        // ---- lists of more than GROUP entries: the whole wave takes them one at a time, 64 entries per sweep
        unsigned long long longs = __ballot(live && is_long && j == 0);
        while (longs) {
            const int kk = (__ffsll((long long)longs) - 1) >> 3;
            longs &= longs - 1;
            const int li = __shfl(ic, kk * GROUP, 64), lbeg = __shfl(beg, kk * GROUP, 64), ldeg = __shfl(deg, kk * GROUP, 64);
            const float llea = __shfl(lea, kk * GROUP, 64);
            const float2 lxi = make_float2(__shfl(xi.x, kk * GROUP, 64), __shfl(xi.y, kk * GROUP, 64));
            const float4 lad4 = lin2(lxi, cf.cd0, cf.cd1);
            const int cnt = ldeg + 1;
            auto score = [&](int idx, int& pp, float2& xx) {
                float eea;
                rd.entry(li, lbeg, ldeg, llea, idx, pp, xx, eea);
                return idx <= ldeg ? lrelu4(cf.raw(xx, lad4, eea)) : ninf4;
            };
            float4 mm = ninf4;
            for (int c0 = 0; c0 < cnt; c0 += 64) {
                int pp;
                float2 xx;
                const float4 ee = score(c0 + lane, pp, xx);
                mm.x = fmaxf(mm.x, ee.x); mm.y = fmaxf(mm.y, ee.y); mm.z = fmaxf(mm.z, ee.z); mm.w = fmaxf(mm.w, ee.w);
            }
            mm = wave_reduce(mm, MaxOp());
            float4 den = zero4;
            for (int c0 = 0; c0 < cnt; c0 += 64) {
                int pp;
                float2 xx;
                acc4(den, exp_sub4(score(c0 + lane, pp, xx), mm));
            }
            const float4 linv = inv_den4(wave_reduce(den, SumOp()));
            float4 la0 = zero4, la1 = zero4;
            for (int c0 = 0; c0 < cnt; c0 += 64) {
                int pp;
                float2 xx;
                const float4 aa = mul44(exp_sub4(score(c0 + lane, pp, xx), mm), linv);
                if (c0 + lane < ldeg) alpha_e4[pp] = aa;
                if (c0 + lane == ldeg) alpha_loop4[li] = aa;
                acc4(la0, mul4(aa, xx.x));
                acc4(la1, mul4(aa, xx.y));
            }
            la0 = wave_reduce(la0, SumOp());
            la1 = wave_reduce(la1, SumOp());
            if (lane == 0) {
                *reinterpret_cast<float4*>(&s_xa[wv][kk][0]) = la0;
                *reinterpret_cast<float4*>(&s_xa[wv][kk][4]) = la1;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- row phase: out[i] = elu(W xa[i] + b), two nodes per turn
#pragma unroll
        for (int t = 0; t < NODES_PER_WAVE / 2; ++t) {
            const int kk = 2 * t + hv, ii = base + kk;
            const float c0 = s_xa[wv][kk][h], c1 = s_xa[wv][kk][4 + h];
            const float4 o = make_float4(w.w0.x * c0 + w.w1.x * c1 + b4.x, w.w0.y * c0 + w.w1.y * c1 + b4.y,
                                         w.w0.z * c0 + w.w1.z * c1 + b4.z, w.w0.w * c0 + w.w1.w * c1 + b4.w);
            if (ii < r.end) reinterpret_cast<float4*>(out + (size_t)ii * H)[l] = elu4(o);
        }
        __builtin_amdgcn_wave_barrier();      // the next turn overwrites s_xa
        bd = bd1;
        bd1 = bd2;
        en = en1;
    }
}

// The whole backward pass of the layer, by target (see the head of the file).  Row phase A: g = g_out * elu'(out),
// its sum (bias gradient) and gw_f[h] = <g[h,:], W[h,:,f]> per node; entry phase B: d_raw of every entry and the
// per-head tallies, xa of every node; row phase C: dW += g (x) xa.
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_gat1_bwd(int n, const float2* __restrict__ x, const float* __restrict__ W,
                                                      const float* __restrict__ coef, const float* __restrict__ g_out,
                                                      const float* __restrict__ out, Lists g,
                                                      const float* __restrict__ alpha_e,
                                                      const float* __restrict__ alpha_loop,
                                                      float* __restrict__ partial /*[grid][P_WIDTH]*/) {
    // LDS: the rows of g kept from phase A for phase C while the nodes are walked, the partial rows afterwards
    constexpr int RED_ROWS = 4 * WAVES;      // a row per half-wave from the main walk, another from the long lists
    __shared__ __attribute__((aligned(16))) float s_buf[RED_ROWS * P_WIDTH];
    static_assert(RED_ROWS * P_WIDTH >= WAVES * NODES_PER_WAVE * H, "the g rows must fit where the partial rows go");
    float (*s_g)[NODES_PER_WAVE][H] = reinterpret_cast<float (*)[NODES_PER_WAVE][H]>(s_buf);
    float (*red)[P_WIDTH] = reinterpret_cast<float (*)[P_WIDTH]>(s_buf);
    __shared__ float s_gw[WAVES][NODES_PER_WAVE][8];      // per node: gw[f][h]
    __shared__ float s_xa[WAVES][NODES_PER_WAVE][8];      // per node: xa[f][h]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k = lane >> 3, j = lane & 7;
    const int l = lane & 31, hv = lane >> 5, h = l >> 3;
    const Coef cf(coef);
    const LaneW w = lane_weights(W, l);
    const float4* al_loop4 = reinterpret_cast<const float4*>(alpha_loop);
    const ListReader rd(x, g, n);
    const float4* al_safe = reinterpret_cast<const float4*>(g.row_ptr[n] > 0 ? alpha_e : alpha_loop);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gsum = zero4, dw0 = zero4, dw1 = zero4;             // per channel (this lane's four)
    float4 dce = zero4, ss0 = zero4, ss1 = zero4, sd0 = zero4, sd1 = zero4;   // per head, this lane's entries
    auto slope4 = [&](float4 raw) {
        return make_float4(raw.x > 0.f ? 1.f : NEG_SLOPE, raw.y > 0.f ? 1.f : NEG_SLOPE, raw.z > 0.f ? 1.f : NEG_SLOPE,
                           raw.w > 0.f ? 1.f : NEG_SLOPE);
    };
    NodeRange r = block_nodes(n);
    const int first = r.begin + wv * NODES_PER_WAVE;
    ListReader::Bounds bd = rd.bounds(first + k), bd1 = rd.bounds(first + NODES_PER_BLOCK_TURN + k);
    ListReader::Entry en = rd.entry_of(bd, j);
    for (int base = first; base < r.end; base += NODES_PER_BLOCK_TURN) {
        // requests first: what the entry phase of this turn gathers, the eight rows, and the stages of the turns ahead
        const float2 xs = x[en.s];
        const float4 alv = al_safe[en.p], all4 = al_loop4[bd.ic];
        float4 ro[NODES_PER_WAVE / 2], rg[NODES_PER_WAVE / 2];
#pragma unroll
        for (int t = 0; t < NODES_PER_WAVE / 2; ++t) {
            const int iic = min(base + 2 * t + hv, n - 1);
            ro[t] = ld4(out + (size_t)iic * H + 4 * l);
            rg[t] = ld4(g_out + (size_t)iic * H + 4 * l);
        }
        const ListReader::Entry en1 = rd.entry_of(bd1, j);
        const ListReader::Bounds bd2 = rd.bounds(base + 2 * NODES_PER_BLOCK_TURN + k);
        // ---- row phase A
#pragma unroll
        for (int t = 0; t < NODES_PER_WAVE / 2; ++t) {
            const int kk = 2 * t + hv, ii = base + kk;
            const float4 o = ro[t];
            float4 gg = rg[t];
            if (ii >= r.end) gg = zero4;
            gg.x *= o.x > 0.f ? 1.f : o.x + 1.f; gg.y *= o.y > 0.f ? 1.f : o.y + 1.f;
            gg.z *= o.z > 0.f ? 1.f : o.z + 1.f; gg.w *= o.w > 0.f ? 1.f : o.w + 1.f;
            *reinterpret_cast<float4*>(&s_g[wv][kk][4 * l]) = gg;
            acc4(gsum, gg);
            const float v0 = head_sum(dot4(gg, w.w0)), v1 = head_sum(dot4(gg, w.w1));
            if ((l & 7) == 0) {
                s_gw[wv][kk][h] = v0;
                s_gw[wv][kk][4 + h] = v1;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- entry phase B: node base + k, entry j
        const int i = base + k;
        const bool live = i < r.end;
        const int deg = bd.deg;
        const float2 xi = bd.xi;
        const bool is_long = deg + 1 > GROUP;
        {
            const float ea = en.ea;
            const float4 al = (live && !is_long) ? (j < deg ? alv : (j == deg ? all4 : zero4)) : zero4;
            const float4 gw0 = ld4(&s_gw[wv][k][0]), gw1 = ld4(&s_gw[wv][k][4]);
            const float4 slope = slope4(cf.raw(xs, lin2(xi, cf.cd0, cf.cd1), ea));
            const float4 da = lin2(xs, gw0, gw1);
            const float4 sdot = group_reduce(mul44(al, da), SumOp());
            const float4 dr = make_float4(al.x * (da.x - sdot.x) * slope.x, al.y * (da.y - sdot.y) * slope.y,
                                          al.z * (da.z - sdot.z) * slope.z, al.w * (da.w - sdot.w) * slope.w);
            acc4(dce, mul4(dr, ea));
            acc4(ss0, mul4(dr, xs.x));
            acc4(ss1, mul4(dr, xs.y));
            acc4(sd0, mul4(dr, xi.x));
            acc4(sd1, mul4(dr, xi.y));
            const float4 xa0 = group_reduce(mul4(al, xs.x), SumOp()), xa1 = group_reduce(mul4(al, xs.y), SumOp());
            if (j == 0) {
                *reinterpret_cast<float4*>(&s_xa[wv][k][0]) = xa0;
                *reinterpret_cast<float4*>(&s_xa[wv][k][4]) = xa1;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- row phase C
#pragma unroll
        for (int t = 0; t < NODES_PER_WAVE / 2; ++t) {
            const int kk = 2 * t + hv;
            const float4 gg = ld4(&s_g[wv][kk][4 * l]);
            acc4(dw0, mul4(gg, s_xa[wv][kk][h]));
            acc4(dw1, mul4(gg, s_xa[wv][kk][4 + h]));
        }
        __builtin_amdgcn_wave_barrier();      // the next turn overwrites s_gw / s_xa / s_g
        bd = bd1;
        bd1 = bd2;
        en = en1;
    }
    // The main walk's sums go to their partial rows now, so that the registers behind them are free for the second
    // walk (which is rare work with many values in flight: kept in one piece with the main walk it set the
    // kernel's register count and cost it a wave per SIMD).
    auto flush = [&](int row) {
        dce = wave_reduce(dce, SumOp());
        ss0 = wave_reduce(ss0, SumOp());
        ss1 = wave_reduce(ss1, SumOp());
        sd0 = wave_reduce(sd0, SumOp());
        sd1 = wave_reduce(sd1, SumOp());
        // (the per-head tallies of both half-waves were summed over the whole wave: taken from the first)
        if (hv) dce = ss0 = ss1 = sd0 = sd1 = zero4;
        *reinterpret_cast<float4*>(&red[row][P_BIAS + 4 * l]) = gsum;
        *reinterpret_cast<float4*>(&red[row][P_DW0 + 4 * l]) = dw0;
        *reinterpret_cast<float4*>(&red[row][P_DW1 + 4 * l]) = dw1;
        if (l == 0) {
            *reinterpret_cast<float4*>(&red[row][P_CE]) = dce;
            *reinterpret_cast<float4*>(&red[row][P_SS]) = ss0;
            *reinterpret_cast<float4*>(&red[row][P_SS + 4]) = ss1;
            *reinterpret_cast<float4*>(&red[row][P_SD]) = sd0;
            *reinterpret_cast<float4*>(&red[row][P_SD + 4]) = sd1;
        }
        gsum = dw0 = dw1 = dce = ss0 = ss1 = sd0 = sd1 = zero4;
    };
    __syncthreads();      // every wave is through with its rows of g: the partial rows take their place
    flush(2 * wv + hv);
    // ---- lists of more than GROUP entries (left out above: their alpha was taken as zero): a second walk over the
    // block's nodes that only reads the list bounds, and the whole wave on every such node, 64 entries per sweep
    for (int base = r.begin + wv * NODES_PER_WAVE; base < r.end; base += NODES_PER_BLOCK_TURN) {
        const int i = base + k;
        const int ic = min(i, n - 1);
        const int beg = g.row_ptr[ic], deg = g.row_ptr[ic + 1] - beg;
        unsigned long long longs = __ballot(i < r.end && deg + 1 > GROUP && j == 0);
        while (longs) {
            const int kk = (__ffsll((long long)longs) - 1) >> 3;
            longs &= longs - 1;
            const int li = base + kk, lbeg = __shfl(beg, kk * GROUP, 64), ldeg = __shfl(deg, kk * GROUP, 64);
            const float llea = g.loop_ea[li];
            const float2 lxi = x[li];
            const float4 lad4 = lin2(lxi, cf.cd0, cf.cd1);
            const float4 lall4 = al_loop4[li];
            // the node's row again (both half-waves hold it; only the first adds to the channel sums)
            const float4 o = ld4(out + (size_t)li * H + 4 * l);
            float4 gg = ld4(g_out + (size_t)li * H + 4 * l);
            gg.x *= o.x > 0.f ? 1.f : o.x + 1.f; gg.y *= o.y > 0.f ? 1.f : o.y + 1.f;
            gg.z *= o.z > 0.f ? 1.f : o.z + 1.f; gg.w *= o.w > 0.f ? 1.f : o.w + 1.f;
            const float v0 = head_sum(dot4(gg, w.w0)), v1 = head_sum(dot4(gg, w.w1));
            const float4 gw0 = make_float4(__shfl(v0, 0, 64), __shfl(v0, 8, 64), __shfl(v0, 16, 64), __shfl(v0, 24, 64));
            const float4 gw1 = make_float4(__shfl(v1, 0, 64), __shfl(v1, 8, 64), __shfl(v1, 16, 64), __shfl(v1, 24, 64));
            const int cnt = ldeg + 1;
            auto entry = [&](int idx, float2& xx, float& eea, float4& al) {
                int pp;
                rd.entry(li, lbeg, ldeg, llea, idx, pp, xx, eea);
                const float4 alv = al_safe[pp];
                al = idx < ldeg ? alv : (idx == ldeg ? lall4 : zero4);
            };
            float4 sdot = zero4, la0 = zero4, la1 = zero4;
            for (int c0 = 0; c0 < cnt; c0 += 64) {
                float2 xx;
                float eea;
                float4 al;
                entry(c0 + lane, xx, eea, al);
                acc4(sdot, mul44(al, lin2(xx, gw0, gw1)));
                acc4(la0, mul4(al, xx.x));
                acc4(la1, mul4(al, xx.y));
            }
            sdot = wave_reduce(sdot, SumOp());
            la0 = wave_reduce(la0, SumOp());
            la1 = wave_reduce(la1, SumOp());
            for (int c0 = 0; c0 < cnt; c0 += 64) {
                float2 xx;
                float eea;
                float4 al;
                entry(c0 + lane, xx, eea, al);
                const float4 slope = slope4(cf.raw(xx, lad4, eea));
                const float4 da = lin2(xx, gw0, gw1);
                const float4 dr = make_float4(al.x * (da.x - sdot.x) * slope.x, al.y * (da.y - sdot.y) * slope.y,
                                              al.z * (da.z - sdot.z) * slope.z, al.w * (da.w - sdot.w) * slope.w);
                acc4(dce, mul4(dr, eea));
                acc4(ss0, mul4(dr, xx.x));
                acc4(ss1, mul4(dr, xx.y));
                acc4(sd0, mul4(dr, lxi.x));
                acc4(sd1, mul4(dr, lxi.y));
            }
            if (hv == 0) {
                acc4(dw0, mul4(gg, pick4(la0, h)));
                acc4(dw1, mul4(gg, pick4(la1, h)));
            }
        }
    }
    flush(2 * WAVES + 2 * wv + hv);
    // block partials: the rows added in fixed order
    __syncthreads();
    for (int c = threadIdx.x; c < P_WIDTH; c += THREADS) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < RED_ROWS; ++q) s += red[q][c];
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

constexpr int BWD_BLOCKS = 2048;     // partial rows the workspace holds
constexpr int BWD_GRID = 768;        // blocks launched: four waves per SIMD are resident, and a block's closing sums are
                                     // a fixed cost - fewer, longer blocks (measured at 100k nodes: 768 31 us, 1024 33, 2048 36)

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
    if (n == 0) {
        hipLaunchKernelGGL(k_gat1_coef, dim3(1), dim3(64), 0, s, W, att_src, att_dst, lin_edge_w, att_edge, coef);
        AIG_LAUNCH_CHECK();
        return AIG_OK;
    }
    AIG_REQUIRE(x && bias && row_ptr && loop_ea && out && alpha_loop, "aig_gat1_fwd: null pointer");
    const Lists g{row_ptr, csr_src, csr_ea, loop_ea};
    AIG_LAUNCH_TIMED(AIG_KT_GAT1_FWD, k_gat1_fwd, dim3(grid_blocks(n)), dim3(THREADS), 0, s, (int)n,
                     reinterpret_cast<const float2*>(x), W, att_src, att_dst, lin_edge_w, att_edge, coef, bias, g, out, alpha_e,
                     alpha_loop);
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
        static const int env_blocks = [] {       // tuning aid
            const char* e = getenv("AIG_GAT1_BWD_BLOCKS");
            const int v = e ? atoi(e) : 0;
            return v >= 8 && v <= BWD_BLOCKS ? v / 8 * 8 : 0;
        }();
        nb = grid_blocks(n, env_blocks ? env_blocks : BWD_GRID);
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
