// Edge-aware graph attention layer of the critic (GATConv with edge_dim=1, heads=4, 32 channels
// per head, concat) as CSR-by-target gather / segmented-softmax / aggregate kernels, forward and
// backward.  Everything here is HBM/L2-bound: rows of 512 B are gathered by source id, each target
// owned by half a wavefront (32 lanes x float4), no floating-point atomics anywhere, so results
// are bitwise reproducible run to run.
#include <climits>
#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"
#include "gat_common.h"
#include "radix.h"

namespace aig {
namespace {

constexpr int H = GAT_H, HEADS = GAT_HEADS, THREADS = GAT_THREADS, HW_PER_BLOCK = GAT_HW_PER_BLOCK;

// ------------------------------------------------------------------------------------------ CSR
__global__ void k_csr_keys(int n, int E, const long long* __restrict__ ei, unsigned* __restrict__ key, int* __restrict__ ids) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < E; e += gridDim.x * blockDim.x) {
        long long s = ei[e], d = ei[(size_t)E + e];
        bool ok = s != d && s >= 0 && s < n && d >= 0 && d < n;  // self loops are replaced, strays dropped
        key[e] = ok ? (unsigned)d : (unsigned)n;
        ids[e] = e;
    }
}

__device__ __forceinline__ int lower_bound_u(const unsigned* __restrict__ a, int len, unsigned v) {
    int lo = 0, hi = len;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void k_csr_ptr(int n, int E, const unsigned* __restrict__ sorted_key, int* __restrict__ ptr) {
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v <= n; v += gridDim.x * blockDim.x)
        ptr[v] = lower_bound_u(sorted_key, E, (unsigned)v);
}

__global__ void k_csr_fill(int n, int E, const long long* __restrict__ ei, const float* __restrict__ ea,
                           const unsigned* __restrict__ sorted_key, const int* __restrict__ sorted_eid,
                           int* __restrict__ csr_src, float* __restrict__ csr_ea, int* __restrict__ csr_eid,
                           unsigned* __restrict__ key_src, int* __restrict__ ids) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < E; p += gridDim.x * blockDim.x) {
        unsigned k = sorted_key[p];
        int eid = sorted_eid[p];
        unsigned ks = (unsigned)n;
        if (k < (unsigned)n) {
            int s = (int)ei[eid];
            csr_src[p] = s;
            csr_ea[p] = ea ? ea[eid] : 0.f;
            csr_eid[p] = eid;
            ks = (unsigned)s;
        }
        key_src[p] = ks;
        ids[p] = p;
    }
}

__global__ void k_csc_fill(int E, const int* __restrict__ row_ptr_n, const unsigned* __restrict__ sorted_dst,
                           const int* __restrict__ csc_pos, int* __restrict__ csc_dst) {
    const int ek = row_ptr_n[0];
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < ek; q += gridDim.x * blockDim.x)
        csc_dst[q] = (int)sorted_dst[csc_pos[q]];
}

// self-loop attribute = mean attribute of the edges entering the node (add_self_loops fill 'mean')
__global__ void k_loop_attr(int n, const int* __restrict__ row_ptr, const float* __restrict__ csr_ea,
                            float* __restrict__ loop_ea) {
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
        int b = row_ptr[v], e = row_ptr[v + 1];
        float s = 0.f;
        for (int p = b; p < e; ++p) s += csr_ea[p];
        loop_ea[v] = e > b ? s / (float)(e - b) : 0.f;
    }
}

int key_bits(int64_t n) {
    int b = 1;
    while ((1ll << b) <= n) ++b;
    return b;
}

size_t sort_temp_bytes(int64_t E, int bits) {
    return rs::temp_bytes<unsigned, int>((size_t)E, bits);
}

constexpr int SEG = 32;         // an out-edge list of at most this many edges is one half-wave's work in the backward pass
constexpr int HUB_CHUNK = 256;  // a longer list (a "hub") is cut into chunks of this many edges, one block each
constexpr int HUB_BLOCKS = 1024;        // blocks that gather hub chunks (multiple of 8)
constexpr int HUB_FINISH_BLOCKS = 64;   // blocks that add up the chunk rows, a hub at a time
constexpr int BWD_DST_BLOCKS = 1024;    // grids of the two backward kernels: what is resident at once at their occupancy
constexpr int BWD_SRC_BLOCKS = 1280;
// upper bounds for E edges: a hub has more than SEG edges, and its chunks number fan-out / HUB_CHUNK rounded up
inline size_t hubs_max(int64_t E) { return (size_t)(E > 0 ? E : 0) / (SEG + 1) + 1; }
inline size_t hub_chunks_max(int64_t E) { return (size_t)(E > 0 ? E : 0) / HUB_CHUNK + hubs_max(E); }

// per node: is it a hub, and how many chunks does its list make
__global__ void k_hub_count(int n, const int* __restrict__ col_ptr, int* __restrict__ flag, int* __restrict__ chunks) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j <= n; j += gridDim.x * blockDim.x) {
        const int d = j < n ? col_ptr[j + 1] - col_ptr[j] : 0;
        flag[j] = d > SEG;
        chunks[j] = d > SEG ? (d + HUB_CHUNK - 1) / HUB_CHUNK : 0;
    }
}
// tab[0] = {hubs, chunks of all hubs, 0, 0}; tab[1 + k] = {node, first position in the by-source list,
// fan-out, first chunk} of the k-th hub in node order (rank / cbase: exclusive sums of flag / chunks)
__global__ void k_hub_tab(int n, const int* __restrict__ col_ptr, const int* __restrict__ rank,
                          const int* __restrict__ cbase, int4* __restrict__ tab) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j <= n; j += gridDim.x * blockDim.x) {
        if (j == n) { tab[0] = make_int4(rank[n], cbase[n], 0, 0); continue; }
        if (rank[j + 1] != rank[j]) tab[1 + rank[j]] = make_int4(j, col_ptr[j], col_ptr[j + 1] - col_ptr[j], cbase[j]);
    }
}

size_t scan_temp_bytes(int64_t n) {
    size_t bytes = 0;
    int* p = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, p, p, 0, (size_t)n, rocprim::plus<int>(), hipStream_t(0));
    return bytes;
}

// a_s[i,h] = <xw[i,h,:], att_src[h,:]>, a_d likewise; block 0 also leaves the edge coefficients
// ce[h] = <lin_edge.weight[h,:], att_edge[h,:]> (edge_dim = 1: the attention term of an edge is attr * ce[h])
__global__ __launch_bounds__(THREADS) void k_gat_node_alpha(int n, const float* __restrict__ xw,
                                                            const float* __restrict__ att_src,
                                                            const float* __restrict__ att_dst,
                                                            const float* __restrict__ lin_edge_w,
                                                            const float* __restrict__ att_edge,
                                                            float* __restrict__ a_s, float* __restrict__ a_d,
                                                            float* __restrict__ ce) {
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5, h = l >> 3;
    if (blockIdx.x == 0 && threadIdx.x < HEADS && ce) {
        float s = 0.f;
        for (int c = 0; c < H / HEADS; ++c)
            s += lin_edge_w[threadIdx.x * (H / HEADS) + c] * att_edge[threadIdx.x * (H / HEADS) + c];
        ce[threadIdx.x] = s;
    }
    const float4 as4 = reinterpret_cast<const float4*>(att_src)[l], ad4 = reinterpret_cast<const float4*>(att_dst)[l];
    NodeRange r = block_nodes(n);
    for (int i = r.begin + hw; i < r.end; i += HW_PER_BLOCK) {
        float4 x = reinterpret_cast<const float4*>(xw + (size_t)i * H)[l];
        float s = head_sum(dot4(x, as4)), d = head_sum(dot4(x, ad4));
        if ((l & 7) == 0) {
            a_s[i * HEADS + h] = s;
            a_d[i * HEADS + h] = d;
        }
    }
}

// out[i] = elu( sum_e alpha_e * xw[src_e] + bias ),  alpha = softmax_e leaky_relu(a_s[src]+a_d[i]+ea*ce)
// over the edges entering i plus its self loop (taken last).  alpha is kept for the backward pass.
//
// One half-wave per node.  The kernel is latency-bound unless loads are kept in flight, and a node
// needs row_ptr -> sources -> rows, three dependent round trips, so every level issues all of its
// loads at once, unconditionally (a load under a condition compiles to a branch with a wait behind
// it; indices are clamped instead):
//   level 1  row_ptr[i], row_ptr[i+1], a_d[i] (four heads), loop_ea[i]
//   level 2  lane k reads entry k of the node's list: source and edge attribute, coalesced; the
//            self loop is entry `deg`
//   level 3  lane k reads a_s[source_k] (four heads, 16 bytes); the first four source ids are
//            broadcast and their rows fetched, 16 bytes per lane
// Scores, softmax and alpha are computed entry-per-lane for the four heads, reduced across lanes
// (two DPP steps when the list has at most four entries, the usual case for an AIG), alpha is
// stored 16 bytes per lane and handed to the row accumulation through LDS.
__global__ __launch_bounds__(THREADS) void k_gat_aggregate(
    int n, const float* __restrict__ xw, const float* __restrict__ a_s, const float* __restrict__ a_d,
    const float* __restrict__ ce, const float* __restrict__ bias, const int* __restrict__ row_ptr,
    const int* __restrict__ csr_src, const float* __restrict__ csr_ea, const float* __restrict__ loop_ea,
    float* __restrict__ out, float* __restrict__ alpha_e, float* __restrict__ alpha_loop,
    uint2* __restrict__ clear, int clear_count) {
    __shared__ float s_alpha[HW_PER_BLOCK][32][HEADS];
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5, h = l >> 3;
    // a buffer the next kernel on the stream wants zeroed (the readout's table of maxima): a few stores per block
    // here instead of a launch of its own in between
    for (int q = blockIdx.x * THREADS + threadIdx.x; q < clear_count; q += gridDim.x * THREADS) clear[q] = make_uint2(0u, 0u);
    const float4 ce4 = make_float4(ce[0], ce[1], ce[2], ce[3]);
    const float4 b4 = reinterpret_cast<const float4*>(bias)[l];
    const float4* xw4 = reinterpret_cast<const float4*>(xw);
    const float4* as4p = reinterpret_cast<const float4*>(a_s);
    const float4* ad4p = reinterpret_cast<const float4*>(a_d);
    float4* alpha_e4 = reinterpret_cast<float4*>(alpha_e);
    float4* alpha_loop4 = reinterpret_cast<float4*>(alpha_loop);
    // clamped list reads need one valid element even when the graph has no edge at all
    const bool has_edges = row_ptr[n] > 0;
    const int* src_safe = has_edges ? csr_src : row_ptr;
    const float* ea_safe = has_edges ? csr_ea : loop_ea;
    NodeRange r = block_nodes(n);
    const float NEG_INF = -__builtin_huge_valf();

    for (int i = r.begin + hw; i < r.end; i += HW_PER_BLOCK) {
        const int beg = row_ptr[i], deg = row_ptr[i + 1] - beg;
        const float4 ad4 = ad4p[i];
        const float lea = loop_ea[i];
        // entry idx of the list: edges 0..deg-1, then the self loop
        auto entry = [&](int idx, int& s, float4& e) {
            const bool is_edge = idx < deg;
            const int p = is_edge ? beg + idx : 0;
            const int sv = src_safe[p];
            const float eav = ea_safe[p];
            s = is_edge ? sv : i;
            const float ea = is_edge ? eav : lea;
            const float4 as4 = as4p[s];
            e.x = lrelu(as4.x + ad4.x + ea * ce4.x); e.y = lrelu(as4.y + ad4.y + ea * ce4.y);
            e.z = lrelu(as4.z + ad4.z + ea * ce4.z); e.w = lrelu(as4.w + ad4.w + ea * ce4.w);
            if (idx > deg) e = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
        };
        auto put_alpha = [&](int idx, float4 al) {
            if (idx < deg) alpha_e4[beg + idx] = al;
            if (idx == deg) alpha_loop4[i] = al;
            *reinterpret_cast<float4*>(&s_alpha[hw][l][0]) = al;
        };
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (deg < 32) {
            int s;
            float4 e;
            entry(l, s, e);
            const int s0 = __shfl(s, 0, 32), s1 = __shfl(s, 1, 32), s2 = __shfl(s, 2, 32), s3 = __shfl(s, 3, 32);
            const float4 r0 = xw4[(size_t)s0 * 32 + l], r1 = xw4[(size_t)s1 * 32 + l], r2 = xw4[(size_t)s2 * 32 + l],
                         r3 = xw4[(size_t)s3 * 32 + l];
            const bool wide = deg > 3;
            const float4 m = halfwave_reduce(e, wide, MaxOp());
            float4 w = make_float4(fast_exp(e.x - m.x), fast_exp(e.y - m.y), fast_exp(e.z - m.z), fast_exp(e.w - m.w));
            const float4 den = halfwave_reduce(w, wide, SumOp());
            w.x *= 1.f / (den.x + SOFTMAX_EPS); w.y *= 1.f / (den.y + SOFTMAX_EPS);
            w.z *= 1.f / (den.z + SOFTMAX_EPS); w.w *= 1.f / (den.w + SOFTMAX_EPS);
            put_alpha(l, w);
            __builtin_amdgcn_wave_barrier();
            axpy4(acc, s_alpha[hw][0][h], r0);
            axpy4(acc, s_alpha[hw][1][h], r1);
            axpy4(acc, s_alpha[hw][2][h], r2);
            axpy4(acc, s_alpha[hw][3][h], r3);
            for (int k0 = 4; k0 <= deg; k0 += 4) {  // lanes past the list carry alpha 0 and the node's own id
                const int t0 = __shfl(s, k0, 32), t1 = __shfl(s, k0 + 1, 32), t2 = __shfl(s, k0 + 2, 32),
                          t3 = __shfl(s, k0 + 3, 32);
                const float4 q0 = xw4[(size_t)t0 * 32 + l], q1 = xw4[(size_t)t1 * 32 + l],
                             q2 = xw4[(size_t)t2 * 32 + l], q3 = xw4[(size_t)t3 * 32 + l];
                axpy4(acc, s_alpha[hw][k0][h], q0);
                axpy4(acc, s_alpha[hw][k0 + 1][h], q1);
                axpy4(acc, s_alpha[hw][k0 + 2][h], q2);
                axpy4(acc, s_alpha[hw][k0 + 3][h], q3);
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            // long list: chunks of 32 entries, three sweeps (max, denominator, alpha + rows)
            const int cnt = deg + 1;
            float4 m = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
            for (int c0 = 0; c0 < cnt; c0 += 32) {
                int s;
                float4 e;
                entry(c0 + l, s, e);
                m.x = fmaxf(m.x, e.x); m.y = fmaxf(m.y, e.y); m.z = fmaxf(m.z, e.z); m.w = fmaxf(m.w, e.w);
            }
            m = halfwave_reduce(m, true, MaxOp());
            float4 den = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int c0 = 0; c0 < cnt; c0 += 32) {
                int s;
                float4 e;
                entry(c0 + l, s, e);
                den.x += fast_exp(e.x - m.x); den.y += fast_exp(e.y - m.y);
                den.z += fast_exp(e.z - m.z); den.w += fast_exp(e.w - m.w);
            }
            den = halfwave_reduce(den, true, SumOp());
            const float4 inv = make_float4(1.f / (den.x + SOFTMAX_EPS), 1.f / (den.y + SOFTMAX_EPS),
                                           1.f / (den.z + SOFTMAX_EPS), 1.f / (den.w + SOFTMAX_EPS));
            for (int c0 = 0; c0 < cnt; c0 += 32) {
                int s;
                float4 e;
                entry(c0 + l, s, e);
                put_alpha(c0 + l, make_float4(fast_exp(e.x - m.x) * inv.x, fast_exp(e.y - m.y) * inv.y,
                                              fast_exp(e.z - m.z) * inv.z, fast_exp(e.w - m.w) * inv.w));
                __builtin_amdgcn_wave_barrier();
                const int lim = min(32, cnt - c0);
                for (int k0 = 0; k0 < lim; k0 += 4) {
                    const int t0 = __shfl(s, k0, 32), t1 = __shfl(s, k0 + 1, 32), t2 = __shfl(s, k0 + 2, 32),
                              t3 = __shfl(s, k0 + 3, 32);
                    const float4 q0 = xw4[(size_t)t0 * 32 + l], q1 = xw4[(size_t)t1 * 32 + l],
                                 q2 = xw4[(size_t)t2 * 32 + l], q3 = xw4[(size_t)t3 * 32 + l];
                    axpy4(acc, s_alpha[hw][k0][h], q0);
                    axpy4(acc, s_alpha[hw][k0 + 1][h], q1);
                    axpy4(acc, s_alpha[hw][k0 + 2][h], q2);
                    axpy4(acc, s_alpha[hw][k0 + 3][h], q3);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        const float4 o = make_float4(acc.x + b4.x, acc.y + b4.y, acc.z + b4.z, acc.w + b4.w);
        reinterpret_cast<float4*>(out + (size_t)i * H)[l] = elu4(o);
    }
}

// ------------------------------------------------------------------------------------------ backward
// Target side.  g = g_out * elu'(out);  dalpha_e = <g_h, xw[src_e]_h>;  softmax and leaky-relu
// derivatives give d_raw_e (per edge and head) and d_a_d[i]; per-block partial sums of g (bias
// gradient) and of d_raw*ea (edge-coefficient gradient) go to `partial`.
// Same shape as the forward kernel: half-wave per node, list entry k (edges, then the self loop)
// lives in lane k with its four heads, loads of a level issued together with clamped indices; the
// dot products against the gathered rows are made row-per-half-wave and handed to the entry lanes
// through LDS.
// POOLED: the layer feeds the level readout directly (the critic's second layer, discriminator.py:61-91), and its
// incoming gradient is a function of the readout's 256-float gradient and a table lookup (readout_pull) - formed
// here, row by row, instead of being written as an [n,128] array by one kernel and read back by this one.
struct PoolGrad {
    const float* __restrict__ g_r;        // [256]: gradient w.r.t. the readout's output
    const int* __restrict__ node_level;   // [n]
    const float* __restrict__ w;          // [n]
    const int* __restrict__ argmax;       // [(L+1)][128]
    const int* __restrict__ meta;         // plan header or NULL
    int L;
};
template <bool POOLED>
__global__ __launch_bounds__(THREADS) void k_gat_bwd_dst(
    int n, const float* __restrict__ g_out, PoolGrad pg, const float* __restrict__ out, const float* __restrict__ xw,
    const float* __restrict__ a_s, const float* __restrict__ a_d, const float* __restrict__ ce,
    const int* __restrict__ row_ptr, const int* __restrict__ csr_src, const float* __restrict__ csr_ea,
    const float* __restrict__ loop_ea, const float* __restrict__ alpha_e, const float* __restrict__ alpha_loop,
    float* __restrict__ g_pre, float* __restrict__ d_raw_e, float* __restrict__ d_raw_loop,
    float* __restrict__ d_a_d, float* __restrict__ partial /*[grid][H + HEADS]*/) {
    __shared__ float red[HW_PER_BLOCK][H + HEADS];
    __shared__ float s_da[HW_PER_BLOCK][32][HEADS];
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5, h = l >> 3;
    const float4 ce4 = make_float4(ce[0], ce[1], ce[2], ce[3]);
    const float4* xw4 = reinterpret_cast<const float4*>(xw);
    const float4* as4p = reinterpret_cast<const float4*>(a_s);
    const float4* ad4p = reinterpret_cast<const float4*>(a_d);
    const float4* al_loop4 = reinterpret_cast<const float4*>(alpha_loop);
    float4* d_raw_e4 = reinterpret_cast<float4*>(d_raw_e);
    float4* d_raw_loop4 = reinterpret_cast<float4*>(d_raw_loop);
    float4* d_a_d4 = reinterpret_cast<float4*>(d_a_d);
    const bool has_edges = row_ptr[n] > 0;  // clamped list reads need one valid element
    const int* src_safe = has_edges ? csr_src : row_ptr;
    const float* ea_safe = has_edges ? csr_ea : loop_ea;
    const float4* al_safe = reinterpret_cast<const float4*>(has_edges ? alpha_e : alpha_loop);
    float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f), dce = gsum;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 g_mean = zero4, g_max = zero4;
    float inv_levels = 0.f;
    if (POOLED) {
        g_mean = reinterpret_cast<const float4*>(pg.g_r)[l];
        g_max = reinterpret_cast<const float4*>(pg.g_r + H)[l];
        inv_levels = 1.f / (float)(level_count(pg.L, pg.meta));
    }
    NodeRange r = block_nodes(n);
    for (int i = r.begin + hw; i < r.end; i += HW_PER_BLOCK) {
        const float4 o = reinterpret_cast<const float4*>(out + (size_t)i * H)[l];
        float4 g;
        if (POOLED)
            g = readout_pull(i, pg.node_level[i], pg.w[i], inv_levels, g_mean, g_max, reinterpret_cast<const int4*>(pg.argmax), l);
        else
            g = reinterpret_cast<const float4*>(g_out + (size_t)i * H)[l];
        const int beg = row_ptr[i], deg = row_ptr[i + 1] - beg;
        const float4 ad4 = ad4p[i];
        const float lea = loop_ea[i];
        const float4 all4 = al_loop4[i];
        g.x *= o.x > 0.f ? 1.f : o.x + 1.f; g.y *= o.y > 0.f ? 1.f : o.y + 1.f;
        g.z *= o.z > 0.f ? 1.f : o.z + 1.f; g.w *= o.w > 0.f ? 1.f : o.w + 1.f;
        reinterpret_cast<float4*>(g_pre + (size_t)i * H)[l] = g;
        gsum.x += g.x; gsum.y += g.y; gsum.z += g.z; gsum.w += g.w;
        // entry idx: edges 0..deg-1, then the self loop; lanes past the list get alpha 0
        auto entry = [&](int idx, int& s, float& ea, float4& al, float4& slope) {
            const bool is_edge = idx < deg;
            const int p = is_edge ? beg + idx : 0;
            const int sv = src_safe[p];
            const float eav = ea_safe[p];
            const float4 alv = al_safe[p];
            s = is_edge ? sv : i;
            ea = is_edge ? eav : lea;
            al = is_edge ? alv : (idx == deg ? all4 : zero4);
            const float4 as4 = as4p[s];
            slope.x = as4.x + ad4.x + ea * ce4.x > 0.f ? 1.f : NEG_SLOPE;
            slope.y = as4.y + ad4.y + ea * ce4.y > 0.f ? 1.f : NEG_SLOPE;
            slope.z = as4.z + ad4.z + ea * ce4.z > 0.f ? 1.f : NEG_SLOPE;
            slope.w = as4.w + ad4.w + ea * ce4.w > 0.f ? 1.f : NEG_SLOPE;
        };
        // dalpha of entries k0..k0+3 of the chunk held in `s`, left in s_da[hw][k0..k0+3]
        auto dots = [&](int s, int k0) {
            const int t0 = __shfl(s, k0, 32), t1 = __shfl(s, k0 + 1, 32), t2 = __shfl(s, k0 + 2, 32),
                      t3 = __shfl(s, k0 + 3, 32);
            const float4 q0 = xw4[(size_t)t0 * 32 + l], q1 = xw4[(size_t)t1 * 32 + l], q2 = xw4[(size_t)t2 * 32 + l],
                         q3 = xw4[(size_t)t3 * 32 + l];
            const float v0 = head_sum(dot4(g, q0)), v1 = head_sum(dot4(g, q1)), v2 = head_sum(dot4(g, q2)),
                        v3 = head_sum(dot4(g, q3));
            if ((l & 7) == 0) {
                s_da[hw][k0][h] = v0;
                s_da[hw][k0 + 1][h] = v1;
                s_da[hw][k0 + 2][h] = v2;
                s_da[hw][k0 + 3][h] = v3;
            }
        };
        auto put = [&](int idx, float4 dr) {
            if (idx < deg) d_raw_e4[beg + idx] = dr;
            if (idx == deg) d_raw_loop4[i] = dr;
        };
        if (deg < 32) {
            int s;
            float ea;
            float4 al, slope;
            entry(l, s, ea, al, slope);
            dots(s, 0);
            for (int k0 = 4; k0 <= deg; k0 += 4) dots(s, k0);
            __builtin_amdgcn_wave_barrier();
            float4 da = *reinterpret_cast<const float4*>(&s_da[hw][l][0]);
            if (l > deg) da = zero4;  // never written for this node
            __builtin_amdgcn_wave_barrier();
            const bool wide = deg > 3;
            const float4 sdot = halfwave_reduce(make_float4(al.x * da.x, al.y * da.y, al.z * da.z, al.w * da.w), wide, SumOp());
            const float4 dr = make_float4(al.x * (da.x - sdot.x) * slope.x, al.y * (da.y - sdot.y) * slope.y,
                                          al.z * (da.z - sdot.z) * slope.z, al.w * (da.w - sdot.w) * slope.w);
            const float4 dad = halfwave_reduce(dr, wide, SumOp());
            put(l, dr);
            if (l == 0) d_a_d4[i] = dad;
            dce.x += dr.x * ea; dce.y += dr.y * ea; dce.z += dr.z * ea; dce.w += dr.w * ea;
        } else {
            // long list: chunks of 32 entries, two sweeps (sdot, then d_raw)
            const int cnt = deg + 1;
            float4 sdot = zero4;
            for (int c0 = 0; c0 < cnt; c0 += 32) {
                int s;
                float ea;
                float4 al, slope;
                entry(c0 + l, s, ea, al, slope);
                const int lim = min(32, cnt - c0);
                for (int k0 = 0; k0 < lim; k0 += 4) dots(s, k0);
                __builtin_amdgcn_wave_barrier();
                float4 da = *reinterpret_cast<const float4*>(&s_da[hw][l][0]);
                if (l >= lim) da = zero4;
                __builtin_amdgcn_wave_barrier();
                sdot.x += al.x * da.x; sdot.y += al.y * da.y; sdot.z += al.z * da.z; sdot.w += al.w * da.w;
            }
            sdot = halfwave_reduce(sdot, true, SumOp());
            float4 dad = zero4;
            for (int c0 = 0; c0 < cnt; c0 += 32) {
                int s;
                float ea;
                float4 al, slope;
                entry(c0 + l, s, ea, al, slope);
                const int lim = min(32, cnt - c0);
                for (int k0 = 0; k0 < lim; k0 += 4) dots(s, k0);
                __builtin_amdgcn_wave_barrier();
                float4 da = *reinterpret_cast<const float4*>(&s_da[hw][l][0]);
                if (l >= lim) da = zero4;
                __builtin_amdgcn_wave_barrier();
                const float4 dr = make_float4(al.x * (da.x - sdot.x) * slope.x, al.y * (da.y - sdot.y) * slope.y,
                                              al.z * (da.z - sdot.z) * slope.z, al.w * (da.w - sdot.w) * slope.w);
                put(c0 + l, dr);
                dad.x += dr.x; dad.y += dr.y; dad.z += dr.z; dad.w += dr.w;
                dce.x += dr.x * ea; dce.y += dr.y * ea; dce.z += dr.z * ea; dce.w += dr.w * ea;
            }
            dad = halfwave_reduce(dad, true, SumOp());
            if (l == 0) d_a_d4[i] = dad;
        }
    }
    // block partials, summed over the 8 half-waves in fixed order
    dce = halfwave_reduce(dce, true, SumOp());
    reinterpret_cast<float4*>(&red[hw][0])[l] = gsum;
    if (l == 0) *reinterpret_cast<float4*>(&red[hw][H]) = dce;
    __syncthreads();
    if (threadIdx.x < H + HEADS) {
        float s = 0.f;
        for (int k = 0; k < HW_PER_BLOCK; ++k) s += red[k][threadIdx.x];
        partial[(size_t)blockIdx.x * (H + HEADS) + threadIdx.x] = s;
    }
}

// Source side.  d_xw[j] = sum_{e: src=j} alpha_e g_pre[dst_e] + alpha_loop g_pre[j]
//                        + d_a_s[j] att_src + d_a_d[j] att_dst,   d_a_s[j] = sum d_raw over out-edges.
// Fan-out is heavily skewed: a generated graph hangs most of its gates on two or three sources (tens of
// thousands of out-edges each) while every other node has a handful.  So a half-wave owns one NODE when
// its out-edge list has at most SEG edges, and the lists of the few "hubs" (listed in hub_tab) are cut
// into chunks of HUB_CHUNK edges that whole blocks gather, one partial row per chunk;
// k_gat_bwd_hub_finish adds the rows of a hub in chunk order and finishes the node.  Every sum has a
// fixed shape, so results are reproducible whatever the fan-out.
struct SrcFinish {
    float4 as4, ad4;
};
struct SrcNode {  // what finishing node j needs besides the accumulated row
    float drl, al, dad;
    float4 gj, xj;
};
__device__ __forceinline__ SrcNode src_node_load(int j, int l, int h, const float4* __restrict__ g4,
                                                 const float* __restrict__ xw, const float* __restrict__ alpha_loop,
                                                 const float* __restrict__ d_raw_loop,
                                                 const float* __restrict__ d_a_d) {
    SrcNode v;
    v.drl = d_raw_loop[j * HEADS + h];
    v.al = alpha_loop[j * HEADS + h];
    v.dad = d_a_d[j * HEADS + h];
    v.gj = g4[(size_t)j * 32 + l];
    v.xj = reinterpret_cast<const float4*>(xw + (size_t)j * H)[l];
    return v;
}
__device__ __forceinline__ void src_finish(int j, int l, float4 acc, float das, const SrcFinish& f, const SrcNode& v,
                                           float* __restrict__ d_xw, float4& ps, float4& pd) {
    das += v.drl;
    acc.x += v.al * v.gj.x + das * f.as4.x + v.dad * f.ad4.x;
    acc.y += v.al * v.gj.y + das * f.as4.y + v.dad * f.ad4.y;
    acc.z += v.al * v.gj.z + das * f.as4.z + v.dad * f.ad4.z;
    acc.w += v.al * v.gj.w + das * f.as4.w + v.dad * f.ad4.w;
    reinterpret_cast<float4*>(d_xw + (size_t)j * H)[l] = acc;
    ps.x += das * v.xj.x; ps.y += das * v.xj.y; ps.z += das * v.xj.z; ps.w += das * v.xj.w;
    pd.x += v.dad * v.xj.x; pd.y += v.dad * v.xj.y; pd.z += v.dad * v.xj.z; pd.w += v.dad * v.xj.w;
}

// row `row` of partial[][2H] = the block's sums of ps | pd, half-waves added in fixed order
__device__ __forceinline__ void block_partial_2h(float4 ps, float4 pd, float* __restrict__ partial, int row) {
    __shared__ float red[HW_PER_BLOCK][2 * H];
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    reinterpret_cast<float4*>(&red[hw][0])[l] = ps;
    reinterpret_cast<float4*>(&red[hw][H])[l] = pd;
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < HW_PER_BLOCK; ++k) s += red[k][threadIdx.x];
    partial[(size_t)row * (2 * H) + threadIdx.x] = s;
}

__device__ __forceinline__ float pick(float4 v, int h) { return h == 0 ? v.x : h == 1 ? v.y : h == 2 ? v.z : v.w; }

struct SrcEdges {
    const int* __restrict__ csc_pos;
    const int* __restrict__ csc_dst;
    const float4* __restrict__ alpha_e4;
    const float4* __restrict__ d_raw_e4;
    const float4* __restrict__ g4;
};
// One half-wave, cnt (1..32) edges at positions q0.. of the by-source list: lane k takes edge k (its
// position in the by-target order and its target), alpha goes through LDS, target rows come four at a
// time.  acc += sum alpha_e g_pre[dst_e] (this lane's four columns), dr += d_raw of this lane's edge.
__device__ __forceinline__ void src_edges32(const SrcEdges& a, int q0, int cnt, int l, int h, float (*s_al)[HEADS],
                                            float4& acc, float4& dr) {
    const bool on = l < cnt;
    const int q = on ? q0 + l : q0;
    const int p = a.csc_pos[q], dv = a.csc_dst[q];
    const int dst = dv;   // lanes past the list read the first edge again: its target, with alpha 0
    const float4 alv = a.alpha_e4[p], drv = a.d_raw_e4[p];
    const float4 al = on ? alv : make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) { dr.x += drv.x; dr.y += drv.y; dr.z += drv.z; dr.w += drv.w; }
    *reinterpret_cast<float4*>(&s_al[l][0]) = al;
    __builtin_amdgcn_wave_barrier();
    for (int k0 = 0; k0 < cnt; k0 += 4) {
        const int t0 = __shfl(dst, k0, 32), t1 = __shfl(dst, k0 + 1, 32), t2 = __shfl(dst, k0 + 2, 32),
                  t3 = __shfl(dst, k0 + 3, 32);
        const float4 r0 = a.g4[(size_t)t0 * 32 + l], r1 = a.g4[(size_t)t1 * 32 + l], r2 = a.g4[(size_t)t2 * 32 + l],
                     r3 = a.g4[(size_t)t3 * 32 + l];
        axpy4(acc, s_al[k0][h], r0);
        axpy4(acc, s_al[k0 + 1][h], r1);
        axpy4(acc, s_al[k0 + 2][h], r2);
        axpy4(acc, s_al[k0 + 3][h], r3);
    }
    __builtin_amdgcn_wave_barrier();
}

// Blocks 0..hub_blocks-1 gather the chunks of the hubs (they are dispatched first, their work is the
// longest), the other blocks walk the nodes, one half-wave per node, XCD-grouped.
__global__ __launch_bounds__(THREADS) void k_gat_bwd_src(
    int n, int hub_blocks, const float* __restrict__ g_pre, const float* __restrict__ xw,
    const float* __restrict__ att_src, const float* __restrict__ att_dst, const int4* __restrict__ hub_tab,
    const int* __restrict__ col_ptr, const int* __restrict__ csc_pos, const int* __restrict__ csc_dst,
    const float* __restrict__ alpha_e, const float* __restrict__ alpha_loop, const float* __restrict__ d_raw_e,
    const float* __restrict__ d_raw_loop, const float* __restrict__ d_a_d, float* __restrict__ d_xw,
    float* __restrict__ part_acc /*[chunks][H]*/, float* __restrict__ part_das /*[chunks][HEADS]*/,
    float* __restrict__ partial /*[node blocks][2H]*/) {
    __shared__ float s_alpha[HW_PER_BLOCK][32][HEADS];
    __shared__ float s_acc[HW_PER_BLOCK][H];
    __shared__ float s_das[HW_PER_BLOCK][HEADS];
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5, h = l >> 3;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const SrcEdges ed{csc_pos, csc_dst, reinterpret_cast<const float4*>(alpha_e),
                      reinterpret_cast<const float4*>(d_raw_e), reinterpret_cast<const float4*>(g_pre)};
    if ((int)blockIdx.x < hub_blocks) {
        const int4 hdr = hub_tab[0];
        const int n_hub = hdr.x, total = hdr.y;
        for (int c = blockIdx.x; c < total; c += hub_blocks) {
            int lo = 0, hi = n_hub;  // the hub owning chunk c: the last one whose first chunk is <= c
            while (lo + 1 < hi) {
                const int mid = (lo + hi) >> 1;
                if (hub_tab[1 + mid].w <= c) lo = mid; else hi = mid;
            }
            const int4 t = hub_tab[1 + lo];
            const int e0 = t.y + (c - t.w) * HUB_CHUNK, e1 = min(t.y + t.z, e0 + HUB_CHUNK);
            const int q0 = e0 + hw * 32, cnt = max(0, min(32, e1 - q0));
            float4 acc = zero4, dr = zero4;
            if (cnt > 0) src_edges32(ed, q0, cnt, l, h, s_alpha[hw], acc, dr);
            const float4 das4 = halfwave_reduce(dr, true, SumOp());
            reinterpret_cast<float4*>(&s_acc[hw][0])[l] = acc;
            if (l == 0) *reinterpret_cast<float4*>(&s_das[hw][0]) = das4;
            __syncthreads();
            if (threadIdx.x < H + HEADS) {
                const bool row = threadIdx.x < H;
                const int col = row ? threadIdx.x : threadIdx.x - H;
                float sum = 0.f;
                for (int k = 0; k < HW_PER_BLOCK; ++k) sum += row ? s_acc[k][col] : s_das[k][col];
                if (row) part_acc[(size_t)c * H + col] = sum;
                else part_das[(size_t)c * HEADS + col] = sum;
            }
            __syncthreads();
        }
        return;
    }
    const int b = blockIdx.x - hub_blocks;
    SrcFinish f{reinterpret_cast<const float4*>(att_src)[l], reinterpret_cast<const float4*>(att_dst)[l]};
    float4 ps = zero4, pd = zero4;
    NodeRange r = block_nodes_of(n, b, gridDim.x - hub_blocks);
    for (int j = r.begin + hw; j < r.end; j += HW_PER_BLOCK) {
        const int beg = col_ptr[j], cnt = col_ptr[j + 1] - beg;
        if (cnt > SEG) continue;  // a hub
        const SrcNode v = src_node_load(j, l, h, ed.g4, xw, alpha_loop, d_raw_loop, d_a_d);
        float4 acc = zero4, dr = zero4;
        if (cnt > 0) src_edges32(ed, beg, cnt, l, h, s_alpha[hw], acc, dr);
        const float4 das4 = halfwave_reduce(dr, cnt > 4, SumOp());
        src_finish(j, l, acc, pick(das4, h), f, v, d_xw, ps, pd);
    }
    block_partial_2h(ps, pd, partial, b);
}

// One block per hub at a time: half-wave hw adds the chunk rows hw, hw+8, ... of the hub (four loads in
// flight), the eight sums are combined in half-wave order and half-wave 0 finishes the node.
__global__ __launch_bounds__(THREADS) void k_gat_bwd_hub_finish(
    const float* __restrict__ g_pre, const float* __restrict__ xw, const float* __restrict__ att_src,
    const float* __restrict__ att_dst, const int4* __restrict__ hub_tab, const float* __restrict__ alpha_loop,
    const float* __restrict__ d_raw_loop, const float* __restrict__ d_a_d, const float* __restrict__ part_acc,
    const float* __restrict__ part_das, float* __restrict__ d_xw, float* __restrict__ partial /*[grid][2H]*/) {
    __shared__ float s_acc[HW_PER_BLOCK][H];
    __shared__ float s_das[HW_PER_BLOCK][HEADS];
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5, h = l >> 3;
    SrcFinish f{reinterpret_cast<const float4*>(att_src)[l], reinterpret_cast<const float4*>(att_dst)[l]};
    const float4* g4 = reinterpret_cast<const float4*>(g_pre);
    float4 ps = make_float4(0.f, 0.f, 0.f, 0.f), pd = ps;
    const int4 hdr = hub_tab[0];
    const int n_hub = hdr.x, total = hdr.y;
    for (int k = blockIdx.x; k < n_hub; k += gridDim.x) {
        const int4 t = hub_tab[1 + k];
        const int c0 = t.w, c1 = k + 1 < n_hub ? hub_tab[2 + k].w : total;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float das = 0.f;
        constexpr int INFL = 16;  // rows in flight per half-wave (added in row order whatever this is)
        for (int c = c0 + hw; c < c1; c += INFL * HW_PER_BLOCK) {
            float4 a[INFL];
            float dd[INFL];
#pragma unroll
            for (int u = 0; u < INFL; ++u) {
                const size_t row = (size_t)min(c + u * HW_PER_BLOCK, c1 - 1);
                a[u] = reinterpret_cast<const float4*>(part_acc + row * H)[l];
                dd[u] = part_das[row * HEADS + h];
            }
#pragma unroll
            for (int u = 0; u < INFL; ++u)
                if (c + u * HW_PER_BLOCK < c1) {
                    acc.x += a[u].x; acc.y += a[u].y; acc.z += a[u].z; acc.w += a[u].w;
                    das += dd[u];
                }
        }
        reinterpret_cast<float4*>(&s_acc[hw][0])[l] = acc;
        if ((l & 7) == 0) s_das[hw][h] = das;
        __syncthreads();
        if (hw == 0) {
            const int j = t.x;
            const SrcNode v = src_node_load(j, l, h, g4, xw, alpha_loop, d_raw_loop, d_a_d);
            float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
            float dt = 0.f;
            for (int q = 0; q < HW_PER_BLOCK; ++q) {
                const float4 a = reinterpret_cast<const float4*>(&s_acc[q][0])[l];
                tot.x += a.x; tot.y += a.y; tot.z += a.z; tot.w += a.w;
                dt += s_das[q][h];
            }
            src_finish(j, l, tot, dt, f, v, d_xw, ps, pd);
        }
        __syncthreads();
    }
    block_partial_2h(ps, pd, partial, blockIdx.x);
}

}  // namespace
}  // namespace aig

using namespace aig;

extern "C" size_t aig_gat_csr_ws_bytes(int64_t n, int64_t E) {
    if (E <= 0) return 256;
    size_t t = sort_temp_bytes(E, key_bits(n)), t2 = scan_temp_bytes(n + 1);
    return 4 * align_up((size_t)E * 4) + 4 * align_up((size_t)(n + 1) * 4) + align_up(t > t2 ? t : t2) + 256;
}

extern "C" int aig_gat_csr(int64_t n, int64_t E, const int64_t* edge_index, const float* edge_attr,
                           int32_t* row_ptr, int32_t* csr_src, float* csr_ea, int32_t* csr_eid, float* loop_ea,
                           int32_t* col_ptr, int32_t* csc_pos, int32_t* csc_dst, int32_t* hub_tab, void* ws,
                           size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX && E >= 0 && E < INT_MAX, "aig_gat_csr: bad sizes n=%lld E=%lld", (long long)n,
                (long long)E);
    AIG_REQUIRE(row_ptr && col_ptr && hub_tab && (n == 0 || loop_ea), "aig_gat_csr: null output");
    int4* tab = reinterpret_cast<int4*>(hub_tab);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (E == 0) {
        AIG_HIP_CHECK(fill_words(row_ptr, 0u, (size_t)(n + 1) * 4, s));
        AIG_HIP_CHECK(fill_words(col_ptr, 0u, (size_t)(n + 1) * 4, s));
        if (n) AIG_HIP_CHECK(fill_words(loop_ea, 0u, (size_t)n * 4, s));
        AIG_HIP_CHECK(fill_words(tab, 0u, sizeof(int4), s));   // no hubs
        return AIG_OK;
    }
    AIG_REQUIRE(edge_index && csr_src && csr_ea && csr_eid && csc_pos && csc_dst && ws, "aig_gat_csr: null pointer");
    if (ws_bytes < aig_gat_csr_ws_bytes(n, E))
        return fail(AIG_EWORKSPACE, "aig_gat_csr: workspace %zu < %zu", ws_bytes, aig_gat_csr_ws_bytes(n, E));
    const int bits = key_bits(n);
    Carver c(ws);
    unsigned* key = c.take<unsigned>(E);
    unsigned* skey = c.take<unsigned>(E);
    int* ids = c.take<int>(E);
    int* seid = c.take<int>(E);
    int* flag = c.take<int>(n + 1);
    int* chunks = c.take<int>(n + 1);
    int* rank = c.take<int>(n + 1);
    int* cbase = c.take<int>(n + 1);
    size_t tb = sort_temp_bytes(E, bits), tb2 = scan_temp_bytes(n + 1);
    if (tb2 > tb) tb = tb2;
    void* temp = c.take<char>(tb);
    const int g = grid_for(E, 256, 4096), gn = grid_for(n + 1, 256, 4096);
    const long long* ei = reinterpret_cast<const long long*>(edge_index);
    hipLaunchKernelGGL(k_csr_keys, dim3(g), dim3(256), 0, s, (int)n, (int)E, ei, key, ids);
    AIG_LAUNCH_CHECK();
    AIG_HIP_CHECK(rs::sort_pairs(temp, tb, key, skey, ids, seid, (size_t)E, bits, s));
    hipLaunchKernelGGL(k_csr_ptr, dim3(gn), dim3(256), 0, s, (int)n, (int)E, skey, row_ptr);
    // key <- source of each kept edge (in target-sorted order), ids <- that position
    hipLaunchKernelGGL(k_csr_fill, dim3(g), dim3(256), 0, s, (int)n, (int)E, ei, edge_attr, skey, seid, csr_src, csr_ea,
                       csr_eid, key, ids);
    AIG_LAUNCH_CHECK();
    // seid is free again: reuse it for the sorted source keys
    unsigned* ssrc = reinterpret_cast<unsigned*>(seid);
    AIG_HIP_CHECK(rs::sort_pairs(temp, tb, key, ssrc, ids, csc_pos, (size_t)E, bits, s));
    hipLaunchKernelGGL(k_csr_ptr, dim3(gn), dim3(256), 0, s, (int)n, (int)E, ssrc, col_ptr);
    hipLaunchKernelGGL(k_csc_fill, dim3(g), dim3(256), 0, s, (int)E, row_ptr + n, skey, csc_pos, csc_dst);
    hipLaunchKernelGGL(k_loop_attr, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, (int)n, row_ptr, csr_ea, loop_ea);
    hipLaunchKernelGGL(k_hub_count, dim3(gn), dim3(256), 0, s, (int)n, col_ptr, flag, chunks);
    AIG_LAUNCH_CHECK();
    AIG_HIP_CHECK(rocprim::exclusive_scan(temp, tb, flag, rank, 0, (size_t)(n + 1), rocprim::plus<int>(), s));
    AIG_HIP_CHECK(rocprim::exclusive_scan(temp, tb, chunks, cbase, 0, (size_t)(n + 1), rocprim::plus<int>(), s));
    hipLaunchKernelGGL(k_hub_tab, dim3(gn), dim3(256), 0, s, (int)n, col_ptr, rank, cbase, tab);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

extern "C" int aig_gat_node_alpha(int64_t n, const float* xw, const float* att_src, const float* att_dst,
                                  const float* lin_edge_w, const float* att_edge, float* a_s, float* a_d, float* ce,
                                  void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX, "aig_gat_node_alpha: n=%lld out of range", (long long)n);
    if (n == 0) return AIG_OK;
    AIG_REQUIRE(xw && att_src && att_dst && a_s && a_d, "aig_gat_node_alpha: null pointer");
    AIG_REQUIRE(!ce || (lin_edge_w && att_edge), "aig_gat_node_alpha: ce needs lin_edge_w and att_edge");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_gat_node_alpha, dim3(grid_blocks(n)), dim3(THREADS), 0, s, (int)n, xw, att_src, att_dst,
                       lin_edge_w, att_edge, a_s, a_d, ce);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

extern "C" int aig_gat_fwd_clearing(int64_t n, const float* xw, const float* a_s, const float* a_d, const float* ce,
                                    const float* bias, const int32_t* row_ptr, const int32_t* csr_src, const float* csr_ea,
                                    const float* loop_ea, float* out, float* alpha_e, float* alpha_loop, void* clear,
                                    size_t clear_bytes, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX, "aig_gat_fwd: n=%lld out of range", (long long)n);
    AIG_REQUIRE(clear_bytes == 0 || (clear && (reinterpret_cast<uintptr_t>(clear) & 7) == 0 && clear_bytes % 8 == 0 &&
                                     clear_bytes / 8 < (size_t)INT_MAX),
                "aig_gat_fwd_clearing: the buffer to clear must be 8-byte aligned and a multiple of 8 bytes long");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) {
        if (clear_bytes) AIG_HIP_CHECK(fill_words(clear, 0u, clear_bytes, s));
        return AIG_OK;
    }
    AIG_REQUIRE(xw && a_s && a_d && ce && bias && row_ptr && loop_ea && out && alpha_loop, "aig_gat_fwd: null pointer");
    AIG_LAUNCH_TIMED(AIG_KT_GAT_AGGREGATE, k_gat_aggregate, dim3(grid_blocks(n)), dim3(THREADS), 0, s, (int)n, xw, a_s, a_d,
                     ce, bias, row_ptr, csr_src, csr_ea, loop_ea, out, alpha_e, alpha_loop, static_cast<uint2*>(clear),
                     (int)(clear_bytes / 8));
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

extern "C" int aig_gat_fwd(int64_t n, const float* xw, const float* a_s, const float* a_d, const float* ce,
                           const float* bias, const int32_t* row_ptr, const int32_t* csr_src, const float* csr_ea,
                           const float* loop_ea, float* out, float* alpha_e, float* alpha_loop, void* stream) {
    return aig_gat_fwd_clearing(n, xw, a_s, a_d, ce, bias, row_ptr, csr_src, csr_ea, loop_ea, out, alpha_e, alpha_loop, nullptr,
                                0, stream);
}

extern "C" size_t aig_gat_bwd_ws_bytes(int64_t n, int64_t E) {
    size_t N = (size_t)(n > 0 ? n : 1), M = (size_t)(E > 0 ? E : 1), chunks = hub_chunks_max(E);
    return align_up(N * H * 4) /*g_pre*/ + align_up(M * HEADS * 4) + 2 * align_up(N * HEADS * 4) +
           align_up(chunks * H * 4) + align_up(chunks * HEADS * 4) + align_up((size_t)4097 * 2 * H * 4) +
           align_up((size_t)2048 * (H + HEADS) * 4) + 256;
}

namespace {
int gat_bwd(int64_t n, int64_t E, const float* g_out, const PoolGrad* pooled, const float* out, const float* xw,
                           const float* att_src, const float* att_dst, const float* ce, const float* a_s,
                           const float* a_d, const int32_t* row_ptr, const int32_t* csr_src, const float* csr_ea,
                           const float* loop_ea, const int32_t* col_ptr, const int32_t* csc_pos,
                           const int32_t* csc_dst, const int32_t* hub_tab,
                           const float* alpha_e,
                           const float* alpha_loop, const float* lin_edge_w, const float* att_edge, float* d_xw,
                           float* d_att_src, float* d_att_dst, float* d_ce, float* d_bias, float* d_lin_edge_w,
                           float* d_att_edge, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX && E >= 0, "aig_gat_bwd: bad sizes");
    AIG_REQUIRE(d_att_src && d_att_dst && d_ce && d_bias, "aig_gat_bwd: null output");
    AIG_REQUIRE((d_lin_edge_w == nullptr) == (d_att_edge == nullptr) && (!d_lin_edge_w || (lin_edge_w && att_edge)),
                "aig_gat_bwd: d_lin_edge_w and d_att_edge come together and need lin_edge_w and att_edge");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) {
        AIG_HIP_CHECK(fill_words(d_att_src, 0u, H * 4, s));
        AIG_HIP_CHECK(fill_words(d_att_dst, 0u, H * 4, s));
        AIG_HIP_CHECK(fill_words(d_ce, 0u, HEADS * 4, s));
        AIG_HIP_CHECK(fill_words(d_bias, 0u, H * 4, s));
        if (d_lin_edge_w) {
            AIG_HIP_CHECK(fill_words(d_lin_edge_w, 0u, H * 4, s));
            AIG_HIP_CHECK(fill_words(d_att_edge, 0u, H * 4, s));
        }
        return AIG_OK;
    }
    AIG_REQUIRE((g_out || pooled) && out && xw && att_src && att_dst && ce && a_s && a_d && row_ptr && loop_ea && col_ptr &&
                    hub_tab && alpha_loop && d_xw && ws,
                "aig_gat_bwd: null pointer");
    AIG_REQUIRE(!pooled || (pooled->g_r && pooled->node_level && pooled->w && pooled->argmax && pooled->L >= 0),
                "aig_gat_bwd_pooled: null pointer or negative level count");
    if (ws_bytes < aig_gat_bwd_ws_bytes(n, E))
        return fail(AIG_EWORKSPACE, "aig_gat_bwd: workspace %zu < %zu", ws_bytes, aig_gat_bwd_ws_bytes(n, E));
    Carver c(ws);
    float* g_pre = c.take<float>((size_t)n * H);
    float* d_raw_e = c.take<float>((size_t)(E > 0 ? E : 1) * HEADS);
    float* d_raw_loop = c.take<float>((size_t)n * HEADS);
    float* d_a_d = c.take<float>((size_t)n * HEADS);
    float* part_acc = c.take<float>(hub_chunks_max(E) * H);
    float* part_das = c.take<float>(hub_chunks_max(E) * HEADS);
    float* partial = c.take<float>((size_t)4097 * 2 * H);       // source-side kernels: [gs + g][2H]
    float* partial_dst = c.take<float>((size_t)2048 * (H + HEADS));  // target-side kernel: [g][H + HEADS]
    // the target-side kernel holds 4 waves per SIMD (104 registers): 1024 blocks are resident at once, and a grid
    // of 1568 left the second round half empty
    const int g = grid_blocks(n, BWD_DST_BLOCKS);
    if (pooled)
        AIG_LAUNCH_TIMED(AIG_KT_GAT_BWD_DST, k_gat_bwd_dst<true>, dim3(g), dim3(THREADS), 0, s, (int)n, g_out, *pooled, out, xw,
                         a_s, a_d, ce, row_ptr, csr_src, csr_ea, loop_ea, alpha_e, alpha_loop, g_pre, d_raw_e, d_raw_loop,
                         d_a_d, partial_dst);
    else
        AIG_LAUNCH_TIMED(AIG_KT_GAT_BWD_DST, k_gat_bwd_dst<false>, dim3(g), dim3(THREADS), 0, s, (int)n, g_out, PoolGrad{}, out,
                         xw, a_s, a_d, ce, row_ptr, csr_src, csr_ea, loop_ea, alpha_e, alpha_loop, g_pre, d_raw_e,
                         d_raw_loop, d_a_d, partial_dst);
    AIG_LAUNCH_CHECK();
    // blocks of the by-source kernel: HUB_BLOCKS for the hubs' chunks (they leave at once when the graph
    // has none), then gs over the nodes; the finishing kernel takes a hub per block
    const int gs = grid_blocks(n, BWD_SRC_BLOCKS);
    const int4* tab = reinterpret_cast<const int4*>(hub_tab);
    AIG_LAUNCH_TIMED(AIG_KT_GAT_BWD_SRC, k_gat_bwd_src, dim3(HUB_BLOCKS + gs), dim3(THREADS), 0, s, (int)n, HUB_BLOCKS, g_pre,
                     xw, att_src, att_dst, tab, col_ptr, csc_pos, csc_dst, alpha_e, alpha_loop, d_raw_e, d_raw_loop, d_a_d,
                     d_xw, part_acc, part_das, partial);
    hipLaunchKernelGGL(k_gat_bwd_hub_finish, dim3(HUB_FINISH_BLOCKS), dim3(THREADS), 0, s, g_pre, xw, att_src, att_dst, tab,
                       alpha_loop, d_raw_loop, d_a_d, part_acc, part_das, d_xw, partial + (size_t)gs * 2 * H);
    AIG_LAUNCH_CHECK();
    // bias gradient = first H columns of the target-side partials, edge-coefficient gradient = their
    // last HEADS columns; the attention vectors' gradients = the two halves of the source-side ones
    ColSums sums{};
    sums.j[0] = ColSum{partial_dst, d_bias, g, H + HEADS, H};
    // d ce[h] also gives d lin_edge.weight[h,:] = d ce[h] * att_edge[h,:] and d att_edge[h,:] = d ce[h] * lin_edge.weight[h,:]
    sums.j[1] = ColSum{partial_dst + H, d_ce, g, H + HEADS, HEADS, d_lin_edge_w ? att_edge : nullptr, d_lin_edge_w,
                       lin_edge_w, d_att_edge, H / HEADS};
    sums.j[2] = ColSum{partial, d_att_src, gs + HUB_FINISH_BLOCKS, 2 * H, H};
    sums.j[3] = ColSum{partial + H, d_att_dst, gs + HUB_FINISH_BLOCKS, 2 * H, H};
    hipLaunchKernelGGL(k_reduce_cols, dim3(3 * H + HEADS), dim3(256), 0, s, sums);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}
}  // namespace

extern "C" int aig_gat_bwd(int64_t n, int64_t E, const float* g_out, const float* out, const float* xw,
                           const float* att_src, const float* att_dst, const float* ce, const float* a_s,
                           const float* a_d, const int32_t* row_ptr, const int32_t* csr_src, const float* csr_ea,
                           const float* loop_ea, const int32_t* col_ptr, const int32_t* csc_pos,
                           const int32_t* csc_dst, const int32_t* hub_tab, const float* alpha_e,
                           const float* alpha_loop, const float* lin_edge_w, const float* att_edge, float* d_xw,
                           float* d_att_src, float* d_att_dst, float* d_ce, float* d_bias, float* d_lin_edge_w,
                           float* d_att_edge, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n == 0 || g_out, "aig_gat_bwd: null g_out");
    return gat_bwd(n, E, g_out, nullptr, out, xw, att_src, att_dst, ce, a_s, a_d, row_ptr, csr_src, csr_ea, loop_ea, col_ptr,
                   csc_pos, csc_dst, hub_tab, alpha_e, alpha_loop, lin_edge_w, att_edge, d_xw, d_att_src, d_att_dst, d_ce,
                   d_bias, d_lin_edge_w, d_att_edge, ws, ws_bytes, stream);
}

extern "C" int aig_gat_bwd_pooled(int64_t n, int64_t E, int64_t L, const int32_t* lvl_meta, const float* g_r,
                                  const int32_t* node_level, const float* w, const int32_t* argmax, const float* out,
                                  const float* xw, const float* att_src, const float* att_dst, const float* ce,
                                  const float* a_s, const float* a_d, const int32_t* row_ptr, const int32_t* csr_src,
                                  const float* csr_ea, const float* loop_ea, const int32_t* col_ptr,
                                  const int32_t* csc_pos, const int32_t* csc_dst, const int32_t* hub_tab,
                                  const float* alpha_e, const float* alpha_loop, const float* lin_edge_w,
                                  const float* att_edge, float* d_xw, float* d_att_src, float* d_att_dst, float* d_ce,
                                  float* d_bias, float* d_lin_edge_w, float* d_att_edge, void* ws, size_t ws_bytes,
                                  void* stream) {
    AIG_REQUIRE(L >= 0 && L < INT_MAX, "aig_gat_bwd_pooled: bad level count");
    const PoolGrad pg{g_r, node_level, w, argmax, lvl_meta, (int)L};
    return gat_bwd(n, E, nullptr, &pg, out, xw, att_src, att_dst, ce, a_s, a_d, row_ptr, csr_src, csr_ea, loop_ea, col_ptr,
                   csc_pos, csc_dst, hub_tab, alpha_e, alpha_loop, lin_edge_w, att_edge, d_xw, d_att_src, d_att_dst, d_ce,
                   d_bias, d_lin_edge_w, d_att_edge, ws, ws_bytes, stream);
}
