// Device helpers shared by the attention-layer kernels (gat.hip, gat1.hip) and the readout's backward (gp.hip):
// block -> node mapping, half-wave reductions, the column-sum kernel behind the parameter gradients.
// Everything sits in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include <cstdlib>

#include "common.h"

namespace aig {
namespace {

constexpr int GAT_H = AIG_HIDDEN;  // heads * channels = 128
constexpr int GAT_HEADS = 4;
constexpr int GAT_THREADS = 256;
constexpr int GAT_HW_PER_BLOCK = GAT_THREADS / 32;  // half-waves (one target each) per block
constexpr float NEG_SLOPE = 0.2f;
constexpr float SOFTMAX_EPS = 1e-16f;

// ------------------------------------------------------------------------------------------ block -> node range
// Blocks b, b+8, b+16, ... share an XCD (and its L2) under round-robin dispatch, so logical block
// ids are dealt such that one XCD walks one contiguous eighth of the nodes.  Affinity only.
__device__ __forceinline__ int logical_block(int b, int nb) { return (b & 7) * (nb >> 3) + (b >> 3); }

struct NodeRange {
    int begin, end;
};
__device__ __forceinline__ NodeRange block_nodes_of(int n, int block, int nb /* multiple of 8 */) {
    int lb = logical_block(block, nb);
    int chunk = (n + nb - 1) / nb;
    int b = lb * chunk, e = min(n, b + chunk);
    return {b, e};
}
__device__ __forceinline__ NodeRange block_nodes(int n) { return block_nodes_of(n, blockIdx.x, gridDim.x); }

// value of the lane selected by a DPP control word within the row of 16
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E;  // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;           // lane j <-> 7-j within each group of 8

__device__ __forceinline__ float head_sum(float v) {  // sum over the 8 lanes of one head, result in all 8
    v += dpp<DPP_XOR1>(v);
    v += dpp<DPP_XOR2>(v);
    v += dpp<DPP_HALF_MIRROR>(v);  // quads are uniform by now, the mirror lane sits in the other quad
    return v;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float lrelu(float x) { return x > 0.f ? x : NEG_SLOPE * x; }

// exp through the hardware exp2 (v_exp_f32, ~1 ulp of the scaled argument): the accurate library
// routines cost tens of instructions per call and made this bandwidth kernel ALU-bound.
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }

__device__ __forceinline__ float4 elu4(float4 o) {
    o.x = o.x > 0.f ? o.x : fast_exp(o.x) - 1.f; o.y = o.y > 0.f ? o.y : fast_exp(o.y) - 1.f;
    o.z = o.z > 0.f ? o.z : fast_exp(o.z) - 1.f; o.w = o.w > 0.f ? o.w : fast_exp(o.w) - 1.f;
    return o;
}
__device__ __forceinline__ void axpy4(float4& acc, float w, float4 r) {
    acc.x += w * r.x; acc.y += w * r.y; acc.z += w * r.z; acc.w += w * r.w;
}

struct MaxOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); }
};
struct SumOp {
    __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};
// reduction over the lanes of a half-wave, result in every lane; `wide` (uniform per half-wave)
// says whether lanes beyond the first quad hold data at all
template <class Op>
__device__ __forceinline__ float4 halfwave_reduce(float4 v, bool wide, Op op) {
    v.x = op(v.x, dpp<DPP_XOR1>(v.x)); v.y = op(v.y, dpp<DPP_XOR1>(v.y));
    v.z = op(v.z, dpp<DPP_XOR1>(v.z)); v.w = op(v.w, dpp<DPP_XOR1>(v.w));
    v.x = op(v.x, dpp<DPP_XOR2>(v.x)); v.y = op(v.y, dpp<DPP_XOR2>(v.y));
    v.z = op(v.z, dpp<DPP_XOR2>(v.z)); v.w = op(v.w, dpp<DPP_XOR2>(v.w));
    if (wide) {
#pragma unroll
        for (int off = 4; off < 32; off <<= 1) {
            v.x = op(v.x, __shfl_xor(v.x, off, 32)); v.y = op(v.y, __shfl_xor(v.y, off, 32));
            v.z = op(v.z, __shfl_xor(v.z, off, 32)); v.w = op(v.w, __shfl_xor(v.w, off, 32));
        }
    } else {  // every lane learns the first quad's result
        v.x = __shfl(v.x, 0, 32); v.y = __shfl(v.y, 0, 32); v.z = __shfl(v.z, 0, 32); v.w = __shfl(v.w, 0, 32);
    }
    return v;
}

// out[c] = sum_b partial[b][c]: one block per column, strided partial sums then a fixed-shape tree.
// One launch serves up to four such sums (the parameter gradients of one backward pass): block x
// belongs to the first job whose column count it falls into.
struct ColSum {
    const float* partial;  // [nb][width], the job's first column at partial[0]
    float* out;
    int nb, width, cols;
    // optional: column c's sum times fac_a[c*fan + t] -> out_a[c*fan + t], likewise b (t < fan <= 256)
    const float* fac_a;
    float* out_a;
    const float* fac_b;
    float* out_b;
    int fan;
};
struct ColSums {
    ColSum j[4];
};
[[maybe_unused]] __global__ __launch_bounds__(256) void k_reduce_cols(ColSums jobs) {
    __shared__ float red[256];
    int c = blockIdx.x, which = 0;
    while (which < 3 && c >= jobs.j[which].cols) c -= jobs.j[which++].cols;
    const float* __restrict__ partial = jobs.j[which].partial;
    float* __restrict__ out = jobs.j[which].out;
    const int nb = jobs.j[which].nb, width = jobs.j[which].width;
    const int t = threadIdx.x;
    float s = 0.f;
    for (int b = t; b < nb; b += 256) s += partial[(size_t)b * width + c];
    red[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) out[c] = red[0];
    const ColSum& job = jobs.j[which];
    if (job.fac_a && t < job.fan) {
        const float total = red[0];
        job.out_a[c * job.fan + t] = total * job.fac_a[c * job.fan + t];
        job.out_b[c * job.fan + t] = total * job.fac_b[c * job.fan + t];
    }
}

// Grid of the node-parallel kernels: a multiple of 8 (XCD grouping), at least 8 nodes per half-wave
// before more blocks are added, capped at 2048 (measured: finer cuts than co-residency win here).
// AIG_GAT_GRID_CAP overrides the cap (tuning aid).
[[maybe_unused]] int grid_blocks(int64_t n, int limit = 2048) {
    static const int env_cap = [] {
        const char* e = getenv("AIG_GAT_GRID_CAP");
        int c = e ? atoi(e) : 0;
        return c >= 8 && c <= 2048 ? c / 8 * 8 : 2048;  // the partial-sum areas hold 2048 blocks
    }();
    const int cap = env_cap < limit ? env_cap : limit;
    int64_t g = (n + 63) / 64;
    g = (g + 7) / 8 * 8;
    if (g > cap) g = cap;
    if (g < 8) g = 8;
    return (int)g;
}

// gradient of the level readout w.r.t. one node row, "pull" form: every (level, feature) maximum belongs
// to exactly one node, so a node asks whether it is that node instead of being scattered to
//   d_h[i,c] = w[i] g[c] + [argmax[level(i),c] == i] g[128+c] / (L+1)
__device__ __forceinline__ float4 readout_pull(int i, int lvl, float wi, float inv_levels, float4 g_mean, float4 g_max,
                                               const int4* __restrict__ argmax4, int l) {
    int4 am = make_int4(-1, -1, -1, -1);
    if (lvl >= 0) am = argmax4[lvl * (GAT_H / 4) + l];
    float4 r;
    r.x = wi * g_mean.x + (am.x == i ? g_max.x * inv_levels : 0.f);
    r.y = wi * g_mean.y + (am.y == i ? g_max.y * inv_levels : 0.f);
    r.z = wi * g_mean.z + (am.z == i ? g_max.z * inv_levels : 0.f);
    r.w = wi * g_mean.w + (am.w == i ? g_max.w * inv_levels : 0.f);
    return r;
}
// number of levels L + 1: from the plan's device-side header when the plan was built without a host read
// (meta[0] = deepest level), else from the caller's L
__device__ __forceinline__ int level_count(int L, const int* __restrict__ meta) { return (meta ? meta[0] : L) + 1; }

}  // namespace
}  // namespace aig
