// The distinct rows of the gradient penalty's edge-free critic pass (train.py:133-152).
//
// On an empty edge list the critic sees a node only through its interpolated features alpha x_r + (1 - alpha) x_f
// (one alpha per call), its level, and whether it holds a column's maximum in its level.  Node features are a type and
// a count of inverted fan-ins, so the (x_r, x_f) pairs of a minibatch take a few dozen values: nodes of one level
// with one pair - a CELL - go through both layers as identical rows, and every sum over nodes in the penalty and in
// its parameter gradient is a sum over cells with multiplicities.  Only the level maxima tell the nodes of a cell
// apart: a column's maximum belongs to the first node (lowest id) among those that attain it, which is the first
// node of its cell.  So a cell is represented by at most two rows: its first node with multiplicity 1 (it may hold
// maxima), and - when the cell has more nodes - its second node standing for the other m - 1.
//
// How a multiplicity enters without touching the penalty's kernels: the forward readout takes the weight m w of a
// row (w = 1 / (level size * levels), from the full plan), and the backward takes sqrt(m) w, so that a row's whole
// backward chain - linear in the readout's gradient - carries the factor sqrt(m); the squared norm, and every
// parameter gradient (a sum over rows of products of two such chains), then carry m.  First nodes have m = 1, so the
// maxima they hold are unscaled.
//
// The rows are emitted in node order (a stable compaction), so "first by id" keeps its meaning among them; their
// number stays on the device (n_rows), the caller sizes its buffers and launches by the node count.
// Pairs are told apart exactly (bit patterns of the four floats) through a small open-addressing table whose slots
// are claimed by node id (the key of a slot is its owner's features: nothing to publish); a minibatch with more
// than GPR_CLASSES distinct pairs keeps the overflowing nodes as rows of their own.
#include <climits>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace aig {
namespace {

constexpr int GPR_CLASSES = 256;       // slots of the pair table
constexpr int GPR_PROBES = 64;         // slots a node looks at before it gives up and stays a row of its own

struct Pair {
    unsigned a, b, c, d;
};
__device__ __forceinline__ Pair pair_of(const float* __restrict__ xr, const float* __restrict__ xf, int i) {
    const uint2 r = reinterpret_cast<const uint2*>(xr)[i], f = reinterpret_cast<const uint2*>(xf)[i];
    return {r.x, r.y, f.x, f.y};
}
__device__ __forceinline__ bool same(const Pair& p, const Pair& q) { return p.a == q.a && p.b == q.b && p.c == q.c && p.d == q.d; }
__device__ __forceinline__ unsigned hash_of(const Pair& p) {
    unsigned h = p.a * 0x9E3779B1u;
    h = (h ^ (h >> 15)) + p.b * 0x85EBCA77u;
    h = (h ^ (h >> 13)) + p.c * 0xC2B2AE3Du;
    h = (h ^ (h >> 16)) + p.d * 0x27D4EB2Fu;
    return h ^ (h >> 15);
}

// a node's level in the plan: its entry of node_level, or 0 for every node when the plan's header says that no node
// carried a valid depth (all nodes are then pooled in one level, discriminator.py:68-72)
__device__ __forceinline__ int level_of(const int* __restrict__ node_level, const int* __restrict__ meta, int i) {
    return (meta && (meta[2] & 1)) ? 0 : node_level[i];
}

// The lanes of a wave grouped by a key: `body(key, lanes of the group, am I its lowest lane)` runs once per distinct
// key among the lanes that take part, for the lanes of that group.  Every lane of the wave must call it.
template <class F>
__device__ __forceinline__ void for_each_cell(bool part, int key, F body) {
    unsigned long long todo = __ballot(part);
    while (todo) {
        const int lead = __ffsll(static_cast<long long>(todo)) - 1;
        const int k0 = __builtin_amdgcn_readlane(key, lead);
        const bool mine = part && key == k0;
        const unsigned long long members = __ballot(mine);
        if (mine) body(k0, members, (int)(threadIdx.x & 63) == lead);
        todo &= ~members;
    }
}

// cls[i] = slot of node i's pair (-1: table full for it, -2: the node has no level); per cell (level, slot): count
// and lowest node id
// Nodes are taken in the order of the plan's `perm` (by level, ids ascending within a level): the lanes of a wave then
// share a level or two and a handful of cells, so the per-wave merging of the atomics below has something to merge
// (in node order a wave of the 100k-node workload holds sixty different cells and the kernel took 120 us).
__global__ __launch_bounds__(1024) void k_gpr_class(int nvalid, const int* __restrict__ meta, const int* __restrict__ perm, int L,
                            const float* __restrict__ xr, const float* __restrict__ xf,
                            const int* __restrict__ node_level, int* __restrict__ owner, int* __restrict__ cls,
                            int* __restrict__ cnt, int* __restrict__ first) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (meta) nvalid = min(nvalid, meta[1]);      // (nodes with a level: the plan's header)
    const bool live = t < nvalid;
    const int i = live ? perm[t] : 0;
    const int lvl = live ? level_of(node_level, meta, i) : -1;
    const Pair me = pair_of(xr, xf, i);
    int h = (int)(hash_of(me) & (GPR_CLASSES - 1));
    int c = -1;
    // The lanes of a wave probe in step.  A free slot is claimed once per wave - by the lowest lane that wants it, for
    // all of them - and what a block has learnt about a slot stays in LDS: when the kernel starts, every wave of
    // the grid finds the table empty, and one compare-and-swap per NODE on a handful of addresses (a minibatch has a
    // few dozen pairs) took 300 us at 100k nodes.
    __shared__ int known[GPR_CLASSES], asks[GPR_CLASSES];
    for (int k = threadIdx.x; k < GPR_CLASSES; k += blockDim.x) {
        known[k] = -1;
        asks[k] = INT_MAX;
    }
    __syncthreads();
    // the slots the block's nodes look at first are settled by the block as a whole: one claim per (block, slot)
    for_each_cell(lvl >= 0 && lvl <= L, h, [&](int slot, unsigned long long, bool leader) {
        if (leader) atomicMin(&asks[slot], i);
    });
    __syncthreads();
    for (int k = threadIdx.x; k < GPR_CLASSES; k += blockDim.x) {
        if (asks[k] == INT_MAX) continue;
        int o = __hip_atomic_load(&owner[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o < 0) {
            o = atomicCAS(&owner[k], -1, asks[k]);
            if (o < 0) o = asks[k];
        }
        known[k] = o;
    }
    __syncthreads();
    bool need = lvl >= 0 && lvl <= L;       // (a level past the cell table: a row of its own)
    for (int probe = 0; probe < GPR_PROBES && __any(need); ++probe) {
        int o = need ? __hip_atomic_load(&known[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
        if (need && o < 0) o = __hip_atomic_load(&owner[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for_each_cell(need && o < 0, h, [&](int slot, unsigned long long members, bool leader) {
            int got = 0;
            if (leader) {
                got = atomicCAS(&owner[slot], -1, i);
                if (got < 0) got = i;
            }
            o = __shfl(got, __ffsll(static_cast<long long>(members)) - 1);
        });
        if (need) {
            __hip_atomic_store(&known[h], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // (one value per slot, whoever stores it)
            if (o == i || same(me, pair_of(xr, xf, o))) {
                c = h;
                need = false;
            } else {
                h = (h + 1) & (GPR_CLASSES - 1);
            }
        }
    }
    if (live) cls[i] = c;
    // one update per (wave, cell) instead of one per node: neighbouring nodes share levels and pairs, and a hundred
    // thousand atomics on a few thousand addresses queue up behind each other (212 us at 100k nodes, now ~10)
    for_each_cell(c >= 0, c >= 0 ? lvl * GPR_CLASSES + c : -1, [&](int cell, unsigned long long members, bool leader) {
        if (leader) {      // the lowest lane of the cell: ids ascend with the lane
            atomicAdd(&cnt[cell], __popcll(members));
            if (first[cell] > i) atomicMin(&first[cell], i);
        }
    });
}

// second[cell] = lowest node id of the cell besides its first
__global__ void k_gpr_second(int nvalid, const int* __restrict__ meta, const int* __restrict__ perm,
                             const int* __restrict__ node_level, const int* __restrict__ cls,
                             const int* __restrict__ first, int* __restrict__ second) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (meta) nvalid = min(nvalid, meta[1]);
    const int i = t < nvalid ? perm[t] : 0;
    const int c = t < nvalid ? cls[i] : -1;
    const int cell_i = c >= 0 ? level_of(node_level, meta, i) * GPR_CLASSES + c : -1;
    const bool cand = c >= 0 && first[cell_i] != i;
    for_each_cell(cand, cell_i, [&](int cell, unsigned long long, bool leader) {
        if (leader && second[cell] > i) atomicMin(&second[cell], i);     // the group's lowest id
    });
}

// keep[i] = multiplicity of the row node i stands for (0: it is represented by another node)
__global__ void k_gpr_keep(int n, const int* __restrict__ meta, const int* __restrict__ node_level, const int* __restrict__ cls,
                           const int* __restrict__ cnt, const int* __restrict__ first, const int* __restrict__ second,
                           int* __restrict__ mult, int* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int lvl = level_of(node_level, meta, i);
    const int c = lvl < 0 ? -2 : cls[i];      // (nodes without a level were never classified)
    int m = 0;
    if (c == -1) {
        m = 1;
    } else if (c >= 0) {
        const size_t cell = (size_t)lvl * GPR_CLASSES + c;
        if (first[cell] == i) m = 1;
        else if (second[cell] == i) m = cnt[cell] - 1;
    }
    mult[i] = m;
    flag[i] = m > 0 ? 1 : 0;
}

// rows in node order; everything past the last row gets neutral values (no level, weight 0)
__global__ void k_gpr_emit(int n, const int* __restrict__ meta, const float* __restrict__ xr, const float* __restrict__ xf,
                           const int* __restrict__ node_level, const float* __restrict__ w, const int* __restrict__ mult,
                           const int* __restrict__ pos, float* __restrict__ xr_rows, float* __restrict__ xf_rows,
                           long long* __restrict__ depth_rows, float* __restrict__ w_fwd, float* __restrict__ w_bwd,
                           int* __restrict__ n_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int total = pos[n - 1] + (mult[n - 1] > 0 ? 1 : 0);
    if (i == 0) n_rows[0] = total;
    const int m = mult[i];
    if (m > 0) {
        const int r = pos[i];
        reinterpret_cast<float2*>(xr_rows)[r] = reinterpret_cast<const float2*>(xr)[i];
        reinterpret_cast<float2*>(xf_rows)[r] = reinterpret_cast<const float2*>(xf)[i];
        depth_rows[r] = level_of(node_level, meta, i);
        const float wi = w[i];
        w_fwd[r] = (float)m * wi;
        w_bwd[r] = sqrtf((float)m) * wi;
    }
    if (i >= total) {
        reinterpret_cast<float2*>(xr_rows)[i] = make_float2(0.f, 0.f);
        reinterpret_cast<float2*>(xf_rows)[i] = make_float2(0.f, 0.f);
        depth_rows[i] = -1;
        w_fwd[i] = 0.f;
        w_bwd[i] = 0.f;
    }
}

size_t scan_bytes(int64_t n) {
    size_t bytes = 0;
    int* p = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, p, p, 0, (size_t)(n > 0 ? n : 1), rocprim::plus<int>(), hipStream_t(0));
    return bytes;
}

}  // namespace
}  // namespace aig

using namespace aig;

extern "C" size_t aig_gp_rows_ws_bytes(int64_t n, int64_t L) {
    const size_t N = (size_t)(n > 0 ? n : 1), cells = (size_t)(L + 1) * GPR_CLASSES;
    return 4 * align_up(N * 4) + 3 * align_up(cells * 4) + align_up(GPR_CLASSES * 4) + align_up(scan_bytes(n)) + 256;
}

extern "C" int aig_gp_rows(int64_t n, int64_t nvalid, int64_t L, const int32_t* lvl_meta, const int32_t* perm,
                           const float* x_real, const float* x_fake, const int32_t* node_level, const float* w, float* xr_rows, float* xf_rows, int64_t* depth_rows, float* w_fwd,
                           float* w_bwd, int32_t* n_rows, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n > 0 && n < INT_MAX && L >= 0 && (L + 1) * (int64_t)GPR_CLASSES < INT_MAX, "aig_gp_rows: bad sizes n=%lld L=%lld",
                (long long)n, (long long)L);
    AIG_REQUIRE(nvalid >= 0 && nvalid <= n && perm, "aig_gp_rows: the full plan's perm and its number of nodes with a level are needed");
    AIG_REQUIRE(x_real && x_fake && node_level && w && xr_rows && xf_rows && depth_rows && w_fwd && w_bwd && n_rows && ws,
                "aig_gp_rows: null pointer");
    if (ws_bytes < aig_gp_rows_ws_bytes(n, L)) return fail(AIG_EWORKSPACE, "aig_gp_rows: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t cells = (size_t)(L + 1) * GPR_CLASSES;
    Carver c(ws);
    int* cls = c.take<int>(n);
    int* mult = c.take<int>(n);
    int* flag = c.take<int>(n);
    int* pos = c.take<int>(n);
    int* cnt = c.take<int>(cells);
    int* first = c.take<int>(cells);
    int* second = c.take<int>(cells);
    int* owner = c.take<int>(GPR_CLASSES);
    size_t sb = scan_bytes(n);
    void* temp = c.take<char>(sb);
    AIG_HIP_CHECK(fill_words(cnt, 0u, cells * 4, s));
    AIG_HIP_CHECK(fill_words(first, 0x7fffffffu, cells * 4, s));
    AIG_HIP_CHECK(fill_words(second, 0x7fffffffu, cells * 4, s));
    AIG_HIP_CHECK(fill_words(owner, 0xffffffffu, GPR_CLASSES * 4, s));
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_gpr_class, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, s, (int)nvalid, lvl_meta, perm, (int)L, x_real, x_fake, node_level, owner, cls,
                       cnt, first);
    hipLaunchKernelGGL(k_gpr_second, grid, block, 0, s, (int)nvalid, lvl_meta, perm, node_level, cls, first, second);
    hipLaunchKernelGGL(k_gpr_keep, grid, block, 0, s, (int)n, lvl_meta, node_level, cls, cnt, first, second, mult, flag);
    AIG_LAUNCH_CHECK();
    AIG_HIP_CHECK(rocprim::exclusive_scan(temp, sb, flag, pos, 0, (size_t)n, rocprim::plus<int>(), s));
    hipLaunchKernelGGL(k_gpr_emit, grid, block, 0, s, (int)n, lvl_meta, x_real, x_fake, node_level, w, mult, pos, xr_rows, xf_rows,
                       reinterpret_cast<long long*>(depth_rows), w_fwd, w_bwd, n_rows);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}
