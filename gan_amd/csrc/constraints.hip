// AIG structural constraints as linear-time integer kernels: repair (enforce), violation counts
// (penalties), strict validation verdicts, attribute resync.  All byte/integer work: sorts for
// grouping and de-duplication, integer atomics (order independent), level-synchronous traversals.
#include <climits>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"
#include "radix.h"

namespace aig {
namespace {

constexpr int PI_T = 0, PO_T = 1, AND_T = 2;
constexpr int TRAV_THREADS = 1024;

struct Cnt4 {
    int a, b, c, d;
};
struct Cnt4Plus {
    __host__ __device__ Cnt4 operator()(const Cnt4& x, const Cnt4& y) const {
        return {x.a + y.a, x.b + y.b, x.c + y.c, x.d + y.d};
    }
};

// ------------------------------------------------------------------------------------------ types and pos
__global__ void k_type_flags(int n, const float* __restrict__ x, int* __restrict__ types, Cnt4* __restrict__ flags) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += gridDim.x * blockDim.x) {
        if (i == n) { flags[n] = {0, 0, 0, 0}; continue; }
        int t = static_cast<int>(x[2 * i]);
        types[i] = t;
        flags[i] = {t == PI_T, t == AND_T, t == PO_T, (t != PI_T && t != AND_T && t != PO_T)};
    }
}
// A set of B graphs of n nodes each, worked on as one graph of N = B n nodes: node i of graph g is union
// node g n + i.  Edge slot e of graph g: source ei[(2g) ecap + e], target ei[(2g + 1) ecap + e], inversion
// value ea[g ecap + e]; only the first ecount[g] slots of a graph hold edges (all ecap when ecount is null).
// A single graph is the set B = 1, ecap = E.  Graphs never share an edge, so sorts, scans and level-by-level
// walks over the union give every graph the result it gets alone; what is per graph by definition
// (ranks among a graph's own nodes, counts, the cycle verdict) is taken per graph below.
struct GraphSet {
    int B, n, ecap;
    const int* ecount;
    __host__ __device__ int N() const { return B * n; }
    __host__ __device__ int M() const { return B * ecap; }
    // a graph without a single edge takes the early exits of the reference (constraints.py:264-269,517-522)
    __device__ bool empty(int g) const { return ecount && ecount[g] <= 0; }
};
__device__ __forceinline__ Cnt4 operator-(const Cnt4& x, const Cnt4& y) { return {x.a - y.a, x.b - y.b, x.c - y.c, x.d - y.d}; }

// pos = rank in (all PI, all AND, all PO, anything else) among the nodes of the node's own graph, index
// order inside a group (constraints.py:118-132).  small[3g + p] = union id of the PI/AND node of graph g
// at pos p < 3.
__global__ void k_pos(GraphSet gs, const int* __restrict__ types, const Cnt4* __restrict__ scan, int* __restrict__ pos,
                      int* __restrict__ small /*[3 B]*/) {
    const int N = gs.N();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int g = i / gs.n;
        const Cnt4 base = scan[g * gs.n];
        const Cnt4 tot = scan[(g + 1) * gs.n] - base;
        const int t = types[i];
        const Cnt4 s = scan[i] - base;
        int p = t == PI_T ? s.a : t == AND_T ? tot.a + s.b : t == PO_T ? tot.a + tot.b + s.c : tot.a + tot.b + tot.c + s.d;
        pos[i] = p;
        if (small && p < 3 && (t == PI_T || t == AND_T)) small[3 * g + p] = i;
    }
}

__device__ __forceinline__ bool in_range(long long u, long long v, int n) { return u >= 0 && u < n && v >= 0 && v < n; }

// Edge slot `e` of the set (0 <= e < B ecap): its graph and its endpoints as the input holds them (ids inside
// the graph, whatever their value); false for a slot past its graph's edge count.
__device__ __forceinline__ bool load_edge(const GraphSet& gs, const long long* __restrict__ ei, int e, int& g,
                                          long long& u, long long& v) {
    g = gs.B == 1 ? 0 : e / gs.ecap;
    const int le = e - g * gs.ecap;
    if (gs.ecount && le >= gs.ecount[g]) return false;
    const size_t row = (size_t)(2 * g) * gs.ecap + le;
    u = ei[row];
    v = ei[row + gs.ecap];
    return true;
}

// ------------------------------------------------------------------------------------------ sort helpers
int bits_for(unsigned long long maxval) {
    int b = 1;
    while (b < 64 && (1ull << b) <= maxval) ++b;
    return b;
}
template <typename K>
size_t sort_temp(int64_t E, int bits) {
    return rs::temp_bytes<K, int>((size_t)(E > 0 ? E : 1), bits);
}
size_t scan_temp(int64_t n) {
    size_t bytes = 0;
    int* p = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, p, p, 0, (size_t)n, rocprim::plus<int>(), hipStream_t(0));
    size_t b4 = 0;
    Cnt4* q = nullptr;
    (void)rocprim::exclusive_scan(nullptr, b4, q, q, Cnt4{0, 0, 0, 0}, (size_t)n, Cnt4Plus(), hipStream_t(0));
    return bytes > b4 ? bytes : b4;
}

template <typename K>
__global__ void k_segment_ptr(int n, int E, const K* __restrict__ skey, K scale, int* __restrict__ ptr) {
    // ptr[v] = first sorted position whose key is >= v*scale, v = 0..n
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v <= n; v += gridDim.x * blockDim.x) {
        const K target = (K)v * scale;
        int lo = 0, hi = E;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (skey[mid] < target) lo = mid + 1; else hi = mid;
        }
        ptr[v] = lo;
    }
}

// ------------------------------------------------------------------------------------------ traversal
// Level-synchronous frontier walk by ONE workgroup over CSR (ptr, adj).
//   MODE 0: breadth-first levels; depth[] preset to 0 on the sources and -1 elsewhere.
//   MODE 1: Kahn peel; state[] = indegree, every node released in the round after its last
//           predecessor, so depth = longest-path level (-1: never released, the node sits on or behind a
//           cycle); *popped counts released nodes.
//           (state[] is scratch: it is counted down in place or left as it was.)
// Every wave leaves through the same `break` (the frontier size is read between two barriers),
// and the loop ends after at most n rounds because a node enters a queue at most once.
constexpr int HEAVY_DEG = 256;   // adjacency lists longer than this are expanded by the whole block
constexpr int HEAVY_CAP = 2048;  // heavy nodes remembered per level (the rest are expanded in place)

template <int MODE>
__device__ __forceinline__ void visit(int v, int level, int* __restrict__ state, int* __restrict__ depth,
                                      int* __restrict__ qn, int* s_next) {
    if (MODE == 0) {
        // peek first (an L1-bypassing load): a popular node is claimed once and then only read, instead
        // of tens of thousands of compare-and-swaps queueing on one address
        if (__hip_atomic_load(&depth[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != -1) return;
        if (atomicCAS(&depth[v], -1, level + 1) == -1) qn[atomicAdd(s_next, 1)] = v;
    } else {
        if (atomicSub(&state[v], 1) == 1) {
            depth[v] = level + 1;
            qn[atomicAdd(s_next, 1)] = v;
        }
    }
}

// One traversal per block: block b works on job[b] (independent walks share a launch and overlap).
struct TravJob {
    const int* ptr;
    const int* adj;
    int* state;
    int* depth;
    int* q0;
    int* q1;
    int* popped;
};
struct TravJobs {
    TravJob j[2];
    const int* gate;   // per graph of a set: walk the graph only when gate[graph] != 0 (null: always)
};

// A set of graphs is walked by one block per graph (blockIdx.y; no edge leaves a graph): block (job, g)
// owns the n nodes from g n on, ids in queues and adjacency lists stay ids of the whole set.
template <int MODE>
__global__ __launch_bounds__(TRAV_THREADS) void k_traverse(int n, TravJobs jobs) {
    if (jobs.gate && jobs.gate[blockIdx.y] == 0) return;   // (uniform over the block, before any barrier)
    const TravJob jb = jobs.j[blockIdx.x];
    const int base = blockIdx.y * n;
    const int* __restrict__ ptr = jb.ptr;
    const int* __restrict__ adj = jb.adj;
    int* __restrict__ state = jb.state;
    int* __restrict__ depth = jb.depth;
    int* __restrict__ q0 = jb.q0 + base;
    int* __restrict__ q1 = jb.q1 + base;
    int* __restrict__ popped = gridDim.y == 1 ? jb.popped : nullptr;
    __shared__ int s_cnt[2];
    __shared__ int s_total;
    __shared__ int s_nheavy;
    __shared__ int s_heavy[HEAVY_CAP];
    const int tid = threadIdx.x;
    if (tid == 0) { s_cnt[0] = s_cnt[1] = 0; s_total = 0; s_nheavy = 0; }
    __syncthreads();
    for (int l = tid; l < n; l += TRAV_THREADS) {
        const int i = base + l;
        bool start;
        if (MODE == 0) start = depth[i] == 0;
        else { start = state[i] == 0; depth[i] = start ? 0 : -1; }
        if (start) q0[atomicAdd(&s_cnt[0], 1)] = i;
    }
    int cur = 0;
    for (int level = 0; level <= n; ++level) {
        __syncthreads();
        const int c = s_cnt[cur];
        __syncthreads();
        if (c == 0) break;
        if (tid == 0) { s_cnt[cur ^ 1] = 0; s_total += c; s_nheavy = 0; }
        __syncthreads();
        const int* qc = cur ? q1 : q0;
        int* qn = cur ? q0 : q1;
        // short adjacency lists: one thread each; long ones are set aside for whole waves
        for (int idx = tid; idx < c; idx += TRAV_THREADS) {
            const int u = qc[idx];
            const int b = ptr[u], e = ptr[u + 1];
            if (e - b > HEAVY_DEG) {
                const int slot = atomicAdd(&s_nheavy, 1);
                if (slot < HEAVY_CAP) { s_heavy[slot] = u; continue; }
            }
            for (int k = b; k < e; ++k) visit<MODE>(adj[k], level, state, depth, qn, &s_cnt[cur ^ 1]);
        }
        __syncthreads();
        // long lists: the whole block strides over one list at a time (a generated graph can hang
        // most of its edges on a single source)
        const int nh = min(s_nheavy, HEAVY_CAP);
        for (int hidx = 0; hidx < nh; ++hidx) {
            const int u = s_heavy[hidx];
            for (int k = ptr[u] + tid; k < ptr[u + 1]; k += TRAV_THREADS)
                visit<MODE>(adj[k], level, state, depth, qn, &s_cnt[cur ^ 1]);
        }
        cur ^= 1;
    }
    __syncthreads();
    if (tid == 0 && popped) *popped = s_total;
}

// ------------------------------------------------------------------------------------------ traversal, LDS-resident
// The same walks with the per-level latency cut to about one global round trip: a level of an AIG
// holds a few hundred nodes, so a level-synchronous walk is a chain of dependent memory accesses
// and nothing else.  Here
//   - the frontier queues live in LDS (entries beyond TRAV_QCAP spill to the global queues),
//   - MODE 0 keeps its visited set as a bitmap in LDS (test-and-set there; depth[] is only stored);
//     MODE 1 keeps the in-degree counters in LDS as bytes while n <= TRAV_KAHN_LDS_MAX_N and no
//     in-degree exceeds 255 (global atomics otherwise),
//   - the adjacency header of a node and its first two neighbours come in one 16-byte record,
//   - two barriers per level (three rotating counters): frontier nodes take the two neighbours
//     that came with their record, longer lists are set aside and walked by a wave or the block,
//   - depth[] is written by a wave of its own, so that no expanding wave waits behind stores.
// Used while the bitmap fits (n <= TRAV_LDS_MAX_N); larger graphs take k_traverse.
constexpr int TRAV_QCAP = 2048;
constexpr int TRAV_LDS_MAX_N = 1 << 20;
constexpr int TRAV_KAHN_LDS_MAX_N = 1 << 17;  // Kahn: one byte of LDS per node for the in-degree counters
constexpr int TRAV_BIG_CAP = 960;   // adjacency lists set aside per level for a wave each
constexpr int TRAV_HEAVY_CAP = 32;  // ... and for the whole block (beyond either: walked in place)

// {first, one past last, neighbour 0, neighbour 1} of every node's adjacency list.  (A longer
// record with more neighbours inline was measured slower: the records of a 100k-node graph then
// span 6.4 MB instead of 1.6 MB, and the per-level dependent load pays for it.)
__global__ void k_node_rec(int n, const int* __restrict__ ptr, const int* __restrict__ adj, int4* __restrict__ rec) {
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < n; u += gridDim.x * blockDim.x) {
        const int b = ptr[u], e = ptr[u + 1];
        rec[u] = make_int4(b, e, b < e ? adj[b] : -1, b + 1 < e ? adj[b + 1] : -1);
    }
}

struct TravLds {
    int base;        // first node of the block's graph: bitmap and LDS counters are indexed by v - base
    unsigned* bits;  // MODE 0: visited bitmap; MODE 1: in-degree counters, one byte per node (see lds_cnt)
    bool lds_cnt;    // MODE 1: the counters live in LDS (every in-degree <= 255 and n small enough)
    unsigned* dummy; // 64 scratch words
    int* lq;         // two LDS queues of TRAV_QCAP entries, back to back
    int* gq0;        // their global spill areas
    int* gq1;
    int* cnt;        // three rotating frontier sizes
};
// queue `which`, entry idx.  Written as explicit LDS / global branches: a pointer picked at run time
// between the two memories turns every access into a flat instruction, and those make the
// compiler drain all outstanding stores before each one.
__device__ __forceinline__ void q_put(const TravLds& t, int which, int idx, int v) {
    if (idx < TRAV_QCAP) t.lq[which * TRAV_QCAP + idx] = v;
    else if (which) t.gq1[idx - TRAV_QCAP] = v;
    else t.gq0[idx - TRAV_QCAP] = v;
}
// Readers walk the LDS part and the spilled part of a queue in two loops of their own (a single
// loop with a branch per entry is merged back into one flat load by the compiler).
__device__ __forceinline__ int q_get_lds(const TravLds& t, int which, int idx) { return t.lq[which * TRAV_QCAP + idx]; }
__device__ __forceinline__ int q_get_spill(const TravLds& t, int which, int idx) {
    return which ? t.gq1[idx - TRAV_QCAP] : t.gq0[idx - TRAV_QCAP];
}

// Claims v for the next frontier.  No global store here: in-order completion counting would make
// the next level's record load wait for this level's scattered depth stores (measured: +1400
// cycles per level); depth[] is written by a wave of its own, see k_traverse_lds.
template <int MODE>
__device__ __forceinline__ void visit_lds(int v, const TravLds& t, int nx, int ncnt, int* __restrict__ state) {
    const int l = v - t.base;
    if (MODE == 0) {
        const unsigned m = 1u << (l & 31);
        if (t.bits[l >> 5] & m) return;  // peek: a popular node is claimed once and then only read
        if (atomicOr(&t.bits[l >> 5], m) & m) return;
    } else if (t.lds_cnt) {
        const int sh = 8 * (l & 3);
        if (((atomicSub(&t.bits[l >> 2], 1u << sh) >> sh) & 255u) != 1u) return;
    } else {
        if (atomicSub(&state[v], 1) != 1) return;
    }
    q_put(t, nx, atomicAdd(&t.cnt[ncnt], 1), v);
}

// The two neighbours that come with a node's record, claimed side by side: both bitmap reads, then
// both test-and-sets, are in flight together, and one counter update covers the pair.  An absent
// neighbour (on == false) goes through the same instructions with an empty mask.
template <int MODE>
__device__ __forceinline__ void visit2_lds(int self, int v0, bool on0, int v1, bool on1, const TravLds& t, int nx,
                                           int ncnt, int* __restrict__ state) {
    bool c0, c1;
    const int l0 = v0 - t.base, l1 = v1 - t.base;
    if (MODE == 0) {
        // (an absent neighbour aims at a per-lane dummy word: same-address atomics are served one
        //  lane at a time, a shared dummy would serialise the idle lanes of every wave)
        unsigned* const dummy = t.dummy + (threadIdx.x & 63);
        unsigned* const q0 = on0 ? &t.bits[l0 >> 5] : dummy;
        unsigned* const q1 = on1 ? &t.bits[l1 >> 5] : dummy;
        const unsigned m0 = on0 ? 1u << (l0 & 31) : 0u, m1 = on1 && v1 != v0 ? 1u << (l1 & 31) : 0u;
        const unsigned p0 = *q0, p1 = *q1;
        const unsigned a0 = m0 & ~p0, a1 = m1 & ~p1;  // still unclaimed at the peek
        const unsigned o0 = atomicOr(q0, a0), o1 = atomicOr(q1, a1);
        c0 = (a0 & ~o0) != 0u;
        c1 = (a1 & ~o1) != 0u;
    } else if (t.lds_cnt) {
        // byte counters in LDS: the decrement of byte b of a word is a subtraction of 1 << 8b (a
        // counter never goes below zero, so no borrow reaches its neighbour)
        unsigned* const dummy = t.dummy + (threadIdx.x & 63);
        const int s0 = 8 * (l0 & 3), s1 = 8 * (l1 & 3);
        const unsigned o0 = atomicSub(on0 ? &t.bits[l0 >> 2] : dummy, on0 ? 1u << s0 : 0u);
        const unsigned o1 = atomicSub(on1 ? &t.bits[l1 >> 2] : dummy, on1 ? 1u << s1 : 0u);
        c0 = on0 && ((o0 >> s0) & 255u) == 1u;
        c1 = on1 && ((o1 >> s1) & 255u) == 1u;
    } else {
        // both counter updates in flight together; an absent neighbour subtracts 0 from the
        // expanding node's own (already released) counter, a distinct address per lane
        const int o0 = atomicSub(&state[on0 ? v0 : self], on0 ? 1 : 0);
        const int o1 = atomicSub(&state[on1 ? v1 : self], on1 ? 1 : 0);
        c0 = on0 && o0 == 1;
        c1 = on1 && o1 == 1;
    }
    // queue slots for the whole wave with one counter update (a per-lane update with differing
    // amounts is turned by the compiler into a loop over the active lanes)
    const unsigned long long b0 = __ballot(c0), b1 = __ballot(c1);
    if (b0 | b1) {
        const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1ull;
        int base = 0;
        if ((__ballot(1) & below) == 0ull) base = atomicAdd(&t.cnt[ncnt], __popcll(b0) + __popcll(b1));
        base = __builtin_amdgcn_readfirstlane(base) + __popcll(b0 & below) + __popcll(b1 & below);
        if (c0) q_put(t, nx, base, v0);
        if (c1) q_put(t, nx, base + (int)c0, v1);
    }
}

template <int MODE>
__global__ __launch_bounds__(TRAV_THREADS) void k_traverse_lds(int n, TravJobs jobs, const int4* __restrict__ rec0,
                                                               const int4* __restrict__ rec1) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    if (jobs.gate && jobs.gate[blockIdx.y] == 0) return;   // (uniform over the block, before any barrier)
    const TravJob jb = jobs.j[blockIdx.x];
    const int base = blockIdx.y * n;   // one block per graph of a set, see k_traverse
    const int4* __restrict__ rec = blockIdx.x ? rec1 : rec0;
    const int* __restrict__ adj = jb.adj;
    int* __restrict__ state = jb.state;
    int* __restrict__ depth = jb.depth;
    __shared__ int s_cnt[3];
    __shared__ int s_total;
    __shared__ int s_nbig[2], s_nheavy[2];
    __shared__ int2 s_big[TRAV_BIG_CAP], s_heavy[TRAV_HEAVY_CAP];
    __shared__ unsigned s_dummy[64];
    const int tid = threadIdx.x;
    __shared__ int s_wide;
    const int nwords = MODE == 0 ? (n + 31) >> 5 : (n <= TRAV_KAHN_LDS_MAX_N ? (n + 3) >> 2 : 0);
    TravLds t;
    t.base = base;
    t.bits = reinterpret_cast<unsigned*>(smem);
    t.lds_cnt = false;
    t.dummy = s_dummy;
    t.lq = smem + nwords;
    t.gq0 = jb.q0 + base;
    t.gq1 = jb.q1 + base;
    t.cnt = s_cnt;
    if (tid == 0) { s_cnt[0] = s_cnt[1] = s_cnt[2] = 0; s_total = 0; s_nbig[0] = s_nbig[1] = 0; s_nheavy[0] = s_nheavy[1] = 0; s_wide = 0; }
    if (MODE == 0) {
        for (int w = tid; w < nwords; w += TRAV_THREADS) t.bits[w] = 0u;
    } else {
        // Kahn: the in-degree counters go to LDS, four to a word, unless one of them needs more than a byte
        __syncthreads();
        bool wide = false;
        for (int w = tid; w < nwords; w += TRAV_THREADS) {
            unsigned word = 0u;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int i = 4 * w + b;
                const int d = i < n ? state[base + i] : 0;
                wide |= d > 255 || d < 0;
                word |= ((unsigned)d & 255u) << (8 * b);
            }
            t.bits[w] = word;
        }
        if (wide) s_wide = 1;
    }
    __syncthreads();
    if (MODE == 1) t.lds_cnt = nwords > 0 && s_wide == 0;
    for (int l = tid; l < n; l += TRAV_THREADS) {
        const int i = base + l;
        bool start;
        if (MODE == 0) start = depth[i] == 0;
        else { start = state[i] == 0; depth[i] = -1; }
        if (start) {
            if (MODE == 0) atomicOr(&t.bits[l >> 5], 1u << (l & 31));
            q_put(t, 0, atomicAdd(&s_cnt[0], 1), i);
        }
    }
    // The last wave only writes depth[] (level of every node of the current frontier) and never
    // waits for those stores; the other waves expand the frontier and never store.
    constexpr int EXPANDERS = TRAV_THREADS - 64;
    const bool writer = tid >= EXPANDERS;
    const int lane = tid & 63, wave = tid >> 6;
    // level L reads queue L&1 / counter L%3, fills queue (L+1)&1 / counter (L+1)%3, and clears
    // counter (L+2)%3 and the list sizes of parity (L+1)&1, which nobody touches during L.
    // Every wave leaves through the same `break`.
    constexpr int XWAVES = EXPANDERS / 64;
    for (int level = 0; level <= n; ++level) {
        const int cur = level & 1, nx = cur ^ 1, ccnt = level % 3, ncnt = (level + 1) % 3;
        __syncthreads();
        const int c = s_cnt[ccnt];
        if (c == 0) break;
        if (tid == 0) { s_cnt[(level + 2) % 3] = 0; s_total += c; s_nbig[nx] = 0; s_nheavy[nx] = 0; }
        // phase A: one thread per frontier node takes the two neighbours that came with the record;
        // a longer list is set aside so that a wave (or, when very long, the block) walks it
        const int c_lds = min(c, TRAV_QCAP);
        if (writer) {
            for (int idx = lane; idx < c_lds; idx += 64) depth[q_get_lds(t, cur, idx)] = level;
            for (int idx = TRAV_QCAP + lane; idx < c; idx += 64) depth[q_get_spill(t, cur, idx)] = level;
        } else {
            auto expand = [&](int u) {
                const int4 r = rec[u];
                const int deg = r.y - r.x;
                visit2_lds<MODE>(u, r.z, deg > 0, r.w, deg > 1, t, nx, ncnt, state);
                if (deg > 2) {
                    const bool heavy = deg - 2 > HEAVY_DEG;
                    const int slot = atomicAdd(heavy ? &s_nheavy[cur] : &s_nbig[cur], 1);
                    if (slot < (heavy ? TRAV_HEAVY_CAP : TRAV_BIG_CAP)) {
                        (heavy ? s_heavy : s_big)[slot] = make_int2(r.x + 2, r.y);
                    } else {  // list full: walk it here
                        for (int k = r.x + 2; k < r.y; ++k) visit_lds<MODE>(adj[k], t, nx, ncnt, state);
                    }
                }
            };
            for (int idx = tid; idx < c_lds; idx += EXPANDERS) expand(q_get_lds(t, cur, idx));
            for (int idx = TRAV_QCAP + tid; idx < c; idx += EXPANDERS) expand(q_get_spill(t, cur, idx));
        }
        __syncthreads();
        // phase B
        const int nb = min(s_nbig[cur], TRAV_BIG_CAP), nh = min(s_nheavy[cur], TRAV_HEAVY_CAP);
        if (!writer) {
            for (int i = wave; i < nb; i += XWAVES) {
                const int2 be = s_big[i];
                for (int k = be.x + lane; k < be.y; k += 64) visit_lds<MODE>(adj[k], t, nx, ncnt, state);
            }
            for (int i = 0; i < nh; ++i) {
                const int2 be = s_heavy[i];
                for (int k = be.x + tid; k < be.y; k += EXPANDERS) visit_lds<MODE>(adj[k], t, nx, ncnt, state);
            }
        }
    }
    __syncthreads();
    if (tid == 0 && jb.popped && gridDim.y == 1) *jb.popped = s_total;
}

// rec0 / rec1: one 16-byte record per node for job 0 / job 1 (workspace of N entries each).  The N nodes are
// `parts` graphs of N / parts nodes: one block per job and graph (the LDS regimes go by a graph's node count).
template <int MODE>
int run_traverse(hipStream_t s, int N, const TravJobs& jobs, int njobs, int4* rec0, int4* rec1, int parts = 1) {
    const int total = N;
    N = total / parts;
    if (N > TRAV_LDS_MAX_N) {
        hipLaunchKernelGGL(k_traverse<MODE>, dim3(njobs, parts), dim3(TRAV_THREADS), 0, s, N, jobs);
        AIG_LAUNCH_CHECK();
        return AIG_OK;
    }
    AIG_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_traverse_lds<MODE>), (size_t)((TRAV_LDS_MAX_N / 8) + 2 * TRAV_QCAP * 4)));
    const int g = (total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096;
    hipLaunchKernelGGL(k_node_rec, dim3(g), dim3(256), 0, s, total, jobs.j[0].ptr, jobs.j[0].adj, rec0);
    if (njobs > 1) hipLaunchKernelGGL(k_node_rec, dim3(g), dim3(256), 0, s, total, jobs.j[1].ptr, jobs.j[1].adj, rec1);
    const size_t lds = (MODE == 0 ? (size_t)((N + 31) >> 5) * 4 : (N <= TRAV_KAHN_LDS_MAX_N ? (size_t)((N + 3) >> 2) * 4 : 0)) +
                       (size_t)2 * TRAV_QCAP * 4;
    hipLaunchKernelGGL(k_traverse_lds<MODE>, dim3(njobs, parts), dim3(TRAV_THREADS), lds, s, N, jobs, rec0, rec1);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

// adjacency in sorted order: adj[p] = other endpoint (union id) of edge slot sid[p]; which = 1: the target
// (`relabel`, may be null: the lists then hold relabel[id] instead of the id - nodes renamed by their place in an order)
__global__ void k_gather_adj(GraphSet gs, const int* __restrict__ sid, const long long* __restrict__ ei, int which,
                             int* __restrict__ adj, const int* __restrict__ relabel) {
    const int m = gs.M();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < m; p += gridDim.x * blockDim.x) {
        int g;
        long long u = -1, v = -1;
        const bool have = load_edge(gs, ei, sid[p], g, u, v);
        const long long w = which ? v : u;
        const int id = have && w >= 0 && w < gs.n ? g * gs.n + (int)w : 0;   // entries past the valid keys are never read
        adj[p] = relabel ? relabel[id] : id;
    }
}

__global__ void k_seed_depth(int n, const int* __restrict__ types, int want_type, int* __restrict__ depth) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        depth[i] = types[i] == want_type ? 0 : -1;
}

// ------------------------------------------------------------------------------------------ levels by a sweep
// Breadth-first levels from the PIs over the CLEANED edges without walking level by level.  A cleaned edge runs
// from a lower to a higher position in the graph's PI / AND / PO order (pos[u] < pos[v] is what "cleaned" means,
// constraints.py:541-546), so that order is a topological one and the level of a node is
//     d[v] = 0 for a PI,  1 + min d[u] over its predecessors that have a level,  none (-1) otherwise
// - one pass over the positions.  One workgroup per graph takes the positions a tile of 1024 at a time: the levels
// of the last RING positions live in LDS, a tile first takes the minimum over its predecessors from before the
// tile (final values) and then relaxes the predecessors inside the tile until nothing changes (values only ever
// decrease, so concurrent reads and writes are harmless and the loop ends).  The dependent steps cost an LDS
// round trip instead of the level walk's two memory round trips per level (0.6 ms -> 0.1 ms for the graph an
// untrained generator wires on 100k nodes: 326 levels); what a tile needs from memory - list bounds, the first
// predecessors, levels from before the LDS window - is requested one and two tiles ahead.
// Lists are kept in POSITION space (sorted by target position, entries = source positions), so consecutive
// threads read consecutive list bounds and neighbouring lists.
constexpr int SWEEP_THREADS = 1024;
constexpr int SWEEP_PT = 4;            // positions per thread
constexpr int SWEEP_T = SWEEP_THREADS * SWEEP_PT;   // positions per tile
constexpr int SWEEP_RING = 16384;      // levels kept in LDS (64 KB)
constexpr int SWEEP_INL = 2;           // predecessors per node kept in registers; longer lists are re-read
constexpr int SWEEP_NONE = 0x3fffffff;
constexpr int SWEEP_ROUNDS = 4;        // relaxation rounds between two barriers

// key = union position of the target, value = union position of the source; dropped edges sort last
__global__ void k_sweep_keys(GraphSet gs, const long long* __restrict__ ei, const int* __restrict__ pos,
                             unsigned* __restrict__ key, int* __restrict__ val) {
    const int M = gs.M(), N = gs.N();
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < M; e += gridDim.x * blockDim.x) {
        int g;
        long long u = -1, v = -1;
        bool ok = load_edge(gs, ei, e, g, u, v) && u != v && in_range(u, v, gs.n);
        int pu = 0, pv = 0;
        if (ok) {
            pu = pos[g * gs.n + (int)u];
            pv = pos[g * gs.n + (int)v];
            ok = pu < pv && pv < gs.n;     // (a node of no known type sits past the POs; pos stays below n)
        }
        key[e] = ok ? (unsigned)(g * gs.n + pv) : (unsigned)N;
        val[e] = ok ? g * gs.n + pu : 0;
    }
}
// seed[position] = 0 for a PI, none otherwise
__global__ void k_sweep_seed(GraphSet gs, const int* __restrict__ types, const int* __restrict__ pos, int* __restrict__ seed) {
    const int N = gs.N();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int p = pos[i];
        if (p < gs.n) seed[(i / gs.n) * gs.n + p] = types[i] == PI_T ? 0 : SWEEP_NONE;
    }
}
// depth[node] = level at the node's position (none -> -1)
__global__ void k_sweep_out(GraphSet gs, const int* __restrict__ pos, const int* __restrict__ dpos, int* __restrict__ depth) {
    const int N = gs.N();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int p = pos[i];
        const int d = p < gs.n ? dpos[(i / gs.n) * gs.n + p] : SWEEP_NONE;
        depth[i] = d >= SWEEP_NONE ? -1 : d;
    }
}

struct SweepNode {          // what a thread holds of one position ahead of its tile
    int lo, hi;             // its list in tsrc
    int seed;
    int q[SWEEP_INL];       // first predecessors (positions inside the graph)
    int dq[SWEEP_INL];      // their levels when they lie before the LDS window of the tile (else unused)
};
struct SweepTile {
    SweepNode nd[SWEEP_PT];  // positions p0 + j * SWEEP_THREADS + tid
};

// A tile is SWEEP_PT positions per thread: what a tile costs is mostly the memory latency that shows between two
// tiles (a barrier waits for the requests made ahead), so fewer, larger tiles - the relaxation inside a tile runs
// at LDS speed.
// MODE 0: breadth-first level from the seeds (min over the predecessors that have a level).  MODE 1: longest-path
// level (0 without predecessors, else 1 + max over the predecessors; `seed` unused) - levels only ever INCREASE in
// the relaxation there.  `gate` (may be null): a graph is swept only when gate[graph] == 0.
struct SweepJob {
    const int* tptr;
    const int* tsrc;
    const int* seed;
    int* dpos;
};
template <int MODE>
__device__ __forceinline__ void sweep_body(const GraphSet& gs, const SweepJob& jb, const int graph) {
    const int* __restrict__ tptr = jb.tptr;
    const int* __restrict__ tsrc = jb.tsrc;
    const int* __restrict__ seed = jb.seed;
    int* __restrict__ dpos = jb.dpos;
    __shared__ int ring[SWEEP_RING + 1];   // the last slot: a neutral level for "no predecessor"
    __shared__ int flag[3];
    volatile int* ring_v = ring;     // (the rounds between two barriers must re-read LDS)
    const int n = gs.n, base = graph * gs.n, tid = threadIdx.x;
    const int ntiles = (n + SWEEP_T - 1) / SWEEP_T;
    constexpr int MASK = SWEEP_RING - 1;
    // Every load below is unconditional on a clamped index (a load under a condition compiles to a branch with a
    // wait behind it, which would put the memory round trips of a thread's positions one after the other).
    const int last_e = max(gs.M() - 1, 0);
    auto bounds = [&](int t, SweepTile& tl) {            // stage 1: list bounds and seed of tile t
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j) {
            SweepNode& nd = tl.nd[j];
            const int p = t * SWEEP_T + j * SWEEP_THREADS + tid;
            const bool live = t < ntiles && p < n;
            const int pc = base + (live ? p : 0);
            const int lo = tptr[pc], hi = tptr[pc + 1], sd = MODE == 0 ? seed[pc] : SWEEP_NONE;
            nd.lo = live ? lo : 0;
            nd.hi = live ? hi : 0;
            nd.seed = live ? sd : SWEEP_NONE;
        }
    };
    auto lists = [&](int t, SweepTile& tl) {             // stage 2: first predecessors, and their levels when final
        const int window_lo = t * SWEEP_T - (SWEEP_RING - SWEEP_T);   // positions from here on are in LDS for tile t
        int raw[SWEEP_PT][SWEEP_INL];
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j)
#pragma unroll
            for (int k = 0; k < SWEEP_INL; ++k) raw[j][k] = tsrc[min(tl.nd[j].lo + k, last_e)];
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j)
#pragma unroll
            for (int k = 0; k < SWEEP_INL; ++k)
                tl.nd[j].q[k] = tl.nd[j].lo + k < tl.nd[j].hi ? raw[j][k] - base : -1;
        int old[SWEEP_PT][SWEEP_INL];
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j)
#pragma unroll
            for (int k = 0; k < SWEEP_INL; ++k) {
                const int q = tl.nd[j].q[k];
                old[j][k] = dpos[base + (q >= 0 && q < window_lo ? q : 0)];
            }
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j)
#pragma unroll
            for (int k = 0; k < SWEEP_INL; ++k) {
                const int q = tl.nd[j].q[k];
                tl.nd[j].dq[k] = q >= 0 && q < window_lo ? old[j][k] : (MODE == 0 ? SWEEP_NONE : -1);
            }
    };
    SweepTile cur, nxt, nn;
    int par = 0;
    if (tid < 3) flag[tid] = 0;
    if (tid == 0) ring[SWEEP_RING] = MODE == 0 ? SWEEP_NONE : -1;
    __syncthreads();
    bounds(0, cur);
    lists(0, cur);
    bounds(1, nxt);
    for (int t = 0; t < ntiles; ++t) {
        const int p0 = t * SWEEP_T;
        const int window_lo = p0 - (SWEEP_RING - SWEEP_T);
        // requests for the tiles ahead go out before this tile's dependent work
        lists(t + 1, nxt);
        bounds(t + 2, nn);
        int d[SWEEP_PT], inl[SWEEP_PT][SWEEP_INL];
        bool anylong = false;
        // predecessors from before the tile: final
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j) {
            const SweepNode& nd = cur.nd[j];
            int best = MODE == 0 ? SWEEP_NONE : -1;
#pragma unroll
            for (int k = 0; k < SWEEP_INL; ++k) {
                const int q = nd.q[k];
                const int rv = ring[(q >= 0 ? q : 0) & MASK];
                // inside the tile: relaxed below (none: the neutral slot)
                inl[j][k] = q >= p0 ? (q & MASK) : SWEEP_RING;
                if (q >= 0 && q < p0) {
                    const int dv = q < window_lo ? nd.dq[k] : rv;
                    best = MODE == 0 ? min(best, dv) : max(best, dv);
                }
            }
            if (nd.hi - nd.lo > SWEEP_INL) {
                anylong = true;
                for (int e = nd.lo + SWEEP_INL; e < nd.hi; ++e) {
                    const int q = tsrc[e] - base;
                    if (q < p0) {
                        const int dv = q < window_lo ? dpos[base + q] : ring[q & MASK];
                        best = MODE == 0 ? min(best, dv) : max(best, dv);
                    }
                }
            }
            if (MODE == 0) d[j] = nd.seed == 0 ? 0 : (best < SWEEP_NONE ? best + 1 : SWEEP_NONE);
            else d[j] = best + 1;
        }
        __syncthreads();                    // every read of the slots this tile overwrites has been made
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j) {
            const int p = p0 + j * SWEEP_THREADS + tid;
            if (p < n) ring[p & MASK] = d[j];
        }
        __syncthreads();
        // predecessors inside the tile: relax until no level changes.  Levels only decrease and LDS is coherent
        // within the workgroup, so a few rounds run between two barriers (a round sees whatever the other waves
        // have written by then); the barrier only decides whether anybody changed anything in the whole set of
        // rounds - if nobody did, every value read in it was already final.
        while (true) {
            bool ch = false;
#pragma unroll 1
            for (int round = 0; round < SWEEP_ROUNDS; ++round) {
                int got[SWEEP_PT][SWEEP_INL];
                asm volatile("" ::: "memory");        // (LDS is read again in every round)
#pragma unroll
                for (int j = 0; j < SWEEP_PT; ++j)
#pragma unroll
                    for (int k = 0; k < SWEEP_INL; ++k) got[j][k] = ring[inl[j][k]];
#pragma unroll
                for (int j = 0; j < SWEEP_PT; ++j) {
                    const SweepNode& nd = cur.nd[j];
                    int nv = d[j];
#pragma unroll
                    for (int k = 0; k < SWEEP_INL; ++k) nv = MODE == 0 ? min(nv, got[j][k] + 1) : max(nv, got[j][k] + 1);
                    if (MODE == 0 && nd.seed == 0) nv = 0;
                    if (anylong && nd.hi - nd.lo > SWEEP_INL) {
                        for (int e = nd.lo + SWEEP_INL; e < nd.hi; ++e) {
                            const int q = tsrc[e] - base;
                            if (q >= p0) nv = MODE == 0 ? min(nv, ring_v[q & MASK] + 1) : max(nv, ring_v[q & MASK] + 1);
                        }
                    }
                    if ((MODE == 0 ? nv < d[j] : nv > d[j]) && p0 + j * SWEEP_THREADS + tid < n) {
                        d[j] = nv;
                        ring[(p0 + j * SWEEP_THREADS + tid) & MASK] = nv;
                        ch = true;
                    }
                }
            }
            if (__any(ch) && (tid & 63) == 0) flag[par] = 1;
            // a barrier that waits for this wave's LDS traffic only: the requests made for the tiles ahead stay in
            // flight through the relaxation (__syncthreads would wait for them here)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const bool go = flag[par] != 0;
            // three flags in turn: the one cleared here was last read before the barrier above and is next raised
            // behind the next barrier (with two, a fast wave could raise the other flag before this store lands)
            if (tid == 0) flag[par == 0 ? 2 : par - 1] = 0;
            par = par == 2 ? 0 : par + 1;
            if (!go) break;
        }
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j) {
            const int p = p0 + j * SWEEP_THREADS + tid;
            if (p < n) dpos[base + p] = d[j];
        }
        cur = nxt;
        nxt = nn;
    }
}

template <int MODE>
__global__ __launch_bounds__(SWEEP_THREADS) void k_sweep(GraphSet gs, SweepJob job, const int* __restrict__ gate) {
    if (gate && gate[blockIdx.x] != 0) return;
    sweep_body<MODE>(gs, job, blockIdx.x);
}

// Which positions reach a seed along the edges?  The mirror image of the reachability sweep above, for the
// direction in which a PULL would read the lists by source - and the graphs an untrained generator wires hang tens
// of thousands of targets on two sources.  So this one PUSHES along the lists by target (a handful of entries per
// position): the positions are taken a tile at a time from the LAST to the first; every edge runs upward, so when a
// tile's turn comes, every position above it has already pushed its mark down.  Inside a tile a marked position
// marks its predecessors - in the tile's LDS flags when they lie in the tile, in memory below it - exactly once;
// the waves poll their positions' flags until a count of marks not yet pushed on reaches zero.  reach[position] = 0
// (reaches a seed; the seeds are preset) or SWEEP_NONE.  One workgroup per graph.
constexpr int REACH_SPINS = 1 << 22;      // polls of a tile at most (a chain of marks needs a few per hop)
__device__ __forceinline__ void reach_back_body(const GraphSet& gs, const int* __restrict__ tptr,
                                                const int* __restrict__ tsrc, int* __restrict__ reach, const int graph) {
    __shared__ int cur[SWEEP_T];
    __shared__ int pending;
    volatile int* cur_v = cur;
    const int n = gs.n, base = graph * gs.n, tid = threadIdx.x;
    const int ntiles = (n + SWEEP_T - 1) / SWEEP_T;
    const int last_e = max(gs.M() - 1, 0);
    struct Bounds {
        int lo[SWEEP_PT], hi[SWEEP_PT];
    };
    struct Node {
        int lo, hi, q[SWEEP_INL];
    };
    // what a tile needs from memory comes in two dependent steps (list bounds, then the first predecessors): the
    // bounds are requested two tiles ahead and the predecessors one tile ahead, with unconditional loads on clamped
    // indices, so that no tile waits for a round trip of its own
    auto bounds = [&](int t, Bounds& bd) {
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j) {
            const int p = t * SWEEP_T + j * SWEEP_THREADS + tid;
            const bool live = t >= 0 && p < n;
            const int pc = base + (live ? p : 0);
            const int lo = tptr[pc], hi = tptr[pc + 1];
            bd.lo[j] = live ? lo : 0;
            bd.hi[j] = live ? hi : 0;
        }
    };
    auto lists = [&](const Bounds& bd, Node* nd) {
        int raw[SWEEP_PT][SWEEP_INL];
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j)
#pragma unroll
            for (int k = 0; k < SWEEP_INL; ++k) raw[j][k] = tsrc[min(bd.lo[j] + k, last_e)];
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j) {
            nd[j].lo = bd.lo[j];
            nd[j].hi = bd.hi[j];
#pragma unroll
            for (int k = 0; k < SWEEP_INL; ++k) nd[j].q[k] = raw[j][k] - base;
        }
    };
    Bounds bnext;
    Node nx[SWEEP_PT];
    bounds(ntiles - 1, bnext);
    lists(bnext, nx);
    bounds(ntiles - 2, bnext);
    for (int t = ntiles - 1; t >= 0; --t) {
        const int a = t * SWEEP_T;
        Node nd[SWEEP_PT];
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j) nd[j] = nx[j];
        // marks pushed into this tile from above.  Every access to `reach` in here is a workgroup-scope atomic: the
        // only readers and writers are the waves of this one workgroup, which share a CU and its cache - device scope
        // would send every access past the XCD's L2 and make the fence below write that L2 back (24 us per tile)
        int got[SWEEP_PT];
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j) {
            const int p = a + j * SWEEP_THREADS + tid;
            got[j] = __hip_atomic_load(&reach[base + min(p, n - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        lists(bnext, nx);          // the next tile's predecessors and the bounds of the one after it: requested before
        bounds(t - 2, bnext);      // this tile's rounds
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j)
            cur[j * SWEEP_THREADS + tid] = a + j * SWEEP_THREADS + tid < n ? got[j] : SWEEP_NONE;
        // `pending` = positions of the tile that carry a mark their owner has not pushed on yet.  Whoever lowers a flag
        // of the tile raises it (an exchange tells the first marker from the others), the owner lowers it after pushing
        // the mark to the position's predecessors; the waves poll their own positions until it reaches zero.  A chain
        // of marks inside the tile then advances at the pace of an LDS poll (rounds between barriers advanced it a hop
        // or two per barrier: the first tiles of what an untrained generator wires hold chains of a hundred hops and
        // took 110 us each, 77 us now).  What is left is the chain itself: some 900 marked positions of such a tile hang
        // one behind the other, two LDS round trips a hop; polling faster (all flags of a poll in flight at once, one
        // word per wave instead of its flags) measured slower, the polls then stand in the way of the marks.
        // REACH_SPINS bounds the polling whatever happens.
        // (the count is raised BEFORE the flag is lowered and taken back when another marker was first: raised after,
        // the owner could push the mark on and lower the count to zero in between, and waves would leave early)
        auto mark = [&](int at) {
            atomicAdd(&pending, 1);
            if (atomicExch(&cur[at], 0) == 0) atomicSub(&pending, 1);
        };
        bool done[SWEEP_PT];
        int mine = 0;
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j) {
            done[j] = false;
            mine += (a + j * SWEEP_THREADS + tid < n && got[j] == 0) ? 1 : 0;
        }
        if (tid == 0) pending = 0;
        __syncthreads();
        {
            const unsigned long long any_mine = __ballot(mine > 0);
            if (any_mine) {      // one LDS add per wave
                int tot = mine;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
                if ((tid & 63) == 0) atomicAdd(&pending, tot);
            }
        }
        __syncthreads();
        for (int spin = 0; spin < REACH_SPINS; ++spin) {
            if (__hip_atomic_load(&pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) break;
            bool fresh[SWEEP_PT];
            bool some = false;
#pragma unroll
            for (int j = 0; j < SWEEP_PT; ++j) {
                fresh[j] = !done[j] && cur_v[j * SWEEP_THREADS + tid] == 0;
                some = some || fresh[j];
            }
            if (!__any(some)) {
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            int below[SWEEP_PT][SWEEP_INL];
            int pushed = 0;
#pragma unroll
            for (int j = SWEEP_PT - 1; j >= 0; --j) {
                done[j] = done[j] || fresh[j];
                pushed += fresh[j] ? 1 : 0;
#pragma unroll
                for (int k = 0; k < SWEEP_INL; ++k) {
                    const int q = nd[j].q[k];
                    const bool has = fresh[j] && nd[j].lo + k < nd[j].hi;
                    below[j][k] = has && q < a ? q : -1;
                    if (has && q >= a && cur_v[q - a] != 0) mark(q - a);
                }
            }
            // below the tile: plain stores, nothing waits for them before the tile ends (a load first - is the mark
            // there already? - would keep the wave from seeing new marks on its positions for a memory round trip)
#pragma unroll
            for (int j = 0; j < SWEEP_PT; ++j)
#pragma unroll
                for (int k = 0; k < SWEEP_INL; ++k)
                    if (below[j][k] >= 0)
                        __hip_atomic_store(&reach[base + below[j][k]], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            // lists longer than the inline entries (rare): entry by entry
#pragma unroll
            for (int j = 0; j < SWEEP_PT; ++j) {
                if (!fresh[j]) continue;
                for (int e = nd[j].lo + SWEEP_INL; e < nd[j].hi; ++e) {
                    const int q = tsrc[e] - base;
                    if (q >= a) {
                        if (cur_v[q - a] != 0) mark(q - a);
                    } else {
                        __hip_atomic_store(&reach[base + q], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            if (pushed) atomicSub(&pending, pushed);     // (after this thread's marks: the count never reads zero early)
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SWEEP_PT; ++j) {
            const int p = a + j * SWEEP_THREADS + tid;
            if (p < n && cur[j * SWEEP_THREADS + tid] == 0 && got[j] != 0)
                __hip_atomic_store(&reach[base + p], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __threadfence_block();   // the marks pushed below this tile are visible before any wave loads the next tile
        __syncthreads();      // (and the tile's flags are overwritten next)
    }
}

// Both reachability passes of a graph in one launch: blockIdx.y = 0 sweeps forward from the seeds of `fwd` (levels,
// of which only "has one" is used), blockIdx.y = 1 marks what reaches a seed of `back_reach` (preset there).
__global__ __launch_bounds__(SWEEP_THREADS) void k_reach_pair(GraphSet gs, SweepJob fwd, const int* __restrict__ tptr,
                                                             const int* __restrict__ tsrc, int* __restrict__ back_reach,
                                                             const int* __restrict__ gate) {
    if (gate && gate[blockIdx.x] != 0) return;
    if (blockIdx.y == 0) sweep_body<0>(gs, fwd, blockIdx.x);
    else reach_back_body(gs, tptr, tsrc, back_reach, blockIdx.x);
}

// The resync's edges in position space (every edge with both ends in range counts there, AIGgan.py:29-62):
// key = union position of the target, value = union position of the source.  unordered[g] is raised when an edge
// of graph g does not run from a lower to a higher position (a self loop included): then the position order is no
// topological order of that graph and its levels come from the Kahn walk instead of the sweep.
__global__ void k_rs_sweep_keys(GraphSet gs, const long long* __restrict__ ei, const int* __restrict__ pos,
                                unsigned* __restrict__ key, int* __restrict__ val, int* __restrict__ unordered) {
    const int M = gs.M(), N = gs.N();
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < M; e += gridDim.x * blockDim.x) {
        int g;
        long long u = -1, v = -1;
        const bool ok = load_edge(gs, ei, e, g, u, v) && in_range(u, v, gs.n);
        int pu = 0, pv = 0;
        if (ok) {
            pu = pos[g * gs.n + (int)u];
            pv = pos[g * gs.n + (int)v];
            if (pu >= pv) unordered[g] = 1;     // (same value from every thread that stores it)
        }
        key[e] = ok ? (unsigned)(g * gs.n + pv) : (unsigned)N;
        val[e] = ok ? g * gs.n + pu : 0;
    }
}
// depth[node] = level at the node's position, for the graphs the sweep took
__global__ void k_rs_sweep_out(GraphSet gs, const int* __restrict__ pos, const int* __restrict__ dpos,
                               const int* __restrict__ unordered, int* __restrict__ depth) {
    const int N = gs.N();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int g = i / gs.n;
        if (unordered[g] == 0) depth[i] = dpos[g * gs.n + pos[i]];
    }
}

// ------------------------------------------------------------------------------------------ enforce
// key of a cleaned edge = v*n + u (grouped by target, then source); dropped edges sort last
__global__ void k_enf_keys(GraphSet gs, const long long* __restrict__ ei, const int* __restrict__ pos,
                           unsigned long long* __restrict__ key, unsigned* __restrict__ key_src, int* __restrict__ ids) {
    const int M = gs.M(), N = gs.N();
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < M; e += gridDim.x * blockDim.x) {
        int g;
        long long u = -1, v = -1;
        bool ok = load_edge(gs, ei, e, g, u, v) && u != v && in_range(u, v, gs.n);
        const int gu = g * gs.n + (int)u, gv = g * gs.n + (int)v;
        ok = ok && pos[gu] < pos[gv];
        key[e] = ok ? (unsigned long long)gv * (unsigned long long)N + (unsigned long long)gu : ~0ull;
        key_src[e] = ok ? (unsigned)gu : (unsigned)N;
        ids[e] = e;
    }
}

// head[p] = first of a run of equal (u,v); minv[p] at a head = merged inversion value of the run:
// the first occurrence's value, raised to 1 if it was 0 and any repeat carries 1 (constraints.py:36-41)
__global__ void k_enf_merge(int m_total, const unsigned long long* __restrict__ skey, const int* __restrict__ sid,
                            const long long* __restrict__ inv, long long* __restrict__ minv) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < m_total; p += gridDim.x * blockDim.x) {
        const unsigned long long k = skey[p];
        if (k == ~0ull) continue;
        if (p > 0 && skey[p - 1] == k) { minv[p] = LLONG_MIN; continue; }  // not a head
        long long val = inv ? inv[sid[p]] : 0;
        for (int q = p + 1; q < m_total && skey[q] == k; ++q)
            if (val == 0 && inv && inv[sid[q]] == 1) val = 1;
        minv[p] = val;
    }
}

struct Pick {
    int s0, s1;
    long long i0, i1;
    int count;
};

__device__ __forceinline__ long long depth_trunc(int kind, const void* depth, size_t u) {
    if (kind == 1) return (long long)static_cast<const float*>(depth)[u];  // int(): toward zero
    if (kind == 2) return static_cast<const long long*>(depth)[u];
    return 0;
}

// one thread per target: choose the fan-ins to keep (constraints.py:558-620) and sum inversions
// (ids are union ids here; depth holds depth_stride entries per graph - 0: one vector shared by all graphs)
__global__ void k_enf_select(GraphSet gs, const int* __restrict__ types, const int* __restrict__ pos,
                             const int* __restrict__ small, const Cnt4* __restrict__ scan,
                             const unsigned long long* __restrict__ skey, const long long* __restrict__ minv,
                             const int* __restrict__ vptr, int depth_kind, const void* __restrict__ depth,
                             long long depth_stride, Pick* __restrict__ picks, int* __restrict__ pick_cnt,
                             float* __restrict__ x_out, const float* __restrict__ x) {
    const int n = gs.N();
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v <= n; v += gridDim.x * blockDim.x) {
        if (v == n) { pick_cnt[n] = 0; continue; }
        const int g = v / gs.n;
        const Cnt4 tot = scan[(g + 1) * gs.n] - scan[g * gs.n];
        const int n_src_nodes = tot.a + tot.b;  // PI + AND of the node's graph
        const int t = types[v];
        const int beg = vptr[v], end = vptr[v + 1];
        Pick pk{-1, -1, 0, 0, 0};
        long long inv_sum = 0;
        int nl = 0;
        int b_and = 0; long long b_dep = 0;  // PO: best so far
        int p1 = INT_MAX, p2 = INT_MAX;      // AND: two smallest pos
        for (int p = beg; p < end; ++p) {
            const long long iv = minv[p];
            if (iv == LLONG_MIN) continue;  // repeat of the previous pair
            inv_sum += iv;
            const int u = (int)(skey[p] - (unsigned long long)v * (unsigned long long)n);  // n = nodes of the union
            const int tu = types[u];
            if (!(tu == PI_T || tu == AND_T)) continue;
            ++nl;
            if (t == PO_T) {
                const int is_and = tu == AND_T;
                const long long dep = depth_trunc(depth_kind, depth, (size_t)g * depth_stride + (u - g * gs.n));
                bool better;
                if (pk.s0 < 0) better = true;
                else if (is_and != b_and) better = is_and > b_and;
                else if (dep != b_dep) better = dep > b_dep;
                else better = u < pk.s0;
                if (better) { pk.s0 = u; pk.i0 = iv; b_and = is_and; b_dep = dep; }
            } else if (t == AND_T) {
                const int pu = pos[u];
                if (pu < p1) { p2 = p1; pk.s1 = pk.s0; pk.i1 = pk.i0; p1 = pu; pk.s0 = u; pk.i0 = iv; }
                else if (pu < p2) { p2 = pu; pk.s1 = u; pk.i1 = iv; }
            }
        }
        if (gs.empty(g)) pk.count = 0;   // no edge at all: nothing is repaired
        else if (t == PO_T) pk.count = nl >= 1 ? 1 : 0;
        else if (t == AND_T) {
            if (nl >= 2) pk.count = 2;
            else if (n_src_nodes - 1 >= 2) {
                // widen with every other PI/AND node (inversion 0): the two smallest pos besides v
                int c0 = -1, c1 = -1;
                for (int j = 0; j < 3; ++j) {
                    int u = j < n_src_nodes ? small[3 * g + j] : -1;
                    if (u < 0 || u == v) continue;
                    if (c0 < 0) c0 = u; else if (c1 < 0) c1 = u;
                }
                const int legal = nl == 1 ? pk.s0 : -1;
                const long long legal_inv = pk.i0;
                pk.s0 = c0; pk.i0 = (c0 == legal) ? legal_inv : 0;
                pk.s1 = c1; pk.i1 = (c1 == legal) ? legal_inv : 0;
                pk.count = (c0 >= 0 && c1 >= 0) ? 2 : 0;
            }
        }
        picks[v] = pk;
        pick_cnt[v] = pk.count;
        x_out[2 * v] = x[2 * v];
        x_out[2 * v + 1] = (float)(inv_sum > 2 ? 2 : inv_sum);
    }
}

// graph g's repaired edges go to its own slots: sources out_ei[(2g) cap + ..], targets out_ei[(2g + 1) cap + ..]
// (ids inside the graph again), out_count[g] of them, in target order
__global__ void k_enf_write(GraphSet gs, long long cap, const Pick* __restrict__ picks, const int* __restrict__ off,
                            long long* __restrict__ out_ei, long long* __restrict__ out_ea, int* __restrict__ out_count) {
    const int N = gs.N();
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
        const Pick pk = picks[v];
        const int g = v / gs.n, first = g * gs.n;
        const int o = off[v] - off[first];
        long long* src = out_ei + (size_t)(2 * g) * cap;
        long long* dst = src + cap;
        long long* ea = out_ea + (size_t)g * cap;
        if (pk.count >= 1) { src[o] = pk.s0 - first; dst[o] = v - first; ea[o] = pk.i0; }
        if (pk.count >= 2) { src[o + 1] = pk.s1 - first; dst[o + 1] = v - first; ea[o + 1] = pk.i1; }
        if (v == first) out_count[g] = off[first + gs.n] - off[first];
    }
}

// breadth-first levels as int64; a graph without edges reports -1 everywhere (constraints.py:517-522)
__global__ void k_enf_depth_out(GraphSet gs, const int* __restrict__ in, long long* __restrict__ out) {
    const int N = gs.N();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
        out[i] = gs.empty(i / gs.n) ? -1 : in[i];
}

// ------------------------------------------------------------------------------------------ penalties / validate
// counters (PEN_CTRS per graph): 0 pi_in, 1 po_out, 2 backedge, 3 selfloop, 4 fanin_type
constexpr int PEN_CTRS = 16;

// counters[g][idx] += amount, one atomic per wave for the lanes that share the first lane's graph (neighbouring
// lanes hold neighbouring edges / nodes, so that is all of them except where a wave straddles two graphs):
// a per-lane atomic on a handful of addresses is served one lane at a time.  Every lane of the wave must call.
__device__ __forceinline__ void wave_count(int* __restrict__ counters, int g, int idx, int amount) {
    const int g0 = __builtin_amdgcn_readfirstlane(g);
    int mine = g == g0 ? amount : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&counters[g0 * PEN_CTRS + idx], mine);
    if (g != g0 && amount) atomicAdd(&counters[g * PEN_CTRS + idx], amount);
}
__global__ void k_pen_edges(GraphSet gs, const long long* __restrict__ ei, const int* __restrict__ types,
                            const int* __restrict__ pos, int* __restrict__ counters, int* __restrict__ indeg,
                            unsigned long long* __restrict__ pair_key, unsigned* __restrict__ key_src,
                            unsigned* __restrict__ key_dst, int* __restrict__ ids, unsigned char* __restrict__ eflag,
                            const int* __restrict__ ord, int* __restrict__ unordered) {
    const int M = gs.M(), N = gs.N();
    // whole waves stay in the loop (a lane past the end carries an absent edge): the counts are summed per wave
    for (int base = blockIdx.x * blockDim.x; base < M; base += gridDim.x * blockDim.x) {
        const int e = base + threadIdx.x;
        int g = 0;
        long long lu = -1, lv = -1;
        const bool have = e < M && load_edge(gs, ei, e, g, lu, lv);
        const bool ok = have && in_range(lu, lv, gs.n);
        const int u = ok ? g * gs.n + (int)lu : 0, v = ok ? g * gs.n + (int)lv : 0;
        const int tu = types[u], tv = types[v];
        const bool self = ok && u == v, to_pi = ok && tv == PI_T, from_po = ok && tu == PO_T;
        const bool back = ok && !(pos[u] < pos[v]);
        const bool bad_src = ok && (tv == AND_T || tv == PO_T) && !(tu == PI_T || tu == AND_T);
        if (counters) {
            wave_count(counters, g, 0, to_pi);
            wave_count(counters, g, 1, from_po);
            wave_count(counters, g, 2, back);
            wave_count(counters, g, 3, self);
            wave_count(counters, g, 4, bad_src);
        }
        if (e >= M) continue;
        ids[e] = e;
        // (the reference cannot index an edge with an endpoint out of range; it is left out of every count)
        if (eflag) eflag[e] = (unsigned char)((self ? 1 : 0) | (to_pi ? 4 : 0) | (from_po ? 8 : 0) | (bad_src ? 16 : 0));
        if (ok) atomicAdd(&indeg[v], 1);
        pair_key[e] = ok ? (unsigned long long)u * (unsigned long long)N + (unsigned long long)v : ~0ull;
        // the lists behind the reachability passes: by node id, or - with `ord` - by the nodes' places in an order
        // in which every edge of a well-formed graph runs upward (unordered[g] says where that does not hold)
        const int ou = ord ? ord[u] : u, ov = ord ? ord[v] : v;
        if (ord && ok && u != v && !(ou < ov)) unordered[g] = 1;      // (same value from every thread that stores it)
        key_src[e] = ok ? (unsigned)ou : (unsigned)N;
        key_dst[e] = ok ? (unsigned)ov : (unsigned)N;
    }
}

// counters[5] += repeats of an already seen (u,v); flags those edges with bit 2 (validator)
__global__ void k_pen_dups(GraphSet gs, const unsigned long long* __restrict__ skey, const int* __restrict__ sid,
                           int* __restrict__ counters, unsigned char* __restrict__ eflag) {
    const int M = gs.M();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < M; p += gridDim.x * blockDim.x) {
        if (p == 0 || skey[p] == ~0ull || skey[p] != skey[p - 1]) continue;
        if (counters) {
            const int g = gs.B == 1 ? 0 : (int)(skey[p] / (unsigned long long)gs.N()) / gs.n;   // the pair's source
            atomicAdd(&counters[g * PEN_CTRS + 5], 1);
        }
        if (eflag) eflag[sid[p]] |= 2;
    }
}

// counters[6] += |indeg - target|; counters[7] += node not (reached forward from a PI and backward from a PO)
// (with `ord`: fwd / rev and the sweeps' fwd_sweep / rev_sweep are indexed by ord[node]; a graph's verdicts come
//  from the sweeps unless unordered[graph] is raised)
__global__ void k_pen_nodes(GraphSet gs, const int* __restrict__ types, const int* __restrict__ indeg,
                            const int* __restrict__ fwd, const int* __restrict__ rev, int* __restrict__ counters,
                            unsigned char* __restrict__ on_path, const int* __restrict__ ord,
                            const int* __restrict__ unordered, const int* __restrict__ fwd_sweep,
                            const int* __restrict__ rev_sweep) {
    const int N = gs.N();
    for (int base = blockIdx.x * blockDim.x; base < N; base += gridDim.x * blockDim.x) {   // whole waves: see k_pen_edges
        const int i = base + threadIdx.x;
        const bool live = i < N;
        const int j = live ? i : 0;
        const int t = types[j];
        const int target = t == PO_T ? 1 : (t == AND_T ? 2 : 0);
        const int d = indeg[j] - target;
        const int g = j / gs.n;
        const int o = ord ? ord[j] : j;
        const bool walked = !ord || unordered[g] != 0;
        const bool ok = walked ? (fwd[o] >= 0 && rev[o] >= 0) : (fwd_sweep[o] < SWEEP_NONE && rev_sweep[o] < SWEEP_NONE);
        wave_count(counters, g, 6, live ? (d < 0 ? -d : d) : 0);
        wave_count(counters, g, 7, live && !ok);
        if (live && on_path) on_path[i] = ok;
    }
}

// An order for the reachability sweeps.  hint (may be null): a caller's claim "place of every node in an order in
// which sources precede targets" (ids inside the graph, stride 0: one vector shared by the graphs); it is used
// for a graph only when it is a permutation of 0..n-1 there (inv: scratch; a value claimed twice leaves one claimant
// disagreeing with inv), else - and without a hint - the PI / AND / PO position is the order.  ord = union id.
__global__ void k_ord_scatter(GraphSet gs, const int* __restrict__ hint, long long stride, int* __restrict__ inv) {
    const int N = gs.N();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int g = i / gs.n, li = i - g * gs.n;
        const int h = hint[(size_t)g * stride + li];
        if (h >= 0 && h < gs.n) inv[g * gs.n + h] = i;
    }
}
__global__ void k_ord_check(GraphSet gs, const int* __restrict__ hint, long long stride, const int* __restrict__ inv,
                            int* __restrict__ bad) {
    const int N = gs.N();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int g = i / gs.n, li = i - g * gs.n;
        const int h = hint[(size_t)g * stride + li];
        if (!(h >= 0 && h < gs.n) || inv[g * gs.n + h] != i) bad[g] = 1;
    }
}
__global__ void k_ord_pick(GraphSet gs, const int* __restrict__ hint, long long stride, const int* __restrict__ pos,
                           const int* __restrict__ bad, int* __restrict__ ord) {
    const int N = gs.N();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int g = i / gs.n, li = i - g * gs.n;
        const bool use = hint && bad[g] == 0;
        ord[i] = g * gs.n + (use ? hint[(size_t)g * stride + li] : pos[i]);
    }
}
// seeds of the two reachability passes at the nodes' places: for the walks (0 / -1) and the sweeps (0 / none)
__global__ void k_reach_seeds(int n, const int* __restrict__ types, const int* __restrict__ ord, int* __restrict__ fwd,
                              int* __restrict__ rev, int* __restrict__ fwd_seed, int* __restrict__ rev_seed) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int o = ord[i], t = types[i];
        fwd[o] = t == PI_T ? 0 : -1;
        rev[o] = t == PO_T ? 0 : -1;
        fwd_seed[o] = t == PI_T ? 0 : SWEEP_NONE;
        rev_seed[o] = t == PO_T ? 0 : SWEEP_NONE;
    }
}

// one thread per graph: out[9 g ..]
__global__ void k_pen_finish(GraphSet gs, const int* __restrict__ counters, float* __restrict__ out9) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= gs.B) return;
    const int* c = counters + g * PEN_CTRS;
    float* out = out9 + 9 * g;
    if (gs.empty(g)) {   // no edge: every term is zero (constraints.py:264-269)
        for (int i = 0; i < 9; ++i) out[i] = 0.f;
        return;
    }
    // pi_in, po_out, backedge, selfloop, dup_fanin, indeg_mse, fanin_type, reach_violation, total
    float pi_in = (float)c[0], po_out = (float)c[1], back = (float)c[2], self = (float)c[3], ftype = (float)c[4],
          dup = (float)c[5], mse = (float)c[6], reach = (float)c[7];
    out[0] = pi_in; out[1] = po_out; out[2] = back; out[3] = self; out[4] = dup; out[5] = mse; out[6] = ftype;
    out[7] = reach;
    // summed in the reference's dict order, weights 1,1,2,3,1,1,1,0.5 (constraints.py:341-364)
    float total = 0.f;
    total += 1.0f * pi_in; total += 1.0f * po_out; total += 2.0f * back; total += 3.0f * self;
    total += 1.0f * dup; total += 1.0f * mse; total += 1.0f * ftype; total += 0.5f * reach;
    out[8] = total;
}

// per-node verdicts of the strict validator (constraints.py:398-462)
__global__ void k_val_nodes(int n, const int* __restrict__ types, const int* __restrict__ indeg,
                            const int* __restrict__ dptr, const int* __restrict__ dadj /*src by target, edge order*/,
                            unsigned char* __restrict__ ncode, int* __restrict__ pred2) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int t = types[i], d = indeg[i];
        const int b = dptr[i];
        const int f0 = d > 0 ? dadj[b] : -1, f1 = d > 1 ? dadj[b + 1] : -1;
        pred2[2 * i] = f0;
        pred2[2 * i + 1] = f1;
        unsigned char code = 0;
        if (t != PI_T && t != PO_T && t != AND_T) code = 1;
        else if (t == PO_T && d != 1) code = 2;
        else if (t == AND_T && d != 2) code = 3;
        else if (t == AND_T && f0 == f1) code = 4;
        ncode[i] = code;
    }
}

__global__ void k_val_summary(int n, const int* __restrict__ popped, const int* __restrict__ counters,
                              int* __restrict__ summary) {
    summary[0] = popped[0] != n;  // Kahn left nodes behind: a cycle (a self loop counts)
    summary[1] = counters[7];
}

// ------------------------------------------------------------------------------------------ resync
__global__ void k_rs_edges(GraphSet gs, const long long* __restrict__ ei, const long long* __restrict__ inv,
                           int* __restrict__ inv_cnt, int* __restrict__ indeg, unsigned* __restrict__ key_src,
                           int* __restrict__ ids) {
    const int M = gs.M(), N = gs.N();
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < M; e += gridDim.x * blockDim.x) {
        int g;
        long long u = -1, v = -1;
        const bool have = load_edge(gs, ei, e, g, u, v);
        ids[e] = e;
        const bool ok = have && in_range(u, v, gs.n);
        const int gv = g * gs.n + (int)v;
        if (have && v >= 0 && v < gs.n && inv && inv[e] == 1) atomicAdd(&inv_cnt[gv], 1);
        key_src[e] = ok ? (unsigned)(g * gs.n + (int)u) : (unsigned)N;
        if (ok) atomicAdd(&indeg[gv], 1);
    }
}
// cyclic[g] = 1 when the Kahn peel left a node of graph g behind
__global__ void k_rs_cyclic(GraphSet gs, const int* __restrict__ depth, int* __restrict__ cyclic) {
    const int N = gs.N();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
        if (depth[i] < 0) cyclic[i / gs.n] = 1;
}
__global__ void k_rs_finish(GraphSet gs, const int* __restrict__ cyclic, const int* __restrict__ depth,
                            const int* __restrict__ inv_cnt, long long* __restrict__ depth_out,
                            long long* __restrict__ inv_out) {
    const int N = gs.N();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        // networkx raises on a cycle and the depth of that graph falls back to zeros
        depth_out[i] = cyclic[i / gs.n] ? 0 : depth[i];
        inv_out[i] = inv_cnt[i];
    }
}

inline int g256(int64_t items) { return grid_for(items, 256, 4096); }

// types/pos for the N nodes of the set into (types, pos, scan[N+1]); `flags` and `temp` are scratch
int run_types_pos(hipStream_t s, const GraphSet& gs, const float* x, int* types, int* pos, int* small, Cnt4* flags,
                  Cnt4* scan, void* temp, size_t tb) {
    const int n = gs.N();
    hipLaunchKernelGGL(k_type_flags, dim3(g256(n + 1)), dim3(256), 0, s, n, x, types, flags);
    AIG_LAUNCH_CHECK();
    AIG_HIP_CHECK(rocprim::exclusive_scan(temp, tb, flags, scan, Cnt4{0, 0, 0, 0}, (size_t)(n + 1), Cnt4Plus(), s));
    if (small) AIG_HIP_CHECK(fill_words(small, 0xFFFFFFFFu, (size_t)3 * gs.B * sizeof(int), s));
    hipLaunchKernelGGL(k_pos, dim3(g256(n)), dim3(256), 0, s, gs, types, scan, pos, small);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

// CSR over the edge slots whose key (an endpoint's union id) is < N: ptr[N+1], adj = the other endpoint
int run_csr(hipStream_t s, const GraphSet& gs, int bits, unsigned* key, unsigned* skey, int* ids, int* sid,
            const long long* ei, int other_is_target, int* ptr, int* adj, void* temp, size_t tb,
            const int* relabel = nullptr) {
    const int N = gs.N(), M = gs.M();
    AIG_HIP_CHECK(rs::sort_pairs(temp, tb, key, skey, ids, sid, (size_t)M, bits, s));
    hipLaunchKernelGGL(k_segment_ptr<unsigned>, dim3(g256(N + 1)), dim3(256), 0, s, N, M, skey, 1u, ptr);
    hipLaunchKernelGGL(k_gather_adj, dim3(g256(M)), dim3(256), 0, s, gs, sid, ei, other_is_target, adj, relabel);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

// depth32[node] = breadth-first level from the PIs over the cleaned edges (-1: not reached), by the sweep over
// positions; seed / dpos: N ints of scratch each
int run_sweep_levels(hipStream_t s, const GraphSet& gs, const long long* ei, const int* types, const int* pos, int bits,
                     unsigned* key, unsigned* skey, int* val, int* sval, int* tptr, int* seed, int* dpos, int* depth32,
                     void* temp, size_t tb) {
    const int N = gs.N(), M = gs.M();
    hipLaunchKernelGGL(k_sweep_keys, dim3(g256(M)), dim3(256), 0, s, gs, ei, pos, key, val);
    AIG_LAUNCH_CHECK();
    AIG_HIP_CHECK(rs::sort_pairs(temp, tb, key, skey, val, sval, (size_t)M, bits, s));
    hipLaunchKernelGGL(k_segment_ptr<unsigned>, dim3(g256(N + 1)), dim3(256), 0, s, N, M, skey, 1u, tptr);
    hipLaunchKernelGGL(k_sweep_seed, dim3(g256(N)), dim3(256), 0, s, gs, types, pos, seed);
    hipLaunchKernelGGL(k_sweep<0>, dim3(gs.B), dim3(SWEEP_THREADS), 0, s, gs, SweepJob{tptr, sval, seed, dpos},
                       (const int*)nullptr);
    hipLaunchKernelGGL(k_sweep_out, dim3(g256(N)), dim3(256), 0, s, gs, pos, dpos, depth32);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

struct Temps {
    int kb;       // bits of a pair key
    int nb;       // bits of a node key (N itself marks an absent edge)
    size_t bytes; // scratch for the widest sort / scan
};
Temps temps_for(int64_t N, int64_t M) {
    Temps t;
    const unsigned long long n = (unsigned long long)(N > 0 ? N : 1);
    t.kb = bits_for(n * n + n);
    t.nb = bits_for(n);
    const size_t t64 = sort_temp<unsigned long long>(M, t.kb), t32 = sort_temp<unsigned>(M, 32),
                 tnb = sort_temp<unsigned>(M, t.nb), ts = scan_temp(N + 2);
    t.bytes = t64 > t32 ? t64 : t32;
    t.bytes = t.bytes > tnb ? t.bytes : tnb;
    t.bytes = t.bytes > ts ? t.bytes : ts;
    return t;
}

int check_set(const char* who, int64_t B, int64_t n, int64_t ecap) {
    AIG_REQUIRE(B >= 1 && n > 0 && ecap > 0, "%s: needs B >= 1, n > 0 and edge slots > 0 (got %lld, %lld, %lld)", who,
                (long long)B, (long long)n, (long long)ecap);
    AIG_REQUIRE(B * n < INT_MAX && B * ecap < INT_MAX, "%s: %lld graphs of %lld nodes / %lld edge slots exceed 2^31", who,
                (long long)B, (long long)n, (long long)ecap);
    return AIG_OK;
}

size_t enforce_ws(int64_t B, int64_t n, int64_t ecap) {
    const size_t N = (size_t)(B * n > 0 ? B * n : 1), M = (size_t)(B * ecap > 0 ? B * ecap : 1);
    const Temps t = temps_for((int64_t)N, (int64_t)M);
    return 2 * align_up((N + 1) * sizeof(Cnt4)) + 6 * align_up((N + 2) * 4) + align_up(N * sizeof(Pick)) +
           2 * align_up(M * 8) + 5 * align_up(M * 4) + align_up(M * 8) + align_up((N + 2) * 16) +
           align_up((size_t)(3 * B + 16) * 4) + align_up(t.bytes) + 1024;
}

int enforce(const GraphSet& gs, const float* x, const int64_t* edge_index, const int64_t* edge_attr, int depth_kind,
            const void* depth, int64_t depth_stride, int64_t* out_edge_index, int64_t cap, int64_t* out_edge_attr,
            int32_t* out_count, int64_t* node_depth, float* x_out, void* ws, size_t ws_bytes, void* stream,
            const char* who) {
    AIG_REQUIRE(x && edge_index && out_edge_index && out_edge_attr && out_count && x_out && ws, "%s: null pointer", who);
    AIG_REQUIRE(cap >= 2 * (int64_t)gs.n, "%s: output capacity %lld < 2n", who, (long long)cap);
    AIG_REQUIRE(depth_kind >= 0 && depth_kind <= 2 && (depth_kind == 0 || depth), "%s: bad depth argument", who);
    AIG_REQUIRE(depth_stride == 0 || depth_stride == gs.n, "%s: depth stride is 0 (shared) or n", who);
    if (ws_bytes < enforce_ws(gs.B, gs.n, gs.ecap))
        return fail(AIG_EWORKSPACE, "%s: workspace %zu < %zu", who, ws_bytes, enforce_ws(gs.B, gs.n, gs.ecap));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int N = gs.N(), M = gs.M();
    Carver c(ws);
    Cnt4* flags = c.take<Cnt4>(N + 1);
    Cnt4* scan = c.take<Cnt4>(N + 1);
    int* types = c.take<int>(N + 2);
    int* pos = c.take<int>(N + 2);
    int* vptr = c.take<int>(N + 2);
    int* sptr = c.take<int>(N + 2);
    int* pick_cnt = c.take<int>(N + 2);
    int* pick_off = c.take<int>(N + 2);
    Pick* picks = c.take<Pick>(N);
    unsigned long long* key64 = c.take<unsigned long long>(M);
    unsigned long long* skey64 = c.take<unsigned long long>(M);
    unsigned* key_src = c.take<unsigned>(M);
    unsigned* skey_src = c.take<unsigned>(M);
    int* ids = c.take<int>(M);
    int* sid = c.take<int>(M);
    int* adj = c.take<int>(M);
    long long* minv = c.take<long long>(M);
    int* small = c.take<int>(3 * gs.B + 16);
    int4* rec = c.take<int4>(N + 2);
    const Temps t = temps_for(N, M);
    size_t tb = t.bytes;
    void* temp = c.take<char>(tb);
    const long long* ei = reinterpret_cast<const long long*>(edge_index);
    const long long* inv = reinterpret_cast<const long long*>(edge_attr);

    int rc = run_types_pos(s, gs, x, types, pos, small, flags, scan, temp, tb);
    if (rc) return rc;
    hipLaunchKernelGGL(k_enf_keys, dim3(g256(M)), dim3(256), 0, s, gs, ei, pos, key64, key_src, ids);
    AIG_LAUNCH_CHECK();
    // by (target, source): runs of equal pairs are adjacent, first occurrence first (stable)
    AIG_HIP_CHECK(rs::sort_pairs(temp, tb, key64, skey64, ids, sid, (size_t)M, t.kb, s));
    hipLaunchKernelGGL(k_enf_merge, dim3(g256(M)), dim3(256), 0, s, M, skey64, sid, inv, minv);
    hipLaunchKernelGGL(k_segment_ptr<unsigned long long>, dim3(g256(N + 1)), dim3(256), 0, s, N, M, skey64,
                       (unsigned long long)N, vptr);
    hipLaunchKernelGGL(k_enf_select, dim3(g256(N + 1)), dim3(256), 0, s, gs, types, pos, small, scan, skey64, minv, vptr,
                       depth_kind, depth, (long long)depth_stride, picks, pick_cnt, x_out, x);
    AIG_LAUNCH_CHECK();
    AIG_HIP_CHECK(rocprim::exclusive_scan(temp, tb, pick_cnt, pick_off, 0, (size_t)(N + 1), rocprim::plus<int>(), s));
    hipLaunchKernelGGL(k_enf_write, dim3(g256(N)), dim3(256), 0, s, gs, (long long)cap, picks, pick_off,
                       reinterpret_cast<long long*>(out_edge_index), reinterpret_cast<long long*>(out_edge_attr),
                       out_count);
    AIG_LAUNCH_CHECK();
    if (!node_depth) return AIG_OK;  // the caller recomputes the depths anyway (sampling: resync follows)
    // node_depth: breadth-first level from the PIs over the cleaned (pre-repair) edges
    int* depth32 = pick_cnt;  // free again
    int* q0 = reinterpret_cast<int*>(flags);  // (N+1)*16 bytes each: room for N ints
    int* q1 = reinterpret_cast<int*>(scan);
    (void)adj; (void)rec;
    if ((rc = run_sweep_levels(s, gs, ei, types, pos, t.nb, key_src, skey_src, ids, sid, sptr, q0, q1, depth32, temp, tb)))
        return rc;
    hipLaunchKernelGGL(k_enf_depth_out, dim3(g256(N)), dim3(256), 0, s, gs, depth32, reinterpret_cast<long long*>(node_depth));
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

// The repair's breadth-first levels alone (the walk the repair ends with, for callers that take the repaired
// edges first and the levels later): levels from the PIs over the cleaned edges of the INPUT graph.
int clean_depth(const GraphSet& gs, const float* x, const int64_t* edge_index, int64_t* node_depth, void* ws,
                size_t ws_bytes, void* stream, const char* who) {
    AIG_REQUIRE(x && edge_index && node_depth && ws, "%s: null pointer", who);
    if (ws_bytes < enforce_ws(gs.B, gs.n, gs.ecap))
        return fail(AIG_EWORKSPACE, "%s: workspace %zu < %zu", who, ws_bytes, enforce_ws(gs.B, gs.n, gs.ecap));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int N = gs.N(), M = gs.M();
    Carver c(ws);
    Cnt4* flags = c.take<Cnt4>(N + 1);
    Cnt4* scan = c.take<Cnt4>(N + 1);
    int* types = c.take<int>(N + 2);
    int* pos = c.take<int>(N + 2);
    int* sptr = c.take<int>(N + 2);
    int* depth32 = c.take<int>(N + 2);
    unsigned long long* key64 = c.take<unsigned long long>(M);
    unsigned* key_src = c.take<unsigned>(M);
    unsigned* skey_src = c.take<unsigned>(M);
    int* ids = c.take<int>(M);
    int* sid = c.take<int>(M);
    int* adj = c.take<int>(M);
    int4* rec = c.take<int4>(N + 2);
    const Temps t = temps_for(N, M);
    size_t tb = t.bytes;
    void* temp = c.take<char>(tb);
    const long long* ei = reinterpret_cast<const long long*>(edge_index);
    int rc = run_types_pos(s, gs, x, types, pos, nullptr, flags, scan, temp, tb);
    if (rc) return rc;
    int* q0 = reinterpret_cast<int*>(flags);  // (N+1)*16 bytes each: room for N ints
    int* q1 = reinterpret_cast<int*>(scan);
    (void)key64; (void)adj; (void)rec;
    if ((rc = run_sweep_levels(s, gs, ei, types, pos, t.nb, key_src, skey_src, ids, sid, sptr, q0, q1, depth32, temp, tb)))
        return rc;
    hipLaunchKernelGGL(k_enf_depth_out, dim3(g256(N)), dim3(256), 0, s, gs, depth32, reinterpret_cast<long long*>(node_depth));
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

size_t graph_checks_ws(int64_t B, int64_t n, int64_t ecap) {
    const size_t N = (size_t)(B * n > 0 ? B * n : 1), M = (size_t)(B * ecap > 0 ? B * ecap : 1);
    const Temps t = temps_for((int64_t)N, (int64_t)M);
    return 2 * align_up((N + 1) * sizeof(Cnt4)) + 18 * align_up((N + 2) * 4) + 2 * align_up(M * 8) +
           8 * align_up(M * 4) + 2 * align_up((N + 2) * 16) + align_up((size_t)(B + 1) * PEN_CTRS * 4) +
           2 * align_up((size_t)(B + 16) * 4) + align_up(t.bytes) + 2048;
}

// shared body: counters (penalties, any B) and/or flags (validator, one graph)
// order_hint / hint_stride (penalties only, may be null): see k_ord_scatter.
int graph_checks(const GraphSet& gs, const float* x, const int64_t* edge_index, float* pen_out, unsigned char* eflag,
                 unsigned char* ncode, int* indeg_out, int* pred2, unsigned char* on_path, int* summary, void* ws,
                 size_t ws_bytes, void* stream, const char* who, const int32_t* order_hint = nullptr,
                 int64_t hint_stride = 0) {
    AIG_REQUIRE(x && edge_index && ws, "%s: null pointer", who);
    AIG_REQUIRE(gs.B == 1 || !ncode, "%s: the validator takes one graph at a time", who);
    if (ws_bytes < graph_checks_ws(gs.B, gs.n, gs.ecap)) return fail(AIG_EWORKSPACE, "%s: workspace too small", who);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int N = gs.N(), M = gs.M();
    Carver c(ws);
    Cnt4* flags = c.take<Cnt4>(N + 1);
    Cnt4* scan = c.take<Cnt4>(N + 1);
    int* types = c.take<int>(N + 2);
    int* pos = c.take<int>(N + 2);
    int* indeg = c.take<int>(N + 2);
    int* sptr = c.take<int>(N + 2);
    int* dptr = c.take<int>(N + 2);
    int* fwd = c.take<int>(N + 2);
    int* rev = c.take<int>(N + 2);
    int* kdeg = c.take<int>(N + 2);
    int* kdepth = c.take<int>(N + 2);
    int* qx = c.take<int>(N + 2);
    int* q2 = c.take<int>(N + 2);
    int* q3 = c.take<int>(N + 2);
    unsigned long long* key64 = c.take<unsigned long long>(M);
    unsigned long long* skey64 = c.take<unsigned long long>(M);
    unsigned* key_src = c.take<unsigned>(M);
    unsigned* key_dst = c.take<unsigned>(M);
    unsigned* skey32 = c.take<unsigned>(M);
    int* ids = c.take<int>(M);
    int* sid = c.take<int>(M);
    int* sadj = c.take<int>(M);
    int* dadj = c.take<int>(M);
    int* q1 = qx;                                                 // second queue (N entries)
    int* ord = c.take<int>(N + 2);
    int* inv = c.take<int>(N + 2);
    int* fwd_seed = c.take<int>(N + 2);
    int* fwd_sweep = c.take<int>(N + 2);
    int* rev_sweep = c.take<int>(N + 2);
    int* bad = c.take<int>(gs.B + 16);
    int* unordered = c.take<int>(gs.B + 16);
    int* counters = c.take<int>((size_t)(gs.B + 1) * PEN_CTRS);   // PEN_CTRS per graph, then [0] Kahn popped
    int* popped = counters + (size_t)gs.B * PEN_CTRS;
    int4* rec0 = c.take<int4>(N + 2);
    int4* rec1 = c.take<int4>(N + 2);
    const Temps t = temps_for(N, M);
    size_t tb = t.bytes;
    void* temp = c.take<char>(tb);
    const long long* ei = reinterpret_cast<const long long*>(edge_index);

    int rc = run_types_pos(s, gs, x, types, pos, nullptr, flags, scan, temp, tb);
    if (rc) return rc;
    AIG_HIP_CHECK(fill_words(counters, 0u, (size_t)(gs.B + 1) * PEN_CTRS * sizeof(int), s));
    AIG_HIP_CHECK(fill_words(indeg, 0u, (size_t)(N + 2) * sizeof(int), s));
    // The penalties' reachability from the PIs and back from the POs (constraints.py:310-339) is one pass over the
    // nodes in an order in which every edge runs upward, and one in reverse - when such an order is at hand: the
    // caller's hint (the generator wires every target to nodes before it in its processing order) or the
    // PI / AND / PO position (what a repaired graph obeys).  Whether a graph obeys the order is found on the device;
    // the graphs that do not are walked level by level as before.  Lists, seeds and verdicts then live at the
    // nodes' places in that order.  The validator keeps node ids (its other outputs are indexed by them).
    const bool use_ord = ncode == nullptr;
    if (use_ord) {
        AIG_HIP_CHECK(fill_words(bad, 0u, (size_t)(gs.B + 16) * sizeof(int), s));
        AIG_HIP_CHECK(fill_words(unordered, 0u, (size_t)(gs.B + 16) * sizeof(int), s));
        if (order_hint) {
            hipLaunchKernelGGL(k_ord_scatter, dim3(g256(N)), dim3(256), 0, s, gs, order_hint, (long long)hint_stride, inv);
            hipLaunchKernelGGL(k_ord_check, dim3(g256(N)), dim3(256), 0, s, gs, order_hint, (long long)hint_stride, inv, bad);
        }
        hipLaunchKernelGGL(k_ord_pick, dim3(g256(N)), dim3(256), 0, s, gs, order_hint, (long long)hint_stride, pos, bad, ord);
        AIG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_pen_edges, dim3(g256(M)), dim3(256), 0, s, gs, ei, types, pos, counters, indeg, key64, key_src,
                       key_dst, ids, eflag, use_ord ? (const int*)ord : nullptr, unordered);
    AIG_LAUNCH_CHECK();
    AIG_HIP_CHECK(rs::sort_pairs(temp, tb, key64, skey64, ids, sid, (size_t)M, t.kb, s));
    hipLaunchKernelGGL(k_pen_dups, dim3(g256(M)), dim3(256), 0, s, gs, skey64, sid, counters, eflag);
    AIG_LAUNCH_CHECK();
    // forward reachability from the PIs, backward from the POs
    const int* relabel = use_ord ? ord : nullptr;
    if ((rc = run_csr(s, gs, t.nb, key_src, skey32, ids, sid, ei, 1, sptr, sadj, temp, tb, relabel))) return rc;
    if ((rc = run_csr(s, gs, t.nb, key_dst, skey32, ids, sid, ei, 0, dptr, dadj, temp, tb, relabel))) return rc;
    int* q0 = reinterpret_cast<int*>(flags);
    TravJobs jobs{};  // the two reachability walks are independent: one launch, one block each
    jobs.j[0] = TravJob{sptr, sadj, nullptr, fwd, q0, q1, nullptr};
    jobs.j[1] = TravJob{dptr, dadj, nullptr, rev, q2, q3, nullptr};
    if (use_ord) {
        hipLaunchKernelGGL(k_reach_seeds, dim3(g256(N)), dim3(256), 0, s, N, types, ord, fwd, rev, fwd_seed, rev_sweep);
        // both passes run along the lists by target (a node's predecessors): forward pulls, places ascending;
        // backward pushes, places descending (rev_sweep is preset with the seeds).  One launch.
        hipLaunchKernelGGL(k_reach_pair, dim3(gs.B, 2), dim3(SWEEP_THREADS), 0, s, gs, SweepJob{dptr, dadj, fwd_seed, fwd_sweep},
                           (const int*)dptr, (const int*)dadj, rev_sweep, (const int*)unordered);
        AIG_LAUNCH_CHECK();
        jobs.gate = unordered;      // the walks take the graphs the sweeps left alone
    } else {
        hipLaunchKernelGGL(k_seed_depth, dim3(g256(N)), dim3(256), 0, s, N, types, PI_T, fwd);
        hipLaunchKernelGGL(k_seed_depth, dim3(g256(N)), dim3(256), 0, s, N, types, PO_T, rev);
    }
    if ((rc = run_traverse<0>(s, N, jobs, 2, rec0, rec1, gs.B))) return rc;
    hipLaunchKernelGGL(k_pen_nodes, dim3(g256(N)), dim3(256), 0, s, gs, types, indeg, fwd, rev, counters, on_path, relabel,
                       (const int*)unordered, (const int*)fwd_sweep, (const int*)rev_sweep);
    AIG_LAUNCH_CHECK();
    if (pen_out) hipLaunchKernelGGL(k_pen_finish, dim3((gs.B + 63) / 64), dim3(64), 0, s, gs, counters, pen_out);
    if (ncode) {
        hipLaunchKernelGGL(k_val_nodes, dim3(g256(N)), dim3(256), 0, s, N, types, indeg, dptr, dadj, ncode, pred2);
        AIG_HIP_CHECK(hipMemcpyAsync(kdeg, indeg, (size_t)N * sizeof(int), hipMemcpyDeviceToDevice, s));
        if (indeg_out) AIG_HIP_CHECK(hipMemcpyAsync(indeg_out, indeg, (size_t)N * sizeof(int), hipMemcpyDeviceToDevice, s));
        TravJobs kj{};
        kj.j[0] = TravJob{sptr, sadj, kdeg, kdepth, q0, q1, popped};
        if ((rc = run_traverse<1>(s, N, kj, 1, rec0, nullptr))) return rc;
        hipLaunchKernelGGL(k_val_summary, dim3(1), dim3(1), 0, s, N, popped, counters, summary);
    }
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

size_t resync_ws(int64_t B, int64_t n, int64_t ecap) {
    const size_t N = (size_t)(B * n > 0 ? B * n : 1), M = (size_t)(B * ecap > 0 ? B * ecap : 1);
    const size_t t32 = sort_temp<unsigned>((int64_t)M, 32), tnb = sort_temp<unsigned>((int64_t)M, bits_for(N));
    return 7 * align_up((N + 2) * 4) + 5 * align_up(M * 4) + align_up((N + 2) * 16) + align_up((size_t)(B + 16) * 4) +
           align_up(t32 > tnb ? t32 : tnb) + 1024;
}

// `x` (may be null): node features whose first column is the node type.  With it, the levels of every graph whose
// edges all run from a lower to a higher position in the PI / AND / PO order (what a repaired graph looks like)
// come from one sweep over the positions (k_sweep<1>) instead of the Kahn walk; the walk only runs for the graphs
// where that does not hold (both are launched, each leaves the other's graphs alone).
size_t resync_typed_ws(int64_t B, int64_t n, int64_t ecap) {
    const size_t N = (size_t)(B * n > 0 ? B * n : 1), M = (size_t)(B * ecap > 0 ? B * ecap : 1);
    const size_t t32 = sort_temp<unsigned>((int64_t)M, 32), ts = scan_temp((int64_t)N + 2);
    return resync_ws(B, n, ecap) + 2 * align_up((N + 1) * sizeof(Cnt4)) + 3 * align_up((N + 2) * 4) + 4 * align_up(M * 4) +
           align_up((size_t)(B + 16) * 4) + align_up(t32 > ts ? t32 : ts) + 1024;
}

int resync(const GraphSet& gs, const float* x, const int64_t* edge_index, const int64_t* edge_attr, int64_t* inv_cnt,
           int64_t* depth, void* ws, size_t ws_bytes, void* stream, const char* who) {
    AIG_REQUIRE(edge_index && inv_cnt && depth && ws, "%s: null pointer", who);
    if (ws_bytes < (x ? resync_typed_ws(gs.B, gs.n, gs.ecap) : resync_ws(gs.B, gs.n, gs.ecap)))
        return fail(AIG_EWORKSPACE, "%s: workspace too small", who);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int N = gs.N(), M = gs.M();
    Carver c(ws);
    int* icnt = c.take<int>(N + 2);
    int* indeg = c.take<int>(N + 2);
    int* sptr = c.take<int>(N + 2);
    int* d32 = c.take<int>(N + 2);
    int* q0 = c.take<int>(N + 2);
    int* q1 = c.take<int>(N + 2);
    (void)c.take<int>(N + 2);      // (room the workspace size still counts)
    unsigned* key = c.take<unsigned>(M);
    unsigned* skey = c.take<unsigned>(M);
    int* ids = c.take<int>(M);
    int* sid = c.take<int>(M);
    int* adj = c.take<int>(M);
    int4* rec = c.take<int4>(N + 2);
    int* cyclic = c.take<int>(gs.B + 16);
    size_t tb = sort_temp<unsigned>(M, 32);
    {
        const size_t tnb = sort_temp<unsigned>(M, bits_for((unsigned long long)N));
        if (tnb > tb) tb = tnb;
    }
    void* temp = c.take<char>(tb);
    const long long* ei = reinterpret_cast<const long long*>(edge_index);
    AIG_HIP_CHECK(fill_words(icnt, 0u, (size_t)(N + 2) * 4, s));
    AIG_HIP_CHECK(fill_words(indeg, 0u, (size_t)(N + 2) * 4, s));
    AIG_HIP_CHECK(fill_words(cyclic, 0u, (size_t)(gs.B + 16) * 4, s));
    hipLaunchKernelGGL(k_rs_edges, dim3(g256(M)), dim3(256), 0, s, gs, ei, reinterpret_cast<const long long*>(edge_attr),
                       icnt, indeg, key, ids);
    AIG_LAUNCH_CHECK();
    int rc;
    const int* gate = nullptr;
    if (x) {
        Cnt4* flags = c.take<Cnt4>(N + 1);
        Cnt4* scan = c.take<Cnt4>(N + 1);
        int* types = c.take<int>(N + 2);
        int* pos = c.take<int>(N + 2);
        int* tptr = c.take<int>(N + 2);
        unsigned* tkey = c.take<unsigned>(M);
        unsigned* tskey = c.take<unsigned>(M);
        int* tval = c.take<int>(M);
        int* tsval = c.take<int>(M);
        int* unordered = c.take<int>(gs.B + 16);
        AIG_HIP_CHECK(fill_words(unordered, 0u, (size_t)(gs.B + 16) * 4, s));
        size_t tb2 = scan_temp(N + 2);
        if (tb2 < tb) tb2 = tb;
        void* temp2 = c.take<char>(tb2);
        if ((rc = run_types_pos(s, gs, x, types, pos, nullptr, flags, scan, temp2, tb2))) return rc;
        hipLaunchKernelGGL(k_rs_sweep_keys, dim3(g256(M)), dim3(256), 0, s, gs, ei, pos, tkey, tval, unordered);
        AIG_LAUNCH_CHECK();
        AIG_HIP_CHECK(rs::sort_pairs(temp2, tb2, tkey, tskey, tval, tsval, (size_t)M, bits_for((unsigned long long)N), s));
        hipLaunchKernelGGL(k_segment_ptr<unsigned>, dim3(g256(N + 1)), dim3(256), 0, s, N, M, tskey, 1u, tptr);
        int* dpos = reinterpret_cast<int*>(flags);   // free again: (N+1)*16 bytes
        hipLaunchKernelGGL(k_sweep<1>, dim3(gs.B), dim3(SWEEP_THREADS), 0, s, gs, SweepJob{tptr, tsval, nullptr, dpos},
                           (const int*)unordered);
        hipLaunchKernelGGL(k_rs_sweep_out, dim3(g256(N)), dim3(256), 0, s, gs, pos, dpos, unordered, d32);
        AIG_LAUNCH_CHECK();
        gate = unordered;
    }
    rc = run_csr(s, gs, bits_for((unsigned long long)N), key, skey, ids, sid, ei, 1, sptr, adj, temp, tb);
    if (rc) return rc;
    TravJobs jobs{};
    jobs.j[0] = TravJob{sptr, adj, indeg, d32, q0, q1, nullptr};
    jobs.gate = gate;
    if ((rc = run_traverse<1>(s, N, jobs, 1, rec, nullptr, gs.B))) return rc;
    hipLaunchKernelGGL(k_rs_cyclic, dim3(g256(N)), dim3(256), 0, s, gs, d32, cyclic);
    hipLaunchKernelGGL(k_rs_finish, dim3(g256(N)), dim3(256), 0, s, gs, cyclic, d32, icnt,
                       reinterpret_cast<long long*>(depth), reinterpret_cast<long long*>(inv_cnt));
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

// edge slots of B graphs -> the padded minibatch the critic takes (data_process.py pad_data, then the
// concatenation with node offsets): graph g owns columns [g max_edges, (g+1) max_edges), its first
// min(count, max_edges) edges with union ids, the rest -1 shifted like every other entry
__global__ void k_pad_batch(GraphSet gs, const long long* __restrict__ ei, const long long* __restrict__ ea,
                            int max_edges, long long* __restrict__ out_ei, long long* __restrict__ out_ea) {
    const long long cols = (long long)gs.B * max_edges;
    for (long long c = blockIdx.x * (long long)blockDim.x + threadIdx.x; c < cols; c += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(c / max_edges), j = (int)(c - (long long)g * max_edges);
        const int have = gs.ecount ? min(gs.ecount[g], gs.ecap) : gs.ecap;
        long long u = -1, v = -1, a = 0;
        if (j < have) {
            const size_t row = (size_t)(2 * g) * gs.ecap + j;
            u = ei[row];
            v = ei[row + gs.ecap];
            a = ea ? ea[(size_t)g * gs.ecap + j] : 0;
        }
        const long long shift = (long long)g * gs.n;
        out_ei[c] = u + shift;
        out_ei[cols + c] = v + shift;
        out_ea[c] = a;
    }
}

}  // namespace
}  // namespace aig

using namespace aig;

// ============================================================================================ enforce
extern "C" size_t aig_enforce_ws_bytes(int64_t n, int64_t E) { return enforce_ws(1, n, E); }

extern "C" int aig_enforce(int64_t n, int64_t E, const float* x, const int64_t* edge_index, const int64_t* edge_attr,
                           int depth_kind, const void* depth, int64_t* out_edge_index, int64_t cap,
                           int64_t* out_edge_attr, int32_t* out_count, int64_t* node_depth, float* x_out, void* ws,
                           size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n > 0 && n < INT_MAX && E > 0 && E < INT_MAX, "aig_enforce: needs n > 0 and E > 0 (got %lld, %lld)",
                (long long)n, (long long)E);
    const GraphSet gs{1, (int)n, (int)E, nullptr};
    return enforce(gs, x, edge_index, edge_attr, depth_kind, depth, 0, out_edge_index, cap, out_edge_attr, out_count,
                   node_depth, x_out, ws, ws_bytes, stream, "aig_enforce");
}

extern "C" size_t aig_enforce_batch_ws_bytes(int64_t B, int64_t n, int64_t ecap) { return enforce_ws(B, n, ecap); }

extern "C" int aig_enforce_batch(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const float* x,
                                 const int64_t* edge_index, const int64_t* edge_attr, int depth_kind, const void* depth,
                                 int64_t depth_stride, int64_t* out_edge_index, int64_t cap, int64_t* out_edge_attr,
                                 int32_t* out_count, int64_t* node_depth, float* x_out, void* ws, size_t ws_bytes,
                                 void* stream) {
    int rc = check_set("aig_enforce_batch", B, n, ecap);
    if (rc) return rc;
    const GraphSet gs{(int)B, (int)n, (int)ecap, ecount};
    return enforce(gs, x, edge_index, edge_attr, depth_kind, depth, depth_stride, out_edge_index, cap, out_edge_attr,
                   out_count, node_depth, x_out, ws, ws_bytes, stream, "aig_enforce_batch");
}

extern "C" int aig_clean_depth_batch(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const float* x,
                                     const int64_t* edge_index, int64_t* node_depth, void* ws, size_t ws_bytes,
                                     void* stream) {
    int rc = check_set("aig_clean_depth_batch", B, n, ecap);
    if (rc) return rc;
    const GraphSet gs{(int)B, (int)n, (int)ecap, ecount};
    return clean_depth(gs, x, edge_index, node_depth, ws, ws_bytes, stream, "aig_clean_depth_batch");
}

// ============================================================================================ penalties / validate
extern "C" size_t aig_penalties_ws_bytes(int64_t n, int64_t E) { return graph_checks_ws(1, n, E); }

extern "C" int aig_penalties(int64_t n, int64_t E, const float* x, const int64_t* edge_index, float* out9, void* ws,
                             size_t ws_bytes, void* stream) {
    AIG_REQUIRE(out9, "aig_penalties: null output");
    AIG_REQUIRE(n > 0 && n < INT_MAX && E > 0 && E < INT_MAX, "aig_penalties: needs n > 0 and E > 0 (got %lld, %lld)",
                (long long)n, (long long)E);
    const GraphSet gs{1, (int)n, (int)E, nullptr};
    return graph_checks(gs, x, edge_index, out9, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes,
                        stream, "aig_penalties");
}

extern "C" size_t aig_penalties_batch_ws_bytes(int64_t B, int64_t n, int64_t ecap) { return graph_checks_ws(B, n, ecap); }

extern "C" int aig_penalties_batch(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const float* x,
                                   const int64_t* edge_index, float* out9, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(out9, "aig_penalties_batch: null output");
    int rc = check_set("aig_penalties_batch", B, n, ecap);
    if (rc) return rc;
    const GraphSet gs{(int)B, (int)n, (int)ecap, ecount};
    return graph_checks(gs, x, edge_index, out9, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes,
                        stream, "aig_penalties_batch");
}

extern "C" int aig_penalties_batch_hinted(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const float* x,
                                          const int64_t* edge_index, const int32_t* order_hint, int64_t hint_stride,
                                          float* out9, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(out9, "aig_penalties_batch_hinted: null output");
    AIG_REQUIRE(hint_stride == 0 || hint_stride == n, "aig_penalties_batch_hinted: the hint's stride is 0 (shared) or n");
    int rc = check_set("aig_penalties_batch_hinted", B, n, ecap);
    if (rc) return rc;
    const GraphSet gs{(int)B, (int)n, (int)ecap, ecount};
    return graph_checks(gs, x, edge_index, out9, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes,
                        stream, "aig_penalties_batch_hinted", order_hint, hint_stride);
}

extern "C" size_t aig_validate_ws_bytes(int64_t n, int64_t E) { return graph_checks_ws(1, n, E); }

extern "C" int aig_validate(int64_t n, int64_t E, const float* x, const int64_t* edge_index, uint8_t* eflag,
                            uint8_t* ncode, int32_t* indeg, int32_t* pred2, uint8_t* on_path, int32_t* summary,
                            void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(eflag && ncode && indeg && pred2 && on_path && summary, "aig_validate: null output");
    AIG_REQUIRE(n > 0 && n < INT_MAX && E > 0 && E < INT_MAX, "aig_validate: needs n > 0 and E > 0 (got %lld, %lld)",
                (long long)n, (long long)E);
    const GraphSet gs{1, (int)n, (int)E, nullptr};
    return graph_checks(gs, x, edge_index, nullptr, eflag, ncode, indeg, pred2, on_path, summary, ws, ws_bytes, stream,
                        "aig_validate");
}

// ============================================================================================ resync
extern "C" size_t aig_resync_ws_bytes(int64_t n, int64_t E) { return resync_ws(1, n, E); }

extern "C" int aig_resync(int64_t n, int64_t E, const int64_t* edge_index, const int64_t* edge_attr,
                          int64_t* inv_cnt, int64_t* depth, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n > 0 && n < INT_MAX && E > 0 && E < INT_MAX, "aig_resync: needs n > 0 and E > 0");
    const GraphSet gs{1, (int)n, (int)E, nullptr};
    return resync(gs, nullptr, edge_index, edge_attr, inv_cnt, depth, ws, ws_bytes, stream, "aig_resync");
}

extern "C" size_t aig_resync_batch_ws_bytes(int64_t B, int64_t n, int64_t ecap) { return resync_ws(B, n, ecap); }

extern "C" int aig_resync_batch(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const int64_t* edge_index,
                                const int64_t* edge_attr, int64_t* inv_cnt, int64_t* depth, void* ws, size_t ws_bytes,
                                void* stream) {
    int rc = check_set("aig_resync_batch", B, n, ecap);
    if (rc) return rc;
    const GraphSet gs{(int)B, (int)n, (int)ecap, ecount};
    return resync(gs, nullptr, edge_index, edge_attr, inv_cnt, depth, ws, ws_bytes, stream, "aig_resync_batch");
}

// ============================================================================================ padded minibatch
extern "C" size_t aig_resync_typed_ws_bytes(int64_t n, int64_t E) { return resync_typed_ws(1, n, E); }

extern "C" int aig_resync_typed(int64_t n, int64_t E, const float* x, const int64_t* edge_index, const int64_t* edge_attr,
                                int64_t* inv_cnt, int64_t* depth, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_set("aig_resync_typed", 1, n, E);
    if (rc) return rc;
    AIG_REQUIRE(x, "aig_resync_typed: null pointer");
    const GraphSet gs{1, (int)n, (int)E, nullptr};
    return resync(gs, x, edge_index, edge_attr, inv_cnt, depth, ws, ws_bytes, stream, "aig_resync_typed");
}

extern "C" size_t aig_resync_typed_batch_ws_bytes(int64_t B, int64_t n, int64_t ecap) { return resync_typed_ws(B, n, ecap); }

extern "C" int aig_resync_typed_batch(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const float* x,
                                      const int64_t* edge_index, const int64_t* edge_attr, int64_t* inv_cnt, int64_t* depth,
                                      void* ws, size_t ws_bytes, void* stream) {
    int rc = check_set("aig_resync_typed_batch", B, n, ecap);
    if (rc) return rc;
    AIG_REQUIRE(x, "aig_resync_typed_batch: null pointer");
    const GraphSet gs{(int)B, (int)n, (int)ecap, ecount};
    return resync(gs, x, edge_index, edge_attr, inv_cnt, depth, ws, ws_bytes, stream, "aig_resync_typed_batch");
}

extern "C" int aig_pad_batch(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const int64_t* edge_index,
                             const int64_t* edge_attr, int64_t max_edges, int64_t* out_edge_index,
                             int64_t* out_edge_attr, void* stream) {
    int rc = check_set("aig_pad_batch", B, n, ecap);
    if (rc) return rc;
    AIG_REQUIRE(max_edges > 0 && B * max_edges < INT_MAX, "aig_pad_batch: bad max_edges %lld", (long long)max_edges);
    AIG_REQUIRE(edge_index && out_edge_index && out_edge_attr, "aig_pad_batch: null pointer");
    const GraphSet gs{(int)B, (int)n, (int)ecap, ecount};
    hipLaunchKernelGGL(k_pad_batch, dim3(g256(B * max_edges)), dim3(256), 0, static_cast<hipStream_t>(stream), gs,
                       reinterpret_cast<const long long*>(edge_index), reinterpret_cast<const long long*>(edge_attr),
                       (int)max_edges, reinterpret_cast<long long*>(out_edge_index),
                       reinterpret_cast<long long*>(out_edge_attr));
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

// ------------------------------------------------------------------------------------------ the sort, on its own
extern "C" size_t aig_sort_pairs_ws_bytes(int64_t items, int key_bytes, int bits) {
    const size_t m = (size_t)(items > 0 ? items : 1);
    return key_bytes == 8 ? rs::temp_bytes<unsigned long long, int>(m, bits) : rs::temp_bytes<unsigned, int>(m, bits);
}

extern "C" int aig_sort_pairs(int64_t items, int key_bytes, int bits, const void* keys_in, void* keys_out,
                              const int32_t* values_in, int32_t* values_out, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(key_bytes == 4 || key_bytes == 8, "aig_sort_pairs: keys are 4 or 8 bytes wide (got %d)", key_bytes);
    AIG_REQUIRE(items >= 0 && items < INT_MAX, "aig_sort_pairs: bad item count %lld", (long long)items);
    AIG_REQUIRE(bits >= 0 && bits <= 8 * key_bytes, "aig_sort_pairs: bad bit count %d", bits);
    if (items == 0) return AIG_OK;
    AIG_REQUIRE(keys_in && keys_out && values_in && values_out && ws, "aig_sort_pairs: null pointer");
    AIG_REQUIRE(keys_in != keys_out && values_in != values_out, "aig_sort_pairs: input and output must be different buffers");
    if (ws_bytes < aig_sort_pairs_ws_bytes(items, key_bytes, bits)) return fail(AIG_EWORKSPACE, "aig_sort_pairs: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (key_bytes == 8) {
        AIG_HIP_CHECK(rs::sort_pairs(ws, ws_bytes, static_cast<const unsigned long long*>(keys_in),
                                     static_cast<unsigned long long*>(keys_out), values_in, values_out, (size_t)items, bits, s));
    } else {
        AIG_HIP_CHECK(rs::sort_pairs(ws, ws_bytes, static_cast<const unsigned*>(keys_in), static_cast<unsigned*>(keys_out),
                                     values_in, values_out, (size_t)items, bits, s));
    }
    return AIG_OK;
}
