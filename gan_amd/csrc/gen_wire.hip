// Generator wiring kernels: candidate scores, tie-exact top-k select, edge assembly.
#include <climits>

#include "common.h"

namespace aig {
namespace {

constexpr int H = AIG_HIDDEN;

// Several graphs of n nodes in one launch: blockIdx.y is the graph, and every per-graph array advances by its
// stride (in elements) from one graph to the next - 0 for an array all graphs share (the plan and the node
// features when the graphs are samples of one template).  A single graph is a grid of height 1.
struct GenSet {
    long long node;     // order, rank, want, cnt, fb
    long long eoff;     // eoff (n + 1 entries per graph)
    long long cand;     // cand (n K entries per graph)
    long long x;        // x (2 n floats per graph)
    long long proj;     // rows of the projection planes per graph
};

// ------------------------------------------------------------------------------------------ scores
// One wave per slot; lane i scores candidate i (and i+64, ... for wider windows):
//   logit = b2 + sum_j w2[j] * relu(A[cand][j] + B[target][j]),   j ascending.
// The target row and w2 are wave-uniform, so they come through scalar loads; every candidate row
// runs the same instruction sequence, hence equal candidate rows give bit-equal logits.
__global__ __launch_bounds__(256) void k_score(int n, int K, const int* __restrict__ order,
                                               const int* __restrict__ cnt, const int* __restrict__ cand,
                                               const float* __restrict__ A, const float* __restrict__ B,
                                               const float* __restrict__ w2, const float* __restrict__ b2,
                                               float* __restrict__ logits, GenSet gs) {
    const int lane = threadIdx.x & 63;
    int r = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (r >= n) return;
    {
        const long long g = blockIdx.y;
        order += g * gs.node; cnt += g * gs.node; cand += g * gs.cand;
        A += g * gs.proj * H; B += g * gs.proj * H; logits += g * (long long)n * K;
    }
    const int c = cnt[r];
    if (c <= 0) return;
    const int v = __builtin_amdgcn_readfirstlane(order[r]);
    const float4* Bv = reinterpret_cast<const float4*>(B + static_cast<size_t>(v) * H);
    const float4* W = reinterpret_cast<const float4*>(w2);
    const float bias = b2[0];
    const size_t row0 = static_cast<size_t>(r) * K;
    for (int base = 0; base < c; base += 64) {
        const int i = base + lane;
        const bool active = i < c;
        const int ci = cand[row0 + (active ? i : 0)];
        const float4* Ar = reinterpret_cast<const float4*>(A + static_cast<size_t>(ci) * H);
        float acc = 0.f;
#pragma unroll 8
        for (int j = 0; j < H / 4; ++j) {
            float4 a = Ar[j], b = Bv[j], w = W[j];
            acc = fmaf(fmaxf(a.x + b.x, 0.f), w.x, acc);
            acc = fmaf(fmaxf(a.y + b.y, 0.f), w.y, acc);
            acc = fmaf(fmaxf(a.z + b.z, 0.f), w.z, acc);
            acc = fmaf(fmaxf(a.w + b.w, 0.f), w.w, acc);
        }
        if (active) logits[row0 + i] = acc + bias;
    }
}

// ------------------------------------------------------------------------------------------ select
// ATen's CPU top-k comparator (largest=True): NaN ranks first, then the larger value.
__device__ __forceinline__ bool gt(float x, float y) { return (isnan(x) && !isnan(y)) || (x > y); }

// Per-thread view of a row while std::nth_element's steps are replayed on it: position p holds the original index
// idx[p] AND its value val[p] - both move together, so that a scan step is ONE LDS read (values left in place and
// looked up through the index made every comparison two dependent reads, and a row's replay is a serial walk of a
// few hundred of them).  Both live in LDS, laid out [element][thread] so that neighbouring threads touch neighbouring
// banks.
template <int STRIDE>
struct RowView {
    unsigned char* idx;  // idx[p * STRIDE]
    float* val;          // val[p * (STRIDE + 1)]  (padded against the transposing fill)
    __device__ __forceinline__ int I(int p) const { return idx[p * STRIDE]; }
    __device__ __forceinline__ float V(int p) const { return val[p * (STRIDE + 1)]; }
    __device__ __forceinline__ void set(int p, int i, float v) {
        idx[p * STRIDE] = static_cast<unsigned char>(i);
        val[p * (STRIDE + 1)] = v;
    }
    __device__ __forceinline__ void move(int to, int from) { set(to, I(from), V(from)); }
    __device__ __forceinline__ void swap(int p, int q) {
        const int a = I(p), b = I(q);
        const float va = V(p), vb = V(q);
        set(p, b, vb);
        set(q, a, va);
    }
};

// libstdc++ heap primitives on positions [first, first+len), the value in hand carried as (original index, value)
template <int S>
__device__ void push_heap_(RowView<S>& q, int first, int hole, int top, int vi, float vv) {
    int parent = (hole - 1) / 2;
    while (hole > top && gt(q.V(first + parent), vv)) {
        q.move(first + hole, first + parent);
        hole = parent;
        parent = (hole - 1) / 2;
    }
    q.set(first + hole, vi, vv);
}
template <int S>
__device__ void adjust_heap_(RowView<S>& q, int first, int hole, int len, int vi, float vv) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (gt(q.V(first + child), q.V(first + child - 1))) --child;
        q.move(first + hole, first + child);
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        q.move(first + hole, first + child - 1);
        hole = child - 1;
    }
    push_heap_(q, first, hole, top, vi, vv);
}
template <int S>
__device__ void heap_select_(RowView<S>& q, int first, int middle, int last) {
    const int len = middle - first;
    if (len >= 2) {
        int parent = (len - 2) / 2;
        while (true) {
            const int vi = q.I(first + parent);
            const float vv = q.V(first + parent);
            adjust_heap_(q, first, parent, len, vi, vv);
            if (parent == 0) break;
            --parent;
        }
    }
    for (int i = middle; i < last; ++i) {
        if (gt(q.V(i), q.V(first))) {  // pop_heap(first, middle, i)
            const int vi = q.I(i);
            const float vv = q.V(i);
            q.move(i, first);
            adjust_heap_(q, first, 0, len, vi, vv);
        }
    }
}

// std::nth_element(first=0, nth, last=c) of libstdc++: introselect with median-of-three pivots,
// a depth limit of 2*floor(log2 c) that falls back to heap-select, and a final insertion sort on
// ranges of at most three elements.
template <int S>
__device__ void nth_element_(RowView<S>& q, int nth, int c) {
    int first = 0, last = c;
    int depth = 2 * (31 - __clz(c));
    while (last - first > 3) {
        if (depth == 0) {
            heap_select_(q, first, nth + 1, last);
            q.swap(first, nth);
            return;
        }
        --depth;
        // median of (first+1, mid, last-1) moved to first
        const int a = first + 1, b = first + (last - first) / 2, cpos = last - 1;
        const float va = q.V(a), vb = q.V(b), vc = q.V(cpos);
        int m;
        if (gt(va, vb)) m = gt(vb, vc) ? b : (gt(va, vc) ? cpos : a);
        else m = gt(va, vc) ? a : (gt(vb, vc) ? cpos : b);
        q.swap(first, m);
        // unguarded partition of [first+1, last) around the pivot at first
        const float pv = q.V(first);
        int lo = first + 1, hi = last;
        while (true) {
            while (gt(q.V(lo), pv)) ++lo;
            --hi;
            while (gt(pv, q.V(hi))) --hi;
            if (!(lo < hi)) break;
            q.swap(lo, hi);
            ++lo;
        }
        if (lo <= nth) first = lo; else last = lo;
    }
    // insertion sort of [first, last)
    for (int i = first + 1; i < last; ++i) {
        const int vi = q.I(i);
        const float vv = q.V(i);
        if (gt(vv, q.V(first))) {
            for (int p = i; p > first; --p) q.move(p, p - 1);
            q.set(first, vi, vv);
        } else {
            int p = i;
            while (gt(vv, q.V(p - 1))) {
                q.move(p, p - 1);
                --p;
            }
            q.set(p, vi, vv);
        }
    }
}

// One thread per row (fused != 0: rows with k*64 <= c already have their picks, from k_score_tiled).  Rows with
// k*64 <= c replay std::partial_sort, whose heap of k <= 2 entries
// fits in registers and whose scan never re-reads a displaced element; shorter rows (c < 128)
// replay std::nth_element on an LDS copy.
constexpr int SEL_THREADS = 128;
constexpr int SEL_CAP = 127;  // nth_element rows have c < 64*k <= 128; 127 lets two blocks share a CU

// T rows (threads) a block: 128 - two full waves - down to 8, an eighth of one.  A row's replay is a serial walk whose
// branches differ from row to row, so a wave spends the SUM of its rows' walks: the fewer rows there are in all, the
// fewer a wave is given (launch_select), as long as the GPU has waves to spare.
// Scores by class instead of an array of logits (k_score_classes' values, never stored): the value of entry i of
// slot r is T[cls[cand[r*ld + i]] * C + cls[order[r]]].  T == nullptr: the kernel reads `logits`.
struct ClassScores {
    const int* cand;
    const int* cls;
    const int* order;
    const float* T;
    int C;
};

template <int T, bool CLS = false>
__global__ __launch_bounds__(T) void k_topk_select(int rows, int ld, int cap, const float* __restrict__ logits,
                                                   const int* __restrict__ cnt,
                                                   const int* __restrict__ want, int* __restrict__ sel,
                                                   int fused, GenSet gs, ClassScores cs) {
    static_assert(T == 128 || T == 32 || T == 16 || T == 8, "two waves, half a wave, a quarter or an eighth");
    constexpr int W = T < 64 ? T : 64;            // rows (lanes) of a wave
    constexpr int PARTS = 128 / W;                // loads that cover a row of fewer than 128 entries
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    {
        const long long g = blockIdx.y;
        cnt += g * gs.node; want += g * gs.node; sel += g * 2ll * rows;
        if (CLS) { cs.cand += g * gs.cand; cs.cls += g * gs.node; cs.order += g * gs.node; cs.T += g * (long long)cs.C * cs.C; }
        else logits += g * (long long)rows * ld;
    }
    constexpr bool by_class = CLS;      // (a template parameter: the form that reads logits compiles as if this one did not exist)
    float* val = reinterpret_cast<float*>(smem);                           // [cap][T+1]
    unsigned char* idx = smem + (size_t)cap * (T + 1) * sizeof(float);     // [cap][T]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row_base = blockIdx.x * T;

    // cooperative fill: each wave copies the short rows of its own threads, coalesced along the row, eight rows
    // (rows have fewer than 128 entries: PARTS loads each) in flight at a time
    const int my_r = row_base + tid;
    const int my_c = my_r < rows ? cnt[my_r] : 0;
    const int my_k = my_r < rows ? min(want[my_r], my_c) : 0;
    // replayed on the LDS copy (scores by class: every row is staged, the rows std::partial_sort serves included -
    // a value costs three dependent loads there, which a thread walking its row alone would pay one after the other)
    const bool my_need = my_k > 0 && (by_class || my_k * 64 > my_c);
    const int my_ct = (by_class && my_need) ? cs.cls[cs.order[my_r]] : 0;
    const bool any_need = __syncthreads_or(my_need);
    for (int t0 = 0; any_need && t0 < W; t0 += 8) {
        float v[8][PARTS];
        int cc[8];
        if constexpr (by_class) {
            int ci[8][PARTS], ct[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u;
                cc[u] = __shfl(my_need ? my_c : 0, t, 64);
                ct[u] = __shfl(my_ct, t, 64);
                const int* crow = cs.cand + static_cast<size_t>(min(row_base + wave * 64 + t, rows - 1)) * ld;
#pragma unroll
                for (int part = 0; part < PARTS; ++part) ci[u][part] = crow[min(lane + W * part, max(cc[u] - 1, 0))];
            }
            int cl[8][PARTS];
            // (entries past a slot's count are not written by anybody: positions are clamped to the count above, and
            //  what a slot without candidates holds at position 0 is clamped to a node here - its value is not used)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int part = 0; part < PARTS; ++part) cl[u][part] = cs.cls[min(max(ci[u][part], 0), rows - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int part = 0; part < PARTS; ++part) v[u][part] = cs.T[cl[u][part] * cs.C + ct[u]];
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u;
                cc[u] = __shfl(my_need ? my_c : 0, t, 64);
                // unconditional loads at clamped addresses (a load under a condition becomes a branch with a wait)
                const float* row = logits + static_cast<size_t>(min(row_base + wave * 64 + t, rows - 1)) * ld;
#pragma unroll
                for (int part = 0; part < PARTS; ++part) v[u][part] = row[min(lane + W * part, ld - 1)];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int slot = wave * 64 + t0 + u;
#pragma unroll
            for (int part = 0; part < PARTS; ++part)
                if (lane + W * part < cc[u]) val[(lane + W * part) * (T + 1) + slot] = v[u][part];
        }
    }
    __syncthreads();

    const int r = row_base + tid;
    if (r >= rows) return;
    const int c = cnt[r], k = min(want[r], c);
    int s0 = -1, s1 = -1;
    if (k > 0) {
        if (k * 64 <= c) {
            if (fused) return;   // picked by the kernel that scored the slot
            // entry i of the row: from the array of logits, or (scores by class) from the block's LDS copy
            const float* row = by_class ? val + tid : logits + static_cast<size_t>(r) * ld;
            const int step = by_class ? T + 1 : 1;
            if (k == 1) {
                // heap of one: an element replaces the top only when strictly ahead of it
                int top = 0;
                float tv = row[0];
                for (int i = 1; i < c; ++i) {
                    float v = row[i * step];
                    if (gt(v, tv)) { top = i; tv = v; }
                }
                s0 = top;
            } else {
                // heap of two kept as (top, other): top is the entry every newcomer is compared with
                int top, oth;
                float tv, ov;
                {   // make_heap on positions 0,1
                    float v0 = row[0], v1 = row[step];
                    if (gt(v1, v0)) { top = 0; tv = v0; oth = 1; ov = v1; }
                    else { top = 1; tv = v1; oth = 0; ov = v0; }
                }
                for (int i = 2; i < c; ++i) {
                    float v = row[i * step];
                    if (gt(v, tv)) {  // pop_heap: newcomer and the old `other` re-form the heap
                        if (gt(ov, v)) { top = i; tv = v; }
                        else { top = oth; tv = ov; oth = i; ov = v; }
                    }
                }
                s0 = oth;  // sort_heap moves the top to the back
                s1 = top;
            }
        } else {
            RowView<T> q{idx + tid, val + tid};
            for (int j = 0; j < c; ++j) q.idx[j * T] = static_cast<unsigned char>(j);      // (the values are in place)
            nth_element_(q, k - 1, c);
            // std::sort of the first k-1 <= 1 elements changes nothing
            s0 = q.I(0);
            if (k == 2) s1 = q.I(1);
        }
    }
    sel[2 * r] = s0;
    sel[2 * r + 1] = s1;
}

// LDS of a block whose longest replayed row has `cap` entries (rows are at most ld long): with the 64
// candidates of the default window three blocks share a CU instead of two
constexpr size_t sel_lds(int cap, int threads = SEL_THREADS) { return (size_t)cap * ((threads + 1) * sizeof(float) + threads); }

// ------------------------------------------------------------------------------------------ emit
__device__ __forceinline__ int inv_bit(const float* __restrict__ A2, const float* __restrict__ B2,
                                       const float* __restrict__ w, float bias, int s, int v) {
    const float4* a = reinterpret_cast<const float4*>(A2 + static_cast<size_t>(s) * H);
    const float4* b = reinterpret_cast<const float4*>(B2 + static_cast<size_t>(v) * H);
    const float4* ww = reinterpret_cast<const float4*>(w);
    float acc = 0.f;
    for (int j = 0; j < H / 4; ++j) {
        float4 x = a[j], y = b[j], z = ww[j];
        acc = fmaf(fmaxf(x.x + y.x, 0.f), z.x, acc);
        acc = fmaf(fmaxf(x.y + y.y, 0.f), z.y, acc);
        acc = fmaf(fmaxf(x.z + y.z, 0.f), z.z, acc);
        acc = fmaf(fmaxf(x.w + y.w, 0.f), z.w, acc);
    }
    return (acc + bias) > 0.f ? 1 : 0;  // sigmoid(l) > 0.5
}

// Packed IEEE-half arithmetic of the reduced-precision option: planes hold halfs, a sum of two of them is rounded to
// half (v_pk_add_f16), the products with the half-rounded weights are accumulated in f32 two at a time
// (v_dot2c_f32_f16), channels ascending.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float term_h(unsigned a, unsigned b, unsigned w, float acc) {
    half2_t t = __builtin_bit_cast(half2_t, a) + __builtin_bit_cast(half2_t, b);
    const half2_t zero = {0, 0};
    t = __builtin_elementwise_max(t, zero);
    return __builtin_amdgcn_fdot2(t, __builtin_bit_cast(half2_t, w), acc, false);
}
__device__ __forceinline__ unsigned pack_h(float lo, float hi) {
    half2_t v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ int inv_bit_h(const _Float16* __restrict__ A2, const _Float16* __restrict__ B2,
                                         const float* __restrict__ w, float bias, int s, int v) {
    const uint4* a = reinterpret_cast<const uint4*>(A2 + static_cast<size_t>(s) * H);
    const uint4* b = reinterpret_cast<const uint4*>(B2 + static_cast<size_t>(v) * H);
    const float4* ww = reinterpret_cast<const float4*>(w);
    float acc = 0.f;
    for (int j = 0; j < H / 8; ++j) {
        const uint4 x = a[j], y = b[j];
        const float4 z0 = ww[2 * j], z1 = ww[2 * j + 1];
        acc = term_h(x.x, y.x, pack_h(z0.x, z0.y), acc);
        acc = term_h(x.y, y.y, pack_h(z0.z, z0.w), acc);
        acc = term_h(x.z, y.z, pack_h(z1.x, z1.y), acc);
        acc = term_h(x.w, y.w, pack_h(z1.z, z1.w), acc);
    }
    return (acc + bias) > 0.f ? 1 : 0;
}

// E: slots between the source row and the target row of a graph's edge list (its edge count for a single
// graph, the slot capacity for a set, where graph g's lists start 2 E g slots in and ecount[g] gets its count)
template <bool HALF>
__global__ void k_emit(int n, int K, long long E, const float* __restrict__ x, const int* __restrict__ order,
                       const int* __restrict__ want, const int* __restrict__ cnt, const int* __restrict__ cand,
                       const int* __restrict__ fb, const int* __restrict__ eoff, const int* __restrict__ sel,
                       const void* __restrict__ A2v, const void* __restrict__ B2v, const float* __restrict__ w2a,
                       const float* __restrict__ b2a, long long* __restrict__ edge_index,
                       long long* __restrict__ edge_attr, float* __restrict__ x_out, GenSet gs,
                       int* __restrict__ ecount, const int* __restrict__ cls = nullptr, int C = 0,
                       const int* __restrict__ pair_bit = nullptr) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (cls) {      // the inversion bit of a (source class, target class) pair is read from the graph's table
        cls += static_cast<long long>(blockIdx.y) * gs.node;
        pair_bit += static_cast<size_t>(blockIdx.y) * C * C;
    }
    {
        const long long g = blockIdx.y;
        x += g * gs.x; order += g * gs.node; want += g * gs.node; cnt += g * gs.node; cand += g * gs.cand;
        fb += g * gs.node; eoff += g * gs.eoff; sel += g * 2ll * n;
        edge_index += g * 2 * E; edge_attr += g * E; x_out += g * 2ll * n;
        if (ecount && r == 0) ecount[g] = eoff[n];
    }
    const size_t plane_off = static_cast<size_t>(blockIdx.y) * gs.proj * H;
    const float* A2 = static_cast<const float*>(A2v) + plane_off;
    const float* B2 = static_cast<const float*>(B2v) + plane_off;
    const _Float16* A2h = static_cast<const _Float16*>(A2v) + plane_off;
    const _Float16* B2h = static_cast<const _Float16*>(B2v) + plane_off;
    auto bit = [&](int src, int tgt, float bias) {
        if (cls) return pair_bit[cls[src] * C + cls[tgt]];
        return HALF ? inv_bit_h(A2h, B2h, w2a, bias, src, tgt) : inv_bit(A2, B2, w2a, bias, src, tgt);
    };
    const int v = order[r];
    const int off = eoff[r], ne = eoff[r + 1] - off;
    int invs = 0;
    if (ne > 0) {
        const int c = cnt[r];
        int s[2], b[2];
        if (c == 0) {
            s[0] = s[1] = fb[r];
            b[0] = b[1] = 0;
        } else {
            const float bias = b2a[0];
            s[0] = cand[static_cast<size_t>(r) * K + sel[2 * r]];
            b[0] = bit(s[0], v, bias);
            if (min(want[r], c) == 2) {
                s[1] = cand[static_cast<size_t>(r) * K + sel[2 * r + 1]];
                b[1] = bit(s[1], v, bias);
            } else {  // fewer candidates than fan-ins: the first pick is repeated
                s[1] = s[0];
                b[1] = b[0];
            }
        }
        for (int j = 0; j < ne; ++j) {
            edge_index[off + j] = s[j];
            edge_index[E + off + j] = v;
            edge_attr[off + j] = b[j];
            invs += b[j];
        }
    }
    x_out[2 * v] = x[2 * v];
    x_out[2 * v + 1] = static_cast<float>(invs);
}

// The same scores with the candidate rows staged in LDS.  The candidate set of slot r+1 is the set of
// slot r plus at most the node processed at r, so T consecutive slots only ever touch the rows of
// cand[r0] and of order[r0 .. r0+T-2]: K+T-1 rows per block instead of K*T gathered rows from L2.
// A candidate that was processed inside the block's range (rank >= r0, r0 = the block's first scored
// slot) sits at row K + rank - r0, any other one at its position in the ascending id list of slot r0.  Same per-pair
// arithmetic (one fmaf chain over the 128 channels in order), hence bit-identical logits.
constexpr int SCORE_T = 64;        // slots per block while the window holds at most 128 candidates
constexpr int SCORE_T_WIDE = 32;   // ... and for windows of up to 256 (the tile must fit the 160 KB of LDS)
constexpr int SCORE_K_TILED = 256; // wider windows take the gather kernel
constexpr int SCORE_THREADS = 1024; // sixteen waves share one tile: the per-pair chains are latency-bound
constexpr int SCORE_LD = H + 4;    // padded row: lanes read different rows at the same column
constexpr int SCORE_LD_H = H + 8;  // the same padding (16 bytes) for rows of halfs
constexpr int SCORE_ROUNDS = SCORE_K_TILED / 64;   // a lane scores up to this many candidates of a slot

// A slot's picks inside the wave that has just scored it: lane j of round q holds the logit of candidate 64 q + j.
// Replays the heap of k <= 2 entries that std::partial_sort keeps while it scans a row (k_topk_select's register
// path, taken when k*64 <= c) - a newcomer changes the heap only when it is strictly ahead of the heap's top, so the
// wave looks for the first such lane, applies that one step, and looks again among the lanes after it with the new top.
__device__ __forceinline__ float readlane_f(float v, int j) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}
__device__ __forceinline__ void wave_partial_sort(const float (&l)[SCORE_ROUNDS], int c, int k, int lane, int& s0, int& s1) {
    int top = 0, oth = -1;
    float tv = readlane_f(l[0], 0), ov = 0.f;
    if (k == 2) {   // make_heap on positions 0, 1
        const float v1 = readlane_f(l[0], 1);
        if (gt(v1, tv)) { oth = 1; ov = v1; }
        else { oth = 0; ov = tv; top = 1; tv = v1; }
    }
#pragma unroll
    for (int q = 0; q < SCORE_ROUNDS; ++q) {
        if (q * 64 < c) {
            const int i = q * 64 + lane;
            const bool in = i >= k && i < c;
            unsigned long long m = __ballot(in && gt(l[q], tv));
            while (m) {
                const int j = __ffsll(static_cast<long long>(m)) - 1;
                const float v = readlane_f(l[q], j);
                const int e = q * 64 + j;
                if (k == 1 || gt(ov, v)) { top = e; tv = v; }
                else { top = oth; tv = ov; oth = e; ov = v; }   // pop_heap: newcomer and the old `other` re-form the heap
                m = __ballot(in && lane > j && gt(l[q], tv));
            }
        }
    }
    if (k == 1) { s0 = top; s1 = -1; }
    else { s0 = oth; s1 = top; }   // sort_heap moves the top to the back
}

// HALF: the planes hold IEEE halfs and a pair is scored in packed half arithmetic (term_h), two accumulators for the
// even and the odd groups of eight channels.  sel != null: the picks of every slot that std::partial_sort would
// serve (k*64 <= c) are made here and k_topk_select is left with the other slots.
template <int T, bool HALF>
__global__ __launch_bounds__(SCORE_THREADS) void k_score_tiled(int n, int K, const int* __restrict__ order,
                                                     const int* __restrict__ rank, const int* __restrict__ cnt,
                                                     const int* __restrict__ want, const int* __restrict__ cand,
                                                     const void* __restrict__ Av, const void* __restrict__ Bv_,
                                                     const float* __restrict__ w2, const float* __restrict__ b2,
                                                     float* __restrict__ logits, int* __restrict__ sel, GenSet gs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LD = HALF ? SCORE_LD_H * 2 : SCORE_LD * 4;   // bytes per tile row
    constexpr int ROWB = HALF ? H * 2 : H * 4;                 // bytes per plane row
    constexpr int LPR = ROWB / 16;                             // lanes that copy one row
    {
        const long long g = blockIdx.y;
        order += g * gs.node; rank += g * gs.node; cnt += g * gs.node; want += g * gs.node; cand += g * gs.cand;
        logits += g * (long long)n * K;
        if (sel) sel += g * 2ll * n;
    }
    const unsigned char* A = static_cast<const unsigned char*>(Av) + static_cast<size_t>(blockIdx.y) * gs.proj * ROWB;
    const unsigned char* B = static_cast<const unsigned char*>(Bv_) + static_cast<size_t>(blockIdx.y) * gs.proj * ROWB;
    unsigned char* tile = smem;                                              // [K + T] rows of LD bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = tid % LPR, hw = tid / LPR;
    // base slot: the first slot of the block that is scored at all (a slot that is not a target has
    // cnt = 0 and no candidate list)
    __shared__ int s_base;
    if (tid < 64) {   // T <= 64
        const int rr = blockIdx.x * T + tid;
        const unsigned long long m = __ballot(tid < T && rr < n && cnt[rr] > 0);
        if (tid == 0) s_base = m ? blockIdx.x * T + (__ffsll((long long)m) - 1) : -1;
    }
    __syncthreads();
    const int r0 = s_base;
    if (r0 < 0) return;
    const int r_end = min(n, (int)(blockIdx.x + 1) * T);
    const int c0 = cnt[r0];
    const size_t row00 = (size_t)r0 * K;
    // rows: [0, c0) = candidates of slot r0, [K, K + T - 1) = nodes processed at r0 .. r0 + T - 2
    for (int row = hw; row < K + T - 1; row += SCORE_THREADS / LPR) {
        int node = -1;
        if (row < c0) node = cand[row00 + row];
        else if (row >= K && r0 + row - K < r_end) node = order[r0 + row - K];
        if (node >= 0)
            *reinterpret_cast<uint4*>(tile + (size_t)row * LD + 16 * l) =
                reinterpret_cast<const uint4*>(A + (size_t)node * ROWB)[l];
    }
    // the second layer's weights: f32 through uniform loads, or rounded to half once and kept in registers
    const float4* W = reinterpret_cast<const float4*>(w2);
    uint4 wh[HALF ? H / 8 : 1];
    if (HALF) {
#pragma unroll
        for (int j = 0; j < H / 8; ++j) {
            const float4 z0 = W[2 * j], z1 = W[2 * j + 1];
            wh[j] = make_uint4(pack_h(z0.x, z0.y), pack_h(z0.z, z0.w), pack_h(z1.x, z1.y), pack_h(z1.z, z1.w));
        }
    }
    __syncthreads();
    const float bias = b2[0];
    for (int r = r0 + wave; r < r_end; r += SCORE_THREADS / 64) {
        const int c = cnt[r];
        if (c <= 0) continue;
        const int v = __builtin_amdgcn_readfirstlane(order[r]);
        const unsigned char* Brow = B + (size_t)v * ROWB;
        const size_t row0 = (size_t)r * K;
        // Where a candidate's row sits in the tile.  The list of a slot holds the largest ids among the nodes
        // processed before it (gen_plan.hip), ascending: of slot r0's list a SUFFIX is still there (a newcomer pushes
        // out the smallest id), in the same order, and the nodes processed since r0 sit between its entries.  So the
        // j-th entry of this slot's list that was processed before r0 is entry (c0 - olds) + j of slot r0's list, with
        // `olds` such entries in all; an entry processed at rank >= r0 sits at row K + rank - r0.
        int rowq[SCORE_ROUNDS];
        {
            int rk[SCORE_ROUNDS];
#pragma unroll
            for (int q = 0; q < SCORE_ROUNDS; ++q) {
                const int i = q * 64 + lane;
                rk[q] = -1;
                if (q * 64 < c) rk[q] = rank[cand[row0 + (i < c ? i : 0)]];
            }
            int before[SCORE_ROUNDS], olds = 0;
#pragma unroll
            for (int q = 0; q < SCORE_ROUNDS; ++q) {
                const unsigned long long m = __ballot(q * 64 + lane < c && rk[q] < r0);
                before[q] = olds + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                olds += __popcll(m);
            }
#pragma unroll
            for (int q = 0; q < SCORE_ROUNDS; ++q) {
                const bool active = q * 64 + lane < c;
                rowq[q] = !active ? 0 : (rk[q] >= r0 ? K + rk[q] - r0 : c0 - olds + before[q]);
            }
        }
        float lg[SCORE_ROUNDS];
#pragma unroll
        for (int q = 0; q < SCORE_ROUNDS; ++q) {
            lg[q] = 0.f;
            if (q * 64 < c) {
                const int i = q * 64 + lane;
                const bool active = i < c;
                const int row = rowq[q];
                float acc = 0.f;
                if (HALF) {
                    const uint4* Ar = reinterpret_cast<const uint4*>(tile + (size_t)row * LD);
                    const uint4* Bq = reinterpret_cast<const uint4*>(Brow);
                    float acc1 = 0.f;
#pragma unroll
                    for (int j = 0; j < H / 8; j += 2) {
                        const uint4 a0 = Ar[j], a1 = Ar[j + 1], b0 = Bq[j], b1 = Bq[j + 1];
                        acc = term_h(a0.x, b0.x, wh[j].x, acc);
                        acc1 = term_h(a1.x, b1.x, wh[j + 1].x, acc1);
                        acc = term_h(a0.y, b0.y, wh[j].y, acc);
                        acc1 = term_h(a1.y, b1.y, wh[j + 1].y, acc1);
                        acc = term_h(a0.z, b0.z, wh[j].z, acc);
                        acc1 = term_h(a1.z, b1.z, wh[j + 1].z, acc1);
                        acc = term_h(a0.w, b0.w, wh[j].w, acc);
                        acc1 = term_h(a1.w, b1.w, wh[j + 1].w, acc1);
                    }
                    acc += acc1;
                } else {
                    const float4* Ar = reinterpret_cast<const float4*>(tile + (size_t)row * LD);
                    const float4* Bq = reinterpret_cast<const float4*>(Brow);
#pragma unroll 8
                    for (int j = 0; j < H / 4; ++j) {
                        float4 a = Ar[j], b = Bq[j], w = W[j];
                        acc = fmaf(fmaxf(a.x + b.x, 0.f), w.x, acc);
                        acc = fmaf(fmaxf(a.y + b.y, 0.f), w.y, acc);
                        acc = fmaf(fmaxf(a.z + b.z, 0.f), w.z, acc);
                        acc = fmaf(fmaxf(a.w + b.w, 0.f), w.w, acc);
                    }
                }
                lg[q] = acc + bias;
                if (active) logits[row0 + i] = lg[q];
            }
        }
        if (sel) {
            const int k = min(want[r], c);
            if (k > 0 && k * 64 <= c) {
                int s0, s1;
                wave_partial_sort(lg, c, k, lane, s0, s1);
                if (lane == 0) *reinterpret_cast<int2*>(sel + 2 * (size_t)r) = make_int2(s0, s1);
            }
        }
    }
}

// half planes with a window the tile does not hold: one wave per slot, candidate rows gathered from memory
__global__ __launch_bounds__(256) void k_score_h(int n, int K, const int* __restrict__ order, const int* __restrict__ cnt,
                                                 const int* __restrict__ cand, const _Float16* __restrict__ A,
                                                 const _Float16* __restrict__ B, const float* __restrict__ w2,
                                                 const float* __restrict__ b2, float* __restrict__ logits, GenSet gs) {
    const int lane = threadIdx.x & 63;
    int r = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (r >= n) return;
    {
        const long long g = blockIdx.y;
        order += g * gs.node; cnt += g * gs.node; cand += g * gs.cand;
        A += g * gs.proj * H; B += g * gs.proj * H; logits += g * (long long)n * K;
    }
    const int c = cnt[r];
    if (c <= 0) return;
    const int v = __builtin_amdgcn_readfirstlane(order[r]);
    const uint4* Bq = reinterpret_cast<const uint4*>(B + static_cast<size_t>(v) * H);
    const float4* W = reinterpret_cast<const float4*>(w2);
    const float bias = b2[0];
    const size_t row0 = static_cast<size_t>(r) * K;
    for (int base = 0; base < c; base += 64) {
        const int i = base + lane;
        const bool active = i < c;
        const int ci = cand[row0 + (active ? i : 0)];
        const uint4* Ar = reinterpret_cast<const uint4*>(A + static_cast<size_t>(ci) * H);
        float acc = 0.f, acc1 = 0.f;
#pragma unroll 4
        for (int j = 0; j < H / 8; j += 2) {   // the order of k_score_tiled's half form: same bits
            const uint4 a0 = Ar[j], a1 = Ar[j + 1], b0 = Bq[j], b1 = Bq[j + 1];
            const float4 z0 = W[2 * j], z1 = W[2 * j + 1], z2 = W[2 * j + 2], z3 = W[2 * j + 3];
            acc = term_h(a0.x, b0.x, pack_h(z0.x, z0.y), acc);
            acc1 = term_h(a1.x, b1.x, pack_h(z2.x, z2.y), acc1);
            acc = term_h(a0.y, b0.y, pack_h(z0.z, z0.w), acc);
            acc1 = term_h(a1.y, b1.y, pack_h(z2.z, z2.w), acc1);
            acc = term_h(a0.z, b0.z, pack_h(z1.x, z1.y), acc);
            acc1 = term_h(a1.z, b1.z, pack_h(z3.x, z3.y), acc1);
            acc = term_h(a0.w, b0.w, pack_h(z1.z, z1.w), acc);
            acc1 = term_h(a1.w, b1.w, pack_h(z3.z, z3.w), acc1);
        }
        if (active) logits[row0 + i] = acc + acc1 + bias;
    }
}

// ------------------------------------------------------------------------------------------ classes of equal nodes
// A node enters the generator through its feature row alone (generator.py:115-117: node_enc and node_proj act row by
// row, z is the graph's), and a template's rows take a handful of values (a type and a count).  The planes then hold
// ONE row per distinct feature row - a class - and a pair's logit and inversion bit depend on (class of the source,
// class of the target) only: a C x C table per graph, made with the very chains the per-pair kernels run (so the
// numbers are the same bits), and scoring a slot is C-table lookups.
//   T[ca * C + ct] = w2 . relu(A[ca] + B[ct]) + b2,     Tb[ca * C + ct] = (w2a . relu(A2[ca] + B2[ct]) + b2a > 0)
template <bool HALF>
__global__ __launch_bounds__(256) void k_pair_table(int C, long long plane_rows, const void* __restrict__ Av,
                                                    const void* __restrict__ Bv, const void* __restrict__ A2v,
                                                    const void* __restrict__ B2v, const float* __restrict__ w2,
                                                    const float* __restrict__ b2, const float* __restrict__ w2a,
                                                    const float* __restrict__ b2a, float* __restrict__ T,
                                                    int* __restrict__ Tb) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= C * C) return;
    const int ca = p / C, ct = p - ca * C;
    const size_t goff = static_cast<size_t>(blockIdx.y) * plane_rows * H;
    T += static_cast<size_t>(blockIdx.y) * C * C;
    Tb += static_cast<size_t>(blockIdx.y) * C * C;
    if (HALF) {
        const _Float16 *A = static_cast<const _Float16*>(Av) + goff, *B = static_cast<const _Float16*>(Bv) + goff;
        const uint4* Ar = reinterpret_cast<const uint4*>(A + static_cast<size_t>(ca) * H);
        const uint4* Bq = reinterpret_cast<const uint4*>(B + static_cast<size_t>(ct) * H);
        const float4* W = reinterpret_cast<const float4*>(w2);
        float acc = 0.f, acc1 = 0.f;
        for (int j = 0; j < H / 8; j += 2) {   // the order of k_score_tiled's half form
            const uint4 a0 = Ar[j], a1 = Ar[j + 1], b0 = Bq[j], b1 = Bq[j + 1];
            const float4 z0 = W[2 * j], z1 = W[2 * j + 1], z2 = W[2 * j + 2], z3 = W[2 * j + 3];
            acc = term_h(a0.x, b0.x, pack_h(z0.x, z0.y), acc);
            acc1 = term_h(a1.x, b1.x, pack_h(z2.x, z2.y), acc1);
            acc = term_h(a0.y, b0.y, pack_h(z0.z, z0.w), acc);
            acc1 = term_h(a1.y, b1.y, pack_h(z2.z, z2.w), acc1);
            acc = term_h(a0.z, b0.z, pack_h(z1.x, z1.y), acc);
            acc1 = term_h(a1.z, b1.z, pack_h(z3.x, z3.y), acc1);
            acc = term_h(a0.w, b0.w, pack_h(z1.z, z1.w), acc);
            acc1 = term_h(a1.w, b1.w, pack_h(z3.z, z3.w), acc1);
        }
        acc += acc1;
        T[p] = acc + b2[0];
        Tb[p] = inv_bit_h(static_cast<const _Float16*>(A2v) + goff, static_cast<const _Float16*>(B2v) + goff, w2a, b2a[0], ca, ct);
    } else {
        const float *A = static_cast<const float*>(Av) + goff, *B = static_cast<const float*>(Bv) + goff;
        const float4* Ar = reinterpret_cast<const float4*>(A + static_cast<size_t>(ca) * H);
        const float4* Bq = reinterpret_cast<const float4*>(B + static_cast<size_t>(ct) * H);
        const float4* W = reinterpret_cast<const float4*>(w2);
        float acc = 0.f;
        for (int j = 0; j < H / 4; ++j) {      // k_score's chain
            float4 a = Ar[j], b = Bq[j], w = W[j];
            acc = fmaf(fmaxf(a.x + b.x, 0.f), w.x, acc);
            acc = fmaf(fmaxf(a.y + b.y, 0.f), w.y, acc);
            acc = fmaf(fmaxf(a.z + b.z, 0.f), w.z, acc);
            acc = fmaf(fmaxf(a.w + b.w, 0.f), w.w, acc);
        }
        T[p] = acc + b2[0];
        Tb[p] = inv_bit(static_cast<const float*>(A2v) + goff, static_cast<const float*>(B2v) + goff, w2a, b2a[0], ca, ct);
    }
}

// One wave per slot: logits[r*K + i] = T[cls[cand[r*K + i]] * C + cls[order[r]]]; with sel given (K <= 256) the slots
// that std::partial_sort would serve get their picks here, as in k_score_tiled.
__global__ __launch_bounds__(256) void k_score_classes(int n, int K, int C, const int* __restrict__ order,
                                                       const int* __restrict__ cnt, const int* __restrict__ want,
                                                       const int* __restrict__ cand, const int* __restrict__ cls,
                                                       const float* __restrict__ T, float* __restrict__ logits,
                                                       int* __restrict__ sel, GenSet gs) {
    const int lane = threadIdx.x & 63;
    const int r = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (r >= n) return;
    {
        const long long g = blockIdx.y;
        order += g * gs.node; cnt += g * gs.node; want += g * gs.node; cand += g * gs.cand; cls += g * gs.node;
        T += g * (long long)C * C;
        logits += g * (long long)n * K;
        if (sel) sel += g * 2ll * n;
    }
    const int c = cnt[r];
    if (c <= 0) return;
    const int ct = cls[order[r]];
    const size_t row0 = static_cast<size_t>(r) * K;
    if (K <= SCORE_K_TILED) {
        float lg[SCORE_ROUNDS];
#pragma unroll
        for (int q = 0; q < SCORE_ROUNDS; ++q) {
            lg[q] = 0.f;
            const int i = q * 64 + lane;
            if (q * 64 < c) {
                const int ci = cand[row0 + (i < c ? i : 0)];
                lg[q] = T[cls[ci] * C + ct];
                if (i < c) logits[row0 + i] = lg[q];
            }
        }
        if (sel) {
            const int k = min(want[r], c);
            if (k > 0 && k * 64 <= c) {
                int s0, s1;
                wave_partial_sort(lg, c, k, lane, s0, s1);
                if (lane == 0) *reinterpret_cast<int2*>(sel + 2 * (size_t)r) = make_int2(s0, s1);
            }
        }
    } else {
        for (int i = lane; i < c; i += 64) logits[row0 + i] = T[cls[cand[row0 + i]] * C + ct];
    }
}

template <int T, bool HALF>
int launch_score_tiled(hipStream_t s, int B, int n, int K, const int* order, const int* rank, const int* cnt,
                       const int* want, const int* cand, const void* projA, const void* projB, const float* w2,
                       const float* b2, float* logits, int* sel, const GenSet& gs) {
    const size_t lds = (size_t)(K + T) * (HALF ? SCORE_LD_H * 2 : SCORE_LD * 4);
    AIG_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_score_tiled<T, HALF>), lds));
    hipLaunchKernelGGL((k_score_tiled<T, HALF>), dim3((unsigned)((n + T - 1) / T), B), dim3(SCORE_THREADS), lds, s, n, K,
                       order, rank, cnt, want, cand, projA, projB, w2, b2, logits, sel, gs);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

// Scores of B graphs.  half: the planes hold IEEE halfs.  sel and want given: slots that std::partial_sort would serve
// get their picks here, and *fused says so (then launch_select leaves those slots alone).
int launch_score(hipStream_t s, int B, int n, int K, const int* order, const int* rank, const int* cnt, const int* want,
                 const int* cand, const void* projA, const void* projB, bool half, const float* w2, const float* b2,
                 float* logits, int* sel, bool* fused, const GenSet& gs) {
    if (fused) *fused = false;
    if (K <= SCORE_K_TILED) {
        if (!want) sel = nullptr;
        if (fused) *fused = sel != nullptr;
        if (half)   // rows of halfs: 64 slots per tile at every window
            return launch_score_tiled<SCORE_T, true>(s, B, n, K, order, rank, cnt, want, cand, projA, projB, w2, b2, logits, sel, gs);
        if (K <= 128) return launch_score_tiled<SCORE_T, false>(s, B, n, K, order, rank, cnt, want, cand, projA, projB, w2, b2, logits, sel, gs);
        return launch_score_tiled<SCORE_T_WIDE, false>(s, B, n, K, order, rank, cnt, want, cand, projA, projB, w2, b2, logits, sel, gs);
    }
    const unsigned blocks = (unsigned)(((int64_t)n * 64 + 255) / 256);
    if (half)
        hipLaunchKernelGGL(k_score_h, dim3(blocks, B), dim3(256), 0, s, n, K, order, cnt, cand,
                           static_cast<const _Float16*>(projA), static_cast<const _Float16*>(projB), w2, b2, logits, gs);
    else
        hipLaunchKernelGGL(k_score, dim3(blocks, B), dim3(256), 0, s, n, K, order, cnt, cand,
                           static_cast<const float*>(projA), static_cast<const float*>(projB), w2, b2, logits, gs);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

int launch_select(hipStream_t s, int B, int rows, int ld, const float* logits, const int* cnt, const int* want, int* sel,
                  bool fused, const GenSet& gs, const ClassScores& cs = ClassScores{}) {
    const int cap = ld < SEL_CAP ? ld : SEL_CAP;
    // rows a block: the fewest that still leave the GPU with waves to spare (~6,500 in flight: measured 172 -> 72 us
    // with 8 rows a block on 26k rows, where 8 rows a block on 100k rows took 186 us against 56 with 16)
    const long long all_rows = (long long)rows * B;
    auto go = [&](auto kernel, int threads) -> hipError_t {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), sel_lds(SEL_CAP, threads));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3((unsigned)((rows + threads - 1) / threads), B), dim3(threads), sel_lds(cap, threads), s, rows,
                           ld, cap, logits, cnt, want, sel, fused ? 1 : 0, gs, cs);
        return hipSuccess;
    };
    static const int forced = [] { const char* e = getenv("AIG_SELECT_ROWS"); return e ? atoi(e) : 0; }();      // (a tuning aid)
    // (rows of 128 candidates and more are served while they are scored and only a few percent of the rows are replayed
    //  here: the blocks' fixed work counts then, 206 against 186 M nodes/s at candidate-k 256 - two full waves a block)
    const int threads = forced ? forced
                        : ld >= 128 ? SEL_THREADS
                        : all_rows <= 52000 ? 8 : all_rows <= 104000 ? 16 : all_rows <= 262144 ? 32 : SEL_THREADS;
    if (cs.T) {
        if (threads == 8) AIG_HIP_CHECK(go(&k_topk_select<8, true>, 8));
        else if (threads == 16) AIG_HIP_CHECK(go(&k_topk_select<16, true>, 16));
        else if (threads == 32) AIG_HIP_CHECK(go(&k_topk_select<32, true>, 32));
        else AIG_HIP_CHECK(go(&k_topk_select<SEL_THREADS, true>, SEL_THREADS));
    } else if (threads == 8) AIG_HIP_CHECK(go(&k_topk_select<8>, 8));
    else if (threads == 16) AIG_HIP_CHECK(go(&k_topk_select<16>, 16));
    else if (threads == 32) AIG_HIP_CHECK(go(&k_topk_select<32>, 32));
    else AIG_HIP_CHECK(go(&k_topk_select<SEL_THREADS>, SEL_THREADS));
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

}  // namespace
}  // namespace aig

using namespace aig;

namespace {
int gen_score(const char* who, int64_t n, int64_t K, const int32_t* order, const int32_t* rank, const int32_t* cnt,
              const int32_t* cand, const void* projA, const void* projB, bool half, const float* w2, const float* b2,
              float* logits, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX && K >= 1, "%s: bad sizes n=%lld K=%lld", who, (long long)n, (long long)K);
    if (n == 0) return AIG_OK;
    AIG_REQUIRE(order && rank && cnt && cand && projA && projB && w2 && b2 && logits, "%s: null pointer", who);
    return launch_score(static_cast<hipStream_t>(stream), 1, (int)n, (int)K, order, rank, cnt, nullptr, cand, projA, projB,
                        half, w2, b2, logits, nullptr, nullptr, GenSet{});
}

int gen_emit(const char* who, bool half, int64_t n, int64_t K, int64_t E, const float* x, const int32_t* order,
             const int32_t* want, const int32_t* cnt, const int32_t* cand, const int32_t* fb, const int32_t* eoff,
             const int32_t* sel, const void* projA2, const void* projB2, const float* w2a, const float* b2a,
             int64_t* edge_index, int64_t* edge_attr, float* x_out, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX && K >= 1 && E >= 0, "%s: bad sizes", who);
    if (n == 0) return AIG_OK;
    AIG_REQUIRE(x && order && want && cnt && cand && fb && eoff && sel && projA2 && projB2 && w2a && b2a && x_out,
                "%s: null pointer", who);
    AIG_REQUIRE(E == 0 || (edge_index && edge_attr), "%s: null edge output", who);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((n + 255) / 256));
    long long* ei = reinterpret_cast<long long*>(edge_index);
    long long* ea = reinterpret_cast<long long*>(edge_attr);
    if (half)
        hipLaunchKernelGGL(k_emit<true>, grid, dim3(256), 0, s, (int)n, (int)K, (long long)E, x, order, want, cnt, cand, fb,
                           eoff, sel, projA2, projB2, w2a, b2a, ei, ea, x_out, GenSet{}, nullptr);
    else
        hipLaunchKernelGGL(k_emit<false>, grid, dim3(256), 0, s, (int)n, (int)K, (long long)E, x, order, want, cnt, cand, fb,
                           eoff, sel, projA2, projB2, w2a, b2a, ei, ea, x_out, GenSet{}, nullptr);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}
}  // namespace

extern "C" int aig_gen_score(int64_t n, int64_t K, const int32_t* order, const int32_t* rank, const int32_t* cnt,
                             const int32_t* cand, const float* projA, const float* projB, const float* w2,
                             const float* b2, float* logits, void* stream) {
    return gen_score("aig_gen_score", n, K, order, rank, cnt, cand, projA, projB, false, w2, b2, logits, stream);
}

extern "C" int aig_gen_score_half(int64_t n, int64_t K, const int32_t* order, const int32_t* rank, const int32_t* cnt,
                                  const int32_t* cand, const void* projA, const void* projB, const float* w2,
                                  const float* b2, float* logits, void* stream) {
    return gen_score("aig_gen_score_half", n, K, order, rank, cnt, cand, projA, projB, true, w2, b2, logits, stream);
}

extern "C" size_t aig_topk_select_ws_bytes(int64_t, int64_t) { return 256; }

extern "C" int aig_topk_select(int64_t rows, int64_t ld, const float* logits, const int32_t* cnt,
                               const int32_t* want, int32_t* sel, void*, size_t, void* stream) {
    AIG_REQUIRE(rows >= 0 && rows < INT_MAX && ld >= 1 && ld < INT_MAX, "aig_topk_select: bad sizes rows=%lld ld=%lld",
                (long long)rows, (long long)ld);
    if (rows == 0) return AIG_OK;
    AIG_REQUIRE(logits && cnt && want && sel, "aig_topk_select: null pointer");
    return launch_select(static_cast<hipStream_t>(stream), 1, (int)rows, (int)ld, logits, cnt, want, sel, false, GenSet{});
}

extern "C" int aig_gen_emit(int64_t n, int64_t K, int64_t E, const float* x, const int32_t* order,
                            const int32_t* want, const int32_t* cnt, const int32_t* cand, const int32_t* fb,
                            const int32_t* eoff, const int32_t* sel, const float* projA2, const float* projB2,
                            const float* w2a, const float* b2a, int64_t* edge_index, int64_t* edge_attr,
                            float* x_out, void* stream) {
    return gen_emit("aig_gen_emit", false, n, K, E, x, order, want, cnt, cand, fb, eoff, sel, projA2, projB2, w2a, b2a,
                    edge_index, edge_attr, x_out, stream);
}

extern "C" int aig_gen_emit_half(int64_t n, int64_t K, int64_t E, const float* x, const int32_t* order,
                                 const int32_t* want, const int32_t* cnt, const int32_t* cand, const int32_t* fb,
                                 const int32_t* eoff, const int32_t* sel, const void* projA2, const void* projB2,
                                 const float* w2a, const float* b2a, int64_t* edge_index, int64_t* edge_attr,
                                 float* x_out, void* stream) {
    return gen_emit("aig_gen_emit_half", true, n, K, E, x, order, want, cnt, cand, fb, eoff, sel, projA2, projB2, w2a, b2a,
                    edge_index, edge_attr, x_out, stream);
}

extern "C" int aig_gen_wire_batch(const aig_gen_wire_set* a, void* stream) {
    AIG_REQUIRE(a, "aig_gen_wire_batch: null argument block");
    AIG_REQUIRE(a->B >= 1 && a->B <= 65535 && a->n >= 1 && a->K >= 1 && a->ecap >= 1 && (a->cls || a->proj_rows >= a->n),
                "aig_gen_wire_batch: bad sizes B=%lld n=%lld K=%lld ecap=%lld rows=%lld", (long long)a->B, (long long)a->n,
                (long long)a->K, (long long)a->ecap, (long long)a->proj_rows);
    AIG_REQUIRE(a->B * a->proj_rows < INT_MAX && a->B * a->ecap < INT_MAX && a->n * a->K < INT_MAX,
                "aig_gen_wire_batch: set too large for 32-bit indices");
    AIG_REQUIRE(a->x && a->order && a->rank && a->want && a->cnt && a->fb && a->eoff && a->cand && a->proj && a->sc2_w &&
                    a->sc2_b && a->at2_w && a->at2_b && a->sel && a->edge_index && a->edge_attr &&
                    a->ecount && a->x_out,
                "aig_gen_wire_batch: null pointer");
    // (the scores need not be stored when they come from the class table and every row is short enough for the
    //  selection kernel's LDS copy: it then forms them itself, and the pass that writes n K logits falls away)
    AIG_REQUIRE(a->logits || (a->cls && a->K <= SEL_CAP),
                "aig_gen_wire_batch: logits may be NULL only with scores by class (cls) and K <= %d", SEL_CAP);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = (int)a->B, n = (int)a->n, K = (int)a->K;
    GenSet gs{};
    if (!a->plan_shared) { gs.node = n; gs.eoff = n + 1; gs.cand = (long long)n * K; }
    gs.x = a->x_shared ? 0 : 2ll * n;
    gs.proj = a->proj_rows;
    const bool half = a->proj_half != 0;
    const size_t plane = (size_t)a->B * a->proj_rows * H * (half ? 2 : 4);   // bytes
    const unsigned char* proj = static_cast<const unsigned char*>(a->proj);
    bool fused = false;
    int rc;
    const int C = (int)a->classes;
    if (a->cls) {
        AIG_REQUIRE(a->classes >= 1 && a->classes <= a->proj_rows && a->classes * a->classes * a->B < INT_MAX && a->pair_logit &&
                        a->pair_bit,
                    "aig_gen_wire_batch: classes=%lld needs 1 <= classes <= proj_rows and the two pair tables",
                    (long long)a->classes);
        const dim3 tgrid((unsigned)((C * C + 255) / 256), B);
        if (half)
            hipLaunchKernelGGL(k_pair_table<true>, tgrid, dim3(256), 0, s, C, (long long)a->proj_rows, proj, proj + plane,
                               proj + 2 * plane, proj + 3 * plane, a->sc2_w, a->sc2_b, a->at2_w, a->at2_b, a->pair_logit,
                               a->pair_bit);
        else
            hipLaunchKernelGGL(k_pair_table<false>, tgrid, dim3(256), 0, s, C, (long long)a->proj_rows, proj, proj + plane,
                               proj + 2 * plane, proj + 3 * plane, a->sc2_w, a->sc2_b, a->at2_w, a->at2_b, a->pair_logit,
                               a->pair_bit);
        if (!a->logits) {
            const ClassScores cs{a->cand, a->cls, a->order, a->pair_logit, C};
            if ((rc = launch_select(s, B, n, K, nullptr, a->cnt, a->want, a->sel, false, gs, cs))) return rc;
        } else {
            fused = K <= SCORE_K_TILED;
            hipLaunchKernelGGL(k_score_classes, dim3((unsigned)(((int64_t)n * 64 + 255) / 256), B), dim3(256), 0, s, n, K, C,
                               a->order, a->cnt, a->want, a->cand, a->cls, a->pair_logit, a->logits, fused ? a->sel : nullptr, gs);
            AIG_LAUNCH_CHECK();
        }
    } else {
        rc = launch_score(s, B, n, K, a->order, a->rank, a->cnt, a->want, a->cand, proj, proj + plane, half, a->sc2_w,
                          a->sc2_b, a->logits, a->sel, &fused, gs);
        if (rc) return rc;
    }
    if (a->logits && (rc = launch_select(s, B, n, K, a->logits, a->cnt, a->want, a->sel, fused, gs))) return rc;
    const dim3 grid((unsigned)((n + 255) / 256), B);
    long long* ei = reinterpret_cast<long long*>(a->edge_index);
    long long* ea = reinterpret_cast<long long*>(a->edge_attr);
    if (half)
        hipLaunchKernelGGL(k_emit<true>, grid, dim3(256), 0, s, n, K, (long long)a->ecap, a->x, a->order, a->want, a->cnt,
                           a->cand, a->fb, a->eoff, a->sel, proj + 2 * plane, proj + 3 * plane, a->at2_w, a->at2_b, ei, ea,
                           a->x_out, gs, a->ecount, a->cls, C, a->pair_bit);
    else
        hipLaunchKernelGGL(k_emit<false>, grid, dim3(256), 0, s, n, K, (long long)a->ecap, a->x, a->order, a->want, a->cnt,
                           a->cand, a->fb, a->eoff, a->sel, proj + 2 * plane, proj + 3 * plane, a->at2_w, a->at2_b, ei, ea,
                           a->x_out, gs, a->ecount, a->cls, C, a->pair_bit);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}
