// Level-wise readout of the critic: per depth level mean || max of the node embeddings, empty
// levels count as zero vectors, then the mean over levels 0..L (models/discriminator.py:64-91).
//
//   mean part  = sum_i w_i h_i,            w_i = 1 / ((L+1) * |level(i)|)
//   max part   = 1/(L+1) * sum_l max_{i in l} h_i
//
// One pass over h in level-sorted order: each half-wave walks a fixed chunk of sorted rows, keeps a
// running max (with the smallest node id among equals) and flushes it with a packed 64-bit integer
// atomic max when the level changes - order independent, hence reproducible.  The weighted sum uses
// per-block partials added in block order.  No floating-point atomics.
#include <climits>
#include <cstring>
#include <mutex>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"
#include "radix.h"

namespace aig {
namespace {

constexpr int H = AIG_HIDDEN;
constexpr int THREADS = 256;
constexpr int HW_PER_BLOCK = THREADS / 32;
// Sorted rows walked by one half-wave: 8 while that still leaves the walk short of a few thousand blocks (at
// 100k nodes 32 rows made 391 blocks, six waves per CU, and the walk ran at latency, not bandwidth), 32 beyond.
__host__ __device__ inline int rows_per_halfwave(long long n) { return n >= (1 << 19) ? 32 : 8; }

// Plan header, four int32 in device memory, for a plan built without a host read of the depth vector
// (aig_level_plan_dyn): {deepest level L, nodes with a level, flags, 0}; flag bit 0 = no node carries a valid
// depth (every node is then pooled in level 0, discriminator.py:68-72), bit 1 = a depth exceeded the caller's cap.
constexpr int META_L = 0, META_NVALID = 1, META_FLAGS = 2;

__global__ void k_level_keys(int n, int L, const int* __restrict__ meta, const long long* __restrict__ depth,
                             unsigned* __restrict__ key, int* __restrict__ ids) {
    // with a header, L is the capacity: ignored nodes sort behind every level the cap allows
    const unsigned ignored = (unsigned)(L + 1);
    const bool flat = meta && (meta[META_FLAGS] & 1);
    if (meta) L = meta[META_L];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        long long d = (depth && !flat) ? depth[i] : 0;
        key[i] = (d >= 0 && d <= L) ? (unsigned)d : ignored;  // negative depth = ignored node
        ids[i] = i;
    }
}

// deepest level and number of nodes with a level, by integer atomics into the zeroed header (meta[0] starts at -1)
__global__ void k_level_scan(int n, int L_cap, const long long* __restrict__ depth, int* __restrict__ meta) {
    int top = -1, cnt = 0, over = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const long long d = depth[i];
        if (d >= 0) {
            if (d > L_cap) { over = 1; continue; }
            top = max(top, (int)d);
            ++cnt;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        top = max(top, __shfl_xor(top, off));
        cnt += __shfl_xor(cnt, off);
        over |= __shfl_xor(over, off);
    }
    // one set of atomics per block (every wave updating the three words itself cost 38 us at 100k nodes)
    __shared__ int s_top[4], s_cnt[4], s_over[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_top[wave] = top; s_cnt[wave] = cnt; s_over[wave] = over; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { top = max(top, s_top[k]); cnt += s_cnt[k]; over |= s_over[k]; }
        if (top >= 0) atomicMax(meta + META_L, top);
        if (cnt) atomicAdd(meta + META_NVALID, cnt);
        if (over) atomicOr(meta + META_FLAGS, 2);
    }
}
// (a kernel, not hipMemsetAsync: see fill_words in common.h)
__global__ void k_level_meta_init(int* __restrict__ meta) {
    if (threadIdx.x < 4) meta[threadIdx.x] = threadIdx.x == META_L ? -1 : 0;
}
__global__ void k_level_meta_finish(int n, int* __restrict__ meta) {
    if (meta[META_L] < 0) {  // no valid depth at all: one level holding every node
        meta[META_L] = 0;
        meta[META_NVALID] = n;
        meta[META_FLAGS] |= 1;
    }
}

// (L is the capacity when the plan has a header: the entries past the deepest level all equal the valid count)
__global__ void k_level_ptr(int n, int L, const unsigned* __restrict__ skey, int* __restrict__ lvl_ptr) {
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v <= L + 1; v += gridDim.x * blockDim.x) {
        int lo = 0, hi = n;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (skey[mid] < (unsigned)v) lo = mid + 1; else hi = mid;
        }
        lvl_ptr[v] = lo;
    }
}

// node_level[i] (or -1) and w[i]
__global__ void k_level_weights(int n, int L, const int* __restrict__ meta, const unsigned* __restrict__ skey,
                                const int* __restrict__ perm, const int* __restrict__ lvl_ptr,
                                int* __restrict__ node_level, float* __restrict__ w) {
    if (meta) L = meta[META_L];
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        unsigned l = skey[p];
        int i = perm[p];
        if (l <= (unsigned)L) {
            node_level[i] = (int)l;
            w[i] = 1.f / ((float)(L + 1) * (float)(lvl_ptr[l + 1] - lvl_ptr[l]));
        } else {
            node_level[i] = -1;
            w[i] = 0.f;
        }
    }
}

__device__ __forceinline__ unsigned long long pack_max(float v, int idx) {
    unsigned b = __float_as_uint(v);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // order-preserving
    return ((unsigned long long)b << 32) | (unsigned)(~(unsigned)idx);  // larger = smaller node id
}
__device__ __forceinline__ float unpack_val(unsigned long long p) {
    unsigned b = (unsigned)(p >> 32);
    b = (b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b;
    return __uint_as_float(b);
}

__global__ __launch_bounds__(THREADS) void k_pool_walk(int nvalid, int rows_per_hw, const int* __restrict__ meta,
                                                       const float* __restrict__ h,
                                                       const int* __restrict__ perm,
                                                       const int* __restrict__ node_level,
                                                       const float* __restrict__ w,
                                                       unsigned long long* __restrict__ table /*[L+1][H]*/,
                                                       float* __restrict__ partial /*[grid][H]*/) {
    __shared__ float red[HW_PER_BLOCK][H];
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const int item = blockIdx.x * HW_PER_BLOCK + hw;
    // with a header the grid covers every node and half-waves past the valid rows add nothing (never more rows
    // than the launch was sized for, whatever the header says)
    if (meta) nvalid = min(nvalid, meta[META_NVALID]);
    const int p0 = item * rows_per_hw, p1 = min(nvalid, p0 + rows_per_hw);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 mx = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 mi = make_int4(0, 0, 0, 0);
    int cur = -1;
    auto flush = [&]() {
        unsigned long long* t = table + (size_t)cur * H + l * 4;
        atomicMax(t + 0, pack_max(mx.x, mi.x));
        atomicMax(t + 1, pack_max(mx.y, mi.y));
        atomicMax(t + 2, pack_max(mx.z, mi.z));
        atomicMax(t + 3, pack_max(mx.w, mi.w));
    };
    // lane q fetches the index, level and weight of row q of the item (one coalesced load, two
    // gathers); the rows themselves then come four at a time
    const int pq = p0 + l;
    const bool onq = pq < p1;
    const int iq = onq ? perm[pq] : 0;
    const int levq = onq ? node_level[iq] : -1;
    const float wq = onq ? w[iq] : 0.f;
    const int nrows = p1 - p0;
    for (int k0 = 0; k0 < nrows; k0 += 4) {
        int ii[4], lv[4];
        float ww[4];
        float4 vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = min(k0 + u, nrows - 1);
            ii[u] = __shfl(iq, k, 32);
            lv[u] = __shfl(levq, k, 32);
            ww[u] = __shfl(wq, k, 32);
            vv[u] = reinterpret_cast<const float4*>(h + (size_t)ii[u] * H)[l];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + u >= nrows) break;
            const int i = ii[u], lev = lv[u];
            const float wi = ww[u];
            const float4 v = vv[u];
            sum.x += wi * v.x; sum.y += wi * v.y; sum.z += wi * v.z; sum.w += wi * v.w;
            if (lev != cur) {
                if (cur >= 0) flush();
                cur = lev;
                mx = v;
                mi = make_int4(i, i, i, i);
            } else {  // rows of a level arrive by ascending node id: strict > keeps the first maximum
                if (v.x > mx.x) { mx.x = v.x; mi.x = i; }
                if (v.y > mx.y) { mx.y = v.y; mi.y = i; }
                if (v.z > mx.z) { mx.z = v.z; mi.z = i; }
                if (v.w > mx.w) { mx.w = v.w; mi.w = i; }
            }
        }
    }
    // The segment a half-wave ends with is not flushed by itself: the eight half-waves of a block walk
    // 256 consecutive sorted rows, mostly of one or two levels, so their closing segments are merged per
    // level in LDS first - one atomic per block, level and column instead of one per half-wave.
    __shared__ unsigned long long seg[HW_PER_BLOCK][H];
    __shared__ int seg_level[HW_PER_BLOCK];
    if (l == 0) seg_level[hw] = cur;
    if (cur >= 0) {
        seg[hw][l * 4 + 0] = pack_max(mx.x, mi.x);
        seg[hw][l * 4 + 1] = pack_max(mx.y, mi.y);
        seg[hw][l * 4 + 2] = pack_max(mx.z, mi.z);
        seg[hw][l * 4 + 3] = pack_max(mx.w, mi.w);
    }
    reinterpret_cast<float4*>(&red[hw][0])[l] = sum;
    __syncthreads();
    if (threadIdx.x < H) {
        const int f = threadIdx.x;
        float s = 0.f;
        for (int k = 0; k < HW_PER_BLOCK; ++k) s += red[k][f];
        partial[(size_t)blockIdx.x * H + f] = s;
        int lev = -1;
        unsigned long long best = 0ull;
        for (int k = 0; k < HW_PER_BLOCK; ++k) {  // levels ascend with k (sorted rows)
            const int lk = seg_level[k];
            if (lk < 0) continue;
            const unsigned long long v = seg[k][f];
            if (lk != lev) {
                if (lev >= 0) atomicMax(table + (size_t)lev * H + f, best);
                lev = lk;
                best = v;
            } else if (v > best) {
                best = v;
            }
        }
        if (lev >= 0) atomicMax(table + (size_t)lev * H + f, best);
    }
}

// One block per feature f: out[f] = sum of the walk's partials, out[H+f] = 1/(L+1) * sum over
// non-empty levels of the level maximum (strided partial sums, then a fixed-shape tree: reproducible);
// argmax[l*H + f] = node holding that maximum (-1 for an empty level).
__global__ __launch_bounds__(256) void k_pool_finish(int nb, int L, const int* __restrict__ meta,
                                                     const float* __restrict__ partial,
                                                     const unsigned long long* __restrict__ table,
                                                     const int* __restrict__ lvl_ptr, float* __restrict__ out,
                                                     int* __restrict__ argmax) {
    __shared__ float red_s[256], red_m[256];
    const int f = blockIdx.x, t = threadIdx.x;
    if (meta) L = meta[META_L];
    float s = 0.f, m = 0.f;
    for (int b = t; b < nb; b += 256) s += partial[(size_t)b * H + f];
    for (int l = t; l <= L; l += 256) {
        const bool has = lvl_ptr[l + 1] > lvl_ptr[l];
        const unsigned long long v = table[(size_t)l * H + f];
        if (has) m += unpack_val(v);
        argmax[(size_t)l * H + f] = has ? (int)(~(unsigned)(v & 0xFFFFFFFFull)) : -1;
    }
    red_s[t] = s;
    red_m[t] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) { red_s[t] += red_s[t + o]; red_m[t] += red_m[t + o]; }
        __syncthreads();
    }
    if (t == 0) {
        out[f] = red_s[0];
        out[H + f] = red_m[0] / (float)(L + 1);
    }
}

int key_bits(int64_t n) {
    int b = 1;
    while ((1ll << b) <= n) ++b;
    return b;
}
size_t sort_temp_bytes(int64_t n, int bits) {
    return rs::temp_bytes<unsigned, int>((size_t)n, bits);
}

}  // namespace
}  // namespace aig

using namespace aig;

extern "C" size_t aig_level_plan_ws_bytes(int64_t n, int64_t L) {
    if (n <= 0) return 256;
    return 3 * align_up((size_t)n * 4) + align_up(sort_temp_bytes(n, key_bits(L + 1))) + 256;
}

namespace {
int level_plan(int64_t n, int64_t L, const int64_t* depth, int32_t* perm, int32_t* lvl_ptr, int32_t* node_level,
               float* w, int32_t* meta, bool dyn, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX && L >= 0 && L < INT_MAX - 2, "aig_level_plan: bad sizes n=%lld L=%lld",
                (long long)n, (long long)L);
    AIG_REQUIRE(lvl_ptr && (!dyn || meta), "aig_level_plan: null lvl_ptr or header");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dyn) {
        hipLaunchKernelGGL(k_level_meta_init, dim3(1), dim3(64), 0, s, meta);
        AIG_LAUNCH_CHECK();
        if (depth && n > 0) {
            hipLaunchKernelGGL(k_level_scan, dim3(grid_for(n, 256 * 8, 128)), dim3(256), 0, s, (int)n, (int)L,
                               reinterpret_cast<const long long*>(depth), meta);
            AIG_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(k_level_meta_finish, dim3(1), dim3(1), 0, s, (int)n, meta);
        AIG_LAUNCH_CHECK();
    }
    if (n == 0) {
        AIG_HIP_CHECK(fill_words(lvl_ptr, 0u, (size_t)(L + 2) * 4, s));
        return AIG_OK;
    }
    AIG_REQUIRE(perm && node_level && w && ws, "aig_level_plan: null pointer");
    if (ws_bytes < aig_level_plan_ws_bytes(n, L))
        return fail(AIG_EWORKSPACE, "aig_level_plan: workspace %zu < %zu", ws_bytes, aig_level_plan_ws_bytes(n, L));
    const int bits = key_bits(L + 1);
    Carver c(ws);
    unsigned* key = c.take<unsigned>(n);
    unsigned* skey = c.take<unsigned>(n);
    int* ids = c.take<int>(n);
    size_t tb = sort_temp_bytes(n, bits);
    void* temp = c.take<char>(tb);
    const int g = grid_for(n, 256, 4096);
    hipLaunchKernelGGL(k_level_keys, dim3(g), dim3(256), 0, s, (int)n, (int)L, (const int*)meta,
                       reinterpret_cast<const long long*>(depth), key, ids);
    AIG_LAUNCH_CHECK();
    AIG_HIP_CHECK(rs::sort_pairs(temp, tb, key, skey, ids, perm, (size_t)n, bits, s));
    hipLaunchKernelGGL(k_level_ptr, dim3(grid_for(L + 2, 256, 4096)), dim3(256), 0, s, (int)n, (int)L, skey, lvl_ptr);
    hipLaunchKernelGGL(k_level_weights, dim3(g), dim3(256), 0, s, (int)n, (int)L, (const int*)meta, skey, perm, lvl_ptr,
                       node_level, w);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}
}  // namespace

extern "C" int aig_level_plan(int64_t n, int64_t L, const int64_t* depth, int32_t* perm, int32_t* lvl_ptr,
                              int32_t* node_level, float* w, void* ws, size_t ws_bytes, void* stream) {
    return level_plan(n, L, depth, perm, lvl_ptr, node_level, w, nullptr, false, ws, ws_bytes, stream);
}

extern "C" int aig_level_plan_dyn(int64_t n, int64_t L_cap, const int64_t* depth, int32_t* perm, int32_t* lvl_ptr,
                                  int32_t* node_level, float* w, int32_t* lvl_meta, void* ws, size_t ws_bytes,
                                  void* stream) {
    return level_plan(n, L_cap, depth, perm, lvl_ptr, node_level, w, lvl_meta, true, ws, ws_bytes, stream);
}

extern "C" size_t aig_level_pool_fwd_ws_bytes(int64_t n, int64_t L) {
    const size_t rows = (size_t)rows_per_halfwave(n);
    size_t items = ((size_t)(n > 0 ? n : 1) + rows - 1) / rows;
    size_t blocks = (items + HW_PER_BLOCK - 1) / HW_PER_BLOCK;
    return align_up((size_t)(L + 1) * H * 8) + align_up(blocks * H * 4) + 256;
}

extern "C" size_t aig_level_pool_table_bytes(int64_t L) { return (size_t)(L > 0 ? L + 1 : 1) * H * 8; }

namespace {
int pool_fwd(int64_t n, int64_t nvalid, int64_t L, const int32_t* lvl_meta, const float* h, const int32_t* perm,
             const int32_t* lvl_ptr, const int32_t* node_level, const float* w, float* out, int32_t* argmax, void* ws,
             size_t ws_bytes, bool table_cleared, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX && L >= 0 && nvalid >= 0 && nvalid <= n, "aig_level_pool_fwd: bad sizes");
    AIG_REQUIRE(out && argmax && lvl_ptr && ws, "aig_level_pool_fwd: null pointer");
    if (ws_bytes < aig_level_pool_fwd_ws_bytes(n, L))
        return fail(AIG_EWORKSPACE, "aig_level_pool_fwd: workspace %zu < %zu", ws_bytes,
                    aig_level_pool_fwd_ws_bytes(n, L));
    hipStream_t s = static_cast<hipStream_t>(stream);
    Carver c(ws);
    unsigned long long* table = c.take<unsigned long long>((size_t)(L + 1) * H);
    if (!table_cleared) AIG_HIP_CHECK(fill_words(table, 0u, (size_t)(L + 1) * H * 8, s));
    int blocks = 0;
    float* partial = nullptr;
    if (nvalid > 0) {
        AIG_REQUIRE(h && perm && node_level && w, "aig_level_pool_fwd: null pointer");
        const int rows = rows_per_halfwave(n);
        int64_t items = (nvalid + rows - 1) / rows;
        blocks = (int)((items + HW_PER_BLOCK - 1) / HW_PER_BLOCK);
        partial = c.take<float>((size_t)blocks * H);
        AIG_LAUNCH_TIMED(AIG_KT_POOL_WALK, k_pool_walk, dim3(blocks), dim3(THREADS), 0, s, (int)nvalid, rows, lvl_meta, h, perm,
                         node_level, w, table, partial);
        AIG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_pool_finish, dim3(H), dim3(256), 0, s, blocks, (int)L, lvl_meta, partial, table, lvl_ptr, out,
                       argmax);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}
}  // namespace

extern "C" int aig_level_pool_fwd(int64_t n, int64_t nvalid, int64_t L, const int32_t* lvl_meta, const float* h,
                                  const int32_t* perm, const int32_t* lvl_ptr, const int32_t* node_level,
                                  const float* w, float* out, int32_t* argmax, void* ws, size_t ws_bytes,
                                  void* stream) {
    return pool_fwd(n, nvalid, L, lvl_meta, h, perm, lvl_ptr, node_level, w, out, argmax, ws, ws_bytes, false, stream);
}

extern "C" int aig_level_pool_fwd_cleared(int64_t n, int64_t nvalid, int64_t L, const int32_t* lvl_meta, const float* h,
                                          const int32_t* perm, const int32_t* lvl_ptr, const int32_t* node_level,
                                          const float* w, float* out, int32_t* argmax, void* ws, size_t ws_bytes,
                                          void* stream) {
    return pool_fwd(n, nvalid, L, lvl_meta, h, perm, lvl_ptr, node_level, w, out, argmax, ws, ws_bytes, true, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// The readout for callers that must not allocate: its packed-maximum table and block partials live in a
// file-scope device array instead of a caller-provided workspace.  Every GPU the library runs on holds its
// own copy of the array (at its own address) until the process ends.
namespace {
static __device__ __align__(256) unsigned char scratch[AIG_NOALLOC_SCRATCH_BYTES];

constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[MAX_DEVICES] = {};   // device address of `scratch` on device d, once looked up there

// Device address of this device's copy of `scratch`.  The array's name is only a host-side handle: the address
// comes from hipGetSymbolAddress, asked once per device with that device current.
int scratch_address(void** out, hipStream_t stream) {
    int dev = -1;
    AIG_HIP_CHECK(hipGetDevice(&dev));
    AIG_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "aig_level_pool_fwd_noalloc: device index %d beyond %d", dev, MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        // The look-up may load the code object and is no stream operation: not something to do while the caller's
        // stream is being recorded into a graph.  The first call on a device has to come outside any recording.
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
            return fail(AIG_EINVAL, "aig_level_pool_fwd_noalloc: the first call on a device looks up the scratch array's "
                        "address and cannot be recorded into a graph - call it once outside the recording first");
        void* p = nullptr;
        AIG_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        AIG_REQUIRE(p != nullptr, "aig_level_pool_fwd_noalloc: no device address for the scratch array");
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return AIG_OK;
}
}  // namespace

extern "C" int aig_level_pool_fwd_noalloc(int64_t n, int64_t nvalid, int64_t L, const int32_t* lvl_meta, const float* h,
                                          const int32_t* perm, const int32_t* lvl_ptr, const int32_t* node_level,
                                          const float* w, float* out, int32_t* argmax, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX && L >= 0 && L < INT_MAX - 2, "aig_level_pool_fwd_noalloc: bad sizes");
    const size_t need = aig_level_pool_fwd_ws_bytes(n, L);
    if (need > (size_t)AIG_NOALLOC_SCRATCH_BYTES)
        return fail(AIG_ELIMIT, "aig_level_pool_fwd_noalloc: n=%lld nodes with %lld levels need %zu bytes of scratch, the "
                    "built-in array holds %zu (use aig_level_pool_fwd with a workspace of your own)",
                    (long long)n, (long long)L + 1, need, (size_t)AIG_NOALLOC_SCRATCH_BYTES);
    void* ptr = nullptr;
    if (int rc = scratch_address(&ptr, static_cast<hipStream_t>(stream))) return rc;
    // the array keeps the previous call's contents: aig_level_pool_fwd clears the table it carves out of it
    return aig_level_pool_fwd(n, nvalid, L, lvl_meta, h, perm, lvl_ptr, node_level, w, out, argmax, ptr,
                              (size_t)AIG_NOALLOC_SCRATCH_BYTES, stream);
}
