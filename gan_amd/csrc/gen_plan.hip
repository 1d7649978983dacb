// Generator plan kernels: processing order and candidate windows.
// They depend on the template (types, node_depth, K) only, so one plan serves every z.
#include <climits>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace aig {
namespace {

constexpr int PI_T = 0, PO_T = 1, AND_T = 2;
constexpr int BLK = 256;  // ranks walked by one wave

// ------------------------------------------------------------------------------------------ order
// Sort key of node i: depth as an order-preserving uint32 (generator.py:76-81).  Without depth the
// key is 0 for a PI and i otherwise.  The stable radix sort on (key) keeps index order among ties.
__global__ void k_order_keys(int n, const float* __restrict__ x, int kind, const void* __restrict__ depth,
                             int* __restrict__ types, uint32_t* __restrict__ keys, int* __restrict__ ids) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int t = static_cast<int>(x[2 * i]);
        types[i] = t;
        ids[i] = i;
        uint32_t key;
        if (kind == 0) {
            key = (t == PI_T) ? 0u : static_cast<uint32_t>(i);
        } else if (kind == 1) {
            float f = static_cast<const float*>(depth)[i];
            if (f == 0.0f) f = 0.0f;  // -0.0 and +0.0 compare equal in the reference's sort
            uint32_t b = __float_as_uint(f);
            key = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
        } else {
            long long d = static_cast<const long long*>(depth)[i];
            d = d < INT_MIN ? INT_MIN : (d > INT_MAX ? INT_MAX : d);
            key = static_cast<uint32_t>(static_cast<int>(d)) ^ 0x80000000u;
        }
        keys[i] = key;
    }
}

__global__ void k_rank(int n, const int* __restrict__ order, int* __restrict__ rank) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) rank[order[r]] = r;
}

// ------------------------------------------------------------------------------------------ candidates
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

// blkid[u] = index of the rank block in which node u becomes a possible source (never for a PO);
// wmin[w]  = min blkid over the 64 ids of word w; bmax[b] = largest node id processed in block b.
__global__ void k_cand_blkid(int n, const int* __restrict__ types, const int* __restrict__ rank,
                             int* __restrict__ blkid, int* __restrict__ wmin) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nwords = (n + 63) >> 6;
    if (wave >= nwords) return;
    int u = wave * 64 + lane, b = INT_MAX;
    if (u < n) {
        if (types[u] != PO_T) b = rank[u] / BLK;
        blkid[u] = b;
    }
    b = wave_min(b);
    if (lane == 0) wmin[wave] = b;
}

__global__ void k_cand_bmax(int n, const int* __restrict__ order, int* __restrict__ bmax) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nblk = (n + BLK - 1) / BLK;
    if (wave >= nblk) return;
    int m = -1;
    for (int i = lane; i < BLK; i += 64) {
        int r = wave * BLK + i;
        if (r < n) m = max(m, order[r]);
    }
    m = wave_max(m);
    if (lane == 0) bmax[wave] = m;
}

// prefix[b*K ..) = the K largest ids u with blkid[u] < b, descending; pcnt[b] of them.
// One wave per block boundary scans id space from the top, skipping 64-id words through wmin.
__global__ void k_cand_prefix(int n, int K, const int* __restrict__ blkid, const int* __restrict__ wmin,
                              int* __restrict__ prefix, int* __restrict__ pcnt) {
    int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nblk = (n + BLK - 1) / BLK;
    if (b >= nblk) return;
    int nwords = (n + 63) >> 6;
    int filled = 0;
    for (int top = nwords - 1; top >= 0 && filled < K; top -= 64) {
        int w = top - lane;  // lane 0 looks at the highest word
        int wm = (w >= 0) ? wmin[w] : INT_MAX;
        unsigned long long m = __ballot(wm < b);
        while (m && filled < K) {
            int l = __ffsll(static_cast<long long>(m)) - 1;
            m &= m - 1;
            int id = (top - l) * 64 + (63 - lane);  // lane 0 holds the highest id of the word
            bool q = id < n && blkid[id] < b;
            unsigned long long qm = __ballot(q);
            int pos = filled + __popcll(qm & ((1ull << lane) - 1ull));
            if (q && pos < K) prefix[static_cast<size_t>(b) * K + pos] = id;
            filled += __popcll(qm);
        }
    }
    if (lane == 0) pcnt[b] = min(filled, K);
}

// Sorted (descending) list of up to K ids spread over the wave: element e lives in register e/64 of lane e%64.
template <int ELEMS>
__device__ __forceinline__ bool list_insert(int (&L)[ELEMS], int x, int K, int lane) {
    int pos = 0;
#pragma unroll
    for (int r = 0; r < ELEMS; ++r) pos += __popcll(__ballot(L[r] > x));
    if (pos >= K) return false;
#pragma unroll
    for (int r = ELEMS - 1; r >= 0; --r) {
        int up = __shfl_up(L[r], 1);
        if (r > 0) {
            int carry = __builtin_amdgcn_readlane(L[r - 1], 63);
            if (lane == 0) up = carry;
        }
        int e = r * 64 + lane;
        int v = (e < pos) ? L[r] : (e == pos ? x : up);
        L[r] = (e < K) ? v : -1;
    }
    return true;
}

// One wave walks the BLK ranks of block b in processing order (generator.py:131-189): a target takes
// a snapshot of the list (its candidates, stored ascending), then every non-PO node joins the list.
template <int ELEMS>
__global__ void k_cand_walk(int n, int K, const int* __restrict__ types, const int* __restrict__ order,
                            const int* __restrict__ prefix, const int* __restrict__ pcnt,
                            const int* __restrict__ bmax, int* __restrict__ want, int* __restrict__ cnt,
                            int* __restrict__ cand, int* __restrict__ fb, int* __restrict__ ne) {
    int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    int nblk = (n + BLK - 1) / BLK;
    if (b >= nblk) return;
    int L[ELEMS];
    int nl = pcnt[b];
#pragma unroll
    for (int r = 0; r < ELEMS; ++r) {
        int e = r * 64 + lane;
        L[r] = (e < nl) ? prefix[static_cast<size_t>(b) * K + e] : -1;
    }
    int rm = -1;  // largest id seen so far, any type
    for (int i = lane; i < b; i += 64) rm = max(rm, bmax[i]);
    rm = wave_max(rm);
    int ord[BLK / 64], typ[BLK / 64];
#pragma unroll
    for (int q = 0; q < BLK / 64; ++q) {
        int r = b * BLK + q * 64 + lane;
        ord[q] = (r < n) ? order[r] : -1;
        typ[q] = (r < n) ? types[ord[q]] : 0;
    }
#pragma unroll
    for (int q = 0; q < BLK / 64; ++q) {
        for (int i = 0; i < 64; ++i) {
            int r = b * BLK + q * 64 + i;
            if (r >= n) break;
            int v = __builtin_amdgcn_readlane(ord[q], i);
            int t = __builtin_amdgcn_readlane(typ[q], i);
            int w = (t == PI_T) ? 0 : (t == AND_T ? 2 : 1);
            if (w) {
#pragma unroll
                for (int rr = 0; rr < ELEMS; ++rr) {
                    int e = rr * 64 + lane;
                    if (e < nl) cand[static_cast<size_t>(r) * K + (nl - 1 - e)] = L[rr];
                }
            }
            if (lane == 0) {
                want[r] = w;
                cnt[r] = w ? nl : 0;
                fb[r] = (w && nl == 0) ? rm : -1;
                ne[r] = (w && (nl > 0 || rm >= 0)) ? w : 0;
            }
            rm = max(rm, v);
            if (t != PO_T) {
                if (list_insert<ELEMS>(L, v, K, lane)) nl = min(nl + 1, K);
            }
        }
    }
}

template <int ELEMS>
void launch_walk(int nblk, hipStream_t s, int n, int K, const int* types, const int* order, const int* prefix,
                 const int* pcnt, const int* bmax, int* want, int* cnt, int* cand, int* fb, int* ne) {
    int grid = (nblk * 64 + 255) / 256;
    hipLaunchKernelGGL(k_cand_walk<ELEMS>, dim3(grid), dim3(256), 0, s, n, K, types, order, prefix, pcnt, bmax, want,
                       cnt, cand, fb, ne);
}

size_t sort_temp_bytes(int64_t n) {
    size_t bytes = 0;
    uint32_t* k = nullptr;
    int* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, static_cast<size_t>(n), 0, 32, hipStream_t(0));
    return bytes;
}
size_t scan_temp_bytes(int64_t n) {
    size_t bytes = 0;
    int* p = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, p, p, 0, static_cast<size_t>(n), rocprim::plus<int>(), hipStream_t(0));
    return bytes;
}

}  // namespace
}  // namespace aig

using namespace aig;

extern "C" size_t aig_gen_order_ws_bytes(int64_t n) {
    if (n <= 0) return 256;
    return 3 * align_up(static_cast<size_t>(n) * 4) + align_up(sort_temp_bytes(n)) + 256;
}

extern "C" int aig_gen_order(int64_t n, const float* x, int depth_kind, const void* depth, int32_t* types,
                             int32_t* order, int32_t* rank, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX, "aig_gen_order: n=%lld out of range", (long long)n);
    if (n == 0) return AIG_OK;
    AIG_REQUIRE(x && types && order && rank && ws, "aig_gen_order: null pointer");
    AIG_REQUIRE(depth_kind >= 0 && depth_kind <= 2, "aig_gen_order: depth_kind=%d", depth_kind);
    AIG_REQUIRE(depth_kind == 0 || depth, "aig_gen_order: depth_kind=%d without a depth pointer", depth_kind);
    if (ws_bytes < aig_gen_order_ws_bytes(n))
        return fail(AIG_EWORKSPACE, "aig_gen_order: workspace %zu < %zu", ws_bytes, aig_gen_order_ws_bytes(n));
    hipStream_t s = static_cast<hipStream_t>(stream);
    Carver c(ws);
    uint32_t* keys_in = c.take<uint32_t>(n);
    uint32_t* keys_out = c.take<uint32_t>(n);
    int* ids = c.take<int>(n);
    size_t tb = sort_temp_bytes(n);
    void* temp = c.take<char>(tb);
    hipLaunchKernelGGL(k_order_keys, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, (int)n, x, depth_kind, depth, types,
                       keys_in, ids);
    AIG_LAUNCH_CHECK();
    AIG_HIP_CHECK(rocprim::radix_sort_pairs(temp, tb, keys_in, keys_out, ids, order, static_cast<size_t>(n), 0, 32, s));
    hipLaunchKernelGGL(k_rank, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, (int)n, order, rank);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

extern "C" size_t aig_gen_candidates_ws_bytes(int64_t n, int64_t K) {
    if (n <= 0 || K <= 0) return 256;
    size_t nblk = (n + BLK - 1) / BLK, nwords = (n + 63) / 64;
    return align_up(n * 4) /*blkid*/ + align_up(nwords * 4) + 2 * align_up(nblk * 4) /*bmax,pcnt*/ +
           align_up(nblk * K * 4) /*prefix*/ + align_up((n + 1) * 4) /*ne*/ + align_up(scan_temp_bytes(n + 1)) + 256;
}

extern "C" int aig_gen_candidates(int64_t n, int64_t K, const int32_t* types, const int32_t* order,
                                  const int32_t* rank, int32_t* want, int32_t* cnt, int32_t* cand, int32_t* fb,
                                  int32_t* eoff, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX, "aig_gen_candidates: n=%lld out of range", (long long)n);
    AIG_REQUIRE(K >= 1, "aig_gen_candidates: candidate window K=%lld must be >= 1", (long long)K);
    if (K > AIG_MAX_CANDIDATE_K)
        return fail(AIG_ELIMIT, "aig_gen_candidates: candidate window K=%lld exceeds the limit %d", (long long)K,
                    AIG_MAX_CANDIDATE_K);
    hipStream_t s = static_cast<hipStream_t>(stream);
    AIG_REQUIRE(eoff, "aig_gen_candidates: null eoff");
    if (n == 0) {
        AIG_HIP_CHECK(fill_words(eoff, 0u, 4, s));
        return AIG_OK;
    }
    AIG_REQUIRE(types && order && rank && want && cnt && cand && fb && ws, "aig_gen_candidates: null pointer");
    if (ws_bytes < aig_gen_candidates_ws_bytes(n, K))
        return fail(AIG_EWORKSPACE, "aig_gen_candidates: workspace %zu < %zu", ws_bytes,
                    aig_gen_candidates_ws_bytes(n, K));
    int nblk = (int)((n + BLK - 1) / BLK), nwords = (int)((n + 63) / 64);
    Carver c(ws);
    int* blkid = c.take<int>(n);
    int* wmin = c.take<int>(nwords);
    int* bmax = c.take<int>(nblk);
    int* pcnt = c.take<int>(nblk);
    int* prefix = c.take<int>(static_cast<size_t>(nblk) * K);
    int* ne = c.take<int>(n + 1);
    size_t tb = scan_temp_bytes(n + 1);
    void* temp = c.take<char>(tb);
    AIG_HIP_CHECK(fill_words(ne + n, 0u, 4, s));
    hipLaunchKernelGGL(k_cand_blkid, dim3((nwords * 64 + 255) / 256), dim3(256), 0, s, (int)n, types, rank, blkid, wmin);
    hipLaunchKernelGGL(k_cand_bmax, dim3((nblk * 64 + 255) / 256), dim3(256), 0, s, (int)n, order, bmax);
    hipLaunchKernelGGL(k_cand_prefix, dim3((nblk * 64 + 255) / 256), dim3(256), 0, s, (int)n, (int)K, blkid, wmin,
                       prefix, pcnt);
    AIG_LAUNCH_CHECK();
    if (K <= 64) launch_walk<1>(nblk, s, n, K, types, order, prefix, pcnt, bmax, want, cnt, cand, fb, ne);
    else if (K <= 128) launch_walk<2>(nblk, s, n, K, types, order, prefix, pcnt, bmax, want, cnt, cand, fb, ne);
    else if (K <= 256) launch_walk<4>(nblk, s, n, K, types, order, prefix, pcnt, bmax, want, cnt, cand, fb, ne);
    else if (K <= 512) launch_walk<8>(nblk, s, n, K, types, order, prefix, pcnt, bmax, want, cnt, cand, fb, ne);
    else launch_walk<16>(nblk, s, n, K, types, order, prefix, pcnt, bmax, want, cnt, cand, fb, ne);
    AIG_LAUNCH_CHECK();
    AIG_HIP_CHECK(rocprim::exclusive_scan(temp, tb, ne, eoff, 0, static_cast<size_t>(n + 1), rocprim::plus<int>(), s));
    return AIG_OK;
}
