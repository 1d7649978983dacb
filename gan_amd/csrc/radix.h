// Stable least-significant-digit radix sort of (key, value) pairs for the sizes this library sorts all the time:
// a few thousand to half a million edge slots or nodes with keys of up to 27 bits (node ids, levels).
//
// The library sort (rocprim::radix_sort_pairs) takes its merge-sort path below a million items - a block sort and
// log2(items / block) merge passes, ten launches and ~60 us for 200,000 pairs - and its one-sweep path costs ~90 us
// at any of these sizes.  Here a pass over a 9-bit digit is two launches: per-tile digit histograms, then a scatter
// in which every tile adds up the histograms of the tiles before it by itself (at most RS_MAX_TILES tiles of 8,192 or
// 16,384 items, so that sum is 32 coalesced loads per thread) - no scan kernel, no look-back state, nothing to clear.
// 17-bit node ids are two passes = four launches (~31 us of kernel time at 200,000 pairs).  More than three passes,
// more than 524,288 items (262,144 with 8-byte keys) or fewer than 1,025 go to the library sort.
//
// Stability (equal keys keep their input order) is what the callers rely on: edge lists by target keep the order of
// the edge slots, which fixes the order of every floating-point sum over a node's edges.
#pragma once
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace aig {
namespace rs {
namespace {      // (every translation unit that sorts holds its own copy of the kernels)

constexpr int RS_TILE_MIN = 8192;      // items per tile: 8192 (1024 threads x 8) or 16384 (512 threads x 32: the longer
                                       // register arrays need the larger register budget of a smaller block)
constexpr int RS_DIGIT = 9;
constexpr int RS_BINS = 1 << RS_DIGIT;
constexpr int RS_MAX_TILES = 32;
constexpr int RS_MAX_PASSES = 3;      // (four passes cost what the library sort costs)
constexpr size_t RS_MIN_ITEMS = 1025;                          // below: one block of the library sort
constexpr size_t RS_MAX_ITEMS = (size_t)RS_MAX_TILES * RS_TILE_MIN * 2;   // 524,288

// digit histogram of every tile: hist[tile * RS_BINS + digit]
template <typename K, int IPT, int RS_THREADS>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const K* __restrict__ keys, int M, int shift, unsigned mask,
                                                        int* __restrict__ hist) {
    __shared__ int bins[RS_BINS];
    const int tid = threadIdx.x;
    const int base = blockIdx.x * (RS_THREADS * IPT);
    K key[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int i = base + r * RS_THREADS + tid;
        key[r] = keys[i < M ? i : 0];
    }
    for (int i = tid; i < RS_BINS; i += RS_THREADS) bins[i] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < IPT; ++r)
        if (base + r * RS_THREADS + tid < M) atomicAdd(&bins[(unsigned)(key[r] >> shift) & mask], 1);
    __syncthreads();
    for (int i = tid; i < RS_BINS; i += RS_THREADS) hist[blockIdx.x * RS_BINS + i] = bins[i];
}

// One tile: wave w owns the contiguous slice [w * 64 * IPT, (w + 1) * 64 * IPT) of it and walks it 64 items at a
// time, so (wave, round, lane) order is input order.  A round ranks its 64 items among those of equal digit by
// comparing digits across the wave (one ballot per digit bit); the running count per (wave, digit) lives in LDS.
// The tile is then put in digit order in LDS and written out from there, so that the items of one digit leave as
// one run of consecutive addresses (written straight from the registers, every lane of a store would hit a
// different cache line, and a tile's sixteen thousand such stores queue up behind one CU's address unit).
template <typename K, typename V, int IPT, int RS_THREADS>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const K* __restrict__ kin, const V* __restrict__ vin,
                                                           K* __restrict__ kout, V* __restrict__ vout, int M, int shift,
                                                           unsigned mask, int nbits, const int* __restrict__ hist,
                                                           int tiles) {
    static_assert(RS_THREADS >= RS_BINS, "one thread per digit");
    constexpr int RS_WAVES = RS_THREADS / 64, TILE = RS_THREADS * IPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_smem[];
    int(*cnt)[RS_BINS] = reinterpret_cast<int(*)[RS_BINS]>(rs_smem);
    int* delta = reinterpret_cast<int*>(rs_smem) + RS_WAVES * RS_BINS;
    K* skey = reinterpret_cast<K*>(delta + RS_BINS);
    V* sval = reinterpret_cast<V*>(skey + TILE);
    __shared__ int2 wsum[RS_BINS / 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int t = blockIdx.x;
    const int base = t * TILE + w * (64 * IPT);
    constexpr bool EARLY = IPT * sizeof(K) <= 128;     // values requested with the keys while the registers allow
    K key[IPT];
    V val[EARLY ? IPT : 1];
    int lr[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int i = base + r * 64 + lane;
        key[r] = kin[i < M ? i : 0];
        if (EARLY) val[r] = vin[i < M ? i : 0];
    }
    // what every tile holds of digit `tid` (threads 0 .. RS_BINS-1): requested now, added up after the ranking
    int h[RS_MAX_TILES];
#pragma unroll
    for (int tt = 0; tt < RS_MAX_TILES; ++tt) h[tt] = hist[(tt < tiles ? tt : 0) * RS_BINS + (tid & (RS_BINS - 1))];
#pragma unroll
    for (int i = 0; i < RS_BINS / 64; ++i) cnt[w][i * 64 + lane] = 0;
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const bool valid = base + r * 64 + lane < M;
        const unsigned d = (unsigned)(key[r] >> shift) & mask;
        // lanes of the same digit: one ballot per digit bit (bits above the digit's width are 0 in every lane and
        // change nothing), the mask kept as two 32-bit halves
        const unsigned long long live = __ballot(valid);
        unsigned plo = (unsigned)live, phi = (unsigned)(live >> 32);
#pragma unroll
        for (int b = 0; b < RS_DIGIT; ++b) {
            const int m = -(int)((d >> b) & 1u);          // all ones where the lane's bit is set
            const unsigned long long bal = __ballot(m != 0);
            plo &= ~((unsigned)bal ^ (unsigned)m);
            phi &= ~((unsigned)(bal >> 32) ^ (unsigned)m);
        }
        const int rk = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));   // same digit, lower lane
        __builtin_amdgcn_wave_barrier();
        const int old = valid ? cnt[w][d] : 0;
        __builtin_amdgcn_wave_barrier();           // (every lane's read of the count is issued before the update)
        if (valid && rk == 0) cnt[w][d] = old + __popc(plo) + __popc(phi);
        lr[r] = old + rk;
    }
    __syncthreads();
    // digit d = tid: where its items start inside the tile (all smaller digits of the tile first) and in the output
    // (all items of smaller digits, then the items of digit d in the tiles before this one)
    int before = 0, total = 0, mine = 0;
#pragma unroll
    for (int tt = 0; tt < RS_MAX_TILES; ++tt) {
        total += tt < tiles ? h[tt] : 0;
        before += tt < t ? h[tt] : 0;
    }
    if (tid < RS_BINS) {
#pragma unroll
        for (int ww = 0; ww < RS_WAVES; ++ww) mine += cnt[ww][tid];
    }
    int incl_t = mine, incl_g = total;          // inclusive scans over threads 0 .. RS_BINS-1
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int a = __shfl_up(incl_t, off), b = __shfl_up(incl_g, off);
        if (lane >= off) { incl_t += a; incl_g += b; }
    }
    if (tid < RS_BINS && lane == 63) wsum[w] = make_int2(incl_t, incl_g);
    __syncthreads();
    if (tid < RS_BINS) {
        int start_t = incl_t - mine, start_g = incl_g - total + before;
        for (int ww = 0; ww < w; ++ww) { start_t += wsum[ww].x; start_g += wsum[ww].y; }
        delta[tid] = start_g - start_t;
#pragma unroll
        for (int ww = 0; ww < RS_WAVES; ++ww) {
            const int c = cnt[ww][tid];
            cnt[ww][tid] = start_t;
            start_t += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        if (base + r * 64 + lane < M) {
            const unsigned d = (unsigned)(key[r] >> shift) & mask;
            const int j = cnt[w][d] + lr[r];
            skey[j] = key[r];
            sval[j] = EARLY ? val[r] : vin[base + r * 64 + lane];
        }
    }
    __syncthreads();
    const int here = min(TILE, M - t * TILE);
    for (int j = tid; j < here; j += RS_THREADS) {
        const K k = skey[j];
        const int pos = j + delta[(unsigned)(k >> shift) & mask];
        kout[pos] = k;
        vout[pos] = sval[j];
    }
}

template <typename K, typename V, int IPT, int THREADS>
constexpr size_t scatter_lds_bytes() {
    return (size_t)(THREADS / 64 * RS_BINS + RS_BINS) * 4 + (size_t)THREADS * IPT * (sizeof(K) + sizeof(V));
}

inline int max_passes() {      // AIG_RS_MAX_PASSES overrides (tuning aid)
    static const int v = [] {
        const char* e = getenv("AIG_RS_MAX_PASSES");
        const int x = e ? atoi(e) : 0;
        return x >= 1 && x <= 7 ? x : RS_MAX_PASSES;
    }();
    return v;
}
inline bool takes(size_t items, int bits, size_t key_bytes) {
    // (8-byte keys: a 16384-item tile with its values would not fit the LDS)
    return items >= RS_MIN_ITEMS && items <= (key_bytes > 4 ? RS_MAX_ITEMS / 2 : RS_MAX_ITEMS) && bits <= RS_DIGIT * max_passes();
}

template <typename K, typename V>
size_t library_temp_bytes(size_t items, int bits) {
    size_t bytes = 0;
    K* k = nullptr;
    V* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, items > 0 ? items : 1, 0, bits, hipStream_t(0));
    return bytes;
}

// scratch for sort_pairs: whichever of the two sorts the call takes
template <typename K, typename V>
size_t temp_bytes(size_t items, int bits) {
    const size_t lib = library_temp_bytes<K, V>(items, bits);
    if (!takes(items, bits, sizeof(K))) return lib;
    const size_t own = align_up(items * sizeof(K)) + align_up(items * sizeof(V)) + align_up((size_t)RS_MAX_TILES * RS_BINS * 4);
    return own > lib ? own : lib;
}

template <typename K, typename V, int IPT, int THREADS>
hipError_t launch_pass(const K* kin, const V* vin, K* kout, V* vout, int M, int shift, int nbits, int* hist, hipStream_t s) {
    const int tiles = (M + THREADS * IPT - 1) / (THREADS * IPT);
    const unsigned mask = (1u << nbits) - 1u;
    constexpr size_t lds = scatter_lds_bytes<K, V, IPT, THREADS>();
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&k_rs_scatter<K, V, IPT, THREADS>), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_rs_hist<K, IPT, THREADS>), dim3(tiles), dim3(THREADS), 0, s, kin, M, shift, mask, hist);
    hipLaunchKernelGGL((k_rs_scatter<K, V, IPT, THREADS>), dim3(tiles), dim3(THREADS), lds, s, kin, vin, kout, vout, M, shift,
                       mask, nbits, hist, tiles);
    return hipSuccess;
}

// Sorts `items` pairs by bits [0, bits) of the key, stably; inputs are left as they were, the result is in
// (kout, vout).  `temp` holds temp_bytes<K, V>(items, bits) bytes.  In and out buffers must not overlap.
template <typename K, typename V>
hipError_t sort_pairs(void* temp, size_t tb, const K* kin, K* kout, const V* vin, V* vout, size_t items, int bits,
                      hipStream_t s) {
    if (!takes(items, bits, sizeof(K))) return rocprim::radix_sort_pairs(temp, tb, kin, kout, vin, vout, items, 0, bits, s);
    if (tb < temp_bytes<K, V>(items, bits)) return hipErrorInvalidValue;
    char* p = static_cast<char*>(temp);
    K* ktmp = reinterpret_cast<K*>(p);
    V* vtmp = reinterpret_cast<V*>(p + align_up(items * sizeof(K)));
    int* hist = reinterpret_cast<int*>(p + align_up(items * sizeof(K)) + align_up(items * sizeof(V)));
    const int M = (int)items;
    const int passes = bits <= 0 ? 1 : (bits + RS_DIGIT - 1) / RS_DIGIT;
    const K* ksrc = kin;
    const V* vsrc = vin;
    for (int pass = 0; pass < passes; ++pass) {
        const bool to_out = ((passes - 1 - pass) & 1) == 0;
        K* kdst = to_out ? kout : ktmp;
        V* vdst = to_out ? vout : vtmp;
        const int shift = pass * RS_DIGIT;
        const int nbits = bits <= 0 ? 0 : (bits - shift < RS_DIGIT ? bits - shift : RS_DIGIT);
        hipError_t e = hipSuccess;
        if (items <= (size_t)RS_MAX_TILES * RS_TILE_MIN) e = launch_pass<K, V, 8, 1024>(ksrc, vsrc, kdst, vdst, M, shift, nbits, hist, s);
        else if constexpr (sizeof(K) <= 4) e = launch_pass<K, V, 32, 512>(ksrc, vsrc, kdst, vdst, M, shift, nbits, hist, s);
        if (e != hipSuccess) return e;
        ksrc = kdst;
        vsrc = vdst;
    }
    return hipGetLastError();
}

}  // namespace
}  // namespace rs
}  // namespace aig
