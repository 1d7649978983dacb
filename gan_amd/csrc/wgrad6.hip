// dW[j][f] = sum_n g[n][j] x[n][f] for g, x [n,128] f32 on the bf16 matrix cores at f32 accuracy - the weight
// gradient of the critic's 128x128 projections (aig_wgrad with F = 128: GATConv.lin of the second attention layer,
// the products of the closed-form gradient penalty).
//
// As in dense6.hip every f32 number is cut into three bf16 pieces that add up to it exactly, and a product of two
// numbers is taken as the six piece products that reach its last bit (hh | hm mh | hl lh mm), accumulated in f32:
// 6/16 of the f32 matrix time, which leaves the kernel bound by reading its two operands.  The sum runs over the
// NODES, so the matrix instruction wants, per lane, eight consecutive nodes of one column - a column-major read of
// tiles that arrive row-major.  The tiles' planes are stored row-major in LDS (256-byte rows, 16-byte chunks
// permuted by an XOR of the row bits so that both the 8-byte stores and the reads below are conflict-free) and read
// with ds_read_b64_tr_b16, which hands a lane four consecutive rows of one column.
//
// One workgroup of eight waves per CU; wave (rw, ch) owns rows 32 rw .. + 31 of dW and the two 32-column blocks
// 2 ch, 2 ch + 1.  Tiles of 32 nodes (three planes of g and of x: 48 KB, double-buffered); units of 32 nodes are
// dealt round-robin over the blocks, every block writes one [128,128] partial, k_wgrad_reduce adds them in block
// order (dense.hip).  Loads run two tiles ahead in registers; the two waves of a SIMD (ch = 0, 1) take the products
// and the refill of a trip in opposite order, so that one cuts and stores while the other multiplies.
#include <climits>

#include "common.h"

namespace aig {
namespace {

constexpr int H = AIG_HIDDEN;
constexpr int TN = 32;                      // nodes per tile
constexpr int THREADS = 512;
constexpr int PLANE_B = TN * H * 2;         // bytes per plane of a tile (bf16)
constexpr int BUF_B = 6 * PLANE_B;          // g: h m l, x: h m l
constexpr size_t WG6_LDS = (size_t)2 * BUF_B;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
using lds_s16x4 = __attribute__((address_space(3))) s16x4;

__device__ __forceinline__ void cut3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x) & 0xFFFF0000u;
    const float r = x - __uint_as_float(h);
    m = __float_as_uint(r) & 0xFFFF0000u;
    l = __float_as_uint(r - __uint_as_float(m));
}
__device__ __forceinline__ unsigned pair(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// byte offset of 16-byte chunk `chunk` (8 columns) of row `row` inside a plane
__device__ __forceinline__ int off_of(int row, int chunk) { return 256 * row + 16 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// eight consecutive nodes (rows r0 .. r0 + 7 of the plane) of one column per lane: two transposed reads of four rows
__device__ __forceinline__ bf16x8 read8(const char* lo, const char* hi) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lo);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)hi);
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(THREADS, 1) void k_wgrad128_x6(int n, const int* __restrict__ row_limit,
                                                           const float* __restrict__ g, const float* __restrict__ x,
                                                           float* __restrict__ partial /*[grid][128][128]*/) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    n = limited_rows(n, row_limit);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rw = wave & 3, ch = wave >> 2;
    const int G_ = gridDim.x;
    const int units = (n + TN - 1) / TN;
    const int nt = units > (int)blockIdx.x ? (units - (int)blockIdx.x + G_ - 1) / G_ : 0;   // tiles of this block: blockIdx.x, + G_, ...
    f32x16 acc[2] = {{0}, {0}}, low[2] = {{0}, {0}};
    if (nt > 0) {
        // loader: float4 q * 512 + tid of a tile's g and of its x, q = 0, 1 (row = idx / 32, columns 4 (idx % 32) .. + 3)
        auto request = [&](int t, float4 (&pg)[2], float4 (&px)[2]) {
            const long long row0 = ((long long)blockIdx.x + (long long)t * G_) * TN;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int idx = q * THREADS + tid;
                const long long row = min(row0 + (idx >> 5), (long long)n - 1);      // (rows past n: zeroed in deposit)
                pg[q] = *reinterpret_cast<const float4*>(g + row * H + 4 * (idx & 31));
                px[q] = *reinterpret_cast<const float4*>(x + row * H + 4 * (idx & 31));
            }
        };
        auto deposit = [&](int t, const float4 (&pg)[2], const float4 (&px)[2]) {
            char* base = lds + (t & 1) * BUF_B;
            const long long row0 = ((long long)blockIdx.x + (long long)t * G_) * TN;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int idx = q * THREADS + tid, row = idx >> 5, c4 = idx & 31;
                const bool live = row0 + row < n;
                char* p = base + off_of(row, c4 >> 1) + 8 * (c4 & 1);
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    const float4 v = which ? px[q] : pg[q];
                    unsigned h[4], m[4], l[4];
                    cut3(live ? v.x : 0.f, h[0], m[0], l[0]);
                    cut3(live ? v.y : 0.f, h[1], m[1], l[1]);
                    cut3(live ? v.z : 0.f, h[2], m[2], l[2]);
                    cut3(live ? v.w : 0.f, h[3], m[3], l[3]);
                    char* pw = p + which * 3 * PLANE_B;
                    *reinterpret_cast<uint2*>(pw) = make_uint2(pair(h[0], h[1]), pair(h[2], h[3]));
                    *reinterpret_cast<uint2*>(pw + PLANE_B) = make_uint2(pair(m[0], m[1]), pair(m[2], m[3]));
                    *reinterpret_cast<uint2*>(pw + 2 * PLANE_B) = make_uint2(pair(l[0], l[1]), pair(l[2], l[3]));
                }
            }
        };
        // transposed reads: lane 4 q + p of a group of sixteen supplies row r0 + q, chunk c0 + p / 2, half p % 2, and
        // receives column 8 c0 + (lane % 16) of the four rows.  Group G: columns 16 (G & 1) .. + 15 of the 32-column
        // block, nodes 8 (G / 2) .. + 7 of the sixteen of a step.
        const int Gq = lane >> 4, iq = (lane & 15) >> 2, ip = lane & 3;
        int a_off[2], b_off[2][2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int row = 8 * (Gq >> 1) + 4 * h2 + iq;
            a_off[h2] = off_of(row, 4 * rw + 2 * (Gq & 1) + (ip >> 1)) + 8 * (ip & 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) b_off[t][h2] = off_of(row, 4 * (2 * ch + t) + 2 * (Gq & 1) + (ip >> 1)) + 8 * (ip & 1) + 3 * PLANE_B;
        }
        auto products = [&](int it) {
            const char* base = lds + (it & 1) * BUF_B;
#pragma unroll
            for (int s = 0; s < TN / 16; ++s) {      // sixteen nodes a step (row + 16 s: the XOR term does not change)
                const char* sb = base + s * 16 * 256;
                const bf16x8 ah = read8(sb + a_off[0], sb + a_off[1]);
                const bf16x8 am = read8(sb + a_off[0] + PLANE_B, sb + a_off[1] + PLANE_B);
                const bf16x8 al = read8(sb + a_off[0] + 2 * PLANE_B, sb + a_off[1] + 2 * PLANE_B);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const bf16x8 bh = read8(sb + b_off[t][0], sb + b_off[t][1]);
                    const bf16x8 bm = read8(sb + b_off[t][0] + PLANE_B, sb + b_off[t][1] + PLANE_B);
                    const bf16x8 bl = read8(sb + b_off[t][0] + 2 * PLANE_B, sb + b_off[t][1] + 2 * PLANE_B);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
                    low[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, low[t], 0, 0, 0);
                    low[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, low[t], 0, 0, 0);
                    low[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, low[t], 0, 0, 0);
                    low[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, low[t], 0, 0, 0);
                    low[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, low[t], 0, 0, 0);
                }
            }
        };
        float4 g0[2], x0[2], g1[2], x1[2];
        request(0, g0, x0);
        if (nt > 1) request(1, g1, x1);
        deposit(0, g0, x0);
        if (nt > 2) request(2, g0, x0);
        __syncthreads();
        auto trip = [&](int it, float4 (&ng)[2], float4 (&nx)[2]) {      // tile it + 1 waits in (ng, nx)
            auto refill = [&] {
                if (it + 1 < nt) {
                    deposit(it + 1, ng, nx);
                    if (it + 3 < nt) request(it + 3, ng, nx);
                }
            };
            if (ch == 0) {
                products(it);
                refill();
            } else {
                refill();
                products(it);
            }
            __syncthreads();
        };
        for (int it = 0; it < nt; it += 2) {
            trip(it, g1, x1);
            if (it + 1 < nt) trip(it + 1, g0, x0);
        }
    }
    // accumulator register r of lane (col, kh): row (r & 3) + 8 (r >> 2) + 4 kh of the 32 x 32 block
    float* out = partial + (size_t)blockIdx.x * H * H;
    const int col = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
            out[(size_t)(32 * rw + row) * H + 32 * (2 * ch + t) + col] = acc[t][r] + low[t][r];
        }
}

}  // namespace

// partial sums of g^T x (F = 128) into partial[blocks][128][128]; blocks <= 256
int launch_wgrad128_x6(hipStream_t s, int n, int blocks, const float* g, const float* x, float* partial) {
    AIG_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_wgrad128_x6), WG6_LDS));
    AIG_LAUNCH_TIMED(AIG_KT_WGRAD128, k_wgrad128_x6, dim3(blocks), dim3(THREADS), WG6_LDS, s, n, g_row_limit, g, x, partial);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

}  // namespace aig
