// Y = act(X W^T + bias) for X [n,128] f32 on the bf16 matrix cores at f32 accuracy - the critic's 128x128 layers
// (aig_linear128: GATConv.lin of the second attention layer, its input gradient, the products of the closed-form
// gradient penalty).
//
// The f32 matrix instructions run at 157 TFLOP/s, the bf16 ones at sixteen times that, and a product of one
// [n,128] f32 array with a 128x128 matrix sits exactly at the ridge of the f32 rate (32 flop per byte): run as f32
// products it needs the matrix cores and the memory flat out at once and reached half of either.  Here every f32
// number is cut into three bf16 pieces, x = x1 + x2 + x3 EXACTLY (each piece takes the next eight bits of the
// significand: truncation, no rounding anywhere), and the product is taken piece by piece,
//     x w = x1 w1 + (x1 w2 + x2 w1) + (x1 w3 + x3 w1 + x2 w2) + [terms below 2^-24 |x w|, dropped]
// six bf16 products accumulated in f32: the same answer as an f32 product to within its own rounding (bf16 products
// are exact in f32; what is dropped is below the last bit of the leading term), at 6/16 of the f32 matrix time.  The
// kernel is then bound by reading X and writing Y, as it should be.
//
// One workgroup of eight waves per CU; wave (cw, uh) owns output columns 32 cw .. + 31 and rows 32 uh .. + 31 of
// every 64-row tile.  The three bf16 planes of the wave's weight slice live in registers (96 per lane) as the first
// matrix operand, so a lane ends up with four consecutive output columns of its row; the tile's planes live in LDS
// (double-buffered): every thread loads 64 bytes of the tile after next from memory, cuts them and writes the
// planes while the matrix cores work on the current tile.
#include <climits>
#include <cstdlib>

#include "common.h"

namespace aig {
namespace {

constexpr int H = AIG_HIDDEN;
constexpr int BM = 64;
constexpr int THREADS = 512;
constexpr int LDA = H / 2 + 4;     // dwords per tile row of a plane (two bf16 a dword), padded: conflict-free 16-byte reads
constexpr int PLANE = BM * LDA;    // dwords per plane of a tile
constexpr int ST_LD = 32 + 4;        // floats per row of a wave's output staging block (32 x 32), padded
constexpr size_t X6_LDS = ((size_t)2 * 3 * PLANE + (size_t)(THREADS / 64) * 32 * ST_LD + 4 * H) * sizeof(unsigned);

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// x = h + m + l exactly; every piece has at most eight significant bits (the upper half of an f32 = a bf16)
__device__ __forceinline__ void cut3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x) & 0xFFFF0000u;
    const float r = x - __uint_as_float(h);
    m = __float_as_uint(r) & 0xFFFF0000u;
    l = __float_as_uint(r - __uint_as_float(m));
}
// the bf16 halves of a (element k) and b (element k + 1) as one dword
__device__ __forceinline__ unsigned pair(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

struct Cut8 {
    u32x4 h, m, l;
};
__device__ __forceinline__ Cut8 cut8(const float (&v)[8]) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cut3(v[e], h[e], m[e], l[e]);
    Cut8 c;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        c.h[e] = pair(h[2 * e], h[2 * e + 1]);
        c.m[e] = pair(m[2 * e], m[2 * e + 1]);
        c.l[e] = pair(l[2 * e], l[2 * e + 1]);
    }
    return c;
}
// v of this lane + v of the lane 32 away, in both (one vector instruction: the two halves of a wave swap registers;
// a shuffle through the LDS crossbar costs its round trip, and the compiler waits for each one in turn)
__device__ __forceinline__ float across_halves(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int ACT>
__device__ __forceinline__ float act1(float y) {
    if (ACT == 1) return fmaxf(y, 0.f);
    if (ACT == 2) return y > 0.f ? y : expm1f(y);
    return y;
}

// The node terms of an attention layer whose projection this is (GATConv: a_s[i][h] = <y[i], att_src> over the 32
// columns of head h, a_d likewise; ce[h] = <lin_edge.weight, att_edge> over the same columns): a wave holds exactly one
// head of its rows when it stores them, so the terms cost a few dozen multiply-adds there instead of a pass of their
// own over the [n,128] output.  Read by the TERMS form of the kernel only (which takes no bias and no activation: the
// layer's projection has neither).
struct NodeTerms {
    const float* att_src;
    const float* att_dst;
    const float* lin_edge_w;
    const float* att_edge;
    float* a_s;
    float* a_d;
    float* ce;
};

template <int ACT, bool TERMS>
__global__ __launch_bounds__(THREADS, 1) void k_linear128_x6(int M, const int* __restrict__ row_limit,
                                                            const float* __restrict__ X,
                                                            const float* __restrict__ W, int ldw, int wt,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            NodeTerms nt_) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    M = limited_rows(M, row_limit);
    if (M <= 0 && !TERMS) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cw = wave & 3, uh = wave >> 2, frow = lane & 31, fh = lane >> 5;
    // the attention vectors wait in LDS behind the staging blocks (the weight planes fill the registers)
    float* att = reinterpret_cast<float*>(lds + 2 * 3 * PLANE) + (THREADS / 64) * 32 * ST_LD;
    const int G = gridDim.x;
    const int ntiles = (M + BM - 1) / BM;
    const int nt = (ntiles - (int)blockIdx.x + G - 1) / G;      // tiles of this block: blockIdx.x, + G, ...
    if (nt <= 0) {      // (only under a row limit below the launch's rows; the coefficients are owed all the same)
        if (TERMS && blockIdx.x == gridDim.x - 1 && tid < 4) {
            float c = 0.f;
            for (int k = 0; k < H / 4; ++k) c += nt_.lin_edge_w[tid * (H / 4) + k] * nt_.att_edge[tid * (H / 4) + k];
            nt_.ce[tid] = c;
        }
        return;
    }
    // (the first two tiles are requested right away - see the loader below - ahead of the weight loads)
    float4 pre0[4], pre1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = q * THREADS + tid;
        const long long r0 = min((long long)blockIdx.x * BM + (idx >> 5), (long long)M - 1);
        const long long r1 = min(((long long)blockIdx.x + G) * BM + (idx >> 5), (long long)M - 1);
        pre0[q] = *reinterpret_cast<const float4*>(X + r0 * H + 4 * (idx & 31));
        pre1[q] = *reinterpret_cast<const float4*>(X + r1 * H + 4 * (idx & 31));
    }
    // (the edge coefficients are taken by the last block - the one with the fewest tiles - after its tiles)
    if (TERMS) {
        if (tid < 2 * H) att[tid] = tid < H ? nt_.att_src[tid] : nt_.att_dst[tid - H];
        if (blockIdx.x == gridDim.x - 1 && tid < 2 * H) att[2 * H + tid] = tid < H ? nt_.lin_edge_w[tid] : nt_.att_edge[tid - H];
    }
    // the wave's weight slice as matrix fragments: lane (frow, fh) holds W[32 cw + frow][16 s + 8 fh .. + 7]
    u32x4 wh[8], wm[8], wl[8];
    {
        const int c = cw * 32 + frow;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v[8];
            const int k0 = 16 * s + 8 * fh;
            if (wt) {   // stored [in][out]
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = W[(size_t)(k0 + e) * ldw + c];
            } else {
                const float4 a = *reinterpret_cast<const float4*>(W + (size_t)c * ldw + k0);
                const float4 b = *reinterpret_cast<const float4*>(W + (size_t)c * ldw + k0 + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            }
            const Cut8 c8 = cut8(v);
            wh[s] = c8.h; wm[s] = c8.m; wl[s] = c8.l;
        }
    }
    float4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        bv[g] = (bias && !TERMS) ? *reinterpret_cast<const float4*>(bias + cw * 32 + 8 * g + 4 * fh) : make_float4(0, 0, 0, 0);

    // loader: thread takes float4 q * 512 + tid of a tile, q = 0..3 (row = idx / 32, columns 4 (idx % 32) .. + 3).
    // Requests run TWO tiles ahead of the products (tile t waits in register set t % 2): the products of a tile
    // take about a microsecond, less than a trip to memory, and 32 KB in flight per CU do not fill the memory.
    auto request = [&](int t, float4 (&pre)[4]) {      // tile number t of this block (rows past M repeat row M - 1)
        const long long row0 = ((long long)blockIdx.x + (long long)t * G) * BM;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = q * THREADS + tid;
            const long long row = min(row0 + (idx >> 5), (long long)M - 1);
            pre[q] = *reinterpret_cast<const float4*>(X + row * H + 4 * (idx & 31));
        }
    };
    auto deposit = [&](int buf, const float4 (&pre)[4]) {    // cut a requested tile into the three planes of LDS buffer `buf`
        unsigned* base = lds + buf * 3 * PLANE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = q * THREADS + tid, row = idx >> 5, c4 = idx & 31;
            unsigned h[4], m[4], l[4];
            cut3(pre[q].x, h[0], m[0], l[0]);
            cut3(pre[q].y, h[1], m[1], l[1]);
            cut3(pre[q].z, h[2], m[2], l[2]);
            cut3(pre[q].w, h[3], m[3], l[3]);
            unsigned* p = base + row * LDA + 2 * c4;
            *reinterpret_cast<uint2*>(p) = make_uint2(pair(h[0], h[1]), pair(h[2], h[3]));
            *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(pair(m[0], m[1]), pair(m[2], m[3]));
            *reinterpret_cast<uint2*>(p + 2 * PLANE) = make_uint2(pair(l[0], l[1]), pair(l[2], l[3]));
        }
    };
    // One trip: the products of tile `it` (LDS buffer it % 2) and its outputs; tile it + 1 cut from `next` into the
    // other buffer (last read a trip ago, behind a barrier) and tile it + 3 requested in its place.  The two halves do
    // not depend on each other, and a wave's own vector work does not run under its own matrix instructions - but
    // under another wave's it does: the two waves of a SIMD (uh = 0 and 1 of a column group) take the halves in
    // opposite order, so that one cuts and stores while the other multiplies.
    auto products = [&](int it) {
        const unsigned* xr = lds + (it & 1) * 3 * PLANE + (uh * 32 + frow) * LDA + 4 * fh;
        f32x16 acc = {0}, low = {0};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bf16x8 xh = as_bf(*reinterpret_cast<const u32x4*>(xr + 8 * s));
            const bf16x8 xm = as_bf(*reinterpret_cast<const u32x4*>(xr + PLANE + 8 * s));
            const bf16x8 xl = as_bf(*reinterpret_cast<const u32x4*>(xr + 2 * PLANE + 8 * s));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(wh[s]), xh, acc, 0, 0, 0);
            low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(wh[s]), xm, low, 0, 0, 0);
            low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(wm[s]), xh, low, 0, 0, 0);
            low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(wh[s]), xl, low, 0, 0, 0);
            low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(wl[s]), xh, low, 0, 0, 0);
            low = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(wm[s]), xm, low, 0, 0, 0);
        }
        // registers 4g .. 4g+3 of a 32x32 block: columns 8g + 4fh .. + 3 of the wave's 32, row frow of its unit.
        // Stored from there an instruction would write 32-byte pieces of 32 different rows; the block goes through a
        // staging area of the wave's own instead and leaves as whole 128-byte lines, eight rows per instruction.
        float* stage = reinterpret_cast<float*>(lds + 2 * 3 * PLANE) + wave * 32 * ST_LD;
        float ps[4], pd[4];
        {
            float* o = stage + frow * ST_LD + 4 * fh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 y;
                y.x = TERMS ? acc[4 * g] + low[4 * g] : act1<ACT>(acc[4 * g] + low[4 * g] + bv[g].x);
                y.y = TERMS ? acc[4 * g + 1] + low[4 * g + 1] : act1<ACT>(acc[4 * g + 1] + low[4 * g + 1] + bv[g].y);
                y.z = TERMS ? acc[4 * g + 2] + low[4 * g + 2] : act1<ACT>(acc[4 * g + 2] + low[4 * g + 2] + bv[g].z);
                y.w = TERMS ? acc[4 * g + 3] + low[4 * g + 3] : act1<ACT>(acc[4 * g + 3] + low[4 * g + 3] + bv[g].w);
                *reinterpret_cast<float4*>(o + 8 * g) = y;
                if (TERMS) {
                    const float4 as4 = *reinterpret_cast<const float4*>(att + cw * 32 + 8 * g + 4 * fh);
                    const float4 ad4 = *reinterpret_cast<const float4*>(att + H + cw * 32 + 8 * g + 4 * fh);
                    ps[g] = y.x * as4.x + y.y * as4.y + y.z * as4.z + y.w * as4.w;
                    pd[g] = y.x * ad4.x + y.y * ad4.y + y.z * ad4.z + y.w * ad4.w;
                }
            }
        }
        if (TERMS) {
            // eight groups of four columns make a head (group 2g + fh here); added in the order of the layer's own
            // pass: neighbours, then pairs of neighbours, then the two halves
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                ps[g] = across_halves(ps[g]);
                pd[g] = across_halves(pd[g]);
            }
            const float s_ = (ps[0] + ps[1]) + (ps[2] + ps[3]), d_ = (pd[0] + pd[1]) + (pd[2] + pd[3]);
            const long long row = ((long long)blockIdx.x + (long long)it * G) * BM + uh * 32 + frow;
            if (fh == 0 && row < M) {
                nt_.a_s[row * 4 + cw] = s_;
                nt_.a_d[row * 4 + cw] = d_;
            }
        }
        __builtin_amdgcn_wave_barrier();
        {
            const long long row0 = ((long long)blockIdx.x + (long long)it * G) * BM + uh * 32;
            const int r8 = lane >> 3, c = 4 * (lane & 7);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 8 * q + r8;
                const float4 y = *reinterpret_cast<const float4*>(stage + r * ST_LD + c);
                if (row0 + r < M) *reinterpret_cast<float4*>(out + (row0 + r) * H + cw * 32 + c) = y;
            }
        }
        __builtin_amdgcn_wave_barrier();      // (the wave's next block overwrites its staging area)
    };
    auto refill = [&](int it, float4 (&next)[4]) {
        if (it + 1 < nt) {
            deposit((it + 1) & 1, next);
            if (it + 3 < nt) request(it + 3, next);
        }
    };
    auto trip = [&](int it, float4 (&next)[4]) {
        if (uh == 0) {
            products(it);
            refill(it, next);
        } else {
            refill(it, next);
            products(it);
        }
        __syncthreads();
    };
    deposit(0, pre0);
    if (nt > 2) request(2, pre0);
    __syncthreads();
    for (int it = 0; it < nt; it += 2) {
        trip(it, pre1);                       // tile it + 1 waits in set 1
        if (it + 1 < nt) trip(it + 1, pre0);  // tile it + 2 in set 0
    }
    if (TERMS && blockIdx.x == gridDim.x - 1 && tid < 4) {      // (the sum as the layer's own pass takes it: ascending, one lane per head)
        float c = 0.f;
        for (int k = 0; k < H / 4; ++k) c += att[2 * H + tid * (H / 4) + k] * att[3 * H + tid * (H / 4) + k];
        nt_.ce[tid] = c;
    }
}

template <int ACT, bool TERMS = false>
int launch(hipStream_t s, int M, const float* X, const float* W, int ldw, int wt, const float* bias, float* out,
           const NodeTerms& terms = NodeTerms{}) {
    AIG_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_linear128_x6<ACT, TERMS>), X6_LDS));
    const int tiles = (M + BM - 1) / BM;
    const int gx = tiles < 256 ? tiles : 256;      // one workgroup of eight waves per CU
    AIG_LAUNCH_TIMED(AIG_KT_LINEAR128, (k_linear128_x6<ACT, TERMS>), dim3(gx), dim3(THREADS), X6_LDS, s, M, g_row_limit, X, W, ldw, wt, bias, out, terms);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

}  // namespace

int launch_linear128_x6(hipStream_t s, int M, const float* X, const float* W, int ldw, int wt, const float* bias, int act,
                        float* out) {
    if (M <= 0) return AIG_OK;
    if (act == 1) return launch<1>(s, M, X, W, ldw, wt, bias, out);
    if (act == 2) return launch<2>(s, M, X, W, ldw, wt, bias, out);
    return launch<0>(s, M, X, W, ldw, wt, bias, out);
}


// Y = X W^T (or X W) together with the node terms of the attention layer it feeds, see NodeTerms
int launch_linear128_x6_terms(hipStream_t s, int M, const float* X, const float* W, int ldw, int wt, float* out,
                              const float* att_src, const float* att_dst, const float* lin_edge_w, const float* att_edge,
                              float* a_s, float* a_d, float* ce) {
    if (M <= 0) return AIG_OK;
    return launch<0, true>(s, M, X, W, ldw, wt, nullptr, out, NodeTerms{att_src, att_dst, lin_edge_w, att_edge, a_s, a_d, ce});
}

}  // namespace aig
