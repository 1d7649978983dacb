// Weight gradient of the critic's dense projections: dW[j][f] = sum_n g[n][j] * x[n][f]
// (g = gradient w.r.t. x W^T, [n,128]; x [n,F]; dW [128,F] in PyTorch's [out,in] layout).
// The reduction runs over the NODES (n ~ 1e5..1e6) while the output is tiny, a shape library GEMMs
// serve with a handful of tiles; here the node range is split over the whole chip, every block
// accumulates a full [128,F] partial on the f32 matrix cores straight from global memory (both
// operands are read as 128-byte row segments, no staging), and the partials are added in block
// order, so the result is reproducible.
#include <climits>
#include <cstdint>
#include <cstdlib>

#include "common.h"

namespace aig {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int H = AIG_HIDDEN;
constexpr int WG_BLOCKS_MAX = 256;         // F = 128: one 64 KB partial per CU
constexpr int WG_SMALL_BLOCKS_MAX = 512;   // small F: partials are at most 2 KB, two blocks per CU

// F == 128.  Node tiles of TILE rows of g and x go global -> LDS by direct 16-byte loads (no
// register stop; one wave instruction fills two whole 512-byte rows), double-buffered: tile t+1 arrives
// while the matrix cores work on tile t, one barrier per tile.  EIGHT waves per workgroup, two per SIMD
// (one wave's operand reads and address arithmetic run under the other's products): wave w owns rows
// 32(w&3) .. +31 of dW and the two 32-column tiles 2(w>>2), 2(w>>2)+1.  MFMA 32x32x2: A[i][k] =
// g[n0+k][32(w&3)+i], B[k][c] = x[n0+k][32t+c], k = lane>>5, i/c = lane&31; the two k rows of an operand
// are read by the two halves of the wave, which the LDS serves in separate cycles, so rows need no padding.
// A tile is two 32-row units; units are dealt round-robin over the blocks (unit u belongs to block
// u % gridDim.x: at most one unit of imbalance, and at any moment the grid reads one contiguous stretch
// of rows).  The next tile's 8 loads per wave are issued one every fourth pair of node rows inside the
// product loop - all at once at the top of a tile, by every workgroup of the grid in the same
// microsecond, they stall the waves at issue.
constexpr int TILE = 64;
constexpr size_t WGRAD_LDS = (size_t)2 * 2 * TILE * H * sizeof(float);

__global__ __launch_bounds__(512, 1) void k_wgrad128(int n, const int* __restrict__ row_limit, const float* __restrict__ g,
                                                     const float* __restrict__ x,
                                                     float* __restrict__ partial /*[grid][128][128]*/) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [buf][g|x][TILE][H]
    n = limited_rows(n, row_limit);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int rw = w & 3, ch = w >> 2;
    const int col = lane & 31, kh = lane >> 5;
    const int G_ = gridDim.x;
    const int units = (n + 31) / 32;
    const int nu = (units - (int)blockIdx.x + G_ - 1) / G_;   // this block's units: blockIdx.x + k * gridDim.x
    const int ntiles = nu > 0 ? (nu + 1) / 2 : 0;
    auto unit_row = [&](int k) { return ((long long)blockIdx.x + (long long)k * G_) * 32; };
    // Direct load q (0..3) of a tile: wave w fills tile rows 2(8q+w) and 2(8q+w)+1 of g and of x; lane ->
    // (row parity kh, 16-byte column col).  Rows past n read row n-1 (zeroed in LDS before use, see below).
    // The loads are issued from inline asm: after the builtin the compiler waits for ALL memory traffic
    // before the next LDS read, i.e. for tile t+1 before the products of tile t.  These are this kernel's
    // only memory operations until the epilogue, so "none outstanding" at the top of a trip is exact.
    const unsigned lds0 = lds_address(lds);
    auto fetch1 = [&](int q, int tile, int buf) {
        const unsigned Gd = lds0 + (unsigned)((size_t)buf * 2 * TILE * H * sizeof(float));
        const unsigned Xd = Gd + (unsigned)(TILE * H * sizeof(float));
        const int pr = 2 * (8 * q + w);
        const long long row = unit_row(2 * tile + (q >> 1)) + (pr & 31) + kh;
        const size_t off = (size_t)(row < n ? row : n - 1) * H + 4 * col;
        load_to_lds16_uncounted(g + off, __builtin_amdgcn_readfirstlane(Gd + (unsigned)(pr * H * sizeof(float))));
        load_to_lds16_uncounted(x + off, __builtin_amdgcn_readfirstlane(Xd + (unsigned)(pr * H * sizeof(float))));
    };
    f32x16 acc[2] = {{0}, {0}};
    if (ntiles > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) fetch1(q, 0, 0);
    }
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's loads of tile `tile` ...
        __syncthreads();  // ... and everybody else's: the tile is complete, the other buffer no longer read
        // rows of g that must not contribute: the second unit of an odd last tile, and the rows past n of the
        // very last unit
        const long long r0 = unit_row(2 * tile), r1 = unit_row(2 * tile + 1);
        const int have0 = (int)min(32LL, (long long)n - r0);
        const int have1 = 2 * tile + 1 < nu ? (int)min(32LL, (long long)n - r1) : 0;
        if (have0 < 32 || have1 < 32) {   // uniform over the block
            float* Gl = lds + (size_t)buf * 2 * TILE * H;
            for (int i = t; i < TILE * H; i += 512) {
                const int row = i >> 7;
                if ((row < 32 && row >= have0) || (row >= 32 && row - 32 >= have1)) Gl[i] = 0.f;
            }
            __syncthreads();
        }
        const bool more = tile + 1 < ntiles;
        const float* Gp = lds + (size_t)buf * 2 * TILE * H + 32 * rw + col;
        const float* Xp = lds + (size_t)buf * 2 * TILE * H + TILE * H + 64 * ch + col;
        // operands of pair kk+1 are requested before the two products of pair kk are issued
        float a = Gp[kh * H];
        float b0 = Xp[kh * H], b1 = Xp[kh * H + 32];
#pragma unroll
        for (int kk = 0; kk < TILE / 2; ++kk) {
            const int rn = (2 * (kk + 1 < TILE / 2 ? kk + 1 : kk) + kh) * H;
            const float na = Gp[rn];
            const float n0 = Xp[rn], n1 = Xp[rn + 32];
            if ((kk & 7) == 0 && more) fetch1(kk >> 3, tile + 1, buf ^ 1);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
            a = na; b0 = n0; b1 = n1;
        }
    }
    float* out = partial + (size_t)blockIdx.x * H * H;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;  // accumulator register -> tile row
            out[(size_t)(32 * rw + row) * H + 32 * (2 * ch + tt) + col] = acc[tt][r];
        }
}

// small F (the first layer has F = 2): half-wave per node row, 16-byte loads of g, the F inputs of
// the row broadcast; the eight row groups of a block are added in fixed order through LDS
template <int F>
__global__ __launch_bounds__(256) void k_wgrad_small(int n, const int* __restrict__ row_limit, const float* __restrict__ g,
                                                     const float* __restrict__ x,
                                                     float* __restrict__ partial /*[grid][128][F]*/) {
    __shared__ float4 red[8][F][32];
    n = limited_rows(n, row_limit);
    const int c4 = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    const int chunk = (n + gridDim.x - 1) / gridDim.x;
    const int n0 = blockIdx.x * chunk, n1 = min(n, n0 + chunk);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4 acc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int row = n0 + r0; row < n1; row += 8) {
        const float4 gv = g4[(size_t)row * (H / 4) + c4];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float xv = x[(size_t)row * F + f];
            acc[f].x = fmaf(gv.x, xv, acc[f].x);
            acc[f].y = fmaf(gv.y, xv, acc[f].y);
            acc[f].z = fmaf(gv.z, xv, acc[f].z);
            acc[f].w = fmaf(gv.w, xv, acc[f].w);
        }
    }
#pragma unroll
    for (int f = 0; f < F; ++f) red[r0][f][c4] = acc[f];
    __syncthreads();
    if (r0 < F) {
        float4 s4 = red[0][r0][c4];
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            const float4 v = red[q][r0][c4];
            s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
        }
        float* out = partial + (size_t)blockIdx.x * H * F + (size_t)(4 * c4) * F + r0;
        out[0] = s4.x; out[F] = s4.y; out[2 * F] = s4.z; out[3 * F] = s4.w;
    }
}

// out[e] = sum_b partial[b][e]: 64 elements per block, sixteen stripes of b per element (ascending
// within a stripe), stripes combined by a fixed tree
__global__ __launch_bounds__(1024) void k_wgrad_reduce(int nb, int elems, const float* __restrict__ partial,
                                                        float* __restrict__ out) {
    __shared__ float red[16][64];
    const int el = threadIdx.x & 63, st = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // independent chains keep loads in flight
    if (e < elems) {
        int b = st;
        for (; b + 48 < nb; b += 64) {
            s0 += partial[(size_t)b * elems + e];
            s1 += partial[(size_t)(b + 16) * elems + e];
            s2 += partial[(size_t)(b + 32) * elems + e];
            s3 += partial[(size_t)(b + 48) * elems + e];
        }
        for (; b < nb; b += 16) s0 += partial[(size_t)b * elems + e];
    }
    red[st][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int half = 8; half > 0; half >>= 1) {
        if (st < half) red[st][el] += red[st + half][el];
        __syncthreads();
    }
    if (st == 0 && e < elems) out[e] = red[0][el];
}

int wg_blocks(int64_t n, int per_block, int cap) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// y[n][c] = x[n][0] w[c][0] + x[n][1] w[c][1]: the first attention layer's projection of the two node
// features (models/discriminator.py:56, GATConv(2, 32, heads=4).lin).  Bound by the 512 bytes written per
// node; a half-wave writes a row, a lane four columns.
__global__ __launch_bounds__(256) void k_linear_in2(int n, const float* __restrict__ x, const float* __restrict__ w,
                                                    float* __restrict__ y) {
    const int l = threadIdx.x & 31;
    float w0[4], w1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        w0[u] = w[2 * (4 * l + u)];
        w1[u] = w[2 * (4 * l + u) + 1];
    }
    const int rows_per_pass = gridDim.x * (256 / 32);
    for (int r = blockIdx.x * (256 / 32) + (threadIdx.x >> 5); r < n; r += rows_per_pass) {
        const float2 xv = *reinterpret_cast<const float2*>(x + 2 * (size_t)r);
        float4 o;
        o.x = fmaf(xv.y, w1[0], xv.x * w0[0]);
        o.y = fmaf(xv.y, w1[1], xv.x * w0[1]);
        o.z = fmaf(xv.y, w1[2], xv.x * w0[2]);
        o.w = fmaf(xv.y, w1[3], xv.x * w0[3]);
        reinterpret_cast<float4*>(y + (size_t)r * H)[l] = o;
    }
}

}  // namespace
}  // namespace aig

using namespace aig;

extern "C" int aig_linear_in2(int64_t n, const float* x, const float* w, float* y, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX, "aig_linear_in2: n=%lld out of range", (long long)n);
    if (n == 0) return AIG_OK;
    AIG_REQUIRE(x && w && y, "aig_linear_in2: null pointer");
    AIG_REQUIRE((reinterpret_cast<uintptr_t>(x) & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                "aig_linear_in2: x must be 8-byte aligned, y 16-byte aligned");
    const int64_t want = (n + 7) / 8;
    const int blocks = (int)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(k_linear_in2, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), (int)n, x, w, y);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

extern "C" size_t aig_wgrad_ws_bytes(int64_t n, int64_t F) {
    (void)n;
    const size_t blocks = F == 128 ? WG_BLOCKS_MAX : WG_SMALL_BLOCKS_MAX;
    return align_up(blocks * H * (size_t)(F > 0 ? F : 1) * 4) + 256;
}

namespace aig {
// partial sums of g^T x over the node range into `partial` ([blocks][128][F]); returns the number of blocks
static int wgrad_partials(hipStream_t s, int64_t n, int64_t F, const float* g, const float* x, float* partial, int* nb_out) {
    AIG_REQUIRE((reinterpret_cast<uintptr_t>(g) & 15) == 0 && (F != 128 || (reinterpret_cast<uintptr_t>(x) & 15) == 0),
                "aig_wgrad: rows are read as 16-byte vectors (g, and x when it has 128 columns, must be 16-byte aligned)");
    int nb;
    if (F == 128) {
        // the six-product form on the bf16 matrix cores (wgrad6.hip) unless AIG_WGRAD128_F32=1 asks for the f32 products
        static const bool f32_products = [] { const char* e = getenv("AIG_WGRAD128_F32"); return e && e[0] == '1'; }();
        if (!f32_products) {
            nb = wg_blocks(n, 32, WG_BLOCKS_MAX);
            int rc = launch_wgrad128_x6(s, (int)n, nb, g, x, partial);
            if (rc) return rc;
            *nb_out = nb;
            return AIG_OK;
        }
        AIG_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_wgrad128), (size_t)(WGRAD_LDS)));
        nb = wg_blocks(n, 2 * TILE, WG_BLOCKS_MAX);
        AIG_LAUNCH_TIMED(AIG_KT_WGRAD128, k_wgrad128, dim3(nb), dim3(512), WGRAD_LDS, s, (int)n, g_row_limit, g, x, partial);
    } else if (F == 1) {
        nb = wg_blocks(n, 64, WG_SMALL_BLOCKS_MAX);
        hipLaunchKernelGGL(k_wgrad_small<1>, dim3(nb), dim3(256), 0, s, (int)n, g_row_limit, g, x, partial);
    } else if (F == 2) {
        nb = wg_blocks(n, 64, WG_SMALL_BLOCKS_MAX);
        hipLaunchKernelGGL(k_wgrad_small<2>, dim3(nb), dim3(256), 0, s, (int)n, g_row_limit, g, x, partial);
    } else {
        nb = wg_blocks(n, 64, WG_SMALL_BLOCKS_MAX);
        hipLaunchKernelGGL(k_wgrad_small<4>, dim3(nb), dim3(256), 0, s, (int)n, g_row_limit, g, x, partial);
    }
    AIG_LAUNCH_CHECK();
    *nb_out = nb;
    return AIG_OK;
}

// dW = g1^T x1 + g2^T x2: the partial sums of both products side by side, added up by one launch (a weight that
// collects two products - the closed-form gradient penalty's - would otherwise cost two reductions and an addition).
// ws: 2 * aig_wgrad_ws_bytes(n, F) bytes.
int wgrad_sum2(int64_t n, int64_t F, const float* g1, const float* x1, const float* g2, const float* x2, float* dW, void* ws,
               size_t ws_bytes, hipStream_t s) {
    AIG_REQUIRE(n > 0 && n < INT_MAX && (F == 128 || F == 1 || F == 2 || F == 4), "wgrad_sum2: bad sizes");
    AIG_REQUIRE(g1 && x1 && g2 && x2 && dW && ws, "wgrad_sum2: null pointer");
    if (ws_bytes < 2 * aig_wgrad_ws_bytes(n, F)) return fail(AIG_EWORKSPACE, "wgrad_sum2: workspace too small");
    float* partial = static_cast<float*>(ws);
    const int elems = (int)(H * F);
    int nb = 0, nb2 = 0, rc;
    if ((rc = wgrad_partials(s, n, F, g1, x1, partial, &nb))) return rc;
    if ((rc = wgrad_partials(s, n, F, g2, x2, partial + (size_t)nb * elems, &nb2))) return rc;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((elems + 63) / 64), dim3(1024), 0, s, nb + nb2, elems, partial, dW);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}
}  // namespace aig

extern "C" int aig_wgrad(int64_t n, int64_t F, const float* g, const float* x, float* dW, void* ws,
                         size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT_MAX, "aig_wgrad: n=%lld out of range", (long long)n);
    AIG_REQUIRE(F == 128 || F == 1 || F == 2 || F == 4, "aig_wgrad: input width %lld not built (1, 2, 4 or 128)",
                (long long)F);
    AIG_REQUIRE(dW, "aig_wgrad: null output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) {
        AIG_HIP_CHECK(fill_words(dW, 0u, (size_t)H * F * 4, s));
        return AIG_OK;
    }
    AIG_REQUIRE(g && x && ws, "aig_wgrad: null pointer");
    if (ws_bytes < aig_wgrad_ws_bytes(n, F)) return fail(AIG_EWORKSPACE, "aig_wgrad: workspace too small");
    float* partial = static_cast<float*>(ws);
    int nb = 0, rc;
    if ((rc = wgrad_partials(s, n, F, g, x, partial, &nb))) return rc;
    const int elems = (int)(H * F);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((elems + 63) / 64), dim3(1024), 0, s, nb, elems, partial, dW);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}
