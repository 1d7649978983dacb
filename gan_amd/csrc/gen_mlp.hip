// Generator node pipeline: node_enc, node_proj and the node-wise halves of the two edge MLPs,
// as chained [M,128] x [128,128] products on the f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Why f32 MFMA: the reference's pick among tied candidates (torch.topk on logits that are equal
// for equal node features) survives only if equal input rows give bit-equal outputs.  The f32 MFMA
// result is a k-ordered fmaf chain per output element, the same chain for every row and tile
// position, so equal rows stay equal whatever tile or lane they land in.
#include "common.h"

namespace aig {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int H = AIG_HIDDEN;   // 128: inner dimension of every product here
constexpr int BM = 64;          // rows per tile
constexpr int LDS_LD = H + 4;   // padded row (floats): 16-B aligned, conflict-free ds_read_b128
constexpr int THREADS = 256;

struct Chunk {          // one 128-column slice of the output
    const float* w;     // weight rows of this slice, [128 rows][ldw], row = output column
    const float* bias;  // [128] or nullptr
    float* out;         // [M][128]
    int ldw;
    int wt;             // 1: w is stored [in][out] (ldw = row length of that layout)
};
struct ChunkSet {
    Chunk c[4];
};

// z-dependent bias vectors: out[v*128 + j] = b_v[j] + sum_k W_v[j][off_v + k] * z[k]
__global__ void k_zbias(const float* __restrict__ z, const float* __restrict__ w0, const float* __restrict__ b0,
                        int ld0, int off0, const float* __restrict__ w1, const float* __restrict__ b1, int ld1,
                        int off1, const float* __restrict__ w2, const float* __restrict__ b2, int ld2, int off2,
                        float* __restrict__ out) {
    int v = blockIdx.x, j = threadIdx.x;
    const float* w = v == 0 ? w0 : (v == 1 ? w1 : w2);
    const float* b = v == 0 ? b0 : (v == 1 ? b1 : b2);
    int ld = v == 0 ? ld0 : (v == 1 ? ld1 : ld2), off = v == 0 ? off0 : (v == 1 ? off1 : off2);
    float acc = 0.f;
    const float* row = w + static_cast<size_t>(j) * ld + off;
    for (int k = 0; k < H; ++k) acc = fmaf(row[k], z[k], acc);
    out[v * H + j] = acc + b[j];
}

// Y[:, chunk] = act(X W_chunk^T + bias_chunk) for a persistent grid of row tiles (ACT 0 none, 1 ReLU, 2 ELU).
// FROM_X2: the input tile is relu(enc0(x)) computed on the fly from the [M,2] features.
//
// LDS: the weight slice (padded rows) stays for the life of the block; input tiles of BM rows are
// double-buffered.  A plain input tile goes global -> LDS by direct 16-byte loads (no register
// stop), one wave instruction per pair of 512-byte rows, so its image cannot be padded; instead
// the 16-byte chunk c of row r is kept at position c ^ (r & 15) (the permutation is applied to the
// global address of each lane), which spreads the matrix-core fragment reads - 32 rows, one chunk
// each - over all banks.  Rows past M read row M-1 and are never stored.
__device__ __forceinline__ void load_to_lds16(const float* src, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int ACT, bool FROM_X2>
__global__ __launch_bounds__(THREADS) void k_linear128(int M, const float* __restrict__ X,
                                                       const float* __restrict__ enc_w,
                                                       const float* __restrict__ enc_b, ChunkSet cs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Wl = lds;                       // [128][LDS_LD]
    float* Xl = lds + H * LDS_LD;          // [2][BM][H], chunks permuted within a row
    const Chunk ck = cs.c[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = (M + BM - 1) / BM;

    // stage the weight slice once: 128 rows x 32 float4 (row = output column)
    if (ck.wt) {  // the slice is stored [in][out]: transpose while staging (all loads first)
        float4 v[(H * H / 4) / THREADS];
#pragma unroll
        for (int q = 0; q < (H * H / 4) / THREADS; ++q) {
            const int idx = q * THREADS + tid, k = idx >> 5, c4 = idx & 31;
            v[q] = *reinterpret_cast<const float4*>(ck.w + static_cast<size_t>(k) * ck.ldw + c4 * 4);
        }
#pragma unroll
        for (int q = 0; q < (H * H / 4) / THREADS; ++q) {
            const int idx = q * THREADS + tid, k = idx >> 5, j = 4 * (idx & 31);
            Wl[j * LDS_LD + k] = v[q].x;
            Wl[(j + 1) * LDS_LD + k] = v[q].y;
            Wl[(j + 2) * LDS_LD + k] = v[q].z;
            Wl[(j + 3) * LDS_LD + k] = v[q].w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < (H * H / 4) / THREADS; ++q) {
            int idx = q * THREADS + tid, row = idx >> 5, c4 = idx & 31;
            float4 v = *reinterpret_cast<const float4*>(ck.w + static_cast<size_t>(row) * ck.ldw + c4 * 4);
            *reinterpret_cast<float4*>(Wl + row * LDS_LD + c4 * 4) = v;
        }
    }

    // plain input: wave w, instruction q fills tile rows 2(4q+w) and 2(4q+w)+1
    auto fetch = [&](int tile, int buf) {
#pragma unroll
        for (int q = 0; q < BM / 8; ++q) {
            const int pr = 2 * (4 * q + wave);
            const int row = pr + (lane >> 5);
            const int gr = min(tile * BM + row, M - 1);
            const int c4 = (lane & 31) ^ (row & 15);
            load_to_lds16(X + static_cast<size_t>(gr) * H + 4 * c4, Xl + (buf * BM + pr) * H);
        }
    };
    // FROM_X2: raw features of 8 rows per thread, encoded when they are written to LDS
    float2 xs[BM / 8];
    auto fetch_x2 = [&](int tile) {
#pragma unroll
        for (int q = 0; q < BM / 8; ++q) {
            const int gr = min(tile * BM + q * 8 + (tid >> 5), M - 1);
            xs[q] = *reinterpret_cast<const float2*>(X + static_cast<size_t>(gr) * 2);
        }
    };
    auto stash_x2 = [&](int buf) {
        const int c4 = tid & 31;
        float w0[4], w1[4], bb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            w0[u] = enc_w[2 * (c4 * 4 + u)];
            w1[u] = enc_w[2 * (c4 * 4 + u) + 1];
            bb[u] = enc_b[c4 * 4 + u];
        }
#pragma unroll
        for (int q = 0; q < BM / 8; ++q) {
            const int row = q * 8 + (tid >> 5);
            float o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = fmaxf(fmaf(xs[q].y, w1[u], xs[q].x * w0[u]) + bb[u], 0.f);
            *reinterpret_cast<float4*>(Xl + (buf * BM + row) * H + 4 * (c4 ^ (row & 15))) =
                make_float4(o[0], o[1], o[2], o[3]);
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) {
        if constexpr (FROM_X2) {
            fetch_x2(tile);
            stash_x2(0);
        } else {
            fetch(tile, 0);
        }
    }

    const int frow = lane & 31, fh = lane >> 5;
    const float* Wfrag = Wl + (wave * 32 + frow) * LDS_LD + 4 * fh;
    float bias_v = ck.bias ? ck.bias[wave * 32 + frow] : 0.f;
    const int swz = frow & 15;  // rows frow and frow+32 share it

    for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        const int next = tile + gridDim.x;
        if constexpr (FROM_X2) {
            __syncthreads();
            if (next < ntiles) fetch_x2(next);
        } else {
            // the tile's direct loads are older than the 32 output stores of the previous trip:
            // wait until only those stores may be outstanding, then meet the other waves
            // (first trip: nothing but the tile and the weight slice is in flight - wait for all)
            if (it == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (next < ntiles) fetch(next, buf ^ 1);
        }

        f32x16 acc0 = {0}, acc1 = {0};
        const float* Xf0 = Xl + (buf * BM + frow) * H;
        const float* Xf1 = Xf0 + 32 * H;
        // lane half fh takes k = 8g + 4fh + t at step t: A and B agree, so any k order is a valid sum.
        // The fragments of group g+1 are requested before the eight products of group g are issued.
        float4 a0 = *reinterpret_cast<const float4*>(Xf0 + 4 * (fh ^ swz));
        float4 a1 = *reinterpret_cast<const float4*>(Xf1 + 4 * (fh ^ swz));
        float4 b = *reinterpret_cast<const float4*>(Wfrag);
#pragma unroll
        for (int g = 0; g < H / 8; ++g) {
            float4 n0 = a0, n1 = a1, nb = b;
            if (g + 1 < H / 8) {
                const int pc = 4 * ((2 * (g + 1) + fh) ^ swz);
                n0 = *reinterpret_cast<const float4*>(Xf0 + pc);
                n1 = *reinterpret_cast<const float4*>(Xf1 + pc);
                nb = *reinterpret_cast<const float4*>(Wfrag + 8 * (g + 1));
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
            a0 = n0; a1 = n1; b = nb;
            // pin the issue order: this group's three LDS reads first, then its eight products
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
        // epilogue: register r of a 32x32 tile is row (r&3) + 8(r>>2) + 4*fh, column = frow
        const int col = wave * 32 + frow;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
            // rows past M land on row M-1 of the same column with the value that row has anyway
            // (their input was row M-1): every trip issues exactly 32 stores, which the counted
            // wait above relies on
            int g0 = min(tile * BM + row, M - 1), g1 = min(tile * BM + row + 32, M - 1);
            float y0 = acc0[r] + bias_v, y1 = acc1[r] + bias_v;
            if (ACT == 1) {
                y0 = fmaxf(y0, 0.f);
                y1 = fmaxf(y1, 0.f);
            } else if (ACT == 2) {  // ELU
                y0 = y0 > 0.f ? y0 : expm1f(y0);
                y1 = y1 > 0.f ? y1 : expm1f(y1);
            }
            ck.out[static_cast<size_t>(g0) * H + col] = y0;
            ck.out[static_cast<size_t>(g1) * H + col] = y1;
        }
        if constexpr (FROM_X2) {
            if (next < ntiles) stash_x2(buf ^ 1);
        }
    }
}

constexpr size_t LINEAR_LDS = (H * LDS_LD + 2 * BM * H) * sizeof(float);

template <int ACT, bool FROM_X2>
int launch_linear(hipStream_t s, int M, const float* X, const float* enc_w, const float* enc_b, const ChunkSet& cs,
                  int nchunks) {
    static bool attr_set = false;  // raising the dynamic LDS limit is idempotent; racing threads set the same value
    if (!attr_set) {
        AIG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear128<ACT, FROM_X2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LINEAR_LDS));
        attr_set = true;
    }
    int ntiles = (M + BM - 1) / BM;
    int gx = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL((k_linear128<ACT, FROM_X2>), dim3(gx, nchunks), dim3(THREADS), LINEAR_LDS, s, M, X, enc_w,
                       enc_b, cs);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

}  // namespace
}  // namespace aig

using namespace aig;

extern "C" size_t aig_gen_node_proj_ws_bytes(int64_t n) {
    if (n <= 0) return 256;
    return 2 * align_up(static_cast<size_t>(n) * H * 4) + align_up(3 * H * 4) + 256;
}

extern "C" int aig_gen_node_proj(int64_t n, const float* x, const float* z, const aig_gen_weights* w,
                                 float* h_cond, float* proj, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT32_MAX, "aig_gen_node_proj: n=%lld out of range", (long long)n);
    if (n == 0) return AIG_OK;
    AIG_REQUIRE(x && z && w && proj && ws, "aig_gen_node_proj: null pointer");
    AIG_REQUIRE(w->enc0_w && w->enc0_b && w->enc2_w && w->enc2_b && w->proj0_w && w->proj0_b && w->proj2_w &&
                    w->proj2_b && w->sc0_w && w->sc0_b && w->at0_w && w->at0_b,
                "aig_gen_node_proj: null weight pointer");
    if (ws_bytes < aig_gen_node_proj_ws_bytes(n))
        return fail(AIG_EWORKSPACE, "aig_gen_node_proj: workspace %zu < %zu", ws_bytes, aig_gen_node_proj_ws_bytes(n));
    hipStream_t s = static_cast<hipStream_t>(stream);
    Carver c(ws);
    float* bufA = c.take<float>(static_cast<size_t>(n) * H);
    float* bufB = c.take<float>(static_cast<size_t>(n) * H);
    float* zb = c.take<float>(3 * H);  // [0] node_proj.0, [1] edge_scorer.0, [2] edge_attr_pred.0
    const int M = static_cast<int>(n);

    hipLaunchKernelGGL(k_zbias, dim3(3), dim3(H), 0, s, z, w->proj0_w, w->proj0_b, 2 * H, H, w->sc0_w, w->sc0_b, 3 * H,
                       2 * H, w->at0_w, w->at0_b, 3 * H, 2 * H, zb);
    AIG_LAUNCH_CHECK();

    ChunkSet cs{};
    int rc;
    // node_enc: Linear(2,128)+ReLU computed while staging, Linear(128,128)+ReLU on the matrix cores
    cs.c[0] = Chunk{w->enc2_w, w->enc2_b, bufA, H, 0};
    if ((rc = launch_linear<1, true>(s, M, x, w->enc0_w, w->enc0_b, cs, 1))) return rc;
    // node_proj.0 on cat[h, z]: the z block is the bias vector zb[0]
    cs.c[0] = Chunk{w->proj0_w, zb, bufB, 2 * H, 0};
    if ((rc = launch_linear<1, false>(s, M, bufA, nullptr, nullptr, cs, 1))) return rc;
    float* hc = h_cond ? h_cond : bufA;
    cs.c[0] = Chunk{w->proj2_w, w->proj2_b, hc, H, 0};
    if ((rc = launch_linear<1, false>(s, M, bufB, nullptr, nullptr, cs, 1))) return rc;
    // first Linear of both edge MLPs, split into source block / target block (+ z block and bias)
    const size_t plane = static_cast<size_t>(n) * H;
    cs.c[0] = Chunk{w->sc0_w, nullptr, proj, 3 * H, 0};
    cs.c[1] = Chunk{w->sc0_w + H, zb + H, proj + plane, 3 * H, 0};
    cs.c[2] = Chunk{w->at0_w, nullptr, proj + 2 * plane, 3 * H, 0};
    cs.c[3] = Chunk{w->at0_w + H, zb + 2 * H, proj + 3 * plane, 3 * H, 0};
    if ((rc = launch_linear<0, false>(s, M, hc, nullptr, nullptr, cs, 4))) return rc;
    return AIG_OK;
}

extern "C" int aig_linear128(int64_t n, const float* x, const float* w, int w_in_major, const float* bias, int act,
                             float* y, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT32_MAX, "aig_linear128: n=%lld out of range", (long long)n);
    if (n == 0) return AIG_OK;
    AIG_REQUIRE(x && w && y, "aig_linear128: null pointer");
    AIG_REQUIRE(x != y, "aig_linear128: in-place operation is not supported");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ChunkSet cs{};
    cs.c[0] = Chunk{w, bias, y, H, w_in_major ? 1 : 0};
    AIG_REQUIRE(act >= 0 && act <= 2, "aig_linear128: act=%d (0 none, 1 ReLU, 2 ELU)", act);
    if (act == 1) return launch_linear<1, false>(s, static_cast<int>(n), x, nullptr, nullptr, cs, 1);
    if (act == 2) return launch_linear<2, false>(s, static_cast<int>(n), x, nullptr, nullptr, cs, 1);
    return launch_linear<0, false>(s, static_cast<int>(n), x, nullptr, nullptr, cs, 1);
}
