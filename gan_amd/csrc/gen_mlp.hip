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

// Y[:, chunk] = act(X W_chunk^T + bias_chunk) for a persistent grid of row tiles.
// FROM_X2: the input tile is relu(enc0(x)) computed on the fly from the [M,2] features.
template <bool RELU, bool FROM_X2>
__global__ __launch_bounds__(THREADS) void k_linear128(int M, const float* __restrict__ X,
                                                       const float* __restrict__ enc_w,
                                                       const float* __restrict__ enc_b, ChunkSet cs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Wl = lds;                       // [128][LDS_LD]
    float* Xl = lds + H * LDS_LD;          // [2][BM][LDS_LD]
    const Chunk ck = cs.c[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = (M + BM - 1) / BM;

    // stage the weight slice once: 128 rows x 32 float4
#pragma unroll
    for (int q = 0; q < (H * H / 4) / THREADS; ++q) {
        int idx = q * THREADS + tid, row = idx >> 5, c4 = idx & 31;
        float4 v = *reinterpret_cast<const float4*>(ck.w + static_cast<size_t>(row) * ck.ldw + c4 * 4);
        *reinterpret_cast<float4*>(Wl + row * LDS_LD + c4 * 4) = v;
    }

    float4 stage[(BM * H / 4) / THREADS];  // 8 float4 per thread = one tile
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int q = 0; q < (BM * H / 4) / THREADS; ++q) {
            int idx = q * THREADS + tid, row = idx >> 5, c4 = idx & 31;
            int gr = tile * BM + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < M) {
                if constexpr (FROM_X2) {
                    float2 xi = *reinterpret_cast<const float2*>(X + static_cast<size_t>(gr) * 2);
                    float o[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        int j = c4 * 4 + u;
                        float s = fmaf(xi.y, enc_w[2 * j + 1], xi.x * enc_w[2 * j]) + enc_b[j];
                        o[u] = fmaxf(s, 0.f);
                    }
                    v = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    v = *reinterpret_cast<const float4*>(X + static_cast<size_t>(gr) * H + c4 * 4);
                }
            }
            stage[q] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int q = 0; q < (BM * H / 4) / THREADS; ++q) {
            int idx = q * THREADS + tid, row = idx >> 5, c4 = idx & 31;
            *reinterpret_cast<float4*>(Xl + (buf * BM + row) * LDS_LD + c4 * 4) = stage[q];
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) {
        load_tile(tile);
        store_tile(0);
    }
    __syncthreads();

    const int frow = lane & 31, fh = lane >> 5;
    const float* Wfrag = Wl + (wave * 32 + frow) * LDS_LD + 4 * fh;
    float bias_v = ck.bias ? ck.bias[wave * 32 + frow] : 0.f;

    for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        const int next = tile + gridDim.x;
        if (next < ntiles) load_tile(next);

        f32x16 acc0 = {0}, acc1 = {0};
        const float* Xf0 = Xl + (buf * BM + frow) * LDS_LD + 4 * fh;
        const float* Xf1 = Xf0 + 32 * LDS_LD;
#pragma unroll
        for (int g = 0; g < H / 8; ++g) {
            // lane half fh takes k = 8g + 4fh + t at step t: A and B agree, so any k order is a valid sum
            float4 a0 = *reinterpret_cast<const float4*>(Xf0 + 8 * g);
            float4 a1 = *reinterpret_cast<const float4*>(Xf1 + 8 * g);
            float4 b = *reinterpret_cast<const float4*>(Wfrag + 8 * g);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
        }
        // epilogue: register r of a 32x32 tile is row (r&3) + 8(r>>2) + 4*fh, column = frow
        const int col = wave * 32 + frow;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
            int g0 = tile * BM + row, g1 = g0 + 32;
            float y0 = acc0[r] + bias_v, y1 = acc1[r] + bias_v;
            if (RELU) {
                y0 = fmaxf(y0, 0.f);
                y1 = fmaxf(y1, 0.f);
            }
            if (g0 < M) ck.out[static_cast<size_t>(g0) * H + col] = y0;
            if (g1 < M) ck.out[static_cast<size_t>(g1) * H + col] = y1;
        }
        if (next < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }
}

constexpr size_t LINEAR_LDS = (H * LDS_LD + 2 * BM * LDS_LD) * sizeof(float);

template <bool RELU, bool FROM_X2>
int launch_linear(hipStream_t s, int M, const float* X, const float* enc_w, const float* enc_b, const ChunkSet& cs,
                  int nchunks) {
    static bool attr_set = false;  // raising the dynamic LDS limit is idempotent; racing threads set the same value
    if (!attr_set) {
        AIG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linear128<RELU, FROM_X2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LINEAR_LDS));
        attr_set = true;
    }
    int ntiles = (M + BM - 1) / BM;
    int gx = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL((k_linear128<RELU, FROM_X2>), dim3(gx, nchunks), dim3(THREADS), LINEAR_LDS, s, M, X, enc_w,
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
    cs.c[0] = Chunk{w->enc2_w, w->enc2_b, bufA, H};
    if ((rc = launch_linear<true, true>(s, M, x, w->enc0_w, w->enc0_b, cs, 1))) return rc;
    // node_proj.0 on cat[h, z]: the z block is the bias vector zb[0]
    cs.c[0] = Chunk{w->proj0_w, zb, bufB, 2 * H};
    if ((rc = launch_linear<true, false>(s, M, bufA, nullptr, nullptr, cs, 1))) return rc;
    float* hc = h_cond ? h_cond : bufA;
    cs.c[0] = Chunk{w->proj2_w, w->proj2_b, hc, H};
    if ((rc = launch_linear<true, false>(s, M, bufB, nullptr, nullptr, cs, 1))) return rc;
    // first Linear of both edge MLPs, split into source block / target block (+ z block and bias)
    const size_t plane = static_cast<size_t>(n) * H;
    cs.c[0] = Chunk{w->sc0_w, nullptr, proj, 3 * H};
    cs.c[1] = Chunk{w->sc0_w + H, zb + H, proj + plane, 3 * H};
    cs.c[2] = Chunk{w->at0_w, nullptr, proj + 2 * plane, 3 * H};
    cs.c[3] = Chunk{w->at0_w + H, zb + 2 * H, proj + 3 * plane, 3 * H};
    if ((rc = launch_linear<false, false>(s, M, hc, nullptr, nullptr, cs, 4))) return rc;
    return AIG_OK;
}
