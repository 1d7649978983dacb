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

// z-dependent bias vectors: out[v*128 + j] = b_v[j] + sum_k W_v[j][off_v + k] * z[k], k < zdim ascending
__global__ void k_zbias(const float* __restrict__ z, const float* __restrict__ w0, const float* __restrict__ b0,
                        int ld0, int off0, const float* __restrict__ w1, const float* __restrict__ b1, int ld1,
                        int off1, const float* __restrict__ w2, const float* __restrict__ b2, int ld2, int off2,
                        int zdim, float* __restrict__ out) {
    int v = blockIdx.x, j = threadIdx.x;
    z += static_cast<size_t>(blockIdx.y) * zdim;        // blockIdx.y: the sample (one z and four vectors each)
    out += static_cast<size_t>(blockIdx.y) * 4 * H;
    if (v == 3) {  // a zero vector for layers without bias
        out[v * H + j] = 0.f;
        return;
    }
    const float* w = v == 0 ? w0 : (v == 1 ? w1 : w2);
    const float* b = v == 0 ? b0 : (v == 1 ? b1 : b2);
    int ld = v == 0 ? ld0 : (v == 1 ? ld1 : ld2), off = v == 0 ? off0 : (v == 1 ? off1 : off2);
    float acc = 0.f;
    const float* row = w + static_cast<size_t>(j) * ld + off;
    // (one chain, k ascending - the checker's order; the loads four at a time and a few groups ahead: with the width a
    //  run-time number, the plain loop is 128 dependent round trips: 21 us against 6)
    const float4* row4 = reinterpret_cast<const float4*>(row);
    const float4* z4 = reinterpret_cast<const float4*>(z);
#pragma unroll 8
    for (int q = 0; q < zdim / 4; ++q) {
        const float4 r = row4[q], zz = z4[q];
        acc = fmaf(r.x, zz.x, acc);
        acc = fmaf(r.y, zz.y, acc);
        acc = fmaf(r.z, zz.z, acc);
        acc = fmaf(r.w, zz.w, acc);
    }
    out[v * H + j] = acc + b[j];
}

// ---------------------------------------------------------------------------------------------
// Node pipeline of the generator with the weights in registers.
//
// Every wave only ever needs its own 32 output columns of a 128x128 layer: lane (col, fh) keeps
// W[col][8g + 4fh .. + 3], g = 0..15, in 64 registers per layer for the life of the block (one wave per
// SIMD: 512 registers per lane), which leaves LDS to tiles: a tile of BM rows is read as matrix-core
// fragments from one LDS buffer and the layer's output is written to another (same chunk permutation
// as aig_linear128's input tiles), either as the next layer's input or as a staging copy of an output
// that the NEXT layer's product loop streams to memory, whole 512-byte rows at a time, one 16-byte
// piece per thread every other group of products - so stores never come in a burst at the end of a
// tile, when all 256 workgroups (which run in step) would otherwise issue theirs together.
// The weights are the FIRST matrix-core operand and the tile the second, so a lane ends up with four
// consecutive output columns (8j + 4fh .. + 3 of the wave's 32, j = 0..3) of tile rows frow and
// frow + 32: 16-byte LDS writes.  The products of an output element and their order are those of
// k_linear128 below, so both give the same bits.
// ---------------------------------------------------------------------------------------------
constexpr int TILE_F = BM * H;  // floats per LDS tile

// rows of a staged tile -> global rows grow0 .. (rows past M: onto row M-1, whose values they repeat)
struct Drain {
    const float* stage;  // LDS tile (permuted chunks)
    float* gout;         // [M][128]
    int grow0, M;
};

__device__ __forceinline__ float4 drain_read(const Drain& d, int q, int tid) {
    return *reinterpret_cast<const float4*>(d.stage + (q * 8 + (tid >> 5)) * H + 4 * (tid & 31));
}
__device__ __forceinline__ void drain_write(const Drain& d, int q, int tid, float4 v) {
    const int row = q * 8 + (tid >> 5);
    const int grow = min(d.grow0 + row, d.M - 1);
    *reinterpret_cast<float4*>(d.gout + static_cast<size_t>(grow) * H + 4 * ((tid & 31) ^ (row & 15))) = v;
}

// One layer of a tile: out = act(tile W^T + bias) into the LDS tile tout.  NDRAIN of the 8 pieces per
// thread of a staged tile, from piece Q0 on, go to memory on the way.  HALF: only tile rows 0..31.
template <bool RELU, int NDRAIN, int Q0, bool HALF>
__device__ __forceinline__ void tile_layer(const float* tin, const float4 (&wr)[H / 8], const float* bias, float* tout,
                                           const Drain& dr, int tid) {
    const int lane = tid & 63, wave = tid >> 6, frow = lane & 31, fh = lane >> 5;
    const int swz = frow & 15;
    const float* Xf0 = tin + frow * H;
    const float* Xf1 = Xf0 + 32 * H;
    float4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const float4*>(bias + wave * 32 + 8 * j + 4 * fh);
    f32x16 acc0 = {0}, acc1 = {0};
    float4 a0 = *reinterpret_cast<const float4*>(Xf0 + 4 * (fh ^ swz));
    float4 a1 = make_float4(0, 0, 0, 0);
    if (!HALF) a1 = *reinterpret_cast<const float4*>(Xf1 + 4 * (fh ^ swz));
    constexpr int NRD = HALF ? 1 : 2;  // fragment reads per group of products
    __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);  // the first group's reads: a slot of their own
    constexpr int STEP = NDRAIN ? (H / 8) / NDRAIN : 1;  // a piece is read in group STEP*i, stored in the next
    float4 piece = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int g = 0; g < H / 8; ++g) {
        float4 n0 = a0, n1 = a1;
        if (g + 1 < H / 8) {
            const int pc = 4 * ((2 * (g + 1) + fh) ^ swz);
            n0 = *reinterpret_cast<const float4*>(Xf0 + pc);
            if (!HALF) n1 = *reinterpret_cast<const float4*>(Xf1 + pc);
        }
        const bool rd = NDRAIN && g % STEP == 0, wrt = NDRAIN && g % STEP == 1;
        if (wrt) drain_write(dr, Q0 + g / STEP, tid, piece);
        if (rd) piece = drain_read(dr, Q0 + g / STEP, tid);
        const float4 b = wr[g];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a0.x, acc0, 0, 0, 0);
        if (!HALF) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a0.y, acc0, 0, 0, 0);
        if (!HALF) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a0.z, acc0, 0, 0, 0);
        if (!HALF) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a0.w, acc0, 0, 0, 0);
        if (!HALF) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a1.w, acc1, 0, 0, 0);
        a0 = n0; a1 = n1;
        // issue order: the LDS reads for the next group (and a piece to drain), a drained piece's store,
        // then this group's products
        if (g + 1 < H / 8) {
            if (rd) __builtin_amdgcn_sched_group_barrier(0x100, NRD + 1, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
        } else if (rd) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (wrt) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, HALF ? 4 : 8, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // registers 4j..4j+3 of a 32x32 block: columns 8j + 4fh .. + 3, tile row frow
        float4 y0 = make_float4(acc0[4 * j] + bv[j].x, acc0[4 * j + 1] + bv[j].y, acc0[4 * j + 2] + bv[j].z,
                                acc0[4 * j + 3] + bv[j].w);
        float4 y1 = make_float4(acc1[4 * j] + bv[j].x, acc1[4 * j + 1] + bv[j].y, acc1[4 * j + 2] + bv[j].z,
                                acc1[4 * j + 3] + bv[j].w);
        if (RELU) {
            y0 = make_float4(fmaxf(y0.x, 0.f), fmaxf(y0.y, 0.f), fmaxf(y0.z, 0.f), fmaxf(y0.w, 0.f));
            y1 = make_float4(fmaxf(y1.x, 0.f), fmaxf(y1.y, 0.f), fmaxf(y1.z, 0.f), fmaxf(y1.w, 0.f));
        }
        // rows frow and frow + 32 share frow & 15: one chunk position for both
        const int pos = 4 * ((wave * 8 + 2 * j + fh) ^ swz);
        *reinterpret_cast<float4*>(tout + frow * H + pos) = y0;
        if (!HALF) *reinterpret_cast<float4*>(tout + (frow + 32) * H + pos) = y1;
    }
}

// the two variants of a layer; `half` is uniform over the block
#define TILE_LAYER(RELU, ND_FULL, ND_HALF, Q0, ...)                             \
    do {                                                                        \
        if (half) tile_layer<RELU, ND_HALF, Q0, true>(__VA_ARGS__);             \
        else tile_layer<RELU, ND_FULL, Q0, false>(__VA_ARGS__);                 \
    } while (0)

// the first NQ of the 8 pieces per thread of a staged tile (4: tile rows 0..31), outside a product loop:
// the last tile of a block
template <int NQ>
__device__ __forceinline__ void drain_all(const Drain& d, int tid) {
    float4 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = drain_read(d, q, tid);
#pragma unroll
    for (int q = 0; q < NQ; ++q) drain_write(d, q, tid, v[q]);
}

// Rows are dealt to workgroups in contiguous runs of a multiple of 32 rows (the same for all, so that all
// finish together); a run is worked through in tiles of BM = 64 rows, the last of which may be a half tile.
__host__ __device__ inline int rows_per_block(int M, int blocks) {
    const int units = (M + 31) / 32;
    return 32 * ((units + blocks - 1) / blocks);
}

__device__ __forceinline__ void load_rows(float4 (&wr)[H / 8], const float* w, int ldw, int col, int fh) {
#pragma unroll
    for (int g = 0; g < H / 8; ++g)
        wr[g] = *reinterpret_cast<const float4*>(w + static_cast<size_t>(col) * ldw + 8 * g + 4 * fh);
}

// node_enc + node_proj: h_cond = relu(W3 relu(W2' relu(W1 relu(enc0(x)) + b1) + zb) + b3), a tile at a time
// without leaving the CU; only x (8 B per row) is read from memory and only h_cond (512 B per row) written.
struct Mlp3 {
    const float* x;        // [M][2]
    const float* enc0_w;   // [128][2]
    const float* enc0_b;   // [128]
    const float* w[3];     // [128][ldw]
    const float* b[3];     // [128]
    int ldw[3];
    float* out;            // [M][128]
    // several graphs of M rows (blockIdx.y): floats from one graph to the next in x, out and the z-dependent b[1]
    long long gx, gout, gzb;
#ifdef MLP3_TIMING
    long long* dbg;
#endif
};

constexpr size_t MLP3_LDS = (3 * TILE_F + 2 * BM) * sizeof(float);

__global__ __launch_bounds__(THREADS) void k_node_mlp3(int M, int rpb, Mlp3 p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // tiles t0, t1, staging tile, x of the next tile
    {
        const long long g = blockIdx.y;
        p.x += g * p.gx; p.out += g * p.gout; p.b[1] += g * p.gzb;
    }
    float* const stage = lds + 2 * TILE_F;
    float2* const xs = reinterpret_cast<float2*>(lds + 3 * TILE_F);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    const int col = wave * 32 + frow;
    const int row_begin = blockIdx.x * rpb;
    if (row_begin >= M) return;
    const int row_end = min(M, row_begin + rpb);
    const int nt = (row_end - row_begin + BM - 1) / BM;

    float4 wr0[H / 8], wr1[H / 8], wr2[H / 8];
    load_rows(wr0, p.w[0], p.ldw[0], col, fh);
    load_rows(wr1, p.w[1], p.ldw[1], col, fh);
    load_rows(wr2, p.w[2], p.ldw[2], col, fh);

    // enc0 = Linear(2,128) + ReLU: thread (c4 = tid & 31) makes columns 4c4..4c4+3 of rows q*8 + tid/32
    const int c4 = tid & 31;
    float e0[4], e1[4], eb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        e0[u] = p.enc0_w[2 * (c4 * 4 + u)];
        e1[u] = p.enc0_w[2 * (c4 * 4 + u) + 1];
        eb[u] = p.enc0_b[c4 * 4 + u];
    }
    // x of a tile: every thread fetches row tid & 63 (four threads a row, the same value); rows past M
    // repeat row M-1, and so do all values computed from them
    auto fetch_x = [&](int row0) {
        return *reinterpret_cast<const float2*>(p.x + static_cast<size_t>(min(row0 + (tid & 63), M - 1)) * 2);
    };
    auto encode = [&](float* t) {
#pragma unroll
        for (int q = 0; q < BM / 8; ++q) {
            const int row = q * 8 + (tid >> 5);
            const float2 xv = xs[row];
            float o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = fmaxf(fmaf(xv.y, e1[u], xv.x * e0[u]) + eb[u], 0.f);
            *reinterpret_cast<float4*>(t + row * H + 4 * (c4 ^ (row & 15))) = make_float4(o[0], o[1], o[2], o[3]);
        }
    };

    xs[tid & 63] = fetch_x(row_begin);
    __syncthreads();
    encode(lds);
    Drain dr{stage, p.out, 0, M};
    bool half = false;
    for (int it = 0; it < nt; ++it) {
        float* ta = lds + (it & 1) * TILE_F;
        float* tb = lds + ((it & 1) ^ 1) * TILE_F;
        const int row0 = row_begin + it * BM;
        half = row_end - row0 <= 32;
        __syncthreads();  // the encoded tile is in ta, the last tile's output in the staging tile
#ifdef MLP3_TIMING
        long long tm[8]; long long rt0 = wall_clock64();
#define TM(i) tm[i] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#else
#define TM(i)
#endif
        TM(0);
        const float2 xn = fetch_x(row0 + BM);
        if (it) TILE_LAYER(true, 4, 4, 0, ta, wr0, p.b[0], tb, dr, tid);
        else TILE_LAYER(true, 0, 0, 0, ta, wr0, p.b[0], tb, dr, tid);
        xs[tid & 63] = xn;
        TM(1);
        __syncthreads();
        TM(2);
        if (it) TILE_LAYER(true, 4, 4, 4, tb, wr1, p.b[1], ta, dr, tid);
        else TILE_LAYER(true, 0, 0, 0, tb, wr1, p.b[1], ta, dr, tid);
        TM(3);
        __syncthreads();  // every thread has drained its pieces of the staging tile
        TM(4);
        TILE_LAYER(true, 0, 0, 0, ta, wr2, p.b[2], stage, dr, tid);
        dr.grow0 = row0;
        TM(5);
        if (it + 1 < nt) encode(tb);  // tb was last read by the second layer: free since the barrier above
#ifdef MLP3_TIMING
        TM(6);
        if (p.dbg && blockIdx.x == 0 && tid == 0) {
            for (int i = 0; i < 7; ++i) p.dbg[it * 8 + i] = tm[i];
            p.dbg[it * 8 + 7] = wall_clock64() - rt0;
        }
#endif
    }
    __syncthreads();
    if (half) drain_all<4>(dr, tid);
    else drain_all<8>(dr, tid);
}

// The first Linear of both edge MLPs on h_cond, split into source block / target block: four independent
// 128-column outputs of the same input tile.  The tile is loaded once (direct to LDS, double-buffered,
// a tile ahead), each output is staged in LDS and drained by the next output's product loop.
struct Proj4 {
    const float* x;      // [M][128]
    const float* w[4];   // [128][ldw]
    const float* b[4];   // [128] (never null: a zero vector where the layer has no bias)
    int ldw[4];
    float* out[4];       // [M][128] each
    // several graphs of M rows (blockIdx.y): floats from one graph to the next in x, every out and every b
    long long gx, gout, gzb;
};

constexpr size_t PROJ4_LDS = 4 * TILE_F * sizeof(float);

__global__ __launch_bounds__(THREADS) void k_node_proj4(int M, int rpb, Proj4 p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // input tiles x0, x1, staging tiles s0, s1
    {
        const long long g = blockIdx.y;
        p.x += g * p.gx;
#pragma unroll
        for (int i = 0; i < 4; ++i) { p.out[i] += g * p.gout; p.b[i] += g * p.gzb; }
    }
    float* const s0 = lds + 2 * TILE_F;
    float* const s1 = lds + 3 * TILE_F;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, fh = lane >> 5;
    const int col = wave * 32 + frow;
    const int row_begin = blockIdx.x * rpb;
    if (row_begin >= M) return;
    const int row_end = min(M, row_begin + rpb);
    const int nt = (row_end - row_begin + BM - 1) / BM;

    // wave w, instruction q fills tile rows 2(4q+w) and 2(4q+w)+1 (chunk c of row r at position c ^ (r & 15));
    // rows past M repeat row M-1
    const unsigned lds0 = lds_address(lds);
    auto fetch = [&](int row0, int buf) {
#pragma unroll
        for (int q = 0; q < BM / 8; ++q) {
            const int pr = 2 * (4 * q + wave);
            const int row = pr + (lane >> 5);
            const int gr = min(row0 + row, M - 1);
            const int c4 = (lane & 31) ^ (row & 15);
            load_to_lds16_uncounted(p.x + static_cast<size_t>(gr) * H + 4 * c4,
                                    __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((buf * BM + pr) * H * sizeof(float))));
        }
    };
    fetch(row_begin, 0);

    float4 wr0[H / 8], wr1[H / 8], wr2[H / 8], wr3[H / 8];
    load_rows(wr0, p.w[0], p.ldw[0], col, fh);
    load_rows(wr1, p.w[1], p.ldw[1], col, fh);
    load_rows(wr2, p.w[2], p.ldw[2], col, fh);
    load_rows(wr3, p.w[3], p.ldw[3], col, fh);

    int prev_row0 = 0;
    bool half = false;
    for (int it = 0; it < nt; ++it) {
        const float* xa = lds + (it & 1) * TILE_F;
        const int row0 = row_begin + it * BM;
        half = row_end - row0 <= 32;
        // This tile's direct loads were issued a trip ago, ahead of at least 28 stores and bias loads:
        // "at most 16 outstanding" (completion is in issue order) means it has landed.
        if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __syncthreads();  // ... for every wave; nobody still reads the other input tile, s1 is complete
        if (it + 1 < nt) fetch(row0 + BM, (it & 1) ^ 1);
        // each output is staged while the one before it (the last tile's fourth, for the first) drains;
        // a half tile's outputs have 4 pieces per thread, the last tile before it was a whole one
        if (it) TILE_LAYER(false, 8, 8, 0, xa, wr0, p.b[0], s0, (Drain{s1, p.out[3], prev_row0, M}), tid);
        else TILE_LAYER(false, 0, 0, 0, xa, wr0, p.b[0], s0, (Drain{s1, p.out[3], 0, M}), tid);
        __syncthreads();  // s0 complete, s1 drained
        TILE_LAYER(false, 8, 4, 0, xa, wr1, p.b[1], s1, (Drain{s0, p.out[0], row0, M}), tid);
        __syncthreads();
        TILE_LAYER(false, 8, 4, 0, xa, wr2, p.b[2], s0, (Drain{s1, p.out[1], row0, M}), tid);
        __syncthreads();
        TILE_LAYER(false, 8, 4, 0, xa, wr3, p.b[3], s1, (Drain{s0, p.out[2], row0, M}), tid);
        prev_row0 = row0;
    }
    __syncthreads();
    if (half) drain_all<4>(Drain{s1, p.out[3], prev_row0, M}, tid);
    else drain_all<8>(Drain{s1, p.out[3], prev_row0, M}, tid);
}

// ---------------------------------------------------------------------------------------------
// The same node pipeline on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, f32 accumulate): inputs of
// every product - activations and weights - are rounded to bf16 (round to nearest even), sums, biases
// and activations stay f32; h_cond is stored as f32, the four planes as IEEE halfs (clamped to +-32752).  One kernel for all seven 128x128 products of a
// 64-row tile: a wave owns 32 output columns of every layer and keeps their weights, packed as matrix
// fragments, in registers (7 x 32 registers); tiles go LDS -> LDS between layers as bf16, a staging
// tile turns each output into whole rows for the store.  Eight products per 32x32 block instead
// of 64, so the kernel is bound by the 1 KB of output per row, not by the matrix cores.  As in the f32
// pipeline a result depends on its own input row only (one fixed chain of operations per element), so
// equal rows give equal bits wherever they sit.
// Rows may belong to several samples of one template that differ in z alone (bulk sampling): then the
// z-dependent bias vectors are per sample, and a tile never straddles two samples.
// ---------------------------------------------------------------------------------------------
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
constexpr int BF_LD = H + 8;    // bf16 elements per LDS row: 272 bytes, rows stay 16-byte aligned
constexpr int ST_LD = H + 4;    // floats per row of the staging tile
constexpr int PIPE_LAYERS = 7;  // enc2, proj0 (node half), proj2, scorer src / tgt, attr src / tgt
constexpr int SH_LD = H + 8;    // halfs per row of a staging tile of halfs (rows stay 16-byte aligned)
// two bf16 tiles and the staging area: one f32 tile (h_cond) or two tiles of halfs (the planes, alternating)
constexpr size_t PIPE_STAGE = BM * ST_LD * sizeof(float) > 2 * BM * SH_LD * 2 ? BM * ST_LD * sizeof(float) : 2 * BM * SH_LD * 2;
constexpr int PIPE_VEC = PIPE_LAYERS * H;   // floats of one tile's bias vectors
// ... and, twice (this tile's and the next one's, fetched a tile ahead), the bias vectors and the x rows of a tile
constexpr size_t PIPE_LDS = 2 * BM * BF_LD * sizeof(__bf16) + PIPE_STAGE + 2 * (PIPE_VEC * sizeof(float) + BM * sizeof(float2));
static_assert(PIPE_STAGE % 16 == 0, "the bias vectors behind the staging area are read in 16-byte pieces");

struct PipeBf16 {
    const float* x;        // [x_rows][2]
    const float* enc0_w;   // [128][2]
    const float* enc0_b;   // [128]
    const __bf16* wpack;   // [7][4 waves][8 steps][64 lanes][8]: fragment-major weights
    const float* b[PIPE_LAYERS];     // bias vectors (never null)
    int zdep[PIPE_LAYERS];           // 1: the vector depends on z - sample s uses b + s * zstride
    int zstride;
    int rows_per_sample;   // 0: one sample; else tile t belongs to sample t*64 / rows_per_sample and reads x row (row % rows_per_sample)
    int x_rows;            // rows of x per sample (a sample's rows past them repeat the last one)
    int x_sample_rows;     // rows of x from one sample to the next (0: all samples read the same x)
    float* h_cond;         // [M][128] or null
    _Float16* out[4];      // [M][128] each, IEEE half
};
constexpr float HALF_PLANE_MAX = 32752.f;   // planes are clamped to this: the sum of two entries stays finite in half

// wpack[l][wave][s][lane][j] = bf16(W_l[wave*32 + (lane & 31)][16 s + 8 (lane >> 5) + j])
struct PackSrc {
    const float* w[PIPE_LAYERS];
    int ldw[PIPE_LAYERS];
};
__global__ void k_pack_bf16(PackSrc src, __bf16* __restrict__ wpack) {
    const int l = blockIdx.x / 8, idx = (blockIdx.x % 8) * 256 + threadIdx.x;  // 4 waves x 8 steps x 64 lanes fragments a layer
    const int lane = idx & 63, s = (idx >> 6) & 7, wave = idx >> 9;
    const float* row = src.w[l] + static_cast<size_t>(wave * 32 + (lane & 31)) * src.ldw[l] + 16 * s + 8 * (lane >> 5);
    const float4 a = *reinterpret_cast<const float4*>(row), b = *reinterpret_cast<const float4*>(row + 4);
    bf16x8 v;
    v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
    v[4] = (__bf16)b.x; v[5] = (__bf16)b.y; v[6] = (__bf16)b.z; v[7] = (__bf16)b.w;
    *reinterpret_cast<bf16x8*>(wpack + (static_cast<size_t>(l) * 2048 + idx) * 8) = v;
}

// one layer of a 64-row tile: acc = tile W^T (A operand = the wave's weight fragments, B operand = tile rows,
// so a lane ends up with four groups of four consecutive output columns of tile rows frow and frow + 32)
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half4_t to_half4(float4 y) {
    half4_t h;
    h[0] = (_Float16)__builtin_amdgcn_fmed3f(y.x, -HALF_PLANE_MAX, HALF_PLANE_MAX);
    h[1] = (_Float16)__builtin_amdgcn_fmed3f(y.y, -HALF_PLANE_MAX, HALF_PLANE_MAX);
    h[2] = (_Float16)__builtin_amdgcn_fmed3f(y.z, -HALF_PLANE_MAX, HALF_PLANE_MAX);
    h[3] = (_Float16)__builtin_amdgcn_fmed3f(y.w, -HALF_PLANE_MAX, HALF_PLANE_MAX);
    return h;
}

// TO_STAGE 1: the output goes to the staging tile as f32 rows of ST_LD floats; 2: as rows of SH_LD halfs
template <bool RELU, bool TO_BF16, int TO_STAGE>
__device__ __forceinline__ void pipe_layer(const __bf16* tin, const bf16x8 (&w)[8], const float* __restrict__ bias,
                                           __bf16* tout, float* stage, int tid) {
    const int lane = tid & 63, wave = tid >> 6, frow = lane & 31, fh = lane >> 5;
    f32x16 acc0 = {0}, acc1 = {0};
    const __bf16* r0 = tin + frow * BF_LD + 8 * fh;
    const __bf16* r1 = r0 + 32 * BF_LD;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(r0 + 16 * s);
        const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(r1 + 16 * s);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s], x0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s], x1, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = wave * 32 + 8 * q + 4 * fh;
        const float4 bv = *reinterpret_cast<const float4*>(bias + col);
        float4 y0 = make_float4(acc0[4 * q] + bv.x, acc0[4 * q + 1] + bv.y, acc0[4 * q + 2] + bv.z, acc0[4 * q + 3] + bv.w);
        float4 y1 = make_float4(acc1[4 * q] + bv.x, acc1[4 * q + 1] + bv.y, acc1[4 * q + 2] + bv.z, acc1[4 * q + 3] + bv.w);
        if (RELU) {
            y0 = make_float4(fmaxf(y0.x, 0.f), fmaxf(y0.y, 0.f), fmaxf(y0.z, 0.f), fmaxf(y0.w, 0.f));
            y1 = make_float4(fmaxf(y1.x, 0.f), fmaxf(y1.y, 0.f), fmaxf(y1.z, 0.f), fmaxf(y1.w, 0.f));
        }
        if (TO_BF16) {
            bf16x4 p0, p1;
            p0[0] = (__bf16)y0.x; p0[1] = (__bf16)y0.y; p0[2] = (__bf16)y0.z; p0[3] = (__bf16)y0.w;
            p1[0] = (__bf16)y1.x; p1[1] = (__bf16)y1.y; p1[2] = (__bf16)y1.z; p1[3] = (__bf16)y1.w;
            *reinterpret_cast<bf16x4*>(tout + frow * BF_LD + col) = p0;
            *reinterpret_cast<bf16x4*>(tout + (frow + 32) * BF_LD + col) = p1;
        }
        if (TO_STAGE == 1) {
            *reinterpret_cast<float4*>(stage + frow * ST_LD + col) = y0;
            *reinterpret_cast<float4*>(stage + (frow + 32) * ST_LD + col) = y1;
        }
        if (TO_STAGE == 2) {
            _Float16* sh = reinterpret_cast<_Float16*>(stage);
            *reinterpret_cast<half4_t*>(sh + frow * SH_LD + col) = to_half4(y0);
            *reinterpret_cast<half4_t*>(sh + (frow + 32) * SH_LD + col) = to_half4(y1);
        }
    }
}

__global__ __launch_bounds__(THREADS) void k_node_pipe_bf16(int M, PipeBf16 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pipe[];
    __bf16* const t0 = reinterpret_cast<__bf16*>(smem_pipe);
    __bf16* const t1 = t0 + BM * BF_LD;
    float* const stage = reinterpret_cast<float*>(t1 + BM * BF_LD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16x8 wr[PIPE_LAYERS][8];
#pragma unroll
    for (int l = 0; l < PIPE_LAYERS; ++l)
#pragma unroll
        for (int s = 0; s < 8; ++s)
            wr[l][s] = *reinterpret_cast<const bf16x8*>(p.wpack + ((static_cast<size_t>(l) * 4 + wave) * 8 + s) * 512 + lane * 8);
    const int c4 = tid & 31;
    float e0[4], e1[4], eb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        e0[u] = p.enc0_w[2 * (c4 * 4 + u)];
        e1[u] = p.enc0_w[2 * (c4 * 4 + u) + 1];
        eb[u] = p.enc0_b[c4 * 4 + u];
    }
    const int ntiles = (M + BM - 1) / BM;
    // rows of the staging tile -> global rows row0 .. (rows past M are not stored)
    auto drain = [&](float* gout, int row0) {
#pragma unroll
        for (int q = 0; q < BM / 8; ++q) {
            const int row = q * 8 + (tid >> 5);
            if (row0 + row < M)
                *reinterpret_cast<float4*>(gout + static_cast<size_t>(row0 + row) * H + 4 * c4) =
                    *reinterpret_cast<const float4*>(stage + row * ST_LD + 4 * c4);
        }
    };
    // ... and of a staging tile of halfs: 16 pieces of 16 bytes per row
    auto drain_half = [&](const float* st, _Float16* gout, int row0) {
        const _Float16* sh = reinterpret_cast<const _Float16*>(st);
#pragma unroll
        for (int q = 0; q < BM / 16; ++q) {
            const int row = q * 16 + (tid >> 4);
            if (row0 + row < M)
                *reinterpret_cast<uint4*>(gout + static_cast<size_t>(row0 + row) * H + 8 * (tid & 15)) =
                    *reinterpret_cast<const uint4*>(sh + row * SH_LD + 8 * (tid & 15));
        }
    };
    // What a tile needs from memory besides the weights - its seven bias vectors (three of them its sample's) and its
    // 64 rows of x - is fetched while the tile before it is worked on and handed over through the LDS: with one wave
    // per SIMD nothing else would cover the latency of a load in front of every layer.
    float* const vec = reinterpret_cast<float*>(smem_pipe + 2 * BM * BF_LD * sizeof(__bf16) + PIPE_STAGE);   // [2][7][128]
    float2* const xs = reinterpret_cast<float2*>(vec + 2 * PIPE_VEC);                                        // [2][64]
    float4 fb = make_float4(0, 0, 0, 0);
    float2 fx = make_float2(0, 0);
    auto fetch = [&](int tile) {
        const int row0 = tile * BM;
        const int sample = p.rows_per_sample > 0 ? row0 / p.rows_per_sample : 0;
        const int xrow0 = row0 - sample * p.rows_per_sample;
        if (tid < PIPE_LAYERS * 32) {
            const float* b = p.b[0];
            bool zd = p.zdep[0];
#pragma unroll
            for (int l = 1; l < PIPE_LAYERS; ++l)
                if ((tid >> 5) == l) { b = p.b[l]; zd = p.zdep[l]; }
            fb = *reinterpret_cast<const float4*>(b + (zd ? static_cast<size_t>(sample) * p.zstride : 0) + 4 * (tid & 31));
        }
        if (tid < BM)
            fx = *reinterpret_cast<const float2*>(
                p.x + (static_cast<size_t>(sample) * p.x_sample_rows + min(xrow0 + tid, p.x_rows - 1)) * 2);
    };
    auto hand_over = [&](int buf) {
        if (tid < PIPE_LAYERS * 32) *reinterpret_cast<float4*>(vec + buf * PIPE_VEC + 4 * tid) = fb;
        if (tid < BM) xs[buf * BM + tid] = fx;
    };
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    fetch(tile);
    hand_over(0);
    __syncthreads();
    for (int it = 0; tile < ntiles; ++it, tile += gridDim.x) {
        const int row0 = tile * BM;
        const int buf = it & 1;
        const bool more = tile + (int)gridDim.x < ntiles;
        if (more) fetch(tile + gridDim.x);
        const float* bias = vec + buf * PIPE_VEC;
        // enc0 = Linear(2,128) + ReLU in f32, rounded to bf16 into t0 (rows past the end repeat the last row)
#pragma unroll
        for (int q = 0; q < BM / 8; ++q) {
            const int row = q * 8 + (tid >> 5);
            const float2 xv = xs[buf * BM + row];
            bf16x4 o;
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = (__bf16)fmaxf(fmaf(xv.y, e1[u], xv.x * e0[u]) + eb[u], 0.f);
            *reinterpret_cast<bf16x4*>(t0 + row * BF_LD + 4 * c4) = o;
        }
        __syncthreads();
        pipe_layer<true, true, 0>(t0, wr[0], bias, t1, stage, tid);
        __syncthreads();
        pipe_layer<true, true, 0>(t1, wr[1], bias + H, t0, stage, tid);
        __syncthreads();
        if (p.h_cond) {
            pipe_layer<true, true, 1>(t0, wr[2], bias + 2 * H, t1, stage, tid);
            __syncthreads();
            drain(p.h_cond, row0);
        } else {
            pipe_layer<true, true, 0>(t0, wr[2], bias + 2 * H, t1, stage, tid);
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            // the two staging tiles alternate: an output is drained while the next one is formed in the other tile
            float* const st = stage + (o & 1) * (BM * SH_LD / 2);
            pipe_layer<false, false, 2>(t1, wr[3 + o], bias + (3 + o) * H, t0, st, tid);
            if (o == 3 && more) hand_over(buf ^ 1);   // (its last readers were the tile before this one)
            __syncthreads();
            drain_half(st, p.out[o], row0);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Y = act(X W^T + bias), one 128x128 layer (ACT 0 none, 1 ReLU, 2 ELU): the layer the critic calls
// (aig_linear128).  One workgroup of EIGHT waves per CU, the weight slice in registers.
//
// The LDS holds the padded weight slice only while the lanes pick their fragments out of it (lane (col, fh)
// keeps W[col][8g + 4fh .. + 3], g = 0..15, for the life of the block, as the node pipeline does); afterwards
// the same LDS is two input tiles of 64 rows.  Wave w owns output columns 32 (w & 3) .. + 31 and the 32-row
// unit (w >> 2) of every tile: two waves per SIMD, so one wave's bias / activation / address arithmetic runs
// under the other's products, and the per-tile barrier keeps the two in step (two independent 4-wave
// workgroups per CU did not: the older one wins every issue slot, finishes early and leaves the other to run
// alone at one wave per SIMD).  32-row units are dealt round-robin over the blocks (at most one unit of
// imbalance), at any moment the blocks read and write one contiguous stretch of rows.
// Memory traffic rides inside the product loop: the tile after this one arrives by direct global -> LDS loads
// issued one per group of products in the first groups (loads the compiler does not count - after a counted one
// it waits for ALL memory traffic before the next LDS read, which serialises the prefetch with the products -
// guarded by counted waits of our own: completion is in issue order), and the outputs of the tile before this
// one, held in registers, are stored one 16-byte piece per group further on.  Stores issued in a burst behind
// the last product of a tile - by every workgroup of the grid at once, since they run in step - stall the waves
// at issue: 44.5 -> 37.9 us at 100k rows with both changes, 398 -> 333 us at a million.
// The weights are the first matrix-core operand, so a lane ends up with four consecutive output columns of its
// row.  Products and their order per output element are those of the node pipeline (same bits).
// ---------------------------------------------------------------------------------------------
constexpr size_t LINEAR_R_LDS = H * LDS_LD * sizeof(float);
static_assert(LINEAR_R_LDS >= 2 * TILE_F * sizeof(float), "two input tiles must fit where the weight slice was");

constexpr int THREADS8 = 512;   // eight waves: two per SIMD

template <int ACT>
__device__ __forceinline__ float4 bias_act(const f32x16& acc, int j, const float4& bv) {
    float4 y = make_float4(acc[4 * j] + bv.x, acc[4 * j + 1] + bv.y, acc[4 * j + 2] + bv.z, acc[4 * j + 3] + bv.w);
    if (ACT == 1) {
        y = make_float4(fmaxf(y.x, 0.f), fmaxf(y.y, 0.f), fmaxf(y.z, 0.f), fmaxf(y.w, 0.f));
    } else if (ACT == 2) {  // ELU
        y.x = y.x > 0.f ? y.x : expm1f(y.x);
        y.y = y.y > 0.f ? y.y : expm1f(y.y);
        y.z = y.z > 0.f ? y.z : expm1f(y.z);
        y.w = y.w > 0.f ? y.w : expm1f(y.w);
    }
    return y;
}

template <int ACT>
__global__ __launch_bounds__(THREADS8, 1) void k_linear128(int M, const float* __restrict__ X, Chunk ck) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cw = wave & 3, uh = wave >> 2;   // column group of the wave, and which 32-row unit of a tile it takes
    const int frow = lane & 31, fh = lane >> 5;
    // 32-row units are dealt round-robin: unit u belongs to block u % gridDim.x, and a block's units 2i, 2i+1
    // are the two halves of its tile i (an odd unit out: the second half of the last tile is idle)
    const int G = gridDim.x;
    const int units = (M + 31) / 32;
    const int nu = (units - (int)blockIdx.x + G - 1) / G;
    if (nu <= 0) return;
    const int nt = (nu + 1) / 2;

    // the weight slice, padded rows (row = output column), staged once
    constexpr int WQ = (H * H / 4) / THREADS8;
    if (ck.wt) {  // stored [in][out]: transpose while staging (all loads first)
        float4 v[WQ];
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int idx = q * THREADS8 + tid, k = idx >> 5, c4 = idx & 31;
            v[q] = *reinterpret_cast<const float4*>(ck.w + static_cast<size_t>(k) * ck.ldw + c4 * 4);
        }
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int idx = q * THREADS8 + tid, k = idx >> 5, j = 4 * (idx & 31);
            lds[j * LDS_LD + k] = v[q].x;
            lds[(j + 1) * LDS_LD + k] = v[q].y;
            lds[(j + 2) * LDS_LD + k] = v[q].z;
            lds[(j + 3) * LDS_LD + k] = v[q].w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int idx = q * THREADS8 + tid, row = idx >> 5, c4 = idx & 31;
            const float4 v = *reinterpret_cast<const float4*>(ck.w + static_cast<size_t>(row) * ck.ldw + c4 * 4);
            *reinterpret_cast<float4*>(lds + row * LDS_LD + c4 * 4) = v;
        }
    }
    float4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        bv[j] = ck.bias ? *reinterpret_cast<const float4*>(ck.bias + cw * 32 + 8 * j + 4 * fh) : make_float4(0, 0, 0, 0);
    __syncthreads();
    float4 wr[H / 8];
#pragma unroll
    for (int g = 0; g < H / 8; ++g)
        wr[g] = *reinterpret_cast<const float4*>(lds + (cw * 32 + frow) * LDS_LD + 8 * g + 4 * fh);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // weights, bias: nothing of ours in flight from here
    __syncthreads();  // every fragment is in registers: the LDS is tiles from here on

    const unsigned lds0 = lds_address(lds);
    auto unit_row = [&](int k) { return ((long long)blockIdx.x + (long long)k * G) * 32; };   // block's k-th unit
    // Direct load q (0..3) of a tile: wave w fills tile rows 2(8q+w) and 2(8q+w)+1 (chunk c of row r at position
    // c ^ (r & 15)); rows past M repeat row M-1
    auto fetch1 = [&](int q, long long ra, long long rb, unsigned dst) {
        const int pr = 2 * (8 * q + wave);
        const int row = pr + (lane >> 5);
        const long long gr = min((q < 2 ? ra : rb - 32) + row, (long long)M - 1);
        const int c4 = (lane & 31) ^ (row & 15);
        load_to_lds16_uncounted(X + gr * H + 4 * c4, __builtin_amdgcn_readfirstlane(dst + (unsigned)(pr * H * sizeof(float))));
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) fetch1(q, unit_row(0), unit_row(1), lds0);

    float* const out = ck.out + cw * 32 + 4 * fh;
    float4 hold[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) hold[j] = make_float4(0, 0, 0, 0);
    float* hold_dst = nullptr;   // this lane's row of the held outputs
    const int swz = frow & 15;
    for (int it = 0; it < nt; ++it) {
        const float* Xf = lds + ((it & 1) * BM + uh * 32 + frow) * H;
        const bool mine = 2 * it + uh < nu;   // (false: second half of an odd last tile)
        // In flight from the trip before, in issue order: this tile's 4 loads, then the 4 stores of the tile
        // before it (none during the first trip).  Completion is in issue order.
        if (it <= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... for every wave; nobody still reads the other tile
        const bool more = it + 1 < nt;
        const long long na = unit_row(2 * it + 2), nb = unit_row(2 * it + 3);
        const unsigned ndst = lds0 + (unsigned)(((it & 1) ^ 1) * TILE_F * sizeof(float));
        f32x16 acc = {0};
        if (mine) {
            // fragment reads run AHEAD groups in front of the products that use them
            constexpr int AHEAD = 2;
            float4 a[H / 8];
#pragma unroll
            for (int g = 0; g < AHEAD; ++g) a[g] = *reinterpret_cast<const float4*>(Xf + 4 * ((2 * g + fh) ^ swz));
#pragma unroll
            for (int g = 0; g < H / 8; ++g) {
                if (g + AHEAD < H / 8) a[g + AHEAD] = *reinterpret_cast<const float4*>(Xf + 4 * ((2 * (g + AHEAD) + fh) ^ swz));
                // the next tile's loads ride in the first groups of products, the held tile's stores further on
                if (g < 4) {
                    if (more) fetch1(g, na, nb, ndst);
                } else if (g >= 8 && g < 12) {
                    if (hold_dst) *reinterpret_cast<float4*>(hold_dst + 8 * (g - 8)) = hold[g - 8];
                }
                const float4 b = wr[g];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a[g].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a[g].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a[g].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a[g].w, acc, 0, 0, 0);
            }
        } else {
            // (only ever the last trip: nothing to load; the held tile's stores follow the loop)
        }
        if (!mine) break;
        // registers 4j..4j+3 of a 32x32 block: columns 8j + 4fh .. + 3 of the wave's 32, row frow of the unit.
        // Rows past M land on row M-1 with the values that row has anyway (their input was row M-1), so a tile
        // always issues 4 stores, which the counted wait relies on.
#pragma unroll
        for (int j = 0; j < 4; ++j) hold[j] = bias_act<ACT>(acc, j, bv[j]);
        hold_dst = out + min(unit_row(2 * it + uh) + frow, (long long)M - 1) * H;
    }
    // the outputs still held (the last tile's; for a wave idle in the last trip, the one before)
    if (hold_dst) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(hold_dst + 8 * j) = hold[j];
    }
}

template <int ACT>
int launch_linear(hipStream_t s, int M, const float* X, const Chunk& ck) {
    AIG_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_linear128<ACT>), (size_t)(LINEAR_R_LDS)));
    const int pairs = ((M + 31) / 32 + 1) / 2;
    const int gx = pairs < 256 ? pairs : 256;   // one workgroup of eight waves per CU
    AIG_LAUNCH_TIMED(AIG_KT_LINEAR128, (k_linear128<ACT>), dim3(gx), dim3(THREADS8), LINEAR_R_LDS, s, M, X, ck);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

}  // namespace
}  // namespace aig

using namespace aig;

namespace aig {
namespace {

int check_gen_weights(const char* who, const float* x, const float* z, const aig_gen_weights* w, const float* h_cond,
                      const float* proj, const void* ws) {
    AIG_REQUIRE(x && z && w && proj && ws, "%s: null pointer", who);
    AIG_REQUIRE(w->enc0_w && w->enc0_b && w->enc2_w && w->enc2_b && w->proj0_w && w->proj0_b && w->proj2_w &&
                    w->proj2_b && w->sc0_w && w->sc0_b && w->at0_w && w->at0_b,
                "%s: null weight pointer", who);
    // the kernels read weights, biases and rows in 16-byte pieces (x in 8-byte pieces)
    auto aligned = [](const void* q, uintptr_t a) { return (reinterpret_cast<uintptr_t>(q) & (a - 1)) == 0; };
    AIG_REQUIRE(aligned(x, 8) && aligned(z, 16) && aligned(proj, 16) && aligned(h_cond, 16) && aligned(ws, 16) && aligned(w->enc2_w, 16) &&
                    aligned(w->enc2_b, 16) && aligned(w->proj0_w, 16) && aligned(w->proj2_w, 16) &&
                    aligned(w->proj2_b, 16) && aligned(w->sc0_w, 16) && aligned(w->at0_w, 16),
                "%s: x must be 8-byte aligned, z, outputs, weights and biases 16-byte aligned", who);
    AIG_REQUIRE(w->z_dim == 0 || (w->z_dim >= 4 && w->z_dim <= 1024 && w->z_dim % 4 == 0),
                "%s: z_dim=%d (a multiple of 4 in 4..1024; 0 = 128)", who, (int)w->z_dim);
    return AIG_OK;
}

// width of z, and the row lengths of the layers that take it: node_proj.0 reads [h, z], the edge MLPs [h_u, h_v, z]
int z_dim_of(const aig_gen_weights* w) { return w->z_dim > 0 ? w->z_dim : H; }
int ld_proj0(const aig_gen_weights* w) { return H + z_dim_of(w); }
int ld_edge0(const aig_gen_weights* w) { return 2 * H + z_dim_of(w); }

void launch_zbias(hipStream_t s, int samples, const float* z, const aig_gen_weights* w, float* zb) {
    // per sample: [0] node_proj.0, [1] edge_scorer.0, [2] edge_attr_pred.0, [3] zeros
    hipLaunchKernelGGL(k_zbias, dim3(4, (unsigned)samples), dim3(H), 0, s, z, w->proj0_w, w->proj0_b, ld_proj0(w), H, w->sc0_w,
                       w->sc0_b, ld_edge0(w), 2 * H, w->at0_w, w->at0_b, ld_edge0(w), 2 * H, z_dim_of(w), zb);
}

size_t proj_f32_ws(int64_t B, int64_t n) {
    if (n <= 0) return 256;
    return align_up(static_cast<size_t>(B) * n * H * 4) + align_up(static_cast<size_t>(B) * 4 * H * 4) + 256;
}

// B graphs of n rows: x advances by x_stride floats per graph (0: shared), the output planes hold `rows` rows per graph
int node_proj_f32(hipStream_t s, int B, int n, const float* x, long long x_stride, const float* z,
                  const aig_gen_weights* w, float* h_cond, float* proj, int64_t rows, void* ws) {
    Carver c(ws);
    float* bufA = c.take<float>(static_cast<size_t>(B) * n * H);  // h_cond when the caller does not want it
    float* zb = c.take<float>(static_cast<size_t>(B) * 4 * H);
    const int M = n;
    launch_zbias(s, B, z, w, zb);
    AIG_LAUNCH_CHECK();
    const int per_graph = 256 / B > 0 ? 256 / B : 1;   // workgroups per graph: about one per CU for the set
    const int rpb = rows_per_block(M, per_graph);
    const dim3 grid((M + rpb - 1) / rpb, B);

    // node_enc (Linear(2,128)+ReLU while staging, Linear(128,128)+ReLU) and node_proj (on cat[h, z]: the z
    // block of node_proj.0 is the bias vector zb[0]) in one pass over the rows
    float* hc = h_cond ? h_cond : bufA;
    {
        AIG_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_node_mlp3), (size_t)(MLP3_LDS)));
        Mlp3 p{};
        p.x = x;
        p.enc0_w = w->enc0_w;
        p.enc0_b = w->enc0_b;
        p.w[0] = w->enc2_w;  p.b[0] = w->enc2_b;  p.ldw[0] = H;
        p.w[1] = w->proj0_w; p.b[1] = zb;         p.ldw[1] = ld_proj0(w);
        p.w[2] = w->proj2_w; p.b[2] = w->proj2_b; p.ldw[2] = H;
        p.out = hc;
        p.gx = x_stride; p.gout = static_cast<long long>(n) * H; p.gzb = 4 * H;
#ifdef MLP3_TIMING
        p.dbg = reinterpret_cast<long long*>(bufA);  // timing builds are run with h_cond given
#endif
        hipLaunchKernelGGL(k_node_mlp3, grid, dim3(THREADS), MLP3_LDS, s, M, rpb, p);
        AIG_LAUNCH_CHECK();
    }
    // first Linear of both edge MLPs, split into source block / target block (+ z block and bias)
    const size_t plane = static_cast<size_t>(B) * rows * H;
    {
        AIG_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_node_proj4), (size_t)(PROJ4_LDS)));
        Proj4 q{};
        q.x = hc;
        q.w[0] = w->sc0_w;     q.b[0] = zb + 3 * H; q.out[0] = proj;
        q.w[1] = w->sc0_w + H; q.b[1] = zb + H;     q.out[1] = proj + plane;
        q.w[2] = w->at0_w;     q.b[2] = zb + 3 * H; q.out[2] = proj + 2 * plane;
        q.w[3] = w->at0_w + H; q.b[3] = zb + 2 * H; q.out[3] = proj + 3 * plane;
        for (int i = 0; i < 4; ++i) q.ldw[i] = ld_edge0(w);
        q.gx = static_cast<long long>(n) * H; q.gout = rows * H; q.gzb = 4 * H;
        hipLaunchKernelGGL(k_node_proj4, grid, dim3(THREADS), PROJ4_LDS, s, M, rpb, q);
        AIG_LAUNCH_CHECK();
    }
    return AIG_OK;
}

size_t proj_bf16_ws(int64_t samples) {
    const size_t S = static_cast<size_t>(samples > 0 ? samples : 1);
    return align_up(static_cast<size_t>(PIPE_LAYERS) * H * H * sizeof(__bf16)) + align_up(S * 4 * H * 4) + 256;
}

// `samples` graphs of n rows; every graph starts on a tile boundary of the output planes (a tile of rows belongs
// to one graph), the rows between a graph's last node and the next boundary repeat its last row.  x advances by
// x_sample_rows rows per graph (0: one template, the graphs differ in z alone).
int node_proj_bf16(hipStream_t s, int n, int samples, const float* x, int x_sample_rows, const float* z,
                   const aig_gen_weights* w, float* h_cond, void* proj, void* ws, const char* who, bool tiled = false) {
    Carver c(ws);
    __bf16* wpack = c.take<__bf16>(static_cast<size_t>(PIPE_LAYERS) * H * H);
    float* zb = c.take<float>(static_cast<size_t>(samples) * 4 * H);
    const int64_t stride = (samples > 1 || tiled) ? (static_cast<int64_t>(n) + BM - 1) / BM * BM : n;
    AIG_REQUIRE(stride * samples < INT32_MAX, "%s: %lld rows", who, (long long)(stride * samples));
    const int M = static_cast<int>(stride * samples);
    AIG_REQUIRE(samples <= 65535, "%s: at most 65535 graphs per call", who);
    launch_zbias(s, samples, z, w, zb);
    AIG_LAUNCH_CHECK();
    PackSrc src{};
    src.w[0] = w->enc2_w;      src.ldw[0] = H;
    src.w[1] = w->proj0_w;     src.ldw[1] = ld_proj0(w);
    src.w[2] = w->proj2_w;     src.ldw[2] = H;
    src.w[3] = w->sc0_w;       src.ldw[3] = ld_edge0(w);
    src.w[4] = w->sc0_w + H;   src.ldw[4] = ld_edge0(w);
    src.w[5] = w->at0_w;       src.ldw[5] = ld_edge0(w);
    src.w[6] = w->at0_w + H;   src.ldw[6] = ld_edge0(w);
    hipLaunchKernelGGL(k_pack_bf16, dim3(PIPE_LAYERS * 8), dim3(256), 0, s, src, wpack);
    AIG_LAUNCH_CHECK();
    AIG_HIP_CHECK(ensure_dynamic_lds(reinterpret_cast<const void*>(&k_node_pipe_bf16), PIPE_LDS));
    const size_t plane = static_cast<size_t>(M) * H;
    PipeBf16 p{};
    p.x = x; p.enc0_w = w->enc0_w; p.enc0_b = w->enc0_b; p.wpack = wpack;
    p.b[0] = w->enc2_b;  p.zdep[0] = 0;
    p.b[1] = zb;         p.zdep[1] = 1;
    p.b[2] = w->proj2_b; p.zdep[2] = 0;
    p.b[3] = zb + 3 * H; p.zdep[3] = 0;
    p.b[4] = zb + H;     p.zdep[4] = 1;
    p.b[5] = zb + 3 * H; p.zdep[5] = 0;
    p.b[6] = zb + 2 * H; p.zdep[6] = 1;
    p.zstride = 4 * H;
    p.rows_per_sample = samples > 1 ? static_cast<int>(stride) : 0;
    p.x_rows = n;
    p.x_sample_rows = x_sample_rows;
    p.h_cond = h_cond;
    for (int o = 0; o < 4; ++o) p.out[o] = static_cast<_Float16*>(proj) + o * plane;
    const int ntiles = (M + BM - 1) / BM;
    hipLaunchKernelGGL(k_node_pipe_bf16, dim3(ntiles < 256 ? ntiles : 256), dim3(THREADS), PIPE_LDS, s, M, p);
    AIG_LAUNCH_CHECK();
    return AIG_OK;
}

}  // namespace
}  // namespace aig

extern "C" size_t aig_gen_node_proj_ws_bytes(int64_t n) { return proj_f32_ws(1, n); }

extern "C" int aig_gen_node_proj(int64_t n, const float* x, const float* z, const aig_gen_weights* w,
                                 float* h_cond, float* proj, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT32_MAX, "aig_gen_node_proj: n=%lld out of range", (long long)n);
    if (n == 0) return AIG_OK;
    int rc = check_gen_weights("aig_gen_node_proj", x, z, w, h_cond, proj, ws);
    if (rc) return rc;
    if (ws_bytes < aig_gen_node_proj_ws_bytes(n))
        return fail(AIG_EWORKSPACE, "aig_gen_node_proj: workspace %zu < %zu", ws_bytes, aig_gen_node_proj_ws_bytes(n));
    return node_proj_f32(static_cast<hipStream_t>(stream), 1, static_cast<int>(n), x, 0, z, w, h_cond, proj, n, ws);
}

extern "C" size_t aig_gen_node_proj_bf16_ws_bytes(int64_t samples) { return proj_bf16_ws(samples); }

extern "C" int aig_gen_node_proj_bf16(int64_t n, int64_t samples, const float* x, const float* z, const aig_gen_weights* w,
                                      float* h_cond, void* proj, void* ws, size_t ws_bytes, void* stream) {
    AIG_REQUIRE(n >= 0 && samples >= 1 && n * samples < INT32_MAX, "aig_gen_node_proj_bf16: n=%lld samples=%lld out of range",
                (long long)n, (long long)samples);
    if (n == 0) return AIG_OK;
    int rc = check_gen_weights("aig_gen_node_proj_bf16", x, z, w, h_cond, static_cast<const float*>(proj), ws);
    if (rc) return rc;
    if (ws_bytes < aig_gen_node_proj_bf16_ws_bytes(samples))
        return fail(AIG_EWORKSPACE, "aig_gen_node_proj_bf16: workspace %zu < %zu", ws_bytes,
                    aig_gen_node_proj_bf16_ws_bytes(samples));
    return node_proj_bf16(static_cast<hipStream_t>(stream), static_cast<int>(n), static_cast<int>(samples), x, 0, z, w,
                          h_cond, proj, ws, "aig_gen_node_proj_bf16");
}

extern "C" int64_t aig_gen_batch_rows(int64_t n) { return (n + BM - 1) / BM * BM; }

extern "C" size_t aig_gen_node_proj_batch_ws_bytes(int64_t B, int64_t n, int bf16) {
    return bf16 ? proj_bf16_ws(B) : proj_f32_ws(B, n);
}

extern "C" int aig_gen_node_proj_batch(int64_t B, int64_t n, const float* x, int x_shared, const float* z,
                                       const aig_gen_weights* w, int bf16, void* proj, void* ws, size_t ws_bytes,
                                       void* stream) {
    AIG_REQUIRE(B >= 1 && B <= 65535 && n >= 1 && B * aig_gen_batch_rows(n) < INT32_MAX,
                "aig_gen_node_proj_batch: B=%lld n=%lld out of range", (long long)B, (long long)n);
    int rc = check_gen_weights("aig_gen_node_proj_batch", x, z, w, nullptr, static_cast<const float*>(proj), ws);
    if (rc) return rc;
    if (ws_bytes < aig_gen_node_proj_batch_ws_bytes(B, n, bf16))
        return fail(AIG_EWORKSPACE, "aig_gen_node_proj_batch: workspace %zu < %zu", ws_bytes,
                    aig_gen_node_proj_batch_ws_bytes(B, n, bf16));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (bf16) {
        // (a set of one graph still lays its rows out on the tile grid: the planes hold aig_gen_batch_rows(n) rows)
        return node_proj_bf16(s, static_cast<int>(n), static_cast<int>(B), x, x_shared ? 0 : static_cast<int>(n), z, w,
                              nullptr, proj, ws, "aig_gen_node_proj_batch", true);
    }
    return node_proj_f32(s, static_cast<int>(B), static_cast<int>(n), x, x_shared ? 0 : 2 * n, z, w, nullptr,
                         static_cast<float*>(proj), aig_gen_batch_rows(n), ws);
}

extern "C" int aig_linear128(int64_t n, const float* x, const float* w, int w_in_major, const float* bias, int act,
                             float* y, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT32_MAX, "aig_linear128: n=%lld out of range", (long long)n);
    if (n == 0) return AIG_OK;
    AIG_REQUIRE(x && w && y, "aig_linear128: null pointer");
    AIG_REQUIRE(x != y, "aig_linear128: in-place operation is not supported");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const Chunk ck{w, bias, y, H, w_in_major ? 1 : 0};
    AIG_REQUIRE(act >= 0 && act <= 2, "aig_linear128: act=%d (0 none, 1 ReLU, 2 ELU)", act);
    // the six-product form on the bf16 matrix cores (dense6.hip) unless AIG_LINEAR128_F32=1 asks for the f32 products
    static const bool f32_products = [] { const char* e = getenv("AIG_LINEAR128_F32"); return e && e[0] == '1'; }();
    if (!f32_products) return launch_linear128_x6(s, static_cast<int>(n), x, w, H, w_in_major ? 1 : 0, bias, act, y);
    if (act == 1) return launch_linear<1>(s, static_cast<int>(n), x, ck);
    if (act == 2) return launch_linear<2>(s, static_cast<int>(n), x, ck);
    return launch_linear<0>(s, static_cast<int>(n), x, ck);
}

extern "C" int aig_linear128_node_terms(int64_t n, const float* x, const float* w, int w_in_major, const float* att_src,
                                        const float* att_dst, const float* lin_edge_w, const float* att_edge, float* y,
                                        float* a_s, float* a_d, float* ce, void* stream) {
    AIG_REQUIRE(n >= 0 && n < INT32_MAX, "aig_linear128_node_terms: n=%lld out of range", (long long)n);
    AIG_REQUIRE(att_src && att_dst && lin_edge_w && att_edge && ce, "aig_linear128_node_terms: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const bool f32_products = [] { const char* e = getenv("AIG_LINEAR128_F32"); return e && e[0] == '1'; }();
    if (n == 0 || f32_products) {      // (no rows: only ce; the f32 products: the two passes one after the other)
        int rc = aig_linear128(n, x, w, w_in_major, nullptr, 0, y, stream);
        if (rc) return rc;
        return aig_gat_node_alpha(n, y, att_src, att_dst, lin_edge_w, att_edge, a_s, a_d, ce, stream);
    }
    AIG_REQUIRE(x && w && y && a_s && a_d, "aig_linear128_node_terms: null pointer");
    AIG_REQUIRE(x != y, "aig_linear128_node_terms: in-place operation is not supported");
    return launch_linear128_x6_terms(s, static_cast<int>(n), x, w, H, w_in_major ? 1 : 0, y, att_src, att_dst, lin_edge_w,
                                     att_edge, a_s, a_d, ce);
}
