/*
 * aig_abi.h - C ABI of libaig_hip.so, the MI355X (gfx950) kernels behind the AIG-GAN hot path.
 *
 * The reference (tyjcj/GAN) is pure Python and has no FFI layer; its boundary for this path is
 * the Python API of models/generator.py, models/discriminator.py, models/constraints.py and
 * AIGgan.py.  Each entry point below names the reference lines whose work it takes over.  The
 * Python modules under gan_amd/ keep the reference's signatures and call these through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - `stream` is a hipStream_t; work is enqueued on it and the call returns without
 *     synchronising unless stated; no entry point allocates or frees device memory:
 *     outputs and workspaces belong to the caller (sizes from the *_ws_bytes queries);
 *   - return value 0 = success, negative = error (AIG_E*), never an exception or abort;
 *     aig_last_error() gives the message of the calling thread's last failure;
 *   - node ids, ranks and offsets are int32 inside the library (N < 2^31); edge_index tensors
 *     at the boundary keep the reference's int64.
 */
#ifndef AIG_ABI_H
#define AIG_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIG_OK 0
#define AIG_EINVAL (-1)      /* bad argument (null pointer, negative size, unsupported value) */
#define AIG_EWORKSPACE (-2)  /* workspace too small */
#define AIG_EHIP (-3)        /* a HIP runtime call or kernel launch failed */
#define AIG_ELIMIT (-4)      /* size beyond a documented limit */

#define AIG_MAX_CANDIDATE_K 1024 /* candidate window limit of the generator kernels */
#define AIG_NOALLOC_SCRATCH_BYTES (4u << 20) /* built-in scratch of aig_level_pool_fwd_noalloc, per GPU */
#define AIG_HIDDEN 128           /* hidden width the kernels are built for (reference default) */

int aig_abi_version(void);
const char* aig_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Generator: plan (depends on the template only, not on z)
 * ------------------------------------------------------------------------------------------- */

/* Processing order - models/generator.py:70-82 (_build_order) and :110-112 (types).
 *   x          float32 [n,2] template features; types[i] = (int)x[i,0]
 *   depth_kind 0: no usable node_depth (key = 0 for a PI, i otherwise); 1: float32; 2: int64
 *   order[r]   node processed r-th (stable sort by (depth, index)); rank[order[r]] = r */
size_t aig_gen_order_ws_bytes(int64_t n);
int aig_gen_order(int64_t n, const float* x, int depth_kind, const void* depth,
                  int32_t* types, int32_t* order, int32_t* rank,
                  void* ws, size_t ws_bytes, void* stream);

/* Candidate windows - models/generator.py:131-158 and the edge counts of :142-154,:178-182.
 * Slot r describes the node order[r]:
 *   want[r]   0 for a PI (not a target), 2 for an AND, 1 for anything else
 *   cnt[r]    number of candidates = min(K, #{u : rank[u] < r, type[u] != PO})
 *   cand[r*K .. r*K+cnt[r])  those candidates, the K largest ids, ascending
 *   fb[r]     largest id among nodes with rank < r when cnt[r] == 0 and the slot is a target, else -1
 *   eoff[r]   exclusive prefix sum of the edges emitted per slot (want[r] if cnt[r] > 0 or fb[r] >= 0);
 *             eoff[n] = total edge count E */
size_t aig_gen_candidates_ws_bytes(int64_t n, int64_t K);
int aig_gen_candidates(int64_t n, int64_t K, const int32_t* types, const int32_t* order, const int32_t* rank,
                       int32_t* want, int32_t* cnt, int32_t* cand, int32_t* fb, int32_t* eoff,
                       void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Generator: per-sample kernels
 * ------------------------------------------------------------------------------------------- */

/* Weights of AIGGenerator (models/generator.py:40-67), PyTorch layout [out,in] row-major fp32. */
typedef struct aig_gen_weights {
    const float *enc0_w, *enc0_b;   /* node_enc.0        [128,2],   [128] */
    const float *enc2_w, *enc2_b;   /* node_enc.2        [128,128], [128] */
    const float *proj0_w, *proj0_b; /* node_proj.0       [128,256], [128] (z_dim = 128) */
    const float *proj2_w, *proj2_b; /* node_proj.2       [128,128], [128] */
    const float *sc0_w, *sc0_b;     /* edge_scorer.0     [128,384], [128] (z_dim = 128) */
    const float *sc2_w, *sc2_b;     /* edge_scorer.2     [1,128],   [1]   */
    const float *at0_w, *at0_b;     /* edge_attr_pred.0  [128,384], [128] (z_dim = 128) */
    const float *at2_w, *at2_b;     /* edge_attr_pred.2  [1,128],   [1]   */
    int32_t z_dim;                  /* width of z (--z-dim of train.py:374 / AIGgan.py:78); 0 means 128.  With z_dim = Z
                                     * node_proj.0 is [128,128+Z] and the two edge MLPs' first layers [128,256+Z]; Z must
                                     * be a multiple of 4 in 4..1024 (weight rows are read as 16-byte vectors) */
} aig_gen_weights;

/* Node encoder + conditioning + the node-wise halves of both edge MLPs -
 * models/generator.py:115-117 and the first Linear of :94-98 / :169, split by input block:
 *   h_cond = node_proj(cat[node_enc(x), z])                                   (optional output)
 *   proj[0] = h_cond @ W_sc[:, 0:128]^T                       (source half of edge_scorer.0)
 *   proj[1] = h_cond @ W_sc[:, 128:256]^T + W_sc[:, 256:384] z + b_sc   (target half + z + bias)
 *   proj[2], proj[3] likewise for edge_attr_pred.0
 * so that the hidden layer of a (candidate c, target v) pair is relu(proj[0][c] + proj[1][v]).
 * proj is 4 consecutive [n,128] fp32 matrices; h_cond may be NULL.  f32 MFMA, k-ordered chains:
 * equal input rows give bit-equal output rows wherever they sit. */
size_t aig_gen_node_proj_ws_bytes(int64_t n);
int aig_gen_node_proj(int64_t n, const float* x, const float* z, const aig_gen_weights* w_host,
                      float* h_cond, float* proj, void* ws, size_t ws_bytes, void* stream);

/* The same node pipeline with the inputs of its seven 128x128 products - activations and weights - rounded to bf16
 * (round to nearest even) and multiplied on the bf16 matrix cores with f32 accumulation; biases and activations stay
 * f32, h_cond is stored as f32 and the four planes of proj as IEEE halfs, clamped to +-32752 so that the sum of two
 * entries stays finite (the reference computes in f32 throughout, models/generator.py:115-117,94-98: this is the
 * reduced-precision option of bulk sampling, off by default; its planes are read by aig_gen_score_half and
 * aig_gen_emit_half).  Results differ from aig_gen_node_proj by bf16 rounding (about 2^-8 relative per product
 * input) and the 2^-11 of the stored halfs; like it, a result depends on its own input row only, so equal rows give
 * bit-equal outputs and candidates with equal features still tie exactly.
 *   samples > 1: `samples` draws of z for ONE template in one call - z [samples,128], x [n,2] shared, rows laid out
 *   sample after sample, each sample starting on a multiple of 64 rows: with n64 = n rounded up to a multiple of 64,
 *   h_cond is [samples*n64,128], proj 4 x [samples*n64,128], sample s owns rows s*n64 .. s*n64+n-1 (the rows up to
 *   the next sample repeat its last row). */
size_t aig_gen_node_proj_bf16_ws_bytes(int64_t samples);
int aig_gen_node_proj_bf16(int64_t n, int64_t samples, const float* x, const float* z, const aig_gen_weights* w_host,
                           float* h_cond, void* proj, void* ws, size_t ws_bytes, void* stream);

/* Candidate scores - models/generator.py:84-99 (_score_candidates + edge_scorer):
 *   logits[r*K + i] = w2 . relu(projA[cand[r*K+i]] + projB[order[r]]) + b2,   i < cnt[r] */
int aig_gen_score(int64_t n, int64_t K, const int32_t* order, const int32_t* rank, const int32_t* cnt, const int32_t* cand,
                  const float* projA, const float* projB, const float* w2, const float* b2,
                  float* logits, void* stream);
/* The same scores from planes of IEEE halfs (aig_gen_node_proj_bf16), in packed half arithmetic: a sum
 * projA[..][j] + projB[..][j] is rounded to half, w2 is rounded to half, the products are accumulated in f32 two
 * channels at a time, channels ascending in two interleaved chains (groups of eight channels alternate between them).
 * One fixed sequence per pair, so equal candidate rows give bit-equal logits here too. */
int aig_gen_score_half(int64_t n, int64_t K, const int32_t* order, const int32_t* rank, const int32_t* cnt,
                       const int32_t* cand, const void* projA, const void* projB, const float* w2, const float* b2,
                       float* logits, void* stream);

/* Tie-exact top-k - torch.topk(logits, k, largest=True, sorted=True) as called at
 * models/generator.py:162-168, i.e. ATen's CPU kernel (native/TopKImpl.h: partial_sort when
 * k*64 <= C, else nth_element) including the order libstdc++ leaves equal values in.
 *   row r holds cnt[r] values at logits + r*ld; k = min(want[r], cnt[r]) in {0,1,2}
 *   sel[2r + j] = position (0-based, within the row) of the j-th pick, -1 when absent */
size_t aig_topk_select_ws_bytes(int64_t rows, int64_t ld);
int aig_topk_select(int64_t rows, int64_t ld, const float* logits, const int32_t* cnt, const int32_t* want,
                    int32_t* sel, void* ws, size_t ws_bytes, void* stream);

/* Edge assembly - models/generator.py:145-154 (fallback), :168-186 (picks, inversion bit, padding by
 * repeating the first pick), :196-209 (edge tensors in processing order, x[:,1] = inverted in-edges).
 *   inversion bit = (w2a . relu(projA2[src] + projB2[v]) + b2a > 0)   (sigmoid(l) > 0.5  <=>  l > 0)
 *   edge_index int64 [2,E] (row 0 sources, row 1 targets), edge_attr int64 [E], x_out float32 [n,2] */
int aig_gen_emit(int64_t n, int64_t K, int64_t E, const float* x, const int32_t* order, const int32_t* want,
                 const int32_t* cnt, const int32_t* cand, const int32_t* fb, const int32_t* eoff,
                 const int32_t* sel, const float* projA2, const float* projB2, const float* w2a, const float* b2a,
                 int64_t* edge_index, int64_t* edge_attr, float* x_out, void* stream);
/* ... with the inversion bit from planes of IEEE halfs, in the arithmetic of aig_gen_score_half (one chain) */
int aig_gen_emit_half(int64_t n, int64_t K, int64_t E, const float* x, const int32_t* order, const int32_t* want,
                      const int32_t* cnt, const int32_t* cand, const int32_t* fb, const int32_t* eoff,
                      const int32_t* sel, const void* projA2, const void* projB2, const float* w2a, const float* b2a,
                      int64_t* edge_index, int64_t* edge_attr, float* x_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Critic: graph attention layer (torch_geometric.nn.GATConv(in, 32, heads=4, edge_dim=1) as called at
 * models/discriminator.py:30-31,60-61; the layer itself is a third-party dependency of the reference).
 * ------------------------------------------------------------------------------------------- */

/* Edge lists by target and by source.  Self loops and out-of-range endpoints are dropped (GATConv
 * removes self loops and re-adds one per node); kept edges are stably sorted by target.
 *   edge_index int64 [2,E]; edge_attr float32 [E] or NULL (zeros)
 *   row_ptr[n+1], csr_src/csr_ea/csr_eid [E]: edges entering node v are row_ptr[v]..row_ptr[v+1]
 *       (row_ptr[n] = number of kept edges); csr_eid = position in the caller's edge list
 *   loop_ea[n]: attribute of the implicit self loop = mean attribute of the entering edges, 0 if none
 *   col_ptr[n+1], csc_pos/csc_dst [E]: edges leaving node u; csc_pos indexes the by-target arrays
 *   hub_tab[4*(E/33 + 2)]: the nodes with more than 32 out-edges ("hubs"; a generated graph hangs most gates on
 *       two or three of them), whose lists the backward pass cuts into chunks of 256 edges:
 *       entry 0 = {hubs, chunks of all hubs, 0, 0}, entry 1+k = {node, first position in the by-source list,
 *       fan-out, first chunk} of the k-th hub in node order */
size_t aig_gat_csr_ws_bytes(int64_t n, int64_t E);
int aig_gat_csr(int64_t n, int64_t E, const int64_t* edge_index, const float* edge_attr,
                int32_t* row_ptr, int32_t* csr_src, float* csr_ea, int32_t* csr_eid, float* loop_ea,
                int32_t* col_ptr, int32_t* csc_pos, int32_t* csc_dst, int32_t* hub_tab,
                void* ws, size_t ws_bytes, void* stream);

/* Attention logits' node terms after the dense projection xw = x W^T ([n,128], 4 heads x 32 channels):
 *   a_s[i,h] = <xw[i,h,:], att_src[h,:]>,  a_d[i,h] = <xw[i,h,:], att_dst[h,:]>        ([n,4] each)
 * and, when ce is not NULL, the edge coefficients ce[h] = <lin_edge.weight[h,:], att_edge[h,:]> ([4]; lin_edge_w and
 * att_edge are [128] = 4 heads x 32) that aig_gat_fwd / aig_gat_bwd take. */
int aig_gat_node_alpha(int64_t n, const float* xw, const float* att_src, const float* att_dst,
                       const float* lin_edge_w, const float* att_edge, float* a_s, float* a_d, float* ce,
                       void* stream);

/* Fused attention / softmax / aggregation / bias / ELU of one layer (one kernel):
 *   e = leaky_relu(a_s[src] + a_d[dst] + edge_attr * ce[h], 0.2)   ce[h] = <lin_edge.weight[h,:], att_edge[h,:]>
 *   alpha = exp(e - max) / (sum exp(e - max) + 1e-16) over the edges entering dst, self loop included
 *   out[dst] = elu(sum alpha * xw[src] + bias)                     (the F.elu of discriminator.py:60-61 is fused)
 * alpha_e [E,4] (by-target order) and alpha_loop [n,4] are kept for the backward pass. */
int aig_gat_fwd(int64_t n, const float* xw, const float* a_s, const float* a_d, const float* ce,
                const float* bias, const int32_t* row_ptr, const int32_t* csr_src, const float* csr_ea,
                const float* loop_ea, float* out, float* alpha_e, float* alpha_loop, void* stream);
/* The same launch, which also zeroes `clear_bytes` bytes at `clear` (8-byte aligned, a multiple of 8; nothing when
 * 0): what the next kernel on the stream wants cleared - the readout's table of maxima, see
 * aig_level_pool_fwd_cleared - costs a few stores per block here instead of a launch of its own (a launch that does
 * nothing takes ~4.6 us inside a replayed HIP graph).  Same reference lines as aig_gat_fwd (discriminator.py:56-61). */
int aig_gat_fwd_clearing(int64_t n, const float* xw, const float* a_s, const float* a_d, const float* ce,
                         const float* bias, const int32_t* row_ptr, const int32_t* csr_src, const float* csr_ea,
                         const float* loop_ea, float* out, float* alpha_e, float* alpha_loop, void* clear,
                         size_t clear_bytes, void* stream);

/* Backward of the same layer: given g_out = dLoss/d out, returns d_xw [n,128], d_att_src [128],
 * d_att_dst [128], d_ce [4], d_bias [128] and, when asked for (both or neither; they need lin_edge_w and att_edge),
 * d_lin_edge_w [128] and d_att_edge [128], the gradients of the two parameters ce is made of.  (Gradients w.r.t.
 * edge_attr are not produced: the only caller that differentiates edge attributes in the reference is dead code,
 * train.py:154-185.) */
size_t aig_gat_bwd_ws_bytes(int64_t n, int64_t E);
int aig_gat_bwd(int64_t n, int64_t E, const float* g_out, const float* out, const float* xw,
                const float* att_src, const float* att_dst, const float* ce, const float* a_s, const float* a_d,
                const int32_t* row_ptr, const int32_t* csr_src, const float* csr_ea, const float* loop_ea,
                const int32_t* col_ptr, const int32_t* csc_pos, const int32_t* csc_dst, const int32_t* hub_tab,
                const float* alpha_e, const float* alpha_loop, const float* lin_edge_w, const float* att_edge,
                float* d_xw, float* d_att_src, float* d_att_dst, float* d_ce, float* d_bias,
                float* d_lin_edge_w, float* d_att_edge, void* ws, size_t ws_bytes, void* stream);

/* The same backward pass for a layer whose output feeds the level readout directly (the critic's second layer,
 * discriminator.py:61-91): instead of g_out [n,128] it takes what the readout's backward pass would build it from -
 * g_r [256] = dLoss/d(readout), the plan's node_level / w, the forward readout's argmax table, L and the plan
 * header (aig_level_pool_bwd's arguments) - and forms each row of g_out on the fly.  Bit-identical to
 * aig_level_pool_bwd followed by aig_gat_bwd, without writing and re-reading the [n,128] array between them. */
int aig_gat_bwd_pooled(int64_t n, int64_t E, int64_t L, const int32_t* lvl_meta, const float* g_r,
                       const int32_t* node_level, const float* w, const int32_t* argmax, const float* out,
                       const float* xw, const float* att_src, const float* att_dst, const float* ce, const float* a_s,
                       const float* a_d, const int32_t* row_ptr, const int32_t* csr_src, const float* csr_ea,
                       const float* loop_ea, const int32_t* col_ptr, const int32_t* csc_pos, const int32_t* csc_dst,
                       const int32_t* hub_tab, const float* alpha_e, const float* alpha_loop, const float* lin_edge_w,
                       const float* att_edge, float* d_xw, float* d_att_src, float* d_att_dst, float* d_ce,
                       float* d_bias, float* d_lin_edge_w, float* d_att_edge, void* ws, size_t ws_bytes, void* stream);

/* The critic's FIRST attention layer as one kernel forward and one backward - GATConv(2, 32, heads=4, edge_dim=1)
 * followed by F.elu (models/discriminator.py:30,60) on x [n,2] (node type, inverter count), W = gat1.lin.weight
 * [128,2].  Same result as aig_linear_in2 + aig_gat_node_alpha + aig_gat_fwd up to rounding (the projected row of a
 * source is two multiply-adds away from its 8 bytes of features, so sums over a node's list are taken in feature
 * space and projected once), without any [n,128] intermediate: no projected rows, no a_s / a_d arrays.
 *   coef [20] (output of the forward call, input of the backward call): the layer's attention coefficients
 *       cs[f][h] = <W[h,:,f], att_src[h,:]> (8), cd likewise from att_dst (8), ce[h] (4)
 * aig_gat1_bwd: given g_out [n,128] = dLoss/d out, the gradients of every parameter of the layer - d_W [128,2],
 * d_att_src / d_att_dst / d_bias [128], d_ce [4] and (both or neither) d_lin_edge_w / d_att_edge [128] - from ONE
 * pass over the by-target lists (no by-source lists needed).  The node features get no gradient: the reference
 * differentiates them only on the edge-free interpolate of the gradient penalty (train.py:141-152), which does not
 * come through here. */
int aig_gat1_fwd(int64_t n, const float* x, const float* W, const float* att_src, const float* att_dst,
                 const float* lin_edge_w, const float* att_edge, const float* bias, const int32_t* row_ptr,
                 const int32_t* csr_src, const float* csr_ea, const float* loop_ea, float* coef, float* out,
                 float* alpha_e, float* alpha_loop, void* stream);
size_t aig_gat1_bwd_ws_bytes(int64_t n);
int aig_gat1_bwd(int64_t n, const float* x, const float* W, const float* att_src, const float* att_dst,
                 const float* lin_edge_w, const float* att_edge, const float* coef, const float* g_out, const float* out,
                 const int32_t* row_ptr, const int32_t* csr_src, const float* csr_ea, const float* loop_ea,
                 const float* alpha_e, const float* alpha_loop, float* d_W, float* d_att_src, float* d_att_dst,
                 float* d_ce, float* d_bias, float* d_lin_edge_w, float* d_att_edge, void* ws, size_t ws_bytes,
                 void* stream);

/* Dense projection of the critic's second attention layer, y = act(x W^T + bias) with x [n,128],
 * y [n,128] on the f32 matrix cores - torch.nn.functional.linear inside GATConv.lin (discriminator.py:61) and
 * its input gradient g W.  w_in_major = 0: w is [out,in] as PyTorch stores a Linear weight; 1: w is [in,out]
 * (so that passing the same Linear weight computes g W).  bias [128] or NULL; act 0 none, 1 ReLU, 2 ELU (the
 * F.elu of discriminator.py:61 on the edge-free path).  x and y must not overlap. */
int aig_linear128(int64_t n, const float* x, const float* w, int w_in_major, const float* bias, int act,
                  float* y, void* stream);

/* aig_linear128 (no bias, no activation) and aig_gat_node_alpha of its output in one launch: y = x W^T together with
 * a_s, a_d [n,4] and ce [4] of the attention layer the projection belongs to - GATConv.lin and the (x_src * att_src)
 * .sum(-1) / (x_dst * att_dst).sum(-1) that follow it inside GATConv.forward (models/discriminator.py:61).  A wave
 * holds one head of its output rows when it stores them, so the node terms need no pass of their own over y.
 * Same y; a_s, a_d, ce as aig_gat_node_alpha gives them (the sums are taken in the same order). */
int aig_linear128_node_terms(int64_t n, const float* x, const float* w, int w_in_major, const float* att_src,
                             const float* att_dst, const float* lin_edge_w, const float* att_edge, float* y,
                             float* a_s, float* a_d, float* ce, void* stream);

/* Projection of the critic's FIRST attention layer, y = x W^T with x [n,2] (node type, inverter count), w [128,2],
 * y [n,128] - torch.nn.functional.linear inside GATConv(2, 32, heads=4).lin (models/discriminator.py:31,56). */
int aig_linear_in2(int64_t n, const float* x, const float* w, float* y, void* stream);

/* Critic head: score = out(mlp(r)) for the pooled graph vector r [256] - Linear(256,128), ReLU, Linear(128,64),
 * ReLU, Linear(64,1) (models/discriminator.py:36-40,98-100) - in one launch, and its backward pass in one launch
 * (torch runs three GEMV launches forward and a dozen small launches backward, ten times per training step).
 *   w1 [128,256], b1 [128], w2 [64,128], b2 [64], w3 [64], b3 [1] as the Linear layers store them
 *   act [192]: the activations of the two hidden layers, kept for the backward calls
 * aig_head_bwd: d_score [1] on the device; any of the seven outputs may be NULL.  With d_score = 1 its d_r is the
 * input gradient ds/dr the gradient penalty starts from (train.py:143-150 on the head).
 * aig_head_input_grad_wgrad: gradient of <ds/dr, u> w.r.t. w1, w2, w3 (u [256]; the biases get none) - the head's
 * share of the second-order term of the gradient penalty, which autograd obtains by differentiating twice. */
int aig_head_fwd(const float* r, const float* w1, const float* b1, const float* w2, const float* b2,
                 const float* w3, const float* b3, float* score, float* act, void* stream);
int aig_head_bwd(const float* d_score, const float* r, const float* w1, const float* w2, const float* w3,
                 const float* act, float* d_r, float* d_w1, float* d_b1, float* d_w2, float* d_b2, float* d_w3,
                 float* d_b3, void* stream);
int aig_head_input_grad_wgrad(const float* u, const float* w1, const float* w2, const float* w3, const float* act,
                              float* d_w1, float* d_w2, float* d_w3, void* stream);
/* The same with every output multiplied by out_scale after the fact (one more f32 rounding per element - what
 * `gp_lambda * gradient` of train.py:258 does to autograd's gradient - without a launch for it). */
int aig_head_input_grad_wgrad_scaled(const float* u, const float* w1, const float* w2, const float* w3, const float* act,
                                     float* d_w1, float* d_w2, float* d_w3, float out_scale, void* stream);

/* Weight gradient of a dense projection xw = x W^T over the nodes (the reduction torch.nn.functional.linear's
 * backward performs for gat{1,2}.lin at discriminator.py:60-61): dW[j][f] = sum_n g[n][j] * x[n][f],
 * g [n,128], x [n,F], dW [128,F]; F in {1, 2, 4, 128}. */
size_t aig_wgrad_ws_bytes(int64_t n, int64_t F);
int aig_wgrad(int64_t n, int64_t F, const float* g, const float* x, float* dW, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Critic: level-wise readout - models/discriminator.py:64-91
 * ------------------------------------------------------------------------------------------- */

/* Groups nodes by depth level 0..L (negative depth = ignored node, as for padding).
 *   depth int64 [n] or NULL (every node in level 0); L = largest level, read by the caller
 *   perm[n]: node ids stably sorted by level, ignored nodes last; lvl_ptr[L+2]: start of each level in perm
 *   node_level[n]: level or -1; w[n] = 1 / ((L+1) * size of the node's level), 0 for ignored nodes */
size_t aig_level_plan_ws_bytes(int64_t n, int64_t L);
int aig_level_plan(int64_t n, int64_t L, const int64_t* depth, int32_t* perm, int32_t* lvl_ptr,
                   int32_t* node_level, float* w, void* ws, size_t ws_bytes, void* stream);

/* The same plan without the caller's host read of the deepest level (discriminator.py:76 reads it with .item()):
 * the level count is found on the device and left in a four-word header
 *   lvl_meta[0] = L, the deepest level (0 when no node carries a valid depth: every node is then pooled in level 0,
 *                 discriminator.py:68-72), lvl_meta[1] = nodes with a level, lvl_meta[2] = flags (bit 0: no valid
 *                 depth; bit 1: a depth exceeded L_cap - such nodes are left out, the caller must re-plan), lvl_meta[3] = 0
 * L_cap bounds the level the caller expects (sizes: lvl_ptr[L_cap+2], the argmax table and workspaces of the
 * consumers are sized with L_cap in place of L).  The consumers below take the header as `lvl_meta`; with a header
 * their `L` (and `nvalid` = n) arguments are capacities, with NULL they are the values themselves.  A sequence of
 * calls built this way has fixed launch shapes for a given (n, L_cap) and can be recorded in a HIP graph and
 * replayed on new depth vectors of that shape. */
int aig_level_plan_dyn(int64_t n, int64_t L_cap, const int64_t* depth, int32_t* perm, int32_t* lvl_ptr,
                       int32_t* node_level, float* w, int32_t* lvl_meta, void* ws, size_t ws_bytes, void* stream);

/* out[0:128] = mean over levels 0..L of the level means, out[128:256] = mean over levels of the level
 * maxima (an empty level counts as a zero vector); argmax[l*128+f] = smallest node id attaining the
 * maximum of feature f in level l (-1 for an empty level), which is where the gradient goes.
 * nvalid = number of nodes with a level (= lvl_ptr[L+1]). */
size_t aig_level_pool_fwd_ws_bytes(int64_t n, int64_t L);
int aig_level_pool_fwd(int64_t n, int64_t nvalid, int64_t L, const int32_t* lvl_meta, const float* h,
                       const int32_t* perm, const int32_t* lvl_ptr, const int32_t* node_level, const float* w,
                       float* out, int32_t* argmax, void* ws, size_t ws_bytes, void* stream);
/* The same readout (discriminator.py:63-91) when the first aig_level_pool_table_bytes(L) bytes of `ws` - its table
 * of packed level maxima - are already zero when the call's first kernel runs, cleared by an earlier kernel on the
 * same stream (aig_gat_fwd_clearing, aig_gp_forward_clearing): one launch less. */
size_t aig_level_pool_table_bytes(int64_t L);
int aig_level_pool_fwd_cleared(int64_t n, int64_t nvalid, int64_t L, const int32_t* lvl_meta, const float* h,
                               const int32_t* perm, const int32_t* lvl_ptr, const int32_t* node_level, const float* w,
                               float* out, int32_t* argmax, void* ws, size_t ws_bytes, void* stream);

/* The same readout for callers that cannot allocate device memory or are not allowed to - code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE (which allocates
 * freely through torch): same arguments and results as aig_level_pool_fwd, without the workspace.  The table of
 * level maxima and the block partials live in an array built into the library, AIG_NOALLOC_SCRATCH_BYTES (4 MiB) on
 * every GPU the library runs on, held until the process ends; the call performs no hipMalloc / hipFree and calls no
 * allocating routine.  Limit: aig_level_pool_fwd_ws_bytes(n, L) <= AIG_NOALLOC_SCRATCH_BYTES (e.g. one million nodes
 * with up to 2,000 levels, or 100,000 nodes with 3,900); beyond it the call returns AIG_ELIMIT and touches nothing.
 * All calls on one GPU share that GPU's array: it keeps the last call's contents (the call clears what it needs
 * zeroed itself), and two calls whose kernels overlap in time on one GPU - different streams, different host
 * threads - race on it.  Serialise them, or use aig_level_pool_fwd with separate workspaces.  The FIRST call on a
 * device looks up the array's device address (hipGetSymbolAddress, kept per device afterwards) and must not come
 * while `stream` is being recorded into a graph - it is refused there with AIG_EINVAL; later calls can be recorded. */
int aig_level_pool_fwd_noalloc(int64_t n, int64_t nvalid, int64_t L, const int32_t* lvl_meta, const float* h,
                               const int32_t* perm, const int32_t* lvl_ptr, const int32_t* node_level, const float* w,
                               float* out, int32_t* argmax, void* stream);

/* Backward of the readout (what autograd derives from discriminator.py:64-91): with g = dLoss/d out [256],
 *   d_h[i,c] = w[i] g[c] + [argmax[level(i),c] == i] g[128+c] / (L+1)
 * and its adjoint (needed when the gradient penalty differentiates a second time): for gg [n,128]
 *   out[c] = sum_i w[i] gg[i,c],   out[128+c] = sum_i [argmax[level(i),c] == i] gg[i,c] / (L+1). */
int aig_level_pool_bwd(int64_t n, int64_t L, const int32_t* lvl_meta, const float* g, const int32_t* node_level,
                       const float* w, const int32_t* argmax, float* d_h, void* stream);
size_t aig_level_pool_bwd_adj_ws_bytes(int64_t n);
int aig_level_pool_bwd_adj(int64_t n, int64_t L, const int32_t* lvl_meta, const float* gg, const int32_t* node_level,
                           const float* w, const int32_t* argmax, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * One critic iteration's update - opt_D.zero_grad(); loss_D.backward(); opt_D.step() of train.py:253,265-266 once
 * the three terms' gradients exist: grad = g0 + g1 + g2 (in that order; a NULL term is skipped) and the Adam update
 * (torch.optim.Adam without weight decay / amsgrad, train.py:221) of param, exp_avg, exp_avg_sq, in one launch.
 * chunks: nchunks device records {float* param, *grad; const float* g0, *g1, *g2; float* exp_avg, *exp_avg_sq;
 * int64 count} (64 bytes each), every record at most aig_adam_chunk_elems() consecutive elements of one tensor;
 * step_ptrs: nsteps device pointers to the optimizer's float step counters, all set to step_value (the count
 * INCLUDING this update); bias_correction1 = 1 - beta1^step, bias_correction2_sqrt = sqrt(1 - beta2^step).
 * mode 3: both halves; mode 1: the sums alone (data-parallel ranks exchange the gradients next); mode 2: the update
 * alone, from grad as it stands.
 * ------------------------------------------------------------------------------------------- */
int aig_adam_chunk_elems(void);
int aig_adam_sum3(int64_t nchunks, const void* chunks, int64_t nsteps, const void* step_ptrs, float step_value, double lr,
                  double beta1, double beta2, double eps, float bias_correction1, float bias_correction2_sqrt, int mode,
                  void* stream);

/* The scalars a training step logs (train.py:263-272,289-296, 302-311) from the device scalars the terms leave:
 * out8 = {D_real, D_fake, gap = D_real - D_fake, gp, grad_norm, loss_D = (D_fake - D_real) + gp_lambda gp,
 * loss_G = -D_raw + lambda_cons pen, pen = pen_scale * mean over the B graphs of pen9[g][8]} (pen9: the [B,9] rows of
 * aig_penalties_batch; pen_scale: the save step's validation factor, 1 otherwise).  One launch for a dozen. */
int aig_step_metrics(int64_t B, const float* d_real, const float* d_fake, const float* gp, const float* grad_norm,
                     const float* d_raw, const float* pen9, float gp_lambda, float lambda_cons, float pen_scale,
                     float* out8, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stable sort of (key, value) pairs by bits [0, bits) of the key - no counterpart in the reference (torch's sorts,
 * scatters and PyG's index bookkeeping stand where the library sorts: constraints.py:88-112, discriminator.py:83-96,
 * GATConv's aggregation).  The building block of the edge lists by target / by source, the level plans and the
 * repair's pair keys, exported for its tests: keys are 4 or 8 bytes wide (key_bytes), values int32; inputs are left
 * as they were; input and output buffers must differ.  Equal keys keep their input order.
 * ------------------------------------------------------------------------------------------- */
size_t aig_sort_pairs_ws_bytes(int64_t items, int key_bytes, int bits);
int aig_sort_pairs(int64_t items, int key_bytes, int bits, const void* keys_in, void* keys_out, const int32_t* values_in,
                   int32_t* values_out, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gradient penalty on the edge-free interpolate - gradient_penalty(use_edges=False), train.py:133-152, through
 * AIGDiscriminator.forward on an empty edge list (discriminator.py:42-100), in closed form.  Without edges a
 * GAT layer is elu(x W^T + b), so with r = readout(h2), s = head(r):
 *   aig_gp_forward     x^ = alpha x_real + (1-alpha) x_fake [n,2]; h1 = elu(x^ W1^T + b1); h2 = elu(h1 W2^T + b2)
 *                      (the caller runs aig_level_pool_fwd on h2 and its own head, and hands back g_r = ds/dr [256])
 *   aig_gp_input_grad  g_z2 = (readout^T g_r) * elu'(z2); g_h1 = g_z2 W2; g_z1 = g_h1 * elu'(z1); g_x = g_z1 W1 [n,2];
 *                      *norm_sq = sum g_x^2
 *   aig_gp_scalars     out3 = {norm = sqrt(norm_sq), gp = (norm - 1)^2, coef = scale * 2 (norm - 1) / norm}: the last
 *                      lines of train.py:150-151 and the factor d gp / d norm / norm every parameter gradient of gp
 *                      carries, times the caller's scale (gp_lambda); coef = 0 at norm = 0 (the zero subgradient
 *                      torch.norm has there).  One launch instead of ten scalar ones
 *   aig_gp_param_grad  gradient of gp w.r.t. W1 [128,2], b1, W2 [128,128], b2 and ubar_gr = d gp / d g_r [256]
 *                      (the caller back-propagates ubar_gr through its head to get the head's weight gradients).
 * W1/b1 = gat1.lin.weight / gat1.bias, W2/b2 = gat2.lin.weight / gat2.bias; alpha, norm_sq, coef are device scalars;
 * lvl_meta = header of a plan from aig_level_plan_dyn, or NULL when L is the plan's deepest level itself.
 * n_rows_dev (may be NULL): a device int holding the number of rows really present, for callers whose row count
 * only the device knows (aig_gp_rows): buffers and launches are sized by n, the kernels work on min(n, *n_rows_dev)
 * rows, and rows past that count must hold neutral values (no level, weight 0) - results do not depend on it.
 *
 *   aig_gp_rows        the DISTINCT rows of this computation (no counterpart in the reference: train.py:133-152 runs
 *                      the critic over every node).  Nodes of one level with the same (x_real, x_fake) features go
 *                      through both layers as identical rows, and only the first of them (lowest id) can hold one of
 *                      its level's maxima.  Per such cell this emits its first node as a row of multiplicity 1 and,
 *                      for cells of m > 1 nodes, its second node as a row standing for the other m - 1; rows come in
 *                      node order: xr_rows / xf_rows [n,2], depth_rows int64 [n] (the row's level; -1 past the last
 *                      row - feed it to aig_level_plan_dyn), w_fwd = m w and w_bwd = sqrt(m) w with w the node's
 *                      weight in the FULL plan (nvalid, lvl_meta, perm, node_level, w: that plan's sizes and arrays;
 *                      L: its level capacity), *n_rows.
 *                      With w_fwd as the readout's weights in aig_level_pool_fwd and w_bwd as `w` of
 *                      aig_gp_input_grad / aig_gp_param_grad the penalty, its norm and every gradient equal those of
 *                      the run over all nodes (a row's backward chain is linear in the readout's gradient, so it
 *                      carries sqrt(m) and every sum over rows of a product of two chains carries m).  More than 256
 *                      distinct feature pairs: the surplus nodes stay rows of their own.
 * ------------------------------------------------------------------------------------------- */
size_t aig_gp_rows_ws_bytes(int64_t n, int64_t L);
int aig_gp_rows(int64_t n, int64_t nvalid, int64_t L, const int32_t* lvl_meta, const int32_t* perm, const float* x_real,
                const float* x_fake, const int32_t* node_level, const float* w,
                float* xr_rows, float* xf_rows, int64_t* depth_rows, float* w_fwd, float* w_bwd, int32_t* n_rows,
                void* ws, size_t ws_bytes, void* stream);
int aig_gp_forward(int64_t n, const int32_t* n_rows_dev, const float* x_real, const float* x_fake, const float* alpha, const float* W1,
                   const float* b1, const float* W2, const float* b2, float* xhat, float* h1, float* h2, void* stream);
/* aig_gp_forward that also zeroes `clear_bytes` bytes at `clear` for the readout that follows it (as
 * aig_gat_fwd_clearing does; the same lines of train.py:133-152) */
int aig_gp_forward_clearing(int64_t n, const int32_t* n_rows_dev, const float* x_real, const float* x_fake, const float* alpha,
                            const float* W1, const float* b1, const float* W2, const float* b2, float* xhat, float* h1,
                            float* h2, void* clear, size_t clear_bytes, void* stream);
int aig_gp_scalars(const float* norm_sq, float scale, float* out3, void* stream);
size_t aig_gp_input_grad_ws_bytes(int64_t n);
int aig_gp_input_grad(int64_t n, const int32_t* n_rows_dev, int64_t L, const int32_t* lvl_meta, const float* g_r, const int32_t* node_level,
                      const float* w, const int32_t* argmax, const float* W1, const float* W2, const float* h1, const float* h2,
                      float* g_z2, float* g_h1, float* g_z1, float* g_x, float* norm_sq,
                      void* ws, size_t ws_bytes, void* stream);
/* aig_gp_input_grad followed by aig_gp_scalars(norm_sq, grad_scale, out3) with one launch less: the launch that adds
 * up |g_x|^2 also forms the three scalars (same sums, same order, same bits; train.py:150-151). */
int aig_gp_input_grad_scalars(int64_t n, const int32_t* n_rows_dev, int64_t L, const int32_t* lvl_meta, const float* g_r,
                              const int32_t* node_level, const float* w, const int32_t* argmax, const float* W1, const float* W2,
                              const float* h1, const float* h2, float* g_z2, float* g_h1, float* g_z1, float* g_x,
                              float* norm_sq, float grad_scale, float* out3, void* ws, size_t ws_bytes, void* stream);
size_t aig_gp_param_grad_ws_bytes(int64_t n);
int aig_gp_param_grad(int64_t n, const int32_t* n_rows_dev, int64_t L, const int32_t* lvl_meta, const float* coef, const float* g_r,
                      const int32_t* node_level, const float* w, const int32_t* argmax, const float* W1, const float* W2, const float* xhat,
                      const float* h1, const float* h2, const float* g_z2, const float* g_h1, const float* g_z1,
                      const float* g_x, float* dW1, float* db1, float* dW2, float* db2, float* ubar_gr,
                      void* ws, size_t ws_bytes, void* stream);
/* The same with dW1, db1, dW2, db2 multiplied by out_scale after the fact (one more f32 rounding per element, as
 * `gp_lambda * gradient` of train.py:258 gives; ubar_gr, an intermediate of the head's share, is not scaled). */
int aig_gp_param_grad_scaled(int64_t n, const int32_t* n_rows_dev, int64_t L, const int32_t* lvl_meta, const float* coef,
                             const float* g_r, const int32_t* node_level, const float* w, const int32_t* argmax, const float* W1,
                             const float* W2, const float* xhat, const float* h1, const float* h2, const float* g_z2,
                             const float* g_h1, const float* g_z1, const float* g_x, float* dW1, float* db1, float* dW2,
                             float* db2, float* ubar_gr, float out_scale, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Structural constraints - models/constraints.py.  One graph per call, n > 0 and E > 0 (the callers
 * handle the empty cases, which need no kernel); edge_index int64 [2,E], x float32 [n,2].
 * ------------------------------------------------------------------------------------------- */

/* Repair - enforce_aig_constraints (constraints.py:508-669, strict=False).
 *   edge_attr int64 [E] inversion values (NULL = zeros); depth_kind 0 none / 1 float32 / 2 int64 node_depth
 *   of the INPUT graph (used to rank a PO's candidate drivers)
 * Cleaning: self loops dropped, repeated (u,v) merged (first value kept, raised to 1 when it is 0 and a
 * repeat carries 1), out-of-range endpoints dropped, edges with pos[u] >= pos[v] dropped (pos = rank in
 * PI < AND < PO order).  Selection per target in id order: a PO keeps the legal driver minimising
 * (-is_and, -int(depth), id); an AND keeps its two smallest-pos legal fan-ins, or - with fewer than two -
 * the two smallest-pos PI/AND nodes other than itself (a kept legal fan-in keeps its inversion value).
 * Outputs: out_edge_index rows of stride `cap` (>= 2n), out_edge_attr [cap], *out_count = edges written;
 * node_depth int64 [n] = breadth-first level from the PIs over the CLEANED edges (-1 unreachable), or NULL
 * to leave that walk out (AIGgan.py:105-110 calls resync_attrs right after the repair, which replaces it);
 * x_out [n,2] = (x[:,0], min(2, sum of merged inversion values entering the node over the cleaned edges)). */
size_t aig_enforce_ws_bytes(int64_t n, int64_t E);
int aig_enforce(int64_t n, int64_t E, const float* x, const int64_t* edge_index, const int64_t* edge_attr,
                int depth_kind, const void* depth, int64_t* out_edge_index, int64_t cap, int64_t* out_edge_attr,
                int32_t* out_count, int64_t* node_depth, float* x_out, void* ws, size_t ws_bytes, void* stream);

/* Violation counts - aig_constraint_penalties (constraints.py:253-366), default weights.
 * out9 = pi_in, po_out, backedge, selfloop, dup_fanin, indeg_mse, fanin_type, reach_violation, total.
 * Edges with an endpoint outside [0,n) cannot be indexed by the reference and are left out of every count. */
size_t aig_penalties_ws_bytes(int64_t n, int64_t E);
int aig_penalties(int64_t n, int64_t E, const float* x, const int64_t* edge_index, float* out9,
                  void* ws, size_t ws_bytes, void* stream);

/* Verdicts of validate_strict_aig (constraints.py:375-505); the caller turns them into the message list.
 *   eflag[E]  bit 0 self loop, bit 1 repeat of an earlier (u,v), bit 2 enters a PI, bit 3 leaves a PO,
 *             bit 4 source type not PI/AND for an AND/PO target
 *   ncode[n]  0 fine, 1 unknown type, 2 PO with indegree != 1, 3 AND with indegree != 2, 4 AND fed twice by one node
 *   indeg[n]; pred2[2n] first two sources entering each node in edge order (-1 when absent)
 *   on_path[n] 1 when reached forward from a PI and backward from a PO
 *   summary[0] graph has a cycle (Kahn), summary[1] nodes not on a PI->PO path */
size_t aig_validate_ws_bytes(int64_t n, int64_t E);
int aig_validate(int64_t n, int64_t E, const float* x, const int64_t* edge_index, uint8_t* eflag, uint8_t* ncode,
                 int32_t* indeg, int32_t* pred2, uint8_t* on_path, int32_t* summary,
                 void* ws, size_t ws_bytes, void* stream);

/* Attribute resync of the sampling CLI - resync_attrs (AIGgan.py:29-62).
 *   inv_cnt[v] = number of entering edges whose inversion value is 1 (not clamped)
 *   depth[v]   = longest-path level over the in-range edges, all zeros when the graph has a cycle */
size_t aig_resync_ws_bytes(int64_t n, int64_t E);
int aig_resync(int64_t n, int64_t E, const int64_t* edge_index, const int64_t* edge_attr,
               int64_t* inv_cnt, int64_t* depth, void* ws, size_t ws_bytes, void* stream);
/* The same results (AIGgan.py:29-62) for a caller that also hands over the node features x [n,2] (column 0 =
 * node type), as resync_attrs has them in data.x: when every in-range edge runs from a lower to a higher position
 * in the PI / AND / PO order - what enforce_aig_constraints emits (constraints.py:558-628) unless the graph has
 * fewer than two PIs - that order is a topological one and the levels come from one sweep over the positions
 * instead of a walk level by level; any other graph (a cycle included) takes the walk of aig_resync. */
size_t aig_resync_typed_ws_bytes(int64_t n, int64_t E);
int aig_resync_typed(int64_t n, int64_t E, const float* x, const int64_t* edge_index, const int64_t* edge_attr,
                     int64_t* inv_cnt, int64_t* depth, void* ws, size_t ws_bytes, void* stream);

/* ---- per-kernel GPU time (a measurement aid; no counterpart in the reference) --------------------------
 * aig_kernel_timing(mask) selects the kernels to time (bit k = kernel k of the list below, 0 = none, the
 * initial state); going from none to some forgets earlier records.  Every launch of a selected kernel carries a
 * start and a stop HIP event of its own, stamped by the dispatch on the stream it is launched on (launches that
 * are being recorded into a HIP graph are left alone).  aig_kernel_timing_read waits for the recorded launches of one kernel and returns
 * their summed duration and their number.  Two event records per launch: for benchmarks, not for
 * production runs. */
enum {
    AIG_KT_GAT_AGGREGATE = 0, /* aig_gat_fwd */
    AIG_KT_GAT_BWD_DST,       /* aig_gat_bwd: target-side pass */
    AIG_KT_GAT_BWD_SRC,       /* aig_gat_bwd: source-side pass */
    AIG_KT_POOL_WALK,         /* aig_level_pool_fwd: the walk over the nodes */
    AIG_KT_LINEAR128,         /* aig_linear128 */
    AIG_KT_WGRAD128,          /* aig_wgrad with 128 input columns */
    AIG_KT_GAT1_FWD,          /* aig_gat1_fwd */
    AIG_KT_GAT1_BWD,          /* aig_gat1_bwd: its one pass over the nodes */
    AIG_KT_COUNT
};
int aig_kernel_timing(unsigned kernel_mask);
int aig_kernel_timing_read(int kernel, double* total_ms, int64_t* launches);
const char* aig_kernel_timing_name(int kernel);

/* ---- sets of graphs -----------------------------------------------------------------------------------
 * The reference repairs, scores and resyncs the graphs of a minibatch (train.py:242-245,287) and the samples
 * of a bulk run (AIGgan.py:104-110) one after another.  These entry points take B graphs of n nodes each in
 * one call, a handful of launches for the whole set: every graph gets exactly the result the single-graph
 * entry point above gives it.
 *   edge slots: graph g's sources at edge_index[(2g) ecap + e], targets at edge_index[(2g+1) ecap + e],
 *   inversion values at edge_attr[g ecap + e], e < ecount[g] (ecount NULL: all ecap slots hold edges);
 *   node arrays (x [B n, 2], node_depth / inv_cnt / depth [B n]) are the graphs' arrays back to back. */

/* Node projections (aig_gen_node_proj) of B graphs in one call: x holds 2n floats per graph, or one
 * template's when x_shared (then only z differs between the graphs); z [B][128]; bf16 != 0 takes the bf16
 * matrix-core pipeline of aig_gen_node_proj_bf16, whose planes are IEEE halfs (f32 otherwise).  proj [4][B rows][128]
 * with rows = aig_gen_batch_rows(n) rows per graph (n rounded up to the 64-row tile; rows past n are unspecified). */
int64_t aig_gen_batch_rows(int64_t n);
size_t aig_gen_node_proj_batch_ws_bytes(int64_t B, int64_t n, int bf16);
int aig_gen_node_proj_batch(int64_t B, int64_t n, const float* x, int x_shared, const float* z,
                            const aig_gen_weights* w, int bf16, void* proj, void* ws, size_t ws_bytes, void* stream);

/* Scores, tie-exact selection and edge assembly (aig_gen_score, aig_topk_select, aig_gen_emit) of B graphs:
 * three launches for the set.  Plan arrays are [B][...] back to back, or one plan when plan_shared.  With windows of
 * up to 256 candidates the wave that scores a slot also makes its picks when torch.topk would take its
 * partial_sort path (k*64 <= cnt), same picks as aig_topk_select's. */
typedef struct aig_gen_wire_set {
    int64_t B, n, K;
    int64_t ecap;             /* edge slots per graph in the outputs (>= every graph's eoff[n]) */
    int64_t plan_shared;      /* 1: order .. cand describe one template used by all graphs */
    int64_t x_shared;         /* 1: x is one template's [n][2] */
    const float* x;           /* [B n][2] or [n][2] */
    const int32_t *order, *rank, *want, *cnt, *fb;   /* [B][n] or [n] */
    const int32_t* eoff;      /* [B][n+1] or [n+1] */
    const int32_t* cand;      /* [B][n][K] or [n][K] */
    const void* proj;         /* [4][B proj_rows][128] from aig_gen_node_proj_batch: f32, or IEEE halfs (proj_half) */
    int64_t proj_rows;
    const float *sc2_w, *sc2_b, *at2_w, *at2_b;      /* edge_scorer.2 / edge_attr_pred.2 */
    float* logits;            /* scratch [B][n][K]; may be NULL with scores by class (cls) and K <= 127: the selection
                               * then forms the scores itself and they are never stored */
    int32_t* sel;             /* scratch [B][n][2] */
    int64_t* edge_index;      /* out [B][2][ecap] */
    int64_t* edge_attr;       /* out [B][ecap] */
    int32_t* ecount;          /* out [B] */
    float* x_out;             /* out [B n][2] */
    int64_t proj_half;        /* 1: proj holds halfs (the bf16 pipeline's planes), scored as aig_gen_score_half does */
    /* Classes of equal nodes (NULL / 0: none).  A node enters the generator through its feature row alone
     * (generator.py:115-117), so nodes with equal rows have equal projections: with cls given, the planes of proj hold
     * one row per CLASS (rows 0 .. classes-1 of a graph's proj_rows) instead of one per node, cls[node] names a
     * node's class ([B][n], or [n] with plan_shared), and a pair's logit and inversion bit are read from per-graph
     * tables over (source class, target class) that the call fills first, with the per-pair chains of
     * aig_gen_score / aig_gen_emit - the same bits as without classes. */
    const int32_t* cls;
    int64_t classes;
    float* pair_logit;        /* scratch [B][classes][classes] */
    int32_t* pair_bit;        /* scratch [B][classes][classes] */
} aig_gen_wire_set;
int aig_gen_wire_batch(const aig_gen_wire_set* a, void* stream);

/* aig_enforce on B graphs.  depth holds depth_stride entries per graph: n, or 0 when all graphs share one
 * depth vector (samples of one template).  Outputs in the same slot layout with `cap` (>= 2n) slots per
 * graph and out_count[B]. */
size_t aig_enforce_batch_ws_bytes(int64_t B, int64_t n, int64_t ecap);
int aig_enforce_batch(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const float* x,
                      const int64_t* edge_index, const int64_t* edge_attr, int depth_kind, const void* depth,
                      int64_t depth_stride, int64_t* out_edge_index, int64_t cap, int64_t* out_edge_attr,
                      int32_t* out_count, int64_t* node_depth, float* x_out, void* ws, size_t ws_bytes, void* stream);

/* The node_depth output of aig_enforce_batch alone (breadth-first levels from the PIs over the cleaned edges of
 * the INPUT graphs, constraints.py:88-112,640-642), for callers that pass node_depth = NULL to the repair, use the
 * repaired edges at once and need the levels later.  Workspace: aig_enforce_batch_ws_bytes(B, n, ecap). */
int aig_clean_depth_batch(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const float* x,
                          const int64_t* edge_index, int64_t* node_depth, void* ws, size_t ws_bytes, void* stream);

/* aig_penalties on B graphs: out9 [B][9]. */
size_t aig_penalties_batch_ws_bytes(int64_t B, int64_t n, int64_t ecap);
int aig_penalties_batch(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const float* x,
                        const int64_t* edge_index, float* out9, void* ws, size_t ws_bytes, void* stream);

/* The same with a hint that makes the reachability term cheap (same results, whatever the hint holds).
 * reach_violation (constraints.py:310-339: nodes not both reached forward from a PI and backward from a PO) is a
 * walk level by level in general; when an order of the nodes is known in which every edge runs from an earlier to
 * a later node, it is one pass over the nodes in that order and one in reverse.
 *   order_hint int32 [B][n] (hint_stride = n) or [n] shared by all graphs (hint_stride = 0), may be NULL: the place
 *       of every node in such an order - e.g. the generator's processing rank (aig_gen_order), in which every
 *       emitted edge runs upward (generator.py:131-189).
 * The hint is checked on the device, per graph: it must be a permutation of 0..n-1, else the graph's PI / AND / PO
 * position is tried instead (what a repaired graph obeys; also the order tried by aig_penalties /
 * aig_penalties_batch); a graph with an edge (self loops aside) that does not run upward in the order tried is
 * walked level by level as before.  Workspace: aig_penalties_batch_ws_bytes. */
int aig_penalties_batch_hinted(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const float* x,
                               const int64_t* edge_index, const int32_t* order_hint, int64_t hint_stride, float* out9,
                               void* ws, size_t ws_bytes, void* stream);

/* aig_resync on B graphs (a graph with a cycle gets zeros, the others their levels). */
size_t aig_resync_batch_ws_bytes(int64_t B, int64_t n, int64_t ecap);
int aig_resync_batch(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const int64_t* edge_index,
                     const int64_t* edge_attr, int64_t* inv_cnt, int64_t* depth, void* ws, size_t ws_bytes,
                     void* stream);
/* aig_resync_typed on B graphs (x [B n, 2]); the sweep / walk choice is made per graph, on the device. */
size_t aig_resync_typed_batch_ws_bytes(int64_t B, int64_t n, int64_t ecap);
int aig_resync_typed_batch(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const float* x,
                           const int64_t* edge_index, const int64_t* edge_attr, int64_t* inv_cnt, int64_t* depth,
                           void* ws, size_t ws_bytes, void* stream);

/* Edge slots of B graphs -> the edge tensors of the padded minibatch the critic takes: pad_data
 * (utilities/data_process.py:30-68) on every graph, then PyG's concatenation with node offsets
 * (train.py:243-245).  out_edge_index [2][B max_edges]: graph g owns columns [g max_edges, (g+1) max_edges),
 * its first min(ecount[g], max_edges) edges with g n added to both endpoints, then -1 + g n (the loader
 * shifts the pad value like any other entry); out_edge_attr [B max_edges] with 0 on pad columns. */
int aig_pad_batch(int64_t B, int64_t n, int64_t ecap, const int32_t* ecount, const int64_t* edge_index,
                  const int64_t* edge_attr, int64_t max_edges, int64_t* out_edge_index, int64_t* out_edge_attr,
                  void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AIG_ABI_H */
