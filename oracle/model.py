"""TEST INFRASTRUCTURE ONLY - fp32 torch-CPU restatement of the floating-point half of the path.

Imported by tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``; never by anything under ``gan_amd/``.

* ``generator_forward`` follows models/generator.py:101-217 target by target with the
  same dense ops on the same shapes, the C checker for order / candidate windows and the
  libstdc++ routines ATen's top-k uses.  Parity status: PINNED (tests/golden/gen_*.npz).
* ``gat_conv`` / ``discriminator_forward`` follow models/discriminator.py:42-100.  The
  attention layer itself is ``torch_geometric.nn.GATConv``, an unpinned third-party
  dependency that is absent from the reference tree and from this image; its published
  forward (PyG 2.3-2.6, ``edge_dim=1``, ``concat=True``, ``add_self_loops=True``,
  ``fill_value='mean'``, ``negative_slope=0.2``) is restated here.  Parity status:
  PARITY UNPINNED - the reference holds no test, fixture or output for this layer.
* ``gradient_penalty`` follows train.py:133-152 (the only live branch, use_edges=False), ``gradient_penalty_edges``
  the branch the loop never takes (train.py:154-185).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import cpu

PI, PO, AND = 0, 1, 2


# ------------------------------------------------------------------------------------ generator
def _mlp2(x, w0, b0, w1, b1, last_relu=True):
    h = F.linear(torch.relu(F.linear(x, w0, b0)), w1, b1)
    return torch.relu(h) if last_relu else h


def generator_forward(state: Dict[str, torch.Tensor], x: torch.Tensor, node_depth: Optional[torch.Tensor],
                      z: torch.Tensor, candidate_k: int, keep: bool = False):
    """Returns dict(edge_index, edge_attr, x) (+ per-target records when ``keep``).

    Runs the dense ops on ONE thread: the reference's pick among tied candidates is only
    defined when equal input rows give bit-equal logits, and a multi-threaded CPU GEMM
    rounds the remainder rows of each thread's slice differently (observed: 8 threads
    split 30 identical rows into 3 logit values).  The golden vectors were captured the
    same way."""
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        return _generator_forward(state, x, node_depth, z, candidate_k, keep)
    finally:
        torch.set_num_threads(nthreads)


def _generator_forward(state, x, node_depth, z, candidate_k, keep):
    s = {k: v.detach().cpu().float() for k, v in state.items()}
    x = x.detach().cpu()
    z = z.detach().cpu().float().view(-1)
    n = x.size(0)
    x_in = x.float()
    types = x[:, 0].long().numpy()
    h = _mlp2(x_in, s["node_enc.0.weight"], s["node_enc.0.bias"], s["node_enc.2.weight"], s["node_enc.2.bias"])
    h_cond = _mlp2(torch.cat([h, z.unsqueeze(0).expand(n, -1)], 1), s["node_proj.0.weight"], s["node_proj.0.bias"],
                   s["node_proj.2.weight"], s["node_proj.2.bias"])
    depth = None
    if node_depth is not None and node_depth.numel() == n:
        depth = node_depth.detach().cpu().to(torch.float64).numpy()
    order, _ = cpu.build_order(types, depth)
    tgt, cand, cnt, fb = cpu.candidates(types, order, candidate_k)
    src_out, dst_out, inv_out = [], [], []
    rec = {"tgt": [], "cand": [], "logit": []}
    for t in range(tgt.shape[0]):
        v = int(tgt[t])
        want_k = 2 if types[v] == AND else 1
        c = int(cnt[t])
        if c == 0:
            if fb[t] >= 0:
                src_out += [int(fb[t])] * want_k
                dst_out += [v] * want_k
                inv_out += [0] * want_k
            continue
        ci = torch.from_numpy(cand[t, :c].copy())
        edge_in = torch.cat([h_cond[ci], h_cond[v:v + 1].expand(c, -1), z.unsqueeze(0).expand(c, -1)], 1)
        logits = _mlp2(edge_in, s["edge_scorer.0.weight"], s["edge_scorer.0.bias"], s["edge_scorer.2.weight"],
                       s["edge_scorer.2.bias"], last_relu=False).view(-1)
        k_eff = min(want_k, c)
        topi = cpu.topk(logits.numpy(), k_eff)
        inv_logit = _mlp2(edge_in[torch.from_numpy(topi.copy())], s["edge_attr_pred.0.weight"],
                          s["edge_attr_pred.0.bias"], s["edge_attr_pred.2.weight"], s["edge_attr_pred.2.bias"],
                          last_relu=False).view(-1)
        inv_bits = (torch.sigmoid(inv_logit) > 0.5).long().tolist()
        srcs = [int(cand[t, i]) for i in topi]
        while len(srcs) < want_k:  # repeat the first pick (generator.py:178-182)
            srcs.append(srcs[0])
            inv_bits.append(inv_bits[0])
        src_out += srcs
        dst_out += [v] * want_k
        inv_out += inv_bits
        if keep:
            rec["tgt"].append(v)
            rec["cand"].append(cand[t, :c].copy())
            rec["logit"].append(logits.numpy().copy())
    ei = torch.tensor([src_out, dst_out], dtype=torch.long).view(2, -1)
    ea = torch.tensor(inv_out, dtype=torch.long).view(-1, 1)
    x_out = x.clone()
    inv_count = torch.zeros(n, dtype=torch.long)
    if ei.numel() > 0:
        inv_count.scatter_add_(0, ei[1], (ea.view(-1) == 1).long())
    x_out[:, 1] = inv_count
    out = {"edge_index": ei, "edge_attr": ea, "x": x_out, "h_cond": h_cond, "order": order,
           "tgt": tgt, "cand": cand, "cnt": cnt, "fb": fb}
    if keep:
        out["rec"] = rec
    return out


def generator_forward_canonical(state: Dict[str, torch.Tensor], x: torch.Tensor,
                                node_depth: Optional[torch.Tensor], z: torch.Tensor, candidate_k: int):
    """models/generator.py:101-217 with the dense layers evaluated as fixed-order fmaf chains
    (aig_oracle.c, "position-independent form") instead of through the CPU BLAS.  Same order,
    windows, top-k routine, padding and assembly as ``generator_forward``; logits agree with it
    to ~1e-6 relative, and equal candidate rows always tie exactly.  This is the form the HIP
    kernels are compared with bit for bit."""
    x = x.detach().cpu()
    n = x.size(0)
    types = x[:, 0].long().numpy()
    depth = None
    if node_depth is not None and node_depth.numel() == n:
        depth = node_depth.detach().cpu().to(torch.float64).numpy()
    order, _ = cpu.build_order(types, depth)
    tgt, cand, cnt, fb = cpu.candidates(types, order, candidate_k)
    h_cond, proj = cpu.node_proj(state, x.float().numpy(), z.detach().cpu().float().numpy())
    T = tgt.shape[0]
    logits = cpu.score_rows(tgt, cnt, np.where(cand < 0, 0, cand), proj[0], proj[1],
                            state["edge_scorer.2.weight"], state["edge_scorer.2.bias"])
    want = np.where(types[tgt] == AND, 2, 1).astype(np.int64)
    sel = cpu.topk_rows(np.nan_to_num(logits, nan=0.0) if T else logits, cnt, want)
    k_eff = np.minimum(want, cnt)
    ne = np.where((cnt > 0) | (fb >= 0), want, 0)
    eoff = np.concatenate([[0], np.cumsum(ne)])
    E = int(eoff[-1])
    src, dst, use_fb = np.zeros(E, np.int64), np.zeros(E, np.int64), np.zeros(E, bool)
    rows = np.arange(T)
    for j in (0, 1):
        m = ne > j
        pick = np.where(k_eff > j, sel[:, j], sel[:, 0])           # missing second pick repeats the first
        s = np.where(cnt > 0, cand[rows, np.maximum(pick, 0)], fb)
        src[eoff[:-1][m] + j] = s[m]
        dst[eoff[:-1][m] + j] = tgt[m]
        use_fb[eoff[:-1][m] + j] = (cnt == 0)[m]
    bits = cpu.inv_bits(src, dst, proj[2], proj[3], state["edge_attr_pred.2.weight"], state["edge_attr_pred.2.bias"])
    bits[use_fb] = 0
    x_out = x.clone()
    inv_count = np.zeros(n, np.int64)
    np.add.at(inv_count, dst, bits)
    x_out[:, 1] = torch.from_numpy(inv_count).to(x_out.dtype)
    return {"edge_index": torch.from_numpy(np.stack([src, dst])), "edge_attr": torch.from_numpy(bits).view(-1, 1),
            "x": x_out, "h_cond": h_cond, "proj": proj, "order": order, "tgt": tgt, "cand": cand, "cnt": cnt,
            "fb": fb, "logits": logits, "sel": sel}


def node_proj_bf16(state: Dict[str, torch.Tensor], x: torch.Tensor, z: torch.Tensor):
    """The node pipeline of the bf16 scoring option (an extension of this repository, no counterpart in the
    reference): activations and weights entering each of the seven 128x128 products rounded to bf16 (nearest
    even), products summed exactly (float64 here; the matrix cores accumulate in f32), biases and ReLU in f32;
    h_cond stays f32, the four planes are clamped to +-32752 and rounded to IEEE half (returned as the f32 values
    of those halfs).  -> (h_cond [n,128], proj [4,n,128]).  Checked to a tolerance, not bit for bit: the order in
    which the hardware adds the sixteen products of an instruction is not documented."""
    def bf(t):
        return t.to(torch.bfloat16).to(torch.float64)

    def lin(h, w, b):
        return (bf(h) @ bf(w).t()).to(torch.float32) + b.float()

    x, z = x.detach().cpu().float(), z.detach().cpu().float().reshape(-1)
    st = {k: v.detach().cpu().float() for k, v in state.items()}
    h = torch.relu(x @ st["node_enc.0.weight"].t() + st["node_enc.0.bias"])
    h = torch.relu(lin(h, st["node_enc.2.weight"], st["node_enc.2.bias"]))
    w0 = st["node_proj.0.weight"]
    zb0 = (w0[:, 128:].double() @ z.double()).float() + st["node_proj.0.bias"]
    h = torch.relu(lin(h, w0[:, :128], zb0))
    h_cond = torch.relu(lin(h, st["node_proj.2.weight"], st["node_proj.2.bias"]))
    outs = []
    for name in ("edge_scorer.0", "edge_attr_pred.0"):
        w, b = st[name + ".weight"], st[name + ".bias"]
        zb = (w[:, 256:].double() @ z.double()).float() + b
        outs.append(lin(h_cond, w[:, :128], torch.zeros(128)))
        outs.append(lin(h_cond, w[:, 128:256], zb))
    return h_cond, torch.stack(outs).clamp(-32752.0, 32752.0).to(torch.float16).to(torch.float32)


def score_half(proj_a_rows: torch.Tensor, proj_b_row: torch.Tensor, w2: torch.Tensor, b2: float) -> torch.Tensor:
    """Logits of the packed-half scoring that follows the bf16 pipeline (same extension): ``proj_a_rows`` [c,128]
    candidate rows and ``proj_b_row`` [128] the target's row, both holding half values; every sum a+b is rounded
    to half, ReLU, w2 rounded to half, products exact, accumulated in f32 two channels at a time in two chains
    (groups of eight channels alternate between them), chains added, then the bias.  The hardware instruction's
    own rounding of "two products plus accumulator" is not documented: one rounding per instruction is assumed
    here and the comparison is made at a tolerance."""
    import numpy as np
    a = proj_a_rows.detach().cpu().to(torch.float16).double().numpy()
    b = proj_b_row.detach().cpu().to(torch.float16).double().numpy()
    w = w2.detach().cpu().reshape(-1).to(torch.float16).double().numpy()
    t = np.maximum((a + b[None, :]).astype(np.float16).astype(np.float64), 0.0)       # exact sum, one rounding
    prod = t * w[None, :]
    acc = [np.zeros(a.shape[0], dtype=np.float32), np.zeros(a.shape[0], dtype=np.float32)]
    for g in range(16):                       # group of eight channels -> chain g & 1, four instructions
        for i in range(4):
            j = 8 * g + 2 * i
            acc[g & 1] = (acc[g & 1].astype(np.float64) + prod[:, j] + prod[:, j + 1]).astype(np.float32)
    return torch.from_numpy(((acc[0] + acc[1]).astype(np.float32) + np.float32(b2)).astype(np.float32))


# ------------------------------------------------------------------------------------ critic
def gat_self_loops(edge_index: torch.Tensor, edge_attr: torch.Tensor, n: int):
    """Drop self loops, then append one per node whose attribute is the mean attribute of
    the edges entering that node (0 when none)."""
    keep = edge_index[0] != edge_index[1]
    ei, ea = edge_index[:, keep], edge_attr[keep]
    ssum = torch.zeros((n, ea.size(1)), dtype=ea.dtype).index_add_(0, ei[1], ea)
    deg = torch.zeros(n, dtype=ea.dtype).index_add_(0, ei[1], torch.ones(ei.size(1), dtype=ea.dtype))
    loop_attr = ssum / deg.clamp(min=1).unsqueeze(1)
    loops = torch.arange(n, dtype=ei.dtype)
    return torch.cat([ei, torch.stack([loops, loops])], 1), torch.cat([ea, loop_attr], 0)


def gat_conv(p: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, edge_index: torch.Tensor,
             edge_attr: torch.Tensor, heads: int = 4) -> torch.Tensor:
    """GATConv(in, C, heads, edge_dim=1, concat=True): x [N,F] -> [N, heads*C]."""
    n = x.size(0)
    w = p[prefix + "lin.weight"]                      # [heads*C, F]
    c = w.size(0) // heads
    ei, ea = gat_self_loops(edge_index, edge_attr, n)
    xw = F.linear(x, w).view(n, heads, c)
    a_s = (xw * p[prefix + "att_src"]).sum(-1)        # [N, heads]
    a_d = (xw * p[prefix + "att_dst"]).sum(-1)
    ew = F.linear(ea, p[prefix + "lin_edge.weight"]).view(-1, heads, c)
    a_e = (ew * p[prefix + "att_edge"]).sum(-1)       # [E', heads]
    j, i = ei[0], ei[1]
    e = F.leaky_relu(a_s[j] + a_d[i] + a_e, 0.2)
    emax = torch.full((n, heads), float("-inf"), dtype=e.dtype).scatter_reduce(
        0, i.unsqueeze(1).expand(-1, heads), e.detach(), reduce="amax", include_self=True)
    ex = torch.exp(e - emax[i])
    den = torch.zeros((n, heads), dtype=e.dtype).index_add_(0, i, ex)
    alpha = ex / (den[i] + 1e-16)
    out = torch.zeros((n, heads, c), dtype=x.dtype).index_add_(0, i, alpha.unsqueeze(-1) * xw[j])
    return out.reshape(n, heads * c) + p[prefix + "bias"]


def level_readout(h: torch.Tensor, node_depth: Optional[torch.Tensor]) -> torch.Tensor:
    """models/discriminator.py:64-96 -> [1, 2H]."""
    def glob(hh):
        return torch.cat([hh.mean(0, keepdim=True), hh.max(0)[0].unsqueeze(0)], 1)
    if node_depth is None:
        return glob(h)
    nd = node_depth
    valid = nd >= 0
    if int(valid.sum()) == 0:
        return glob(h)
    nd = nd.clone()
    nd[~valid] = -1
    vecs = []
    for l in range(int(nd.max().item()) + 1):
        m = nd == l
        if int(m.sum()) == 0:
            vecs.append(torch.zeros(2 * h.size(1), dtype=h.dtype))
            continue
        sub = h[m]
        vecs.append(torch.cat([sub.mean(0), sub.max(0)[0]], 0))
    return torch.stack(vecs, 0).mean(0, keepdim=True)


def level_readout_segmented(h: torch.Tensor, node_depth: Optional[torch.Tensor]) -> torch.Tensor:
    """The same readout as ``level_readout`` (models/discriminator.py:64-96) without its loop over levels, for
    graphs of 10^5 nodes and hundreds of levels where the loop's backward pass fills an [N, H] array per level.
    Nodes are ordered by (level, node id); a level's mean is its segment's sum over its count; a level's maximum
    is read from ONE row per column - the first node of the level, in node order, that holds the maximum, the row
    ``sub.max(0)`` differentiates through - so first and second derivatives follow the loop's.  The sums are
    added in another order than ``mean(0)`` adds them: equal to the loop within rounding, not bit for bit
    (tests/test_oracle_golden.py holds it to the loop, values and gradients)."""
    if node_depth is None:
        return level_readout(h, None)
    nd = node_depth.long()
    valid = nd >= 0
    if int(valid.sum()) == 0:
        return level_readout(h, node_depth)
    ids = torch.nonzero(valid).reshape(-1)
    lv = nd[ids]
    order = torch.argsort(lv, stable=True)          # ids ascend, so a level's nodes stay in node order
    ids, lv = ids[order], lv[order]
    levels = int(lv.max().item()) + 1
    rows = h[ids]                                   # [M, H]
    hdim = h.size(1)
    cnt = torch.zeros(levels, dtype=h.dtype).index_add_(0, lv, torch.ones(lv.numel(), dtype=h.dtype))
    mean = torch.zeros((levels, hdim), dtype=h.dtype).index_add(0, lv, rows) / cnt.clamp(min=1).unsqueeze(1)
    idx = lv.unsqueeze(1).expand(-1, hdim)
    top = torch.full((levels, hdim), float("-inf"), dtype=h.dtype).scatter_reduce(
        0, idx, rows.detach(), reduce="amax", include_self=True)
    pos = torch.arange(rows.size(0)).unsqueeze(1).expand(-1, hdim)
    first = torch.where(rows.detach() == top[lv], pos, torch.full_like(pos, rows.size(0)))
    arg = torch.full((levels, hdim), rows.size(0), dtype=torch.long).scatter_reduce(0, idx, first, reduce="amin", include_self=True)
    filled = cnt > 0
    arg = torch.where(filled.unsqueeze(1), arg, torch.zeros_like(arg))
    mx = rows.gather(0, arg) * filled.unsqueeze(1).to(h.dtype)          # an empty level is a row of zeros
    return torch.cat([mean, mx], 1).mean(0, keepdim=True)


def discriminator_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, edge_index: torch.Tensor,
                          edge_attr: Optional[torch.Tensor], node_depth: Optional[torch.Tensor],
                          heads: int = 4, readout=None) -> torch.Tensor:
    """models/discriminator.py:42-100 -> tensor [1].  ``readout``: ``level_readout`` (the loop of
    discriminator.py:78-87, the default) or ``level_readout_segmented``."""
    x = x.float()
    if edge_attr is None:
        edge_attr = torch.zeros((edge_index.size(1), 1), dtype=x.dtype)
    edge_attr = edge_attr.float()
    h = F.elu(gat_conv(p, "gat1.", x, edge_index, edge_attr, heads))
    h = F.elu(gat_conv(p, "gat2.", h, edge_index, edge_attr, heads))
    g = (readout or level_readout)(h, node_depth)
    g = torch.relu(F.linear(g, p["mlp.0.weight"], p["mlp.0.bias"]))
    g = torch.relu(F.linear(g, p["mlp.2.weight"], p["mlp.2.bias"]))
    return F.linear(g, p["out.weight"], p["out.bias"]).view(-1)


def gradient_penalty(p: Dict[str, torch.Tensor], real_x: torch.Tensor, fake_x: torch.Tensor,
                     node_depth: torch.Tensor, alpha: torch.Tensor, readout=None):
    """train.py:141-152 with a given alpha [1,1]; -> (gp, grad_norm)."""
    x_inter = (alpha * real_x + (1 - alpha) * fake_x).requires_grad_(True)
    ei = torch.zeros((2, 0), dtype=torch.long)
    ea = torch.zeros((0, 1), dtype=torch.float32)
    out = discriminator_forward(p, x_inter, ei, ea, node_depth, readout=readout)
    grads = torch.autograd.grad(out.sum(), [x_inter], create_graph=True, retain_graph=True, only_inputs=True)
    gn = torch.norm(grads[0].reshape(-1), p=2)
    return (gn - 1.0) ** 2, gn.detach()


def gradient_penalty_edges(p: Dict[str, torch.Tensor], real, fake, alpha: torch.Tensor, readout=None):
    """train.py:154-185 (use_edges=True) with a given alpha [1,1] on two padded graphs (objects with x, edge_index,
    edge_attr, node_depth); -> (gp, grad_norm).  Without a slot that holds an edge on both sides: train.py:156-166,
    the node-only form."""
    m_r = real.edge_index.min(dim=0).values >= 0 if real.edge_index.numel() else torch.zeros(0, dtype=torch.bool)
    m_f = fake.edge_index.min(dim=0).values >= 0 if fake.edge_index.numel() else torch.zeros(0, dtype=torch.bool)
    if m_r.numel() == 0 or m_f.numel() == 0 or int((m_r & m_f).sum()) == 0:
        return gradient_penalty(p, real.x, fake.x, real.node_depth, alpha, readout=readout)
    both = m_r & m_f
    x_inter = (alpha * real.x + (1 - alpha) * fake.x).requires_grad_(True)
    ea_inter = (alpha * real.edge_attr.float()[both] + (1 - alpha) * fake.edge_attr.float()[both]).requires_grad_(True)
    out = discriminator_forward(p, x_inter, real.edge_index[:, both], ea_inter, real.node_depth, readout=readout)
    g_x, g_ea = torch.autograd.grad(out.sum(), [x_inter, ea_inter], create_graph=True, retain_graph=True, only_inputs=True)
    gn = torch.norm(torch.cat([g_x.reshape(-1), g_ea.reshape(-1)], dim=0), p=2)
    return (gn - 1.0) ** 2, gn.detach()


def sanitize_edges(edge_index: torch.Tensor, edge_attr: torch.Tensor):
    """train.py:108-123: drop every column holding a negative index."""
    if edge_index.numel() == 0:
        return edge_index, edge_attr
    m = edge_index.min(dim=0).values >= 0
    return edge_index[:, m], edge_attr[m]


def init_discriminator_params(seed: int = 0, in_dim: int = 2, hidden: int = 128, heads: int = 4):
    """Random critic parameters with GATConv's names and shapes (glorot / zeros)."""
    g = torch.Generator().manual_seed(seed)
    c = hidden // heads

    def glorot(*shape):
        fan = shape[-1] + shape[-2]
        a = (6.0 / fan) ** 0.5
        return (torch.rand(*shape, generator=g) * 2 - 1) * a

    p = {}
    for name, fin in (("gat1.", in_dim), ("gat2.", hidden)):
        p[name + "lin.weight"] = glorot(hidden, fin)
        p[name + "att_src"] = glorot(1, heads, c)
        p[name + "att_dst"] = glorot(1, heads, c)
        p[name + "att_edge"] = glorot(1, heads, c)
        p[name + "lin_edge.weight"] = glorot(hidden, 1)
        p[name + "bias"] = torch.rand(hidden, generator=g) * 0.1
    p["mlp.0.weight"], p["mlp.0.bias"] = glorot(hidden, 2 * hidden), torch.rand(hidden, generator=g) * 0.1
    p["mlp.2.weight"], p["mlp.2.bias"] = glorot(hidden // 2, hidden), torch.rand(hidden // 2, generator=g) * 0.1
    p["out.weight"], p["out.bias"] = glorot(1, hidden // 2), torch.rand(1, generator=g) * 0.1
    return p
