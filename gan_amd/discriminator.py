"""Edge-aware GAT critic on MI355X kernels.

Drop-in for ``models.discriminator.AIGDiscriminator`` of the reference
(models/discriminator.py:13-100): same constructor, same parameter names as a PyG
``GATConv(edge_dim=1)`` state dict (``gat1.lin.weight``, ``gat1.att_src`` ...; the older
``lin_src`` / ``lin_dst`` spelling is accepted on load), same ``forward(data) -> Tensor[1]``
(ONE score per call, also for a ``Batch`` of several graphs: all of them are pooled by level
together, discriminator.py:89-100).

Where the work runs (every item is a kernel of libaig_hip.so; torch supplies memory, streams, autograd
bookkeeping and the optimiser):
* dense projections ``x W^T`` (``aig_linear_in2`` for the two input features, ``aig_linear128`` on the f32
  matrix cores for the second layer), their input gradients and weight gradients (``aig_wgrad``);
* attention logits, per-target softmax, neighbour aggregation, bias and ELU: one fused kernel over a
  by-target CSR (``aig_gat_fwd``), backward in ``aig_gat_bwd``;
* level-wise mean/max readout: one pass (``aig_level_pool_fwd``); its backward (``aig_level_pool_bwd``) is
  linear in the incoming gradient and has its own adjoint kernel, so it can be differentiated again;
* the 256-128-64-1 head: one launch forward, one backward (``aig_head_fwd`` / ``aig_head_bwd``); torch modules
  only where the autograd form of the gradient penalty differentiates the head twice;
* a graph without edges (the gradient-penalty interpolate of train.py:141-152) needs no attention kernel:
  softmax over the lone self loop is exactly 1, so a layer is ``elu(x W^T + bias)``; the penalty and its
  parameter gradient have a closed form (``gradient_penalty_closed_form``).

Graph structure (by-target CSR, level plan) is either derived from the tensors of a call and cached, or - for
inputs that live in fixed buffers and are refilled every step - held in ``GraphCSR`` / ``DynLevelPlan`` objects
the caller rebuilds in place (``data._csr`` / ``data._plan``), which is what lets a whole critic term be recorded
once per tensor SHAPE as a HIP graph and replayed on every minibatch of that shape.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd.function import once_differentiable

from . import _abi
from .data import Data

HIDDEN = 128
# Switches for the tests that compare the fused forms with the separate kernels they replace (both stay in the
# library): the first layer without projected rows (aig_gat1_*), the readout's backward inside the second layer's,
# the second layer's node terms out of its projection's launch (aig_linear128_node_terms).
FUSED_FIRST_LAYER = True
FUSED_READOUT_BACKWARD = True
FUSED_NODE_TERMS = True
# ... which pays from a few hundred thousand rows on (1M nodes: the step 2 % faster; 100k: 0.7 % slower, the longer
# epilogue shows while the projection is not yet bound by memory; small minibatches: 10 % slower)
NODE_TERMS_MIN_ROWS = 400_000


class GraphCSR:
    """By-target and by-source edge lists of one (batched) graph, built on the GPU.  ``E`` counts the columns
    handed in; columns with an endpoint outside ``[0, n)`` (the ``-1`` padding of train.py:108-123) and self
    loops are dropped by the builder, so a padded edge list needs no host-side filtering.  ``rebuild`` fills
    the same buffers from a new edge list of the same width."""

    def __init__(self, edge_index: torch.Tensor, edge_attr: Optional[torch.Tensor], n: int):
        dev = edge_index.device
        E = int(edge_index.size(1))
        self.n, self.E = int(n), E
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        self.row_ptr = torch.empty(n + 1, **i32)
        self.col_ptr = torch.empty(n + 1, **i32)
        self.csr_src = torch.empty(max(E, 1), **i32)
        self.csr_eid = torch.empty(max(E, 1), **i32)
        self.csc_pos = torch.empty(max(E, 1), **i32)
        self.csc_dst = torch.empty(max(E, 1), **i32)
        self.hub_tab = torch.empty((E // 33 + 2, 4), **i32)   # nodes with more than 32 out-edges, see aig_abi.h
        self.csr_ea = torch.empty(max(E, 1), **f32)
        self.loop_ea = torch.empty(max(n, 1), **f32)
        self.rebuild(edge_index, edge_attr)

    def rebuild(self, edge_index: torch.Tensor, edge_attr: Optional[torch.Tensor]) -> None:
        lib = _abi.lib()
        n, E, dev = self.n, self.E, edge_index.device
        if int(edge_index.size(1)) != E:
            raise _abi.AigAbiError(f"GraphCSR.rebuild: {int(edge_index.size(1))} edge columns, built for {E}")
        ei = edge_index.contiguous()
        ea = None if edge_attr is None else edge_attr.reshape(-1).float().contiguous()
        ws = _abi.workspace(lib.aig_gat_csr_ws_bytes(n, E), dev)
        _abi.check(lib.aig_gat_csr(n, E, _abi.ptr(ei) if E else None, _abi.ptr(ea) if E else None,
                                   _abi.ptr(self.row_ptr), _abi.ptr(self.csr_src), _abi.ptr(self.csr_ea),
                                   _abi.ptr(self.csr_eid), _abi.ptr(self.loop_ea), _abi.ptr(self.col_ptr),
                                   _abi.ptr(self.csc_pos), _abi.ptr(self.csc_dst), _abi.ptr(self.hub_tab),
                                   _abi.ptr(ws), ws.numel(),
                                   _abi.stream()), "aig_gat_csr")


class LevelPlan:
    """Nodes grouped by depth level (negative depth = ignored)."""

    def __init__(self, node_depth: Optional[torch.Tensor], n: int, device):
        lib = _abi.lib()
        L = 0
        depth = None
        nvalid = None
        if node_depth is not None and n > 0:
            nd = node_depth.to(device)
            # the one host read per depth vector (the plan is cached): deepest level and nodes with a level
            top, nvalid = (int(v) for v in torch.stack([nd.max().long(), (nd >= 0).sum()]).tolist())
            if top >= 0:                # otherwise no valid level: pool over all nodes (discriminator.py:68-72)
                L = top
                depth = nd.to(torch.int64).contiguous()
            else:
                nvalid = None
        self.n, self.L = int(n), int(L)
        i32 = dict(dtype=torch.int32, device=device)
        self.perm = torch.empty(max(n, 1), **i32)
        self.node_level = torch.empty(max(n, 1), **i32)
        self.lvl_ptr = torch.empty(L + 2, **i32)
        self.w = torch.empty(max(n, 1), dtype=torch.float32, device=device)
        ws = _abi.workspace(lib.aig_level_plan_ws_bytes(n, L), device)
        _abi.check(lib.aig_level_plan(n, L, _abi.ptr(depth), _abi.ptr(self.perm), _abi.ptr(self.lvl_ptr),
                                      _abi.ptr(self.node_level), _abi.ptr(self.w), _abi.ptr(ws), ws.numel(),
                                      _abi.stream()), "aig_level_plan")
        # lvl_ptr[L + 1] = nodes in levels 0..L; without a depth vector every node sits in level 0
        self.nvalid = (nvalid if nvalid is not None else n) if n > 0 else 0
        self.meta = None      # L and nvalid are host values (see DynLevelPlan for the other case)


class DynLevelPlan:
    """The same plan for a depth vector that lives in a fixed buffer and changes between uses: the deepest
    level is found on the device and kept in a small header (``meta``) that the readout kernels read, so
    neither building nor using the plan needs a host read, and the launches have the same shapes for every
    depth vector with ``n`` entries and levels up to ``level_cap``.  ``L`` / ``nvalid`` hold the capacities."""

    def __init__(self, n: int, level_cap: int, device):
        self.n, self.L, self.nvalid = int(n), int(level_cap), int(n)
        i32 = dict(dtype=torch.int32, device=device)
        self.perm = torch.empty(max(n, 1), **i32)
        self.node_level = torch.empty(max(n, 1), **i32)
        self.lvl_ptr = torch.empty(self.L + 2, **i32)
        self.w = torch.empty(max(n, 1), dtype=torch.float32, device=device)
        self.meta = torch.zeros(4, **i32)

    def rebuild(self, node_depth: torch.Tensor) -> None:
        """Enqueue the plan of ``node_depth`` (int64 ``[n]``, negative = ignored node) on the current stream."""
        lib = _abi.lib()
        if node_depth.dtype != torch.int64 or int(node_depth.numel()) != self.n or not node_depth.is_contiguous():
            raise _abi.AigAbiError("DynLevelPlan.rebuild takes a contiguous int64 depth vector of the planned length")
        ws = _abi.workspace(lib.aig_level_plan_ws_bytes(self.n, self.L), node_depth.device)
        _abi.check(lib.aig_level_plan_dyn(self.n, self.L, _abi.ptr(node_depth), _abi.ptr(self.perm), _abi.ptr(self.lvl_ptr),
                                          _abi.ptr(self.node_level), _abi.ptr(self.w), _abi.ptr(self.meta),
                                          _abi.ptr(ws), ws.numel(), _abi.stream()), "aig_level_plan_dyn")

    def overflowed(self) -> bool:
        """True when a depth of the last rebuild exceeded ``level_cap`` (one host read; a debugging aid)."""
        return bool(int(self.meta[2].item()) & 2)


class GpRows:
    """The distinct rows of the closed-form gradient penalty (``aig_gp_rows``): nodes of one level with the same
    (real, generated) features are one row with a multiplicity, kept in buffers of fixed address so that the
    penalty can be recorded once and replayed over whatever ``rebuild`` last left in them.  ``rebuild`` sorts and
    scans (library routines), so it runs eagerly, before the replay - once per (real, generated) pair, the random
    ``alpha`` of a critic iteration does not enter."""

    def __init__(self, n: int, level_cap: int, device):
        self.n, self.L = int(n), int(level_cap)
        f32 = dict(dtype=torch.float32, device=device)
        self.x_real, self.x_fake = torch.zeros((n, 2), **f32), torch.zeros((n, 2), **f32)
        self.depth = torch.full((n,), -1, dtype=torch.int64, device=device)
        self.w_fwd, self.w_bwd = torch.zeros(n, **f32), torch.zeros(n, **f32)
        self.n_rows = torch.zeros(1, dtype=torch.int32, device=device)
        self.plan = DynLevelPlan(n, self.L, device)

    def rebuild(self, x_real: torch.Tensor, x_fake: torch.Tensor, plan) -> None:
        """Rows of the penalty between ``x_real`` and ``x_fake`` ([n,2] f32) whose nodes have the levels of ``plan``
        (a ``LevelPlan`` / ``DynLevelPlan`` of the same n with at most this object's level capacity)."""
        lib = _abi.lib()
        if int(x_real.size(0)) != self.n or int(x_fake.size(0)) != self.n or plan.n != self.n or plan.L > self.L:
            raise _abi.AigAbiError("GpRows.rebuild: sizes differ from the planned ones")
        xr, xf = x_real.float().contiguous(), x_fake.float().contiguous()
        ws = _abi.workspace(lib.aig_gp_rows_ws_bytes(self.n, self.L), xr.device)
        _abi.check(lib.aig_gp_rows(self.n, plan.nvalid, self.L, _abi.ptr(plan.meta), _abi.ptr(plan.perm), _abi.ptr(xr), _abi.ptr(xf),
                                   _abi.ptr(plan.node_level), _abi.ptr(plan.w),
                                   _abi.ptr(self.x_real), _abi.ptr(self.x_fake), _abi.ptr(self.depth), _abi.ptr(self.w_fwd),
                                   _abi.ptr(self.w_bwd), _abi.ptr(self.n_rows), _abi.ptr(ws), ws.numel(), _abi.stream()),
                   "aig_gp_rows")
        self.plan.rebuild(self.depth)


class _GATAggregate(torch.autograd.Function):
    """softmax-attention aggregation + bias + ELU over a GraphCSR (first-order differentiable)."""

    @staticmethod
    def forward(ctx, xw, att_src, att_dst, lin_edge_w, att_edge, bias, csr: GraphCSR, terms=None):
        lib = _abi.lib()
        n, dev = csr.n, xw.device
        xw = xw.contiguous()
        att_src, att_dst, lin_edge_w, att_edge, bias = (t.detach().contiguous().view(-1)
                                                        for t in (att_src, att_dst, lin_edge_w, att_edge, bias))
        f32 = dict(dtype=torch.float32, device=dev)
        out = torch.empty((n, HIDDEN), **f32)
        alpha_e = torch.empty((max(csr.E, 1), 4), **f32)
        alpha_loop = torch.empty((max(n, 1), 4), **f32)
        st = _abi.stream()
        if terms is not None:       # (the projection's launch made them, see _ProjectTerms)
            a_s, a_d, ce = terms
        else:
            ce = torch.empty(4, **f32)
            a_s, a_d = torch.empty((max(n, 1), 4), **f32), torch.empty((max(n, 1), 4), **f32)
            _abi.check(_abi.timed("aig_gat_node_alpha", lib.aig_gat_node_alpha, n, _abi.ptr(xw), _abi.ptr(att_src),
                                  _abi.ptr(att_dst), _abi.ptr(lin_edge_w), _abi.ptr(att_edge), _abi.ptr(a_s), _abi.ptr(a_d),
                                  _abi.ptr(ce), st), "aig_gat_node_alpha")
        _abi.check(_abi.timed("aig_gat_fwd", lib.aig_gat_fwd, n, _abi.ptr(xw), _abi.ptr(a_s), _abi.ptr(a_d),
                              _abi.ptr(ce), _abi.ptr(bias), _abi.ptr(csr.row_ptr), _abi.ptr(csr.csr_src),
                              _abi.ptr(csr.csr_ea), _abi.ptr(csr.loop_ea), _abi.ptr(out), _abi.ptr(alpha_e),
                              _abi.ptr(alpha_loop), st, meta=(n, csr.E)), "aig_gat_fwd")
        ctx.csr = csr
        ctx.save_for_backward(xw, att_src, att_dst, ce, a_s, a_d, out, alpha_e, alpha_loop, lin_edge_w, att_edge)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out):
        lib = _abi.lib()
        csr: GraphCSR = ctx.csr
        xw, att_src, att_dst, ce, a_s, a_d, out, alpha_e, alpha_loop, lin_edge_w, att_edge = ctx.saved_tensors
        n, dev = csr.n, xw.device
        g_out = g_out.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        d_xw = torch.empty((n, HIDDEN), **f32)
        d_as, d_ad, d_bias = torch.empty(HIDDEN, **f32), torch.empty(HIDDEN, **f32), torch.empty(HIDDEN, **f32)
        d_ce = torch.empty(4, **f32)
        d_lw, d_ae = torch.empty((HIDDEN, 1), **f32), torch.empty((1, 4, HIDDEN // 4), **f32)
        ws = _abi.workspace(lib.aig_gat_bwd_ws_bytes(n, csr.E), dev)
        _abi.check(_abi.timed("aig_gat_bwd", lib.aig_gat_bwd, n, csr.E, _abi.ptr(g_out), _abi.ptr(out), _abi.ptr(xw), _abi.ptr(att_src),
                                   _abi.ptr(att_dst), _abi.ptr(ce), _abi.ptr(a_s), _abi.ptr(a_d),
                                   _abi.ptr(csr.row_ptr), _abi.ptr(csr.csr_src), _abi.ptr(csr.csr_ea),
                                   _abi.ptr(csr.loop_ea), _abi.ptr(csr.col_ptr), _abi.ptr(csr.csc_pos),
                                   _abi.ptr(csr.csc_dst), _abi.ptr(csr.hub_tab), _abi.ptr(alpha_e),
                                   _abi.ptr(alpha_loop), _abi.ptr(lin_edge_w), _abi.ptr(att_edge),
                                   _abi.ptr(d_xw), _abi.ptr(d_as), _abi.ptr(d_ad), _abi.ptr(d_ce), _abi.ptr(d_bias),
                                   _abi.ptr(d_lw), _abi.ptr(d_ae),
                                   _abi.ptr(ws), ws.numel(), _abi.stream(), meta=(n, csr.E)), "aig_gat_bwd")
        return d_xw, d_as, d_ad, d_lw, d_ae, d_bias, None, None


class _GAT1Fused(torch.autograd.Function):
    """The whole first attention layer (projection, attention, aggregation, bias, ELU) on the two node features
    without any [n,128] intermediate: ``aig_gat1_fwd`` / ``aig_gat1_bwd``.  Differentiable once, w.r.t. the layer's
    parameters; the node features get no gradient (callers whose features need one take the generic path)."""

    @staticmethod
    def forward(ctx, x, w, att_src, att_dst, lin_edge_w, att_edge, bias, csr: GraphCSR):
        lib = _abi.lib()
        n, dev = csr.n, x.device
        x = x.detach().contiguous()
        w, att_src, att_dst, lin_edge_w, att_edge, bias = (t.detach().contiguous() for t in (w, att_src, att_dst, lin_edge_w,
                                                                                           att_edge, bias))
        f32 = dict(dtype=torch.float32, device=dev)
        coef = torch.empty(20, **f32)
        out = torch.empty((n, HIDDEN), **f32)
        alpha_e = torch.empty((max(csr.E, 1), 4), **f32)
        alpha_loop = torch.empty((max(n, 1), 4), **f32)
        _abi.check(_abi.timed("aig_gat1_fwd", lib.aig_gat1_fwd, n, _abi.ptr(x), _abi.ptr(w), _abi.ptr(att_src),
                              _abi.ptr(att_dst), _abi.ptr(lin_edge_w), _abi.ptr(att_edge), _abi.ptr(bias),
                              _abi.ptr(csr.row_ptr), _abi.ptr(csr.csr_src), _abi.ptr(csr.csr_ea), _abi.ptr(csr.loop_ea),
                              _abi.ptr(coef), _abi.ptr(out), _abi.ptr(alpha_e), _abi.ptr(alpha_loop), _abi.stream(),
                              meta=(n, csr.E)), "aig_gat1_fwd")
        ctx.csr = csr
        ctx.save_for_backward(x, w, att_src, att_dst, lin_edge_w, att_edge, coef, out, alpha_e, alpha_loop)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out):
        lib = _abi.lib()
        csr: GraphCSR = ctx.csr
        x, w, att_src, att_dst, lin_edge_w, att_edge, coef, out, alpha_e, alpha_loop = ctx.saved_tensors
        n, dev = csr.n, x.device
        g_out = g_out.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        d_w = torch.empty((HIDDEN, 2), **f32)
        d_as, d_ad, d_bias = torch.empty((1, 4, HIDDEN // 4), **f32), torch.empty((1, 4, HIDDEN // 4), **f32), torch.empty(HIDDEN, **f32)
        d_ce = torch.empty(4, **f32)
        d_lw, d_ae = torch.empty((HIDDEN, 1), **f32), torch.empty((1, 4, HIDDEN // 4), **f32)
        ws = _abi.workspace(lib.aig_gat1_bwd_ws_bytes(n), dev)
        _abi.check(_abi.timed("aig_gat1_bwd", lib.aig_gat1_bwd, n, _abi.ptr(x), _abi.ptr(w), _abi.ptr(att_src), _abi.ptr(att_dst),
                              _abi.ptr(lin_edge_w), _abi.ptr(att_edge), _abi.ptr(coef), _abi.ptr(g_out), _abi.ptr(out),
                              _abi.ptr(csr.row_ptr), _abi.ptr(csr.csr_src), _abi.ptr(csr.csr_ea), _abi.ptr(csr.loop_ea),
                              _abi.ptr(alpha_e), _abi.ptr(alpha_loop), _abi.ptr(d_w), _abi.ptr(d_as), _abi.ptr(d_ad),
                              _abi.ptr(d_ce), _abi.ptr(d_bias), _abi.ptr(d_lw), _abi.ptr(d_ae), _abi.ptr(ws), ws.numel(),
                              _abi.stream(), meta=(n, csr.E)), "aig_gat1_bwd")
        return None, d_w, d_as, d_ad, d_lw, d_ae, d_bias, None


class _GATAggregatePool(torch.autograd.Function):
    """An attention layer's aggregation (as ``_GATAggregate``) followed by the level readout (as ``_LevelPool``),
    i.e. the critic's second layer and what consumes it (discriminator.py:61-91), as ONE autograd node: the backward
    pass forms every row of the aggregation's incoming gradient from the readout's 256-float gradient inside the
    attention kernel (``aig_gat_bwd_pooled``) instead of writing it as an [n,128] array and reading it back.
    Same numbers as the two nodes in a row, bit for bit.  Differentiable once."""

    @staticmethod
    def forward(ctx, xw, att_src, att_dst, lin_edge_w, att_edge, bias, csr: GraphCSR, plan, terms=None):
        lib = _abi.lib()
        n, dev = csr.n, xw.device
        xw = xw.contiguous()
        att_src, att_dst, lin_edge_w, att_edge, bias = (t.detach().contiguous().view(-1)
                                                        for t in (att_src, att_dst, lin_edge_w, att_edge, bias))
        f32 = dict(dtype=torch.float32, device=dev)
        out = torch.empty((n, HIDDEN), **f32)
        alpha_e = torch.empty((max(csr.E, 1), 4), **f32)
        alpha_loop = torch.empty((max(n, 1), 4), **f32)
        st = _abi.stream()
        if terms is not None:       # (the projection's launch made them, see _ProjectTerms)
            a_s, a_d, ce = terms
        else:
            ce = torch.empty(4, **f32)
            a_s, a_d = torch.empty((max(n, 1), 4), **f32), torch.empty((max(n, 1), 4), **f32)
            _abi.check(_abi.timed("aig_gat_node_alpha", lib.aig_gat_node_alpha, n, _abi.ptr(xw), _abi.ptr(att_src),
                                  _abi.ptr(att_dst), _abi.ptr(lin_edge_w), _abi.ptr(att_edge), _abi.ptr(a_s), _abi.ptr(a_d),
                                  _abi.ptr(ce), st), "aig_gat_node_alpha")
        # (the aggregation's launch also clears the readout's table of maxima at the head of the readout's workspace)
        ws = _abi.workspace(lib.aig_level_pool_fwd_ws_bytes(plan.n, plan.L), dev)
        _abi.check(_abi.timed("aig_gat_fwd", lib.aig_gat_fwd_clearing, n, _abi.ptr(xw), _abi.ptr(a_s), _abi.ptr(a_d),
                              _abi.ptr(ce), _abi.ptr(bias), _abi.ptr(csr.row_ptr), _abi.ptr(csr.csr_src),
                              _abi.ptr(csr.csr_ea), _abi.ptr(csr.loop_ea), _abi.ptr(out), _abi.ptr(alpha_e),
                              _abi.ptr(alpha_loop), _abi.ptr(ws), lib.aig_level_pool_table_bytes(plan.L), st,
                              meta=(n, csr.E)), "aig_gat_fwd_clearing")
        r = torch.empty((1, 2 * HIDDEN), **f32)
        argmax = torch.empty((plan.L + 1, HIDDEN), dtype=torch.int32, device=dev)
        _abi.check(_abi.timed("aig_level_pool_fwd", lib.aig_level_pool_fwd_cleared, plan.n, plan.nvalid, plan.L, _abi.ptr(plan.meta),
                              _abi.ptr(out), _abi.ptr(plan.perm), _abi.ptr(plan.lvl_ptr), _abi.ptr(plan.node_level),
                              _abi.ptr(plan.w), _abi.ptr(r), _abi.ptr(argmax), _abi.ptr(ws), ws.numel(), st),
                   "aig_level_pool_fwd")
        ctx.csr, ctx.plan = csr, plan
        ctx.save_for_backward(xw, att_src, att_dst, ce, a_s, a_d, out, alpha_e, alpha_loop, lin_edge_w, att_edge, argmax)
        return r

    @staticmethod
    @once_differentiable
    def backward(ctx, g_r):
        lib = _abi.lib()
        csr, plan = ctx.csr, ctx.plan
        xw, att_src, att_dst, ce, a_s, a_d, out, alpha_e, alpha_loop, lin_edge_w, att_edge, argmax = ctx.saved_tensors
        n, dev = csr.n, xw.device
        g_r = g_r.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        d_xw = torch.empty((n, HIDDEN), **f32)
        d_as, d_ad, d_bias = torch.empty(HIDDEN, **f32), torch.empty(HIDDEN, **f32), torch.empty(HIDDEN, **f32)
        d_ce = torch.empty(4, **f32)
        d_lw, d_ae = torch.empty((HIDDEN, 1), **f32), torch.empty((1, 4, HIDDEN // 4), **f32)
        ws = _abi.workspace(lib.aig_gat_bwd_ws_bytes(n, csr.E), dev)
        _abi.check(_abi.timed("aig_gat_bwd", lib.aig_gat_bwd_pooled, n, csr.E, plan.L, _abi.ptr(plan.meta), _abi.ptr(g_r),
                              _abi.ptr(plan.node_level), _abi.ptr(plan.w), _abi.ptr(argmax), _abi.ptr(out), _abi.ptr(xw),
                              _abi.ptr(att_src), _abi.ptr(att_dst), _abi.ptr(ce), _abi.ptr(a_s), _abi.ptr(a_d),
                              _abi.ptr(csr.row_ptr), _abi.ptr(csr.csr_src), _abi.ptr(csr.csr_ea), _abi.ptr(csr.loop_ea),
                              _abi.ptr(csr.col_ptr), _abi.ptr(csr.csc_pos), _abi.ptr(csr.csc_dst), _abi.ptr(csr.hub_tab),
                              _abi.ptr(alpha_e), _abi.ptr(alpha_loop), _abi.ptr(lin_edge_w), _abi.ptr(att_edge),
                              _abi.ptr(d_xw), _abi.ptr(d_as), _abi.ptr(d_ad), _abi.ptr(d_ce), _abi.ptr(d_bias),
                              _abi.ptr(d_lw), _abi.ptr(d_ae), _abi.ptr(ws), ws.numel(), _abi.stream(), meta=(n, csr.E)),
                   "aig_gat_bwd_pooled")
        return d_xw, d_as, d_ad, d_lw, d_ae, d_bias, None, None, None


class _WGrad(torch.autograd.Function):
    """dW[j,f] = sum_n g[n,j] x[n,f]: the node-dimension reduction of a projection's backward pass."""

    @staticmethod
    def forward(ctx, g, x):
        lib = _abi.lib()
        g, x = g.contiguous(), x.contiguous()
        n, F_in = int(x.size(0)), int(x.size(1))
        dW = torch.empty((HIDDEN, F_in), dtype=torch.float32, device=x.device)
        ws = _abi.workspace(lib.aig_wgrad_ws_bytes(n, F_in), x.device)
        _abi.check(_abi.timed("aig_wgrad", lib.aig_wgrad, n, F_in, _abi.ptr(g), _abi.ptr(x), _abi.ptr(dW),
                              _abi.ptr(ws), ws.numel(), _abi.stream()), "aig_wgrad")
        ctx.save_for_backward(g, x)
        return dW

    @staticmethod
    def backward(ctx, gg):
        g, x = ctx.saved_tensors
        return _matmul_wt(x, gg), _matmul_w(g, gg)


def _linear128(x: torch.Tensor, w: torch.Tensor, in_major: bool) -> torch.Tensor:
    """x w^T (in_major False) or x w (in_major True) for x [n,128], w [128,128] on the projection kernel."""
    x, w = x.contiguous(), w.contiguous()
    y = torch.empty_like(x)
    _abi.check(_abi.timed("aig_linear128", _abi.lib().aig_linear128, int(x.size(0)), _abi.ptr(x), _abi.ptr(w),
                          1 if in_major else 0, None, 0, _abi.ptr(y), _abi.stream()), "aig_linear128")
    return y


def _linear_in2(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """x w^T for the two node features, x [n,2], w [128,2] (the first attention layer's projection)."""
    x, w = x.contiguous(), w.contiguous()
    y = torch.empty((x.size(0), HIDDEN), dtype=torch.float32, device=x.device)
    _abi.check(_abi.timed("aig_linear_in2", _abi.lib().aig_linear_in2, int(x.size(0)), _abi.ptr(x), _abi.ptr(w),
                          _abi.ptr(y), _abi.stream()), "aig_linear_in2")
    return y


def _is_in2(x: torch.Tensor, w: torch.Tensor) -> bool:
    return (x.dim() == 2 and x.size(1) == 2 and tuple(w.shape) == (HIDDEN, 2)
            and x.dtype == torch.float32 and x.is_cuda)


def _is128(x: torch.Tensor, w: torch.Tensor) -> bool:
    return x.dim() == 2 and x.size(1) == HIDDEN and tuple(w.shape) == (HIDDEN, HIDDEN) and x.dtype == torch.float32


def _matmul_wt(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """x @ w.t(), differentiable (through _Project) when the shapes fit the kernel."""
    return _Project.apply(x, w) if _is128(x, w) else x @ w.t()


def _matmul_w(g: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """g @ w, differentiable (through _ProjectBack) when the shapes fit the kernel."""
    return _ProjectBack.apply(g, w) if _is128(g, w) else g @ w


class _Project(torch.autograd.Function):
    """x W^T with the weight gradient on the kernel above; differentiable twice (its backward is
    made of differentiable pieces), which the gradient penalty needs."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        if _is_in2(x, w):
            return _linear_in2(x, w)
        return _linear128(x, w, False) if _is128(x, w) else x @ w.t()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        d_x = _ProjectBack.apply(g, w) if ctx.needs_input_grad[0] else None
        d_w = _WGrad.apply(g, x) if ctx.needs_input_grad[1] else None
        return d_x, d_w


class _ProjectBack(torch.autograd.Function):
    """g W (the input gradient of a projection); its own weight gradient - reached when the
    gradient penalty differentiates a second time - is again a reduction over the nodes."""

    @staticmethod
    def forward(ctx, g, w):
        ctx.save_for_backward(g, w)
        return _linear128(g, w, True) if _is128(g, w) else g @ w

    @staticmethod
    def backward(ctx, gg):
        g, w = ctx.saved_tensors
        d_g = _matmul_wt(gg, w) if ctx.needs_input_grad[0] else None
        d_w = _WGrad.apply(g, gg) if ctx.needs_input_grad[1] else None
        return d_g, d_w


class _ProjectTerms(torch.autograd.Function):
    """``_Project`` for the projection of an attention layer, which also hands over the layer's node terms
    ``a_s``, ``a_d`` and edge coefficients ``ce`` (``aig_linear128_node_terms``: one launch instead of a pass of
    their own over the projected rows).  The three carry no gradient here: the aggregation that consumes them
    differentiates through them itself, as when it computes them."""

    @staticmethod
    def forward(ctx, x, w, att_src, att_dst, lin_edge_w, att_edge):
        ctx.save_for_backward(x, w)
        x_, w_ = x.contiguous(), w.contiguous()
        vecs = [t.detach().contiguous().view(-1) for t in (att_src, att_dst, lin_edge_w, att_edge)]
        n = int(x_.size(0))
        f32 = dict(dtype=torch.float32, device=x_.device)
        y = torch.empty_like(x_)
        a_s, a_d, ce = torch.empty((max(n, 1), 4), **f32), torch.empty((max(n, 1), 4), **f32), torch.empty(4, **f32)
        _abi.check(_abi.timed("aig_linear128", _abi.lib().aig_linear128_node_terms, n, _abi.ptr(x_), _abi.ptr(w_), 0,
                              _abi.ptr(vecs[0]), _abi.ptr(vecs[1]), _abi.ptr(vecs[2]), _abi.ptr(vecs[3]), _abi.ptr(y),
                              _abi.ptr(a_s), _abi.ptr(a_d), _abi.ptr(ce), _abi.stream()), "aig_linear128_node_terms")
        ctx.mark_non_differentiable(a_s, a_d, ce)
        return y, a_s, a_d, ce

    @staticmethod
    def backward(ctx, g, *_unused):
        x, w = ctx.saved_tensors
        d_x = _ProjectBack.apply(g, w) if ctx.needs_input_grad[0] else None
        d_w = _WGrad.apply(g, x) if ctx.needs_input_grad[1] else None
        return d_x, d_w, None, None, None, None


class _LevelPool(torch.autograd.Function):
    """[n,128] -> [1,256] level-wise mean || max readout."""

    @staticmethod
    def forward(ctx, h, plan: LevelPlan):
        lib = _abi.lib()
        dev = h.device
        h = h.contiguous()
        out = torch.empty((1, 2 * HIDDEN), dtype=torch.float32, device=dev)
        argmax = torch.empty((plan.L + 1, HIDDEN), dtype=torch.int32, device=dev)
        ws = _abi.workspace(lib.aig_level_pool_fwd_ws_bytes(plan.n, plan.L), dev)
        _abi.check(_abi.timed("aig_level_pool_fwd", lib.aig_level_pool_fwd, plan.n, plan.nvalid, plan.L, _abi.ptr(plan.meta), _abi.ptr(h), _abi.ptr(plan.perm),
                                          _abi.ptr(plan.lvl_ptr), _abi.ptr(plan.node_level), _abi.ptr(plan.w),
                                          _abi.ptr(out), _abi.ptr(argmax), _abi.ptr(ws), ws.numel(), _abi.stream()),
                   "aig_level_pool_fwd")
        ctx.plan = plan
        ctx.n = int(h.size(0))
        ctx.save_for_backward(argmax)
        return out

    @staticmethod
    def backward(ctx, g):
        (argmax,) = ctx.saved_tensors
        return _LevelPoolBwd.apply(g, ctx.plan, argmax, ctx.n), None


class _LevelPoolBwd(torch.autograd.Function):
    """Gradient of the readout w.r.t. the node rows; linear in ``g``, so its own backward (reached
    when the gradient penalty differentiates a second time) is the adjoint map."""

    @staticmethod
    def forward(ctx, g, plan: LevelPlan, argmax, n: int):
        g = g.contiguous()
        d_h = torch.empty((n, HIDDEN), dtype=torch.float32, device=g.device)
        _abi.check(_abi.lib().aig_level_pool_bwd(n, plan.L, _abi.ptr(plan.meta), _abi.ptr(g), _abi.ptr(plan.node_level), _abi.ptr(plan.w),
                                                 _abi.ptr(argmax), _abi.ptr(d_h), _abi.stream()), "aig_level_pool_bwd")
        ctx.plan, ctx.n = plan, n
        ctx.save_for_backward(argmax)
        return d_h

    @staticmethod
    def backward(ctx, gg):
        lib = _abi.lib()
        (argmax,) = ctx.saved_tensors
        plan, n = ctx.plan, ctx.n
        gg = gg.contiguous()
        out = torch.empty((1, 2 * HIDDEN), dtype=torch.float32, device=gg.device)
        ws = _abi.workspace(lib.aig_level_pool_bwd_adj_ws_bytes(n), gg.device)
        _abi.check(lib.aig_level_pool_bwd_adj(n, plan.L, _abi.ptr(plan.meta), _abi.ptr(gg), _abi.ptr(plan.node_level), _abi.ptr(plan.w),
                                              _abi.ptr(argmax), _abi.ptr(out), _abi.ptr(ws), ws.numel(), _abi.stream()),
                   "aig_level_pool_bwd_adj")
        return out, None, None, None


def _glorot_(t: torch.Tensor) -> torch.Tensor:
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        return t.uniform_(-a, a)


class GATLayer(nn.Module):
    """Parameters of ``GATConv(in_dim, hidden/heads, heads, edge_dim=1, concat=True)``; the forward
    pass also applies the ELU that follows each layer in the critic (fused in the kernel)."""

    def __init__(self, in_dim: int, out_per_head: int, heads: int):
        super().__init__()
        self.heads, self.out_per_head = heads, out_per_head
        self.lin = nn.Linear(in_dim, heads * out_per_head, bias=False)
        self.lin_edge = nn.Linear(1, heads * out_per_head, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_per_head))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_per_head))
        self.att_edge = nn.Parameter(torch.empty(1, heads, out_per_head))
        self.bias = nn.Parameter(torch.zeros(heads * out_per_head))
        for t in (self.lin.weight, self.lin_edge.weight, self.att_src, self.att_dst, self.att_edge):
            _glorot_(t)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # older PyG releases store the shared projection as lin_src (and a tied lin_dst)
        src, dst = prefix + "lin_src.weight", prefix + "lin_dst.weight"
        if src in state_dict and prefix + "lin.weight" not in state_dict:
            state_dict[prefix + "lin.weight"] = state_dict.pop(src)
            state_dict.pop(dst, None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward_by_torch_ops(self, x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor) -> torch.Tensor:
        """The layer (and the ELU behind it) out of torch operations alone, for the one caller that differentiates
        the critic TWICE through its edges and w.r.t. its edge attributes: ``gradient_penalty(use_edges=True)``
        (train.py:154-185, a branch the reference's loop never takes).  The kernels' backward passes are
        differentiable once; this form is for correctness there, not for speed.  GATConv's steps: self loops are
        dropped and one per node is appended whose attribute is the mean attribute of the edges entering the node
        (0 without any); e = leaky_relu(a_s[src] + a_d[dst] + a_e, 0.2); softmax per target with the maximum
        taken out (no gradient through it) and 1e-16 added to the sum; bias after the heads are concatenated."""
        n, H, C = int(x.size(0)), self.heads, self.out_per_head
        ea = edge_attr.reshape(-1, 1).to(x.dtype)
        ok = (edge_index.min(dim=0).values >= 0) & (edge_index[0] != edge_index[1])
        src, dst, ea = edge_index[0][ok], edge_index[1][ok], ea[ok]
        entering = torch.zeros(n, dtype=x.dtype, device=x.device).index_add(0, dst, torch.ones_like(ea[:, 0]))
        loop_ea = torch.zeros((n, 1), dtype=x.dtype, device=x.device).index_add(0, dst, ea) / entering.clamp(min=1).unsqueeze(1)
        ids = torch.arange(n, device=x.device, dtype=src.dtype)
        src, dst, ea = torch.cat([src, ids]), torch.cat([dst, ids]), torch.cat([ea, loop_ea])
        xw = (x @ self.lin.weight.t()).view(n, H, C)
        a_s, a_d = (xw * self.att_src).sum(-1), (xw * self.att_dst).sum(-1)
        a_e = ((ea @ self.lin_edge.weight.t()).view(-1, H, C) * self.att_edge).sum(-1)
        e = F.leaky_relu(a_s[src] + a_d[dst] + a_e, 0.2)
        top = torch.full((n, H), float("-inf"), dtype=x.dtype, device=x.device).scatter_reduce(
            0, dst.unsqueeze(1).expand(-1, H), e.detach(), "amax", include_self=True)
        w = torch.exp(e - top[dst])
        total = torch.zeros((n, H), dtype=x.dtype, device=x.device).index_add(0, dst, w)
        alpha = w / (total[dst] + 1e-16)
        out = torch.zeros((n, H, C), dtype=x.dtype, device=x.device).index_add(0, dst, alpha.unsqueeze(-1) * xw[src])
        return F.elu(out.reshape(n, H * C) + self.bias)

    def forward(self, x: torch.Tensor, csr: Optional[GraphCSR], plan=None) -> torch.Tensor:
        """``plan``: a level plan - the layer's output then goes straight into the level readout and the
        readout's [1,256] row is returned (the critic's second layer; edge-carrying graphs only)."""
        if (plan is None and csr is not None and csr.E > 0 and _is_in2(x, self.lin.weight) and not x.requires_grad
                and FUSED_FIRST_LAYER):
            # two input features: the whole layer without its projected rows (see _GAT1Fused)
            return _GAT1Fused.apply(x, self.lin.weight, self.att_src, self.att_dst, self.lin_edge.weight, self.att_edge,
                                    self.bias, csr)
        has_edges = csr is not None and csr.E > 0
        terms = None
        if (has_edges and _is128(x, self.lin.weight) and x.is_cuda and FUSED_NODE_TERMS
                and int(x.size(0)) >= NODE_TERMS_MIN_ROWS):
            # the projection's launch also makes the node terms of the attention (see _ProjectTerms)
            xw, a_s, a_d, ce = _ProjectTerms.apply(x, self.lin.weight, self.att_src, self.att_dst, self.lin_edge.weight,
                                                   self.att_edge)
            terms = (a_s, a_d, ce)
        elif x.size(1) in (1, 2, 4, HIDDEN) and self.lin.weight.size(0) == HIDDEN:
            xw = _Project.apply(x, self.lin.weight)
        else:
            xw = self.lin(x)
        if not has_edges:
            # only self loops: alpha = 1/(1 + 1e-16) = 1 in fp32
            return F.elu(xw + self.bias)
        # edge_dim = 1: an edge's attention term is attr * ce[h], ce[h] = <lin_edge.weight[h,:], att_edge[h,:]>,
        # formed (and differentiated) inside the kernels
        if plan is not None:
            return _GATAggregatePool.apply(xw, self.att_src.reshape(-1), self.att_dst.reshape(-1), self.lin_edge.weight,
                                           self.att_edge, self.bias, csr, plan, terms)
        return _GATAggregate.apply(xw, self.att_src.reshape(-1), self.att_dst.reshape(-1), self.lin_edge.weight,
                                   self.att_edge, self.bias, csr, terms)


class _CriticHead(torch.autograd.Function):
    """``out(mlp(r))`` for one pooled row (discriminator.py:98-100): one launch forward, one backward.
    Differentiable once; the gradient penalty's second-order path uses the torch modules or the closed form."""

    @staticmethod
    def forward(ctx, r, w1, b1, w2, b2, w3, b3):
        lib = _abi.lib()
        tensors = [t.contiguous() for t in (r, w1, b1, w2, b2, w3, b3)]
        r_, w1_, b1_, w2_, b2_, w3_, b3_ = tensors
        score = torch.empty(1, dtype=torch.float32, device=r.device)
        act = torch.empty(192, dtype=torch.float32, device=r.device)
        _abi.check(lib.aig_head_fwd(_abi.ptr(r_), _abi.ptr(w1_), _abi.ptr(b1_), _abi.ptr(w2_), _abi.ptr(b2_),
                                    _abi.ptr(w3_), _abi.ptr(b3_), _abi.ptr(score), _abi.ptr(act), _abi.stream()),
                   "aig_head_fwd")
        ctx.save_for_backward(r_, w1_, w2_, w3_, act)
        return score

    @staticmethod
    @once_differentiable
    def backward(ctx, g_score):
        lib = _abi.lib()
        r, w1, w2, w3, act = ctx.saved_tensors
        f32 = dict(dtype=torch.float32, device=r.device)
        d_r = torch.empty((1, 256), **f32)
        d_w1, d_b1 = torch.empty((128, 256), **f32), torch.empty(128, **f32)
        d_w2, d_b2 = torch.empty((64, 128), **f32), torch.empty(64, **f32)
        d_w3, d_b3 = torch.empty((1, 64), **f32), torch.empty(1, **f32)
        _abi.check(lib.aig_head_bwd(_abi.ptr(g_score.contiguous()), _abi.ptr(r), _abi.ptr(w1), _abi.ptr(w2), _abi.ptr(w3),
                                    _abi.ptr(act), _abi.ptr(d_r), _abi.ptr(d_w1), _abi.ptr(d_b1), _abi.ptr(d_w2),
                                    _abi.ptr(d_b2), _abi.ptr(d_w3), _abi.ptr(d_b3), _abi.stream()), "aig_head_bwd")
        return d_r, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3


class _Cache(OrderedDict):
    def __init__(self, cap: int = 16):
        super().__init__()
        self.cap = cap

    def lookup(self, key, make):
        if key in self:
            self.move_to_end(key)
            return self[key]
        v = make()
        self[key] = v
        if len(self) > self.cap:
            self.popitem(last=False)
        return v


def _tkey(t: Optional[torch.Tensor]):
    return None if t is None else (t.data_ptr(), t._version, tuple(t.shape), t.dtype)


class AIGDiscriminator(nn.Module):
    def __init__(self, in_dim: int = 2, hidden_dim: int = 128, heads: int = 4):
        super().__init__()
        assert hidden_dim % heads == 0, "hidden_dim must be divisible by heads"
        self.hidden_dim = hidden_dim
        self.heads = heads
        self.gat1 = GATLayer(in_dim, hidden_dim // heads, heads)
        self.gat2 = GATLayer(hidden_dim, hidden_dim // heads, heads)
        self.mlp = nn.Sequential(nn.Linear(hidden_dim * 2, hidden_dim), nn.ReLU(),
                                 nn.Linear(hidden_dim, hidden_dim // 2), nn.ReLU())
        self.out = nn.Linear(hidden_dim // 2, 1)
        if (hidden_dim, heads) != (128, 4):
            raise NotImplementedError("the MI355X kernels are built for hidden_dim=128, heads=4 (reference defaults)")
        self._csr_cache = _Cache()
        self._lvl_cache = _Cache()

    # graph structure is weight-independent, so the several critic evaluations of one training
    # step (train.py:246-283 calls D ten times on three edge lists) share one CSR / level plan
    def graph_csr(self, edge_index, edge_attr, n) -> GraphCSR:
        key = (_tkey(edge_index), _tkey(edge_attr), n)
        # keep the keyed tensors alive with the entry so a recycled address cannot alias
        return self._csr_cache.lookup(key, lambda: (GraphCSR(edge_index, edge_attr, n), edge_index, edge_attr))[0]

    def level_plan(self, node_depth, n, device) -> LevelPlan:
        key = (_tkey(node_depth), n)
        return self._lvl_cache.lookup(key, lambda: (LevelPlan(node_depth, n, device), node_depth))[0]

    def gradient_penalty_closed_form(self, x_real: torch.Tensor, x_fake: torch.Tensor, alpha: torch.Tensor,
                                     node_depth: Optional[torch.Tensor], plan=None, grad_scale: float = 1.0,
                                     rows: Optional["GpRows"] = None, out_scale: float = 1.0):
        """``gradient_penalty(use_edges=False)`` of train.py:133-152 and its gradient w.r.t. the critic's
        parameters without autograd on the node tensors: on an empty edge list each attention layer
        is ``elu(x W^T + b)``, which makes the penalty a closed form (four [n,128]x[128,128]
        products, three node-dimension weight gradients, a few row-wise passes; the head's input
        gradient and its second-order share come from the head kernels).  Returns ``(gp, grad_norm, grads)`` with ``grads`` a dict
        parameter -> gradient of ``gp`` (parameters the edge-free pass never touches are absent).  ``plan``: a
        level plan of ``node_depth`` the caller already holds (else it is looked up / built here); ``grad_scale``
        multiplies every returned gradient (the loss weights the penalty by gp_lambda).  ``rows``: a ``GpRows``
        whose last ``rebuild`` was made for these very features and levels - the computation then runs over its
        distinct rows (same numbers up to the order of the sums); ``x_real`` / ``x_fake`` / ``plan`` are not read.
        ``rows=True``: the rows are built here, for this call.  ``out_scale`` multiplies every returned gradient
        AFTER the fact - one more rounding per element, the bits of ``out_scale * gradient`` as train.py:258 forms
        ``gp_lambda * gp`` - inside the launches that write them."""
        lib = _abi.lib()
        dev = x_real.device
        n = int(x_real.size(0))
        f32 = dict(dtype=torch.float32, device=dev)
        xr, xf = x_real.float().contiguous(), x_fake.float().contiguous()
        n_dev = None
        if rows is True:        # rows of this very call, in buffers kept per (n, level capacity)
            if plan is None:
                plan = self.level_plan(node_depth, n, dev)
            cache = self.__dict__.setdefault("_gp_rows_cache", {})
            rows = cache.get((n, plan.L))
            if rows is None:
                if len(cache) >= 4:
                    cache.pop(next(iter(cache)))
                rows = cache[(n, plan.L)] = GpRows(n, plan.L, dev)
            rows.rebuild(xr, xf, plan)
        if rows is not None:
            if rows.n != n:
                raise _abi.AigAbiError("gradient_penalty_closed_form: rows were planned for another node count")
            xr, xf, plan, n_dev = rows.x_real, rows.x_fake, rows.plan, rows.n_rows
        w_fwd = rows.w_fwd if rows is not None else None
        w_bwd = rows.w_bwd if rows is not None else None
        al = alpha.reshape(-1)[:1].float().contiguous()
        W1, b1 = self.gat1.lin.weight.detach().contiguous(), self.gat1.bias.detach().contiguous()
        W2, b2 = self.gat2.lin.weight.detach().contiguous(), self.gat2.bias.detach().contiguous()
        if tuple(W1.shape) != (HIDDEN, 2):
            raise NotImplementedError("closed-form gradient penalty is built for in_dim=2")
        if plan is None:
            plan = self.level_plan(node_depth, n, dev)
        st = _abi.stream()
        xhat = torch.empty((n, 2), **f32)
        h1, h2 = torch.empty((n, HIDDEN), **f32), torch.empty((n, HIDDEN), **f32)
        if w_fwd is None:
            w_fwd = w_bwd = plan.w
        # (the forward launch also clears the readout's table of maxima at the head of the readout's workspace)
        ws = _abi.workspace(lib.aig_level_pool_fwd_ws_bytes(n, plan.L), dev)
        _abi.check(lib.aig_gp_forward_clearing(n, _abi.ptr(n_dev), _abi.ptr(xr), _abi.ptr(xf), _abi.ptr(al), _abi.ptr(W1),
                                               _abi.ptr(b1), _abi.ptr(W2), _abi.ptr(b2), _abi.ptr(xhat), _abi.ptr(h1),
                                               _abi.ptr(h2), _abi.ptr(ws), lib.aig_level_pool_table_bytes(plan.L), st),
                   "aig_gp_forward_clearing")
        r = torch.empty((1, 2 * HIDDEN), **f32)
        argmax = torch.empty((plan.L + 1, HIDDEN), dtype=torch.int32, device=dev)
        _abi.check(lib.aig_level_pool_fwd_cleared(n, plan.nvalid, plan.L, _abi.ptr(plan.meta), _abi.ptr(h2), _abi.ptr(plan.perm), _abi.ptr(plan.lvl_ptr),
                                          _abi.ptr(plan.node_level), _abi.ptr(w_fwd), _abi.ptr(r), _abi.ptr(argmax),
                                          _abi.ptr(ws), ws.numel(), st), "aig_level_pool_fwd")
        # head: s = out(mlp(r)), g_r = ds/dr (the backward kernel of the head with ds = 1)
        hw1, hb1 = self.mlp[0].weight.detach().contiguous(), self.mlp[0].bias.detach().contiguous()
        hw2, hb2 = self.mlp[2].weight.detach().contiguous(), self.mlp[2].bias.detach().contiguous()
        hw3, hb3 = self.out.weight.detach().contiguous(), self.out.bias.detach().contiguous()
        score, act = torch.empty(1, **f32), torch.empty(192, **f32)
        _abi.check(lib.aig_head_fwd(_abi.ptr(r), _abi.ptr(hw1), _abi.ptr(hb1), _abi.ptr(hw2), _abi.ptr(hb2),
                                    _abi.ptr(hw3), _abi.ptr(hb3), _abi.ptr(score), _abi.ptr(act), st), "aig_head_fwd")
        g_r_c = torch.empty((1, 2 * HIDDEN), **f32)
        one = self.__dict__.get("_one")       # (a constant kept per device: no fill launch per call)
        if one is None or one.device != dev:
            one = self.__dict__["_one"] = torch.ones(1, **f32)
        _abi.check(lib.aig_head_bwd(_abi.ptr(one), _abi.ptr(r), _abi.ptr(hw1), _abi.ptr(hw2), _abi.ptr(hw3), _abi.ptr(act),
                                    _abi.ptr(g_r_c), None, None, None, None, None, None, st), "aig_head_bwd")
        g_z2, g_h1, g_z1 = (torch.empty((n, HIDDEN), **f32) for _ in range(3))
        g_x, norm_sq = torch.empty((n, 2), **f32), torch.empty(1, **f32)
        # norm, gp = (norm - 1)^2 and coef = grad_scale * d gp / d norm / norm come out of the launch that adds up
        # |g_x|^2.  d gp / d g_x = coef * g_x; at norm = 0 autograd's torch.norm has the zero subgradient
        # (train.py:150), so the penalty's gradient vanishes there instead of becoming 0 * inf (decided on the
        # device, no host read)
        sc = torch.empty(3, **f32)
        ws = _abi.workspace(lib.aig_gp_input_grad_ws_bytes(n), dev)
        _abi.check(lib.aig_gp_input_grad_scalars(n, _abi.ptr(n_dev), plan.L, _abi.ptr(plan.meta), _abi.ptr(g_r_c),
                                                 _abi.ptr(plan.node_level), _abi.ptr(w_bwd), _abi.ptr(argmax), _abi.ptr(W1),
                                                 _abi.ptr(W2), _abi.ptr(h1), _abi.ptr(h2), _abi.ptr(g_z2), _abi.ptr(g_h1),
                                                 _abi.ptr(g_z1), _abi.ptr(g_x), _abi.ptr(norm_sq), float(grad_scale),
                                                 _abi.ptr(sc), _abi.ptr(ws), ws.numel(), st), "aig_gp_input_grad_scalars")
        norm, gp, coef = sc[0], sc[1], sc[2:3]
        dW1, db1 = torch.empty((HIDDEN, 2), **f32), torch.empty(HIDDEN, **f32)
        dW2, db2 = torch.empty((HIDDEN, HIDDEN), **f32), torch.empty(HIDDEN, **f32)
        u_gr = torch.empty((1, 2 * HIDDEN), **f32)
        ws = _abi.workspace(lib.aig_gp_param_grad_ws_bytes(n), dev)
        _abi.check(lib.aig_gp_param_grad_scaled(n, _abi.ptr(n_dev), plan.L, _abi.ptr(plan.meta), _abi.ptr(coef), _abi.ptr(g_r_c),
                                                _abi.ptr(plan.node_level), _abi.ptr(w_bwd), _abi.ptr(argmax), _abi.ptr(W1),
                                                _abi.ptr(W2), _abi.ptr(xhat), _abi.ptr(h1), _abi.ptr(h2), _abi.ptr(g_z2),
                                                _abi.ptr(g_h1), _abi.ptr(g_z1), _abi.ptr(g_x), _abi.ptr(dW1), _abi.ptr(db1),
                                                _abi.ptr(dW2), _abi.ptr(db2), _abi.ptr(u_gr), float(out_scale), _abi.ptr(ws),
                                                ws.numel(), st), "aig_gp_param_grad_scaled")
        grads = {self.gat1.lin.weight: dW1, self.gat1.bias: db1, self.gat2.lin.weight: dW2, self.gat2.bias: db2}
        # the head's share: gradient of <g_r, u_gr> w.r.t. its weights (its biases only move the ReLU masks)
        dhw1, dhw2, dhw3 = torch.empty((128, 256), **f32), torch.empty((64, 128), **f32), torch.empty((1, 64), **f32)
        _abi.check(lib.aig_head_input_grad_wgrad_scaled(_abi.ptr(u_gr), _abi.ptr(hw1), _abi.ptr(hw2), _abi.ptr(hw3), _abi.ptr(act),
                                                        _abi.ptr(dhw1), _abi.ptr(dhw2), _abi.ptr(dhw3), float(out_scale), st),
                   "aig_head_input_grad_wgrad_scaled")
        grads[self.mlp[0].weight], grads[self.mlp[2].weight], grads[self.out.weight] = dhw1, dhw2, dhw3
        return gp.detach(), norm.detach(), grads

    def _head_is_fused(self, r: torch.Tensor) -> bool:
        return tuple(r.shape) == (1, 2 * HIDDEN) and r.dtype == torch.float32 and self.hidden_dim == HIDDEN

    def forward(self, data: Data, twice_differentiable: bool = False) -> torch.Tensor:
        """The fused head kernel is differentiable once.  When the node features themselves require a
        gradient - the autograd form of the gradient penalty differentiates the critic w.r.t. its input and
        then once more (train.py:143-150) - or when ``twice_differentiable`` is set, the head runs as torch
        modules, which can be differentiated again."""
        x = data.x.float()
        twice_differentiable = twice_differentiable or x.requires_grad
        if x.device.type != "cuda":
            raise _abi.AigAbiError("AIGDiscriminator.forward needs the graph on a HIP device (no CPU path exists)")
        n = int(x.size(0))
        edge_index = data.edge_index
        has_edges = edge_index is not None and edge_index.numel() > 0
        edge_attr = data.edge_attr if (hasattr(data, "edge_attr") and data.edge_attr is not None) else None
        if twice_differentiable and has_edges:
            # differentiated twice THROUGH the edges (gradient_penalty(use_edges=True), train.py:154-185): the
            # attention layers out of torch operations, the readout and the head in their twice-differentiable forms
            _abi.lib()        # (the readout below is a kernel: no library, no critic)
            ea = edge_attr if edge_attr is not None else torch.zeros((edge_index.size(1), 1), dtype=x.dtype, device=x.device)
            h = self.gat2.forward_by_torch_ops(self.gat1.forward_by_torch_ops(x, edge_index, ea), edge_index, ea)
            nd_ = data.node_depth if (hasattr(data, "node_depth") and data.node_depth is not None) else None
            plan_ = data.__dict__.get("_plan") if hasattr(data, "__dict__") else None
            if plan_ is None:
                plan_ = self.level_plan(nd_, n, x.device)
            return self.out(self.mlp(_LevelPool.apply(h, plan_))).view(-1)
        # a caller that keeps its inputs in fixed buffers hands over the structures it rebuilt in place
        csr = data.__dict__.get("_csr") if hasattr(data, "__dict__") else None
        if csr is None:
            csr = self.graph_csr(edge_index, edge_attr, n) if has_edges else None
        elif csr.n != n:
            raise _abi.AigAbiError(f"data._csr was built for {csr.n} nodes, the graph has {n}")
        nd = data.node_depth if (hasattr(data, "node_depth") and data.node_depth is not None) else None
        plan = data.__dict__.get("_plan") if hasattr(data, "__dict__") else None
        if plan is None:
            plan = self.level_plan(nd, n, x.device)
        elif plan.n != n:
            raise _abi.AigAbiError(f"data._plan was built for {plan.n} nodes, the graph has {n}")
        h = self.gat1(x, csr)
        if csr is not None and csr.E > 0 and not twice_differentiable and FUSED_READOUT_BACKWARD:
            # second layer and readout as one autograd node (their backward passes share a kernel)
            graph_repr = self.gat2(h, csr, plan)
        else:
            h = self.gat2(h, csr)
            graph_repr = _LevelPool.apply(h, plan)           # [1, 2H]
        if not twice_differentiable and self._head_is_fused(graph_repr):
            return _CriticHead.apply(graph_repr, self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight,
                                     self.mlp[2].bias, self.out.weight, self.out.bias)
        g = self.mlp(graph_repr)
        return self.out(g).view(-1)
