"""WGAN-GP training loop of the AIG-GAN on MI355X kernels.

Same loop, flags and artefacts as train.py of the reference (train.py:191-387): per minibatch
the generator wires a fake per template once, the fakes are repaired and padded, the critic
takes ``--n-critic`` steps of ``(D_fake - D_real) + gp_lambda * GP`` with a node-only gradient
penalty, then the generator step evaluates ``-D(raw) + lambda_cons * penalties``; checkpoints and
``metrics.csv`` per epoch, sample ``.pt`` files every ``--save-every`` steps.

What differs, on purpose:
* every model / constraint call runs on the GPU kernels of this package (no CPU path);
* ``sanitize_for_D`` of the real batch is hoisted out of the critic loop (its result does not change
  between iterations), which lets the critic reuse one CSR per edge list; the fake batches go to the
  critic with their ``-1`` padding columns, which the CSR builder drops itself (a boolean-mask copy
  would make the host wait for the GPU), and the columns ``sanitize_for_D`` would keep are only counted;
* with ``WORLD_SIZE > 1`` (torchrun) the shuffled minibatches are dealt to the ranks and the
  critic gradient is averaged over ranks after every ``loss_D.backward()`` (gan_amd/parallel.py);
* the real-side critic term and the gradient penalty of a minibatch are recorded as HIP graphs
  and replayed (``CapturedCriticTerms``; ``--no-graph-capture`` keeps the eager path); the critic's
  fakes are generated and repaired on a second stream underneath the first replay of the
  real-side term (``--no-overlap`` keeps one stream);
* the graphs of a minibatch all carry the padded node count, so the generator pass, the repair, the
  penalties and the padding of the fakes take the minibatch as ONE set of graphs (``gan_amd/graphset.py``):
  a dozen launches and no host read per pass instead of that per graph (``--no-sets`` keeps the
  graph-by-graph calls; ``--unpad-templates`` implies them);
* extension flags: ``--seed``, ``--unpad-templates`` (the reference's templates stay padded,
  see gan_amd/dataset.py), ``--no-progress``, ``--no-graph-capture``, ``--autograd-gp``, ``--no-overlap``,
  ``--no-sets``;
* the gradient penalty and its parameter gradient come from a closed form
  (``AIGDiscriminator.gradient_penalty_closed_form``) unless ``--autograd-gp`` is given.
"""
from __future__ import annotations

import argparse
import contextlib
import csv
import gc
import os
import time
import traceback
from collections import OrderedDict
from typing import Dict, List, Optional

import torch

from . import parallel
from .constraints import (aig_constraint_penalties, clean_depth_set, enforce_aig_constraints, enforce_begin,
                          enforce_finish, enforce_set, pad_set, penalties_set, validate_strict_aig)
from .data import Batch, Data, GraphLoader
from .graphset import GraphSet
from .dataset import load_pt_dataset, pad_data, save_data_list_as_pt, unpad_data
from .discriminator import AIGDiscriminator, DynLevelPlan, GpRows, GraphCSR
from .generator import AIGGenerator

METRIC_FIELDS = ["step", "epoch", "D_real", "D_fake", "gap", "grad_norm", "loss_D", "loss_G", "gp", "pen"]


# the four figures of train.py:79-102: file name -> (title, y label, series); a series drawn on a
# second y axis is marked with "right"
PLOTS = {
    "losses.png": ("Losses", "loss", [("loss_D", "loss_D", "left"), ("loss_G", "loss_G", "left")]),
    "scores.png": ("Discriminator scores", "score", [("D_real", "D_real", "left"), ("D_fake", "D_fake", "left")]),
    "gap_grad.png": ("Gap & Grad-norm", None, [("gap", "gap (D_real - D_fake)", "left"), ("grad_norm", "grad_norm", "right")]),
    "gp_pen.png": ("GP & Structure penalty", "value", [("gp", "gp", "left"), ("pen", "pen", "left")]),
}


class MetricsLogger:
    """Rows of train.py:29-41 written as ``metrics.csv`` (train.py:43-50) and drawn as the four PNG
    figures of train.py:52-102 (skipped with a note when matplotlib is not installed)."""

    def __init__(self, out_dir: str):
        self.out_dir = out_dir
        os.makedirs(out_dir, exist_ok=True)
        self.rows: List[Dict[str, float]] = []

    def log(self, step, epoch, d_real, d_fake, gap, grad_norm, loss_d, loss_g, gp, pen):
        vals = [float(step), int(epoch), float(d_real), float(d_fake), float(gap), float(grad_norm), float(loss_d),
                float(loss_g), float(gp), float(pen)]
        self.rows.append(dict(zip(METRIC_FIELDS, vals)))

    def flush_csv(self):
        if not self.rows:
            return
        with open(os.path.join(self.out_dir, "metrics.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=METRIC_FIELDS)
            w.writeheader()
            w.writerows(self.rows)

    def plot_all(self) -> List[str]:
        """Writes the figures next to ``metrics.csv``; returns the paths written."""
        if not self.rows:
            return []
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except ImportError:
            print("[WARN] matplotlib is not installed: metrics.csv written, figures skipped")
            return []
        steps = [r["step"] for r in self.rows]
        written = []
        for fname, (title, ylabel, series) in PLOTS.items():
            fig, left = plt.subplots()
            right = left.twinx() if any(side == "right" for _, _, side in series) else None
            lines = []
            for key, label, side in series:
                ax = right if side == "right" else left
                lines += ax.plot(steps, [r[key] for r in self.rows], label=label, linestyle="--" if side == "right" else "-")
                if right is not None:
                    ax.set_ylabel(label)
            left.set_xlabel("step")
            if ylabel is not None:
                left.set_ylabel(ylabel)
            left.set_title(title)
            left.legend(lines, [ln.get_label() for ln in lines], loc="best")
            fig.tight_layout()
            path = os.path.join(self.out_dir, fname)
            fig.savefig(path)
            plt.close(fig)
            written.append(path)
        return written


def valid_edge_mask(d: Data) -> torch.Tensor:
    """Columns of ``edge_index`` without a negative (padding) index (train.py:126-130)."""
    if d.edge_index is None or d.edge_index.numel() == 0:
        return torch.zeros(0, dtype=torch.bool, device=d.x.device)
    return d.edge_index.min(dim=0).values >= 0


def sanitize_for_D(padded: Data) -> Data:
    """Critic input without the -1 padding edges (train.py:108-123)."""
    dev = padded.x.device
    d = Data(x=padded.x)
    if padded.edge_index is not None and padded.edge_index.numel() > 0:
        keep = valid_edge_mask(padded)
        d.edge_index = padded.edge_index[:, keep]
        ea = getattr(padded, "edge_attr", None)
        if ea is not None and ea.numel() > 0:
            d.edge_attr = ea[keep]
        else:
            d.edge_attr = torch.zeros((d.edge_index.size(1), 1), dtype=torch.long, device=dev)
    else:
        d.edge_index = torch.zeros((2, 0), dtype=torch.long, device=dev)
        d.edge_attr = torch.zeros((0, 1), dtype=torch.long, device=dev)
    d.node_depth = padded.node_depth
    return d


def mean_score(D, data: Data) -> torch.Tensor:
    """``torch.mean(D(data))`` (train.py:247-248,289): the critic returns one score per call, whose mean is
    that score - taken as a view, not as a reduction launch with its own backward launch."""
    s = D(data)
    return s.reshape(()) if s.numel() == 1 else torch.mean(s)


def gradient_penalty(D, real_pad: Data, fake_pad: Data, device, use_edges: bool = True,
                     alpha: Optional[torch.Tensor] = None, plan=None):
    """The gradient penalty on an interpolate of the two padded minibatches (train.py:133-185).

    Returns ``(gp, grad_norm)``.  ``use_edges=False`` (what the reference's loop passes, train.py:261), or no edge
    slot that holds an edge on both sides: node features only, on an edge-free interpolate (train.py:141-152,
    154-166).  Otherwise the interpolate keeps the real side's edges in the slots where both sides hold one, with
    interpolated attributes, and the norm is taken over the gradients w.r.t. features and attributes together
    (train.py:168-185) - the critic then runs out of torch operations (``GATLayer.forward_by_torch_ops``).
    ``plan``: a level plan of ``real_pad.node_depth`` the caller already holds."""
    if alpha is None:
        alpha = torch.rand(1, 1, device=device)
    x_inter = (alpha * real_pad.x + (1 - alpha) * fake_pad.x).requires_grad_(True)
    common = None
    if use_edges:
        mask_real, mask_fake = valid_edge_mask(real_pad), valid_edge_mask(fake_pad)
        if mask_real.numel() > 0 and mask_fake.numel() > 0:
            common = mask_real & mask_fake
            if int(common.sum().item()) == 0:
                common = None
    if common is None:
        inter = Data(x=x_inter, edge_index=torch.zeros((2, 0), dtype=torch.long, device=device),
                     edge_attr=torch.zeros((0, 1), dtype=torch.float32, device=device), node_depth=real_pad.node_depth)
        wrt = [x_inter]
    else:
        ea_inter = (alpha * real_pad.edge_attr.float()[common] + (1 - alpha) * fake_pad.edge_attr.float()[common]).requires_grad_(True)
        inter = Data(x=x_inter, edge_index=real_pad.edge_index[:, common], edge_attr=ea_inter, node_depth=real_pad.node_depth)
        wrt = [x_inter, ea_inter]
    if plan is not None:
        inter._plan = plan
    out = D(inter)
    grads = torch.autograd.grad(out.sum(), wrt, create_graph=True, retain_graph=True, only_inputs=True)
    flat = grads[0].reshape(-1) if len(grads) == 1 else torch.cat([g.reshape(-1) for g in grads], dim=0)
    grad_norm = torch.norm(flat, p=2)
    gp = (grad_norm - 1.0) ** 2
    return gp, grad_norm.detach()


def count_edges(edge_index: Optional[torch.Tensor]) -> torch.Tensor:
    """Edge columns that are edges: no negative (padding) endpoint and no self loop (batching turns the ``-1``
    padding of every graph but the first into self loops, see gan_amd/data.py).  A device scalar, no host read."""
    if edge_index is None or edge_index.numel() == 0:
        return torch.zeros((), dtype=torch.long, device=None if edge_index is None else edge_index.device)
    return ((edge_index.min(dim=0).values >= 0) & (edge_index[0] != edge_index[1])).sum()


def depth_top(batch: Data) -> int:
    """Deepest level of a minibatch (-1 without one).  ``GraphLoader`` notes it on the batches it makes from its
    host-side data; for any other batch it is read once (a host read when the batch is on the GPU) and kept on
    the object."""
    top = batch.__dict__.get("_depth_top")
    if top is None:
        nd = getattr(batch, "node_depth", None)
        top = -1 if nd is None or nd.numel() == 0 else int(nd.max().item())
        batch._depth_top = top
    return top


@contextlib.contextmanager
def _no_cyclic_gc():
    was_on = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_on:
            gc.enable()


class _GeneratedSide:
    """Fixed buffers for one generated minibatch (same shapes as the real one) and what the critic's recorded pass
    over it needs: edge lists and level plan rebuilt in place by ``load``."""

    def __init__(self, n: int, e_cols: int, cap: int, device: torch.device, x: torch.Tensor):
        f32 = dict(dtype=torch.float32, device=device)
        self.n, self.e_cols = n, e_cols
        self.x = x
        self.ei = torch.full((2, e_cols), -1, dtype=torch.int64, device=device)
        self.ea = torch.zeros((e_cols, 1), **f32)
        self.depth = torch.full((n,), -1, dtype=torch.int64, device=device)
        self.plan = DynLevelPlan(n, cap, device)
        self.csr = None
        self.for_D = Data(x=self.x, edge_index=self.ei, edge_attr=self.ea, node_depth=self.depth)
        self.graph = None
        self.d = None
        self.grads = ([], [])

    def load(self, batch: Batch, check: bool = True) -> bool:
        self.x.copy_(batch.x)
        self.load_edges(batch.edge_index, getattr(batch, "edge_attr", None))
        return self.load_depth(getattr(batch, "node_depth", None), check)

    def load_edges(self, edge_index: Optional[torch.Tensor], edge_attr: Optional[torch.Tensor]) -> None:
        """The edge half of ``load`` (the edge lists by target / by source rebuilt in place): all it needs are the
        minibatch's edges, which a repaired minibatch has before its level walk is through."""
        if self.e_cols:
            self.ei.copy_(edge_index)
            if edge_attr is not None and edge_attr.numel() == self.e_cols:
                self.ea.copy_(edge_attr.reshape(-1, 1))
            else:
                self.ea.zero_()
            if self.csr is None:
                self.csr = GraphCSR(self.ei, self.ea, self.n)
                self.for_D._csr = self.csr
            else:
                self.csr.rebuild(self.ei, self.ea)

    def load_depth(self, node_depth: Optional[torch.Tensor], check: bool = True) -> bool:
        """The level half of ``load``; with ``check``, one host read: does the recording hold this many levels?"""
        if node_depth is not None:
            self.depth.copy_(node_depth)
        else:
            self.depth.zero_()
        self.plan.rebuild(self.depth)
        self.for_D._plan = self.plan
        return not (check and self.plan.overflowed())


class CapturedCriticTerms:
    """``-mean(D(real))`` and ``gp_lambda * GP`` with their gradients w.r.t. the critic's parameters,
    recorded ONCE PER MINIBATCH SHAPE as HIP graphs and replayed for every minibatch of that shape.

    Every minibatch of a run is padded to ``(max_nodes, max_edges)`` (train.py:200-204), so the real batch
    always has the same tensor shapes.  Its tensors are copied into fixed buffers (``load``); what depends on
    their CONTENT is rebuilt in place, in buffers of fixed address, before the replays: the by-target /
    by-source edge lists (the builder drops the ``-1`` padding columns itself) and the level plan (its level
    count lives in device memory, see ``DynLevelPlan``).  Two graphs: the real-side term (does not depend on the
    fakes, so its first replay runs while they are still being generated) and the penalty term.

    What is recorded is made of this package's kernels and elementwise torch operations only.  The structure
    builders (device-wide sorts) and anything that reduces through torch are kept out of the recordings: a
    memset node of a recorded graph writes garbage on this ROCm release from the second replay on
    (tools/graph_memset_probe2.py), and library sorts and torch's multi-block reductions clear their scratch
    with memsets.  For the same reason the autograd form of the penalty (``--autograd-gp``) is evaluated
    eagerly every iteration; only its closed form is recorded.

    Together the two terms are about half of the ~1300 kernel launches of a training step, which is bound by
    launch rate rather than by the GPU.  ``loss_D`` is a sum, so its gradient is the replayed gradient plus the
    eagerly computed gradient of ``mean(D(fake))`` - the same numbers as one backward pass up to the order of
    one floating-point addition per parameter."""

    def __init__(self, D: AIGDiscriminator, params: List[torch.Tensor], example: Batch, level_cap: int,
                 gp_lambda: float, device: torch.device, closed_form_gp: bool = True, fake_side: bool = False,
                 fake_level_cap: int = 0, raw_side: bool = False, raw_grads: bool = True, gp_rows: bool = True):
        n = int(example.x.size(0))
        e_cols = 0 if example.edge_index is None else int(example.edge_index.size(1))
        self.n, self.e_cols, self.level_cap = n, e_cols, int(level_cap)
        f32 = dict(dtype=torch.float32, device=device)
        self.x_real = torch.zeros((n, 2), **f32)
        self.ei = torch.full((2, e_cols), -1, dtype=torch.int64, device=device)
        self.ea = torch.zeros((e_cols, 1), **f32)
        self.depth = torch.full((n,), -1, dtype=torch.int64, device=device)
        self.x_fake = torch.zeros((n, 2), **f32)
        self.alpha = torch.full((1, 1), 0.5, **f32)
        self.csr = GraphCSR(self.ei, self.ea, n) if e_cols > 0 else None
        self.plan = DynLevelPlan(n, self.level_cap, device)
        self.real_for_D = Data(x=self.x_real, edge_index=self.ei, edge_attr=self.ea, node_depth=self.depth)
        self.real_for_D._csr, self.real_for_D._plan = self.csr, self.plan
        real_pad = Data(x=self.x_real, node_depth=self.depth)
        fake_pad = Data(x=self.x_fake)
        self._params = params
        self.load(example)
        self.x_fake.copy_(self.x_real)
        # the penalty over its distinct rows (nodes of one level with equal features are one row, see GpRows): the
        # rows are rebuilt, eagerly, whenever one of the three buffers they are made of has been written since
        self.gp_rows = GpRows(n, self.level_cap, device) if (closed_form_gp and gp_rows) else None
        self._rows_key = None
        self._refresh_rows()
        # the generated side: score of a generated minibatch (the critic's repaired fakes, the generator step's
        # raw graphs - same shapes as the real minibatch) and its gradient, over buffers of its own
        self.fake_side = bool(fake_side)
        self.fake_cap = max(int(fake_level_cap), self.level_cap)
        # ... and, with ``raw_side``, a second set for the generator step's raw graphs, so that they can be made and
        # loaded while the critic iterations still replay over the first (their levels are the templates': no
        # deeper than the real minibatch's)
        self.fk = self.rw = None
        if self.fake_side:
            self.fk = _GeneratedSide(n, e_cols, self.fake_cap, device, self.x_fake)
            self.fk.load(example, check=False)
            if raw_side:
                self.rw = _GeneratedSide(n, e_cols, self.fake_cap, device, torch.zeros((n, 2), **f32))
                self.rw.load(example, check=False)

        # seeds of the backward passes as constants: autograd would make them with a fill launch per pass, and the
        # real side's sign with a negation launch forward and another one backward (the gradient of -d is the
        # gradient of d seeded with -1: the same bits)
        # (kept on the object: the recordings read them at every replay)
        self._seeds = plus_one, minus_one = torch.ones((), **f32), -torch.ones((), **f32)

        def compute_real():
            d_real = mean_score(D, self.real_for_D)
            return d_real.detach(), torch.autograd.grad(d_real, params, grad_outputs=minus_one, allow_unused=True)

        def compute_gp():
            if closed_form_gp:
                # (scaled after the fact - ``out_scale``, inside the launches that write the gradients - not through
                #  ``grad_scale``: the same roundings as lambda * autograd's gradient; Adam turns a last-bit
                #  difference in a near-zero gradient element into a full step)
                gp, gn, gp_grads = D.gradient_penalty_closed_form(self.x_real, self.x_fake, self.alpha, self.depth,
                                                                  plan=self.plan, rows=self.gp_rows, out_scale=gp_lambda)
                return gp, gn, [gp_grads.get(p) for p in params]
            gp, gn = gradient_penalty(D, real_pad, fake_pad, device, use_edges=False, alpha=self.alpha, plan=self.plan)
            return gp.detach(), gn, torch.autograd.grad(gp_lambda * gp, params, allow_unused=True)

        def compute_side(side):
            if side is self.rw and not raw_grads:     # the generator step's score alone (see Trainer.skip_unused_backward)
                with torch.no_grad():
                    return mean_score(D, side.for_D).detach(), [None] * len(params)
            d_side = mean_score(D, side.for_D)
            return d_side.detach(), torch.autograd.grad(d_side, params, grad_outputs=plus_one, allow_unused=True)

        self._compute_gp = compute_gp
        cur = torch.cuda.current_stream(device)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):      # sizes the allocator's pools before anything is recorded
            for _ in range(2):
                compute_real()
                compute_gp()
                for side_ in (self.fk, self.rw):
                    if side_ is not None:
                        compute_side(side_)
        cur.wait_stream(side)
        torch.cuda.synchronize(device)
        # thread-local capture: a communicator's watchdog thread may query events while we record.  The cyclic
        # collector is held off while a recording is open: it can run in any thread at any allocation (the
        # autograd worker's included), and a collected object that owns device resources - an earlier recording,
        # say - would release them in the middle of the capture, which the runtime answers with an abort.
        self.graph_real = torch.cuda.CUDAGraph()
        with _no_cyclic_gc(), torch.cuda.graph(self.graph_real, capture_error_mode="thread_local"):
            self.d_real, grads_real = compute_real()
        self._grads_real = self._present(grads_real)
        self.graph_gp = None
        if closed_form_gp:
            self.graph_gp = torch.cuda.CUDAGraph()
            with _no_cyclic_gc(), torch.cuda.graph(self.graph_gp, capture_error_mode="thread_local"):
                self.gp, self.grad_norm, grads_gp = compute_gp()
            self._grads_gp = self._present(grads_gp)
        for side_ in (self.fk, self.rw):
            if side_ is not None:
                side_.graph = torch.cuda.CUDAGraph()
                with _no_cyclic_gc(), torch.cuda.graph(side_.graph, capture_error_mode="thread_local"):
                    side_.d, grads_side = compute_side(side_)
                side_.grads = self._present(grads_side)

    def release(self) -> None:
        """Drop the recordings and what refers back to this object now (called when a recording is replaced),
        instead of whenever the cyclic collector gets to it."""
        self.graph_real = self.graph_gp = None
        self._compute_gp = None
        self.__dict__.pop("_fused_update", None)
        self._grads_real = self._grads_gp = ([], [])
        for side_ in (self.fk, self.rw):
            if side_ is not None:
                side_.graph, side_.grads = None, ([], [])

    def _present(self, grads):
        """(parameters, gradients) of the parameters a term reaches."""
        return ([p for p, g in zip(self._params, grads) if g is not None], [g for g in grads if g is not None])

    def fits(self, batch: Batch) -> bool:
        e_cols = 0 if batch.edge_index is None else int(batch.edge_index.size(1))
        return int(batch.x.size(0)) == self.n and e_cols == self.e_cols and depth_top(batch) <= self.level_cap

    def load(self, batch: Batch) -> None:
        """Make ``batch`` the real minibatch of the following replays: four copies into the fixed buffers, then
        its edge lists and level plan rebuilt in place (eagerly: a few dozen launches per training step)."""
        self.x_real.copy_(batch.x)
        if self.e_cols:
            self.ei.copy_(batch.edge_index)
            ea = getattr(batch, "edge_attr", None)
            if ea is not None and ea.numel() == self.e_cols:
                self.ea.copy_(ea.reshape(-1, 1))
            else:
                self.ea.zero_()
        nd = getattr(batch, "node_depth", None)
        if nd is not None:
            self.depth.copy_(nd)
        else:
            self.depth.zero_()      # no depth vector: every node in one level (discriminator.py:92-96)
        if self.csr is not None:
            self.csr.rebuild(self.ei, self.ea)
        self.plan.rebuild(self.depth)
        # the minibatch's edge count is a property of the minibatch: counted once, kept on the object
        count = batch.__dict__.get("_edge_count")
        if count is None:
            count = batch._edge_count = count_edges(batch.edge_index)
        self.edges_real = count

    # the critic's fakes (their features are also what the penalty term reads: ``x_fake``)
    plan_f = property(lambda self: self.fk.plan)
    csr_f = property(lambda self: self.fk.csr)
    fake_for_D = property(lambda self: self.fk.for_D)
    d_fake = property(lambda self: self.fk.d)

    def load_generated(self, batch: Batch, check: bool = True) -> bool:
        """Make ``batch`` (same shapes as the real minibatch) the input of the following ``replay_fake``: four
        copies, its edge lists and level plan rebuilt in place.  With ``check``, one host read tells whether its
        deepest level fits the recording (the same read the eager path spends on sizing its level plan); on
        False the caller evaluates this minibatch eagerly."""
        if not self.fake_side:
            self.x_fake.copy_(batch.x)
            return False
        return self.fk.load(batch, check)

    def replay_real(self) -> None:
        self.graph_real.replay()

    def replay_fake(self) -> None:
        self.fk.graph.replay()

    def add_fake_gradients(self, sign: float = 1.0) -> None:
        ps, gs = self.fk.grads
        torch._foreach_add_([p.grad for p in ps], gs, alpha=sign)

    def replay_raw(self) -> None:
        self.rw.graph.replay()

    def add_raw_gradients(self, sign: float = 1.0) -> None:
        ps, gs = self.rw.grads
        if ps:
            torch._foreach_add_([p.grad for p in ps], gs, alpha=sign)

    def _refresh_rows(self) -> None:
        if self.gp_rows is None:
            return
        key = (self.x_real._version, self.x_fake._version, self.depth._version)      # (every write is a torch copy)
        if key != self._rows_key:
            self.gp_rows.rebuild(self.x_real, self.x_fake, self.plan)
            self._rows_key = key

    def replay_gp(self, alpha: torch.Tensor) -> None:
        """The penalty term between the real minibatch and the generated one of the last ``load_generated``."""
        self.alpha.copy_(alpha.reshape(1, 1))
        self._refresh_rows()
        if self.graph_gp is not None:
            self.graph_gp.replay()
        else:
            self.gp, self.grad_norm, grads_gp = self._compute_gp()
            self._grads_gp = self._present(grads_gp)

    def replay(self, x_fake: torch.Tensor, alpha: torch.Tensor) -> None:
        self.x_fake.copy_(x_fake)
        self.replay_real()
        self.replay_gp(alpha)

    def add_gradients(self) -> None:
        for ps, gs in (self._grads_real, self._grads_gp):
            torch._foreach_add_([p.grad for p in ps], gs)


class FusedCriticUpdate:
    """``opt_D.zero_grad(); (the three recorded terms' gradients summed into .grad); opt_D.step()`` of one critic
    iteration (train.py:253,265-266) as ONE launch (``aig_adam_sum3``) instead of six: the gradients are the same
    sums in the same order, the update is Adam's with the optimizer's own state tensors (its step counters included),
    so iterations may go through this object or through ``opt_D.step()`` in any mix.

    Built for one ``CapturedCriticTerms`` (whose gradient buffers have fixed addresses) once the optimizer holds
    its state, i.e. after the first ordinary ``opt_D.step()``; ``valid()`` tells whether the tensors it points at are
    still the parameters' (an eager iteration replaces the ``.grad`` tensors)."""

    def __init__(self, opt: torch.optim.Adam, params: List[torch.Tensor], terms: "CapturedCriticTerms"):
        import numpy as np
        from . import _abi
        group = opt.param_groups[0]
        if len(opt.param_groups) != 1 or group.get("weight_decay", 0) != 0 or group.get("amsgrad") or group.get("maximize"):
            raise ValueError("plain Adam only")
        self.lr, (self.beta1, self.beta2), self.eps = float(group["lr"]), group["betas"], float(group["eps"])
        lib = _abi.lib()
        chunk = int(lib.aig_adam_chunk_elems())
        # the three terms in the order their gradients are added (no terms: an update from .grad as it stands)
        src = [dict(zip(ps, gs)) for ps, gs in ((terms.fk.grads, terms._grads_real, terms._grads_gp) if terms else ())]
        src += [{}] * (3 - len(src))
        recs, steps, key = [], [], []
        for p in params:
            st = opt.state.get(p, {})
            if p.grad is None or not all(k in st for k in ("exp_avg", "exp_avg_sq", "step")):
                raise ValueError("optimizer state not there yet")
            gs = [d.get(p) for d in src]
            tensors = [p, p.grad, st["exp_avg"], st["exp_avg_sq"]] + [g for g in gs if g is not None]
            if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel() for t in tensors):
                raise ValueError("contiguous float32 tensors on the device only")
            if not (st["step"].is_cuda and st["step"].dtype == torch.float32):
                raise ValueError("the optimizer's step counters must be float32 device tensors (fused Adam)")
            steps.append(st["step"].data_ptr())
            key.append((p.data_ptr(), p.grad.data_ptr()))
            for off in range(0, p.numel(), chunk):
                b = 4 * off
                recs.append((p.data_ptr() + b, p.grad.data_ptr() + b) + tuple(0 if g is None else g.data_ptr() + b for g in gs) +
                            (st["exp_avg"].data_ptr() + b, st["exp_avg_sq"].data_ptr() + b, min(chunk, p.numel() - off)))
        self._keep = [opt.state[p] for p in params]
        self.params, self.key = params, key
        dev = params[0].device
        self.table = torch.from_numpy(np.asarray(recs, dtype=np.int64).reshape(-1)).to(dev)
        self.steps = torch.tensor(steps, dtype=torch.int64, device=dev)
        self.nchunks, self.nsteps = len(recs), len(steps)

    def valid(self) -> bool:
        return all(p.grad is not None and (p.data_ptr(), p.grad.data_ptr()) == k for p, k in zip(self.params, self.key))

    def apply(self, update_no: int, mode: int = 3) -> None:
        """``update_no``: how many updates the optimizer has made, this one included.  ``mode`` 3: sums and update;
        1: the sums alone (the ranks of a data-parallel run average the gradients next); 2: the update alone."""
        from . import _abi
        n = max(int(update_no), 1)
        bc1 = 1.0 - self.beta1 ** n
        bc2 = 1.0 - self.beta2 ** n
        _abi.check(_abi.lib().aig_adam_sum3(self.nchunks, _abi.ptr(self.table), self.nsteps, _abi.ptr(self.steps),
                                            float(n), self.lr, float(self.beta1), float(self.beta2), self.eps,
                                            float(bc1), float(bc2 ** 0.5), int(mode), _abi.stream()), "aig_adam_sum3")


_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def _side_stream(device: torch.device) -> "torch.cuda.Stream":
    """The second stream of the training step, one per device for the life of the process.  Streams share a handful
    of hardware queues; a trainer made after other streams have been taken from the pool could get one that shares
    its queue with the first stream, and the step's two streams would then run one after the other."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _SIDE_STREAMS:
        _SIDE_STREAMS[idx] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[idx]


class Trainer:
    """State of one training run; ``step(batch)`` is one pass of the reference's inner loop body."""

    def __init__(self, args, max_nodes: int, max_edges: int, device: torch.device, rank: int = 0, world: int = 1):
        self.args, self.device, self.rank, self.world = args, device, rank, world
        # gradients are exchanged between ranks (or through a one-rank communicator made to, see gan_amd/parallel.py)
        self.exchange = world > 1 or parallel.exchanging()
        self.max_nodes, self.max_edges = max_nodes, max_edges
        self.G = AIGGenerator(node_in_dim=2, hidden_dim=args.g_hidden, z_dim=args.z_dim,
                              candidate_k=args.candidate_k).to(device)
        self.D = AIGDiscriminator(in_dim=2, hidden_dim=args.d_hidden).to(device)
        parallel.broadcast_parameters([self.G, self.D])
        # the same update as train.py:221-222; on a HIP device the single-kernel ("fused") form of it, one
        # launch per step instead of a dozen multi-tensor launches
        fused = {"fused": True} if device.type == "cuda" else {}
        self.opt_G = torch.optim.Adam(self.G.parameters(), lr=args.lr, betas=(0.5, 0.9), **fused)
        self.opt_D = torch.optim.Adam(self.D.parameters(), lr=args.lr, betas=(0.5, 0.9), **fused)
        self.reducer = parallel.FlatGradAllReduce(self.D.parameters())
        self._d_params = [p for p in self.D.parameters() if p.requires_grad]
        self._captured: "OrderedDict[tuple, CapturedCriticTerms]" = OrderedDict()   # by minibatch shape
        self.captures = 0     # recordings made so far (one per minibatch shape, unless a level cap was outgrown)
        self.graph_capture = bool(getattr(args, "graph_capture", True))
        self.closed_form_gp = bool(getattr(args, "closed_form_gp", True))
        self.global_step = 0
        self._templates: Dict[int, Data] = {}
        self.last: Dict[str, torch.Tensor] = {}
        self._edges_last = None   # device scalar: edges of the last step's critic inputs (real + fake)
        self.level_cap = int(getattr(args, "level_cap", 0) or 0)   # deepest level the recordings are sized for
        # random draws of the loop (train.py:143,241,276); tests substitute recorded values
        self.draw_z = lambda: torch.randn(args.z_dim, device=device)
        self.draw_alpha = lambda: torch.rand(1, 1, device=device)
        # the structural penalties only need the raw graphs and their latency-bound traversals keep
        # one CU busy: they run on a second stream underneath the critic pass of the generator step
        self.side_stream = _side_stream(device)
        # ... and the generator step's raw graphs (which depend on nothing the critic iterations compute) are
        # made, padded and loaded into their recording's buffers on a third stream underneath those iterations
        self.raw_stream = self.side_stream
        # ... and the level sweep of the critic's fakes on one more (a chain of single-workgroup kernels, see step)
        self.depth_stream = torch.cuda.Stream(device=device)
        # ... and so do the critic's fakes (generator pass and repair), underneath the recorded
        # real-side term of the first critic iteration
        self.overlap_fakes = not bool(getattr(args, "no_overlap", False))
        self.use_sets = not bool(getattr(args, "no_sets", False))
        self.capture_fake = str(getattr(args, "capture_fake", "auto"))
        # ``loss_G.backward()`` (train.py:298) reaches the critic's parameters only - the generator's outputs are
        # integers and copies - and the critic's gradients are zeroed before their next use (train.py:253): that
        # backward pass changes nothing a run can observe.  Off by default (the step then does what the reference's
        # does); with it on, the generator step is the critic's forward pass over the raw graphs alone.
        self.skip_unused_backward = bool(getattr(args, "skip_unused_backward", False))
        # one launch per critic iteration for zero_grad + the three gradient sums + Adam (FusedCriticUpdate) whenever
        # all three terms of the iteration are recorded and no gradient exchange stands between them and the update
        self.fused_update = not bool(getattr(args, "no_fused_update", False))
        # the recorded penalty term over its distinct rows (same numbers up to the order of the sums)
        self.gp_rows = not bool(getattr(args, "gp_all_nodes", False))
        self._d_updates = 0      # updates opt_D has made (its step counters hold the same number)
        self._fake_caps: Dict[tuple, int] = {}   # by minibatch shape: level cap the generated side was found to need
        self.marks: Optional[list] = None   # tuning aid: [(label, event on the main stream, host time)] of the last step
        self.probes: Optional[dict] = None  # diagnosis aid: {label: event on whatever stream was current} of the last step

    def _probe(self, label: str) -> None:
        if self.probes is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.probes[label] = ev

    def _mark(self, label: str) -> None:
        if self.marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((label, ev, time.perf_counter()))

    # templates are the same tensors every epoch: keep them (and their generator plan) per sample
    def templates_of(self, real_padded: Batch) -> List[Data]:
        ids = getattr(real_padded, "sample_ids", None)
        parts = None
        out = []
        for k in range(real_padded.num_graphs):
            key = ids[k] if ids is not None else None
            if key is None or key not in self._templates:
                if parts is None:
                    parts = real_padded.to_data_list()
                d = parts[k]
                if getattr(self.args, "unpad_templates", False):
                    keep = int((d.node_depth >= 0).sum().item())
                    d._orig_n, d._orig_e = keep, int(valid_edge_mask(d).sum().item())
                t = unpad_data(d)   # without _orig_n the template keeps its padded size, as in the reference
                t = Data(x=t.x.clone(), node_depth=None if t.node_depth is None else t.node_depth.clone())
                if key is None:
                    out.append(t)
                    continue
                self._templates[key] = t
            out.append(self._templates[key])
        return out

    def _captured_terms(self, real_padded: Batch) -> Optional[CapturedCriticTerms]:
        """The recorded real-side / gradient-penalty terms for minibatches of this shape (None when graph
        capture is off).  A recording is sized for depth levels up to a cap; a minibatch deeper than that
        replaces it by one with twice the room."""
        if not self.graph_capture:
            return None
        e_cols = 0 if real_padded.edge_index is None else int(real_padded.edge_index.size(1))
        key = (int(real_padded.x.size(0)), e_cols)
        terms = self._captured.get(key)
        if terms is not None and terms.fits(real_padded):
            self._captured.move_to_end(key)
            return terms
        top = max(depth_top(real_padded), 0)
        cap = max(self.level_cap, 2 * top, 64)
        old = self._captured.pop(key, None)
        if old is not None:
            old.release()
        # the generated side is recorded as well: small minibatches are bound by launch rate (an eager critic pass
        # over them costs the host more than the GPU), and on large ones the ~250 small launches of an eager pass
        # still leave gaps between the wide kernels (8.7 -> 8.0 ms/step at 100k nodes)
        fake_side = self.capture_fake in ("on", "auto")
        terms = CapturedCriticTerms(self.D, self._d_params, real_padded, cap, self.args.gp_lambda, self.device,
                                    closed_form_gp=self.closed_form_gp, fake_side=fake_side,
                                    fake_level_cap=self._fake_caps.get(key, 0),
                                    raw_side=fake_side and self.overlap_fakes and self.use_sets,
                                    raw_grads=not self.skip_unused_backward, gp_rows=self.gp_rows)
        self.captures += 1
        self._captured[key] = terms
        if len(self._captured) > 4:       # every entry pins the activations of one minibatch shape
            self._captured.popitem(last=False)[1].release()
        return terms

    def _outgrown(self, terms: CapturedCriticTerms, generated: Batch) -> None:
        """The generated minibatch is deeper than the recording of the generated side holds: this step scores it
        eagerly, the next one records anew with twice the room."""
        key = (terms.n, terms.e_cols)
        self._fake_caps[key] = 2 * int(generated.node_depth.max().item())
        self._captured.pop(key, None)

    def _fused_for(self, terms: CapturedCriticTerms) -> Optional[FusedCriticUpdate]:
        """The one-launch update for an iteration whose three terms are all recorded in ``terms`` (None: take the
        separate launches - a generated side that is not recorded, optimizer state that does not exist before the
        first ordinary step).  With more than one rank the sums and the update are two launches around the
        gradient exchange."""
        if not self.fused_update or terms.fk is None or terms.graph_gp is None:
            return None
        fused = terms.__dict__.get("_fused_update")
        if fused is not None and fused.valid():
            return fused
        try:
            fused = FusedCriticUpdate(self.opt_D, self._d_params, terms)
        except ValueError:
            return None
        terms._fused_update = fused
        return fused

    def _dp_update(self) -> FusedCriticUpdate:
        """The Adam half alone, from the averaged gradients in ``.grad`` (slices of the exchange buffer, so their
        addresses never change); the optimizer's state is created here when no update has been made yet."""
        upd = self.__dict__.get("_dp_adam")
        if upd is None or not upd.valid():
            for p in self._d_params:
                st = self.opt_D.state[p]
                if "exp_avg" not in st:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            upd = self._dp_adam = FusedCriticUpdate(self.opt_D, self._d_params, None)
        return upd

    @property
    def edges_last_step(self) -> int:
        """Edges the critic saw in the last step: real + fake, padding columns and self loops excluded (one host read)."""
        return 0 if self._edges_last is None else int(self._edges_last.item())

    def _pad_batch(self, graphs: List[Data]) -> Batch:
        return Batch.from_data_list([pad_data(g, self.max_nodes, self.max_edges) for g in graphs]).to(self.device)

    def _pad_set(self, gset: GraphSet, edges=None) -> Batch:
        """``_pad_batch(gset.graphs())`` in one launch (the graphs already have ``max_nodes`` nodes each);
        ``edges``: the padded edge tensors when the caller has made them already."""
        ei, ea = pad_set(gset, self.max_edges) if edges is None else edges
        nd = gset.node_depth
        if nd is None:
            nd = torch.zeros(gset.B * gset.n, dtype=torch.long, device=self.device)    # pad_data: depth 0 when absent
        elif nd.numel() == gset.n and gset.B > 1:
            nd = nd.repeat(gset.B)
        x = gset.x if gset.x.dtype == gset.x_dtype else gset.x.to(gset.x_dtype)
        return Batch.from_uniform(gset.B, gset.n, self.max_edges, x=x, edge_index=ei, edge_attr=ea,
                                  node_depth=nd.to(torch.long))

    def _as_set(self, templates: List[Data]) -> bool:
        """The graphs of the minibatch go through the generator and the constraint kernels as one set when
        they all have the padded node count (the reference's default: templates keep their padding)."""
        return self.use_sets and all(int(t.x.size(0)) == self.max_nodes for t in templates)

    def step(self, real_padded: Batch, epoch: int = 0, save_dir: Optional[str] = None) -> Dict[str, float]:
        a, dev, G, D = self.args, self.device, self.G, self.D
        real_padded = real_padded.to(dev)
        templates = self.templates_of(real_padded)

        # ---- critic: fakes are generated once and reused for all n_critic iterations (train.py:241-245)
        # A rank whose step raises still enters every gradient all-reduce of the step (with a
        # "not valid" flag), then re-raises: the other ranks never wait on a missing partner.
        failed: Optional[BaseException] = None
        main_stream = torch.cuda.current_stream(dev)
        self._mark("start")
        self._probe("main: step start")
        real_replayed = False
        try:
            terms = self._captured_terms(real_padded)
            fakes_stream = self.side_stream if (terms is not None and self.overlap_fakes) else main_stream
            fakes_stream.wait_stream(main_stream)
            # With the fakes on their own stream they are enqueued FIRST: the real minibatch's structures (two
            # sorts for its edge lists, one for its levels - dozens of tiny launches) are then rebuilt underneath
            # the generator pass instead of in front of everything.
            real_first = terms is None or fakes_stream is main_stream
            if terms is None:
                real_for_D, edges_real = sanitize_for_D(real_padded), count_edges(real_padded.edge_index)
            elif real_first:
                terms.load(real_padded)
            as_set = self._as_set(templates)
            # The penalty term needs the fakes' node features, the critic's pass over the fakes their edges and
            # levels.  The levels come out of a level-by-level walk (one workgroup per graph, most of a millisecond
            # on a large graph): when the fakes are made on their own stream, the repair hands over features and
            # edges first and the walk follows, so that the main stream can go on to the penalty term meanwhile.
            split = as_set and fakes_stream is not main_stream
            deferred_wait = False
            with torch.cuda.stream(fakes_stream):
                zs_once = [self.draw_z() for _ in templates]
                if split:
                    raw_fakes_set = G.forward_set(templates, torch.stack(zs_once))
                    raw_ready = torch.cuda.Event()
                    raw_ready.record()
                    self._probe("side: fakes generated")
            if split and not real_first:
                # The host enqueues the real side right behind the generator pass (half a millisecond of GPU
                # work) and the repair's ~150 small launches after that: enqueued in front of the real side they
                # kept the host busy for longer than the generator pass runs, and the first stream idle meanwhile.
                terms.load(real_padded)
                terms.replay_real()       # the first iteration's real-side term runs next to the fakes
                real_replayed = True
            finish_fakes = None
            with torch.cuda.stream(fakes_stream):
                if split:
                    corr_set = enforce_set(raw_fakes_set, want_depth=False)
                    features_ready = torch.cuda.Event()
                    features_ready.record()
                    self._probe("side: fakes repaired")
            # The generated graphs' levels only need the raw graphs (breadth-first levels over their cleaned edges):
            # three sorts and one workgroup that sweeps for a quarter of a millisecond.  They run on a stream of their
            # own from the moment the raw graphs exist - next to the repair on the second stream and the real-side
            # replay on the first (a chain of single-workgroup kernels: it takes nothing from either) - so that the
            # first stream goes from the real side straight to the penalty term and the second from the repair straight
            # to the padded edges and edge lists.
            depth_early = depth_ready = None
            if split and real_replayed and terms.fake_side:
                ds = self.depth_stream
                ds.wait_event(raw_ready)
                with torch.cuda.stream(ds):
                    depth_early = clean_depth_set(raw_fakes_set)
                    depth_ready = torch.cuda.Event()
                    depth_ready.record(ds)
                for t_ in (raw_fakes_set.x, raw_fakes_set.edge_index, raw_fakes_set.count):
                    if t_ is not None:
                        t_.record_stream(ds)         # (allocated under the second stream, read by this one)
                self._probe("main: generated graphs' levels")
            with torch.cuda.stream(fakes_stream):
                if split:

                    def finish_fakes():
                        # the level walk behind the fakes' depths and the padded minibatch: enqueued by the first
                        # critic iteration AFTER its penalty term (host order: the penalty term only waits for
                        # the features, and the walk's set-up is dozens of small launches)
                        with torch.cuda.stream(fakes_stream):
                            edges = pad_set(corr_set, self.max_edges)
                            ready = torch.cuda.Event()
                            ready.record()        # the critic's edge lists of the fakes can be built from here on
                            # ... and with a recording of the generated side they are, right here: the two dozen
                            # small launches behind its edge lists and level plan run on this stream, underneath
                            # the first stream's real-side and penalty terms, instead of in front of its replay
                            if terms.fake_side:
                                terms.fk.load_edges(edges[0], edges[1])
                            if depth_early is not None:
                                # The padded minibatch and the generated side's level plan follow the penalty term on
                                # the FIRST stream (which has the time: the second still builds the edge lists when
                                # the levels arrive); the recorded pass over the generated side, on the second stream,
                                # waits for them.
                                with torch.cuda.stream(main_stream):
                                    main_stream.wait_event(ready)
                                    main_stream.wait_event(depth_ready)
                                    for t_ in edges:
                                        if torch.is_tensor(t_):
                                            t_.record_stream(main_stream)
                                    depth_early.record_stream(main_stream)
                                    corr_set.node_depth = depth_early
                                    padded = self._pad_set(corr_set, edges)
                                    terms.fk.load_depth(padded.node_depth, check=False)
                                    loaded = torch.cuda.Event()
                                    loaded.record(main_stream)
                                    n_edges = count_edges(padded.edge_index)
                                fakes_stream.wait_event(loaded)
                                self._probe("side: fakes' levels and padding")
                                return padded, n_edges, ready
                            corr_set.node_depth = clean_depth_set(raw_fakes_set)
                            padded = self._pad_set(corr_set, edges)
                            if terms.fake_side:
                                terms.fk.load_depth(padded.node_depth, check=False)
                            self._probe("side: fakes' levels and padding")
                            return padded, count_edges(padded.edge_index), ready
                    fake_padded = None
                elif as_set:
                    # generator, repair and padding for the whole minibatch: a dozen launches, no host read
                    fake_padded = self._pad_set(enforce_set(G.forward_set(templates, torch.stack(zs_once))))
                else:
                    raw_fakes_once = [G(t, z) for t, z in zip(templates, zs_once)]
                    # every repair is enqueued before the first edge count is awaited: one wait per minibatch
                    repairs = [enforce_begin(r) for r in raw_fakes_once]
                    corr_fakes_once = [enforce_finish(h) for h in repairs]
                    fake_padded = self._pad_batch(corr_fakes_once)
                fake_for_D = fake_padded
                edges_fake = count_edges(fake_padded.edge_index) if fake_padded is not None else None
            if not real_first and not real_replayed:
                terms.load(real_padded)
                terms.replay_real()       # the first iteration's real-side term runs next to the fakes
                real_replayed = True
            if terms is not None:
                real_for_D, edges_real = terms.real_for_D, terms.edges_real
            fake_recorded = False
            fake_pending = False     # the generated side is recorded, its inputs are loaded once the walk is through
            if split:
                main_stream.wait_event(features_ready)
                terms.x_fake.copy_(corr_set.x)      # all the penalty term reads of the fakes
                deferred_wait = True                # the rest is awaited right before the critic's pass over them
                fake_pending = terms.fake_side
            else:
                main_stream.wait_stream(fakes_stream)
                self._mark("fakes ready")
            if terms is not None:
                if not split:
                    fake_recorded = terms.load_generated(fake_padded)
                    if terms.fake_side and not fake_recorded:
                        self._outgrown(terms, fake_padded)
                if fake_recorded or fake_pending:
                    for p_ in self._d_params:      # all three terms are recorded: their gradients are added
                        if p_.grad is None:        # into .grad tensors that are zeroed in place
                            p_.grad = torch.zeros_like(p_)
        except Exception as e:
            if not self.exchange:
                raise
            failed = e
        # The generator step's inputs early: with a recording of their own for the raw graphs (``terms.rw``), the
        # generator pass, the padded minibatch, its edge lists and level plan and the structural penalties are
        # enqueued on a third stream right after the first critic iteration and run underneath the other
        # iterations; the generator step is then one replay.  The draws keep the order of train.py:241-276
        # (z of the fakes, one alpha per critic iteration, z of the raw graphs).
        early_raw = (failed is None and terms is not None and terms.rw is not None and as_set
                     and terms is self._captured.get((terms.n, terms.e_cols)))
        alphas = zs = None
        raw_set = loss_pen = raw_loaded = None
        if early_raw:
            alphas = [self.draw_alpha() for _ in range(a.n_critic)]
            zs = [self.draw_z() for _ in templates]
            step_start = torch.cuda.Event()
            step_start.record(main_stream)       # (what the third stream writes was last read before this point)
        raw_from = min(1, a.n_critic - 1)    # from this iteration on the second stream belongs to the raw graphs
        # the logged scalars in one launch at the end of the step (aig_step_metrics) when every number they are
        # made of sits in a buffer of the recordings until then
        metrics_late = False
        pen_all = None
        for it in range(a.n_critic):
            updated = False      # this iteration's update has been made (by the one-launch form)
            if failed is None:
                try:
                    # (an eager backward pass over the fakes installs its gradients as the .grad tensors - no
                    #  kernel - and the recorded terms are added to them; zeroed tensors would cost it an
                    #  accumulation launch per parameter)
                    if terms is None:
                        self.opt_D.zero_grad(set_to_none=True)
                    # (with recorded terms the gradients are zeroed further down, right before their first use:
                    #  when all three terms are recorded, one launch overwrites them instead)
                    if terms is not None:
                        # (once the raw graphs' pipeline occupies the second stream, the recorded pass over the
                        #  fakes follows the other two terms on this one)
                        raw_turn = early_raw and it >= raw_from
                        fake_on_side = fake_recorded and not fake_pending and not raw_turn
                        fake_after = fake_recorded and not fake_pending and raw_turn
                        if fake_on_side:      # the three recorded terms are independent: the generated side
                            self.side_stream.wait_stream(main_stream)          # runs next to the other two
                            with torch.cuda.stream(self.side_stream):
                                terms.replay_fake()
                        if not (it == 0 and real_replayed):
                            terms.replay_real()
                        terms.replay_gp(alphas[it] if alphas is not None else self.draw_alpha())
                        self._probe(f"main: real and penalty terms {it}")
                        edges_ready = None
                        if finish_fakes is not None:
                            fake_padded, edges_fake, edges_ready = finish_fakes()
                            fake_for_D, finish_fakes = fake_padded, None
                        if deferred_wait and not fake_pending:
                            # no recording of the generated side: the eager pass below reads the padded minibatch
                            # on this stream
                            main_stream.wait_stream(fakes_stream)
                            deferred_wait = False
                            self._mark("fakes ready")
                        if fake_pending and edges_ready is not None and depth_early is not None:
                            # the generated side's structures were loaded on the second stream: its recorded pass
                            # follows them there, next to this stream's penalty term (one host read on that stream:
                            # does the recording hold their depth?)
                            fake_pending = False
                            with torch.cuda.stream(fakes_stream):
                                fake_recorded = not terms.fk.plan.overflowed()
                                if fake_recorded:
                                    terms.replay_fake()
                            main_stream.wait_stream(fakes_stream)
                            deferred_wait = False
                            self._mark("fakes ready")
                            if not fake_recorded:
                                self._outgrown(terms, fake_padded)
                            fake_here = fake_recorded
                        elif fake_pending:
                            if deferred_wait:
                                main_stream.wait_stream(fakes_stream)
                                deferred_wait = False
                                self._mark("fakes ready")
                            # first iteration with the fakes' level walk behind the two replays above: their
                            # level plan is rebuilt now (one host read: does the recording hold their depth?)
                            # and the recorded pass over them follows on this stream
                            fake_pending = False
                            if edges_ready is not None:      # (loaded on the second stream, see finish_fakes)
                                fake_recorded = not terms.fk.plan.overflowed()
                            else:
                                fake_recorded = terms.load_generated(fake_padded)
                            if fake_recorded:
                                terms.replay_fake()
                            else:
                                self._outgrown(terms, fake_padded)
                            fake_here = fake_recorded
                        elif fake_after:
                            terms.replay_fake()
                            self._probe(f"main: generated side replayed {it}")
                            fake_here = True
                        else:
                            fake_here = False
                        if fake_on_side:
                            main_stream.wait_stream(self.side_stream)
                        if fake_on_side or fake_here:
                            if it == a.n_critic - 1:
                                metrics_late = early_raw and self.fused_update
                                if not metrics_late:      # (the buffer is written again by the generator step)
                                    D_fake_mean = terms.d_fake.clone()
                            fused = self._fused_for(terms)
                            if fused is not None and not self.exchange:
                                fused.apply(self._d_updates + 1)   # zero_grad + the three sums + opt_D.step()
                                self._d_updates += 1               # (counted once the launch is in the queue)
                                updated = True
                            elif fused is not None:
                                fused.apply(0, mode=1)             # zero_grad + the three sums; the exchange is next
                            else:
                                self.opt_D.zero_grad(set_to_none=False)
                                terms.add_fake_gradients()
                                terms.add_gradients()
                            self._probe(f"main: generated side's gradients added {it}")
                        else:
                            self.opt_D.zero_grad(set_to_none=True)
                            D_fake_mean = mean_score(D, fake_for_D)
                            D_fake_mean.backward()
                            terms.add_gradients()
                        self._probe(f"main: critic gradients {it}")
                        if it == a.n_critic - 1 and not metrics_late:   # the logged scalars are those of the last iteration (train.py:263-272)
                            D_real_mean, gp_term, grad_norm = terms.d_real.clone(), terms.gp.clone(), terms.grad_norm.clone()
                            gap = D_real_mean - D_fake_mean.detach()
                            loss_D = (D_fake_mean.detach() - D_real_mean) + a.gp_lambda * gp_term
                    else:
                        D_real_mean = mean_score(D, real_for_D)
                        D_fake_mean = mean_score(D, fake_for_D)
                        gap = D_real_mean - D_fake_mean
                        if self.closed_form_gp:
                            gp_term, grad_norm, gp_grads = D.gradient_penalty_closed_form(
                                real_padded.x, fake_padded.x, self.draw_alpha(), real_padded.node_depth,
                                rows=True if self.gp_rows else None)
                            (D_fake_mean - D_real_mean).backward()
                            for p_, g_ in gp_grads.items():
                                p_.grad = a.gp_lambda * g_ if p_.grad is None else p_.grad.add_(g_, alpha=a.gp_lambda)
                            loss_D = (D_fake_mean - D_real_mean).detach() + a.gp_lambda * gp_term
                        else:
                            gp_term, grad_norm = gradient_penalty(D, real_padded, fake_padded, dev, use_edges=False,
                                                                  alpha=self.draw_alpha())
                            loss_D = (D_fake_mean - D_real_mean) + a.gp_lambda * gp_term
                            loss_D.backward()
                    if it == a.n_critic - 1 and not metrics_late:
                        self.last = dict(D_real=D_real_mean.detach(), D_fake=D_fake_mean.detach(), gap=gap.detach(),
                                         gp=gp_term.detach(), grad_norm=grad_norm.detach(), loss_D=loss_D.detach())
                except Exception as e:
                    if not self.exchange:
                        raise
                    failed = e
            if failed is None and updated:
                continue
            if self.reducer.reduce(failed is None):
                if self.exchange and self.fused_update:
                    # every rank, every iteration through the same kernel: the ranks' weights stay bit-identical
                    self._dp_update().apply(self._d_updates + 1, mode=2)
                else:
                    self.opt_D.step()
                self._d_updates += 1      # (after the update is enqueued: a step that raises before leaves the count alone)
        if failed is None and early_raw:
            # Host order: behind the replays of every critic iteration (cheap to enqueue), so the first stream has
            # its work queued while the host issues the ~100 small launches of the raw graphs' pipeline; on the
            # GPU the second stream has been free since iteration ``raw_from`` and starts on them at once.
            try:
                self.raw_stream.wait_event(step_start)
                with torch.cuda.stream(self.raw_stream):
                    raw_set = G.forward_set(templates, torch.stack(zs))
                    self._probe("side: raw graphs generated")
                    terms.rw.load(self._pad_set(raw_set), check=False)
                    raw_loaded = torch.cuda.Event()
                    raw_loaded.record()
                    self._probe("side: raw graphs loaded")
                    pen_all = penalties_set(raw_set)
                    loss_pen = pen_all if metrics_late else pen_all[:, 8].mean()
                    self._probe("side: penalties")
            except Exception as e:
                if not self.exchange:
                    raise
                failed = e
        if failed is not None:
            raise failed
        self._edges_last = edges_fake + edges_real

        # ---- generator (train.py:275-299); its output carries no gradient, so backward only touches D
        self._mark("critic done")
        self.opt_G.zero_grad()
        early_raw = early_raw and raw_set is not None
        pen_stream = self.raw_stream if early_raw else self.side_stream
        if early_raw:
            raw_gen = None
        else:
            zs = [self.draw_z() for _ in templates]
            if as_set:
                raw_set = G.forward_set(templates, torch.stack(zs))
                raw_gen = None        # the separate graphs are only made when a save step asks for them
            else:
                raw_gen = [G(t, z) for t, z in zip(templates, zs)]
            self.side_stream.wait_stream(main_stream)
            with torch.cuda.stream(self.side_stream):
                if as_set:
                    loss_pen = penalties_set(raw_set)[:, 8].mean()
                else:
                    pen_terms = [aig_constraint_penalties(r) for r in raw_gen]
                    loss_pen = torch.stack([pt["total"] for pt in pen_terms]).mean()
            raw_batch_for_D = self._pad_set(raw_set) if as_set else self._pad_batch(raw_gen)
        # the generator hands the template's depths through unchanged (generator.py:211-217) and the templates
        # are the real minibatch's graphs, so the padded depth vector of the raw batch is the real batch's: no
        # deeper than the recordings hold, and the eager pass can reuse the level plan rebuilt for this step
        raw_recorded = terms is not None and terms.fake_side and terms is self._captured.get((terms.n, terms.e_cols))
        if early_raw:
            main_stream.wait_event(raw_loaded)
            terms.replay_raw()
            loss_G_adv = None if metrics_late else -terms.rw.d.clone()
            self._probe("main: raw graphs scored")
        elif raw_recorded and not self.skip_unused_backward:
            terms.load_generated(raw_batch_for_D, check=False)
            terms.replay_fake()
            loss_G_adv = -terms.d_fake.clone()
        elif self.skip_unused_backward:
            raw_recorded = False
            if terms is not None:
                raw_batch_for_D._plan = terms.plan
            with torch.no_grad():
                loss_G_adv = -mean_score(D, raw_batch_for_D)
        else:
            if terms is not None:
                raw_batch_for_D._plan = terms.plan
            loss_G_adv = -mean_score(D, raw_batch_for_D)
            # loss_G.backward() before the penalties are awaited: they carry no gradient (the generator's outputs
            # are integers and copies), so the gradient of loss_G is the gradient of its adversarial term, and
            # the penalties' level walks run underneath the backward pass as well
            loss_G_adv.backward()
        main_stream.wait_stream(pen_stream)
        loss_pen.record_stream(main_stream)
        pen_scale = None
        if (self.global_step % a.save_every) == 0:
            if raw_gen is None:
                raw_gen = raw_set.graphs()
            scales = [10.0 if len(validate_strict_aig(r, check_all_on_path=True)) > 0 else 1.0 for r in raw_gen]
            pen_scale = torch.tensor(scales, dtype=torch.float32).mean()
            if not metrics_late:
                loss_pen = loss_pen * pen_scale.to(dev)
        if metrics_late:
            from . import _abi
            out8 = torch.empty(8, dtype=torch.float32, device=dev)
            _abi.check(_abi.lib().aig_step_metrics(int(pen_all.size(0)), _abi.ptr(terms.d_real), _abi.ptr(terms.d_fake),
                                                   _abi.ptr(terms.gp), _abi.ptr(terms.grad_norm), _abi.ptr(terms.rw.d),
                                                   _abi.ptr(pen_all), float(a.gp_lambda), float(a.lambda_cons),
                                                   1.0 if pen_scale is None else float(pen_scale), _abi.ptr(out8),
                                                   _abi.stream()), "aig_step_metrics")
            self.last = dict(zip(("D_real", "D_fake", "gap", "gp", "grad_norm", "loss_D"), out8[:6].unbind()))
            loss_G, loss_pen = out8[6], out8[7]
        else:
            loss_G = loss_G_adv.detach() + a.lambda_cons * loss_pen
        if early_raw:
            # loss_G.backward(): only the critic's parameters receive a gradient (train.py:297-299) - the
            # recorded gradient of D(raw), negated
            terms.add_raw_gradients(sign=-1.0)
        elif raw_recorded:
            terms.add_fake_gradients(sign=-1.0)
        self.opt_G.step()
        self._probe("main: step end")
        self.last.update(loss_G=loss_G.detach(), pen=loss_pen.detach())
        self._mark("generator done")

        if save_dir is not None and (self.global_step % a.save_every) == 0:
            self._save_samples(raw_gen if raw_gen is not None else raw_set.graphs(), epoch, save_dir)
        self.global_step += 1
        return self.last

    def _save_samples(self, raw_gen: List[Data], epoch: int, out_dir: str) -> None:
        a, step = self.args, self.global_step
        k = min(len(raw_gen), a.save_num)
        # the reference repairs the very objects it then writes as "raw" (train.py:324-327)
        corr_list = [enforce_aig_constraints(r) for r in raw_gen[:k]]
        tag = f"ep{epoch}_st{step}" if self.world == 1 else f"ep{epoch}_st{step}_r{self.rank}"
        save_data_list_as_pt(corr_list, out_dir, prefix="raw_" + tag, start_idx=0)
        save_data_list_as_pt(corr_list, out_dir, prefix="corr_" + tag, start_idx=0)
        valid = []
        for c in corr_list:
            violations = validate_strict_aig(c, check_all_on_path=True)
            has_edges = c.edge_index is not None and c.edge_index.numel() > 0
            and_count = int((c.x[:, 0] == 2).sum().item()) if has_edges else 0
            max_lev = int(c.node_depth.max().item()) if c.node_depth is not None and c.node_depth.numel() > 0 else 0
            if len(violations) == 0 and and_count > 0 and max_lev > 0:
                valid.append(c)
        if valid:
            save_data_list_as_pt(valid, os.path.join("results", "ISCAS85", "pt_fake"), prefix="valid_" + tag)


def train(args) -> None:
    rank, local_rank, world = parallel.init_distributed()
    if not torch.cuda.is_available():
        raise RuntimeError("gan_amd.train needs a HIP device: the models have no CPU path")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    os.makedirs(args.out_dir, exist_ok=True)
    if args.seed is not None:
        torch.manual_seed(args.seed)
        torch.cuda.manual_seed_all(args.seed)

    data_list = load_pt_dataset(args.dataset_dir, max_files=args.max_files, allow_pickle=getattr(args, "allow_pickle", False))
    if len(data_list) == 0:
        raise RuntimeError("No .pt files found in dataset_dir")
    max_nodes = max(d.x.size(0) for d in data_list) if args.max_nodes is None else args.max_nodes
    max_edges = (max((d.edge_index.size(1) if d.edge_index is not None else 0) for d in data_list)
                 if args.max_edges is None else args.max_edges)
    if rank == 0:
        print(f"[INFO] dataset size {len(data_list)}, max_nodes={max_nodes}, max_edges={max_edges}")
    padded_list = [pad_data(d, max_nodes=max_nodes, max_edges=max_edges) for d in data_list]
    loader = GraphLoader(padded_list, batch_size=args.batch_size, shuffle=True,
                         seed=args.seed if world == 1 else (args.seed or 0), rank=rank, world_size=world)
    # size the recorded critic terms for the deepest graph of the dataset, so that no minibatch outgrows them
    args.level_cap = max([0] + loader.depth_tops)
    trainer = Trainer(args, max_nodes, max_edges, device, rank, world)
    logger = MetricsLogger(args.out_dir)

    for epoch in range(args.epochs):
        it = loader
        if not args.no_progress and rank == 0:
            try:
                from tqdm import tqdm
                it = tqdm(loader, desc=f"Epoch {epoch}", dynamic_ncols=True)
            except ImportError:
                pass
        for real_padded in it:
            try:
                step_no = trainer.global_step
                m = trainer.step(real_padded, epoch, save_dir=args.out_dir)
                logger.log(step_no, epoch, d_real=m["D_real"].item(), d_fake=m["D_fake"].item(), gap=m["gap"].item(),
                           grad_norm=m["grad_norm"].item(), loss_d=m["loss_D"].item(), loss_g=m["loss_G"].item(),
                           gp=m["gp"].item(), pen=m["pen"].item())
                if step_no % args.log_every == 0 and hasattr(it, "set_postfix"):
                    r = logger.rows[-1]
                    it.set_postfix({k: f"{r[k]:.4f}" for k in ("loss_D", "loss_G", "D_real", "D_fake", "gp", "pen")})
            except Exception as e:  # the reference reports and moves on (train.py:349-352)
                print("[ERROR] exception in training step:", e)
                traceback.print_exc()
                continue
        if rank == 0:
            torch.save(trainer.G.state_dict(), os.path.join(args.out_dir, f"generator_epoch{epoch}.pt"))
            torch.save(trainer.D.state_dict(), os.path.join(args.out_dir, f"discriminator_epoch{epoch}.pt"))
            logger.flush_csv()
            logger.plot_all()
            print(f"[INFO] saved checkpoints & metrics epoch {epoch}")
        if epoch == 0:
            # Everything alive after the first epoch lives as long as the run: the dataset, the per-sample
            # templates and generator plans, the recorded critic terms.  They are taken out of the cyclic collector's
            # sight; otherwise a full collection walks a few hundred thousand objects - a stall of ~100 ms in
            # the middle of a step - whenever enough short-lived containers have come and gone.
            gc.collect()
            gc.freeze()
    if rank == 0:
        print("[INFO] training finished")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser()
    p.add_argument("--dataset-dir", type=str, required=True, help="Directory containing the dataset.")
    p.add_argument("--out-dir", type=str, default="out_gen", help="Output directory.")
    p.add_argument("--epochs", type=int, default=10)
    p.add_argument("--batch-size", type=int, default=1)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--g-hidden", type=int, default=128)
    p.add_argument("--d-hidden", type=int, default=128)
    p.add_argument("--z-dim", type=int, default=128)
    p.add_argument("--candidate-k", type=int, default=512, help="Number of candidate predecessors per node.")
    p.add_argument("--max-nodes", type=int, default=None)
    p.add_argument("--max-edges", type=int, default=None)
    p.add_argument("--lambda-cons", type=float, default=1e-2)
    p.add_argument("--gp-lambda", type=float, default=10.0)
    p.add_argument("--n-critic", type=int, default=3)
    p.add_argument("--save-every", type=int, default=10)
    p.add_argument("--save-num", type=int, default=2)
    p.add_argument("--log-every", type=int, default=1)
    p.add_argument("--max-files", type=int, default=None)
    # extensions
    p.add_argument("--seed", type=int, default=None, help="Seed torch's generators (the reference sets none).")
    p.add_argument("--unpad-templates", action="store_true",
                   help="Strip padding rows from templates (the reference keeps them padded).")
    p.add_argument("--no-progress", action="store_true")
    p.add_argument("--allow-pickle", action="store_true",
                   help="also read dataset files that need a full unpickle (pickled torch_geometric objects); trusted files only")
    p.add_argument("--autograd-gp", dest="closed_form_gp", action="store_false",
                   help="compute the gradient penalty by double backward (autograd) instead of its closed form")
    p.add_argument("--no-overlap", dest="no_overlap", action="store_true",
                   help="generate the critic's fakes on the main stream instead of underneath the real-side term")
    p.add_argument("--capture-fake", choices=["auto", "on", "off"], default="auto",
                   help="also record the critic's pass over generated minibatches (fakes, raw graphs) as a HIP graph "
                        "(auto = on; off evaluates it eagerly every time)")
    p.add_argument("--no-fused-update", dest="no_fused_update", action="store_true",
                   help="zero the critic's gradients, add the recorded terms' gradients and step its optimizer as separate "
                        "launches instead of one")
    p.add_argument("--gp-all-nodes", dest="gp_all_nodes", action="store_true",
                   help="run the recorded gradient-penalty term over every node instead of over the distinct rows "
                        "(nodes of one level with equal features are one row with a multiplicity)")
    p.add_argument("--skip-unused-backward", dest="skip_unused_backward", action="store_true",
                   help="leave out the backward pass of the generator step: it only reaches the critic's gradients, which "
                        "are zeroed before their next use (weights, metrics and samples are unchanged)")
    p.add_argument("--no-sets", dest="no_sets", action="store_true",
                   help="run generator, repair and penalties graph by graph instead of once per minibatch")
    p.add_argument("--no-graph-capture", dest="graph_capture", action="store_false",
                   help="evaluate the real-side critic term and the gradient penalty eagerly instead of replaying a HIP graph")
    return p


def main(argv=None) -> None:
    train(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
