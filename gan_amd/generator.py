"""Template-conditioned AIG generator on MI355X kernels.

Drop-in for ``models.generator.AIGGenerator`` of the reference (models/generator.py:10-217):
same constructor arguments, same sub-module names (so state dicts interchange), same
``forward(template, z) -> Data`` with ``x`` (template dtype), ``edge_index`` int64 [2,E],
``edge_attr`` int64 [E,1] and a clone of ``node_depth``.

How the per-node Python loop of the reference maps to kernels:

* the processing order and every target's candidate window depend on the template only
  (types, node_depth, candidate_k), never on earlier picks, so they are computed once per
  template (``GeneratorPlan``) by a radix sort and a wave-parallel walk;
* ``edge_scorer.0`` / ``edge_attr_pred.0`` act on ``[h_src | h_tgt | z]``: their weight splits
  into a source block, a target block and a z block, so the hidden layer of a pair is
  ``relu(A[src] + B[tgt])`` with ``A``/``B`` computed once per node on the f32 matrix cores;
* scoring, ATen-exact top-k and edge assembly are one kernel each over all targets.

Like the reference, the output carries no gradient: every emitted tensor is integer-valued
or a copy, so ``loss_G.backward()`` never reaches these parameters (train.py:297-299).
"""
from __future__ import annotations

import ctypes as C
C_ = C      # (inside the methods that count classes, C is a number)
from typing import Optional

import torch
import torch.nn as nn

from . import _abi
from .data import Data

PI, PO, AND = 0, 1, 2


class GeneratorPlan:
    """Order, candidate windows and edge offsets of one template for one candidate_k."""

    __slots__ = ("n", "K", "E", "types", "order", "rank", "want", "cnt", "cand", "fb", "eoff", "key", "C", "cls", "x_cls")

    @staticmethod
    def build(x: torch.Tensor, node_depth: Optional[torch.Tensor], K: int) -> "GeneratorPlan":
        lib = _abi.lib()
        dev = x.device
        n = int(x.size(0))
        p = GeneratorPlan()
        p.n, p.K = n, int(K)
        i32 = dict(dtype=torch.int32, device=dev)
        p.types, p.order, p.rank = torch.empty(n, **i32), torch.empty(n, **i32), torch.empty(n, **i32)
        kind, dptr = 0, None
        if node_depth is not None and node_depth.numel() == n:
            nd = node_depth.to(dev)
            if nd.dtype.is_floating_point:
                nd, kind = nd.to(torch.float32).contiguous(), 1
            else:
                nd, kind = nd.to(torch.int64).contiguous(), 2
            dptr = _abi.ptr(nd)
        ws = _abi.workspace(lib.aig_gen_order_ws_bytes(n), dev)
        _abi.check(lib.aig_gen_order(n, _abi.ptr(x), kind, dptr, _abi.ptr(p.types), _abi.ptr(p.order),
                                     _abi.ptr(p.rank), _abi.ptr(ws), ws.numel(), _abi.stream()), "aig_gen_order")
        p.want, p.cnt, p.fb = torch.empty(n, **i32), torch.empty(n, **i32), torch.empty(n, **i32)
        p.cand = torch.empty((max(n, 1), p.K), **i32)
        p.eoff = torch.empty(n + 1, **i32)
        ws2 = _abi.workspace(lib.aig_gen_candidates_ws_bytes(n, p.K), dev)
        _abi.check(lib.aig_gen_candidates(n, p.K, _abi.ptr(p.types), _abi.ptr(p.order), _abi.ptr(p.rank),
                                          _abi.ptr(p.want), _abi.ptr(p.cnt), _abi.ptr(p.cand), _abi.ptr(p.fb),
                                          _abi.ptr(p.eoff), _abi.ptr(ws2), ws2.numel(), _abi.stream()),
                   "aig_gen_candidates")
        p.E = int(p.eoff[n].item())  # the one host read per template: sizes the output tensors
        # classes of equal feature rows (exact bit patterns): the generator sees a node through its row alone, so the
        # node pipeline runs over one row per class and pairs are scored per (source class, target class)
        p.C, p.cls, p.x_cls = 0, None, None
        if n > 0:
            bits = x.contiguous().view(torch.int32).to(torch.int64)
            rowkey = (bits[:, 0] << 32) | (bits[:, 1] & 0xFFFFFFFF)
            # (one stable sort: classes numbered in key order, each represented by its lowest node - no reduction that
            # sends every node's atomic to one of a handful of addresses)
            skey, by_key = torch.sort(rowkey, stable=True)
            opens = torch.ones(n, dtype=torch.bool, device=dev)
            opens[1:] = skey[1:] != skey[:-1]
            first = by_key[opens]
            p.C = int(first.numel())
            inv = torch.empty(n, dtype=torch.int64, device=dev)
            inv[by_key] = torch.cumsum(opens, 0) - 1
            p.cls, p.x_cls = inv.to(torch.int32).contiguous(), x[first].contiguous()
        return p


CLASS_LIMIT = 1024      # distinct feature rows up to which a template is scored per class (pair tables of C x C entries)
SELECT_FORMS_SCORES_MAX_K = 127     # windows up to which aig_gen_wire_batch takes logits = NULL with scores by class
# ... and the slots of a call from which on that pays: the selection kernel then runs two full waves a block and forming
# the scores in it costs less than the pass that stores them (370-400 against 334 + 242 us for a million slots); below, its
# blocks are fractions of a wave and the two kernels are faster (89 against 103 us for 100k slots)
SELECT_FORMS_SCORES_MIN_SLOTS = 262145


def _plan_key(x: torch.Tensor, nd: Optional[torch.Tensor], K: int):
    return (K, x.data_ptr(), x._version, tuple(x.shape),
            None if nd is None else (nd.data_ptr(), nd._version, nd.dtype))


class AIGGenerator(nn.Module):
    def __init__(self, node_in_dim: int = 2, hidden_dim: int = 128, z_dim: int = 128, candidate_k: int = 512,
                 device: torch.device = torch.device("cpu"), score_dtype: str = "fp32"):
        super().__init__()
        if score_dtype not in ("fp32", "bf16"):
            raise ValueError("score_dtype is 'fp32' (the reference's arithmetic) or 'bf16'")
        # "bf16": the node projections behind the scores run on the bf16 matrix cores with f32 accumulation, their
        # planes are stored as IEEE halfs and the pairs are scored in packed half arithmetic
        # (an extension for bulk sampling; candidate windows, tie handling and assembly are unchanged)
        self.score_dtype = score_dtype
        # templates whose nodes take at most this many distinct feature rows are scored by class (same outputs, bit
        # for bit: equal rows give equal projections whatever row of a tile they sit in); 0 switches it off
        self.class_limit = CLASS_LIMIT
        self.hidden_dim = hidden_dim
        self.z_dim = z_dim
        self.candidate_k = candidate_k
        self.device = device
        self.node_enc = nn.Sequential(nn.Linear(node_in_dim, hidden_dim), nn.ReLU(),
                                      nn.Linear(hidden_dim, hidden_dim), nn.ReLU())
        self.node_proj = nn.Sequential(nn.Linear(hidden_dim + z_dim, hidden_dim), nn.ReLU(),
                                       nn.Linear(hidden_dim, hidden_dim), nn.ReLU())
        self.edge_scorer = nn.Sequential(nn.Linear(hidden_dim * 2 + z_dim, hidden_dim), nn.ReLU(),
                                         nn.Linear(hidden_dim, 1))
        self.edge_attr_pred = nn.Sequential(nn.Linear(hidden_dim * 2 + z_dim, hidden_dim), nn.ReLU(),
                                            nn.Linear(hidden_dim, 1))
        if (node_in_dim, hidden_dim) != (2, 128):
            raise NotImplementedError(
                "the MI355X kernels are built for node_in_dim=2, hidden_dim=128 (the reference defaults)")
        if not (4 <= z_dim <= 1024 and z_dim % 4 == 0):
            raise NotImplementedError("the MI355X kernels take z_dim as a multiple of 4 in 4..1024 (default 128)")

    # ------------------------------------------------------------------ plumbing
    def _weights(self) -> _abi.GenWeights:
        """Pointer table of the 16 parameter tensors, rebuilt only when one of them moved (``.to()``,
        ``load_state_dict`` on a new device): it is needed once per forward pass."""
        # Walking the module tree costs ~40 us, a fifth of a forward pass on an ISCAS85-sized graph: the list
        # of the 16 Parameter objects is kept (``.to()`` and ``load_state_dict`` change their storage, not the
        # objects) and re-read every 256 calls in case somebody assigned a new Parameter.
        d = self.__dict__
        plist, age = d.get("_plist"), d.get("_plist_age", 0)
        if plist is None or age >= 256:
            plist, age = list(self.parameters()), 0
            d["_plist"] = plist
        d["_plist_age"] = age + 1
        key = tuple(t.data_ptr() for t in plist)
        cached = self.__dict__.get("_weights_cache")
        if cached is not None and cached[0] == key:
            return cached[1]
        w = self._weights_uncached()
        self.__dict__["_weights_cache"] = (key, w)
        return w

    def _weights_uncached(self) -> _abi.GenWeights:
        def p(t):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise _abi.AigAbiError("generator parameters must be contiguous float32 tensors on a HIP device")
            return t.data_ptr()
        w = _abi.GenWeights()
        w.enc0_w, w.enc0_b = p(self.node_enc[0].weight), p(self.node_enc[0].bias)
        w.enc2_w, w.enc2_b = p(self.node_enc[2].weight), p(self.node_enc[2].bias)
        w.proj0_w, w.proj0_b = p(self.node_proj[0].weight), p(self.node_proj[0].bias)
        w.proj2_w, w.proj2_b = p(self.node_proj[2].weight), p(self.node_proj[2].bias)
        w.sc0_w, w.sc0_b = p(self.edge_scorer[0].weight), p(self.edge_scorer[0].bias)
        w.sc2_w, w.sc2_b = p(self.edge_scorer[2].weight), p(self.edge_scorer[2].bias)
        w.at0_w, w.at0_b = p(self.edge_attr_pred[0].weight), p(self.edge_attr_pred[0].bias)
        w.at2_w, w.at2_b = p(self.edge_attr_pred[2].weight), p(self.edge_attr_pred[2].bias)
        w.z_dim = int(self.z_dim)
        return w

    def plan_for(self, template: Data, x: torch.Tensor, nd: Optional[torch.Tensor]) -> GeneratorPlan:
        """Cached on the template object; rebuilt when its tensors or candidate_k change."""
        key = _plan_key(x, nd, int(self.candidate_k))
        plan = template.__dict__.get("_gen_plan")
        if plan is None or plan.key != key:
            plan = GeneratorPlan.build(x, nd, int(self.candidate_k))
            plan.key = key
            template.__dict__["_gen_plan"] = plan
        return plan

    @torch.no_grad()
    def forward_many(self, template: Data, zs: torch.Tensor) -> list:
        """``[self(template, z) for z in zs]`` for bulk sampling from ONE template (``zs`` is ``[S, z_dim]``): the
        plan is shared, and with ``score_dtype="bf16"`` the node projections of all S samples are ONE launch
        (the z of a sample only enters as bias vectors), which is what gives the matrix cores rows to work on
        when the template is a few thousand nodes.  Scores, selection and assembly follow per sample; every
        returned graph has the tensors ``forward`` returns for its z."""
        lib = _abi.lib()
        dev = template.x.device
        if dev.type != "cuda":
            raise _abi.AigAbiError("AIGGenerator.forward_many needs the template on a HIP device (no CPU path exists)")
        zs = zs.to(dev).float().contiguous()
        if zs.dim() != 2 or int(zs.size(1)) != self.z_dim:
            raise _abi.AigAbiError(f"zs must be [samples, z_dim={self.z_dim}], got {tuple(zs.shape)}")
        if zs.data_ptr() % 16:      # (a view into a larger tensor: the kernels read z as 16-byte vectors)
            zs = zs.clone()
        S = int(zs.size(0))
        if self.score_dtype != "bf16" or S == 1:
            return [self(template, zs[i]) for i in range(S)]
        x_src = template.x
        x_in = x_src if (x_src.dtype == torch.float32 and x_src.is_contiguous()) else x_src.float().contiguous()
        n = int(x_in.size(0))
        nd = getattr(template, "node_depth", None)
        nd_dev = nd.to(dev) if nd is not None else None
        plan = self.plan_for(template, x_in, nd_dev)
        K, E = plan.K, plan.E
        st = _abi.stream()
        n64 = (n + 63) // 64 * 64
        proj = torch.empty((4, S * n64, 128), dtype=torch.float16, device=dev)
        ws = _abi.workspace(lib.aig_gen_node_proj_bf16_ws_bytes(S), dev)
        w = self._weights()
        _abi.check(_abi.timed("aig_gen_node_proj_bf16", lib.aig_gen_node_proj_bf16, n, S, _abi.ptr(x_in), _abi.ptr(zs),
                              C.byref(w), None, _abi.ptr(proj), _abi.ptr(ws), ws.numel(), st, meta=(n, S)),
                   "aig_gen_node_proj_bf16")
        outs = []
        for i in range(S):
            rows = slice(i * n64, i * n64 + n)
            outs.append(self._wire(plan, x_in, x_src, nd, proj[0, rows], proj[1, rows], proj[2, rows], proj[3, rows], n, K, E, dev))
        return outs

    # ------------------------------------------------------------------ sets of graphs
    def _stacked_plans(self, templates, dev):
        """Plan arrays, node features and depths of several templates of one node count, back to back.  The
        last few stacks are kept: the two generator passes of a training step use the same templates."""
        xs, nds, plans = [], [], []
        for t in templates:
            x_src = t.x
            x_in = x_src if (x_src.dtype == torch.float32 and x_src.is_contiguous()) else x_src.float().contiguous()
            nd = getattr(t, "node_depth", None)
            nd_dev = nd.to(dev) if nd is not None else None
            plans.append(self.plan_for(t, x_in, nd_dev))
            xs.append(x_in)
            nds.append(nd_dev)
        key = tuple(id(p) for p in plans)
        cache = self.__dict__.setdefault("_plan_stacks", {})
        hit = cache.get(key)
        if hit is not None and all(a is b for a, b in zip(hit["plans"], plans)):
            return hit
        n = plans[0].n
        if any(p.n != n for p in plans):
            raise ValueError("forward_set needs templates of one node count")
        if len(plans) == 1:     # nothing to stack: the plan's own tensors
            st = {"plans": plans, "n": n, "E": [plans[0].E], "x": xs[0]}
            for name in ("order", "rank", "want", "cnt", "fb", "eoff", "cand"):
                st[name] = getattr(plans[0], name)
            st["nd"] = nds[0] if nds[0] is not None and nds[0].numel() == n else None
            st["C"], st["cls"], st["x_cls"] = plans[0].C, plans[0].cls, plans[0].x_cls
            return st
        st = {"plans": plans, "n": n, "E": [p.E for p in plans], "x": torch.cat(xs).contiguous()}
        # classes: every template's class rows padded to the largest class count (the padding repeats row 0)
        C = max(p.C for p in plans)
        st["C"] = C
        st["cls"] = torch.stack([p.cls for p in plans]).contiguous()
        st["x_cls"] = torch.cat([torch.cat([p.x_cls, p.x_cls[:1].expand(C - p.C, 2)]) for p in plans]).contiguous()
        for name in ("order", "rank", "want", "cnt", "fb", "eoff", "cand"):
            st[name] = torch.stack([getattr(p, name) for p in plans]).contiguous()
        st["nd"] = torch.cat(nds).contiguous() if all(v is not None and v.numel() == n for v in nds) else None
        if len(cache) >= 4:
            cache.pop(next(iter(cache)))
        cache[key] = st
        return st

    @torch.no_grad()
    def forward_set(self, templates, zs: torch.Tensor):
        """``self(template, z)`` for a whole set of graphs in five launches: ``templates`` is ONE template (the
        graphs are samples of it that differ in z alone - bulk sampling) or a list of B templates of equal node
        count (the graphs of a padded minibatch), ``zs`` is ``[B, z_dim]``.  Returns a ``GraphSet`` whose graph
        ``g`` holds exactly the tensors ``self(template_g, zs[g])`` returns."""
        from .graphset import GraphSet
        lib = _abi.lib()
        shared = not isinstance(templates, (list, tuple))
        first = templates if shared else templates[0]
        dev = first.x.device
        if dev.type != "cuda":
            raise _abi.AigAbiError("AIGGenerator.forward_set needs the templates on a HIP device (no CPU path exists)")
        zs = zs.to(dev).float().contiguous()
        if zs.dim() != 2 or int(zs.size(1)) != self.z_dim:
            raise _abi.AigAbiError(f"zs must be [samples, z_dim={self.z_dim}], got {tuple(zs.shape)}")
        if zs.data_ptr() % 16:      # (a view into a larger tensor: the kernels read z as 16-byte vectors)
            zs = zs.clone()
        B = int(zs.size(0))
        K = int(self.candidate_k)
        if shared:
            x_src = first.x
            x_all = x_src if (x_src.dtype == torch.float32 and x_src.is_contiguous()) else x_src.float().contiguous()
            nd = getattr(first, "node_depth", None)
            nd_all = nd.to(dev) if nd is not None else None
            plan = self.plan_for(first, x_all, nd_all)
            n, Es = plan.n, [plan.E] * B
            arrays = {name: getattr(plan, name) for name in ("order", "rank", "want", "cnt", "fb", "eoff", "cand")}
            if nd_all is not None:
                nd_all = nd_all.clone()
        else:
            if len(templates) != B:
                raise ValueError("forward_set: one z per template")
            st = self._stacked_plans(templates, dev)
            n, Es, x_all, arrays = st["n"], st["E"], st["x"], st
            nd_all = None if st["nd"] is None else st["nd"].clone()
        if shared:
            C, cls, x_cls = plan.C, plan.cls, plan.x_cls
        else:
            C, cls, x_cls = st["C"], st["cls"], st["x_cls"]
        ecap = max(max(Es), 1)
        _, _, edge_index, edge_attr, ecount, x_out = self._wire_set(B, n, K, ecap, shared, x_all, arrays, zs, C, cls, x_cls)
        out = GraphSet(B, n, ecap, x_out, edge_index, edge_attr, ecount, nd_all, counts_host=Es, x_dtype=first.x.dtype)
        out.order_hint = arrays["rank"]      # every emitted edge runs from an earlier to a later node of this order
        return out

    def _wire_set(self, B, n, K, ecap, shared, x_all, arrays, zs, C, cls, x_cls, h_cond=None, keep=None):
        """Node projections, scores, selection and assembly of B graphs of n nodes (the launches behind ``forward_set``
        and, with B = 1, ``forward``).  When the templates' nodes take at most ``class_limit`` distinct feature rows
        the projections are made per class and pairs are scored through per-graph (source class, target class)
        tables - the same bits, since a projection depends on its own input row only.  ``keep``: a dict that receives
        the projection planes (and the class-wise h_cond when ``h_cond`` is asked for) for debugging."""
        lib = _abi.lib()
        dev = x_all.device
        stream = _abi.stream()
        bf16 = 1 if self.score_dtype == "bf16" else 0
        w = self._weights()
        by_class = bool(self.class_limit) and 0 < C <= self.class_limit
        m = C if by_class else n                       # rows per graph that go through the node pipeline
        x_rows = x_cls if by_class else x_all
        rows = int(lib.aig_gen_batch_rows(m))
        pdt = torch.float16 if bf16 else torch.float32
        if h_cond is not None:          # (one graph: the single-graph entry points also hand out h_cond)
            assert B == 1
            proj = torch.empty((4, m, 128), dtype=pdt, device=dev)
            hc = torch.empty((m, 128), dtype=torch.float32, device=dev)
            if bf16:
                ws = _abi.workspace(lib.aig_gen_node_proj_bf16_ws_bytes(1), dev)
                _abi.check(_abi.timed("aig_gen_node_proj_bf16", lib.aig_gen_node_proj_bf16, m, 1, _abi.ptr(x_rows), _abi.ptr(zs),
                                      C_.byref(w), _abi.ptr(hc), _abi.ptr(proj), _abi.ptr(ws), ws.numel(), stream),
                           "aig_gen_node_proj_bf16")
            else:
                ws = _abi.workspace(lib.aig_gen_node_proj_ws_bytes(m), dev)
                _abi.check(_abi.timed("aig_gen_node_proj", lib.aig_gen_node_proj, m, _abi.ptr(x_rows), _abi.ptr(zs), C_.byref(w),
                                      _abi.ptr(hc), _abi.ptr(proj), _abi.ptr(ws), ws.numel(), stream), "aig_gen_node_proj")
            rows = m
        else:
            proj = torch.empty((4, B * rows, 128), dtype=pdt, device=dev)
            ws = _abi.workspace(lib.aig_gen_node_proj_batch_ws_bytes(B, m, bf16), dev)
            _abi.check(_abi.timed("aig_gen_node_proj_batch", lib.aig_gen_node_proj_batch, B, m, _abi.ptr(x_rows), int(shared),
                                  _abi.ptr(zs), C_.byref(w), bf16, _abi.ptr(proj), _abi.ptr(ws), ws.numel(), stream,
                                  meta=(m, B)), "aig_gen_node_proj_batch")
            hc = None
        a = _abi.GenWireSet()
        a.B, a.n, a.K, a.ecap, a.plan_shared, a.x_shared = B, n, K, ecap, int(shared), int(shared)
        a.x = _abi.ptr(x_all)
        for name in ("order", "rank", "want", "cnt", "fb", "eoff", "cand"):
            setattr(a, name, _abi.ptr(arrays[name]))
        a.proj, a.proj_rows, a.proj_half = _abi.ptr(proj), rows, bf16
        if by_class:
            pair_logit = torch.empty((B, C, C), dtype=torch.float32, device=dev)
            pair_bit = torch.empty((B, C, C), dtype=torch.int32, device=dev)
            a.cls, a.classes, a.pair_logit, a.pair_bit = _abi.ptr(cls), C, _abi.ptr(pair_logit), _abi.ptr(pair_bit)
        a.sc2_w, a.sc2_b = _abi.ptr(self.edge_scorer[2].weight), _abi.ptr(self.edge_scorer[2].bias)
        a.at2_w, a.at2_b = _abi.ptr(self.edge_attr_pred[2].weight), _abi.ptr(self.edge_attr_pred[2].bias)
        # (scores by class and windows short enough for the selection kernel's LDS copy: it forms the scores itself
        #  and the n K logits are never stored - unless the caller wants to look at them)
        logits = None
        if keep is not None or not by_class or K > SELECT_FORMS_SCORES_MAX_K or B * n < SELECT_FORMS_SCORES_MIN_SLOTS:
            logits = torch.empty((B, n, K), dtype=torch.float32, device=dev)
            if keep is not None:
                logits.fill_(float("nan"))
        sel = torch.empty((B, n, 2), dtype=torch.int32, device=dev)
        edge_index = torch.empty((B, 2, ecap), dtype=torch.int64, device=dev)
        edge_attr = torch.empty((B, ecap), dtype=torch.int64, device=dev)
        ecount = torch.empty(B, dtype=torch.int32, device=dev)
        x_out = torch.empty((B * n, 2), dtype=torch.float32, device=dev)
        a.logits, a.sel = _abi.ptr(logits), _abi.ptr(sel)
        a.edge_index, a.edge_attr, a.ecount, a.x_out = _abi.ptr(edge_index), _abi.ptr(edge_attr), _abi.ptr(ecount), _abi.ptr(x_out)
        _abi.check(_abi.timed("aig_gen_wire_batch", lib.aig_gen_wire_batch, C_.byref(a), stream, meta=(n, B, K)),
                   "aig_gen_wire_batch")
        if keep is not None:
            idx = cls.long() if by_class else None
            keep["proj"] = proj[:, idx] if by_class else proj
            keep["h_cond"] = None if hc is None else (hc[idx] if by_class else hc)
            keep["by_class"] = by_class
        return logits, sel, edge_index, edge_attr, ecount, x_out

    def _wire(self, plan, x_in, x_src, nd, pa, pb, pa2, pb2, n, K, E, dev, debug=None) -> Data:
        """Scores, tie-exact selection and edge assembly from the four projection planes (generator.py:139-209);
        planes of halfs (the bf16 pipeline's) take the packed-half forms of the two entry points that read them."""
        lib = _abi.lib()
        st = _abi.stream()
        half = pa.dtype == torch.float16
        score_name, emit_name = ("aig_gen_score_half", "aig_gen_emit_half") if half else ("aig_gen_score", "aig_gen_emit")
        logits = torch.empty((max(n, 1), K), dtype=torch.float32, device=dev)
        if debug is not None:
            logits.fill_(float("nan"))
        _abi.check(_abi.timed(score_name, getattr(lib, score_name), n, K, _abi.ptr(plan.order), _abi.ptr(plan.rank), _abi.ptr(plan.cnt), _abi.ptr(plan.cand),
                                     _abi.ptr(pa), _abi.ptr(pb), _abi.ptr(self.edge_scorer[2].weight),
                                     _abi.ptr(self.edge_scorer[2].bias), _abi.ptr(logits), st), score_name)
        sel = torch.empty((max(n, 1), 2), dtype=torch.int32, device=dev)
        _abi.check(_abi.timed("aig_topk_select", lib.aig_topk_select, n, K, _abi.ptr(logits), _abi.ptr(plan.cnt), _abi.ptr(plan.want),
                                       _abi.ptr(sel), None, 0, st), "aig_topk_select")
        edge_index = torch.empty((2, E), dtype=torch.int64, device=dev)
        edge_attr = torch.empty((E, 1), dtype=torch.int64, device=dev)
        x_out = torch.empty((n, 2), dtype=torch.float32, device=dev)
        _abi.check(_abi.timed(emit_name, getattr(lib, emit_name), n, K, E, _abi.ptr(x_in), _abi.ptr(plan.order), _abi.ptr(plan.want),
                                    _abi.ptr(plan.cnt), _abi.ptr(plan.cand), _abi.ptr(plan.fb), _abi.ptr(plan.eoff),
                                    _abi.ptr(sel), _abi.ptr(pa2), _abi.ptr(pb2),
                                    _abi.ptr(self.edge_attr_pred[2].weight), _abi.ptr(self.edge_attr_pred[2].bias),
                                    _abi.ptr(edge_index) if E else None, _abi.ptr(edge_attr) if E else None,
                                    _abi.ptr(x_out), st), emit_name)
        if x_src.dtype != torch.float32:
            x_out = x_out.to(x_src.dtype)
        out = Data(x=x_out, edge_index=edge_index, edge_attr=edge_attr,
                   node_depth=(nd.clone().to(dev) if nd is not None else None))
        if debug is not None:
            debug.update({"logits": logits, "sel": sel})
        return out

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, template: Data, z: torch.Tensor, return_debug: bool = False) -> Data:
        lib = _abi.lib()
        dev = template.x.device if hasattr(template, "x") else self.device
        if dev.type != "cuda":
            raise _abi.AigAbiError("AIGGenerator.forward needs the template on a HIP device (no CPU path exists)")
        if z.dim() == 2 and z.size(0) == 1:
            z = z.squeeze(0)
        z = z.to(dev).float().contiguous()
        if z.numel() != self.z_dim:
            raise _abi.AigAbiError(f"z must hold z_dim={self.z_dim} values, got {tuple(z.shape)}")
        if z.data_ptr() % 16:       # (a view into a larger tensor: the kernels read z as 16-byte vectors)
            z = z.clone()
        x_src = template.x
        x_in = x_src if (x_src.dtype == torch.float32 and x_src.is_contiguous()) else x_src.float().contiguous()
        n = int(x_in.size(0))
        nd = getattr(template, "node_depth", None)
        nd_dev = nd.to(dev) if nd is not None else None
        plan = self.plan_for(template, x_in, nd_dev)
        K, E = plan.K, plan.E
        st = _abi.stream()
        f32 = dict(dtype=torch.float32, device=dev)

        if self.class_limit and 0 < plan.C <= self.class_limit and n > 0:
            # per class: the launches of forward_set for a set of one graph
            arrays = {name: getattr(plan, name) for name in ("order", "rank", "want", "cnt", "fb", "eoff", "cand")}
            keep = {} if return_debug else None
            logits, sel, ei, ea, _, x_out = self._wire_set(1, n, K, max(E, 1), True, x_in, arrays, z.reshape(1, -1), plan.C,
                                                           plan.cls, plan.x_cls, h_cond=True if return_debug else None, keep=keep)
            if x_src.dtype != torch.float32:
                x_out = x_out.to(x_src.dtype)
            out = Data(x=x_out, edge_index=ei[0, :, :E], edge_attr=ea[0, :E].reshape(-1, 1),
                       node_depth=(nd.clone().to(dev) if nd is not None else None))
            if return_debug:
                out._debug = {"plan": plan, "h_cond": keep["h_cond"], "proj": keep["proj"], "logits": logits[0], "sel": sel[0]}
            return out
        proj = torch.empty((4, max(n, 1), 128), dtype=torch.float16 if self.score_dtype == "bf16" else torch.float32,
                           device=dev)
        h_cond = torch.empty((max(n, 1), 128), **f32) if return_debug else None
        w = self._weights()
        if self.score_dtype == "bf16":
            ws = _abi.workspace(lib.aig_gen_node_proj_bf16_ws_bytes(1), dev)
            _abi.check(_abi.timed("aig_gen_node_proj_bf16", lib.aig_gen_node_proj_bf16, n, 1, _abi.ptr(x_in), _abi.ptr(z),
                                  C.byref(w), _abi.ptr(h_cond), _abi.ptr(proj), _abi.ptr(ws), ws.numel(), st),
                       "aig_gen_node_proj_bf16")
        else:
            ws = _abi.workspace(lib.aig_gen_node_proj_ws_bytes(n), dev)
            _abi.check(_abi.timed("aig_gen_node_proj", lib.aig_gen_node_proj, n, _abi.ptr(x_in), _abi.ptr(z), C.byref(w),
                                  _abi.ptr(h_cond), _abi.ptr(proj), _abi.ptr(ws), ws.numel(), st), "aig_gen_node_proj")
        debug = {"plan": plan, "h_cond": h_cond, "proj": proj} if return_debug else None
        out = self._wire(plan, x_in, x_src, nd, proj[0], proj[1], proj[2], proj[3], n, K, E, dev, debug)
        if return_debug:
            out._debug = debug
        return out
