"""Sets of equally sized graphs that go through the kernels together.

The reference handles the graphs of a minibatch (train.py:242-245,287) and the samples of a bulk run
(AIGgan.py:104-110) one after another; here ``B`` graphs of ``n`` nodes each are one set of tensors, and the
generator's wiring, the repair, the penalties and the resync take the whole set in a handful of launches.

Layout (what the ``*_batch`` entry points of the library take):

* ``x``            float32 ``[B*n, 2]`` - the graphs' node features back to back
* ``edge_index``   int64 ``[B, 2, ecap]`` - graph ``g`` keeps its sources in ``[g, 0, :count[g]]`` and its
  targets in ``[g, 1, :count[g]]`` (node ids inside the graph); the slots behind are unspecified
* ``edge_attr``    int64 ``[B, ecap]``
* ``count``        int32 ``[B]`` on the device (``None``: every slot holds an edge)
* ``node_depth``   ``[B*n]``, or ``[n]`` when all graphs share one template's depths, or ``None``
* ``order_hint``   int32 ``[B*n]`` / ``[n]`` / ``None``: see ``GraphSet.__init__``
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .data import Data


class GraphSet:
    __slots__ = ("B", "n", "ecap", "x", "edge_index", "edge_attr", "count", "node_depth", "_counts_host", "x_dtype",
                 "order_hint", "_views")

    def __init__(self, B: int, n: int, ecap: int, x: torch.Tensor, edge_index: torch.Tensor, edge_attr: torch.Tensor,
                 count: Optional[torch.Tensor], node_depth: Optional[torch.Tensor] = None,
                 counts_host: Optional[Sequence[int]] = None, x_dtype: torch.dtype = torch.float32):
        self.B, self.n, self.ecap = int(B), int(n), int(ecap)
        self.x, self.edge_index, self.edge_attr, self.count, self.node_depth = x, edge_index, edge_attr, count, node_depth
        self._counts_host = None if counts_host is None else [int(c) for c in counts_host]
        self.x_dtype = x_dtype      # dtype the per-graph views hand out for x (the template's)
        # int32 [B*n] or [n] (shared), or None: place of every node in an order in which the set's edges run from
        # earlier to later nodes - the generator leaves its processing rank here; the penalties use it as a
        # (device-checked) shortcut for their reachability term
        self.order_hint = None
        self._views = None

    # ------------------------------------------------------------------ host side of the counts
    def counts(self) -> List[int]:
        """Edges per graph on the host (one device read unless they were known when the set was made)."""
        if self._counts_host is None:
            self._counts_host = [self.ecap] * self.B if self.count is None else [int(c) for c in self.count.tolist()]
        return self._counts_host

    @property
    def depth_shared(self) -> bool:
        return self.node_depth is not None and self.B > 1 and self.node_depth.numel() == self.n

    # ------------------------------------------------------------------ per-graph views
    def graph(self, g: int, copy: bool = True) -> Data:
        """Graph ``g`` as the ``Data`` the single-graph functions return.  ``copy=False`` hands out views of
        the set's tensors instead of tensors of its own (``edge_index`` is then a strided ``[2, E]`` view)."""
        m = self.counts()[g]
        n = self.n
        if self._views is None:      # per-graph views of the node and slot arrays, cut once for the whole set
            nd = self.node_depth
            self._views = (self.x.view(self.B, n, -1).unbind(0), self.edge_index.unbind(0), self.edge_attr.unbind(0),
                           None if nd is None or nd.numel() == n else nd.view(self.B, n).unbind(0))
        xs, eis, eas, nds = self._views
        x = xs[g]
        nd = self.node_depth if nds is None else nds[g]
        ei, ea = eis[g].narrow(1, 0, m), eas[g].narrow(0, 0, m).unsqueeze(1)
        if copy:
            x, ei, ea = x.clone(), ei.contiguous(), ea.clone()
            nd = None if nd is None else nd.clone()
        if x.dtype != self.x_dtype:
            x = x.to(self.x_dtype)
        return Data(x=x, edge_index=ei, edge_attr=ea, node_depth=nd)

    def graphs(self, copy: bool = True) -> List[Data]:
        return [self.graph(g, copy) for g in range(self.B)]

    def cpu(self) -> "GraphSet":
        """The set on the host (a few large copies; its graphs are then cut out there)."""
        counts = self.counts()
        return GraphSet(self.B, self.n, self.ecap, self.x.cpu(), self.edge_index.cpu(), self.edge_attr.cpu(), None,
                        None if self.node_depth is None else self.node_depth.cpu(), counts_host=counts,
                        x_dtype=self.x_dtype)

    # ------------------------------------------------------------------ from separate graphs
    @staticmethod
    def from_data_list(graphs: Sequence[Data]) -> "GraphSet":
        """Stack graphs of equal node count (their edge counts may differ)."""
        n = int(graphs[0].x.size(0))
        if any(int(d.x.size(0)) != n for d in graphs):
            raise ValueError("a GraphSet holds graphs of one node count")
        dev = graphs[0].x.device
        B = len(graphs)
        es = [0 if getattr(d, "edge_index", None) is None else int(d.edge_index.size(1)) for d in graphs]
        ecap = max(max(es), 1)
        ei = torch.zeros((B, 2, ecap), dtype=torch.int64, device=dev)
        ea = torch.zeros((B, ecap), dtype=torch.int64, device=dev)
        for g, d in enumerate(graphs):
            if es[g]:
                ei[g, :, :es[g]] = d.edge_index
                a = getattr(d, "edge_attr", None)
                if a is not None and a.numel():
                    ea[g, :es[g]] = a.reshape(-1)[:es[g]]
        x = torch.cat([d.x.float() for d in graphs]).contiguous()
        nds = [getattr(d, "node_depth", None) for d in graphs]
        nd = None
        if all(v is not None and v.numel() == n for v in nds):
            nd = torch.cat([v.to(dev) for v in nds]).contiguous()
        count = torch.tensor(es, dtype=torch.int32, device=dev)
        return GraphSet(B, n, ecap, x, ei, ea, count, nd, counts_host=es, x_dtype=graphs[0].x.dtype)
