"""Graph containers for the AIG-GAN hot path.

The reference keeps every graph in a ``torch_geometric.data.Data`` attribute bag
(reference: models/generator.py:211-217, utilities/data_process.py:60-68) and
batches them with ``Batch.from_data_list`` / ``to_data_list`` (train.py:236,245).
torch_geometric is not a dependency here: this module owns the small subset of
that interface the path uses, with the same observable behaviour:

* ``x`` / ``edge_index`` / ``edge_attr`` read as ``None`` when never set, any
  other attribute raises ``AttributeError`` (so ``hasattr`` works as callers expect);
* ``.to()`` / ``.cpu()`` move tensors in place and return ``self``;
* batching adds the node offset to *every* ``edge_index`` entry, ``-1`` pad
  sentinels included (train.py:113 relies on ``min >= 0`` afterwards), and only
  public attributes survive a ``from_data_list`` / ``to_data_list`` round trip
  (``_orig_n`` / ``_orig_e`` written by ``pad_data`` do not), which is why training
  templates keep their padded size in the reference.
"""
from __future__ import annotations

from typing import Any, Dict, Iterator, List, Optional, Sequence

import torch

__all__ = ["Data", "Batch", "GraphLoader"]


class Data:
    """Attribute bag holding one graph (or, as ``Batch``, several)."""

    def __init__(self, x=None, edge_index=None, edge_attr=None, **kwargs: Any):
        self.x = x
        self.edge_index = edge_index
        self.edge_attr = edge_attr
        for k, v in kwargs.items():
            setattr(self, k, v)

    # ------------------------------------------------------------------ basics
    def keys(self) -> List[str]:
        """Public attribute names that hold a value."""
        return [k for k, v in self.__dict__.items() if not k.startswith("_") and v is not None]

    def __contains__(self, key: str) -> bool:
        return key in self.keys()

    def __getitem__(self, key: str):
        return getattr(self, key)

    def __setitem__(self, key: str, value) -> None:
        setattr(self, key, value)

    @property
    def num_nodes(self) -> int:
        if self.x is not None:
            return int(self.x.size(0))
        if self.edge_index is not None and self.edge_index.numel() > 0:
            return int(self.edge_index.max()) + 1
        return 0

    @property
    def num_edges(self) -> int:
        return 0 if self.edge_index is None else int(self.edge_index.size(1))

    def to(self, device, non_blocking: bool = False) -> "Data":
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                self.__dict__[k] = v.to(device, non_blocking=non_blocking)
        return self

    def cpu(self) -> "Data":
        return self.to("cpu")

    def cuda(self, device=None) -> "Data":
        return self.to("cuda" if device is None else device)

    def clone(self) -> "Data":
        out = self.__class__.__new__(self.__class__)
        for k, v in self.__dict__.items():
            out.__dict__[k] = v.clone() if torch.is_tensor(v) else v
        return out

    def to_dict(self) -> Dict[str, Any]:
        """Plain-dict form ({x, edge_index, edge_attr, node_depth, ...}) for ``.pt`` I/O.

        The reference exporters accept this form (utilities/pt2bench.py:218,
        utilities/pt2graphml.py:12), so emitted samples need no pickled class.
        """
        return {k: getattr(self, k) for k in self.keys()}

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "Data":
        return cls(**d)

    def __repr__(self) -> str:
        parts = []
        for k in self.keys():
            v = getattr(self, k)
            parts.append(f"{k}={list(v.shape)}" if torch.is_tensor(v) else f"{k}={v!r}")
        return f"{self.__class__.__name__}({', '.join(parts)})"


class Batch(Data):
    """Several graphs stacked into one disjoint graph (train.py:245,281).

    ``batch`` (graph id per node) and ``ptr`` (node offsets) are built on first use: nothing on the
    training path reads them, and building them on a GPU costs two blocking host-to-device copies
    per minibatch."""

    def __getattr__(self, name: str):   # reached only when the attribute is not set yet
        if name in ("batch", "ptr") and "_node_counts" in self.__dict__:
            counts, dev = self.__dict__["_node_counts"], self.x.device
            self.__dict__["batch"] = torch.repeat_interleave(
                torch.arange(len(counts), device=dev), torch.tensor(counts, device=dev), output_size=sum(counts))
            self.__dict__["ptr"] = torch.tensor(self.__dict__["_offsets"], dtype=torch.long, device=dev)
            return self.__dict__[name]
        raise AttributeError(f"{self.__class__.__name__!r} object has no attribute {name!r}")

    def keys(self) -> List[str]:
        ks = super().keys()
        if "_node_counts" in self.__dict__:
            ks += [k for k in ("batch", "ptr") if k not in ks]
        return ks

    @classmethod
    def from_data_list(cls, data_list: Sequence[Data]) -> "Batch":
        if len(data_list) == 0:
            raise ValueError("from_data_list needs at least one graph")
        out = cls()
        keys = data_list[0].keys()
        node_counts = [int(d.x.size(0)) for d in data_list]
        offsets = [0]
        for n in node_counts:
            offsets.append(offsets[-1] + n)
        slices: Dict[str, List[int]] = {}
        for key in keys:
            vals = [getattr(d, key) for d in data_list]
            if not all(torch.is_tensor(v) for v in vals):
                setattr(out, key, list(vals))
                continue
            if key == "edge_index":
                # every entry is shifted, -1 pad sentinels included
                shifted = [v + off for v, off in zip(vals, offsets[:-1])]
                setattr(out, key, torch.cat(shifted, dim=1))
                cuts = [0]
                for v in vals:
                    cuts.append(cuts[-1] + int(v.size(1)))
            else:
                vals = [v.unsqueeze(0) if v.dim() == 0 else v for v in vals]
                setattr(out, key, torch.cat(vals, dim=0))
                cuts = [0]
                for v in vals:
                    cuts.append(cuts[-1] + int(v.size(0)))
            slices[key] = cuts
        out._node_counts, out._offsets = node_counts, offsets
        out._slices = slices
        out._num_graphs = len(data_list)
        return out

    @classmethod
    def from_uniform(cls, num_graphs: int, nodes: int, edge_cols: int, **tensors) -> "Batch":
        """A batch whose concatenated tensors already exist: ``num_graphs`` graphs of ``nodes`` nodes and
        ``edge_cols`` edge columns each (``edge_index`` with the node offsets added), e.g. straight from a
        kernel that pads a whole set of graphs."""
        out = cls(**tensors)
        out._node_counts = [int(nodes)] * int(num_graphs)
        out._offsets = [g * int(nodes) for g in range(int(num_graphs) + 1)]
        out._slices = {k: [g * (int(edge_cols) if k in ("edge_index", "edge_attr") else int(nodes))
                           for g in range(int(num_graphs) + 1)] for k, v in tensors.items() if torch.is_tensor(v)}
        out._num_graphs = int(num_graphs)
        return out

    @property
    def num_graphs(self) -> int:
        return int(self._num_graphs)

    def to_data_list(self) -> List[Data]:
        outs: List[Data] = []
        ptr = self._offsets
        for g in range(self._num_graphs):
            d = Data()
            for key, cuts in self._slices.items():
                v = getattr(self, key)
                lo, hi = cuts[g], cuts[g + 1]
                if key == "edge_index":
                    setattr(d, key, v[:, lo:hi] - ptr[g])
                else:
                    setattr(d, key, v[lo:hi])
            outs.append(d)
        return outs


class GraphLoader:
    """Shuffling minibatch loader over a list of (padded) graphs.

    Stands in for the ``torch_geometric.loader.DataLoader`` call at train.py:206-214.
    ``rank`` / ``world_size`` give every data-parallel rank the same shuffled
    order and every ``world_size``-th minibatch of it; batches are dropped at the
    tail so that all ranks run the same number of steps (a rank left without a
    batch would not enter the gradient all-reduce and the others would block).
    """

    def __init__(self, data_list: Sequence[Data], batch_size: int = 1, shuffle: bool = True,
                 seed: Optional[int] = None, rank: int = 0, world_size: int = 1):
        self.data_list = list(data_list)
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.rank, self.world_size = int(rank), int(world_size)
        self._gen = torch.Generator()
        if seed is None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
        self._gen.manual_seed(seed)
        # deepest level per graph, from the host-side data: consumers that size device structures by it
        # (gan_amd.train.depth_top) then need no host read of a batch that already lives on the GPU
        self.depth_tops = []
        for d in self.data_list:
            nd = getattr(d, "node_depth", None)
            self.depth_tops.append(-1 if nd is None or nd.numel() == 0 else int(nd.max().item()))

    def _batches(self) -> List[List[int]]:
        n = len(self.data_list)
        perm = torch.randperm(n, generator=self._gen).tolist() if self.shuffle else list(range(n))
        batches = [perm[i:i + self.batch_size] for i in range(0, n, self.batch_size)]
        if self.world_size > 1:
            usable = (len(batches) // self.world_size) * self.world_size
            batches = batches[:usable][self.rank::self.world_size]
        return batches

    def __len__(self) -> int:
        nb = (len(self.data_list) + self.batch_size - 1) // self.batch_size
        return nb // self.world_size if self.world_size > 1 else nb

    def __iter__(self) -> Iterator[Batch]:
        for idxs in self._batches():
            b = Batch.from_data_list([self.data_list[i] for i in idxs])
            b.sample_ids = list(idxs)
            b._depth_top = max(self.depth_tops[i] for i in idxs)
            yield b
