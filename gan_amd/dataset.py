"""Dataset plumbing for the AIG-GAN path: ``.pt`` files in, fixed-size padded graphs out.

Behaviour follows utilities/data_process.py of the reference:

* ``pad_data`` (:30-68): ``x`` grows to ``max_nodes`` rows of zeros (a zero row reads as a PI),
  ``node_depth`` becomes int64 with -1 on the padding (0 for real nodes when the graph has none),
  ``edge_index`` becomes int64 ``[2, max_edges]`` with -1 columns behind the first
  ``min(E, max_edges)`` edges, ``edge_attr`` int64 ``[max_edges, 1]`` with zeros; the true
  sizes ride along as ``_orig_n`` / ``_orig_e``;
* ``unpad_data`` (:70-82): cuts back to those sizes while they are still attached and keeps
  the padded size otherwise - private attributes do not survive ``Batch.from_data_list`` /
  ``to_data_list``, which is how the reference's training templates stay padded;
* files carry the plain dict ``{x, edge_index, edge_attr, node_depth}`` (the form the
  reference's exporters accept too).  A pickled torch_geometric object - what the reference's own
  ``torch.load(..., weights_only=False)`` reads (:18) - executes code from the file when loaded, so it is read
  only when the caller says the files are trusted (``allow_pickle=True``, ``--allow-pickle`` on the CLIs) and
  torch_geometric itself can be imported.  The other direction is one-way too: samples written here are plain
  dicts, which the reference's exporters read but its ``load_pt_dataset`` skips (it keeps ``Data`` instances only).
"""
from __future__ import annotations

import glob
import os
from typing import Any, List, Optional

import torch

from .data import Data

__all__ = ["load_pt_dataset", "pad_data", "unpad_data", "save_data_list_as_pt"]


def _to_graph(obj: Any) -> Optional[Data]:
    if isinstance(obj, Data):
        return obj
    if isinstance(obj, dict):
        return Data(**obj) if "x" in obj else None
    if isinstance(obj, (tuple, list)):          # (graph, label) pairs: the graph comes first
        return _to_graph(obj[0]) if len(obj) else None
    if hasattr(obj, "x") and hasattr(obj, "edge_index"):   # attribute bag of another library
        extra = {"node_depth": obj.node_depth} if getattr(obj, "node_depth", None) is not None else {}
        return Data(x=obj.x, edge_index=obj.edge_index, edge_attr=getattr(obj, "edge_attr", None), **extra)
    return None


def _read(path: str, allow_pickle: bool = False):
    try:
        return torch.load(path, weights_only=True)
    except Exception:
        if not allow_pickle:
            raise
        import torch_geometric  # noqa: F401 - unpickling a PyG object needs its classes
        return torch.load(path, weights_only=False)


def load_pt_dataset(dataset_dir: str, max_files: Optional[int] = None, allow_pickle: bool = False) -> List[Data]:
    """Graphs of every ``*.pt`` under ``dataset_dir`` in sorted file order; unreadable files are
    reported and skipped.  ``allow_pickle``: also read files that only a full unpickle can read (trusted
    files only: unpickling runs code from the file)."""
    graphs: List[Data] = []
    for path in sorted(glob.glob(os.path.join(dataset_dir, "*.pt")))[:max_files]:
        try:
            g = _to_graph(_read(path, allow_pickle))
        except Exception as exc:
            print(f"[WARN] failed loading {path}: {exc}")
            continue
        if g is not None:
            graphs.append(g)
    return graphs


def _edge_count(g: Data) -> int:
    ei = g.edge_index
    return int(ei.size(1)) if ei is not None and ei.numel() > 0 else 0


def pad_data(data: Data, max_nodes: int, max_edges: int) -> Data:
    x = data.x
    n, e = int(x.size(0)), _edge_count(data)
    kept = min(e, max_edges)
    i64 = dict(dtype=torch.long, device=x.device)

    x_pad = x.new_zeros((max_nodes, x.size(1)))
    x_pad[:n] = x

    depth = torch.full((max_nodes,), -1, **i64)
    src_depth = getattr(data, "node_depth", None)
    if src_depth is None:
        depth[:n] = 0
    else:
        depth[:src_depth.size(0)] = src_depth        # implicit cast to int64

    ei = torch.full((2, max_edges), -1, **i64)
    ea = torch.zeros((max_edges, 1), **i64)
    if kept:
        ei[:, :kept] = data.edge_index[:, :kept]
        ea[:kept] = data.edge_attr[:kept]

    out = Data(x=x_pad, edge_index=ei, edge_attr=ea, node_depth=depth)
    out._orig_n, out._orig_e = n, e
    return out


def unpad_data(padded: Data) -> Data:
    n = getattr(padded, "_orig_n", padded.x.size(0))
    e = getattr(padded, "_orig_e", padded.edge_index.numel() // 2)
    if e > 0:
        ei, ea = padded.edge_index[:, :e], padded.edge_attr[:e]
    else:
        ei, ea = torch.zeros((2, 0), dtype=torch.long), torch.zeros((0, 1), dtype=torch.long)
    depth = padded.node_depth[:n] if hasattr(padded, "node_depth") else None
    return Data(x=padded.x[:n], edge_index=ei, edge_attr=ea, node_depth=depth)


def save_data_list_as_pt(data_list: List[Data], out_dir: str, prefix: str = "sample", start_idx: int = 0) -> None:
    os.makedirs(out_dir, exist_ok=True)
    for k, g in enumerate(data_list, start=start_idx):
        payload = {name: (val.cpu() if torch.is_tensor(val) else val) for name, val in g.to_dict().items()}
        torch.save(payload, os.path.join(out_dir, f"{prefix}_{k:06d}.pt"))
