"""ctypes binding of libaig_hip.so (declarations: include/aig_abi.h).

There is no CPU fallback: if the shared library is missing or a tensor is not on a
HIP device, the call fails with an error instead of computing somewhere else.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libaig_hip.so")

P = C.c_void_p
I64 = C.c_int64
SZ = C.c_size_t
INT = C.c_int


class GenWeights(C.Structure):
    """Mirror of ``aig_gen_weights``."""
    _fields_ = [(n, P) for n in (
        "enc0_w", "enc0_b", "enc2_w", "enc2_b", "proj0_w", "proj0_b", "proj2_w", "proj2_b",
        "sc0_w", "sc0_b", "sc2_w", "sc2_b", "at0_w", "at0_b", "at2_w", "at2_b")] + [("z_dim", C.c_int32)]


class GenWireSet(C.Structure):
    """Mirror of ``aig_gen_wire_set``."""
    _fields_ = ([(n, I64) for n in ("B", "n", "K", "ecap", "plan_shared", "x_shared")] +
                [(n, P) for n in ("x", "order", "rank", "want", "cnt", "fb", "eoff", "cand", "proj")] +
                [("proj_rows", I64)] +
                [(n, P) for n in ("sc2_w", "sc2_b", "at2_w", "at2_b", "logits", "sel", "edge_index", "edge_attr",
                                  "ecount", "x_out")] +
                [("proj_half", I64), ("cls", P), ("classes", I64), ("pair_logit", P), ("pair_bit", P)])


# name -> (restype, argtypes); every symbol of include/aig_abi.h appears here
SIGNATURES = {
    "aig_abi_version": (INT, []),
    "aig_last_error": (C.c_char_p, []),
    "aig_gen_order_ws_bytes": (SZ, [I64]),
    "aig_gen_order": (INT, [I64, P, INT, P, P, P, P, P, SZ, P]),
    "aig_gen_candidates_ws_bytes": (SZ, [I64, I64]),
    "aig_gen_candidates": (INT, [I64, I64, P, P, P, P, P, P, P, P, P, SZ, P]),
    "aig_gen_node_proj_ws_bytes": (SZ, [I64]),
    "aig_gen_node_proj": (INT, [I64, P, P, C.POINTER(GenWeights), P, P, P, SZ, P]),
    "aig_gen_node_proj_bf16_ws_bytes": (SZ, [I64]),
    "aig_gen_node_proj_bf16": (INT, [I64, I64, P, P, C.POINTER(GenWeights), P, P, P, SZ, P]),
    "aig_gen_score": (INT, [I64, I64, P, P, P, P, P, P, P, P, P, P]),
    "aig_gen_score_half": (INT, [I64, I64, P, P, P, P, P, P, P, P, P, P]),
    "aig_gen_emit_half": (INT, [I64, I64, I64, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "aig_topk_select_ws_bytes": (SZ, [I64, I64]),
    "aig_topk_select": (INT, [I64, I64, P, P, P, P, P, SZ, P]),
    "aig_gen_emit": (INT, [I64, I64, I64, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "aig_gat_csr_ws_bytes": (SZ, [I64, I64]),
    "aig_gat_csr": (INT, [I64, I64] + [P] * 11 + [P, SZ, P]),
    "aig_gat_node_alpha": (INT, [I64, P, P, P, P, P, P, P, P, P]),
    "aig_gat_fwd": (INT, [I64] + [P] * 12 + [P]),
    "aig_gat_fwd_clearing": (INT, [I64] + [P] * 12 + [P, SZ, P]),
    "aig_gat_bwd_ws_bytes": (SZ, [I64, I64]),
    "aig_gat_bwd": (INT, [I64, I64] + [P] * 27 + [P, SZ, P]),
    "aig_gat_bwd_pooled": (INT, [I64, I64, I64] + [P] * 31 + [P, SZ, P]),
    "aig_gat1_fwd": (INT, [I64] + [P] * 15 + [P]),
    "aig_gat1_bwd_ws_bytes": (SZ, [I64]),
    "aig_gat1_bwd": (INT, [I64] + [P] * 22 + [P, SZ, P]),
    "aig_level_pool_bwd": (INT, [I64, I64, P, P, P, P, P, P, P]),
    "aig_level_pool_bwd_adj_ws_bytes": (SZ, [I64]),
    "aig_level_pool_bwd_adj": (INT, [I64, I64, P, P, P, P, P, P, P, SZ, P]),
    "aig_gp_rows_ws_bytes": (SZ, [I64, I64]),
    "aig_gp_rows": (INT, [I64, I64, I64] + [P] * 12 + [P, SZ, P]),
    "aig_gp_forward": (INT, [I64, P] + [P] * 10 + [P]),
    "aig_gp_forward_clearing": (INT, [I64, P] + [P] * 10 + [P, SZ, P]),
    "aig_gp_scalars": (INT, [P, C.c_float, P, P]),
    "aig_adam_chunk_elems": (INT, []),
    "aig_adam_sum3": (INT, [I64, P, I64, P, C.c_float, C.c_double, C.c_double, C.c_double, C.c_double, C.c_float,
                            C.c_float, INT, P]),
    "aig_step_metrics": (INT, [I64, P, P, P, P, P, P, C.c_float, C.c_float, C.c_float, P, P]),
    "aig_sort_pairs_ws_bytes": (SZ, [I64, INT, INT]),
    "aig_sort_pairs": (INT, [I64, INT, INT, P, P, P, P, P, SZ, P]),
    "aig_gp_input_grad_ws_bytes": (SZ, [I64]),
    "aig_gp_input_grad": (INT, [I64, P, I64] + [P] * 14 + [P, SZ, P]),
    "aig_gp_param_grad_ws_bytes": (SZ, [I64]),
    "aig_gp_param_grad": (INT, [I64, P, I64] + [P] * 20 + [P, SZ, P]),
    "aig_gp_param_grad_scaled": (INT, [I64, P, I64] + [P] * 20 + [C.c_float, P, SZ, P]),
    "aig_gp_input_grad_scalars": (INT, [I64, P, I64] + [P] * 14 + [C.c_float, P, P, SZ, P]),
    "aig_linear128": (INT, [I64, P, P, INT, P, INT, P, P]),
    "aig_linear128_node_terms": (INT, [I64, P, P, INT, P, P, P, P, P, P, P, P, P]),
    "aig_linear_in2": (INT, [I64, P, P, P, P]),
    "aig_head_fwd": (INT, [P] * 9 + [P]),
    "aig_head_bwd": (INT, [P] * 13 + [P]),
    "aig_head_input_grad_wgrad": (INT, [P] * 8 + [P]),
    "aig_head_input_grad_wgrad_scaled": (INT, [P] * 8 + [C.c_float, P]),
    "aig_wgrad_ws_bytes": (SZ, [I64, I64]),
    "aig_wgrad": (INT, [I64, I64, P, P, P, P, SZ, P]),
    "aig_level_plan_ws_bytes": (SZ, [I64, I64]),
    "aig_level_plan": (INT, [I64, I64, P, P, P, P, P, P, SZ, P]),
    "aig_level_plan_dyn": (INT, [I64, I64, P, P, P, P, P, P, P, SZ, P]),
    "aig_level_pool_fwd_ws_bytes": (SZ, [I64, I64]),
    "aig_level_pool_fwd": (INT, [I64, I64, I64, P, P, P, P, P, P, P, P, P, SZ, P]),
    "aig_level_pool_table_bytes": (SZ, [I64]),
    "aig_level_pool_fwd_cleared": (INT, [I64, I64, I64, P, P, P, P, P, P, P, P, P, SZ, P]),
    "aig_level_pool_fwd_noalloc": (INT, [I64, I64, I64, P, P, P, P, P, P, P, P, P]),
    "aig_enforce_ws_bytes": (SZ, [I64, I64]),
    "aig_enforce": (INT, [I64, I64, P, P, P, INT, P, P, I64, P, P, P, P, P, SZ, P]),
    "aig_penalties_ws_bytes": (SZ, [I64, I64]),
    "aig_penalties": (INT, [I64, I64, P, P, P, P, SZ, P]),
    "aig_validate_ws_bytes": (SZ, [I64, I64]),
    "aig_validate": (INT, [I64, I64, P, P, P, P, P, P, P, P, P, SZ, P]),
    "aig_resync_ws_bytes": (SZ, [I64, I64]),
    "aig_resync": (INT, [I64, I64, P, P, P, P, P, SZ, P]),
    "aig_resync_typed_ws_bytes": (SZ, [I64, I64]),
    "aig_resync_typed": (INT, [I64, I64, P, P, P, P, P, P, SZ, P]),
    "aig_kernel_timing": (INT, [C.c_uint]),
    "aig_kernel_timing_read": (INT, [INT, P, P]),
    "aig_kernel_timing_name": (C.c_char_p, [INT]),
    "aig_gen_batch_rows": (I64, [I64]),
    "aig_gen_node_proj_batch_ws_bytes": (SZ, [I64, I64, INT]),
    "aig_gen_node_proj_batch": (INT, [I64, I64, P, INT, P, P, INT, P, P, SZ, P]),
    "aig_gen_wire_batch": (INT, [P, P]),
    "aig_enforce_batch_ws_bytes": (SZ, [I64, I64, I64]),
    "aig_enforce_batch": (INT, [I64, I64, I64, P, P, P, P, INT, P, I64, P, I64, P, P, P, P, P, SZ, P]),
    "aig_clean_depth_batch": (INT, [I64, I64, I64, P, P, P, P, P, SZ, P]),
    "aig_penalties_batch_ws_bytes": (SZ, [I64, I64, I64]),
    "aig_penalties_batch": (INT, [I64, I64, I64, P, P, P, P, P, SZ, P]),
    "aig_penalties_batch_hinted": (INT, [I64, I64, I64, P, P, P, P, I64, P, P, SZ, P]),
    "aig_resync_batch_ws_bytes": (SZ, [I64, I64, I64]),
    "aig_resync_batch": (INT, [I64, I64, I64, P, P, P, P, P, P, SZ, P]),
    "aig_resync_typed_batch_ws_bytes": (SZ, [I64, I64, I64]),
    "aig_resync_typed_batch": (INT, [I64, I64, I64, P, P, P, P, P, P, P, SZ, P]),
    "aig_pad_batch": (INT, [I64, I64, I64, P, P, P, I64, P, P, P]),
}

_lib = None


class AigAbiError(RuntimeError):
    pass


def lib():
    """The loaded library; raises if it has not been built (``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AigAbiError(
                f"{LIB_PATH} is missing: the HIP kernels are not built and gan_amd has no other compute path")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class _Timeline:
    """Optional per-entry-point GPU timing with stream events (used by bench.py)."""

    def __init__(self):
        self.enabled = False
        self.events = {}
        self.meta = {}
        self.only = None
        self.sizes_of = None     # names whose problem sizes (``meta``) are noted without any event

    def reset(self):
        self.events = {}
        self.meta = {}

    def totals_ms(self):
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in self.events.items()}


timeline = _Timeline()


def timed(name: str, fn, *args, meta=None):
    """Call an entry point; when the timeline is on, bracket it with events on the current stream
    (the stream the kernels are enqueued on) and remember ``meta`` (problem sizes) per launch.
    ``timeline.only`` (a set of names, or None) restricts the bracketing: an event pair costs tens
    of microseconds of queue time, so a timed run keeps only the entry point it reports on."""
    if timeline.sizes_of is not None and name in timeline.sizes_of and meta is not None:
        timeline.meta.setdefault(name, []).append(meta)
        return fn(*args)
    if not timeline.enabled or (timeline.only is not None and name not in timeline.only):
        return fn(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    rc = fn(*args)
    b.record()
    timeline.events.setdefault(name, []).append((a, b))
    if meta is not None:
        timeline.meta.setdefault(name, []).append(meta)
    return rc


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().aig_last_error().decode("utf-8", "replace")
        raise AigAbiError(f"{what or 'libaig_hip'} failed ({rc}): {msg}")


def ptr(t: Optional[torch.Tensor]):
    """Device address of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise AigAbiError("gan_amd kernels take tensors on a HIP device; got a CPU tensor")
    if not t.is_contiguous():
        raise AigAbiError("gan_amd kernels take contiguous tensors")
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream() -> C.c_void_p:
    """The current stream of the current device as the ``hipStream_t`` the entry points take.  Asked for once per
    entry-point call - a dozen and more times per training step - so it goes through torch's two C calls (current
    device, its current raw stream: well under a microsecond) where they exist; building a ``torch.cuda.Stream``
    object to read its handle costs ~9 us a call."""
    if _raw_stream is not None and _cur_device is not None:
        try:
            return C.c_void_p(_raw_stream(_cur_device()))
        except Exception:       # (no device: the public call below says so in torch's own words)
            pass
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
