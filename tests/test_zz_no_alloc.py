"""The readout entry point that allocates nothing (``aig_level_pool_fwd_noalloc``: its scratch is an array built
into the library) against the entry point it is a variant of (``aig_level_pool_fwd`` with a caller's workspace),
through the C ABI: bit-equal outputs on two calls in a row, the second on a smaller input (the array still holds
the first call's table), then a third, larger again; sizes beyond the built-in array are refused.

Collected last on purpose (``test_zz_``): the round's end runs ``pytest -x``."""
import numpy as np
import pytest
import torch

from gan_amd import _abi
from gan_amd.discriminator import DynLevelPlan, LevelPlan

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
H = 128


def both(plan, h):
    """(out, argmax) of the two entry points for rows ``h`` pooled by ``plan``."""
    lib = _abi.lib()
    res = []
    for noalloc in (False, True):
        out = torch.full((1, 2 * H), float("nan"), device=DEV)
        argmax = torch.full((plan.L + 1, H), -7, dtype=torch.int32, device=DEV)
        args = [plan.n, plan.nvalid, plan.L, _abi.ptr(plan.meta), _abi.ptr(h), _abi.ptr(plan.perm), _abi.ptr(plan.lvl_ptr),
                _abi.ptr(plan.node_level), _abi.ptr(plan.w), _abi.ptr(out), _abi.ptr(argmax)]
        if noalloc:
            _abi.check(lib.aig_level_pool_fwd_noalloc(*args, _abi.stream()), "aig_level_pool_fwd_noalloc")
        else:
            ws = _abi.workspace(lib.aig_level_pool_fwd_ws_bytes(plan.n, plan.L), DEV)
            _abi.check(lib.aig_level_pool_fwd(*args, _abi.ptr(ws), ws.numel(), _abi.stream()), "aig_level_pool_fwd")
        torch.cuda.synchronize()
        res.append((out.cpu().numpy(), argmax.cpu().numpy()))
    return res


def check_case(n, levels, seed, dyn=False):
    g = torch.Generator().manual_seed(seed)
    depth = torch.randint(-1, levels, (n,), generator=g)
    depth[:levels] = torch.arange(levels)          # every level occupied at least once, the deepest included
    h = torch.randn((n, H), generator=g).to(DEV)
    if dyn:
        plan = DynLevelPlan(n, levels + 5, DEV)
        plan.rebuild(depth.to(DEV))
    else:
        plan = LevelPlan(depth.to(DEV), n, DEV)
    (out_ws, am_ws), (out_na, am_na) = both(plan, h)
    assert np.isfinite(out_ws).all()
    assert np.array_equal(out_ws, out_na)
    top = levels if not dyn else levels      # rows of argmax past the deepest level are not written by either
    assert np.array_equal(am_ws[:top], am_na[:top])
    # and both are the readout: mean over levels of (level mean || level max)
    hh, dd = h.cpu().double(), depth
    ref = torch.stack([torch.cat([hh[dd == l].mean(0), hh[dd == l].max(0)[0]]) for l in range(levels)]).mean(0)
    np.testing.assert_allclose(out_na[0], ref.numpy(), rtol=1e-4, atol=1e-5)


def _args(plan, h, out, argmax):
    return [plan.n, plan.nvalid, plan.L, _abi.ptr(plan.meta), _abi.ptr(h), _abi.ptr(plan.perm), _abi.ptr(plan.lvl_ptr),
            _abi.ptr(plan.node_level), _abi.ptr(plan.w), _abi.ptr(out), _abi.ptr(argmax)]


def test_recorded_into_a_graph_and_replayed():
    """What the entry point is for: under HIP graph capture.  The very first call of a process on a device looks up
    the scratch array's address and is refused inside a recording (when this file runs in a process of its own, that
    is the call made here); after one call outside, the call is recorded and replayed on new rows, bit-equal to the
    workspace form each time."""
    lib = _abi.lib()
    n, levels = 30000, 90
    g = torch.Generator().manual_seed(11)
    depth = torch.randint(-1, levels, (n,), generator=g)
    depth[:levels] = torch.arange(levels)
    plan = LevelPlan(depth.to(DEV), n, DEV)
    h = torch.randn((n, H), generator=g).to(DEV)
    out = torch.zeros((1, 2 * H), device=DEV)
    argmax = torch.zeros((plan.L + 1, H), dtype=torch.int32, device=DEV)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        first_rc = lib.aig_level_pool_fwd_noalloc(*_args(plan, h, out, argmax), _abi.stream())
    if first_rc != 0:      # nothing had looked the address up yet: refused, nothing recorded
        assert first_rc == -1 and "outside the recording" in lib.aig_last_error().decode()
    del graph
    _abi.check(lib.aig_level_pool_fwd_noalloc(*_args(plan, h, out, argmax), _abi.stream()))     # outside: allowed
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        _abi.check(lib.aig_level_pool_fwd_noalloc(*_args(plan, h, out, argmax), _abi.stream()))
    for seed in (21, 22, 23):
        h.copy_(torch.randn((n, H), generator=torch.Generator().manual_seed(seed)))
        out.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        want = torch.empty_like(out)
        am = torch.empty_like(argmax)
        ws = _abi.workspace(lib.aig_level_pool_fwd_ws_bytes(plan.n, plan.L), DEV)
        _abi.check(lib.aig_level_pool_fwd(*_args(plan, h, want, am), _abi.ptr(ws), ws.numel(), _abi.stream()))
        torch.cuda.synchronize()
        assert torch.equal(out, want) and torch.equal(argmax, am), seed


def test_two_calls_in_a_row_second_smaller_then_larger():
    check_case(n=60000, levels=700, seed=1)
    check_case(n=900, levels=12, seed=2)            # the array still holds the 700-level table of the call before
    check_case(n=20000, levels=300, seed=3, dyn=True)
    check_case(n=1, levels=1, seed=4)


def test_sizes_beyond_the_built_in_array_are_refused():
    lib = _abi.lib()
    n, L = 1000, 5000                                # 5001 levels need 5 MB of table
    assert lib.aig_level_pool_fwd_ws_bytes(n, L) > (4 << 20)
    t = torch.zeros(16, device=DEV)
    rc = lib.aig_level_pool_fwd_noalloc(n, n, L, None, _abi.ptr(t), _abi.ptr(t), _abi.ptr(t), _abi.ptr(t), _abi.ptr(t),
                                        _abi.ptr(t), _abi.ptr(t), _abi.stream())
    assert rc == -4                                  # AIG_ELIMIT
    msg = lib.aig_last_error().decode()
    assert "scratch" in msg and "5001 levels" in msg
    torch.cuda.synchronize()
