"""The library's stable pair sort (aig_sort_pairs, behind every edge list by target / by source, level plan and the
repair's pair keys) against numpy's stable argsort: sizes on both sides of the hand-over to the library sort, one to
four 9-bit passes, 4- and 8-byte keys, keys with bits above the sorted range, heavy duplicates."""
import ctypes as C

import numpy as np
import pytest
import torch

from gan_amd import _abi

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def run_sort(keys: np.ndarray, bits: int):
    lib = _abi.lib()
    kb = keys.dtype.itemsize
    n = keys.size
    # (torch has no unsigned 64-bit arithmetic; the bytes travel as signed)
    kin = torch.from_numpy(keys.view(np.int64 if kb == 8 else np.int32).copy()).to(DEV)
    vin = torch.arange(n, dtype=torch.int32, device=DEV)
    kout, vout = torch.empty_like(kin), torch.empty_like(vin)
    ws = _abi.workspace(lib.aig_sort_pairs_ws_bytes(n, kb, bits), DEV)
    _abi.check(lib.aig_sort_pairs(n, kb, bits, _abi.ptr(kin), _abi.ptr(kout), _abi.ptr(vin), _abi.ptr(vout), _abi.ptr(ws),
                                  ws.numel(), _abi.stream()), "aig_sort_pairs")
    torch.cuda.synchronize()
    assert torch.equal(kin.cpu(), torch.from_numpy(keys.view(np.int64 if kb == 8 else np.int32)))     # inputs untouched
    return kout.cpu().numpy().view(keys.dtype), vout.cpu().numpy()


@pytest.mark.parametrize("kb", [4, 8])
@pytest.mark.parametrize("n,bits", [(1, 5), (1000, 17), (1025, 9), (1025, 10), (4097, 17), (20000, 3), (65536, 18),
                                    (200001, 17), (262144, 27), (262145, 20), (524288, 19), (524289, 19), (1048576, 17),
                                    (1048577, 17), (300000, 28), (150000, 0)])
def test_sort_pairs_matches_stable_argsort(n, bits, kb):
    rng = np.random.default_rng(n + bits)
    dt = np.uint64 if kb == 8 else np.uint32
    hi = (1 << min(bits + 3, 8 * kb)) - 1            # a few bits above the sorted range: they must not matter
    keys = rng.integers(0, hi, size=n, dtype=dt, endpoint=True)
    mask = dt((1 << bits) - 1)
    order = np.argsort(keys & mask, kind="stable")
    got_k, got_v = run_sort(keys, bits)
    assert np.array_equal(got_v, order.astype(np.int32))
    assert np.array_equal(got_k, keys[order])


@pytest.mark.parametrize("n,bits", [(200000, 34), (400000, 36), (50000, 40)])
def test_sort_pairs_wide_keys(n, bits):
    rng = np.random.default_rng(bits)
    keys = rng.integers(0, (1 << bits) - 1, size=n, dtype=np.uint64, endpoint=True)
    order = np.argsort(keys, kind="stable")
    got_k, got_v = run_sort(keys, bits)
    assert np.array_equal(got_v, order.astype(np.int32)) and np.array_equal(got_k, keys[order])


def test_sort_pairs_heavy_duplicates():
    """Most keys equal (an untrained generator hangs most edges on two sources), a few unique ones in between."""
    n = 300000
    rng = np.random.default_rng(3)
    keys = np.where(rng.random(n) < 0.85, rng.integers(0, 2, size=n) * 77777, rng.integers(0, 100000, size=n)).astype(np.uint32)
    order = np.argsort(keys, kind="stable")
    got_k, got_v = run_sort(keys, 17)
    assert np.array_equal(got_v, order.astype(np.int32)) and np.array_equal(got_k, keys[order])


def test_sort_pairs_rejects_bad_arguments():
    lib = _abi.lib()
    k = torch.zeros(8, dtype=torch.int32, device=DEV)
    v = torch.zeros(8, dtype=torch.int32, device=DEV)
    ws = _abi.workspace(1024, DEV)
    assert lib.aig_sort_pairs(8, 2, 8, _abi.ptr(k), _abi.ptr(k), _abi.ptr(v), _abi.ptr(v), _abi.ptr(ws), ws.numel(), _abi.stream()) != 0
    assert lib.aig_sort_pairs(8, 4, 8, _abi.ptr(k), _abi.ptr(k), _abi.ptr(v), _abi.ptr(v), _abi.ptr(ws), ws.numel(), _abi.stream()) != 0
    assert lib.aig_sort_pairs(8, 4, 40, _abi.ptr(k), _abi.ptr(v), _abi.ptr(v), _abi.ptr(k), _abi.ptr(ws), ws.numel(), _abi.stream()) != 0
