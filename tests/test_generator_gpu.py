"""GPU parity of the generator path (all calls go through the C ABI of libaig_hip.so).

Bars: order, candidate windows, picks, edges and x bit-exact against the reference's own
outputs (tests/golden) and against the CPU checker; h_cond / logits within 1e-4 relative.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_names
from gan_amd import _abi
from gan_amd.data import Data
from gan_amd.generator import AIGGenerator, GeneratorPlan
from gan_amd.synth import synth_aig
from oracle import cpu as ocpu
from oracle import model as omodel

pytestmark = pytest.mark.gpu
DEV = "cuda"


from util_golden import gen_state, golden_targets, load, tie_consistent


def golden_generator(K):
    G = AIGGenerator(candidate_k=K)
    G.load_state_dict(gen_state())
    return G.to(DEV).eval()


def cpu_state():
    return gen_state()


GEN_CASES = [c for c in golden_names("gen_") if c != "weights"]


@pytest.mark.parametrize("case", GEN_CASES)
def test_generator_against_reference_vectors(case):
    """Against the reference's own outputs: windows exact, scores within 1e-4 relative, picks exact
    when the select kernel is fed the reference's scores, and exact end to end wherever the
    reference's scores are a function of the candidate features (see util_golden.tie_consistent)."""
    d = load("gen_" + case)
    K = int(d["candidate_k"])
    G = golden_generator(K)
    tpl = Data(x=torch.from_numpy(d["x"]).to(DEV))
    if "node_depth" in d.files:
        tpl.node_depth = torch.from_numpy(d["node_depth"]).to(DEV)
    z = torch.from_numpy(d["z"]).to(DEV)
    out = G(tpl, z, return_debug=True)
    torch.cuda.synchronize()
    assert out.edge_index.dtype == torch.int64 and out.edge_attr.dtype == torch.int64
    assert out.edge_index.shape == tuple(d["out_edge_index"].shape)
    assert np.array_equal(out.edge_index[1].cpu().numpy(), d["out_edge_index"][1])
    if "node_depth" in d.files:
        assert np.array_equal(out.node_depth.cpu().numpy(), d["node_depth"])
    plan = out._debug["plan"]
    if "h_cond" in d.files:
        np.testing.assert_allclose(out._debug["h_cond"].cpu().numpy()[: d["x"].shape[0]], d["h_cond"],
                                   rtol=1e-4, atol=1e-5)
    rank = plan.rank.cpu().numpy()
    cand = plan.cand.cpu().numpy()
    cnt = plan.cnt.cpu().numpy()
    want = plan.want.cpu().numpy()
    logits = out._debug["logits"].cpu().numpy()
    sel = out._debug["sel"].cpu().numpy()
    targets = golden_targets(d)
    ref_rows = np.zeros((len(targets), K), np.float32)
    ref_cnt, ref_want, ref_picks = [], [], []
    all_consistent = True
    for t, (v, ref_c, ref_l, picks) in enumerate(targets):
        r = rank[v]
        assert cnt[r] == len(ref_c)
        assert np.array_equal(cand[r, :cnt[r]], ref_c)
        np.testing.assert_allclose(logits[r, :cnt[r]], ref_l, rtol=1e-4, atol=1e-6)
        ref_rows[t, :len(ref_l)] = ref_l
        ref_cnt.append(len(ref_c))
        ref_want.append(want[r])
        ref_picks.append(picks + [-1] * (2 - len(picks)))
        if tie_consistent(d["x"], ref_c, ref_l):
            assert sel[r, :len(picks)].tolist() == picks, (case, v)
        else:
            all_consistent = False
    got = run_select(ref_rows, np.asarray(ref_cnt), np.asarray(ref_want))
    assert np.array_equal(got, np.asarray(ref_picks))
    if all_consistent:
        assert np.array_equal(out.edge_index.cpu().numpy(), d["out_edge_index"])
        assert np.array_equal(out.edge_attr.cpu().numpy(), d["out_edge_attr"])
        assert np.array_equal(out.x.cpu().numpy(), d["out_x"])


@pytest.mark.parametrize("case", GEN_CASES)
def test_generator_bit_exact_against_checker(case):
    """Against the CPU checker's fixed-order form: every tensor identical, scores bit for bit."""
    d = load("gen_" + case)
    K = int(d["candidate_k"])
    G = golden_generator(K)
    x, z = torch.from_numpy(d["x"]), torch.from_numpy(d["z"])
    nd = torch.from_numpy(d["node_depth"]) if "node_depth" in d.files else None
    tpl = Data(x=x.to(DEV))
    if nd is not None:
        tpl.node_depth = nd.to(DEV)
    out = G(tpl, z.to(DEV), return_debug=True)
    ref = omodel.generator_forward_canonical(cpu_state(), x, nd, z, K)
    n = x.size(0)
    assert np.array_equal(out._debug["h_cond"].cpu().numpy()[:n].view(np.int32), ref["h_cond"].view(np.int32))
    assert np.array_equal(out._debug["proj"].cpu().numpy()[:, :n].view(np.int32), ref["proj"].view(np.int32))
    rank = out._debug["plan"].rank.cpu().numpy()
    logits = out._debug["logits"].cpu().numpy()
    for t, v in enumerate(ref["tgt"]):
        c = int(ref["cnt"][t])
        assert np.array_equal(logits[rank[v], :c].view(np.int32), ref["logits"][t, :c].view(np.int32))
    assert np.array_equal(out.edge_index.cpu().numpy(), ref["edge_index"].numpy())
    assert np.array_equal(out.edge_attr.cpu().numpy(), ref["edge_attr"].numpy())
    assert np.array_equal(out.x.cpu().numpy(), ref["x"].numpy())


@pytest.mark.parametrize("n", [1, 31, 32, 33, 64, 65, 97, 4097, 8192 + 33, 256 * 96, 256 * 160 + 7])
def test_node_projection_row_runs_bit_exact(n):
    """The node pipeline deals rows to workgroups in runs of whole and half tiles and streams each
    output to memory from the product loop of the next: every run shape (one half tile, whole + half,
    several whole + half, a ragged last run) against the CPU checker, bit for bit."""
    import ctypes as C
    G = golden_generator(8)
    lib = _abi.lib()
    gen = torch.Generator().manual_seed(n)
    x = torch.stack([torch.randint(0, 3, (n,), generator=gen).float(), torch.randint(0, 3, (n,), generator=gen).float()], 1)
    z = torch.randn(128, generator=gen)
    xd, zd = x.to(DEV).contiguous(), z.to(DEV)
    proj = torch.full((4, n, 128), float("nan"), device=DEV)
    hc = torch.full((n, 128), float("nan"), device=DEV)
    guard = torch.full((4096,), 7.0, device=DEV)   # allocated right after: rows past n must not be written
    ws = _abi.workspace(lib.aig_gen_node_proj_ws_bytes(n), DEV)
    w = G._weights()
    _abi.check(lib.aig_gen_node_proj(n, _abi.ptr(xd), _abi.ptr(zd), C.byref(w), _abi.ptr(hc), _abi.ptr(proj), _abi.ptr(ws),
                                     ws.numel(), _abi.stream()), "aig_gen_node_proj")
    torch.cuda.synchronize()
    ref_h, ref_p = ocpu.node_proj(cpu_state(), x.numpy(), z.numpy())
    assert np.array_equal(hc.cpu().numpy().view(np.int32), ref_h.view(np.int32))
    assert np.array_equal(proj.cpu().numpy().view(np.int32), ref_p.view(np.int32))
    assert bool((guard == 7.0).all())


def test_node_projection_rejects_misaligned_rows():
    """The kernels read x in 8-byte and everything else in 16-byte pieces: a pointer that is not aligned
    so is refused up front instead of faulting on the GPU."""
    import ctypes as C
    G = golden_generator(8)
    lib = _abi.lib()
    n = 100
    x = torch.zeros(2 * n + 1, device=DEV)[1:].view(n, 2)   # 4 bytes past an aligned address
    z = torch.zeros(128, device=DEV)
    proj = torch.empty((4, n, 128), device=DEV)
    ws = _abi.workspace(lib.aig_gen_node_proj_ws_bytes(n), DEV)
    w = G._weights()
    with pytest.raises(_abi.AigAbiError):
        _abi.check(lib.aig_gen_node_proj(n, _abi.ptr(x), _abi.ptr(z), C.byref(w), None, _abi.ptr(proj), _abi.ptr(ws),
                                         ws.numel(), _abi.stream()), "aig_gen_node_proj")


def check_plan_against_checker(tpl_cpu: Data, K: int):
    x = tpl_cpu.x
    n = x.size(0)
    nd = getattr(tpl_cpu, "node_depth", None)
    plan = GeneratorPlan.build(x.to(DEV).float().contiguous(), None if nd is None else nd.to(DEV), K)
    torch.cuda.synchronize()
    types = x[:, 0].long().numpy()
    depth = None if nd is None else nd.to(torch.float64).numpy()
    order, rank = ocpu.build_order(types, depth)
    assert np.array_equal(plan.order.cpu().numpy(), order)
    assert np.array_equal(plan.rank.cpu().numpy(), rank)
    assert np.array_equal(plan.types.cpu().numpy(), types)
    tgt, cand, cnt, fb = ocpu.candidates(types, order, K)
    want = plan.want.cpu().numpy()
    slots = np.nonzero(want)[0]
    assert np.array_equal(order[slots], tgt)
    assert np.array_equal(plan.cnt.cpu().numpy()[slots], cnt)
    assert np.array_equal(plan.fb.cpu().numpy()[slots], fb)
    got = plan.cand.cpu().numpy()[slots]
    mask = np.arange(K)[None, :] < cnt[:, None]
    assert np.array_equal(np.where(mask, got, -1), np.where(mask, cand, -1))
    # edge offsets: want[r] edges whenever the slot has a candidate or a fallback source
    ne = np.zeros(n, np.int64)
    ne[slots] = np.where((cnt > 0) | (fb >= 0), want[slots], 0)
    eoff = plan.eoff.cpu().numpy()
    assert np.array_equal(eoff, np.concatenate([[0], np.cumsum(ne)]))
    assert plan.E == int(ne.sum())
    return plan


@pytest.mark.parametrize("K", [1, 3, 64, 65, 200, 512, 1000])
@pytest.mark.parametrize("kind", ["forward", "dataset", "shuffled", "nodepth", "longdepth"])
def test_plan_matches_checker(kind, K):
    t = synth_aig(50, 900, 40, seed=K, window=300, orientation="dataset" if kind == "dataset" else "forward",
                  shuffle_ids=(kind == "shuffled"))
    if kind == "nodepth":
        del t.node_depth
    if kind == "longdepth":
        t.node_depth = t.node_depth.long() - 1
    check_plan_against_checker(t, K)


def test_plan_large_graph_and_limits():
    t = synth_aig(1000, 98000, 1000, seed=0, window=4096)
    check_plan_against_checker(t, 64)
    with pytest.raises(_abi.AigAbiError):
        GeneratorPlan.build(t.x.to(DEV), None, 5000)  # beyond AIG_MAX_CANDIDATE_K
    # empty template
    e = GeneratorPlan.build(torch.zeros((0, 2), device=DEV), None, 8)
    assert e.E == 0


def run_select(values, cnt, want):
    lib = _abi.lib()
    rows, ld = values.shape
    v = torch.from_numpy(values).to(DEV)
    c = torch.from_numpy(cnt.astype(np.int32)).to(DEV)
    w = torch.from_numpy(want.astype(np.int32)).to(DEV)
    sel = torch.full((rows, 2), -7, dtype=torch.int32, device=DEV)
    _abi.check(lib.aig_topk_select(rows, ld, _abi.ptr(v), _abi.ptr(c), _abi.ptr(w), _abi.ptr(sel), None, 0,
                                   _abi.stream()))
    torch.cuda.synchronize()
    return sel.cpu().numpy()


def test_select_matches_torch_topk_table():
    d = load("topk_kat")
    offs, vals = d["offsets"], d["values"]
    rows = len(offs) - 1  # 2,560 rows: C = 1..512, five tie patterns each
    mat = np.zeros((rows, 512), np.float32)
    cnt = np.diff(offs)
    for r in range(rows):
        mat[r, :cnt[r]] = vals[offs[r]:offs[r + 1]]
    s1 = run_select(mat, cnt, np.ones(rows))
    assert np.array_equal(s1[:, 0], d["top1"])
    assert (s1[:, 1] == -1).all()
    s2 = run_select(mat, cnt, np.full(rows, 2))
    two = cnt >= 2
    assert np.array_equal(s2[two], d["top2"][two])
    # a single candidate asked for two fan-ins: one pick, the caller repeats it
    assert np.array_equal(s2[~two][:, 0], np.zeros((~two).sum())) and (s2[~two][:, 1] == -1).all()


def test_select_matches_checker_on_random_rows():
    rng = np.random.default_rng(3)
    rows, ld = 5000, 1024
    cnt = rng.integers(0, ld + 1, rows)
    cnt[:200] = rng.integers(0, 130, 200)
    want = rng.integers(0, 3, rows)
    mat = np.zeros((rows, ld), np.float32)
    for r in range(rows):
        nlev = int(rng.integers(1, 12))
        lv = rng.standard_normal(nlev).astype(np.float32)
        row = rng.choice(lv, size=cnt[r]) if r % 3 else rng.standard_normal(cnt[r]).astype(np.float32)
        if r % 17 == 0 and cnt[r] > 3:
            row[rng.integers(0, cnt[r], 2)] = np.nan
        if r % 19 == 0 and cnt[r] > 3:
            row[rng.integers(0, cnt[r], 2)] = np.inf
        mat[r, :cnt[r]] = row
    got = run_select(mat, cnt, want)
    ref = ocpu.topk_rows(mat, cnt, want)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("n_and,K", [(3000, 64), (1500, 256)])
def test_generator_matches_checker_beyond_golden_sizes(n_and, K):
    t = synth_aig(150, n_and, 80, seed=n_and, window=700, orientation="dataset")
    z = torch.randn(128, generator=torch.Generator().manual_seed(1))
    G = golden_generator(K)
    out = G(Data(x=t.x.to(DEV), node_depth=t.node_depth.to(DEV)), z.to(DEV))
    ref = omodel.generator_forward_canonical(cpu_state(), t.x, t.node_depth, z, K)
    assert np.array_equal(out.edge_index.cpu().numpy(), ref["edge_index"].numpy())
    assert np.array_equal(out.edge_attr.cpu().numpy(), ref["edge_attr"].numpy())
    assert np.array_equal(out.x.cpu().numpy(), ref["x"].numpy())


def test_generator_full_size_properties():
    """100k-node template (too slow for the CPU checker): structural invariants of the output."""
    t = synth_aig(1000, 98000, 1000, seed=0, window=4096)
    G = golden_generator(64)
    tpl = Data(x=t.x.to(DEV), node_depth=t.node_depth.to(DEV))
    z = torch.randn(128, generator=torch.Generator().manual_seed(5)).to(DEV)
    out = G(tpl, z)
    out2 = G(tpl, z)  # cached plan, same z: identical
    assert torch.equal(out.edge_index, out2.edge_index) and torch.equal(out.edge_attr, out2.edge_attr)
    ei = out.edge_index.cpu().numpy()
    types = t.x[:, 0].long().numpy()
    depth = t.node_depth.numpy()
    assert ei.shape[1] == 2 * 98000 + 1000
    src, dst = ei
    assert (types[src] != 1).all() and (types[dst] != 0).all() and (src != dst).all()
    # sources are processed before their targets: (depth, id) strictly smaller
    assert ((depth[src] < depth[dst]) | ((depth[src] == depth[dst]) & (src < dst))).all()
    indeg = np.bincount(dst, minlength=types.size)
    assert np.array_equal(indeg, np.where(types == 2, 2, np.where(types == 1, 1, 0)))
    inv = out.edge_attr.cpu().numpy().reshape(-1)
    assert set(np.unique(inv)) <= {0, 1}
    assert np.array_equal(out.x.cpu().numpy()[:, 1], np.bincount(dst, weights=inv, minlength=types.size))


def test_generator_requires_gpu_tensors():
    G = golden_generator(8)
    with pytest.raises(_abi.AigAbiError):
        G(Data(x=torch.zeros(4, 2)), torch.zeros(128))


def test_bulk_sampler_matches_one_at_a_time():
    """Samples kept in flight on several streams are the samples of the plain loop, in job order."""
    from gan_amd.sample import sample_many, sample_one
    from gan_amd.synth import synth_aig
    torch.manual_seed(5)
    G = AIGGenerator(candidate_k=16).to(DEV)
    tpls = [synth_aig(8, 300 + 40 * k, 6, seed=k, window=64, orientation="dataset").to(DEV) for k in range(3)]
    zs = [torch.randn(128, device=DEV) for _ in range(7)]
    jobs = [(tpls[i % 3], zs[i]) for i in range(7)]
    want = [sample_one(G, Data(x=t.x.clone(), node_depth=t.node_depth.clone()), z) for t, z in jobs]
    for in_flight in (1, 3, 4):
        got = list(sample_many(G, ((t, (lambda z=z: z)) for t, z in jobs), in_flight=in_flight))
        assert len(got) == len(want)
        for a, b in zip(got, want):
            for k in ("x", "edge_index", "edge_attr", "node_depth"):
                assert torch.equal(getattr(a, k), getattr(b, k)), (in_flight, k)


# ------------------------------------------------------------------------------------------------------
# bf16 scoring option (bulk sampling): node projections on the bf16 matrix cores, planes stored as halfs, pairs
# scored in packed half arithmetic.  Stated tolerance: every logit within BF16_TOL of the f32 path's, relative to
# the largest logit of its row.
BF16_TOL = 3e-2


def bf16_generator(K):
    G = AIGGenerator(candidate_k=K, score_dtype="bf16")
    G.load_state_dict(gen_state())
    return G.to(DEV).eval()


@pytest.mark.parametrize("case", ["n486_dataset_k64", "n700_dataset_k256", "n486_distinct_k64", "n243_padded300_k64",
                                  "pofirst_nodepth_k4"])
def test_bf16_scoring_against_the_f32_path(case):
    d = load("gen_" + case)
    K = int(d["candidate_k"])
    x, z = torch.from_numpy(d["x"]), torch.from_numpy(d["z"])
    tpl = Data(x=x.to(DEV))
    if "node_depth" in d.files:
        tpl.node_depth = torch.from_numpy(d["node_depth"]).to(DEV)
    out32 = golden_generator(K)(tpl, z.to(DEV), return_debug=True)
    out16 = bf16_generator(K)(tpl, z.to(DEV), return_debug=True)
    n = x.size(0)
    # the plan (order, candidate windows, edge offsets) does not depend on the arithmetic: same object contents
    p32, p16 = out32._debug["plan"], out16._debug["plan"]
    for name in ("order", "rank", "want", "cnt", "cand", "fb", "eoff"):
        assert torch.equal(getattr(p32, name), getattr(p16, name)), name
    # projections against the restated bf16 pipeline (exact sums of bf16-rounded inputs)
    h_ref, proj_ref = omodel.node_proj_bf16(gen_state(), x, z)
    h16, proj16h = out16._debug["h_cond"].cpu()[:n], out16._debug["proj"].cpu()[:, :n]
    assert proj16h.dtype == torch.float16
    proj16 = proj16h.float()
    scale = float(proj_ref.abs().max())
    assert float((h16 - h_ref).abs().max()) <= 1e-2 * float(h_ref.abs().max())
    assert float((proj16 - proj_ref).abs().max()) <= 1e-2 * scale
    assert float((proj16 - proj_ref).abs().median()) <= 1e-5 * scale       # all but boundary roundings agree closely
    # equal feature rows give bit-equal projections (ties between equal candidates survive)
    _, inv = np.unique(d["x"], axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    first = {int(g): int(np.flatnonzero(inv == g)[0]) for g in np.unique(inv)}
    rep = torch.tensor([first[int(g)] for g in inv])
    assert torch.equal(proj16h.view(torch.int16), proj16h[:, rep].view(torch.int16))
    # logits within the stated tolerance of the f32 path's; picks identical wherever f32 separates the
    # chosen candidates from the rest (and from each other) by more than twice that
    l32, l16 = out32._debug["logits"].cpu().numpy(), out16._debug["logits"].cpu().numpy()
    s32, s16 = out32._debug["sel"].cpu().numpy(), out16._debug["sel"].cpu().numpy()
    cnt, want = p32.cnt.cpu().numpy(), p32.want.cpu().numpy()
    # the packed-half scoring itself, from the planes the device made: against its restatement
    st = gen_state()
    order_h, cand_h = p32.order.cpu().numpy(), p32.cand.cpu().numpy()
    w2, b2 = st["edge_scorer.2.weight"], float(st["edge_scorer.2.bias"])
    checked = 0
    for r in range(0, n, max(1, n // 40)):
        c = int(cnt[r])
        if c == 0:
            continue
        ref = omodel.score_half(proj16h[0][cand_h[r, :c]], proj16h[1][order_h[r]], w2, b2).numpy()
        got = out16._debug["logits"].cpu().numpy()[r, :c]
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6 * max(float(np.abs(ref).max()), 1.0), err_msg=f"{case} {r}")
        checked += 1
    assert checked > 0
    compared = decided = 0
    for r in range(n):
        c = int(cnt[r])
        if c == 0 or want[r] == 0:
            continue
        a, b = l32[r, :c], l16[r, :c]
        tol = BF16_TOL * max(float(np.abs(a).max()), 1e-6)
        assert float(np.abs(a - b).max()) <= tol, (case, r)
        compared += 1
        k = min(int(want[r]), c)
        top = np.sort(a)[::-1]
        clear = all(top[i] - top[i + 1] > 2 * tol for i in range(min(k, c - 1)))
        if clear:
            assert s16[r, :k].tolist() == s32[r, :k].tolist(), (case, r)
            decided += 1
    assert compared > 0
    # rows of equal candidates tie in bf16 exactly as they do in f32: equal logit bits for equal candidate features
    cand = p32.cand.cpu().numpy()
    for r in range(0, n, max(1, n // 50)):
        c = int(cnt[r])
        if c < 2:
            continue
        groups = inv[cand[r, :c]]
        for gid in np.unique(groups):
            bits = l16[r, :c][groups == gid].view(np.int32)
            assert np.unique(bits).size == 1, (case, r)


def test_bf16_projection_of_several_samples_in_one_call():
    """z of several samples against one template in one launch (every sample starts on a 64-row boundary) = the
    same rows as one call per sample, bit for bit."""
    import ctypes as C
    lib = _abi.lib()
    G = bf16_generator(64)
    n, S = 64 * 5 + 7, 3
    n64 = (n + 63) // 64 * 64
    g = synth_aig(20, n - 30, 10, seed=2, window=40)
    x = g.x.to(DEV).contiguous()
    zs = torch.randn(S, 128, generator=torch.Generator().manual_seed(3)).to(DEV)
    w = G._weights()
    proj = torch.empty((4, S * n64, 128), dtype=torch.float16, device=DEV)
    h = torch.empty((S * n64, 128), device=DEV)
    ws = _abi.workspace(lib.aig_gen_node_proj_bf16_ws_bytes(S), x.device)
    _abi.check(lib.aig_gen_node_proj_bf16(n, S, _abi.ptr(x), _abi.ptr(zs), C.byref(w), _abi.ptr(h), _abi.ptr(proj),
                                          _abi.ptr(ws), ws.numel(), _abi.stream()))
    for i in range(S):
        p1 = torch.empty((4, n, 128), dtype=torch.float16, device=DEV)
        h1 = torch.empty((n, 128), device=DEV)
        ws1 = _abi.workspace(lib.aig_gen_node_proj_bf16_ws_bytes(1), x.device)
        _abi.check(lib.aig_gen_node_proj_bf16(n, 1, _abi.ptr(x), _abi.ptr(zs[i].contiguous()), C.byref(w), _abi.ptr(h1),
                                              _abi.ptr(p1), _abi.ptr(ws1), ws1.numel(), _abi.stream()))
        assert torch.equal(h[i * n64:i * n64 + n], h1)
        assert torch.equal(proj[:, i * n64:i * n64 + n], p1)


def test_bulk_sampling_in_blocks_equals_one_sample_at_a_time():
    from gan_amd.sample import sample_bulk, sample_one
    g = synth_aig(12, 330, 9, seed=5, window=60, orientation="dataset")
    tpl = Data(x=g.x.to(DEV), node_depth=g.node_depth.to(DEV))
    zs = torch.randn(7, 128, generator=torch.Generator().manual_seed(8)).to(DEV)
    for dtype in ("bf16", "fp32"):
        G = AIGGenerator(candidate_k=32, score_dtype=dtype)
        G.load_state_dict(gen_state())
        G = G.to(DEV).eval()
        bulk = list(sample_bulk(G, tpl, [zs[:4], zs[4:]], in_flight=3))
        assert len(bulk) == 7
        for i, d in enumerate(bulk):
            ref = sample_one(G, tpl, zs[i])
            for name in ("x", "edge_index", "edge_attr", "node_depth"):
                assert torch.equal(getattr(d, name), getattr(ref, name)), (dtype, i, name)


@pytest.mark.parametrize("K,half", [(256, False), (256, True), (128, False), (192, True)])
def test_picks_made_while_scoring_equal_the_selection_kernel(K, half):
    """With windows of up to 256 candidates aig_gen_wire_batch lets the wave that scores a slot pick for it when
    torch.topk would run partial_sort (k*64 <= cnt); those picks against aig_topk_select on the logits the same
    call wrote.  The planes are made so that the logits are long runs of equal values, +-inf and NaN: a logit is
    relu(a0) - relu(a1) + 0 * relu(a2), with a2 = inf where a NaN is wanted."""
    import ctypes as C
    lib = _abi.lib()
    g = synth_aig(40, 900, 20, seed=K + half, window=K + 50)
    x = g.x.to(DEV).float().contiguous()
    n = x.size(0)
    G = golden_generator(K)
    plan = G.plan_for(Data(x=x), x, None)
    gen = torch.Generator().manual_seed(K * 2 + half)
    B = 3
    rows = int(lib.aig_gen_batch_rows(n))
    levels = torch.tensor([0.0, 0.0, 0.0, 1.0, 1.0, 2.5, 7.0, float("inf")])
    a0 = levels[torch.randint(0, len(levels), (B * rows,), generator=gen)]
    a1 = levels[torch.randint(0, len(levels), (B * rows,), generator=gen)]
    a1[torch.rand(B * rows, generator=gen) < 0.5] = 0.0
    a2 = torch.where(torch.rand(B * rows, generator=gen) < 0.03, torch.tensor(float("inf")), torch.tensor(0.0))
    a1[a0 == float("inf")] = 0.0                       # (inf - inf is a NaN of its own; keep the two kinds apart)
    dt = torch.float16 if half else torch.float32
    proj = torch.zeros((4, B * rows, 128), dtype=dt)
    proj[0, :, 0], proj[0, :, 1], proj[0, :, 2] = a0.to(dt), a1.to(dt), a2.to(dt)
    proj = proj.to(DEV)
    w2 = torch.zeros((1, 128)); w2[0, 0], w2[0, 1] = 1.0, -1.0
    w2, b2 = w2.to(DEV), torch.zeros(1, device=DEV)
    E = max(plan.E, 1)
    a = _abi.GenWireSet()
    a.B, a.n, a.K, a.ecap, a.plan_shared, a.x_shared = B, n, K, E, 1, 1
    a.x = _abi.ptr(x)
    for name in ("order", "rank", "want", "cnt", "fb", "eoff", "cand"):
        setattr(a, name, _abi.ptr(getattr(plan, name)))
    a.proj, a.proj_rows, a.proj_half = _abi.ptr(proj), rows, int(half)
    a.sc2_w, a.sc2_b, a.at2_w, a.at2_b = _abi.ptr(w2), _abi.ptr(b2), _abi.ptr(w2), _abi.ptr(b2)
    logits = torch.full((B, n, K), float("nan"), device=DEV)
    sel = torch.full((B, n, 2), -9, dtype=torch.int32, device=DEV)
    ei = torch.empty((B, 2, E), dtype=torch.int64, device=DEV)
    ea = torch.empty((B, E), dtype=torch.int64, device=DEV)
    ecount = torch.empty(B, dtype=torch.int32, device=DEV)
    x_out = torch.empty((B * n, 2), device=DEV)
    a.logits, a.sel, a.edge_index, a.edge_attr, a.ecount, a.x_out = (_abi.ptr(t) for t in (logits, sel, ei, ea, ecount, x_out))
    _abi.check(lib.aig_gen_wire_batch(C.byref(a), _abi.stream()), "aig_gen_wire_batch")
    torch.cuda.synchronize()
    cnt, want = plan.cnt.cpu().numpy(), plan.want.cpu().numpy()
    k = np.minimum(want, cnt)
    fused_rows = (k > 0) & (k * 64 <= cnt)
    assert fused_rows.sum() > 100 and (~fused_rows & (k > 0)).sum() > 10       # both kinds of slot occur
    lg = logits.cpu().numpy()
    assert np.isnan(lg[0][fused_rows][:, :64]).any() and np.isinf(lg[0][fused_rows][:, :64]).any()
    for b in range(B):
        alone = torch.full((n, 2), -9, dtype=torch.int32, device=DEV)
        _abi.check(lib.aig_topk_select(n, K, _abi.ptr(logits[b]), _abi.ptr(plan.cnt), _abi.ptr(plan.want), _abi.ptr(alone),
                                       None, 0, _abi.stream()), "aig_topk_select")
        torch.cuda.synchronize()
        assert torch.equal(sel[b], alone), (b, np.flatnonzero((sel[b] != alone).any(1).cpu().numpy())[:10])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_scoring_by_class_equals_scoring_by_node(dtype):
    """Templates whose nodes take few distinct feature rows are projected per class and scored through (source class,
    target class) tables; with ``class_limit = 0`` every node goes through the node pipeline and every pair through
    its own chain.  Same logits, picks, edges and features, bit for bit - one graph, a set of samples of one
    template, and a set of different templates whose class counts differ."""
    from gan_amd.sample import sample_one
    K = 48
    tpls = []
    for seed, extra in ((1, 0), (2, 3), (3, 7)):
        g = synth_aig(15, 400, 10, seed=seed, window=70, orientation="dataset")
        x = g.x.clone().float()
        if extra:                                  # a few more distinct rows: class counts differ between templates
            x[5:5 + extra, 1] = torch.arange(3, 3 + extra).float()
        tpls.append(Data(x=x.to(DEV), node_depth=g.node_depth.to(DEV)))
    zs = torch.randn(3, 128, generator=torch.Generator().manual_seed(4)).to(DEV)

    def make(limit):
        G = AIGGenerator(candidate_k=K, score_dtype=dtype)
        G.load_state_dict(gen_state())
        G = G.to(DEV).eval()
        G.class_limit = limit
        return G

    Gc, Gn = make(1024), make(0)
    counts = []
    for t, z in zip(tpls, zs):
        a, b = Gc(t, z, return_debug=True), Gn(t, z, return_debug=True)
        counts.append(a._debug["plan"].C)
        for name in ("x", "edge_index", "edge_attr"):
            assert torch.equal(getattr(a, name), getattr(b, name)), name
        la, lb = a._debug["logits"], b._debug["logits"]
        assert torch.equal(torch.nan_to_num(la, nan=-7.0).view(torch.int32), torch.nan_to_num(lb, nan=-7.0).view(torch.int32))
        assert torch.equal(a._debug["sel"], b._debug["sel"])
        assert torch.equal(a._debug["h_cond"], b._debug["h_cond"])
        pa, pb = a._debug["proj"], b._debug["proj"][:, :t.x.size(0)]
        assert pa.dtype == pb.dtype and torch.equal(pa.view(torch.int16), pb.view(torch.int16))
    assert len(set(counts)) > 1 and max(counts) < 40
    # sets: samples of one template, and different templates
    for templates, z_ in ((tpls[1], zs), (tpls, zs)):
        sa, sb = Gc.forward_set(templates, z_), Gn.forward_set(templates, z_)
        assert torch.equal(sa.count, sb.count) and torch.equal(sa.x, sb.x)
        for ga, gb in zip(sa.graphs(), sb.graphs()):          # (slots past a graph's count are not written)
            for name in ("x", "edge_index", "edge_attr"):
                assert torch.equal(getattr(ga, name), getattr(gb, name)), name
    # more classes than the limit: the node-wise path is taken (and is the same numbers anyway)
    Gl = make(5)
    c = Gl(tpls[2], zs[2])
    d = Gn(tpls[2], zs[2])
    assert torch.equal(c.edge_index, d.edge_index) and torch.equal(c.edge_attr, d.edge_attr)


# ------------------------------------------------------------------------------------------------------
# --z-dim other than 128 (train.py:374, AIGgan.py:78): z only enters through three bias vectors, one per layer
# that reads it, so the width of z is a row length of three weight matrices and a loop bound.
@pytest.mark.parametrize("zd,K", [(32, 64), (64, 16), (200, 64), (4, 8)])
def test_other_z_widths_against_the_checker_and_the_reference_formula(zd, K):
    g = torch.Generator().manual_seed(zd)
    torch.manual_seed(zd)
    G = AIGGenerator(z_dim=zd, candidate_k=K).to(DEV).eval()
    state = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    assert tuple(state["node_proj.0.weight"].shape) == (128, 128 + zd)
    assert tuple(state["edge_scorer.0.weight"].shape) == (128, 256 + zd)
    t = synth_aig(40, 900, 25, seed=zd, window=200, orientation="dataset")
    z = torch.randn(zd, generator=g)
    tpl = Data(x=t.x.to(DEV), node_depth=t.node_depth.to(DEV))
    out = G(tpl, z.to(DEV), return_debug=True)
    n = t.x.size(0)
    # the checker's fixed-order form: bit for bit
    ref = omodel.generator_forward_canonical(state, t.x, t.node_depth, z, K)
    assert np.array_equal(out._debug["h_cond"].cpu().numpy()[:n].view(np.int32), ref["h_cond"].view(np.int32))
    assert np.array_equal(out._debug["proj"].cpu().numpy()[:, :n].view(np.int32), ref["proj"].view(np.int32))
    assert np.array_equal(out.edge_index.cpu().numpy(), ref["edge_index"].numpy())
    assert np.array_equal(out.edge_attr.cpu().numpy(), ref["edge_attr"].numpy())
    assert np.array_equal(out.x.cpu().numpy(), ref["x"].numpy())
    # the reference's formulation (cat[h, z] through nn.Linear, models/generator.py:115-117): 1e-4
    plain = omodel.generator_forward(state, t.x, t.node_depth, z, K)
    np.testing.assert_allclose(out._debug["h_cond"].cpu().numpy()[:n], plain["h_cond"].numpy(), rtol=1e-4, atol=1e-5)
    assert np.array_equal(out.edge_index[1].cpu().numpy(), plain["edge_index"][1].numpy())
    # the set path and the bulk path give the one-at-a-time graphs
    zs = torch.randn((3, zd), generator=g).to(DEV)
    singles = [G(Data(x=t.x.to(DEV), node_depth=t.node_depth.to(DEV)), zs[i]) for i in range(3)]
    many = G.forward_many(tpl, zs)
    graphs = G.forward_set([tpl] * 3, zs).graphs()
    for a, b, c in zip(singles, many, graphs):
        for k in ("x", "edge_index", "edge_attr"):
            assert torch.equal(getattr(a, k), getattr(b, k)) and torch.equal(getattr(a, k), getattr(c, k)), k
    # a z of another width is refused before any kernel reads past it
    with pytest.raises(_abi.AigAbiError):
        G(tpl, torch.randn(zd + 4).to(DEV))


def test_other_z_widths_bf16_projection_and_refusals():
    zd = 96
    torch.manual_seed(3)
    G32 = AIGGenerator(z_dim=zd, candidate_k=32).to(DEV).eval()
    G16 = AIGGenerator(z_dim=zd, candidate_k=32, score_dtype="bf16").to(DEV).eval()
    G16.load_state_dict(G32.state_dict())
    t = synth_aig(30, 600, 20, seed=9, window=150, orientation="dataset")
    tpl = Data(x=t.x.to(DEV), node_depth=t.node_depth.to(DEV))
    z = torch.randn(zd, generator=torch.Generator().manual_seed(4)).to(DEV)
    a, b = G32(tpl, z, return_debug=True), G16(tpl, z, return_debug=True)
    n = t.x.size(0)
    ha, hb = a._debug["h_cond"][:n].float().cpu().numpy(), b._debug["h_cond"][:n].float().cpu().numpy()
    assert np.abs(ha - hb).max() <= BF16_TOL * max(1.0, np.abs(ha).max())
    assert torch.equal(a.edge_index[1], b.edge_index[1])
    for bad in (0, 2, 130, 1028):
        with pytest.raises(NotImplementedError):
            AIGGenerator(z_dim=bad)


def test_z_width_is_checked_at_the_c_abi():
    """``aig_gen_weights.z_dim`` outside a multiple of 4 in 4..1024 is refused with a message, not read past."""
    import ctypes as C
    lib = _abi.lib()
    G = AIGGenerator(z_dim=64, candidate_k=8).to(DEV).eval()
    w = G._weights_uncached()
    n = 40
    x = torch.zeros((n, 2), device=DEV)
    z = torch.zeros(64, device=DEV)
    proj = torch.empty((4, n, 128), device=DEV)
    ws = _abi.workspace(lib.aig_gen_node_proj_ws_bytes(n), torch.device(DEV))
    call = lambda: lib.aig_gen_node_proj(n, _abi.ptr(x), _abi.ptr(z), C.byref(w), None, _abi.ptr(proj), _abi.ptr(ws),
                                         ws.numel(), _abi.stream())
    assert call() == 0
    torch.cuda.synchronize()
    for bad in (6, 2, 1028, -4):
        w.z_dim = bad
        assert call() != 0
        assert b"z_dim" in lib.aig_last_error()
    w.z_dim = 64
    assert call() == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("n_and,K", [(2500, 16), (2500, 64), (2500, 100), (70000, 64), (150000, 8)])
def test_selection_that_forms_its_scores_equals_the_two_kernels(n_and, K, monkeypatch):
    """Scores by class never stored (``aig_gen_wire_batch`` with ``logits = NULL``: the selection kernel forms them -
    the default from 262,145 slots a call on) against the default below that size (scores written by one kernel,
    picked by the next): the same graphs, one at a time and as a set; block shapes of 8, 16 and 32 rows."""
    import gan_amd.generator as gmod
    torch.manual_seed(K)
    G = AIGGenerator(candidate_k=K).to(DEV).eval()
    t = synth_aig(60, n_and, 40, seed=n_and + K, window=max(300, n_and // 30), orientation="dataset")
    tpl = Data(x=t.x.to(DEV), node_depth=t.node_depth.to(DEV))
    zs = torch.randn((3, 128), generator=torch.Generator().manual_seed(2)).to(DEV)
    assert 3 * t.x.size(0) < gmod.SELECT_FORMS_SCORES_MIN_SLOTS or n_and > 80000
    monkeypatch.setattr(gmod, "SELECT_FORMS_SCORES_MIN_SLOTS", 1 << 40)
    want_one = G(tpl, zs[0])
    want_set = G.forward_set([tpl] * 3, zs).graphs()
    monkeypatch.setattr(gmod, "SELECT_FORMS_SCORES_MIN_SLOTS", 0)
    got_one = G(tpl, zs[0])
    got_set = G.forward_set([tpl] * 3, zs).graphs()
    for a, b in [(want_one, got_one)] + list(zip(want_set, got_set)):
        for k in ("x", "edge_index", "edge_attr"):
            assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert torch.equal(want_one.edge_index, want_set[0].edge_index)


def test_z_taken_from_the_middle_of_a_larger_tensor():
    """A z that is a view at an odd offset (the kernels read z as 16-byte vectors): copied, same graph."""
    torch.manual_seed(1)
    G = AIGGenerator(candidate_k=16).to(DEV).eval()
    t = synth_aig(20, 400, 12, seed=77, window=80, orientation="dataset")
    tpl = Data(x=t.x.to(DEV), node_depth=t.node_depth.to(DEV))
    big = torch.randn(3 * 128 + 5, device=DEV)
    z = big[1:129]
    assert z.data_ptr() % 16 != 0
    a, b = G(tpl, z), G(tpl, z.clone())
    assert torch.equal(a.edge_index, b.edge_index) and torch.equal(a.edge_attr, b.edge_attr)
    zs = big[1:1 + 3 * 128].view(3, 128)
    for x, y in zip(G.forward_many(tpl, zs), G.forward_many(tpl, zs.clone())):
        assert torch.equal(x.edge_index, y.edge_index)
