"""The one-launch critic update (aig_adam_sum3) and the logged scalars (aig_step_metrics) through the C ABI, against
torch.optim.Adam and plain tensor arithmetic."""
import numpy as np
import pytest
import torch

from gan_amd import _abi

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
SHAPES = [(128, 2), (128,), (1, 4, 32), (128, 128), (64, 128), (1, 64), (1,), (300,), (257,)]


def make_table(params, grads, srcs, exp_avg, exp_avg_sq):
    chunk = int(_abi.lib().aig_adam_chunk_elems())
    recs = []
    for i, p in enumerate(params):
        for off in range(0, p.numel(), chunk):
            b = 4 * off
            recs.append((p.data_ptr() + b, grads[i].data_ptr() + b) +
                        tuple(0 if s[i] is None else s[i].data_ptr() + b for s in srcs) +
                        (exp_avg[i].data_ptr() + b, exp_avg_sq[i].data_ptr() + b, min(chunk, p.numel() - off)))
    return torch.from_numpy(np.asarray(recs, dtype=np.int64).reshape(-1)).to(DEV), len(recs)


def run(mode, table, nchunks, steps, step_no, lr=1e-4, b1=0.5, b2=0.9, eps=1e-8):
    lib = _abi.lib()
    sp = torch.tensor([s.data_ptr() for s in steps], dtype=torch.int64, device=DEV)
    n = max(step_no, 1)
    _abi.check(lib.aig_adam_sum3(nchunks, _abi.ptr(table), len(steps), _abi.ptr(sp), float(n), lr, b1, b2, eps,
                                 float(1 - b1 ** n), float((1 - b2 ** n) ** 0.5), mode, _abi.stream()), "aig_adam_sum3")
    torch.cuda.synchronize()


def fresh(seed):
    g = torch.Generator().manual_seed(seed)
    params = [torch.randn(s, generator=g).to(DEV) for s in SHAPES]
    srcs = [[(torch.randn(s, generator=g) * 0.1).to(DEV) for s in SHAPES] for _ in range(3)]
    srcs[2][1] = None          # a term that does not reach every parameter (the penalty and the attention vectors)
    srcs[2][2] = None
    return params, srcs


def test_update_matches_torch_adam_over_several_steps():
    params, srcs = fresh(1)
    ref = [torch.nn.Parameter(p.clone()) for p in params]
    opt = torch.optim.Adam(ref, lr=1e-4, betas=(0.5, 0.9), fused=True)
    grads = [torch.full_like(p, 7.0) for p in params]            # (stale contents: the launch overwrites them)
    m, v = [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params]
    steps = [torch.zeros((), dtype=torch.float32, device=DEV) for _ in params]
    table, nchunks = make_table(params, grads, srcs, m, v)
    for it in range(1, 6):
        for i, r in enumerate(ref):
            r.grad = sum(s[i] for s in srcs if s[i] is not None) * it
        for s in srcs:                                       # fresh gradients every step, same buffers
            for t in s:
                if t is not None:
                    t.mul_(it / max(it - 1, 1)) if it > 1 else None
        want = [((0.0 + srcs[0][i]) + srcs[1][i]) + (srcs[2][i] if srcs[2][i] is not None else 0.0) for i in range(len(params))]
        for i, r in enumerate(ref):
            r.grad = want[i].clone()
        opt.step()
        run(3, table, nchunks, steps, it)
        for i, p in enumerate(params):
            assert torch.equal(grads[i], want[i]), i                                     # the sums, bit for bit
            assert float((p - ref[i].detach()).abs().max()) <= 1e-7 + 1e-6 * float(ref[i].abs().max()), (it, i)
            st = opt.state[ref[i]]
            assert float((m[i] - st["exp_avg"]).abs().max()) <= 1e-9 + 1e-6 * float(st["exp_avg"].abs().max())
            assert float((v[i] - st["exp_avg_sq"]).abs().max()) <= 1e-12 + 1e-6 * float(st["exp_avg_sq"].abs().max())
            assert float(steps[i]) == float(it) == float(st["step"])


def test_sum_then_update_equals_both_at_once():
    (pa, srcs_a), (pb, srcs_b) = fresh(2), fresh(2)
    outs = []
    for params, srcs, modes in ((pa, srcs_a, (3,)), (pb, srcs_b, (1, 2))):
        grads = [torch.zeros_like(p) for p in params]
        m, v = [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params]
        steps = [torch.zeros((), dtype=torch.float32, device=DEV) for _ in params]
        table, nchunks = make_table(params, grads, srcs, m, v)
        for it in (1, 2, 3):
            for mode in modes:
                run(mode, table, nchunks, steps, it)
        outs.append((params, grads, m, v, steps))
    for a, b in zip(outs[0], outs[1]):
        assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_update_rejects_bad_arguments():
    lib = _abi.lib()
    t = torch.zeros(8, dtype=torch.int64, device=DEV)
    assert lib.aig_adam_sum3(0, _abi.ptr(t), 0, None, 1.0, 1e-4, 0.5, 0.9, 1e-8, 0.5, 0.3, 3, _abi.stream()) != 0
    assert lib.aig_adam_sum3(1, _abi.ptr(t), 0, None, 1.0, 1e-4, 0.5, 0.9, 1e-8, 0.5, 0.3, 4, _abi.stream()) != 0
    assert lib.aig_adam_sum3(1, _abi.ptr(t), 0, None, 1.0, 1e-4, 0.5, 0.9, 1e-8, 0.0, 0.3, 3, _abi.stream()) != 0


@pytest.mark.parametrize("B,scale", [(1, 1.0), (3, 1.0), (11, 5.5)])
def test_step_metrics(B, scale):
    lib = _abi.lib()
    g = torch.Generator().manual_seed(B)
    d_real, d_fake, gp, gn, d_raw = (torch.randn(1, generator=g).to(DEV) for _ in range(5))
    pen9 = (torch.rand(B, 9, generator=g) * 1000).to(DEV)
    out = torch.empty(8, dtype=torch.float32, device=DEV)
    _abi.check(lib.aig_step_metrics(B, _abi.ptr(d_real), _abi.ptr(d_fake), _abi.ptr(gp), _abi.ptr(gn), _abi.ptr(d_raw),
                                    _abi.ptr(pen9), 10.0, 0.01, scale, _abi.ptr(out), _abi.stream()), "aig_step_metrics")
    pen = pen9[:, 8].mean() * scale if scale != 1.0 else pen9[:, 8].mean()
    want = torch.stack([d_real[0], d_fake[0], d_real[0] - d_fake[0], gp[0], gn[0], (d_fake[0] - d_real[0]) + 10.0 * gp[0],
                        -d_raw[0] + 0.01 * pen, pen])
    assert float((out - want).abs().max()) <= 1e-6 * float(want.abs().max())
    assert torch.equal(out[:5], want[:5]), (out[:5] - want[:5]).tolist()
