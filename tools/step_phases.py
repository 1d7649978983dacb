"""Phase timing of one critic iteration on the bench workload (tuning aid)."""
import os, sys, torch, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gan_amd.train import Trainer, build_parser, sanitize_for_D, gradient_penalty
from gan_amd.constraints import enforce_aig_constraints, aig_constraint_penalties
from gan_amd.data import Batch
from gan_amd.dataset import pad_data
from gan_amd.synth import synth_aig

dev = torch.device("cuda")
args = build_parser().parse_args(["--dataset-dir", "x", "--out-dir", "/tmp/o", "--candidate-k", "64"])
g = synth_aig(1000, 98000, 1000, seed=0, window=4096, orientation="dataset")
n, E = g.x.size(0), g.edge_index.size(1)
tr = Trainer(args, n, 2 * n, dev)
real = Batch.from_data_list([pad_data(g, n, 2 * n)]).to(dev); real.sample_ids = [0]
for _ in range(3): tr.step(real)
torch.cuda.synchronize()
def timed(fn, reps=5):
    out = None
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        s.record(); out = fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2], out
G, D = tr.G, tr.D
tpl = tr.templates_of(real)
t_gen, raw = timed(lambda: [G(t, tr.draw_z()) for t in tpl])
t_enf, corr = timed(lambda: [enforce_aig_constraints(r) for r in raw])
t_pad, fake_padded = timed(lambda: tr._pad_batch(corr))
t_san, fake_for_D = timed(lambda: sanitize_for_D(fake_padded))
real_for_D = sanitize_for_D(real)
t_dreal, dr = timed(lambda: torch.mean(D(real_for_D)))
t_dfake, df = timed(lambda: torch.mean(D(fake_for_D)))
t_gp, (gp, gn) = timed(lambda: gradient_penalty(D, real, fake_padded, dev, use_edges=False, alpha=tr.draw_alpha()))
def full():
    tr.opt_D.zero_grad()
    a = torch.mean(D(real_for_D)); b = torch.mean(D(fake_for_D))
    gp, _ = gradient_penalty(D, real, fake_padded, dev, use_edges=False, alpha=tr.draw_alpha())
    loss = (b - a) + 10.0 * gp
    loss.backward()
    return loss
t_full, _ = timed(full)
def gp_only():
    tr.opt_D.zero_grad()
    gp, _ = gradient_penalty(D, real, fake_padded, dev, use_edges=False, alpha=tr.draw_alpha())
    gp.backward()
t_gpfb, _ = timed(gp_only)
def dd_only():
    tr.opt_D.zero_grad()
    a = torch.mean(D(real_for_D)); b = torch.mean(D(fake_for_D)); (b - a).backward()
t_dd, _ = timed(dd_only)
t_opt, _ = timed(lambda: tr.opt_D.step())
t_pen, _ = timed(lambda: [aig_constraint_penalties(r) for r in raw])
t_step, _ = timed(lambda: tr.step(real))
print(f"gen {t_gen:.2f}  enforce {t_enf:.2f}  pad {t_pad:.2f}  sanitize {t_san:.2f}  D(real) fwd {t_dreal:.2f}  D(fake) fwd {t_dfake:.2f}  "
      f"GP fwd {t_gp:.2f}\ncritic iteration fwd+bwd {t_full:.2f}  (GP fwd+bwd alone {t_gpfb:.2f}, D real+fake fwd+bwd alone {t_dd:.2f})  "
      f"Adam {t_opt:.2f}  penalties {t_pen:.2f}  full step {t_step:.2f} ms")
