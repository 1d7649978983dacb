"""GPU time of each recorded term of the training step alone on the device (100k-node workload): the real-side term,
the gradient penalty, the generated side, the raw side - events around 20 replays each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_amd.data import Batch
from gan_amd.dataset import pad_data
from gan_amd.synth import synth_aig
from gan_amd.train import Trainer, build_parser

dev = torch.device("cuda", 0)
n_and = int(os.environ.get("N_AND", "98000"))
graphs = [synth_aig(1000, n_and, 1000, seed=s, window=4096, orientation="dataset") for s in range(2)]
mn = max(g.x.size(0) for g in graphs); me = max(g.edge_index.size(1) for g in graphs)
batches = [Batch.from_data_list([pad_data(g, mn, me)]).to(dev) for g in graphs]
for i, b in enumerate(batches): b.sample_ids = [i]
args = build_parser().parse_args(["--dataset-dir", "unused", "--candidate-k", "64", "--save-every", "1000000"])
tr = Trainer(args, mn, me, dev); tr.global_step = 1
for i in range(4): tr.step(batches[i % 2])
torch.cuda.synchronize()
terms = next(iter(tr._captured.values()))
alpha = torch.rand(1, 1, device=dev)
def timed(name, fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name:28s} {a.elapsed_time(b) / reps * 1e3:8.1f} us")
timed("replay_real", terms.replay_real)
timed("replay_gp", lambda: terms.replay_gp(alpha))
timed("replay_fake", terms.replay_fake)
timed("replay_raw", terms.replay_raw)
timed("add_gradients + fake", lambda: (terms.add_fake_gradients(), terms.add_gradients()))
timed("opt_D.step", tr.opt_D.step)
timed("zero_grad", lambda: tr.opt_D.zero_grad(set_to_none=False))
