"""Host-side profile of the training step on the bench workload (tuning aid)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd.train import Trainer, build_parser
from gan_amd.data import Batch
from gan_amd.dataset import pad_data
from gan_amd.synth import synth_aig

dev = torch.device("cuda")
g = synth_aig(1000, 98000, 1000, seed=0, window=4096, orientation="dataset")
n = g.x.size(0)
real = Batch.from_data_list([pad_data(g, n, 2 * n)]).to(dev); real.sample_ids = [0]
args = build_parser().parse_args(["--dataset-dir", "x", "--out-dir", "/tmp/o", "--candidate-k", "64", "--save-every", "1000000000"])
tr = Trainer(args, n, 2 * n, dev)
for _ in range(4): tr.step(real)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): tr.step(real)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
st.print_callers("item")
st.print_callers("torch.tensor")
