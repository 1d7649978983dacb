"""End-to-end check of the bulk sampling command line at the bulk configuration (tuning aid, not a test):
writes a one-graph dataset and a generator checkpoint, runs AIGgan.py's main for N samples, reports the rate
including the .pt writes."""
import os, sys, tempfile, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_amd import sample as cli
from gan_amd.dataset import save_data_list_as_pt
from gan_amd.generator import AIGGenerator
from gan_amd.synth import ISCAS85_SHAPES, synth_aig

n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 512
name, n_pi, n_po, n_and = ISCAS85_SHAPES[-1]
root = tempfile.mkdtemp()
g = synth_aig(n_pi, n_and, n_po, seed=0, window=max(8, n_and // 8 + n_pi), orientation="dataset")
g.edge_attr = g.edge_attr.long()
save_data_list_as_pt([g], os.path.join(root, "ds"), prefix="g")
torch.manual_seed(1)
torch.save(AIGGenerator(candidate_k=256).state_dict(), os.path.join(root, "gen.pt"))
t0 = time.perf_counter()
cli.main(["--dataset-dir", os.path.join(root, "ds"), "--gen-ckpt", os.path.join(root, "gen.pt"), "--out", os.path.join(root, "out", "s.pt"),
          "--candidate-k", "256", "--num-samples", str(n_samples), "--score-dtype", "bf16", "--seed", "3"])
dt = time.perf_counter() - t0
files = sorted(os.listdir(os.path.join(root, "out")))
d = torch.load(os.path.join(root, "out", files[-1]), weights_only=True)
print(f"{len(files)} files in {dt:.2f} s = {len(files) / dt:.0f} samples/s incl. start-up and writes; last: "
      f"{ {k: tuple(v.shape) for k, v in d.items()} }")
