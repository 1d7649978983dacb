"""Host-side profile of the training step on the ISCAS85-like minibatch (tuning aid)."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gan_amd.train import Trainer, build_parser
from gan_amd.data import Batch
from gan_amd.dataset import pad_data

dev = torch.device("cuda")
graphs = bench.workload_minibatches("iscas85", 0, 1)[0][0]
mn = max(g.x.size(0) for g in graphs); me = max(g.edge_index.size(1) for g in graphs)
real = Batch.from_data_list([pad_data(g, mn, me) for g in graphs]).to(dev); real.sample_ids = list(range(len(graphs)))
args = build_parser().parse_args(["--dataset-dir", "x", "--out-dir", "/tmp/o", "--candidate-k", "64", "--save-every", "1000000000"])
tr = Trainer(args, mn, me, dev)
for _ in range(4): tr.step(real)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): tr.step(real)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) * 100)
pr = cProfile.Profile()
pr.enable()
for _ in range(10): tr.step(real)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(40)
st.sort_stats("tottime").print_stats(25)
