"""How long the host needs to ISSUE one scoring step (all launches of a batch), GPU idle before each."""
import os, sys, time, types, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
args = bench.parse_args() if hasattr(bench, "parse_args") else None
sys.argv = [sys.argv[0]]
args = bench.parse_args()
args.impressions = 4000
D = bench.Dist(args)
W = bench.build_workload("mind-small-default", args, D, args.impressions)
sc = bench.Scorer(W, args, D)
for _ in range(30):
    sc.step()
torch.cuda.synchronize()
ts = []
for _ in range(40):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); sc.step(); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("host issue time per step (GPU idle at start): median %.0f us, min %.0f us, max %.0f us" % (np.median(ts) * 1e6, min(ts) * 1e6, max(ts) * 1e6))
t0 = time.perf_counter()
for _ in range(200):
    sc.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("200 steps: host loop %.1f ms, until the GPU is done %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
