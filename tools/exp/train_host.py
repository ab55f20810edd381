"""Is the training step host-bound?  Enqueue time per step (no sync inside the loop) vs wall time per step (run on the GPU box)."""
import os, sys, time, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

args = types.SimpleNamespace(news=0, train_news_encoder="table", projection="auto", batch=4096, train_precision="fp32")
D = types.SimpleNamespace(rank=0, world=1, dev=torch.device("cuda:0"), device_index=0)
W = bench.build_workload("mind-small-default", args, D, 4096, trainable=True)
from digat_amd.trainer import SyntheticTrainSet, Trainer
cfg = W.cfg
cfg.epoch, cfg.batch_size, cfg.lr, cfg.weight_decay, cfg.gradient_clip_norm = 1, 64, 1e-4, 0.0, 1.0
cfg.train_precision = "fp32"
ts = SyntheticTrainSet(W.corpus, 4, seed=0)
ts.negative_sampling()
tr = Trainer(W.model, cfg, W.dc, ts, local_rank=-1)
tr.model.train()
nb = len(ts) // 64
k = 0
def step():
    global k
    idx = (np.arange(64) + 64 * (k % nb)) % len(ts)
    k += 1
    return tr.train_step(idx, read_loss=False)
for _ in range(20):
    step()
torch.cuda.synchronize()
for n in (20, 20):
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"steps {n}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, wall {1e3 * (t2 - t0) / n:.3f} ms/step, drain after loop {1e3 * (t2 - t1):.2f} ms")
