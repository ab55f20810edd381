#!/usr/bin/env python3
"""Training-step timing on one GPU (development aid): batch 64 x (1+4) candidates, MIND-small default shapes."""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from digat_amd import synthetic, util  # noqa: E402
from digat_amd.model import Model, PrecomputedNewsEncoder  # noqa: E402
from digat_amd.trainer import SyntheticTrainSet, Trainer  # noqa: E402

dev = torch.device("cuda:0")
neighbors = int(sys.argv[1]) if len(sys.argv) > 1 else 3
spec = synthetic.SynthSpec(news_num=4096, sag_neighbors=neighbors, sag_hops=2, impressions=512, seed=0)
corpus = synthetic.make_corpus(spec)
cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=spec.news_graph_size, max_history_num=50,
                            category_num=17, graph_depth=3, dropout_rate=0.2, epoch=1, batch_size=64, lr=1e-4, weight_decay=0.0,
                            gradient_clip_norm=1.0)
model = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(corpus.news_embedding), trainable=True))
model.initialize()
model = model.to(dev)
dc = util.DeviceCorpus.from_numpy(corpus, dev)
ts = SyntheticTrainSet(corpus, 4, 0)
ts.negative_sampling()
tr = Trainer(model, cfg, dc, ts)
model.train()
idx = np.arange(64)
t_pre = time.perf_counter()
while time.perf_counter() - t_pre < 0.5:          # bring the GPU out of its idle power state (see bench.py)
    tr.train_step(idx)
    torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for i in range(K):
    tr.train_step((idx + 64 * i) % len(ts))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"train step: N={spec.news_graph_size} U=67 d=400 L=3, batch 64x5 = 320 rows: {dt*1e3:.2f} ms/step, {320/dt:.0f} rows/s, "
      f"{64/dt:.0f} behaviours/s")
