#!/usr/bin/env python3
"""How long the host needs to enqueue one scoring step vs how long the GPU needs to run it (development aid)."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from digat_amd import synthetic, util  # noqa: E402
from digat_amd.model import Model, PrecomputedNewsEncoder  # noqa: E402

dev = torch.device("cuda:0")
spec = synthetic.SynthSpec(news_num=8192, sag_neighbors=3, sag_hops=2, impressions=1024, seed=0)
corpus = synthetic.make_corpus(spec)
state = synthetic.make_state_dict(400, 17, 3, seed=0, bias_std=0.05)
cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=spec.news_graph_size, max_history_num=50,
                            category_num=17, graph_depth=3, dropout_rate=0.2)
model = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(corpus.news_embedding)))
model.graph_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
model = model.to(dev).eval()
dc = util.DeviceCorpus.from_numpy(corpus, dev)
util.prepare_news_side(model.graph_encoder, dc, 1024)
imp = corpus.row_impression
B = 1024
nb = dc.rows // B


def step(i):
    s = (i % nb) * B
    with torch.no_grad():
        return model.inference_grouped(*util.gather_batch_grouped(dc, s, s + B, imp))


from digat_amd import _lib  # noqa: E402
mode = os.environ.get("HO_MODE", "")
buf = torch.empty(B, device=dev)
if "copy" in mode:
    _step = step
    def step(i):                     # noqa: F811
        buf[:] = _step(i)
for i in range(5):
    step(i)
torch.cuda.synchronize()
if "prof" in mode:
    _lib.profile_start(10000)
    _lib.lib().digat_profile_pause(1)
if "fence" in mode:
    torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for i in range(K):
    step(5 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/K:.3f} ms/step, total {1e3*(t2-t0)/K:.3f} ms/step")
t0 = time.perf_counter()
for i in range(K):
    s = ((5 + i) % nb) * B
    g = util.gather_batch_grouped(dc, s, s + B, imp)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"gather only: host {1e3*(t1-t0)/K:.3f} ms/step")

# upper bound of what overlapping the gathers with the previous batch could buy: inputs gathered beforehand
pre = [util.gather_batch_grouped(dc, (i % nb) * B, (i % nb) * B + B, imp) for i in range(nb)]
torch.cuda.synchronize()
for i in range(10):
    with torch.no_grad():
        model.inference_grouped(*pre[i % nb])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K):
    with torch.no_grad():
        model.inference_grouped(*pre[(5 + i) % nb])
torch.cuda.synchronize()
print(f"inference only (inputs pre-gathered): {1e3*(time.perf_counter()-t0)/K:.3f} ms/step")
