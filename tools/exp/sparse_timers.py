"""Phase timers of the wave-per-centre Eq. 8 kernel (experiment build: hipcc -DDIGAT_SPARSE_TIMERS -> tools/exp/libdigat_timers.so)."""
import os, sys, ctypes as C, types, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["DIGAT_HIP_LIB"] = os.path.join(HERE, "libdigat_timers.so")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from digat_amd import _lib, synthetic, util
from digat_amd.model import Model, PrecomputedNewsEncoder
spec = synthetic.SynthSpec(news_num=8192, impressions=600, seed=0)
corpus = synthetic.make_corpus(spec)
state = synthetic.make_state_dict(400, 17, 3, seed=0, bias_std=0.05)
cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=10, max_history_num=50, category_num=17, graph_depth=3, dropout_rate=0.2)
dev = torch.device("cuda:0")
model = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(corpus.news_embedding)))
model.graph_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
model = model.to(dev).eval()
dc = util.DeviceCorpus.from_numpy(corpus, dev)
util.prepare_news_side(model.graph_encoder, dc, 1024)
lib = C.CDLL(_lib.LIB_PATH)
lib.digat_debug_sparse_timers.argtypes = [C.POINTER(C.c_double)]
util.score_rows(model, dc, 0, 8192, 1024)
torch.cuda.synchronize()
out = (C.c_double * 9)()
lib.digat_debug_sparse_timers(out)
util.score_rows(model, dc, 0, 16384, 1024)      # 16 batches x 2 row-list launches
torch.cuda.synchronize()
lib.digat_debug_sparse_timers(out)
v = np.array(list(out)); waves = v[8]
names = ["list entry", "setup loads (A row, a, X, flags)", "score: wait for P' (+Q first)", "score: compute + reduce", "softmax",
         "agg: wait for h", "agg: fma", "store"]
print("sampled waves", waves)
for nme, x in zip(names, v[:8]):
    print(f"{nme:36s} {x / waves:9.0f} cycles per centre")
print("sum", v[:8].sum() / waves)
