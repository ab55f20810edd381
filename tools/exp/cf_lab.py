"""Fused user-context kernel lab (round 6): phase timers of a -DDIGAT_CF_TIMERS build.
  DIGAT_HIP_LIB=tools/exp/lib_cft.so python tools/exp/cf_lab.py [B=4096]"""
import ctypes as C, os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from digat_amd import _lib, synthetic
from digat_amd.graphEncoders import DIGAT
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
N, H, Cn, d, L = 10, 50, 17, 400, 3
cfg = types.SimpleNamespace(news_graph_size=N, max_history_num=H, category_num=Cn, graph_depth=L, dropout_rate=0.2)
enc = DIGAT(cfg, d)
enc.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(d, Cn, L, seed=0, bias_std=0.05).items()})
enc = enc.to(dev).eval(); enc.projection_mode = "fp16x3"
batch = synthetic.make_encoder_batch(B, N, H, Cn, d, seed=1)
keys = ("news_graph_embeddings", "news_graph", "news_graph_mask", "user_news_embedding", "user_graph", "user_category_mask", "user_category_indices")
args = [torch.from_numpy(np.ascontiguousarray(batch[k])).to(dev) for k in keys]
Lb = _lib.lib()
with torch.no_grad():
    for _ in range(5): enc(*args)
    torch.cuda.synchronize()
    if hasattr(Lb, "digat_debug_cf_timers"):
        Lb.digat_debug_cf_timers.argtypes = [C.POINTER(C.c_double)]
        o = (C.c_double * 16)(); Lb.digat_debug_cf_timers(o)
        for _ in range(10): enc(*args)
        torch.cuda.synchronize(); Lb.digat_debug_cf_timers(o); v = list(o); n = max(v[15], 1)
        names = ["prologue (first loads + slots)", "row 0 total", "row 1 total", "row 2 total", "row 3 total", "tiles barrier", "phase 2 K loop", "epilogue + scores",
                 "softmax + output", "rows: select + scores", "rows: barrier 1", "rows: softmax", "rows: barrier 2", "rows: fp32 MFMA", "-", "-"]
        print(f"{int(n)} workgroups sampled; cycles per workgroup (wave 0, lane 0):")
        for k in range(14):
            print(f"  [{k:2d}] {names[k]:32s} {v[k] / n:10.0f}")
        print("  rows 0-3 stamps [1..4] include [9..13] (and the split stores + next row's load issue)")
