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
if os.environ.get("DIGAT_BENCH_CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(25)
for n in (20, 20, 60, 100, 20):
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"steps {n}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, wall {1e3 * (t2 - t0) / n:.3f} ms/step, drain after loop {1e3 * (t2 - t1):.2f} ms")

# which torch ops launch what: one profiled step, ops with their input shapes (where do the element-wise adds come from?)
if os.environ.get("TRAIN_OPS", "1") != "0":
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    rows = {}
    for e in prof.events():
        if e.name.startswith("aten::") or "Backward" in e.name or e.name.startswith("Optimizer"):
            key = (e.name, str(e.input_shapes)[:90])
            n, cpu, dev = rows.get(key, (0, 0.0, 0.0))
            rows[key] = (n + 1, cpu + e.cpu_time_total, dev + e.device_time_total)
    print(f"{'op':44s} {'shapes':90s} {'n/step':>6s} {'cpu us':>8s} {'dev us':>8s}")
    for (name, shapes), (n, cpu, dev) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{name[:44]:44s} {shapes:90s} {n / 3:6.1f} {cpu / 3:8.1f} {dev / 3:8.1f}")
    print()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))

# per-step host timestamps of a long run: is the slow part periodic (allocator growth, garbage collection, epoch wrap)?
if os.environ.get("TRAIN_STAMPS"):
    import gc
    torch.cuda.synchronize()
    gc_events = []
    gc.callbacks.append(lambda phase, info: gc_events.append((time.perf_counter(), phase, info.get("generation"))))
    stamps = [time.perf_counter()]
    for _ in range(300):
        step()
        stamps.append(time.perf_counter())
    torch.cuda.synchronize()
    end = time.perf_counter()
    dt = np.diff(np.array(stamps)) * 1e3
    print(f"300 steps: wall {1e3 * (end - stamps[0]) / 300:.3f} ms/step; host per step: median {np.median(dt):.3f}, p90 {np.percentile(dt, 90):.3f}, max {dt.max():.3f} ms")
    slow = np.flatnonzero(dt > 1.5 * np.median(dt))
    print("slow steps (index: ms):", ", ".join(f"{i}: {dt[i]:.1f}" for i in slow[:40]))
    g2 = [(t - stamps[0]) * 1e3 for t, ph, gen in gc_events if ph == "start" and gen == 2]
    print(f"gc: {sum(1 for e in gc_events if e[1] == 'start')} collections, generation 2 at ms:", [round(x) for x in g2[:20]])
