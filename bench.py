#!/usr/bin/env python3
"""bench.py — MIND-shaped dev impressions scored per second through the HIP DIGAT path.

A step = one pass of the hot path over one batch of B=1024 (impression, candidate) rows of a synthetic
MIND-small-shaped dev set: on-device gather of the batch from the HBM-resident corpus tables
(util.py:65-67 of the reference) + ``Model.inference`` (DIGAT.inference, graph_depth 3, N=10, U=67,
d=400, fp32) + dot-product logits.  Inputs are resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      the dominant kernel of the step, timed live with HIP events over the timed region
  cpu_baseline  the oracle (unfused reference algorithm, torch-CPU) on a bounded sample, N=1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (the 5 PF headline is 2:1 sparse)

WORKLOADS = {
    # BASELINE.json configs[1]
    "mind-small-default": dict(sag_neighbors=3, sag_hops=2, depth=3, category_num=17,
                               label="MIND-small default: --graph_encoder=DIGAT neighbors=3 hops=2 (N=10, U=67), "
                                     "d=400, graph_depth=3, fp32 dev inference"),
    # BASELINE.json configs[2]
    "mind-small-stress": dict(sag_neighbors=8, sag_hops=2, depth=7, category_num=17,
                              label="MIND-small stress: neighbors=8 hops=2 (N=65, U=67), d=400, graph_depth=7"),
    # BASELINE.json configs[3] shape (18 categories)
    "mind-large-default": dict(sag_neighbors=5, sag_hops=2, depth=3, category_num=18,
                               label="MIND-large default shape: neighbors=5 hops=2 (N=26, U=68), d=400, graph_depth=3"),
}


def usable_cores() -> int:
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="rows per step (reference: batch_size*16 = 1024, main.py:42)")
    ap.add_argument("--workload", default="mind-small-default", choices=sorted(WORKLOADS))
    ap.add_argument("--impressions", type=int, default=1024, help="synthetic impressions per rank (~37 rows each)")
    ap.add_argument("--news", type=int, default=8192, help="synthetic news corpus size per rank")
    ap.add_argument("--cpu-rows", type=int, default=1536, help="max rows of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="time bound of the CPU-baseline sample")
    ap.add_argument("--per-row-users", action="store_true",
                    help="expand the user tensors per row as the reference's driver does (default: once per impression)")
    ap.add_argument("--projection", default="bf16x6", choices=["bf16x6", "bf16x6-pq3", "fp32"],
                    help="node projections: split-bf16 (fp32-equivalent) on the bf16 matrix cores, or fp32 MFMA")
    return ap.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    # test hook (one-GPU boxes): DIGAT_BENCH_TEST_SHARED_GPU=1 puts every rank on cuda:0 and runs the control-plane
    # collectives (barrier, max / sum of three scalars) over gloo; the data path has no collective either way
    shared_gpu = os.environ.get("DIGAT_BENCH_TEST_SHARED_GPU") == "1"
    device_index = 0 if shared_gpu else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    ctl_dev = torch.device("cpu") if shared_gpu else dev       # where the timing scalars are reduced
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if shared_gpu else "nccl")          # nccl = RCCL on ROCm

    from digat_amd import _lib, synthetic, util
    from digat_amd.model import Model, PrecomputedNewsEncoder

    wl = WORKLOADS[args.workload]
    spec = synthetic.SynthSpec(news_num=args.news, sag_neighbors=wl["sag_neighbors"], sag_hops=wl["sag_hops"],
                               category_num=wl["category_num"], impressions=args.impressions, seed=rank)
    corpus = synthetic.make_corpus(spec)       # each rank owns its shard of the dev rows (weak scaling)
    N, H, C, d, L = spec.news_graph_size, spec.max_history_num, spec.category_num, spec.embedding_dim, wl["depth"]
    state = synthetic.make_state_dict(d, C, L, seed=0, bias_std=0.05)
    cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=N, max_history_num=H,
                                category_num=C, graph_depth=L, dropout_rate=0.2)
    model = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(corpus.news_embedding)))
    model.graph_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(dev).eval()
    model.graph_encoder.projection_mode = args.projection

    dc = util.DeviceCorpus.from_numpy(corpus, dev)
    util.prepare_news_side(model.graph_encoder, dc, args.batch)     # news cache + c_n0 (setup, untimed)
    B = args.batch
    nbatches = max(1, dc.rows // B)
    mean_cand = corpus.rows / float(spec.impressions)

    imp_host = corpus.row_impression

    # The driver's scoring loop (util.score_rows): the grouped inputs of batch k+1 are gathered on a side stream while
    # batch k is scored (util.GroupedBatchPipeline).  The pipeline takes batches in order, so `step` ignores its
    # argument and walks a cursor over the dev rows, cycled.
    cursor = {"k": 0, "pipe": None, "base": 0}
    CHUNK = 512                                    # batches per pipeline instance (index arrays are built per instance)

    # consecutive batches alternate over two HIP streams, as util.score_rows does (util.batch_streams): batch k+1's
    # opening kernels run under batch k's last layer.  cursor["lanes"] = 1 puts every batch on the current stream.
    lanes = util.batch_streams(dev, int(os.environ.get("DIGAT_BENCH_LANES", "2")))
    lane_scores = [torch.empty(B, dtype=torch.float32, device=dev) for _ in lanes]
    cursor["lanes"] = len(lanes)

    def step(_i):
        k = cursor["k"]
        if cursor["pipe"] is None or k - cursor["base"] >= CHUNK:
            cursor["base"] = k
            order = [(((k + j) % nbatches) * B, min(((k + j) % nbatches) * B + B, dc.rows)) for j in range(CHUNK)]
            cursor["order"] = order
            cursor["pipe"] = None if args.per_row_users else util.GroupedBatchPipeline(dc, order, imp_host)
            for extra in lanes[1:]:
                extra.wait_stream(lanes[0])
        s, e = cursor["order"][k - cursor["base"]]
        lane = k % cursor["lanes"]
        with torch.no_grad(), torch.cuda.stream(lanes[lane]):
            inputs = cursor["pipe"].take(k - cursor["base"]) if cursor["pipe"] is not None else None
            if inputs is None:
                lane_scores[lane][:e - s] = model.inference(*util.gather_batch(dc, s, e))
            else:
                lane_scores[lane][:e - s] = model.inference_grouped(*inputs)
            if cursor["pipe"] is not None:
                cursor["pipe"].scored(k - cursor["base"])
        cursor["k"] = k + 1
        return e - s

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    # per-kernel HIP events cost the host two hipEventRecord calls per launch (~140 per step): every PROFILE_EVERY-th
    # step of the timed region is recorded, the others run unobserved.  The profiler is set up (thousands of
    # hipEventCreate) BEFORE the warm-up, so that the timed region follows the warm-up without an idle gap.
    PROFILE_EVERY = int(os.environ.get("DIGAT_BENCH_PROFILE_EVERY", "4"))
    _lib.profile_start(64 * (args.steps // PROFILE_EVERY + 2) * (L + 1))
    _lib.lib().digat_profile_pause(1)
    # Setup, untimed: bring the GPU out of its idle power state (the setup above leaves it idle for ~100 ms and the
    # clocks need tens of milliseconds of load to come back: with 5 warm-up steps = 10 ms the first timed steps ran at
    # half speed).  A dev run scores ~2 600 such batches back to back; steady state is what the metric means.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.3:
        for i in range(8):
            step(i)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    fence()
    profiled_steps = 0
    rows_done = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        sampled = i % PROFILE_EVERY == 0
        _lib.lib().digat_profile_pause(0 if sampled else 1)
        profiled_steps += int(sampled)
        rows_done += step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    prof = _lib.profile_stop()
    live_fraction = float(_lib.lib().digat_profile_live_row_fraction())

    # The timed region overlaps the news-graph kernels with the user graph's on a side stream, so the launch
    # durations above include the sharing.  A second, untimed pass on one stream gives each kernel's duration
    # with the chip to itself (reported as roofline.isolated_*; `frac` stays the timed region's).
    prev = _lib.lib().digat_set_side_stream(0)
    cursor["lanes"] = 1                          # ... and every batch on the same caller stream
    _lib.profile_start(64 * (args.steps + 1) * (L + 1))
    for i in range(min(args.steps, 10)):
        step(args.warmup + i)
    torch.cuda.synchronize()
    prof_iso = _lib.profile_stop()
    _lib.lib().digat_set_side_stream(prev)
    cursor["lanes"] = len(lanes)

    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rows_done, corpus.rows, spec.impressions], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rows_total = float(r[0].item())
        mean_cand = float(r[1].item()) / float(r[2].item())      # candidates per impression over every rank's shard
    else:
        rows_total = float(rows_done)

    if rank != 0:
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (HIP events on the launch stream, summed over the timed region)
    kinds = {k: v for k, v in prof.items() if v["launches"] > 0}
    # dominant = the kind with the largest SOLO time per step (the untimed single-stream pass): inside the timed region the
    # chip is shared by two batches and the side stream, and a launch's duration there says how long it waited, not what it cost
    iso_ms = {k: prof_iso[k]["ms"] / max(1, min(args.steps, 10)) for k in kinds if prof_iso.get(k, {}).get("launches", 0) > 0}
    dom = max(iso_ms, key=iso_ms.get) if iso_ms else max(kinds, key=lambda k: kinds[k]["ms"])

    kernel_ms = {k: round(v["ms"] / max(1, profiled_steps), 4) for k, v in kinds.items()}

    # HBM traffic of the roofline kernels comes from separate rocprofv3 --pmc passes (FETCH_SIZE and
    # WRITE_SIZE cannot share a pass); their per-launch means are kept under profiles/ and quoted here.
    symbols = {"proj": "gemm_bf16x6s_kernel<3>" if getattr(model.graph_encoder, "projection_mode", "").startswith("bf16x6")
               else "gemm_f32_kernel<128, 80, 4, 1, 1, 1>", "xattn": "xattn_sparse_kernel" if model.graph_encoder.resolved_xattn_mode("user") == "sparse" else "xattn_score_kernel",
               "agg": "xattn_agg_kernel",
               "topic": "topic_pool_kernel", "pool": "attn_pool_kernel"}

    def pmc_traffic(kind):
        import glob
        if args.workload != "mind-small-default" or args.batch != 1024:
            return None
        for path in sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc.json")), reverse=True):
            try:
                table = json.load(open(path))["kernels"]
            except (OSError, ValueError, KeyError):
                continue
            for name, v in table.items():
                if symbols.get(kind, "\0") in name:
                    return {"bytes_per_launch": v["hbm_bytes_mean"], "source": os.path.relpath(path, REPO)}
        return None

    def roof(kind):
        out = roof_of(kind, kinds[kind])
        iso = prof_iso.get(kind)
        if iso and iso["launches"] > 0:
            o2 = roof_of(kind, iso)
            out["isolated_achieved"], out["isolated_frac"] = o2["achieved"], o2["frac"]
            out["isolated_avg_launch_ms"] = o2["avg_launch_ms"]
            out["note"] = ("achieved/frac: launch durations inside the timed region, where news-graph kernels share the chip "
                           "on a side stream; isolated_*: the same launches on a single stream (untimed pass)")
        return out

    def roof_of(kind, v):
        per_launch_ms = v["ms"] / v["launches"]
        rate = v["work"] / (v["ms"] * 1e-3)
        pmode = getattr(model.graph_encoder, "projection_mode", "fp32")
        if kind == "proj" and pmode.startswith("bf16x6"):
            # the projections run as 6 bf16 MFMA products per fp32 product (exact 3-way operand split; "bf16x6-pq3": 6 for h,
            # 3 for P and Q = 4 on average): price the EXECUTED bf16 flops against the dense bf16 peak, and quote the
            # fp32-equivalent rate
            nprod = 6 if pmode == "bf16x6" else 4
            return {"kernel": "proj (gemm_bf16x6s_kernel)", "bound": "mfma", "achieved": nprod * rate / 1e12,
                    "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": nprod * rate / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                    "traffic": pmc_traffic(kind),
                    "mfma_dtype": "bf16 (3-way split of f32, f32 accumulate; products per fp32 product: %s)"
                                  % ("6" if nprod == 6 else "6 for h, 3 for P and Q"),
                    "fp32_equivalent_tflops": rate / 1e12, "algorithmic_flops_per_launch": v["work"] / v["launches"],
                    "executed_flops_per_launch": nprod * v["work"] / v["launches"],
                    "peak_note": "nominal dense bf16 peak (2.4 GHz); on random operands the chip holds about 1.9-2.0 GHz "
                                 "(MI355X_MICROARCH.md), i.e. about 2.0 PFLOP/s",
                    "avg_launch_ms": per_launch_ms, "launches": v["launches"]}
        if kind in ("proj", "linear"):
            return {"kernel": kind, "bound": "mfma", "achieved": rate / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": rate / 1e12 / MFMA_F32_PEAK_TFLOPS, "traffic": pmc_traffic(kind),
                    "mfma_dtype": "f32", "algorithmic_flops_per_launch": v["work"] / v["launches"],
                    "avg_launch_ms": per_launch_ms, "launches": v["launches"]}
        return {"kernel": kind, "bound": "hbm", "achieved": rate / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": rate / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic(kind),
                "algorithmic_bytes_per_launch": v["work"] / v["launches"], "avg_launch_ms": per_launch_ms,
                "launches": v["launches"]}

    # ---- CPU baseline + AUC match on a bounded sample (rank 0, N=1 only)
    cpu_baseline, auc_match = None, None
    if world == 1 and args.cpu_rows > 0:
        from oracle import digat_oracle as O      # the checker / baseline, never the product
        imp_np = corpus.row_impression
        cores = usable_cores()
        torch.set_num_threads(cores)
        p = O.as_params(state)
        emb = torch.from_numpy(corpus.news_embedding)
        ids = torch.from_numpy(corpus.news_node_ID.astype(np.int64))
        sa = emb.index_select(0, ids.flatten()).view(ids.shape[0], -1, d)
        masks, graphs = torch.from_numpy(corpus.news_graph_mask), torch.from_numpy(corpus.news_graph)
        # whole impressions, at most --cpu-rows rows and about --cpu-seconds of CPU work
        imp_starts = np.r_[0, np.flatnonzero(np.diff(imp_np)) + 1, corpus.rows]
        with torch.no_grad():
            c_n0 = O.news_graph_context(p, sa, masks)
            cpu_scores, n_rows, n_imps = [], 0, 0
            t1 = time.perf_counter()
            while n_imps + 1 < len(imp_starts):
                s, e = int(imp_starts[n_imps]), int(imp_starts[n_imps + 1])
                # batch a few impressions together: up to 64 rows per oracle call
                k = n_imps + 1
                while k + 1 < len(imp_starts) and int(imp_starts[k + 1]) - s <= 64:
                    k += 1
                e = int(imp_starts[k])
                if e > args.cpu_rows and n_rows > 0:
                    break
                imp = torch.from_numpy(corpus.row_impression[s:e])
                cand = torch.from_numpy(corpus.row_candidate[s:e].astype(np.int64))
                hist = torch.from_numpy(corpus.history.astype(np.int64)).index_select(0, imp)
                ue = emb.index_select(0, hist.flatten()).view(e - s, H, d)
                cpu_scores.append(O.row_logits(
                    p, L, ue, torch.from_numpy(corpus.user_graph).index_select(0, imp),
                    torch.from_numpy(corpus.user_category_mask).index_select(0, imp),
                    torch.from_numpy(corpus.user_category_indices).index_select(0, imp),
                    sa.index_select(0, cand), graphs.index_select(0, cand), masks.index_select(0, cand),
                    c_n0.index_select(0, cand)))
                n_rows, n_imps = e, k
                if time.perf_counter() - t1 > args.cpu_seconds:
                    break
            cpu_s = time.perf_counter() - t1
        last_imp = n_imps
        cpu_scores = torch.cat(cpu_scores).numpy()
        cpu_baseline = {"value": (n_rows / mean_cand) / cpu_s, "unit": "impressions/s", "cores": cores,
                        "kind": "port", "rows_per_s": n_rows / cpu_s,
                        "sample": f"first {n_rows} rows ({last_imp} whole impressions) of the same synthetic dev set, "
                                  f"unfused reference algorithm (oracle/digat_oracle.py, torch-CPU fp32, B=64 batches), "
                                  f"{cpu_s:.1f}s"}
        gpu_scores = util.score_rows(model, dc, 0, n_rows, B).cpu().numpy()
        from digat_amd import evaluate
        lab, ri = corpus.row_label[:n_rows], corpus.row_impression[:n_rows]
        mg = evaluate.scoring(lab, evaluate.impression_ranks(gpu_scores, ri), ri)
        mc = evaluate.scoring(lab, evaluate.impression_ranks(cpu_scores, ri), ri)
        auc_match = {"max_abs_metric_diff": float(np.max(np.abs(np.array(mg) - np.array(mc)))),
                     "max_abs_score_diff": float(np.max(np.abs(gpu_scores - cpu_scores))),
                     "gpu": [round(v, 6) for v in mg], "cpu": [round(v, 6) for v in mc],
                     "metrics": ["AUC", "MRR", "nDCG@5", "nDCG@10"], "tolerance": 1e-4}

    value = (rows_total / mean_cand) / elapsed
    out = {
        "metric": "MIND dev impressions scored/sec (AUC-matched)",
        "value": value,
        "unit": "impressions/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl["label"], "projection": args.projection,
                   "user_side": "per row" if args.per_row_users else "once per impression (row_group index)",
                   "user_graph_eq8": model.graph_encoder.resolved_xattn_mode("user") + " (chosen from the corpus: mean adjacency entries per node)",
                   "news_graph_eq8": ("small-graph kernel (n <= 16)" if N <= 16 else model.graph_encoder.resolved_xattn_mode("news")),
                   "rows_per_step": B, "N": N, "U": H + C, "d": d, "graph_depth": L,
                   "mean_candidates_per_impression": round(mean_cand, 3), "parallelism": f"dp{world} (row shards, no data-path collective)"},
        "rows_per_s": rows_total / elapsed,
        "roofline": roof(dom),
        "roofline_xattn": roof("xattn") if "xattn" in kinds else None,
        "kernel_ms_per_step": kernel_ms,
        "kernel_ms_per_step_single_stream": {k: round(v["ms"] / max(1, min(args.steps, 10)), 4)
                                             for k, v in prof_iso.items() if v["launches"] > 0},
        # rows projected / rows nominal over the row-list launches (user-graph layers >= 1 and featureAffine): the encoder
        # leaves out nodes and topic buckets that cannot reach its outputs; the proj roofline prices EXECUTED flops
        "live_row_fraction": live_fraction if live_fraction >= 0 else None,
        "cpu_baseline": cpu_baseline,
        "auc_match": auc_match,
    }
    print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
