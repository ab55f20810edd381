"""GPU suite: bench.py end to end, as the driver launches it (one process; two ranks under torch.distributed.run with
both ranks on the box's single GPU and gloo for the control-plane scalars)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}


def _last_json_line(out):
    """The driver's view: stdout carries exactly ONE JSON line, the last line, small enough for an 8 KB tail."""
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.strip().splitlines()[-1] == lines[0], out[-3000:]
    assert len(lines[0]) < 6000, len(lines[0])
    return json.loads(lines[0])


def _detail(path):
    with open(path) as f:
        return json.load(f)


def test_bench_single_gpu_prints_the_contract_line(tmp_path):
    detail = str(tmp_path / "detail.json")
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2", "--impressions", "600", "--news", "2048",
           "--cpu-rows", "256", "--cpu-seconds", "5", "--extra-steps", "2", "--e2e-impressions", "1500", "--detail", detail]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    # ---- the compact line alone satisfies the contract (round 4's single 26 KB line did not survive the driver's stdout tail)
    short = _last_json_line(res.stdout)
    assert REQUIRED <= set(short) and "cpu_baseline" in short
    assert short["n_gpus"] == 1 and short["steps"] == 6 and short["warmup"] == 2 and short["value"] > 0
    assert short["unit"] == "impressions/s" and short["dtype"] == "f32" and short["vs_baseline"] is None and short["valid"] is True
    assert "AUC-matched" in short["metric"] and "workload" in short["config"] and short["config"]["rows_per_step"] == 4096
    sroof = short["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(sroof) and 0 < sroof["frac"] < 1
    assert sroof["traffic"] is None or sroof["traffic"] > 0
    assert short["cpu_baseline"]["kind"] == "port" and short["cpu_baseline"]["value"] > 0 and short["cpu_baseline"]["cores"] >= 1
    assert short["auc_match"]["max_abs_metric_diff"] <= short["auc_match"]["tolerance"]
    assert short["auc_match_trained"]["max_abs_metric_diff"] <= 1e-4
    assert 0 < short["roofline_xattn"]["frac"] < 1 and 0 < short["roofline_step"]["frac"] < 1
    assert {"twin", "l0", "news"} <= set(short["roofline_xattn"]["parts"])
    for part in short["roofline_xattn"]["parts"].values():
        assert part["us"] > 0 and 0 < part["frac"] < 1 and part["alg_MB"] > 0
    # one number per extra workload, the oracle check of every extra that scores a different SHAPE, and configs[4]'s verdict
    sx = short["extra_workloads"]
    assert all(v == "error" or v["value"] > 0 for v in sx.values()), sx
    for k in ("mind-small-stress", "mind-large-default", "mind-small-heavy-history"):
        assert sx[k]["auc_diff"] <= 1e-4, (k, sx[k])
    assert "pq-bf16" in short["configs4_inference"] and "pq-fp8" in short["configs4_inference"]
    assert short["detail"]
    # ---- the full document
    line = _detail(detail)
    assert REQUIRED <= set(line) and "cpu_baseline" in line
    assert abs(line["value"] - short["value"]) <= 1e-3 * line["value"]
    roof = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof) and 0 < roof["frac"] < 1
    # round 6: `frac` is the SOLO fraction (single stream: what a kernel trace reproduces); the in-region figure keeps its own name
    assert roof["frac"] == roof["isolated_frac"] and 0 < roof["frac_overlapped"] < 1 and roof["avg_launch_ms_overlapped"] > 0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1
    assert line["auc_match"]["max_abs_metric_diff"] <= line["auc_match"]["tolerance"] and line["valid"] is True
    assert "AUC-matched" in line["metric"]
    rx = line["roofline_xattn"]
    assert rx["bound"] == "hbm" and 0 < rx["frac"] < 1
    # the three Eq. 8 kernels apart: time alone and in the region, algorithmic bytes (each distinct row once)
    for name in ("twin", "l0", "news"):
        e = rx["parts"][name]
        assert e["launches"] > 0 and e["algorithmic_bytes_per_launch"] > 0 and 0 < e["isolated_frac"] < 1
        assert e["frac"] == e["isolated_frac"] and 0 < e["frac_overlapped"] < 1
    # layer 0 of grouped rows reads each GROUP's rows once: its algorithmic bytes are far below a layer >= 1 launch's
    assert rx["parts"]["l0"]["algorithmic_bytes_per_launch"] < 0.6 * rx["parts"]["twin"]["isolated_algorithmic_bytes_per_launch"]
    assert line["setup_ms"] > 0 and line["config"]["news_num"] == 2048
    # the same criterion on a model that ranks: trained weights on the planted-signal dev split vs the imported reference's scores
    tr = line["auc_match_trained"]
    assert tr["rows"] > 70000 and tr["reference"][0] > 0.60 and tr["max_abs_metric_diff"] <= 1e-4 and tr["ranks_equal_fraction"] > 0.999
    assert line["fp16x3_range_overflow"] is False
    # the end-to-end dev run (title tokens -> rank file), here on 1 500 impressions
    e2e = line["e2e"]
    assert e2e["impressions"] == 1500 and e2e["seconds"] > 0 and e2e["rank_file_bytes"] > 0
    assert abs(sum(e2e["breakdown_s"].values()) - e2e["seconds"]) < 0.05 * e2e["seconds"] + 0.01
    ex = line["extra_workloads"]
    assert set(ex) == {"mind-small-stress", "mind-large-default", "mind-small-heavy-history", "mind-small-default/pq-bf16",
                       "mind-small-default/pq-fp8", "mind-small-default/bf16x6", "mind-small-default/train-step",
                       "mind-small-default/reference-batch-1024", "mind-small-default/drop-in"}
    # a step is one launch set of util.LAUNCH_ROWS rows; the reference's own 1024-row chunking is reported next to it
    assert line["config"]["rows_per_step"] == 4096 and line["config"]["reference_dev_batch_rows"] == 1024
    assert ex["mind-small-default/reference-batch-1024"]["rows_per_step"] == 1024
    assert e2e["rows_per_launch_set"] == 4096 and e2e["launch_sets"] <= e2e["reference_batches"]
    assert all(v["value"] > 0 for v in ex.values()), ex
    assert "fp16x3" in line["config"]["projection"] and "two fp16 pieces" in line["config"]["projection_format"]
    assert ex["mind-small-default/bf16x6"]["max_abs_metric_diff_vs_fp32_oracle"] <= 1e-4
    # BASELINE configs[4], inference half — held to what the DEFAULT bench line prints: the drift on the trained, reference-pinned
    # 2 000-impression dev set (the random-click rows of the CPU sample are reported next to it, not asserted: 41 impressions)
    lo = ex["mind-small-default/pq-bf16"]
    assert lo["max_abs_metric_diff_vs_reference_trained_2k"] <= 1e-4 and lo["within_1e-4"] is True
    # pq-fp8: e4m3 storage of P', Q is OVER the reference's 1e-4 on the trained fixture (measured 1.9e-4): the line says so, the
    # mode stays opt-in; a silent improvement (or a regression past 3e-4) fails here so that the line's verdict gets rewritten
    f8 = ex["mind-small-default/pq-fp8"]
    assert f8["within_1e-4"] is False and 1e-4 < f8["max_abs_metric_diff_vs_reference_trained_2k"] <= 3e-4
    assert f8["ranks_equal_fraction_trained_2k"] > 0.98
    assert f8["projection_gemm_result_bytes_per_row"] == 1600 + 2 * 448
    # what the reference's own driver gets from the two-line swap
    di = ex["mind-small-default/drop-in"]
    assert di["rows_per_step"] == 1024 and di["one_stream"]["value"] > 0 and di["three_streams"]["value"] > 0
    # the other adjacency regime: its own rooflines, both Eq. 8 variants timed, scores held to the oracle
    hv = ex["mind-small-heavy-history"]
    assert hv["adjacency_entries_per_node"] > 12 and hv["live_row_fraction"] > 0.7
    assert hv["auc_match"]["max_abs_metric_diff"] <= 1e-4 and len(hv["user_graph_eq8_variants"]) == 3
    # configs[2] / configs[3]: the N > 16 news-table path of these extras is held to the oracle in the same line
    for k in ("mind-small-stress", "mind-large-default"):
        assert ex[k]["auc_match"]["max_abs_metric_diff"] <= 1e-4 and ex[k]["auc_match"]["rows"] >= 64, ex[k]["auc_match"]
    rs = line["roofline_step"]
    assert 0 < rs["frac"] < 1 and rs["floor_ms"] == max(rs["mfma_floor_ms"], rs["hbm_floor_ms"]) and rs["bound"] in ("hbm", "mfma")
    assert set(line["untimed_seconds_before_the_timed_region"]) == {"prewarm", "lane_tuning"}
    assert 0 < ex["mind-small-default/train-step"]["roofline"]["frac"] < 1
    assert ex["mind-small-stress"]["config"]["N"] == 65 and ex["mind-large-default"]["config"]["N"] == 26
    for k in ("mind-small-stress", "mind-large-default", "mind-small-heavy-history"):        # configs[2] / configs[3]: their own rooflines
        assert 0 < ex[k]["roofline"]["frac"] < 1 and 0 < ex[k]["roofline_xattn"]["frac"] < 1 and ex[k]["kernel_ms_per_step_single_stream"]


def test_bench_two_ranks_sum_their_rows():
    env = dict(os.environ, DIGAT_BENCH_TEST_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2", "--impressions", "600", "--news", "2048"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=420, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _last_json_line(res.stdout)
    assert REQUIRED <= set(line)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert "cpu_baseline" not in line or line["cpu_baseline"] is None       # rank 0 at N=1 only
    assert line["config"]["N"] == 26 and "MIND-large" in line["config"]["workload"]     # BASELINE configs[3]'s shape at N > 1
    assert line["auc_match"]["max_abs_metric_diff"] <= 1e-4
    assert len(line["per_rank_impressions_per_s"]) == 2 and all(v > 0 for v in line["per_rank_impressions_per_s"])
    # value = the sum of what the ranks scored over the slowest rank's clock; the line carries its own one-GPU reference of the
    # SAME (MIND-large) workload, so that nobody divides it by the MIND-small N = 1 headline
    assert line["value"] <= sum(line["per_rank_impressions_per_s"]) * 1.001
    n1 = line["n1_same_workload"]
    assert n1["value"] > 0 and abs(line["scaling_efficiency"] - line["value"] / (2 * n1["value"])) < 1e-5      # the compact line rounds its numbers
    assert len(line["devices"]) == 2 and len(line["all_gather_ms_by_rank"]) == 2 and all(v > 0 for v in line["all_gather_ms_by_rank"])


def test_bench_two_gpus_starts_its_own_launcher():
    """``python bench.py --gpus 2`` with no outer launcher (how a driver may start the scaling runs): the parent starts
    torch.distributed.run as a child before touching the GPU and relays its JSON line and exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(DIGAT_BENCH_TEST_SHARED_GPU="1")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "2", "--impressions", "600", "--news", "2048"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=420, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _last_json_line(res.stdout)
    assert REQUIRED <= set(line)
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["ranks_in_process_group"] == 2
    assert line["config"]["backend"] == "gloo"            # the one-GPU test hook; "nccl" (RCCL) on a multi-GPU node
    assert line["auc_match"]["max_abs_metric_diff"] <= 1e-4


def test_bench_train_mode_two_ranks_ddp():
    """--mode train under torch.distributed.run: DistributedDataParallel around Model.forward / the HIP backward."""
    env = dict(os.environ, DIGAT_BENCH_TEST_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29733", "bench.py", "--gpus", "2", "--mode", "train", "--steps", "3", "--warmup", "1",
           "--impressions", "400", "--news", "2048"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=420, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _last_json_line(res.stdout)
    assert line["n_gpus"] == 2 and line["unit"] == "rows/s" and line["value"] > 0 and "ddp2" in line["config"]["parallelism"]
    import math
    assert math.isfinite(line["final_loss"])
    # the gradient all-reduce as DistributedDataParallel times it, and the ranks that really rendezvoused
    assert line["ranks_in_process_group"] == 2 and line["backend"] == "gloo"
    assert line["ddp_timers"] and ("avg_backward_comm_time" in line["ddp_timers"] or "error" in line["ddp_timers"])


def test_bench_eight_ranks_dress_rehearsal_on_one_gpu():
    """Round 6: the driver's 8-GPU scaling run has never had hardware.  The same command line — ``bench.py --gpus 8``,
    the launcher started by bench.py itself before any GPU call — with all eight ranks on cuda:0 and gloo collectives
    (DIGAT_BENCH_TEST_SHARED_GPU): rank count, row shards, ports, the all-gather of the scores and the summed line, before the first
    real 8-GPU lease.  Eight ranks, eight shards, every rank's rows in the sum."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(DIGAT_BENCH_TEST_SHARED_GPU="1")
    cmd = [sys.executable, "bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1", "--impressions", "300", "--news", "2048"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _last_json_line(res.stdout)
    assert REQUIRED <= set(line)
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["ranks_in_process_group"] == 8 and line["config"]["backend"] == "gloo"
    assert "dp8" in line["config"]["parallelism"] and line["config"]["N"] == 26            # BASELINE configs[3]: MIND-large shapes at N > 1
    per = line["per_rank_impressions_per_s"]
    assert len(per) == 8 and all(v > 0 for v in per) and len(line["devices"]) == 8
    assert len(line["all_gather_ms_by_rank"]) == 8 and all(v > 0 for v in line["all_gather_ms_by_rank"])
    assert line["value"] <= sum(per) * 1.001                       # the ranks' rows over the slowest rank's clock
    assert line["auc_match"]["max_abs_metric_diff"] <= 1e-4
    n1 = line["n1_same_workload"]
    assert n1["value"] > 0 and abs(line["scaling_efficiency"] - line["value"] / (8 * n1["value"])) < 1e-5


def test_bench_train_mode_eight_ranks_ddp_rehearsal():
    """... and --mode train with eight DistributedDataParallel ranks on the one GPU (gloo): the gradient buckets of eight ranks meet."""
    env = dict(os.environ, DIGAT_BENCH_TEST_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29735", "bench.py", "--gpus", "8", "--mode", "train", "--steps", "2", "--warmup", "1",
           "--impressions", "400", "--news", "2048"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _last_json_line(res.stdout)
    import math
    assert line["n_gpus"] == 8 and line["unit"] == "rows/s" and line["value"] > 0 and "ddp8" in line["config"]["parallelism"]
    assert math.isfinite(line["final_loss"]) and line["ranks_in_process_group"] == 8 and line["backend"] == "gloo"


def test_bench_train_mode_with_the_msa_news_encoder():
    """--mode train --train-news-encoder msa: the reference's full step (title text -> MSA -> graph encoder), native forward and
    backward end to end."""
    cmd = [sys.executable, "bench.py", "--mode", "train", "--train-news-encoder", "msa", "--steps", "3", "--warmup", "1",
           "--impressions", "400", "--news", "2048"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert res.returncode == 0, res.stderr[-2000:]
    line = _last_json_line(res.stdout)
    import math
    assert line["unit"] == "rows/s" and line["value"] > 0 and math.isfinite(line["final_loss"])
    assert "MSA" in line["config"]["news_encoder"]
    assert line["roofline"]["bound"] == "mfma" and 0 < line["roofline"]["frac"] < 1


def test_bench_e2e_mode_prints_its_line():
    """--mode e2e (README: the dev run from title tokens to the rank file) as a stand-alone invocation: one JSON line, seconds."""
    cmd = [sys.executable, "bench.py", "--mode", "e2e", "--e2e-impressions", "600", "--news", "4096"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["unit"] == "s" and line["higher_is_better"] is False and line["value"] > 0
    assert line["steps"] == line["e2e"]["launch_sets"] > 0 and line["ms_per_step"] > 0
    assert line["e2e"]["fp16x3_range_overflow"] is False
