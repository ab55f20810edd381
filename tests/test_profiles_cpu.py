"""CPU suite over the committed evidence under profiles/: the numbers bench.py prints have to follow from the counter and trace
summaries of the same round (round 4's line priced the layer-0 Eq. 8 launch at 4.7x the bytes the counters saw).  No GPU, no
oracle: JSON against JSON."""
import glob
import json
import os
import re

import pytest

from conftest import REPO

PROFILES = os.path.join(REPO, "profiles")


def _newest(pattern):
    """The newest round that has BOTH a counter summary and a bench document (r05_final_pmc.json + r05_final_bench.json)."""
    rounds = sorted({re.match(r"(r\d+)_", os.path.basename(p)).group(1) for p in glob.glob(os.path.join(PROFILES, "r*_final_pmc.json"))}, reverse=True)
    for r in rounds:
        path = os.path.join(PROFILES, pattern.format(r=r))
        if os.path.exists(path):
            return r, path
    return None, None


def _bench_doc(path):
    text = open(path).read().strip()
    doc = json.loads(text.splitlines()[-1])
    return doc


def test_algorithmic_bytes_do_not_exceed_the_counter_bytes():
    """For every Eq. 8 kernel the line prices (roofline_xattn.parts): algorithmic HBM bytes per launch <= 1.05 x what the
    FETCH_SIZE / WRITE_SIZE counters saw for that kernel in the same round's collection.  Algorithmic = every distinct row once; the
    counters also see re-reads (a group's rows fetched by several XCDs' L2s, Infinity-Cache hits), never fewer bytes."""
    r, pmc_path = _newest("{r}_final_pmc.json")
    assert r is not None, "no profiles/rNN_final_pmc.json"
    bench_path = os.path.join(PROFILES, f"{r}_final_bench.json")
    if not os.path.exists(bench_path):
        pytest.skip(f"profiles/{r}_final_bench.json not collected")
    doc = _bench_doc(bench_path)
    parts = (doc.get("roofline_xattn") or {}).get("parts")
    if not parts:
        pytest.skip(f"profiles/{r}_final_bench.json predates the per-kernel Eq. 8 accounting (round 5)")
    kernels = json.load(open(pmc_path))["kernels"]
    symbol = {"twin": "xattn_sparse_twin", "l0": "xattn_sparse_l0", "news": "xattn_small_lds"}
    checked = 0
    for name, e in parts.items():
        if name not in symbol:
            continue
        match = [v for k, v in kernels.items() if symbol[name] in k]
        assert match, f"{symbol[name]} is priced in the line but absent from {os.path.basename(pmc_path)}"
        counted = match[0]["hbm_bytes_mean"]
        alg = e.get("isolated_algorithmic_bytes_per_launch", e["algorithmic_bytes_per_launch"])
        assert alg <= 1.05 * counted, f"{name}: algorithmic {alg / 1e6:.1f} MB per launch > 1.05 x {counted / 1e6:.1f} MB counted"
        assert alg >= 0.4 * counted, f"{name}: algorithmic {alg / 1e6:.1f} MB is implausibly far below the {counted / 1e6:.1f} MB counted"
        checked += 1
    assert checked == 3
    # ... and the whole step: compulsory bytes of the step <= what the counters saw over every kernel of a step
    rs = doc.get("roofline_step")
    assert rs and rs["compulsory_hbm_bytes_per_step"] > 0


def test_solo_fractions_follow_from_the_solo_kernel_table():
    """The per-kernel solo fractions of the line = algorithmic bytes / the kernel's average duration in the same round's single-stream
    kernel table (profiles/rNN_default_solo_kernels.txt), within the run-to-run spread of two runs on different boxes (15 %)."""
    r, pmc_path = _newest("{r}_final_pmc.json")
    bench_path, solo_path = os.path.join(PROFILES, f"{r}_final_bench.json"), os.path.join(PROFILES, f"{r}_default_solo_kernels.txt")
    if not (os.path.exists(bench_path) and os.path.exists(solo_path)):
        pytest.skip("round's bench document or solo kernel table not collected")
    parts = (_bench_doc(bench_path).get("roofline_xattn") or {}).get("parts")
    if not parts:
        pytest.skip("bench document predates the per-kernel Eq. 8 accounting")
    table = {}
    for line in open(solo_path):
        m = re.match(r"(?:void )?(\S+?)(?:<[^>]*>)?\s+grid\s+(\d+)\s+wg\s+\d+\s+n/step\s+([\d.]+)\s+avg\s+([\d.]+) us", line.strip())
        if m:
            table.setdefault(m.group(1), []).append((float(m.group(3)), float(m.group(4))))
    symbol = {"twin": "xattn_sparse_twin_kernel", "l0": "xattn_sparse_l0_kernel", "news": "xattn_small_lds_kernel"}
    for name, sym in symbol.items():
        rows = table.get(sym)
        assert rows, f"{sym} not in {os.path.basename(solo_path)}"
        avg_us = sum(n * us for n, us in rows) / sum(n for n, _ in rows)
        e = parts[name]
        assert abs(e["isolated_avg_launch_us"] - avg_us) <= 0.15 * avg_us, (name, e["isolated_avg_launch_us"], avg_us)
        frac = e.get("isolated_algorithmic_bytes_per_launch", e["algorithmic_bytes_per_launch"]) / (avg_us * 1e-6) / 8e12
        assert abs(frac - e["isolated_frac"]) <= 0.15 * frac + 0.01, (name, frac, e["isolated_frac"])


def test_fetch_size_correction_is_calibrated_on_known_byte_counts():
    """Round 6: the `traffic` figures are 2 x FETCH_SIZE + WRITE_SIZE.  The factor is not applied blindly any
    more: profiles/r06_fetch_calib.json holds it measured on known byte counts in every access shape of the product kernels
    (tools/exp/fetch_calib.hip): streamed and row-shaped float4 loads, LDS-DMA, the GEMM's 128-byte row pieces, gathered rows."""
    doc = json.load(open(os.path.join(PROFILES, "r06_fetch_calib.json")))
    shapes = doc["shapes"]
    assert {"stream_f4", "stream_dma", "rows128_dma", "rows1600_f4", "rows1600_f4g", "store_f4"} <= set(shapes)
    for name, v in shapes.items():
        lo, hi = (0.98, 1.02) if name == "store_f4" else (1.7, 2.02)       # reads: half-counted, up to 12 % of line straddling on top
        assert lo <= v["factor"] <= hi, (name, v["factor"])
        assert abs(v["known_bytes"] / (v["counter_kb"] * 1024.0) - v["factor"]) < 1e-9


def test_counter_bytes_bracket_the_algorithmic_bytes_of_every_hot_kernel():
    """counter bytes >= 0.95 x algorithmic AND <= 1.6 x algorithmic for the kernels whose algorithmic bytes are counted on the device
    (the three Eq. 8 kernels): a blind factor of two on a kernel that does not deserve it would show here."""
    r, pmc_path = _newest("{r}_final_pmc.json")
    bench_path = os.path.join(PROFILES, f"{r}_final_bench.json")
    if not os.path.exists(bench_path):
        pytest.skip(f"profiles/{r}_final_bench.json not collected")
    parts = (_bench_doc(bench_path).get("roofline_xattn") or {}).get("parts")
    if not parts:
        pytest.skip("bench document predates the per-kernel Eq. 8 accounting")
    kernels = json.load(open(pmc_path))["kernels"]
    symbol = {"twin": "xattn_sparse_twin", "l0": "xattn_sparse_l0", "news": "xattn_small_lds"}
    for name, sym in symbol.items():
        counted = [v for k, v in kernels.items() if sym in k][0]["hbm_bytes_mean"]
        alg = parts[name].get("isolated_algorithmic_bytes_per_launch", parts[name]["algorithmic_bytes_per_launch"])
        assert 0.95 * alg <= counted <= 1.6 * alg, f"{name}: counted {counted / 1e6:.1f} MB vs algorithmic {alg / 1e6:.1f} MB"
