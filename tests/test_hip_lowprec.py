"""GPU suite: the 2 000-impression reference-pinned dev set (fp32 path), and BASELINE configs[4]'s reduced-precision Eq. 8 —
P' = K3 + K1 and Q = K2 of the user graph stored in bf16 (``projection_mode = "pq-bf16"``; the reference's own
"faster inference" idea is a quantised K3 + K1 + K2, README.md:62-66, with "no AUC/MRR/nDCG degradation accurate to 1e-4").

tests/golden/devset_2k.npz: 2 000 impressions / 74 239 rows at the MIND-small default shapes, scored by the imported
reference (oracle/make_golden.py devset_2k): scores, per-row ranks, evaluate.scoring's four metrics.
"""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build_2k():
    from digat_amd import synthetic, util
    from digat_amd.model import Model, PrecomputedNewsEncoder
    fx = load_golden("devset_2k.npz")
    spec = synthetic.SynthSpec(news_num=4096, sag_neighbors=3, sag_hops=2, impressions=2000, seed=47)
    corpus = synthetic.make_corpus(spec)
    chk = (float(corpus.news_embedding.astype(np.float64).sum()) + float(corpus.user_graph.sum()) + float(corpus.news_graph.sum())
           + float(corpus.row_candidate.astype(np.float64).sum()))
    assert abs(chk - float(fx["input_checksum"])) <= 1e-6 * abs(chk), "synthetic generator drifted from the fixture's"
    L = int(fx["depth"])
    state = synthetic.make_state_dict(spec.embedding_dim, spec.category_num, L, seed=spec.seed + 1, bias_std=0.05)
    cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=spec.news_graph_size,
                                max_history_num=spec.max_history_num, category_num=spec.category_num, graph_depth=L, dropout_rate=0.2)
    model = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(corpus.news_embedding)))
    model.graph_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(DEV).eval()
    dc = util.DeviceCorpus.from_numpy(corpus, torch.device(DEV))
    return fx, corpus, model, dc


def report(scores, fx, what):
    ref = fx["scores"].astype(np.float64)
    rel = np.abs(scores - ref) / (np.abs(ref) + 1e-3)
    print(f"\n[{what}] scores vs the reference: max rel {rel.max():.3e}, mean rel {rel.mean():.3e}, "
          f"rms of scores {np.sqrt((ref ** 2).mean()):.1f}")
    return rel


def test_devset_2k_fp32_metrics_match_the_reference():
    """74 k rows are enough for near-ties between candidates to occur: AUC / MRR / nDCG@5 / nDCG@10 within 1e-4 of
    evaluate.scoring on the reference's scores, scores within 1e-4 relative, ranks equal except at near-ties."""
    from digat_amd import evaluate, util
    fx, corpus, model, dc = build_2k()
    scores, metrics = util.compute_scores(model, dc, 1024, labels=corpus.row_label)
    rel = report(scores, fx, "fp32 (bf16x6 projections)")
    assert rel.max() < 1e-4
    np.testing.assert_allclose(metrics, fx["metrics"], rtol=0, atol=1e-4)
    ranks = evaluate.impression_ranks(scores, corpus.row_impression)
    assert (np.asarray(ranks) == fx["ranks"].astype(np.int64)).mean() > 0.999


def quantize_e4m3_strips(x, strip=80):
    """The GEMM epilogue's block-scaled OCP e4m3 storage of P' / Q, restated in torch: one block per (row, 80-channel strip),
    scale = absmax / 448 (fp32), codes = round-to-nearest-even e4m3 of value / scale."""
    shp = x.shape
    xs = x.reshape(*shp[:-1], shp[-1] // strip, strip)
    amax = xs.abs().amax(-1, keepdim=True)
    scale = torch.where(amax > 0, amax * np.float32(1.0 / 448.0), torch.ones_like(amax))
    q = (xs * (1.0 / scale)).to(torch.float8_e4m3fn).to(torch.float32)
    return (q * scale).reshape(shp)


@pytest.mark.parametrize("pq,fmt", [(0, 0), (1, 0), (2, 0), (2, 1)])
def test_low_precision_eq8_layer_against_its_emulation(pq, fmt):
    """digat_xattn_fwd_lowprec: ONE Eq. 8 layer of MIND-shaped user graphs with P' = K3 + K1 and Q = K2 stored in fp32 / bf16 /
    block-scaled e4m3, against the oracle's unfused Eq. 8 with the same quantiser applied to the two operands before the
    broadcast-add.  fp32: 1e-5.  Quantised: the projections here and the oracle's differ in the last fp32 bits, so a value
    that sits on a rounding boundary may take the neighbouring code (1/16 of its magnitude for e4m3) — the outputs then
    agree to ~1e-3 of their scale in the worst element and much better on average, while quantised and unquantised
    outputs differ by 10-100x more: layout, scales and strip mapping are what this pins."""
    import torch.nn.functional as F
    from digat_amd import _lib, synthetic
    from oracle import digat_oracle as O
    B, N, H, C, d, L = 48, 10, 50, 17, 400, 1
    state = synthetic.make_state_dict(d, C, L, seed=171, bias_std=0.05)
    batch = synthetic.make_encoder_batch(B, N, H, C, d, seed=172, empty_history_rows=(2,))
    batch["user_graph"][5, 3, :] = False                # a row without any entry: uniform over all nodes (E1)
    p = O.as_params(state)
    tb = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in batch.items()}
    quant = {0: (lambda x: x), 1: (lambda x: x.to(torch.bfloat16).to(torch.float32)), 2: quantize_e4m3_strips}[pq]
    pre = "user_graph_attention_"
    with torch.no_grad():
        Xu = O.user_nodes(p, tb["user_news_embedding"])
        c_n = O.news_graph_context(p, tb["news_graph_embeddings"], tb["news_graph_mask"])
        U = Xu.shape[1]
        h = O._linear(Xu, p, f"{pre}W.0")
        K3 = O._linear(c_n, p, f"{pre}ffn3.0").view(B, 1, d)
        Pq = quant(K3 + O._linear(Xu, p, f"{pre}ffn1.0")).unsqueeze(1)
        Qq = quant(O._linear(Xu, p, f"{pre}ffn2.0")).unsqueeze(2)
        s = F.linear(F.relu(Pq + Qq), p[f"{pre}a.0.weight"]).squeeze(3)
        alpha = F.softmax(F.leaky_relu(s, 0.2).masked_fill(tb["user_graph"] == 0, O.MASK_FILL), dim=2)
        want = F.relu(torch.bmm(alpha, h)) + Xu
        exact = O.cross_graph_attention(p, "user", 0, Xu, tb["user_graph"], c_n)
    Lb = _lib.lib()
    dev = torch.device(DEV)
    w = {k: torch.from_numpy(state[f"{pre}{k}"]).to(dev).contiguous() for k in
         ("W.0.weight", "W.0.bias", "ffn1.0.weight", "ffn2.0.weight", "ffn3.0.weight", "ffn3.0.bias", "a.0.weight")}
    wsplit = torch.empty(Lb.digat_split_weights_bytes(3 * d, d), dtype=torch.uint8, device=dev)
    _lib.check(Lb.digat_split_proj_weights(w["W.0.weight"].data_ptr(), w["ffn1.0.weight"].data_ptr(), w["ffn2.0.weight"].data_ptr(), d,
                                           wsplit.data_ptr(), fmt, _lib.stream_ptr()), "split")
    dX, dA, dc = Xu.to(dev).contiguous(), tb["user_graph"].to(dev).contiguous(), c_n.to(dev).contiguous()
    out = torch.full((B, U, d), float("nan"), device=dev)
    nbytes = Lb.digat_xattn_workspace_bytes(B, U, d)
    ws = torch.full((nbytes,), 255, dtype=torch.uint8, device=dev)          # NaN patterns: nothing may rely on the scratch
    _lib.check(Lb.digat_xattn_fwd_lowprec(dX.data_ptr(), dA.data_ptr(), dc.data_ptr(), w["W.0.weight"].data_ptr(), w["W.0.bias"].data_ptr(),
                                          w["ffn1.0.weight"].data_ptr(), w["ffn2.0.weight"].data_ptr(), w["ffn3.0.weight"].data_ptr(),
                                          w["ffn3.0.bias"].data_ptr(), w["a.0.weight"].data_ptr(), wsplit.data_ptr(), fmt, out.data_ptr(),
                                          B, U, d, pq, ws.data_ptr(), nbytes, _lib.stream_ptr()), "digat_xattn_fwd_lowprec")
    torch.cuda.synchronize()
    got = out.cpu()
    assert torch.isfinite(got).all()
    scale = float(want.abs().max())
    err = (got - want).abs()
    gap = (exact - want).abs()
    print(f"\n[Eq. 8 layer, pq={pq}, format={fmt}] vs emulation: max {err.max():.3e} mean {err.mean():.3e} (output scale {scale:.2f}); "
          f"quantised vs exact: max {gap.max():.3e} mean {gap.mean():.3e}")
    if pq == 0:
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)
    else:
        assert err.mean() <= 0.05 * max(float(gap.mean()), 1e-7) + 1e-6, (err.mean(), gap.mean())
        assert err.max() <= 2e-3 * scale
        assert gap.max() > 10 * 1e-5, "the quantiser changed nothing: the test does not test"


def test_low_precision_eq8_entry_rejects_what_it_cannot_run():
    from digat_amd import _lib
    Lb = _lib.lib()
    t = torch.zeros(64, device=DEV)
    args = [t.data_ptr()] * 11
    # d % 80 != 0; n <= 16; fewer than 2048 node rows; unknown pq
    assert Lb.digat_xattn_fwd_lowprec(*args, 0, t.data_ptr(), 64, 67, 64, 2, t.data_ptr(), 1 << 40, None) == 2
    assert Lb.digat_xattn_fwd_lowprec(*args, 0, t.data_ptr(), 1024, 10, 400, 2, t.data_ptr(), 1 << 40, None) == 2
    assert Lb.digat_xattn_fwd_lowprec(*args, 0, t.data_ptr(), 8, 67, 400, 2, t.data_ptr(), 1 << 40, None) == 2
    assert Lb.digat_xattn_fwd_lowprec(*args, 0, t.data_ptr(), 64, 67, 400, 3, t.data_ptr(), 1 << 40, None) == 1
    assert Lb.digat_xattn_fwd_lowprec(*args, 0, t.data_ptr(), 64, 67, 400, 2, t.data_ptr(), 16, None) == 3


@pytest.mark.parametrize("mode", ["pq-bf16", "pq-bf16-x1", "pq-fp8"])
def test_bf16_eq8_operands_keep_the_metrics(mode):
    """configs[4]: P' and Q of the user graph's layers >= 1 in bf16 (and, "-x1", computed with one bf16 product).  The metric
    drift against the reference stays within the reference's own 1e-4 criterion; element-wise the scores move by ~1e-4."""
    from digat_amd import util
    fx, corpus, model, dc = build_2k()
    base, base_metrics = util.compute_scores(model, dc, 1024, labels=corpus.row_label)
    model.graph_encoder.projection_mode = mode
    scores, metrics = util.compute_scores(model, dc, 1024, labels=corpus.row_label)
    assert not np.array_equal(scores, base), "the reduced-precision path did not run"
    rel = report(scores, fx, mode)
    drift = np.abs(np.array(metrics) - fx["metrics"])
    print(f"[{mode}] metric drift vs the reference {np.round(drift, 7)}; vs this library's fp32 path "
          f"{np.round(np.abs(np.array(metrics) - np.array(base_metrics)), 7)}")
    assert rel.mean() < (5e-4 if mode == "pq-bf16" else 2e-3) and rel.max() < (3e-2 if mode != "pq-fp8" else 0.3)
    if mode == "pq-bf16":
        assert drift.max() <= 1e-4, drift
    else:
        # one bf16 product for P and Q / block-scaled e4m3 on Xavier weights and random clicks (logits of rms ~600, AUC 0.5):
        # reported, held to a looser bound; the yardstick for "AUC-matched" is the trained model below
        assert drift.max() <= (5e-4 if mode != "pq-fp8" else 1e-3), drift


@pytest.mark.parametrize("name", ["devset_default.npz", "devset_tiny.npz"])
def test_bf16_eq8_operands_on_the_small_devsets(name):
    """The same on the two small reference-pinned dev sets (24 and 200 impressions).  devset_tiny (d = 64, U = 15) has no
    launch large enough for the bf16 segments (>= 2048 projected rows per layer are needed): the mode must then be a no-op."""
    from test_hip_parity import DEVSETS
    from digat_amd import synthetic, util
    from digat_amd.model import Model, PrecomputedNewsEncoder
    fx = load_golden(name)
    spec = synthetic.SynthSpec(**DEVSETS[name])
    corpus = synthetic.make_corpus(spec)
    L = int(fx["depth"])
    state = synthetic.make_state_dict(spec.embedding_dim, spec.category_num, L, seed=spec.seed + 1, bias_std=0.05)
    cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=spec.news_graph_size,
                                max_history_num=spec.max_history_num, category_num=spec.category_num, graph_depth=L, dropout_rate=0.2)
    model = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(corpus.news_embedding)))
    model.graph_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(DEV).eval()
    model.graph_encoder.projection_mode = "pq-bf16"
    dc = util.DeviceCorpus.from_numpy(corpus, torch.device(DEV))
    scores, metrics = util.compute_scores(model, dc, 1024, labels=corpus.row_label)
    np.testing.assert_allclose(metrics, fx["metrics"], rtol=0, atol=1e-4)
    ref = fx["scores"]
    print(f"\n[{name} pq-bf16] max rel score diff {np.max(np.abs(scores - ref) / (np.abs(ref) + 1e-3)):.3e}")


@pytest.mark.parametrize("mode", ["pq-bf16", "pq-fp8"])
def test_reduced_precision_operands_on_the_news_graph_too(mode):
    """MIND-large shapes (N = 26: the news graph takes the sparse Eq. 8 kernel, so ITS P', Q are stored in the reduced format at
    every layer, next to the user graph's layers >= 1): finite scores, really different from the fp32-grade run, and close to it
    (random weights, logits of rms ~600: mean relative difference below 1e-3 for bf16, 5e-3 for e4m3)."""
    from digat_amd import synthetic, util
    from digat_amd.model import Model, PrecomputedNewsEncoder
    spec = synthetic.SynthSpec(news_num=1024, sag_neighbors=5, sag_hops=2, category_num=18, impressions=90, mean_candidates=30.0,
                               max_candidates=60, seed=141)
    corpus = synthetic.make_corpus(spec)
    L = 3
    state = synthetic.make_state_dict(spec.embedding_dim, spec.category_num, L, seed=142, bias_std=0.05)
    cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=spec.news_graph_size,
                                max_history_num=spec.max_history_num, category_num=spec.category_num, graph_depth=L, dropout_rate=0.1)
    model = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(corpus.news_embedding)))
    model.graph_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(DEV).eval()
    dc = util.DeviceCorpus.from_numpy(corpus, torch.device(DEV))
    base, _ = util.compute_scores(model, dc, 1024, labels=corpus.row_label)
    assert model.graph_encoder.resolved_xattn_mode("news") == "sparse"
    model.graph_encoder.projection_mode = mode
    scores, _ = util.compute_scores(model, dc, 1024, labels=corpus.row_label)
    assert np.isfinite(scores).all() and not np.array_equal(scores, base)
    rel = np.abs(scores - base) / (np.abs(base) + 1e-3)
    print(f"\n[{mode}, N = {spec.news_graph_size}] vs the fp32-grade run: mean rel {rel.mean():.3e}, max rel {rel.max():.3e}")
    assert rel.mean() < (1e-3 if mode == "pq-bf16" else 5e-3)


def build_trained():
    from conftest import planted_devset
    from digat_amd import util
    from digat_amd.model import Model, PrecomputedNewsEncoder
    fx, corpus, state = planted_devset()
    spec = corpus.spec
    cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=spec.news_graph_size,
                                max_history_num=spec.max_history_num, category_num=spec.category_num, graph_depth=int(fx["depth"]),
                                dropout_rate=0.2)
    model = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(corpus.news_embedding)))
    model.graph_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(DEV).eval()
    dc = util.DeviceCorpus.from_numpy(corpus, torch.device(DEV))
    return fx, corpus, model, dc


@pytest.mark.parametrize("mode,metric_tol,rank_match", [("bf16x6", 1e-4, 0.9995), ("fp16x3", 1e-4, 0.9995), ("auto", 1e-4, 0.9995),
                                                        ("pq-bf16", 1e-4, 0.99), ("fp32", 1e-4, 0.9995), ("pq-fp8", 3e-4, 0.98)])
def test_trained_model_metrics_match_the_reference(mode, metric_tol, rank_match):
    """"AUC-matched" on a model that RANKS: trained weights on the planted-signal corpus (reference AUC 0.644, logits of rms ~10 —
    near-ties between candidates as a trained model has them, not the widely spread scores of Xavier weights on random clicks).
    Every operand format of the projections keeps AUC / MRR / nDCG@5 / nDCG@10 within the reference's own 1e-4 of
    evaluate.scoring on the reference's scores; the fp32-grade formats also reproduce the scores to 2e-5 and > 99.95 % of
    the per-row ranks.  "pq-fp8" (block-scaled e4m3 P', Q: the fp8 half of configs[4]) is held to what it MEASURES on this
    fixture — AUC 4.1e-5, MRR 1.7e-4, nDCG@5 1.9e-4, nDCG@10 6.5e-5, 98.8 % of the ranks — which is over the reference's 1e-4: the
    mode is opt-in and never what "auto" resolves to (three mantissa bits per operand: a finer block does not help, the error is
    the code's relative step, not the scale's)."""
    from digat_amd import evaluate, util
    fx, corpus, model, dc = build_trained()
    model.graph_encoder.projection_mode = mode
    scores, metrics = util.compute_scores(model, dc, 1024, labels=corpus.row_label)
    ref = fx["scores"].astype(np.float64)
    err = np.abs(scores - ref)
    drift = np.abs(np.array(metrics) - fx["metrics"])
    ranks = np.asarray(evaluate.impression_ranks(scores, corpus.row_impression))
    same = float((ranks == fx["ranks"].astype(np.int64)).mean())
    print(f"\n[trained, {mode} -> {model.graph_encoder.resolved_projection_mode()}] reference metrics {np.round(fx['metrics'], 6)}; "
          f"metric drift {np.round(drift, 8)}; scores: max abs diff {err.max():.3e}, rms of scores {np.sqrt((ref ** 2).mean()):.2f}; "
          f"ranks equal {same:.5f}")
    assert metrics[0] > 0.60
    assert drift.max() <= metric_tol, drift
    assert same >= rank_match, same
    if not mode.startswith("pq-"):
        assert err.max() <= 2e-5 * max(1.0, np.abs(ref).max()), err.max()


def test_fp16x3_projections_keep_scores_and_metrics():
    """projection_mode "fp16x3" (split images of format DIGAT_GEMM_F16X3): every matrix-core operand as two fp16 pieces, three products.  On the
    reference-pinned 2 000-impression dev set the scores stay within 1e-4 relative of the reference's and the metrics within
    1e-5 — a tenth of the reference's criterion — while the projections take 0.7x the time."""
    from digat_amd import _lib, util
    fx, corpus, model, dc = build_2k()
    model.graph_encoder.projection_mode = "bf16x6"
    base, base_metrics = util.compute_scores(model, dc, 1024, labels=corpus.row_label)
    assert not model.graph_encoder._params().flags & _lib.PARAMS_GEMM_F16X3
    model.graph_encoder.projection_mode = "fp16x3"
    scores, metrics = util.compute_scores(model, dc, 1024, labels=corpus.row_label)
    assert model.graph_encoder._params().flags & _lib.PARAMS_GEMM_F16X3
    assert not np.array_equal(scores, base), "the fp16x3 path did not run"
    rel = report(scores, fx, "fp16x3")
    drift = np.abs(np.array(metrics) - fx["metrics"])
    print(f"[fp16x3] metric drift vs the reference {np.round(drift, 8)}; scores vs this library's bf16x6 path: max rel "
          f"{np.max(np.abs(scores - base) / (np.abs(base) + 1e-3)):.3e}")
    assert rel.max() < 1e-4 and drift.max() <= 1e-5, (rel.max(), drift)


@pytest.mark.parametrize("M,N,K,scale", [(4100, 400, 400, 1.0), (34304, 1200, 400, 1.0), (4100, 400, 400, 100.0), (2500, 160, 72, 0.01),
                                         (4100, 400, 400, 1e-3), (4100, 400, 400, 1e-4)])
def test_fp16x3_linear_error_against_fp64(M, N, K, scale):
    """The two-piece fp16 product against an fp64 product, next to the fp32-MFMA kernel (an exact k-ordered fp32 fma chain) on the
    same data: mean and max error at most 1.1x the chain's (measured 0.75x / 0.7x at unit scale, 1.0x for |x| ~ 0.01).  Below
    |x| ~ 1e-3 the low pieces are fp16 subnormals (quantum 2^-24 after the format's 2^4 scaling = 3.7e-9 of x): the error then
    has an ABSOLUTE floor — K terms of 2^-29 |w| each — which is reported and bounded, not hidden (the chain's own error shrinks
    with x; the format's does not)."""
    from digat_amd import _lib
    rng = np.random.default_rng(M + N + K)
    x = (rng.standard_normal((M, K)) * scale).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    want = x.astype(np.float64) @ w.astype(np.float64).T + b
    xd, wd, bd = (torch.from_numpy(a).to(DEV) for a in (x, w, b))
    L = _lib.lib()
    y16 = torch.full((M, N), float("nan"), device=DEV)
    y32 = torch.full((M, N), float("nan"), device=DEV)
    ws = torch.empty(L.digat_split_weights_bytes(N, K), dtype=torch.uint8, device=DEV)
    _lib.check(L.digat_linear_f32x3(xd.data_ptr(), K, wd.data_ptr(), bd.data_ptr(), y16.data_ptr(), N, M, N, K, ws.data_ptr(),
                                    _lib.GEMM_F16X3, _lib.stream_ptr()), "digat_linear_f32x3")
    _lib.check(L.digat_linear_f32(xd.data_ptr(), K, wd.data_ptr(), bd.data_ptr(), y32.data_ptr(), N, M, N, K, _lib.stream_ptr()),
               "digat_linear_f32")
    torch.cuda.synchronize()
    e16 = np.abs(y16.cpu().numpy().astype(np.float64) - want)
    e32 = np.abs(y32.cpu().numpy().astype(np.float64) - want)
    print(f"\n[fp16x3 {M}x{N}x{K} x{scale}] mean {e16.mean():.3e} (fp32 chain {e32.mean():.3e}), max {e16.max():.3e} ({e32.max():.3e})")
    assert np.isfinite(e16).all()
    if scale >= 0.01:
        assert e16.mean() <= 1.1 * e32.mean() + 1e-9 and e16.max() <= 1.1 * e32.max() + 1e-8
    else:
        # subnormal low pieces: an absolute floor of ~sqrt(K) * 2^-29 * |w| per output (|w| ~ 1 / sqrt(K): ~2e-9), i.e. relative
        # to outputs of size `scale` 2e-6 at 1e-3 and 2e-5 at 1e-4 — fp32-grade no longer, still far inside the 1e-4 the path
        # is held to; "auto" therefore wants features of ordinary size (news representations: 0.1 .. 10)
        floor = np.sqrt(K) * 2.0 ** -29 / np.sqrt(K) * 4
        print(f"    relative to the outputs' scale {scale}: mean {e16.mean() / scale:.2e}, max {e16.max() / scale:.2e}; floor bound {floor:.2e}")
        assert e16.mean() <= e32.mean() + floor and e16.max() <= e32.max() + 8 * floor
