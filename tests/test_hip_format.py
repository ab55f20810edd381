"""GPU suite: the matrix-core operand format as a property of the split image (no process-wide state), the fp16x3 range
flag and its fallback, an evaluation and a training step on two host threads, and the dev evaluation of a model whose news
encoder is being trained."""
import threading
import types
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _small_world(seed=5, d=80, news_num=512, impressions=96, scale=0.5, depth=3):
    from digat_amd import synthetic, util
    from digat_amd.model import Model, PrecomputedNewsEncoder
    spec = synthetic.SynthSpec(news_num=news_num, sag_neighbors=3, sag_hops=2, max_history_num=50, category_num=17,
                               embedding_dim=d, impressions=impressions, seed=seed, embedding_scale=scale)
    corpus = synthetic.make_corpus(spec)
    state = synthetic.make_state_dict(d, spec.category_num, depth, seed=seed + 1, bias_std=0.05)
    cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=spec.news_graph_size,
                                max_history_num=spec.max_history_num, category_num=spec.category_num, graph_depth=depth, dropout_rate=0.2)
    model = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(corpus.news_embedding)))
    model.graph_encoder.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to(DEV).eval()
    dc = util.DeviceCorpus.from_numpy(corpus, torch.device(DEV))
    return corpus, model, dc, cfg


def test_mismatched_split_image_is_refused():
    """An image split in one format cannot be read by the other format's kernel: the library remembers what it split."""
    from digat_amd import _lib
    L = _lib.lib()
    M, N, K = 2304, 160, 64
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) * 0.1
    y = torch.empty(M, N, device=DEV)
    ws = torch.empty(L.digat_split_weights_bytes(N, K), dtype=torch.uint8, device=DEV)
    for fmt in (_lib.GEMM_BF16X6, _lib.GEMM_F16X3):
        _lib.check(L.digat_linear_f32x3(x.data_ptr(), K, w.data_ptr(), None, y.data_ptr(), N, M, N, K, ws.data_ptr(), fmt, _lib.stream_ptr()), "x3")
        torch.cuda.synchronize()
        assert float((y - x @ w.t()).abs().max()) < 1e-4
    assert L.digat_linear_f32x3(x.data_ptr(), K, w.data_ptr(), None, y.data_ptr(), N, M, N, K, ws.data_ptr(), 7, _lib.stream_ptr()) != 0
    # the encoder's parameter block: images split as fp16x3, flags claiming bf16x6 -> DIGAT_ERR_ARG from the projection launch
    corpus, model, dc, _ = _small_world()
    enc = model.graph_encoder
    enc.projection_mode = "fp16x3"
    P = enc._params()
    assert P.flags & _lib.PARAMS_GEMM_F16X3
    from digat_amd import util
    util.prepare_news_side(enc, dc, 1024)
    P.flags &= ~_lib.PARAMS_GEMM_F16X3
    X = torch.randn(4096, enc.news_embedding_dim, device=DEV)
    out = torch.empty(3, 4096, enc.news_embedding_dim, device=DEV)
    rc = L.digat_user_project0(P, X.data_ptr(), out.data_ptr(), 4096, _lib.stream_ptr())
    assert rc == 1, rc                      # DIGAT_ERR_ARG
    P.flags |= _lib.PARAMS_GEMM_F16X3
    assert L.digat_user_project0(P, X.data_ptr(), out.data_ptr(), 4096, _lib.stream_ptr()) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K,fmt", [(1024, 400, 400, 1), (1024, 1200, 400, 1), (1000, 400, 800, 1), (37, 80, 40, 1), (1, 160, 72, 1),
                                       (1024, 400, 400, 0), (700, 1200, 408, 0), (2047, 240, 64, 1)])
def test_skinny_linear_on_split_images_is_fp32_grade(M, N, K, fmt):
    """The [B,d] linears (M < 2048) on the weights' split images (gemm_skinny_split_kernel): against fp64 next to the fp32-MFMA
    skinny kernel on the same data — mean and largest error at most 1.1x / 1.5x the fp32 chain's."""
    from digat_amd import _lib
    rng = np.random.default_rng(M + N + K + fmt)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    want = x.astype(np.float64) @ w.astype(np.float64).T + b
    xd, wd, bd = (torch.from_numpy(a).to(DEV) for a in (x, w, b))
    L = _lib.lib()
    ys = torch.full((M, N), float("nan"), device=DEV)
    y32 = torch.full((M, N), float("nan"), device=DEV)
    ws = torch.empty(L.digat_split_weights_bytes(N, K), dtype=torch.uint8, device=DEV)
    _lib.check(L.digat_linear_f32x3(xd.data_ptr(), K, wd.data_ptr(), bd.data_ptr(), ys.data_ptr(), N, M, N, K, ws.data_ptr(), fmt,
                                    _lib.stream_ptr()), "digat_linear_f32x3")
    _lib.check(L.digat_linear_f32(xd.data_ptr(), K, wd.data_ptr(), bd.data_ptr(), y32.data_ptr(), N, M, N, K, _lib.stream_ptr()), "digat_linear_f32")
    torch.cuda.synchronize()
    es = np.abs(ys.cpu().numpy().astype(np.float64) - want)
    e32 = np.abs(y32.cpu().numpy().astype(np.float64) - want)
    print(f"\n[skinny split {M}x{N}x{K} fmt {fmt}] mean {es.mean():.3e} (fp32 chain {e32.mean():.3e}), max {es.max():.3e} ({e32.max():.3e})")
    assert np.isfinite(es).all()
    slack = 1.1 if K >= 256 else 1.5           # short sums: a few terms, the two roundings of the operand split show
    assert es.mean() <= slack * e32.mean() + 1e-9 and es.max() <= 1.5 * e32.max() + 1e-7


def test_fp16x3_range_flag_and_fallback_at_layers_above_zero():
    """Node features of layers >= 1 are X + relu(alpha h): with a large layer-0 W they leave fp16x3's range (|x| >= 4094) although
    the weights (< 32) and the corpus's news representations (< 256) pass the host-side guard.  The GEMM raises the device
    flag; under "auto" util.compute_scores re-scores in bf16x6 (bit-identical to an explicit bf16x6 run) and stays there; an
    explicit "fp16x3" refuses to return such scores."""
    from digat_amd import _lib, util
    corpus, model, dc, _ = _small_world(scale=8.0, d=400, news_num=384, impressions=64)
    enc = model.graph_encoder
    with torch.no_grad():
        for g in ("news", "user"):
            getattr(enc, f"{g}_graph_attention_W")[0].weight.mul_(300.0)
        wmax = max(float(getattr(enc, f"{g}_graph_attention_W")[0].weight.abs().max()) for g in ("news", "user"))
    assert wmax < 32.0 and float(dc.news_embedding.abs().max()) < 256.0
    enc.projection_mode = "bf16x6"
    want, _ = util.compute_scores(model, dc, 1024, labels=corpus.row_label)
    assert np.isfinite(want).all()
    enc.projection_mode = "auto"
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got, _ = util.compute_scores(model, dc, 1024, labels=corpus.row_label)
    assert any("bf16x6" in str(w.message) for w in caught), "the fallback must say so"
    assert enc.range_fallback and enc.resolved_projection_mode() == "bf16x6"
    np.testing.assert_array_equal(got, want)
    enc.range_fallback = False
    enc.projection_mode = "fp16x3"
    with pytest.raises(_lib.DigatHipError):
        util.compute_scores(model, dc, 1024, labels=corpus.row_label)
    # ordinary features: the flag stays down and "auto" runs fp16x3
    corpus, model, dc, _ = _small_world(scale=0.5, d=400, news_num=384, impressions=64)
    enc = model.graph_encoder
    util.compute_scores(model, dc, 1024, labels=corpus.row_label)
    assert enc.resolved_projection_mode() == "fp16x3" and not enc.range_fallback and not enc.range_overflowed()


def test_score_rows_itself_watches_the_range_flag():
    """Direct callers of util.score_rows (not only compute_scores) get the check: an "auto" run whose deep-layer features leave
    fp16x3's range is redone in bf16x6 and equals an explicit bf16x6 run; an explicit "fp16x3" raises."""
    from digat_amd import _lib, util
    corpus, model, dc, _ = _small_world(scale=8.0, d=400, news_num=384, impressions=64)
    enc = model.graph_encoder
    with torch.no_grad():
        for g in ("news", "user"):
            getattr(enc, f"{g}_graph_attention_W")[0].weight.mul_(300.0)
    enc.projection_mode = "bf16x6"
    util.prepare_news_side(enc, dc, 1024)
    want = util.score_rows(model, dc, 0, dc.rows, 1024).cpu().numpy()
    enc.projection_mode = "auto"
    util.prepare_news_side(enc, dc, 1024)
    assert enc.resolved_projection_mode() == "fp16x3"
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = util.score_rows(model, dc, 0, dc.rows, 1024).cpu().numpy()
    assert any("bf16x6" in str(w.message) for w in caught)
    assert enc.range_fallback and enc.resolved_projection_mode() == "bf16x6"
    np.testing.assert_array_equal(got, want)
    enc.range_fallback = False
    enc.projection_mode = "fp16x3"
    util.prepare_news_side(enc, dc, 1024)
    with pytest.raises(_lib.DigatHipError):
        util.score_rows(model, dc, 0, dc.rows, 1024)


def test_evaluation_and_training_on_two_host_threads_are_bit_stable():
    """An fp16x3 dev evaluation on one thread / stream and bf16x6 training steps on another, at the same time: each must produce
    exactly what it produces alone (the operand format travels with the call; nothing process-wide is flipped)."""
    from digat_amd import synthetic, util
    from digat_amd.model import Model, PrecomputedNewsEncoder
    from digat_amd.trainer import SyntheticTrainSet, Trainer
    corpus, model, dc, cfg = _small_world(seed=9, d=400, news_num=512, impressions=160)
    model.graph_encoder.projection_mode = "fp16x3"

    def make_trainer():
        tcfg = types.SimpleNamespace(**vars(cfg), epoch=1, batch_size=64, lr=1e-3, weight_decay=0.0, gradient_clip_norm=1.0)
        tcfg.dropout_rate = 0.0
        torch.manual_seed(0)
        tm = Model(tcfg, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(corpus.news_embedding), trainable=True))
        tm.initialize()
        tm = tm.to(DEV)
        ts = SyntheticTrainSet(corpus, 4, seed=0)
        ts.negative_sampling()
        return Trainer(tm, tcfg, util.DeviceCorpus.from_numpy(corpus, torch.device(DEV)), ts), tm

    def train_steps(out, n=6):
        with torch.cuda.stream(torch.cuda.Stream(device=DEV)):
            tr, tm = make_trainer()
            tm.train()
            losses = [tr.train_step(np.arange(64) + 7 * i) for i in range(n)]
            torch.cuda.synchronize()
            out["losses"] = losses
            out["w"] = tm.graph_encoder.user_graph_attention_ffn1[0].weight.detach().cpu().numpy().copy()

    def evaluate(out, n=3):
        with torch.cuda.stream(torch.cuda.Stream(device=DEV)):
            runs = [util.compute_scores(model, dc, 1024, labels=corpus.row_label)[0] for _ in range(n)]
            out["scores"] = runs

    alone_t, alone_e = {}, {}
    train_steps(alone_t)
    evaluate(alone_e, 1)
    both_t, both_e, errors = {}, {}, []

    def guarded(fn, out):
        try:
            fn(out)
        except Exception as exc:            # surfaced below: a thread's exception must fail the test
            errors.append(exc)
    threads = [threading.Thread(target=guarded, args=(train_steps, both_t)), threading.Thread(target=guarded, args=(evaluate, both_e))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert both_t["losses"] == alone_t["losses"]
    np.testing.assert_array_equal(both_t["w"], alone_t["w"])
    for run in both_e["scores"]:
        np.testing.assert_array_equal(run, alone_e["scores"][0])


def test_two_threads_with_different_launch_options_do_not_flip_each_other():
    """Side stream and live-row lists are per-call options (digat_params.flags through DIGAT.launch_options, thread-local): two
    host threads scoring the SAME encoder — one with the side stream on and live-row lists off, one the other way round — each
    see their own flags in every call and both reproduce the single-thread scores bit for bit (round 3 had process-wide setters
    here: one thread's digat_set_side_stream reached the other's calls)."""
    from digat_amd import _lib, util
    corpus, model, dc, cfg = _small_world(seed=19, d=400, news_num=512, impressions=120)
    enc = model.graph_encoder
    util.prepare_news_side(enc, dc, 1024)
    want = util.score_rows(model, dc, 0, dc.rows, 1024).cpu().numpy()
    seen, errors = {"a": set(), "b": set()}, []
    barrier = threading.Barrier(2)

    def run(tag, opts, out):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=DEV)), enc.launch_options(**opts):
                barrier.wait(timeout=60)
                runs = []
                for _ in range(4):
                    seen[tag].add(int(enc._params().flags) & (_lib.PARAMS_SIDE_STREAM_OFF | _lib.PARAMS_SIDE_STREAM_ON | _lib.PARAMS_NO_LIVE_ROWS))
                    runs.append(util.score_rows(model, dc, 0, dc.rows, 1024, streams=1).cpu().numpy())
                out[tag] = runs
        except Exception as exc:
            errors.append(exc)
    out = {}
    threads = [threading.Thread(target=run, args=("a", dict(side_stream="on", live_rows=False), out)),
               threading.Thread(target=run, args=("b", dict(side_stream="off", live_rows=True), out))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert seen["a"] == {_lib.PARAMS_SIDE_STREAM_ON | _lib.PARAMS_NO_LIVE_ROWS} and seen["b"] == {_lib.PARAMS_SIDE_STREAM_OFF}
    for tag in ("a", "b"):
        for r in out[tag]:
            np.testing.assert_array_equal(r, want)
    assert not int(enc._params().flags) & (_lib.PARAMS_SIDE_STREAM_OFF | _lib.PARAMS_SIDE_STREAM_ON | _lib.PARAMS_NO_LIVE_ROWS)


def test_dev_evaluation_follows_a_news_encoder_in_training():
    """Trainer + MSA news encoder on title text + dev labels (trainer.py:109-120 with util.py:24-33): after every epoch the dev
    scores must come from news representations re-encoded with the CURRENT news-encoder weights, not from the ones cached
    before training."""
    from digat_amd import synthetic, util
    from digat_amd.model import Model
    from digat_amd.trainer import SyntheticTrainSet, Trainer
    Lw, V, dm, heads, dk, att = 16, 300, 40, 4, 20, 24
    d = heads * dk
    spec = synthetic.SynthSpec(news_num=256, sag_neighbors=3, sag_hops=1, max_history_num=10, category_num=5, embedding_dim=d,
                               impressions=64, mean_candidates=10.0, max_candidates=24, seed=5)
    corpus = synthetic.make_corpus(spec)
    cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=spec.news_graph_size,
                                max_history_num=spec.max_history_num, category_num=spec.category_num, graph_depth=2,
                                dropout_rate=0.1, epoch=2, batch_size=16, lr=1e-3, weight_decay=0.0, gradient_clip_norm=1.0,
                                vocabulary_size=V, word_embedding_dim=dm, max_title_length=Lw, MSA_head_num=heads, MSA_head_dim=dk,
                                attention_dim=att)
    torch.manual_seed(0)
    model = Model(cfg)
    model.initialize()
    with torch.no_grad():
        model.news_encoder.word_embedding.weight.mul_(0.1)
    model = model.to(DEV)
    dc = util.DeviceCorpus.from_numpy(corpus, torch.device(DEV))
    text, mask = synthetic.make_titles(spec.news_num, Lw, V, seed=7)
    dc.title_text = torch.from_numpy(text).to(torch.int32).to(DEV)
    dc.title_mask = torch.from_numpy(mask).to(DEV)
    stale = dc.news_embedding.clone()                    # the synthetic table: NOT what the news encoder produces
    trainer = Trainer(model, cfg, dc, SyntheticTrainSet(corpus, 4, seed=0), dev_labels=corpus.row_label)
    trainer.train()
    assert len(trainer.auc) == 2
    model.eval()
    fresh = util.cache_news_representations(model.news_encoder, dc.title_text, dc.title_mask, 4096)
    # the best epoch's weights were restored at the end of train(): one more evaluation re-encodes for them
    util.compute_scores(model, dc, 256, labels=corpus.row_label)
    assert dc.news_key is not None
    assert float((dc.news_embedding - fresh).abs().max()) <= 1e-6 * max(1.0, float(fresh.abs().max()))
    assert float((dc.news_embedding - stale).abs().max()) > 1e-2
    # unchanged weights: no re-encoding
    before = dc.news_embedding.data_ptr()
    util.compute_scores(model, dc, 256, labels=corpus.row_label)
    assert dc.news_embedding.data_ptr() == before
