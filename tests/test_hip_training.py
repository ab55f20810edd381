"""GPU suite: training-mode forward/backward through the HIP primitives (digat_amd/training.py) against
the loss and gradients the REFERENCE's autograd produced for one step (tests/golden/train_step.npz,
dropout 0 so that train mode is deterministic; trainer.py:98-102, model.py:54-77).

Tolerance: gradients are long fp32 sums (over B*n rows, n*n pairs, d channels) evaluated in a different
order than ATen's -> 2e-4 relative + 2e-6 absolute on each gradient tensor.
"""
import types

import numpy as np
import pytest
import torch

from conftest import check_grad_digest, load_golden, regenerate_train, split_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(got, want, what, rtol=2e-4, atol=2e-6):
    got = got.detach().cpu().numpy()
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite"
    scale = max(float(np.abs(want).max()), 1e-12)
    err = np.abs(got - want)
    tol = atol + rtol * np.maximum(np.abs(want), 0.05 * scale)
    if (err > tol).any():
        idx = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: max|diff|={err.max():.3e} (scale {scale:.3e}) at {idx}: got {got[idx]:.7g} want {want[idx]:.7g}")


def build(fx, dropout):
    from digat_amd.graphEncoders import DIGAT
    B, K, N, H, C, d, L = (int(v) for v in fx["meta"])
    ins, w, outs = split_fixture(fx)
    cfg = types.SimpleNamespace(news_graph_size=N, max_history_num=H, category_num=C, graph_depth=L, dropout_rate=dropout)
    enc = DIGAT(cfg, d)
    enc.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()})
    enc = enc.to(DEV).train()
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in ins.items()}
    return enc, t, outs, (B, K, N, H, C, d, L)


def run_step(enc, t, dims):
    B, K, N, H, C, d, L = dims
    Xn = t["news_graph_embeddings"].clone().requires_grad_(True)
    ue = t["user_news_embedding"].clone().requires_grad_(True)

    def expand(x):                                                   # model.py:64-71
        return x.unsqueeze(1).expand(B, K, *x.shape[1:]).contiguous().view(B * K, *x.shape[1:])

    n, u = enc(Xn, t["news_graph"], t["news_graph_mask"], expand(ue), expand(t["user_graph"]),
               expand(t["user_category_mask"]), expand(t["user_category_indices"]))
    logits = (u.view(B, K, d) * n.view(B, K, d)).sum(dim=2)
    loss = (-torch.log_softmax(logits, dim=1).select(1, 0)).mean()    # trainer.py:100
    loss.backward()
    return logits, loss, Xn, ue


def test_training_step_matches_reference_autograd():
    fx = load_golden("train_step.npz")
    enc, t, outs, dims = build(fx, dropout=0.0)
    logits, loss, Xn, ue = run_step(enc, t, dims)
    torch.cuda.synchronize()
    close(logits, outs["logits"], "logits", rtol=1e-5, atol=1e-5)
    close(loss, outs["loss"], "loss", rtol=1e-5, atol=1e-6)
    close(Xn.grad, fx["g_in_news_graph_embeddings"], "d news_graph_embeddings")
    close(ue.grad, fx["g_in_user_news_embedding"], "d user_news_embedding")
    for name, p in enc.named_parameters():
        assert p.grad is not None, name
        close(p.grad, fx["g_" + name], "grad " + name)


def test_training_step_with_dropout_live_matches_reference_autograd_under_the_same_masks(monkeypatch):
    """Dropout LIVE (rate 0.2), held to the reference: tests/golden/train_step_dropout.npz is the reference's autograd with its
    nn.Dropout modules drawing the library's counter-hash masks, seed 1001 + site in the order of graphEncoders.py:177-187.  Here
    ``training._seed`` hands out the same seeds: every fused dropout (layer inputs, alpha, the gate, the pooled topics, the topic
    nodes) must land on the same elements with the same scale, forward and backward."""
    from digat_amd import training
    fx = load_golden("train_step_dropout.npz")
    tape = iter(range(int(fx["first_seed"]), int(fx["first_seed"]) + 1000))
    monkeypatch.setattr(training, "_seed", lambda: next(tape))
    for use_ext in (True, False):
        tape = iter(range(int(fx["first_seed"]), int(fx["first_seed"]) + 1000))
        enc, t, outs, dims = build(fx, dropout=float(fx["dropout_rate"]))
        if not use_ext:
            monkeypatch.setattr(training._lib, "ext", lambda: None)
        logits, loss, Xn, ue = run_step(enc, t, dims)
        torch.cuda.synchronize()
        assert next(tape) - int(fx["first_seed"]) == int(fx["sites"]), "a dropout site was added, dropped or reordered"
        close(logits, outs["logits"], "logits", rtol=1e-5, atol=1e-5)
        close(loss, outs["loss"], "loss", rtol=1e-5, atol=1e-6)
        close(Xn.grad, fx["g_in_news_graph_embeddings"], "d news_graph_embeddings")
        close(ue.grad, fx["g_in_user_news_embedding"], "d user_news_embedding")
        for name, p in enc.named_parameters():
            assert p.grad is not None, name
            close(p.grad, fx["g_" + name], "grad " + name)


def build_default(dropout=0.0):
    """train_step_default.npz: the production shapes (N=10, U=67, d=400, L=3; 40 rows -> 2 680 user-node rows, so the
    bf16x6 training GEMMs (`_x3_ok`), multi-slice `gemm_tn_kernel` launches and the pairwise backward at n=67 all run)."""
    from digat_amd.graphEncoders import DIGAT
    fx = load_golden("train_step_default.npz")
    dims, w, flat, users = regenerate_train(fx)
    B, K, N, H, C, d, L = dims
    cfg = types.SimpleNamespace(news_graph_size=N, max_history_num=H, category_num=C, graph_depth=L, dropout_rate=dropout)
    enc = DIGAT(cfg, d)
    enc.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()})
    enc = enc.to(DEV).train()
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in flat.items() if k.startswith("news_")}
    t.update({k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in users.items() if k.startswith("user_")})
    return fx, enc, t, dims


def test_training_step_at_production_shapes_matches_reference_autograd():
    from digat_amd import training
    fx, enc, t, dims = build_default()
    B, K, N, H, C, d, L = dims
    assert training._x3_ok(B * K * (H + C), d, d), "the fixture must reach the bf16x6 training GEMMs"
    logits, loss, Xn, ue = run_step(enc, t, dims)
    torch.cuda.synchronize()
    close(logits, fx["out_logits"], "logits", rtol=2e-5, atol=2e-5)
    close(loss, fx["out_loss"], "loss", rtol=1e-5, atol=1e-6)
    close(Xn.grad, fx["g_in_news_graph_embeddings"], "d news_graph_embeddings")
    close(ue.grad, fx["g_in_user_news_embedding"], "d user_news_embedding")
    for name, p in enc.named_parameters():
        assert p.grad is not None, name
        check_grad_digest(fx, name, p.grad.detach().cpu().numpy(), 2e-4, "grad ")


def test_split_images_made_once_per_step_change_no_bit(monkeypatch):
    """digat_split_jobs: every split image of a step in one launch, handed to the entries (training.step_images), against each call
    splitting its own weights: the same images, so loss and every gradient bit for bit — with dropout live, through both bindings;
    and the launch is really taken (images for every layer of the user graph at these shapes)."""
    from digat_amd import training
    made = []
    real = training.split_images
    monkeypatch.setattr(training, "split_images", lambda jobs, dev: made.append(len(jobs)) or real(jobs, dev))

    def grads(once, use_ext):
        fx, enc, t, dims = build_default(dropout=0.2)
        enc.split_weights_once_per_step = once
        tape = iter(range(500, 1500))
        monkeypatch.setattr(training, "_seed", lambda: next(tape))
        if not use_ext:
            monkeypatch.setattr(training._lib, "ext", lambda: None)
        logits, loss, Xn, ue = run_step(enc, t, dims)
        torch.cuda.synchronize()
        return [loss.detach().clone(), Xn.grad.clone(), ue.grad.clone()] + [p.grad.clone() for p in enc.parameters()]

    for use_ext in (True, False):
        made.clear()
        a, b = grads(True, use_ext), grads(False, use_ext)
        assert made == [2 * 3], made          # the user graph's three layers, forward and backward images (news / topics: below 2 048 rows)
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_bf16_training_precision_at_production_shapes():
    """BASELINE configs[4], training half: digat_set_train_precision(1) — one bf16 product for the >= 2048-row GEMMs, fp32
    master weights / accumulation / weight gradients.  Against the reference's fp32 autograd: loss within 1e-2 relative, every
    gradient's norm within 5 % and its direction within 1 - cos < 2e-2 of this library's fp32-grade gradient (measured worst 7.4e-3); and the fp32-grade default must be restored bit for bit."""
    from digat_amd import _lib
    fx, enc, t, dims = build_default()
    base_logits, base_loss, _, _ = run_step(enc, t, dims)
    base_grads = {n: p.grad.clone() for n, p in enc.named_parameters()}
    prev = _lib.lib().digat_set_train_precision(1)
    try:
        fx, enc2, t2, dims = build_default()
        logits, loss, Xn, ue = run_step(enc2, t2, dims)
        torch.cuda.synchronize()
    finally:
        _lib.lib().digat_set_train_precision(prev)
    assert not torch.equal(logits, base_logits), "the bf16 path did not run"
    assert abs(float(loss.detach()) - float(fx["out_loss"])) <= 1e-2 * abs(float(fx["out_loss"])) + 1e-4      # bf16: 2^-8 per product
    close(logits, fx["out_logits"], "bf16 logits", rtol=2e-2, atol=2e-2 * float(np.abs(fx["out_logits"]).max()))
    worst = 0.0
    for name, p in enc2.named_parameters():
        g = p.grad.detach().cpu().numpy().astype(np.float64).reshape(-1)
        ref_norm = float(fx["gn_" + name])
        norm = float(np.sqrt((g ** 2).sum()))
        assert abs(norm - ref_norm) <= 0.05 * ref_norm + 1e-9, (name, norm, ref_norm)
        b = base_grads[name].detach().cpu().numpy().astype(np.float64).reshape(-1)
        cos = float((g * b).sum() / (np.linalg.norm(g) * np.linalg.norm(b) + 1e-30))
        worst = max(worst, 1.0 - cos)
        assert cos > 0.98, (name, cos)          # measured worst: 0.9926 (user_graph_attention_ffn1.1.weight)
    print(f"\n[bf16 training] loss {float(loss.detach()):.6f} vs reference {float(fx['out_loss']):.6f}; worst 1 - cos(grad, fp32 grad) = {worst:.2e}")
    fx, enc3, t3, dims = build_default()
    logits3, _, _, _ = run_step(enc3, t3, dims)
    assert torch.equal(logits3, base_logits)


def test_training_gradients_are_reproducible():
    fx = load_golden("train_step.npz")
    grads = []
    for _ in range(2):
        enc, t, outs, dims = build(fx, dropout=0.0)
        run_step(enc, t, dims)
        grads.append({n: p.grad.clone() for n, p in enc.named_parameters()})
    for n in grads[0]:
        assert torch.equal(grads[0][n], grads[1][n]), n            # ordered reductions: bit-identical


def test_both_bindings_and_both_ways_of_summing_shared_gradients_agree():
    """The training entries through the thin torch extension (the default) and through the ctypes table: the same C calls, the same
    bits — with dropout live (same seeds from torch's generator).  And the shared context weights' gradients summed inside the
    library's weight-gradient launches (training.StepSink, the default) against autograd's own sum of four per-call gradients:
    the same numbers up to the order of three additions."""
    from digat_amd import _lib
    assert _lib.ext() is not None, "digat_torch_ext.so has not been built (python -m digat_amd.build)"
    fx = load_golden("train_step.npz")

    def grads(use_ext, sink, dropout):
        enc, t, outs, dims = build(fx, dropout=dropout)
        enc.sum_shared_gradients_in_library = sink
        torch.manual_seed(1234)
        _lib.USE_TORCH_EXT = use_ext
        try:
            assert (_lib.ext() is not None) == use_ext
            logits, loss, Xn, ue = run_step(enc, t, dims)
        finally:
            _lib.USE_TORCH_EXT = True
        out = {n: p.grad.clone() for n, p in enc.named_parameters()}
        out["Xn"], out["ue"], out["loss"] = Xn.grad.clone(), ue.grad.clone(), loss.detach().clone()
        return out
    for dropout in (0.0, 0.2):
        a, b = grads(True, True, dropout), grads(False, True, dropout)
        for n in a:
            assert torch.equal(a[n], b[n]), (dropout, n)
    a, b = grads(True, True, 0.0), grads(True, False, 0.0)
    for n in a:
        scale = float(b[n].abs().max())
        assert torch.allclose(a[n], b[n], rtol=1e-5, atol=1e-6 * max(scale, 1e-12)), (n, float((a[n] - b[n]).abs().max()), scale)


def test_an_unused_encoder_output_fails_loudly_instead_of_dropping_gradients():
    """StepSink hands the shared weights' gradients to autograd with the LAST context call's backward: a loss that never reaches
    one of the calls would leave the sum incomplete, and the backward pass raises."""
    fx = load_golden("train_step.npz")
    enc, t, outs, dims = build(fx, dropout=0.0)
    B, K, N, H, C, d, L = dims

    def expand(x):
        return x.unsqueeze(1).expand(B, K, *x.shape[1:]).contiguous().view(B * K, *x.shape[1:])
    n, u = enc(t["news_graph_embeddings"], t["news_graph"], t["news_graph_mask"], expand(t["user_news_embedding"]), expand(t["user_graph"]),
               expand(t["user_category_mask"]), expand(t["user_category_indices"]))
    with pytest.raises(RuntimeError, match="StepSink"):
        n.sum().backward()                      # c_u's last user-context call never reaches this loss


def test_dropout_kernel_statistics_and_train_mode_runs():
    from digat_amd import training
    x = torch.ones(1 << 20, device=DEV)
    y = training.Dropout.apply(x, 0.2)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.8) < 5e-3
    assert torch.allclose(y[y != 0], torch.full_like(y[y != 0], 1.25))
    y2 = training.Dropout.apply(x, 0.2)
    assert not torch.equal(y, y2)                                   # fresh seed per call
    fx = load_golden("train_step.npz")
    enc, t, outs, dims = build(fx, dropout=0.2)
    logits, loss, Xn, ue = run_step(enc, t, dims)
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in enc.parameters())
    # eval mode of the same module is untouched by the training path
    enc.eval()
    with torch.no_grad():
        B, K = dims[0], dims[1]

        def expand(x):
            return x.unsqueeze(1).expand(B, K, *x.shape[1:]).contiguous().view(B * K, *x.shape[1:])
        n, u = enc(t["news_graph_embeddings"], t["news_graph"], t["news_graph_mask"], expand(t["user_news_embedding"]),
                   expand(t["user_graph"]), expand(t["user_category_mask"]), expand(t["user_category_indices"]))
    d = dims[5]
    lg = (u.view(B, K, d) * n.view(B, K, d)).sum(dim=2)
    close(lg, outs["logits"], "eval logits after training", rtol=1e-5, atol=1e-5)


def test_trainer_reduces_loss_on_a_tiny_synthetic_task():
    """H3 end to end: Model.forward (9 tensors) -> loss -> backward through the HIP kernels -> clip -> Adam."""
    from digat_amd import synthetic, util
    from digat_amd.model import Model, PrecomputedNewsEncoder
    from digat_amd.trainer import SyntheticTrainSet, Trainer
    spec = synthetic.SynthSpec(news_num=256, sag_neighbors=3, sag_hops=1, max_history_num=10, category_num=5,
                               embedding_dim=64, impressions=64, mean_candidates=10.0, max_candidates=24, seed=5)
    corpus = synthetic.make_corpus(spec)
    cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=spec.news_graph_size,
                                max_history_num=spec.max_history_num, category_num=spec.category_num, graph_depth=2,
                                dropout_rate=0.1, epoch=6, batch_size=16, lr=1e-3, weight_decay=0.0, gradient_clip_norm=1.0)
    torch.manual_seed(0)
    model = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(corpus.news_embedding), trainable=True))
    model.initialize()
    model = model.to(DEV)
    dc = util.DeviceCorpus.from_numpy(corpus, torch.device(DEV))
    trainer = Trainer(model, cfg, dc, SyntheticTrainSet(corpus, 4, seed=0), dev_labels=corpus.row_label)
    losses = trainer.train()
    assert len(losses) == 6 and all(np.isfinite(losses)), losses
    assert losses[-1] < 0.85 * losses[0], losses         # fits the clicked candidates of 64 impressions
    # the epoch loop of trainer.py:107-188: dev metrics after every epoch through the HIP inference path (the per-news caches
    # follow the weights: util.weights_key), the best epoch's weights are the result
    assert len(trainer.auc) == 6 and all(0.0 <= v <= 1.0 for v in trainer.auc)
    assert 1 <= trainer.best_dev_epoch <= 6 and trainer.best_state is not None
    assert trainer.auc[-1] > trainer.auc[0], trainer.auc             # training on these impressions must show on their dev AUC


@pytest.mark.parametrize("n,d", [(128, 64), (113, 48), (67, 400), (1, 32)])
def test_eq8_and_gat_layers_backward_at_the_largest_graphs(n, d):
    """The two layer pairs (digat_xattn_fwd_train / _bwd, digat_gat_fwd_train / _bwd) alone against the oracle's autograd at the
    largest graph the ABI admits (DIGAT_MAX_NODES = 128: the pairwise backward then runs 32-channel chunks, 113 is the first
    size that needs them), at the user graph's size and at a single node."""
    from digat_amd import training
    from oracle import digat_oracle as O
    g = torch.Generator().manual_seed(n * 1000 + d)
    B = 3
    X = torch.randn(B, n, d, generator=g)
    A = (torch.rand(B, n, n, generator=g) < min(1.0, 6.0 / n))
    A |= torch.eye(n, dtype=torch.bool).unsqueeze(0)
    A[1, 0] = False                                                  # a centre without any entry: uniform attention (E5)
    ctx = torch.randn(B, d, generator=g)
    dOut = torch.randn(B, n, d, generator=g)
    w = {k: (torch.randn(*shape, generator=g) * scale) for k, shape, scale in
         [("W", (d, d), d ** -0.5), ("bW", (d,), 0.1), ("F1", (d, d), d ** -0.5), ("F2", (d, d), d ** -0.5), ("F3", (d, d), d ** -0.5),
          ("b3", (d,), 0.1), ("a", (1, d), d ** -0.5), ("a1", (1, d), d ** -0.5), ("a2", (1, d), d ** -0.5)]}
    p = {"user_graph_attention_W.0.weight": w["W"], "user_graph_attention_W.0.bias": w["bW"],
         "user_graph_attention_ffn1.0.weight": w["F1"], "user_graph_attention_ffn2.0.weight": w["F2"],
         "user_graph_attention_ffn3.0.weight": w["F3"], "user_graph_attention_ffn3.0.bias": w["b3"],
         "user_graph_attention_a.0.weight": w["a"], "user_graph_attention_a1.0.weight": w["a1"],
         "user_graph_attention_a2.0.weight": w["a2"]}
    p = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    Xo, co = X.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    (O.cross_graph_attention(p, "user", 0, Xo, A, co) * dOut).sum().backward()
    want_x = {"X": Xo.grad, "ctx": co.grad, **{k: p["user_graph_attention_" + n_].grad for k, n_ in
              [("W", "W.0.weight"), ("bW", "W.0.bias"), ("F1", "ffn1.0.weight"), ("F2", "ffn2.0.weight"), ("F3", "ffn3.0.weight"),
               ("b3", "ffn3.0.bias"), ("a", "a.0.weight")]}}
    for v in p.values():
        v.grad = None
    Xg = X.clone().requires_grad_(True)
    (O.gat_layer(p, "user", 0, Xg, A) * dOut).sum().backward()
    want_g = {"X": Xg.grad, "W": p["user_graph_attention_W.0.weight"].grad, "bW": p["user_graph_attention_W.0.bias"].grad,
              "a1": p["user_graph_attention_a1.0.weight"].grad, "a2": p["user_graph_attention_a2.0.weight"].grad}

    dv = {k: v.to(DEV).requires_grad_(True) for k, v in w.items()}
    Ab = A.to(torch.uint8).to(DEV).contiguous()
    Xd, cd = X.to(DEV).requires_grad_(True), ctx.to(DEV).requires_grad_(True)
    out = training.XattnFused.apply(Xd, Ab, cd, dv["W"], dv["bW"], dv["F1"], dv["F2"], dv["F3"], dv["b3"], dv["a"], 0.0)
    (out * dOut.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    got = {"X": Xd.grad, "ctx": cd.grad, **{k: dv[k].grad for k in ("W", "bW", "F1", "F2", "F3", "b3", "a")}}
    for k in want_x:
        close(got[k], want_x[k].numpy(), f"Eq. 8 n={n} grad {k}")
    for v in dv.values():
        v.grad = None
    Xd2 = X.to(DEV).requires_grad_(True)
    out = training.GatFused.apply(Xd2, Ab, dv["W"], dv["bW"], dv["a1"], dv["a2"], 0.0)
    (out * dOut.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    got = {"X": Xd2.grad, "W": dv["W"].grad, "bW": dv["bW"].grad, "a1": dv["a1"].grad, "a2": dv["a2"].grad}
    for k in want_g:
        close(got[k], want_g[k].numpy(), f"GAT n={n} grad {k}")


@pytest.mark.parametrize("n,d,B,p_in,per_node,mode", [(67, 64, 3, 0.0, 6, 0), (10, 64, 3, 0.0, 6, 0), (128, 32, 3, 0.0, 6, 0), (67, 80, 32, 0.25, 6, 0),
                                                      (10, 64, 3, 0.25, 6, 0), (67, 64, 3, 0.0, 40, 0), (128, 400, 2, 0.0, 9, 0),
                                                      (67, 64, 3, 0.0, 6, 1), (67, 64, 3, 0.0, 40, 1), (67, 64, 3, 0.0, 6, 2), (10, 64, 3, 0.0, 6, 1)])
def test_eq8_layer_with_attention_dropout_live(n, d, B, p_in, per_node, mode, monkeypatch):
    """digat_xattn_fwd_train / _bwd with the attention dropout LIVE (p = 0.3) against the oracle's autograd under the same keep bits:
    the dropout is applied inside the score kernels (the tile kernel at 67 and 128 nodes, the small-graph kernel — which also
    aggregates in place — at 10), forward output and every gradient.  Graphs of more than 16 nodes go to the wave-per-centre kernel
    when the batch's adjacency is sparse (~6 or ~9 entries per node here) and to the tile kernel + aggregation when it is not (~40):
    the choice is made on the device (mode 0) or by the caller (1: entry-wise, also on the dense graph; 2: all-pairs, also on the
    sparse one — the same function either way).  With p_in > 0 the library also applies the layer's INPUT
    dropout and returns the gradient of the undropped input: through the epilogue of the bf16x6 input-gradient product at
    32 x 67 = 2 144 rows, through a dropout launch below 2 048 rows."""
    from digat_amd import training
    from oracle import digat_oracle as O
    g = torch.Generator().manual_seed(n * 77 + d)
    p_alpha, seed, seed_in = 0.3, 4242, 977
    if p_in > 0 and B * n >= 2048:
        assert training._x3_ok(B * n, d, 3 * d)
    X = torch.randn(B, n, d, generator=g)
    A = (torch.rand(B, n, n, generator=g) < min(1.0, float(per_node) / n))
    A |= torch.eye(n, dtype=torch.bool).unsqueeze(0)
    A[1, 0] = False                                                  # a centre without any entry: uniform attention (E5)
    A[1, 1] = False
    A[1, 1, 1] = True                                                # ... and one whose only entry is its self loop
    ctx = torch.randn(B, d, generator=g)
    dOut = torch.randn(B, n, d, generator=g)
    w = {k: (torch.randn(*shape, generator=g) * scale) for k, shape, scale in
         [("W", (d, d), d ** -0.5), ("bW", (d,), 0.1), ("F1", (d, d), d ** -0.5), ("F2", (d, d), d ** -0.5), ("F3", (d, d), d ** -0.5),
          ("b3", (d,), 0.1), ("a", (1, d), d ** -0.5)]}
    names = {"W": "W.0.weight", "bW": "W.0.bias", "F1": "ffn1.0.weight", "F2": "ffn2.0.weight", "F3": "ffn3.0.weight", "b3": "ffn3.0.bias",
             "a": "a.0.weight"}
    p = {"user_graph_attention_" + names[k]: v.clone().requires_grad_(True) for k, v in w.items()}
    Xo, co = X.clone().requires_grad_(True), ctx.clone().requires_grad_(True)

    def drop(x, frac):          # frac 0.5: the layer's input site (identity when the caller drops it: p_in = 0); 1.0: alpha — the kernels' bits
        if frac == 0.5:
            return O.hash_dropout(x.contiguous(), p_in, seed_in) if p_in > 0 else x
        return O.hash_dropout(x.contiguous(), p_alpha, seed)

    want_out = O.cross_graph_attention(p, "user", 0, Xo, A, co, drop=drop)
    (want_out * dOut).sum().backward()
    want = {"X": Xo.grad, "ctx": co.grad, **{k: p["user_graph_attention_" + names[k]].grad for k in w}}

    seeds = iter([seed_in, seed] if p_in > 0 else [seed])
    monkeypatch.setattr(training, "_seed", lambda: next(seeds))
    dv = {k: v.to(DEV).requires_grad_(True) for k, v in w.items()}
    Ab = A.to(torch.uint8).to(DEV).contiguous()
    Xd, cd = X.to(DEV).requires_grad_(True), ctx.to(DEV).requires_grad_(True)
    out = training.XattnFused.apply(Xd, Ab, cd, dv["W"], dv["bW"], dv["F1"], dv["F2"], dv["F3"], dv["b3"], dv["a"], p_alpha, None, p_in, mode)
    (out * dOut.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    close(out, want_out.detach().numpy(), f"Eq. 8 n={n} out under dropout", rtol=2e-5, atol=2e-5)
    got = {"X": Xd.grad, "ctx": cd.grad, **{k: dv[k].grad for k in w}}
    for k in want:
        close(got[k], want[k].numpy(), f"Eq. 8 n={n} dropout grad {k}")


def test_table_lookup_of_two_id_lists_has_one_dense_gradient_equal_to_two_embedding_backwards():
    """training.TableLookup2 (the table-backed news encoder's two lookups of a training batch, one launch for the table gradient)
    against torch's own embedding backward: many repeated ids (a history repeats news, candidates share neighbours), ids only one
    list holds, rows nobody looks up (zero gradient), a row stride that is not 64 floats; twice: bit-reproducible."""
    from digat_amd import training
    from digat_amd.model import PrecomputedNewsEncoder
    g = torch.Generator().manual_seed(5)
    V, dm = 5000, 400
    table = torch.randn(V, dm, generator=g)
    ids_a = torch.randint(0, 300, (320, 10), generator=g)          # heavy repetition
    ids_b = torch.randint(100, V, (64, 50), generator=g)
    ids_b[:, :5] = 7                                               # one id 320 times in the second list, absent from the first
    wa, wb = torch.randn(320, 10, dm, generator=g), torch.randn(64, 50, dm, generator=g)
    t0 = table.clone().requires_grad_(True)
    ((torch.nn.functional.embedding(ids_a, t0) * wa).sum() + (torch.nn.functional.embedding(ids_b, t0) * wb).sum()).backward()
    grads = []
    for _ in range(2):
        t1 = table.to(DEV).requires_grad_(True)
        a, b = training.TableLookup2.apply(t1, ids_a.to(DEV), ids_b.to(DEV))
        assert torch.equal(a.cpu(), table[ids_a]) and torch.equal(b.cpu(), table[ids_b])
        ((a * wa.to(DEV)).sum() + (b * wb.to(DEV)).sum()).backward()
        torch.cuda.synchronize()
        grads.append(t1.grad.clone())
    assert torch.equal(grads[0], grads[1])
    close(grads[0], t0.grad.numpy(), "table gradient", rtol=1e-5, atol=1e-6)      # sums of up to 320 rows in another order
    untouched = torch.ones(V, dtype=torch.bool)
    untouched[ids_a.flatten()] = False
    untouched[ids_b.flatten()] = False
    assert untouched.any() and float(grads[0].cpu()[untouched].abs().max()) == 0.0
    # through the encoder stand-in, as Model.forward calls it ([.., 1] "titles" of one token), and only one output used
    enc = PrecomputedNewsEncoder(table, trainable=True).to(DEV)
    ca, hb = enc.encode_pair(ids_a.to(DEV).unsqueeze(-1), ids_b.to(DEV).unsqueeze(-1))
    assert ca.shape == (320, 10, dm) and hb.shape == (64, 50, dm)
    (hb * wb.to(DEV)).sum().backward()
    t2 = table.clone().requires_grad_(True)
    (torch.nn.functional.embedding(ids_b, t2) * wb).sum().backward()
    close(enc.embed_table.grad, t2.grad.numpy(), "table gradient, second list only", rtol=1e-5, atol=1e-6)


def test_step_head_and_tail_match_torch_autograd():
    """training.RowLogits (model.py:75) and trainer.training_loss on the GPU (training.ClickLoss, trainer.py:100) against the torch
    expressions they replace: values and gradients, with an incoming gradient other than 1."""
    from digat_amd import training, trainer
    g = torch.Generator().manual_seed(3)
    B, K, d = 64, 5, 400
    n, u = torch.randn(B * K, d, generator=g), torch.randn(B * K, d, generator=g) * 0.3
    n0, u0 = n.clone().requires_grad_(True), u.clone().requires_grad_(True)
    lg0 = (u0.view(B, K, d) * n0.view(B, K, d)).sum(dim=2)
    loss0 = (-torch.log_softmax(lg0, dim=1).select(1, 0)).mean()
    (loss0 * 1.7).backward()
    n1, u1 = n.to(DEV).requires_grad_(True), u.to(DEV).requires_grad_(True)
    lg1 = training.RowLogits.apply(n1, u1).view(B, K)
    loss1 = trainer.training_loss(lg1)
    assert type(loss1.grad_fn).__name__.startswith("ClickLoss")
    (loss1 * 1.7).backward()
    torch.cuda.synchronize()
    close(lg1, lg0.detach().numpy(), "logits", rtol=1e-5, atol=1e-5)
    close(loss1, loss0.detach().numpy(), "loss", rtol=1e-6, atol=1e-6)
    close(n1.grad, n0.grad.numpy(), "d news_ctx", rtol=2e-5, atol=1e-8)
    close(u1.grad, u0.grad.numpy(), "d user_ctx", rtol=2e-5, atol=1e-8)


@pytest.mark.parametrize("max_norm,wd", [(1.0, 0.0), (0.0, 0.0), (0.05, 0.01)])
def test_clip_adam_matches_clip_grad_norm_and_torch_adam(max_norm, wd):
    """optim.ClipAdam (clipping by the global norm + Adam in three launches) against clip_grad_norm_ + torch.optim.Adam on the CPU over
    several steps: tensors of odd sizes (the vector path's tails, a tensor smaller than a chunk, one of several chunks, an unaligned
    view), two weight-decay groups, gradients large enough to be clipped."""
    from digat_amd.optim import ClipAdam
    g = torch.Generator().manual_seed(11)
    shapes = [(400, 400), (400,), (3,), (70001,), (1, 400), (33, 17)]
    base = [torch.randn(*s_, generator=g) for s_ in shapes]
    ref = [b.clone().requires_grad_(True) for b in base]
    big = torch.zeros(70001 + 3, device=DEV)
    dev = [b.to(DEV).requires_grad_(True) for b in base]
    opt_ref = torch.optim.Adam([{"params": ref[:3], "weight_decay": wd}, {"params": ref[3:], "weight_decay": 0.0}], lr=1e-2)
    opt_dev = ClipAdam([{"params": dev[:3], "weight_decay": wd}, {"params": dev[3:], "weight_decay": 0.0}], lr=1e-2)
    for step in range(4):
        grads = [torch.randn(*s_, generator=g) * (3.0 if step % 2 == 0 else 0.01) for s_ in shapes]
        for p, q, gr in zip(ref, dev, grads):
            p.grad = gr.clone()
            q.grad = gr.to(DEV)
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(ref, max_norm)
        opt_ref.step()
        opt_dev.step(max_norm=max_norm)
        torch.cuda.synchronize()
        for k, (p, q) in enumerate(zip(ref, dev)):
            # (where coef g and wd p nearly cancel, g' is of the size of eps and the update m / (sqrt(v) + eps) feels the last bit of the
            # norm — summed per chunk here, per tensor in torch: 2e-6 of an O(1) parameter in the third case)
            close(q, p.detach().numpy(), f"step {step} tensor {k}", rtol=2e-5, atol=1e-6)
        assert torch.equal(dev[0].grad.cpu(), grads[0])            # the gradients are read, not rescaled
    del big


def test_round6_entry_points_reject_bad_arguments_and_handle_empty_work():
    """digat_split_jobs / digat_embedding_bwd_unsorted / digat_clip_adam_step / digat_click_loss: error codes for bad arguments (no launch),
    empty work is a no-op, ids outside the table receive nothing."""
    import ctypes as C
    from digat_amd import _lib
    L = _lib.lib()
    ERR_ARG = 1
    d = 80
    w = torch.randn(d, d, device=DEV)
    img = torch.empty(L.digat_split_job_bytes(d, d, 0, 1), dtype=torch.uint8, device=DEV)
    assert L.digat_split_job_bytes(d, d, 2, 1) == 0 and L.digat_split_job_bytes(d, d, 0, 2) == 0
    job = (_lib.SplitJob * 1)(_lib.SplitJob(w.data_ptr(), None, None, d, d, 0, 0, img.data_ptr()))
    assert L.digat_split_jobs(job, 0, None) == 0
    assert L.digat_split_jobs(None, 1, None) == ERR_ARG
    assert L.digat_split_jobs(job, 25, None) == ERR_ARG
    bad = (_lib.SplitJob * 1)(_lib.SplitJob(w.data_ptr(), w.data_ptr(), None, d, d, 0, 0, img.data_ptr()))      # two of three matrices
    assert L.digat_split_jobs(bad, 1, None) == ERR_ARG
    bad = (_lib.SplitJob * 1)(_lib.SplitJob(w.data_ptr(), None, None, d, d, 3, 0, img.data_ptr()))
    assert L.digat_split_jobs(bad, 1, None) == ERR_ARG
    # the image of one job equals the per-call split of the same weight (digat_split_weights, bf16x6)
    ref = torch.empty_like(img)
    assert L.digat_split_jobs(job, 1, _lib.stream_ptr()) == 0
    assert L.digat_split_weights(w.data_ptr(), d, d, ref.data_ptr(), 0, _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(img, ref)
    # table gradient: nothing to do, and ids outside [0, V)
    V, dm = 50, 16
    table_grad = torch.zeros(V, dm, device=DEV)
    ws = torch.empty(L.digat_embedding_bwd_unsorted_workspace_bytes(8, dm, V), dtype=torch.uint8, device=DEV)
    assert L.digat_embedding_bwd_unsorted(None, None, dm, 0, None, None, dm, 0, dm, V, table_grad.data_ptr(), ws.data_ptr(), ws.numel(), None) == 0
    ids = torch.tensor([3, -1, 70, 3, 49, 0, 3, 49], dtype=torch.int64, device=DEV)
    g = torch.arange(8 * dm, dtype=torch.float32, device=DEV).view(8, dm)
    assert L.digat_embedding_bwd_unsorted(ids.data_ptr(), g.data_ptr(), dm, 8, None, None, dm, 0, dm, V, table_grad.data_ptr(), ws.data_ptr(),
                                          ws.numel(), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    want = torch.zeros(V, dm)
    for k, i in enumerate(ids.tolist()):
        if 0 <= i < V:
            want[i] += g[k].cpu()
    assert torch.equal(table_grad.cpu(), want)
    assert L.digat_embedding_bwd_unsorted(ids.data_ptr(), g.data_ptr(), dm, 8, None, None, dm, 0, 6, V, table_grad.data_ptr(), ws.data_ptr(),
                                          ws.numel(), None) == 2                                        # dm % 4: SHAPE
    assert L.digat_embedding_bwd_unsorted(ids.data_ptr(), g.data_ptr(), dm, 8, None, None, dm, 0, dm, V, table_grad.data_ptr(), ws.data_ptr(), 16,
                                          None) == 3                                                    # WORKSPACE
    assert L.digat_clip_adam_step(None, None, None, 1, None, 1.0, 1e-3, 0.9, 0.999, 1e-8, 1, None) == ERR_ARG
    assert L.digat_clip_adam_step(ws.data_ptr(), ws.data_ptr(), ws.data_ptr(), 0, ws.data_ptr(), 1.0, 1e-3, 0.9, 0.999, 1e-8, 1, None) == 0
    assert L.digat_clip_adam_step(ws.data_ptr(), ws.data_ptr(), ws.data_ptr(), 1, ws.data_ptr(), 1.0, 1e-3, 0.9, 0.999, 1e-8, 0, None) == ERR_ARG   # step >= 1
    assert L.digat_click_loss(None, 4, 5, None, None, None) == ERR_ARG
