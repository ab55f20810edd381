"""GPU suite for the MSA news encoder: digat_msa_fwd (through newsEncoders.MSA in eval mode) against the
vectors minted from the reference's layers.py modules and against the oracle."""
import types

import numpy as np
import pytest
import torch

from conftest import check_grad_digest, load_golden

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _encoder(V, dm, h, dk, att, Lw, state):
    from digat_amd import newsEncoders
    cfg = types.SimpleNamespace(vocabulary_size=V, word_embedding_dim=dm, max_title_length=Lw, dropout_rate=0.2,
                                MSA_head_num=h, MSA_head_dim=dk, attention_dim=att)
    enc = newsEncoders.MSA(cfg)
    missing = enc.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return enc.to(_dev()).eval()


@pytest.mark.parametrize("name", ["msa_tiny.npz", "msa_default.npz"])
def test_msa_hip_matches_reference_vectors(name):
    from digat_amd import synthetic
    fx = load_golden(name)
    T_, Lw, V, dm, h, dk, att = (int(v) for v in fx["meta"])
    s_w, s_t = (int(v) for v in fx["seeds"])
    state = synthetic.make_msa_state(V, dm, h, dk, att, seed=s_w)
    text, mask = synthetic.make_titles(T_, Lw, V, seed=s_t)
    enc = _encoder(V, dm, h, dk, att, Lw, state)
    with torch.no_grad():
        got = enc(torch.from_numpy(text).to(_dev()).unsqueeze(1), torch.from_numpy(mask).to(_dev()).unsqueeze(1)).squeeze(1)
    np.testing.assert_allclose(got.cpu().numpy(), fx["out_news_representation"], rtol=1e-5, atol=2e-6)


def test_msa_hip_batch_on_the_matrix_core_path_matches_oracle():
    """Production shapes and enough titles (T*Lw >= 2048) for the bf16x6 GEMM with the embedding lookup folded in."""
    from digat_amd import synthetic
    from oracle import news_oracle
    T_, Lw, V, dm, h, dk, att = 300, 32, 2000, 300, 16, 25, 256
    state = synthetic.make_msa_state(V, dm, h, dk, att, seed=61)
    text, mask = synthetic.make_titles(T_, Lw, V, seed=62)
    enc = _encoder(V, dm, h, dk, att, Lw, state)
    with torch.no_grad():
        got = enc(torch.from_numpy(text).to(_dev()), torch.from_numpy(mask).to(_dev()))
        want = news_oracle.msa_forward({k: torch.from_numpy(v) for k, v in state.items()}, torch.from_numpy(text),
                                       torch.from_numpy(mask), h)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-5, atol=2e-6)
    # eval-mode forward with grad enabled runs the training pair without dropout: same numbers within fp32 reassociation
    stock = enc(torch.from_numpy(text[:8]).to(_dev()).unsqueeze(0), torch.from_numpy(mask[:8]).to(_dev()).unsqueeze(0))
    np.testing.assert_allclose(stock.detach().cpu().numpy()[0], want.numpy()[:8], rtol=1e-4, atol=1e-5)


def test_news_cache_in_batches_equals_one_call():
    """util.cache_news_representations (util.py:24-33) over ragged batches vs one call: same bits."""
    from digat_amd import synthetic, util
    T_, Lw, V, dm, h, dk, att = 1000, 32, 3000, 300, 16, 25, 256
    state = synthetic.make_msa_state(V, dm, h, dk, att, seed=71)
    text, mask = synthetic.make_titles(T_, Lw, V, seed=72)
    enc = _encoder(V, dm, h, dk, att, Lw, state)
    tt, tm = torch.from_numpy(text).to(_dev()), torch.from_numpy(mask).to(_dev())
    a = util.cache_news_representations(enc, tt, tm, 384)
    with torch.no_grad():
        b = enc(tt, tm)
    assert torch.equal(a, b)


# ---- training: digat_msa_fwd_train / digat_msa_bwd / digat_embedding_bwd ------------------------------------------------
def _train_case(name, dropout=0.0):
    from digat_amd import synthetic
    fx = load_golden(name)
    T_, Lw, V, dm, h, dk, att = (int(v) for v in fx["meta"])
    s_w, s_t, s_r = (int(v) for v in fx["seeds"])
    state = synthetic.make_msa_state(V, dm, h, dk, att, seed=s_w)
    text, mask = synthetic.make_titles(T_, Lw, V, seed=s_t)
    text[2], mask[2] = 0, False
    R = np.random.default_rng(s_r).standard_normal((T_, h * dk)).astype(np.float32)
    tot = sum(float(np.asarray(v, dtype=np.float64).sum()) for v in [text, mask, R] + list(state.values()))
    assert abs(tot - float(fx["input_checksum"])) <= 1e-6 * max(1.0, abs(tot)), "synthetic generator drifted from the fixture's"
    enc = _encoder(V, dm, h, dk, att, Lw, state)
    enc.dropout.p = dropout
    return fx, enc.train(), torch.from_numpy(text).to(_dev()), torch.from_numpy(mask).to(_dev()), torch.from_numpy(R).to(_dev())


def _close(got, want, what, rtol=2e-4, atol=2e-6):
    got, want = got.detach().cpu().numpy(), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite"
    scale = max(float(np.abs(want).max()), 1e-12)
    err = np.abs(got - want)
    tol = atol + rtol * np.maximum(np.abs(want), 0.05 * scale)
    assert not (err > tol).any(), f"{what}: max|diff| {err.max():.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("name", ["msa_train_tiny.npz", "msa_train_default.npz"])
def test_msa_training_step_matches_reference_autograd(name):
    """Output, loss and every gradient (word embedding included) of one training step, dropout 0, against the reference's
    autograd.  Tolerance as tests/test_hip_training.py: long fp32 sums in another order -> 2e-4 relative + 2e-6 absolute."""
    fx, enc, text, mask, R = _train_case(name)
    out = enc(text.unsqueeze(0), mask.unsqueeze(0)).squeeze(0)
    loss = (out * R).sum()
    loss.backward()
    torch.cuda.synchronize()
    _close(out, fx["out_news_representation"], "news representation", rtol=1e-5, atol=2e-6)
    _close(loss, fx["out_loss"], "loss", rtol=2e-5, atol=1e-5)
    for k, p in enc.named_parameters():
        assert p.grad is not None, k
        if "g_" + k in fx:
            _close(p.grad, fx["g_" + k], "grad " + k)
        else:
            check_grad_digest(fx, k, p.grad.detach().cpu().numpy(), 2e-4, "grad ")


def test_msa_training_with_dropout_live_matches_the_oracle_under_the_same_mask():
    """The embedding dropout LIVE (newsEncoders.py:77): the library draws its keep bits from the counter hash of (seed, element of the
    [T Lw, dm] embedded tokens) and applies the backward in the epilogue of the 2 304-row input-gradient product (300 columns in 320-column
    strips).  The oracle's autograd under the same mask: output and every gradient, the word embedding's included."""
    from oracle import digat_oracle as O
    from oracle import news_oracle as N
    fx, enc, text, mask, R = _train_case("msa_train_default.npz", dropout=0.2)
    T_, Lw, V, dm, h, dk, att = (int(v) for v in fx["meta"])
    assert T_ * Lw >= 2048 and dm % 80 != 0
    torch.manual_seed(77)
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())           # what MsaFused.forward will draw
    torch.manual_seed(77)
    out = enc(text.unsqueeze(0), mask.unsqueeze(0)).squeeze(0)
    (out * R).sum().backward()
    torch.cuda.synchronize()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    want = N.msa_forward(p, text.cpu().long(), mask.cpu(), h, drop=lambda w: O.hash_dropout(w.contiguous(), 0.2, seed))
    (want * R.cpu()).sum().backward()
    _close(out, want.detach().numpy(), "news representation under dropout", rtol=1e-5, atol=2e-6)
    for k, q in enc.named_parameters():
        _close(q.grad, p[k].grad.numpy(), "grad " + k + " under dropout")


def test_msa_training_with_dropout_is_reproducible_and_finite():
    def once():
        torch.manual_seed(11)
        fx, enc, text, mask, R = _train_case("msa_train_default.npz", dropout=0.2)
        out = enc(text.unsqueeze(0), mask.unsqueeze(0)).squeeze(0)
        (out * R).sum().backward()
        torch.cuda.synchronize()
        return out.detach().clone(), [p.grad.clone() for p in enc.parameters()]
    o1, g1 = once()
    o2, g2 = once()
    assert torch.isfinite(o1).all() and all(torch.isfinite(g).all() for g in g1)
    assert torch.equal(o1, o2) and all(torch.equal(a, b) for a, b in zip(g1, g2))
    fx = load_golden("msa_train_default.npz")
    assert float((o1.cpu() - torch.from_numpy(fx["out_news_representation"])).abs().max()) > 1e-3      # the dropout is live


@pytest.mark.parametrize("M,V,dm", [(5000, 50, 300), (700, 3000, 64), (64, 10, 8), (3, 4, 4), (20000, 3, 32)])
def test_embedding_backward_sums_rows_per_token(M, V, dm):
    """digat_embedding_bwd against index_add_ in float64: few tokens with very long runs (several reduction levels), many
    tokens with short runs, a single chunk."""
    from digat_amd import _lib
    g = torch.Generator().manual_seed(M + V)
    tokens = torch.randint(0, V, (M,), generator=g)
    if M > 100:
        tokens[: M // 2] = 0                                       # the padding token: one very long run
    rows = torch.randn(M, dm, generator=g)
    want = torch.zeros(V, dm, dtype=torch.float64).index_add_(0, tokens, rows.double())
    dev = _dev()
    stok, order = torch.sort(tokens.to(dev), stable=True)
    stok, order = stok.to(torch.int32), order.to(torch.int32)
    rows_d = rows.to(dev)
    table = torch.zeros(V, dm, device=dev)
    L = _lib.lib()
    nb = L.digat_embedding_bwd_workspace_bytes(M, dm)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    _lib.check(L.digat_embedding_bwd(rows_d.data_ptr(), dm, order.data_ptr(), stok.data_ptr(), M, dm, table.data_ptr(), ws.data_ptr(), nb,
                                     _lib.stream_ptr()), "digat_embedding_bwd")
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    assert float((table.cpu().double() - want).abs().max()) <= 2e-5 * scale


def test_model_training_step_with_native_msa_equals_stock_msa():
    """Model.forward (model.py:54-77) with the MSA news encoder feeding the DIGAT graph encoder: one training step (dropout 0)
    with the news encoder on its HIP pair against the same step with the news encoder on stock PyTorch modules — the autograd
    chain text -> MsaFused -> graph-encoder Functions -> loss, every parameter gradient."""
    from digat_amd import synthetic
    from digat_amd.model import Model
    B, K, N, H, C, Lw, V, dm, heads, dk, att, L = 4, 3, 4, 10, 5, 16, 200, 40, 4, 20, 24, 2
    cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=N, max_history_num=H, category_num=C,
                                graph_depth=L, dropout_rate=0.0, vocabulary_size=V, word_embedding_dim=dm, max_title_length=Lw,
                                MSA_head_num=heads, MSA_head_dim=dk, attention_dim=att)
    torch.manual_seed(3)
    model = Model(cfg)
    model.initialize()
    with torch.no_grad():
        model.graph_encoder.topic_node_embedding.normal_(0, 0.02)
        model.news_encoder.word_embedding.weight.mul_(0.1)         # GloVe-like magnitudes: logits of order one (with N(0,1) rows
        # the loss is ~10 and a 3e-7 difference between the two news encoders flips ReLU gates of the graph encoder: its own
        # gradients then differ by percents between the two runs — measured; an ill-conditioned test, not a defect)
    model = model.to(_dev()).train()
    flat = synthetic.make_encoder_batch(B * K, N, H, C, heads * dk, seed=5)
    users = synthetic.make_encoder_batch(B, N, H, C, heads * dk, seed=6, empty_history_rows=(1,))
    nt, nm = synthetic.make_titles(B * K * N, Lw, V, seed=7)
    ut, um = synthetic.make_titles(B * H, Lw, V, seed=8)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(_dev())
    args = (d(ut).view(B, H, Lw), d(um).view(B, H, Lw), d(users["user_graph"]), d(users["user_category_mask"]),
            d(users["user_category_indices"]), d(nt).view(B, K, N, Lw), d(nm).view(B, K, N, Lw),
            d(flat["news_graph"]).view(B, K, N, N), d(flat["news_graph_mask"]).view(B, K, N))

    def step():
        model.zero_grad(set_to_none=True)
        logits = model(*args)
        loss = (-torch.log_softmax(logits, dim=1).select(1, 0)).mean()
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}
    loss_hip, g_hip = step()
    enc = model.news_encoder
    enc.forward = enc.forward_stock
    try:
        loss_stock, g_stock = step()
    finally:
        del enc.forward
    assert abs(float(loss_hip) - float(loss_stock)) <= 1e-5 * max(1.0, abs(float(loss_stock)))
    for k in g_stock:
        _close(g_hip[k], g_stock[k].cpu().numpy(), "grad " + k, rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("T_,Lw,V,dm,h,dk,att", [(0, 16, 50, 32, 2, 8, 12), (1, 1, 30, 16, 1, 4, 4), (140, 20, 300, 64, 5, 16, 100),
                                                  (70, 32, 200, 36, 3, 32, 20), (5, 7, 40, 12, 4, 3, 8)])
def test_msa_training_shapes_against_the_oracle_autograd(T_, Lw, V, dm, h, dk, att):
    """Other shapes than the fixtures': no title, a one-token title, 2 800 token rows with 16-wide heads (the matrix-core paths
    with a padded attention_dim), the widest head (32), odd sizes on the fp32 paths — output and gradients vs the oracle."""
    from digat_amd import synthetic
    from oracle import news_oracle
    state = synthetic.make_msa_state(V, dm, h, dk, att, seed=T_ + Lw)
    text, mask = synthetic.make_titles(T_, Lw, V, seed=T_ + Lw + 1)
    R = np.random.default_rng(T_).standard_normal((T_, h * dk)).astype(np.float32)
    enc = _encoder(V, dm, h, dk, att, Lw, state).train()
    enc.dropout.p = 0.0
    out = enc(torch.from_numpy(text).to(_dev()).unsqueeze(0), torch.from_numpy(mask).to(_dev()).unsqueeze(0)).squeeze(0)
    (out * torch.from_numpy(R).to(_dev())).sum().backward()
    torch.cuda.synchronize()
    if T_ == 0:                                   # nothing to encode: an empty output and zero gradients
        assert out.shape == (0, h * dk)
        assert all(v.grad is not None and float(v.grad.abs().max()) == 0.0 for v in enc.parameters())
        return
    p = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in state.items()}
    want = news_oracle.msa_forward(p, torch.from_numpy(text), torch.from_numpy(mask), h)
    (want * torch.from_numpy(R)).sum().backward()
    _close(out, want.detach().numpy(), "news representation", rtol=1e-5, atol=2e-6)
    for k, v in enc.named_parameters():
        assert v.grad is not None, k
        _close(v.grad, p[k].grad.numpy(), "grad " + k)
