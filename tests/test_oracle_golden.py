"""CPU suite: the oracle (oracle/digat_oracle.py) against the golden vectors minted from the
imported reference (oracle/make_golden.py).  No GPU, no /root/reference needed.

Tolerance: fp32, same ATen ops in the same order as the reference -> 1e-5 absolute / 1e-5 relative (values reach ~10 at depth 7).
"""
import numpy as np
import pytest
import torch

from conftest import check_grad_digest, load_golden, regenerate, regenerate_ablation_train, regenerate_train, split_fixture
from oracle import digat_oracle as O

RTOL, ATOL = 1e-5, 1e-5


def close(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b).max()
    assert np.allclose(a, b, rtol=RTOL, atol=ATOL), f"{what}: max|diff|={err:.3e}"


def run_oracle(batch, state, L):
    p = O.as_params(state)
    Xn, An, Mn, ue, Au, cm, ci = O.batch_tensors(batch)
    H = ue.shape[1]
    out = {}
    with torch.no_grad():
        Xu = O.user_nodes(p, ue)
        c_n0 = O.news_graph_context(p, Xn, Mn)
        c_u0 = O.user_graph_context(p, Xu, cm, ci, c_n0, H)
        out["a3_news_ctx"] = c_n0
        out["a4_user_ctx"] = c_u0
        out["a6_sdpa_candidate"] = O.scaled_dot_attention(p, "candidate_attention", Xn, Xn[:, 0], Mn)
        out["a1_news_emb_l0"] = O.cross_graph_attention(p, "news", 0, Xn, An, c_u0)
        out["a2_user_emb_l0"] = O.cross_graph_attention(p, "user", 0, Xu, Au, c_n0)
        out["a5_forward_news"], out["a5_forward_user"] = O.encoder_forward(p, L, Xn, An, Mn, ue, Au, cm, ci)
        out["a5_inference_news"], out["a5_inference_user"] = O.encoder_inference(p, L, Xn, An, Mn, ue, Au, cm, ci, c_n0)
        out["h1_logits"] = O.row_logits(p, L, ue, Au, cm, ci, Xn, An, Mn, c_n0)
    return {k: v.numpy() for k, v in out.items()}


@pytest.mark.parametrize("name", ["tiny.npz", "edges.npz"])
def test_oracle_matches_reference_stored_inputs(name):
    fx = load_golden(name)
    ins, w, outs = split_fixture(fx)
    L = int(fx["meta"][-1])
    got = run_oracle(ins, w, L)
    for k, v in outs.items():
        close(got[k], v, f"{name}:{k}")


@pytest.mark.parametrize("name", ["default_b8.npz", "codedefault_b4.npz", "stress_b2.npz"])
def test_oracle_matches_reference_regenerated_inputs(name):
    fx = load_golden(name)
    batch, state = regenerate(fx)
    L = int(fx["meta"][-1])
    got = run_oracle(batch, state, L)
    for k, v in fx.items():
        if k.startswith("out_"):
            close(got[k[4:]], v, f"{name}:{k}")


def test_forward_equals_inference_in_eval():
    fx = load_golden("tiny.npz")
    _, _, outs = split_fixture(fx)
    assert np.array_equal(outs["a5_forward_news"], outs["a5_inference_news"])
    assert np.array_equal(outs["a5_forward_user"], outs["a5_inference_user"])


def test_segment_ops_against_naive_loop():
    rng = np.random.default_rng(5)
    B, H, C1, d = 7, 23, 6, 9
    a = rng.standard_normal((B, H)).astype(np.float32) * 3
    idx = rng.integers(0, C1, size=(B, H))
    idx[0] = 2                      # one segment holds everything
    idx[1] = np.arange(H) % C1
    got = O.segment_softmax(torch.from_numpy(a), torch.from_numpy(idx), C1).numpy()
    want = O.segment_softmax_naive(a, idx, C1)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7)
    src = rng.standard_normal((B, H, d)).astype(np.float32)
    got = O.segment_sum(torch.from_numpy(src), torch.from_numpy(idx), C1).numpy()
    want = np.zeros((B, C1, d), np.float32)
    for b in range(B):
        for t in range(H):
            want[b, idx[b, t]] += src[b, t]
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    assert np.all(got[0, [0, 1, 3, 4, 5]] == 0)          # empty segments are exactly zero


def test_fully_masked_attention_row_is_uniform():
    """E1: -1e9 (not -inf) => an all-masked row averages every node, padded ones included."""
    fx = load_golden("edges.npz")
    ins, w, outs = split_fixture(fx)
    X = ins["news_graph_embeddings"]
    assert not ins["news_graph_mask"][2].any()
    np.testing.assert_allclose(outs["a6_sdpa_candidate"][2], X[2].mean(axis=0), rtol=1e-5, atol=1e-6)


def test_train_step_loss_and_gradients():
    fx = load_golden("train_step.npz")
    B, K, N, H, C, d, L = (int(v) for v in fx["meta"])
    ins, w, outs = split_fixture(fx)
    p = {k: v.clone().requires_grad_(True) for k, v in O.as_params(w).items()}
    Xn = torch.from_numpy(ins["news_graph_embeddings"]).view(B, K, N, d).clone().requires_grad_(True)
    ue = torch.from_numpy(ins["user_news_embedding"]).clone().requires_grad_(True)
    logits = O.training_logits(p, L, ue, torch.from_numpy(ins["user_graph"]),
                               torch.from_numpy(ins["user_category_mask"]),
                               torch.from_numpy(ins["user_category_indices"]), Xn,
                               torch.from_numpy(ins["news_graph"]).view(B, K, N, N),
                               torch.from_numpy(ins["news_graph_mask"]).view(B, K, N))
    loss = O.training_loss(logits)
    loss.backward()
    close(logits.detach().numpy(), outs["logits"], "logits")
    close(loss.detach().numpy(), outs["loss"], "loss")
    close(Xn.grad.view(B * K, N, d).numpy(), fx["g_in_news_graph_embeddings"], "dX_news")
    close(ue.grad.numpy(), fx["g_in_user_news_embedding"], "dX_user")
    for k, v in p.items():
        close(v.grad.numpy(), fx["g_" + k], "grad " + k)


def test_train_step_with_dropout_live_under_the_kernels_masks():
    """train_step_dropout.npz: the REFERENCE's autograd with its three nn.Dropout modules drawing the HIP path's counter-hash masks
    (seed 1001 + site).  The oracle's train mode (the same hook) must reproduce loss and every gradient; the mask generator's
    restatement is pinned here too (the fixture's values depend on every mask bit)."""
    fx = load_golden("train_step_dropout.npz")
    B, K, N, H, C, d, L = (int(v) for v in fx["meta"])
    ins, w, outs = split_fixture(fx)
    p = {k: v.clone().requires_grad_(True) for k, v in O.as_params(w).items()}
    Xn = torch.from_numpy(ins["news_graph_embeddings"]).view(B, K, N, d).clone().requires_grad_(True)
    ue = torch.from_numpy(ins["user_news_embedding"]).clone().requires_grad_(True)
    tape = O.SeedTape(float(fx["dropout_rate"]), first=int(fx["first_seed"]))
    logits = O.training_logits(p, L, ue, torch.from_numpy(ins["user_graph"]), torch.from_numpy(ins["user_category_mask"]),
                               torch.from_numpy(ins["user_category_indices"]), Xn,
                               torch.from_numpy(ins["news_graph"]).view(B, K, N, N),
                               torch.from_numpy(ins["news_graph_mask"]).view(B, K, N), drop=tape)
    assert tape.next - int(fx["first_seed"]) == int(fx["sites"]) == 3 + 6 * L       # topics, c_n0, c_u0; per layer 2 x (X, alpha) + 2 contexts
    loss = O.training_loss(logits)
    loss.backward()
    close(logits.detach().numpy(), outs["logits"], "logits")
    close(loss.detach().numpy(), outs["loss"], "loss")
    close(Xn.grad.view(B * K, N, d).numpy(), fx["g_in_news_graph_embeddings"], "dX_news")
    close(ue.grad.numpy(), fx["g_in_user_news_embedding"], "dX_user")
    for k, v in p.items():
        close(v.grad.numpy(), fx["g_" + k], "grad " + k)
    keep = O.hash_dropout_keep(1 << 20, 0.2, 7)
    assert abs(keep.mean() - 0.8) < 2e-3 and not np.array_equal(keep, O.hash_dropout_keep(1 << 20, 0.2, 8))


@pytest.mark.parametrize("name", ["devset_tiny.npz", "devset_default.npz", "devset_large.npz", "devset_stress.npz"])
def test_devset_scores_ranks_metrics(name):
    from digat_amd import synthetic
    fx = load_golden(name)
    spec = synthetic.SynthSpec(**synthetic.DEVSET_FIXTURES[name[:-4]][0])
    assert int(fx["depth"]) == synthetic.DEVSET_FIXTURES[name[:-4]][1]
    corpus = synthetic.make_corpus(spec)
    L = int(fx["depth"])
    p = O.as_params(synthetic.make_state_dict(spec.embedding_dim, spec.category_num, L,
                                              seed=spec.seed + 1, bias_std=0.05))
    emb = torch.from_numpy(corpus.news_embedding)
    ids = torch.from_numpy(corpus.news_node_ID.astype(np.int64))
    sa = emb.index_select(0, ids.flatten()).view(ids.shape[0], -1, emb.shape[1])
    masks, graphs = torch.from_numpy(corpus.news_graph_mask), torch.from_numpy(corpus.news_graph)
    with torch.no_grad():
        c_n0 = O.news_graph_context(p, sa, masks)
        close(c_n0[:64].numpy(), fx["c_n0_head"], "c_n0")
        imp = torch.from_numpy(corpus.row_impression)
        cand = torch.from_numpy(corpus.row_candidate.astype(np.int64))
        hist = torch.from_numpy(corpus.history.astype(np.int64)).index_select(0, imp)
        ue = emb.index_select(0, hist.flatten()).view(len(imp), -1, emb.shape[1])
        scores = O.row_logits(p, L, ue, torch.from_numpy(corpus.user_graph).index_select(0, imp),
                              torch.from_numpy(corpus.user_category_mask).index_select(0, imp),
                              torch.from_numpy(corpus.user_category_indices).index_select(0, imp),
                              sa.index_select(0, cand), graphs.index_select(0, cand),
                              masks.index_select(0, cand), c_n0.index_select(0, cand)).numpy()
    np.testing.assert_allclose(scores, fx["scores"], rtol=1e-4, atol=2e-5)
    # ranking rule + metrics restatement, fed the reference's own scores so ties cannot differ
    ranks = O.impression_ranks(fx["scores"].tolist(), corpus.row_impression.tolist())
    assert "\n".join(O.rank_lines(ranks)) == str(fx["rank_lines"])
    labels = [[] for _ in ranks]
    for i, r in zip(corpus.row_impression.tolist(), corpus.row_label.tolist()):
        labels[i].append(int(r))
    got = O.ranking_metrics(labels, ranks)
    np.testing.assert_allclose(got, fx["metrics"], rtol=0, atol=1e-9)
    # and end to end from the oracle's own scores: the repo-stated metric tolerance
    got2 = O.ranking_metrics(labels, O.impression_ranks(scores.tolist(), corpus.row_impression.tolist()))
    np.testing.assert_allclose(got2, fx["metrics"], rtol=0, atol=1e-4)


# --------------------------------------------------------------------------------------------------
# MSA news encoder: oracle/news_oracle.py vs vectors minted from the reference's layers.py modules
# --------------------------------------------------------------------------------------------------
def _msa_case(name):
    from digat_amd import synthetic
    fx = load_golden(name)
    T_, Lw, V, dm, h, dk, att = (int(v) for v in fx["meta"])
    s_w, s_t = (int(v) for v in fx["seeds"])
    state = synthetic.make_msa_state(V, dm, h, dk, att, seed=s_w)
    text, mask = synthetic.make_titles(T_, Lw, V, seed=s_t)
    tot = float(np.asarray(text, dtype=np.float64).sum() + np.asarray(mask, dtype=np.float64).sum())
    tot += sum(float(np.asarray(v, dtype=np.float64).sum()) for v in state.values())
    assert abs(tot - float(fx["input_checksum"])) < 1e-6 * max(1.0, abs(tot)), "regenerated MSA inputs differ from the minted ones"
    if "in_title_text" in fx:
        assert np.array_equal(fx["in_title_text"], text) and np.array_equal(fx["in_title_mask"], mask)
    return state, text, mask, h, fx["out_news_representation"]


@pytest.mark.parametrize("name", ["msa_tiny.npz", "msa_default.npz"])
def test_msa_oracle_matches_reference_layers(name):
    from oracle import news_oracle
    state, text, mask, h, want = _msa_case(name)
    p = {k: torch.from_numpy(v) for k, v in state.items()}
    with torch.no_grad():
        got = news_oracle.msa_forward(p, torch.from_numpy(text), torch.from_numpy(mask), h).numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


# --------------------------------------------------------------------------------------------------
# the five ablation encoders — oracle vs vectors minted from the reference's classes
# --------------------------------------------------------------------------------------------------
def _ablation_case(name, tag):
    from digat_amd import synthetic
    fx = load_golden(f"ablation_{name}_{tag}.npz")
    B, N, H, C, d, L = (int(v) for v in fx["meta"])
    s_w, s_b = (int(v) for v in fx["seeds"])
    state = synthetic.make_ablation_state_dict(name, d, C, L, seed=s_w)
    batch = synthetic.make_encoder_batch(B, N, H, C, d, seed=s_b, empty_history_rows=(1,), isolated_news_rows=(2,))
    tot = sum(float(np.asarray(v, dtype=np.float64).sum()) for v in list(batch.values()) + list(state.values()))
    assert abs(tot - float(fx["input_checksum"])) < 1e-6 * max(1.0, abs(tot)), "regenerated inputs differ from the minted ones"
    return fx, state, batch, L


@pytest.mark.parametrize("tag", ["tiny", "default"])
@pytest.mark.parametrize("name", list(O.ABLATIONS))
def test_ablation_oracle_matches_reference(name, tag):
    fx, state, batch, L = _ablation_case(name, tag)
    p = O.as_params(state)
    b = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in batch.items()}
    args = (b["news_graph_embeddings"], b["news_graph"], b["news_graph_mask"], b["user_news_embedding"], b["user_graph"],
            b["user_category_mask"], b["user_category_indices"])
    with torch.no_grad():
        fn, fu = O.ablation_encode(name, p, L, *args)
        c0 = args[0][:, 0] if name == "wo_SA" else O.news_graph_context(p, args[0], args[2])
        inn, inu = O.ablation_encode(name, p, L, *args, c_n=c0)
    for got, key in ((fn, "out_forward_news"), (fu, "out_forward_user"), (inn, "out_inference_news"), (inu, "out_inference_user")):
        np.testing.assert_allclose(got.numpy(), fx[key], rtol=1e-5, atol=ATOL, err_msg=f"{name}/{tag}/{key}")


def test_train_step_at_production_shapes():
    """The oracle's autograd against the reference's at N=10, U=67, d=400, L=3, 40 rows (train_step_default.npz: logits,
    loss, contexts and input gradients whole, parameter gradients as digests)."""
    fx = load_golden("train_step_default.npz")
    (B, K, N, H, C, d, L), w, flat, users = regenerate_train(fx)
    p = {k: v.clone().requires_grad_(True) for k, v in O.as_params(w).items()}
    Xn = torch.from_numpy(flat["news_graph_embeddings"]).view(B, K, N, d).clone().requires_grad_(True)
    ue = torch.from_numpy(users["user_news_embedding"]).clone().requires_grad_(True)
    logits = O.training_logits(p, L, ue, torch.from_numpy(users["user_graph"]),
                               torch.from_numpy(users["user_category_mask"]),
                               torch.from_numpy(users["user_category_indices"]), Xn,
                               torch.from_numpy(flat["news_graph"]).view(B, K, N, N),
                               torch.from_numpy(flat["news_graph_mask"]).view(B, K, N))
    loss = O.training_loss(logits)
    loss.backward()
    close(logits.detach().numpy(), fx["out_logits"], "logits")
    close(loss.detach().numpy(), fx["out_loss"], "loss")
    close(Xn.grad.view(B * K, N, d).numpy(), fx["g_in_news_graph_embeddings"], "dX_news")
    close(ue.grad.numpy(), fx["g_in_user_news_embedding"], "dX_user")
    for k, v in p.items():
        check_grad_digest(fx, k, v.grad.numpy(), 2e-5, "oracle grad ")


def test_oracle_on_a_slice_of_the_2k_devset():
    """devset_2k.npz (2 000 impressions scored by the reference): the oracle on its first impressions (a CPU-sized slice);
    the GPU suite holds the HIP path to all 74 k rows (tests/test_hip_lowprec.py)."""
    from digat_amd import synthetic
    fx = load_golden("devset_2k.npz")
    spec = synthetic.SynthSpec(news_num=4096, sag_neighbors=3, sag_hops=2, impressions=2000, seed=47)
    corpus = synthetic.make_corpus(spec)
    L = int(fx["depth"])
    p = O.as_params(synthetic.make_state_dict(spec.embedding_dim, spec.category_num, L, seed=spec.seed + 1, bias_std=0.05))
    rows = int(np.searchsorted(corpus.row_impression, 8))            # the first 8 impressions
    emb = torch.from_numpy(corpus.news_embedding)
    cand = torch.from_numpy(corpus.row_candidate[:rows].astype(np.int64))
    ids = torch.from_numpy(corpus.news_node_ID.astype(np.int64)).index_select(0, cand)
    sa = emb.index_select(0, ids.flatten()).view(rows, -1, spec.embedding_dim)
    masks = torch.from_numpy(corpus.news_graph_mask).index_select(0, cand)
    graphs = torch.from_numpy(corpus.news_graph).index_select(0, cand)
    imp = torch.from_numpy(corpus.row_impression[:rows])
    hist = torch.from_numpy(corpus.history.astype(np.int64)).index_select(0, imp)
    ue = emb.index_select(0, hist.flatten()).view(rows, spec.max_history_num, spec.embedding_dim)
    with torch.no_grad():
        c_n0 = O.news_graph_context(p, sa, masks)
        got = O.row_logits(p, L, ue, torch.from_numpy(corpus.user_graph).index_select(0, imp),
                           torch.from_numpy(corpus.user_category_mask).index_select(0, imp),
                           torch.from_numpy(corpus.user_category_indices).index_select(0, imp), sa, graphs, masks, c_n0).numpy()
    want = fx["scores"][:rows]
    assert np.allclose(got, want, rtol=2e-5, atol=1e-4), np.abs(got - want).max()


def test_oracle_on_a_slice_of_the_trained_devset():
    """devset_trained_2k.npz: TRAINED weights (AUC 0.64, logits of rms ~10) on the planted-signal corpus, scored by the reference.
    The oracle on the first impressions; the fixture's labels are not random: its stored metrics must say the model ranks."""
    from conftest import planted_devset
    fx, corpus, state = planted_devset()
    assert fx["metrics"][0] > 0.60 and 3.0 < float(np.sqrt((fx["scores"].astype(np.float64) ** 2).mean())) < 30.0
    L = int(fx["depth"])
    p = O.as_params(state)
    spec = corpus.spec
    rows = int(np.searchsorted(corpus.row_impression, 8))            # the first 8 impressions
    emb = torch.from_numpy(corpus.news_embedding)
    cand = torch.from_numpy(corpus.row_candidate[:rows].astype(np.int64))
    ids = torch.from_numpy(corpus.news_node_ID.astype(np.int64)).index_select(0, cand)
    sa = emb.index_select(0, ids.flatten()).view(rows, -1, spec.embedding_dim)
    masks = torch.from_numpy(corpus.news_graph_mask).index_select(0, cand)
    graphs = torch.from_numpy(corpus.news_graph).index_select(0, cand)
    imp = torch.from_numpy(corpus.row_impression[:rows])
    hist = torch.from_numpy(corpus.history.astype(np.int64)).index_select(0, imp)
    ue = emb.index_select(0, hist.flatten()).view(rows, spec.max_history_num, spec.embedding_dim)
    with torch.no_grad():
        c_n0 = O.news_graph_context(p, sa, masks)
        got = O.row_logits(p, L, ue, torch.from_numpy(corpus.user_graph).index_select(0, imp),
                           torch.from_numpy(corpus.user_category_mask).index_select(0, imp),
                           torch.from_numpy(corpus.user_category_indices).index_select(0, imp), sa, graphs, masks, c_n0).numpy()
    want = fx["scores"][:rows]
    assert np.allclose(got, want, rtol=2e-5, atol=2e-5), np.abs(got - want).max()


def test_planted_signal_corpus_is_rankable_and_the_default_corpus_is_unchanged():
    """synthetic.SynthSpec(signal=...): a trivial scorer (mean history embedding . candidate embedding) ranks the planted corpus
    well above chance; with signal = 0 the generator is bit for bit what the older fixtures were minted with."""
    from digat_amd import evaluate, synthetic
    spec = synthetic.SynthSpec(news_num=1024, impressions=300, seed=3, signal=2.0, embedding_scale=0.12)
    c = synthetic.make_corpus(spec)
    valid = c.history > 0
    um = (c.news_embedding[c.history] * valid[..., None]).sum(1) / np.maximum(valid.sum(1, keepdims=True), 1)
    sc = (um[c.row_impression] * c.news_embedding[c.row_candidate]).sum(1)
    auc = evaluate.scoring(c.row_label, evaluate.impression_ranks(sc, c.row_impression), c.row_impression)[0]
    assert auc > 0.62, auc
    sub = synthetic.slice_impressions(c, 100, 200)
    assert sub.spec.impressions == 100 and sub.row_impression.min() == 0 and sub.row_impression.max() == 99
    assert sub.rows == int(((c.row_impression >= 100) & (c.row_impression < 200)).sum())
    fx = load_golden("devset_2k.npz")                            # minted before the signal option existed
    base = synthetic.make_corpus(synthetic.SynthSpec(news_num=4096, sag_neighbors=3, sag_hops=2, impressions=2000, seed=47))
    chk = (float(base.news_embedding.astype(np.float64).sum()) + float(base.user_graph.sum()) + float(base.news_graph.sum())
           + float(base.row_candidate.astype(np.float64).sum()))
    assert abs(chk - float(fx["input_checksum"])) <= 1e-6 * abs(chk)


ABLATION_TRAIN = [(n, "tiny") for n in O.ABLATIONS] + [("wo_interaction", "default")]


@pytest.mark.parametrize("name,tag", ABLATION_TRAIN)
def test_ablation_train_step_matches_reference_autograd(name, tag):
    """One training step (dropout 0) of every ablation encoder: the oracle's autograd against the reference's
    (ablation_train_*.npz; tiny shapes with whole gradients, wo_interaction also at N=10, U=67, d=400 with digests)."""
    fx = load_golden(f"ablation_train_{name}_{tag}.npz")
    (B, K, N, H, C, d, L), w, flat, users = regenerate_ablation_train(fx, name)
    p = {k: v.clone().requires_grad_(True) for k, v in O.as_params(w).items()}
    Xn = torch.from_numpy(flat["news_graph_embeddings"]).clone().requires_grad_(True)
    ue = torch.from_numpy(users["user_news_embedding"]).clone().requires_grad_(True)

    def expand(t):
        t = torch.from_numpy(t) if isinstance(t, np.ndarray) else t
        return t.unsqueeze(1).expand(B, K, *t.shape[1:]).reshape(B * K, *t.shape[1:])

    n, u = O.ablation_encode(name, p, L, Xn, torch.from_numpy(flat["news_graph"]), torch.from_numpy(flat["news_graph_mask"]),
                             expand(ue), expand(users["user_graph"]), expand(users["user_category_mask"]),
                             expand(users["user_category_indices"]))
    logits = (u.view(B, K, d) * n.view(B, K, d)).sum(dim=2)
    loss = O.training_loss(logits)
    loss.backward()
    close(logits.detach().numpy(), fx["out_logits"], "logits")
    close(loss.detach().numpy(), fx["out_loss"], "loss")
    close(Xn.grad.numpy(), fx["g_in_news_graph_embeddings"], "dX_news")
    close(ue.grad.numpy(), fx["g_in_user_news_embedding"], "dX_user")
    for k, v in p.items():
        if tag == "tiny":
            close(v.grad.numpy(), fx["g_" + k], "grad " + k)
        else:
            check_grad_digest(fx, k, v.grad.numpy(), 2e-5, "grad ")


@pytest.mark.parametrize("name", ["msa_train_tiny.npz", "msa_train_default.npz"])
def test_msa_train_step_oracle_matches_reference_autograd(name):
    """The news oracle's autograd (oracle/news_oracle.py) against the reference modules' (msa_train_*.npz)."""
    from digat_amd import synthetic
    from oracle import news_oracle
    fx = load_golden(name)
    T_, Lw, V, dm, h, dk, att = (int(v) for v in fx["meta"])
    s_w, s_t, s_r = (int(v) for v in fx["seeds"])
    state = synthetic.make_msa_state(V, dm, h, dk, att, seed=s_w)
    text, mask = synthetic.make_titles(T_, Lw, V, seed=s_t)
    text[2], mask[2] = 0, False
    R = np.random.default_rng(s_r).standard_normal((T_, h * dk)).astype(np.float32)
    p = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in state.items()}
    out = news_oracle.msa_forward(p, torch.from_numpy(text), torch.from_numpy(mask), h)
    loss = (out * torch.from_numpy(R)).sum()
    loss.backward()
    close(out.detach().numpy(), fx["out_news_representation"], "news representation")
    close(loss.detach().numpy(), fx["out_loss"], "loss")
    for k, v in p.items():
        if "g_" + k in fx:
            close(v.grad.numpy(), fx["g_" + k], "grad " + k)
        else:
            check_grad_digest(fx, k, v.grad.numpy(), 2e-5, "grad ")
