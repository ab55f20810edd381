#!/usr/bin/env python3
"""Mint golden vectors for the DIGAT hot path by running the REAL reference on CPU.

Runs only in the build container (needs /root/reference, read-only).  It imports the reference's
``graphEncoders.py`` / ``evaluate.py`` unchanged, feeds them the synthetic MIND-shaped inputs of
``digat_amd.synthetic`` and stores inputs/outputs as small ``.npz`` fixtures under
``tests/golden/``.  Only data is written: no reference source, bytecode or text.

Three modules the reference imports are not installable here (no network) and are stubbed in
``sys.modules`` before the import:
  * ``torchtext.vocab.GloVe``, ``sentence_transformers.SentenceTransformer`` — only pulled in
    transitively (config.py:7-8 -> MIND_corpus.py:8-9 -> construct_SAG.py:4); never called.
  * ``torch_scatter`` — ``scatter_softmax`` / ``scatter_sum`` (graphEncoders.py:7,129,130) are
    replaced by a deliberately naive per-segment loop written from the library's documented
    semantics (max-shifted exp, segment sum, empty segment -> 0).  The reference has no test that
    pins this boundary, so this stub *defines* it; the oracle's vectorised version is
    cross-checked against the same naive loop in tests/.

Usage:  python oracle/make_golden.py [name ...]   (no names: rewrites every fixture; deterministic)
"""
from __future__ import annotations

import io
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
GOLDEN = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from digat_amd import synthetic  # noqa: E402


# --------------------------------------------------------------------------------------------
# reference import harness
# --------------------------------------------------------------------------------------------
def _naive_scatter_softmax(src, index, dim=-1, eps=1e-12):
    assert dim in (1, -1) and src.dim() == 2
    out = torch.zeros_like(src)
    for b in range(src.shape[0]):
        for s in index[b].unique().tolist():
            sel = (index[b] == s).nonzero().flatten()
            v = src[b, sel]
            e = (v - v.max()).exp()
            out[b, sel] = e / e.sum()
    return out


def _naive_scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    assert dim == 1 and src.dim() == 3
    res = torch.zeros((src.shape[0], dim_size, src.shape[2]), dtype=src.dtype)
    for b in range(src.shape[0]):
        for t in range(src.shape[1]):          # sequential in t, like the CPU scatter_add
            res[b, int(index[b, t])] = res[b, int(index[b, t])] + src[b, t]
    return res


def import_reference():
    if not os.path.isdir(REFERENCE):
        raise SystemExit("make_golden.py needs /root/reference (build container only)")
    tt = types.ModuleType("torchtext")
    ttv = types.ModuleType("torchtext.vocab")
    ttv.GloVe = object
    tt.vocab = ttv
    st = types.ModuleType("sentence_transformers")
    st.SentenceTransformer = object
    ts = types.ModuleType("torch_scatter")
    ts.scatter_softmax = _naive_scatter_softmax
    ts.scatter_sum = _naive_scatter_sum
    sys.modules.update({"torchtext": tt, "torchtext.vocab": ttv, "sentence_transformers": st,
                        "torch_scatter": ts})
    sys.path.insert(0, REFERENCE)
    argv, sys.argv = sys.argv, ["x"]
    try:
        import graphEncoders  # noqa
        import evaluate       # noqa
    finally:
        sys.argv = argv
    return graphEncoders, evaluate


def reference_encoder(graphEncoders, N, H, C, d, L, state, dropout=0.0):
    cfg = types.SimpleNamespace(news_graph_size=N, max_history_num=H, category_num=C,
                                graph_depth=L, dropout_rate=dropout)
    enc = graphEncoders.DIGAT(cfg, d)
    enc.initialize()
    missing = enc.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return enc.eval()


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def save(name, **arrays):
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {os.path.relpath(path, REPO)}  {os.path.getsize(path) / 1024:.1f} KiB")


def checksum(batch, state):
    tot = 0.0
    for v in list(batch.values()) + list(state.values()):
        tot += float(np.asarray(v, dtype=np.float64).sum())
    return np.float64(tot)


def run_all_functions(enc, batch, L):
    """Outputs of every function on the path, eval mode."""
    b = {k: T(v) for k, v in batch.items()}
    H = b["user_news_embedding"].shape[1]
    out = {}
    with torch.no_grad():
        Xn, An, Mn = b["news_graph_embeddings"], b["news_graph"], b["news_graph_mask"]
        Au, cm, ci = b["user_graph"], b["user_category_mask"], b["user_category_indices"]
        Xu = torch.cat([b["user_news_embedding"],
                        enc.topic_node_embedding.unsqueeze(0).expand(Xn.shape[0], -1, -1)], dim=1)
        c_n0 = enc.compute_news_graph_context(Xn, Mn)                       # a3
        c_u0 = enc.compute_user_graph_context(Xu, cm, ci, c_n0)             # a4
        out["a3_news_ctx"] = c_n0.numpy()
        out["a4_user_ctx"] = c_u0.numpy()
        out["a6_sdpa_candidate"] = enc.candidate_attention(Xn, Xn[:, 0], mask=Mn).numpy()   # a6
        out["a1_news_emb_l0"] = enc.compute_news_graph_embeddings(0, Xn, An, c_u0).numpy()  # a1
        out["a2_user_emb_l0"] = enc.compute_user_graph_embeddings(0, Xu, Au, c_n0).numpy()  # a2
        fn, fu = enc.forward(Xn, An, Mn, b["user_news_embedding"], Au, cm, ci)              # a5
        out["a5_forward_news"], out["a5_forward_user"] = fn.numpy(), fu.numpy()
        inn, inu = enc.inference(Xn, An, Mn, b["user_news_embedding"], Au, cm, ci, c_n0)
        out["a5_inference_news"], out["a5_inference_user"] = inn.numpy(), inu.numpy()
        out["h1_logits"] = (inu * inn).sum(dim=1).numpy()                                   # model.py:89
    return out


# --------------------------------------------------------------------------------------------
# fixtures
# --------------------------------------------------------------------------------------------
def fixture_tiny(ge):
    """(i) BASELINE configs[0]: B=4, neighbors 3 / hops 1 -> N=4, H=10, C=5, d=64, L=1."""
    B, N, H, C, d, L = 4, 4, 10, 5, 64, 1
    state = synthetic.make_state_dict(d, C, L, seed=11, bias_std=0.05)
    batch = synthetic.make_encoder_batch(B, N, H, C, d, seed=12)
    enc = reference_encoder(ge, N, H, C, d, L, state)
    out = run_all_functions(enc, batch, L)
    save("tiny.npz", meta=np.array([B, N, H, C, d, L]),
         **{"in_" + k: v for k, v in batch.items()}, **{"w_" + k: v for k, v in state.items()},
         **{"out_" + k: v for k, v in out.items()})


def fixture_default(ge):
    """(ii) MIND-small default shape, one batch B=8: N=10, U=67, d=400, L=3.  Inputs and weights
    are regenerated from seeds (make_encoder_batch / make_state_dict); only outputs are stored."""
    for tag, (B, N, L, seed) in {"default_b8": (8, 10, 3, 21), "stress_b2": (2, 65, 7, 23),
                                 "codedefault_b4": (4, 26, 3, 25)}.items():
        H, C, d = 50, 17, 400
        state = synthetic.make_state_dict(d, C, L, seed=seed, bias_std=0.05)
        batch = synthetic.make_encoder_batch(B, N, H, C, d, seed=seed + 1)
        enc = reference_encoder(ge, N, H, C, d, L, state)
        out = run_all_functions(enc, batch, L)
        save(f"{tag}.npz", meta=np.array([B, N, H, C, d, L]), seeds=np.array([seed, seed + 1]),
             input_checksum=checksum(batch, state), **{"out_" + k: v for k, v in out.items()})


def fixture_edges(ge):
    """(iii) E1-E5: fully-masked news rows, empty histories, single-category histories, padded-only
    nodes, and adjacency rows with no edge at all (not produced by the reference's data prep, but
    the -1e9 fill gives them a defined uniform result the kernels must reproduce)."""
    B, N, H, C, d, L = 6, 10, 12, 4, 32, 2
    state = synthetic.make_state_dict(d, C, L, seed=31, bias_std=0.05)
    batch = synthetic.make_encoder_batch(B, N, H, C, d, seed=32, empty_history_rows=(1, 4),
                                         isolated_news_rows=(2, 4))
    # row 3: every history item in one category
    hist_len = np.full(B, H, dtype=np.int64)
    cats = np.zeros((B, H), dtype=np.int64)
    g, cm, ci = synthetic.build_user_graphs(cats, hist_len, C)
    batch["user_graph"][3], batch["user_category_mask"][3], batch["user_category_indices"][3] = g[3], cm[3], ci[3]
    # row 5: adjacency rows with no edge (incl. no self loop) on both graphs, one fully-masked
    batch["news_graph"][5, 1, :] = False
    batch["user_graph"][5, 0, :] = False
    batch["user_graph"][5, H + 1, :] = False
    enc = reference_encoder(ge, N, H, C, d, L, state)
    out = run_all_functions(enc, batch, L)
    save("edges.npz", meta=np.array([B, N, H, C, d, L]),
         **{"in_" + k: v for k, v in batch.items()}, **{"w_" + k: v for k, v in state.items()},
         **{"out_" + k: v for k, v in out.items()})


def reference_scores(enc, corpus, batch_size):
    """util.compute_scores' flow (util.py:34-69) with the reference encoder, on the synthetic corpus."""
    emb = T(corpus.news_embedding)
    sa = emb.index_select(0, T(corpus.news_node_ID.astype(np.int64)).flatten()).view(
        corpus.news_node_ID.shape[0], -1, emb.shape[1])
    masks = T(corpus.news_graph_mask)
    graphs = T(corpus.news_graph)
    with torch.no_grad():
        c_n0 = torch.cat([enc.compute_news_graph_context(sa[i:i + batch_size], masks[i:i + batch_size])
                          for i in range(0, sa.shape[0], batch_size)])
        scores = []
        for s in range(0, corpus.rows, batch_size):
            imp = T(corpus.row_impression[s:s + batch_size])
            cand = T(corpus.row_candidate[s:s + batch_size].astype(np.int64))
            hist = T(corpus.history.astype(np.int64)).index_select(0, imp)
            user_rep = emb.index_select(0, hist.flatten()).view(len(imp), -1, emb.shape[1])
            n, u = enc.inference(sa.index_select(0, cand), graphs.index_select(0, cand),
                                 masks.index_select(0, cand), user_rep,
                                 T(corpus.user_graph).index_select(0, imp),
                                 T(corpus.user_category_mask).index_select(0, imp),
                                 T(corpus.user_category_indices).index_select(0, imp),
                                 c_n0.index_select(0, cand))
            scores.append((u * n).sum(dim=1))
    return torch.cat(scores).numpy(), c_n0.numpy()


def fixture_devset(ge, ev, only=None):
    """(iv) synthetic dev sets scored by the reference encoder, ranked by util.py:70-80's rule,
    metrics by the reference's evaluate.scoring (sklearn AUC).  Specs: digat_amd.synthetic.DEVSET_FIXTURES."""
    specs = {tag: (synthetic.SynthSpec(**kw), L) for tag, (kw, L) in synthetic.DEVSET_FIXTURES.items() if only is None or tag in only}
    for tag, (spec, L) in specs.items():
        corpus = synthetic.make_corpus(spec)
        state = synthetic.make_state_dict(spec.embedding_dim, spec.category_num, L, seed=spec.seed + 1,
                                          bias_std=0.05)
        enc = reference_encoder(ge, spec.news_graph_size, spec.max_history_num, spec.category_num,
                                spec.embedding_dim, L, state)
        scores, c_n0 = reference_scores(enc, corpus, 128)
        # util.py:70-80
        sub = [[] for _ in range(int(corpus.row_impression[-1]) + 1)]
        for i, imp in enumerate(corpus.row_impression.tolist()):
            sub[imp].append([float(scores[i]), len(sub[imp])])
        lines, truth = [], []
        labels = [[] for _ in sub]
        for i, imp in enumerate(corpus.row_impression.tolist()):
            labels[imp].append(int(corpus.row_label[i]))
        for i, s in enumerate(sub):
            s.sort(key=lambda x: x[0], reverse=True)
            res = [0] * len(s)
            for j in range(len(s)):
                res[s[j][1]] = j + 1
            lines.append(str(i + 1) + " " + str(res).replace(" ", ""))
            truth.append(str(i + 1) + " " + str(labels[i]).replace(" ", ""))
        auc, mrr, n5, n10 = ev.scoring(io.StringIO("\n".join(truth)), io.StringIO("\n".join(lines)))
        print(f"  {tag}: rows={corpus.rows} AUC={auc:.6f} MRR={mrr:.6f} nDCG5={n5:.6f} nDCG10={n10:.6f}")
        save(f"{tag}.npz", depth=np.array(L), scores=scores.astype(np.float32),
             c_n0_head=c_n0[:64].astype(np.float32),
             rank_lines=np.array("\n".join(lines)), truth_lines=np.array("\n".join(truth)),
             metrics=np.array([auc, mrr, n5, n10], dtype=np.float64),
             input_checksum=np.float64(float(corpus.news_embedding.astype(np.float64).sum())
                                       + float(corpus.user_graph.sum()) + float(corpus.news_graph.sum())
                                       + float(corpus.row_candidate.astype(np.float64).sum())))


def fixture_train_step(ge):
    """(v) one training step, dropout 0 (so train mode is deterministic): loss of trainer.py:100,
    gradients of every parameter and of the two embedding inputs, from the reference's autograd."""
    B, K, N, H, C, d, L = 3, 5, 4, 10, 5, 32, 2
    state = synthetic.make_state_dict(d, C, L, seed=51, bias_std=0.05)
    flat = synthetic.make_encoder_batch(B * K, N, H, C, d, seed=52)
    users = synthetic.make_encoder_batch(B, N, H, C, d, seed=53)
    enc = reference_encoder(ge, N, H, C, d, L, state, dropout=0.0).train()
    Xn = T(flat["news_graph_embeddings"]).requires_grad_(True)
    ue = T(users["user_news_embedding"]).requires_grad_(True)

    def expand(t):                                                          # model.py:64-71
        return t.unsqueeze(1).expand(B, K, *t.shape[1:]).contiguous().view(B * K, *t.shape[1:])

    n, u = enc(Xn, T(flat["news_graph"]), T(flat["news_graph_mask"]), expand(ue),
               expand(T(users["user_graph"])), expand(T(users["user_category_mask"])),
               expand(T(users["user_category_indices"])))
    logits = (u.view(B, K, d) * n.view(B, K, d)).sum(dim=2)
    loss = (-torch.log_softmax(logits, dim=1).select(1, 0)).mean()
    loss.backward()
    grads = {"g_" + k: v.grad.numpy() for k, v in enc.named_parameters()}
    save("train_step.npz", meta=np.array([B, K, N, H, C, d, L]),
         in_news_graph_embeddings=flat["news_graph_embeddings"], in_news_graph=flat["news_graph"],
         in_news_graph_mask=flat["news_graph_mask"], in_user_news_embedding=users["user_news_embedding"],
         in_user_graph=users["user_graph"], in_user_category_mask=users["user_category_mask"],
         in_user_category_indices=users["user_category_indices"],
         **{"w_" + k: v for k, v in state.items()},
         out_logits=logits.detach().numpy(), out_loss=loss.detach().numpy(),
         g_in_news_graph_embeddings=Xn.grad.numpy(), g_in_user_news_embedding=ue.grad.numpy(), **grads)


def fixture_train_step_dropout(ge):
    """(v') the same step with dropout LIVE (rate 0.2): the reference's three nn.Dropout modules (graphEncoders.py:22-24) are swapped
    for modules that draw their masks from the HIP path's counter-hash generator (oracle.digat_oracle.hash_dropout), site k of the
    forward pass with seed 1001 + k — so the reference's own autograd yields the loss and every gradient under exactly the masks
    the kernels draw when digat_amd.training._seed hands out 1001, 1002, ... (tests/test_hip_training.py)."""
    from oracle import digat_oracle as O
    B, K, N, H, C, d, L = 3, 5, 4, 10, 5, 32, 2
    rate = 0.2
    state = synthetic.make_state_dict(d, C, L, seed=51, bias_std=0.05)
    flat = synthetic.make_encoder_batch(B * K, N, H, C, d, seed=52)
    users = synthetic.make_encoder_batch(B, N, H, C, d, seed=53)
    enc = reference_encoder(ge, N, H, C, d, L, state, dropout=rate).train()
    tape = O.SeedTape(rate, first=1001)

    class TapeDropout(torch.nn.Module):
        def __init__(self, frac):
            super().__init__()
            self.frac = frac

        def forward(self, x):
            return tape(x, self.frac) if self.training else x

    enc.dropout, enc.dropout_, enc.dropout__ = TapeDropout(1.0), TapeDropout(1.0), TapeDropout(0.5)
    Xn = T(flat["news_graph_embeddings"]).requires_grad_(True)
    ue = T(users["user_news_embedding"]).requires_grad_(True)

    def expand(t):                                                          # model.py:64-71
        return t.unsqueeze(1).expand(B, K, *t.shape[1:]).contiguous().view(B * K, *t.shape[1:])

    n, u = enc(Xn, T(flat["news_graph"]), T(flat["news_graph_mask"]), expand(ue),
               expand(T(users["user_graph"])), expand(T(users["user_category_mask"])),
               expand(T(users["user_category_indices"])))
    logits = (u.view(B, K, d) * n.view(B, K, d)).sum(dim=2)
    loss = (-torch.log_softmax(logits, dim=1).select(1, 0)).mean()
    loss.backward()
    grads = {"g_" + k: v.grad.numpy() for k, v in enc.named_parameters()}
    # the oracle's train mode (the same hook) must agree with the reference it restates
    p = O.as_params(state)
    lo = O.training_logits(p, L, T(users["user_news_embedding"]), T(users["user_graph"]), T(users["user_category_mask"]),
                           T(users["user_category_indices"]), T(flat["news_graph_embeddings"]).view(B, K, N, d),
                           T(flat["news_graph"]).view(B, K, N, N), T(flat["news_graph_mask"]).view(B, K, N),
                           drop=O.SeedTape(rate, first=1001))
    assert float((lo - logits.detach()).abs().max()) < 1e-5, "oracle train mode drifted from the reference"
    save("train_step_dropout.npz", meta=np.array([B, K, N, H, C, d, L]), dropout_rate=np.float64(rate),
         first_seed=np.int64(1001), sites=np.int64(tape.next - 1001),
         in_news_graph_embeddings=flat["news_graph_embeddings"], in_news_graph=flat["news_graph"],
         in_news_graph_mask=flat["news_graph_mask"], in_user_news_embedding=users["user_news_embedding"],
         in_user_graph=users["user_graph"], in_user_category_mask=users["user_category_mask"],
         in_user_category_indices=users["user_category_indices"],
         **{"w_" + k: v for k, v in state.items()},
         out_logits=logits.detach().numpy(), out_loss=loss.detach().numpy(),
         g_in_news_graph_embeddings=Xn.grad.numpy(), g_in_user_news_embedding=ue.grad.numpy(), **grads)


def grad_digest(name, g):
    """Compact, order-sensitive digest of one gradient tensor: its L2 norm, its dot product with a fixed pseudo-random
    probe (seeded by the parameter name), and every ``stride``-th element.  21 MB of production-shape gradients would not
    belong in the repository; a wrong element almost surely moves the probe, a wrong scale moves the norm, and the
    samples localise a failure."""
    flat = np.asarray(g, dtype=np.float32).reshape(-1)
    seed = int.from_bytes(name.encode()[-8:].rjust(8, b"\0"), "little") % (2 ** 32)
    probe = np.random.default_rng(seed).standard_normal(flat.size).astype(np.float32)
    stride = max(1, flat.size // 2048)
    return {"gn_" + name: np.float64(np.sqrt((flat.astype(np.float64) ** 2).sum())),
            "gp_" + name: np.float64((flat.astype(np.float64) * probe).sum()),
            "gs_" + name: flat[::stride].copy()}


def fixture_train_step_default(ge):
    """(v-b) one training step at the PRODUCTION shapes (MIND-small default: N=10, U=67, d=400, L=3; B=8 impressions x
    K=5 candidates = 40 rows -> 2 680 user-node rows, so the >= 2048-row GEMM paths are taken), dropout 0, from the
    reference's autograd.  Inputs and weights regenerate from seeds; logits, loss and the two input gradients are stored
    whole, parameter gradients as digests (``grad_digest``)."""
    B, K, N, H, C, d, L = 8, 5, 10, 50, 17, 400, 3
    seeds = (91, 92, 93)
    state = synthetic.make_state_dict(d, C, L, seed=seeds[0], bias_std=0.05)
    flat = synthetic.make_encoder_batch(B * K, N, H, C, d, seed=seeds[1])
    users = synthetic.make_encoder_batch(B, N, H, C, d, seed=seeds[2], empty_history_rows=(3,))
    enc = reference_encoder(ge, N, H, C, d, L, state, dropout=0.0).train()
    Xn = T(flat["news_graph_embeddings"]).requires_grad_(True)
    ue = T(users["user_news_embedding"]).requires_grad_(True)

    def expand(t):                                                          # model.py:64-71
        return t.unsqueeze(1).expand(B, K, *t.shape[1:]).contiguous().view(B * K, *t.shape[1:])

    n, u = enc(Xn, T(flat["news_graph"]), T(flat["news_graph_mask"]), expand(ue),
               expand(T(users["user_graph"])), expand(T(users["user_category_mask"])),
               expand(T(users["user_category_indices"])))
    logits = (u.view(B, K, d) * n.view(B, K, d)).sum(dim=2)
    loss = (-torch.log_softmax(logits, dim=1).select(1, 0)).mean()
    loss.backward()
    digests = {}
    for k, v in enc.named_parameters():
        digests.update(grad_digest(k, v.grad.numpy()))
    both = dict(flat)
    both.update({"u_" + k: v for k, v in users.items()})
    save("train_step_default.npz", meta=np.array([B, K, N, H, C, d, L]), seeds=np.array(seeds),
         input_checksum=checksum(both, state),
         out_logits=logits.detach().numpy(), out_loss=loss.detach().numpy(),
         out_news_ctx=n.detach().numpy(), out_user_ctx=u.detach().numpy(),
         g_in_news_graph_embeddings=Xn.grad.numpy(),
         g_in_user_news_embedding=ue.grad.numpy(), **digests)


def fixture_devset_2k(ge, ev):
    """(iv-b) a 2 000-impression dev set at the MIND-small default shapes (d=400, N=10, U=67, L=3, ~37 candidates per
    impression: ~74 k rows), scored by the reference encoder (about ten minutes of CPU in the build container), ranked by
    util.py:70-80's rule, metrics by evaluate.scoring.  Large enough that near-ties between candidates occur."""
    spec = synthetic.SynthSpec(news_num=4096, sag_neighbors=3, sag_hops=2, impressions=2000, seed=47)
    L = 3
    corpus = synthetic.make_corpus(spec)
    state = synthetic.make_state_dict(spec.embedding_dim, spec.category_num, L, seed=spec.seed + 1, bias_std=0.05)
    enc = reference_encoder(ge, spec.news_graph_size, spec.max_history_num, spec.category_num, spec.embedding_dim, L, state)
    scores, c_n0 = reference_scores(enc, corpus, 128)
    sub = [[] for _ in range(int(corpus.row_impression[-1]) + 1)]
    labels = [[] for _ in sub]
    for i, imp in enumerate(corpus.row_impression.tolist()):
        sub[imp].append([float(scores[i]), len(sub[imp])])
        labels[imp].append(int(corpus.row_label[i]))
    lines, truth = [], []
    for i, s in enumerate(sub):                                             # util.py:70-80
        s.sort(key=lambda x: x[0], reverse=True)
        res = [0] * len(s)
        for j in range(len(s)):
            res[s[j][1]] = j + 1
        lines.append(str(i + 1) + " " + str(res).replace(" ", ""))
        truth.append(str(i + 1) + " " + str(labels[i]).replace(" ", ""))
    auc, mrr, n5, n10 = ev.scoring(io.StringIO("\n".join(truth)), io.StringIO("\n".join(lines)))
    print(f"  devset_2k: rows={corpus.rows} AUC={auc:.6f} MRR={mrr:.6f} nDCG5={n5:.6f} nDCG10={n10:.6f}")
    ranks = np.concatenate([np.array(eval(l.split(" ", 1)[1]), dtype=np.int16) for l in lines])   # our own output lines
    save("devset_2k.npz", depth=np.array(L), scores=scores.astype(np.float32), ranks=ranks,
         c_n0_head=c_n0[:64].astype(np.float32), metrics=np.array([auc, mrr, n5, n10], dtype=np.float64),
         input_checksum=np.float64(float(corpus.news_embedding.astype(np.float64).sum())
                                   + float(corpus.user_graph.sum()) + float(corpus.news_graph.sum())
                                   + float(corpus.row_candidate.astype(np.float64).sum())))


def fixture_devset_trained_2k(ge, ev):
    """(iv-c) the dev split of the PLANTED-SIGNAL corpus scored by the reference with TRAINED weights.  Every other dev fixture
    uses Xavier weights on random clicks (AUC 0.5, logits of rms ~600: rank equality on widely spread scores).  Here the clicks
    follow sub-topic preferences (synthetic.PLANTED_SPEC), the graph encoder was trained on impressions [2000, 6000) of that
    corpus by tools/train_planted.py on the GPU box (digat_amd's own trainer: Adam 3e-4, 4 epochs, dropout 0.2) and its state
    dict travels as data (tests/golden/trained_planted_state.npz); the reference scores the held-out impressions [0, 2000):
    AUC ~0.64, logits of rms ~10, near-ties between candidates as a trained model has them."""
    full = synthetic.make_corpus(synthetic.SynthSpec(**synthetic.PLANTED_SPEC))
    corpus = synthetic.slice_impressions(full, 0, synthetic.PLANTED_DEV_IMPRESSIONS)
    spec = corpus.spec
    state = {k: v for k, v in np.load(os.path.join(GOLDEN, "trained_planted_state.npz")).items()}
    L = sum(1 for k in state if k.startswith("user_graph_attention_a."))
    enc = reference_encoder(ge, spec.news_graph_size, spec.max_history_num, spec.category_num, spec.embedding_dim, L, state)
    scores, c_n0 = reference_scores(enc, corpus, 128)
    sub = [[] for _ in range(int(corpus.row_impression[-1]) + 1)]
    labels = [[] for _ in sub]
    for i, imp in enumerate(corpus.row_impression.tolist()):
        sub[imp].append([float(scores[i]), len(sub[imp])])
        labels[imp].append(int(corpus.row_label[i]))
    lines, truth = [], []
    for i, s in enumerate(sub):                                             # util.py:70-80
        s.sort(key=lambda x: x[0], reverse=True)
        res = [0] * len(s)
        for j in range(len(s)):
            res[s[j][1]] = j + 1
        lines.append(str(i + 1) + " " + str(res).replace(" ", ""))
        truth.append(str(i + 1) + " " + str(labels[i]).replace(" ", ""))
    auc, mrr, n5, n10 = ev.scoring(io.StringIO("\n".join(truth)), io.StringIO("\n".join(lines)))
    print(f"  devset_trained_2k: rows={corpus.rows} AUC={auc:.6f} MRR={mrr:.6f} nDCG5={n5:.6f} nDCG10={n10:.6f} "
          f"logits rms {float(np.sqrt((scores.astype(np.float64) ** 2).mean())):.2f}")
    ranks = np.concatenate([np.array(eval(l.split(" ", 1)[1]), dtype=np.int16) for l in lines])   # our own output lines
    save("devset_trained_2k.npz", depth=np.array(L), scores=scores.astype(np.float32), ranks=ranks,
         c_n0_head=c_n0[:64].astype(np.float32), metrics=np.array([auc, mrr, n5, n10], dtype=np.float64),
         input_checksum=np.float64(float(corpus.news_embedding.astype(np.float64).sum())
                                   + float(corpus.user_graph.sum()) + float(corpus.news_graph.sum())
                                   + float(corpus.row_candidate.astype(np.float64).sum()) + float(corpus.row_label.sum())),
         state_checksum=np.float64(sum(float(np.asarray(v, dtype=np.float64).sum()) for v in state.values())))


def fixture_ablations(ge):
    """the five ablation encoders (graphEncoders.py:201-842), eval-mode forward and inference, at the tiny
    shapes (inputs stored) and at the default shapes (inputs and weights regenerate from seeds).  Loading the
    generated state dict with strict=True pins the parameter names of every class."""
    for name in ("wo_SA", "Seq_SA", "wo_interaction", "News_graph_wo_inter", "User_graph_wo_inter"):
        for tag, (B, N, H, C, d, L, seed) in {"tiny": (4, 4, 10, 5, 64, 2, 81), "default": (6, 10, 50, 17, 400, 3, 83)}.items():
            state = synthetic.make_ablation_state_dict(name, d, C, L, seed=seed)
            batch = synthetic.make_encoder_batch(B, N, H, C, d, seed=seed + 1, empty_history_rows=(1,), isolated_news_rows=(2,))
            cfg = types.SimpleNamespace(news_graph_size=N, max_history_num=H, category_num=C, graph_depth=L, dropout_rate=0.0)
            enc = getattr(ge, name)(cfg, d)
            enc.initialize()
            missing = enc.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
            assert not missing.missing_keys and not missing.unexpected_keys
            enc.eval()
            b = {k: T(v) for k, v in batch.items()}
            args = (b["news_graph_embeddings"], b["news_graph"], b["news_graph_mask"], b["user_news_embedding"], b["user_graph"],
                    b["user_category_mask"], b["user_category_indices"])
            with torch.no_grad():
                fn, fu = enc(*args)
                c0 = fn if name == "wo_SA" else (enc.compute_news_sequence_context if name == "Seq_SA"
                                                  else enc.compute_news_graph_context)(args[0], args[2])
                inn, inu = enc.inference(*args, c0)
            extra = {}
            if tag == "tiny" and name == "wo_interaction":      # one class keeps its inputs and weights on disk as well
                extra = dict(**{"in_" + k: v for k, v in batch.items()}, **{"w_" + k: v for k, v in state.items()})
            save(f"ablation_{name}_{tag}.npz", meta=np.array([B, N, H, C, d, L]), seeds=np.array([seed, seed + 1]),
                 input_checksum=checksum(batch, state), out_forward_news=fn.numpy(), out_forward_user=fu.numpy(),
                 out_inference_news=inn.numpy(), out_inference_user=inu.numpy(), **extra)


def fixture_ablation_train(ge):
    """SURVEY §8f-3, training: one training step (dropout 0, loss of trainer.py:100) of each of the five ablation encoders
    from the reference's autograd.  Tiny shapes with every gradient stored whole; wo_interaction (vanilla GAT on both graphs)
    also at the production shapes with parameter gradients as digests.  Inputs and weights regenerate from seeds."""
    cases = [(name, "tiny", (3, 4, 4, 10, 5, 32, 2, 85)) for name in
             ("wo_SA", "Seq_SA", "wo_interaction", "News_graph_wo_inter", "User_graph_wo_inter")]
    cases.append(("wo_interaction", "default", (8, 5, 10, 50, 17, 400, 3, 87)))
    for name, tag, (B, K, N, H, C, d, L, seed) in cases:
        state = synthetic.make_ablation_state_dict(name, d, C, L, seed=seed)
        flat = synthetic.make_encoder_batch(B * K, N, H, C, d, seed=seed + 1, isolated_news_rows=(2,))
        users = synthetic.make_encoder_batch(B, N, H, C, d, seed=seed + 2, empty_history_rows=(1,))
        cfg = types.SimpleNamespace(news_graph_size=N, max_history_num=H, category_num=C, graph_depth=L, dropout_rate=0.0)
        enc = getattr(ge, name)(cfg, d)
        enc.initialize()
        enc.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
        enc.train()
        Xn = T(flat["news_graph_embeddings"]).requires_grad_(True)
        ue = T(users["user_news_embedding"]).requires_grad_(True)

        def expand(t):                                                          # model.py:64-71
            return t.unsqueeze(1).expand(B, K, *t.shape[1:]).contiguous().view(B * K, *t.shape[1:])

        n, u = enc(Xn, T(flat["news_graph"]), T(flat["news_graph_mask"]), expand(ue), expand(T(users["user_graph"])),
                   expand(T(users["user_category_mask"])), expand(T(users["user_category_indices"])))
        logits = (u.view(B, K, d) * n.view(B, K, d)).sum(dim=2)
        loss = (-torch.log_softmax(logits, dim=1).select(1, 0)).mean()
        loss.backward()
        grads = {}
        for k, v in enc.named_parameters():
            if tag == "tiny":
                grads["g_" + k] = v.grad.numpy()
            else:
                grads.update(grad_digest(k, v.grad.numpy()))
        both = dict(flat)
        both.update({"u_" + k: v for k, v in users.items()})
        save(f"ablation_train_{name}_{tag}.npz", meta=np.array([B, K, N, H, C, d, L]), seeds=np.array([seed, seed + 1, seed + 2]),
             input_checksum=checksum(both, state), out_logits=logits.detach().numpy(), out_loss=loss.detach().numpy(),
             g_in_news_graph_embeddings=Xn.grad.numpy(), g_in_user_news_embedding=ue.grad.numpy(), **grads)


def fixture_msa():
    """MSA news encoder: the reference's own layers.MultiHeadAttention / layers.Attention modules composed
    as newsEncoders.MSA.forward composes them (newsEncoders.py:70-82; NewsEncoder.__init__ itself needs the dataset's
    word-embedding pickle, so the modules are built directly), eval mode."""
    import layers  # the reference's layers.py (already on sys.path)
    for tag, (T_, Lw, V, dm, h, dk, att, seed) in {"msa_tiny": (6, 8, 40, 20, 2, 8, 12, 51),
                                                   "msa_default": (12, 32, 500, 300, 16, 25, 256, 52)}.items():
        state = synthetic.make_msa_state(V, dm, h, dk, att, seed=seed)
        text, mask = synthetic.make_titles(T_, Lw, V, seed=seed + 1)
        mha = layers.MultiHeadAttention(h, dm, Lw, Lw, dk, dk)
        attn = layers.Attention(h * dk, att)
        mha.load_state_dict({k[len("multiheadSelfattention."):]: T(v) for k, v in state.items() if k.startswith("multiheadSelfattention.")})
        attn.load_state_dict({k[len("attention."):]: T(v) for k, v in state.items() if k.startswith("attention.")})
        with torch.no_grad():
            w = torch.nn.functional.embedding(T(text), T(state["word_embedding.weight"]))      # newsEncoders.py:76 (dropout off)
            hfeat = torch.relu(mha(w, w, w))                                                   # :78
            out = attn(hfeat, mask=T(mask.astype(np.int64)))                                   # :80
        extra = {}
        if tag == "msa_tiny":
            extra = dict(in_title_text=text, in_title_mask=mask, **{"w_" + k: v for k, v in state.items()})
        save(f"{tag}.npz", meta=np.array([T_, Lw, V, dm, h, dk, att]), seeds=np.array([seed, seed + 1]),
             input_checksum=checksum({"t": text, "m": mask}, state), out_news_representation=out.numpy(), **extra)


def fixture_msa_train():
    """MSA news encoder, training: the same module composition as fixture_msa, train mode with dropout 0, loss =
    sum(out * R) for a seeded random R, gradients of every parameter including the word embedding from the reference's autograd.
    Tiny shapes with whole gradients; the default shapes (16 heads x 25, 300-d words, 32 tokens; 72 titles = 2 304 token rows, so
    the >= 2048-row matrix-core paths are taken) with digests.  One title is all padding (its pooling is uniform, layers.py:111)."""
    import layers
    for tag, (T_, Lw, V, dm, h, dk, att, seed) in {"msa_train_tiny": (6, 8, 40, 20, 2, 8, 12, 55),
                                                   "msa_train_default": (72, 32, 500, 300, 16, 25, 256, 56)}.items():
        state = synthetic.make_msa_state(V, dm, h, dk, att, seed=seed)
        text, mask = synthetic.make_titles(T_, Lw, V, seed=seed + 1)
        text[2], mask[2] = 0, False                                                            # an empty (all-padding) title
        R = np.random.default_rng(seed + 2).standard_normal((T_, h * dk)).astype(np.float32)
        mha = layers.MultiHeadAttention(h, dm, Lw, Lw, dk, dk)
        attn = layers.Attention(h * dk, att)
        mha.load_state_dict({k[len("multiheadSelfattention."):]: T(v) for k, v in state.items() if k.startswith("multiheadSelfattention.")})
        attn.load_state_dict({k[len("attention."):]: T(v) for k, v in state.items() if k.startswith("attention.")})
        emb = torch.nn.Embedding(V, dm)
        emb.weight.data.copy_(T(state["word_embedding.weight"]))
        mha.train(); attn.train()
        w = emb(T(text))                                                                       # newsEncoders.py:76 (dropout 0)
        hfeat = torch.relu(mha(w, w, w))                                                       # :78
        out = attn(hfeat, mask=T(mask.astype(np.int64)))                                       # :80
        loss = (out * T(R)).sum()
        loss.backward()
        named = [("word_embedding.weight", emb.weight)] + [("multiheadSelfattention." + k, v) for k, v in mha.named_parameters()] \
            + [("attention." + k, v) for k, v in attn.named_parameters()]
        grads = {}
        for k, v in named:
            if tag == "msa_train_tiny":
                grads["g_" + k] = v.grad.numpy()
            else:
                grads.update(grad_digest(k, v.grad.numpy()))
        save(f"{tag}.npz", meta=np.array([T_, Lw, V, dm, h, dk, att]), seeds=np.array([seed, seed + 1, seed + 2]),
             input_checksum=checksum({"t": text, "m": mask, "r": R}, state), out_news_representation=out.detach().numpy(),
             out_loss=loss.detach().numpy(), **grads)


def fixture_sag():
    """SAG construction.  generate_news_graph (construct_SAG.py:449-485) is the reference function run
    unchanged on synthetic similarity dictionaries.  generate_cos_similarities (:112-162) moves its tensors with
    ``.cuda()`` / ``device='cuda'`` (no GPU in this container), so it is run with the module's ``torch`` global replaced
    by a proxy that drops the device (the same ATen CPU ops in the same order), writing its pickle cache under a scratch
    directory inside the repository that is removed afterwards."""
    import shutil
    import construct_SAG as ref   # the reference's module (sentence_transformers stubbed above)
    for tag, (news_num, top_M, hop, seed) in {"sag_graph_default": (300, 5, 2, 61), "sag_graph_small": (200, 3, 2, 62),
                                               "sag_graph_hop1": (120, 3, 1, 63), "sag_graph_hop3": (150, 4, 3, 64)}.items():
        rng = np.random.default_rng(seed)
        ids, cos, length = synthetic.make_similarity_lists(rng, news_num, top_M, isolated_frac=0.05)
        length[rng.random(news_num) < 0.2] = max(1, top_M - 2)                 # lists shorter than top_M
        length[0] = 0
        nn = synthetic.news_graph_size(top_M, hop)
        sim, news_ID_dict = synthetic.similarity_dict(ids, cos, length)
        node_ID, graph, mask = ref.generate_news_graph("x", sim, news_ID_dict, top_M, hop, nn)
        save(f"{tag}.npz", meta=np.array([news_num, top_M, hop, nn]), in_sim_index=ids, in_sim_cos=cos, in_sim_len=length,
             out_news_node_ID=node_ID, out_news_graph=np.packbits(graph), out_news_graph_mask=mask)

    class _CpuTorch:
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def zeros(*a, device=None, **k):
            return torch.zeros(*a, **k)

        @staticmethod
        def device(*_):
            return torch.device("cpu")

    scratch = os.path.join(REPO, ".golden_scratch")
    real_torch, real_cuda = ref.torch, torch.Tensor.cuda
    ref.torch = _CpuTorch()
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for tag, (n, m, dim, top_M, seed) in {"sag_cos_small": (40, 32, 64, 5, 71), "sag_cos_clamped": (24, 4, 32, 5, 72),
                                              "sag_cos_mpnet": (20, 16, 768, 5, 73)}.items():
            shutil.rmtree(scratch, ignore_errors=True)
            os.makedirs(os.path.join(scratch, "x-SAG", "cos"))
            title, content = synthetic.make_semantic_embeddings(max(n, m), dim, seed=seed)
            args = [T(title[:n]), T(content[:n]), T(title[:m]), T(content[:m])]   # the corpus is a subset of the category's news (rows [:n] query, [:m] corpus)
            res = ref.generate_cos_similarities(os.path.join(scratch, "x"), top_M, "c", *args)
            names = [f"out_{kind}_{what}" for kind in ("title", "content", "title_content", "content_title", "average")
                     for what in ("values", "indices")]
            save(f"{tag}.npz", meta=np.array([n, m, dim, top_M]), in_title_all=title, in_content_all=content,
                 **{k: v.numpy() for k, v in zip(names, res)})
    finally:
        ref.torch = real_torch
        torch.Tensor.cuda = real_cuda
        shutil.rmtree(scratch, ignore_errors=True)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    ge, ev = import_reference()
    print("reference imported from", REFERENCE)
    jobs = {"tiny": lambda: fixture_tiny(ge), "edges": lambda: fixture_edges(ge), "train_step": lambda: fixture_train_step(ge),
            "train_step_default": lambda: fixture_train_step_default(ge), "train_step_dropout": lambda: fixture_train_step_dropout(ge), "devset": lambda: fixture_devset(ge, ev, ("devset_tiny", "devset_default")),
            "devset_large": lambda: fixture_devset(ge, ev, ("devset_large",)), "devset_stress": lambda: fixture_devset(ge, ev, ("devset_stress",)),
            "default": lambda: fixture_default(ge), "ablations": lambda: fixture_ablations(ge),
            "ablation_train": lambda: fixture_ablation_train(ge), "msa": fixture_msa,
            "msa_train": fixture_msa_train,
            "sag": fixture_sag, "devset_2k": lambda: fixture_devset_2k(ge, ev),
            "devset_trained_2k": lambda: fixture_devset_trained_2k(ge, ev)}
    only = [a for a in sys.argv[1:] if not a.startswith("-")]        # python oracle/make_golden.py [name ...]
    for name in (only or list(jobs)):
        jobs[name]()


if __name__ == "__main__":
    main()
