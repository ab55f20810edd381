"""ORACLE (test infrastructure, not product code): CPU restatement of DIGAT's dual-graph hot path.

This file restates, in plain fp32 torch-CPU ops, the *unfused* algorithm of the reference's
``graphEncoders.DIGAT`` and of the callers either side of it.  It exists so that the HIP path can
be checked against the reference's results on a box where ``/root/reference`` does not exist.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package ``digat_amd`` never does.

Pinning: the reference ships no tests and no golden vectors for this path, so the
oracle is pinned against outputs of the reference itself: ``oracle/make_golden.py`` imports
``/root/reference/graphEncoders.py`` (in the build container only) and writes the fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` holds this file to them.

Third-party arithmetic restated here: ``torch_scatter.scatter_softmax`` / ``scatter_sum``
(rusty1s/pytorch_scatter, pinned 2.0.9 in README.md:12 and 2.1.1 in install_dependencies.sh:16;
not vendored, not installable here): per-segment max-shifted exp / segment sum, empty segments
sum to zero.  The 2.0.x ``eps=1e-12`` in the denominator is below fp32 resolution because every
non-empty segment's denominator is >= 1.

Parameters are a dict of fp32 tensors keyed by the reference's state_dict names
(graphEncoders.py:52-73), ``nn.Linear`` weights are [out, in].
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]
MASK_FILL = -1e9  # graphEncoders.py:152,172 ; layers.py:202 — NOT -inf


def as_params(state: Dict[str, "np.ndarray | torch.Tensor"]) -> Params:
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))).float()
            for k, v in state.items()}


def _linear(x: torch.Tensor, p: Params, name: str) -> torch.Tensor:
    return F.linear(x, p[name + ".weight"], p.get(name + ".bias"))


# ------------------------------------------------------------------------------------------
# torch_scatter restatement (call sites graphEncoders.py:129-130)
# ------------------------------------------------------------------------------------------
def segment_softmax(src: torch.Tensor, index: torch.Tensor, segments: int) -> torch.Tensor:
    """scatter_softmax(src[B,H], index[B,H], dim=1): softmax within groups of equal index."""
    B, H = src.shape
    seg_max = torch.full((B, segments), -float("inf"), dtype=src.dtype)
    seg_max = seg_max.scatter_reduce(1, index, src, reduce="amax", include_self=True)
    shifted = (src - seg_max.gather(1, index)).exp()
    seg_sum = torch.zeros((B, segments), dtype=src.dtype).scatter_add(1, index, shifted)
    return shifted / seg_sum.gather(1, index)


def segment_sum(src: torch.Tensor, index: torch.Tensor, segments: int) -> torch.Tensor:
    """scatter_sum(src[B,H,d], index[B,H], dim=1, dim_size=segments); empty segments -> 0."""
    B, H, d = src.shape
    out = torch.zeros((B, segments, d), dtype=src.dtype)
    return out.scatter_add(1, index.unsqueeze(2).expand(B, H, d), src)


def segment_softmax_naive(src: np.ndarray, index: np.ndarray, segments: int) -> np.ndarray:
    """Pure-Python per-segment loop; the cross-check for ``segment_softmax`` (small cases only)."""
    out = np.zeros_like(src)
    for b in range(src.shape[0]):
        for s in range(segments):
            sel = np.nonzero(index[b] == s)[0]
            if sel.size == 0:
                continue
            v = src[b, sel]
            e = np.exp(v - v.max())
            out[b, sel] = e / e.sum()
    return out


# ------------------------------------------------------------------------------------------
# a6: ScaledDotProductAttention (layers.py:199-206)
# ------------------------------------------------------------------------------------------
def scaled_dot_attention(p: Params, name: str, feature: torch.Tensor, query: torch.Tensor,
                         mask: torch.Tensor) -> torch.Tensor:
    d_att = p[name + ".K.weight"].shape[0]
    keys = F.linear(feature, p[name + ".K.weight"])                         # [B,n,d_att]
    q = F.linear(query, p[name + ".Q.weight"], p[name + ".Q.bias"])         # [B,d_att]
    a = torch.bmm(keys, q.unsqueeze(2)).squeeze(2) / math.sqrt(float(d_att))
    alpha = F.softmax(a.masked_fill(mask == 0, MASK_FILL), dim=1)           # -1e9: all-masked row -> uniform
    return torch.bmm(alpha.unsqueeze(1), feature).squeeze(1)                # values are the RAW features


# ------------------------------------------------------------------------------------------
# a3: news-graph context (graphEncoders.py:109-114), eval mode (dropout__ is identity)
# ------------------------------------------------------------------------------------------
def _keep(x: torch.Tensor, drop, frac: float) -> torch.Tensor:
    """Train mode: ``drop(x, frac)`` is the caller's dropout at rate ``frac * dropout_rate`` (graphEncoders.py:22-24: ``dropout`` and
    ``dropout_`` are the full rate, ``dropout__`` half of it); ``None`` = eval mode (identity).  The hook is called once per
    dropout site, in the reference's order of evaluation."""
    return x if drop is None else drop(x, frac)


def news_graph_context(p: Params, X: torch.Tensor, node_mask: torch.Tensor, drop=None) -> torch.Tensor:
    local = X[:, 0]
    glob = scaled_dot_attention(p, "candidate_attention", X, local, node_mask)
    gate = torch.sigmoid(_keep(_linear(torch.cat([local, glob], dim=1), p, "news_graph_W"), drop, 0.5))   # dropout__ (:111)
    return gate * local + (1 - gate) * glob


# ------------------------------------------------------------------------------------------
# a4 (+a7): user-graph context (graphEncoders.py:123-134), eval mode
# ------------------------------------------------------------------------------------------
def topic_pooling(p: Params, Xu: torch.Tensor, cat_idx: torch.Tensor, c_n: torch.Tensor,
                  H: int) -> torch.Tensor:
    """Topic-level attention: returns the pooled [B, C+1, d] before featureAffine (:124-130)."""
    d = Xu.shape[2]
    C1 = p["topic_node_embedding"].shape[0] + 1
    hist = Xu[:, :H]
    k = _linear(hist, p, "user_news_K")
    q = _linear(c_n, p, "user_news_Q").unsqueeze(2)
    a = torch.bmm(k, q).squeeze(2) / math.sqrt(float(d))
    alpha = segment_softmax(a, cat_idx, C1).unsqueeze(2)
    return segment_sum(alpha * hist, cat_idx, C1)


def user_graph_context(p: Params, Xu: torch.Tensor, cat_mask: torch.Tensor, cat_idx: torch.Tensor,
                       c_n: torch.Tensor, H: int, drop=None) -> torch.Tensor:
    topics = topic_pooling(p, Xu, cat_idx, c_n, H)
    topics = _keep(F.relu(_linear(topics, p, "featureAffine")) + topics, drop, 1.0)   # :131 (dropout)
    return scaled_dot_attention(p, "userAttention", topics, c_n, cat_mask)    # :133


# ------------------------------------------------------------------------------------------
# a1 / a2: Eq. 8 dual-interaction GAT layer (graphEncoders.py:143-154 / :163-174), eval mode
# ------------------------------------------------------------------------------------------
def cross_graph_attention(p: Params, graph: str, layer: int, X: torch.Tensor, adj: torch.Tensor,
                          ctx: torch.Tensor, return_alpha: bool = False, drop=None):
    """Unfused, exactly as the reference evaluates it: materialises [B,n,n,d].

    K1 (ffn1) is indexed by the neighbour j, K2 (ffn2) by the centre i, softmax over j (E5);
    ``K3 + K1 + K2`` is evaluated left to right.
    """
    B, n, d = X.shape
    pre = f"{graph}_graph_attention_"
    X = _keep(X, drop, 0.5)                                                   # dropout__ (:145 / :165): the residual uses the dropped input
    h = _linear(X, p, f"{pre}W.{layer}")
    K1 = _linear(X, p, f"{pre}ffn1.{layer}").unsqueeze(1)                     # [B,1,n,d]
    K2 = _linear(X, p, f"{pre}ffn2.{layer}").unsqueeze(2)                     # [B,n,1,d]
    K3 = _linear(ctx, p, f"{pre}ffn3.{layer}").view(B, 1, 1, d)
    s = F.linear(F.relu(K3 + K1 + K2), p[f"{pre}a.{layer}.weight"]).squeeze(3)  # [B,n,n]
    e = F.leaky_relu(s, 0.2)
    alpha = F.softmax(e.masked_fill(adj == 0, MASK_FILL), dim=2)
    out = F.relu(torch.bmm(_keep(alpha, drop, 1.0), h)) + X                   # dropout_ on alpha (:152 / :172)
    return (out, alpha) if return_alpha else out


# ------------------------------------------------------------------------------------------
# a5: orchestration (graphEncoders.py:177-198), eval mode
# ------------------------------------------------------------------------------------------
def user_nodes(p: Params, user_news_embedding: torch.Tensor, drop=None) -> torch.Tensor:
    B = user_news_embedding.shape[0]
    topic = _keep(p["topic_node_embedding"].unsqueeze(0).expand(B, -1, -1), drop, 0.5)    # dropout__ (:179)
    return torch.cat([user_news_embedding, topic], dim=1)                     # [history | topics] (E4)


def encoder_inference(p: Params, depth: int, Xn, An, Mn, user_news_embedding, Au, cat_mask, cat_idx,
                      c_n, trace: "List | None" = None, drop=None, Xu=None):
    H = user_news_embedding.shape[1]
    if Xu is None:
        Xu = user_nodes(p, user_news_embedding)
    c_u = user_graph_context(p, Xu, cat_mask, cat_idx, c_n, H, drop)
    for i in range(depth):
        Xn_next = cross_graph_attention(p, "news", i, Xn, An, c_u, drop=drop)
        Xu_next = cross_graph_attention(p, "user", i, Xu, Au, c_n, drop=drop)
        Xn, Xu = Xn_next, Xu_next
        c_n = c_n + news_graph_context(p, Xn, Mn, drop)
        c_u = c_u + user_graph_context(p, Xu, cat_mask, cat_idx, c_n, H, drop)     # uses the UPDATED c_n
        if trace is not None:
            trace.append((Xn, Xu, c_n, c_u))
    return c_n, c_u


def encoder_forward(p: Params, depth: int, Xn, An, Mn, user_news_embedding, Au, cat_mask, cat_idx,
                    trace: "List | None" = None, drop=None):
    """``drop`` (train mode): the dropout hook of ``_keep``, called in the order of graphEncoders.py:177-187 — the topic nodes (:179),
    then per function as the reference evaluates it."""
    Xu = user_nodes(p, user_news_embedding, drop)                             # :179
    c_n0 = news_graph_context(p, Xn, Mn, drop)                                # :180
    return encoder_inference(p, depth, Xn, An, Mn, user_news_embedding, Au, cat_mask, cat_idx, c_n0, trace, drop, Xu)


# ------------------------------------------------------------------------------------------
# train mode: the dropout generator of the HIP path, restated (digat_amd/csrc/digat_train.inc, dropout_fwd_kernel)
# ------------------------------------------------------------------------------------------
def _hash32(x: np.ndarray) -> np.ndarray:
    m = np.uint64(0xFFFFFFFF)
    x = x & m
    x = x ^ (x >> np.uint64(16)); x = (x * np.uint64(0x7FEB352D)) & m
    x = x ^ (x >> np.uint64(15)); x = (x * np.uint64(0x846CA68B)) & m
    return x ^ (x >> np.uint64(16))


def hash_dropout_keep(n: int, p: float, seed: int) -> np.ndarray:
    """keep[e] of a dropout over n < 2^32 elements in flat (row-major) order: a counter-based hash of (seed, e) against
    floor(float32(p) * 2^32).  Dropout masks are the one thing a reimplementation cannot share with the reference (torch's Philox
    stream); restating the HIP path's generator lets the REFERENCE's autograd run a training step under the very masks the kernels
    draw (oracle/make_golden.py train_step_dropout)."""
    assert 0 <= n < 2 ** 32
    e = np.arange(n, dtype=np.uint64)
    inner = _hash32(np.array([seed], dtype=np.uint64))[0]
    h = _hash32(((e * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)) + inner)
    thr = np.uint64(int(float(np.float32(p)) * 4294967296.0))
    return h >= thr


def hash_dropout(x: torch.Tensor, p: float, seed: int) -> torch.Tensor:
    """x * keep / (1 - p) in fp32, as the kernels evaluate it (scale = 1 / (1 - float32(p)) rounded to fp32)."""
    keep = torch.from_numpy(hash_dropout_keep(x.numel(), p, seed)).view(x.shape)
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    return torch.where(keep, x * scale, torch.zeros_like(x))


class SeedTape:
    """Dropout hook for ``encoder_forward(drop=...)``: site k of a forward pass uses seed ``first + k``."""

    def __init__(self, rate: float, first: int = 1001):
        self.rate, self.next = float(rate), int(first)

    def __call__(self, x: torch.Tensor, frac: float) -> torch.Tensor:
        seed, self.next = self.next, self.next + 1
        return hash_dropout(x.contiguous(), self.rate * frac, seed)


# ------------------------------------------------------------------------------------------
# H1: Model.inference / Model.forward glue (model.py:54-90), graph-encoder side only
# ------------------------------------------------------------------------------------------
def row_logits(p: Params, depth: int, user_news_embedding, Au, cat_mask, cat_idx, Xn, An, Mn, c_n0):
    """model.py:87-90: one logit per (impression, candidate) row."""
    news_rep, user_rep = encoder_inference(p, depth, Xn, An, Mn, user_news_embedding, Au, cat_mask,
                                           cat_idx, c_n0)
    return (user_rep * news_rep).sum(dim=1)


def training_logits(p: Params, depth: int, user_news_embedding, Au, cat_mask, cat_idx, Xn, An, Mn, drop=None):
    """model.py:54-77 from the encoder inputs on: [B, 1+neg] candidates per user, user tensors
    expanded per candidate, dot-product logits.  Xn [B,K,N,d], An [B,K,N,N], Mn [B,K,N]."""
    B, K = Xn.shape[:2]

    def expand(t):
        return t.unsqueeze(1).expand(B, K, *t.shape[1:]).reshape(B * K, *t.shape[1:])

    news_rep, user_rep = encoder_forward(
        p, depth, Xn.reshape(B * K, *Xn.shape[2:]), An.reshape(B * K, *An.shape[2:]),
        Mn.reshape(B * K, -1), expand(user_news_embedding), expand(Au), expand(cat_mask), expand(cat_idx), drop=drop)
    return (user_rep.view(B, K, -1) * news_rep.view(B, K, -1)).sum(dim=2)


def training_loss(logits: torch.Tensor) -> torch.Tensor:
    """trainer.py:100 — the clicked candidate is column 0."""
    return (-torch.log_softmax(logits, dim=1).select(1, 0)).mean()


# ------------------------------------------------------------------------------------------
# H2: ranking + metrics (util.py:70-80, evaluate.py:7-89)
# ------------------------------------------------------------------------------------------
def impression_ranks(scores: Sequence[float], row_impression: Sequence[int]) -> List[List[int]]:
    """Per impression: stable descending sort of the scores, 1-based rank of every candidate."""
    groups: List[List[Tuple[float, int]]] = [[] for _ in range(int(row_impression[-1]) + 1)]
    for s, imp in zip(scores, row_impression):
        groups[int(imp)].append((float(s), len(groups[int(imp)])))
    ranks = []
    for g in groups:
        order = sorted(g, key=lambda t: t[0], reverse=True)                   # Python sort is stable
        r = [0] * len(g)
        for place, (_, pos) in enumerate(order):
            r[pos] = place + 1
        ranks.append(r)
    return ranks


def rank_lines(ranks: List[List[int]]) -> List[str]:
    """The rank-file lines ``"<impression id> [r1,r2,...]"`` (util.py:74-80)."""
    return [f"{i + 1} " + str(r).replace(" ", "") for i, r in enumerate(ranks)]


def _dcg(y_true: np.ndarray, y_score: np.ndarray, k: int) -> float:
    order = np.argsort(y_score)[::-1]
    gains = 2 ** np.take(y_true, order[:k]) - 1
    return float(np.sum(gains / np.log2(np.arange(len(gains)) + 2)))


def _auc(y_true: np.ndarray, y_score: np.ndarray) -> float:
    """Mann-Whitney AUC with average ranks for ties (= sklearn.roc_auc_score on binary labels)."""
    order = np.argsort(y_score, kind="mergesort")
    sorted_scores = y_score[order]
    ranks = np.empty(len(y_score), dtype=np.float64)
    i = 0
    while i < len(order):
        j = i
        while j + 1 < len(order) and sorted_scores[j + 1] == sorted_scores[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    pos = y_true > 0
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def ranking_metrics(labels: List[List[int]], ranks: List[List[int]]):
    """AUC, MRR, nDCG@5, nDCG@10 averaged over impressions, on 1/rank scores (evaluate.py:32-89)."""
    aucs, mrrs, n5s, n10s = [], [], [], []
    for y, r in zip(labels, ranks):
        if len(y) == 0:
            continue
        y_true = np.asarray(y, dtype=np.float32)
        y_score = np.asarray([1.0 / v for v in r], dtype=np.float64)
        aucs.append(_auc(y_true, y_score))
        order = np.argsort(y_score)[::-1]
        hit = np.take(y_true, order)
        mrrs.append(float(np.sum(hit / (np.arange(len(hit)) + 1)) / np.sum(y_true)))
        n5s.append(_dcg(y_true, y_score, 5) / _dcg(y_true, y_true, 5))
        n10s.append(_dcg(y_true, y_score, 10) / _dcg(y_true, y_true, 10))
    return float(np.mean(aucs)), float(np.mean(mrrs)), float(np.mean(n5s)), float(np.mean(n10s))


# ------------------------------------------------------------------------------------------
# the five ablation encoders (graphEncoders.py:201-842), eval mode
# ------------------------------------------------------------------------------------------
ABLATIONS = ("wo_SA", "Seq_SA", "wo_interaction", "News_graph_wo_inter", "User_graph_wo_inter")


def gat_layer(p: Params, graph: str, layer: int, X: torch.Tensor, adj: torch.Tensor) -> torch.Tensor:
    """Vanilla GAT update (graphEncoders.py:493-503 / :510-519): e_ij = leaky_relu(a1.h_j + a2.h_i)."""
    B, n, _ = X.shape
    pre = f"{graph}_graph_attention_"
    h = _linear(X, p, f"{pre}W.{layer}")
    a1 = F.linear(h, p[f"{pre}a1.{layer}.weight"]).view(B, 1, n)               # over the neighbour j
    a2 = F.linear(h, p[f"{pre}a2.{layer}.weight"])                             # [B,n,1]: over the centre i
    e = F.leaky_relu(a1 + a2, 0.2)
    alpha = F.softmax(e.masked_fill(adj == 0, MASK_FILL), dim=2)
    return F.relu(torch.bmm(alpha, h)) + X


def ablation_encode(name: str, p: Params, depth: int, Xn, An, Mn, user_news_embedding, Au, cat_mask, cat_idx, c_n=None):
    """``forward`` (c_n None: computed here where the class computes it) / ``inference`` (c_n given) of an ablation
    encoder, eval mode.  Returns (news context, user context)."""
    assert name in ABLATIONS
    H = user_news_embedding.shape[1]
    Xu = user_nodes(p, user_news_embedding)
    if name == "wo_SA":                                   # :276-293 — no news graph; context = the candidate itself
        c = Xn[:, 0]
        for i in range(depth):
            Xu = cross_graph_attention(p, "user", i, Xu, Au, c)
        return c, user_graph_context(p, Xu, cat_mask, cat_idx, c, H)
    if c_n is None:
        c_n = news_graph_context(p, Xn, Mn)               # compute_news_{graph,sequence}_context
    c_u = user_graph_context(p, Xu, cat_mask, cat_idx, c_n, H)
    for i in range(depth):
        if name == "Seq_SA":                              # :390-408 — the neighbourhood is a sequence: pooled once, never updated
            Xu = cross_graph_attention(p, "user", i, Xu, Au, c_n)
            c_u = c_u + user_graph_context(p, Xu, cat_mask, cat_idx, c_n, H)
            continue
        if name == "wo_interaction":                      # :523-549
            Xn_next, Xu_next = gat_layer(p, "news", i, Xn, An), gat_layer(p, "user", i, Xu, Au)
        elif name == "News_graph_wo_inter":               # :672-696
            Xn_next, Xu_next = gat_layer(p, "news", i, Xn, An), cross_graph_attention(p, "user", i, Xu, Au, c_n)
        else:                                             # User_graph_wo_inter :819-842
            Xn_next, Xu_next = cross_graph_attention(p, "news", i, Xn, An, c_u), gat_layer(p, "user", i, Xu, Au)
        Xn, Xu = Xn_next, Xu_next
        c_n = c_n + news_graph_context(p, Xn, Mn)
        c_u = c_u + user_graph_context(p, Xu, cat_mask, cat_idx, c_n, H)
    return c_n, c_u


# ------------------------------------------------------------------------------------------
# convenience for tests / bench: numpy in, numpy out
# ------------------------------------------------------------------------------------------
def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


def batch_tensors(batch: Dict[str, np.ndarray]):
    """The 7 encoder inputs of synthetic.make_encoder_batch as torch tensors, reference order."""
    keys = ("news_graph_embeddings", "news_graph", "news_graph_mask", "user_news_embedding",
            "user_graph", "user_category_mask", "user_category_indices")
    return tuple(_t(batch[k]) for k in keys)
