"""Ranking + metrics of the dev/test driver (the reference's util.py:70-80 and evaluate.py:7-89).

``impression_ranks``: per impression, stable descending sort of the scores -> 1-based rank of every
candidate (ties keep candidate order, as Python's stable ``list.sort(reverse=True)`` does).
``scoring``: AUC / MRR / nDCG@5 / nDCG@10 averaged over impressions on ``1/rank`` scores.
Vectorised numpy; AUC by the rank-sum identity (ranks are distinct, so no tie handling is needed —
identical to ``sklearn.metrics.roc_auc_score`` on these inputs).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def impression_ranks(scores: np.ndarray, row_impression: np.ndarray) -> np.ndarray:
    """ranks[r] = 1-based rank of row r inside its impression (rows are impression-major)."""
    scores = np.asarray(scores, dtype=np.float64)
    imp = np.asarray(row_impression, dtype=np.int64)
    n = len(scores)
    # stable sort by (impression ascending, score descending): lexsort is stable, last key is primary
    order = np.lexsort((-scores, imp))
    starts = np.r_[0, np.flatnonzero(np.diff(imp[order])) + 1]
    first_of_group = np.repeat(starts, np.diff(np.r_[starts, n]))
    ranks = np.empty(n, dtype=np.int64)
    ranks[order] = np.arange(n) - first_of_group + 1
    return ranks


def rank_lines(ranks: np.ndarray, row_impression: np.ndarray) -> List[str]:
    """``"<impression id> [r1,r2,...]"`` per impression (util.py:74-80), ids are 1-based."""
    imp = np.asarray(row_impression, dtype=np.int64)
    lines = []
    bounds = np.r_[0, np.flatnonzero(np.diff(imp)) + 1, len(imp)]
    count = int(imp[-1]) + 1 if len(imp) else 0
    per_imp = {int(imp[s]): ranks[s:e] for s, e in zip(bounds[:-1], bounds[1:])}
    for i in range(count):
        r = per_imp.get(i, np.zeros(0, dtype=np.int64))
        lines.append(f"{i + 1} " + str(r.tolist()).replace(" ", ""))      # the reference's own formatting (util.py:80), at C speed
    return lines


def _dcg(labels_sorted: np.ndarray, k: int) -> float:
    g = labels_sorted[:k]
    return float(np.sum((2.0 ** g - 1.0) / np.log2(np.arange(len(g)) + 2.0)))


def scoring(labels: np.ndarray, ranks: np.ndarray, row_impression: np.ndarray) -> Tuple[float, float, float, float]:
    """(AUC, MRR, nDCG@5, nDCG@10), mean over impressions that have at least one row."""
    labels = np.asarray(labels, dtype=np.float64)
    imp = np.asarray(row_impression, dtype=np.int64)
    bounds = np.r_[0, np.flatnonzero(np.diff(imp)) + 1, len(imp)]
    aucs, mrrs, n5, n10 = [], [], [], []
    for s, e in zip(bounds[:-1], bounds[1:]):
        y, r = labels[s:e], ranks[s:e].astype(np.float64)
        n_pos = float(y.sum())
        n_neg = float(len(y) - n_pos)
        # score = 1/rank is strictly decreasing in rank, so ascending-score rank = len - rank + 1
        asc = len(y) - r + 1.0
        aucs.append((asc[y > 0].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))
        by_rank = y[np.argsort(r, kind="stable")]
        mrrs.append(float(np.sum(by_rank / (np.arange(len(y)) + 1.0)) / n_pos))
        ideal = np.sort(y)[::-1]
        n5.append(_dcg(by_rank, 5) / _dcg(ideal, 5))
        n10.append(_dcg(by_rank, 10) / _dcg(ideal, 10))
    return float(np.mean(aucs)), float(np.mean(mrrs)), float(np.mean(n5)), float(np.mean(n10))


def rank_file_bytes(ranks: np.ndarray, row_impression: np.ndarray) -> bytes:
    """``"\\n".join(rank_lines(ranks, row_impression))`` as bytes, formatted by the library's host-side C routine
    (``digat_format_rank_file``): the Python loop spends 0.3 s of a 3 s MIND-small dev run on 2.7 M integers."""
    import ctypes
    from . import _lib
    imp = np.asarray(row_impression, dtype=np.int64)
    r = np.ascontiguousarray(ranks, dtype=np.int64)
    if imp.size == 0:
        return b""
    count = int(imp[-1]) + 1
    starts = np.ascontiguousarray(np.r_[0, np.cumsum(np.bincount(imp, minlength=count))], dtype=np.int64)
    L = _lib.lib()
    need = L.digat_format_rank_file(r.ctypes.data, starts.ctypes.data, count, None, 0)
    buf = ctypes.create_string_buffer(int(need))
    got = L.digat_format_rank_file(r.ctypes.data, starts.ctypes.data, count, ctypes.addressof(buf), need)
    assert got == need
    return buf.raw[:got]


def device_ranks_and_metrics(scores, row_impression: np.ndarray, labels: np.ndarray = None):
    """Ranks (and, with labels, the four metrics) computed on the GPU by ``digat_rank_metrics``.

    ``scores``: a CUDA float32 tensor [R] in impression-major row order; ``row_impression`` / ``labels``: host
    arrays.  Returns ``(ranks int64 numpy [R], (auc, mrr, ndcg5, ndcg10) or None)`` — the same values as
    ``impression_ranks`` + ``scoring`` (ranks identical; metrics equal to float64 rounding)."""
    import torch
    from . import _lib
    imp = np.asarray(row_impression, dtype=np.int64)
    R = len(imp)
    dev = _lib.require_device(scores)
    if R == 0:
        return np.zeros(0, dtype=np.int64), None
    if np.any(np.diff(imp) < 0):
        raise ValueError("rows must be impression-major")
    starts = np.r_[0, np.flatnonzero(np.diff(imp)) + 1, R].astype(np.int64)
    I = len(starts) - 1
    sc = scores.detach().to(torch.float32).contiguous()
    st = torch.from_numpy(starts).to(dev)
    ranks = torch.empty(R, dtype=torch.int32, device=dev)
    lab = per = mean = None
    if labels is not None:
        lab = torch.from_numpy(np.ascontiguousarray(np.asarray(labels) > 0).view(np.uint8)).to(dev)
        per = torch.empty((I, 4), dtype=torch.float64, device=dev)
        mean = torch.empty(4, dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().digat_rank_metrics(sc.data_ptr(), lab.data_ptr() if lab is not None else None, st.data_ptr(), I,
                                             ranks.data_ptr(), per.data_ptr() if per is not None else None,
                                             mean.data_ptr() if mean is not None else None, _lib.stream_ptr()),
               "digat_rank_metrics")
    metrics = tuple(float(v) for v in mean.cpu().numpy()) if mean is not None else None
    return ranks.cpu().numpy().astype(np.int64), metrics


class AvgMetric:
    """util.py:100-121: the model-selection average."""

    def __init__(self, auc, mrr, ndcg5, ndcg10):
        self.auc, self.mrr, self.ndcg5, self.ndcg10 = auc, mrr, ndcg5, ndcg10
        self.avg = (auc + mrr + (ndcg5 + ndcg10) / 2) / 3

    def __gt__(self, o): return self.avg > o.avg
    def __ge__(self, o): return self.avg >= o.avg
    def __lt__(self, o): return self.avg < o.avg
    def __le__(self, o): return self.avg <= o.avg

    def __str__(self):
        return '%.4f\nAUC = %.4f\nMRR = %.4f\nnDCG@5  = %.4f\nnDCG@10 = %.4f' % (
            self.avg, self.auc, self.mrr, self.ndcg5, self.ndcg10)
