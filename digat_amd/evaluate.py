"""Ranking + metrics of the dev/test driver (the reference's util.py:70-80 and evaluate.py:7-89).

``impression_ranks``: per impression, stable descending sort of the scores -> 1-based rank of every
candidate (ties keep candidate order, as Python's stable ``list.sort(reverse=True)`` does).
``scoring``: AUC / MRR / nDCG@5 / nDCG@10 averaged over impressions on ``1/rank`` scores.
Vectorised numpy; AUC by the rank-sum identity (ranks are distinct, so no tie handling is needed —
identical to ``sklearn.metrics.roc_auc_score`` on these inputs).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

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
        lines.append(f"{i + 1} [" + ",".join(str(int(v)) for v in r) + "]")
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
