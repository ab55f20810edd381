"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's SAG construction steps.

``generate_cos_similarities`` follows construct_SAG.py:112-162 (per news: ``torch.nn.CosineSimilarity`` of its title /
content embedding against every corpus title / content embedding, their mean, ``torch.topk(k = top_M + 1)`` of each),
without the pickle caching around it; ``generate_news_graph`` follows :449-485 (the breadth-first walk over the
similar-news lists) on integer arrays instead of dictionaries keyed by news-ID strings (``lists_from_dict`` converts).

Pinned by ``tests/golden/sag_*.npz``, minted by ``oracle/make_golden.py`` from the reference's own functions:
``generate_news_graph`` runs unchanged; ``generate_cos_similarities`` asks for ``.cuda()`` tensors, so it is run with its
``torch`` global redirected to the CPU (same ATen ops, same order).  Only ``tests/``, ``__graft_entry__.smoke()`` and
benchmark CPU legs may import this module.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

KINDS = ("title", "content", "title_content", "content_title", "average")
SIMILARITY_THRESHOLD = 0.5          # construct_SAG.py:10


def generate_cos_similarities(title: torch.Tensor, content: torch.Tensor, corpus_title: torch.Tensor,
                              corpus_content: torch.Tensor, top_M: int) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    """{kind: (values [n, k] f32, indices [n, k] int32)} with k = min(top_M, m - 1) + 1 (construct_SAG.py:115,127-162)."""
    n, m = title.size(0), corpus_title.size(0)
    k = min(top_M, m - 1) + 1                                                            # :115
    out = {kind: (torch.zeros(n, k, dtype=torch.float32), torch.zeros(n, k, dtype=torch.int32)) for kind in KINDS}
    cos = torch.nn.CosineSimilarity()                                                    # :141 (dim=1, eps=1e-8)
    pairs = {"title": (title, corpus_title), "content": (content, corpus_content),      # :144, :148
             "title_content": (title, corpus_content), "content_title": (content, corpus_title)}   # :152, :156
    with torch.no_grad():
        for i in range(n):
            row = {kind: cos(q[i:i + 1].expand(m, -1), c) for kind, (q, c) in pairs.items()}
            row["average"] = (row["title"] + row["content"] + row["title_content"] + row["content_title"]) / 4   # :160
            for kind in KINDS:
                v, j = torch.topk(row[kind], k=k)
                out[kind][0][i] = v
                out[kind][1][i] = j.to(torch.int32)
    return out


def lists_from_dict(news_similarity_dict: Dict[str, Sequence], news_ID_dict: Dict[str, int], top_M: int):
    """The similar-news lists of ``aggregate`` (construct_SAG.py:425-446: {news_ID: [[news_ID, cos], ...]}) as arrays indexed
    by news index: sim_index [news_num, top_M] int32, sim_cos [news_num, top_M] f32, sim_len [news_num] int32."""
    news_num = len(news_ID_dict)
    sim_index = np.zeros((news_num, top_M), dtype=np.int32)
    sim_cos = np.zeros((news_num, top_M), dtype=np.float32)
    sim_len = np.zeros(news_num, dtype=np.int32)
    for news_ID, i in news_ID_dict.items():
        entries = news_similarity_dict.get(news_ID, [])
        assert len(entries) <= top_M
        sim_len[i] = len(entries)
        for e, (other, c) in enumerate(entries):
            sim_index[i, e] = news_ID_dict[other]
            sim_cos[i, e] = c
    return sim_index, sim_cos, sim_len


def generate_news_graph(sim_index: np.ndarray, sim_cos: np.ndarray, sim_len: np.ndarray, top_M: int, hop: int,
                        news_node_num: int, threshold: float = SIMILARITY_THRESHOLD):
    """(news_node_ID [num, nn] int32, news_graph [num, nn, nn] bool, news_graph_mask [num, nn] bool), construct_SAG.py:449-485."""
    news_num = len(sim_len)
    node_ID = np.zeros((news_num, news_node_num), dtype=np.int32)
    graph = np.zeros((news_num, news_node_num, news_node_num), dtype=bool)
    mask = np.zeros((news_num, news_node_num), dtype=bool)
    mask[:, 0] = True                                                                    # :455
    for i in range(1, news_num):                                                         # :456
        node_ID[i, 0] = i
        where = {i: 0}
        depth = [0] * news_node_num
        head, rear = 0, 1
        while head < rear:
            if depth[head] != hop:                                                       # :464
                cur = int(node_ID[i, head])
                for e in range(int(sim_len[cur])):
                    if depth[head] > 0 and (float(sim_cos[cur, e]) < threshold or e == top_M - 1):   # :470
                        break
                    nb = int(sim_index[cur, e])
                    pos = where.get(nb)
                    if pos is None:
                        if rear >= news_node_num:
                            raise IndexError("news graph needs more than news_node_num nodes")
                        pos = rear
                        node_ID[i, pos] = nb
                        mask[i, pos] = True
                        where[nb] = pos
                        depth[pos] = depth[head] + 1
                        rear += 1
                    graph[i, head, pos] = True
                    graph[i, pos, head] = True
            head += 1
    return node_ID, graph, mask
