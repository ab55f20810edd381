"""Semantic-augmented-graph construction on the GPU: the reference's construct_SAG.py, device steps.

``generate_cos_similarities`` (construct_SAG.py:112-162) and ``generate_news_graph`` (:449-485) keep the reference's
names, argument order and return values; both run on ``cuda:0`` through the C ABI (``digat_sag_cos_topk``,
``digat_sag_news_graph``) and there is no CPU path.  The sentence-transformer embedding step (:13-109), the
JSON/pickle caches between the steps and the news-ID bookkeeping (:237-422) are the reference's storage layer and stay
out of scope; ``similarity_lists`` converts the dictionary ``aggregate`` (:425-446) produces into the arrays the walk
kernel reads.
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
import torch

from . import _lib

similarity_threshold = 0.5          # construct_SAG.py:10
_MAX_K = 32


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.DigatHipError("digat_amd.construct_SAG runs on the GPU only; there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def cos_topk_device(title: torch.Tensor, content: torch.Tensor, corpus_title: torch.Tensor, corpus_content: torch.Tensor,
                    top_M: int):
    """Device tensors in, device tensors out: (values [5, n, k] f32, indices [5, n, k] int32), k = min(top_M, m - 1) + 1;
    kinds in the order ``generate_cos_similarities`` returns them."""
    dev = _lib.require_device(title, content, corpus_title, corpus_content)
    title, content, corpus_title, corpus_content = (_lib.f32(t) for t in (title, content, corpus_title, corpus_content))
    n, dim = title.shape
    m = corpus_title.shape[0]
    if content.shape != (n, dim) or corpus_title.shape != (m, dim) or corpus_content.shape != (m, dim):
        raise ValueError("title/content must be [n, dim] and the corpus embeddings [m, dim]")
    if m < 1:
        raise ValueError("empty corpus")
    k = min(top_M, m - 1) + 1                                            # :115
    if k > _MAX_K:
        raise ValueError(f"top_M + 1 = {k} exceeds the kernel's limit of {_MAX_K}")
    L = _lib.lib()
    values = torch.empty((5, n, k), dtype=torch.float32, device=dev)
    indices = torch.empty((5, n, k), dtype=torch.int32, device=dev)
    ws = _lib.workspace(L.digat_sag_cos_topk_workspace_bytes(n, m, dim), dev, "sag")
    _lib.check(L.digat_sag_cos_topk(title.data_ptr(), content.data_ptr(), n, corpus_title.data_ptr(), corpus_content.data_ptr(), m,
                                    dim, k, values.data_ptr(), indices.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
               "digat_sag_cos_topk")
    return values, indices


def generate_cos_similarities(dataset_type, top_M, category, title_semantic_embeddings, content_semantic_embeddings,
                              corpus_title_semantic_embeddings, corpus_content_semantic_embeddings):
    """construct_SAG.py:112-233 without the pickle cache (``dataset_type`` / ``category`` only name the reference's cache
    files and are ignored): ten CPU tensors, (values [n, k] f32, indices [n, k] int32) for title, content, title-content,
    content-title and average, each row as ``torch.topk`` returns it."""
    dev = _device()
    args = [torch.as_tensor(t, dtype=torch.float32).to(dev) for t in
            (title_semantic_embeddings, content_semantic_embeddings, corpus_title_semantic_embeddings, corpus_content_semantic_embeddings)]
    values, indices = cos_topk_device(*args, top_M=top_M)
    values, indices = values.cpu(), indices.cpu()
    out = []
    for kind in range(5):
        out += [values[kind], indices[kind]]
    return tuple(out)


def similarity_lists(news_similarity_dict: Dict[str, Sequence], news_ID_dict: Dict[str, int], top_M: int):
    """{news_ID: [[news_ID, cos], ...]} (``aggregate``, :425-446) -> (sim_index [num, top_M] int32, sim_cos f32, sim_len int32)
    indexed by ``news_ID_dict`` value."""
    news_num = len(news_ID_dict)
    sim_index = np.zeros((news_num, top_M), dtype=np.int32)
    sim_cos = np.zeros((news_num, top_M), dtype=np.float32)
    sim_len = np.zeros(news_num, dtype=np.int32)
    for news_ID, row in news_ID_dict.items():
        entries = news_similarity_dict[news_ID]
        if len(entries) > top_M:
            raise ValueError(f"{news_ID}: {len(entries)} similar news, more than top_M = {top_M}")
        sim_len[row] = len(entries)
        for e, (other, cos) in enumerate(entries):
            sim_index[row, e] = news_ID_dict[other]
            sim_cos[row, e] = cos
    return sim_index, sim_cos, sim_len


def news_graph_device(sim_index: torch.Tensor, sim_cos: torch.Tensor, sim_len: torch.Tensor, top_M: int, hop: int,
                      news_node_num: int, threshold: float = similarity_threshold):
    """Device arrays in, device tensors out: (news_node_ID int32 [num, nn], news_graph bool [num, nn, nn], mask bool [num, nn])."""
    dev = _lib.require_device(sim_index, sim_cos, sim_len)
    num = sim_len.shape[0]
    if sim_index.dtype != torch.int32 or sim_len.dtype != torch.int32 or sim_cos.dtype != torch.float32:
        raise ValueError("sim_index / sim_len must be int32 and sim_cos float32")
    if tuple(sim_index.shape) != (num, top_M) or tuple(sim_cos.shape) != (num, top_M):
        raise ValueError("sim_index / sim_cos must be [news_num, top_M]")
    sim_index, sim_cos, sim_len = sim_index.contiguous(), sim_cos.contiguous(), sim_len.contiguous()
    node_ID = torch.empty((num, news_node_num), dtype=torch.int32, device=dev)
    graph = torch.empty((num, news_node_num, news_node_num), dtype=torch.uint8, device=dev)
    mask = torch.empty((num, news_node_num), dtype=torch.uint8, device=dev)
    overflow = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().digat_sag_news_graph(sim_index.data_ptr(), sim_cos.data_ptr(), sim_len.data_ptr(), num, top_M, hop,
                                               news_node_num, float(threshold), node_ID.data_ptr(), graph.data_ptr(),
                                               mask.data_ptr(), overflow.data_ptr(), _lib.stream_ptr()),
               "digat_sag_news_graph")
    if int(overflow.item()):
        raise IndexError("a news graph needs more than news_node_num nodes")      # the reference's numpy IndexError at :474
    return node_ID, graph.view(torch.bool), mask.view(torch.bool)


def generate_news_graph(dataset_type, news_similarity_dict, news_ID_dict, top_M, hop, news_node_num):
    """construct_SAG.py:449-485: (news_node_ID int32 [num, nn], news_graph bool [num, nn, nn], news_graph_mask bool [num, nn])
    as numpy arrays (``dataset_type`` is unused there too)."""
    dev = _device()
    arrays = similarity_lists(news_similarity_dict, news_ID_dict, top_M)
    node_ID, graph, mask = news_graph_device(*(torch.from_numpy(a).to(dev) for a in arrays), top_M=top_M, hop=hop,
                                             news_node_num=news_node_num)
    return node_ID.cpu().numpy(), graph.cpu().numpy(), mask.cpu().numpy()
