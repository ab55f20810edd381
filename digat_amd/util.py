"""Dev/test scoring driver: the counterpart of the reference's ``util.compute_scores`` (util.py:10-85).

Same flow — news-representation cache -> SA gather -> c_n0 precompute -> batched
``Model.inference`` -> per-impression stable ranking -> AUC/MRR/nDCG — re-laid for one MI355X:

* every corpus table lives in HBM for the whole run (288 GB: MIND-small's user graphs are 328 MB,
  the SA cache 1 GB); a batch is assembled by on-device ``index_select`` from (impression, candidate)
  ids instead of 32 DataLoader worker processes building ``[1024,67,67]`` bool arrays item by item
  (util.py:51-52, MIND_dataset.py:97-102);
* rows shard across ranks in contiguous impression-aligned blocks (rows are independent, the ranking
  is per impression), one ``all_gather`` of fp32 scores at the end — no collective on the data path.

The graph encoder is the HIP plugin (``digat_amd.graphEncoders.DIGAT``); nothing here falls back to CPU.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from . import evaluate


NEWS_TABLE_MAX_BYTES = 48 << 30     # layer-0 [h|P|Q] tables of news graphs of more than 16 nodes are kept up to this size (288 GB of HBM per GPU)
NEWS_TABLE_FREE_FRACTION = 0.5      # ... and never beyond this share of the memory that is free when the table is built
SPARSE_ENTRIES_PER_NODE = 20        # the library's own threshold for DIGAT_XATTN_AUTO (digat_kernels.hip); measured at 15.7 entries per node
                                    # (heavy histories, bench.py): sparse 5.34 vs dense 6.11 ms per 4096-row step; break-even extrapolates to ~21


@dataclass
class DeviceCorpus:
    """Device-resident corpus tables with the reference's array names (MIND_corpus.py:189-298)."""
    news_embedding: torch.Tensor         # [news_num, d]   cached_news_representations (util.py:24-33)
    news_node_ID: torch.Tensor           # [news_num, N]   int64
    news_graph: torch.Tensor             # [news_num, N, N] bool
    news_graph_mask: torch.Tensor        # [news_num, N]   bool
    history: torch.Tensor                # [I, H] int64
    user_graph: torch.Tensor             # [I, U, U] bool
    user_category_mask: torch.Tensor     # [I, C+1] bool
    user_category_indices: torch.Tensor  # [I, H] int64
    row_impression: torch.Tensor         # [R] int64
    row_candidate: torch.Tensor          # [R] int64
    SA_news_representations: Optional[torch.Tensor] = None   # [news_num, N, d] (util.py:36)
    c_n0: Optional[torch.Tensor] = None                      # [news_num, d]    (util.py:37-44)
    news_hpq0: Optional[torch.Tensor] = None                 # [3, news_num, N, d]: layer 0's [h|P|Q] of every news graph
    user_hpq0: Optional[torch.Tensor] = None                 # [3, news_num, d]: layer 0's user-graph [h|P|Q] of every news as a history node
    topic_hpq0: Optional[torch.Tensor] = None                # [3, C, d]: ... of the topic nodes
    ctxq0: Optional[torch.Tensor] = None                     # [3, news_num, d]: topic query | user query | layer-0 user K3 of every c_n0
    weights_key: Optional[tuple] = None                      # the weight version the five caches above were computed from
    title_text: Optional[torch.Tensor] = None                # [news_num, Lw] int32 token ids (MIND_corpus.py: news_title_text)
    title_mask: Optional[torch.Tensor] = None                # [news_num, Lw] bool                       (news_title_mask)
    news_key: Optional[tuple] = None                         # the news-encoder weight version news_embedding was computed from
    xattn_hint: Optional[dict] = None                        # THIS corpus's sparse / dense choice per graph (prepare_news_side); applied to the
                                                             # encoder whenever this corpus is scored (two corpora may share one encoder)
    range_overflow_at_prepare: bool = False                  # an fp16x3 GEMM left the format's range while the per-news tables were built

    @classmethod
    def from_numpy(cls, corpus, device) -> "DeviceCorpus":
        def t(a, dtype=None):
            x = torch.from_numpy(np.ascontiguousarray(a))
            return (x.to(dtype) if dtype is not None else x).to(device)
        return cls(t(corpus.news_embedding), t(corpus.news_node_ID, torch.int64), t(corpus.news_graph),
                   t(corpus.news_graph_mask), t(corpus.history, torch.int64), t(corpus.user_graph),
                   t(corpus.user_category_mask), t(corpus.user_category_indices, torch.int64),
                   t(corpus.row_impression, torch.int64), t(corpus.row_candidate, torch.int64))

    @property
    def rows(self) -> int:
        return int(self.row_impression.shape[0])


def shard_rows(row_impression: np.ndarray, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, impression-aligned block [start, end) of rows for ``rank``.

    Boundaries are the impression starts nearest to the even split, so every impression is scored by
    exactly one rank and the blocks concatenate back in row order."""
    imp = np.asarray(row_impression)
    R = len(imp)
    if world_size <= 1:
        return 0, R
    starts = np.r_[0, np.flatnonzero(np.diff(imp)) + 1, R]

    def cut(k):
        if k <= 0:
            return 0
        if k >= world_size:
            return R
        target = (R * k) // world_size
        return int(starts[np.searchsorted(starts, target, side="left")])
    return cut(rank), cut(rank + 1)


def cache_news_representations(news_encoder, title_text: torch.Tensor, title_mask: torch.Tensor, batch_size: int) -> torch.Tensor:
    """util.py:24-33: the representation of every news, computed once per dev/test run in batches.
    ``title_text`` / ``title_mask`` [news_num, max_title_length] on the GPU.  With ``newsEncoders.MSA`` in eval mode this
    runs on the HIP kernels (``digat_msa_fwd``: embedding lookup folded into the bf16x6 projection GEMM, attention on the
    matrix cores)."""
    news_num = title_text.shape[0]
    out = torch.empty((news_num, news_encoder.news_embedding_dim), dtype=torch.float32, device=title_text.device)
    if hasattr(news_encoder, "eval"):
        news_encoder.eval()
    with torch.no_grad():
        for s in range(0, news_num, batch_size):
            e = min(s + batch_size, news_num)
            out[s:e] = news_encoder(title_text[s:e].unsqueeze(1), title_mask[s:e].unsqueeze(1)).squeeze(1)   # :30-32
    return out


def prepare_news_side(encoder, dc: DeviceCorpus, batch_size: int) -> None:
    """util.py:34-44: gather the SA neighbourhood embeddings and precompute c_n0 for every news."""
    news_num, N = dc.news_node_ID.shape
    d = dc.news_embedding.shape[1]
    # Eq. 8 of the user graph: the corpus is known here, so the sparse / dense choice the library would otherwise make on
    # the device per batch (both variants launched, one returning at once) is made once, on the host, from the mean
    # number of adjacency entries per node (MIND user graphs: ~4 of 67).  An explicit "dense" / "sparse" setting wins.
    if hasattr(encoder, "resolved_xattn_mode"):
        hint = {}
        if dc.user_graph.numel() > 0:
            U = dc.user_graph.shape[1]
            per_node = float(dc.user_graph.sum(dtype=torch.float64) / (dc.user_graph.shape[0] * U))
            hint["user"] = "sparse" if per_node <= SPARSE_ENTRIES_PER_NODE else "dense"
        if N > 16 and dc.news_graph.numel() > 0:
            per_node = float(dc.news_graph.sum(dtype=torch.float64) / (news_num * N))
            hint["news"] = "sparse" if per_node <= SPARSE_ENTRIES_PER_NODE else "dense"
        encoder.corpus_xattn_hint = hint          # in force while the encoder's own setting is "auto"
        dc.xattn_hint = dict(hint)                # ... and re-applied whenever THIS corpus is scored (apply_corpus_hint)
    if hasattr(encoder, "corpus_activation_max") and dc.news_embedding.numel() > 0:
        # the range of the node features the projections will see ("auto" projection format: graphEncoders.resolved_projection_mode)
        encoder.corpus_activation_max = float(dc.news_embedding.abs().max())
    dc.SA_news_representations = dc.news_embedding.index_select(0, dc.news_node_ID.flatten()).view(news_num, N, d)
    c_n0 = torch.empty((news_num, d), dtype=torch.float32, device=dc.news_embedding.device)
    with torch.no_grad():
        for s in range(0, news_num, batch_size):
            e = min(s + batch_size, news_num)
            c_n0[s:e] = encoder.compute_news_graph_context(dc.SA_news_representations[s:e], dc.news_graph_mask[s:e])
    dc.c_n0 = c_n0
    # ... and, in the same spirit, layer 0's projections of the news graph, which depend on the news alone (small news graphs:
    # the kernel that consumes them adds K3 itself).  [3, news_num, N, d] fp32: 3.1 GB for MIND-small at N = 10.
    # Larger news graphs (N = 26, 65) keep the table too when their Eq. 8 runs on the sparse kernel, which reads the candidates' rows
    # in place (round 4: the layer-0 projection GEMM of B N rows was the largest launch of a MIND-large step); 20 GB at MIND-large
    # scale and at MIND-small stress scale, of 288 — bounded by NEWS_TABLE_MAX_BYTES.
    dc.news_hpq0 = None
    table_bytes = 3 * news_num * N * d * 4
    dev = dc.news_embedding.device
    budget = NEWS_TABLE_MAX_BYTES
    if dev.type == "cuda":          # a smaller-HBM part, or a process that also holds training state: stay within what is free
        budget = min(budget, int(NEWS_TABLE_FREE_FRACTION * torch.cuda.mem_get_info(dev)[0]))
    # the table holds fp32 P, Q: under the reduced-precision storage modes ("pq-bf16", "pq-fp8": P', Q of news graphs of more than
    # 16 nodes leave the GEMM epilogue in bf16 / e4m3 at EVERY layer) the in-batch projection stays, so that the configuration
    # measured is the one documented
    pq_low = getattr(encoder, "projection_mode", "") in ("pq-bf16", "pq-bf16-x1", "pq-fp8")
    big_ok = (N > 16 and table_bytes <= budget and hasattr(encoder, "resolved_xattn_mode") and not pq_low
              and encoder.resolved_xattn_mode("news") == "sparse" and news_num < 2 ** 31)
    if (N <= 16 or big_ok) and d % 4 == 0 and d <= 1024 and hasattr(encoder, "project_news_layer0") and getattr(encoder, "graph_depth", 0) > 0:
        try:
            table = torch.empty((3, news_num, N, d), dtype=torch.float32, device=dev)
            chunk = max(batch_size, 4096)
            with torch.no_grad():
                for s in range(0, news_num, chunk):
                    e = min(s + chunk, news_num)
                    table[:, s:e] = encoder.project_news_layer0(dc.SA_news_representations[s:e])
            dc.news_hpq0 = table
        except torch.OutOfMemoryError:          # the in-batch projection needs no table: slower, same bits
            table = None
            dc.news_hpq0 = None
            torch.cuda.empty_cache()
    # the user graph's layer-0 projections are row-wise: per news (a history node is a news) and per topic node
    dc.user_hpq0 = dc.topic_hpq0 = None
    if hasattr(encoder, "project_user_layer0") and getattr(encoder, "graph_depth", 0) > 0 and d % 4 == 0:
        with torch.no_grad():
            dc.user_hpq0 = encoder.project_user_layer0(dc.news_embedding)
            dc.topic_hpq0 = encoder.project_user_layer0(encoder.topic_node_embedding.detach())
    # ... and the first link of the [B,d] chain: the two queries and the user graph's layer-0 K3 are linear maps of c_n0
    dc.ctxq0 = None
    if hasattr(encoder, "news_context_queries") and not getattr(encoder, "training", False) and d % 4 == 0:
        with torch.no_grad():
            dc.ctxq0 = encoder.news_context_queries(dc.c_n0)
    dc.weights_key = weights_key(encoder, dc)
    # the layer-0 projections of news graphs of more than 16 nodes now run HERE, not during scoring: an activation that leaves the
    # fp16x3 range while the tables are built must still reach the scoring run's range check (score_rows clears the flag first)
    dc.range_overflow_at_prepare = bool(hasattr(encoder, "range_overflowed") and encoder.gemm_format() == 1
                                        and encoder.range_overflowed(reset=False))


def apply_corpus_hint(encoder, dc: DeviceCorpus) -> None:
    """Put THIS corpus's sparse / dense choice (made by ``prepare_news_side``) back in force on the encoder: with two corpora on one
    encoder (dev and test) the encoder otherwise keeps the hint of whichever was prepared last, and a kept layer-0 table of news
    graphs of more than 16 nodes would meet an encoder that no longer names the sparse kernel (DIGAT_ERR_ARG) — or the other way."""
    if dc.xattn_hint is not None and hasattr(encoder, "corpus_xattn_hint"):
        encoder.corpus_xattn_hint = dict(dc.xattn_hint)


def weights_key(encoder, dc: DeviceCorpus) -> tuple:
    """What the per-news caches of ``prepare_news_side`` depend on: every encoder parameter's storage and version counter
    (an optimizer step or ``load_state_dict`` bumps them) and the news representations they were computed from."""
    params = tuple((p.data_ptr(), p._version) for p in encoder.parameters()) if hasattr(encoder, "parameters") else ()
    pm = encoder.resolved_projection_mode() if hasattr(encoder, "resolved_projection_mode") else getattr(encoder, "projection_mode", None)
    fmt = encoder.gemm_format() if hasattr(encoder, "gemm_format") else None
    # ... and the kernel that computes the context-query table (the encoder's pass_rows names it: same bits as inside a pass)
    # ... and whether the news graph's Eq. 8 reads the layer-0 table in place (larger news graphs: only the sparse kernel can)
    # — the RESOLVED mode: the explicit setting, or the corpus hint in force (apply_corpus_hint puts the scored corpus's own there)
    news_mode = encoder.resolved_xattn_mode("news") if hasattr(encoder, "resolved_xattn_mode") else None
    return params + (dc.news_embedding.data_ptr(), dc.news_embedding._version, pm, fmt, getattr(encoder, "pass_rows", 0) >= 2048, news_mode)


def gather_batch(dc: DeviceCorpus, start: int, end: int):
    """The 8 inputs of ``Model.inference`` for rows [start, end) (util.py:57-67), all on device."""
    imp = dc.row_impression[start:end]
    cand = dc.row_candidate[start:end]
    H = dc.history.shape[1]
    d = dc.news_embedding.shape[1]
    hist = dc.history.index_select(0, imp)
    user_rep = dc.news_embedding.index_select(0, hist.flatten()).view(end - start, H, d)
    return (user_rep, dc.user_graph.index_select(0, imp), dc.user_category_mask.index_select(0, imp),
            dc.user_category_indices.index_select(0, imp), dc.SA_news_representations.index_select(0, cand),
            dc.news_graph.index_select(0, cand), dc.news_graph_mask.index_select(0, cand),
            dc.c_n0.index_select(0, cand))


def gather_batch_grouped(dc: DeviceCorpus, start: int, end: int, row_impression_host: np.ndarray):
    """Like ``gather_batch`` but the user side is gathered once per impression: returns the 9 inputs of
    ``Model.inference_grouped``.  Rows are impression-major, so the groups of a batch are runs of equal
    impression id; the run boundaries are found on the host copy of the ids (no device sync)."""
    imp_host = row_impression_host[start:end]
    first = np.r_[True, imp_host[1:] != imp_host[:-1]]
    dev = dc.news_embedding.device
    uniq = torch.from_numpy(imp_host[first].astype(np.int64)).to(dev, non_blocking=True)
    row_group = torch.from_numpy((np.cumsum(first) - 1).astype(np.int32)).to(dev, non_blocking=True)
    cand = dc.row_candidate[start:end]
    H, d = dc.history.shape[1], dc.news_embedding.shape[1]
    hist = dc.history.index_select(0, uniq)
    user_rep = dc.news_embedding.index_select(0, hist.flatten()).view(uniq.shape[0], H, d)
    return (user_rep, dc.user_graph.index_select(0, uniq), dc.user_category_mask.index_select(0, uniq),
            dc.user_category_indices.index_select(0, uniq), row_group, dc.SA_news_representations.index_select(0, cand),
            dc.news_graph.index_select(0, cand), dc.news_graph_mask.index_select(0, cand), dc.c_n0.index_select(0, cand))


class GroupedBatchPipeline:
    """Assembles the grouped inputs of batch k+1 on a side stream while batch k is being scored.

    The gathers of a batch are ~10 small kernels (60 us) that would otherwise sit between two encoder calls on the
    same stream.  ``batches`` lists the (start, end) row ranges in the order they will be taken.  The group structure
    of every batch (unique impressions, row -> group index) is computed once on the host and uploaded once; ``nsets`` sets
    of static buffers (user side sized for batch_size // 4 groups, the most the grouped entry takes) take turns, and
    events order gather -> score -> gather on each set.  A set is refilled only once the batch that last used it has been
    scored, so at most ``nsets`` batches are in flight: a driver that alternates over n streams wants nsets = n
    (``score_rows`` passes its lane count; with two sets a third lane only overlapped its prologue)."""

    def __init__(self, dc: DeviceCorpus, batches, row_impression_host: np.ndarray, nsets: int = 2, in_place_tables: bool = True,
                 news_sparse: Optional[bool] = None):
        """``news_sparse``: whether the encoder's Eq. 8 of the NEWS graph resolves to the sparse kernel at call time — the only
        reader that can take layer 0 of news graphs of more than 16 nodes from the per-news table; False leaves the table unused
        (in-batch projection).  None: trust the table's presence (prepare_news_side built it under the sparse mode)."""
        self.dc, self.batches = dc, list(batches)
        self.nsets = nsets = max(2, int(nsets))
        dev = dc.news_embedding.device
        self.dev = dev
        B = max(e - s for s, e in self.batches)
        Gmax = max(1, B // 4)
        H, d = dc.history.shape[1], dc.news_embedding.shape[1]
        U, C1 = dc.user_graph.shape[1], dc.user_category_mask.shape[1]
        N = dc.news_graph.shape[1]

        import os
        in_place_tables = in_place_tables and os.environ.get("DIGAT_IN_PLACE", "1") != "0"          # A/B switch for measurements
        in_place = dc.news_hpq0 is not None and d % 4 == 0 and d <= 1024 and in_place_tables and not (N > 16 and news_sparse is False)
        gathered_tables = dc.news_hpq0 is not None and not in_place and N <= 16        # larger news graphs: in place or not at all

        def bufs():
            return dict(hist=torch.empty((Gmax, H), dtype=torch.int64, device=dev),
                        user_rep=torch.empty((Gmax, H, d), dtype=torch.float32, device=dev),
                        user_graph=torch.empty((Gmax, U, U), dtype=dc.user_graph.dtype, device=dev),
                        cat_mask=torch.empty((Gmax, C1), dtype=dc.user_category_mask.dtype, device=dev),
                        cat_idx=torch.empty((Gmax, H), dtype=torch.int64, device=dev),
                        sa=(torch.empty((B, N, d), dtype=torch.float32, device=dev) if not in_place else None),
                        news_graph=torch.empty((B, N, N), dtype=dc.news_graph.dtype, device=dev),
                        news_mask=torch.empty((B, N), dtype=dc.news_graph_mask.dtype, device=dev),
                        c_n0=torch.empty((B, d), dtype=torch.float32, device=dev),
                        hpq=(torch.empty((3 * B * N * d,), dtype=torch.float32, device=dev) if gathered_tables else None),
                        hist_hpq=(torch.empty((3 * Gmax * H * d,), dtype=torch.float32, device=dev) if dc.user_hpq0 is not None else None),
                        ctxq=(torch.empty((3 * B * d,), dtype=torch.float32, device=dev) if dc.ctxq0 is not None else None))
        # the news side's layer-0 tables (and the node table behind them) are read IN PLACE through the candidate ids when the
        # encoder can (small news graphs: digat_encoder_fwd_grouped_cached's news_index): no gathered copies of 65 MB per batch
        self.in_place = in_place
        self.sets = [bufs() for _ in range(nsets)]
        uniq_parts, rg_parts, self.uo, self.ro = [], [], [0], [0]
        for s, e in self.batches:
            imp_b = row_impression_host[s:e]
            first = np.r_[True, imp_b[1:] != imp_b[:-1]]
            uniq_parts.append(imp_b[first].astype(np.int64))
            rg_parts.append((np.cumsum(first) - 1).astype(np.int32))
            self.uo.append(self.uo[-1] + int(first.sum()))
            self.ro.append(self.ro[-1] + (e - s))
        self.uniq_all = torch.from_numpy(np.concatenate(uniq_parts)).to(dev)
        self.row_group_all = torch.from_numpy(np.concatenate(rg_parts)).to(dev)
        self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))       # the corpus tables and the index arrays above
        self.ready = [torch.cuda.Event() for _ in range(nsets)]
        self.done = [None] * nsets
        self.meta = [None] * nsets
        self._gather(0)

    def _gather(self, k):
        if k >= len(self.batches):
            return
        s, e = self.batches[k]
        par = k % self.nsets
        G, n = self.uo[k + 1] - self.uo[k], e - s
        if 4 * G > n:                             # too few rows per group for the grouped entry: per-row path
            self.meta[par] = (k, None)
            return
        dc, b = self.dc, self.sets[par]
        uniq = self.uniq_all[self.uo[k]:self.uo[k + 1]]
        if self.done[par] is not None:
            self.stream.wait_event(self.done[par])                    # this set's previous batch has been scored
        with torch.cuda.stream(self.stream):
            # every gather of the batch in one launch (digat_gather_tables): 15 index_select kernels of ~9 us each before —
            # 7 % of the kernel time of a scoring step, on a stream that shares its hardware queue with the encoder's
            from . import _lib
            cand_ptr = dc.row_candidate.data_ptr() + 8 * s
            uniq_ptr = uniq.data_ptr()
            H, d = dc.history.shape[1], dc.news_embedding.shape[1]
            N = dc.news_graph.shape[1]
            hist_tab = dc.history.data_ptr()
            jobs = []

            def job(src, dst, row_bytes, rows, idx, idx2=0, inner=1):
                jobs.append((src, dst, row_bytes, rows, idx, idx2, inner))

            def rowb(t):
                return t[0].numel() * t.element_size()
            job(hist_tab, b["hist"].data_ptr(), rowb(dc.history), G, uniq_ptr)
            job(dc.news_embedding.data_ptr(), b["user_rep"].data_ptr(), d * 4, G * H, uniq_ptr, hist_tab, H)
            job(dc.user_graph.data_ptr(), b["user_graph"].data_ptr(), rowb(dc.user_graph), G, uniq_ptr)
            job(dc.user_category_mask.data_ptr(), b["cat_mask"].data_ptr(), rowb(dc.user_category_mask), G, uniq_ptr)
            job(dc.user_category_indices.data_ptr(), b["cat_idx"].data_ptr(), rowb(dc.user_category_indices), G, uniq_ptr)
            if b["sa"] is not None:
                job(dc.SA_news_representations.data_ptr(), b["sa"].data_ptr(), rowb(dc.SA_news_representations), n, cand_ptr)
            job(dc.news_graph.data_ptr(), b["news_graph"].data_ptr(), rowb(dc.news_graph), n, cand_ptr)
            job(dc.news_graph_mask.data_ptr(), b["news_mask"].data_ptr(), rowb(dc.news_graph_mask), n, cand_ptr)
            job(dc.c_n0.data_ptr(), b["c_n0"].data_ptr(), d * 4, n, cand_ptr)
            if b["hist_hpq"] is not None:
                for t in range(3):             # [3, G*H, d] <- user_hpq0[t][history of the impression]
                    job(dc.user_hpq0[t].data_ptr(), b["hist_hpq"].data_ptr() + 4 * t * G * H * d, d * 4, G * H, uniq_ptr, hist_tab, H)
            if b["ctxq"] is not None:
                for t in range(3):             # [3, n, d] <- ctxq0[t][candidate]
                    job(dc.ctxq0[t].data_ptr(), b["ctxq"].data_ptr() + 4 * t * n * d, d * 4, n, cand_ptr)
            if b["hpq"] is not None:
                for t in range(3):             # contiguous [3, n, N, d] for this batch's n rows
                    job(dc.news_hpq0[t].data_ptr(), b["hpq"].data_ptr() + 4 * t * n * N * d, N * d * 4, n, cand_ptr)
            arr = (_lib.GatherJob * len(jobs))(*[_lib.GatherJob(*j) for j in jobs])
            _lib.check(_lib.lib().digat_gather_tables(arr, len(jobs), _lib.stream_ptr()), "digat_gather_tables")
            self.ready[par].record(self.stream)
        self.meta[par] = (k, (G, n, self.row_group_all[self.ro[k]:self.ro[k + 1]]))

    def take(self, k):
        """The 9 inputs of ``Model.inference_grouped`` for batch k, or None (use ``gather_batch`` + ``inference``)."""
        par = k % self.nsets
        assert self.meta[par] is not None and self.meta[par][0] == k, "batches must be taken in order"
        info = self.meta[par][1]
        if info is None:
            return None
        G, n, row_group = info
        b = self.sets[par]
        torch.cuda.current_stream(self.dev).wait_event(self.ready[par])
        s0, e0 = self.batches[k]
        sa = b["sa"][:n] if b["sa"] is not None else self.dc.SA_news_representations
        out = (b["user_rep"][:G], b["user_graph"][:G], b["cat_mask"][:G], b["cat_idx"][:G], row_group,
               sa, b["news_graph"][:n], b["news_mask"][:n], b["c_n0"][:n])
        N_, d_ = self.dc.news_graph.shape[1], self.dc.news_embedding.shape[1]
        H_ = b["hist"].shape[1]
        news_hpq = b["hpq"][:3 * n * N_ * d_].view(3, n, N_, d_) if b["hpq"] is not None else (self.dc.news_hpq0 if self.in_place else None)
        news_index = self.dc.row_candidate[s0:e0] if self.in_place else None
        hist_hpq = b["hist_hpq"][:3 * G * H_ * d_].view(3, G, H_, d_) if b["hist_hpq"] is not None else None
        ctxq = b["ctxq"][:3 * n * d_].view(3, n, d_) if b["ctxq"] is not None else None
        if news_hpq is not None or hist_hpq is not None or ctxq is not None:
            out = out + (news_hpq, hist_hpq, self.dc.topic_hpq0 if hist_hpq is not None else None, ctxq, news_index)
        return out

    def scored(self, k):
        """Call once batch k's kernels are enqueued: the other buffer set may then be refilled for batch k+1."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self.done[k % self.nsets] = ev
        self._gather(k + 1)

    def drain(self):
        """Make the current stream wait for every batch scored so far and for the gather stream: after this the pipeline (and
        its buffers, which the caching allocator may hand to someone else) can be dropped."""
        cur = torch.cuda.current_stream(self.dev)
        for ev in self.done:
            if ev is not None:
                cur.wait_event(ev)
        cur.wait_stream(self.stream)


_batch_streams = {}


def batch_streams(device, count: int = 3):
    """[current stream, extra streams ...] for scoring consecutive batches on alternating streams: a batch begins with
    small input-only kernels (user-node build, liveness, layer-0 projections of the groups) and ends with the last
    user context on the library's side stream — on one stream the GPU idles through both; with more, batch k+1's opening
    runs under batch k's last layer.  Every stream has its own scratch (``_lib.workspace`` is keyed by stream).
    Measured per 1024-row step (MIND-small shapes): 1 stream 1.48 ms, 2: 1.43, **3: 1.36**, 4: 1.48 — three batches in
    flight fill the phases where a batch has a single latency-bound kernel running (a kernel trace of the 2-stream run:
    no kernel 11 % of the time, one kernel 36 %).  The optimum depends on how the runtime maps the 2 x count + 1 streams to
    its 4 hardware queues: with 5 or more queues (GPU_MAX_HW_QUEUES) every count is slower (2.0-2.2 ms) — more kernels truly
    concurrent contend for the chip; without the library's side streams (DIGAT_SINGLE_STREAM=1) two streams do as well as
    three with them (1.37 ms), one is worse (1.63 vs 1.48)."""
    cur = torch.cuda.current_stream(device)
    extra = _batch_streams.setdefault((device, cur.cuda_stream), [])
    while len(extra) < count - 1:
        extra.append(torch.cuda.Stream(device=device))
    return [cur] + extra[:count - 1]


# Rows one launch set scores.  The reference's dev batch (batch_size * 16 = 1024 rows, main.py:42) is what a 24 GB card's
# [B, n, n, d] activations allow; here rows are independent and nothing of that size exists, so consecutive dev batches are
# scored together: 4 096 rows per launch set fill the chip's 256 CUs better (the [B,d] linears, poolings and the news graph
# are 1-4 waves of workgroups at 1 024 rows).  Measured per 1 024 rows, three launch sets in flight: 1 024: 0.93 ms, 2 048:
# 0.92, 4 096: 0.80, 8 192: 0.80 — scores equal to the last fp32 bits or so (the [B,d] linears run on the tiled kernel from 2 048
# rows per pass up: DIGAT.pass_rows).
LAUNCH_ROWS = 4096


def launch_batches(start: int, end: int, batch_size: int, launch_rows: Optional[int] = None) -> List[Tuple[int, int]]:
    """Row ranges of the launch sets over [start, end): whole multiples of the caller's batch (``launch_rows`` rounded down
    to one, at least one batch)."""
    rows = LAUNCH_ROWS if launch_rows is None else launch_rows
    step = batch_size * max(1, rows // max(1, batch_size))
    return [(s, min(s + step, end)) for s in range(start, end, step)]


def _pass_rows(batch_size: int, launch_rows: Optional[int] = None) -> int:
    """Rows of a full launch set (what ``launch_batches`` steps by)."""
    rows = LAUNCH_ROWS if launch_rows is None else launch_rows
    return batch_size * max(1, rows // max(1, batch_size))


def range_overflow_any_rank(enc, world_size: int = 1, group=None, also: bool = False) -> bool:
    """Has an fp16x3 GEMM of ``enc`` met an activation beyond the format's range since the flag was last cleared — on ANY rank of
    ``group``?  Every rank gets the same answer (MAX all-reduce), so that all of them fall back to bf16x6 together: a per-rank
    decision would all-gather scores of two formats and leave the ranks' caches and weights_key diverged."""
    if not hasattr(enc, "range_overflowed"):
        return False
    hit = bool(enc.range_overflowed()) or bool(also)         # `also`: this rank's latch from prepare_news_side (DeviceCorpus.range_overflow_at_prepare)
    if world_size > 1:
        import torch.distributed as dist
        backend = dist.get_backend(group)
        dev = enc.topic_node_embedding.device if backend == "nccl" else torch.device("cpu")
        t = torch.tensor([1 if hit else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        hit = bool(int(t.item()))
    return hit


def score_rows(model, dc: DeviceCorpus, start: int, end: int, batch_size: int, grouped: bool = True,
               streams: int = 3, in_place_tables: bool = True, launch_rows: Optional[int] = None,
               check_range: bool = True) -> torch.Tensor:
    """Scores of rows [start, end): the hot loop of util.py:51-69.  ``grouped`` passes each impression's user
    tensors once (bit-identical scores, less work in layer 0); it needs ``model.inference_grouped``.  ``streams``:
    consecutive launch sets alternate over this many HIP streams (same kernels, same bits: see ``batch_streams``).
    ``batch_size`` is the reference's dev batch; ``launch_rows`` (default ``LAUNCH_ROWS``) how many rows — whole batches — one
    pass through the encoder takes (``launch_rows=batch_size``: the reference's own chunking).

    ``check_range`` (default): when the projections ran in the range-limited fp16x3 format (what "auto" resolves to once
    ``prepare_news_side`` has seen the corpus), the encoder's range flag is read after the run — one host synchronisation — and a
    run that met an activation at or beyond |x| = 4094 is redone in bf16x6 ("auto"; the encoder stays there) or refused
    (explicit "fp16x3").  ``compute_scores`` passes False and makes the same decision for all ranks together.  Callers that drive
    ``model.inference`` / ``inference_grouped`` themselves after ``prepare_news_side`` (bench.py's timed loop) must read
    ``graph_encoder.range_overflowed()`` themselves."""
    enc0 = getattr(model, "graph_encoder", None)
    if enc0 is not None:
        apply_corpus_hint(enc0, dc)
    watch = check_range and hasattr(enc0, "range_overflowed") and enc0.gemm_format() == 1
    if watch:
        enc0.range_overflowed()            # clear what earlier calls may have left in the flag
    scores = _score_rows_once(model, dc, start, end, batch_size, grouped, streams, in_place_tables, launch_rows)
    # ... or the tables this run read were built from out-of-range activations (the layer-0 GEMMs run in prepare_news_side)
    if watch and (enc0.range_overflowed() or (dc.range_overflow_at_prepare and enc0.gemm_format() == 1)):
        _range_fallback(enc0, dc, batch_size)
        scores = _score_rows_once(model, dc, start, end, batch_size, grouped, streams, in_place_tables, launch_rows)
    return scores


def _range_fallback(enc, dc, batch_size):
    """An fp16x3 GEMM met an activation at or beyond the format's range (|x| >= 4094: node features of a deep layer, say): its
    results are degraded or inf.  Under "auto" the encoder moves to the range-free bf16x6 format (and stays there) and the
    per-news tables are rebuilt; an explicit "fp16x3" is the caller's word against the data's — refuse to return such scores."""
    if enc.projection_mode == "fp16x3":
        from ._lib import DigatHipError
        raise DigatHipError("projection_mode='fp16x3': an activation left the format's range (|x| >= 4094); use 'bf16x6' or 'auto'")
    import warnings
    warnings.warn("digat_amd: fp16x3 projections met activations beyond the format's range; re-scoring in bf16x6")
    enc.range_fallback = True
    prepare_news_side(enc, dc, batch_size)


def _score_rows_once(model, dc, start, end, batch_size, grouped, streams, in_place_tables, launch_rows):
    dev = dc.news_embedding.device
    scores = torch.empty(end - start, dtype=torch.float32, device=dev)
    grouped = grouped and hasattr(model, "inference_grouped")
    batches = launch_batches(start, end, batch_size, launch_rows)
    enc = getattr(model, "graph_encoder", None)
    if enc is not None:
        apply_corpus_hint(enc, dc)
    if hasattr(enc, "pass_rows") and batches:
        # the launch-set size names the kernel of the [B,d] linears for the whole run (tail set included); per-news tables made
        # under the other name are rebuilt (they hold the same linears' results)
        enc.pass_rows = _pass_rows(batch_size, launch_rows)
        if dc.weights_key is not None and dc.weights_key != weights_key(enc, dc):
            prepare_news_side(enc, dc, batch_size)
    lanes = batch_streams(dev, max(1, streams))
    # a single lane has nothing else to put under the user graph's kernels: there the library's side stream (news chain) stays on
    # for big passes too (4096 rows, one lane: 3.50 vs 3.64 ms; three lanes: 3.17 without it vs 3.29 with it — the default)
    # (a per-call, per-thread launch option — digat_params.flags — not a process-wide switch)
    if len(lanes) == 1 and hasattr(enc, "launch_options") and enc._launch_option("side_stream") == "auto":
        with enc.launch_options(side_stream="on"):
            return _score_sets(model, dc, start, batches, scores, lanes, grouped, in_place_tables)
    return _score_sets(model, dc, start, batches, scores, lanes, grouped, in_place_tables)


def _score_sets(model, dc, start, batches, scores, lanes, grouped, in_place_tables):
    enc = getattr(model, "graph_encoder", None)
    if not grouped and hasattr(enc, "launch_options") and enc._launch_option("detect_shared_users"):
        # grouped=False means the per-row entry as the reference's driver would call it WITHOUT the encoder's own search for
        # shared users (tests compare the two paths; drivers that want the search call model.inference themselves)
        with enc.launch_options(shared_users=False):
            return _score_sets(model, dc, start, batches, scores, lanes, grouped, in_place_tables)
    with torch.no_grad():
        news_sparse = (enc.resolved_xattn_mode("news") == "sparse") if hasattr(enc, "resolved_xattn_mode") else None
        pipe = (GroupedBatchPipeline(dc, batches, dc.row_impression.cpu().numpy(), nsets=len(lanes), in_place_tables=in_place_tables,
                                     news_sparse=news_sparse)
                if grouped and batches else None)
        # the parameter block (split weights, folded queries) is (re)built on the first lane BEFORE the other lanes are
        # ordered after it: a rebuild inside the loop would run on one lane while the next batch reads it on the other
        if enc is not None and hasattr(enc, "_params") and batches:
            with torch.cuda.stream(lanes[0]):
                enc._params()
        for extra in lanes[1:]:
            extra.wait_stream(lanes[0])                       # the corpus tables, `scores`, the pipeline's index arrays
        for k, (s, e) in enumerate(batches):
            with torch.cuda.stream(lanes[k % len(lanes)]):
                inputs = pipe.take(k) if pipe is not None else None
                if inputs is not None:
                    scores[s - start:e - start] = model.inference_grouped(*inputs)
                else:
                    scores[s - start:e - start] = model.inference(*gather_batch(dc, s, e))
                if pipe is not None:
                    pipe.scored(k)
    for extra in lanes[1:]:
        lanes[0].wait_stream(extra)
    if pipe is not None:
        with torch.cuda.stream(lanes[0]):
            pipe.drain()                  # its buffers go back to the allocator when this function returns
    return scores


def all_gather_scores(local: torch.Tensor, counts: List[int], group=None) -> torch.Tensor:
    """Concatenate every rank's block of scores in rank order (blocks are ragged: pad to the max)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    width = max(counts)
    padded = torch.zeros(width, dtype=local.dtype, device=local.device)
    padded[:local.numel()] = local
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)])


_HEAP_FROZEN = False


def freeze_host_heap(force: bool = False) -> bool:
    """Take the objects alive NOW out of the reach of Python's cyclic garbage collector (``gc.collect(); gc.freeze()``).

    A driver loop of this package enqueues hundreds of launches per pass from Python and allocates a few thousand short-lived
    container objects per step; every few dozen steps that triggers a generation-2 collection, which walks EVERY tracked object of
    the process — with a corpus's Python-side structures resident that is 75-110 ms on the GPU boxes' hosts (round 5: one such pause
    inside a 60-step training region, 7.06 instead of 6.28 ms per step; the device sat idle at its end).  The reference's MIND_Corpus
    keeps dictionaries of millions of entries: the same pauses, longer.  Frozen objects are still freed by reference counting;
    they are just never walked again — which also means cyclic garbage that forms later AMONG them is never reclaimed: a
    process-global side effect the embedding application may not want.  So (round 6): at most ONCE per process however often the
    library's entry points are called (``compute_scores`` runs every dev epoch; ``force=True`` freezes again, for a driver that has
    just built another corpus), and not at all under ``DIGAT_FREEZE_HEAP=0``.  Returns whether this call froze anything."""
    global _HEAP_FROZEN
    if os.environ.get("DIGAT_FREEZE_HEAP", "1") == "0" or (_HEAP_FROZEN and not force):
        return False
    import gc
    gc.collect()
    gc.freeze()
    _HEAP_FROZEN = True
    return True


def compute_scores(model, dc: DeviceCorpus, batch_size: int, labels: Optional[np.ndarray] = None,
                   result_file: Optional[str] = None, rank: int = 0, world_size: int = 1, group=None,
                   score_fn: Optional[Callable] = None):
    """Score every row, rank candidates per impression, write the rank file, return
    ``(scores [R] cpu, (auc, mrr, ndcg5, ndcg10) or None)``.

    ``model`` needs ``.graph_encoder`` (for ``compute_news_graph_context``) and ``.inference``.
    With ``world_size > 1`` each rank scores its impression-aligned block and the blocks are
    all-gathered; metrics and the rank file are produced on rank 0 (the others return ``None`` metrics).
    ``score_fn(model, dc, start, end, batch_size)`` replaces the scorer (tests)."""
    if hasattr(model, "eval"):
        model.eval()
    freeze_host_heap()
    ne = getattr(model, "news_encoder", None)
    if score_fn is None and dc.title_text is not None and ne is not None and not hasattr(ne, "table"):
        # a text news encoder (MSA): util.py:24-33 re-encodes every news at the start of each dev / test run — here whenever the
        # encoder's weights have moved on since news_embedding was computed (training epochs), not otherwise
        nk = tuple((p.data_ptr(), p._version) for p in ne.parameters())
        if dc.news_key != nk:
            dc.news_embedding = cache_news_representations(ne, dc.title_text, dc.title_mask, max(batch_size, 4096))
            dc.news_key = nk               # a new tensor: weights_key below changes and the per-news caches follow
    if score_fn is None and hasattr(model.graph_encoder, "range_overflowed"):
        model.graph_encoder.range_overflowed()                      # clear what earlier calls may have left in the flag
    if score_fn is None and hasattr(model.graph_encoder, "pass_rows"):
        model.graph_encoder.pass_rows = _pass_rows(batch_size)      # what score_rows will pass per call: names the [B,d] kernel
    if score_fn is None:
        apply_corpus_hint(model.graph_encoder, dc)
    if score_fn is None and (dc.c_n0 is None or dc.weights_key != weights_key(model.graph_encoder, dc)):
        prepare_news_side(model.graph_encoder, dc, batch_size)      # first use, or the weights moved on since (an optimizer
                                                                    # step, load_state_dict): the per-news caches are stale
    row_imp = dc.row_impression.cpu().numpy()
    start, end = shard_rows(row_imp, world_size, rank)
    enc = getattr(model, "graph_encoder", None)
    watch_range = score_fn is None and hasattr(enc, "range_overflowed")
    local = score_fn(model, dc, start, end, batch_size) if score_fn else score_rows(model, dc, start, end, batch_size, check_range=False)
    # the fp16x3 range check of score_rows, decided for ALL ranks together (a rank whose shard overflowed must not fall back alone)
    if watch_range and range_overflow_any_rank(enc, world_size, group, also=dc.range_overflow_at_prepare and enc.gemm_format() == 1):
        _range_fallback(enc, dc, batch_size)
        local = score_rows(model, dc, start, end, batch_size, check_range=False)
    if world_size > 1:
        counts = [shard_rows(row_imp, world_size, r) for r in range(world_size)]
        scores = all_gather_scores(local, [e - s for s, e in counts], group)
    else:
        scores = local
    scores_np = scores.detach().cpu().numpy()
    if rank != 0:
        return scores_np, None
    if scores.is_cuda:          # ranking + metrics on the device (digat_rank_metrics); one wave per impression
        ranks, metrics = evaluate.device_ranks_and_metrics(scores, row_imp, labels)
    else:                       # host tensors only occur in the CPU harness tests (score_fn)
        ranks = evaluate.impression_ranks(scores_np, row_imp)
        metrics = evaluate.scoring(labels, ranks, row_imp) if labels is not None else None
    if result_file is not None:
        with open(result_file, "wb") as f:
            f.write(evaluate.rank_file_bytes(ranks, row_imp))
    return scores_np, metrics
