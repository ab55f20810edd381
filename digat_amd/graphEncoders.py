"""Drop-in ``--graph_encoder=DIGAT`` plugin backed by the gfx950 HIP kernels.

Mirrors the reference's class contract (graphEncoders.py:10-198): constructor
``DIGAT(config, news_embedding_dim)`` reading ``config.{news_graph_size, max_history_num,
category_num, graph_depth, dropout_rate}``; methods ``initialize``, ``forward`` (7 tensors),
``inference`` (8 tensors), ``compute_news_graph_context``, ``compute_user_graph_context``,
``compute_news_graph_embeddings``, ``compute_user_graph_embeddings``; attribute
``max_history_num``; and the same parameter names, so reference checkpoints load and the
trainer's ``'graph_encoder.'`` no-decay match still works.

All arithmetic happens in ``libdigat_hip.so`` through the C ABI (include/digat_hip.h).  There is
no eager / CPU fallback: tensors must be on the GPU and the extension must be built.
"""
from __future__ import annotations

import contextlib
import math
import threading

import torch
import torch.nn as nn

from . import _lib
from .layers import ScaledDotProductAttention

_TLS = threading.local()          # per-thread launch-option overrides, keyed by encoder (DIGAT.launch_options)


class GraphEncoder(nn.Module):
    def __init__(self, config, news_embedding_dim: int):
        super().__init__()
        self.news_graph_size = config.news_graph_size
        self.user_graph_size = config.max_history_num + config.category_num
        self.max_history_num = config.max_history_num
        self.category_num = config.category_num + 1          # +1: the padding bucket (E3)
        self.news_embedding_dim = news_embedding_dim
        self.graph_depth = config.graph_depth
        self.dropout_rate = float(config.dropout_rate)
        self.attention_scalar = math.sqrt(float(self.news_embedding_dim))
        self.topic_node_embedding = nn.Parameter(torch.zeros([config.category_num, self.news_embedding_dim]))

    def initialize(self):
        nn.init.zeros_(self.topic_node_embedding)

    def forward(self, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph,
                user_category_mask, user_category_indices):
        raise Exception('Function forward must be implemented at sub-class')

    def inference(self, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph,
                  user_category_mask, user_category_indices, news_graph_context):
        raise Exception('Function inference must be implemented at sub-class')


class DIGAT(GraphEncoder):
    def __init__(self, config, news_embedding_dim: int):
        super().__init__(config, news_embedding_dim)
        d, L = self.news_embedding_dim, self.graph_depth
        if d % 4 != 0:
            raise ValueError("digat_amd needs news_embedding_dim % 4 == 0 (float4 rows)")
        if L > _lib.DIGAT_MAX_DEPTH or max(self.news_graph_size, self.user_graph_size) > _lib.DIGAT_MAX_NODES:
            raise ValueError("graph_depth <= %d and graph sizes <= %d are supported"
                             % (_lib.DIGAT_MAX_DEPTH, _lib.DIGAT_MAX_NODES))
        # compute_news_graph_context
        self.candidate_attention = ScaledDotProductAttention(d, d, d)
        self.news_graph_W = nn.Linear(d * 2, d, bias=True)
        # compute_user_graph_context
        self.user_news_K = nn.Linear(d, d, bias=False)
        self.user_news_Q = nn.Linear(d, d, bias=True)
        self.featureAffine = nn.Linear(d, d, bias=True)
        self.userAttention = ScaledDotProductAttention(d, d, d)
        # Eq. 8 layers
        for g in ("news", "user"):
            setattr(self, f"{g}_graph_attention_W", nn.ModuleList([nn.Linear(d, d, bias=True) for _ in range(L)]))
            setattr(self, f"{g}_graph_attention_ffn1", nn.ModuleList([nn.Linear(d, d, bias=False) for _ in range(L)]))
            setattr(self, f"{g}_graph_attention_ffn2", nn.ModuleList([nn.Linear(d, d, bias=False) for _ in range(L)]))
            setattr(self, f"{g}_graph_attention_ffn3", nn.ModuleList([nn.Linear(d, d, bias=True) for _ in range(L)]))
            setattr(self, f"{g}_graph_attention_a", nn.ModuleList([nn.Linear(d, 1, bias=False) for _ in range(L)]))
        self._param_block = None
        # node projections: "bf16x6" = fp32-equivalent product on the bf16 matrix cores (default), "bf16x6-pq3" = the same
        # for h, three of the six products for P and Q (they only feed the score: DIGAT_PROJ_PQ_X3), "fp32" = v_mfma_f32_16x16x4_f32;
        # BASELINE configs[4]: "pq-bf16" = pq3 + P', Q of the user graph's layers >= 1 STORED in bf16 (DIGAT_PQ_BF16; the
        # reference's quantised K3 + K1 + K2, README.md:62-66), "pq-bf16-x1" = the same with one bf16 product for P and Q;
        # "pq-fp8" = the same launches with P', Q stored as block-scaled OCP e4m3 (DIGAT_PQ_FP8: one fp32 scale per row and
        # 80-channel strip; the fp8 half of configs[4]);
        # "fp16x3" = every operand as two fp16 pieces, three products (digat_set_gemm_format(1): 0.7x the GEMM time, error at or
        # below an fp32 fma chain's against fp64 for |w| < 63, |x| < 4094 — fp16's range after the format's scaling);
        # "auto" (default) = "fp16x3" when every projected weight is below 32 in magnitude, else "bf16x6" (no range limit)
        self.projection_mode = "auto"
        self._resolved_pm = None
        self._range_flag = None            # device word raised by the fp16x3 GEMMs on out-of-range activations (range_flag())
        self.range_fallback = False        # set by util.compute_scores after such a run: "auto" then resolves to bf16x6
        self.corpus_activation_max = None  # max |news representation| of the corpus, set by util.prepare_news_side
        # rows the driver passes through inference() per call (util.score_rows sets it to its launch-set size, util.LAUNCH_ROWS):
        # it NAMES the kernel of the [B,d] linears (>= 2048: tiled split-operand, below: split-image; DIGAT_PARAMS_BD_TILED) —
        # the row count of a call does not, so a row's bits do not depend on the batch it sits in
        self.pass_rows = 4096
        # Eq. 8 of the user graph: "auto" (the device counts the adjacency entries of the batch and runs the sparse
        # edge-list kernel or the dense tile + MFMA pair), "dense", "sparse" (digat_params.flags, include/digat_hip.h)
        self.user_xattn_mode = "auto"
        # ... and of news graphs of more than 16 nodes: "dense", "sparse" (DIGAT_NEWS_XATTN_SPARSE) or "auto" = dense until
        # util.prepare_news_side has looked at the corpus (there is no device-side decision for this graph)
        self.news_xattn_mode = "auto"
        self.corpus_xattn_hint = {}        # {"user": "sparse" | "dense", "news": ...}: set by util.prepare_news_side
        # per-call launch options (digat_params.flags bits 9-11; results do not depend on them, bit for bit):
        # side_stream "auto" = the news chain of a layer on the library's side stream for passes below 2 048 rows, "on" / "off" =
        # always / never; live_rows False = project, score and write every user-graph node (default: dead nodes are skipped).
        # ``launch_options(...)`` overrides them for the calling THREAD only — nothing here is process-wide.
        self.side_stream = "auto"
        self.live_rows = True
        # compute_user_graph_context of the folded inference path as ONE launch (csrc/digat_ctxfused.inc: topic pooling, featureAffine
        # and the SDPA pooling without T / T' leaving the CU; fp16x3 format, H <= 52, 192 < d <= 448, C + 1 <= 20).  OFF by default:
        # measured at parity with the three launches it replaces (round 6: 0.73 against 0.78 ms of kernel time per 4 096-row pass alone,
        # the same step time with three passes in flight, 3 % slower at 1 024 rows per pass — DESIGN.md section 4).  Read when the
        # parameter block is built (a weight version): set it before the first call.
        self.fused_user_context = False

    # ------------------------------------------------------------------ init (graphEncoders.py:76-101)
    def initialize(self):
        super().initialize()
        relu_gain = nn.init.calculate_gain('relu')
        leaky_gain = nn.init.calculate_gain('leaky_relu', 0.2)
        for g in ("news", "user"):
            for i in range(self.graph_depth):
                nn.init.xavier_uniform_(getattr(self, f"{g}_graph_attention_W")[i].weight)
                nn.init.zeros_(getattr(self, f"{g}_graph_attention_W")[i].bias)
                nn.init.xavier_uniform_(getattr(self, f"{g}_graph_attention_a")[i].weight, gain=leaky_gain)
                for f in ("ffn1", "ffn2", "ffn3"):
                    nn.init.xavier_uniform_(getattr(self, f"{g}_graph_attention_{f}")[i].weight, gain=relu_gain)
                nn.init.zeros_(getattr(self, f"{g}_graph_attention_ffn3")[i].bias)
        self.candidate_attention.initialize()
        nn.init.xavier_uniform_(self.news_graph_W.weight)
        nn.init.zeros_(self.news_graph_W.bias)
        nn.init.xavier_uniform_(self.user_news_K.weight)
        nn.init.xavier_uniform_(self.user_news_Q.weight)
        nn.init.zeros_(self.user_news_Q.bias)
        nn.init.xavier_uniform_(self.featureAffine.weight, gain=relu_gain)
        nn.init.zeros_(self.featureAffine.bias)
        self.userAttention.initialize()

    @contextlib.contextmanager
    def launch_options(self, side_stream=None, live_rows=None, shared_users=None):
        """Override ``side_stream`` / ``live_rows`` for the calls the CURRENT THREAD makes inside the block (util.score_rows turns
        the side stream on for single-lane runs this way): another thread driving the same encoder keeps its own settings."""
        table = _TLS.__dict__.setdefault("opts", {})
        prev = table.get(id(self))
        cur = dict(prev or {})
        if side_stream is not None:
            if side_stream not in ("auto", "on", "off"):
                raise ValueError("side_stream must be 'auto', 'on' or 'off'")
            cur["side_stream"] = side_stream
        if live_rows is not None:
            cur["live_rows"] = bool(live_rows)
        if shared_users is not None:          # inference(): look for runs of identical consecutive user rows (False: the per-row entry as is)
            cur["detect_shared_users"] = bool(shared_users)
        table[id(self)] = cur
        try:
            yield self
        finally:
            if prev is None:
                table.pop(id(self), None)
            else:
                table[id(self)] = prev

    def _launch_option(self, name):
        return _TLS.__dict__.get("opts", {}).get(id(self), {}).get(name, getattr(self, name))

    def resolved_xattn_mode(self, g: str) -> str:
        """The Eq. 8 variant in force for graph ``g``: the explicit setting, or what ``util.prepare_news_side`` found in the
        corpus it was last shown while the setting is "auto" (``corpus_xattn_hint``); "auto" otherwise."""
        mode = getattr(self, f"{g}_xattn_mode", "auto")
        return self.corpus_xattn_hint.get(g, "auto") if mode == "auto" else mode

    # ------------------------------------------------------------------ parameter block for the C ABI
    def _apply(self, fn, *args, **kwargs):
        self._param_block = None          # .cuda() / .to() move the storages
        return super()._apply(fn, *args, **kwargs)

    def _params(self) -> "_lib.Params":
        # the matrix-core operand format ("fp16x3": two fp16 pieces, three products; everything else: three bf16 pieces, six
        # products) is a property of the split images made below and travels in P.flags: nothing process-wide
        pm = self.resolved_projection_mode()
        fmt = self.gemm_format()
        ptrs = tuple(p.data_ptr() for p in self.parameters())
        if self._param_block is not None and self._param_block[0] == ptrs and self._param_block[2] == self._fold_key():
            # the Eq. 8 variant and the launch options only select kernels (flags): they never invalidate the split weights or
            # the folded queries, so changing them (util.prepare_news_side's corpus hint, launch_options) must not rebuild them
            return self._call_block(self._param_block[1])
        for p in self.parameters():
            if p.device.type != "cuda" or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.DigatHipError("DIGAT parameters must be contiguous float32 CUDA tensors "
                                         "(call model.cuda()); there is no CPU path")
        P = _lib.Params()
        P.d, P.depth, P.category_num = self.news_embedding_dim, self.graph_depth, self.category_num - 1
        P.flags = self._flags()
        P.topic_node_embedding = self.topic_node_embedding.data_ptr()
        P.cand_K = self.candidate_attention.K.weight.data_ptr()
        P.cand_Q = self.candidate_attention.Q.weight.data_ptr()
        P.cand_bQ = self.candidate_attention.Q.bias.data_ptr()
        P.news_graph_W = self.news_graph_W.weight.data_ptr()
        P.news_graph_b = self.news_graph_W.bias.data_ptr()
        P.user_news_K = self.user_news_K.weight.data_ptr()
        P.user_news_Q = self.user_news_Q.weight.data_ptr()
        P.user_news_bQ = self.user_news_Q.bias.data_ptr()
        P.featureAffine_W = self.featureAffine.weight.data_ptr()
        P.featureAffine_b = self.featureAffine.bias.data_ptr()
        P.userAtt_K = self.userAttention.K.weight.data_ptr()
        P.userAtt_Q = self.userAttention.Q.weight.data_ptr()
        P.userAtt_bQ = self.userAttention.Q.bias.data_ptr()
        for g, arr in (("news", P.news), ("user", P.user)):
            for i in range(self.graph_depth):
                lp = arr[i]
                W = getattr(self, f"{g}_graph_attention_W")[i]
                lp.W, lp.bW = W.weight.data_ptr(), W.bias.data_ptr()
                lp.F1 = getattr(self, f"{g}_graph_attention_ffn1")[i].weight.data_ptr()
                lp.F2 = getattr(self, f"{g}_graph_attention_ffn2")[i].weight.data_ptr()
                F3 = getattr(self, f"{g}_graph_attention_ffn3")[i]
                lp.F3, lp.b3 = F3.weight.data_ptr(), F3.bias.data_ptr()
                lp.a = getattr(self, f"{g}_graph_attention_a")[i].weight.data_ptr()
        # bf16x6 projections: split [W | ffn1 | ffn2] of every layer into three bf16 planes (once per weight version)
        P._splits = []
        if pm in ("bf16x6", "bf16x6-pq3", "pq-bf16", "pq-bf16-x1", "pq-fp8", "fp16x3") and self.news_embedding_dim % 80 == 0:
            L_ = _lib.lib()
            d = self.news_embedding_dim
            nbytes = L_.digat_split_weights_bytes(3 * d, d)
            for g, arr in (("news", P.news), ("user", P.user)):
                for i in range(self.graph_depth):
                    buf = _lib.split_buffer(nbytes, self.topic_node_embedding.device)
                    _lib.check(L_.digat_split_proj_weights(
                        getattr(self, f"{g}_graph_attention_W")[i].weight.data_ptr(),
                        getattr(self, f"{g}_graph_attention_ffn1")[i].weight.data_ptr(),
                        getattr(self, f"{g}_graph_attention_ffn2")[i].weight.data_ptr(), d, buf.data_ptr(), fmt,
                        _lib.stream_ptr()), "digat_split_proj_weights")
                    arr[i].wsplit = buf.data_ptr()
                    P._splits.append(buf)
            buf = _lib.split_buffer(L_.digat_split_weights_bytes(d, d), self.topic_node_embedding.device)
            _lib.check(L_.digat_split_weights(self.featureAffine.weight.data_ptr(), d, d, buf.data_ptr(), fmt, _lib.stream_ptr()),
                       "digat_split_weights")
            P.featureAffine_wsplit = buf.data_ptr()
            P._splits.append(buf)
            if fmt == _lib.GEMM_F16X3:           # one device word the fp16x3 GEMMs raise when an activation leaves the format's range
                P.range_flag = self.range_flag().data_ptr()
        # inference: fold the key projections into the query weights once per weight version
        P._folds = None
        if not self.training:
            P._folds = self._fold_attention()
            (P.cand_fold_W, P.cand_fold_b, P.user_news_fold_W, P.user_news_fold_b,
             P.userAtt_fold_W, P.userAtt_fold_b) = (t.data_ptr() for t in P._folds)
            if P._splits:
                # the [B,d] linears of the folded path on split images too (gemm_skinny_split_kernel): K3 of the news graph,
                # the candidate query, the gate, and per layer the three matrices applied to the news context
                L_, d, dev = _lib.lib(), self.news_embedding_dim, self.topic_node_embedding.device

                def image(rows, K, fn, *args):
                    buf = _lib.split_buffer(L_.digat_split_weights_bytes(rows, K), dev)
                    _lib.check(fn(*args, buf.data_ptr(), fmt, _lib.stream_ptr()), "split")
                    P._splits.append(buf)
                    return buf.data_ptr()
                for i in range(self.graph_depth):
                    P.news[i].f3_wsplit = image(d, d, L_.digat_split_weights, self.news_graph_attention_ffn3[i].weight.data_ptr(), d, d)
                P.cand_fold_wsplit = image(d, d, L_.digat_split_weights, P.cand_fold_W, d, d)
                P.gate_wsplit = image(d, 2 * d, L_.digat_split_weights, self.news_graph_W.weight.data_ptr(), d, 2 * d)
                for l in range(self.graph_depth + 1):
                    third = self.user_graph_attention_ffn3[l].weight.data_ptr() if l < self.graph_depth else P.userAtt_fold_W
                    P.ctx_wsplit[l] = image(3 * d, d, L_.digat_split_proj_weights, P.user_news_fold_W, P.userAtt_fold_W, third, d)
                if fmt == _lib.GEMM_F16X3 and self.fused_user_context:
                    # compute_user_graph_context as one launch (csrc/digat_ctxfused.inc): featureAffine in the fused kernel's lane order
                    buf = _lib.split_buffer(L_.digat_split_ctx_fused_bytes(d), dev)
                    _lib.check(L_.digat_split_ctx_fused_weights(self.featureAffine.weight.data_ptr(), d, buf.data_ptr(), _lib.stream_ptr()),
                               "digat_split_ctx_fused_weights")
                    P._splits.append(buf)
                    P.featureAffine_fsplit = buf.data_ptr()
        self._param_block = (ptrs, P, self._fold_key())
        return self._call_block(P)

    def _call_block(self, P):
        """The parameter block one call hands to the library: a COPY of the cached block (2.6 KB of pointers) with this call's
        flags — two threads with different launch options never write the same struct."""
        Q = _lib.Params.from_buffer_copy(P)
        Q.flags = self._flags()
        return Q

    def resolved_projection_mode(self) -> str:
        """``projection_mode`` with "auto" resolved from the weights' range (once per weight version; one host sync)."""
        return self._auto_base() if self.projection_mode == "auto" else self.projection_mode

    def gemm_format(self) -> int:
        """The operand format of this encoder's split weight images: fp16x3 when asked for, and under "auto" / "pq-bf16" whenever
        the range conditions of ``_auto_base`` hold; bf16x6 otherwise."""
        pm = self.projection_mode
        if pm == "fp16x3" or (pm in ("auto", "pq-bf16", "pq-fp8") and self._auto_base() == "fp16x3"):
            return _lib.GEMM_F16X3
        return _lib.GEMM_BF16X6

    def _auto_base(self) -> str:
        if self.news_embedding_dim % 80 != 0:
            return "bf16x6"
        ws = [m.weight for g in ("news", "user") for f in ("W", "ffn1", "ffn2") for m in getattr(self, f"{g}_graph_attention_{f}")]
        ws.append(self.featureAffine.weight)
        key = tuple((w.data_ptr(), w._version) for w in ws + [self.topic_node_embedding])
        key += (self.corpus_activation_max, self.range_fallback)
        if self._resolved_pm is None or self._resolved_pm[0] != key:
            amax = self.corpus_activation_max
            # fp16x3 needs a driver that looks at the range flag after the run: util.score_rows and util.compute_scores do (and
            # redo the run in bf16x6); "auto" therefore picks it only once util.prepare_news_side has seen the corpus — plain
            # forward / inference calls get the range-free bf16x6.  A caller that runs prepare_news_side and then drives
            # inference / inference_grouped itself (bench.py's timed loop) must read range_overflowed() after its run.  Then: weights below 32 (the format holds 63), the topic nodes and the
            # corpus's news representations below 256 (the format holds 4094; what the features of layers >= 1 grow to is
            # checked on the device, by the GEMM itself); nan compares false
            ok = amax is not None and not self.range_fallback
            if ok:
                wmax = float(torch.stack([w.detach().abs().max() for w in ws]).max())
                tmax = float(self.topic_node_embedding.detach().abs().max())
                ok = wmax < 32.0 and amax < 256.0 and tmax < 256.0
            self._resolved_pm = (key, "fp16x3" if ok else "bf16x6")
        return self._resolved_pm[1]

    def range_flag(self) -> torch.Tensor:
        """The device word the fp16x3 GEMMs OR 1 into when an activation reaches the format's range (digat_params.range_flag)."""
        dev = self.topic_node_embedding.device
        if self._range_flag is None or self._range_flag.device != dev:
            self._range_flag = torch.zeros(1, dtype=torch.int32, device=dev)
        return self._range_flag

    def range_overflowed(self, reset: bool = True) -> bool:
        """True when an fp16x3 GEMM of this encoder has seen an activation at or beyond the format's range since the last
        reset (one host synchronisation: call it once per scoring run, as util.compute_scores does)."""
        if self._range_flag is None:
            return False
        hit = bool(int(self._range_flag.item()))
        if reset and hit:
            self._range_flag.zero_()
        return hit

    def _flags(self) -> int:
        """digat_params.flags (include/digat_hip.h): Eq. 8 variant of the user graph (bits 0-1), DIGAT_PROJ_PQ_X3 (bit 2),
        DIGAT_NEWS_XATTN_SPARSE (bit 3), ..., DIGAT_PARAMS_BD_TILED (bit 7: ``pass_rows``)."""
        pm = self.projection_mode
        return ({"auto": 0, "dense": 1, "sparse": 2}[self.resolved_xattn_mode("user")]
                | (4 if pm in ("bf16x6-pq3", "pq-bf16", "pq-fp8") else 0)
                | (8 if self.resolved_xattn_mode("news") == "sparse" else 0)
                | (16 if pm in ("pq-bf16", "pq-bf16-x1") else 0)         # DIGAT_PQ_BF16: P', Q of Eq. 8 stored in bf16
                | (32 if pm == "pq-bf16-x1" else 0)                      # DIGAT_PQ_X1: ... and computed with one bf16 product
                | (256 if pm == "pq-fp8" else 0)                         # DIGAT_PQ_FP8: P', Q of Eq. 8 stored as block-scaled e4m3
                | (_lib.PARAMS_GEMM_F16X3 if self.gemm_format() == _lib.GEMM_F16X3 else 0)
                | (_lib.PARAMS_BD_TILED if self.pass_rows >= 2048 else 0)
                | {"auto": 0, "off": _lib.PARAMS_SIDE_STREAM_OFF, "on": _lib.PARAMS_SIDE_STREAM_ON}[self._launch_option("side_stream")]
                | (0 if self._launch_option("live_rows") else _lib.PARAMS_NO_LIVE_ROWS))

    def _fold_sources(self):
        ca, ua = self.candidate_attention, self.userAttention
        return ((ca.K.weight, ca.Q.weight, ca.Q.bias), (self.user_news_K.weight, self.user_news_Q.weight,
                                                        self.user_news_Q.bias), (ua.K.weight, ua.Q.weight, ua.Q.bias))

    def _fold_key(self):
        key = (self.training, self.resolved_projection_mode(), self.gemm_format()) + tuple(t._version for trio in self._fold_sources() for t in trio)
        for g in ("news", "user"):
            for f in ("W", "ffn1", "ffn2"):
                key += tuple(m.weight._version for m in getattr(self, f"{g}_graph_attention_{f}"))
        return key + (self.featureAffine.weight._version, bool(self.fused_user_context))

    def _fold_attention(self):
        """(K x).(Q c + b) = x.(Wf c + bf) with Wf = K^T Q, bf = K^T b, computed by the library itself."""
        L = _lib.lib()
        d = self.news_embedding_dim
        dev = self.topic_node_embedding.device
        nbytes = L.digat_fold_workspace_bytes(d)
        ws = _lib.workspace(nbytes, dev, "fold")
        out = []
        for K, Q, b in self._fold_sources():
            Wf = torch.empty((d, d), dtype=torch.float32, device=dev)
            bf = torch.empty((d,), dtype=torch.float32, device=dev)
            _lib.check(L.digat_fold_attention(K.data_ptr(), Q.data_ptr(), b.data_ptr(), Wf.data_ptr(), bf.data_ptr(), d,
                                              ws.data_ptr(), nbytes, _lib.stream_ptr()), "digat_fold_attention")
            out += [Wf, bf]
        return out

    def _eval_only(self, what: str):
        if self.training and self.dropout_rate > 0 and torch.is_grad_enabled():
            raise NotImplementedError(
                f"{what}: the HIP path implements eval-mode semantics (dropout = identity); "
                "call model.eval() / torch.no_grad(), or train through digat_amd.training")

    # ------------------------------------------------------------------ a3 (graphEncoders.py:109-114)
    def compute_news_graph_context(self, news_graph_embeddings, news_graph_mask):
        X = _lib.f32(news_graph_embeddings)
        dev = _lib.require_device(X, news_graph_mask)
        B, N, d = X.shape
        mask = _lib.as_bytes(news_graph_mask)
        out = torch.empty((B, d), dtype=torch.float32, device=dev)
        if B == 0:
            return out
        L = _lib.lib()
        nbytes = L.digat_news_ctx_workspace_bytes(B, N, d)
        ws = _lib.workspace(nbytes, dev, "ctx")
        ca, g = self.candidate_attention, self.news_graph_W
        _lib.check(L.digat_news_ctx_fwd(X.data_ptr(), mask.data_ptr(), ca.K.weight.data_ptr(), ca.Q.weight.data_ptr(),
                                        ca.Q.bias.data_ptr(), g.weight.data_ptr(), g.bias.data_ptr(), None,
                                        out.data_ptr(), B, N, d, ws.data_ptr(), nbytes, _lib.stream_ptr()),
                   "digat_news_ctx_fwd")
        return out

    # ------------------------------------------------------------------ a4 (graphEncoders.py:123-134)
    def compute_user_graph_context(self, user_graph_embeddings, user_category_mask, user_category_indices,
                                   news_graph_context):
        Xu = _lib.f32(user_graph_embeddings)
        c_n = _lib.f32(news_graph_context)
        dev = _lib.require_device(Xu, user_category_mask, user_category_indices, c_n)
        B, U, d = Xu.shape
        H, C1 = self.max_history_num, self.category_num
        mask = _lib.as_bytes(user_category_mask)
        idx = user_category_indices.to(torch.int64).contiguous()
        out = torch.empty((B, d), dtype=torch.float32, device=dev)
        if B == 0:
            return out
        L = _lib.lib()
        nbytes = L.digat_user_ctx_workspace_bytes(B, U, H, C1, d)
        ws = _lib.workspace(nbytes, dev, "ctx")
        ua = self.userAttention
        _lib.check(L.digat_user_ctx_fwd(Xu.data_ptr(), mask.data_ptr(), idx.data_ptr(), c_n.data_ptr(),
                                        self.user_news_K.weight.data_ptr(), self.user_news_Q.weight.data_ptr(),
                                        self.user_news_Q.bias.data_ptr(), self.featureAffine.weight.data_ptr(),
                                        self.featureAffine.bias.data_ptr(), ua.K.weight.data_ptr(),
                                        ua.Q.weight.data_ptr(), ua.Q.bias.data_ptr(), None, out.data_ptr(),
                                        B, U, H, C1, d, ws.data_ptr(), nbytes, _lib.stream_ptr()),
                   "digat_user_ctx_fwd")
        return out

    # ------------------------------------------------------------------ a1 / a2 (graphEncoders.py:143-174)
    def _xattn(self, g: str, index: int, X, A, ctx, return_alpha: bool = False):
        X, ctx = _lib.f32(X), _lib.f32(ctx)
        dev = _lib.require_device(X, A, ctx)
        B, n, d = X.shape
        adj = _lib.as_bytes(A)
        out = torch.empty_like(X)
        alpha = torch.empty((B, n, n), dtype=torch.float32, device=dev) if return_alpha else None
        if B == 0:
            return (out, alpha) if return_alpha else out
        L = _lib.lib()
        nbytes = L.digat_xattn_workspace_bytes(B, n, d)
        ws = _lib.workspace(nbytes, dev, "xattn")
        W = getattr(self, f"{g}_graph_attention_W")[index]
        F3 = getattr(self, f"{g}_graph_attention_ffn3")[index]
        mode = self.resolved_xattn_mode(g)
        if mode == "sparse" and not return_alpha and n > 16 and d <= 1024:      # the caller knows the graphs are sparse
            _lib.check(L.digat_xattn_fwd_mode(X.data_ptr(), adj.data_ptr(), ctx.data_ptr(), W.weight.data_ptr(), W.bias.data_ptr(),
                                              getattr(self, f"{g}_graph_attention_ffn1")[index].weight.data_ptr(),
                                              getattr(self, f"{g}_graph_attention_ffn2")[index].weight.data_ptr(),
                                              F3.weight.data_ptr(), F3.bias.data_ptr(),
                                              getattr(self, f"{g}_graph_attention_a")[index].weight.data_ptr(),
                                              out.data_ptr(), B, n, d, 2, ws.data_ptr(), nbytes, _lib.stream_ptr()),
                       "digat_xattn_fwd_mode")
            return out
        _lib.check(L.digat_xattn_fwd(X.data_ptr(), adj.data_ptr(), ctx.data_ptr(), W.weight.data_ptr(),
                                     W.bias.data_ptr(),
                                     getattr(self, f"{g}_graph_attention_ffn1")[index].weight.data_ptr(),
                                     getattr(self, f"{g}_graph_attention_ffn2")[index].weight.data_ptr(),
                                     F3.weight.data_ptr(), F3.bias.data_ptr(),
                                     getattr(self, f"{g}_graph_attention_a")[index].weight.data_ptr(),
                                     out.data_ptr(), _lib.ptr(alpha), B, n, d, ws.data_ptr(), nbytes,
                                     _lib.stream_ptr()), "digat_xattn_fwd")
        return (out, alpha) if return_alpha else out

    def compute_news_graph_embeddings(self, index, news_graph_embeddings, news_graph, user_graph_context):
        self._eval_only("compute_news_graph_embeddings")
        return self._xattn("news", index, news_graph_embeddings, news_graph, user_graph_context)

    def compute_user_graph_embeddings(self, index, user_graph_embeddings, user_graph, news_graph_context):
        self._eval_only("compute_user_graph_embeddings")
        return self._xattn("user", index, user_graph_embeddings, user_graph, news_graph_context)

    # ------------------------------------------------------------------ a5 (graphEncoders.py:177-198)
    def _encode(self, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph,
                user_category_mask, user_category_indices, news_graph_context, shared: bool = False):
        Xn, ue = _lib.f32(news_graph_embeddings), _lib.f32(user_news_embedding)
        dev = _lib.require_device(Xn, news_graph, news_graph_mask, ue, user_graph, user_category_mask,
                                  user_category_indices)
        B, N, d = Xn.shape
        H, C = self.max_history_num, self.category_num - 1
        if ue.shape != (B, H, d) or user_graph.shape[1] != H + C or N != self.news_graph_size:
            raise _lib.DigatHipError("input shapes do not match the encoder's configuration")
        An, Mn = _lib.as_bytes(news_graph), _lib.as_bytes(news_graph_mask)
        Au, cm = _lib.as_bytes(user_graph), _lib.as_bytes(user_category_mask)
        ci = user_category_indices.to(torch.int64).contiguous()
        c0 = None if news_graph_context is None else _lib.f32(news_graph_context)
        out_n = torch.empty((B, d), dtype=torch.float32, device=dev)
        out_u = torch.empty((B, d), dtype=torch.float32, device=dev)
        if B == 0:                       # empty tensors have no storage to point at
            return out_n, out_u
        L = _lib.lib()
        # shared: digat_encoder_fwd_shared — the same arguments, runs of identical consecutive user rows found on the device
        nbytes = (L.digat_encoder_shared_workspace_bytes if shared else L.digat_encoder_workspace_bytes)(B, N, H, C, d, self.graph_depth)
        ws = _lib.workspace(nbytes, dev, "encoder")
        P = self._params()
        X = _lib.ext()
        if X is not None:         # the thin torch extension: tensors in, the same C entry point behind it
            X.encoder_fwd(_lib.addressof(P), bool(shared), Xn, An, Mn, ue, Au, cm, ci, c0, out_n, out_u, ws)
            return out_n, out_u
        fn, what = (L.digat_encoder_fwd_shared, "digat_encoder_fwd_shared") if shared else (L.digat_encoder_fwd, "digat_encoder_fwd")
        _lib.check(fn(P, Xn.data_ptr(), An.data_ptr(), Mn.data_ptr(), ue.data_ptr(), Au.data_ptr(),
                      cm.data_ptr(), ci.data_ptr(), _lib.ptr(c0), out_n.data_ptr(), out_u.data_ptr(),
                      B, N, H, ws.data_ptr(), nbytes, _lib.stream_ptr()), what)
        return out_n, out_u

    def project_news_layer0(self, news_graph_embeddings):
        """[h|P|Q] of layer 0 of the news graph for M news graphs ([M,N,d] -> [3,M,N,d]): they depend on the news alone, so a
        driver can keep them per news next to the news representations and c_n0 (``util.prepare_news_side``) and hand the
        batch's rows to ``inference_grouped(news_hpq0=...)``.  Same launch, same bits as inside the encoder."""
        X = _lib.f32(news_graph_embeddings)
        dev = _lib.require_device(X)
        M, N, d = X.shape
        out = torch.empty((3, M, N, d), dtype=torch.float32, device=dev)
        if M:
            _lib.check(_lib.lib().digat_news_project0(self._params(), X.data_ptr(), out.data_ptr(), M, N, _lib.stream_ptr()),
                       "digat_news_project0")
        return out

    def project_user_layer0(self, rows):
        """[h|P|Q] of layer 0 of the USER graph for M node embeddings ([M,d] -> [3,M,d]): row-wise, so they can be kept per
        news (history nodes) and per topic (``self.topic_node_embedding``) — see ``util.prepare_news_side``."""
        X = _lib.f32(rows)
        dev = _lib.require_device(X)
        M, d = X.shape
        out = torch.empty((3, M, d), dtype=torch.float32, device=dev)
        if M:
            _lib.check(_lib.lib().digat_user_project0(self._params(), X.data_ptr(), out.data_ptr(), M, _lib.stream_ptr()),
                       "digat_user_project0")
        return out

    def news_context_queries(self, news_graph_context):
        """[topic query | user-attention query | K3 of the user graph's layer 0] of M news contexts ([M,d] -> [3,M,d]): linear maps
        of the candidate's cached c_n0, so a driver keeps them per news next to c_n0 (``util.prepare_news_side``) and hands the
        batch's rows to ``inference_grouped(ctxq0=...)``.  Same launch, same bits as inside the encoder (eval mode)."""
        c = _lib.f32(news_graph_context)
        dev = _lib.require_device(c)
        M, d = c.shape
        out = torch.empty((3, M, d), dtype=torch.float32, device=dev)
        if self.graph_depth == 0:
            out[2].zero_()
        if M:
            _lib.check(_lib.lib().digat_news_context_queries(self._params(), c.data_ptr(), out.data_ptr(), M, _lib.stream_ptr()),
                       "digat_news_context_queries")
        return out

    def inference_grouped(self, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph,
                          user_category_mask, user_category_indices, row_group, news_graph_context, news_hpq0=None,
                          hist_hpq0=None, topic_hpq0=None, ctxq0=None, news_index=None):
        """``inference`` for rows that share users (not in the reference: its driver expands the user tensors per
        row, util.py:57-67).  The four user tensors are given once per GROUP ([G,...]) and ``row_group`` [B] maps
        each row to its group; results are bit-identical to ``inference`` on the expanded tensors.
        ``news_index`` [B] int64: ``news_graph_embeddings`` and ``news_hpq0`` are then the per-news TABLES ([news_num, N, d],
        [3, news_num, N, d]) and row b reads their row ``news_index[b]`` in place (no gathered copies)."""
        Xn, ue = _lib.f32(news_graph_embeddings), _lib.f32(user_news_embedding)
        dev = _lib.require_device(Xn, news_graph, news_graph_mask, ue, user_graph, user_category_mask,
                                  user_category_indices, row_group)
        _, N, d = Xn.shape
        B = news_graph.shape[0]
        if news_index is None and Xn.shape[0] != B:
            raise ValueError("news_graph_embeddings must have one graph per row (or pass news_index with the per-news tables)")
        G = ue.shape[0]
        H, C = self.max_history_num, self.category_num - 1
        if B == 0 or 4 * G > B or self.training:       # empty batch, or too few rows per group to pay off: expand, plain path
            rg = row_group.long()
            if news_index is not None:
                Xn = Xn.index_select(0, news_index.long())
            return self._encode(Xn, news_graph, news_graph_mask, ue.index_select(0, rg), user_graph.index_select(0, rg),
                                user_category_mask.index_select(0, rg), user_category_indices.index_select(0, rg),
                                news_graph_context)
        An, Mn = _lib.as_bytes(news_graph), _lib.as_bytes(news_graph_mask)
        Au, cm = _lib.as_bytes(user_graph), _lib.as_bytes(user_category_mask)
        ci = user_category_indices.to(torch.int64).contiguous()
        rg = row_group.to(torch.int32).contiguous()
        c0 = _lib.f32(news_graph_context)
        out_n = torch.empty((B, d), dtype=torch.float32, device=dev)
        out_u = torch.empty((B, d), dtype=torch.float32, device=dev)
        L = _lib.lib()
        nbytes = L.digat_encoder_grouped_workspace_bytes(B, N, H, C, d, self.graph_depth)
        ws = _lib.workspace(nbytes, dev, "encoder")
        P = self._params()
        X = _lib.ext()
        hpq = hh = th = cq = ni = None
        if news_hpq0 is not None or hist_hpq0 is not None or ctxq0 is not None or news_index is not None:
            M = 0
            if news_index is not None:
                if news_hpq0 is None or news_graph_context is None:
                    raise ValueError("news_index needs news_hpq0 (the per-news table) and news_graph_context")
                ni = news_index.to(torch.int64).contiguous()
                M = Xn.shape[0]
                if ni.shape[0] != B or tuple(news_hpq0.shape) != (3, M, N, d):
                    raise ValueError("with news_index: news_graph_embeddings [M, N, d], news_hpq0 [3, M, N, d], news_index [B]")
            if ctxq0 is not None:
                cq = _lib.f32(ctxq0)
                if tuple(cq.shape) != (3, B, d):
                    raise ValueError("ctxq0 must be [3, B, d] (news_context_queries of the batch's news contexts)")
            if news_hpq0 is not None:
                hpq = _lib.f32(news_hpq0)
                if ni is None and tuple(hpq.shape) != (3, B, N, d):
                    raise ValueError("news_hpq0 must be [3, B, N, d] (project_news_layer0 of the batch's candidates)")
            if hist_hpq0 is not None:
                hh, th = _lib.f32(hist_hpq0), _lib.f32(topic_hpq0)
                if tuple(hh.shape) != (3, G, H, d) or tuple(th.shape) != (3, C, d):
                    raise ValueError("hist_hpq0 must be [3, G, H, d] and topic_hpq0 [3, C, d] (project_user_layer0)")
            if X is not None:
                X.encoder_fwd_grouped(_lib.addressof(P), Xn, An, Mn, ue, Au, cm, ci, rg, c0, hpq, hh, th, cq, ni, out_n, out_u, ws)
                return out_n, out_u
            _lib.check(L.digat_encoder_fwd_grouped_cached(P, Xn.data_ptr(), An.data_ptr(), Mn.data_ptr(), ue.data_ptr(),
                                                          Au.data_ptr(), cm.data_ptr(), ci.data_ptr(), rg.data_ptr(), c0.data_ptr(),
                                                          _lib.ptr(hpq), _lib.ptr(hh), _lib.ptr(th), _lib.ptr(cq), _lib.ptr(ni), M,
                                                          out_n.data_ptr(),
                                                          out_u.data_ptr(), B, G, N, H, ws.data_ptr(), nbytes, _lib.stream_ptr()),
                       "digat_encoder_fwd_grouped_cached")
            return out_n, out_u
        if X is not None:
            X.encoder_fwd_grouped(_lib.addressof(P), Xn, An, Mn, ue, Au, cm, ci, rg, c0, None, None, None, None, None, out_n, out_u, ws)
            return out_n, out_u
        _lib.check(L.digat_encoder_fwd_grouped(P, Xn.data_ptr(), An.data_ptr(), Mn.data_ptr(), ue.data_ptr(), Au.data_ptr(),
                                               cm.data_ptr(), ci.data_ptr(), rg.data_ptr(), c0.data_ptr(), out_n.data_ptr(),
                                               out_u.data_ptr(), B, G, N, H, ws.data_ptr(), nbytes, _lib.stream_ptr()),
                   "digat_encoder_fwd_grouped")
        return out_n, out_u

    def forward(self, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph,
                user_category_mask, user_category_indices):
        if self.training and torch.is_grad_enabled():
            from .training import digat_forward_train
            return digat_forward_train(self, news_graph_embeddings, news_graph, news_graph_mask,
                                       user_news_embedding, user_graph, user_category_mask, user_category_indices)
        return self._encode(news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph,
                            user_category_mask, user_category_indices, None)

    def inference(self, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph,
                  user_category_mask, user_category_indices, news_graph_context):
        """graphEncoders.py:189-198.  The reference's driver expands an impression's user tensors once per candidate
        (util.py:57-67), so consecutive rows of a dev batch carry bit-identical users: with ``detect_shared_users`` (default) the
        call goes to ``digat_encoder_fwd_shared`` — the runs are found on the device (every byte of the four user tensors compared
        with the previous row's), nothing is read back by the host, and layer 0 of the user graph is computed once per run.
        Bit-identical to the per-row entry; rows that share nothing cost the comparison pass on top of it."""
        want = bool(self._launch_option("detect_shared_users") and not self.training and self.graph_depth > 0
                    and news_graph_embeddings.shape[0] >= self.SHARED_USERS_MIN_ROWS)
        if (want and self.user_xattn_mode == "auto" and "user" not in self.corpus_xattn_hint and user_graph.is_cuda
                and not torch.cuda.is_current_stream_capturing()):
            # a driver that never showed the corpus to util.prepare_news_side (the reference's own loop): the sparse / dense choice
            # of the user graph's Eq. 8 is made here, ONCE, from the first batch (one host read; the library's per-batch device-side
            # choice launches both variants, and only the sparse one can share layer 0 between the rows of a run)
            from .util import SPARSE_ENTRIES_PER_NODE
            per_node = float(user_graph.sum(dtype=torch.float64) / max(1, user_graph.shape[0] * user_graph.shape[1]))
            self.corpus_xattn_hint = dict(self.corpus_xattn_hint, user="sparse" if per_node <= SPARSE_ENTRIES_PER_NODE else "dense")
        shared = want and self.resolved_xattn_mode("user") == "sparse"
        return self._encode(news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph,
                            user_category_mask, user_category_indices, news_graph_context, shared=shared)

    detect_shared_users = True         # inference(): look for runs of identical consecutive user rows (see there)
    SHARED_USERS_MIN_ROWS = 128        # ... in batches of at least this many rows

    def _shared_user_runs(self, Xn, ue, Au, cm, ci):
        """(row_group [B] int32, leaders [G] int64) when the rows of this batch are runs of identical users worth grouping, else None:
        the search of ``digat_encoder_fwd_shared`` as a stand-alone call (``digat_user_row_runs``; one host read of the run count) for
        drivers that want to call ``inference_grouped`` themselves, and for the tests."""
        B = ue.shape[0]
        if (not self._launch_option("detect_shared_users") or self.training or B < self.SHARED_USERS_MIN_ROWS or self.graph_depth == 0
                or self.resolved_xattn_mode("user") == "dense" or ue.dtype != torch.float32 or not ue.is_cuda
                or torch.cuda.is_current_stream_capturing()):          # the run count is read on the host: not inside a graph capture
            return None
        H, d = self.max_history_num, self.news_embedding_dim
        U, C1 = self.user_graph_size, self.category_num
        if tuple(ue.shape) != (B, H, d) or tuple(Au.shape) != (B, U, U) or tuple(cm.shape) != (B, C1) or tuple(ci.shape) != (B, H):
            return None
        dev = _lib.require_device(ue, Au, cm, ci)
        ue_c, Au_b, cm_b = _lib.f32(ue), _lib.as_bytes(Au), _lib.as_bytes(cm)
        ci_c = ci.to(torch.int64).contiguous()
        row_group = torch.empty(B, dtype=torch.int32, device=dev)
        leaders = torch.empty(B, dtype=torch.int64, device=dev)
        count = torch.empty(1, dtype=torch.int32, device=dev)
        ws = _lib.workspace(B, dev, "runs")
        X = _lib.ext()
        if X is not None:
            X.user_row_runs(ue_c, Au_b, cm_b, ci_c, row_group, leaders, count, ws)
        else:
            _lib.check(_lib.lib().digat_user_row_runs(ue_c.data_ptr(), Au_b.data_ptr(), cm_b.data_ptr(), ci_c.data_ptr(), B, H, U, C1, d,
                                                      row_group.data_ptr(), leaders.data_ptr(), count.data_ptr(), ws.data_ptr(), B,
                                                      _lib.stream_ptr()), "digat_user_row_runs")
        G = int(count.item())              # the one host read of the drop-in path
        if 4 * G > B:
            return None
        return row_group, leaders[:G]


# ======================================================================================================
# the five ablation encoders of the reference (graphEncoders.py:201-842), on the same kernels
# ======================================================================================================
class _Ablation(GraphEncoder):
    """Shared machinery.  A subclass names which graph runs Eq. 8 (``EQ8``), which runs the vanilla-GAT layer (``GAT``)
    and whether the news context exists (``NEWS_CONTEXT``); parameter names are the reference's, so its checkpoints
    load.  Inference / eval-mode forward run on the HIP kernels (``digat_xattn_fwd``, ``digat_gat_fwd``,
    ``digat_news_ctx_fwd``, ``digat_user_ctx_fwd``); training-mode forward (with autograd) goes through
    ``training.ablation_forward_train`` on the ``digat_*_fwd_train`` / ``digat_*_bwd`` pairs."""
    EQ8: tuple = ()
    GAT: tuple = ()
    NEWS_CONTEXT = True

    def __init__(self, config, news_embedding_dim: int):
        super().__init__(config, news_embedding_dim)
        d, L = self.news_embedding_dim, self.graph_depth
        if d % 4 != 0:
            raise ValueError("digat_amd needs news_embedding_dim % 4 == 0 (float4 rows)")
        if self.NEWS_CONTEXT:
            self.candidate_attention = ScaledDotProductAttention(d, d, d)
            self.news_graph_W = nn.Linear(d * 2, d, bias=True)
        self.user_news_K = nn.Linear(d, d, bias=False)
        self.user_news_Q = nn.Linear(d, d, bias=True)
        self.featureAffine = nn.Linear(d, d, bias=True)
        self.userAttention = ScaledDotProductAttention(d, d, d)
        for g in self.EQ8:
            setattr(self, f"{g}_graph_attention_W", nn.ModuleList([nn.Linear(d, d, bias=True) for _ in range(L)]))
            setattr(self, f"{g}_graph_attention_ffn1", nn.ModuleList([nn.Linear(d, d, bias=False) for _ in range(L)]))
            setattr(self, f"{g}_graph_attention_ffn2", nn.ModuleList([nn.Linear(d, d, bias=False) for _ in range(L)]))
            setattr(self, f"{g}_graph_attention_ffn3", nn.ModuleList([nn.Linear(d, d, bias=True) for _ in range(L)]))
            setattr(self, f"{g}_graph_attention_a", nn.ModuleList([nn.Linear(d, 1, bias=False) for _ in range(L)]))
        for g in self.GAT:
            setattr(self, f"{g}_graph_attention_W", nn.ModuleList([nn.Linear(d, d, bias=True) for _ in range(L)]))
            setattr(self, f"{g}_graph_attention_a1", nn.ModuleList([nn.Linear(d, 1, bias=False) for _ in range(L)]))
            setattr(self, f"{g}_graph_attention_a2", nn.ModuleList([nn.Linear(d, 1, bias=False) for _ in range(L)]))

    def initialize(self):
        super().initialize()
        relu_gain = nn.init.calculate_gain('relu')
        leaky_gain = nn.init.calculate_gain('leaky_relu', 0.2)
        for g in self.EQ8 + self.GAT:
            for i in range(self.graph_depth):
                nn.init.xavier_uniform_(getattr(self, f"{g}_graph_attention_W")[i].weight)
                nn.init.zeros_(getattr(self, f"{g}_graph_attention_W")[i].bias)
        for g in self.EQ8:
            for i in range(self.graph_depth):
                nn.init.xavier_uniform_(getattr(self, f"{g}_graph_attention_a")[i].weight, gain=leaky_gain)
                for f in ("ffn1", "ffn2", "ffn3"):
                    nn.init.xavier_uniform_(getattr(self, f"{g}_graph_attention_{f}")[i].weight, gain=relu_gain)
                nn.init.zeros_(getattr(self, f"{g}_graph_attention_ffn3")[i].bias)
        for g in self.GAT:
            for i in range(self.graph_depth):
                nn.init.xavier_uniform_(getattr(self, f"{g}_graph_attention_a1")[i].weight, gain=leaky_gain)
                nn.init.xavier_uniform_(getattr(self, f"{g}_graph_attention_a2")[i].weight, gain=leaky_gain)
        if self.NEWS_CONTEXT:
            self.candidate_attention.initialize()
            nn.init.xavier_uniform_(self.news_graph_W.weight)
            nn.init.zeros_(self.news_graph_W.bias)
        nn.init.xavier_uniform_(self.user_news_K.weight)
        nn.init.xavier_uniform_(self.user_news_Q.weight)
        nn.init.zeros_(self.user_news_Q.bias)
        nn.init.xavier_uniform_(self.featureAffine.weight, gain=relu_gain)
        nn.init.zeros_(self.featureAffine.bias)
        self.userAttention.initialize()

    # the context functions and the Eq. 8 layer are DIGAT's (same attribute names)
    _eval_only = DIGAT._eval_only
    compute_news_graph_context = DIGAT.compute_news_graph_context
    compute_user_graph_context = DIGAT.compute_user_graph_context
    _xattn = DIGAT._xattn
    resolved_xattn_mode = DIGAT.resolved_xattn_mode
    user_xattn_mode = "auto"      # Eq. 8 variant per graph for digat_xattn_fwd_mode ("auto" = dense until
    news_xattn_mode = "auto"      # util.prepare_news_side has looked at the corpus)
    corpus_xattn_hint: dict = {}  # replaced per instance by util.prepare_news_side

    def _gat(self, g: str, index: int, X, A):
        """Vanilla GAT update layer (graphEncoders.py:493-519)."""
        self._eval_only("vanilla GAT layer")
        X = _lib.f32(X)
        dev = _lib.require_device(X, A)
        B, n, d = X.shape
        out = torch.empty_like(X)
        if B == 0:
            return out
        adj = _lib.as_bytes(A)
        L = _lib.lib()
        nbytes = L.digat_gat_workspace_bytes(B, n, d)
        ws = _lib.workspace(nbytes, dev, "gat")
        W = getattr(self, f"{g}_graph_attention_W")[index]
        _lib.check(L.digat_gat_fwd(X.data_ptr(), adj.data_ptr(), W.weight.data_ptr(), W.bias.data_ptr(),
                                   getattr(self, f"{g}_graph_attention_a1")[index].weight.data_ptr(),
                                   getattr(self, f"{g}_graph_attention_a2")[index].weight.data_ptr(), out.data_ptr(), B, n, d,
                                   ws.data_ptr(), nbytes, _lib.stream_ptr()), "digat_gat_fwd")
        return out

    def _user_nodes(self, user_news_embedding):
        ue = _lib.f32(user_news_embedding)
        return torch.cat([ue, self.topic_node_embedding.unsqueeze(0).expand(ue.shape[0], -1, -1)], dim=1).contiguous()

    def _layer(self, g: str, i: int, X, A, ctx):
        if g in self.EQ8:
            self._eval_only("Eq. 8 layer")
            return self._xattn(g, i, X, A, ctx)
        return self._gat(g, i, X, A)

    def _encode(self, Xn, An, Mn, ue, Au, cm, ci, c_n):
        Xu = self._user_nodes(ue)
        c_u = self.compute_user_graph_context(Xu, cm, ci, c_n)
        for i in range(self.graph_depth):
            # both updates read the PREVIOUS contexts; tuple assignment keeps the reference's order of evaluation
            Xn, Xu = self._layer("news", i, Xn, An, c_u), self._layer("user", i, Xu, Au, c_n)
            c_n = c_n + self.compute_news_graph_context(Xn, Mn)
            c_u = c_u + self.compute_user_graph_context(Xu, cm, ci, c_n)
        return c_n, c_u

    def forward(self, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph,
                user_category_mask, user_category_indices):
        if self.training and torch.is_grad_enabled():
            from .training import ablation_forward_train
            return ablation_forward_train(self, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding,
                                          user_graph, user_category_mask, user_category_indices)
        c_n = self.compute_news_graph_context(news_graph_embeddings, news_graph_mask)
        return self._encode(_lib.f32(news_graph_embeddings), news_graph, news_graph_mask, user_news_embedding, user_graph,
                            user_category_mask, user_category_indices, c_n)

    def inference(self, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph,
                  user_category_mask, user_category_indices, news_graph_context):
        return self._encode(_lib.f32(news_graph_embeddings), news_graph, news_graph_mask, user_news_embedding, user_graph,
                            user_category_mask, user_category_indices, _lib.f32(news_graph_context))


class wo_SA(_Ablation):
    """graphEncoders.py:201-293 — no semantic-augmentation graph: the candidate's own representation is the context."""
    EQ8, GAT, NEWS_CONTEXT = ("user",), (), False

    def compute_user_graph_embeddings(self, index, user_graph_embeddings, user_graph, news_graph_context):
        return self._layer("user", index, user_graph_embeddings, user_graph, news_graph_context)

    def _run(self, news_graph_embeddings, user_news_embedding, user_graph, user_category_mask, user_category_indices):
        c = _lib.f32(news_graph_embeddings)[:, 0].contiguous()
        Xu = self._user_nodes(user_news_embedding)
        for i in range(self.graph_depth):
            Xu = self._layer("user", i, Xu, user_graph, c)
        return c, self.compute_user_graph_context(Xu, user_category_mask, user_category_indices, c)

    def forward(self, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph,
                user_category_mask, user_category_indices):
        if self.training and torch.is_grad_enabled():
            from .training import ablation_forward_train
            return ablation_forward_train(self, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding,
                                          user_graph, user_category_mask, user_category_indices)
        return self._run(news_graph_embeddings, user_news_embedding, user_graph, user_category_mask, user_category_indices)

    def inference(self, news_graph_embeddings, news_graph, news_graph_mask, user_news_embedding, user_graph,
                  user_category_mask, user_category_indices, news_graph_context):
        return self._run(news_graph_embeddings, user_news_embedding, user_graph, user_category_mask, user_category_indices)


class Seq_SA(_Ablation):
    """graphEncoders.py:295-408 — the neighbourhood as a sequence: pooled once into the news context, never updated."""
    EQ8, GAT = ("user",), ()

    def compute_news_sequence_context(self, news_graph_embeddings, news_graph_mask):
        return self.compute_news_graph_context(news_graph_embeddings, news_graph_mask)

    def compute_user_graph_embeddings(self, index, user_graph_embeddings, user_graph, news_graph_context):
        return self._layer("user", index, user_graph_embeddings, user_graph, news_graph_context)

    def _encode(self, Xn, An, Mn, ue, Au, cm, ci, c_n):
        Xu = self._user_nodes(ue)
        c_u = self.compute_user_graph_context(Xu, cm, ci, c_n)
        for i in range(self.graph_depth):
            Xu = self._layer("user", i, Xu, Au, c_n)
            c_u = c_u + self.compute_user_graph_context(Xu, cm, ci, c_n)
        return c_n, c_u


class wo_interaction(_Ablation):
    """graphEncoders.py:410-549 — vanilla GAT layers on both graphs."""
    EQ8, GAT = (), ("news", "user")

    def compute_news_graph_embeddings(self, index, news_graph_embeddings, news_graph):
        return self._gat("news", index, news_graph_embeddings, news_graph)

    def compute_user_graph_embeddings(self, index, user_graph_embeddings, user_graph):
        return self._gat("user", index, user_graph_embeddings, user_graph)


class News_graph_wo_inter(_Ablation):
    """graphEncoders.py:551-696 — vanilla GAT on the news graph, Eq. 8 on the user graph."""
    EQ8, GAT = ("user",), ("news",)

    def compute_news_graph_embeddings(self, index, news_graph_embeddings, news_graph):
        return self._gat("news", index, news_graph_embeddings, news_graph)

    def compute_user_graph_embeddings(self, index, user_graph_embeddings, user_graph, news_graph_context):
        return self._layer("user", index, user_graph_embeddings, user_graph, news_graph_context)


class User_graph_wo_inter(_Ablation):
    """graphEncoders.py:698-842 — Eq. 8 on the news graph, vanilla GAT on the user graph."""
    EQ8, GAT = ("news",), ("user",)

    def compute_news_graph_embeddings(self, index, news_graph_embeddings, news_graph, user_graph_context):
        return self._layer("news", index, news_graph_embeddings, news_graph, user_graph_context)

    def compute_user_graph_embeddings(self, index, user_graph_embeddings, user_graph):
        return self._gat("user", index, user_graph_embeddings, user_graph)
