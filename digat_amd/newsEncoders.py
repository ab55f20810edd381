"""News encoders: UPSTREAM of the hot path.  ``MSA`` on the GPU runs on the HIP kernels: inference on
``digat_msa_fwd`` (``csrc/digat_news.inc``), training forward / backward on ``digat_msa_fwd_train`` / ``digat_msa_bwd``
(``csrc/digat_news_train.inc``); on the CPU (tests without a GPU) and for ``CNN`` the stock PyTorch modules below.

Their output ``[., news_embedding_dim]`` is the graph encoder's input.  They are restated here
so that ``Model.forward`` (training) and the news-representation cache of ``compute_scores``
have a producer with the reference's parameter names (``word_embedding``, ``multiheadSelfattention.
W_{K,Q,V}``, ``attention.affine{1,2}``, ``conv.conv``) and the same semantics:
word embedding -> dropout -> MSA (16 heads x 25) + ReLU | Conv1d + ReLU -> additive tanh attention.
GloVe initialisation needs the downloaded vectors; without them the table keeps its random init.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class MultiHeadAttention(nn.Module):
    """layers.py:50-88 (no output projection, K without bias)."""

    def __init__(self, h: int, d_model: int, d_k: int, d_v: int):
        super().__init__()
        self.h, self.d_k, self.d_v = h, d_k, d_v
        self.W_K = nn.Linear(d_model, h * d_k, bias=False)
        self.W_Q = nn.Linear(d_model, h * d_k, bias=True)
        self.W_V = nn.Linear(d_model, h * d_v, bias=True)

    def initialize(self):
        nn.init.zeros_(self.W_Q.bias)
        nn.init.zeros_(self.W_V.bias)

    def forward(self, x):
        B, T, _ = x.shape
        q = self.W_Q(x).view(B, T, self.h, self.d_k).transpose(1, 2)
        k = self.W_K(x).view(B, T, self.h, self.d_k).transpose(1, 2)
        v = self.W_V(x).view(B, T, self.h, self.d_v).transpose(1, 2)
        alpha = F.softmax(q @ k.transpose(2, 3) / math.sqrt(float(self.d_k)), dim=3)
        return (alpha @ v).transpose(1, 2).reshape(B, T, self.h * self.d_v)


class Attention(nn.Module):
    """layers.py:91-115: additive attention pooling, -1e9 mask."""

    def __init__(self, feature_dim: int, attention_dim: int):
        super().__init__()
        self.affine1 = nn.Linear(feature_dim, attention_dim, bias=True)
        self.affine2 = nn.Linear(attention_dim, 1, bias=False)

    def initialize(self):
        nn.init.xavier_uniform_(self.affine1.weight, gain=nn.init.calculate_gain('tanh'))
        nn.init.zeros_(self.affine1.bias)
        nn.init.xavier_uniform_(self.affine2.weight)

    def forward(self, feature, mask=None):
        a = self.affine2(torch.tanh(self.affine1(feature))).squeeze(2)
        if mask is not None:
            a = a.masked_fill(mask == 0, -1e9)
        return (F.softmax(a, dim=1).unsqueeze(1) @ feature).squeeze(1)


class _Conv(nn.Module):
    def __init__(self, in_channels, kernels, window):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, kernels, kernel_size=window, padding=(window - 1) // 2)

    def initialize(self):
        pass

    def forward(self, x):
        return F.relu(self.conv(x))


class MsaFused(torch.autograd.Function):
    """The MSA news encoder as one library call per direction (``digat_msa_fwd_train`` / ``digat_msa_bwd``,
    ``digat_embedding_bwd`` for the word-embedding rows)."""

    @staticmethod
    def _params(table, WQ, bQ, WK, WV, bV, A1, b1, a2, heads, dk):
        from . import _lib
        P = _lib.MsaParams(word_embedding_dim=table.shape[1], head_num=heads, head_dim=dk, attention_dim=A1.shape[0])
        for name, w in zip(("word_embedding", "W_Q", "b_Q", "W_K", "W_V", "b_V", "A1", "b1", "a2"), (table, WQ, bQ, WK, WV, bV, A1, b1, a2)):
            setattr(P, name, w.data_ptr())
        return P

    @staticmethod
    def forward(ctx, tokens, mask, table, WQ, bQ, WK, WV, bV, A1, b1, a2, heads, dk, p_drop):
        from . import _lib
        L = _lib.lib()
        ws_ = [w.detach().float().contiguous() for w in (table, WQ, bQ, WK, WV, bV, A1, b1, a2)]
        dev = _lib.require_device(tokens, mask, *ws_)
        T, Lw = tokens.shape
        dm, att = ws_[0].shape[1], ws_[6].shape[0]
        P = MsaFused._params(*ws_, heads, dk)
        out = torch.empty((T, heads * dk), dtype=torch.float32, device=dev)
        nsave = L.digat_msa_train_save_bytes(T, Lw, dm, heads, dk, att)
        nws = L.digat_msa_train_workspace_bytes(T, Lw, dm, heads, dk, att)
        save = torch.empty(max(int(nsave), 256), dtype=torch.uint8, device=dev)
        ws = _lib.workspace(nws, dev, "msa_train")
        p = float(p_drop)
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0 else 0
        if T:
            _lib.check(L.digat_msa_fwd_train(P, tokens.data_ptr(), mask.data_ptr(), out.data_ptr(), p, seed, T, Lw, save.data_ptr(),
                                             nsave, ws.data_ptr(), nws, _lib.stream_ptr()), "digat_msa_fwd_train")
        ctx.save_for_backward(tokens, mask, save, *ws_)
        ctx.dims, ctx.p, ctx.sizes = (heads, dk), p, (nsave, nws)
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import _lib
        L = _lib.lib()
        tokens, mask, save, *ws_ = ctx.saved_tensors
        table, WQ, bQ, WK, WV, bV, A1, b1, a2 = ws_
        heads, dk = ctx.dims
        T, Lw = tokens.shape
        dm, hd, att = table.shape[1], heads * dk, A1.shape[0]
        dev = tokens.device
        if T == 0:                                # no title: every gradient is zero
            z = [torch.zeros_like(w) for w in ws_]
            return (None, None, z[0] if ctx.needs_input_grad[2] else None, *z[1:], None, None, None)
        dout = dout.float().contiguous()
        P = MsaFused._params(*ws_, heads, dk)
        nsave, nws = ctx.sizes
        ws = _lib.workspace(nws, dev, "msa_train")
        f = dict(dtype=torch.float32, device=dev)
        ld = int(L.digat_msa_row_grad_ld(T, Lw, dm))
        row_grad = torch.empty((T * Lw, ld), **f)
        dW3 = torch.empty((3, hd, dm), **f)       # one buffer: the library writes its single [3 hd, dm] weight-gradient product in place
        dWQ, dWK, dWV = dW3[0], dW3[1], dW3[2]
        dbQ, dbV = torch.empty(hd, **f), torch.empty(hd, **f)
        dA1, db1, da2 = torch.empty((att, hd), **f), torch.empty(att, **f), torch.empty(att, **f)
        _lib.check(L.digat_msa_bwd(P, tokens.data_ptr(), mask.data_ptr(), dout.data_ptr(), ctx.p, save.data_ptr(), nsave,
                                   row_grad.data_ptr(), ld, dWQ.data_ptr(), dbQ.data_ptr(), dWK.data_ptr(), dWV.data_ptr(), dbV.data_ptr(),
                                   dA1.data_ptr(), db1.data_ptr(), da2.data_ptr(), T, Lw, ws.data_ptr(), nws, _lib.stream_ptr()),
                   "digat_msa_bwd")
        dtable = None
        if ctx.needs_input_grad[2]:
            # index plumbing only: the rows in token order (stable); the sums run in the library, in that fixed order
            stok, order = torch.sort(tokens.reshape(-1).to(torch.int64), stable=True)
            stok, order = stok.to(torch.int32), order.to(torch.int32)
            dtable = torch.zeros_like(table)
            nb = L.digat_embedding_bwd_workspace_bytes(T * Lw, dm)
            ews = _lib.workspace(nb, dev, "emb_bwd")
            _lib.check(L.digat_embedding_bwd(row_grad.data_ptr(), ld, order.data_ptr(), stok.data_ptr(), T * Lw, dm, dtable.data_ptr(),
                                             ews.data_ptr(), nb, _lib.stream_ptr()), "digat_embedding_bwd")
        return None, None, dtable, dWQ, dbQ, dWK, dWV, dbV, dA1, db1, da2.view_as(a2), None, None, None


class NewsEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embedding_dim = config.word_embedding_dim
        self.max_sentence_length = config.max_title_length
        self.word_embedding = nn.Embedding(config.vocabulary_size, self.word_embedding_dim)
        self.dropout = nn.Dropout(p=config.dropout_rate)

    def initialize(self):
        pass

    def _words(self, title_text):
        B, n = title_text.shape[:2]
        w = self.dropout(self.word_embedding(title_text.long()))
        return w.view(B * n, self.max_sentence_length, self.word_embedding_dim), B, n


class MSA(NewsEncoder):
    """newsEncoders.py:58-82."""

    def __init__(self, config):
        super().__init__(config)
        self.multiheadSelfattention = MultiHeadAttention(config.MSA_head_num, config.word_embedding_dim,
                                                         config.MSA_head_dim, config.MSA_head_dim)
        self.news_embedding_dim = config.MSA_head_num * config.MSA_head_dim
        self.attention = Attention(self.news_embedding_dim, config.attention_dim)

    def initialize(self):
        self.multiheadSelfattention.initialize()
        self.attention.initialize()

    def forward(self, title_text, title_mask):
        if title_text.is_cuda and not torch.is_grad_enabled():
            if not self.training or self.dropout.p == 0:
                return self.encode_hip(title_text, title_mask)  # inference: the HIP kernels (digat_msa_fwd)
        if title_text.is_cuda and torch.is_grad_enabled() and title_text.shape[-1] <= 32:
            return self.train_hip(title_text, title_mask)       # training: digat_msa_fwd_train / digat_msa_bwd
        return self.forward_stock(title_text, title_mask)       # CPU (tests without a GPU)

    def forward_stock(self, title_text, title_mask):
        """The same function on stock PyTorch modules (CPU runs; the yardstick of tools/kbench.py msa-train)."""
        w, B, n = self._words(title_text)
        h = F.relu(self.multiheadSelfattention(w))
        return self.attention(h, mask=title_mask.view(B * n, -1)).view(B, n, self.news_embedding_dim)

    def train_hip(self, title_text, title_mask):
        """Forward with autograd through the HIP pair (newsEncoders.py:70-82; dropout on the embedded tokens in train mode)."""
        shape = title_text.shape
        Lw = shape[-1]
        mha, att = self.multiheadSelfattention, self.attention
        out = MsaFused.apply(title_text.reshape(-1, Lw).to(torch.int32).contiguous(),
                             (title_mask.reshape(-1, Lw) != 0).to(torch.uint8).contiguous(),
                             self.word_embedding.weight, mha.W_Q.weight, mha.W_Q.bias, mha.W_K.weight, mha.W_V.weight, mha.W_V.bias,
                             att.affine1.weight, att.affine1.bias, att.affine2.weight, mha.h, mha.d_k,
                             float(self.dropout.p) if self.training else 0.0)
        return out.view(*shape[:-1], self.news_embedding_dim)

    # ---- inference on the HIP kernels (digat_news.inc)
    def _hip_params(self):
        from . import _lib
        ws = [self.word_embedding.weight, self.multiheadSelfattention.W_Q.weight, self.multiheadSelfattention.W_Q.bias,
              self.multiheadSelfattention.W_K.weight, self.multiheadSelfattention.W_V.weight,
              self.multiheadSelfattention.W_V.bias, self.attention.affine1.weight, self.attention.affine1.bias,
              self.attention.affine2.weight]
        key = tuple((w.data_ptr(), w._version) for w in ws)
        cached = getattr(self, "_hip_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        L = _lib.lib()
        mha = self.multiheadSelfattention
        dm, hd, att = self.word_embedding_dim, mha.h * mha.d_k, self.attention.affine1.out_features
        dev = ws[0].device
        keep = [w.detach().float().contiguous() for w in ws]
        P = _lib.MsaParams(word_embedding_dim=dm, head_num=mha.h, head_dim=mha.d_k, attention_dim=att)
        for name, w in zip(("word_embedding", "W_Q", "b_Q", "W_K", "W_V", "b_V", "A1", "b1", "a2"), keep):
            setattr(P, name, w.data_ptr())
        if hd % 80 == 0 and dm % 4 == 0 and dm >= 32:          # the bf16x6 matrix-core path (fp32-grade)
            qkv = _lib.split_buffer(L.digat_msa_split_bytes(dm, mha.h, mha.d_k), dev)
            _lib.check(L.digat_split_msa_weights(keep[1].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(), dm, hd,
                                                 qkv.data_ptr(), _lib.stream_ptr()), "digat_split_msa_weights")
            a1 = _lib.split_buffer(L.digat_split_weights_bytes(att, hd), dev)
            # the MSA encoder's operand format is bf16x6 (no range limit: word embeddings are whatever the vocabulary file holds)
            _lib.check(L.digat_split_weights(keep[6].data_ptr(), att, hd, a1.data_ptr(), _lib.GEMM_BF16X6, _lib.stream_ptr()), "digat_split_weights")
            P.qkv_wsplit, P.a1_wsplit = qkv.data_ptr(), a1.data_ptr()
            keep += [qkv, a1]
        self._hip_cache = (key, (P, keep))
        return P, keep

    def encode_hip(self, title_text, title_mask):
        """title_text / title_mask [B, n, Lw] (or [T, Lw]) on the GPU -> [B, n, news_embedding_dim] ([T, ...])."""
        from . import _lib
        shape = title_text.shape
        Lw = shape[-1]
        tok = title_text.reshape(-1, Lw).to(torch.int32).contiguous()
        msk = (title_mask.reshape(-1, Lw) != 0).to(torch.uint8).contiguous()
        dev = _lib.require_device(tok, msk)
        T = tok.shape[0]
        P, _keep = self._hip_params()
        out = torch.empty((T, self.news_embedding_dim), dtype=torch.float32, device=dev)
        if T:
            L = _lib.lib()
            mha = self.multiheadSelfattention
            nbytes = L.digat_msa_workspace_bytes(T, Lw, self.word_embedding_dim, mha.h, mha.d_k, self.attention.affine1.out_features)
            ws = _lib.workspace(nbytes, dev, "msa")
            _lib.check(L.digat_msa_fwd(P, tok.data_ptr(), msk.data_ptr(), out.data_ptr(), T, Lw, ws.data_ptr(), nbytes,
                                       _lib.stream_ptr()), "digat_msa_fwd")
        return out.view(*shape[:-1], self.news_embedding_dim)


class CNN(NewsEncoder):
    """newsEncoders.py:30-54 with the 'naive' Conv1D (layers.py:13-14)."""

    def __init__(self, config):
        super().__init__(config)
        self.conv = _Conv(config.word_embedding_dim, config.cnn_kernel_num, config.cnn_window_size)
        self.news_embedding_dim = config.cnn_kernel_num
        self.attention = Attention(self.news_embedding_dim, config.attention_dim)

    def initialize(self):
        self.attention.initialize()

    def forward(self, title_text, title_mask):
        w, B, n = self._words(title_text)
        h = self.dropout(self.conv(w.permute(0, 2, 1)).permute(0, 2, 1))
        return self.attention(h, mask=title_mask.view(B * n, -1)).view(B, n, self.news_embedding_dim)
