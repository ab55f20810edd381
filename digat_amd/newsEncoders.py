"""News encoders: UPSTREAM of the hot path, kept as stock PyTorch-ROCm modules (rocBLAS / MIOpen).

SURVEY.md §2 marks the reference's ``newsEncoders.py`` / ``layers.py:7-115`` out of scope for the HIP
work: their output ``[., news_embedding_dim]`` is the graph encoder's input.  They are restated here
only so that ``Model.forward`` (training) and the news-representation cache of ``compute_scores``
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
        w, B, n = self._words(title_text)
        h = F.relu(self.multiheadSelfattention(w))
        return self.attention(h, mask=title_mask.view(B * n, -1)).view(B, n, self.news_embedding_dim)


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
