"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's MSA news encoder.

``newsEncoders.MSA.forward`` (newsEncoders.py:70-82) in eval mode: word embedding lookup ->
``layers.MultiHeadAttention`` (layers.py:50-88: W_Q/W_V with bias, W_K without, scores / sqrt(d_k),
softmax over the keys, NO padding mask inside the self-attention, no output projection) -> ReLU ->
``layers.Attention`` (layers.py:91-115: tanh(affine1) . affine2, -1e9 on masked positions, softmax over the
title, weighted sum of the features).  Plain fp32 torch-CPU ops; parameters keyed by the reference's
state_dict names.  Pinned by ``tests/golden/msa_*.npz`` (minted by ``oracle/make_golden.py`` from the
reference's own ``layers.MultiHeadAttention`` / ``layers.Attention`` modules).  Only ``tests/`` may import it.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F


def msa_forward(p: Dict[str, torch.Tensor], title_text: torch.Tensor, title_mask: torch.Tensor, head_num: int, drop=None) -> torch.Tensor:
    """title_text [T, Lw] int64, title_mask [T, Lw] (0 = padding) -> news representation [T, head_num * d_k].
    ``drop`` (train mode): the dropout on the embedded tokens (newsEncoders.py:77), a callable on the [T, Lw, dm] tensor; None = eval."""
    w = p["word_embedding.weight"].index_select(0, title_text.reshape(-1)).view(*title_text.shape, -1)   # :76
    if drop is not None:
        w = drop(w)                                                                                      # :77
    T, Lw, _ = w.shape
    q = F.linear(w, p["multiheadSelfattention.W_Q.weight"], p["multiheadSelfattention.W_Q.bias"])       # layers.py:78
    k = F.linear(w, p["multiheadSelfattention.W_K.weight"])                                             # :79 (no bias)
    v = F.linear(w, p["multiheadSelfattention.W_V.weight"], p["multiheadSelfattention.W_V.bias"])       # :80
    dk = q.shape[-1] // head_num
    q, k, v = (x.view(T, Lw, head_num, dk).transpose(1, 2) for x in (q, k, v))                          # [T,h,Lw,dk]
    alpha = torch.softmax(q @ k.transpose(2, 3) / math.sqrt(float(dk)), dim=3)                          # :84-85
    h = torch.relu((alpha @ v).transpose(1, 2).reshape(T, Lw, head_num * dk))                           # :86-87, newsEncoders.py:78
    a = F.linear(torch.tanh(F.linear(h, p["attention.affine1.weight"], p["attention.affine1.bias"])),
                 p["attention.affine2.weight"]).squeeze(2)                                              # layers.py:108-109
    a = a.masked_fill(title_mask == 0, -1e9)                                                            # :111
    return (torch.softmax(a, dim=1).unsqueeze(1) @ h).squeeze(1)                                        # :114
