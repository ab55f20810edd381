"""Parameter containers with the reference's names (layers.py:181-191).

Only what ``--graph_encoder=DIGAT`` instantiates is mirrored: ``ScaledDotProductAttention`` holds
``K`` (no bias) and ``Q`` (bias) so that ``candidate_attention.K.weight`` etc. keep their
state_dict keys.  The arithmetic (layers.py:199-206) runs inside the HIP context kernels.
"""
import math

import torch.nn as nn


class ScaledDotProductAttention(nn.Module):
    def __init__(self, feature_dim: int, query_dim: int, attention_dim: int):
        super().__init__()
        self.K = nn.Linear(feature_dim, attention_dim, bias=False)
        self.Q = nn.Linear(query_dim, attention_dim, bias=True)
        self.attention_scalar = math.sqrt(float(attention_dim))

    def initialize(self):
        nn.init.xavier_uniform_(self.K.weight)
        nn.init.xavier_uniform_(self.Q.weight)
        nn.init.zeros_(self.Q.bias)

    def forward(self, feature, query, mask=None):  # pragma: no cover - never called on the product path
        raise RuntimeError("ScaledDotProductAttention is evaluated inside digat_amd's HIP kernels; "
                           "call DIGAT.compute_*_graph_context instead")
