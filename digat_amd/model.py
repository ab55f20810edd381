"""Model assembly: the counterpart of the reference's ``model.py`` (plugin selection + glue).

``config.graph_encoder == 'DIGAT'`` selects the HIP plugin (model.py:18-19); ``forward`` (9 tensors,
training) and ``inference`` (8 tensors, dev/test) keep the reference's signatures and reshapes
(model.py:54-90).  The five ablation encoders run on the same kernels in eval mode /
inference; unknown names raise the reference's ``'<name> is not implemented'`` exception.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib, graphEncoders, newsEncoders


def _row_logits(news_rep, user_rep, logits):
    """model.py:75,90: logits = sum_d(user_ctx * news_ctx), one wave per row (digat_row_logits)."""
    X = _lib.ext()
    if X is not None:
        X.row_logits(news_rep, user_rep, logits)
        return
    B, d = news_rep.shape
    _lib.check(_lib.lib().digat_row_logits(news_rep.data_ptr(), user_rep.data_ptr(), logits.data_ptr(), B, d, _lib.stream_ptr()),
               "digat_row_logits")


class Model(nn.Module):
    def __init__(self, config, news_encoder: nn.Module = None):
        super().__init__()
        if news_encoder is not None:
            self.news_encoder = news_encoder            # any module exposing .news_embedding_dim
        elif config.news_encoder == 'CNN':
            self.news_encoder = newsEncoders.CNN(config)
        elif config.news_encoder == 'MSA':
            self.news_encoder = newsEncoders.MSA(config)
        else:
            raise Exception(config.news_encoder + ' is not implemented')
        variants = {'DIGAT': graphEncoders.DIGAT, 'wo_SA': graphEncoders.wo_SA, 'Seq_SA': graphEncoders.Seq_SA,
                    'wo_interaction': graphEncoders.wo_interaction, 'news_graph_wo_inter': graphEncoders.News_graph_wo_inter,
                    'user_graph_wo_inter': graphEncoders.User_graph_wo_inter}       # model.py:18-31
        if config.graph_encoder in variants:
            self.graph_encoder = variants[config.graph_encoder](config, self.news_encoder.news_embedding_dim)
        else:
            raise Exception(config.graph_encoder + ' is not implemented')
        self.model_name = str(getattr(config, 'news_encoder', 'MSA')) + '-' + config.graph_encoder
        self.max_title_length = getattr(config, 'max_title_length', 32) if news_encoder is None else getattr(config, 'max_title_length', 1)
        self.max_history_num = config.max_history_num
        self.category_num = config.category_num + 1
        self.news_embedding_dim = self.news_encoder.news_embedding_dim
        self.representation_dim = self.news_embedding_dim
        self.news_graph_size = config.news_graph_size
        self.user_graph_size = config.max_history_num + config.category_num

    def initialize(self):
        if hasattr(self.news_encoder, "initialize"):
            self.news_encoder.initialize()
        self.graph_encoder.initialize()

    # model.py:54-77
    def forward(self, user_title_text, user_title_mask, user_graph, user_category_mask, user_category_indices,
                news_title_text, news_title_mask, news_graph, news_graph_mask):
        batch_size, news_num = news_graph.size(0), news_graph.size(1)
        bn = batch_size * news_num
        news_title_text = news_title_text.view([bn, self.news_graph_size, self.max_title_length])
        news_title_mask = news_title_mask.view([bn, self.news_graph_size, self.max_title_length])
        news_graph = news_graph.view([bn, self.news_graph_size, self.news_graph_size])
        news_graph_mask = news_graph_mask.view([bn, self.news_graph_size])

        def per_candidate(t):
            return t.unsqueeze(1).expand(-1, news_num, *t.shape[1:]).contiguous().view([bn, *t.shape[1:]])

        user_graph = per_candidate(user_graph)
        user_category_mask = per_candidate(user_category_mask)
        user_category_indices = per_candidate(user_category_indices)
        if hasattr(self.news_encoder, "encode_pair"):      # a table-backed encoder looks both id lists up at once (one table gradient)
            candidate_news_embedding, user_news_embedding = self.news_encoder.encode_pair(news_title_text, user_title_text)
            user_news_embedding = per_candidate(user_news_embedding)
        else:
            candidate_news_embedding = self.news_encoder(news_title_text, news_title_mask)
            user_news_embedding = per_candidate(self.news_encoder(user_title_text, user_title_mask))
        news_rep, user_rep = self.graph_encoder(candidate_news_embedding, news_graph, news_graph_mask,
                                                user_news_embedding, user_graph, user_category_mask,
                                                user_category_indices)
        if news_rep.is_cuda and news_rep.dtype == torch.float32 and news_rep.dim() == 2:
            from .training import RowLogits          # the same dot products, one launch each way (digat_row_logits / digat_row_logits_bwd)
            return RowLogits.apply(news_rep, user_rep).view([batch_size, news_num])
        news_rep = news_rep.view([batch_size, news_num, self.representation_dim])
        user_rep = user_rep.view([batch_size, news_num, self.representation_dim])
        return (user_rep * news_rep).sum(dim=2)

    # model.py:87-90
    def inference(self, user_news_embedding, user_graph, user_category_mask, user_category_indices,
                  candidate_news_embedding, news_graph, news_graph_mask, c_n0):
        news_rep, user_rep = self.graph_encoder.inference(candidate_news_embedding, news_graph, news_graph_mask,
                                                          user_news_embedding, user_graph, user_category_mask,
                                                          user_category_indices, c_n0)
        B, d = news_rep.shape
        logits = torch.empty(B, dtype=torch.float32, device=news_rep.device)
        if B:
            _row_logits(news_rep, user_rep, logits)
        return logits


    def inference_grouped(self, user_news_embedding, user_graph, user_category_mask, user_category_indices, row_group,
                          candidate_news_embedding, news_graph, news_graph_mask, c_n0, news_hpq0=None, hist_hpq0=None,
                          topic_hpq0=None, ctxq0=None, news_index=None):
        """``inference`` with the user tensors given once per impression ([G,...]) + ``row_group`` [B]; optional rows of the
        per-news layer-0 projection tables (``DIGAT.project_news_layer0`` / ``project_user_layer0``)."""
        kw = {k: v for k, v in (("news_hpq0", news_hpq0), ("hist_hpq0", hist_hpq0), ("topic_hpq0", topic_hpq0), ("ctxq0", ctxq0),
                                ("news_index", news_index)) if v is not None}
        news_rep, user_rep = self.graph_encoder.inference_grouped(candidate_news_embedding, news_graph, news_graph_mask,
                                                                  user_news_embedding, user_graph, user_category_mask,
                                                                  user_category_indices, row_group, c_n0, **kw)
        B, d = news_rep.shape
        logits = torch.empty(B, dtype=torch.float32, device=news_rep.device)
        if B:
            _row_logits(news_rep, user_rep, logits)
        return logits


class PrecomputedNewsEncoder(nn.Module):
    """Stand-in producer for synthetic runs: news id -> embedding row (no title text exists).
    Takes ids shaped [B, n] or [B, n, 1] (a "title" of one token = the news id; the mask argument is ignored)
    and returns [B, n, d].  ``trainable=True`` makes the table a parameter (named ``embed_table`` so that the
    trainer's 'embed' no-decay rule applies, trainer.py:25)."""

    def __init__(self, table: torch.Tensor, trainable: bool = False):
        super().__init__()
        if trainable:
            self.embed_table = nn.Parameter(table.clone())
        else:
            self.register_buffer("embed_table", table)
        self.news_embedding_dim = int(table.shape[1])

    @property
    def table(self):
        return self.embed_table

    @staticmethod
    def _ids(news_ids):
        return news_ids.squeeze(-1) if news_ids.dim() >= 3 and news_ids.shape[-1] == 1 else news_ids

    def encode_pair(self, candidate_ids, history_ids):
        """Model.forward's two lookups (model.py:72-73) in one autograd node: on the GPU with a trainable table the gradient of both is
        ONE dense table gradient from one launch (training.TableLookup2); otherwise two plain lookups."""
        a, b = self._ids(candidate_ids), self._ids(history_ids)
        if self.embed_table.requires_grad and torch.is_grad_enabled() and self.embed_table.is_cuda and self.embed_table.shape[1] % 4 == 0 \
                and self.embed_table.shape[1] <= 1024 and self.embed_table.dtype == torch.float32:
            from .training import TableLookup2
            return TableLookup2.apply(self.embed_table, a, b)
        return self.forward(a), self.forward(b)

    def forward(self, news_ids, _mask=None):
        if news_ids.dim() == 3 and news_ids.shape[2] == 1:
            news_ids = news_ids.squeeze(2)
        # F.embedding, not table[ids]: its backward is one dense-embedding kernel, where advanced indexing goes through
        # index_put_(accumulate=True) — sort-based, 0.72 ms per call at the training shapes (1.45 of a 11.9 ms step)
        return torch.nn.functional.embedding(news_ids.long(), self.embed_table)
