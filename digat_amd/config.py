"""Flags: the counterpart of the reference's ``config.py`` for the path this repo owns.

Same flag names, defaults and dataset overrides (config.py:14-75): MIND-small forces dropout 0.2 / 16 epochs,
MIND-large 0.1 / 7; ``news_graph_size = 1 + M + M(M-1) + ...``.  Differences, all forced by the environment:
real MIND cannot be downloaded, so the corpus is synthetic (``--synthetic_news``, ``--synthetic_impressions``);
``--local_rank`` also accepts torch >= 2.0's ``--local-rank`` spelling and the ``LOCAL_RANK`` variable.
"""
from __future__ import annotations

import argparse
import os
import random

import numpy as np
import torch

from .synthetic import news_graph_size


class Config:
    def __init__(self, argv=None):
        p = argparse.ArgumentParser(description='DIGAT (MI355X HIP path) experiments')
        p.add_argument('--mode', default='train', choices=['train', 'dev', 'test'])
        p.add_argument('--news_encoder', default='MSA', choices=['MSA', 'CNN'])
        p.add_argument('--graph_encoder', default='DIGAT',
                       choices=['DIGAT', 'wo_SA', 'Seq_SA', 'wo_interaction', 'news_graph_wo_inter', 'user_graph_wo_inter'])
        p.add_argument('--seed', type=int, default=0)
        p.add_argument('--local_rank', '--local-rank', type=int, default=int(os.environ.get('LOCAL_RANK', -1)))
        p.add_argument('--dataset', default='MIND-small', choices=['MIND-small', 'MIND-large'])
        p.add_argument('--negative_sample_num', type=int, default=4)
        p.add_argument('--max_history_num', type=int, default=50)
        p.add_argument('--epoch', type=int, default=16)
        p.add_argument('--batch_size', type=int, default=64)
        p.add_argument('--lr', type=float, default=1e-4)
        p.add_argument('--weight_decay', type=float, default=0)
        p.add_argument('--gradient_clip_norm', type=float, default=1)
        p.add_argument('--early_stopping_epoch', type=int, default=5)
        p.add_argument('--dev_criterion', default='avg', choices=['auc', 'mrr', 'ndcg5', 'ndcg10', 'avg'])
        p.add_argument('--train_precision', default='fp32', choices=['fp32', 'bf16'],
                       help='bf16: bf16 matrix-core operands for the large training GEMMs (fp32 master weights / accumulation)')
        p.add_argument('--dropout_rate', type=float, default=0.2)
        p.add_argument('--graph_depth', type=int, default=3)
        p.add_argument('--SAG_hops', type=int, default=2)
        p.add_argument('--SAG_neighbors', type=int, default=5)
        p.add_argument('--news_embedding_dim', type=int, default=400, help='MSA: 16 heads x 25')
        p.add_argument('--synthetic_news', type=int, default=8192)
        p.add_argument('--synthetic_impressions', type=int, default=2048)
        p.add_argument('--max_steps', type=int, default=0, help='stop training after this many steps (0 = all epochs)')
        a = p.parse_args(argv)
        self.attribute_dict = dict(vars(a))
        for k, v in self.attribute_dict.items():
            setattr(self, k, v)
        if self.dataset == 'MIND-small':
            self.dropout_rate, self.epoch, self.category_num = 0.2, 16, 17
        else:
            self.dropout_rate, self.epoch, self.category_num = 0.1, 7, 18
        self.news_graph_size = news_graph_size(self.SAG_neighbors, self.SAG_hops)
        self.max_title_length = 1                                   # synthetic "titles" are news ids

    def set_device(self):
        assert torch.cuda.is_available(), 'GPU is not available'
        if self.local_rank == -1:
            torch.cuda.set_device(0)
        else:
            import datetime
            import torch.distributed as dist
            torch.cuda.set_device(self.local_rank)
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group(backend='nccl', timeout=datetime.timedelta(0, 43200))   # RCCL on ROCm
        torch.manual_seed(self.seed)
        random.seed(self.seed)
        np.random.seed(self.seed)
