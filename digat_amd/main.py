"""Entry point: ``python -m digat_amd.main --mode={train,dev,test} --graph_encoder=DIGAT ...``

The counterpart of the reference's ``main.py`` on a synthetic MIND-shaped corpus: ``train`` runs the
``Trainer`` (DDP when launched with one process per GPU) and then scores the dev rows; ``dev`` / ``test``
score them and print AUC / MRR / nDCG@5 / nDCG@10 and the inference time (main.py:66-72).
"""
from __future__ import annotations

import time

import torch

from . import synthetic, util
from .config import Config
from .model import Model, PrecomputedNewsEncoder
from .trainer import SyntheticTrainSet, Trainer


def main(argv=None):
    config = Config(argv)
    config.set_device()
    dev = torch.device('cuda', torch.cuda.current_device())
    spec = synthetic.SynthSpec(news_num=config.synthetic_news, sag_neighbors=config.SAG_neighbors, sag_hops=config.SAG_hops,
                               max_history_num=config.max_history_num, category_num=config.category_num,
                               embedding_dim=config.news_embedding_dim, impressions=config.synthetic_impressions,
                               seed=config.seed)
    corpus = synthetic.make_corpus(spec)
    model = Model(config, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(corpus.news_embedding),
                                                              trainable=config.mode == 'train'))
    model.initialize()
    model = model.to(dev)
    dc = util.DeviceCorpus.from_numpy(corpus, dev)
    if config.mode == 'train':
        trainer = Trainer(model, config, dc, SyntheticTrainSet(corpus, config.negative_sample_num, config.seed),
                          local_rank=config.local_rank, dev_labels=corpus.row_label)
        trainer.train(max_steps=config.max_steps or None, log_every=50)
        if config.local_rank != -1:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
        if not trainer.is_main_rank:
            return
    if config.local_rank in (-1, 0):
        start = time.time()
        dc.news_embedding = model.news_encoder.table.detach()
        scores, metrics = util.compute_scores(model, dc, config.batch_size * 16, labels=corpus.row_label)
        print('AUC : %.4f\nMRR : %.4f\nnDCG@5 : %.4f\nnDCG@10 : %.4f' % metrics)
        print('Inference time : %.1fs' % (time.time() - start))


if __name__ == '__main__':
    main()
