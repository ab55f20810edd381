"""CPU suite for the trainer's host logic (no kernels): lr schedule, parameter groups, negative sampling,
loss — against the reference's rules (trainer.py:25-32,81,100; MIND_dataset.py:26-47)."""
import types

import torch

from digat_amd import synthetic
from digat_amd.trainer import SyntheticTrainSet, lr_decay_epoch, parameter_groups, training_loss


def test_lr_decay_epoch_matches_reference_rule():
    # trainer.py:32,81: decay when e == epoch - ((epoch-1)//10 + 1) + 1
    assert lr_decay_epoch(16) == 15          # MIND-small: 16 epochs -> lr/10 from epoch 15
    assert lr_decay_epoch(7) == 7            # MIND-large: 7 epochs -> last epoch
    assert lr_decay_epoch(1) == 1


def test_parameter_groups_follow_the_no_decay_rule():
    from digat_amd.model import Model, PrecomputedNewsEncoder
    cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=4, max_history_num=10,
                                category_num=5, graph_depth=1, dropout_rate=0.2)
    model = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.zeros(16, 64), trainable=True))
    decay, no_decay = parameter_groups(model, 0.01)
    assert decay["params"] == []                               # everything here is graph_encoder.* or an embedding
    assert len(no_decay["params"]) == len(list(model.parameters()))
    assert no_decay["weight_decay"] == 0.0 and decay["weight_decay"] == 0.01


def test_negative_sampling_and_loss():
    spec = synthetic.SynthSpec(news_num=256, sag_neighbors=3, sag_hops=1, max_history_num=10, category_num=5,
                               embedding_dim=16, impressions=40, mean_candidates=8.0, max_candidates=20, seed=3)
    corpus = synthetic.make_corpus(spec)
    ts = SyntheticTrainSet(corpus, negative_sample_num=4, seed=0)
    ts.negative_sampling()
    assert 0 < len(ts) <= int((corpus.row_label == 1).sum())       # one behaviour per clicked candidate
    for i, (imp, click, negs) in enumerate(ts.behaviors):
        assert ts.samples[i, 0] == click
        assert set(ts.samples[i, 1:].tolist()) <= set(negs.tolist())
        if len(set(negs.tolist())) == len(negs) > 4:
            assert len(set(ts.samples[i, 1:].tolist())) == 4       # positions drawn without replacement
    logits = torch.tensor([[2.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    want = (-torch.log_softmax(logits, dim=1)[:, 0]).mean()
    assert torch.allclose(training_loss(logits), want)


def test_per_epoch_dev_selection_and_early_stopping(monkeypatch):
    """trainer.py:109-172 on the host: the best epoch by the dev criterion (`>=`: the later of two equal epochs), early
    stopping after `early_stopping_epoch` epochs without improvement, the best epoch's weights restored at the end."""
    import numpy as np
    from digat_amd import trainer as T
    from digat_amd.model import Model, PrecomputedNewsEncoder
    cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=4, max_history_num=10,
                                category_num=5, graph_depth=1, dropout_rate=0.2, epoch=10, batch_size=4, lr=1e-3,
                                early_stopping_epoch=2, dev_criterion="auc")
    model = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.zeros(16, 64), trainable=True))
    spec = synthetic.SynthSpec(news_num=64, sag_neighbors=3, sag_hops=1, max_history_num=10, category_num=5,
                               embedding_dim=64, impressions=12, mean_candidates=6.0, max_candidates=12, seed=4)
    corpus = synthetic.make_corpus(spec)
    dc = types.SimpleNamespace(news_embedding=torch.zeros(1))
    tr = T.Trainer(model, cfg, dc, T.SyntheticTrainSet(corpus, 4, 0), dev_labels=corpus.row_label)
    aucs = iter([0.50, 0.60, 0.60, 0.55, 0.58, 0.59, 0.70])           # epochs 1..: best = 3 (>=), stop after epoch 6
    marks = []

    def fake_step(idx, read_loss=True):
        with torch.no_grad():
            model.graph_encoder.topic_node_embedding.add_(1.0)          # the weights move every step
        return 0.0 if read_loss else torch.zeros(())                    # Trainer.train sums the loss tensors, one read per epoch

    def fake_dev(net, dc_, labels, bs, as_tuple=False):
        marks.append(float(net.graph_encoder.topic_node_embedding[0, 0]))
        return (next(aucs), 0.3, 0.3, 0.3)
    monkeypatch.setattr(tr, "train_step", fake_step)
    monkeypatch.setattr(tr, "batches", lambda e: iter([np.arange(2)]))
    monkeypatch.setattr(T, "evaluate_dev", fake_dev)
    tr.train()
    assert tr.best_dev_epoch == 3 and tr.auc == [0.50, 0.60, 0.60, 0.55, 0.58, 0.59]
    assert len(tr.losses) == 6                                          # epochs 4, 5, 6 did not improve: 3 > 2 -> stop
    assert float(model.graph_encoder.topic_node_embedding[0, 0]) == marks[2] == 3.0     # epoch 3's weights are the result
