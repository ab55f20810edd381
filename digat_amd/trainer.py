"""Training harness: the counterpart of the reference's ``trainer.py`` for the path this repo owns.

Same step as trainer.py:71-105 — Adam(lr) with the ``['.bias', 'embed', 'graph_encoder.']`` no-decay
split (:25-30), loss = mean(-log_softmax(logits)[:, 0]) (:100), clip_grad_norm_ (:103-104), lr / 10 at
epoch ``E - ((E-1)//10 + 1) + 1`` (:32, :81-82), DistributedDataParallel + DistributedSampler when
launched with one process per GPU (:19, :78-80; backend "nccl" is RCCL on ROCm) — around
``Model.forward``, whose graph encoder runs on the HIP kernels forward and backward.

Real MIND is not reachable from this environment, so the data side is ``SyntheticTrainSet``: behaviours
with one clicked and ``negative_sample_num`` sampled non-clicked candidates (MIND_dataset.py:26-47),
indexed into the device-resident synthetic corpus (no DataLoader workers: a batch is a few index_selects).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from . import util
from .evaluate import AvgMetric


def training_loss(logits: torch.Tensor) -> torch.Tensor:
    """trainer.py:100 — the clicked candidate is column 0.  On the GPU: the loss and its gradient from one launch (training.ClickLoss)."""
    if logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and logits.shape[0] > 0:
        from .training import ClickLoss
        return ClickLoss.apply(logits)
    return (-torch.log_softmax(logits, dim=1).select(1, 0)).mean()


def lr_decay_epoch(epochs: int) -> int:
    """First epoch that runs at lr/10 (trainer.py:32,81): E - ((E-1)//10 + 1) + 1."""
    return epochs - ((epochs - 1) // 10 + 1) + 1


def parameter_groups(model: nn.Module, weight_decay: float):
    no_decay = ['.bias', 'embed', 'graph_encoder.']                     # trainer.py:25
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    return [
        {'params': [p for n, p in named if not any(nd in n.lower() for nd in no_decay)], 'weight_decay': weight_decay},
        {'params': [p for n, p in named if any(nd in n.lower() for nd in no_decay)], 'weight_decay': 0.0},
    ]


class SyntheticTrainSet:
    """Training behaviours on a synthetic corpus: (impression, clicked news, non-clicked pool)."""

    def __init__(self, corpus, negative_sample_num: int = 4, seed: int = 0):
        self.neg = negative_sample_num
        self.rng = np.random.default_rng(seed)
        imp, cand, lab = corpus.row_impression, corpus.row_candidate, corpus.row_label
        bounds = np.r_[0, np.flatnonzero(np.diff(imp)) + 1, len(imp)]
        self.behaviors = []                                              # (impression, clicked, [non-clicked])
        for s, e in zip(bounds[:-1], bounds[1:]):
            pos, negs = cand[s:e][lab[s:e] == 1], cand[s:e][lab[s:e] == 0]
            if len(negs) == 0:
                continue
            for c in pos:
                self.behaviors.append((int(imp[s]), int(c), negs))
        self.samples = np.zeros((len(self.behaviors), 1 + self.neg), dtype=np.int64)
        self.impression = np.array([b[0] for b in self.behaviors], dtype=np.int64)

    def negative_sampling(self):
        """MIND_dataset.py:26-47: without replacement when the pool is large enough, cyclic otherwise."""
        for i, (_, click, negs) in enumerate(self.behaviors):
            self.samples[i, 0] = click
            if len(negs) <= self.neg:
                self.samples[i, 1:] = negs[np.arange(self.neg) % len(negs)]
            else:
                self.samples[i, 1:] = self.rng.choice(negs, size=self.neg, replace=False)

    def __len__(self):
        return len(self.behaviors)


class Trainer:
    def __init__(self, model: nn.Module, config, dc: "util.DeviceCorpus", train_set: SyntheticTrainSet,
                 local_rank: int = -1, dev_labels: Optional[np.ndarray] = None, model_dir: Optional[str] = None):
        self.local_rank = local_rank
        self.is_main_rank = local_rank in (-1, 0)
        if local_rank == -1:
            self.model = model
        else:
            from torch.nn.parallel import DistributedDataParallel as DDP
            self.model = DDP(model, device_ids=[local_rank], output_device=local_rank)
        # BASELINE configs[4]: --train_precision bf16 = bf16 matrix-core operands for the large training GEMMs (fp32 master
        # weights, activations, accumulation and weight gradients); default fp32-grade
        self.train_precision = getattr(config, "train_precision", "fp32")
        if self.train_precision not in ("fp32", "bf16"):
            raise ValueError("train_precision must be 'fp32' or 'bf16'")
        from . import _lib
        _lib.lib().digat_set_train_precision(1 if self.train_precision == "bf16" else 0)
        self.epochs = config.epoch
        self.batch_size = config.batch_size
        # the reference's Adam (trainer.py:30), as ONE fused launch per parameter group on the GPU: the default multi-tensor form walks
        # every parameter ten times (the synthetic news table alone is 104 MB: 0.63 ms of a 7.1 ms step, 20 launches); the update is
        # the same arithmetic per element
        groups = parameter_groups(self.model, getattr(config, "weight_decay", 0.0))
        on_gpu = all(p.is_cuda for g in groups for p in g["params"])
        # round 6: on the GPU the clipping and the update are ONE pass over (p, g, m, v) + two small launches for the norm (optim.ClipAdam:
        # 3 launches where clip_grad_norm_ + the fused Adam are 15); config.optimizer_impl = "torch" keeps torch's pair
        if on_gpu and getattr(config, "optimizer_impl", "hip") == "hip" and all(p.dtype == torch.float32 for g in groups for p in g["params"]):
            from .optim import ClipAdam
            self.optimizer = ClipAdam(groups, lr=config.lr)
        else:
            self.optimizer = optim.Adam(groups, lr=config.lr, **({"fused": True} if on_gpu and getattr(config, "fused_adam", True) else {}))
        self.gradient_clip_norm = getattr(config, "gradient_clip_norm", 1.0)
        self.decay_epoch = lr_decay_epoch(self.epochs)
        self.dc, self.train_set = dc, train_set
        self.losses = []
        # per-epoch dev evaluation and model selection on the main rank (trainer.py:52-69, :109-172)
        self.dev_labels = dev_labels
        self.dev_criterion = getattr(config, "dev_criterion", "avg")
        self.early_stopping_epoch = getattr(config, "early_stopping_epoch", 5)
        self.model_dir = model_dir
        self.auc, self.mrr, self.ndcg5, self.ndcg10 = [], [], [], []
        self.best_dev_epoch, self.best_dev = 0, None
        self.epoch_not_increase = 0
        self.best_state = None
        from . import util
        # Eq. 8 of the user graph in training: the corpus is known here, so the entry-wise / all-pairs choice the library would otherwise
        # make on the device per batch (a decision launch, and the launches of the side not taken returning at once: ~40 us per layer)
        # is made once, on the host, as util.prepare_news_side does for scoring; an explicit encoder.user_xattn_mode wins
        enc = getattr(model, "graph_encoder", None)
        if enc is not None and hasattr(enc, "corpus_xattn_hint") and "user" not in enc.corpus_xattn_hint and dc is not None \
                and getattr(dc, "user_graph", None) is not None and dc.user_graph.numel() > 0:
            per_node = float(dc.user_graph.sum(dtype=torch.float64) / (dc.user_graph.shape[0] * dc.user_graph.shape[1]))
            enc.corpus_xattn_hint = dict(enc.corpus_xattn_hint, user="sparse" if per_node <= util.SPARSE_ENTRIES_PER_NODE else "dense")
        util.freeze_host_heap()           # the corpus's host-side structures: out of the garbage collector's walks (util.freeze_host_heap)

    def lr_decay(self):
        for group in self.optimizer.param_groups:
            group['lr'] = group['lr'] / 10

    def batches(self, epoch: int):
        n = len(self.train_set)
        order = np.random.default_rng(1000 + epoch).permutation(n)
        if self.local_rank != -1:                                        # DistributedSampler: strided shards
            import torch.distributed as dist
            world, rank = dist.get_world_size(), dist.get_rank()
            total = (n + world - 1) // world * world
            order = np.r_[order, order[: total - n]][rank::world]
        for s in range(0, len(order), self.batch_size):
            yield order[s:s + self.batch_size]

    def gather(self, idx: np.ndarray):
        """The 9 inputs of Model.forward (trainer.py:88-96) for the behaviours ``idx`` — news ids stand in for
        title text (the synthetic news 'encoder' is an embedding table)."""
        dc, dev = self.dc, self.dc.news_embedding.device
        # the step's indices travel as ONE pinned, asynchronous copy: a pageable .to(device) is a blocking copy — the host waits there
        # until the device has drained the previous step, every step (round 5: 1.5 ms of host time per 7 ms step)
        samples = self.train_set.samples[idx]                                      # [B, 1+neg]
        B, K = samples.shape
        packed = torch.from_numpy(np.concatenate([self.train_set.impression[idx], samples.reshape(-1)]).astype(np.int64, copy=False))
        if dev.type == "cuda":
            packed = packed.pin_memory().to(dev, non_blocking=True)
        imp, news = packed[:B], packed[B:].view(B, K)
        node_ids = dc.news_node_ID.index_select(0, news.flatten()).view(B, K, -1, 1)  # [B,K,N,1]
        hist = dc.history.index_select(0, imp).unsqueeze(2)                        # [B,H,1]
        if dc.title_text is not None:          # a text news encoder (MSA): the titles of the history and of the SAG nodes, as
            Lw = dc.title_text.shape[1]        # MIND_dataset.py hands them to model.forward (trainer.py:88-96)
            def titles(ids):
                flat = ids.reshape(-1)
                return (dc.title_text.index_select(0, flat).view(*ids.shape[:-1], Lw),
                        dc.title_mask.index_select(0, flat).view(*ids.shape[:-1], Lw))
            ht, hm = titles(hist)
            nt, nm = titles(node_ids)
            return (ht, hm, dc.user_graph.index_select(0, imp), dc.user_category_mask.index_select(0, imp),
                    dc.user_category_indices.index_select(0, imp), nt, nm,
                    dc.news_graph.index_select(0, news.flatten()).view(B, K, *dc.news_graph.shape[1:]),
                    dc.news_graph_mask.index_select(0, news.flatten()).view(B, K, -1))
        return (hist, torch.ones_like(hist, dtype=torch.bool), dc.user_graph.index_select(0, imp),
                dc.user_category_mask.index_select(0, imp), dc.user_category_indices.index_select(0, imp),
                node_ids, torch.ones_like(node_ids, dtype=torch.bool),
                dc.news_graph.index_select(0, news.flatten()).view(B, K, *dc.news_graph.shape[1:]),
                dc.news_graph_mask.index_select(0, news.flatten()).view(B, K, -1))

    def train_step(self, idx: np.ndarray, read_loss: bool = True):
        """One optimisation step (trainer.py:98-105).  ``read_loss`` (default): return the loss as a Python float, as the reference's
        loop reads it every step (``loss.item()``: a host synchronisation per step, which serialises the host's enqueue of the next
        step with the device's work on this one); False: return the loss TENSOR (detached, on the device) and read nothing — the
        caller sums on the device and reads once per epoch (``Trainer.train``), and the step after this one is enqueued while the
        device is still busy."""
        logits = self.model(*self.gather(idx))                          # [B, 1+neg]
        loss = training_loss(logits)
        self.optimizer.zero_grad()
        loss.backward()
        if hasattr(self.optimizer, "_chunk"):                           # optim.ClipAdam: the clipping coefficient is applied inside the update
            self.optimizer.step(max_norm=self.gradient_clip_norm)
        else:
            if self.gradient_clip_norm > 0:
                nn.utils.clip_grad_norm_(self.model.parameters(), self.gradient_clip_norm)
            self.optimizer.step()
        return float(loss.item()) if read_loss else loss.detach()

    def _criterion(self, metrics):
        """trainer.py:121-165: the value the best epoch is chosen by (``>=`` keeps the later of two equal epochs)."""
        auc, mrr, ndcg5, ndcg10 = metrics
        return {"auc": auc, "mrr": mrr, "ndcg5": ndcg5, "ndcg10": ndcg10}.get(self.dev_criterion, AvgMetric(*metrics))

    def dev_epoch(self, e: int) -> bool:
        """Main rank, end of epoch e (trainer.py:109-172): score the dev rows through the HIP inference path, track the best
        epoch, keep / save its state dict.  Returns True when early stopping says stop."""
        net = self.model.module if hasattr(self.model, "module") else self.model
        was_training = net.training
        metrics = evaluate_dev(net, self.dc, self.dev_labels, self.batch_size * 16, as_tuple=True)
        net.train(was_training)
        for acc, v in zip((self.auc, self.mrr, self.ndcg5, self.ndcg10), metrics):
            acc.append(v)
        print('Epoch %d : dev done\nDev criterions' % e)
        print('AUC = {:.4f}\nMRR = {:.4f}\nnDCG@5  = {:.4f}\nnDCG@10 = {:.4f}'.format(*metrics), flush=True)
        value = self._criterion(metrics)
        if self.best_dev is None or value >= self.best_dev:
            self.best_dev, self.best_dev_epoch, self.epoch_not_increase = value, e, 0
            self.best_state = {k: v.detach().clone() for k, v in net.state_dict().items()}
            if self.model_dir is not None:                               # trainer.py:169-170
                import os
                os.makedirs(self.model_dir, exist_ok=True)
                torch.save({net.model_name: net.state_dict()}, os.path.join(self.model_dir, f"{net.model_name}-{e}"))
        else:
            self.epoch_not_increase += 1
        print('Best epoch :', self.best_dev_epoch, flush=True)
        return self.epoch_not_increase > self.early_stopping_epoch

    def train(self, max_steps: Optional[int] = None, log_every: int = 0):
        step = 0
        distributed = self.local_rank != -1
        for e in range(1, self.epochs + 1):
            self.train_set.negative_sampling()
            if e == self.decay_epoch:
                self.lr_decay()
            self.model.train()
            # the epoch's loss is summed ON THE DEVICE (float64: the sum of the same fp32 losses the reference adds up as Python floats,
            # trainer.py:105) and read once at the end: the reference's per-step loss.item() drains the device every step
            epoch_loss_dev, nb = torch.zeros((), dtype=torch.float64, device=self.dc.news_embedding.device), 0
            for idx in self.batches(e):
                log_now = bool(log_every and self.is_main_rank and (step + 1) % log_every == 0)
                loss = self.train_step(idx, read_loss=False)
                epoch_loss_dev += loss.double()
                nb += 1
                step += 1
                if log_now:
                    print(f"epoch {e} step {step} loss {float(loss.item()):.4f}", flush=True)
                if max_steps is not None and step >= max_steps:
                    break
            epoch_loss = float(epoch_loss_dev.item())
            self.losses.append(epoch_loss / max(nb, 1))
            if self.is_main_rank:
                print(f"Epoch {e} : train done\nloss = {self.losses[-1]}", flush=True)
            stop = max_steps is not None and step >= max_steps
            if self.dev_labels is not None and self.is_main_rank:
                stop = self.dev_epoch(e) or stop
            if distributed:
                # the reference breaks out on the main rank only (trainer.py:171-172) and leaves the others waiting in the next
                # all-reduce until the 12 h timeout; here the decision is broadcast
                import torch.distributed as dist
                flag = torch.tensor([int(stop)], device=self.dc.news_embedding.device)
                dist.broadcast(flag, src=0)
                stop = bool(flag.item())
            if stop:
                break
        if self.is_main_rank and self.best_state is not None:           # trainer.py:188: the best epoch's weights are the result
            net = self.model.module if hasattr(self.model, "module") else self.model
            net.load_state_dict(self.best_state)
            if self.model_dir is not None:
                import os
                torch.save({net.model_name: net.state_dict()}, os.path.join(self.model_dir, net.model_name))
        return self.losses


def evaluate_dev(model, dc, labels, batch_size: int, as_tuple: bool = False):
    """Per-epoch dev evaluation (trainer.py:109-120): AUC / MRR / nDCG through the HIP inference path.  The per-news caches
    (c_n0, layer-0 projection tables) are recomputed when the weights have moved on (``util.weights_key``)."""
    net = model.module if hasattr(model, "module") else model
    if hasattr(net.news_encoder, "table"):
        dc.news_embedding = net.news_encoder.table.detach()             # a trainable table is the news-representation cache
    scores, metrics = util.compute_scores(net, dc, batch_size, labels=labels)
    return tuple(metrics) if as_tuple else AvgMetric(*metrics)
