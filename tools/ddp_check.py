#!/usr/bin/env python3
"""DDP gradient check (run under torch.distributed.run, one process per rank; tests/test_hip_ddp.py).

Every rank wraps the same Model in DistributedDataParallel (trainer.py:19), takes ITS share of one batch of training
behaviours through Model.forward / loss / backward (the HIP forward and backward kernels, launched from autograd
Functions on torch's current stream), and rank 0 compares the all-reduced (averaged) gradients with the gradients a
single process gets from the whole batch.  Dropout is 0 so that both are deterministic.

DIGAT_BENCH_TEST_SHARED_GPU=1: every rank on cuda:0, collectives over gloo (a one-GPU box); otherwise one GPU per rank
over RCCL ("nccl").  Prints one JSON line on rank 0; exit code 1 on mismatch.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from digat_amd import synthetic, util  # noqa: E402
from digat_amd.model import Model, PrecomputedNewsEncoder  # noqa: E402
from digat_amd.trainer import SyntheticTrainSet, Trainer, training_loss  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    shared = os.environ.get("DIGAT_BENCH_TEST_SHARED_GPU") == "1"
    index = 0 if shared else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(index)
    dev = torch.device("cuda", index)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo" if shared else "nccl")
    depth = int(os.environ.get("DDP_CHECK_DEPTH", "2"))
    per_rank = int(os.environ.get("DDP_CHECK_BEHAVIOURS", "8"))
    spec = synthetic.SynthSpec(news_num=1024, sag_neighbors=3, sag_hops=2, impressions=96, seed=7)
    corpus = synthetic.make_corpus(spec)                      # the same corpus on every rank
    cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=spec.news_graph_size,
                                max_history_num=spec.max_history_num, category_num=spec.category_num, graph_depth=depth,
                                dropout_rate=0.0, epoch=1, batch_size=per_rank, lr=1e-4, weight_decay=0.0, gradient_clip_norm=0.0)

    def make_model():
        torch.manual_seed(0)
        m = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(corpus.news_embedding), trainable=True))
        m.initialize()
        with torch.no_grad():
            m.graph_encoder.topic_node_embedding.normal_(0, 0.02)
        return m.to(dev)

    dc = util.DeviceCorpus.from_numpy(corpus, dev)
    ts = SyntheticTrainSet(corpus, 4, seed=0)
    ts.negative_sampling()                                    # same seed -> the same negatives on every rank
    tr = Trainer(make_model(), cfg, dc, ts, local_rank=index)
    tr.model.train()
    union = np.arange(world * per_rank)
    mine = union[rank * per_rank:(rank + 1) * per_rank]
    loss = training_loss(tr.model(*tr.gather(mine)))
    tr.optimizer.zero_grad()
    loss.backward()                                           # DDP all-reduces (averages) the gradients here
    torch.cuda.synchronize()
    ok, worst = True, {}
    if rank == 0:
        ref = make_model().train()
        ref_loss = training_loss(ref(*tr.gather(union)))
        ref_loss.backward()
        torch.cuda.synchronize()
        got = dict(tr.model.module.named_parameters())
        for name, p in ref.named_parameters():
            g, w = got[name].grad, p.grad
            assert g is not None and w is not None, name
            scale = float(w.abs().max())
            err = float((g - w).abs().max())
            if err > 1e-6 + 1e-5 * scale:
                ok = False
            if scale > 0:
                worst[name] = err / scale
        top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
        print(json.dumps({"ok": ok, "world": world, "rows_per_rank": per_rank * 5, "backend": dist.get_backend(),
                          "worst_relative": top, "params": len(worst)}))
    flag = torch.tensor([int(ok)], device="cpu" if shared else dev)
    dist.broadcast(flag, src=0)
    # every rank must hold the same averaged gradients
    probe = torch.stack([p.grad.double().sum() for p in tr.model.parameters()]).sum().reshape(1)
    probe = probe.cpu() if shared else probe
    lo, hi = probe.clone(), probe.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same = bool((hi - lo).abs().item() <= 1e-9 * max(1.0, abs(hi.item())))
    dist.destroy_process_group()
    sys.exit(0 if (bool(flag.item()) and same) else 1)


if __name__ == "__main__":
    main()
