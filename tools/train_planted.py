#!/usr/bin/env python3
"""Train a DIGAT graph encoder on a planted-signal synthetic corpus (GPU box) and keep the weights as data.

The synthetic corpora every older fixture was minted with draw their clicks at random: a Xavier-initialised model ranks them
at AUC 0.5 with logits of rms ~600, so "AUC-matched" there is a statement about rank equality of widely spread random
scores.  ``synthetic.SynthSpec(signal=...)`` plants a click signal (sub-topic preferences); this script trains the HIP path
on impressions [DEV, I) of such a corpus with the reference's step (digat_amd/trainer.py: Adam, clip-norm 1, the three
dropouts), reports the held-out dev metrics of impressions [0, DEV), and writes the graph encoder's state dict to
``gpurun_out/trained/<tag>.npz``.  ``oracle/make_golden.py devset_trained_2k`` then runs those weights through the imported
reference in the build container (tests/golden/devset_trained_2k.npz).

  python tools/train_planted.py [--lr 3e-4] [--epochs 4] [--tag planted]
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from digat_amd import synthetic, util  # noqa: E402
from digat_amd.model import Model, PrecomputedNewsEncoder  # noqa: E402
from digat_amd.trainer import SyntheticTrainSet, Trainer  # noqa: E402

# the corpus of the trained-model fixture: shared with oracle/make_golden.py and tests/ through synthetic.PLANTED_SPEC
SPEC = synthetic.PLANTED_SPEC
DEV_IMPRESSIONS = synthetic.PLANTED_DEV_IMPRESSIONS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--tag", default="planted")
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "trained"))
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    full = synthetic.make_corpus(synthetic.SynthSpec(**SPEC))
    dev_c = synthetic.slice_impressions(full, 0, DEV_IMPRESSIONS)
    train_c = synthetic.slice_impressions(full, DEV_IMPRESSIONS, full.spec.impressions)
    spec = full.spec
    cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder="DIGAT", news_graph_size=spec.news_graph_size,
                                max_history_num=spec.max_history_num, category_num=spec.category_num, graph_depth=args.depth,
                                dropout_rate=0.2, epoch=args.epochs, batch_size=64, lr=args.lr, weight_decay=0.0,
                                gradient_clip_norm=1.0, early_stopping_epoch=args.epochs)
    model = Model(cfg, news_encoder=PrecomputedNewsEncoder(torch.from_numpy(full.news_embedding), trainable=False))
    model.initialize()
    model = model.to(dev)
    dc_train = util.DeviceCorpus.from_numpy(train_c, dev)
    dc_dev = util.DeviceCorpus.from_numpy(dev_c, dev)
    ts = SyntheticTrainSet(train_c, 4, seed=0)

    def dev_metrics():
        model.eval()
        scores, m = util.compute_scores(model, dc_dev, 1024, labels=dev_c.row_label)
        return scores, m

    s0, m0 = dev_metrics()
    print(f"untrained: dev AUC {m0[0]:.4f} MRR {m0[1]:.4f} nDCG@5 {m0[2]:.4f} nDCG@10 {m0[3]:.4f}; logits rms {np.sqrt((s0 ** 2).mean()):.2f}",
          flush=True)
    tr = Trainer(model, cfg, dc_train, ts)
    t0 = time.time()
    for e in range(1, args.epochs + 1):
        ts.negative_sampling()
        if e == tr.decay_epoch and args.epochs > 1:
            tr.lr_decay()
        model.train()
        losses = [tr.train_step(idx) for idx in tr.batches(e)]
        s, m = dev_metrics()
        print(f"epoch {e}: {len(losses)} steps, loss {np.mean(losses):.4f} (last 20: {np.mean(losses[-20:]):.4f}); dev AUC {m[0]:.4f} MRR {m[1]:.4f} "
              f"nDCG@5 {m[2]:.4f} nDCG@10 {m[3]:.4f}; logits rms {np.sqrt((s ** 2).mean()):.2f} max {np.abs(s).max():.2f}; {time.time() - t0:.1f}s",
              flush=True)
    os.makedirs(args.out, exist_ok=True)
    state = {k: v.detach().cpu().numpy() for k, v in model.graph_encoder.state_dict().items()}
    path = os.path.join(args.out, args.tag + ".npz")
    np.savez_compressed(path, **state)
    amax = max(float(np.abs(v).max()) for v in state.values())
    print(f"wrote {path} ({os.path.getsize(path) / 2**20:.1f} MiB, {sum(v.size for v in state.values())} parameters, max |w| {amax:.3f}); "
          f"dev scores head {np.round(s[:6], 4).tolist()}")
    np.save(os.path.join(args.out, args.tag + "_dev_scores.npy"), s.astype(np.float32))


if __name__ == "__main__":
    main()
