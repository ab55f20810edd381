"""CPU suite for the host side of the path: ranking/metrics (evaluate.py), impression-aligned row
sharding and the world_size-2 score gather over gloo (util.py).  The HIP scorer is replaced by a
deterministic stand-in (``score_fn``) — no GPU compute happens here."""
import os
import socket

import numpy as np
import pytest
import torch

from conftest import load_golden
from digat_amd import evaluate, synthetic, util

TINY = synthetic.SynthSpec(news_num=512, sag_neighbors=3, sag_hops=1, max_history_num=10, category_num=5,
                           embedding_dim=64, impressions=200, mean_candidates=12.0, max_candidates=40, seed=41)


def test_ranks_and_metrics_match_reference_outputs():
    fx = load_golden("devset_tiny.npz")
    corpus = synthetic.make_corpus(TINY)
    ranks = evaluate.impression_ranks(fx["scores"], corpus.row_impression)
    assert "\n".join(evaluate.rank_lines(ranks, corpus.row_impression)) == str(fx["rank_lines"])
    got = evaluate.scoring(corpus.row_label, ranks, corpus.row_impression)
    np.testing.assert_allclose(got, fx["metrics"], rtol=0, atol=1e-9)      # sklearn AUC, MRR, nDCG@5/10


def test_ranking_is_stable_on_ties():
    scores = np.array([0.5, 0.5, 0.9, 0.5, 0.1, 0.1], dtype=np.float32)
    imp = np.array([0, 0, 0, 0, 1, 1])
    assert evaluate.impression_ranks(scores, imp).tolist() == [2, 3, 1, 4, 1, 2]


def test_metrics_against_sklearn_on_random_impressions():
    from sklearn.metrics import roc_auc_score
    rng = np.random.default_rng(0)
    imp = np.repeat(np.arange(50), rng.integers(2, 30, size=50))
    labels = np.zeros(len(imp), dtype=np.int8)
    for i in range(50):
        sel = np.flatnonzero(imp == i)
        labels[rng.choice(sel, size=rng.integers(1, len(sel)), replace=False)] = 1
    scores = rng.standard_normal(len(imp)).astype(np.float32)
    ranks = evaluate.impression_ranks(scores, imp)
    auc = evaluate.scoring(labels, ranks, imp)[0]
    want = np.mean([roc_auc_score(labels[imp == i], 1.0 / ranks[imp == i]) for i in range(50)])
    assert abs(auc - want) < 1e-12


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_rows_is_an_impression_aligned_partition(world):
    corpus = synthetic.make_corpus(TINY)
    imp = corpus.row_impression
    blocks = [util.shard_rows(imp, world, r) for r in range(world)]
    assert blocks[0][0] == 0 and blocks[-1][1] == len(imp)
    for (s0, e0), (s1, e1) in zip(blocks[:-1], blocks[1:]):
        assert e0 == s1
    for s, e in blocks:
        if 0 < s < len(imp):
            assert imp[s] != imp[s - 1]          # never splits an impression
    sizes = [e - s for s, e in blocks]
    assert max(sizes) - min(sizes) <= 2 * TINY.max_candidates


def _fake_scores(model, dc, start, end, batch_size):
    idx = torch.arange(start, end, dtype=torch.float64)
    return torch.sin(idx * 12.9898).to(torch.float32) * 3.0


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    corpus = synthetic.make_corpus(TINY)
    dc = util.DeviceCorpus.from_numpy(corpus, torch.device("cpu"))
    scores, metrics = util.compute_scores(None, dc, 64, labels=corpus.row_label, rank=rank, world_size=world,
                                          score_fn=_fake_scores)
    q.put((rank, scores, metrics))
    dist.destroy_process_group()


def test_two_rank_gloo_gather_equals_single_process():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict()
    for _ in procs:
        r, scores, metrics = q.get(timeout=120)
        results[r] = (scores, metrics)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    corpus = synthetic.make_corpus(TINY)
    dc = util.DeviceCorpus.from_numpy(corpus, torch.device("cpu"))
    want_scores, want_metrics = util.compute_scores(None, dc, 64, labels=corpus.row_label, score_fn=_fake_scores)
    np.testing.assert_array_equal(results[0][0], want_scores)
    np.testing.assert_array_equal(results[1][0], want_scores)       # all_gather: every rank holds all scores
    assert results[1][1] is None                                    # metrics on rank 0 only
    np.testing.assert_allclose(results[0][1], want_metrics, rtol=0, atol=0)


class _FlagEncoder:
    """What util.range_overflow_any_rank needs of an encoder: the flag and a tensor that says where it lives."""
    def __init__(self, hit):
        self.hit, self.topic_node_embedding = hit, torch.zeros(1)

    def range_overflowed(self):
        hit, self.hit = self.hit, False
        return hit


def _flag_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first = util.range_overflow_any_rank(_FlagEncoder(rank == 1), world)       # only rank 1's shard overflowed
    second = util.range_overflow_any_rank(_FlagEncoder(False), world)
    q.put((rank, first, second))
    dist.destroy_process_group()


def test_fp16x3_range_fallback_is_decided_for_all_ranks_together():
    """A rank whose shard left fp16x3's range must not fall back alone (the all_gather would mix scores of two formats and the
    ranks' caches would diverge): the flag is MAX-reduced over the group, every rank sees the same answer."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_flag_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [(0, True, False), (1, True, False)]
    assert util.range_overflow_any_rank(_FlagEncoder(True)) and not util.range_overflow_any_rank(_FlagEncoder(False))


def test_rank_file_bytes_equal_the_joined_rank_lines():
    """evaluate.rank_file_bytes (the library's host-side C formatter) against "\\n".join(rank_lines(...)): impressions without
    rows, one-row impressions, three-digit ranks, six-digit ids."""
    from digat_amd import evaluate
    rng = np.random.default_rng(0)
    for trial in range(4):
        cnt = rng.integers(1, 320, size=1200)
        if trial == 1:
            cnt[[5, 100, 1199]] = 0
        imp = np.repeat(np.arange(len(cnt)), cnt)
        ranks = rng.integers(1, 400, size=len(imp)).astype(np.int64)
        if trial == 2:
            imp, ranks = np.repeat(np.arange(3), [2, 0, 3]), np.array([1, 2, 10, 3, 100])
        if trial == 3:
            imp, ranks = np.array([0, 0, 123456]), np.array([2, 1, 1])
        assert evaluate.rank_file_bytes(ranks, imp) == "\n".join(evaluate.rank_lines(ranks, imp)).encode()
    assert evaluate.rank_file_bytes(np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)) == b""


def test_launch_sets_are_whole_batches_covering_the_range():
    """util.launch_batches: the row ranges one pass through the encoder takes — whole multiples of the caller's dev batch."""
    from digat_amd import util
    assert util.launch_batches(0, 10000, 1024) == [(0, 4096), (4096, 8192), (8192, 10000)]          # LAUNCH_ROWS = 4096
    assert util.launch_batches(5, 3000, 1024, 1024) == [(5, 1029), (1029, 2053), (2053, 3000)]      # the reference's own chunking
    assert util.launch_batches(0, 5000, 600, 2500) == [(0, 2400), (2400, 4800), (4800, 5000)]       # rounded down to whole batches
    assert util.launch_batches(0, 5000, 8192) == [(0, 5000)]                                        # never below one batch
    assert util.launch_batches(7, 7, 1024) == []
    for start, end, b, rows in [(0, 99999, 1024, None), (3, 4099, 64, 1000), (0, 1, 1, 1)]:
        sets = util.launch_batches(start, end, b, rows)
        assert sets[0][0] == start and sets[-1][1] == end and all(a[1] == c[0] for a, c in zip(sets, sets[1:]))
        assert all((e - s) % b == 0 for s, e in sets[:-1])


def test_heavy_history_profile_is_the_other_adjacency_regime():
    """SynthSpec(history_profile="heavy"): every user with a full history in 2-4 categories (what real MIND's truncated long tail
    looks like): 13-26 adjacency entries per history node and ~80 % of the nodes live, against 4-5 entries and ~45 % for the
    default profile — the regime in which the sparse / dense choice of Eq. 8 is a real question.  The graphs obey
    MIND_corpus.py:153-176 (symmetric, self loops, same-category items fully connected, padding bucket never unmasked), and the
    default profile's random stream is untouched by the new branch (the golden fixtures' checksums hold that elsewhere)."""
    spec = synthetic.SynthSpec(news_num=2048, impressions=120, seed=3, history_profile="heavy")
    c = synthetic.make_corpus(spec)
    H, C = spec.max_history_num, spec.category_num
    g = c.user_graph
    assert (c.history > 0).all() and (c.extra["history_len"] == H).all()
    ent = g.sum(-1)
    assert 12.0 < ent.mean() < 24.0 and (ent > 1).mean() > 0.7
    cats = c.user_category_mask[:, :C].sum(1)
    assert cats.min() >= 1 and cats.max() <= 4 and not c.user_category_mask[:, C].any()
    assert (g == g.transpose(0, 2, 1)).all() and g[:, np.arange(H + C), np.arange(H + C)].all()
    ci = c.user_category_indices
    same = ci[:, :, None] == ci[:, None, :]
    assert (g[:, :H, :H] == same).all()                        # same-category history items are fully connected, others not
    d = synthetic.make_corpus(synthetic.SynthSpec(news_num=2048, impressions=120, seed=3))
    ent_d = d.user_graph.sum(-1)
    assert ent_d.mean() < 7.0 and (ent_d > 1).mean() < 0.6
    with pytest.raises(ValueError):
        synthetic.make_corpus(synthetic.SynthSpec(news_num=64, impressions=4, history_profile="nope"))
