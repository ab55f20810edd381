import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def planted_devset():
    """(fixture, dev corpus, trained state dict) of tests/golden/devset_trained_2k.npz: the held-out impressions [0, 2000) of the
    planted-signal corpus, scored by the imported reference with the weights tools/train_planted.py trained on the GPU box
    (tests/golden/trained_planted_state.npz).  AUC ~0.64, logits of rms ~10: a model that ranks."""
    from digat_amd import synthetic
    fx = load_golden("devset_trained_2k.npz")
    full = synthetic.make_corpus(synthetic.SynthSpec(**synthetic.PLANTED_SPEC))
    corpus = synthetic.slice_impressions(full, 0, synthetic.PLANTED_DEV_IMPRESSIONS)
    state = load_golden("trained_planted_state.npz")
    chk = (float(corpus.news_embedding.astype(np.float64).sum()) + float(corpus.user_graph.sum()) + float(corpus.news_graph.sum())
           + float(corpus.row_candidate.astype(np.float64).sum()) + float(corpus.row_label.sum()))
    assert abs(chk - float(fx["input_checksum"])) <= 1e-6 * abs(chk), "synthetic generator drifted from the fixture's"
    wchk = sum(float(np.asarray(v, dtype=np.float64).sum()) for v in state.values())
    assert abs(wchk - float(fx["state_checksum"])) <= 1e-9 * max(1.0, abs(wchk)), "trained weights differ from the ones the fixture was minted with"
    return fx, corpus, state


def split_fixture(fx):
    """(inputs, weights, outputs) of a fixture that stores all three."""
    ins = {k[3:]: v for k, v in fx.items() if k.startswith("in_")}
    w = {k[2:]: v for k, v in fx.items() if k.startswith("w_")}
    outs = {k[4:]: v for k, v in fx.items() if k.startswith("out_")}
    return ins, w, outs


def regenerate(fx):
    """Inputs and weights of a seeds-only fixture (default_b8 & co.), rebuilt by the shared generator."""
    from digat_amd import synthetic
    B, N, H, C, d, L = (int(v) for v in fx["meta"])
    s_w, s_b = (int(v) for v in fx["seeds"])
    state = synthetic.make_state_dict(d, C, L, seed=s_w, bias_std=0.05)
    batch = synthetic.make_encoder_batch(B, N, H, C, d, seed=s_b)
    tot = 0.0
    for v in list(batch.values()) + list(state.values()):
        tot += float(np.asarray(v, dtype=np.float64).sum())
    assert abs(tot - float(fx["input_checksum"])) <= 1e-6 * max(1.0, abs(tot)), \
        "synthetic generator drifted from the one that minted the fixture"
    return batch, state


def grad_digest(name, g):
    """The digest oracle/make_golden.py stores for a production-shape gradient: (L2 norm, dot product with a fixed
    pseudo-random probe seeded by the parameter name, every stride-th element)."""
    flat = np.asarray(g, dtype=np.float32).reshape(-1)
    seed = int.from_bytes(name.encode()[-8:].rjust(8, b"\0"), "little") % (2 ** 32)
    probe = np.random.default_rng(seed).standard_normal(flat.size).astype(np.float32)
    stride = max(1, flat.size // 2048)
    return (float(np.sqrt((flat.astype(np.float64) ** 2).sum())), float((flat.astype(np.float64) * probe).sum()),
            flat[::stride].copy())


def check_grad_digest(fx, name, g, rtol, what=""):
    """Hold gradient ``g`` of parameter ``name`` to the digest in fixture ``fx``; tolerances relative to the gradient's norm."""
    norm, dot, samples = grad_digest(name, g)
    want_norm, want_dot, want_samples = float(fx["gn_" + name]), float(fx["gp_" + name]), fx["gs_" + name]
    scale = max(want_norm, 1e-30)
    assert abs(norm - want_norm) <= rtol * scale, f"{what}{name}: |g| = {norm:.6e}, reference {want_norm:.6e}"
    # the probe is a unit-variance random vector: |<g - g_ref, probe>| ~ |g - g_ref|
    assert abs(dot - want_dot) <= 4 * rtol * scale, f"{what}{name}: probe {dot:.6e}, reference {want_dot:.6e} (|g| {want_norm:.3e})"
    n = max(1, np.asarray(g).size)
    tol = rtol * scale / np.sqrt(n) * 30 + rtol * np.abs(want_samples)          # elementwise: 30 x the rms budget + relative
    bad = np.abs(samples - want_samples) > tol
    assert not bad.any(), f"{what}{name}: {int(bad.sum())} of {len(samples)} sampled elements off, worst {np.abs(samples - want_samples).max():.3e}"


def regenerate_train(fx):
    """Inputs and weights of train_step_default.npz: (meta, state, per-row news batch, per-impression user batch)."""
    from digat_amd import synthetic
    B, K, N, H, C, d, L = (int(v) for v in fx["meta"])
    s_w, s_n, s_u = (int(v) for v in fx["seeds"])
    state = synthetic.make_state_dict(d, C, L, seed=s_w, bias_std=0.05)
    flat = synthetic.make_encoder_batch(B * K, N, H, C, d, seed=s_n)
    users = synthetic.make_encoder_batch(B, N, H, C, d, seed=s_u, empty_history_rows=(3,))
    both = dict(flat)
    both.update({"u_" + k: v for k, v in users.items()})
    tot = 0.0
    for v in list(both.values()) + list(state.values()):
        tot += float(np.asarray(v, dtype=np.float64).sum())
    assert abs(tot - float(fx["input_checksum"])) <= 1e-6 * max(1.0, abs(tot)), \
        "synthetic generator drifted from the one that minted the fixture"
    return (B, K, N, H, C, d, L), state, flat, users


def regenerate_ablation_train(fx, name):
    """Inputs and weights of ablation_train_<name>_<tag>.npz: (meta, state, per-row news batch, per-impression user batch)."""
    from digat_amd import synthetic
    B, K, N, H, C, d, L = (int(v) for v in fx["meta"])
    s_w, s_n, s_u = (int(v) for v in fx["seeds"])
    state = synthetic.make_ablation_state_dict(name, d, C, L, seed=s_w)
    flat = synthetic.make_encoder_batch(B * K, N, H, C, d, seed=s_n, isolated_news_rows=(2,))
    users = synthetic.make_encoder_batch(B, N, H, C, d, seed=s_u, empty_history_rows=(1,))
    both = dict(flat)
    both.update({"u_" + k: v for k, v in users.items()})
    tot = 0.0
    for v in list(both.values()) + list(state.values()):
        tot += float(np.asarray(v, dtype=np.float64).sum())
    assert abs(tot - float(fx["input_checksum"])) <= 1e-6 * max(1.0, abs(tot)), \
        "synthetic generator drifted from the one that minted the fixture"
    return (B, K, N, H, C, d, L), state, flat, users


@pytest.fixture(scope="session")
def golden():
    return load_golden
