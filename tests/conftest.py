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


@pytest.fixture(scope="session")
def golden():
    return load_golden
