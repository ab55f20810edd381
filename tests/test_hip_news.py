"""GPU suite for the MSA news encoder: digat_msa_fwd (through newsEncoders.MSA in eval mode) against the
vectors minted from the reference's layers.py modules and against the oracle."""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _encoder(V, dm, h, dk, att, Lw, state):
    from digat_amd import newsEncoders
    cfg = types.SimpleNamespace(vocabulary_size=V, word_embedding_dim=dm, max_title_length=Lw, dropout_rate=0.2,
                                MSA_head_num=h, MSA_head_dim=dk, attention_dim=att)
    enc = newsEncoders.MSA(cfg)
    missing = enc.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return enc.to(_dev()).eval()


@pytest.mark.parametrize("name", ["msa_tiny.npz", "msa_default.npz"])
def test_msa_hip_matches_reference_vectors(name):
    from digat_amd import synthetic
    fx = load_golden(name)
    T_, Lw, V, dm, h, dk, att = (int(v) for v in fx["meta"])
    s_w, s_t = (int(v) for v in fx["seeds"])
    state = synthetic.make_msa_state(V, dm, h, dk, att, seed=s_w)
    text, mask = synthetic.make_titles(T_, Lw, V, seed=s_t)
    enc = _encoder(V, dm, h, dk, att, Lw, state)
    with torch.no_grad():
        got = enc(torch.from_numpy(text).to(_dev()).unsqueeze(1), torch.from_numpy(mask).to(_dev()).unsqueeze(1)).squeeze(1)
    np.testing.assert_allclose(got.cpu().numpy(), fx["out_news_representation"], rtol=1e-5, atol=2e-6)


def test_msa_hip_batch_on_the_matrix_core_path_matches_oracle():
    """Production shapes and enough titles (T*Lw >= 2048) for the bf16x6 GEMM with the embedding lookup folded in."""
    from digat_amd import synthetic
    from oracle import news_oracle
    T_, Lw, V, dm, h, dk, att = 300, 32, 2000, 300, 16, 25, 256
    state = synthetic.make_msa_state(V, dm, h, dk, att, seed=61)
    text, mask = synthetic.make_titles(T_, Lw, V, seed=62)
    enc = _encoder(V, dm, h, dk, att, Lw, state)
    with torch.no_grad():
        got = enc(torch.from_numpy(text).to(_dev()), torch.from_numpy(mask).to(_dev()))
        want = news_oracle.msa_forward({k: torch.from_numpy(v) for k, v in state.items()}, torch.from_numpy(text),
                                       torch.from_numpy(mask), h)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-5, atol=2e-6)
    # eval-mode forward with grad enabled falls back to the stock modules: same numbers within fp32 reassociation
    stock = enc(torch.from_numpy(text[:8]).to(_dev()).unsqueeze(0), torch.from_numpy(mask[:8]).to(_dev()).unsqueeze(0))
    np.testing.assert_allclose(stock.detach().cpu().numpy()[0], want.numpy()[:8], rtol=1e-4, atol=1e-5)


def test_news_cache_in_batches_equals_one_call():
    """util.cache_news_representations (util.py:24-33) over ragged batches vs one call: same bits."""
    from digat_amd import synthetic, util
    T_, Lw, V, dm, h, dk, att = 1000, 32, 3000, 300, 16, 25, 256
    state = synthetic.make_msa_state(V, dm, h, dk, att, seed=71)
    text, mask = synthetic.make_titles(T_, Lw, V, seed=72)
    enc = _encoder(V, dm, h, dk, att, Lw, state)
    tt, tm = torch.from_numpy(text).to(_dev()), torch.from_numpy(mask).to(_dev())
    a = util.cache_news_representations(enc, tt, tm, 384)
    with torch.no_grad():
        b = enc(tt, tm)
    assert torch.equal(a, b)
