"""GPU suite for the five ablation encoders: eval-mode forward and inference on the HIP kernels against
the vectors minted from the reference's classes (graphEncoders.py:201-842)."""
import types

import numpy as np
import pytest
import torch

from conftest import check_grad_digest, load_golden, regenerate_ablation_train
from oracle import digat_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.mark.parametrize("tag", ["tiny", "default"])
@pytest.mark.parametrize("name", list(O.ABLATIONS))
def test_ablation_encoder_matches_reference(name, tag):
    from digat_amd import graphEncoders, synthetic
    fx = load_golden(f"ablation_{name}_{tag}.npz")
    B, N, H, C, d, L = (int(v) for v in fx["meta"])
    s_w, s_b = (int(v) for v in fx["seeds"])
    state = synthetic.make_ablation_state_dict(name, d, C, L, seed=s_w)
    batch = synthetic.make_encoder_batch(B, N, H, C, d, seed=s_b, empty_history_rows=(1,), isolated_news_rows=(2,))
    cfg = types.SimpleNamespace(news_graph_size=N, max_history_num=H, category_num=C, graph_depth=L, dropout_rate=0.2)
    enc = getattr(graphEncoders, name)(cfg, d)
    missing = enc.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    enc = enc.to(_dev()).eval()
    b = {k: torch.from_numpy(np.ascontiguousarray(v)).to(_dev()) for k, v in batch.items()}
    args = (b["news_graph_embeddings"], b["news_graph"], b["news_graph_mask"], b["user_news_embedding"], b["user_graph"],
            b["user_category_mask"], b["user_category_indices"])
    with torch.no_grad():
        fn, fu = enc(*args)
        c0 = args[0][:, 0] if name == "wo_SA" else enc.compute_news_graph_context(args[0], args[2])
        inn, inu = enc.inference(*args, c0)
    torch.cuda.synchronize()
    for got, key in ((fn, "out_forward_news"), (fu, "out_forward_user"), (inn, "out_inference_news"), (inu, "out_inference_user")):
        np.testing.assert_allclose(got.cpu().numpy(), fx[key], rtol=1e-5, atol=1e-5, err_msg=f"{name}/{tag}/{key}")


def test_model_selects_every_graph_encoder():
    """model.py:18-31: the --graph_encoder choice list of config.py:19."""
    from digat_amd.model import Model, PrecomputedNewsEncoder
    table = torch.zeros(8, 64)
    for choice, cls in (("DIGAT", "DIGAT"), ("wo_SA", "wo_SA"), ("Seq_SA", "Seq_SA"), ("wo_interaction", "wo_interaction"),
                        ("news_graph_wo_inter", "News_graph_wo_inter"), ("user_graph_wo_inter", "User_graph_wo_inter")):
        cfg = types.SimpleNamespace(news_encoder="MSA", graph_encoder=choice, news_graph_size=4, max_history_num=10,
                                    category_num=5, graph_depth=1, dropout_rate=0.2)
        m = Model(cfg, news_encoder=PrecomputedNewsEncoder(table))
        assert type(m.graph_encoder).__name__ == cls
    cfg.graph_encoder = "nope"
    with pytest.raises(Exception, match="nope is not implemented"):
        Model(cfg, news_encoder=PrecomputedNewsEncoder(table))


# ---- training (digat_gat_fwd_train / digat_gat_bwd and the Eq. 8 / context pairs through training.ablation_forward_train) ----
def _train_case(name, tag, dropout=0.0):
    from digat_amd import graphEncoders
    fx = load_golden(f"ablation_train_{name}_{tag}.npz")
    dims, w, flat, users = regenerate_ablation_train(fx, name)
    B, K, N, H, C, d, L = dims
    cfg = types.SimpleNamespace(news_graph_size=N, max_history_num=H, category_num=C, graph_depth=L, dropout_rate=dropout)
    enc = getattr(graphEncoders, name)(cfg, d)
    enc.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}, strict=True)
    enc = enc.to(_dev()).train()
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(_dev()) for k, v in flat.items() if k.startswith("news_")}
    t.update({k: torch.from_numpy(np.ascontiguousarray(v)).to(_dev()) for k, v in users.items() if k.startswith("user_")})
    return fx, enc, t, dims


def _train_step(enc, t, dims):
    B, K, N, H, C, d, L = dims
    Xn = t["news_graph_embeddings"].clone().requires_grad_(True)
    ue = t["user_news_embedding"].clone().requires_grad_(True)

    def expand(x):                                                   # model.py:64-71
        return x.unsqueeze(1).expand(B, K, *x.shape[1:]).contiguous().view(B * K, *x.shape[1:])

    n, u = enc(Xn, t["news_graph"], t["news_graph_mask"], expand(ue), expand(t["user_graph"]),
               expand(t["user_category_mask"]), expand(t["user_category_indices"]))
    logits = (u.view(B, K, d) * n.view(B, K, d)).sum(dim=2)
    loss = (-torch.log_softmax(logits, dim=1).select(1, 0)).mean()    # trainer.py:100
    loss.backward()
    torch.cuda.synchronize()
    return logits, loss, Xn, ue


def _close(got, want, what, rtol=2e-4, atol=2e-6):
    got, want = got.detach().cpu().numpy(), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite"
    scale = max(float(np.abs(want).max()), 1e-12)
    err = np.abs(got - want)
    tol = atol + rtol * np.maximum(np.abs(want), 0.05 * scale)
    assert not (err > tol).any(), f"{what}: max|diff| {err.max():.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("name,tag", [(n, "tiny") for n in O.ABLATIONS] + [("wo_interaction", "default")])
def test_ablation_training_step_matches_reference_autograd(name, tag):
    """Loss, logits and every gradient of one training step (dropout 0) against the reference's autograd.  Tolerance as
    tests/test_hip_training.py: fp32 sums in a different order than ATen's -> 2e-4 relative + 2e-6 absolute."""
    fx, enc, t, dims = _train_case(name, tag)
    logits, loss, Xn, ue = _train_step(enc, t, dims)
    _close(logits, fx["out_logits"], "logits", rtol=2e-5, atol=2e-5)
    _close(loss, fx["out_loss"], "loss", rtol=1e-5, atol=1e-6)
    _close(Xn.grad, fx["g_in_news_graph_embeddings"], "d news_graph_embeddings")
    _close(ue.grad, fx["g_in_user_news_embedding"], "d user_news_embedding")
    for k, p in enc.named_parameters():
        assert p.grad is not None, k
        if tag == "tiny":
            _close(p.grad, fx["g_" + k], "grad " + k)
        else:
            check_grad_digest(fx, k, p.grad.detach().cpu().numpy(), 2e-4, "grad ")


def test_ablation_training_with_dropout_runs_and_is_reproducible():
    """Dropout live (p = 0.2): finite loss and gradients, and the same seed gives the same bits twice."""
    def once():
        torch.manual_seed(5)
        fx, enc, t, dims = _train_case("wo_interaction", "default", dropout=0.2)
        logits, loss, Xn, ue = _train_step(enc, t, dims)
        return loss.detach().clone(), Xn.grad.clone(), [p.grad.clone() for p in enc.parameters()]
    l1, x1, g1 = once()
    l2, x2, g2 = once()
    assert torch.isfinite(l1) and torch.isfinite(x1).all() and all(torch.isfinite(g).all() for g in g1)
    assert torch.equal(l1, l2) and torch.equal(x1, x2) and all(torch.equal(a, b) for a, b in zip(g1, g2))
