"""GPU suite for the five ablation encoders: eval-mode forward and inference on the HIP kernels against
the vectors minted from the reference's classes (graphEncoders.py:201-842)."""
import types

import numpy as np
import pytest
import torch

from conftest import load_golden
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
