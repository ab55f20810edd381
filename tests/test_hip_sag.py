"""GPU suite for the SAG construction steps: digat_sag_cos_topk / digat_sag_news_graph through the
construct_SAG mirror, against the vectors minted from the reference's construct_SAG.py and against the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_sag import COS_FIXTURES, GRAPH_FIXTURES, KINDS, assert_topk_matches, unpack_graph

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", GRAPH_FIXTURES)
def test_news_graph_hip_matches_reference_vectors(name):
    from digat_amd import construct_SAG, synthetic
    fx = load_golden(name)
    news_num, top_M, hop, nn = (int(v) for v in fx["meta"])
    sim, news_ID_dict = synthetic.similarity_dict(fx["in_sim_index"], fx["in_sim_cos"], fx["in_sim_len"])
    node_ID, graph, mask = construct_SAG.generate_news_graph("small", sim, news_ID_dict, top_M, hop, nn)
    assert node_ID.dtype == np.int32 and graph.dtype == bool and mask.dtype == bool
    np.testing.assert_array_equal(node_ID, fx["out_news_node_ID"])
    np.testing.assert_array_equal(graph, unpack_graph(fx))
    np.testing.assert_array_equal(mask, fx["out_news_graph_mask"])


@pytest.mark.parametrize("news_num,top_M,hop", [(20000, 5, 2), (5000, 3, 3), (3000, 6, 1), (1, 3, 2), (2, 3, 2)])
def test_news_graph_hip_matches_oracle_on_corpus_sized_inputs(news_num, top_M, hop):
    from digat_amd import construct_SAG, synthetic
    from oracle import sag_oracle
    rng = np.random.default_rng(news_num + top_M)
    ids, cos, length = synthetic.make_similarity_lists(rng, news_num, top_M, isolated_frac=0.03)
    length[rng.random(news_num) < 0.15] = 1
    length[0] = 0
    nn = synthetic.news_graph_size(top_M, hop)
    want = sag_oracle.generate_news_graph(ids, cos, length, top_M, hop, nn)
    got = construct_SAG.news_graph_device(*(torch.from_numpy(a).to(_dev()) for a in (ids, cos, length)), top_M=top_M, hop=hop,
                                          news_node_num=nn)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g.cpu().numpy(), w)


def test_news_graph_hip_reports_a_walk_that_outgrows_the_node_budget():
    from digat_amd import construct_SAG
    ids = torch.tensor([[0, 0], [2, 3], [1, 3], [1, 2]], dtype=torch.int32, device=_dev())
    cos = torch.full((4, 2), 0.9, dtype=torch.float32, device=_dev())
    length = torch.tensor([0, 2, 2, 2], dtype=torch.int32, device=_dev())
    with pytest.raises(IndexError):
        construct_SAG.news_graph_device(ids, cos, length, top_M=2, hop=2, news_node_num=2)
    node_ID, graph, mask = construct_SAG.news_graph_device(ids, cos, length, top_M=2, hop=2, news_node_num=3)
    np.testing.assert_array_equal(node_ID.cpu().numpy(), [[0, 0, 0], [1, 2, 3], [2, 1, 3], [3, 1, 2]])


@pytest.mark.parametrize("name", COS_FIXTURES)
def test_cos_topk_hip_matches_reference_vectors(name):
    from digat_amd import construct_SAG
    fx = load_golden(name)
    n, m, dim, top_M = (int(v) for v in fx["meta"])
    title, content = torch.from_numpy(fx["in_title_all"]), torch.from_numpy(fx["in_content_all"])
    res = construct_SAG.generate_cos_similarities("small", top_M, "news", title[:n], content[:n], title[:m], content[:m])
    assert len(res) == 10
    for i, kind in enumerate(KINDS):
        values, indices = res[2 * i], res[2 * i + 1]
        assert values.dtype == torch.float32 and indices.dtype == torch.int32 and values.device.type == "cpu"
        assert_topk_matches(values.numpy(), indices.numpy(), fx[f"out_{kind}_values"], fx[f"out_{kind}_indices"])


def _float64_topk(title, content, m, k):
    """All five similarity matrices in float64 (normalise, then dot: F.cosine_similarity's order) and their top-k."""
    q = [torch.from_numpy(x).double() for x in (title, content)]
    qn = [x / x.norm(dim=1, keepdim=True).clamp_min(1e-8) for x in q]
    cn = [x[:m] for x in qn]
    tt, cc, tc, ct = qn[0] @ cn[0].T, qn[1] @ cn[1].T, qn[0] @ cn[1].T, qn[1] @ cn[0].T
    return [torch.topk(s, k, dim=1) for s in (tt, cc, tc, ct, (tt + cc + tc + ct) / 4)]


@pytest.mark.parametrize("n,m,dim,top_M", [(3000, 2600, 768, 5),      # 2n >= 2048: the bf16x6 matrix-core GEMM
                                           (4500, 500, 64, 3),        # more than one 4096-row query chunk
                                           (700, 333, 48, 7), (300, 200, 32, 12), (200, 150, 16, 20), (50, 3, 32, 5)])
def test_cos_topk_hip_matches_float64_and_oracle(n, m, dim, top_M):
    from digat_amd import construct_SAG, synthetic
    from oracle import sag_oracle
    title, content = synthetic.make_semantic_embeddings(max(n, m), dim, seed=n + m)
    dev = _dev()
    k = min(top_M, m - 1) + 1
    values, indices = construct_SAG.cos_topk_device(torch.from_numpy(title[:n]).to(dev), torch.from_numpy(content[:n]).to(dev),
                                                    torch.from_numpy(title[:m]).to(dev), torch.from_numpy(content[:m]).to(dev), top_M)
    assert tuple(values.shape) == (5, n, k)
    values, indices = values.cpu().numpy(), indices.cpu().numpy()
    for kind, (wv, wi) in enumerate(_float64_topk(title[:n], content[:n], m, k)):
        assert_topk_matches(values[kind], indices[kind], wv.numpy(), wi.numpy(), atol=3e-6)
    rows = min(n, 48)
    want = sag_oracle.generate_cos_similarities(*(torch.from_numpy(x) for x in (title[:rows], content[:rows], title[:m], content[:m])), top_M)
    for kind, name in enumerate(KINDS):
        assert_topk_matches(values[kind][:rows], indices[kind][:rows], want[name][0].numpy(), want[name][1].numpy(), atol=3e-6)
    assert (np.diff(values, axis=2) <= 0).all()                      # every row sorted descending
    assert indices.min() >= 0 and indices.max() < m


def test_sag_host_mirror_refuses_cpu_tensors_and_oversized_k():
    from digat_amd import _lib, construct_SAG
    x = torch.zeros(8, 32)
    with pytest.raises(_lib.DigatHipError):
        construct_SAG.cos_topk_device(x, x, x, x, 3)
    big = torch.zeros(64, 32, device=_dev())
    with pytest.raises(ValueError):
        construct_SAG.cos_topk_device(big, big, big, big, 40)
