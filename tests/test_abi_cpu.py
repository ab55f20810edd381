"""CPU suite: the C-ABI library loads and exports every symbol include/digat_hip.h declares, the
host-side shape planning answers sanely, and the product path refuses to run without a GPU
(no compute calls here)."""
import os
import re
import types

import numpy as np
import pytest
import torch

from conftest import REPO


def declared_functions():
    text = open(os.path.join(REPO, "include", "digat_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef DIGAT_LAB.*?#endif", "", text, flags=re.S)          # LAB builds only: not part of the product ABI
    return sorted(set(re.findall(r"\b(digat_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from digat_amd import _lib, build
    build.build(verbose=False)
    L = _lib.lib()
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/digat_hip.h but not exported"
    assert set(names) == set(_lib.EXPORTED), "ctypes signature table out of sync with the header"
    assert L.digat_version() == _lib.ABI_VERSION == 4
    assert b"workspace" in L.digat_error_string(3)


def test_torch_extension_is_built_and_binds_the_same_library():
    """The thin torch extension over the C ABI (csrc/digat_torch_ext.cpp; BASELINE north_star's binding): built in-tree next to
    libdigat_hip.so, importable without a GPU, same ABI version, and it refuses CPU tensors (no compute call is made here)."""
    from digat_amd import _lib, build
    build.build(verbose=False)
    assert os.path.exists(_lib.EXT_PATH)
    X = _lib.ext()
    assert X is not None and X.abi_version() == _lib.lib().digat_version()
    assert {"encoder_fwd", "encoder_fwd_grouped", "row_logits", "user_row_runs"} <= set(dir(X))
    with pytest.raises(RuntimeError):
        X.row_logits(torch.zeros(2, 4), torch.zeros(2, 4), torch.zeros(2))
    # the extension is host code only: every kernel lives in libdigat_hip.so, which it links by rpath
    import subprocess
    needed = subprocess.run(["readelf", "-d", _lib.EXT_PATH], capture_output=True, text=True).stdout
    assert "libdigat_hip.so" in needed and "$ORIGIN" in needed


def test_workspace_queries():
    from digat_amd import _lib
    L = _lib.lib()
    B, N, H, C, d, depth = 1024, 10, 50, 17, 400, 3
    U = H + C
    x = L.digat_xattn_workspace_bytes(B, U, d)
    assert x >= 3 * B * U * d * 4 + B * d * 4
    e = L.digat_encoder_workspace_bytes(B, N, H, C, d, depth)
    assert e >= 2 * B * U * d * 4 + 2 * B * N * d * 4 + x
    assert e < 2 << 30          # far below the 288 GB of one MI355X


def test_struct_layout_matches_header():
    """digat_params: 4 int32 + 14 pointers + 2 * DIGAT_MAX_DEPTH * 9 pointers + 6 folded-query pointers + 1 split-weight pointer
    + 2 + (DIGAT_MAX_DEPTH + 1) split images of the [B,d] linears + the range-flag pointer."""
    import ctypes
    from digat_amd import _lib
    assert ctypes.sizeof(_lib.LayerParams) == 9 * 8
    assert ctypes.sizeof(_lib.Params) == 16 + 14 * 8 + 2 * 16 * 9 * 8 + 7 * 8 + 2 * 8 + 17 * 8 + 8 + 8      # ... range_flag, featureAffine_fsplit (round 6)


def test_module_mirrors_reference_parameter_names():
    from digat_amd import synthetic
    from digat_amd.graphEncoders import DIGAT
    cfg = types.SimpleNamespace(news_graph_size=10, max_history_num=50, category_num=17, graph_depth=3, dropout_rate=0.2)
    enc = DIGAT(cfg, 400)
    enc.initialize()
    want = synthetic.make_state_dict(400, 17, 3, seed=0)
    got = enc.state_dict()
    assert set(got) == set(want)
    for k in want:
        assert tuple(got[k].shape) == want[k].shape, k
    assert sum(p.numel() for p in enc.parameters()) == 5_296_000
    assert enc.max_history_num == 50 and enc.category_num == 18 and enc.user_graph_size == 67
    assert float(enc.topic_node_embedding.detach().abs().sum()) == 0.0             # zero init (graphEncoders.py:28)


def test_no_cpu_fallback():
    from digat_amd import _lib, synthetic
    from digat_amd.graphEncoders import DIGAT
    cfg = types.SimpleNamespace(news_graph_size=4, max_history_num=10, category_num=5, graph_depth=1, dropout_rate=0.2)
    enc = DIGAT(cfg, 64).eval()
    b = synthetic.make_encoder_batch(2, 4, 10, 5, 64, seed=0)
    with pytest.raises(_lib.DigatHipError):
        enc(*(torch.from_numpy(np.ascontiguousarray(b[k])) for k in
              ("news_graph_embeddings", "news_graph", "news_graph_mask", "user_news_embedding", "user_graph",
               "user_category_mask", "user_category_indices")))


def test_no_process_wide_operand_format():
    """The matrix-core operand format travels with the split image / digat_params.flags: the library has no setter for it
    and no mutable global behind one."""
    from digat_amd import _lib
    L = _lib.lib()
    assert not hasattr(L, "digat_set_gemm_format") and not hasattr(L, "digat_get_gemm_format")
    csrc = os.path.join(REPO, "digat_amd", "csrc")
    for f in os.listdir(csrc):
        text = open(os.path.join(csrc, f)).read()
        assert "g_gemm_format" not in text and "GemmFormatScope" not in text, f


def test_no_environment_switch_and_no_mutable_mode_in_the_product_library():
    """Round-3 verdict, items 8 / 9 / 12: the wrong-result timing ablations (DIGAT_*_SKIP), every other environment knob, the
    LDS-staged Eq. 8 variants and the process-wide setters (side stream, live-row lists, staged mode) are LAB-build material
    (-DDIGAT_LAB); the product library neither reads an environment variable nor exports a mode setter."""
    from digat_amd import _lib, build
    build.build(verbose=False)
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"_SKIP", b"DIGAT_SINGLE_STREAM", b"DIGAT_NO_SKIP", b"DIGAT_XATTN_STAGED", b"DIGAT_STREAM_TIMERS", b"DIGAT_L0_LIVE",
                 b"DIGAT_SPARSE_PER_NODE", b"DIGAT_NEWS_LDS", b"DIGAT_POOL_RESIDENT", b"DIGAT_SKINNY_SPLIT", b"DIGAT_SPARSE_XCD",
                 b"xattn_staged_kernel", b"getenv"):
        assert name not in blob, name
    L = _lib.lib()
    for setter in ("digat_set_side_stream", "digat_set_live_row_skipping", "digat_set_staged_xattn", "digat_set_gemm_format"):
        assert not hasattr(L, setter), setter
    text = open(os.path.join(REPO, "digat_amd", "csrc", "digat_kernels.hip")).read()
    assert "g_live_rows_on" not in text and "g_side_stream_on" not in text


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(REPO, "digat_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
