#!/usr/bin/env python3
"""Kernel micro-benchmarks on the GPU box (development aid, not part of the product or the tests).

  python tools/kbench.py xattn  [B n d]      Eq. 8 pairwise kernel alone (digat_xattn_pairwise_fwd)
  python tools/kbench.py linear [M N K]      fp32 MFMA linear (digat_linear_f32)
  python tools/kbench.py encoder             whole DIGAT.inference, per-kernel-kind breakdown
Timing with torch events on the current stream, median of --iters launches.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from digat_amd import _lib  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def bench_xattn(B=1024, n=67, d=400, density=None):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    P, Q, h, X = (torch.randn(B, n, d, device=dev, generator=g) for _ in range(4))
    r = torch.randn(B, d, device=dev, generator=g)
    a = torch.randn(d, device=dev, generator=g) * 0.1
    if density == "mind":
        from digat_amd import synthetic
        batch = synthetic.make_encoder_batch(B, 10, 50, n - 50, d, seed=0)
        A = torch.from_numpy(batch["user_graph"]).to(dev).view(torch.uint8)
        print(f"   MIND-like user graphs: element density {float(batch['user_graph'].mean()):.3f}")
    elif density is None:
        A = torch.ones(B, n, n, dtype=torch.uint8, device=dev)
    else:
        A = (torch.rand(B, n, n, device=dev, generator=g) < density).to(torch.uint8)
        A |= torch.eye(n, dtype=torch.uint8, device=dev)[None]
    out = torch.empty_like(X)
    alpha = torch.empty(B, n, n, device=dev)
    L = _lib.lib()

    def run():
        _lib.check(L.digat_xattn_pairwise_fwd(P.data_ptr(), Q.data_ptr(), h.data_ptr(), X.data_ptr(),
                                              a.data_ptr(), A.data_ptr(), out.data_ptr(), alpha.data_ptr(), B, n, d,
                                              _lib.stream_ptr()), "xattn")
    med, best = timeit(run)
    bytes_b = B * (5.0 * n * d * 4 + d * 4 + n * n) + 4 * d
    lane_ops = 3.0 * B * n * n * d + 2.0 * B * n * n * d / 2
    print(f"xattn B={B} n={n} d={d} skip={os.environ.get('DIGAT_XATTN_SKIP', '0')}: median {med*1e3:.1f} us  best {best*1e3:.1f} us  "
          f"{bytes_b/med/1e6:.0f} GB/s algorithmic  ({bytes_b/1e6:.0f} MB)")


def bench_xattn_entry(B=1024, n=67, d=400, per_node=0):
    """digat_xattn_fwd_mode (K3 + projections + Eq. 8) on MIND-shaped user graphs: the dense pair against the sparse kernel."""
    from digat_amd import synthetic
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    batch = synthetic.make_encoder_batch(B, 10, 50, n - 50, d, seed=0)
    if per_node > 0:       # random graphs with about per_node entries per node (self loops included) instead of MIND-shaped ones
        rnd = torch.rand(B, n, n, generator=torch.Generator().manual_seed(1)) < (per_node - 1) / (n - 1)
        batch["user_graph"] = (rnd | torch.eye(n, dtype=torch.bool)[None]).numpy()
    A = torch.from_numpy(batch["user_graph"]).to(dev).view(torch.uint8)
    X = torch.randn(B, n, d, device=dev, generator=g)
    ctx = torch.randn(B, d, device=dev, generator=g)
    W, F1, F2, F3 = (torch.randn(d, d, device=dev, generator=g) / d ** 0.5 for _ in range(4))
    bW, b3 = torch.randn(d, device=dev, generator=g) * 0.1, torch.randn(d, device=dev, generator=g) * 0.1
    a = torch.randn(d, device=dev, generator=g) * 0.1
    L = _lib.lib()
    nbytes = L.digat_xattn_workspace_bytes(B, n, d)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    outs = {}
    for name, mode in (("dense", 1), ("sparse", 2)):
        out = torch.empty_like(X)

        def run():
            _lib.check(L.digat_xattn_fwd_mode(X.data_ptr(), A.data_ptr(), ctx.data_ptr(), W.data_ptr(), bW.data_ptr(), F1.data_ptr(),
                                              F2.data_ptr(), F3.data_ptr(), b3.data_ptr(), a.data_ptr(), out.data_ptr(), B, n, d, mode,
                                              ws.data_ptr(), nbytes, _lib.stream_ptr()), "xattn_fwd_mode")
        med, best = timeit(run)
        outs[name] = out
        print(f"digat_xattn_fwd_mode {name:6s} B={B} n={n} d={d} (fp32 MFMA projections of all rows included): median {med*1e3:.1f} us  best {best*1e3:.1f} us")
    print(f"   max |dense - sparse| = {float((outs['dense'] - outs['sparse']).abs().max()):.2e}"
          f"   adjacency entries per node {float(batch['user_graph'].sum() / (B * n)):.2f}")


def bench_linear(M=68608, N=400, K=400):
    dev = torch.device("cuda:0")
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    L = _lib.lib()

    def run():
        _lib.check(L.digat_linear_f32(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, N, K,
                                      _lib.stream_ptr()), "linear")
    med, best = timeit(run)
    fl = 2.0 * M * N * K
    print(f"linear M={M} N={N} K={K}: median {med*1e3:.1f} us best {best*1e3:.1f} us  {fl/med/1e9:.1f} TFLOP/s")
    if M >= 2048 and N % 80 == 0 and K % 8 == 0:
        y6 = torch.empty(M, N, device=dev)
        ws = torch.empty(L.digat_split_weights_bytes(N, K), dtype=torch.uint8, device=dev)

        def run6():
            _lib.check(L.digat_linear_f32x3(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y6.data_ptr(), N, M, N, K,
                                            ws.data_ptr(), int(os.environ.get("KBENCH_GEMM_FORMAT", "0")), _lib.stream_ptr()), "linear x3")
        m6, b6 = timeit(run6)
        print(f"   bf16x6 (incl. weight split): median {m6*1e3:.1f} us best {b6*1e3:.1f} us  {fl/m6/1e9:.1f} fp32-equivalent TFLOP/s"
              f"  max|diff vs fp32 kernel| {float((y6 - y).abs().max()):.2e}")
    ref = torch.addmm(b, x, w.t())
    t_ref, _ = timeit(lambda: torch.addmm(b, x, w.t()))
    print(f"   (rocBLAS addmm for scale: {t_ref*1e3:.1f} us, max|diff| {float((ref - y).abs().max()):.2e})")


def bench_topic(B=1024, H=50, C=17, d=400):
    dev = torch.device("cuda:0")
    U = H + C
    Xu = torch.randn(B, U, d, device=dev)
    kq = torch.randn(B, d, device=dev)
    idx = torch.randint(0, C + 1, (B, H), device=dev, dtype=torch.int64)
    out = torch.empty(B, C + 1, d, device=dev)
    L = _lib.lib()

    def run():
        _lib.check(L.digat_topic_pool_fwd(Xu.data_ptr(), kq.data_ptr(), idx.data_ptr(), out.data_ptr(), B, U, H, C + 1, d,
                                          _lib.stream_ptr()), "topic")
    med, best = timeit(run)
    by = B * (H * d * 4 + d * 4 + H * 8 + (C + 1) * d * 4)
    print(f"topic B={B} H={H} C1={C+1} d={d} skip={os.environ.get('DIGAT_TOPIC_SKIP', '0')}: median {med*1e3:.1f} us best {best*1e3:.1f} us "
          f"{by/med/1e6:.0f} GB/s algorithmic ({by/1e6:.0f} MB)")


def bench_msa(T=8192, Lw=32, V=30000, dm=300, h=16, dk=25, att=256):
    import types
    from digat_amd import newsEncoders, synthetic
    dev = torch.device("cuda:0")
    state = synthetic.make_msa_state(V, dm, h, dk, att, seed=1)
    text, mask = synthetic.make_titles(T, Lw, V, seed=2)
    cfg = types.SimpleNamespace(vocabulary_size=V, word_embedding_dim=dm, max_title_length=Lw, dropout_rate=0.2,
                                MSA_head_num=h, MSA_head_dim=dk, attention_dim=att)
    enc = newsEncoders.MSA(cfg)
    enc.load_state_dict({k_: torch.from_numpy(v) for k_, v in state.items()})
    enc = enc.to(dev).eval()
    tt, tm = torch.from_numpy(text).to(dev), torch.from_numpy(mask).to(dev)

    def hip():
        with torch.no_grad():
            return enc(tt, tm)

    def stock():
        with torch.no_grad():
            return enc.forward_stock(tt.unsqueeze(0), tm.unsqueeze(0))
    a = hip()
    b = stock().detach()[0]
    m1, _ = timeit(hip, iters=10)
    m2, _ = timeit(stock, iters=10)
    flops = T * Lw * (2.0 * dm * 3 * h * dk + 2.0 * h * dk * att) + T * h * 4.0 * Lw * Lw * dk
    print(f"MSA news encoder T={T} titles x {Lw} tokens: HIP {m1:.2f} ms ({T/m1/1e3:.2f} M titles/s, {flops/m1/1e9:.1f} TFLOP/s fp32-eq)"
          f"   stock torch {m2:.2f} ms   max|diff| {float((a - b).abs().max()):.2e}")


def bench_msa_train(T=6400, Lw=32, V=30000, dm=300, h=16, dk=25, att=256):
    """One training step of the MSA news encoder (forward + backward, dropout 0.2): digat_msa_fwd_train / digat_msa_bwd /
    digat_embedding_bwd against the stock PyTorch modules.  T = 6400 titles is the reference's step (64 impressions x
    (5 candidates x 10 SAG nodes + 50 history items))."""
    import types
    from digat_amd import newsEncoders, synthetic
    dev = torch.device("cuda:0")
    state = synthetic.make_msa_state(V, dm, h, dk, att, seed=1)
    text, mask = synthetic.make_titles(T, Lw, V, seed=2)
    cfg = types.SimpleNamespace(vocabulary_size=V, word_embedding_dim=dm, max_title_length=Lw, dropout_rate=0.2,
                                MSA_head_num=h, MSA_head_dim=dk, attention_dim=att)
    enc = newsEncoders.MSA(cfg)
    enc.load_state_dict({k_: torch.from_numpy(v) for k_, v in state.items()})
    enc = enc.to(dev).train()
    tt, tm = torch.from_numpy(text).to(dev).unsqueeze(0), torch.from_numpy(mask).to(dev).unsqueeze(0)
    R = torch.randn(1, T, h * dk, device=dev)

    def step(fn):
        def run():
            enc.zero_grad(set_to_none=True)
            (fn(tt, tm) * R).sum().backward()
        return run
    m1, _ = timeit(step(enc), iters=10)
    m2, _ = timeit(step(enc.forward_stock), iters=10)
    M = T * Lw
    flops = 3 * (M * (2.0 * dm * 3 * h * dk + 2.0 * h * dk * att)) + T * h * Lw * Lw * dk * 2.0 * 7
    print(f"MSA training step T={T} titles x {Lw} tokens: HIP {m1:.2f} ms ({flops/m1/1e9:.1f} TFLOP/s fp32-eq)   stock torch {m2:.2f} ms")
    from digat_amd import _lib
    _lib.lib().digat_profile_start(4096)
    step(enc)()
    torch.cuda.synchronize()
    import ctypes as C
    ms = (C.c_double * 7)(); wk = (C.c_double * 7)(); cn = (C.c_int * 7)()
    _lib.lib().digat_profile_stop(ms, wk, cn)
    print("   library kernel ms by kind (proj, linear, xattn, pool, topic, glue, agg):", [round(v, 3) for v in ms], list(cn))


def bench_sag(n=30000, m=30000, dim=768, top_M=5, news_num=65238, hop=2, cpu_rows=32):
    """SAG construction: cosine top-k of one category of n news against an m-news corpus, and the walk over
    news_num similarity lists; the reference's per-news loop (oracle restatement) timed on cpu_rows rows beside it."""
    import time
    from digat_amd import construct_SAG, synthetic
    from oracle import sag_oracle
    dev = torch.device("cuda:0")
    title, content = synthetic.make_semantic_embeddings(max(n, m), dim, seed=3, clusters=200)
    t, c = torch.from_numpy(title).to(dev), torch.from_numpy(content).to(dev)
    ms, _ = timeit(lambda: construct_SAG.cos_topk_device(t[:n], c[:n], t[:m], c[:m], top_M), iters=3, warm=1)
    flops = 4 * 2.0 * n * m * dim
    t0 = time.perf_counter()
    sag_oracle.generate_cos_similarities(*(torch.from_numpy(x) for x in (title[:cpu_rows], content[:cpu_rows], title[:m], content[:m])), top_M)
    cpu_ms_row = (time.perf_counter() - t0) * 1e3 / cpu_rows
    print(f"SAG cosine top-{top_M + 1}: n={n} x m={m} x dim={dim}: HIP {ms:.1f} ms ({n / ms * 1e3:.0f} news/s, {flops / ms / 1e9:.1f} TFLOP/s fp32-eq)"
          f"   CPU per-news loop {cpu_ms_row:.1f} ms/news ({1e3 / cpu_ms_row:.0f} news/s, {torch.get_num_threads()} threads, {cpu_rows} rows)")
    rng = np.random.default_rng(4)
    ids, cos, length = synthetic.make_similarity_lists(rng, news_num, top_M, isolated_frac=0.02)
    nn = synthetic.news_graph_size(top_M, hop)
    dv = [torch.from_numpy(a).to(dev) for a in (ids, cos, length)]
    ms2, _ = timeit(lambda: construct_SAG.news_graph_device(*dv, top_M=top_M, hop=hop, news_node_num=nn), iters=5, warm=1)
    sub = min(news_num, 4000)
    sub_ids = np.minimum(ids[:sub], sub - 1)
    t0 = time.perf_counter()
    sag_oracle.generate_news_graph(sub_ids, cos[:sub], length[:sub], top_M, hop, nn)
    cpu2 = (time.perf_counter() - t0) * 1e3 / sub
    print(f"SAG walk: {news_num} news, top_M={top_M}, hop={hop}, {nn} nodes: HIP {ms2:.2f} ms ({news_num / ms2 / 1e3:.2f} M news/s)"
          f"   CPU walk {cpu2 * news_num:.0f} ms ({1 / cpu2:.1f} k news/s, 1 thread, {sub} rows)")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "xattn"
    nums = [int(v) for v in sys.argv[2:]]
    if what == "xattn":
        bench_xattn(*nums)
    elif what == "xattn-entry":
        bench_xattn_entry(*nums)
    elif what == "xattn-mind":
        bench_xattn(*nums, density="mind")
    elif what == "msa":
        bench_msa(*nums)
    elif what == "msa-train":
        bench_msa_train(*nums)
    elif what == "sag":
        bench_sag(*nums)
    elif what == "topic":
        bench_topic(*nums)
    elif what == "linear":
        bench_linear(*nums)
