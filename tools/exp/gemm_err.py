import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from digat_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
for (M, N, K, xs) in [(4100, 400, 400, 1.0), (34304, 1200, 400, 1.0), (4100, 400, 400, 0.01), (4100, 400, 400, 100.0)]:
    rng = np.random.default_rng(M + N + K)
    x = (rng.standard_normal((M, K)) * xs).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    want = x.astype(np.float64) @ w.astype(np.float64).T + b
    xd, wd, bd = (torch.from_numpy(a).to(dev) for a in (x, w, b))
    y6 = torch.empty((M, N), device=dev); y32 = torch.empty((M, N), device=dev)
    ws = torch.empty(L.digat_split_weights_bytes(N, K), dtype=torch.uint8, device=dev)
    _lib.check(L.digat_linear_f32x3(xd.data_ptr(), K, wd.data_ptr(), bd.data_ptr(), y6.data_ptr(), N, M, N, K, ws.data_ptr(), int(os.environ.get('FMT', '0')), _lib.stream_ptr()), "x3")
    _lib.check(L.digat_linear_f32(xd.data_ptr(), K, wd.data_ptr(), bd.data_ptr(), y32.data_ptr(), N, M, N, K, _lib.stream_ptr()), "f32")
    yt = torch.addmm(bd, xd, wd.t())
    torch.cuda.synchronize()
    sc = np.abs(want).max()
    for name, y in (("split kernel", y6), ("fp32 MFMA kernel", y32), ("torch addmm (rocBLAS fp32)", yt)):
        e = np.abs(y.cpu().numpy().astype(np.float64) - want)
        print(f"{M}x{N}x{K} x*{xs}: {name:28s} mean {e.mean()/sc:.2e}  max {e.max()/sc:.2e} (of the output scale {sc:.3g})")
