"""featureAffine-sized launch of the one-strip GEMM tile (M x 400 x 400 with relu + residual through digat_linear_f32x3-like entry is not
exposed; this times digat_linear_f32x3 minus its weight split by timing the split alone too).  python tools/exp/fa_lab.py lib1.so lib2.so"""
import os, sys, subprocess
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch, numpy as np
    sys.path.insert(0, ROOT)
    from digat_amd import _lib
    L = _lib.lib(); dev = torch.device("cuda:0")
    for M in (7000, 18432, 2304):
        N = K = 400
        x = torch.randn(M, K, device=dev) * 0.5; w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev) * 0.1
        y = torch.empty(M, N, device=dev)
        ws = torch.empty(L.digat_split_weights_bytes(N, K), dtype=torch.uint8, device=dev)
        def run():
            _lib.check(L.digat_linear_f32x3(x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, M, N, K, ws.data_ptr(), 1, _lib.stream_ptr()), "x3")
        def split():
            _lib.check(L.digat_split_weights(w.data_ptr(), N, K, ws.data_ptr(), 1, _lib.stream_ptr()), "split")
        out = []
        for fn in (run, split):
            for _ in range(10): fn()
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): fn()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
            out.append(best)
        if hasattr(L, "digat_debug_gemm_timers"):
            import ctypes as C
            L.digat_debug_gemm_timers.argtypes = [C.POINTER(C.c_double)]
            o = (C.c_double * 8)(); L.digat_debug_gemm_timers(o)
            for _ in range(10): run()
            torch.cuda.synchronize(); L.digat_debug_gemm_timers(o); v = list(o); wv = max(v[6], 1); st = max(v[3], 1)
            print(f"    per sampled wave: total {v[7]/wv:9.0f} ticks; epilogue {v[5]/wv:8.0f}; per step: wait {v[0]/st:7.1f} barrier {v[1]/st:7.1f} body {v[2]/st:7.1f} (steps/wave {st/wv:.1f})")
        err = float((y.double() - (x.double() @ w.double().t() + b.double())).abs().max())
        print(f"{os.path.basename(_lib.LIB_PATH):22s} M={M:6d}: gemm+split {out[0]:6.1f} us, split {out[1]:5.1f} us -> gemm ~{out[0]-out[1]:6.1f} us  max err {err:.2e}")
    sys.exit(0)
for lib in sys.argv[1:]:
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, DIGAT_HIP_LIB=os.path.abspath(lib)))
