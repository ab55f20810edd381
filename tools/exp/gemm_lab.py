"""Projection-GEMM lab: times digat_user_project0 ([M,400] x [1200,400]^T, pre-split weights: the encoder's own launch) for the libraries
given on the command line, and prints the phase timers of experiment builds (-DDIGAT_GEMM_TIMERS).
  python tools/exp/gemm_lab.py [fmt=1] [M=34304] lib1.so lib2.so ..."""
import os, sys, subprocess, json
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    fmt, M = int(sys.argv[2]), int(sys.argv[3])
    import ctypes as C, types, numpy as np, torch
    sys.path.insert(0, ROOT)
    from digat_amd import _lib, synthetic
    from digat_amd.graphEncoders import DIGAT
    dev = torch.device("cuda:0")
    d, Cn, L = 400, 17, 3
    cfg = types.SimpleNamespace(news_graph_size=10, max_history_num=50, category_num=Cn, graph_depth=L, dropout_rate=0.2)
    enc = DIGAT(cfg, d)
    enc.load_state_dict({k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(d, Cn, L, seed=0, bias_std=0.05).items()})
    enc = enc.to(dev).eval()
    enc.projection_mode = "fp16x3" if fmt else "bf16x6"
    P = enc._params()
    X = torch.randn(M, d, device=dev) * 0.5
    out = torch.empty(3, M, d, device=dev)
    Lb = _lib.lib()
    Nn = int(os.environ.get("LAB_N", "1200"))              # LAB_N=400: one [400,400] linear on one-strip tiles (featureAffine's launch)
    if Nn != 1200:
        Wl = torch.randn(Nn, d, device=dev) * 0.05
        bl = torch.randn(Nn, device=dev) * 0.05
        img = torch.empty(Lb.digat_split_weights_bytes(Nn, d), dtype=torch.uint8, device=dev)
        _lib.check(Lb.digat_split_weights(Wl.data_ptr(), Nn, d, img.data_ptr(), fmt, _lib.stream_ptr()), "split")
        out = torch.empty(1, M, Nn, device=dev)
    def run():
        if Nn != 1200:
            _lib.check(Lb.digat_linear_f32x3(X.data_ptr(), d, Wl.data_ptr(), bl.data_ptr(), out.data_ptr(), Nn, M, Nn, d, img.data_ptr(), fmt,
                                             _lib.stream_ptr()), "linear")
        else:
            _lib.check(Lb.digat_user_project0(P, X.data_ptr(), out.data_ptr(), M, _lib.stream_ptr()), "project0")
    for _ in range(20): run()
    torch.cuda.synchronize()
    has_t = hasattr(Lb, "digat_debug_gemm_timers")
    if has_t:
        Lb.digat_debug_gemm_timers.argtypes = [C.POINTER(C.c_double)]
        o = (C.c_double * 8)(); Lb.digat_debug_gemm_timers(o)
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    ref = (X.double() @ enc.user_graph_attention_W[0].weight.double().t() + enc.user_graph_attention_W[0].bias.double()) if Nn == 1200 \
        else (X.double() @ Wl.double().t() + bl.double())
    err = float((out[0].double() - ref).abs().max())
    fl = 2.0 * M * Nn * 400
    msg = f"{os.path.basename(_lib.LIB_PATH):28s} fmt={fmt} M={M} N={Nn}: median {sorted(ts)[2]:7.1f} us  best {min(ts):7.1f} us  {fl / min(ts) / 1e6:6.1f} TF fp32-eq  max err {err:.2e}"
    if has_t:
        Lb.digat_debug_gemm_timers(o); v = list(o); w = max(v[6], 1); st = max(v[3], 1)
        msg += (f"\n    per sampled wave: total {v[7]/w:9.0f} ticks; prologue+loop {(v[7]-v[5])/w:9.0f}; epilogue {v[5]/w:8.0f}; per step: wait {v[0]/st:7.1f} "
                f"barrier {v[1]/st:7.1f} body {v[2]/st:7.1f} (steps/wave {st/w:.1f})")
    print(msg)
    sys.exit(0)
args = sys.argv[1:]
fmt = int(args.pop(0)) if args and args[0].isdigit() else 1
M = int(args.pop(0)) if args and args[0].isdigit() else 34304
for lib in args or [os.path.join(ROOT, "digat_amd", "lib", "libdigat_hip.so")]:
    env = dict(os.environ, DIGAT_HIP_LIB=os.path.abspath(lib))
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(fmt), str(M)], env=env)
