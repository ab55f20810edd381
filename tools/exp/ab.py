#!/usr/bin/env python3
"""A/B of library builds / environment switches on the default bench workload, alternating runs (run on the GPU box):

  python tools/exp/ab.py [--runs 2] [--bench "extra bench args"] name1:VAR=V,VAR2=V2 name2:DIGAT_HIP_LIB=tools/exp/lib_x.so ...

Each configuration is one `python bench.py` child per run (short: 60 timed steps, no CPU leg, no extras); prints step time,
throughput, per-kind solo times and the three Eq. 8 kernels' solo launch times from the run's detail document."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    args = sys.argv[1:]
    runs, extra = 2, []
    while args and args[0].startswith("--"):
        if args[0] == "--runs":
            runs = int(args[1]); args = args[2:]
        elif args[0] == "--bench":
            extra = args[1].split(); args = args[2:]
        else:
            raise SystemExit("unknown option " + args[0])
    configs = []
    for a in args:
        name, _, envs = a.partition(":")
        configs.append((name, dict(kv.split("=", 1) for kv in envs.split(",") if kv)))
    for r in range(runs):
        for name, env in configs:
            with tempfile.NamedTemporaryFile(suffix=".json") as f:
                cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "10", "--cpu-rows", "0", "--extra-steps", "0",
                       "--e2e-impressions", "0", "--detail", f.name] + extra
                res = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True)
                if res.returncode != 0:
                    print(f"{name:16s} FAILED rc={res.returncode}: {res.stderr[-300:]}")
                    continue
                j = json.load(open(f.name))
            iso = j.get("kernel_ms_per_step_single_stream") or {}
            parts = (j.get("roofline_xattn") or {}).get("parts") or {}
            print(f"{name:16s} ms/step {j['ms_per_step']:.4f}  imp/s {j['value']:.0f}  lanes {j['batches_in_flight']}  solo "
                  + " ".join(f"{k}={v:.3f}" for k, v in iso.items())
                  + "  | eq8 solo us " + " ".join(f"{k}={v.get('isolated_avg_launch_us', 0):.1f}" for k, v in parts.items())
                  + "  | in-region us " + " ".join(f"{k}={v.get('avg_launch_us_overlapped', v['avg_launch_us']):.1f}" for k, v in parts.items()), flush=True)


if __name__ == "__main__":
    main()
