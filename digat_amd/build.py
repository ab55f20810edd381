"""In-tree build of the HIP extension: ``python -m digat_amd.build``.

One translation unit, one shared object: ``digat_amd/lib/libdigat_hip.so`` (git-ignored, but it
travels to the GPU box with the repo snapshot).  hipcc cross-compiles gfx950 without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "digat_kernels.hip")
HEADER = os.path.join(os.path.dirname(HERE), "include", "digat_hip.h")
OUT = os.path.join(HERE, "lib", "libdigat_hip.so")
# -fno-slp-vectorize: hipcc otherwise packs adjacent scalar f32 FMAs into v_pk_fma_f32, which on
# gfx950 issues slower than two v_fma_f32 (MI355X_MICROARCH.md, cycle constants) — measured 487 -> 452 us
# on the Eq. 8 score kernel of that time.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC"]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    csrc = os.path.dirname(SRC)
    sources = [HEADER] + [os.path.join(csrc, n) for n in os.listdir(csrc) if n.endswith((".hip", ".inc", ".h"))]
    return os.path.getmtime(OUT) < max(os.path.getmtime(p) for p in sources)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["hipcc", *FLAGS, "-o", OUT, SRC]
    if verbose:
        print("[digat_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
