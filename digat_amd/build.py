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


EXT_SRC = os.path.join(HERE, "csrc", "digat_torch_ext.cpp")
EXT_OUT = os.path.join(HERE, "lib", "digat_torch_ext.so")


def build_torch_ext(force: bool = False, verbose: bool = True) -> str:
    """The thin torch extension over the C ABI (csrc/digat_torch_ext.cpp): host code only, compiled with g++ against the installed
    torch's headers and linked to libdigat_hip.so next to it (rpath $ORIGIN).  In-tree, like the HIP library."""
    if not force and os.path.exists(EXT_OUT) and os.path.getmtime(EXT_OUT) >= max(os.path.getmtime(EXT_SRC), os.path.getmtime(HEADER),
                                                                                 os.path.getmtime(OUT)):
        return EXT_OUT
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    tlib = ce.library_paths()[0]
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=digat_torch_ext",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}", "-I/opt/rocm/include"]
    cmd += [EXT_SRC, "-o", EXT_OUT, f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python",
            f"-L{os.path.dirname(OUT)}", "-ldigat_hip", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print("[digat_amd.build]", " ".join(cmd[:6]), "...", EXT_SRC, "->", EXT_OUT, flush=True)
    subprocess.run(cmd, check=True)
    return EXT_OUT


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    csrc = os.path.dirname(SRC)
    sources = [HEADER] + [os.path.join(csrc, n) for n in os.listdir(csrc) if n.endswith((".hip", ".inc", ".h"))]
    return os.path.getmtime(OUT) < max(os.path.getmtime(p) for p in sources)


def build(force: bool = False, verbose: bool = True) -> str:
    if force or needs_build():
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        cmd = ["hipcc", *FLAGS, "-o", OUT, SRC]
        if verbose:
            print("[digat_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    build_torch_ext(force=force, verbose=verbose)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
