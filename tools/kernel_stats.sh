#!/bin/bash
# Register / LDS / spill figures of every kernel (device code only): tools/kernel_stats.sh [name filter]
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/digat_dev.co
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize --offload-device-only -c -o $OUT digat_amd/csrc/digat_kernels.hip ${DIGAT_EXTRA_FLAGS}
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$OUT --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$OUT.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $OUT.elf | python3 -c "
import sys, re
txt = sys.stdin.read()
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for blk in txt.split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk)
    if not name: continue
    g = lambda k: (re.search(r'\.' + k + r':\s+(\d+)', blk) or [None, '?'])[1]
    import subprocess
    n = subprocess.run(['c++filt', name.group(1)], capture_output=True, text=True).stdout.strip()
    n = re.sub(r'\(.*', '', n)
    if flt in n:
        print(f'{n:70s} vgpr {g(\"vgpr_count\"):>4} sgpr {g(\"sgpr_count\"):>4} spill {g(\"vgpr_spill_count\"):>3} lds {g(\"group_segment_fixed_size\"):>6} scratch {g(\"private_segment_fixed_size\"):>4}')
" "$1"
