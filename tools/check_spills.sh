#!/bin/bash
# every instantiation of the kernel-spec translation units, K = 1..8: no VGPR / SGPR spill, no private segment (tests/test_isa_guards.py checks K = 4)
cd "$(dirname "$0")/.."; mkdir -p build/tmp
for f in exl3_gemm3 exl3_gemv4 exl3_gemv2; do for K in 1 2 3 4 5 6 7 8; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=off -Iinclude -DG2_K=$K -S --cuda-device-only -o build/tmp/${f}_k$K.s exllamav3_amd/csrc/$f.kspec.hip 2>/dev/null
  python3 - <<PY
import re
t = open('build/tmp/${f}_k$K.s').read()
n = bad = 0
for b in t.split('  - .agpr_count:')[1:]:
    n += 1
    v = [int(re.search(r'\.' + k + r':\s+(\d+)', b).group(1)) for k in ('vgpr_spill_count', 'sgpr_spill_count', 'private_segment_fixed_size')]
    if any(v): bad += 1; print('  SCRATCH', re.search(r'\.name:\s+(\S+)', b).group(1), v)
print('$f K=$K kernels', n, 'with scratch', bad)
PY
done; done
