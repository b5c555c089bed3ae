#!/bin/bash
# bs 1 + attention over a 1000-token cache: context splits of the matrix-pipe kernel (EXL3_HIP_ATTN_WIDE_WGS caps workgroups = splits x kv heads x sequences)
cd "$(dirname "$0")/.." || exit 1
one() { EXL3_HIP_ATTN_WIDE_WGS=$1 python bench.py --attention --no-prefill --no-cpu --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgs_cap=$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do one 512; one 64; one 32; one 96; done
