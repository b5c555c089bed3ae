#!/bin/bash
# bs 1 with decode attention over a 1000-token cache: the q|k|v epilogue inside the attention's context-split launch (EXL3_HIP_ATTN_QKV_IN_SPLIT=1, default) against
# glue_qkv_rs + split as two launches (=0); three alternations on one box
cd "$(dirname "$0")/.." || exit 1
one() { EXL3_HIP_ATTN_QKV_IN_SPLIT=$1 python bench.py --attention --no-prefill --no-cpu --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qkv_in_split=$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do one 0; one 1; done
