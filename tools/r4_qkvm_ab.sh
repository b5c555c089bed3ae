#!/bin/bash
# linears-only decode step: the q|k|v epilogue inside o_proj's launch (EXL3_HIP_QKV_IN_OPROJ=1, default) against glue_qkv_rs + o_proj (=0); alternations on one box
cd "$(dirname "$0")/.." || exit 1
one() { EXL3_HIP_QKV_IN_OPROJ=$1 python bench.py --no-prefill --no-cpu --no-extra --steps 40 --warmup 5 $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qkv_in_oproj=$1 $2', d['value'], d['ms_per_step'], (d.get('logits_check') or {}).get('ok'))"; }
for i in 1 2 3; do one 0; one 1; done
for i in 1 2; do one 0 "--model llama-3.2-1b"; one 1 "--model llama-3.2-1b"; done
