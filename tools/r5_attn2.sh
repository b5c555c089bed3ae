#!/bin/bash
# decode attention: 4 / 8 / 16 waves per workgroup of the matrix-pipe kernel (tests, then the whole step at 1000 / 4000 / 16 000 tokens, alternating)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rope_cache.py tests/test_gpu_path.py -q -m gpu -x -k "long_context_kernel or eight_and_sixteen or qkv_in_split or merge_inside_oproj" 2>&1 | tail -3
for r in 1 2; do for nw in 4 8; do
  echo "== NW=$nw round $r"; EXL3_HIP_ATTN_WIDE_NW=$nw timeout 300 python tools/bench_decode_ctx.py 1000 4000 16000 2>&1 | tail -4
done; done | tee $O/attn_nw.txt
