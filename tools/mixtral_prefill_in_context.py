"""Where a Mixtral-8x7B prefill chunk spends its time: events around every ext.* call of the chunk (few layers), grouped by op name and GEMM shape.
python tools/mixtral_prefill_in_context.py [layers=2]"""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext, linear, moe_path
from exllamav3_amd.mixtral_path import MIXTRAL_8X7B, SyntheticEXL3Mixtral
dev = torch.device("cuda:0"); ext.init(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mm = SyntheticEXL3Mixtral(MIXTRAL_8X7B, K=4, cb=2, device=dev, layers=L)
mm.prefill_chunk(4096); torch.cuda.synchronize()
rec = []
names = ["hgemm_nt", "gemm_nt_mfma", "gemm_nt_grouped", "index_select_placeholder", "reconstruct_had_multi_t", "reconstruct_had_slice_t", "silu_mul_2d", "silu_mul", "moe_scatter", "routing_std", "rms_norm", "rope_strided", "quant_cache_paged_strided", "hgemm", "had_r_128", "reconstruct"]
orig = {n: getattr(ext, n) for n in names if hasattr(ext, n)}
def wrap(n, f):
    def g(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); r = f(*a, **k); e1.record()
        key = n
        if n == "gemm_nt_grouped": key = f"{n} R={a[0].shape[0]} k={a[0].shape[1]} n={a[1].shape[1]} epi={a[4]}"
        if n in ("hgemm_nt", "gemm_nt_mfma"): key = f"{n} m={a[0].shape[0]} k={a[0].shape[1]} n={a[1].shape[0]} c={str(a[2].dtype)[6:]}"
        rec.append((key, e0, e1)); return r
    return g
for n, f in orig.items(): setattr(ext, n, wrap(n, f))
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); mm.prefill_chunk(4096); t1.record(); torch.cuda.synchronize()
acc = collections.defaultdict(list)
for k, e0, e1 in rec: acc[k].append(e0.elapsed_time(e1) * 1e3)
tot = t0.elapsed_time(t1) * 1e3
print("chunk us per layer", round(tot / L, 1))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("  %-70s calls/layer %5.1f  avg_us %8.1f  us/layer %8.1f" % (k, len(v) / L, sum(v) / len(v), sum(v) / L))
print("  (sum of timed ops per layer: %.1f)" % (sum(sum(v) for v in acc.values()) / L))
