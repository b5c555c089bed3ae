"""lm_head GEMV (k=4096, n=128256, 4 bpw): PLAIN (pre-rotated input) vs NORM mode, waves per workgroup sweep."""
import sys, torch
sys.path.insert(0, "/root/repo")
from exllamav3_amd import ext
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama
dev = torch.device("cuda:0")
cb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = SyntheticEXL3Llama(SHAPES["llama-3.1-8b"], K=4, cb=cb, device=dev, kv_bits=4, layers=1)
model.alloc_state(1)
model.decode_step_fused(); model.decode_step_fused_v1(); torch.cuda.synchronize()
lm = model.lm_head
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(8): fn()
        g.replay(); st.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(st); g.replay(); e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 8
for mw in (16, 8, 4):
    ext.set_gemv_max_waves(mw)
    a = timeit(lambda: ext.exl3_gemv_ex(None, model.xh3[:1], model.xs3[:1], [lm.trellis], [model.logits], None, [lm.svh], 1, lm.mcg, lm.mul1, ext.GEMV_IN_ROTATED))
    b = timeit(lambda: ext.exl3_gemv_ex_norm(model.x, model.final_norm, model.ss, model.eps, [lm.trellis], [model.logits], [lm.suh], [lm.svh], 1, lm.mcg, lm.mul1, 0))
    print(f"max waves {mw:2d}: PLAIN(rotated) {a:6.1f} us   NORM {b:6.1f} us   ({262.7 / a:.2f} / {262.7 / b:.2f} TB/s)")
