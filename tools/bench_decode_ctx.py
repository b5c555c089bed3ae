"""bs-1 decode step with the attention core at several context lengths (hipGraph replay; STEP=fx (launch-per-op) | fused | persistent (the attention inside the ONE launch)).
python tools/bench_decode_ctx.py [ctx ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama
dev = torch.device("cuda:0")
ctxs = [int(a) for a in sys.argv[1:]] or [1000, 4000, 16000]
STEP = os.environ.get("STEP", "fx")
model = SyntheticEXL3Llama(SHAPES[os.environ.get("MODEL", "llama-3.1-8b")], K=4, cb=2, device=dev, kv_bits=4, max_ctx=max(ctxs) // 256 * 256 + 256)
for wa in (False, True):
    model.with_attention = wa
    for ctx in (ctxs if wa else ctxs[:1]):
        model.alloc_state(1, pos=ctx)
        step = {"fx": model.decode_step_fx, "fused": model.decode_step_fused, "persistent": model.decode_step_persistent}[STEP]
        step(); torch.cuda.synchronize()
        st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                step()
            g.replay(); st.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(30): g.replay()
            e1.record(st); st.synchronize()
        ms = e0.elapsed_time(e1) / 30
        if not wa: base = ms
        if STEP == "persistent" and getattr(model, "_pstep", None) is not None: assert not model._pstep.error(), "time-out in the persistent step"
        print(f"step={STEP} attention={wa} ctx={ctx}: {ms:.4f} ms/step, {1000 / ms:.1f} tok/s" + (f", attention sublayer +{(ms - base) * 1e3 / model.n_layers:.2f} us/layer" if wa else ""), flush=True)
