#!/usr/bin/env python3
"""
bench.py — decode tok/s (+ prefill tok/s) of the EXL3 quantized-linear hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W        (N > 1: launched under torch.distributed.run)
  * a "step" = one pass of the hot path over one batch of synthetic input = one decode step (bs tokens) through every
    quantized Linear + RMSNorm + RoPE + KV-cache quantized append of Llama-3.1-8B @ 4.0 bpw (BASELINE.json metric /
    configs[2]); random-init packed EXL3 tensors of that architecture, all resident in HBM before the timed region;
  * W untimed warmup steps, then EXACTLY K timed steps bracketed by barrier + torch.cuda.synchronize() on both sides,
    MAX over ranks; rank 0 prints ONE JSON line;
  * N > 1 = tensor parallel (column/row shards + RCCL all-reduce over xGMI): total work is fixed -> "scaling": "strong".

Extra objects on the JSON line (tier contract 4):
  roofline     : the dominant kernel (the fused EXL3 GEMV, all of its launches in a decode step), bound "hbm":
                 achieved = algorithmic bytes per launch / average launch duration, measured live with HIP events on
                 the launch stream; peak = 8000 GB/s (HBM3E spec, MI355X_MICROARCH.md); traffic = PMC FETCH bytes per
                 launch when profiles/traffic.json from a rocprofv3 --pmc pass is present, else null.
  cpu_baseline : the reference's own CPU implementation of the tile format (cpu/moe_mul1.cpp built into oracle/_ref, kind "reference", layer built
                 once, forward calls timed) and, under "torch_port", the SURVEY 8(d) baseline: the oracle's reconstruct + matmul in torch on the
                 host cores, median of 5 after a warm-up, reconstruct and matmul separately; both on a bounded sample, scaled by packed bytes.
  repeat_ms_per_step : the contract's K steps plus two more blocks of K steps (run-to-run spread).
  other_configs: the other BASELINE.json configs that fit one GPU (8B bs 16, 8B bs 1 with decode attention, Llama-3.2-1B bs 1), same timing.
  prefill      : prefill tok/s of a 4096-token chunk through the same linears (reconstruct_had + MFMA GEMM), N = 1 only.
"""
from __future__ import annotations
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4-copy ceiling)
MFMA_PEAK_TFLOPS = 2500.0       # dense fp16/bf16 MFMA peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="llama-3.1-8b")
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--codebook", default="mul1", choices=["3inst", "mcg", "mul1"],
                    help="mul1 = default of new EXL3 conversions (conversion/convert_model.py:49)")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--kv-bits", type=int, default=4)
    ap.add_argument("--layers", type=int, default=0, help="debug: fewer layers (invalidates the number)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-prefill", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE configs (1B bs 1, 8B bs 16, 8B bs 1 with attention) on the JSON line")
    ap.add_argument("--prefill-tokens", type=int, default=4096)
    ap.add_argument("--variant", type=int, default=1)
    ap.add_argument("--attention", action="store_true", help="include decode attention over the quantized KV cache (context = 1000 tokens) in the timed step")
    ap.add_argument("--unfused", action="store_true", help="one launch per reference op instead of the fused pipeline")
    ap.add_argument("--pipeline", choices=["auto", "tail", "glue", "resid", "fx", "unfused", "persistent"], default="auto",
                    help="auto (default): persistent where it applies (batch 1, one rank, mul1 codebook, 4-bit cache, no attention core), else fx; "
                         "persistent: the whole decode step in ONE launch (exl3_pstep.hip); fx (batch <= 4 on one rank, otherwise = glue): residual stream in a 64-bit fixed-point accumulator, o_proj / "
                         "down_proj add into it with integer atomics, 5 launches/layer (fastest measured); glue: deferred-epilogue GEMVs + glue kernels "
                         "(8 launches/layer); tail: sublayer boundaries run inside the GEMV launches (4 launches/layer; the in-kernel cross-workgroup "
                         "hand-off through memory costs more than a launch)")
    return ap.parse_args()


def cpu_baseline(shape, K, cb):
    """SURVEY.md 8(d) CPU baseline: the oracle's reconstruct + matmul in torch on ALL host cores (oracle/exl3_oracle_torch.py, checked bit for
    bit against the numpy oracle), median of 5 runs after a warm-up, reconstruct and matmul reported separately.  Sample = the q (k x k) and gate
    (k x inter) linears of one layer at m = 1 (config 1's 4096 x 4096 tile is the first of them for the 8B shapes); scaled to tok/s by packed bytes.
    Second figure (mul1 only): the reference's own cpu/moe_mul1.cpp with the layer built ONCE outside the timed loop."""
    import torch
    from oracle import exl3_oracle_torch as ot
    s = shape.linear_shapes()
    # thread count: all host cores is the spec, but torch's integer element-wise ops get SLOWER with 256 threads on this box (4.1 s vs 0.4 s
    # for one 4096 x 4096 reconstruct): probe a few pool sizes on the q linear and keep the fastest, stated in `cores`
    ncpu = os.cpu_count() or 1
    probe = {}
    for th in sorted({min(ncpu, t) for t in (8, 32, 64, ncpu)}):
        torch.set_num_threads(th)
        r = ot.time_linear(s["q"][0], s["q"][1], K, cb, m=1, runs=1, seed=7)
        probe[th] = r["reconstruct_s"] + r["matmul_s"]
        if probe[th] > 3.0 and th < ncpu:
            break
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)
    t_rec = t_mm = 0.0
    nbytes = 0
    parts = {}
    t_start = time.perf_counter()
    for name in ("q", "gate"):
        k, n = s[name]
        r = ot.time_linear(k, n, K, cb, m=1, runs=5, seed=7)
        parts[name] = {"shape": [k, n], "reconstruct_ms": round(r["reconstruct_s"] * 1e3, 2), "matmul_ms": round(r["matmul_s"] * 1e3, 3)}
        t_rec += r["reconstruct_s"]; t_mm += r["matmul_s"]; nbytes += r["bytes"]
    scale = shape.decode_bytes_per_token(K) / nbytes
    tok_s = 1.0 / ((t_rec + t_mm) * scale)
    out = {"value": round(tok_s, 4), "unit": "tok/s", "cores": threads, "kind": "port", "host_cores": ncpu,
           "thread_probe_s": {str(k): round(v, 3) for k, v in probe.items()},
           "reconstruct_s_per_token": round(t_rec * scale, 3), "matmul_s_per_token": round(t_mm * scale, 4), "linears": parts,
           "extrapolated": True,
           "sample": f"EXTRAPOLATION from a byte-scaled sample (two linears, not a whole-model run): torch-CPU oracle (reconstruct = trellis decode + tile permute; matmul = had(x*suh) @ W_hat -> had * svh), {threads} threads "
                     f"(fastest of the probed pool sizes on {ncpu} host cores), "
                     f"q and gate linears of one layer ({nbytes / 1e6:.1f} MB packed), median of 5 after 1 warm-up ({time.perf_counter() - t_start:.1f} s "
                     f"in all), scaled by packed bytes to the {shape.decode_bytes_per_token(K) / 1e9:.3f} GB/token model"}
    ref = cpu_baseline_reference(shape, K) if cb == 2 else None
    if ref is not None:
        # the reference's own multi-threaded CPU kernels are the stronger baseline: primary figure; the torch port of the oracle rides along
        ref["torch_port"] = out
        return ref
    return out


def cpu_baseline_reference(shape, K):
    """The reference's OWN CPU implementation of the EXL3 tile format (exllamav3_ext/cpu/moe_mul1.cpp, mul1 codebook), compiled from
    /root/reference into oracle/_ref/ by oracle/build_ref.sh (travels to the GPU box as a prebuilt .so).  One gateless expert =
    up (hidden -> inter) + down (inter -> hidden) of the benchmark model, m = 1, its own thread pool on all host cores, its default
    (fastest available, int8-approximate VNNI where present) ISA tier.  The layer is built ONCE; ~5 s of forward calls are timed; scaled to
    tokens/s by packed-weight bytes.  None if the library is not present."""
    import ctypes
    import numpy as np
    from oracle import exl3_oracle as orc
    path = os.path.join(ROOT, "oracle", "_ref", "libexl3_ref_mul1.so")
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
        lib.ref_mul1_mlp_make.restype = ctypes.c_longlong
        lib.ref_mul1_mlp_forward.argtypes = [ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        lib.ref_mul1_mlp_free.argtypes = [ctypes.c_longlong]
    except (OSError, AttributeError):
        return None
    s = shape.linear_shapes()
    H, I = s["up"]
    I = min(I, 8192)                     # the reference CPU path asserts k <= 8192 (cpu/moe_mul1.cpp:1895); throughput is scaled by bytes
    ut, us, uv = orc.synth_linear(H, I, K, seed=11)
    dt, ds, dv = orc.synth_linear(I, H, K, seed=12)
    x = np.random.default_rng(0).standard_normal((1, H)).astype(np.float16)
    out = np.zeros((1, H), dtype=np.float32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    threads = os.cpu_count() or 1
    h = lib.ref_mul1_mlp_make(P(ut), P(us), P(uv), P(dt), P(ds), P(dv), H, I, K, K)
    if h < 0:
        return None
    call = lambda: lib.ref_mul1_mlp_forward(h, P(x), 1, H, P(out), threads)
    if call() != 0:
        return None
    for _ in range(3):
        call()
    times = []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 5.0 and len(times) < 20000:
        a = time.perf_counter(); call(); times.append(time.perf_counter() - a)
    lib.ref_mul1_mlp_free(h)
    per_call = float(np.median(times))
    bytes_call = 2 * (H * I * K // 8) + 2 * 2 * (H + I)
    tok_s = 1.0 / (per_call * shape.decode_bytes_per_token(K) / bytes_call)
    return {"value": round(tok_s, 3), "unit": "tok/s", "cores": threads, "kind": "reference", "us_per_call_median": round(per_call * 1e6, 1),
            "extrapolated": True,
            "sample": f"EXTRAPOLATION from a byte-scaled sample (one up+down pair, not a whole-model run): reference cpu/moe_mul1.cpp (oracle/_ref, default ISA tier, {threads} threads), layer built once: up+down linears "
                      f"({bytes_call / 1e6:.1f} MB packed) x {len(times)} forward calls, median; scaled by bytes to the "
                      f"{shape.decode_bytes_per_token(K) / 1e9:.3f} GB/token model"}


def dist_backend_name(backend):
    try:
        import torch.distributed as dist
        return str(dist.get_backend()) if dist.is_initialized() else None
    except Exception:
        return None


def persistent_phases(model_name):
    """Per-op phase medians of the persistent step's final kernel (in-kernel stamps of a separate builder-side run of the C++ harness with stamps on: the stamps cost
    ~ 8 % of the step, so they are NOT collected inside the timed region) -- profiles/r06_persistent_layer_phases.json, labelled as such."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r06_persistent_layer_phases.json"))).get(model_name)
        if not d:
            return None
        out = {"source": "NOT measured in this run: in-kernel phase stamps (100 MHz) of the same kernel in a builder-side run with stamps on, profiles/r06_persistent_layer_phases.json",
               "step_us_with_stamps_on": d.get("total_us"), "per_op_median_us": {}}
        for op, rec in d.get("ops", {}).items():
            if op == "head":
                out["per_op_median_us"][op] = {k: round(v, 2) for k, v in rec.items() if isinstance(v, (int, float))}
            else:
                out["per_op_median_us"][op] = {"period": round(rec.get("period_us", 0.0), 2),
                                               **{k: round(v["median_us"], 2) for k, v in rec.items() if isinstance(v, dict) and "median_us" in v}}
        return out
    except Exception:
        return None


def pinned_logits_check(model_name, K, cb, bsz, dev, pipeline, tol=3e-2, ctx=None):
    """Correctness gate of the timed pipeline (VERDICT r3 weak #1 d; replaces `isfinite(logits)`): ONE layer of the benchmark's shape + a 2048-column
    lm_head built from a fixed host seed (SyntheticEXL3Llama.pin_model: the same tensors on every machine) goes through the SAME decode-step function
    bench.py times, eagerly and as a replayed hipGraph, and the logits are compared with the ORACLE's logits for that model, committed as
    tests/golden/bench_pins.json (made by tests/golden/make_bench_pins.py on the CPU; tests/test_bench_pins.py re-derives them from the oracle).
    Bar: max |d| <= 3e-2 * RMS, the end-to-end bar of every pipeline-vs-oracle test.  The oracle itself is never imported here.
    ctx: the pin WITH the decode attention in the step (round 6): the new token behind a host-seeded pre-filled 4-bit cache of `ctx` tokens -- the gate of the
    ..._with_attention_ctx* lines (at 16 000 tokens a context split of the persistent step's attention takes several 128-token steps)."""
    import numpy as np
    import torch
    from exllamav3_amd.llama_path import SyntheticEXL3Llama
    key = SyntheticEXL3Llama.pin_key(model_name, K, cb, bsz, ctx)
    pf = os.path.join(ROOT, "tests", "golden", "bench_pins.json")
    pins = json.load(open(pf))["pins"] if os.path.exists(pf) else {}
    if key not in pins:
        # no oracle pin for this configuration (tests/golden/make_bench_pins.py lists the pinned ones): the timed number is then gated only by the
        # isfinite assertion of the caller -- said loudly in the record
        return {"pinned": False, "key": key, "WARNING": "NO ORACLE PIN for this configuration: the value on this line is gated by isfinite(logits) only; add the "
                                                        "configuration to tests/golden/make_bench_pins.py to pin it"}
    m = SyntheticEXL3Llama.pin_model(model_name, K, cb, dev, bsz, ctx)
    if ctx is not None:
        m.attn_merge_in_oproj_hd64 = True
        cache0 = [(c.clone(), s_.clone()) for c, s_ in m.kcache + m.vcache]           # (every run appends the new token's row: the same bits each time, restored anyway)
    step = {"tail": m.decode_step_tail, "glue": m.decode_step_fused, "resid": m.decode_step_resid, "fx": m.decode_step_fx, "unfused": m.decode_step,
            "persistent": m.decode_step_persistent, "bc_runners": m.decode_step_bc_runners}[pipeline]
    if pipeline == "persistent":
        assert m.persistent_applies(), "bench.py: the persistent step does not cover this configuration"
    ref = np.asarray(pins[key]["logits"], dtype=np.float32).reshape(pins[key]["shape"])
    eager = step().float().cpu().numpy().copy()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            step()
    if ctx is not None:
        for (c, s_), (c0, s0) in zip(m.kcache + m.vcache, cache0): c.copy_(c0); s_.copy_(s0)
    m.logits.zero_(); g.replay(); torch.cuda.synchronize()
    replay = m.logits.float().cpu().numpy()
    rms = float(np.sqrt((ref.astype(np.float64) ** 2).mean()))
    err = float(np.abs(eager - ref).max() / rms) if np.isfinite(eager).all() else float("inf")
    out = {"pinned": True, "key": key, "rel_err_vs_oracle": round(err, 5), "tol": tol, "ok": bool(err < tol),
           "graph_replay_bit_equal": bool(np.array_equal(replay, eager)), "step": step.__name__,
           "note": "max |logits - oracle logits| / RMS over one layer of the benchmark shape + a 2048-column head (host-seeded tensors; oracle values: tests/golden/bench_pins.json)"
                   + ("; the decode attention over a host-seeded %d-token 4-bit cache in the step" % ctx if ctx is not None else "")}
    if ctx is not None and pipeline == "persistent":
        out["attention_splits_tokens_steps"] = list(m._pstep.attn_geometry(ctx + 1))
    if pipeline == "persistent":
        out["edge_timeout"] = bool(m._pstep.error())
        assert not out["edge_timeout"], f"bench.py: the persistent step reported a timed-out edge: {out}"
    del m, g
    torch.cuda.empty_cache()
    assert out["ok"] and out["graph_replay_bit_equal"], f"bench.py: the timed pipeline does not reproduce the oracle's pinned logits: {out}"
    return out


def main():
    args = parse()
    import torch
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama
    from exllamav3_amd.tp import TPBackendRCCL

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py: --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    # one rank per GPU; EXL3_HIP_TP_BACKEND=gloo (test hook) lets several ranks share a device so that the multi-rank control flow of this
    # script can be exercised on a single-GPU box (RCCL refuses two ranks on one device)
    tp_backend = os.environ.get("EXL3_HIP_TP_BACKEND") or None
    dev = torch.device(f"cuda:{local_rank % torch.cuda.device_count() if tp_backend == 'gloo' else local_rank}")
    torch.cuda.set_device(dev)
    backend = TPBackendRCCL(rank, world, dev, backend=tp_backend)
    ext.init(dev.index)
    ext.set_gemv_variant(args.variant)

    from exllamav3_amd.mixtral_path import MIXTRAL_SHAPES, SyntheticEXL3Mixtral
    cb = {"3inst": 0, "mcg": 1, "mul1": 2}[args.codebook]
    is_moe = args.model in MIXTRAL_SHAPES
    if is_moe:
        # BASELINE config 5: Mixtral layer = attention (tensor-parallel) + sparse-MoE block (expert-parallel over the ranks), 4-bit KV
        shape = MIXTRAL_SHAPES[args.model]
        model = SyntheticEXL3Mixtral(shape, K=args.bits, cb=cb, device=dev, backend=backend, kv_bits=args.kv_bits, layers=args.layers or None)
    else:
        shape = SHAPES[args.model]
        model = SyntheticEXL3Llama(shape, K=args.bits, cb=cb, device=dev, backend=backend, kv_bits=args.kv_bits,
                                   layers=args.layers or None)
    model.alloc_state(args.batch)
    model.with_attention = bool(args.attention)
    model.prefill_attention = bool(args.attention)          # the prefill leg then includes dequant_cache_paged + causal attention over the pages

    # ---- decode: eager warm-up (also creates library contexts), graph capture, timed replays
    pipeline = "unfused" if args.unfused else args.pipeline
    if pipeline == "auto":
        # the persistent decode step (ONE launch per step) wherever it applies -- batch 1, one rank, mul1, 4-bit cache, no attention core -- and is not switched
        # off (EXL3_HIP_PSTEP=0); else the fixed-point-residual launch-per-op pipeline
        # N > 1 (round 6): the tensor-parallel step too -- the all-reduce behind o_proj / down_proj happens on the step's row edges (every rank pushes its partial lines into
        # every rank's exchange buffer: ext.PersistentStep(tp=...)).  Only over a real collective backend: ranks that SHARE a GPU (the gloo test hook) cannot both keep a
        # whole-chip grid resident (EXL3_HIP_PSTEP_TP=1 forces it, =0 keeps the launch-per-op TP branch)
        tp_ps = world == 1 or (os.environ.get("EXL3_HIP_PSTEP_TP", "") == "1" or (dist_backend_name(backend) == "nccl" and os.environ.get("EXL3_HIP_PSTEP_TP", "") != "0"))
        pipeline = "persistent" if (not is_moe and tp_ps and model.persistent_applies() and model.persistent is not False) else "fx"
    if pipeline == "tail" and world > 1:
        pipeline = "glue"                                  # TP ranks all-reduce between o/down and the norm ("fx" has its own TP form: llama_path._decode_step_fx_tp)
    fused = pipeline != "unfused"
    run_step = (model.decode_step_fx if pipeline == "fx" else model.decode_step) if is_moe else \
        {"tail": model.decode_step_tail, "glue": model.decode_step_fused, "resid": model.decode_step_resid, "fx": model.decode_step_fx,
         "unfused": model.decode_step, "persistent": model.decode_step_persistent}[pipeline]
    if pipeline == "persistent":
        assert not is_moe and model.persistent_applies(), "--pipeline persistent: batch 1, 4-bit cache, hidden <= 8192"
    # tensor-parallel decode: the o_proj / down_proj all-reduces go through the one-shot IPC push (exl3_allreduce.hip, fused with the residual
    # add) unless EXL3_HIP_TP_ALLREDUCE=rccl; the set-up self-tests against the collective library and every rank falls back together
    ipc_on = False
    if world > 1 and os.environ.get("EXL3_HIP_TP_ALLREDUCE", "ipc") == "ipc":
        ipc_on = backend.enable_ipc_allreduce(max(args.batch, 1) * shape.hidden)

    # safety net of the driver's line: the persistent step needs the whole grid co-resident (one workgroup per CU); should a box ever not give that, its bounded
    # waits time out (flagged, results invalid) -- then the launch-per-op pipeline (the same HIP library, round 4's default) is timed instead, and the line says so
    persistent_fallback = None
    if pipeline == "persistent":
        run_step(); run_step()
        torch.cuda.synchronize()
        ps_bad = model._pstep is None or bool(model._pstep.error())
        if world > 1:
            # the ranks decide together (a rank that keeps the step while a peer left it would wait for lines that never come)
            tb = torch.tensor([1.0 if ps_bad else 0.0], dtype=torch.float64, device=dev)
            backend.all_reduce_max(tb)
            ps_bad = float(tb.item()) != 0.0
        if ps_bad:
            print("bench.py: WARNING: the persistent decode step reported a time-out on this device (grid not co-resident?); timing the launch-per-op fx pipeline instead",
                  file=sys.stderr, flush=True)
            persistent_fallback = "the persistent step timed out in the warm-up on this device: launch-per-op fx pipeline timed"
            pipeline = "fx"
            model.persistent = False
            model._pstep = None
            run_step = model.decode_step_fx

    def capture():
        run_step()
        torch.cuda.synchronize()
        g_ = None
        if args.no_graph:
            return None
        try:
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                run_step()
                st.synchronize()
                g_ = torch.cuda.CUDAGraph()
                # multi-rank: the process group's watchdog thread issues event queries of its own; thread-local capture mode keeps those from
                # invalidating this thread's capture
                with torch.cuda.graph(g_, stream=st, **({"capture_error_mode": "thread_local"} if world > 1 else {})):
                    run_step()
            torch.cuda.synchronize()
        except Exception as e:          # e.g. a collective that cannot be captured: fall back to eager launches
            if rank == 0:
                print(f"bench.py: graph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            g_ = None
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
        if world > 1:
            # all ranks must agree on the mode (a rank replaying a graph while another launches eagerly would deadlock RCCL)
            ok = torch.tensor([1.0 if g_ is not None else 0.0], dtype=torch.float64, device=dev)
            neg = -ok
            backend.all_reduce_max(neg)                 # max(-ok) = -min(ok)
            if float(neg.item()) != -1.0:
                g_ = None
        return g_

    graph = capture()
    if ipc_on:
        # a peer that never arrived leaves the error word set (bounded spins): then every rank drops to RCCL and captures again
        # (collective: error words + epoch lockstep; on failure every rank barriers, unmaps and frees together -- tp.poll_ipc_allreduce)
        if not backend.poll_ipc_allreduce():
            if rank == 0:
                print("bench.py: IPC all-reduce reported a timed-out peer / epoch mismatch; falling back to RCCL all-reduce", file=sys.stderr)
            ipc_on = False
            graph = capture()
    def step():
        if graph is not None:
            graph.replay()
        else:
            run_step()

    for _ in range(args.warmup):
        step()
    backend.fwd_barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    backend.fwd_barrier()
    t1 = time.perf_counter()
    el = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    backend.all_reduce_max(el)
    elapsed = float(el.item())
    ms_per_step = elapsed * 1e3 / args.steps
    tok_s = args.batch * args.steps / elapsed
    assert torch.isfinite(model.logits.float()).all(), "non-finite logits"
    # the pipeline that was just timed, on a pinned one-layer model of the same shape, against the oracle's committed logits (Llama shapes, one rank)
    logits_check = None
    if world == 1 and not is_moe:
        # (--attention: the pin with the decode attention over a host-seeded 1000-token cache in the step)
        logits_check = pinned_logits_check(args.model, args.bits, cb, args.batch, dev, pipeline, ctx=1000 if args.attention else None)
    ipc_fell_back_after_timing = False
    if ipc_on and not backend.poll_ipc_allreduce():
        # a timed-out push poisons its elements with NaN (caught by the assert above when it reaches the logits); reaching this line means the
        # error word / epochs disagree without visible damage: report it, the timed region ran on the IPC path
        ipc_on = False; ipc_fell_back_after_timing = True
        graph = capture()
    # run-to-run spread: two more blocks of K steps, timed the same way (the contract's `value` is the first block above)
    repeat_ms = [round(ms_per_step, 4)]
    for _ in range(2):
        backend.fwd_barrier(); torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(); backend.fwd_barrier()
        eb = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device=dev)
        backend.all_reduce_max(eb)
        repeat_ms.append(round(float(eb.item()) * 1e3 / args.steps, 4))

    # ---- N > 1: latency of one decode all-reduce (batch x hidden fp32 + residual add), IPC push and the collective library, 64 back-to-back calls
    allreduce = None
    if world > 1:
        def time_calls(fn, n=64):
            fn(); torch.cuda.synchronize(); backend.fwd_barrier()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record(); torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) * 1e3 / n], dtype=torch.float64, device=dev)
            backend.all_reduce_max(t)
            return round(float(t.item()), 2)
        yb = torch.randn((args.batch, shape.hidden), device=dev)
        rb = torch.zeros((args.batch, shape.hidden), dtype=torch.half, device=dev)
        sb = torch.zeros((args.batch, shape.hidden // 128), device=dev)
        ipc_requested = os.environ.get("EXL3_HIP_TP_ALLREDUCE", "ipc") == "ipc"
        # per-rank weight bytes streamed per token (the rank's column / row shards + its lm_head shard): gathered so the line shows every rank's share
        my_bytes = torch.tensor([float(sum((k * n * args.bits // 8 + 2 * (k + n)) * cnt for (k, n, cnt) in model.gemv_launches_per_step()))],
                                dtype=torch.float64, device=dev)
        all_bytes = [torch.zeros_like(my_bytes) for _ in range(world)]
        torch.distributed.all_gather(all_bytes, my_bytes)
        try:
            rccl_ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl_ver = None
        allreduce = {"world_size": world, "collective_backend": dist_backend_name(backend), "rccl_version": rccl_ver,
                     "message_bytes": args.batch * shape.hidden * 4, "per_step": 2 * model.n_layers,
                     "ipc_requested": ipc_requested, "ipc_enabled": bool(ipc_on), "ipc_fell_back": bool(ipc_requested and not ipc_on),
                     "ipc_fell_back_after_timing": ipc_fell_back_after_timing,
                     "bytes_per_token_per_rank": [int(b.item()) for b in all_bytes],
                     "path": ("inside the persistent step: every rank pushes the partial lines of its row shards into every rank's exchange buffer (no all-reduce launch)"
                              if pipeline == "persistent" else
                              ("ipc one-shot push + fused residual add (exl3_allreduce.hip)" if ipc_on else "collective library all_reduce + glue_resid")),
                     "inside_persistent_step": pipeline == "persistent"}
        if ipc_on:
            allreduce["ipc_us"] = time_calls(lambda: backend.ipc.reduce(yb, resid=rb, ss_part=sb, m=args.batch))
        saved_ipc, backend.ipc = backend.ipc, None
        allreduce["library_us"] = time_calls(lambda: backend.all_reduce_resid(yb, rb, sb, args.batch))
        backend.ipc = saved_ipc

    # ---- roofline leg: every fused-GEMV launch of a decode step, bracketed by HIP events on the launch stream
    roofline = None
    if is_moe:
        # whole-step figure: the step streams the attention linears + the routed experts' packed weights once (bytes_per_token); per-launch
        # event timing of the indexed exl3_mgemm launches is in tools/bench_moe.py
        bpt = shape.decode_bytes_per_token(args.bits)
        ach = bpt / (ms_per_step * 1e-3) / 1e9 / max(world, 1)
        roofline = {"bound": "hbm", "kernel": "whole decode step (quantized GEMV / indexed exl3_mgemm launches + glue)", "achieved": round(ach, 1),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None,
                    "note": "algorithmic bytes per token per GPU / step time (not a single kernel)"}
    elif rank == 0 or world > 1:
        # GPU-side timing only: the launches are captured into a hipGraph per call type (all layers' instances, i.e.
        # distinct cold weights) and the replay is bracketed by HIP events on the replay stream, so host/ctypes time
        # is excluded; the figure still contains the ~1-2 us inter-kernel gap of back-to-back graph nodes.
        bsz = args.batch
        calls = model.gemv_calls("fx" if pipeline == "persistent" else pipeline)        # (persistent step: the launch-per-op GEMVs of the same shapes, for the per-launch table)
        per_layer = 4
        groups = [calls[i:len(calls) - 1:per_layer] for i in range(per_layer)] + [[calls[-1]] * 4]
        total_us, launches = 0.0, 0
        group_us = []                                       # average microseconds per launch of each call type (q|k|v, o, gate|up, down, lm_head)
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for grp in groups:
                for c in grp:
                    c()
                st.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    for c in grp:
                        c()
                g.replay(); st.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                reps = 5
                e0.record(st)
                for _ in range(reps):
                    g.replay()
                e1.record(st)
                st.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / reps
                if grp is groups[-1]:
                    us, n = us / 4, 1                  # lm_head: 4 replays of the single call
                else:
                    n = len(grp)
                total_us += us; launches += n
                group_us.append(us / n)
        torch.cuda.synchronize()
        ov = 0.0
        K = args.bits
        bytes_step = sum((k * n * K // 8 + 2 * (k + n) + 2 * bsz * (k + n)) * cnt for (k, n, cnt) in model.gemv_launches_per_step())
        launches_step = sum(cnt for (_, _, cnt) in model.gemv_launches_per_step())
        avg_us = total_us / launches
        bytes_per_launch = bytes_step / launches_step
        achieved = bytes_per_launch / avg_us / 1e3                            # GB/s
        # roofline.traffic is NOT measured in this run (a --pmc pass re-runs the step under the profiler: too heavy for the driver line): it is
        # the per-launch FETCH_SIZE of the last committed PMC pass (tools/final_profiles.sh -> profiles/traffic.json), and says so
        traffic = None; traffic_src = None; kernel_only = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                traffic = tj.get("fetch_bytes_per_launch")
                traffic_src = ("NOT measured in this run: per-launch FETCH_SIZE of an earlier rocprofv3 --pmc pass over the same command, "
                               f"profiles/traffic.json (collected {tj.get('collected', 'round 2')}, {tj.get('launches_sampled')} launches)")
                ko = tj.get("kernel_only")
                if ko and world == 1:
                    # rocprofv3 --kernel-trace durations of the same launches (no graph-node gaps): builder-kept, from the profiles/ CSV named
                    kernel_only = dict(ko)
                    kernel_only["frac"] = round(bytes_step / launches_step / (ko["avg_launch_us"] * 1e3) / HBM_PEAK_GBPS, 4)
                    kernel_only["note"] = "NOT measured in this run: average kernel duration of the step's GEMV launches in the committed rocprofv3 stats CSV"
            except Exception:
                traffic = None
        # per call type (VERDICT r4 task 7: the roofline must be re-derivable from this record alone): HIP-event microseconds per launch, algorithmic
        # bytes, GB/s -- and the least-squares fit  t = const_us + bytes / stream_rate  over the five call types
        per_launch, fit = None, None
        if len(group_us) == 5:
            names = ["q|k|v", "o", "gate|up", "down", "lm_head"]
            shp = model.gemv_launches_per_step()
            per_launch = []
            for nm, us_, (k_, n_, cnt_) in zip(names, group_us, shp):
                b_ = k_ * n_ * K // 8 + 2 * (k_ + n_) + 2 * bsz * (k_ + n_)
                per_launch.append({"launch": nm, "k": int(k_), "n": int(n_), "per_step": int(cnt_), "us": round(us_, 2), "bytes": int(b_), "GBps": round(b_ / us_ / 1e3, 1)})
            import numpy as _np
            A_ = _np.array([[1.0, pl["bytes"]] for pl in per_launch]); y_ = _np.array([pl["us"] for pl in per_launch])
            w_ = _np.array([pl["per_step"] for pl in per_launch], dtype=_np.float64) ** 0.5          # weight by how often the launch occurs in a step
            c_, *_ = _np.linalg.lstsq(A_ * w_[:, None], y_ * w_, rcond=None)
            fit = {"const_us": round(float(c_[0]), 2), "stream_TBps": round(1e-6 / float(c_[1]), 3) if c_[1] > 0 else None,
                   "constants_share_of_step": round(float(c_[0]) * launches_step / (ms_per_step * 1e3), 3),
                   "note": "t = const_us + bytes / stream_rate, weighted least squares over the call types above (weights = launches per step); the step also has its glue launches"}
        gemv_roofline = None
        if pipeline == "persistent":
            # the dominant kernel IS the step: exl3_pstep_kernel<K>, one launch per step.  achieved = the step's algorithmic bytes (the same sum as the
            # launch-per-op table: every quantized matrix once + activations) / the launch's duration, HIP events around hipGraph replays of the step on the
            # replay stream (the step's set-up launch fx_init_prep, ~4 us, is inside: it is a node of the same graph)
            stp = torch.cuda.Stream(); stp.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stp):
                model.decode_step_persistent(); stp.synchronize()
                gp = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gp, stream=stp):
                    model.decode_step_persistent()
                gp.replay(); stp.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(stp)
                for _ in range(20): gp.replay()
                e1.record(stp); stp.synchronize()
            step_us = e0.elapsed_time(e1) * 1e3 / 20
            assert not model._pstep.error(), "bench.py: the persistent step reported a timed-out edge"
            gemv_roofline = {"kernel": "exl3_gemv4_kernel, the launch-per-op pipeline's GEMV launches of the same shapes (reference table; NOT the timed step)",
                             "achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBPS, 4), "avg_launch_us": round(avg_us, 2),
                             "bytes_per_launch": int(bytes_per_launch), "launches_per_step": launches_step, "per_launch": per_launch, "fit": fit,
                             "traffic": traffic, "traffic_source": traffic_src, "kernel_only": kernel_only}
            ptraffic = None; ptraffic_src = None
            try:
                tj = json.load(open(tf))
                pj = tj.get("persistent_step", {}).get(args.model)
                if pj:
                    ptraffic = pj.get("fetch_bytes_per_launch")
                    ptraffic_src = f"NOT measured in this run: FETCH_SIZE of exl3_pstep_kernel in an earlier rocprofv3 --pmc pass over the same command, profiles/traffic.json (collected {pj.get('collected')})"
            except Exception:
                pass
            ach_p = bytes_step / step_us / 1e3
            roofline = {"bound": "hbm", "kernel": f"exl3_pstep_kernel<{K}> (the whole decode step in ONE launch: trellis decode + Hadamards + MFMA GEMV of every linear, RMSNorm, q|k|v epilogue, silu*mul, residual adds)",
                        "achieved": round(ach_p, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach_p / HBM_PEAK_GBPS, 4),
                        "traffic": ptraffic, "traffic_source": ptraffic_src,
                        "avg_launch_us": round(step_us, 2), "bytes_per_launch": int(bytes_step), "launches_per_step": 1,
                        "plan": model._pstep.describe(),
                        "phases": persistent_phases(args.model),
                        "launch_per_op_gemv": gemv_roofline,
                        "note": "HIP events around hipGraph replays of the step (one exl3_pstep_kernel launch + the set-up launch); algorithmic bytes = every packed weight once + scales + activations"}
        else:
          roofline = {"bound": "hbm", "kernel": ("exl3_gemv4_kernel" if bsz <= 4 else "exl3_gemm3_kernel") + " (fused trellis decode + Hadamard + MFMA GEMV), all GEMV launches of a decode step",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "kernel_only": kernel_only,
                    "avg_launch_us": round(avg_us, 2), "bytes_per_launch": int(bytes_per_launch),
                    "launches_per_step": launches_step, "per_launch": per_launch, "fit": fit,
                    "note": "HIP events around hipGraph replays of the step's GEMV launches (all layers, cold weights); includes inter-node gaps"
                            + ("" if fused else " and the split-k reduce launch") + "; compare profiles/ for rocprofv3 kernel-only durations"}

    # ---- prefill leg: one chunk through the same linears.  N > 1: every rank runs its column / row shards of the chunk and the (tokens, hidden)
    # partial sums of o_proj / down_proj go through the collective library (RCCL ring: bandwidth matters at 4096 x hidden; reference
    # model/model_tp_backend.py:119-126, method eval/perf.py:36-56); barrier + synchronize on both sides, MAX over ranks like the decode leg
    prefill = None
    if not args.no_prefill:
        toks = args.prefill_tokens
        model.prefill_chunk(toks)                       # warm-up (GEMM autotune, allocator)
        torch.cuda.synchronize(); backend.fwd_barrier()
        # five chunks, each timed on its own (barrier + synchronize on both sides, MAX over ranks): value = the MEDIAN, the spread rides along
        reps = 2
        chunk_s = []
        for _ in range(5):
            t0 = time.perf_counter()
            model.prefill_chunk(toks)
            torch.cuda.synchronize(); backend.fwd_barrier()
            dtt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            backend.all_reduce_max(dtt)
            chunk_s.append(float(dtt.item()))
        dt = sorted(chunk_s)[len(chunk_s) // 2]
        # whole job (all ranks); Mixtral: attention linears + top-2 of 8 experts per token (the grouped-by-expert prefill tier, moe_path.forward_prefill)
        flops = (model.prefill_flops_per_token() * toks) if is_moe else shape.prefill_flops_per_token() * toks * (model.n_layers / shape.layers)
        peak_all = MFMA_PEAK_TFLOPS * world
        prefill = {"metric": "prefill tok/s", "value": round(toks / dt, 1), "unit": "tok/s", "chunk_tokens": toks, "n_gpus": world,
                   "ms_per_chunk": round(dt * 1e3, 2), "repeat_ms_per_chunk": [round(v * 1e3, 2) for v in chunk_s],
                   "roofline": {"bound": "mfma", "achieved": round(flops / dt / 1e12, 1), "peak": peak_all, "unit": "TFLOP/s",
                                "frac": round(flops / dt / 1e12 / peak_all, 4),
                                "note": "linears' 2*k*n flops (whole job) over the whole chunk time (includes reconstruct_had, norms, rope, kv-quant"
                                        + (", the o_proj / down_proj all-reduces" if world > 1 else "") + "); peak = dense fp16 MFMA x n_gpus"}}
        prefill["contraction"] = ("exl3_gemm_nt2_mfma: hand-written NT MFMA GEMM, assembly K-loop, one wave per SIMD (fused residual-add / silu*mul epilogues)"
                                  if ext.gemm_nt_own_default() else "hipBLASLt (EXL3_HIP_GEMM_NT=0)")
        if world == 1 and ext.gemm_nt_own_default():
            # the same chunk through the library GEMM (hipBLASLt + the separate silu * mul pass), same process, right behind the timed one: the number the own kernel replaces
            try:
                ext._GEMM_NT_OWN = False
                model.prefill_chunk(toks); torch.cuda.synchronize()
                ls = []
                for _ in range(3):
                    t0 = time.perf_counter(); model.prefill_chunk(toks); torch.cuda.synchronize(); ls.append(time.perf_counter() - t0)
                ldt = sorted(ls)[1]
                prefill["library_route"] = {"value": round(toks / ldt, 1), "unit": "tok/s", "ms_per_chunk": round(ldt * 1e3, 2), "frac": round(flops / ldt / 1e12 / peak_all, 4),
                                            "note": "ext.hgemm_nt forced to hipBLASLt for every GEMM of the chunk (+ silu_mul_2d as its own pass)"}
            except Exception as e:                      # (the comparison must not take the timed number down)
                prefill["library_route"] = {"error": repr(e)[:200]}
            finally:
                ext._GEMM_NT_OWN = True
        if world > 1:
            # the chunk's exchange step alone: 2 all-reduces per layer of the (tokens, hidden) partial sums, in the dtype the TP branch reduces
            msg = torch.zeros((toks, shape.hidden), dtype=model.prefill_allreduce_dtype(), device=dev)
            backend.all_reduce(msg); torch.cuda.synchronize(); backend.fwd_barrier()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                backend.all_reduce(msg)
            e1.record(); torch.cuda.synchronize()
            tms = torch.tensor([e0.elapsed_time(e1) / 8], dtype=torch.float64, device=dev)
            backend.all_reduce_max(tms)
            prefill["allreduce"] = {"message_bytes": msg.numel() * msg.element_size(), "per_chunk": 2 * model.n_layers, "ms_each": round(float(tms.item()), 4),
                                    "prefill_ms": round(float(tms.item()) * 2 * model.n_layers, 3), "path": "collective library (RCCL) all_reduce",
                                    "note": "back-to-back, not overlapped with compute in the chunk"}

        # the same chunk with the attention core in the timed region (append to the quantized cache, expand the pages, causal attention over them):
        # reported beside the headline, which follows BASELINE.json's linears-only shapes (science/qgemm_benchmark.py)
        if not args.attention and world == 1 and not is_moe:
            model.prefill_attention = True
            model.prefill_chunk(toks); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                model.prefill_chunk(toks)
            torch.cuda.synchronize()
            dta = (time.perf_counter() - t0) / reps
            model.prefill_attention = False
            hd_ = shape.head_dim
            aflops = 4.0 * hd_ * shape.heads_q * (toks * (toks + 1) / 2) * model.n_layers
            prefill["with_attention"] = {"value": round(toks / dta, 1), "unit": "tok/s", "ms_per_chunk": round(dta * 1e3, 2),
                                         "tflops": round((flops + aflops) / dta / 1e12, 1),
                                         "note": "linears + dequant_cache_paged + causal chunk attention (exl3_attn_prefill_paged); flops = linears + causal attention"}

        # MI355X option: reconstructed fp16 W kept resident across chunks (LinearEXL3.cache_reconstructed; 0.5 GB per 8B layer).
        # Reported separately: the chunk above reconstructs every matrix per forward exactly like the reference.
        if world == 1 and not is_moe:
            from exllamav3_amd.linear import LinearEXL3
            LinearEXL3.cache_reconstructed = True
            model.prefill_chunk(toks); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                model.prefill_chunk(toks)
            torch.cuda.synchronize()
            dtc = (time.perf_counter() - t0) / reps
            LinearEXL3.cache_reconstructed = False
            for L in model.layers:
                for lin in L.values():
                    if hasattr(lin, "_w_cache"): del lin._w_cache
            prefill["resident_w_option"] = {"value": round(toks / dtc, 1), "unit": "tok/s", "ms_per_chunk": round(dtc * 1e3, 2),
                                            "tflops": round(flops / dtc / 1e12, 1),
                                            "note": "later chunks with the reconstructed fp16 weights left resident in HBM (not the reference's per-forward reconstruct)"}

    if prefill is not None and world == 1 and not is_moe:
        # where the chunk's time goes: HIP events around every op of ONE more chunk (the events cost a few percent: shares, not the timed number)
        try:
            import collections
            names = ["hgemm_nt", "gemm_nt_mfma", "reconstruct_had_multi_t", "reconstruct_had_slice_t", "silu_mul_2d", "silu_mul", "rms_norm", "rope_strided", "rope",
                     "quant_cache_paged_strided", "quant_cache_paged", "add"]
            from exllamav3_amd import linear as _lin
            orig, recs, depth = {n: getattr(ext, n) for n in names if hasattr(ext, n)}, [], [0]
            def _wrap(n, f):
                def g(*a, **k):
                    if depth[0]: return f(*a, **k)              # (ext.hgemm_nt calls ext.gemm_nt_mfma: count the outer call only)
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    depth[0] += 1
                    try:
                        e0.record(); r = f(*a, **k); e1.record()
                    finally:
                        depth[0] -= 1
                    recs.append((n, e0, e1)); return r
                return g
            for n, f in orig.items(): setattr(ext, n, _wrap(n, f))
            try:
                model.prefill_chunk(toks); torch.cuda.synchronize()
            finally:
                for n, f in orig.items(): setattr(ext, n, f)
            acc = collections.defaultdict(float)
            for n, e0, e1 in recs: acc[n] += e0.elapsed_time(e1) * 1e3
            grp = {"contraction (NT GEMM incl. its fused epilogues)": acc["hgemm_nt"] + acc["gemm_nt_mfma"],
                   "reconstruct_had (W^T of every Linear, per chunk as the reference does)": acc["reconstruct_had_multi_t"] + acc["reconstruct_had_slice_t"],
                   "rms_norm": acc["rms_norm"], "rope + kv-quant": acc["rope_strided"] + acc["rope"] + acc["quant_cache_paged_strided"] + acc["quant_cache_paged"],
                   "silu * mul as its own pass": acc["silu_mul_2d"] + acc["silu_mul"], "residual add as its own pass": acc["add"]}
            tot = sum(grp.values())
            prefill["per_layer_us"] = {k: round(v / model.n_layers, 1) for k, v in grp.items() if v > 0}
            prefill["per_layer_share"] = {k: round(v / tot, 3) for k, v in grp.items() if v > 0}
        except Exception as e:          # (diagnostics only)
            prefill["per_layer_us"] = {"error": repr(e)[:200]}

    # ---- the other BASELINE.json configs that fit one GPU, same process, same timing (hipGraph replay, K steps after W warm-ups)
    extra = None
    if world == 1 and not args.no_extra and args.model == "llama-3.1-8b" and args.batch == 1 and not args.attention and not args.layers:
        def timed_decode(mdl, fn, bsz):
            fn(); torch.cuda.synchronize()
            stx = torch.cuda.Stream(); stx.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stx):
                fn(); stx.synchronize()
                gx = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gx, stream=stx):
                    fn()
            torch.cuda.synchronize()
            for _ in range(args.warmup):
                gx.replay()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(args.steps):
                gx.replay()
            torch.cuda.synchronize()
            msx = (time.perf_counter() - ta) * 1e3 / args.steps
            assert torch.isfinite(mdl.logits.float()).all()
            bpt = mdl.shape.decode_bytes_per_token(mdl.K)
            return {"tok_s": round(bsz * 1e3 / msx, 1), "ms_per_step": round(msx, 4), "batch": bsz,
                    "frac_of_hbm_roofline": round((1e3 / msx) / (HBM_PEAK_GBPS * 1e9 / bpt), 4)}
        pipe_x = "fx" if pipeline == "persistent" else pipeline          # (the other configs run the launch-per-op form of the timed pipeline)
        fx_step_desc = "decode_step_fx: launch-per-op pipeline, 5 launches per layer (fixed-point residual, atomic epilogues)"
        extra = {}
        # config 3, bs 16 (generation-3 GEMM + glue_rotate route)
        model.alloc_state(16)
        extra["llama-3.1-8b_bs16"] = timed_decode(model, model.decode_step_fx if pipe_x == "fx" else model.decode_step_fused, 16)
        # bs 1 with the quant-cache-direct decode attention over a 1000-token context in the step
        model.alloc_state(1)
        model.with_attention = True
        att_persistent = pipeline == "persistent" and model.persistent_applies()         # (the attention INSIDE the persistent step: head_dim 128)
        att_step = lambda mdl: (mdl.decode_step_auto if att_persistent else (mdl.decode_step_fx if pipe_x == "fx" else mdl.decode_step_fused))
        att_desc = ("decode_step_persistent with the decode attention inside o_proj's preparation (one (kv head, context split) item per CU on the service waves, records as tagged lines)"
                    if att_persistent else fx_step_desc + " + the attention core (q|k|v epilogue inside the context-split launch, merge inside o_proj)")
        extra["llama-3.1-8b_bs1_with_attention_ctx1000"] = timed_decode(model, att_step(model), 1)
        extra["llama-3.1-8b_bs1_with_attention_ctx1000"]["step"] = att_desc
        # ... and the same step (attention over the same 1000-token cache) through the reference's RUNNER objects: what the reference stack over the installed stub delivers
        try:
            extra["llama-3.1-8b_bs1_with_attention_ctx1000_through_bc_runners"] = timed_decode(model, model.decode_step_bc_runners, 1)
            extra["llama-3.1-8b_bs1_with_attention_ctx1000_through_bc_runners"].update({
                "step": "per layer: rms_norm, ext.BC_Attention.run, add, rms_norm, ext.BC_GatedMLP.run_bszN, add; BC_LinearEXL3.run for the head -- the launch-per-op kernels "
                        "behind the reference's per-module surface (libtorch/attention.cpp:506, libtorch/mlp.cpp:93), one hipGraph per step",
                "logits_check": pinned_logits_check("llama-3.1-8b", args.bits, cb, 1, dev, "bc_runners", ctx=1000)})
        except Exception as e:
            extra["llama-3.1-8b_bs1_with_attention_ctx1000_through_bc_runners"] = {"error": repr(e)[:300]}
        att_pipe = "persistent" if att_persistent else (pipe_x if pipe_x != "unfused" else "glue")
        extra["llama-3.1-8b_bs1_with_attention_ctx1000"]["logits_check"] = pinned_logits_check("llama-3.1-8b", args.bits, cb, 1, dev, att_pipe, ctx=1000)
        if att_persistent:
            extra["llama-3.1-8b_bs1_with_attention_ctx1000"]["edge_timeout"] = bool(model._pstep.error())
            assert not extra["llama-3.1-8b_bs1_with_attention_ctx1000"]["edge_timeout"], "bench.py: the persistent step (attention inside) reported a time-out"
            model._pstep = None
            extra["llama-3.1-8b_bs1_with_attention_ctx1000_launch_per_op"] = timed_decode(model, model.decode_step_fx, 1)
            extra["llama-3.1-8b_bs1_with_attention_ctx1000_launch_per_op"]["logits_check"] = pinned_logits_check("llama-3.1-8b", args.bits, cb, 1, dev, "fx", ctx=1000)
        # ... and over a 16 000-token context (VERDICT r4 task 4: the long-context cost of the quantized-cache attention on the driver line)
        try:
            import copy as _copy
            m16 = SyntheticEXL3Llama(shape, K=args.bits, cb=cb, device=dev, backend=backend, kv_bits=args.kv_bits, max_ctx=16384)
            m16.layers, m16.lm_head, m16.final_norm = model.layers, model.lm_head, model.final_norm      # the same weights: only the cache is longer
            m16.alloc_state(1, pos=16000)
            m16.with_attention = True
            extra["llama-3.1-8b_bs1_with_attention_ctx16000"] = timed_decode(m16, att_step(m16), 1)
            extra["llama-3.1-8b_bs1_with_attention_ctx16000"]["step"] = att_desc
            extra["llama-3.1-8b_bs1_with_attention_ctx16000"]["logits_check"] = pinned_logits_check("llama-3.1-8b", args.bits, cb, 1, dev, att_pipe, ctx=16000)
            if att_persistent:
                extra["llama-3.1-8b_bs1_with_attention_ctx16000"]["edge_timeout"] = bool(m16._pstep.error())
                assert not extra["llama-3.1-8b_bs1_with_attention_ctx16000"]["edge_timeout"], "bench.py: the persistent step (attention inside) reported a time-out"
                m16._pstep = None
                extra["llama-3.1-8b_bs1_with_attention_ctx16000_launch_per_op"] = timed_decode(m16, m16.decode_step_fx, 1)
                extra["llama-3.1-8b_bs1_with_attention_ctx16000_launch_per_op"]["logits_check"] = pinned_logits_check("llama-3.1-8b", args.bits, cb, 1, dev, "fx", ctx=16000)
            kvb = 2 * 16000 * model.hkv * shape.head_dim * args.kv_bits // 8 + 2 * 16000 * model.hkv * shape.head_dim // 32 * 2
            d_ms = extra["llama-3.1-8b_bs1_with_attention_ctx16000"]["ms_per_step"] - ms_per_step
            extra["llama-3.1-8b_bs1_with_attention_ctx16000"].update({
                "cache_bytes_per_layer": int(kvb), "attention_sublayer_us_per_layer": round(d_ms * 1e3 / model.n_layers, 2),
                "cache_words_frac_of_hbm": round(kvb / (d_ms * 1e-3 / model.n_layers) / 1e9 / HBM_PEAK_GBPS, 4) if d_ms > 0 else None,
                "note": "attention_sublayer_us_per_layer = (this step - the headline step without attention) / layers: the q|k|v epilogue + context-split kernel + merge inside "
                        "o_proj; cache_words_frac_of_hbm = the layer's quantized K / V words and scales over that time, against 8 TB/s"})
            del m16
            torch.cuda.empty_cache()
        except Exception as e:          # (an out-of-memory on a shared box must not take the headline down)
            extra["llama-3.1-8b_bs1_with_attention_ctx16000"] = {"error": repr(e)[:200]}
        model.with_attention = False
        model._pstep = None
        # bs 1 with the EXACT GEMV variant (MFMA operands = the reference's fp16-rounded weights bit for bit; the headline runs the default variant,
        # unrounded lo + hi / raw byte sums, inside the same 1e-2 bound)
        if args.variant != 0:
            ext.set_gemv_variant(0)
            extra["llama-3.1-8b_bs1_gemv_variant0_exact"] = timed_decode(model, model.decode_step_fx if pipe_x == "fx" else model.decode_step_fused, 1)
            ext.set_gemv_variant(args.variant)
        extra["llama-3.1-8b_bs16"]["logits_check"] = pinned_logits_check("llama-3.1-8b", args.bits, cb, 16, dev, pipe_x if pipe_x != "unfused" else "glue")
        # the headline is the persistent step: the launch-per-op fx pipeline on the same tensors beside it (same gate) ...
        if pipeline == "persistent":
            model._pstep = None
            extra["llama-3.1-8b_bs1_launch_per_op"] = timed_decode(model, model.decode_step_fx, 1)
            extra["llama-3.1-8b_bs1_launch_per_op"]["step"] = fx_step_desc
            extra["llama-3.1-8b_bs1_launch_per_op"]["logits_check"] = pinned_logits_check("llama-3.1-8b", args.bits, cb, 1, dev, "fx")
        # ... or the other way round (--pipeline fx / EXL3_HIP_PSTEP=0)
        elif pipe_x == "fx" and cb == 2 and model.persistent_applies():
            extra["llama-3.1-8b_bs1_persistent_step"] = timed_decode(model, model.decode_step_persistent, 1)
            extra["llama-3.1-8b_bs1_persistent_step"]["edge_timeout"] = bool(model._pstep.error())
            extra["llama-3.1-8b_bs1_persistent_step"]["plan"] = model._pstep.describe()
            extra["llama-3.1-8b_bs1_persistent_step"]["logits_check"] = pinned_logits_check("llama-3.1-8b", args.bits, cb, 1, dev, "persistent")
            model._pstep = None
        # the 3INST codebook (the older public quants, quant/codebook.cuh:56-90; SURVEY.md 8d "run 3INST and mul1"): same shapes, same pipeline
        if cb != 0:
            m3 = SyntheticEXL3Llama(shape, K=args.bits, cb=0, device=dev, backend=backend, kv_bits=args.kv_bits)
            m3.alloc_state(1)
            p3 = pipeline == "persistent" and m3.persistent_applies()          # (round 6: the persistent step takes the 3INST / mcg codebooks: exl3_pstep.kspec.hip)
            extra["llama-3.1-8b_bs1_3inst"] = timed_decode(m3, m3.decode_step_auto if p3 else (m3.decode_step_fx if pipe_x == "fx" else m3.decode_step_fused), 1)
            if p3 and m3._pstep is not None:
                extra["llama-3.1-8b_bs1_3inst"]["step"] = "decode_step_persistent (exact fp16 3INST weights per quad operand)"
                extra["llama-3.1-8b_bs1_3inst"]["plan"] = m3._pstep.describe()
                extra["llama-3.1-8b_bs1_3inst"]["edge_timeout"] = bool(m3._pstep.error())
                assert not extra["llama-3.1-8b_bs1_3inst"]["edge_timeout"], "bench.py: the persistent step (3INST) reported a time-out"
                extra["llama-3.1-8b_bs1_3inst"]["logits_check"] = pinned_logits_check("llama-3.1-8b", args.bits, 0, 1, dev, "persistent")
                m3._pstep = None
                extra["llama-3.1-8b_bs1_3inst_launch_per_op"] = timed_decode(m3, m3.decode_step_fx, 1)
                extra["llama-3.1-8b_bs1_3inst_launch_per_op"]["logits_check"] = pinned_logits_check("llama-3.1-8b", args.bits, 0, 1, dev, "fx")
            else:
                extra["llama-3.1-8b_bs1_3inst"]["logits_check"] = pinned_logits_check("llama-3.1-8b", args.bits, 0, 1, dev, pipe_x if pipe_x != "unfused" else "glue")
            del m3
            torch.cuda.empty_cache()
        # a fractional-bpw checkpoint (4.5 bpw-style: the reference's allocator bumps whole qgroups by one bit, conversion/allocation.py:131-141): half the layers' MLPs
        # one bit up, the lm_head at 6 bits -- the persistent step runs the ops in runs of equal width (exl3_pstep_kernel<K, K + 1, 6, ...>)
        if pipeline == "persistent" and cb == 2 and args.bits <= 5:
            try:
                kb = args.bits
                mm = SyntheticEXL3Llama(shape, K=kb, cb=cb, device=dev, backend=backend, kv_bits=args.kv_bits, head_K=6,
                                        layer_K=lambda li, g: kb + 1 if (li % 2 == 1 and g in ("gu", "d")) else kb)
                mm.alloc_state(1)
                if mm.persistent_applies():
                    key = "llama-3.1-8b_bs1_mixed_%d_%dbit_head6" % (kb, kb + 1)
                    extra[key] = timed_decode(mm, mm.decode_step_auto, 1)
                    bpm = sum((k_ * n_ * l_.K // 8 + 2 * (k_ + n_)) for L in mm.layers for nm, l_ in L.items() if hasattr(l_, "trellis")
                              for (k_, n_) in [(l_.trellis.shape[0] * 16, l_.trellis.shape[1] * 16)]) + shape.hidden * shape.vocab * 6 // 8
                    extra[key].update({"bytes_per_token": int(bpm), "frac_of_hbm_roofline": round((1e3 / extra[key]["ms_per_step"]) / (HBM_PEAK_GBPS * 1e9 / bpm), 4),
                                       "plan": mm._pstep.describe() if mm._pstep is not None else None, "edge_timeout": bool(mm._pstep.error()) if mm._pstep is not None else None,
                                       "note": "odd layers' gate|up and down at %d bits, everything else at %d, lm_head at 6; gated by the GPU suite's oracle test "
                                               "(tests/test_gpu_pstep.py::test_persistent_step_two_adjacent_widths_in_the_layers), no bench pin" % (kb + 1, kb)})
                    mm._pstep = None
                    extra[key + "_launch_per_op"] = timed_decode(mm, mm.decode_step_fx, 1)
                del mm
                torch.cuda.empty_cache()
            except Exception as e:
                extra["llama-3.1-8b_bs1_mixed_width"] = {"error": repr(e)[:200]}
        # config 2: Llama-3.2-1B, bs 1
        m1 = SyntheticEXL3Llama(SHAPES["llama-3.2-1b"], K=args.bits, cb=cb, device=dev, backend=backend, kv_bits=args.kv_bits)
        m1.alloc_state(1)
        if pipe_x == "fx" and m1.persistent_applies() and (m1.persistent is None or m1.persistent):
            # the persistent decode step (one launch per step: exl3_pstep.hip) is this model's default -- the launch-per-op fx pipeline beside it
            extra["llama-3.2-1b_bs1"] = timed_decode(m1, m1.decode_step_auto, 1)
            extra["llama-3.2-1b_bs1"]["step"] = "decode_step_persistent (one launch per step; 12 streaming + 4 service waves per CU, decode-ahead, tagged slab / row lines)"
            extra["llama-3.2-1b_bs1"]["plan"] = m1._pstep.describe()
            extra["llama-3.2-1b_bs1"]["edge_timeout"] = bool(m1._pstep.error())
            extra["llama-3.2-1b_bs1"]["logits_check"] = pinned_logits_check("llama-3.2-1b", args.bits, cb, 1, dev, "persistent")
            extra["llama-3.2-1b_bs1_launch_per_op"] = timed_decode(m1, m1.decode_step_fx, 1)
            extra["llama-3.2-1b_bs1_launch_per_op"]["logits_check"] = pinned_logits_check("llama-3.2-1b", args.bits, cb, 1, dev, "fx")
        else:
            extra["llama-3.2-1b_bs1"] = timed_decode(m1, m1.decode_step_fx if pipe_x == "fx" else m1.decode_step_fused, 1)
            extra["llama-3.2-1b_bs1"]["logits_check"] = pinned_logits_check("llama-3.2-1b", args.bits, cb, 1, dev, pipe_x if pipe_x != "unfused" else "glue")
        if pipe_x == "fx":
            # ... and with the decode attention over a 1000-token 4-bit cache (head_dim 64: two kv heads per 128-value block of the matrix-pipe split kernel)
            m1.with_attention = True
            m1._pstep = None
            att1 = pipeline == "persistent" and m1.persistent_applies()
            extra["llama-3.2-1b_bs1_with_attention_ctx1000"] = timed_decode(m1, m1.decode_step_auto if att1 else m1.decode_step_fx, 1)
            extra["llama-3.2-1b_bs1_with_attention_ctx1000"]["logits_check"] = pinned_logits_check("llama-3.2-1b", args.bits, cb, 1, dev, "persistent" if att1 else "fx", ctx=1000)
            if att1:
                extra["llama-3.2-1b_bs1_with_attention_ctx1000"]["step"] = att_desc
                extra["llama-3.2-1b_bs1_with_attention_ctx1000"]["edge_timeout"] = bool(m1._pstep.error())
                assert not extra["llama-3.2-1b_bs1_with_attention_ctx1000"]["edge_timeout"], "bench.py: the persistent step (attention inside) reported a time-out"
                m1._pstep = None
                extra["llama-3.2-1b_bs1_with_attention_ctx1000_launch_per_op"] = timed_decode(m1, m1.decode_step_fx, 1)
                extra["llama-3.2-1b_bs1_with_attention_ctx1000_launch_per_op"]["logits_check"] = pinned_logits_check("llama-3.2-1b", args.bits, cb, 1, dev, "fx", ctx=1000)
            m1.with_attention = False
        del m1
        torch.cuda.empty_cache()
        # config 4's compute leg: ONE rank of Llama-3.1-70B at 3 bpw under TP = 8 (q 8192->1024, k/v 8192->128, o 1024->8192, gate/up 8192->3584,
        # down 3584->8192, 16128-column head shard; all 80 layers), collectives replaced by no-ops (tp.OneRankOfMany) -- the part of config 4 one GPU can
        # measure (modules/quant/exl3.py:284-330).  The 160 all-reduces per token are NOT in the step time; an estimate from the one-GPU two-process
        # measurement of the IPC push (6.7 us per all-reduce + residual add) is stated beside it, labelled as such.
        from exllamav3_amd.tp import OneRankOfMany
        m70 = SyntheticEXL3Llama(SHAPES["llama-3.1-70b"], K=3, cb=cb, device=dev, backend=OneRankOfMany(8, dev), kv_bits=args.kv_bits)
        m70.alloc_state(1)
        p70 = pipeline == "persistent" and m70.persistent_applies()            # (round 6: the persistent step takes rows up to 8192 wide)
        r70 = timed_decode(m70, m70.decode_step_auto if p70 else (m70.decode_step_fx if pipe_x == "fx" else m70.decode_step_fused), 1)
        if p70 and m70._pstep is not None:
            r70["step"] = "decode_step_persistent over the rank's shards (one launch per step; the o_proj / down_proj partial rows are NOT exchanged: compute leg only)"
            r70["plan"] = m70._pstep.describe()
            r70["edge_timeout"] = bool(m70._pstep.error())
            assert not r70["edge_timeout"], "bench.py: the persistent step (70B rank shapes) reported a time-out"
            m70._pstep = None
            extra["llama-3.1-70b_tp8_rank_bs1_launch_per_op"] = timed_decode(m70, m70.decode_step_fx, 1)
            extra["llama-3.1-70b_tp8_rank_bs1_launch_per_op"]["launches_per_layer"] = 7
        else:
            r70["launches_per_layer"] = 7 if pipe_x == "fx" else 10
        rank_bytes = sum((k * n * 3 // 8 + 2 * (k + n)) * cnt for (k, n, cnt) in m70.gemv_launches_per_step())
        r70.update({"bits": 3, "tp": 8, "rank_bytes_per_token": int(rank_bytes),
                    "frac_of_hbm_roofline": round((1e3 / r70["ms_per_step"]) / (HBM_PEAK_GBPS * 1e9 / rank_bytes), 4),
                    "allreduce_estimate_ms": round(160 * 6.7e-3, 3),
                    "tok_s_with_allreduce_estimate": round(1e3 / (r70["ms_per_step"] + 160 * 6.7e-3), 1),
                    "note": "one rank's compute leg only (the TP form of the timed pipeline with the exchange left out of its all-reduce launches: tp.OneRankOfMany); allreduce_estimate_ms = 160 x 6.7 us, the IPC push + residual add measured between two "
                            "processes on ONE GPU -- an ESTIMATE, no xGMI link was crossed; frac_of_hbm_roofline uses the rank's own bytes"})
        extra["llama-3.1-70b_tp8_rank_bs1"] = r70
        del m70
        torch.cuda.empty_cache()
        if cb == 2:
            # the same kernels at Llama-3.1-70B's FULL width against the oracle's pin (one layer + a 2048-column head on one rank; the rank-shape step above
            # has no single-GPU oracle value: its collectives are no-ops)
            r70["logits_check_full_width_layer"] = pinned_logits_check("llama-3.1-70b", 3, cb, 1, dev, pipe_x if pipe_x != "unfused" else "glue")
        # config 5 on one GPU: Mixtral 8x7B (23 GB of packed weights), bs 1, 4-bit KV; `--gpus 2 --model mixtral-8x7b` runs it TP = 2 / EP = 2
        from exllamav3_amd.mixtral_path import MIXTRAL_8X7B
        mm = SyntheticEXL3Mixtral(MIXTRAL_8X7B, K=args.bits, cb=cb, device=dev, backend=backend, kv_bits=args.kv_bits)
        mm.alloc_state(1)
        extra["mixtral-8x7b_bs1"] = timed_decode(mm, mm.decode_step_fx if pipe_x == "fx" else mm.decode_step, 1)
        if not args.no_prefill:
            # config 5's prefill leg: one chunk through the attention linears + the grouped-by-expert MoE tier (moe_path.forward_prefill); median of 3
            toks_m = args.prefill_tokens
            mm.prefill_chunk(toks_m); torch.cuda.synchronize()
            cs = []
            for _ in range(3):
                t0 = time.perf_counter(); mm.prefill_chunk(toks_m); torch.cuda.synchronize(); cs.append(time.perf_counter() - t0)
            dtm = sorted(cs)[1]
            fl = mm.prefill_flops_per_token() * toks_m
            extra["mixtral-8x7b_prefill"] = {"tok_s": round(toks_m / dtm, 1), "chunk_tokens": toks_m, "ms_per_chunk": round(dtm * 1e3, 2),
                                             "repeat_ms_per_chunk": [round(v * 1e3, 2) for v in cs], "tflops": round(fl / dtm / 1e12, 1),
                                             "frac_of_mfma_peak": round(fl / dtm / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                             "note": "routed flops (attention linears + top-2 of 8 experts per token) over the whole chunk time"}
        del mm
        torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(shape, args.bits, cb)

    # the fx pipeline's launches per layer: 6; silu(g) * u inside the down launch: -1; without the attention core the q|k|v epilogue inside o_proj's launch: -1
    fx_act_in = bool(getattr(model, "fx_act_in_gemv", False) or getattr(model, "fx_gu_atomic", False))
    fx_qkvm = bool(getattr(model, "qkv_in_oproj", False)) and not args.attention and not getattr(model, "fx_gu_atomic", False) and args.kv_bits == 4 and not is_moe
    fx_desc = ("fixed-point-residual pipeline (%d launches/layer: o_proj / down_proj add into a 64-bit fixed-point residual with integer atomics, no split-k-reduce / "
               "residual launches%s%s)" % (6 - int(fx_act_in) - int(fx_qkvm), "; silu(g) * u formed inside the down launch" if fx_act_in else "",
                                           "; the q|k|v epilogue (RoPE, K / V append) runs inside o_proj's launch" if fx_qkvm else ""))
    if rank == 0:
        out = {
            "metric": ("decode tok/s, Mixtral-8x7B EXL3 %d.0bpw hot path (attention linears + top-2 of 8 experts per layer + RMSNorm + RoPE + KV-quant), bs=%d" % (args.bits, args.batch)) if is_moe else "decode tok/s, %s EXL3 %d.0bpw hot path (all quantized linears + RMSNorm + RoPE + KV-quant), bs=%d" % (
                {"llama-3.1-8b": "Llama-3.1-8B", "llama-3.2-1b": "Llama-3.2-1B", "llama-3.1-70b": "Llama-3.1-70B"}.get(shape.name, shape.name), args.bits, args.batch),
            "value": round(tok_s, 2), "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{shape.name} EXL3 {args.bits}.0bpw {args.codebook} codebook, decode bs={args.batch}, "
                                   f"{model.n_layers} layers, {'attention TP=%d + expert-parallel MoE over %d rank(s)' % (world, world) if is_moe else 'TP=%d' % world}, {args.kv_bits}-bit KV append, "
                                   f"{'hipGraph replay' if graph is not None else 'eager launches'}, { 'glue pipeline + indexed exl3_mgemm MoE block' if is_moe else {'tail': 'tail-epilogue pipeline (4 launches/layer)', 'glue': 'fused glue pipeline (%d launches/layer)' % (7 if (args.batch == 1 and model.act_in_gemv and world == 1) else ((8 if (world == 1 and model.fold_rotate) else 10) if args.batch > 4 else 8)), 'resid': 'resid-in-GEMV pipeline (5 launches/layer: residual add + RMSNorm finished inside the consumer GEMV)' if (args.batch <= 4 and world == 1) else 'fused glue pipeline', 'fx': fx_desc if (args.batch <= model.fx_max_bsz and world == 1) else 'fused glue pipeline', 'unfused': 'one launch per reference op', 'persistent': 'persistent decode step: ONE launch per step (exl3_pstep.hip: 12 streaming + 4 service waves per CU, three decode-ahead units, tagged slab lines, direct residual edges; residual kept in fp32)'}[pipeline] }; "
                                   f"{'attention core INCLUDED: quant-cache-direct decode attention over a 1000-token context' if args.attention else 'attention core excluded (SURVEY.md 2.1)'}",
                       "bytes_per_token": shape.decode_bytes_per_token(args.bits), "hbm_roofline_tok_s": round(HBM_PEAK_GBPS * 1e9 / shape.decode_bytes_per_token(args.bits), 1),
                       "frac_of_hbm_roofline": round(tok_s / args.batch / (HBM_PEAK_GBPS * 1e9 / shape.decode_bytes_per_token(args.bits)), 4),
                       "parallelism": f"tp{world}", "gemv_variant": args.variant},
            "repeat_ms_per_step": repeat_ms, "allreduce": allreduce, "persistent_fallback": persistent_fallback,
            "logits_check": logits_check,
            "roofline": roofline, "cpu_baseline": cpu, "prefill": prefill, "other_configs": extra,
        }
        print(json.dumps(out), flush=True)
    backend.close()


if __name__ == "__main__":
    main()
