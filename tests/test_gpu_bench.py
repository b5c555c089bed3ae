"""bench.py's contract on the GPU box: one JSON line with the required keys, single process and as two torch.distributed ranks."""
import json
import os
import socket
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "cpu_baseline", "repeat_ms_per_step", "other_configs"}


def _line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_process_line(dev):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--layers", "2", "--steps", "3", "--warmup", "1", "--no-prefill", "--no-cpu"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "workload" in d["config"]
    # the default step of this configuration is the persistent one: ONE launch per step is the dominant kernel, the launch-per-op GEMV table rides along
    assert rf["launches_per_step"] == 1 and "exl3_pstep_kernel" in rf["kernel"] and len(rf["launch_per_op_gemv"]["per_launch"]) == 5
    assert d["logits_check"]["step"] == "decode_step_persistent" and not d["logits_check"]["edge_timeout"]
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--layers", "2", "--steps", "3", "--warmup", "1", "--no-prefill", "--no-cpu", "--pipeline", "fx"],
                        capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r2.returncode == 0, r2.stderr[-2000:]
    d2 = _line(r2.stdout)
    assert d2["roofline"]["launches_per_step"] > 1 and len(d2["roofline"]["per_launch"]) == 5 and d2["roofline"]["fit"]["const_us"] > 0


def test_bench_two_ranks_control_flow(dev):
    """`--gpus 2` as launched by the driver (torch.distributed.run, one rank per process).  This box has one GPU, so the test hook
    EXL3_HIP_TP_BACKEND=gloo lets both ranks share it (RCCL refuses that) -- the collectives are gloo's, everything else is the
    multi-rank path of bench.py: per-rank shards, barriers, max-over-ranks timing, rank 0 printing the one line."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, EXL3_HIP_TP_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--layers", "2", "--steps", "2", "--warmup", "1",
                        "--no-graph", "--prefill-tokens", "512"], capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = _line(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["parallelism"] == "tp2" and d["cpu_baseline"] is None
    # the decode all-reduces ran through the IPC push (two processes mapping each other's buffers on the shared GPU) and were timed
    ar = d["allreduce"]
    assert ar["path"].startswith("ipc") and ar["ipc_us"] > 0 and ar["library_us"] > 0 and ar["per_step"] == 4
    assert ar["ipc_enabled"] and not ar["ipc_fell_back"] and len(ar["bytes_per_token_per_rank"]) == 2
    # the prefill leg runs under TP too (task 3): whole-job tok/s, the chunk's all-reduces priced separately
    pf = d["prefill"]
    assert pf["n_gpus"] == 2 and pf["value"] > 0 and pf["allreduce"]["per_chunk"] == 4 and pf["allreduce"]["prefill_ms"] > 0
    assert pf["allreduce"]["message_bytes"] == 512 * 4096 * 4


@pytest.mark.parametrize("name,K,cb,bsz,pipeline", [("llama-3.1-8b", 4, 2, 1, "fx"), ("llama-3.1-8b", 4, 0, 1, "fx"), ("llama-3.1-8b", 4, 2, 16, "fx"),
                                                     ("llama-3.2-1b", 4, 2, 1, "fx"), ("llama-3.1-8b", 4, 2, 1, "glue"), ("llama-3.1-70b", 3, 2, 1, "fx"),
                                                     ("llama-3.1-8b", 4, 2, 1, "persistent"), ("llama-3.2-1b", 4, 2, 1, "persistent")])
def test_bench_pinned_logits_gate(dev, name, K, cb, bsz, pipeline):
    """bench.py's correctness gate: the pipeline it times, over the pin model of the benchmark's shape, reproduces the oracle's committed logits
    (tests/golden/bench_pins.json, re-derived from the oracle by tests/test_bench_pins.py) within 3e-2 * RMS, and its hipGraph replay equals the eager
    step bit for bit.  The function asserts; here also that the pin exists (a missing pin would silently skip the gate)."""
    sys.path.insert(0, ROOT)
    import bench
    r = bench.pinned_logits_check(name, K, cb, bsz, dev, pipeline)
    assert r["pinned"] and r["ok"] and r["graph_replay_bit_equal"] and r["rel_err_vs_oracle"] < 3e-2


@pytest.mark.parametrize("name,ctx,pipeline", [("llama-3.1-8b", 1000, "persistent"), ("llama-3.1-8b", 16000, "persistent"), ("llama-3.2-1b", 1000, "persistent"),
                                               ("llama-3.1-8b", 1000, "fx"), ("llama-3.1-8b", 16000, "fx"), ("llama-3.2-1b", 1000, "fx"),
                                               # the step through the reference's runner objects (ext.BC_Attention / BC_GatedMLP / BC_LinearEXL3: llama_path.decode_step_bc_runners)
                                               ("llama-3.1-8b", 1000, "bc_runners"), ("llama-3.2-1b", 1000, "bc_runners")])
def test_bench_pinned_logits_gate_with_attention(dev, name, ctx, pipeline):
    """The gate of bench.py's ..._with_attention_ctx* lines (round 6: they carried `edge_timeout: false` only): the timed step with the decode attention over a host-seeded
    pre-filled 4-bit cache reproduces the oracle's committed logits; at 16 000 tokens the persistent step's context splits take several 128-token steps."""
    sys.path.insert(0, ROOT)
    import bench
    r = bench.pinned_logits_check(name, 4, 2, 1, dev, pipeline, ctx=ctx)
    assert r["pinned"] and r["ok"] and r["graph_replay_bit_equal"] and r["rel_err_vs_oracle"] < 3e-2
    if pipeline == "persistent" and ctx == 16000:
        assert r["attention_splits_tokens_steps"][2] > 1


def test_bench_pinned_logits_gate_catches_a_wrong_pipeline(dev, monkeypatch):
    """The gate fails when the step is wrong: with the norm weights of the pin model perturbed after construction the assertion fires."""
    sys.path.insert(0, ROOT)
    import bench
    from exllamav3_amd.llama_path import SyntheticEXL3Llama
    orig = SyntheticEXL3Llama.pin_model.__func__

    def broken(cls, *a, **kw):
        m = orig(cls, *a, **kw)
        m.layers[0]["norm2"].mul_(1.5)
        return m
    monkeypatch.setattr(SyntheticEXL3Llama, "pin_model", classmethod(broken))
    with pytest.raises(AssertionError):
        bench.pinned_logits_check("llama-3.2-1b", 4, 2, 1, dev, "fx")
