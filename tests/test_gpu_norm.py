"""GPU parity of RMSNorm / silu_mul / add against the oracle and the reference-Python fixtures."""
import numpy as np
import pytest
import torch
from oracle import exl3_oracle as o

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("dim", [8, 128, 2048, 4096, 12288])
@pytest.mark.parametrize("rows", [1, 3, 64])
@pytest.mark.parametrize("dt_in,dt_out", [(np.float16, np.float16), (np.float32, np.float16), (np.float16, np.float32)])
def test_rms_norm(dev, dim, rows, dt_in, dt_out):
    # tests/test_ext_norm_.py:11-40: dims, rows, dtype combos, tolerance 1e-3
    from exllamav3_amd import ext
    rng = np.random.default_rng(dim + rows)
    x = (rng.standard_normal((rows, dim)) * 1.5).astype(dt_in)
    w = (1 + 0.1 * rng.standard_normal(dim)).astype(np.float16)
    ref = o.rms_norm(x, w, 1e-6, out_dtype=dt_out).astype(np.float32)
    y = torch.empty((rows, dim), dtype=torch.float if dt_out == np.float32 else torch.half, device=dev)
    ext.rms_norm(_t(x, dev), _t(w, dev), y, 1e-6)
    assert np.allclose(y.float().cpu().numpy(), ref, rtol=1e-3, atol=1e-3)


def test_rms_norm_golden_and_modes(dev, golden):
    from exllamav3_amd import ext
    for tag in "abc":
        eps, cb, cs = golden[f"rms_{tag}_cfg"]
        x, w = golden[f"rms_{tag}_x"], golden[f"rms_{tag}_w"]
        y = torch.empty(x.shape, dtype=torch.half, device=dev)
        ext.rms_norm(_t(x, dev), _t(w, dev), y, float(eps), float(cb), float(cs))
        assert np.allclose(y.float().cpu().numpy(), golden[f"rms_{tag}_y"].astype(np.float32), rtol=2e-3, atol=2e-3)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((4, 4096)).astype(np.float16)
    w = (1 + 0.1 * rng.standard_normal(4096)).astype(np.float16)
    # RES_IN: r += x; y = norm(r), fp16 and fp32 residual streams
    for rdt in (np.float16, np.float32):
        r = rng.standard_normal((4, 4096)).astype(rdt)
        y_ref, r_ref = o.rms_norm(x, w, 1e-5, residual_in=r)
        rt = _t(r, dev)
        y = torch.empty((4, 4096), dtype=torch.half, device=dev)
        ext.rms_norm_res_in(_t(x, dev), _t(w, dev), y, rt, 1e-5)
        assert np.allclose(rt.float().cpu().numpy(), r_ref.astype(np.float32), rtol=1e-3, atol=1e-3)
        assert np.allclose(y.float().cpu().numpy(), y_ref.astype(np.float32), rtol=2e-3, atol=2e-3)
    # RES_POST: y += norm(x)
    base = rng.standard_normal((4, 4096)).astype(np.float16)
    y_ref = o.rms_norm(x, w, 1e-5, add_residual_to=base)
    y = _t(base, dev)
    ext.rms_norm(_t(x, dev), _t(w, dev), y, 1e-5, 0.0, 1.0, False, True)
    assert np.allclose(y.float().cpu().numpy(), y_ref.astype(np.float32), rtol=2e-3, atol=2e-3)
    # bf16 weights, unweighted
    wb = torch.from_numpy(w.astype(np.float32)).to(torch.bfloat16)
    y = torch.empty((4, 4096), dtype=torch.half, device=dev)
    ext.rms_norm(_t(x, dev), wb.to(dev), y, 1e-5)
    ref = o.rms_norm(x, wb.float().numpy(), 1e-5)
    assert np.allclose(y.float().cpu().numpy(), ref.astype(np.float32), rtol=2e-3, atol=2e-3)
    ext.rms_norm(_t(x, dev), None, y, 1e-5)
    assert np.allclose(y.float().cpu().numpy(), o.rms_norm(x, None, 1e-5).astype(np.float32), rtol=2e-3, atol=2e-3)


def test_silu_mul_add(dev):
    from exllamav3_amd import ext
    rng = np.random.default_rng(0)
    g = rng.standard_normal((3, 1024)).astype(np.float32) * 3
    u = rng.standard_normal((3, 1024)).astype(np.float32)
    ref = g / (1 + np.exp(-g)) * u
    for dt in (np.float16, np.float32):
        y = torch.empty((3, 1024), dtype=torch.half, device=dev)
        ext.silu_mul(_t(g.astype(dt), dev), _t(u.astype(dt), dev), y)
        r = (g.astype(dt).astype(np.float32)); r = r / (1 + np.exp(-r)) * u.astype(dt).astype(np.float32)
        assert np.allclose(y.float().cpu().numpy(), r, rtol=3e-3, atol=3e-3)
    a = _t(g, dev); ext.add(a, _t(u.astype(np.float16), dev))
    assert np.allclose(a.cpu().numpy(), g + u.astype(np.float16).astype(np.float32), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("act", ["silu", "gelu", "relu2", "relu", "silu_oai"])
@pytest.mark.parametrize("limit", [0.0, 1.5])
def test_act_mul_activations_and_limits(dev, act, limit):
    """exl3_act_mul (activation.cu gelu_mul / relu2_mul / silu_oai_mul and their act_limit clamps) against the oracle's restatement of the reference
    kernels, fp16 and fp32 inputs.  The device computes in fp32 and rounds once; the reference's fp16 kernel rounds after every step: 3e-3 covers both."""
    from exllamav3_amd import ext
    rng = np.random.default_rng(len(act) + int(limit * 10))
    g = (rng.standard_normal((5, 512)) * 3).astype(np.float32); u = (rng.standard_normal((5, 512)) * 2).astype(np.float32)
    code = {"silu": ext.ACT_SILU, "gelu": ext.ACT_GELU, "relu2": ext.ACT_RELU2, "relu": ext.ACT_RELU, "silu_oai": ext.ACT_SILU_OAI}[act]
    for dt in (np.float16, np.float32):
        y = torch.full((5, 512), float("nan"), dtype=torch.half, device=dev)
        ext.act_mul(_t(g.astype(dt), dev), _t(u.astype(dt), dev), y, code, limit)
        ref = o.act_mul(g.astype(dt), u.astype(dt), act, limit).astype(np.float32)
        got = y.float().cpu().numpy()
        assert np.isfinite(got).all()
        assert np.allclose(got, ref, rtol=3e-3, atol=3e-3), np.abs(got - ref).max()
    with pytest.raises(RuntimeError):
        ext.act_mul(_t(g, dev), _t(u, dev), torch.empty((5, 512), dtype=torch.half, device=dev), 7, 0.0)


@pytest.mark.parametrize("dim", [2048, 4096, 8192])
@pytest.mark.parametrize("rows", [64, 257, 4096])
def test_rms_norm_many_rows_fast_path(dev, dim, rows):
    """>= 64 fp16 rows of 2048 / 4096 / 8192: the one-wave-per-row kernel (prefill), plain and RES_IN, fp16 and bf16 weights, with the
    constant bias / scale arguments; same tolerance as the block-per-row kernel."""
    from exllamav3_amd import ext
    rng = np.random.default_rng(dim + rows)
    x = (rng.standard_normal((rows, dim)) * 1.5).astype(np.float16)
    w = (1 + 0.1 * rng.standard_normal(dim)).astype(np.float16)
    y = torch.empty((rows, dim), dtype=torch.half, device=dev)
    ext.rms_norm(_t(x, dev), _t(w, dev), y, 1e-6)
    assert np.allclose(y.float().cpu().numpy(), o.rms_norm(x, w, 1e-6).astype(np.float32), rtol=1e-3, atol=1e-3)
    ext.rms_norm(_t(x, dev), _t(w, dev), y, 1e-5, 1.0, 0.5)
    assert np.allclose(y.float().cpu().numpy(), o.rms_norm(x, w, 1e-5, constant_bias=1.0, constant_scale=0.5).astype(np.float32), rtol=2e-3, atol=2e-3)
    wb = torch.from_numpy(w.astype(np.float32)).to(torch.bfloat16)
    ext.rms_norm(_t(x, dev), wb.to(dev), y, 1e-5)
    assert np.allclose(y.float().cpu().numpy(), o.rms_norm(x, wb.float().numpy(), 1e-5).astype(np.float32), rtol=2e-3, atol=2e-3)
    r = rng.standard_normal((rows, dim)).astype(np.float16)
    y_ref, r_ref = o.rms_norm(x, w, 1e-5, residual_in=r)
    rt = _t(r, dev)
    ext.rms_norm_res_in(_t(x, dev), _t(w, dev), y, rt, 1e-5)
    assert np.allclose(rt.float().cpu().numpy(), r_ref.astype(np.float32), rtol=1e-3, atol=1e-3)
    assert np.allclose(y.float().cpu().numpy(), y_ref.astype(np.float32), rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("shape", [(1, 1, 4, 128), (2, 5, 32, 64), (3, 17, 6, 96)])
def test_attention_gate_ops_vs_oracle(dev, shape):
    """mul_sigmoid_ / mul_sigmoid_broadcast_ / mul_softplus_broadcast_ / deinterleave_qg (activation.cu:526-785) against the oracle, which runs
    against the reference's own kernels on the host (tests/test_oracle_pins.py): the fp16 sigmoid to 2 fp16 ulps of the product (the device exp is not the
    correctly rounded one), the fp32 softplus gate to 1 ulp, the deinterleave bit for bit; gates from -12 to 12."""
    from exllamav3_amd import ext
    from oracle import exl3_oracle as o
    b, s_, h, d = shape
    rng = np.random.default_rng(h * d)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x = (rng.standard_normal(shape) * 2).astype(np.float16)
    y_full = rng.uniform(-12, 12, size=shape).astype(np.float16)
    y_head = rng.uniform(-12, 12, size=shape[:3]).astype(np.float16)
    ulps = lambda got, ref: float((np.abs(got.astype(np.float32) - ref.astype(np.float32)) / (np.maximum(np.abs(ref.astype(np.float32)), 2.0 ** -14) * 2.0 ** -10)).max())
    t = T(x); ext.mul_sigmoid_(t, T(y_full))
    assert ulps(t.cpu().numpy(), o.mul_sigmoid(x, y_full)) <= 2
    t = T(x); ext.mul_sigmoid_broadcast_(t, T(y_head))
    assert ulps(t.cpu().numpy(), o.mul_sigmoid(x, y_head)) <= 2
    t = T(x); ext.mul_softplus_broadcast_(t, T(y_head))
    assert ulps(t.cpu().numpy(), o.mul_softplus_broadcast(x, y_head)) <= 1
    if d % 8 == 0:
        qg = rng.standard_normal((b, s_, h * 2 * d)).astype(np.float16)
        q = torch.zeros((b, s_, h * d), dtype=torch.half, device=dev); g = torch.zeros_like(q)
        ext.deinterleave_qg(T(qg), q, g, d)
        rq, rg = o.deinterleave_qg(qg, d)
        assert np.array_equal(q.cpu().numpy(), rq) and np.array_equal(g.cpu().numpy(), rg)
    with pytest.raises(RuntimeError):
        ext.mul_sigmoid_broadcast_(T(x), T(y_full))                        # y must be [B, S, H]


def test_shared_expert_gate_ops_and_fp16_cache_append_vs_oracle(dev):
    """add_sigmoid_gate / add_sigmoid_gate_proj (activation.cu:480-524, 662-714: the shared-expert merge of the sparse-MoE block), relu_mul, and
    paged_kv_cache_update (generator/cache.cu:186-240: fp16 cache append at cache_seqlens[b] + t through the block table) against the oracle / numpy."""
    from exllamav3_amd import ext
    from oracle import exl3_oracle as o
    rng = np.random.default_rng(3)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x = rng.standard_normal((5, 384)).astype(np.float32); z = rng.standard_normal((5, 384)).astype(np.float32)
    y1 = (rng.standard_normal((5, 1)) * 3).astype(np.float32)
    tz = T(z); ext.add_sigmoid_gate(T(x), T(y1), tz)
    assert np.allclose(tz.cpu().numpy(), o.add_sigmoid_gate(x, y1, z), rtol=1e-5, atol=1e-5)
    yh = rng.standard_normal((5, 384)).astype(np.float16); w = (rng.standard_normal((384, 1)) * 0.1).astype(np.float16)
    yh[4] = (-40 * np.sign(w[:, 0].astype(np.float32))).astype(np.float16)              # a gate far below 1e-8: the row stays untouched
    tz = T(z); ext.add_sigmoid_gate_proj(T(x), T(yh), tz, T(w))
    ref = o.add_sigmoid_gate_proj(x, yh, z, w)
    assert np.allclose(tz.cpu().numpy(), ref, rtol=1e-4, atol=1e-4) and np.array_equal(tz.cpu().numpy()[4], z[4])
    g16 = rng.standard_normal((3, 256)).astype(np.float16); u16 = rng.standard_normal((3, 256)).astype(np.float16)
    t = torch.zeros((3, 256), dtype=torch.half, device=dev); ext.relu_mul(T(g16), T(u16), t)
    assert np.array_equal(t.cpu().numpy(), o.act_mul(g16, u16, "relu", 0.0))
    # fp16 cache append: 3 sequences, 5 new tokens each, scattered pages, one sequence crossing a page edge
    B, S, H, D, pps = 3, 5, 2, 64, 2
    k = rng.standard_normal((B, S, H, D)).astype(np.float16); v = rng.standard_normal((B, S, H, D)).astype(np.float16)
    bt = rng.permutation(B * pps).reshape(B, pps).astype(np.int32)
    lens = np.array([0, 254, 300], np.int32)
    kc = torch.zeros((B * pps, 256, H, D), dtype=torch.half, device=dev); vc = torch.zeros_like(kc)
    ext.paged_kv_cache_update(T(k), T(v), kc, vc, T(bt), T(lens))
    kcn, vcn = kc.cpu().numpy(), vc.cpu().numpy()
    written = 0
    for b in range(B):
        for t_ in range(S):
            pos = lens[b] + t_
            assert np.array_equal(kcn[bt[b, pos // 256], pos % 256], k[b, t_]) and np.array_equal(vcn[bt[b, pos // 256], pos % 256], v[b, t_])
            written += 1
    assert int((np.abs(kcn).sum((2, 3)) > 0).sum()) == written
