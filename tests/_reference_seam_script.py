"""Run by tests/test_reference_seam.py / test_gpu_reference_seam.py in a subprocess whose PYTHONPATH = [stub dir, repo, /root/reference]:
drives the REFERENCE's own Python classes (LinearEXL3, RMSNorm, RoPE) whose `ext` is this build's module.
argv[1] = device ("cpu": every op must stop at this build's "tensor must be on a GPU device" check, i.e. the reference's call reached the
mirror with a compatible signature; "cuda:0": results are compared with the oracle)."""
import sys, types
import numpy as np
import torch
dev = sys.argv[1]
on_gpu = dev.startswith("cuda")
from oracle import exl3_oracle as o
import exllamav3_ext
assert exllamav3_ext.__implementation__.__name__ == "exllamav3_amd.ext"
from exllamav3.modules.quant.exl3 import LinearEXL3
from exllamav3.modules.rmsnorm import RMSNorm
from exllamav3.util.rope import RoPE, RopeSettings, RopeStyle


def expect_device_error(fn, what):
    try:
        fn()
    except RuntimeError as e:
        assert "GPU device" in str(e), f"{what}: unexpected error {e!r}"
        return
    raise SystemExit(f"{what}: expected this build's device check to fire on CPU tensors")


T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
rel = lambda got, ref: float(np.abs(got.astype(np.float32) - ref.astype(np.float32)).max() / (np.sqrt((ref.astype(np.float32) ** 2).mean()) + 1e-9))

# ---- LinearEXL3 (modules/quant/exl3.py:16-218): kernel route at 1 / 16 rows, reconstruct + hgemm at 145, fused reconstruct at 1100
k, n, K = 512, 384, 4
tr, su, sv = o.synth_linear(k, n, K, seed=3, realistic=True)
lin = LinearEXL3(None, k, n, suh=T(su), svh=T(sv), trellis=T(tr), mul1=torch.zeros(1, dtype=torch.int32), key="seam.test")
for rows in (1, 16, 145, 1100):
    x = np.random.default_rng(rows).standard_normal((rows, k)).astype(np.float16)
    if on_gpu:
        y = lin.forward(T(x), {}).float().cpu().numpy()
        ref = o.linear_forward(x, tr, su, sv, K, 2)
        assert y.shape == (rows, n) and rel(y, ref) < (1e-2 if rows <= 144 else 2e-2), (rows, rel(y, ref))
    else:
        expect_device_error(lambda: lin.forward(T(x), {}), f"LinearEXL3.forward rows={rows}")
if on_gpu:
    w = lin.get_weight_tensor().float().cpu().numpy()
    assert rel(w, o.weight_tensor(tr, su, sv, K, 2)) < 2e-2

# ---- RMSNorm.forward (modules/rmsnorm.py:98-160) called unbound on a namespace with the attributes it reads
xs = (np.random.default_rng(1).standard_normal((3, 512)) * 2).astype(np.float16)
wn = (1 + 0.1 * np.random.default_rng(2).standard_normal(512)).astype(np.float16)
ns = types.SimpleNamespace(weight=T(wn), rms_norm_eps=1e-5, constant_bias=0.0, constant_scale=1.0, span_heads=False, out_dtype=torch.half, key="n")
if on_gpu:
    y = RMSNorm.forward(ns, T(xs), {}).float().cpu().numpy()
    assert np.allclose(y, o.rms_norm(xs, wn, 1e-5).astype(np.float32), rtol=1e-3, atol=1e-3)
    r = (np.random.default_rng(4).standard_normal((3, 512))).astype(np.float16)
    tr_ = T(r)
    y = RMSNorm.forward(ns, T(xs), {}, residual_in=tr_).float().cpu().numpy()
    yr, rr = o.rms_norm(xs, wn, 1e-5, residual_in=r)
    assert np.allclose(y, yr.astype(np.float32), rtol=2e-3, atol=2e-3) and np.allclose(tr_.float().cpu().numpy(), rr.astype(np.float32), rtol=1e-3, atol=1e-3)
else:
    expect_device_error(lambda: RMSNorm.forward(ns, T(xs), {}), "RMSNorm.forward")

# ---- RoPE.apply (util/rope.py:469-530) with llama3 scaling
rs = RopeSettings(head_dim=128, rope_theta=500000.0, rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0,
                  "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}, max_position_embeddings=131072, rope_style=RopeStyle.NEOX)
rope = RoPE(dev, rs)
q = np.random.default_rng(5).standard_normal((2, 5, 4, 128)).astype(np.float16)
kk = np.random.default_rng(6).standard_normal((2, 5, 2, 128)).astype(np.float16)
if on_gpu:
    qo, ko = rope.apply(T(q), T(kk), position=37)
    rq, rk = o.rope(q, kk, rope.inv_freq.float().cpu().numpy(), position=37, rope_mode=o.ROPE_NEOX)
    assert np.allclose(qo.float().cpu().numpy(), rq.astype(np.float32), atol=3e-3, rtol=3e-3)
    assert np.allclose(ko.float().cpu().numpy(), rk.astype(np.float32), atol=3e-3, rtol=3e-3)
else:
    expect_device_error(lambda: rope.apply(T(q), T(kk), position=37), "RoPE.apply")
print("REFERENCE_CALLS_OK", dev)
