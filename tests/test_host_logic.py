"""Host logic of the LinearEXL3 mirror on CPU: which route a forward takes (modules/quant/exl3.py:114-218 thresholds), shard arithmetic, and the
argument checks of the Python binding -- with the kernel calls replaced by a recorder (no GPU, no compute)."""
import numpy as np
import pytest
import torch
from oracle import exl3_oracle as o


class Recorder:
    """Stands in for exllamav3_amd.ext inside linear.py: records (name, shapes) and leaves outputs untouched."""
    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        def f(*args, **kw):
            self.calls.append((name, [tuple(a.shape) if isinstance(a, torch.Tensor) else a for a in args]))
        return f


class FakeBC:
    def __init__(self, rec): self.rec = rec
    def run_alloc(self, x, n, fp32):
        self.rec.calls.append(("bc.run_alloc", [tuple(x.shape), n, fp32]))
        return torch.zeros(x.shape[:-1] + (n,), dtype=torch.float if fp32 else torch.half)


def _linear(k=256, n=384, K=3):
    from exllamav3_amd.linear import LinearEXL3
    tr, suh, svh = o.synth_linear(k, n, K, seed=1)
    return LinearEXL3(k, n, torch.from_numpy(tr), torch.from_numpy(suh), torch.from_numpy(svh))


def test_route_selection_by_row_count(monkeypatch):
    from exllamav3_amd import linear
    rec = Recorder()
    monkeypatch.setattr(linear, "ext", rec)
    lin = _linear()
    lin.bc = FakeBC(rec)
    names = lambda: [c[0] for c in rec.calls]
    # rows <= 144: the fused small-m kernel (exl3.py:133-137), flattened over leading dims
    for shape in ((1, 256), (144, 256), (2, 72, 256)):
        rec.calls.clear()
        y = lin.forward(torch.zeros(shape, dtype=torch.half))
        assert names() == ["bc.run_alloc"] and y.shape == shape[:-1] + (384,)
    # above: reconstruct + GEMM; this library's fused-W route from 145 rows
    rec.calls.clear()
    lin.forward(torch.zeros((145, 256), dtype=torch.half))
    assert names() == ["reconstruct_had_slice_t", "hgemm_nt"]
    # the reference's switch point restored: unfused route below 1024 rows (had -> reconstruct -> hgemm -> had + svh)
    monkeypatch.setattr(linear, "FUSED_RECONSTRUCT_MIN_ROWS", 1024)
    rec.calls.clear()
    lin.forward(torch.zeros((145, 256), dtype=torch.half))
    assert names()[:3] == ["had_r_128", "reconstruct", "hgemm"] and "had_r_128" in names()[3:]
    # params force either route regardless of rows (exl3.py:114-131)
    rec.calls.clear()
    lin.forward(torch.zeros((4, 256), dtype=torch.half), {"reconstruct": True})
    assert "reconstruct" in names() and "bc.run_alloc" not in names()
    rec.calls.clear()
    lin.forward(torch.zeros((2000, 256), dtype=torch.half), {"no_reconstruct": True})
    assert names() == ["bc.run_alloc"]
    # wide outputs are processed in column slices (exl3.py:199-211)
    monkeypatch.setattr(linear, "MAX_RECONSTRUCT_SLICE_N", 128)
    rec.calls.clear()
    lin.forward(torch.zeros((2000, 256), dtype=torch.half))
    assert sum(1 for n in names() if n.startswith("reconstruct")) == 3 and sum(1 for n in names() if n == "hgemm") == 3
    with pytest.raises(AssertionError):
        lin.forward(torch.zeros((4, 512), dtype=torch.half)[:, ::2])


def test_forward_multi_applicability(monkeypatch):
    from exllamav3_amd import linear
    rec = Recorder()
    monkeypatch.setattr(linear, "ext", rec)
    a, b = _linear(256, 384), _linear(256, 128, K=5)
    rec.calls.clear()                                                                    # the constructors built their BC_LinearEXL3 through ext
    outs = linear.LinearEXL3.forward_multi([a, b], torch.zeros((200, 256), dtype=torch.half))
    assert [tuple(t.shape) for t in outs] == [(200, 384), (200, 128)] and outs[1].stride(0) == 512 and outs[1].storage_offset() == 384
    assert [c[0] for c in rec.calls] == ["reconstruct_had_slice_t", "reconstruct_had_slice_t", "hgemm_nt"]
    assert linear.LinearEXL3.forward_multi([a, b], torch.zeros((100, 256), dtype=torch.half)) is None         # small-m territory
    assert linear.LinearEXL3.forward_multi([a, _linear(384, 128)], torch.zeros((200, 256), dtype=torch.half)) is None   # different k


def test_tp_shard_arithmetic():
    lin = _linear(256, 384)
    lin.bias = torch.arange(384, dtype=torch.float16)
    col = lin.tp_shard(128, 384, "n")
    assert (col.in_features, col.out_features) == (256, 256) and torch.equal(col.svh, lin.svh[128:]) and torch.equal(col.bias, lin.bias[128:])
    assert torch.equal(col.trellis, lin.trellis[:, 8:24]) and col.suh is lin.suh
    row0, row1 = lin.tp_shard(0, 128, "k"), lin.tp_shard(128, 256, "k")
    assert row0.bias is not None and row1.bias is None                                   # the all-reduce must add the bias once
    assert torch.equal(row1.suh, lin.suh[128:]) and row1.svh is lin.svh and torch.equal(row1.trellis, lin.trellis[8:])
    with pytest.raises(AssertionError):
        lin.tp_shard(0, 100, "n")


def test_binding_argument_checks_raise_before_any_launch():
    """The ctypes binding validates like TORCH_CHECK in the reference (exllamav3_ext/util.h:24-35): RuntimeError with a message, on CPU tensors
    too, because the checks run before the library is touched."""
    from exllamav3_amd import ext
    h = lambda *s: torch.zeros(s, dtype=torch.half)
    with pytest.raises(RuntimeError):
        ext.hgemm_nt(h(4, 8), h(3, 16), h(4, 3))                                            # k mismatch
    with pytest.raises(RuntimeError):
        ext.exl3_gemv_ex(h(2, 128), None, None, [torch.zeros((16, 8, 64), dtype=torch.int16)], None, None, None, 2, False, True, 0)   # width != k
    with pytest.raises(RuntimeError):
        ext.rope_strided(h(10, 100), None, torch.zeros(64), 0, None, None, 1.0, 1, 10)      # width not heads x 128
    with pytest.raises(RuntimeError):
        ext.quant_cache_paged_strided(h(4, 64), torch.zeros((1, 256, 8), dtype=torch.int32), h(1, 256, 2), h(4, 32), torch.zeros((1, 256, 8), dtype=torch.int32),
                                      h(1, 256, 2), torch.zeros(1, dtype=torch.int32), torch.zeros((1, 1), dtype=torch.int32), 256, 4)


def test_rope_inv_freq_matches_reference_llama3_and_default(golden):
    """config.json rope_scaling -> inv_freq: llama3 and default agree with the reference's RoPE object (fixtures from util/rope.py)."""
    import torch
    from exllamav3_amd.llama_path import rope_inv_freq
    rs = {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}
    got = rope_inv_freq(128, 500000.0, rs).numpy()
    np.testing.assert_allclose(got, golden["rope_neox_inv_freq"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(rope_inv_freq(64, 10000.0, None).numpy(), golden["rope_default64_inv_freq"], rtol=1e-6, atol=0)
    lin = rope_inv_freq(64, 10000.0, {"type": "linear", "factor": 4.0}).numpy()
    np.testing.assert_allclose(lin * 4.0, golden["rope_default64_inv_freq"], rtol=1e-6, atol=0)
    with pytest.raises(NotImplementedError):
        rope_inv_freq(64, 10000.0, {"rope_type": "yarn", "factor": 4.0})
