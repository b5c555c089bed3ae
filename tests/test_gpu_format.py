"""GPU parity (through the C-ABI) of the integer / byte / index ops: BIT-EXACT against the oracle."""
import numpy as np
import pytest
import torch
from oracle import exl3_oracle as o

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("K", range(1, 9))
def test_unpack_pack_bit_exact(dev, K):
    from exllamav3_amd import ext
    rng = np.random.default_rng(100 + K)
    tr = rng.integers(0, 65536, size=(5, 9, 16 * K), dtype=np.uint16).view(np.int16)
    ref = o.unpack_trellis(tr, K)
    packed = _t(tr, dev)
    unpacked = torch.empty((5, 9, 256), dtype=torch.int16, device=dev)
    ext.unpack_trellis(unpacked, packed, K)
    assert np.array_equal(unpacked.cpu().numpy().view(np.uint16), ref)
    repacked = torch.zeros_like(packed)
    ext.pack_trellis(repacked, unpacked, K)
    assert np.array_equal(repacked.cpu().numpy(), tr)
    # pack from full 16-bit states of a constructive tail-biting sequence (tests/test_quant_fn.py:94-128)
    st = o.random_tailbiting_states(rng, (3, 4), K)
    p2 = torch.zeros((3, 4, 16 * K), dtype=torch.int16, device=dev)
    ext.pack_trellis(p2, _t(st.view(np.int16), dev), K)
    assert np.array_equal(p2.cpu().numpy(), o.pack_trellis(st, K))
    u2 = torch.empty((3, 4, 256), dtype=torch.int16, device=dev)
    ext.unpack_trellis(u2, p2, K)
    assert np.array_equal(u2.cpu().numpy().view(np.uint16), st)


@pytest.mark.parametrize("cb", [0, 1, 2])
def test_decode_all_states_bit_exact(dev, cb):
    from exllamav3_amd import ext
    allst = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    ref = o.decode(allst, cb)
    idx = _t(allst.view(np.int16), dev)
    out = torch.empty(65536, dtype=torch.half, device=dev)
    ext.decode(idx, out, cb == 1, cb == 2)
    assert np.array_equal(out.cpu().numpy().view(np.uint16), ref.view(np.uint16))
    out32 = torch.empty(65536, dtype=torch.float, device=dev)
    ext.decode(idx, out32, cb == 1, cb == 2)
    assert np.array_equal(out32.cpu().numpy(), ref.astype(np.float32))


@pytest.mark.parametrize("K", range(1, 9))
@pytest.mark.parametrize("cb", [0, 1, 2])
def test_reconstruct_bit_exact(dev, K, cb):
    from exllamav3_amd import ext
    k, n = 256, 384
    tr, _, _ = o.synth_linear(k, n, K)
    ref = o.reconstruct(tr, K, cb)
    w = torch.empty((k, n), dtype=torch.half, device=dev)
    ext.reconstruct(w, _t(tr, dev), K, cb == 1, cb == 2)
    assert np.array_equal(w.cpu().numpy().view(np.uint16), ref.view(np.uint16))
    ws = torch.empty((k, 128), dtype=torch.half, device=dev)
    ext.reconstruct_slice(ws, _t(tr, dev), K, cb == 1, cb == 2, 256)
    assert np.array_equal(ws.cpu().numpy().view(np.uint16), ref[:, 256:384].view(np.uint16))


def test_reconstruct_reference_test_shapes(dev):
    """tests/test_reconstruct_had.py:35-42 shapes/seeds."""
    from exllamav3_amd import ext
    for (k, n, K, cb) in [(256, 128, 3, 0), (512, 384, 2, 0), (1024, 512, 5, 0), (384, 256, 4, 1), (256, 512, 3, 2)]:
        tr, _, _ = o.synth_linear(k, n, K)
        w = torch.empty((k, n), dtype=torch.half, device=dev)
        ext.reconstruct(w, _t(tr, dev), K, cb == 1, cb == 2)
        assert np.array_equal(w.cpu().numpy().view(np.uint16), o.reconstruct(tr, K, cb).view(np.uint16))


def test_pack_signs(dev, golden):
    from exllamav3_amd import ext
    signs = _t(golden["bf_out"], dev)
    packed = torch.zeros(signs.numel() // 16, dtype=torch.int16, device=dev)
    ext.pack_signs(packed, signs)
    assert np.array_equal(packed.cpu().numpy(), golden["bf_in"])


def test_error_behaviour(dev):
    from exllamav3_amd import ext
    w = torch.empty((256, 100), dtype=torch.half, device=dev)
    tr = torch.zeros((16, 8, 64), dtype=torch.int16, device=dev)
    with pytest.raises(RuntimeError):
        ext.reconstruct_slice(w, tr, 4, False, False, 0)          # N not divisible by 128
    with pytest.raises(RuntimeError):
        ext.had_r_128(torch.empty((2, 100), dtype=torch.half, device=dev), torch.empty((2, 100), dtype=torch.half, device=dev), None, None)


@pytest.mark.parametrize("fp32", [False, True])
def test_had_r_128(dev, fp32):
    from exllamav3_amd import ext
    rng = np.random.default_rng(7)
    x = rng.standard_normal((13, 512)).astype(np.float32 if fp32 else np.float16)
    sc = (np.sign(rng.standard_normal(512)) * (0.5 + rng.random(512))).astype(np.float16)
    for pre, post in [(None, None), (sc, None), (None, sc)]:
        ref = o.had_r_128(x, pre_scale=pre, post_scale=post)
        xi = _t(x, dev)
        y = torch.empty_like(xi)
        ext.had_r_128(xi, y, None if pre is None else _t(pre, dev), None if post is None else _t(post, dev), 1.0)
        got = y.cpu().numpy()
        if fp32:
            assert np.allclose(got, ref, rtol=1e-5, atol=1e-5)
        else:
            # fp32 butterflies in the same order + identical rounding points -> bit-exact
            assert np.array_equal(got.view(np.uint16), ref.view(np.uint16))
    # in place
    xi = _t(x, dev)
    ext.had_r_128(xi, xi, None, None, 1.0)
    assert np.allclose(xi.float().cpu().numpy(), o.had_r_128(x).astype(np.float32), rtol=1e-3, atol=1e-3)
