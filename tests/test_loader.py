"""EXL3 checkpoint reader (exllamav3_amd/loader.py) on synthetic safetensors files with the reference's tensor naming
(modules/linear.py:389-425): suh/svh and legacy packed su/sv, codebook markers, bias; CPU only (no kernels are launched)."""
import numpy as np
import pytest
import torch
from safetensors.torch import save_file
from oracle import exl3_oracle as o


def _write(tmp_path):
    k, n, K = 256, 384, 3
    tr, suh, svh = o.synth_linear(k, n, K, seed=5)                       # +-1 scales
    t = {
        "model.layers.0.mlp.up_proj.trellis": torch.from_numpy(tr), "model.layers.0.mlp.up_proj.suh": torch.from_numpy(suh),
        "model.layers.0.mlp.up_proj.svh": torch.from_numpy(svh), "model.layers.0.mlp.up_proj.mul1": torch.zeros(1, dtype=torch.int32),
        "model.layers.0.mlp.up_proj.bias": torch.arange(n, dtype=torch.float16),
    }
    # legacy group: packed sign bitfields instead of suh / svh
    su = torch.from_numpy(o.pack_signs(suh))
    sv = torch.from_numpy(o.pack_signs(svh))
    t2 = {"model.layers.0.mlp.down_proj.trellis": torch.from_numpy(tr), "model.layers.0.mlp.down_proj.su": su, "model.layers.0.mlp.down_proj.sv": sv,
          "model.norm.weight": torch.ones(k, dtype=torch.float16)}
    save_file(t, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file(t2, str(tmp_path / "model-00002-of-00002.safetensors"))
    return tr, suh, svh, K


def test_collection_and_linear_loading(tmp_path):
    from exllamav3_amd import loader
    tr, suh, svh, K = _write(tmp_path)
    stc = loader.SafetensorsCollection(str(tmp_path))
    assert stc.linear_keys() == ["model.layers.0.mlp.down_proj", "model.layers.0.mlp.up_proj"]
    assert stc.has_tensor("model.norm.weight") and not stc.has_tensor("nope")
    with pytest.raises(KeyError):
        stc.get_tensor("nope")
    up = loader.load_linear_exl3(stc, "model.layers.0.mlp.up_proj", "cpu")
    assert (up.in_features, up.out_features, up.K, up.mul1, up.mcg) == (256, 384, K, True, False)
    assert np.array_equal(up.trellis.numpy(), tr) and np.array_equal(up.suh.numpy(), suh) and np.array_equal(up.svh.numpy(), svh)
    assert up.bias is not None and float(up.bias[5]) == 5.0
    down = loader.load_linear_exl3(stc, "model.layers.0.mlp.down_proj", "cpu")
    assert (down.mul1, down.mcg, down.bias) == (False, False, None)
    assert np.array_equal(down.suh.numpy(), suh) and np.array_equal(down.svh.numpy(), svh)       # unpacked from the sign bitfields
    with pytest.raises(KeyError):
        loader.load_linear_exl3(stc, "model.norm", "cpu")


def test_unpack_bf_matches_oracle():
    from exllamav3_amd import loader
    rng = np.random.default_rng(0)
    bf = rng.integers(-32768, 32768, size=40, dtype=np.int16)
    assert np.array_equal(loader.unpack_bf(torch.from_numpy(bf)).numpy(), o.unpack_bf(bf))


def test_tp_slices_equal_full_load_then_shard(tmp_path):
    """A rank's shard read straight from the file (load_linear_exl3 tp_slice) == full load + LinearEXL3.tp_shard, for column and row
    shards, suh/svh and legacy sign-bit groups, with the bias kept only on the first row shard."""
    from exllamav3_amd import loader
    _write(tmp_path)
    stc = loader.SafetensorsCollection(str(tmp_path))
    for key in ("model.layers.0.mlp.up_proj", "model.layers.0.mlp.down_proj"):
        full = loader.load_linear_exl3(stc, key, "cpu")
        for (first, last, dim) in ((0, 128, "n"), (128, 384, "n"), (0, 128, "k"), (128, 256, "k")):
            a = loader.load_linear_exl3(stc, key, "cpu", tp_slice=(first, last, dim))
            b = full.tp_shard(first, last, dim)
            assert (a.in_features, a.out_features, a.K, a.mul1, a.mcg) == (b.in_features, b.out_features, b.K, b.mul1, b.mcg)
            assert torch.equal(a.trellis, b.trellis) and torch.equal(a.suh, b.suh) and torch.equal(a.svh, b.svh)
            assert (a.bias is None) == (b.bias is None) and (a.bias is None or torch.equal(a.bias, b.bias))
    with pytest.raises(ValueError):
        loader.load_linear_exl3(stc, "model.layers.0.mlp.up_proj", "cpu", tp_slice=(0, 100, "n"))
    with pytest.raises(ValueError):
        loader.load_linear_exl3(stc, "model.layers.0.mlp.up_proj", "cpu", tp_slice=(256, 512, "n"))


def test_tensor_storage_map(tmp_path):
    """quantization_config.json's tensor_storage rebuilt from the files (conversion/quant_config.py:14-55)."""
    from exllamav3_amd import loader
    _, _, _, K = _write(tmp_path)
    stc = loader.SafetensorsCollection(str(tmp_path))
    ts = stc.tensor_storage()
    assert sorted(ts) == ["model.layers.0.mlp.down_proj", "model.layers.0.mlp.up_proj"]
    up = ts["model.layers.0.mlp.up_proj"]
    assert up["quant_format"] == "exl3" and up["bits_per_weight"] == K
    st = up["stored_tensors"]
    assert st["model.layers.0.mlp.up_proj.trellis"] == {"shape": [16, 24, 16 * K], "dtype": "I16", "n_bytes": 16 * 24 * 16 * K * 2}
    assert st["model.layers.0.mlp.up_proj.suh"]["shape"] == [256]
    assert "mul1_multiplier" not in up                         # the marker's value is 0 in this file (presence selects the codebook)
    assert set(ts["model.layers.0.mlp.down_proj"]["stored_tensors"]) == {"model.layers.0.mlp.down_proj.trellis", "model.layers.0.mlp.down_proj.su", "model.layers.0.mlp.down_proj.sv"}


def test_from_checkpoint_tp_ranks_load_their_own_shards(tmp_path):
    """SyntheticEXL3Llama.from_checkpoint under TP = 2 (construction only, CPU): each rank's linears equal the whole model's tp_shard --
    q/k/v/gate/up/lm_head by columns, o/down by rows -- read as slices of the files."""
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    from exllamav3_amd.tp import split_points

    class FakeBackend:
        def __init__(self, rank, world): self.rank, self.world_size = rank, world

    shape = LlamaShape("tiny", 256, 512, 2, 4, 2, 128, 384)
    whole = SyntheticEXL3Llama(shape, K=3, cb=0, device="cpu", kv_bits=4, max_ctx=2048, head_K=4)
    whole.save_checkpoint(str(tmp_path))
    ipts, vpts = split_points(512, 2), split_points(384, 2)
    for rank in (0, 1):
        m = SyntheticEXL3Llama.from_checkpoint(str(tmp_path), device="cpu", backend=FakeBackend(rank, 2), max_ctx=2048)
        assert (m.hq, m.hkv, m.inter_local, m.vocab_local) == (2, 1, ipts[rank + 1] - ipts[rank], vpts[rank + 1] - vpts[rank])
        for Lw, Lr in zip(whole.layers, m.layers):
            exp = {"q": Lw["q"].tp_shard(rank * 256, (rank + 1) * 256, "n"), "k": Lw["k"].tp_shard(rank * 128, (rank + 1) * 128, "n"),
                   "v": Lw["v"].tp_shard(rank * 128, (rank + 1) * 128, "n"), "o": Lw["o"].tp_shard(rank * 256, (rank + 1) * 256, "k"),
                   "gate": Lw["gate"].tp_shard(ipts[rank], ipts[rank + 1], "n"), "up": Lw["up"].tp_shard(ipts[rank], ipts[rank + 1], "n"),
                   "down": Lw["down"].tp_shard(ipts[rank], ipts[rank + 1], "k")}
            for nm, e in exp.items():
                assert torch.equal(Lr[nm].trellis, e.trellis) and torch.equal(Lr[nm].suh, e.suh) and torch.equal(Lr[nm].svh, e.svh), nm
        e = whole.lm_head.tp_shard(vpts[rank], vpts[rank + 1], "n")
        assert torch.equal(m.lm_head.trellis, e.trellis) and torch.equal(m.lm_head.svh, e.svh)
