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
