"""Run by tests/test_reference_seam.py in a subprocess whose PYTHONPATH = [stub dir, repo, /root/reference] (CPU, needs the reference):
pins the on-disk format reader / writer of this build (exllamav3_amd/loader.py, SyntheticEXL3Llama.save_checkpoint) against the REFERENCE's own
reader and tensor-naming code (task 4a):

  A. a checkpoint written by this build's save_checkpoint is read by the reference's SafetensorsCollection (loader/safetensors.py:185) +
     Linear.load_exl3 (modules/linear.py:389-425): every tensor of every quantized Linear arrives bit for bit, codebook markers included;
  B. a checkpoint assembled by the reference's own naming code (LinearEXL3.get_tensors, modules/quant/exl3.py:96-112) from reference-constructed
     linears -- 3INST / mcg / mul1 markers, a bias, the legacy packed su / sv sign bitfields -- is read by this build's loader: bit for bit;
  C. legacy files that STORE .su / .sv (old checkpoints): both readers unpack them to the same +-1 vectors;
  D. tensor-parallel shards read by this build (tp_slice) equal the slices of the tensors the reference loaded whole.
"""
import json
import os
import sys
import tempfile
import types
import numpy as np
import torch
from safetensors.torch import save_file

import exllamav3_ext
assert exllamav3_ext.__implementation__.__name__ == "exllamav3_amd.ext"
from exllamav3.loader.safetensors import SafetensorsCollection as RefSTC
from exllamav3.modules.linear import Linear as RefLinear
from exllamav3.modules.quant.exl3 import LinearEXL3 as RefLinearEXL3
from exllamav3_amd import loader as L
from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
from oracle import exl3_oracle as o

eq = lambda a, b: a.dtype == b.dtype and a.shape == b.shape and bool(torch.equal(a.cpu(), b.cpu()))


def ref_load(stc, key, k, n):
    """The reference's own load path for one quantized Linear: Linear(config with .stc).load_exl3(key)."""
    cfg = types.SimpleNamespace(stc=stc)
    lin = RefLinear.__new__(RefLinear)
    # the attributes load_exl3 / is_exl3_storage read (modules/linear.py:380-425); the Module / Linear constructors need a full model Config
    lin.config, lin.key, lin.alt_key, lin.device = cfg, key, None, torch.device("cpu")
    lin.in_features, lin.out_features, lin.out_dtype = k, n, None
    assert lin.load_exl3(key) is True, key
    return lin.inner


with tempfile.TemporaryDirectory() as td:
    # ---- A: this build writes, the reference reads
    d_a = os.path.join(td, "a"); os.makedirs(d_a)
    shape = LlamaShape("tiny", 256, 384, 2, 4, 2, 64, 512)
    checked = 0
    for cb in (0, 1, 2):
        model = SyntheticEXL3Llama(shape, K=3 + cb, cb=cb, device="cpu")
        dd = os.path.join(d_a, f"cb{cb}"); model.save_checkpoint(dd)
        rstc = RefSTC(dd, load_method="python")
        mine = L.SafetensorsCollection(dd)
        keys = mine.linear_keys()
        assert len(keys) == 2 * 7 + 1
        for key in keys:
            a = L.load_linear_exl3(mine, key, "cpu")
            r = ref_load(rstc, key, a.in_features, a.out_features)
            assert eq(r.trellis, a.trellis) and eq(r.suh, a.suh) and eq(r.svh, a.svh), key
            assert r.mcg == a.mcg == (cb == 1) and r.mul1 == a.mul1 == (cb == 2) and r.K == a.K == 3 + cb, key
            assert (r.bias is None) == (a.bias is None)
            checked += 1
        # the reference's tensor inventory of a group == this build's (names, shapes, byte counts)
        try:
            rl = rstc.list_tensors("model.layers.0.self_attn.q_proj", only_serializable=True)
            ml = mine.list_tensors("model.layers.0.self_attn.q_proj")
            assert set(rl) == set(ml) and all(rl[k_]["shape"] == ml[k_]["shape"] and rl[k_]["n_bytes"] == ml[k_]["n_bytes"] for k_ in rl)
        except ImportError:
            pass                                           # marisa_trie (the reference's prefix index) is not in this image
    print("A_OK", checked)

    # ---- B: the reference names the tensors, this build reads
    d_b = os.path.join(td, "b"); os.makedirs(d_b)
    rng = np.random.default_rng(0)
    tensors, made = {}, {}
    def bits_of(v):                                        # +-1 vector -> the packed int16 sign bitfield of old checkpoints (bit i of word j = channel 16 j + i, set = negative)
        b = (np.asarray(v) < 0).astype(np.uint16).reshape(-1, 16)
        return torch.from_numpy((b << np.arange(16, dtype=np.uint16)).sum(axis=1).astype(np.uint16).view(np.int16))
    for i, (k, n, K, cb, bias, legacy) in enumerate([(256, 384, 4, 0, False, False), (384, 128, 2, 1, True, False), (128, 256, 5, 2, False, False),
                                                     (256, 256, 3, 2, True, True)]):
        tr, su, sv = o.synth_linear(k, n, K, seed=i, realistic=not legacy)
        kw = {}
        if legacy: kw.update(su=bits_of(su), sv=bits_of(sv))
        else: kw.update(suh=torch.from_numpy(su), svh=torch.from_numpy(sv))
        if cb == 1: kw["mcg"] = torch.tensor([0xCBAC1FED - (1 << 32)], dtype=torch.int32)
        if cb == 2: kw["mul1"] = torch.tensor([0x83DCD12D - (1 << 32)], dtype=torch.int32)
        if bias: kw["bias"] = torch.from_numpy((rng.standard_normal(n) * 0.1).astype(np.float16))
        key = f"model.layers.{i}.mlp.down_proj"
        rl = RefLinearEXL3(None, k, n, trellis=torch.from_numpy(tr), key=key, **kw)
        tensors.update(rl.get_tensors(key))                 # the reference's own naming code
        made[key] = (rl, cb, K, su, sv, legacy)
    save_file({k_: v.contiguous() for k_, v in tensors.items()}, os.path.join(d_b, "model.safetensors"))
    mine = L.SafetensorsCollection(d_b)
    assert mine.linear_keys() == sorted(made)
    for key, (rl, cb, K, su, sv, legacy) in made.items():
        a = L.load_linear_exl3(mine, key, "cpu")
        assert eq(a.trellis, rl.trellis) and eq(a.suh, rl.suh) and eq(a.svh, rl.svh), key
        assert a.mcg == rl.mcg == (cb == 1) and a.mul1 == rl.mul1 == (cb == 2) and a.K == rl.K == K
        assert (a.bias is None) == (rl.bias is None) and (a.bias is None or eq(a.bias, rl.bias))
        if legacy:
            assert np.array_equal(a.suh.numpy(), su.astype(np.float16)) and np.array_equal(a.svh.numpy(), sv.astype(np.float16))
    ts = mine.tensor_storage()
    assert ts["model.layers.2.mlp.down_proj"]["bits_per_weight"] == 5 and ts["model.layers.2.mlp.down_proj"]["mul1_multiplier"] == 0x83DCD12D
    assert ts["model.layers.1.mlp.down_proj"]["mcg_multiplier"] == 0xCBAC1FED and "mul1_multiplier" not in ts["model.layers.0.mlp.down_proj"]
    print("B_OK", len(made))

    # ---- C: a legacy file that stores .su / .sv: both readers
    d_c = os.path.join(td, "c"); os.makedirs(d_c)
    k, n, K = 256, 384, 4
    tr, su, sv = o.synth_linear(k, n, K, seed=77)
    key = "model.layers.0.self_attn.o_proj"
    save_file({key + ".trellis": torch.from_numpy(tr), key + ".su": bits_of(su), key + ".sv": bits_of(sv)}, os.path.join(d_c, "model.safetensors"))
    a = L.load_linear_exl3(L.SafetensorsCollection(d_c), key, "cpu")
    r = ref_load(RefSTC(d_c, load_method="python"), key, k, n)
    assert eq(a.suh, r.suh) and eq(a.svh, r.svh) and eq(a.trellis, r.trellis)
    assert np.array_equal(a.suh.numpy(), su.astype(np.float16)) and np.array_equal(a.svh.numpy(), sv.astype(np.float16))
    print("C_OK")

    # ---- D: this build's tensor-parallel shard reads == slices of what the reference loaded whole
    dd = os.path.join(d_a, "cb2")
    rstc, mine = RefSTC(dd, load_method="python"), L.SafetensorsCollection(dd)
    key = "model.layers.1.mlp.gate_proj"
    whole = ref_load(rstc, key, 256, 384)
    for first, last in ((0, 128), (128, 384)):
        s = L.load_linear_exl3(mine, key, "cpu", tp_slice=(first, last, "n"))
        assert eq(s.trellis, whole.trellis[:, first // 16: last // 16].contiguous()) and eq(s.svh, whole.svh[first:last]) and eq(s.suh, whole.suh)
    key = "model.layers.1.mlp.down_proj"
    whole = ref_load(rstc, key, 384, 256)
    for first, last in ((0, 256), (256, 384)):
        s = L.load_linear_exl3(mine, key, "cpu", tp_slice=(first, last, "k"))
        assert eq(s.trellis, whole.trellis[first // 16: last // 16].contiguous()) and eq(s.suh, whole.suh[first:last]) and eq(s.svh, whole.svh)
    print("D_OK")
print("REFERENCE_LOADER_OK")
