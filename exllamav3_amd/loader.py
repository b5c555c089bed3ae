"""
On-disk format reader for EXL3 checkpoints (reference: loader/safetensors.py SafetensorsCollection,
modules/linear.py:389-425 Linear.load_exl3, modules/quant/exl3.py:16-112).

A quantized Linear `key` is stored as
    key.trellis   int16 (k/16, n/16, 16*K)         packed codebook states (the layout the HIP kernels read directly: no re-packing at load)
    key.suh / key.svh   fp16 (k) / (n)             input / output sign*scale vectors      (newer checkpoints)
    key.su  / key.sv    int16 packed sign bits     (older checkpoints: bit i of word j = sign of channel 16 j + i; unpacked to +-1 here,
                                                    modules/quant/exl3.py:142-158)
    key.mcg / key.mul1  marker tensors             codebook selection (absent: 3INST)
    key.bias      fp16 (n), optional
Tensors go straight from the safetensors file to the target device; nothing is converted except the legacy sign bitfields.
"""
from __future__ import annotations
import glob
import json
import os
import torch
from safetensors import safe_open
from .linear import LinearEXL3


class SafetensorsCollection:
    """Index of every tensor in the *.safetensors files of a directory (lazy: tensors are read on demand)."""

    def __init__(self, directory: str):
        self.directory = directory
        self.files = sorted(glob.glob(os.path.join(directory, "*.safetensors")))
        if not self.files:
            raise FileNotFoundError(f"no .safetensors files in {directory}")
        self.index: dict[str, str] = {}
        for f in self.files:
            with safe_open(f, framework="pt", device="cpu") as st:
                for k in st.keys():
                    self.index[k] = f
        qc = os.path.join(directory, "quantization_config.json")
        self.quantization_config = json.load(open(qc)) if os.path.exists(qc) else None

    def has_tensor(self, key: str) -> bool:
        return key in self.index

    def get_tensor(self, key: str, device: str | torch.device = "cpu", optional: bool = False) -> torch.Tensor | None:
        f = self.index.get(key)
        if f is None:
            if optional:
                return None
            raise KeyError(key)
        with safe_open(f, framework="pt", device=str(device)) as st:
            return st.get_tensor(key)

    def linear_keys(self) -> list[str]:
        return sorted(k[: -len(".trellis")] for k in self.index if k.endswith(".trellis"))


def unpack_bf(bitfield: torch.Tensor) -> torch.Tensor:
    """int16 packed sign bits -> fp16 vector of +-1 (modules/quant/exl3.py:142-158): bit i of word j is channel 16 j + i, set = negative."""
    b = bitfield.to(torch.int32) & 0xFFFF
    bits = (b.unsqueeze(-1) >> torch.arange(16, dtype=torch.int32, device=b.device)) & 1
    return (1.0 - 2.0 * bits.reshape(-1).to(torch.float32)).to(torch.float16)


def is_exl3_storage(stc: SafetensorsCollection, key: str) -> bool:
    return (stc.has_tensor(key + ".trellis") and (stc.has_tensor(key + ".suh") or stc.has_tensor(key + ".su"))
            and (stc.has_tensor(key + ".svh") or stc.has_tensor(key + ".sv")))


def load_linear_exl3(stc: SafetensorsCollection, key: str, device: str | torch.device, out_dtype: torch.dtype | None = None) -> LinearEXL3:
    """Linear.load_exl3 (modules/linear.py:389-425): builds the inner LinearEXL3 of `key` on `device`."""
    if not is_exl3_storage(stc, key):
        raise KeyError(f"{key}: not an EXL3 tensor group")
    trellis = stc.get_tensor(key + ".trellis", device)
    if trellis.dtype != torch.int16 or trellis.dim() != 3 or trellis.shape[2] % 16 != 0:
        raise ValueError(f"{key}.trellis: expected int16 (k/16, n/16, 16*K)")
    suh = stc.get_tensor(key + ".suh", device, optional=True)
    if suh is None:
        suh = unpack_bf(stc.get_tensor(key + ".su", "cpu")).to(device)
    svh = stc.get_tensor(key + ".svh", device, optional=True)
    if svh is None:
        svh = unpack_bf(stc.get_tensor(key + ".sv", "cpu")).to(device)
    k, n = trellis.shape[0] * 16, trellis.shape[1] * 16
    if suh.numel() != k or svh.numel() != n:
        raise ValueError(f"{key}: scale vectors do not match the trellis shape")
    bias = stc.get_tensor(key + ".bias", device, optional=True)
    return LinearEXL3(k, n, trellis.contiguous(), suh.to(torch.float16).contiguous(), svh.to(torch.float16).contiguous(),
                      mcg=stc.has_tensor(key + ".mcg"), mul1=stc.has_tensor(key + ".mul1"), bias=bias, out_dtype=out_dtype)
