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

    def get_slice(self, key: str, dim: int, first: int, last: int, device: str | torch.device = "cpu") -> torch.Tensor:
        """Elements [first, last) along `dim` of a stored tensor; only those bytes are read for dim 0 (safetensors lazy slices).
        This is how a tensor-parallel rank loads its shard: every rank (one process per GPU) reads its own slice of the file
        straight to its device -- no rank-0 load + export/import round (the reference's tp_export / tp_import,
        modules/quant/exl3.py:284-330, which exists because its ranks are spawned after the load)."""
        f = self.index.get(key)
        if f is None:
            raise KeyError(key)
        with safe_open(f, framework="pt", device=str(device)) as st:
            sl = st.get_slice(key)
            if dim == 0:
                return sl[first:last]
            if dim == 1:
                return sl[:, first:last]
            raise ValueError("get_slice: dim must be 0 or 1")

    def list_tensors(self, prefix: str) -> dict:
        """{key: {"shape", "dtype", "n_bytes"}} of every stored tensor under `prefix.` (loader/safetensors.py list_tensors)."""
        out = {}
        by_file: dict[str, list[str]] = {}
        for k, f in self.index.items():
            if k == prefix or k.startswith(prefix + "."):
                by_file.setdefault(f, []).append(k)
        for f, keys in by_file.items():
            with safe_open(f, framework="pt", device="cpu") as st:
                for k in keys:
                    sl = st.get_slice(k)
                    shape = list(sl.get_shape())
                    dt = str(sl.get_dtype())
                    esz = {"F16": 2, "BF16": 2, "F32": 4, "I16": 2, "I32": 4, "I64": 8, "U8": 1, "I8": 1, "F64": 8, "BOOL": 1}.get(dt, 0)
                    n = 1
                    for d in shape:
                        n *= d
                    out[k] = {"shape": shape, "dtype": dt, "n_bytes": n * esz}
        return dict(sorted(out.items()))

    def linear_keys(self) -> list[str]:
        return sorted(k[: -len(".trellis")] for k in self.index if k.endswith(".trellis"))

    def tensor_storage(self) -> dict:
        """The `tensor_storage` map of quantization_config.json (conversion/quant_config.py:14-55), rebuilt from the files: per EXL3
        tensor group its stored tensors, quant_format, bits_per_weight and codebook multipliers."""
        out = {}
        for key in self.linear_keys():
            d = {"stored_tensors": self.list_tensors(key), "quant_format": "exl3"}
            d["bits_per_weight"] = d["stored_tensors"][key + ".trellis"]["shape"][-1] // 16
            for marker, field in (("mul1", "mul1_multiplier"), ("mcg", "mcg_multiplier")):
                if self.has_tensor(f"{key}.{marker}"):
                    v = int(self.get_tensor(f"{key}.{marker}").view(-1)[0].item()) & 0xFFFFFFFF
                    if v:
                        d[field] = v
            out[key] = d
        return out


def unpack_bf(bitfield: torch.Tensor) -> torch.Tensor:
    """int16 packed sign bits -> fp16 vector of +-1 (modules/quant/exl3.py:142-158): bit i of word j is channel 16 j + i, set = negative."""
    b = bitfield.to(torch.int32) & 0xFFFF
    bits = (b.unsqueeze(-1) >> torch.arange(16, dtype=torch.int32, device=b.device)) & 1
    return (1.0 - 2.0 * bits.reshape(-1).to(torch.float32)).to(torch.float16)


def is_exl3_storage(stc: SafetensorsCollection, key: str) -> bool:
    return (stc.has_tensor(key + ".trellis") and (stc.has_tensor(key + ".suh") or stc.has_tensor(key + ".su"))
            and (stc.has_tensor(key + ".svh") or stc.has_tensor(key + ".sv")))


def load_linear_exl3(stc: SafetensorsCollection, key: str, device: str | torch.device, out_dtype: torch.dtype | None = None,
                     tp_slice: tuple[int, int, str] | None = None) -> LinearEXL3:
    """Linear.load_exl3 (modules/linear.py:389-425): builds the inner LinearEXL3 of `key` on `device`.
    tp_slice = (first, last, "n" | "k"): load only that column ('n', out-features) or row ('k', in-features) shard, boundaries multiples
    of 128, with LinearEXL3.tp_shard's semantics (a row shard keeps the whole svh, and the bias only where first == 0) -- the tensors are
    identical to load + tp_shard, but only the shard's bytes leave the file."""
    if not is_exl3_storage(stc, key):
        raise KeyError(f"{key}: not an EXL3 tensor group")
    if tp_slice is None:
        trellis = stc.get_tensor(key + ".trellis", device)
        k_sl = n_sl = None
    else:
        first, last, dim = tp_slice
        if first % 128 or last % 128 or not first < last or dim not in ("n", "k"):
            raise ValueError(f"{key}: tensor-parallel shards are [first, last) in multiples of 128 along 'n' or 'k'")
        trellis = stc.get_slice(key + ".trellis", 1 if dim == "n" else 0, first // 16, last // 16, device)
        if trellis.shape[1 if dim == "n" else 0] * 16 != last - first:
            raise ValueError(f"{key}: shard [{first}, {last}) exceeds the tensor")
        k_sl = (first, last) if dim == "k" else None
        n_sl = (first, last) if dim == "n" else None
    if trellis.dtype != torch.int16 or trellis.dim() != 3 or trellis.shape[2] % 16 != 0:
        raise ValueError(f"{key}.trellis: expected int16 (k/16, n/16, 16*K)")

    def scale(name_h: str, name_bits: str, sl):
        if stc.has_tensor(f"{key}.{name_h}"):
            return stc.get_tensor(f"{key}.{name_h}", device) if sl is None else stc.get_slice(f"{key}.{name_h}", 0, sl[0], sl[1], device)
        v = unpack_bf(stc.get_tensor(f"{key}.{name_bits}", "cpu"))
        return (v if sl is None else v[sl[0]: sl[1]]).to(device)

    suh = scale("suh", "su", k_sl)
    svh = scale("svh", "sv", n_sl)
    k, n = trellis.shape[0] * 16, trellis.shape[1] * 16
    if suh.numel() != k or svh.numel() != n:
        raise ValueError(f"{key}: scale vectors do not match the trellis shape")
    bias = None
    if stc.has_tensor(key + ".bias") and (k_sl is None or k_sl[0] == 0):
        bias = stc.get_tensor(key + ".bias", device) if n_sl is None else stc.get_slice(key + ".bias", 0, n_sl[0], n_sl[1], device)
    return LinearEXL3(k, n, trellis.contiguous(), suh.to(torch.float16).contiguous(), svh.to(torch.float16).contiguous(),
                      mcg=stc.has_tensor(key + ".mcg"), mul1=stc.has_tensor(key + ".mul1"), bias=bias, out_dtype=out_dtype)
