"""
CPU BASELINE (test infrastructure, NOT product code): the oracle's reconstruct + matmul restated in torch so that it runs on ALL
host cores (SURVEY.md 8d "CPU baseline timing": torch.set_num_threads(os.cpu_count()), median of >= 5 runs after a warm-up, reconstruct
time and matmul time reported separately).  Same algorithm, same citations as oracle/exl3_oracle.py (unpack_trellis: quant/exl3_dq.cuh:15-31;
codebooks: quant/codebook.cuh:56-90; tile permutation: exl3_lib/quantize.py:21-44; forward: modules/quant/exl3.py:114-139,227-237); checked
against the numpy oracle bit for bit (reconstruct) by tests/test_oracle_pins.py.  Only bench.py's cpu_baseline leg and tests import it.
"""
from __future__ import annotations
import time
import numpy as np
import torch

MUL1_MULT, MCG_MULT = 0x83DCD12D, 0xCBAC1FED


def _perm() -> torch.Tensor:
    t = torch.arange(256)
    l, j = t // 8, t % 8
    return ((2 * (l % 4) + (j & 1) + 8 * ((j >> 1) & 1)) * 16 + l // 4 + 8 * (j >> 2)).long()


def reconstruct(trellis: torch.Tensor, K: int, cb: int) -> torch.Tensor:
    """Packed int16 (k/16, n/16, 16K) -> W_hat (k, n) fp16, rotated basis (= oracle.reconstruct)."""
    kt, nt, _ = trellis.shape
    w16 = (trellis.to(torch.int64) & 0xFFFF).view(kt, nt, 8 * K, 2)
    w = w16[..., 0] | (w16[..., 1] << 16)                                    # little-endian u32 words as int64
    nw = 8 * K
    t = torch.arange(256, dtype=torch.int64)
    b0 = t * K + K - 16 + 256 * K
    b1 = b0 + 16
    i0, i1 = (b0 // 32) % nw, ((b1 - 1) // 32) % nw
    sh = (((b1 - 1) // 32) + 1) * 32 - b1
    st = (((w[..., i0] << 32) | w[..., i1]) >> sh) & 0xFFFF                  # (kt, nt, 256) states, stream order
    if cb == 2:
        x = (st * MUL1_MULT) & 0xFFFFFFFF
        b = (x & 0xFF) + ((x >> 8) & 0xFF) + ((x >> 16) & 0xFF) + (x >> 24)
        h = (0x6400 + b).to(torch.int16).view(torch.float16).double()       # fp16(1024 + b), exact
        k_inv = torch.tensor([0x1EEE], dtype=torch.int16).view(torch.float16).double()
        k_bias = torch.tensor([0xC931 - 65536], dtype=torch.int16).view(torch.float16).double()
        v = (h * k_inv + k_bias).half()                                      # exact in f64, one rounding = hfma
    else:
        x = (st * 89226354 + 64248484) & 0xFFFFFFFF if cb == 0 else (st * MCG_MULT) & 0xFFFFFFFF
        x = (x & 0x8FFF8FFF) ^ 0x3B603B60
        lo = (x & 0xFFFF); hi = x >> 16
        as_h = lambda u: torch.where(u >= 32768, u - 65536, u).to(torch.int16).view(torch.float16).double()
        v = (as_h(lo) + as_h(hi)).half()                                     # exact in f64, one rounding = hadd
    tile = torch.empty_like(v)
    tile[..., _perm()] = v
    return tile.view(kt, nt, 16, 16).permute(0, 2, 1, 3).reshape(kt * 16, nt * 16).contiguous()


def _had128(x: torch.Tensor) -> torch.Tensor:
    """x (..., 128 * b) fp32 -> blockwise H128 (Sylvester, unscaled) by seven butterfly stages."""
    shp = x.shape
    v = x.reshape(-1, 128)
    h = 1
    while h < 128:
        v = v.view(-1, 128 // (2 * h), 2, h)
        v = torch.stack((v[:, :, 0] + v[:, :, 1], v[:, :, 0] - v[:, :, 1]), dim=2)
        h *= 2
    return v.reshape(shp)


def forward(x: torch.Tensor, w_hat: torch.Tensor, suh: torch.Tensor, svh: torch.Tensor) -> torch.Tensor:
    """y = ((x * suh) H / sqrt(128)) @ W_hat H / sqrt(128) * svh, fp16 x, fp32 math, fp16 result."""
    s = 0.088388347648
    xh = (_had128((x * suh).float()) * s).half()
    acc = xh.float() @ w_hat.float()
    return ((_had128(acc) * s).half() * svh)


def time_linear(k: int, n: int, K: int, cb: int, m: int = 1, runs: int = 5, seed: int = 0) -> dict:
    """Median reconstruct and matmul seconds of one k x n linear on the current torch thread pool (1 warm-up + `runs` timed)."""
    g = torch.Generator().manual_seed(seed)
    tr = torch.randint(-32768, 32768, (k // 16, n // 16, 16 * K), dtype=torch.int16, generator=g)
    suh = torch.where(torch.rand(k, generator=g) < 0.5, -1.0, 1.0).half()
    svh = torch.where(torch.rand(n, generator=g) < 0.5, -1.0, 1.0).half()
    x = torch.randn(m, k, generator=g).half()
    t_rec, t_mm = [], []
    for it in range(runs + 1):
        t0 = time.perf_counter(); w = reconstruct(tr, K, cb); t1 = time.perf_counter()
        y = forward(x, w, suh, svh); t2 = time.perf_counter()
        if it:
            t_rec.append(t1 - t0); t_mm.append(t2 - t1)
    assert torch.isfinite(y.float()).all()
    return {"reconstruct_s": float(np.median(t_rec)), "matmul_s": float(np.median(t_mm)), "bytes": k * n * K // 8 + 2 * (k + n)}
