"""
Host-side mirror of the reference's inner quantized-linear op `LinearEXL3`
(/root/reference/exllamav3/modules/quant/exl3.py:16-389): same path selection, slicing and TP-shard semantics, on top
of the C-ABI ops in exllamav3_amd.ext.

  forward(x, params, out_dtype):
     rows <= AUTO_RECONSTRUCT_THRESHOLD (144)  -> fused HIP GEMV / small-m GEMM (exl3.py:133-137)
     otherwise reconstruct_hgemm (exl3.py:161-218):
        rows <  FUSED_RECONSTRUCT_MIN_ROWS: had_r_128(x, suh) -> reconstruct -> hgemm -> had_r_128(y, svh)
        rows >= FUSED_RECONSTRUCT_MIN_ROWS: reconstruct_had_slice (original-basis W, both Hadamards on the matrix pipe) -> hgemm on raw x
        (the reference switches at 1024 rows; on MI355X the fused route is ahead from the first row above the small-m threshold)
        out_features > MAX_RECONSTRUCT_SLICE_N: column slices
"""
from __future__ import annotations
import os
import torch
from . import ext

AUTO_RECONSTRUCT_THRESHOLD = 144          # exl3.py:10
MAX_RECONSTRUCT_SLICE_N = 32768           # exl3.py:11
# exl3.py:184 switches to the fused route at 1024 rows.  Measured on MI355X (tools/bench_mid_rows.py, Llama-3.1-8B shapes, 144..512 rows) the fused
# route is 20-35 % faster than had_r_128 + reconstruct + hgemm + had_r_128 at every row count, so it starts right above the small-m threshold.
FUSED_RECONSTRUCT_MIN_ROWS = AUTO_RECONSTRUCT_THRESHOLD + 1


#: the fused prefill GEMMs reconstruct their matrices with ONE launch (ext.reconstruct_had_multi_t); EXL3_HIP_RECON_MULTI=0: one launch per matrix
RECON_MULTI = os.environ.get("EXL3_HIP_RECON_MULTI", "1") != "0"


def _same_kind(*lins) -> bool:
    """Same bits per weight and codebook (what one multi-matrix launch needs), and the multi-matrix launch switched on."""
    return RECON_MULTI and len({(l.K, bool(l.mcg), bool(l.mul1)) for l in lins}) == 1


class LinearEXL3:
    quant_type = "exl3"

    def __init__(self, in_features: int, out_features: int, trellis: torch.Tensor, suh: torch.Tensor, svh: torch.Tensor,
                 mcg: bool = False, mul1: bool = False, bias: torch.Tensor | None = None, out_dtype: torch.dtype | None = None,
                 key: str | None = None):
        assert trellis.dtype == torch.int16 and trellis.dim() == 3, "trellis must be a 3-D int16 tensor"
        assert suh.dtype == torch.half and svh.dtype == torch.half, "suh / svh must be float16"
        assert trellis.shape[0] * 16 == in_features and trellis.shape[1] * 16 == out_features, "trellis shape mismatch"
        if bias is not None and bias.dtype == torch.float:
            bias = bias.to(torch.half)
        self.in_features, self.out_features = in_features, out_features
        self.trellis, self.suh, self.svh, self.bias = trellis.contiguous(), suh.contiguous(), svh.contiguous(), bias
        self.K = trellis.shape[-1] // 16
        self.mcg, self.mul1 = bool(mcg), bool(mul1)
        self.out_dtype = out_dtype
        self.default_out_dtype = out_dtype or torch.half
        self.key = key
        self.bc = ext.BC_LinearEXL3(self.trellis, self.suh, self.svh, self.K, self.bias, self.mcg, self.mul1, None)

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, params: dict | None = None, out_dtype: torch.dtype | None = None) -> torch.Tensor:
        params = params or {}
        assert x.is_contiguous(), f"LinearEXL3 {self.key}: non-contiguous input {tuple(x.shape)}"      # exl3.py:130
        if not params.get("reconstruct"):
            rows = x.numel() // x.shape[-1]
            if rows <= AUTO_RECONSTRUCT_THRESHOLD or params.get("no_reconstruct"):
                dtype = out_dtype or self.default_out_dtype
                return self.bc.run_alloc(x, self.out_features, dtype == torch.float)
        return self.reconstruct_hgemm(x, out_dtype)

    def reconstruct_hgemm(self, x: torch.Tensor, out_dtype: torch.dtype | None) -> torch.Tensor:
        shape = x.shape
        rows = x.numel() // shape[-1]
        y = torch.empty(shape[:-1] + (self.out_features,), dtype=out_dtype or self.default_out_dtype, device=x.device)
        x2 = x.view(rows, self.in_features)
        y2 = y.view(rows, self.out_features)
        use_fused = self.in_features % 128 == 0 and self.out_features % 128 == 0 and rows >= FUSED_RECONSTRUCT_MIN_ROWS
        if use_fused:
            xh = x2
        else:
            xh = torch.empty_like(x2)
            ext.had_r_128(x2, xh, self.suh, None, 1.0)
        dev = self.trellis.device
        if self.out_features <= MAX_RECONSTRUCT_SLICE_N:
            if use_fused:
                ext.hgemm_nt(xh, self._reconstructed_w(), y2)           # W^T, k contiguous: both GEMM operands K-major
                self._release_w()
            else:
                w = torch.empty((self.in_features, self.out_features), dtype=torch.half, device=dev)
                ext.reconstruct(w, self.trellis, self.K, self.mcg, self.mul1)
                ext.hgemm(xh, w, y2)
        else:
            step = (MAX_RECONSTRUCT_SLICE_N // 128) * 128
            w_ = torch.empty((self.in_features * step,), dtype=torch.half, device=dev)
            for n0 in range(0, self.out_features, step):
                n1 = min(n0 + step, self.out_features)
                w = w_[: self.in_features * (n1 - n0)].view(self.in_features, n1 - n0)
                if use_fused:
                    ext.reconstruct_had_slice(w, self.trellis, self.suh, self.svh[n0:], self.K, self.mcg, self.mul1, n0)
                else:
                    ext.reconstruct_slice(w, self.trellis, self.K, self.mcg, self.mul1, n0)
                ext.hgemm(xh, w, y2[:, n0:n1])
        if not use_fused:
            ext.had_r_128(y2, y2, None, self.svh, 1.0)
        if self.bias is not None:
            y += self.bias
        return y

    def forward_add_residual(self, x: torch.Tensor, resid: torch.Tensor) -> None:
        """resid (fp16, in place) += linear(x) for the prefill route: reconstruct_had_slice + hgemm whose epilogue adds the residual
        (the reference runs the GEMM with fp32 output and adds afterwards, modules/quant/exl3.py:161-218 + rms_norm_res_in / add;
        the sum is rounded to fp16 once either way).  Saves the fp32 (tokens, hidden) round trip and the add launch."""
        rows = x.numel() // x.shape[-1]
        fused = (self.in_features % 128 == 0 and self.out_features % 128 == 0 and rows >= FUSED_RECONSTRUCT_MIN_ROWS
                 and self.out_features <= MAX_RECONSTRUCT_SLICE_N and self.bias is None and rows > AUTO_RECONSTRUCT_THRESHOLD)
        if not fused:
            # forward() wants a contiguous x; prefill_chunk hands over strided column ranges of the fused q|k|v output
            y = self.forward(x if x.is_contiguous() else x.contiguous(), out_dtype=torch.float)
            ext.add(resid, y.view(resid.shape))
            return
        x2 = x if (x.dim() == 2 and x.stride(1) == 1) else x.view(rows, self.in_features)     # a strided 2-D column range is taken as it is
        ext.hgemm_nt(x2, self._reconstructed_w(), resid.view(rows, self.out_features), accumulate=True)
        self._release_w()

    @staticmethod
    def forward_gate_up_silu(gate: "LinearEXL3", up: "LinearEXL3", x: torch.Tensor) -> torch.Tensor:
        """a = silu(gate(x)) * up(x) for the prefill route with ONE GEMM: both W^T are reconstructed into one (2n, k) buffer (rows of W^T
        stack along n for free), hgemm_nt writes (rows, 2n), silu_mul_2d reads the two column halves.  Same values as two forwards + silu_mul
        (the library may pick another tile for the wider GEMM: fp32 summation order only)."""
        rows = x.numel() // x.shape[-1]
        k, n = gate.in_features, gate.out_features
        ok = (up.in_features == k and up.out_features == n and k % 128 == 0 and n % 128 == 0 and rows >= FUSED_RECONSTRUCT_MIN_ROWS
              and 2 * n <= 2 * MAX_RECONSTRUCT_SLICE_N and gate.bias is None and up.bias is None and LinearEXL3.ahead is None
              and not LinearEXL3.cache_reconstructed)
        dev = x.device
        a = torch.empty(x.shape[:-1] + (n,), dtype=torch.half, device=dev)
        if not ok:
            ext.silu_mul(gate.forward(x), up.forward(x), a)
            return a
        wt = torch.empty((2 * n, k), dtype=torch.half, device=dev)
        x2 = x.view(rows, k)
        if (_same_kind(gate, up) and rows >= 256 and k % 64 == 0 and ext.gemm_nt_own_default() and x2.stride(0) % 8 == 0 and x2.data_ptr() % 16 == 0
                and ext.gemm_nt_own_fills_chip(rows, 2 * n, x2.device, wide_only=True)):
            # the own GEMM's fused epilogue: W^T with the 128-row blocks of gate and up alternating (every 256-row tile = the gate | up rows of the same 128
            # outputs), silu(g) * u applied to the tile before it leaves the CU -- no (rows, 2n) round trip, no activation launch
            ext.reconstruct_had_multi_t(wt, [gate.trellis, up.trellis], [gate.suh, up.suh], [gate.svh, up.svh], gate.K, gate.mcg, gate.mul1, True)
            ext.gemm_nt_mfma(x2, wt, a.view(rows, n), 2)
            return a
        if _same_kind(gate, up):
            ext.reconstruct_had_multi_t(wt, [gate.trellis, up.trellis], [gate.suh, up.suh], [gate.svh, up.svh], gate.K, gate.mcg, gate.mul1)
        else:
            ext.reconstruct_had_slice_t(wt[:n], gate.trellis, gate.suh, gate.svh, gate.K, gate.mcg, gate.mul1, 0)
            ext.reconstruct_had_slice_t(wt[n:], up.trellis, up.suh, up.svh, up.K, up.mcg, up.mul1, 0)
        y = torch.empty((rows, 2 * n), dtype=torch.half, device=dev)
        ext.hgemm_nt(x.view(rows, k), wt, y)
        ext.silu_mul_2d(y[:, :n], y[:, n:], a.view(rows, n))
        return a

    @staticmethod
    def forward_multi(lins: list["LinearEXL3"], x: torch.Tensor) -> list[torch.Tensor] | None:
        """[lin(x) for lin in lins] for the prefill route with ONE GEMM: the W^T of all matrices are reconstructed into one (sum n, k) buffer
        and hgemm_nt writes one (rows, sum n) output; the results are its column ranges (row stride sum n: consumers take strided views --
        ext.rope_strided, ext.quant_cache_paged_strided, hgemm_nt's A operand).  q|k|v of Llama-3.1-8B: 96 + 38 + 38 us -> one GEMM at the wide
        shape's efficiency (k / v alone run at 0.9 PFLOP/s).  Returns None when the route does not apply (caller falls back to separate forwards)."""
        rows = x.numel() // x.shape[-1]
        k = lins[0].in_features
        ok = (all(l.in_features == k and l.out_features % 128 == 0 and l.bias is None for l in lins) and k % 128 == 0
              and rows >= FUSED_RECONSTRUCT_MIN_ROWS and sum(l.out_features for l in lins) <= MAX_RECONSTRUCT_SLICE_N
              and LinearEXL3.ahead is None and not LinearEXL3.cache_reconstructed)
        if not ok:
            return None
        dev = x.device
        ntot = sum(l.out_features for l in lins)
        wt = torch.empty((ntot, k), dtype=torch.half, device=dev)
        if len(lins) <= 4 and _same_kind(*lins):
            # one launch for all of them (k and v alone are one workgroup per CU: 10.4 us each for 2 MB of W^T)
            ext.reconstruct_had_multi_t(wt, [l.trellis for l in lins], [l.suh for l in lins], [l.svh for l in lins], lins[0].K, lins[0].mcg, lins[0].mul1)
        else:
            n0 = 0
            for l in lins:
                ext.reconstruct_had_slice_t(wt[n0: n0 + l.out_features], l.trellis, l.suh, l.svh, l.K, l.mcg, l.mul1, 0)
                n0 += l.out_features
        y = torch.empty((rows, ntot), dtype=torch.half, device=dev)
        ext.hgemm_nt(x.view(rows, k), wt, y)
        outs, n0 = [], 0
        for l in lins:
            outs.append(y[:, n0: n0 + l.out_features])
            n0 += l.out_features
        return outs

    #: MI355X option (not in the reference): keep the reconstructed original-basis fp16 W of every Linear resident after its first
    #: prefill use -- 16 GB for an 8B model, 141 GB for 70B, both fit the 288 GB of one MI355X next to the packed weights -- so later
    #: prefill chunks are pure MFMA GEMMs.  Off by default: the reference reconstructs per forward (exl3.py:161-218).
    cache_reconstructed = False

    #: optional ReconstructAhead (below) that rebuilds W on a side stream ahead of use
    ahead = None

    def _release_w(self) -> None:
        if LinearEXL3.ahead is not None:
            LinearEXL3.ahead.release(self)

    def _reconstructed_w(self) -> torch.Tensor:
        if LinearEXL3.ahead is not None:
            w = LinearEXL3.ahead.acquire(self)
            if w is not None:
                return w
        w = getattr(self, "_w_cache", None)
        if w is not None:
            return w
        w = torch.empty((self.out_features, self.in_features), dtype=torch.half, device=self.trellis.device)     # W^T
        ext.reconstruct_had_slice_t(w, self.trellis, self.suh, self.svh, self.K, self.mcg, self.mul1, 0)
        if self.cache_reconstructed:
            self._w_cache = w
        return w

    # ---- weights -----------------------------------------------------------------------------------
    def get_inner_weight_tensor(self) -> torch.Tensor:
        w = torch.empty((self.in_features, self.out_features), dtype=torch.half, device=self.trellis.device)
        ext.reconstruct(w, self.trellis, self.K, self.mcg, self.mul1)
        return w

    def get_weight_tensor(self) -> torch.Tensor:
        """exl3.py:227-237: original-basis weights."""
        w = torch.empty((self.in_features, self.out_features), dtype=torch.half, device=self.trellis.device)
        ext.reconstruct_had_slice(w, self.trellis, self.suh, self.svh, self.K, self.mcg, self.mul1, 0)
        return w

    # ---- tensor-parallel shards (exl3.py:284-330 tp_import_split) --------------------------------------
    def tp_shard(self, first: int, last: int, split_dim: str) -> "LinearEXL3":
        """Column ('n', out-features) or row ('k', in-features) shard [first, last); boundaries are multiples of 128
        (Hadamard blocks).  Column shards slice trellis dim 1, svh and bias; row shards slice trellis dim 0 and suh, keep
        the full svh and keep the bias only on the shard that starts at 0 (so that the all-reduce adds it once)."""
        assert first % 128 == 0 and last % 128 == 0 and first < last
        if split_dim == "n":
            assert last <= self.out_features
            return LinearEXL3(self.in_features, last - first, self.trellis[:, first // 16: last // 16].contiguous(),
                              self.suh, self.svh[first:last].contiguous(), self.mcg, self.mul1,
                              None if self.bias is None else self.bias[first:last].contiguous(), self.out_dtype, self.key)
        assert split_dim == "k" and last <= self.in_features
        return LinearEXL3(last - first, self.out_features, self.trellis[first // 16: last // 16].contiguous(),
                          self.suh[first:last].contiguous(), self.svh, self.mcg, self.mul1,
                          self.bias if first == 0 else None, self.out_dtype, self.key)


class ReconstructAhead:
    """Prefill scheduler piece (no counterpart in the reference, which reconstructs inline on one stream): the reconstructed fp16 W of
    the next `depth` Linears of a known sequence is rebuilt on a second HIP stream while the GEMM of the current one runs on the main
    stream.  The GEMM is MFMA-bound, reconstruct_had_slice is VALU / latency-bound: the two share the chip well.  W never depends on the
    activations, so the only ordering is buffer reuse (ring of depth + 1 buffers) -- expressed with HIP events, no host synchronisation.

        ahead = ReconstructAhead(sequence_of_linears); LinearEXL3.ahead = ahead; ahead.begin(); ...forward calls in that order...;
        ahead.end(); LinearEXL3.ahead = None
    """

    def __init__(self, linears, depth: int = 2):
        self.seq = list(linears)
        self.nbuf = depth + 1
        dev = self.seq[0].trellis.device
        cap = max(l.in_features * l.out_features for l in self.seq)
        self.bufs = [torch.empty((cap,), dtype=torch.half, device=dev) for _ in range(self.nbuf)]
        self.side = torch.cuda.Stream(device=dev)
        self.ready, self.released, self.w = {}, {}, {}
        self.pos = 0

    def _issue(self, i: int):
        if i >= len(self.seq):
            return
        lin = self.seq[i]
        with torch.cuda.stream(self.side):
            if i - self.nbuf in self.released:
                self.side.wait_event(self.released.pop(i - self.nbuf))       # the GEMM that read this buffer last has finished
            w = self.bufs[i % self.nbuf][: lin.in_features * lin.out_features].view(lin.out_features, lin.in_features)     # W^T
            ext.reconstruct_had_slice_t(w, lin.trellis, lin.suh, lin.svh, lin.K, lin.mcg, lin.mul1, 0)
            ev = torch.cuda.Event(); ev.record(self.side)
        self.ready[i], self.w[i] = ev, w

    def begin(self):
        self.pos = 0
        self.side.wait_stream(torch.cuda.current_stream())
        for i in range(self.nbuf):
            self._issue(i)

    def acquire(self, lin):
        if self.pos >= len(self.seq) or self.seq[self.pos] is not lin:
            return None                                                       # not part of the announced sequence: caller reconstructs inline
        torch.cuda.current_stream().wait_event(self.ready.pop(self.pos))
        return self.w.pop(self.pos)

    def release(self, lin):
        if self.pos >= len(self.seq) or self.seq[self.pos] is not lin:
            return
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
        self.released[self.pos] = ev
        i = self.pos
        self.pos += 1
        self._issue(i + self.nbuf)

    def end(self):
        torch.cuda.current_stream().wait_stream(self.side)



class Linear:
    """The module-level wrapper of the reference (modules/linear.py:32-120,561-602) around a quantized inner op, host logic only:
      * in / out features are padded up to a multiple of `pad_to` (128 = the Hadamard block; modules/linear.py:69-72) -- the stored tensors
        already have the padded shape;
      * an input narrower than the padded weight is zero-extended (exact: the padded weight rows are zeros, :575-576);
      * padded output columns are trimmed (contiguous copy) when `trim_padded_out` (:589-590);
      * softcap (ext.softcap) and post_scale are applied after the inner op (:598-601).
    LoRA, H capture and the unquantized / fp16 inner types are outside this path."""

    def __init__(self, inner: LinearEXL3, in_features: int, out_features: int, pad_to: int = 128, trim_padded_out: bool = True,
                 softcap: float = 0.0, post_scale: float = 1.0, out_dtype: torch.dtype | None = None):
        self.in_features_unpadded, self.out_features_unpadded = in_features, out_features
        self.in_features = (in_features + pad_to - 1) // pad_to * pad_to
        self.out_features = (out_features + pad_to - 1) // pad_to * pad_to
        if (inner.in_features, inner.out_features) != (self.in_features, self.out_features):
            raise RuntimeError(f"Linear: inner op is {inner.in_features} x {inner.out_features}, expected the padded "
                               f"{self.in_features} x {self.out_features}")
        self.inner, self.trim_padded_out, self.softcap, self.post_scale, self.out_dtype = inner, trim_padded_out, softcap, post_scale, out_dtype

    def forward(self, x: torch.Tensor, out_dtype: torch.dtype | None = None) -> torch.Tensor:
        if self.out_features == 0:
            return x.new_empty((*x.shape[:-1], 0), dtype=out_dtype or self.out_dtype or torch.half)
        if x.shape[-1] < self.in_features:
            x = torch.nn.functional.pad(x, (0, self.in_features - x.shape[-1]))
        y = self.inner.forward(x.contiguous(), out_dtype=out_dtype or self.out_dtype)
        if self.trim_padded_out and self.out_features != self.out_features_unpadded:
            y = y[..., :self.out_features_unpadded].contiguous()
        if self.softcap != 0.0:
            ext.softcap(y, y, self.softcap)
        if self.post_scale != 1.0:
            y *= self.post_scale
        return y
