/*
 * exl3_hip.h — C-ABI of the MI355X-native EXL3 quantized-linear hot path (libexl3_hip.so).
 *
 * This is the drop-in boundary for the path named by BASELINE.json `north_star`: every entry point
 * replaces one op of the reference's native extension `exllamav3_ext` (pybind11 module,
 * /root/reference/exllamav3/exllamav3_ext/bindings.cpp:69-226).  The reference binds at::Tensor
 * arguments; here every argument is a plain device pointer / size / scalar so that any host language
 * can bind it (the Python mirror of the reference op surface lives in exllamav3_amd/ext.py and is what
 * INTEGRATION.md installs as the module `exllamav3_ext`).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on `device`-current HBM unless marked host; the caller owns all
 *     buffers (reference ownership contract: SURVEY.md 8b) and must keep them alive until the stream drains;
 *   - `stream` is a hipStream_t (NULL = default stream); all calls are asynchronous w.r.t. the GPU;
 *   - return 0 on success, <0 on error (EXL3_ERR_*); the message is available from exl3_last_error()
 *     (thread-local).  The reference raises RuntimeError via TORCH_CHECK for the same argument violations
 *     (exllamav3_ext/util.h:24-35); exllamav3_amd/ext.py converts a non-zero return into RuntimeError;
 *   - fp16 = IEEE binary16 (`half`), trellis = int16 [k/16][n/16][16K] exactly as stored in EXL3 checkpoints;
 *   - cb (codebook): 0 = 3INST, 1 = mcg, 2 = mul1  (reference passes two bools `mcg`, `mul1`;
 *     reconstruct.cu:128-130 — mcg wins).
 *   - all kernels are graph-capture safe after exl3_init(device) has run once outside capture.
 *   - ONE STREAM PER DEVICE AT A TIME.  The split-k partial slabs, the arrival tickets, the generation-3 rotated-activation scratch and the
 *     hgemm workspace are one per device (like the reference's DevCtx lock buffer + workspace, quant/exl3_devctx.cuh:8-19), and slab-writing
 *     GEMV launches alternate between two workspace regions in issue order (a launch may read its predecessor's slabs: exl3_gemv_ex_act,
 *     exl3_glue_*).  Calls on one device must therefore be issued from one host thread at a time.  When the issuing stream changes between two
 *     calls that use the workspace, the library orders the new stream behind the previous one with an event edge (so two streams never overlap
 *     inside the workspace); a change while the previous stream is still capturing a graph is refused with an error -- all launches of one graph
 *     go onto one stream.  The reference has the same contract (its A_had scratch and lock buffer are shared by every Linear of a device,
 *     SURVEY.md 8b Ownership).  Different devices are independent.
 */
#ifndef EXL3_HIP_H
#define EXL3_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXL3_OK        0
#define EXL3_ERR_ARG  (-1)   /* shape / alignment / range violation (reference: TORCH_CHECK)      */
#define EXL3_ERR_HIP  (-2)   /* HIP runtime error (reference: cuda_check, util.cuh:92-100)         */
#define EXL3_ERR_INIT (-3)   /* per-device context missing while the stream is capturing           */

#define EXL3_ABI_VERSION 4   /* bumped whenever an exported symbol is removed or changes meaning (round 3: 3; round 6: 4 -- exl3_pstep_linear_t carries K and codebook) */

const char* exl3_last_error(void);
int  exl3_abi_version(void);

/* Per-device context: split-k workspace + zero-initialised ticket words (reference: DevCtx,
 * quant/exl3_devctx.cuh:8-19, .cu:48-69).  Idempotent; must run once outside graph capture. */
int  exl3_init(int device);
/* bindings.cpp g_get_cc / g_get_num_sms equivalents: compute units and gfx arch number (950). */
int  exl3_device_info(int device, int* num_cus, int* gfx_arch, int64_t* hbm_bytes);

/* ---- format ops -------------------------------------------------------------------------------- */

/* pack_trellis(packed, unpacked, K)       quant/pack.cu:68-95.   unpacked u16 [tk][tn][256], packed i16 [tk][tn][16K] */
int exl3_pack_trellis(void* packed, const void* unpacked, int tiles_k, int tiles_n, int K, void* stream);
/* unpack_trellis(unpacked, packed, K)     quant/pack.cu:148-175 */
int exl3_unpack_trellis(void* unpacked, const void* packed, int tiles_k, int tiles_n, int K, void* stream);
/* pack_signs(packed, unpacked)            quant/pack.cu:177-226.  signs fp16 [numel] (numel % 16 == 0) -> i16 [numel/16] */
int exl3_pack_signs(void* packed, const void* signs, int64_t numel, void* stream);
/* decode(idx, out, mcg, mul1)             quant/quantize.cu:89-168.  states u16 [numel] -> fp16 or fp32 [numel] */
int exl3_decode(const void* states, void* out, int64_t numel, int out_fp32, int cb, void* stream);

/* reconstruct / reconstruct_slice         quant/reconstruct.cu:98-144,375-386.
 * out fp16 [16*tiles_k][n_size] (row stride n_size) = W_hat[:, n_offset : n_offset+n_size]; n_offset, n_size % 128 == 0 */
int exl3_reconstruct(void* out, const void* trellis, int tiles_k, int tiles_n, int K, int cb,
                     int64_t n_offset, int64_t n_size, void* stream);
/* reconstruct_had_slice                   quant/reconstruct.cu:324-373.  Original-basis weights
 * W = diag(suh) H W_hat H diag(svh); svh is pre-offset by the caller (points at element n_offset's scale). */
int exl3_reconstruct_had(void* out, const void* trellis, const void* suh, const void* svh,
                         int tiles_k, int tiles_n, int K, int cb, int64_t n_offset, int64_t n_size, void* stream);
/* reconstruct_had_slice written transposed: out[n_size][ld_out] = W^T (row = output feature, k contiguous, ld_out >= k). */
int exl3_reconstruct_had_t(void* out, int64_t ld_out, const void* trellis, const void* suh, const void* svh,
                           int tiles_k, int tiles_n, int K, int cb, int64_t n_offset, int64_t n_size, void* stream);
/* W^T of up to 4 whole matrices with the same k, bits per weight and codebook in ONE launch, stacked along n: out [sum n_i][ld_out] fp16 (matrix i's
 * rows behind matrix i - 1's), tiles_n[i] = n_i / 16, n_i % 128 == 0.  The fused q|k|v / gate|up GEMMs of the prefill route (modules/quant/exl3.py:161-218
 * reconstructs each Linear on its own) read this buffer as one B operand. */
int exl3_reconstruct_had_multi_t(void* out, int64_t ld_out, const void* const* trellis, const void* const* suh, const void* const* svh,
                                 const int* tiles_n, int count, int tiles_k, int K, int cb, void* stream);
/* The same launch with the matrices' 128-row blocks interleaved (block row j of matrix i at rows (j * count + i) * 128; equal n_i): for count = 2 every 256-row tile of
 * the output is 128 gate rows | the 128 up rows of the same outputs -- the B operand of exl3_gemm_nt2_mfma's fused silu(gate) * up epilogue (activation.cu silu_mul
 * behind modules/mlp.py's gate / up forwards; the reference runs the two GEMMs and the activation as separate launches). */
int exl3_reconstruct_had_multi_t_interleaved(void* out, int64_t ld_out, const void* const* trellis, const void* const* suh, const void* const* svh,
                                             const int* tiles_n, int count, int tiles_k, int K, int cb, void* stream);

/* had_r_128(input, output, pre_scale, post_scale, scale)   quant/hadamard.cu:88-173.
 * rows x cols, cols % 128 == 0; fp32 = 0: fp16 in/out, 1: fp32 in/out; scales fp16 [cols] or NULL; in-place allowed */
int exl3_had_r_128(const void* in, void* out, const void* pre_scale, const void* post_scale, float scale,
                   int rows, int cols, int fp32, void* stream);

/* ---- quantized GEMM / GEMV --------------------------------------------------------------------- */

/* exl3_gemm(A, B, C, suh, A_had, svh, force_shape_idx, mcg, mul1, force_num_sms)   quant/exl3_gemm.cuh:21-33
 * + bias add of BC_LinearEXL3::run (libtorch/linear.cpp:34-71).
 *   C[m][n] = ((A[m][k] * suh) H) @ dequant(B) H * svh (+ bias)
 * A fp16 [m][k] contiguous; B int16 [k/16][n/16][16K]; C fp16 or fp32 [m][n]; suh [k], svh [n], bias [n] fp16 (bias may be NULL).
 * k % 128 == 0, n % 128 == 0, 1 <= K <= 8.  Any m >= 1 (16-row passes; callers switch to reconstruct+hgemm above
 * 144 rows as modules/quant/exl3.py:135 does).  force_split: 0 = heuristic, >0 = k-split count.
 * Returns the kernel id (>= 1) on success like the reference, < 0 on error. */
int exl3_gemm(const void* A, const void* B, void* C, const void* suh, const void* svh, const void* bias,
              int m, int k, int n, int K, int cb, int c_fp32, int force_split, void* stream);

/* exl3_mgemm (broadcast form)    quant/exl3_gemm.cuh:58-78 with indices == NULL, bszm matrices sharing one A:
 * host arrays of `count` device pointers (B_i, C_i, suh_i, svh_i) and widths n_i; every matrix has the same k, K, cb.
 * This is the fused q/k/v and gate/up launch (libtorch/mlp.cpp:40, attention.cpp:286-365). */
int exl3_mgemm(const void* A, const void* const* Bs, void* const* Cs, const void* const* suhs, const void* const* svhs,
               const int* ns, int count, int m, int k, int K, int cb, int c_fp32, int force_split, void* stream);

/* Tuning hooks (no reference equivalent; the reference's knobs are env vars EXL3_GEMV / EXL3_GEMV_SMEM, doc/env_vars.md):
 * variant 0 = EXACT (reference fp16 weights bit-for-bit into the MFMA), 1 = FAST (default; see exl3_gemv2.kspec.hip). */
int exl3_set_gemv_variant(int variant);
int exl3_set_gemv_max_waves(int max_waves_per_workgroup);   /* 0 = heuristic (up to 16) */
int exl3_set_gemv_gen4(int on);                              /* 1 (default; env EXL3_HIP_GEMV_GEN4): 1..4-row launches take the generation-4 kernel
                                                                (exl3_gemv4.kspec.hip: activation quads in the A-broadcast register layout, no per-wave
                                                                prologue); 0 pins generation 2 for them */
int exl3_set_gemm3_cpw(int column_blocks_per_workgroup);    /* generation 3, <= 16-row passes: 1 / 2 / 4 column blocks (4- / 8- / 16-wave workgroups) share one activation tile; 0 = the library's cost model (default; env EXL3_HIP_GEMM3_CPW) */
int exl3_set_attn_wide_waves(int waves_per_workgroup);       /* decode attention, matrix-pipe kernel: 4 / 8 waves per workgroup (one / two dependent token-step chains per SIMD); 0 = by the split length (default; env EXL3_HIP_ATTN_WIDE_NW).  No reference counterpart: A/B runs and tests */
int exl3_set_gemm3_min_rows(int min_rows);                   /* passes with >= min_rows rows use the LDS-transpose kernel (exl3_gemm3.kspec.hip); default 5 (9 for raw input), 0 = never */
int exl3_set_gemv_defer_wg_per_cu(int workgroups_per_cu);      /* deferred-epilogue k-split target, 0 = default (2) */

/* ---- fused decode pipeline (m <= 16): the reference chains these steps as separate graph nodes inside its BC_* runners
 * (libtorch/attention.cpp:246-504 BC_Attention::run_gr, libtorch/mlp.cpp:14-91 BC_GatedMLP::run_bszN_gr); on MI355X each
 * node is a ~4.5 us latency-bound launch, so the GEMVs run with a deferred epilogue and three "glue" kernels do everything
 * in between (exl3_glue.hip).
 *
 * exl3_gemv_ex: exl3_mgemm with flags: EXL3_GEMV_IN_ROTATED (1): per-matrix pre-rotated inputs xhs[i] = had128(x * suh_i) fp16 [m][k]
 * (xsums[i], per-128-block sums fp32 [m][k/128], are accepted for compatibility and no longer read: the kernel sums the fragments it builds) instead of A; EXL3_GEMV_OUT_DEFERRED (2): no output Hadamard, raw fp32 partial
 * slabs [n_i/128][S][m][128] are left in the per-device workspace, slabs_out[i] (host array) receives their device address and
 * *S_out the split count.  Arrays are host arrays of `count` (<= 4) entries. */
#define EXL3_GEMV_IN_ROTATED   1
#define EXL3_GEMV_OUT_DEFERRED 2
#define EXL3_GEMV_OUT_ATOMIC   32  /* m <= 4: Cs[i] is a 64-bit fixed-point accumulator int64 [m][n] (value * 2^32, the residual stream of the "fx" decode
                                      pipeline below); every workgroup ADDS its split-k share of the finished rows (out-Hadamard * svh, + bias once) with
                                      integer atomics: no split-k reduce, no residual launch, bit-reproducible (integer addition is associative).  Replaces
                                      the gemm epilogue + `x += y` of the reference's decode graph (libtorch/attention.cpp, libtorch/mlp.cpp:14-91) */
int exl3_gemv_ex(const void* A, const void* const* xhs, const float* const* xsums, const void* const* Bs, void* const* Cs,
                 const void* const* suhs, const void* const* svhs, const void* const* biases, const int* ns, int count,
                 int m, int k, int K, int cb, int c_fp32, int flags, int force_split, float** slabs_out, int* S_out, void* stream);
/* glue 1: [y = reduce slabs + out-Hadamard + svh (+bias), or y = y_dense fp32 [m][hidden] (e.g. after a TP all-reduce);
 * residual(fp16) += y]  (skipped when both are NULL) -> RMSNorm(w, eps) (norm.cu rms_norm_res_in semantics) -> for each of `count`
 * (<= 3) consumers: xh_i = had128(xn * suh_i), xsum_i.  xn_out optional. */
int exl3_glue_norm(const float* y_slabs, int y_S, const float* y_dense, const void* svh, const void* bias, void* resid, const void* w, float eps,
                   const void* const* suhs, void* const* xhs, float* const* xsums, int count, int m, int hidden,
                   void* xn_out, void* stream);
/* glue 2: q/k/v epilogue (head_dim 128): reduce + out-Hadamard + svh -> rope (rope.cu semantics, positions[m]) on q, k -> q_out fp16,
 * k/v -> quantized paged-cache append at positions[m] (q_cache_kernels.cuh semantics; k_cache == NULL skips) and optional fp16 k_out/v_out. */
int exl3_glue_qkv(const float* sq, const float* sk, const float* sv, int S, const void* svh_q, const void* svh_k, const void* svh_v,
                  void* q_out, void* k_out, void* v_out, const float* inv_freq, const int32_t* positions,
                  void* k_cache, void* k_scales, void* v_cache, void* v_scales, const int32_t* block_table, int blocks_per_seq,
                  int page_size, int k_bits, int v_bits, int m, int heads_q, int heads_kv, int head_dim, int rope_mode,
                  float attn_factor, void* stream);
/* glue 3: gate/up epilogue: reduce + out-Hadamard + svh for g and u -> a = silu(g) * u (activation.cu) -> xh_d = had128(a * suh_down), xsum_d. */
int exl3_glue_act(const float* sg, const float* su, int S, const void* svh_g, const void* svh_u, const void* suh_d,
                  void* xh_d, float* xsum_d, void* a_out, int m, int inter, void* stream);

/* exl3_mgemm with pointer tables, indices and routing weights (MoE form)   quant/exl3_gemm.cuh:58-78, exl3_gemm_kernel.cuh:88-292
 * tbl_B / tbl_suh / tbl_svh: device arrays of device addresses (int64), one entry per matrix (all k x n, same K and codebook).
 * Slot j of bszm uses matrix indices[j] (device int64; NULL: matrix j).  min_index >= 0: only indices in [min_index, max_index) run,
 * compacted to the front in order and re-based to min_index (expert-parallel ranks); the other slots are left unwritten.
 * A [bszm_in][m][k] fp16 with bszm_in == 1 (shared input) or bszm; C [bszm][m][n] fp16/fp32.
 * weights (fp16 [bszm], NULL: none): slot output scaled by its weight inside the output-Hadamard scale, then every group of
 * bszm / num_tokens consecutive slots is summed into C[t] (fp16: sequential half adds, fp32: float adds).  m <= 16. */
int exl3_mgemm_indexed(const void* A, int bszm_in, const void* tbl_B, const void* tbl_suh, const void* tbl_svh,
                       const int64_t* indices, const void* weights, int bszm, void* C, int m, int k, int n, int K, int cb, int c_fp32,
                       int min_index, int max_index, int num_tokens, void* stream);

/* routing_std(hidden, gate, scores, topk_indices, topk_weights, per_expert_scale, gate_t, bias)   routing.cu:955-1010, kernel :457-590
 * scores[bsz][E] fp16 = hidden[bsz][H] @ gate[H][E]; top-K logits (descending, ties to the lower index) -> topk_indices int64 [bsz][K],
 * topk_weights fp16 [bsz][K] = softmax over the K selected logits.  bias (fp16 [E], optional) is added to the logits before selection. */
int exl3_routing_std(const void* hidden, const void* gate, const void* bias, void* scores, int64_t* topk_indices, void* topk_weights,
                     int bsz, int hidden_size, int num_experts, int K, void* stream);

/* ---- GEMV launches with an in-kernel tail epilogue (decode, m <= 16) ------------------------------------------------------
 * The workgroup that finishes a 128-column block last (device-memory arrival ticket) reduces the split-k partials of that
 * block and runs the sublayer boundary the reference executes as separate graph nodes between two exl3_gemm calls
 * (libtorch/attention.cpp:246-504, libtorch/mlp.cpp:14-91).  Inputs: pre-rotated (xh = had128(x * suh), xsum = per-block sums,
 * as produced by these same epilogues / exl3_glue_norm) or raw (A + suh).  Results are bit-identical to the
 * exl3_gemv_ex(GEMV_OUT_DEFERRED) + exl3_glue_* pair. */

/* o_proj / down_proj:  y = linear(x) (fp32, + bias) ; resid += y (fp16, norm.cu:193-218 rms_norm_res_in semantics) ;
 * xn = rms_norm(resid) * norm_w ; for each of t_count consumers: t_xhs[i] = had128(xn * t_suhs[i]), t_xsums[i] = block sums. */
/* o_proj / down_proj with the glue_resid step inside the launch (agent-scope tail hand-off; kept on one XCD only after exl3_set_tail_xcd_local(1) and when n/128 % 8 == 0):
 * resid (fp16 [m][n], in place) += linear(x); ss_out [m][n/128] = per-block sums of squares of the new residual. */
int exl3_gemv_resid(const void* A, const void* xh, const float* xsum, const void* B, const void* suh, const void* svh, const void* bias,
                    int m, int k, int n, int K, int cb, void* resid, float* ss_out, int force_split, void* stream);
/* XCD-local tail hand-off of exl3_gemv_resid: OFF by default (agent-scope, placement-independent hand-off).  enable != 0 first probes that
 * workgroup i runs on XCD i % 8 on this device (not a HIP guarantee) and returns EXL3_ERR_ARG, leaving the mode off, if any probe workgroup did not. */
int exl3_set_tail_xcd_local(int enable);
int exl3_gemv_norm(const void* A, const void* xh, const float* xsum, const void* B, const void* suh, const void* svh, const void* bias,
                   int m, int k, int n, int K, int cb, void* resid, const void* norm_w, float eps,
                   const void* const* t_suhs, void* const* t_xhs, float* const* t_xsums, int t_count, void* xn_out, void* stream);
/* gate_proj + up_proj (Bs[0], Bs[1]): a = fp16(silu(g) * u) (activation.cu) ; xh_d = had128(a * suh_d), xsum_d ; a_out optional. */
int exl3_gemv_act(const void* A, const void* const* xhs, const float* const* xsums, const void* const* Bs, const void* const* suhs,
                  const void* const* svhs, int m, int k, int inter, int K, int cb,
                  const void* suh_d, void* xh_d, float* xsum_d, void* a_out, void* stream);
/* q/k/v projections (Bs[0..2]), head_dim 128: RoPE on q and k from the per-step table (exl3_rope_table), q_out fp16
 * [m][heads_q*128]; k, v appended to the quantized paged cache (cache.cu quant_cache_paged semantics) and/or written to k_out/v_out. */
int exl3_gemv_qkv(const void* A, const void* const* xhs, const float* const* xsums, const void* const* Bs, const void* const* suhs,
                  const void* const* svhs, int m, int k, int K, int cb, void* q_out, void* k_out, void* v_out,
                  const float* rope_sin, const float* rope_cos, const int32_t* positions,
                  void* k_cache, void* k_scales, void* v_cache, void* v_scales, const int32_t* block_table, int blocks_per_seq,
                  int page_size, int k_bits, int v_bits, int heads_q, int heads_kv, int head_dim, int rope_mode, void* stream);
/* sin_out/cos_out[m][64] = sincosf(inv_freq[f] * positions[row]) * attn_factor (rope.cu:60-120 evaluates the same per launch). */
int exl3_rope_table(const float* inv_freq, const int32_t* positions, float attn_factor, int m, float* sin_out, float* cos_out, void* stream);

/* exl3_gemv_ex whose input is the fp16 residual stream: x = rms_norm(resid) * norm_w (norm.cu:20-120) is formed inside the GEMV from
 * the per-block sums of squares ss_part [m][k/128] (exl3_glue_resid), then (x * suh) H as usual.  flags: EXL3_GEMV_OUT_DEFERRED optional. */
int exl3_gemv_ex_norm(const void* resid, const void* norm_w, const float* ss_part, float eps, const void* const* Bs, void* const* Cs,
                      const void* const* suhs, const void* const* svhs, const void* const* biases, const int* ns, int count,
                      int m, int k, int K, int cb, int c_fp32, int flags, int force_split, float** slabs_out, int* S_out, void* stream);
/* glue 1a: [reduce y_slabs + out-Hadamard + svh (+bias) | y_dense] -> resid += y (fp16, norm.cu:193-218) -> ss_part[row][block] =
 * sum of squares of each 128-block of the updated residual.  y_slabs == y_dense == NULL: sums of squares only. */
int exl3_glue_resid(const float* y_slabs, int y_S, const float* y_dense, const void* svh, const void* bias, void* resid,
                    float* ss_part, int m, int hidden, void* stream);

/* glue 1b (batches above 4 rows): xh_i = had128(rms_norm(resid) * w * suh_i) for up to 3 consumers from resid + ss_part (exl3_glue_resid);
 * distributed over (row, block); the consumer GEMVs then run with EXL3_GEMV_IN_ROTATED.  xsums optional. */
int exl3_set_glue_threads(int threads);   /* tuning: threads per workgroup of the glue kernels (0 = heuristic: 64 up to 512 half-wave tasks, else 256) */
int exl3_glue_rotate(const void* resid, const float* ss_part, const void* w, float eps, const void* const* suhs, void* const* xhs,
                     float* const* xsums, int count, int m, int hidden, void* stream);

/* down_proj on a = fp16(silu(g) * u) taken straight from the gate / up launch's deferred slabs (m <= 4): the reduce + output Hadamards +
 * svh + silu*mul (activation.cu) + input Hadamard of exl3_glue_act happen while this GEMV builds its activation fragments. */
/* ---- residual add folded into the consumer GEMV (decode, m <= 4): 5 launches per Llama layer instead of 7 ------------------------------------
 * exl3_gemv_ex_resid = exl3_glue_resid + exl3_gemv_ex_norm in one launch: every workgroup finishes the PRODUCER linear's output (its deferred split-k
 * slabs: out-Hadamard, svh, + resid_in) for the Hadamard blocks of its own k-slice; the workgroups of column block 0 write resid_out (must differ
 * from resid_in) and ss_out [m][k/128].  The RMSNorm scale applied to the activations is the one of resid_in (ss_prev); because the quantized
 * linear is linear, whoever finishes this launch's slabs multiplies by r_new / r_prev (exl3_glue_qkv_rs, exl3_gemv_ex_act_rs) -- same value as
 * rms_norm_res_in (norm.cu:193-299) + exl3_mgemm up to the rounding point of the normalised activation (fp16 at scale r_prev instead of r_new).
 * Replaces the same reference graph nodes as exl3_glue_resid + exl3_gemv_ex_norm (libtorch/mlp.cpp:14-91, libtorch/attention.cpp:246-330). */
int exl3_gemv_ex_resid(const void* resid_in, const void* norm_w, const float* ss_prev, float eps, const float* prod_slabs, int prod_S,
                       const void* prod_svh, void* resid_out, float* ss_out, const void* const* Bs, const void* const* suhs, const int* ns,
                       int count, int m, int k, int K, int cb, int cpw, int force_split, float** slabs_out, int* S_out, void* stream);
/* ---- "fx" decode pipeline (m <= 4, one rank): the residual stream is a fixed-point accumulator R int64 [m][hidden] (value * 2^32) -------------------
 * exl3_fx_init: R = x (fp16) and ss [m][hidden/128] = block sums of squares of x.   exl3_fx_finish: x = fp16(R / 2^32) (x may be NULL), ss of it.
 * exl3_gemv_ex(..., EXL3_GEMV_OUT_ATOMIC, Cs = {R}): o_proj / down_proj add their rows into R (see the flag).
 * exl3_gemv_ex_fx: exl3_gemv_ex_norm reading R.  The RMSNorm scale it applies is the PREVIOUS residual's (ss_prev, complete); the sums of squares of R
 * itself go to ss_out (!= ss_prev), and the consumer of this launch's deferred slabs multiplies by r_new / r_prev (exl3_glue_qkv_rs, exl3_glue_act_rs,
 * exl3_gemv_ex_act_rs) -- the protocol of exl3_gemv_ex_resid, same rounding point.  5-6 launches per Llama layer instead of 7-8: neither the split-k
 * reduce nor rms_norm_res_in's residual add (norm.cu:193-218) is a launch any more. */
int exl3_fx_init(const void* x, void* R, float* ss, int m, int hidden, void* stream);
int exl3_fx_finish(const void* R, void* x, float* ss, int m, int hidden, void* stream);
/* The same two step-level boundaries with one kernel launch less each: exl3_fx_init_prep = exl3_fx_init + exl3_qkv_prep (independent set-up work of a decode
 * step, one launch); exl3_fx_finish_rotate = exl3_fx_finish + exl3_glue_rotate for ONE consumer (the lm_head): x_out / ss_out optional, xh = rotated
 * normalised row [m][hidden] fp16, xsum its block sums [m][hidden/128] (optional).  Values identical to the two-launch forms. */
int exl3_fx_init_prep(const void* x, void* R, float* ss, int m, int hidden, const float* inv_freq, const int32_t* positions, float attn_factor,
                      int head_dim, const int32_t* block_table, int blocks_per_seq, int page_size, float* sin_out, float* cos_out,
                      int64_t* slots, void* stream);
int exl3_fx_finish_rotate(const void* R, void* x_out, float* ss_out, const void* norm_w, float eps, const void* suh, void* xh, float* xsum,
                          int m, int hidden, void* stream);
int exl3_gemv_ex_fx(const void* R, const void* norm_w, const float* ss_prev, float* ss_out, float eps, const void* const* Bs,
                    const void* const* suhs, const int* ns, int count, int m, int k, int K, int cb, int force_split,
                    float** slabs_out, int* S_out, void* stream);
/* ... with the gate / up rows ADDED into fixed-point accumulators accs[i] (int64 [m][n_i], zero on entry) instead of slabs, and the down_proj that
 * forms silu(g) * u from them (row-scale correction from ss_prev / ss_new) while it builds its activation quads: the silu_mul node of
 * libtorch/mlp.cpp:14-91 without a launch and without a slab reduction.  exl3_fx_zero_next(ptr, bytes): the NEXT GEMV entry-point call made by THIS
 * host thread on THIS device clears that buffer as a side job (the o_proj launch zeroes the accumulators of the gate|up launch behind it: no memset
 * node).  One-shot: the request never outlives that call -- if it fails an argument check, or is not a plain (non-table) generation-4 launch, it
 * returns EXL3_ERR_ARG ("cannot clear the buffer") and the request is dropped; (nullptr, 0) cancels.
 * Non-finite values: a NaN / Inf / |v| >= 2^20 contribution REPLACES its accumulator with a poison value and every reader of an accumulator
 * (the GEMV_IN_FX / GEMV_IN_ACTFX launches, exl3_fx_finish*, the fx router) turns a poisoned accumulator into NaN, so the failure reaches the logits as
 * it does through the reference's fp16 residual (norm.cu:193-218). */
int exl3_gemv_ex_fx_atomic(const void* R, const void* norm_w, const float* ss_prev, float* ss_out, float eps, const void* const* Bs,
                           void* const* accs, const void* const* suhs, const void* const* svhs, const int* ns, int count, int m, int k, int K,
                           int cb, int force_split, int* S_out, void* stream);
int exl3_gemv_ex_actfx(const void* g_acc, const void* u_acc, const float* ss_prev, const float* ss_new, int hidden, float eps,
                       const void* B, void* C, const void* suh, const void* svh, const void* bias, int m, int k, int n, int K, int cb,
                       int flags, int force_split, float** slab_out, int* S_out, void* stream);
/* Tensor-parallel boundary of the fx pipeline.  exl3_fx_add: R += the finished rows of a row-sharded linear that did not add them itself -- dense fp32
 * rows (after a collective-library all-reduce of the ranks' partials), or deferred slabs + svh.  exl3_ar_reduce_fx (with the IPC all-reduce context,
 * below): the same with the exchange inside -- push this rank's partial, sum the W partials in rank order, R += sum -- one launch per boundary,
 * accumulators bit-identical on every rank (model/model_tp_backend.py:119-126 + the residual add). */
int exl3_fx_add(void* R, const float* y, const float* slabs, int S, const void* svh, int m, int hidden, void* stream);
int exl3_ar_reduce_fx(void* ctx, const float* y, const float* slabs, int S, const void* svh, void* R, int m, int hidden, void* stream);
int exl3_fx_zero_next(void* ptr, int64_t bytes);
/* exl3_gemv_ex (raw input, deferred output, m <= 4) in the wave-per-column-block layout used by the three launches above and below: `cpw` (1..16)
 * column blocks of ONE matrix per workgroup, each wave streams the whole k-slice of its column block and writes its own slab.  The fused
 * prologues then re-read the producer's slabs once per cpw column blocks instead of once per column block. */
int exl3_gemv_ex_wpc(const void* A, const void* const* Bs, const void* const* suhs, const int* ns, int count, int m, int k, int K, int cb,
                     int cpw, int force_split, float** slabs_out, int* S_out, void* stream);
int exl3_gemv_ex_act_rs(const float* g_slabs, const float* u_slabs, int act_S, const void* svh_g, const void* svh_u,
                        const float* ss_prev, const float* ss_new, int hidden, float eps,
                        const void* B, void* C, const void* suh, const void* svh, const void* bias, int m, int k, int n, int K, int cb,
                        int c_fp32, int flags, int cpw /* 0: classic layout */, int force_split, float** slab_out, int* S_out, void* stream);
int exl3_glue_qkv_rs(const float* sq, const float* sk, const float* sv, int S, const void* svh_q, const void* svh_k, const void* svh_v,
                     void* q_out, void* k_out, void* v_out, const float* inv_freq, const int32_t* positions,
                     void* k_cache, void* k_scales, void* v_cache, void* v_scales, const int32_t* block_table, int blocks_per_seq,
                     int page_size, int k_bits, int v_bits, int m, int heads_q, int heads_kv, int head_dim, int rope_mode,
                     float attn_factor, const float* ss_prev, const float* ss_new, int hidden, float eps, void* stream);
int exl3_gemv_ex_act(const float* g_slabs, const float* u_slabs, int act_S, const void* svh_g, const void* svh_u,
                     const void* B, void* C, const void* suh, const void* svh, const void* bias, int m, int k, int n, int K, int cb,
                     int c_fp32, int flags, int force_split, float** slab_out, int* S_out, void* stream);
/* Per-step tables for exl3_glue_qkv_tab (every layer of a decode step shares positions and block table): sin / cos [m][64] fp32 of
 * position x inv_freq (x attn_factor; head_dim / 2 entries per row used) and slots[r] = the physical row of token r in the paged cache.
 * exl3_glue_qkv_tab = exl3_glue_qkv_rs reading them instead of building its own sin / cos table and walking positions -> block_table in every
 * layer (rope.cu:60-120, q_cache_kernels.cuh:300-318 do that per launch); rope_sin == rope_cos == slots == NULL: identical to exl3_glue_qkv_rs. */
int exl3_qkv_prep(const float* inv_freq, const int32_t* positions, float attn_factor, int m, int head_dim, const int32_t* block_table,
                  int blocks_per_seq, int page_size, float* sin_out, float* cos_out, int64_t* slots, void* stream);
int exl3_glue_qkv_tab(const float* sq, const float* sk, const float* sv, int S, const void* svh_q, const void* svh_k, const void* svh_v,
                      void* q_out, void* k_out, void* v_out, const float* inv_freq, const int32_t* positions,
                      void* k_cache, void* k_scales, void* v_cache, void* v_scales, const int32_t* block_table, int blocks_per_seq,
                      int page_size, int k_bits, int v_bits, int m, int heads_q, int heads_kv, int head_dim, int rope_mode,
                      float attn_factor, const float* ss_prev, const float* ss_new, int hidden, float eps,
                      const float* rope_sin, const float* rope_cos, const int64_t* slots, void* stream);
/* ---- MoE block tail in one launch (one rank holds all experts): exl3_mgemm_indexed_act_deferred = the weighted down launch of
 * exl3_mgemm_indexed_act leaving raw split-k slabs [slot][n/128][S][m][128] (no svh, no routing weight, no slot sum); exl3_glue_resid_moe finishes
 * them per (token, 128-block) -- out-Hadamard, x (1/sqrt(128) * w_slot) x svh[expert], slots summed in order from zero (the arithmetic of the
 * reference's weighted exl3_mgemm, quant/exl3_gemm_kernel.cuh:241-290) -- and adds the result to the fp16 residual stream, leaving the per-block
 * sums of squares for the next norm.  Replaces three launches (split-k reduce, slot sum, residual add). */
int exl3_mgemm_indexed_act_deferred(const void* G, const void* U, const void* tbl_B, const void* tbl_suh, const int64_t* indices, int bszm,
                                    int m, int k, int n, int K, int cb, float** slab_out, int* S_out, void* stream);
int exl3_glue_resid_moe(const float* slabs, int S, const void* tbl_svh, const int64_t* indices, const void* weights, int top_k, void* resid,
                        float* ss_part, int tokens, int hidden, void* stream);

/* ---- batches above 4 rows: exl3_glue_resid + exl3_glue_rotate in one launch (8 launches per Llama layer instead of 10) --------------------------
 * resid += y (pending slabs + svh, or a dense fp32 tensor); ss_new [m][hidden/128] = block sums of squares of the new residual;
 * xh_i = had128(fp16(resid_new * w * r_prev) * suh_i) for up to 3 consumers, r_prev = rsqrt(mean(ss_prev) + eps) of the PREVIOUS residual
 * (ss_prev != ss_new).  The consumers' outputs are finished with r_new / r_prev by exl3_glue_qkv_rs / exl3_glue_act_rs (same reasoning and
 * rounding point as exl3_gemv_ex_resid above).  Replaces rms_norm_res_in (norm.cu:193-299) + the input transform of exl3_gemm (quant/exl3_gemm.cu:139-145). */
int exl3_glue_resid_rotate(const float* y_slabs, int y_S, const float* y_dense, const void* svh, const void* bias, void* resid,
                           const float* ss_prev, float* ss_new, const void* w, float eps, const void* const* suhs, void* const* xhs,
                           float* const* xsums, int count, int m, int hidden, void* stream);
int exl3_glue_act_rs(const float* sg, const float* su, int S, const void* svh_g, const void* svh_u, const void* suh_d,
                     void* xh_d, float* xsum_d, void* a_out, int m, int inter, const float* ss_prev, const float* ss_new, int hidden,
                     float eps, void* stream);

/* Hand-written NT MFMA GEMM of the prefill route (exl3_gemm_nt.hip; reference: hgemm.cu:19-78 behind modules/quant/exl3.py:161-218):
 * c[m][n] = a[m][k] @ bt[n][k]^T, fp16 in, fp32 accumulate, fp16 out; a / bt / c row-major with leading dimensions lda / ldb / ldc (elements).
 * epi 0: store; 1: c = fp16(c + y) (the residual add of o_proj / down_proj); 2: bt stacks, per 256 rows, 128 gate rows then the 128 up rows of the
 * same outputs and c[m][n/2] = fp16(silu(fp16 g) * fp16 u) (activation.cu silu_mul fused into the gate|up GEMM).  k % 64 == 0, n % 256 == 0,
 * 16-byte aligned operands; other shapes return EXL3_ERR_ARG (use exl3_hgemm_nt*, the hipBLASLt route). */
int exl3_gemm_nt_mfma(const void* a, int64_t lda, const void* bt, int64_t ldb, void* c, int64_t ldc, int m, int k, int n, int epi, void* stream);

/* Generation 2 of the same contraction (exl3_gemm_nt2.hip; same reference, same contract and epilogues): one wave per SIMD with a 128 x 128 wave tile, the K-loop one
 * hand-allocated assembly statement (v_mfma_f32_32x32x16_f16 on 256 accumulator registers, four LDS buffers filled three K-tiles ahead, one barrier per K-tile).
 * k % 128 == 0, n % 256 == 0, 16-byte aligned operands; other shapes return EXL3_ERR_ARG (generation 1 or exl3_hgemm_nt* take them). */
int exl3_gemm_nt2_mfma(const void* a, int64_t lda, const void* bt, int64_t ldb, void* c, int64_t ldc, int m, int k, int n, int epi, void* stream);

/* The same kernel as a GROUPED launch: `count` problems c_e[m_e][n] = a_e[m_e][k] @ bt_e[n][k]^T of one k and n in ONE grid -- the experts of a MoE block over the
 * assignments sorted by expert (the reference runs one GEMM per expert, sized on the host from expert_count.tolist(): quant/exl3_moe.cu:99-301,
 * modules/block_sparse_mlp.py:1169-1330).  a / c hold the problems' rows back to back: problem e = rows [rows[e], rows[e + 1]); `rows` is DEVICE memory (int32
 * [count + 1], e.g. the cumulative sum of the router's bincount: the sizes never visit the host); bt_e = bt + e * bt_stride elements; max_rows >= rows[count] sizes the
 * grid.  epi 0 store, 1 c += (fp16), 2 silu(gate) * up (bt_e as exl3_reconstruct_had_multi_t_interleaved writes it; c has n / 2 columns), 3 fp32 output (c float, ldc
 * in floats: the reference's fp32 expert outputs in front of its index_add_).  k % 64 == 0, n % 256 == 0, 16-byte aligned operands. */
int exl3_gemm_nt2_grouped(const void* a, int64_t lda, const void* bt, int64_t ldb, int64_t bt_stride, void* c, int64_t ldc, const int* rows, int count, int max_rows,
                          int k, int n, int epi, void* stream);

/* y[rows][cols] = silu(g) * u (activation.cu) where g and u are fp16 column ranges of wider matrices (row strides ld_g, ld_u). */
int exl3_silu_mul_2d(const void* g, const void* u, void* y, int64_t rows, int64_t cols, int64_t ld_g, int64_t ld_u, void* stream);

/* Decode attention straight from the quantized paged cache (one new token per sequence, GQA, head_dim 128):
 *   out[b][h] = softmax(q[b][h] . K[b][:len]^T * scale) @ V[b][:len]     with K, V the dequantized cache (cache/q_cache_kernels.cuh semantics)
 * reference: libtorch/attention.cpp:246-504 (dequant_cache_paged + fp16 attention).  K / V are never materialised: scores and the output are
 * computed in the cache's rotated (H32) domain.  cache_seqlens[b] = length INCLUDING the new token (append it first).  max_len bounds the
 * lengths (host value: no sync).  workspace: >= bsz * heads_q * ceil(max_len / 32) * 132 floats for the flash-decoding partials (may be NULL
 * when max_len <= 32). */
int exl3_attn_decode_qcache(const void* q, void* out, const void* k_cache, const void* k_scales, const void* v_cache, const void* v_scales,
                            const int32_t* block_table, const int32_t* cache_seqlens, int bsz, int blocks_per_seq, int page_size,
                            int k_bits, int v_bits, int heads_q, int heads_kv, int head_dim, int max_len, float scale,
                            float* workspace, int64_t workspace_floats, void* stream);
/* exl3_attn_decode_qcache with learned attention sinks (gpt-oss style): sinks[heads_q] fp32, one logit per query head in the units of the scaled scores,
 * joining the softmax denominator only (modules/attention_fn/triton_paged.py:1030-1050; the combine kernel of libtorch/attention.cpp:463-480).  The workspace
 * is required (the merge kernel finishes every head, also with one split). */
int exl3_attn_decode_qcache_sinks(const void* q, void* out, const void* k_cache, const void* k_scales, const void* v_cache, const void* v_scales,
                                  const int32_t* block_table, const int32_t* cache_seqlens, int bsz, int blocks_per_seq, int page_size,
                                  int k_bits, int v_bits, int heads_q, int heads_kv, int head_dim, int max_len, float scale,
                                  float* workspace, int64_t workspace_floats, const float* sinks, void* stream);

/* The context-split half of exl3_attn_decode_qcache only (head_dim 128): the partial records {m, l, -, -, o[128]} of every (sequence, 128-value kv block,
 * query index, split) stay in `workspace` ([bsz][blocks][gq][*nsplit_out][132] fp32, written also with a single split; at most 32 splits) and the
 * consumer of the attention output merges them: exl3_gemv_ex_attm is o_proj with that merge as the preparation task of each (row, head) -- the
 * arithmetic of the merge kernel operation for operation, so attention + o_proj give the same bits with one launch less
 * (libtorch/attention.cpp:246-504: attention, then the o_proj linear).  flags: EXL3_GEMV_OUT_DEFERRED / EXL3_GEMV_OUT_ATOMIC / 0. */
int exl3_attn_decode_qcache_split(const void* q, const void* k_cache, const void* k_scales, const void* v_cache, const void* v_scales,
                                  const int32_t* block_table, const int32_t* cache_seqlens, int bsz, int blocks_per_seq, int page_size,
                                  int k_bits, int v_bits, int heads_q, int heads_kv, int head_dim, int max_len, float scale,
                                  float* workspace, int64_t workspace_floats, int* nsplit_out, void* stream);

/* exl3_glue_qkv_tab + exl3_attn_decode_qcache_split as ONE launch (round 4; libtorch/attention.cpp:386-440 -- rope, cache append, split kernel -- as one node):
 * every workgroup of the context-split kernel finishes the query heads it needs from the q|k|v launch's deferred slabs (sq / sk / sv, S slices; svh_*;
 * row-scale correction ss_prev / ss_new / hidden / eps as in exl3_glue_qkv_rs; RoPE from exl3_qkv_prep's tables), and the workgroup whose split holds the new
 * token (index cache_seqlens[b] - 1, physical row slots[b]) appends that token's K / V to the 4-bit cache.  Applies where the matrix-pipe split kernel applies
 * (head_dim 128, 4-bit K and V, length bound >= two 64-token splits); otherwise the two launches run (same results: shared device functions).
 * q_out [bsz][heads_q][128] fp16 is required (scratch of the two-launch form; the fused form's split 0 writes the finished queries there).
 * *fused_out (optional): 1 = one launch, 0 = two.  *nsplit_out: splits of the partial records in `workspace` (consumer: exl3_gemv_ex_attm). */
int exl3_attn_decode_qcache_split_qkv(const float* sq, const float* sk, const float* sv, int S, const void* svh_q, const void* svh_k, const void* svh_v,
                                      void* q_out, const float* inv_freq, const int32_t* positions, float attn_factor, int rope_mode,
                                      const float* ss_prev, const float* ss_new, int hidden, float eps,
                                      const float* rope_sin, const float* rope_cos, const int64_t* slots,
                                      void* k_cache, void* k_scales, void* v_cache, void* v_scales,
                                      const int32_t* block_table, const int32_t* cache_seqlens, int bsz, int blocks_per_seq, int page_size,
                                      int k_bits, int v_bits, int heads_q, int heads_kv, int head_dim, int max_len, float scale,
                                      float* workspace, int64_t workspace_floats, int* nsplit_out, int* fused_out, void* stream);
int exl3_gemv_ex_attm(const float* part, int nsplit, int gq, int blocks, int head_dim, const void* B, void* C, const void* suh, const void* svh, const void* bias,
                      int m, int k, int n, int K, int cb, int c_fp32, int flags, int force_split, float** slab_out, int* S_out, void* stream);

/* o_proj fed straight by the q|k|v launch's deferred slabs -- the decode step WITHOUT the attention core, where o_proj's input is the finished q:
 * exl3_glue_qkv_tab's work (split-k reduce, output Hadamard, row-scale correction, svh, RoPE from the per-step tables of exl3_qkv_prep, the 4-bit append of
 * K / V at slots[row]) runs inside this launch; bit-identical to exl3_glue_qkv_tab + exl3_gemv_ex, one launch less per layer.  head_dim 64 | 128, 4-bit K
 * and V, m <= 4; q_out (optional, [m][k] fp16) receives the finished queries.  Other arguments as exl3_gemv_ex.
 * reference nodes replaced: libtorch/attention.cpp:283-400 (q / k / v epilogues, rope, cache append) + :497-508 (o_proj). */
int exl3_gemv_ex_qkvm(const float* sq, const float* sk, const float* sv, int S_qkv, const void* svh_q, const void* svh_k, const void* svh_v,
                      const float* rope_sin, const float* rope_cos, const int64_t* slots, const float* ss_prev, const float* ss_new, int hidden, float eps,
                      int rope_mode, int head_dim, int heads_kv, void* q_out, void* k_cache, void* k_scales, void* v_cache, void* v_scales,
                      const void* B, void* C, const void* suh, const void* svh, const void* bias, int m, int k, int n, int K, int cb, int c_fp32,
                      int flags, int force_split, float** slab_out, int* S_out, void* stream);
/* Prefill (multi-token) causal attention over paged fp16 K/V -- the attention step of the reference's prefill path: cache/quant.py:83-117
 * dequantizes the pages (exl3_dequant_cache_paged), then flash_attn_with_kvcache(q, k_pages, v_pages, block_table, cache_seqlens, causal) attends.
 * q / out fp16 [bsz][q_len][heads_q][head_dim] (head_dim 128 or 64); k_pages / v_pages fp16 [pages][page_size][heads_kv][head_dim];
 * cache_seqlens[b] = tokens of sequence b in the cache INCLUDING the q_len new ones (appended before the call);
 * query i of the chunk attends to keys 0 .. cache_seqlens[b] - q_len + i.  fp32 softmax / accumulation, fp16 probabilities and output. */
int exl3_attn_prefill_paged(const void* q, void* out, const void* k_pages, const void* v_pages, const int32_t* block_table,
                            const int32_t* cache_seqlens, int bsz, int q_len, int heads_q, int heads_kv, int head_dim,
                            int blocks_per_seq, int page_size, float scale, void* stream);
/* ... with q as a column range of a wider row-major matrix (ldq halves per token, a multiple of 8): the fused q|k|v prefill GEMM output in place */
int exl3_attn_prefill_paged_strided(const void* q, int64_t ldq, void* out, const void* k_pages, const void* v_pages, const int32_t* block_table,
                                    const int32_t* cache_seqlens, int bsz, int q_len, int heads_q, int heads_kv, int head_dim,
                                    int blocks_per_seq, int page_size, float scale, void* stream);

/* Diagnostics only: copy [byte_offset, byte_offset + nbytes) of the per-device split-k workspace to dst (tools/gemv_timeline.py). */
int exl3_debug_copy_workspace(void* dst, int64_t byte_offset, int64_t nbytes, void* stream);

/* hgemm(a, b, c)      hgemm.cu:19-102:  c[m][n] (row stride ldc elements, fp16 or fp32) = a[m][k] @ b[k][n], fp32 accumulate.
 * m, k, n arbitrary multiples of 16/32/16. */
int exl3_hgemm(const void* a, const void* b, void* c, int m, int k, int n, int64_t ldc, int c_fp32, void* stream);
/* c (fp16, in place) = fp16(a @ b + c): hgemm with the residual add of the o_proj / down_proj boundary (norm.cu:193-218, add.cu) in the
 * GEMM epilogue; same single rounding as fp32 output + add. */
int exl3_hgemm_acc(const void* a, const void* b, void* c, int m, int k, int n, int64_t ldc, void* stream);
/* c[m][n] = a[m][k] @ bt[n][k]^T: bt is B^T row-major with row stride ldb >= k (both operands K-major, the layout the library's MFMA kernels
 * run 15-28 % faster on MI355X); accumulate != 0: c (fp16) += product. */
int exl3_hgemm_nt(const void* a, const void* bt, void* c, int m, int k, int n, int64_t ldb, int64_t ldc, int c_fp32, int accumulate, void* stream);
/* the same with a row stride for a (lda >= k): a is a column range of a wider row-major matrix */
int exl3_hgemm_nt_lda(const void* a, int64_t lda, const void* bt, void* c, int m, int k, int n, int64_t ldb, int64_t ldc, int c_fp32, int accumulate, void* stream);

/* ---- RMSNorm     norm.cuh:7-39, norm.cu:155-299 --------------------------------------------------- */
/* mode 0: y = norm(x)*w ; 1: y += norm(x)*w (add_residual) ; 2: r += x; y = norm(r)*w (rms_norm_res_in).
 * x_fp32 / y_fp32 / r_fp32 select fp16 or fp32 buffers; w fp16 (w_bf16 = 0) or bf16 (1) or NULL. dim % 4 == 0. */
int exl3_rms_norm(const void* x, const void* w, void* y, void* r, float eps, float constant_bias, float constant_scale,
                  int rows, int dim, int x_fp32, int y_fp32, int r_fp32, int w_bf16, int mode, void* stream);

/* ---- RoPE        rope.cuh:51-72, rope.cu:16-296 -------------------------------------------------- */
/* q [bsz][seq][heads_q][head_dim], k [bsz][seq][heads_k][head_dim] fp16 (k/out_k may be NULL); inv_freq fp32 [head_dim/2]
 * (or [bsz*seq][head_dim/2] when inv_freq_per_token).  Position of (b,t) = position + t | positions[b] + t | position_ids[b][t]
 * (int32 device arrays or NULL).  rope_mode 1 = GPTJ (interleaved), 2 = NEOX (half split).  q_norm/k_norm: optional fp16 [head_dim]
 * per-head RMSNorm weights applied before the rotation (eps, constant_bias). */
int exl3_rope(const void* q, void* out_q, const void* k, void* out_k, const float* inv_freq,
              int bsz, int seq_len, int heads_q, int heads_k, int head_dim,
              uint32_t position, const int32_t* positions, const int32_t* position_ids,
              int rope_mode, float attn_factor, const void* q_norm, const void* k_norm, float norm_eps,
              float norm_constant_bias, void* stream);

/* The rest of rope.cuh:51-72's argument list (kernel rope.cu:16-305, host checks rope.cu:345-470):
 *   partial_head_dim   rotated width = 2 * (columns of inv_freq) <= head_dim; elements outside the rotated sub-ranges pass through
 *   rotate_dims 1..4   sub-range r covers [rotate_offset + partial_head_dim * r, + partial_head_dim) and takes position_ids[b][t][r] when
 *                      position_ids_stride == rotate_dims (3-D position ids), else the token's one position
 *   inv_freq_table     inv_freq holds ANGLES indexed [batch * inv_freq_stride + pos * partial_head_dim / 2 + pair] (stride 0: one table for all)
 *   rope_mode 3        NANOCHAT: NEOX pairs with the opposite sign of sin
 *   norm_bf16          q_norm / k_norm are bfloat16 (v * rmf * (float(w) + bias), one rounding) instead of fp16 (reference's fp16 arithmetic)
 *   post_rope_norm     unweighted RMSNorm of the whole head after the rotation
 *   l4_beta > 0        query heads only, whole head: *= 1 + l4_beta * ln(1 + pos / l4_orig) (integer division), after the rotation
 *   q_head_stride / k_head_stride   halves between consecutive heads (>= head_dim; token stride = heads * head stride as in the reference) */
int exl3_rope_ex(const void* q, void* out_q, const void* k, void* out_k, const float* inv_freq,
                 int bsz, int seq_len, int heads_q, int heads_k, int head_dim, int q_head_stride, int k_head_stride, int partial_head_dim,
                 uint32_t position, const int32_t* positions, const int32_t* position_ids, int position_ids_stride,
                 int rope_mode, float attn_factor, const void* q_norm, const void* k_norm, int norm_bf16, float norm_eps, float norm_constant_bias,
                 int inv_freq_table, int inv_freq_stride, float l4_beta, int l4_orig, int post_rope_norm, int rotate_dims, int rotate_offset,
                 void* stream);

/* ---- KV-cache quantization   cache/q_cache.cuh:48-62, q_cache_kernels.cuh:61-236 ------------------ */
/* quant_cache_cont(in, out, out_scales): in fp16 [tokens][dim], out u32 [tokens][dim/32*bits], scales fp16 [tokens][dim/32] */
int exl3_quant_cache_cont(const void* in, void* out, void* out_scales, int64_t tokens, int dim, int bits, void* stream);
int exl3_dequant_cache_cont(const void* in, const void* in_scales, void* out, int64_t tokens, int dim, int bits, void* stream);
/* quant_cache_paged: append seq_len new tokens per sequence to the paged quantized cache.
 * k_in/v_in fp16 [bsz][seq_len][dim] (in_contiguous) ; caches u32 [pages][page_size][dim/32*bits], scales fp16 [pages][page_size][dim/32];
 * cache_seqlens int32 [bsz] (tokens already in cache), block_table int32 [bsz][blocks_per_seq]. */
int exl3_quant_cache_paged(const void* k_in, void* k_out, void* k_scales, const void* v_in, void* v_out, void* v_scales,
                           const int32_t* cache_seqlens, const int32_t* block_table, int bsz, int blocks_per_seq,
                           int page_size, int seq_len, int dim, int k_bits, int v_bits, void* stream);
/* the same append with k_in / v_in as column ranges of a wider row-major matrix (ld_k, ld_v = halves per token): consumes the prefill
 * route's fused q|k|v GEMM output without a split copy */
int exl3_quant_cache_paged_strided(const void* k_in, void* k_out, void* k_scales, const void* v_in, void* v_out, void* v_scales,
                                   const int32_t* cache_seqlens, const int32_t* block_table, int bsz, int blocks_per_seq,
                                   int page_size, int seq_len, int dim, int k_bits, int v_bits, int64_t ld_k, int64_t ld_v, void* stream);
/* in-place NEOX rope, head_dim 128, no head norm, on q / k column ranges of a wider matrix (rope.cu semantics otherwise) */
int exl3_rope_strided(void* q, void* k, const float* inv_freq, int bsz, int seq_len, int heads_q, int heads_k, int64_t ld_q, int64_t ld_k,
                      uint32_t position, const int32_t* positions, const int32_t* position_ids, float attn_factor, void* stream);
/* dequant_cache_paged: expand every page referenced by block_table up to cache_seqlens[b] (+ nothing beyond) into fp16 pages
 * k_out/v_out fp16 [pages][page_size][dim]. */
int exl3_dequant_cache_paged(const void* k_in, const void* k_scales, void* k_out, const void* v_in, const void* v_scales, void* v_out,
                             const int32_t* cache_seqlens, const int32_t* block_table, int bsz, int blocks_per_seq,
                             int page_size, int dim, int k_bits, int v_bits, void* stream);

/* The cache ops with the reference's remaining arguments (cache/q_cache.cuh:6-92; cache/quant.py:83-117 passes them on every call):
 * compand_a > 0 (in (0, 1)): cubic level compander of cache/lmq.cuh instead of the midpoint grid (0 = off);
 * sliding_window > 0: dequant_cache_paged leaves the rows before the window untouched, by the reference's own rule (whole 256-chunk thread
 * blocks that end at or before cache_seqlens[b] - sliding_window are skipped; a chunk = 4 groups of one token);
 * compact_out, bonus_len: dequant_cache_paged_window -- page p of sequence b goes to row (b * blocks_per_seq + p) * page_size of the scratch
 * and rows up to cache_seqlens[b] + bonus_len are expanded (sliding_window is ignored by the reference there: pass 0). */
int exl3_quant_cache_cont_ex(const void* in, void* out, void* out_scales, int64_t tokens, int dim, int bits, float compand_a, void* stream);
int exl3_dequant_cache_cont_ex(const void* in, const void* in_scales, void* out, int64_t tokens, int dim, int bits, float compand_a, void* stream);
int exl3_quant_cache_paged_ex(const void* k_in, void* k_out, void* k_scales, const void* v_in, void* v_out, void* v_scales,
                              const int32_t* cache_seqlens, const int32_t* block_table, int bsz, int blocks_per_seq,
                              int page_size, int seq_len, int dim, int k_bits, int v_bits, int64_t ld_k, int64_t ld_v, float compand_a,
                              int in_contiguous /* 0: k_in / v_in are a flat fp16 cache read at each token's own physical row */, void* stream);
int exl3_dequant_cache_paged_ex(const void* k_in, const void* k_scales, void* k_out, const void* v_in, const void* v_scales, void* v_out,
                                const int32_t* cache_seqlens, const int32_t* block_table, int bsz, int blocks_per_seq,
                                int page_size, int dim, int k_bits, int v_bits, int sliding_window, float compand_a, int compact_out,
                                int bonus_len, void* stream);

/* ---- elementwise glue used by the decode step (activation.cu silu_mul, add.cu) -------------------- */
/* y = silu(g) * u ; g, u fp16 or fp32 (in_fp32) [rows][dim]; y fp16 */
int exl3_silu_mul(const void* g, const void* u, void* y, int64_t numel, int in_fp32, void* stream);
/* y = fp16(act(g) * u): the other activations of the reference's gated MLP (activation.cu gelu_mul :181-254, relu2_mul :259-330, silu_oai_mul :103-176; kernels
 * activation_kernels.cuh:142-254).  act: 0 SiLU, 1 GELU (tanh form), 2 relu^2, 3 relu, 4 gpt-oss clamped swiglu.  act_limit != 0: up clamped to [-limit, limit], the
 * activated gate to <= limit (act 4: gate and up clamped BEFORE the activation).  g, u fp16 or fp32 (in_fp32), numel % 4 == 0. */
#define EXL3_ACT_SILU 0
#define EXL3_ACT_GELU 1
#define EXL3_ACT_RELU2 2
#define EXL3_ACT_RELU 3
#define EXL3_ACT_SILU_OAI 4
int exl3_act_mul(const void* g, const void* u, void* y, int64_t numel, int in_fp32, int act, float act_limit, void* stream);

/* attention output gates   activation.cuh:124-160, activation.cu:526-660 (mul_sigmoid_, mul_sigmoid_broadcast_, mul_softplus_broadcast_):
 * x fp16 [numel] *= sigmoid(y[i]) (bcast = 0, y [numel]) or *= sigmoid / softplus(y[i / bcast]) (one gate per `bcast` consecutive values = per head).
 * The sigmoid is fp16 arithmetic like the reference's (activation_kernels.cuh:132-139), the softplus gate fp32 with one rounding of the product. */
int exl3_mul_gate(void* x, const void* y, int64_t numel, int bcast, int softplus, void* stream);
/* deinterleave_qg (activation.cu:716-785): qg [heads_total][2][head_dim] fp16 -> q, g [heads_total][head_dim] */
int exl3_deinterleave_qg(const void* qg, void* q, void* g, int64_t heads_total, int head_dim, void* stream);
/* shared-expert merge of the sparse-MoE block (activation.cuh add_sigmoid_gate / add_sigmoid_gate_proj; activation.cu:480-524, 662-714), fp32:
 * z[i] += x[i] * sigmoid(y[i / dim])   and   z[b][:] += x[b][:] * sigmoid(y[b][:] . w)  (y, w fp16; a gate below 1e-8 leaves the row untouched) */
int exl3_add_sigmoid_gate(const float* x, const float* y, float* z, int64_t numel, int dim, void* stream);
int exl3_add_sigmoid_gate_proj(const float* x, const void* y, float* z, const void* w, int rows, int dim, void* stream);
/* fp16 paged cache append (generator/cache.cu:140-240 paged_kv_cache_update): k / v fp16 [bsz][seq_len][heads][dim] -> row cache_seqlens[b] + t of the
 * sequence's pages (k_cache / v_cache fp16 [pages][256][heads][dim]) */
int exl3_paged_kv_cache_update(const void* k, const void* v, void* k_cache, void* v_cache, const int32_t* block_table, const int32_t* cache_seqlens,
                               int bsz, int seq_len, int heads, int dim, int pages_per_seq, void* stream);
/* x (fp16 or fp32) += y (fp16 or fp32) */
int exl3_add(void* x, const void* y, int64_t numel, int x_fp32, int y_fp32, void* stream);
/* softcap(x, y, scale)   softcap.cu:59-100: y = scale * tanh(x / scale) (fp32 math; fp16 or fp32 tensors; y == x allowed); Linear.forward's
 * post-op for soft-capped logits (modules/linear.py:598-599) */
int exl3_softcap(const void* x, void* y, int64_t numel, float scale, int is_fp32, void* stream);

/* ---- MoE block fusions (block_sparse_mlp.py:1099-1478 at decode): fewer launches around the indexed exl3_mgemm ------------------------------
 * exl3_routing_std_slots: routing_std (routing.cu:955-1010) that also writes the [gate slots | up slots] index list of one indexed launch over the
 * concatenated gate | up pointer tables.  exl3_mgemm_indexed_act: the down launch whose per-slot input is fp16(silu(G_j) * U_j) (silu_mul folded in). */
int exl3_routing_std_slots(const void* hidden, const void* gate, const void* bias, void* scores, int64_t* topk_indices, void* topk_weights,
                           int64_t* gu_slots, int bsz, int hidden_size, int num_experts, int K, void* stream);
/* ... with per_expert_scale (bf16 [num_experts], routing.cu:955-1010 argument): weight_k *= scale[expert_k] after the softmax */
int exl3_routing_std_scaled(const void* hidden, const void* gate, const void* bias, const void* per_expert_scale, void* scores,
                            int64_t* topk_indices, void* topk_weights, int64_t* gu_slots, int bsz, int hidden_size, int num_experts, int K,
                            void* stream);
/* exl3_mgemm with per-matrix output widths -- replaces the size_n_list / c_ptrs form of quant/exl3_gemm.cuh:54-55,76-77 (launcher
 * exl3_gemm.cu:433-447, kernel exl3_gemm_kernel.cuh:176-182; caller libtorch/dsv4_attn.cpp:88-98).  Matrix i is size_n_list[i] columns wide (device
 * int32 [num matrices], multiples of 128, <= n_max) and writes its [m][size_n_list[i]] output (fp16 / fp32 by c_fp32) at c_ptrs[i] (device table of
 * device addresses); slot j of bszm runs matrix indices ? indices[j] : j on A[j] (or the one shared A).  No routing weights / expert range / token
 * groups (the reference rejects those combinations too).  1..16 rows. */
int exl3_mgemm_indexed_nlist(const void* A, int bszm_in, const void* tbl_B, const void* tbl_suh, const void* tbl_svh, const int64_t* indices,
                             int bszm, const int32_t* size_n_list, const void* c_ptrs, int m, int k, int n_max, int K, int cb, int c_fp32, void* stream);

/* ... on the RMSNorm of the residual stream formed inside the launch: xn = fp16(resid * norm_w * rsqrt(mean(resid^2) + eps)), mean square from
 * ss_part [bsz][hidden/128] (exl3_glue_resid); xn_out receives xn for the expert launches.  Replaces the rms_norm launch + routing of
 * modules/block_sparse_mlp.py:1099-1130. */
int exl3_routing_std_norm(const void* resid, const void* norm_w, const float* ss_part, float eps, void* xn_out, const void* gate, const void* bias,
                          void* scores, int64_t* topk_indices, void* topk_weights, int64_t* gu_slots, int bsz, int hidden_size,
                          int num_experts, int K, void* stream);
int exl3_mgemm_indexed_act(const void* G, const void* U, const void* tbl_B, const void* tbl_suh, const void* tbl_svh,
                           const int64_t* indices, const void* weights, int bszm, void* C, int m, int k, int n, int K, int cb, int c_fp32,
                           int min_index, int max_index, int num_tokens, void* stream);

/* ---- MoE block of the fx decode pipeline: 3 launches (modules/block_sparse_mlp.py:1099-1130 runs norm, router, the gate / up / down exl3_mgemm launches,
 * silu_mul and the residual add as separate ops; quant/exl3_gemm_kernel.cuh:88-292 is the indexed kernel).
 * exl3_routing_std_fx: exl3_routing_std_norm whose input is the residual stream in 64-bit fixed point (int64 [bsz][hidden], value * 2^32, the accumulator
 *   of the EXL3_GEMV_OUT_ATOMIC launches); the launch takes the row's exact mean square itself and leaves the block sums of squares in ss_out
 *   [bsz][hidden/128] for the next exl3_gemv_ex_fx / exl3_glue_qkv_rs.
 * exl3_mgemm_indexed_deferred: the indexed gate|up launch (generation-4 GEMV, raw x, bszm_in == 1: one shared row set) leaving raw split-k slabs
 *   [slot][n/128][S][m][128]; indices = the router's [gate slots | up slots] list over the tables [gate_0..gate_E-1, up_0..up_E-1].
 * exl3_mgemm_indexed_act_fx: the indexed down launch: slot j's input fp16(silu(g_j) * u_j) is finished from those slabs (gate slots first, then the up
 *   slots; gu_svh_tbl = that launch's svh table, up svh = entry + up_off) while the activation fragments are built, and its output rows -- output
 *   Hadamard, svh and the routing weight applied per split-k partial -- are ADDED into R (int64 [num_tokens][m][n]) with integer atomics:
 *   no silu_mul, split-k reduce, slot sum or residual launch.  force_split: k-slices (0: the dispatcher balances the chip). */
int exl3_routing_std_fx(const void* resid_fx, const void* norm_w, float* ss_out, float eps, void* xn_out, const void* gate, const void* bias,
                        void* scores, int64_t* topk_indices, void* topk_weights, int64_t* gu_slots, int bsz, int hidden_size,
                        int num_experts, int K, void* stream);
int exl3_mgemm_indexed_deferred(const void* A, int bszm_in, const void* tbl_B, const void* tbl_suh, const int64_t* indices, int bszm,
                                int m, int k, int n, int K, int cb, int force_split, float** slab_out, int* S_out, void* stream);
int exl3_mgemm_indexed_act_fx(const float* gu_slabs, int act_S, const void* gu_svh_tbl, int up_off, const void* tbl_B, const void* tbl_suh,
                              const void* tbl_svh, const int64_t* indices, const void* weights, int bszm, void* R, int m, int k, int n,
                              int K, int cb, int num_tokens, int force_split, void* stream);

/* exl3_moe (quant/exl3_moe.cu:99-301) pieces that keep the op free of host round trips (capturable): the slot list of its indexed launches is built
 * on the device.  slot j < max_slots (>= min(E, T) + T / rows_per_slot): slot_expert[j] = expert with 0 < count <= max_rows, or -1 (skipped by the
 * indexed exl3_mgemm launches); slot_tok [max_slots][rows_per_slot] = token of each row (rows past a chunk repeat its last assignment);
 * rowmap[p] = j * rows_per_slot + r of assignment p, -1 if its expert is not accepted.  exl3_moe_scatter: out[t] += sum over the token's accepted
 * assignments, in ascending p, of weight_sorted[p] * D[rowmap[p]] (fp32; fixed order: bit-reproducible, unlike an atomic scatter). */
int exl3_moe_build_slots(const int64_t* expert_count, const int64_t* token_sorted, int num_experts, int num_assignments, int max_rows,
                         int rows_per_slot, int max_slots, int64_t* slot_expert, int64_t* slot_tok, int32_t* rowmap, void* stream);
int exl3_moe_scatter(const float* D, const int32_t* rowmap, const int64_t* token_sorted, const void* weight_sorted, float* out,
                     int bsz, int num_assignments, int hidden, void* stream);

/* ---- tensor-parallel decode all-reduce: one-shot push over IPC-mapped peer buffers (xGMI), fused with the residual add -------------------------
 * Replaces TPBackendNCCL.all_reduce (model/model_tp_backend.py:119-126) / the native small-message all-reduce (exllamav3_ext/parallel/all_reduce.cu:18-232)
 * for the (tokens x hidden) fp32 partial sums after o_proj / down_proj at decode.  One process per GPU:
 *   exl3_ar_create(world, rank, max_elems, &ctx, handle64)   allocates this rank's fine-grained receive buffer, returns its 64-byte IPC handle;
 *   [exchange the handles between the ranks with any host-side channel]
 *   exl3_ar_open_peer(ctx, r, handle_of_rank_r)              maps rank r's buffer (hipIpcOpenMemHandle);
 *   exl3_ar_reduce(ctx, y, y_out, resid, ss_part, m, hidden, stream)
 *        every rank pushes its partial y [m][hidden] fp32 to all ranks as 8-byte {value, epoch} granules and sums the W partials it received in
 *        rank order (same bits on every rank).  y_out (optional) = the sum; resid (optional, fp16 [m][hidden]) += sum with the rounding of
 *        rms_norm_res_in (norm.cu:193-218); ss_part (optional) = per-128-block sums of squares of the new residual (exl3_glue_resid's output).
 *        Graph-capturable (epochs live in device memory).  Every rank must issue the same sequence of calls.
 *        m * hidden / 128 <= 4096 (the grid has to be co-resident on every rank: larger messages go to the collective library).
 *   exl3_ar_error(ctx, stream)  1 if a bounded spin gave up since the last query (a peer never arrived), else 0; synchronises the stream.
 *        The elements such a spin could not complete are written as NaN (y_out / resid / ss_part), never as partial sums.
 *   exl3_ar_epoch(ctx, &epoch, stream)  number of reductions this rank's buffer has completed; ranks must agree (host-side lockstep check). */
int exl3_ar_create(int world, int rank, int64_t max_elems, void** ctx_out, void* handle_out);
int exl3_ar_open_peer(void* ctx, int peer_rank, const void* handle);
int exl3_ar_destroy(void* ctx);
int exl3_ar_error(void* ctx, void* stream);
int exl3_ar_epoch(void* ctx, uint32_t* epoch_out, void* stream);
int exl3_ar_reduce(void* ctx, const float* y, float* y_out, void* resid, float* ss_part, int m, int hidden, void* stream);
/* exl3_ar_reduce whose local partial is given as the deferred split-k slabs of the row-sharded o_proj / down_proj launch
 * ([hidden/128][S][m][128] fp32 from exl3_gemv_ex*, EXL3_GEMV_OUT_DEFERRED) + that linear's svh instead of a dense tensor (y == NULL): the launch
 * finishes the slabs (sum over S, out-Hadamard, x svh in fp32), pushes, reduces and adds to the residual -- one launch per tensor-parallel sublayer
 * boundary where the reference runs the GEMM epilogue, the all-reduce and the residual add separately (modules/attn.py:915-960, mlp.py:833-891). */
int exl3_ar_reduce_slabs(void* ctx, const float* y, const float* slabs, int S, const void* svh, float* y_out, void* resid, float* ss_part,
                         int m, int hidden, void* stream);

/* ---- the persistent decode step (generation 5, exl3_pstep.hip) ---------------------------------------------------------------------------
 * ONE launch for every quantized linear of a batch-1 decode step of a Llama-type model (all layers + lm_head) and the glue between them
 * (RMSNorm, q|k|v epilogue with RoPE + 4-bit K / V append, silu * mul, residual adds).  Replaces, per layer, the graphs of
 * exllamav3_ext/libtorch/attention.cpp:246-330 (attention core excluded: o_proj consumes q) and libtorch/mlp.cpp:14-91, whose linears each keep
 * input Hadamard -> stream -> output Hadamard inside one cooperative launch (quant/exl3_gemv_kernel.cuh:138-402); here the whole chain is one
 * launch of one 16-wave workgroup per CU; what one CU writes for another travels as tagged 16-byte granules (data and flag in one store: no arrival
 * counter on the critical path), a linear that follows a residual add gathers the partial rows of its own k-slice itself (one hop), and the waves
 * decode their first three weight units of the NEXT linear while they wait for its input.
 *   exl3_pstep_create   builds the plan (device-resident op / tile tables, slab and counter buffers) for the given tensors.  K in {2,3,4,5,6,8},
 *                       cb = 2 (mul1), hidden a multiple of 128 and <= 4096, head_dim 64 | 128, 4-bit cache.  flags: bit 0 = record phase stamps,
 *                       bit 1 = the owner form of the residual edges (two hops; A/B runs; env EXL3_HIP_PSTEP_OWNERS=1), bits 8..11 = the lm_head's bits per weight when they differ from the layers' K (instantiated: 6), bit 2 = the decode attention over the
 *                       4-bit paged cache INSIDE the step (libtorch/attention.cpp:246-504 at q_len 1: o_proj consumes the attention output instead of q; head_dim 128).
 *   exl3_pstep_run      one decode step: R = the int64 fixed-point residual holding the embedded token (exl3_fx_init / exl3_fx_init_prep, which
 *                       also produce rope_sin / rope_cos / slots); logits fp16 [vocab]; q_out optional fp16 [heads_q * head_dim].  Graph-capturable.
 *   exl3_pstep_run_attn the step of a plan created with flags bit 2: block_table int32 [blocks_per_seq] and cache_seqlens int32 [1] (length INCLUDING the new token)
 *                       of the one sequence, page size (a multiple of 16), softmax scale.  Graph-capturable (the length is read on the device).
 *                       bit 4 = stream the checkpoint's tile-row-major tensors as they are (default: exl3_pstep_create copies every op's packed words ONCE into the
 *                       order the plan's workgroups and streaming waves read them -- one contiguous run per wave instead of 1 KiB row pieces at a stride of
 *                       n / 16 x 16 K bytes; SURVEY 8(f)4: a legal load-time transform; the plan then owns a second copy of the weights, the caller's tensors
 *                       are not read by the step any more.  env EXL3_HIP_PSTEP_REPACK=0 | 1 overrides).
 *                       At create the kernel's occupancy is checked (the step needs one co-resident workgroup per CU): EXL3_ERR_ARG if it does not fit.
 *   exl3_pstep_set      decode-ahead units 0..3 (-1: keep; default 3), spin limit of the bounded waits (0: keep).
 *   exl3_pstep_error    synchronises the stream; 1 if a wait ever timed out (results invalid: the step wrote NaN logits), else 0; clears the flag.
 *   exl3_pstep_error_peek  the same flag WITHOUT a synchronisation (a pinned host word the kernel sets when a wait times out): what a caller that replays a captured
 *                       step checks between replays; does not clear.
 *   exl3_pstep_attn_geometry  (plans with the attention inside) out3 = { context splits in use, tokens per split, 128-token steps per split } at sequence length len
 *                       (the kernel derives the same on the device from cache_seqlens).
 *   exl3_pstep_unpack_op  the inverse of the load-time repack: matrix `mat` of op `op` (op = 4 layer + {0 q|k|v, 1 o, 2 gate|up, 3 down}, last = lm_head) copied from
 *                       the plan's order back into a checkpoint-layout tensor [k / 16][n / 16][16 K] -- equality with the original proves the permutation.
 *   exl3_pstep_stamps   copies the phase stamps of the last run ([nops][ncu][32] x u64, 100 MHz) to host memory; returns nops * ncu * 32. */
/* K: bits per weight of this tensor, cb: its codebook (0 3INST, 1 mcg, 2 mul1: exllamav3_ext/quant/codebook.cuh:56-90).  K = 0: the create call's K and cb.
 * The tensors of one fused linear (q / k / v; gate / up) share K and cb -- a reference qgroup (modules/attn.py:244-308, modules/mlp.py:537-574) is quantized as one;
 * different fused linears may differ by one bit (fractional-bpw checkpoints: conversion/allocation.py:131-141 bumps whole qgroups). */
typedef struct { const void* trellis; const void* suh; const void* svh; int k, n; int K, cb; } exl3_pstep_linear_t;
typedef struct
{
    exl3_pstep_linear_t q, k, v, o, gate, up, down;
    const void* norm1; const void* norm2;
    void* k_cache; void* k_scales; void* v_cache; void* v_scales;
} exl3_pstep_layer_t;
int exl3_pstep_create(void** handle_out, const exl3_pstep_layer_t* layers, int n_layers, const exl3_pstep_linear_t* head, const void* final_norm,
                      int hidden, int heads_q, int heads_kv, int head_dim, int K, int cb, float eps, int rope_mode, int flags);
/* the planner alone, host logic (no device needed): rectangles [ncu][12] = {matrix, first column block, column blocks, first Hadamard block, blocks, slice,
 * side task, flags, first work unit in the op's repacked weights, pA, pB, pC} of one op kind (0 q|k|v, 1 o_proj, 2 gate|up, 3 down, 4 lm_head) for a chip of ncu CUs;
 * pA / pB / pC = n << 2 | e: the work units each streaming wave of age group A (waves 0-3) / B (4-7) / C (8-11) takes of the rectangle (the first e waves of the group
 * one more; all zero: the uniform partition) -- a SIMD runs its waves oldest-first, so equal shares leave the youngest streaming alone at the end;
 * env EXL3_HIP_PSTEP_SHARES="fA,fB" (permille) sets the ratio, "0,0" = uniform everywhere.  *S_out = the op's k-slices */
int exl3_pstep_plan_tiles(int hidden, int inter, int heads_q, int heads_kv, int head_dim, int vocab, int ncu, int op_kind, int32_t* tiles_out, int* S_out);
int exl3_pstep_run(void* handle, void* R, void* logits, void* q_out, const float* rope_sin, const float* rope_cos, const int64_t* slots, void* stream);
int exl3_pstep_run_attn(void* handle, void* R, void* logits, void* q_out, const float* rope_sin, const float* rope_cos, const int64_t* slots,
                        const int32_t* block_table, const int32_t* cache_seqlens, int blocks_per_seq, int page_size, float scale, void* stream);
int exl3_pstep_error(void* handle, void* stream);
int exl3_pstep_error_peek(void* handle);
int exl3_pstep_attn_geometry(void* handle, int len, int* out3);
int exl3_pstep_unpack_op(void* handle, int op, int mat, void* trellis_out, void* stream);
/* Tensor parallelism INSIDE the step (exl3_pstep_create flags bits 12..15 = ranks (2..8), bits 16..19 = this rank; the tensors passed are this rank's shards, heads_q / heads_kv
 * the rank's, hidden the model's).  Replaces the all-reduce launches behind o_proj / down_proj (model/model_tp_backend.py:119-126; modules/attn.py:547, modules/mlp.py:770):
 * every rank pushes the partial rows of its row shards as tagged lines into EVERY rank's exchange buffer over IPC-mapped addresses; the consumers of all ranks sum the same
 * ranks x slices lines in the same order (bit-identical residual rows on every rank), no launch, flag or fence in between.
 *   exl3_pstep_tp_handle     the 64-byte IPC handle of this rank's exchange buffer (send it to the peers over the process group);
 *   exl3_pstep_tp_open_peer  maps rank r's buffer from its handle;
 *   exl3_pstep_tp_commit     after every peer is mapped; the caller barriers the ranks behind it and before the first step.  All ranks then run the same sequence of steps. */
int exl3_pstep_tp_handle(void* handle, void* handle64_out);
int exl3_pstep_tp_open_peer(void* handle, int peer_rank, const void* handle64);
int exl3_pstep_tp_commit(void* handle);
int64_t exl3_pstep_tp_peek(void* handle, void* host_out, int64_t max_bytes);    /* diagnostics: this rank's exchange buffer -> host memory (synchronises); bytes copied */
int exl3_pstep_set(void* handle, int decode_ahead_units, int spin_limit);
int exl3_pstep_describe(void* handle, char* buf, int buf_bytes);
int64_t exl3_pstep_stamps(void* handle, uint64_t* host_out, int64_t max_words, void* stream);
int exl3_pstep_destroy(void* handle);

#ifdef __cplusplus
}
#endif
#endif
