// "Glue" kernels of the fused decode step (m <= 16 rows): everything between two quantized GEMVs of a Llama layer in ONE
// launch each.  The GEMVs run with a deferred epilogue (raw rotated-basis fp32 partial slabs [colblock][S][m][128] in the
// per-device workspace) and pre-rotated inputs, so each glue kernel
//     reduces the split-k slabs -> output Hadamard * svh (+bias) -> op-specific middle -> (x * suh) input Hadamard of the
//     NEXT linear(s) + per-block sums
// and replaces what the reference runs as separate graph nodes inside its BC_* runners: the split-k hand-off + output
// Hadamard of exl3_gemm, rms_norm / rms_norm_res_in, rope, quant_cache_paged, silu_mul and the input Hadamard
// (libtorch/attention.cpp:246-504, libtorch/mlp.cpp:14-91, quant/hadamard_inner.cuh:283-413 fuses the last two as well).
// At batch 1 every one of those is a ~4.5 us latency-bound launch on MI355X (profiles/r01_bench_decode_kernel_stats.csv);
// a 128-wide Hadamard block is handled by one 32-lane half-wave with 4 values per lane, so all of this is register work.
//
// Rounding points follow the unfused ops exactly (they are the same device functions), so fused and unfused pipelines agree
// to fp32 summation order.
#include <string.h>
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include "exl3_glue_device.cuh"


// ------------------------------------------------------------------------------------------------
// G1: [reduce + out-had + svh (+bias)] -> residual += y -> RMSNorm -> in-had for up to 3 next linears.  Single workgroup.
//     y is the fp32 output of o_proj / down_proj (architecture/llama.py:95,111 out_dtype float); the residual stream is fp16.
//     has_y == 0: first layer, no pending sublayer output (plain rms_norm of the residual).
// ------------------------------------------------------------------------------------------------
struct NormTargets { const half_t* suh[3]; half_t* xh[3]; float* xsum[3]; int count; };

#define GN_MAXT 16       // tasks (row, block) per half-wave: m * hidden/128 <= 32 * GN_MAXT

__global__ __launch_bounds__(1024)
void glue_norm_kernel(SlabRef y, int has_y, const float* __restrict__ y_dense, const half_t* __restrict__ svh, const half_t* __restrict__ bias,
                      half_t* __restrict__ resid, const half_t* __restrict__ w, float eps, NormTargets tg, int m, int hidden,
                      half_t* __restrict__ xn_out)
{
    __shared__ float ss_part[16 * 128];          // [row][block] partial sums of squares (hidden <= 16384)
    const int tid = threadIdx.x, l = tid & 31, hw = tid >> 5, nhw = blockDim.x >> 5;
    const int nblk = hidden >> 7;
    const int tasks = m * nblk;
    // phase 1: residual update + sum of squares per (row, block); the updated residual stays in registers
    half4_t r_first = { 0, 0, 0, 0 };                 // the half-wave's first task keeps its row in registers; further tasks re-read what they stored (or never changed)
    // phase-2 operands of this half-wave's first task are fetched now, with the phase-1 loads (at batch 1 there is only one
    // task per half-wave, and a load issued after the barrier would add a full memory latency to a ~5 us kernel)
    half4_t w_first, suh_first[3];
    {
        const int t0 = hw < tasks ? hw : 0;
        const int blk0 = t0 % nblk;
        w_first = ((const half4_t*) (w + blk0 * 128))[l];
        #pragma unroll
        for (int i = 0; i < 3; ++i) suh_first[i] = i < tg.count ? ((const half4_t*) (tg.suh[i] + blk0 * 128))[l] : half4_t{ 0, 0, 0, 0 };
    }
    #pragma unroll
    for (int it = 0; it < GN_MAXT; ++it)
    {
        const int t = it * nhw + hw;
        if (it * nhw >= tasks) break;
        const bool act = t < tasks;
        const int row = act ? t / nblk : 0, blk = act ? t % nblk : 0;
        half4_t r = ((const half4_t*) (resid + (size_t) row * hidden + blk * 128))[l];
        float r0 = (float) r.x, r1 = (float) r.y, r2 = (float) r.z, r3 = (float) r.w;
        if (has_y)
        {
            float h0, h1, h2, h3;
            if (y_dense)
            {
                // finished fp32 sublayer output (e.g. after a tensor-parallel all-reduce)
                float4_t yv = ((const float4_t*) (y_dense + (size_t) row * hidden + blk * 128))[l];
                h0 = yv.x; h1 = yv.y; h2 = yv.z; h3 = yv.w;
            }
            else
            {
                const half4_t sc = ((const half4_t*) (svh + blk * 128))[l];
                out_had(slab_sum(y, blk, row, m, l), l, h0, h1, h2, h3);
                h0 *= (float) sc.x; h1 *= (float) sc.y; h2 *= (float) sc.z; h3 *= (float) sc.w;
                if (bias) { half4_t b = ((const half4_t*) (bias + blk * 128))[l]; h0 += (float) b.x; h1 += (float) b.y; h2 += (float) b.z; h3 += (float) b.w; }
            }
            // r += y, rounded to the residual dtype (norm.cu:193-218)
            r = half4_t{ f2h(r0 + h0), f2h(r1 + h1), f2h(r2 + h2), f2h(r3 + h3) };
            r0 = (float) r.x; r1 = (float) r.y; r2 = (float) r.z; r3 = (float) r.w;
            if (act) ((half4_t*) (resid + (size_t) row * hidden + blk * 128))[l] = r;
        }
        if (it == 0) r_first = r;
        float ss = r0 * r0;
        ss = __builtin_fmaf(r1, r1, ss); ss = __builtin_fmaf(r2, r2, ss); ss = __builtin_fmaf(r3, r3, ss);
        #pragma unroll
        for (int i = 1; i < 32; i <<= 1) ss += xor_lane(ss, i);
        if (act && l == 0) ss_part[row * 128 + blk] = ss;
    }
    __syncthreads();
    // phase 2: normalise (fp32, one rounding to fp16 like rms_norm) and rotate for every consumer
    #pragma unroll
    for (int it = 0; it < GN_MAXT; ++it)
    {
        const int t = it * nhw + hw;
        if (it * nhw >= tasks) break;
        const bool act = t < tasks;
        const int row = act ? t / nblk : 0, blk = act ? t % nblk : 0;
        // row sum of squares: fixed-order tree over the block partials (deterministic), redundantly per half-wave
        float s2 = 0.0f;
        for (int b0 = 0; b0 < nblk; b0 += 32)
        {
            float v = (b0 + l < nblk) ? ss_part[row * 128 + b0 + l] : 0.0f;
            #pragma unroll
            for (int i = 1; i < 32; i <<= 1) v += xor_lane(v, i);
            s2 += v;
        }
        const float rmf = __frsqrt_rn(s2 / (float) hidden + eps);
        half4_t r = it == 0 ? r_first : ((const half4_t*) (resid + (size_t) row * hidden + blk * 128))[l];
        half4_t wv = it == 0 ? w_first : ((const half4_t*) (w + blk * 128))[l];
        half4_t xn = { f2h((float) r.x * (float) wv.x * rmf), f2h((float) r.y * (float) wv.y * rmf),
                       f2h((float) r.z * (float) wv.z * rmf), f2h((float) r.w * (float) wv.w * rmf) };
        if (xn_out && act) ((half4_t*) (xn_out + (size_t) row * hidden + blk * 128))[l] = xn;
        #pragma unroll
        for (int i = 0; i < 3; ++i)
        {
            if (i < tg.count)
            {
                const half4_t sv = it == 0 ? suh_first[i] : ((const half4_t*) (tg.suh[i] + blk * 128))[l];
                float sum = in_had_store_v(xn, sv, tg.xh[i] + (size_t) row * hidden + blk * 128, l, act);
                if (act && l == 0 && tg.xsum[i]) tg.xsum[i][(size_t) row * nblk + blk] = sum;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// G1a: the distributed half of G1: per (row, 128-block) half-wave: [reduce + out-had + svh (+bias)] -> residual += y (fp16) -> sum of
//      squares of the block -> ss_part[row][block].  The consumer GEMVs (GEMV_IN_NORM) finish the RMSNorm while they build their
//      activation fragments, so the norm needs neither a single-workgroup pass over all slabs nor its own launch.
//      Same arithmetic as glue_norm_kernel phase 1 (bit-identical residual and partial sums).
// ------------------------------------------------------------------------------------------------
// ROT (batches above 4 rows, one rank): the task also does glue_rotate's work for the consumers of the NEW residual -- normalised with the
// PREVIOUS residual's 1/rms (rot.ss_prev, complete before this launch; the new row sum does not exist until every block of the row is done),
// so x = fp16(resid_new * w * r_prev) and whoever finishes the consumers' outputs multiplies by r_new / r_prev (GemvRescale: exl3_glue_qkv_rs,
// exl3_glue_act_rs; the quantized linear commutes with the row scalar).  ss_part must then be a different buffer than rot.ss_prev.
struct ResidRotate { const float* ss_prev; const half_t* w; float eps; NormTargets tg; };

// Arguments of glue_resid_kernel.  The kernel is a chain of latencies, not of work (4.7 us for < 100 KB): everything a task needs is in the
// first cache line of this block and is read in one batch, the task index is split with a multiply-high (gemv_udiv), half-wave tasks per
// workgroup come from here instead of blockDim (an implicit argument on another line), and every load a task needs -- residual, scales, bias,
// the slab lines -- is issued before the first value is used.
struct ResidArgs
{
    const float* y_base; const float* y_dense; const half_t* svh; const half_t* bias; half_t* resid; float* ss_part;     // 48 B
    int y_S, has_y, m, hidden, nblk, tpw; uint32_t magic_nblk; int pad_;                                                // 80 B
    ResidRotate rot;
};

template <bool ROT>
__global__ __launch_bounds__(256)
void glue_resid_kernel(const ResidArgs a)
{
    const float* const y_base = a.y_base; const float* const y_dense = a.y_dense; const half_t* const svh = a.svh; const half_t* const bias = a.bias;
    half_t* const resid = a.resid; float* const ss_part = a.ss_part;
    const int y_S = a.y_S, has_y = a.has_y, m = a.m, hidden = a.hidden, nblk = a.nblk, tpw = a.tpw;
    const uint32_t magic_nblk = a.magic_nblk;
    const SlabRef y = { y_base, y_S };
    const int tid = threadIdx.x, l = tid & 31, hw = tid >> 5;
    const int tasks = m * nblk;
    const int t = blockIdx.x * tpw + hw;                               // half-wave tasks per workgroup = blockDim / 32
    const bool act = t < tasks;
    const int tt = act ? t : 0;
    const int row = gemv_udiv(tt, magic_nblk), blk = tt - row * nblk;
    // ---- all loads first
    half4_t r = ((const half4_t*) (resid + (size_t) row * hidden + blk * 128))[l];
    half4_t wv = { 0, 0, 0, 0 }, sv[3];
    float ssp0 = 0.0f;
    if constexpr (ROT)
    {
        wv = ((const half4_t*) (a.rot.w + blk * 128))[l];
        #pragma unroll
        for (int i = 0; i < 3; ++i) sv[i] = i < a.rot.tg.count ? ((const half4_t*) (a.rot.tg.suh[i] + blk * 128))[l] : half4_t{ 0, 0, 0, 0 };
        ssp0 = l < nblk ? a.rot.ss_prev[(size_t) row * nblk + l] : 0.0f;
    }
    float r0, r1, r2, r3;
    if (has_y)
    {
        float h0, h1, h2, h3;
        if (y_dense)
        {
            float4_t yv = ((const float4_t*) (y_dense + (size_t) row * hidden + blk * 128))[l];
            h0 = yv.x; h1 = yv.y; h2 = yv.z; h3 = yv.w;
        }
        else
        {
            const half4_t sc = ((const half4_t*) (svh + blk * 128))[l];
            half4_t b = { 0, 0, 0, 0 };
            if (bias) b = ((const half4_t*) (bias + blk * 128))[l];
            out_had(slab_sum(y, blk, row, m, l), l, h0, h1, h2, h3);
            h0 *= (float) sc.x; h1 *= (float) sc.y; h2 *= (float) sc.z; h3 *= (float) sc.w;
            if (bias) { h0 += (float) b.x; h1 += (float) b.y; h2 += (float) b.z; h3 += (float) b.w; }
        }
        r = half4_t{ f2h((float) r.x + h0), f2h((float) r.y + h1), f2h((float) r.z + h2), f2h((float) r.w + h3) };
        if (act) ((half4_t*) (resid + (size_t) row * hidden + blk * 128))[l] = r;
    }
    r0 = (float) r.x; r1 = (float) r.y; r2 = (float) r.z; r3 = (float) r.w;
    float ss = r0 * r0;
    ss = __builtin_fmaf(r1, r1, ss); ss = __builtin_fmaf(r2, r2, ss); ss = __builtin_fmaf(r3, r3, ss);
    #pragma unroll
    for (int i = 1; i < 32; i <<= 1) ss += xor_lane(ss, i);
    if (act && l == 0) ss_part[(size_t) row * nblk + blk] = ss;
    if constexpr (ROT)
    {
        // 1/rms of the previous residual: the same fixed-order sum as glue_rotate / GEMV_IN_NORM
        float s2 = 0.0f;
        for (int b0 = 0; b0 < nblk; b0 += 32)
        {
            float v = b0 == 0 ? ssp0 : ((b0 + l < nblk) ? a.rot.ss_prev[(size_t) row * nblk + b0 + l] : 0.0f);
            #pragma unroll
            for (int i = 1; i < 32; i <<= 1) v += xor_lane(v, i);
            s2 += v;
        }
        const float rmf = __frsqrt_rn(s2 / (float) hidden + a.rot.eps);
        const half4_t xn = { f2h(r0 * (float) wv.x * rmf), f2h(r1 * (float) wv.y * rmf), f2h(r2 * (float) wv.z * rmf), f2h(r3 * (float) wv.w * rmf) };
        #pragma unroll
        for (int i = 0; i < 3; ++i)
        {
            if (i < a.rot.tg.count)
            {
                float sum = in_had_store_v(xn, sv[i], a.rot.tg.xh[i] + (size_t) row * hidden + blk * 128, l, act);
                if (act && l == 0 && a.rot.tg.xsum[i]) a.rot.tg.xsum[i][(size_t) row * nblk + blk] = sum;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// G1m: glue_resid for a MoE block whose weighted down launch was deferred (exl3_mgemm_indexed_act_deferred): per (token, 128-block) half-wave,
//      y = sum over the token's top_k slots, in slot order from zero, of  had128(sum_s slab) * (1/sqrt(128) * w_slot) * svh[expert of the slot]
//      (fp32: the arithmetic and order of exl3_gemv_reduce_kernel + mgemm_slot_reduce_kernel, i.e. of the reference's weighted exl3_mgemm,
//      quant/exl3_gemm_kernel.cuh:241-290), then resid = fp16(resid + y) and the block's sum of squares as in glue_resid.
//      Replaces the split-k reduce launch, the slot-sum launch and the residual launch behind a MoE block by one.
// ------------------------------------------------------------------------------------------------
struct ResidMoeArgs
{
    const float* slabs; const uint64_t* tbl_svh; const int64_t* indices; const half_t* weights; half_t* resid; float* ss_part;     // 48 B
    int S, top_k, tokens, hidden, nblk, tpw; uint32_t magic_nblk; int pad_;
};

__global__ __launch_bounds__(256)
void glue_resid_moe_kernel(const ResidMoeArgs a)
{
    const float* const slabs = a.slabs; const uint64_t* const tbl_svh = a.tbl_svh; const int64_t* const indices = a.indices;
    const half_t* const weights = a.weights; half_t* const resid = a.resid; float* const ss_part = a.ss_part;
    const int S = a.S, top_k = a.top_k, tokens = a.tokens, hidden = a.hidden, nblk = a.nblk, tpw = a.tpw;
    const uint32_t magic_nblk = a.magic_nblk;
    const int tid = threadIdx.x, l = tid & 31, hw = tid >> 5;
    const int t = blockIdx.x * tpw + hw;
    const bool act = t < tokens * nblk;
    const int tt = act ? t : 0;
    const int row = gemv_udiv(tt, magic_nblk), blk = tt - row * nblk;
    half4_t r = ((const half4_t*) (resid + (size_t) row * hidden + blk * 128))[l];
    float y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f;
    for (int j = 0; j < top_k; ++j)
    {
        const int slot = row * top_k + j;
        const int e = (int) indices[slot];
        const float w = (float) weights[slot];
        const half4_t sc = ((const half4_t*) ((const half_t*) tbl_svh[e] + blk * 128))[l];
        const SlabRef sr = { slabs + (size_t) slot * nblk * S * 128, S };     // m = 1 row per slot: [slot][block][S][128]
        const float4_t v = slab_sum(sr, blk, 0, 1, l);
        float h0 = v.x, h1 = v.y, h2 = v.z, h3 = v.w;
        had128_f32x4(h0, h1, h2, h3, l);
        const float os = HAD_R_SCALE_128 * w;
        h0 *= os; h1 *= os; h2 *= os; h3 *= os;
        y0 += h0 * (float) sc.x; y1 += h1 * (float) sc.y; y2 += h2 * (float) sc.z; y3 += h3 * (float) sc.w;
    }
    r = half4_t{ f2h((float) r.x + y0), f2h((float) r.y + y1), f2h((float) r.z + y2), f2h((float) r.w + y3) };
    if (act) ((half4_t*) (resid + (size_t) row * hidden + blk * 128))[l] = r;
    const float r0 = (float) r.x, r1 = (float) r.y, r2 = (float) r.z, r3 = (float) r.w;
    float ss = r0 * r0;
    ss = __builtin_fmaf(r1, r1, ss); ss = __builtin_fmaf(r2, r2, ss); ss = __builtin_fmaf(r3, r3, ss);
    #pragma unroll
    for (int i = 1; i < 32; i <<= 1) ss += xor_lane(ss, i);
    if (act && l == 0) ss_part[(size_t) row * nblk + blk] = ss;
}

// ------------------------------------------------------------------------------------------------
// G1b: the other half of the norm boundary for batches above 4 rows: per (row, 128-block) half-wave: 1/rms of the row from the per-block
//      sums of squares (same fixed-order sum as everywhere else) -> x = fp16(resid * w / rms) -> (x * suh_i) input Hadamard for up to 3
//      consumers.  At m <= 4 the consumer GEMVs do this themselves (GEMV_IN_NORM); at m = 16 that would repeat 16 Hadamards per block
//      in every one of the 48..224 column-block workgroups, 6x the VALU time of the weight decode (tools/gemv_timeline.py).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void glue_rotate_kernel(const half_t* __restrict__ resid, const float* __restrict__ ss_part, const half_t* __restrict__ w, float eps,
                        NormTargets tg, int m, int hidden)
{
    const int tid = threadIdx.x, l = tid & 31, hw = tid >> 5;
    const int nblk = hidden >> 7;
    const int tasks = m * nblk;
    const int t = blockIdx.x * (blockDim.x >> 5) + hw;                 // half-wave tasks per workgroup = blockDim / 32
    const bool act = t < tasks;
    const int row = act ? t / nblk : 0, blk = act ? t % nblk : 0;
    const half4_t r = ((const half4_t*) (resid + (size_t) row * hidden + blk * 128))[l];
    const half4_t wv = ((const half4_t*) (w + blk * 128))[l];
    half4_t sv[3];
    #pragma unroll
    for (int i = 0; i < 3; ++i) sv[i] = i < tg.count ? ((const half4_t*) (tg.suh[i] + blk * 128))[l] : half4_t{ 0, 0, 0, 0 };
    float s2 = 0.0f;
    for (int b0 = 0; b0 < nblk; b0 += 32)
    {
        float v = (b0 + l < nblk) ? ss_part[(size_t) row * nblk + b0 + l] : 0.0f;
        #pragma unroll
        for (int i = 1; i < 32; i <<= 1) v += xor_lane(v, i);
        s2 += v;
    }
    const float rmf = __frsqrt_rn(s2 / (float) hidden + eps);
    half4_t xn = { f2h((float) r.x * (float) wv.x * rmf), f2h((float) r.y * (float) wv.y * rmf),
                   f2h((float) r.z * (float) wv.z * rmf), f2h((float) r.w * (float) wv.w * rmf) };
    #pragma unroll
    for (int i = 0; i < 3; ++i)
    {
        if (i < tg.count)
        {
            float sum = in_had_store_v(xn, sv[i], tg.xh[i] + (size_t) row * hidden + blk * 128, l, act);
            if (act && l == 0 && tg.xsum[i]) tg.xsum[i][(size_t) row * nblk + blk] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// G2: q/k/v epilogue per (row, head), head_dim == 128 == one Hadamard block:
//     reduce + out-had + fp16 svh -> RoPE (NEOX or GPTJ; sin/cos once per (row, frequency) in LDS) on q and k ->
//     q fp16 out; k, v -> quantized paged cache append (and optional fp16 copies).
// ------------------------------------------------------------------------------------------------

struct QkvArgs
{
    SlabRef sq, sk, sv;
    const half_t* svh_q; const half_t* svh_k; const half_t* svh_v;
    half_t* q_out; half_t* k_out; half_t* v_out;          // [m][heads*128] fp16 (k_out / v_out optional)
    const float* inv_freq;                                  // [64]
    const int32_t* positions;                               // [m] absolute position of each row's token
    uint32_t* k_cache; half_t* k_scales; uint32_t* v_cache; half_t* v_scales;      // paged quantized cache of this layer (optional)
    const int32_t* block_table; int blocks_per_seq; int page_size;
    int m, hq, hkv, rope_mode;                              // hq / hkv count 128-wide blocks (= heads at head_dim 128, head pairs at 64)
    int hd;
    float attn_factor;
    GemvRescale rs;                                         // ss_new != nullptr: q, k, v came from an exl3_gemv_ex_resid launch (row scale correction)
    // per-step tables of exl3_qkv_prep (all layers of a decode step share positions and block table): sin / cos [m][64] fp32 (x attn_factor)
    // and the physical cache row of every token.  With them the kernel has no sincosf, no LDS table, no barrier and no dependent
    // positions -> block_table load chain; null: it computes all of that itself (the form the reference's per-layer ops have)
    const float* rope_sin; const float* rope_cos; const int64_t* slots;
    int tpw; uint32_t magic_heads;                          // half-wave tasks per workgroup; gemv_magic(hq + 2 hkv)
};

typedef float f2_t __attribute__((ext_vector_type(2)));

template <int KB, int VB, bool TAB>
__global__ __launch_bounds__(256)
void glue_qkv_kernel(const QkvArgs a)
{
    __shared__ float sn_s[TAB ? 1 : 16 * 64], cs_s[TAB ? 1 : 16 * 64];
    // every scalar argument in one batch (see ResidArgs)
    const int a_m = a.m, a_hq = a.hq, a_hkv = a.hkv, a_hd = a.hd, rope_mode = a.rope_mode, tpw = a.tpw, page_size = a.page_size, bps = a.blocks_per_seq;
    const uint32_t magic_heads = a.magic_heads;
    const float attn_factor = a.attn_factor;
    half_t* const q_out = a.q_out; half_t* const k_out = a.k_out; half_t* const v_out = a.v_out;
    uint32_t* const k_cache = a.k_cache; half_t* const k_scales = a.k_scales; uint32_t* const v_cache = a.v_cache; half_t* const v_scales = a.v_scales;
    const int32_t* const positions = a.positions; const int32_t* const block_table = a.block_table; const float* const inv_freq = a.inv_freq;
    const float* const rope_sin = a.rope_sin; const float* const rope_cos = a.rope_cos; const int64_t* const slots = a.slots;
    const GemvRescale rs = a.rs;
    // (pinned: the compiler otherwise fetches these in four more scalar round trips, each behind a branch, before the first slab line is requested)
    asm volatile("" :: "s"(a.sq.base), "s"(a.sk.base), "s"(a.sv.base), "s"(a.sq.S), "s"(a.sk.S), "s"(a.sv.S), "s"(a.svh_q), "s"(a.svh_k), "s"(a.svh_v));
    asm volatile("" :: "s"(q_out), "s"(k_cache), "s"(k_scales), "s"(v_cache), "s"(v_scales), "s"(rope_sin), "s"(rope_cos), "s"(slots), "s"(rs.ss_prev), "s"(rs.ss_new), "s"(rs.k), "s"(rs.eps));
    const int tid = threadIdx.x, l = tid & 31, hw = tid >> 5;
    const int heads = a_hq + 2 * a_hkv;
    const int tasks = a_m * heads;
    const int t = blockIdx.x * tpw + hw;                               // half-wave tasks per workgroup = blockDim / 32
    const bool act = t < tasks;
    const int tt = act ? t : 0;
    const int row = gemv_udiv(tt, magic_heads), head = tt - row * heads;
    const int kind = head < a_hq ? 0 : (head < a_hq + a_hkv ? 1 : 2);
    const int hi = kind == 0 ? head : (kind == 1 ? head - a_hq : head - a_hq - a_hkv);
    const SlabRef& sr = kind == 0 ? a.sq : (kind == 1 ? a.sk : a.sv);
    const half_t* svh = (kind == 0 ? a.svh_q : (kind == 1 ? a.svh_k : a.svh_v)) + hi * 128;
    // ---- small loads first, then the slab lines; nothing is used before the slab sum
    const half4_t sc = ((const half4_t*) svh)[l];
    float rs_p = 0.0f, rs_n = 0.0f;
    if (rs.ss_new && l < (rs.k >> 7)) { rs_p = rs.ss_prev[(size_t) row * (rs.k >> 7) + l]; rs_n = rs.ss_new[(size_t) row * (rs.k >> 7) + l]; }
    const int ph = a_hd >> 3;                                           // NEOX partner distance in lanes: 16 (head_dim 128) or 8 (64)
    float4_t sn4 = { 0.f, 0.f, 0.f, 0.f }, cs4 = { 0.f, 0.f, 0.f, 0.f };
    int64_t token_pos = 0;
    if constexpr (TAB)
    {
        // NEOX: frequencies 4 (l mod ph) .. + 3; GPTJ: 2 (l mod hd/4), + 1 (the upper two lanes of the float4 are not used)
        const int f = rope_mode == 2 ? 4 * (l & (ph - 1)) : 2 * (l & ((a_hd >> 2) - 1));
        if (rope_mode == 2) { sn4 = *((const float4_t*) (rope_sin + row * 64 + f)); cs4 = *((const float4_t*) (rope_cos + row * 64 + f)); }
        else { const f2_t s2 = *((const f2_t*) (rope_sin + row * 64 + f)), c2 = *((const f2_t*) (rope_cos + row * 64 + f)); sn4.x = s2.x; sn4.y = s2.y; cs4.x = c2.x; cs4.y = c2.y; }
        if (k_cache) token_pos = slots[row];
    }
    const float4_t ysum = slab_sum(sr, hi, row, a_m, l);
    if constexpr (!TAB)
    {
        const int nfreq = a_hd >> 1;                                    // 64 (head_dim 128) or 32 (head_dim 64: two heads per block)
        for (int i = tid; i < a_m * nfreq; i += 32 * tpw)
        {
            const int rw = a_hd == 128 ? i >> 6 : i >> 5, f = i & (nfreq - 1);
            float sn, cs;
            sincosf(inv_freq[f] * (float) positions[rw], &sn, &cs);
            sn_s[rw * 64 + f] = sn * attn_factor; cs_s[rw * 64 + f] = cs * attn_factor;
        }
        __syncthreads();
    }
    if constexpr (!TAB)
    {
        if (rope_mode == 2) { const int f = 4 * (l & (ph - 1)); sn4 = *((const float4_t*) (sn_s + row * 64 + f)); cs4 = *((const float4_t*) (cs_s + row * 64 + f)); }
        else
        {
            const int lf = 2 * (l & ((a_hd >> 2) - 1));
            sn4.x = sn_s[row * 64 + lf]; sn4.y = sn_s[row * 64 + lf + 1]; cs4.x = cs_s[row * 64 + lf]; cs4.y = cs_s[row * 64 + lf + 1];
        }
    }
    // reduce -> out-Hadamard -> row-scale correction -> fp16 x svh -> RoPE on q and k (lane l holds dims 4l..4l+3): exl3_glue_device.cuh
    const half4_t y = qkv_block_finish(ysum, sc, rs, row, l, rs_p, rs_n, kind != 2, rope_mode, ph, sn4, cs4);
    if (kind == 0 && act) ((half4_t*) (q_out + ((size_t) row * a_hq + hi) * 128))[l] = y;
    half_t* dense = kind == 1 ? k_out : (kind == 2 ? v_out : nullptr);
    if (dense && act && kind != 0) ((half4_t*) (dense + ((size_t) row * a_hkv + hi) * 128))[l] = y;
    // quantized append (all lanes take part in the shuffles; stores are predicated)
    const bool do_q = act && kind != 0 && k_cache != nullptr;
    if constexpr (!TAB)
    {
        const int pos = positions[row];
        const int page_idx = pos / page_size;
        token_pos = k_cache ? (int64_t) block_table[row * bps + page_idx] * page_size + (pos % page_size) : 0;
    }
    const int groups_per_token = a_hkv * 4;
    const int64_t gbase = token_pos * groups_per_token + hi * 4 + (l >> 3);
    {
        float v0 = (float) y.x, v1 = (float) y.y, v2 = (float) y.z, v3 = (float) y.w;
        kv_quant_regs<KB>(v0, v1, v2, v3, k_cache ? k_cache + gbase * KB : nullptr, k_scales ? k_scales + gbase : nullptr, do_q && kind == 1, tid & 63);
        kv_quant_regs<VB>(v0, v1, v2, v3, v_cache ? v_cache + gbase * VB : nullptr, v_scales ? v_scales + gbase : nullptr, do_q && kind == 2, tid & 63);
    }
}

// ------------------------------------------------------------------------------------------------
// G3: gate/up epilogue per (row, 128-block of the intermediate dim): reduce + out-had + fp16 svh for g and u ->
//     a = fp16(silu(g) * u) (activation.cu) -> in-had with suh_down -> xh_down + block sum.
// ------------------------------------------------------------------------------------------------
// arguments in one block, read in one batch; multiply-high task split; tasks per workgroup from here (see ResidArgs)
struct ActArgs
{
    SlabRef sg, su; const half_t* svh_g; const half_t* svh_u; const half_t* suh_d; half_t* xh_d; float* xsum_d; half_t* a_out;      // 80 B
    int m, inter, nblk, tpw; uint32_t magic_nblk; int pad_;
    GemvRescale rs;
};

__global__ __launch_bounds__(256)
void glue_act_kernel(const ActArgs a)
{
    const SlabRef sg = a.sg, su = a.su;
    const half_t* const svh_g = a.svh_g; const half_t* const svh_u = a.svh_u; const half_t* const suh_d = a.suh_d;
    half_t* const xh_d = a.xh_d; float* const xsum_d = a.xsum_d; half_t* const a_out = a.a_out;
    const int m = a.m, inter = a.inter, nblk = a.nblk, tpw = a.tpw;
    const uint32_t magic_nblk = a.magic_nblk;
    const GemvRescale rs = a.rs;
    // ONE batch of scalar loads: left alone the compiler fetched the slab pointers behind the first branch (a second scalar round trip in front of the
    // slab loads of a kernel that is nothing but a latency chain) and the output pointers at the tail
    asm volatile("" :: "s"(sg.base), "s"(su.base), "s"(sg.S), "s"(su.S), "s"(xh_d), "s"(xsum_d), "s"(a_out), "s"(rs.ss_prev), "s"(rs.ss_new), "s"(rs.k), "s"(inter));
    const int tid = threadIdx.x, l = tid & 31, hw = tid >> 5;
    const int tasks = m * nblk;
    const int t = blockIdx.x * tpw + hw;                               // half-wave tasks per workgroup = blockDim / 32
    const bool act = t < tasks;
    const int tt = act ? t : 0;
    const int row = gemv_udiv(tt, magic_nblk), blk = tt - row * nblk;
    // all independent loads first (scales of this block), then the slabs
    const half4_t svg = ((const half4_t*) (svh_g + blk * 128))[l], svu = ((const half4_t*) (svh_u + blk * 128))[l];
    const half4_t sud = ((const half4_t*) (suh_d + blk * 128))[l];
    float g0, g1, g2, g3, u0, u1, u2, u3;
    float4_t vg, vu;
    float rs_p = 0.0f, rs_n = 0.0f;
    if (rs.ss_new && l < (rs.k >> 7)) { rs_p = rs.ss_prev[(size_t) row * (rs.k >> 7) + l]; rs_n = rs.ss_new[(size_t) row * (rs.k >> 7) + l]; }
    slab_sum2(sg, su, blk, row, m, l, vg, vu);
    out_had(vg, l, g0, g1, g2, g3);
    out_had(vu, l, u0, u1, u2, u3);
    if (rs.ss_new)
    {
        // gate / up were computed from a row normalised with the previous residual's 1/rms (glue_resid ROT): r_new / r_prev on both
        const float rsc = gemv_rescale(rs, row, l, rs_p, rs_n);
        g0 *= rsc; g1 *= rsc; g2 *= rsc; g3 *= rsc; u0 *= rsc; u1 *= rsc; u2 *= rsc; u3 *= rsc;
    }
    half4_t gh = half4_t{ f2h(g0), f2h(g1), f2h(g2), f2h(g3) } * svg;
    half4_t uh = half4_t{ f2h(u0), f2h(u1), f2h(u2), f2h(u3) } * svu;
    auto silu_mul = [] (half_t g, half_t u) -> half_t { float gf = (float) g; return f2h(gf / (1.0f + __expf(-gf)) * (float) u); };
    half4_t av = { silu_mul(gh.x, uh.x), silu_mul(gh.y, uh.y), silu_mul(gh.z, uh.z), silu_mul(gh.w, uh.w) };
    if (a_out && act) ((half4_t*) (a_out + (size_t) row * inter + blk * 128))[l] = av;
    float sum = in_had_store_v(av, sud, xh_d + (size_t) row * inter + blk * 128, l, act);
    if (act && l == 0 && xsum_d) xsum_d[(size_t) row * nblk + blk] = sum;
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------

extern "C" int exl3_glue_norm(const float* y_slabs, int y_S, const float* y_dense, const void* svh, const void* bias, void* resid, const void* w, float eps,
                              const void* const* suhs, void* const* xhs, float* const* xsums, int count, int m, int hidden,
                              void* xn_out, void* stream)
{
    EXL3_CHECK_ARG(resid && w, "glue_norm: null pointer");
    EXL3_CHECK_ARG(m >= 1 && m <= 16, "glue_norm: 1 <= m <= 16");
    EXL3_CHECK_ARG(hidden % 128 == 0 && hidden <= 16384, "glue_norm: hidden must be a multiple of 128, <= 16384");
    EXL3_CHECK_ARG(m * (hidden / 128) <= 32 * GN_MAXT, "glue_norm: m * hidden / 128 must be <= 512");
    EXL3_CHECK_ARG(count >= 0 && count <= 3, "glue_norm: at most 3 consumers");
    EXL3_CHECK_ARG(!y_slabs || (svh && y_S >= 1), "glue_norm: pending output needs svh");
    NormTargets tg; tg.count = count;
    for (int i = 0; i < 3; ++i)
    {
        tg.suh[i] = i < count ? (const half_t*) suhs[i] : nullptr;
        tg.xh[i] = i < count ? (half_t*) xhs[i] : nullptr;
        tg.xsum[i] = (i < count && xsums) ? xsums[i] : nullptr;
        EXL3_CHECK_ARG(i >= count || (tg.suh[i] && tg.xh[i]), "glue_norm: null consumer pointer");
    }
    SlabRef y = { y_slabs, y_S };
    int tasks = m * (hidden / 128);
    int threads = tasks * 32; if (threads > 1024) threads = 1024; if (threads < 64) threads = 64;
    threads = (threads + 63) / 64 * 64;
    EXL3_CHECK_ARG((tasks + threads / 32 - 1) / (threads / 32) <= GN_MAXT, "glue_norm: too many (row, block) tasks");
    glue_norm_kernel<<<1, threads, 0, (hipStream_t) stream>>>(y, (y_slabs || y_dense) ? 1 : 0, y_dense, (const half_t*) svh, (const half_t*) bias, (half_t*) resid,
                                                             (const half_t*) w, eps, tg, m, hidden, (half_t*) xn_out);
    return exl3_check_launch("glue_norm");
}

template <int KB, bool TAB>
static void launch_qkv_t(int vb, dim3 grid, int threads, hipStream_t st, const QkvArgs& a)
{
    switch (vb)
    {
        case 2: glue_qkv_kernel<KB, 2, TAB><<<grid, threads, 0, st>>>(a); break; case 3: glue_qkv_kernel<KB, 3, TAB><<<grid, threads, 0, st>>>(a); break;
        case 4: glue_qkv_kernel<KB, 4, TAB><<<grid, threads, 0, st>>>(a); break; case 5: glue_qkv_kernel<KB, 5, TAB><<<grid, threads, 0, st>>>(a); break;
        case 6: glue_qkv_kernel<KB, 6, TAB><<<grid, threads, 0, st>>>(a); break; case 7: glue_qkv_kernel<KB, 7, TAB><<<grid, threads, 0, st>>>(a); break;
        default: glue_qkv_kernel<KB, 8, TAB><<<grid, threads, 0, st>>>(a); break;
    }
}
template <int KB>
static void launch_qkv(int vb, dim3 grid, int threads, hipStream_t st, const QkvArgs& a)
{
    if (a.rope_sin) launch_qkv_t<KB, true>(vb, grid, threads, st, a); else launch_qkv_t<KB, false>(vb, grid, threads, st, a);
}

// half-wave tasks of the glue kernels are spread over as many CUs as possible: a task's slab lines (S x 512 B) come in at the per-CU load rate, so
// few fat workgroups are slower than many thin ones (a one-workgroup-per-row fusion of glue_resid + glue_rotate measured 2.2 us SLOWER per
// boundary than the two launches).  64-thread workgroups (2 tasks) up to 512 tasks, 256-thread ones above.
static int g_glue_threads = 0;
extern "C" int exl3_set_glue_threads(int t) { g_glue_threads = t; return EXL3_OK; }
static int glue_threads(int tasks)
{
    if (g_glue_threads > 0) return g_glue_threads;
    return tasks <= 512 ? 64 : 256;
}

extern "C" int exl3_glue_qkv(const float* sq, const float* sk, const float* sv, int S, const void* svh_q, const void* svh_k, const void* svh_v,
                             void* q_out, void* k_out, void* v_out, const float* inv_freq, const int32_t* positions,
                             void* k_cache, void* k_scales, void* v_cache, void* v_scales, const int32_t* block_table, int blocks_per_seq,
                             int page_size, int k_bits, int v_bits, int m, int heads_q, int heads_kv, int head_dim, int rope_mode,
                             float attn_factor, void* stream)
{
    return exl3_glue_qkv_rs(sq, sk, sv, S, svh_q, svh_k, svh_v, q_out, k_out, v_out, inv_freq, positions, k_cache, k_scales, v_cache, v_scales,
                            block_table, blocks_per_seq, page_size, k_bits, v_bits, m, heads_q, heads_kv, head_dim, rope_mode, attn_factor,
                            nullptr, nullptr, 0, 0.0f, stream);
}

// exl3_glue_qkv for slabs produced by exl3_gemv_ex_resid: ss_prev / ss_new [m][hidden/128] + eps give the row scale correction (null: none)
extern "C" int exl3_glue_qkv_rs(const float* sq, const float* sk, const float* sv, int S, const void* svh_q, const void* svh_k, const void* svh_v,
                                void* q_out, void* k_out, void* v_out, const float* inv_freq, const int32_t* positions,
                                void* k_cache, void* k_scales, void* v_cache, void* v_scales, const int32_t* block_table, int blocks_per_seq,
                                int page_size, int k_bits, int v_bits, int m, int heads_q, int heads_kv, int head_dim, int rope_mode,
                                float attn_factor, const float* ss_prev, const float* ss_new, int hidden, float eps, void* stream)
{
    return exl3_glue_qkv_tab(sq, sk, sv, S, svh_q, svh_k, svh_v, q_out, k_out, v_out, inv_freq, positions, k_cache, k_scales, v_cache, v_scales,
                             block_table, blocks_per_seq, page_size, k_bits, v_bits, m, heads_q, heads_kv, head_dim, rope_mode, attn_factor,
                             ss_prev, ss_new, hidden, eps, nullptr, nullptr, nullptr, stream);
}

// Per-step tables for exl3_glue_qkv_tab: sin / cos [m][64] fp32 of (position x inverse frequency) x attn_factor -- the same sincosf() values the
// rope / glue_qkv kernels compute -- and, with a block table, the physical cache row slots[r] = block_table[r][pos / page] * page + pos % page of
// every token.  One launch per decode step instead of that work in every layer (the reference recomputes it in every rope / cache launch:
// rope.cu:60-120, q_cache_kernels.cuh:300-318).
__global__ void qkv_prep_kernel(const float* __restrict__ inv_freq, const int32_t* __restrict__ positions, float attn_factor, int m, int nfreq,
                                const int32_t* __restrict__ block_table, int blocks_per_seq, int page_size,
                                float* __restrict__ sin_out, float* __restrict__ cos_out, int64_t* __restrict__ slots)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * 64) return;
    const int r = i >> 6, f = i & 63;
    const int pos = positions[r];
    if (f < nfreq)
    {
        float sn, cs;
        sincosf(inv_freq[f] * (float) pos, &sn, &cs);
        sin_out[i] = sn * attn_factor; cos_out[i] = cs * attn_factor;
    }
    if (f == 0 && slots && block_table) slots[r] = (int64_t) block_table[r * blocks_per_seq + pos / page_size] * page_size + (pos % page_size);
}

extern "C" int exl3_qkv_prep(const float* inv_freq, const int32_t* positions, float attn_factor, int m, int head_dim, const int32_t* block_table,
                             int blocks_per_seq, int page_size, float* sin_out, float* cos_out, int64_t* slots, void* stream)
{
    EXL3_CHECK_ARG(inv_freq && positions && sin_out && cos_out && m >= 1, "qkv_prep: bad arguments");
    EXL3_CHECK_ARG(head_dim == 128 || head_dim == 64, "qkv_prep: head_dim must be 128 or 64");
    EXL3_CHECK_ARG(!slots || (block_table && page_size > 0 && blocks_per_seq > 0), "qkv_prep: slots need the block table");
    qkv_prep_kernel<<<(m * 64 + 255) / 256, 256, 0, (hipStream_t) stream>>>(inv_freq, positions, attn_factor, m, head_dim / 2, block_table, blocks_per_seq,
                                                                          page_size, sin_out, cos_out, slots);
    return exl3_check_launch("qkv_prep");
}

// exl3_glue_qkv_rs with the per-step tables of exl3_qkv_prep (rope_sin / rope_cos [m][64] fp32, slots int64 [m]; all null: computed in the kernel)
extern "C" int exl3_glue_qkv_tab(const float* sq, const float* sk, const float* sv, int S, const void* svh_q, const void* svh_k, const void* svh_v,
                                 void* q_out, void* k_out, void* v_out, const float* inv_freq, const int32_t* positions,
                                 void* k_cache, void* k_scales, void* v_cache, void* v_scales, const int32_t* block_table, int blocks_per_seq,
                                 int page_size, int k_bits, int v_bits, int m, int heads_q, int heads_kv, int head_dim, int rope_mode,
                                 float attn_factor, const float* ss_prev, const float* ss_new, int hidden, float eps,
                                 const float* rope_sin, const float* rope_cos, const int64_t* slots, void* stream)
{
    EXL3_CHECK_ARG((!rope_sin && !rope_cos && !slots) || (rope_sin && rope_cos && (slots || !k_cache)), "glue_qkv_tab: sin, cos and (with a cache) slots go together");
    EXL3_CHECK_ARG(!ss_new || (ss_prev && hidden > 0 && hidden % 128 == 0), "glue_qkv_rs: rescale needs ss_prev and hidden");
    EXL3_CHECK_ARG(sq && sk && sv && svh_q && svh_k && svh_v && q_out && inv_freq && positions, "glue_qkv: null pointer");
    EXL3_CHECK_ARG(head_dim == 128 || head_dim == 64, "glue_qkv: head_dim must be 128 or 64 (one or two heads per Hadamard block)");
    EXL3_CHECK_ARG((heads_q * head_dim) % 128 == 0 && (heads_kv * head_dim) % 128 == 0, "glue_qkv: heads * head_dim must be a multiple of 128");
    EXL3_CHECK_ARG(m >= 1 && m <= 16, "glue_qkv: 1 <= m <= 16");
    EXL3_CHECK_ARG(rope_mode == 1 || rope_mode == 2, "glue_qkv: rope_mode must be 1 (GPTJ) or 2 (NEOX)");
    EXL3_CHECK_ARG(!k_cache || (k_scales && v_cache && v_scales && block_table && page_size > 0), "glue_qkv: incomplete cache arguments");
    EXL3_CHECK_ARG(!k_cache || (k_bits >= 2 && k_bits <= 8 && v_bits >= 2 && v_bits <= 8), "glue_qkv: cache bits must be in [2, 8]");
    QkvArgs a;
    a.sq = { sq, S }; a.sk = { sk, S }; a.sv = { sv, S };
    a.svh_q = (const half_t*) svh_q; a.svh_k = (const half_t*) svh_k; a.svh_v = (const half_t*) svh_v;
    a.q_out = (half_t*) q_out; a.k_out = (half_t*) k_out; a.v_out = (half_t*) v_out;
    a.inv_freq = inv_freq; a.positions = positions;
    a.k_cache = (uint32_t*) k_cache; a.k_scales = (half_t*) k_scales; a.v_cache = (uint32_t*) v_cache; a.v_scales = (half_t*) v_scales;
    a.block_table = block_table; a.blocks_per_seq = blocks_per_seq; a.page_size = page_size > 0 ? page_size : 256;
    a.m = m; a.hq = heads_q * head_dim / 128; a.hkv = heads_kv * head_dim / 128; a.hd = head_dim; a.rope_mode = rope_mode; a.attn_factor = attn_factor;
    a.rs = GemvRescale{ ss_prev, ss_new, hidden, eps };
    a.rope_sin = rope_sin; a.rope_cos = rope_cos; a.slots = slots;
    int tasks = m * (a.hq + 2 * a.hkv);
    const int th = glue_threads(tasks), tpw = th / 32;
    a.tpw = tpw; a.magic_heads = gemv_magic((uint32_t) (a.hq + 2 * a.hkv));
    dim3 grid((tasks + tpw - 1) / tpw);
    hipStream_t st = (hipStream_t) stream;
    int kb = k_cache ? k_bits : 8, vb = k_cache ? v_bits : 8;
    switch (kb)
    {
        case 2: launch_qkv<2>(vb, grid, th, st, a); break; case 3: launch_qkv<3>(vb, grid, th, st, a); break; case 4: launch_qkv<4>(vb, grid, th, st, a); break;
        case 5: launch_qkv<5>(vb, grid, th, st, a); break; case 6: launch_qkv<6>(vb, grid, th, st, a); break; case 7: launch_qkv<7>(vb, grid, th, st, a); break;
        default: launch_qkv<8>(vb, grid, th, st, a); break;
    }
    return exl3_check_launch("glue_qkv");
}

extern "C" int exl3_glue_act(const float* sg, const float* su, int S, const void* svh_g, const void* svh_u, const void* suh_d,
                             void* xh_d, float* xsum_d, void* a_out, int m, int inter, void* stream)
{
    return exl3_glue_act_rs(sg, su, S, svh_g, svh_u, suh_d, xh_d, xsum_d, a_out, m, inter, nullptr, nullptr, 0, 0.0f, stream);
}

// exl3_glue_act for gate / up slabs computed from a row that exl3_glue_resid_rotate normalised with the previous residual's 1/rms:
// ss_prev / ss_new [m][hidden/128] + eps give r_new / r_prev (null: none)
extern "C" int exl3_glue_act_rs(const float* sg, const float* su, int S, const void* svh_g, const void* svh_u, const void* suh_d,
                                void* xh_d, float* xsum_d, void* a_out, int m, int inter, const float* ss_prev, const float* ss_new, int hidden,
                                float eps, void* stream)
{
    EXL3_CHECK_ARG(!ss_new || (ss_prev && hidden > 0 && hidden % 128 == 0), "glue_act_rs: rescale needs ss_prev and hidden");
    EXL3_CHECK_ARG(sg && su && svh_g && svh_u && suh_d && xh_d, "glue_act: null pointer");
    EXL3_CHECK_ARG(m >= 1 && m <= 16 && inter % 128 == 0, "glue_act: bad dimensions");
    int tasks = m * (inter / 128);
    SlabRef g = { sg, S }, u = { su, S };
    const int th = glue_threads(tasks), tpw = th / 32;
    ActArgs aa;
    memset((void*) &aa, 0, sizeof(aa));
    aa.sg = g; aa.su = u; aa.svh_g = (const half_t*) svh_g; aa.svh_u = (const half_t*) svh_u; aa.suh_d = (const half_t*) suh_d;
    aa.xh_d = (half_t*) xh_d; aa.xsum_d = xsum_d; aa.a_out = (half_t*) a_out;
    aa.m = m; aa.inter = inter; aa.nblk = inter / 128; aa.tpw = tpw; aa.magic_nblk = gemv_magic((uint32_t) (inter / 128));
    aa.rs = GemvRescale{ ss_prev, ss_new, hidden, eps };
    glue_act_kernel<<<(tasks + tpw - 1) / tpw, th, 0, (hipStream_t) stream>>>(aa);
    return exl3_check_launch("glue_act");
}

// sin / cos of (position x inverse frequency), scaled by attn_factor, for every row of a decode step: [m][64] fp32 each.
// All layers of a step share the positions, so the fused pipeline builds this once per step (the reference recomputes it
// inside every rope launch, rope.cu:60-120); values are the same sincosf() results the rope / glue_qkv kernels use.
__global__ void rope_table_kernel(const float* __restrict__ inv_freq, const int32_t* __restrict__ positions, float attn_factor, int m,
                                  float* __restrict__ sin_out, float* __restrict__ cos_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * 64) return;
    float sn, cs;
    sincosf(inv_freq[i & 63] * (float) positions[i >> 6], &sn, &cs);
    sin_out[i] = sn * attn_factor; cos_out[i] = cs * attn_factor;
}

extern "C" int exl3_rope_table(const float* inv_freq, const int32_t* positions, float attn_factor, int m, float* sin_out, float* cos_out, void* stream)
{
    EXL3_CHECK_ARG(inv_freq && positions && sin_out && cos_out && m >= 1, "rope_table: bad arguments");
    rope_table_kernel<<<(m * 64 + 255) / 256, 256, 0, (hipStream_t) stream>>>(inv_freq, positions, attn_factor, m, sin_out, cos_out);
    return exl3_check_launch("rope_table");
}

extern "C" int exl3_glue_resid(const float* y_slabs, int y_S, const float* y_dense, const void* svh, const void* bias, void* resid,
                               float* ss_part, int m, int hidden, void* stream)
{
    EXL3_CHECK_ARG(resid && ss_part, "glue_resid: null pointer");
    EXL3_CHECK_ARG(m >= 1 && m <= 16 && hidden % 128 == 0, "glue_resid: bad dimensions");
    EXL3_CHECK_ARG(!y_slabs || (svh && y_S >= 1), "glue_resid: pending output needs svh");
    SlabRef y = { y_slabs, y_S };
    const int tasks = m * (hidden / 128);
    const int th = glue_threads(tasks), tpw = th / 32;
    ResidArgs ra;
    memset((void*) &ra, 0, sizeof(ra));
    ra.y_base = y.base; ra.y_dense = y_dense; ra.svh = (const half_t*) svh; ra.bias = (const half_t*) bias; ra.resid = (half_t*) resid; ra.ss_part = ss_part;
    ra.y_S = y.S; ra.has_y = (y_slabs || y_dense) ? 1 : 0; ra.m = m; ra.hidden = hidden; ra.nblk = hidden / 128; ra.tpw = tpw;
    ra.magic_nblk = gemv_magic((uint32_t) (hidden / 128));
    glue_resid_kernel<false><<<(tasks + tpw - 1) / tpw, th, 0, (hipStream_t) stream>>>(ra);
    return exl3_check_launch("glue_resid");
}

// glue_resid_moe_kernel: slabs = exl3_mgemm_indexed_act_deferred's output ([tokens * top_k slots][hidden/128][S][128] fp32, one row per slot),
// tbl_svh = the down projections' svh pointer table, indices / weights = the router's [tokens][top_k] outputs
extern "C" int exl3_glue_resid_moe(const float* slabs, int S, const void* tbl_svh, const int64_t* indices, const void* weights, int top_k, void* resid,
                                   float* ss_part, int tokens, int hidden, void* stream)
{
    EXL3_CHECK_ARG(slabs && tbl_svh && indices && weights && resid && ss_part, "glue_resid_moe: null pointer");
    EXL3_CHECK_ARG(S >= 1 && top_k >= 1 && tokens >= 1 && tokens <= 16 && hidden % 128 == 0, "glue_resid_moe: bad dimensions");
    const int tasks = tokens * (hidden / 128);
    const int th = glue_threads(tasks), tpw = th / 32;
    ResidMoeArgs a;
    memset((void*) &a, 0, sizeof(a));
    a.slabs = slabs; a.tbl_svh = (const uint64_t*) tbl_svh; a.indices = indices; a.weights = (const half_t*) weights; a.resid = (half_t*) resid; a.ss_part = ss_part;
    a.S = S; a.top_k = top_k; a.tokens = tokens; a.hidden = hidden; a.nblk = hidden / 128; a.tpw = tpw; a.magic_nblk = gemv_magic((uint32_t) (hidden / 128));
    glue_resid_moe_kernel<<<(tasks + tpw - 1) / tpw, th, 0, (hipStream_t) stream>>>(a);
    return exl3_check_launch("glue_resid_moe");
}

// glue_resid + glue_rotate in one launch (see glue_resid_kernel ROT): resid += y; ss_new = block sums of squares of the new residual;
// xh_i = had128(fp16(resid_new * w * r_prev) * suh_i) with r_prev from ss_prev (a different buffer than ss_new).  The consumers of xh_i
// must be finished with exl3_glue_qkv_rs / exl3_glue_act_rs (ss_prev, ss_new).
extern "C" int exl3_glue_resid_rotate(const float* y_slabs, int y_S, const float* y_dense, const void* svh, const void* bias, void* resid,
                                      const float* ss_prev, float* ss_new, const void* w, float eps, const void* const* suhs, void* const* xhs,
                                      float* const* xsums, int count, int m, int hidden, void* stream)
{
    EXL3_CHECK_ARG(resid && ss_prev && ss_new && ss_prev != ss_new && w && suhs && xhs, "glue_resid_rotate: null pointer / ss_prev == ss_new");
    EXL3_CHECK_ARG(m >= 1 && m <= 16 && hidden % 128 == 0, "glue_resid_rotate: bad dimensions");
    EXL3_CHECK_ARG(count >= 1 && count <= 3, "glue_resid_rotate: 1..3 consumers");
    EXL3_CHECK_ARG((y_slabs && svh && y_S >= 1) || y_dense, "glue_resid_rotate: needs a pending output (slabs + svh, or a dense tensor)");
    ResidRotate rot;
    rot.ss_prev = ss_prev; rot.w = (const half_t*) w; rot.eps = eps;
    rot.tg.count = count;
    for (int i = 0; i < 3; ++i)
    {
        rot.tg.suh[i] = i < count ? (const half_t*) suhs[i] : nullptr;
        rot.tg.xh[i] = i < count ? (half_t*) xhs[i] : nullptr;
        rot.tg.xsum[i] = (i < count && xsums) ? xsums[i] : nullptr;
        EXL3_CHECK_ARG(i >= count || (suhs[i] && xhs[i]), "glue_resid_rotate: null consumer pointer");
    }
    SlabRef y = { y_slabs, y_S };
    const int tasks = m * (hidden / 128);
    const int th = glue_threads(tasks), tpw = th / 32;
    ResidArgs ra;
    memset((void*) &ra, 0, sizeof(ra));
    ra.y_base = y.base; ra.y_dense = y_dense; ra.svh = (const half_t*) svh; ra.bias = (const half_t*) bias; ra.resid = (half_t*) resid; ra.ss_part = ss_new;
    ra.y_S = y.S; ra.has_y = 1; ra.m = m; ra.hidden = hidden; ra.nblk = hidden / 128; ra.tpw = tpw;
    ra.magic_nblk = gemv_magic((uint32_t) (hidden / 128));
    ra.rot = rot;
    glue_resid_kernel<true><<<(tasks + tpw - 1) / tpw, th, 0, (hipStream_t) stream>>>(ra);
    return exl3_check_launch("glue_resid_rotate");
}

extern "C" int exl3_glue_rotate(const void* resid, const float* ss_part, const void* w, float eps, const void* const* suhs, void* const* xhs,
                                float* const* xsums, int count, int m, int hidden, void* stream)
{
    EXL3_CHECK_ARG(resid && ss_part && w && suhs && xhs, "glue_rotate: null pointer");
    EXL3_CHECK_ARG(m >= 1 && m <= 16 && hidden % 128 == 0, "glue_rotate: bad dimensions");
    EXL3_CHECK_ARG(count >= 1 && count <= 3, "glue_rotate: between 1 and 3 consumers");
    NormTargets tg; tg.count = count;
    for (int i = 0; i < 3; ++i)
    {
        tg.suh[i] = i < count ? (const half_t*) suhs[i] : nullptr;
        tg.xh[i] = i < count ? (half_t*) xhs[i] : nullptr;
        tg.xsum[i] = (i < count && xsums) ? xsums[i] : nullptr;
        EXL3_CHECK_ARG(i >= count || (tg.suh[i] && tg.xh[i]), "glue_rotate: null consumer pointer");
    }
    const int tasks = m * (hidden / 128);
    const int th = glue_threads(tasks), tpw = th / 32;
    glue_rotate_kernel<<<(tasks + tpw - 1) / tpw, th, 0, (hipStream_t) stream>>>((const half_t*) resid, ss_part, (const half_t*) w, eps, tg, m, hidden);
    return exl3_check_launch("glue_rotate");
}

// ------------------------------------------------------------------------------------------------
// "fx" decode pipeline (round 3): the residual stream as a 64-bit fixed-point accumulator R [m][hidden] (value * 2^32) that the o_proj / down_proj
// launches add into with integer atomics (GEMV_OUT_ATOMIC) and the q|k|v / gate|up launches read (exl3_gemv_ex_fx).  These two kernels are its ends:
//   fx_init:   R = x (fp16, exact in fixed point) and ss[row][block] = the block sums of squares of x  (the arithmetic of glue_resid without a linear)
//   fx_finish: x = fp16(R / 2^32) and ss of those fp16 values: the residual handed to the final norm / lm_head (or to any fp16 consumer)
// One 32-lane half-wave per (row, 128-block), 4 values per lane.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void fx_init_kernel(const half_t* __restrict__ x, long long* __restrict__ R, float* __restrict__ ss, int m, int hidden)
{
    const int tid = threadIdx.x, l = tid & 31, hw = tid >> 5;
    const int nblk = hidden >> 7;
    const int t = blockIdx.x * 8 + hw;
    const bool act = t < m * nblk;
    const int row = act ? t / nblk : 0, blk = act ? t % nblk : 0;
    const half4_t r = ((const half4_t*) (x + (size_t) row * hidden + blk * 128))[l];
    const float r0 = (float) r.x, r1 = (float) r.y, r2 = (float) r.z, r3 = (float) r.w;
    if (act)
    {
        long long* o = R + (size_t) row * hidden + blk * 128 + 4 * l;
        o[0] = fx_from_float(r0); o[1] = fx_from_float(r1);                       // a NaN / Inf input row poisons its accumulator (exl3_gemv_args.h)
        o[2] = fx_from_float(r2); o[3] = fx_from_float(r3);
    }
    float s2 = r0 * r0;
    s2 = __builtin_fmaf(r1, r1, s2); s2 = __builtin_fmaf(r2, r2, s2); s2 = __builtin_fmaf(r3, r3, s2);
    #pragma unroll
    for (int i = 1; i < 32; i <<= 1) s2 += xor_lane(s2, i);
    if (act && l == 0) ss[(size_t) row * nblk + blk] = s2;
}

__global__ __launch_bounds__(256)
void fx_finish_kernel(const long long* __restrict__ R, half_t* __restrict__ x, float* __restrict__ ss, int m, int hidden)
{
    const int tid = threadIdx.x, l = tid & 31, hw = tid >> 5;
    const int nblk = hidden >> 7;
    const int t = blockIdx.x * 8 + hw;
    const bool act = t < m * nblk;
    const int row = act ? t / nblk : 0, blk = act ? t % nblk : 0;
    const uint4_t* fp = (const uint4_t*) (R + (size_t) row * hidden + blk * 128) + 2 * l;
    const uint4_t f0 = fp[0], f1 = fp[1];
    // the conversion of the generation-4 GEMV's fixed-point input mode (exl3_gemv4.kspec.hip): same fp16 values
    auto fx = [] (uint32_t lo, uint32_t hi) -> half_t { return f2h(fx_to_float(lo, hi)); };
    const half4_t r = { fx(f0.x, f0.y), fx(f0.z, f0.w), fx(f1.x, f1.y), fx(f1.z, f1.w) };
    if (act && x) ((half4_t*) (x + (size_t) row * hidden + blk * 128))[l] = r;
    const float r0 = (float) r.x, r1 = (float) r.y, r2 = (float) r.z, r3 = (float) r.w;
    float s2 = r0 * r0;
    s2 = __builtin_fmaf(r1, r1, s2); s2 = __builtin_fmaf(r2, r2, s2); s2 = __builtin_fmaf(r3, r3, s2);
    #pragma unroll
    for (int i = 1; i < 32; i <<= 1) s2 += xor_lane(s2, i);
    if (act && l == 0 && ss) ss[(size_t) row * nblk + blk] = s2;
}

// fx_init + qkv_prep in one launch (the two step-level set-up kernels are independent of each other): workgroups [0, fx_blocks) are fx_init's,
// the rest exl3_qkv_prep's.  One kernel boundary less per decode step.
__global__ __launch_bounds__(256)
void fx_init_prep_kernel(const half_t* __restrict__ x, long long* __restrict__ R, float* __restrict__ ss, int m, int hidden, int fx_blocks,
                         const float* __restrict__ inv_freq, const int32_t* __restrict__ positions, float attn_factor, int nfreq,
                         const int32_t* __restrict__ block_table, int blocks_per_seq, int page_size,
                         float* __restrict__ sin_out, float* __restrict__ cos_out, int64_t* __restrict__ slots)
{
    if ((int) blockIdx.x >= fx_blocks)
    {
        const int i = ((int) blockIdx.x - fx_blocks) * 256 + threadIdx.x;
        if (i >= m * 64) return;
        const int r = i >> 6, f = i & 63;
        const int pos = positions[r];
        if (f < nfreq)
        {
            float sn, cs;
            sincosf(inv_freq[f] * (float) pos, &sn, &cs);
            sin_out[i] = sn * attn_factor; cos_out[i] = cs * attn_factor;
        }
        if (f == 0 && slots && block_table) slots[r] = (int64_t) block_table[r * blocks_per_seq + pos / page_size] * page_size + (pos % page_size);
        return;
    }
    const int tid = threadIdx.x, l = tid & 31, hw = tid >> 5;
    const int nblk = hidden >> 7;
    const int t = blockIdx.x * 8 + hw;
    const bool act = t < m * nblk;
    const int row = act ? t / nblk : 0, blk = act ? t % nblk : 0;
    const half4_t r = ((const half4_t*) (x + (size_t) row * hidden + blk * 128))[l];
    const float r0 = (float) r.x, r1 = (float) r.y, r2 = (float) r.z, r3 = (float) r.w;
    if (act)
    {
        long long* o = R + (size_t) row * hidden + blk * 128 + 4 * l;
        o[0] = fx_from_float(r0); o[1] = fx_from_float(r1);                       // a NaN / Inf input row poisons its accumulator (exl3_gemv_args.h)
        o[2] = fx_from_float(r2); o[3] = fx_from_float(r3);
    }
    float s2 = r0 * r0;
    s2 = __builtin_fmaf(r1, r1, s2); s2 = __builtin_fmaf(r2, r2, s2); s2 = __builtin_fmaf(r3, r3, s2);
    #pragma unroll
    for (int i = 1; i < 32; i <<= 1) s2 += xor_lane(s2, i);
    if (act && l == 0) ss[(size_t) row * nblk + blk] = s2;
}

extern "C" int exl3_fx_init_prep(const void* x, void* R, float* ss, int m, int hidden, const float* inv_freq, const int32_t* positions, float attn_factor,
                                 int head_dim, const int32_t* block_table, int blocks_per_seq, int page_size, float* sin_out, float* cos_out,
                                 int64_t* slots, void* stream)
{
    EXL3_CHECK_ARG(x && R && ss && m >= 1 && hidden % 128 == 0, "exl3_fx_init_prep: null pointer / hidden not a multiple of 128");
    EXL3_CHECK_ARG(inv_freq && positions && sin_out && cos_out, "exl3_fx_init_prep: bad rope arguments");
    EXL3_CHECK_ARG(head_dim == 128 || head_dim == 64, "exl3_fx_init_prep: head_dim must be 128 or 64");
    EXL3_CHECK_ARG(!slots || (block_table && page_size > 0 && blocks_per_seq > 0), "exl3_fx_init_prep: slots need the block table");
    const int fx_blocks = (m * (hidden / 128) + 7) / 8, prep_blocks = (m * 64 + 255) / 256;
    fx_init_prep_kernel<<<fx_blocks + prep_blocks, 256, 0, (hipStream_t) stream>>>((const half_t*) x, (long long*) R, ss, m, hidden, fx_blocks, inv_freq, positions,
                                                                                  attn_factor, head_dim / 2, block_table, blocks_per_seq, page_size, sin_out, cos_out, slots);
    return exl3_check_launch("fx_init_prep");
}

// fx_finish + glue_rotate in one launch: one workgroup per row (32 half-waves walk the row's Hadamard blocks), x = fp16(R / 2^32) and the block sums of
// squares as exl3_fx_finish leaves them, a workgroup barrier instead of a kernel boundary, then exl3_glue_rotate's arithmetic for ONE consumer
// (the lm_head): xh = had128(fp16(x * w * rsqrt(mean(x^2) + eps)) * suh) / sqrt(128), block sums of xh.  Same values as the two launches.
__global__ __launch_bounds__(1024)
void fx_finish_rotate_kernel(const long long* __restrict__ R, half_t* __restrict__ x_out, float* __restrict__ ss_out, const half_t* __restrict__ w, float eps,
                             const half_t* __restrict__ suh, half_t* __restrict__ xh, float* __restrict__ xsum, int hidden)
{
    __shared__ float ss_s[256];
    const int tid = threadIdx.x, l = tid & 31, hw = tid >> 5, nhw = blockDim.x >> 5, row = blockIdx.x;
    const int nblk = hidden >> 7;
    auto fx = [] (uint32_t lo, uint32_t hi) -> half_t { return f2h(fx_to_float(lo, hi)); };
    // first pass: every block of the row (a half-wave takes blocks hw, hw + nhw, ...; hidden <= 4096: exactly one, kept in registers)
    half4_t r0v = { 0, 0, 0, 0 };
    for (int blk = hw; blk < nblk; blk += nhw)
    {
        const uint4_t* fp = (const uint4_t*) (R + (size_t) row * hidden + blk * 128) + 2 * l;
        const uint4_t f0 = fp[0], f1 = fp[1];
        const half4_t r = { fx(f0.x, f0.y), fx(f0.z, f0.w), fx(f1.x, f1.y), fx(f1.z, f1.w) };
        if (blk == hw) r0v = r;
        if (x_out) ((half4_t*) (x_out + (size_t) row * hidden + blk * 128))[l] = r;
        const float a0 = (float) r.x, a1 = (float) r.y, a2 = (float) r.z, a3 = (float) r.w;
        float s2 = a0 * a0;
        s2 = __builtin_fmaf(a1, a1, s2); s2 = __builtin_fmaf(a2, a2, s2); s2 = __builtin_fmaf(a3, a3, s2);
        #pragma unroll
        for (int i = 1; i < 32; i <<= 1) s2 += xor_lane(s2, i);
        if (l == 0) { ss_s[blk] = s2; if (ss_out) ss_out[(size_t) row * nblk + blk] = s2; }
    }
    __syncthreads();
    float s2 = 0.0f;
    for (int b0 = 0; b0 < nblk; b0 += 32)
    {
        float v = (b0 + l < nblk) ? ss_s[b0 + l] : 0.0f;
        #pragma unroll
        for (int i = 1; i < 32; i <<= 1) v += xor_lane(v, i);
        s2 += v;
    }
    const float rmf = __frsqrt_rn(s2 / (float) hidden + eps);
    for (int blk = hw; blk < nblk; blk += nhw)
    {
        half4_t r = r0v;
        if (blk != hw)
        {
            const uint4_t* fp = (const uint4_t*) (R + (size_t) row * hidden + blk * 128) + 2 * l;
            const uint4_t f0 = fp[0], f1 = fp[1];
            r = half4_t{ fx(f0.x, f0.y), fx(f0.z, f0.w), fx(f1.x, f1.y), fx(f1.z, f1.w) };
        }
        const half4_t wv = ((const half4_t*) (w + blk * 128))[l];
        const half4_t xn = { f2h((float) r.x * (float) wv.x * rmf), f2h((float) r.y * (float) wv.y * rmf),
                             f2h((float) r.z * (float) wv.z * rmf), f2h((float) r.w * (float) wv.w * rmf) };
        const float sum = in_had_store(xn, suh + blk * 128, xh + (size_t) row * hidden + blk * 128, l, true);
        if (l == 0 && xsum) xsum[(size_t) row * nblk + blk] = sum;
    }
}

extern "C" int exl3_fx_finish_rotate(const void* R, void* x_out, float* ss_out, const void* norm_w, float eps, const void* suh, void* xh, float* xsum,
                                     int m, int hidden, void* stream)
{
    EXL3_CHECK_ARG(R && norm_w && suh && xh && m >= 1, "exl3_fx_finish_rotate: null pointer");
    EXL3_CHECK_ARG(hidden % 128 == 0 && hidden <= 32768, "exl3_fx_finish_rotate: hidden must be a multiple of 128, at most 32768");
    const int nblk = hidden / 128;
    const int threads = 32 * (nblk < 32 ? nblk : 32);
    fx_finish_rotate_kernel<<<m, threads, 0, (hipStream_t) stream>>>((const long long*) R, (half_t*) x_out, ss_out, (const half_t*) norm_w, eps,
                                                                     (const half_t*) suh, (half_t*) xh, xsum, hidden);
    return exl3_check_launch("fx_finish_rotate");
}

// R += the finished rows of a linear whose result was NOT added by its own launch: dense fp32 rows y [m][hidden] (e.g. after a collective-library
// all-reduce of the ranks' partial sums) or deferred split-k slabs + svh (slab sum, output Hadamard, x svh in fp32: the arithmetic of the IPC all-reduce
// launch's slab route).  One half-wave per (row, 128-block); every element has one owner, so a plain read-modify-write.  The fx pipeline's
// tensor-parallel boundary without the IPC push (exl3_ar_reduce_fx is this + the exchange in one launch).
__global__ __launch_bounds__(256)
void fx_add_kernel(long long* __restrict__ R, const float* __restrict__ y, SlabRef sr, const half_t* __restrict__ svh, int m, int hidden)
{
    const int tid = threadIdx.x, l = tid & 31, hw = tid >> 5;
    const int nblk = hidden >> 7;
    const int t = blockIdx.x * 8 + hw;
    const bool act = t < m * nblk;
    const int row = act ? t / nblk : 0, blk = act ? t % nblk : 0;
    float4_t v = { 0.f, 0.f, 0.f, 0.f };
    if (sr.base)
    {
        const half4_t sc = ((const half4_t*) (svh + blk * 128))[l];
        const float4_t sm = slab_sum(sr, blk, row, m, l);
        float h0 = sm.x, h1 = sm.y, h2 = sm.z, h3 = sm.w;
        had128_f32x4(h0, h1, h2, h3, l);
        h0 *= HAD_R_SCALE_128; h1 *= HAD_R_SCALE_128; h2 *= HAD_R_SCALE_128; h3 *= HAD_R_SCALE_128;
        v = float4_t{ h0 * (float) sc.x, h1 * (float) sc.y, h2 * (float) sc.z, h3 * (float) sc.w };
    }
    else if (act) v = *((const float4_t*) (y + (size_t) row * hidden + blk * 128 + 4 * l));
    if (!act) return;
    long long* r = R + (size_t) row * hidden + blk * 128 + 4 * l;
    const float sv[4] = { v.x, v.y, v.z, v.w };
    #pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const long long old = r[i];
        const bool bad = !(__builtin_fabsf(sv[i]) < GEMV_FX_LIMIT) || (unsigned long long) (old + (1ll << 60)) > (2ull << 60);
        r[i] = bad ? (long long) GEMV_FX_POISON : old + __double2ll_rn((double) sv[i] * GEMV_FX_SCALE);
    }
}

extern "C" int exl3_fx_add(void* R, const float* y, const float* slabs, int S, const void* svh, int m, int hidden, void* stream)
{
    EXL3_CHECK_ARG(R && ((y && !slabs) || (!y && slabs && svh && S >= 1)) && m >= 1 && hidden % 128 == 0, "exl3_fx_add: give y, or slabs + svh; hidden a multiple of 128");
    SlabRef sr = { slabs, S };
    const int tasks = m * (hidden / 128);
    fx_add_kernel<<<(tasks + 7) / 8, 256, 0, (hipStream_t) stream>>>((long long*) R, y, sr, (const half_t*) svh, m, hidden);
    return exl3_check_launch("fx_add");
}

extern "C" int exl3_fx_init(const void* x, void* R, float* ss, int m, int hidden, void* stream)
{
    EXL3_CHECK_ARG(x && R && ss && m >= 1 && hidden % 128 == 0, "exl3_fx_init: null pointer / hidden not a multiple of 128");
    const int tasks = m * (hidden / 128);
    fx_init_kernel<<<(tasks + 7) / 8, 256, 0, (hipStream_t) stream>>>((const half_t*) x, (long long*) R, ss, m, hidden);
    return exl3_check_launch("fx_init");
}

extern "C" int exl3_fx_finish(const void* R, void* x, float* ss, int m, int hidden, void* stream)
{
    EXL3_CHECK_ARG(R && (x || ss) && m >= 1 && hidden % 128 == 0, "exl3_fx_finish: null pointer / hidden not a multiple of 128");
    const int tasks = m * (hidden / 128);
    fx_finish_kernel<<<(tasks + 7) / 8, 256, 0, (hipStream_t) stream>>>((const long long*) R, (half_t*) x, ss, m, hidden);
    return exl3_check_launch("fx_finish");
}
