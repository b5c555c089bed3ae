// EXL3 quantized GEMV / small-m GEMM for gfx950 (decode path):   C = ((A * suh) H) @ dequant(B) H * svh (+ bias)
//
// Host dispatcher + the split-k reduce kernels.  Replaces the reference's cooperative stream-K kernels (quant/exl3_gemv_kernel.cuh:138-402,
// quant/exl3_gemm_kernel.cuh:8-80) with a design that needs no grid sync (a grid sync costs 26 us on MI355X, MI355X_MICROARCH.md
// "barrier-cg"): grid = (n/128 column blocks) x S k-slices; a workgroup owns 128 output columns = one output Hadamard block and a k-slice
// that is a whole number of 128-wide input Hadamard blocks, so BOTH rotations are workgroup-local: the input Hadamard is recomputed per
// workgroup in the prologue from x (L2-resident) and the output Hadamard runs in the epilogue (S == 1), in the split-k reduce kernel below
// (S > 1) or in the consumer of a deferred launch (exl3_glue.hip).  The kernels themselves: exl3_gemv2.kspec.hip (1..16 rows per weight
// pass, column-pair-per-lane decode into v_mfma_f32_4x4x4_16B_f16) and exl3_gemm3.kspec.hip (5..64 rows, 16x16x32 MFMAs fed from the decoding lanes' registers).
// (The first-generation kernel of round 1 -- 4-wave workgroups, 16x16x32 MFMA on B-fragment-ordered lanes -- was an A/B baseline only and
// has been removed; git history has it.)
//
// Variants (VAR):
//   0 EXACT : B operand = the reference's fp16 weights bit-for-bit (one fp16 add / fma per weight).
//   1 FAST  : cb0/cb1 "split": weight = lo + hi (two fp16 halves of the codebook word) is fed to the MFMA as
//             two k-slots against a duplicated activation, skipping the fp16 add and all packing
//             (unrounded lo+hi: closer to exact arithmetic than the reference by <= 2^-12 relative per weight);
//             cb2 "raw": the MFMA multiplies fp16(1024 + bytesum) and the affine map k_inv * acc + k_bias * sum(x)
//             is applied once per output in the epilogue.
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include <string.h>
#include <stdlib.h>

#include "exl3_gemv_args.h"

// Split-k reduce + output Hadamard + svh (+ bias).  One half-wave per (row, column block).
__global__ __launch_bounds__(256)
void exl3_gemv_reduce_kernel(const GemvArgs a, int total_colblocks)
{
    const int l = threadIdx.x & 31, hw = threadIdx.x >> 5;
    const int m = a.m;
    const int64_t item = (int64_t) blockIdx.x * 8 + hw;            // (cbg, row)
    const bool act = item < (int64_t) total_colblocks * m;
    const int cbg = act ? (int) (item / m) : 0;
    const int row = act ? (int) (item % m) : 0;
    int mi = 0, cbl, n, ws_off;
    const half_t* svh_m; const half_t* bias_m = nullptr; void* C_m;
    float out_scale = HAD_R_SCALE_128;
    bool skip = false;
    if (a.tbl.B)
    {
        const int slot = cbg / a.tbl.cbs_per_mat;
        const SlotRef_t sr = resolve_slot(a.tbl, slot);
        skip = sr.mat_index < 0;
        cbl = cbg - slot * a.tbl.cbs_per_mat; n = a.tbl.n;
        ws_off = slot * a.tbl.cbs_per_mat * a.S * a.m * 128;
        svh_m = skip ? nullptr : (const half_t*) a.tbl.svh[sr.mat_index];
        C_m = a.c_fp32 ? (void*) ((float*) a.tbl.C + (size_t) slot * a.tbl.c_slot_stride) : (void*) ((half_t*) a.tbl.C + (size_t) slot * a.tbl.c_slot_stride);
        if (a.tbl.n_list && !skip) { n = a.tbl.n_list[sr.mat_index]; C_m = (void*) a.tbl.c_list[sr.mat_index]; skip = cbl >= (n >> 7); }
        out_scale = HAD_R_SCALE_128 * sr.weight;
    }
    else
    {
        #pragma unroll
        for (int i = 1; i < GEMV_MAX_MATS; ++i) if (i < a.num_mats && cbg >= a.mat[i].cb_first) mi = i;
        cbl = cbg - a.mat[mi].cb_first; n = a.mat[mi].n; ws_off = a.mat[mi].ws_offset;
        svh_m = a.mat[mi].svh; bias_m = a.mat[mi].bias; C_m = a.mat[mi].C;
    }
    const float* slab = a.workspace + ws_off + ((size_t) cbl * a.S) * (size_t) m * 128 + (size_t) row * 128;
    float4_t v = { 0.f, 0.f, 0.f, 0.f };
    for (int s = 0; s < a.S; ++s)
    {
        float4_t p = ((const float4_t*) (slab + (size_t) s * m * 128))[l];
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    float h0 = v.x, h1 = v.y, h2 = v.z, h3 = v.w;
    had128_f32x4(h0, h1, h2, h3, l);
    h0 *= out_scale; h1 *= out_scale; h2 *= out_scale; h3 *= out_scale;
    if (!act || skip) return;
    const half_t* svh = svh_m + cbl * 128;
    const half_t* bias = bias_m ? bias_m + cbl * 128 : nullptr;
    half4_t sc = ((const half4_t*) svh)[l];
    size_t off = ((size_t) a.c_row_offset + row) * n + cbl * 128 + 4 * l;
    if (a.c_fp32)
    {
        float4_t o = { h0 * (float) sc.x, h1 * (float) sc.y, h2 * (float) sc.z, h3 * (float) sc.w };
        if (bias) { half4_t bv = ((const half4_t*) bias)[l]; o.x += (float) bv.x; o.y += (float) bv.y; o.z += (float) bv.z; o.w += (float) bv.w; }
        *((float4_t*) ((float*) C_m + off)) = o;
    }
    else
    {
        half4_t o = { f2h(h0), f2h(h1), f2h(h2), f2h(h3) };
        o = o * sc;
        if (bias) o = o + ((const half4_t*) bias)[l];
        *((half4_t*) ((half_t*) C_m + off)) = o;
    }
}

// Final step of the weighted (MoE) mgemm: every group of `stride` consecutive slots is summed into output row-block t, in slot order
// (fp16: sequential __hadd from zero, fp32: sequential adds -- quant/exl3_gemm_kernel.cuh:241-290).  With an expert range the in-range slots were
// compacted to the front and only THEY are summed (the reference continues with bszm = the in-range count, exl3_gemm_kernel.cuh:101-127: slots this
// launch never wrote do not enter the sum, no in-range slot at all gives zeros) -- C needs no zero fill.
__global__ void mgemm_slot_reduce_kernel(void* C, int c_fp32, int num_tokens, int stride, int64_t mn, const int64_t* __restrict__ indices, int bszm,
                                         int min_index, int max_index)
{
    const int64_t col = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (min_index >= 0)
    {
        // the in-range slot count is uniform across the launch: one thread of the workgroup counts, the others read it from LDS (every
        // thread used to re-read all bszm indices)
        __shared__ int cnt_s;
        if (threadIdx.x == 0)
        {
            int cnt = 0;
            for (int i = 0; i < bszm; ++i) { const int64_t ix = indices[i]; cnt += (ix >= min_index && ix < max_index) ? 1 : 0; }
            cnt_s = cnt;
        }
        __syncthreads();
        stride = cnt_s;                                              // num_tokens == 1 with an expert range (host-checked)
    }
    if (col >= mn) return;
    for (int t = 0; t < num_tokens; ++t)
    {
        if (c_fp32)
        {
            const float* p = (const float*) C + (int64_t) t * stride * mn + col;
            float sum = 0.0f;
            for (int j = 0; j < stride; ++j) sum += p[(int64_t) j * mn];
            ((float*) C)[(int64_t) t * mn + col] = sum;
        }
        else
        {
            const half_t* p = (const half_t*) C + (int64_t) t * stride * mn + col;
            half_t sum = (half_t) 0.0f;
            for (int j = 0; j < stride; ++j) sum = sum + p[(int64_t) j * mn];
            ((half_t*) C)[(int64_t) t * mn + col] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------

static int g_gemv_variant = -1;      // -1: read EXL3_HIP_GEMV_VARIANT (default 1 = FAST)
static int g_gemv_nwv = 0;           // 0 = heuristic; otherwise cap on waves per workgroup (tuning / tests)
static int g_gemv_defer_wg_per_cu = 0;
// XCD-local tail hand-off (exl3_gemv_resid): OPT-IN.  It is only correct if workgroup i of a 1-D grid runs on XCD i % 8, which HIP does not
// promise (CU masking, harvested XCDs, other firmware): enabling it runs a probe over the grid sizes the mode is used with and refuses
// (EXL3_ERR_ARG, mode stays off) unless every workgroup reported XCC_ID == blockIdx % 8 on several launches.  Default: agent-scope hand-off.
static int g_tail_xcd_local = 0;
__global__ void exl3_xcc_probe_kernel(uint32_t* bad)
{
    if (threadIdx.x == 0)
    {
        uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        if ((x & 15u) != (blockIdx.x & 7u)) atomicAdd(bad, 1u);
    }
}
extern "C" int exl3_set_tail_xcd_local(int v)
{
    if (!v) { g_tail_xcd_local = 0; return EXL3_OK; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(0, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)
    { exl3_set_error("exl3_set_tail_xcd_local: cannot probe the workgroup -> XCD mapping during graph capture"); return EXL3_ERR_ARG; }
    uint32_t* bad = nullptr;
    EXL3_CHECK_HIP(hipMalloc(&bad, 4), "exl3_set_tail_xcd_local");
    uint32_t h = 0;
    hipError_t e = hipMemset(bad, 0, 4);
    const int grids[] = { 64, 256, 512, 1024, 2048 };
    for (int rep = 0; rep < 3 && e == hipSuccess; ++rep)
        for (int g : grids) exl3_xcc_probe_kernel<<<g, 256>>>(bad);
    if (e == hipSuccess) e = hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    (void) hipFree(bad);
    EXL3_CHECK_HIP(e, "exl3_set_tail_xcd_local");
    if (h != 0)
    {
        g_tail_xcd_local = 0;
        exl3_set_error("exl3_set_tail_xcd_local: %u probe workgroups did not run on XCD blockIdx %% 8 on this device; the XCD-local hand-off stays off", h);
        return EXL3_ERR_ARG;
    }
    g_tail_xcd_local = 1;
    return EXL3_OK;
}
static int g_gemm3_min_rows = 5;     // passes with at least this many rows take generation 3 (0 = never); raw (unrotated) input: >= 9
extern "C" int exl3_set_gemm3_min_rows(int v) { g_gemm3_min_rows = v; return EXL3_OK; }
// generation 3, 16-row passes: column blocks per workgroup (1, 2 or 4 = 4- / 8- / 16-wave workgroups sharing one activation tile); 0 = cost model
static int g_gemm3_cpw = -1;
static int gemm3_cpw() { if (g_gemm3_cpw < 0) { const char* e = getenv("EXL3_HIP_GEMM3_CPW"); const int v = e ? atoi(e) : 0; g_gemm3_cpw = (v == 1 || v == 2 || v == 4) ? v : 0; } return g_gemm3_cpw; }
extern "C" int exl3_set_gemm3_cpw(int v) { g_gemm3_cpw = (v == 1 || v == 2 || v == 4) ? v : 0; return EXL3_OK; }
static int gemm3_max_waves(int K, int mt, int rot)
{
    switch (K)
    {
        case 1: return exl3_gemm3_max_waves_k1(mt, rot); case 2: return exl3_gemm3_max_waves_k2(mt, rot); case 3: return exl3_gemm3_max_waves_k3(mt, rot);
        case 4: return exl3_gemm3_max_waves_k4(mt, rot); case 5: return exl3_gemm3_max_waves_k5(mt, rot); case 6: return exl3_gemm3_max_waves_k6(mt, rot);
        case 7: return exl3_gemm3_max_waves_k7(mt, rot); default: return exl3_gemm3_max_waves_k8(mt, rot);
    }
}
// LDS a generation-3 workgroup may use: one 16-wave workgroup per CU, two 8-wave ones, four of 4 waves (160 KB per CU, some left for the runtime)
static int gemm3_lds_budget(int cpw3) { return cpw3 == 4 ? 139264 : (cpw3 == 2 ? 73728 : 36864); }
static int gemm3_chunk_blocks(int mp, int cpw3, int bps)
{
    static const int chunk_bytes_env = [] { const char* e = getenv("EXL3_HIP_GEMM3_CHUNK_BYTES"); return e ? atoi(e) : 0; }();
    const int budget = chunk_bytes_env > 0 ? chunk_bytes_env : gemm3_lds_budget(cpw3);
    int chunk = (budget / (mp * 2) - 16) / 128;
    if (chunk > 16 * cpw3) chunk = 16 * cpw3;                // the rotated-input copy maps a chunk row onto <= 64 x waves 16-byte pieces (16 per block)
    if (chunk > 32) chunk = 32;
    if (chunk < 1) chunk = 1;
    if (chunk >= bps) return bps;
    // several chunks: every chunk but the last must be a whole number of the kernel's trips (G3_NR = 4 decode steps = 2 Hadamard blocks)
    chunk &= ~1; if (chunk < 2) chunk = 2;
    return chunk;
}
// Cost model of a generation-3 launch at <= 16 rows (us; fitted to tools/gemv_timeline.py at batch 16, round 4): `groups` workgroup columns of cpw3
// column blocks, s k-slices of b Hadamard blocks.  A wave alone on its SIMD needs ~250 ns per decode step (4 tile rows x 32 columns), w waves sharing a
// SIMD ~{250, 202, 190, 181} ns of SIMD time per step (issue-bound from 2 waves); a CU holds 4 / cpw3 workgroups; a round of workgroups pays ~1.5 us
// before it streams (launch, first rows, activation tile) and a slab per slice is written and read again.
static double gemm3_cost(int groups, int cpw3, int s, int b, int cus, bool deferred)
{
    // (w = 4 above the issue-bound figure: with four waves per SIMD the launch's first microseconds -- every workgroup staging its activation tile behind
    // everyone's first weight rows -- stretch to 3..12 us; sweep of the whole step, tools/r4_g3_sweep.sh: q|k|v 8 / gate|up 2 / down 16 slices at two
    // column blocks per workgroup 70.9 us per layer against 74.9 for the round-3 choices; 16-wave workgroups measured 10 % slower than 8-wave ones)
    static const double tw[5] = { 0.0, 0.250, 0.202, 0.195, 0.210 };
    const long wgs = (long) groups * s;
    const int cap = 4 / cpw3;
    const long rounds = (wgs + (long) cus * cap - 1) / ((long) cus * cap);
    long per_cu = (wgs + cus - 1) / cus; if (per_cu > cap) per_cu = cap;
    const int w = (int) per_cu * cpw3;
    const double t_round = 1.5 + 2.0 * b * w * tw[w] * (cpw3 == 4 ? 1.25 : 1.0);
    return rounds * t_round + 0.03 * s + ((s > 1 && !deferred) ? 5.0 : 0.0);
}

extern "C" int exl3_set_gemv_defer_wg_per_cu(int v) { g_gemv_defer_wg_per_cu = v; return EXL3_OK; }
extern "C" int exl3_set_gemv_max_waves(int v) { g_gemv_nwv = v; return EXL3_OK; }


// generation 4 (exl3_gemv4.kspec.hip) takes the 1..4-row launches it supports (plain / rotated / RMSNorm / silu-mul input, final or deferred
// output, slices of at most 32 Hadamard blocks); 0 pins generation 2 (A/B runs, the bit-identity tests between generation-2 pipelines)
static int g_gemv_gen4 = -1;
static int gemv_gen4()
{
    if (g_gemv_gen4 < 0) { const char* e = getenv("EXL3_HIP_GEMV_GEN4"); g_gemv_gen4 = e ? (atoi(e) != 0) : 1; }
    return g_gemv_gen4;
}
extern "C" int exl3_set_gemv_gen4(int v) { g_gemv_gen4 = v ? 1 : 0; return EXL3_OK; }

// fx pipeline: exl3_fx_zero_next(ptr, bytes) asks the NEXT run_mgemm call OF THIS THREAD ON THIS DEVICE to clear a buffer as a side job (the gate / up
// accumulators, cleared by the o_proj launch in front of the gate|up launch that adds into them).  The request is one-shot and never outlives that
// call: run_mgemm takes it on entry (take_fx_zero), so an argument error, a launch that is not a plain generation-4 launch, or a launch on another
// device fails loudly ("cannot clear") instead of leaving the pointer armed for an unrelated later launch.
struct FxZeroReq { void* ptr; int64_t bytes; int device; };
static thread_local FxZeroReq g_fx_zero = { nullptr, 0, -1 };
extern "C" int exl3_fx_zero_next(void* ptr, int64_t bytes)
{
    g_fx_zero = FxZeroReq{ nullptr, 0, -1 };
    EXL3_CHECK_ARG(!ptr || (bytes > 0 && bytes % 16 == 0 && (uintptr_t) ptr % 16 == 0 && bytes / 16 < (1ll << 31)), "exl3_fx_zero_next: 16-byte aligned pointer and size");
    if (!ptr) return EXL3_OK;
    int dev = -1;
    EXL3_CHECK_HIP(hipGetDevice(&dev), "exl3_fx_zero_next");
    g_fx_zero = FxZeroReq{ ptr, bytes, dev };
    return EXL3_OK;
}
static FxZeroReq take_fx_zero() { const FxZeroReq r = g_fx_zero; g_fx_zero = FxZeroReq{ nullptr, 0, -1 }; return r; }

static int gemv_variant()
{
    if (g_gemv_variant < 0)
    {
        const char* e = getenv("EXL3_HIP_GEMV_VARIANT");
        g_gemv_variant = e ? atoi(e) : 1;
        if (g_gemv_variant != 0) g_gemv_variant = 1;
    }
    return g_gemv_variant;
}

extern "C" int exl3_set_gemv_variant(int v) { g_gemv_variant = v ? 1 : 0; return EXL3_OK; }

static int choose_split(int gen, int total_colblocks, int k, int m, int num_cus, int force_split)
{
    const int nb = k / 128;
    int S;
    if (force_split > 0) S = force_split;
    else if (gen == 3)
    {
        // gen 3: 4-wave workgroups, four resident per CU (LDS / registers); fill those slots, slices of at least 4 Hadamard blocks (shorter
        // slices lose more to the reduce launch and the per-workgroup prologue than they gain: tools/bench_gemm3.py, 4096 x 4096)
        // (PMC on gate|up at 16 rows with S = 1: 224 workgroups = one wave per SIMD, VALU active 29 % of the kernel)
        S = (4 * num_cus + total_colblocks / 2) / total_colblocks;
        int maxS = nb / 4; if (maxS < 1) maxS = 1;
        if (S > maxS) S = maxS;
        const long ws_cap = (long) (EXL3_WS_REGION_BYTES / ((long) total_colblocks * m * 512));      // the slabs must fit one workspace region
        if (S > ws_cap) S = ws_cap < 1 ? 1 : (int) ws_cap;
    }
    else
    {
        // gen 2: a workgroup is up to 16 waves that split its k-slice, so one workgroup per CU already fills the CU.
        // Split k across workgroups only until every CU has one (no reduce launch when the column blocks suffice).
        if (4 * total_colblocks >= 3 * num_cus) S = 1;
        else S = (num_cus + total_colblocks - 1) / total_colblocks;
    }
    if (S < 1) S = 1;
    if (S > nb) S = nb;
    int blocks_per_slice = (nb + S - 1) / S;
    // normalise S so that no slice is empty
    S = (nb + blocks_per_slice - 1) / blocks_per_slice;
    return S;
}

// GEMV_IN_RESID operands: the producer linear's deferred slabs + svh, and where the workgroups of column block 0 publish the new residual
struct GemvResidIn { const float* slab; int S; const void* svh; void* resid_out; float* ss_out; };

static int run_mgemm(const void* A, const void* const* Bs, void* const* Cs, const void* const* suhs, const void* const* svhs,
                     const void* const* biases, const int* ns, int count, int m, int k, int K, int cb, int c_fp32,
                     int force_split, hipStream_t st, int flags = 0, const void* const* xhs = nullptr, const float* const* xsums = nullptr,
                     float** slabs_out = nullptr, int* S_out = nullptr, const GemvEpi* epi = nullptr,
                     const void* norm_w = nullptr, const float* ss_part = nullptr, float eps = 0.0f, const GemvTable* tbl = nullptr,
                     const float* act_g = nullptr, const float* act_u = nullptr, int act_S = 0, const void* act_svh_g = nullptr,
                     const void* act_svh_u = nullptr, const GemvResidIn* rsd = nullptr, const GemvRescale* act_rs = nullptr, int cpw = 0,
                     float* fx_ss_out = nullptr, const GemvAttm* attm = nullptr, const GemvQkvm* qkvm = nullptr)
{
    FxZeroReq fxz = take_fx_zero();          // one-shot: whatever happens below, the request does not survive this call
    if (rsd) flags |= GEMV_IN_RESID;
    // cpw > 0: wave-per-column-block layout (exl3_gemv2.kspec.hip): cpw column blocks of one matrix per workgroup, one wave each
    EXL3_CHECK_ARG(cpw >= 0 && cpw <= 16, "exl3_gemv_ex: column blocks per workgroup must be in [0, 16]");
    EXL3_CHECK_ARG(!rsd || cpw > 0, "exl3_gemv_ex_resid: needs the wave-per-column-block layout (cpw >= 1)");
    EXL3_CHECK_ARG(cpw == 0 || ((flags & GEMV_OUT_DEFERRED) && m <= 4 && !tbl && !epi && !(flags & GEMV_IN_ROTATED)),
                   "exl3_gemv_ex (cpw): wave-per-column-block launches are deferred, m <= 4, raw / resid / act input");
    if (act_g) flags |= GEMV_IN_ACT;
    if (attm) flags |= GEMV_IN_ATTM;
    EXL3_CHECK_ARG(!attm || (attm->part && attm->nsplit >= 1 && (attm->hd == 128 || attm->hd == 64) && attm->nsplit <= (attm->hd == 128 ? 32 : 16) && attm->gq >= 1 && attm->blocks >= 1
                             && count == 1 && m <= 4 && !tbl && !epi && cpw == 0 && !rsd && !act_g && !(flags & (GEMV_IN_ROTATED | GEMV_IN_NORM))
                             && k == attm->gq * attm->blocks * 128),
                   "exl3_gemv_ex_attm: one matrix, m <= 4, at most 32 (head_dim 128) / 16 (head_dim 64) context splits, k = heads_q x head_dim");
    if (qkvm) flags |= GEMV_IN_QKVM;
    EXL3_CHECK_ARG(!qkvm || (qkvm->sq && qkvm->sk && qkvm->sv && qkvm->S >= 1 && qkvm->svh_q && qkvm->svh_k && qkvm->svh_v && qkvm->rope_sin && qkvm->rope_cos && qkvm->slots
                             && qkvm->k_cache && qkvm->k_scales && qkvm->v_cache && qkvm->v_scales && (qkvm->hd == 64 || qkvm->hd == 128) && qkvm->kvb >= 1
                             && (qkvm->rope_mode == 1 || qkvm->rope_mode == 2) && count == 1 && m <= 4 && !tbl && !epi && cpw == 0 && !rsd && !act_g && !attm
                             && !(flags & (GEMV_IN_ROTATED | GEMV_IN_NORM))),
                   "exl3_gemv_ex_qkvm: one matrix, m <= 4, the q|k|v slabs + scales, the rope tables and cache rows of exl3_qkv_prep, a 4-bit cache");
    if (epi) flags |= GEMV_OUT_DEFERRED;
    // GEMV_OUT_ATOMIC (generation 4): no slabs, every workgroup adds its share of the output into the fixed-point accumulator Cs[i]; like a deferred
    // launch it keeps every workgroup resident (the split is chosen the same way) and it needs svhs
    const bool atomic_out = (flags & GEMV_OUT_ATOMIC) != 0, in_fx = (flags & GEMV_IN_FX) != 0;
    EXL3_CHECK_ARG(!atomic_out || (!(flags & GEMV_OUT_DEFERRED) && ((Cs && svhs) || (tbl && tbl->C && tbl->svh && tbl->slots_per_token >= 1)) && m <= 4 && !epi && cpw == 0 && !rsd),
                   "exl3_gemv_ex: GEMV_OUT_ATOMIC needs the accumulators (Cs), svhs, m <= 4 and excludes GEMV_OUT_DEFERRED");
    EXL3_CHECK_ARG(!in_fx || ((flags & GEMV_IN_NORM) && m <= 4 && fx_ss_out), "exl3_gemv_ex_fx: needs GEMV_IN_NORM, m <= 4 and ss_out");
    const bool deferred = (flags & (GEMV_OUT_DEFERRED | GEMV_OUT_ATOMIC)) != 0, rotated = (flags & GEMV_IN_ROTATED) != 0;
    EXL3_CHECK_ARG(!(deferred || rotated) || m <= 16, "exl3_gemv_ex: at most 16 rows");
    const int epi_sets = !epi ? 0 : (epi->mode == GEMV_EPI_ACT ? 2 : 1);
    EXL3_CHECK_ARG(!rotated || xhs, "exl3_gemv_ex: rotated input requires xh pointers");
    const bool in_norm = (flags & GEMV_IN_NORM) != 0;
    EXL3_CHECK_ARG(!in_norm || (norm_w && ss_part && !rotated && m <= 16), "exl3_gemv_ex: GEMV_IN_NORM needs norm_w, ss_part, raw input and m <= 16");
    EXL3_CHECK_ARG(count >= 1 && (tbl || count <= GEMV_MAX_MATS), "exl3_mgemm: between 1 and %d matrices per launch", GEMV_MAX_MATS);
    EXL3_CHECK_ARG(!tbl || (m <= 16 && !rotated && !epi), "exl3_mgemm (indexed): at most 16 rows per slot");
    EXL3_CHECK_ARG(K >= 1 && K <= 8, "exl3_gemm: K must be in [1, 8]");
    EXL3_CHECK_ARG(cb >= 0 && cb <= 2, "exl3_gemm: bad codebook");
    EXL3_CHECK_ARG(k % 128 == 0 && k > 0, "exl3_gemm: k must be divisible by 128");
    EXL3_CHECK_ARG(m >= 1, "exl3_gemm: m must be >= 1");
    const bool in_act = (flags & GEMV_IN_ACT) != 0;
    EXL3_CHECK_ARG(!in_act || (act_g && act_u && ((act_svh_g && act_svh_u && count == 1) || (tbl && tbl->act_svh)) && act_S >= 1 && !rotated && m <= 4),
                   "exl3_gemv_ex_act: needs gate / up slabs + svh, one matrix, m <= 4");
    EXL3_CHECK_ARG(A || rotated || in_act || attm || qkvm, "exl3_gemm: null A");
    EXL3_CHECK_ARG(!rsd || (in_norm && deferred && m <= 4 && rsd->slab && rsd->S >= 1 && rsd->svh && rsd->resid_out && rsd->ss_out && rsd->resid_out != A),
                   "exl3_gemv_ex_resid: needs GEMV_IN_NORM, deferred output, m <= 4, producer slabs + svh and a resid_out buffer other than resid_in");
    int total_cb = 0;
    if (tbl)
    {
        EXL3_CHECK_ARG(ns[0] % 128 == 0 && ns[0] > 0, "exl3_gemm: n must be divisible by 128");
        total_cb = count * (ns[0] / 128);
    }
    for (int i = 0; i < (tbl ? 0 : count); ++i)
    {
        EXL3_CHECK_ARG(ns[i] % 128 == 0 && ns[i] > 0, "exl3_gemm: n must be divisible by 128");
        EXL3_CHECK_ARG(Bs[i] && (deferred || (Cs && Cs[i] && svhs && svhs[i])) && (rotated || (suhs && suhs[i])), "exl3_gemm: null pointer");
        EXL3_CHECK_ARG(!atomic_out || (Cs[i] && svhs[i]), "exl3_gemv_ex: GEMV_OUT_ATOMIC needs an accumulator and svh per matrix");
        total_cb += ns[i] / 128;
    }
    Exl3DevCtx* ctx = exl3_get_ctx(st);
    if (!ctx) return EXL3_ERR_INIT;

    const int var = gemv_variant();
    // generation 3 streams the weights once per up to 64 rows (exl3_gemm3.kspec.hip); the special input / output modes stay on generation 2
    const bool g3_ok = g_gemm3_min_rows > 0 && !in_norm && !tbl && !in_act && !epi && true;
    const int pass_rows = (g3_ok && !deferred && !rotated && m > 16) ? (m > 32 ? 64 : 32) : 16;
    for (int m0 = 0; m0 < m; m0 += pass_rows)
    {
        const int mp = (m - m0) < pass_rows ? (m - m0) : pass_rows;
        GemvArgs args;
        memset((void*) &args, 0, sizeof(args));
        // measured in the fused pipeline (tools/prof_tail.py, Llama-3.1-8B shapes): generation 3 is ahead from 5 rows with rotated input; with raw
        // input its 8 half-waves per workgroup would spend longer on the input Hadamards than generation 2's 16..32, so it starts at 9 rows
        // and, for wide launches, rotates the activations ONCE first (the role of the reference's A_had temporary, "storage for input transform": quant/exl3_gemm.cu:30, 139-145)
        const bool g3 = cpw == 0 && g3_ok && mp >= (rotated ? g_gemm3_min_rows : (g_gemm3_min_rows > 9 ? g_gemm3_min_rows : 9));
        const int gen = (cpw == 0 && g3) ? 3 : 2;
        int pass_flags = flags;
        const void* pre_xh[GEMV_MAX_MATS] = { nullptr };
        const size_t xh_bytes = (size_t) mp * k * 2;
        if (g3 && !rotated && (long) total_cb * mp >= 1024 && xh_bytes * count <= (size_t) EXL3_WS_XH_BYTES)
        {
            char* xh0 = (char*) ctx->workspace + EXL3_WS_XH_OFFSET;
            for (int i = 0; i < count; ++i)
            {
                // identical suh -> one rotation serves both (gate / up of some checkpoints share it; cheap pointer test)
                int same = -1;
                for (int j = 0; j < i; ++j) if (suhs[j] == suhs[i]) { same = j; break; }
                if (same >= 0) { pre_xh[i] = pre_xh[same]; continue; }
                void* dst = xh0 + (size_t) i * xh_bytes;
                int rc = exl3_had_r_128((const half_t*) A + (size_t) m0 * k, dst, suhs[i], nullptr, 1.0f, mp, k, 0, st);
                if (rc != EXL3_OK) return rc;
                pre_xh[i] = dst;
            }
            pass_flags |= GEMV_IN_ROTATED;
        }
        // generation 3: column blocks per workgroup and the split are chosen together (16-row passes; 32- / 64-row passes keep 4-wave workgroups)
        int g3cpw = 1, g3fs = 0;
        if (g3)
        {
            const int mt_ = mp > 32 ? 4 : (mp > 16 ? 2 : 1);
            const int maxw = gemm3_max_waves(K, mt_, (pass_flags & GEMV_IN_ROTATED) ? 1 : 0);
            const int nbk = k / 128, cus = ctx->num_cus;
            int forced_S = 0;
            if (force_split > 0)
            {
                const int fsq = force_split > nbk ? nbk : (force_split > 64 ? 64 : force_split);        // (the search below stops at 64 slices)
                const int fb = (nbk + fsq - 1) / fsq; forced_S = (nbk + fb - 1) / fb;
            }
            double best = 1e30;
            // two column blocks per workgroup wherever the instantiation's register budget allows 8 waves (measured best in the whole step; one block
            // otherwise; four only on request: exl3_set_gemm3_cpw / EXL3_HIP_GEMM3_CPW) -- the split then follows the cost model for that choice
            const int c3_want = (gemm3_cpw() && 4 * gemm3_cpw() <= maxw) ? gemm3_cpw() : (maxw >= 8 ? 2 : 1);
            for (int c3 = c3_want; c3 <= c3_want; ++c3)
            {
                int groups = 0;
                for (int i = 0; i < count; ++i) groups += (ns[i] / 128 + c3 - 1) / c3;
                for (int s_ = 1; s_ <= nbk && s_ <= 64; ++s_)
                {
                    const int b = (nbk + s_ - 1) / s_;
                    if ((nbk + b - 1) / b != s_) continue;          // not a normalised split
                    if (forced_S > 0 && s_ != forced_S) continue;
                    // the slabs must fit one workspace region -- a launch that writes none (one slice, finished output) always fits (with
                    // every candidate rejected the split stayed 0 and the slice length below divided by it: lm_head-sized n at 33..64 rows)
                    if ((s_ > 1 || deferred) && (long) total_cb * s_ * mp * 512 > (long) EXL3_WS_REGION_BYTES) continue;
                    const double c = gemm3_cost(groups, c3, s_, b, cus, deferred);
                    if (c < best) { best = c; g3cpw = c3; g3fs = s_; }
                }
            }
            if (g3fs < 1) { g3fs = 1; g3cpw = c3_want; }            // (forced split that is not normalised / nothing fits: one slice)
        }
        args.act_g = act_g; args.act_u = act_u; args.act_S = act_S;
        args.act_svh_g = (const half_t*) act_svh_g; args.act_svh_u = (const half_t*) act_svh_u;
        args.norm_w = (const half_t*) norm_w; args.ss_part = ss_part; args.eps = eps;
        if (in_fx) args.rs_ss_out = fx_ss_out;
        if (rsd) { args.rs_slab = rsd->slab; args.rs_S = rsd->S; args.rs_svh = (const half_t*) rsd->svh; args.rs_resid_out = (half_t*) rsd->resid_out; args.rs_ss_out = rsd->ss_out; }
        if (act_rs) args.act_rs = *act_rs;
        if (attm) { args.attm = *attm; args.attm.magic_gq = gemv_magic((uint32_t) attm->gq); }
        if (qkvm) args.qkvm = *qkvm;
        int fs = force_split;
        if (g3) fs = g3fs;
        else if (deferred && fs == 0)
        {
            // deferred epilogue: the reduction is free (a glue kernel / tail epilogue does it), so pick the split that balances the
            // chip.  All workgroups of these launches are resident at once and the kernel is VALU-bound per CU, so the launch
            // takes as long as the busiest CU: ceil(workgroups / CUs) * blocks per workgroup.  Fewest slabs among the minima;
            // a light preference for >= 2 workgroups per CU (tools/sweep_split.py: gate/up 3 -> 8 is -3 us per layer).
            const int nbk = k / 128, cus = ctx->num_cus;
            double best_cost = 1e30;
            fs = 1;
            for (int s = 1; s <= nbk && s <= 64; ++s)
            {
                const int b = (nbk + s - 1) / s;
                if ((nbk + b - 1) / b != s) continue;              // not a normalised split
                const long wg = (long) total_cb * s;
                const double cost = (double) ((wg + cus - 1) / cus) * b + (wg < 2l * cus ? 0.25 : 0.0) + 1e-3 * s;
                if (cost < best_cost) { best_cost = cost; fs = s; }
            }
            if (g_gemv_defer_wg_per_cu > 0) fs = (g_gemv_defer_wg_per_cu * ctx->num_cus + total_cb - 1) / total_cb;
            if (fs < 1) fs = 1;
            if (epi && fs > 64) fs = 64;                         // tail epilogue: one pass of slabs must fit the workgroup LDS
        }
        int total_groups = 0;
        if (cpw > 0)
        {
            for (int i = 0; i < count; ++i) total_groups += (ns[i] / 128 + cpw - 1) / cpw;
            if (force_split == 0)
            {
                // default: about 16 waves per CU in all (4 per SIMD at this layout's 128 VGPRs); tuned values come from the caller
                const int nbk = k / 128;
                long want = ((long) ctx->num_cus * 16 + (long) total_groups * cpw - 1) / ((long) total_groups * cpw);
                if (want < 1) want = 1;
                if (want > nbk) want = nbk;
                fs = (int) want;
            }
            else fs = force_split;
        }
        const int S = g3 ? g3fs : choose_split(gen, total_cb, k, mp, ctx->num_cus, fs);
        EXL3_CHECK_ARG(S >= 1, "exl3_gemm: internal: split factor %d", S);
        const int nb = k / 128;
        const int bps = (nb + S - 1) / S;
        int cbf = 0; int64_t wso = 0;
        if (tbl) { args.tbl = *tbl; wso = (int64_t) total_cb * S * mp * 128; }
        for (int i = 0; i < (tbl ? 0 : count); ++i)
        {
            args.mat[i].B = (const uint32_t*) Bs[i];
            args.mat[i].suh = suhs ? (const half_t*) suhs[i] : nullptr;
            args.mat[i].svh = svhs ? (const half_t*) svhs[i] : nullptr;
            args.mat[i].bias = biases ? (const half_t*) biases[i] : nullptr;
            args.mat[i].C = Cs ? Cs[i] : nullptr;
            args.mat[i].xh = pre_xh[i] ? (const half_t*) pre_xh[i] : (xhs ? (const half_t*) xhs[i] : nullptr);
            args.mat[i].xsum = xsums ? xsums[i] : nullptr;
            args.mat[i].n = ns[i];
            args.mat[i].cb_first = cbf;
            args.mat[i].ws_offset = (int) wso;
            cbf += cpw > 0 ? (ns[i] / 128 + cpw - 1) / cpw : (g3 ? (ns[i] / 128 + g3cpw - 1) / g3cpw : ns[i] / 128);
            wso += (int64_t) (ns[i] / 128) * S * mp * 128;
        }
        if (epi)
        {
            args.epi = *epi;
            args.epi.tickets = ctx->tickets;
            args.epi.ticket_global = total_cb;
            args.epi.ss_offset = (int) wso;
            if (epi->mode == GEMV_EPI_NORM) wso += (int64_t) mp * (ns[0] / 128);
            // one XCD per column block needs (column blocks x S) / 8 consecutive logical workgroups to hold whole column blocks
            args.epi.xcd_local = (epi->mode == GEMV_EPI_RESID && g_tail_xcd_local && total_cb % 8 == 0) ? 1 : 0;
            EXL3_CHECK_ARG(total_cb + 1 <= EXL3_NUM_TICKETS, "exl3_gemv: too many column blocks for the ticket table");
            EXL3_CHECK_ARG(S <= 128, "exl3_gemv: split too deep for the tail epilogue");
        }
        EXL3_CHECK_ARG((S == 1 && !deferred) || atomic_out || wso * 4 <= EXL3_WS_REGION_BYTES, "exl3_gemm: split-k workspace too small");
        float* ws_region = ctx->workspace;
        if ((S > 1 || deferred) && !atomic_out)
        {
            ws_region += ctx->ws_toggle ? EXL3_WS_REGION_BYTES / 4 : 0;
            ctx->ws_toggle ^= 1;
            if (slabs_out) for (int i = 0; i < (tbl ? 0 : count); ++i) slabs_out[i] = ws_region + args.mat[i].ws_offset;
            if (slabs_out && tbl) slabs_out[0] = ws_region;                // indexed form: [slot][column block][S][m][128]
        }
        if (S_out) *S_out = S;
        args.flags = pass_flags;
        args.A = (const half_t*) A + (size_t) m0 * k;
        args.workspace = ws_region;
        {
            // diagnostics builds (-DG2_TIMING / -DG4_TIMING): phase stamps at 48 MiB of the workspace; EXL3_HIP_TIMING_SLOTS=n (<= 8) rotates n 512-KB slots
            // over successive launches so that a whole captured step keeps the stamps of its last n GEMV launches (slots 4..7 overlap the
            // pre-rotation region at 50 MiB: diagnostics only)
            static const int slots = [] { const char* e = getenv("EXL3_HIP_TIMING_SLOTS"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
            static unsigned launch_no = 0;
            args.ws_debug = ctx->workspace + (48ll << 20) / 4 + (size_t) (slots > 1 ? (launch_no++ % (unsigned) slots) : 0) * (512ll << 10) / 4;
        }
        args.num_mats = tbl ? 1 : count;
        args.m = mp;
        args.k = k;
        args.S = S;
        args.kslice = bps * 128;
        args.c_fp32 = c_fp32;
        args.c_row_offset = m0;

        args.cpw = cpw;
        dim3 grid((unsigned) ((cpw > 0 ? total_groups : total_cb) * S));
        EXL3_CHECK_ARG(!(fxz.ptr && gen == 3), "exl3_fx_zero_next: the launch that followed cannot clear the buffer (not a generation-4 launch: rows > 4, a slice of more than 32 Hadamard blocks, a tail / wave-per-column-block mode, or generation 4 switched off)");
        if (gen == 3)
        {
            const int mt = mp > 32 ? 4 : (mp > 16 ? 2 : 1);
            // workgroup = 4 g3cpw waves = g3cpw adjacent column blocks of one matrix sharing ONE activation tile in LDS (16 rows keep a whole 8-block
            // slice resident at 4 waves, a whole k = 4096 at 16: one chunk, no second prologue / barrier inside the stream)
            const int nwt = 4 * g3cpw;
            const int chunk = gemm3_chunk_blocks(mp, g3cpw, bps);
            args.chunk_blocks = chunk;
            const size_t lds = exl3_gemm3_lds_bytes(mp, chunk, nwt);
            int total_groups3 = 0;
            for (int i = 0; i < count; ++i) total_groups3 += (ns[i] / 128 + g3cpw - 1) / g3cpw;
            EXL3_CHECK_ARG(total_groups3 <= 65535, "exl3_gemm: too many column blocks for one launch");
            grid = dim3((unsigned) S, (unsigned) total_groups3);              // (k-slices, groups of g3cpw column blocks), as for generation 2 below
            for (int i = 1; i < GEMV_MAX_MATS; ++i) args.cbf[i - 1] = args.mat[i].cb_first;
            args.magic_m = gemv_magic((uint32_t) mp);
            switch (K)
            {
                case 1: exl3_gemm3_launch_k1(cb, mt, var, nwt, grid, lds, st, args); break;
                case 2: exl3_gemm3_launch_k2(cb, mt, var, nwt, grid, lds, st, args); break;
                case 3: exl3_gemm3_launch_k3(cb, mt, var, nwt, grid, lds, st, args); break;
                case 4: exl3_gemm3_launch_k4(cb, mt, var, nwt, grid, lds, st, args); break;
                case 5: exl3_gemm3_launch_k5(cb, mt, var, nwt, grid, lds, st, args); break;
                case 6: exl3_gemm3_launch_k6(cb, mt, var, nwt, grid, lds, st, args); break;
                case 7: exl3_gemm3_launch_k7(cb, mt, var, nwt, grid, lds, st, args); break;
                case 8: exl3_gemm3_launch_k8(cb, mt, var, nwt, grid, lds, st, args); break;
            }
        }
        else
        {
            const int ng = mp <= 4 ? 1 : (mp <= 8 ? 2 : 4);
            const bool rot_pass = (pass_flags & GEMV_IN_ROTATED) != 0;
            // table launches (MoE): raw-x / slab-act inputs; the fp16 gate / up act input (tbl->act_u) stays generation 2's
            bool g4 = gemv_gen4() && ng == 1 && (!tbl || (!tbl->act_u && !tbl->n_list)) && !epi && cpw == 0 && !rsd && bps <= 32;
            if (g4 && rot_pass && var == 1 && cb == 2)
                for (int i = 0; i < count; ++i) if (!args.mat[i].xsum) g4 = false;      // the mul1 FAST variant needs the producer's block sums
            EXL3_CHECK_ARG(g4 || !fxz.ptr, "exl3_fx_zero_next: the launch that followed cannot clear the buffer (not a generation-4 launch: rows > 4, a slice of more than 32 Hadamard blocks, a tail / wave-per-column-block mode, or generation 4 switched off)");
            if (g4)
            {
                const int mode = qkvm ? 9 : attm ? 8 : tbl ? (in_act ? 7 : 6) : in_act ? ((flags & GEMV_IN_ACTFX) ? 5 : 3) : (in_norm ? (in_fx ? 4 : 2) : (rot_pass ? 0 : 1));
                if (fxz.ptr)
                {
                    int dev_now = -1;
                    EXL3_CHECK_HIP(hipGetDevice(&dev_now), "exl3_gemv_ex");
                    EXL3_CHECK_ARG(!tbl && dev_now == fxz.device, "exl3_fx_zero_next: the launch that follows must be a plain (non-table) generation-4 launch on the device the request was made on");
                    args.fx_zero = fxz.ptr; args.fx_zero_n16 = (int) (fxz.bytes / 16); fxz.ptr = nullptr;
                }
                const int units4 = bps * 4;
                int nwv4 = 8;
                if (nwv4 > (bps > 4 ? bps : 4)) nwv4 = bps > 4 ? bps : 4;
                if (deferred && nwv4 > 4 && bps <= 16) nwv4 = 4;                       // (generation 2's measured choices, below)
                if (g_gemv_nwv < 0) nwv4 = -g_gemv_nwv < 8 ? -g_gemv_nwv : 8;
                if (nwv4 > units4) nwv4 = units4;
                if (g_gemv_nwv > 0 && nwv4 > g_gemv_nwv) nwv4 = g_gemv_nwv;
                if (nwv4 < 1) nwv4 = 1;
                const size_t lds4 = exl3_gemv4_lds_bytes(mode, nwv4, mp, bps);
                EXL3_CHECK_ARG(grid.x / (unsigned) S <= 65535u, "exl3_gemm: too many column blocks for one launch");
                grid = dim3((unsigned) S, grid.x / (unsigned) S);
                if (tbl)
                {
                    EXL3_CHECK_ARG(count <= 65535, "exl3_mgemm: too many slots for one launch");
                    grid = dim3((unsigned) S, (unsigned) (ns[0] / 128), (unsigned) count);       // (k-slices, column blocks of one matrix, slots)
                }
                for (int i = 1; i < GEMV_MAX_MATS; ++i) args.cbf[i - 1] = args.mat[i].cb_first;
                args.nwv = nwv4;
                args.magic_m = gemv_magic((uint32_t) mp); args.magic_nwv = gemv_magic((uint32_t) nwv4); args.magic_nhw = gemv_magic((uint32_t) (2 * nwv4));
                switch (K)
                {
                    case 1: exl3_gemv4_launch_k1(cb, var, mode, nwv4, grid, lds4, st, args); break;
                    case 2: exl3_gemv4_launch_k2(cb, var, mode, nwv4, grid, lds4, st, args); break;
                    case 3: exl3_gemv4_launch_k3(cb, var, mode, nwv4, grid, lds4, st, args); break;
                    case 4: exl3_gemv4_launch_k4(cb, var, mode, nwv4, grid, lds4, st, args); break;
                    case 5: exl3_gemv4_launch_k5(cb, var, mode, nwv4, grid, lds4, st, args); break;
                    case 6: exl3_gemv4_launch_k6(cb, var, mode, nwv4, grid, lds4, st, args); break;
                    case 7: exl3_gemv4_launch_k7(cb, var, mode, nwv4, grid, lds4, st, args); break;
                    case 8: exl3_gemv4_launch_k8(cb, var, mode, nwv4, grid, lds4, st, args); break;
                }
            }
            else
            {
            EXL3_CHECK_ARG(!atomic_out && !in_fx && !attm && !qkvm && !(flags & GEMV_IN_ACTFX) && !(tbl && in_act), "exl3_gemv_ex: GEMV_OUT_ATOMIC / GEMV_IN_FX / GEMV_IN_ACTFX / GEMV_IN_ATTM / GEMV_IN_QKVM / slab-act table launches are generation-4 launches (m <= 4, slices of <= 32 Hadamard blocks, generation 4 enabled)");
            int nwv = 16 / ng;                                   // partial-sum LDS: nwv * 4*ng rows * 512 B <= 32 KB
            const int units = bps * (8 / G2_PF);                 // the waves split the slice's tile rows in units of G2_PF
            // measured on MI355X (tools/prof_tail.py, batch 1): one wave per Hadamard block of the slice, but at least 4 waves --
            // more waves than that do not help (the kernel is VALU / fixed-latency bound), fewer starve the short slices
            if (nwv > (bps > 4 ? bps : 4)) nwv = bps > 4 ? bps : 4;
            // deferred launches keep every workgroup resident (balanced split above): 4 waves = one per SIMD per workgroup with equal
            // rows each, so no SIMD ends up with an extra wave (tools/gemv_timeline.py: down_proj's 7-wave workgroups left a
            // 4/4/3/3 SIMD load and a 1.8 us wait at the workgroup barrier)
            if (deferred && nwv > 4 && bps <= 16) nwv = 4;
            // variants above 64 VGPRs (3INST/MCG decode, NORM prep, K >= 5 rings) fit 7 or 6 waves per SIMD: three 8-wave workgroups per
            // CU instead of one 16-wave workgroup (tools/bench_lmhead.py: lm_head 3INST 105 -> 87 us, NORM 92 -> 81 us)
            if (!deferred && ng == 1 && (cb != 2 || tbl || K >= 5) && nwv > 8) nwv = 8;
            if (in_act && nwv > 4) nwv = 4;                      // (classic layout) ACT-mode kernels are built for 4-wave workgroups (256 VGPRs available)
            if (g_gemv_nwv < 0) nwv = (-g_gemv_nwv < 16 / ng) ? -g_gemv_nwv : 16 / ng;      // tuning: forced wave count
            if (in_act && nwv > 4) nwv = 4;
            if (nwv > units) nwv = units;
            if (g_gemv_nwv > 0 && nwv > g_gemv_nwv) nwv = g_gemv_nwv;
            if (nwv < 1) nwv = 1;
            if (cpw > 0) nwv = cpw;                              // one wave per column block of the group
            // activation fragments of one chunk of Hadamard blocks, built once per workgroup: up to ~48 KB, the whole slice when it fits
            const int AHh = (var == 1 && cb != 2) ? 32 : 16;
            int chunk = (int) ((size_t) 49152 / ((size_t) 8 * mp * AHh * 2));
            if (chunk < 1) chunk = 1;
            if (chunk > bps) chunk = bps;
            EXL3_CHECK_ARG(cpw == 0 || chunk >= bps, "exl3_gemv_ex (cpw): the slice's activation fragments must fit one LDS chunk (use a deeper split)");
            args.chunk_blocks = chunk;
            size_t lds = exl3_gemv2_lds_bytes(ng, var, cb, nwv, mp, chunk);
            if (epi)
            {
                // tail epilogue: the last workgroup of a column block stages rows_per_pass rows x sets x S slab lines (512 B) in LDS
                const size_t line_bytes = (size_t) epi_sets * S * 512;
                if (lds < line_bytes) lds = line_bytes;
                int rpp = (int) (lds / line_bytes);
                if (rpp > mp) rpp = mp;
                if (rpp > 2 * nwv) rpp = 2 * nwv;
                args.epi.rows_per_pass = rpp;
            }
            // 2-D grid (k-slices, column blocks or groups): dispatched x-fastest, i.e. in the order column block * S + slice; the kernel reads
            // its pair from blockIdx instead of dividing a linear id (exl3_gemv2.kspec.hip prologue)
            EXL3_CHECK_ARG(grid.x / (unsigned) S <= 65535u, "exl3_gemm: too many column blocks for one launch");
            grid = dim3((unsigned) S, grid.x / (unsigned) S);
            for (int i = 1; i < GEMV_MAX_MATS; ++i) args.cbf[i - 1] = args.mat[i].cb_first;
            args.nwv = nwv;
            args.magic_m = gemv_magic((uint32_t) mp); args.magic_nwv = gemv_magic((uint32_t) nwv); args.magic_nhw = gemv_magic((uint32_t) (2 * nwv));
            switch (K)
            {
                case 1: exl3_gemv2_launch_k1(cb, var, ng, nwv, grid, lds, st, args); break;
                case 2: exl3_gemv2_launch_k2(cb, var, ng, nwv, grid, lds, st, args); break;
                case 3: exl3_gemv2_launch_k3(cb, var, ng, nwv, grid, lds, st, args); break;
                case 4: exl3_gemv2_launch_k4(cb, var, ng, nwv, grid, lds, st, args); break;
                case 5: exl3_gemv2_launch_k5(cb, var, ng, nwv, grid, lds, st, args); break;
                case 6: exl3_gemv2_launch_k6(cb, var, ng, nwv, grid, lds, st, args); break;
                case 7: exl3_gemv2_launch_k7(cb, var, ng, nwv, grid, lds, st, args); break;
                case 8: exl3_gemv2_launch_k8(cb, var, ng, nwv, grid, lds, st, args); break;
            }
            }
        }
        int rc = exl3_check_launch("exl3_gemv");
        if (rc) return rc;
        if (S > 1 && !deferred)
        {
            // (GEMV_OUT_ATOMIC launches count as deferred: nothing to reduce)
            // the reduce kernel walks COLUMN BLOCKS: a launch whose grid counted groups of column blocks (generation 3 with 8- / 16-wave workgroups, the
            // wave-per-column-block layout) hands it the matrices' first column blocks, not their first groups
            if (!tbl) { int cbc = 0; for (int i = 0; i < count; ++i) { args.mat[i].cb_first = cbc; cbc += ns[i] / 128; } }
            int64_t items = (int64_t) total_cb * mp;
            exl3_gemv_reduce_kernel<<<dim3((unsigned) ((items + 7) / 8)), dim3(256), 0, st>>>(args, total_cb);
            rc = exl3_check_launch("exl3_gemv_reduce");
            if (rc) return rc;
        }
    }
    return 1;
}

extern "C" int exl3_gemm(const void* A, const void* B, void* C, const void* suh, const void* svh, const void* bias,
                         int m, int k, int n, int K, int cb, int c_fp32, int force_split, void* stream)
{
    const void* Bs[1] = { B }; void* Cs[1] = { C }; const void* su[1] = { suh }; const void* sv[1] = { svh };
    const void* bi[1] = { bias }; int ns[1] = { n };
    return run_mgemm(A, Bs, Cs, su, sv, bi, ns, 1, m, k, K, cb, c_fp32, force_split, (hipStream_t) stream);
}

extern "C" int exl3_mgemm(const void* A, const void* const* Bs, void* const* Cs, const void* const* suhs, const void* const* svhs,
                          const int* ns, int count, int m, int k, int K, int cb, int c_fp32, int force_split, void* stream)
{
    EXL3_CHECK_ARG(Bs && Cs && suhs && svhs && ns, "exl3_mgemm: null table");
    return run_mgemm(A, Bs, Cs, suhs, svhs, nullptr, ns, count, m, k, K, cb, c_fp32, force_split, (hipStream_t) stream);
}

// Extended launch used by the fused decode pipeline (exl3_glue.hip): optional pre-rotated inputs per matrix and/or deferred
// epilogue (raw fp32 partial slabs left in the per-device workspace for a glue kernel).  slabs_out[i] receives the device
// pointer of matrix i's slabs [n_i/128][S][m][128]; *S_out the split count.
extern "C" int exl3_gemv_ex(const void* A, const void* const* xhs, const float* const* xsums, const void* const* Bs, void* const* Cs,
                            const void* const* suhs, const void* const* svhs, const void* const* biases, const int* ns, int count,
                            int m, int k, int K, int cb, int c_fp32, int flags, int force_split, float** slabs_out, int* S_out, void* stream)
{
    EXL3_CHECK_ARG(Bs && ns, "exl3_gemv_ex: null table");
    EXL3_CHECK_ARG(!(flags & GEMV_IN_NORM), "exl3_gemv_ex: use exl3_gemv_ex_norm for GEMV_IN_NORM");
    return run_mgemm(A, Bs, Cs, suhs, svhs, biases, ns, count, m, k, K, cb, c_fp32, force_split, (hipStream_t) stream, flags, xhs, xsums, slabs_out, S_out);
}

// exl3_gemv_ex whose input A is the fp16 residual stream: RMSNorm (norm_w, eps; per-block sums of squares ss_part [m][k/128] from
// exl3_glue_resid) is applied while the activation fragments are built.  flags: GEMV_OUT_DEFERRED optional.
extern "C" int exl3_gemv_ex_norm(const void* resid, const void* norm_w, const float* ss_part, float eps, const void* const* Bs, void* const* Cs,
                                 const void* const* suhs, const void* const* svhs, const void* const* biases, const int* ns, int count,
                                 int m, int k, int K, int cb, int c_fp32, int flags, int force_split, float** slabs_out, int* S_out, void* stream)
{
    EXL3_CHECK_ARG(Bs && ns && resid, "exl3_gemv_ex_norm: null table");
    return run_mgemm(resid, Bs, Cs, suhs, svhs, biases, ns, count, m, k, K, cb, c_fp32, force_split, (hipStream_t) stream,
                     (flags & GEMV_OUT_DEFERRED) | GEMV_IN_NORM, nullptr, nullptr, slabs_out, S_out, nullptr, norm_w, ss_part, eps);
}

// ------------------------------------------------------------------------------------------------
// GEMV launches with an in-kernel tail epilogue (exl3_gemv2_tail.cuh): a whole sublayer boundary of the decode step per launch.
// Inputs are either pre-rotated (xhs / xsums from the producing tail or glue kernel) or raw (A + suhs: the kernel rotates).
// ------------------------------------------------------------------------------------------------
extern "C" int exl3_gemv_norm(const void* A, const void* xh, const float* xsum, const void* B, const void* suh, const void* svh, const void* bias,
                              int m, int k, int n, int K, int cb, void* resid, const void* norm_w, float eps,
                              const void* const* t_suhs, void* const* t_xhs, float* const* t_xsums, int t_count, void* xn_out, void* stream)
{
    EXL3_CHECK_ARG(B && svh && resid && norm_w, "exl3_gemv_norm: null pointer");
    EXL3_CHECK_ARG(m >= 1 && m <= 16, "exl3_gemv_norm: 1 <= m <= 16");
    EXL3_CHECK_ARG(t_count >= 0 && t_count <= 3, "exl3_gemv_norm: at most 3 consumers");
    GemvEpi e; memset((void*) &e, 0, sizeof(e));
    e.mode = GEMV_EPI_NORM;
    e.resid = (half_t*) resid; e.norm_w = (const half_t*) norm_w; e.eps = eps; e.xn_out = (half_t*) xn_out; e.t_count = t_count;
    for (int i = 0; i < t_count; ++i)
    {
        EXL3_CHECK_ARG(t_suhs && t_xhs && t_suhs[i] && t_xhs[i], "exl3_gemv_norm: null consumer pointer");
        e.t_suh[i] = (const half_t*) t_suhs[i]; e.t_xh[i] = (half_t*) t_xhs[i]; e.t_xsum[i] = t_xsums ? t_xsums[i] : nullptr;
    }
    const void* Bs[1] = { B }; const void* su[1] = { suh }; const void* sv[1] = { svh }; const void* bi[1] = { bias };
    const void* xhs[1] = { xh }; const float* xss[1] = { xsum }; int ns[1] = { n };
    return run_mgemm(A, Bs, nullptr, su, sv, bi, ns, 1, m, k, K, cb, 1, 0, (hipStream_t) stream, xh ? GEMV_IN_ROTATED : 0,
                     xh ? xhs : nullptr, xh ? xss : nullptr, nullptr, nullptr, &e);
}

// o_proj / down_proj with glue_resid inside the launch: resid (fp16, in place) += linear(x), ss_out[m][n/128] = per-block sums of squares of the
// new residual (what the consumer's GEMV_IN_NORM needs).  The workgroup that finishes a column block last does it; with n/128 % 8 == 0 all
// slices of a column block are placed on one XCD and the hand-off stays in its L2.
extern "C" int exl3_gemv_resid(const void* A, const void* xh, const float* xsum, const void* B, const void* suh, const void* svh, const void* bias,
                               int m, int k, int n, int K, int cb, void* resid, float* ss_out, int force_split, void* stream)
{
    EXL3_CHECK_ARG(B && svh && resid && ss_out, "exl3_gemv_resid: null pointer");
    EXL3_CHECK_ARG(m >= 1 && m <= 16, "exl3_gemv_resid: 1 <= m <= 16");
    GemvEpi e; memset((void*) &e, 0, sizeof(e));
    e.mode = GEMV_EPI_RESID;
    e.resid = (half_t*) resid; e.ss_out = ss_out;
    const void* Bs[1] = { B }; const void* su[1] = { suh }; const void* sv[1] = { svh }; const void* bi[1] = { bias };
    const void* xhs[1] = { xh }; const float* xss[1] = { xsum }; int ns[1] = { n };
    return run_mgemm(A, Bs, nullptr, su, sv, bi, ns, 1, m, k, K, cb, 1, force_split, (hipStream_t) stream, xh ? GEMV_IN_ROTATED : 0,
                     xh ? xhs : nullptr, xh ? xss : nullptr, nullptr, nullptr, &e);
}

extern "C" int exl3_gemv_act(const void* A, const void* const* xhs, const float* const* xsums, const void* const* Bs, const void* const* suhs,
                             const void* const* svhs, int m, int k, int inter, int K, int cb,
                             const void* suh_d, void* xh_d, float* xsum_d, void* a_out, void* stream)
{
    EXL3_CHECK_ARG(Bs && svhs && svhs[0] && svhs[1] && suh_d && xh_d, "exl3_gemv_act: null pointer");
    EXL3_CHECK_ARG(m >= 1 && m <= 16, "exl3_gemv_act: 1 <= m <= 16");
    GemvEpi e; memset((void*) &e, 0, sizeof(e));
    e.mode = GEMV_EPI_ACT;
    e.t_count = 1; e.t_suh[0] = (const half_t*) suh_d; e.t_xh[0] = (half_t*) xh_d; e.t_xsum[0] = xsum_d; e.a_out = (half_t*) a_out;
    int ns[2] = { inter, inter };
    return run_mgemm(A, Bs, nullptr, suhs, svhs, nullptr, ns, 2, m, k, K, cb, 0, 0, (hipStream_t) stream, xhs ? GEMV_IN_ROTATED : 0,
                     xhs, xsums, nullptr, nullptr, &e);
}

extern "C" int exl3_gemv_qkv(const void* A, const void* const* xhs, const float* const* xsums, const void* const* Bs, const void* const* suhs,
                             const void* const* svhs, int m, int k, int K, int cb, void* q_out, void* k_out, void* v_out,
                             const float* rope_sin, const float* rope_cos, const int32_t* positions,
                             void* k_cache, void* k_scales, void* v_cache, void* v_scales, const int32_t* block_table, int blocks_per_seq,
                             int page_size, int k_bits, int v_bits, int heads_q, int heads_kv, int head_dim, int rope_mode, void* stream)
{
    EXL3_CHECK_ARG(Bs && svhs && svhs[0] && svhs[1] && svhs[2] && q_out && rope_sin && rope_cos && positions, "exl3_gemv_qkv: null pointer");
    EXL3_CHECK_ARG(head_dim == 128, "exl3_gemv_qkv: head_dim must be 128 (one Hadamard block per head)");
    EXL3_CHECK_ARG(m >= 1 && m <= 16, "exl3_gemv_qkv: 1 <= m <= 16");
    EXL3_CHECK_ARG(rope_mode == 1 || rope_mode == 2, "exl3_gemv_qkv: rope_mode must be 1 (GPTJ) or 2 (NEOX)");
    EXL3_CHECK_ARG(!k_cache || (k_scales && v_cache && v_scales && block_table && page_size > 0), "exl3_gemv_qkv: incomplete cache arguments");
    EXL3_CHECK_ARG(!k_cache || (k_bits >= 2 && k_bits <= 8 && v_bits >= 2 && v_bits <= 8), "exl3_gemv_qkv: cache bits must be in [2, 8]");
    GemvEpi e; memset((void*) &e, 0, sizeof(e));
    e.mode = GEMV_EPI_QKV;
    e.q_out = (half_t*) q_out; e.k_out = (half_t*) k_out; e.v_out = (half_t*) v_out;
    e.rope_sin = rope_sin; e.rope_cos = rope_cos; e.positions = positions;
    e.k_cache = (uint32_t*) k_cache; e.k_scales = (half_t*) k_scales; e.v_cache = (uint32_t*) v_cache; e.v_scales = (half_t*) v_scales;
    e.block_table = block_table; e.blocks_per_seq = blocks_per_seq; e.page_size = page_size > 0 ? page_size : 256;
    e.k_bits = k_bits; e.v_bits = v_bits; e.hq = heads_q; e.hkv = heads_kv; e.rope_mode = rope_mode;
    int ns[3] = { heads_q * 128, heads_kv * 128, heads_kv * 128 };
    return run_mgemm(A, Bs, nullptr, suhs, svhs, nullptr, ns, 3, m, k, K, cb, 0, 0, (hipStream_t) stream, xhs ? GEMV_IN_ROTATED : 0,
                     xhs, xsums, nullptr, nullptr, &e);
}

// Indexed / weighted multi-matrix launch (MoE): exl3_mgemm with pointer tables (quant/exl3_gemm.cuh:58-78, kernel exl3_gemm_kernel.cuh:88-292).
//   slot j of bszm: matrix = indices ? indices[j] : j (device int64); with an expert range, in-range entries are compacted to the front and
//   re-based to min_index, the remaining slots are skipped; A [bszm_in][m][k] (bszm_in == 1: shared), C [bszm][m][n];
//   weights (fp16 [bszm]): each slot's output is scaled by its weight inside the output Hadamard scale, then every group of
//   bszm / num_tokens slots is summed into C[t].
static int mgemm_indexed_impl(const void* A, const void* act_u, int bszm_in, const void* tbl_B, const void* tbl_suh, const void* tbl_svh,
                              const int64_t* indices, const void* weights, int bszm, void* C, int m, int k, int n, int K, int cb, int c_fp32,
                              int min_index, int max_index, int num_tokens, void* stream);

extern "C" int exl3_mgemm_indexed(const void* A, int bszm_in, const void* tbl_B, const void* tbl_suh, const void* tbl_svh,
                                  const int64_t* indices, const void* weights, int bszm, void* C, int m, int k, int n, int K, int cb, int c_fp32,
                                  int min_index, int max_index, int num_tokens, void* stream)
{
    return mgemm_indexed_impl(A, nullptr, bszm_in, tbl_B, tbl_suh, tbl_svh, indices, weights, bszm, C, m, k, n, K, cb, c_fp32, min_index, max_index,
                              num_tokens, stream);
}

// exl3_mgemm with per-matrix output widths (quant/exl3_gemm.cuh:54-55, exl3_gemm.cu:433-447, kernel exl3_gemm_kernel.cuh:176-182): matrix i is
// size_n_list[i] columns wide (device int32, multiples of 128, <= n_max) and its [m][size_n_list[i]] output goes to c_ptrs[i] (device int64 table);
// slot j runs matrix indices ? indices[j] : j.  As in the reference: no routing weights, no expert range, one token.
extern "C" int exl3_mgemm_indexed_nlist(const void* A, int bszm_in, const void* tbl_B, const void* tbl_suh, const void* tbl_svh, const int64_t* indices,
                                        int bszm, const int32_t* size_n_list, const void* c_ptrs, int m, int k, int n_max, int K, int cb, int c_fp32,
                                        void* stream)
{
    EXL3_CHECK_ARG(A && tbl_B && tbl_suh && tbl_svh && size_n_list && c_ptrs, "exl3_mgemm (per-matrix widths): null pointer");
    EXL3_CHECK_ARG(bszm >= 1 && (bszm_in == 1 || bszm_in == bszm), "exl3_mgemm: A must have 1 or bszm slots");
    EXL3_CHECK_ARG(m >= 1 && m <= 16, "exl3_mgemm (per-matrix widths): 1..16 rows per slot");
    GemvTable t; memset((void*) &t, 0, sizeof(t));
    t.B = (const uint64_t*) tbl_B; t.suh = (const uint64_t*) tbl_suh; t.svh = (const uint64_t*) tbl_svh;
    t.indices = indices; t.C = (void*) c_ptrs;                        // (never written: every slot's output pointer comes from c_list)
    t.bszm = bszm; t.min_index = -1; t.max_index = -1; t.n = n_max; t.cbs_per_mat = n_max / 128;
    t.a_slot_stride = bszm_in == 1 ? 0 : (int64_t) m * k;
    t.c_slot_stride = 0;
    t.n_list = size_n_list; t.c_list = (const uint64_t*) c_ptrs;
    const void* Bs[1] = { tbl_B }; int ns[1] = { n_max };
    const void* su[1] = { tbl_suh }; const void* sv[1] = { tbl_svh }; void* Cs[1] = { (void*) c_ptrs };
    return run_mgemm(A, Bs, Cs, su, sv, nullptr, ns, bszm, m, k, K, cb, c_fp32, 0, (hipStream_t) stream, 0, nullptr, nullptr, nullptr, nullptr,
                     nullptr, nullptr, nullptr, 0.0f, &t);
}

// exl3_mgemm_indexed whose input is fp16(silu(G_j) * U_j) per slot (G, U: [bszm][m][k] fp16, the gate / up outputs of the routed experts): the
// silu_mul launch between the gate|up and the down exl3_mgemm of a MoE block (block_sparse_mlp.py / activation.cu) folded into the down launch.
extern "C" int exl3_mgemm_indexed_act(const void* G, const void* U, const void* tbl_B, const void* tbl_suh, const void* tbl_svh,
                                      const int64_t* indices, const void* weights, int bszm, void* C, int m, int k, int n, int K, int cb, int c_fp32,
                                      int min_index, int max_index, int num_tokens, void* stream)
{
    EXL3_CHECK_ARG(U, "exl3_mgemm_indexed_act: null up tensor");
    return mgemm_indexed_impl(G, U, bszm, tbl_B, tbl_suh, tbl_svh, indices, weights, bszm, C, m, k, n, K, cb, c_fp32, min_index, max_index,
                              num_tokens, stream);
}

static int mgemm_indexed_impl(const void* A, const void* act_u, int bszm_in, const void* tbl_B, const void* tbl_suh, const void* tbl_svh,
                              const int64_t* indices, const void* weights, int bszm, void* C, int m, int k, int n, int K, int cb, int c_fp32,
                              int min_index, int max_index, int num_tokens, void* stream)
{
    EXL3_CHECK_ARG(A && tbl_B && tbl_suh && tbl_svh && C, "exl3_mgemm: null pointer");
    EXL3_CHECK_ARG(bszm >= 1 && (bszm_in == 1 || bszm_in == bszm), "exl3_mgemm: A must have 1 or bszm slots");
    EXL3_CHECK_ARG(num_tokens >= 1 && bszm % num_tokens == 0, "exl3_mgemm: bszm must be divisible by num_tokens");
    EXL3_CHECK_ARG(num_tokens == 1 || min_index < 0, "exl3_mgemm: multi-token reduction (num_tokens > 1) is not compatible with expert-range filtering (min_index >= 0); TP-sharded experts must use num_tokens == 1");
    EXL3_CHECK_ARG(min_index < 0 || indices, "exl3_mgemm: an expert range needs indices");
    GemvTable t; memset((void*) &t, 0, sizeof(t));
    t.B = (const uint64_t*) tbl_B; t.suh = (const uint64_t*) tbl_suh; t.svh = (const uint64_t*) tbl_svh;
    t.indices = indices; t.weights = (const half_t*) weights; t.C = C;
    t.bszm = bszm; t.min_index = min_index; t.max_index = max_index; t.n = n; t.cbs_per_mat = n / 128;
    t.a_slot_stride = bszm_in == 1 ? 0 : (int64_t) m * k;
    t.c_slot_stride = (int64_t) m * n;
    t.act_u = (const half_t*) act_u;
    const void* Bs[1] = { tbl_B }; int ns[1] = { n };
    const void* su[1] = { tbl_suh }; const void* sv[1] = { tbl_svh }; void* Cs[1] = { C };
    int rc = run_mgemm(A, Bs, Cs, su, sv, nullptr, ns, bszm, m, k, K, cb, c_fp32, 0, (hipStream_t) stream, 0, nullptr, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, 0.0f, &t);
    if (rc < 0) return rc;
    if (weights)
    {
        const int64_t mn = (int64_t) m * n;
        mgemm_slot_reduce_kernel<<<dim3((unsigned) ((mn + 255) / 256)), dim3(256), 0, (hipStream_t) stream>>>(C, c_fp32, num_tokens, bszm / num_tokens, mn, indices, bszm,
                                                                                                               min_index, max_index);
        return exl3_check_launch("exl3_mgemm slot reduce") < 0 ? EXL3_ERR_HIP : rc;
    }
    return rc;
}

// exl3_mgemm_indexed_act with a deferred epilogue: the launch leaves raw rotated-basis split-k slabs [slot][n/128][S][m][128] fp32 (no svh, no
// routing weights, no slot sum); exl3_glue_resid_moe finishes them together with the residual add.  No expert range (one rank holds all experts).
extern "C" int exl3_mgemm_indexed_act_deferred(const void* G, const void* U, const void* tbl_B, const void* tbl_suh, const int64_t* indices, int bszm,
                                               int m, int k, int n, int K, int cb, float** slab_out, int* S_out, void* stream)
{
    EXL3_CHECK_ARG(G && U && tbl_B && tbl_suh && indices && slab_out && S_out, "exl3_mgemm_indexed_act_deferred: null pointer");
    EXL3_CHECK_ARG(bszm >= 1 && m >= 1 && m <= 4, "exl3_mgemm_indexed_act_deferred: 1..4 rows per slot");
    GemvTable t; memset((void*) &t, 0, sizeof(t));
    t.B = (const uint64_t*) tbl_B; t.suh = (const uint64_t*) tbl_suh; t.svh = nullptr;
    t.indices = indices; t.weights = nullptr; t.C = nullptr;
    t.bszm = bszm; t.min_index = -1; t.max_index = -1; t.n = n; t.cbs_per_mat = n / 128;
    t.a_slot_stride = (int64_t) m * k;
    t.c_slot_stride = (int64_t) m * n;
    t.act_u = (const half_t*) U;
    const void* Bs[1] = { tbl_B }; int ns[1] = { n };
    const void* su[1] = { tbl_suh };
    int rc = run_mgemm(G, Bs, nullptr, su, nullptr, nullptr, ns, bszm, m, k, K, cb, 0, 0, (hipStream_t) stream, GEMV_OUT_DEFERRED, nullptr, nullptr, slab_out,
                       S_out, nullptr, nullptr, nullptr, 0.0f, &t);
    return rc < 0 ? rc : EXL3_OK;
}

// Indexed gate|up launch of a MoE block with a deferred epilogue (generation 4): raw x (bszm_in == 1: one shared row set), pointer tables over
// [gate_0..gate_E-1, up_0..up_E-1], indices = the router's [gate slots | up slots] list; leaves slabs [slot][n/128][S][m][128] for
// exl3_mgemm_indexed_act_fx.
extern "C" int exl3_mgemm_indexed_deferred(const void* A, int bszm_in, const void* tbl_B, const void* tbl_suh, const int64_t* indices, int bszm,
                                           int m, int k, int n, int K, int cb, int force_split, float** slab_out, int* S_out, void* stream)
{
    EXL3_CHECK_ARG(A && tbl_B && tbl_suh && slab_out && S_out, "exl3_mgemm_indexed_deferred: null pointer");
    EXL3_CHECK_ARG(bszm >= 1 && (bszm_in == 1 || bszm_in == bszm) && m >= 1 && m <= 4, "exl3_mgemm_indexed_deferred: 1..4 rows per slot; A has 1 or bszm slots");
    GemvTable t; memset((void*) &t, 0, sizeof(t));
    t.B = (const uint64_t*) tbl_B; t.suh = (const uint64_t*) tbl_suh;
    t.indices = indices; t.bszm = bszm; t.min_index = -1; t.max_index = -1; t.n = n; t.cbs_per_mat = n / 128;
    t.a_slot_stride = bszm_in == 1 ? 0 : (int64_t) m * k;
    t.c_slot_stride = (int64_t) m * n;
    const void* Bs[1] = { tbl_B }; int ns[1] = { n }; const void* su[1] = { tbl_suh };
    int rc = run_mgemm(A, Bs, nullptr, su, nullptr, nullptr, ns, bszm, m, k, K, cb, 0, force_split, (hipStream_t) stream, GEMV_OUT_DEFERRED, nullptr, nullptr, slab_out,
                       S_out, nullptr, nullptr, nullptr, 0.0f, &t);
    return rc < 0 ? rc : EXL3_OK;
}

// Indexed down launch of a MoE block in the fx pipeline (generation 4): slot j's input is fp16(silu(g_j) * u_j) finished from the slabs of
// exl3_mgemm_indexed_deferred (gate slots first, then the up slots: gu_slabs, act_S; gu_svh_tbl = that launch's svh table, up svh = entry + up_off),
// its output rows -- out-Hadamard, svh and the routing weight applied per split-k partial -- are ADDED into the fixed-point residual accumulator
// R [num_tokens][m][n] int64 (slot j -> token j / (bszm / num_tokens)): no slot sum, no split-k reduce, no residual launch.
extern "C" int exl3_mgemm_indexed_act_fx(const float* gu_slabs, int act_S, const void* gu_svh_tbl, int up_off, const void* tbl_B, const void* tbl_suh,
                                         const void* tbl_svh, const int64_t* indices, const void* weights, int bszm, void* R, int m, int k, int n,
                                         int K, int cb, int num_tokens, int force_split, void* stream)
{
    EXL3_CHECK_ARG(gu_slabs && gu_svh_tbl && tbl_B && tbl_suh && tbl_svh && R, "exl3_mgemm_indexed_act_fx: null pointer");
    EXL3_CHECK_ARG(bszm >= 1 && m >= 1 && m <= 4 && act_S >= 1, "exl3_mgemm_indexed_act_fx: 1..4 rows per slot");
    EXL3_CHECK_ARG(num_tokens >= 1 && bszm % num_tokens == 0, "exl3_mgemm_indexed_act_fx: bszm must be divisible by num_tokens");
    GemvTable t; memset((void*) &t, 0, sizeof(t));
    t.B = (const uint64_t*) tbl_B; t.suh = (const uint64_t*) tbl_suh; t.svh = (const uint64_t*) tbl_svh;
    t.indices = indices; t.weights = (const half_t*) weights; t.C = R;
    t.bszm = bszm; t.min_index = -1; t.max_index = -1; t.n = n; t.cbs_per_mat = n / 128;
    t.c_slot_stride = (int64_t) m * n;
    t.act_svh = (const uint64_t*) gu_svh_tbl; t.act_up_off = up_off; t.slots_per_token = bszm / num_tokens;
    const size_t sstride = (size_t) (k / 128) * act_S * m * 128;
    const void* Bs[1] = { tbl_B }; int ns[1] = { n }; const void* su[1] = { tbl_suh }; const void* sv[1] = { tbl_svh };
    int rc = run_mgemm(nullptr, Bs, nullptr, su, sv, nullptr, ns, bszm, m, k, K, cb, 0, force_split, (hipStream_t) stream, GEMV_OUT_ATOMIC, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, 0.0f, &t, gu_slabs, gu_slabs + (size_t) bszm * sstride, act_S);
    return rc < 0 ? rc : EXL3_OK;
}

// down_proj whose input a = fp16(silu(g) * u) is finished from the gate / up launch's deferred slabs while the activation fragments are
// built (m <= 4): replaces exl3_glue_act + exl3_gemv_ex(IN_ROTATED).  g_slabs / u_slabs / act_S: as returned by the gate / up exl3_gemv_ex
// call (they live in the other workspace region than this launch's own slabs).  flags: EXL3_GEMV_OUT_DEFERRED optional.
// exl3_gemv_ex_norm that also finishes the PRODUCER linear's residual add (replaces the exl3_glue_resid launch between o_proj / down_proj and the
// next q|k|v / gate|up launch at m <= 4).  resid_in: fp16 [m][k] residual BEFORE the producer's output is added (read only); prod_slabs / prod_S /
// prod_svh: the producer's deferred slabs (exl3_gemv_ex*, GEMV_OUT_DEFERRED; they live in the other workspace region); every workgroup rebuilds
// resid_new = resid_in + linear_out for the Hadamard blocks of its k-slice, the column-block-0 workgroups write resid_out (a different buffer)
// and ss_out [m][k/128].  The RMSNorm row scale comes from ss_prev (sums of squares of resid_in): consumers of THIS launch's deferred slabs apply
// rsqrt(mean(resid_new^2) + eps) / rsqrt(mean(resid_in^2) + eps) (exl3_glue_qkv_rs, exl3_gemv_ex_act_rs).  Output: always deferred slabs.
extern "C" int exl3_gemv_ex_resid(const void* resid_in, const void* norm_w, const float* ss_prev, float eps, const float* prod_slabs, int prod_S,
                                  const void* prod_svh, void* resid_out, float* ss_out, const void* const* Bs, const void* const* suhs, const int* ns,
                                  int count, int m, int k, int K, int cb, int cpw, int force_split, float** slabs_out, int* S_out, void* stream)
{
    EXL3_CHECK_ARG(Bs && ns && resid_in && suhs, "exl3_gemv_ex_resid: null table");
    GemvResidIn rsd = { prod_slabs, prod_S, prod_svh, resid_out, ss_out };
    return run_mgemm(resid_in, Bs, nullptr, suhs, nullptr, nullptr, ns, count, m, k, K, cb, 0, force_split, (hipStream_t) stream,
                     GEMV_OUT_DEFERRED | GEMV_IN_NORM, nullptr, nullptr, slabs_out, S_out, nullptr, norm_w, ss_prev, eps, nullptr,
                     nullptr, nullptr, 0, nullptr, nullptr, &rsd, nullptr, cpw > 0 ? cpw : 4);
}

// The "fx" decode pipeline (round 3): the residual stream lives in a 64-bit fixed-point accumulator R [m][hidden] (value * 2^32).  o_proj / down_proj
// launches ADD their output rows into it (exl3_gemv_ex with EXL3_GEMV_OUT_ATOMIC: integer atomics, order-independent, so no split-k reduce and no
// residual launch), and this launch is the consumer: exl3_gemv_ex_norm whose input is R.  The RMSNorm scale it applies is the PREVIOUS residual's
// (ss_prev [m][k/128], complete) because the sums of squares of R itself are only known once every block has been read: the workgroups of column
// block 0 write them to ss_out, and whoever finishes this launch's slabs multiplies by r_new / r_prev (exl3_glue_qkv_rs, exl3_glue_act_rs) -- the
// protocol of exl3_gemv_ex_resid without its redundant slab reduction (the memory system has already summed the slices).  Output: deferred slabs.
extern "C" int exl3_gemv_ex_fx(const void* R, const void* norm_w, const float* ss_prev, float* ss_out, float eps, const void* const* Bs,
                               const void* const* suhs, const int* ns, int count, int m, int k, int K, int cb, int force_split,
                               float** slabs_out, int* S_out, void* stream)
{
    EXL3_CHECK_ARG(Bs && ns && R && suhs && ss_prev && ss_out && ss_prev != ss_out, "exl3_gemv_ex_fx: null table / ss_out must differ from ss_prev");
    return run_mgemm(R, Bs, nullptr, suhs, nullptr, nullptr, ns, count, m, k, K, cb, 0, force_split, (hipStream_t) stream,
                     GEMV_OUT_DEFERRED | GEMV_IN_NORM | GEMV_IN_FX, nullptr, nullptr, slabs_out, S_out, nullptr, norm_w, ss_prev, eps, nullptr,
                     nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, ss_out);
}

// exl3_gemv_ex_fx whose outputs are ADDED into fixed-point accumulators accs[i] (int64 [m][n_i], zero on entry: exl3_fx_zero_next) instead of being
// left as slabs: gate|up of the fx pipeline.  The rows still carry the previous residual's 1/rms: the consumer (exl3_gemv_ex_actfx) corrects.
extern "C" int exl3_gemv_ex_fx_atomic(const void* R, const void* norm_w, const float* ss_prev, float* ss_out, float eps, const void* const* Bs,
                                      void* const* accs, const void* const* suhs, const void* const* svhs, const int* ns, int count, int m, int k, int K,
                                      int cb, int force_split, int* S_out, void* stream)
{
    EXL3_CHECK_ARG(Bs && ns && R && suhs && svhs && accs && ss_prev && ss_out && ss_prev != ss_out, "exl3_gemv_ex_fx_atomic: null table");
    return run_mgemm(R, Bs, accs, suhs, svhs, nullptr, ns, count, m, k, K, cb, 0, force_split, (hipStream_t) stream,
                     GEMV_OUT_ATOMIC | GEMV_IN_NORM | GEMV_IN_FX, nullptr, nullptr, nullptr, S_out, nullptr, norm_w, ss_prev, eps, nullptr,
                     nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, ss_out);
}

// down_proj of the fx pipeline: input = fp16(silu(g) * u) formed from the fixed-point gate / up accumulators (int64 [m][k]) with the row-scale
// correction r_new / r_prev (ss_prev / ss_new [m][hidden/128]); output added into the residual accumulator R (EXL3_GEMV_OUT_ATOMIC) or left as slabs.
extern "C" int exl3_gemv_ex_actfx(const void* g_acc, const void* u_acc, const float* ss_prev, const float* ss_new, int hidden, float eps,
                                  const void* B, void* C, const void* suh, const void* svh, const void* bias, int m, int k, int n, int K, int cb,
                                  int flags, int force_split, float** slab_out, int* S_out, void* stream)
{
    EXL3_CHECK_ARG(g_acc && u_acc && B && suh && m <= 4, "exl3_gemv_ex_actfx: null pointer / m <= 4");
    const void* Bs[1] = { B }; void* Cs[1] = { C }; const void* su[1] = { suh }; const void* sv[1] = { svh }; const void* bi[1] = { bias };
    int ns[1] = { n };
    GemvRescale rs = { ss_prev, ss_new, hidden, eps };
    // the accumulators travel in the act_g / act_u fields; act_S = 1 and dummy svh pointers satisfy the slab-mode argument checks
    return run_mgemm(nullptr, Bs, C ? Cs : nullptr, su, svh ? sv : nullptr, bi, ns, 1, m, k, K, cb, 0, force_split, (hipStream_t) stream,
                     (flags & (GEMV_OUT_DEFERRED | GEMV_OUT_ATOMIC)) | GEMV_IN_ACTFX, nullptr, nullptr, slab_out, S_out, nullptr, nullptr, nullptr, 0.0f, nullptr,
                     (const float*) g_acc, (const float*) u_acc, 1, suh, suh, nullptr, (ss_prev && ss_new) ? &rs : nullptr, 0);
}

// exl3_gemv_ex (raw input A + suhs, deferred output, m <= 4) in the wave-per-column-block layout: cpw column blocks of one matrix per workgroup,
// every wave streams the whole k-slice of its column block and writes its own slab (no cross-wave reduction).  cpw in [1, 16].
extern "C" int exl3_gemv_ex_wpc(const void* A, const void* const* Bs, const void* const* suhs, const int* ns, int count, int m, int k, int K, int cb,
                                int cpw, int force_split, float** slabs_out, int* S_out, void* stream)
{
    EXL3_CHECK_ARG(Bs && ns && A && suhs && cpw >= 1, "exl3_gemv_ex_wpc: null table / cpw");
    return run_mgemm(A, Bs, nullptr, suhs, nullptr, nullptr, ns, count, m, k, K, cb, 0, force_split, (hipStream_t) stream, GEMV_OUT_DEFERRED,
                     nullptr, nullptr, slabs_out, S_out, nullptr, nullptr, nullptr, 0.0f, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                     nullptr, nullptr, cpw);
}

// exl3_gemv_ex_act whose gate / up slabs come from an exl3_gemv_ex_resid launch: g and u are multiplied by the exact-over-estimated RMSNorm
// scale of their row (ss_prev / ss_new [m][hidden/128], eps) before svh and silu.
extern "C" int exl3_gemv_ex_act_rs(const float* g_slabs, const float* u_slabs, int act_S, const void* svh_g, const void* svh_u,
                                   const float* ss_prev, const float* ss_new, int hidden, float eps,
                                   const void* B, void* C, const void* suh, const void* svh, const void* bias, int m, int k, int n, int K, int cb,
                                   int c_fp32, int flags, int cpw, int force_split, float** slab_out, int* S_out, void* stream)
{
    EXL3_CHECK_ARG(B && suh && ss_prev && ss_new && hidden % 128 == 0, "exl3_gemv_ex_act_rs: null pointer");
    const void* Bs[1] = { B }; void* Cs[1] = { C }; const void* su[1] = { suh }; const void* sv[1] = { svh }; const void* bi[1] = { bias };
    int ns[1] = { n };
    GemvRescale rs = { ss_prev, ss_new, hidden, eps };
    return run_mgemm(nullptr, Bs, C ? Cs : nullptr, su, svh ? sv : nullptr, bi, ns, 1, m, k, K, cb, c_fp32, force_split, (hipStream_t) stream,
                     (flags & (GEMV_OUT_DEFERRED | GEMV_OUT_ATOMIC)), nullptr, nullptr, slab_out, S_out, nullptr, nullptr, nullptr, 0.0f, nullptr,
                     g_slabs, u_slabs, act_S, svh_g, svh_u, nullptr, &rs, cpw);
}

// o_proj whose input is the decode attention's output, finished inside the launch: `part` = the context-split partial records exl3_attn_decode_qcache_split
// left ([m][blocks][gq][nsplit][132] fp32), merged per (row, query head) by the preparation task that needs that head (head_dim 128: one head = one
// Hadamard block of o_proj's input) with the arithmetic of the merge kernel -- same bits as attn_decode_qcache + exl3_gemv_ex, one launch less.
// flags: GEMV_OUT_DEFERRED or GEMV_OUT_ATOMIC (C = the fixed-point residual) or 0 (final output).  reference: libtorch/attention.cpp:246-504 (attention, then o_proj).
extern "C" int exl3_gemv_ex_attm(const float* part, int nsplit, int gq, int blocks, int head_dim, const void* B, void* C, const void* suh, const void* svh, const void* bias,
                                 int m, int k, int n, int K, int cb, int c_fp32, int flags, int force_split, float** slab_out, int* S_out, void* stream)
{
    EXL3_CHECK_ARG(part && B && suh, "exl3_gemv_ex_attm: null pointer");
    const void* Bs[1] = { B }; void* Cs[1] = { C }; const void* su[1] = { suh }; const void* sv[1] = { svh }; const void* bi[1] = { bias };
    int ns[1] = { n };
    GemvAttm at = { part, nsplit, gq, blocks, 0u, head_dim };
    return run_mgemm(nullptr, Bs, C ? Cs : nullptr, su, svh ? sv : nullptr, bi, ns, 1, m, k, K, cb, c_fp32, force_split, (hipStream_t) stream,
                     (flags & (GEMV_OUT_DEFERRED | GEMV_OUT_ATOMIC)), nullptr, nullptr, slab_out, S_out, nullptr, nullptr, nullptr, 0.0f, nullptr,
                     nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, &at);
}

// o_proj fed straight by the q|k|v launch's deferred slabs (the decode step WITHOUT the attention core: o_proj's input is the finished q): the q|k|v epilogue
// -- exl3_glue_qkv_tab's work: split-k reduce, output Hadamard, row-scale correction, svh, RoPE from the per-step tables, the 4-bit append of K / V --
// runs inside this launch: a preparation task finishes the q block it needs, idle half-waves of the column-block-0 workgroups append K / V.  Same bits
// as exl3_glue_qkv_tab + exl3_gemv_ex (shared device functions); one launch less per layer.  head_dim 64 | 128, 4-bit K and V.
// reference: libtorch/attention.cpp:283-400 (q / k / v epilogues, rope, cache append) + :497-508 (o_proj).
extern "C" int exl3_gemv_ex_qkvm(const float* sq, const float* sk, const float* sv, int S_qkv, const void* svh_q, const void* svh_k, const void* svh_v,
                                 const float* rope_sin, const float* rope_cos, const int64_t* slots, const float* ss_prev, const float* ss_new, int hidden, float eps,
                                 int rope_mode, int head_dim, int heads_kv, void* q_out, void* k_cache, void* k_scales, void* v_cache, void* v_scales,
                                 const void* B, void* C, const void* suh, const void* svh, const void* bias, int m, int k, int n, int K, int cb, int c_fp32,
                                 int flags, int force_split, float** slab_out, int* S_out, void* stream)
{
    EXL3_CHECK_ARG(B && suh, "exl3_gemv_ex_qkvm: null pointer");
    EXL3_CHECK_ARG(!ss_new || (ss_prev && hidden > 0 && hidden % 128 == 0), "exl3_gemv_ex_qkvm: rescale needs ss_prev and hidden");
    EXL3_CHECK_ARG((head_dim == 64 || head_dim == 128) && heads_kv >= 1 && (heads_kv * head_dim) % 128 == 0, "exl3_gemv_ex_qkvm: head_dim 64 | 128, whole 128-value kv blocks");
    const void* Bs[1] = { B }; void* Cs[1] = { C }; const void* su[1] = { suh }; const void* svp[1] = { svh }; const void* bi[1] = { bias };
    int ns[1] = { n };
    GemvQkvm q; memset((void*) &q, 0, sizeof(q));
    q.sq = sq; q.sk = sk; q.sv = sv; q.S = S_qkv; q.rope_mode = rope_mode;
    q.svh_q = (const half_t*) svh_q; q.svh_k = (const half_t*) svh_k; q.svh_v = (const half_t*) svh_v;
    q.rope_sin = rope_sin; q.rope_cos = rope_cos; q.slots = slots;
    q.rs = GemvRescale{ ss_prev, ss_new, hidden, eps };
    q.q_out = (half_t*) q_out; q.k_cache = (uint32_t*) k_cache; q.k_scales = (half_t*) k_scales; q.v_cache = (uint32_t*) v_cache; q.v_scales = (half_t*) v_scales;
    q.hd = head_dim; q.kvb = heads_kv * head_dim / 128;
    return run_mgemm(nullptr, Bs, C ? Cs : nullptr, su, svh ? svp : nullptr, bi, ns, 1, m, k, K, cb, c_fp32, force_split, (hipStream_t) stream,
                     (flags & (GEMV_OUT_DEFERRED | GEMV_OUT_ATOMIC)), nullptr, nullptr, slab_out, S_out, nullptr, nullptr, nullptr, 0.0f, nullptr,
                     nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, &q);
}

extern "C" int exl3_gemv_ex_act(const float* g_slabs, const float* u_slabs, int act_S, const void* svh_g, const void* svh_u,
                                const void* B, void* C, const void* suh, const void* svh, const void* bias, int m, int k, int n, int K, int cb,
                                int c_fp32, int flags, int force_split, float** slab_out, int* S_out, void* stream)
{
    EXL3_CHECK_ARG(B && suh, "exl3_gemv_ex_act: null pointer");
    const void* Bs[1] = { B }; void* Cs[1] = { C }; const void* su[1] = { suh }; const void* sv[1] = { svh }; const void* bi[1] = { bias };
    int ns[1] = { n };
    return run_mgemm(nullptr, Bs, C ? Cs : nullptr, su, svh ? sv : nullptr, bi, ns, 1, m, k, K, cb, c_fp32, force_split, (hipStream_t) stream,
                     (flags & (GEMV_OUT_DEFERRED | GEMV_OUT_ATOMIC)), nullptr, nullptr, slab_out, S_out, nullptr, nullptr, nullptr, 0.0f, nullptr,
                     g_slabs, u_slabs, act_S, svh_g, svh_u);
}
