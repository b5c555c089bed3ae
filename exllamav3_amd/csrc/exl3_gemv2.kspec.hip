// EXL3 quantized GEMV / small-m GEMM for gfx950, kernel generation 2 ("column-pair per lane").
//
//   C = ((A * suh) H) @ dequant(B) H * svh (+ bias)        reference: quant/exl3_gemv_kernel.cuh:138-402,
//                                                          quant/exl3_gemm_kernel.cuh:8-80 (semantics only)
//
// Data-layout observation that drives the design: in the EXL3 tile bitstream the 32 weights with stream
// indices [32c, 32c+32) are exactly columns c and c+8 of the 16x16 tile (all 16 rows of each), and they occupy
// exactly K consecutive, word-aligned u32 words (words [cK, cK+K) of the tile).  So if lane (8T + c) of a wave
// loads K consecutive words at word offset (8T + c) * K of a tile row, then
//   * the wave's load is ONE fully contiguous 256*K-byte run (K = 4: a 1 KiB global_load_dwordx4),
//   * the lane owns two complete output columns (16T + c, 16T + c + 8) of the 128-column block for 16 k-rows,
//   * every bit-window shift is a compile-time constant for every K in 1..8 (one alignbit/bfe per weight),
//   * the only cross-lane traffic is the previous lane's last word (16 - K carry-in bits of the trellis state).
// A wave therefore owns a whole 128-column output block (= one output Hadamard block) and walks down k.
//
// MAC: v_mfma_f32_4x4x4_16B_f16 with A-broadcast (cbsz = 4): the instruction computes, for every lane j of the
// wave, D[0..3][j] += sum_{k<4} A[0..3][k] * B[k][j] where B[.][j] is lane j's own 4 halves and A comes from
// lanes 4*abid .. 4*abid+3.  That is a per-lane dot product against up to 4 activation rows -- exactly a GEMV with
// per-lane columns, on the matrix pipe, with no cross-lane reduction at all.  Rows 4..15 use abid = 1..3.
//
// Workgroup = 4 waves = one column block x one k-slice; the waves split the slice, reduce through LDS, and the
// output Hadamard runs in the epilogue (S == 1) or in the split-k reduce kernel (S > 1, exl3_gemv.hip).
// The input Hadamard of each 128-block of x is computed by the wave that consumes it, just in time, into a
// wave-private double-buffered LDS fragment store (no workgroup barrier in the main loop).
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include "exl3_gemv_args.h"
#include "exl3_lane_decode.cuh"
#include "exl3_gemv2_tail.cuh"
#include "exl3_glue_device.cuh"

#include <type_traits>

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

#define G2_MODE_PLAIN 0     // rotated or raw input
#define G2_MODE_NORM  1     // GEMV_IN_NORM: RMSNorm of the residual stream while building the activation fragments
#define G2_MODE_TAIL  2     // in-kernel tail epilogue (exl3_gemv2_tail.cuh)
#define G2_MODE_TABLE 3     // device-side pointer tables + indices + routing weights (MoE exl3_mgemm)
#define G2_MODE_ACT   4     // GEMV_IN_ACT: down_proj whose input is finished from the gate / up slabs (replaces the glue_act launch at m <= 4)
#define G2_MODE_RNORM 5     // GEMV_IN_NORM | GEMV_IN_RESID: the residual add of the PREVIOUS linear (its split-k slabs) + RMSNorm while the fragments are built
#define G2_MODE_WPLAIN 6    // PLAIN in the wave-per-column-block layout
#define G2_MODE_WACT   7    // ACT in the wave-per-column-block layout
// Wave-per-column-block layout (RNORM, WPLAIN, WACT; deferred output, m <= 4): a workgroup = `cpw` column blocks of ONE matrix x one k-slice, wave w
// streams ALL tile rows of the slice for column block (group * cpw + w) and writes its own slab -- no cross-wave reduction, no barrier after the
// streaming loop -- while the activation fragments of the slice are still built once per workgroup.  What it buys: the fused prologues (RESID,
// ACT) re-read the producer's slabs once per workgroup, i.e. per cpw column blocks instead of per column block (29 MB -> 2-4 MB per launch for
// Llama-3.1-8B's gate|up and down), which is what makes folding the glue launches into the consumer GEMVs pay.
#define G2_IS_WPC(M) ((M) == G2_MODE_RNORM || (M) == G2_MODE_WPLAIN || (M) == G2_MODE_WACT)
#define G2_IS_ACT(M) ((M) == G2_MODE_ACT || (M) == G2_MODE_WACT)
// The modes are separate instantiations because the hot loop needs 62 of the 64 VGPRs that allow two 16-wave workgroups per CU: code
// of a cold path that is merely present makes the allocator spill (scratch also slows every launch by ~1 us, measured).
// register budget (waves per SIMD) of an instantiation: chosen so that NO variant spills (tests/test_no_scratch.py)
constexpr int g2_waves_per_eu(int K, int CB, int NG, int MODE)
{
    if (MODE == G2_MODE_ACT) return 2;
    if (G2_IS_WPC(MODE)) return 4;                // up to 16 waves per workgroup, 128 VGPRs: 8 prefetched slab lines (32 VGPRs) on top of the prep
    if (NG == 4) return 3;
    if (NG == 2) return 4;
    if (G2_PF > 2) return 6;
    const bool normish = MODE == G2_MODE_NORM || MODE == G2_MODE_RNORM;
    int w = (K >= 5 && normish && CB != EXL3_CB_MUL1) ? 5
          : ((K >= 5 || ((normish || MODE == G2_MODE_TAIL) && CB != EXL3_CB_MUL1)) ? 6
          : ((normish || MODE == G2_MODE_TABLE || MODE == G2_MODE_TAIL || CB != EXL3_CB_MUL1) ? 7 : 8));
    return w;
}

template <int K, int CB, int VAR, int NG, int MODE>
// m <= 4 (NG == 1): two 16-wave workgroups per CU need <= 64 VGPRs (K <= 4, MUL1: the hot loop uses 59..62).  Variants that do not fit
// (3INST/MCG, NORM prep, K >= 5 rings) get the next budget instead of spilling: any scratch use costs every launch ~1-2 us
// (profiles/r01_launch_chain_microbench.json)
#ifdef G2_ABL_ILP
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4)))
#else
__global__ __launch_bounds__(MODE == G2_MODE_ACT ? 256 : 1024 / NG) __attribute__((amdgpu_waves_per_eu(g2_waves_per_eu(K, CB, NG, MODE))))
#endif
void exl3_gemv2_kernel(const GemvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_tail_flag;
    constexpr bool SPLIT = (VAR == 1) && (CB != EXL3_CB_MUL1);
    constexpr bool RAW = (VAR == 1) && (CB == EXL3_CB_MUL1);
    constexpr int MR = 4 * NG;                       // activation rows held by the A operand
    constexpr int AH = SPLIT ? 32 : 16;              // halves per (tile row, activation row)
    constexpr int NW = 8 * K;
#ifndef G2_WPC_PF
#define G2_WPC_PF 4
#endif
    // tile rows per work unit = depth of the per-wave weight-row register ring.  The wave-per-column-block layouts run 4 waves per SIMD (128 VGPRs)
    // and every wave walks its slice alone: a 2-row ring left them waiting on HBM (o_proj 9.9 vs 5.5 us), so they keep 4 rows in flight
    constexpr int PF = G2_IS_WPC(MODE) ? G2_WPC_PF : G2_PF;

#ifdef G2_TIMING
    // diagnostics build: wave 0 of every workgroup leaves 100 MHz timestamps of its phases in the workspace tail (48 MiB offset)
    uint64_t tstamp[6];
    tstamp[0] = __builtin_amdgcn_s_memrealtime();
    #define G2_T(i) tstamp[i] = __builtin_amdgcn_s_memrealtime()
#else
    #define G2_T(i)
#endif
    // Every scalar argument of the prologue is read HERE, in one batch of scalar loads: left at their use sites they become a chain of 6-10
    // dependent scalar-cache round trips (~200 ns each) ahead of the first vector load (tools/gemv_timeline.py: 1.1 us (PLAIN) to 2.1 us (NORM)
    // from workgroup entry to "loads issued").  For the same reason the divisions by launch constants are multiply-highs (gemv_udiv) and the
    // (k-slice, column block) pair comes from a 2-D grid instead of a division of the linear block id.
    const int a_S = a.S, a_k = a.k, a_kslice = a.kslice, a_flags = a.flags, a_chb = a.chunk_blocks, a_nm = a.num_mats;
    const int a_cbf[GEMV_MAX_MATS] = { 0, a.cbf[0], a.cbf[1], a.cbf[2] };
    const uint32_t mg_m = a.magic_m, mg_nwv = a.magic_nwv, mg_nhw = a.magic_nhw;
    const int a_nwv = a.nwv;
    // ACT modes: the slab bases of the producer launch live on another line of the argument block: same batch (dead code in the other modes)
    const float* const a_act_g = a.act_g; const float* const a_act_u = a.act_u;
    const half_t* const a_act_svh_g = a.act_svh_g; const half_t* const a_act_svh_u = a.act_svh_u;
    const int a_act_S = a.act_S;
    const float* const a_act_rs_new = a.act_rs.ss_new;
    if constexpr (G2_IS_ACT(MODE)) asm volatile("" :: "s"(a_act_g), "s"(a_act_S));     // ... and make it THIS batch (the compiler would sink it to the first use)
    if constexpr (MODE != G2_MODE_TABLE)
    {
        // touch the four matrix records (5 cache lines) in the same batch: the record of THIS workgroup's matrix is chosen from the values above
        // (a dependent load), which then finds its line in the scalar cache instead of paying a second miss
        const void* t0 = a.mat[0].B; const void* t1 = a.mat[1].B; const void* t2 = a.mat[2].B; const void* t3 = a.mat[3].B; const void* t4 = a.mat[3].xsum;
        asm volatile("" :: "s"(t0), "s"(t1), "s"(t2), "s"(t3), "s"(t4));
    }
    const half_t* const a_A = a.A;
    const half_t* const a_norm_w = a.norm_w;
    const float* const a_ss_part = a.ss_part;
    const float a_eps = a.eps;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int m = a.m;

    // xcd_local tail epilogues: workgroup i runs on XCD i % 8 (tools/ubench_xcc.hip), so logical workgroup (i % 8) * (grid / 8) + i / 8 puts each
    // run of grid / 8 consecutive logical ids -- hence all S slices of a column block -- on one XCD (host: column blocks % 8 == 0)
    // grid = (S k-slices, column blocks): the linear dispatch order (x fastest) is the former 1-D order column block * S + slice
    int s = blockIdx.x;
    int cbg = blockIdx.y;                            // classic: global column block of the workgroup; WPC: column-block GROUP of the workgroup
    if constexpr (MODE == G2_MODE_TAIL)
    {
        if (a.epi.xcd_local)
        {
            const int lin = blockIdx.y * a_S + blockIdx.x, total = gridDim.x * gridDim.y;
            const int bid = (lin & 7) * (total >> 3) + (lin >> 3);
            s = bid % a_S; cbg = bid / a_S;
        }
    }
    constexpr bool WPC = G2_IS_WPC(MODE);
    int mi = 0;
    const uint32_t* __restrict__ Bm;
    const half_t* __restrict__ suh;
    const half_t* A_in = a_A;
    int n, cbl, ws_off;
    bool wave_live = true;                           // WPC: this wave has a column block (the last group of a matrix may be partial)
    if constexpr (MODE == G2_MODE_TABLE)
    {
        const int slot = cbg / a.tbl.cbs_per_mat;
        const SlotRef_t sr = resolve_slot(a.tbl, slot);
        if (sr.mat_index < 0) return;                                  // filtered-out slot: nothing is written (as the reference)
        Bm = (const uint32_t*) a.tbl.B[sr.mat_index];
        suh = (const half_t*) a.tbl.suh[sr.mat_index];
        n = a.tbl.n; cbl = cbg - slot * a.tbl.cbs_per_mat;
        if (a.tbl.n_list) { n = a.tbl.n_list[sr.mat_index]; if (cbl >= (n >> 7)) return; }      // per-matrix widths: the grid is sized for the widest
        A_in = a_A + (size_t) slot * a.tbl.a_slot_stride;
        ws_off = slot * a.tbl.cbs_per_mat * a_S * m * 128;
    }
    else if constexpr (WPC)
    {
        // groups never straddle matrices (q|k|v, gate|up have different suh): mat[i].cb_first holds the first GROUP of matrix i
        #pragma unroll
        for (int i = 1; i < GEMV_MAX_MATS; ++i) if (i < a_nm && cbg >= a_cbf[i]) mi = i;
        Bm = a.mat[mi].B; suh = a.mat[mi].suh;
        n = a.mat[mi].n; ws_off = a.mat[mi].ws_offset;
        cbl = (cbg - a.mat[mi].cb_first) * a.cpw + wave;
        wave_live = cbl < (n >> 7);
        if (!wave_live) cbl = (n >> 7) - 1;
    }
    else
    {
        #pragma unroll
        for (int i = 1; i < GEMV_MAX_MATS; ++i) if (i < a_nm && cbg >= a_cbf[i]) mi = i;
        Bm = a.mat[mi].B; suh = a.mat[mi].suh;
        n = a.mat[mi].n; cbl = cbg - a.mat[mi].cb_first; ws_off = a.mat[mi].ws_offset;
    }
    const int tiles_n = n >> 4;
    const int k0s = s * a_kslice;
    const int k1s = min(k0s + a_kslice, a_k);
    const int nb = (k1s - k0s) >> 7;                 // 128-blocks in the workgroup's slice
    const int nwv = a_nwv;                           // waves per workgroup (1..16) = blockDim.x / 64
    // Work split inside the workgroup: the slice is walked in units of PF (= 2) tile rows, unit u belongs to wave u % nwv.  All waves
    // therefore advance down k together, which lets the workgroup build the activation fragments of a CHUNK of Hadamard blocks ONCE,
    // cooperatively (one (block, row) task per 32-lane half-wave), instead of every wave rotating every block it touches: at batch 16 the
    // per-wave version spent 6x more VALU time on input Hadamards than on decoding weights (tools/gemv_timeline.py).
    const int units = nb * (8 / PF);
    // unit i of this wave = ubase + i * ustride.  One chunk (the usual case: the whole slice's fragments fit the LDS budget): contiguous
    // ranges per wave, [units*w/nwv, units*(w+1)/nwv); several chunks: unit u belongs to wave u % nwv so that every wave has rows in
    // every chunk.
    const bool one_chunk = a_chb >= nb;
    // WPC: every live wave walks ALL units of the slice (its own column block); the host guarantees one chunk
    const int ubase = WPC ? 0 : (one_chunk ? gemv_udiv(units * wave, mg_nwv) : wave);
    const int ustride = (WPC || one_chunk) ? 1 : nwv;
    const int nunits_w = WPC ? (wave_live ? units : 0)
                             : (one_chunk ? gemv_udiv(units * (wave + 1), mg_nwv) - ubase : (wave < units ? gemv_udiv(units - wave + nwv - 1, mg_nwv) : 0));

    // LDS carve: fragments of one chunk [blk][tile row 8][row m][AH halves] | partials [nwv][MR][128] fp32 + per-wave row sums |
    //            tile-row sums [blk * 8][m] fp32 (RAW) | 1/rms per row [16] fp32 (NORM)
    const int chb = a_chb;                           // blocks per chunk (host: LDS budget)
    const size_t frag_halves = (size_t) chb * 8 * m * AH;
    half_t* xa = (half_t*) smem;
    float* part = (float*) (smem + ((frag_halves * 2 + 15) & ~(size_t) 15));
    float* tsum = part + (size_t) nwv * MR * 128 + (size_t) nwv * MR;
    float* rmf_s = tsum + (size_t) chb * 8 * m;

    float rowsum[2 * NG];
    #pragma unroll
    for (int i = 0; i < 2 * NG; ++i) rowsum[i] = 0.0f;
    const int l32 = lane & 31, hw = lane >> 5;
    const bool in_rotated = (a_flags & GEMV_IN_ROTATED) != 0;
    const half_t* __restrict__ xh_in = MODE == G2_MODE_TABLE ? nullptr : a.mat[mi].xh;
    const half_t* __restrict__ x_src = in_rotated ? xh_in : A_in;          // one scalar select (two pointers picked per load became a stack table)
    const int npass = (m + 1) >> 1;                  // row pairs (2p, 2p + 1) held by the two half-waves of the streaming loop
    const int hwid = tid >> 5, nhw = nwv * 2;        // half-wave id in the workgroup: the prep task owner

    // GEMV_IN_NORM: A is the fp16 residual stream; x = fp16(resid * norm_w * rsqrt(mean(resid^2) + eps)) is formed in the prep tasks, per
    // row from the per-block sums of squares a glue kernel left behind (same arithmetic and summation order as glue_norm_kernel /
    // rms_norm: norm.cu:20-120), so the RMSNorm between two linears costs no launch and no single-workgroup pass.
    constexpr bool in_norm = MODE == G2_MODE_NORM || MODE == G2_MODE_RNORM;
    // RNORM: this launch also finishes the producer linear's output for the blocks it needs: resid_new = resid_in + out-had(slab sum) * svh.  The
    // workgroups of column block 0 of matrix 0 (one per k-slice) publish resid_new and the per-block sums of squares; the row scale used here
    // is the PREVIOUS residual's 1/rms (ss_part), corrected by whoever finishes this launch's outputs (GemvRescale)
    constexpr bool RES = MODE == G2_MODE_RNORM;
    const bool res_writer = RES && cbg == 0;                 // WPC: group 0 (= the first column blocks of matrix 0), one workgroup per k-slice

    // ---- streaming state: issue the first weight rows before anything else
    const int T = lane >> 3, c = lane & 7;
    const uint32_t* __restrict__ strip = Bm + ((size_t) (k0s >> 4) * tiles_n + (size_t) cbl * 8) * NW + (size_t) lane * K;
    const size_t row_stride = (size_t) tiles_n * NW;
    const int prev_lane_addr = ((lane & ~7) | ((lane - 1) & 7)) << 2;   // ds_bpermute byte address
    const int last_unit = nunits_w > 0 ? ubase + (nunits_w - 1) * ustride : 0;

    float4_t acc_c[NG], acc_d[NG];
    #pragma unroll
    for (int gq = 0; gq < NG; ++gq) { acc_c[gq] = float4_t{ 0.f, 0.f, 0.f, 0.f }; acc_d[gq] = float4_t{ 0.f, 0.f, 0.f, 0.f }; }


    // prep task fetch: task t = it * nhw + hwid of chunk (c0, cnt) -> (block c0 + t / m, row t % m); loads only
    // NORM at m <= 4 and hidden <= 4096: the task owner reduces its row's 32 partial sums of squares itself (loaded with the task's
    // operands: one memory latency, no extra workgroup barrier); otherwise 1/rms per row goes through LDS (rmf_s) once per launch.
    const bool norm_in_task = in_norm && NG == 1 && (a_k >> 7) <= 32;
    // ACT mode: the first 8 gate and 8 up slab lines of the task's block travel with the task operands (issued before the weight rows)
    constexpr int ACT_PRE = MODE == G2_MODE_ACT ? 8 : (MODE == G2_MODE_WACT ? 2 : 1);   // WPC layouts: 128-VGPR budget
    constexpr int SLAB_PRE = MODE == G2_MODE_ACT ? 8 : (MODE == G2_MODE_WACT ? 2 : (RES ? 4 : 1));                // slab lines of the first set that travel with the task operands
    struct PrepIn { half4_t xv, sv, wv; float ss, ssn; float4_t sg[SLAB_PRE], su[ACT_PRE]; half4_t svg, svu; };
    auto fetch = [&] (int c0, int cnt, int it) -> PrepIn
    {
        PrepIn r; r.xv = half4_t{ 0, 0, 0, 0 }; r.sv = r.xv; r.wv = r.xv; r.ss = 0.0f; r.ssn = 0.0f;
        const int t = min(it * nhw + hwid, cnt * m - 1);
        const int tq = gemv_udiv(t, mg_m);
        const int blk = c0 + tq, row = t - tq * m;
        const size_t kofs = (size_t) k0s + 128 * blk;
        if constexpr (!G2_IS_ACT(MODE)) r.xv = ((const half4_t*) (x_src + (size_t) row * a_k + kofs))[l32];
        else
        {
            const int blk_abs = (k0s >> 7) + blk;
            const float* pg = a_act_g + ((size_t) blk_abs * a_act_S * m + row) * 128;
            const float* pu = a_act_u + ((size_t) blk_abs * a_act_S * m + row) * 128;
            const size_t st = (size_t) m * 128;
            #pragma unroll
            for (int i = 0; i < ACT_PRE; ++i)
            {
                r.sg[i] = ((const float4_t*) (pg + (size_t) min(i, a_act_S - 1) * st))[l32];
                r.su[i] = ((const float4_t*) (pu + (size_t) min(i, a_act_S - 1) * st))[l32];
            }
            r.svg = ((const half4_t*) (a_act_svh_g + blk_abs * 128))[l32];
            r.svu = ((const half4_t*) (a_act_svh_u + blk_abs * 128))[l32];
            if (a_act_rs_new)
            {
                const int nbh = a.act_rs.k >> 7;
                if (l32 < nbh) { r.ss = a.act_rs.ss_prev[(size_t) row * nbh + l32]; r.ssn = a_act_rs_new[(size_t) row * nbh + l32]; }
            }
        }
        if constexpr (RES)
        {
            const int blk_abs = (k0s >> 7) + blk;
            const float* pg = a.rs_slab + ((size_t) blk_abs * a.rs_S * m + row) * 128;
            const size_t st = (size_t) m * 128;
            #pragma unroll
            for (int i = 0; i < SLAB_PRE; ++i) r.sg[i] = ((const float4_t*) (pg + (size_t) min(i, a.rs_S - 1) * st))[l32];
            r.svg = ((const half4_t*) (a.rs_svh + blk_abs * 128))[l32];
        }
        if constexpr (MODE == G2_MODE_TABLE)
        {
            if (a.tbl.act_u) r.wv = ((const half4_t*) (a.tbl.act_u + (A_in - a_A) + (size_t) row * a_k + kofs))[l32];   // the slot's `up` row
        }
        if (!in_rotated)
        {
            r.sv = ((const half4_t*) (suh + kofs))[l32];
            if constexpr (in_norm)
            {
                r.wv = ((const half4_t*) (a_norm_w + kofs))[l32];
                if (norm_in_task && l32 < (a_k >> 7)) r.ss = a_ss_part[(size_t) row * (a_k >> 7) + l32];
            }
        }
        return r;
    };
    // the first task's operands are requested first (before the 1/rms loads of NORM mode) ...
    PrepIn nx = fetch(0, min(chb, nb), 0);

    // ... and only then the first weight rows: loads return in issue order per wave, so with the weights first the (small, L2-resident)
    // activation operands could not be consumed -- and the input Hadamards could not start -- before the first weight rows had arrived from
    // HBM; this way the prep computes underneath the weight latency
    LaneWords<K> ring[PF];
#ifdef G2_ABL_LATE_RING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // diagnostics build: the first weight rows are requested only after the task operands arrived
#endif
    if (nunits_w > 0)
    {
        #pragma unroll
        for (int u = 0; u < PF; ++u) load_lane_words<K>(ring[u], strip + (size_t) (PF * ubase + u) * row_stride);
    }
    G2_T(1);

    if (in_norm && !norm_in_task)
    {
        // 1/rms of row h by half-wave h (m <= 16 <= half-waves of any launch with >= 8 waves; fewer waves loop)
        const int nblk_k = a_k >> 7;
        for (int base = 0; base < m; base += nhw)                       // uniform trip count: the butterflies need whole waves
        {
            const int row = min(base + hwid, m - 1);
            float s2 = 0.0f;
            for (int bb = 0; bb < nblk_k; bb += 32)
            {
                float v = (bb + l32 < nblk_k) ? a_ss_part[(size_t) row * nblk_k + bb + l32] : 0.0f;
                #pragma unroll
                for (int i = 1; i < 32; i <<= 1) v += xor_lane(v, i);
                s2 += v;
            }
            if (l32 == 0 && base + hwid < m) rmf_s[row] = __frsqrt_rn(s2 / (float) a_k + a_eps);
        }
        __syncthreads();
    }

    // this lane's A row in the streaming loop; lanes whose row is >= m read row m-1 (their MFMA output rows are never stored)
    const half_t* arow = xa + (size_t) min(lane & 15, m - 1) * AH;
    const size_t astep = (size_t) m * AH;                               // halves per tile row
    int ui = 0;                                                         // index of this wave's next unit

    for (int c0 = 0; c0 < nb; c0 += chb)
    {
        const int cnt = min(chb, nb - c0);
        if (c0 > 0) __syncthreads();                                    // the previous chunk's fragments are no longer read

        // ---- cooperative prep of blocks [c0, c0 + cnt): task t = (block t / m, row t % m), one per half-wave, software pipelined by one
        {
            const int ntask = cnt * m;
            const int trips = gemv_udiv(ntask + nhw - 1, mg_nhw);
            if (c0 > 0) nx = fetch(c0, cnt, 0);
            for (int it = 0; it < trips; ++it)
            {
                const PrepIn cur = nx;
                // software pipelining by one task, except in ACT mode where a task's operands are 64 VGPRs of slab lines (fetched after the task)
                if constexpr (!G2_IS_ACT(MODE) && !RES) { if (it + 1 < trips) nx = fetch(c0, cnt, it + 1); }
                const int t = it * nhw + hwid;
                const bool act = t < ntask;
                const int tc = min(t, ntask - 1);
                const int blk_l = gemv_udiv(tc, mg_m), row = tc - blk_l * m;  // chunk-local block
                half2_t o01, o23;
                if (in_rotated)
                {
                    o01 = half2_t{ cur.xv.x, cur.xv.y }; o23 = half2_t{ cur.xv.z, cur.xv.w };
                }
                else
                {
                    half4_t xv = cur.xv;
                    if constexpr (MODE == G2_MODE_TABLE)
                    {
                        if (a.tbl.act_u)
                        {
                            // x = fp16(silu(g) * u): the arithmetic of silu_mul (exl3_elementwise.hip / activation.cu), per element
                            auto silu_mul1 = [] (half_t g, half_t u) -> half_t { float gf = (float) g; return f2h(gf / (1.0f + __expf(-gf)) * (float) u); };
                            xv = half4_t{ silu_mul1(xv.x, cur.wv.x), silu_mul1(xv.y, cur.wv.y), silu_mul1(xv.z, cur.wv.z), silu_mul1(xv.w, cur.wv.w) };
                        }
                    }
                    if constexpr (G2_IS_ACT(MODE))
                    {
                        // a = fp16(silu(g) * u) of this (row, block): split-k reduce of the gate / up slabs, output Hadamards, svh -- the arithmetic of
                        // glue_act_kernel (same device functions), done by the half-wave that needs the block
                        const int blk_abs = (k0s >> 7) + c0 + blk_l;
                        const half4_t svg = cur.svg, svu = cur.svu;
                        // slice-order sums (the order of slab_sum2 / glue_act_kernel): the prefetched lines first, any further slices from memory
                        float4_t vg = { 0.f, 0.f, 0.f, 0.f }, vu = vg;
                        #pragma unroll
                        for (int i = 0; i < ACT_PRE; ++i) if (i < a_act_S)
                        {
                            vg.x += cur.sg[i].x; vg.y += cur.sg[i].y; vg.z += cur.sg[i].z; vg.w += cur.sg[i].w;
                            vu.x += cur.su[i].x; vu.y += cur.su[i].y; vu.z += cur.su[i].z; vu.w += cur.su[i].w;
                        }
                        for (int sl = ACT_PRE; sl < a_act_S; sl += 4)        // further slices: 4 + 4 independent loads per round, slice-order sums
                        {
                            float4_t tg[4], tu[4];
                            const float* pg = a_act_g + ((size_t) blk_abs * a_act_S * m + row) * 128;
                            const float* pu = a_act_u + ((size_t) blk_abs * a_act_S * m + row) * 128;
                            #pragma unroll
                            for (int i = 0; i < 4; ++i)
                            {
                                tg[i] = ((const float4_t*) (pg + (size_t) min(sl + i, a_act_S - 1) * m * 128))[l32];
                                tu[i] = ((const float4_t*) (pu + (size_t) min(sl + i, a_act_S - 1) * m * 128))[l32];
                            }
                            #pragma unroll
                            for (int i = 0; i < 4; ++i) if (sl + i < a_act_S)
                            {
                                vg.x += tg[i].x; vg.y += tg[i].y; vg.z += tg[i].z; vg.w += tg[i].w;
                                vu.x += tu[i].x; vu.y += tu[i].y; vu.z += tu[i].z; vu.w += tu[i].w;
                            }
                        }
                        float g0, g1, g2, g3, u0, u1, u2, u3;
                        out_had(vg, l32, g0, g1, g2, g3);
                        out_had(vu, l32, u0, u1, u2, u3);
                        if (a_act_rs_new)
                        {
                            // gate / up were computed from fp16(x * w * r_prev): apply r_new / r_prev (exact sums of squares of the new residual)
                            const float rsc = gemv_rescale(a.act_rs, row, l32, cur.ss, cur.ssn);
                            g0 *= rsc; g1 *= rsc; g2 *= rsc; g3 *= rsc; u0 *= rsc; u1 *= rsc; u2 *= rsc; u3 *= rsc;
                        }
                        const half4_t gh = half4_t{ f2h(g0), f2h(g1), f2h(g2), f2h(g3) } * svg;
                        const half4_t uh = half4_t{ f2h(u0), f2h(u1), f2h(u2), f2h(u3) } * svu;
                        auto silu_mul = [] (half_t g, half_t u) -> half_t { float gf = (float) g; return f2h(gf / (1.0f + __expf(-gf)) * (float) u); };
                        xv = half4_t{ silu_mul(gh.x, uh.x), silu_mul(gh.y, uh.y), silu_mul(gh.z, uh.z), silu_mul(gh.w, uh.w) };
                    }
                    if constexpr (RES)
                    {
                        // resid_new of this (row, block): the arithmetic of glue_resid_kernel (same device functions, slice-order slab sum)
                        const int blk_abs = (k0s >> 7) + c0 + blk_l;
                        float4_t vy = { 0.f, 0.f, 0.f, 0.f };
                        #pragma unroll
                        for (int i = 0; i < SLAB_PRE; ++i) if (i < a.rs_S) { vy.x += cur.sg[i].x; vy.y += cur.sg[i].y; vy.z += cur.sg[i].z; vy.w += cur.sg[i].w; }
                        // further slices: independent loads in batches of 8 (a load-add chain costs a memory latency per slab line: 5 us at 16 slices)
                        for (int sl = SLAB_PRE; sl < a.rs_S; sl += 8)
                        {
                            float4_t ty[8];
                            const float* pl = a.rs_slab + ((size_t) blk_abs * a.rs_S * m + row) * 128;
                            #pragma unroll
                            for (int i = 0; i < 8; ++i) ty[i] = ((const float4_t*) (pl + (size_t) min(sl + i, a.rs_S - 1) * m * 128))[l32];
                            #pragma unroll
                            for (int i = 0; i < 8; ++i) if (sl + i < a.rs_S) { vy.x += ty[i].x; vy.y += ty[i].y; vy.z += ty[i].z; vy.w += ty[i].w; }
                        }
                        float h0, h1, h2, h3;
                        out_had(vy, l32, h0, h1, h2, h3);
                        const half4_t sc = cur.svg;
                        h0 *= (float) sc.x; h1 *= (float) sc.y; h2 *= (float) sc.z; h3 *= (float) sc.w;
                        xv = half4_t{ f2h((float) xv.x + h0), f2h((float) xv.y + h1), f2h((float) xv.z + h2), f2h((float) xv.w + h3) };
                        if (res_writer)
                        {
                            const float r0 = (float) xv.x, r1 = (float) xv.y, r2 = (float) xv.z, r3 = (float) xv.w;
                            float ssq = r0 * r0;
                            ssq = __builtin_fmaf(r1, r1, ssq); ssq = __builtin_fmaf(r2, r2, ssq); ssq = __builtin_fmaf(r3, r3, ssq);
                            #pragma unroll
                            for (int i = 1; i < 32; i <<= 1) ssq += xor_lane(ssq, i);
                            if (act)
                            {
                                ((half4_t*) (a.rs_resid_out + (size_t) row * a_k + (size_t) blk_abs * 128))[l32] = xv;
                                if (l32 == 0) a.rs_ss_out[(size_t) row * (a_k >> 7) + blk_abs] = ssq;
                            }
                        }
                    }
                    if constexpr (in_norm)
                    {
                        float r;
                        if (norm_in_task)
                        {
                            float s2 = cur.ss;
                            #pragma unroll
                            for (int i = 1; i < 32; i <<= 1) s2 += xor_lane(s2, i);
                            r = __frsqrt_rn((0.0f + s2) / (float) a_k + a_eps);
                        }
                        else r = rmf_s[row];
                        xv = half4_t{ f2h((float) xv.x * (float) cur.wv.x * r), f2h((float) xv.y * (float) cur.wv.y * r),
                                      f2h((float) xv.z * (float) cur.wv.z * r), f2h((float) xv.w * (float) cur.wv.w * r) };
                    }
                    xv = xv * cur.sv;
                    float h0 = (float) xv.x, h1 = (float) xv.y, h2 = (float) xv.z, h3 = (float) xv.w;
                    had128_f32x4(h0, h1, h2, h3, l32);
                    o01 = half2_t{ f2h(h0 * HAD_R_SCALE_128), f2h(h1 * HAD_R_SCALE_128) };
                    o23 = half2_t{ f2h(h2 * HAD_R_SCALE_128), f2h(h3 * HAD_R_SCALE_128) };
                }
#ifdef G2_DEBUG_FRAG
                // diagnostics build: workgroup 0 dumps the rotated activations it built (fp16 [blk][row][128]) at 40 MiB
                if (blockIdx.x == 0 && blockIdx.y == 0 && act)
                {
                    half_t* dbg = (half_t*) ((char*) a.ws_debug + (4ll << 20)) + ((size_t) (k0s / 128 + c0 + blk_l) * m + row) * 128 + 4 * l32;
                    dbg[0] = o01.x; dbg[1] = o01.y; dbg[2] = o23.x; dbg[3] = o23.y;
                }
#endif
                // elements 4*l32 .. +3 of the block: tile row r8 = l32 >> 2, rows 4*(l32&3) .. +3 of the tile
                const int r8 = l32 >> 2;
                if constexpr (RAW)
                {
                    // sum of the fp16 fragment values per (tile row, activation row): the FAST mul1 variant's bias term
                    float ts = ((float) o01.x + (float) o01.y) + ((float) o23.x + (float) o23.y);
                    ts += xor_lane(ts, 1);
                    ts += xor_lane(ts, 2);
                    if (act && (l32 & 3) == 0) tsum[(size_t) (blk_l * 8 + r8) * m + row] = ts;
                }
                if (act)
                {
                    const int q0 = 2 * (l32 & 1);
                    const int sp = (l32 >> 1) & 1;                          // slot pair: rows {2q,2q+1} (0) or {2q+8,2q+9} (1)
                    half_t* base = xa + ((size_t) (blk_l * 8 + r8) * m + row) * AH;
                    if constexpr (SPLIT)
                    {
                        half4_t d01 = { o01.x, o01.x, o01.y, o01.y }, d23 = { o23.x, o23.x, o23.y, o23.y };
                        *((half4_t*) (base + q0 * 8 + sp * 4)) = d01;
                        *((half4_t*) (base + (q0 + 1) * 8 + sp * 4)) = d23;
                    }
                    else
                    {
                        *((half2_t*) (base + q0 * 4 + sp * 2)) = o01;
                        *((half2_t*) (base + (q0 + 1) * 4 + sp * 2)) = o23;
                    }
                }
                if constexpr (G2_IS_ACT(MODE) || RES) { if (it + 1 < trips) nx = fetch(c0, cnt, it + 1); }
            }
        }
        __syncthreads();
        if (c0 == 0) { G2_T(2); }

        const int row_end = (c0 + cnt) * 8;                             // slice-local tile row bound of the chunk
        const int u_end = min(nunits_w, (PF * ubase < row_end) ? ((row_end / PF - 1 - ubase) / ustride + 1) : 0);   // this wave's first unit index beyond the chunk
        if constexpr (RAW)
        {
            // sum(x) over the tile rows this wave streams in this chunk, per activation row of the half-wave: lane i takes the wave's
            // i-th unit of the chunk, then a 32-lane butterfly (outside the streaming loop: no LDS wait on its critical path)
            #pragma unroll
            for (int p = 0; p < 2 * NG; ++p)
            {
                if (p < npass)
                {
                    const int rowp = min(2 * p + hw, m - 1);
                    float v = 0.0f;
                    for (int i = ui + l32; i < u_end; i += 32)
                    {
                        const float* tsr = tsum + (size_t) (PF * (ubase + i * ustride) - c0 * 8) * m + rowp;
                        #pragma unroll
                        for (int rr = 0; rr < PF; ++rr) v += tsr[rr * m];
                    }
                    #pragma unroll
                    for (int i = 1; i < 32; i <<= 1) v += xor_lane(v, i);
                    rowsum[p] += v;
                }
            }
        }

        // ---- pure streaming loop over this wave's units inside the chunk
        for (; ui < u_end; ++ui)                                       // plain counted loop: an early exit made the compiler drain vmcnt every iteration
        {
            const int unit = ubase + ui * ustride;
            const int row0 = PF * unit;
            const int nxt = min(unit + ustride, last_unit);                // next unit of this wave (clamped: a harmless reload at the end)
            #pragma unroll
            for (int u = 0; u < PF; ++u)
            {
                const int row = row0 + u;
                uint32_t Wx[K + 1];
                #pragma unroll
                for (int i = 0; i < K; ++i) Wx[i + 1] = ring[u].w[i];
#ifdef G2_ABL_NOBPERM
                Wx[0] = ring[u].w[K - 1] * 11u;                          // diagnostics build: no cross-lane carry
#else
#ifdef G2_CARRY_BPERMUTE
                Wx[0] = (uint32_t) __builtin_amdgcn_ds_bpermute(prev_lane_addr, (int) ring[u].w[K - 1]);
#else
                // carry-in = last word of the previous lane of the 8-lane tile group (lane c = 0 wraps to c = 7: the tile stream is circular).
                // Two DPP row rotates + a select (register file only) instead of ds_bpermute (LDS crossbar: -6 % on lm_head when ablated):
                // row_ror:1 gives lane i <- i - 1 within the 16-lane row, row_ror:9 gives i <- i - 9 = i + 7 (mod 16), i.e. the wrap for c = 0
                {
                    const uint32_t wl = ring[u].w[K - 1];
                    const uint32_t r1 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x121, 0xf, 0xf, true);
                    const uint32_t r9 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x129, 0xf, 0xf, true);
                    Wx[0] = (lane & 7) ? r1 : r9;
                }
#endif
#endif

                // refill the slot
#ifdef G2_ABL_HOT
                load_lane_words<K>(ring[u], strip + (size_t) ((PF * nxt + u) & 1) * row_stride);     // diagnostics build: the same two (cache-hot) rows every step
#else
                load_lane_words<K>(ring[u], strip + (size_t) (PF * nxt + u) * row_stride);
#endif

                // A fragments of this tile row for this lane's activation row
                const half_t* ap = arow + (size_t) (row - c0 * 8) * astep;
                half8_t af[AH / 8];
#ifdef G2_ABL_NOAFRAG
                #pragma unroll
                for (int i = 0; i < AH / 8; ++i) af[i] = half8_t{ 1, 2, 3, 4, 5, 6, 7, 8 };   // diagnostics build: no LDS fragment reads
                (void) ap;
#else
                #pragma unroll
                for (int i = 0; i < AH / 8; ++i) af[i] = ((const half8_t*) ap)[i];
#endif
                static_for<0, 4>([&] (auto qc)
                {
                    constexpr int q = decltype(qc)::value;
                    half4_t bc[2], bd[2];
                    // weights 8q..8q+3 -> column c ; 8q+4..8q+7 -> column c + 8
                    decode_quad<K, CB, VAR, 8 * q>(Wx, bc);
                    decode_quad<K, CB, VAR, 8 * q + 4>(Wx, bd);
                    if constexpr (SPLIT)
                    {
                        half8_t f = af[q];
                        half4_t a0 = { f[0], f[1], f[2], f[3] }, a1 = { f[4], f[5], f[6], f[7] };
                        static_for<0, NG>([&] (auto gc)
                        {
                            constexpr int gq = decltype(gc)::value;
                            acc_c[gq] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, bc[0], acc_c[gq], 4, gq, 0);
                            acc_d[gq] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, bd[0], acc_d[gq], 4, gq, 0);
                            acc_c[gq] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, bc[1], acc_c[gq], 4, gq, 0);
                            acc_d[gq] = __builtin_amdgcn_mfma_f32_4x4x4f16(a1, bd[1], acc_d[gq], 4, gq, 0);
                        });
                    }
                    else
                    {
                        half8_t f = af[q >> 1];
                        half4_t a0 = (q & 1) ? half4_t{ f[4], f[5], f[6], f[7] } : half4_t{ f[0], f[1], f[2], f[3] };
#ifdef G2_ABL_NOMFMA
                        asm volatile("" :: "v"(bc[0]), "v"(bd[0]), "v"(a0));     // diagnostics build: decode only, no MAC
#else
                        static_for<0, NG>([&] (auto gc)
                        {
                            constexpr int gq = decltype(gc)::value;
                            acc_c[gq] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, bc[0], acc_c[gq], 4, gq, 0);
                            acc_d[gq] = __builtin_amdgcn_mfma_f32_4x4x4f16(a0, bd[0], acc_d[gq], 4, gq, 0);
                        });
#endif
                    }
#ifndef G2_ABL_ILP
                    // bound live ranges: 8 weights in flight at a time (occupancy > ILP; letting the compiler overlap the quads in the 128-VGPR
                    // wave-per-column-block layouts changed nothing: 81.9 vs 82.2 us per layer)
                    __builtin_amdgcn_sched_barrier(0);
#endif
                });
            }
        }
    }

    G2_T(3);
    // ---- epilogue: per-wave partials -> LDS, cross-wave sum, output Hadamard
    if constexpr (RAW)
    {
        // row sums live in lane 0 / lane 32 of the wave (rows 2p / 2p+1): publish through the partial area
        float* sx = part + (size_t) nwv * MR * 128 + wave * MR;
        #pragma unroll
        for (int p = 0; p < 2 * NG; ++p) if (l32 == 0 && 2 * p + hw < MR) sx[2 * p + hw] = rowsum[p];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const float kinv = (float) u16_as_half(0x1eeeu), kbias = (float) u16_as_half(0xc931u);
        #pragma unroll
        for (int gq = 0; gq < NG; ++gq)
        {
            #pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                float sxv = sx[4 * gq + i];
                acc_c[gq][i] = acc_c[gq][i] * kinv + kbias * sxv;
                acc_d[gq][i] = acc_d[gq][i] * kinv + kbias * sxv;
            }
        }
    }
    {
        float* pw = part + (size_t) wave * MR * 128;
        const int col = 16 * T + c;
        #pragma unroll
        for (int gq = 0; gq < NG; ++gq)
        {
            #pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                const int row = 4 * gq + i;
                if (row < m) { pw[row * 128 + col] = acc_c[gq][i]; pw[row * 128 + col + 8] = acc_d[gq][i]; }
            }
        }
    }
    if constexpr (WPC)
    {
        // the wave owns its column block for the whole slice: its partials ARE the slab lines.  Through the wave's own LDS rows for 16-byte
        // coalesced stores (row r by half-wave r & 1); wave-level ordering only, no workgroup barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (wave_live)
        {
            const float* pw = part + (size_t) wave * MR * 128;
            float* slab = a.workspace + ws_off + ((size_t) cbl * a_S + s) * (size_t) m * 128;
            for (int row = hw; row < m; row += 2)
                ((float4_t*) (slab + row * 128))[l32] = ((const float4_t*) (pw + row * 128))[l32];
        }
#ifdef G2_TIMING
        if (tid == 0)
        {
            tstamp[4] = tstamp[5] = __builtin_amdgcn_s_memrealtime();
            uint64_t* dbg = (uint64_t*) a.ws_debug + (size_t) (blockIdx.y * gridDim.x + blockIdx.x) * 8;
            for (int i = 0; i < 6; ++i) dbg[i] = tstamp[i];
            uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            dbg[6] = xcc; dbg[7] = 0;
        }
#endif
        return;
    }
    __syncthreads();
    G2_T(4);

    const int l = tid & 31, hw8 = tid >> 5;
    const size_t wstride = (size_t) MR * 128;
    if (a_S > 1 || (a_flags & GEMV_OUT_DEFERRED))
    {
        float* slab = a.workspace + ws_off + ((size_t) cbl * a_S + s) * (size_t) m * 128;
        for (int row = hw8; row < m; row += nwv * 2)
        {
            const float* p0 = part + row * 128;
            float4_t v = ((const float4_t*) p0)[l];
            for (int w = 1; w < nwv; ++w)
            {
                float4_t t = ((const float4_t*) (p0 + w * wstride))[l];
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
#ifdef G2_SLAB_SC1
            st_agent(slab + row * 128 + 4 * l, v);                                         // experiment: write-through slabs in every mode
#else
            if constexpr (MODE == G2_MODE_TAIL)
            {
                if (a.epi.xcd_local) ((float4_t*) (slab + row * 128))[l] = v;              // reader is on this XCD: the L2 copy is enough
                else st_agent(slab + row * 128 + 4 * l, v);                                // read by another workgroup of this launch
            }
            else ((float4_t*) (slab + row * 128))[l] = v;
#endif
        }
#ifdef G2_TIMING
        if (tid == 0)
        {
            tstamp[5] = __builtin_amdgcn_s_memrealtime();
            uint64_t* dbg = (uint64_t*) a.ws_debug + (size_t) (blockIdx.y * gridDim.x + blockIdx.x) * 8;
            for (int i = 0; i < 6; ++i) dbg[i] = tstamp[i];
            uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            uint32_t hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            dbg[6] = xcc; dbg[7] = hwid;
        }
#endif
        if constexpr (MODE == G2_MODE_TAIL) gemv_tail(a, mi, cbl, cbg, tid, nwv, &s_tail_flag, (float4_t*) smem);
        return;
    }

    // output side resolved here (not kept live across the streaming loop: 64-VGPR budget)
    const half_t* svh; const half_t* bias = nullptr; void* C_m;
    float out_scale = HAD_R_SCALE_128;
    if constexpr (MODE == G2_MODE_TABLE)
    {
        const int slot = cbg / a.tbl.cbs_per_mat;
        const SlotRef_t sr = resolve_slot(a.tbl, slot);
        svh = (const half_t*) a.tbl.svh[sr.mat_index] + cbl * 128;
        C_m = a.c_fp32 ? (void*) ((float*) a.tbl.C + (size_t) slot * a.tbl.c_slot_stride) : (void*) ((half_t*) a.tbl.C + (size_t) slot * a.tbl.c_slot_stride);
        if (a.tbl.c_list) C_m = (void*) a.tbl.c_list[sr.mat_index];
        out_scale = HAD_R_SCALE_128 * sr.weight;                       // reference: scale *= weight, then one multiply (kernel.cuh:216-217)
    }
    else
    {
        svh = a.mat[mi].svh + cbl * 128;
        if (a.mat[mi].bias) bias = a.mat[mi].bias + cbl * 128;
        C_m = a.mat[mi].C;
    }
    for (int base = 0; base < m; base += nwv * 2)
    {
        int row = base + hw8;
        bool act = row < m;
        const float* p0 = part + (act ? row : 0) * 128;
        float4_t v = ((const float4_t*) p0)[l];
        for (int w = 1; w < nwv; ++w)
        {
            float4_t t = ((const float4_t*) (p0 + w * wstride))[l];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        float h0 = v.x, h1 = v.y, h2 = v.z, h3 = v.w;
        had128_f32x4(h0, h1, h2, h3, l);
        h0 *= out_scale; h1 *= out_scale; h2 *= out_scale; h3 *= out_scale;
        if (!act) continue;
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
}

// ------------------------------------------------------------------------------------------------
// launch (called from exl3_gemv.hip's dispatcher).  One translation unit per K: built with -DG2_K=1..8.
// ------------------------------------------------------------------------------------------------
#ifndef G2_K
#error "compile with -DG2_K=<bits per weight>"
#endif

template <int CB, int MODE>
static void launch_mode(int var, int ng, int nwv, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args)
{
    #define L(V, N) exl3_gemv2_kernel<G2_K, CB, V, N, MODE><<<grid, dim3(64 * nwv), lds, st>>>(args)
    if constexpr (G2_IS_ACT(MODE) || G2_IS_WPC(MODE))
    {
        // m <= 4 only (host-checked): one instantiation per variant
        if (var == 0) L(0, 1); else L(1, 1);
    }
    else
    {
        if (var == 0) { if (ng == 1) L(0, 1); else if (ng == 2) L(0, 2); else L(0, 4); }
        else          { if (ng == 1) L(1, 1); else if (ng == 2) L(1, 2); else L(1, 4); }
    }
    #undef L
}

template <int CB>
static void launch_cb(int var, int ng, int nwv, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args)
{
    if (args.tbl.B) launch_mode<CB, G2_MODE_TABLE>(var, ng, nwv, grid, lds, st, args);
    else if (args.cpw > 0)
    {
        // wave-per-column-block layout (host: deferred output, m <= 4, one chunk)
        if (args.flags & GEMV_IN_ACT) launch_mode<CB, G2_MODE_WACT>(var, ng, nwv, grid, lds, st, args);
        else if (args.flags & GEMV_IN_RESID) launch_mode<CB, G2_MODE_RNORM>(var, ng, nwv, grid, lds, st, args);
        else launch_mode<CB, G2_MODE_WPLAIN>(var, ng, nwv, grid, lds, st, args);
    }
    else if (args.flags & GEMV_IN_ACT) launch_mode<CB, G2_MODE_ACT>(var, ng, nwv, grid, lds, st, args);
    else if (args.epi.mode != GEMV_EPI_NONE) launch_mode<CB, G2_MODE_TAIL>(var, ng, nwv, grid, lds, st, args);
    else if (args.flags & GEMV_IN_NORM) launch_mode<CB, G2_MODE_NORM>(var, ng, nwv, grid, lds, st, args);
    else                                launch_mode<CB, G2_MODE_PLAIN>(var, ng, nwv, grid, lds, st, args);
}

#define G2_CAT_(a, b) a##b
#define G2_CAT(a, b) G2_CAT_(a, b)

void G2_CAT(exl3_gemv2_launch_k, G2_K)(int cb, int var, int ng, int nwv, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args)
{
    if (cb == 0) launch_cb<0>(var, ng, nwv, grid, lds, st, args);
    else if (cb == 1) launch_cb<1>(var, ng, nwv, grid, lds, st, args);
    else launch_cb<2>(var, ng, nwv, grid, lds, st, args);
}

#if G2_K == 4
size_t exl3_gemv2_lds_bytes(int ng, int var, int cb, int nwv, int m, int chunk_blocks)
{
    const int MR = 4 * ng;
    const int AH = (var == 1 && cb != 2) ? 32 : 16;
    size_t frag = ((size_t) chunk_blocks * 8 * m * AH * 2 + 15) & ~(size_t) 15;
    size_t part = (size_t) nwv * MR * 128 * 4 + (size_t) nwv * MR * 4;
    size_t tsum = (size_t) chunk_blocks * 8 * m * 4 + 64;          // tile-row sums + 1/rms per row
    return frag + part + tsum;
}
#endif
