// EXL3 quantized GEMV for gfx950, kernel generation 4: 1..4 rows (decode at batch 1..4), "no per-wave prologue".
//
//   C = ((A * suh) H) @ dequant(B) H * svh (+ bias)        reference: quant/exl3_gemv_kernel.cuh:138-402 (semantics only)
//
// Why it exists (profiles/r03_gemv_valu_breakdown.json): at batch 1 a generation-2 wave streams only 2..4 work units (a unit =
// 2 tile rows = 64 weights per lane = 235 VALU instructions), and every wave of every workgroup ran ~450 VALU instructions of prologue and
// epilogue around them (activation-fragment tasks executed by all waves whether they own a task or not, a run-time division, RAW row sums
// per wave, chunk bookkeeping, SGPR spills): 5.5 issued VALU per weight against 3.4 in the loop, in a kernel that is bound by VALU issue.
// Generation 4 keeps generation 2's streaming loop (column pair per lane, compile-time bit windows, v_mfma_f32_4x4x4_16B_f16 A-broadcast)
// and removes what surrounded it:
//
//   * The A operand of the 4x4x4 A-broadcast MFMA comes from lanes 4*abid .. 4*abid+3.  Generation 2 used abid 0..3 for rows 0..15 and fetched
//     the fragment of the current tile row from LDS in every unit.  Here abid addresses ACTIVATION QUADS instead: lane 4g+i of ONE VGPR pair
//     holds the 4 halves of (tile row g>>2, quad g&3) for row i, so a single 8-byte-per-lane read (one ds_read_b64, lane-linear: 512
//     contiguous bytes, conflict-free) covers 4 tile rows x 4 rows of activations, and the 32 MFMAs of those 4 tile rows pick their quad
//     with the immediate abid = 0..15.  No LDS traffic in the unit, no per-row fragment address.
//   * Pre-rotated input (GEMV_IN_ROTATED, the producer left fp16 had(x * suh) and per-block sums): the wave loads its quads straight from
//     global memory into that register layout -- no LDS, no workgroup barrier before streaming, no preparation code at all.
//   * Raw / RMSNorm input: the (block, row) Hadamard tasks are executed only by the waves that own one (m = 1, 4 blocks: 2 of 4 waves),
//     the others go straight to the barrier with their weight rows in flight.
//   * mul1 FAST variant: the affine map k_inv * acc + k_bias * sum(x) is applied once per workgroup by the half-wave that reduces the
//     waves' partials (block sums from the producer / the task), not per wave from per-tile-row sums.
//   * No chunks (slice <= 32 Hadamard blocks), no division, no tables, no tail epilogues, no wave-per-column-block layout: those stay
//     generation 2's (exl3_gemv2.kspec.hip), which remains the kernel for 5..16 rows, MoE tables and the opt-in pipelines.
//
// Grid = (k-slices S, column blocks); workgroup = nwv waves that split the slice's units into contiguous ranges; partial sums meet in
// LDS; output = deferred slabs [colblock][S][m][128] fp32 (S > 1 or GEMV_OUT_DEFERRED) or the final rows (S == 1).
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include "exl3_gemv_args.h"
#include "exl3_lane_decode.cuh"
#include "exl3_glue_device.cuh"

#include <type_traits>

template <int I, int N, typename F>
__device__ __forceinline__ void g4_static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); g4_static_for<I + 1, N>(f); }
}

#define G4_MODE_ROT  0      // GEMV_IN_ROTATED: mat[i].xh (+ xsum for the mul1 FAST variant)
#define G4_MODE_RAWX 1      // raw x: x * suh -> 128-point Hadamard in the task
#define G4_MODE_NORM 2      // GEMV_IN_NORM: RMSNorm of the residual stream, then as RAWX
#define G4_MODE_ACT  3      // GEMV_IN_ACT: silu(g) * u finished from the producer's gate / up slabs, then as RAWX
#define G4_MODE_NORMFX 4    // GEMV_IN_NORM | GEMV_IN_FX: as NORM, the residual read from the 64-bit fixed-point accumulator (row scale = the previous residual's)
#define G4_MODE_ACTFX 5     // GEMV_IN_ACTFX: silu(g) * u from the fixed-point gate / up accumulators a GEMV_OUT_ATOMIC gate|up launch added into
// Table modes (MoE / indexed exl3_mgemm, a.tbl): grid = (k-slices, column blocks of ONE matrix, slots); slot -> (matrix, routing weight) resolved on the
// device from the router's index list; the input side is RAWX's / ACT's
#define G4_MODE_TRAWX 6     // slot input = raw x (one shared row set, or a_slot_stride apart)
#define G4_MODE_TACT 7      // slot input = silu(g) * u from the slabs a TRAWX gate|up launch over [gate_0..gate_E-1, up_0..up_E-1] left (slot j: gate, slot bszm + j: up)
#define G4_MODE_ATTM 8      // GEMV_IN_ATTM: the flash-decoding merge of the attention's context-split partials, then as RAWX (a.A is not read)
#define G4_MODE_QKVM 9      // GEMV_IN_QKVM: q block of the q|k|v launch's slabs finished in the task (+ the K / V append as a side job), then as RAWX (a.A is not read)
constexpr int g4_input_mode(int MODE) { return MODE == G4_MODE_TRAWX ? G4_MODE_RAWX : (MODE == G4_MODE_TACT ? G4_MODE_ACT : MODE); }

constexpr int g4_waves_per_eu(int K, int CB, int MODE_)
{
    const int MODE = g4_input_mode(MODE_);
    if (MODE == G4_MODE_QKVM) return 2;                      // eight slab lines + rope values travel with a task (as ATTM: these launches are one wave per SIMD anyway)
    if (MODE == G4_MODE_ATTM) return 2;                      // eight 16-byte split outputs travel with a task, two tasks in flight in the pipelined task loop: 128 VGPRs spill (4 VGPRs + 79 SGPRs at budget 4); o_proj launches are one wave per SIMD anyway
    if (MODE == G4_MODE_ACT) return MODE_ == G4_MODE_TACT ? 4 : 3;     // 8 slab lines travel with a task; the non-table form also carries the row-scale correction
    if (MODE == G4_MODE_ACTFX && K >= 5) return 5;          // four 16-byte accumulator loads in flight per task next to a 10..16-word ring
    if (K >= 5) return 6;
    if (MODE == G4_MODE_NORMFX || MODE == G4_MODE_ACTFX) return 6;
    return (MODE == G4_MODE_ROT && CB == EXL3_CB_MUL1) ? 8 : 7;
}
// A/B builds: G4_WPE_DELTA = n asks for n fewer waves per SIMD (a larger register budget) in every instantiation
#ifndef G4_WPE_DELTA
#define G4_WPE_DELTA 0
#endif

// One work unit = 2 tile rows of the wave's column block.  HALF selects which half of the 4-tile-row activation group the unit is (abid 0..7 or 8..15).
// Units in flight per wave (weight-row register ring = 2 * PFU rows).  One unit of lookahead, like generation 2.  Measured (round 3, same box,
// three alternations, gpurun_out/r3d): TWO units in flight -- tried because a q|k|v wave streams only 2 units and tools/gemv_timeline.py shows it
// waiting for its second unit's rows -- is 6 % SLOWER on the whole step (535 vs 568 tok/s): every wave of a launch issues its first requests in the
// same microsecond, and doubling that burst delays every dependent small load (activations, slabs) behind twice as many weight lines.
#ifndef G4_PFU
#define G4_PFU(K) 1
#endif

template <int K, int CB, int VAR, int HALF, int NR>
__device__ __forceinline__ void g4_unit(LaneWords<K> (&ringall)[NR], const uint32_t* __restrict__ refill, size_t row_stride, int lane,
                                        half4_t ag0, half4_t ag1, float4_t& acc_c, float4_t& acc_d)
{
    constexpr bool SPLIT = (VAR == 1) && (CB != EXL3_CB_MUL1);
    constexpr int R0 = NR == 4 ? 2 * HALF : 0;               // two units in flight: the unit's slot pair alternates with HALF
    g4_static_for<0, 2>([&] (auto uc)
    {
        constexpr int u = decltype(uc)::value;
        LaneWords<K>* ring = ringall + R0;
        uint32_t Wx[K + 1];
        #pragma unroll
        for (int i = 0; i < K; ++i) Wx[i + 1] = ring[u].w[i];
        // carry-in = last word of the previous lane of the 8-lane tile group (lane c = 0 wraps to c = 7: the tile stream is circular): two DPP row
        // rotates + a select, register file only (exl3_gemv2.kspec.hip)
        {
            const uint32_t wl = ring[u].w[K - 1];
            const uint32_t r1 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x121, 0xf, 0xf, true);
            const uint32_t r9 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x129, 0xf, 0xf, true);
            Wx[0] = (lane & 7) ? r1 : r9;
        }
        load_lane_words<K>(ring[u], refill + (size_t) u * row_stride);           // refill the slot with the next unit's row
        g4_static_for<0, 4>([&] (auto qc)
        {
            constexpr int q = decltype(qc)::value;
            constexpr int ABID = 8 * HALF + 4 * u + q;
            half4_t bc[2], bd[2];
            decode_quad<K, CB, VAR, 8 * q>(Wx, bc);                               // weights 8q..8q+3 -> column c
            decode_quad<K, CB, VAR, 8 * q + 4>(Wx, bd);                           // 8q+4..8q+7 -> column c + 8
            acc_c = __builtin_amdgcn_mfma_f32_4x4x4f16(ag0, bc[0], acc_c, 4, ABID, 0);
            acc_d = __builtin_amdgcn_mfma_f32_4x4x4f16(ag0, bd[0], acc_d, 4, ABID, 0);
            if constexpr (SPLIT)
            {
                acc_c = __builtin_amdgcn_mfma_f32_4x4x4f16(ag1, bc[1], acc_c, 4, ABID, 0);
                acc_d = __builtin_amdgcn_mfma_f32_4x4x4f16(ag1, bd[1], acc_d, 4, ABID, 0);
            }
            // 8 weights in flight at a time (occupancy over ILP).  A/B builds: G4_NO_SCHED = free scheduling, G4_SCHED2 = 16 weights in flight
#if defined(G4_NO_SCHED)
#elif defined(G4_SCHED2)
            if constexpr (q & 1) __builtin_amdgcn_sched_barrier(0);
#else
            __builtin_amdgcn_sched_barrier(0);
#endif
        });
    });
}

template <int K, int CB, int VAR, int XMODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(g4_waves_per_eu(K, CB, XMODE) - G4_WPE_DELTA)))
void exl3_gemv4_kernel(const GemvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MODE = g4_input_mode(XMODE);                 // the input side of a table launch is RAWX's / ACT's
    constexpr bool TBL = XMODE == G4_MODE_TRAWX || XMODE == G4_MODE_TACT;
    constexpr bool SPLIT = (VAR == 1) && (CB != EXL3_CB_MUL1);
    constexpr bool RAW = (VAR == 1) && (CB == EXL3_CB_MUL1);
    constexpr int NW = 8 * K;
    constexpr bool IN_LDS = MODE != G4_MODE_ROT;

    // every prologue scalar in one batch of scalar loads out of the first two lines of the argument block (exl3_gemv_args.h)
    const int a_S = a.S, a_k = a.k, a_kslice = a.kslice, a_flags = a.flags, a_nm = a.num_mats, nwv = a.nwv, m = a.m;
    const int a_cbf[GEMV_MAX_MATS] = { 0, a.cbf[0], a.cbf[1], a.cbf[2] };
    const uint32_t mg_m = a.magic_m, mg_nwv = a.magic_nwv;
    const half_t* a_A = a.A;
    const half_t* const a_norm_w = a.norm_w;
    const float* const a_ss_part = a.ss_part;
    const float a_eps = a.eps;
    const float* a_act_g = a.act_g; const float* a_act_u = a.act_u;
    const half_t* a_act_svh_g = a.act_svh_g; const half_t* a_act_svh_u = a.act_svh_u;
    const int a_act_S = a.act_S;
    if constexpr (MODE == G4_MODE_ACT) asm volatile("" :: "s"(a_act_g), "s"(a_act_S));
    {
        const void* t0 = a.mat[0].B; const void* t1 = a.mat[1].B; const void* t2 = a.mat[2].B; const void* t3 = a.mat[3].B; const void* t4 = a.mat[3].xsum;
        asm volatile("" :: "s"(t0), "s"(t1), "s"(t2), "s"(t3), "s"(t4));     // the workgroup's matrix record (a dependent load) then hits the scalar cache
    }
#ifdef G4_TIMING
    // diagnostics build: wave 0 of every workgroup leaves 100 MHz timestamps of its phases in the workspace tail (48 MiB offset; tools/gemv_timeline.py)
    uint64_t tstamp[6];
    tstamp[0] = __builtin_amdgcn_s_memrealtime();
    const uint64_t cyc0 = __builtin_amdgcn_s_memtime();
    #define G4_T(i) tstamp[i] = __builtin_amdgcn_s_memrealtime()
#else
    #define G4_T(i)
#endif
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int s = blockIdx.x, cbg = blockIdx.y;
    int mi = 0;
    #pragma unroll
    for (int i = 1; i < GEMV_MAX_MATS; ++i) if (!TBL && i < a_nm && cbg >= a_cbf[i]) mi = i;
    const uint32_t* __restrict__ Bm = a.mat[mi].B;
    const half_t* __restrict__ suh = a.mat[mi].suh;
    int n = a.mat[mi].n, cbl = cbg - a.mat[mi].cb_first, ws_off = a.mat[mi].ws_offset;
    int slot = 0, tmat = 0; float tweight = 1.0f;
    if constexpr (TBL)
    {
        slot = blockIdx.z;
        const SlotRef_t sr = resolve_slot(a.tbl, slot);
        if (sr.mat_index < 0) return;                           // slot filtered out by the expert range (whole workgroup, before any barrier)
        tmat = sr.mat_index; tweight = sr.weight;
        Bm = (const uint32_t*) a.tbl.B[tmat];
        suh = (const half_t*) a.tbl.suh[tmat];
        n = a.tbl.n; cbl = cbg; ws_off = slot * a.tbl.cbs_per_mat * a_S * m * 128;
        if constexpr (MODE == G4_MODE_RAWX) a_A += (size_t) slot * a.tbl.a_slot_stride;
        if constexpr (MODE == G4_MODE_ACT)
        {
            // gate / up slabs of this slot [slot][k/128][act_S][m][128] (act_u = the same launch's slot bszm + j); their svh from the gate|up table
            const size_t sstride = (size_t) (a_k >> 7) * a_act_S * m * 128;
            a_act_g += (size_t) slot * sstride; a_act_u += (size_t) slot * sstride;
            a_act_svh_g = (const half_t*) a.tbl.act_svh[tmat];
            a_act_svh_u = (const half_t*) a.tbl.act_svh[tmat + a.tbl.act_up_off];
        }
    }
    const int tiles_n = n >> 4;
    const int k0s = s * a_kslice;
    const int k1s = min(k0s + a_kslice, a_k);
    const int nb = (k1s - k0s) >> 7;                         // Hadamard blocks of the slice (<= 32, host-checked)
    const int units = nb * 4;
    const int ubase = gemv_udiv(units * wave, mg_nwv);
    const int nun = gemv_udiv(units * (wave + 1), mg_nwv) - ubase;     // this wave's units: [ubase, ubase + nun)
    const int l32 = lane & 31, hwid = tid >> 5, nhw = nwv * 2;

    // LDS: activation quads [tile row][quad][row < m] x 8 bytes (+ one group of slack for the over-read of a partial last group) | block sums
    // [nb][m] fp32 | partial sums [nwv][m][128] fp32, which REUSE the quad area after the streaming loop (barrier in between)
    const int quad_bytes = IN_LDS ? (nb * 8 + 4) * 4 * m * 8 : 0;
    const int part_bytes = nwv * m * 512;
    char* quads = smem;
    float* part = (float*) smem;
    float* bsum = (float*) (smem + (((quad_bytes > part_bytes ? quad_bytes : part_bytes) + 15) & ~15));

    // ---- streaming state
    const int T = lane >> 3, c = lane & 7;
    const size_t row_stride = (size_t) tiles_n * NW;
    const uint32_t* __restrict__ strip = Bm + ((size_t) (k0s >> 4) * tiles_n + (size_t) cbl * 8) * NW + (size_t) lane * K;
    const int last_unit = ubase + (nun > 0 ? nun - 1 : 0);
#if defined(G4_ABL_HOT_FIRST) || defined(G4_ABL_HOT_ALL)
    // speed-only ablations (results are garbage): the wave's FIRST ring of weight rows (HOT_FIRST) or every weight row (HOT_ALL) is read from the
    // matrix's first tile rows -- the same 2 KB per lane position for every workgroup of the launch, i.e. L2-resident after the first touch per XCD.
    // Upper bound for "the predecessor's drain warms the successor's first rows" (VERDICT r3 task 5 i); profiles/r04_bs1_variant_queue.txt
    const uint32_t* __restrict__ strip_hot = Bm + (size_t) lane * K;
#endif

    // ---- preparation tasks (raw / norm / act input): task t = (block t / m, row t % m), one per half-wave; only waves that own a task run them
    // (ACT: the first 4 slab lines of gate and up and their svh travel with the task, i.e. they are requested BEFORE the wave's first weight rows -- loaded
    // inside the task they queued behind those rows)
    constexpr int NSL = MODE == G4_MODE_ACT ? 4 : 1;
#ifndef G4_ATTM_NMO
#define G4_ATTM_NMO 8
#endif
    constexpr int NMO = MODE == G4_MODE_ATTM ? G4_ATTM_NMO : 1;         // ATTM: the first 8 splits' outputs travel with the task (16: 254 VGPRs and a private segment -- the pipelined task loop holds two tasks)
    constexpr int NQL = MODE == G4_MODE_QKVM ? 8 : 1;         // QKVM: the first 8 slab lines of the q block travel with the task
    struct PrepIn { half4_t xv, sv, wv; float ss, ssn; uint4_t f0, f1, f2, f3; float4_t ga[NSL], ua[NSL]; half4_t svg, svu; float4_t mo[NMO]; float2 mst;
                    float4_t ql[NQL], sn4, cs4; };
    const int at_nsplit = MODE == G4_MODE_ATTM ? a.attm.nsplit : 1, at_gq = MODE == G4_MODE_ATTM ? a.attm.gq : 1;
    const float* const at_part = MODE == G4_MODE_ATTM ? a.attm.part : nullptr;
    // partial records of (row, Hadamard block of o_proj's input).  head_dim 128: the block is query head `blk` = record (kv block blk / gq, query index
    // blk % gq), lane l32 owns accumulators 4 l32 .. + 3 and the statistics pair at offset 0.  head_dim 64: the block holds query heads 2 blk and 2 blk + 1
    // (lanes 0-15 / 16-31); head qh belongs to kv head qh / gq = half `kv & 1` of the records of kv block kv >> 1 -- per-lane record base, statistics at
    // offset 2 * half, accumulators at 4 + 64 * half + 4 * (l32 & 15)
    const bool at_hd64 = MODE == G4_MODE_ATTM && a.attm.hd == 64;
    const int at_lr = at_hd64 ? (l32 & 15) : l32;                                      // lane inside its head: holds the statistics of split at_lr
    struct AttmRec { const float* p; int st_off, acc_off; };
    auto attm_rec = [&] (int row, int blk) -> AttmRec
    {
        const int qh = at_hd64 ? 2 * blk + (l32 >> 4) : blk;
        const int kv = gemv_udiv(qh, a.attm.magic_gq), i = qh - kv * at_gq;
        const int h = at_hd64 ? kv >> 1 : kv, half = at_hd64 ? (kv & 1) : 0;
        AttmRec r;
        r.p = at_part + ((((size_t) row * a.attm.blocks + h) * at_gq + i) * at_nsplit) * 132;
        r.st_off = 2 * half; r.acc_off = 4 + 64 * half + 4 * at_lr;
        return r;
    };
    const int ntask = nb * m;
    auto fetch = [&] (int it) -> PrepIn
    {
        PrepIn r; r.xv = half4_t{ 0, 0, 0, 0 }; r.sv = r.xv; r.wv = r.xv; r.ss = 0.0f; r.ssn = 0.0f;
        const int t = min(it * nhw + hwid, ntask - 1);
        const int blk = gemv_udiv(t, mg_m), row = t - blk * m;
        const size_t kofs = (size_t) k0s + 128 * blk;
        if constexpr (MODE == G4_MODE_RAWX || MODE == G4_MODE_NORM) r.xv = ((const half4_t*) (a_A + (size_t) row * a_k + kofs))[l32];
        if constexpr (MODE == G4_MODE_NORMFX)
        {
            const uint4_t* fp = (const uint4_t*) ((const int64_t*) a_A + (size_t) row * a_k + kofs) + 2 * l32;      // 4 x int64 per lane
            r.f0 = fp[0]; r.f1 = fp[1];
        }
        if constexpr (MODE == G4_MODE_ACTFX)
        {
            const uint4_t* gp = (const uint4_t*) ((const int64_t*) a_act_g + (size_t) row * a_k + kofs) + 2 * l32;
            const uint4_t* up = (const uint4_t*) ((const int64_t*) a_act_u + (size_t) row * a_k + kofs) + 2 * l32;
            r.f0 = gp[0]; r.f1 = gp[1]; r.f2 = up[0]; r.f3 = up[1];
        }
        r.sv = ((const half4_t*) (suh + kofs))[l32];
        if constexpr (MODE == G4_MODE_ATTM)
        {
            const AttmRec rc = attm_rec(row, (k0s >> 7) + blk);
            r.mst = *((const float2*) (rc.p + (size_t) min(at_lr, at_nsplit - 1) * 132 + rc.st_off));      // lane s of a head: {m, l} of split s (masked at its use)
            #pragma unroll
            for (int u = 0; u < NMO; ++u) r.mo[u] = *((const float4_t*) (rc.p + (size_t) min(u, at_nsplit - 1) * 132 + rc.acc_off));
        }
        if constexpr (MODE == G4_MODE_QKVM)
        {
            // q block blk_abs of this row: its first slab lines, column scales, the row's rope values and rescale sums -- requested here, i.e. before the
            // wave's first weight rows (inside the task they would queue behind those)
            const int blk_abs = (k0s >> 7) + blk, qS = a.qkvm.S;
            const float* pq = a.qkvm.sq + ((size_t) blk_abs * qS * m + row) * 128;
            #pragma unroll
            for (int i = 0; i < NQL; ++i) r.ql[i] = ((const float4_t*) (pq + (size_t) min(i, qS - 1) * m * 128))[l32];
            r.svg = ((const half4_t*) (a.qkvm.svh_q + blk_abs * 128))[l32];
            const int ph = a.qkvm.hd >> 3;
            r.sn4 = float4_t{ 0.f, 0.f, 0.f, 0.f }; r.cs4 = r.sn4;
            if (a.qkvm.rope_mode == 2)
            {
                const int f = 4 * (l32 & (ph - 1));
                r.sn4 = *((const float4_t*) (a.qkvm.rope_sin + row * 64 + f)); r.cs4 = *((const float4_t*) (a.qkvm.rope_cos + row * 64 + f));
            }
            else
            {
                const int f = 2 * (l32 & ((a.qkvm.hd >> 2) - 1));
                r.sn4.x = a.qkvm.rope_sin[row * 64 + f]; r.sn4.y = a.qkvm.rope_sin[row * 64 + f + 1]; r.cs4.x = a.qkvm.rope_cos[row * 64 + f]; r.cs4.y = a.qkvm.rope_cos[row * 64 + f + 1];
            }
            if (a.qkvm.rs.ss_new)
            {
                const int nbh = a.qkvm.rs.k >> 7;
                r.ss = a.qkvm.rs.ss_prev[(size_t) row * nbh + min(l32, nbh - 1)]; r.ssn = a.qkvm.rs.ss_new[(size_t) row * nbh + min(l32, nbh - 1)];
                if (l32 >= nbh) { r.ss = 0.0f; r.ssn = 0.0f; }
            }
        }
        if constexpr (MODE == G4_MODE_ACT)
        {
            const int blk_abs = (k0s >> 7) + blk;
            const float* pa = a_act_g + ((size_t) blk_abs * a_act_S * m + row) * 128;
            const float* pb = a_act_u + ((size_t) blk_abs * a_act_S * m + row) * 128;
            #pragma unroll
            for (int i = 0; i < NSL; ++i)
            {
                const size_t so = (size_t) min(i, a_act_S - 1) * m * 128;
                r.ga[i] = ((const float4_t*) (pa + so))[l32]; r.ua[i] = ((const float4_t*) (pb + so))[l32];
            }
            r.svg = ((const half4_t*) (a_act_svh_g + blk_abs * 128))[l32];
            r.svu = ((const half4_t*) (a_act_svh_u + blk_abs * 128))[l32];
        }
        if constexpr (!TBL && (MODE == G4_MODE_ACT || MODE == G4_MODE_ACTFX))
        {
            // gate / up came from a launch that normalised with the previous residual's 1/rms (GEMV_IN_RESID / GEMV_IN_FX): the first 32 block sums
            // of squares of both residuals travel with the task (gemv_rescale).  (Table launches never carry a rescale.)
            if (a.act_rs.ss_new)
            {
                const int nbh = a.act_rs.k >> 7;
                if (l32 < nbh) { r.ss = a.act_rs.ss_prev[(size_t) row * nbh + l32]; r.ssn = a.act_rs.ss_new[(size_t) row * nbh + l32]; }
            }
        }
        if constexpr (MODE == G4_MODE_NORM || MODE == G4_MODE_NORMFX)
        {
            r.wv = ((const half4_t*) (a_norm_w + kofs))[l32];
            // unconditional (clamped index, masked where it is used): behind a lane condition the compiler cannot count the outstanding loads and waits
            // with vmcnt(0) in front of the first task -- i.e. for the wave's first WEIGHT rows, which were requested after these operands precisely
            // so that the task would run underneath their latency (ISA + tools/gemv_timeline.py, round 3: 1.7 / 5.4 us from "loads issued" to "prep done")
            r.ss = a_ss_part[(size_t) row * (a_k >> 7) + min(l32, (a_k >> 7) - 1)];
        }
        return r;
    };
    const bool prep_wave = IN_LDS && 2 * wave < ntask;
    PrepIn nx;
    if constexpr (IN_LDS) { if (prep_wave) nx = fetch(0); }

    // first weight rows: requested after the first task's (small, L2-resident) operands so that the task computes underneath the HBM latency
    constexpr int PFU = G4_PFU(K), NR = 2 * PFU;
    LaneWords<K> ring[NR];
#ifdef G4_ABL_PREP_FIRST
    // experiment: a wave that owns a preparation task requests its weight rows only after the task's operands have arrived
    if constexpr (IN_LDS) { if (prep_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
    auto issue_ring = [&] ()
    {
#ifdef G4_ABL_ONE_ROW
        // experiment: only the first row before the preparation barrier (half the initial burst of weight requests); the second after it
        if (nun > 0) load_lane_words<K>(ring[0], strip + (size_t) (2 * ubase) * row_stride);
#else
        // unconditional (the host gives every wave at least one unit; the row index is clamped into the slice regardless): under `if (nun > 0)` the number
        // of outstanding loads after the merge is path-dependent and the waits of the preparation task below degrade to vmcnt(0)
#ifdef G4_RAMP
        // experiment (G4_PFU = 2): a two-unit ring whose second unit is requested only in front of the streaming loop (issue_ring_second): the deeper
        // lookahead of the steady state without doubling the launch's initial burst of weight requests.  Measured in the whole step (round 3, same box, two
        // alternations, tools/experiments/decode_step_harness.hip): 603.3 / 603.6 vs 622.2 / 630.9 tok/s -- slower like the plain two-unit ring; not the default
        #pragma unroll
        for (int u = 0; u < 2; ++u) load_lane_words<K>(ring[u], strip + (size_t) min(min(2 * ubase + u, 2 * last_unit + 1), 2 * units - 1) * row_stride);
#else
#if defined(G4_ABL_HOT_FIRST) || defined(G4_ABL_HOT_ALL)
        #pragma unroll
        for (int u = 0; u < NR; ++u) load_lane_words<K>(ring[u], strip_hot + (size_t) (u & 1) * row_stride);
#else
        #pragma unroll
        for (int u = 0; u < NR; ++u) load_lane_words<K>(ring[u], strip + (size_t) min(min(2 * ubase + u, 2 * last_unit + 1), 2 * units - 1) * row_stride);
#endif
#endif
#endif
    };
#ifdef G4_RAMP
    auto issue_ring_second = [&] ()
    {
        #pragma unroll
        for (int u = 2; u < NR; ++u) load_lane_words<K>(ring[u], strip + (size_t) min(min(2 * ubase + u, 2 * last_unit + 1), 2 * units - 1) * row_stride);
    };
#endif
    if constexpr (IN_LDS) issue_ring();              // (rotated input: behind the first activation group, below)
    G4_T(1);
    // ---- activation quads of this wave's first group
    // lane 4g + i: tile row (group base + (g >> 2)), quad g & 3, row min(i, m - 1)
    const int gq = lane >> 2, gi = min(lane & 3, m - 1);
    half4_t agc0 = { 0, 0, 0, 0 }, agc1 = agc0;               // current group (agc1: the duplicated second pair of the SPLIT variants)
    uint2_t agn = { 0u, 0u };                                 // next group, raw
    const half_t* xh_lane = nullptr;
    int quad_lane = 0;
    const int tr_last = nb * 8 - 1;
    auto load_group = [&] (int tr0) -> uint2_t                 // tr0: slice-local first tile row of the group
    {
        uint2_t r;
        if constexpr (IN_LDS) r = *((const uint2_t*) (quads + (size_t) tr0 * 4 * m * 8 + quad_lane));
        else
        {
            const int tr = min(tr0 + (gq >> 2), tr_last);
            const uint32_t* p = (const uint32_t*) (xh_lane + 16 * tr);
            r.x = p[0]; r.y = p[4];                           // halves {2q, 2q+1} and {2q+8, 2q+9} of the tile row
        }
        return r;
    };
    auto set_group = [&] (uint2_t raw)
    {
        if constexpr (SPLIT)
        {
            // weight = lo + hi fed as two k-slots against a duplicated activation: (x0, x0, x1, x1), (x2, x2, x3, x3)
            agc0 = u2_as_half4(__builtin_amdgcn_perm(raw.x, raw.x, 0x01000100u), __builtin_amdgcn_perm(raw.x, raw.x, 0x03020302u));
            agc1 = u2_as_half4(__builtin_amdgcn_perm(raw.y, raw.y, 0x01000100u), __builtin_amdgcn_perm(raw.y, raw.y, 0x03020302u));
        }
        else agc0 = u2_as_half4(raw.x, raw.y);
    };
    // mul1 FAST: sum of the slice's rotated activations per row, for the reducing half-waves (rows hwid < m); ROT: from the producer's block sums.  A plain
    // load for every lane (clamped indices, masked in the epilogue): an accumulating loop here waits for each value on the spot
    float xs_pre = 0.0f;
    if constexpr (RAW && !IN_LDS) xs_pre = a.mat[mi].xsum[(size_t) min(hwid, m - 1) * (a_k >> 7) + (k0s >> 7) + min(l32, nb - 1)];
    if constexpr (!IN_LDS)
    {
        // Order of the first requests = the order of a streaming trip (activation group, then weight rows).  The compiler's wait counts at the loop
        // head are the minimum over the entry path and the back edge: with the group requested AFTER the first weight rows it was the youngest load on
        // entry, the head waited with vmcnt(1) / vmcnt(0) in EVERY trip, i.e. for the weight rows refilled a few hundred cycles earlier (ISA, round 3).
        xh_lane = a.mat[mi].xh + (size_t) gi * a_k + k0s + 2 * (gq & 3);
        agn = load_group(2 * ubase);
        issue_ring();
#ifdef G4_ABL_ONE_ROW
        if (nun > 0) load_lane_words<K>(ring[1], strip + (size_t) (2 * ubase + 1) * row_stride);
#endif
    }
    else quad_lane = (gq * m + gi) * 8;

    if constexpr (IN_LDS)
    {
        // one preparation task: operands `cur` of round `it`
        auto do_task = [&] (const PrepIn& cur, int it)
        {
            {
                const int t = it * nhw + hwid;
                const bool act = t < ntask;
                const int tc = min(t, ntask - 1);
                const int blk = gemv_udiv(tc, mg_m), row = tc - blk * m;
                half4_t xv = cur.xv;
                float ssq_pub = 0.0f;
                if constexpr (MODE == G4_MODE_ACT)
                {
                    // a = fp16(silu(g) * u) of this (row, block): split-k reduce of the producer's gate / up slabs, output Hadamards, svh -- the
                    // arithmetic of glue_act_kernel (same device functions, slice-order sums)
                    const int blk_abs = (k0s >> 7) + blk;
                    // slice-order sums: lines 0..3 came with the task, the rest (splits deeper than 4) are fetched here
                    float4_t vg = { 0.f, 0.f, 0.f, 0.f }, vu = vg;
                    #pragma unroll
                    for (int i = 0; i < NSL; ++i) if (i < a_act_S)
                    {
                        vg.x += cur.ga[i].x; vg.y += cur.ga[i].y; vg.z += cur.ga[i].z; vg.w += cur.ga[i].w;
                        vu.x += cur.ua[i].x; vu.y += cur.ua[i].y; vu.z += cur.ua[i].z; vu.w += cur.ua[i].w;
                    }
                    if (a_act_S > NSL)
                    {
                        const float* pa = a_act_g + ((size_t) blk_abs * a_act_S * m + row) * 128;
                        const float* pb = a_act_u + ((size_t) blk_abs * a_act_S * m + row) * 128;
                        for (int sl = NSL; sl < a_act_S; sl += 4)
                        {
                            float4_t ta[4], tb[4];
                            #pragma unroll
                            for (int i = 0; i < 4; ++i)
                            {
                                const size_t so = (size_t) min(sl + i, a_act_S - 1) * m * 128;
                                ta[i] = ((const float4_t*) (pa + so))[l32]; tb[i] = ((const float4_t*) (pb + so))[l32];
                            }
                            #pragma unroll
                            for (int i = 0; i < 4; ++i) if (sl + i < a_act_S)
                            {
                                vg.x += ta[i].x; vg.y += ta[i].y; vg.z += ta[i].z; vg.w += ta[i].w;
                                vu.x += tb[i].x; vu.y += tb[i].y; vu.z += tb[i].z; vu.w += tb[i].w;
                            }
                        }
                    }
                    const half4_t svg = cur.svg, svu = cur.svu;
                    float g0, g1, g2, g3, u0, u1, u2, u3;
                    out_had(vg, l32, g0, g1, g2, g3);
                    out_had(vu, l32, u0, u1, u2, u3);
                    if (!TBL && a.act_rs.ss_new)
                    {
                        const float rsc = gemv_rescale(a.act_rs, row, l32, cur.ss, cur.ssn);        // r_new / r_prev of the row
                        g0 *= rsc; g1 *= rsc; g2 *= rsc; g3 *= rsc; u0 *= rsc; u1 *= rsc; u2 *= rsc; u3 *= rsc;
                    }
                    const half4_t gh = half4_t{ f2h(g0), f2h(g1), f2h(g2), f2h(g3) } * svg;
                    const half4_t uh = half4_t{ f2h(u0), f2h(u1), f2h(u2), f2h(u3) } * svu;
                    auto silu_mul = [] (half_t g, half_t u) -> half_t { float gf = (float) g; return f2h(gf / (1.0f + __expf(-gf)) * (float) u); };
                    xv = half4_t{ silu_mul(gh.x, uh.x), silu_mul(gh.y, uh.y), silu_mul(gh.z, uh.z), silu_mul(gh.w, uh.w) };
                }
                if constexpr (MODE == G4_MODE_QKVM)
                {
                    // x = q block blk_abs of this row, finished as exl3_glue_qkv_tab finishes it: slab lines summed in slice order from zero (slab_sum),
                    // then qkv_block_finish (output Hadamard, r_new / r_prev, fp16, x svh, RoPE) -- the same device function, the same bits
                    const int blk_abs = (k0s >> 7) + blk, qS = a.qkvm.S;
                    float4_t v = { 0.f, 0.f, 0.f, 0.f };
                    #pragma unroll
                    for (int i = 0; i < NQL; ++i) if (i < qS) { v.x += cur.ql[i].x; v.y += cur.ql[i].y; v.z += cur.ql[i].z; v.w += cur.ql[i].w; }
                    if (qS > NQL)
                    {
                        const float* pq = a.qkvm.sq + ((size_t) blk_abs * qS * m + row) * 128;
                        for (int sl = NQL; sl < qS; sl += 4)
                        {
                            float4_t t4[4];
                            #pragma unroll
                            for (int i = 0; i < 4; ++i) t4[i] = ((const float4_t*) (pq + (size_t) min(sl + i, qS - 1) * m * 128))[l32];
                            #pragma unroll
                            for (int i = 0; i < 4; ++i) if (sl + i < qS) { v.x += t4[i].x; v.y += t4[i].y; v.z += t4[i].z; v.w += t4[i].w; }
                        }
                    }
                    xv = qkv_block_finish(v, cur.svg, a.qkvm.rs, row, l32, cur.ss, cur.ssn, true, a.qkvm.rope_mode, a.qkvm.hd >> 3, cur.sn4, cur.cs4);
                    if (cbg == 0 && act && a.qkvm.q_out) ((half4_t*) (a.qkvm.q_out + (size_t) row * a_k + (size_t) blk_abs * 128))[l32] = xv;
                }
                if constexpr (MODE == G4_MODE_ATTM)
                {
                    // x = the attention output of query head `blk_abs` of this row: merge of the context splits' partial records -- the arithmetic of
                    // attn_merge_kernel<128> (exl3_attn_decode.hip) operation for operation, so the bits are those of the two-launch route: statistics by
                    // butterflies, the splits in four consecutive groups of ceil(nsplit / 4) summed sequentially and combined ((P0 + P1) + P2) + P3
                    // (its four helper half-waves), 1 / L, the inverse 32-point rotation, fp16.  nsplit <= 32 (host-checked): one chunk of statistics.
                    // head_dim 64 (attn_merge_kernel<64>): two heads per block, 16 lanes each -- the butterflies stop at 8, at most 16 splits
                    const AttmRec rc = attm_rec(row, (k0s >> 7) + blk);
                    const float* p = rc.p;
                    const bool has = at_lr < at_nsplit;
                    const float m_s = has ? cur.mst.x : -1.0e30f, l_s = has ? cur.mst.y : 0.0f;
                    float M = m_s;
                    #pragma unroll
                    for (int i = 1; i < 16; i <<= 1) M = fmaxf(M, xor_lane(M, i));
                    if (!at_hd64) M = fmaxf(M, xor_lane(M, 16));
                    const float e_s = m_s > -1.0e29f ? __expf(m_s - M) : 0.0f;
                    float L = l_s * e_s;
                    #pragma unroll
                    for (int i = 1; i < 16; i <<= 1) L += xor_lane(L, i);
                    if (!at_hd64) L += xor_lane(L, 16);
                    const int per = (at_nsplit + 3) >> 2, lbase = lane - at_lr;
                    float4_t P[4];
                    #pragma unroll
                    for (int hh = 0; hh < 4; ++hh) P[hh] = float4_t{ 0.f, 0.f, 0.f, 0.f };
                    for (int s0 = 0; s0 < at_nsplit; s0 += NMO)
                    {
                        float4_t ov[NMO];
                        #pragma unroll
                        for (int u = 0; u < NMO; ++u) ov[u] = s0 == 0 ? cur.mo[u] : *((const float4_t*) (p + (size_t) min(s0 + u, at_nsplit - 1) * 132 + rc.acc_off));
                        #pragma unroll
                        for (int u = 0; u < NMO; ++u)
                        {
                            const int sg = min(s0 + u, at_nsplit - 1);
                            const float ev = __shfl(e_s, lbase + sg, 64);
                            const int hh = sg >= 3 * per ? 3 : (sg >= 2 * per ? 2 : (sg >= per ? 1 : 0));
                            if (s0 + u < at_nsplit)
                            {
                                #pragma unroll
                                for (int q = 0; q < 4; ++q) if (q == hh) { P[q].x += ov[u].x * ev; P[q].y += ov[u].y * ev; P[q].z += ov[u].z * ev; P[q].w += ov[u].w * ev; }
                            }
                        }
                    }
                    float o0 = P[0].x, o1 = P[0].y, o2 = P[0].z, o3 = P[0].w;
                    #pragma unroll
                    for (int hh = 1; hh < 4; ++hh) { o0 += P[hh].x; o1 += P[hh].y; o2 += P[hh].z; o3 += P[hh].w; }
                    const float inv = L > 0.0f ? 1.0f / L : 0.0f;
                    float v0 = o0 * inv, v1 = o1 * inv, v2 = o2 * inv, v3 = o3 * inv;
                    kvg_had32(v0, v1, v2, v3, lane);
                    const float r32 = 0.17677669529663688110f;
                    xv = half4_t{ f2h(v0 * r32), f2h(v1 * r32), f2h(v2 * r32), f2h(v3 * r32) };
                }
                if constexpr (MODE == G4_MODE_ACTFX)
                {
                    // g, u of this (row, block) are complete in the accumulators (out-Hadamard and svh were applied per split-k partial, the sum is
                    // exact integer addition): row-scale correction, one rounding to fp16, silu * mul
                    auto fxf = [] (uint32_t lo, uint32_t hi) -> float { return fx_to_float(lo, hi); };      // NaN for a poisoned accumulator
                    float g0 = fxf(cur.f0.x, cur.f0.y), g1 = fxf(cur.f0.z, cur.f0.w), g2 = fxf(cur.f1.x, cur.f1.y), g3 = fxf(cur.f1.z, cur.f1.w);
                    float u0 = fxf(cur.f2.x, cur.f2.y), u1 = fxf(cur.f2.z, cur.f2.w), u2 = fxf(cur.f3.x, cur.f3.y), u3 = fxf(cur.f3.z, cur.f3.w);
                    if (a.act_rs.ss_new)
                    {
                        const float rsc = gemv_rescale(a.act_rs, row, l32, cur.ss, cur.ssn);
                        g0 *= rsc; g1 *= rsc; g2 *= rsc; g3 *= rsc; u0 *= rsc; u1 *= rsc; u2 *= rsc; u3 *= rsc;
                    }
                    auto silu_mul = [] (float g, float u) -> half_t { const float gf = (float) f2h(g); return f2h(gf / (1.0f + __expf(-gf)) * (float) f2h(u)); };
                    xv = half4_t{ silu_mul(g0, u0), silu_mul(g1, u1), silu_mul(g2, u2), silu_mul(g3, u3) };
                }
                if constexpr (MODE == G4_MODE_NORMFX)
                {
                    // the residual stream in 64-bit fixed point (value * 2^32, GEMV_OUT_ATOMIC launches add into it): x = fp16(hi + lo / 2^32)
                    auto fx = [] (uint32_t lo, uint32_t hi) -> half_t { return f2h(fx_to_float(lo, hi)); };             // NaN for a poisoned accumulator
                    xv = half4_t{ fx(cur.f0.x, cur.f0.y), fx(cur.f0.z, cur.f0.w), fx(cur.f1.x, cur.f1.y), fx(cur.f1.z, cur.f1.w) };
                    if (cbg == 0)
                    {
                        // block sums of squares of THIS residual for whoever finishes this launch's outputs (exl3_glue_qkv_rs / _act_rs) and for
                        // the next consumer's estimate: one workgroup per k-slice (column block 0 of matrix 0) covers every block
                        const float r0 = (float) xv.x, r1 = (float) xv.y, r2 = (float) xv.z, r3 = (float) xv.w;
                        float ssq = r0 * r0;
                        ssq = __builtin_fmaf(r1, r1, ssq); ssq = __builtin_fmaf(r2, r2, ssq); ssq = __builtin_fmaf(r3, r3, ssq);
                        #pragma unroll
                        for (int i = 1; i < 32; i <<= 1) ssq += xor_lane(ssq, i);
                        ssq_pub = ssq;                       // stored at the end of the task: a conditional store HERE makes the load count unknown to the
                                                             // compiler (vmcnt counts stores) and the wait for the row's sums below becomes vmcnt(0)
                    }
                }
                if constexpr (MODE == G4_MODE_NORM || MODE == G4_MODE_NORMFX)
                {
                    // x = fp16(resid * norm_w * rsqrt(mean(resid^2) + eps)): the row's mean square from the per-block sums a glue kernel left behind,
                    // same arithmetic and summation order as generation 2 / glue_norm_kernel / rms_norm (norm.cu:20-120)
                    const int nblk_k = a_k >> 7;
                    float s2 = l32 < nblk_k ? cur.ss : 0.0f;      // the first 32 blocks travel with the task (straight-line code: countable waits)
                    #pragma unroll
                    for (int i = 1; i < 32; i <<= 1) s2 += xor_lane(s2, i);
                    for (int b0 = 32; b0 < nblk_k; b0 += 32)       // hidden > 4096 only
                    {
                        float v = (b0 + l32 < nblk_k) ? a_ss_part[(size_t) row * nblk_k + b0 + l32] : 0.0f;
                        #pragma unroll
                        for (int i = 1; i < 32; i <<= 1) v += xor_lane(v, i);
                        s2 += v;
                    }
                    const float r = __frsqrt_rn(s2 / (float) a_k + a_eps);
                    xv = half4_t{ f2h((float) xv.x * (float) cur.wv.x * r), f2h((float) xv.y * (float) cur.wv.y * r),
                                  f2h((float) xv.z * (float) cur.wv.z * r), f2h((float) xv.w * (float) cur.wv.w * r) };
                }
                xv = xv * cur.sv;
                float h0 = (float) xv.x, h1 = (float) xv.y, h2 = (float) xv.z, h3 = (float) xv.w;
                had128_f32x4(h0, h1, h2, h3, l32);
                const half2_t o01 = { f2h(h0 * HAD_R_SCALE_128), f2h(h1 * HAD_R_SCALE_128) };
                const half2_t o23 = { f2h(h2 * HAD_R_SCALE_128), f2h(h3 * HAD_R_SCALE_128) };
                if constexpr (RAW)
                {
                    float ts = ((float) o01.x + (float) o01.y) + ((float) o23.x + (float) o23.y);
                    #pragma unroll
                    for (int i = 1; i < 32; i <<= 1) ts += xor_lane(ts, i);
                    if (act && l32 == 0) bsum[blk * m + row] = ts;
                }
                if (act)
                {
                    // elements 4*l32 .. +3 of the block: tile row l32 >> 2; offsets 4*(l32 & 3) .. +3 of the tile row = quads q0, q0 + 1, slot pair sp
                    const int tr = blk * 8 + (l32 >> 2);
                    const int q0 = 2 * (l32 & 1), sp = (l32 >> 1) & 1;
                    char* base = quads + ((size_t) (tr * 4 + q0) * m + row) * 8 + sp * 4;
                    *((half2_t*) base) = o01;
                    *((half2_t*) (base + (size_t) m * 8)) = o23;
                }
                if constexpr (MODE == G4_MODE_NORMFX) { if (cbg == 0 && act && l32 == 0) a.rs_ss_out[(size_t) row * (a_k >> 7) + (k0s >> 7) + blk] = ssq_pub; }
            }
        };
#ifndef G4_NO_PRIO
        // the workgroup's critical path (its preparation tasks, and below the half-waves that finish its outputs) ahead of the streaming waves that share
        // the SIMD: +1.5 % on the whole batch-1 step (round 4, tools/experiments/variant_queue.sh, same box: 646.4 against 637.6 / 635.4 tok/s)
        if (prep_wave) __builtin_amdgcn_s_setprio(3);
#endif
        if (prep_wave)
        {
            if (ntask <= nhw) do_task(nx, 0);       // the usual case, one round: a straight line in which the compiler counts the outstanding loads and the
                                                    // task waits for ITS operands only (vmcnt(NR)), not for the weight rows requested behind them -- in
                                                    // the pipelined loop below the count differs per path and every wait is vmcnt(0)
            else
            {
                for (int it = 0; it * nhw + 2 * wave < ntask; ++it)
                {
                    const PrepIn cur = nx;
                    if ((it + 1) * nhw + 2 * wave < ntask) nx = fetch(it + 1);
                    do_task(cur, it);
                }
            }
        }
        if constexpr (MODE == G4_MODE_QKVM)
        {
            // side job of the column-block-0 workgroups (one per k-slice): the new token's K / V rows -- (row, K | V, 128-value block) tasks t = slice + S * j,
            // j over this workgroup's half-waves starting with those that had NO preparation task (they would idle at the barrier below): at batch 1 a
            // slice has 4 preparation tasks and 2 of these.  Same arithmetic as glue_qkv_kernel (qkv_block_finish + kv_quant_regs), 4-bit cache.
            if (cbg == 0)
            {
                const int kvb = a.qkvm.kvb, nkv = m * 2 * kvb;
                const int nj = s < nkv ? (nkv - s + a_S - 1) / a_S : 0;                // this slice's tasks t = s + S * j, j < nj (workgroup-uniform)
                const int busy = min(ntask, nhw);                                      // half-waves that own a preparation task
                auto rank = [&] (int hw) { return hw >= busy ? hw - busy : hw + (nhw - busy); };      // idle half-waves first
                const int jw = rank(hwid), jmin = min(rank(2 * wave), rank(2 * wave + 1));
                for (int jr = 0; jr * nhw < nj; ++jr)
                {
                    if (jmin + nhw * jr >= nj) continue;                                // wave-uniform: neither half-wave of this wave has a task this round
                    const int j = jw + nhw * jr;
                    const bool actk = j < nj;
                    const int tc = s + a_S * min(j, nj - 1);
                    const int row = tc / (2 * kvb), rem = tc - row * 2 * kvb, isv = rem >= kvb ? 1 : 0, hb = rem - isv * kvb;
                    const SlabRef sr = { isv ? a.qkvm.sv : a.qkvm.sk, a.qkvm.S };
                    const half4_t sc = ((const half4_t*) ((isv ? a.qkvm.svh_v : a.qkvm.svh_k) + hb * 128))[l32];
                    float rs_p = 0.0f, rs_n = 0.0f;
                    if (a.qkvm.rs.ss_new && l32 < (a.qkvm.rs.k >> 7)) { rs_p = a.qkvm.rs.ss_prev[(size_t) row * (a.qkvm.rs.k >> 7) + l32]; rs_n = a.qkvm.rs.ss_new[(size_t) row * (a.qkvm.rs.k >> 7) + l32]; }
                    const int ph = a.qkvm.hd >> 3;
                    float4_t sn4 = { 0.f, 0.f, 0.f, 0.f }, cs4 = sn4;
                    if (a.qkvm.rope_mode == 2)
                    {
                        const int f = 4 * (l32 & (ph - 1));
                        sn4 = *((const float4_t*) (a.qkvm.rope_sin + row * 64 + f)); cs4 = *((const float4_t*) (a.qkvm.rope_cos + row * 64 + f));
                    }
                    else
                    {
                        const int f = 2 * (l32 & ((a.qkvm.hd >> 2) - 1));
                        sn4.x = a.qkvm.rope_sin[row * 64 + f]; sn4.y = a.qkvm.rope_sin[row * 64 + f + 1]; cs4.x = a.qkvm.rope_cos[row * 64 + f]; cs4.y = a.qkvm.rope_cos[row * 64 + f + 1];
                    }
                    const int64_t token_pos = a.qkvm.slots[row];
                    const float4_t ysum = slab_sum(sr, hb, row, m, l32);
                    const half4_t y = qkv_block_finish(ysum, sc, a.qkvm.rs, row, l32, rs_p, rs_n, !isv, a.qkvm.rope_mode, ph, sn4, cs4);
                    const int64_t gb = token_pos * (kvb * 4) + hb * 4 + (l32 >> 3);
                    uint32_t* cw = isv ? a.qkvm.v_cache : a.qkvm.k_cache; half_t* csc = isv ? a.qkvm.v_scales : a.qkvm.k_scales;
                    kv_quant_regs<4>((float) y.x, (float) y.y, (float) y.z, (float) y.w, cw + gb * 4, csc + gb, actk, lane);
                }
            }
        }
#ifndef G4_NO_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        __syncthreads();
#ifdef G4_ABL_ONE_ROW
        if (nun > 0) load_lane_words<K>(ring[1], strip + (size_t) (2 * ubase + 1) * row_stride);
#endif
        agn = load_group(2 * ubase);
    }

    float4_t acc_c = { 0.f, 0.f, 0.f, 0.f }, acc_d = acc_c;
#ifdef G4_RAMP
    issue_ring_second();
#endif
    G4_T(2);

    // ---- streaming: groups of 2 units (4 tile rows share one activation register pair), then an odd last unit
    const int ngrp = nun >> 1;
    int unit = ubase;
    for (int j = 0; j < ngrp; ++j)                              // plain counted loop (an early exit makes the compiler drain vmcnt every trip)
    {
        set_group(agn);
        agn = load_group(2 * min(unit + 2, last_unit));         // next group (clamped: a harmless reload at the end)
        // a unit refills its slots with the rows of the unit PFU ahead (clamped: a harmless reload at the end)
#ifdef G4_ABL_HOT_ALL
        g4_unit<K, CB, VAR, 0, NR>(ring, strip_hot, row_stride, lane, agc0, agc1, acc_c, acc_d);
        g4_unit<K, CB, VAR, 1, NR>(ring, strip_hot, row_stride, lane, agc0, agc1, acc_c, acc_d);
#else
        g4_unit<K, CB, VAR, 0, NR>(ring, strip + (size_t) (2 * min(unit + PFU, last_unit)) * row_stride, row_stride, lane, agc0, agc1, acc_c, acc_d);
        g4_unit<K, CB, VAR, 1, NR>(ring, strip + (size_t) (2 * min(unit + 1 + PFU, last_unit)) * row_stride, row_stride, lane, agc0, agc1, acc_c, acc_d);
#endif
        unit += 2;
    }
    if (nun & 1)
    {
        set_group(agn);
        g4_unit<K, CB, VAR, 0, NR>(ring, strip + (size_t) (2 * last_unit) * row_stride, row_stride, lane, agc0, agc1, acc_c, acc_d);
    }

    G4_T(3);
    {
        // the scalars of the output side (matrix record, row offset) are fetched HERE, under the partial-sum exchange: loaded where they are used they
        // were two dependent scalar round trips between the last barrier and the svh load / the atomics of every workgroup
        const void* e0 = a.mat[mi].svh; const void* e1 = a.mat[mi].bias; const void* e2 = a.mat[mi].C; const int64_t e3 = a.c_row_offset;
        asm volatile("" :: "s"(e0), "s"(e1), "s"(e2), "s"(e3));
    }
    // ---- partial sums -> LDS (the area of the activation quads: every wave is past its last quad read after this barrier)
    if constexpr (IN_LDS) __syncthreads();
    {
        float* pw = part + (size_t) wave * m * 128;
        const int col = 16 * T + c;
        #pragma unroll
        for (int i = 0; i < 4; ++i) if (i < m) { pw[i * 128 + col] = acc_c[i]; pw[i * 128 + col + 8] = acc_d[i]; }
    }
    __syncthreads();
    G4_T(4);

    // ---- half-wave h takes rows h, h + nhw, ...: sum of the waves' partials, the mul1 FAST affine map, then the slab line or the final output row
#ifndef G4_NO_PRIO
    if (hwid < m) __builtin_amdgcn_s_setprio(3);
#endif
    const int l = l32;
    for (int row = hwid; row < m; row += nhw)
    {
        float4_t v = ((const float4_t*) (part + row * 128))[l];
        for (int w = 1; w < nwv; ++w)
        {
            const float4_t t = ((const float4_t*) (part + ((size_t) w * m + row) * 128))[l];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if constexpr (RAW)
        {
            float xs = 0.0f;
            if constexpr (IN_LDS) { for (int b0 = 0; b0 < nb; b0 += 32) if (b0 + l < nb) xs += bsum[(b0 + l) * m + row]; }
            else if (row == hwid)
            {
                xs = l < nb ? xs_pre : 0.0f;                   // requested at kernel entry (blocks 0..31 of the slice)
                const float* xsr = a.mat[mi].xsum + (size_t) row * (a_k >> 7) + (k0s >> 7);
                for (int b0 = 32; b0 < nb; b0 += 32) if (b0 + l < nb) xs += xsr[b0 + l];
            }
            else
            {
                // (one-wave workgroups with more than two rows only) the further rows' block sums are fetched here
                const float* xsr = a.mat[mi].xsum + (size_t) row * (a_k >> 7) + (k0s >> 7);
                for (int b0 = 0; b0 < nb; b0 += 32) if (b0 + l < nb) xs += xsr[b0 + l];
            }
            #pragma unroll
            for (int i = 1; i < 32; i <<= 1) xs += xor_lane(xs, i);
            const float kinv = (float) u16_as_half(0x1eeeu), kbias = (float) u16_as_half(0xc931u);
            const float b = kbias * xs;
            v.x = v.x * kinv + b; v.y = v.y * kinv + b; v.z = v.z * kinv + b; v.w = v.w * kinv + b;
        }
        if ((a_S > 1 && !(a_flags & GEMV_OUT_ATOMIC)) || (a_flags & GEMV_OUT_DEFERRED))
        {
            float* slab = a.workspace + ws_off + ((size_t) cbl * a_S + s) * (size_t) m * 128;
            ((float4_t*) (slab + row * 128))[l] = v;
            continue;
        }
        float h0 = v.x, h1 = v.y, h2 = v.z, h3 = v.w;
        had128_f32x4(h0, h1, h2, h3, l);
        h0 *= HAD_R_SCALE_128; h1 *= HAD_R_SCALE_128; h2 *= HAD_R_SCALE_128; h3 *= HAD_R_SCALE_128;
        const half_t* svh_m = a.mat[mi].svh; void* C_m = a.mat[mi].C; size_t c_row = (size_t) a.c_row_offset + row;
        const half_t* bias = (!TBL && a.mat[mi].bias) ? a.mat[mi].bias + cbl * 128 : nullptr;
        if constexpr (TBL)
        {
            svh_m = (const half_t*) a.tbl.svh[tmat];
            h0 *= tweight; h1 *= tweight; h2 *= tweight; h3 *= tweight;       // reference: scale *= weight, then one multiply (exl3_gemm_kernel.cuh:216-217)
            if (a_flags & GEMV_OUT_ATOMIC) { C_m = a.tbl.C; c_row = (size_t) (slot / a.tbl.slots_per_token) * m + row; }     // every slot of a token adds into that token's rows
            else C_m = a.c_fp32 ? (void*) ((float*) a.tbl.C + (size_t) slot * a.tbl.c_slot_stride) : (void*) ((half_t*) a.tbl.C + (size_t) slot * a.tbl.c_slot_stride);
        }
        const half4_t sc = ((const half4_t*) (svh_m + cbl * 128))[l];
        if (a_flags & GEMV_OUT_ATOMIC)
        {
            // the slice's share of the output rows (out-Hadamard and svh applied to the partial: both linear), added into the fixed-point
            // accumulator; fp32 arithmetic of glue_resid up to the sum, which here is exact integer addition in any order
            float o[4] = { h0 * (float) sc.x, h1 * (float) sc.y, h2 * (float) sc.z, h3 * (float) sc.w };
            if (bias && s == 0) { const half4_t bv = ((const half4_t*) bias)[l]; o[0] += (float) bv.x; o[1] += (float) bv.y; o[2] += (float) bv.z; o[3] += (float) bv.w; }
            unsigned long long* acc = (unsigned long long*) C_m + c_row * n + cbl * 128 + 4 * l;
            // (a NaN / Inf / out-of-range share poisons the accumulator instead of adding finite garbage: fx_atomic_add, exl3_gemv_args.h)
            #pragma unroll
            for (int i = 0; i < 4; ++i) fx_atomic_add(acc + i, o[i]);
            continue;
        }
        const size_t off = c_row * n + cbl * 128 + 4 * l;
        if (a.c_fp32)
        {
            float4_t o = { h0 * (float) sc.x, h1 * (float) sc.y, h2 * (float) sc.z, h3 * (float) sc.w };
            if (bias) { const half4_t bv = ((const half4_t*) bias)[l]; o.x += (float) bv.x; o.y += (float) bv.y; o.z += (float) bv.z; o.w += (float) bv.w; }
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
    if (a.fx_zero)
    {
        // side job (fx pipeline): clear this workgroup's share of a buffer a LATER launch accumulates into.  Last thing the workgroup does: between the
        // first loads and the preparation tasks its conditional stores made the outstanding-load count unknown to the compiler (vmcnt(0) in front of the tasks)
        const int nwg = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
        const int per = (a.fx_zero_n16 + nwg - 1) / nwg;
        const int c0 = wg * per, c1 = min(c0 + per, a.fx_zero_n16);
        for (int cidx = c0 + tid; cidx < c1; cidx += 64 * nwv) ((uint4_t*) a.fx_zero)[cidx] = uint4_t{ 0u, 0u, 0u, 0u };
    }

#ifdef G4_TIMING
    if (tid == 0)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tstamp[5] = __builtin_amdgcn_s_memrealtime();
        uint64_t* dbg = (uint64_t*) a.ws_debug + (size_t) (blockIdx.y * gridDim.x + blockIdx.x) * 8;
        for (int i = 0; i < 6; ++i) dbg[i] = tstamp[i];
        uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        dbg[6] = xcc; dbg[7] = __builtin_amdgcn_s_memtime() - cyc0;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// launch (called from exl3_gemv.hip's dispatcher).  One translation unit per K: built with -DG2_K=1..8.
// ------------------------------------------------------------------------------------------------
#ifndef G2_K
#error "compile with -DG2_K=<bits per weight>"
#endif

template <int CB>
static void g4_launch_cb(int var, int mode, int nwv, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args)
{
    #define L(V, M) exl3_gemv4_kernel<G2_K, CB, V, M><<<grid, dim3(64 * nwv), lds, st>>>(args)
    #define LV(M) { if (var == 0) L(0, M); else L(1, M); }
    switch (mode)
    {
        case G4_MODE_ROT:  LV(G4_MODE_ROT)  break;
        case G4_MODE_RAWX: LV(G4_MODE_RAWX) break;
        case G4_MODE_NORM: LV(G4_MODE_NORM) break;
        case G4_MODE_NORMFX: LV(G4_MODE_NORMFX) break;
        case G4_MODE_ACTFX: LV(G4_MODE_ACTFX) break;
        case G4_MODE_TRAWX: LV(G4_MODE_TRAWX) break;
        case G4_MODE_TACT: LV(G4_MODE_TACT) break;
        case G4_MODE_ATTM: LV(G4_MODE_ATTM) break;
        case G4_MODE_QKVM: LV(G4_MODE_QKVM) break;
        default:           LV(G4_MODE_ACT)  break;
    }
    #undef LV
    #undef L
}

#define G4_CAT_(a, b) a##b
#define G4_CAT(a, b) G4_CAT_(a, b)

void G4_CAT(exl3_gemv4_launch_k, G2_K)(int cb, int var, int mode, int nwv, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args)
{
    if (cb == 0) g4_launch_cb<0>(var, mode, nwv, grid, lds, st, args);
    else if (cb == 1) g4_launch_cb<1>(var, mode, nwv, grid, lds, st, args);
    else g4_launch_cb<2>(var, mode, nwv, grid, lds, st, args);
}

#if G2_K == 4
size_t exl3_gemv4_lds_bytes(int mode, int nwv, int m, int blocks_per_slice)
{
    const size_t quad = mode == G4_MODE_ROT ? 0 : (size_t) (blocks_per_slice * 8 + 4) * 4 * m * 8;
    const size_t part = (size_t) nwv * m * 512;
    return (((quad > part ? quad : part) + 15) & ~(size_t) 15) + (size_t) blocks_per_slice * m * 4 + 16;
}
#endif
