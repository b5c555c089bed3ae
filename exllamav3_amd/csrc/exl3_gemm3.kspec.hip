// EXL3 quantized small-m GEMM for gfx950, kernel generation 3 ("decode -> 16x16x32 MFMA straight from the decoding lane's registers"), 5..64 rows per pass.
//
//   C = ((A * suh) H) @ dequant(B) H * svh (+ bias)        reference: quant/exl3_gemm_kernel.cuh:8-80, quant/exl3_gemm_inner.cuh
//                                                          (semantics only; the reference streams 16 rows per pass)
//
// Why a third kernel: generation 2 (exl3_gemv2.kspec.hip) multiplies straight out of the decoding lane's registers with
// v_mfma_f32_4x4x4_16B_f16, which does a quarter of the matrix pipe's work per cycle.  That is free at m <= 4 (the decode VALU work
// dominates) but at 16 rows the kernel is MFMA-bound: 1024 MFMA cycles against ~660 decode cycles per tile row.  Here the fp16 weights
// feed v_mfma_f32_16x16x32_f16 (B operand; see "No transpose" below): 64 MFMA cycles per 16 rows per decode step, so the weight pass is
// decode-bound again and a 32-row pass costs about what a 4-row pass does.
//
// Work split: a workgroup = 4 waves = one 128-column block x one k-slice, as in generation 2, but the waves split the COLUMNS (32 each),
// not k: every wave walks the whole slice.  Nothing is reduced across waves -- no partial sums through LDS, no barrier after the
// streaming loop, 8 accumulator VGPRs per 16 rows -- and all waves have exactly the same amount of work.
//
// Round 4: a workgroup may hold 4, 8 or 16 waves (NWT) = 1, 2 or 4 ADJACENT column blocks of one matrix over the same k-slice.  The waves are as
// independent as before; what they share is the activation tile in LDS.  With one column block per workgroup every one of the 224 gate|up column
// blocks copied the same 16 x k activations: 30 MB of L2 -> LDS traffic per launch beside 59 MB of weights, all of it requested in the launch's first
// microsecond in front of the workgroups' weight rows -- tools/gemv_timeline.py (BSZ=16) showed 2..12 us from entry to "activations staged" once four
// workgroups shared a CU, so higher occupancy bought nothing (gate|up S = 2 / 4 / 8: the same 24 us).  Four column blocks per workgroup quarter
// that traffic and the LDS footprint (a whole k = 4096 slice of 16 rows fits: no chunk loop for the lm_head), one workgroup fills a CU.
//
// Lane geometry of a decode step (4 tile rows x the wave's 2 tiles): lane = 16 g + 8 t2 + c owns, as in exl3_lane_decode.cuh, the K
// words [cK, cK + K) of tile (2 wave + t2) of tile row 4 step + g = columns c, c + 8 of that tile, 16 k each: one 16-byte load per lane
// (K = 4), 256 contiguous bytes per 16-lane group.
//
// No transpose (round 3).  The decoding lane's registers ARE a B operand of v_mfma_f32_16x16x32_f16: operand lane l supplies column l % 16 and
// contraction slots 8 (l / 16) .. + 7, and lane 16 g + 8 t2 + c holds, for column c (and c + 8) of tile t2, the 16 k of tile row g as two natural-order
// runs of 8 (low words: k 0..7, high words: k 8..15).  Reading "column" as the virtual column 8 t2 + c and "contraction group" as tile row g, the four
// instructions of a step are (k 0..7 | k 8..15 of the four tile rows) x (columns c | c + 8 of the two tiles), their A operand is the 16 bytes
// x[row][16 (4 step + g) + 8 h ..] of the row-major activations, and their accumulators hold the outputs of the lane's own columns.  (Rounds 1-2 wrote
// the decoded step to a wave-private LDS buffer and read it back in the natural layout: 4 ds_write_b128 + 4 ds_read_b128 + two wave-level fences
// per step and 18 KB of LDS per workgroup for nothing -- the contraction index of a matrix product can be enumerated in any order both operands agree on.)
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include "exl3_gemv_args.h"
#include "exl3_lane_decode.cuh"

#include <type_traits>

template <int I, int N, typename F>
__device__ __forceinline__ void g3_static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); g3_static_for<I + 1, N>(f); }
}

#ifndef G3_XPAD
#define G3_XPAD 16
#endif
#ifndef G3_NR
#define G3_NR 4            // weight-ring depth in decode steps (A/B builds: 8)
#endif

constexpr int g3_waves_per_eu(int K, int MT, bool ROT, bool RAW)
{
    (void) RAW;                                      // the raw variant is instantiated for the rotated-input prologue only, whose budgets cover its row-sum accumulators
    if (MT == 4 && K == 7 && !ROT) return 2;         // (round 4, likewise: 7-word ring slots)
    if (MT == 2 && K == 3 && !ROT) return 4;         // (round 4: the step-granular ring needs 2 more registers here than the 96 of five waves per SIMD)
    return MT == 4 ? (ROT ? (K >= 5 ? 2 : 3) : ((K == 4 || K <= 2) ? 4 : 3)) : ((K >= 7 || (ROT && K >= 5)) ? 3 : ((ROT || K >= 5) ? 4 : 5));
}
// waves per workgroup an instantiation can run with: 16 waves on 4 SIMDs need the 4-waves-per-SIMD register budget (128 VGPRs), 8 waves the 2-waves one
constexpr bool g3_nwt_ok(int K, int MT, bool ROT, bool RAW, int NWT) { return NWT == 4 || (MT == 1 && g3_waves_per_eu(K, MT, ROT, RAW) >= NWT / 4); }

template <int K, int CB, int MT, bool ROT, bool RAWV, int NWT>
// RAWV (mul1 codebook, rotated input = the fused decode pipeline, the library's default GEMV variant): the weights enter the matrix instructions as the packed byte sums 1024 + s the
// decode produces (exact fp16 integers) instead of fp16(kinv * (1024 + s) + kbias); the affine map is applied once per output,
// out = kinv * acc + kbias * sum_k x[row][k], with the row sums taken from the producer's per-block sums (mat[].xsum; launches without them take the
// fp16-weight variant).  Removes a v_pk_fma_f16 per weight pair, 14 % of the streaming loop's VALU instructions; the same arithmetic as generation
// 2's / 4's default variant.
// ROT: the input is already rotated (fused decode pipeline) -- a separate instantiation so that neither prologue's registers burden the other.
// five 4-wave workgroups per CU (LDS: 5 x 31 KB) need <= 96 VGPRs; the 64-row passes, the ROT prologue (16 registers of activation copy in
// flight next to the weight ring) and the wide rings of K >= 5 take the next register budgets instead of spilling (any scratch use slows
// every launch); the table below is what hipcc 7.2 needs for zero scratch in every (K, codebook, MT, ROT) instantiation
__global__ __launch_bounds__(64 * NWT) __attribute__((amdgpu_waves_per_eu(g3_waves_per_eu(K, MT, ROT, RAWV && CB == EXL3_CB_MUL1))))
void exl3_gemm3_kernel(const GemvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 8 * K;

#ifdef G2_TIMING
    // diagnostics build: the stamps of exl3_gemv2.kspec.hip (tools/gemv_timeline.py reads both)
    uint64_t tstamp[6];
    tstamp[0] = __builtin_amdgcn_s_memrealtime();
    const uint64_t cyc0 = __builtin_amdgcn_s_memtime();          // shader-clock counter: cycles / realtime = the clock the launch ran at
    #define G3_T(i) tstamp[i] = __builtin_amdgcn_s_memrealtime()
#else
    #define G3_T(i)
#endif
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // one batch of scalar loads for the whole prologue, matrix records prefetched, (k-slice, column block) from a 2-D grid, multiply-high
    // instead of runtime divisions: see the prologue of exl3_gemv2.kspec.hip
    const int m = a.m, a_S = a.S, a_k = a.k, a_kslice = a.kslice, a_chb = a.chunk_blocks, a_nm = a.num_mats;
    const int a_cbf[GEMV_MAX_MATS] = { 0, a.cbf[0], a.cbf[1], a.cbf[2] };
    const uint32_t mg_m = a.magic_m;
    const half_t* const a_A = a.A;
    {
        const void* t0 = a.mat[0].B; const void* t1 = a.mat[1].B; const void* t2 = a.mat[2].B; const void* t3 = a.mat[3].B; const void* t4 = a.mat[3].xsum;
        asm volatile("" :: "s"(t0), "s"(t1), "s"(t2), "s"(t3), "s"(t4));
    }
    const int s = blockIdx.x;
    const int cbg = blockIdx.y;
    int mi = 0;
    #pragma unroll
    for (int i = 1; i < GEMV_MAX_MATS; ++i) if (i < a_nm && cbg >= a_cbf[i]) mi = i;
    const uint32_t* __restrict__ Bm = a.mat[mi].B;
    const half_t* __restrict__ suh = a.mat[mi].suh;
    const int n = a.mat[mi].n, ws_off = a.mat[mi].ws_offset;
    // the wave's column block: group (cbg - first group of the matrix) x CPW + wave / 4; wv = the wave's 32-column quarter of it.  A matrix whose column
    // blocks do not fill its last group leaves waves without one: they stage and synchronise with the others, stream a clamped (valid) block and store nothing
    constexpr int CPW = NWT / 4;
    const int cb_want = (cbg - a.mat[mi].cb_first) * CPW + (wave >> 2), wv = wave & 3;
    const bool has_cb = cb_want < (n >> 7);
    const int cbl = has_cb ? cb_want : (n >> 7) - 1;
    {
        // everything the first vector loads need, in the SECOND batch of scalar loads (the matrix record depends on mi): left alone, the compiler sank
        // mat[mi].B and chunk_blocks behind the first branch -- a third scalar round trip between entry and the first weight request (ISA, round 3)
        const void* p0 = Bm; const void* p1 = a.mat[mi].xh; const void* p2 = a.mat[mi].xsum; const void* p3 = suh;
        asm volatile("" :: "s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(a_chb), "s"(n), "s"(ws_off));
    }
    const int tiles_n = n >> 4;
    const int k0s = s * a_kslice;
    const int k1s = min(k0s + a_kslice, a_k);
    const int nb = (k1s - k0s) >> 7;                 // Hadamard blocks in the slice; every wave streams all of them
    const int chb = a_chb;

    // LDS: the activations of one chunk, row-major fp16; the S == 1 epilogue reuses it for [m][128] fp32
    const int ldx = chb * 128 + G3_XPAD;
    half_t* xa = (half_t*) smem;

    const int l32 = lane & 31;
    constexpr bool in_rotated = ROT;
    const half_t* __restrict__ x_src = in_rotated ? a.mat[mi].xh : a_A;
    const int hwid = tid >> 5, nhw = NWT * 2;

    // decode geometry
    const int g = lane >> 4, t2 = (lane >> 3) & 1, c = lane & 7;
    const size_t row_stride = (size_t) tiles_n * NW;
    const uint32_t* __restrict__ strip = Bm + ((size_t) (k0s >> 4) + g) * row_stride + (size_t) (cbl * 8 + 2 * wv + t2) * NW + (size_t) c * K;

    constexpr bool RAW = RAWV && CB == EXL3_CB_MUL1;
    float4_t acc[MT][2];
    #pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i][0] = float4_t{ 0.f, 0.f, 0.f, 0.f }; acc[i][1] = acc[i][0]; }
    // RAW: sum_k x[row][k] over the slice for the rows of acc[i][*], from the producer's per-block sums (mat[mi].xsum [m][k/128], the launcher
    // takes this variant only when they exist) -- requested here, needed in the epilogue.  (Rounds 1-2 accumulated them with one extra matrix
    // instruction per 32 k against a ones operand: 2 of every 6 matrix instructions of the loop.)
    float4_t rsum[RAW ? MT : 1];
    #pragma unroll
    for (int i = 0; i < (RAW ? MT : 1); ++i) rsum[i] = float4_t{ 0.f, 0.f, 0.f, 0.f };
    constexpr bool RSUM_EARLY = MT == 1;             // the 32- / 64-row passes fetch them in the epilogue instead (their register budgets are full)
    if constexpr (RAW && RSUM_EARLY)
    {
        // the 16 lanes that share these rows (one DPP row) take different blocks each: one round of 4 loads per lane covers 16 blocks, nothing waits
        // on them before the epilogue, where a row sum over the 16 lanes finishes the job.  (Accumulating block after block here stalled every
        // workgroup for a memory round trip per block BEFORE its weight loads were issued: tools/gemv_timeline.py, 1.7 us from entry to "loads issued".)
        // (plain loads at a clamped block index, no arithmetic on them here -- an add would make the compiler wait for the value on the spot; lanes
        // past the slice's last block are zeroed in the epilogue, slices beyond 16 blocks fetch the rest there)
        const float* xsr = a.mat[mi].xsum + (k0s >> 7);
        const int nbk = a_k >> 7, kg0 = lane >> 4, blk0 = min(lane & 15, nb - 1);
        #pragma unroll
        for (int i = 0; i < MT; ++i)
            #pragma unroll
            for (int r = 0; r < 4; ++r) rsum[i][r] = xsr[(size_t) min(16 * i + 4 * kg0 + r, m - 1) * nbk + blk0];
    }

    // ---- activation fetch helpers
    struct PrepIn { half4_t xv, sv; };
    auto fetch = [&] (int c0, int cnt, int it) -> PrepIn
    {
        PrepIn r;
        const int t = min(it * nhw + hwid, cnt * m - 1);
        const int tq = gemv_udiv(t, mg_m);
        const int blk = c0 + tq, row = t - tq * m;
        const size_t kofs = (size_t) k0s + 128 * blk;
        r.xv = ((const half4_t*) (x_src + (size_t) row * a_k + kofs))[l32];
        r.sv = ((const half4_t*) (suh + kofs))[l32];
        return r;
    };
    // already rotated input (glue_rotate / glue_act): a straight copy in 16-byte pieces.  A row of the chunk is <= 2^cp_sl pieces; thread =
    // (row tid >> cp_sl of the instruction's 256 >> cp_sl rows, piece tid & (2^cp_sl - 1)); four instructions in flight per thread, so one batch
    // covers 16 rows at up to 4 blocks per chunk (the decode shapes), 32 rows at 2, 64 rows at 1
    const int cp_sl = chb <= 1 ? 4 : (chb <= 2 ? 5 : (chb <= 4 ? 6 : (chb <= 8 ? 7 : (chb <= 16 ? 8 : 9))));       // (host: 16 chb <= 64 NWT)
    const int cp_piece = tid & ((1 << cp_sl) - 1), cp_rsub = tid >> cp_sl, cp_rows = (64 * NWT) >> cp_sl;
    auto copy_load = [&] (int c0, int cnt, int base, uint4_t (&v)[4])
    {
        const half_t* src0 = x_src + (size_t) k0s + 128 * c0 + 8 * min(cp_piece, cnt * 16 - 1);
        #pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *((const uint4_t*) (src0 + (size_t) min(base + cp_rows * j + cp_rsub, m - 1) * a_k));
    };
    auto copy_store = [&] (int cnt, int base, const uint4_t (&v)[4])
    {
        #pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const int row = base + cp_rows * j + cp_rsub;
            if (row < m && cp_piece < cnt * 16) *((uint4_t*) (xa + (size_t) row * ldx + 8 * cp_piece)) = v[j];
        }
    };
    // activation operands first, weight rows second: loads return in issue order per wave (see exl3_gemv2.kspec.hip)
    PrepIn nx = { half4_t{ 0, 0, 0, 0 }, half4_t{ 0, 0, 0, 0 } };
    uint4_t cv[4];
    if constexpr (in_rotated) copy_load(0, min(chb, nb), 0, cv); else nx = fetch(0, min(chb, nb), 0);

    // weight ring: G3_NR slots of ONE decode step (4 tile rows) each, consumed round-robin by a loop that is unrolled G3_NR steps per trip, so every slot
    // is a fixed register set and every wait is countable: in front of step j exactly G3_NR - 1 younger ring loads are outstanding (round 4: the previous
    // two-block ring with its slot swap for odd block counts and the next chunk's activation rows requested across the loop made the compiler wait with
    // vmcnt(0) at the loop head, i.e. for the refill issued a step earlier -- one full memory round trip exposed per trip: tools/isa_blocks.py, the reason
    // the 16-row pass ran at 45 % of its decode VALU bound)
    constexpr int NR = G3_NR;
    const int nsteps = 2 * nb;
    LaneWords<K> ring[NR];
#ifdef G3_ABL_HOT
    auto step_ptr = [&] (int st) -> const uint32_t* { return strip + (size_t) (4 * (min(st, nsteps - 1) & 3)) * row_stride; };      // ablation: cache-hot rows
#else
    auto step_ptr = [&] (int st) -> const uint32_t* { return strip + (size_t) (4 * min(st, nsteps - 1)) * row_stride; };
#endif
    #pragma unroll
    for (int u = 0; u < NR; ++u) load_lane_words<K>(ring[u], step_ptr(u));
    G3_T(1);

    // MFMA operand geometry
    const int mj = lane & 15, kg = lane >> 4;
    const half_t* arow[MT];                              // A operand of (row 16 i + mj, tile row kg of the step): + 16-byte half h
    #pragma unroll
    for (int i = 0; i < MT; ++i) arow[i] = xa + (size_t) min(16 * i + mj, m - 1) * ldx + 16 * kg;

    // one decode step (4 tile rows x the wave's 2 tiles) from ring slot `slot`; kloc = chunk-local k of the step; the slot is refilled with step `refill`
    auto do_step = [&] (LaneWords<K>& slot, int kloc, int refill)
    {
        uint32_t Wx[K + 1];
        #pragma unroll
        for (int i = 0; i < K; ++i) Wx[i + 1] = slot.w[i];
        // carry-in = last word of the previous lane of the 8-lane tile group (c = 0 wraps to c = 7): two DPP row rotates + a select, register
        // file only (generation 4's form; this was a ds_bpermute, i.e. an LDS-crossbar round trip in front of every decode step)
        {
            const uint32_t wl = slot.w[K - 1];
            const uint32_t r1 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x121, 0xf, 0xf, true);
            const uint32_t r9 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x129, 0xf, 0xf, true);
            Wx[0] = (lane & 7) ? r1 : r9;
        }
#ifdef G3_REFILL_EARLY
        load_lane_words<K>(slot, step_ptr(refill));     // A/B: refill right after the words were copied out (the compiler then copies ring registers at the loop head and waits with vmcnt(0))
#endif

        // A operands of the step: 16-row passes request them before the decode and consume them after it (8 registers); the 32- / 64-row passes read
        // them in front of each half's matrix instructions (their register budgets are full: any spill costs 1.5..3 us per launch)
        constexpr bool AF_EARLY = MT == 1;
        half8_t af[2][MT];
#ifdef G3_ABL_NOLDS
        // ablation: no A-operand reads (garbage activations from registers)
        #pragma unroll
        for (int h = 0; h < 2; ++h)
            #pragma unroll
            for (int i = 0; i < MT; ++i) { union { uint32_t w[4]; half8_t h8; } u_; u_.w[0] = Wx[1]; u_.w[1] = kloc; u_.w[2] = lane; u_.w[3] = h; af[h][i] = u_.h8; }
#else
        if constexpr (AF_EARLY)
        {
            #pragma unroll
            for (int h = 0; h < 2; ++h)
                #pragma unroll
                for (int i = 0; i < MT; ++i) af[h][i] = *((const half8_t*) (arow[i] + kloc + 8 * h));
        }
#endif

        // exact fp16 weights: quad q of column c / c + 8 = rows {2q, 2q+1 | 2q+8, 2q+9}: low words -> rows 0..7, high words -> rows 8..15
        uint32_t clo[4], chi[4], dlo[4], dhi[4];
#ifdef G3_ABL_NODECODE
        // ablation: no decode arithmetic (the loaded words go to the matrix instructions as they are)
        #pragma unroll
        for (int q = 0; q < 4; ++q) { clo[q] = Wx[1 + (q % K)]; chi[q] = Wx[1 + ((q + 1) % K)] ^ Wx[0]; dlo[q] = Wx[1 + ((q + 2) % K)]; dhi[q] = Wx[1 + ((q + 3) % K)]; }
        if (false)
#endif
        g3_static_for<0, 4>([&] (auto qc)
        {
            constexpr int q = decltype(qc)::value;
            half4_t bc[2], bd[2];
            decode_quad<K, CB, RAW ? 1 : 0, 8 * q>(Wx, bc);
            decode_quad<K, CB, RAW ? 1 : 0, 8 * q + 4>(Wx, bd);
            union { half4_t h; uint32_t w[2]; } uc, ud; uc.h = bc[0]; ud.h = bd[0];
            clo[q] = uc.w[0]; chi[q] = uc.w[1]; dlo[q] = ud.w[0]; dhi[q] = ud.w[1];
            __builtin_amdgcn_sched_barrier(0);          // bound live ranges: 8 weights in flight at a time (occupancy > ILP here)
        });
#ifndef G3_REFILL_EARLY
        // refill the slot once its words are dead (clamped: a harmless reload at the end of the slice): the load lands in place, the ring needs no register
        // copies, and the compiler's waits in the one-chunk loop are counted -- vmcnt(3) / (3) / (2) / (2) at NR = 4 (hipcc 7.2 ISA; tests/test_isa_guards.py)
        load_lane_words<K>(slot, step_ptr(refill));
#endif
        union { uint32_t w[4]; half8_t h; } bc[2], bd[2];           // [k 0..7 | k 8..15 of the lane's tile row] of column c / c + 8
        #pragma unroll
        for (int q = 0; q < 4; ++q) { bc[0].w[q] = clo[q]; bc[1].w[q] = chi[q]; bd[0].w[q] = dlo[q]; bd[1].w[q] = dhi[q]; }
        #pragma unroll
        for (int h = 0; h < 2; ++h)
        {
#ifndef G3_ABL_NOLDS
            if constexpr (!AF_EARLY)
            {
                #pragma unroll
                for (int i = 0; i < MT; ++i) af[h][i] = *((const half8_t*) (arow[i] + kloc + 8 * h));
            }
#endif
            #pragma unroll
            for (int i = 0; i < MT; ++i)
            {
#ifdef G3_ABL_NOMFMA
                // ablation: no matrix instructions (the operands stay live through one cheap VALU op each)
                acc[i][0][0] += (float) af[h][i][0] + (float) bc[h].h[0] + (float) bc[h].h[2] + (float) bc[h].h[4] + (float) bc[h].h[6];
                acc[i][1][0] += (float) af[h][i][1] + (float) bd[h].h[0] + (float) bd[h].h[2] + (float) bd[h].h[4] + (float) bd[h].h[6];
#else
                acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[h][i], bc[h].h, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[h][i], bd[h].h, acc[i][1], 0, 0, 0);
#endif
            }
        }
    };

    // activations of blocks [c0, c0 + cnt) of the slice -> LDS, row-major (the chunk's first batch of rows may already be in `cv` / `nx`)
    auto stage_chunk = [&] (int c0, int cnt, bool first)
    {
        if constexpr (in_rotated)
        {
            for (int base = 0; base < m; base += 4 * cp_rows)
            {
                if (base > 0 || !first) copy_load(c0, cnt, base, cv);
                copy_store(cnt, base, cv);
            }
        }
        else
        {
            // cooperative input Hadamards: task t = (block t / m, row t % m), one per half-wave, software pipelined by one
            const int ntask = cnt * m;
            const int trips = (ntask + nhw - 1) / nhw;
            if (!first) nx = fetch(c0, cnt, 0);
            for (int it = 0; it < trips; ++it)
            {
                const PrepIn cur = nx;
                if (it + 1 < trips) nx = fetch(c0, cnt, it + 1);
                const int t = it * nhw + hwid;
                const bool act = t < ntask;
                const int tc = min(t, ntask - 1);
                const int blk_l = gemv_udiv(tc, mg_m), row = tc - blk_l * m;
                const half4_t xv = cur.xv * cur.sv;
                float h0 = (float) xv.x, h1 = (float) xv.y, h2 = (float) xv.z, h3 = (float) xv.w;
                had128_f32x4(h0, h1, h2, h3, l32);
                const half4_t o = { f2h(h0 * HAD_R_SCALE_128), f2h(h1 * HAD_R_SCALE_128), f2h(h2 * HAD_R_SCALE_128), f2h(h3 * HAD_R_SCALE_128) };
                if (act) *((half4_t*) (xa + (size_t) row * ldx + blk_l * 128 + 4 * l32)) = o;
            }
        }
    };
    // the chunk's streaming: `trips` trips of NR steps from slice-local step st (slot 0 first: every chunk but the slice's last is a whole number of trips)
    auto stream = [&] (int st, int trips)
    {
        int kl = 0;
        for (int t = 0; t < trips; ++t)                  // plain counted loop (an early exit makes the compiler drain vmcnt every trip)
        {
            g3_static_for<0, NR>([&] (auto uc) { constexpr int u = decltype(uc)::value; do_step(ring[u], kl + 64 * u, st + u + NR); });
            kl += 64 * NR; st += NR;
        }
        return kl;
    };

    // chunks of `chb` Hadamard blocks of activations in LDS.  The host makes every chunk but the last a whole number of trips (2 chb % NR == 0), so the
    // ring position at a chunk boundary is always slot 0 and only the slice's last chunk has a remainder (steps are even: 0 or 2 of them at NR = 4).
    // The one-chunk slice (every 16-row launch of the decode step but the lm_head) is its own straight-line path: inside the chunk loop the streaming
    // loop is a NESTED loop, and there the compiler's wait-count analysis gives up (vmcnt(0) at the loop head again)
    int kl_last = 0, last_cnt = nb;
    if (nb <= chb)
    {
        stage_chunk(0, nb, true);
        __syncthreads();
        G3_T(2);
        kl_last = stream(0, (2 * nb) / NR);
    }
    else
    {
        // several chunks, ONE flat loop over the slice's trips with the chunk boundary as a conditional block at the top of a trip (a nested loop per
        // chunk lost the counted waits, see above): at a boundary every wave has finished the previous chunk's LDS reads (barrier), the chunk is staged,
        // and the trip continues with chunk-local k = 0.  Every chunk but the last is a whole number of trips (host: 2 chb % NR == 0).
        stage_chunk(0, chb, true);
        __syncthreads();
        G3_T(2);
        const int tpc = (2 * chb) / NR;                  // trips per full chunk
        const int total = (2 * nb) / NR;
        int kl = 0, st = 0, next_b = tpc, c0 = 0;
        for (int t = 0; t < total; ++t)
        {
            if (t == next_b)
            {
                c0 += chb; next_b += tpc; kl = 0;
                __syncthreads();
                stage_chunk(c0, min(chb, nb - c0), false);
                __syncthreads();
            }
            g3_static_for<0, NR>([&] (auto uc) { constexpr int u = decltype(uc)::value; do_step(ring[u], kl + 64 * u, st + u + NR); });
            kl += 64 * NR; st += NR;
        }
        // (a last chunk shorter than a trip is staged here; its steps are the remainder below)
        if (total == next_b && c0 + chb < nb)
        {
            c0 += chb; kl = 0;
            __syncthreads();
            stage_chunk(c0, nb - c0, false);
            __syncthreads();
        }
        kl_last = kl; last_cnt = nb - c0;
    }
    // remainder of the slice's last chunk: the first (2 last_cnt) % NR slots, in order
    {
        const int rem = (2 * last_cnt) % NR;
        g3_static_for<0, NR - 1>([&] (auto uc) { constexpr int u = decltype(uc)::value; if (u < rem) do_step(ring[u], kl_last + 64 * u, nsteps - 1); });
    }
    G3_T(3);

    // ---- epilogue.  D layout: rows 16 i + 4 kg + r; acc[i][0]: column c of tile t2 = 32 wave + 16 (mj >> 3) + (mj & 7), acc[i][1]: that + 8
    // (lane index laundered through an empty asm: the output addresses are computed here, not hoisted above the streaming loop where they
    // would cost registers -- the prologue spilled to scratch otherwise, and any scratch use slows every launch)
    if constexpr (RAW)
    {
        const float kinv = (float) u16_as_half(0x1eeeu), kbias = (float) u16_as_half(0xc931u);
        #pragma unroll
        for (int i = 0; i < MT; ++i)
            #pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                float rs = (RSUM_EARLY && (lane & 15) < nb) ? rsum[i][r] : 0.0f;
                for (int blk = (lane & 15) + (RSUM_EARLY ? 16 : 0); blk < nb; blk += 16)
                    rs += a.mat[mi].xsum[(size_t) min(16 * i + 4 * (lane >> 4) + r, m - 1) * (a.k >> 7) + (k0s >> 7) + blk];
                #pragma unroll
                for (int o = 1; o < 16; o <<= 1) rs += xor_lane(rs, o);                      // the 16 lanes' blocks
                const float b = kbias * rs;
                acc[i][0][r] = acc[i][0][r] * kinv + b; acc[i][1][r] = acc[i][1][r] * kinv + b;
            }
    }
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int mj_e = lane_e & 15, kg_e = lane_e >> 4;
    if (a.S > 1 || (a.flags & GEMV_OUT_DEFERRED))
    {
        float* slab = a.workspace + ws_off + ((size_t) cbl * a.S + s) * (size_t) m * 128 + 32 * wv + 16 * (mj_e >> 3) + (mj_e & 7);
        #pragma unroll
        for (int i = 0; i < MT; ++i)
        {
            #pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int row = 16 * i + 4 * kg_e + r;
                if (row < m && has_cb) { slab[row * 128] = acc[i][0][r]; slab[row * 128 + 8] = acc[i][1][r]; }
            }
        }
#ifdef G2_TIMING
        if (tid == 0)
        {
            tstamp[4] = tstamp[3];
            tstamp[5] = __builtin_amdgcn_s_memrealtime();
            uint64_t* dbg = (uint64_t*) a.ws_debug + (size_t) (blockIdx.y * gridDim.x + blockIdx.x) * 8;
            for (int i = 0; i < 6; ++i) dbg[i] = tstamp[i];
            uint32_t xcc, hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            dbg[6] = (xcc & 15u) | ((uint64_t) hwid << 8); dbg[7] = __builtin_amdgcn_s_memtime() - cyc0;      // which CU the workgroup ran on (placement statistics)
        }
#endif
        return;
    }

    // S == 1: the output Hadamard needs whole 128-column rows -> through LDS (over the activations): [column block of the workgroup][m][128] fp32
    __syncthreads();
    float* part = (float*) smem + (size_t) (wave >> 2) * m * 128;
    #pragma unroll
    for (int i = 0; i < MT; ++i)
    {
        #pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const int row = 16 * i + 4 * kg_e + r;
            const int col = 32 * wv + 16 * (mj_e >> 3) + (mj_e & 7);
            if (row < m) { part[row * 128 + col] = acc[i][0][r]; part[row * 128 + col + 8] = acc[i][1][r]; }
        }
    }
    __syncthreads();

    // task t = (column block t / m of the workgroup, row t % m), one per half-wave
    const int l = lane_e & 31, hw_e = 2 * wave + (lane_e >> 5);
    const int cb0 = (cbg - a.mat[mi].cb_first) * CPW, ncb_here = min(CPW, (n >> 7) - cb0);
    void* C_m = a.mat[mi].C;
    for (int t = hw_e; t < ncb_here * m; t += 2 * NWT)
    {
        const int cbw = gemv_udiv(t, mg_m), row = t - cbw * m, cbo = cb0 + cbw;
        const half_t* svh = a.mat[mi].svh + cbo * 128;
        const half_t* bias = a.mat[mi].bias ? a.mat[mi].bias + cbo * 128 : nullptr;
        float4_t v = ((const float4_t*) ((const float*) smem + ((size_t) cbw * m + row) * 128))[l];
        float h0 = v.x, h1 = v.y, h2 = v.z, h3 = v.w;
        had128_f32x4(h0, h1, h2, h3, l);
        h0 *= HAD_R_SCALE_128; h1 *= HAD_R_SCALE_128; h2 *= HAD_R_SCALE_128; h3 *= HAD_R_SCALE_128;
        half4_t sc = ((const half4_t*) svh)[l];
        size_t off = ((size_t) a.c_row_offset + row) * n + cbo * 128 + 4 * l;
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
#ifndef G2_K
#error "compile with -DG2_K=<bits per weight>"
#endif

template <int CB>
static void g3_launch_cb(int mt, bool raw, int nwt, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args)
{
    const bool rot = (args.flags & GEMV_IN_ROTATED) != 0;
    for (int i = 0; i < args.num_mats; ++i) if (!args.mat[i].xsum) raw = false;       // the raw variant takes the activation sums from the producer's block sums
    #define L3(M, R, V, N) { if constexpr (g3_nwt_ok(G2_K, M, R, V && CB == EXL3_CB_MUL1, N)) exl3_gemm3_kernel<G2_K, CB, M, R, V, N><<<grid, dim3(64 * N), lds, st>>>(args); }
    // 8- / 16-wave workgroups (2 / 4 column blocks sharing one activation tile): 16-row passes only
    #define L2(M, R, V) { if (M == 1 && nwt == 16) L3(M, R, V, (M == 1 ? 16 : 4)) else if (M == 1 && nwt == 8) L3(M, R, V, (M == 1 ? 8 : 4)) else L3(M, R, V, 4) }
    // the raw variant serves the fused decode pipeline (rotated input); the standalone op keeps the reference's fp16-rounded weights
    #define L(M, R) { if constexpr (CB == EXL3_CB_MUL1 && R) { if (raw) L2(M, R, true) else L2(M, R, false) } else L2(M, R, false) }
    if (mt == 1)      { if (rot) L(1, true) else L(1, false) }
    else if (mt == 2) { if (rot) L(2, true) else L(2, false) }
    else              { if (rot) L(4, true) else L(4, false) }
    #undef L
    #undef L2
    #undef L3
}

#define G3_CAT_(a, b) a##b
#define G3_CAT(a, b) G3_CAT_(a, b)

// mt = row tiles of 16 per pass: 1, 2 or 4; nwt = waves per workgroup: 4, 8 or 16 (exl3_gemm3_max_waves: what the instantiation can run with)
void G3_CAT(exl3_gemm3_launch_k, G2_K)(int cb, int mt, int var, int nwt, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args)
{
    if (cb == 0) g3_launch_cb<0>(mt, false, nwt, grid, lds, st, args);
    else if (cb == 1) g3_launch_cb<1>(mt, false, nwt, grid, lds, st, args);
    else g3_launch_cb<2>(mt, var == 1, nwt, grid, lds, st, args);
}

// largest workgroup (in waves) the (K, row tiles, rotated input) instantiations can run with: the register budgets of g3_waves_per_eu
int G3_CAT(exl3_gemm3_max_waves_k, G2_K)(int mt, int rot)
{
    if (mt != 1) return 4;
    const bool r = rot != 0;
    return g3_nwt_ok(G2_K, 1, r, r, 16) ? 16 : (g3_nwt_ok(G2_K, 1, r, r, 8) ? 8 : 4);
}

#if G2_K == 4
// LDS bytes of a launch: activations of one chunk; the S == 1 epilogue's [m][128] fp32 overlays them
size_t exl3_gemm3_lds_bytes(int m, int chunk_blocks, int nwt)
{
    const size_t stream = (size_t) m * (chunk_blocks * 128 + G3_XPAD) * 2;
    const size_t part = (size_t) m * 128 * 4 * (nwt / 4);
    return stream > part ? stream : part;
}
#endif
