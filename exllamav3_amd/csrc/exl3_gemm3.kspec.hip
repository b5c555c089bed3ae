// EXL3 quantized small-m GEMM for gfx950, kernel generation 3 ("decode -> LDS transpose -> 16x16x32 MFMA"), 5..64 rows per pass.
//
//   C = ((A * suh) H) @ dequant(B) H * svh (+ bias)        reference: quant/exl3_gemm_kernel.cuh:8-80, quant/exl3_gemm_inner.cuh
//                                                          (semantics only; the reference streams 16 rows per pass)
//
// Why a third kernel: generation 2 (exl3_gemv2.kspec.hip) multiplies straight out of the decoding lane's registers with
// v_mfma_f32_4x4x4_16B_f16, which does a quarter of the matrix pipe's work per cycle.  That is free at m <= 4 (the decode VALU work
// dominates) but at 16 rows the kernel is MFMA-bound: 1024 MFMA cycles against ~660 decode cycles per tile row.  Here the fp16 weights
// go through a wave-private 4.5 KB LDS buffer into the B-operand layout of v_mfma_f32_16x16x32_f16: 64 MFMA cycles per 16 rows per
// decode step, so the weight pass is decode-bound again and a 32-row pass costs about what a 4-row pass does.
//
// Work split: a workgroup = 4 waves = one 128-column block x one k-slice, as in generation 2, but the waves split the COLUMNS (32 each),
// not k: every wave walks the whole slice.  Nothing is reduced across waves -- no partial sums through LDS, no barrier after the
// streaming loop, 8 accumulator VGPRs per 16 rows -- and all waves have exactly the same amount of work.
//
// Lane geometry of a decode step (4 tile rows x the wave's 2 tiles): lane = 16 g + 8 t2 + c owns, as in exl3_lane_decode.cuh, the K
// words [cK, cK + K) of tile (2 wave + t2) of tile row 4 step + g = columns c, c + 8 of that tile, 16 k each: one 16-byte load per lane
// (K = 4), 256 contiguous bytes per 16-lane group.
//
// LDS transpose.  The decoded step is 4 chunk rows r = 2 h + half (h: column c or c + 8, half: k 0..7 or 8..15 of the tile row), each
// 64 lanes x 16 B, row pitch 1152 B; written with ds_write_b128 at r * 1152 + lane * 16 (contiguous).  MFMA (p, t) -- tile rows 2p, 2p + 1
// of the step x tile t -- reads for lane (j, kg) the chunk of writer lane 16 (2p + (kg >> 1)) + 8 t + (j & 7) in row 2 (j >> 3) + (kg & 1):
// the 128-byte pitch offset puts the four 64-byte runs of every 16-lane ds_read_b128 group on disjoint banks.
// k inside an MFMA is natural (k = 8 kg + i over the two tile rows), so the A operand comes straight from a row-major fp16 copy of the
// rotated activations (row pitch padded by 32 B: conflict-free ds_read_b128 across the 16 rows).
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

#define G3_STG_ROW 1152
#define G3_STG_BYTES (4 * G3_STG_ROW)
#define G3_XPAD 16
#define G3_WAVES 4

constexpr int g3_waves_per_eu(int K, int MT, bool ROT, bool RAW)
{
    (void) RAW;                                      // the raw variant is instantiated for the rotated-input prologue only, whose budgets cover its row-sum accumulators
    return MT == 4 ? (ROT ? (K >= 5 ? 2 : 3) : ((K == 4 || K <= 2) ? 4 : 3)) : ((K >= 7 || (ROT && K >= 5)) ? 3 : ((ROT || K >= 5) ? 4 : 5));
}

template <int K, int CB, int MT, bool ROT, bool RAWV>
// RAWV (mul1 codebook, rotated input = the fused decode pipeline, the library's default GEMV variant): the weights enter the matrix instructions as the packed byte sums 1024 + s the
// decode produces (exact fp16 integers) instead of fp16(kinv * (1024 + s) + kbias); the affine map is applied once per output,
// out = kinv * acc + kbias * sum_k x[row][k], with the row sums accumulated by one extra matrix instruction per 32 k against a ones operand (they
// land in the accumulator layout of the outputs).  Removes a v_pk_fma_f16 per weight pair, 14 % of the streaming loop's VALU instructions; the
// same arithmetic as generation 2's default variant.
// ROT: the input is already rotated (fused decode pipeline) -- a separate instantiation so that neither prologue's registers burden the other.
// five 4-wave workgroups per CU (LDS: 5 x 31 KB) need <= 96 VGPRs; the 64-row passes, the ROT prologue (16 registers of activation copy in
// flight next to the weight ring) and the wide rings of K >= 5 take the next register budgets instead of spilling (any scratch use slows
// every launch); the table below is what hipcc 7.2 needs for zero scratch in every (K, codebook, MT, ROT) instantiation
__global__ __launch_bounds__(64 * G3_WAVES) __attribute__((amdgpu_waves_per_eu(g3_waves_per_eu(K, MT, ROT, RAWV && CB == EXL3_CB_MUL1))))
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
    const int n = a.mat[mi].n, cbl = cbg - a.mat[mi].cb_first, ws_off = a.mat[mi].ws_offset;
    const int tiles_n = n >> 4;
    const int k0s = s * a_kslice;
    const int k1s = min(k0s + a_kslice, a_k);
    const int nb = (k1s - k0s) >> 7;                 // Hadamard blocks in the slice; every wave streams all of them
    const int chb = a_chb;

    // LDS carve: [wave-private transpose buffers] [activations of one chunk, row-major fp16]; the S == 1 epilogue reuses it for [m][128] fp32
    const int ldx = chb * 128 + G3_XPAD;
    char* stg = smem + (size_t) wave * G3_STG_BYTES;
    half_t* xa = (half_t*) (smem + (size_t) G3_WAVES * G3_STG_BYTES);

    const int l32 = lane & 31;
    constexpr bool in_rotated = ROT;
    const half_t* __restrict__ x_src = in_rotated ? a.mat[mi].xh : a_A;
    const int hwid = tid >> 5, nhw = G3_WAVES * 2;

    // decode geometry
    const int g = lane >> 4, t2 = (lane >> 3) & 1, c = lane & 7;
    const size_t row_stride = (size_t) tiles_n * NW;
    const uint32_t* __restrict__ strip = Bm + ((size_t) (k0s >> 4) + g) * row_stride + (size_t) (cbl * 8 + 2 * wave + t2) * NW + (size_t) c * K;
    const int prev_lane_addr = ((lane & ~7) | ((lane - 1) & 7)) << 2;

    constexpr bool RAW = RAWV && CB == EXL3_CB_MUL1;
    float4_t acc[MT][2];
    #pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i][0] = float4_t{ 0.f, 0.f, 0.f, 0.f }; acc[i][1] = acc[i][0]; }
    float4_t rsum[RAW ? MT : 1];                     // RAW: sum_k x[row][k] for the rows of acc[i][*] (every column lane holds the same value)
    #pragma unroll
    for (int i = 0; i < (RAW ? MT : 1); ++i) rsum[i] = float4_t{ 0.f, 0.f, 0.f, 0.f };

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
    const int cp_sl = chb <= 1 ? 4 : (chb <= 2 ? 5 : (chb <= 4 ? 6 : 7));
    const int cp_piece = tid & ((1 << cp_sl) - 1), cp_rsub = tid >> cp_sl, cp_rows = (64 * G3_WAVES) >> cp_sl;
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

    // weight ring: two slots of one Hadamard block (8 tile rows = 2 decode steps) each
    LaneWords<K> ring[2][2];
    auto load_block = [&] (LaneWords<K> (&slot)[2], int blk)
    {
        const uint32_t* p = strip + (size_t) (8 * min(blk, nb - 1)) * row_stride;
        load_lane_words<K>(slot[0], p);
        load_lane_words<K>(slot[1], p + 4 * row_stride);
    };
    load_block(ring[0], 0);
    load_block(ring[1], 1);
    G3_T(1);

    // MFMA operand geometry
    const int mj = lane & 15, kg = lane >> 4;
    const char* bsrc = stg + (2 * (mj >> 3) + (kg & 1)) * G3_STG_ROW + ((kg >> 1) * 16 + (mj & 7)) * 16;   // + (32 p + 8 t) * 16
    char* bdst = stg + lane * 16;                                                                        // + r * G3_STG_ROW
    const half_t* arow[MT];
    #pragma unroll
    for (int i = 0; i < MT; ++i) arow[i] = xa + (size_t) min(16 * i + mj, m - 1) * ldx + 8 * kg;

    // one Hadamard block (2 decode steps) from a ring slot; kloc = chunk-local k of the block
    auto do_block = [&] (LaneWords<K> (&slot)[2], int kloc, int refill_blk)
    {
        #pragma unroll
        for (int sub = 0; sub < 2; ++sub)
        {
            uint32_t Wx[K + 1];
            #pragma unroll
            for (int i = 0; i < K; ++i) Wx[i + 1] = slot[sub].w[i];
            Wx[0] = (uint32_t) __builtin_amdgcn_ds_bpermute(prev_lane_addr, (int) slot[sub].w[K - 1]);
            // refill the slot half that was just consumed
            load_lane_words<K>(slot[sub], strip + (size_t) (8 * refill_blk + 4 * sub) * row_stride);

            // exact fp16 weights: quad q of column c / c + 8 = rows {2q, 2q+1 | 2q+8, 2q+9}: low words -> rows 0..7, high words -> rows 8..15
            uint32_t clo[4], chi[4], dlo[4], dhi[4];
            g3_static_for<0, 4>([&] (auto qc)
            {
                constexpr int q = decltype(qc)::value;
                half4_t bc[2], bd[2];
                decode_quad<K, CB, RAW ? 1 : 0, 8 * q>(Wx, bc);
                decode_quad<K, CB, RAW ? 1 : 0, 8 * q + 4>(Wx, bd);
                union { half4_t h; uint32_t w[2]; } uc, ud; uc.h = bc[0]; ud.h = bd[0];
                clo[q] = uc.w[0]; chi[q] = uc.w[1]; dlo[q] = ud.w[0]; dhi[q] = ud.w[1];
                __builtin_amdgcn_sched_barrier(0);      // bound live ranges: 8 weights in flight at a time (occupancy > ILP here)
            });
            *((uint4_t*) (bdst + 0 * G3_STG_ROW)) = uint4_t{ clo[0], clo[1], clo[2], clo[3] };
            *((uint4_t*) (bdst + 1 * G3_STG_ROW)) = uint4_t{ chi[0], chi[1], chi[2], chi[3] };
            *((uint4_t*) (bdst + 2 * G3_STG_ROW)) = uint4_t{ dlo[0], dlo[1], dlo[2], dlo[3] };
            *((uint4_t*) (bdst + 3 * G3_STG_ROW)) = uint4_t{ dhi[0], dhi[1], dhi[2], dhi[3] };
            // the wave's own LDS operations complete in order: no wait between its stores and the loads below, only compiler ordering
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int p = 0; p < 2; ++p)
            {
                half8_t af[MT];
                #pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = *((const half8_t*) (arow[i] + kloc + 64 * sub + 32 * p));
                if constexpr (RAW)
                {
                    const half8_t ones = { 1, 1, 1, 1, 1, 1, 1, 1 };
                    #pragma unroll
                    for (int i = 0; i < MT; ++i) rsum[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], ones, rsum[i], 0, 0, 0);
                }
                #pragma unroll
                for (int t = 0; t < 2; ++t)
                {
                    const half8_t bf = *((const half8_t*) (bsrc + (32 * p + 8 * t) * 16));
                    #pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf, acc[i][t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    };

    for (int c0 = 0; c0 < nb; c0 += chb)
    {
        const int cnt = min(chb, nb - c0);
        if (c0 > 0) __syncthreads();

        // ---- activations of blocks [c0, c0 + cnt) -> LDS, row-major
        if constexpr (in_rotated)
        {
            // the first batch of rows is already in registers (requested before the weight ring, or during the previous chunk's streaming)
            for (int base = 0; base < m; base += 4 * cp_rows)
            {
                if (base > 0) copy_load(c0, cnt, base, cv);
                copy_store(cnt, base, cv);
            }
        }
        else
        {
            // cooperative input Hadamards: task t = (block t / m, row t % m), one per half-wave, software pipelined by one
            const int ntask = cnt * m;
            const int trips = (ntask + nhw - 1) / nhw;
            if (c0 > 0) nx = fetch(c0, cnt, 0);
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
        __syncthreads();
        if (c0 == 0) { G3_T(2); }
        // next chunk's first rows: in flight underneath this chunk's streaming
        if constexpr (in_rotated) { if (c0 + chb < nb) copy_load(c0 + chb, min(chb, nb - c0 - chb), 0, cv); }

        // ---- streaming: blocks c0 .. c0 + cnt - 1, two per trip (one per ring slot); an odd tail swaps the slots
        int b = c0;
        for (; b + 1 < c0 + cnt; b += 2)
        {
            do_block(ring[0], 128 * (b - c0), min(b + 2, nb - 1));
            do_block(ring[1], 128 * (b + 1 - c0), min(b + 3, nb - 1));
        }
        if (b < c0 + cnt)
        {
            do_block(ring[0], 128 * (b - c0), min(b + 2, nb - 1));
            #pragma unroll
            for (int sub = 0; sub < 2; ++sub) { LaneWords<K> t = ring[0][sub]; ring[0][sub] = ring[1][sub]; ring[1][sub] = t; }
        }
    }
    G3_T(3);

    // ---- epilogue.  D layout of acc[i][t]: rows 16 i + 4 kg + r, column 32 wave + 16 t + mj
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
                const float b = kbias * rsum[i][r];
                acc[i][0][r] = acc[i][0][r] * kinv + b; acc[i][1][r] = acc[i][1][r] * kinv + b;
            }
    }
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int mj_e = lane_e & 15, kg_e = lane_e >> 4;
    if (a.S > 1 || (a.flags & GEMV_OUT_DEFERRED))
    {
        float* slab = a.workspace + ws_off + ((size_t) cbl * a.S + s) * (size_t) m * 128 + 32 * wave + mj_e;
        #pragma unroll
        for (int i = 0; i < MT; ++i)
        {
            #pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int row = 16 * i + 4 * kg_e + r;
                if (row < m) { slab[row * 128] = acc[i][0][r]; slab[row * 128 + 16] = acc[i][1][r]; }
            }
        }
#ifdef G2_TIMING
        if (tid == 0)
        {
            tstamp[4] = tstamp[3];
            tstamp[5] = __builtin_amdgcn_s_memrealtime();
            uint64_t* dbg = (uint64_t*) a.ws_debug + (size_t) (blockIdx.y * gridDim.x + blockIdx.x) * 8;
            for (int i = 0; i < 6; ++i) dbg[i] = tstamp[i];
            uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            dbg[6] = xcc; dbg[7] = __builtin_amdgcn_s_memtime() - cyc0;
        }
#endif
        return;
    }

    // S == 1: the output Hadamard needs whole 128-column rows -> through LDS (over the transpose buffers / activations)
    __syncthreads();
    float* part = (float*) smem;
    #pragma unroll
    for (int i = 0; i < MT; ++i)
    {
        #pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const int row = 16 * i + 4 * kg_e + r;
            if (row < m) { part[row * 128 + 32 * wave + mj_e] = acc[i][0][r]; part[row * 128 + 32 * wave + 16 + mj_e] = acc[i][1][r]; }
        }
    }
    __syncthreads();

    const int l = lane_e & 31, hw8 = 2 * wave + (lane_e >> 5);
    const half_t* svh = a.mat[mi].svh + cbl * 128;
    const half_t* bias = a.mat[mi].bias ? a.mat[mi].bias + cbl * 128 : nullptr;
    void* C_m = a.mat[mi].C;
    for (int base = 0; base < m; base += G3_WAVES * 2)
    {
        int row = base + hw8;
        bool act = row < m;
        float4_t v = ((const float4_t*) (part + (act ? row : 0) * 128))[l];
        float h0 = v.x, h1 = v.y, h2 = v.z, h3 = v.w;
        had128_f32x4(h0, h1, h2, h3, l);
        h0 *= HAD_R_SCALE_128; h1 *= HAD_R_SCALE_128; h2 *= HAD_R_SCALE_128; h3 *= HAD_R_SCALE_128;
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
#ifndef G2_K
#error "compile with -DG2_K=<bits per weight>"
#endif

template <int CB>
static void g3_launch_cb(int mt, bool raw, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args)
{
    const bool rot = (args.flags & GEMV_IN_ROTATED) != 0;
    #define L2(M, R, V) exl3_gemm3_kernel<G2_K, CB, M, R, V><<<grid, dim3(64 * G3_WAVES), lds, st>>>(args)
    // the raw variant serves the fused decode pipeline (rotated input); the standalone op keeps the reference's fp16-rounded weights
    #define L(M, R) { if constexpr (CB == EXL3_CB_MUL1 && R) { if (raw) L2(M, R, true); else L2(M, R, false); } else L2(M, R, false); }
    if (mt == 1)      { if (rot) L(1, true) else L(1, false) }
    else if (mt == 2) { if (rot) L(2, true) else L(2, false) }
    else              { if (rot) L(4, true) else L(4, false) }
    #undef L
    #undef L2
}

#define G3_CAT_(a, b) a##b
#define G3_CAT(a, b) G3_CAT_(a, b)

// mt = row tiles of 16 per pass: 1, 2 or 4
void G3_CAT(exl3_gemm3_launch_k, G2_K)(int cb, int mt, int var, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args)
{
    if (cb == 0) g3_launch_cb<0>(mt, false, grid, lds, st, args);
    else if (cb == 1) g3_launch_cb<1>(mt, false, grid, lds, st, args);
    else g3_launch_cb<2>(mt, var == 1, grid, lds, st, args);
}

#if G2_K == 4
// LDS bytes of a launch: transpose buffers + activations of one chunk; the S == 1 epilogue's [m][128] fp32 overlays them
size_t exl3_gemm3_lds_bytes(int m, int chunk_blocks)
{
    const size_t stream = (size_t) G3_WAVES * G3_STG_BYTES + (size_t) m * (chunk_blocks * 128 + G3_XPAD) * 2;
    const size_t part = (size_t) m * 128 * 4;
    return stream > part ? stream : part;
}
#endif
