// EXPERIMENT (round 3), not part of the product library: built together with mlp1_harness.hip against libexl3_hip.so (see that file).  Measured on
// MI355X: parity with the three-launch form to 4e-4 of the output RMS; 35.6-36.3 vs 31.5-32.5 us per MLP with the counter-based exchange
// (profiles/r03_mlp1_fused_launch.json), 31.9-32.6 vs 32.0-32.7 us with -DM1_TAGGED (profiles/r03_mlp1_fused_launch_tagged.json): equal time, one launch instead of
// three -- no gain at Llama-3.1-8B shapes (profiles/NOTES.md B section 6): the full-row prologue and the reductions are exposed because every CU runs one workgroup in lockstep.
//
// The MLP block of a batch-1 decode step in ONE launch (K = 4, mul1 codebook, FAST variant, one row):
//
//   R += down( silu(gate(n)) * up(n) ),   n = RMSNorm(R) * w          reference: libtorch/mlp.cpp:14-91 (BC_GatedMLP::run_bsz1), activation.cu silu_mul,
//                                                                     three exl3_gemv launches + one activation launch there
//
// It replaces three launches of the fixed-point-residual pipeline (exl3_gemv_ex_fx over gate|up, exl3_glue_act_rs, exl3_gemv_ex ROTATED | ATOMIC
// over down: exl3_gemv4.kspec.hip, exl3_glue.hip) and keeps their arithmetic -- same decode, same matrix instruction, same rounding points; only the
// fp32 summation order over k differs (full-k workgroups here, split-k slices there).
//
// Why one launch is possible without a grid-wide barrier.  gate|up is split by COLUMNS and a workgroup contracts the whole k (= hidden) for its
// 128 columns, so its outputs are complete when its own waves are done: nothing is reduced across workgroups before the non-linearity.  down is split
// into k-slices of 8 Hadamard blocks (1024 intermediate channels); the 8 gate and 8 up column blocks that feed one slice are computed by one TEAM of
// 16 workgroups, which then turn into the 16 workgroups that multiply that slice (2 of down's 32 column blocks each).  The only cross-workgroup
// step is a 16-way all-gather of 256 bytes per member inside a team (agent-scope stores, one arrival counter per team, bounded poll) -- not a grid
// barrier -- and the waves request their first down rows before they wait.  Per address of R the launch adds inter / 1024 = 14 partial rows (the
// three-launch form: 16).  Grid = 16 x (inter / 1024) workgroups of 16 waves, one per CU, ALL co-resident (the launcher refuses more workgroups than
// the device has CUs): 224 for Llama-3.1-8B.
//
// tools/ubench_decode_occupancy.hip (profiles/r03_decode_occupancy_ubench.json) is why 16-wave workgroups (4 waves per SIMD) are acceptable: with a
// two-unit weight ring the decode loop streams at 440 ns per unit per SIMD at 4 waves against 415-420 at 8.
//
// R is read (phase 1, every workgroup: the whole row) and added into (phase 4) in place.  A workgroup may only add once EVERY workgroup has its row
// in registers: a grid-wide arrival counter that each workgroup bumps after phase 1 and checks -- one load, long satisfied -- in front of its atomics.
// All counters clean themselves (the last workgroup through resets them), so a captured graph replays correctly.
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include "exl3_gemv_args.h"
#include "exl3_lane_decode.cuh"
#include "exl3_glue_device.cuh"

#include <type_traits>
#include <stdlib.h>

template <int I, int N, typename F>
__device__ __forceinline__ void m1_static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); m1_static_for<I + 1, N>(f); }
}

#define M1_WAVES 16
#define M1_TEAM 16            // workgroups per team = gate blocks + up blocks of one down k-slice
#define M1_TBLK 8             // Hadamard blocks (128 intermediate channels) per team
#define M1_SPIN_LIMIT (1 << 17)
#define M1_PAD_LDS_BYTES (64 << 10)

struct Mlp1Args
{
    const int64_t* R;                 // [hidden] fixed-point residual (value * 2^32), read in phase 1
    unsigned long long* R_acc;        // the same buffer, added into in phase 4
    const half_t* norm_w; const float* ss_prev; float* ss_out; float eps;
    int hidden, inter, nteam, nwg;
    const uint32_t* Bg; const uint32_t* Bu; const uint32_t* Bd;
    const half_t* suh_g; const half_t* suh_u; const half_t* svh_g; const half_t* svh_u; const half_t* suh_d; const half_t* svh_d;
    unsigned long long* exch;         // [nteam][16 members][32 lanes] x 8 bytes: a member's 128 finished fp16 outputs (M1_TAGGED: [..][64] granules of { 2 x fp16, tag })
    uint32_t* cnt;                    // zero on entry, zero on exit: team t: arrive at [32 t], depart at [32 t + 16]; grid: read gate at [32 nteam], finish at [32 nteam + 16]
    uint32_t* err;                    // sticky: 1 = a team never completed, 2 = the read gate never completed (bounded polls)
    unsigned long long* dbg;          // diagnostics (EXL3_HIP_MLP1_TIMING=1): 8 x 100 MHz timestamps per workgroup, else null
};

// one work unit = 2 tile rows of the wave's 128-column block (exl3_gemv4.kspec.hip g4_unit, two units in flight: slots 2 HALF, 2 HALF + 1)
template <int K, int HALF>
__device__ __forceinline__ void m1_unit(LaneWords<K> (&ring)[4], const uint32_t* __restrict__ refill, size_t row_stride, int lane, half4_t ag,
                                        float4_t& vc, float4_t& vd)
{
    m1_static_for<0, 2>([&] (auto uc)
    {
        constexpr int u = decltype(uc)::value;
        LaneWords<K>& slot = ring[2 * HALF + u];
        uint32_t Wx[K + 1];
        #pragma unroll
        for (int i = 0; i < K; ++i) Wx[i + 1] = slot.w[i];
        {
            const uint32_t wl = slot.w[K - 1];
            const uint32_t r1 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x121, 0xf, 0xf, true);
            const uint32_t r9 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x129, 0xf, 0xf, true);
            Wx[0] = (lane & 7) ? r1 : r9;
        }
        load_lane_words<K>(slot, refill + (size_t) u * row_stride);
        m1_static_for<0, 4>([&] (auto qc)
        {
            constexpr int q = decltype(qc)::value;
            constexpr int ABID = 8 * HALF + 4 * u + q;
            half4_t bc[2], bd[2];
            decode_quad<K, EXL3_CB_MUL1, 1, 8 * q>(Wx, bc);
            decode_quad<K, EXL3_CB_MUL1, 1, 8 * q + 4>(Wx, bd);
            vc = __builtin_amdgcn_mfma_f32_4x4x4f16(ag, bc[0], vc, 4, ABID, 0);
            vd = __builtin_amdgcn_mfma_f32_4x4x4f16(ag, bd[0], vd, 4, ABID, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    });
}

// stream units [ubase, ubase + nun) (nun even) of one 128-column strip against the activation quads in LDS; ring holds units ubase, ubase + 1 on entry
template <int K>
__device__ __forceinline__ void m1_stream(LaneWords<K> (&ring)[4], const uint32_t* __restrict__ strip, size_t row_stride, const char* quads, int ubase, int nun,
                                          int lane, float4_t& vc, float4_t& vd)
{
    const int last_unit = ubase + nun - 1;
    const int quad_lane = (lane >> 2) * 8;                     // lane 4 g + i: (tile row g >> 2 of the group, quad g & 3); one row: every i reads row 0
    auto load_group = [&] (int tr0) -> uint2_t { return *((const uint2_t*) (quads + (size_t) tr0 * 32 + quad_lane)); };
    uint2_t agn = load_group(2 * ubase);
    int unit = ubase;
    for (int j = 0; j < (nun >> 1); ++j)
    {
        const half4_t ag = u2_as_half4(agn.x, agn.y);
        agn = load_group(2 * min(unit + 2, last_unit));
        m1_unit<K, 0>(ring, strip + (size_t) (2 * min(unit + 2, last_unit)) * row_stride, row_stride, lane, ag, vc, vd);
        m1_unit<K, 1>(ring, strip + (size_t) (2 * min(unit + 3, last_unit)) * row_stride, row_stride, lane, ag, vc, vd);
        unit += 2;
    }
}

template <int K>
__device__ __forceinline__ void m1_issue_ring(LaneWords<K> (&ring)[4], const uint32_t* __restrict__ strip, size_t row_stride, int ubase)
{
    #pragma unroll
    for (int u = 0; u < 4; ++u) load_lane_words<K>(ring[u], strip + (size_t) (2 * ubase + u) * row_stride);
}

template <int K>
__global__ __launch_bounds__(64 * M1_WAVES)
void exl3_mlp1_kernel(const Mlp1Args a)
{
    constexpr int NW = 8 * K;
    __shared__ __attribute__((aligned(16))) char quads1[(32 * 8 + 4) * 32];        // activation quads of gate / up's input: 32 blocks x 8 tile rows x 4 quads x 8 bytes (+ slack)
    __shared__ __attribute__((aligned(16))) char quads2[(M1_TBLK * 8 + 4) * 32];   // ... of down's k-slice
    __shared__ __attribute__((aligned(16))) float part[M1_WAVES * 128];
    __shared__ float bsum1[32], bsum2[M1_TBLK], ssq_s[32];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, l32 = lane & 31, hwid = tid >> 5;
    const int j = blockIdx.x, t = blockIdx.y;
#ifdef M1_TAGGED
    // tagged exchange: every 8-byte granule = { two fp16 values, tag }; tag = launch epoch + 1 (the epoch word lives in memory and is bumped by the last
    // workgroup of a launch, so a replayed graph sees a new tag; a stale granule carries an older one): data and flag travel in ONE store, no counter
    const uint32_t tag = a.cnt[32 * a.nteam + 8] + 1u;
#endif
    #define M1_T(i) do { if (a.dbg && tid == 0) a.dbg[(size_t) (t * M1_TEAM + j) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    M1_T(0);
    const int hidden = a.hidden, inter = a.inter, nb1 = hidden >> 7;
    const int mi = j >> 3;                                     // 0: gate, 1: up
    const int cb1 = t * M1_TBLK + (j & 7);                     // column block of the gate / up matrix
    const uint32_t* __restrict__ B1 = mi ? a.Bu : a.Bg;
    const half_t* __restrict__ suh1 = mi ? a.suh_u : a.suh_g;
    const half_t* __restrict__ svh1 = mi ? a.svh_u : a.svh_g;

    // ---------------- phase 1: the residual row -> RMSNorm (previous residual's scale, corrected below) -> x * suh -> 128-point Hadamards -> quads
    // task = Hadamard block, one per half-wave (hidden <= 4096: one round).  Operands first, weight rows behind them.
    const int pb = min(hwid, nb1 - 1);
    const bool pact = hwid < nb1;
    uint4_t f0, f1; half4_t sv, wv; float ssp;
    {
        const uint4_t* fp = (const uint4_t*) (a.R + (size_t) 128 * pb) + 2 * l32;       // 4 x int64 per lane
        f0 = fp[0]; f1 = fp[1];
        sv = ((const half4_t*) (suh1 + 128 * pb))[l32];
        wv = ((const half4_t*) (a.norm_w + 128 * pb))[l32];
        ssp = a.ss_prev[min(l32, nb1 - 1)];
    }
    const int tiles_n1 = inter >> 4;
    const size_t rs1 = (size_t) tiles_n1 * NW;
    const uint32_t* __restrict__ strip1 = B1 + (size_t) cb1 * 8 * NW + (size_t) lane * K;
    const int nun1 = (nb1 * 4) / M1_WAVES;                     // units per wave (hidden % 1024 == 0: even)
    const int ub1 = wave * nun1;
    LaneWords<K> ring[4];
    m1_issue_ring<K>(ring, strip1, rs1, ub1);

    float r_prev;
    {
        auto fx = [] (uint32_t lo, uint32_t hi) -> half_t { return f2h((float) (int32_t) hi + (float) lo * 2.3283064365386963e-10f); };
        half4_t xv = { fx(f0.x, f0.y), fx(f0.z, f0.w), fx(f1.x, f1.y), fx(f1.z, f1.w) };
        const float r0 = (float) xv.x, r1 = (float) xv.y, r2 = (float) xv.z, r3 = (float) xv.w;
        float ssq = r0 * r0;
        ssq = __builtin_fmaf(r1, r1, ssq); ssq = __builtin_fmaf(r2, r2, ssq); ssq = __builtin_fmaf(r3, r3, ssq);
        #pragma unroll
        for (int i = 1; i < 32; i <<= 1) ssq += xor_lane(ssq, i);
        float s2 = l32 < nb1 ? ssp : 0.0f;
        #pragma unroll
        for (int i = 1; i < 32; i <<= 1) s2 += xor_lane(s2, i);
        r_prev = __frsqrt_rn(s2 / (float) hidden + a.eps);
        xv = half4_t{ f2h((float) xv.x * (float) wv.x * r_prev), f2h((float) xv.y * (float) wv.y * r_prev),
                      f2h((float) xv.z * (float) wv.z * r_prev), f2h((float) xv.w * (float) wv.w * r_prev) };
        xv = xv * sv;
        float h0 = (float) xv.x, h1 = (float) xv.y, h2 = (float) xv.z, h3 = (float) xv.w;
        had128_f32x4(h0, h1, h2, h3, l32);
        const half2_t o01 = { f2h(h0 * HAD_R_SCALE_128), f2h(h1 * HAD_R_SCALE_128) };
        const half2_t o23 = { f2h(h2 * HAD_R_SCALE_128), f2h(h3 * HAD_R_SCALE_128) };
        float ts = ((float) o01.x + (float) o01.y) + ((float) o23.x + (float) o23.y);
        #pragma unroll
        for (int i = 1; i < 32; i <<= 1) ts += xor_lane(ts, i);
        if (pact)
        {
            if (l32 == 0) { bsum1[pb] = ts; ssq_s[pb] = ssq; }
            const int tr = pb * 8 + (l32 >> 2);
            const int q0 = 2 * (l32 & 1), sp = (l32 >> 1) & 1;
            char* base = quads1 + (size_t) (tr * 4 + q0) * 8 + sp * 4;
            *((half2_t*) base) = o01;
            *((half2_t*) (base + 8)) = o23;
        }
    }
    __syncthreads();
    M1_T(1);
    // this workgroup has its copy of the row: tell the grid (checked in front of the atomics of phase 4)
    if (tid == 0) __hip_atomic_fetch_add(a.cnt + 32 * a.nteam, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---------------- phase 2: gate / up column block cb1 over the whole k: 4 nb1 units split over the 16 waves
    float4_t vc = { 0.f, 0.f, 0.f, 0.f }, vd = vc;
    m1_stream<K>(ring, strip1, rs1, quads1, ub1, nun1, lane, vc, vd);
    M1_T(2);

    // down's geometry: the team's k-slice = rows [1024 t, +1024); this workgroup's column blocks 2 j, 2 j + 1; wave = (column block wave >> 3, Hadamard block wave & 7)
    const int tiles_n2 = hidden >> 4;
    const size_t rs2 = (size_t) tiles_n2 * NW;
    const int cb2 = 2 * j + (wave >> 3);
    const uint32_t* __restrict__ strip2 = a.Bd + ((size_t) (t * (M1_TBLK * 8)) * tiles_n2 + (size_t) cb2 * 8) * NW + (size_t) lane * K;
    const int ub2 = 4 * (wave & 7);
    // operands of phase 3's tasks (half-waves 0..7) and of the two output half-waves, requested here
    half4_t sv2 = { 0, 0, 0, 0 }, svo = { 0, 0, 0, 0 };
    if (hwid < M1_TBLK) sv2 = ((const half4_t*) (a.suh_d + (size_t) t * (M1_TBLK * 128) + 128 * hwid))[l32];
    if (hwid < 2) svo = ((const half4_t*) (a.svh_d + (size_t) (2 * j + hwid) * 128))[l32];
    const half4_t sv1o = ((const half4_t*) (svh1 + (size_t) cb1 * 128))[l32];
    {
        const int T = lane >> 3, c = lane & 7, col = 16 * T + c;
        part[wave * 128 + col] = vc[0]; part[wave * 128 + col + 8] = vd[0];
    }
    if (wave != 0) m1_issue_ring<K>(ring, strip2, rs2, ub2);  // (wave 0 first publishes: its vmcnt(0) below would wait for these rows too)
    __syncthreads();

    // half-wave 0: this workgroup's 128 finished outputs: sum of the waves' partials, mul1 affine map, output Hadamard, row-scale correction, svh -> fp16
    if (tid < 32)
    {
        float4_t v = ((const float4_t*) part)[l32];
        for (int w = 1; w < M1_WAVES; ++w)
        {
            const float4_t p = ((const float4_t*) (part + w * 128))[l32];
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        float xs = l32 < nb1 ? bsum1[l32] : 0.0f;
        float sn = l32 < nb1 ? ssq_s[l32] : 0.0f, sp_ = l32 < nb1 ? ssp : 0.0f;
        #pragma unroll
        for (int i = 1; i < 32; i <<= 1) { xs += xor_lane(xs, i); sn += xor_lane(sn, i); sp_ += xor_lane(sp_, i); }
        const float kinv = (float) u16_as_half(0x1eeeu), kbias = (float) u16_as_half(0xc931u);
        const float b = kbias * xs;
        v.x = v.x * kinv + b; v.y = v.y * kinv + b; v.z = v.z * kinv + b; v.w = v.w * kinv + b;
        float g0, g1, g2, g3;
        out_had(v, l32, g0, g1, g2, g3);
        // r_new / r_prev (gemv_rescale's arithmetic: both sums through the same 32-lane tree)
        const float rp = __frsqrt_rn(sp_ / (float) hidden + a.eps), rn = __frsqrt_rn(sn / (float) hidden + a.eps);
        const float rsc = rn / rp;
        g0 *= rsc; g1 *= rsc; g2 *= rsc; g3 *= rsc;
        const half4_t gh = half4_t{ f2h(g0), f2h(g1), f2h(g2), f2h(g3) } * sv1o;
#ifdef M1_TAGGED
        union { half4_t h; uint32_t w[2]; } pk; pk.h = gh;
        unsigned long long* dst = a.exch + ((size_t) (t * M1_TEAM + j) * 64 + 2 * l32);
        __hip_atomic_store(dst, (unsigned long long) pk.w[0] | ((unsigned long long) tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, (unsigned long long) pk.w[1] | ((unsigned long long) tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        union { half4_t h; unsigned long long u; } pk; pk.h = gh;
        __hip_atomic_store(a.exch + ((size_t) (t * M1_TEAM + j) * 32 + l32), pk.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // written through and acknowledged before the arrival is announced
        if (l32 == 0) __hip_atomic_fetch_add(a.cnt + 32 * t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    M1_T(3);
#ifdef M1_TAGGED
    if (wave == 0) m1_issue_ring<K>(ring, strip2, rs2, ub2);
#else
    if (wave == 0)
    {
        m1_issue_ring<K>(ring, strip2, rs2, ub2);
        // the team's 16 members have published: bounded poll by one wave
        uint32_t seen = 0; int spins = 0;
        do
        {
            seen = __hip_atomic_load(a.cnt + 32 * t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen < M1_TEAM) __builtin_amdgcn_s_sleep(2);
        } while (seen < M1_TEAM && ++spins < M1_SPIN_LIMIT);
        if (lane == 0)
        {
            if (seen < M1_TEAM) __hip_atomic_fetch_or(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the last member through cleans the team's counters (every member has seen the full count before it departs)
            const uint32_t old = __hip_atomic_fetch_add(a.cnt + 32 * t + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == M1_TEAM - 1)
            {
                __hip_atomic_store(a.cnt + 32 * t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.cnt + 32 * t + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();
#endif
    M1_T(4);

    // ---------------- phase 3: a = fp16(silu(g) * u) of the team's 8 blocks (glue_act's arithmetic) -> * suh_d -> Hadamard -> quads of down's k-slice
    if (hwid < M1_TBLK)
    {
#ifdef M1_TAGGED
        union { half4_t h; uint32_t w[2]; } pg, pu;
        {
            // each lane polls its own four granules (two of the gate block, two of the up block) until all carry this launch's tag; bounded
            const unsigned long long* sg = a.exch + ((size_t) (t * M1_TEAM + hwid) * 64 + 2 * l32);
            const unsigned long long* su = a.exch + ((size_t) (t * M1_TEAM + M1_TBLK + hwid) * 64 + 2 * l32);
            int spins = 0; bool ok;
            do
            {
                const unsigned long long g0 = __hip_atomic_load(sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), g1 = __hip_atomic_load(sg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long u0 = __hip_atomic_load(su, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), u1 = __hip_atomic_load(su + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (uint32_t) (g0 >> 32) == tag && (uint32_t) (g1 >> 32) == tag && (uint32_t) (u0 >> 32) == tag && (uint32_t) (u1 >> 32) == tag;
                pg.w[0] = (uint32_t) g0; pg.w[1] = (uint32_t) g1; pu.w[0] = (uint32_t) u0; pu.w[1] = (uint32_t) u1;
            } while (__builtin_amdgcn_ballot_w64(!ok) != 0ull && ++spins < M1_SPIN_LIMIT);
            if (!ok) __hip_atomic_fetch_or(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#else
        union { half4_t h; unsigned long long u; } pg, pu;
        pg.u = __hip_atomic_load(a.exch + ((size_t) (t * M1_TEAM + hwid) * 32 + l32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pu.u = __hip_atomic_load(a.exch + ((size_t) (t * M1_TEAM + M1_TBLK + hwid) * 32 + l32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        auto silu_mul = [] (half_t g, half_t u) -> half_t { float gf = (float) g; return f2h(gf / (1.0f + __expf(-gf)) * (float) u); };
        half4_t xv = { silu_mul(pg.h.x, pu.h.x), silu_mul(pg.h.y, pu.h.y), silu_mul(pg.h.z, pu.h.z), silu_mul(pg.h.w, pu.h.w) };
        xv = xv * sv2;
        float h0 = (float) xv.x, h1 = (float) xv.y, h2 = (float) xv.z, h3 = (float) xv.w;
        had128_f32x4(h0, h1, h2, h3, l32);
        const half2_t o01 = { f2h(h0 * HAD_R_SCALE_128), f2h(h1 * HAD_R_SCALE_128) };
        const half2_t o23 = { f2h(h2 * HAD_R_SCALE_128), f2h(h3 * HAD_R_SCALE_128) };
        float ts = ((float) o01.x + (float) o01.y) + ((float) o23.x + (float) o23.y);
        #pragma unroll
        for (int i = 1; i < 32; i <<= 1) ts += xor_lane(ts, i);
        if (l32 == 0) bsum2[hwid] = ts;
        const int tr = hwid * 8 + (l32 >> 2);
        const int q0 = 2 * (l32 & 1), sp = (l32 >> 1) & 1;
        char* base = quads2 + (size_t) (tr * 4 + q0) * 8 + sp * 4;
        *((half2_t*) base) = o01;
        *((half2_t*) (base + 8)) = o23;
    }
    __syncthreads();
    M1_T(5);

    // ---------------- phase 4: down: wave = (column block, Hadamard block of the slice): 4 units
    vc = float4_t{ 0.f, 0.f, 0.f, 0.f }; vd = vc;
#ifdef M1_TAGGED
    // the read gate's value requested HERE (almost always complete by now): the check in front of the adds then costs no round trip
    uint32_t gate_seen = 0;
    if (hwid < 2) gate_seen = __hip_atomic_load(a.cnt + 32 * a.nteam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    m1_stream<K>(ring, strip2, rs2, quads2, ub2, 4, lane, vc, vd);
    M1_T(6);
    {
        const int T = lane >> 3, c = lane & 7, col = 16 * T + c;
        part[wave * 128 + col] = vc[0]; part[wave * 128 + col + 8] = vd[0];
    }
    __syncthreads();
    if (hwid < 2)
    {
        // half-wave h: column block 2 j + h = the sum of waves 8 h .. 8 h + 7; affine map; output Hadamard; svh; added into R
        float4_t v = ((const float4_t*) (part + (8 * hwid) * 128))[l32];
        for (int w = 1; w < 8; ++w)
        {
            const float4_t p = ((const float4_t*) (part + (8 * hwid + w) * 128))[l32];
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        float xs = l32 < M1_TBLK ? bsum2[l32] : 0.0f;
        #pragma unroll
        for (int i = 1; i < 32; i <<= 1) xs += xor_lane(xs, i);
        const float kinv = (float) u16_as_half(0x1eeeu), kbias = (float) u16_as_half(0xc931u);
        const float b = kbias * xs;
        v.x = v.x * kinv + b; v.y = v.y * kinv + b; v.z = v.z * kinv + b; v.w = v.w * kinv + b;
        float h0, h1, h2, h3;
        out_had(v, l32, h0, h1, h2, h3);
        const float o[4] = { h0 * (float) svo.x, h1 * (float) svo.y, h2 * (float) svo.z, h3 * (float) svo.w };
        // every workgroup of the grid has read its copy of R?  (long true by now: one load; bounded)
#ifdef M1_TAGGED
        uint32_t seen = gate_seen; int spins = 0;
        while (seen < (uint32_t) a.nwg && ++spins < M1_SPIN_LIMIT) seen = __hip_atomic_load(a.cnt + 32 * a.nteam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        uint32_t seen = 0; int spins = 0;
        do
        {
            seen = __hip_atomic_load(a.cnt + 32 * a.nteam, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (seen < (uint32_t) a.nwg && ++spins < M1_SPIN_LIMIT);
#endif
        if (seen < (uint32_t) a.nwg && l32 == 0) __hip_atomic_fetch_or(a.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long* acc = a.R_acc + (size_t) (2 * j + hwid) * 128 + 4 * l32;
        #pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const long long f = __double2ll_rn((double) o[i] * GEMV_FX_SCALE);
            __hip_atomic_fetch_add(acc + i, (unsigned long long) f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // this residual's block sums of squares for whoever normalises the NEXT residual with them (the next q|k|v launch): one workgroup
    if (j == 0 && t == 0 && tid < nb1) a.ss_out[tid] = ssq_s[tid];
    __syncthreads();
    if (a.dbg && tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); M1_T(7); }
    if (tid == 0)
    {
        // the last workgroup to finish cleans the grid counters (every workgroup has passed its read-gate check before it counts here)
        const uint32_t old = __hip_atomic_fetch_add(a.cnt + 32 * a.nteam + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (uint32_t) a.nwg - 1)
        {
            __hip_atomic_store(a.cnt + 32 * a.nteam, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.cnt + 32 * a.nteam + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef M1_TAGGED
            __hip_atomic_fetch_add(a.cnt + 32 * a.nteam + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // next launch's tag
#endif
        }
    }
}

// ------------------------------------------------------------------------------------------------
#define M1_EXCH_OFFSET_BYTES (48ll << 20)          // the workspace's diagnostics tail [48, 50) MiB: 16 x 30 x 256 bytes at most
#define M1_CNT_OFFSET (EXL3_NUM_TICKETS - 1024)    // the last 1024 tickets: 32 per team (<= 30 teams) + 32 for the grid + the error word

extern "C" int exl3_mlp1_fx(void* R, const void* norm_w, const float* ss_prev, float* ss_out, float eps,
                            const void* B_gate, const void* B_up, const void* suh_g, const void* suh_u, const void* svh_g, const void* svh_u,
                            const void* B_down, const void* suh_d, const void* svh_d, int m, int hidden, int inter, int K, int cb, void* stream)
{
    EXL3_CHECK_ARG(R && norm_w && ss_prev && ss_out && ss_prev != ss_out && B_gate && B_up && suh_g && suh_u && svh_g && svh_u && B_down && suh_d && svh_d,
                   "exl3_mlp1_fx: null pointer / ss_out must differ from ss_prev");
    EXL3_CHECK_ARG(m == 1 && K == 4 && cb == EXL3_CB_MUL1, "exl3_mlp1_fx: one row, 4 bits per weight, mul1 codebook (other configurations take the three-launch form)");
    EXL3_CHECK_ARG(hidden >= 1024 && hidden <= 4096 && hidden % 1024 == 0 && inter >= 1024 && inter % 1024 == 0 && hidden / 128 == 32,
                   "exl3_mlp1_fx: hidden must be 4096 and inter a multiple of 1024 (16 workgroups x 2 column blocks of down per team)");
    hipStream_t st = (hipStream_t) stream;
    Exl3DevCtx* ctx = exl3_get_ctx(st);
    if (!ctx) return EXL3_ERR_INIT;
    const int nteam = inter / 1024, nwg = nteam * M1_TEAM;
    EXL3_CHECK_ARG(nteam <= 30 && nwg <= ctx->num_cus, "exl3_mlp1_fx: every workgroup must be resident at once (16 * inter / 1024 <= CUs)");
    Mlp1Args a;
    a.R = (const int64_t*) R; a.R_acc = (unsigned long long*) R;
    a.norm_w = (const half_t*) norm_w; a.ss_prev = ss_prev; a.ss_out = ss_out; a.eps = eps;
    a.hidden = hidden; a.inter = inter; a.nteam = nteam; a.nwg = nwg;
    a.Bg = (const uint32_t*) B_gate; a.Bu = (const uint32_t*) B_up; a.Bd = (const uint32_t*) B_down;
    a.suh_g = (const half_t*) suh_g; a.suh_u = (const half_t*) suh_u; a.svh_g = (const half_t*) svh_g; a.svh_u = (const half_t*) svh_u;
    a.suh_d = (const half_t*) suh_d; a.svh_d = (const half_t*) svh_d;
    a.exch = (unsigned long long*) ((char*) ctx->workspace + M1_EXCH_OFFSET_BYTES);
    a.cnt = ctx->tickets + M1_CNT_OFFSET;
    a.err = ctx->tickets + M1_CNT_OFFSET + 32 * 31;
    static int timing = -1;
    if (timing < 0) { const char* e = getenv("EXL3_HIP_MLP1_TIMING"); timing = e ? atoi(e) : 0; }
    a.dbg = timing ? (unsigned long long*) ((char*) ctx->workspace + M1_EXCH_OFFSET_BYTES + (256 << 10)) : nullptr;      // read back with exl3_debug_copy_workspace
    // one workgroup per CU: the static LDS (19.5 KB) plus this unused dynamic allocation exceeds half of the CU's 160 KB, so the dispatcher cannot
    // place two of the 16-wave workgroups on one CU while another CU stays empty
    static int pad = -1;
    if (pad < 0)
    {
        const char* e = getenv("EXL3_HIP_MLP1_PAD_LDS");          // diagnostics: 0 = let the dispatcher co-locate workgroups
        pad = e ? atoi(e) : M1_PAD_LDS_BYTES;
        if (pad > 0) EXL3_CHECK_HIP(hipFuncSetAttribute((const void*) exl3_mlp1_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, pad), "hipFuncSetAttribute");
    }
    exl3_mlp1_kernel<4><<<dim3(M1_TEAM, (unsigned) nteam), dim3(64 * M1_WAVES), (size_t) pad, st>>>(a);
    return exl3_check_launch("exl3_mlp1_fx");
}

// the sticky error word of exl3_mlp1_fx (0 = every bounded poll completed); reading clears it.  Synchronises the stream.
extern "C" int exl3_mlp1_error(int* out, void* stream)
{
    EXL3_CHECK_ARG(out, "exl3_mlp1_error: null pointer");
    hipStream_t st = (hipStream_t) stream;
    Exl3DevCtx* ctx = exl3_get_ctx(st);
    if (!ctx) return EXL3_ERR_INIT;
    uint32_t v = 0;
    EXL3_CHECK_HIP(hipMemcpyAsync(&v, ctx->tickets + M1_CNT_OFFSET + 32 * 31, 4, hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
    EXL3_CHECK_HIP(hipStreamSynchronize(st), "hipStreamSynchronize");
    if (v) EXL3_CHECK_HIP(hipMemsetAsync(ctx->tickets + M1_CNT_OFFSET + 32 * 31, 0, 4, st), "hipMemsetAsync");
    *out = (int) v;
    return EXL3_OK;
}
