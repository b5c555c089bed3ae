// Persistent decode-step kernel body (generation 5), included by exl3_pstep.hip.  Design: exl3_pstep.cuh.  K = bits per weight, mul1 codebook, one row.
//
// One 16-wave workgroup per CU, two roles, NO workgroup barrier inside the op loop (the waves meet through monotonic LDS counters):
//   * waves 0..11 STREAM: for every op they own a run of work units (2 tile rows x 128 columns) of the workgroup's rectangle.  While the op's activations
//     are not ready they DECODE AHEAD: the first unit into registers, the second into LDS (the weights do not depend on the activations); when the
//     service waves publish the activation quads they spend one MFMA pass per decoded unit and stream the rest (generation 4's unit, exl3_gemv4.kspec.hip
//     g4_unit); the last unit's ring refill already requests the wave's first rows of the NEXT op; partial rows -> LDS, counter, on to the next op.
//   * waves 12..15 SERVE (one per SIMD; their memory queues hold no weight rows, so their small dependent loads are not parked behind HBM streams --
//     vmcnt returns in order): poll the edge, read R / the producer's tagged slab lines, RMSNorm | q finish + RoPE | silu * mul, input Hadamard -> LDS quads;
//     when the streamers are done: sum the partial rows, mul1 affine map, then slab line (tagged granules, no edge) | output Hadamard + atomics into R
//     (drain, arrive at the edge) | final fp16 row; the K / V append of the new token as a side job.
// Cross-workgroup protocol: R is read only after the edge of the op that added into it; an op that adds into R first passes the READ GATE of the
// op that last read it (every workgroup arrives at the gate once its R values are in registers); slab lines carry their own tags.
#pragma once
#include "exl3_pstep.cuh"
#include "exl3_api_internal.h"
#include "exl3_gemv_args.h"
#include "exl3_lane_decode.cuh"
#include "exl3_glue_device.cuh"
#include "exl3_attn_device.cuh"
#include <type_traits>

template <int I, int N, typename F>
__device__ __forceinline__ void ps_static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); ps_static_for<I + 1, N>(f); }
}

// The plan tables are immutable during the launch: reading them through the CONSTANT address space makes every field a scalar load (s_load) with a
// wave-uniform result -- through a plain pointer the compiler must assume the kernel's own stores alias them and emits vector loads + vmcnt(0) waits +
// readfirstlane (and waterfall loops around every buffer instruction whose resource comes from such a value).  Pointers taken from the tables are
// generic to the compiler: weight / scale loads cast them to the GLOBAL address space (flat_load would also count against lgkmcnt and couple the
// weight stream to every LDS wait).
#define PS_CONST __attribute__((address_space(4)))
#define PS_GLOBAL __attribute__((address_space(1)))
typedef const PsOp PS_CONST* ps_op_p;
typedef const PsMat PS_CONST* ps_mat_p;
template <class T> __device__ __forceinline__ const T PS_GLOBAL* ps_g(const T* p) { return (const T PS_GLOBAL*) p; }
__device__ __forceinline__ PsTile ps_load_tile(const PsTile* tiles, size_t idx)
{
    const int PS_CONST* t = (const int PS_CONST*) (tiles + idx);
    PsTile r; r.mat = t[0]; r.cb0 = t[1]; r.ncb = t[2]; r.b0 = t[3]; r.nb = t[4]; r.slice = t[5]; r.side = t[6]; r.flags = t[7]; r.ubase = t[8];
    r.r0_ = t[9]; r.r1_ = t[10]; r.r2_ = t[11];
    return r;
}
template <int K>
__device__ __forceinline__ void ps_load_row(LaneWords<K>& d, const uint32_t* __restrict__ p)
{
    // p = this lane's first word of the tile row; K consecutive words (contiguous across the wave), non-temporal, global address space
    if constexpr (K == 4) { const uint4_t v = __builtin_nontemporal_load((const uint4_t PS_GLOBAL*) p); d.w[0] = v.x; d.w[1] = v.y; d.w[2] = v.z; d.w[3] = v.w; }
    else if constexpr (K == 8)
    {
        const uint4_t v = __builtin_nontemporal_load((const uint4_t PS_GLOBAL*) p), u = __builtin_nontemporal_load((const uint4_t PS_GLOBAL*) p + 1);
        d.w[0] = v.x; d.w[1] = v.y; d.w[2] = v.z; d.w[3] = v.w; d.w[4] = u.x; d.w[5] = u.y; d.w[6] = u.z; d.w[7] = u.w;
    }
    else if constexpr (K == 2) { const uint2_t v = __builtin_nontemporal_load((const uint2_t PS_GLOBAL*) p); d.w[0] = v.x; d.w[1] = v.y; }
    else
    {
        #pragma unroll
        for (int i = 0; i < K; ++i) d.w[i] = __builtin_nontemporal_load((const uint32_t PS_GLOBAL*) p + i);
    }
}

// agent-scope (sc1) accesses: everything one workgroup writes for another inside the launch goes through these (exl3_gemv2_tail.cuh has the same pair)
__device__ __forceinline__ unsigned long long ps_ld64(const void* p) { return __hip_atomic_load((unsigned long long*) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte agent-scope loads / stores: raw buffer instructions with the sc1 bit (aux = 16 on gfx950); the compiler tracks their wait counts
typedef __amdgpu_buffer_rsrc_t ps_rsrc_t;
__device__ __forceinline__ ps_rsrc_t ps_rsrc(const void* base) { return __builtin_amdgcn_make_buffer_rsrc((void*) base, 0, 0x7ffffff0, 0x00020000); }
// (PS_SCOPE_AUX: 16 = sc1, agent scope; 17 = sc0 | sc1, system scope -- what lines pushed into a PEER GPU's buffer and read from a buffer peers push into need)
#ifndef PS_SCOPE_AUX
#define PS_SCOPE_AUX 16
#endif
__device__ __forceinline__ uint4_t ps_ld128(ps_rsrc_t r, uint32_t byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int) byte_off, 0, PS_SCOPE_AUX); }
__device__ __forceinline__ void ps_st128(ps_rsrc_t r, uint32_t byte_off, uint4_t v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int) byte_off, 0, PS_SCOPE_AUX); }
// ... and the system-scope pair (sc0 | sc1) for the lines of a tensor-parallel plan: a partial row is pushed into EVERY rank's exchange buffer (its own included) over the
// IPC-mapped addresses, and the consumer reads a buffer that peer GPUs write while its kernel runs (fine-grained memory: exl3_pstep.hip)
__device__ __forceinline__ uint4_t ps_ld128_sys(ps_rsrc_t r, uint32_t byte_off) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int) byte_off, 0, 17); }
__device__ __forceinline__ void ps_st128_sys(ps_rsrc_t r, uint32_t byte_off, uint4_t v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int) byte_off, 0, 17); }

// Slab lines are TAGGED: a line = 128 values = 64 granule pairs of 16 bytes { v0, tag, v1, tag }, each written by ONE 16-byte agent-scope store; lane l of the
// writing half-wave owns values 4l..4l+3 = pair l ({4l, 4l+1}) and pair 32 + l ({4l+2, 4l+3}), so a store / load instruction covers 512 contiguous bytes.
// tag = (run epoch, producer op).  The consumer needs no edge: it loads the lines of its block and re-loads until every tag is the producer's (data and
// flag in one store: the guide's handoff-1to1 granules).  Lines summed in slice order from zero (slab_sum of exl3_glue_device.cuh).
#define PS_LINE_BYTES 1024
// PARTIAL lines -- a slice's partial row of a column block on its way to the consumer that sums the slices (slab lines of q|k|v and gate|up, the partial rows of o / down) --
// travel as fp16 pairs: ONE granule { v0 v1, tag, v2 v3, tag } per lane, 512 bytes per line.  Half the bytes of every hand-off edge and half the registers of a gather round
// (a q|k|v / gate|up op now gathers the partial lines of all its <= 4 blocks in ONE round trip, o_proj sums its 8 slab lines in one).  The sums stay fp32; what is rounded is
// each slice's contribution (2^-11 relative, like the activations the consumer forms from it).  The residual row's blocks (rbuf) stay fp32 lines.
// A/B builds: PS_FP32_PARTIALS keeps the 1 KiB fp32 lines.
#ifndef PS_FP32_PARTIALS
#define PS_PLINE_BYTES 512
struct PsPl { uint4_t a; };
__device__ __forceinline__ PsPl ps_pl_load(ps_rsrc_t r, uint32_t line_off, int l) { PsPl p; p.a = ps_ld128(r, line_off + (uint32_t) l * 16); return p; }
// (the scope is a COMPILE-TIME choice at every gather site: a per-load select between the two instructions cost the single-rank step 3.6-4.2 % -- the gathers of a
//  tensor-parallel plan are instantiated a second time instead and picked by one wave-uniform branch)
template <bool SYS>
__device__ __forceinline__ PsPl ps_pl_load_t(ps_rsrc_t r, uint32_t line_off, int l)
{
    PsPl p;
    if constexpr (SYS) p.a = ps_ld128_sys(r, line_off + (uint32_t) l * 16); else p.a = ps_ld128(r, line_off + (uint32_t) l * 16);
    return p;
}
__device__ __forceinline__ bool ps_pl_ok(const PsPl& p, uint32_t tag) { return (p.a.y == tag) & (p.a.w == tag); }
__device__ __forceinline__ float4_t ps_pl_val(const PsPl& p, uint32_t mask = 0xffffffffu)
{
    const half2_t a01 = u32_as_half2(p.a.x & mask), a23 = u32_as_half2(p.a.z & mask);
    return float4_t{ (float) a01.x, (float) a01.y, (float) a23.x, (float) a23.y };
}
__device__ __forceinline__ void ps_pl_store(ps_rsrc_t r, uint32_t line_off, int l, float4_t v, uint32_t tag)
{
    ps_st128(r, line_off + (uint32_t) l * 16, uint4_t{ half2_as_u32(half2_t{ f2h(v.x), f2h(v.y) }), tag, half2_as_u32(half2_t{ f2h(v.z), f2h(v.w) }), tag });
}
__device__ __forceinline__ void ps_pl_store_sys(ps_rsrc_t r, uint32_t line_off, int l, float4_t v, uint32_t tag)
{
    ps_st128_sys(r, line_off + (uint32_t) l * 16, uint4_t{ half2_as_u32(half2_t{ f2h(v.x), f2h(v.y) }), tag, half2_as_u32(half2_t{ f2h(v.z), f2h(v.w) }), tag });
}
#else
#define PS_PLINE_BYTES 1024
struct PsPl { uint4_t a, b; };
__device__ __forceinline__ PsPl ps_pl_load(ps_rsrc_t r, uint32_t line_off, int l) { PsPl p; p.a = ps_ld128(r, line_off + (uint32_t) l * 16); p.b = ps_ld128(r, line_off + 512 + (uint32_t) l * 16); return p; }
__device__ __forceinline__ bool ps_pl_ok(const PsPl& p, uint32_t tag) { return (p.a.y == tag) & (p.a.w == tag) & (p.b.y == tag) & (p.b.w == tag); }
__device__ __forceinline__ float4_t ps_pl_val(const PsPl& p, uint32_t mask = 0xffffffffu)
{
    return float4_t{ __uint_as_float(p.a.x & mask), __uint_as_float(p.a.z & mask), __uint_as_float(p.b.x & mask), __uint_as_float(p.b.z & mask) };
}
__device__ __forceinline__ void ps_pl_store(ps_rsrc_t r, uint32_t line_off, int l, float4_t v, uint32_t tag)
{
    ps_st128(r, line_off + (uint32_t) l * 16, uint4_t{ __float_as_uint(v.x), tag, __float_as_uint(v.y), tag });
    ps_st128(r, line_off + 512 + (uint32_t) l * 16, uint4_t{ __float_as_uint(v.z), tag, __float_as_uint(v.w), tag });
}
#endif

template <int NB, bool SYS = false>
__device__ __forceinline__ float4_t ps_slab_sum(ps_rsrc_t r, uint32_t blk_off, int S, int l, uint32_t tag, bool& ok)
{
    float4_t v = { 0.f, 0.f, 0.f, 0.f };
    for (int s = 0; s < S; s += NB)
    {
        PsPl t[NB];
        #pragma unroll
        for (int i = 0; i < NB; ++i) t[i] = ps_pl_load_t<SYS>(r, blk_off + (uint32_t) min(s + i, S - 1) * PS_PLINE_BYTES, l);
        #pragma unroll
        for (int i = 0; i < NB; ++i) if (s + i < S)
        {
            ok &= ps_pl_ok(t[i], tag);                                  // (bitwise: && makes an exec-mask branch per line)
            const float4_t x = ps_pl_val(t[i]);
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
        }
    }
    return v;
}

// two slab sets of one block at once (gate & up): both sets' lines of a round are in flight before the first add
template <int NB>
__device__ __forceinline__ void ps_slab_sum2(ps_rsrc_t ra, ps_rsrc_t rb, uint32_t blk_off, int S, int l, uint32_t tag, bool& ok, float4_t& va, float4_t& vb)
{
    va = float4_t{ 0.f, 0.f, 0.f, 0.f }; vb = va;
    for (int s = 0; s < S; s += NB)
    {
        PsPl ta[NB], tb[NB];
        #pragma unroll
        for (int i = 0; i < NB; ++i)
        {
            const uint32_t o = blk_off + (uint32_t) min(s + i, S - 1) * PS_PLINE_BYTES;
            ta[i] = ps_pl_load(ra, o, l); tb[i] = ps_pl_load(rb, o, l);
        }
        #pragma unroll
        for (int i = 0; i < NB; ++i) if (s + i < S)
        {
            ok &= ps_pl_ok(ta[i], tag) & ps_pl_ok(tb[i], tag);
            const float4_t x = ps_pl_val(ta[i]), y = ps_pl_val(tb[i]);
            va.x += x.x; va.y += x.y; va.z += x.z; va.w += x.w;
            vb.x += y.x; vb.y += y.y; vb.z += y.z; vb.w += y.w;
        }
    }
}

// one work unit = 2 tile rows of the wave's 128-column block against the activation group `ag` (exl3_gemv4.kspec.hip g4_unit, mul1 FAST variant)
// CB: the codebook of the op's tensors.  mul1: the FAST operand (fp16 of 1024 + byte sum; the affine map is applied once per output by the service waves).  3INST / mcg:
//   * units decoded AHEAD hold the EXACT fp16 weight (lo + hi halves in one fp16 add: the reference's bits, quant/codebook.cuh:56-77) -- ONE operand per quad like mul1, so a
//     decoded unit takes the same 32 registers / 8 KiB and the same MFMA pass; the three extra slow VALU instructions per weight pair are spent while the wave waits for
//     its input anyway;
//   * STREAMED units take generation 4's FAST form: the masked product's two fp16 halves go to the matrix pipe as two k-slots against duplicated activations
//     ((a0 a0 a1 a1), (a2 a2 a3 a3)) -- two more MFMAs per quad, no VALU instruction for the sum (exact streaming measured no faster than the launch-per-op step:
//     586.7 vs 582.8 tok/s at Llama-3.1-8B).  The number of units decoded ahead is then FIXED by the plan (not by when the input arrives), so a step's bits do not depend
//     on timing.
#define PS_VAR(CB) ((CB) == EXL3_CB_MUL1 ? 1 : 0)
template <int K, int CB, int HALF>
__device__ __forceinline__ void ps_unit(LaneWords<K> (&ring)[2], const uint32_t* __restrict__ refill, size_t refill_rs, int lane, int lofs, half4_t ag,
                                        float4_t& acc_c, float4_t& acc_d)
{
    // (3INST / mcg: the activation quad with every value twice: the k-slots of a weight's lo and hi halves)
    half4_t ag_lo = ag, ag_hi = ag;
    if constexpr (CB != EXL3_CB_MUL1)
    {
        union { half4_t h; uint32_t u[2]; } c; c.h = ag;
        ag_lo = u2_as_half4(__builtin_amdgcn_perm(c.u[0], c.u[0], 0x01000100u), __builtin_amdgcn_perm(c.u[0], c.u[0], 0x03020302u));
        ag_hi = u2_as_half4(__builtin_amdgcn_perm(c.u[1], c.u[1], 0x01000100u), __builtin_amdgcn_perm(c.u[1], c.u[1], 0x03020302u));
    }
    ps_static_for<0, 2>([&] (auto uc)
    {
        constexpr int u = decltype(uc)::value;
        uint32_t Wx[K + 1];
        #pragma unroll
        for (int i = 0; i < K; ++i) Wx[i + 1] = ring[u].w[i];
        {
            const uint32_t wl = ring[u].w[K - 1];
            const uint32_t r1 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x121, 0xf, 0xf, true);
            const uint32_t r9 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x129, 0xf, 0xf, true);
            Wx[0] = (lane & 7) ? r1 : r9;
        }
        if (refill) ps_load_row<K>(ring[u], refill + (size_t) u * refill_rs + lofs);      // (null: nothing to request)
        ps_static_for<0, 4>([&] (auto qc)
        {
            constexpr int q = decltype(qc)::value;
            constexpr int ABID = 8 * HALF + 4 * u + q;
            half4_t bc[2], bd[2];
#ifdef PS_ABL_NODECODE
            // speed-only ablation (results are garbage): what does the streaming phase cost without the decode arithmetic?
            bc[0] = u2_as_half4(Wx[q % (K + 1)], Wx[(q + 1) % (K + 1)]); bd[0] = u2_as_half4(Wx[(q + 1) % (K + 1)], Wx[q % (K + 1)]);
#else
            decode_quad<K, CB, 1, 8 * q>(Wx, bc);
            decode_quad<K, CB, 1, 8 * q + 4>(Wx, bd);
#endif
            if constexpr (CB == EXL3_CB_MUL1)
            {
                acc_c = __builtin_amdgcn_mfma_f32_4x4x4f16(ag, bc[0], acc_c, 4, ABID, 0);
                acc_d = __builtin_amdgcn_mfma_f32_4x4x4f16(ag, bd[0], acc_d, 4, ABID, 0);
            }
            else
            {
                // bc[0] = { lo0 hi0 lo1 hi1 }, bc[1] = { lo2 hi2 lo3 hi3 } of the quad's four weights: two k-slots per weight
                acc_c = __builtin_amdgcn_mfma_f32_4x4x4f16(ag_lo, bc[0], acc_c, 4, ABID, 0);
                acc_d = __builtin_amdgcn_mfma_f32_4x4x4f16(ag_lo, bd[0], acc_d, 4, ABID, 0);
                acc_c = __builtin_amdgcn_mfma_f32_4x4x4f16(ag_hi, bc[1], acc_c, 4, ABID, 0);
                acc_d = __builtin_amdgcn_mfma_f32_4x4x4f16(ag_hi, bd[1], acc_d, 4, ABID, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    });
}

// decode-ahead: the same unit decoded into 16 B operands (no activations needed), the ring refilled as a streamed unit refills it
template <int K, int CB>
__device__ __forceinline__ void ps_predecode(LaneWords<K> (&ring)[2], const uint32_t* __restrict__ refill, size_t refill_rs, int lane, int lofs, half4_t (&dec)[16])
{
    ps_static_for<0, 2>([&] (auto uc)
    {
        constexpr int u = decltype(uc)::value;
        uint32_t Wx[K + 1];
        #pragma unroll
        for (int i = 0; i < K; ++i) Wx[i + 1] = ring[u].w[i];
        {
            const uint32_t wl = ring[u].w[K - 1];
            const uint32_t r1 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x121, 0xf, 0xf, true);
            const uint32_t r9 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x129, 0xf, 0xf, true);
            Wx[0] = (lane & 7) ? r1 : r9;
        }
        if (refill) ps_load_row<K>(ring[u], refill + (size_t) u * refill_rs + lofs);
        ps_static_for<0, 4>([&] (auto qc)
        {
            constexpr int q = decltype(qc)::value;
            half4_t bc[2], bd[2];
            decode_quad<K, CB, PS_VAR(CB), 8 * q>(Wx, bc);
            decode_quad<K, CB, PS_VAR(CB), 8 * q + 4>(Wx, bd);
            dec[(4 * u + q) * 2] = bc[0]; dec[(4 * u + q) * 2 + 1] = bd[0];
        });
    });
}

// ... the same unit decoded straight into the wave's LDS slots (the second decode-ahead unit): no 32-register staging array
template <int K, int CB>
__device__ __forceinline__ void ps_predecode_lds(LaneWords<K> (&ring)[2], const uint32_t* __restrict__ refill, size_t refill_rs, int lane, int lofs, char* pdec_w)
{
    ps_static_for<0, 2>([&] (auto uc)
    {
        constexpr int u = decltype(uc)::value;
        uint32_t Wx[K + 1];
        #pragma unroll
        for (int i = 0; i < K; ++i) Wx[i + 1] = ring[u].w[i];
        {
            const uint32_t wl = ring[u].w[K - 1];
            const uint32_t r1 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x121, 0xf, 0xf, true);
            const uint32_t r9 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x129, 0xf, 0xf, true);
            Wx[0] = (lane & 7) ? r1 : r9;
        }
        if (refill) ps_load_row<K>(ring[u], refill + (size_t) u * refill_rs + lofs);
        ps_static_for<0, 4>([&] (auto qc)
        {
            constexpr int q = decltype(qc)::value;
            half4_t bc[2], bd[2];
            decode_quad<K, CB, PS_VAR(CB), 8 * q>(Wx, bc);
            decode_quad<K, CB, PS_VAR(CB), 8 * q + 4>(Wx, bd);
            *((half4_t*) (pdec_w + ((4 * u + q) * 2) * 512)) = bc[0];
            *((half4_t*) (pdec_w + ((4 * u + q) * 2 + 1) * 512)) = bd[0];
        });
    });
}

template <int HALF>
__device__ __forceinline__ void ps_consume(const half4_t (&dec)[16], half4_t ag, float4_t& acc_c, float4_t& acc_d)
{
    ps_static_for<0, 8>([&] (auto ic)
    {
        constexpr int i = decltype(ic)::value;                   // i = 4 u + q
        constexpr int ABID = 8 * HALF + i;
        acc_c = __builtin_amdgcn_mfma_f32_4x4x4f16(ag, dec[2 * i], acc_c, 4, ABID, 0);
        acc_d = __builtin_amdgcn_mfma_f32_4x4x4f16(ag, dec[2 * i + 1], acc_d, 4, ABID, 0);
    });
}

// lines per round trip of a slab sum: the registers of 4 fp32 lines hold 8 fp16 lines
#ifndef PS_FP32_PARTIALS
#define PS_SLAB_NB 8
#define PS_SLAB2_NB 4
#else
#define PS_SLAB_NB 4
#define PS_SLAB2_NB 3
#endif
// PS_SUM_HALFWAVES (A/B builds): the partial rows of a rectangle summed one column per service HALF-wave (the form before the end of round 5) instead of one per service wave
#ifndef PS_COOP_MIN
#define PS_COOP_MIN 8                  // a direct row edge gathers cooperatively (all eight service half-waves, through LDS) above this many producer slices
#endif
#ifndef PS_POLL_SLEEP
#define PS_POLL_SLEEP 2                // s_sleep between two polls of tagged lines (x 64 clocks)
#endif
#define PS_SW 12                       // streaming waves (3 per SIMD); waves PS_SW .. 15 serve
#define PS_NSV (PS_WAVES - PS_SW)      // service waves
// monotonic LDS counters (targets are derived by every wave from the uniform op / tile tables)
#define PS_C_EDGE 0                    // = op + 1 once the edge in front of op's R read is satisfied (service wave 0 polls)
#define PS_C_A 1                       // + PS_NSV per RMSNorm op: the service waves' block sums of squares are in LDS
#define PS_C_T 2                       // + PS_NSV per op: the service waves' activation quads of the op are in LDS
#define PS_C_S 3                       // + PS_SW per op: the streaming waves' partial rows of the op are in LDS
#define PS_C_R 4                       // + PS_NSV per op that adds into R: the service waves' atomics are acknowledged
#define PS_C_G 5                       // = op + 1 once the read gate in front of the row lines op's owners overwrite is satisfied
#define PS_C_X 7                       // + 8 per barrier of the attention item's eight waves
#define PS_C_Q 8                       // = op + 1 once the attention item's rotated queries (and the new token's words) are in LDS
#define PS_C_O 6                       // + PS_NSV per op whose blocks this workgroup owns: the service half-waves' gathered sums are in LDS
#define PS_C_ABORT 9                   // != 0 once a bounded wait of this workgroup timed out (the service waves then give up every later wait after one poll)

// Issue priority of the streaming waves.  A SIMD arbitrates its ready waves oldest-first: of the three streaming waves it hosts (w, w + 4, w + 8) the oldest streams at its
// own latency-bound pace and the youngest gets what is left -- by the phase stamps of an 8B gate|up op the waves 0-3 were done 6.3 us after the quads, the waves 8-11 after
// 12.8 us, with the SIMD half idle behind a single wave for the last third.  PS_PRIO_ROT: every wave walks its priority 0 -> 1 -> 2 -> 0 per work unit (offset by its
// group): over a run every wave spends the same number of units at every level.  PS_PRIO_INV (diagnostic): static 0 / 1 / 2 for the groups, youngest highest.
__device__ __forceinline__ void ps_prio_rot(int n)
{
#ifdef PS_PRIO_ROT
    const int r = n % 3;
    if (r == 0) __builtin_amdgcn_s_setprio(0); else if (r == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
#endif
}

// The run of work units [u0, u1) streaming wave w takes of a rectangle of T units.  UNIFORM (the tile's r0_ = r1_ = r2_ = 0): [T w / 12, T (w + 1) / 12).  WEIGHTED (round 6):
// a SIMD arbitrates its ready waves oldest-first, so of the three streaming waves it hosts (w, w + 4, w + 8) the oldest streams at its own memory-latency-bound pace
// (~ 0.95 us per unit by the per-unit stamps of an 8B gate|up op), the second nearly so (1.1) and the youngest gets the issue slots that are left (2.4-2.9) -- with equal
// runs the two older waves were done 7.4 / 8.5 us after the activations arrived and the youngest then streamed its last third ALONE, latency-bound, until 13.6 us.  The
// planner therefore cuts the rectangle by AGE GROUP: waves 0-3 / 4-7 / 8-11 take n_A >= n_B >= n_C units each (r0_ / r1_ / r2_ = n_g << 2 | e_g: the first e_g waves of
// the group take one more), chosen so that the three finish together (exl3_pstep.hip: wave_partition).  Everything here is wave-uniform (scalar arithmetic).
struct PsRange { int u0, u1; };
__device__ __forceinline__ PsRange ps_wave_range(const PsTile& t, int T, int w)
{
    PsRange r;
    if ((t.r0_ | t.r1_ | t.r2_) == 0) { r.u0 = (T * w) / PS_SW; r.u1 = (T * (w + 1)) / PS_SW; return r; }
    const int g = w >> 2, i = w & 3;
    const int GA = 4 * (t.r0_ >> 2) + (t.r0_ & 3), GB = 4 * (t.r1_ >> 2) + (t.r1_ & 3);
    const int pg = g == 0 ? t.r0_ : (g == 1 ? t.r1_ : t.r2_), base = g == 0 ? 0 : (g == 1 ? GA : GA + GB);
    const int n = pg >> 2, e = pg & 3;
    r.u0 = base + i * n + min(i, e); r.u1 = r.u0 + n + (i < e ? 1 : 0);
    return r;
}

// a streaming wave's run of work units inside its workgroup's rectangle: column-major over (column block j, unit i); at most two column blocks (planner: ncb <= PS_SW)
template <int K>
struct PsSeg
{
    const uint32_t* stripA; const uint32_t* stripB; size_t rs;      // wave-uniform pointers (lane 0's words) of the first unit of segment 0 / 1; words per tile row
    int j0, i0, len0, len1, n;
    __device__ __forceinline__ const uint32_t* unit_ptr(int q) const       // 0 <= q < n
    {
#ifdef PS_ABL_HOT
        return stripA;
#endif
        return q < len0 ? stripA + (size_t) (2 * q) * rs : stripB + (size_t) (2 * (q - len0)) * rs;
    }
};

template <int K>
__device__ __forceinline__ PsSeg<K> ps_make_seg(ps_op_p O, const PsTile& t, int wave)
{
    constexpr int NW = 8 * K;
    PsSeg<K> s;
    s.stripA = nullptr; s.stripB = nullptr; s.rs = 0; s.j0 = 0; s.i0 = 0; s.len0 = 0; s.len1 = 0; s.n = 0;
    if (t.mat >= 0)
    {
        const int H = 4 * t.nb, T = H * t.ncb;
#ifdef PS_TEST_SW8
        // experiment: only 8 of the 12 streaming waves take units (does the streaming phase scale with the number of streaming waves?)
        const int u0 = wave < 8 ? (T * wave) / 8 : T, u1 = wave < 8 ? (T * (wave + 1)) / 8 : T;
#else
        const PsRange rg = ps_wave_range(t, T, wave);
        const int u0 = rg.u0, u1 = rg.u1;
#endif
        s.n = u1 - u0;
        s.j0 = u0 / H; s.i0 = u0 - s.j0 * H;
        s.len0 = min(s.n, H - s.i0); s.len1 = s.n - s.len0;
        const uint32_t* const Bp = O->Bp;
        if (Bp)
        {
            // REPACKED weights (exl3_pstep.cuh: PsOp::Bp): the rectangle's units lie in the order the waves take them, a unit = 2 tile rows of 128 columns = 16 NW
            // contiguous words -> this wave's whole run (both segments) is ONE contiguous range
            s.rs = (size_t) 8 * NW;
            s.stripA = Bp + (size_t) (t.ubase + u0) * (16 * NW);
            s.stripB = s.stripA + (size_t) s.len0 * (16 * NW);
#ifdef PS_ABL_HOT
            // speed-only ablation (results are garbage): every unit of the rectangle reads the SAME 2 tile rows per wave -- the weight stream without HBM behind it
            s.stripA = Bp + (size_t) (t.ubase + wave) * (16 * NW); s.stripB = s.stripA;
#endif
        }
        else
        {
            const ps_mat_p M = &O->mat[t.mat];
            const uint32_t* B = M->B; const int tn = M->tiles_n;
            s.rs = (size_t) tn * NW;
            s.stripA = B + ((size_t) (t.b0 * 8 + 2 * s.i0) * tn + (size_t) (t.cb0 + s.j0) * 8) * NW;
            s.stripB = B + ((size_t) (t.b0 * 8) * tn + (size_t) (t.cb0 + s.j0 + 1) * 8) * NW;
        }
    }
    return s;
}

// K / K2: bits per weight of the layers' fused linears -- K2 = K, or K + 1 for a fractional-bpw checkpoint (the reference's allocator bumps whole qgroups by one bit:
// conversion/allocation.py:131-141); CB: the codebook of every tensor (mul1 | 3INST | mcg).
// ATT: the plan has the decode attention inside o_proj's preparation (PS_ATTN).  A separate instantiation: with the attention code compiled in, the step WITHOUT attention
// ran 2.6 % (8B) / 6 % (1B) slower (250 instead of 78 scalar spills on the service path, a third more code)
// KH: bits per weight of the lm_head (= K, or 6: the head of a real checkpoint)
// TP: the plan is one rank of a tensor-parallel job (the row shards' partial lines go to every rank's exchange buffer, the row edges read with system-scope loads).  Its own
// instantiations: compiled into the one kernel -- as run-time branches on the plan's rank count -- the tensor-parallel paths cost the SINGLE-rank step 3.5-4 % (1B 0.398 ->
// 0.414 ms, same box: a fifth more scalar spills on the service path), the same lesson as ATT.
template <int K, int K2, int KH, int CB, bool ATT, bool TP = false>
__global__ __launch_bounds__(PS_NT) void exl3_pstep_kernel(const PsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (the activation quads lie IN FRONT of the gather area: a slice of more than 32 Hadamard blocks -- only the lm_head of a model wider than 4096, one slice of the whole
    //  row -- runs up to 8 KiB into it; that op gathers nothing)
    char* const quads = smem + PS_MISC_BYTES + PS_PART_BYTES + PS_PDEC_BYTES;
    float* const ssblk = (float*) smem;
    float* const bsum2 = ssblk + 64;                                // [2 (op parity)][64]: block sums of the rotated activations
    int* const seginfo2 = (int*) (bsum2 + 128);                     // [2 (op parity)][16][4]: j0, len0, len1 of every streaming wave's run
    uint32_t* const lctl = (uint32_t*) (seginfo2 + 128);            // the PS_C_* counters
    float* const part = (float*) (smem + PS_MISC_BYTES);
    char* const pdec = smem + PS_MISC_BYTES + PS_PART_BYTES;
    float* const gath = (float*) (smem + PS_MISC_BYTES + PS_PART_BYTES + PS_PDEC_BYTES + PS_QUADS_BYTES);      // [4 owned blocks][8 half-waves][128]

    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cu = blockIdx.x, ncu = a.ncu, nops = a.nops;
    unsigned long long* const dbg = a.dbg;
    // phase stamps (100 MHz, PS_DBG_SLOTS per op and workgroup): lane 0 of streaming wave 0 writes slots 0..2, lane 0 of service wave 0 slots 3..12; slot 16 + w: streaming
    // wave w has finished its run, slot 28 + s: service wave s has published its quads
    // (tools/pstep_stamps.py names them)
    #define PS_T(i) do { if (dbg && lane == 0) dbg[((size_t) op * ncu + cu) * PS_DBG_SLOTS + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    // (slots 13 / 14: the SHADER clock counter (s_memtime) at stamps 1 / 2 of streaming wave 0 -- (14 - 13) / (2 - 1) = the core clock during the op's streaming phase)
    #define PS_TC(i) do { if (dbg && lane == 0) dbg[((size_t) op * ncu + cu) * PS_DBG_SLOTS + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

    if (tid < 16) lctl[tid] = 0u;
    const uint32_t epoch = (uint32_t) __builtin_amdgcn_readfirstlane((int) *a.epoch);      // bumped by workgroup 0 when it leaves: every replay tags afresh
    __syncthreads();

    auto c_load = [&] (int i) -> uint32_t { return __hip_atomic_load(lctl + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto c_wait = [&] (int i, uint32_t target) { while ((int32_t) (c_load(i) - target) < 0) __builtin_amdgcn_s_sleep(1); };
    auto c_spin = [&] (int i, uint32_t target) { while ((int32_t) (c_load(i) - target) < 0) { } };          // service waves: no sleep granularity on the latency chain
    auto c_inc = [&] (int i) -> uint32_t
    {
        uint32_t old = 0u;
        if (lane == 0) old = __hip_atomic_fetch_add(lctl + i, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (uint32_t) __builtin_amdgcn_readfirstlane((int) old);
    };
    auto c_set = [&] (int i, uint32_t v) { if (lane == 0) __hip_atomic_store(lctl + i, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); };


    // ---- the attention item's token loop, shared by the four service waves (slots 0..3) and -- when the workgroup has an item -- streaming waves 0..3 (slots 4..7): eight
    // waves, two per SIMD, take 16 tokens each per 128-token step (attn_decode_wide_kernel<..., NW = 8> of exl3_attn_decode.hip).  LDS: the V tiles of the eight waves start at
    // the gather area (free in an o_proj op) and run into the attention area; behind them the new token's words, the waves' statistics and the rotated queries.
    struct AttItem { int h, split, t0, t1, st_tok, len, nsteps, G; bool owner; const uint32_t* kc; const half_t* ks; const uint32_t* vc; const half_t* vs; };
    struct AttWords { uint4_t k, v; half_t ks, vs; };
    char* const att_base = smem + (PS_MISC_BYTES + PS_PART_BYTES + PS_PDEC_BYTES + PS_QUADS_BYTES);      // = the gather area
    half_t* const att_vt = (half_t*) att_base;                                  // [8 waves][16 * AW_VS]; after the loop: partial outputs [8][8][128] fp32 (32 KB <= 34 KB)
    uint32_t* const att_newkv = (uint32_t*) (att_base + 8 * 16 * AW_VS * 2);     // [2][16]: the new token's K / V words of this kv head
    half_t* const att_newsc = (half_t*) (att_base + 8 * 16 * AW_VS * 2 + 128);   // [2][4]: their group scales
    float* const att_ml = (float*) (att_base + 8 * 16 * AW_VS * 2 + 256);        // [8 waves][8 heads][2]
    half_t* const att_q = (half_t*) (att_base + 8 * 16 * AW_VS * 2 + 768);       // [8][128] rotated, pre-scaled queries in pair order
    static_assert(8 * 16 * AW_VS * 2 + 768 + 2048 <= PS_GATH_BYTES + PS_ATT_BYTES, "attention item: LDS map");
    auto att_nse = [] (int len, int ns) -> int { return min(max((len + 127) >> 7, 1), ns); };       // (head_dim 64: the plan's nsplit is <= 16: one lane per split in a 16-lane head)       // splits in use: enough for one 128-token step each, at most the plan's
    auto att_make = [&] (ps_op_p O, int item, int len, int ns, int nse, int hkv) -> AttItem
    {
        AttItem it;
        it.h = item / ns; it.split = item - it.h * ns; it.len = len;
        it.st_tok = (((len + nse - 1) / nse) + 15) & ~15;                       // tokens per split: a multiple of the 16 tokens a wave takes per step
        it.t0 = it.split * it.st_tok; it.t1 = min(len, it.t0 + it.st_tok);
        it.nsteps = (it.st_tok + 127) >> 7; it.G = (hkv * O->hd) >> 5;           // 32-groups per token
        it.owner = len - 1 >= it.t0 && len - 1 < it.t0 + it.st_tok;             // the split that holds the new token finishes and appends its K / V
        it.kc = O->k_cache; it.ks = O->k_scales; it.vc = O->v_cache; it.vs = O->v_scales;
        return it;
    };
    auto att_load = [&] (const AttItem& it, int slot, int lane_v, int st_) -> AttWords
    {
        AttWords r;
        const int c = lane_v & 15, kg = lane_v >> 4;
        const int tb = it.t0 + 128 * st_ + 16 * slot;                          // the wave's 16 tokens: 16-aligned, inside ONE page (the page size is a multiple of 16)
        const int tk = max(min(tb + c, it.t1 - 1), 0);
        const int page = a.page_size;
        // the page id is wave-uniform: a scalar load (the table is not written during the launch) instead of a vector load in front of the cache loads' addresses
        const int tpg = max(tb < it.t1 ? tb : it.t1 - 1, 0) / page;
        const int64_t pg = (int64_t) ((const int PS_CONST*) a.block_table)[min(tpg, a.blocks_per_seq - 1)];
        const int64_t gbase = (pg * page + (tk % page)) * it.G + it.h * 4 + kg;
        r.k = *ps_g((const uint4_t*) (it.kc + gbase * 4)); r.v = *ps_g((const uint4_t*) (it.vc + gbase * 4));
        r.ks = ps_g(it.ks)[gbase]; r.vs = ps_g(it.vs)[gbase];
        return r;
    };
    auto att_tokens = [&] (const AttItem& it, int slot, int lane_v, AttWords w0, uint32_t& tgt_x)
    {
        const int c = lane_v & 15, kg = lane_v >> 4;
        auto att_bar = [&] () { tgt_x += 8u; c_inc(PS_C_X); c_spin(PS_C_X, tgt_x); };
        // (the query fragments are re-read from LDS in every step: 16 registers the token loop does not have -- the persistent kernel's budget is 128)
        const half_t* const qrow = att_q + min(c, 7) * 128 + 32 * kg;          // rows >= gq are zero rows; lanes c >= 8 read row 7 and are masked below
        float m_run = -1.0e30f, l_run = 0.0f;                                   // of query head c (lanes with c >= gq carry zero queries)
        float4_t oc[8];
        #pragma unroll
        for (int nbk = 0; nbk < 8; ++nbk) oc[nbk] = float4_t{ 0.f, 0.f, 0.f, 0.f };
        half_t* const vw = att_vt + (size_t) slot * (16 * AW_VS);
        uint32_t mk_v = 0x001E001Eu;
        asm volatile("" : "+v"(mk_v));
        for (int st = 0; st < it.nsteps; ++st)
        {
            const int tb_ = it.t0 + 128 * st + 16 * slot;                       // the wave's 16 tokens of this step
            if (tb_ >= it.t1) break;                                            // wave-uniform
            const int tk = min(tb_ + c, it.t1 - 1);
            if (it.owner && tk == it.len - 1)
            {
                // the new token's words come from this workgroup's LDS copy (its cache row is being written by this very launch)
                w0.k = uint4_t{ att_newkv[kg * 4], att_newkv[kg * 4 + 1], att_newkv[kg * 4 + 2], att_newkv[kg * 4 + 3] };
                w0.v = uint4_t{ att_newkv[16 + kg * 4], att_newkv[16 + kg * 4 + 1], att_newkv[16 + kg * 4 + 2], att_newkv[16 + kg * 4 + 3] };
                w0.ks = att_newsc[kg]; w0.vs = att_newsc[4 + kg];
            }
            float4_t scv = { 0.f, 0.f, 0.f, 0.f };
            {
                const half_t k4 = w0.ks * (half_t) 4.0f;
                #pragma unroll
                for (int s = 0; s < 4; ++s)
                {
                    const half8_t ka = aw_dequant8(s == 0 ? w0.k.x : (s == 1 ? w0.k.y : (s == 2 ? w0.k.z : w0.k.w)), half2_t{ k4, k4 }, mk_v);
                    half8_t qv = *((const half8_t*) (qrow + 8 * s));
                    if (c >= 8) qv = half8_t{ 0, 0, 0, 0, 0, 0, 0, 0 };
                    scv = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qv, scv, 0, 0, 0);
                }
                const half_t v4 = w0.vs * (half_t) 4.0f;
                #pragma unroll
                for (int s = 0; s < 4; ++s)
                    *((half8_t*) (vw + c * AW_VS + 32 * kg + 8 * s)) = aw_dequant8(s == 0 ? w0.v.x : (s == 1 ? w0.v.y : (s == 2 ? w0.v.z : w0.v.w)), half2_t{ v4, v4 }, mk_v);
            }
            // the next step's words are requested HERE -- after this step's words are consumed (one set of them in registers: the budget is 128), ahead of the
            // softmax and the value product (two steps ahead measured slower here: 468 vs 476 tok/s at 16 000 tokens)
            if (st + 1 < it.nsteps) w0 = att_load(it, slot, lane_v, st + 1);
            float mx = m_run;
            #pragma unroll
            for (int r = 0; r < 4; ++r) { if (tb_ + 4 * kg + r >= it.t1) scv[r] = -1.0e30f; mx = fmaxf(mx, scv[r]); }
            mx = fmaxf(mx, xor_lane(mx, 16)); mx = fmaxf(mx, xor_lane(mx, 32));
            if (!(mx > m_run + 8.0f)) mx = m_run;                               // lazy reference (exl3_attn_decode.hip)
            const float corr = __builtin_amdgcn_exp2f(m_run - mx);
            float p[4], psum = 0.0f;
            #pragma unroll
            for (int r = 0; r < 4; ++r) { p[r] = scv[r] > -1.0e29f ? __builtin_amdgcn_exp2f(scv[r] - mx) : 0.0f; psum += p[r]; }
            psum += xor_lane(psum, 16); psum += xor_lane(psum, 32);
            l_run = l_run * corr + psum; m_run = mx;
            const half4_t pa = { (half_t) p[0], (half_t) p[1], (half_t) p[2], (half_t) p[3] };
            if (__any(corr != 1.0f))
            {
                float cr[4];
                #pragma unroll
                for (int r = 0; r < 4; ++r) cr[r] = __shfl(corr, 4 * kg + r, 64);
                #pragma unroll
                for (int nbk = 0; nbk < 8; ++nbk) { oc[nbk].x *= cr[0]; oc[nbk].y *= cr[1]; oc[nbk].z *= cr[2]; oc[nbk].w *= cr[3]; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            #pragma unroll
            for (int nbk = 0; nbk < 8; ++nbk)
            {
                const half4_t vb = aw_tr16(vw + (4 * kg + (c >> 2)) * AW_VS + 16 * nbk + 4 * (c & 3));
                oc[nbk] = __builtin_amdgcn_mfma_f32_16x16x16f16(pa, vb, oc[nbk], 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- the eight waves' partial results (statistics of head c from lanes (c, kg = 0); outputs of heads 4 kg + r, column 16 nbk + c in pair order from every lane)
        att_bar();                                                             // the V tiles are dead: their space takes the partial outputs [slot][head][128] fp32
        float* const o_s = (float*) att_vt;
        if (kg == 0 && c < 8) { att_ml[(slot * 8 + c) * 2] = m_run; att_ml[(slot * 8 + c) * 2 + 1] = l_run; }
        #pragma unroll
        for (int nbk = 0; nbk < 8; ++nbk)
            #pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int head = 4 * kg + r;
                if (head < 8) o_s[(slot * 8 + head) * 128 + 16 * nbk + c] = oc[nbk][r];
            }
        att_bar();
    };

    if (wave < PS_SW)
    {
        // =========================================================================================== streaming waves
        // the op loop of a streaming wave over ops [op0, op1), for matrices of KK bits per weight: the layers' linears share K, the lm_head may have its own (KH: real
        // checkpoints keep the head at 6 bits) -- then the loop runs twice, the second time over the head alone (no rows requested across the two)
        uint32_t tgt_x = 0u;
        auto stream_ops = [&] (auto kc, auto pmc, const int op0, const int op1)
        {
        constexpr int KK = decltype(kc)::value;
        constexpr int PMC = decltype(pmc)::value;          // decode-ahead units this pass may hold (a head of its own K: two -- the third unit's registers on top of a 6-bit
                                                           // ring are what pushed the mixed instantiations into scratch, and one unit of ~ 26 per wave is nothing there)
        const int quad_lane = (lane >> 2) * 8, lofs = lane * KK;
        const int pmax = min(a.pmax, PMC);
#ifdef PS_PRIO_INV
        if ((wave >> 2) == 1) __builtin_amdgcn_s_setprio(1); else if ((wave >> 2) == 2) __builtin_amdgcn_s_setprio(2);
#endif
        LaneWords<KK> ring[2];
        #pragma unroll
        for (int i = 0; i < KK; ++i) { ring[0].w[i] = 0u; ring[1].w[i] = 0u; }
        char* const pdec_w = pdec + (size_t) wave * (16 * 64 * 8) + (size_t) lane * 8;

        const ps_op_p ops_c = (ps_op_p) a.ops;
        PsTile tl = ps_load_tile(a.tiles, (size_t) op0 * ncu + cu);
        PsSeg<KK> cur = ps_make_seg<KK>(ops_c + op0, tl, wave);
        bool ring_ready = false;
        for (int op = op0; op < op1; ++op)
        {
            const ps_op_p O = ops_c + op;
            if (wave == 0) PS_T(0);
            // the next op's rectangle and this wave's run in it (pointers only): the last streamed unit of this op requests its first rows
            PsTile tn; tn.mat = -1; tn.cb0 = 0; tn.ncb = 0; tn.b0 = 0; tn.nb = 0; tn.slice = 0; tn.side = -1; tn.flags = 0; tn.ubase = 0; tn.r0_ = 0; tn.r1_ = 0; tn.r2_ = 0;
            if (op + 1 < op1) tn = ps_load_tile(a.tiles, (size_t) (op + 1) * ncu + cu);
            const PsSeg<KK> nxt = ps_make_seg<KK>(O + 1, tn, wave);
            const uint32_t* const after_all = nxt.n > 0 ? nxt.stripA : nullptr;           // (null: the last unit does not refill)
            const size_t after_rs = nxt.rs;
            if (lane == 0) { int* si = seginfo2 + (op & 1) * 64 + wave * 4; si[0] = cur.j0; si[1] = cur.len0; si[2] = cur.len1; si[3] = 0; }

            if (ATT && wave < 4 && ((O->in_type & (0xff | PS_ATTN)) == (PS_IN_QKV | PS_ATTN)))
            {
                // this workgroup's attention item (the service waves' preparation of this op, below): streaming waves 0..3 are its slots 4..7
                const PsAtt PS_CONST* const AT = (const PsAtt PS_CONST*) &O->mat[1];
                const int ns = AT->nsplit;
                const int len = __builtin_amdgcn_readfirstlane(ps_g(a.seqlens)[0]);
                const int nse = att_nse(len, ns);
                if (tl.side >= 0 && tl.side % ns < nse)
                {
                    const AttItem it = att_make(O, tl.side, len, ns, nse, AT->hkv);
                    AttWords w0 = att_load(it, 4 + wave, lane, 0);
                    c_wait(PS_C_Q, (uint32_t) (op + 1));
                    att_tokens(it, 4 + wave, lane, w0, tgt_x);
                }
            }
            // (the decode-ahead units live inside the op: declared here they are dead during the attention item above)
            half4_t dec0[16], dec1[16];
            #pragma unroll
            for (int i = 0; i < 16; ++i) { dec0[i] = half4_t{ 0, 0, 0, 0 }; dec1[i] = dec0[i]; }
#ifdef PS_DBG_UNITS
            int dbg_u = 0;
            #define PS_TP() do { if (dbg && op == PS_DBG_UNITS && (wave & 3) == 0 && lane == 0 && dbg_u < PS_DBG_SLOTS) dbg[((size_t) (nops + (wave >> 2)) * ncu + cu) * PS_DBG_SLOTS + dbg_u++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
            #define PS_TP() do { } while (0)
#endif
            PS_TP();                                       // [0] op start
            // ---- decode-ahead while the service waves have not published the op's activation quads
            const uint32_t tgt_t = (uint32_t) PS_NSV * (uint32_t) (op + 1);
            int P = 0;
            if (cur.n > 0)
            {
                if (!ring_ready)
                {
                    ps_load_row<KK>(ring[0], cur.stripA + lofs);
                    ps_load_row<KK>(ring[1], cur.stripA + cur.rs + lofs);
                }
                const int Pm = min(pmax, cur.len0);
                // (3INST / mcg: units decoded ahead and streamed units differ in rounding, so HOW MANY are decoded ahead must not depend on when the input arrives)
                constexpr bool FIXED_P = CB != EXL3_CB_MUL1;
                if (Pm >= 1 && (FIXED_P || (int32_t) (c_load(PS_C_T) - tgt_t) < 0))
                {
                    PS_TP();                               // [1] first rows there? (the predecode waits for them)
                    ps_predecode<KK, CB>(ring, cur.unit_ptr(min(1, cur.n - 1)), cur.rs, lane, lofs, dec0);
                    P = 1;
                    PS_TP();
                    if (Pm >= 2 && (FIXED_P || (int32_t) (c_load(PS_C_T) - tgt_t) < 0))
                    {
                        // (the SECOND unit in registers too where the pass may hold three: the op's input then meets two register units -- all of a q|k|v wave's work at
                        //  Llama-3.1-8B -- and only the third comes back through 8 KiB of LDS.  PS_UNIT2_LDS: the order before round 6, second unit in LDS)
#ifndef PS_UNIT2_LDS
                        if constexpr (PMC >= 3) ps_predecode<KK, CB>(ring, cur.unit_ptr(min(2, cur.n - 1)), cur.rs, lane, lofs, dec1);
                        else
#endif
                        ps_predecode_lds<KK, CB>(ring, cur.unit_ptr(min(2, cur.n - 1)), cur.rs, lane, lofs, pdec_w);
                        P = 2;
                        PS_TP();
                        // a THIRD unit, in registers again (Llama-3.2-1B's gate|up rectangle is 32 units = 2.67 per wave: with two units ahead eight waves streamed one
                        // more after the quads were there -- 1.5 us of decode on the critical path of every layer)
                        if constexpr (PMC >= 3)
                        {
                            if (Pm >= 3 && (FIXED_P || (int32_t) (c_load(PS_C_T) - tgt_t) < 0))
                            {
#ifndef PS_UNIT2_LDS
                                ps_predecode_lds<KK, CB>(ring, cur.unit_ptr(min(3, cur.n - 1)), cur.rs, lane, lofs, pdec_w);
#else
                                ps_predecode<KK, CB>(ring, cur.unit_ptr(min(3, cur.n - 1)), cur.rs, lane, lofs, dec1);
#endif
                                P = 3;
                                PS_TP();
                            }
                        }
                    }
                }
            }
            c_wait(PS_C_T, tgt_t);
            PS_TP();                                       // quads seen
            if (wave == 0) { PS_T(1); PS_TC(13); }

            // ---- this wave's run of work units: decode-ahead units first (MFMA only), the rest streamed
            auto run_seg = [&] (const uint32_t* strip, int len, int pre, const uint32_t* after, size_t aft_rs, int qoff, float* pslot)
            {
                float4_t acc_c = { 0.f, 0.f, 0.f, 0.f }, acc_d = acc_c;
                const char* qb = quads + qoff + quad_lane;
                const size_t rs = cur.rs;
#ifdef PS_ABL_HOT
                auto up = [&] (int q) -> const uint32_t* { return q < len ? strip : after; };
#else
                auto up = [&] (int q) -> const uint32_t* { return q < len ? strip + (size_t) (2 * q) * rs : after; };
#endif
                auto ur = [&] (int q) -> size_t { return q < len ? rs : aft_rs; };          // the refill target may lie in the NEXT op's matrix (another row pitch)
                int p = 0;
#ifdef PS_DBG_UNITS
                // diagnostic build: streaming waves 0 / 4 / 8 (one SIMD's three) stamp the start of every work unit of op PS_DBG_UNITS into three extra stamp areas behind
                // the ops' (exl3_pstep.hip allocates them in this build): tools/pstep_unit_timeline.py
                #define PS_TU() do { if (dbg && op == PS_DBG_UNITS && (wave & 3) == 0 && lane == 0 && dbg_u < PS_DBG_SLOTS) dbg[((size_t) (nops + (wave >> 2)) * ncu + cu) * PS_DBG_SLOTS + dbg_u++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
                #define PS_TU() do { } while (0)
#endif
                if (pre > 0)
                {
                    const uint2_t raw = *((const uint2_t*) qb);
                    const half4_t ag = u2_as_half4(raw.x, raw.y);
                    PS_TU();
                    ps_consume<0>(dec0, ag, acc_c, acc_d);
                    if (len > 1)
                    {
                        if (pre > 1)
                        {
#ifndef PS_UNIT2_LDS
                            if constexpr (PMC >= 3) { PS_TU(); ps_consume<1>(dec1, ag, acc_c, acc_d); }
                            else
#endif
                            {
                                half4_t tmp[16];
                                #pragma unroll
                                for (int i = 0; i < 16; ++i) tmp[i] = *((const half4_t*) (pdec_w + i * 512));
                                PS_TU();
                                ps_consume<1>(tmp, ag, acc_c, acc_d);
                            }
                        }
                        else { PS_TU(); ps_unit<KK, CB, 1>(ring, up(2), ur(2), lane, lofs, ag, acc_c, acc_d); }
                    }
                    p = 2;
                    if constexpr (PMC >= 3)
                    {
                        if (pre > 2)                                     // (pre > 2 implies len > 2: decode-ahead stays inside segment 0)
                        {
                            const uint2_t raw2 = *((const uint2_t*) (qb + 2 * 64));
                            const half4_t ag2 = u2_as_half4(raw2.x, raw2.y);
                            PS_TU();
#ifndef PS_UNIT2_LDS
                            {
                                half4_t tmp[16];
                                #pragma unroll
                                for (int i = 0; i < 16; ++i) tmp[i] = *((const half4_t*) (pdec_w + i * 512));
                                ps_consume<0>(tmp, ag2, acc_c, acc_d);
                            }
#else
                            ps_consume<0>(dec1, ag2, acc_c, acc_d);
#endif
                            if (len > 3) { PS_TU(); ps_unit<KK, CB, 1>(ring, up(4), ur(4), lane, lofs, ag2, acc_c, acc_d); }
                            p = 4;
                        }
                    }
                }
                for (; p + 1 < len; p += 2)
                {
                    const uint2_t raw = *((const uint2_t*) (qb + p * 64));
                    const half4_t ag = u2_as_half4(raw.x, raw.y);
                    ps_prio_rot(p + (wave >> 2));
                    PS_TU();
                    ps_unit<KK, CB, 0>(ring, up(p + 1), ur(p + 1), lane, lofs, ag, acc_c, acc_d);
                    ps_prio_rot(p + 1 + (wave >> 2));
                    PS_TU();
                    ps_unit<KK, CB, 1>(ring, up(p + 2), ur(p + 2), lane, lofs, ag, acc_c, acc_d);
                }
                if (p < len)
                {
                    const uint2_t raw = *((const uint2_t*) (qb + p * 64));
                    const half4_t ag = u2_as_half4(raw.x, raw.y);
                    PS_TU();
                    ps_unit<KK, CB, 0>(ring, up(p + 1), ur(p + 1), lane, lofs, ag, acc_c, acc_d);
                }
                PS_TU();
                const int col = 16 * (lane >> 3) + (lane & 7);
                pslot[col] = acc_c[0]; pslot[col + 8] = acc_d[0];
                #undef PS_TU
            };
            if (cur.n > 0)
            {
                float* pw = part + (size_t) wave * 256;
                run_seg(cur.stripA, cur.len0, P, cur.len1 > 0 ? cur.stripB : after_all, cur.len1 > 0 ? cur.rs : after_rs, cur.i0 * 64, pw);
                if (cur.len1 > 0) run_seg(cur.stripB, cur.len1, 0, after_all, after_rs, 0, pw + 128);
            }
#ifndef PS_SUM_HALFWAVES
            else
            {
                // (a wave without a unit in this op leaves a zero row: the service wave that sums a one-column rectangle then adds all twelve rows unmasked)
                float* pw = part + (size_t) wave * 256;
                const int col = 16 * (lane >> 3) + (lane & 7);
                pw[col] = 0.0f; pw[col + 8] = 0.0f;
            }
#endif
            ring_ready = cur.n > P && nxt.n > 0;
            c_inc(PS_C_S);
            if (wave == 0) { PS_T(2); PS_TC(14); }
            PS_T(16 + wave);
            cur = nxt; tl = tn;
        }
        };
        // (8-bit layers: two units ahead as well -- a 16-word ring + three decoded units sat at the 128-register limit of a 16-wave workgroup, one change away from scratch)
        // the ops in RUNS of equal bits per weight (PsArgs::runs = { end op, K } pairs; a uniform model with the head at the layers' K: one run; no rows are requested across
        // a run boundary).  The lm_head's pass and 6-bit layer passes of a mixed plan hold two decode-ahead units (register budget: see above)
        // (3INST / mcg: two units ahead -- their exact decode costs 1.8x a mul1 unit and runs whether or not the input is already there: same box 638 vs 631 tok/s)
        constexpr int PM_MAIN = (K >= 8 || CB != EXL3_CB_MUL1) ? 2 : 3;
        constexpr int PM_2 = (K2 >= 6 || CB != EXL3_CB_MUL1) ? 2 : 3;
        const int PS_CONST* const runs = (const int PS_CONST*) a.runs;
        int op0 = 0;
        for (int r = 0; op0 < nops; ++r)
        {
            const int op1 = runs[2 * r], Kr = runs[2 * r + 1];
            if (Kr == K) stream_ops(std::integral_constant<int, K>{}, std::integral_constant<int, PM_MAIN>{}, op0, op1);
            else
            {
                if constexpr (K2 != K) { if (Kr == K2) stream_ops(std::integral_constant<int, K2>{}, std::integral_constant<int, PM_2>{}, op0, op1); }
                if constexpr (KH != K && KH != K2) { if (Kr == KH) stream_ops(std::integral_constant<int, KH>{}, std::integral_constant<int, 2>{}, op0, op1); }
            }
            op0 = op1;
        }
    }
    else
    {
        // =========================================================================================== service waves
        const int sw = wave - PS_SW, shw = 2 * sw + (lane >> 5);          // service wave 0..3, service half-wave 0..7
        __builtin_amdgcn_s_setprio(3);                                    // the workgroup's latency chain runs here
        // bounded waits: every wait of this wave gives up after `slim` polls.  The FIRST time-out anywhere in the workgroup (an edge counter, a tagged line that never came: the
        // grid is not co-resident, a producer died) sets the sticky error word, drops this wave's limit to zero and raises an LDS flag the other service waves pick up once per
        // op -- every later wait then fails after one poll instead of spinning its full limit again (129 ops x several waits x 2^17 polls: minutes), and the lm_head writes NaN
        // logits (the fx pipeline's poison convention): a timed-out step cannot be consumed silently
        int slim = a.spin_limit;
        auto ps_timeout = [&] (uint32_t errbit)
        {
            slim = 0;
            if (lane == 0)
            {
                __hip_atomic_fetch_or(a.err, errbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (the pinned host mirror's address lives behind the error word, not in the kernel arguments: two scalar registers less in every path that never times out)
                uint32_t* const eh = *((uint32_t* const*) (a.err + 2));
                if (eh) __hip_atomic_store(eh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(lctl + PS_C_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        uint32_t tgt_a = 0u, tgt_o = 0u, tgt_x = 0u;
        float r_last = 1.0f;                                              // the row scale (1 / rms) this workgroup last knew: DIRECT RMSNorm ops form their quads with it
        auto poll_cnt = [&] (int cop, uint32_t errbit)                    // every workgroup has arrived at counter `cop` (8 XCD shards)
        {
            if (slim == 0) return;
            const uint32_t* cbase = a.cnt; asm volatile("" : "+s"(cbase));            // (address formed here, not hoisted: see the op loop)
            const uint32_t PS_GLOBAL* c = ps_g(cbase + ((size_t) cop * 8 + (lane & 7)) * 16);
            const uint32_t expect = (uint32_t) ((ncu - (lane & 7) + 7) >> 3);
            // two polls in flight, half a round trip apart: the arrival of the last workgroup is seen after ~ half a memory round trip on average
            uint32_t v0 = __hip_atomic_load((uint32_t PS_GLOBAL*) c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_sleep(12);
            for (int spins = 0;; ++spins)
            {
                uint32_t v1 = __hip_atomic_load((uint32_t PS_GLOBAL*) c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_ballot_w64(v0 < expect) == 0ull) break;
                v0 = __hip_atomic_load((uint32_t PS_GLOBAL*) c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_ballot_w64(v1 < expect) == 0ull) break;
                if (spins > slim) { ps_timeout(errbit); break; }
            }
        };
        auto arrive = [&] (int cop) { if (lane == 0) __hip_atomic_fetch_add((uint32_t PS_GLOBAL*) (a.cnt + ((size_t) cop * 8 + (cu & 7)) * 16), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

        const ps_op_p ops_c = (ps_op_p) a.ops;
        PsTile tl_next = ps_load_tile(a.tiles, (size_t) cu);
        for (int op = 0; op < nops; ++op)
        {
            const ps_op_p O = ops_c + op;
            const PsTile tl = tl_next;
            // the lane coordinates are re-derived per op from an opaque copy of the thread id (and once more behind the attention item): everything computed from them then
            // lives inside the op -- hoisted out of the op loop, a dozen per-lane offsets stayed live across the attention's token loop, whose registers the allocator then
            // found in scratch (and a kernel with scratch does not keep the grid co-resident)
            int tid_v = tid;
            asm volatile("" : "+v"(tid_v));
            int lane = tid_v & 63, l32 = lane & 31, shw = 2 * sw + (lane >> 5);
            // (per-lane addresses off these bases are formed where they are used: hoisted out of the op loop they stay live for the whole launch and are
            //  the values the allocator spills -- 36 B of scratch; the empty asm makes the base a fresh value per op)
            const float* rope_sin_p = a.rope_sin; const float* rope_cos_p = a.rope_cos; half_t* q_out_p = a.q_out; half_t* logits_p = a.logits; uint32_t* cnt_p = a.cnt;
            asm volatile("" : "+s"(rope_sin_p), "+s"(rope_cos_p), "+s"(q_out_p), "+s"(logits_p), "+s"(cnt_p));
            const bool active = tl.mat >= 0;
            const int b0 = tl.b0, nb = active ? tl.nb : 0, W = active ? tl.ncb : 0;
            const int in_raw = O->in_type, out_raw = O->out_type, in_type = in_raw & 0xff, out_type = out_raw & 0xff, kk = O->k, nblk = kk >> 7;
            const bool in_direct = (in_raw & PS_DIRECT) != 0, out_direct = (out_raw & PS_DIRECT) != 0;       // DIRECT residual edge (exl3_pstep.cuh)
            const bool in_attn = ATT && (in_raw & PS_ATTN) != 0;                                               // the attention inside o_proj's preparation
            const uint32_t tag_out = (epoch << 12) | (uint32_t) (op + 1), tag_in = (epoch << 12) | (uint32_t) op;       // tag of op i = (epoch, i + 1), never 0
            float* const bsum = bsum2 + (op & 1) * 64;
            const int* const seginfo = seginfo2 + (op & 1) * 64;
            if (sw == 0) PS_T(3);

            if (in_type == PS_IN_QKV && in_attn)
            {
                // (a) of the attention inside o_proj's preparation (see the preparation branch below): this workgroup's item, BEFORE the op's other operands are requested
                // (the token loop needs the registers)
                if (sw == 0) PS_T(4);
                const PsAtt PS_CONST* const AT = (const PsAtt PS_CONST*) &O->mat[1];
                const int gq = AT->gq, ns = AT->nsplit;                // gq: query heads per kv head = query BLOCKS per kv block
                const bool hd64 = O->hd == 64;
                const int rows = hd64 ? 2 * gq : gq;                   // rows of the query operand (<= 8)
                const ps_rsrc_t rrec = ps_rsrc(AT->rec), rst = ps_rsrc(AT->stats);
                auto svc_bar = [&] () { tgt_o += PS_NSV; c_inc(PS_C_O); c_spin(PS_C_O, tgt_o); };
                // splits in use: enough for one 128-token step of the item's eight waves each, at most the plan's nsplit; items of the other splits have nothing to do
                const int len = __builtin_amdgcn_readfirstlane(ps_g(a.seqlens)[0]);
                const int ns_eff = att_nse(len, ns);
                const int item = (tl.side >= 0 && tl.side % ns < ns_eff) ? tl.side : -1;
                if (item >= 0)
                {
                    const AttItem it = att_make(O, item, len, ns, ns_eff, AT->hkv);
                    AttWords w0 = att_load(it, sw, lane, 0);
                    // ---- tasks, one per half-wave: 0 .. gq - 1 = query head h * gq + task, gq = the new token's K row, gq + 1 = its V row (owner only); a second round for gq = 7, 8
                    {
                        const uint32_t set_k = (uint32_t) ((const char*) O->in_slab[1] - (const char*) O->in_slab[0]), set_v = (uint32_t) ((const char*) O->in_slab[2] - (const char*) O->in_slab[0]);
                        const ps_rsrc_t rq = ps_rsrc(O->in_slab[0]);                        // (q, k and v slab sets lie in one allocation: one resource, per-lane offsets)
                        const int S_q = O->S_in;
                        // head_dim 64 (Llama-3.2-1B): a 128-value block holds TWO heads -- the tasks are still 128-value blocks (gq query blocks of two adjacent heads each, the K and
                        // the V block of the block's two kv heads), the rope partner distance and the frequency index follow the 64-wide head, the operand has 2 gq rows: row
                        // i = (sub, qi) keeps its 64 dims in half `sub` of the 128-wide row and zeros in the other (attn_decode_wide_kernel's HD64 form)
                        const int ph = O->hd >> 3;
                        float4_t sn4 = { 0.f, 0.f, 0.f, 0.f }, cs4 = sn4;
                        if (O->rope_mode == 2) { const int f = 4 * (l32 & (ph - 1)); sn4 = *ps_g((const float4_t*) (rope_sin_p + f)); cs4 = *ps_g((const float4_t*) (rope_cos_p + f)); }
                        else { const int f = 2 * (l32 & (2 * ph - 1)); sn4.x = ps_g(rope_sin_p)[f]; sn4.y = ps_g(rope_sin_p)[f + 1]; cs4.x = ps_g(rope_cos_p)[f]; cs4.y = ps_g(rope_cos_p)[f + 1]; }
                        #pragma nounroll
                        for (int r = 0; r < (gq + 2 + 7) / 8; ++r)
                        {
                            if (r > 0 && !it.owner) break;
                            const int task = r * 8 + shw;
                            const int kind = task < gq ? 0 : task - gq + 1;                 // 0: query, 1: K row, 2: V row, >= 3: nothing
                            const int tw = r * 8 + 2 * sw;
                            const bool wave_has = tw < gq || (it.owner && tw + 1 >= gq && tw <= gq + 1);      // wave-uniform: tasks tw, tw + 1
                            if (wave_has)
                            {
                                const bool kvt = kind == 1 || kind == 2;
                                const int cblk = kvt ? it.h : it.h * gq + min(task, gq - 1);
                                const half_t* svp = (kind == 1 ? O->in_svh[1] : (kind == 2 ? O->in_svh[2] : O->in_svh[0])) + (size_t) cblk * 128;
                                const half4_t sc = ps_g((const half4_t*) svp)[l32];
                                const uint32_t boff = (kind == 1 ? set_k : (kind == 2 ? set_v : 0u)) + (uint32_t) cblk * (uint32_t) S_q * PS_PLINE_BYTES;
                                float4_t ysum;
                                for (int spins = 0;; ++spins)
                                {
                                    bool ok = true;
                                    ysum = ps_slab_sum<PS_SLAB_NB>(rq, boff, S_q, l32, tag_in, ok);
                                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                                    if (spins > slim) { ps_timeout(2u); break; }
                                    __builtin_amdgcn_s_sleep(PS_POLL_SLEEP);
                                }
                                const GemvRescale rs0 = { nullptr, nullptr, 0, 0.0f };
                                const half4_t y = qkv_block_finish(ysum, sc, rs0, 0, l32, 0.0f, 0.0f, kind != 2, O->rope_mode, ph, sn4, cs4);
                                float v0 = (float) y.x, v1 = (float) y.y, v2 = (float) y.z, v3 = (float) y.w;
                                const int64_t token_pos = a.slots[0];
                                const int64_t gb = token_pos * it.G + it.h * 4 + (l32 >> 3);
                                uint32_t* cw = kind == 2 ? O->v_cache : O->k_cache; half_t* cs = kind == 2 ? O->v_scales : O->k_scales;
                                const bool actkv = it.owner && kvt;
                                kv_quant_regs<4>(v0, v1, v2, v3, cw + gb * 4, cs + gb, actkv, lane);
                                kv_quant_regs<4>(v0, v1, v2, v3, &att_newkv[(kind == 2 ? 16 : 0) + (l32 >> 3) * 4], &att_newsc[(kind == 2 ? 4 : 0) + (l32 >> 3)], actkv, lane);
                                if (kind == 0 && q_out_p && it.split == 0) ((half4_t PS_GLOBAL*) (q_out_p + (size_t) cblk * 128))[l32] = y;
                                kvg_had32(v0, v1, v2, v3, lane);
                                const float fq = ATT_R32 * a.att_scale * 1.44269504f;
                                const float vv[4] = { v0 * fq, v1 * fq, v2 * fq, v3 * fq };
                                if (kind == 0)
                                {
                                    if (hd64)
                                    {
                                        // lanes 0-15: head 2 task, lanes 16-31: head 2 task + 1
                                        const int i = 2 * task + (l32 >> 4), sub = i >= gq ? 1 : 0;
                                        #pragma unroll
                                        for (int e = 0; e < 4; ++e)
                                        {
                                            const int d = 64 * sub + 4 * (l32 & 15) + e, d8 = d & 7, dz = d ^ 64;
                                            att_q[i * 128 + (d & ~7) + 2 * (d8 & 3) + (d8 >> 2)] = (half_t) vv[e];
                                            att_q[i * 128 + (dz & ~7) + 2 * (d8 & 3) + (d8 >> 2)] = (half_t) 0.0f;
                                        }
                                    }
                                    else
                                    {
                                        #pragma unroll
                                        for (int e = 0; e < 4; ++e)
                                        {
                                            const int d = 4 * l32 + e, d8 = d & 7;
                                            att_q[task * 128 + (d & ~7) + 2 * (d8 & 3) + (d8 >> 2)] = (half_t) vv[e];
                                        }
                                    }
                                }
                            }
                            if (r == 0 && shw >= rows)
                            {
                                #pragma unroll
                                for (int e = 0; e < 4; ++e) att_q[shw * 128 + 4 * l32 + e] = (half_t) 0.0f;      // rows >= rows of the query operand stay zero
                            }
                        }
                    }
                    svc_bar();
                    c_set(PS_C_Q, (uint32_t) (op + 1));                    // streaming waves 0..3 join the token loop
                    if (sw == 0) PS_T(8);
                    att_tokens(it, sw, lane, w0, tgt_x);
                    if (shw < rows)
                    {
                        // half-wave i finishes row i of the kv block: lane l owns natural dims 4 l .. 4 l + 3; the record = 32 tagged granules + one statistics granule.
                        // head_dim 64: row i = (sub, qi) -> record qi of the block, statistics `sub`, its own 64 of the 128 accumulators (the lanes of that half)
                        const int i = shw;
                        float M = -1.0e30f;
                        #pragma unroll
                        for (int w = 0; w < 8; ++w) M = fmaxf(M, att_ml[(w * 8 + i) * 2]);
                        float L = 0.0f, Oa[4] = { 0.f, 0.f, 0.f, 0.f };
                        #pragma unroll
                        for (int w = 0; w < 8; ++w)
                        {
                            const float mw = att_ml[(w * 8 + i) * 2];
                            const float e = mw > -1.0e29f ? __builtin_amdgcn_exp2f(mw - M) : 0.0f;
                            L += att_ml[(w * 8 + i) * 2 + 1] * e;
                            #pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4)
                            {
                                const int d = 4 * l32 + e4, d8 = d & 7;
                                Oa[e4] += ((const float*) att_vt)[(w * 8 + i) * 128 + (d & ~7) + 2 * (d8 & 3) + (d8 >> 2)] * e;
                            }
                        }
                        const int sub = hd64 ? (i >= gq ? 1 : 0) : 0, qi = i - sub * gq;
                        const int rec_ = it.h * gq + qi;
                        // (the record's 128 accumulators as fp16 pairs: ONE 16-byte granule { O0 O1, tag, O2 O3, tag } per lane, 512 bytes per record -- the merge reads
                        //  nb x splits of them per workgroup; the merged output is rounded to fp16 anyway)
                        const uint32_t ro = ((uint32_t) rec_ * (uint32_t) ns + (uint32_t) it.split) * 512u + (uint32_t) l32 * 16u;
                        if (!hd64 || (l32 >> 4) == sub)
                            ps_st128(rrec, ro, uint4_t{ half2_as_u32(half2_t{ f2h(Oa[0]), f2h(Oa[1]) }), tag_out, half2_as_u32(half2_t{ f2h(Oa[2]), f2h(Oa[3]) }), tag_out });
                        const int srow = hd64 ? 2 * rec_ + sub : rec_;
                        if (l32 == 0) ps_st128(rst, ((uint32_t) srow * PS_ATT_MAX_SPLITS + (uint32_t) it.split) * 16u, uint4_t{ __float_as_uint(M * 0.69314718f), tag_out, __float_as_uint(L), tag_out });
                    }
                }
                if (sw == 0) PS_T(8);
                asm volatile("" : "+v"(tid_v));
                lane = tid_v & 63; l32 = lane & 31; shw = 2 * sw + (lane >> 5);
            }
            // ---- static operands of the preparation tasks (weights: requested before any wait)
            half4_t wv[4], sv[4], sva = { 0, 0, 0, 0 }, svb = sva;
            float4_t sn4 = { 0.f, 0.f, 0.f, 0.f }, cs4 = sn4;
            #pragma unroll
            for (int i = 0; i < 4; ++i) { wv[i] = sva; sv[i] = sva; }
            const half_t* const suh_m = active ? O->mat[tl.mat].suh : nullptr;
            // NORM: service half-wave shw holds blocks shw + 8 it of the row (and rotates those that lie in the slice); QKV / ACT: task block b0 + shw (planner: nb <= 8)
            if (in_type == PS_IN_NORM && !in_direct)
            {
                #pragma unroll
                for (int it = 0; it < 4; ++it)
                {
                    const int blk = min(shw + 8 * it, nblk - 1);
                    wv[it] = ps_g((const half4_t*) (O->norm_w + (size_t) blk * 128))[l32];
                    if (active) sv[it] = ps_g((const half4_t*) (suh_m + (size_t) blk * 128))[l32];
                }
            }
            else if (active)
            {
                const int blk = b0 + min(shw, nb - 1);
                sv[0] = ps_g((const half4_t*) (suh_m + (size_t) blk * 128))[l32];
                sva = ps_g((const half4_t*) (O->in_svh[0] + (size_t) blk * 128))[l32];
                if (in_type == PS_IN_NORM) wv[0] = ps_g((const half4_t*) (O->norm_w + (size_t) blk * 128))[l32];      // DIRECT: block b0 + shw is this half-wave's task
                else if (in_type == PS_IN_ACT) svb = ps_g((const half4_t*) (O->in_svh[1] + (size_t) blk * 128))[l32];
                else
                {
                    const int ph = O->hd >> 3;
                    if (O->rope_mode == 2)
                    {
                        const int f = 4 * (l32 & (ph - 1));
                        sn4 = *ps_g((const float4_t*) (rope_sin_p + f)); cs4 = *ps_g((const float4_t*) (rope_cos_p + f));
                    }
                    else
                    {
                        const int f = 2 * (l32 & ((O->hd >> 2) - 1));
                        sn4.x = ps_g(rope_sin_p)[f]; sn4.y = ps_g(rope_sin_p)[f + 1]; cs4.x = ps_g(rope_cos_p)[f]; cs4.y = ps_g(rope_cos_p)[f + 1];
                    }
                }
            }

            // the output side's operands too (pointers, the column scales of the two column blocks this half-wave may finish): requested here, used after the streaming
            const ps_mat_p Mo = &O->mat[active ? tl.mat : 0];
            // (S_op: partial lines per column block of this op's output; a tensor-parallel rank's own lines are line0 .. line0 + S - 1 of the tp_world x S, and every
            //  rank's exchange buffer receives them: exl3_pstep.cuh)
            unsigned long long* const slab_p = Mo->slab; const half_t* const svh_p = Mo->svh; const int S_op = TP ? O->S_all : O->S;
            const int tpw = TP ? O->tp_world : 1, line0 = TP ? O->line0 : 0;
            constexpr bool tp_sys = TP;       // (requested HERE with the op's other descriptor fields: a scalar load inside the
                                                                                            //  publish sat on the chain of every op: 8B +2.5 %, 1B +4 %)
            half4_t scp[2];
            #pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
            {
                scp[r2] = half4_t{ 0, 0, 0, 0 };
                if (active && out_type == PS_OUT_ATOMIC) scp[r2] = ps_g((const half4_t*) (svh_p + (size_t) (tl.cb0 + min(shw + 8 * r2, W - 1)) * 128))[l32];
            }
            asm volatile("" :: "s"(slab_p), "s"(S_op), "s"(tpw), "s"(line0));

            // one block: x * suh -> 128-point Hadamard -> fp16 quads in LDS (+ the block's sum for the mul1 affine term)
            auto rotate_store = [&] (half4_t xv, half4_t svv, int blk_local, bool act)
            {
                xv = xv * svv;
                float h0 = (float) xv.x, h1 = (float) xv.y, h2 = (float) xv.z, h3 = (float) xv.w;
                had128_f32x4(h0, h1, h2, h3, l32);
                const half2_t o01 = { f2h(h0 * HAD_R_SCALE_128), f2h(h1 * HAD_R_SCALE_128) };
                const half2_t o23 = { f2h(h2 * HAD_R_SCALE_128), f2h(h3 * HAD_R_SCALE_128) };
                float ts = ((float) o01.x + (float) o01.y) + ((float) o23.x + (float) o23.y);
                #pragma unroll
                for (int i = 1; i < 32; i <<= 1) ts += xor_lane(ts, i);
                if (act)
                {
                    if (l32 == 0) bsum[blk_local] = ts;
                    const int tr = blk_local * 8 + (l32 >> 2);
                    const int q0 = 2 * (l32 & 1), sp = (l32 >> 1) & 1;
                    char* base = quads + (size_t) (tr * 4 + q0) * 8 + sp * 4;
                    *((half2_t*) base) = o01;
                    *((half2_t*) (base + 8)) = o23;
                }
            };

            float fac_norm = 1.0f, r_next = r_last;                        // DIRECT: (true row scale) / (the scale the quads were formed with) multiplies the partial sums
            if (in_type == PS_IN_NORM && in_direct)
            {
                // DIRECT residual edge: this workgroup finishes the blocks of ITS slice of the row's new version from the producer op's partial lines -- ONE hop
                // (partial lines -> here) instead of two (partial lines -> owner -> owner's line -> every workgroup).  All eight service half-waves gather (half-wave h:
                // lines h, h + 8, ... of a block, two blocks per round while S_in <= 16), the block's half-wave adds the eight sums in the owners' order, output
                // Hadamard, the producer's svh, + the block of the previous version; one workgroup per slice (PS_TILE_Q_OUT) publishes the new blocks and their
                // sums of squares.  The quads are formed with r_last; the true scale needs every block's sum of squares: see the finish below
                if (sw == 0) PS_T(4);
                const int rver = O->rver;
                // (a producer with <= 8 slices: the block's own half-wave sums its partial lines in one round trip -- no gather through LDS, no counter among the service
                //  waves; more slices: all eight half-waves gather)
                const bool coop = O->S_in > PS_COOP_MIN;
                // the previous version's block of this half-wave's task (tagged by the workgroup that published it two ops ago; version 0 = the caller's R) is REQUESTED here, in
                // front of the gather of the producer's partial lines, and checked behind it: it was a second dependent memory round trip after the gather (~ 0.8 us of every
                // direct RMSNorm op by the stamps -- the block has been in memory for two ops)
                const bool has_task = active && 2 * sw < nb;               // wave-uniform
                const int tb = min(shw, max(nb - 1, 0)), blk = b0 + tb;
                const bool act = active && shw < nb;
                const ps_rsrc_t rb = ps_rsrc(a.rbuf);
                const uint32_t ro_old = rver == 1 ? (uint32_t) blk * 1024u + (uint32_t) l32 * 32u
                                                  : (uint32_t) ((rver - 1) & 1) * PS_RROW_BYTES + (uint32_t) blk * PS_LINE_BYTES + (uint32_t) l32 * 16u;
                uint4_t pra = { 0u, 0u, 0u, 0u }, prc = pra;
#ifndef PS_ROLD_LATE
                if (has_task)
                {
                    if (rver == 1) { const ps_rsrc_t r0 = ps_rsrc(a.R); pra = ps_ld128(r0, ro_old); prc = ps_ld128(r0, ro_old + 16u); }
                    else { pra = ps_ld128(rb, ro_old); prc = ps_ld128(rb, ro_old + 512u); }
                }
#endif
                float4_t ys_own = { 0.f, 0.f, 0.f, 0.f };
                auto gather_direct = [&] (auto sysc)
                {
                constexpr bool SYS = decltype(sysc)::value;                 // a tensor-parallel plan reads its exchange buffer with system-scope loads (peer GPUs write it)
                if (active && !coop && 2 * sw < nb)
                {
                    const ps_rsrc_t rp = ps_rsrc(O->in_slab[0]);
                    const int blk_o = b0 + min(shw, nb - 1);
                    for (int spins = 0;; ++spins)
                    {
                        bool ok = true;
                        ys_own = ps_slab_sum<8, SYS>(rp, (uint32_t) blk_o * (uint32_t) O->S_in * PS_PLINE_BYTES, O->S_in, l32, tag_in, ok);
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                        if (spins > slim) { ps_timeout(8u); break; }
                        __builtin_amdgcn_s_sleep(PS_POLL_SLEEP);
                    }
                }
                if (active && coop)
                {
                    const ps_rsrc_t rp = ps_rsrc(O->in_slab[0]);
                    const int S_p = O->S_in;
                    // NSLOT lines per half-wave and round trip: with <= 16 slices a half-wave has two lines per block -> NSLOT / 2 blocks per round (all four with the fp16
                    // lines); with up to 32 slices four lines per block
#ifndef PS_FP32_PARTIALS
                    constexpr int NSLOT = 8;
#else
                    constexpr int NSLOT = 4;
#endif
                    const bool two = S_p <= 16;
                    const int lpb = two ? 2 : 4, bpr = NSLOT / lpb;           // lines per block and half-wave, blocks per round
                    const int rounds = (nb + bpr - 1) / bpr;
                    for (int r = 0; r < rounds; ++r)
                    {
                        float4_t acc[NSLOT / 2];
                        for (int spins = 0;; ++spins)
                        {
                            PsPl t[NSLOT];
                            bool ok = true;
                            #pragma unroll
                            for (int i = 0; i < NSLOT; ++i)
                            {
                                const int jj = r * bpr + (two ? (i >> 1) : (i >> 2)), s_ = shw + 8 * (two ? (i & 1) : (i & 3));
                                t[i] = ps_pl_load_t<SYS>(rp, ((uint32_t) (b0 + min(jj, nb - 1)) * (uint32_t) S_p + (uint32_t) min(s_, S_p - 1)) * PS_PLINE_BYTES, l32);
                            }
                            #pragma unroll
                            for (int q = 0; q < NSLOT / 2; ++q) acc[q] = float4_t{ 0.f, 0.f, 0.f, 0.f };
                            auto take = [&] (auto twoc)
                            {
                                constexpr bool TWO = decltype(twoc)::value;
                                ps_static_for<0, NSLOT>([&] (auto ic)
                                {
                                    constexpr int i = decltype(ic)::value, bi = TWO ? (i >> 1) : (i >> 2);
                                    const int jj = r * bpr + bi, s_ = shw + 8 * (TWO ? (i & 1) : (i & 3));
                                    const bool use = jj < nb && s_ < S_p;
                                    ok &= !use | ps_pl_ok(t[i], tag_in);
                                    const float4_t x = ps_pl_val(t[i], use ? 0xffffffffu : 0u);
                                    acc[bi].x += x.x; acc[bi].y += x.y; acc[bi].z += x.z; acc[bi].w += x.w;
                                });
                            };
                            if (two) take(std::true_type{}); else take(std::false_type{});
                            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                            if (spins > slim) { ps_timeout(8u); break; }
                            __builtin_amdgcn_s_sleep(PS_POLL_SLEEP);
                        }
                        #pragma unroll
                        for (int q = 0; q < NSLOT / 2; ++q)
                        {
                            const int jj = r * bpr + q;
                            if (q < bpr && jj < nb) ((float4_t*) (gath + ((size_t) (jj & 3) * 8 + shw) * 128))[l32] = acc[q];
                        }
                    }
                    tgt_o += PS_NSV;
                    c_inc(PS_C_O);
                }
                };
                gather_direct(std::integral_constant<bool, TP>{});
                float4_t rold = { 0.f, 0.f, 0.f, 0.f };
                if (has_task)
                {
                    if (rver == 1)
                    {
#ifdef PS_ROLD_LATE
                        { const ps_rsrc_t r0 = ps_rsrc(a.R); pra = ps_ld128(r0, ro_old); prc = ps_ld128(r0, ro_old + 16u); }
#endif
                        rold = float4_t{ fx_to_float(pra.x, pra.y), fx_to_float(pra.z, pra.w), fx_to_float(prc.x, prc.y), fx_to_float(prc.z, prc.w) };
                    }
                    else
                    {
                        const uint32_t tag_old = (epoch << 12) | (uint32_t) (op - 2);
#ifdef PS_ROLD_LATE
                        pra = ps_ld128(rb, ro_old); prc = ps_ld128(rb, ro_old + 512u);
#endif
                        for (int spins = 0;; ++spins)
                        {
                            rold = float4_t{ __uint_as_float(pra.x), __uint_as_float(pra.z), __uint_as_float(prc.x), __uint_as_float(prc.z) };
                            const bool ok = (pra.y == tag_old) & (pra.w == tag_old) & (prc.y == tag_old) & (prc.w == tag_old);
                            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                            if (spins > slim) { ps_timeout(1u); break; }
                            __builtin_amdgcn_s_sleep(PS_POLL_SLEEP);
                            pra = ps_ld128(rb, ro_old); prc = ps_ld128(rb, ro_old + 512u);
                        }
                    }
                }
                // every row-related load of this wave is in registers: the read gate (all workgroups arrive, also those without a rectangle)
                tgt_a += PS_NSV;
                { const uint32_t old = c_inc(PS_C_A); if (old + 1u == tgt_a) arrive(op); }
                if (active && coop) c_spin(PS_C_O, tgt_o);
                if (sw == 0) PS_T(8);
                if (has_task)
                {
                    float4_t ys = ys_own;
                    if (coop)
                    {
                        #pragma unroll
                        for (int h = 0; h < 8; ++h)
                        {
                            const float4_t t = ((const float4_t*) (gath + ((size_t) (tb & 3) * 8 + h) * 128))[l32];
                            ys.x += t.x; ys.y += t.y; ys.z += t.z; ys.w += t.w;
                        }
                    }
                    float h0, h1, h2, h3;
                    out_had(ys, l32, h0, h1, h2, h3);
                    const float n0 = rold.x + h0 * (float) sva.x, n1 = rold.y + h1 * (float) sva.y, n2 = rold.z + h2 * (float) sva.z, n3 = rold.w + h3 * (float) sva.w;
                    float ssq = n0 * n0;
                    ssq = __builtin_fmaf(n1, n1, ssq); ssq = __builtin_fmaf(n2, n2, ssq); ssq = __builtin_fmaf(n3, n3, ssq);
                    #pragma unroll
                    for (int i = 1; i < 32; i <<= 1) ssq += xor_lane(ssq, i);
                    if (act && (tl.flags & PS_TILE_Q_OUT))
                    {
                        const uint32_t no = (uint32_t) (rver & 1) * PS_RROW_BYTES + (uint32_t) blk * PS_LINE_BYTES + (uint32_t) l32 * 16u;
                        ps_st128(rb, no, uint4_t{ __float_as_uint(n0), tag_in, __float_as_uint(n1), tag_in });
                        ps_st128(rb, no + 512, uint4_t{ __float_as_uint(n2), tag_in, __float_as_uint(n3), tag_in });
                        if (l32 == 0) ps_st128(rb, 2u * PS_RROW_BYTES + (uint32_t) ((rver & 1) * PS_MAX_ROW_BLOCKS + blk) * 16u, uint4_t{ __float_as_uint(ssq), tag_in, 0u, tag_in });
                    }
                    const half4_t xv = { f2h(n0 * (float) wv[0].x * r_last), f2h(n1 * (float) wv[0].y * r_last), f2h(n2 * (float) wv[0].z * r_last), f2h(n3 * (float) wv[0].w * r_last) };
                    rotate_store(xv, sv[0], tb, act);
                }
                if (sw == 0) PS_T(9);
            }
#ifndef PS_ABL_NO_COLD
            else if (in_type == PS_IN_NORM)
            {
                if (sw == 0) PS_T(4);
                // exact RMSNorm: every workgroup reads the whole row -- service half-wave shw takes blocks shw + 8 i (block sums of squares in rms_norm's
                // order: norm.cu:20-120) -- then rotates the blocks of its slice.  Version 0 of the row is the caller's fixed-point R; later versions are
                // tagged fp32 lines written by the owners of the previous op's column blocks (no edge: the lines carry the producer's tag)
                half4_t xr[PS_ROW_IT];
                const int rver = O->rver;
                if (rver == 0)
                {
                    const ps_rsrc_t rR = ps_rsrc(a.R);
                    #pragma unroll
                    for (int it = 0; it < PS_ROW_IT; ++it)
                    {
                        xr[it] = half4_t{ 0, 0, 0, 0 };
                        if (8 * it < nblk)
                        {
                            const uint32_t o = (uint32_t) min(shw + 8 * it, nblk - 1) * 1024u + (uint32_t) l32 * 32u;
                            const uint4_t ra = ps_ld128(rR, o), rb = ps_ld128(rR, o + 16u);
                            auto fx = [] (uint32_t lo, uint32_t hi) -> half_t { return f2h(fx_to_float(lo, hi)); };
                            xr[it] = half4_t{ fx(ra.x, ra.y), fx(ra.z, ra.w), fx(rb.x, rb.y), fx(rb.z, rb.w) };
                        }
                    }
                }
                else
                {
                    const ps_rsrc_t rR = ps_rsrc(a.rbuf + (size_t) (rver & 1) * PS_MAX_ROW_BLOCKS * 128);
                    for (int spins = 0;; ++spins)
                    {
                        bool ok = true;
                        #pragma unroll
                        for (int it = 0; it < PS_ROW_IT; ++it)
                        {
                            xr[it] = half4_t{ 0, 0, 0, 0 };
                            if (8 * it < nblk)
                            {
                                const uint32_t o = (uint32_t) min(shw + 8 * it, nblk - 1) * PS_LINE_BYTES + (uint32_t) l32 * 16u;
                                const uint4_t ra = ps_ld128(rR, o), rb = ps_ld128(rR, o + 512u);
                                ok &= (ra.y == tag_in) & (ra.w == tag_in) & (rb.y == tag_in) & (rb.w == tag_in);
                                xr[it] = half4_t{ f2h(__uint_as_float(ra.x)), f2h(__uint_as_float(ra.z)), f2h(__uint_as_float(rb.x)), f2h(__uint_as_float(rb.z)) };
                            }
                        }
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                        if (spins > slim) { ps_timeout(1u); break; }
                        __builtin_amdgcn_s_sleep(PS_POLL_SLEEP);
                    }
                }
                #pragma unroll
                for (int it = 0; it < PS_ROW_IT; ++it) if (8 * it < nblk)
                {
                    const int blk = shw + 8 * it;
                    const float f0 = (float) xr[it].x, f1 = (float) xr[it].y, f2 = (float) xr[it].z, f3 = (float) xr[it].w;
                    float ssq = f0 * f0;
                    ssq = __builtin_fmaf(f1, f1, ssq); ssq = __builtin_fmaf(f2, f2, ssq); ssq = __builtin_fmaf(f3, f3, ssq);
                    #pragma unroll
                    for (int i = 1; i < 32; i <<= 1) ssq += xor_lane(ssq, i);
                    if (blk < nblk && l32 == 0) ssblk[blk] = ssq;
                }
                if (sw == 0) PS_T(8);
                // the row is in registers: this workgroup passes the read gate in front of the op that will overwrite this version's lines (loads only: no drain)
                tgt_a += PS_NSV;
                { const uint32_t old = c_inc(PS_C_A); if (old + 1u == tgt_a) arrive(op); }
                c_spin(PS_C_A, tgt_a);
                if (sw == 0) PS_T(9);
                float s2 = (l32 < nblk ? ssblk[l32] : 0.0f) + (l32 + 32 < nblk ? ssblk[l32 + 32] : 0.0f);
                #pragma unroll
                for (int i = 1; i < 32; i <<= 1) s2 += xor_lane(s2, i);
                const float r = __frsqrt_rn(s2 / (float) kk + O->eps);
                r_next = r;
                #pragma unroll
                for (int it = 0; it < PS_ROW_IT; ++it) if (8 * it < nblk)
                {
                    const int blk = shw + 8 * it;                           // the block held in xr[it]
                    const bool act = active && blk >= b0 && blk < b0 + nb;
                    if (__builtin_amdgcn_ballot_w64(act) != 0ull)           // (wave-uniform: neither half-wave holds a block of the slice -> nothing to rotate)
                    {
                        half4_t w_ = wv[it & 3], s_ = sv[it & 3];
                        if (it >= 4)
                        {
                            // (blocks 32 .. 63 of a row wider than 4096: norm weight and input scale fetched here, not ahead -- eight register pairs more per path otherwise)
                            const int bc_ = min(blk, nblk - 1);
                            w_ = ps_g((const half4_t*) (O->norm_w + (size_t) bc_ * 128))[l32];
                            s_ = ps_g((const half4_t*) (suh_m + (size_t) bc_ * 128))[l32];
                        }
                        const half4_t xv = { f2h((float) xr[it].x * (float) w_.x * r), f2h((float) xr[it].y * (float) w_.y * r),
                                             f2h((float) xr[it].z * (float) w_.z * r), f2h((float) xr[it].w * (float) w_.w * r) };
                        rotate_store(xv, s_, min(max(blk - b0, 0), max(nb - 1, 0)), act);
                    }
                }
            }
#endif
            else if (in_type == PS_IN_QKV && in_attn)
            {
                // ATTENTION inside o_proj's preparation (exl3_pstep.cuh: PS_ATTN).  (a) this workgroup's item = (kv head h, context split) of the decode attention over the
                // 4-bit paged cache: the matrix-pipe split kernel of exl3_attn_decode.hip (attn_decode_wide_kernel<GQ, FUSED>, head_dim 128) on the four service waves -- its
                // workgroup barriers are the service waves' LDS counter, its slab sums read tagged lines, its partial record leaves as a tagged line + a statistics granule;
                // (b) the merge of the records of the query heads of this workgroup's k-slice (exl3_gemv4's ATTM tasks; sequential over the splits) -> o_proj's quads.
                const PsAtt PS_CONST* const AT = (const PsAtt PS_CONST*) &O->mat[1];
                const int ns = AT->nsplit;
                const ps_rsrc_t rrec = ps_rsrc(AT->rec), rst = ps_rsrc(AT->stats);
                if (sw == 0) PS_T(9);
                // ---- (b) the attention output of the query heads b0 .. b0 + nb - 1 (= the Hadamard blocks of o_proj's slice): merge of all splits' records
                if (active && 2 * sw < nb)
                {
                    const bool act = shw < nb;
                    const int tb = min(shw, nb - 1), blk = b0 + tb;
                    const int len = __builtin_amdgcn_readfirstlane(ps_g(a.seqlens)[0]);
                    const int nse = att_nse(len, ns);                          // the splits in use (as the items compute it)
                    // head_dim 128: block blk = query head blk = record blk, lane l32 holds the statistics of split l32.  head_dim 64: the block holds query heads 2 blk and
                    // 2 blk + 1 (lanes 0-15 / 16-31); head qh belongs to kv head qh / gq = half (kv & 1) of the records of kv block kv >> 1: per-lane record, statistics
                    // row and granule lane; lane (l32 & 15) of a head holds split (l32 & 15) (exl3_gemv4's ATTM tasks)
                    const bool hd64 = O->hd == 64;
                    const int gq_ = AT->gq;
                    const int at_lr = hd64 ? (l32 & 15) : l32;
                    const int qh = hd64 ? 2 * blk + (l32 >> 4) : blk;
                    const int kv = qh / gq_, qi = qh - kv * gq_;
                    const int rec_ = hd64 ? (kv >> 1) * gq_ + qi : blk, half_ = hd64 ? (kv & 1) : 0;
                    const int srow = hd64 ? 2 * rec_ + half_ : rec_;
                    const uint32_t glane = hd64 ? (uint32_t) (16 * half_ + at_lr) : (uint32_t) l32;      // the granule of the record that holds this lane's four dims
                    float m_s = -1.0e30f, l_s = 0.0f;
                    for (int spins = 0;; ++spins)
                    {
                        const uint4_t g = ps_ld128(rst, ((uint32_t) srow * PS_ATT_MAX_SPLITS + (uint32_t) min(at_lr, nse - 1)) * 16u);
                        const bool has = at_lr < nse;
                        m_s = has ? __uint_as_float(g.x) : -1.0e30f; l_s = has ? __uint_as_float(g.z) : 0.0f;
                        const bool ok = !has | ((g.y == tag_out) & (g.w == tag_out));
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                        if (spins > slim) { ps_timeout(2u); break; }
                        __builtin_amdgcn_s_sleep(PS_POLL_SLEEP);
                    }
                    float M = m_s;
                    #pragma unroll
                    for (int i = 1; i < 16; i <<= 1) M = fmaxf(M, xor_lane(M, i));
                    if (!hd64) M = fmaxf(M, xor_lane(M, 16));
                    const float e_s = m_s > -1.0e29f ? __expf(m_s - M) : 0.0f;
                    float L = l_s * e_s;
                    #pragma unroll
                    for (int i = 1; i < 16; i <<= 1) L += xor_lane(L, i);
                    if (!hd64) L += xor_lane(L, 16);
                    const int lbase = lane - at_lr;
                    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
                    for (int s0 = 0; s0 < nse; s0 += 8)                       // eight records per round trip
                    {
                        uint4_t t[8];
                        for (int spins = 0;; ++spins)
                        {
                            bool ok = true;
                            #pragma unroll
                            for (int u = 0; u < 8; ++u)
                                t[u] = ps_ld128(rrec, ((uint32_t) rec_ * (uint32_t) ns + (uint32_t) min(s0 + u, nse - 1)) * 512u + glane * 16u);
                            #pragma unroll
                            for (int u = 0; u < 8; ++u) ok &= (t[u].y == tag_out) & (t[u].w == tag_out);
                            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                            if (spins > slim) { ps_timeout(2u); break; }
                            __builtin_amdgcn_s_sleep(PS_POLL_SLEEP);
                        }
                        #pragma unroll
                        for (int u = 0; u < 8; ++u)
                        {
                            const float ev = (s0 + u < nse) ? __shfl(e_s, lbase + min(s0 + u, nse - 1), 64) : 0.0f;
                            const half2_t a01 = u32_as_half2(t[u].x), a23 = u32_as_half2(t[u].z);
                            o0 += (float) a01.x * ev; o1 += (float) a01.y * ev; o2 += (float) a23.x * ev; o3 += (float) a23.y * ev;
                        }
                    }
                    const float inv = L > 0.0f ? 1.0f / L : 0.0f;
                    float v0 = o0 * inv, v1 = o1 * inv, v2 = o2 * inv, v3 = o3 * inv;
                    kvg_had32(v0, v1, v2, v3, lane);
                    const half4_t xa = { f2h(v0 * ATT_R32), f2h(v1 * ATT_R32), f2h(v2 * ATT_R32), f2h(v3 * ATT_R32) };
                    rotate_store(xa, sv[0], tb, act);
                }
            }
            else if (in_type == PS_IN_QKV)
            {
                // q block (b0 + shw) finished from the q|k|v op's tagged slab lines exactly as exl3_glue_qkv_tab finishes it (qkv_block_finish), then o_proj's input rotation
                if (sw == 0) PS_T(4);
                if (active && 2 * sw < nb)                                 // wave-uniform: this wave owns at least one task
                {
                    const bool act = shw < nb;
                    const int tb = min(shw, nb - 1), blk = b0 + tb;
                    const ps_rsrc_t rq = ps_rsrc(O->in_slab[0]);
                    float4_t ys;
                    for (int spins = 0;; ++spins)
                    {
                        bool ok = true;
                        ys = ps_slab_sum<PS_SLAB_NB>(rq, (uint32_t) blk * (uint32_t) O->S_in * PS_PLINE_BYTES, O->S_in, l32, tag_in, ok);
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                        if (spins > slim) { ps_timeout(2u); break; }
                        __builtin_amdgcn_s_sleep(PS_POLL_SLEEP);
                    }
                    if (sw == 0) PS_T(8);
                    const GemvRescale rs0 = { nullptr, nullptr, 0, 0.0f };
                    const half4_t xq = qkv_block_finish(ys, sva, rs0, 0, l32, 0.0f, 0.0f, true, O->rope_mode, O->hd >> 3, sn4, cs4);
                    if (act && (tl.flags & PS_TILE_Q_OUT) && q_out_p) ((half4_t PS_GLOBAL*) (q_out_p + (size_t) blk * 128))[l32] = xq;
                    rotate_store(xq, sv[0], tb, act);
                }
            }
            else
            {
                // silu(g) * u of block (b0 + shw): slab lines in slice order, output Hadamards, svh -- the arithmetic of glue_act_kernel / generation 4's ACT tasks
                if (sw == 0) PS_T(4);
                if (active && 2 * sw < nb)
                {
                    const bool act = shw < nb;
                    const int tb = min(shw, nb - 1), blk = b0 + tb;
                    const ps_rsrc_t rg = ps_rsrc(O->in_slab[0]), ru = ps_rsrc(O->in_slab[1]);
                    float4_t vg, vu;
                    for (int spins = 0;; ++spins)
                    {
                        bool ok = true;
                        ps_slab_sum2<PS_SLAB2_NB>(rg, ru, (uint32_t) blk * (uint32_t) O->S_in * PS_PLINE_BYTES, O->S_in, l32, tag_in, ok, vg, vu);
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                        if (spins > slim) { ps_timeout(2u); break; }
                        __builtin_amdgcn_s_sleep(PS_POLL_SLEEP);
                    }
                    if (sw == 0) PS_T(8);
                    float g0, g1, g2, g3, u0, u1, u2, u3;
                    out_had(vg, l32, g0, g1, g2, g3);
                    out_had(vu, l32, u0, u1, u2, u3);
                    const half4_t gh = half4_t{ f2h(g0), f2h(g1), f2h(g2), f2h(g3) } * sva;
                    const half4_t uh = half4_t{ f2h(u0), f2h(u1), f2h(u2), f2h(u3) } * svb;
                    auto silu_mul = [] (half_t g, half_t u) -> half_t { float gf = (float) g; return f2h(gf / (1.0f + __expf(-gf)) * (float) u); };
                    const half4_t xa = { silu_mul(gh.x, uh.x), silu_mul(gh.y, uh.y), silu_mul(gh.z, uh.z), silu_mul(gh.w, uh.w) };
                    rotate_store(xa, sv[0], tb, act);
                }
            }
            c_inc(PS_C_T);                                                   // (release: this wave's quads are in LDS)
            PS_T(28 + sw);
            if (slim != 0 && __hip_atomic_load(lctl + PS_C_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) slim = 0;      // (another wave of the workgroup timed out)
            if (op + 1 < nops)
            {
                // the next op's rectangle and the cache lines of its descriptor, requested under the streaming (scalar-cache misses otherwise open the next op)
                tl_next = ps_load_tile(a.tiles, (size_t) (op + 1) * ncu + cu);
                const int PS_CONST* on = (const int PS_CONST*) (O + 1);
                const int t0 = on[0], t1 = on[16], t2 = on[32], t3 = on[48], t4 = on[64];
                asm volatile("" :: "s"(t0), "s"(t1), "s"(t2), "s"(t3), "s"(t4));
            }
            if (sw == 0) PS_T(5);

#ifndef PS_SUM_HALFWAVES
            // ---- the sum of the streaming waves' partial rows, prepared UNDER the streaming.  One column block per service WAVE (column index wave-uniform: the partition
            // arithmetic is scalar and done here, off the chain), its candidate rows split between the wave's two half-waves and the halves added through v_permlane32_swap: a
            // one-column rectangle = rows 0-5 | 6-11 (unit-less waves leave zero rows: no masks), a wider one = candidates 0-3 | 4-7 of the partition formula.  It was one
            // column per HALF-wave: a per-lane column index -> ~200 VALU instructions of integer arithmetic and masks between "last streaming wave done" and "partial line
            // published" (0.9-1.2 us by the stamps), and with W <= 2 one service wave did it all.
            // (mul1: value = kinv * (1024 + byte sum) + kbias, applied here once per output; 3INST / mcg operands are the weights themselves)
            const bool op_mul1 = CB == EXL3_CB_MUL1;
            const float kinv_s = op_mul1 ? (float) u16_as_half(0x1eeeu) : 1.0f;
            float bb_s = 0.0f;
            uint32_t ctab0 = 0u, ctab1 = 0u;
            // (the waves whose run touches column j and which of their two partial rows belongs to it, straight from the partition (ps_wave_range): a run that STARTS in the
            //  column has its row in segment 0, one that started in the column before in segment 1.  At most EIGHT consecutive waves touch a column -- the uniform partition
            //  by arithmetic (tests/test_pstep_plan.py), a weighted one because the planner checks it (wave_partition).  Packed: first candidate wave | candidate i counts << 8
            //  | its row is segment 1 << 16)
            auto col_table = [&] (int j) -> uint32_t
            {
                const int H = 4 * nb, T = H * W;
                const int lo_u = j * H, hi_u = lo_u + H;
                uint32_t inb = 0u, sgb = 0u;
                #pragma unroll
                for (int w = 0; w < PS_SW; ++w)
                {
                    const PsRange rg = ps_wave_range(tl, T, w);
                    const bool in_col = rg.u1 > rg.u0 && rg.u0 < hi_u && rg.u1 > lo_u;
                    inb |= (in_col ? 1u : 0u) << w; sgb |= (rg.u0 >= lo_u ? 0u : 1u) << w;
                }
                const int w_first = inb ? __builtin_ctz(inb) : 0;
                return (uint32_t) w_first | (((inb >> w_first) & 0xffu) << 8) | (((sgb >> w_first) & 0xffu) << 16);
            };
            half4_t scw0 = { 0, 0, 0, 0 }, scw1 = scw0;                     // (the lm_head: the column scales of this wave's first two column blocks)
            if (sw < W)
            {
                if (out_type == PS_OUT_FINAL)
                {
                    scw0 = ps_g((const half4_t*) (svh_p + (size_t) (tl.cb0 + sw) * 128))[l32];
                    scw1 = ps_g((const half4_t*) (svh_p + (size_t) (tl.cb0 + min(sw + PS_NSV, W - 1)) * 128))[l32];
                }
                c_spin(PS_C_T, (uint32_t) PS_NSV * (uint32_t) (op + 1));      // (every service wave's block sums are in LDS: true long before the streaming ends)
                float xs = 0.0f;
                for (int q0 = 0; q0 < nb; q0 += 32) if (q0 + l32 < nb) xs += bsum[q0 + l32];
                #pragma unroll
                for (int i = 1; i < 32; i <<= 1) xs += xor_lane(xs, i);
                bb_s = op_mul1 ? (float) u16_as_half(0xc931u) * xs : 0.0f;
                if (W > 1) ctab0 = col_table(sw);                             // (this wave's first two columns; a rectangle wider than eight: the third after them, on the chain)
                if (W > sw + PS_NSV) ctab1 = col_table(sw + PS_NSV);
            }
#endif
            // ---- an op that produces a new version of the row: its owners overwrite the lines of the version before the previous one -- the read gate of
            // the op that read THAT version (every workgroup had it in registers long ago); polled under the streaming
            if (out_type == PS_OUT_ATOMIC && !out_direct)
            {
                if (sw == 0) { const int gop = O->gate_op; if (gop >= 0) poll_cnt(gop, 4u); c_set(PS_C_G, (uint32_t) (op + 1)); }
            }
            // DIRECT RMSNorm op: the true row scale from the blocks' sums of squares (published by one workgroup per slice during ITS preparation, one hop
            // away: here long before the streaming ends); the partial sums are linear in the scale the quads were formed with
            if (in_type == PS_IN_NORM && in_direct && active)
            {
                const ps_rsrc_t rb = ps_rsrc(a.rbuf);
                const uint32_t go = 2u * PS_RROW_BYTES + (uint32_t) ((O->rver & 1) * PS_MAX_ROW_BLOCKS + min(l32, nblk - 1)) * 16u;
                const uint32_t go2 = 2u * PS_RROW_BYTES + (uint32_t) ((O->rver & 1) * PS_MAX_ROW_BLOCKS + min(l32 + 32, nblk - 1)) * 16u;      // (blocks 32 .. 63 of a row wider than 4096)
                float s2 = 0.0f;
                for (int spins = 0;; ++spins)
                {
                    const uint4_t g = ps_ld128(rb, go);
                    s2 = l32 < nblk ? __uint_as_float(g.x) : 0.0f;
                    bool ok = (g.y == tag_in) & (g.w == tag_in);
                    if (nblk > 32)
                    {
                        const uint4_t g2 = ps_ld128(rb, go2);
                        s2 += l32 + 32 < nblk ? __uint_as_float(g2.x) : 0.0f;
                        ok &= (g2.y == tag_in) & (g2.w == tag_in);
                    }
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                    if (spins > slim) { ps_timeout(1u); break; }
                    __builtin_amdgcn_s_sleep(PS_POLL_SLEEP);
                }
                #pragma unroll
                for (int i = 1; i < 32; i <<= 1) s2 += xor_lane(s2, i);
                r_next = __frsqrt_rn(s2 / (float) kk + O->eps);
                fac_norm = r_next / r_last;
            }

            // ---- the streaming waves' partial rows are in LDS: service half-wave shw finishes column blocks shw, shw + 8 of the rectangle
            // (the one LONG wait of a service wave sleeps between its polls: spinning at priority 3 it took ~ 10 % of its SIMD's VALU issue away from the three streaming
            //  waves it was waiting for -- same box 1.2797 -> 1.2734 ms; PS_WAIT_S_SPIN: the busy loop)
#ifdef PS_WAIT_S_SPIN
            c_spin(PS_C_S, (uint32_t) PS_SW * (uint32_t) (op + 1));
#else
            c_wait(PS_C_S, (uint32_t) PS_SW * (uint32_t) (op + 1));
#endif
            if (sw == 0) PS_T(6);
            // the lm_head poisons its logits if any wait of the launch has timed out by now: this workgroup's own flag, or the device's sticky error word (one load per step,
            // requested here, consumed behind the output Hadamard)
            uint32_t poison = 0u;                                            // OR-ed into the fp16 pairs of the logits: 0x7e00 | x is a NaN whatever x
            if (out_type == PS_OUT_FINAL)
            {
                const uint32_t e = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                poison = (slim == 0 || __builtin_amdgcn_readfirstlane((int) e) != 0) ? 0x7e007e00u : 0u;
            }
#ifndef PS_SUM_HALFWAVES
            {
                if (sw < W)
                {
                    const int l = l32, hi = lane >> 5;
                    const ps_rsrc_t rsl = ps_rsrc(slab_p);
                    auto halves = [] (float x) -> float
                    {
                        // r[0] = { x.lo, x.lo }, r[1] = { x.hi, x.hi } (lanes 0-31 | 32-63): the same sum lo + hi in every lane
                        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
                        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
                    };
                    auto publish = [&] (float4_t v, int j)
                    {
                        v.x = halves(v.x); v.y = halves(v.y); v.z = halves(v.z); v.w = halves(v.w);
                        v.x = (v.x * kinv_s + bb_s) * fac_norm; v.y = (v.y * kinv_s + bb_s) * fac_norm; v.z = (v.z * kinv_s + bb_s) * fac_norm; v.w = (v.w * kinv_s + bb_s) * fac_norm;
                        if (out_type == PS_OUT_FINAL)
                        {
                            // the lm_head (round 6: on the service-wave path too, so its rectangles can be shared by age group like the layers'): output Hadamard, column
                            // scale, fp16 logits; NaN if any wait of the launch timed out
                            const int cbl = tl.cb0 + j;
                            const half4_t sc = j == sw ? scw0 : (j == sw + PS_NSV ? scw1 : ps_g((const half4_t*) (svh_p + (size_t) cbl * 128))[l]);
                            float h0, h1, h2, h3;
                            out_had(v, l, h0, h1, h2, h3);
                            half4_t o = { f2h(h0), f2h(h1), f2h(h2), f2h(h3) };
                            o = o * sc;
                            { union { half4_t h; uint32_t u[2]; } ob; ob.h = o; ob.u[0] |= poison; ob.u[1] |= poison; o = ob.h; }
                            if (!hi) ((half4_t PS_GLOBAL*) (logits_p + (size_t) cbl * 128))[l] = o;
                            return;
                        }
                        const uint32_t loff = ((uint32_t) (tl.cb0 + j) * (uint32_t) S_op + (uint32_t) (line0 + tl.slice)) * PS_PLINE_BYTES;
                        if (!(tp_sys && out_type == PS_OUT_ATOMIC)) { if (!hi) ps_pl_store(rsl, loff, l, v, tag_out); }
                        else
                        {
                            // a row shard's partial row goes to EVERY rank (this one included): the consumers of all ranks then sum the same tp_world x S lines in the same
                            // order -- the all-reduce of model/model_tp_backend.py:119-126 without a launch, a flag or a fence (the tag is the flag)
                            const unsigned long long PS_CONST* const peers = *((const unsigned long long PS_CONST* const PS_CONST*) (a.err + 4));
                            for (int pr = 0; pr < tpw; ++pr)
                            {
                                const ps_rsrc_t rp = ps_rsrc((const char*) peers[pr] + O->xoff);
                                if (!hi) ps_pl_store_sys(rp, loff, l, v, tag_out);
                            }
                        }
                    };
                    if (W == 1)
                    {
                        float4_t t[6];
                        #pragma unroll
                        for (int i = 0; i < 6; ++i) t[i] = ((const float4_t*) (part + (size_t) (6 * hi + i) * 256))[l];
                        float4_t v = t[0];
                        #pragma unroll
                        for (int i = 1; i < 6; ++i) v += t[i];             // (vector adds: v_pk_add_f32)
                        if (sw == 0) PS_T(10);
                        publish(v, 0);
                    }
                    else
                    {
                        for (int j = sw; j < W; j += PS_NSV)
                        {
                            const uint32_t ct = j == sw ? ctab0 : (j == sw + PS_NSV ? ctab1 : col_table(j));      // (scalar; the first two formed under the streaming, above)
                            const uint32_t inh = ((ct >> 8) & 0xffu) >> (4 * hi), sgh = (ct >> 16) >> (4 * hi);
                            const int wb = (int) (ct & 0xffu) + 4 * hi;
                            float4_t t[4];
                            #pragma unroll
                            for (int c = 0; c < 4; ++c)
                                t[c] = ((const float4_t*) (part + (size_t) min(wb + c, PS_SW - 1) * 256 + (size_t) ((sgh >> c) & 1u) * 128))[l];
                            float4_t v = { 0.f, 0.f, 0.f, 0.f };
                            #pragma unroll
                            for (int c = 0; c < 4; ++c)
                            {
                                const uint32_t mk = (uint32_t) (-(int32_t) ((inh >> c) & 1u));
                                v.x += __uint_as_float(__float_as_uint(t[c].x) & mk); v.y += __uint_as_float(__float_as_uint(t[c].y) & mk);
                                v.z += __uint_as_float(__float_as_uint(t[c].z) & mk); v.w += __uint_as_float(__float_as_uint(t[c].w) & mk);
                            }
                            if (sw == 0 && j == 0) PS_T(10);
                            publish(v, j);
                        }
                    }
                }
            }
#else
            for (int j = shw; j < W; j += 2 * PS_NSV)
            {
                const int l = l32;
                float4_t v = { 0.f, 0.f, 0.f, 0.f };
#ifdef PS_SUM_BRANCHY
                #pragma unroll
                for (int w = 0; w < PS_SW; ++w)
                {
                    const int sj = seginfo[w * 4], s0 = seginfo[w * 4 + 1], s1 = seginfo[w * 4 + 2];
                    const float4_t t0 = ((const float4_t*) (part + (size_t) w * 256))[l], t1 = ((const float4_t*) (part + (size_t) w * 256 + 128))[l];
                    if (s0 > 0 && sj == j) { v.x += t0.x; v.y += t0.y; v.z += t0.z; v.w += t0.w; }
                    if (s1 > 0 && sj + 1 == j) { v.x += t1.x; v.y += t1.y; v.z += t1.z; v.w += t1.w; }
                }
#else
                if (W == 1)
                {
                    // one column block (the usual rectangle of q|k|v, o, down): every wave's segment-0 row counts if the wave has a unit at all -- which follows
                    // from the rectangle's size alone (scalar arithmetic): no records, twelve independent LDS reads, one round trip
                    const int T1 = 4 * nb;
                    #pragma unroll
                    for (int w0 = 0; w0 < PS_SW; w0 += 6)                  // (two rounds of six: twelve at once cost the kernel its last free registers)
                    {
                        float4_t t[6];
                        #pragma unroll
                        for (int i = 0; i < 6; ++i) t[i] = ((const float4_t*) (part + (size_t) (w0 + i) * 256))[l];
                        #pragma unroll
                        for (int i = 0; i < 6; ++i)
                        {
                            const int w = w0 + i;
                            const uint32_t mk = ((T1 * (w + 1)) / PS_SW - (T1 * w) / PS_SW) > 0 ? 0xffffffffu : 0u;
                            v.x += __uint_as_float(__float_as_uint(t[i].x) & mk); v.y += __uint_as_float(__float_as_uint(t[i].y) & mk);
                            v.z += __uint_as_float(__float_as_uint(t[i].z) & mk); v.w += __uint_as_float(__float_as_uint(t[i].w) & mk);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                else
                {
                    // the waves whose run touches column block j follow from the partition (ps_make_seg: wave w takes units [T w / 12, T (w + 1) / 12) of the rectangle's
                    // T = 4 nb W units, column-major): at most eight consecutive waves from one before floor(12 j / W) on (checked for every nb <= 32, W <= 12); a wave whose
                    // run STARTS in the column has its row in segment 0, one that started in the column before in segment 1 (a run spans at most two columns).  ONE round of
                    // eight masked LDS reads -- it was three dependent rounds of records + both rows of all twelve waves (1.4 us by the stamps; 0.8 for one column block)
                    const int H = 4 * nb, T = H * W;
                    const int lo_u = j * H, hi_u = lo_u + H;
                    const int w_first = max((PS_SW * lo_u) / T - 1, 0);
                    float4_t t[8]; uint32_t mk[8];
                    #pragma unroll
                    for (int i = 0; i < 8; ++i)
                    {
                        const int w = min(w_first + i, PS_SW - 1);
                        const int u0 = (T * w) / PS_SW, u1 = (T * (w + 1)) / PS_SW;
                        const bool in_col = (w_first + i < PS_SW) && u1 > u0 && u0 < hi_u && u1 > lo_u;
                        t[i] = ((const float4_t*) (part + (size_t) w * 256 + (u0 >= lo_u ? 0 : 128)))[l];
                        mk[i] = in_col ? 0xffffffffu : 0u;
                    }
                    #pragma unroll
                    for (int i = 0; i < 8; ++i)
                    {
                        v.x += __uint_as_float(__float_as_uint(t[i].x) & mk[i]); v.y += __uint_as_float(__float_as_uint(t[i].y) & mk[i]);
                        v.z += __uint_as_float(__float_as_uint(t[i].z) & mk[i]); v.w += __uint_as_float(__float_as_uint(t[i].w) & mk[i]);
                    }
                }
#endif
                float xs = 0.0f;
                for (int q0 = 0; q0 < nb; q0 += 32) if (q0 + l < nb) xs += bsum[q0 + l];
                #pragma unroll
                for (int i = 1; i < 32; i <<= 1) xs += xor_lane(xs, i);
                if (sw == 0 && j == shw) PS_T(10);
                const float kinv = CB == EXL3_CB_MUL1 ? (float) u16_as_half(0x1eeeu) : 1.0f, kbias = CB == EXL3_CB_MUL1 ? (float) u16_as_half(0xc931u) : 0.0f;
                const float bb = kbias * xs;
                v.x = (v.x * kinv + bb) * fac_norm; v.y = (v.y * kinv + bb) * fac_norm; v.z = (v.z * kinv + bb) * fac_norm; v.w = (v.w * kinv + bb) * fac_norm;
                const int cbl = tl.cb0 + j;
                if (out_type != PS_OUT_FINAL)
                {
                    // the slice's partial row of column block cbl as one tagged line (both op kinds: no atomics, no drain, no edge)
                    const ps_rsrc_t rsl = ps_rsrc(slab_p);
                    ps_pl_store(rsl, ((uint32_t) cbl * (uint32_t) S_op + (uint32_t) tl.slice) * PS_PLINE_BYTES, l, v, tag_out);
                }
                else
                {
                    float h0, h1, h2, h3;
                    out_had(v, l, h0, h1, h2, h3);
                    const half4_t sc = j == shw ? scp[0] : scp[1];
                    half4_t o = { f2h(h0), f2h(h1), f2h(h2), f2h(h3) };
                    o = o * sc;
                    { union { half4_t h; uint32_t u[2]; } ob; ob.h = o; ob.u[0] |= poison; ob.u[1] |= poison; o = ob.h; }      // (a wait timed out: NaN, never a plausible row)
                    ((half4_t PS_GLOBAL*) (logits_p + (size_t) cbl * 128))[l] = o;
                }
            }
#endif
            if (sw == 0) PS_T(11);
            r_last = r_next;
#ifndef PS_ABL_NO_COLD
            if (out_type == PS_OUT_ATOMIC && !out_direct && active && tl.slice == 0)
            {
                // OWNERS of the residual row's blocks cb0 .. cb0 + W - 1 (the slice-0 workgroup of the column group).  All eight service half-waves gather: half-wave h
                // takes the partial lines s = h, h + 8, ... of every owned block (tagged: re-loaded until complete; one round of loads), leaves its sum in LDS, and
                // the half-wave that owns the block adds the eight sums in order h = 0..7, applies the output Hadamard and svh, adds the block of the previous
                // row version and publishes the new version's block as ONE tagged line -- every workgroup's next RMSNorm reads it.  Replaces integer atomics
                // into R + drain + arrival counter + poll (6.5-7 us from "streaming done" to "next op has the row" by the phase stamps; now ~4).
                const ps_rsrc_t rsl = ps_rsrc(slab_p);
                const int rv = O->rver;                                       // the version this op produces (>= 1)
                for (int jj = 0; jj < W; ++jj)
                {
                    const int cbl = tl.cb0 + jj;
                    float4_t ys = { 0.f, 0.f, 0.f, 0.f };
                    if (shw < S_op)
                    {
                        const int nl = (S_op - shw + 7) >> 3;                  // lines shw, shw + 8, ...: at most 4 (S <= 32)
                        for (int spins = 0;; ++spins)
                        {
                            bool ok = true;
                            PsPl t[4];
                            #pragma unroll
                            for (int i = 0; i < 4; ++i)
                            {
                                const uint32_t lo_ = ((uint32_t) cbl * (uint32_t) S_op + (uint32_t) (shw + 8 * min(i, nl - 1))) * PS_PLINE_BYTES;
                                t[i] = ps_pl_load_t<TP>(rsl, lo_, l32);
                            }
                            ys = float4_t{ 0.f, 0.f, 0.f, 0.f };
                            #pragma unroll
                            for (int i = 0; i < 4; ++i) if (i < nl)
                            {
                                ok &= ps_pl_ok(t[i], tag_out);
                                const float4_t x = ps_pl_val(t[i]);
                                ys.x += x.x; ys.y += x.y; ys.z += x.z; ys.w += x.w;
                            }
                            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                            if (spins > slim) { ps_timeout(8u); break; }
                            __builtin_amdgcn_s_sleep(PS_POLL_SLEEP);
                        }
                    }
                    ((float4_t*) (gath + ((size_t) (jj & 3) * 8 + shw) * 128))[l32] = ys;
                }
                tgt_o += PS_NSV;
                c_inc(PS_C_O);
                for (int jj = shw; jj < W; jj += 2 * PS_NSV)
                {
                    const int cbl = tl.cb0 + jj, l = l32;
                    float4_t rold;
                    if (rv == 1)
                    {
                        const ps_rsrc_t r0 = ps_rsrc(a.R);
                        const uint32_t ro = (uint32_t) cbl * 1024u + (uint32_t) l * 32u;
                        const uint4_t ra = ps_ld128(r0, ro), rb = ps_ld128(r0, ro + 16u);
                        rold = float4_t{ fx_to_float(ra.x, ra.y), fx_to_float(ra.z, ra.w), fx_to_float(rb.x, rb.y), fx_to_float(rb.z, rb.w) };
                    }
                    else
                    {
                        // (tagged by whoever published version rv - 1: the owners of op - 2, or -- DIRECT -- the publishing workgroups of op - 1; the same tag)
                        const ps_rsrc_t r0 = ps_rsrc(a.rbuf + (size_t) ((rv - 1) & 1) * PS_MAX_ROW_BLOCKS * 128);
                        const uint32_t ro = (uint32_t) cbl * PS_LINE_BYTES + (uint32_t) l * 16u, tag_old = (epoch << 12) | (uint32_t) (op - 1);
                        for (int spins = 0;; ++spins)
                        {
                            const uint4_t ra = ps_ld128(r0, ro), rb = ps_ld128(r0, ro + 512u);
                            rold = float4_t{ __uint_as_float(ra.x), __uint_as_float(ra.z), __uint_as_float(rb.x), __uint_as_float(rb.z) };
                            const bool ok = (ra.y == tag_old) & (ra.w == tag_old) & (rb.y == tag_old) & (rb.w == tag_old);
                            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                            if (spins > slim) { ps_timeout(1u); break; }
                            __builtin_amdgcn_s_sleep(PS_POLL_SLEEP);
                        }
                    }
                    c_spin(PS_C_O, tgt_o);
                    float4_t ys = { 0.f, 0.f, 0.f, 0.f };
                    #pragma unroll
                    for (int h = 0; h < 8; ++h)
                    {
                        const float4_t t = ((const float4_t*) (gath + ((size_t) (jj & 3) * 8 + h) * 128))[l];
                        ys.x += t.x; ys.y += t.y; ys.z += t.z; ys.w += t.w;
                    }
                    float h0, h1, h2, h3;
                    out_had(ys, l, h0, h1, h2, h3);
                    const half4_t sc = jj == shw ? scp[0] : scp[1];
                    const float n0 = rold.x + h0 * (float) sc.x, n1 = rold.y + h1 * (float) sc.y, n2 = rold.z + h2 * (float) sc.z, n3 = rold.w + h3 * (float) sc.w;
                    if (sw != 0) c_spin(PS_C_G, (uint32_t) (op + 1));
                    const ps_rsrc_t rn = ps_rsrc(a.rbuf + (size_t) (rv & 1) * PS_MAX_ROW_BLOCKS * 128);
                    const uint32_t no = (uint32_t) cbl * PS_LINE_BYTES + (uint32_t) l * 16u;
                    ps_st128(rn, no, uint4_t{ __float_as_uint(n0), tag_out, __float_as_uint(n1), tag_out });
                    ps_st128(rn, no + 512, uint4_t{ __float_as_uint(n2), tag_out, __float_as_uint(n3), tag_out });
                }
            }
#endif
            if (sw == 0) PS_T(12);
            if (sw == 0) PS_T(7);

            if (in_type == PS_IN_QKV && !in_attn && tl.side >= 0 && sw == PS_NSV - 1)
            {
                // side job: one (K | V, 128-value block) of the new token: finished like q (RoPE on K only) and appended to the 4-bit paged cache
                // (the arithmetic of glue_qkv_kernel: qkv_block_finish + kv_quant_regs); both half-waves compute it, the upper one stores.  After the
                // workgroup's arrival: nothing in the launch reads the cache, and the k / v slab lines stay valid until the next layer's q|k|v op,
                // which is two edges away
                const int kvb = O->kvb, tsk = tl.side, isv = tsk >= kvb ? 1 : 0, hb = tsk - isv * kvb;
                const half4_t sc = ps_g((const half4_t*) (O->in_svh[1 + isv] + (size_t) hb * 128))[l32];
                const ps_rsrc_t rk = ps_rsrc(O->in_slab[1 + isv]);
                float4_t ys;
                for (int spins = 0;; ++spins)
                {
                    bool ok = true;
                    ys = ps_slab_sum<PS_SLAB_NB>(rk, (uint32_t) hb * (uint32_t) O->S_in * PS_PLINE_BYTES, O->S_in, l32, tag_in, ok);
                    if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                    if (spins > slim) { ps_timeout(2u); break; }
                    __builtin_amdgcn_s_sleep(PS_POLL_SLEEP);
                }
                const GemvRescale rs0 = { nullptr, nullptr, 0, 0.0f };
                const int ph = O->hd >> 3;
                float4_t ksn = { 0.f, 0.f, 0.f, 0.f }, kcs = ksn;
                if (O->rope_mode == 2)
                {
                    const int f = 4 * (l32 & (ph - 1));
                    ksn = *ps_g((const float4_t*) (rope_sin_p + f)); kcs = *ps_g((const float4_t*) (rope_cos_p + f));
                }
                else
                {
                    const int f = 2 * (l32 & ((O->hd >> 2) - 1));
                    ksn.x = ps_g(rope_sin_p)[f]; ksn.y = ps_g(rope_sin_p)[f + 1]; kcs.x = ps_g(rope_cos_p)[f]; kcs.y = ps_g(rope_cos_p)[f + 1];
                }
                const half4_t y = qkv_block_finish(ys, sc, rs0, 0, l32, 0.0f, 0.0f, !isv, O->rope_mode, ph, ksn, kcs);
                const int64_t token_pos = a.slots[0];
                const int64_t gb = token_pos * (kvb * 4) + hb * 4 + (l32 >> 3);
                uint32_t* cw = isv ? O->v_cache : O->k_cache; half_t* csc = isv ? O->v_scales : O->k_scales;
                kv_quant_regs<4>((float) y.x, (float) y.y, (float) y.z, (float) y.w, cw + gb * 4, csc + gb, (lane >> 5) == 1, lane);
            }
        }
        if (cu == 0 && sw == 0)
        {
            if (lane == 0) *a.epoch = epoch + 1u;
            if constexpr (ATT)
            {
                // the tags keep 20 bits of the run epoch.  Every line of the step is re-written in every run EXCEPT the attention records / statistics of splits that are
                // not in use at the current length: left alone, one written 2^20 runs ago would carry the current tag again.  The run that wraps the 20 bits
                // clears them (tag 0 = no producer): once per 2^20 runs, after this workgroup's last op -- nothing in the launch touches them behind o_proj of the last layer.
                if (((epoch + 1u) & 0xfffffu) == 0u && nops >= 2)
                {
                    const PsAtt PS_CONST* const AT = (const PsAtt PS_CONST*) &ops_c[1].mat[1];
                    const ps_rsrc_t rrec = ps_rsrc(AT->rec), rst = ps_rsrc(AT->stats);
                    const uint4_t z = { 0u, 0u, 0u, 0u };
                    const int n_st = AT->hq * PS_ATT_MAX_SPLITS, n_rec = AT->hq * AT->nsplit * 32;
                    for (int i = lane; i < n_st; i += 64) ps_st128(rst, (uint32_t) i * 16u, z);
                    for (int i = lane; i < n_rec; i += 64) ps_st128(rrec, (uint32_t) i * 16u, z);
                }
            }
        }
    }
    #undef PS_T
    #undef PS_TC
    #undef PS_TP
}
