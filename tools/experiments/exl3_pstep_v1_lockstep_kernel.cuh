// Persistent decode-step kernel body (generation 5), included by exl3_pstep.hip.  Design: exl3_pstep.cuh.  K = bits per weight, mul1 codebook, one row.
//
// Per op, per workgroup (one per CU, 16 waves):
//   [static operands of the preparation requested]  ->  wave 0 polls the previous op's edge  ->  B1
//   preparation tasks (half-wave per 128-value block of the workgroup's k-slice: RMSNorm | q finish + RoPE | silu * mul, then the input Hadamard) -> LDS quads -> B3
//   every wave: its decode-ahead units (MFMA only), then the rest of its run of work units (generation 4's unit: exl3_gemv4.kspec.hip g4_unit); the LAST
//   unit's ring refill already requests the wave's first rows of the NEXT op  ->  partial rows to LDS  ->  B4
//   half-wave j finishes column block j of the rectangle (sum of the waves' partials, mul1 affine map, slab line | output Hadamard + atomics | final row);
//   the last finishing wave announces the workgroup's arrival at this op's edge
//   every wave: decode-ahead of its first unit(s) of the next op (registers; the second unit in LDS) -- this is what fills the edge's wait
#pragma once
#include "exl3_pstep.cuh"
#include "exl3_api_internal.h"
#include "exl3_gemv_args.h"
#include "exl3_lane_decode.cuh"
#include "exl3_glue_device.cuh"
#include <type_traits>

template <int I, int N, typename F>
__device__ __forceinline__ void ps_static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); ps_static_for<I + 1, N>(f); }
}

// agent-scope (sc1) accesses: everything one workgroup writes for another inside the launch goes through these (exl3_gemv2_tail.cuh has the same pair)
__device__ __forceinline__ unsigned long long ps_ld64(const void* p) { return __hip_atomic_load((unsigned long long*) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ps_st64(void* p, unsigned long long v) { __hip_atomic_store((unsigned long long*) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float4_t ps_ld_f4(const float* p)
{
    union { unsigned long long u[2]; float4_t f; } c;
    c.u[0] = ps_ld64(p); c.u[1] = ps_ld64(p + 2);
    return c.f;
}
__device__ __forceinline__ void ps_st_f4(float* p, float4_t v)
{
    union { unsigned long long u[2]; float4_t f; } c; c.f = v;
    ps_st64(p, c.u[0]); ps_st64(p + 2, c.u[1]);
}

// slab lines of one column block summed in slice order from zero (slab_sum of exl3_glue_device.cuh with agent-scope loads); every load is issued before the first add
template <int NB>
__device__ __forceinline__ float4_t ps_slab_sum(const float* base, int S, int l)
{
    float4_t v = { 0.f, 0.f, 0.f, 0.f };
    for (int s = 0; s < S; s += NB)
    {
        float4_t t[NB];
        #pragma unroll
        for (int i = 0; i < NB; ++i) t[i] = ps_ld_f4(base + (size_t) min(s + i, S - 1) * 128 + 4 * l);
        #pragma unroll
        for (int i = 0; i < NB; ++i) if (s + i < S) { v.x += t[i].x; v.y += t[i].y; v.z += t[i].z; v.w += t[i].w; }
    }
    return v;
}

// one work unit = 2 tile rows of the wave's 128-column block against the activation group `ag` (exl3_gemv4.kspec.hip g4_unit, mul1 FAST variant)
template <int K, int HALF>
__device__ __forceinline__ void ps_unit(LaneWords<K> (&ring)[2], const uint32_t* __restrict__ refill, size_t refill_rs, int lane, int lofs, half4_t ag,
                                        float4_t& acc_c, float4_t& acc_d)
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
        load_lane_words<K>(ring[u], refill + (size_t) u * refill_rs + lofs);
        ps_static_for<0, 4>([&] (auto qc)
        {
            constexpr int q = decltype(qc)::value;
            constexpr int ABID = 8 * HALF + 4 * u + q;
            half4_t bc[2], bd[2];
            decode_quad<K, EXL3_CB_MUL1, 1, 8 * q>(Wx, bc);
            decode_quad<K, EXL3_CB_MUL1, 1, 8 * q + 4>(Wx, bd);
            acc_c = __builtin_amdgcn_mfma_f32_4x4x4f16(ag, bc[0], acc_c, 4, ABID, 0);
            acc_d = __builtin_amdgcn_mfma_f32_4x4x4f16(ag, bd[0], acc_d, 4, ABID, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    });
}

// decode-ahead: the same unit decoded into 16 B operands (no activations needed), the ring refilled as a streamed unit refills it
template <int K>
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
        load_lane_words<K>(ring[u], refill + (size_t) u * refill_rs + lofs);
        ps_static_for<0, 4>([&] (auto qc)
        {
            constexpr int q = decltype(qc)::value;
            half4_t bc[2], bd[2];
            decode_quad<K, EXL3_CB_MUL1, 1, 8 * q>(Wx, bc);
            decode_quad<K, EXL3_CB_MUL1, 1, 8 * q + 4>(Wx, bd);
            dec[(4 * u + q) * 2] = bc[0]; dec[(4 * u + q) * 2 + 1] = bd[0];
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

// a wave's run of work units inside its workgroup's rectangle: column-major over (column block j, unit i); at most two column blocks (planner: ncb <= 16)
template <int K>
struct PsSeg
{
    const uint32_t* stripA; const uint32_t* stripB; size_t rs;      // wave-uniform pointers (lane 0's words) of the first unit of segment 0 / 1; words per tile row
    int j0, i0, len0, len1, n;
    __device__ __forceinline__ const uint32_t* unit_ptr(int q) const       // 0 <= q < n
    {
        return q < len0 ? stripA + (size_t) (2 * q) * rs : stripB + (size_t) (2 * (q - len0)) * rs;
    }
};

template <int K>
__device__ __forceinline__ PsSeg<K> ps_make_seg(const PsOp* __restrict__ O, const PsTile& t, int wave)
{
    constexpr int NW = 8 * K;
    PsSeg<K> s;
    s.stripA = nullptr; s.stripB = nullptr; s.rs = 0; s.j0 = 0; s.i0 = 0; s.len0 = 0; s.len1 = 0; s.n = 0;
    if (t.mat >= 0)
    {
        const int H = 4 * t.nb, T = H * t.ncb;
        const int u0 = (T * wave) >> 4, u1 = (T * (wave + 1)) >> 4;
        s.n = u1 - u0;
        s.j0 = u0 / H; s.i0 = u0 - s.j0 * H;
        s.len0 = min(s.n, H - s.i0); s.len1 = s.n - s.len0;
        const PsMat* M = &O->mat[t.mat];
        const uint32_t* B = M->B; const int tn = M->tiles_n;
        s.rs = (size_t) tn * NW;
        s.stripA = B + ((size_t) (t.b0 * 8 + 2 * s.i0) * tn + (size_t) (t.cb0 + s.j0) * 8) * NW;
        s.stripB = B + ((size_t) (t.b0 * 8) * tn + (size_t) (t.cb0 + s.j0 + 1) * 8) * NW;
    }
    return s;
}

template <int K>
__global__ __launch_bounds__(PS_NT) void exl3_pstep_kernel(const PsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const quads = smem;
    float* const ssblk = (float*) (smem + PS_QUADS_BYTES);
    float* const bsum = ssblk + 64;
    int* const seginfo2 = (int*) (bsum + 64);                     // [2 (op parity)][16][4]: j0, len0, len1 of every wave's run
    uint32_t* const lctl = (uint32_t*) (seginfo2 + 128);            // [0]: finished reducer waves of the current op
    float* const part = (float*) (smem + PS_QUADS_BYTES + PS_MISC_BYTES);
    char* const pdec = smem + PS_QUADS_BYTES + PS_MISC_BYTES + PS_PART_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, hwid = tid >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cu = blockIdx.x, ncu = a.ncu, nops = a.nops;
    const int quad_lane = (lane >> 2) * 8, lofs = lane * K;
    const int pmax = a.pmax;
    unsigned long long* const dbg = a.dbg;
    #define PS_T(i) do { if (dbg && tid == 0) dbg[((size_t) op * ncu + cu) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)

    if (tid == 0) lctl[0] = 0u;

    LaneWords<K> ring[2];
    half4_t dec0[16];
    #pragma unroll
    for (int i = 0; i < 16; ++i) dec0[i] = half4_t{ 0, 0, 0, 0 };
    #pragma unroll
    for (int i = 0; i < K; ++i) { ring[0].w[i] = 0u; ring[1].w[i] = 0u; }
    char* const pdec_w = pdec + (size_t) wave * (16 * 64 * 8) + (size_t) lane * 8;

    // decode-ahead of the wave's first unit(s) of an op: returns the number of units decoded (unit 0 -> dec0, unit 1 -> LDS)
    auto decode_ahead = [&] (const PsSeg<K>& s, bool ring_loaded) -> int
    {
        if (s.n <= 0) return 0;
        if (!ring_loaded)
        {
            load_lane_words<K>(ring[0], s.stripA + lofs);
            load_lane_words<K>(ring[1], s.stripA + s.rs + lofs);
        }
        const int P = min(pmax, s.len0);
        if (P >= 1) ps_predecode<K>(ring, s.unit_ptr(min(1, s.n - 1)), s.rs, lane, lofs, dec0);
        if (P >= 2)
        {
            half4_t tmp[16];
            ps_predecode<K>(ring, s.unit_ptr(min(2, s.n - 1)), s.rs, lane, lofs, tmp);
            #pragma unroll
            for (int i = 0; i < 16; ++i) *((half4_t*) (pdec_w + i * 512)) = tmp[i];
        }
        return P;
    };

    PsTile tl = a.tiles[cu];
    PsSeg<K> cur = ps_make_seg<K>(a.ops, tl, wave);
    int P = decode_ahead(cur, false);
    bool aborted = false;
    __syncthreads();

    for (int op = 0; op < nops; ++op)
    {
        const PsOp* __restrict__ O = a.ops + op;
        const bool active = tl.mat >= 0;
        const int b0 = tl.b0, nb = active ? tl.nb : 0, W = active ? tl.ncb : 0;
        const int in_type = O->in_type, out_type = O->out_type, kk = O->k;
        PS_T(0);

        // ---- the next op's rectangle and this wave's run in it (pointers only): the last streamed unit of this op requests its first rows
        PsTile tn; tn.mat = -1; tn.cb0 = 0; tn.ncb = 0; tn.b0 = 0; tn.nb = 0; tn.slice = 0; tn.side = -1; tn.flags = 0;
        if (op + 1 < nops) tn = a.tiles[(size_t) (op + 1) * ncu + cu];
        const PsSeg<K> nxt = ps_make_seg<K>(O + 1, tn, wave);
        const uint32_t* const after_all = nxt.n > 0 ? nxt.stripA : (cur.n > 0 ? cur.unit_ptr(cur.n - 1) : nullptr);
        const size_t after_rs = nxt.n > 0 ? nxt.rs : cur.rs;

        // ---- static operands of the preparation tasks (weights: requested before the wait)
        half4_t wv[2], sv[2], sva = { 0, 0, 0, 0 }, svb = sva;
        float4_t sn4 = { 0.f, 0.f, 0.f, 0.f }, cs4 = sn4;
        wv[0] = sva; wv[1] = sva; sv[0] = sva; sv[1] = sva;
        const half_t* const suh_m = active ? O->mat[tl.mat].suh : nullptr;
        const int nblk = kk >> 7;
        const int tb = min(hwid, max(nb - 1, 0));                 // QKV / ACT: this half-wave's slice-local block (clamped)
        if (active)
        {
            if (in_type == PS_IN_NORM)
            {
                #pragma unroll
                for (int it = 0; it < 2; ++it)
                {
                    const int blk = min(hwid + 32 * it, nblk - 1);
                    wv[it] = ((const half4_t*) (O->norm_w + (size_t) blk * 128))[l32];
                    sv[it] = ((const half4_t*) (suh_m + (size_t) blk * 128))[l32];
                }
            }
            else
            {
                const int blk = b0 + tb;
                sv[0] = ((const half4_t*) (suh_m + (size_t) blk * 128))[l32];
                sva = ((const half4_t*) (O->in_svh[0] + (size_t) blk * 128))[l32];
                if (in_type == PS_IN_ACT) svb = ((const half4_t*) (O->in_svh[1] + (size_t) blk * 128))[l32];
                else
                {
                    const int ph = O->hd >> 3;
                    if (O->rope_mode == 2)
                    {
                        const int f = 4 * (l32 & (ph - 1));
                        sn4 = *((const float4_t*) (a.rope_sin + f)); cs4 = *((const float4_t*) (a.rope_cos + f));
                    }
                    else
                    {
                        const int f = 2 * (l32 & ((O->hd >> 2) - 1));
                        sn4.x = a.rope_sin[f]; sn4.y = a.rope_sin[f + 1]; cs4.x = a.rope_cos[f]; cs4.y = a.rope_cos[f + 1];
                    }
                }
            }
        }
        int* const seginfo = seginfo2 + (op & 1) * 64;
        if (lane == 0) { int* si = seginfo + wave * 4; si[0] = cur.j0; si[1] = cur.len0; si[2] = cur.len1; }

        // ---- the edge: every workgroup has finished (and drained) the previous op
        if (wave == 0 && op > 0 && !aborted)
        {
            const uint32_t* c = a.cnt + ((size_t) (op - 1) * 8 + (lane & 7)) * 16;
            const uint32_t expect = (uint32_t) ((ncu - (lane & 7) + 7) >> 3);
            int spins = 0;
            for (;;)
            {
                const uint32_t v = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__builtin_amdgcn_ballot_w64(v < expect) == 0ull) break;
                if (++spins > a.spin_limit)
                {
                    aborted = true;
                    if (lane == 0) __hip_atomic_fetch_or(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();                                                                 // B1
        PS_T(1);

        // ---- preparation tasks -> activation quads (+ block sums of the rotated activations for the mul1 affine term)
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
        if (in_type == PS_IN_NORM)
        {
            // exact RMSNorm: every workgroup reads the whole row (block sums of squares in rms_norm's order: norm.cu:20-120), then rotates its slice
            half4_t xr[2];
            #pragma unroll
            for (int it = 0; it < 2; ++it)
            {
                const int blk = hwid + 32 * it;
                xr[it] = half4_t{ 0, 0, 0, 0 };
                if (32 * it < nblk)
                {
                    const unsigned long long* rp = a.R + (size_t) min(blk, nblk - 1) * 128 + 4 * l32;
                    const unsigned long long r0 = ps_ld64(rp), r1 = ps_ld64(rp + 1), r2 = ps_ld64(rp + 2), r3 = ps_ld64(rp + 3);
                    auto fx = [] (unsigned long long v) -> half_t { return f2h(fx_to_float((uint32_t) v, (uint32_t) (v >> 32))); };
                    xr[it] = half4_t{ fx(r0), fx(r1), fx(r2), fx(r3) };
                    const float f0 = (float) xr[it].x, f1 = (float) xr[it].y, f2 = (float) xr[it].z, f3 = (float) xr[it].w;
                    float ssq = f0 * f0;
                    ssq = __builtin_fmaf(f1, f1, ssq); ssq = __builtin_fmaf(f2, f2, ssq); ssq = __builtin_fmaf(f3, f3, ssq);
                    #pragma unroll
                    for (int i = 1; i < 32; i <<= 1) ssq += xor_lane(ssq, i);
                    if (blk < nblk && l32 == 0) ssblk[blk] = ssq;
                }
            }
            __syncthreads();                                                             // B2
            float s2 = l32 < nblk ? ssblk[l32] : 0.0f;
            #pragma unroll
            for (int i = 1; i < 32; i <<= 1) s2 += xor_lane(s2, i);
            if (nblk > 32)
            {
                float v = (32 + l32 < nblk) ? ssblk[32 + l32] : 0.0f;
                #pragma unroll
                for (int i = 1; i < 32; i <<= 1) v += xor_lane(v, i);
                s2 += v;
            }
            const float r = __frsqrt_rn(s2 / (float) kk + O->eps);
            #pragma unroll
            for (int it = 0; it < 2; ++it)
            {
                const int blk = hwid + 32 * it;
                if (32 * it < nblk)
                {
                    const bool act = active && blk >= b0 && blk < b0 + nb;
                    const half4_t xv = { f2h((float) xr[it].x * (float) wv[it].x * r), f2h((float) xr[it].y * (float) wv[it].y * r),
                                         f2h((float) xr[it].z * (float) wv[it].z * r), f2h((float) xr[it].w * (float) wv[it].w * r) };
                    rotate_store(xv, sv[it], min(max(blk - b0, 0), max(nb - 1, 0)), act);
                }
            }
        }
        else if (in_type == PS_IN_QKV)
        {
            // q block (b0 + tb) finished from the q|k|v op's slabs exactly as exl3_glue_qkv_tab finishes it (qkv_block_finish), then o_proj's input rotation
            if (active && 2 * wave < nb)                                    // wave-uniform: this wave owns at least one task
            {
                const bool act = hwid < nb;
                const int blk = b0 + tb;
                const float4_t ys = ps_slab_sum<8>(O->in_slab[0] + (size_t) blk * O->S_in * 128, O->S_in, l32);
                const GemvRescale rs0 = { nullptr, nullptr, 0, 0.0f };
                const half4_t xq = qkv_block_finish(ys, sva, rs0, 0, l32, 0.0f, 0.0f, true, O->rope_mode, O->hd >> 3, sn4, cs4);
                if (act && (tl.flags & PS_TILE_Q_OUT) && a.q_out) ((half4_t*) (a.q_out + (size_t) blk * 128))[l32] = xq;
                rotate_store(xq, sv[0], tb, act);
            }
            if (tl.side >= 0 && wave == PS_WAVES - 1)
            {
                // side job: one (K | V, 128-value block) of the new token: finished like q (RoPE on K only) and appended to the 4-bit paged cache
                // (the arithmetic of glue_qkv_kernel: qkv_block_finish + kv_quant_regs); both half-waves compute it, the upper one stores
                const int kvb = O->kvb, tsk = tl.side, isv = tsk >= kvb ? 1 : 0, hb = tsk - isv * kvb;
                const half4_t sc = ((const half4_t*) (O->in_svh[1 + isv] + (size_t) hb * 128))[l32];
                const float4_t ys = ps_slab_sum<8>(O->in_slab[1 + isv] + (size_t) hb * O->S_in * 128, O->S_in, l32);
                const GemvRescale rs0 = { nullptr, nullptr, 0, 0.0f };
                const int ph = O->hd >> 3;
                float4_t ksn = { 0.f, 0.f, 0.f, 0.f }, kcs = ksn;
                if (O->rope_mode == 2)
                {
                    const int f = 4 * (l32 & (ph - 1));
                    ksn = *((const float4_t*) (a.rope_sin + f)); kcs = *((const float4_t*) (a.rope_cos + f));
                }
                else
                {
                    const int f = 2 * (l32 & ((O->hd >> 2) - 1));
                    ksn.x = a.rope_sin[f]; ksn.y = a.rope_sin[f + 1]; kcs.x = a.rope_cos[f]; kcs.y = a.rope_cos[f + 1];
                }
                const half4_t y = qkv_block_finish(ys, sc, rs0, 0, l32, 0.0f, 0.0f, !isv, O->rope_mode, ph, ksn, kcs);
                const int64_t token_pos = a.slots[0];
                const int64_t gb = token_pos * (kvb * 4) + hb * 4 + (l32 >> 3);
                uint32_t* cw = isv ? O->v_cache : O->k_cache; half_t* csc = isv ? O->v_scales : O->k_scales;
                kv_quant_regs<4>((float) y.x, (float) y.y, (float) y.z, (float) y.w, cw + gb * 4, csc + gb, hwid == 2 * PS_WAVES - 1, lane);
            }
        }
        else
        {
            // silu(g) * u of block (b0 + tb): slab lines in slice order, output Hadamards, svh -- the arithmetic of glue_act_kernel / generation 4's ACT tasks
            if (active && 2 * wave < nb)
            {
                const bool act = hwid < nb;
                const int blk = b0 + tb;
                const float4_t vg = ps_slab_sum<4>(O->in_slab[0] + (size_t) blk * O->S_in * 128, O->S_in, l32);
                const float4_t vu = ps_slab_sum<4>(O->in_slab[1] + (size_t) blk * O->S_in * 128, O->S_in, l32);
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
        __syncthreads();                                                                 // B3
        PS_T(2);

        // ---- this wave's run of work units: decode-ahead units first (MFMA only), the rest streamed
        auto run_seg = [&] (const uint32_t* strip, int len, int pre, const uint32_t* after, size_t after_rs, int qoff, float* pslot)
        {
            float4_t acc_c = { 0.f, 0.f, 0.f, 0.f }, acc_d = acc_c;
            const char* qb = quads + qoff + quad_lane;
            const size_t rs = cur.rs;
            auto up = [&] (int q) -> const uint32_t* { return q < len ? strip + (size_t) (2 * q) * rs : after; };
            auto ur = [&] (int q) -> size_t { return q < len ? rs : after_rs; };          // the refill target may lie in the NEXT op's matrix (another row pitch)
            int p = 0;
            if (pre > 0)
            {
                const uint2_t raw = *((const uint2_t*) qb);
                const half4_t ag = u2_as_half4(raw.x, raw.y);
                ps_consume<0>(dec0, ag, acc_c, acc_d);
                if (len > 1)
                {
                    if (pre > 1)
                    {
                        half4_t tmp[16];
                        #pragma unroll
                        for (int i = 0; i < 16; ++i) tmp[i] = *((const half4_t*) (pdec_w + i * 512));
                        ps_consume<1>(tmp, ag, acc_c, acc_d);
                    }
                    else ps_unit<K, 1>(ring, up(2), ur(2), lane, lofs, ag, acc_c, acc_d);
                }
                p = 2;
            }
            for (; p + 1 < len; p += 2)
            {
                const uint2_t raw = *((const uint2_t*) (qb + p * 64));
                const half4_t ag = u2_as_half4(raw.x, raw.y);
                ps_unit<K, 0>(ring, up(p + 1), ur(p + 1), lane, lofs, ag, acc_c, acc_d);
                ps_unit<K, 1>(ring, up(p + 2), ur(p + 2), lane, lofs, ag, acc_c, acc_d);
            }
            if (p < len)
            {
                const uint2_t raw = *((const uint2_t*) (qb + p * 64));
                const half4_t ag = u2_as_half4(raw.x, raw.y);
                ps_unit<K, 0>(ring, up(p + 1), ur(p + 1), lane, lofs, ag, acc_c, acc_d);
            }
            const int col = 16 * (lane >> 3) + (lane & 7);
            pslot[col] = acc_c[0]; pslot[col + 8] = acc_d[0];
        };
        if (cur.n > 0)
        {
            float* pw = part + (size_t) wave * 256;
            run_seg(cur.stripA, cur.len0, P, cur.len1 > 0 ? cur.stripB : after_all, cur.len1 > 0 ? cur.rs : after_rs, cur.i0 * 64, pw);
            if (cur.len1 > 0) run_seg(cur.stripB, cur.len1, 0, after_all, after_rs, 0, pw + 128);
        }
        const bool ring_has_next = cur.n > P && nxt.n > 0;
        PS_T(3);
        __syncthreads();                                                                 // B4
        PS_T(4);

        // ---- half-wave j finishes column block j of the rectangle
        const int nred = active ? (W + 1) >> 1 : 1;
        if (active && hwid < W)
        {
            const int j = hwid, l = l32;
            float4_t v = { 0.f, 0.f, 0.f, 0.f };
            for (int w = 0; w < PS_WAVES; ++w)
            {
                const int sj = seginfo[w * 4], s0 = seginfo[w * 4 + 1], s1 = seginfo[w * 4 + 2];
                if (s0 > 0 && sj == j) { const float4_t t = ((const float4_t*) (part + (size_t) w * 256))[l]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                if (s1 > 0 && sj + 1 == j) { const float4_t t = ((const float4_t*) (part + (size_t) w * 256 + 128))[l]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            }
            float xs = 0.0f;
            for (int q0 = 0; q0 < nb; q0 += 32) if (q0 + l < nb) xs += bsum[q0 + l];
            #pragma unroll
            for (int i = 1; i < 32; i <<= 1) xs += xor_lane(xs, i);
            const float kinv = (float) u16_as_half(0x1eeeu), kbias = (float) u16_as_half(0xc931u);
            const float bb = kbias * xs;
            v.x = v.x * kinv + bb; v.y = v.y * kinv + bb; v.z = v.z * kinv + bb; v.w = v.w * kinv + bb;
            const PsMat* M = &O->mat[tl.mat];
            const int cbl = tl.cb0 + j;
            if (out_type == PS_OUT_SLAB) ps_st_f4(M->slab + ((size_t) cbl * O->S + tl.slice) * 128 + 4 * l, v);
            else
            {
                float h0 = v.x, h1 = v.y, h2 = v.z, h3 = v.w;
                had128_f32x4(h0, h1, h2, h3, l);
                h0 *= HAD_R_SCALE_128; h1 *= HAD_R_SCALE_128; h2 *= HAD_R_SCALE_128; h3 *= HAD_R_SCALE_128;
                const half4_t sc = ((const half4_t*) (M->svh + (size_t) cbl * 128))[l];
                if (out_type == PS_OUT_ATOMIC)
                {
                    const float o[4] = { h0 * (float) sc.x, h1 * (float) sc.y, h2 * (float) sc.z, h3 * (float) sc.w };
                    unsigned long long* acc = a.R + (size_t) cbl * 128 + 4 * l;
                    #pragma unroll
                    for (int i = 0; i < 4; ++i) fx_atomic_add(acc + i, o[i]);
                }
                else
                {
                    half4_t o = { f2h(h0), f2h(h1), f2h(h2), f2h(h3) };
                    o = o * sc;
                    ((half4_t*) (a.logits + (size_t) cbl * 128))[l] = o;
                }
            }
        }
        if (wave < nred)
        {
            // this wave's output stores / atomics are acknowledged, then the last such wave announces the workgroup at the edge
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
            {
                const uint32_t old = __hip_atomic_fetch_add(lctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if ((int) old == nred - 1)
                {
                    __hip_atomic_store(lctl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(a.cnt + ((size_t) op * 8 + (cu & 7)) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        PS_T(5);

        // ---- decode-ahead of the next op: fills the wait at the edge
        P = decode_ahead(nxt, ring_has_next);
        PS_T(6);
        cur = nxt; tl = tn;
    }
    #undef PS_T
}
