// Tail epilogues of the gen-2 GEMV: what the fused decode step ran as separate "glue" launches (exl3_glue.hip) now runs inside
// the GEMV launch itself, on the workgroup that finishes a column block LAST.
//
// Every workgroup writes its raw fp32 partial slab (write-through stores, see st_agent below) and takes an arrival ticket for
// its column block.  The workgroup that draws the last ticket sums the S slabs of that 128-column block in a fixed order (run-to-run deterministic) and finishes the block:
//   NORM : out-had * svh (+bias) -> residual += y (fp16) -> partial sum of squares; a second, launch-wide ticket elects the
//          workgroup that normalises the full row and applies the input Hadamard of every consumer      (was glue_norm)
//   ACT  : gate & up blocks -> fp16(silu(g) * u) -> (x * suh_down) input Hadamard of down_proj             (was glue_act)
//   QKV  : one head (head_dim 128 == one Hadamard block) -> RoPE -> q out / quantized paged KV append      (was glue_qkv)
// A 128-wide block is one 32-lane half-wave with 4 values per lane; the arithmetic is the glue kernels' (same device
// functions, same rounding points), so both pipelines produce identical bits.
// Replaces, in the reference, the graph nodes between two exl3_gemm calls of BC_Attention / BC_GatedMLP
// (libtorch/attention.cpp:246-504, libtorch/mlp.cpp:14-91) and the lock-based split-k hand-off of the GEMM kernel
// (quant/exl3_gemm_inner.cuh barrier + reduce).
#pragma once
#include "exl3_gemv_args.h"
#include "exl3_glue_device.cuh"

// Cross-workgroup hand-off inside one launch.  The 8 XCD L2s are not coherent with each other for plain accesses, and an
// agent-scope release/acquire FENCE costs an L2 write-back / invalidate per wave (measured: ~115 us per launch with 8192 waves).
// So every datum that crosses workgroups (split-k slabs, NORM residual block + sum of squares) is written and read with
// agent-scope relaxed ATOMIC accesses instead -- write-through / L2-bypassing (sc1) loads and stores -- and the only
// ordering needed is "my stores have been acknowledged before my ticket is taken": s_waitcnt vmcnt(0) + workgroup barrier.
__device__ __forceinline__ void st_agent(float* p, float4_t v)
{
    union { float4_t f; uint64_t u[2]; } c; c.f = v;
    __hip_atomic_store((uint64_t*) p, c.u[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((uint64_t*) p + 1, c.u[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4_t ld_agent(const float* p)
{
    union { float4_t f; uint64_t u[2]; } c;
    c.u[0] = __hip_atomic_load((uint64_t*) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    c.u[1] = __hip_atomic_load((uint64_t*) p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return c.f;
}
__device__ __forceinline__ void st_agent(half_t* p, half4_t v)
{
    union { half4_t h; uint64_t u; } c; c.h = v;
    __hip_atomic_store((uint64_t*) p, c.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ half4_t ld_agent(const half_t* p)
{
    union { half4_t h; uint64_t u; } c;
    c.u = __hip_atomic_load((uint64_t*) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return c.h;
}
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent1(const float* p) { return __hip_atomic_load((float*) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// returns true for the workgroup that arrived last at `ticket` (and re-arms the ticket)
// local: every participant runs on the same XCD (GemvEpi::xcd_local).  Stores are acknowledged by that XCD's L2 and atomics execute there, so a
// plain store -> s_waitcnt -> L2 atomic -> plain load chain is ordered without any memory-side (sc1) round trip: the hand-off costs an L2
// latency instead of the ~2 us per hop of the agent-scope version (profiles/NOTES.md B 4.2).
__device__ __forceinline__ bool tail_arrive(uint32_t* ticket, uint32_t expected, int tid, int* s_flag, bool local = false)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores are acknowledged (by L2; write-through ones by memory)
    __syncthreads();
    if (tid == 0)
    {
        bool last = true;
        if (expected > 1)
        {
            uint32_t old = local ? __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                 : __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = old == expected - 1;
            if (last)
            {
                if (local) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        *s_flag = last ? 1 : 0;
    }
    __syncthreads();
    return *s_flag != 0;
}

// Split-k slabs of column block c, rows [row0, row0 + rows), NS slab sets: every half-wave of the workgroup fetches one
// (row, set, slice) slab line into LDS (all loads in flight at once: one L2/HBM latency instead of a dependent chain), then
// half-wave j sums row j's S lines in slice order -- the same order as slab_sum(), so glue and tail pipelines agree bit for bit.
template <int NS>
__device__ __forceinline__ void tail_gather(const float* const (&bases)[NS], const int S, const int c, const int m, const int row0, const int rows,
                                            float4_t* buf, const int hw8, const int nhw, const int l, const int rr, float4_t (&v)[NS], const bool local = false)
{
    const int items = rows * NS * S;
    for (int it = hw8; it < items; it += nhw)
    {
        const int sl = it % S, q = it / S, set = q % NS, r = q / NS;
        const float* base = bases[0];
        #pragma unroll
        for (int i = 1; i < NS; ++i) if (set == i) base = bases[i];
        const float* src = base + ((size_t) (c * S + sl) * m + row0 + r) * 128 + 4 * l;
        buf[it * 32 + l] = local ? *((const float4_t*) src) : ld_agent(src);
    }
    __syncthreads();
    #pragma unroll
    for (int set = 0; set < NS; ++set)
    {
        float4_t acc = { 0.f, 0.f, 0.f, 0.f };
        const float4_t* p = buf + (size_t) ((rr * NS + set) * S) * 32 + l;
        for (int sl = 0; sl < S; ++sl) { float4_t t = p[sl * 32]; acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
        v[set] = acc;
    }
}

__device__ __forceinline__ void gemv_tail(const GemvArgs& a, const int mi, const int cbl, const int cbg, const int tid, const int nwv, int* s_flag, float4_t* buf)
{
    const GemvEpi& e = a.epi;
    const int R = e.rows_per_pass;
    const int m = a.m, S = a.S;
    const int l = tid & 31, hw8 = tid >> 5, nhw = nwv * 2;
    const int group = e.mode == GEMV_EPI_ACT ? cbl : cbg;
    const uint32_t expected = e.mode == GEMV_EPI_ACT ? 2u * S : (uint32_t) S;
    const bool local = e.xcd_local != 0;
    if (!tail_arrive(e.tickets + group, expected, tid, s_flag, local)) return;

    if (e.mode == GEMV_EPI_ACT)
    {
        const SlabRef sg = { a.workspace + a.mat[0].ws_offset, S }, su = { a.workspace + a.mat[1].ws_offset, S };
        const int inter = a.mat[0].n, nblk = inter >> 7;
        const float* const bases[2] = { sg.base, su.base };
        for (int row0 = 0; row0 < m; row0 += R)
        {
            const int rows = min(R, m - row0);
            const bool act = hw8 < rows;
            const int rr = act ? hw8 : 0, row = row0 + rr;
            float4_t v[2];
            tail_gather<2>(bases, S, cbl, m, row0, rows, buf, hw8, nhw, l, rr, v);
            float g0, g1, g2, g3, u0, u1, u2, u3;
            out_had(v[0], l, g0, g1, g2, g3);
            out_had(v[1], l, u0, u1, u2, u3);
            half4_t gh = half4_t{ f2h(g0), f2h(g1), f2h(g2), f2h(g3) } * ((const half4_t*) (a.mat[0].svh + cbl * 128))[l];
            half4_t uh = half4_t{ f2h(u0), f2h(u1), f2h(u2), f2h(u3) } * ((const half4_t*) (a.mat[1].svh + cbl * 128))[l];
            auto silu_mul = [] (half_t g, half_t u) -> half_t { float gf = (float) g; return f2h(gf / (1.0f + __expf(-gf)) * (float) u); };
            half4_t av = { silu_mul(gh.x, uh.x), silu_mul(gh.y, uh.y), silu_mul(gh.z, uh.z), silu_mul(gh.w, uh.w) };
            if (e.a_out && act) ((half4_t*) (e.a_out + (size_t) row * inter + cbl * 128))[l] = av;
            float sum = in_had_store(av, e.t_suh[0] + cbl * 128, e.t_xh[0] + (size_t) row * inter + cbl * 128, l, act);
            if (act && l == 0 && e.t_xsum[0]) e.t_xsum[0][(size_t) row * nblk + cbl] = sum;
            __syncthreads();
        }
        return;
    }

    if (e.mode == GEMV_EPI_QKV)
    {
        const int kind = mi, hi = cbl;
        const SlabRef sr = { a.workspace + a.mat[mi].ws_offset, S };
        const half_t* svh = a.mat[mi].svh + hi * 128;
        const float* const bases[1] = { sr.base };
        for (int row0 = 0; row0 < m; row0 += R)
        {
            const int rows = min(R, m - row0);
            const bool act = hw8 < rows;
            const int rr = act ? hw8 : 0, row = row0 + rr;
            float4_t v[1];
            tail_gather<1>(bases, S, hi, m, row0, rows, buf, hw8, nhw, l, rr, v);
            float h0, h1, h2, h3;
            out_had(v[0], l, h0, h1, h2, h3);
            half4_t y = half4_t{ f2h(h0), f2h(h1), f2h(h2), f2h(h3) } * ((const half4_t*) svh)[l];
            const int pos = e.positions[row];
            if (kind != 2)
            {
                float v0 = (float) y.x, v1 = (float) y.y, v2 = (float) y.z, v3 = (float) y.w;
                // sin/cos of (position x frequency) come from the per-step table [row][64] (exl3_rope_table: all layers of a step share
                // the positions, so the table is built once per step instead of once per layer)
                const float* sn = e.rope_sin + row * 64; const float* cs = e.rope_cos + row * 64;
                if (e.rope_mode == 2)
                {
                    // NEOX pairs (d, d+64): partner lane l ^ 16, frequency index d & 63
                    float p0 = xor_lane(v0, 16), p1 = xor_lane(v1, 16), p2 = xor_lane(v2, 16), p3 = xor_lane(v3, 16);
                    const float4_t s4 = ((const float4_t*) sn)[l & 15], c4 = ((const float4_t*) cs)[l & 15];
                    const bool upper = l >= 16;
                    float r0 = upper ? v0 * c4.x + p0 * s4.x : v0 * c4.x - p0 * s4.x;
                    float r1 = upper ? v1 * c4.y + p1 * s4.y : v1 * c4.y - p1 * s4.y;
                    float r2 = upper ? v2 * c4.z + p2 * s4.z : v2 * c4.z - p2 * s4.z;
                    float r3 = upper ? v3 * c4.w + p3 * s4.w : v3 * c4.w - p3 * s4.w;
                    y = half4_t{ f2h(r0), f2h(r1), f2h(r2), f2h(r3) };
                }
                else
                {
                    // GPTJ pairs (2i, 2i+1), both in this lane: frequencies 2l, 2l+1
                    const float sa = sn[2 * l], sb = sn[2 * l + 1], ca = cs[2 * l], cb = cs[2 * l + 1];
                    y = half4_t{ f2h(v0 * ca - v1 * sa), f2h(v1 * ca + v0 * sa),
                                 f2h(v2 * cb - v3 * sb), f2h(v3 * cb + v2 * sb) };
                }
            }
            if (kind == 0 && act) ((half4_t*) (e.q_out + ((size_t) row * e.hq + hi) * 128))[l] = y;
            half_t* dense = kind == 1 ? e.k_out : (kind == 2 ? e.v_out : nullptr);
            if (dense && act) ((half4_t*) (dense + ((size_t) row * e.hkv + hi) * 128))[l] = y;
            if (kind != 0 && e.k_cache)                                     // uniform per workgroup
            {
                const int page_idx = pos / e.page_size;
                const int64_t token_pos = (int64_t) e.block_table[row * e.blocks_per_seq + page_idx] * e.page_size + (pos % e.page_size);
                const int64_t gbase = token_pos * (e.hkv * 4) + hi * 4 + (l >> 3);
                const int bits = kind == 1 ? e.k_bits : e.v_bits;
                uint32_t* cache = kind == 1 ? e.k_cache : e.v_cache;
                half_t* scales = kind == 1 ? e.k_scales : e.v_scales;
                kv_quant_regs_rt(bits, (float) y.x, (float) y.y, (float) y.z, (float) y.w, cache + gbase * bits, scales + gbase, act, tid & 63);
            }
            __syncthreads();
        }
        return;
    }

    // ---- NORM
    {
        const int hidden = a.mat[0].n, nblk = hidden >> 7;
        const SlabRef sr = { a.workspace + a.mat[0].ws_offset, S };
        float* ss_part = e.mode == GEMV_EPI_RESID ? e.ss_out : a.workspace + e.ss_offset;   // [m][nblk]
        const half_t* svh = a.mat[0].svh + cbl * 128;
        const half_t* bias = a.mat[0].bias ? a.mat[0].bias + cbl * 128 : nullptr;
        const float* const bases[1] = { sr.base };
        for (int row0 = 0; row0 < m; row0 += R)
        {
            const int rows = min(R, m - row0);
            const bool act = hw8 < rows;
            const int rr = act ? hw8 : 0, row = row0 + rr;
            half_t* rp = e.resid + (size_t) row * hidden + cbl * 128;
            half4_t r = ((const half4_t*) rp)[l];
            float4_t v[1];
            tail_gather<1>(bases, S, cbl, m, row0, rows, buf, hw8, nhw, l, rr, v, local);
            float h0, h1, h2, h3;
            out_had(v[0], l, h0, h1, h2, h3);
            half4_t sc = ((const half4_t*) svh)[l];
            h0 *= (float) sc.x; h1 *= (float) sc.y; h2 *= (float) sc.z; h3 *= (float) sc.w;
            if (bias) { half4_t b = ((const half4_t*) bias)[l]; h0 += (float) b.x; h1 += (float) b.y; h2 += (float) b.z; h3 += (float) b.w; }
            r = half4_t{ f2h((float) r.x + h0), f2h((float) r.y + h1), f2h((float) r.z + h2), f2h((float) r.w + h3) };
            const float r0 = (float) r.x, r1 = (float) r.y, r2 = (float) r.z, r3 = (float) r.w;
            if (act) { if (e.mode == GEMV_EPI_RESID) ((half4_t*) rp)[l] = r; else st_agent(rp + 4 * l, r); }    // RESID: read by the NEXT launch only
            float ss = r0 * r0;
            ss = __builtin_fmaf(r1, r1, ss); ss = __builtin_fmaf(r2, r2, ss); ss = __builtin_fmaf(r3, r3, ss);
            #pragma unroll
            for (int i = 1; i < 32; i <<= 1) ss += xor_lane(ss, i);
            if (act && l == 0) { if (e.mode == GEMV_EPI_RESID) ss_part[row * nblk + cbl] = ss; else st_agent(ss_part + row * nblk + cbl, ss); }
            __syncthreads();
        }
        if (e.mode == GEMV_EPI_RESID) return;                               // the row-wide RMSNorm happens in the consumer (GEMV_IN_NORM)
        if (!tail_arrive(e.tickets + e.ticket_global, (uint32_t) nblk, tid, s_flag)) return;

        const int tasks = m * nblk;
        for (int base = 0; base < tasks; base += nhw)
        {
            const bool act = base + hw8 < tasks;
            const int t = act ? base + hw8 : 0;
            const int row = t / nblk, blk = t % nblk;
            float s2 = 0.0f;
            for (int b0 = 0; b0 < nblk; b0 += 32)
            {
                float v = (b0 + l < nblk) ? ld_agent1(ss_part + row * nblk + b0 + l) : 0.0f;
                #pragma unroll
                for (int i = 1; i < 32; i <<= 1) v += xor_lane(v, i);
                s2 += v;
            }
            const float rmf = __frsqrt_rn(s2 / (float) hidden + e.eps);
            half4_t r = ld_agent(e.resid + (size_t) row * hidden + blk * 128 + 4 * l);
            half4_t wv = ((const half4_t*) (e.norm_w + blk * 128))[l];
            half4_t xn = { f2h((float) r.x * (float) wv.x * rmf), f2h((float) r.y * (float) wv.y * rmf),
                           f2h((float) r.z * (float) wv.z * rmf), f2h((float) r.w * (float) wv.w * rmf) };
            if (e.xn_out && act) ((half4_t*) (e.xn_out + (size_t) row * hidden + blk * 128))[l] = xn;
            #pragma unroll
            for (int i = 0; i < 3; ++i)
            {
                if (i < e.t_count)
                {
                    float sum = in_had_store(xn, e.t_suh[i] + blk * 128, e.t_xh[i] + (size_t) row * hidden + blk * 128, l, act);
                    if (act && l == 0 && e.t_xsum[i]) e.t_xsum[i][(size_t) row * nblk + blk] = sum;
                }
            }
        }
    }
}
