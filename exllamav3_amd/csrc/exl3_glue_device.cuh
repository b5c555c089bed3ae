// Device helpers shared by the glue kernels (exl3_glue.hip) and the in-kernel tail epilogues of the gen-2 GEMV
// (exl3_gemv2.kspec.hip): split-k slab reduction, 128-wide output / input Hadamard on a 32-lane half-wave with 4 values
// per lane, and the quantized KV-cache append on register values (same arithmetic as exl3_rope_cache.hip).
#pragma once
#include "exl3_common.cuh"

#include "exl3_gemv_args.h"
#include "exl3_kvq.cuh"

// r_new / r_prev of one row (GemvRescale): both sums in the fixed order every consumer of ss_part uses (32-lane tree per 32 blocks, then the
// partial trees in sequence).  p0 / n0: the lane's first-32 values, loaded by the caller with its other operands (0 for lanes >= k/128).
__device__ __forceinline__ float gemv_rescale(const GemvRescale& rs, int row, int l32, float p0, float n0)
{
    const int nb = rs.k >> 7;
    float sp = 0.0f, sn = 0.0f;
    for (int b0 = 0; b0 < nb; b0 += 32)
    {
        float vp = p0, vn = n0;
        if (b0 > 0)
        {
            vp = (b0 + l32 < nb) ? rs.ss_prev[(size_t) row * nb + b0 + l32] : 0.0f;
            vn = (b0 + l32 < nb) ? rs.ss_new[(size_t) row * nb + b0 + l32] : 0.0f;
        }
        #pragma unroll
        for (int i = 1; i < 32; i <<= 1) { vp += xor_lane(vp, i); vn += xor_lane(vn, i); }
        sp += vp; sn += vn;
    }
    const float rp = __frsqrt_rn(sp / (float) rs.k + rs.eps), rn = __frsqrt_rn(sn / (float) rs.k + rs.eps);
    return rn / rp;
}

struct SlabRef { const float* base; int S; };       // slab(c, s, row) = base + ((c*S + s)*m + row)*128

__device__ __forceinline__ float4_t slab_sum(const SlabRef& sr, int c, int row, int m, int l)
{
    // independent 16-byte loads in batches of 16 (a dependent load-add chain costs ~0.6 us of L2/HBM latency per slab);
    // the summation order is fixed, so results are run-to-run deterministic
    const float* p = sr.base + ((size_t) c * sr.S * m + row) * 128;
    const size_t st = (size_t) m * 128;
    float4_t v = { 0.f, 0.f, 0.f, 0.f };
    for (int s = 0; s < sr.S; s += 16)
    {
        float4_t t[16];
        #pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = ((const float4_t*) (p + (size_t) min(s + i, sr.S - 1) * st))[l];
        #pragma unroll
        for (int i = 0; i < 16; ++i) if (s + i < sr.S) { v.x += t[i].x; v.y += t[i].y; v.z += t[i].z; v.w += t[i].w; }
    }
    return v;
}

// two slab sets at once (gate & up): both sets' loads are in flight before the first add; NB lines per set per round (register budget:
// 2 * NB float4).  The summation order is slice order for every NB, so all callers agree bit for bit.
template <int NB = 8>
__device__ __forceinline__ void slab_sum2(const SlabRef& sa, const SlabRef& sb, int c, int row, int m, int l, float4_t& va, float4_t& vb)
{
    const float* pa = sa.base + ((size_t) c * sa.S * m + row) * 128;
    const float* pb = sb.base + ((size_t) c * sb.S * m + row) * 128;
    const size_t st = (size_t) m * 128;
    va = float4_t{ 0.f, 0.f, 0.f, 0.f }; vb = va;
    const int S = sa.S;                                   // both sets come from one launch: same split
    for (int s = 0; s < S; s += NB)
    {
        float4_t ta[NB], tb[NB];
        #pragma unroll
        for (int i = 0; i < NB; ++i) { ta[i] = ((const float4_t*) (pa + (size_t) min(s + i, S - 1) * st))[l]; tb[i] = ((const float4_t*) (pb + (size_t) min(s + i, S - 1) * st))[l]; }
        #pragma unroll
        for (int i = 0; i < NB; ++i) if (s + i < S)
        {
            va.x += ta[i].x; va.y += ta[i].y; va.z += ta[i].z; va.w += ta[i].w;
            vb.x += tb[i].x; vb.y += tb[i].y; vb.z += tb[i].z; vb.w += tb[i].w;
        }
    }
}

// out-Hadamard of a reduced block -> fp32 (h *= 1/sqrt(128)); caller applies svh in the dtype of the logical output
__device__ __forceinline__ void out_had(float4_t v, int l, float& h0, float& h1, float& h2, float& h3)
{
    h0 = v.x; h1 = v.y; h2 = v.z; h3 = v.w;
    had128_f32x4(h0, h1, h2, h3, l);
    h0 *= HAD_R_SCALE_128; h1 *= HAD_R_SCALE_128; h2 *= HAD_R_SCALE_128; h3 *= HAD_R_SCALE_128;
}

// One 128-wide block of the q|k|v launch's output, finished: out-Hadamard of the split-k sum -> row-scale correction (r_new / r_prev, launches that
// normalised with the previous residual's 1/rms) -> fp16 -> x svh in fp16 (the output semantics of exl3_gemm) -> RoPE on the fp16 values (q and k;
// sn4 / cs4 = the lane's sin / cos: NEOX frequencies 4 (l mod ph) .. + 3, GPTJ 2 (l mod hd/4), + 1 in .x / .y).  ph = head_dim / 8 = the NEOX partner
// distance in lanes (16 at head_dim 128, 8 at 64).  Shared by glue_qkv_kernel and the attention kernel that does this work itself
// (exl3_attn_decode.hip, fused form): one arithmetic, bit-identical results.
__device__ __forceinline__ half4_t qkv_block_finish(float4_t ysum, half4_t sc, const GemvRescale& rs, int row, int l, float rs_p, float rs_n,
                                                    bool rope, int rope_mode, int ph, float4_t sn4, float4_t cs4)
{
    float h0, h1, h2, h3;
    out_had(ysum, l, h0, h1, h2, h3);
    if (rs.ss_new) { const float rsc = gemv_rescale(rs, row, l, rs_p, rs_n); h0 *= rsc; h1 *= rsc; h2 *= rsc; h3 *= rsc; }
    half4_t y = half4_t{ f2h(h0), f2h(h1), f2h(h2), f2h(h3) } * sc;
    if (rope)
    {
        float v0 = (float) y.x, v1 = (float) y.y, v2 = (float) y.z, v3 = (float) y.w;
        if (rope_mode == 2)
        {
            // NEOX: pairs (d, d + hd/2) inside a head: partner lane l ^ (hd/8), frequency index d mod hd/2
            float p0, p1, p2, p3;
            if (ph == 16) { p0 = xor_lane(v0, 16); p1 = xor_lane(v1, 16); p2 = xor_lane(v2, 16); p3 = xor_lane(v3, 16); }
            else          { p0 = xor_lane(v0, 8);  p1 = xor_lane(v1, 8);  p2 = xor_lane(v2, 8);  p3 = xor_lane(v3, 8); }
            const bool upper = (l & ph) != 0;
            // lower half: r1 = v1*cos - v2*sin ; upper half: r2 = v2*cos + v1*sin   (v1 = lower element, v2 = upper element)
            float r0 = upper ? v0 * cs4.x + p0 * sn4.x : v0 * cs4.x - p0 * sn4.x;
            float r1 = upper ? v1 * cs4.y + p1 * sn4.y : v1 * cs4.y - p1 * sn4.y;
            float r2 = upper ? v2 * cs4.z + p2 * sn4.z : v2 * cs4.z - p2 * sn4.z;
            float r3 = upper ? v3 * cs4.w + p3 * sn4.w : v3 * cs4.w - p3 * sn4.w;
            y = half4_t{ f2h(r0), f2h(r1), f2h(r2), f2h(r3) };
        }
        else
        {
            // GPTJ: pairs (2i, 2i+1) both in this lane: frequencies 2l', 2l'+1 with l' the lane inside the head
            y = half4_t{ f2h(v0 * cs4.x - v1 * sn4.x), f2h(v1 * cs4.x + v0 * sn4.x),
                         f2h(v2 * cs4.y - v3 * sn4.y), f2h(v3 * cs4.y + v2 * sn4.y) };
        }
    }
    return y;
}

// input Hadamard of the next linear: xh = fp16(had(fp16 x * suh) / sqrt(128)); returns the block sum of the fp16 outputs.
// The suh values are passed in registers so callers can issue that load at kernel entry (every dependent global load on the
// path of these latency-bound kernels costs ~1 us).
__device__ __forceinline__ float in_had_store_v(half4_t x, half4_t sv, half_t* __restrict__ xh_blk, int l, bool act)
{
    half4_t t = x * sv;
    float h0 = (float) t.x, h1 = (float) t.y, h2 = (float) t.z, h3 = (float) t.w;
    had128_f32x4(h0, h1, h2, h3, l);
    half4_t o = { f2h(h0 * HAD_R_SCALE_128), f2h(h1 * HAD_R_SCALE_128), f2h(h2 * HAD_R_SCALE_128), f2h(h3 * HAD_R_SCALE_128) };
    float sum = ((float) o.x + (float) o.y) + ((float) o.z + (float) o.w);
    #pragma unroll
    for (int i = 1; i < 32; i <<= 1) sum += xor_lane(sum, i);
    if (act) ((half4_t*) xh_blk)[l] = o;
    return sum;
}

__device__ __forceinline__ float in_had_store(half4_t x, const half_t* __restrict__ suh_blk, half_t* __restrict__ xh_blk, int l, bool act)
{
    return in_had_store_v(x, ((const half4_t*) suh_blk)[l], xh_blk, l, act);
}

// ---- KV-cache quantization on register values, 4 values per lane (the layout the 128-point output Hadamard leaves a head in): thin views of
// the generic group code in exl3_kvq.cuh (KvGroup<4>: 8 lanes per 32-group)
__device__ __forceinline__ void kvg_had32(float& v0, float& v1, float& v2, float& v3, int lane)
{
    float v[4] = { v0, v1, v2, v3 };
    KvGroup<4>::hadamard(v, lane);
    v0 = v[0]; v1 = v[1]; v2 = v[2]; v3 = v[3];
}

// `bits` may be a runtime value (GEMV tail epilogue) or a compile-time constant (glue_qkv_kernel) -- the arithmetic is identical
__device__ __forceinline__ void kv_quant_regs_rt(const int bits, float v0, float v1, float v2, float v3, uint32_t* __restrict__ out,
                                                 half_t* __restrict__ out_scale, bool active, int lane)
{
    float v[4] = { v0, v1, v2, v3 };
    KvGroup<4>::quantize(bits, v, out, out_scale, active, lane);
}

template <int BITS>
__device__ __forceinline__ void kv_quant_regs(float v0, float v1, float v2, float v3, uint32_t* __restrict__ out, half_t* __restrict__ out_scale, bool active, int lane)
{
    kv_quant_regs_rt(BITS, v0, v1, v2, v3, out, out_scale, active, lane);
}

// Rotated-domain values of one quantized 32-group (8 lanes x 4 values): u = (level - (2^(b-1) - 0.5)) * scale / 2^(b-1).  The dequantized
// K / V the reference materialises is x' = H32(u) / sqrt(32) (cache/q_cache_kernels.cuh:150-236); attention can stay in the rotated domain
// because H32 / sqrt(32) is orthonormal and symmetric.
__device__ __forceinline__ void kv_dequant_vals_rt(const int bits, const uint32_t* __restrict__ in, const half_t* __restrict__ in_scale, int lane,
                                                   float& v0, float& v1, float& v2, float& v3)
{
    float u[4];
    KvGroup<4>::levels<false>(bits, in, in_scale, lane, u);
    v0 = u[0]; v1 = u[1]; v2 = u[2]; v3 = u[3];
}
