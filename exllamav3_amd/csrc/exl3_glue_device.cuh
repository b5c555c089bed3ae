// Device helpers shared by the glue kernels (exl3_glue.hip) and the in-kernel tail epilogues of the gen-2 GEMV
// (exl3_gemv2.kspec.hip): split-k slab reduction, 128-wide output / input Hadamard on a 32-lane half-wave with 4 values
// per lane, and the quantized KV-cache append on register values (same arithmetic as exl3_rope_cache.hip).
#pragma once
#include "exl3_common.cuh"

#include "exl3_gemv_args.h"

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

__device__ __forceinline__ void kvg_had32(float& v0, float& v1, float& v2, float& v3, int lane)
{
    float s0 = v0 + v1, d0 = v0 - v1, s1 = v2 + v3, d1 = v2 - v3;
    v0 = s0 + s1; v1 = d0 + d1; v2 = s0 - s1; v3 = d0 - d1;
    #pragma unroll
    for (int i = 1; i < 8; i <<= 1)
    {
        float p0 = xor_lane(v0, i), p1 = xor_lane(v1, i), p2 = xor_lane(v2, i), p3 = xor_lane(v3, i);
        bool neg = (lane & i) != 0;
        v0 = (neg ? -v0 : v0) + p0; v1 = (neg ? -v1 : v1) + p1; v2 = (neg ? -v2 : v2) + p2; v3 = (neg ? -v3 : v3) + p3;
    }
}

template <int W>
__device__ __forceinline__ void kvg_pack_plane(uint32_t* __restrict__ out, int word_base, int sl, uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3, bool active)
{
    uint32_t field = f0 | (f1 << W) | (f2 << (2 * W)) | (f3 << (3 * W));
    constexpr int LPW = 8 / W;
    int off = sl * 4 * W;
    uint32_t contrib = field << (off & 31);
    #pragma unroll
    for (int i = 1; i < LPW; i <<= 1) contrib |= (uint32_t) xor_lane((int) contrib, i);
    if (active && (sl % LPW) == 0) out[word_base + (off >> 5)] = contrib;
}

// same arithmetic as kv_quant_group in exl3_rope_cache.hip, input already in registers (fp16-rounded values); `bits` may be a
// runtime value (GEMV tail epilogue) or a compile-time constant (glue_qkv_kernel) -- the arithmetic is identical.
__device__ __forceinline__ void kv_quant_regs_rt(const int bits, float v0, float v1, float v2, float v3, uint32_t* __restrict__ out,
                                                 half_t* __restrict__ out_scale, bool active, int lane)
{
    const float mf = (float) (1 << (bits - 1));
    const int qmax = (1 << bits) - 1;
    const int sl = lane & 7;
    kvg_had32(v0, v1, v2, v3, lane);
    const float r32 = 0.17677669529663688110f;
    v0 *= r32; v1 *= r32; v2 *= r32; v3 *= r32;
    float s = fmaxf(fmaxf(fabsf(v0), fabsf(v1)), fmaxf(fabsf(v2), fabsf(v3))) + 1e-10f;
    #pragma unroll
    for (int i = 1; i < 8; i <<= 1) s = fmaxf(s, xor_lane(s, i));
    const float inv_s = 1.0f / s;
    auto quant1 = [&] (float v) -> uint32_t { int qi = (int) floorf(__builtin_fmaf(v * inv_s, mf, mf)); return (uint32_t) max(min(qi, qmax), 0); };
    uint32_t q0 = quant1(v0), q1 = quant1(v1), q2 = quant1(v2), q3 = quant1(v3);
    int rem = bits, wb = 0;
    if (bits & 8) { rem -= 8; kvg_pack_plane<8>(out, wb, sl, (q0 >> rem) & 255, (q1 >> rem) & 255, (q2 >> rem) & 255, (q3 >> rem) & 255, active); wb += 8; }
    if (bits & 4) { rem -= 4; kvg_pack_plane<4>(out, wb, sl, (q0 >> rem) & 15, (q1 >> rem) & 15, (q2 >> rem) & 15, (q3 >> rem) & 15, active); wb += 4; }
    if (bits & 2) { rem -= 2; kvg_pack_plane<2>(out, wb, sl, (q0 >> rem) & 3, (q1 >> rem) & 3, (q2 >> rem) & 3, (q3 >> rem) & 3, active); wb += 2; }
    if (bits & 1) { kvg_pack_plane<1>(out, wb, sl, q0 & 1, q1 & 1, q2 & 1, q3 & 1, active); }
    if (active && sl == 0) *out_scale = f2h(s);
}

template <int BITS>
__device__ __forceinline__ void kv_quant_regs(float v0, float v1, float v2, float v3, uint32_t* __restrict__ out, half_t* __restrict__ out_scale, bool active, int lane)
{
    kv_quant_regs_rt(BITS, v0, v1, v2, v3, out, out_scale, active, lane);
}

// Rotated-domain values of one quantized 32-group (8 lanes x 4 values): u = (level - (2^(b-1) - 0.5)) * scale / 2^(b-1).  The dequantized
// K / V the reference materialises is x' = H32(u) / sqrt(32) (cache/q_cache_kernels.cuh:150-236, exl3_rope_cache.hip kv_dequant_group); attention
// can stay in the rotated domain because H32/sqrt(32) is orthonormal and symmetric.  `bits` may be a runtime value.
__device__ __forceinline__ void kv_dequant_vals_rt(const int bits, const uint32_t* __restrict__ in, const half_t* __restrict__ in_scale, int lane,
                                                   float& v0, float& v1, float& v2, float& v3)
{
    // branch-free: the (up to 4) plane words and the scale are loaded unconditionally (absent planes re-read word 0) so that a caller that
    // unrolls over several tokens gets all of their loads in flight at once; absent planes are masked out with selects
    const int sl = lane & 7;
    uint32_t word[4];
    int wb = 0;
    #pragma unroll
    for (int pi = 0; pi < 4; ++pi)
    {
        const int w = 8 >> pi;
        const bool has = (bits & w) != 0;
        const int off = sl * 4 * w;
        word[pi] = in[has ? wb + (off >> 5) : 0];
        wb += has ? w : 0;
    }
    const float scale = (float) *in_scale;
    uint32_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    #pragma unroll
    for (int pi = 0; pi < 4; ++pi)
    {
        const int w = 8 >> pi;
        const bool has = (bits & w) != 0;
        const int off = sl * 4 * w;
        const uint32_t x = word[pi] >> (off & 31);
        const uint32_t mask = (1u << w) - 1u;
        q0 = has ? ((q0 << w) | (x & mask)) : q0;
        q1 = has ? ((q1 << w) | ((x >> w) & mask)) : q1;
        q2 = has ? ((q2 << w) | ((x >> (2 * w)) & mask)) : q2;
        q3 = has ? ((q3 << w) | ((x >> (3 * w)) & mask)) : q3;
    }
    const int m = 1 << (bits - 1);
    const float sm = scale * (1.0f / (float) m);
    const float mh = (float) m - 0.5f;
    v0 = ((float) (int) q0 - mh) * sm; v1 = ((float) (int) q1 - mh) * sm;
    v2 = ((float) (int) q2 - mh) * sm; v3 = ((float) (int) q3 - mh) * sm;
}
