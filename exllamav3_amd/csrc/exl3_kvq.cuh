// KV-cache quantization of one 32-value group on gfx950, written from the cache FORMAT (reference semantics:
// cache/q_cache_kernels.cuh:61-236 -- pinned bit for bit through oracle/_ref):
//   v = H32(x) / sqrt(32);  s = max|v| + 1e-10;  q_e = clamp(floor(fma(v_e / s, 2^(b-1), 2^(b-1))), 0, 2^b - 1);  scale = fp16(s)
//   bit planes: the set bits of b among (8, 4, 2, 1), most significant plane first; the plane of width w is w consecutive u32 words holding
//   field e = (q_e >> rem) & (2^w - 1) at bit e * w (rem = bits below the plane)
//   dequant: u_e = (q_e - (2^(b-1) - 0.5)) * fp32(scale) / sqrt(32) / 2^(b-1);  x' = H32(u)
//
// Lane mapping (wave64): a group is spread over LPG = 32 / VPL lanes with VPL CONSECUTIVE values per lane.
//   VPL = 2 -> LPG = 16: one group per 16-lane DPP row, four groups (one 128-wide head) per wave.  Every cross-lane step -- the four Hadamard
//             butterflies above the in-lane one, the max reduction, the OR-reduction of plane words -- is a DPP row operation (quad_perm,
//             row_half_mirror, row_mirror): no LDS, no permlane, no ds_bpermute.  Used by the standalone cache kernels (exl3_rope_cache.hip).
//   VPL = 4 -> LPG = 8 : the layout in which the fused decode kernels already hold a head (4 values per lane after the 128-point output
//             Hadamard: exl3_glue.hip, exl3_gemv2_tail.cuh, exl3_attn_decode.hip).
// All plane arithmetic is generic in (VPL, w): a lane's VPL fields occupy bits [gl * VPL * w, (gl + 1) * VPL * w) of the plane, i.e. word
// (gl * VPL * w) >> 5 at shift (gl * VPL * w) & 31, and 32 / (VPL * w) neighbouring lanes share a word.
#pragma once
#include "exl3_common.cuh"

#define KVQ_R32 0.17677669529663688110f     // 1 / sqrt(32)

// Optional cubic compander of the level grid (reference: cache/lmq.cuh LMCubic, used by quant_block_x4 / dequant_block_x4 when compand_a > 0):
//   decode: t = (2 q + 1) / 2^b - 1;  level = a t + (1 - a) t^3            (fma order as below)
//   encode: the real root of (1 - a) t^3 + a t = x by Cardano;  q = clamp(floor(fma(t, 2^(b-1), 2^(b-1))), 0, 2^b - 1)
// sqrtf is correctly rounded here; cbrtf is the device library's (<= 1 ulp), so an x within an ulp of a cell boundary may land in the
// neighbouring cell compared with another libm -- the parity tests allow that and nothing else.
struct KvCompander
{
    float a, b, inv_b, p3_cub;
    __device__ __forceinline__ explicit KvCompander(float a_) : a(a_), b(1.0f - a_), inv_b(1.0f / (1.0f - a_))
    {
        const float p3 = a * inv_b * (1.0f / 3.0f);
        p3_cub = p3 * p3 * p3;
    }
    __device__ __forceinline__ float decode(uint32_t q, int bits) const
    {
        const float t = __builtin_fmaf(2.0f * (float) (int) q + 1.0f, 1.0f / (float) (1 << bits), -1.0f);
        return t * __builtin_fmaf(t * t, b, a);
    }
    __device__ __forceinline__ uint32_t encode(float x, int bits) const
    {
        const float q_half = x * inv_b * 0.5f;
        const float delta = __builtin_fmaf(q_half, q_half, p3_cub);
        const float s = sqrtf(delta);
        const float t = cbrtf(q_half + s) + cbrtf(q_half - s);
        const float half_n = (float) (1 << (bits - 1));
        const int qi = (int) floorf(__builtin_fmaf(t, half_n, half_n));
        return (uint32_t) max(min(qi, (1 << bits) - 1), 0);
    }
};

template <int VPL>
struct KvGroup
{
    static_assert(VPL == 2 || VPL == 4, "2 or 4 values per lane");
    static constexpr int LPG = 32 / VPL;

    // H32 over the group: Sylvester order, value index = gl * VPL + i.  In-lane stages first (index bits 0 .. log2(VPL) - 1), then lane bits.
    __device__ static __forceinline__ void hadamard(float (&v)[VPL], int lane)
    {
        if constexpr (VPL == 2) { const float a = v[0] + v[1], b = v[0] - v[1]; v[0] = a; v[1] = b; }
        else
        {
            const float s0 = v[0] + v[1], d0 = v[0] - v[1], s1 = v[2] + v[3], d1 = v[2] - v[3];
            v[0] = s0 + s1; v[1] = d0 + d1; v[2] = s0 - s1; v[3] = d0 - d1;
        }
        #pragma unroll
        for (int i = 1; i < LPG; i <<= 1)
        {
            const bool neg = (lane & i) != 0;
            #pragma unroll
            for (int j = 0; j < VPL; ++j) { const float p = xor_lane(v[j], i); v[j] = (neg ? -v[j] : v[j]) + p; }
        }
    }

    __device__ static __forceinline__ float group_max(float s)
    {
        #pragma unroll
        for (int i = 1; i < LPG; i <<= 1) s = fmaxf(s, xor_lane(s, i));
        return s;
    }

    // one plane of width W: this lane's VPL fields -> the shared word, OR-reduced over the lanes of that word; the first lane of the word stores it
    template <int W>
    __device__ static __forceinline__ void store_plane(uint32_t* __restrict__ out, int word_base, int gl, const uint32_t (&q)[VPL], int rem, bool active)
    {
        constexpr int LPW = (32 / (VPL * W)) > 0 ? 32 / (VPL * W) : 1;        // lanes per word
        uint32_t field = 0;
        #pragma unroll
        for (int j = 0; j < VPL; ++j) field |= ((q[j] >> rem) & ((1u << W) - 1u)) << (j * W);
        const int off = gl * VPL * W;
        uint32_t word = field << (off & 31);
        #pragma unroll
        for (int i = 1; i < LPW; i <<= 1) word |= xor_lane(word, i);
        if (active && (gl % LPW) == 0) out[word_base + (off >> 5)] = word;
    }

    // quantize the group held in v (already fp16-rounded inputs as fp32); `bits` may be a compile-time or a run-time value (same arithmetic)
    __device__ static __forceinline__ void quantize(const int bits, float (&v)[VPL], uint32_t* __restrict__ out, half_t* __restrict__ out_scale,
                                                    bool active, int lane, const float compand_a = 0.0f)
    {
        const int gl = lane % LPG;
        hadamard(v, lane);
        float s = 0.0f;
        #pragma unroll
        for (int j = 0; j < VPL; ++j) { v[j] *= KVQ_R32; s = fmaxf(s, fabsf(v[j])); }
        s = group_max(s) + 1e-10f;
        const float inv_s = 1.0f / s;                          // IEEE division (the oracle's definition; the reference builds with fast-math)
        const float half_range = (float) (1 << (bits - 1));
        const int qmax = (1 << bits) - 1;
        uint32_t q[VPL];
        if (compand_a > 0.0f)
        {
            const KvCompander lm(compand_a);
            #pragma unroll
            for (int j = 0; j < VPL; ++j) q[j] = lm.encode(v[j] * inv_s, bits);
        }
        else
        {
            #pragma unroll
            for (int j = 0; j < VPL; ++j)
            {
                const int qi = (int) floorf(__builtin_fmaf(v[j] * inv_s, half_range, half_range));
                q[j] = (uint32_t) max(min(qi, qmax), 0);
            }
        }
        int rem = bits, wb = 0;
        if (bits & 8) { rem -= 8; store_plane<8>(out, wb, gl, q, rem, active); wb += 8; }
        if (bits & 4) { rem -= 4; store_plane<4>(out, wb, gl, q, rem, active); wb += 4; }
        if (bits & 2) { rem -= 2; store_plane<2>(out, wb, gl, q, rem, active); wb += 2; }
        if (bits & 1) { rem -= 1; store_plane<1>(out, wb, gl, q, rem, active); }
        if (active && gl == 0) *out_scale = f2h(s);
    }

    // rotated-domain values u of the group (before the inverse H32): branch-free -- all (up to four) plane words and the scale are requested
    // before any is used, absent planes re-read word 0 and are masked out, so a caller unrolling over tokens gets every load in flight at once
    // WITH_R32: include the 1/sqrt(32) of the inverse Hadamard (dequantization to x'); without it the caller folds that factor elsewhere
    // (decode attention keeps K / V in the rotated domain)
    template <bool WITH_R32>
    __device__ static __forceinline__ void levels(const int bits, const uint32_t* __restrict__ in, const half_t* __restrict__ in_scale, int lane, float (&u)[VPL],
                                                  const float compand_a = 0.0f)
    {
        const int gl = lane % LPG;
        uint32_t word[4];
        int wb = 0;
        #pragma unroll
        for (int pi = 0; pi < 4; ++pi)
        {
            const int w = 8 >> pi;
            const bool has = (bits & w) != 0;
            word[pi] = in[has ? wb + ((gl * VPL * w) >> 5) : 0];
            wb += has ? w : 0;
        }
        const float scale = (float) *in_scale;
        uint32_t q[VPL];
        #pragma unroll
        for (int j = 0; j < VPL; ++j) q[j] = 0;
        #pragma unroll
        for (int pi = 0; pi < 4; ++pi)
        {
            const int w = 8 >> pi;
            const bool has = (bits & w) != 0;
            const uint32_t x = word[pi] >> ((gl * VPL * w) & 31);
            #pragma unroll
            for (int j = 0; j < VPL; ++j) q[j] = has ? ((q[j] << w) | ((x >> (j * w)) & ((1u << w) - 1u))) : q[j];
        }
        if (compand_a > 0.0f)
        {
            const KvCompander lm(compand_a);
            const float sc = WITH_R32 ? scale * KVQ_R32 : scale;
            #pragma unroll
            for (int j = 0; j < VPL; ++j) u[j] = lm.decode(q[j], bits) * sc;
            return;
        }
        const int m = 1 << (bits - 1);
        const float sm = (WITH_R32 ? scale * KVQ_R32 : scale) * (1.0f / (float) m);
        const float mh = (float) m - 0.5f;
        #pragma unroll
        for (int j = 0; j < VPL; ++j) u[j] = ((float) (int) q[j] - mh) * sm;
    }
};
