// RoPE (+ optional per-head RMSNorm) and KV-cache quantization for gfx950.
// reference: exllamav3_ext/rope.cu:16-296 ; cache/q_cache_kernels.cuh:29-399, cache/q_cache.cu:19-434.
// Both are tiny HBM-bound elementwise/pack kernels (a few KB per token): the design goal is few launches, wide
// accesses and no LDS atomics, not FLOPs.
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include "exl3_kvq.cuh"

// ------------------------------------------------------------------------------------------------
// RoPE.  One wave per (token, head); lane t owns rotation pair t (+ 64, + 128 ... for head_dim > 128).
// NEOX pairs (t, t + d/2), GPTJ pairs (2t, 2t + 1).  Optional q/k head RMSNorm first, with the reference's
// rounding points (normalised value rounded to fp16, multiplied by fp16 (w + bias) in fp16: rope.cu:199-233).
// ------------------------------------------------------------------------------------------------
#define ROPE_MAX_PAIRS_PER_LANE 4      // head_dim <= 512

template <int MODE>     // 1 = GPTJ, 2 = NEOX
__global__ __launch_bounds__(256)
void rope_kernel(const half_t* __restrict__ q, half_t* __restrict__ out_q, const half_t* __restrict__ k, half_t* __restrict__ out_k,
                 const float* __restrict__ inv_freq, int seq_len, int heads_q, int heads_k, int head_dim,
                 uint32_t position, const int32_t* __restrict__ positions, const int32_t* __restrict__ position_ids,
                 float attn_factor, const half_t* __restrict__ q_norm, const half_t* __restrict__ k_norm,
                 float norm_eps, float norm_constant_bias)
{
    // one workgroup per token: sin / cos of (position x frequency) are evaluated once (accurate sincosf: arguments reach 1e5 rad) and
    // shared through LDS by all heads of the token (at prefill that is 40x fewer evaluations than one per (token, head, pair))
    __shared__ float sn_s[64 * ROPE_MAX_PAIRS_PER_LANE], cs_s[64 * ROPE_MAX_PAIRS_PER_LANE];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int token = blockIdx.x, batch = blockIdx.y;
    const int heads = heads_q + heads_k;
    {
        int pos0 = token + (int) position;
        if (positions) pos0 = token + positions[batch];
        else if (position_ids) pos0 = position_ids[(int64_t) batch * seq_len + token];
        for (int t = threadIdx.x; t < (head_dim >> 1); t += blockDim.x)
        {
            float sn, cs;
            sincosf(inv_freq[t] * (float) pos0, &sn, &cs);
            sn_s[t] = sn * attn_factor; cs_s[t] = cs * attn_factor;
        }
    }
    __syncthreads();
    for (int head = wave; head < heads; head += 4)
    {
    const bool is_q = head < heads_q;
    const int hi = is_q ? head : head - heads_q;
    const int64_t tok = (int64_t) batch * seq_len + token;
    const half_t* src = is_q ? q + (tok * heads_q + hi) * head_dim : k + (tok * heads_k + hi) * head_dim;
    half_t* dst = is_q ? out_q + (tok * heads_q + hi) * head_dim : out_k + (tok * heads_k + hi) * head_dim;
    const half_t* nw = is_q ? q_norm : k_norm;

    const int half_dim = head_dim >> 1;
    float v1[ROPE_MAX_PAIRS_PER_LANE], v2[ROPE_MAX_PAIRS_PER_LANE];
    float ss = 0.0f;
    #pragma unroll
    for (int i = 0; i < ROPE_MAX_PAIRS_PER_LANE; ++i)
    {
        int t = lane + 64 * i;
        v1[i] = 0.f; v2[i] = 0.f;
        if (t < half_dim)
        {
            int i1 = MODE == 2 ? t : 2 * t, i2 = MODE == 2 ? t + half_dim : 2 * t + 1;
            v1[i] = (float) src[i1]; v2[i] = (float) src[i2];
            ss += v1[i] * v1[i] + v2[i] * v2[i];
        }
    }
    if (nw)
    {
        #pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += xor_lane(ss, o);
        const float rmf = __frsqrt_rn(ss / (float) head_dim + norm_eps);
        const half_t bias_h = f2h(norm_constant_bias);
        #pragma unroll
        for (int i = 0; i < ROPE_MAX_PAIRS_PER_LANE; ++i)
        {
            int t = lane + 64 * i;
            if (t < half_dim)
            {
                int i1 = MODE == 2 ? t : 2 * t, i2 = MODE == 2 ? t + half_dim : 2 * t + 1;
                half_t w1 = nw[i1] + bias_h, w2 = nw[i2] + bias_h;
                v1[i] = (float) (w1 * f2h(v1[i] * rmf));
                v2[i] = (float) (w2 * f2h(v2[i] * rmf));
            }
        }
    }
    #pragma unroll
    for (int i = 0; i < ROPE_MAX_PAIRS_PER_LANE; ++i)
    {
        int t = lane + 64 * i;
        if (t < half_dim)
        {
            const float sn = sn_s[t], cs = cs_s[t];
            int i1 = MODE == 2 ? t : 2 * t, i2 = MODE == 2 ? t + half_dim : 2 * t + 1;
            dst[i1] = f2h(v1[i] * cs - v2[i] * sn);
            dst[i2] = f2h(v2[i] * cs + v1[i] * sn);
        }
    }
    }
}

// NEOX, head_dim 128, no head norm (Llama / Mixtral prefill): 8 lanes per head, 16-byte accesses -- lane l of a head holds elements
// 8l .. 8l+7 and their partners 64 + 8l .. 64 + 8l + 7.  The general kernel above moves 2 bytes per lane per access (128 B per wave
// instruction): 31.5 us for a 4096-token q + k (2.7 TB/s, profiles/r01_prefill_chunk_kernel_stats.csv).  Same arithmetic per element.
__global__ __launch_bounds__(256)
void rope_neox128_kernel(const half_t* __restrict__ q, half_t* __restrict__ out_q, const half_t* __restrict__ k, half_t* __restrict__ out_k,
                         const float* __restrict__ inv_freq, int seq_len, int heads_q, int heads_k,
                         uint32_t position, const int32_t* __restrict__ positions, const int32_t* __restrict__ position_ids, float attn_factor,
                         int64_t ld_q, int64_t ld_k)
{
    // ld_q / ld_k: halves between consecutive tokens (heads * 128 when contiguous; larger when q / k are column ranges of one fused
    // q|k|v GEMM output)
    __shared__ __attribute__((aligned(16))) float sn_s[64], cs_s[64];
    const int token = blockIdx.x, batch = blockIdx.y;
    {
        int pos0 = token + (int) position;
        if (positions) pos0 = token + positions[batch];
        else if (position_ids) pos0 = position_ids[(int64_t) batch * seq_len + token];
        if (threadIdx.x < 64)
        {
            float sn, cs;
            sincosf(inv_freq[threadIdx.x] * (float) pos0, &sn, &cs);
            sn_s[threadIdx.x] = sn * attn_factor; cs_s[threadIdx.x] = cs * attn_factor;
        }
    }
    // the first unit's data is requested before the barrier
    const int heads = heads_q + heads_k;
    const int64_t tok = (int64_t) batch * seq_len + token;
    const int l = threadIdx.x & 7;
    auto ptrs = [&] (int head, const half_t*& src, half_t*& dst)
    {
        const bool is_q = head < heads_q;
        const int hi = is_q ? head : head - heads_q;
        src = is_q ? q + tok * ld_q + hi * 128 : k + tok * ld_k + hi * 128;
        dst = is_q ? out_q + tok * ld_q + hi * 128 : out_k + tok * ld_k + hi * 128;
    };
    int head = threadIdx.x >> 3;
    half8_t a = {}, b = {};
    const half_t* src; half_t* dst = nullptr;
    if (head < heads) { ptrs(head, src, dst); a = ((const half8_t*) src)[l]; b = ((const half8_t*) (src + 64))[l]; }
    __syncthreads();
    float sn[8], cs[8];
    {
        const float4_t s0 = ((const float4_t*) sn_s)[2 * l], s1 = ((const float4_t*) sn_s)[2 * l + 1];
        const float4_t c0 = ((const float4_t*) cs_s)[2 * l], c1 = ((const float4_t*) cs_s)[2 * l + 1];
        sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
        cs[0] = c0.x; cs[1] = c0.y; cs[2] = c0.z; cs[3] = c0.w; cs[4] = c1.x; cs[5] = c1.y; cs[6] = c1.z; cs[7] = c1.w;
    }
    for (; head < heads; head += 32)
    {
        half8_t na = a, nb = b;
        const half_t* nsrc; half_t* ndst = nullptr;
        if (head + 32 < heads) { ptrs(head + 32, nsrc, ndst); na = ((const half8_t*) nsrc)[l]; nb = ((const half8_t*) (nsrc + 64))[l]; }
        half8_t oa, ob;
        #pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            const float v1 = (float) a[j], v2 = (float) b[j];
            oa[j] = f2h(v1 * cs[j] - v2 * sn[j]);
            ob[j] = f2h(v2 * cs[j] + v1 * sn[j]);
        }
        ((half8_t*) dst)[l] = oa; ((half8_t*) (dst + 64))[l] = ob;
        a = na; b = nb; dst = ndst;
    }
}

extern "C" int exl3_rope(const void* q, void* out_q, const void* k, void* out_k, const float* inv_freq,
                         int bsz, int seq_len, int heads_q, int heads_k, int head_dim,
                         uint32_t position, const int32_t* positions, const int32_t* position_ids,
                         int rope_mode, float attn_factor, const void* q_norm, const void* k_norm, float norm_eps,
                         float norm_constant_bias, void* stream)
{
    EXL3_CHECK_ARG(q && out_q && inv_freq, "rope: null pointer");
    EXL3_CHECK_ARG(heads_k == 0 || (k && out_k), "rope: k given without out_k");
    EXL3_CHECK_ARG(head_dim % 2 == 0 && head_dim > 0 && head_dim <= 128 * ROPE_MAX_PAIRS_PER_LANE, "rope: head_dim must be even and <= 512");
    EXL3_CHECK_ARG(rope_mode == 1 || rope_mode == 2, "rope: rope_mode must be 1 (GPTJ) or 2 (NEOX)");
    if (bsz == 0 || seq_len == 0) return EXL3_OK;
    dim3 grid(seq_len, bsz, 1);
    hipStream_t st = (hipStream_t) stream;
    if (rope_mode == 2 && head_dim == 128 && !q_norm && !k_norm && seq_len >= 16)
    {
        rope_neox128_kernel<<<grid, 256, 0, st>>>((const half_t*) q, (half_t*) out_q, (const half_t*) k, (half_t*) out_k, inv_freq, seq_len, heads_q, heads_k,
                                                  position, positions, position_ids, attn_factor, (int64_t) heads_q * 128, (int64_t) heads_k * 128);
        return exl3_check_launch("rope");
    }
    if (rope_mode == 2)
        rope_kernel<2><<<grid, 256, 0, st>>>((const half_t*) q, (half_t*) out_q, (const half_t*) k, (half_t*) out_k, inv_freq, seq_len, heads_q, heads_k,
                                             head_dim, position, positions, position_ids, attn_factor, (const half_t*) q_norm, (const half_t*) k_norm, norm_eps, norm_constant_bias);
    else
        rope_kernel<1><<<grid, 256, 0, st>>>((const half_t*) q, (half_t*) out_q, (const half_t*) k, (half_t*) out_k, inv_freq, seq_len, heads_q, heads_k,
                                             head_dim, position, positions, position_ids, attn_factor, (const half_t*) q_norm, (const half_t*) k_norm, norm_eps, norm_constant_bias);
    return exl3_check_launch("rope");
}

// ------------------------------------------------------------------------------------------------
// The reference's whole argument list (rope.cu:16-65, host side :307-470): rotated width p = 2 * (inv_freq columns) < head_dim ("partial rotary"),
// up to four rotated sub-ranges with their own position each (`rotate_dims`, 3-D position ids: multimodal rope), a start offset, an angle TABLE
// instead of frequencies, NANOCHAT sign convention, bf16 head-norm weights, an unweighted RMSNorm AFTER the rotation, the llama-4 position scale on
// query heads, and a head stride wider than head_dim (q / k as the trailing columns of wider heads).
// One workgroup per token; the sin / cos of every (sub-range, pair) are evaluated once into LDS; one wave per head at a time.  A head lives in a
// wave-private fp16 LDS line between the stages because the stages address it differently (norm / scale / store: lane owns elements 2l, 2l + 1 of
// every 128; rotation: lane owns pair t of a sub-range, whose partner sits p/2 away) and because the reference rounds to fp16 between them.
// ------------------------------------------------------------------------------------------------
struct RopeExArgs
{
    const half_t* q; half_t* out_q; const half_t* k; half_t* out_k; const float* inv_freq;
    int seq_len, heads_q, heads_k, head_dim, q_head_stride, k_head_stride, p2;
    uint32_t position; const int32_t* positions; const int32_t* position_ids; int position_ids_stride;
    float attn_factor; const void* q_norm; const void* k_norm; float norm_eps, norm_constant_bias;
    int inv_freq_table, inv_freq_stride; float l4_beta; int l4_orig; int post_rope_norm, rotate_dims, rotate_offset;
};

template <int MODE, bool NORM_BF16>     // 1 = GPTJ, 2 = NEOX, 3 = NANOCHAT
__global__ __launch_bounds__(256)
void rope_ex_kernel(const RopeExArgs a)
{
    __shared__ float sn_s[4 * 256], cs_s[4 * 256];
    __shared__ __attribute__((aligned(16))) half_t head_s[4][512];
    __shared__ float l4_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int token = blockIdx.x, batch = blockIdx.y;
    const int64_t tok = (int64_t) batch * a.seq_len + token;
    auto pos_of = [&] (int rdim) -> int
    {
        if (a.positions) return token + a.positions[batch];
        if (a.position_ids) return a.position_ids[tok * a.position_ids_stride + (a.position_ids_stride > 1 ? rdim : 0)];
        return token + (int) a.position;
    };
    for (int e = threadIdx.x; e < a.rotate_dims * a.p2; e += blockDim.x)
    {
        const int rdim = e / a.p2, t = e - rdim * a.p2;
        const int pos = pos_of(rdim);
        const float ang = a.inv_freq_table ? a.inv_freq[(int64_t) batch * a.inv_freq_stride + (int64_t) pos * a.p2 + t] : a.inv_freq[t] * (float) pos;
        float sn, cs;
        sincosf(ang, &sn, &cs);
        sn_s[rdim * 256 + t] = sn * a.attn_factor; cs_s[rdim * 256 + t] = cs * a.attn_factor;
    }
    if (threadIdx.x == 0) l4_s = a.l4_beta > 0.0f ? 1.0f + a.l4_beta * __logf(1.0f + (float) (pos_of(0) / a.l4_orig)) : 1.0f;
    __syncthreads();
    const float l4 = l4_s;
    half_t* hb = head_s[wave];
    const int heads = a.heads_q + a.heads_k;
    for (int head = wave; head < heads; head += 4)
    {
        const bool is_q = head < a.heads_q;
        const int hi = is_q ? head : head - a.heads_q;
        const half_t* src = is_q ? a.q + (tok * a.heads_q + hi) * a.q_head_stride : a.k + (tok * a.heads_k + hi) * a.k_head_stride;
        half_t* dst = is_q ? a.out_q + (tok * a.heads_q + hi) * a.q_head_stride : a.out_k + (tok * a.heads_k + hi) * a.k_head_stride;
        const void* nw = is_q ? a.q_norm : a.k_norm;
        // ---- load (+ head norm)
        float v[8];
        float ss = 0.0f;
        #pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const int e = 2 * lane + 128 * i;
            v[2 * i] = 0.f; v[2 * i + 1] = 0.f;
            if (e < a.head_dim) { const half2_t h = *(const half2_t*) (src + e); v[2 * i] = (float) h[0]; v[2 * i + 1] = (float) h[1]; }
            ss += v[2 * i] * v[2 * i] + v[2 * i + 1] * v[2 * i + 1];
        }
        if (a.q_norm)
        {
            #pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss += xor_lane(ss, o);
            const float rmf = __frsqrt_rn(ss / (float) a.head_dim + a.norm_eps);
            const half_t bias_h = f2h(a.norm_constant_bias);
            #pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                const int e = 2 * lane + 128 * i;
                if (e < a.head_dim)
                {
                    #pragma unroll
                    for (int j = 0; j < 2; ++j)
                    {
                        if constexpr (NORM_BF16)
                        {
                            const uint32_t wb = (uint32_t) ((const uint16_t*) nw)[e + j] << 16;
                            float w; __builtin_memcpy(&w, &wb, 4);
                            v[2 * i + j] = (float) f2h((v[2 * i + j] * rmf) * (w + a.norm_constant_bias));
                        }
                        else
                        {
                            const half_t w = ((const half_t*) nw)[e + j] + bias_h;
                            v[2 * i + j] = (float) (w * f2h(v[2 * i + j] * rmf));
                        }
                    }
                }
            }
        }
        #pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const int e = 2 * lane + 128 * i;
            if (e < a.head_dim) { half2_t h; h[0] = f2h(v[2 * i]); h[1] = f2h(v[2 * i + 1]); *(half2_t*) (hb + e) = h; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- rotation, one sub-range after the other (disjoint element ranges; pairs within a sub-range are disjoint)
        for (int rdim = 0; rdim < a.rotate_dims; ++rdim)
        {
            const int off = a.rotate_offset + 2 * a.p2 * rdim;
            for (int t = lane; t < a.p2; t += 64)
            {
                const int i1 = off + (MODE == 1 ? 2 * t : t), i2 = off + (MODE == 1 ? 2 * t + 1 : t + a.p2);
                const float sn = MODE == 3 ? -sn_s[rdim * 256 + t] : sn_s[rdim * 256 + t], cs = cs_s[rdim * 256 + t];
                const float v1 = (float) hb[i1], v2 = (float) hb[i2];
                hb[i1] = f2h(v1 * cs - v2 * sn);
                hb[i2] = f2h(v2 * cs + v1 * sn);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- unweighted norm after the rotation, llama-4 scale (query heads), store
        ss = 0.0f;
        #pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const int e = 2 * lane + 128 * i;
            v[2 * i] = 0.f; v[2 * i + 1] = 0.f;
            if (e < a.head_dim) { const half2_t h = *(const half2_t*) (hb + e); v[2 * i] = (float) h[0]; v[2 * i + 1] = (float) h[1]; }
            ss += v[2 * i] * v[2 * i] + v[2 * i + 1] * v[2 * i + 1];
        }
        if (a.post_rope_norm)
        {
            #pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss += xor_lane(ss, o);
            const float rmf = __frsqrt_rn(ss / (float) a.head_dim + a.norm_eps);
            #pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float) f2h(v[j] * rmf);
        }
        if (is_q && l4 != 1.0f)
        {
            #pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float) f2h(v[j] * l4);
        }
        #pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const int e = 2 * lane + 128 * i;
            if (e < a.head_dim) { half2_t h; h[0] = f2h(v[2 * i]); h[1] = f2h(v[2 * i + 1]); *(half2_t*) (dst + e) = h; }
        }
        __builtin_amdgcn_wave_barrier();          // the line is rewritten by this wave's next head
    }
}

extern "C" int exl3_rope_ex(const void* q, void* out_q, const void* k, void* out_k, const float* inv_freq,
                            int bsz, int seq_len, int heads_q, int heads_k, int head_dim, int q_head_stride, int k_head_stride, int partial_head_dim,
                            uint32_t position, const int32_t* positions, const int32_t* position_ids, int position_ids_stride,
                            int rope_mode, float attn_factor, const void* q_norm, const void* k_norm, int norm_bf16, float norm_eps, float norm_constant_bias,
                            int inv_freq_table, int inv_freq_stride, float l4_beta, int l4_orig, int post_rope_norm, int rotate_dims, int rotate_offset,
                            void* stream)
{
    EXL3_CHECK_ARG(q && out_q && inv_freq, "rope: null pointer");
    EXL3_CHECK_ARG(heads_k == 0 || (k && out_k), "rope: k given without out_k");
    EXL3_CHECK_ARG(head_dim % 2 == 0 && head_dim > 0 && head_dim <= 512, "rope: head_dim must be even and <= 512");
    EXL3_CHECK_ARG(rope_mode >= 1 && rope_mode <= 3, "rope: rope_mode must be 1 (GPTJ), 2 (NEOX) or 3 (NANOCHAT)");
    EXL3_CHECK_ARG(partial_head_dim > 0 && partial_head_dim % 2 == 0, "rope: rotated width must be even");
    EXL3_CHECK_ARG(rotate_dims > 0 && rotate_dims <= 4, "rotate_dims out of range");                                             // rope.cu:361
    EXL3_CHECK_ARG(rotate_dims == 1 || head_dim == partial_head_dim * rotate_dims, "rotate_dims is inconsistent with inv_freq and head_dim");   // :362
    EXL3_CHECK_ARG(rotate_offset >= 0 && rotate_offset + partial_head_dim * rotate_dims <= head_dim, "rotate_offset out of range");            // :363
    EXL3_CHECK_ARG(!(positions && position_ids), "rope: positions and position_ids are mutually exclusive");                             // :403
    EXL3_CHECK_ARG(position_ids_stride == 1 || (position_ids && position_ids_stride == rotate_dims), "rope: position_ids stride must be 1 or rotate_dims");
    EXL3_CHECK_ARG(q_head_stride >= head_dim && q_head_stride % 2 == 0 && (heads_k == 0 || (k_head_stride >= head_dim && k_head_stride % 2 == 0)),
                   "rope: head strides must be even and >= head_dim");
    EXL3_CHECK_ARG(!q_norm == !k_norm || heads_k == 0, "rope: q_norm and k_norm must be given together");
    EXL3_CHECK_ARG(l4_beta <= 0.0f || l4_orig > 0, "rope: llama_4_scaling_original must be positive");
    if (bsz == 0 || seq_len == 0) return EXL3_OK;
    RopeExArgs a;
    a.q = (const half_t*) q; a.out_q = (half_t*) out_q; a.k = (const half_t*) k; a.out_k = (half_t*) out_k; a.inv_freq = inv_freq;
    a.seq_len = seq_len; a.heads_q = heads_q; a.heads_k = heads_k; a.head_dim = head_dim; a.q_head_stride = q_head_stride; a.k_head_stride = k_head_stride;
    a.p2 = partial_head_dim / 2; a.position = position; a.positions = positions; a.position_ids = position_ids; a.position_ids_stride = position_ids_stride;
    a.attn_factor = attn_factor; a.q_norm = q_norm; a.k_norm = k_norm; a.norm_eps = norm_eps; a.norm_constant_bias = norm_constant_bias;
    a.inv_freq_table = inv_freq_table; a.inv_freq_stride = inv_freq_stride; a.l4_beta = l4_beta; a.l4_orig = l4_orig > 0 ? l4_orig : 1;
    a.post_rope_norm = post_rope_norm; a.rotate_dims = rotate_dims; a.rotate_offset = rotate_offset;
    dim3 grid(seq_len, bsz, 1);
    hipStream_t st = (hipStream_t) stream;
    const bool bf = norm_bf16 && q_norm;
    if (rope_mode == 1)      { if (bf) rope_ex_kernel<1, true><<<grid, 256, 0, st>>>(a); else rope_ex_kernel<1, false><<<grid, 256, 0, st>>>(a); }
    else if (rope_mode == 2) { if (bf) rope_ex_kernel<2, true><<<grid, 256, 0, st>>>(a); else rope_ex_kernel<2, false><<<grid, 256, 0, st>>>(a); }
    else                     { if (bf) rope_ex_kernel<3, true><<<grid, 256, 0, st>>>(a); else rope_ex_kernel<3, false><<<grid, 256, 0, st>>>(a); }
    return exl3_check_launch("rope_ex");
}

// In-place NEOX rope (head_dim 128, no head norm) on q / k that are column ranges of a wider row-major matrix: ld_q, ld_k = halves per token.
// Used by the prefill route's fused q|k|v GEMM (linear.LinearEXL3.forward_multi); same kernel as exl3_rope's fast path.
extern "C" int exl3_rope_strided(void* q, void* k, const float* inv_freq, int bsz, int seq_len, int heads_q, int heads_k, int64_t ld_q, int64_t ld_k,
                                 uint32_t position, const int32_t* positions, const int32_t* position_ids, float attn_factor, void* stream)
{
    EXL3_CHECK_ARG(q && inv_freq && (heads_k == 0 || k), "rope_strided: null pointer");
    EXL3_CHECK_ARG(ld_q >= (int64_t) heads_q * 128 && ld_k >= (int64_t) heads_k * 128 && ld_q % 8 == 0 && ld_k % 8 == 0, "rope_strided: bad row strides");
    if (bsz == 0 || seq_len == 0) return EXL3_OK;
    rope_neox128_kernel<<<dim3(seq_len, bsz, 1), 256, 0, (hipStream_t) stream>>>((const half_t*) q, (half_t*) q, (const half_t*) k, (half_t*) k, inv_freq, seq_len,
                                                                                 heads_q, heads_k, position, positions, position_ids, attn_factor, ld_q, ld_k);
    return exl3_check_launch("rope_strided");
}

// ------------------------------------------------------------------------------------------------
// KV-cache quantization (quant_cache_cont / _paged, dequant_cache_cont / _paged): 32-value groups, H32 rotation, absmax scale, midpoint grid,
// power-of-two bit planes.  Group code: exl3_kvq.cuh.  These kernels map ONE GROUP TO ONE 16-LANE DPP ROW (2 consecutive values per lane), so
// a wave64 holds four groups = one 128-wide head, the butterflies / max / plane OR-reductions are DPP row operations, a group's fp16 values
// are one 64-byte row access and its scale + plane words are written by that row alone.
// ------------------------------------------------------------------------------------------------

// One 32-group per 16-lane DPP row, two consecutive values per lane, four groups (128 values) per wave: exl3_kvq.cuh KvGroup<2>.
template <int BITS>
__device__ __forceinline__ void kv_quant_group(const half_t* __restrict__ in, uint32_t* __restrict__ out, half_t* __restrict__ out_scale, bool active, int lane,
                                               float compand_a)
{
    float v[2] = { 0.0f, 0.0f };
    if (active) { const half2_t x = ((const half2_t*) in)[lane & 15]; v[0] = (float) x.x; v[1] = (float) x.y; }
    KvGroup<2>::quantize(BITS, v, out, out_scale, active, lane, compand_a);
}

template <int BITS>
__device__ __forceinline__ void kv_dequant_group(const uint32_t* __restrict__ in, const half_t* __restrict__ in_scale, half_t* __restrict__ out, bool active, int lane,
                                                 float compand_a)
{
    float u[2];
    KvGroup<2>::levels<true>(BITS, in, in_scale, lane, u, compand_a);      // inactive lanes read group 0 (valid memory), results discarded
    KvGroup<2>::hadamard(u, lane);
    if (active) ((half2_t*) out)[lane & 15] = half2_t{ f2h(u[0]), f2h(u[1]) };
}

// contiguous: group g of the flat tensor
template <int BITS>
__global__ __launch_bounds__(256)
void kv_quant_cont_kernel(const half_t* __restrict__ in, uint32_t* __restrict__ out, half_t* __restrict__ scales, int64_t num_groups, float compand_a)
{
    int64_t g = (int64_t) blockIdx.x * 16 + (threadIdx.x >> 4);
    bool active = g < num_groups;
    int64_t gs = active ? g : 0;
    kv_quant_group<BITS>(in + gs * 32, out + gs * BITS, scales + gs, active, threadIdx.x & 63, compand_a);
}

template <int BITS>
__global__ __launch_bounds__(256)
void kv_dequant_cont_kernel(const uint32_t* __restrict__ in, const half_t* __restrict__ scales, half_t* __restrict__ out, int64_t num_groups, float compand_a)
{
    int64_t g = (int64_t) blockIdx.x * 16 + (threadIdx.x >> 4);
    bool active = g < num_groups;
    int64_t gs = active ? g : 0;
    kv_dequant_group<BITS>(in + gs * BITS, scales + gs, out + gs * 32, active, threadIdx.x & 63, compand_a);
}

// paged append: grid (ceil(groups_per_token/16), seq_len, bsz); K and V in one launch (blockIdx.x parity split would
// halve occupancy; instead each thread-group does K then V like the reference, q_cache_kernels.cuh:291-326)
template <int KB, int VB>
__global__ __launch_bounds__(256)
void kv_quant_paged_kernel(const half_t* __restrict__ k_in, uint32_t* __restrict__ k_out, half_t* __restrict__ k_scales,
                           const half_t* __restrict__ v_in, uint32_t* __restrict__ v_out, half_t* __restrict__ v_scales,
                           const int32_t* __restrict__ cache_seqlens, const int32_t* __restrict__ block_table,
                           int blocks_per_seq, int page_size, int groups_per_token, int64_t ld_k, int64_t ld_v, float compand_a, int in_contiguous)
{
    // ld_k / ld_v: halves between consecutive input tokens (groups_per_token * 32 when contiguous)
    const int batch = blockIdx.z;
    const int token_idx = blockIdx.y + cache_seqlens[batch];
    const int page_idx = token_idx / page_size;
    const int64_t token_pos = (int64_t) block_table[blocks_per_seq * batch + page_idx] * page_size + (token_idx % page_size);
    // in_contiguous = 0: k_in / v_in are a flat fp16 staging cache read at the token's own physical row (q_cache_kernels.cuh:316-318)
    const int64_t in_pos = in_contiguous ? (int64_t) batch * gridDim.y + blockIdx.y : token_pos;
    const int g = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool active = g < groups_per_token;
    const int gs = active ? g : 0;
    const int64_t base = token_pos * groups_per_token + gs;
    const int lane = threadIdx.x & 63;
    kv_quant_group<KB>(k_in + in_pos * ld_k + gs * 32, k_out + base * KB, k_scales + base, active, lane, compand_a);
    kv_quant_group<VB>(v_in + in_pos * ld_v + gs * 32, v_out + base * VB, v_scales + base, active, lane, compand_a);
}

// paged dequant of the cached tokens: grid (ceil(groups_per_token/16), max_tokens, bsz).  Which rows are written follows the reference kernel
// (cache/q_cache_kernels.cuh:342-399): logical positions [0, cache_seqlens[b] + bonus_len); with sliding_window > 0 the reference skips whole
// thread blocks of 256 "chunks" (a chunk = 4 groups of one token, ceil(groups_per_token / 4) chunks per token) that end at or before
// max_len - sliding_window -- the same test is applied per group here, so exactly the same rows are left untouched; compact_out writes page p of
// sequence b densely at row (b * blocks_per_seq + p) * page_size instead of the physical page (dequant_cache_paged_window's scratch layout).
template <int KB, int VB>
__global__ __launch_bounds__(256)
void kv_dequant_paged_kernel(const uint32_t* __restrict__ k_in, const half_t* __restrict__ k_scales, half_t* __restrict__ k_out,
                             const uint32_t* __restrict__ v_in, const half_t* __restrict__ v_scales, half_t* __restrict__ v_out,
                             const int32_t* __restrict__ cache_seqlens, const int32_t* __restrict__ block_table,
                             int blocks_per_seq, int page_size, int groups_per_token, int sliding_window, float compand_a, int compact_out, int bonus_len)
{
    const int batch = blockIdx.z;
    const int token_idx = blockIdx.y;
    const int max_token_idx = cache_seqlens[batch] + bonus_len;
    if (token_idx >= max_token_idx) return;
    const int g = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool in_range = g < groups_per_token;
    const int gs = in_range ? g : 0;
    bool active = in_range;
    if (sliding_window > 0)
    {
        const int chunks_per_token = (groups_per_token + 3) >> 2;
        const int64_t chunk_id = (int64_t) token_idx * chunks_per_token + (gs >> 2);
        const int64_t next_block_chunk = ((chunk_id >> 8) + 1) << 8;               // 8 warps x 32 iterations of the reference's thread block
        active = active && !((int) (next_block_chunk / chunks_per_token) <= max_token_idx - sliding_window);
    }
    const int page_idx = token_idx / page_size;
    const int64_t token_pos = (int64_t) block_table[blocks_per_seq * batch + page_idx] * page_size + (token_idx % page_size);
    const int64_t out_pos = compact_out ? ((int64_t) batch * blocks_per_seq + page_idx) * page_size + (token_idx % page_size) : token_pos;
    const int64_t base = token_pos * groups_per_token + gs, base_out = out_pos * groups_per_token + gs;
    const int lane = threadIdx.x & 63;
    kv_dequant_group<KB>(k_in + base * KB, k_scales + base, k_out + base_out * 32, active, lane, compand_a);
    kv_dequant_group<VB>(v_in + base * VB, v_scales + base, v_out + base_out * 32, active, lane, compand_a);
}

#define BITS_SWITCH(b, CALL) switch (b) { \
    case 2: { constexpr int BB = 2; CALL; } break; case 3: { constexpr int BB = 3; CALL; } break; case 4: { constexpr int BB = 4; CALL; } break; \
    case 5: { constexpr int BB = 5; CALL; } break; case 6: { constexpr int BB = 6; CALL; } break; case 7: { constexpr int BB = 7; CALL; } break; \
    case 8: { constexpr int BB = 8; CALL; } break; }

extern "C" int exl3_quant_cache_cont(const void* in, void* out, void* out_scales, int64_t tokens, int dim, int bits, void* stream)
{
    return exl3_quant_cache_cont_ex(in, out, out_scales, tokens, dim, bits, 0.0f, stream);
}

static bool compand_ok(float a) { return a == 0.0f || (a > 0.0f && a < 1.0f); }      // (1 - a) is a divisor in the encoder

extern "C" int exl3_quant_cache_cont_ex(const void* in, void* out, void* out_scales, int64_t tokens, int dim, int bits, float compand_a, void* stream)
{
    EXL3_CHECK_ARG(compand_ok(compand_a), "quant_cache_cont: compand_a must be 0 (off) or in (0, 1)");
    EXL3_CHECK_ARG(in && out && out_scales, "quant_cache_cont: null pointer");
    EXL3_CHECK_ARG(dim % 32 == 0, "quant_cache_cont: dim must be divisible by 32");
    EXL3_CHECK_ARG(bits >= 2 && bits <= 8, "quant_cache_cont: bits must be in [2, 8]");
    int64_t groups = tokens * (dim / 32);
    if (groups == 0) return EXL3_OK;
    dim3 grid((unsigned) ((groups + 15) / 16));
    BITS_SWITCH(bits, (kv_quant_cont_kernel<BB><<<grid, 256, 0, (hipStream_t) stream>>>((const half_t*) in, (uint32_t*) out, (half_t*) out_scales, groups, compand_a)));
    return exl3_check_launch("quant_cache_cont");
}

extern "C" int exl3_dequant_cache_cont(const void* in, const void* in_scales, void* out, int64_t tokens, int dim, int bits, void* stream)
{
    return exl3_dequant_cache_cont_ex(in, in_scales, out, tokens, dim, bits, 0.0f, stream);
}

extern "C" int exl3_dequant_cache_cont_ex(const void* in, const void* in_scales, void* out, int64_t tokens, int dim, int bits, float compand_a, void* stream)
{
    EXL3_CHECK_ARG(compand_ok(compand_a), "dequant_cache_cont: compand_a must be 0 (off) or in (0, 1)");
    EXL3_CHECK_ARG(in && out && in_scales, "dequant_cache_cont: null pointer");
    EXL3_CHECK_ARG(dim % 32 == 0, "dequant_cache_cont: dim must be divisible by 32");
    EXL3_CHECK_ARG(bits >= 2 && bits <= 8, "dequant_cache_cont: bits must be in [2, 8]");
    int64_t groups = tokens * (dim / 32);
    if (groups == 0) return EXL3_OK;
    dim3 grid((unsigned) ((groups + 15) / 16));
    BITS_SWITCH(bits, (kv_dequant_cont_kernel<BB><<<grid, 256, 0, (hipStream_t) stream>>>((const uint32_t*) in, (const half_t*) in_scales, (half_t*) out, groups, compand_a)));
    return exl3_check_launch("dequant_cache_cont");
}

template <int KB>
static void launch_quant_paged(int vb, dim3 grid, hipStream_t st, const void* k_in, void* k_out, void* k_scales, const void* v_in, void* v_out, void* v_scales,
                               const int32_t* sl, const int32_t* bt, int bps, int ps, int gpt, int64_t ld_k, int64_t ld_v, float ca, int in_cont)
{
    BITS_SWITCH(vb, (kv_quant_paged_kernel<KB, BB><<<grid, 256, 0, st>>>((const half_t*) k_in, (uint32_t*) k_out, (half_t*) k_scales, (const half_t*) v_in,
                                                                         (uint32_t*) v_out, (half_t*) v_scales, sl, bt, bps, ps, gpt, ld_k, ld_v, ca, in_cont)));
}

template <int KB>
static void launch_dequant_paged(int vb, dim3 grid, hipStream_t st, const void* k_in, const void* k_scales, void* k_out, const void* v_in, const void* v_scales, void* v_out,
                                 const int32_t* sl, const int32_t* bt, int bps, int ps, int gpt, int sw, float ca, int compact, int bonus)
{
    BITS_SWITCH(vb, (kv_dequant_paged_kernel<KB, BB><<<grid, 256, 0, st>>>((const uint32_t*) k_in, (const half_t*) k_scales, (half_t*) k_out, (const uint32_t*) v_in,
                                                                           (const half_t*) v_scales, (half_t*) v_out, sl, bt, bps, ps, gpt, sw, ca, compact, bonus)));
}

static int quant_cache_paged_impl(const void* k_in, void* k_out, void* k_scales, const void* v_in, void* v_out, void* v_scales,
                                  const int32_t* cache_seqlens, const int32_t* block_table, int bsz, int blocks_per_seq,
                                  int page_size, int seq_len, int dim, int k_bits, int v_bits, int64_t ld_k, int64_t ld_v, float compand_a, int in_contiguous, void* stream)
{
    EXL3_CHECK_ARG(compand_ok(compand_a), "quant_cache_paged: compand_a must be 0 (off) or in (0, 1)");
    EXL3_CHECK_ARG(k_in && k_out && k_scales && v_in && v_out && v_scales && cache_seqlens && block_table, "quant_cache_paged: null pointer");
    EXL3_CHECK_ARG(dim % 32 == 0 && page_size > 0, "quant_cache_paged: dim must be divisible by 32");
    EXL3_CHECK_ARG(k_bits >= 2 && k_bits <= 8 && v_bits >= 2 && v_bits <= 8, "quant_cache_paged: bits must be in [2, 8]");
    EXL3_CHECK_ARG(ld_k >= dim && ld_v >= dim && ld_k % 4 == 0 && ld_v % 4 == 0, "quant_cache_paged: bad token strides");
    if (bsz == 0 || seq_len == 0) return EXL3_OK;
    const int gpt = dim / 32;
    dim3 grid((gpt + 15) / 16, seq_len, bsz);
    hipStream_t st = (hipStream_t) stream;
    #define QP(KBv) case KBv: launch_quant_paged<KBv>(v_bits, grid, st, k_in, k_out, k_scales, v_in, v_out, v_scales, cache_seqlens, block_table, blocks_per_seq, page_size, gpt, ld_k, ld_v, compand_a, in_contiguous); break;
    switch (k_bits) { QP(2) QP(3) QP(4) QP(5) QP(6) QP(7) QP(8) }
    #undef QP
    return exl3_check_launch("quant_cache_paged");
}

extern "C" int exl3_quant_cache_paged(const void* k_in, void* k_out, void* k_scales, const void* v_in, void* v_out, void* v_scales,
                                      const int32_t* cache_seqlens, const int32_t* block_table, int bsz, int blocks_per_seq,
                                      int page_size, int seq_len, int dim, int k_bits, int v_bits, void* stream)
{
    return quant_cache_paged_impl(k_in, k_out, k_scales, v_in, v_out, v_scales, cache_seqlens, block_table, bsz, blocks_per_seq, page_size, seq_len, dim,
                                  k_bits, v_bits, dim, dim, 0.0f, 1, stream);
}

// ... with the reference's optional level compander (cache/lmq.cuh; quant.py passes compand_a to every cache op) and explicit token strides
extern "C" int exl3_quant_cache_paged_ex(const void* k_in, void* k_out, void* k_scales, const void* v_in, void* v_out, void* v_scales,
                                         const int32_t* cache_seqlens, const int32_t* block_table, int bsz, int blocks_per_seq,
                                         int page_size, int seq_len, int dim, int k_bits, int v_bits, int64_t ld_k, int64_t ld_v, float compand_a, int in_contiguous,
                                         void* stream)
{
    return quant_cache_paged_impl(k_in, k_out, k_scales, v_in, v_out, v_scales, cache_seqlens, block_table, bsz, blocks_per_seq, page_size, seq_len, dim,
                                  k_bits, v_bits, ld_k, ld_v, compand_a, in_contiguous, stream);
}

// k_in / v_in are column ranges of a wider row-major matrix: ld_k, ld_v = halves per token (the prefill route's fused q|k|v GEMM output)
extern "C" int exl3_quant_cache_paged_strided(const void* k_in, void* k_out, void* k_scales, const void* v_in, void* v_out, void* v_scales,
                                              const int32_t* cache_seqlens, const int32_t* block_table, int bsz, int blocks_per_seq,
                                              int page_size, int seq_len, int dim, int k_bits, int v_bits, int64_t ld_k, int64_t ld_v, void* stream)
{
    return quant_cache_paged_impl(k_in, k_out, k_scales, v_in, v_out, v_scales, cache_seqlens, block_table, bsz, blocks_per_seq, page_size, seq_len, dim,
                                  k_bits, v_bits, ld_k, ld_v, 0.0f, 1, stream);
}

extern "C" int exl3_dequant_cache_paged(const void* k_in, const void* k_scales, void* k_out, const void* v_in, const void* v_scales, void* v_out,
                                        const int32_t* cache_seqlens, const int32_t* block_table, int bsz, int blocks_per_seq,
                                        int page_size, int dim, int k_bits, int v_bits, void* stream)
{
    return exl3_dequant_cache_paged_ex(k_in, k_scales, k_out, v_in, v_scales, v_out, cache_seqlens, block_table, bsz, blocks_per_seq, page_size, dim,
                                       k_bits, v_bits, 0, 0.0f, 0, 0, stream);
}

// sliding_window > 0: rows the reference's dequant_cache_paged leaves untouched before the window stay untouched here (same rule, see the kernel);
// compact_out + bonus_len: dequant_cache_paged_window (dense per-sequence scratch, rows up to cache_seqlens + bonus_len)
extern "C" int exl3_dequant_cache_paged_ex(const void* k_in, const void* k_scales, void* k_out, const void* v_in, const void* v_scales, void* v_out,
                                           const int32_t* cache_seqlens, const int32_t* block_table, int bsz, int blocks_per_seq,
                                           int page_size, int dim, int k_bits, int v_bits, int sliding_window, float compand_a, int compact_out,
                                           int bonus_len, void* stream)
{
    EXL3_CHECK_ARG(compand_ok(compand_a), "dequant_cache_paged: compand_a must be 0 (off) or in (0, 1)");
    EXL3_CHECK_ARG(bonus_len >= 0, "dequant_cache_paged: bonus_len must be >= 0");
    EXL3_CHECK_ARG(k_in && k_out && k_scales && v_in && v_out && v_scales && cache_seqlens && block_table, "dequant_cache_paged: null pointer");
    EXL3_CHECK_ARG(dim % 32 == 0 && page_size > 0, "dequant_cache_paged: dim must be divisible by 32");
    EXL3_CHECK_ARG(k_bits >= 2 && k_bits <= 8 && v_bits >= 2 && v_bits <= 8, "dequant_cache_paged: bits must be in [2, 8]");
    if (bsz == 0 || blocks_per_seq == 0) return EXL3_OK;
    const int gpt = dim / 32;
    dim3 grid((gpt + 15) / 16, blocks_per_seq * page_size, bsz);
    hipStream_t st = (hipStream_t) stream;
    #define DP(KBv) case KBv: launch_dequant_paged<KBv>(v_bits, grid, st, k_in, k_scales, k_out, v_in, v_scales, v_out, cache_seqlens, block_table, blocks_per_seq, page_size, gpt, sliding_window, compand_a, compact_out, bonus_len); break;
    switch (k_bits) { DP(2) DP(3) DP(4) DP(5) DP(6) DP(7) DP(8) }
    #undef DP
    return exl3_check_launch("dequant_cache_paged");
}
