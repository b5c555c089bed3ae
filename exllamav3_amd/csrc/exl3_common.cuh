// EXL3 tile format + codebooks, gfx950 device helpers.  Written for CDNA4 only (wave64, no CUDA paths).
//
// Format (reference: exllamav3_ext/quant/pack.cu:9-57, quant/exl3_dq.cuh:15-31, quant/codebook.cuh:56-90,
// modules/quant/exl3_lib/quantize.py:21-44; restated in oracle/exl3_oracle.py):
//   tile = 16x16 weights = 8K little-endian u32 words; circular MSB-first bitstream; weight t's 16-bit
//   trellis state = stream bits [(t+1)K-16, (t+1)K); t = 8l+j -> row(k) = 2(l%4)+(j&1)+8((j>>1)&1),
//   col(n) = l/4 + 8(j>>2).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float    float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t  __attribute__((ext_vector_type(2)));
typedef uint32_t uint4_t  __attribute__((ext_vector_type(4)));

#define EXL3_CB_3INST 0
#define EXL3_CB_MCG   1
#define EXL3_CB_MUL1  2

#define HAD_R_SCALE_128 0.088388347648f      // exllamav3_ext/quant/hadamard.cu:103

__device__ __forceinline__ half_t u16_as_half(uint32_t v)
{
    union { uint16_t u; half_t h; } c; c.u = (uint16_t) v; return c.h;
}
__device__ __forceinline__ uint32_t half_as_u16(half_t h)
{
    union { uint16_t u; half_t h; } c; c.h = h; return c.u;
}
__device__ __forceinline__ half2_t u32_as_half2(uint32_t v)
{
    union { uint32_t u; half2_t h; } c; c.u = v; return c.h;
}
__device__ __forceinline__ uint32_t half2_as_u32(half2_t h)
{
    union { uint32_t u; half2_t h; } c; c.h = h; return c.u;
}

// state (16 significant bits) -> 32-bit codebook product
template <int CB>
__device__ __forceinline__ uint32_t cb_product(uint32_t s)
{
    if constexpr (CB == EXL3_CB_3INST) return s * 89226354u + 64248484u;
    else if constexpr (CB == EXL3_CB_MCG) return s * 0xCBAC1FEDu;
    else return s * 0x83DCD12Du;
}

// cb0/cb1: the two fp16 halves whose sum is the weight, packed (lo16 | hi16 << 16)
__device__ __forceinline__ uint32_t cb_mask3inst(uint32_t x)
{
    return (x & 0x8fff8fffu) ^ 0x3b603b60u;
}

// Bit-exact decode: exactly one fp16 RN op per weight (v_add_f16 / v_fma_f16), as the reference's
// __hadd / __hfma (codebook.cuh:64-66,85-89).
template <int CB>
__device__ __forceinline__ half_t decode_exact(uint32_t s)
{
    uint32_t x = cb_product<CB>(s);
    if constexpr (CB != EXL3_CB_MUL1)
    {
        x = cb_mask3inst(x);
        half_t lo = u16_as_half(x & 0xffffu);
        half_t hi = u16_as_half(x >> 16);
        return lo + hi;                                   // v_add_f16, one RN rounding
    }
    else
    {
        uint32_t sum = __builtin_amdgcn_sad_u8(x, 0u, 0x6400u);       // 0x6400 + bytesum = fp16(1024 + b)
        half_t h = u16_as_half(sum);
        half_t k_inv = u16_as_half(0x1eeeu);
        half_t k_bias = u16_as_half(0xc931u);
        return __builtin_fmaf16(h, k_inv, k_bias);        // v_fma_f16, one RN rounding
    }
}

// Generic window read from a tile's words (LDS or global pointer): state of stream index t, any K in 1..8.
// Equivalent to exl3_dq.cuh:18-30 (the high word is always word(i1-1); when the window lies inside one
// word the high word is shifted out).
template <int K>
__device__ __forceinline__ uint32_t tile_state(const uint32_t* __restrict__ w, int t)
{
    constexpr int NW = 8 * K;
    int pe = (t + 1) * K - 1;                 // last bit of the window (0..256K-1)
    int i1 = pe >> 5;
    int i0 = (i1 + NW - 1) % NW;
    int sh = 31 - (pe & 31);
    uint64_t merged = ((uint64_t) w[i0] << 32) | (uint64_t) w[i1];
    return (uint32_t) (merged >> sh) & 0xffffu;
}

// stream index of tile element (row r = k offset, col c = n offset): inverse of tensor_core_perm
__device__ __forceinline__ int tile_stream_index(int r, int c)
{
    int l = 4 * (c & 7) + ((r & 7) >> 1);
    int j = (r & 1) + 2 * (r >> 3) + 4 * (c >> 3);
    return 8 * l + j;
}

// float -> half with the reference's rounding: the fp32 operation rounds to fp32 first, then RNE to fp16 (what __float2half_rn of
// an fp32 expression does in the reference kernels and what numpy does in the oracle).  Without the barrier hipcc folds
// fptrunc(fmul / fadd / fma) into v_fma_mixlo_f16, which rounds ONCE; the result differs from the two-step rounding in rare
// cases and -- because the fold depends on the surrounding code -- two kernels computing the same expression disagree by 1 ulp.
__device__ __forceinline__ half_t f2h(float v) { asm("" : "+v"(v)); return (half_t) v; }

// Butterfly exchange v[lane ^ I] for a compile-time I.  hipcc lowers __shfl_xor to ds_bpermute_b32 (LDS crossbar, ~100+ cycles of
// dependent latency per stage); the latency-bound decode kernels are chains of these, so use the register-file paths instead:
//   I = 1, 2 : one DPP quad_perm (folds into the consuming VALU op)
//   I = 4    : row_half_mirror (i ^ 7) then quad reverse (i ^ 3)
//   I = 8    : row_mirror (i ^ 15) then row_half_mirror (i ^ 7)
//   I = 16/32: gfx950 v_permlane16_swap / v_permlane32_swap + one select
__device__ __forceinline__ uint32_t xor_lane_u32(uint32_t v, const int i)
{
    #define EXL3_DPP(x, ctrl) ((uint32_t) __builtin_amdgcn_update_dpp(0, (int) (x), ctrl, 0xf, 0xf, true))
    switch (i)
    {
        case 1: return EXL3_DPP(v, 0xB1);
        case 2: return EXL3_DPP(v, 0x4E);
        case 4: { uint32_t t = EXL3_DPP(v, 0x141); return EXL3_DPP(t, 0x1B); }
        case 8: { uint32_t t = EXL3_DPP(v, 0x140); return EXL3_DPP(t, 0x141); }
        case 16:
        {
            auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
            const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            return (lane & 16) ? r[0] : r[1];
        }
        case 32:
        {
            auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
            const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            return (lane & 32) ? r[0] : r[1];
        }
        default: return (uint32_t) __shfl_xor((int) v, i, 64);
    }
    #undef EXL3_DPP
}
__device__ __forceinline__ float xor_lane(float v, const int i) { return __uint_as_float(xor_lane_u32(__float_as_uint(v), i)); }
__device__ __forceinline__ uint32_t xor_lane(uint32_t v, const int i) { return xor_lane_u32(v, i); }
__device__ __forceinline__ int xor_lane(int v, const int i) { return (int) xor_lane_u32((uint32_t) v, i); }

// 128-point Sylvester Hadamard over one 32-lane half-wave, 4 elements (4t..4t+3) per lane, fp32.
// Stage order matches the reference (hadamard_inner.cuh:117-131: in-lane H4, then lane bits 0..4).
__device__ __forceinline__ void had128_f32x4(float& h0, float& h1, float& h2, float& h3, int lane32)
{
    float s0 = h0 + h1, d0 = h0 - h1, s1 = h2 + h3, d1 = h2 - h3;
    h0 = s0 + s1; h1 = d0 + d1; h2 = s0 - s1; h3 = d0 - d1;
    #pragma unroll
    for (int i = 1; i < 32; i <<= 1)
    {
        float p0 = xor_lane(h0, i);
        float p1 = xor_lane(h1, i);
        float p2 = xor_lane(h2, i);
        float p3 = xor_lane(h3, i);
        bool neg = (lane32 & i) != 0;
        h0 = (neg ? -h0 : h0) + p0;
        h1 = (neg ? -h1 : h1) + p1;
        h2 = (neg ? -h2 : h2) + p2;
        h3 = (neg ? -h3 : h3) + p3;
    }
}
