// Lane-local EXL3 decode helpers shared by the GEMV (exl3_gemv2.kspec.hip) and reconstruct_had kernels.
//
// Lane (8T + c) of a wave owns words [(8T + c)K, +K) of a tile row = the 32 weights with stream indices [32c, 32c+32) of tile T
// = columns c and c+8 (16 rows each).  Weight t = 8q + j: rows {2q, 2q+1, 2q+8, 2q+9}[j & 3], column c + 8 (j >> 2).
#pragma once
#include "exl3_common.cuh"

// ---- compile-time bit-window extraction ----------------------------------------------------------------
// Wx[0] = previous lane's last word (carry-in), Wx[1..K] = the lane's K words.  Weight t (0..31): window of 16 bits
// ending at extended-stream bit 32 + (t+1)K.
template <int K, int T>
__device__ __forceinline__ uint32_t lane_state(const uint32_t (&Wx)[K + 1])
{
    constexpr int e = 32 + (T + 1) * K;          // exclusive end, 33..288
    constexpr int lo = (e - 1) >> 5;
    constexpr int hi = (e - 16) >> 5;
    constexpr int sh = 32 * (lo + 1) - e;        // 0..31
    if constexpr (hi == lo)
    {
        if constexpr (sh == 0) return Wx[lo] & 0xffffu;
        else if constexpr (sh == 16) return Wx[lo] >> 16;
        else return __builtin_amdgcn_ubfe(Wx[lo], sh, 16);
    }
    // a window that straddles two words (sh > 16): all such windows of one word boundary lie inside the middle 32 bits of the pair, so they share
    // ONE v_alignbit (common subexpression) and cost one v_bfe each, instead of an alignbit + mask per window
    else return __builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(Wx[hi], Wx[lo], 16), sh - 16, 16);
}

template <int K> struct LaneWords { uint32_t w[K]; };

// EXL3_LOAD_PLAIN (A/B builds, tools/experiments/build_lite.py): the weight rows with the default cache policy instead of non-temporal loads
#ifdef EXL3_LOAD_PLAIN
#define EXL3_WLOAD(ptr) (*(ptr))
#else
#define EXL3_WLOAD(ptr) __builtin_nontemporal_load(ptr)
#endif

template <int K>
__device__ __forceinline__ void load_lane_words(LaneWords<K>& d, const uint32_t* __restrict__ p)
{
    // p = this lane's first word of the tile row; K consecutive words (contiguous across the wave)
    if constexpr (K == 4) { uint4_t v = EXL3_WLOAD((const uint4_t*) p); d.w[0] = v.x; d.w[1] = v.y; d.w[2] = v.z; d.w[3] = v.w; }
    else if constexpr (K == 8)
    {
        uint4_t v = EXL3_WLOAD((const uint4_t*) p), u = EXL3_WLOAD((const uint4_t*) p + 1);
        d.w[0] = v.x; d.w[1] = v.y; d.w[2] = v.z; d.w[3] = v.w; d.w[4] = u.x; d.w[5] = u.y; d.w[6] = u.z; d.w[7] = u.w;
    }
    else if constexpr (K == 2) { uint2_t v = EXL3_WLOAD((const uint2_t*) p); d.w[0] = v.x; d.w[1] = v.y; }
    else
    {
        #pragma unroll
        for (int i = 0; i < K; ++i) d.w[i] = EXL3_WLOAD(p + i);
    }
}

__device__ __forceinline__ half4_t u2_as_half4(uint32_t a, uint32_t b)
{
    union { uint32_t u[2]; half4_t h; } c; c.u[0] = a; c.u[1] = b; return c.h;
}

// Decode the 4 weights T0..T0+3 into MFMA B operands.  out[0] (and out[1] for SPLIT) are half4 operands.
template <int K, int CB, int VAR, int T0>
__device__ __forceinline__ void decode_quad(const uint32_t (&Wx)[K + 1], half4_t (&out)[2])
{
    uint32_t x0 = cb_product<CB>(lane_state<K, T0 + 0>(Wx));
    uint32_t x1 = cb_product<CB>(lane_state<K, T0 + 1>(Wx));
    uint32_t x2 = cb_product<CB>(lane_state<K, T0 + 2>(Wx));
    uint32_t x3 = cb_product<CB>(lane_state<K, T0 + 3>(Wx));
    if constexpr (CB != EXL3_CB_MUL1)
    {
        x0 = cb_mask3inst(x0); x1 = cb_mask3inst(x1); x2 = cb_mask3inst(x2); x3 = cb_mask3inst(x3);
        if constexpr (VAR == 1) { out[0] = u2_as_half4(x0, x1); out[1] = u2_as_half4(x2, x3); }
        else
        {
            half2_t s01 = u32_as_half2(__builtin_amdgcn_perm(x1, x0, 0x05040100u)) + u32_as_half2(__builtin_amdgcn_perm(x1, x0, 0x07060302u));
            half2_t s23 = u32_as_half2(__builtin_amdgcn_perm(x3, x2, 0x05040100u)) + u32_as_half2(__builtin_amdgcn_perm(x3, x2, 0x07060302u));
            out[0] = u2_as_half4(half2_as_u32(s01), half2_as_u32(s23));
        }
    }
    else
    {
        uint32_t h01 = __builtin_amdgcn_sad_hi_u8(x1, 0u, __builtin_amdgcn_sad_u8(x0, 0u, 0x64006400u));
        uint32_t h23 = __builtin_amdgcn_sad_hi_u8(x3, 0u, __builtin_amdgcn_sad_u8(x2, 0u, 0x64006400u));
        if constexpr (VAR == 1) out[0] = u2_as_half4(h01, h23);
        else
        {
            const half2_t kinv = { u16_as_half(0x1eeeu), u16_as_half(0x1eeeu) };
            const half2_t kbias = { u16_as_half(0xc931u), u16_as_half(0xc931u) };
            half2_t ra = __builtin_elementwise_fma(u32_as_half2(h01), kinv, kbias);      // v_pk_fma_f16
            half2_t rb = __builtin_elementwise_fma(u32_as_half2(h23), kinv, kbias);
            out[0] = u2_as_half4(half2_as_u32(ra), half2_as_u32(rb));
        }
    }
}
