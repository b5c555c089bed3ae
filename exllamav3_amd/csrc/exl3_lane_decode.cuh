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
    else return __builtin_amdgcn_alignbit(Wx[hi], Wx[lo], sh) & 0xffffu;
}

template <int K> struct LaneWords { uint32_t w[K]; };

template <int K>
__device__ __forceinline__ void load_lane_words(LaneWords<K>& d, const uint32_t* __restrict__ p)
{
    // p = this lane's first word of the tile row; K consecutive words (contiguous across the wave)
    if constexpr (K == 4) { uint4_t v = __builtin_nontemporal_load((const uint4_t*) p); d.w[0] = v.x; d.w[1] = v.y; d.w[2] = v.z; d.w[3] = v.w; }
    else if constexpr (K == 8)
    {
        uint4_t v = __builtin_nontemporal_load((const uint4_t*) p), u = __builtin_nontemporal_load((const uint4_t*) p + 1);
        d.w[0] = v.x; d.w[1] = v.y; d.w[2] = v.z; d.w[3] = v.w; d.w[4] = u.x; d.w[5] = u.y; d.w[6] = u.z; d.w[7] = u.w;
    }
    else if constexpr (K == 2) { uint2_t v = __builtin_nontemporal_load((const uint2_t*) p); d.w[0] = v.x; d.w[1] = v.y; }
    else
    {
        #pragma unroll
        for (int i = 0; i < K; ++i) d.w[i] = __builtin_nontemporal_load(p + i);
    }
}

