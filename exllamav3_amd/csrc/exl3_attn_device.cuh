// Device helpers of the matrix-pipe decode attention (exl3_attn_decode.hip: attn_decode_wide_kernel; exl3_pstep_kernel.cuh: the attention item inside the persistent
// decode step): the V tile's LDS pitch, the transposed LDS read of the value product's B operand, the 4-bit cache words -> fp16 dequantization in pair order.
#pragma once
#include "exl3_common.cuh"

#define ATT_R32 0.17677669529663688110f
#define AW_VS 136          // V tile row stride in halves (272 B: the transpose-read groups tile the banks)

__device__ __forceinline__ half4_t aw_tr16(const half_t* p)
{
    typedef short s16x4_t __attribute__((__vector_size__(4 * sizeof(short))));
    const s16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*) p);
    return __builtin_bit_cast(half4_t, r);
}

// 8 four-bit levels of one dword -> 8 halves (level - 7.5) * sc4 / 4 ... in pair order (n0, n4), (n1, n5), (n2, n6), (n3, n7); sc4 = 4 * scale as half2
// (mk = 0x001E001E held in a VGPR by the caller: with both constants as literals the compiler needs v_and + v_or -- one literal per instruction -- instead of v_and_or_b32)
__device__ __forceinline__ half8_t aw_dequant8(uint32_t x, half2_t sc4, uint32_t mk)
{
    const half2_t off = { u16_as_half(0xcc0fu), u16_as_half(0xcc0fu) };         // -(16 + 15/64)
    union { uint32_t u[4]; half8_t h; } r;
    #pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const uint32_t sh = i == 0 ? (x << 1) : (x >> (4 * i - 1));
        const uint32_t m = (sh & mk) | 0x4C004C00u;
        const half2_t v = (u32_as_half2(m) + off) * sc4;
        r.u[i] = half2_as_u32(v);
    }
    return r.h;
}

