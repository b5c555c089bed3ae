// Microbenchmark for the "LDS codebook LUT" question (task 2): can part of the EXL3 decode move from the VALU to the LDS pipe?
//
// A 65536-entry fp16 table (128 KiB) of the mul1 codebook lives in LDS (one workgroup per CU, NW waves).  Per lane and step: 32 weights
// (K = 4: 4 words + carry-in).  Of every 8 weights, NLUT are decoded by `ds_read_u16_d16(_hi)` at byte address state * 2 (random states ->
// realistic bank conflicts) and 8 - NLUT by the VALU recipe of the shipped kernel (window extraction + v_mul_lo_u32 + v_sad_u8); every 4 weights
// feed one v_mfma_f32_4x4x4_16B_f16 like the GEMV.  Reports ns per wave-step per SIMD, the 4-bpw TB/s equivalent for 256 CUs, and the table fill time.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_lds_lut tools/ubench_lds_lut.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <type_traits>
typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
#define STEPS 512

__device__ __forceinline__ half_t decode_mul1(uint32_t s)
{
    uint32_t x = s * 0x83DCD12Du;
    uint32_t sum = __builtin_amdgcn_sad_u8(x, 0u, 0x6400u);
    union { uint16_t u; half_t h; } c, ki, kb; c.u = (uint16_t) sum; ki.u = 0x1eee; kb.u = 0xc931;
    return __builtin_fmaf16(c.h, ki.h, kb.h);
}

// state of weight j (0..7) of a word pair (hi = previous word, lo = this word), K = 4: 16-bit window ending at bit 4 (j + 1) of lo (MSB first)
template <int J> __device__ __forceinline__ uint32_t st(uint32_t hi, uint32_t lo)
{
    constexpr int sh = 32 - 4 * (J + 1);                 // 28, 24, ..., 0
    if constexpr (sh == 0) return lo & 0xffffu;
    else if constexpr (sh == 16) return lo >> 16;
    else if constexpr (sh < 16) return __builtin_amdgcn_ubfe(lo, sh, 16);
    else return __builtin_amdgcn_alignbit(hi, lo, sh) & 0xffffu;
}
// byte address state * 2 of the same window (LUT path): one op fewer than state-then-shift
template <int J> __device__ __forceinline__ uint32_t st2(uint32_t hi, uint32_t lo)
{
    constexpr int sh = 32 - 4 * (J + 1);
    if constexpr (sh == 0) return (lo << 1) & 0x1fffeu;
    else if constexpr (sh == 16) return (lo >> 15) & 0x1fffeu;
    else if constexpr (sh < 16) return (lo >> (sh - 1)) & 0x1fffeu;
    else return __builtin_amdgcn_alignbit(hi, lo, sh - 1) & 0x1fffeu;
}

template <int NLUT, int NW>
__global__ __launch_bounds__(64 * NW) void lut_mix(uint32_t* out, uint32_t seed, uint64_t* fill_clk)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* lut = (half_t*) smem;                        // LDS offset 0: byte address = state * 2
    const int tid = threadIdx.x, lane = tid & 63;
    uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    if (NLUT > 0)
    {
        for (int i = tid; i < 32768; i += 64 * NW)
        {
            half2_t v = { decode_mul1(2 * i), decode_mul1(2 * i + 1) };
            ((half2_t*) lut)[i] = v;
        }
        __syncthreads();
    }
    uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && fill_clk) fill_clk[blockIdx.x] = t1 - t0;

    float4_t acc0 = {0,0,0,0}, acc1 = {0,0,0,0};
    const uint32_t M = 0x83DCD12Du;
    const half4_t a = { (half_t) 1.0f, (half_t) 0.5f, (half_t) 0.25f, (half_t) 2.0f };
    uint32_t soft = seed * 0x9E3779B9u + tid * 0x85EBCA6Bu + blockIdx.x * 0xC2B2AE35u;
    for (int s = 0; s < STEPS; ++s)
    {
        uint32_t W[5];
        // xorshift words: random states, like real trellis data
        #pragma unroll
        for (int i = 0; i < 5; ++i) { soft ^= soft << 13; soft ^= soft >> 17; soft ^= soft << 5; W[i] = soft; }
        #pragma unroll
        for (int wd = 0; wd < 4; ++wd)
        {
            const uint32_t hi = W[wd], lo = W[wd + 1];
            uint32_t h[4];                                // 8 weights as 4 packed half pairs
            auto pair = [&] (auto jc) -> uint32_t
            {
                constexpr int j = decltype(jc)::value;   // weights j, j + 1
                if constexpr (j + 1 < NLUT)
                {
                    const uint32_t a0 = st2<j>(hi, lo), a1 = st2<j + 1>(hi, lo);
                    uint32_t r;
                    asm volatile("ds_read_u16_d16 %0, %1\n\tds_read_u16_d16_hi %0, %2" : "=&v"(r) : "v"(a0), "v"(a1));
                    return r;
                }
                else if constexpr (j < NLUT)              // mixed pair: low via LUT (exact fp16), high via VALU (raw 1024 + b): only a rate probe
                {
                    const uint32_t a0 = st2<j>(hi, lo);
                    uint32_t r;
                    asm volatile("ds_read_u16 %0, %1" : "=v"(r) : "v"(a0));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r));
                    return __builtin_amdgcn_sad_hi_u8(st<j + 1>(hi, lo) * M, 0u, r);
                }
                else return __builtin_amdgcn_sad_hi_u8(st<j + 1>(hi, lo) * M, 0u, __builtin_amdgcn_sad_u8(st<j>(hi, lo) * M, 0u, 0x64006400u));
            };
            h[0] = pair(std::integral_constant<int, 0>{}); h[1] = pair(std::integral_constant<int, 2>{});
            h[2] = pair(std::integral_constant<int, 4>{}); h[3] = pair(std::integral_constant<int, 6>{});
            if (NLUT > 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]));
            union { uint32_t u[2]; half4_t h; } b0, b1; b0.u[0] = h[0]; b0.u[1] = h[1]; b1.u[0] = h[2]; b1.u[1] = h[3];
            acc0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b0.h, acc0, 4, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b1.h, acc1, 4, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + tid] = (uint32_t) (acc0.x + acc1.y + acc0.z + acc1.w);
}

template <int NLUT, int NW> static void run(uint32_t* d, uint64_t* clk, int cus)
{
    const size_t lds = NLUT > 0 ? 131072 : 0;
    auto k = lut_mix<NLUT, NW>;
    hipFuncSetAttribute((const void*) k, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = cus * (NLUT > 0 ? 1 : 1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), lds, 0, d, 1u, clk); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 4; ++r)
    {
        hipEventRecord(e0, 0); hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * NW), lds, 0, d, (uint32_t) r + 2, clk);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    uint64_t c[4]; hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost);
    const double fill_us = c[0] / 100.0;                                    // s_memrealtime: 100 MHz
    const double stream_ms = best - fill_us * 1e-3;
    const double waves_per_simd = NW / 4.0;
    const double ns_step = stream_ms * 1e6 / (waves_per_simd * STEPS);       // per wave-step per SIMD
    // one wave-step = 64 lanes x 32 weights = 2048 weights = 1024 bytes at 4 bpw; 1024 SIMDs
    const double tbs = 1024.0 * 1024.0 / ns_step * 1e9 / 1e12;
    printf("{\"lut_of_8\": %d, \"waves_per_cu\": %d, \"ms\": %.4f, \"fill_us\": %.2f, \"ns_per_wave_step_per_simd\": %.1f, \"tbs_equiv_4bpw\": %.2f}\n",
           NLUT, NW, best, fill_us, ns_step, tbs);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint32_t* d; hipMalloc(&d, (size_t) cus * 1024 * 4);
    uint64_t* clk; hipMalloc(&clk, (size_t) cus * 8); hipMemset(clk, 0, (size_t) cus * 8);
    run<0, 16>(d, clk, cus); run<0, 8>(d, clk, cus);
    run<1, 16>(d, clk, cus); run<2, 16>(d, clk, cus); run<3, 16>(d, clk, cus); run<4, 16>(d, clk, cus); run<6, 16>(d, clk, cus); run<8, 16>(d, clk, cus);
    run<2, 8>(d, clk, cus); run<4, 8>(d, clk, cus); run<8, 8>(d, clk, cus);
    return 0;
}
