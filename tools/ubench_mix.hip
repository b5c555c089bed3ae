// Microbenchmark of the EXL3 mul1 decode instruction mix (per 32 weights: 32 window extractions, 32 v_mul_lo_u32,
// 16 v_sad_u8 + 16 v_sad_hi_u8) with 0 / 8 / 16 v_mfma_f32_4x4x4_16b_f16 per 32 weights and optional ds_bpermute, no memory.
// Reports cycles per 32-weight "step" per SIMD at 8 waves/SIMD.   build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
#define STEPS 2048

typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

// FEAT bits: 1 = ds_bpermute carry, 2 = two ds_read_b128 A-fragment reads per step, 4 = one 1 KiB nontemporal global load per
// step through a 2-deep register ring (cache-resident source), 8 = sched_barrier per step
template <int NMFMA, int FEAT>
__global__ __launch_bounds__(256) void mix2(uint32_t* out, const uint4_t* __restrict__ src, uint32_t seed)
{
    __shared__ __attribute__((aligned(16))) _Float16 frag[4][64 * 16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 64 * 16; i += 64) frag[wave][i] = (_Float16) (0.01f * (i & 31));
    __syncthreads();
    float4_t acc0 = {0,0,0,0}, acc1 = {0,0,0,0};
    const uint32_t M = 0x83DCD12Du;
    uint32_t sink = 0;
    const uint4_t* p = src + (size_t) (blockIdx.x & 63) * 64 * 64 + lane;
    uint4_t ring[2];
    ring[0] = __builtin_nontemporal_load(p); ring[1] = __builtin_nontemporal_load(p + 64);
    uint32_t soft = seed + threadIdx.x;
    for (int s0 = 0; s0 < STEPS; s0 += 2)
    {
        #pragma unroll
        for (int u = 0; u < 2; ++u)
        {
            const int s = s0 + u;
            uint32_t W[5];
            if (FEAT & 4) { W[1] = ring[u].x; W[2] = ring[u].y; W[3] = ring[u].z; W[4] = ring[u].w; ring[u] = __builtin_nontemporal_load(p + (size_t) ((s + 2) & 63) * 64); }
            else { soft += 0x9E3779B9u; W[1] = soft; W[2] = soft * 3u; W[3] = soft * 5u; W[4] = soft * 7u; }
            if (FEAT & 1) W[0] = (uint32_t) __builtin_amdgcn_ds_bpermute(((lane & ~7) | ((lane - 1) & 7)) << 2, (int) W[4]); else W[0] = W[4] * 11u;
            half8_t f0, f1;
            if (FEAT & 2) { const half8_t* ap = (const half8_t*) (&frag[wave][((s & 7) * 4 + (lane & 3)) * 16]); f0 = ap[0]; f1 = ap[1]; }
            else { f0 = half8_t{1,2,3,4,5,6,7,8}; f1 = f0; }
            #pragma unroll
            for (int wd = 0; wd < 4; ++wd)
            {
                uint32_t hi = W[wd], lo = W[wd + 1];
                uint32_t s0_ = __builtin_amdgcn_alignbit(hi, lo, 28) & 0xffff, s1 = __builtin_amdgcn_alignbit(hi, lo, 24) & 0xffff;
                uint32_t s2 = __builtin_amdgcn_alignbit(hi, lo, 20) & 0xffff, s3 = lo >> 16;
                uint32_t s4 = __builtin_amdgcn_ubfe(lo, 12, 16), s5 = __builtin_amdgcn_ubfe(lo, 8, 16), s6 = __builtin_amdgcn_ubfe(lo, 4, 16), s7 = lo & 0xffff;
                uint32_t h01 = __builtin_amdgcn_sad_hi_u8(s1 * M, 0u, __builtin_amdgcn_sad_u8(s0_ * M, 0u, 0x64006400u));
                uint32_t h23 = __builtin_amdgcn_sad_hi_u8(s3 * M, 0u, __builtin_amdgcn_sad_u8(s2 * M, 0u, 0x64006400u));
                uint32_t h45 = __builtin_amdgcn_sad_hi_u8(s5 * M, 0u, __builtin_amdgcn_sad_u8(s4 * M, 0u, 0x64006400u));
                uint32_t h67 = __builtin_amdgcn_sad_hi_u8(s7 * M, 0u, __builtin_amdgcn_sad_u8(s6 * M, 0u, 0x64006400u));
                union { uint32_t u[2]; half4_t h; } b0, b1; b0.u[0] = h01; b0.u[1] = h23; b1.u[0] = h45; b1.u[1] = h67;
                half8_t f = (wd >> 1) ? f1 : f0;
                half4_t a = (wd & 1) ? half4_t{ f[4], f[5], f[6], f[7] } : half4_t{ f[0], f[1], f[2], f[3] };
                if (NMFMA >= 8)
                {
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b0.h, acc0, 4, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b1.h, acc1, 4, 0, 0);
                }
                else sink ^= h01 ^ h23 ^ h45 ^ h67;
            }
            if (FEAT & 8) __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink ^ (uint32_t) (acc0.x + acc1.y + acc0.z + acc1.w);
}

template <int NMFMA, int FEAT> static double run2(uint32_t* d, const uint4_t* src, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mix2<NMFMA, FEAT>), dim3(blocks), dim3(256), 0, 0, d, src, 1u); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) { hipEventRecord(e0, 0); hipLaunchKernelGGL((mix2<NMFMA, FEAT>), dim3(blocks), dim3(256), 0, 0, d, src, (uint32_t) r + 2);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    return best;
}

template <int NMFMA, int BPERM>
__global__ __launch_bounds__(256) void mix(uint32_t* out, uint32_t seed)
{
    uint32_t w0 = seed + threadIdx.x, w1 = w0 * 3u, w2 = w0 * 5u, w3 = w0 * 7u, p = w0 * 11u;
    float4_t acc0 = {0,0,0,0}, acc1 = {0,0,0,0};
    const uint32_t M = 0x83DCD12Du;
    half4_t a = { (_Float16) 1.0f, (_Float16) 0.5f, (_Float16) 0.25f, (_Float16) 2.0f };
    uint32_t sink = 0;
    for (int s = 0; s < STEPS; ++s)
    {
        uint32_t W[5] = { p, w0, w1, w2, w3 };
        if (BPERM) W[0] = (uint32_t) __builtin_amdgcn_ds_bpermute(((threadIdx.x & 63) ^ 1) << 2, (int) w3);
        #pragma unroll
        for (int wd = 0; wd < 4; ++wd)
        {
            uint32_t hi = W[wd], lo = W[wd + 1];
            uint32_t s0 = __builtin_amdgcn_alignbit(hi, lo, 28) & 0xffff, s1 = __builtin_amdgcn_alignbit(hi, lo, 24) & 0xffff;
            uint32_t s2 = __builtin_amdgcn_alignbit(hi, lo, 20) & 0xffff, s3 = lo >> 16;
            uint32_t s4 = __builtin_amdgcn_ubfe(lo, 12, 16), s5 = __builtin_amdgcn_ubfe(lo, 8, 16), s6 = __builtin_amdgcn_ubfe(lo, 4, 16), s7 = lo & 0xffff;
            uint32_t h01 = __builtin_amdgcn_sad_hi_u8(s1 * M, 0u, __builtin_amdgcn_sad_u8(s0 * M, 0u, 0x64006400u));
            uint32_t h23 = __builtin_amdgcn_sad_hi_u8(s3 * M, 0u, __builtin_amdgcn_sad_u8(s2 * M, 0u, 0x64006400u));
            uint32_t h45 = __builtin_amdgcn_sad_hi_u8(s5 * M, 0u, __builtin_amdgcn_sad_u8(s4 * M, 0u, 0x64006400u));
            uint32_t h67 = __builtin_amdgcn_sad_hi_u8(s7 * M, 0u, __builtin_amdgcn_sad_u8(s6 * M, 0u, 0x64006400u));
            union { uint32_t u[2]; half4_t h; } b0, b1; b0.u[0] = h01; b0.u[1] = h23; b1.u[0] = h45; b1.u[1] = h67;
            if (NMFMA >= 8)
            {
                acc0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b0.h, acc0, 4, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b1.h, acc1, 4, 0, 0);
                if (NMFMA >= 16)
                {
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b1.h, acc0, 4, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b0.h, acc1, 4, 0, 0);
                }
            }
            else sink ^= h01 ^ h23 ^ h45 ^ h67;
        }
        p = w3; w0 += 0x9E3779B9u; w1 ^= w0; w2 += w1; w3 ^= w2;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink ^ (uint32_t) (acc0.x + acc1.y + acc0.z + acc1.w);
}

template <int NMFMA, int BPERM> static double run(uint32_t* d, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mix<NMFMA, BPERM>), dim3(blocks), dim3(256), 0, 0, d, 1u); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) { hipEventRecord(e0, 0); hipLaunchKernelGGL((mix<NMFMA, BPERM>), dim3(blocks), dim3(256), 0, 0, d, (uint32_t) r + 2);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    return best;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount, blocks = cus * 8;
    uint32_t* d; hipMalloc(&d, (size_t) blocks * 256 * 4);
    double t;
    #define R(N, B) t = run<N, B>(d, blocks); printf("{\"mfma_per_step\": %d, \"bpermute\": %d, \"ms\": %.4f, \"ns_per_wave_step_per_simd\": %.1f, \"cycles_at_2.38GHz\": %.0f}\n", N, B, t, t * 1e6 / (8.0 * STEPS), t * 1e6 / (8.0 * STEPS) * 2.38);
    R(0, 0) R(8, 0) R(16, 0) R(0, 1) R(8, 1)
    uint4_t* src; hipMalloc(&src, 64 * 64 * 64 * 16); hipMemset(src, 0x5a, 64 * 64 * 64 * 16);
    #define R2(N, F) t = run2<N, F>(d, src, blocks); printf("{\"mix2_mfma\": %d, \"feat\": %d, \"ms\": %.4f, \"ns_per_wave_step_per_simd\": %.1f, \"cycles_at_2.38GHz\": %.0f}\n", N, F, t, t * 1e6 / (8.0 * STEPS), t * 1e6 / (8.0 * STEPS) * 2.38);
    R2(8, 0) R2(8, 1) R2(8, 2) R2(8, 3) R2(8, 4) R2(8, 7) R2(8, 15) R2(8, 8)
    return 0;
}
