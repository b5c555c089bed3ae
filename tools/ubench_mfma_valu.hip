// Can a matrix-phase wave and a VALU-phase wave that share a SIMD overlap on MI355X?  (round 3: the two-group prefill attention kernel measured
// its K Q^T / P V block and its softmax block as ADDITIVE although one group of waves runs the first while the other runs the second.)
// One 512-thread workgroup per CU (2 waves per SIMD; waves w and w + 4 share SIMD order[w % 4]), NB blocks separated by s_barrier.  Per block:
//   group A (waves 0..3): NM matrix instructions (independent accumulators, register operands)        group B (waves 4..7): NV VALU instructions
//   mode 0: A only   mode 1: B only   mode 2: A and B in the same block (ping-pong)   mode 3: every wave does NM/2... (all waves: matrix, then VALU)
// VALU kinds: 0 v_fma_f32, 1 v_exp_f32, 2 mix of 1 exp + 3 fma (softmax-like).  MFMA kinds: 0 16x16x32 f16, 1 32x32x16 f16.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_valu.hip -o tools/bin/ubench_mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int MK>
__device__ __forceinline__ void mfma_block(int nm, half8 a, half8 b, float4v (&acc)[8], float16v (&acc32)[4])
{
    for (int i = 0; i < nm; i += 8)
    {
        if constexpr (MK == 0)
        {
            #pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
        }
        else
        {
            #pragma unroll
            for (int j = 0; j < 4; ++j) acc32[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc32[j], 0, 0, 0);
        }
    }
}

template <int VK>
__device__ __forceinline__ void valu_block(int nv, float (&x)[8])
{
    for (int i = 0; i < nv; i += 32)
    {
        #pragma unroll
        for (int r = 0; r < 4; ++r)
            #pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                if constexpr (VK == 0) x[j] = __builtin_fmaf(x[j], 1.0001f, 0.25f);
                else if constexpr (VK == 1) x[j] = __builtin_amdgcn_exp2f(x[j]);
                else { if (r == 0) x[j] = __builtin_amdgcn_exp2f(x[j]); else x[j] = __builtin_fmaf(x[j], 0.999f, 0.125f); }
            }
    }
}

template <int MK, int VK>
__global__ __launch_bounds__(512) void k(int mode, int nb, int nm, int nv, float* out)
{
    const int tid = threadIdx.x, wave = tid >> 6, grp = wave >> 2;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16) (tid * 0.001f + i); b[i] = (_Float16) (i * 0.5f - tid * 0.002f); }
    float4v acc[8]; float16v acc32[4];
    for (int j = 0; j < 8; ++j) acc[j] = float4v{ 0.f, 0.f, 0.f, 0.f };
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc32[j][e] = 0.f;
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = tid * 0.01f + j;
    for (int n = 0; n < nb; ++n)
    {
        __builtin_amdgcn_s_barrier();
        if (mode == 3) { mfma_block<MK>(nm / 2, a, b, acc, acc32); valu_block<VK>(nv / 2, x); }
        else if (mode == 4) { if ((n + grp) & 1) valu_block<VK>(nv, x); else mfma_block<MK>(nm, a, b, acc, acc32); }      // roles alternate per block
        else if (grp == 0) { if (mode != 1) mfma_block<MK>(nm, a, b, acc, acc32); }
        else { if (mode != 0) valu_block<VK>(nv, x); }
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + x[j];
    for (int j = 0; j < 4; ++j) s += acc32[j][0];
    if (s == 12345.678f) out[0] = s;
}

template <int MK, int VK>
static void run(int nm, int nv)
{
    float* out; hipMalloc(&out, 4);
    const int nb = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("mfma kind %d (%s) x %d per block, valu kind %d x %d per block:", MK, MK ? "32x32x16" : "16x16x32", nm, VK, nv);
    for (int mode = 0; mode < 5; ++mode)
    {
        k<MK, VK><<<256, 512>>>(mode, nb, nm, nv, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<MK, VK><<<256, 512>>>(mode, nb, nm, nv, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  mode%d %.0f ns/block", mode, ms * 1e6 / nb);
    }
    printf("\n");
    hipFree(out);
}

template <int MK>
static void peak(int waves_mode)
{
    // pure matrix stream: mode 0 = 4 waves per CU (one per SIMD), mode 3 with nv = 0 = 8 waves (two per SIMD, nm / 2 each)
    float* out; hipMalloc(&out, 4);
    const int nb = 4000, nm = 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MK, 0><<<256, 512>>>(waves_mode, nb, nm, 0, out); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MK, 0><<<256, 512>>>(waves_mode, nb, nm, 0, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (waves_mode == 0 ? 4.0 * nm : 8.0 * (nm / 2)) * nb * 256.0 * (MK ? 32768.0 : 16384.0);
    printf("pure %s, %d waves per SIMD: %.0f ns/block, %.0f TFLOP/s\n", MK ? "32x32x16" : "16x16x32", waves_mode == 0 ? 1 : 2, ms * 1e6 / nb, mf / (ms * 1e-3) * 1e-12);
    hipFree(out);
}

int main()
{
    peak<0>(0); peak<0>(3); peak<1>(0); peak<1>(3);
    run<0, 0>(64, 256); run<0, 1>(64, 64); run<0, 2>(64, 256);
    run<1, 0>(32, 256); run<1, 1>(32, 64); run<1, 2>(32, 256);
    run<0, 2>(64, 128); run<0, 2>(64, 512);
    return 0;
}
