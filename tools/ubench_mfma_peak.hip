// What does the fp16 matrix pipe of THIS chip deliver with nothing in its way?  (the round-3 figure of ~1.45-1.5 PFLOP/s came from
// tools/ubench_mfma_valu.hip, which has an s_barrier every 64 instructions and 1-2 waves per SIMD; the guide quotes 2495 TFLOP/s for 32x32x16.)
// Register-only, barrier-free: every wave issues `iters` rounds of independent matrix instructions (8 accumulators of 16x16x32 or 4 of 32x32x16) on register
// operands; 1 / 2 / 4 / 8 waves per SIMD (256-thread workgroups, wps workgroups per CU, `rounds` waves of workgroups); operands all-zero, constant or
// pseudo-random (the switching activity of the multiplier array is data-dependent, and with it power and the clock the chip holds).  The shader clock is measured inside the
// kernel: s_memtime (core clock domain) against s_memrealtime (constant 100 MHz) over the whole loop of wave 0 of every workgroup.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_peak.hip -o tools/bin/ubench_mfma_peak        run: tools/bin/ubench_mfma_peak > profiles/r04_mfma_peak_ubench.json
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MK>
__global__ __launch_bounds__(256) void peak_kernel(int iters, int data, float* out, uint64_t* clk)
{
    extern __shared__ char occupancy_pad[];                   // dynamic LDS sized by the host so that exactly wps workgroups fit on a CU
    const int tid = threadIdx.x;
    if (iters < 0) occupancy_pad[tid] = 1;
    half8 a, b;
    for (int i = 0; i < 8; ++i)
    {
        if (data == 0) { a[i] = (_Float16) 0.f; b[i] = (_Float16) 0.f; }
        else if (data == 1) { a[i] = (_Float16) 1.f; b[i] = (_Float16) 0.5f; }
        else
        {
            uint32_t h = (uint32_t) (tid * 8 + i) * 2654435761u + blockIdx.x * 40503u;
            a[i] = (_Float16) (((int) (h & 0xffff) - 32768) * (1.0f / 32768.f));
            b[i] = (_Float16) (((int) (h >> 16) - 32768) * (1.0f / 32768.f));
        }
    }
    float4v acc[8]; float16v acc32[4];
    for (int j = 0; j < 8; ++j) acc[j] = float4v{ 0.f, 0.f, 0.f, 0.f };
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc32[j][e] = 0.f;
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it)
    {
        if constexpr (MK == 0)
        {
            #pragma unroll
            for (int r = 0; r < 4; ++r)
                #pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
        }
        else
        {
            #pragma unroll
            for (int r = 0; r < 4; ++r)
                #pragma unroll
                for (int j = 0; j < 4; ++j) acc32[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc32[j], 0, 0, 0);
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][3];
    for (int j = 0; j < 4; ++j) s += acc32[j][0] + acc32[j][15];
    if (s == 123.456f) out[0] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main()
{
    CK(hipSetDevice(0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* out; uint64_t* clk;
    const int max_wg = cus * 8 * 40;
    CK(hipMalloc(&out, 4096)); CK(hipMalloc(&clk, (size_t) max_wg * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"_about\": \"tools/ubench_mfma_peak.hip: register-only, barrier-free fp16 MFMA streams; wps = 256-thread workgroups per CU = waves per SIMD (enforced with a dynamic-LDS pad of 160 KB / wps); rounds = workgroups per resident slot (4: ~8 ms launches, 40: ~80 ms); data 0 = zero operands, 1 = constants, 2 = pseudo-random in [-1, 1); clock_mhz = s_memtime ticks / s_memrealtime ticks x 100 MHz over the loop (median over workgroups); clockrate_attr_khz = %d, cus = %d\",\n \"runs\": [\n", prop.clockRate, cus);
    bool first = true;
    CK(hipFuncSetAttribute((const void*) peak_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*) peak_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int long_run = 0; long_run < 2; ++long_run)
    for (int mk = 0; mk < 2; ++mk)
        for (int data = 0; data < 3; ++data)
            for (int wps : { 1, 2, 4, 8 })
            {
                if (long_run && (data != 2 || wps < 4)) continue;        // the sustained (~80 ms) runs: random operands only
                const int rounds = long_run ? 40 : 4, grid = cus * wps * rounds;
                const size_t lds = (size_t) (160 * 1024 / wps) - (wps == 1 ? 0 : 1024);
                const int per_iter = mk == 0 ? 32 : 16;                  // matrix instructions per wave per iteration
                const double flop_per = mk == 0 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16;
                const int iters = (mk == 0 ? 4096 : 4096) / wps * 2;
                auto launch = [&] () { if (mk == 0) peak_kernel<0><<<grid, 256, lds>>>(iters, data, out, clk); else peak_kernel<1><<<grid, 256, lds>>>(iters, data, out, clk); };
                launch(); CK(hipDeviceSynchronize());
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep)
                {
                    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
                }
                std::vector<uint64_t> h((size_t) grid * 2);
                CK(hipMemcpy(h.data(), clk, (size_t) grid * 16, hipMemcpyDeviceToHost));
                std::vector<double> mhz, cyc;
                for (int i = 0; i < grid; ++i) if (h[2 * i + 1] > 0) { mhz.push_back((double) h[2 * i] / (double) h[2 * i + 1] * 100.0); cyc.push_back((double) h[2 * i]); }
                std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
                const double total = (double) grid * 4 * iters * per_iter * flop_per;
                const double tf = total / (best * 1e-3) / 1e12;
                // per-SIMD issue interval: loop ticks of a workgroup's wave 0 / (instructions of the wps waves sharing its SIMD)
                const double cyc_per_inst = cyc[cyc.size() / 2] / ((double) iters * per_iter * wps);
                printf("%s  {\"mfma\": \"%s\", \"data\": %d, \"waves_per_simd\": %d, \"rounds\": %d, \"workgroups\": %d, \"ms\": %.4f, \"tflops\": %.1f, \"clock_mhz_median\": %.0f, \"clock_mhz_min\": %.0f, "
                       "\"memtime_ticks_per_instruction_per_simd\": %.2f}",
                       first ? "" : ",\n", mk == 0 ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_32x32x16_f16", data, wps, rounds, grid, best, tf, mhz[mhz.size() / 2], mhz[0], cyc_per_inst);
                first = false;
            }
    printf("\n ]}\n");
    return 0;
}
