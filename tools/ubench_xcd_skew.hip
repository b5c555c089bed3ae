// Do the 8 XCDs of an MI355X start a kernel at the same time?  (round 3: tools/gemv_timeline.py shows first-workgroup-entry stamps 1.5 - 4.7 us apart
// between XCDs; are those unsynchronised s_memrealtime counters or real dispatch skew?)
//   phase A (clock check): every workgroup spins on a flag that workgroup 0 sets, then stamps s_memrealtime: per-XCD spread of those stamps = counter
//            offset + flag propagation;
//   phase B (dispatch): NK dependent kernels in one hipGraph (the decode step's shape: grid G workgroups of 256 threads); each workgroup stamps its entry;
//            reported per XCD: first entry relative to the earliest XCD of the same launch, mean / min / max over the launches, corrected by phase A's offsets.
// build: hipcc --offload-arch=gfx950 -O3 -w tools/ubench_xcd_skew.hip -o tools/bin/ubench_xcd_skew
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

__device__ __forceinline__ uint32_t xcc_id() { uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 15; }

__global__ void k_sync(uint32_t* flag, uint64_t* stamps, uint32_t* xcc)
{
    if (threadIdx.x == 0)
    {
        if (blockIdx.x == 0)
        {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) { }                 // 20 us: every workgroup is spinning by then
            __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) { }
        stamps[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        xcc[blockIdx.x] = xcc_id();
    }
}

__global__ void k_stamp(uint64_t* stamps, uint32_t* xcc, int launch, int grid, int spin)
{
    const uint64_t t = __builtin_amdgcn_s_memrealtime();
    float v = (float) threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;                           // stand-in for ~spin * 4 cycles of work
    if (threadIdx.x == 0)
    {
        stamps[(size_t) launch * grid + blockIdx.x] = t;
        xcc[(size_t) launch * grid + blockIdx.x] = xcc_id();
        if (v == -1.f) stamps[0] = 0;
    }
}

int main()
{
    const int G = 512, NK = 64;
    uint32_t* flag; uint64_t* st; uint32_t* xc;
    hipMalloc(&flag, 4); hipMalloc(&st, sizeof(uint64_t) * G * NK); hipMalloc(&xc, 4 * G * NK);
    std::vector<uint64_t> hs(G * NK); std::vector<uint32_t> hx(G * NK);
    double off[8] = { 0 };
    for (int rep = 0; rep < 3; ++rep)
    {
        hipMemset(flag, 0, 4);
        k_sync<<<256, 64>>>(flag, st, xc);                                           // one workgroup per CU: all resident
        hipDeviceSynchronize();
        hipMemcpy(hs.data(), st, 8 * 256, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), xc, 4 * 256, hipMemcpyDeviceToHost);
        double mn[8], mx[8]; int cnt[8] = { 0 };
        for (int x = 0; x < 8; ++x) { mn[x] = 1e30; mx[x] = -1e30; }
        uint64_t base = *std::min_element(hs.begin(), hs.begin() + 256);
        for (int i = 0; i < 256; ++i) { const int x = hx[i] & 7; const double t = (hs[i] - base) * 0.01; mn[x] = std::min(mn[x], t); mx[x] = std::max(mx[x], t); cnt[x]++; }
        printf("clock check rep %d: per-XCD [min, max] us after the flag:", rep);
        for (int x = 0; x < 8; ++x) { printf("  x%d[%.2f %.2f]n%d", x, mn[x], mx[x], cnt[x]); if (rep == 2) off[x] = mn[x]; }
        printf("\n");
    }
    for (int spin : { 0, 500, 2000 })
    {
        hipStream_t s; hipStreamCreate(&s);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int l = 0; l < NK; ++l) k_stamp<<<G, 256, 0, s>>>(st, xc, l, G, spin);
        hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        hipMemcpy(hs.data(), st, 8 * G * NK, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), xc, 4 * G * NK, hipMemcpyDeviceToHost);
        double sum[8] = { 0 }, lo[8], hi[8]; double per_launch = 0;
        for (int x = 0; x < 8; ++x) { lo[x] = 1e30; hi[x] = -1e30; }
        for (int l = 8; l < NK; ++l)
        {
            double first[8]; for (int x = 0; x < 8; ++x) first[x] = 1e30;
            for (int i = 0; i < G; ++i) { const int x = hx[l * G + i] & 7; first[x] = std::min(first[x], (double) hs[l * G + i] * 0.01 - off[x]); }
            const double e = *std::min_element(first, first + 8);
            for (int x = 0; x < 8; ++x) { const double d = first[x] - e; sum[x] += d; lo[x] = std::min(lo[x], d); hi[x] = std::max(hi[x], d); }
        }
        per_launch = ((double) *std::min_element(hs.begin() + (NK - 1) * G, hs.begin() + NK * G) - (double) *std::min_element(hs.begin() + 8 * G, hs.begin() + 9 * G)) * 0.01 / (NK - 9);
        printf("graph of %d dependent kernels (%d x 256 threads, spin %d): %.2f us per kernel; first entry per XCD after the earliest XCD, mean [min max] us:", NK, G, spin, per_launch);
        for (int x = 0; x < 8; ++x) printf("  x%d %.2f[%.2f %.2f]", x, sum[x] / (NK - 8), lo[x], hi[x]);
        printf("\n");
    }
    return 0;
}
