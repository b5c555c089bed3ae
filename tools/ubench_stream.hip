// Streaming microbenchmark: what does HBM give for the EXL3 GEMV access pattern?
// Every wave reads `steps` chunks of 1 KiB (64 lanes x 16 B, nontemporal) that are `stride` bytes apart, PF loads in
// flight, and folds them into one xor (no other work).  stride = 1 KiB reproduces a band-contiguous re-layout, stride =
// n/16*128 B the checkpoint layout [k/16][n/16][16K] where consecutive k tile-rows of a column block are ~1 MB apart.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o tools/bin/ubench_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));

template <int PF>
__global__ __launch_bounds__(1024) void stream_kernel(const uint4_t* __restrict__ base, uint32_t* out, size_t cb_stride16, size_t row_stride16, int steps)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint4_t* p = base + (size_t) blockIdx.x * cb_stride16 + (size_t) (w * steps) * row_stride16 + lane;
    uint4_t ring[PF];
    #pragma unroll
    for (int u = 0; u < PF; ++u) ring[u] = __builtin_nontemporal_load(p + (size_t) (u < steps ? u : steps - 1) * row_stride16);
    uint4_t acc = { 0, 0, 0, 0 };
    for (int s0 = 0; s0 < steps; s0 += PF)
    {
        #pragma unroll
        for (int u = 0; u < PF; ++u)
        {
            uint4_t v = ring[u];
            int nx = s0 + u + PF; if (nx >= steps) nx = steps - 1;
            ring[u] = __builtin_nontemporal_load(p + (size_t) nx * row_stride16);
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x * 16 + w] = acc.x;
}

int main()
{
    // lm_head geometry: 1002 column blocks x 256 tile rows x 1 KiB = 262.7 MB ; gate/up: 224 x 256
    uint4_t* d; uint32_t* o;
    const size_t maxbytes = (size_t) 1002 * 256 * 1024;
    hipMalloc(&d, maxbytes * 2); hipMalloc(&o, 1 << 22);
    hipMemset(d, 1, maxbytes * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("{\"results\": [\n");
    int cbs_list[2] = { 1002, 224 };
    for (int ci = 0; ci < 2; ++ci)
    for (int contiguous = 0; contiguous < 2; ++contiguous)
    for (int W = 4; W <= 16; W *= 2)
    for (int pf = 2; pf <= 8; pf *= 2)
    {
        const int cbs = cbs_list[ci], rows = 256, steps = rows / W;
        const size_t bytes = (size_t) cbs * rows * 1024;
        size_t cb_stride16 = contiguous ? (size_t) rows * 64 : 64;
        size_t row_stride16 = contiguous ? 64 : (size_t) cbs * 64;
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep)
        {
            const uint4_t* b = d + (rep & 1) * (maxbytes / 16);
            hipEventRecord(e0, 0);
            if (pf == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(cbs), dim3(64 * W), 0, 0, b, o, cb_stride16, row_stride16, steps);
            if (pf == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(cbs), dim3(64 * W), 0, 0, b, o, cb_stride16, row_stride16, steps);
            if (pf == 8) hipLaunchKernelGGL(stream_kernel<8>, dim3(cbs), dim3(64 * W), 0, 0, b, o, cb_stride16, row_stride16, steps);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2 && ms < best) best = ms;
        }
        printf("  {\"colblocks\": %d, \"layout\": \"%s\", \"waves_per_wg\": %d, \"pf\": %d, \"us\": %.1f, \"TBps\": %.2f},\n",
               cbs, contiguous ? "band-contiguous" : "checkpoint", W, pf, best * 1e3, bytes / (best * 1e-3) / 1e12);
    }
    printf("  {}]}\n");
    return 0;
}
