// Launch-chain microbenchmark: what does one dependent kernel launch cost inside a hipGraph on MI355X, as a function of grid
// size, workgroup size, dynamic LDS and kernarg size?  (The decode step is a chain of ~260 dependent launches.)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_launch.hip -o tools/bin/ubench_launch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

struct Big { uint64_t v[60]; };

__global__ void k_empty(float* p) { if (p && threadIdx.x == 4096) p[0] = 1.0f; }
__global__ void k_big(Big b, float* p) { if (p && b.v[3] == 77 && threadIdx.x == 4096) p[0] = 1.0f; }
__global__ void k_lds(float* p) { extern __shared__ float sm[]; if (p && threadIdx.x == 4096) p[0] = sm[5]; }
// scratch: a dynamically indexed private array forces a scratch (private segment) allocation for every wave
__global__ void k_scratch(float* p, int idx)
{
    volatile float priv[8];
    for (int i = 0; i < 8; ++i) priv[i] = (float) (i + threadIdx.x);
    if (p && priv[idx & 7] == -5.0f) p[0] = 1.0f;
}
// touch: every workgroup reads 16 B per thread from a buffer and (never) writes: adds one global-load latency
__global__ void k_touch(const float4* src, float* p) { float4 v = src[blockIdx.x * blockDim.x + threadIdx.x]; if (v.x == 123.456f) p[0] = v.y; }

template <typename F>
static float time_graph(hipStream_t st, int n, F launch)
{
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; ++i) launch();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r)
    {
        hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best * 1e3f / n;
}

int main()
{
    hipStream_t st; hipStreamCreate(&st);
    float* d; hipMalloc(&d, 1 << 26);
    hipMemset(d, 0, 1 << 26);
    const int n = 200;
    Big b; for (int i = 0; i < 60; ++i) b.v[i] = i;
    printf("{\"results\": [\n");
    int grids[3] = { 1, 512, 4096 };
    int threads[2] = { 256, 1024 };
    for (int gi = 0; gi < 3; ++gi) for (int ti = 0; ti < 2; ++ti)
    {
        const int G = grids[gi], T = threads[ti];
        float a = time_graph(st, n, [&] { k_empty<<<G, T, 0, st>>>(d); });
        float c = time_graph(st, n, [&] { k_big<<<G, T, 0, st>>>(b, d); });
        float l = time_graph(st, n, [&] { k_lds<<<G, T, 65536, st>>>(d); });
        float sc = time_graph(st, n, [&] { k_scratch<<<G, T, 0, st>>>(d, gi); });
        float t = (size_t) G * T * 16 <= (1u << 26) ? time_graph(st, n, [&] { k_touch<<<G, T, 0, st>>>((const float4*) d, d); }) : -1.f;
        printf("  {\"grid\": %d, \"threads\": %d, \"empty_us\": %.2f, \"kernarg480B_us\": %.2f, \"lds64k_us\": %.2f, \"one_load_us\": %.2f, \"scratch_us\": %.2f},\n", G, T, a, c, l, t, sc);
    }
    printf("  {}]}\n");
    return 0;
}
