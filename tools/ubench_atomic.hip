// What does finishing a split-k GEMV with atomics cost on MI355X?  (round 3: could the o_proj / down_proj epilogue add its partial rows straight into
// a fixed-point residual accumulator and make the glue_resid launches disappear?)
// Grid = (S slices, column blocks) like the decode GEMVs; every workgroup waits ~DELAY us (stand-in for the streaming phase), then ONE half-wave adds
// 128 values (4 per lane) of its column block into acc[cb*128 ..]:
//   mode 0: plain 16-byte slab store (today's deferred epilogue)          mode 1: 64-bit integer atomic add, agent scope, no return (deterministic)
//   mode 2: fp32 atomic add, agent scope, no return                       mode 3: 64-bit integer atomic add, all 4 waves' partials separately (4x the ops)
// Reported: us per launch inside a hipGraph of 100 back-to-back launches, each followed by a tiny dependent reader kernel (the consumer).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_atomic.hip -o tools/bin/ubench_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(256) void k_epi(float* slabs, unsigned long long* acc64, float* acc32, int S, int spin)
{
    const int s = blockIdx.x, cb = blockIdx.y, tid = threadIdx.x, l = tid & 31;
    // stand-in for the streaming loop
    float v = (float) tid;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    const bool writer = MODE == 3 ? true : (tid < 32);
    if (!writer) { if (v == -1.f) slabs[0] = v; return; }
    float4 p = { v, v + 1.f, v + 2.f, v + 3.f };
    if (MODE == 0) ((float4*) (slabs + ((size_t) cb * S + s) * 128))[l] = p;
    else if (MODE == 2)
    {
        float* a = acc32 + cb * 128 + 4 * l;
        __hip_atomic_fetch_add(a + 0, p.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add(a + 1, p.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a + 2, p.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add(a + 3, p.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else
    {
        unsigned long long* a = acc64 + cb * 128 + 4 * l;
        const float q[4] = { p.x, p.y, p.z, p.w };
        #pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const long long f = (long long) ((double) q[i] * 4294967296.0);
            __hip_atomic_fetch_add(a + i, (unsigned long long) f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the consumer: 64 workgroups read the whole accumulator row (or the slabs) -- the dependent kernel that follows
template <int MODE>
__global__ void k_consume(const float* slabs, const unsigned long long* acc64, const float* acc32, int n, int S, float* out)
{
    const int hw = threadIdx.x >> 5, l = threadIdx.x & 31, cb = blockIdx.x * 8 + hw;
    if (cb * 128 >= n) return;
    float4 v = { 0.f, 0.f, 0.f, 0.f };
    if (MODE == 0)
    {
        float4 t[16];
        #pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = ((const float4*) (slabs + ((size_t) cb * S + (k < S ? k : S - 1)) * 128))[l];
        #pragma unroll
        for (int k = 0; k < 16; ++k) if (k < S) { v.x += t[k].x; v.y += t[k].y; v.z += t[k].z; v.w += t[k].w; }
    }
    else
    {
        const unsigned long long* a = acc64 + cb * 128 + 4 * l;
        v.x = (float) (long long) a[0]; v.y = (float) (long long) a[1]; v.z = (float) (long long) a[2]; v.w = (float) (long long) a[3];
    }
    ((float4*) (out + cb * 128))[l] = v;
}

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
    float* slabs; unsigned long long* acc64; float* acc32; float* out;
    hipMalloc(&slabs, 64 << 20); hipMalloc(&acc64, 1 << 20); hipMalloc(&acc32, 1 << 20); hipMalloc(&out, 1 << 20);
    hipMemset(slabs, 0, 64 << 20); hipMemset(acc64, 0, 1 << 20); hipMemset(acc32, 0, 1 << 20);
    printf("{\"results\": [\n");
    const int shapes[3][2] = { { 16, 32 }, { 8, 224 }, { 16, 48 } };           // (S, column blocks): o / down, gate|up, q|k|v
    const int spins[2] = { 0, 400 };
    bool first = true;
    for (int si = 0; si < 3; ++si) for (int sp = 0; sp < 2; ++sp)
    {
        const int S = shapes[si][0], CB = shapes[si][1], spin = spins[sp];
        dim3 grid(S, CB);
        const int n = CB * 128;
        // producer alone per mode (kernel completion includes the completion of its stores / atomics), then producer + a light consumer
        // (one half-wave per column block, the shape of glue_resid) for the slab and the 64-bit atomic forms
        float p[4];
        p[0] = time_graph(st, 100, [&] { k_epi<0><<<grid, 256, 0, st>>>(slabs, acc64, acc32, S, spin); });
        p[1] = time_graph(st, 100, [&] { k_epi<1><<<grid, 256, 0, st>>>(slabs, acc64, acc32, S, spin); });
        p[2] = time_graph(st, 100, [&] { k_epi<2><<<grid, 256, 0, st>>>(slabs, acc64, acc32, S, spin); });
        p[3] = time_graph(st, 100, [&] { k_epi<3><<<grid, 256, 0, st>>>(slabs, acc64, acc32, S, spin); });
        float c0 = time_graph(st, 100, [&] { k_epi<0><<<grid, 256, 0, st>>>(slabs, acc64, acc32, S, spin); k_consume<0><<<(CB + 7) / 8, 256, 0, st>>>(slabs, acc64, acc32, n, S, out); });
        float c1 = time_graph(st, 100, [&] { k_epi<1><<<grid, 256, 0, st>>>(slabs, acc64, acc32, S, spin); k_consume<1><<<(CB + 7) / 8, 256, 0, st>>>(slabs, acc64, acc32, n, S, out); });
        printf("%s {\"S\": %d, \"column_blocks\": %d, \"spin\": %d, \"producer_us\": {\"slab_store\": %.2f, \"atomic_i64\": %.2f, \"atomic_f32\": %.2f, \"atomic_i64_x4_waves\": %.2f}, "
               "\"producer_plus_consumer_us\": {\"slab_store_then_reduce\": %.2f, \"atomic_i64_then_read\": %.2f}}",
               first ? "" : ",\n", S, CB, spin, p[0], p[1], p[2], p[3], c0, c1);
        first = false;
    }
    printf("\n]}\n");
    return 0;
}
