// Feasibility microbenchmark for a persistent "chain" decode kernel: what does ONE cross-workgroup stage boundary cost inside a launch
// (every workgroup publishes a little data, signals, waits for all the others, reads another workgroup's data) compared with the ~1.6 us
// node-to-node gap + ramp of a hipGraph launch boundary?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_chain_barrier tools/ubench_chain_barrier.hip && /tmp/ubench_chain_barrier
// Prints one JSON line per configuration: {"wgs", "threads", "shards", "stages", "us_per_stage", "errors"}.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7; }

// counters: [stage][shards] ints, 64 bytes apart.  data: [stage & 1][wgs][64] floats
template <int SHARDS, bool WORK, bool FENCE, bool SLEEP>
__global__ __launch_bounds__(1024) void chain_kernel(int* counters, float* data, int nstages, int* errors, const float* dummy_w, int w_stride)
{
    const int P = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
    const int shard = SHARDS == 1 ? 0 : (xcc_id() % SHARDS);
    __shared__ int s_ok;
    float acc = 0.0f;
    int err = 0;
    for (int s = 0; s < nstages; ++s)
    {
        float* mine = data + ((size_t) (s & 1) * P + w) * 64;
        if (WORK)
        {
            // a little streaming work per stage (64 KiB per workgroup), like a weight strip
            const float4* src = (const float4*) (dummy_w + (size_t) ((s * P + w) % w_stride) * 16384);
            for (int i = tid; i < 4096; i += blockDim.x) { float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
        }
        if (tid < 64) __hip_atomic_store(mine + tid, (float) (s * 1000 + w) + (WORK ? acc * 0.0f : 0.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!FENCE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                 // all stores of the workgroup issued ...
        if (tid == 0)
        {
            if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // ... and visible before the signal
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // write-through stores: acknowledged = at the coherence point
            __hip_atomic_fetch_add(counters + ((size_t) s * SHARDS + shard) * 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // wait: wave 0 polls the SHARDS counters of this stage (lane i -> shard i), bounded
        if (tid < 64)
        {
            int spins = 0, sum;
            do
            {
                int v = tid < SHARDS ? __hip_atomic_load(counters + ((size_t) s * SHARDS + tid) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                sum = v;
                #pragma unroll
                for (int i = 1; i < 8; i <<= 1) sum += __shfl_xor(sum, i);
                sum = __shfl(sum, 0);
                if (SLEEP && sum < P) __builtin_amdgcn_s_sleep(8);
            } while (sum < P && ++spins < 2000000);
            if (tid == 0) s_ok = sum >= P;
        }
        __syncthreads();
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (!s_ok) { if (tid == 0) atomicAdd(errors, 1000000); return; }
        // read another workgroup's data of this stage
        const int other = (w * 7 + 13 + s) % P;
        if (tid < 64)
        {
            float v = __hip_atomic_load(data + ((size_t) (s & 1) * P + other) * 64 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != (float) (s * 1000 + other)) ++err;
        }
    }
    if (err) atomicAdd(errors, err);
    if (WORK && acc == 123.456f) errors[1] = 1;
}

template <int SHARDS, bool WORK, bool FENCE, bool SLEEP>
static void run(int wgs, int threads, int nstages, int* counters, float* data, int* errors, const float* dummy, int w_stride)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    int errs = 0;
    for (int rep = 0; rep < 6; ++rep)
    {
        CK(hipMemsetAsync(counters, 0, (size_t) nstages * SHARDS * 64, 0));
        CK(hipMemsetAsync(errors, 0, 8, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((chain_kernel<SHARDS, WORK, FENCE, SLEEP>), dim3(wgs), dim3(threads), 0, 0, counters, data, nstages, errors, dummy, w_stride);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
        int h[2]; CK(hipMemcpy(h, errors, 8, hipMemcpyDeviceToHost));
        errs += h[0];
    }
    printf("{\"wgs\": %d, \"threads\": %d, \"shards\": %d, \"work\": %d, \"fence\": %d, \"sleep\": %d, \"stages\": %d, \"us_per_stage\": %.3f, \"errors\": %d}\n",
           wgs, threads, SHARDS, (int) WORK, (int) FENCE, (int) SLEEP, nstages, best * 1000.0f / nstages, errs);
    fflush(stdout);
}

int main()
{
    const int nstages = 256;
    int* counters; float* data; int* errors; float* dummy;
    CK(hipMalloc(&counters, (size_t) nstages * 8 * 64));
    CK(hipMalloc(&data, (size_t) 2 * 2048 * 64 * 4));
    CK(hipMalloc(&errors, 8));
    const int w_stride = 8192;                                           // 512 MiB of dummy weights: beyond the caches
    CK(hipMalloc(&dummy, (size_t) w_stride * 65536));
    CK(hipMemset(dummy, 0, (size_t) w_stride * 65536));
    for (int wgs : { 64, 256, 512 })
        for (int threads : { 256, 1024 })
        {
            if (wgs * threads > 256 * 2048) continue;                    // must be co-resident
            run<8, false, true, false>(wgs, threads, nstages, counters, data, errors, dummy, w_stride);
            run<1, false, false, false>(wgs, threads, nstages, counters, data, errors, dummy, w_stride);
            run<8, false, false, false>(wgs, threads, nstages, counters, data, errors, dummy, w_stride);
            run<8, false, false, true>(wgs, threads, nstages, counters, data, errors, dummy, w_stride);
            run<8, true, false, false>(wgs, threads, nstages, counters, data, errors, dummy, w_stride);
        }
    return 0;
}
