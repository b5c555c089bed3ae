// Can a dependent chain of short kernels hide its launch gaps and prologues by running neighbours on TWO queues, with the dependency carried by a
// completion counter in memory instead of the queue's barrier?  (round 3: a decode step is 6 dependent launches per layer, ~1.6 us of gap between two
// kernels of a graph plus 1-2 us until a kernel's first weight rows arrive: ~27 us of 49 us per layer are per-launch constants.)
//   every kernel: G workgroups x 256 threads; (1) request its "weights" (WB bytes per workgroup, independent of the predecessor), (2) wait for the
//   predecessor, (3) read the predecessor's output vector, (4) consume the weights, write its own slice of the output vector, (5) signal.
//   mode 0: one stream, the queue's barrier between kernels (steps 2 / 5 skipped)          -- what the decode step does today
//   mode 1: kernels alternate between two streams of one graph (no edge between neighbours): kernel k+1 becomes resident while kernel k runs, has its
//           weights in flight and waits in step 2 for k's done flag (every workgroup of k adds 1 to a counter after a release fence, the last one raises 64 flag copies on 64 lines; bounded wait)
//   mode 2: as mode 1 but without a graph (two plain streams)
// checks: the chained vector is exact (every element == number of kernels), no wait timed out.
// build: hipcc --offload-arch=gfx950 -O3 -w tools/ubench_chain_overlap.hip -o tools/bin/ubench_chain_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <stdlib.h>

typedef uint32_t u4v __attribute__((ext_vector_type(4)));
#define NV 4096
#define TS_STRIDE 128
#define CNT_STRIDE 32          // uint32 words between two copies of a counter (128 bytes)
#define CNT_COPIES 64

__global__ __launch_bounds__(256) void k_stage(const uint4* __restrict__ w, size_t w_per_wg16, int nload, const float* xin, float* xout,
                                               uint32_t* cnt_wait, uint32_t wait_target, uint32_t* cnt_signal, uint32_t* err, int per_wg, unsigned long long* ts)
{
    if (ts && threadIdx.x == 0 && blockIdx.x == 0) ts[0] = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, b = blockIdx.x;
    // (1) weights: nload x 16 bytes per thread, requested before anything that depends on the predecessor
    u4v acc[8];
    const u4v* wp = (const u4v*) w + (size_t) b * w_per_wg16 + tid;
    #pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = i < nload ? __builtin_nontemporal_load(wp + (size_t) i * 256) : u4v{ 0, 0, 0, 0 };
    // (2) wait for the predecessor (wave 0 polls ONE of its 64 done flags, the rest of the workgroup sits in the barrier)
    if (cnt_wait)
    {
        if (tid < 64)
        {
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            const uint32_t* fl = cnt_wait + (1 + (b & (CNT_COPIES - 1))) * CNT_STRIDE;
            bool ok = false;
            while (!ok)
            {
                ok = __builtin_amdgcn_readfirstlane(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0;
                if (!ok)
                {
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 100000) { if (tid == 0) atomicAdd(err, 1u); break; }      // 1 ms
                    __builtin_amdgcn_s_sleep(8);
                }
            }
        }
        __syncthreads();
#ifdef USE_FENCES
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        if (ts && threadIdx.x == 0 && blockIdx.x == 0) ts[TS_STRIDE] = __builtin_amdgcn_s_memrealtime();
    }
    // (3) + (4): the predecessor's vector, this workgroup's slice of the next one
    uint32_t ws = 0;
    for (int r = 8; r < nload; r += 8)              // more weights than registers: stream the rest
    {
        #pragma unroll
        for (int i = 0; i < 8; ++i) { ws += acc[i].x ^ acc[i].y ^ acc[i].z ^ acc[i].w; acc[i] = r + i < nload ? __builtin_nontemporal_load(wp + (size_t) (r + i) * 256) : u4v{ 0, 0, 0, 0 }; }
    }
    #pragma unroll
    for (int i = 0; i < 8; ++i) ws += acc[i].x ^ acc[i].y ^ acc[i].z ^ acc[i].w;
    if (tid < per_wg)
    {
        const int i = b * per_wg + tid;
#ifdef USE_FENCES
        if (i < NV) xout[i] = xin[(i * 7 + 3) & (NV - 1)] + 1.0f + (ws == 0x9e3779b9u ? 1.0f : 0.0f);
#else
        // no cache-wide fences: the chained data itself travels with agent-scope (L2-bypassing / write-through) accesses
        if (i < NV) __hip_atomic_store(xout + i, __hip_atomic_load(xin + ((i * 7 + 3) & (NV - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1.0f
                                       + (ws == 0x9e3779b9u ? 1.0f : 0.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    // (5) signal: one counter; the last workgroup to arrive raises the done flags
    if (cnt_signal)
    {
#ifdef USE_FENCES
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        if (tid < 64)
        {
            uint32_t old = 0;
            if (tid == 0) old = __hip_atomic_fetch_add(cnt_signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == gridDim.x - 1 && tid < CNT_COPIES)
                __hip_atomic_store(cnt_signal + (1 + tid) * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == gridDim.x - 1 && tid == 0 && ts) ts[2 * TS_STRIDE] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

int main()
{
    const int NK = 96;
    const size_t WBUF = (size_t) 1 << 30;                   // 1 GiB of "weights", walked through so that every kernel reads cold lines
    uint4* w; float* x; uint32_t* cnt; uint32_t* err; unsigned long long* ts;
    hipMalloc(&ts, TS_STRIDE * 3 * 8);
    hipMalloc(&w, WBUF); hipMemset(w, 1, WBUF);
    hipMalloc(&x, 2 * NV * sizeof(float)); hipMalloc(&cnt, (size_t) NK * (CNT_COPIES + 1) * CNT_STRIDE * 4); hipMalloc(&err, 4);
    hipStream_t s[2]; hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
    hipEvent_t fork, join, e0, e1; hipEventCreateWithFlags(&fork, hipEventDisableTiming); hipEventCreateWithFlags(&join, hipEventDisableTiming);
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct Shape { int G; int kb_per_wg; };
    const Shape shapes[] = { { 512, 16 }, { 1792, 32 }, { 768, 16 } };
    for (const Shape& sh : shapes)
    {
        const int G = sh.G, per_wg = NV / G > 0 ? (NV + G - 1) / G : 1;
        const int nload = sh.kb_per_wg * 1024 / (256 * 16);
        const size_t w_per_wg16 = (size_t) nload * 256;
        const size_t w_per_k16 = w_per_wg16 * G;
        printf("G %d, %d KiB per workgroup (%.1f MiB per kernel), %d chained kernels:", G, sh.kb_per_wg, w_per_k16 * 16.0 / 1048576.0, NK);
        for (int mode = 0; mode < 3; ++mode)
        {
            auto enqueue = [&]()
            {
                hipMemsetAsync(x, 0, 2 * NV * sizeof(float), s[0]);
                hipMemsetAsync(ts, 0xff, TS_STRIDE * 2 * 8, s[0]); hipMemsetAsync(ts + 2 * TS_STRIDE, 0, TS_STRIDE * 8, s[0]);
                hipMemsetAsync(cnt, 0, (size_t) NK * (CNT_COPIES + 1) * CNT_STRIDE * 4, s[0]);
                if (mode) { hipEventRecord(fork, s[0]); hipStreamWaitEvent(s[1], fork, 0); }
                for (int k = 0; k < NK; ++k)
                {
                    hipStream_t st = mode ? s[k & 1] : s[0];
                    const uint4* wk = w + ((size_t) k * w_per_k16) % (WBUF / 16 - w_per_k16);
                    uint32_t* cw = (mode && k > 0) ? cnt + (size_t) (k - 1) * (CNT_COPIES + 1) * CNT_STRIDE : nullptr;
                    uint32_t* cs = mode ? cnt + (size_t) k * (CNT_COPIES + 1) * CNT_STRIDE : nullptr;
                    k_stage<<<G, 256, 0, st>>>(wk, w_per_wg16, nload, x + (k & 1) * NV, x + ((k + 1) & 1) * NV, cw, (uint32_t) G, cs, err, per_wg, ts + k);
                }
                if (mode) { hipEventRecord(join, s[1]); hipStreamWaitEvent(s[0], join, 0); }
            };
            hipMemset(err, 0, 4);
            float ms = 0.f;
            if (mode < 2)
            {
                hipGraph_t g; hipGraphExec_t ge;
                hipStreamBeginCapture(s[0], hipStreamCaptureModeGlobal);
                enqueue();
                hipStreamEndCapture(s[0], &g);
                if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf(" instantiate failed"); continue; }
                for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, s[0]);
                hipStreamSynchronize(s[0]);
                hipEventRecord(e0, s[0]);
                for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, s[0]);
                hipEventRecord(e1, s[0]); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1); ms /= 10;
                hipGraphExecDestroy(ge); hipGraphDestroy(g);
            }
            else
            {
                for (int i = 0; i < 2; ++i) enqueue();
                hipStreamSynchronize(s[0]);
                hipEventRecord(e0, s[0]);
                for (int i = 0; i < 5; ++i) enqueue();
                hipEventRecord(e1, s[0]); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            }
            std::vector<float> hx(NV); uint32_t herr = 0;
            hipMemcpy(hx.data(), x + (NK & 1) * NV, NV * sizeof(float), hipMemcpyDeviceToHost); hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
            int bad = 0; for (int i = 0; i < per_wg * G && i < NV; ++i) bad += hx[i] != (float) NK;
            printf("  mode%d %.2f us/kernel (bad %d, timeouts %u)", mode, ms * 1000.0 / NK, bad, herr);
            if (getenv("CHAIN_TIMELINE"))
            {
                std::vector<unsigned long long> ht(TS_STRIDE * 3);
                hipMemcpy(ht.data(), ts, TS_STRIDE * 3 * 8, hipMemcpyDeviceToHost);
                printf("\n    kernel: start / dependency seen / last workgroup done, us after kernel 40's start:");
                for (int k = 40; k < 48; ++k)
                    printf("  k%d %.2f/%.2f/%.2f", k, (double) (long long) (ht[k] - ht[40]) * 0.01, mode ? (double) (long long) (ht[TS_STRIDE + k] - ht[40]) * 0.01 : 0.0,
                           (double) (long long) (ht[2 * TS_STRIDE + k] - ht[40]) * 0.01);
                printf("\n");
            }
        }
        printf("\n");
    }
    return 0;
}
