// Streaming microbenchmark for the generation-3 (5..64 rows) access pattern, round 4: a workgroup = one 128-column block x one k-slice, 4 waves of 32
// columns; per decode step a wave reads 4 tile rows x its 2 tiles = four 256-byte pieces, rows (n/16 x 128 B) apart -- against generation 4's pattern
// (a wave reads ONE tile row of the whole column block: 1 KiB contiguous).  No decode work: what does the memory system give each pattern at the
// launch shapes of the batch-16 decode step (gate|up: 2 x 4096 x 14336 at 4 bpw)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_stream_g3 tools/ubench_stream_g3.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

// PAT 0: generation 3 (lane 16 g + 8 t2 + c: tile row 4 step + g, tile 2 wave + t2, 16 bytes at c * 16); PAT 1: generation 4 (a wave = one tile row of the
// column block per load, the workgroup's waves split the slice's tile rows round-robin)
template <int PAT, int NR>
__global__ __launch_bounds__(256) void stream_g3(const uint4_t* __restrict__ base, uint32_t* out, int tiles_n, int rows_per_slice, int work)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x, cb = blockIdx.y;
    const size_t row16 = (size_t) tiles_n * 8;                         // 16-byte units per tile row (128 B per tile at 4 bpw)
    const uint4_t* p; size_t step16; int steps;
    if (PAT == 0)
    {
        const int g = lane >> 4, t2 = (lane >> 3) & 1, c = lane & 7;
        p = base + ((size_t) s * rows_per_slice + g) * row16 + (size_t) (cb * 8 + 2 * wave + t2) * 8 + c;
        step16 = 4 * row16; steps = rows_per_slice / 4;
    }
    else
    {
        p = base + ((size_t) s * rows_per_slice + wave) * row16 + (size_t) cb * 64 + lane;
        step16 = 4 * row16; steps = rows_per_slice / 4;
    }
    uint4_t ring[NR];
    #pragma unroll
    for (int u = 0; u < NR; ++u) ring[u] = __builtin_nontemporal_load(p + (size_t) (u < steps ? u : steps - 1) * step16);
    uint4_t acc = { 0, 0, 0, 0 };
    for (int s0 = 0; s0 < steps; s0 += NR)
    {
        #pragma unroll
        for (int u = 0; u < NR; ++u)
        {
            uint4_t v = ring[u];
            int nx = s0 + u + NR; if (nx >= steps) nx = steps - 1;
            for (int w = 0; w < work; ++w) { v.x = v.x * 0x83DCD12Du + v.y; v.y ^= v.x >> 7; }       // optional dependent VALU work per step
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            ring[u] = __builtin_nontemporal_load(p + (size_t) nx * step16);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave] = acc.x;
}

int main()
{
    const int k = 4096, n = 28672;                                      // gate|up as ONE matrix of 224 column blocks (same strides as two of 112)
    const int tiles_n = n / 16, rows = k / 16;
    const size_t bytes = (size_t) rows * tiles_n * 128;
    uint4_t* d; uint32_t* o;
    CK(hipMalloc(&d, bytes * 4)); CK(hipMalloc(&o, 1 << 24)); CK(hipMemset(d, 1, bytes * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"bytes\": %zu, \"results\": [\n", bytes);
    for (int pat = 0; pat < 2; ++pat)
    for (int S = 2; S <= 16; S *= 2)
    for (int nr = 4; nr <= 8; nr *= 2)
    for (int work = 0; work <= 64; work += 64)
    {
        float best = 1e30f;
        for (int rep = 0; rep < 7; ++rep)
        {
            const uint4_t* b = d + (size_t) (rep & 3) * (bytes / 16);   // rotate through 4 copies (> L2 + Infinity Cache)
            dim3 grid(S, n / 128);
            CK(hipEventRecord(e0, 0));
            #define LAUNCH(P, R) hipLaunchKernelGGL((stream_g3<P, R>), grid, dim3(256), 0, 0, b, o, tiles_n, rows / S, work)
            if (pat == 0) { if (nr == 4) LAUNCH(0, 4); else LAUNCH(0, 8); } else { if (nr == 4) LAUNCH(1, 4); else LAUNCH(1, 8); }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2 && ms < best) best = ms;
        }
        printf("  {\"pattern\": \"%s\", \"S\": %d, \"workgroups\": %d, \"ring\": %d, \"valu_work\": %d, \"us\": %.1f, \"TBps\": %.2f},\n",
               pat == 0 ? "gen3 4x256B" : "gen4 1KiB", S, S * (n / 128), nr, work, best * 1e3, bytes / (best * 1e-3) / 1e12);
    }
    printf("  {}]}\n");
    return 0;
}
