// EXPERIMENT (round 5), not part of the product library: the one-wave-per-SIMD form of the prefill NT GEMM (VERDICT r4 task 2: "one wave per SIMD owning
// the register file if hipcc allows").  It does not: compiled for gfx950 with hipcc 7.2 (-O3, waves_per_eu(1, 1)) the kernel below gets
//     VGPRs 256, AGPRs 256, scratch 732 B / lane, 261 spilled VGPRs; 128 v_mfma in the main loop against 1302 v_accvgpr_read / _write and 190 scratch
//     accesses (scratch reloads + s_waitcnt vmcnt(0) between the global_load_lds of one K-tile)
// -- the allocator treats the unified file as one pool, parks FRAGMENTS in the accumulation half and shuffles accumulators through the vector half around
// every matrix instruction.  Pinning the accumulators with inline asm ("+a" / "v" constraints, the variant kept below) made it worse (1737 moves, 736 B).
// The structure needs hand-allocated registers, i.e. an assembly main loop; it was not run on the GPU (a kernel with this much scratch also breaks the
// co-residency the rest of the round relied on).  What the structure buys on paper is in the comment of the kernel; the shipped 8-wave kernel
// (exl3_gemm_nt.hip) and the library route are unchanged.  Build check only:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I exllamav3_amd/csrc -c tools/experiments/gemm_nt2_one_wave_per_simd.hip -Rpass-analysis=kernel-resource-usage
#include "exl3_common.cuh"
#include "exl3_api_internal.h"

#define GNT_BM 256
#define GNT_BN 256
#define GNT_EPI_STORE 0
#define GNT_EPI_ACC 1
#define GNT_EPI_SILU_MUL 2
#define GNT_CPITCH 264
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct GntArgs { const half_t* A; const half_t* Bt; half_t* C; int64_t lda, ldb, ldc; int M, N, K, epi; int tiles_m, tiles_n; };
__device__ __forceinline__ void gnt_glds16(const void* gptr, void* lds_uniform_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) gptr, (__attribute__((address_space(3))) void*) lds_uniform_base, 16, 0, 0);
}
// ------------------------------------------------------------------------------------------------------------------------------------------------
// Version 2 (round 5): ONE wave per SIMD, 128 x 128 per wave, accumulators in the upper half of the unified register file.
//
// Why: the 8-wave kernel above is LDS-bound by construction.  Per 64-deep K-tile a CU's eight 128 x 64 waves read 8 x 24 fragments x 1 KiB = 192 KiB
// from LDS and the staging writes 64 KiB: 2048 cycles of the 128 B / clock LDS port -- exactly the 2048 cycles its 512 matrix instructions take, so
// any wait exposes the matrix pipe (SQ_WAIT_ANY 37 % against the library's 16 %, profiles/r03_gemm_nt_pmc.json).  Fragment traffic per flop depends
// only on the wave tile, (TM + TN) / (TM TN): four 128 x 128 waves read 128 KiB per K-tile instead of 192 (1536 port cycles against the same 2048
// matrix cycles), and the single wave of a SIMD overlaps its own fragment reads with its own matrix instructions (reads of K-tile t + 1 are issued
// between the 64 MFMAs of K-tile t; 2 x 16 fragment registers).
//   * 256 x 256 x 32 K-tiles, FOUR LDS buffers of 32 KiB (A 16 | B 16): a tile is requested three tiles before its fragments are read (>= 3000 cycles
//     of flight for the global_load_lds), waited for with a COUNTED vmcnt (16 younger requests stay in flight), one workgroup barrier per K-tile.
//   * LDS image: row pitch 64 B (32 halves); chunk c (16 B) of row r sits at position c ^ ((r >> 2) & 3): the 16 rows x one chunk of a fragment read
//     cover all 64 banks exactly once (tests/test_lds_swizzle.py checks the map), and the permutation is applied to the SOURCE address of the
//     lane-linear global_load_lds image (lane L of an instruction = row L >> 2, position L & 3 of a 16-row group).
//   * 64 accumulator tiles of v_mfma_f32_16x16x32_f16 per wave (256 registers); epilogue through LDS as above.
#define GN2_BK 32
#define GN2_OP_BYTES (256 * GN2_BK * 2)               // one operand, one K-tile: 16 KiB
#define GN2_BUF_BYTES (2 * GN2_OP_BYTES)              // A | B
#define GN2_NBUF 4
#define GN2_LDS_BYTES (256 * GNT_CPITCH * 2)          // the epilogue tile (135168 B) >= 4 buffers (131072 B)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void exl3_gemm_nt2_kernel(const GntArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int nwg = a.tiles_m * a.tiles_n;
    int t;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int tm, tn;
    if ((a.tiles_m & 7) == 0 && (a.tiles_n & 3) == 0)
    {
        const int g = t >> 5, j = t & 31, gm = a.tiles_m >> 3;
        tm = (g % gm) * 8 + (j & 7); tn = (g / gm) * 4 + (j >> 3);
    }
    else { tm = t % a.tiles_m; tn = t / a.tiles_m; }
    const int m0 = tm * GNT_BM, n0 = tn * GNT_BN;

    // ---- staging: 4 + 4 global_load_lds per thread per K-tile.  Instruction i of wave w covers rows (i * 4 + w) * 16 .. + 15 of the operand tile;
    // lane L supplies LDS position (L & 3) of row (L >> 2) of that group and fetches global chunk (L & 3) ^ ((row >> 2) & 3)
    const half_t* srcA[4]; const half_t* srcB[4];
    #pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const int row = (i * 4 + wave) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        srcA[i] = a.A + (size_t) min(m0 + row, a.M - 1) * a.lda + chunk * 8;
        srcB[i] = a.Bt + (size_t) min(n0 + row, a.N - 1) * a.ldb + chunk * 8;
    }
    auto stage = [&] (int kt)
    {
        char* base = lds + (kt & (GN2_NBUF - 1)) * GN2_BUF_BYTES;
        #pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            gnt_glds16(srcA[i] + (size_t) kt * GN2_BK, base + (i * 4 + wave) * 1024);
            gnt_glds16(srcB[i] + (size_t) kt * GN2_BK, base + GN2_OP_BYTES + (i * 4 + wave) * 1024);
        }
    };
    // fragment of 16 rows x 32 k: lane = row (lane & 15), chunk (lane >> 4) of the row's four
    const int fr = lane & 15, kc = lane >> 4;
    const int offA = (wm * 128 + fr) * 64 + ((kc ^ ((fr >> 2) & 3)) << 4);
    const int offB = GN2_OP_BYTES + (wn * 128 + fr) * 64 + ((kc ^ ((fr >> 2) & 3)) << 4);

    f32x4 acc[8][8];
    #pragma unroll
    for (int i = 0; i < 8; ++i)
        #pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{ 0.f, 0.f, 0.f, 0.f };

    const int nk = a.K / GN2_BK;
    half8_t fa[2][8], fb[2][8];
    auto rd = [&] (int set, int kt)
    {
        const char* base = lds + (kt & (GN2_NBUF - 1)) * GN2_BUF_BYTES;
        #pragma unroll
        for (int i = 0; i < 8; ++i) { fa[set][i] = *((const half8_t*) (base + offA + i * 16 * 64)); fb[set][i] = *((const half8_t*) (base + offB + i * 16 * 64)); }
    };
    // prologue: three tiles in flight, the first one's fragments in registers
    #pragma unroll
    for (int p = 0; p < GN2_NBUF - 1; ++p) if (p < nk) stage(p);
    if (nk >= 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rd(0, 0);

    auto body = [&] (int kt, int cs, int ns, bool more)
    {
        // 64 matrix instructions on fragment set cs; the 16 fragment reads of K-tile kt + 1 (set ns) are issued among them
        #pragma unroll
        for (int im = 0; im < 8; ++im)
        {
            if (more)
            {
                const char* base = lds + ((kt + 1) & (GN2_NBUF - 1)) * GN2_BUF_BYTES;
                fa[ns][im] = *((const half8_t*) (base + offA + im * 16 * 64)); fb[ns][im] = *((const half8_t*) (base + offB + im * 16 * 64));
            }
            #pragma unroll
            for (int jn = 0; jn < 8; ++jn)
                // accumulators pinned to the accumulation half of the register file, fragments to the vector half (left to itself the allocator shuffles
                // both halves around every instruction: 1302 v_accvgpr moves for 128 MFMAs and scratch reloads between the staging requests)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[im][jn]) : "v"(fa[cs][im]), "v"(fb[cs][jn]));
        }
    };
    for (int kt = 0; kt < nk; kt += 2)
    {
        // --- even tile: fragments in set 0
        if (kt + 3 < nk) stage(kt + 3);
        // tile kt + 1 has landed (this wave's requests: the two younger tiles stay in flight) and this wave's reads of tile kt are complete
        if (kt + 3 < nk) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        else if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        body(kt, 0, 1, kt + 1 < nk);
        if (kt + 1 >= nk) break;
        // --- odd tile: fragments in set 1
        if (kt + 4 < nk) stage(kt + 4);
        if (kt + 4 < nk) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
        else if (kt + 3 < nk) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        body(kt + 1, 1, 0, kt + 2 < nk);
    }
    __syncthreads();                                            // every wave is past its last fragment read: the buffers become the C tile

    // ---- epilogue through LDS: D tile (i, j) of the wave: row = wm*128 + i*16 + (lane >> 4)*4 + reg, col = wn*128 + j*16 + (lane & 15)
    half_t* ct = (half_t*) lds;
    #pragma unroll
    for (int i = 0; i < 8; ++i)
        #pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            const int col = wn * 128 + j * 16 + (lane & 15);
            #pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                const int row = wm * 128 + i * 16 + (lane >> 4) * 4 + r;
                ct[row * GNT_CPITCH + col] = f2h(acc[i][j][r]);
            }
        }
    __syncthreads();
    if (a.epi == GNT_EPI_SILU_MUL)
    {
        const int nout0 = n0 >> 1;
        for (int row = tid >> 4; row < GNT_BM; row += 16)
        {
            if (m0 + row >= a.M) break;
            const int c8 = (tid & 15) * 8;
            const half8_t g = *((const half8_t*) (ct + row * GNT_CPITCH + c8));
            const half8_t u = *((const half8_t*) (ct + row * GNT_CPITCH + 128 + c8));
            half8_t o;
            #pragma unroll
            for (int e = 0; e < 8; ++e)
            {
                const float gf = (float) g[e];
                o[e] = f2h(gf * __builtin_amdgcn_rcpf(1.0f + __expf(-gf)) * (float) u[e]);
            }
            *((half8_t*) (a.C + (size_t) (m0 + row) * a.ldc + nout0 + c8)) = o;
        }
        return;
    }
    for (int row = tid >> 5; row < GNT_BM; row += 8)
    {
        if (m0 + row >= a.M) break;
        const int c8 = (tid & 31) * 8;
        if (n0 + c8 >= a.N) continue;
        half8_t v = *((const half8_t*) (ct + row * GNT_CPITCH + c8));
        half_t* dst = a.C + (size_t) (m0 + row) * a.ldc + n0 + c8;
        if (a.epi == GNT_EPI_ACC)
        {
            const half8_t old = *((const half8_t*) dst);
            #pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = f2h((float) old[e] + (float) v[e]);
        }
        *((half8_t*) dst) = v;
    }
}

