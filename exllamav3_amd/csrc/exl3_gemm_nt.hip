// Hand-written NT MFMA GEMM for the prefill route (rows > 144):   C[m][n] (+)= A[m][k] @ Wt[n][k]^T
//
//   A  = the activations (tokens x k, fp16, k contiguous), Wt = W^T as exl3_reconstruct_had_slice_t writes it (n x k, k contiguous): both operands
//   K-major.  reference path: modules/quant/exl3.py:161-218 (reconstruct + hgemm), hgemm.cu:19-78 (cublasGemmEx); north_star: "prefill dequants a
//   tile into LDS then runs MFMA".  This is the contraction; the dequant + both Hadamards stay in exl3_reconstruct_had.hip (the
//   Hadamards cannot sit inside a tiled GEMM without doubling its work).  hipBLASLt (exl3_hgemm.hip) is the default route (it measured 1.2-1.3x
//   faster on every prefill shape) and the fallback for shapes this kernel does not take; the host side (ext.hgemm_nt*) takes this kernel only
//   when EXL3_HIP_GEMM_NT=1 is set (unset or 0 = library).
//
// Structure (cdna_hip_programming.md section 5, written for gfx950 only):
//   * 256 x 256 x 64 tile, 8 waves as 2 (m) x 4 (n): a wave owns 128 x 64 of C = 8 x 4 tiles of v_mfma_f32_16x16x32_f16, 128 accumulator VGPRs.
//   * Operands go HBM -> LDS with global_load_lds (16 B per lane, no VGPR round trip, no ds_write pass).  The LDS image of a tile is the lane-linear
//     image the instruction writes, [row][64 halves] = 128 B per row, and the bank-conflict-free read pattern is obtained by permuting the SOURCE
//     address: the 16-byte chunk at LDS position p of row r holds global chunk p ^ ((r >> 1) & 7).  A fragment read (16 lanes = 16 consecutive rows,
//     the same k chunk) then touches 16 distinct (row parity, slot) pairs = all 64 banks exactly once.
//   * Two LDS buffers of one K-tile each (2 x 64 KB); the next K-tile is requested at the start of a tile and waited for at its end; the two wave groups run
//     one step apart (main loop), so that LDS reads and MFMAs of the two waves of a SIMD alternate instead of coinciding.
//   * Workgroup -> tile mapping: every XCD gets a contiguous chunk of the tile list (workgroup i runs on XCD i % 8), walked in groups of 8 (m) x 4 (n)
//     tiles so that the 32 workgroups that share an L2 at any time read 8 A panels and 4 B panels, not 32 + 32.
//   * Epilogue through LDS (the operand buffers are free by then): accumulators -> fp16 tile -> full 512-byte rows out, with what the library GEMM
//     cannot fuse:  EPI_ACC  c = fp16(c + acc)  (the residual add of o_proj / down_proj: one rounding, as fp32 output + `x += y`),
//                   EPI_SILU_MUL  the tile holds 128 gate columns | 128 up columns of the same 128 outputs (the caller stacks W^T rows that way):
//                                 c = fp16(silu(fp16 g) * fp16 u) -- activation.cu silu_mul, and half the output bytes.
#include "exl3_common.cuh"
#include "exl3_api_internal.h"

#define GNT_BM 256
#define GNT_BN 256
#define GNT_BK 64
#define GNT_EPI_STORE 0
#define GNT_EPI_ACC 1
#define GNT_EPI_SILU_MUL 2
#define GNT_TILE_BYTES (256 * 64 * 2)                 // one operand, one K-tile: 32 KB
#define GNT_BUF_BYTES (2 * GNT_TILE_BYTES)            // A | B
#define GNT_CPITCH 264                                // epilogue tile row pitch in halves (528 B: rows 8 banks apart)
#define GNT_LDS_BYTES (256 * GNT_CPITCH * 2)          // 135168 B >= 2 * GNT_BUF_BYTES

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GntArgs
{
    const half_t* A; const half_t* Bt; half_t* C;
    int64_t lda, ldb, ldc;
    int M, N, K, epi;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ void gnt_glds16(const void* gptr, void* lds_uniform_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) gptr, (__attribute__((address_space(3))) void*) lds_uniform_base, 16, 0, 0);
}

__global__ __launch_bounds__(512)
void exl3_gemm_nt_kernel(const GntArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the two wave groups of the main loop: waves 0-3 / 4-7.  Wave w runs on SIMD w % 4 (measured: pairing waves (w, w ^ 1) or (w, w ^ 2) instead
    // put both waves of a SIMD in the same group and cost 45 % -- tools/bench_gemm_nt_small.py history), so each SIMD holds one wave of each group
    const int wm = wave >> 2, wn = wave & 3;

    // ---- tile of this workgroup (XCD-aware, bijective for any grid size)
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

    // ---- staging: 4 + 4 global_load_lds per thread per K-tile.  Instruction i of wave w covers rows (i*8 + w)*8 .. +7 of the operand tile; lane L
    // supplies LDS position (L & 7) of row (L >> 3) of that group and fetches global chunk (L & 7) ^ swz(row), swz(row) = (row >> 1) & 7
    const half_t* srcA[4]; const half_t* srcB[4];
    #pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const int row = (i * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        srcA[i] = a.A + (size_t) min(m0 + row, a.M - 1) * a.lda + chunk * 8;
        srcB[i] = a.Bt + (size_t) min(n0 + row, a.N - 1) * a.ldb + chunk * 8;
    }
    auto stage = [&] (int buf, int kt)
    {
        char* base = lds + buf * GNT_BUF_BYTES;
        #pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            gnt_glds16(srcA[i] + (size_t) kt * GNT_BK, base + (i * 8 + wave) * 1024);
            gnt_glds16(srcB[i] + (size_t) kt * GNT_BK, base + GNT_TILE_BYTES + (i * 8 + wave) * 1024);
        }
    };

    // ---- fragment read offsets (bytes inside an operand tile): row = 16-row block base + (lane & 15), chunk = ks * 4 + (lane >> 4)
    const int fr = lane & 15, kg = lane >> 4, sw = (lane >> 1) & 7;
    int offA[2], offB[2];
    #pragma unroll
    for (int ks = 0; ks < 2; ++ks)
    {
        offA[ks] = (wm * 128 + fr) * 128 + (((ks * 4 + kg) ^ sw) << 4);
        offB[ks] = GNT_TILE_BYTES + (wn * 64 + fr) * 128 + (((ks * 4 + kg) ^ sw) << 4);
    }

    f32x4 acc[8][4];
    #pragma unroll
    for (int i = 0; i < 8; ++i)
        #pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{ 0.f, 0.f, 0.f, 0.f };

    // ---- main loop: the two wave groups (wm = 0: waves 0-3, wm = 1: waves 4-7; one wave of each per SIMD) run the same four steps per K-tile
    //      R0: 12 fragment reads (first 32-deep half)   M0: 32 MFMAs   R1: 12 fragment reads (second half)   M1: 32 MFMAs
    // separated by workgroup barriers, but ONE STEP APART: while a SIMD's group-0 wave multiplies, its group-1 wave reads LDS, and vice versa -- the
    // matrix pipe and the LDS port are both busy in every slot (in lockstep both waves of a SIMD read together, then multiply together: the first
    // version of this kernel, 1.19 PFLOP/s at 4096^3).  Slots of K-tile t:        4t+0          4t+1   4t+2   4t+3
    //                                                            group 0:  stage(t+1), R0     M0     R1     M1, wait(t+1)
    //                                                            group 1:  stage(t+1), M1(t-1) R0     M0     R1, wait(t+1)
    // stage(t+1) overwrites the buffer of tile t-1, last read in slot 4(t-1)+3; wait(t+1) = s_waitcnt vmcnt(0), three slots after the request.
    const int nk = a.K / GNT_BK;
    half8_t fa[8], fb[4];
    auto R = [&] (const char* base, int ks)
    {
        #pragma unroll
        for (int jn = 0; jn < 4; ++jn) fb[jn] = *((const half8_t*) (base + offB[ks] + jn * 16 * 128));
        #pragma unroll
        for (int im = 0; im < 8; ++im) fa[im] = *((const half8_t*) (base + offA[ks] + im * 16 * 128));
    };
    auto M = [&] ()
    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        #pragma unroll
        for (int im = 0; im < 8; ++im)
            #pragma unroll
            for (int jn = 0; jn < 4; ++jn) acc[im][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[im], fb[jn], acc[im][jn], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wm == 0)
    {
        for (int kt = 0; kt < nk; ++kt)
        {
            const char* base = lds + (kt & 1) * GNT_BUF_BYTES;
            if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
            R(base, 0);
            __builtin_amdgcn_s_barrier();
            M();
            __builtin_amdgcn_s_barrier();
            R(base, 1);
            __builtin_amdgcn_s_barrier();
            M();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    else
    {
        for (int kt = 0; kt < nk; ++kt)
        {
            const char* base = lds + (kt & 1) * GNT_BUF_BYTES;
            if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
            if (kt > 0) M();                                    // M1 of the previous K-tile
            __builtin_amdgcn_s_barrier();
            R(base, 0);
            __builtin_amdgcn_s_barrier();
            M();
            __builtin_amdgcn_s_barrier();
            R(base, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        M();
    }
    __syncthreads();                                            // every wave is past its last fragment read: the buffers become the C tile

    // ---- epilogue through LDS: D tile (i, j) of the wave: row = wm*128 + i*16 + (lane >> 4)*4 + reg, col = wn*64 + j*16 + (lane & 15)
    half_t* ct = (half_t*) lds;
    #pragma unroll
    for (int i = 0; i < 8; ++i)
        #pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const int col = wn * 64 + j * 16 + (lane & 15);
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
        // columns 0..127 = gate, 128..255 = up of outputs (n0 / 2) .. +127: 16 threads per row, 8 outputs each
        const int nout0 = n0 >> 1;
        for (int row = tid >> 4; row < GNT_BM; row += 32)
        {
            if (m0 + row >= a.M) break;
            const int c8 = (tid & 15) * 8;
            const half8_t g = *((const half8_t*) (ct + row * GNT_CPITCH + c8));
            const half8_t u = *((const half8_t*) (ct + row * GNT_CPITCH + 128 + c8));
            half8_t o;
            #pragma unroll
            for (int e = 0; e < 8; ++e)
            {
                // silu(g) * u with v_exp_f32 + v_rcp_f32 (1 ulp each): the full-precision division of the standalone silu_mul kernel costs
                // ~9 us per tile here, where no MFMA work is left to hide it (measured: +91 us on the 4096 x 28672 GEMM)
                const float gf = (float) g[e];
                o[e] = f2h(gf * __builtin_amdgcn_rcpf(1.0f + __expf(-gf)) * (float) u[e]);
            }
            *((half8_t*) (a.C + (size_t) (m0 + row) * a.ldc + nout0 + c8)) = o;
        }
        return;
    }
    for (int row = tid >> 5; row < GNT_BM; row += 16)
    {
        if (m0 + row >= a.M) break;
        const int c8 = (tid & 31) * 8;
        if (n0 + c8 >= a.N) continue;
        half8_t v = *((const half8_t*) (ct + row * GNT_CPITCH + c8));
        half_t* dst = a.C + (size_t) (m0 + row) * a.ldc + n0 + c8;
        if (a.epi == GNT_EPI_ACC)
        {
            // c = fp16(c + y): y is the fp16-rounded GEMM output here; the library route (beta = 1) adds the fp32 accumulator -- both inside the
            // prefill tolerance, this one is the reference's `x += y` on fp16 tensors exactly
            const half8_t old = *((const half8_t*) dst);
            #pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = f2h((float) old[e] + (float) v[e]);
        }
        *((half8_t*) dst) = v;
    }
}

// c[m][n] (ldc) = a[m][k] (lda) @ bt[n][k]^T (ldb); epi: 0 store, 1 c += (fp16), 2 silu(gate) * up on 128 | 128 column pairs (c has n / 2 columns).
// Returns EXL3_ERR_ARG for shapes outside the kernel (the caller falls back to the library): k % 64, n % 256 (silu: the tile pairing), alignment.
extern "C" int exl3_gemm_nt_mfma(const void* a, int64_t lda, const void* bt, int64_t ldb, void* c, int64_t ldc, int m, int k, int n, int epi, void* stream)
{
    EXL3_CHECK_ARG(a && bt && c && m >= 1 && k >= 64 && n >= 256, "gemm_nt_mfma: null pointer / empty problem");
    EXL3_CHECK_ARG(k % 64 == 0 && n % 256 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "gemm_nt_mfma: k % 64, n % 256, leading dimensions % 8");
    EXL3_CHECK_ARG(((uintptr_t) a | (uintptr_t) bt | (uintptr_t) c) % 16 == 0, "gemm_nt_mfma: 16-byte aligned operands");
    EXL3_CHECK_ARG(epi >= 0 && epi <= 2, "gemm_nt_mfma: bad epilogue");
    static bool attr_set = false;
    if (!attr_set)
    {
        EXL3_CHECK_HIP(hipFuncSetAttribute((const void*) exl3_gemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GNT_LDS_BYTES), "gemm_nt_mfma: LDS size");
        attr_set = true;
    }
    GntArgs g;
    g.A = (const half_t*) a; g.Bt = (const half_t*) bt; g.C = (half_t*) c; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = m; g.N = n; g.K = k; g.epi = epi;
    g.tiles_m = (m + GNT_BM - 1) / GNT_BM; g.tiles_n = n / GNT_BN;
    exl3_gemm_nt_kernel<<<g.tiles_m * g.tiles_n, 512, GNT_LDS_BYTES, (hipStream_t) stream>>>(g);
    return exl3_check_launch("gemm_nt_mfma");
}
