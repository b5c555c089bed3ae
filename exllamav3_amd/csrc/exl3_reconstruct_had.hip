// reconstruct_had_slice: original-basis weights  W = diag(suh) . H128 . W_hat . H128 . diag(svh)  (fp16), one 128x128
// block per workgroup.  reference: exllamav3_ext/quant/reconstruct.cu:147-373 (shared-memory butterflies).
//
// MI355X design: the two 128-point Hadamard transforms are done on the MATRIX pipe as two 128x128x128 fp16 GEMMs
// against the +-1 Sylvester matrix, whose MFMA operand fragments are generated in registers from lane indices
// (sign = parity(popcount(i & j))) -- no Hadamard matrix in memory, no 7-stage butterfly with LDS round trips,
// and exact fp32 accumulation of exactly representable products.  (The float -> half conversions here are left to the compiler, which
// folds scale-multiply + convert into v_fma_mixlo_f16: one rounding instead of the reference's two, half the VALU work of the kernel's
// hottest non-MFMA part, and inside the op's 2e-3 tolerance; the decode-path kernels use f2h() because pipelines are compared bit for bit.)  Rounding points follow the reference: the
// intermediate (after the k-side transform and suh) is rounded to fp16, the final result is rounded once.
// The first GEMM is computed transposed so that its accumulators ARE the second GEMM's A operand (no LDS round trip for
// the intermediate, one 34.8 KB LDS image per workgroup).
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include "exl3_lane_decode.cuh"
#include <string.h>

// diagnostics builds (tools/alt_lib.py rh_x exl3_reconstruct_had.o -DRH_ABL_NOMFMA / -DRH_ABL_NOSTORE): timing-only ablations.  Round 6, gate shape (4096 x 14336): 42.7 us whole,
// 28.6 without the matrix instructions, 34.2 without the global stores, 13.9 without both -- the three parts ADD (no overlap to speak of between the four workgroups of a CU);
// packing the 2-byte LDS stores of the W^T epilogue into 8-byte ones changed nothing (44.1 us), nor did starting the four workgroups of a CU a quarter of a block apart (42.3 / 43.3)
#ifdef RH_ABL_NOMFMA
#define RH_MFMA(a, b, c) (c)
#else
#define RH_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#endif
#define RH_LD 136          // padded leading dimension (halves) of the 128x128 LDS images: 272 B rows -> conflict-free b128 reads

// Sylvester structure: H128[a][b] = H8[a >> 4][b >> 4] * H16[a & 15][b & 15].  For an MFMA fragment whose k-slots are
// b = 32 ks + 8 g + s (s = 0..7) and whose row/column index is a = 16 t + j, the 8 halves are
//     H8[t][2 ks + (g >> 1)] * H16[j][8 (g & 1) + s]
// i.e. ONE per-lane base fragment (depends on j, g & 1 only) times a +-1 scalar per (t, ks): 4 xors instead of 25 VALU ops.
__device__ __forceinline__ void had_base_frag(int j, int g, uint32_t (&f)[4])
{
    const int b0 = 8 * (g & 1);
    #pragma unroll
    for (int r = 0; r < 4; ++r)
    {
        uint32_t s0 = __builtin_popcount(j & (b0 + 2 * r)) & 1, s1 = __builtin_popcount(j & (b0 + 2 * r + 1)) & 1;
        f[r] = 0x3C003C00u ^ (s0 << 15) ^ (s1 << 31);
    }
}

__device__ __forceinline__ half8_t had_frag_signed(const uint32_t (&base)[4], int t, int ks, int g)
{
    const uint32_t neg = (__builtin_popcount(t & (2 * ks + (g >> 1))) & 1) ? 0x80008000u : 0u;
    union { uint32_t u[4]; half8_t h; } f;
    f.u[0] = base[0] ^ neg; f.u[1] = base[1] ^ neg; f.u[2] = base[2] ^ neg; f.u[3] = base[3] ^ neg;
    return f.h;
}

// row r (0..15) of the lane's column c (HI = 0) or c + 8 (HI = 1) is weight t = 8q + j
template <int R, int HI> struct RowToWeight { static constexpr int q = (R & 7) >> 1, j = (R & 1) + 2 * (R >> 3) + 4 * HI, t = 8 * q + j; };

// Up to 4 matrices of one launch (same k, bits and codebook), stacked along n in `out` in this order: q | k | v of the prefill route are 1536 workgroups
// of one launch instead of 1024 + 256 + 256 of three (k and v alone are one workgroup per CU: 10.4 us each for 2 MB).
#define RH_MAX_MATS 4
struct ReconMats
{
    const uint32_t* packed[RH_MAX_MATS]; const half_t* suh[RH_MAX_MATS]; const half_t* svh[RH_MAX_MATS];
    int tiles_n_total[RH_MAX_MATS], tile_n_offset[RH_MAX_MATS], nb_first[RH_MAX_MATS];      // nb_first: first 128-column block of matrix i in the grid
    int count;
    int interleave;                                       // 1: block row nb of matrix i is written at position nb * count + i (equal n_i): the tile pairing of a fused epilogue
};

#ifndef RH_WAVES
#define RH_WAVES 4          // (3: the compiler's own choice, 143 registers; 4: 97 registers, no scratch, four workgroups per CU: gate shape 43.3 -> 38.0 us)
#endif
template <int K, int CB, bool TR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RH_WAVES, RH_WAVES)))
void reconstruct_had_kernel(half_t* __restrict__ out, const ReconMats mt, int64_t out_stride)
{
    int mi = 0;
    #pragma unroll
    for (int i = 1; i < RH_MAX_MATS; ++i) if (i < mt.count && (int) blockIdx.x >= mt.nb_first[i]) mi = i;
    const uint32_t* __restrict__ packed = mt.packed[mi];
    const half_t* __restrict__ suh = mt.suh[mi];
    const half_t* __restrict__ svh = mt.svh[mi];
    const int tiles_n_total = mt.tiles_n_total[mi], tile_n_offset = mt.tile_n_offset[mi];
    const int nb_out = mt.interleave ? ((int) blockIdx.x - mt.nb_first[mi]) * mt.count + mi : (int) blockIdx.x;          // position in the stacked output
    constexpr int NW = 8 * K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* Wt = (half_t*) smem;                          // [n][RH_LD]  W_hat transposed; later the output staging [k'][RH_LD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kb = blockIdx.y, nb = (int) blockIdx.x - mt.nb_first[mi];         // block within its matrix
    // the scale vectors this lane needs (its k' rows in GEMM 1, its n' column of every column tile in GEMM 2) are requested FIRST, next to the packed
    // words: loaded where they are used, each was a dependent L2 round trip in the middle of the kernel (the 8 svh values one per column-tile trip),
    // and the kernel without decode, matrix instructions and stores still took 33 of 59 us (round 3 ablation, gate_proj shape)
    const int jq = lane & 15;
    const half_t su0 = suh[kb * 128 + 16 * (2 * wave) + jq], su1 = suh[kb * 128 + 16 * (2 * wave + 1) + jq];
    half_t svr[8];
    #pragma unroll
    for (int ct = 0; ct < 8; ++ct) svr[ct] = svh[nb * 128 + 16 * ct + jq];

    // ---- decode: wave w handles tile rows 2w, 2w+1; lane (8T + c) owns columns c, c+8 of tile T (exl3_lane_decode.cuh):
    //      one contiguous 256*K-byte load per wave and tile row, constant-shift windows, bit-exact fp16 values.
    {
        const int Tt = lane >> 3, c = lane & 7;
        const int prev_lane_addr = ((lane & ~7) | ((lane - 1) & 7)) << 2;
        #pragma unroll
        for (int rr = 0; rr < 2; ++rr)
        {
            const int tr = 2 * wave + rr;
            const uint32_t* src = packed + ((int64_t) (kb * 8 + tr) * tiles_n_total + tile_n_offset + nb * 8) * NW + (size_t) lane * K;
            LaneWords<K> lw;
            load_lane_words<K>(lw, src);
            uint32_t Wx[K + 1];
            #pragma unroll
            for (int i = 0; i < K; ++i) Wx[i + 1] = lw.w[i];
            Wx[0] = (uint32_t) __builtin_amdgcn_ds_bpermute(prev_lane_addr, (int) lw.w[K - 1]);
            // Exact fp16 weights four at a time (exl3_lane_decode.cuh decode_quad, EXACT variant: the reference's values bit for bit with
            // packed ops: 2.5 VALU per weight instead of ~5).  Quad t = 8q .. 8q+3 is rows (2q, 2q+1, 2q+8, 2q+9) of column c, quad
            // 8q+4 .. 8q+7 the same rows of column c + 8: the low dword of quad q is dword q of rows 0-7, the high dword that of rows 8-15.
            union { uint32_t u[4]; half8_t h; } lo0u, lo1u, hi0u, hi1u;
            #define DQ(Q) { half4_t qa[2], qb[2]; decode_quad<K, CB, 0, 8 * Q>(Wx, qa); decode_quad<K, CB, 0, 8 * Q + 4>(Wx, qb); \
                            union { half4_t h; uint32_t u[2]; } ca, cb2; ca.h = qa[0]; cb2.h = qb[0]; \
                            lo0u.u[Q] = ca.u[0]; lo1u.u[Q] = ca.u[1]; hi0u.u[Q] = cb2.u[0]; hi1u.u[Q] = cb2.u[1]; }
            DQ(0) DQ(1) DQ(2) DQ(3)
            #undef DQ
            const half8_t lo0 = lo0u.h, lo1 = lo1u.h, hi0 = hi0u.h, hi1 = hi1u.h;   // column c rows 0-7 / 8-15 ; column c+8 rows 0-7 / 8-15
            half_t* w0 = Wt + (size_t) (16 * Tt + c) * RH_LD + 16 * tr;
            half_t* w1 = Wt + (size_t) (16 * Tt + c + 8) * RH_LD + 16 * tr;
            *((half8_t*) w0) = lo0; *((half8_t*) (w0 + 8)) = lo1;
            *((half8_t*) w1) = hi0; *((half8_t*) (w1 + 8)) = hi1;
        }
    }
    __syncthreads();

    const int j = lane & 15, g = lane >> 4;
    const float r128 = HAD_R_SCALE_128;
    uint32_t hbase[4];
    had_base_frag(j, g, hbase);

    // ---- GEMM 1 (transposed): D1t[n][k'] = sum_k W_hat^T[n][k] * H[k][k'];  Tt = fp16(D1t * suh[k'] / sqrt(128))
    //      A = rows of Wt (LDS), B = Hadamard fragment.  The accumulator layout (lane: k' = j, four consecutive n = 4g..4g+3 per n-tile)
    //      is exactly what GEMM 2 needs for its A operand (row k' = j, n along the k-slots), so the intermediate never leaves registers:
    //      k-slot (g, s) of GEMM-2 step u holds n = 16(2u) + 4g + s for s < 4 and n = 16(2u+1) + 4g + s - 4 for s >= 4 -- a permutation of
    //      n that the generated Hadamard B fragments of GEMM 2 simply follow.
    half8_t ta[2][4];                                     // [k' tile of this wave][step u]
    {
        half8_t hb[2][4];
        #pragma unroll
        for (int rt = 0; rt < 2; ++rt)
            #pragma unroll
            for (int ks = 0; ks < 4; ++ks) hb[rt][ks] = had_frag_signed(hbase, 2 * wave + rt, ks, g);
        const float sc0 = (float) su0 * r128;
        const float sc1 = (float) su1 * r128;
        #pragma unroll
        for (int u = 0; u < 4; ++u)
        {
            float4_t acc[2][2];                           // [n-tile parity][k' tile]
            #pragma unroll
            for (int e = 0; e < 2; ++e)
            {
                acc[e][0] = float4_t{ 0.f, 0.f, 0.f, 0.f }; acc[e][1] = acc[e][0];
                #pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                {
                    half8_t af = *((const half8_t*) (Wt + (size_t) (16 * (2 * u + e) + j) * RH_LD + 32 * ks + 8 * g));
                    acc[e][0] = RH_MFMA(af, hb[0][ks], acc[e][0]);
                    acc[e][1] = RH_MFMA(af, hb[1][ks], acc[e][1]);
                }
            }
            ta[0][u] = half8_t{ (half_t) (acc[0][0][0] * sc0), (half_t) (acc[0][0][1] * sc0), (half_t) (acc[0][0][2] * sc0), (half_t) (acc[0][0][3] * sc0),
                                (half_t) (acc[1][0][0] * sc0), (half_t) (acc[1][0][1] * sc0), (half_t) (acc[1][0][2] * sc0), (half_t) (acc[1][0][3] * sc0) };
            ta[1][u] = half8_t{ (half_t) (acc[0][1][0] * sc1), (half_t) (acc[0][1][1] * sc1), (half_t) (acc[0][1][2] * sc1), (half_t) (acc[0][1][3] * sc1),
                                (half_t) (acc[1][1][0] * sc1), (half_t) (acc[1][1][1] * sc1), (half_t) (acc[1][1][2] * sc1), (half_t) (acc[1][1][3] * sc1) };
        }
    }
    __syncthreads();                                      // every wave is done reading Wt: it becomes the output staging buffer

    // ---- GEMM 2: Out[k'][n'] = sum_n T[k'][n] * H[n][n'];   out = fp16(Out * svh[n'] / sqrt(128))
    {
        // H16[4g + s][j] for s = 0..3: the per-lane base of the permuted-n Hadamard fragment
        uint32_t h16[2];
        #pragma unroll
        for (int r = 0; r < 2; ++r)
        {
            uint32_t s0 = __builtin_popcount(j & (4 * g + 2 * r)) & 1, s1 = __builtin_popcount(j & (4 * g + 2 * r + 1)) & 1;
            h16[r] = 0x3C003C00u ^ (s0 << 15) ^ (s1 << 31);
        }
        #pragma unroll
        for (int ct = 0; ct < 8; ++ct)
        {
            float4_t acc0 = { 0.f, 0.f, 0.f, 0.f }, acc1 = { 0.f, 0.f, 0.f, 0.f };
            #pragma unroll
            for (int u = 0; u < 4; ++u)
            {
                // signs H8[2u][ct] and H8[2u+1][ct]
                const uint32_t na = (__builtin_popcount((2 * u) & ct) & 1) ? 0x80008000u : 0u;
                const uint32_t nb2 = (__builtin_popcount((2 * u + 1) & ct) & 1) ? 0x80008000u : 0u;
                union { uint32_t w[4]; half8_t h; } f;
                f.w[0] = h16[0] ^ na; f.w[1] = h16[1] ^ na; f.w[2] = h16[0] ^ nb2; f.w[3] = h16[1] ^ nb2;
                acc0 = RH_MFMA(ta[0][u], f.h, acc0);
                acc1 = RH_MFMA(ta[1][u], f.h, acc1);
            }
            const float sv = (float) svr[ct] * r128;
            #pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                int k0r = 16 * (2 * wave) + 4 * g + r, k1r = k0r + 16;
                if constexpr (TR)
                {
                    // transposed output W^T[n'][k'] (k contiguous): the layout the MFMA GEMM library is fastest on (both operands K-major)
                    Wt[(size_t) (16 * ct + j) * RH_LD + k0r] = (half_t) (acc0[r] * sv);
                    Wt[(size_t) (16 * ct + j) * RH_LD + k1r] = (half_t) (acc1[r] * sv);
                }
                else
                {
                    Wt[(size_t) k0r * RH_LD + 16 * ct + j] = (half_t) (acc0[r] * sv);
                    Wt[(size_t) k1r * RH_LD + 16 * ct + j] = (half_t) (acc1[r] * sv);
                }
            }
        }
    }
    __syncthreads();

    // ---- coalesced store: 16 threads x 16 B per row
    #pragma unroll
    for (int it = 0; it < 8; ++it)
    {
        int row = (tid >> 4) + 16 * it, seg = tid & 15;
        half8_t v = *((const half8_t*) (Wt + (size_t) row * RH_LD + 8 * seg));
#ifdef RH_ABL_NOSTORE
        if (v[0] == (half_t) 12345.0f)
#endif
        if constexpr (TR) *((half8_t*) (out + ((int64_t) nb_out * 128 + row) * out_stride + (int64_t) kb * 128 + 8 * seg)) = v;     // row = n'
        else              *((half8_t*) (out + ((int64_t) kb * 128 + row) * out_stride + (int64_t) nb_out * 128 + 8 * seg)) = v;     // row = k'

    }
}

static int reconstruct_had_launch(void* out, int64_t ld_out, int transposed, const ReconMats& mt, int total_nb, int tiles_k, int K, int cb, void* stream)
{
    dim3 grid((unsigned) total_nb, (unsigned) (tiles_k / 8));
    size_t lds = (size_t) 128 * RH_LD * 2;                 // 34.8 KB: four workgroups per CU
    hipStream_t st = (hipStream_t) stream;
    #define RC(KK, CC) case KK * 3 + CC: \
        if (transposed) reconstruct_had_kernel<KK, CC, true><<<grid, 256, lds, st>>>((half_t*) out, mt, ld_out); \
        else reconstruct_had_kernel<KK, CC, false><<<grid, 256, lds, st>>>((half_t*) out, mt, ld_out); break;
    switch (K * 3 + cb)
    {
        RC(1,0) RC(1,1) RC(1,2) RC(2,0) RC(2,1) RC(2,2) RC(3,0) RC(3,1) RC(3,2) RC(4,0) RC(4,1) RC(4,2)
        RC(5,0) RC(5,1) RC(5,2) RC(6,0) RC(6,1) RC(6,2) RC(7,0) RC(7,1) RC(7,2) RC(8,0) RC(8,1) RC(8,2)
    }
    #undef RC
    return exl3_check_launch("reconstruct_had_slice");
}

static int reconstruct_had_impl(void* out, int64_t ld_out, int transposed, const void* trellis, const void* suh, const void* svh,
                                int tiles_k, int tiles_n, int K, int cb, int64_t n_offset, int64_t n_size, void* stream)
{
    EXL3_CHECK_ARG(out && trellis && suh && svh, "reconstruct_had_slice: null pointer");
    EXL3_CHECK_ARG(K >= 1 && K <= 8, "reconstruct_had_slice: K must be in [1, 8]");
    EXL3_CHECK_ARG(cb >= 0 && cb <= 2, "reconstruct_had_slice: bad codebook");
    EXL3_CHECK_ARG(tiles_k % 8 == 0, "reconstruct_had_slice: K dimension must be divisible by 128");
    EXL3_CHECK_ARG(n_size % 128 == 0, "unpacked N dimension must be divisible by 128");
    EXL3_CHECK_ARG(n_offset % 128 == 0 && n_offset >= 0, "n_offset must be a non-negative multiple of 128");
    EXL3_CHECK_ARG(n_offset + n_size <= (int64_t) tiles_n * 16, "reconstruct slice exceeds packed tensor bounds");
    EXL3_CHECK_ARG(ld_out >= (transposed ? (int64_t) tiles_k * 16 : n_size) && ld_out % 8 == 0, "reconstruct_had_slice: bad output row stride");
    if (n_size == 0 || tiles_k == 0) return EXL3_OK;
    ReconMats mt; memset((void*) &mt, 0, sizeof(mt));
    mt.packed[0] = (const uint32_t*) trellis; mt.suh[0] = (const half_t*) suh; mt.svh[0] = (const half_t*) svh;      // (the slice's own svh, as in the reference's narrowed tensor)
    mt.tiles_n_total[0] = tiles_n; mt.tile_n_offset[0] = (int) (n_offset / 16); mt.nb_first[0] = 0; mt.count = 1;
    return reconstruct_had_launch(out, ld_out, transposed, mt, (int) (n_size / 128), tiles_k, K, cb, stream);
}

// W^T of up to 4 matrices with the same k, bits per weight and codebook, stacked along n in `out` ([sum n_i][ld_out], matrix i's rows behind matrix
// i - 1's): one launch for the fused q|k|v (or gate|up) GEMM of the prefill route.  tiles_n[i] = n_i / 16 (whole matrices; n_i % 128 == 0).
extern "C" int exl3_reconstruct_had_multi_t(void* out, int64_t ld_out, const void* const* trellis, const void* const* suh, const void* const* svh,
                                            const int* tiles_n, int count, int tiles_k, int K, int cb, void* stream)
{
    EXL3_CHECK_ARG(out && trellis && suh && svh && tiles_n, "reconstruct_had_multi_t: null pointer");
    EXL3_CHECK_ARG(count >= 1 && count <= RH_MAX_MATS, "reconstruct_had_multi_t: between 1 and 4 matrices");
    EXL3_CHECK_ARG(K >= 1 && K <= 8 && cb >= 0 && cb <= 2, "reconstruct_had_multi_t: K must be in [1, 8], codebook in [0, 2]");
    EXL3_CHECK_ARG(tiles_k % 8 == 0 && tiles_k > 0, "reconstruct_had_multi_t: K dimension must be divisible by 128");
    EXL3_CHECK_ARG(ld_out >= (int64_t) tiles_k * 16 && ld_out % 8 == 0, "reconstruct_had_multi_t: bad output row stride");
    ReconMats mt; memset((void*) &mt, 0, sizeof(mt));
    int nb = 0;
    for (int i = 0; i < count; ++i)
    {
        EXL3_CHECK_ARG(trellis[i] && suh[i] && svh[i] && tiles_n[i] > 0 && tiles_n[i] % 8 == 0, "reconstruct_had_multi_t: null matrix / n not divisible by 128");
        mt.packed[i] = (const uint32_t*) trellis[i]; mt.suh[i] = (const half_t*) suh[i]; mt.svh[i] = (const half_t*) svh[i];
        mt.tiles_n_total[i] = tiles_n[i]; mt.tile_n_offset[i] = 0; mt.nb_first[i] = nb;
        nb += tiles_n[i] / 8;
    }
    mt.count = count;
    return reconstruct_had_launch(out, ld_out, 1, mt, nb, tiles_k, K, cb, stream);
}

// The same launch with the matrices' 128-row blocks INTERLEAVED: block row j of matrix i lands at rows (j * count + i) * 128 (all n_i equal).  count = 2 is the operand the
// prefill GEMM's fused silu(gate) * up epilogue wants: every 256-row tile of W^T = 128 gate rows | the 128 up rows of the same outputs (exl3_gemm_nt2_mfma, epi 2).
extern "C" int exl3_reconstruct_had_multi_t_interleaved(void* out, int64_t ld_out, const void* const* trellis, const void* const* suh, const void* const* svh,
                                                        const int* tiles_n, int count, int tiles_k, int K, int cb, void* stream)
{
    EXL3_CHECK_ARG(out && trellis && suh && svh && tiles_n, "reconstruct_had_multi_t_interleaved: null pointer");
    EXL3_CHECK_ARG(count >= 1 && count <= RH_MAX_MATS, "reconstruct_had_multi_t_interleaved: between 1 and 4 matrices");
    EXL3_CHECK_ARG(K >= 1 && K <= 8 && cb >= 0 && cb <= 2, "reconstruct_had_multi_t_interleaved: K must be in [1, 8], codebook in [0, 2]");
    EXL3_CHECK_ARG(tiles_k % 8 == 0 && tiles_k > 0, "reconstruct_had_multi_t_interleaved: K dimension must be divisible by 128");
    EXL3_CHECK_ARG(ld_out >= (int64_t) tiles_k * 16 && ld_out % 8 == 0, "reconstruct_had_multi_t_interleaved: bad output row stride");
    ReconMats mt; memset((void*) &mt, 0, sizeof(mt));
    int nb = 0;
    for (int i = 0; i < count; ++i)
    {
        EXL3_CHECK_ARG(trellis[i] && suh[i] && svh[i] && tiles_n[i] > 0 && tiles_n[i] % 8 == 0 && tiles_n[i] == tiles_n[0],
                       "reconstruct_had_multi_t_interleaved: null matrix / n not divisible by 128 / matrices of different n");
        mt.packed[i] = (const uint32_t*) trellis[i]; mt.suh[i] = (const half_t*) suh[i]; mt.svh[i] = (const half_t*) svh[i];
        mt.tiles_n_total[i] = tiles_n[i]; mt.tile_n_offset[i] = 0; mt.nb_first[i] = nb;
        nb += tiles_n[i] / 8;
    }
    mt.count = count; mt.interleave = 1;
    return reconstruct_had_launch(out, ld_out, 1, mt, nb, tiles_k, K, cb, stream);
}

extern "C" int exl3_reconstruct_had(void* out, const void* trellis, const void* suh, const void* svh,
                                    int tiles_k, int tiles_n, int K, int cb, int64_t n_offset, int64_t n_size, void* stream)
{
    return reconstruct_had_impl(out, n_size, 0, trellis, suh, svh, tiles_k, tiles_n, K, cb, n_offset, n_size, stream);
}

// Same reconstruction, written TRANSPOSED: out[n_size][ld_out] holds W^T (row = output feature, k contiguous; ld_out >= k, so several
// matrices can share one buffer row-wise).  The prefill GEMM then has both operands K-major ("NT"), the layout hipBLASLt's MFMA kernels
// are 15-28 % faster on (tools/bench_gemm_layouts.py).
extern "C" int exl3_reconstruct_had_t(void* out, int64_t ld_out, const void* trellis, const void* suh, const void* svh,
                                      int tiles_k, int tiles_n, int K, int cb, int64_t n_offset, int64_t n_size, void* stream)
{
    return reconstruct_had_impl(out, ld_out, 1, trellis, suh, svh, tiles_k, tiles_n, K, cb, n_offset, n_size, stream);
}
