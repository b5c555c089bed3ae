// reconstruct_had_slice: original-basis weights  W = diag(suh) . H128 . W_hat . H128 . diag(svh)  (fp16), one 128x128
// block per workgroup.  reference: exllamav3_ext/quant/reconstruct.cu:147-373 (shared-memory butterflies).
//
// MI355X design: the two 128-point Hadamard transforms are done on the MATRIX pipe as two 128x128x128 fp16 GEMMs
// against the +-1 Sylvester matrix, whose MFMA operand fragments are generated in registers from lane indices
// (sign = parity(popcount(i & j))) -- no Hadamard matrix in memory, no 7-stage butterfly with LDS round trips,
// and exact fp32 accumulation of exactly representable products.  Rounding points follow the reference: the
// intermediate (after the k-side transform and suh) is rounded to fp16, the final result is rounded once.
#include "exl3_common.cuh"
#include "exl3_api_internal.h"

#define RH_LD 136          // padded leading dimension (halves) of the 128x128 LDS images: 272 B rows -> conflict-free b128 reads

// fp16 +-1 Hadamard fragment for v_mfma_f32_16x16x32_f16: 8 halves, element s = H[a][b0 + s], b0 % 8 == 0
__device__ __forceinline__ half8_t had_frag(int a, int b0)
{
    const uint32_t p0 = __builtin_popcount(a & b0) & 1;
    // parity((a & 7) & s) for s = 0..7 as a bit mask
    const int a3 = a & 7;
    uint32_t m8 = 0;
    #pragma unroll
    for (int s = 0; s < 8; ++s) m8 |= (uint32_t) (__builtin_popcount(a3 & s) & 1) << s;
    m8 ^= p0 ? 0xffu : 0u;
    union { uint32_t u[4]; half8_t h; } f;
    #pragma unroll
    for (int r = 0; r < 4; ++r)
        f.u[r] = 0x3C003C00u ^ (((m8 >> (2 * r)) & 1u) << 15) ^ (((m8 >> (2 * r + 1)) & 1u) << 31);
    return f.h;
}

template <int K, int CB>
__global__ __launch_bounds__(256)
void reconstruct_had_kernel(half_t* __restrict__ out, const uint32_t* __restrict__ packed, const half_t* __restrict__ suh,
                            const half_t* __restrict__ svh, int tiles_n_total, int tile_n_offset, int64_t out_stride)
{
    constexpr int NW = 8 * K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* Wt = (half_t*) smem;                          // [n][RH_LD]  W_hat transposed; later the output staging [k'][RH_LD]
    half_t* T = Wt + 128 * RH_LD;                         // [k'][RH_LD] intermediate
    uint32_t* words = (uint32_t*) (T + 128 * RH_LD);      // [8 tile rows][8 tiles][NW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kb = blockIdx.y, nb = blockIdx.x;

    // ---- packed words of the 8x8 tiles
    for (int i = tid; i < 8 * 8 * NW; i += 256)
    {
        int tr = i / (8 * NW), rest = i % (8 * NW);
        words[i] = packed[((int64_t) (kb * 8 + tr) * tiles_n_total + tile_n_offset + nb * 8) * NW + rest];
    }
    __syncthreads();

    // ---- decode: thread -> (tile, column, row half): 8 consecutive k of one n -> one 16-byte store into Wt[n][k]
    #pragma unroll 2
    for (int it = 0; it < 8; ++it)
    {
        int unit = tid + 256 * it;
        int tile = unit >> 5, c = (unit & 31) >> 1, rh = unit & 1;
        int tr = tile >> 3, tcn = tile & 7;
        const uint32_t* w = words + (size_t) tile * NW;
        half8_t v;
        #pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = decode_exact<CB>(tile_state<K>(w, tile_stream_index(8 * rh + r, c)));
        *((half8_t*) (Wt + (size_t) (16 * tcn + c) * RH_LD + 16 * tr + 8 * rh)) = v;
    }
    __syncthreads();

    const int j = lane & 15, g = lane >> 4;
    const float r128 = HAD_R_SCALE_128;

    // ---- GEMM 1: D1[k'][n] = sum_k H[k'][k] * W_hat[k][n];   T = fp16(D1 * suh[k'] / sqrt(128))
    {
        half8_t ha[2][4];
        #pragma unroll
        for (int rt = 0; rt < 2; ++rt)
            #pragma unroll
            for (int ks = 0; ks < 4; ++ks) ha[rt][ks] = had_frag(16 * (2 * wave + rt) + j, 32 * ks + 8 * g);
        #pragma unroll 2
        for (int ct = 0; ct < 8; ++ct)
        {
            float4_t acc0 = { 0.f, 0.f, 0.f, 0.f }, acc1 = { 0.f, 0.f, 0.f, 0.f };
            #pragma unroll
            for (int ks = 0; ks < 4; ++ks)
            {
                half8_t bf = *((const half8_t*) (Wt + (size_t) (16 * ct + j) * RH_LD + 32 * ks + 8 * g));
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha[0][ks], bf, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha[1][ks], bf, acc1, 0, 0, 0);
            }
            #pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                int k0r = 16 * (2 * wave) + 4 * g + r, k1r = k0r + 16;
                float s0 = (float) suh[kb * 128 + k0r] * r128, s1 = (float) suh[kb * 128 + k1r] * r128;
                T[(size_t) k0r * RH_LD + 16 * ct + j] = (half_t) (acc0[r] * s0);
                T[(size_t) k1r * RH_LD + 16 * ct + j] = (half_t) (acc1[r] * s1);
            }
        }
    }
    __syncthreads();

    // ---- GEMM 2: Out[k'][n'] = sum_n T[k'][n] * H[n][n'];   out = fp16(Out * svh[n'] / sqrt(128))   (staged in Wt)
    {
        half8_t ta[2][4];
        #pragma unroll
        for (int rt = 0; rt < 2; ++rt)
            #pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                ta[rt][ks] = *((const half8_t*) (T + (size_t) (16 * (2 * wave + rt) + j) * RH_LD + 32 * ks + 8 * g));
        #pragma unroll 2
        for (int ct = 0; ct < 8; ++ct)
        {
            float4_t acc0 = { 0.f, 0.f, 0.f, 0.f }, acc1 = { 0.f, 0.f, 0.f, 0.f };
            #pragma unroll
            for (int ks = 0; ks < 4; ++ks)
            {
                half8_t hb = had_frag(16 * ct + j, 32 * ks + 8 * g);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ta[0][ks], hb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ta[1][ks], hb, acc1, 0, 0, 0);
            }
            const float sv = (float) svh[nb * 128 + 16 * ct + j] * r128;
            #pragma unroll
            for (int r = 0; r < 4; ++r)
            {
                int k0r = 16 * (2 * wave) + 4 * g + r, k1r = k0r + 16;
                Wt[(size_t) k0r * RH_LD + 16 * ct + j] = (half_t) (acc0[r] * sv);
                Wt[(size_t) k1r * RH_LD + 16 * ct + j] = (half_t) (acc1[r] * sv);
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
        *((half8_t*) (out + ((int64_t) kb * 128 + row) * out_stride + (int64_t) nb * 128 + 8 * seg)) = v;
    }
}

extern "C" int exl3_reconstruct_had(void* out, const void* trellis, const void* suh, const void* svh,
                                    int tiles_k, int tiles_n, int K, int cb, int64_t n_offset, int64_t n_size, void* stream)
{
    EXL3_CHECK_ARG(out && trellis && suh && svh, "reconstruct_had_slice: null pointer");
    EXL3_CHECK_ARG(K >= 1 && K <= 8, "reconstruct_had_slice: K must be in [1, 8]");
    EXL3_CHECK_ARG(cb >= 0 && cb <= 2, "reconstruct_had_slice: bad codebook");
    EXL3_CHECK_ARG(tiles_k % 8 == 0, "reconstruct_had_slice: K dimension must be divisible by 128");
    EXL3_CHECK_ARG(n_size % 128 == 0, "unpacked N dimension must be divisible by 128");
    EXL3_CHECK_ARG(n_offset % 128 == 0 && n_offset >= 0, "n_offset must be a non-negative multiple of 128");
    EXL3_CHECK_ARG(n_offset + n_size <= (int64_t) tiles_n * 16, "reconstruct slice exceeds packed tensor bounds");
    if (n_size == 0 || tiles_k == 0) return EXL3_OK;
    dim3 grid((unsigned) (n_size / 128), (unsigned) (tiles_k / 8));
    size_t lds = (size_t) 2 * 128 * RH_LD * 2 + (size_t) 64 * 8 * K * 4;
    hipStream_t st = (hipStream_t) stream;
    switch (K * 3 + cb)
    {
        #define RC(KK, CC) case KK * 3 + CC: reconstruct_had_kernel<KK, CC><<<grid, 256, lds, st>>>((half_t*) out, (const uint32_t*) trellis, \
            (const half_t*) suh, (const half_t*) svh, tiles_n, (int) (n_offset / 16), n_size); break;
        RC(1,0) RC(1,1) RC(1,2) RC(2,0) RC(2,1) RC(2,2) RC(3,0) RC(3,1) RC(3,2) RC(4,0) RC(4,1) RC(4,2)
        RC(5,0) RC(5,1) RC(5,2) RC(6,0) RC(6,1) RC(6,2) RC(7,0) RC(7,1) RC(7,2) RC(8,0) RC(8,1) RC(8,2)
        #undef RC
    }
    return exl3_check_launch("reconstruct_had_slice");
}
