// Format ops for EXL3 tensors on gfx950: pack / unpack / pack_signs / decode / reconstruct / had_r_128.
// All integer work is bit-exact with the reference semantics (oracle/exl3_oracle.py); these kernels are
// HBM-bound byte shufflers: one coalesced 16 B access per lane wherever the layout allows.
#include "exl3_common.cuh"
#include "exl3_api_internal.h"

// ------------------------------------------------------------------------------------------------
// unpack_trellis: packed tile -> 256 16-bit states in stream order.  reference: quant/pack.cu:97-138
// One wave per tile; lane handles 4 consecutive stream indices (one 8-byte store per lane).
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256)
void unpack_trellis_kernel(uint16_t* __restrict__ unpacked, const uint32_t* __restrict__ packed, int64_t num_tiles)
{
    constexpr int NW = 8 * K;
    __shared__ uint32_t s_w[4][NW];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t tile = (int64_t) blockIdx.x * 4 + wave;
    if (tile < num_tiles)
    {
        if (lane < NW) s_w[wave][lane] = packed[tile * NW + lane];
    }
    __syncthreads();
    if (tile >= num_tiles) return;
    uint32_t s0 = tile_state<K>(s_w[wave], lane * 4 + 0);
    uint32_t s1 = tile_state<K>(s_w[wave], lane * 4 + 1);
    uint32_t s2 = tile_state<K>(s_w[wave], lane * 4 + 2);
    uint32_t s3 = tile_state<K>(s_w[wave], lane * 4 + 3);
    uint2_t o = { s0 | (s1 << 16), s2 | (s3 << 16) };
    ((uint2_t*) (unpacked + tile * 256))[lane] = o;
}

// ------------------------------------------------------------------------------------------------
// pack_trellis: low K bits of each state, MSB-first, 16-bit chunks stored pair-swapped.  pack.cu:9-57
// One wave per tile.  Output word i (u32) = stream bits [32i, 32i+32) MSB-first.
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256)
void pack_trellis_kernel(uint32_t* __restrict__ packed, const uint16_t* __restrict__ unpacked, int64_t num_tiles)
{
    constexpr int NW = 8 * K;
    __shared__ uint16_t s_u[4][256];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t tile = (int64_t) blockIdx.x * 4 + wave;
    if (tile < num_tiles)
        ((uint2_t*) s_u[wave])[lane] = ((const uint2_t*) (unpacked + tile * 256))[lane];
    __syncthreads();
    if (tile >= num_tiles || lane >= NW) return;
    // word `lane` covers stream bits [32*lane, 32*lane + 32)
    int p0 = 32 * lane;
    uint32_t word = 0;
    int t = p0 / K;
    int p = t * K;                               // first bit of symbol t
    while (p < p0 + 32)
    {
        uint32_t sym = (uint32_t) s_u[wave][t] & ((1u << K) - 1u);
        int sh = (p0 + 32) - (p + K);            // left shift of the symbol's LSB inside the word
        if (sh >= 0) word |= sym << sh; else word |= sym >> (-sh);
        p += K; t++;
    }
    // the head symbol may start before p0: bits above the word are dropped by the 32-bit shift
    packed[tile * NW + lane] = word;
}

// pack_signs: fp16 [numel] -> int16 [numel/16], bit b of word w = sign of element 16w+b.  pack.cu:177-201
__global__ __launch_bounds__(256)
void pack_signs_kernel(uint16_t* __restrict__ packed, const uint16_t* __restrict__ signs, int64_t words)
{
    int64_t w = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    const uint4_t* src = (const uint4_t*) (signs + w * 16);
    uint4_t a = src[0], b = src[1];
    uint32_t v[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
    uint32_t r = 0;
    #pragma unroll
    for (int i = 0; i < 8; ++i)
    {
        r |= ((v[i] >> 15) & 1u) << (2 * i);
        r |= ((v[i] >> 31) & 1u) << (2 * i + 1);
    }
    packed[w] = (uint16_t) r;
}

// decode: states -> codebook values.  quantize.cu:89-168
template <int CB, bool FP32>
__global__ __launch_bounds__(256)
void decode_kernel(const uint16_t* __restrict__ states, void* __restrict__ out, int64_t numel)
{
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numel) return;
    half_t v = decode_exact<CB>((uint32_t) states[i]);
    if constexpr (FP32) ((float*) out)[i] = (float) v;
    else ((half_t*) out)[i] = v;
}

// ------------------------------------------------------------------------------------------------
// reconstruct: packed -> W_hat fp16 row-major (rotated basis).  reconstruct.cu:13-84
// Block = 256 threads = 16 rows x 128 columns (8 tiles).  Thread (r, seg) decodes 8 consecutive columns of
// row r and issues one 16-byte store; a row's 16 threads write 256 contiguous bytes.
// ------------------------------------------------------------------------------------------------
template <int K, int CB>
__global__ __launch_bounds__(256)
void reconstruct_kernel(half_t* __restrict__ out, const uint32_t* __restrict__ packed,
                        int tiles_n_total, int tile_n_offset, int64_t out_stride)
{
    constexpr int NW = 8 * K;
    __shared__ uint32_t s_w[8][NW];
    int t = threadIdx.x;
    int kt = blockIdx.y;
    int nt0 = blockIdx.x * 8;
    const uint32_t* src = packed + ((int64_t) kt * tiles_n_total + tile_n_offset + nt0) * NW;
    for (int i = t; i < 8 * NW; i += 256) (&s_w[0][0])[i] = src[i];
    __syncthreads();

    int r = t >> 4, seg = t & 15;
    int tile = seg >> 1, chalf = seg & 1;
    half8_t v;
    #pragma unroll
    for (int c = 0; c < 8; ++c)
    {
        int ti = tile_stream_index(r, c + 8 * chalf);
        v[c] = decode_exact<CB>(tile_state<K>(s_w[tile], ti));
    }
    half_t* dst = out + ((int64_t) kt * 16 + r) * out_stride + (int64_t) nt0 * 16 + seg * 8;
    *((half8_t*) dst) = v;
}

// ------------------------------------------------------------------------------------------------
// had_r_128: y = (x.view(-1,128) @ H128) * scale / sqrt(128), optional fp16 pre / post scale.  hadamard.cu:88-173
// One 32-lane half-wave per 128-vector, 4 elements per lane; block = 256 threads = 8 vectors.
// ------------------------------------------------------------------------------------------------
template <bool FP32, int SCALE_MODE>   // 0 none, 1 pre, 2 post
__global__ __launch_bounds__(256)
void had_r_128_kernel(const void* __restrict__ in, void* __restrict__ out, const half_t* __restrict__ scale,
                      float r_scale, int64_t num_vecs, int blocks_per_row)
{
    int64_t vec = (int64_t) blockIdx.x * 8 + (threadIdx.x >> 5);
    int l = threadIdx.x & 31;
    bool active = vec < num_vecs;
    int64_t vsafe = active ? vec : 0;
    int cb = (int) (vsafe % blocks_per_row);
    float h0, h1, h2, h3;
    half4_t sc = {};
    if constexpr (SCALE_MODE != 0) sc = ((const half4_t*) (scale + cb * 128))[l];
    if constexpr (FP32)
    {
        float4_t v = ((const float4_t*) ((const float*) in + vsafe * 128))[l];
        if constexpr (SCALE_MODE == 1) { v.x *= (float) sc.x; v.y *= (float) sc.y; v.z *= (float) sc.z; v.w *= (float) sc.w; }
        h0 = v.x; h1 = v.y; h2 = v.z; h3 = v.w;
    }
    else
    {
        half4_t v = ((const half4_t*) ((const half_t*) in + vsafe * 128))[l];
        if constexpr (SCALE_MODE == 1) v = v * sc;          // fp16 multiply, as __hmul2
        h0 = (float) v.x; h1 = (float) v.y; h2 = (float) v.z; h3 = (float) v.w;
    }
    had128_f32x4(h0, h1, h2, h3, l);
    h0 *= r_scale; h1 *= r_scale; h2 *= r_scale; h3 *= r_scale;
    if (!active) return;
    if constexpr (FP32)
    {
        float4_t o = { h0, h1, h2, h3 };
        if constexpr (SCALE_MODE == 2) { o.x *= (float) sc.x; o.y *= (float) sc.y; o.z *= (float) sc.z; o.w *= (float) sc.w; }
        ((float4_t*) ((float*) out + vec * 128))[l] = o;
    }
    else
    {
        half4_t o = { f2h(h0), f2h(h1), f2h(h2), f2h(h3) };
        if constexpr (SCALE_MODE == 2) o = o * sc;
        ((half4_t*) ((half_t*) out + vec * 128))[l] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Host launchers (C-ABI bodies)
// ------------------------------------------------------------------------------------------------

#define K_SWITCH(K, CALL) \
    switch (K) { case 1: { constexpr int KK = 1; CALL; } break; case 2: { constexpr int KK = 2; CALL; } break; \
                 case 3: { constexpr int KK = 3; CALL; } break; case 4: { constexpr int KK = 4; CALL; } break; \
                 case 5: { constexpr int KK = 5; CALL; } break; case 6: { constexpr int KK = 6; CALL; } break; \
                 case 7: { constexpr int KK = 7; CALL; } break; case 8: { constexpr int KK = 8; CALL; } break; }

#define CB_SWITCH(cb, CALL) \
    switch (cb) { case 0: { constexpr int CC = 0; CALL; } break; case 1: { constexpr int CC = 1; CALL; } break; \
                  case 2: { constexpr int CC = 2; CALL; } break; }

extern "C" int exl3_unpack_trellis(void* unpacked, const void* packed, int tiles_k, int tiles_n, int K, void* stream)
{
    EXL3_CHECK_ARG(unpacked && packed, "unpack_trellis: null pointer");
    EXL3_CHECK_ARG(K >= 1 && K <= 8, "unpack_trellis: K must be in [1, 8]");
    int64_t tiles = (int64_t) tiles_k * tiles_n;
    if (tiles == 0) return EXL3_OK;
    dim3 grid((unsigned) ((tiles + 3) / 4));
    K_SWITCH(K, (unpack_trellis_kernel<KK><<<grid, dim3(256), 0, (hipStream_t) stream>>>((uint16_t*) unpacked, (const uint32_t*) packed, tiles)));
    return exl3_check_launch("unpack_trellis");
}

extern "C" int exl3_pack_trellis(void* packed, const void* unpacked, int tiles_k, int tiles_n, int K, void* stream)
{
    EXL3_CHECK_ARG(unpacked && packed, "pack_trellis: null pointer");
    EXL3_CHECK_ARG(K >= 1 && K <= 8, "pack_trellis: K must be in [1, 8]");
    int64_t tiles = (int64_t) tiles_k * tiles_n;
    if (tiles == 0) return EXL3_OK;
    dim3 grid((unsigned) ((tiles + 3) / 4));
    K_SWITCH(K, (pack_trellis_kernel<KK><<<grid, dim3(256), 0, (hipStream_t) stream>>>((uint32_t*) packed, (const uint16_t*) unpacked, tiles)));
    return exl3_check_launch("pack_trellis");
}

extern "C" int exl3_pack_signs(void* packed, const void* signs, int64_t numel, void* stream)
{
    EXL3_CHECK_ARG(packed && signs, "pack_signs: null pointer");
    EXL3_CHECK_ARG(numel % 16 == 0, "pack_signs: numel must be divisible by 16");
    int64_t words = numel / 16;
    if (words == 0) return EXL3_OK;
    pack_signs_kernel<<<dim3((unsigned) ((words + 255) / 256)), dim3(256), 0, (hipStream_t) stream>>>((uint16_t*) packed, (const uint16_t*) signs, words);
    return exl3_check_launch("pack_signs");
}

extern "C" int exl3_decode(const void* states, void* out, int64_t numel, int out_fp32, int cb, void* stream)
{
    EXL3_CHECK_ARG(states && out, "decode: null pointer");
    EXL3_CHECK_ARG(cb >= 0 && cb <= 2, "decode: bad codebook");
    if (numel == 0) return EXL3_OK;
    dim3 grid((unsigned) ((numel + 255) / 256));
    if (out_fp32) { CB_SWITCH(cb, (decode_kernel<CC, true><<<grid, dim3(256), 0, (hipStream_t) stream>>>((const uint16_t*) states, out, numel))); }
    else          { CB_SWITCH(cb, (decode_kernel<CC, false><<<grid, dim3(256), 0, (hipStream_t) stream>>>((const uint16_t*) states, out, numel))); }
    return exl3_check_launch("decode");
}

extern "C" int exl3_reconstruct(void* out, const void* trellis, int tiles_k, int tiles_n, int K, int cb,
                                int64_t n_offset, int64_t n_size, void* stream)
{
    EXL3_CHECK_ARG(out && trellis, "reconstruct: null pointer");
    EXL3_CHECK_ARG(K >= 1 && K <= 8, "reconstruct: K must be in [1, 8]");
    EXL3_CHECK_ARG(cb >= 0 && cb <= 2, "reconstruct: bad codebook");
    EXL3_CHECK_ARG(n_size % 128 == 0, "unpacked N dimension must be divisible by 128");
    EXL3_CHECK_ARG(n_offset % 128 == 0, "n_offset must be divisible by 128");
    EXL3_CHECK_ARG(n_offset >= 0, "n_offset must be non-negative");
    EXL3_CHECK_ARG(n_offset + n_size <= (int64_t) tiles_n * 16, "reconstruct slice exceeds packed tensor bounds");
    if (n_size == 0 || tiles_k == 0) return EXL3_OK;
    dim3 grid((unsigned) (n_size / 128), (unsigned) tiles_k);
    K_SWITCH(K, CB_SWITCH(cb, (reconstruct_kernel<KK, CC><<<grid, dim3(256), 0, (hipStream_t) stream>>>((half_t*) out, (const uint32_t*) trellis, tiles_n, (int) (n_offset / 16), n_size))));
    return exl3_check_launch("reconstruct");
}

extern "C" int exl3_had_r_128(const void* in, void* out, const void* pre_scale, const void* post_scale, float scale,
                              int rows, int cols, int fp32, void* stream)
{
    EXL3_CHECK_ARG(in && out, "had_r_128: null pointer");
    EXL3_CHECK_ARG(cols % 128 == 0, "had_r_128: dim 1 must be divisible by 128");
    int64_t vecs = (int64_t) rows * (cols / 128);
    if (vecs == 0) return EXL3_OK;
    float r_scale = scale * HAD_R_SCALE_128;
    dim3 grid((unsigned) ((vecs + 7) / 8));
    int bpr = cols / 128;
    hipStream_t s = (hipStream_t) stream;
    #define HAD_LAUNCH(F, M, SC) had_r_128_kernel<F, M><<<grid, dim3(256), 0, s>>>(in, out, (const half_t*) (SC), r_scale, vecs, bpr)
    if (fp32) { if (pre_scale) HAD_LAUNCH(true, 1, pre_scale); else if (post_scale) HAD_LAUNCH(true, 2, post_scale); else HAD_LAUNCH(true, 0, nullptr); }
    else      { if (pre_scale) HAD_LAUNCH(false, 1, pre_scale); else if (post_scale) HAD_LAUNCH(false, 2, post_scale); else HAD_LAUNCH(false, 0, nullptr); }
    #undef HAD_LAUNCH
    return exl3_check_launch("had_r_128");
}
