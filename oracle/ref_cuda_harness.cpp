// ORACLE / TEST INFRASTRUCTURE -- not product code.
//
// extern "C" harness around the REFERENCE's own device code for the codebooks, the trellis window readers and the KV-cache quantizer,
// compiled for the host from where it lies under /root/reference (oracle/build_ref.sh -> oracle/_ref/libexl3_ref_cuda.so) on top of
// oracle/cuda_host_shim.h.  Nothing from the reference is copied into this repository; the headers are #included by path:
//   quant/codebook.cuh:56-123   decode_3inst<cb>, decode_3inst_2<cb>
//   quant/exl3_dq.cuh:15-293    dq<bits, cb>, dq_dispatch<bits, cb> (the per-bitrate fast paths the GEMM kernels use)
//   cache/q_cache_kernels.cuh   quant_block_x4 / dequant_block_x4 (:61-236), quant_cache_paged_kernel / dequant_cache_paged_kernel (:289-399)
// tests/test_oracle_pins.py compares oracle/exl3_oracle.py with these bit for bit.
#include "cuda_host_shim.h"
#include "quant/exl3_dq.cuh"               // -I/root/reference/exllamav3/exllamav3_ext ; pulls quant/codebook.cuh
#include "cache/q_cache_kernels.cuh"       // pulls cache/lmq.cuh

template <int cb> static void decode_all(uint16_t* single, uint16_t* pair_lo, uint16_t* pair_hi)
{
    for (uint32_t s = 0; s < 65536; ++s)
    {
        single[s] = decode_3inst<cb>(s).bits;
        // the two-at-a-time form the kernels use: (s, 65535 - s)
        const half2 p = decode_3inst_2<cb>(s, 65535u - s);
        pair_lo[s] = p.x.bits; pair_hi[s] = p.y.bits;
    }
}

template <int bits, int cb> static void dq_tile(const uint32_t* words, uint16_t* generic, uint16_t* dispatch)
{
    for (int t = 0; t < 256; ++t) generic[t] = dq<bits, cb>(words, t).bits;
    for (int t = 0; t < 256; t += 8)
    {
        FragB f0, f1;
        dq_dispatch<bits, cb>(words, t, f0, f1);
        dispatch[t + 0] = f0[0].x.bits; dispatch[t + 1] = f0[0].y.bits; dispatch[t + 2] = f0[1].x.bits; dispatch[t + 3] = f0[1].y.bits;
        dispatch[t + 4] = f1[0].x.bits; dispatch[t + 5] = f1[0].y.bits; dispatch[t + 6] = f1[1].x.bits; dispatch[t + 7] = f1[1].y.bits;
    }
}

template <int cb> static int dq_tile_bits(int bits, const uint32_t* w, uint16_t* g, uint16_t* d)
{
    switch (bits)
    {
        case 1: dq_tile<1, cb>(w, g, d); return 0;
        case 2: dq_tile<2, cb>(w, g, d); return 0;
        case 3: dq_tile<3, cb>(w, g, d); return 0;
        case 4: dq_tile<4, cb>(w, g, d); return 0;
        case 5: dq_tile<5, cb>(w, g, d); return 0;
        case 6: dq_tile<6, cb>(w, g, d); return 0;
        case 7: dq_tile<7, cb>(w, g, d); return 0;
        case 8: dq_tile<8, cb>(w, g, d); return 0;
    }
    return -1;
}

template <int bits> static void kv_quant_groups(const half* in, uint32_t* out, half* scales, int num_groups, float compand_a)
{
    // one emulated warp per 4 groups, exactly the span quant_cache_cont_kernel hands to quant_block_x4
    for (int g0 = 0; g0 < num_groups; g0 += 4)
    {
        static uint32_t sh_pack[32];
        const int active = min(4, num_groups - g0);
        shim::run_warp([&] { quant_block_x4<bits>(in + g0 * 32, out + g0 * bits, scales + g0, sh_pack, active, compand_a); });
    }
}
template <int bits> static void kv_dequant_groups(const uint32_t* in, const half* scales, half* out, int num_groups, float compand_a)
{
    for (int g0 = 0; g0 < num_groups; g0 += 4)
    {
        const int active = min(4, num_groups - g0);
        shim::run_warp([&] { dequant_block_x4<bits>(in + g0 * bits, scales + g0, out + g0 * 32, active, compand_a); });
    }
}

extern "C" {

// all 65536 states of codebook cb (0 = 3INST, 1 = mcg, 2 = mul1): fp16 bit patterns
int ref_cuda_decode_all(int cb, uint16_t* single, uint16_t* pair_lo, uint16_t* pair_hi)
{
    if (cb == 0) decode_all<0>(single, pair_lo, pair_hi); else if (cb == 1) decode_all<1>(single, pair_lo, pair_hi);
    else if (cb == 2) decode_all<2>(single, pair_lo, pair_hi); else return -1;
    return 0;
}

// one 16x16 tile (8 * bits u32 words): the 256 decoded weights in stream order through the generic reader and through dq_dispatch
int ref_cuda_dq_tile(int bits, int cb, const uint32_t* words, uint16_t* generic, uint16_t* dispatch)
{
    if (cb == 0) return dq_tile_bits<0>(bits, words, generic, dispatch);
    if (cb == 1) return dq_tile_bits<1>(bits, words, generic, dispatch);
    if (cb == 2) return dq_tile_bits<2>(bits, words, generic, dispatch);
    return -1;
}

int ref_cuda_kv_quant(int bits, const uint16_t* in, uint32_t* out, uint16_t* scales, int num_groups, float compand_a)
{
    const half* i = (const half*) in; half* s = (half*) scales;
    switch (bits)
    {
        case 2: kv_quant_groups<2>(i, out, s, num_groups, compand_a); return 0;
        case 3: kv_quant_groups<3>(i, out, s, num_groups, compand_a); return 0;
        case 4: kv_quant_groups<4>(i, out, s, num_groups, compand_a); return 0;
        case 5: kv_quant_groups<5>(i, out, s, num_groups, compand_a); return 0;
        case 6: kv_quant_groups<6>(i, out, s, num_groups, compand_a); return 0;
        case 7: kv_quant_groups<7>(i, out, s, num_groups, compand_a); return 0;
        case 8: kv_quant_groups<8>(i, out, s, num_groups, compand_a); return 0;
    }
    return -1;
}

int ref_cuda_kv_dequant(int bits, const uint32_t* in, const uint16_t* scales, uint16_t* out, int num_groups, float compand_a)
{
    const half* s = (const half*) scales; half* o = (half*) out;
    switch (bits)
    {
        case 2: kv_dequant_groups<2>(in, s, o, num_groups, compand_a); return 0;
        case 3: kv_dequant_groups<3>(in, s, o, num_groups, compand_a); return 0;
        case 4: kv_dequant_groups<4>(in, s, o, num_groups, compand_a); return 0;
        case 5: kv_dequant_groups<5>(in, s, o, num_groups, compand_a); return 0;
        case 6: kv_dequant_groups<6>(in, s, o, num_groups, compand_a); return 0;
        case 7: kv_dequant_groups<7>(in, s, o, num_groups, compand_a); return 0;
        case 8: kv_dequant_groups<8>(in, s, o, num_groups, compand_a); return 0;
    }
    return -1;
}

// The paged kernels with their own index arithmetic (block table, cache_seqlens, in_contiguous, sliding window), launched the way
// cache/q_cache.cu does but with one 32-thread warp per block (the kernels derive their group offsets from blockDim.x).
// k_bits == v_bits == bits here (the instance table is indexed [k_bits - 2][v_bits - 2]).
int ref_cuda_quant_cache_paged(int bits, const uint16_t* k_in, uint32_t* k_out, uint16_t* k_scales, const uint16_t* v_in, uint32_t* v_out, uint16_t* v_scales,
                               const uint32_t* cache_seqlens, const uint32_t* block_table, int blocks_per_seq, int groups_per_token,
                               int bsz, int seq_len, float compand_a, int in_contiguous)
{
    if (bits < 2 || bits > 8) return -1;
    auto kern = quant_cache_paged_kernel_instances[bits - 2][bits - 2];
    blockDim.x = 32; gridDim.x = (unsigned) ((groups_per_token + 3) / 4); gridDim.y = (unsigned) seq_len; gridDim.z = (unsigned) bsz;
    for (unsigned z = 0; z < gridDim.z; ++z) for (unsigned y = 0; y < gridDim.y; ++y) for (unsigned x = 0; x < gridDim.x; ++x)
    {
        blockIdx.x = x; blockIdx.y = y; blockIdx.z = z;
        shim::run_warp([&] { kern((const half*) k_in, k_out, (half*) k_scales, (const half*) v_in, v_out, (half*) v_scales, cache_seqlens, block_table,
                                  blocks_per_seq, groups_per_token, compand_a, in_contiguous); });
    }
    return 0;
}

int ref_cuda_dequant_cache_paged(int bits, const uint32_t* k_in, const uint16_t* k_scales, uint16_t* k_out, const uint32_t* v_in, const uint16_t* v_scales,
                                 uint16_t* v_out, const uint32_t* cache_seqlens, const uint32_t* block_table, int pages_per_seq, int groups_per_token,
                                 int bsz, int max_tokens, int sliding_window, float compand_a, int compact_out, int bonus_len)
{
    // launch geometry of cache/q_cache.cu:309-320 (the sliding-window skip works on whole thread blocks, so the geometry is part of the result):
    // blockDim = min(32 * chunks_per_seq, 32 * MAX_WARPS), grid.x = ceil(ceil(chunks_per_seq / MAX_WARPS) / ITER_PER_TB); the warps of a
    // block do not communicate, so they are run one after the other.  max_tokens = pages_per_seq * page_size of the real call (or fewer to
    // save time: blocks past the longest sequence do nothing).
    if (bits < 2 || bits > 8) return -1;
    auto kern = dequant_cache_paged_kernel_instances[bits - 2][bits - 2];
    const int chunks_per_token = (groups_per_token + 3) / 4;
    const long chunks_per_seq = (long) pages_per_seq * CQ_PAGE_SIZE * chunks_per_token;
    const int warps = (int) (chunks_per_seq < MAX_WARPS ? chunks_per_seq : MAX_WARPS);
    blockDim.x = 32 * warps; gridDim.y = (unsigned) bsz; gridDim.z = 1;
    const long used_chunks = (long) max_tokens * chunks_per_token;
    gridDim.x = (unsigned) (((used_chunks + warps - 1) / warps + ITER_PER_TB - 1) / ITER_PER_TB);
    for (unsigned y = 0; y < gridDim.y; ++y) for (unsigned x = 0; x < gridDim.x; ++x) for (int w = 0; w < warps; ++w)
    {
        blockIdx.x = x; blockIdx.y = y; blockIdx.z = 0;
        shim::run_warp([&] { kern(k_in, (const half*) k_scales, (half*) k_out, v_in, (const half*) v_scales, (half*) v_out, cache_seqlens, block_table,
                                  pages_per_seq, groups_per_token, chunks_per_token, sliding_window, compand_a, compact_out, bonus_len); }, w);
    }
    return 0;
}

}
