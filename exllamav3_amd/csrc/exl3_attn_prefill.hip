// Prefill (multi-token) causal attention over the paged fp16 K/V that dequant_cache_paged expands -- the attention step of the reference's
// prefill path: CacheLayer_quant.get_kv (cache/quant.py:83-117) dequantizes the pages, then flash_attn_with_kvcache(q, k_pages, v_pages,
// block_table, cache_seqlens, causal=True) attends (modules/attn.py).  SURVEY.md 8(f)3, "next" row: the decode branch is exl3_attn_decode.hip.
//
// gfx950 design: flash-attention forward, one workgroup = 64 consecutive queries of the GW query heads that share one kv head of one sequence,
// K/V tiles of 64 keys staged in LDS.  Everything is shaped so that no register transpose is needed between the two matrix products
// (v_mfma_f32_16x16x32_f16; operand lane (g = lane / 16, c = lane % 16) holds row / column c and contraction slots 8g .. 8g+7, the result lane
// holds column c, rows 4g .. 4g+3):
//   * scores are computed TRANSPOSED, S^T = K Q^T (A = K rows from LDS, B = the wave's Q rows, kept in registers for the whole kernel), so a
//     lane ends up with the probabilities of ONE query (column c) for keys {16 kb + 4g + j}: exactly an A operand of P V if contraction slot
//     8g + 4h + j of the second product is DEFINED to be key 16 (2 kb2 + h) + 4g + j -- a contraction index can be permuted freely as long
//     as both operands agree;
//   * V is staged row-major like K (16-byte LDS writes); the matching B operand (4 consecutive keys of one output column) is gathered by
//     ds_read_b64_tr_b16, gfx950's LDS transpose read - one instruction per 4 keys, no transposing write pass;
//   * the GW (1, 2 or 4) query heads that share a kv head sit in ONE workgroup, so a K/V tile is staged once for all of them;
//   * softmax statistics live with the query's column lanes; the output accumulator's rows are queries 4g + j, so the running rescale factor
//     of those four queries is fetched from lanes 4g + j (ds_bpermute, 4 per tile).
// fp32 softmax and accumulation, fp16 probabilities into the second product (as flash-attention does), fp16 output.
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include "exl3_gemv_args.h"

#define PA_BM 64          // queries per workgroup
#define PA_BN 64          // keys per tile

struct PrefillAttnArgs
{
    const half_t* q; half_t* out;                               // [bsz][q_len][hq][HD]
    const half_t* k_pages; const half_t* v_pages;               // [pages][page_size][hkv][HD] fp16 (dequant_cache_paged's output, or an fp16 cache)
    const int32_t* block_table; const int32_t* cache_seqlens;   // [bsz][blocks_per_seq]; [bsz] = tokens in the cache INCLUDING the q_len new ones
    int q_len, hq, hkv, blocks_per_seq, page_size;
    int64_t ldq;                                                // halves between consecutive tokens of q (hq * HD when contiguous)
    float scale;
};

typedef short s16x4_t __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ half4_t lds_read_tr16(const half_t* p)
{
    const s16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*) p);
    return __builtin_bit_cast(half4_t, r);
}

// GW query heads of ONE kv head per workgroup (GW in {1, 2, 4}, GW | heads_q / heads_kv): all of its waves share every staged K/V tile, so the
// staging work per matrix instruction drops by GW (with one head per workgroup the four q heads of a Llama kv head each re-staged it).
// A wave owns QG 16-query groups of one head (4 / QG waves per head): with QG = 2 every K and V fragment read from LDS feeds two matrix
// instructions at 2 waves per SIMD, with QG = 1 the kernel fits 128 VGPRs and runs 4 waves per SIMD; the launcher picks by workgroup size.
template <int HD, int GW, int QG>
__global__ __launch_bounds__(256 * GW / QG)
void attn_prefill_kernel(const PrefillAttnArgs a)
{
    constexpr int WPH = 4 / QG;                                 // waves per head: each owns QG 16-query groups of the 64-query tile
    constexpr int NT = 64 * GW * WPH;
    constexpr int KS = HD + 8;                                  // tile row stride in halves (16-byte aligned rows, rotated banks)
    __shared__ __attribute__((aligned(16))) half_t Ks[PA_BN * KS];
    constexpr int VS = HD + 16;                                 // V rows 8 dwords apart (mod 64 banks): the 16 four-lane row segments of a transpose read tile the banks
    __shared__ __attribute__((aligned(16))) half_t Vs[PA_BN * VS];   // row-major like K; the PV operand is gathered with the LDS transpose read
    const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6, wave = wave_all % WPH, g = lane >> 4, c = lane & 15;
    // Dispatch order = blockIdx.x fastest: the head group is the fast index and the query tile the slow one, LONGEST tiles first, so the workgroups
    // reach the CUs in order of decreasing length (a causal tile qt walks qt + 1 key tiles; with the tile index fast, the second wave of workgroups
    // handed 64-tile jobs to CUs that had already worked 17, 33, 49 tiles: 1.7x imbalance).  Consecutive workgroups also go to consecutive XCDs, so
    // with 8 kv heads each head's K/V stays in one XCD's L2.
    const int qt = gridDim.y - 1 - blockIdx.y, hg = blockIdx.x, h = hg * GW + wave_all / WPH, b = blockIdx.z;
    const int q_len = a.q_len, hq = a.hq, hkv = a.hkv, page_size = a.page_size;
    const int kvh = (hg * GW) / (hq / hkv);             // the same for all GW heads of the workgroup
    const int kv_len = a.cache_seqlens[b];
    const int ctx = kv_len - q_len;                             // query i sits at position ctx + i and sees keys 0 .. ctx + i
    const int q0 = qt * PA_BM + wave * 16 * QG;                 // this wave's first query; group u covers q0 + 16 u .. + 15
    const int32_t* bt = a.block_table + (size_t) b * a.blocks_per_seq;
    const float sl = a.scale * 1.44269504f;                      // scale * log2(e)

    // Q rows of the wave as B operands of S^T = K Q^T: contraction slots 8g .. 8g+7 of step ks are head dims 32 ks + 8g ..
    half8_t qf[QG][HD / 32];
    int qpos[QG];                                                // this lane's query (column of S^T) per group; clamped rows are never stored
    #pragma unroll
    for (int u = 0; u < QG; ++u)
    {
        const int qi = min(q0 + 16 * u + c, q_len - 1);
        qpos[u] = ctx + qi;
        const half_t* qp = a.q + ((size_t) b * q_len + qi) * a.ldq + (size_t) h * HD;
        #pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) qf[u][ks] = *((const half8_t*) (qp + 32 * ks + 8 * g));
    }
    float m_run[QG], l_run[QG];
    #pragma unroll
    for (int u = 0; u < QG; ++u) { m_run[u] = -1.0e30f; l_run[u] = 0.0f; }
    float4_t oc[QG][HD / 16];
    #pragma unroll
    for (int u = 0; u < QG; ++u)
        #pragma unroll
        for (int nb = 0; nb < HD / 16; ++nb) oc[u][nb] = float4_t{ 0.f, 0.f, 0.f, 0.f };

    const int q_last = min(qt * PA_BM + PA_BM, q_len) - 1;      // last query of the workgroup: the tiles needed are those up to its position
    const int ntiles = (ctx + q_last) / PA_BN + 1;
    // K/V tiles travel HBM -> registers -> LDS; the loads of tile kt + 1 are issued before the matrix work of tile kt and land in LDS after
    // the barrier that ends it (one register set: 2 x 16 bytes per staging slot)
    constexpr int CH = HD / 8;                                  // 16-byte chunks per key
    constexpr int SLOTS = (PA_BN * CH + NT - 1) / NT;
    half8_t kreg[SLOTS], vreg[SLOTS];
    auto fetch_tile = [&](int kt)
    {
        const int key0 = kt * PA_BN;                            // a tile never straddles a page (64 | page size)
        const int64_t page = bt[min(key0 / page_size, a.blocks_per_seq - 1)];
        #pragma unroll
        for (int j = 0; j < SLOTS; ++j)
        {
            const int idx = tid + NT * j;
            const int key = idx / CH, ch = idx % CH;
            const bool ok = key0 + key < kv_len && idx < PA_BN * CH;
            const size_t row = ((size_t) page * page_size + ((key0 + key) % page_size)) * hkv + kvh;
            half8_t kv = { 0, 0, 0, 0, 0, 0, 0, 0 }, vv = kv;
            if (ok) { kv = *((const half8_t*) (a.k_pages + row * HD + 8 * ch)); vv = *((const half8_t*) (a.v_pages + row * HD + 8 * ch)); }
            kreg[j] = kv; vreg[j] = vv;
        }
    };
    fetch_tile(0);
    for (int kt = 0; kt < ntiles; ++kt)
    {
        const int key0 = kt * PA_BN;
        __syncthreads();                                        // the previous tile is no longer read
        #pragma unroll
        for (int j = 0; j < SLOTS; ++j)
        {
            const int idx = tid + NT * j;
            if ((PA_BN * CH) % NT != 0 && idx >= PA_BN * CH) break;
            const int key = idx / CH, ch = idx % CH;
            *((half8_t*) (Ks + key * KS + 8 * ch)) = kreg[j];
            *((half8_t*) (Vs + key * VS + 8 * ch)) = vreg[j];
        }
        __syncthreads();
        if (kt + 1 < ntiles) fetch_tile(kt + 1);
        // ---- S^T block kb: keys key0 + 16 kb + (4g + j) x this lane's query of either group
        float4_t st[QG][PA_BN / 16];
        #pragma unroll
        for (int kb = 0; kb < PA_BN / 16; ++kb)
        {
            float4_t acc[QG];
            #pragma unroll
            for (int u = 0; u < QG; ++u) acc[u] = float4_t{ 0.f, 0.f, 0.f, 0.f };
            #pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks)
            {
                const half8_t ka = *((const half8_t*) (Ks + (16 * kb + c) * KS + 32 * ks + 8 * g));
                #pragma unroll
                for (int u = 0; u < QG; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qf[u][ks], acc[u], 0, 0, 0);
            }
            #pragma unroll
            for (int u = 0; u < QG; ++u) st[u][kb] = acc[u];
        }
        // ---- online softmax for this lane's queries: its 16 values here + the 3 other lane groups of the column.  Scores stay unscaled:
        // p = 2^((s - max) * scale * log2 e) is one fma + v_exp_f32 per value.  Only tiles on the diagonal or at the end of the sequence are masked
        // (a wave-uniform branch); a masked score is -1e30 and underflows to p = 0 because every query has key 0 in tile 0 (finite running max).
        half4_t pa[QG][PA_BN / 16];
        float corr[QG];
        const bool masked = key0 + PA_BN - 1 > ctx + min(q0, q_len - 1) || key0 + PA_BN > kv_len;
        if (masked)
        {
            #pragma unroll
            for (int u = 0; u < QG; ++u)
                #pragma unroll
                for (int kb = 0; kb < PA_BN / 16; ++kb)
                    #pragma unroll
                    for (int j = 0; j < 4; ++j)
                    {
                        const int key = key0 + 16 * kb + 4 * g + j;
                        if (!(key <= qpos[u] && key < kv_len)) st[u][kb][j] = -1.0e30f;
                    }
        }
        // Deferred running max: the reference point m_run of the exponent moves only when some query of the wave saw a score more than 8 (log2 units)
        // above it -- p <= 2^8 otherwise, well inside the fp16 / fp32 range -- so the accumulator rescale (64 multiplies + 8 ds_bpermute behind the
        // other waves' fragment reads) runs in a few early tiles instead of most of them (with 32 queries per wave SOME query's max moves in most tiles)
        float mxu[QG];
        bool need = false;
        #pragma unroll
        for (int u = 0; u < QG; ++u)
        {
            float mx = st[u][0][0];
            #pragma unroll
            for (int kb = 0; kb < PA_BN / 16; ++kb)
                mx = fmaxf(fmaxf(mx, fmaxf(st[u][kb][0], st[u][kb][1])), fmaxf(st[u][kb][2], st[u][kb][3]));
            mx = fmaxf(mx, xor_lane(mx, 16)); mx = fmaxf(mx, xor_lane(mx, 32));
            mxu[u] = mx;
            need = need || (mx - m_run[u]) * sl > 8.0f;
        }
        const bool upd = __any(need);
        #pragma unroll
        for (int u = 0; u < QG; ++u)
        {
            float mx = m_run[u];
            corr[u] = 1.0f;
            if (upd)
            {
                mx = fmaxf(mx, mxu[u]);
                corr[u] = __builtin_amdgcn_exp2f((m_run[u] - mx) * sl);
            }
            const float mxs = mx * sl;
            float ps = 0.0f;
            #pragma unroll
            for (int kb = 0; kb < PA_BN / 16; ++kb)
            {
                float p[4];
                #pragma unroll
                for (int j = 0; j < 4; ++j) { p[j] = __builtin_amdgcn_exp2f(fmaf(st[u][kb][j], sl, -mxs)); ps += p[j]; }
                pa[u][kb] = half4_t{ (half_t) p[0], (half_t) p[1], (half_t) p[2], (half_t) p[3] };
            }
            ps += xor_lane(ps, 16); ps += xor_lane(ps, 32);
            l_run[u] = l_run[u] * corr[u] + ps;
            m_run[u] = mx;
        }
        // ---- rescale the accumulator rows (queries 4g + j of each group) where the running max moved, and add P V
        if (upd)
        {
            float cr[QG][4];
            #pragma unroll
            for (int u = 0; u < QG; ++u)
                #pragma unroll
                for (int j = 0; j < 4; ++j) cr[u][j] = __shfl(corr[u], 4 * g + j, 64);
            #pragma unroll
            for (int u = 0; u < QG; ++u)
                #pragma unroll
                for (int nb = 0; nb < HD / 16; ++nb)
                {
                    oc[u][nb].x *= cr[u][0]; oc[u][nb].y *= cr[u][1]; oc[u][nb].z *= cr[u][2]; oc[u][nb].w *= cr[u][3];
                }
        }
        #pragma unroll
        for (int nb = 0; nb < HD / 16; ++nb)
        {
            #pragma unroll
            for (int k2 = 0; k2 < PA_BN / 32; ++k2)
            {
                // contraction slot 8g + 4h + j = key 16 (2 k2 + h) + 4g + j: A = [P block 2 k2 | P block 2 k2 + 1], B = the same keys of V column (16 nb + c).
                // ds_read_b64_tr_b16: the 16 lanes of a group hand in the addresses of a [4 keys][16 dims] block (lane c: key c >> 2, dims 4 (c & 3) ..)
                // and lane c gets column c of it, i.e. V[key 0 .. 3][16 nb + c] - the transposition the B operand needs, done by the LDS unit
                const half_t* vr = Vs + (32 * k2 + 4 * g + (c >> 2)) * VS + 16 * nb + 4 * (c & 3);
                const half4_t v0 = lds_read_tr16(vr), v1 = lds_read_tr16(vr + 16 * VS);
                const half8_t vB = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
                #pragma unroll
                for (int u = 0; u < QG; ++u)
                {
                    const half4_t p0 = pa[u][2 * k2], p1 = pa[u][2 * k2 + 1];
                    const half8_t pA = { p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w };
                    oc[u][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pA, vB, oc[u][nb], 0, 0, 0);
                }
            }
        }
    }
    // ---- normalise and store: rows = queries q0 + 16 u + 4g + j, columns 16 nb + c
    #pragma unroll
    for (int u = 0; u < QG; ++u)
    {
        float li[4];
        #pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            // a query without any visible key (cache_seqlens[b] < q_len: not a valid call, but it must not produce garbage) keeps its running max at the
            // mask value; its row is written as zeros
            const float lv = __shfl(l_run[u], 4 * g + j, 64), mv = __shfl(m_run[u], 4 * g + j, 64);
            li[j] = (lv > 0.0f && mv > -1.0e29f) ? 1.0f / lv : 0.0f;
        }
        #pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const int qr = q0 + 16 * u + 4 * g + j;
            if (qr < q_len)
            {
                half_t* op = a.out + (((size_t) b * q_len + qr) * hq + h) * HD + c;
                #pragma unroll
                for (int nb = 0; nb < HD / 16; ++nb) op[16 * nb] = (half_t) (oc[u][nb][j] * li[j]);
            }
        }
    }
}

// q / out: fp16 [bsz][q_len][heads_q][head_dim]; k_pages / v_pages: fp16 [pages][page_size][heads_kv][head_dim] (what exl3_dequant_cache_paged writes);
// cache_seqlens[b] = tokens of sequence b in the cache INCLUDING the q_len new ones (they were appended before the call, as the reference does);
// causal: query i of the chunk attends to keys 0 .. cache_seqlens[b] - q_len + i.
extern "C" int exl3_attn_prefill_paged(const void* q, void* out, const void* k_pages, const void* v_pages, const int32_t* block_table,
                                       const int32_t* cache_seqlens, int bsz, int q_len, int heads_q, int heads_kv, int head_dim,
                                       int blocks_per_seq, int page_size, float scale, void* stream)
{
    return exl3_attn_prefill_paged_strided(q, (int64_t) heads_q * head_dim, out, k_pages, v_pages, block_table, cache_seqlens, bsz, q_len, heads_q, heads_kv,
                                           head_dim, blocks_per_seq, page_size, scale, stream);
}

// q as a column range of a wider row-major matrix (ldq halves per token): the prefill route's fused q|k|v GEMM output, consumed in place
extern "C" int exl3_attn_prefill_paged_strided(const void* q, int64_t ldq, void* out, const void* k_pages, const void* v_pages, const int32_t* block_table,
                                               const int32_t* cache_seqlens, int bsz, int q_len, int heads_q, int heads_kv, int head_dim,
                                               int blocks_per_seq, int page_size, float scale, void* stream)
{
    EXL3_CHECK_ARG(ldq >= (int64_t) heads_q * head_dim && ldq % 8 == 0, "attn_prefill: bad q token stride");
    EXL3_CHECK_ARG(head_dim == 128 || head_dim == 64, "attn_prefill: head_dim must be 128 or 64");
    EXL3_CHECK_ARG(heads_kv >= 1 && heads_q % heads_kv == 0, "attn_prefill: heads_q must be a multiple of heads_kv");
    EXL3_CHECK_ARG(page_size > 0 && page_size % PA_BN == 0 && blocks_per_seq >= 1, "attn_prefill: page size must be a multiple of 64");
    EXL3_CHECK_ARG(q_len >= 1 && (q_len + PA_BM - 1) / PA_BM <= 65535 && heads_q <= 65535, "attn_prefill: bad q_len / heads");
    EXL3_CHECK_ARG(bsz >= 0 && bsz <= 65535, "attn_prefill: bsz out of range");
    if (bsz == 0) return EXL3_OK;
    EXL3_CHECK_ARG(q && out && k_pages && v_pages && block_table && cache_seqlens, "attn_prefill: null pointer");
    PrefillAttnArgs a;
    a.q = (const half_t*) q; a.out = (half_t*) out; a.k_pages = (const half_t*) k_pages; a.v_pages = (const half_t*) v_pages;
    a.block_table = block_table; a.cache_seqlens = cache_seqlens;
    a.q_len = q_len; a.hq = heads_q; a.hkv = heads_kv; a.blocks_per_seq = blocks_per_seq; a.page_size = page_size; a.scale = scale; a.ldq = ldq;
    const int gq = heads_q / heads_kv;
    int gw = gq % 4 == 0 ? 4 : (gq % 2 == 0 ? 2 : 1);                                // query heads per workgroup (they share a kv head) ...
    const int64_t qtiles = (int64_t) ((q_len + PA_BM - 1) / PA_BM) * bsz;
    while (gw > 1 && qtiles * (heads_q / gw) < 512) gw >>= 1;                         // ... fewer when the launch would not fill the 256 CUs twice
    dim3 grid(heads_q / gw, (q_len + PA_BM - 1) / PA_BM, bsz);
    hipStream_t st = (hipStream_t) stream;
    // 16 queries per wave and 4 waves per SIMD (QG = 1) where the registers allow it, else 32 queries per wave at 2 waves per SIMD
    #define PA_L(HDv, QGv) { if (gw == 4) attn_prefill_kernel<HDv, 4, QGv><<<grid, 1024 / QGv, 0, st>>>(a); else if (gw == 2) attn_prefill_kernel<HDv, 2, QGv><<<grid, 512 / QGv, 0, st>>>(a); \
                             else attn_prefill_kernel<HDv, 1, QGv><<<grid, 256 / QGv, 0, st>>>(a); }
    // 32 queries per wave (every K / V fragment feeds two matrix instructions; 216 VGPRs, 2 waves per SIMD) for the 8-wave workgroups, 16 queries per
    // wave (<= 128 VGPRs, 4 waves per SIMD) for the smaller ones, whose 32-query variant would leave one wave per SIMD
    static const int qg_env = [] { const char* e = getenv("EXL3_HIP_ATTN_QG"); return e ? atoi(e) : 0; }();
    const int qg = qg_env ? qg_env : (gw == 4 ? 2 : 1);
    if (head_dim == 128) { if (qg == 2) PA_L(128, 2) else PA_L(128, 1) } else { if (qg == 2) PA_L(64, 2) else PA_L(64, 1) }
    #undef PA_L
    return exl3_check_launch("attn_prefill");
}
