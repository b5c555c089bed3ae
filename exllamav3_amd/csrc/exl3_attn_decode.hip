// Decode attention straight from the quantized paged KV cache ("quant-cache-direct flash decoding").
//
//   out[b][hq][:] = softmax(q[b][hq] . K[b][:len][kvh]^T * scale) @ V[b][:len][kvh]         one new token per sequence, GQA
//
// reference: libtorch/attention.cpp:246-504 decodes through dequant_cache_paged (cache/q_cache.cu) + an fp16 attention kernel.  Here K and V are
// never materialised: a cached 32-group is x' = H u with H = H32/sqrt(32) orthonormal and symmetric and u the rotated-domain levels*scale, so
//   q . x' = (H q) . u            -> q is rotated once per head, scores are dot products with the raw dequantized levels
//   sum_t p_t x'_t = H (sum_t p_t u_t)   -> the output is accumulated in the rotated domain and rotated back once at the end.
// The only deviation from the reference's arithmetic is that it rounds x' to fp16 before use (2^-11 relative): inside the 1e-2 tolerance.
//
// Mapping: a 32-lane half-wave per cached token (lane = 4 consecutive head dims = one 8-lane quantization subgroup position), the 8 half-waves
// of a workgroup stride over the tokens of one (sequence, kv head, context split); online softmax per half-wave, merged through LDS; context
// splits are merged by a second small kernel (flash decoding).  The work is tiny (144 B per token and kv head at 4 bits): the design goal is
// two short launches, not throughput.
// head_dim 64 (Llama-3.2-1B): a 128-value block of the token vector holds TWO kv heads; the half-wave then carries two independent 16-lane
// attention problems (score reductions over 16 lanes, per-half statistics), everything else is unchanged.
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include "exl3_glue_device.cuh"

#include "exl3_attn_device.cuh"

#define ATT_MAX_GQ 8

struct AttnArgs
{
    const half_t* q; half_t* out;                               // [bsz][hq][128]
    const uint32_t* k_cache; const half_t* k_scales; const uint32_t* v_cache; const half_t* v_scales;
    const int32_t* block_table; const int32_t* cache_seqlens;   // [bsz][blocks_per_seq], [bsz] (length INCLUDING the new token)
    float* part;                                                // [bsz][hq][nsplit][132] partial (m, l, o[128]) when nsplit > 1
    int blocks_per_seq, page_size, k_bits, v_bits, hq, hkv, nsplit, split_tokens;
    float scale;
    int force_part;                                             // write partial records even with one split (the consumer merges: exl3_gemv_ex_attm)
};
// hq / hkv in AttnArgs count HEADS; a workgroup handles one 128-value block of the kv vector = 128 / HD kv heads

// KVB: bits of both caches when known at compile time (4 or 8: one packed field per value, the dequantization folds to a shift and a mask per value
// instead of four predicated field passes), 0 = read a.k_bits / a.v_bits (any 2-8 bit mix)
template <int GQ, int HD, int KVB>
__global__ __launch_bounds__(256)
void attn_decode_kernel(const AttnArgs a)
{
    const int k_bits = KVB ? KVB : a.k_bits, v_bits = KVB ? KVB : a.v_bits;
    constexpr int NSUB = 128 / HD;                              // kv heads per 128-value block (1 or 2)
    constexpr int RW = HD / 4;                                  // lanes per head (32 or 16)
    __shared__ float ml_s[8][GQ][NSUB][2];
    __shared__ float o_s[8][GQ][128];
    const int tid = threadIdx.x, l = tid & 31, hwid = tid >> 5, lane = tid & 63;
    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;          // h: 128-value block of the kv vector
    const int sub = l / RW;                                     // which kv head of the block this lane belongs to
    const int kvh = h * NSUB + sub;
    const int len = a.cache_seqlens[b];
    const int t0 = split * a.split_tokens, t1 = min(len, t0 + a.split_tokens);
    const int G = a.hkv * HD / 32;                              // quantization groups per token
    const int g = l >> 3;

    // rotated, pre-scaled queries of the GQ heads sharing this kv head: qh = H32(q) / sqrt(32) * softmax scale
    float qh[GQ][4];
    #pragma unroll
    for (int i = 0; i < GQ; ++i)
    {
        const half4_t qv = ((const half4_t*) (a.q + ((size_t) b * a.hq + kvh * GQ + i) * HD))[l % RW];
        float v0 = (float) qv.x, v1 = (float) qv.y, v2 = (float) qv.z, v3 = (float) qv.w;
        kvg_had32(v0, v1, v2, v3, lane);
        const float f = ATT_R32 * a.scale;
        qh[i][0] = v0 * f; qh[i][1] = v1 * f; qh[i][2] = v2 * f; qh[i][3] = v3 * f;
    }
    float mx[GQ], ls[GQ], oa[GQ][4];
    #pragma unroll
    for (int i = 0; i < GQ; ++i) { mx[i] = -1.0e30f; ls[i] = 0.0f; oa[i][0] = oa[i][1] = oa[i][2] = oa[i][3] = 0.0f; }

    // the split's page: one block-table read per workgroup when the split lies inside one page (always, for the host's 32 / 64-token splits)
    // (neither depends on the sequence length: the block-table read is issued together with the q and length loads, and tokens beyond the
    // length inside the page are fetched speculatively -- the page is allocated memory -- and masked afterwards)
    const int t1_max = t0 + a.split_tokens;
    const bool one_page = (t0 / a.page_size) == ((t1_max - 1) / a.page_size) && t0 / a.page_size < a.blocks_per_seq;
    const int64_t page_phys = a.block_table[(size_t) b * a.blocks_per_seq + min(t0 / a.page_size, a.blocks_per_seq - 1)];
    const int ntok = max(t1 - t0, 0);
    const int trips = one_page ? a.split_tokens / 8 : (ntok + 7) / 8;   // uniform trip count (the butterflies need whole waves), length-independent when possible
    // a split never straddles a page when split_tokens divides the page size (the host guarantees that or single-page splits are not assumed):
    // the page lookup is still per token, but batches of ATT_UNROLL tokens are fetched and dequantized before any softmax update so that the
    // cache loads of a batch are in flight together (this loop is a chain of ~1 us memory latencies otherwise)
    constexpr int ATT_UNROLL = 4;
    for (int it0 = 0; it0 < trips; it0 += ATT_UNROLL)
    {
        float kv[ATT_UNROLL][8];
        bool actv[ATT_UNROLL];
        #pragma unroll
        for (int u = 0; u < ATT_UNROLL; ++u)
        {
            const int t = t0 + (it0 + u) * 8 + hwid;
            actv[u] = (it0 + u) < trips && t < t1;
            const int tc = (actv[u] || one_page) ? t : max(t1 - 1, 0);
            const int64_t pg = one_page ? page_phys : (int64_t) a.block_table[(size_t) b * a.blocks_per_seq + tc / a.page_size];
            const int64_t token_pos = pg * a.page_size + (tc % a.page_size);
            const int64_t gb = token_pos * G + h * 4 + g;
            kv_dequant_vals_rt(k_bits, a.k_cache + gb * k_bits, a.k_scales + gb, lane, kv[u][0], kv[u][1], kv[u][2], kv[u][3]);
            kv_dequant_vals_rt(v_bits, a.v_cache + gb * v_bits, a.v_scales + gb, lane, kv[u][4], kv[u][5], kv[u][6], kv[u][7]);
        }
        #pragma unroll
        for (int u = 0; u < ATT_UNROLL; ++u)
        {
            #pragma unroll
            for (int i = 0; i < GQ; ++i)
            {
                float s = qh[i][0] * kv[u][0] + qh[i][1] * kv[u][1] + qh[i][2] * kv[u][2] + qh[i][3] * kv[u][3];
                #pragma unroll
                for (int j = 1; j < RW; j <<= 1) s += xor_lane(s, j);     // over the lanes of THIS head
                if (actv[u])
                {
                    const float mn = fmaxf(mx[i], s);
                    const float corr = __expf(mx[i] - mn), p = __expf(s - mn);
                    ls[i] = ls[i] * corr + p;
                    oa[i][0] = oa[i][0] * corr + p * kv[u][4]; oa[i][1] = oa[i][1] * corr + p * kv[u][5];
                    oa[i][2] = oa[i][2] * corr + p * kv[u][6]; oa[i][3] = oa[i][3] * corr + p * kv[u][7];
                    mx[i] = mn;
                }
            }
        }
    }
    // merge the 8 half-waves
    #pragma unroll
    for (int i = 0; i < GQ; ++i)
    {
        if (l % RW == 0) { ml_s[hwid][i][sub][0] = mx[i]; ml_s[hwid][i][sub][1] = ls[i]; }
        *((float4_t*) &o_s[hwid][i][4 * l]) = float4_t{ oa[i][0], oa[i][1], oa[i][2], oa[i][3] };
    }
    __syncthreads();
    for (int i = hwid; i < GQ; i += 8)
    {
        float M = -1.0e30f;
        #pragma unroll
        for (int k = 0; k < 8; ++k) M = fmaxf(M, ml_s[k][i][sub][0]);
        float L = 0.0f; float4_t O = { 0.f, 0.f, 0.f, 0.f };
        #pragma unroll
        for (int k = 0; k < 8; ++k)
        {
            const float e = __expf(ml_s[k][i][sub][0] - M);
            L += ml_s[k][i][sub][1] * e;
            const float4_t ov = *((const float4_t*) &o_s[k][i][4 * l]);
            O.x += ov.x * e; O.y += ov.y * e; O.z += ov.z * e; O.w += ov.w * e;
        }
        const int head = kvh * GQ + i;                          // query head of this lane
        if (a.nsplit == 1 && !a.force_part)
        {
            const float inv = L > 0.0f ? 1.0f / L : 0.0f;
            float v0 = O.x * inv, v1 = O.y * inv, v2 = O.z * inv, v3 = O.w * inv;
            kvg_had32(v0, v1, v2, v3, lane);
            ((half4_t*) (a.out + ((size_t) b * a.hq + head) * HD))[l % RW] = half4_t{ f2h(v0 * ATT_R32), f2h(v1 * ATT_R32), f2h(v2 * ATT_R32), f2h(v3 * ATT_R32) };
        }
        else
        {
            // one record per (sequence, 128-value block, query index i): {m, l} of each kv head of the block, then the 128 accumulators
            float* p = a.part + ((((size_t) b * gridDim.y + h) * GQ + i) * a.nsplit + split) * 132;
            if (l % RW == 0) { p[2 * sub] = M; p[2 * sub + 1] = L; }
            *((float4_t*) (p + 4 + 4 * l)) = O;                 // record = {m0, l0, m1, l1, o[128]}
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// Long contexts: the same computation on the matrix pipe (head_dim 128, 4-bit K and V).  The kernel above spends ~160 VALU instructions per pair
// of cached tokens (per-head butterflies, per-lane softmax updates): 43.6 us per layer at 16 000 tokens.  Here a wave takes 16 tokens per step:
//   * lane (c = lane % 16, kg = lane / 16) loads ONE dword per 32-group = the 8 four-bit levels of dims 8 kg .. 8 kg + 7 of token c and turns them
//     into fp16 (level - 7.5) * scale / 8 with exact half arithmetic: (x >> (4r - 1)) & 0x001E001E | 0x4C004C00 is the pair (16 + n_r / 32,
//     16 + n_{r+4} / 32), minus (16 + 15/64) is exact, times 4 * scale rounds once -- the value the reference's dequantization rounds to fp16 too.
//     The pair order (0, 4, 1, 5, 2, 6, 3, 7) is a permutation of the contraction index, applied to the rotated query fragments as well;
//   * scores = K q^T with v_mfma_f32_16x16x32_f16 (A = the dequantized K rows, B = rotated queries, one instruction per 32-group): lane (head c, kg)
//     ends up with the scores of tokens 4 kg .. 4 kg + 3 for query head c -- which is the A operand layout of the second product, so the
//     probabilities never move between lanes (the S^T trick of exl3_attn_prefill.hip);
//   * V is dequantized the same way, staged row-major in a wave-private LDS tile and gathered as the B operand (4 tokens of one dim) by
//     ds_read_b64_tr_b16; output columns are therefore in the pair order and are un-permuted once, when the partial record is written.
// Partial records and the merge kernel are those of the kernel above.
struct AttnWideArgs { AttnArgs a; };

// The q|k|v launch's epilogue done by the attention kernel itself (FUSED form, round 4): what exl3_glue_qkv_tab does in a launch of its own --
// split-k reduce of the deferred slabs, output Hadamard, row-scale correction, svh, RoPE from the per-step tables, the 4-bit append of the new token's
// K / V -- happens in the preparation phase of every workgroup for the query heads it needs; the workgroup whose context split holds the new token
// (index len - 1) also finishes that token's K and V rows of its kv head, writes them to the cache page AND keeps the words in LDS, from where its
// streaming loop takes them (no read-back of its own stores).  reference: libtorch/attention.cpp:386-440 (rope, cache append, split kernel).
struct AttnQkvArgs
{
    SlabRef sq, sk, sv;                                         // deferred slabs of the q|k|v launch
    const half_t* svh_q; const half_t* svh_k; const half_t* svh_v;
    const float* rope_sin; const float* rope_cos; const int64_t* slots;        // exl3_qkv_prep: [m][64], [m][64], physical cache row of the new token [m]
    GemvRescale rs;
    half_t* q_out;                                              // optional [m][hq][128]: the finished (roped) queries, written by context split 0
    uint32_t* k_cache_w; half_t* k_scales_w; uint32_t* v_cache_w; half_t* v_scales_w;
    int rope_mode, m;
};

// HD64 (head_dim 64, Llama-3.2-1B): the 128-value block of the kv vector holds TWO kv heads.  The kernel runs unchanged on "rows" = the query heads of both
// (GQ = 2 x heads per kv head <= 8): row (sub, qi) carries its 64 query dims at dims 64 sub .. 64 sub + 63 of a 128-wide row and ZEROS in the other half, so
// the four score instructions over the block's four 32-groups give every row the dot product with ITS kv head only; the value product computes 128 dims per
// row of which the row's own half is kept.  Records come out in the NSUB = 2 form of the half-wave kernel ({m0, l0, m1, l1, o[128]} per query index).
// NW = waves per workgroup: 4 (one dependent chain per SIMD: the short contexts, where a wave has one step anyway) or 8 (two chains per SIMD: a step is
// cache words -> dequantize -> 4 score MFMAs -> softmax -> V tile -> 8 transposed reads + MFMAs, ~0.8 us of one wave's latency; at 16 000 tokens the
// one-wave-per-SIMD form ran 8 such steps back to back per wave)
template <int GQ, bool FUSED, bool HD64 = false, int NW = 4>
__global__ __launch_bounds__(64 * NW)
void attn_decode_wide_kernel(const AttnArgs a, const AttnQkvArgs x)
{
    constexpr int HD = 128;
    __shared__ __attribute__((aligned(16))) uint32_t new_kv[2][16];            // FUSED: the new token's K / V words of this kv head (4 groups x 4 words)
    __shared__ __attribute__((aligned(8))) half_t new_sc[2][4];                //        and their group scales
    __shared__ __attribute__((aligned(16))) half_t q_s[8 * 128];               // rotated, pre-scaled queries in pair order (rows >= GQ stay zero)
    __shared__ __attribute__((aligned(16))) half_t vt[NW][16 * AW_VS];          // wave-private dequantized V tiles; after the loop: the waves' partial outputs
    __shared__ float ml_s[NW][8][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, kg = lane >> 4;
    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;              // h: kv head (one 128-value block of the kv vector)
    const int len = a.cache_seqlens[b];
    const int t0 = split * a.split_tokens, t1 = min(len, t0 + a.split_tokens);
    const int G = a.hkv * (HD64 ? 64 : 128) / 32;

    // ---- rotated queries: half-wave i rotates head i as the kernel above does and stores it in pair order; natural-log scores become log2 scores
    // FUSED: the new token (index len - 1) lies in exactly one context split; that workgroup finishes and appends its K / V
    const bool owner = FUSED && len - 1 >= t0 && len - 1 < t0 + a.split_tokens;
    // the wave's FIRST 16 tokens are requested here, ahead of the query preparation (which does not depend on them): at a 1000-token context a wave
    // has exactly one step, and the chain length -> block table -> cache words otherwise starts only after the queries are ready
    // (round 5) cache words as ONE 16-byte load per lane and operand: lane (token c, kg) takes the four words of 32-GROUP kg -- the 8 levels of word s are
    // dims 32 kg + 8 s .. + 7, a permutation of the contraction index that the query fragments and the V tile follow -- instead of word kg of each of the
    // four groups (four 4-byte loads); and the loop below keeps the words of TWO steps ahead in registers (it was one memory round trip per 16-token step:
    // 17 us per layer at a 16 000-token context, 0.14 of the cache words' HBM time).  The page ids of the split (<= 4 pages) are read once, here.
    const int pg_first = t0 / a.page_size;
    int64_t pgs[4];
    #pragma unroll
    for (int i = 0; i < 4; ++i) pgs[i] = a.block_table[(size_t) b * a.blocks_per_seq + min(pg_first + i, a.blocks_per_seq - 1)];
    auto page_of = [&] (const AttnArgs& aa, int tk) -> int64_t
    {
        const int pi = tk / aa.page_size - pg_first;
        return pi == 0 ? pgs[0] : (pi == 1 ? pgs[1] : (pi == 2 ? pgs[2] : (pi == 3 ? pgs[3] : (int64_t) aa.block_table[(size_t) b * aa.blocks_per_seq + tk / aa.page_size])));
    };
    struct StepWords { uint4_t k, v; half_t ks, vs; };
    auto load_step = [&] (const AttnArgs& aa, int st_) -> StepWords
    {
        StepWords r;
        const int tk = max(min(t0 + 16 * NW * st_ + 16 * wave + c, t1 - 1), 0);
        const int64_t gbase = (page_of(aa, tk) * aa.page_size + (tk % aa.page_size)) * G + h * 4 + kg;
        r.k = *((const uint4_t*) (aa.k_cache + gbase * 4)); r.v = *((const uint4_t*) (aa.v_cache + gbase * 4));
        r.ks = aa.k_scales[gbase]; r.vs = aa.v_scales[gbase];
        return r;
    };
    StepWords w0 = load_step(a, 0), w1 = load_step(a, 1);      // (four steps in flight measured the same: 22.2 vs 21.1 us per layer at 16 000 tokens -- the step is a dependent chain per wave, not a memory wait)
    if constexpr (FUSED)
    {
        // tasks, one per half-wave: 0 .. GQ - 1 = query head h * GQ + task; GQ = the new token's K row, GQ + 1 = its V row (kv head h; only in the
        // workgroup whose split holds the new token).  One code path for the three kinds (per-lane operand pointers, like glue_qkv_kernel), a second
        // round only for GQ = 7, 8.  Every half-wave of a participating wave runs the arithmetic (the butterflies want whole half-waves); stores are
        // predicated.  The finished values are those of exl3_glue_qkv_tab bit for bit (qkv_block_finish).
        // HD64: the tasks are 128-value BLOCKS as well -- NT = GQ / 2 query blocks (two adjacent heads each: rows 2 task, 2 task + 1 of the operand),
        // then the K and the V block of the two kv heads; the rope partner distance and the frequency index follow the 64-wide head
        const int l = tid & 31, hw = tid >> 5;
        constexpr int NT = HD64 ? GQ / 2 : GQ;
        constexpr int PH = HD64 ? 8 : 16;
        constexpr int NHW = 2 * NW;                                                           // half-waves = tasks per round
        #pragma nounroll
        for (int r = 0; r < (NT + 2 + NHW - 1) / NHW; ++r)
        {
            if (r > 0 && !owner) break;                                                       // workgroup-uniform
            const int task = r * NHW + hw;
            const int kind = task < NT ? 0 : task - NT + 1;                                    // 0: query, 1: K row, 2: V row, >= 3: nothing
            const int tw = r * NHW + 2 * wave;
            const bool wave_has = tw < NT || (owner && tw + 1 >= NT && tw <= NT + 1);          // wave-uniform: tasks tw, tw + 1
            if (wave_has)
            {
                const bool kvt = kind == 1 || kind == 2;
                const int cblk = kvt ? h : h * NT + min(task, NT - 1);
                const float* sbase = kind == 1 ? x.sk.base : (kind == 2 ? x.sv.base : x.sq.base);
                const half_t* svh = (kind == 1 ? x.svh_k : (kind == 2 ? x.svh_v : x.svh_q)) + cblk * 128;
                const half4_t sc = ((const half4_t*) svh)[l];
                float rs_p = 0.0f, rs_n = 0.0f;
                if (x.rs.ss_new && l < (x.rs.k >> 7)) { rs_p = x.rs.ss_prev[(size_t) b * (x.rs.k >> 7) + l]; rs_n = x.rs.ss_new[(size_t) b * (x.rs.k >> 7) + l]; }
                float4_t sn4 = { 0.f, 0.f, 0.f, 0.f }, cs4 = sn4;
                const int f = x.rope_mode == 2 ? 4 * (l & (PH - 1)) : 2 * (l & (2 * PH - 1));
                if (x.rope_mode == 2) { sn4 = *((const float4_t*) (x.rope_sin + b * 64 + f)); cs4 = *((const float4_t*) (x.rope_cos + b * 64 + f)); }
                else { sn4.x = x.rope_sin[b * 64 + f]; sn4.y = x.rope_sin[b * 64 + f + 1]; cs4.x = x.rope_cos[b * 64 + f]; cs4.y = x.rope_cos[b * 64 + f + 1]; }
                const int64_t token_pos = x.slots[b];
                const SlabRef sr = { sbase, x.sq.S };                                          // one launch wrote the three slab sets: one split count
                const float4_t ysum = slab_sum(sr, cblk, b, x.m, l);
                const half4_t y = qkv_block_finish(ysum, sc, x.rs, b, l, rs_p, rs_n, kind != 2, x.rope_mode, PH, sn4, cs4);
                float v0 = (float) y.x, v1 = (float) y.y, v2 = (float) y.z, v3 = (float) y.w;
                // K / V row: 4-bit append to the cache page and the workgroup's own copy of the words
                const int64_t gb = token_pos * G + h * 4 + (l >> 3);
                uint32_t* cw = kind == 2 ? x.v_cache_w : x.k_cache_w; half_t* cs = kind == 2 ? x.v_scales_w : x.k_scales_w;
                const bool actkv = owner && kvt;
                kv_quant_regs<4>(v0, v1, v2, v3, cw + gb * 4, cs + gb, actkv, lane);
                kv_quant_regs<4>(v0, v1, v2, v3, &new_kv[kind == 2 ? 1 : 0][(l >> 3) * 4], &new_sc[kind == 2 ? 1 : 0][l >> 3], actkv, lane);
                // query: rotated into the cache's H32 domain, pre-scaled, in pair order (the unfused form's staging below)
                if (kind == 0 && x.q_out && split == 0) ((half4_t*) (x.q_out + (size_t) b * a.hq * (HD64 ? 64 : 128) + (size_t) cblk * 128))[l] = y;
                kvg_had32(v0, v1, v2, v3, lane);
                const float fq = ATT_R32 * a.scale * 1.44269504f;
                const float vv[4] = { v0 * fq, v1 * fq, v2 * fq, v3 * fq };
                if (kind == 0)
                {
                    if constexpr (HD64)
                    {
                        // lanes 0-15: head 2 task, lanes 16-31: head 2 task + 1; row i = (sub, qi) keeps its 64 dims in half `sub`, zeros in the other
                        const int i = 2 * task + (l >> 4), sub = i / NT;
                        #pragma unroll
                        for (int e = 0; e < 4; ++e)
                        {
                            const int d = 64 * sub + 4 * (l & 15) + e, d8 = d & 7, dz = d ^ 64;
                            q_s[i * 128 + (d & ~7) + 2 * (d8 & 3) + (d8 >> 2)] = (half_t) vv[e];
                            q_s[i * 128 + (dz & ~7) + 2 * (d8 & 3) + (d8 >> 2)] = (half_t) 0.0f;
                        }
                    }
                    else
                    {
                        #pragma unroll
                        for (int e = 0; e < 4; ++e)
                        {
                            const int d = 4 * l + e, d8 = d & 7;
                            q_s[task * 128 + (d & ~7) + 2 * (d8 & 3) + (d8 >> 2)] = (half_t) vv[e];
                        }
                    }
                }
            }
            if (r == 0 && hw >= GQ && hw < 8)
            {
                #pragma unroll
                for (int e = 0; e < 4; ++e) q_s[hw * 128 + 4 * l + e] = (half_t) 0.0f;          // rows >= GQ of the query operand stay zero
            }
        }
    }
    else if constexpr (HD64)
    {
        // row i = sub * (GQ / 2) + qi = query head (2 h + sub) * (GQ / 2) + qi: lanes 0-15 of the half-wave hold its 64 dims, which go to dims 64 sub ..
        // of the row; lanes 16-31 write the zeros of the other half
        constexpr int G2 = GQ / 2;
        const int l = tid & 31, i = tid >> 5;
        const int sub = i < GQ ? i / G2 : 0, qi = i - sub * G2;          // (rows >= GQ are all zeros: sub 0 keeps their stores inside the row)
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (i < GQ && l < 16)
        {
            const half4_t qv = ((const half4_t*) (a.q + ((size_t) b * a.hq + (2 * h + sub) * G2 + qi) * 64))[l];
            v0 = (float) qv.x; v1 = (float) qv.y; v2 = (float) qv.z; v3 = (float) qv.w;
        }
        kvg_had32(v0, v1, v2, v3, lane);
        const float f = ATT_R32 * a.scale * 1.44269504f;
        const float vv[4] = { v0 * f, v1 * f, v2 * f, v3 * f };
        const int dbase = l < 16 ? 64 * sub + 4 * l : 64 * (1 - sub) + 4 * (l - 16);
        #pragma unroll
        for (int e = 0; e < 4; ++e)
        {
            const int d = dbase + e, d8 = d & 7;
            if (i < 8) q_s[i * 128 + (d & ~7) + 2 * (d8 & 3) + (d8 >> 2)] = (i < GQ && l < 16) ? (half_t) vv[e] : (half_t) 0.0f;        // (NW = 8: half-waves 8 .. 15 have no row)
        }
    }
    else
    {
        const int l = tid & 31, i = tid >> 5;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (i < GQ)
        {
            const half4_t qv = ((const half4_t*) (a.q + ((size_t) b * a.hq + h * GQ + i) * HD))[l];
            v0 = (float) qv.x; v1 = (float) qv.y; v2 = (float) qv.z; v3 = (float) qv.w;
        }
        kvg_had32(v0, v1, v2, v3, lane);
        const float f = ATT_R32 * a.scale * 1.44269504f;
        const float vv[4] = { v0 * f, v1 * f, v2 * f, v3 * f };
        #pragma unroll
        for (int e = 0; e < 4; ++e)
        {
            const int d = 4 * l + e, d8 = d & 7;
            if (i < 8) q_s[i * 128 + (d & ~7) + 2 * (d8 & 3) + (d8 >> 2)] = i < GQ ? (half_t) vv[e] : (half_t) 0.0f;
        }
    }
    // FUSED: the preparation above holds ~35 scalar argument registers (slab bases, scales, tables, rescale sums); with the streaming loop's own
    // arguments (cache pointers, block table, partial records) live across it the kernel ran out of SGPRs (24 spills and a 20-byte private segment =
    // +1.5 .. 3 us per launch).  The loop's arguments are therefore read from the kernel-argument segment AFTER the preparation: the segment pointer
    // goes through an empty asm, so the compiler cannot hoist those loads to the top
    AttnArgs al;
#if defined(__HIP_DEVICE_COMPILE__)                                                            // (the host pass of the compiler has no address space 4)
    if constexpr (FUSED)
    {
        const __attribute__((address_space(4))) char* kp = (const __attribute__((address_space(4))) char*) __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        al = *((const __attribute__((address_space(4))) AttnArgs*) kp);                        // `a` is the first kernel parameter: offset 0
    }
    else
#endif
    al = a;
    __syncthreads();
    half8_t qf[4];
    #pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = c < 8 ? *((const half8_t*) (q_s + c * 128 + 32 * kg + 8 * s)) : half8_t{ 0, 0, 0, 0, 0, 0, 0, 0 };

    float m_run = -1.0e30f, l_run = 0.0f;                       // of query head c (lanes with c >= GQ carry zero queries)
    float4_t oc[8];
    #pragma unroll
    for (int nb = 0; nb < 8; ++nb) oc[nb] = float4_t{ 0.f, 0.f, 0.f, 0.f };
    half_t* vw = vt[wave];

    const int nsteps = (al.split_tokens + 16 * NW - 1) / (16 * NW);
    uint32_t mk_v = 0x001E001Eu;
    asm volatile("" : "+v"(mk_v));
    for (int st = 0; st < nsteps; ++st)
    {
        const int tb = t0 + 16 * NW * st + 16 * wave;           // the wave's 16 tokens of this step
        if (tb >= t1) break;                                    // wave-uniform: nothing left for this wave (its LDS tile is private)
        const int tk = min(tb + c, t1 - 1);
        uint32_t kw[4] = { w0.k.x, w0.k.y, w0.k.z, w0.k.w }, vw4[4] = { w0.v.x, w0.v.y, w0.v.z, w0.v.w };
        half_t ksc = w0.ks, vsc = w0.vs;
        w0 = w1;
        if (st + 2 < nsteps) w1 = load_step(al, st + 2);            // (two steps ahead; clamped addresses past the split's end are never used)
        if constexpr (FUSED)
        {
            // the new token's words come from this workgroup's LDS copy (its cache row is being written by this very launch)
            if (owner && tk == len - 1)
            {
                #pragma unroll
                for (int s = 0; s < 4; ++s) { kw[s] = new_kv[0][kg * 4 + s]; vw4[s] = new_kv[1][kg * 4 + s]; }
                ksc = new_sc[0][kg]; vsc = new_sc[1][kg];
            }
        }
        // ---- scores of the 16 tokens for all heads: D[token][head]; lane (head c, kg) holds tokens 4 kg .. 4 kg + 3
        float4_t sc = { 0.f, 0.f, 0.f, 0.f };
        #pragma unroll
        for (int s = 0; s < 4; ++s)
        {
            const half_t k4 = ksc * (half_t) 4.0f;
            const half8_t ka = aw_dequant8(kw[s], half2_t{ k4, k4 }, mk_v);
            sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qf[s], sc, 0, 0, 0);
        }
        // ---- V tile of the wave: token c, dims 32 s + 8 kg .. (pair order) -> LDS row c
        #pragma unroll
        for (int s = 0; s < 4; ++s)
        {
            const half_t v4 = vsc * (half_t) 4.0f;
            *((half8_t*) (vw + c * AW_VS + 32 * kg + 8 * s)) = aw_dequant8(vw4[s], half2_t{ v4, v4 }, mk_v);
        }
        // ---- online softmax of head c over the step's tokens (log2 domain)
        float mx = m_run;
        #pragma unroll
        for (int r = 0; r < 4; ++r) { if (tb + 4 * kg + r >= t1) sc[r] = -1.0e30f; mx = fmaxf(mx, sc[r]); }
        mx = fmaxf(mx, xor_lane(mx, 16)); mx = fmaxf(mx, xor_lane(mx, 32));
        // lazy reference: the running reference only moves when the maximum grew by more than 2^8 -- probabilities then stay <= 256 (exact enough in fp16: the
        // relative precision does not depend on the magnitude; sums and accumulators are fp32) and the accumulator rescale below (36 accumulator reads + 16 packed
        // multiplies + 4 cross-lane reads per step) runs in the first step and almost never again; the merge only needs (reference, sum) to be consistent
        if (!(mx > m_run + 8.0f)) mx = m_run;
        const float corr = __builtin_amdgcn_exp2f(m_run - mx);
        float p[4], ps = 0.0f;
        #pragma unroll
        for (int r = 0; r < 4; ++r) { p[r] = sc[r] > -1.0e29f ? __builtin_amdgcn_exp2f(sc[r] - mx) : 0.0f; ps += p[r]; }
        ps += xor_lane(ps, 16); ps += xor_lane(ps, 32);
        l_run = l_run * corr + ps; m_run = mx;
        const half4_t pa = { (half_t) p[0], (half_t) p[1], (half_t) p[2], (half_t) p[3] };
        // ---- rescale the accumulators (rows = heads 4 kg + r live in lanes (dim, kg)) where a running max moved, then add P V
        if (__any(corr != 1.0f))
        {
            float cr[4];
            #pragma unroll
            for (int r = 0; r < 4; ++r) cr[r] = __shfl(corr, 4 * kg + r, 64);
            #pragma unroll
            for (int nb = 0; nb < 8; ++nb) { oc[nb].x *= cr[0]; oc[nb].y *= cr[1]; oc[nb].z *= cr[2]; oc[nb].w *= cr[3]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        #pragma unroll
        for (int nb = 0; nb < 8; ++nb)
        {
            const half4_t vb = aw_tr16(vw + (4 * kg + (c >> 2)) * AW_VS + 16 * nb + 4 * (c & 3));
            oc[nb] = __builtin_amdgcn_mfma_f32_16x16x16f16(pa, vb, oc[nb], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- merge the NW waves: statistics of head c from lanes (c, kg = 0); outputs of heads 4 kg + r, column 16 nb + c (pair order) from every lane
    __syncthreads();                                            // the V tiles are dead: their space takes the partial outputs [wave][head][128] fp32
    float* o_s = (float*) &vt[0][0];                            // NW * 8 * 128 * 4 B = 16 KB (NW = 4) <= sizeof(vt) = 17 KB
    static_assert(NW * 8 * 128 * 4 <= (int) sizeof(vt), "partial outputs must fit the V tiles");
    if (kg == 0 && c < 8) { ml_s[wave][c][0] = m_run; ml_s[wave][c][1] = l_run; }
    #pragma unroll
    for (int nb = 0; nb < 8; ++nb)
        #pragma unroll
        for (int r = 0; r < 4; ++r)
        {
            const int head = 4 * kg + r;
            if (head < 8) o_s[(wave * 8 + head) * 128 + 16 * nb + c] = oc[nb][r];
        }
    __syncthreads();
    // half-wave i finishes head i: lane l owns natural dims 4 l .. 4 l + 3
    {
        const int l = tid & 31, i = tid >> 5;
        if (i < GQ)
        {
            float M = -1.0e30f;
            #pragma unroll
            for (int w = 0; w < NW; ++w) M = fmaxf(M, ml_s[w][i][0]);
            float L = 0.0f, O[4] = { 0.f, 0.f, 0.f, 0.f };
            #pragma unroll
            for (int w = 0; w < NW; ++w)
            {
                const float e = ml_s[w][i][0] > -1.0e29f ? __builtin_amdgcn_exp2f(ml_s[w][i][0] - M) : 0.0f;
                L += ml_s[w][i][1] * e;
                #pragma unroll
                for (int e4 = 0; e4 < 4; ++e4)
                {
                    const int d = 4 * l + e4, d8 = d & 7;
                    O[e4] += o_s[(w * 8 + i) * 128 + (d & ~7) + 2 * (d8 & 3) + (d8 >> 2)] * e;
                }
            }
            if constexpr (HD64)
            {
                // row i = (sub, qi) -> record qi of the block, statistics pair `sub`, its own 64 of the 128 accumulators (lanes of that half)
                constexpr int G2 = GQ / 2;
                const int sub = i / G2, qi = i - sub * G2;
                float* pr = al.part + ((((size_t) b * gridDim.y + h) * G2 + qi) * al.nsplit + split) * 132;
                if (l == 16 * sub) { pr[2 * sub] = M * 0.69314718f; pr[2 * sub + 1] = L; }
                if ((l >> 4) == sub) *((float4_t*) (pr + 4 + 4 * l)) = float4_t{ O[0], O[1], O[2], O[3] };
            }
            else
            {
                float* pr = al.part + ((((size_t) b * gridDim.y + h) * GQ + i) * al.nsplit + split) * 132;
                if (l == 0) { pr[0] = M * 0.69314718f; pr[1] = L; }              // the merge kernel works in natural-log units
                *((float4_t*) (pr + 4 + 4 * l)) = float4_t{ O[0], O[1], O[2], O[3] };
            }
        }
    }
}

// merge of the context splits.  Latency, not work: 32 items x 32 splits x 528 B at a 1000-token context.  The first version gave one half-wave
// per item three dependent load rounds (maxima, then (max, sum) again, then the split outputs 8 at a time): 6.6 us under rocprofv3.  Now FOUR
// half-waves share an item, each takes a quarter of the splits, and everything a half-wave needs -- the (max, sum) statistics of ALL splits
// (lane s holds split s's pair, up to 8 chunks of rw splits in registers) and its own first 8 split outputs -- is requested in ONE round; the
// statistics reductions are DPP butterflies, the four partial outputs are summed in a fixed order through LDS, and the item -> head index
// arithmetic uses multiply-highs.
#define ATT_MERGE_HELPERS 4
template <int HD>
__global__ __launch_bounds__(256)
void attn_merge_kernel(const float* __restrict__ part, half_t* __restrict__ out, int items_total, int nsplit, int gq, int blocks, int hq,
                       uint32_t magic_gq, uint32_t magic_bg, uint32_t magic_blocks, const float* __restrict__ sinks)
{
    constexpr int hd = HD;
    __shared__ float comb[2][ATT_MERGE_HELPERS][32][4];
    // item = (sequence b, 128-value block h, query index i); at head_dim 64 lanes 0-15 / 16-31 belong to the block's two kv heads
    const int tid = threadIdx.x, l = tid & 31, lane = tid & 63, hwid = tid >> 5;
    const int item_local = hwid >> 2, helper = hwid & 3;
    const int item = blockIdx.x * 2 + item_local;
    const bool act = item < items_total;
    constexpr int rw_shift = hd == 128 ? 5 : 4, rw = 1 << rw_shift, nsub = 128 >> (rw_shift + 2);     // compile-time: the butterflies below unroll into DPP ops
    const int sub = l >> rw_shift;
    const int lr = l & (rw - 1), lbase = lane - lr;
    const float* p = part + (size_t) (act ? item : 0) * nsplit * 132;
    const int per = (nsplit + ATT_MERGE_HELPERS - 1) / ATT_MERGE_HELPERS;          // splits per helper
    const int sb = helper * per, cnt_h = max(0, min(per, nsplit - sb));
    const int nchunk = (nsplit + rw - 1) >> rw_shift;                              // <= 8 (host: nsplit <= 128 at rw = 16, <= 256 at 32)
    // ---- one round of loads: all statistics + this helper's first 8 outputs
    float2 st[8];
    #pragma unroll
    for (int c = 0; c < 8; ++c)
    {
        const int sidx = (c << rw_shift) + lr;
        st[c] = (c < nchunk && sidx < nsplit) ? *((const float2*) (p + (size_t) sidx * 132 + 2 * sub)) : float2{ -1.0e30f, 0.0f };
    }
    float4_t ov[8];
    #pragma unroll
    for (int u = 0; u < 8; ++u) ov[u] = *((const float4_t*) (p + (size_t) min(sb + min(u, max(cnt_h - 1, 0)), nsplit - 1) * 132 + 4 + 4 * l));
    // ---- statistics: M = max over all splits, e_s = exp(m_s - M) (kept in the lane that holds split s), L = sum l_s e_s
    float M = -1.0e30f;
    {
        float m = st[0].x;                                                           // lane-wise max over the chunks first, one butterfly
        #pragma unroll
        for (int c = 1; c < 8; ++c) m = fmaxf(m, st[c].x);
        #pragma unroll
        for (int j = 1; j < rw; j <<= 1) m = fmaxf(m, xor_lane(m, j));
        M = m;
    }
    // learned per-head sink logit (gpt-oss style, modules/attention_fn/triton_paged.py:1030-1050): it joins the softmax DENOMINATOR at the final
    // reduction -- the running maximum and the sum of exponentials -- and contributes no value
    float snk = 0.0f;
    if (sinks)
    {
        const int it_ = act ? item : 0;
        const int ig_ = gemv_udiv(it_, magic_gq), i_ = it_ - ig_ * gq, h_ = ig_ - gemv_udiv(ig_, magic_blocks) * blocks;
        snk = sinks[(h_ * nsub + sub) * gq + i_];
        M = fmaxf(M, snk);
    }
    float ec[8], L = 0.0f;
    #pragma unroll
    for (int c = 0; c < 8; ++c)
    {
        ec[c] = st[c].x > -1.0e29f ? __expf(st[c].x - M) : 0.0f;
        if (c < nchunk)                                                              // uniform; chunk sums in chunk order (the order of the first version)
        {
            float ls = st[c].y * ec[c];
            #pragma unroll
            for (int j = 1; j < rw; j <<= 1) ls += xor_lane(ls, j);
            L += ls;
        }
    }
    if (sinks) L += __expf(snk - M);
    // ---- this helper's weighted partial output
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
    for (int c0 = 0; c0 < cnt_h; c0 += 8)
    {
        if (c0 > 0)
        {
            #pragma unroll
            for (int u = 0; u < 8; ++u) ov[u] = *((const float4_t*) (p + (size_t) (sb + min(c0 + u, cnt_h - 1)) * 132 + 4 + 4 * l));
        }
        #pragma unroll
        for (int u = 0; u < 8; ++u)
        {
            const int sg = sb + min(c0 + u, cnt_h - 1);                              // global split index (uniform per half-wave)
            const int ch = sg >> rw_shift;
            float esel = ec[0];
            #pragma unroll
            for (int c = 1; c < 8; ++c) esel = ch == c ? ec[c] : esel;
            const float ev = __shfl(esel, lbase + (sg & (rw - 1)), 64);              // split sg's weight lives in lane sg % rw of this head's lanes
            if (c0 + u < cnt_h) { o0 += ov[u].x * ev; o1 += ov[u].y * ev; o2 += ov[u].z * ev; o3 += ov[u].w * ev; }
        }
    }
    comb[item_local][helper][l][0] = o0; comb[item_local][helper][l][1] = o1; comb[item_local][helper][l][2] = o2; comb[item_local][helper][l][3] = o3;
    __syncthreads();
    if (helper != 0) return;
    #pragma unroll
    for (int h2 = 1; h2 < ATT_MERGE_HELPERS; ++h2)
    {
        o0 += comb[item_local][h2][l][0]; o1 += comb[item_local][h2][l][1]; o2 += comb[item_local][h2][l][2]; o3 += comb[item_local][h2][l][3];
    }
    const float inv = L > 0.0f ? 1.0f / L : 0.0f;
    float v0 = o0 * inv, v1 = o1 * inv, v2 = o2 * inv, v3 = o3 * inv;
    kvg_had32(v0, v1, v2, v3, lane);
    // item -> query head: b = item / (blocks * gq), h = (item / gq) % blocks, i = item % gq; head = (h * nsub + sub) * gq + i
    const int it = act ? item : 0;
    const int b = gemv_udiv(it, magic_bg), ig = gemv_udiv(it, magic_gq), i = it - ig * gq, h = ig - gemv_udiv(ig, magic_blocks) * blocks;
    const int head = (h * nsub + sub) * gq + i;
    if (act) ((half4_t*) (out + ((size_t) b * hq + head) * hd))[lr] = half4_t{ f2h(v0 * ATT_R32), f2h(v1 * ATT_R32), f2h(v2 * ATT_R32), f2h(v3 * ATT_R32) };
}

// what the fused form needs beyond AttnQkvArgs when it has to fall back to the two-launch form (exl3_glue_qkv_tab, then the split kernels)
struct QkvFuse { AttnQkvArgs x; int S; const float* inv_freq; const int32_t* positions; float attn_factor; int hidden; int* fused_out; };
extern "C" int exl3_glue_qkv_tab(const float* sq, const float* sk, const float* sv, int S, const void* svh_q, const void* svh_k, const void* svh_v,
                                 void* q_out, void* k_out, void* v_out, const float* inv_freq, const int32_t* positions,
                                 void* k_cache, void* k_scales, void* v_cache, void* v_scales, const int32_t* block_table, int blocks_per_seq,
                                 int page_size, int k_bits, int v_bits, int m, int heads_q, int heads_kv, int head_dim, int rope_mode,
                                 float attn_factor, const float* ss_prev, const float* ss_new, int hidden, float eps,
                                 const float* rope_sin, const float* rope_cos, const int64_t* slots, void* stream);

// waves per workgroup of the matrix-pipe kernel: 0 = by the split length (8 from two 64-token steps on), 4 / 8 forced (A/B runs, tests)
static int g_attn_wide_waves = 0;
extern "C" int exl3_set_attn_wide_waves(int v) { g_attn_wide_waves = (v == 4 || v == 8) ? v : 0; return EXL3_OK; }

static int attn_decode_impl(const void* q, void* out, const void* k_cache, const void* k_scales, const void* v_cache, const void* v_scales,
                            const int32_t* block_table, const int32_t* cache_seqlens, int bsz, int blocks_per_seq, int page_size,
                            int k_bits, int v_bits, int heads_q, int heads_kv, int head_dim, int max_len, float scale,
                            float* workspace, int64_t workspace_floats, void* stream, int* nsplit_out, const QkvFuse* fuse = nullptr, const float* sinks = nullptr);

extern "C" int exl3_attn_decode_qcache(const void* q, void* out, const void* k_cache, const void* k_scales, const void* v_cache, const void* v_scales,
                                       const int32_t* block_table, const int32_t* cache_seqlens, int bsz, int blocks_per_seq, int page_size,
                                       int k_bits, int v_bits, int heads_q, int heads_kv, int head_dim, int max_len, float scale,
                                       float* workspace, int64_t workspace_floats, void* stream)
{
    return attn_decode_impl(q, out, k_cache, k_scales, v_cache, v_scales, block_table, cache_seqlens, bsz, blocks_per_seq, page_size, k_bits, v_bits, heads_q, heads_kv,
                            head_dim, max_len, scale, workspace, workspace_floats, stream, nullptr);
}

// exl3_attn_decode_qcache with learned attention sinks: sinks[heads_q] fp32, one logit per query head in the units of the scaled scores; it joins the
// softmax denominator (running maximum and sum of exponentials) and carries no value (modules/attention_fn/triton_paged.py:1030-1050, the combine kernel
// of libtorch/attention.cpp:463-480 with HAS_SINKS).  Partial records are always written and the merge kernel finishes every head (workspace required).
extern "C" int exl3_attn_decode_qcache_sinks(const void* q, void* out, const void* k_cache, const void* k_scales, const void* v_cache, const void* v_scales,
                                             const int32_t* block_table, const int32_t* cache_seqlens, int bsz, int blocks_per_seq, int page_size,
                                             int k_bits, int v_bits, int heads_q, int heads_kv, int head_dim, int max_len, float scale,
                                             float* workspace, int64_t workspace_floats, const float* sinks, void* stream)
{
    EXL3_CHECK_ARG(sinks && workspace, "attn_decode_sinks: needs the sink logits and the workspace");
    return attn_decode_impl(q, out, k_cache, k_scales, v_cache, v_scales, block_table, cache_seqlens, bsz, blocks_per_seq, page_size, k_bits, v_bits, heads_q, heads_kv,
                            head_dim, max_len, scale, workspace, workspace_floats, stream, nullptr, nullptr, sinks);
}

// The context-split half of exl3_attn_decode_qcache only: the partial records {m, l, -, -, o[128]} per (sequence, kv block, query index, split) stay in
// `workspace` ([bsz][blocks][gq][*nsplit_out][132] fp32, always written, also with one split) and whoever consumes the attention output merges them --
// exl3_gemv_ex_attm does it inside o_proj's launch.  head_dim 128 only (one query head = one Hadamard block of o_proj's input).
extern "C" int exl3_attn_decode_qcache_split(const void* q, const void* k_cache, const void* k_scales, const void* v_cache, const void* v_scales,
                                             const int32_t* block_table, const int32_t* cache_seqlens, int bsz, int blocks_per_seq, int page_size,
                                             int k_bits, int v_bits, int heads_q, int heads_kv, int head_dim, int max_len, float scale,
                                             float* workspace, int64_t workspace_floats, int* nsplit_out, void* stream)
{
    EXL3_CHECK_ARG(nsplit_out && workspace && (head_dim == 128 || head_dim == 64), "attn_decode_split: needs the workspace, nsplit_out and head_dim 128 or 64");
    return attn_decode_impl(q, nullptr, k_cache, k_scales, v_cache, v_scales, block_table, cache_seqlens, bsz, blocks_per_seq, page_size, k_bits, v_bits, heads_q, heads_kv,
                            head_dim, max_len, scale, workspace, workspace_floats, stream, nsplit_out);
}

// exl3_glue_qkv_tab + exl3_attn_decode_qcache_split as ONE launch where the matrix-pipe split kernel applies (head_dim 128, 4-bit K and V, a length
// bound of two or more 64-token splits): every workgroup finishes the query heads it needs from the q|k|v launch's deferred slabs, the workgroup whose
// split holds the new token appends that token's K / V (libtorch/attention.cpp:386-440 as one node).  Same values as the two launches, bit for bit
// (shared device functions).  q_out [bsz][heads_q][128] is required: scratch of the two-launch fallback, and written by split 0 of the fused form.
// *fused_out (optional) reports which form ran.  cache_seqlens INCLUDE the new token; slots[b] = its physical cache row (exl3_qkv_prep).
extern "C" int exl3_attn_decode_qcache_split_qkv(const float* sq, const float* sk, const float* sv, int S, const void* svh_q, const void* svh_k, const void* svh_v,
                                                 void* q_out, const float* inv_freq, const int32_t* positions, float attn_factor, int rope_mode,
                                                 const float* ss_prev, const float* ss_new, int hidden, float eps,
                                                 const float* rope_sin, const float* rope_cos, const int64_t* slots,
                                                 void* k_cache, void* k_scales, void* v_cache, void* v_scales,
                                                 const int32_t* block_table, const int32_t* cache_seqlens, int bsz, int blocks_per_seq, int page_size,
                                                 int k_bits, int v_bits, int heads_q, int heads_kv, int head_dim, int max_len, float scale,
                                                 float* workspace, int64_t workspace_floats, int* nsplit_out, int* fused_out, void* stream)
{
    EXL3_CHECK_ARG(nsplit_out && workspace && (head_dim == 128 || head_dim == 64), "attn_decode_split_qkv: needs the workspace, nsplit_out and head_dim 128 or 64");
    EXL3_CHECK_ARG(sq && sk && sv && svh_q && svh_k && svh_v && q_out && inv_freq && positions, "attn_decode_split_qkv: null pointer");
    EXL3_CHECK_ARG(rope_sin && rope_cos && slots, "attn_decode_split_qkv: needs the per-step tables of exl3_qkv_prep");
    EXL3_CHECK_ARG(rope_mode == 1 || rope_mode == 2, "attn_decode_split_qkv: rope_mode must be 1 (GPTJ) or 2 (NEOX)");
    EXL3_CHECK_ARG(bsz >= 1 && bsz <= 16 && S >= 1, "attn_decode_split_qkv: 1 <= bsz <= 16");
    EXL3_CHECK_ARG(!ss_new || (ss_prev && hidden > 0 && hidden % 128 == 0), "attn_decode_split_qkv: rescale needs ss_prev and hidden");
    QkvFuse f;
    f.x.sq = { sq, S }; f.x.sk = { sk, S }; f.x.sv = { sv, S };
    f.x.svh_q = (const half_t*) svh_q; f.x.svh_k = (const half_t*) svh_k; f.x.svh_v = (const half_t*) svh_v;
    f.x.rope_sin = rope_sin; f.x.rope_cos = rope_cos; f.x.slots = slots;
    f.x.rs = GemvRescale{ ss_prev, ss_new, hidden, eps };
    f.x.q_out = (half_t*) q_out;
    f.x.k_cache_w = (uint32_t*) k_cache; f.x.k_scales_w = (half_t*) k_scales; f.x.v_cache_w = (uint32_t*) v_cache; f.x.v_scales_w = (half_t*) v_scales;
    f.x.rope_mode = rope_mode; f.x.m = bsz;
    f.S = S; f.inv_freq = inv_freq; f.positions = positions; f.attn_factor = attn_factor; f.hidden = hidden; f.fused_out = fused_out;
    return attn_decode_impl(q_out, nullptr, k_cache, k_scales, v_cache, v_scales, block_table, cache_seqlens, bsz, blocks_per_seq, page_size, k_bits, v_bits, heads_q, heads_kv,
                            head_dim, max_len, scale, workspace, workspace_floats, stream, nsplit_out, &f);
}

static int attn_decode_impl(const void* q, void* out, const void* k_cache, const void* k_scales, const void* v_cache, const void* v_scales,
                            const int32_t* block_table, const int32_t* cache_seqlens, int bsz, int blocks_per_seq, int page_size,
                            int k_bits, int v_bits, int heads_q, int heads_kv, int head_dim, int max_len, float scale,
                            float* workspace, int64_t workspace_floats, void* stream, int* nsplit_out, const QkvFuse* fuse, const float* sinks)
{
    const bool split_only = nsplit_out != nullptr;
    const AttnQkvArgs* xq = nullptr;
    EXL3_CHECK_ARG(q && (out || split_only) && k_cache && k_scales && v_cache && v_scales && block_table && cache_seqlens, "attn_decode: null pointer");
    EXL3_CHECK_ARG(head_dim == 128 || head_dim == 64, "attn_decode: head_dim must be 128 or 64");
    EXL3_CHECK_ARG((heads_kv * head_dim) % 128 == 0, "attn_decode: heads_kv * head_dim must be a multiple of 128 (whole Hadamard blocks)");
    EXL3_CHECK_ARG(heads_kv >= 1 && heads_q % heads_kv == 0 && heads_q / heads_kv <= ATT_MAX_GQ, "attn_decode: heads_q must be a multiple (<= 8x) of heads_kv");
    EXL3_CHECK_ARG(k_bits >= 2 && k_bits <= 8 && v_bits >= 2 && v_bits <= 8, "attn_decode: cache bits must be in [2, 8]");
    EXL3_CHECK_ARG(page_size > 0 && max_len >= 1, "attn_decode: bad page size / length bound");
    if (bsz == 0) { if (split_only) *nsplit_out = 1; return EXL3_OK; }
    // context splits: enough workgroups to cover the chip, at least 64 tokens (8 per half-wave) each
    const int blocks = heads_kv * head_dim / 128;                                    // 128-value blocks of the kv vector = workgroups per (split, sequence)
    const int gq = heads_q / heads_kv;
    int split_tokens = ((max_len + 31) / 32) * bsz * blocks <= 1024 ? 32 : 64;      // 4 or 8 tokens per half-wave
    int nsplit = (max_len + split_tokens - 1) / split_tokens;
    int cap = 1024 / (bsz * blocks) > 1 ? 1024 / (bsz * blocks) : 1;
    if (cap > (head_dim == 128 ? 256 : 128)) cap = head_dim == 128 ? 256 : 128;       // the merge kernel keeps 8 chunks of 32 / 16 split statistics in registers
    if (nsplit > cap) { nsplit = cap; split_tokens = ((max_len + nsplit - 1) / nsplit + 7) / 8 * 8; nsplit = (max_len + split_tokens - 1) / split_tokens; }
    // (split-only callers hand the records to a consumer that keeps one chunk of 32 split statistics: cap the split count there)
    const int cap_so = head_dim == 128 ? 32 : 16;                                   // (head_dim 64: two heads per half-wave, 16 statistics lanes each)
    if (split_only && nsplit > cap_so) { nsplit = cap_so; split_tokens = ((max_len + nsplit - 1) / nsplit + 7) / 8 * 8; nsplit = (max_len + split_tokens - 1) / split_tokens; }
    EXL3_CHECK_ARG((nsplit == 1 && !split_only && !sinks) || (workspace && workspace_floats >= (int64_t) bsz * blocks * gq * nsplit * 132), "attn_decode: workspace too small for the context splits");
    EXL3_CHECK_ARG(!sinks || !split_only, "attn_decode: attention sinks are merged by the merge kernel (not by the split-only forms)");
    AttnArgs a;
    a.q = (const half_t*) q; a.out = (half_t*) out;
    a.k_cache = (const uint32_t*) k_cache; a.k_scales = (const half_t*) k_scales; a.v_cache = (const uint32_t*) v_cache; a.v_scales = (const half_t*) v_scales;
    a.block_table = block_table; a.cache_seqlens = cache_seqlens; a.part = workspace;
    a.blocks_per_seq = blocks_per_seq; a.page_size = page_size; a.k_bits = k_bits; a.v_bits = v_bits; a.hq = heads_q; a.hkv = heads_kv;
    a.nsplit = nsplit; a.split_tokens = split_tokens; a.scale = scale; a.force_part = (split_only || sinks) ? 1 : 0;      // sinks: the merge kernel finishes every head
    hipStream_t st = (hipStream_t) stream;
    // head_dim 128, 4-bit K and V, a length bound of at least two 64-token steps: the matrix-pipe kernel (it always writes partial records); measured
    // ahead of the half-wave-per-token kernel from a 512-token bound on (463 vs 458 tok/s with attention), far ahead at long contexts
    static const int wide_min = [] { const char* e = getenv("EXL3_HIP_ATTN_WIDE_MIN"); return e ? atoi(e) : 128; }();
    // head_dim 64: the same kernel on the two kv heads of a 128-value block (rows = 2 gq <= 8 query heads)
    static const int wide64 = [] { const char* e = getenv("EXL3_HIP_ATTN_WIDE_HD64"); return e ? atoi(e) : 1; }();
    const bool wide_hd64 = head_dim == 64 && gq <= 4 && wide64;
    if ((head_dim == 128 || wide_hd64) && k_bits == 4 && v_bits == 4 && max_len >= wide_min && page_size % 16 == 0 && workspace)
    {
        int st_tok = 64;
        int ns = (max_len + st_tok - 1) / st_tok;
        const int capw = split_only ? (head_dim == 128 ? 32 : 16) : (head_dim == 128 ? 256 : 128);
        // about two workgroups per CU: more, shorter splits cost more in the merge than they return (16 000 tokens: 380 tok/s at 512, 337 at 2048)
        static const int wg_cap = [] { const char* e = getenv("EXL3_HIP_ATTN_WIDE_WGS"); return e ? atoi(e) : 512; }();
        // (ns bottoms out at 1: with bsz * blocks > wg_cap alone the bound cannot be met and the loop must stop there -- the ns >= 2 test
        // below then hands such batches to the half-wave-per-token kernel)
        while (ns > 1 && (ns > capw || (int64_t) ns * bsz * blocks > wg_cap)) { st_tok += 64; ns = (max_len + st_tok - 1) / st_tok; }
        if (ns >= 2 && workspace_floats >= (int64_t) bsz * blocks * gq * ns * 132)
        {
            a.nsplit = ns; a.split_tokens = st_tok;
            if (fuse) { xq = &fuse->x; if (fuse->fused_out) *fuse->fused_out = 1; }
            dim3 gridw(ns, blocks, bsz);
            // eight waves per workgroup (two dependent chains per SIMD) once a 4-wave workgroup would run two or more steps per wave
            static const int nw_env = [] { const char* e = getenv("EXL3_HIP_ATTN_WIDE_NW"); return e ? atoi(e) : 0; }();
            const int nw_req = g_attn_wide_waves ? g_attn_wide_waves : nw_env;
            const bool nw8 = nw_req ? nw_req == 8 : st_tok >= 128;
            // (sixteen waves -- four chains per SIMD, a 64 KB merge through LDS -- measured like four: 443 vs 443 vs 457 tok/s with eight at 16 000 tokens; not instantiated)
            #define AW_LAUNCH(GQv, HD64v) \
                { if (nw8) { if (xq) attn_decode_wide_kernel<GQv, true, HD64v, 8><<<gridw, 512, 0, st>>>(a, *xq); else attn_decode_wide_kernel<GQv, false, HD64v, 8><<<gridw, 512, 0, st>>>(a, AttnQkvArgs{}); } \
                  else     { if (xq) attn_decode_wide_kernel<GQv, true, HD64v, 4><<<gridw, 256, 0, st>>>(a, *xq); else attn_decode_wide_kernel<GQv, false, HD64v, 4><<<gridw, 256, 0, st>>>(a, AttnQkvArgs{}); } }
            if (wide_hd64)
            {
                switch (gq)
                {
                    case 1: AW_LAUNCH(2, true) break;
                    case 2: AW_LAUNCH(4, true) break;
                    case 3: AW_LAUNCH(6, true) break;
                    default: AW_LAUNCH(8, true) break;
                }
            }
            else
            switch (gq)
            {
                case 1: AW_LAUNCH(1, false) break;
                case 2: AW_LAUNCH(2, false) break;
                case 3: AW_LAUNCH(3, false) break;
                case 4: AW_LAUNCH(4, false) break;
                case 5: AW_LAUNCH(5, false) break;
                case 6: AW_LAUNCH(6, false) break;
                case 7: AW_LAUNCH(7, false) break;
                default: AW_LAUNCH(8, false) break;
            }
            #undef AW_LAUNCH
            int rcw = exl3_check_launch("attn_decode_wide");
            if (rcw) return rcw;
            if (split_only) { *nsplit_out = ns; return EXL3_OK; }
            const int items = bsz * blocks * gq;
            const uint32_t mg = gemv_magic((uint32_t) gq), mbg = gemv_magic((uint32_t) (blocks * gq)), mb = gemv_magic((uint32_t) blocks);
            if (head_dim == 128) attn_merge_kernel<128><<<(items + 1) / 2, 256, 0, st>>>(workspace, (half_t*) out, items, ns, gq, blocks, heads_q, mg, mbg, mb, sinks);
            else                 attn_merge_kernel<64><<<(items + 1) / 2, 256, 0, st>>>(workspace, (half_t*) out, items, ns, gq, blocks, heads_q, mg, mbg, mb, sinks);
            return exl3_check_launch("attn_merge");
        }
    }
    if (fuse)
    {
        // the fused form does not apply (short length bound, other cache widths, too many sequences for the split budget): the q|k|v epilogue as its
        // own launch, then the kernels below on its q
        const AttnQkvArgs& x = fuse->x;
        if (fuse->fused_out) *fuse->fused_out = 0;
        int rcg = exl3_glue_qkv_tab(x.sq.base, x.sk.base, x.sv.base, fuse->S, x.svh_q, x.svh_k, x.svh_v, x.q_out, nullptr, nullptr, fuse->inv_freq, fuse->positions,
                                    x.k_cache_w, x.k_scales_w, x.v_cache_w, x.v_scales_w, block_table, blocks_per_seq, page_size, k_bits, v_bits, bsz, heads_q, heads_kv,
                                    head_dim, x.rope_mode, fuse->attn_factor, x.rs.ss_prev, x.rs.ss_new, fuse->hidden, x.rs.eps, x.rope_sin, x.rope_cos, x.slots, stream);
        if (rcg) return rcg;
    }
    dim3 grid(nsplit, blocks, bsz);
    const int kvb = (k_bits == v_bits && (k_bits == 4 || k_bits == 8)) ? k_bits : 0;
    #define ATT_K(GQv, HDv) { if (kvb == 4) attn_decode_kernel<GQv, HDv, 4><<<grid, 256, 0, st>>>(a); else if (kvb == 8) attn_decode_kernel<GQv, HDv, 8><<<grid, 256, 0, st>>>(a); \
                              else attn_decode_kernel<GQv, HDv, 0><<<grid, 256, 0, st>>>(a); }
    #define ATT_L(GQv) case GQv: if (head_dim == 128) ATT_K(GQv, 128) else ATT_K(GQv, 64) break;
    switch (gq) { ATT_L(1) ATT_L(2) ATT_L(3) ATT_L(4) ATT_L(5) ATT_L(6) ATT_L(7) default: if (head_dim == 128) ATT_K(8, 128) else ATT_K(8, 64) break; }
    #undef ATT_K
    #undef ATT_L
    int rc = exl3_check_launch("attn_decode");
    if (rc) return rc;
    if (split_only) { *nsplit_out = nsplit; return EXL3_OK; }
    if (nsplit > 1 || sinks)
    {
        const int items = bsz * blocks * gq;
        EXL3_CHECK_ARG(nsplit <= 8 * (head_dim == 128 ? 32 : 16), "attn_decode: too many context splits for the merge kernel");
        const uint32_t mg = gemv_magic((uint32_t) gq), mbg = gemv_magic((uint32_t) (blocks * gq)), mb = gemv_magic((uint32_t) blocks);
        if (head_dim == 128) attn_merge_kernel<128><<<(items + 1) / 2, 256, 0, st>>>(workspace, (half_t*) out, items, nsplit, gq, blocks, heads_q, mg, mbg, mb, sinks);
        else                 attn_merge_kernel<64><<<(items + 1) / 2, 256, 0, st>>>(workspace, (half_t*) out, items, nsplit, gq, blocks, heads_q, mg, mbg, mb, sinks);
        rc = exl3_check_launch("attn_merge");
    }
    return rc;
}
