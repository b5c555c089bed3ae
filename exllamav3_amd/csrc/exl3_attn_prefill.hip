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
#include <type_traits>
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

// ------------------------------------------------------------------------------------------------------------------------------------------------
// Round 4 (head_dim 128 and 64; three or more query heads per kv head; chunks that give ~192 workgroups or more): v_mfma_f32_32x32x16_f16, an
// exponent-only softmax, one barrier per 64-key tile.
// A workgroup = 8 waves = four query heads of a kv head x the two 32-query blocks of 64 consecutive queries; a wave owns one head's 32 queries
// (described for head_dim 128; 64 halves the k-steps and the output blocks):
//   * S^T = K Q^T in 32 x 32 blocks (kb: keys): A = K rows from LDS (ds_read_b128, lane (key l % 32, dims 16 ks + 8 (l / 32) ..)), B = the wave's Q rows
//     (LDS, pre-scaled); a lane ends with column (query) l % 32 and rows (keys) 32 kb + 8 (i / 4) + 4 (l / 32) + i % 4, i = 0 .. 15;
//   * those 16 values are, eight at a time, an A operand of O = P V if contraction slot 8 h + 4 e + j of k-step (kb, bp) is DEFINED as key
//     32 kb + 16 bp + 8 e + 4 h + j; the matching B operand (two runs of four consecutive keys of one output column) comes from two
//     ds_read_b64_tr_b16 of the row-major V tile.
// Software pipeline inside a wave: iteration t runs
//   phase 1: S_{t+1} = K_{t+1} Q^T (16 + 2 MFMA), the LDS writes of the staged K_{t+2} / V_{t+1} and the loads of K_{t+3} / V_{t+2} between them
//   phase 2: O += P_t V_t and the row sums (16 + 4 MFMA)  beside  the maximum and the probabilities of tile t + 1
// K runs one tile ahead of V in the LDS ring (two slots each): what iteration t writes is first read after the NEXT barrier -- one barrier per
// tile.  The loop is unrolled by two so that ring slots are immediate offsets of the fragment reads and the probability registers alternate
// without copies.  The output goes through LDS (the wave's query rows are free by then) and leaves as whole 256-byte rows.
// Measured (profiles/r04_attn_prefill_ablations.txt): 4096 tokens 660 -> 755 - 790 TFLOP/s, 8192: 750 -> 825 - 850; what bounds it is the SIMD's
// instruction issue (~320 instructions per wave and tile for 38 matrix instructions at ~5 cycles each for the two waves together), not a pipe.
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// (plain fmaxf on matrix-instruction results makes hipcc canonicalise every operand first -- one extra v_max each; the scores here are never NaN)
__device__ __forceinline__ float pw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// maximum of the 32 scores a lane holds for its query in one tile (two 32-key blocks) and of the query's other lane
__device__ __forceinline__ float pw_rowmax(const f32x16_t& s0, const f32x16_t& s1)
{
    float mx = pw_max3(s0[0], s1[0], s0[1]);
    #pragma unroll
    for (int e = 1; e < 15; ++e) mx = pw_max3(mx, s1[e], s0[e + 1]);
    mx = pw_max3(mx, s1[15], s1[15]);
    return pw_max3(mx, xor_lane(mx, 32), mx);
}
__device__ __forceinline__ uint32_t pw_exp_pair(float a, float b)
{
    const half2_t h = { (half_t) __builtin_amdgcn_exp2f(a), (half_t) __builtin_amdgcn_exp2f(b) };
    uint32_t r = half2_as_u32(h);
    // (an empty volatile asm "uses" the result HERE: without it the optimiser sinks the exponentials of a phase into the next basic block, next to
    // their consumers, and the matrix instructions of this phase run bare)
    asm volatile("" : "+v"(r));
    return r;
}

// LDS: K ring (2 x 64 keys x 256 B) | V ring (same) | Q (8 waves x 32 queries x 272 B); at the end the Q area takes the output rows.
// K / V rows are not padded; the conflict-free read comes from permuting the 16-byte chunks of a row: position p of row r holds chunk p ^ f(r),
//   K: f(r) = r & 15                    (a ds_read_b128 lane group touches 16 rows whose r & 15 are distinct -> 16 distinct positions = all 64 banks)
//   V: f(r) = ((r & 3) << 2) | ((r >> 2) & 3)   (a transpose-read group touches 4 consecutive rows x 4 consecutive chunks: the rows' chunk quads differ)
// (PMC: SQ_LDS_BANK_CONFLICT 0.9 % of SQ_LDS_IDX_ACTIVE.)
//
// Workgroup = 8 waves = the four query heads of a kv head x the two 32-query blocks of a 64-query tile: two waves per SIMD (one wave alone issues an
// instruction every ~5 cycles, i.e. at most ~5 beside each 32-cycle matrix instruction; the 64-queries-per-wave form of this kernel, one wave per
// SIMD, was issue-bound at 9.7 VALU per matrix instruction and, once the softmax was cut down, out of registers -- hipcc spilled accumulators).
// The softmax is written for instruction count:
//   * the queries are pre-multiplied by scale * log2(e) (fp16) and minus the running reference maximum of a lane's query enters its score chains on
//     the matrix pipe (one extra k-step per chain: A = ones, B = { hi, lo } with hi + lo = -max; a lane's 16 results all belong to one query), so a
//     probability is ONE instruction, v_exp_f32 of the accumulator -- no multiply-subtract per score;
//   * the row sums come from the matrix pipe: one more 32-column block of "V" that is all ones (4 instructions per tile instead of 32 adds), and land
//     in the accumulator layout, where the final division needs no cross-lane traffic;
//   * the reference maximum moves only when a score exceeds it by 2^8 (probabilities stay below 2^8 in fp16); the tile's scores, the start values and
//     -- after the tile's P V is complete -- the accumulators are adjusted in rarely taken wave-uniform branches.
// head_dim 64 (template HD): rows of 128 B = 8 chunks; K: f(r) = (r >> 1) & 7 (two rows share a 64-bank line: row parity picks the half, the
// position the 4-bank group), V: f(r) = ((r >> 1) & 1) << 2 (the two rows of a transpose-read group that share a bank half take different chunk
// quads); 4 k-steps per score chain and 2 output blocks, i.e. 8 slices per phase instead of 16, two probability pairs per slice.
template <int HD>
__global__ __launch_bounds__(512)
void attn_prefill_w64_kernel(const PrefillAttnArgs a)
{
    static_assert(HD == 128 || HD == 64, "head_dim 128 or 64");
    constexpr int TILE = 64 * HD;                               // halves of one K / V tile
    constexpr int QS = HD + 8;                                  // query / output row stride in halves (272 / 144 B: b128 reads of 16 rows tile the banks)
    constexpr int CH = HD / 8;                                  // 16-byte chunks per row
    constexpr int KSN = HD / 16;                                // k-steps of a score chain
    constexpr int NB = HD / 32;                                 // 32-column output blocks
    constexpr int NS1 = 2 * KSN, NS2 = 4 * NB;                  // slices (matrix-instruction groups) of the score / the P V phase
    constexpr int NCH = (64 * CH) / 512;                        // staging chunks per thread and operand (2 / 1)
    // schedule knobs: fragment read-ahead (K / V in slices, Q in k-steps), the score-phase slices of the LDS writes and of the next loads.  Swept
    // (read-ahead 1 .. 5, writes at 0 .. 10, loads in either phase, static priority for waves 4-7): all inside the run-to-run spread
    // (profiles/r04_attn_prefill_ablations.txt)
    constexpr int KD = 3, QD = 2, VD = 2, PUT_AT = 0, FETCH_AT = HD == 128 ? 5 : 3;
    __shared__ __attribute__((aligned(1024))) half_t ring[4 * TILE + 8 * 32 * QS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, h = lane >> 5;
    half_t* const qs = ring + 4 * TILE + wave * 32 * QS;                                                    // this wave's queries (later: its output rows)
    const int qt = gridDim.y - 1 - blockIdx.y, b = blockIdx.z;                                              // longest tiles first (see the kernel above)
    const int q_len = a.q_len, hq = a.hq, hkv = a.hkv, page_size = a.page_size;
    // blockIdx.x = (kv head, part): a kv head's query heads in groups of four; the last group of a kv head whose group size is not a multiple of 4
    // has waves without a head -- they compute a duplicate of the last one (same barriers, same staging) and store nothing
    const int gq = hq / hkv, parts = (gq + 3) >> 2;
    const int kvh = (int) blockIdx.x / parts, hpart = (int) blockIdx.x - kvh * parts;
    const int head = kvh * gq + min(4 * hpart + (wave & 3), gq - 1);
    const int kv_len = a.cache_seqlens[b];
    const int ctx = kv_len - q_len;
    const int q0 = qt * 64, q0w = q0 + 32 * (wave >> 2);                                                    // the workgroup's / this wave's first query
    const int32_t* bt = a.block_table + (size_t) b * a.blocks_per_seq;
    const float sl = a.scale * 1.44269504f;
    const int q_last = min(q0 + 64, q_len) - 1;
    const int ntiles = (ctx + q_last) / 64 + 1;

    // chunk swizzles of the K / V tiles (see above)
    auto kswz = [&](int r) __attribute__((always_inline)) { return HD == 128 ? (r & 15) : ((r >> 1) & 7); };
    auto vswz = [&](int r) __attribute__((always_inline)) { return HD == 128 ? (((r & 3) << 2) | ((r >> 2) & 3)) : (((r >> 1) & 1) << 2); };

    // ---- staging: 64 keys x CH 16-byte chunks per tile and operand, NCH per thread; HBM -> registers one iteration before the LDS write
    const int row_halves = hkv * HD;                                                                        // (64 rows x hkv x HD halves: far inside 32 bits)
    // LDS offsets (halves) of chunk j of this thread: recomputed where they are used (two VALU each) rather than held in registers
    // (tid_o: the thread index behind an empty asm, refreshed per iteration -- otherwise the loop-invariant offsets are hoisted back into registers)
    int tid_o = tid;
    auto kdst = [&](int j) __attribute__((always_inline)) { const int idx = tid_o + 512 * j, key = idx / CH, ch = idx % CH; return key * HD + ((ch ^ kswz(key)) * 8); };
    auto vdst = [&](int j) __attribute__((always_inline)) { const int idx = tid_o + 512 * j, key = idx / CH, ch = idx % CH; return key * HD + ((ch ^ vswz(key)) * 8); };
    // ONE register set: what iteration t fetches (early in its score phase) is written to LDS early in iteration t + 1.  (A second set -- written two
    // iterations later -- measured the same: the loads cost their instructions, not their latency.)
    half8_t kreg[NCH], vreg[NCH];
    // Where a tile lives: (tile, page index, first key's offset inside the page), advanced by 64 keys per step without divisions (page_size is a run-time
    // value: every / and % was a ~30-instruction scalar sequence, three per iteration) and clamped at the last tile.  The page id itself (a scalar load
    // from the block table) is read one iteration before the fetch that uses it.
    // `base` = element offset of the tile's first row of this kv head in the page array: + 64 rows per step inside a page (two scalar adds; the
    // 64-bit multiplies of a from-scratch address were ~12 scalar instructions per fetch), rebuilt from the page id when a new page begins
    struct TilePos { int tile, pidx, off; int64_t base; bool newpage; };
    auto set_base = [&](TilePos& p, int32_t page_id) __attribute__((always_inline))
    {
        p.base = ((int64_t) page_id * page_size + p.off) * (int64_t) row_halves + (int64_t) kvh * HD; p.newpage = false;
    };
    auto advance = [&](TilePos& p) __attribute__((always_inline))
    {
        if (p.tile < ntiles - 1)
        {
            ++p.tile; p.off += 64; p.base += (int64_t) 64 * row_halves;
            if (p.off >= page_size) { p.off = 0; ++p.pidx; p.newpage = true; }
        }
    };
    auto page_id_of = [&](const TilePos& p) __attribute__((always_inline)) { return bt[min(p.pidx, a.blocks_per_seq - 1)]; };
    // this thread's chunks of a tile as BYTE offsets from the tile's first row: scalar base + 32-bit lane offset is one load instruction without address
    // arithmetic; only the sequence's last tile (keys beyond the length clamped to the last one: finite duplicates, their scores are masked) recomputes them
    uint32_t goffb[NCH];
    #pragma unroll
    for (int j = 0; j < NCH; ++j) { const int idx = tid + 512 * j; goffb[j] = (uint32_t) ((idx / CH) * row_halves + (idx % CH) * 8) * 2u; }
    auto fetch = [&](const TilePos& p, half8_t (&reg)[NCH], const half_t* pages) __attribute__((always_inline))
    {
        const char* base = (const char*) (pages + p.base);                                                  // wave-uniform
        const int kmax = kv_len - 1 - p.tile * 64;
        if (kmax >= 63)
        {
            #pragma unroll
            for (int j = 0; j < NCH; ++j) reg[j] = *((const half8_t*) (base + goffb[j]));
        }
        else
        {
            #pragma unroll
            for (int j = 0; j < NCH; ++j)
            {
                const int idx = tid + 512 * j;
                reg[j] = *((const half8_t*) (base + (uint32_t) (min(idx / CH, kmax) * row_halves + (idx % CH) * 8) * 2u));
            }
        }
    };
    auto putk = [&](int j, const half8_t (&reg)[NCH], half_t* slot) __attribute__((always_inline)) { *((half8_t*) (slot + kdst(j))) = reg[j]; };
    auto putv = [&](int j, const half8_t (&reg)[NCH], half_t* slot) __attribute__((always_inline)) { *((half8_t*) (slot + vdst(j))) = reg[j]; };
    auto Kslot = [&](int i) __attribute__((always_inline)) { return ring + (i & 1) * TILE; };
    auto Vslot = [&](int i) __attribute__((always_inline)) { return ring + (2 + (i & 1)) * TILE; };

    // ---- queries: global -> x scale log2(e) -> this wave's LDS rows (B operands of S^T are read per k-step: query n, dims 16 ks + 8 h ..)
    const int qpos = ctx + min(q0w + n, q_len - 1);
    #pragma unroll
    for (int j = 0; j < (32 * CH) / 64; ++j)
    {
        const int idx = lane + 64 * j, r = idx / CH, ch = idx % CH;
        const int qi = min(q0w + r, q_len - 1);
        half8_t v = *((const half8_t*) (a.q + ((size_t) b * q_len + qi) * a.ldq + (size_t) head * HD + 8 * ch));
        #pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (half_t) ((float) v[e] * sl);
        *((half8_t*) (qs + r * QS + 8 * ch)) = v;
    }
    // ---- per-lane fragment addresses: everything that depends on the slice is an immediate offset
    // (kept as LDS POINTERS into slot 0 of each ring: the slot of an iteration is a compile-time constant of its unrolled copy, so slot, key block
    // and k-step all go into the read's immediate offset and a fragment read costs no address arithmetic)
    const half_t* kl[KSN]; const half_t* vl0[NB]; const half_t* vl1[NB];
    #pragma unroll
    for (int ks = 0; ks < KSN; ++ks) kl[ks] = ring + n * HD + (((2 * ks + h) ^ kswz(n)) * 8);               // key block kb: + 32 * HD (32 rows: the swizzle repeats)
    {
        const int G = lane >> 4, c = lane & 15;
        const int rl = 4 * (G >> 1) + (c >> 2), L = 2 * (G & 1) + ((c & 3) >> 1);                           // (rows rl and rl + 8 of a k-step; 16 rows further: the same swizzle)
        #pragma unroll
        for (int nb = 0; nb < NB; ++nb)
        {
            vl0[nb] = ring + 2 * TILE + rl * HD + (((4 * nb + L) ^ vswz(rl)) * 8) + 4 * (c & 1);            // k-step kst: + 16 * kst * HD
            vl1[nb] = ring + 2 * TILE + (rl + 8) * HD + (((4 * nb + L) ^ vswz(rl + 8)) * 8) + 4 * (c & 1);
        }
    }
    const half_t* const ql = qs + n * QS + 8 * h;                                                           // k-step ks: + 16

    f32x16_t oc[NB], lacc;                                      // O, the row sums (ones column)
    // minus the reference maximum of the lane's query enters every score chain through ONE extra k-step: A = all ones, B = { hi, lo, 0 ... } in the
    // lanes of the first k half (hi + lo = -max as two fp16; products with 1.0 and the fp32 accumulation are exact) -- 4 registers instead of a
    // 16-register start value for operand C
    float mref = 0.0f;
    half8_t negm = { 0, 0, 0, 0, 0, 0, 0, 0 };
    auto set_negm = [&]() __attribute__((always_inline))
    {
        const half_t hi = (half_t) (-mref), lo = (half_t) (-mref - (float) hi);
        negm[0] = h == 0 ? hi : (half_t) 0.0f; negm[1] = h == 0 ? lo : (half_t) 0.0f;
    };
    #pragma unroll
    for (int i = 0; i < 16; ++i) lacc[i] = 0.0f;
    #pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        #pragma unroll
        for (int i = 0; i < 16; ++i) oc[nb][i] = 0.0f;
    const half8_t ones = { (half_t) 1.0f, (half_t) 1.0f, (half_t) 1.0f, (half_t) 1.0f, (half_t) 1.0f, (half_t) 1.0f, (half_t) 1.0f, (half_t) 1.0f };

    // scores of one tile (causal / length mask): wave-uniform test, per-element select
    auto mask_tile = [&](int t, f32x16_t (&S)[2]) __attribute__((always_inline))
    {
        const int key0 = t * 64;
        if (!(key0 + 63 > ctx + min(q0w, q_len - 1) || key0 + 64 > kv_len)) return;
        int h4 = 4 * h;
        asm volatile("" : "+v"(h4));                            // (keeps the 32 key indices inside this rarely taken branch: hoisted, they cost registers in every iteration)
        #pragma unroll
        for (int kb = 0; kb < 2; ++kb)
            #pragma unroll
            for (int i = 0; i < 16; ++i)
            {
                const int key = key0 + 32 * kb + 8 * (i >> 2) + h4 + (i & 3);
                if (!(key <= qpos && key < kv_len)) S[kb][i] = -1.0e30f;
            }
    };
    // slice s of the score phase: k-step ks = s >> 1 of key block kb = s & 1; slice s of the P V phase: k-step kst = s / NB, output block nb = s % NB
    auto k_frag = [&](int s, int slot) __attribute__((always_inline)) { return *((const half8_t*) (kl[s >> 1] + slot * TILE + (s & 1) * 32 * HD)); };
    auto q_frag = [&](int ks) __attribute__((always_inline)) { return *((const half8_t*) (ql + 16 * ks)); };
    auto v_frag = [&](int s, int slot) __attribute__((always_inline))
    {
        const int kst = s / NB, nb = s % NB;                    // k-step (kb, bp) = (kst >> 1, kst & 1): keys 32 kb + 16 bp + 8 e + 4 h + j
        const half4_t v0 = lds_read_tr16(vl0[nb] + slot * TILE + kst * 16 * HD), v1 = lds_read_tr16(vl1[nb] + slot * TILE + kst * 16 * HD);
        return half8_t{ v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
    };
    auto qk_mfma = [&](int s, half8_t ka, half8_t qf, f32x16_t (&S)[2]) __attribute__((always_inline))
    {
        const f32x16_t zero = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        S[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ka, qf, s < 2 ? zero : S[s & 1], 0, 0, 0);
        if (s >= NS1 - 2) S[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, negm, S[s & 1], 0, 0, 0);  // ... - max, once per chain
    };
    // the ones column rides with nb == 0
    auto pv_mfma = [&](int s, half8_t vB, const uint32_t (&pa)[16]) __attribute__((always_inline))
    {
        const int kst = s / NB, nb = s % NB;
        union { uint32_t u[4]; half8_t h8; } p;
        #pragma unroll
        for (int i = 0; i < 4; ++i) p.u[i] = pa[4 * kst + i];
        oc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p.h8, vB, oc[nb], 0, 0, 0);
        if (nb == 0) lacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(p.h8, ones, lacc, 0, 0, 0);
    };
    // probabilities: value pair p = 0 .. 15 of the lane's 32 scores (key block 0: pairs 0-7, key block 1: pairs 8-15)
    auto exp_pair = [&](int p, const f32x16_t (&S)[2]) __attribute__((always_inline))
    {
        return p < 8 ? pw_exp_pair(S[0][2 * p], S[0][2 * p + 1]) : pw_exp_pair(S[1][2 * p - 16], S[1][2 * p - 15]);
    };
    // the reference maximum of a query moves when a score of the tile exceeds it by more than 2^8: this tile's scores and the chains' start values move
    // down by d now; the accumulators (which still take the PREVIOUS tile's P V) are scaled by 2^-d at the end of the phase
    float corr = 1.0f;
    bool pend = false;                                          // wave-uniform
    auto adjust = [&](f32x16_t (&S)[2], float mx) __attribute__((always_inline))
    {
        if (!__any(mx > 8.0f)) return;
        const float d = mx > 8.0f ? mx : 0.0f;
        #pragma unroll
        for (int i = 0; i < 16; ++i) { S[0][i] -= d; S[1][i] -= d; }
        mref += d; set_negm();
        corr = __builtin_amdgcn_exp2f(-d);
        pend = true;
    };
    auto rescale = [&]() __attribute__((always_inline))
    {
        if (!pend) return;
        pend = false;
        #pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            const float cr = __shfl(corr, 8 * (i >> 2) + 4 * h + (i & 3), 64);                 // row (query) of accumulator element i
            #pragma unroll
            for (int nb = 0; nb < NB; ++nb) oc[nb][i] *= cr;
            lacc[i] *= cr;
        }
    };

    // ---- prologue: K_0, K_1, V_0 in LDS, K_2 and V_1 on their way; S_0 with its own maxima as the first reference
    TilePos pk = { 0, 0, 0, 0, false }, pv;                     // (the fetch positions of K and V: K runs one tile ahead)
    int32_t pg_next = 0;
    {
        // all block-table entries first, then K_0, V_0, K_1 in flight together (one HBM latency, not three)
        TilePos p1 = pk; advance(p1);
        TilePos p2 = p1; advance(p2);
        TilePos p3 = p2; advance(p3);
        const int32_t g0 = page_id_of(pk), g1 = page_id_of(p1), g2 = page_id_of(p2), g3 = page_id_of(p3);
        set_base(pk, g0); set_base(p1, g1); set_base(p2, g2); set_base(p3, g3);
        half8_t k1reg[NCH];
        fetch(pk, kreg, a.k_pages); fetch(pk, vreg, a.v_pages); fetch(p1, k1reg, a.k_pages);               // K_0, V_0, K_1
        #pragma unroll
        for (int j = 0; j < NCH; ++j) { putk(j, kreg, Kslot(0)); putv(j, vreg, Vslot(0)); putk(j, k1reg, Kslot(1)); }
        fetch(p2, kreg, a.k_pages); fetch(p1, vreg, a.v_pages);                                             // written in iteration 0: K_2, V_1
        pv = p2; pk = p3;                                                                                   // iteration 0 fetches V_2 and K_3
    }
    __syncthreads();
    f32x16_t S[2];                                              // [key block]
    #pragma unroll
    for (int s = 0; s < NS1; ++s) qk_mfma(s, k_frag(s, 0), q_frag(s >> 1), S);                             // (negm = 0: true scores)
    mask_tile(0, S);
    {
        float mx = pw_rowmax(S[0], S[1]);
        if (mx < -1.0e29f) mx = 0.0f;                           // a query without any visible key (not a valid call): all its probabilities become 0
        #pragma unroll
        for (int i = 0; i < 16; ++i) { S[0][i] -= mx; S[1][i] -= mx; }
        mref = mx; set_negm();
    }
    uint32_t pab[2][16];                                        // fp16 probability pairs of the current / the next tile (roles alternate with SET)
    #pragma unroll
    for (int p = 0; p < 16; ++p) pab[0][p] = exp_pair(p, S);

    // one iteration: phase 1 = tile t + 1's scores (matrix instructions, fragment reads, the staging traffic), phase 2 = tile t's P V beside tile
    // t + 1's maximum and probabilities.  MORE = tile t + 1 exists.  Ring: at the top of iteration t the slots of K_t and V_{t-1} are dead and take
    // K_{t+2} and V_{t+1} (in registers since the previous iteration); they are first read after the NEXT barrier
    auto iteration = [&](int t, auto more_c, auto set_c) __attribute__((always_inline))
    {
        constexpr bool MORE = decltype(more_c)::value;
        constexpr int SET = decltype(set_c)::value;             // = t & 1: the ring slots and the probability arrays of this unrolled copy
        tid_o = tid; asm volatile("" : "+v"(tid_o));
        __syncthreads();
        constexpr int kbuf = 1 - SET, vbuf = SET;               // ring slots of K_{t+1} and V_t
        // fragment reads run KD / QD / VD slices ahead of their use: a score slice is ONE matrix instruction (32 cycles) and an LDS read under
        // eight waves' traffic takes well over 100
        half8_t kf[NS1], qfr[KSN], vf[NS2];
        if constexpr (MORE)
        {
            #pragma unroll
            for (int i = 0; i < KD; ++i) kf[i] = k_frag(i, kbuf);
            #pragma unroll
            for (int i = 0; i < QD; ++i) qfr[i] = q_frag(i);
        }
        else
        {
            #pragma unroll
            for (int i = 0; i < VD; ++i) vf[i] = v_frag(i, vbuf);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase 1: S_{t+1}
        if constexpr (MORE)
        {
            #pragma unroll
            for (int s = 0; s < NS1; ++s)
            {
                if (s + KD < NS1) kf[s + KD] = k_frag(s + KD, kbuf);
                if (!(s & 1) && (s >> 1) + QD < KSN) qfr[(s >> 1) + QD] = q_frag((s >> 1) + QD);
                if (s >= NS1 - VD) vf[s - (NS1 - VD)] = v_frag(s - (NS1 - VD), vbuf);
                qk_mfma(s, kf[s], qfr[s >> 1], S);
                // the staging traffic rides between the matrix instructions (all eight waves doing it together after the barrier left the pipe idle):
                // the LDS writes of K_{t+2} / V_{t+1} (in registers since the previous iteration) first, then the loads of K_{t+3} / V_{t+2}
                if (s >= PUT_AT && s < PUT_AT + NCH) putk(s - PUT_AT, kreg, Kslot(t));
                if (s >= PUT_AT + NCH && s < PUT_AT + 2 * NCH) putv(s - PUT_AT - NCH, vreg, Vslot(t + 1));
                if (s == FETCH_AT) fetch(pk, kreg, a.k_pages);
                if (s == FETCH_AT + 1)
                {
                    fetch(pv, vreg, a.v_pages); pv = pk; advance(pk);
                    // the block-table entry of the NEXT iteration's K fetch.  (After the first barrier hipcc no longer proves the table unwritten and
                    // reads it with a vector load; asked for here and consumed at the end of the iteration, nobody waits for it.)
                    pg_next = page_id_of(pk);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            mask_tile(t + 1, S);
        }
        // ---- phase 2: O += P_t V_t (and the row sums) beside the maximum and the probabilities of tile t + 1: slice 0 the maximum, slices
        // 1 .. NS2 - 1 the pairs, 16 / NS2 each, the last slice also the rest
        uint32_t (&pa)[16] = pab[SET];
        uint32_t (&pn)[16] = pab[1 - SET];
        constexpr int PPS = 16 / NS2;
        #pragma unroll
        for (int s = 0; s < NS2; ++s)
        {
            if (s + VD < NS2) vf[s + VD] = v_frag(s + VD, vbuf);
            pv_mfma(s, vf[s], pa);
            if constexpr (MORE)
            {
                if (s == 0) adjust(S, pw_rowmax(S[0], S[1]));
                else
                {
                    #pragma unroll
                    for (int p = (s - 1) * PPS; p < (s == NS2 - 1 ? 16 : s * PPS); ++p) pn[p] = exp_pair(p, S);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MORE)
        {
            rescale();
            if (pk.newpage) set_base(pk, __builtin_amdgcn_readfirstlane(pg_next));           // (wave-uniform, once per page)
        }
    };
    {
        using T = std::true_type; using F = std::false_type; using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
        int t = 0;
        for (; t + 2 < ntiles; t += 2) { iteration(t, T{}, S0{}); iteration(t + 1, T{}, S1{}); }
        if (t + 1 < ntiles) { iteration(t, T{}, S0{}); iteration(t + 1, F{}, S1{}); }
        else iteration(t, F{}, S0{});
    }

    // ---- normalise (row sums and outputs share the accumulator layout), stage through this wave's LDS rows (its queries are no longer needed),
    // store whole rows
    // (everything lane-dependent is re-derived from the thread index behind an empty asm: values shared with the prologue are NOT kept alive, or
    // spilled, across the main loop for this)
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wave_e = tid_e >> 6, n_e = lane_e & 31, h_e = lane_e >> 5, head_e = kvh * gq + 4 * hpart + (wave_e & 3);
    const bool has_head = 4 * hpart + (wave_e & 3) < gq;
    half_t* os = ring + 4 * TILE + wave_e * 32 * QS;
    #pragma unroll
    for (int i = 0; i < 16; ++i)
    {
        const int r = 8 * (i >> 2) + 4 * h_e + (i & 3);
        const float li = lacc[i] > 0.0f ? 1.0f / lacc[i] : 0.0f;                             // (a query without any visible key: zeros, as the kernel above)
        #pragma unroll
        for (int nb = 0; nb < NB; ++nb) os[r * QS + 32 * nb + n_e] = (half_t) (oc[nb][i] * li);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    #pragma unroll
    for (int j = 0; j < (32 * CH) / 64; ++j)
    {
        const int idx = lane_e + 64 * j, r = idx / CH, ch = idx % CH;
        if (q0w + r < q_len && has_head)
            *((half8_t*) (a.out + (((size_t) b * q_len + q0w + r) * hq + head_e) * HD + 8 * ch)) = *((const half8_t*) (os + r * QS + 8 * ch));
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
    hipStream_t st = (hipStream_t) stream;
    // head_dim 128 with four query heads per kv head: one wave per SIMD, 64 queries per wave (round 4)
    static const int w64 = [] { const char* e = getenv("EXL3_HIP_ATTN_PREFILL_W64"); return e ? atoi(e) : 1; }();
    // (four heads per workgroup is fixed there: below ~3/4 of a workgroup per CU the older kernel, which then takes fewer heads per workgroup, is ahead --
    // 4096-token context + 512 new tokens: 107 vs 111 us, 1024 tokens: 32.9 vs 34.0, 2048: 79.9 vs 66.2 -- so short chunks stay with it)
    const char* e_min = getenv("EXL3_HIP_ATTN_PREFILL_W64_MIN_WGS");                 // (read per call: the tests force either kernel on small shapes)
    const int w64_min_wgs = e_min ? atoi(e_min) : 192;
    // group sizes that fill at least 60 % of the head slots of their workgroups: 3 .. (4096 tokens, same box: 7 heads per kv head 661 vs 422 TFLOP/s on the
    // older kernel, 3: 580 vs 418, 5: 475 vs 430; 2: 381 vs 444 and 1: 199 vs 296 stay with it)
    const int parts = (gq + 3) / 4;
    const char* e_fill = getenv("EXL3_HIP_ATTN_PREFILL_W64_MIN_FILL");                // (percent of head slots used; tuning / tests)
    const int min_fill = e_fill ? atoi(e_fill) : 60;
    if (w64 && 100 * gq >= min_fill * 4 * parts && q_len >= 64 && (int64_t) (heads_kv * parts) * ((q_len + 63) / 64) * bsz >= w64_min_wgs)
    {
        dim3 gridw(heads_kv * parts, (q_len + 63) / 64, bsz);
        if (head_dim == 128) attn_prefill_w64_kernel<128><<<gridw, 512, 0, st>>>(a); else attn_prefill_w64_kernel<64><<<gridw, 512, 0, st>>>(a);
        return exl3_check_launch("attn_prefill_w64");
    }
    int gw = gq % 4 == 0 ? 4 : (gq % 2 == 0 ? 2 : 1);                                // query heads per workgroup (they share a kv head) ...
    const int64_t qtiles = (int64_t) ((q_len + PA_BM - 1) / PA_BM) * bsz;
    while (gw > 1 && qtiles * (heads_q / gw) < 512) gw >>= 1;                         // ... fewer when the launch would not fill the 256 CUs twice
    dim3 grid(heads_q / gw, (q_len + PA_BM - 1) / PA_BM, bsz);
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
