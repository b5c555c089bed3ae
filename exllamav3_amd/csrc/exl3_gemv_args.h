// Kernel-argument block shared by the GEMV kernel generations.
#pragma once
#include <stddef.h>
#include "exl3_common.cuh"

#ifndef G2_PF
#define G2_PF 2             // gen 2: tile rows per work unit = depth of the per-wave weight-row register ring (2 or 4)
#endif
#define GEMV_IN_ROTATED   1   // the input Hadamard was applied by the producer (glue kernel): mat[i].xh / xsum
#define GEMV_OUT_DEFERRED 2   // write raw rotated-basis partial slabs [colblock][S][m][128] fp32; a glue kernel finishes
#define GEMV_IN_NORM      4   // A is the residual stream: the kernel applies RMSNorm (norm_w, per-block sums of squares ss_part, eps) before the input Hadamard
#define GEMV_IN_ACT       8   // input = silu(g) * u formed from the previous launch's gate / up slabs while the activation fragments are built
#define GEMV_IN_RESID     16  // (with GEMV_IN_NORM) the residual stream is finished HERE: resid_new = resid_in + out-had(sum of the producer's split-k slabs) * svh,
                              // per Hadamard block while the activation fragments are built (replaces the glue_resid launch).  The row scale of the
                              // RMSNorm is taken from the PREVIOUS residual's sums of squares (ss_part); consumers of this launch's outputs
                              // multiply by rsqrt(ms_new + eps) / rsqrt(ms_prev + eps) (GemvRescale) -- the linear map commutes with the scalar
#define GEMV_OUT_ATOMIC    32  // (generation 4) the finished rows are ADDED into a 64-bit fixed-point accumulator (mat[i].C = int64 [m][n], value * 2^32: the
                              // residual stream of the "fx" decode pipeline): every workgroup applies the output Hadamard, svh (and slice 0 the bias) to its
                              // OWN split-k partial -- the Hadamard is linear -- and adds it with agent-scope integer atomics.  Integer addition is
                              // associative, so the sum does not depend on arrival order (bit-reproducible), and no reduce / residual launch follows.
#define GEMV_IN_FX         64  // (generation 4, with GEMV_IN_NORM) A is that accumulator: x = fp16(A / 2^32); ss_part holds the block sums of squares of the
                              // PREVIOUS residual (the row scale is an estimate, corrected downstream exactly as for GEMV_IN_RESID); the workgroups of
                              // column block 0 of matrix 0 write the block sums of squares of THIS residual to rs_ss_out
#define GEMV_FX_SCALE 4294967296.0
// Non-finite or out-of-range contributions must not turn into finite garbage (the reference's fp16 residual carries Inf / NaN through to the logits):
// a value that is NaN, Inf or >= GEMV_FX_LIMIT in magnitude (16x beyond the fp16 range the reference's residual saturates at) REPLACES the accumulator
// with GEMV_FX_POISON by an atomic exchange -- not an add: k poisons cannot cancel mod 2^64 -- and every reader (fx_to_float) turns an accumulator
// beyond +-2^60 into NaN.  Legal adds are < 2^52 each in magnitude: it takes > 768 maximal NEGATIVE adds after the poisoning to bring the
// word (2^62) back under the readers' 2^60 threshold -- a step performs at most (layers x 2 boundaries x splits) ~ 2.6 k adds per address, of
// alternating sign and typical magnitude 2^32..2^36, so this is out of reach in practice but not by construction; fx_add_kernel and the
// all-reduce kernel keep a poisoned word sticky, the atomic path (fx_atomic_add) does not re-check it.
#define GEMV_FX_LIMIT 1048576.0f
#define GEMV_FX_POISON 0x4000000000000000ull
#ifdef __HIPCC__
__device__ __forceinline__ float fx_to_float(uint32_t lo, uint32_t hi)
{
    const float v = (float) (int32_t) hi + (float) lo * 2.3283064365386963e-10f;
    return ((uint32_t) (hi + 0x10000000u) < 0x20000000u) ? v : __builtin_nanf("");
}
__device__ __forceinline__ long long fx_from_float(float v)
{
    return (__builtin_fabsf(v) < GEMV_FX_LIMIT) ? __double2ll_rn((double) v * GEMV_FX_SCALE) : (long long) GEMV_FX_POISON;
}
__device__ __forceinline__ void fx_atomic_add(unsigned long long* acc, float v)
{
    if (__builtin_fabsf(v) < GEMV_FX_LIMIT)
        __hip_atomic_fetch_add(acc, (unsigned long long) __double2ll_rn((double) v * GEMV_FX_SCALE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        __hip_atomic_exchange(acc, (unsigned long long) GEMV_FX_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif
#define GEMV_IN_ACTFX     128  // (generation 4) down_proj whose input silu(g) * u is formed from the gate / up rows that a GEMV_OUT_ATOMIC gate|up launch ADDED into two
                              // fixed-point accumulators (act_g / act_u reinterpreted as int64 [m][k]): no slab reduction in the prologue, 2 KB per task
#define GEMV_IN_ATTM      256  // (generation 4) o_proj whose input is the decode attention's OUTPUT, finished here: the flash-decoding merge of the context-split
                              // partial records (exl3_attn_decode.hip: {m, l, o[128]} per (sequence, kv block, query index, split)) is the preparation task of
                              // the (row, head) that needs it -- head_dim 128: one query head = one Hadamard block -- so the merge launch disappears
#define GEMV_IN_QKVM      512  // (generation 4) o_proj fed by the q|k|v launch's deferred slabs WITHOUT the attention core in between (the linears-only decode step): the
                              // preparation task of (row, block) finishes q block `block` (split-k reduce, output Hadamard, row-scale correction, svh, RoPE:
                              // exl3_glue_device.cuh qkv_block_finish, the arithmetic of glue_qkv_kernel) and half-waves without a preparation task in the
                              // column-block-0 workgroups finish and append the K / V rows -- the glue_qkv launch disappears
#define GEMV_MAX_MATS 4

// r = rsqrt(sum(ss_new[row]) / k + eps) / rsqrt(sum(ss_prev[row]) / k + eps): the exact RMSNorm scale over the estimate a GEMV_IN_RESID launch used.
struct GemvRescale { const float* ss_prev; const float* ss_new; int k; float eps; };

// GEMV_IN_ATTM operands: partial records part[((row * blocks + h) * gq + i) * nsplit + split][132] fp32 = {m, l, -, -, o[128]} of query head h * gq + i
struct GemvAttm { const float* part; int nsplit, gq, blocks; uint32_t magic_gq; int hd; };      // hd: head_dim 128 | 64 (64: two query heads per Hadamard block, records of exl3_attn_decode.hip's NSUB = 2 form)

// GEMV_IN_QKVM operands: the q|k|v launch's slab sets (slab(c, s, row) = base + ((c * S + s) * m + row) * 128) and their column scales, the per-step
// rope tables and cache rows of exl3_qkv_prep, the row-scale correction, the 4-bit paged cache of the layer
struct GemvQkvm
{
    const float* sq; const float* sk; const float* sv; int S; int rope_mode;
    const half_t* svh_q; const half_t* svh_k; const half_t* svh_v;
    const float* rope_sin; const float* rope_cos; const int64_t* slots;
    GemvRescale rs;
    half_t* q_out;                          // optional [m][k]: the finished queries (written by the column-block-0 workgroups)
    uint32_t* k_cache; half_t* k_scales; uint32_t* v_cache; half_t* v_scales;
    int hd, kvb;                            // head_dim (64 | 128), 128-value blocks of the kv vector (heads_kv * hd / 128)
};

struct GemvMat
{
    const uint32_t* B;
    const half_t* suh;
    const half_t* svh;
    const half_t* bias;
    void* C;
    const half_t* xh;      // optional pre-rotated input [m][k] fp16 = had128(x * suh) (flags & GEMV_IN_ROTATED)
    const float* xsum;     // optional per-128-block sums of xh [m][k/128] fp32 (needed by the cb2 FAST variant)
    int n;
    int cb_first;          // first column block of this matrix in the flattened grid
    int ws_offset;         // float offset of this matrix's slabs in the workspace
};

// In-kernel tail epilogue of the gen-2 GEMV (deferred mode only): the LAST workgroup to finish a column block (arrival
// ticket in device memory) reduces the split-k slabs of that block and runs what would otherwise be a separate glue launch.
#define GEMV_EPI_NONE 0
#define GEMV_EPI_NORM 1   // 1 matrix (o_proj / down_proj): out-had*svh(+bias) -> residual += y ; the last block overall: RMSNorm -> in-had
#define GEMV_EPI_ACT  2   // 2 matrices (gate, up): out-had*svh -> silu(g)*u -> in-had of down_proj
#define GEMV_EPI_RESID 4  // 1 matrix (o_proj / down_proj): out-had*svh(+bias) -> residual += y -> per-block sums of squares (glue_resid inside the launch)
#define GEMV_EPI_QKV  3   // 3 matrices (q, k, v), head_dim 128: out-had*svh -> RoPE -> q out / quantized KV-cache append

struct GemvEpi
{
    int mode;
    int rows_per_pass;                       // rows whose slabs fit the workgroup LDS at once (host: lds >= rows * sets * S * 512 B)
    int ticket_global;                       // index of the whole-launch ticket (NORM)
    int xcd_local;                           // all slices of a column block run on ONE XCD (host-arranged): the hand-off stays in that XCD's L2
    float* ss_out;                           // RESID: [m][n/128] per-block sums of squares for the consumer's GEMV_IN_NORM
    uint32_t* tickets;                       // zero on entry, zero on exit
    // consumers of the produced activation (NORM: up to 3, ACT: 1): xh = fp16(had(x * suh)/sqrt(128)), per-block sums
    const half_t* t_suh[3]; half_t* t_xh[3]; float* t_xsum[3]; int t_count;
    // NORM
    half_t* resid; const half_t* norm_w; float eps; int ss_offset /* workspace floats: [m][n/128] sums of squares */; half_t* xn_out;
    // ACT
    half_t* a_out;
    // QKV
    half_t* q_out; half_t* k_out; half_t* v_out;
    const float* rope_sin; const float* rope_cos;   // [m][64] fp32, already scaled by attn_factor (exl3_rope_table)
    const int32_t* positions;
    uint32_t* k_cache; half_t* k_scales; uint32_t* v_cache; half_t* v_scales;
    const int32_t* block_table; int blocks_per_seq; int page_size; int k_bits; int v_bits;
    int hq, hkv, rope_mode;
};

// Table mode (MoE / indexed exl3_mgemm, quant/exl3_gemm_kernel.cuh:88-292): the matrices are device-side pointer tables and the
// slot -> matrix mapping (`indices`, optional expert-range filter, optional routing weights) is resolved on the device.
struct GemvTable
{
    const uint64_t* B; const uint64_t* suh; const uint64_t* svh;   // [num matrices] device addresses
    const int64_t* indices;                                         // [bszm] or null (slot j -> matrix j)
    const half_t* weights;                                          // [bszm] or null
    void* C;                                                        // [bszm][m][n]
    int bszm, min_index, max_index, n, cbs_per_mat;
    int64_t a_slot_stride;                                          // elements between the inputs of two slots (0: one shared input)
    int64_t c_slot_stride;                                          // elements between the outputs of two slots
    const half_t* act_u;                                            // non-null: the input of slot j is fp16(silu(A_j) * act_u_j) (A = gate, act_u = up outputs,
                                                                    // same slot stride): activation.cu silu_mul folded into the down launch of a MoE block
    // generation 4 (exl3_gemv4.kspec.hip, G4_MODE_TACT / GEMV_OUT_ATOMIC table launches)
    const uint64_t* act_svh; int act_up_off;                        // svh table of the gate|up launch whose slabs are this launch's input: gate svh = act_svh[matrix],
                                                                    // up svh = act_svh[matrix + act_up_off]
    int slots_per_token;                                            // GEMV_OUT_ATOMIC: slot j adds into the rows of token j / slots_per_token
    // per-matrix output widths (quant/exl3_gemm.cu:433-447): matrix i is n_list[i] <= n columns wide and writes [m][n_list[i]] at c_list[i];
    // n then only sizes the grid and the slabs (generation 2 table launches only)
    const int32_t* n_list; const uint64_t* c_list;
};

struct SlotRef_t { int mat_index; float weight; };

// slot -> (matrix, routing weight).  With an expert range [min_index, max_index) the in-range entries are compacted to the front in
// order (reference: single-thread pre-pass + grid sync, exl3_gemm_kernel.cuh:101-127); every workgroup recomputes it from <= bszm entries.
__device__ __forceinline__ SlotRef_t resolve_slot(const GemvTable& t, int slot)
{
    SlotRef_t r; r.mat_index = slot; r.weight = 1.0f;
    if (!t.indices) { if (t.weights) r.weight = (float) t.weights[slot]; return r; }
    if (t.min_index < 0)
    {
        r.mat_index = (int) t.indices[slot];
        if (t.weights) r.weight = (float) t.weights[slot];
        return r;
    }
    r.mat_index = -1;
    int cnt = 0;
    for (int i = 0; i < t.bszm; ++i)
    {
        const int idx = (int) t.indices[i];
        if (idx >= t.min_index && idx < t.max_index)
        {
            if (cnt == slot) { r.mat_index = idx - t.min_index; if (t.weights) r.weight = (float) t.weights[i]; }
            ++cnt;
        }
    }
    return r;
}

struct GemvArgs
{
    // ---- hot block: everything a generation-2 workgroup reads before its first vector load sits in the first two 64-byte lines of the
    // argument block (and is read in one batch, exl3_gemv2.kspec.hip prologue): the fields used to be spread over 8 cache lines and the
    // prologue paid a scalar-cache miss per dependent step (tools/gemv_timeline.py)
    int num_mats;
    int m;                 // rows in this pass (1..16)
    int k;
    int S;                 // k-slices
    int kslice;            // elements per slice (multiple of 128)
    int c_fp32;
    int flags;             // GEMV_IN_ROTATED | GEMV_OUT_DEFERRED
    int chunk_blocks;      // gen 2: Hadamard blocks of activation fragments a wave keeps in LDS at a time
    int cpw;               // > 0: wave-per-column-block layout (exl3_gemv2.kspec.hip G2_IS_WPC): column blocks per workgroup = waves per workgroup;
                           // mat[i].cb_first then counts GROUPS of cpw column blocks (groups never straddle matrices)
    int cbf[3];            // copies of mat[1..3].cb_first
    uint32_t magic_m, magic_nwv, magic_nhw;      // exact-division multipliers (gemv_magic / gemv_udiv below; 0 = divisor 1)
    float eps;
    const half_t* A;       // [m][k] (already offset to the first row of this pass)
    float* workspace;      // slab region of this launch
    const half_t* norm_w;  // GEMV_IN_NORM: RMSNorm weight [k]
    const float* ss_part;  // GEMV_IN_NORM: [m][k/128] sums of squares of the residual blocks (exl3_glue_resid)
    float* ws_debug;       // diagnostics builds: fixed area at 48 MiB of the per-device workspace
    int64_t c_row_offset;  // first output row of this pass
    int nwv;               // gen 2: waves per workgroup (= blockDim.x / 64; reading blockDim costs a scalar load from the implicit arguments
                           // BEHIND this block, i.e. one more cache line and one more dependent round trip)
    int pad_hot_[3];       // mat[] starts on the third line
    GemvMat mat[GEMV_MAX_MATS];
    GemvTable tbl;         // table mode when tbl.B != nullptr
    // ACT mode (GEMV_IN_ACT): the input is silu(g) * u of the PREVIOUS launch's deferred gate / up slabs, finished here per Hadamard block
    const float* act_g; const float* act_u;      // slab bases [inter/128][act_S][m][128] fp32 (the other workspace region)
    const half_t* act_svh_g; const half_t* act_svh_u;
    int act_S;
    GemvRescale act_rs;                          // ACT mode: g, u came from a GEMV_IN_RESID launch (ss_new == nullptr: no rescale)
    // RESID mode (GEMV_IN_NORM | GEMV_IN_RESID): A = resid_in (fp16 [m][k], read only); producer slabs [k/128][rs_S][m][128] + its svh; the
    // workgroups of column block 0 of matrix 0 write resid_out (fp16 [m][k], a DIFFERENT buffer) and ss_out [m][k/128]
    const float* rs_slab; const half_t* rs_svh; half_t* rs_resid_out; float* rs_ss_out; int rs_S;
    GemvEpi epi;
    // fx pipeline: a buffer this launch clears as a side job (n16 16-byte chunks spread over the workgroups): the gate / up accumulators of the NEXT
    // gate|up launch are zeroed by the o_proj launch in front of it -- no memset node in the graph
    void* fx_zero; int fx_zero_n16;
    GemvAttm attm;
    GemvQkvm qkvm;
};
static_assert(offsetof(GemvArgs, mat) == 128, "GemvArgs: the hot block is two 64-byte lines");

// floor(x / d) for x < 2^32 / d as one multiply-high: M = floor(2^32 / d) + 1 (d >= 2), M = 0 encodes d == 1.  The kernels divide by launch
// constants (rows, waves per workgroup) in their prologue and prep loops; a runtime integer division is a ~30-instruction v_rcp sequence each.
static inline uint32_t gemv_magic(uint32_t d) { return d <= 1 ? 0u : (uint32_t) ((1ull << 32) / d) + 1u; }
#ifdef __HIPCC__
__device__ __forceinline__ int gemv_udiv(int x, uint32_t M) { return M ? (int) __umulhi((uint32_t) x, M) : x; }
#endif

// generation-2 kernels: one translation unit per K (exl3_gemv2.kspec.hip compiled with -DG2_K=1..8)
size_t exl3_gemv2_lds_bytes(int ng, int var, int cb, int nwv, int m, int chunk_blocks);
#define G2_DECL(KK) void exl3_gemv2_launch_k##KK(int cb, int var, int ng, int nwv, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args);
G2_DECL(1) G2_DECL(2) G2_DECL(3) G2_DECL(4) G2_DECL(5) G2_DECL(6) G2_DECL(7) G2_DECL(8)
#undef G2_DECL
// generation 3 (exl3_gemm3.kspec.hip): 5..64 rows per pass through an LDS transpose into 16x16x32 MFMAs
#define G3_DECL(KK) void exl3_gemm3_launch_k##KK(int cb, int mt, int var, int nwt, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args); \
                    int exl3_gemm3_max_waves_k##KK(int mt, int rot);
G3_DECL(1) G3_DECL(2) G3_DECL(3) G3_DECL(4) G3_DECL(5) G3_DECL(6) G3_DECL(7) G3_DECL(8)
#undef G3_DECL
size_t exl3_gemm3_lds_bytes(int m, int chunk_blocks, int nwt);
// generation 4 (exl3_gemv4.kspec.hip): 1..4 rows, activation quads in the A-broadcast register layout, no per-wave prologue
#define G4_DECL(KK) void exl3_gemv4_launch_k##KK(int cb, int var, int mode, int nwv, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args);
G4_DECL(1) G4_DECL(2) G4_DECL(3) G4_DECL(4) G4_DECL(5) G4_DECL(6) G4_DECL(7) G4_DECL(8)
#undef G4_DECL
size_t exl3_gemv4_lds_bytes(int mode, int nwv, int m, int blocks_per_slice);
