// routing_std: MoE router for one token step.  scores = hidden @ gate (fp32 accumulate, fp16 out), top-K experts by logit
// (descending, ties to the lower index), weights = softmax over the K selected logits (fp16), optional bias on the logits.
// reference: exllamav3_ext/routing.cu:457-590 (routing_std_topk_kernel) + routing_gemv :955-1010; python seam modules/block_sparse_mlp.py:51-93.
// One workgroup per row; E <= 512, K <= 16.  This is the caller of the indexed exl3_mgemm.
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include "exl3_gemv_args.h"           // fx_to_float: the fixed-point residual's reader (NaN for a poisoned accumulator)

#define ROUTING_MAX_EXPERTS 512
#define ROUTING_MAX_K 16

// NORM: `hidden` is the fp16 RESIDUAL stream and the router input is its RMSNorm, xn = fp16(resid * norm_w * rsqrt(mean(resid^2) + eps)), with the
// row's mean square taken from the per-block sums of squares the residual kernel left behind (ss_part [rows][H/128], the same fixed-order sum the
// GEMV_IN_NORM launches use).  The workgroup forms xn once (LDS, dynamic: H halves), writes it to xn_out for the expert launches and routes on it:
// the rms_norm launch in front of a MoE block disappears (modules/block_sparse_mlp.py:1099-1130 runs norm, router and experts as separate ops).
// NORM == 2 (fx pipeline): `hidden` is the residual stream in 64-bit fixed point (int64 [rows][H], value * 2^32: the accumulator the GEMV_OUT_ATOMIC
// launches add into).  The workgroup reads the whole row anyway, so it takes the EXACT sums of squares of this residual (no estimate from the previous
// one, unlike the GEMV_IN_FX launches whose workgroups see one k-slice each) and publishes them per Hadamard block in ss_out for the next
// GEMV_IN_FX launch's estimate and exl3_glue_qkv_rs's correction.
struct RoutingNorm { const half_t* norm_w; const float* ss_part; half_t* xn_out; float eps;
                     const uint16_t* per_expert_scale;         // optional bf16 [E]: weight_k *= scale[expert_k] after the softmax (routing.cu:587-588)
                     float* ss_out; };

template <int NORM>
__global__ __launch_bounds__(256)
void routing_std_kernel(const half_t* __restrict__ hidden, const half_t* __restrict__ gate, const half_t* __restrict__ bias,
                        half_t* __restrict__ scores, int64_t* __restrict__ topk_indices, half_t* __restrict__ topk_weights,
                        int H, int E, int K, int64_t* __restrict__ gu_slots, int rows, RoutingNorm nrm)
{
    extern __shared__ __attribute__((aligned(16))) char dyn_s[];
    __shared__ float logit_s[ROUTING_MAX_EXPERTS];
    __shared__ float sel_logit[ROUTING_MAX_K];
    __shared__ int sel_idx[ROUTING_MAX_K];
    __shared__ float ss_s[256];                                              // NORM == 2: block sums of squares of the row (H <= 32768)
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const half_t* xg = hidden + (size_t) row * H;
    half_t* xn_s = (half_t*) dyn_s;
    // the router input, element h: global memory (plain) or the normalised row in LDS (NORM, valid after form_xn())
    auto xin = [&] (int h) -> half_t { if constexpr (NORM != 0) return xn_s[h]; else return xg[h]; };
    auto form_xn = [&] ()
    {
        if constexpr (NORM == 2)
        {
            const int nblk = H >> 7, l32 = tid & 31, nch4 = H >> 2;
            const int64_t* rg = (const int64_t*) hidden + (size_t) row * H;
            auto fx = [] (uint32_t lo, uint32_t hi) -> half_t { return f2h(fx_to_float(lo, hi)); };
            auto block_ss = [&] (half4_t xv) -> float                        // sum of squares of one Hadamard block (32 consecutive threads), fixed order
            {
                const float r0 = (float) xv.x, r1 = (float) xv.y, r2 = (float) xv.z, r3 = (float) xv.w;
                float ssq = r0 * r0;
                ssq = __builtin_fmaf(r1, r1, ssq); ssq = __builtin_fmaf(r2, r2, ssq); ssq = __builtin_fmaf(r3, r3, ssq);
                #pragma unroll
                for (int i = 1; i < 32; i <<= 1) ssq += xor_lane(ssq, i);
                return ssq;
            };
            auto row_scale = [&] () -> float                                 // the consumers' fixed-order sum of the block sums
            {
                float s2 = 0.0f;
                for (int b0 = 0; b0 < nblk; b0 += 32)
                {
                    float v = (b0 + l32 < nblk) ? ss_s[b0 + l32] : 0.0f;
                    #pragma unroll
                    for (int i = 1; i < 32; i <<= 1) v += xor_lane(v, i);
                    s2 += v;
                }
                return __frsqrt_rn(s2 / (float) H + nrm.eps);
            };
            auto norm4 = [] (half4_t xv, half4_t wv, float rmf) -> half4_t
            {
                return half4_t{ f2h((float) xv.x * (float) wv.x * rmf), f2h((float) xv.y * (float) wv.y * rmf),
                                f2h((float) xv.z * (float) wv.z * rmf), f2h((float) xv.w * (float) wv.w * rmf) };
            };
            if (nch4 <= 1024)
            {
                // H <= 4096: the row (4 x 32 bytes per thread) and the norm weight are requested in ONE batch and stay in registers: one memory round trip
                // (next to the gate rows already in flight), not residual -> barrier -> norm weight
                uint4_t f0[4], f1[4]; half4_t wv[4], xv[4];
                #pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    const int c = min(tid + 256 * j, nch4 - 1);              // 32 consecutive threads = one Hadamard block (H / 4 is a multiple of 32)
                    f0[j] = ((const uint4_t*) rg)[2 * c]; f1[j] = ((const uint4_t*) rg)[2 * c + 1];
                    wv[j] = ((const half4_t*) nrm.norm_w)[c];
                }
                #pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    const int c = tid + 256 * j;
                    xv[j] = half4_t{ fx(f0[j].x, f0[j].y), fx(f0[j].z, f0[j].w), fx(f1[j].x, f1[j].y), fx(f1[j].z, f1[j].w) };    // x = fp16(R / 2^32), as the GEMV_IN_FX launches read it
                    const float ssq = block_ss(xv[j]);
                    if (c < nch4 && l32 == 0) { ss_s[c >> 5] = ssq; nrm.ss_out[(size_t) row * nblk + (c >> 5)] = ssq; }
                }
                __syncthreads();
                const float rmf = row_scale();
                #pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    const int c = tid + 256 * j;
                    if (c < nch4)
                    {
                        const half4_t o = norm4(xv[j], wv[j], rmf);
                        ((half4_t*) xn_s)[c] = o;
                        ((half4_t*) (nrm.xn_out + (size_t) row * H))[c] = o;
                    }
                }
            }
            else
            {
                for (int c = tid; c < nch4; c += 256)
                {
                    const uint4_t f0 = ((const uint4_t*) rg)[2 * c], f1 = ((const uint4_t*) rg)[2 * c + 1];
                    const half4_t xv = { fx(f0.x, f0.y), fx(f0.z, f0.w), fx(f1.x, f1.y), fx(f1.z, f1.w) };
                    ((half4_t*) xn_s)[c] = xv;
                    const float ssq = block_ss(xv);
                    if (l32 == 0) { ss_s[c >> 5] = ssq; nrm.ss_out[(size_t) row * nblk + (c >> 5)] = ssq; }
                }
                __syncthreads();
                const float rmf = row_scale();
                for (int c = tid; c < nch4; c += 256)                        // each thread renormalises the chunks it wrote
                {
                    const half4_t o = norm4(((const half4_t*) xn_s)[c], ((const half4_t*) nrm.norm_w)[c], rmf);
                    ((half4_t*) xn_s)[c] = o;
                    ((half4_t*) (nrm.xn_out + (size_t) row * H))[c] = o;
                }
            }
            __syncthreads();
        }
        if constexpr (NORM == 1)
        {
            const int nblk = H >> 7, l32 = tid & 31;
            float s2 = 0.0f;
            for (int b0 = 0; b0 < nblk; b0 += 32)
            {
                float v = (b0 + l32 < nblk) ? nrm.ss_part[(size_t) row * nblk + b0 + l32] : 0.0f;
                #pragma unroll
                for (int i = 1; i < 32; i <<= 1) v += xor_lane(v, i);
                s2 += v;
            }
            const float rmf = __frsqrt_rn(s2 / (float) H + nrm.eps);
            for (int c = tid; c < (H >> 2); c += 256)
            {
                const half4_t xv = ((const half4_t*) xg)[c], wv = ((const half4_t*) nrm.norm_w)[c];
                const half4_t o = { f2h((float) xv.x * (float) wv.x * rmf), f2h((float) xv.y * (float) wv.y * rmf),
                                    f2h((float) xv.z * (float) wv.z * rmf), f2h((float) xv.w * (float) wv.w * rmf) };
                ((half4_t*) xn_s)[c] = o;
                ((half4_t*) (nrm.xn_out + (size_t) row * H))[c] = o;
            }
            __syncthreads();
        }
    };

    // scores: gate is [H][E] row-major, so a thread reads 8 consecutive experts of one hidden row as ONE 16-byte load.  E a multiple of 8 with
    // E / 8 dividing 256 (8, 16, 32, 64 ... 512 experts): thread t owns expert chunk t % (E / 8) and the hidden rows t / (E / 8) + j * (256 / (E / 8));
    // 16-byte loads, 8 independent FMA chains per thread, then a fixed-order sum of the per-thread partials through LDS (deterministic).
    // (The first version -- one wave per expert striding over H with 2-byte loads at stride 2E -- took 25.6 us per call at E = 8, H = 4096:
    // 13 % of the Mixtral decode step, profiles/r02_bench_mixtral_kernel_stats.csv.)  Other E: the per-expert loop.
    const int nch = E >> 3;
    if ((E & 7) == 0 && nch <= 64 && (256 % nch) == 0)
    {
        __shared__ float part_s[4][64][8];                                   // [wave][expert chunk][expert in chunk]
        const int ch = tid % nch, rows_per_pass = 256 / nch;
        float acc[8] = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
        // 8 gate rows per thread in flight at a time (one workgroup reads the whole gate matrix: 64 KB at E = 8; with one load per loop trip
        // the kernel was a chain of 16 memory round trips, 17.8 us)
        int h = tid / nch;
        // NORM: the first batch of gate rows is requested by EVERY thread (row indices clamped: threads without a full first batch load valid
        // rows they will not use), then form_xn() -- which contains the workgroup barriers -- runs at ONE call site that every thread reaches, with
        // those loads in flight underneath.  (It used to sit inside the loop below, whose trip condition depends on tid: threads that skipped the
        // loop met the barrier at a different site -- ADVICE round 2.)
        half8_t g[8];
        const bool first = h + 7 * rows_per_pass < H;
        #pragma unroll
        for (int u = 0; u < 8; ++u) g[u] = *((const half8_t*) (gate + (size_t) min(h + u * rows_per_pass, H - 1) * E + ch * 8));
        form_xn();
        if (first)
        {
            for (;;)
            {
                half_t xs[8];
                #pragma unroll
                for (int u = 0; u < 8; ++u) xs[u] = xin(h + u * rows_per_pass);
                #pragma unroll
                for (int u = 0; u < 8; ++u)
                {
                    const float xv = (float) xs[u];
                    #pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(xv, (float) g[u][i], acc[i]);
                }
                h += 8 * rows_per_pass;
                if (!(h + 7 * rows_per_pass < H)) break;
                #pragma unroll
                for (int u = 0; u < 8; ++u) g[u] = *((const half8_t*) (gate + (size_t) (h + u * rows_per_pass) * E + ch * 8));
            }
        }
        for (; h < H; h += rows_per_pass)
        {
            const half8_t g = *((const half8_t*) (gate + (size_t) h * E + ch * 8));
            const float xv = (float) xin(h);
            #pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(xv, (float) g[i], acc[i]);
        }
        // fixed-order reduction: butterflies over the lanes of a wave that own the same chunk (lane % nch), then the four waves in sequence
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1)
        {
            if (d >= nch)                                                    // uniform
            {
                #pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += xor_lane(acc[i], d);
            }
        }
        if (lane < nch)
        {
            #pragma unroll
            for (int i = 0; i < 8; ++i) part_s[wave][lane][i] = acc[i];
        }
        __syncthreads();
        for (int e = tid; e < E; e += 256)
        {
            const int c = e >> 3, i = e & 7;
            const half_t sc = f2h(((part_s[0][c][i] + part_s[1][c][i]) + part_s[2][c][i]) + part_s[3][c][i]);    // the reference materialises fp16 scores and routes on them
            scores[(size_t) row * E + e] = sc;
            logit_s[e] = (float) sc + (bias ? (float) bias[e] : 0.0f);
        }
    }
    else
    {
        form_xn();
        for (int e = wave; e < E; e += 4)
        {
            float acc = 0.0f;
            for (int h = lane; h < H; h += 64) acc = __builtin_fmaf((float) xin(h), (float) gate[(size_t) h * E + e], acc);
            #pragma unroll
            for (int i = 1; i < 64; i <<= 1) acc += xor_lane(acc, i);
            if (lane == 0)
            {
                const half_t s = f2h(acc);
                scores[(size_t) row * E + e] = s;
                logit_s[e] = (float) s + (bias ? (float) bias[e] : 0.0f);
            }
        }
    }
    __syncthreads();
    if (wave != 0) return;

    // top-K by repeated arg-max over the wave (candidates e = lane + 64 j); ties go to the lower expert index
    uint64_t taken_lo = 0;                                 // bit j: candidate j of this lane already selected (E <= 512 -> j < 8)
    float max_logit = -1.0e30f;
    for (int k = 0; k < K; ++k)
    {
        float best = -1.0e30f; int best_e = 0x7fffffff;
        for (int j = 0; lane + 64 * j < E; ++j)
        {
            const int e = lane + 64 * j;
            const float v = logit_s[e];
            if (!((taken_lo >> j) & 1) && (v > best || (v == best && e < best_e))) { best = v; best_e = e; }
        }
        #pragma unroll
        for (int i = 1; i < 64; i <<= 1)
        {
            const float ov = xor_lane(best, i); const int oe = xor_lane(best_e, i);
            if (ov > best || (ov == best && oe < best_e)) { best = ov; best_e = oe; }
        }
        if (k == 0) max_logit = best;
        if ((best_e & 63) == lane) taken_lo |= 1ull << (best_e >> 6);
        if (lane == 0) { sel_logit[k] = best; sel_idx[k] = best_e; }
    }
    // softmax over the selected logits
    float ev = lane < K ? expf(sel_logit[lane] - max_logit) : 0.0f;
    float sum = ev;
    #pragma unroll
    for (int i = 1; i < 64; i <<= 1) sum += xor_lane(sum, i);
    ev /= (sum + 1e-20f);
    if (lane < K)
    {
        if (nrm.per_expert_scale) ev *= __uint_as_float((uint32_t) nrm.per_expert_scale[sel_idx[lane]] << 16);
        topk_indices[(size_t) row * K + lane] = (int64_t) sel_idx[lane];
        topk_weights[(size_t) row * K + lane] = f2h(ev);
        if (gu_slots)
        {
            // slot lists of the indexed gate | up launch over the pointer tables [gate_0..gate_E-1, up_0..up_E-1]: [rows * K gate slots | rows * K up slots]
            gu_slots[(size_t) row * K + lane] = (int64_t) sel_idx[lane];
            gu_slots[(size_t) rows * K + (size_t) row * K + lane] = (int64_t) sel_idx[lane] + E;
        }
    }
}

extern "C" int exl3_routing_std(const void* hidden, const void* gate, const void* bias, void* scores, int64_t* topk_indices, void* topk_weights,
                                int bsz, int hidden_size, int num_experts, int K, void* stream)
{
    return exl3_routing_std_slots(hidden, gate, bias, scores, topk_indices, topk_weights, nullptr, bsz, hidden_size, num_experts, K, stream);
}

// routing_std that also writes gu_slots int64 [2][bsz * K] = [selected experts | selected experts + num_experts]: the slot list of ONE indexed
// exl3_mgemm over the concatenated gate | up pointer tables (saves the two torch index kernels between the router and the launch)
extern "C" int exl3_routing_std_slots(const void* hidden, const void* gate, const void* bias, void* scores, int64_t* topk_indices, void* topk_weights,
                                      int64_t* gu_slots, int bsz, int hidden_size, int num_experts, int K, void* stream)
{
    return exl3_routing_std_scaled(hidden, gate, bias, nullptr, scores, topk_indices, topk_weights, gu_slots, bsz, hidden_size, num_experts, K, stream);
}

// ... with the reference's per_expert_scale argument (bf16 [num_experts], routing.cu:955-1010): the softmax weight of a selected expert is multiplied by it
extern "C" int exl3_routing_std_scaled(const void* hidden, const void* gate, const void* bias, const void* per_expert_scale, void* scores,
                                       int64_t* topk_indices, void* topk_weights, int64_t* gu_slots, int bsz, int hidden_size, int num_experts, int K,
                                       void* stream)
{
    EXL3_CHECK_ARG(hidden && gate && scores && topk_indices && topk_weights, "routing_std: null pointer");
    EXL3_CHECK_ARG(num_experts >= 1 && num_experts <= ROUTING_MAX_EXPERTS, "Too many experts");
    EXL3_CHECK_ARG(K >= 1 && K <= ROUTING_MAX_K, "Too many experts per token");
    EXL3_CHECK_ARG(K <= num_experts, "K cannot exceed number of experts");
    if (bsz == 0) return EXL3_OK;
    routing_std_kernel<0><<<bsz, 256, 0, (hipStream_t) stream>>>((const half_t*) hidden, (const half_t*) gate, (const half_t*) bias, (half_t*) scores,
                                                                    topk_indices, (half_t*) topk_weights, hidden_size, num_experts, K, gu_slots, bsz,
                                                                    RoutingNorm{ nullptr, nullptr, nullptr, 0.0f, (const uint16_t*) per_expert_scale, nullptr });
    return exl3_check_launch("routing_std");
}

// exl3_routing_std_slots on the RMSNorm of the residual stream, formed inside the launch (see RoutingNorm): resid fp16 [bsz][hidden], norm_w fp16
// [hidden], ss_part fp32 [bsz][hidden/128] (exl3_glue_resid), xn_out fp16 [bsz][hidden] receives the normalised rows for the expert launches.
extern "C" int exl3_routing_std_norm(const void* resid, const void* norm_w, const float* ss_part, float eps, void* xn_out, const void* gate, const void* bias,
                                     void* scores, int64_t* topk_indices, void* topk_weights, int64_t* gu_slots, int bsz, int hidden_size,
                                     int num_experts, int K, void* stream)
{
    EXL3_CHECK_ARG(resid && norm_w && ss_part && xn_out && gate && scores && topk_indices && topk_weights, "routing_std_norm: null pointer");
    EXL3_CHECK_ARG(hidden_size % 128 == 0 && hidden_size <= 32768, "routing_std_norm: hidden must be a multiple of 128, at most 32768");
    EXL3_CHECK_ARG(num_experts >= 1 && num_experts <= ROUTING_MAX_EXPERTS, "Too many experts");
    EXL3_CHECK_ARG(K >= 1 && K <= ROUTING_MAX_K, "Too many experts per token");
    EXL3_CHECK_ARG(K <= num_experts, "K cannot exceed number of experts");
    if (bsz == 0) return EXL3_OK;
    const RoutingNorm nrm = { (const half_t*) norm_w, ss_part, (half_t*) xn_out, eps, nullptr, nullptr };
    routing_std_kernel<1><<<bsz, 256, (size_t) hidden_size * 2, (hipStream_t) stream>>>((const half_t*) resid, (const half_t*) gate, (const half_t*) bias,
                                                                                          (half_t*) scores, topk_indices, (half_t*) topk_weights, hidden_size,
                                                                                          num_experts, K, gu_slots, bsz, nrm);
    return exl3_check_launch("routing_std_norm");
}

// exl3_routing_std_norm of the fx pipeline: resid_fx = the residual stream in 64-bit fixed point (int64 [bsz][hidden], value * 2^32); the launch takes
// the row's exact mean square itself and leaves the per-block sums of squares in ss_out fp32 [bsz][hidden/128].
extern "C" int exl3_routing_std_fx(const void* resid_fx, const void* norm_w, float* ss_out, float eps, void* xn_out, const void* gate, const void* bias,
                                   void* scores, int64_t* topk_indices, void* topk_weights, int64_t* gu_slots, int bsz, int hidden_size,
                                   int num_experts, int K, void* stream)
{
    EXL3_CHECK_ARG(resid_fx && norm_w && ss_out && xn_out && gate && scores && topk_indices && topk_weights, "routing_std_fx: null pointer");
    EXL3_CHECK_ARG(hidden_size % 128 == 0 && hidden_size <= 32768, "routing_std_fx: hidden must be a multiple of 128, at most 32768");
    EXL3_CHECK_ARG(num_experts >= 1 && num_experts <= ROUTING_MAX_EXPERTS, "Too many experts");
    EXL3_CHECK_ARG(K >= 1 && K <= ROUTING_MAX_K, "Too many experts per token");
    EXL3_CHECK_ARG(K <= num_experts, "K cannot exceed number of experts");
    if (bsz == 0) return EXL3_OK;
    const RoutingNorm nrm = { (const half_t*) norm_w, nullptr, (half_t*) xn_out, eps, nullptr, ss_out };
    routing_std_kernel<2><<<bsz, 256, (size_t) hidden_size * 2, (hipStream_t) stream>>>((const half_t*) resid_fx, (const half_t*) gate, (const half_t*) bias,
                                                                                       (half_t*) scores, topk_indices, (half_t*) topk_weights, hidden_size,
                                                                                       num_experts, K, gu_slots, bsz, nrm);
    return exl3_check_launch("routing_std_fx");
}

// ------------------------------------------------------------------------------------------------
// exl3_moe (quant/exl3_moe.cu:99-301) without a host round trip: the slot list of the indexed launches is built ON THE DEVICE from expert_count /
// token_sorted, with shapes that depend only on the tensor sizes, so the whole op can be captured in a hipGraph (round 2 built the list on the host
// from expert_count.tolist(): one device -> host sync per call).
//   slot j (j < ns_max = min(E, T) + T / m): up to m consecutive assignments of ONE expert e with 0 < count[e] <= max_rows, or unused:
//     slot_expert[j] = e | -1 (the indexed exl3_mgemm launches skip negative entries), slot_tok[j][r] = token of row r (rows past the chunk repeat its
//     last assignment: computed and dropped), rowmap[p] = j * m + r for assignment p of an accepted expert, -1 otherwise.
// One workgroup; expert e's slots start at the exclusive prefix sum of the chunk counts (reference: the expert tickets of exl3_moe_kernel.cuh:17-283).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void moe_build_slots_kernel(const int64_t* __restrict__ expert_count, const int64_t* __restrict__ token_sorted, int E, int T, int max_rows, int m,
                            int ns_max, int64_t* __restrict__ slot_expert, int64_t* __restrict__ slot_tok, int32_t* __restrict__ rowmap)
{
    __shared__ int cnt_s[ROUTING_MAX_EXPERTS], off_s[ROUTING_MAX_EXPERTS], base_s[ROUTING_MAX_EXPERTS];
    __shared__ int total_s;
    const int tid = threadIdx.x;
    for (int e = tid; e < E; e += 256) cnt_s[e] = (int) expert_count[e];
    for (int p = tid; p < T; p += 256) rowmap[p] = -1;
    __syncthreads();
    if (tid == 0)
    {
        int off = 0, base = 0;
        for (int e = 0; e < E; ++e)
        {
            const int c = cnt_s[e];
            off_s[e] = off; base_s[e] = base;
            off += c;
            if (c > 0 && c <= max_rows) base += (c + m - 1) / m;
        }
        total_s = base;
    }
    __syncthreads();
    const int total = total_s;
    for (int j = total + tid; j < ns_max; j += 256)
    {
        slot_expert[j] = -1;
        for (int r = 0; r < m; ++r) slot_tok[(size_t) j * m + r] = 0;
    }
    for (int e = tid; e < E; e += 256)
    {
        const int c = cnt_s[e];
        if (c <= 0 || c > max_rows) continue;
        const int off = off_s[e];
        for (int ch = 0, j = base_s[e]; ch * m < c && j < ns_max; ++ch, ++j)
        {
            const int n = min(m, c - ch * m);
            slot_expert[j] = e;
            for (int r = 0; r < m; ++r)
            {
                const int p = off + ch * m + min(r, n - 1);
                slot_tok[(size_t) j * m + r] = p < T ? token_sorted[p] : 0;
                if (r < n && p < T) rowmap[p] = j * m + r;
            }
        }
    }
}

// out[t] += sum over the token's assignments p (ascending p: a fixed order, so the result is bit-reproducible; index_add_ / atomics are not)
// of weight[p] * D[rowmap[p]].  One workgroup per token; 4 columns per thread per pass.
__global__ __launch_bounds__(256)
void moe_scatter_kernel(const float* __restrict__ D, const int32_t* __restrict__ rowmap, const int64_t* __restrict__ token_sorted,
                        const half_t* __restrict__ weight_sorted, float* __restrict__ out, int T, int hidden)
{
    __shared__ int list_s[64];
    __shared__ int n_s;
    const int t = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) n_s = 0;
    __syncthreads();
    for (int p = tid; p < T; p += 256)
        if (token_sorted[p] == t && rowmap[p] >= 0) { const int i = atomicAdd(&n_s, 1); if (i < 64) list_s[i] = p; }
    __syncthreads();
    if (n_s > 64)
    {
        // more assignments than the list holds (never with top-k routing, where a token has <= 16): walk all of them in ascending p -- the same fixed
        // order, no truncation, every thread reads the same p (broadcast loads)
        for (int c = tid * 4; c < hidden; c += 1024)
        {
            float4_t acc = *((const float4_t*) (out + (size_t) t * hidden + c));
            for (int p = 0; p < T; ++p)
            {
                if (token_sorted[p] != t || rowmap[p] < 0) continue;
                const float w = (float) weight_sorted[p];
                const float4_t d = *((const float4_t*) (D + (size_t) rowmap[p] * hidden + c));
                acc.x += d.x * w; acc.y += d.y * w; acc.z += d.z * w; acc.w += d.w * w;
            }
            *((float4_t*) (out + (size_t) t * hidden + c)) = acc;
        }
        return;
    }
    const int n = n_s;
    if (n == 0) return;
    if (tid == 0)
        for (int i = 1; i < n; ++i) { const int v = list_s[i]; int k = i - 1; while (k >= 0 && list_s[k] > v) { list_s[k + 1] = list_s[k]; --k; } list_s[k + 1] = v; }
    __syncthreads();
    for (int c = tid * 4; c < hidden; c += 1024)
    {
        float4_t acc = *((const float4_t*) (out + (size_t) t * hidden + c));
        for (int i = 0; i < n; ++i)
        {
            const int p = list_s[i];
            const float w = (float) weight_sorted[p];
            const float4_t d = *((const float4_t*) (D + (size_t) rowmap[p] * hidden + c));
            acc.x += d.x * w; acc.y += d.y * w; acc.z += d.z * w; acc.w += d.w * w;
        }
        *((float4_t*) (out + (size_t) t * hidden + c)) = acc;
    }
}

extern "C" int exl3_moe_build_slots(const int64_t* expert_count, const int64_t* token_sorted, int num_experts, int num_assignments, int max_rows,
                                    int rows_per_slot, int max_slots, int64_t* slot_expert, int64_t* slot_tok, int32_t* rowmap, void* stream)
{
    EXL3_CHECK_ARG(expert_count && token_sorted && slot_expert && slot_tok && rowmap, "moe_build_slots: null pointer");
    EXL3_CHECK_ARG(num_experts >= 1 && num_experts <= ROUTING_MAX_EXPERTS, "Too many experts");
    EXL3_CHECK_ARG(num_assignments >= 1 && max_rows >= 1 && rows_per_slot >= 1 && rows_per_slot <= 16 && max_slots >= 1, "moe_build_slots: bad sizes");
    moe_build_slots_kernel<<<1, 256, 0, (hipStream_t) stream>>>(expert_count, token_sorted, num_experts, num_assignments, max_rows, rows_per_slot,
                                                               max_slots, slot_expert, slot_tok, rowmap);
    return exl3_check_launch("moe_build_slots");
}

extern "C" int exl3_moe_scatter(const float* D, const int32_t* rowmap, const int64_t* token_sorted, const void* weight_sorted, float* out,
                                int bsz, int num_assignments, int hidden, void* stream)
{
    EXL3_CHECK_ARG(D && rowmap && token_sorted && weight_sorted && out, "moe_scatter: null pointer");
    EXL3_CHECK_ARG(hidden % 4 == 0 && bsz >= 1 && num_assignments >= 1, "moe_scatter: bad sizes");
    moe_scatter_kernel<<<bsz, 256, 0, (hipStream_t) stream>>>(D, rowmap, token_sorted, (const half_t*) weight_sorted, out, num_assignments, hidden);
    return exl3_check_launch("moe_scatter");
}
