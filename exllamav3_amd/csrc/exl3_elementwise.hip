// RMSNorm (+ fused residual modes), silu*mul, add.  HBM-bound elementwise/reduction kernels for gfx950:
// 8-16 B per lane vector accesses, wave64 shuffle reductions.
// reference: exllamav3_ext/norm.cu:155-299 (rms_norm_kernel), activation.cu (silu_mul), add.cu.
#include "exl3_common.cuh"
#include "exl3_api_internal.h"

__device__ __forceinline__ float4_t load4(const void* p, int64_t idx4, bool fp32)
{
    if (fp32) return ((const float4_t*) p)[idx4];
    half4_t h = ((const half4_t*) p)[idx4];
    return float4_t{ (float) h.x, (float) h.y, (float) h.z, (float) h.w };
}

__device__ __forceinline__ void store4(void* p, int64_t idx4, float4_t v, bool fp32)
{
    if (fp32) ((float4_t*) p)[idx4] = v;
    else ((half4_t*) p)[idx4] = half4_t{ f2h(v.x), f2h(v.y), f2h(v.z), f2h(v.w) };
}

__device__ __forceinline__ float4_t load_w4(const void* w, int idx4, bool bf16)
{
    uint2_t raw = ((const uint2_t*) w)[idx4];
    if (bf16)
        return float4_t{ __uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u),
                         __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xffff0000u) };
    half2_t a = u32_as_half2(raw.x), b = u32_as_half2(raw.y);
    return float4_t{ (float) a.x, (float) a.y, (float) b.x, (float) b.y };
}

__device__ __forceinline__ float block_sum(float v, float* red)
{
    #pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += xor_lane(v, o);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int nw = blockDim.x >> 6;
    if (nw == 1) return v;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = lane < nw ? red[lane] : 0.0f;
    #pragma unroll
    for (int o = 8; o > 0; o >>= 1) t += xor_lane(t, o);
    return __shfl(t, 0, 64);
}

// MODE 0: y = norm(x) * w; 1: y += norm(x) * w; 2: r += x (rounded to r's dtype), y = norm(r) * w
template <int MODE>
__global__ __launch_bounds__(1024)
void rms_norm_kernel(const void* __restrict__ x, const void* __restrict__ w, void* __restrict__ y, void* __restrict__ r,
                     float eps, float constant_bias, float constant_scale, int dim,
                     int x_fp32, int y_fp32, int r_fp32, int w_bf16)
{
    __shared__ float red[16];
    const int row = blockIdx.x;
    const int cols4 = dim >> 2;
    const int64_t base4 = (int64_t) row * cols4;
    float sum = 0.0f;
    for (int c = threadIdx.x; c < cols4; c += blockDim.x)
    {
        float4_t v = load4(x, base4 + c, x_fp32);
        if constexpr (MODE == 2)
        {
            float4_t rv = load4(r, base4 + c, r_fp32);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            store4(r, base4 + c, v, r_fp32);
            if (!r_fp32) { v.x = (float) f2h(v.x); v.y = (float) f2h(v.y); v.z = (float) f2h(v.z); v.w = (float) f2h(v.w); }  // norm sees r's rounding (norm.cu:206-213)
        }
        sum = __builtin_fmaf(v.x, v.x, sum); sum = __builtin_fmaf(v.y, v.y, sum);
        sum = __builtin_fmaf(v.z, v.z, sum); sum = __builtin_fmaf(v.w, v.w, sum);
    }
    sum = block_sum(sum, red);
    const float rmf = __frsqrt_rn(sum / (float) dim + eps) * constant_scale;
    for (int c = threadIdx.x; c < cols4; c += blockDim.x)
    {
        float4_t v = (MODE == 2) ? load4(r, base4 + c, r_fp32) : load4(x, base4 + c, x_fp32);
        if (w)
        {
            float4_t wv = load_w4(w, c, w_bf16);
            if (constant_bias != 0.0f) { wv.x += constant_bias; wv.y += constant_bias; wv.z += constant_bias; wv.w += constant_bias; }
            v.x = v.x * wv.x * rmf; v.y = v.y * wv.y * rmf; v.z = v.z * wv.z * rmf; v.w = v.w * wv.w * rmf;
        }
        else { v.x *= rmf; v.y *= rmf; v.z *= rmf; v.w *= rmf; }
        if constexpr (MODE == 1)
        {
            float4_t o = load4(y, base4 + c, y_fp32);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        store4(y, base4 + c, v, y_fp32);
    }
}

// Many-row variant (prefill: thousands of rows of 2048 / 4096 / 8192 halves): one WAVE per row.  The row lives in registers (NV 16-byte
// pieces per lane), so x is read once, the sum of squares is a DPP butterfly, and there is no workgroup barrier; 4 rows per 256-thread
// workgroup.  The block-per-row kernel above spends a 4096-wide row on 1024 threads x 8 bytes, two barriers and a second pass over x:
// 25.6 us for 4096 x 4096 fp16 (2.6 TB/s, profiles/r01_prefill_chunk_kernel_stats.csv).  Same per-element arithmetic (v * w * rmf in
// fp32, one rounding); the sum of squares is accumulated in another order (fp32), inside the op's 1e-3 tolerance.
template <int MODE, int NV>
__global__ __launch_bounds__(256)
void rms_norm_rows_kernel(const half_t* __restrict__ x, const void* __restrict__ w, half_t* __restrict__ y, half_t* __restrict__ r,
                          float eps, float constant_bias, float constant_scale, int rows, int w_bf16)
{
    constexpr int DIM = NV * 512;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4)
    {
        half8_t v[NV];
        const half8_t* xr = (const half8_t*) (x + (size_t) row * DIM);
        #pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = xr[i * 64 + lane];
        if constexpr (MODE == 2)
        {
            half8_t* rr = (half8_t*) (r + (size_t) row * DIM);
            #pragma unroll
            for (int i = 0; i < NV; ++i)
            {
                const half8_t rv = rr[i * 64 + lane];
                #pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = f2h((float) v[i][j] + (float) rv[j]);      // r += x rounded to r's dtype; the norm sees that value
                rr[i * 64 + lane] = v[i];
            }
        }
        float sum = 0.0f;
        #pragma unroll
        for (int i = 0; i < NV; ++i)
        {
            #pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = (float) v[i][j]; sum = __builtin_fmaf(f, f, sum); }
        }
        #pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += xor_lane(sum, o);
        const float rmf = __frsqrt_rn(sum / (float) DIM + eps) * constant_scale;
        half8_t* yr = (half8_t*) (y + (size_t) row * DIM);
        #pragma unroll
        for (int i = 0; i < NV; ++i)
        {
            half8_t o8;
            if (w)
            {
                const float4_t w0 = load_w4(w, 2 * (i * 64 + lane), w_bf16), w1 = load_w4(w, 2 * (i * 64 + lane) + 1, w_bf16);
                const float wf[8] = { w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w };
                #pragma unroll
                for (int j = 0; j < 8; ++j) o8[j] = f2h((float) v[i][j] * (wf[j] + constant_bias) * rmf);
            }
            else
            {
                #pragma unroll
                for (int j = 0; j < 8; ++j) o8[j] = f2h((float) v[i][j] * rmf);
            }
            yr[i * 64 + lane] = o8;
        }
    }
}

extern "C" int exl3_rms_norm(const void* x, const void* w, void* y, void* r, float eps, float constant_bias, float constant_scale,
                             int rows, int dim, int x_fp32, int y_fp32, int r_fp32, int w_bf16, int mode, void* stream)
{
    EXL3_CHECK_ARG(x && y, "rms_norm: null pointer");
    EXL3_CHECK_ARG(dim % 4 == 0 && dim > 0, "rms_norm: last dimension must be divisible by 4");
    EXL3_CHECK_ARG(mode >= 0 && mode <= 2, "rms_norm: bad mode");
    EXL3_CHECK_ARG(mode != 2 || r, "rms_norm: res_mode RES_IN requires residual tensor");
    if (rows == 0) return EXL3_OK;
    hipStream_t st = (hipStream_t) stream;
    if (rows >= 64 && !x_fp32 && !y_fp32 && (mode != 2 || !r_fp32) && mode != 1 && (dim == 2048 || dim == 4096 || dim == 8192))
    {
        const unsigned grid = (unsigned) ((rows + 3) / 4);
        #define RR(M, NV) rms_norm_rows_kernel<M, NV><<<dim3(grid), dim3(256), 0, st>>>((const half_t*) x, w, (half_t*) y, (half_t*) r, eps, \
                                                                                     constant_bias, constant_scale, rows, w_bf16)
        if (mode == 0) { if (dim == 2048) RR(0, 4); else if (dim == 4096) RR(0, 8); else RR(0, 16); }
        else           { if (dim == 2048) RR(2, 4); else if (dim == 4096) RR(2, 8); else RR(2, 16); }
        #undef RR
        return exl3_check_launch("rms_norm");
    }
    int threads = ((dim / 4 + 63) / 64) * 64;
    if (threads > 1024) threads = 1024;
    #define RN(M) rms_norm_kernel<M><<<dim3(rows), dim3(threads), 0, st>>>(x, w, y, r, eps, constant_bias, \
                                     constant_scale, dim, x_fp32, y_fp32, r_fp32, w_bf16)
    if (mode == 0) RN(0); else if (mode == 1) RN(1); else RN(2);
    #undef RN
    return exl3_check_launch("rms_norm");
}

// y = silu(g) * u      (activation.cu: _silu = x / (1 + exp(-x)), fast-math exp)
__global__ __launch_bounds__(256)
void silu_mul_kernel(const void* __restrict__ g, const void* __restrict__ u, half_t* __restrict__ y, int64_t n4, int in_fp32)
{
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4_t gv = load4(g, i, in_fp32), uv = load4(u, i, in_fp32);
    float4_t o;
    o.x = gv.x / (1.0f + __expf(-gv.x)) * uv.x;
    o.y = gv.y / (1.0f + __expf(-gv.y)) * uv.y;
    o.z = gv.z / (1.0f + __expf(-gv.z)) * uv.z;
    o.w = gv.w / (1.0f + __expf(-gv.w)) * uv.w;
    store4(y, i, o, false);
}

extern "C" int exl3_silu_mul(const void* g, const void* u, void* y, int64_t numel, int in_fp32, void* stream)
{
    EXL3_CHECK_ARG(g && u && y, "silu_mul: null pointer");
    EXL3_CHECK_ARG(numel % 4 == 0, "silu_mul: numel must be divisible by 4");
    if (numel == 0) return EXL3_OK;
    int64_t n4 = numel / 4;
    silu_mul_kernel<<<dim3((unsigned) ((n4 + 255) / 256)), dim3(256), 0, (hipStream_t) stream>>>(g, u, (half_t*) y, n4, in_fp32);
    return exl3_check_launch("silu_mul");
}

// y = act(g) * u for the activations of the reference's gated MLP (activation.cu: silu_mul, gelu_mul, relu2_mul, silu_oai_mul; kernels
// activation_kernels.cuh:142-254).  act: 0 SiLU, 1 GELU (tanh form), 2 relu(x)^2, 3 relu, 4 gpt-oss clamped swiglu ((u + 1) * g * sigmoid(1.702 g), gate clamped
// from above, up symmetrically, before the activation).  act_limit != 0 (acts 0..3): u clamped to [-limit, limit], act(g) to <= limit.  fp32 arithmetic, one
// rounding to fp16 (clamped to the finite range) at the end.
__device__ __forceinline__ float act_mul_one(float g, float u, int act, float limit)
{
    if (act == 4)
    {
        if (limit != 0.0f) { g = fminf(g, limit); u = fminf(fmaxf(u, -limit), limit); }
        return (u + 1.0f) * (g / (1.0f + __expf(-1.702f * g)));
    }
    float a;
    if (act == 0) a = g / (1.0f + __expf(-g));
    else if (act == 1) a = 0.5f * g * (1.0f + tanhf(0.797884560803f * (g + 0.044715f * g * g * g)));
    else if (act == 2) { a = fmaxf(0.0f, g); a = a * a; }
    else a = fmaxf(0.0f, g);
    if (limit != 0.0f) { u = fminf(fmaxf(u, -limit), limit); a = fminf(a, limit); }
    return fminf(fmaxf(a * u, -65504.0f), 65504.0f);
}

__global__ __launch_bounds__(256)
void act_mul_kernel(const void* __restrict__ g, const void* __restrict__ u, half_t* __restrict__ y, int64_t n4, int in_fp32, int act, float limit)
{
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4_t gv = load4(g, i, in_fp32), uv = load4(u, i, in_fp32);
    float4_t o = { act_mul_one(gv.x, uv.x, act, limit), act_mul_one(gv.y, uv.y, act, limit), act_mul_one(gv.z, uv.z, act, limit), act_mul_one(gv.w, uv.w, act, limit) };
    store4(y, i, o, false);
}

extern "C" int exl3_act_mul(const void* g, const void* u, void* y, int64_t numel, int in_fp32, int act, float act_limit, void* stream)
{
    EXL3_CHECK_ARG(g && u && y, "act_mul: null pointer");
    EXL3_CHECK_ARG(numel % 4 == 0, "act_mul: numel must be divisible by 4");
    EXL3_CHECK_ARG(act >= 0 && act <= 4, "act_mul: activation must be 0 (silu), 1 (gelu), 2 (relu2), 3 (relu) or 4 (silu_oai)");
    EXL3_CHECK_ARG(act_limit >= 0.0f, "act_mul: act_limit must be >= 0");
    if (numel == 0) return EXL3_OK;
    int64_t n4 = numel / 4;
    act_mul_kernel<<<dim3((unsigned) ((n4 + 255) / 256)), dim3(256), 0, (hipStream_t) stream>>>(g, u, (half_t*) y, n4, in_fp32, act, act_limit);
    return exl3_check_launch("act_mul");
}

// ------------------------------------------------------------------------------------------------
// Attention output gates (activation.cu:526-660, kernels activation_kernels.cuh:132-139, 300-367): x *= sigmoid(y) elementwise or with one gate per
// `bcast` consecutive values (one per head), or x *= softplus(y) broadcast.  The sigmoid form is fp16 arithmetic end to end like the reference's
// (exp of -y rounded to fp16, 1 + e in fp16, reciprocal rounded to fp16, product in fp16); the softplus gate is evaluated in fp32 and only the product
// rounds (max(y, 0) + log1p(exp(-|y|))).  4 halves per thread; bcast % 4 == 0 or bcast == 0 (elementwise).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ half_t sigmoid_h(half_t y)
{
    const half_t e = f2h(__expf(-(float) y));
    const half_t sum = (half_t) 1.0f + e;
    return f2h(1.0f / (float) sum);
}

template <bool SOFTPLUS>
__global__ __launch_bounds__(256)
void mul_gate_kernel(half_t* __restrict__ x, const half_t* __restrict__ y, int64_t n4, int bcast)
{
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    half4_t xv = ((half4_t*) x)[i];
    if (bcast == 0)
    {
        const half4_t yv = ((const half4_t*) y)[i];
        xv = half4_t{ xv.x * sigmoid_h(yv.x), xv.y * sigmoid_h(yv.y), xv.z * sigmoid_h(yv.z), xv.w * sigmoid_h(yv.w) };
    }
    else
    {
        const half_t g = y[(i * 4) / bcast];
        if constexpr (SOFTPLUS)
        {
            const float gf = (float) g, sp = fmaxf(gf, 0.0f) + log1pf(__expf(-fabsf(gf)));
            xv = half4_t{ f2h((float) xv.x * sp), f2h((float) xv.y * sp), f2h((float) xv.z * sp), f2h((float) xv.w * sp) };
        }
        else
        {
            const half_t sg = sigmoid_h(g);
            xv = half4_t{ xv.x * sg, xv.y * sg, xv.z * sg, xv.w * sg };
        }
    }
    ((half4_t*) x)[i] = xv;
}

extern "C" int exl3_mul_gate(void* x, const void* y, int64_t numel, int bcast, int softplus, void* stream)
{
    EXL3_CHECK_ARG(x && y, "mul_gate: null pointer");
    EXL3_CHECK_ARG(numel % 4 == 0 && bcast >= 0 && bcast % 4 == 0, "mul_gate: numel and the broadcast width must be divisible by 4");
    EXL3_CHECK_ARG(!softplus || bcast > 0, "mul_gate: the softplus gate is the broadcast (one gate per head) form");
    if (numel == 0) return EXL3_OK;
    const int64_t n4 = numel / 4;
    const dim3 grid((unsigned) ((n4 + 255) / 256));
    if (softplus) mul_gate_kernel<true><<<grid, 256, 0, (hipStream_t) stream>>>((half_t*) x, (const half_t*) y, n4, bcast);
    else          mul_gate_kernel<false><<<grid, 256, 0, (hipStream_t) stream>>>((half_t*) x, (const half_t*) y, n4, bcast);
    return exl3_check_launch("mul_gate");
}

// z += x * sigmoid(y)  (activation.cu:480-524, activation_kernels.cuh:278-297: fp32, one gate per `dim` values) and
// z += x * sigmoid(y . w) (activation.cu:662-714, activation_kernels.cuh:369-411: the gate is the row's projection onto w; a gate below 1e-8 leaves z
// untouched): the shared-expert merge of the sparse-MoE block.  One workgroup per row for the projection form.
__global__ __launch_bounds__(256)
void add_sigmoid_gate_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ z, int64_t numel, int dim)
{
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numel) return;
    z[i] += x[i] * (1.0f / (1.0f + __expf(-y[i / dim])));
}

__global__ __launch_bounds__(256)
void add_sigmoid_gate_proj_kernel(const float* __restrict__ x, const half_t* __restrict__ y, float* __restrict__ z, const half_t* __restrict__ w, int dim)
{
    __shared__ float red[16];
    const int b = blockIdx.x;
    float yw = 0.0f;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) yw += (float) w[i] * (float) y[(int64_t) b * dim + i];
    yw = block_sum(yw, red);
    const float g = 1.0f / (1.0f + __expf(-yw));
    if (g < 1e-8f) return;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) z[(int64_t) b * dim + i] += x[(int64_t) b * dim + i] * g;
}

extern "C" int exl3_add_sigmoid_gate(const float* x, const float* y, float* z, int64_t numel, int dim, void* stream)
{
    EXL3_CHECK_ARG(x && y && z && dim > 0 && numel % dim == 0, "add_sigmoid_gate: null pointer / bad sizes");
    if (numel == 0) return EXL3_OK;
    add_sigmoid_gate_kernel<<<dim3((unsigned) ((numel + 255) / 256)), 256, 0, (hipStream_t) stream>>>(x, y, z, numel, dim);
    return exl3_check_launch("add_sigmoid_gate");
}

extern "C" int exl3_add_sigmoid_gate_proj(const float* x, const void* y, float* z, const void* w, int rows, int dim, void* stream)
{
    EXL3_CHECK_ARG(x && y && z && w && dim > 0, "add_sigmoid_gate_proj: null pointer / bad sizes");
    if (rows == 0) return EXL3_OK;
    add_sigmoid_gate_proj_kernel<<<dim3(rows), 256, 0, (hipStream_t) stream>>>(x, (const half_t*) y, z, (const half_t*) w, dim);
    return exl3_check_launch("add_sigmoid_gate_proj");
}

// fp16 paged cache append (generator/cache.cu:140-240 paged_kv_cache_update): k / v [bsz][s][heads][dim] -> row cache_seqlens[b] + t of the sequence's
// pages (page size 256); 16 bytes per thread
__global__ __launch_bounds__(256)
void paged_kv_update_kernel(const half8_t* __restrict__ k, const half8_t* __restrict__ v, half8_t* __restrict__ kc, half8_t* __restrict__ vc,
                            const int32_t* __restrict__ block_table, const int32_t* __restrict__ cache_seqlens, int S, int row8, int pages_per_seq, int64_t total8)
{
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int64_t tok = i / row8, c = i - tok * row8;
    const int64_t b = tok / S, t = tok - b * S;
    const int64_t pos = (int64_t) cache_seqlens[b] + t;
    const int64_t page = block_table[b * pages_per_seq + (pos >> 8)];
    const int64_t dst = (page * 256 + (pos & 255)) * row8 + c;
    kc[dst] = k[i]; vc[dst] = v[i];
}

extern "C" int exl3_paged_kv_cache_update(const void* k, const void* v, void* k_cache, void* v_cache, const int32_t* block_table, const int32_t* cache_seqlens,
                                          int bsz, int seq_len, int heads, int dim, int pages_per_seq, void* stream)
{
    EXL3_CHECK_ARG(k && v && k_cache && v_cache && block_table && cache_seqlens, "paged_kv_cache_update: null pointer");
    EXL3_CHECK_ARG(dim % 8 == 0, "dim must be divisible by 8");
    const int64_t total8 = (int64_t) bsz * seq_len * heads * (dim / 8);
    if (total8 == 0) return EXL3_OK;
    paged_kv_update_kernel<<<dim3((unsigned) ((total8 + 255) / 256)), 256, 0, (hipStream_t) stream>>>((const half8_t*) k, (const half8_t*) v, (half8_t*) k_cache, (half8_t*) v_cache,
                                                                                                     block_table, cache_seqlens, seq_len, heads * (dim / 8), pages_per_seq, total8);
    return exl3_check_launch("paged_kv_cache_update");
}

// [.., heads, (q: head_dim, g: head_dim)] -> contiguous q and g (activation.cu:716-785): 16 bytes per thread
__global__ __launch_bounds__(256)
void deinterleave_qg_kernel(const half8_t* __restrict__ qg, half8_t* __restrict__ q, half8_t* __restrict__ g, int hd8, int64_t n8)
{
    const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    const int64_t h = i / hd8, d = i - h * hd8;
    const int64_t src = h * 2 * hd8 + d;
    q[i] = qg[src];
    g[i] = qg[src + hd8];
}

extern "C" int exl3_deinterleave_qg(const void* qg, void* q, void* g, int64_t heads_total, int head_dim, void* stream)
{
    EXL3_CHECK_ARG(qg && q && g, "deinterleave_qg: null pointer");
    EXL3_CHECK_ARG(head_dim > 0 && head_dim % 8 == 0, "deinterleave_qg: head_dim must be divisible by 8");
    const int64_t n8 = heads_total * (head_dim / 8);
    if (n8 == 0) return EXL3_OK;
    deinterleave_qg_kernel<<<dim3((unsigned) ((n8 + 255) / 256)), 256, 0, (hipStream_t) stream>>>((const half8_t*) qg, (half8_t*) q, (half8_t*) g, head_dim / 8, n8);
    return exl3_check_launch("deinterleave_qg");
}

// row-strided variant (fp16): g and u are column ranges of wider matrices (the fused gate|up prefill GEMM writes one [rows][2*cols] output);
// 8 halves per thread (16-byte accesses)
__global__ __launch_bounds__(256)
void silu_mul_2d_kernel(const half_t* __restrict__ g, const half_t* __restrict__ u, half_t* __restrict__ y, int64_t rows, int cols8,
                        int64_t ld_g, int64_t ld_u)
{
    // grid = (column chunks of 256 x 8 halves, row groups); a workgroup walks its rows with a stride of gridDim.y: no per-thread division,
    // two rows in flight per thread
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols8) return;
    for (int64_t r = blockIdx.y; r < rows; r += 2 * (int64_t) gridDim.y)
    {
        const int64_t r1 = r + gridDim.y;
        const bool two = r1 < rows;
        const half8_t g0 = ((const half8_t*) (g + r * ld_g))[c], u0 = ((const half8_t*) (u + r * ld_u))[c];
        half8_t g1 = g0, u1 = u0;
        if (two) { g1 = ((const half8_t*) (g + r1 * ld_g))[c]; u1 = ((const half8_t*) (u + r1 * ld_u))[c]; }
        half8_t o0, o1;
        #pragma unroll
        for (int j = 0; j < 8; ++j)
        {
            const float a = (float) g0[j], b = (float) g1[j];
            o0[j] = f2h(a / (1.0f + __expf(-a)) * (float) u0[j]);
            o1[j] = f2h(b / (1.0f + __expf(-b)) * (float) u1[j]);
        }
        ((half8_t*) (y + r * (int64_t) cols8 * 8))[c] = o0;
        if (two) ((half8_t*) (y + r1 * (int64_t) cols8 * 8))[c] = o1;
    }
}

extern "C" int exl3_silu_mul_2d(const void* g, const void* u, void* y, int64_t rows, int64_t cols, int64_t ld_g, int64_t ld_u, void* stream)
{
    EXL3_CHECK_ARG(g && u && y, "silu_mul_2d: null pointer");
    EXL3_CHECK_ARG(cols % 8 == 0 && ld_g % 8 == 0 && ld_u % 8 == 0 && ld_g >= cols && ld_u >= cols, "silu_mul_2d: columns and row strides must be multiples of 8");
    if (rows == 0 || cols == 0) return EXL3_OK;
    const int cols8 = (int) (cols / 8);
    const unsigned gx = (unsigned) ((cols8 + 255) / 256);
    unsigned gy = (unsigned) (rows < 2048 ? rows : 2048);                  // >= 8 workgroups per CU in flight for wide matrices
    silu_mul_2d_kernel<<<dim3(gx, gy), dim3(256), 0, (hipStream_t) stream>>>((const half_t*) g, (const half_t*) u, (half_t*) y, rows, cols8, ld_g, ld_u);
    return exl3_check_launch("silu_mul_2d");
}

__global__ __launch_bounds__(256)
void add_kernel(void* __restrict__ x, const void* __restrict__ y, int64_t n4, int x_fp32, int y_fp32)
{
    int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4_t a = load4(x, i, x_fp32), b = load4(y, i, y_fp32);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    store4(x, i, a, x_fp32);
}

extern "C" int exl3_add(void* x, const void* y, int64_t numel, int x_fp32, int y_fp32, void* stream)
{
    EXL3_CHECK_ARG(x && y, "add: null pointer");
    EXL3_CHECK_ARG(numel % 4 == 0, "add: numel must be divisible by 4");
    if (numel == 0) return EXL3_OK;
    int64_t n4 = numel / 4;
    add_kernel<<<dim3((unsigned) ((n4 + 255) / 256)), dim3(256), 0, (hipStream_t) stream>>>(x, y, n4, x_fp32, y_fp32);
    return exl3_check_launch("add");
}


// ------------------------------------------------------------------------------------------------
// softcap(x, y, scale): y = scale * tanh(x / scale), fp32 math, fp16 or fp32 tensors, in place allowed (softcap.cu:11-100; applied by
// Linear.forward after the quantized linear for models with logit / attention soft-capping, modules/linear.py:598-599).
// HBM-bound element-wise pass: 16 bytes per lane.
// ------------------------------------------------------------------------------------------------
template <bool FP32>
__global__ __launch_bounds__(256)
void softcap_kernel(const void* __restrict__ x, void* __restrict__ y, int64_t numel, float scale)
{
    constexpr int V = FP32 ? 4 : 8;
    const int64_t i0 = ((int64_t) blockIdx.x * 256 + threadIdx.x) * V;
    if (i0 >= numel) return;
    const float inv = 1.0f / scale;
    if (i0 + V <= numel)
    {
        if constexpr (FP32)
        {
            float4_t v = *((const float4_t*) ((const float*) x + i0));
            v = float4_t{ tanhf(v.x / scale) * scale, tanhf(v.y / scale) * scale, tanhf(v.z / scale) * scale, tanhf(v.w / scale) * scale };
            *((float4_t*) ((float*) y + i0)) = v;
        }
        else
        {
            half8_t v = *((const half8_t*) ((const half_t*) x + i0));
            half8_t o;
            #pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = f2h(tanhf((float) v[i] / scale) * scale);
            *((half8_t*) ((half_t*) y + i0)) = o;
        }
    }
    else
    {
        for (int64_t i = i0; i < numel; ++i)
        {
            if constexpr (FP32) ((float*) y)[i] = tanhf(((const float*) x)[i] / scale) * scale;
            else ((half_t*) y)[i] = f2h(tanhf((float) ((const half_t*) x)[i] / scale) * scale);
        }
    }
    (void) inv;
}

extern "C" int exl3_softcap(const void* x, void* y, int64_t numel, float scale, int is_fp32, void* stream)
{
    EXL3_CHECK_ARG(x && y && numel >= 0, "softcap: null pointer");
    EXL3_CHECK_ARG(scale != 0.0f, "softcap: scale must be non-zero");
    if (numel == 0) return EXL3_OK;
    const int V = is_fp32 ? 4 : 8;
    const unsigned blocks = (unsigned) ((numel + (int64_t) 256 * V - 1) / ((int64_t) 256 * V));
    if (is_fp32) softcap_kernel<true><<<blocks, 256, 0, (hipStream_t) stream>>>(x, y, numel, scale);
    else softcap_kernel<false><<<blocks, 256, 0, (hipStream_t) stream>>>(x, y, numel, scale);
    return exl3_check_launch("softcap");
}
