// A Llama-3.1-8B prefill chunk (llama_path.prefill_chunk, TP = 1, attention core excluded as in bench.py's prefill line) driven through the C ABI from
// C++: per layer RMSNorm -> reconstruct_had_multi_t(q|k|v) -> hgemm_nt -> rope + paged KV-quant on column ranges -> reconstruct_had_t(o) -> hgemm_nt with the
// residual add in its epilogue -> RMSNorm -> reconstruct_had_multi_t(gate|up) -> hgemm_nt -> silu_mul_2d -> reconstruct_had_t(down) -> hgemm_nt (+ residual).
// Same calls, buffers reused across layers like linear.py's scratch cache.  No Python, no torch on the box: a chunk-level A/B of a prefill change costs
// seconds of GPU time (the library autotunes its hipBLASLt algorithms during the first, untimed chunk).  Timing tool only -- parity of every op is
// tests/test_gpu_prefill.py's job; the checksum printed here lets two library builds be compared for equality.
// WRITTEN AT THE END OF ROUND 3 WITHOUT GPU SECONDS LEFT: compiled against include/exl3_hip.h, not yet run.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include -o tools/bin/prefill_chunk_harness tools/experiments/prefill_chunk_harness.hip \
//         -L exllamav3_amd -lexl3_hip -Wl,-rpath,'$ORIGIN/../../exllamav3_amd'
//   tools/bin/prefill_chunk_harness [tokens = 4096] [layers = 32] [timed chunks = 3]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "exl3_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define CE(x) do { int r_ = (x); if (r_ < 0) { fprintf(stderr, "%s:%d exl3 error %d: %s\n", __FILE__, __LINE__, r_, exl3_last_error()); exit(1); } } while (0)

__global__ void fill_hash(uint32_t* p, size_t n, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    {
        uint32_t x = (uint32_t) i * 0x9E3779B1u + seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = x;
    }
}
// fp16 N(0, 1)-ish rows from a hash (Irwin-Hall of four 8-bit uniforms): the chunk's input
__global__ void fill_rows(__half* p, size_t n, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
    {
        uint32_t x = (uint32_t) i * 0x9E3779B1u + seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const float s = (float) ((x & 255) + ((x >> 8) & 255) + ((x >> 16) & 255) + (x >> 24)) - 510.0f;
        p[i] = __float2half(s * (1.0f / 147.8f));
    }
}
__global__ void checksum(const __half* p, size_t n, double* out)
{
    double s = 0;
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) s += (double) __half2float(p[i]) * (double) ((i % 251) + 1);
    atomicAdd(out, s);
}

static uint64_t rng_s = 0x9876543212345678ull;
static double urand() { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return (double) (rng_s >> 11) / 9007199254740992.0; }
static double nrand() { double u = urand() + 1e-12, v = urand(); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }
static __half* dev_half(const std::vector<float>& v)
{
    std::vector<__half> h(v.size());
    for (size_t i = 0; i < v.size(); ++i) h[i] = __float2half(v[i]);
    __half* d; CK(hipMalloc(&d, v.size() * 2)); CK(hipMemcpy(d, h.data(), v.size() * 2, hipMemcpyHostToDevice));
    return d;
}
static std::vector<float> scale_vec(int n, double mag)
{
    std::vector<float> v(n);
    for (int i = 0; i < n; ++i) v[i] = (float) ((urand() < 0.5 ? -1.0 : 1.0) * mag * exp(0.2 * nrand()));
    return v;
}
struct Lin { uint32_t* B; __half* suh; __half* svh; int k, n; };
static uint32_t g_seed = 1;
static Lin make_lin(int k, int n, int K, double out_scale, hipStream_t st)
{
    Lin l; l.k = k; l.n = n;
    const size_t words = (size_t) k * n * K / 32;
    CK(hipMalloc(&l.B, words * 4));
    fill_hash<<<2048, 256, 0, st>>>(l.B, words, g_seed++ * 7919u);
    l.suh = dev_half(scale_vec(k, 1.0)); l.svh = dev_half(scale_vec(n, out_scale / sqrt((double) k)));
    return l;
}
struct Layer { Lin q, k, v, o, g, u, d; __half *norm1, *norm2; };

int main(int argc, char** argv)
{
    const int hidden = 4096, inter = 14336, hq = 32, hkv = 8, hd = 128, K = 4, cb = 2, page = 256, kv_bits = 4;
    const int T = argc > 1 ? atoi(argv[1]) : 4096, n_layers = argc > 2 ? atoi(argv[2]) : 32, reps = argc > 3 ? atoi(argv[3]) : 3;
    const float eps = 1e-5f;
    const int nq = hq * hd, nkv = hkv * hd, nqkv = nq + 2 * nkv;
    CK(hipSetDevice(0));
    CE(exl3_init(0));
    hipStream_t st; CK(hipStreamCreate(&st));
    std::vector<Layer> L(n_layers);
    for (int i = 0; i < n_layers; ++i)
    {
        // out_scale 0.1 on o / down keeps the fp16 residual stream bounded over 32 layers of random weights
        L[i].q = make_lin(hidden, nq, K, 0.5, st); L[i].k = make_lin(hidden, nkv, K, 0.5, st); L[i].v = make_lin(hidden, nkv, K, 0.5, st);
        L[i].o = make_lin(nq, hidden, K, 0.1, st);
        L[i].g = make_lin(hidden, inter, K, 0.5, st); L[i].u = make_lin(hidden, inter, K, 0.5, st); L[i].d = make_lin(inter, hidden, K, 0.1, st);
        std::vector<float> nw(hidden); for (auto& x : nw) x = (float) (1.0 + 0.05 * nrand());
        L[i].norm1 = dev_half(nw); for (auto& x : nw) x = (float) (1.0 + 0.05 * nrand()); L[i].norm2 = dev_half(nw);
    }
    // chunk state + the scratch W^T buffers every layer reuses
    __half *x0, *x, *xn, *y_qkv, *y_gu, *a, *wt_qkv, *w_o, *wt_gu, *w_d;
    CK(hipMalloc(&x0, (size_t) T * hidden * 2)); CK(hipMalloc(&x, (size_t) T * hidden * 2)); CK(hipMalloc(&xn, (size_t) T * hidden * 2));
    CK(hipMalloc(&y_qkv, (size_t) T * nqkv * 2)); CK(hipMalloc(&y_gu, (size_t) T * 2 * inter * 2)); CK(hipMalloc(&a, (size_t) T * inter * 2));
    CK(hipMalloc(&wt_qkv, (size_t) nqkv * hidden * 2)); CK(hipMalloc(&w_o, (size_t) hidden * nq * 2));
    CK(hipMalloc(&wt_gu, (size_t) 2 * inter * hidden * 2)); CK(hipMalloc(&w_d, (size_t) hidden * inter * 2));
    fill_rows<<<2048, 256, 0, st>>>(x0, (size_t) T * hidden, 4242u);
    const int G = nkv / 32, pages = (T + page - 1) / page;
    uint32_t *kc, *vc; __half *ks, *vs; int32_t *seqlens, *block_table; float* inv_freq; double* csum;
    CK(hipMalloc(&kc, (size_t) pages * page * G * kv_bits * 4)); CK(hipMalloc(&vc, (size_t) pages * page * G * kv_bits * 4));
    CK(hipMalloc(&ks, (size_t) pages * page * G * 2)); CK(hipMalloc(&vs, (size_t) pages * page * G * 2));
    CK(hipMalloc(&seqlens, 4)); CK(hipMalloc(&block_table, pages * 4)); CK(hipMalloc(&inv_freq, 64 * 4)); CK(hipMalloc(&csum, 8));
    {
        float f[64]; for (int i = 0; i < 64; ++i) f[i] = (float) (1.0 / pow(500000.0, (2.0 * i) / hd));
        CK(hipMemcpy(inv_freq, f, sizeof(f), hipMemcpyHostToDevice));
        int32_t z = 0; CK(hipMemcpy(seqlens, &z, 4, hipMemcpyHostToDevice));
        std::vector<int32_t> bt(pages); for (int i = 0; i < pages; ++i) bt[i] = i;
        CK(hipMemcpy(block_table, bt.data(), pages * 4, hipMemcpyHostToDevice));
    }
    CK(hipStreamSynchronize(st));

    auto chunk = [&] ()
    {
        CK(hipMemcpyAsync(x, x0, (size_t) T * hidden * 2, hipMemcpyDeviceToDevice, st));
        for (int i = 0; i < n_layers; ++i)
        {
            Layer& l = L[i];
            CE(exl3_rms_norm(x, l.norm1, xn, nullptr, eps, 0.0f, 1.0f, T, hidden, 0, 0, 0, 0, 0, st));
            {
                const void* Bs[3] = { l.q.B, l.k.B, l.v.B }; const void* su[3] = { l.q.suh, l.k.suh, l.v.suh }; const void* sv[3] = { l.q.svh, l.k.svh, l.v.svh };
                int tn[3] = { nq / 16, nkv / 16, nkv / 16 };
                CE(exl3_reconstruct_had_multi_t(wt_qkv, hidden, Bs, su, sv, tn, 3, hidden / 16, K, cb, st));
                CE(exl3_hgemm_nt(xn, wt_qkv, y_qkv, T, hidden, nqkv, hidden, nqkv, 0, 0, st));
            }
            CE(exl3_rope_strided(y_qkv, y_qkv + nq, inv_freq, 1, T, hq, hkv, nqkv, nqkv, 0u, nullptr, nullptr, 1.0f, st));
            CE(exl3_quant_cache_paged_strided(y_qkv + nq, kc, ks, y_qkv + nq + nkv, vc, vs, seqlens, block_table, 1, pages, page, T, nkv, kv_bits, kv_bits, nqkv, nqkv, st));
            CE(exl3_reconstruct_had_t(w_o, nq, l.o.B, l.o.suh, l.o.svh, nq / 16, hidden / 16, K, cb, 0, hidden, st));
            CE(exl3_hgemm_nt_lda(y_qkv, nqkv, w_o, x, T, nq, hidden, nq, hidden, 0, 1, st));             // x += q @ W_o (residual add in the epilogue)
            CE(exl3_rms_norm(x, l.norm2, xn, nullptr, eps, 0.0f, 1.0f, T, hidden, 0, 0, 0, 0, 0, st));
            {
                const void* Bs[2] = { l.g.B, l.u.B }; const void* su[2] = { l.g.suh, l.u.suh }; const void* sv[2] = { l.g.svh, l.u.svh };
                int tn[2] = { inter / 16, inter / 16 };
                CE(exl3_reconstruct_had_multi_t(wt_gu, hidden, Bs, su, sv, tn, 2, hidden / 16, K, cb, st));
                CE(exl3_hgemm_nt(xn, wt_gu, y_gu, T, hidden, 2 * inter, hidden, 2 * inter, 0, 0, st));
            }
            CE(exl3_silu_mul_2d(y_gu, y_gu + inter, a, T, inter, 2 * inter, 2 * inter, st));
            CE(exl3_reconstruct_had_t(w_d, inter, l.d.B, l.d.suh, l.d.svh, inter / 16, hidden / 16, K, cb, 0, hidden, st));
            CE(exl3_hgemm_nt(a, w_d, x, T, inter, hidden, inter, hidden, 0, 1, st));
        }
    };

    chunk();                                            // autotune + warm-up
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    printf("{\"model\": \"llama-3.1-8b shapes, %d layers, EXL3 4.0 bpw mul1, prefill chunk of %d tokens via the C ABI\", \"ms_per_chunk\": [", n_layers, T);
    for (int r = 0; r < reps; ++r)
    {
        CK(hipEventRecord(e0, st));
        chunk();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double) ms);
        printf("%s%.3f", r ? ", " : "", ms);
    }
    CK(hipMemsetAsync(csum, 0, 8, st));
    checksum<<<256, 256, 0, st>>>(x, (size_t) T * hidden, csum);
    double h = 0; CK(hipMemcpyAsync(&h, csum, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    const double flops = 2.0 * ((double) hidden * nqkv + (double) nq * hidden + 2.0 * (double) hidden * inter + (double) inter * hidden) * n_layers * (double) T;
    printf("], \"tok_s\": %.1f, \"tflops_on_the_linears\": %.1f, \"frac_of_2500\": %.4f, \"residual_checksum\": %.6e}\n",
           T / (best * 1e-3), flops / (best * 1e-3) / 1e12, flops / (best * 1e-3) / 2.5e15, h);
    return 0;
}
