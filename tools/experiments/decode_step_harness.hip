// A whole Llama-3.1-8B batch-1 decode step (the fixed-point-residual pipeline of exllamav3_amd/llama_path.py: decode_step_fx) driven through the C ABI from
// C++ -- no Python, no torch on the box: synthetic EXL3 4.0 bpw mul1 tensors filled on the device, one hipGraph per step, tok/s from graph replays.  A run
// costs a few seconds of GPU time, so same-box A/B alternations of a kernel variant are cheap.  Variant B here: the MLP block as ONE launch
// (exl3_mlp1.hip, -DM1_TAGGED) instead of exl3_gemv_ex_fx(gate|up) -> exl3_glue_act_rs -> exl3_gemv_ex(down); the logits of the two variants are
// compared.  Attention core excluded, as in bench.py's headline line: o_proj reads q after rope.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-rdc -DM1_TAGGED -I include -I exllamav3_amd/csrc -o tools/bin/decode_step_harness \
//         tools/experiments/decode_step_harness.hip tools/experiments/exl3_mlp1.hip -L exllamav3_amd -lexl3_hip -Wl,-rpath,'$ORIGIN/../../exllamav3_amd'
//   tools/bin/decode_step_harness [layers = 32] [alternations = 3] [split qkv = 0] [o = 8] [gate|up = 0] [down = 0] [baseline only = 0] [batch = 1]
//   H_VARIANTS="0,1,2,3": which MLP forms to compare (0 shipped three launches, 1 one launch, 2 silu*mul inside the down launch, 3 gate|up atomics + down);
//   H_MAX_WAVES / H_DEFER_WG_PER_CU / H_GLUE_THREADS / H_GEMV_VARIANT / H_GEMM3_MIN_ROWS: the library's run-time knobs.  (Variants 2 and 3 and the knobs
//   were added after the round's GPU seconds were gone: compiled, not yet run.)
// batch > 4 (e.g. 16): the folded glue pipeline of llama_path._decode_step_fused_folded (8 launches per layer, generation-3 kernels) -- the three-launch
// variant only; this mode was written after the round's GPU seconds were gone: compiled against the header, not yet run.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "exl3_hip.h"

extern "C" int exl3_mlp1_fx(void* R, const void* norm_w, const float* ss_prev, float* ss_out, float eps,
                            const void* B_gate, const void* B_up, const void* suh_g, const void* suh_u, const void* svh_g, const void* svh_u,
                            const void* B_down, const void* suh_d, const void* svh_d, int m, int hidden, int inter, int K, int cb, void* stream);
extern "C" int exl3_mlp1_error(int* out, void* stream);

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

// one EXL3 linear with llama_path._rand_linear's magnitudes (outputs of O(out_scale) for unit-RMS inputs)
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
struct Layer { Lin q, k, v, o, g, u, d; __half *norm1, *norm2; uint32_t *kc, *vc; __half *ks, *vs; };

int main(int argc, char** argv)
{
    const int hidden = 4096, inter = 14336, hq = 32, hkv = 8, hd = 128, vocab = 128256, K = 4, cb = 2, page = 256, max_ctx = 4096, kv_bits = 4, pos = 1000;
    const int n_layers = argc > 1 ? atoi(argv[1]) : 32, alternations = argc > 2 ? atoi(argv[2]) : 3;
    // forced split-k factors of the four GEMV launches of a layer (0 = the library's choice; llama_path.decode_step_fx: o = 8, the others 0)
    const int sp_qkv = argc > 3 ? atoi(argv[3]) : 0, sp_o = argc > 4 ? atoi(argv[4]) : 8, sp_gu = argc > 5 ? atoi(argv[5]) : 0, sp_down = argc > 6 ? atoi(argv[6]) : 0;
    const int bsz = argc > 8 ? atoi(argv[8]) : 1;
    const bool baseline_only = (argc > 7 && atoi(argv[7]) != 0) || bsz > 1;          // time only the three-launch form (split sweeps, batches)
    const float eps = 1e-5f;
    CK(hipSetDevice(0));
    CE(exl3_init(0));
    hipStream_t st; CK(hipStreamCreate(&st));
    // the library's run-time tuning knobs, from the environment (one process per setting: tools/experiments/sweep_splits.sh style sweeps)
    if (const char* e = getenv("H_GEMV_VARIANT")) CE(exl3_set_gemv_variant(atoi(e)));          // 1 FAST (default) / 0 EXACT
    if (const char* e = getenv("H_MAX_WAVES")) CE(exl3_set_gemv_max_waves(atoi(e)));
    if (const char* e = getenv("H_DEFER_WG_PER_CU")) CE(exl3_set_gemv_defer_wg_per_cu(atoi(e)));
    if (const char* e = getenv("H_GLUE_THREADS")) CE(exl3_set_glue_threads(atoi(e)));
    if (const char* e = getenv("H_GEMM3_MIN_ROWS")) CE(exl3_set_gemm3_min_rows(atoi(e)));

    std::vector<Layer> L(n_layers);
    const int G = hkv * hd / 32, n_pages = max_ctx / page;      // pages per sequence (block_table [bsz][n_pages])
    for (int i = 0; i < n_layers; ++i)
    {
        L[i].q = make_lin(hidden, hq * hd, K, 0.5, st); L[i].k = make_lin(hidden, hkv * hd, K, 0.5, st); L[i].v = make_lin(hidden, hkv * hd, K, 0.5, st);
        L[i].o = make_lin(hq * hd, hidden, K, 0.5, st);
        L[i].g = make_lin(hidden, inter, K, 0.5, st); L[i].u = make_lin(hidden, inter, K, 0.5, st); L[i].d = make_lin(inter, hidden, K, 0.5, st);
        std::vector<float> nw(hidden); for (auto& x : nw) x = (float) (1.0 + 0.05 * nrand());
        L[i].norm1 = dev_half(nw); for (auto& x : nw) x = (float) (1.0 + 0.05 * nrand()); L[i].norm2 = dev_half(nw);
        const size_t cw = (size_t) (argc > 8 ? atoi(argv[8]) : 1) * n_pages * page * G * kv_bits, cs = (size_t) (argc > 8 ? atoi(argv[8]) : 1) * n_pages * page * G;
        CK(hipMalloc(&L[i].kc, cw * 4)); CK(hipMalloc(&L[i].vc, cw * 4)); CK(hipMalloc(&L[i].ks, cs * 2)); CK(hipMalloc(&L[i].vs, cs * 2));
        CK(hipMemsetAsync(L[i].kc, 0, cw * 4, st)); CK(hipMemsetAsync(L[i].vc, 0, cw * 4, st)); CK(hipMemsetAsync(L[i].ks, 0, cs * 2, st)); CK(hipMemsetAsync(L[i].vs, 0, cs * 2, st));
    }
    Lin head = make_lin(hidden, vocab, K, 0.5, st);
    std::vector<float> fnw(hidden); for (auto& x : fnw) x = (float) (1.0 + 0.05 * nrand());
    __half* final_norm = dev_half(fnw);

    // step state (llama_path.alloc_state, bsz = 1)
    std::vector<float> x0((size_t) bsz * hidden); for (auto& v : x0) v = (float) nrand();
    __half* dx0 = dev_half(x0);
    __half *q, *xout, *xh_d, *xh_head, *logits; int64_t *R, *slots; float *ssA, *ssB, *xs_d, *xs_head, *rsin, *rcos, *inv_freq; int32_t *positions, *block_table;
    CK(hipMalloc(&q, (size_t) bsz * hq * hd * 2)); CK(hipMalloc(&xout, (size_t) bsz * hidden * 2)); CK(hipMalloc(&xh_d, (size_t) bsz * inter * 2)); CK(hipMalloc(&xh_head, (size_t) bsz * hidden * 2)); CK(hipMalloc(&logits, (size_t) bsz * vocab * 2));
    CK(hipMalloc(&R, (size_t) bsz * hidden * 8)); CK(hipMalloc(&slots, (size_t) bsz * 8)); CK(hipMalloc(&ssA, (size_t) bsz * 32 * 4)); CK(hipMalloc(&ssB, (size_t) bsz * 32 * 4)); CK(hipMalloc(&xs_d, (size_t) bsz * inter / 128 * 4));
    CK(hipMalloc(&xs_head, (size_t) bsz * 32 * 4)); CK(hipMalloc(&rsin, (size_t) bsz * 64 * 4)); CK(hipMalloc(&rcos, (size_t) bsz * 64 * 4)); CK(hipMalloc(&inv_freq, 64 * 4)); CK(hipMalloc(&positions, (size_t) bsz * 4)); CK(hipMalloc(&block_table, (size_t) bsz * n_pages * 4));
    int64_t* GU; CK(hipMalloc(&GU, (size_t) 2 * inter * 8)); CK(hipMemsetAsync(GU, 0, (size_t) 2 * inter * 8, st));      // gate / up accumulators (variant 3)
    // batches: fp16 residual + three rotated-input buffers (q|k|v or gate|up) of the folded glue pipeline
    __half* xres; __half* xh3[3]; float* xs3[3];
    CK(hipMalloc(&xres, (size_t) bsz * hidden * 2));
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&xh3[i], (size_t) bsz * hidden * 2)); CK(hipMalloc(&xs3[i], (size_t) bsz * 32 * 4)); }
    {
        float f[64]; for (int i = 0; i < 64; ++i) f[i] = (float) (1.0 / pow(500000.0, (2.0 * i) / hd));
        CK(hipMemcpy(inv_freq, f, sizeof(f), hipMemcpyHostToDevice));
        std::vector<int32_t> p(bsz, pos); CK(hipMemcpy(positions, p.data(), (size_t) bsz * 4, hipMemcpyHostToDevice));
        std::vector<int32_t> bt((size_t) bsz * n_pages); for (size_t i = 0; i < bt.size(); ++i) bt[i] = (int32_t) i;
        CK(hipMemcpy(block_table, bt.data(), bt.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipStreamSynchronize(st));

    // MLP variants: 0 = gate|up (slabs) -> glue_act_rs -> down [shipped default at 8B]; 1 = one launch (exl3_mlp1.hip); 2 = silu*mul + rotation inside the down launch
    // (llama_path fx_act_in_gemv: 5 launches per layer); 3 = gate|up ADD into fixed-point accumulators, down forms silu*mul from them (fx_gu_atomic: 5 launches)
    auto step = [&] (int variant)
    {
        float* sc = ssA; float* so = ssB;
        CE(exl3_fx_init_prep(dx0, R, sc, 1, hidden, inv_freq, positions, 1.0f, hd, block_table, n_pages, page, rsin, rcos, slots, st));
        for (int i = 0; i < n_layers; ++i)
        {
            Layer& l = L[i];
            {
                const void* Bs[3] = { l.q.B, l.k.B, l.v.B }; const void* su[3] = { l.q.suh, l.k.suh, l.v.suh }; int ns[3] = { l.q.n, l.k.n, l.v.n };
                float* slabs[3] = { nullptr, nullptr, nullptr }; int S = 0;
                CE(exl3_gemv_ex_fx(R, l.norm1, sc, so, eps, Bs, su, ns, 3, 1, hidden, K, cb, sp_qkv, slabs, &S, st));
                CE(exl3_glue_qkv_tab(slabs[0], slabs[1], slabs[2], S, l.q.svh, l.k.svh, l.v.svh, q, nullptr, nullptr, inv_freq, positions, l.kc, l.ks, l.vc, l.vs,
                                     block_table, n_pages, page, kv_bits, kv_bits, 1, hq, hkv, hd, 2, 1.0f, sc, so, hidden, eps, rsin, rcos, slots, st));
                std::swap(sc, so);
            }
            {
                const void* Bs[1] = { l.o.B }; void* Cs[1] = { R }; const void* su[1] = { l.o.suh }; const void* sv[1] = { l.o.svh }; int ns[1] = { hidden };
                if (variant == 3) CE(exl3_fx_zero_next(GU, (int64_t) 2 * inter * 8));      // the o_proj launch clears the gate / up accumulators as a side job
                CE(exl3_gemv_ex(q, nullptr, nullptr, Bs, Cs, su, sv, nullptr, ns, 1, 1, hq * hd, K, cb, 0, EXL3_GEMV_OUT_ATOMIC, sp_o, nullptr, nullptr, st));
            }
            if (variant == 1)
                CE(exl3_mlp1_fx(R, l.norm2, sc, so, eps, l.g.B, l.u.B, l.g.suh, l.u.suh, l.g.svh, l.u.svh, l.d.B, l.d.suh, l.d.svh, 1, hidden, inter, K, cb, st));
            else if (variant == 2)
            {
                const void* Bs[2] = { l.g.B, l.u.B }; const void* su[2] = { l.g.suh, l.u.suh }; int ns[2] = { inter, inter };
                float* slabs[2] = { nullptr, nullptr }; int S = 0;
                CE(exl3_gemv_ex_fx(R, l.norm2, sc, so, eps, Bs, su, ns, 2, 1, hidden, K, cb, sp_gu, slabs, &S, st));
                CE(exl3_gemv_ex_act_rs(slabs[0], slabs[1], S, l.g.svh, l.u.svh, sc, so, hidden, eps, l.d.B, R, l.d.suh, l.d.svh, nullptr, 1, inter, hidden, K, cb,
                                       0, EXL3_GEMV_OUT_ATOMIC, 0, sp_down, nullptr, nullptr, st));
            }
            else if (variant == 3)
            {
                const void* Bs[2] = { l.g.B, l.u.B }; const void* su[2] = { l.g.suh, l.u.suh }; const void* sv[2] = { l.g.svh, l.u.svh }; int ns[2] = { inter, inter };
                void* accs[2] = { GU, GU + inter }; int S = 0;
                CE(exl3_gemv_ex_fx_atomic(R, l.norm2, sc, so, eps, Bs, accs, su, sv, ns, 2, 1, hidden, K, cb, sp_gu, &S, st));
                CE(exl3_gemv_ex_actfx(GU, GU + inter, sc, so, hidden, eps, l.d.B, R, l.d.suh, l.d.svh, nullptr, 1, inter, hidden, K, cb, EXL3_GEMV_OUT_ATOMIC, sp_down,
                                      nullptr, nullptr, st));
            }
            else
            {
                const void* Bs[2] = { l.g.B, l.u.B }; const void* su[2] = { l.g.suh, l.u.suh }; int ns[2] = { inter, inter };
                float* slabs[2] = { nullptr, nullptr }; int S = 0;
                CE(exl3_gemv_ex_fx(R, l.norm2, sc, so, eps, Bs, su, ns, 2, 1, hidden, K, cb, sp_gu, slabs, &S, st));
                CE(exl3_glue_act_rs(slabs[0], slabs[1], S, l.g.svh, l.u.svh, l.d.suh, xh_d, xs_d, nullptr, 1, inter, sc, so, hidden, eps, st));
                const void* xh[1] = { xh_d }; const float* xs[1] = { xs_d }; const void* Bd[1] = { l.d.B }; void* Cs[1] = { R }; const void* sv[1] = { l.d.svh }; int nd[1] = { hidden };
                CE(exl3_gemv_ex(nullptr, xh, xs, Bd, Cs, nullptr, sv, nullptr, nd, 1, 1, inter, K, cb, 0, EXL3_GEMV_IN_ROTATED | EXL3_GEMV_OUT_ATOMIC, sp_down, nullptr, nullptr, st));
            }
            std::swap(sc, so);
        }
        CE(exl3_fx_finish_rotate(R, xout, sc, final_norm, eps, head.suh, xh_head, xs_head, 1, hidden, st));
        const void* xh[1] = { xh_head }; const float* xs[1] = { xs_head }; const void* Bh[1] = { head.B }; void* Cs[1] = { logits }; const void* sv[1] = { head.svh }; int nh[1] = { vocab };
        CE(exl3_gemv_ex(nullptr, xh, xs, Bh, Cs, nullptr, sv, nullptr, nh, 1, 1, hidden, K, cb, 0, EXL3_GEMV_IN_ROTATED, 0, nullptr, nullptr, st));
    };

    // batches above 4 rows: llama_path._decode_step_fused_folded
    auto step_folded = [&] ()
    {
        const int DEF = EXL3_GEMV_OUT_DEFERRED, ROT = EXL3_GEMV_IN_ROTATED;
        float* ss_c = ssA; float* ss_o = ssB;
        CK(hipMemcpyAsync(xres, dx0, (size_t) bsz * hidden * 2, hipMemcpyDeviceToDevice, st));
        CE(exl3_qkv_prep(inv_freq, positions, 1.0f, bsz, hd, block_table, n_pages, page, rsin, rcos, slots, st));
        CE(exl3_glue_resid(nullptr, 0, nullptr, nullptr, nullptr, xres, ss_c, bsz, hidden, st));
        {
            const void* su[3] = { L[0].q.suh, L[0].k.suh, L[0].v.suh }; void* xh[3] = { xh3[0], xh3[1], xh3[2] }; float* xs[3] = { xs3[0], xs3[1], xs3[2] };
            CE(exl3_glue_rotate(xres, ss_c, L[0].norm1, eps, su, xh, xs, 3, bsz, hidden, st));
        }
        const float* rs_prev = nullptr; const float* rs_new = nullptr;
        for (int i = 0; i < n_layers; ++i)
        {
            Layer& l = L[i];
            {
                const void* xh[3] = { xh3[0], xh3[1], xh3[2] }; const float* xs[3] = { xs3[0], xs3[1], xs3[2] };
                const void* Bs[3] = { l.q.B, l.k.B, l.v.B }; int ns[3] = { l.q.n, l.k.n, l.v.n };
                float* slabs[3] = { nullptr, nullptr, nullptr }; int S = 0;
                CE(exl3_gemv_ex(nullptr, xh, xs, Bs, nullptr, nullptr, nullptr, nullptr, ns, 3, bsz, hidden, K, cb, 0, ROT | DEF, sp_qkv, slabs, &S, st));
                CE(exl3_glue_qkv_tab(slabs[0], slabs[1], slabs[2], S, l.q.svh, l.k.svh, l.v.svh, q, nullptr, nullptr, inv_freq, positions, l.kc, l.ks, l.vc, l.vs,
                                     block_table, n_pages, page, kv_bits, kv_bits, bsz, hq, hkv, hd, 2, 1.0f, rs_prev, rs_new, hidden, eps, rsin, rcos, slots, st));
            }
            {
                const void* Bs[1] = { l.o.B }; const void* su[1] = { l.o.suh }; int ns[1] = { hidden };
                float* so[1] = { nullptr }; int So = 0;
                CE(exl3_gemv_ex(q, nullptr, nullptr, Bs, nullptr, su, nullptr, nullptr, ns, 1, bsz, hq * hd, K, cb, 0, DEF, bsz > 4 ? 0 : sp_o, so, &So, st));
                const void* su2[2] = { l.g.suh, l.u.suh }; void* xh[2] = { xh3[0], xh3[1] }; float* xs[2] = { xs3[0], xs3[1] };
                CE(exl3_glue_resid_rotate(so[0], So, nullptr, l.o.svh, nullptr, xres, ss_c, ss_o, l.norm2, eps, su2, xh, xs, 2, bsz, hidden, st));
                rs_prev = ss_c; rs_new = ss_o; std::swap(ss_c, ss_o);
            }
            {
                const void* xh[2] = { xh3[0], xh3[1] }; const float* xs[2] = { xs3[0], xs3[1] };
                const void* Bs[2] = { l.g.B, l.u.B }; int ns[2] = { inter, inter };
                float* slabs[2] = { nullptr, nullptr }; int S = 0;
                CE(exl3_gemv_ex(nullptr, xh, xs, Bs, nullptr, nullptr, nullptr, nullptr, ns, 2, bsz, hidden, K, cb, 0, ROT | DEF, sp_gu, slabs, &S, st));
                CE(exl3_glue_act_rs(slabs[0], slabs[1], S, l.g.svh, l.u.svh, l.d.suh, xh_d, xs_d, nullptr, bsz, inter, rs_prev, rs_new, hidden, eps, st));
                const void* xd[1] = { xh_d }; const float* xsd[1] = { xs_d }; const void* Bd[1] = { l.d.B }; int nd[1] = { hidden };
                float* sd[1] = { nullptr }; int Sd = 0;
                CE(exl3_gemv_ex(nullptr, xd, xsd, Bd, nullptr, nullptr, nullptr, nullptr, nd, 1, bsz, inter, K, cb, 0, ROT | DEF, sp_down, sd, &Sd, st));
                if (i + 1 < n_layers)
                {
                    Layer& nl = L[i + 1];
                    const void* su3[3] = { nl.q.suh, nl.k.suh, nl.v.suh }; void* xh3o[3] = { xh3[0], xh3[1], xh3[2] }; float* xs3o[3] = { xs3[0], xs3[1], xs3[2] };
                    CE(exl3_glue_resid_rotate(sd[0], Sd, nullptr, l.d.svh, nullptr, xres, ss_c, ss_o, nl.norm1, eps, su3, xh3o, xs3o, 3, bsz, hidden, st));
                    rs_prev = ss_c; rs_new = ss_o; std::swap(ss_c, ss_o);
                }
                else CE(exl3_glue_resid(sd[0], Sd, nullptr, l.d.svh, nullptr, xres, ss_c, bsz, hidden, st));
            }
        }
        const void* suh[1] = { head.suh }; void* xhh[1] = { xh3[0] }; float* xsh[1] = { xs3[0] };
        CE(exl3_glue_rotate(xres, ss_c, final_norm, eps, suh, xhh, xsh, 1, bsz, hidden, st));
        const void* xh[1] = { xh3[0] }; const float* xs[1] = { xs3[0] }; const void* Bh[1] = { head.B }; void* Cs[1] = { logits }; const void* sv[1] = { head.svh }; int nh[1] = { vocab };
        CE(exl3_gemv_ex(nullptr, xh, xs, Bh, Cs, nullptr, sv, nullptr, nh, 1, bsz, hidden, K, cb, 0, ROT, 0, nullptr, nullptr, st));
    };
    if (bsz > 4)
    {
        step_folded(); CK(hipStreamSynchronize(st));
        hipGraph_t g; hipGraphExec_t gx;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); step_folded(); CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&gx, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
        hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
        double bestb = 1e30;
        for (int a = 0; a < alternations; ++a)
        {
            for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(gx, st));
            CK(hipEventRecord(a0, st));
            for (int i = 0; i < 30; ++i) CK(hipGraphLaunch(gx, st));
            CK(hipEventRecord(a1, st)); CK(hipEventSynchronize(a1));
            float ms; CK(hipEventElapsedTime(&ms, a0, a1)); bestb = std::min(bestb, (double) ms / 30.0);
        }
        std::vector<__half> lgb((size_t) bsz * vocab); CK(hipMemcpy(lgb.data(), logits, lgb.size() * 2, hipMemcpyDeviceToHost));
        int bad = 0; double r2 = 0; for (auto h : lgb) { const double v = __half2float(h); if (!std::isfinite(v)) ++bad; else r2 += v * v; }
        if (const char* df = getenv("H_DUMP_TIMING"))
        {
            // timing builds of the library (-DG2_TIMING) with EXL3_HIP_TIMING_SLOTS=8: the phase stamps of the step's last 8 GEMV launches, as left by the last replay
            std::vector<char> hb((size_t) 4 << 20); void* db; CK(hipMalloc(&db, hb.size()));
            CE(exl3_debug_copy_workspace(db, (int64_t) 48 << 20, (int64_t) hb.size(), st)); CK(hipStreamSynchronize(st));
            CK(hipMemcpy(hb.data(), db, hb.size(), hipMemcpyDeviceToHost));
            FILE* f = fopen(df, "wb"); if (f) { fwrite(hb.data(), 1, hb.size(), f); fclose(f); }
        }
        printf("{\"model\": \"llama-3.1-8b shapes, %d layers, EXL3 4.0 bpw mul1, bs %d, folded glue pipeline via the C ABI\", \"ms_per_step\": %.4f, \"tok_s\": %.1f, "
               "\"us_per_layer\": %.2f, \"logits_rms\": %.5g, \"nonfinite\": %d, \"splits\": {\"qkv\": %d, \"gate_up\": %d, \"down\": %d}}\n",
               n_layers, bsz, bestb, bsz * 1e3 / bestb, bestb * 1e3 / n_layers, sqrt(r2 / (double) lgb.size()), bad, sp_qkv, sp_gu, sp_down);
        return 0;
    }

    // selected variants (H_VARIANTS="0,1,2,3"; default "0,1"; baseline only: "0"): logits against variant 0, then alternating graph replays
    const char* vnames[4] = { "three_launch_mlp", "one_launch_mlp", "act_in_down", "gate_up_atomic" };
    std::vector<int> vs;
    {
        const char* e = baseline_only ? "0" : (getenv("H_VARIANTS") ? getenv("H_VARIANTS") : "0,1");
        for (const char* c = e; *c; ++c) if (*c >= '0' && *c <= '3') vs.push_back(*c - '0');
        if (vs.empty() || vs[0] != 0) vs.insert(vs.begin(), 0);
    }
    std::vector<std::vector<__half>> lg(vs.size(), std::vector<__half>(vocab));
    for (size_t i = 0; i < vs.size(); ++i)
    {
        step(vs[i]);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(lg[i].data(), logits, (size_t) vocab * 2, hipMemcpyDeviceToHost));
    }
    int err = -1; CE(exl3_mlp1_error(&err, st));
    printf("{\"model\": \"llama-3.1-8b shapes, %d layers, EXL3 4.0 bpw mul1, bs 1, fx pipeline via the C ABI\", \"mlp1_err_word\": %d, \"logits_vs_variant_0\": [", n_layers, err);
    for (size_t i = 0; i < vs.size(); ++i)
    {
        double rms = 0, maxd = 0, sd2 = 0; int nonfinite = 0;
        for (int c = 0; c < vocab; ++c)
        {
            const double a = __half2float(lg[0][c]), b2 = __half2float(lg[i][c]);
            if (!std::isfinite(a) || !std::isfinite(b2)) { ++nonfinite; continue; }
            rms += a * a; sd2 += (a - b2) * (a - b2); maxd = std::max(maxd, fabs(a - b2));
        }
        rms = sqrt(rms / vocab);
        printf("%s{\"variant\": \"%s\", \"logits_rms\": %.5g, \"max_abs_diff\": %.5g, \"rms_diff\": %.5g, \"nonfinite\": %d}", i ? ", " : "", vnames[vs[i]], rms, maxd, sqrt(sd2 / vocab), nonfinite);
    }
    printf("],\n");
    fflush(stdout);

    std::vector<hipGraphExec_t> ge(vs.size());
    for (size_t i = 0; i < vs.size(); ++i)
    {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        step(vs[i]);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge[i], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf(" \"ms_per_step\": [");
    std::vector<double> best(vs.size(), 1e30);
    for (int a = 0; a < alternations; ++a)
        for (size_t i = 0; i < vs.size(); ++i)
        {
            for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge[i], st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < 50; ++r) CK(hipGraphLaunch(ge[i], st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best[i] = std::min(best[i], (double) ms / 50.0);
            printf("%s{\"variant\": \"%s\", \"ms\": %.4f}", (a || i) ? ", " : "", vnames[vs[i]], ms / 50.0);
            fflush(stdout);
        }
    err = -1; CE(exl3_mlp1_error(&err, st));
    printf("],\n \"splits\": {\"qkv\": %d, \"o\": %d, \"gate_up\": %d, \"down\": %d}, \"tok_s\": {", sp_qkv, sp_o, sp_gu, sp_down);
    for (size_t i = 0; i < vs.size(); ++i) printf("%s\"%s\": %.1f", i ? ", " : "", vnames[vs[i]], 1e3 / best[i]);
    printf("}, \"us_per_layer\": {");
    for (size_t i = 0; i < vs.size(); ++i) printf("%s\"%s\": %.2f", i ? ", " : "", vnames[vs[i]], best[i] * 1e3 / n_layers);
    printf("}, \"mlp1_err_word_after_timing\": %d}\n", err);
    return 0;
}
