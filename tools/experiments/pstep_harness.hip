// The persistent decode step (exl3_pstep_*, generation 5) against the shipped launch-per-op fx pipeline, both driven through the C ABI from C++ -- no Python,
// no torch on the box: synthetic EXL3 mul1 tensors filled on the device, one hipGraph per step per variant, alternating replays, logits compared.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o tools/bin/pstep_harness tools/experiments/pstep_harness.hip -L exllamav3_amd -lexl3_hip \
//         -Wl,-rpath,'$ORIGIN/../../exllamav3_amd'
//   tools/bin/pstep_harness [model = 8b | 1b] [layers = model's] [alternations = 3] [decode-ahead units list = "2,1,0"] [stamps file = none]
// Baseline = llama_path.decode_step_fx as bench.py times it: q|k|v (NORMFX, slabs) -> o_proj with the q|k|v epilogue (QKVM, atomics) -> gate|up (NORMFX, slabs)
// -> [8B: glue_act_rs -> down (ROT, atomics) | 1B: down with silu * mul inside (ACT, atomics)] -> fx_finish_rotate -> lm_head.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
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
struct Layer { Lin q, k, v, o, g, u, d; __half *norm1, *norm2; uint32_t *kc, *vc; __half *ks, *vs; };
static exl3_pstep_linear_t pl(const Lin& l) { exl3_pstep_linear_t r; r.trellis = l.B; r.suh = l.suh; r.svh = l.svh; r.k = l.k; r.n = l.n; r.K = 0; r.cb = 0; return r; }      // (K = 0: the create call's K / codebook)

int main(int argc, char** argv)
{
    const bool small = argc > 1 && !strcmp(argv[1], "1b");
    const int hidden = small ? 2048 : 4096, inter = small ? 8192 : 14336, hq = 32, hkv = 8, hd = small ? 64 : 128, vocab = 128256, K = 4, cb = getenv("H_CB") ? atoi(getenv("H_CB")) : 2;      // H_CB: 0 3INST | 1 mcg | 2 mul1 (default)
    const int page = 256, max_ctx = 4096, kv_bits = 4, pos = 1000;
    const int n_layers = argc > 2 && atoi(argv[2]) > 0 ? atoi(argv[2]) : (small ? 16 : 32), alternations = argc > 3 ? atoi(argv[3]) : 3;
    const char* plist = argc > 4 ? argv[4] : "2,1,0";
    const char* stamps_file = argc > 5 ? argv[5] : nullptr;
    const bool act_in_gemv = small;                                  // llama_path: fx_act_in_gemv on for hidden <= 2048
    const float eps = 1e-5f;
    setvbuf(stdout, nullptr, _IONBF, 0);
    const char* only = getenv("H_ONLY");                            // "base" | "ps": run one side only (fault isolation)
    CK(hipSetDevice(0));
    CE(exl3_init(0));
    hipStream_t st; CK(hipStreamCreate(&st));

    std::vector<Layer> L(n_layers);
    const int G = hkv * hd / 32, n_pages = max_ctx / page;
    for (int i = 0; i < n_layers; ++i)
    {
        L[i].q = make_lin(hidden, hq * hd, K, 0.5, st); L[i].k = make_lin(hidden, hkv * hd, K, 0.5, st); L[i].v = make_lin(hidden, hkv * hd, K, 0.5, st);
        L[i].o = make_lin(hq * hd, hidden, K, 0.5, st);
        L[i].g = make_lin(hidden, inter, K, 0.5, st); L[i].u = make_lin(hidden, inter, K, 0.5, st); L[i].d = make_lin(inter, hidden, K, 0.5, st);
        std::vector<float> nw(hidden); for (auto& x : nw) x = (float) (1.0 + 0.05 * nrand());
        L[i].norm1 = dev_half(nw); for (auto& x : nw) x = (float) (1.0 + 0.05 * nrand()); L[i].norm2 = dev_half(nw);
        const size_t cw = (size_t) n_pages * page * G * kv_bits, cs = (size_t) n_pages * page * G;
        CK(hipMalloc(&L[i].kc, cw * 4)); CK(hipMalloc(&L[i].vc, cw * 4)); CK(hipMalloc(&L[i].ks, cs * 2)); CK(hipMalloc(&L[i].vs, cs * 2));
        CK(hipMemsetAsync(L[i].kc, 0, cw * 4, st)); CK(hipMemsetAsync(L[i].vc, 0, cw * 4, st)); CK(hipMemsetAsync(L[i].ks, 0, cs * 2, st)); CK(hipMemsetAsync(L[i].vs, 0, cs * 2, st));
    }
    Lin head = make_lin(hidden, vocab, K, 0.5, st);
    std::vector<float> fnw(hidden); for (auto& x : fnw) x = (float) (1.0 + 0.05 * nrand());
    __half* final_norm = dev_half(fnw);

    std::vector<float> x0(hidden); for (auto& v : x0) v = (float) nrand();
    __half* dx0 = dev_half(x0);
    __half *q, *xout, *xh_d, *xh_head, *logits; int64_t *R, *slots; float *ssA, *ssB, *xs_d, *xs_head, *rsin, *rcos, *inv_freq; int32_t *positions, *block_table;
    const int nbh = hidden / 128;
    CK(hipMalloc(&q, (size_t) hq * hd * 2)); CK(hipMalloc(&xout, (size_t) hidden * 2)); CK(hipMalloc(&xh_d, (size_t) inter * 2)); CK(hipMalloc(&xh_head, (size_t) hidden * 2)); CK(hipMalloc(&logits, (size_t) vocab * 2));
    CK(hipMalloc(&R, (size_t) hidden * 8)); CK(hipMalloc(&slots, 8)); CK(hipMalloc(&ssA, (size_t) nbh * 4)); CK(hipMalloc(&ssB, (size_t) nbh * 4)); CK(hipMalloc(&xs_d, (size_t) inter / 128 * 4));
    CK(hipMalloc(&xs_head, (size_t) nbh * 4)); CK(hipMalloc(&rsin, 64 * 4)); CK(hipMalloc(&rcos, 64 * 4)); CK(hipMalloc(&inv_freq, 64 * 4)); CK(hipMalloc(&positions, 4)); CK(hipMalloc(&block_table, (size_t) n_pages * 4));
    {
        float f[64]; for (int i = 0; i < 64; ++i) f[i] = (float) (1.0 / pow(500000.0, (2.0 * (i % (hd / 2))) / hd));
        CK(hipMemcpy(inv_freq, f, sizeof(f), hipMemcpyHostToDevice));
        int32_t p = pos; CK(hipMemcpy(positions, &p, 4, hipMemcpyHostToDevice));
        std::vector<int32_t> bt(n_pages); for (size_t i = 0; i < bt.size(); ++i) bt[i] = (int32_t) i;
        CK(hipMemcpy(block_table, bt.data(), bt.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipStreamSynchronize(st));

    // ---- the persistent step's plan
    std::vector<exl3_pstep_layer_t> PL(n_layers);
    for (int i = 0; i < n_layers; ++i)
    {
        exl3_pstep_layer_t& p = PL[i]; const Layer& l = L[i];
        p.q = pl(l.q); p.k = pl(l.k); p.v = pl(l.v); p.o = pl(l.o); p.gate = pl(l.g); p.up = pl(l.u); p.down = pl(l.d);
        p.norm1 = l.norm1; p.norm2 = l.norm2; p.k_cache = l.kc; p.k_scales = l.ks; p.v_cache = l.vc; p.v_scales = l.vs;
    }
    exl3_pstep_linear_t ph = pl(head);
    void* ps = nullptr;
    CE(exl3_pstep_create(&ps, PL.data(), n_layers, &ph, final_norm, hidden, hq, hkv, hd, K, cb, eps, 2, stamps_file ? 1 : 0));
    { char buf[1024]; CE(exl3_pstep_describe(ps, buf, sizeof(buf))); printf("{\"plan\": \"%s\",\n", buf); }
    if (const char* e = getenv("H_SPIN_LIMIT")) CE(exl3_pstep_set(ps, -1, atoi(e)));

    auto step_base = [&] ()
    {
        float* sc = ssA; float* so = ssB;
        CE(exl3_fx_init_prep(dx0, R, sc, 1, hidden, inv_freq, positions, 1.0f, hd, block_table, n_pages, page, rsin, rcos, slots, st));
        for (int i = 0; i < n_layers; ++i)
        {
            Layer& l = L[i];
            float* qs[3] = { nullptr, nullptr, nullptr }; int Sq = 0;
            {
                const void* Bs[3] = { l.q.B, l.k.B, l.v.B }; const void* su[3] = { l.q.suh, l.k.suh, l.v.suh }; int ns[3] = { l.q.n, l.k.n, l.v.n };
                CE(exl3_gemv_ex_fx(R, l.norm1, sc, so, eps, Bs, su, ns, 3, 1, hidden, K, cb, 0, qs, &Sq, st));
            }
            CE(exl3_gemv_ex_qkvm(qs[0], qs[1], qs[2], Sq, l.q.svh, l.k.svh, l.v.svh, rsin, rcos, slots, sc, so, hidden, eps, 2, hd, hkv, q, l.kc, l.ks, l.vc, l.vs,
                                 l.o.B, R, l.o.suh, l.o.svh, nullptr, 1, hq * hd, hidden, K, cb, 0, EXL3_GEMV_OUT_ATOMIC, (hq * hd == 4096 && hidden == 4096) ? 8 : 0, nullptr, nullptr, st));
            std::swap(sc, so);
            float* gs[2] = { nullptr, nullptr }; int Sg = 0;
            {
                const void* Bs[2] = { l.g.B, l.u.B }; const void* su[2] = { l.g.suh, l.u.suh }; int ns[2] = { inter, inter };
                CE(exl3_gemv_ex_fx(R, l.norm2, sc, so, eps, Bs, su, ns, 2, 1, hidden, K, cb, 0, gs, &Sg, st));
            }
            if (act_in_gemv)
                CE(exl3_gemv_ex_act_rs(gs[0], gs[1], Sg, l.g.svh, l.u.svh, sc, so, hidden, eps, l.d.B, R, l.d.suh, l.d.svh, nullptr, 1, inter, hidden, K, cb,
                                       0, EXL3_GEMV_OUT_ATOMIC, 0, 0, nullptr, nullptr, st));
            else
            {
                CE(exl3_glue_act_rs(gs[0], gs[1], Sg, l.g.svh, l.u.svh, l.d.suh, xh_d, xs_d, nullptr, 1, inter, sc, so, hidden, eps, st));
                const void* xh[1] = { xh_d }; const float* xs[1] = { xs_d }; const void* Bd[1] = { l.d.B }; void* Cs[1] = { R }; const void* sv[1] = { l.d.svh }; int nd[1] = { hidden };
                CE(exl3_gemv_ex(nullptr, xh, xs, Bd, Cs, nullptr, sv, nullptr, nd, 1, 1, inter, K, cb, 0, EXL3_GEMV_IN_ROTATED | EXL3_GEMV_OUT_ATOMIC, 0, nullptr, nullptr, st));
            }
            std::swap(sc, so);
        }
        CE(exl3_fx_finish_rotate(R, xout, sc, final_norm, eps, head.suh, xh_head, xs_head, 1, hidden, st));
        const void* xh[1] = { xh_head }; const float* xs[1] = { xs_head }; const void* Bh[1] = { head.B }; void* Cs[1] = { logits }; const void* sv[1] = { head.svh }; int nh[1] = { vocab };
        CE(exl3_gemv_ex(nullptr, xh, xs, Bh, Cs, nullptr, sv, nullptr, nh, 1, 1, hidden, K, cb, 0, EXL3_GEMV_IN_ROTATED, 0, nullptr, nullptr, st));
    };
    auto step_ps = [&] ()
    {
        CE(exl3_fx_init_prep(dx0, R, ssA, 1, hidden, inv_freq, positions, 1.0f, hd, block_table, n_pages, page, rsin, rcos, slots, st));
        CE(exl3_pstep_run(ps, R, logits, q, rsin, rcos, slots, st));
    };

    // variants: index 0 = baseline, then one persistent variant per decode-ahead depth in the list
    std::vector<int> pm;
    for (const char* c = plist; *c; ++c) if (*c >= '0' && *c <= '3') pm.push_back(*c - '0');
    const size_t nv = 1 + pm.size();
    auto run_variant = [&] (size_t v) { if (v == 0) step_base(); else { CE(exl3_pstep_set(ps, pm[v - 1], 0)); step_ps(); } };

    // K / V words the step appends (layer 0, the new token's row) + logits, per variant
    const size_t kv_words = (size_t) G * kv_bits;
    std::vector<std::vector<__half>> lg(nv, std::vector<__half>(vocab));
    std::vector<std::vector<uint32_t>> kvw(nv, std::vector<uint32_t>(2 * kv_words));
    std::vector<std::vector<__half>> qv(nv, std::vector<__half>((size_t) hq * hd));
    std::vector<int> errw(nv, 0);
    for (size_t v = 0; v < nv; ++v)
    {
        if (only && ((v == 0) != !strcmp(only, "base"))) continue;
        fprintf(stderr, "[eager run of variant %d]\n", (int) v);
        CK(hipMemsetAsync(logits, 0, (size_t) vocab * 2, st));
        run_variant(v);
        CK(hipStreamSynchronize(st));
        if (v > 0) errw[v] = exl3_pstep_error(ps, st);
        CK(hipMemcpy(lg[v].data(), logits, (size_t) vocab * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(kvw[v].data(), L[0].kc + (size_t) pos * kv_words, kv_words * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(kvw[v].data() + kv_words, L[0].vc + (size_t) pos * kv_words, kv_words * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(qv[v].data(), q, (size_t) hq * hd * 2, hipMemcpyDeviceToHost));
    }
    printf(" \"model\": \"%s shapes, %d layers, EXL3 4.0 bpw mul1, bs 1\", \"vs_baseline\": [", small ? "llama-3.2-1b" : "llama-3.1-8b", n_layers);
    for (size_t v = 0; v < nv; ++v)
    {
        double rms = 0, maxd = 0, sd2 = 0; int nonfinite = 0;
        for (int c = 0; c < vocab; ++c)
        {
            const double a = __half2float(lg[0][c]), b2 = __half2float(lg[v][c]);
            if (!std::isfinite(a) || !std::isfinite(b2)) { ++nonfinite; continue; }
            rms += a * a; sd2 += (a - b2) * (a - b2); maxd = std::max(maxd, fabs(a - b2));
        }
        rms = sqrt(rms / vocab);
        int kvdiff = 0; for (size_t i = 0; i < 2 * kv_words; ++i) if (kvw[v][i] != kvw[0][i]) ++kvdiff;
        double qd = 0; for (size_t i = 0; i < (size_t) hq * hd; ++i) qd = std::max(qd, fabs((double) __half2float(qv[v][i]) - (double) __half2float(qv[0][i])));
        printf("%s{\"variant\": \"%s%d\", \"logits_rms\": %.5g, \"max_abs_diff\": %.5g, \"rms_diff\": %.5g, \"nonfinite\": %d, \"kv_words_differ_layer0\": %d, \"q_last_layer_max_diff\": %.4g, \"edge_timeout\": %d}",
               v ? ", " : "", v ? "persistent_ahead" : "launches", v ? pm[v - 1] : 0, rms, maxd, sqrt(sd2 / vocab), nonfinite, kvdiff, qd, errw[v]);
    }
    printf("],\n");
    fflush(stdout);

    if (only) return 0;
    std::vector<hipGraphExec_t> ge(nv);
    for (size_t v = 0; v < nv; ++v)
    {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        run_variant(v);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge[v], g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf(" \"ms_per_step\": [");
    std::vector<double> best(nv, 1e30);
    for (int a = 0; a < alternations; ++a)
        for (size_t v = 0; v < nv; ++v)
        {
            // (a captured launch carries the arguments of its capture: the decode-ahead depth of variant v is in its graph)
            for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge[v], st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < 50; ++r) CK(hipGraphLaunch(ge[v], st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best[v] = std::min(best[v], (double) ms / 50.0);
            printf("%s{\"v\": %d, \"ms\": %.4f}", (a || v) ? ", " : "", (int) v, ms / 50.0);
            fflush(stdout);
        }
    const int to = exl3_pstep_error(ps, st);
    printf("],\n \"best\": [");
    for (size_t v = 0; v < nv; ++v)
        printf("%s{\"variant\": \"%s%d\", \"ms_per_step\": %.4f, \"tok_s\": %.1f, \"us_per_layer\": %.2f}", v ? ", " : "", v ? "persistent_ahead" : "launches", v ? pm[v - 1] : 0,
               best[v], 1e3 / best[v], best[v] * 1e3 / n_layers);
    printf("], \"edge_timeout_during_replays\": %d}\n", to);

    if (stamps_file)
    {
        // one eager run with the deepest decode-ahead, then the phase stamps [op][cu][32] (100 MHz ticks)
        CE(exl3_pstep_set(ps, pm.empty() ? 2 : pm[0], 0));
        step_ps(); CK(hipStreamSynchronize(st));
        const int nops = 4 * n_layers + 1;
        std::vector<uint64_t> sb((size_t) nops * 512 * 32 + 16);
        const int64_t got = exl3_pstep_stamps(ps, sb.data(), (int64_t) sb.size(), st);
        FILE* f = fopen(stamps_file, "wb");
        if (f && got > 0) { fwrite(sb.data(), 8, (size_t) got, f); fclose(f); fprintf(stderr, "stamps: %lld words -> %s\n", (long long) got, stamps_file); }
    }
    CE(exl3_pstep_destroy(ps));
    return 0;
}
