// Stand-alone check + timing of exl3_mlp1_fx (the MLP block of a batch-1 decode step in one launch, exl3_mlp1.hip) against the three-launch form of the
// fixed-point-residual pipeline (exl3_gemv_ex_fx over gate|up -> exl3_glue_act_rs -> exl3_gemv_ex ROTATED | ATOMIC over down), through the C ABI only --
// no Python, no torch: a run costs seconds of GPU time.  The three-launch form is the one tests/test_gpu_path.py pins against the oracle; this
// harness compares the two on the same random EXL3 tensors (Llama-3.1-8B shapes) and times both under hipGraph replay over 8 rotating weight sets
// (0.7 GB: beyond the 256 MiB Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-rdc -I include -I exllamav3_amd/csrc -o tools/bin/mlp1_harness \
//         tools/experiments/mlp1_harness.hip tools/experiments/exl3_mlp1.hip -L exllamav3_amd -lexl3_hip -Wl,-rpath,'$ORIGIN/../../exllamav3_amd'
//   tools/bin/mlp1_harness            (EXL3_HIP_MLP1_TIMING=1: per-workgroup phase durations of one launch)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "exl3_hip.h"

// the experiment's entry points (exl3_mlp1.hip, linked into this binary; everything else comes from libexl3_hip.so)
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

static uint64_t rng_s = 0x1234567887654321ull;
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

struct Layer { uint32_t *Bg, *Bu, *Bd; __half *suh_g, *suh_u, *svh_g, *svh_u, *suh_d, *svh_d, *norm_w; };

int main(int argc, char** argv)
{
    const int hidden = 4096, inter = argc > 1 ? atoi(argv[1]) : 14336, K = 4, cb = 2, NL = 8;
    const float eps = 1e-5f;
    CK(hipSetDevice(0));
    CE(exl3_init(0));
    hipStream_t st; CK(hipStreamCreate(&st));
    std::vector<Layer> L(NL);
    const size_t wwords = (size_t) hidden * inter * K / 32;
    for (int i = 0; i < NL; ++i)
    {
        CK(hipMalloc(&L[i].Bg, wwords * 4)); CK(hipMalloc(&L[i].Bu, wwords * 4)); CK(hipMalloc(&L[i].Bd, wwords * 4));
        fill_hash<<<1024, 256, 0, st>>>(L[i].Bg, wwords, 11u + 3u * i); fill_hash<<<1024, 256, 0, st>>>(L[i].Bu, wwords, 12u + 3u * i);
        fill_hash<<<1024, 256, 0, st>>>(L[i].Bd, wwords, 13u + 3u * i);
        L[i].suh_g = dev_half(scale_vec(hidden, 1.0)); L[i].suh_u = dev_half(scale_vec(hidden, 1.0));
        L[i].svh_g = dev_half(scale_vec(inter, 1.0 / sqrt((double) hidden))); L[i].svh_u = dev_half(scale_vec(inter, 1.0 / sqrt((double) hidden)));
        L[i].suh_d = dev_half(scale_vec(inter, 1.0)); L[i].svh_d = dev_half(scale_vec(hidden, 0.5 / sqrt((double) inter)));
        std::vector<float> nw(hidden); for (auto& x : nw) x = (float) (1.0 + 0.05 * nrand());
        L[i].norm_w = dev_half(nw);
    }
    // the residual row: unit-RMS x with a few large channels; ss_prev = the block sums of squares of a DIFFERENT (1.21 x) residual, so r_new / r_prev != 1
    std::vector<float> x(hidden); for (auto& v : x) v = (float) nrand(); x[7] = 40.0f; x[1000] = -25.0f;
    __half* dx = dev_half(x);
    int64_t *R0, *Rref, *Rnew; CK(hipMalloc(&R0, hidden * 8)); CK(hipMalloc(&Rref, hidden * 8)); CK(hipMalloc(&Rnew, hidden * 8));
    float *ss0, *ss_prev, *ss_out_ref, *ss_out_new;
    CK(hipMalloc(&ss0, 32 * 4)); CK(hipMalloc(&ss_prev, 32 * 4)); CK(hipMalloc(&ss_out_ref, 32 * 4)); CK(hipMalloc(&ss_out_new, 32 * 4));
    CE(exl3_fx_init(dx, R0, ss0, 1, hidden, st));
    CK(hipStreamSynchronize(st));
    {
        float h[32]; CK(hipMemcpy(h, ss0, 128, hipMemcpyDeviceToHost));
        for (int i = 0; i < 32; ++i) h[i] *= 1.21f;
        CK(hipMemcpy(ss_prev, h, 128, hipMemcpyHostToDevice));
    }
    __half* xh_d; float* xs_d; CK(hipMalloc(&xh_d, inter * 2)); CK(hipMalloc(&xs_d, inter / 128 * 4));

    auto ref_mlp = [&] (const Layer& l, int64_t* R, float* ss_out)
    {
        const void* Bs[2] = { l.Bg, l.Bu }; const void* su[2] = { l.suh_g, l.suh_u }; int ns[2] = { inter, inter };
        float* slabs[2] = { nullptr, nullptr }; int S = 0;
        CE(exl3_gemv_ex_fx(R, l.norm_w, ss_prev, ss_out, eps, Bs, su, ns, 2, 1, hidden, K, cb, 0, slabs, &S, st));
        CE(exl3_glue_act_rs(slabs[0], slabs[1], S, l.svh_g, l.svh_u, l.suh_d, xh_d, xs_d, nullptr, 1, inter, ss_prev, ss_out, hidden, eps, st));
        const void* xh[1] = { xh_d }; const float* xs[1] = { xs_d }; const void* Bd[1] = { l.Bd }; void* Cs[1] = { R }; const void* sv[1] = { l.svh_d }; int nd[1] = { hidden };
        CE(exl3_gemv_ex(nullptr, xh, xs, Bd, Cs, nullptr, sv, nullptr, nd, 1, 1, inter, K, cb, 0, EXL3_GEMV_IN_ROTATED | EXL3_GEMV_OUT_ATOMIC, 0, nullptr, nullptr, st));
    };
    auto new_mlp = [&] (const Layer& l, int64_t* R, float* ss_out)
    {
        CE(exl3_mlp1_fx(R, l.norm_w, ss_prev, ss_out, eps, l.Bg, l.Bu, l.suh_g, l.suh_u, l.svh_g, l.svh_u, l.Bd, l.suh_d, l.svh_d, 1, hidden, inter, K, cb, st));
    };

    // ---- parity: one MLP on each of two weight sets, both forms from the same residual
    printf("{\"hidden\": %d, \"inter\": %d, \"parity\": [", hidden, inter);
    for (int li = 0; li < 2; ++li)
    {
        CK(hipMemcpyAsync(Rref, R0, hidden * 8, hipMemcpyDeviceToDevice, st)); CK(hipMemcpyAsync(Rnew, R0, hidden * 8, hipMemcpyDeviceToDevice, st));
        ref_mlp(L[li], Rref, ss_out_ref);
        new_mlp(L[li], Rnew, ss_out_new);
        int err = -1; CE(exl3_mlp1_error(&err, st));
        CK(hipStreamSynchronize(st));
        std::vector<int64_t> h0(hidden), hr(hidden), hn(hidden);
        CK(hipMemcpy(h0.data(), R0, hidden * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), Rref, hidden * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hn.data(), Rnew, hidden * 8, hipMemcpyDeviceToHost));
        float sr[32], sn[32]; CK(hipMemcpy(sr, ss_out_ref, 128, hipMemcpyDeviceToHost)); CK(hipMemcpy(sn, ss_out_new, 128, hipMemcpyDeviceToHost));
        double rms = 0, maxd = 0, sumd2 = 0; int worst = 0, nonzero_new = 0;
        for (int i = 0; i < hidden; ++i)
        {
            const double dr = (double) (hr[i] - h0[i]) / 4294967296.0, dn = (double) (hn[i] - h0[i]) / 4294967296.0;
            rms += dr * dr; sumd2 += (dr - dn) * (dr - dn);
            if (fabs(dr - dn) > maxd) { maxd = fabs(dr - dn); worst = i; }
            if (hn[i] != h0[i]) ++nonzero_new;
        }
        rms = sqrt(rms / hidden);
        int ss_bad = 0; for (int i = 0; i < 32; ++i) if (sr[i] != sn[i]) ++ss_bad;
        printf("%s{\"set\": %d, \"mlp_out_rms\": %.6g, \"max_abs_diff\": %.6g, \"rms_diff\": %.6g, \"rel_to_rms\": %.3g, \"worst_col\": %d, \"ref_at_worst\": %.6g, \"new_at_worst\": %.6g, "
               "\"cols_changed_by_new\": %d, \"ss_out_mismatches\": %d, \"err_word\": %d}", li ? ", " : "", li, rms, maxd, sqrt(sumd2 / hidden), maxd / (rms + 1e-30), worst,
               (double) (hr[worst] - h0[worst]) / 4294967296.0, (double) (hn[worst] - h0[worst]) / 4294967296.0, nonzero_new, ss_bad, err);
        if (li == 0)
        {
            // first-failure localisation: per down column block (128 columns) the largest difference
            fprintf(stderr, "per column block max |ref - new| (set 0):");
            for (int cbk = 0; cbk < hidden / 128; ++cbk)
            {
                double m = 0; for (int i = 0; i < 128; ++i) { const int c = cbk * 128 + i; m = std::max(m, fabs((double) (hr[c] - hn[c]) / 4294967296.0)); }
                fprintf(stderr, " %.2g", m);
            }
            fprintf(stderr, "\n");
        }
    }
    printf("],\n");
    fflush(stdout);

    // ---- timing: 8 MLPs (one per weight set) per graph, back to back on one residual
    auto time_graph = [&] (bool use_new) -> double
    {
        CK(hipMemcpyAsync(Rref, R0, hidden * 8, hipMemcpyDeviceToDevice, st));
        for (int i = 0; i < NL; ++i) { if (use_new) new_mlp(L[i], Rref, ss_out_ref); else ref_mlp(L[i], Rref, ss_out_ref); }     // eager warm-up
        CK(hipStreamSynchronize(st));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < NL; ++i) { if (use_new) new_mlp(L[i], Rref, ss_out_ref); else ref_mlp(L[i], Rref, ss_out_ref); }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep)
        {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, (double) ms * 1e3 / (10.0 * NL));
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return best;
    };
    if (getenv("EXL3_HIP_MLP1_TIMING") && atoi(getenv("EXL3_HIP_MLP1_TIMING")))
    {
        // per-workgroup phase stamps of ONE launch after the caches have been flushed by the other weight sets (100 MHz counter)
        for (int i = 1; i < NL; ++i) new_mlp(L[i], Rref, ss_out_ref);
        new_mlp(L[0], Rref, ss_out_ref);
        const int nwg = 16 * (inter / 1024);
        std::vector<unsigned long long> d((size_t) nwg * 8);
        CE(exl3_debug_copy_workspace(d.data(), (48ll << 20) + (256 << 10), (int64_t) d.size() * 8, st));
        CK(hipStreamSynchronize(st));
        unsigned long long t0min = ~0ull, t7max = 0;
        for (int w = 0; w < nwg; ++w) { t0min = std::min(t0min, d[(size_t) w * 8]); t7max = std::max(t7max, d[(size_t) w * 8 + 7]); }
        const char* names[7] = { "prep", "stream1", "reduce_publish", "team_wait", "prep2", "stream2", "epilogue" };
        printf(" \"timeline_us\": {\"kernel_first_entry_to_last_exit\": %.2f", (double) (t7max - t0min) * 0.01);
        {
            std::vector<double> v; for (int w = 0; w < nwg; ++w) v.push_back((double) (d[(size_t) w * 8] - t0min) * 0.01);
            std::sort(v.begin(), v.end());
            printf(", \"entry_offset\": [%.2f, %.2f, %.2f]", v[nwg / 10], v[nwg / 2], v[nwg - 1]);
        }
        for (int ph = 0; ph < 7; ++ph)
        {
            std::vector<double> v; for (int w = 0; w < nwg; ++w) v.push_back((double) (d[(size_t) w * 8 + ph + 1] - d[(size_t) w * 8 + ph]) * 0.01);
            std::sort(v.begin(), v.end());
            printf(", \"%s\": [%.2f, %.2f, %.2f, %.2f]", names[ph], v[0], v[nwg / 2], v[nwg * 9 / 10], v[nwg - 1]);
        }
        {
            std::vector<double> v; for (int w = 0; w < nwg; ++w) v.push_back((double) (d[(size_t) w * 8 + 2] - t0min) * 0.01);
            std::sort(v.begin(), v.end());
            printf(", \"stream1_done_at\": [%.2f, %.2f, %.2f]", v[0], v[nwg / 2], v[nwg - 1]);
        }
        printf("},\n");
    }
    const double t_ref = time_graph(false);
    const double t_new = time_graph(true);
    int err = -1; CE(exl3_mlp1_error(&err, st));
    const double bytes = 3.0 * (double) wwords * 4.0;
    printf(" \"us_per_mlp\": {\"three_launches\": %.2f, \"one_launch\": %.2f}, \"tb_per_s\": {\"three_launches\": %.2f, \"one_launch\": %.2f}, \"err_word_after_timing\": %d}\n",
           t_ref, t_new, bytes / (t_ref * 1e-6) / 1e12, bytes / (t_new * 1e-6) / 1e12, err);
    return 0;
}
