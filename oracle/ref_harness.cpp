// ORACLE / TEST INFRASTRUCTURE — not product code.
//
// Thin extern "C" harness around the REFERENCE's own CPU implementation of the EXL3 tile format
// (exllamav3_ext/cpu/moe_mul1.cpp, mul1 codebook, scalar tier :1087-1116 selected with
// EXL3_MOE_CPU_MAX_ISA=scalar).  The reference source is compiled from where it lies under
// /root/reference by oracle/build_ref.sh; nothing from it is copied into this repository.
// Output goes to oracle/_ref/ (git-ignored, travels to the GPU box like our own .so files).
//
// The reference entry points take at::Tensor; this file only wraps caller-owned host buffers with
// at::from_blob and forwards them.  One "gateless relu2" expert = up-projection -> relu^2 -> down-projection,
// i.e. two chained EXL3 linears with both Hadamards and suh/svh, which is what tests/test_oracle_pins.py
// compares the numpy oracle against.

#include <ATen/ATen.h>
#include <vector>
#include <cstdint>
#include "cpu/moe_mul1.h"      // resolved with -I/root/reference/exllamav3/exllamav3_ext

extern "C" {

// up: (hidden -> interm), down: (interm -> hidden).  trellis int16 [k/16][n/16][16K]; suh/svh fp16.
// x fp16 [m][hidden]; out fp32 [m][hidden].  Returns 0 on success, -1 on exception.
int ref_mul1_mlp_relu2(
    const int16_t* up_trellis, const uint16_t* up_suh, const uint16_t* up_svh,
    const int16_t* down_trellis, const uint16_t* down_suh, const uint16_t* down_svh,
    int hidden, int interm, int K_up, int K_down,
    const uint16_t* x, int m, float* out, int threads)
{
    try {
        auto i16 = at::TensorOptions().dtype(at::kShort);
        auto f16 = at::TensorOptions().dtype(at::kHalf);
        auto f32 = at::TensorOptions().dtype(at::kFloat);
        auto i64 = at::TensorOptions().dtype(at::kLong);
        at::Tensor ut = at::from_blob((void*) up_trellis, {hidden / 16, interm / 16, 16 * K_up}, i16);
        at::Tensor us = at::from_blob((void*) up_suh, {hidden}, f16);
        at::Tensor uv = at::from_blob((void*) up_svh, {interm}, f16);
        at::Tensor dt = at::from_blob((void*) down_trellis, {interm / 16, hidden / 16, 16 * K_down}, i16);
        at::Tensor ds = at::from_blob((void*) down_suh, {interm}, f16);
        at::Tensor dv = at::from_blob((void*) down_svh, {hidden}, f16);
        std::vector<at::Tensor> none;
        int64_t h = exl3_moe_cpu_make_layer(none, none, none, {ut}, {us}, {uv}, {dt}, {ds}, {dv},
                                            none, none, none, /*activation relu2*/ 2, 0.0, /*swizzled*/ 0);
        at::Tensor xt = at::from_blob((void*) x, {m, hidden}, f16);
        at::Tensor sel = at::zeros({m, 1}, i64);
        at::Tensor wts = at::ones({m, 1}, f16);
        at::Tensor o = at::from_blob((void*) out, {m, hidden}, f32);
        exl3_moe_cpu_forward(h, xt, sel, wts, o, threads);
        exl3_moe_cpu_free_layer(h);
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "ref_mul1_mlp_relu2: %s\n", e.what());
        return -1;
    }
}

// Build-once variant for timing (bench.py cpu_baseline): the layer (the reference's re-laid-out copy of the packed tensors) is made ONCE,
// forward calls are timed on their own.  Returns a handle (>= 0) or -1.
long long ref_mul1_mlp_make(const int16_t* up_trellis, const uint16_t* up_suh, const uint16_t* up_svh,
                            const int16_t* down_trellis, const uint16_t* down_suh, const uint16_t* down_svh,
                            int hidden, int interm, int K_up, int K_down)
{
    try {
        auto i16 = at::TensorOptions().dtype(at::kShort);
        auto f16 = at::TensorOptions().dtype(at::kHalf);
        at::Tensor ut = at::from_blob((void*) up_trellis, {hidden / 16, interm / 16, 16 * K_up}, i16);
        at::Tensor us = at::from_blob((void*) up_suh, {hidden}, f16);
        at::Tensor uv = at::from_blob((void*) up_svh, {interm}, f16);
        at::Tensor dt = at::from_blob((void*) down_trellis, {interm / 16, hidden / 16, 16 * K_down}, i16);
        at::Tensor ds = at::from_blob((void*) down_suh, {interm}, f16);
        at::Tensor dv = at::from_blob((void*) down_svh, {hidden}, f16);
        std::vector<at::Tensor> none;
        return (long long) exl3_moe_cpu_make_layer(none, none, none, {ut}, {us}, {uv}, {dt}, {ds}, {dv}, none, none, none, 2, 0.0, 0);
    } catch (const std::exception& e) {
        fprintf(stderr, "ref_mul1_mlp_make: %s\n", e.what());
        return -1;
    }
}

int ref_mul1_mlp_forward(long long handle, const uint16_t* x, int m, int hidden, float* out, int threads)
{
    try {
        auto f16 = at::TensorOptions().dtype(at::kHalf);
        auto f32 = at::TensorOptions().dtype(at::kFloat);
        auto i64 = at::TensorOptions().dtype(at::kLong);
        at::Tensor xt = at::from_blob((void*) x, {m, hidden}, f16);
        at::Tensor sel = at::zeros({m, 1}, i64);
        at::Tensor wts = at::ones({m, 1}, f16);
        at::Tensor o = at::from_blob((void*) out, {m, hidden}, f32);
        exl3_moe_cpu_forward((int64_t) handle, xt, sel, wts, o, threads);
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "ref_mul1_mlp_forward: %s\n", e.what());
        return -1;
    }
}

void ref_mul1_mlp_free(long long handle) { try { exl3_moe_cpu_free_layer((int64_t) handle); } catch (...) {} }

}
