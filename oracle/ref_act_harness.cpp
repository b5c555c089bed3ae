// ORACLE / TEST INFRASTRUCTURE -- not product code.
//
// extern "C" harness around the REFERENCE's own activation kernels (exllamav3_ext/activation_kernels.cuh:142-254 act_mul_kernel_h / act_mul_kernel_f
// with the activations of :9-128), compiled for the host from where they lie under /root/reference (oracle/build_ref.sh ->
// oracle/_ref/libexl3_ref_act.so) on top of oracle/cuda_host_shim.h.  Nothing from the reference is copied into this repository; the headers are
// #included by path.  Built with -DUSE_ROCM so that compat.cuh:12-18 takes its exp-based tanh_opt (the branch the reference itself uses on AMD; the
// other branch is a PTX instruction).  The fp16 transcendental intrinsics (hexp, hrcp) are the correctly rounded functions here; the device versions
// are approximations with 1-2 ulp of error, so tests/test_oracle_pins.py compares to a few fp16 ulps, not bit for bit.
#include "cuda_host_shim.h"
#include <algorithm>
#include <math.h>

// ---- the CUDA intrinsics these kernels use beyond the shim's core set (documented semantics of each) ------------------------------------------
struct float2 { float x, y; };
static inline float2 __half22float2(half2 v) { float2 r; r.x = shim::h2f(v.x); r.y = shim::h2f(v.y); return r; }
static inline half2 __float22half2_rn(float2 v) { half2 r; r.x = shim::f2h_rn(v.x); r.y = shim::f2h_rn(v.y); return r; }
static inline half2 __float2half2_rn(float f) { half2 r; r.x = shim::f2h_rn(f); r.y = r.x; return r; }
static inline half __hneg(half a) { half r; r.bits = a.bits ^ 0x8000; return r; }
static inline half2 __hneg2(half2 a) { half2 r; r.x = __hneg(a.x); r.y = __hneg(a.y); return r; }
static inline half shim_round_d(double d) { return shim::f2h_rn((float) d); }        // |d| well inside float range; double -> float -> half
static inline half hexp(half a) { return shim_round_d(exp((double) shim::h2f(a))); }
static inline half hrcp(half a) { return shim_round_d(1.0 / (double) shim::h2f(a)); }
static inline half2 h2exp(half2 a) { half2 r; r.x = hexp(a.x); r.y = hexp(a.y); return r; }
static inline half2 h2rcp(half2 a) { half2 r; r.x = hrcp(a.x); r.y = hrcp(a.y); return r; }
static inline half shim_hmax(half a, half b) { return shim::h2f(a) >= shim::h2f(b) ? a : b; }
static inline half shim_hmin(half a, half b) { return shim::h2f(a) <= shim::h2f(b) ? a : b; }
static inline half2 __hmax2(half2 a, half2 b) { half2 r; r.x = shim_hmax(a.x, b.x); r.y = shim_hmax(a.y, b.y); return r; }
static inline half2 __hmin2(half2 a, half2 b) { half2 r; r.x = shim_hmin(a.x, b.x); r.y = shim_hmin(a.y, b.y); return r; }
#define __expf(x) expf(x)                  /* glibc declares a symbol of this name: a macro instead of a function */
static inline float __fdividef(float a, float b) { return a / b; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float min(float a, float b) { return a < b ? a : b; }
// activation_kernels.cuh:375-411 (a projection kernel this harness never runs) calls the block reduction of reduction.cuh; a declaration is all it needs
static inline float block_reduce_sum_broadcast_f(float v, int) { return v; }

#define NUM_THREADS 256
#define NUM_THREADS_P 1024
#define ACT_SILU 0
#define ACT_GELU 1
#define ACT_RELU2 2
#define ACT_SILU_OAI 3
#define ACT_RELU 4
#include "compat.cuh"                      // -I/root/reference/exllamav3/exllamav3_ext : tanh_opt
#include "activation_kernels.cuh"

template <int ACT> static void run(int in_fp32, const void* x, const void* y, half* z, float act_limit, size_t numel)
{
    const size_t pairs = numel / 2;
    gridDim.x = (unsigned) ((pairs + NUM_THREADS - 1) / NUM_THREADS); blockDim.x = NUM_THREADS;
    for (unsigned b = 0; b < gridDim.x; ++b)
        for (unsigned t = 0; t < NUM_THREADS; ++t)
        {
            blockIdx.x = b; threadIdx.x = t;
            if (in_fp32) act_mul_kernel_f<ACT>((const float*) x, (const float*) y, z, act_limit, numel);
            else act_mul_kernel_h<ACT>((const half*) x, (const half*) y, z, act_limit, numel);
        }
}

extern "C" {

// the reference's attention-gate kernels (activation_kernels.cuh:300-367) run thread by thread: kind 0 mul_sigmoid_kernel_h (y [numel]),
// 1 mul_sigmoid_broadcast_kernel_h, 2 mul_softplus_broadcast_kernel_h (y [numel / dim]); x is updated in place
int ref_mul_gate(int kind, uint16_t* x, const uint16_t* y, size_t numel, size_t dim)
{
    if (numel % 2) return -1;
    const size_t pairs = numel / 2;
    gridDim.x = (unsigned) ((pairs + NUM_THREADS - 1) / NUM_THREADS); blockDim.x = NUM_THREADS;
    for (unsigned b = 0; b < gridDim.x; ++b)
        for (unsigned t = 0; t < NUM_THREADS; ++t)
        {
            blockIdx.x = b; threadIdx.x = t;
            if (kind == 0) mul_sigmoid_kernel_h((half*) x, (const half*) y, numel);
            else if (kind == 1) mul_sigmoid_broadcast_kernel_h((half*) x, (const half*) y, numel, dim);
            else if (kind == 2) mul_softplus_broadcast_kernel_h((half*) x, (const half*) y, numel, dim);
            else return -2;
        }
    return 0;
}

// act: the reference's ACT_* codes (activation.cu:14-18).  x, y: fp16 or fp32 [numel]; z: fp16 [numel]
int ref_act_mul(int act, int in_fp32, const void* x, const void* y, uint16_t* z, float act_limit, size_t numel)
{
    if (numel % 2) return -1;
    half* zh = (half*) z;
    switch (act)
    {
        case ACT_SILU: run<ACT_SILU>(in_fp32, x, y, zh, act_limit, numel); break;
        case ACT_GELU: run<ACT_GELU>(in_fp32, x, y, zh, act_limit, numel); break;
        case ACT_RELU2: run<ACT_RELU2>(in_fp32, x, y, zh, act_limit, numel); break;
        case ACT_SILU_OAI: run<ACT_SILU_OAI>(in_fp32, x, y, zh, act_limit, numel); break;
        case ACT_RELU: run<ACT_RELU>(in_fp32, x, y, zh, act_limit, numel); break;
        default: return -2;
    }
    return 0;
}

}
