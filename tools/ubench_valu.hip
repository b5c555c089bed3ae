// VALU instruction-rate microbenchmark for gfx950 (standalone; build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o ubench_valu).
// Settles SURVEY.md hard part 1: the per-weight ALU budget of the EXL3 decode (is v_mul_lo_u32 full rate? what do
// v_sad_u8 / v_perm_b32 / v_alignbit / v_bfe / v_pk_add_f16 / v_mad_u32_u24 cost?).
// Each kernel runs a long dependent-free stream of one instruction (8 independent chains per lane); we report
// wave-instructions per cycle per SIMD derived from wall time at the measured clock (hipDeviceProp clockRate is
// nominal, so the table is relative to v_add_u32 = full rate).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 4096

#define DEF_KERNEL(NAME, ASM) \
__global__ void NAME(uint32_t* out, uint32_t seed) { \
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u; \
    uint32_t b = seed * 0x9E3779B9u + 12345u, c = seed ^ 0x5bd1e995u; \
    for (int i = 0; i < ITERS; ++i) { \
        asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
    } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; }

#define A_ADD(r)      "v_add_u32 " #r ", " #r ", %8\n"
#define A_MULLO(r)    "v_mul_lo_u32 " #r ", " #r ", %8\n"
#define A_MULU24(r)   "v_mul_u32_u24 " #r ", " #r ", %8\n"
#define A_MADU24(r)   "v_mad_u32_u24 " #r ", " #r ", %8, %9\n"
#define A_SAD(r)      "v_sad_u8 " #r ", " #r ", %8, %9\n"
#define A_SADHI(r)    "v_sad_hi_u8 " #r ", " #r ", %8, %9\n"
#define A_PERM(r)     "v_perm_b32 " #r ", " #r ", %8, %9\n"
#define A_ALIGN(r)    "v_alignbit_b32 " #r ", " #r ", %8, 12\n"
#define A_BFE(r)      "v_bfe_u32 " #r ", " #r ", 4, 16\n"
#define A_AND(r)      "v_and_b32 " #r ", " #r ", %8\n"
#define A_ANDOR(r)    "v_and_or_b32 " #r ", " #r ", %8, %9\n"
#define A_XOR(r)      "v_xor_b32 " #r ", " #r ", %8\n"
#define A_PKADD(r)    "v_pk_add_f16 " #r ", " #r ", %8\n"
#define A_PKFMA(r)    "v_pk_fma_f16 " #r ", " #r ", %8, %9\n"
#define A_PKMUL16(r)  "v_pk_mul_lo_u16 " #r ", " #r ", %8\n"
#define A_LSHLADD(r)  "v_lshl_add_u32 " #r ", " #r ", 3, %8\n"
#define A_FMA(r)      "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define A_DOT2(r)     "v_dot2_f32_f16 " #r ", %8, %9, " #r "\n"
#define A_LSHR64(r)   "v_lshrrev_b32 " #r ", 5, " #r "\n"
#define A_MADU64(r)   "v_mul_hi_u32 " #r ", " #r ", %8\n"
#define A_BFI(r)      "v_bfi_b32 " #r ", %8, " #r ", %9\n"
#define A_XAD(r)      "v_xad_u32 " #r ", " #r ", %8, %9\n"

DEF_KERNEL(k_add, A_ADD)
DEF_KERNEL(k_mullo, A_MULLO)
DEF_KERNEL(k_mulu24, A_MULU24)
DEF_KERNEL(k_madu24, A_MADU24)
DEF_KERNEL(k_sad, A_SAD)
DEF_KERNEL(k_sadhi, A_SADHI)
DEF_KERNEL(k_perm, A_PERM)
DEF_KERNEL(k_align, A_ALIGN)
DEF_KERNEL(k_bfe, A_BFE)
DEF_KERNEL(k_and, A_AND)
DEF_KERNEL(k_andor, A_ANDOR)
DEF_KERNEL(k_xor, A_XOR)
DEF_KERNEL(k_pkadd, A_PKADD)
DEF_KERNEL(k_pkfma, A_PKFMA)
DEF_KERNEL(k_pkmul16, A_PKMUL16)
DEF_KERNEL(k_lshladd, A_LSHLADD)
DEF_KERNEL(k_fma, A_FMA)
DEF_KERNEL(k_dot2, A_DOT2)
DEF_KERNEL(k_lshr, A_LSHR64)
DEF_KERNEL(k_mulhi, A_MADU64)
DEF_KERNEL(k_bfi, A_BFI)
DEF_KERNEL(k_xad, A_XAD)

typedef void (*kern_t)(uint32_t*, uint32_t);

static double run(kern_t k, uint32_t* d, int blocks, int threads)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 1u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r)
    {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, (uint32_t) (r + 2));
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    int threads = 256, blocks = cus * 8;             // 8 waves per SIMD
    uint32_t* d; hipMalloc(&d, (size_t) blocks * threads * 4);
    struct { const char* name; kern_t k; } tests[] = {
        {"v_add_u32", k_add}, {"v_mul_lo_u32", k_mullo}, {"v_mul_u32_u24", k_mulu24}, {"v_mad_u32_u24", k_madu24},
        {"v_sad_u8", k_sad}, {"v_sad_hi_u8", k_sadhi}, {"v_perm_b32", k_perm}, {"v_alignbit_b32", k_align},
        {"v_bfe_u32", k_bfe}, {"v_and_b32", k_and}, {"v_and_or_b32", k_andor}, {"v_xor_b32", k_xor},
        {"v_pk_add_f16", k_pkadd}, {"v_pk_fma_f16", k_pkfma}, {"v_pk_mul_lo_u16", k_pkmul16}, {"v_lshl_add_u32", k_lshladd},
        {"v_fma_f32", k_fma}, {"v_dot2_f32_f16", k_dot2}, {"v_lshrrev_b32", k_lshr}, {"v_mul_hi_u32", k_mulhi},
        {"v_bfi_b32", k_bfi}, {"v_xad_u32", k_xad},
    };
    double base = 0;
    printf("{\"device\": \"%s\", \"cus\": %d, \"results\": [\n", p.gcnArchName, cus);
    int n = sizeof(tests) / sizeof(tests[0]);
    for (int i = 0; i < n; ++i)
    {
        double ms = run(tests[i].k, d, blocks, threads);
        if (i == 0) base = ms;
        // wave-instructions: blocks*4 waves * ITERS * 8 ; per SIMD: / (cus*4)
        double winst_per_simd = (double) blocks * 4 * ITERS * 8 / (cus * 4.0);
        double cyc_at_2p4 = ms * 1e-3 * 2.4e9 / winst_per_simd;
        printf("  {\"inst\": \"%s\", \"ms\": %.4f, \"rel_to_add\": %.2f, \"cycles_per_wave_inst_at_2.4GHz\": %.2f}%s\n",
               tests[i].name, ms, ms / base, cyc_at_2p4, i + 1 < n ? "," : "");
    }
    printf("]}\n");
    return 0;
}
