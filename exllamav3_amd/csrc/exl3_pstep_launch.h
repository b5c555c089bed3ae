// Persistent decode step: the seam between the planner / C ABI (exl3_pstep.hip) and the per-K kernel units (exl3_pstep.kspec.hip compiled with -DG2_K=1..8).
#pragma once
#include "exl3_pstep.cuh"

#define PS_LDS_BYTES (PS_QUADS_BYTES + PS_MISC_BYTES + PS_PART_BYTES + PS_PDEC_BYTES + PS_GATH_BYTES + PS_ATT_BYTES)
static_assert(PS_LDS_BYTES <= 160 * 1024, "persistent step: LDS map exceeds a CU's 160 KiB");

struct PsKernelSet
{
    // both return 0, or 1 if the unit has no kernel for (second layer width K2, head width KH, codebook cb)
    // (tp: the instantiation of a tensor-parallel rank)
    int (*prepare)(int K2, int KH, int cb, bool att, bool tp, int* occupancy);      // dynamic-LDS attribute + workgroups of this kernel that fit one CU (-1: query failed); < 0: HIP error
    int (*launch)(int K2, int KH, int cb, bool att, bool tp, int ncu, hipStream_t st, const PsArgs& args);
};
// nullptr: no instantiation for layers of that width (K = 1, 7)
const PsKernelSet* ps_kernel_set_k1(); const PsKernelSet* ps_kernel_set_k2(); const PsKernelSet* ps_kernel_set_k3(); const PsKernelSet* ps_kernel_set_k4();
const PsKernelSet* ps_kernel_set_k5(); const PsKernelSet* ps_kernel_set_k6(); const PsKernelSet* ps_kernel_set_k7(); const PsKernelSet* ps_kernel_set_k8();
