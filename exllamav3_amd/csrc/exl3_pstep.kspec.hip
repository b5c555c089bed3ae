// Persistent decode step (generation 5): the KERNEL INSTANTIATIONS of one bits-per-weight value of the layers (compiled once per K with -DG2_K=K, like the GEMV
// generations: the units build in parallel).  Kernel body: exl3_pstep_kernel.cuh; planner + C ABI: exl3_pstep.hip, which reaches these through ps_kernel_set_k<K>().
//
// exl3_pstep_kernel<K, K2, KH, CB, ATT>:
//   K   the layers' bits per weight (this unit's G2_K);
//   K2  = K, or K + 1: the second width of a fractional-bpw checkpoint (3.5 / 4.5 bpw ...: the reference's allocator bumps whole qgroups by one bit,
//       conversion/allocation.py:131-141) -- the streaming loop then runs the ops in runs of equal width;
//   KH  the lm_head's: K, or 6 (the head of a real checkpoint stays at 6 bits: conversion defaults);
//   CB  the codebook of every tensor: 2 mul1 (default of new conversions, conversion/convert_model.py:49), 0 3INST (every older public EXL3 quant), 1 mcg
//       (quant/codebook.cuh:56-90);
//   ATT the decode attention inside o_proj's preparation;
//   TP  one rank of a tensor-parallel job (the exchange inside the step).
// Instantiated: mul1 and 3INST: (K, K, K) and (K, K, 6) for K in {2, 3, 4, 5, 6, 8}; mul1 mixed (K, K + 1, 6) for K in {2, 3, 4, 5}; mcg: K in {3, 4}; tensor-parallel
// ranks: the mul1 uniform-width sets.
#include "exl3_pstep_kernel.cuh"
#include "exl3_pstep_launch.h"

#ifndef G2_K
#error "compile with -DG2_K=<bits per weight>"
#endif

#if G2_K == 2 || G2_K == 3 || G2_K == 4 || G2_K == 5 || G2_K == 6 || G2_K == 8
#define PS_HAVE_K 1
#else
#define PS_HAVE_K 0
#endif

#if PS_HAVE_K
namespace
{
constexpr int KL = G2_K;
// X(K2, KH, CB) over this unit's instantiations
#if G2_K == 6
#define PS_SETS_UNIFORM(X, CB) X(6, 6, CB)
#else
#define PS_SETS_UNIFORM(X, CB) X(G2_K, G2_K, CB) X(G2_K, 6, CB)
#endif
#if G2_K >= 2 && G2_K <= 5
#define PS_SETS_MIXED(X) X(G2_K + 1, 6, EXL3_CB_MUL1)
#else
#define PS_SETS_MIXED(X)
#endif
#if G2_K == 3 || G2_K == 4
#define PS_SETS_MCG(X) PS_SETS_UNIFORM(X, EXL3_CB_MCG)
#else
#define PS_SETS_MCG(X)
#endif
#ifdef PS_ONLY_MUL1
#define PS_SETS(X) PS_SETS_UNIFORM(X, EXL3_CB_MUL1)
#else
#define PS_SETS(X) PS_SETS_UNIFORM(X, EXL3_CB_MUL1) PS_SETS_MIXED(X) PS_SETS_UNIFORM(X, EXL3_CB_3INST) PS_SETS_MCG(X)
#endif

// (tensor-parallel instantiations: mul1, one width in the layers)
template <int K2, int CB> constexpr bool ps_has_tp() { return CB == EXL3_CB_MUL1 && K2 == KL; }

template <int K2, int KH, int CB>
int ps_prepare_one(bool att, bool tp, int* occupancy)
{
    const void* f = nullptr;
    if (!tp) f = att ? (const void*) exl3_pstep_kernel<KL, K2, KH, CB, true, false> : (const void*) exl3_pstep_kernel<KL, K2, KH, CB, false, false>;
    else if constexpr (ps_has_tp<K2, CB>()) f = att ? (const void*) exl3_pstep_kernel<KL, K2, KH, CB, true, true> : (const void*) exl3_pstep_kernel<KL, K2, KH, CB, false, true>;
    if (!f) return 1;
    EXL3_CHECK_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, PS_LDS_BYTES), "hipFuncSetAttribute(pstep)");
    int nb = 0;
    *occupancy = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, PS_NT, PS_LDS_BYTES) == hipSuccess ? nb : -1;
    return EXL3_OK;
}

int ps_prepare(int K2, int KH, int cb, bool att, bool tp, int* occupancy)
{
    #define PS_X(KK2, KKH, CCB) if (K2 == (KK2) && KH == (KKH) && cb == (CCB)) return ps_prepare_one<(KK2), (KKH), (CCB)>(att, tp, occupancy);
    PS_SETS(PS_X)
    #undef PS_X
    return 1;                                            // no such instantiation
}

template <int K2, int KH, int CB>
int ps_launch_one(bool att, bool tp, int ncu, hipStream_t st, const PsArgs& args)
{
    if (!tp)
    {
        if (att) exl3_pstep_kernel<KL, K2, KH, CB, true, false><<<dim3(ncu), dim3(PS_NT), PS_LDS_BYTES, st>>>(args);
        else     exl3_pstep_kernel<KL, K2, KH, CB, false, false><<<dim3(ncu), dim3(PS_NT), PS_LDS_BYTES, st>>>(args);
        return 0;
    }
    if constexpr (ps_has_tp<K2, CB>())
    {
        if (att) exl3_pstep_kernel<KL, K2, KH, CB, true, true><<<dim3(ncu), dim3(PS_NT), PS_LDS_BYTES, st>>>(args);
        else     exl3_pstep_kernel<KL, K2, KH, CB, false, true><<<dim3(ncu), dim3(PS_NT), PS_LDS_BYTES, st>>>(args);
        return 0;
    }
    return 1;
}

int ps_launch(int K2, int KH, int cb, bool att, bool tp, int ncu, hipStream_t st, const PsArgs& args)
{
    #define PS_X(KK2, KKH, CCB) if (K2 == (KK2) && KH == (KKH) && cb == (CCB)) return ps_launch_one<(KK2), (KKH), (CCB)>(att, tp, ncu, st, args);
    PS_SETS(PS_X)
    #undef PS_X
    return 1;
}

const PsKernelSet kset = { ps_prepare, ps_launch };
}
#endif

#define PS_CAT2(a, b) a##b
#define PS_CAT(a, b) PS_CAT2(a, b)
const PsKernelSet* PS_CAT(ps_kernel_set_k, G2_K)()
{
#if PS_HAVE_K
    return &kset;
#else
    return nullptr;
#endif
}
