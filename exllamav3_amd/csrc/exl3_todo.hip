// Entry points declared in include/exl3_hip.h whose kernels are not written yet.  They fail loudly.
#include "exl3_api_internal.h"

#define NOT_YET(name) do { exl3_set_error(name ": not implemented yet in this build"); return EXL3_ERR_ARG; } while (0)

extern "C" int exl3_reconstruct_had(void*, const void*, const void*, const void*, int, int, int, int, int64_t, int64_t, void*) { NOT_YET("exl3_reconstruct_had"); }
extern "C" int exl3_hgemm(const void*, const void*, void*, int, int, int, int64_t, int, void*) { NOT_YET("exl3_hgemm"); }
extern "C" int exl3_rope(const void*, void*, const void*, void*, const float*, int, int, int, int, int, uint32_t, const int32_t*, const int32_t*,
                         int, float, const void*, const void*, float, float, void*) { NOT_YET("exl3_rope"); }
extern "C" int exl3_quant_cache_cont(const void*, void*, void*, int64_t, int, int, void*) { NOT_YET("exl3_quant_cache_cont"); }
extern "C" int exl3_dequant_cache_cont(const void*, const void*, void*, int64_t, int, int, void*) { NOT_YET("exl3_dequant_cache_cont"); }
extern "C" int exl3_quant_cache_paged(const void*, void*, void*, const void*, void*, void*, const int32_t*, const int32_t*, int, int, int, int, int, int, int, void*) { NOT_YET("exl3_quant_cache_paged"); }
extern "C" int exl3_dequant_cache_paged(const void*, const void*, void*, const void*, const void*, void*, const int32_t*, const int32_t*, int, int, int, int, int, int, void*) { NOT_YET("exl3_dequant_cache_paged"); }
