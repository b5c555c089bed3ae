// Internal helpers shared by the C-ABI translation units (error reporting, per-device context).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/exl3_hip.h"

void exl3_set_error(const char* fmt, ...);
int  exl3_check_launch(const char* what);

#define EXL3_CHECK_ARG(cond, ...) \
    do { if (!(cond)) { exl3_set_error(__VA_ARGS__); return EXL3_ERR_ARG; } } while (0)

#define EXL3_CHECK_HIP(expr, what) \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) { exl3_set_error("%s: %s", what, hipGetErrorString(e_)); return EXL3_ERR_HIP; } } while (0)

// Per-device context (reference: DevCtx, quant/exl3_devctx.cuh:8-19).  workspace: fp32 split-k partial slabs.
struct Exl3DevCtx
{
    bool   ready;
    int    num_cus;
    float* workspace;          // EXL3_WORKSPACE_BYTES
    uint32_t* tickets;         // zero-initialised, every kernel leaves it zeroed
    int    ws_toggle;          // slab-writing launches alternate between two workspace regions: a launch may read its predecessor's slabs
                               // (GEMV ACT mode) while its own workgroups already write theirs
    // The workspace is one per device: when the issuing stream changes, the new stream is ordered behind the previous one (event edge), so
    // launches that use the workspace never overlap across streams.  A change in the middle of a graph capture cannot be ordered this way and
    // is refused (exl3_get_ctx).
    hipStream_t last_stream; bool last_stream_valid; hipEvent_t switch_event;
};
#define EXL3_WORKSPACE_BYTES (64ll << 20)
#define EXL3_NUM_TICKETS 65536
#define EXL3_WS_XH_OFFSET (50ll << 20)         // rotated activations of one generation-3 pass (exl3_gemv.hip), up to the end of the workspace
#define EXL3_WS_XH_BYTES (14ll << 20)
#define EXL3_WS_REGION_BYTES (24ll << 20)     // two slab regions [0, 24) and [24, 48) MiB; diagnostics builds use the tail

// Returns nullptr (and sets the error) if the context cannot be created (e.g. stream capturing before exl3_init).
Exl3DevCtx* exl3_get_ctx(hipStream_t stream);
