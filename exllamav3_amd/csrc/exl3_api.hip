// C-ABI plumbing: error reporting, per-device context, device info.
#include "exl3_api_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <mutex>

static thread_local char g_err[512] = "";

void exl3_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int exl3_check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
    {
        exl3_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return EXL3_ERR_HIP;
    }
    return EXL3_OK;
}

extern "C" const char* exl3_last_error(void) { return g_err; }
extern "C" int exl3_abi_version(void) { return EXL3_ABI_VERSION; }

#define MAX_DEVICES 64
static Exl3DevCtx g_ctx[MAX_DEVICES];
static std::mutex g_ctx_mutex;

static int init_device(int device)
{
    if (device < 0 || device >= MAX_DEVICES) { exl3_set_error("exl3_init: bad device %d", device); return EXL3_ERR_ARG; }
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    Exl3DevCtx& c = g_ctx[device];
    if (c.ready) return EXL3_OK;
    int prev = 0;
    EXL3_CHECK_HIP(hipGetDevice(&prev), "hipGetDevice");
    EXL3_CHECK_HIP(hipSetDevice(device), "hipSetDevice");
    hipDeviceProp_t prop;
    EXL3_CHECK_HIP(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    c.num_cus = prop.multiProcessorCount;
    EXL3_CHECK_HIP(hipMalloc((void**) &c.workspace, EXL3_WORKSPACE_BYTES), "hipMalloc(workspace)");
    EXL3_CHECK_HIP(hipMalloc((void**) &c.tickets, EXL3_NUM_TICKETS * sizeof(uint32_t)), "hipMalloc(tickets)");
    EXL3_CHECK_HIP(hipMemset(c.tickets, 0, EXL3_NUM_TICKETS * sizeof(uint32_t)), "hipMemset(tickets)");
    EXL3_CHECK_HIP(hipEventCreateWithFlags(&c.switch_event, hipEventDisableTiming), "hipEventCreate");
    EXL3_CHECK_HIP(hipDeviceSynchronize(), "hipDeviceSynchronize");
    c.last_stream_valid = false;
    c.ready = true;
    (void) hipSetDevice(prev);
    return EXL3_OK;
}

extern "C" int exl3_init(int device) { return init_device(device); }

Exl3DevCtx* exl3_get_ctx(hipStream_t stream)
{
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { exl3_set_error("hipGetDevice failed"); return nullptr; }
    if (device < 0 || device >= MAX_DEVICES) { exl3_set_error("bad device"); return nullptr; }
    if (!g_ctx[device].ready)
    {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone)
        {
            exl3_set_error("exl3: call exl3_init(%d) once before capturing a graph", device);
            return nullptr;
        }
        if (init_device(device) != EXL3_OK) return nullptr;
    }
    Exl3DevCtx& c = g_ctx[device];
    // the stream-switch bookkeeping is shared by every host thread that launches on this device: one lock around it (the common one-stream path takes
    // it uncontended).  Ordering covers EAGER launches only: a graph replay that uses the workspace is ordered against eager work on other streams by
    // whoever launches the graph (exl3_hip.h, "Streams").
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    if (c.last_stream_valid && c.last_stream != stream)
    {
        // another stream than the previous workspace user's: order it behind that one (once per change; nothing on the common one-stream path)
        hipStreamCaptureStatus so = hipStreamCaptureStatusNone, sn = hipStreamCaptureStatusNone;
        // (the previous stream may have been destroyed in the meantime: the query then fails and leaves an error behind -- cleared on every path below,
        // a later exl3_check_launch must not report it as a failed launch)
        if (hipStreamIsCapturing(c.last_stream, &so) != hipSuccess) so = hipStreamCaptureStatusNone;
        (void) hipGetLastError();
        if (hipStreamIsCapturing(stream, &sn) != hipSuccess) sn = hipStreamCaptureStatusNone;
        (void) hipGetLastError();
        if (so != hipStreamCaptureStatusNone)
        {
            exl3_set_error("exl3: the per-device workspace was last used by a stream that is still capturing; all launches of one graph must be issued on one stream");
            return nullptr;
        }
        if (sn == hipStreamCaptureStatusNone)
        {
            // (a previous stream that has been destroyed in the meantime fails the record: nothing left to order against)
            if (hipEventRecord(c.switch_event, c.last_stream) == hipSuccess) (void) hipStreamWaitEvent(stream, c.switch_event, 0);
            (void) hipGetLastError();
        }
        // (a capturing new stream after an eager one: the capture's replays are ordered by whoever launches the graph)
    }
    c.last_stream = stream; c.last_stream_valid = true;
    return &c;
}

extern "C" int exl3_device_info(int device, int* num_cus, int* gfx_arch, int64_t* hbm_bytes)
{
    hipDeviceProp_t prop;
    EXL3_CHECK_HIP(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    if (num_cus) *num_cus = prop.multiProcessorCount;
    if (gfx_arch)
    {
        int a = 0;
        const char* p = strstr(prop.gcnArchName, "gfx");
        if (p) a = (int) strtol(p + 3, nullptr, 10);    // "gfx950" -> 950
        *gfx_arch = a;
    }
    if (hbm_bytes) *hbm_bytes = (int64_t) prop.totalGlobalMem;
    return EXL3_OK;
}

// Diagnostics: copy a range of the per-device split-k workspace into a caller buffer (device or host pointer).  Used by
// tools/gemv_timeline.py with a -DG2_TIMING build, whose workgroups leave phase timestamps in the workspace tail.
extern "C" int exl3_debug_copy_workspace(void* dst, int64_t byte_offset, int64_t nbytes, void* stream)
{
    Exl3DevCtx* ctx = exl3_get_ctx((hipStream_t) stream);
    if (!ctx) return EXL3_ERR_INIT;
    EXL3_CHECK_ARG(dst && byte_offset >= 0 && nbytes >= 0 && byte_offset + nbytes <= EXL3_WORKSPACE_BYTES, "debug_copy_workspace: bad range");
    EXL3_CHECK_HIP(hipMemcpyAsync(dst, (const char*) ctx->workspace + byte_offset, (size_t) nbytes, hipMemcpyDefault, (hipStream_t) stream), "hipMemcpyAsync");
    return EXL3_OK;
}
