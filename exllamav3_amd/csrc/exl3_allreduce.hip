// Latency-grade all-reduce for tensor-parallel DECODE on one xGMI node: one-shot push over IPC-mapped peer buffers, fused with the residual add.
//
// Replaces, for the 16-32 KB (tokens x hidden fp32) messages after o_proj / down_proj: TPBackendNCCL.all_reduce (model/model_tp_backend.py:119-126)
// + the reference's native small-message all-reduce (exllamav3_ext/parallel/all_reduce.cu:18-232, a ring over pinned host memory for PCIe boxes).
// On MI355X every pair of the 8 GPUs has a direct xGMI link, so the minimum-latency algorithm is ONE hop: every rank writes its partial sums
// straight into every peer's receive buffer and then adds up the W partials it received, locally, in rank order (so all ranks get the same bits).
// No RCCL on the decode path (RCCL stays for the 64-128 MB prefill messages where ring bandwidth matters).
//
// Transport = data-tagged granules (cdna_hip_programming.md Guideline 16 R2 / MI355X_MICROARCH.md "handoff-1to1"): every fp32 value travels as one
// naturally aligned 8-byte {value, tag} written by ONE system-scope store; the receiver polls the granule itself until the tag equals the current
// epoch.  No separate flag, no fence: an 8-byte store is never torn, and a stale granule carries an old tag.  Tags are per-call epochs kept in
// device memory (the kernel increments them itself, so a captured hipGraph replays correctly); two slot sets alternate by epoch parity: a rank can
// only be one call ahead of a peer (it needs the peer's push of call n+1, which the peer issues after it finished reading call n).
//
// Every spin is bounded: after EXL3_AR_SPIN_LIMIT polls the kernel raises the error word and writes NaN for the elements it could not complete (never
// hangs the GPU, never lets a stale or partial sum through); the host polls the word collectively (TPBackendRCCL.poll_ipc_allreduce).
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include "exl3_glue_device.cuh"
#include <mutex>
#include <string.h>

#define EXL3_AR_MAX_RANKS 8
#define EXL3_AR_SPIN_LIMIT (1 << 22)
#define EXL3_AR_MAX_WGS 512        // launch bound: the grid must be co-resident (every workgroup waits for its peers' twin)

struct ArGranule { float v; uint32_t tag; };

struct ArCtx
{
    int world, rank, device;
    size_t max_elems;                 // fp32 elements per message
    size_t bytes;                     // size of every rank's buffer
    char* own;                        // this rank's receive buffer (fine-grained device memory)
    char* peer[EXL3_AR_MAX_RANKS];    // peer[r] = rank r's buffer as mapped into this process (peer[rank] == own)
    bool opened[EXL3_AR_MAX_RANKS];
    hipIpcMemHandle_t handle;
};

// buffer layout: [header 256 B: epoch (u32), error (u32)] [slot set 0][slot set 1]; slot set = [source rank][max_elems] granules
#define AR_HDR 256
__host__ __device__ __forceinline__ size_t ar_slot_off(size_t max_elems, int world, int set, int src)
{
    return AR_HDR + ((size_t) set * world + src) * max_elems * sizeof(ArGranule);
}

struct ArArgs
{
    char* own; char* peer[EXL3_AR_MAX_RANKS];
    int world, rank; size_t max_elems;
    const float* y;        // this rank's partial sums [m][hidden] fp32 (null when slabs are given)
    const float* slabs; int S; const half_t* svh;     // ... or the deferred split-k slabs [hidden/128][S][m][128] of the row-sharded linear that
                                                      // produced them + its svh: the task finishes them itself (slab sum, out-Hadamard, x svh in
                                                      // fp32 = exl3_gemv_reduce_kernel's fp32 path) -- no reduce launch in front of the all-reduce
    float* y_out;          // optional: the reduced fp32 tensor (plain all_reduce semantics)
    half_t* resid;         // optional: fp16 residual stream, resid += sum (glue_resid semantics, norm.cu:193-218 rounding)
    float* ss_part;        // optional (with resid): per-128-block sums of squares of the new residual [m][hidden/128]
    long long* resid_fx;   // optional: the fx pipeline's residual stream, a 64-bit fixed-point accumulator [m][hidden] (value * 2^32): resid_fx += sum.
                           // Every element belongs to exactly one task of this launch and the sum over the ranks is formed in rank order, so the
                           // accumulators stay bit-identical on every rank; a timed-out (NaN) sum poisons the accumulator (exl3_gemv_args.h)
    int m, hidden;
};

__device__ __forceinline__ void ar_store(ArGranule* p, float v, uint32_t tag)
{
    union { ArGranule g; uint64_t u; } c; c.g.v = v; c.g.tag = tag;
    __hip_atomic_store((uint64_t*) p, c.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool ar_load(const ArGranule* p, uint32_t tag, float& v)
{
    union { ArGranule g; uint64_t u; } c;
    c.u = __hip_atomic_load((const uint64_t*) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    v = c.g.v;
    return c.g.tag == tag;
}

// one 32-lane half-wave per (row, 128-block), 4 values per lane -- the task shape of glue_resid_kernel, so the fused residual add and the
// per-block sums of squares are the same arithmetic
__global__ __launch_bounds__(256)
void ar_push_reduce_kernel(ArArgs a)
{
    const int tid = threadIdx.x, l = tid & 31, hw = tid >> 5;
    const int nblk = a.hidden >> 7;
    const int tasks = a.m * nblk;
    const int t = blockIdx.x * 8 + hw;
    const bool act = t < tasks;
    const int row = act ? t / nblk : 0, blk = act ? t % nblk : 0;
    uint32_t* hdr = (uint32_t*) a.own;
    // epoch of this call: every workgroup reads the same value; the last workgroup to finish publishes epoch + 1 (arrival counter in the header)
    const uint32_t epoch = __hip_atomic_load(hdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int set = (int) (epoch & 1u);
    const size_t e0 = (size_t) row * a.hidden + (size_t) blk * 128 + 4 * l;
    float4_t mine = { 0.f, 0.f, 0.f, 0.f };
    if (a.slabs)
    {
        const half4_t sc = ((const half4_t*) (a.svh + blk * 128))[l];
        const SlabRef sr = { a.slabs, a.S };
        const float4_t v = slab_sum(sr, blk, row, a.m, l);                 // all 32 lanes take part in the Hadamard butterflies
        float h0 = v.x, h1 = v.y, h2 = v.z, h3 = v.w;
        had128_f32x4(h0, h1, h2, h3, l);
        h0 *= HAD_R_SCALE_128; h1 *= HAD_R_SCALE_128; h2 *= HAD_R_SCALE_128; h3 *= HAD_R_SCALE_128;
        mine = float4_t{ h0 * (float) sc.x, h1 * (float) sc.y, h2 * (float) sc.z, h3 * (float) sc.w };
    }
    else if (act) mine = *((const float4_t*) (a.y + e0));
    // 1. push this rank's 4 values to every rank's slot [set][rank] (own buffer included: one code path, and the local copy is what rank r adds)
    if (act)
    {
        for (int p = 0; p < a.world; ++p)
        {
            ArGranule* dst = (ArGranule*) (a.peer[p] + ar_slot_off(a.max_elems, a.world, set, a.rank)) + e0;
            ar_store(dst + 0, mine.x, epoch); ar_store(dst + 1, mine.y, epoch); ar_store(dst + 2, mine.z, epoch); ar_store(dst + 3, mine.w, epoch);
        }
    }
    // 2. gather: the W partials of these 4 elements, summed in rank order (identical on every rank)
    float4_t sum = { 0.f, 0.f, 0.f, 0.f };
    bool timeout = false;
    if (act)
    {
        for (int s = 0; s < a.world; ++s)
        {
            const ArGranule* src = (const ArGranule*) (a.own + ar_slot_off(a.max_elems, a.world, set, s)) + e0;
            float v[4];
            #pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                int spins = 0;
                while (!ar_load(src + i, epoch, v[i]))
                {
                    if (++spins > EXL3_AR_SPIN_LIMIT) { timeout = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            sum.x += v[0]; sum.y += v[1]; sum.z += v[2]; sum.w += v[3];
        }
    }
    if (timeout)
    {
        // a peer did not deliver inside the spin bound: raise the error word AND poison the result, so the failure shows up downstream (NaN
        // residual / logits) instead of stale or partial sums silently entering the residual stream; TPBackendRCCL.poll_ipc_allreduce reads the
        // word collectively and moves every rank back to the collective library
        __hip_atomic_store(hdr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float qnan = __builtin_nanf("");
        sum = float4_t{ qnan, qnan, qnan, qnan };
    }
    // 3. consumers
    if (a.y_out && act) *((float4_t*) (a.y_out + e0)) = sum;
    if (a.resid)
    {
        half4_t r = act ? ((const half4_t*) (a.resid + e0))[0] : half4_t{ 0, 0, 0, 0 };
        r = half4_t{ f2h((float) r.x + sum.x), f2h((float) r.y + sum.y), f2h((float) r.z + sum.z), f2h((float) r.w + sum.w) };
        if (act) ((half4_t*) (a.resid + e0))[0] = r;
        if (a.ss_part)
        {
            const float r0 = (float) r.x, r1 = (float) r.y, r2 = (float) r.z, r3 = (float) r.w;
            float ss = r0 * r0;
            ss = __builtin_fmaf(r1, r1, ss); ss = __builtin_fmaf(r2, r2, ss); ss = __builtin_fmaf(r3, r3, ss);
            #pragma unroll
            for (int i = 1; i < 32; i <<= 1) ss += xor_lane(ss, i);
            if (act && l == 0) a.ss_part[(size_t) row * nblk + blk] = ss;
        }
    }
    if (a.resid_fx && act)
    {
        long long* r = a.resid_fx + e0;
        const float sv[4] = { sum.x, sum.y, sum.z, sum.w };
        #pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const long long old = r[i];
            const bool bad = !(__builtin_fabsf(sv[i]) < GEMV_FX_LIMIT) || (unsigned long long) (old + (1ll << 60)) > (2ull << 60);      // NaN / Inf / out of range, or already poisoned
            r[i] = bad ? (long long) GEMV_FX_POISON : old + __double2ll_rn((double) sv[i] * GEMV_FX_SCALE);
        }
    }
    // 4. epoch hand-over: the last workgroup of this launch bumps the epoch for the next call (arrival counter at hdr[2])
    __syncthreads();
    if (tid == 0)
    {
        const uint32_t arrived = __hip_atomic_fetch_add(hdr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (arrived == gridDim.x)
        {
            __hip_atomic_store(hdr + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(hdr, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
static std::mutex g_ar_mutex;

extern "C" int exl3_ar_create(int world, int rank, int64_t max_elems, void** ctx_out, void* handle_out /* 64 bytes */)
{
    EXL3_CHECK_ARG(ctx_out && handle_out, "ar_create: null pointer");
    EXL3_CHECK_ARG(world >= 1 && world <= EXL3_AR_MAX_RANKS && rank >= 0 && rank < world, "ar_create: 1 <= world <= 8, 0 <= rank < world");
    EXL3_CHECK_ARG(max_elems > 0 && max_elems % 4 == 0, "ar_create: max_elems must be a positive multiple of 4");
    std::lock_guard<std::mutex> lock(g_ar_mutex);
    ArCtx* c = new ArCtx();
    memset((void*) c, 0, sizeof(ArCtx));
    // the one-shot push keeps its whole grid co-resident (exl3_ar_reduce_slabs: at most 8 * EXL3_AR_MAX_WGS tasks of 128 values): a larger request is
    // clamped HERE, so that callers routing by max_elems (tp.py: numel() <= max_elems, else the collective library) never reach the launch bound
    if (max_elems > (int64_t) 8 * EXL3_AR_MAX_WGS * 128) max_elems = (int64_t) 8 * EXL3_AR_MAX_WGS * 128;
    c->world = world; c->rank = rank; c->max_elems = (size_t) max_elems;
    c->bytes = AR_HDR + (size_t) 2 * world * max_elems * sizeof(ArGranule);
    hipError_t e = hipGetDevice(&c->device);
    // fine-grained device memory: peer writes over xGMI must be visible to a kernel that is already running on the owner
    if (e == hipSuccess) e = hipExtMallocWithFlags((void**) &c->own, c->bytes, hipDeviceMallocFinegrained);
    if (e == hipSuccess) e = hipMemset(c->own, 0, c->bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipIpcGetMemHandle(&c->handle, c->own);
    if (e != hipSuccess)
    {
        exl3_set_error("ar_create: %s", hipGetErrorString(e));
        if (c->own) (void) hipFree(c->own);
        delete c;
        return EXL3_ERR_HIP;
    }
    c->peer[rank] = c->own;
    memcpy(handle_out, &c->handle, sizeof(hipIpcMemHandle_t));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    *ctx_out = c;
    return EXL3_OK;
}

extern "C" int exl3_ar_open_peer(void* ctx, int peer_rank, const void* handle /* 64 bytes */)
{
    ArCtx* c = (ArCtx*) ctx;
    EXL3_CHECK_ARG(c && handle && peer_rank >= 0 && peer_rank < c->world && peer_rank != c->rank, "ar_open_peer: bad arguments");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    EXL3_CHECK_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "ar_open_peer: hipIpcOpenMemHandle");
    c->peer[peer_rank] = (char*) p; c->opened[peer_rank] = true;
    return EXL3_OK;
}

extern "C" int exl3_ar_destroy(void* ctx)
{
    ArCtx* c = (ArCtx*) ctx;
    if (!c) return EXL3_OK;
    for (int r = 0; r < c->world; ++r) if (c->opened[r]) (void) hipIpcCloseMemHandle(c->peer[r]);
    if (c->own) (void) hipFree(c->own);
    delete c;
    return EXL3_OK;
}

// epoch counter of this rank's buffer (= number of completed reductions): ranks must stay in lockstep, TPBackendRCCL compares them collectively
extern "C" int exl3_ar_epoch(void* ctx, uint32_t* epoch_out, void* stream)
{
    ArCtx* c = (ArCtx*) ctx;
    EXL3_CHECK_ARG(c && epoch_out, "ar_epoch: null pointer");
    EXL3_CHECK_HIP(hipMemcpyAsync(epoch_out, c->own, 4, hipMemcpyDeviceToHost, (hipStream_t) stream), "ar_epoch");
    EXL3_CHECK_HIP(hipStreamSynchronize((hipStream_t) stream), "ar_epoch");
    return EXL3_OK;
}

// error word (1 = a spin timed out since the last call of this function); resets it
extern "C" int exl3_ar_error(void* ctx, void* stream)
{
    ArCtx* c = (ArCtx*) ctx;
    EXL3_CHECK_ARG(c, "ar_error: null context");
    uint32_t v = 0;
    EXL3_CHECK_HIP(hipMemcpyAsync(&v, c->own + 4, 4, hipMemcpyDeviceToHost, (hipStream_t) stream), "ar_error");
    EXL3_CHECK_HIP(hipStreamSynchronize((hipStream_t) stream), "ar_error");
    if (v) EXL3_CHECK_HIP(hipMemsetAsync(c->own + 4, 0, 4, (hipStream_t) stream), "ar_error");
    return (int) v;
}

// sum over ranks of y [m][hidden] fp32 (every rank's partial).  y_out (optional): the reduced tensor (may alias y).  resid (optional): fp16
// residual stream, resid += sum with glue_resid's rounding; ss_part (optional): per-block sums of squares of the new residual.
extern "C" int exl3_ar_reduce(void* ctx, const float* y, float* y_out, void* resid, float* ss_part, int m, int hidden, void* stream)
{
    return exl3_ar_reduce_slabs(ctx, y, nullptr, 0, nullptr, y_out, resid, ss_part, m, hidden, stream);
}

static int ar_reduce_impl(void* ctx, const float* y, const float* slabs, int S, const void* svh, float* y_out, void* resid, float* ss_part, void* resid_fx,
                          int m, int hidden, void* stream);

// The o_proj / down_proj boundary of a tensor-parallel step of the FX pipeline (llama_path.decode_step_fx under TP): this rank's partial -- dense fp32
// rows y, or the deferred split-k slabs of the row-sharded linear + its svh -- is pushed to every rank, the W partials are summed in rank order and the
// sum is ADDED into the 64-bit fixed-point residual accumulator R [m][hidden] (value * 2^32) that the GEMV_IN_FX launches read: one launch per
// boundary (linear epilogue + all-reduce + residual add), accumulators bit-identical on every rank.  reference: model/model_tp_backend.py:119-126.
extern "C" int exl3_ar_reduce_fx(void* ctx, const float* y, const float* slabs, int S, const void* svh, void* R, int m, int hidden, void* stream)
{
    EXL3_CHECK_ARG(R, "ar_reduce_fx: null accumulator");
    return ar_reduce_impl(ctx, y, slabs, S, svh, nullptr, nullptr, nullptr, R, m, hidden, stream);
}

// ... with this rank's partial given as the deferred split-k slabs of the row-sharded linear (exl3_gemv_ex* with EXL3_GEMV_OUT_DEFERRED:
// [hidden/128][S][m][128] fp32) + its svh instead of a dense tensor (y == NULL): the all-reduce launch also is that linear's epilogue.
extern "C" int exl3_ar_reduce_slabs(void* ctx, const float* y, const float* slabs, int S, const void* svh, float* y_out, void* resid, float* ss_part,
                                    int m, int hidden, void* stream)
{
    return ar_reduce_impl(ctx, y, slabs, S, svh, y_out, resid, ss_part, nullptr, m, hidden, stream);
}

static int ar_reduce_impl(void* ctx, const float* y, const float* slabs, int S, const void* svh, float* y_out, void* resid, float* ss_part, void* resid_fx,
                          int m, int hidden, void* stream)
{
    ArCtx* c = (ArCtx*) ctx;
    EXL3_CHECK_ARG(c && ((y && !slabs) || (!y && slabs && svh && S >= 1)) && (y_out || resid || resid_fx), "ar_reduce: null pointer (give y, or slabs + svh)");
    EXL3_CHECK_ARG(m >= 1 && hidden % 128 == 0 && (size_t) m * hidden <= c->max_elems, "ar_reduce: m * hidden exceeds the buffer / hidden not a multiple of 128");
    for (int r = 0; r < c->world; ++r) EXL3_CHECK_ARG(c->peer[r], "ar_reduce: peer %d not opened", r);
    ArArgs a;
    a.own = c->own; a.world = c->world; a.rank = c->rank; a.max_elems = c->max_elems;
    for (int r = 0; r < EXL3_AR_MAX_RANKS; ++r) a.peer[r] = r < c->world ? c->peer[r] : nullptr;
    a.slabs = slabs; a.S = S; a.svh = (const half_t*) svh;
    a.y = y; a.y_out = y_out; a.resid = (half_t*) resid; a.ss_part = ss_part; a.resid_fx = (long long*) resid_fx; a.m = m; a.hidden = hidden;
    const int tasks = m * (hidden / 128);
    // every workgroup spin-waits on its peers' matching workgroups, so the whole grid has to be co-resident on every rank whatever the dispatch
    // order: at most ~2 workgroups per CU (decode messages are 32..512 tasks; prefill-sized messages belong to the collective library)
    EXL3_CHECK_ARG(tasks <= 8 * EXL3_AR_MAX_WGS, "ar_reduce: m * hidden too large for the one-shot push (use the collective library)");
    ar_push_reduce_kernel<<<(tasks + 7) / 8, 256, 0, (hipStream_t) stream>>>(a);
    return exl3_check_launch("ar_reduce");
}
