// EXL3 quantized small-m GEMM for gfx950, kernel generation 3 ("decode -> LDS transpose -> 16x16x32 MFMA"), 9..32 rows per pass.
//
//   C = ((A * suh) H) @ dequant(B) H * svh (+ bias)        reference: quant/exl3_gemm_kernel.cuh:8-80, quant/exl3_gemm_inner.cuh
//                                                          (semantics only; the reference streams 16 rows per pass)
//
// Why a third kernel: generation 2 (exl3_gemv2.kspec.hip) multiplies straight out of the decoding lane's registers with
// v_mfma_f32_4x4x4_16B_f16, which does a quarter of the matrix pipe's work per cycle.  That is free at m <= 4 (the decode VALU work
// dominates) but at 16 rows the kernel is MFMA-bound: 1024 MFMA cycles against ~800 decode cycles per 2 tile rows.  Here the decoding
// lane still reads one contiguous run of the bitstream (lane 8T + c owns columns c, c + 8 of tile T: exl3_lane_decode.cuh), but the fp16
// weights go through a wave-private LDS buffer into the B-operand layout of v_mfma_f32_16x16x32_f16: 8 MFMAs (128 cycles) per 16 rows
// per 2 tile rows.  The weight pass is decode-bound again up to 32 rows, so a 32-row pass costs about what a 4-row pass does.
//
// LDS transpose.  Unit = 2 tile rows = 32 k x 128 columns = 512 chunks of 8 halves (16 B); chunk (h, kq, T, c) = rows 8 kq .. 8 kq + 7
// of the unit, column 16 T + 8 h + c.  The decoding lane (T, c) writes its 8 chunks with ds_write_b128 at
//     (4 h + kq) * 1152 + (8 T + c) * 16          -- 1 KiB contiguous per instruction, conflict-free
// and MFMA lane (j, kg) reads, for column tile T, the chunk (h = j >> 3, kq = kg, T, c = j & 7): the 1152-byte row pitch puts the four
// 64-byte runs of every 16-lane ds_read_b128 group on disjoint banks.
// The k order inside the unit is natural (k = 8 kq + i), so the A operand is read straight from a row-major fp16 copy of the rotated
// activations (row pitch padded by 32 B: conflict-free ds_read_b128 across the 16 rows).
#include "exl3_common.cuh"
#include "exl3_api_internal.h"
#include "exl3_gemv_args.h"
#include "exl3_lane_decode.cuh"

#include <type_traits>

template <int I, int N, typename F>
__device__ __forceinline__ void g3_static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); g3_static_for<I + 1, N>(f); }
}

#define G3_STG_ROW 1152
#define G3_STG_BYTES (8 * G3_STG_ROW)
#define G3_XPAD 16
#ifndef G3_ROWS
#define G3_ROWS 4           // tile rows per unit = weight rows in flight per wave (2 per MFMA k-step)
#endif

template <int K, int CB, int MT>
__global__ __launch_bounds__(256)
void exl3_gemm3_kernel(const GemvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 8 * K;
    constexpr int MR = 16 * MT;

#ifdef G2_TIMING
    // diagnostics build: the stamps of exl3_gemv2.kspec.hip (tools/gemv_timeline.py reads both)
    uint64_t tstamp[6];
    tstamp[0] = __builtin_amdgcn_s_memrealtime();
    #define G3_T(i) tstamp[i] = __builtin_amdgcn_s_memrealtime()
#else
    #define G3_T(i)
#endif
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int m = a.m;
    const int s = blockIdx.x % a.S;
    const int cbg = blockIdx.x / a.S;
    int mi = 0;
    #pragma unroll
    for (int i = 1; i < GEMV_MAX_MATS; ++i) if (i < a.num_mats && cbg >= a.mat[i].cb_first) mi = i;
    const uint32_t* __restrict__ Bm = a.mat[mi].B;
    const half_t* __restrict__ suh = a.mat[mi].suh;
    const int n = a.mat[mi].n, cbl = cbg - a.mat[mi].cb_first, ws_off = a.mat[mi].ws_offset;
    const int tiles_n = n >> 4;
    const int k0s = s * a.kslice;
    const int k1s = min(k0s + a.kslice, a.k);
    const int nb = (k1s - k0s) >> 7;
    const int nwv = blockDim.x >> 6;
    // the same unit distribution as generation 2, in units of G3_ROWS tile rows; one chunk: contiguous ranges per wave, several: round robin
    const int units = nb * (8 / G3_ROWS);
    const int chb = a.chunk_blocks;
    const bool one_chunk = chb >= nb;
    const int ubase = one_chunk ? (units * wave) / nwv : wave;
    const int ustride = one_chunk ? 1 : nwv;
    const int nunits_w = one_chunk ? (units * (wave + 1)) / nwv - ubase : (wave < units ? (units - wave + nwv - 1) / nwv : 0);

    // LDS carve: [wave-private transpose buffers] [activations of one chunk, row-major fp16] ; the partial sums alias both after the loop
    const int ldx = chb * 128 + G3_XPAD;
    char* stg = smem + (size_t) wave * G3_STG_BYTES;
    half_t* xa = (half_t*) (smem + (size_t) nwv * G3_STG_BYTES);
    float* part = (float*) smem;

    const int l32 = lane & 31;
    const bool in_rotated = (a.flags & GEMV_IN_ROTATED) != 0;
    const half_t* __restrict__ x_src = in_rotated ? a.mat[mi].xh : a.A;
    const int hwid = tid >> 5, nhw = nwv * 2;

    const int T = lane >> 3, c = lane & 7;
    const uint32_t* __restrict__ strip = Bm + ((size_t) (k0s >> 4) * tiles_n + (size_t) cbl * 8) * NW + (size_t) lane * K;
    const size_t row_stride = (size_t) tiles_n * NW;
    const int prev_lane_addr = ((lane & ~7) | ((lane - 1) & 7)) << 2;
    const int last_unit = nunits_w > 0 ? ubase + (nunits_w - 1) * ustride : 0;

    float4_t acc[MT][8];
    #pragma unroll
    for (int i = 0; i < MT; ++i)
    {
        #pragma unroll
        for (int t = 0; t < 8; ++t) acc[i][t] = float4_t{ 0.f, 0.f, 0.f, 0.f };
    }

    struct PrepIn { half4_t xv, sv; };
    auto fetch = [&] (int c0, int cnt, int it) -> PrepIn
    {
        PrepIn r; r.sv = half4_t{ 0, 0, 0, 0 };
        const int t = min(it * nhw + hwid, cnt * m - 1);
        const int blk = c0 + t / m, row = t % m;
        const size_t kofs = (size_t) k0s + 128 * blk;
        r.xv = ((const half4_t*) (x_src + (size_t) row * a.k + kofs))[l32];
        if (!in_rotated) r.sv = ((const half4_t*) (suh + kofs))[l32];
        return r;
    };
    // activation operands first, weight rows second: loads return in issue order per wave (see exl3_gemv2.kspec.hip)
    // already rotated input (glue_rotate / glue_act): a straight copy in 16-byte pieces, four loads in flight per thread
    auto copy_load = [&] (int c0, int cnt, int base, uint4_t (&v)[4])
    {
        const int per_row = cnt * 16, total = m * per_row;
        const half_t* src0 = x_src + (size_t) k0s + 128 * c0;
        #pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const int idx = min(base + j * (int) blockDim.x + tid, total - 1);
            v[j] = *((const uint4_t*) (src0 + (size_t) (idx / per_row) * a.k + (idx % per_row) * 8));
        }
    };
    auto copy_store = [&] (int cnt, int base, const uint4_t (&v)[4])
    {
        const int per_row = cnt * 16, total = m * per_row;
        #pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const int idx = base + j * (int) blockDim.x + tid;
            if (idx < total) *((uint4_t*) (xa + (size_t) (idx / per_row) * ldx + (idx % per_row) * 8)) = v[j];
        }
    };
    PrepIn nx = { half4_t{ 0, 0, 0, 0 }, half4_t{ 0, 0, 0, 0 } };
    uint4_t cv[4];
    if (in_rotated) copy_load(0, min(chb, nb), 0, cv); else nx = fetch(0, min(chb, nb), 0);
    LaneWords<K> ring[G3_ROWS];
    if (nunits_w > 0)
    {
        #pragma unroll
        for (int u = 0; u < G3_ROWS; ++u) load_lane_words<K>(ring[u], strip + (size_t) (G3_ROWS * ubase + u) * row_stride);
    }

    G3_T(1);

    // MFMA operand geometry
    const int mj = lane & 15, kg = lane >> 4;
    const char* bsrc = stg + (4 * (mj >> 3) + kg) * G3_STG_ROW + (mj & 7) * 16;            // + T * 128 per column tile
    char* bdst = stg + lane * 16;                                                        // + (4 h + kq) * G3_STG_ROW per chunk
    const half_t* arow[MT];
    #pragma unroll
    for (int i = 0; i < MT; ++i) arow[i] = xa + (size_t) min(16 * i + mj, m - 1) * ldx + 8 * kg;
    int ui = 0;

    for (int c0 = 0; c0 < nb; c0 += chb)
    {
        const int cnt = min(chb, nb - c0);
        if (c0 > 0) __syncthreads();

        // ---- activations of blocks [c0, c0 + cnt) -> LDS, row-major
        if (in_rotated)
        {
            const int total = m * cnt * 16;
            for (int base = 0; base < total; base += 4 * (int) blockDim.x)
            {
                if (c0 > 0 || base > 0) copy_load(c0, cnt, base, cv);
                copy_store(cnt, base, cv);
            }
        }
        else
        {
            // cooperative input Hadamards: task t = (block t / m, row t % m), one per half-wave, software pipelined by one
            const int ntask = cnt * m;
            const int trips = (ntask + nhw - 1) / nhw;
            if (c0 > 0) nx = fetch(c0, cnt, 0);
            for (int it = 0; it < trips; ++it)
            {
                const PrepIn cur = nx;
                if (it + 1 < trips) nx = fetch(c0, cnt, it + 1);
                const int t = it * nhw + hwid;
                const bool act = t < ntask;
                const int tc = min(t, ntask - 1);
                const int blk_l = tc / m, row = tc % m;
                const half4_t xv = cur.xv * cur.sv;
                float h0 = (float) xv.x, h1 = (float) xv.y, h2 = (float) xv.z, h3 = (float) xv.w;
                had128_f32x4(h0, h1, h2, h3, l32);
                const half4_t o = { f2h(h0 * HAD_R_SCALE_128), f2h(h1 * HAD_R_SCALE_128), f2h(h2 * HAD_R_SCALE_128), f2h(h3 * HAD_R_SCALE_128) };
                if (act) *((half4_t*) (xa + (size_t) row * ldx + blk_l * 128 + 4 * l32)) = o;
            }
        }
        __syncthreads();
        if (c0 == 0) { G3_T(2); }

        const int row_end = (c0 + cnt) * 8;
        const int u_end = min(nunits_w, (G3_ROWS * ubase < row_end) ? ((row_end / G3_ROWS - 1 - ubase) / ustride + 1) : 0);

        for (; ui < u_end; ++ui)
        {
            const int unit = ubase + ui * ustride;
            const int nxt = min(unit + ustride, last_unit);
            #pragma unroll
            for (int hh = 0; hh < G3_ROWS / 2; ++hh)
            {
            #pragma unroll
            for (int u = 0; u < 2; ++u)
            {
                uint32_t Wx[K + 1];
                #pragma unroll
                for (int i = 0; i < K; ++i) Wx[i + 1] = ring[2 * hh + u].w[i];
                Wx[0] = (uint32_t) __builtin_amdgcn_ds_bpermute(prev_lane_addr, (int) ring[2 * hh + u].w[K - 1]);
                load_lane_words<K>(ring[2 * hh + u], strip + (size_t) (G3_ROWS * nxt + 2 * hh + u) * row_stride);

                // exact fp16 weights: quad q of column c / c + 8 = rows {2q, 2q+1 | 2q+8, 2q+9}: low words -> rows 0..7, high words -> rows 8..15
                uint32_t clo[4], chi[4], dlo[4], dhi[4];
#ifdef G3_ABL_NODECODE
                #pragma unroll
                for (int q = 0; q < 4; ++q) { clo[q] = Wx[q + 1]; chi[q] = Wx[q]; dlo[q] = Wx[q + 1] * 3u; dhi[q] = Wx[q] * 5u; }
#else
                g3_static_for<0, 4>([&] (auto qc)
                {
                    constexpr int q = decltype(qc)::value;
                    half4_t bc[2], bd[2];
                    decode_quad<K, CB, 0, 8 * q>(Wx, bc);
                    decode_quad<K, CB, 0, 8 * q + 4>(Wx, bd);
                    union { half4_t h; uint32_t w[2]; } uc, ud; uc.h = bc[0]; ud.h = bd[0];
                    clo[q] = uc.w[0]; chi[q] = uc.w[1]; dlo[q] = ud.w[0]; dhi[q] = ud.w[1];
                });
#endif
#ifdef G3_ABL_NOLDS
                asm volatile("" :: "v"(clo[0] ^ clo[1] ^ clo[2] ^ clo[3] ^ chi[0] ^ chi[1] ^ chi[2] ^ chi[3] ^ dlo[0] ^ dlo[1] ^ dlo[2] ^ dlo[3] ^ dhi[0] ^ dhi[1] ^ dhi[2] ^ dhi[3]));
#else
                *((uint4_t*) (bdst + (0 + 2 * u + 0) * G3_STG_ROW)) = uint4_t{ clo[0], clo[1], clo[2], clo[3] };
                *((uint4_t*) (bdst + (0 + 2 * u + 1) * G3_STG_ROW)) = uint4_t{ chi[0], chi[1], chi[2], chi[3] };
                *((uint4_t*) (bdst + (4 + 2 * u + 0) * G3_STG_ROW)) = uint4_t{ dlo[0], dlo[1], dlo[2], dlo[3] };
                *((uint4_t*) (bdst + (4 + 2 * u + 1) * G3_STG_ROW)) = uint4_t{ dhi[0], dhi[1], dhi[2], dhi[3] };
#endif
            }
            // the wave's own LDS operations complete in order: no wait between its stores and the loads below, only compiler ordering
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();

            half8_t af[MT];
            const int kloc = 16 * G3_ROWS * unit + 32 * hh - 128 * c0;
            #pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = *((const half8_t*) (arow[i] + kloc));
            #pragma unroll
            for (int t = 0; t < 8; ++t)
            {
#ifdef G3_ABL_NOLDS
                const half8_t bf = af[0];
#else
                const half8_t bf = *((const half8_t*) (bsrc + t * 128));
#endif
#ifdef G3_ABL_NOMFMA
                #pragma unroll
                for (int i = 0; i < MT; ++i) { acc[i][t][0] += (float) bf[0] * (float) af[i][0]; }
#else
                #pragma unroll
                for (int i = 0; i < MT; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf, acc[i][t], 0, 0, 0);
#endif
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            }
        }
    }

    // ---- epilogue: per-wave partials -> LDS (aliases the transpose buffers and the activations), cross-wave sum, output Hadamard
    G3_T(3);
    __syncthreads();
    {
        float* pw = part + (size_t) wave * MR * 128;
        #pragma unroll
        for (int i = 0; i < MT; ++i)
        {
            #pragma unroll
            for (int t = 0; t < 8; ++t)
            {
                #pragma unroll
                for (int r = 0; r < 4; ++r)
                {
                    const int row = 16 * i + 4 * kg + r;
                    if (row < m) pw[row * 128 + 16 * t + mj] = acc[i][t][r];
                }
            }
        }
    }
    __syncthreads();
    G3_T(4);

    const int l = tid & 31, hw8 = tid >> 5;
    const size_t wstride = (size_t) MR * 128;
    if (a.S > 1 || (a.flags & GEMV_OUT_DEFERRED))
    {
        float* slab = a.workspace + ws_off + ((size_t) cbl * a.S + s) * (size_t) m * 128;
        for (int row = hw8; row < m; row += nwv * 2)
        {
            const float* p0 = part + row * 128;
            float4_t v = ((const float4_t*) p0)[l];
            for (int w = 1; w < nwv; ++w)
            {
                float4_t t = ((const float4_t*) (p0 + w * wstride))[l];
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            ((float4_t*) (slab + row * 128))[l] = v;
        }
#ifdef G2_TIMING
        if (tid == 0)
        {
            tstamp[5] = __builtin_amdgcn_s_memrealtime();
            uint64_t* dbg = (uint64_t*) a.ws_debug + (size_t) blockIdx.x * 8;
            for (int i = 0; i < 6; ++i) dbg[i] = tstamp[i];
            uint32_t xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            uint32_t hwreg; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwreg));
            dbg[6] = xcc; dbg[7] = hwreg;
        }
#endif
        return;
    }

    const half_t* svh = a.mat[mi].svh + cbl * 128;
    const half_t* bias = a.mat[mi].bias ? a.mat[mi].bias + cbl * 128 : nullptr;
    void* C_m = a.mat[mi].C;
    for (int base = 0; base < m; base += nwv * 2)
    {
        int row = base + hw8;
        bool act = row < m;
        const float* p0 = part + (act ? row : 0) * 128;
        float4_t v = ((const float4_t*) p0)[l];
        for (int w = 1; w < nwv; ++w)
        {
            float4_t t = ((const float4_t*) (p0 + w * wstride))[l];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        float h0 = v.x, h1 = v.y, h2 = v.z, h3 = v.w;
        had128_f32x4(h0, h1, h2, h3, l);
        h0 *= HAD_R_SCALE_128; h1 *= HAD_R_SCALE_128; h2 *= HAD_R_SCALE_128; h3 *= HAD_R_SCALE_128;
        if (!act) continue;
        half4_t sc = ((const half4_t*) svh)[l];
        size_t off = ((size_t) a.c_row_offset + row) * n + cbl * 128 + 4 * l;
        if (a.c_fp32)
        {
            float4_t o = { h0 * (float) sc.x, h1 * (float) sc.y, h2 * (float) sc.z, h3 * (float) sc.w };
            if (bias) { half4_t bv = ((const half4_t*) bias)[l]; o.x += (float) bv.x; o.y += (float) bv.y; o.z += (float) bv.z; o.w += (float) bv.w; }
            *((float4_t*) ((float*) C_m + off)) = o;
        }
        else
        {
            half4_t o = { f2h(h0), f2h(h1), f2h(h2), f2h(h3) };
            o = o * sc;
            if (bias) o = o + ((const half4_t*) bias)[l];
            *((half4_t*) ((half_t*) C_m + off)) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
#ifndef G2_K
#error "compile with -DG2_K=<bits per weight>"
#endif

template <int CB>
static void g3_launch_cb(int mt, int nwv, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args)
{
    if (mt == 1) exl3_gemm3_kernel<G2_K, CB, 1><<<grid, dim3(64 * nwv), lds, st>>>(args);
    else         exl3_gemm3_kernel<G2_K, CB, 2><<<grid, dim3(64 * nwv), lds, st>>>(args);
}

#define G3_CAT_(a, b) a##b
#define G3_CAT(a, b) G3_CAT_(a, b)

void G3_CAT(exl3_gemm3_launch_k, G2_K)(int cb, int mt, int nwv, dim3 grid, size_t lds, hipStream_t st, const GemvArgs& args)
{
    if (cb == 0) g3_launch_cb<0>(mt, nwv, grid, lds, st, args);
    else if (cb == 1) g3_launch_cb<1>(mt, nwv, grid, lds, st, args);
    else g3_launch_cb<2>(mt, nwv, grid, lds, st, args);
}

#if G2_K == 4
// LDS bytes of a launch: max(transpose buffers + activations of one chunk, partial sums)
size_t exl3_gemm3_lds_bytes(int mt, int nwv, int m, int chunk_blocks)
{
    const size_t stream = (size_t) nwv * G3_STG_BYTES + (size_t) m * (chunk_blocks * 128 + G3_XPAD) * 2;
    const size_t part = (size_t) nwv * 16 * mt * 128 * 4;
    return stream > part ? stream : part;
}
#endif
