// How many waves per SIMD does the EXL3 decode loop need to saturate the VALU on MI355X, and does more per-wave ILP / a deeper weight ring
// buy any of it back?  (Round 3, written without GPU minutes left; decides whether "fat workgroup" designs -- a fused gate|up -> silu*mul -> down
// launch with 16-wave workgroups, or a persistent step kernel, both of which run at <= 4 waves per SIMD -- can stream at the rate the shipped
// 7..8-waves-per-SIMD launches do.  profiles/NOTES.md B section 6.)
//
// The loop body is generation 4's work unit (exl3_gemv4.kspec.hip: 2 tile rows = 64 weights per lane, K = 4, mul1 codebook, FAST variant:
// compile-time bit windows, v_mul_lo_u32, v_sad_u8, 16 x v_mfma_f32_4x4x4_16B_f16) on a private stream of weight rows:
//   SCHED = 1: a sched_barrier after every decoded quad (the shipped form: 8 weights in flight, "occupancy over ILP")
//   SCHED = 0: the compiler interleaves the unit's 8 quads freely
//   NR    = ring slots: 2 = one unit of lookahead (shipped), 4 = two units, 6 = three
//   hot   = every wave re-reads the same 64 KiB (L2-resident: VALU issue only); cold = a 2 GiB stream (HBM)
// Grid = CUs x W workgroups of 4 waves (one wave per SIMD each), all resident: W waves per SIMD.  Prints ns per unit per SIMD; the shipped kernels'
// streaming phase runs at about 360 ns per unit per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -o /tmp/ubench_decode_occupancy tools/ubench_decode_occupancy.hip && /tmp/ubench_decode_occupancy
#include "../exllamav3_amd/csrc/exl3_common.cuh"
#include "../exllamav3_amd/csrc/exl3_lane_decode.cuh"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

constexpr int K = 4, CB = EXL3_CB_MUL1, VAR = 1;

// one unit out of ring slots (r0, r0 + 1); the slots are refilled with the rows at `refill`
template <int SCHED, int HALF>
__device__ __forceinline__ void unit(LaneWords<K>& s0, LaneWords<K>& s1, const uint32_t* __restrict__ refill, int lane, half4_t ag, float4_t& acc_c, float4_t& acc_d)
{
    static_for<0, 2>([&] (auto uc)
    {
        constexpr int u = decltype(uc)::value;
        LaneWords<K>& slot = u ? s1 : s0;
        uint32_t Wx[K + 1];
        #pragma unroll
        for (int i = 0; i < K; ++i) Wx[i + 1] = slot.w[i];
        {
            const uint32_t wl = slot.w[K - 1];
            const uint32_t r1 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x121, 0xf, 0xf, true);
            const uint32_t r9 = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) wl, 0x129, 0xf, 0xf, true);
            Wx[0] = (lane & 7) ? r1 : r9;
        }
        load_lane_words<K>(slot, refill + (size_t) u * 256);            // 64 lanes x 4 words = 256 words per row
        static_for<0, 4>([&] (auto qc)
        {
            constexpr int q = decltype(qc)::value;
            constexpr int ABID = 8 * HALF + 4 * u + q;
            half4_t bc[2], bd[2];
            decode_quad<K, CB, VAR, 8 * q>(Wx, bc);
            decode_quad<K, CB, VAR, 8 * q + 4>(Wx, bd);
            acc_c = __builtin_amdgcn_mfma_f32_4x4x4f16(ag, bc[0], acc_c, 4, ABID, 0);
            acc_d = __builtin_amdgcn_mfma_f32_4x4x4f16(ag, bd[0], acc_d, 4, ABID, 0);
            if constexpr (SCHED) __builtin_amdgcn_sched_barrier(0);
        });
    });
}

template <int SCHED, int NR>
__global__ __launch_bounds__(256) void k_stream(const uint32_t* __restrict__ w, size_t wave_stride_words, size_t mask_words, int nunits, float* out)
{
    const int lane = threadIdx.x & 63;
    const size_t gw = (size_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    // row r of this wave: 256 contiguous words (1 KiB per wave row, as a tile row of a 128-column block)
    const size_t base = gw * wave_stride_words;
    auto rowp = [&] (int r) { return w + (((base + (size_t) r * 256) & mask_words) + (size_t) lane * K); };
    LaneWords<K> ring[NR];
    #pragma unroll
    for (int i = 0; i < NR; ++i) load_lane_words<K>(ring[i], rowp(i));
    half4_t ag = { (half_t) (0.01f * (lane & 3)), (half_t) 0.02f, (half_t) -0.01f, (half_t) 0.03f };
    float4_t acc_c = { 0.f, 0.f, 0.f, 0.f }, acc_d = acc_c;
    constexpr int PF = NR / 2;                                           // units of lookahead
    const int rows = 2 * nunits;
    // PF units per trip so that every ring index is a compile-time constant
    for (int u0 = 0; u0 < nunits; u0 += PF)
    {
        static_for<0, PF>([&] (auto pc)
        {
            constexpr int p = decltype(pc)::value;
            const int r_next = min(2 * (u0 + p + PF), rows - 2);
            if (p & 1) unit<SCHED, 1>(ring[2 * p], ring[2 * p + 1], rowp(r_next), lane, ag, acc_c, acc_d);
            else unit<SCHED, 0>(ring[2 * p], ring[2 * p + 1], rowp(r_next), lane, ag, acc_c, acc_d);
        });
    }
    if (acc_c[0] + acc_d[1] == 123.456f) out[gw] = acc_c[0];            // keep the work
}

struct Cfg { const char* name; int sched, nr; };

template <int SCHED, int NR>
static float run(const uint32_t* w, size_t stride, size_t mask, int wgs, int nunits, float* out)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep)
    {
        // cold runs: every repetition starts 2 GiB further into the 8 GiB buffer (nothing of it is left in the 256 MiB Infinity Cache)
        const uint32_t* wr = w + (stride ? ((size_t) rep << 29) : 0);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_stream<SCHED, NR>), dim3(wgs), dim3(256), 0, 0, wr, stride, mask, nunits, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const size_t cold_bytes = (size_t) 2 << 30, hot_bytes = (size_t) 1 << 20;      // power-of-two sizes: wrap with a mask
    uint32_t* w; CK(hipMalloc(&w, 4 * cold_bytes)); CK(hipMemset(w, 0x5a, 4 * cold_bytes));
    float* out; CK(hipMalloc(&out, (size_t) cus * 8 * 4 * 4));
    const int nunits = 96;                                              // per wave: 96 units = 192 KiB of weight rows
    int regs[6];
    {
        hipFuncAttributes a;
        CK(hipFuncGetAttributes(&a, (const void*) k_stream<1, 2>)); regs[0] = a.numRegs;
        CK(hipFuncGetAttributes(&a, (const void*) k_stream<0, 2>)); regs[1] = a.numRegs;
        CK(hipFuncGetAttributes(&a, (const void*) k_stream<1, 4>)); regs[2] = a.numRegs;
        CK(hipFuncGetAttributes(&a, (const void*) k_stream<0, 4>)); regs[3] = a.numRegs;
        CK(hipFuncGetAttributes(&a, (const void*) k_stream<1, 6>)); regs[4] = a.numRegs;
        CK(hipFuncGetAttributes(&a, (const void*) k_stream<0, 6>)); regs[5] = a.numRegs;
    }
    // (a variant above 64 VGPRs cannot hold 8 waves per SIMD: its W = 8 row then runs in two rounds -- read it with the register counts)
    printf("{\"device\": \"%s\", \"cus\": %d, \"units_per_wave\": %d, \"vgprs\": {\"sched_ring2\": %d, \"free_ring2\": %d, \"sched_ring4\": %d, \"free_ring4\": %d, \"sched_ring6\": %d, \"free_ring6\": %d}, \"results\": [\n",
           p.gcnArchName, cus, nunits, regs[0], regs[1], regs[2], regs[3], regs[4], regs[5]);
    bool first = true;
    for (int hot = 1; hot >= 0; --hot)
        for (int W = 1; W <= 8; ++W)
        {
            const int wgs = cus * W;
            const size_t stride = hot ? 0 : (size_t) nunits * 2 * 256;    // hot: every wave walks the same rows; cold: private, contiguous per wave
            const size_t mask = (hot ? hot_bytes : cold_bytes) / 4 - 1;
            float t[6];
            t[0] = run<1, 2>(w, stride, mask, wgs, nunits, out);
            t[1] = run<0, 2>(w, stride, mask, wgs, nunits, out);
            t[2] = run<1, 4>(w, stride, mask, wgs, nunits, out);
            t[3] = run<0, 4>(w, stride, mask, wgs, nunits, out);
            t[4] = run<1, 6>(w, stride, mask, wgs, nunits, out);
            t[5] = run<0, 6>(w, stride, mask, wgs, nunits, out);
            // ns per unit per SIMD: W waves on each SIMD each run nunits units
            printf("%s  {\"data\": \"%s\", \"waves_per_simd\": %d, \"ns_per_unit_per_simd\": {\"sched_ring2\": %.1f, \"free_ring2\": %.1f, \"sched_ring4\": %.1f, \"free_ring4\": %.1f, \"sched_ring6\": %.1f, \"free_ring6\": %.1f}, \"tb_per_s_equiv\": %.2f}",
                   first ? "" : ",\n", hot ? "hot" : "cold", W,
                   t[0] * 1e6 / (nunits * W), t[1] * 1e6 / (nunits * W), t[2] * 1e6 / (nunits * W), t[3] * 1e6 / (nunits * W), t[4] * 1e6 / (nunits * W), t[5] * 1e6 / (nunits * W),
                   (double) wgs * 4 * nunits * 2048.0 / (t[0] * 1e-3) / 1e12);
            first = false; fflush(stdout);
        }
    printf("\n]}\n");
    return 0;
}
