// ORACLE / TEST INFRASTRUCTURE -- not product code.
//
// A host-side stand-in for the small slice of the CUDA device API that the reference's hot-path *headers* use, so that
// those headers can be compiled for the CPU by g++ FROM WHERE THEY LIE under /root/reference (oracle/build_ref.sh) and
// executed as the parity pin of oracle/exl3_oracle.py:
//   exllamav3_ext/quant/codebook.cuh, quant/exl3_dq.cuh          (codebooks, trellis window readers)
//   exllamav3_ext/cache/lmq.cuh, cache/q_cache_kernels.cuh        (KV-cache quant / dequant, paged kernels)
// Nothing here is taken from the reference: every function implements the documented semantics of the CUDA intrinsic
// of the same name.  fp16 arithmetic is exact integer arithmetic with one round-to-nearest-even at the end (what the
// hardware instruction does), not a float emulation.  A 32-lane warp is emulated by 32 OS threads that meet at a
// barrier in every warp collective (__shfl_*_sync, __syncwarp); `__shared__` becomes a process-wide static, so only
// one thread block runs at a time.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <array>
#include <atomic>
#include <functional>
#include <thread>
#include <vector>
#include <pthread.h>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

// ---- fp16 ---------------------------------------------------------------------------------------------------------
struct half { uint16_t bits; };
typedef half __half;
struct half2 { half x, y; };

namespace shim {

// value of an fp16 in units of 2^-24 (every finite fp16 is an integer multiple of its smallest subnormal)
static inline int64_t h_units(half h)
{
    const int e = (h.bits >> 10) & 31, m = h.bits & 1023;
    int64_t v = e ? (int64_t) (1024 + m) << (e - 1) : m;
    return (h.bits & 0x8000) ? -v : v;
}
static inline bool h_special(half h) { return ((h.bits >> 10) & 31) == 31; }

// round an exact integer count of 2^-unit_log2 to fp16, round-to-nearest-even, overflow -> inf
static inline half round_units(__int128 v, int unit_log2)
{
    half r; r.bits = 0;
    if (v == 0) return r;
    const uint16_t sign = v < 0 ? 0x8000 : 0;
    unsigned __int128 a = v < 0 ? (unsigned __int128) (-v) : (unsigned __int128) v;
    int msb = 127; while (!((a >> msb) & 1)) --msb;
    // value = a * 2^-unit_log2; exponent of the leading bit
    int e = msb - unit_log2;                                  // floor(log2(value))
    int drop;                                                 // low bits of `a` below the fp16 ulp
    if (e < -14) drop = unit_log2 - 24;                       // subnormal: ulp 2^-24
    else drop = msb - 10;                                     // normal: 11 significant bits
    unsigned __int128 q;
    if (drop <= 0) q = a << (-drop);
    else
    {
        q = a >> drop;
        const unsigned __int128 rem = a & (((unsigned __int128) 1 << drop) - 1), halfway = (unsigned __int128) 1 << (drop - 1);
        if (rem > halfway || (rem == halfway && (q & 1))) ++q;
    }
    if (e < -14)
    {
        r.bits = sign | (uint16_t) q;                         // q == 1024 rolls into the smallest normal by construction
        return r;
    }
    if (q == 2048) { q = 1024; ++e; }
    if (e > 15) { r.bits = sign | 0x7c00; return r; }
    r.bits = sign | (uint16_t) ((e + 15) << 10) | (uint16_t) (q - 1024);
    return r;
}

static inline float h2f(half h)
{
    if (h_special(h))
    {
        uint32_t u = ((uint32_t) (h.bits & 0x8000) << 16) | 0x7f800000u | ((uint32_t) (h.bits & 1023) << 13);
        float f; memcpy(&f, &u, 4); return f;
    }
    return (float) ldexp((double) h_units(h), -24);           // exact: 11 significant bits
}

static inline half f2h_rn(float f)
{
    half r;
    uint32_t u; memcpy(&u, &f, 4);
    if (((u >> 23) & 255) == 255) { r.bits = (uint16_t) ((u >> 16) & 0x8000) | 0x7c00 | ((u & 0x7fffff) ? 0x200 : 0); return r; }
    if (f == 0.0f) { r.bits = (uint16_t) ((u >> 16) & 0x8000); return r; }
    if (fabsf(f) >= 131072.0f) { r.bits = (uint16_t) ((u >> 16) & 0x8000) | 0x7c00; return r; }
    if (fabsf(f) < ldexpf(1.0f, -40)) { r.bits = (uint16_t) ((u >> 16) & 0x8000); return r; }   // far below half an fp16 subnormal ulp
    // a float in this range is an integer multiple of 2^-64: exact as an __int128 count of that unit
    int ex; const double m = frexp((double) f, &ex);          // f = m * 2^ex, |m| in [0.5, 1)
    const int64_t mant = (int64_t) ldexp(m, 24);              // 24-bit integer mantissa, exact
    const int sh = ex - 24 + 64;                              // f = mant * 2^(ex-24) = (mant << sh) * 2^-64
    __int128 v = sh >= 0 ? ((__int128) mant << sh) : ((__int128) mant >> (-sh));   // sh >= 0 here (|f| >= 2^-40)
    return round_units(v, 64);
}

}  // namespace shim

static inline float __half2float(half h) { return shim::h2f(h); }
static inline half __float2half_rn(float f) { return shim::f2h_rn(f); }
static inline half __float2half(float f) { return shim::f2h_rn(f); }
static inline half __ushort_as_half(uint16_t u) { half h; h.bits = u; return h; }
static inline uint16_t __half_as_ushort(half h) { return h.bits; }
static inline half __low2half(half2 v) { return v.x; }
static inline half __high2half(half2 v) { return v.y; }
static inline half2 __halves2half2(half a, half b) { half2 r; r.x = a; r.y = b; return r; }
static inline half2 __half2half2(half a) { half2 r; r.x = a; r.y = a; return r; }
static inline half2 __lows2half2(half2 a, half2 b) { half2 r; r.x = a.x; r.y = b.x; return r; }
static inline half2 __highs2half2(half2 a, half2 b) { half2 r; r.x = a.y; r.y = b.y; return r; }
static inline half2 __floats2half2_rn(float a, float b) { half2 r; r.x = shim::f2h_rn(a); r.y = shim::f2h_rn(b); return r; }
// one IEEE fp16 operation each (finite operands: the codebooks never produce inf / nan)
static inline half __hadd(half a, half b) { return shim::round_units((__int128) shim::h_units(a) + shim::h_units(b), 24); }
static inline half __hmul(half a, half b) { return shim::round_units((__int128) shim::h_units(a) * shim::h_units(b), 48); }
static inline half __hfma(half a, half b, half c)
{
    return shim::round_units((__int128) shim::h_units(a) * shim::h_units(b) + ((__int128) shim::h_units(c) << 24), 48);
}
static inline half2 __hadd2(half2 a, half2 b) { half2 r; r.x = __hadd(a.x, b.x); r.y = __hadd(a.y, b.y); return r; }
static inline half2 __hmul2(half2 a, half2 b) { half2 r; r.x = __hmul(a.x, b.x); r.y = __hmul(a.y, b.y); return r; }
static inline half2 __hfma2(half2 a, half2 b, half2 c) { half2 r; r.x = __hfma(a.x, b.x, c.x); r.y = __hfma(a.y, b.y, c.y); return r; }

// ---- the reference's own helper unions / fragment types (util.cuh:74-90, ptx.cuh:5-16: plain data layouts) and the PTX-macro trio of
// ptx.cuh:304-315, restated from the PTX ISA semantics of bfe / shf.r.wrap -------------------------------------------
union half2_uint32
{
    uint32_t as_uint32; half2 as_half2;
    half2_uint32(uint32_t v) : as_uint32(v) {}
    half2_uint32(half2 v) : as_half2(v) {}
    half2_uint32() : as_uint32(0) {}
};
union half_uint16
{
    uint16_t as_uint16; half as_half;
    half_uint16(uint16_t v) : as_uint16(v) {}
    half_uint16(half v) : as_half(v) {}
    half_uint16() : as_uint16(0) {}
};
template <typename T, int n> struct Vec { T elems[n]; T& operator[](int i) { return elems[i]; } };
using FragB = Vec<half2, 2>;
static inline uint32_t bfe64(uint32_t lo, uint32_t hi, int offset, int length)
{
    const uint64_t v = ((uint64_t) hi << 32) | lo;
    return (uint32_t) ((v >> offset) & ((length >= 64) ? ~0ull : ((1ull << length) - 1)));
}
#define FSHF_IMM(dst, lo, hi, imm) do { (dst) = (uint32_t) (((((uint64_t) (hi)) << 32) | (uint64_t) (lo)) >> ((imm) & 31)); } while (0)
#define BFE16_IMM(dst, src, imm) do { (dst) = ((uint32_t) (src) >> (imm)) & 0xffffu; } while (0)

// `lop3.b32 d, a, b, c, lut` appears as inline PTX in codebook.cuh with a == d, b and c immediates and lut 0x6a = (a & b) ^ c.  The header is
// compiled untouched: this assembler macro gives the x86 assembler a definition of that mnemonic for exactly that form.
asm(".macro lop3.b32 d, a, b, c, lut\n"
    "  .if \\lut - 0x6a\n  .error \"lop3.b32: only lut 0x6a is defined by the host shim\"\n  .endif\n"
    "  andl $\\b, \\d\n"
    "  xorl $\\c, \\d\n"
    ".endm\n");

// ---- integer / float intrinsics -------------------------------------------------------------------------------------
static inline uint32_t __dp4a(uint32_t a, uint32_t b, uint32_t c)
{
    for (int i = 0; i < 4; ++i) c += ((a >> (8 * i)) & 255u) * ((b >> (8 * i)) & 255u);
    return c;
}
static inline uint32_t __funnelshift_r(uint32_t lo, uint32_t hi, uint32_t shift) { return (uint32_t) (((((uint64_t) hi) << 32) | lo) >> (shift & 31)); }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float2int_rd(float f)
{
    if (f != f) return 0;
    const double d = floor((double) f);
    if (d >= 2147483647.0) return 2147483647;
    if (d <= -2147483648.0) return -2147483647 - 1;
    return (int) d;
}
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }

// ---- execution model: one warp = 32 OS threads ----------------------------------------------------------------------
struct shim_dim3 { unsigned x = 1, y = 1, z = 1; };
static thread_local shim_dim3 threadIdx;
static shim_dim3 blockIdx, blockDim, gridDim;                  // set by the harness between (sequential) blocks

namespace shim {
static pthread_barrier_t g_bar;
static uint64_t g_slot[32];
static inline void bar() { pthread_barrier_wait(&g_bar); }
template <typename T> static inline T shfl(T v, int src)
{
    static_assert(sizeof(T) <= 8, "shuffle payload");
    const int lane = threadIdx.x & 31;
    uint64_t w = 0; memcpy(&w, &v, sizeof(T));
    g_slot[lane] = w;
    bar();
    T r; memcpy(&r, &g_slot[src & 31], sizeof(T));
    bar();
    return r;
}
// run f() once per lane of warp `warp` of the current block (threadIdx.x = 32 * warp + 0..31); blockIdx / blockDim / gridDim as set by the
// caller.  Kernels without cross-warp communication can be run one warp at a time this way under their real blockDim.
static inline void run_warp(const std::function<void()>& f, int warp = 0)
{
    pthread_barrier_init(&g_bar, nullptr, 32);
    std::vector<std::thread> th;
    for (int l = 0; l < 32; ++l) th.emplace_back([l, warp, &f] { threadIdx.x = (unsigned) (32 * warp + l); threadIdx.y = threadIdx.z = 0; f(); });
    for (auto& t : th) t.join();
    pthread_barrier_destroy(&g_bar);
}
}  // namespace shim

template <typename T> static inline T __shfl_xor_sync(unsigned, T v, int lane_mask) { return shim::shfl(v, (int) (threadIdx.x & 31) ^ lane_mask); }
template <typename T> static inline T __shfl_sync(unsigned, T v, int src_lane) { return shim::shfl(v, src_lane); }
static inline void __syncwarp(unsigned = 0xffffffffu) { shim::bar(); }
static inline uint32_t atomicOr(uint32_t* p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
