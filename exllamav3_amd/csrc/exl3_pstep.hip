// Persistent decode step (generation 5): planner + C ABI.  Kernel: exl3_pstep_kernel.cuh; design and structures: exl3_pstep.cuh.
//
// The planner cuts every fused GEMV ("op") of the step into one rectangle per CU: (column groups of each matrix) x (k-slices), chosen so that
// the largest rectangle is as small as possible, a rectangle is at most 16 column blocks wide (a wave's run of work units then crosses at most one
// column-block boundary) and the product of groups x slices fits the CUs.  The tables live in device memory; the kernel reads its rectangle of
// op i + 1 while it streams op i.
#include "exl3_pstep_launch.h"
#include "exl3_api_internal.h"
#include <string>
#include <vector>
#include <stdio.h>
#include <string.h>

#ifdef PS_TEST_NO_ROW_PARITY
#define PS_ROW_PARITY(li) 0
#else
#define PS_ROW_PARITY(li) ((li) & 1)
#endif

namespace
{
struct PsHandle
{
    int K, K2, KH, cb, nops, ncu, pmax, spin_limit, n_layers;
    const PsKernelSet* kset; int* d_runs;
    int tp_world, tp_rank; char* d_xbuf; size_t xbuf_bytes; unsigned long long* d_peers; void* peer_ptr[8];      // tensor parallelism inside the step: this rank's exchange buffer, the ranks' mapped buffers
    PsOp* d_ops; PsTile* d_tiles; uint32_t* d_cnt; uint32_t* d_err; unsigned long long* d_dbg;
    unsigned long long* d_slab_a; unsigned long long* d_slab_b; unsigned long long* d_slab_c; unsigned long long* d_slab_d; unsigned long long* d_rbuf; uint32_t* d_epoch;
    unsigned long long* d_att_rec; unsigned long long* d_att_stats; int attn;
    uint32_t* d_repack; size_t repack_words;               // the ops' weights in plan order (PsOp::Bp); null: the checkpoint layout is streamed as it is
    uint32_t* h_err;                                       // pinned host mirror of the error word: copied behind every run (capturable), read without a synchronisation
    int att_nsplit, att_cap, head_dim;
    std::vector<PsOp> ops; std::vector<PsTile> tiles;      // host copies of the plan (exl3_pstep_unpack_op, exl3_pstep_attn_geometry)
    size_t cnt_bytes, dbg_words;
    std::string desc;
};

struct OpPlan { int S; int g[PS_MAX_MATS]; int G; int wmax, hmax; };

// groups x slices for one op: minimise the largest rectangle (in work units = 2 tile rows x 128 columns); ties -> fewer slices (fewer adds per address / slab lines)
bool plan_op(int ncu, int nblk, const int* ncb, int nmat, int in_type, int out_type, OpPlan& best, bool direct = false, int max_S = 32)
{
    double best_cost = 1e30; bool found = false;
    int total_cb = 0; for (int i = 0; i < nmat; ++i) total_cb += ncb[i];
    for (int S = 1; S <= nblk && S <= max_S; ++S)
    {
        if (out_type == PS_OUT_FINAL && S != 1) continue;
        const int nbmax = (nblk + S - 1) / S;
        if (nbmax > (out_type == PS_OUT_FINAL ? PS_MAX_ROW_BLOCKS : PS_MAX_SLICE_BLOCKS)) continue;      // (quads in LDS; the lm_head's one slice may be the whole row)
        if (in_type != PS_IN_NORM && nbmax > 8) continue;                  // one preparation task per service half-wave
        if (direct && nbmax > 4) continue;                                 // a DIRECT RMSNorm op gathers the partial lines of at most four blocks (LDS: gath)
        const int G = ncu / S;
        if (G < nmat) continue;
        OpPlan p; p.S = S;
        int sum = 0;
        for (int i = 0; i < nmat; ++i) { int g = (int) ((long long) G * ncb[i] / total_cb); if (g < 1) g = 1; if (g > ncb[i]) g = ncb[i]; p.g[i] = g; sum += g; }
        while (sum > G) { int bi = -1; for (int i = 0; i < nmat; ++i) if (p.g[i] > 1 && (bi < 0 || p.g[i] > p.g[bi])) bi = i; if (bi < 0) break; --p.g[bi]; --sum; }
        if (sum > G) continue;
        for (;;)
        {
            // hand a spare group to the matrix with the widest rectangles
            if (sum >= G) break;
            int bi = -1, bw = 0;
            for (int i = 0; i < nmat; ++i) { const int w = (ncb[i] + p.g[i] - 1) / p.g[i]; if (p.g[i] < ncb[i] && w > bw) { bw = w; bi = i; } }
            if (bi < 0) break;
            ++p.g[bi]; ++sum;
        }
        p.G = sum; p.wmax = 0;
        for (int i = 0; i < nmat; ++i) { const int w = (ncb[i] + p.g[i] - 1) / p.g[i]; if (w > p.wmax) p.wmax = w; }
        if (p.wmax > 12) continue;                                          // a streaming wave's run crosses at most one column-block boundary
        if (out_type == PS_OUT_ATOMIC && p.wmax > 4) continue;             // an owner workgroup gathers at most four blocks of the residual row
        p.hmax = 4 * nbmax;
        // cost in work units (~ 0.1 us of a CU's streaming each): the rectangle's units beyond what the waves decode ahead while the input is on its way (those are free), plus
        // what the CONSUMER pays to gather this op's S partial lines per block -- one more round trip per eight lines (EXL3_HIP_PSTEP_COST=0: the round-5 cost, the
        // largest rectangle alone)
        static const int cost_model = getenv("EXL3_HIP_PSTEP_COST") ? atoi(getenv("EXL3_HIP_PSTEP_COST")) : 1;
        const int units = p.wmax * p.hmax;
        const double cost = cost_model == 0 ? (double) units + 0.2 * S
                                            : (double) (units > 30 ? units - 30 : 0) + 6.0 * ((S + 7) / 8) + 0.05 * S + 0.05 * units;
        if (cost < best_cost) { best_cost = cost; best = p; found = true; }
    }
    return found;
}

// How the twelve streaming waves share a rectangle of T = 4 nb x ncb work units (H = 4 nb per column).  A SIMD hosts waves w, w + 4, w + 8 and arbitrates oldest-first: the
// age groups A (0-3) / B (4-7) / C (8-11) stream at ~ 1.05 / 0.89 / 0.44 units per us while all three run (per-unit stamps of an 8B gate|up op: profiles/r06_*), so equal runs
// leave the youngest wave streaming alone, latency-bound, for the last third of the phase.  Every wave decodes `pre` units ahead while the op's input is on its way (those cost
// no time behind the input); the REST of the SIMD's share is cut in the ratio of the groups' rates (permille fA : fB : 1000 - fA - fB).  out[g] = n_g << 2 | e_g: the four waves
// of group g take n_g units, the first e_g of them one more.  All zero = the uniform partition [T w / 12, T (w + 1) / 12): rectangles whose units are all decoded ahead, runs
// that would be longer than a column (a wave keeps two partial rows: its run may cross ONE column boundary), and the lm_head (its finish still uses the closed formulas).
void wave_partition(int T, int H, int pre, int fA, int fB, int* out)
{
    out[0] = out[1] = out[2] = 0;
    if (fA <= 0 || fB <= 0 || fA + fB >= 1000 || T <= 12 * pre) return;
    const double rest = (double) T / 4.0 - 3.0 * pre;                        // units per SIMD streamed behind the input
    int G[3];
    G[0] = (int) (4.0 * (pre + rest * fA / 1000.0) + 0.5);
    G[1] = (int) (4.0 * (pre + rest * fB / 1000.0) + 0.5);
    G[2] = T - G[0] - G[1];
    if (G[2] < 4 * pre || G[0] < G[1] || G[1] < G[2]) return;
    for (int g = 0; g < 3; ++g) if ((G[g] + 3) / 4 > H) return;             // (a run longer than a column would cross two boundaries)
    // the finish reads the partial rows of a column from at most EIGHT consecutive waves (exl3_pstep_kernel.cuh: col_table): check every column of this cut
    {
        int u0[12], u1[12], b = 0;
        for (int g = 0; g < 3; ++g) { const int n = G[g] / 4, e = G[g] % 4; for (int i = 0; i < 4; ++i) { u0[4 * g + i] = b + i * n + (i < e ? i : e); u1[4 * g + i] = u0[4 * g + i] + n + (i < e ? 1 : 0); } b += G[g]; }
        for (int j = 0; j * H < T; ++j)
        {
            int first = -1, last = -1;
            for (int w = 0; w < 12; ++w) if (u1[w] > u0[w] && u0[w] < (j + 1) * H && u1[w] > j * H) { if (first < 0) first = w; last = w; }
            if (first >= 0 && last - first >= 8) return;
        }
    }
    for (int g = 0; g < 3; ++g) out[g] = ((G[g] / 4) << 2) | (G[g] % 4);
}

// the shares of the age groups (permille of the units streamed behind the input): EXL3_HIP_PSTEP_SHARES="fA,fB" overrides ("0,0": uniform partition everywhere)
void wave_shares(int& fA, int& fB)
{
    fA = 460; fB = 370;                    // (same-box sweep on MI355X, Llama-3.1-8B: profiles/r06_pstep_wave_shares_ab.txt)
    if (const char* e = getenv("EXL3_HIP_PSTEP_SHARES")) { int a = 0, b = 0; if (sscanf(e, "%d,%d", &a, &b) == 2) { fA = a; fB = b; } }
}

// the rectangles of one op: workgroup (column group gi, slice s) = gi * S + s; every (column block, Hadamard block) of the op lies in exactly one
// rectangle (tests/test_pstep_plan.py checks the partition on the CPU through exl3_pstep_plan_tiles)
void fill_tiles(PsTile* T, int ncu, const OpPlan& p, const int* ncb, int nmat, int nblk, int side_tasks, bool weighted = true, int pre = 3)
{
    for (int c = 0; c < ncu; ++c) { T[c].mat = -1; T[c].cb0 = 0; T[c].ncb = 0; T[c].b0 = 0; T[c].nb = 0; T[c].slice = 0; T[c].side = -1; T[c].flags = 0; T[c].ubase = 0; T[c].r0_ = T[c].r1_ = T[c].r2_ = 0; }
    int gi = 0;
    for (int i = 0; i < nmat; ++i)
        for (int j = 0; j < p.g[i]; ++j, ++gi)
        {
            const int c0 = (int) ((long long) j * ncb[i] / p.g[i]), c1 = (int) ((long long) (j + 1) * ncb[i] / p.g[i]);
            for (int s = 0; s < p.S; ++s)
            {
                PsTile& t = T[gi * p.S + s];
                t.mat = i; t.cb0 = c0; t.ncb = c1 - c0;
                t.b0 = (int) ((long long) s * nblk / p.S); t.nb = (int) ((long long) (s + 1) * nblk / p.S) - t.b0;
                t.slice = s;
                if (i == 0 && j == 0) t.flags |= PS_TILE_Q_OUT;
            }
        }
    for (int t = 0; t < side_tasks; ++t) T[(int) ((long long) t * ncu / side_tasks)].side = t;
    // first work unit (2 tile rows x 128 columns) of every rectangle in the op's repacked weights: rectangles in workgroup order, 4 nb x ncb units each
    int ub = 0;
    for (int c = 0; c < ncu; ++c) { T[c].ubase = ub; if (T[c].mat >= 0) ub += 4 * T[c].nb * T[c].ncb; }
    // the streaming waves' shares of every rectangle (wave_partition)
    int fA, fB; wave_shares(fA, fB);
    for (int c = 0; c < ncu; ++c) if (weighted && T[c].mat >= 0)
    {
        int o3[3]; wave_partition(4 * T[c].nb * T[c].ncb, 4 * T[c].nb, pre, fA, fB, o3);
        T[c].r0_ = o3[0]; T[c].r1_ = o3[1]; T[c].r2_ = o3[2];
    }
}

void lin_to_mat(const exl3_pstep_linear_t& l, PsMat& m)
{
    m.B = (const uint32_t*) l.trellis; m.suh = (const half_t*) l.suh; m.svh = (const half_t*) l.svh; m.slab = nullptr; m.n = l.n; m.tiles_n = l.n / 16;
}
}

// the kernel unit of the layers' bits per weight (exl3_pstep.kspec.hip, one translation unit per K)
static const PsKernelSet* ps_kset(int K)
{
    switch (K)
    {
        case 1: return ps_kernel_set_k1(); case 2: return ps_kernel_set_k2(); case 3: return ps_kernel_set_k3(); case 4: return ps_kernel_set_k4();
        case 5: return ps_kernel_set_k5(); case 6: return ps_kernel_set_k6(); case 7: return ps_kernel_set_k7(); case 8: return ps_kernel_set_k8();
    }
    return nullptr;
}

// Load-time repack ("re-ordering tiles at load time is a legal one-time transform"): the op's packed words copied into the order the plan streams them --
// [workgroup rectangle][work unit q = column-major (column block j, unit i)][2 tile rows][8 tiles x 8 K words] -- so that streaming wave w of the rectangle reads the
// contiguous range of units [T w / 12, T (w + 1) / 12) instead of 1 KiB row pieces at a stride of n / 16 x 16 K bytes.  Same words, other addresses: decode is untouched.
// inverse != 0: the same walk copies repacked -> checkpoint layout into `plain` for matrix `only_mat` (exl3_pstep_unpack_op: the bit-exactness test of the permutation).
__global__ __launch_bounds__(128) void ps_repack_kernel(const PsOp* __restrict__ op, const PsTile* __restrict__ tiles, uint32_t* __restrict__ packed, int K,
                                                        int inverse, int only_mat, uint32_t* __restrict__ plain)
{
    const PsTile t = tiles[blockIdx.y];
    if (t.mat < 0 || (inverse && t.mat != only_mat)) return;
    const int H = 4 * t.nb, T = H * t.ncb, NW = 8 * K, row_words = 8 * NW;
    const uint32_t* B = op->mat[t.mat].B; const int tn = op->mat[t.mat].tiles_n;
    for (int q = blockIdx.x; q < T; q += gridDim.x)
    {
        const int j = q / H, i = q - j * H;
        for (int r = 0; r < 2; ++r)
        {
            const size_t src = ((size_t) (t.b0 * 8 + 2 * i + r) * tn + (size_t) (t.cb0 + j) * 8) * NW;
            uint32_t* const pk = packed + (size_t) (t.ubase + q) * (2 * row_words) + (size_t) r * row_words;
            if (!inverse) for (int w = threadIdx.x; w < row_words; w += blockDim.x) pk[w] = B[src + w];
            else          for (int w = threadIdx.x; w < row_words; w += blockDim.x) plain[src + w] = pk[w];
        }
    }
}

extern "C" int exl3_pstep_create(void** handle_out, const exl3_pstep_layer_t* layers, int n_layers, const exl3_pstep_linear_t* head, const void* final_norm,
                                 int hidden, int heads_q, int heads_kv, int head_dim, int K, int cb, float eps, int rope_mode, int flags)
{
    EXL3_CHECK_ARG(handle_out && layers && n_layers >= 1 && head && final_norm, "exl3_pstep_create: null argument");
    EXL3_CHECK_ARG(cb >= 0 && cb <= 2, "exl3_pstep_create: codebook 0 (3INST) | 1 (mcg) | 2 (mul1)");
    EXL3_CHECK_ARG(hidden % 128 == 0 && hidden >= 128 && hidden / 128 <= PS_MAX_ROW_BLOCKS, "exl3_pstep_create: hidden must be a multiple of 128 and <= %d", PS_MAX_ROW_BLOCKS * 128);
    EXL3_CHECK_ARG((head_dim == 64 || head_dim == 128) && heads_kv >= 1 && heads_q >= heads_kv && (heads_kv * head_dim) % 128 == 0 && (heads_q * head_dim) % 128 == 0,
                   "exl3_pstep_create: head_dim 64 | 128, whole 128-value blocks of q and kv");
    EXL3_CHECK_ARG(rope_mode == 1 || rope_mode == 2, "exl3_pstep_create: rope_mode 1 (GPT-J) | 2 (NeoX)");
    int dev = 0; EXL3_CHECK_HIP(hipGetDevice(&dev), "hipGetDevice");
    hipDeviceProp_t prop; EXL3_CHECK_HIP(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
    int ncu = prop.multiProcessorCount;
    // (test hook: EXL3_HIP_PSTEP_NCU plans for fewer CUs than the device has -- two tensor-parallel ranks sharing ONE GPU, each launching on a stream whose CU mask
    //  holds that many CUs, so that both grids are co-resident: tests/test_gpu_pstep_tp.py)
    if (const char* e = getenv("EXL3_HIP_PSTEP_NCU")) { const int v = atoi(e); if (v >= 16 && v <= ncu) ncu = v; }
    EXL3_CHECK_ARG(ncu >= 16, "exl3_pstep_create: device has %d CUs", ncu);
    const int qdim = heads_q * head_dim, kvdim = heads_kv * head_dim, kvb = kvdim / 128;
    const int nops = 4 * n_layers + 1;
    EXL3_CHECK_ARG(nops < 4095, "exl3_pstep_create: %d ops: the tags of the step's lines carry the producer op in 12 bits (at most 1023 layers)", nops);
    // ---- tensor parallelism inside the step (flags bits 12..15 = ranks, 16..19 = this rank; the tensors passed are THIS rank's shards: q / k / v / gate / up column shards,
    // o / down row shards, an lm_head column shard; heads_q / heads_kv are the rank's; hidden is the model's).  The all-reduce behind o_proj / down_proj
    // (model/model_tp_backend.py:119-126) happens on the step's row edges: every rank pushes its partial lines into every rank's exchange buffer.
    const int tpw = ((flags >> 12) & 0xf) > 1 ? ((flags >> 12) & 0xf) : 1, tpr = (flags >> 16) & 0xf;
    EXL3_CHECK_ARG(tpw <= 8 && tpr < tpw, "exl3_pstep_create: tensor-parallel rank %d of %d", tpr, tpw);
    // ---- bits per weight / codebook per op.  The tensors of one fused linear share both (a reference qgroup is quantized as one: modules/attn.py:244-308,
    // modules/mlp.py:537-574); the fused linears of the layers may have TWO adjacent widths (a fractional-bpw checkpoint: conversion/allocation.py:131-141 bumps whole
    // qgroups by one bit), the lm_head its own (flags bits 8..11 when its tensor does not say), one codebook for everything
    std::vector<int> opK((size_t) nops, 0);
    int cbm = -1, Klo = 99, Khi = 0;
    {
        auto kind = [&] (const exl3_pstep_linear_t& l, int& Ko, int& cbo) { Ko = l.K ? l.K : K; cbo = l.K ? l.cb : cb; };
        for (int li = 0; li < n_layers; ++li)
        {
            const exl3_pstep_linear_t* grp[4][3] = { { &layers[li].q, &layers[li].k, &layers[li].v }, { &layers[li].o, nullptr, nullptr },
                                                     { &layers[li].gate, &layers[li].up, nullptr }, { &layers[li].down, nullptr, nullptr } };
            for (int g = 0; g < 4; ++g)
                for (int j = 0; j < 3 && grp[g][j]; ++j)
                {
                    int Kj, cbj; kind(*grp[g][j], Kj, cbj);
                    if (cbm < 0) cbm = cbj;
                    EXL3_CHECK_ARG(cbj == cbm, "exl3_pstep_create: layer %d: tensors of different codebooks (%d / %d) in one model", li, cbj, cbm);
                    if (j == 0) opK[(size_t) 4 * li + g] = Kj;
                    EXL3_CHECK_ARG(Kj == opK[(size_t) 4 * li + g], "exl3_pstep_create: layer %d: the tensors of one fused linear (q / k / v; gate / up) differ in bits per weight", li);
                    if (Kj < Klo) Klo = Kj;
                    if (Kj > Khi) Khi = Kj;
                }
        }
        int Kh, cbh; kind(*head, Kh, cbh);
        if (!head->K && ((flags >> 8) & 0xf)) Kh = (flags >> 8) & 0xf;           // flags bits 8..11: the lm_head's bits per weight when its tensor does not carry them
        EXL3_CHECK_ARG(cbh == cbm, "exl3_pstep_create: the lm_head's codebook (%d) differs from the layers' (%d)", cbh, cbm);
        opK[(size_t) nops - 1] = Kh;
    }
    const int KH = opK[(size_t) nops - 1];
    EXL3_CHECK_ARG(Khi - Klo <= 1, "exl3_pstep_create: the layers' linears have %d .. %d bits per weight: the step takes one width or two adjacent ones", Klo, Khi);
    const int cb_all = cbm;
    K = Klo; const int K2 = Khi;
    const PsKernelSet* const kset = ps_kset(K);
    {
        // the step's edges need the WHOLE grid co-resident (one workgroup per CU): the kernel must fit a CU (155 KiB of LDS, 1024 threads, no scratch) -- checked here, at
        // create, instead of discovered as a time-out at the first step; a grid of exactly multiProcessorCount workgroups is then placed one per CU
        int occ = -1;
        const int r = kset ? kset->prepare(K2, KH, cb_all, (flags & 4) != 0, tpw > 1, &occ) : 1;
        if (r < 0) return r;
        EXL3_CHECK_ARG(r == 0, "exl3_pstep_create: no kernel for layers of %d%s bits per weight, a %d-bit lm_head and codebook %d%s (exl3_pstep.kspec.hip lists the instantiations)",
                       K, K2 != K ? " / + 1" : "", KH, cb_all, tpw > 1 ? " as a tensor-parallel rank" : "");
        EXL3_CHECK_ARG(occ != 0, "exl3_pstep_create: the step kernel does not fit a CU of this device (occupancy 0): the grid could not be co-resident");
    }
    // DIRECT residual edges (exl3_pstep.cuh) unless flags bit 1 / EXL3_HIP_PSTEP_OWNERS=1 asks for the owner form everywhere (A/B runs); a plan that cannot keep
    // an RMSNorm op's slice at <= 4 blocks falls back to owners for the whole step
    // attention inside the step (flags bit 2): head_dim 128, <= 8 query heads per kv head, the chip holds one item per (kv head, split) with 2..32 splits
    const bool attn = (flags & 4) != 0;
    const int att_blocks = kvdim / 128, att_gq = heads_q / heads_kv;
    int att_nsplit = 0;
    if (attn)
    {
        EXL3_CHECK_ARG(heads_q % heads_kv == 0 && att_gq * (128 / head_dim) <= 8, "exl3_pstep_create: attention inside the step: at most 8 query heads per 128-value kv block (head_dim 128: 8, head_dim 64: 4 per kv head)");
        const int cap = head_dim == 128 ? PS_ATT_MAX_SPLITS : 16;                  // (head_dim 64: a head's statistics live in 16 lanes)
        att_nsplit = ncu / att_blocks; if (att_nsplit > cap) att_nsplit = cap;
        EXL3_CHECK_ARG(att_nsplit >= 1, "exl3_pstep_create: more kv heads than CUs");
    }
    bool direct = !(flags & 2);
    if (const char* e = getenv("EXL3_HIP_PSTEP_OWNERS")) if (atoi(e)) direct = false;
    {
        const int ncb3[3] = { qdim / 128, kvdim / 128, kvdim / 128 }; const int ncb2[2] = { layers[0].gate.n / 128, layers[0].gate.n / 128 };
        OpPlan t;
        if (direct && !(plan_op(ncu, hidden / 128, ncb3, 3, PS_IN_NORM, PS_OUT_SLAB, t, true) && plan_op(ncu, hidden / 128, ncb2, 2, PS_IN_NORM, PS_OUT_SLAB, t, true))) direct = false;
    }
    std::vector<PsOp> ops((size_t) nops + 1);                       // + 1: the kernel forms the address of ops[nops] (never read)
    std::vector<PsTile> tiles((size_t) nops * ncu);
    memset(ops.data(), 0, ops.size() * sizeof(PsOp));
    size_t slab_a_floats = 0, slab_b_floats = 0, slab_c_floats = 0, slab_d_floats = 0;
    int rver = 0; std::vector<int> reader_of_version;               // op index of the RMSNorm op that reads version v of the residual row
    reader_of_version.push_back(0);
    std::string desc;
    struct Pending { int op; int which; size_t off[PS_MAX_MATS]; int parity; };
    std::vector<Pending> slab_fix;

    auto add_tiles = [&] (int op, const OpPlan& p, const int* ncb, int nmat, int nblk, int side_tasks)
    {
        // (units every wave decodes ahead: the kernel's PM_MAIN / PM_2 -- three, two at 8 bits, for the wider of two widths when it is 6, and for 3INST / mcg)
        fill_tiles(tiles.data() + (size_t) op * ncu, ncu, p, ncb, nmat, nblk, side_tasks, true, (cb_all != EXL3_CB_MUL1 || opK[(size_t) op] >= 8 || (K2 != K && opK[(size_t) op] == K2 && K2 >= 6)) ? 2 : 3);
    };

    int op = 0;
    char line[256];
    for (int li = 0; li < n_layers; ++li)
    {
        const exl3_pstep_layer_t& L = layers[li];
        EXL3_CHECK_ARG(L.q.k == hidden && L.k.k == hidden && L.v.k == hidden && L.q.n == qdim && L.k.n == kvdim && L.v.n == kvdim && L.o.k == qdim && L.o.n == hidden
                       && L.gate.k == hidden && L.up.k == hidden && L.gate.n == L.up.n && L.down.k == L.gate.n && L.down.n == hidden && L.gate.n % 128 == 0,
                       "exl3_pstep_create: layer %d: shapes do not form a Llama block", li);
        EXL3_CHECK_ARG(L.norm1 && L.norm2 && L.k_cache && L.k_scales && L.v_cache && L.v_scales, "exl3_pstep_create: layer %d: null norm / cache pointer", li);
        const int inter = L.gate.n;
        // q|k|v
        {
            PsOp& O = ops[op]; O.in_type = PS_IN_NORM; O.out_type = PS_OUT_SLAB; O.k = hidden; O.nmat = 3; O.eps = eps; O.norm_w = (const half_t*) L.norm1;
            O.rver = rver; O.gate_op = -1; if ((int) reader_of_version.size() <= rver) reader_of_version.resize(rver + 1); reader_of_version[rver] = op;
            lin_to_mat(L.q, O.mat[0]); lin_to_mat(L.k, O.mat[1]); lin_to_mat(L.v, O.mat[2]);
            const int ncb[3] = { qdim / 128, kvdim / 128, kvdim / 128 };
            OpPlan p; EXL3_CHECK_ARG(plan_op(ncu, hidden / 128, ncb, 3, O.in_type, O.out_type, p, direct), "exl3_pstep_create: no plan for q|k|v");
            if (direct && rver >= 1)
            {
                PsOp& Pd = ops[op - 1];                                 // the previous layer's down_proj: its partial lines are gathered here, it has no owner stage
                O.in_type |= PS_DIRECT; Pd.out_type |= PS_DIRECT; O.S_in = Pd.S; O.in_svh[0] = Pd.mat[0].svh;
            }
            O.S = p.S; add_tiles(op, p, ncb, 3, hidden / 128, 0);
            Pending f; f.op = op; f.which = 0; f.parity = li & 1; size_t off = 0;
            for (int i = 0; i < 3; ++i) { f.off[i] = off; off += (size_t) ncb[i] * p.S * 128; }
            slab_fix.push_back(f); if (off > slab_a_floats) slab_a_floats = off;
            if (li == 0) { snprintf(line, sizeof(line), "qkv: S=%d groups=%d+%d+%d tile<=%dx%d units; ", p.S, p.g[0], p.g[1], p.g[2], p.wmax, p.hmax); desc += line; }
            ++op;
        }
        // o_proj (input: q finished from the q|k|v slabs; side jobs: K / V append)
        {
            PsOp& O = ops[op]; const PsOp& Pq = ops[op - 1];
            O.in_type = PS_IN_QKV; O.out_type = PS_OUT_ATOMIC; O.k = qdim; O.nmat = 1; O.eps = eps; O.S_in = Pq.S; O.hd = head_dim; O.kvb = kvb; O.rope_mode = rope_mode;
            lin_to_mat(L.o, O.mat[0]);
            O.in_svh[0] = (const half_t*) L.q.svh; O.in_svh[1] = (const half_t*) L.k.svh; O.in_svh[2] = (const half_t*) L.v.svh;
            O.k_cache = (uint32_t*) L.k_cache; O.k_scales = (half_t*) L.k_scales; O.v_cache = (uint32_t*) L.v_cache; O.v_scales = (half_t*) L.v_scales;
            const int ncb[1] = { hidden / 128 };
            OpPlan p; EXL3_CHECK_ARG(plan_op(ncu, qdim / 128, ncb, 1, O.in_type, O.out_type, p, false, 32 / tpw), "exl3_pstep_create: no plan for o_proj");
            O.S = p.S; add_tiles(op, p, ncb, 1, qdim / 128, attn ? att_blocks * att_nsplit : 2 * kvb);      // side: K / V append tasks, or (PS_ATTN) attention items h * nsplit + s
            if (attn)
            {
                O.in_type |= PS_ATTN;
                PsAtt* A = (PsAtt*) &O.mat[1];
                A->rec = nullptr; A->stats = nullptr; A->gq = att_gq; A->nsplit = att_nsplit; A->hq = heads_q; A->hkv = heads_kv;      // (buffers: below)
            }
            O.rver = ++rver; O.gate_op = direct ? op - 1 : (rver >= 3 ? reader_of_version[rver - 2] : -1);       // (direct: the last readers of the lines an owner overwrites are the op before it)
            { Pending f; f.op = op; f.which = 2; f.off[0] = 0; f.parity = PS_ROW_PARITY(li); slab_fix.push_back(f); const size_t n = (size_t) ncb[0] * p.S * tpw * 128; if (n > slab_c_floats) slab_c_floats = n; }
            if (li == 0) { snprintf(line, sizeof(line), "o: S=%d groups=%d tile<=%dx%d; ", p.S, p.g[0], p.wmax, p.hmax); desc += line; }
            ++op;
        }
        // gate|up
        {
            PsOp& O = ops[op]; O.in_type = PS_IN_NORM; O.out_type = PS_OUT_SLAB; O.k = hidden; O.nmat = 2; O.eps = eps; O.norm_w = (const half_t*) L.norm2;
            O.rver = rver; O.gate_op = -1; if ((int) reader_of_version.size() <= rver) reader_of_version.resize(rver + 1); reader_of_version[rver] = op;
            lin_to_mat(L.gate, O.mat[0]); lin_to_mat(L.up, O.mat[1]);
            const int ncb[2] = { inter / 128, inter / 128 };
            OpPlan p; EXL3_CHECK_ARG(plan_op(ncu, hidden / 128, ncb, 2, O.in_type, O.out_type, p, direct), "exl3_pstep_create: no plan for gate|up");
            if (direct)
            {
                PsOp& Po = ops[op - 1];                                 // this layer's o_proj
                O.in_type |= PS_DIRECT; Po.out_type |= PS_DIRECT; O.S_in = Po.S; O.in_svh[0] = Po.mat[0].svh;
            }
            O.S = p.S; add_tiles(op, p, ncb, 2, hidden / 128, 0);
            Pending f; f.op = op; f.which = 1; f.parity = 0; size_t off = 0;
            for (int i = 0; i < 2; ++i) { f.off[i] = off; off += (size_t) ncb[i] * p.S * 128; }
            slab_fix.push_back(f); if (off > slab_b_floats) slab_b_floats = off;
            if (li == 0) { snprintf(line, sizeof(line), "gate|up: S=%d groups=%d+%d tile<=%dx%d; ", p.S, p.g[0], p.g[1], p.wmax, p.hmax); desc += line; }
            ++op;
        }
        // down (input: silu(g) * u finished from the gate|up slabs)
        {
            PsOp& O = ops[op]; const PsOp& Pg = ops[op - 1];
            O.in_type = PS_IN_ACT; O.out_type = PS_OUT_ATOMIC; O.k = inter; O.nmat = 1; O.eps = eps; O.S_in = Pg.S;
            lin_to_mat(L.down, O.mat[0]);
            O.in_svh[0] = (const half_t*) L.gate.svh; O.in_svh[1] = (const half_t*) L.up.svh;
            const int ncb[1] = { hidden / 128 };
            OpPlan p; EXL3_CHECK_ARG(plan_op(ncu, inter / 128, ncb, 1, O.in_type, O.out_type, p, false, 32 / tpw), "exl3_pstep_create: no plan for down_proj");
            O.S = p.S; add_tiles(op, p, ncb, 1, inter / 128, 0);
            O.rver = ++rver; O.gate_op = direct ? op - 1 : (rver >= 3 ? reader_of_version[rver - 2] : -1);
            { Pending f; f.op = op; f.which = 3; f.off[0] = 0; f.parity = PS_ROW_PARITY(li); slab_fix.push_back(f); const size_t n = (size_t) ncb[0] * p.S * tpw * 128; if (n > slab_d_floats) slab_d_floats = n; }
            if (li == 0) { snprintf(line, sizeof(line), "down: S=%d groups=%d tile<=%dx%d; ", p.S, p.g[0], p.wmax, p.hmax); desc += line; }
            ++op;
        }
    }
    {
        EXL3_CHECK_ARG(head->k == hidden && head->n % 128 == 0, "exl3_pstep_create: lm_head shape");
        PsOp& O = ops[op]; O.in_type = PS_IN_NORM; O.out_type = PS_OUT_FINAL; O.k = hidden; O.nmat = 1; O.eps = eps; O.norm_w = (const half_t*) final_norm;
        O.rver = rver; O.gate_op = -1;
        lin_to_mat(*head, O.mat[0]);
        const int ncb[1] = { head->n / 128 };
        OpPlan p; EXL3_CHECK_ARG(plan_op(ncu, hidden / 128, ncb, 1, O.in_type, O.out_type, p), "exl3_pstep_create: no plan for the lm_head (a rectangle is at most 12 column blocks wide: vocab / 128 <= 12 x CUs)");
        O.S = p.S; fill_tiles(tiles.data() + (size_t) op * ncu, ncu, p, ncb, 1, hidden / 128, 0, true, (cb_all != EXL3_CB_MUL1 || KH != K || KH >= 8) ? 2 : 3);      // (units ahead: the kernel's head pass)
        snprintf(line, sizeof(line), "head: S=%d groups=%d tile<=%dx%d; residual edges: %s%s", p.S, p.g[0], p.wmax, p.hmax, direct ? "direct (consumer gathers)" : "owners", attn ? "; attention inside o_proj's preparation" : ""); desc += line;
        if (attn) { snprintf(line, sizeof(line), " (%d kv blocks x %d splits)", att_blocks, att_nsplit); desc += line; }
        ++op;
    }

    PsHandle* h = new PsHandle();
    h->K = K; h->K2 = K2; h->KH = KH; h->cb = cb_all; h->kset = kset; h->d_runs = nullptr; h->nops = nops;
    h->tp_world = tpw; h->tp_rank = tpr; h->d_xbuf = nullptr; h->xbuf_bytes = 0; h->d_peers = nullptr; for (int i = 0; i < 8; ++i) h->peer_ptr[i] = nullptr; h->ncu = ncu; h->pmax = 3; h->spin_limit = 1 << 17; h->n_layers = n_layers; h->desc = desc;
    h->d_att_rec = nullptr; h->d_att_stats = nullptr; h->attn = attn ? 1 : 0; h->d_repack = nullptr; h->repack_words = 0; h->h_err = nullptr;
    h->d_ops = nullptr; h->d_tiles = nullptr; h->d_cnt = nullptr; h->d_err = nullptr; h->d_dbg = nullptr; h->d_slab_a = nullptr; h->d_slab_b = nullptr; h->d_slab_c = nullptr; h->d_slab_d = nullptr; h->d_rbuf = nullptr; h->d_epoch = nullptr;
    h->cnt_bytes = (size_t) nops * 8 * 16 * 4; h->dbg_words = (flags & 1) ? (size_t) nops * ncu * PS_DBG_SLOTS : 0;
#ifdef PS_DBG_UNITS
    if (flags & 1) h->dbg_words = (size_t) (nops + 3) * ncu * PS_DBG_SLOTS;        // (diagnostic build: three more areas for the per-unit stamps of one SIMD's streaming waves)
#endif
    #define PS_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { exl3_set_error("exl3_pstep_create: %s", hipGetErrorString(e_)); exl3_pstep_destroy(h); return EXL3_ERR_HIP; } } while (0)
    // (the q|k|v slab lines: TWO sets, alternating by layer -- without the attention inside, the K / V append side job of layer i reads the k / v lines after its workgroup
    //  has published o_proj's partial rows, and with direct row edges nothing orders that read against layer i + 1's q|k|v writers: they now write the other set)
    PS_TRY(hipMalloc(&h->d_slab_a, 2 * slab_a_floats * 8)); PS_TRY(hipMalloc(&h->d_slab_b, slab_b_floats * 8));
    PS_TRY(hipMemset(h->d_slab_a, 0, 2 * slab_a_floats * 8)); PS_TRY(hipMemset(h->d_slab_b, 0, slab_b_floats * 8));
    if (tpw > 1)
    {
        // the EXCHANGE BUFFER of a tensor-parallel rank: the partial lines of o_proj and down_proj of ALL ranks ([column block][rank x S + slice] per op kind), one
        // fine-grained allocation (peer GPUs write it while this rank's kernel reads it) that the peers map through its IPC handle (exl3_pstep_tp_handle / _open_peer)
        h->xbuf_bytes = 2 * (slab_c_floats + slab_d_floats) * 8;
        PS_TRY(hipExtMallocWithFlags((void**) &h->d_xbuf, h->xbuf_bytes, hipDeviceMallocFinegrained));
        PS_TRY(hipMemset(h->d_xbuf, 0, h->xbuf_bytes));
        PS_TRY(hipMalloc(&h->d_peers, 8 * sizeof(unsigned long long))); PS_TRY(hipMemset(h->d_peers, 0, 8 * sizeof(unsigned long long)));
        h->peer_ptr[tpr] = h->d_xbuf;
    }
    else
    {
        // (two sets each, alternating by layer, like the q|k|v lines: a rank may be a layer ahead of the slowest reader of a peer)
        PS_TRY(hipMalloc(&h->d_slab_c, 2 * slab_c_floats * 8)); PS_TRY(hipMalloc(&h->d_slab_d, 2 * slab_d_floats * 8));
        PS_TRY(hipMemset(h->d_slab_c, 0, 2 * slab_c_floats * 8)); PS_TRY(hipMemset(h->d_slab_d, 0, 2 * slab_d_floats * 8));
    }
    PS_TRY(hipMalloc(&h->d_rbuf, (size_t) PS_RBUF_BYTES)); PS_TRY(hipMemset(h->d_rbuf, 0, (size_t) PS_RBUF_BYTES));
    if (attn)
    {
        const size_t rec_bytes = (size_t) heads_q * att_nsplit * 512, st_bytes = (size_t) heads_q * PS_ATT_MAX_SPLITS * 16;
        PS_TRY(hipMalloc(&h->d_att_rec, rec_bytes)); PS_TRY(hipMemset(h->d_att_rec, 0, rec_bytes));
        PS_TRY(hipMalloc(&h->d_att_stats, st_bytes)); PS_TRY(hipMemset(h->d_att_stats, 0, st_bytes));
        for (int i = 0; i < nops; ++i) if ((ops[i].in_type & 0xff) == PS_IN_QKV) { PsAtt* A = (PsAtt*) &ops[i].mat[1]; A->rec = h->d_att_rec; A->stats = h->d_att_stats; }
    }
    for (const Pending& f : slab_fix)
    {
        PsOp& O = ops[f.op];
        unsigned long long* const xc = tpw > 1 ? (unsigned long long*) h->d_xbuf : h->d_slab_c;
        unsigned long long* const xd = tpw > 1 ? (unsigned long long*) h->d_xbuf + 2 * slab_c_floats : h->d_slab_d;
        unsigned long long* base = f.which == 0 ? h->d_slab_a + (size_t) f.parity * slab_a_floats : f.which == 1 ? h->d_slab_b
                                 : f.which == 2 ? xc + (size_t) f.parity * slab_c_floats : xd + (size_t) f.parity * slab_d_floats;
        if (f.which >= 2) O.xoff = (long long) ((const char*) base - (const char*) (tpw > 1 ? (unsigned long long*) h->d_xbuf : base));
        for (int i = 0; i < O.nmat; ++i) O.mat[i].slab = base + f.off[i];
        if (f.which >= 2)
        {
            // partial lines of an op that adds into the row: read by its own owners, or (DIRECT) by the RMSNorm op that follows
            if (O.out_type & PS_DIRECT) ops[f.op + 1].in_slab[0] = base + f.off[0];
            continue;
        }
        PsOp& C = ops[f.op + 1];                                      // the consumer op reads these slab sets
        for (int i = 0; i < O.nmat; ++i) C.in_slab[i] = base + f.off[i];
    }
    for (int i = 0; i < nops; ++i) { ops[i].K = opK[(size_t) i]; ops[i].cb = cb_all; ops[i].Bp = nullptr; }
    // lines per column block of every op's output, and what its consumer gathers: tp_world x S for the row shards (o_proj, down_proj) of a tensor-parallel rank
    for (int i = 0; i < nops; ++i)
    {
        PsOp& O = ops[i];
        const bool row_shard = tpw > 1 && (O.out_type & 0xff) == PS_OUT_ATOMIC;
        O.tp_world = tpw; O.S_all = row_shard ? O.S * tpw : O.S; O.line0 = row_shard ? tpr * O.S : 0;
        if (row_shard && (O.out_type & PS_DIRECT)) ops[i + 1].S_in = O.S_all;
    }
    // runs of equal width for the streaming waves: { end op, K } pairs (+ a terminator)
    std::vector<int> runs;
    for (int i = 0; i < nops; ++i) if (i + 1 == nops || opK[(size_t) i + 1] != opK[(size_t) i]) { runs.push_back(i + 1); runs.push_back(opK[(size_t) i]); }
    runs.push_back(nops); runs.push_back(0);
    // the ops' weights in plan order (flags bit 4 / EXL3_HIP_PSTEP_REPACK=0: stream the checkpoint layout as it is -- A/B runs, and callers that cannot afford the second copy:
    // the plan then holds pointers into the caller's tensors only)
    bool repack = !(flags & 16);
    if (const char* e = getenv("EXL3_HIP_PSTEP_REPACK")) repack = atoi(e) != 0;
    std::vector<size_t> rp_off((size_t) nops, 0);
    if (repack)
    {
        size_t words = 0;
        for (int i = 0; i < nops; ++i)
        {
            rp_off[i] = words;
            const PsTile* T = tiles.data() + (size_t) i * ncu;
            size_t units = 0; for (int c = 0; c < ncu; ++c) if (T[c].mat >= 0) units += (size_t) 4 * T[c].nb * T[c].ncb;
            words += units * (size_t) (128 * ops[i].K);
        }
        h->repack_words = words;
        PS_TRY(hipMalloc(&h->d_repack, words * 4));
        for (int i = 0; i < nops; ++i) ops[i].Bp = h->d_repack + rp_off[i];
    }
    PS_TRY(hipMalloc(&h->d_ops, ops.size() * sizeof(PsOp))); PS_TRY(hipMemcpy(h->d_ops, ops.data(), ops.size() * sizeof(PsOp), hipMemcpyHostToDevice));
    PS_TRY(hipMalloc(&h->d_tiles, tiles.size() * sizeof(PsTile))); PS_TRY(hipMemcpy(h->d_tiles, tiles.data(), tiles.size() * sizeof(PsTile), hipMemcpyHostToDevice));
    if (repack)
    {
        for (int i = 0; i < nops; ++i)
            ps_repack_kernel<<<dim3(64, ncu), dim3(128), 0, 0>>>(h->d_ops + i, h->d_tiles + (size_t) i * ncu, h->d_repack + rp_off[i], ops[i].K, 0, 0, nullptr);
        PS_TRY(hipGetLastError());
        PS_TRY(hipDeviceSynchronize());
    }
    h->ops = ops; h->tiles = tiles; h->att_nsplit = att_nsplit; h->att_cap = head_dim == 128 ? PS_ATT_MAX_SPLITS : 16; h->head_dim = head_dim;
    PS_TRY(hipMalloc(&h->d_runs, runs.size() * sizeof(int))); PS_TRY(hipMemcpy(h->d_runs, runs.data(), runs.size() * sizeof(int), hipMemcpyHostToDevice));
    PS_TRY(hipMalloc(&h->d_cnt, h->cnt_bytes)); PS_TRY(hipMemset(h->d_cnt, 0, h->cnt_bytes));
    PS_TRY(hipMalloc(&h->d_err, 64)); PS_TRY(hipMemset(h->d_err, 0, 64));
    PS_TRY(hipHostMalloc((void**) &h->h_err, 64, hipHostMallocMapped)); h->h_err[0] = 0u;
    { void* dp = nullptr; PS_TRY(hipHostGetDevicePointer(&dp, h->h_err, 0)); PS_TRY(hipMemcpy((char*) h->d_err + 8, &dp, 8, hipMemcpyHostToDevice)); }      // (behind the error word: PsArgs::err)
    PS_TRY(hipMalloc(&h->d_epoch, 64)); { const uint32_t one = 1u; PS_TRY(hipMemset(h->d_epoch, 0, 64)); PS_TRY(hipMemcpy(h->d_epoch, &one, 4, hipMemcpyHostToDevice)); }
    if (h->dbg_words) { PS_TRY(hipMalloc(&h->d_dbg, h->dbg_words * 8)); PS_TRY(hipMemset(h->d_dbg, 0, h->dbg_words * 8)); }
    #undef PS_TRY
    *handle_out = h;
    return EXL3_OK;
}

// The planner alone (no device): the rectangles of one op kind of a Llama block for a chip of `ncu` CUs.  op_kind: 0 q|k|v, 1 o_proj, 2 gate|up, 3 down,
// 4 lm_head.  tiles_out: [ncu][12] = {mat, cb0, ncb, b0, nb, slice, side, flags, ubase, pA, pB, pC} (PsTile; pX = n << 2 | e: the streaming waves' shares by age
// group, all zero = uniform: wave_partition); S_out: its k-slices.  Host logic only: callable on a box without a GPU.
extern "C" int exl3_pstep_plan_tiles(int hidden, int inter, int heads_q, int heads_kv, int head_dim, int vocab, int ncu, int op_kind, int32_t* tiles_out, int* S_out)
{
    EXL3_CHECK_ARG(tiles_out && S_out && ncu >= 16 && ncu <= 1024 && hidden % 128 == 0 && inter % 128 == 0 && vocab % 128 == 0 && (head_dim == 64 || head_dim == 128),
                   "exl3_pstep_plan_tiles: bad shape");
    const int qdim = heads_q * head_dim, kvdim = heads_kv * head_dim;
    EXL3_CHECK_ARG(qdim % 128 == 0 && kvdim % 128 == 0, "exl3_pstep_plan_tiles: whole 128-value blocks of q and kv");
    int ncb[3] = { 0, 0, 0 }, nmat = 1, nblk = hidden / 128, in_type = PS_IN_NORM, out_type = PS_OUT_SLAB, side = 0;
    switch (op_kind)
    {
        case 0: ncb[0] = qdim / 128; ncb[1] = ncb[2] = kvdim / 128; nmat = 3; break;
        case 1: ncb[0] = hidden / 128; nblk = qdim / 128; in_type = PS_IN_QKV; out_type = PS_OUT_ATOMIC; side = 2 * (kvdim / 128); break;
        case 2: ncb[0] = ncb[1] = inter / 128; nmat = 2; break;
        case 3: ncb[0] = hidden / 128; nblk = inter / 128; in_type = PS_IN_ACT; out_type = PS_OUT_ATOMIC; break;
        case 4: ncb[0] = vocab / 128; out_type = PS_OUT_FINAL; break;
        default: exl3_set_error("exl3_pstep_plan_tiles: op_kind 0..4"); return EXL3_ERR_ARG;
    }
    OpPlan p;
    // the RMSNorm ops of the layers take the DIRECT-edge plan (slices of <= 4 blocks) when both have one, as exl3_pstep_create decides
    bool direct = false;
    if (op_kind == 0 || op_kind == 2)
    {
        const int ncb3[3] = { qdim / 128, kvdim / 128, kvdim / 128 }; const int ncb2[2] = { inter / 128, inter / 128 };
        OpPlan t;
        direct = plan_op(ncu, hidden / 128, ncb3, 3, PS_IN_NORM, PS_OUT_SLAB, t, true) && plan_op(ncu, hidden / 128, ncb2, 2, PS_IN_NORM, PS_OUT_SLAB, t, true);
        if (const char* e = getenv("EXL3_HIP_PSTEP_OWNERS")) if (atoi(e)) direct = false;
    }
    EXL3_CHECK_ARG(plan_op(ncu, nblk, ncb, nmat, in_type, out_type, p, direct), "exl3_pstep_plan_tiles: no plan for this op on %d CUs", ncu);
    std::vector<PsTile> T((size_t) ncu);
    fill_tiles(T.data(), ncu, p, ncb, nmat, nblk, side, true);
    memcpy(tiles_out, T.data(), (size_t) ncu * sizeof(PsTile));
    *S_out = p.S;
    return EXL3_OK;
}

static int ps_run(PsHandle* h, void* R, void* logits, void* q_out, const float* rope_sin, const float* rope_cos, const int64_t* slots,
                  const int32_t* block_table, const int32_t* seqlens, int blocks_per_seq, int page_size, float scale, void* stream);

extern "C" int exl3_pstep_run(void* handle, void* R, void* logits, void* q_out, const float* rope_sin, const float* rope_cos, const int64_t* slots, void* stream)
{
    PsHandle* h = (PsHandle*) handle;
    EXL3_CHECK_ARG(h && !h->attn, "exl3_pstep_run: this plan has the attention inside (exl3_pstep_run_attn)");
    return ps_run(h, R, logits, q_out, rope_sin, rope_cos, slots, nullptr, nullptr, 0, 0, 0.0f, stream);
}

extern "C" int exl3_pstep_run_attn(void* handle, void* R, void* logits, void* q_out, const float* rope_sin, const float* rope_cos, const int64_t* slots,
                                   const int32_t* block_table, const int32_t* cache_seqlens, int blocks_per_seq, int page_size, float scale, void* stream)
{
    PsHandle* h = (PsHandle*) handle;
    EXL3_CHECK_ARG(h && h->attn, "exl3_pstep_run_attn: the plan was created without the attention (flags bit 2)");
    EXL3_CHECK_ARG(block_table && cache_seqlens && blocks_per_seq >= 1 && page_size >= 16 && page_size % 16 == 0, "exl3_pstep_run_attn: block table / lengths / page size");
    return ps_run(h, R, logits, q_out, rope_sin, rope_cos, slots, block_table, cache_seqlens, blocks_per_seq, page_size, scale, stream);
}

static int ps_run(PsHandle* h, void* R, void* logits, void* q_out, const float* rope_sin, const float* rope_cos, const int64_t* slots,
                  const int32_t* block_table, const int32_t* seqlens, int blocks_per_seq, int page_size, float scale, void* stream)
{
    EXL3_CHECK_ARG(h && R && logits && rope_sin && rope_cos && slots, "exl3_pstep_run: null argument");
    hipStream_t st = (hipStream_t) stream;
    EXL3_CHECK_HIP(hipMemsetAsync(h->d_cnt, 0, h->cnt_bytes, st), "exl3_pstep_run: hipMemsetAsync");
    PsArgs a;
    a.ops = h->d_ops; a.tiles = h->d_tiles; a.nops = h->nops; a.ncu = h->ncu;
    a.R = (unsigned long long*) R; a.logits = (half_t*) logits; a.q_out = (half_t*) q_out;
    a.rope_sin = rope_sin; a.rope_cos = rope_cos; a.slots = slots;
    a.block_table = block_table; a.seqlens = seqlens; a.blocks_per_seq = blocks_per_seq; a.page_size = page_size; a.att_scale = scale;
    a.rbuf = h->d_rbuf; a.cnt = h->d_cnt; a.epoch = h->d_epoch; a.err = h->d_err; a.dbg = h->d_dbg; a.spin_limit = h->spin_limit; a.pmax = h->pmax;
    a.runs = h->d_runs;
    h->kset->launch(h->K2, h->KH, h->cb, h->attn != 0, h->tp_world > 1, h->ncu, st, a);
    return exl3_check_launch("exl3_pstep_run");
}

extern "C" int exl3_pstep_error(void* handle, void* stream)
{
    PsHandle* h = (PsHandle*) handle;
    EXL3_CHECK_ARG(h, "exl3_pstep_error: null handle");
    EXL3_CHECK_HIP(hipStreamSynchronize((hipStream_t) stream), "exl3_pstep_error: hipStreamSynchronize");
    uint32_t e = 0;
    EXL3_CHECK_HIP(hipMemcpy(&e, h->d_err, 4, hipMemcpyDeviceToHost), "exl3_pstep_error: hipMemcpy");
    if (e) EXL3_CHECK_HIP(hipMemset(h->d_err, 0, 4), "exl3_pstep_error: hipMemset");
    if (h->h_err) h->h_err[0] = 0u;
    return e ? 1 : 0;
}

// no synchronisation: the pinned host mirror of the error word, written by the kernel itself when a wait times out (system-scope store).  A caller that replays a captured
// step polls this between replays: != 0 means some EARLIER step timed out (its logits were NaN) -- exl3_pstep_error then confirms and clears.
extern "C" int exl3_pstep_error_peek(void* handle)
{
    PsHandle* h = (PsHandle*) handle;
    EXL3_CHECK_ARG(h, "exl3_pstep_error_peek: null handle");
    return h->h_err && ((volatile uint32_t*) h->h_err)[0] ? 1 : 0;
}

// ---- tensor parallelism inside the step: the ranks exchange the IPC handles of their exchange buffers (64 bytes each, over the process group), map the peers' and commit.
// Every rank must then run the same sequence of steps (a step's lines carry the run epoch: the ranks' epochs advance together).
extern "C" int exl3_pstep_tp_handle(void* handle, void* handle64_out)
{
    PsHandle* h = (PsHandle*) handle;
    EXL3_CHECK_ARG(h && handle64_out && h->tp_world > 1 && h->d_xbuf, "exl3_pstep_tp_handle: the plan was not created as a tensor-parallel rank (flags bits 12..19)");
    hipIpcMemHandle_t ih;
    EXL3_CHECK_HIP(hipIpcGetMemHandle(&ih, h->d_xbuf), "exl3_pstep_tp_handle: hipIpcGetMemHandle");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    memcpy(handle64_out, &ih, 64);
    return EXL3_OK;
}

extern "C" int exl3_pstep_tp_open_peer(void* handle, int peer_rank, const void* handle64)
{
    PsHandle* h = (PsHandle*) handle;
    EXL3_CHECK_ARG(h && handle64 && h->tp_world > 1 && peer_rank >= 0 && peer_rank < h->tp_world && peer_rank != h->tp_rank && !h->peer_ptr[peer_rank], "exl3_pstep_tp_open_peer: bad arguments");
    hipIpcMemHandle_t ih; memcpy(&ih, handle64, 64);
    void* p = nullptr;
    EXL3_CHECK_HIP(hipIpcOpenMemHandle(&p, ih, hipIpcMemLazyEnablePeerAccess), "exl3_pstep_tp_open_peer: hipIpcOpenMemHandle");
    h->peer_ptr[peer_rank] = p;
    return EXL3_OK;
}

// all peers mapped: the table of the ranks' buffers goes to the device (behind the error word: PsArgs::err + 4).  The caller barriers the ranks AFTER this and before the first step.
extern "C" int exl3_pstep_tp_commit(void* handle)
{
    PsHandle* h = (PsHandle*) handle;
    EXL3_CHECK_ARG(h && h->tp_world > 1, "exl3_pstep_tp_commit: not a tensor-parallel plan");
    unsigned long long tab[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int r = 0; r < h->tp_world; ++r) { EXL3_CHECK_ARG(h->peer_ptr[r], "exl3_pstep_tp_commit: rank %d's exchange buffer is not mapped", r); tab[r] = (unsigned long long) h->peer_ptr[r]; }
    EXL3_CHECK_HIP(hipMemcpy(h->d_peers, tab, sizeof(tab), hipMemcpyHostToDevice), "exl3_pstep_tp_commit: hipMemcpy");
    EXL3_CHECK_HIP(hipMemcpy((char*) h->d_err + 16, &h->d_peers, 8, hipMemcpyHostToDevice), "exl3_pstep_tp_commit: hipMemcpy");
    return EXL3_OK;
}

// diagnostics: this rank's exchange buffer copied to host memory (synchronises the device); returns the bytes copied
extern "C" int64_t exl3_pstep_tp_peek(void* handle, void* host_out, int64_t max_bytes)
{
    PsHandle* h = (PsHandle*) handle;
    if (!h || !host_out || !h->d_xbuf) { exl3_set_error("exl3_pstep_tp_peek: not a tensor-parallel plan"); return EXL3_ERR_ARG; }
    const int64_t n = (int64_t) h->xbuf_bytes < max_bytes ? (int64_t) h->xbuf_bytes : max_bytes;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_out, h->d_xbuf, (size_t) n, hipMemcpyDeviceToHost) != hipSuccess) { exl3_set_error("exl3_pstep_tp_peek: hipMemcpy"); return EXL3_ERR_HIP; }
    return n;
}

// the attention item's geometry at a sequence length (host restatement of att_nse / att_make of the kernel): out3 = { splits in use, tokens per split, 128-token steps per split }
extern "C" int exl3_pstep_attn_geometry(void* handle, int len, int* out3)
{
    PsHandle* h = (PsHandle*) handle;
    EXL3_CHECK_ARG(h && out3 && len >= 1, "exl3_pstep_attn_geometry: null argument");
    EXL3_CHECK_ARG(h->attn, "exl3_pstep_attn_geometry: the plan was created without the attention");
    int nse = (len + 127) >> 7; if (nse < 1) nse = 1; if (nse > h->att_nsplit) nse = h->att_nsplit;
    const int st_tok = (((len + nse - 1) / nse) + 15) & ~15;
    out3[0] = nse; out3[1] = st_tok; out3[2] = (st_tok + 127) >> 7;
    return EXL3_OK;
}

// the inverse of the load-time repack for matrix `mat` of op `op`: the repacked words copied back into a checkpoint-layout tensor [k / 16][n / 16][16 K] (device memory,
// the caller's): equality with the original tensor proves the permutation (tests/test_gpu_pstep.py).  EXL3_ERR_ARG if the plan streams the checkpoint layout.
extern "C" int exl3_pstep_unpack_op(void* handle, int op, int mat, void* trellis_out, void* stream)
{
    PsHandle* h = (PsHandle*) handle;
    EXL3_CHECK_ARG(h && trellis_out && op >= 0 && op < h->nops && mat >= 0 && mat < h->ops[(size_t) op].nmat, "exl3_pstep_unpack_op: op / matrix out of range");
    EXL3_CHECK_ARG(h->d_repack && h->ops[(size_t) op].Bp, "exl3_pstep_unpack_op: this plan streams the checkpoint layout (created without the repack)");
    ps_repack_kernel<<<dim3(64, h->ncu), dim3(128), 0, (hipStream_t) stream>>>(h->d_ops + op, h->d_tiles + (size_t) op * h->ncu, (uint32_t*) h->ops[(size_t) op].Bp, h->ops[(size_t) op].K,
                                                                                1, mat, (uint32_t*) trellis_out);
    return exl3_check_launch("exl3_pstep_unpack_op");
}

extern "C" int exl3_pstep_set(void* handle, int decode_ahead_units, int spin_limit)
{
    PsHandle* h = (PsHandle*) handle;
    EXL3_CHECK_ARG(h && decode_ahead_units >= -1 && decode_ahead_units <= 3, "exl3_pstep_set: decode_ahead_units in 0..3 (-1: keep)");
    if (decode_ahead_units >= 0) h->pmax = decode_ahead_units;
    if (spin_limit > 0) h->spin_limit = spin_limit;
    return EXL3_OK;
}

extern "C" int exl3_pstep_describe(void* handle, char* buf, int buf_bytes)
{
    PsHandle* h = (PsHandle*) handle;
    EXL3_CHECK_ARG(h && buf && buf_bytes > 0, "exl3_pstep_describe: null argument");
    snprintf(buf, (size_t) buf_bytes, "ops=%d cus=%d K=%d%s head_K=%d codebook=%s decode_ahead=%d lds=%d weights=%s | %s", h->nops, h->ncu, h->K, h->K2 != h->K ? "+1" : "", h->KH,
             h->cb == 2 ? "mul1" : (h->cb == 1 ? "mcg" : "3inst"), h->pmax, (int) PS_LDS_BYTES,
             h->d_repack ? "repacked (one contiguous run per streaming wave)" : "checkpoint layout", h->desc.c_str());
    return EXL3_OK;
}

extern "C" int64_t exl3_pstep_stamps(void* handle, uint64_t* host_out, int64_t max_words, void* stream)
{
    PsHandle* h = (PsHandle*) handle;
    if (!h || !host_out) { exl3_set_error("exl3_pstep_stamps: null argument"); return EXL3_ERR_ARG; }
    if (!h->d_dbg) return 0;
    if (hipStreamSynchronize((hipStream_t) stream) != hipSuccess) { exl3_set_error("exl3_pstep_stamps: hipStreamSynchronize"); return EXL3_ERR_HIP; }
    const int64_t n = (int64_t) h->dbg_words < max_words ? (int64_t) h->dbg_words : max_words;
    if (hipMemcpy(host_out, h->d_dbg, (size_t) n * 8, hipMemcpyDeviceToHost) != hipSuccess) { exl3_set_error("exl3_pstep_stamps: hipMemcpy"); return EXL3_ERR_HIP; }
    return n;
}

extern "C" int exl3_pstep_destroy(void* handle)
{
    PsHandle* h = (PsHandle*) handle;
    if (!h) return EXL3_OK;
    if (h->d_ops) (void) hipFree(h->d_ops);
    if (h->d_tiles) (void) hipFree(h->d_tiles);
    if (h->d_cnt) (void) hipFree(h->d_cnt);
    if (h->d_err) (void) hipFree(h->d_err);
    if (h->d_epoch) (void) hipFree(h->d_epoch);
    if (h->d_dbg) (void) hipFree(h->d_dbg);
    if (h->d_att_rec) (void) hipFree(h->d_att_rec);
    if (h->d_att_stats) (void) hipFree(h->d_att_stats);
    if (h->d_slab_a) (void) hipFree(h->d_slab_a);
    if (h->d_slab_b) (void) hipFree(h->d_slab_b);
    if (h->d_slab_c) (void) hipFree(h->d_slab_c);
    if (h->d_slab_d) (void) hipFree(h->d_slab_d);
    if (h->d_rbuf) (void) hipFree(h->d_rbuf);
    if (h->d_repack) (void) hipFree(h->d_repack);
    for (int i = 0; i < 8; ++i) if (h->peer_ptr[i] && i != h->tp_rank) (void) hipIpcCloseMemHandle(h->peer_ptr[i]);
    if (h->d_xbuf) (void) hipFree(h->d_xbuf);
    if (h->d_peers) (void) hipFree(h->d_peers);
    if (h->d_runs) (void) hipFree(h->d_runs);
    if (h->h_err) (void) hipHostFree(h->h_err);
    delete h;
    return EXL3_OK;
}
