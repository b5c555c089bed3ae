// Persistent decode step ("generation 5"): plan structures shared by the kernel (exl3_pstep.kspec.hip) and the planner / C ABI (exl3_pstep.hip).
//
// One launch runs EVERY quantized linear of a batch-1 decode step (all layers + lm_head) with the glue between them (RMSNorm, q|k|v epilogue with
// RoPE and the 4-bit K / V append, silu * mul, residual adds); reference graphs it replaces: libtorch/attention.cpp:246-330 + libtorch/mlp.cpp:14-91 per
// layer (the reference itself keeps input Hadamard -> stream -> output Hadamard of ONE linear inside one cooperative launch with two grid syncs:
// quant/exl3_gemv_kernel.cuh:138-402).  Grid = one 16-wave workgroup per CU, all co-resident; an "op" = one fused GEMV (q|k|v, o, gate|up, down,
// lm_head); what one CU hands to another between two ops (the EDGE) travels as TAGGED LINES -- data and flag in one 16-byte store, tag = (run epoch, producer
// op) -- that the consumer re-loads until every tag is the producer's: no arrival counter is polled on the data path.  What a wave does while it waits at an
// edge is the point of the design: the weights of the next op do not depend on the activations, so the wave already has them in flight and DECODES its first
// work units into registers / LDS (decode-ahead: three units); when the activations arrive those units cost one MFMA pass instead of 235 VALU instructions each.
#pragma once
#include <stdint.h>
#include "exl3_common.cuh"

#define PS_WAVES 16
#define PS_NT (64 * PS_WAVES)
#define PS_MAX_MATS 3
#define PS_IN_NORM 0       // input = RMSNorm(R) (R: the 64-bit fixed-point residual), exact row scale (every CU reads the whole row)
#define PS_IN_QKV 1        // input = q finished from the q|k|v op's slabs (output Hadamard, svh, RoPE, fp16); side job: the new token's K / V append
#define PS_IN_ACT 2        // input = silu(g) * u finished from the gate|up op's slabs
#define PS_OUT_SLAB 0      // raw rotated-basis partial rows -> slab[colblock][slice][128] tagged granules (finished by the consumer op's preparation; no edge)
#define PS_OUT_ATOMIC 1    // the op adds into the residual row: partial rows as tagged lines, the OWNER of each 128-value block (the slice-0 workgroup) sums them, applies the
                           // output Hadamard + svh, adds the block of the previous row version and publishes the new version's block as one tagged fp32 line
#define PS_OUT_FINAL 2     // one slice: finished fp16 rows (lm_head)
// DIRECT residual edges (bit 8 of in_type / out_type; the low byte is the type): the consumer RMSNorm op finishes the row's new version ITSELF -- every workgroup
// gathers the producer's partial lines of the blocks of ITS k-slice (one hop instead of partial lines -> owner -> owner's line -> every workgroup: two), adds the
// previous version's blocks, normalises with the scale it last knew and corrects its partial sums by (true scale / that scale) when the block sums of squares
// -- published with the new blocks by one workgroup per slice -- have arrived (long before the streaming ends).  The producer op then has no owner stage.
#define PS_DIRECT 0x100
// ATTENTION inside the step (bit 9 of o_proj's in_type): o_proj's preparation then (a) runs ONE item of the decode attention over the quantized cache -- kv block h, context
// split s of `nsplit` (tile.side = h * nsplit + s): q / k / v finished from the q|k|v op's slab lines as exl3_attn_decode_qcache_split_qkv does (RoPE, 4-bit append of the
// new token by the split that holds it), the matrix-pipe split kernel's token loop on eight waves (the four service waves + streaming waves 0-3: two per SIMD), the
// partial record published as 32 tagged fp16 granules + a statistics granule -- and (b) merges the records of the query heads of ITS k-slice (all splits in use: one hop,
// eight records per round trip) into o_proj's input.  The splits in use follow the length on the device.  Reference: libtorch/attention.cpp:246-504.
#define PS_ATTN 0x200

// LDS map of the kernel (bytes)
#define PS_QUADS_BYTES 8448                       // activation quads of a slice: (nb * 8 + 4) tile rows x 32 B  ->  nb <= 32 Hadamard blocks (a 64-block slice -- the lm_head of a
                                                  // model wider than 4096 -- runs 8 KiB into the gather area behind them: exl3_pstep_kernel.cuh)
#define PS_MISC_BYTES 1536                        // block sums of squares [64] | block sums [2][64] | segment records [2][16][4] | control words
#define PS_PART_BYTES (12 * 2 * 512)              // partial rows [streaming wave][segment][128] fp32
#define PS_PDEC_BYTES (12 * 16 * 64 * 8)          // decode-ahead unit #2 of every streaming wave: [wave][16 operand slots][64 lanes] x 8 B
#define PS_GATH_BYTES (4 * 8 * 512)                // finishers of residual-row blocks (owners; direct RMSNorm ops whose producer has > 8 slices): [4 blocks][8 service half-waves][128] fp32 gathered sums
#define PS_ATT_BYTES 21504                         // attention item (with the gather area in front of it: 37 888 B): V tiles / partial outputs 8 x 4352 | new-token words 128 | scales 128 | wave statistics 512 | queries 2048
#define PS_ATT_MAX_SPLITS 32                       // statistics of all splits of a head in one half-wave (one lane per split)
#define PS_DBG_SLOTS 32                           // phase stamps per op and workgroup (exl3_pstep_stamps): 0..12 streaming wave 0 / service wave 0, 16 + w: streaming wave w done, 28 + s: service wave s published its quads
#define PS_MAX_ROW_BLOCKS 64                      // the residual row: hidden <= 8192 (round 6; Llama-3.1-70B's width): 8 service half-waves x 8 blocks
#define PS_ROW_IT (PS_MAX_ROW_BLOCKS / 8)
#define PS_RROW_BYTES (PS_MAX_ROW_BLOCKS * 1024u)  // one version of the row: a 1 KiB tagged line per block
#define PS_RBUF_BYTES (2 * PS_MAX_ROW_BLOCKS * 1024 + 2 * PS_MAX_ROW_BLOCKS * 16)
#define PS_MAX_SLICE_BLOCKS 32                    // the largest k-slice of an op (quads in LDS); the lm_head's one slice may be the whole row

struct PsMat
{
    const uint32_t* B; const half_t* suh; const half_t* svh;
    unsigned long long* slab;          // PS_OUT_SLAB: [n / 128][S] partial lines (PS_PLINE_BYTES = 512: 32 tagged fp16 granules { v0 v1, tag, v2 v3, tag }; exl3_pstep_kernel.cuh)
    int n, tiles_n;
};

struct PsOp
{
    int in_type, out_type, k, nmat;
    int S, S_in, hd, kvb;              // S: k-slices of this op (= slab lines per column block); S_in: slab lines of the producer op
    int rope_mode, rver;               // rver: version of the residual row a PS_IN_NORM op reads / a PS_OUT_ATOMIC op produces (0 = the caller's fixed-point R)
    float eps; int gate_op;            // PS_OUT_ATOMIC: the op whose read gate protects the row lines this op's owners overwrite (-1: none)
    PsMat mat[PS_MAX_MATS];
    const half_t* norm_w;              // PS_IN_NORM
    const unsigned long long* in_slab[3];   // PS_IN_QKV: q, k, v slab sets of the producer; PS_IN_ACT: gate, up; PS_IN_NORM | PS_DIRECT: [0] = the producer's partial lines (S_in per block)
    const half_t* in_svh[3];                // ... and the producer's column scales
    uint32_t* k_cache; half_t* k_scales; uint32_t* v_cache; half_t* v_scales;      // PS_IN_QKV: the layer's 4-bit paged cache
    // ---- round 6 (the descriptor is 320 B = five 64-byte lines now)
    const uint32_t* Bp;                // REPACKED weights of the op (null: the checkpoint's tile-row-major tensors through mat[].B): the op's work units in the order the
                                       // plan's workgroups and streaming waves take them -- [CU rectangle][unit], unit = 2 tile rows x 128 columns = 128 K contiguous
                                       // words -- so that every streaming wave reads ONE contiguous run (re-ordering tiles at load time is a legal
                                       // one-time transform; exl3_pstep_create permutes on the device, exl3_pstep_unpack_op inverts it for the bit-exactness test)
    int K, cb;                         // bits per weight and codebook of THIS op's matrices (a qgroup shares both: conversion/allocation.py:131-141 bumps whole qgroups)
    // tensor parallelism inside the step (round 6; model/model_tp_backend.py:119-126, modules/attn.py:547, modules/mlp.py:770: the all-reduce behind o_proj / down_proj).
    // A row-sharded op's partial row is pushed into every rank's EXCHANGE BUFFER (fine-grained memory, IPC-mapped by the peers: PsArgs::err + 4 holds the table of the
    // ranks' base addresses) at byte offset xoff: line (column block, line0 + slice) of S_all = tp_world x S lines per block.  Consumers sum the S_all lines in index order:
    // the same bits on every rank.
    int S_all, line0, tp_world, pad0_;
    long long xoff;
    int pad_[6];
};
static_assert(sizeof(PsOp) == 320, "PsOp: five 64-byte lines (the service waves prefetch the next op's descriptor line by line)");

// o_proj with PS_ATTN: overlays mat[1] .. mat[2] of its PsOp (o_proj has one matrix)
struct PsAtt
{
    unsigned long long* rec;           // [hq][nsplit] records of 512 B: the split's 128 output accumulators of the head (rotated domain, un-normalised) as fp16 pairs,
                                       // lane l's granule { O[4l] O[4l+1], tag, O[4l+2] O[4l+3], tag }
    unsigned long long* stats;         // [hq][PS_ATT_MAX_SPLITS] granules { running maximum (natural log units), tag, sum, tag }
    int gq, nsplit, hq, hkv;
};
static_assert(sizeof(PsAtt) <= 2 * sizeof(PsMat), "PsAtt overlays two PsMat slots");

// what ONE workgroup (CU) does in one op: a rectangle of (ncb column blocks of matrix mat) x (nb Hadamard blocks of k); mat < 0: nothing (it still meets the edge)
struct PsTile { int mat, cb0, ncb, b0, nb, slice, side, flags, ubase, r0_, r1_, r2_; };      // ubase: first work unit of the rectangle in the op's repacked weights (PsOp::Bp)
#define PS_TILE_INTS 12
#define PS_TILE_Q_OUT 1               // this workgroup also stores the finished q blocks of its slice (one column group per slice); in a DIRECT RMSNorm op: it publishes
                                      // the new row version's blocks of its slice and their sums of squares

struct PsArgs
{
    const PsOp* ops; const PsTile* tiles; int nops, ncu;
    unsigned long long* R; half_t* logits; half_t* q_out;
    unsigned long long* rbuf;         // [2][PS_MAX_ROW_BLOCKS] lines of 1 KiB: the residual row's versions >= 1 (version v in half v & 1), tagged fp32 pairs;
                                      // then [2][PS_MAX_ROW_BLOCKS] granules of 16 B { sum of squares of the block, tag, 0, tag }
    const float* rope_sin; const float* rope_cos; const int64_t* slots;
    const int32_t* block_table; const int32_t* seqlens;      // PS_ATTN: the sequence's page ids [blocks_per_seq], its length INCLUDING the new token [1]
    int blocks_per_seq, page_size; float att_scale;
    uint32_t* cnt;                    // [nops][8 shards][16 words]: arrivals of edge `op`, zero at launch
    uint32_t* epoch;                  // run counter in device memory (tags of the slab granules): read at entry, + 1 at exit
    uint32_t* err;                    // sticky: bit 0 = an edge timed out, bit 1 = a tagged slab line never arrived (results invalid); words 2..3: the address of the pinned
                                      // host mirror (set to 1 by the wave that times out: a caller reads it without synchronising) or null; words 4..5: the address of the table of the tensor-parallel
                                      // ranks' exchange buffers (device memory, tp_world x 8 B) or null
    unsigned long long* dbg;          // optional phase stamps [nops][ncu][PS_DBG_SLOTS] (100 MHz)
    int spin_limit, pmax;
    const int* runs;                  // the ops in runs of equal bits per weight: { end op, K } pairs (the streaming waves' loop is instantiated per width)
};

