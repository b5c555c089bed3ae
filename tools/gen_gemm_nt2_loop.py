#!/usr/bin/env python3
"""Generator of the hand-scheduled main loop of the prefill NT GEMM (exllamav3_amd/csrc/exl3_gemm_nt2.hip):

    python tools/gen_gemm_nt2_loop.py > exllamav3_amd/csrc/exl3_gemm_nt2_loop.inc
    python tools/gen_gemm_nt2_loop.py clobbers > exllamav3_amd/csrc/exl3_gemm_nt2_clobbers.inc

The output is ONE inline-asm body (a C string literal) with every register named by hand -- hipcc cannot allocate this structure (256 accumulators +
two fragment sets; left to the allocator: 732 B of scratch and 1302 accumulator moves around 128 matrix instructions).  The structure, per workgroup of four waves (one per SIMD):

  * wave tile 128 x 128 = 4 x 4 tiles of v_mfma_f32_32x32x16_f16, accumulators a[0:255] (tile (im, jn) = a[16 (4 im + jn) : +15]).  The W^T fragment is the
    instruction's first source and the activation fragment its second, so D is the TRANSPOSED tile: lane l holds row m = l & 31 of C and four runs of four consecutive
    columns -- the epilogue packs them into 8-byte LDS stores of the fp16 C tile.
  * K-tiles of 32 in FOUR LDS buffers of 32 KiB (A 16 KiB | W^T 16 KiB; 64-byte rows, 16-byte chunk c of row r at position c ^ ((r >> 2) & 3): the sixteen-lane groups of a
    ds_read_b128 -- {0-3, 12-15, 20-27}, ... -- then touch sixteen distinct 16-byte bank slots), filled by global_load_lds_dwordx4 whose SOURCE address carries the
    permutation.  Tile j + 4 is requested while tile j is multiplied; tile j + 1 is waited for with a COUNTED vmcnt(16) (two younger tiles stay in flight), then ONE
    workgroup barrier per K-tile, issued in the shadow of a matrix instruction.
  * two fragment sets of 32 registers: while the 16 matrix instructions of one 16-deep half run on one set, the 8 ds_read_b128 of the next half fill the other.
  * every K-tile issues its 8 requests, also the last four (clamped to the last tile, into a buffer nobody reads): the counts stay uniform, no tail code.

Register map (the statement's clobbers): a[0:255]; v[192:223] / v[224:255] fragment sets 0 / 1 (A 16 | W^T 16); v[160:163] the ds_read addresses + 64 KiB; v[164:179]
epilogue temporaries; s[68:81].  Operands (names below) are placed by the compiler in the registers it keeps.
"""

import os
import sys
FLAGS = set(a for a in sys.argv[1:] if a != "clobbers")          # diagnostics (tools/r6_gn2_variants.sh): noglds noreads nobar spread -- timing only, wrong results
MF = "v_mfma_f32_32x32x16_f16"
SET_A = (192, 224)
SET_B = (208, 240)
out = []


def emit(s):
    if in_loop and (("noglds" in FLAGS and s.startswith("global_load_lds")) or ("noreads" in FLAGS and s.startswith("ds_read")) or ("nobar" in FLAGS and s == "s_barrier")):
        return
    if in_loop and "noglds" in FLAGS and s.startswith("s_waitcnt vmcnt(16)"):
        s = "s_waitcnt lgkmcnt(0)"
    out.append(s)


in_loop = False


def acc(im, jn):
    t = (im * 4 + jn) * 16
    return f"a[{t}:{t + 15}]"


def frag(base, i):
    return f"v[{base + 4 * i}:{base + 4 * i + 3}]"


def mfma(s, i):
    im, jn = i >> 2, i & 3
    emit(f"{MF} {acc(im, jn)}, {frag(SET_B[s], jn)}, {frag(SET_A[s], im)}, {acc(im, jn)}")


def reads(s, buf, ks):
    """the eight fragment reads of (buffer buf, 16-deep half ks) into set s, in the order the matrix instructions want them: A0 B0 A1 B1 ..."""
    hi = buf >= 2
    imm0 = (buf & 1) * 32768
    ra = (f"v{160 + ks}" if hi else f"%[rdA{ks}]")
    rb = (f"v{162 + ks}" if hi else f"%[rdB{ks}]")
    r = []
    for i in range(4):
        r.append(f"ds_read_b128 {frag(SET_A[s], i)}, {ra} offset:{imm0 + i * 2048}")
        r.append(f"ds_read_b128 {frag(SET_B[s], i)}, {rb} offset:{imm0 + i * 2048}")
    return r


def glds_setup(tile_expr_add, tmp="s74"):
    """scalar part of a K-tile's requests: s[76:77] / s[78:79] = operand bases + min(j + add, nk - 1) * 64"""
    if "src128" in FLAGS:                             # timing only: K-tiles 128 bytes apart, wrapped at k / 2 (8 rows x 128 B per request: full cache lines)
        return [f"s_add_u32 {tmp}, s73, {tile_expr_add}",
                f"s_and_b32 {tmp}, {tmp}, s81",
                f"s_lshl_b32 {tmp}, {tmp}, 7",
                f"s_add_u32 s76, s68, {tmp}",
                "s_addc_u32 s77, s69, 0",
                f"s_add_u32 s78, s70, {tmp}",
                "s_addc_u32 s79, s71, 0"]
    return [f"s_add_u32 {tmp}, s73, {tile_expr_add}",
            f"s_min_u32 {tmp}, {tmp}, s72",
            f"s_lshl_b32 {tmp}, {tmp}, 6",
            f"s_add_u32 s76, s68, {tmp}",
            "s_addc_u32 s77, s69, 0",
            f"s_add_u32 s78, s70, {tmp}",
            "s_addc_u32 s79, s71, 0"]


def glds_list():
    g = []
    for i in range(4):
        g.append(f"global_load_lds_dwordx4 %[goA{i}], s[76:77]")
    for i in range(4):
        g.append(f"global_load_lds_dwordx4 %[goB{i}], s[78:79]")
    return g


# ------------------------------------------------------------------------------------------------------------------ prologue
emit("s_mov_b32 s68, %[alo]"); emit("s_mov_b32 s69, %[ahi]"); emit("s_mov_b32 s70, %[blo]"); emit("s_mov_b32 s71, %[bhi]")
emit("s_sub_u32 s72, %[nk], 1"); emit("s_mov_b32 s73, 0"); emit("s_mov_b32 s80, %[ldsw]")
if "src128" in FLAGS:
    emit("s_lshr_b32 s81, %[nk], 1"); emit("s_sub_u32 s81, s81, 1")
emit("v_add_u32 v160, 0x10000, %[rdA0]"); emit("v_add_u32 v161, 0x10000, %[rdA1]"); emit("v_add_u32 v162, 0x10000, %[rdB0]"); emit("v_add_u32 v163, 0x10000, %[rdB1]")
for t in range(4):
    # tiles 0..3 into buffers 0..3 (nk >= 4)
    emit(f"s_add_u32 s76, s68, {t * 64}"); emit("s_addc_u32 s77, s69, 0"); emit(f"s_add_u32 s78, s70, {t * 64}"); emit("s_addc_u32 s79, s71, 0")
    emit(f"s_add_u32 m0, s80, {t * 32768}")
    emit("s_nop 0")
    for k, g in enumerate(glds_list()):
        emit(g)
        if k < 7:
            emit("s_add_u32 m0, m0, 0x1000"); emit("s_nop 0")
for i in range(256):
    emit(f"v_accvgpr_write_b32 a{i}, 0")
emit("s_waitcnt vmcnt(24)")
emit("s_barrier")
for r in reads(0, 0, 0):
    emit(r)
emit("s_waitcnt lgkmcnt(0)")

# ------------------------------------------------------------------------------------------------------------------ main loop, four K-tiles per trip
in_loop = True
emit("L_gnt2_loop%=:")
# slots 0..15 = the matrix instructions of half 0 (set 0), 16..31 = half 1 (set 1) of tile j = s73 + b; `extras[slot]` follow that slot's matrix instruction.
#   * half 0 carries the fragment reads of half 1 of the same tile (into set 1), half 1 those of half 0 of tile j + 1 (into set 0)
#   * after slot 12: tile j + 1 has landed (this wave's part; vmcnt leaves the younger requests in flight) and this wave's reads of buffer b are complete -> barrier;
#     from then on buffer b may be overwritten: the 8 requests of tile j + 4
GL_SLOTS = [24, 25, 26, 27, 28, 29, 30, 31] if "spread" not in FLAGS else [13, 15, 18, 20, 22, 25, 27, 29]
for b in range(4):
    extras = {i: [] for i in range(32)}
    for i, r in enumerate(reads(1, b, 1)):
        extras[i].append(r)
    for i, r in enumerate(reads(0, (b + 1) & 3, 0)):
        extras[16 + i].append(r)
    sal = glds_setup(b + 4)
    assert len(sal) == 7
    for i, x in enumerate(sal):
        extras[8 + i // 2].append(x)                  # (s_add / s_addc pairs stay in order: nothing else here writes SCC)
    # requests issued before the wait at slot 12 of the NEXT tile's half 0 do not exist (all sit at slots >= 13): vmcnt(16) = two younger tiles
    extras[12] += ["s_waitcnt vmcnt(16) lgkmcnt(0)", "s_barrier"]
    gl = glds_list()
    extras[GL_SLOTS[0] - 1].append(f"s_add_u32 m0, s80, {b * 32768}")
    for k in range(8):
        extras[GL_SLOTS[k]].append(gl[k])
        if k < 7:
            extras[GL_SLOTS[k]].append("s_add_u32 m0, m0, 0x1000")
    extras[31].append("s_waitcnt lgkmcnt(0)")
    for i in range(32):
        mfma(i >> 4, i & 15)
        for x in extras[i]:
            emit(x)
in_loop = False
emit("s_add_u32 s73, s73, 4")
emit("s_cmp_lt_u32 s73, %[nk]")
emit("s_cbranch_scc1 L_gnt2_loop%=")

# ------------------------------------------------------------------------------------------------------------------ accumulators -> fp16 C tile in LDS
emit("s_waitcnt vmcnt(0)")
emit("s_barrier")
emit("s_nop 15"); emit("s_nop 15")
n = 0
for im in range(4):
    for jn in range(4):
        for g in range(4):
            a0 = (im * 4 + jn) * 16 + 4 * g
            t = 164 + 8 * (n & 1)                     # two temporary sets: the reads of one group overlap the store of the previous
            for e in range(4):
                emit(f"v_accvgpr_read_b32 v{t + e}, a{a0 + e}")
            emit(f"v_cvt_pk_f16_f32 v{t + 4}, v{t}, v{t + 1}")
            emit(f"v_cvt_pk_f16_f32 v{t + 5}, v{t + 2}, v{t + 3}")
            emit(f"ds_write_b64 %[ct], v[{t + 4}:{t + 5}] offset:{im * 32 * 528 + (jn * 32 + 8 * g) * 2}")
            n += 1
emit("s_waitcnt lgkmcnt(0)")
emit("s_barrier")

print("// generated by tools/gen_gemm_nt2_loop.py -- do not edit")
if len(sys.argv) > 1 and sys.argv[1] == "clobbers":
    for i in range(0, 256, 16):
        print(", ".join(f'"a{j}"' for j in range(i, i + 16)) + ("," if i < 240 else ""))
else:
    for s in out:
        print(f'"{s}\\n\\t"')
