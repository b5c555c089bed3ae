#!/usr/bin/env python3
"""Generator of the hand-scheduled K-loop of the prefill NT GEMM, 64-deep K-tiles and v_mfma_f32_16x16x32_f16 (exllamav3_amd/csrc/exl3_gemm_nt2.hip, -DGN2_K64 body):

    python tools/gen_gemm_nt3_loop.py > exllamav3_amd/csrc/exl3_gemm_nt3_loop.inc

Same skeleton as tools/gen_gemm_nt2_loop.py (one wave per SIMD, 128 x 128 wave tile, 256 accumulator registers, every register named by hand), with the two things the
first version's measurements asked for (profiles/NOTES.md R6.3): operand rows of 128 bytes -- one request = 8 full cache lines instead of 16 halves (half the L2 requests,
half the issue time of a request) -- and the 16x16x32 matrix instruction, which the chip sustains at a higher clock on random operands than 32x32x16.

  * 256 x 256 x 64 tiles in TWO LDS stages of 64 KiB (A 32 KiB | W^T 32 KiB; 128-byte rows, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7): generation 1's map,
    measured conflict-free).  16 requests per wave and K-tile.
  * accumulators: tile (im, jn) of 16 x 16 = a[4 (8 im + jn) : + 3]; W^T fragment first, activation fragment second: lane l holds row m = l & 15 and four consecutive columns.
  * two fragment sets of 64 registers (8 A + 8 W^T fragments of one 32-deep half).  Per K-tile t (stage t & 1), slots = the 128 matrix instructions:
        0..15    reads of half 1 of tile t -> set 1                      (set 0 multiplies: slots 0..63)
        20       vmcnt(0) [tile t + 1 landed] lgkmcnt(0) [stage t & 1 read]  ->  barrier
        21..     the 16 requests of tile t + 2 into stage t & 1, every second slot
        64..79   reads of half 0 of tile t + 1 -> set 0                    (set 1 multiplies: slots 64..127)
  * every K-tile issues its 16 requests, the last two clamped to the last tile (uniform counts, no tail code).

Register map (clobbers): a[0:255]; v[128:191] / v[192:255] fragment sets 0 / 1 (A 32 | W^T 32); v[120:123] read addresses + 64 KiB; v[104:119] epilogue temporaries; s[68:81].
Flags (diagnostics, timing only): noglds noreads nobar
"""
import sys

FLAGS = set(a for a in sys.argv[1:] if a != "clobbers")
# "n128": the 256 x 128 tile (wave tile 128 x 64, four W^T fragments per half, a[0:127]) in THREE stages of 48 KiB, requests three K-tiles ahead with a counted vmcnt --
# for shapes whose 256 x 256 tile count leaves CUs idle in the last round (q|k|v of Llama-3.1-8B: 384 tiles on 256 CUs; 768 narrow tiles = three full rounds)
NB = 4 if "n128" in FLAGS else 8                      # W^T fragments of a wave per 32-deep half
S = 3 if NB == 4 else 2                               # LDS stages
STAGE = [0, 65536] if S == 2 else [0, 49152, 98304]
NG = 8 + NB                                           # requests / fragment reads per wave and K-tile / half
NS = 16 * NB                                          # matrix instructions per wave and K-tile
RP = 2 if "rp2" in FLAGS else 1                       # fragment reads per slot (two per slot with all four waves in step saturate the LDS port: +12 % cycles, measured)
MF = "v_mfma_f32_16x16x32_f16"
SET_A = (128, 192)
SET_B = (160, 224)
out = []
in_loop = False


def emit(s):
    if in_loop and (("noglds" in FLAGS and s.startswith("global_load_lds")) or ("noreads" in FLAGS and s.startswith("ds_read")) or ("nobar" in FLAGS and s == "s_barrier")):
        return
    if in_loop and "noglds" in FLAGS and s.startswith("s_waitcnt vmcnt(") and s.endswith("lgkmcnt(0)"):
        s = "s_waitcnt lgkmcnt(0)"
    out.append(s)


def acc(im, jn):
    t = (im * NB + jn) * 4
    return f"a[{t}:{t + 3}]"


def frag(base, i):
    return f"v[{base + 4 * i}:{base + 4 * i + 3}]"


def mfma(s, i):
    im, jn = i // NB, i % NB
    if (im & 1) and "noserp" not in FLAGS:
        jn = NB - 1 - jn                                  # serpentine: at a change of the A fragment the W^T fragment stays (one operand toggles, not two)
    emit(f"{MF} {acc(im, jn)}, {frag(SET_B[s], jn)}, {frag(SET_A[s], im)}, {acc(im, jn)}")


def reads(s, stage, ks):
    """the sixteen fragment reads of (stage, 32-deep half ks) into set s: A0 B0 B1 A1 B2 B3 ... the order the matrix instructions first touch them (im = 0 runs over all jn)"""
    hi = STAGE[stage] >= 65536
    imm0 = STAGE[stage] - (65536 if hi else 0)
    ra = (f"v{120 + ks}" if hi else f"%[rdA{ks}]")
    rb = (f"v{122 + ks}" if hi else f"%[rdB{ks}]")
    a = [f"ds_read_b128 {frag(SET_A[s], i)}, {ra} offset:{imm0 + i * 2048}" for i in range(8)]
    b = [f"ds_read_b128 {frag(SET_B[s], i)}, {rb} offset:{imm0 + i * 2048}" for i in range(NB)]
    return [a[0]] + b + a[1:]


def glds_setup(add, tmp="s74"):
    return [f"s_add_u32 {tmp}, s73, {add}",
            f"s_min_u32 {tmp}, {tmp}, s72",
            f"s_lshl_b32 {tmp}, {tmp}, 7",
            f"s_add_u32 s76, s68, {tmp}",
            "s_addc_u32 s77, s69, 0",
            f"s_add_u32 s78, s70, {tmp}",
            "s_addc_u32 s79, s71, 0"]


def glds_list():
    return [f"global_load_lds_dwordx4 %[goA{i}], s[76:77]" for i in range(8)] + [f"global_load_lds_dwordx4 %[goB{i}], s[78:79]" for i in range(NB)]


# ------------------------------------------------------------------------------------------------------------------ prologue
emit("s_mov_b32 s68, %[alo]"); emit("s_mov_b32 s69, %[ahi]"); emit("s_mov_b32 s70, %[blo]"); emit("s_mov_b32 s71, %[bhi]")
emit("s_sub_u32 s72, %[nk], 1"); emit("s_mov_b32 s73, 0"); emit("s_mov_b32 s80, %[ldsw]")
emit("v_add_u32 v120, 0x10000, %[rdA0]"); emit("v_add_u32 v121, 0x10000, %[rdA1]"); emit("v_add_u32 v122, 0x10000, %[rdB0]"); emit("v_add_u32 v123, 0x10000, %[rdB1]")
A3 = "a3" in FLAGS
if A3:
    # ---- "a3": THREE stages for the activations (0 / 32 / 64 KiB), two for W^T (96 / 128 KiB) = all 160 KiB of the CU.  After barrier(t) a wave requests W^T(t + 2), then
    # A(t + 3): the requests that are needed first are issued first, the others get a whole K-tile more of flight time (operands that come from HBM rather than the
    # 256 MB cache -- the down projection inside a prefill chunk -- arrived late with two stages each: +10 % there, profiles/NOTES.md R6.3).  Six K-tiles per trip.
    assert NB == 8
    STA = [0, 32768, 65536]; STB = [0, 32768]          # (W^T stages relative to 96 KiB: %[rdB*] and the request base carry the 96 KiB)

    def reads3(s_, ta, tb, ks):
        sa, sb = ta % 3, tb % 2
        ra = f"v{120 + ks}" if sa == 2 else f"%[rdA{ks}]"
        ia = 0 if sa == 2 else STA[sa]
        a = [f"ds_read_b128 {frag(SET_A[s_], i)}, {ra} offset:{ia + i * 2048}" for i in range(8)]
        b = [f"ds_read_b128 {frag(SET_B[s_], i)}, %[rdB{ks}] offset:{STB[sb] + i * 2048}" for i in range(8)]
        return [a[0]] + b + a[1:]

    def setup3(addB, addA):
        return [f"s_add_u32 s74, s73, {addB}", "s_min_u32 s74, s74, s72", "s_lshl_b32 s74, s74, 7", "s_add_u32 s78, s70, s74", "s_addc_u32 s79, s71, 0",
                f"s_add_u32 s75, s73, {addA}", "s_min_u32 s75, s75, s72", "s_lshl_b32 s75, s75, 7", "s_add_u32 s76, s68, s75", "s_addc_u32 s77, s69, 0"]

    gA = [f"global_load_lds_dwordx4 %[goA{i}], s[76:77]" for i in range(8)]
    gB = [f"global_load_lds_dwordx4 %[goB{i}], s[78:79]" for i in range(8)]

    def pro(kind, t):
        emit(f"s_min_u32 s74, {t}, s72"); emit("s_lshl_b32 s74, s74, 7")
        if kind == "A":
            emit("s_add_u32 s76, s68, s74"); emit("s_addc_u32 s77, s69, 0"); emit(f"s_add_u32 m0, s80, {STA[t % 3]}")
        else:
            emit("s_add_u32 s78, s70, s74"); emit("s_addc_u32 s79, s71, 0"); emit(f"s_add_u32 m0, s80, {98304 + STB[t % 2]}")
        emit("s_nop 0")
        for k, g in enumerate(gA if kind == "A" else gB):
            emit(g)
            if k < 7:
                emit("s_add_u32 m0, m0, 0x1000"); emit("s_nop 0")

    pro("A", 0); pro("B", 0); pro("A", 1); pro("B", 1); pro("A", 2)
    for i in range(256):
        emit(f"v_accvgpr_write_b32 a{i}, 0")
    emit("s_waitcnt vmcnt(24)")
    emit("s_barrier")
    for r in reads3(0, 0, 0, 0):
        emit(r)
    emit("s_waitcnt lgkmcnt(0)")
    in_loop = True
    for w in range(1, 4):
        emit(f"s_cmp_eq_u32 %[wid], {w}")
        emit(f"s_cbranch_scc1 L_gnt3_w{w}%=")
    for w in range(4):
        emit(f"L_gnt3_w{w}%=:")
        for tau in range(6):
            extras = {i: [] for i in range(128)}
            for i, r in enumerate(reads3(1, tau, tau, 1)):
                extras[i].append(r)
            for i, r in enumerate(reads3(0, tau + 1, tau + 1, 0)):
                extras[64 + i].append(r)
            sal = setup3(2, 3)
            for i, x in enumerate(sal):
                extras[16 + [0, 0, 0, 1, 1, 2, 2, 2, 3, 3][i]].append(x)
            extras[20] += ["s_waitcnt vmcnt(8) lgkmcnt(0)", "s_barrier", f"s_add_u32 m0, s80, {98304 + STB[tau % 2]}"]
            gl = gB + gA
            for k in range(16):
                slot = 21 + 4 * k + w
                extras[slot].append(gl[k])
                if k == 7:
                    extras[slot].append(f"s_add_u32 m0, s80, {STA[tau % 3]}")
                elif k < 15:
                    extras[slot].append("s_add_u32 m0, m0, 0x1000")
            extras[127] += ["s_waitcnt lgkmcnt(0)", "s_add_u32 s73, s73, 1", "s_cmp_lt_u32 s73, %[nk]"]
            for i in range(128):
                mfma(i >> 6, i & 63)
                for x in extras[i]:
                    emit(x)
            if tau < 5:
                emit("s_cbranch_scc0 L_gnt3_done%=")
        emit(f"s_cbranch_scc1 L_gnt3_w{w}%=")
        if w < 3:
            emit("s_branch L_gnt3_done%=")
    emit("L_gnt3_done%=:")
    in_loop = False
else:
    for t in range(S):
        # tiles 0 .. S - 1 into stages 0 .. S - 1 (clamped: nk may be smaller than S)
        emit(f"s_min_u32 s74, {t}, s72"); emit("s_lshl_b32 s74, s74, 7")
        emit("s_add_u32 s76, s68, s74"); emit("s_addc_u32 s77, s69, 0"); emit("s_add_u32 s78, s70, s74"); emit("s_addc_u32 s79, s71, 0")
        emit(f"s_add_u32 m0, s80, {STAGE[t]}")
        emit("s_nop 0")
        for k, g in enumerate(glds_list()):
            emit(g)
            if k < NG - 1:
                emit("s_add_u32 m0, m0, 0x1000"); emit("s_nop 0")
    for i in range(8 * NB * 4):
        emit(f"v_accvgpr_write_b32 a{i}, 0")
    emit(f"s_waitcnt vmcnt({(S - 1) * NG})")
    emit("s_barrier")
    for r in reads(0, 0, 0):
        emit(r)
    emit("s_waitcnt lgkmcnt(0)")

    # ------------------------------------------------------------------------------------------------------------------ main loop, S K-tiles per trip
    in_loop = True
    # The four waves of the workgroup run FOUR copies of the loop that differ only in the slots of their requests: wave w issues request k after matrix instruction
    # GL0 + 4 k + w, so the CU's one address unit sees one request per slot instead of four at once (measured: a request issued by all four waves in the same slot
    # stalls each of them ~26 cycles beyond the matrix instruction it hides behind -- profiles/NOTES.md R6.3).  "nostagger": one copy, all waves in the same slots.
    # Per K-tile t (stage t % S), slots = its NS matrix instructions:
    #     0 ..           the NG reads of half 1 of tile t -> set 1 (RP per slot)                 (set 0 multiplies: slots 0 .. NS / 2 - 1)
    #     BS             vmcnt((S - 2) NG) [tile t + 1 landed] lgkmcnt(0) [stage t % S read] -> barrier
    #     BS + 1 ..      the NG requests of tile t + S into stage t % S
    #     NS / 2 ..      the NG reads of half 0 of tile t + 1 -> set 0                            (set 1 multiplies: slots NS / 2 .. NS - 1)
    STAGGER = "nostagger" not in FLAGS
    NW = 4 if STAGGER else 1
    RSL = NG // RP                                        # slots that carry fragment reads
    SAL0 = RSL if NB == 8 else RSL - 4                    # the scalar part of the requests: after the reads, or (narrow tile: 64 slots) beside the last four
    BS = SAL0 + 4
    GL0, GL_STEP = (BS + 1, 4) if STAGGER else (BS + 1, 2)
    for f in FLAGS:
        if f.startswith("glstep="):
            GL_STEP = int(f[7:])
    assert GL0 + GL_STEP * (NG - 1) + (NW - 1 if GL_STEP >= 4 else GL_STEP - 1) < NS
    if STAGGER:
        for w in range(1, 4):
            emit(f"s_cmp_eq_u32 %[wid], {w}")
            emit(f"s_cbranch_scc1 L_gnt3_w{w}%=")
    for w in range(NW):
        emit(f"L_gnt3_w{w}%=:")
        for st in range(S):
            extras = {i: [] for i in range(NS)}
            for i, r in enumerate(reads(1, st, 1)):
                extras[i // RP].append(r)
            for i, r in enumerate(reads(0, (st + 1) % S, 0)):
                extras[NS // 2 + i // RP].append(r)
            sal = glds_setup(S)
            for i, x in enumerate(sal):
                extras[SAL0 + i // 2].append(x)
            extras[BS] += [f"s_waitcnt vmcnt({(S - 2) * NG}) lgkmcnt(0)", "s_barrier", f"s_add_u32 m0, s80, {STAGE[st]}"]
            gl = glds_list()
            for k in range(NG):
                slot = GL0 + GL_STEP * k + (w if GL_STEP >= 4 else w % GL_STEP if GL_STEP > 1 else 0)
                extras[slot].append(gl[k])
                if k < NG - 1:
                    extras[slot].append("s_add_u32 m0, m0, 0x1000")
            extras[NS - 1] += ["s_waitcnt lgkmcnt(0)", "s_add_u32 s73, s73, 1", "s_cmp_lt_u32 s73, %[nk]"]
            for i in range(NS):
                mfma(i // (NS // 2), i % (NS // 2))
                for x in extras[i]:
                    emit(x)
            if st < S - 1:
                emit("s_cbranch_scc0 L_gnt3_done%=")
        emit(f"s_cbranch_scc1 L_gnt3_w{w}%=")
        if STAGGER and w < 3:
            emit("s_branch L_gnt3_done%=")
    emit("L_gnt3_done%=:")
    in_loop = False

# ------------------------------------------------------------------------------------------------------------------ accumulators -> C tile in LDS
# "noend": the K-loop statement stops behind its last matrix instruction (the grouped kernel picks its ending at run time: exl3_gemm_nt2_body.inc, GN2_SPLIT_END);
# "end_f16" / "end_f32_p0" / "end_f32_p1": only that ending, as its own statement -- the fp16 C tile, or one 128-row half of the tile as fp32 ([128][260] floats:
# the rows of accumulator tiles im = 4 p .. 4 p + 3 of both wave rows).
END_ONLY = [f for f in FLAGS if f.startswith("end_")]
if END_ONLY:
    out = []
if "noend" in FLAGS or END_ONLY:
    if not END_ONLY:
        emit("s_waitcnt vmcnt(0)")
        emit("s_barrier")
        emit("s_nop 15")
else:
    emit("s_waitcnt vmcnt(0)")
    emit("s_barrier")
    emit("s_nop 15")
if "noend" not in FLAGS and (not END_ONLY or END_ONLY[0] == "end_f16"):
    n = 0
    for im in range(8):
        for jn in range(NB):
            a0 = (im * NB + jn) * 4
            t = 104 + 8 * (n & 1)
            for e in range(4):
                emit(f"v_accvgpr_read_b32 v{t + e}, a{a0 + e}")
            emit(f"v_cvt_pk_f16_f32 v{t + 4}, v{t}, v{t + 1}")
            emit(f"v_cvt_pk_f16_f32 v{t + 5}, v{t + 2}, v{t + 3}")
            emit(f"ds_write_b64 %[ct], v[{t + 4}:{t + 5}] offset:{im * 16 * 528 + jn * 32}")
            n += 1
    emit("s_waitcnt lgkmcnt(0)")
    emit("s_barrier")
elif END_ONLY:
    p_ = int(END_ONLY[0][-1])
    n = 0
    for im in range(4 * p_, 4 * p_ + 4):
        for jn in range(NB):
            a0 = (im * NB + jn) * 4
            t = 104 + 4 * (n & 3)
            for e in range(4):
                emit(f"v_accvgpr_read_b32 v{t + e}, a{a0 + e}")
            emit(f"ds_write_b128 %[ct], v[{t}:{t + 3}] offset:{(im - 4 * p_) * 16 * 1040 + jn * 64}")
            n += 1
    emit("s_waitcnt lgkmcnt(0)")

print("// generated by tools/gen_gemm_nt3_loop.py -- do not edit")
for s in out:
    print(f'"{s}\\n\\t"')
