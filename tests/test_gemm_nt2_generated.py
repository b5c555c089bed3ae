"""Host-side checks of the generated assembly K-loops of the prefill GEMM (exl3_gemm_nt2.hip; tools/gen_gemm_nt2_loop.py, tools/gen_gemm_nt3_loop.py) -- no GPU:
the committed *.inc files ARE what the generators emit; every register a statement names lies inside the clobber list of its asm statement (exl3_gemm_nt2_body.inc);
every LDS-DMA request has exactly one fresh M0 with another instruction between the M0 write and the request (the hazard the assembler does not pad); the counted
waits are the counts the schedule implies; immediates fit their fields; and the LDS images are bank-conflict-free under the lane groups of the ds_read_b128 hardware path
(/opt/skills/guides/MI355X_MICROARCH.md, LDS table)."""
import os
import re
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "exllamav3_amd", "csrc")
FORMS = {"exl3_gemm_nt2_loop.inc": ("gen_gemm_nt2_loop.py", []), "exl3_gemm_nt3_loop.inc": ("gen_gemm_nt3_loop.py", []),
         "exl3_gemm_nt3n_loop.inc": ("gen_gemm_nt3_loop.py", ["n128"]), "exl3_gemm_nt3a_loop.inc": ("gen_gemm_nt3_loop.py", ["a3"]),
         "exl3_gemm_nt2_clobbers.inc": ("gen_gemm_nt2_loop.py", ["clobbers"]),
         # the grouped kernel: the K-loop without an ending + the endings as statements of their own (fp16 C tile, two fp32 half tiles)
         "exl3_gemm_nt3g_loop.inc": ("gen_gemm_nt3_loop.py", ["a3", "noend"]), "exl3_gemm_nt3_end_f16.inc": ("gen_gemm_nt3_loop.py", ["end_f16"]),
         "exl3_gemm_nt3_end_f32_p0.inc": ("gen_gemm_nt3_loop.py", ["end_f32_p0"]), "exl3_gemm_nt3_end_f32_p1.inc": ("gen_gemm_nt3_loop.py", ["end_f32_p1"])}
# requests per K-tile and wave, matrix instructions per K-tile and wave, K-tiles per loop trip, loop copies, the counted wait inside the loop
SHAPE = {"exl3_gemm_nt2_loop.inc": (8, 32, 4, 1, 16), "exl3_gemm_nt3_loop.inc": (16, 128, 2, 4, 0), "exl3_gemm_nt3n_loop.inc": (12, 64, 3, 4, 12),
         "exl3_gemm_nt3a_loop.inc": (16, 128, 6, 4, 8), "exl3_gemm_nt3g_loop.inc": (16, 128, 6, 4, 8)}


def _gen(script, args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + args, check=True, capture_output=True, text=True).stdout


def _lines(text):
    return [m.group(1) for m in re.finditer(r'^"(.*)\\n\\t"$', text, flags=re.M)]


@pytest.mark.parametrize("name", list(FORMS))
def test_committed_loop_files_are_the_generators_output(name):
    script, args = FORMS[name]
    assert open(os.path.join(CSRC, name)).read() == _gen(script, args), f"{name} is stale: python tools/{script} {' '.join(args)} > exllamav3_amd/csrc/{name}"


@pytest.mark.parametrize("name", list(SHAPE))
def test_registers_named_by_the_statement_are_clobbered(name):
    body = open(os.path.join(CSRC, "exl3_gemm_nt2_body.inc")).read()
    k64 = name != "exl3_gemm_nt2_loop.inc"
    part = body[body.index("#ifdef GN2_K64"):body.index("#else\n    // ======== 32-deep")] if k64 else body[body.index("// ======== 32-deep"):]
    clob = part[part.index('"memory", "scc"'):]
    clob = clob[:clob.index(");")]
    vs = {int(x) for x in re.findall(r'"v(\d+)"', clob)}; ss = {int(x) for x in re.findall(r'"s(\d+)"', clob)}
    text = "\n".join(_lines(open(os.path.join(CSRC, name)).read()))
    used_v, used_s, used_a = set(), set(), set()
    for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", text): used_v |= set(range(int(lo), int(hi) + 1))
    used_v |= {int(x) for x in re.findall(r"\bv(\d+)\b", text)}
    for lo, hi in re.findall(r"\bs\[(\d+):(\d+)\]", text): used_s |= set(range(int(lo), int(hi) + 1))
    used_s |= {int(x) for x in re.findall(r"\bs(\d+)\b", text)}
    for lo, hi in re.findall(r"\ba\[(\d+):(\d+)\]", text): used_a |= set(range(int(lo), int(hi) + 1))
    used_a |= {int(x) for x in re.findall(r"\ba(\d+)\b", text)}
    assert used_v and used_v <= vs, sorted(used_v - vs)
    assert used_s and used_s <= ss, sorted(used_s - ss)
    assert used_a and max(used_a) <= 255                                   # (a0 .. a255: exl3_gemm_nt2_clobbers.inc)
    assert len(re.findall(r'"a\d+"', open(os.path.join(CSRC, "exl3_gemm_nt2_clobbers.inc")).read())) == 256


@pytest.mark.parametrize("name", list(SHAPE))
def test_schedule_counts_and_m0_discipline(name):
    ng, nmf, tiles, copies, vm = SHAPE[name]
    ls = _lines(open(os.path.join(CSRC, name)).read())
    start = next(i for i, l in enumerate(ls) if re.match(r"L_gnt\d_(loop|w0)%=:", l))
    end = max(i for i, l in enumerate(ls) if l.startswith("s_cbranch_scc1 L_gnt") and ("loop" in l or "_w" in l)) + 1
    loop = ls[start:end]
    assert sum(l.startswith("v_mfma") for l in loop) == nmf * tiles * copies
    assert sum(l.startswith("global_load_lds_dwordx4") for l in loop) == ng * tiles * copies
    assert sum(l == "s_barrier" for l in loop) == tiles * copies                                  # ONE workgroup barrier per K-tile
    waits = [l for l in loop if l.startswith("s_waitcnt vmcnt")]
    assert len(waits) == tiles * copies and all(l == f"s_waitcnt vmcnt({vm}) lgkmcnt(0)" for l in waits), set(waits)
    # every request: exactly one M0 write since the previous request, and at least one instruction between that write and the request
    last_m0, since_req_m0 = None, 0
    for i, l in enumerate(ls):
        if re.match(r"s_(add_u32|mov_b32) m0,", l):
            last_m0 = i; since_req_m0 += 1
        elif l.startswith("global_load_lds_dwordx4"):
            assert since_req_m0 == 1 and last_m0 is not None and i - last_m0 >= 2, (i, l)
            since_req_m0 = 0
    # immediates: 16-bit unsigned DS offsets
    for l in ls:
        m = re.search(r"^ds_(read|write)\w+ .*offset:(\d+)$", l)
        if m: assert int(m.group(2)) < 65536, l
    # s_addc_u32 follows its s_add_u32 with nothing that writes SCC in between
    for i, l in enumerate(ls):
        if l.startswith("s_addc_u32"):
            j = i - 1
            while not ls[j].startswith("s_"): j -= 1
            assert ls[j].startswith("s_add_u32") and not ls[j].startswith("s_add_u32 m0"), (i, ls[j], l)


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def _slots(addrs):
    return [(a // 16) % 16 for a in addrs]                                                      # a 16-byte access covers 4 of the 64 banks: 16 slots


def test_fragment_reads_are_bank_conflict_free_128_byte_rows():
    """64-deep forms: v_mfma_f32_16x16x32 fragment = 16 rows x four 16-byte chunks; lane l reads row l & 15, chunk 4 ks + (l >> 4) at position chunk ^ ((row >> 1) & 7)."""
    for ks in range(2):
        for grp in B128_GROUPS:
            s = _slots([(l & 15) * 128 + (((4 * ks + (l >> 4)) ^ (((l & 15) >> 1) & 7)) << 4) for l in grp])
            assert sorted(s) == list(range(16)), (ks, grp, s)


def test_fragment_reads_are_bank_conflict_free_64_byte_rows():
    """32-deep form: v_mfma_f32_32x32x16 fragment = 32 rows x two chunks; lane l reads row l & 31, chunk 2 ks + (l >> 5) at position chunk ^ ((row >> 2) & 3)."""
    for ks in range(2):
        for grp in B128_GROUPS:
            s = _slots([(l & 31) * 64 + (((2 * ks + (l >> 5)) ^ (((l & 31) >> 2) & 3)) << 4) for l in grp])
            assert sorted(s) == list(range(16)), (ks, grp, s)


@pytest.mark.parametrize("rows_per_req,chunks", [(8, 8), (16, 4)])
def test_request_source_permutation_matches_the_read_side(rows_per_req, chunks):
    """A request's lane L lands at LDS position L % chunks of row L / chunks (lane-linear 1 KiB) and fetches global chunk position ^ f(row): the reader asks for chunk c at
    position c ^ f(row) -- the same involution, so reads see chunk c."""
    f = (lambda r: (r >> 1) & 7) if chunks == 8 else (lambda r: (r >> 2) & 3)
    for base in range(0, 64, rows_per_req):
        for L in range(64):
            row, pos = base + L // chunks, L % chunks
            fetched = pos ^ f(row)
            assert fetched ^ f(row) == pos and 0 <= fetched < chunks


def test_no_compiler_access_to_accumulation_registers_outside_the_statements():
    """The grouped kernel keeps a[0:255] live ACROSS asm statements (K-loop, then the ending chosen at run time with C++ in between): the compiler must not touch an
    accumulation register or spill there.  Compiles the unit to assembly (hipcc cross-compiles without a GPU) and looks outside ;;#ASMSTART / ;;#ASMEND."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "g.s")
        r = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                            "-S", "--cuda-device-only", os.path.join(CSRC, "exl3_gemm_nt2.hip"), "-o", out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        text = open(out).read()
    kernels = re.findall(r"^(_Z\d+exl3_gemm_nt\w+):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) >= 5
    for name, body in kernels:
        outside, inasm = [], False
        for l in body.splitlines():
            if ";;#ASMSTART" in l: inasm = True; continue
            if ";;#ASMEND" in l: inasm = False; continue
            if not inasm: outside.append(l)
        bad = [l for l in outside if re.search(r"accvgpr|scratch_|[\s,\[]a\d+[\],:\s]|\ba\[\d+:\d+\]", l)]
        assert not bad, (name, bad[:5])
