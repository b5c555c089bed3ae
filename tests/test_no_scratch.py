"""Build-output guard: no kernel of the library may use scratch (private segment) memory.  On MI355X any scratch use costs a launch
1.5-3 us (profiles/r01_launch_chain_microbench.json) -- as much as a whole glue kernel -- so register budgets (`amdgpu_waves_per_eu` per
instantiation) are chosen for zero spills and this test keeps a compiler or code change from silently reintroducing one.
Reads the code objects inside build/obj/*.o (made by __graft_entry__.build(); CPU only)."""
import glob
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

ALLOWED = ()


@pytest.mark.skipif(not glob.glob(os.path.join(ROOT, "build", "obj", "*.o")) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"),
                    reason="needs the objects of __graft_entry__.build() and the ROCm llvm tools")
def test_no_kernel_uses_scratch():
    import scratch_report
    rep = scratch_report.report()
    assert sum(len(v) for v in rep.values()) > 500, "expected the GEMV / GEMM / glue instantiations in build/obj"
    for hot in ("exl3_gemv2_k4.o", "exl3_gemm3_k4.o", "exl3_glue.o", "exl3_rope_cache.o", "exl3_attn_decode.o"):
        assert rep.get(hot), f"{hot}: no kernels found"
    # the opt-in wave-per-column-block modes of the gen-2 GEMV (MODE 5 / 6 / 7, `bench.py --pipeline resid`, measured slower than the default glue
    # pipeline: DESIGN.md 4.4) sit at the SGPR limit; a few K / codebook combinations get a <= 128-byte SGPR-spill frame.  No default route uses them.
    def wpc(name):
        return "exl3_gemv2_kernel" in name and name.endswith(("ELi5EEv8GemvArgs", "ELi6EEv8GemvArgs", "ELi7EEv8GemvArgs"))
    bad = [(obj, name, b) for obj, ks in rep.items() for name, b in ks if b and not any(a in name for a in ALLOWED) and not (wpc(name) and b <= 128)]
    assert not bad, "kernels with scratch: " + ", ".join(f"{o}:{n} ({b} B)" for o, n, b in bad[:10])


@pytest.mark.skipif(not glob.glob(os.path.join(ROOT, "build", "obj", "exl3_pstep_k*.o")) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"),
                    reason="needs the objects of __graft_entry__.build() and the ROCm llvm tools")
def test_persistent_step_instantiations_keep_register_headroom():
    """The persistent step's 16-wave workgroup has 128 registers per lane and must not use scratch (a grid with scratch is not co-resident: DESIGN.md 4.10 Safety).  An
    instantiation that sits AT 128 is one unrelated edit away from a spill (seen at the end of round 5: the mixed-K variants, fixed by two instead of three decode-ahead units
    in the head pass): every instantiation of every per-K unit (exl3_pstep.kspec.hip: mul1 / 3INST / mcg, uniform and two-width plans, with and without the attention) stays
    <= 124 with no VGPR spill."""
    import subprocess, tempfile, re
    LLVM = "/opt/rocm/lib/llvm/bin"
    kernels = {}
    for obj in sorted(glob.glob(os.path.join(ROOT, "build", "obj", "exl3_pstep_k*.o"))):
        with tempfile.TemporaryDirectory() as td:
            fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
            subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"],
                               capture_output=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue                                      # (K = 1, 7: no kernels in the unit)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
        cur = None
        for line in notes.splitlines():
            m = re.match(r"\s*\.name:\s+(\S+)", line)
            if m: cur = m.group(1); kernels[cur] = {}
            m = re.match(r"\s*\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size):\s+(\d+)", line)
            if m and cur: kernels[cur][m.group(1)] = int(m.group(2))
    ps = {k: v for k, v in kernels.items() if "exl3_pstep_kernel" in k}
    assert len(ps) >= 60, sorted(ps)
    bad = {k: v for k, v in ps.items() if v["vgpr_count"] > 124 or v["vgpr_spill_count"] or v["private_segment_fixed_size"]}
    assert not bad, bad
