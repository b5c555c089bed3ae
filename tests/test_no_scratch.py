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
