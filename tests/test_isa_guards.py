"""Compile-time guards on the hot kernels' ISA (CPU suite: hipcc cross-compiles gfx950 without a GPU).

* No kernel of the decode path may use scratch: a private segment costs 1.5 .. 3 us per launch on MI355X (profiles/r01_launch_chain_microbench.json),
  more than the work of a batch-1 launch.  Checked for every instantiation of the 4.0 bpw translation units of generations 3 and 4 (the benchmarked
  bitrate; tools/check_spills.sh runs the same check over K = 1..8).
* The one-chunk streaming loop of the 16-row generation-3 kernel keeps COUNTED waits (vmcnt(3) / (2), never vmcnt(0)): the round-3 loop form drained
  every outstanding load at the loop head."""
import os
import re
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-DG2_K=4", "-S",
         "--cuda-device-only"]


def _asm(tmp_path, src):
    out = tmp_path / (src + ".s")
    r = subprocess.run([HIPCC] + FLAGS + ["-o", str(out), os.path.join(ROOT, "exllamav3_amd", "csrc", src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def _kernels(asm):
    for b in asm.split("  - .agpr_count:")[1:]:
        yield (re.search(r"\.name:\s+(\S+)", b).group(1), int(re.search(r"\.vgpr_spill_count:\s+(\d+)", b).group(1)),
               int(re.search(r"\.sgpr_spill_count:\s+(\d+)", b).group(1)), int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1)))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src,minimum", [("exl3_gemm3.kspec.hip", 30), ("exl3_gemv4.kspec.hip", 40)])
def test_no_kernel_uses_scratch_at_4bpw(tmp_path, src, minimum):
    ks = list(_kernels(_asm(tmp_path, src)))
    assert len(ks) >= minimum
    # (name, VGPR spills, SGPR spills, private segment bytes).  SGPR spills live in the lanes of a spare VGPR (v_writelane / v_readlane: no memory, no private
    # segment): tolerated only in the attention-merge and q|k|v-epilogue input modes of generation 4 (template argument MODE = 8 / 9), whose task code holds
    # the whole argument block plus the partial-record addressing; everything else must not spill at all
    bad = [k for k in ks if k[1] or k[3] or (k[2] and not k[0].endswith(("ELi8EEv8GemvArgs", "ELi9EEv8GemvArgs")))]
    assert not bad, bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_gemm3_one_chunk_loop_has_counted_waits(tmp_path):
    asm = _asm(tmp_path, "exl3_gemm3.kspec.hip")
    fn = "_Z17exl3_gemm3_kernelILi4ELi2ELi1ELb1ELb1ELi8EEv8GemvArgs"           # K = 4, mul1, 16 rows, rotated input, raw variant, 8 waves: the decode step's launches
    body = asm[asm.index(fn + ":"):]
    body = body[:body.index(".Lfunc_end")]
    # basic blocks that branch back to themselves and hold the 16 matrix instructions of a 4-step trip
    loops = []
    for m in re.finditer(r"^(\.LBB\d+_\d+):.*?(?=^\.LBB\d+_\d+:|\Z)", body, flags=re.S | re.M):
        blk = m.group(0)
        if blk.count("v_mfma_f32_16x16x32") == 16 and re.search(r"s_cbranch_\w+ " + re.escape(m.group(1)) + r"\b", blk):
            loops.append(blk)
    assert loops, "streaming loop not found"
    one_chunk = loops[0]
    waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", one_chunk)
    assert waits and "0" not in waits, waits
