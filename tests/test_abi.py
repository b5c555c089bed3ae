"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU and exports every symbol that
include/exl3_hip.h declares; argument errors surface as RuntimeError like the reference's TORCH_CHECK."""
import os
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from exllamav3_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build(verbose=False)
    syms = _lib.check_symbols()
    assert len(syms) >= 20
    assert _lib.lib().exl3_abi_version() == _lib.ABI_VERSION == 4


def test_argument_errors_are_reported_without_gpu():
    from exllamav3_amd import _lib
    l = _lib.lib()
    # K out of range is rejected before any device work
    rc = l.exl3_unpack_trellis(1, 1, 1, 1, 9, None)
    assert rc == -1 and b"K must be in [1, 8]" in l.exl3_last_error()
    rc = l.exl3_reconstruct(1, 1, 1, 8, 4, 0, 64, 128, None)
    assert rc == -1 and b"n_offset must be divisible by 128" in l.exl3_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc)


def test_header_cites_reference():
    src = open(os.path.join(ROOT, "include", "exl3_hip.h")).read()
    for needle in ("quant/exl3_gemm.cuh", "quant/reconstruct.cu", "norm.cu", "rope.cu", "q_cache"):
        assert needle in src


def test_cpp_consumers_of_the_header_still_compile():
    """The Python-free hosts under tools/experiments (whole decode step, prefill chunk, MLP block: C++ programs that call the library through
    include/exl3_hip.h only) are type-checked against the header, so a changed signature cannot silently break the drop-in boundary for a C / C++ consumer
    (INTEGRATION.md section 5).  Syntax check only: no GPU, no link."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not available")
    exp = os.path.join(ROOT, "tools", "experiments")
    srcs = [os.path.join(exp, f) for f in ("decode_step_harness.hip", "prefill_chunk_harness.hip", "mlp1_harness.hip", "exl3_mlp1.hip")]
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-DM1_TAGGED", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "exllamav3_amd", "csrc")] + srcs, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
