"""The drop-in boundary demonstrated with the reference's OWN Python (CPU, needs /root/reference; skipped elsewhere): the CPython stub
exllamav3_amd/stub/exllamav3_ext*.so is found by the import seam of exllamav3/ext.py:20-30 (find_spec + EXTENSION_SUFFIXES), the whole
reference package imports over it unchanged, and every `ext.<name>` the reference's hot-path modules use resolves on it."""
import ast
import glob
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
STUB_DIR = os.path.join(ROOT, "exllamav3_amd", "stub")

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "exllamav3")) or not glob.glob(os.path.join(STUB_DIR, "exllamav3_ext*.so")),
                               reason="needs /root/reference and the stub built by __graft_entry__.build()")

# the callers on the hot path
HOT_MODULES = ["modules/quant/exl3.py", "modules/rmsnorm.py", "util/rope.py", "cache/quant.py", "modules/linear.py",
               # round 4: the modules that own the runners -- attention (BC_Attention, gates, sinks), the dense and sparse MLPs (BC_GatedMLP, BC_MLP,
               # BC_BlockSparseMLP, exl3_moe), the pointer-table holder, the fp16 cache layer
               "modules/attn.py", "modules/mlp.py", "modules/block_sparse_mlp.py", "modules/multilinear.py", "cache/fp16.py", "modules/quant/fp16.py",
               "modules/transformer.py", "modules/embedding.py", "modules/layernorm.py"]
# names those modules use that belong to other subsystems (conversion-time quantizer, LoRA-free fp16 inner, capture) -- outside SURVEY.md 8
OUTSIDE = {"quantize_tiles", "quantize_tiles_multigpu", "had_paley", "had_paley2", "test_distribution", "count_inf_nan", "gated_rms_norm",
           "quantize_error", "gen_mrope_pos_ids",
           # other model families' routers / activations (DeepSeek-V3 and selective-norm routing, xIELU): outside SURVEY.md 8
           "routing_ds3_nogroup", "routing_sel_norm", "xielu"}       # gen_mrope_pos_ids: host-side position-id builder of the multimodal front end (no device work)


def _ext_names(path):
    tree = ast.parse(open(path).read())
    return {n.attr for n in ast.walk(tree) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == "ext"}


@needs_ref
def test_reference_package_imports_over_the_stub():
    code = ("import exllamav3_ext, exllamav3, exllamav3.ext as seam\n"
            "assert seam.is_precompiled_extension_available()\n"
            "assert seam.exllamav3_ext is exllamav3_ext\n"
            "assert exllamav3_ext.__implementation__.__name__ == 'exllamav3_amd.ext'\n"
            "assert exllamav3_ext.__file__.endswith('.so')\n"
            "from exllamav3.modules.quant.exl3 import LinearEXL3\n"
            "from exllamav3.modules.rmsnorm import RMSNorm\n"
            "from exllamav3.util.rope import RoPE\n"
            "from exllamav3.cache.quant import CacheLayer_quant\n"
            "try:\n    exllamav3_ext.argmax_sample\nexcept AttributeError as e:\n    assert 'outside the EXL3' in str(e)\nelse:\n    raise SystemExit('sampler op unexpectedly present')\n"
            "print('SEAM_OK')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([STUB_DIR, ROOT, REF]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
    assert r.returncode == 0 and "SEAM_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


@needs_ref
def test_every_hot_path_ext_name_resolves():
    sys.path.insert(0, ROOT)
    from exllamav3_amd import ext
    missing = {}
    for m in HOT_MODULES:
        names = _ext_names(os.path.join(REF, "exllamav3", m)) - OUTSIDE
        miss = sorted(n for n in names if not hasattr(ext, n))
        if miss:
            missing[m] = miss
    assert not missing, f"hot-path ops the reference calls but the mirror lacks: {missing}"


@needs_ref
def test_reference_classes_reach_the_mirror_with_compatible_signatures():
    """The reference's LinearEXL3.forward (kernel, reconstruct + hgemm and fused-reconstruct routes), RMSNorm.forward and RoPE.apply, constructed
    from the reference's own code over the stub, on CPU tensors: every call must arrive at this build's op with a compatible argument list and
    stop at its "tensor must be on a GPU device" check (the GPU variant of the same script compares the results with the oracle)."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([STUB_DIR, ROOT, REF]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_reference_seam_script.py"), "cpu"], capture_output=True, text=True,
                       timeout=600, env=env, cwd="/tmp")
    assert r.returncode == 0 and "REFERENCE_CALLS_OK cpu" in r.stdout, (r.stdout + r.stderr)[-3000:]


@needs_ref
def test_loader_and_writer_pinned_against_the_reference_reader_and_naming_code():
    """SURVEY.md 8f rank 4, parity pinned (task 4a): a checkpoint written by save_checkpoint is read by the reference's own
    SafetensorsCollection + Linear.load_exl3; a checkpoint named by the reference's LinearEXL3.get_tensors (3INST / mcg / mul1 markers, bias,
    legacy packed su / sv) is read by exllamav3_amd.loader; stored legacy .su / .sv files through both readers; tensor-parallel shard reads
    against slices of the reference's whole tensors -- every tensor bit for bit (tests/_reference_loader_script.py)."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([STUB_DIR, ROOT, REF]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_reference_loader_script.py")], capture_output=True, text=True,
                       timeout=600, env=env, cwd="/tmp")
    assert r.returncode == 0 and "REFERENCE_LOADER_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
    for tag in ("A_OK 45", "B_OK 4", "C_OK", "D_OK"):
        assert tag in r.stdout
