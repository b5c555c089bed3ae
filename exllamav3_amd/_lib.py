"""
ctypes binding of the C-ABI library libexl3_hip.so (include/exl3_hip.h).

The product path has NO fallback: if the shared library is missing or fails to load, importing any op raises.
"""
from __future__ import annotations
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EXL3_HIP_LIB") or os.path.join(_HERE, "libexl3_hip.so")      # override: diagnostics builds only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "exl3_hip.h")

_lib = None
ABI_VERSION = 4            # include/exl3_hip.h EXL3_ABI_VERSION


def declared_symbols() -> list[str]:
    """Every function name declared in include/exl3_hip.h."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    names = re.findall(r"\b(exl3_[a-z0-9_]+)\s*\(", src)
    out = []
    for n in names:
        if n not in out:
            out.append(n)
    return out


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"exllamav3_amd: HIP library not built ({LIB_PATH}); run `python __graft_entry__.py` "
                f"(there is no CPU fallback for the EXL3 hot path)")
        # torch first: its wheel bundles its own ROCm runtime (libamdhip64 & co.), and this library must bind to THAT copy.  Loading
        # libexl3_hip.so before torch pulls in /opt/rocm's runtime with RTLD_GLOBAL and a later `import torch` in the same process segfaults
        # (two HIP runtimes interposing each other's symbols).  A pure-C consumer has one runtime and no such ordering problem.
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        _lib.exl3_last_error.restype = ctypes.c_char_p
        ver = _lib.exl3_abi_version()
        if ver != ABI_VERSION:
            raise RuntimeError(f"exllamav3_amd: ABI version mismatch (library {ver}, binding {ABI_VERSION}): rebuild with `python __graft_entry__.py`")
        _declare(_lib)
    return _lib


def check_symbols() -> list[str]:
    """Load the library (no GPU needed) and verify it exports every symbol the header declares."""
    l = lib()
    missing = [s for s in declared_symbols() if not hasattr(l, s)]
    if missing:
        raise RuntimeError(f"libexl3_hip.so does not export: {missing}")
    return declared_symbols()


def last_error() -> str:
    return lib().exl3_last_error().decode("utf-8", "replace")


class Exl3Error(RuntimeError):
    pass


def check(rc: int) -> int:
    """Reference behaviour: argument violations surface as RuntimeError (TORCH_CHECK, util.h:24-35)."""
    if rc < 0:
        raise Exl3Error(last_error())
    return rc


vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
u32 = ctypes.c_uint32


class PstepLinear(ctypes.Structure):
    """exl3_pstep_linear_t"""
    _fields_ = [("trellis", vp), ("suh", vp), ("svh", vp), ("k", i32), ("n", i32), ("K", i32), ("cb", i32)]


class PstepLayer(ctypes.Structure):
    """exl3_pstep_layer_t"""
    _fields_ = [("q", PstepLinear), ("k", PstepLinear), ("v", PstepLinear), ("o", PstepLinear), ("gate", PstepLinear), ("up", PstepLinear),
                ("down", PstepLinear), ("norm1", vp), ("norm2", vp), ("k_cache", vp), ("k_scales", vp), ("v_cache", vp), ("v_scales", vp)]


def _declare(l):
    def sig(name, *argtypes):
        fn = getattr(l, name)
        fn.argtypes = list(argtypes)
        fn.restype = ctypes.c_int
    sig("exl3_init", i32)
    sig("exl3_device_info", i32, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i64))
    sig("exl3_pack_trellis", vp, vp, i32, i32, i32, vp)
    sig("exl3_unpack_trellis", vp, vp, i32, i32, i32, vp)
    sig("exl3_pack_signs", vp, vp, i64, vp)
    sig("exl3_decode", vp, vp, i64, i32, i32, vp)
    sig("exl3_reconstruct", vp, vp, i32, i32, i32, i32, i64, i64, vp)
    sig("exl3_reconstruct_had", vp, vp, vp, vp, i32, i32, i32, i32, i64, i64, vp)
    sig("exl3_had_r_128", vp, vp, vp, vp, f32, i32, i32, i32, vp)
    sig("exl3_gemm", vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp)
    sig("exl3_mgemm", vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
        ctypes.POINTER(i32), i32, i32, i32, i32, i32, i32, i32, vp)
    sig("exl3_hgemm", vp, vp, vp, i32, i32, i32, i64, i32, vp)
    sig("exl3_hgemm_acc", vp, vp, vp, i32, i32, i32, i64, vp)
    sig("exl3_hgemm_nt", vp, vp, vp, i32, i32, i32, i64, i64, i32, i32, vp)
    sig("exl3_hgemm_nt_lda", vp, i64, vp, vp, i32, i32, i32, i64, i64, i32, i32, vp)
    sig("exl3_reconstruct_had_t", vp, i64, vp, vp, vp, i32, i32, i32, i32, i64, i64, vp)
    sig("exl3_rms_norm", vp, vp, vp, vp, f32, f32, f32, i32, i32, i32, i32, i32, i32, i32, vp)
    sig("exl3_rope", vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, u32, vp, vp, i32, f32, vp, vp, f32, f32, vp)
    sig("exl3_quant_cache_cont", vp, vp, vp, i64, i32, i32, vp)
    sig("exl3_dequant_cache_cont", vp, vp, vp, i64, i32, i32, vp)
    sig("exl3_quant_cache_paged", vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp)
    sig("exl3_quant_cache_cont_ex", vp, vp, vp, i64, i32, i32, f32, vp)
    sig("exl3_dequant_cache_cont_ex", vp, vp, vp, i64, i32, i32, f32, vp)
    sig("exl3_quant_cache_paged_ex", vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i64, i64, f32, i32, vp)
    sig("exl3_dequant_cache_paged_ex", vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, i32, i32, vp)
    sig("exl3_quant_cache_paged_strided", vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i64, i64, vp)
    sig("exl3_rope_ex", vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, u32, vp, vp, i32, i32, f32, vp, vp, i32, f32, f32, i32, i32, f32, i32, i32, i32, i32, vp)
    sig("exl3_rope_strided", vp, vp, vp, i32, i32, i32, i32, i64, i64, u32, vp, vp, f32, vp)
    sig("exl3_dequant_cache_paged", vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp)
    sig("exl3_silu_mul", vp, vp, vp, i64, i32, vp)
    sig("exl3_act_mul", vp, vp, vp, i64, i32, i32, f32, vp)
    sig("exl3_add_sigmoid_gate", vp, vp, vp, i64, i32, vp)
    sig("exl3_add_sigmoid_gate_proj", vp, vp, vp, vp, i32, i32, vp)
    sig("exl3_paged_kv_cache_update", vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp)
    sig("exl3_mul_gate", vp, vp, i64, i32, i32, vp)
    sig("exl3_deinterleave_qg", vp, vp, vp, i64, i32, vp)
    sig("exl3_add", vp, vp, i64, i32, i32, vp)
    sig("exl3_softcap", vp, vp, i64, f32, i32, vp)
    sig("exl3_ar_create", i32, i32, i64, ctypes.POINTER(vp), vp)
    sig("exl3_ar_open_peer", vp, i32, vp)
    sig("exl3_ar_destroy", vp)
    sig("exl3_ar_error", vp, vp)
    sig("exl3_ar_epoch", vp, ctypes.POINTER(ctypes.c_uint32), vp)
    sig("exl3_ar_reduce", vp, vp, vp, vp, vp, i32, i32, vp)
    sig("exl3_ar_reduce_slabs", vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, vp)
    # persistent decode step (exl3_pstep.hip): plan structures mirror include/exl3_hip.h
    sig("exl3_pstep_create", ctypes.POINTER(vp), ctypes.POINTER(PstepLayer), i32, ctypes.POINTER(PstepLinear), vp, i32, i32, i32, i32, i32, i32, f32, i32, i32)
    sig("exl3_pstep_run", vp, vp, vp, vp, vp, vp, vp, vp)
    sig("exl3_pstep_run_attn", vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp)
    sig("exl3_pstep_plan_tiles", i32, i32, i32, i32, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(i32))
    sig("exl3_pstep_error", vp, vp)
    sig("exl3_pstep_error_peek", vp)
    sig("exl3_pstep_attn_geometry", vp, i32, ctypes.POINTER(i32))
    sig("exl3_pstep_unpack_op", vp, i32, i32, vp, vp)
    sig("exl3_pstep_tp_handle", vp, vp)
    sig("exl3_pstep_tp_open_peer", vp, i32, vp)
    sig("exl3_pstep_tp_commit", vp)
    l.exl3_pstep_tp_peek.argtypes = [vp, vp, i64]
    l.exl3_pstep_tp_peek.restype = i64
    sig("exl3_pstep_set", vp, i32, i32)
    sig("exl3_pstep_describe", vp, ctypes.c_char_p, i32)
    sig("exl3_pstep_destroy", vp)
    l.exl3_pstep_stamps.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), i64, vp]
    l.exl3_pstep_stamps.restype = i64
    PP = ctypes.POINTER(vp)
    sig("exl3_reconstruct_had_multi_t", vp, i64, PP, PP, PP, ctypes.POINTER(i32), i32, i32, i32, i32, vp)
    sig("exl3_reconstruct_had_multi_t_interleaved", vp, i64, PP, PP, PP, ctypes.POINTER(i32), i32, i32, i32, i32, vp)
    sig("exl3_gemv_ex", vp, PP, PP, PP, PP, PP, PP, PP, ctypes.POINTER(i32), i32, i32, i32, i32, i32, i32, i32, i32, PP, ctypes.POINTER(i32), vp)
    sig("exl3_glue_norm", vp, i32, vp, vp, vp, vp, vp, f32, PP, PP, PP, i32, i32, i32, vp, vp)
    sig("exl3_glue_qkv", vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp)
    sig("exl3_glue_act", vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp)
    sig("exl3_gemv_norm", vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, f32, vp, vp, vp, i32, vp, vp)
    sig("exl3_gemv_resid", vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp)
    sig("exl3_set_tail_xcd_local", i32)
    sig("exl3_gemv_act", vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp)
    sig("exl3_gemv_qkv", vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp)
    sig("exl3_rope_table", vp, vp, f32, i32, vp, vp, vp)
    sig("exl3_debug_copy_workspace", vp, i64, i64, vp)
    sig("exl3_attn_prefill_paged_strided", vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp)
    sig("exl3_attn_prefill_paged", vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp)
    sig("exl3_attn_decode_qcache", vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, i64, vp)
    sig("exl3_silu_mul_2d", vp, vp, vp, i64, i64, i64, i64, vp)
    sig("exl3_gemv_ex_resid", vp, vp, vp, f32, vp, i32, vp, vp, vp, PP, PP, ctypes.POINTER(i32), i32, i32, i32, i32, i32, i32, i32, PP, ctypes.POINTER(i32), vp)
    sig("exl3_gemv_ex_wpc", vp, PP, PP, ctypes.POINTER(i32), i32, i32, i32, i32, i32, i32, i32, PP, ctypes.POINTER(i32), vp)
    sig("exl3_attn_decode_qcache_split_qkv", vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, f32, i32, vp, vp, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32,
        i32, i32, i32, i32, i32, i32, f32, vp, i64, ctypes.POINTER(i32), ctypes.POINTER(i32), vp)
    sig("exl3_attn_decode_qcache_sinks", vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, i64, vp, vp)
    sig("exl3_attn_decode_qcache_split", vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, i64, ctypes.POINTER(i32), vp)
    sig("exl3_gemv_ex_attm", vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, PP, ctypes.POINTER(i32), vp)
    sig("exl3_gemv_ex_qkvm", vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32,
        i32, i32, ctypes.POINTER(vp), ctypes.POINTER(i32), vp)
    sig("exl3_gemv_ex_act_rs", vp, vp, i32, vp, vp, vp, vp, i32, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, PP, ctypes.POINTER(i32), vp)
    sig("exl3_glue_qkv_rs", vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, i32, f32, vp)
    sig("exl3_gemv_ex_act", vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, PP, ctypes.POINTER(i32), vp)
    sig("exl3_routing_std", vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp)
    sig("exl3_mgemm_indexed_act_deferred", vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, PP, ctypes.POINTER(i32), vp)
    sig("exl3_glue_resid_moe", vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, vp)
    sig("exl3_routing_std_scaled", vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp)
    sig("exl3_routing_std_norm", vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp)
    sig("exl3_routing_std_fx", vp, vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp)
    sig("exl3_mgemm_indexed_deferred", vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, PP, ctypes.POINTER(i32), vp)
    sig("exl3_mgemm_indexed_act_fx", vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp)
    sig("exl3_routing_std_slots", vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp)
    sig("exl3_mgemm_indexed_act", vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp)
    sig("exl3_set_glue_threads", i32)
    sig("exl3_qkv_prep", vp, vp, f32, i32, i32, vp, i32, i32, vp, vp, vp, vp)
    sig("exl3_glue_qkv_tab", vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, i32, f32, vp, vp, vp, vp)
    sig("exl3_glue_resid_rotate", vp, i32, vp, vp, vp, vp, vp, vp, vp, f32, PP, PP, PP, i32, i32, i32, vp)
    sig("exl3_glue_act_rs", vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, f32, vp)
    sig("exl3_glue_rotate", vp, vp, vp, f32, PP, PP, PP, i32, i32, i32, vp)
    sig("exl3_mgemm_indexed", vp, i32, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp)
    sig("exl3_mgemm_indexed_nlist", vp, i32, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp)
    sig("exl3_gemv_ex_norm", vp, vp, vp, f32, PP, PP, PP, PP, PP, ctypes.POINTER(i32), i32, i32, i32, i32, i32, i32, i32, i32, PP, ctypes.POINTER(i32), vp)
    sig("exl3_glue_resid", vp, i32, vp, vp, vp, vp, vp, i32, i32, vp)
    sig("exl3_fx_init", vp, vp, vp, i32, i32, vp)
    sig("exl3_fx_finish", vp, vp, vp, i32, i32, vp)
    sig("exl3_fx_init_prep", vp, vp, vp, i32, i32, vp, vp, f32, i32, vp, i32, i32, vp, vp, vp, vp)
    sig("exl3_fx_finish_rotate", vp, vp, vp, vp, f32, vp, vp, vp, i32, i32, vp)
    sig("exl3_gemv_ex_fx", vp, vp, vp, vp, f32, PP, PP, ctypes.POINTER(i32), i32, i32, i32, i32, i32, i32, PP, ctypes.POINTER(i32), vp)
    sig("exl3_moe_build_slots", vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp)
    sig("exl3_moe_scatter", vp, vp, vp, vp, vp, i32, i32, i32, vp)
    sig("exl3_gemm_nt_mfma", vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp)
    sig("exl3_gemm_nt2_mfma", vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp)
    sig("exl3_gemm_nt2_grouped", vp, i64, vp, i64, i64, vp, i64, vp, i32, i32, i32, i32, i32, vp)
    sig("exl3_gemv_ex_fx_atomic", vp, vp, vp, vp, f32, PP, PP, PP, PP, ctypes.POINTER(i32), i32, i32, i32, i32, i32, i32, ctypes.POINTER(i32), vp)
    sig("exl3_gemv_ex_actfx", vp, vp, vp, vp, i32, f32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, PP, ctypes.POINTER(i32), vp)
    sig("exl3_fx_zero_next", vp, i64)
    sig("exl3_fx_add", vp, vp, vp, i32, vp, i32, i32, vp)
    sig("exl3_ar_reduce_fx", vp, vp, vp, i32, vp, vp, i32, i32, vp)
    sig("exl3_set_gemv_variant", i32)
    sig("exl3_set_gemv_max_waves", i32)
    sig("exl3_set_gemv_gen4", i32)
    sig("exl3_set_gemm3_min_rows", i32)
    sig("exl3_set_gemm3_cpw", i32)
    sig("exl3_set_attn_wide_waves", i32)
    sig("exl3_set_gemv_defer_wg_per_cu", i32)
