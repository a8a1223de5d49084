"""ctypes binding of libslim_hip.so (declared in include/slim_hip.h).

The library is the only compute path of this package: if it is missing the import fails
loudly; there is no CPU or eager-PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libslim_hip.so")

vp, i64, i32, f32, u32 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_uint32

# name -> (restype, argtypes); mirrors include/slim_hip.h one to one
PROTOTYPES = {
    "slim_abi_version": (i32, []),
    "slim_last_error": (C.c_char_p, []),
    "slim_profile_enable": (i32, [i32]),
    "slim_profile_summary": (i32, [C.c_char_p, i64]),
    "slim_colsumsq_workspace_bytes": (i64, [i64, i64, i64]),
    "slim_scaler_row_accum": (i32, [vp, i64, i64, i64, i64, i64, vp, i64, vp, i64, vp]),
    "slim_colsumsq": (i32, [vp, i64, i64, i64, i64, vp, vp, i64, vp]),
    "slim_scaler_row_fold": (i32, [vp, i64, vp, i64, i64, i64, vp]),
    "slim_gram_workspace_bytes": (i64, [i64, i64]),
    "slim_gram_accum": (i32, [vp, i64, i64, i64, vp, i64, i64, vp, i64, vp]),
    "slim_shift_zeros": (i32, [vp, vp, i64, vp]),
    "slim_sqrt_f32": (i32, [vp, vp, i64, vp]),
    "slim_any_zero_f32": (i32, [vp, i64, vp, vp]),
    "slim_outlier_stats": (i32, [vp, i64, i32, vp, vp, vp]),
    "slim_mask_nm": (i32, [vp, vp, i64, i64, i32, i32, vp, vp]),
    "slim_prune_nm_f32": (i32, [vp, i64, i64, i32, i32, vp, vp]),
    "slim_mask_unstructured": (i32, [vp, vp, i64, i64, i64, vp, vp]),
    "slim_prune_unstructured_f32": (i32, [vp, i64, i64, i64, vp, vp]),
    "slim_nm_lut4": (i32, [i32, vp]),
    "slim_hist_abs_bf16": (i32, [vp, i64, vp, i32, vp]),
    "slim_quant_search_workspace_bytes": (i64, [i32]),
    "slim_quant_search": (i32, [vp, i32, i32, vp, vp, vp, vp, i64, vp]),
    "slim_quant_search_noalloc": (i32, [vp, i32, i32, vp, vp, vp, vp]),
    "slim_absmax_scale": (i32, [vp, i32, vp, vp, vp]),
    "slim_quantize_pt": (i32, [vp, i64, vp, i32, vp, vp]),
    "slim_dequantize_pt": (i32, [vp, i64, vp, vp, i32, vp]),
    "slim_fake_quant_pt": (i32, [vp, i64, vp, i32, vp, vp, vp]),
    "slim_error_matrix": (i32, [vp, vp, vp, vp, i32, i64, i64, vp, vp, vp]),
    "slim_error_matrix_deq": (i32, [vp, vp, vp, vp, i64, i64, vp, vp, vp]),
    "slim_important_columns_workspace_bytes": (i64, [i64, i64]),
    "slim_important_columns": (i32, [vp, vp, i64, i64, i64, vp, vp, i64, vp]),
    "slim_scale_columns_bf16": (i32, [vp, vp, i64, i64, f32, vp, vp]),
    "slim_scale_columns_f32": (i32, [vp, vp, i64, i64, f32, vp]),
    "slim_error_matrix_cols": (i32, [vp, vp, vp, vp, i32, vp, i32, i64, i64, vp, vp, vp]),
    "slim_error_matrix_t_rounded": (i32, [vp, vp, vp, vp, i32, i32, i64, i64, vp, vp, vp]),
    "slim_transpose_bf16": (i32, [vp, i64, i64, vp, vp]),
    "slim_gemm_nt_bf16": (i32, [vp, i64, vp, i64, i32, i32, i32, f32, f32, vp, i64, vp, i64, vp, i64, vp]),
    "slim_lora_prune_workspace_bytes": (i64, [i64, i64, i64]),
    "slim_lora_prune_optimize": (i32, [vp, vp, vp, vp, i64, i64, i64, i32, f32, f32, vp, vp, i64, vp]),
    "slim_gemm_set_tile": (None, [i32]),
    "slim_linear24_set_wide": (None, [i32]),
    "slim_gemm_set_family_tile": (None, [i32, i32]),
    "slim_gemm_set_trace": (None, [vp, i32]),
    "slim_eigh_workspace_bytes": (i64, [i32]),
    "slim_tridiagonalize_f32_batched": (i32, [vp, i32, i32, vp, vp, vp, vp, i64, vp]),
    "slim_eigh_tail_f32": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, i64, vp]),
    "slim_eigh_tail_workspace_bytes": (i64, [i32, i32, i32]),
    "slim_eigh_tail_f32_batched": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, i64, vp]),
    "slim_lowrank_fit_set_filter": (i32, [vp, i32, i32]),
    "slim_ritz_split_supported": (i32, [i32, i32]),
    "slim_ritz_split_workspace_bytes": (i64, [i32, i32, i32]),
    "slim_ritz_split_f32_batched": (i32, [vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, i64, vp]),
    "slim_gemm_f32": (i32, [vp, i64, i32, vp, i64, i32, i32, i32, i32, f32, vp, i64, vp]),
    "slim_lowrank_fit_front": (i32, [vp, vp, i64, i64, i32, i32, i32, u32, i32, i32, vp, vp, vp, i64, vp]),
    "slim_lowrank_fit_back": (i32, [i64, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "slim_lowrank_fit_front_batched": (i32, [vp, vp, i32, i64, i64, i32, i32, i32, u32, i32, i32, vp, vp, vp, i64, vp]),
    "slim_lowrank_fit_group_workspace_bytes": (i64, [vp, vp, i32, i64]),
    "slim_lowrank_fit_front_group": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, u32, i32, i32, vp, vp, vp, i64, vp]),
    "slim_lowrank_fit_back_group": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "slim_lowrank_fit_back_batched": (i32, [i32, i64, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "slim_tridiagonalize_f32": (i32, [vp, i32, vp, vp, vp, vp, i64, vp]),
    "slim_eigh_f32": (i32, [vp, i32, i32, vp, vp, vp, i64, vp]),
    "slim_orthonormalize_workspace_bytes": (i64, [i64, i64]),
    "slim_orthonormalize_bf16": (i32, [vp, vp, i64, i32, i32, i32, i32, vp, vp, vp, vp, i64, vp]),
    "slim_lowrank_fit_workspace_bytes": (i64, [i64, i64, i64]),
    "slim_gemm_nt_bf16_splitk": (i32, [vp, i64, vp, i64, i32, i32, i32, f32, vp, i64, vp, i64, i32, vp, vp]),
    "slim_gemm_nt_bf16_splitk_t": (i32, [vp, i64, vp, i64, i32, i32, i32, f32, vp, i64, vp, i64, i32, vp, vp]),
    "slim_lowrank_fit": (i32, [vp, vp, i64, i64, i32, i32, i32, u32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "slim_lowrank_fit_batched_workspace_bytes": (i64, [i32, i64, i64, i64]),
    "slim_lowrank_fit_batched": (i32, [vp, vp, i32, i64, i64, i32, i32, i32, u32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp]),
    "slim_gramfit_workspace_bytes": (i64, [i64, i64]),
    "slim_gramfit_iterate": (i32, [vp, i64, i32, i32, u32, i32, i32, vp, vp, vp, vp, i64, vp]),
    "slim_fmod_probe_f32": (i32, [vp, vp, i64, vp, vp, vp]),
    "slim_floordiv_probe_f32": (i32, [vp, vp, i64, vp, vp, vp]),
    "slim_lora_adapters": (i32, [vp, vp, i64, i64, i32, i32, i32, vp, vp, vp]),
    "slim_gramfit_batched_workspace_bytes": (i64, [i32, i64, i64]),
    "slim_gramfit_iterate_batched": (i32, [vp, i32, i64, i32, i32, u32, i32, i32, vp, vp, vp, vp, vp, i64, vp]),
    "slim_gramfit_check": (i32, [vp, i32, i32, i32, i32, vp, vp, vp]),
    "slim_gramfit_project": (i32, [vp, i64, i64, i32, vp, vp, vp, vp, vp, i64, vp]),
    "slim_gramfit_factors": (i32, [i32, vp, vp, i64, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i64, vp]),
    "slim_lora_assemble": (i32, [vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "slim_outlier_project": (i32, [vp, vp, i64, i64, i32, vp, vp]),
    "slim_error_matrix_t": (i32, [vp, vp, vp, vp, i32, i64, i64, vp, vp, vp]),
    "slim_deflate_apply": (i32, [vp, vp, vp, i64, i64, i32, vp, vp, vp]),
    "slim_outlier_basis_workspace_bytes": (i64, [i32]),
    "slim_outlier_basis": (i32, [vp, i64, vp, i32, i32, vp, vp, vp, i64, vp]),
    "slim_lora_prepare_workspace_bytes": (i64, [i64, i64, i32]),
    "slim_lora_prepare": (i32, [vp, vp, i64, i64, i32, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp,
                                vp, i64, vp]),
    "slim_lora_finish_workspace_bytes": (i64, [i64, i64]),
    "slim_lora_finish": (i32, [vp, vp, i64, i64, i32, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, i64, vp]),
    "slim_lowrank_residual": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]),
    "slim_residual_weight": (i32, [vp, vp, vp, i64, vp, vp, vp]),
    "slim_randn_bf16": (i32, [vp, i64, u32, vp]),
    "slim_scale_rows_bf16": (i32, [vp, vp, i64, i64, vp]),
    "slim_group_absmax": (i32, [vp, i64, i64, i32, vp, vp]),
    "slim_group_quantize": (i32, [vp, vp, i64, i64, i32, i32, vp, vp]),
    "slim_group_dequantize": (i32, [vp, vp, i64, i64, i32, i32, vp, vp]),
    "slim_group_fake_quant": (i32, [vp, i64, i64, i32, i32, vp, vp, vp, vp]),
    "slim_input_fake_quant_fp8": (i32, [vp, i64, vp, vp, vp]),
    "slim_input_fake_quant_int": (i32, [vp, i64, i32, i32, f32, vp, vp]),
    "slim_packed24_bytes": (i32, [i64, i64, vp, vp]),
    "slim_pack24_int4": (i32, [vp, vp, i64, i64, vp, vp, vp, vp]),
    "slim_unpack24_int4": (i32, [vp, vp, vp, i64, i64, vp, vp]),
    "slim_linear24_workspace_bytes": (i64, [i64, i64, i64, i64]),
    "slim_linear24_forward_q4": (i32, [vp, i64, i64, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, i64, i32, f32, vp, i64, vp, i64, vp]),
    "slim_linear24_forward": (i32, [vp, i64, i64, i64, vp, vp, vp, i64, vp, vp, vp, i64, f32, vp, i64, vp, i64, vp]),
}

ERR_INVALID, ERR_LAUNCH, ERR_UNSUPPORTED, ERR_SOLVER = -1, -2, -3, -4


class SlimHipError(RuntimeError):
    def __init__(self, name, code, msg):
        super().__init__(f"{name} failed ({code}): {msg}")
        self.code = code


_lib = None


def load():
    """Load the shared library (once) and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m slim_amd.build` (needs hipcc). "
            "slim_amd has no fallback path without its HIP library.")
    # torch must be imported first so that the process-wide HIP runtime is the one torch uses
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header and library out of sync
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def call(name, *args):
    """Call an int-returning entry point; raise on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.slim_last_error().decode(errors="replace")
        if rc == ERR_UNSUPPORTED:
            raise NotImplementedError(f"{name}: {msg}")
        if rc == ERR_INVALID:
            raise ValueError(f"{name}: {msg}")
        raise SlimHipError(name, rc, msg)


def query(name, *args):
    """Call an entry point that returns a value (workspace sizes, version)."""
    return getattr(load(), name)(*args)


def profile_enable(on=True):
    call("slim_profile_enable", int(bool(on)))


def profile_summary():
    """{scope name: dict(count, ms, bytes, flops)} of the launches recorded since profile_enable; bytes and
    flops are the algorithmic traffic / arithmetic the library attributes to those launches."""
    buf = C.create_string_buffer(1 << 16)
    call("slim_profile_summary", buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms, nbytes, flops = line.split()
        out[name] = dict(count=int(cnt), ms=float(ms), bytes=float(nbytes), flops=float(flops))
    return out
