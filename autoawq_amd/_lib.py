"""ctypes loader for libawq_hip.so (the C ABI declared in include/awq_hip.h).

There is deliberately no fallback: if the shared library is missing, or a tensor is not on a
HIP device, the call raises.  The CPU oracle under oracle/ is test infrastructure and is never
imported from this package.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libawq_hip.so")

_lib = None

c_void_p, c_int, c_int64, c_size_t, c_uint32, c_float = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                           ctypes.c_size_t, ctypes.c_uint32, ctypes.c_float)

class AwqGemmEx(ctypes.Structure):
    """struct AwqGemmEx of include/awq_hip.h (same field order)."""
    _fields_ = [("struct_bytes", ctypes.c_uint32), ("flags", ctypes.c_uint32),
                ("x", c_void_p), ("qweight", c_void_p), ("scales", c_void_p), ("qzeros", c_void_p), ("bias", c_void_p),
                ("y", c_void_p), ("M", c_int64), ("K", c_int64), ("N", c_int64), ("group_size", c_int64),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t), ("stream", c_void_p),
                ("norm_weight", c_void_p), ("norm_eps", c_float), ("residual_in", c_void_p), ("residual_out", c_void_p),
                ("ssq_in", c_void_p), ("ssq_in_tiles", c_int64), ("add_residual", c_void_p), ("ssq_out", c_void_p)]


class AwqGemvEx(ctypes.Structure):
    """struct AwqGemvEx of include/awq_hip.h (same field order)."""
    _fields_ = [("struct_bytes", ctypes.c_uint32), ("flags", ctypes.c_uint32),
                ("x", c_void_p), ("qweight", c_void_p), ("scales", c_void_p), ("qzeros", c_void_p), ("y", c_void_p),
                ("M", c_int64), ("K", c_int64), ("N", c_int64), ("group_size", c_int64), ("zeros_width", c_int64),
                ("stream", c_void_p), ("norm_weight", c_void_p), ("norm_eps", c_float), ("add_residual", c_void_p)]


# name -> (restype, argtypes); must list every symbol include/awq_hip.h declares
# (tests/test_boundary.py cross-checks this table against the header and the .so).
SIGNATURES = {
    "awq_allreduce_staging_bytes": (c_size_t, [c_int64]),
    "awq_allreduce_flag_bytes": (c_size_t, []),
    "awq_allreduce_state_bytes": (c_size_t, []),
    "awq_allreduce_oneshot": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "awq_allreduce_alloc": (c_int, [c_void_p, c_size_t]),
    "awq_allreduce_free": (c_int, [c_void_p]),
    "awq_allreduce_ipc_export": (c_int, [c_void_p, c_void_p]),
    "awq_allreduce_ipc_open": (c_int, [c_void_p, c_void_p]),
    "awq_allreduce_ipc_close": (c_int, [c_void_p]),
    "awq_allreduce_oneshot_group": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "awq_hip_abi_version": (c_int, []),
    "awq_hip_error_string": (ctypes.c_char_p, [c_int]),
    "awq_hip_last_kernel": (ctypes.c_char_p, []),
    "awq_unpack_int4": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "awq_dequantize_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "awq_gemm_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "awq_gemm_workspace_init": (c_int, [c_void_p, c_size_t, c_void_p]),
    "awq_gemm_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                 c_int64, c_int64, c_void_p, c_size_t, c_uint32, c_void_p]),
    "awq_gemm_forward_normed": (c_int, [c_void_p] * 4 + [c_float] + [c_void_p] * 5 + [c_int64] * 4 +
                                [c_void_p, c_size_t, c_uint32, c_void_p]),
    "awq_gemm_ex_ssq_tiles": (c_int64, [c_int64]),
    "awq_gemm_forward_ex": (c_int, [ctypes.POINTER(AwqGemmEx)]),
    "awq_silu_and_mul": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "awq_rmsnorm_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_float, c_void_p]),
    "awq_rope_kv_append": (c_int, [c_void_p] * 7 + [c_int64] * 8 + [c_void_p]),
    "awq_decode_attention_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "awq_decode_attention": (c_int, [c_void_p] * 5 + [c_int64] * 7 + [c_float, c_void_p, c_size_t, c_void_p]),
    "awq_repack_gemv_to_gemm": (c_int, [c_void_p] * 6 + [c_int64] * 4 + [c_void_p]),
    "awq_prefill_attention": (c_int, [c_void_p] * 4 + [c_int64] * 7 + [c_float, c_float, c_void_p, c_void_p]),
    "awq_decode_attention_ex": (c_int, [c_void_p] * 5 + [c_int64] * 7 + [c_float, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "awq_decode_attention_rope": (c_int, [c_void_p] * 7 + [c_int64] * 7 + [c_float, c_void_p, c_size_t, c_void_p]),
    "awq_moe_route": (c_int, [c_void_p] * 6 + [c_int64, c_int64, c_int64, c_int, c_int64, c_void_p]),
    "awq_moe_route_local": (c_int, [c_void_p] * 6 + [c_int64, c_int64, c_int64, c_int, c_int64, c_int64, c_int64, c_void_p]),
    "awq_grouped_gemm_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "awq_grouped_gemm_forward": (c_int, [c_void_p] * 9 + [c_int64] * 8 + [c_void_p, c_size_t, c_void_p]),
    "awq_grouped_gemm_forward_ex": (c_int, [c_void_p] * 9 + [c_int64] * 8 + [c_void_p, c_size_t, c_uint32, c_void_p]),
    "awq_grouped_gemm_prefill": (c_int, [c_void_p] * 6 + [c_int64] * 5 + [c_uint32, c_void_p]),
    "awq_grouped_gemm_prefill_ex": (c_int, [c_void_p] * 8 + [c_int64] * 6 + [c_uint32, c_void_p]),
    "awq_moe_sort_pairs": (c_int, [c_void_p] * 3 + [c_int64, c_int64, c_void_p]),
    "awq_gemv_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                 c_int64, c_uint32, c_void_p]),
    "awq_gemv_auto_kernel": (c_int, [c_int64, c_int64, c_int64, c_int64]),
    "awq_gemv_forward_ex": (c_int, [ctypes.POINTER(AwqGemvEx)]),
    "awq_grouped_gemv_forward": (c_int, [c_void_p] * 7 + [c_int64] * 8 + [c_uint32, c_int64, c_void_p]),
    "awq_gemv_lds_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "awq_dequantize_weights_gemv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                            c_int64, c_void_p]),
    "awq_gemv_fast_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                      c_int64, c_int64, c_uint32, c_void_p]),
    "awq_gemv_fast_lds_bytes_c": (c_size_t, [c_int64, c_int64, c_int64]),
    "awq_gemv_fast_prefill": (c_int, [c_void_p] * 6 + [c_int64] * 5 + [c_uint32, c_void_p]),
    "awq_repack_gemvfast_to_gemm": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "awq_dequantize_weights_gemv_fast": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                                 c_void_p]),
    "awq_gemm_workspace_status": (c_int, [c_void_p, c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    "awq_gemm_auto_kernel": (c_int, [c_int64, c_int64, c_int64, c_int64]),
}


class AwqHipError(RuntimeError):
    code = 0  # the AWQ_ERR_* value when the error came from the library


ERR_UNSUPPORTED = -3  # AWQ_ERR_UNSUPPORTED: a valid AWQ tensor, but no kernel for it


def available():
    return os.path.exists(LIB_PATH)


def lib():
    """Load (once) and return the ctypes handle.  torch is imported first so that the HIP runtime
    already mapped by torch (same SONAME libamdhip64.so.7) is the one our kernels register with."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (side effect: maps torch's libamdhip64 before ours resolves it)

        if not os.path.exists(LIB_PATH):
            raise AwqHipError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python autoawq_amd/csrc/build.py` (or __graft_entry__.build()). "
                "autoawq_amd has no CPU fallback.")
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.awq_hip_abi_version() != 1:
            raise AwqHipError("libawq_hip.so ABI version mismatch")
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().awq_hip_error_string(rc).decode()
        err = AwqHipError(f"{what}: {msg} (code {rc})")
        err.code = rc
        raise err
