"""ctypes binding of ``libaid_hip.so`` (C ABI declared in ``include/aid_hip.h``).

The shared library is built in-tree (``attention-interpolation-diffusion_amd/libaid_hip.so``) by
``__graft_entry__.build()`` / ``csrc/Makefile``.  There is NO fallback: if the library is missing
or a call returns a non-zero code, a ``RuntimeError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AID_LIB_PATH") or os.path.join(PKG_DIR, "libaid_hip.so")   # override: development A/B builds
CSRC_DIR = os.path.join(PKG_DIR, "csrc")

AID_ABI_VERSION = 8
DTYPE_F16, DTYPE_BF16, DTYPE_F32 = 0, 1, 2
MODE_PLAIN, MODE_INNER, MODE_OUTER = 0, 1, 2
GEMM_MAX_PROBLEMS = 6
IP_NONE, IP_SAME, IP_PLAIN = 0, 1, 2

# every symbol include/aid_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    "aid_gemm_nt", "aid_layernorm", "aid_ln_stats", "aid_ln_fold", "aid_attn_fwd", "aid_lerp_kv", "aid_processor_workspace_bytes", "aid_processor_fwd",
    "aid_abi_version", "aid_strerror", "aid_last_attn_variant", "aid_last_gemm_variant", "aid_device_info",
    "aid_profile_begin", "aid_profile_end", "aid_set_tuning", "aid_get_tuning", "aid_stream_capture_id",
)


class AidGemmProblem(C.Structure):
    _fields_ = [
        ("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("bias", C.c_void_p),
        ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
        ("batch", C.c_int32), ("scale", C.c_float),
        ("stride_a", C.c_int64), ("stride_b", C.c_int64), ("stride_c", C.c_int64),
        ("residual", C.c_void_p),
        ("ln_stats", C.c_void_p), ("ln_colsum", C.c_void_p), ("ln_shift", C.c_void_p),
        ("ln_side", C.c_int32), ("trans_rows", C.c_int32), ("stride_stats", C.c_int64),
        ("cu_share", C.c_int32), ("reserved0", C.c_int32),
    ]


class AidAttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p), ("out", C.c_void_p),
        ("coef", C.c_void_p), ("frame_scale", C.c_void_p), ("kv_map", C.c_void_p),
        ("k2", C.c_void_p), ("vt2", C.c_void_p),
        ("n_frames", C.c_int32), ("n_kv", C.c_int32),
        ("s", C.c_int32), ("l", C.c_int32), ("heads", C.c_int32), ("d", C.c_int32),
        ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldvt", C.c_int32), ("ldo", C.c_int32),
        ("q_fs", C.c_int64), ("k_fs", C.c_int64), ("vt_fs", C.c_int64), ("o_fs", C.c_int64),
        ("mode", C.c_int32), ("fused", C.c_int32), ("begin", C.c_int32), ("end", C.c_int32),
        ("accumulate", C.c_int32), ("dtype", C.c_int32),
        ("softmax_scale", C.c_float), ("out_scale", C.c_float),
        ("n_plain", C.c_int32), ("q_prescaled", C.c_int32),
        ("seg_executed", C.c_int32), ("reserved0", C.c_int32),
        ("bias", C.c_void_p), ("bias_fs", C.c_int64), ("bias_hs", C.c_int32), ("bias_rs", C.c_int32),
    ]


class AidProcessorArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ctx", C.c_void_p), ("wq", C.c_void_p), ("wk", C.c_void_p),
        ("wv", C.c_void_p), ("wo", C.c_void_p), ("bo", C.c_void_p), ("y", C.c_void_p),
        ("coef", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("n_frames", C.c_int32), ("s", C.c_int32), ("l", C.c_int32), ("c", C.c_int32),
        ("cc", C.c_int32), ("heads", C.c_int32), ("mode", C.c_int32), ("fused", C.c_int32),
        ("begin", C.c_int32), ("end", C.c_int32), ("dtype", C.c_int32), ("n_ctx", C.c_int32),
        ("ctx_map", C.c_void_p),
        ("n_plain", C.c_int32), ("ln_eps", C.c_float),
        ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("residual", C.c_void_p),
        ("ip", C.c_void_p), ("wk_ip", C.c_void_p), ("wv_ip", C.c_void_p), ("ip_map", C.c_void_p),
        ("ip_frame_scale", C.c_void_p), ("ip_stride", C.c_int64),
        ("n_ip", C.c_int32), ("t_ip", C.c_int32), ("ip_mode", C.c_int32), ("ip_scale", C.c_float),
        ("ip_begin", C.c_int32), ("ip_end", C.c_int32), ("seg_executed", C.c_int32), ("reserved2", C.c_int32),
        ("ln_wq", C.c_void_p), ("ln_wk", C.c_void_p), ("ln_wv", C.c_void_p), ("ln_const", C.c_void_p),
        ("k_cached", C.c_void_p), ("vt_cached", C.c_void_p),
        ("cu_share", C.c_int32), ("reserved1", C.c_int32),
        ("attn_bias", C.c_void_p), ("attn_bias_fs", C.c_int64), ("attn_bias_hs", C.c_int32), ("attn_bias_rs", C.c_int32),
    ]


class AidProfileEntry(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double),
                ("flops_executed", C.c_double)]


_lib: Optional[C.CDLL] = None


def build(verbose: bool = False) -> str:
    """Compile the gfx950 shared library in-tree with hipcc (cross-compiles without a GPU)."""
    proc = subprocess.run(["make", "-C", CSRC_DIR, "-j4"], capture_output=True, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout[-4000:])
        print(proc.stderr[-4000:])
    if proc.returncode != 0:
        raise RuntimeError("building libaid_hip.so failed (see output above)")
    return LIB_PATH


def load() -> C.CDLL:
    """Load libaid_hip.so; raise if it is absent (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            f"(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C {CSRC_DIR}`). "
            "This package has no CPU fallback.")
    _lib = bind(LIB_PATH)
    return _lib


def bind(path: str) -> C.CDLL:
    """dlopen one build of the library and declare the prototypes of include/aid_hip.h (development tools load several builds
    side by side, tools/dev/pp_variants.py)."""
    lib = C.CDLL(path)
    lib.aid_abi_version.restype = C.c_int
    lib.aid_strerror.restype = C.c_char_p
    lib.aid_strerror.argtypes = [C.c_int]
    lib.aid_last_attn_variant.restype = C.c_char_p
    lib.aid_last_gemm_variant.restype = C.c_char_p
    lib.aid_device_info.restype = C.c_int
    lib.aid_device_info.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p]
    lib.aid_gemm_nt.restype = C.c_int
    lib.aid_gemm_nt.argtypes = [C.POINTER(AidGemmProblem), C.c_int, C.c_int, C.c_void_p]
    lib.aid_layernorm.restype = C.c_int
    lib.aid_layernorm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float,
                                  C.c_int32, C.c_void_p]
    lib.aid_ln_stats.restype = C.c_int
    lib.aid_ln_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_int32, C.c_void_p]
    lib.aid_ln_fold.restype = C.c_int
    lib.aid_ln_fold.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                C.c_int32, C.c_void_p]
    lib.aid_attn_fwd.restype = C.c_int
    lib.aid_attn_fwd.argtypes = [C.POINTER(AidAttnArgs), C.c_void_p]
    lib.aid_lerp_kv.restype = C.c_int
    lib.aid_lerp_kv.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
    lib.aid_processor_workspace_bytes.restype = C.c_size_t
    lib.aid_processor_workspace_bytes.argtypes = [C.POINTER(AidProcessorArgs)]
    lib.aid_processor_fwd.restype = C.c_int
    lib.aid_processor_fwd.argtypes = [C.POINTER(AidProcessorArgs), C.c_void_p]
    lib.aid_set_tuning.restype = C.c_int
    lib.aid_set_tuning.argtypes = [C.c_char_p, C.c_int]
    lib.aid_get_tuning.restype = C.c_int
    lib.aid_get_tuning.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    lib.aid_profile_begin.restype = C.c_int
    lib.aid_profile_end.restype = C.c_int
    lib.aid_profile_end.argtypes = [C.POINTER(AidProfileEntry), C.c_int]
    lib.aid_stream_capture_id.restype = C.c_int
    lib.aid_stream_capture_id.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    if lib.aid_abi_version() != AID_ABI_VERSION:
        raise RuntimeError(f"libaid_hip.so ABI version {lib.aid_abi_version()} != expected {AID_ABI_VERSION}; rebuild")
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().aid_strerror(code).decode()
        raise RuntimeError(f"{what} failed with code {code}: {msg}")
