"""ctypes binding of librelpose_hip.so (declared in include/relpose_hip.h).

PyTorch is only plumbing here: it owns device memory and the HIP stream; every hot-path op goes through this
C ABI.  There is NO fallback: a missing library or a non-zero return code raises."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime this library binds to)

from . import _build

_LIB = None
HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "relpose_hip.h")


def _header_contract():
    """(RP_ABI_VERSION, RP_ABI_EXPORTS, declared function names) parsed from include/relpose_hip.h -- the single source of truth the
    library is compiled against and this binding is checked against."""
    import re
    with open(HEADER) as f:
        text = f.read()
    ver = int(re.search(r"#define\s+RP_ABI_VERSION\s+(\d+)", text).group(1))
    cnt = int(re.search(r"#define\s+RP_ABI_EXPORTS\s+(\d+)", text).group(1))
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return ver, cnt, set(re.findall(r"\b(rp_[a-z0-9_]+)\s*\(", code))


ABI_VERSION, ABI_EXPORTS, DECLARED = _header_contract()
RP_ERRORS = {-1: "bad shape", -2: "misaligned pointer/stride", -3: "workspace too small", -4: "unsupported"}


class RpGemm(Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
                ("M", c_int), ("N", c_int), ("K", c_int),
                ("lda", c_int), ("ldb", c_int), ("ldc", c_int),
                ("a_layout", c_int), ("b_layout", c_int), ("batch", c_int),
                ("stride_a", c_longlong), ("stride_b", c_longlong), ("stride_c", c_longlong),
                ("split_k", c_int), ("workspace", c_void_p), ("workspace_bytes", c_size_t),
                ("bias", c_void_p), ("pre_out", c_void_p), ("act", c_int), ("dact", c_int),
                ("aux", c_void_p), ("residual", c_void_p), ("trans_c", c_int), ("precision", c_int),
                ("colsum_part", c_void_p),
                ("ln_x", c_void_p), ("ln_mean", c_void_p), ("ln_rstd", c_void_p), ("ln_gamma", c_void_p), ("ln_part", c_void_p),
                ("ev_start", c_void_p), ("ev_stop", c_void_p), ("io_bf16", c_int), ("defer_reduce", c_int)]


class RpColsumTask(Structure):
    _fields_ = [("in_", c_void_p), ("rows", c_int), ("cols", c_int), ("ld", c_int), ("out", c_void_p)]


class RpBnMask(Structure):
    _fields_ = [("x", c_void_p), ("mean", c_void_p), ("rstd", c_void_p), ("gamma", c_void_p), ("beta", c_void_p)]


class RpSplitkTask(Structure):
    _fields_ = [("ws", c_void_p), ("C", c_void_p), ("M", c_int), ("N", c_int), ("ldc", c_int), ("split_k", c_int), ("trans_c", c_int)]


class RpTransposeTask(Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("rows", c_int), ("cols", c_int)]


RP_COLSUM_MAX = 8
RP_SPLITK_MAX = 8
RP_TRANSPOSE_MAX = 24
P, I, F, L = c_void_p, c_int, c_float, c_longlong
_SIGS = {
    "rp_abi_version": (c_int, []),
    "rp_abi_export_count": (c_int, []),
    "rp_target_arch": (c_char_p, []),
    "rp_gemm": (c_int, [POINTER(RpGemm), P]),
    "rp_gemm_workspace_bytes": (c_size_t, [I, I, I]),
    "rp_splitk_reduce_multi": (c_int, [POINTER(RpSplitkTask), I, P]),
    "rp_transpose_multi": (c_int, [POINTER(RpTransposeTask), I, P]),
    "rp_maxpool3x3s2_fwd": (c_int, [P, P, P, I, I, I, I, I, P]),
    "rp_maxpool3x3s2_bwd": (c_int, [P, P, P, I, I, I, I, I, P]),
    "rp_geodesic_loss": (c_int, [P, P, P, P, P, I, P]),
    "rp_bn_partial_blocks": (c_int, [L]),
    "rp_bn_stats": (c_int, [P, L, I, P, P, P, P, P, F, F, I, P]),
    "rp_bn_apply_fwd": (c_int, [P, P, P, P, P, P, P, L, I, I, I, P]),
    "rp_bn_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, P]),
    "rp_bn_bwd_from_partials": (c_int, [P, P, P, P, P, P, I, P, P, P, P, L, I, P]),
    "rp_layernorm_fwd": (c_int, [P, P, P, P, P, P, I, I, F, P]),
    "rp_layernorm_bwd_blocks": (c_int, [I]),
    "rp_layernorm_bwd": (c_int, [P, P, P, P, P, P, P, P, P, I, I, P]),
    "rp_colsum_workspace_bytes": (c_size_t, [I, I]),
    "rp_colsum": (c_int, [P, I, I, I, P, P, c_size_t, P]),
    "rp_colsum_multi_workspace_bytes": (c_size_t, [POINTER(RpColsumTask), I]),
    "rp_colsum_multi": (c_int, [POINTER(RpColsumTask), I, P, c_size_t, P]),
    "rp_preprocess": (c_int, [P, P, I, I, I, P]),
    "rp_tokens_fwd": (c_int, [P, P, P, I, I, I, P]),
    "rp_tokens_fwd_nhwc": (c_int, [P, P, P, I, I, I, P]),
    "rp_tokens_bwd": (c_int, [P, P, I, I, I, P]),
    "rp_attn_fwd": (c_int, [P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, I, P]),
    "rp_attn_fwd_bf16": (c_int, [P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, P]),
    "rp_attn_bwd_delta_bf16": (c_int, [P, P, P, I, I, I, P]),
    "rp_attn_bwd_bf16": (c_int, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, P, P, P, I, P]),
    "rp_attn_bwd_delta": (c_int, [P, P, P, I, I, I, P]),
    "rp_attn_bwd": (c_int, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, P]),
    "rp_attn_bwd_dkdv_ds": (c_int, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, P, P, I, P]),
    "rp_emm_stats_workspace_bytes": (ctypes.c_size_t, [I, I]),
    "rp_emm_stats": (c_int, [P, P, P, P, P, P, I, I, I, I, F, I, P]),
    "rp_ds_matmul": (c_int, [P, P, P, I, I, I, I, I, I, P, I, P]),
    "rp_attn_bwd_cross": (c_int, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, I, P]),
    "rp_attn_bwd_dkdv": (c_int, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, P]),
    "rp_attn_bwd_dq": (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, I, I, F, I, P]),
    "rp_attn_fwd_savep": (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, I, F, P]),
    "rp_attn_bwd_dkdv_p": (c_int, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, F, P, P, I, P]),
    "rp_ds_matmul_t": (c_int, [P, P, P, I, I, I, I, I, P, I, P]),
    "rp_dw192_f32_splits": (c_int, [I, I]),
    "rp_dw192_f32_workspace_bytes": (c_size_t, [I, I]),
    "rp_dw192_f32": (c_int, [P, I, P, I, I, P, c_size_t, P]),
    "rp_dw192_split3": (c_int, [P, I, P, I, I, P, c_size_t, P]),
    "rp_dw192_bf16_splits": (c_int, [I, I]),
    "rp_dw192_bf16_workspace_bytes": (c_size_t, [I, I]),
    "rp_dw192_bf16": (c_int, [P, I, P, I, I, I, P, c_size_t, P]),
    "rp_dx_lnbwd_bf16_tile_rows": (c_int, []),
    "rp_dx_lnbwd_bf16": (c_int, [P, P, P, P, P, P, P, P, P, I, I, P]),
    "rp_emm_finalize_parts": (c_int, [P, P, I, I, I, I, P]),
    "rp_emm_build_x_bf16": (c_int, [P, P, P, I, I, I, P]),
    "rp_emm_apply_bf16": (c_int, [P, I, P, P, P, P, I, I, F, I, P]),
    "rp_emm_f_bf16": (c_int, [P, P, P, I, I, P]),
    "rp_emm_w_bf16": (c_int, [P, P, P, P, P, P, I, I, P]),
    "rp_emm_dx_bf16": (c_int, [P, P, P, P, P, I, P, I, I, P]),
    "rp_emm_grad_bf16": (c_int, [P, I, P, P, P, P, P, P, P, I, I, F, I, P]),
    "rp_posenc": (c_int, [P, P, P, I, I, P]),
    "rp_emm_build_x": (c_int, [P, P, P, I, I, I, P]),
    "rp_emm_build_x_bwd": (c_int, [P, P, I, I, I, P]),
    "rp_emm_apply": (c_int, [P, I, P, P, P, P, P, P, P, I, I, F, I, I, I, P]),
    "rp_emm_finalize": (c_int, [P, P, I, I, I, P]),
    "rp_emm_finalize_bwd": (c_int, [P, P, I, I, I, P]),
    "rp_rowdot96": (c_int, [P, P, P, L, P]),
    "rp_emm_grad": (c_int, [P, I, P, P, P, P, P, P, P, I, I, F, I, I, I, P]),
    "rp_emm_grad_ds": (c_int, [P, I, P, P, P, P, P, P, P, P, P, I, I, F, I, I, P]),
    "rp_pose_normalize_fwd": (c_int, [P, P, P, I, P]),
    "rp_pose_normalize_bwd": (c_int, [P, P, P, I, P]),
    "rp_preprocess_padded": (c_int, [P, P, I, I, I, I, P]),
    "rp_conv_stem_blocks": (c_int, [I, I, I]),
    "rp_conv_stem_fwd": (c_int, [P, P, P, P, I, I, I, P]),
    "rp_conv_stem_bf16_blocks": (c_int, [I, I, I]),
    "rp_conv_stem_fwd_bf16": (c_int, [P, P, P, P, I, I, I, P]),
    "rp_conv_stem_wgrad_workspace_bytes": (c_size_t, [I]),
    "rp_conv_stem_wgrad_bf16": (c_int, [P, P, P, P, c_size_t, I, I, I, P]),
    "rp_conv_stem_wgrad_f32_workspace_bytes": (c_size_t, [I]),
    "rp_conv_stem_wgrad_f32": (c_int, [P, P, P, P, c_size_t, I, I, I, P]),
    "rp_conv3x3_c64_blocks": (c_int, [I]),
    "rp_conv3x3_c64_bf16": (c_int, [P, P, P, P, P, P, I, I, I, P]),
    "rp_conv3x3_c64_wgrad_blocks": (c_int, [I]),
    "rp_conv3x3_c64_wgrad_workspace_bytes": (c_size_t, [I]),
    "rp_conv3x3_c64_wgrad_bf16": (c_int, [P, P, P, P, c_size_t, I, I, I, P]),
    "rp_conv3x3_c64_wgrad_f32_blocks": (c_int, [I]),
    "rp_conv3x3_c64_wgrad_f32_workspace_bytes": (c_size_t, [I]),
    "rp_conv3x3_c64_wgrad_f32": (c_int, [P, P, P, P, c_size_t, I, I, I, P]),
    "rp_conv3x3_c64_f32_blocks": (c_int, [I]),
    "rp_conv3x3_c64_f32": (c_int, [P, P, P, P, P, POINTER(RpBnMask), I, I, I, I, P]),
    "rp_conv3x3_c128_f32_blocks": (c_int, [I, I]),
    "rp_conv3x3_c128_f32": (c_int, [P, P, P, P, P, I, I, I, I, I, P]),
    "rp_bn_stats_from_partials": (c_int, [P, I, L, I, P, P, P, P, P, F, F, P]),
    "rp_bn_relu_pool_fwd": (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "rp_bn_relu_pool_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "rp_event_create": (c_void_p, []),
    "rp_event_destroy": (None, [P]),
    "rp_event_elapsed_ms": (c_float, [P, P]),
    "rp_linear_rows192_tile_rows": (c_int, []),
    "rp_linear_rows192": (c_int, [P, P, P, P, P, P, F, P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "rp_mlp_fused_workspace_bytes": (ctypes.c_size_t, [I]),
    "rp_mlp_fused_fwd": (c_int, [P, P, P, P, P, P, P, P, P, I, I, I, ctypes.c_float, P, P, P, P, P, I, I, P]),
    "rp_mlp_fused_bwd_workspace_bytes": (ctypes.c_size_t, [I]),
    "rp_mlp_fused_bwd_tile_rows": (c_int, []),
    "rp_mlp_fused_bwd": (c_int, [P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "rp_mlp_fused_bwd_ln_part_rows": (c_int, [I]),
    "rp_mlp_fused_bwd_ln": (c_int, [P, P, P, P, P, P, P, P, I, I, I, I, I, P, P, P, P, P, P]),
    "rp_augment_blocks": (c_int, []),
    "rp_augment_pairs": (c_int, [P, P, P, P, I, I, I, I, I, P]),
    "rp_essential_from_pose": (c_int, [P, P, I, P]),
    "rp_svd3x3": (c_int, [P, P, P, P, I, P]),
    "rp_pose_from_essential": (c_int, [P, P, P, I, P, P, I, P]),
}
EXPORTS = tuple(_SIGS)


def lib_path():
    return _build.LIB


def load():
    """Load (building if the .so is absent) and type the C ABI.  Raises on any failure."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB
    if _build.needs_build():           # missing, or older than a kernel source / the header (never a silently stale .so)
        _build.build(verbose=False)
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise RuntimeError("rel_pose_amd: cannot load HIP extension %s (%s); there is no CPU fallback" % (path, e))
    lib.rp_abi_version.restype = c_int
    if lib.rp_abi_version() != ABI_VERSION:
        raise RuntimeError("rel_pose_amd: %s has ABI version %d, this package binds version %d -- rebuild with "
                           "`python -m rel_pose_amd._build --force`" % (path, lib.rp_abi_version(), ABI_VERSION))
    if set(_SIGS) != DECLARED or len(DECLARED) != ABI_EXPORTS:
        raise RuntimeError("rel_pose_amd: include/relpose_hip.h declares %d entry points (RP_ABI_EXPORTS = %d) but _lib.py binds %d: "
                           "%s" % (len(DECLARED), ABI_EXPORTS, len(_SIGS), sorted(DECLARED ^ set(_SIGS))))
    lib.rp_abi_export_count.restype = c_int
    if lib.rp_abi_export_count() != ABI_EXPORTS:
        raise RuntimeError("rel_pose_amd: %s was compiled with %d entry points, the header declares %d -- rebuild"
                           % (path, lib.rp_abi_export_count(), ABI_EXPORTS))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError = symbol missing = broken build
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc, what):
    if rc != 0:
        if rc < 0:
            raise RuntimeError("rel_pose_amd: %s failed: %s (RP error %d)" % (what, RP_ERRORS.get(rc, "?"), rc))
        raise RuntimeError("rel_pose_amd: %s failed: hipError %d" % (what, rc))
