"""ctypes binding of libapplecider_hip.so (C ABI declared in include/applecider_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or a
kernel is asked to run on a non-GPU tensor, this module raises.  Only the
in-tree build (``applecider_amd/csrc/libapplecider_hip.so``) is ever loaded.
"""

from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("APPLECIDER_HIP_LIB") or os.path.join(_HERE, "csrc", "libapplecider_hip.so")   # override: another build of the same ABI
# the same sources built with IEEE fp16 as the 16-bit operand format (csrc/Makefile, ac_common.h):
# the inference library behind hipops.set_math("f16") (BASELINE configs[4])
LIB_PATH_F16 = os.path.join(_HERE, "csrc", "libapplecider_hip_f16.so")

AC_GEMM_NT, AC_GEMM_NN, AC_GEMM_TN = 0, 1, 2
AC_EINVAL = -22
ACT_NONE, ACT_GELU, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4
MATH_F32, MATH_BF16, MATH_BF16_IN, MATH_BF16X3 = 0, 1, 2, 3
ACT_CODES = {None: ACT_NONE, "none": ACT_NONE, "gelu": ACT_GELU, "relu": ACT_RELU,
             "sigmoid": ACT_SIGMOID, "tanh": ACT_TANH}


class RowMap(C.Structure):
    _fields_ = [("r1", C.c_int32), ("r2", C.c_int32), ("s1", C.c_int64), ("s2", C.c_int64),
                ("s3", C.c_int64)]


class Mat(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("rows", RowMap), ("goff", C.c_void_p)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("math", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("act", C.c_int32), ("dact", C.c_int32), ("accumulate", C.c_int32),
        ("split_k", C.c_int32), ("force_simple", C.c_int32),
        ("alpha", C.c_float), ("tile", C.c_int32),
        ("a", Mat), ("b", Mat), ("c", Mat),
        ("bias", C.c_void_p), ("pre_out", C.c_void_p), ("ld_pre", C.c_int64),
        ("aux", C.c_void_p), ("ld_aux", C.c_int64),
        ("colscale", C.c_void_p), ("residual", C.c_void_p), ("ld_res", C.c_int64),
        ("c16", C.c_void_p), ("ld_c16", C.c_int64),
        ("mask16", C.c_void_p), ("ld_mask16", C.c_int64),
        ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("drop_step", C.c_void_p),
        ("b_hi", C.c_void_p), ("b_lo", C.c_void_p), ("ld_bpl", C.c_int64), ("colsum", C.c_void_p),
    ]


class ConvWinDesc(C.Structure):
    _fields_ = [("a", C.c_void_p), ("a_batch_stride", C.c_int64), ("a_row_stride", C.c_int64),
                ("a_col_off", C.c_int32), ("row_base", C.c_int32),
                ("B", C.c_int32), ("L", C.c_int32), ("C", C.c_int32), ("k", C.c_int32),
                ("w", C.c_void_p), ("w_row_stride", C.c_int64), ("w_tap_stride", C.c_int64),
                ("flip", C.c_int32), ("N", C.c_int32), ("c", C.c_void_p), ("ldc", C.c_int64),
                ("bias", C.c_void_p), ("accumulate", C.c_int32), ("variant", C.c_int32),
                ("c16", C.c_void_p), ("ldc16", C.c_int64), ("a_lo_off", C.c_int64), ("w_lo_off", C.c_int64),
                ("tap_row_step", C.c_int32), ("c_block", C.c_int32), ("c_block_stride", C.c_int64)]


class WgradDesc(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("dy_batch_stride", C.c_int64), ("dy_row_stride", C.c_int64),
                ("dy_row_base", C.c_int32), ("dy_col_off", C.c_int32),
                ("x", C.c_void_p), ("x_batch_stride", C.c_int64), ("x_row_stride", C.c_int64),
                ("x_row_base", C.c_int32), ("x_rows", C.c_int32),
                ("B", C.c_int32), ("L", C.c_int32), ("Cout", C.c_int32), ("Cin", C.c_int32), ("k", C.c_int32),
                ("split_k", C.c_int32), ("dw", C.c_void_p), ("ldw", C.c_int64),
                ("dy_lo_off", C.c_int64), ("x_lo_off", C.c_int64), ("variant", C.c_int32),
                ("tap_row_step", C.c_int32), ("dy_block", C.c_int32), ("dy_block_stride", C.c_int64)]


class FftRowsDesc(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("rows_lo", C.c_void_p), ("spec", C.c_void_p), ("tw", C.c_void_p),
                ("bias", C.c_void_p), ("batch_stride", C.c_int64), ("row_stride", C.c_int64),
                ("col_off", C.c_int32), ("B", C.c_int32), ("L", C.c_int32), ("C", C.c_int32), ("logn", C.c_int32),
                ("radix3", C.c_int32), ("blocks", C.c_int32), ("block_step", C.c_int32), ("shift", C.c_int32),
                ("n_lo", C.c_int32), ("n_hi", C.c_int32), ("accumulate", C.c_int32), ("lds_exact", C.c_int32)]


class TowerDesc(C.Structure):
    _fields_ = ([(n, C.c_void_p) for n in
                 ("x", "w1", "b1", "lnm_g", "lnm_b", "lng_g", "lng_b", "wm", "bm", "wg", "bg", "ws", "bs", "y", "save",
                  "dy", "dx", "dw1", "db1", "dlnm_g", "dlnm_b", "dlng_g", "dlng_b", "dwm", "dbm", "dwg", "dbg",
                  "dws", "dbs")] +
                [(n, C.c_int64) for n in ("ldx", "ldy", "lddy", "lddx")] +
                [(n, C.c_int32) for n in ("n_in", "hid", "n_out", "gather", "group_id")] +
                [("eps", C.c_float), ("cols", C.c_uint8 * 24)])


class AdamSeg(C.Structure):
    _fields_ = [("begin", C.c_int64), ("end", C.c_int64), ("lr", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("decoupled", C.c_int32)]


_P, _I32, _I64, _F, _U64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64

# name -> argtypes (return type is int unless listed in _RESTYPES)
SIGNATURES = {
    "ac_abi_version": [],
    "ac_strerror": [_I32],
    "ac_gemm": [C.POINTER(GemmDesc), _P],
    "ac_conv1d_window_bf16": [C.POINTER(ConvWinDesc), _P],
    "ac_cast_bf16": [_P, _P, _I64, _P],
    "ac_transpose_cast_bf16": [_P, _I64, _P, _I64, _I64, _I32, _P],
    "ac_transpose_cast_segments": [_P, _P, _P, _I32, _I32, _P],
    "ac_layernorm_fwd": [_P, _I64, _P, _P, _P, _I64, _P, _P, _I64, _I32, _F, _I32, _P, _I64, _I32, _P],
    "ac_layernorm_bwd": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _I64, _I32, _I32,
                         _P, _I64, _I32, _I32, _I32, _I32, _I32, _P],
    "ac_layernorm_bwd_split": [_P, _I64, _P, _I64, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _I64, _I32, _I32,
                               _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P],
    "ac_mpt_mask": [_P, _P, _P, _I32, _I32, C.c_double, C.c_uint64, _P, _P],
    "ac_mpt_loss_fwd_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _F, _F, _F, _P],
    "ac_colsum": [_P, _I64, _P, _I64, _I32, _I32, _P],
    "ac_colsum_bf16": [_P, _I64, _P, _I64, _I32, _I32, _P],
    "ac_cast_bf16_colsum": [_P, _I64, _P, _I64, _P, _I64, _I32, _I32, _P],
    "ac_act_bwd": [_P, _P, _P, _I64, _I32, _P],
    "ac_act_fwd": [_P, _P, _I64, _I32, _P],
    "ac_copy2d": [_P, _I64, _P, _I64, _I64, _I32, _P],
    "ac_scale_add_rows": [_P, _P, _P, _P, _I64, _I32, _P],
    "ac_splitk_reduce": [_P, _I32, _P, _P, _P, _P, _P, _I64, _I32, _P],
    "ac_gather_cols": [_P, _I64, _P, _P, _I64, _I64, _I32, _P],
    "ac_gate_fwd": [_P, _P, _P, _P, _I64, _P],
    "ac_gate_bwd": [_P, _P, _P, _P, _P, _I64, _P],
    "ac_dropout": [_P, _P, _I64, _F, _U64, _U64, _P, _P],
    "ac_layerscale_bwd": [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _P],
    "ac_add": [_P, _P, _P, _I64, _F, _P],
    "ac_scale_by_dev": [_P, _I64, _P, _P],
    "ac_stem_patchify": [_P, _P, _I32, _I32, _I32, _I32, _P],
    "ac_dwconv7x7_fwd": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _P],
    "ac_dwconv7x7_bwd": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P],
    "ac_dwconv7x7_fwd_v": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    "ac_dwconv7x7_bwd_v": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    "ac_dwconv7x7_bwd_res": [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    "ac_avgpool_fwd": [_P, _P, _I32, _I32, _I32, _P],
    "ac_avgpool_bwd": [_P, _P, _I32, _I32, _I32, _P],
    "ac_maxpool4_fwd": [_P, _P, _I64, _P, _I32, _I32, _I32, _P],
    "ac_maxpool4_bwd": [_P, _I64, _P, _P, _I32, _I32, _I32, _P],
    "ac_globalmax_fwd": [_P, _P, _P, _I32, _I32, _I32, _P],
    "ac_globalmax_bwd": [_P, _P, _P, _I32, _I32, _I32, _P],
    "ac_pad_rows": [_P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    "ac_pad_rows_bf16": [_P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    "ac_toeplitz_expand": [_P, _P, _I32, _I32, _I32, _I32, _P],
    "ac_toeplitz_fold": [_P, _P, _I32, _I32, _I32, _I32, _P],
    "ac_embed_fwd": [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P],
    "ac_embed_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P],
    "ac_mha_fwd": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _F, _U64, _P, _P],
    "ac_mha_bwd": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _F, _U64, _P, _P],
    "ac_batchnorm_fwd": [_P, _I64, _P, _P, _P, _P, _P, _I64, _P, _P, _I64, _I32, _F, _F, _I32, _I32, _P],
    "ac_batchnorm_bwd": [_P, _I64, _P, _I64, _P, _P, _P, _I64, _P, _P, _P, _I64, _I32, _I32, _I32, _P],
    "ac_conv1d_window_x3": [C.POINTER(ConvWinDesc), _P],
    "ac_conv1d_wgrad_bf16": [C.POINTER(WgradDesc), _P],
    "ac_mha_fwd_mfma": [_P, _P, _P, _P, _I32, _I32, _I32, _I32, _F, _U64, _P, _I32, _P],
    "ac_mha_bwd_mfma": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _F, _U64, _P, _I32, _P],
    "ac_split_bf16": [_P, _P, _P, _I64, _P],
    "ac_transpose_split_bf16": [_P, _I64, _P, _P, _I64, _I64, _I32, _P],
    "ac_pad_rows_split": [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P],
    "ac_moe_top2_fwd": [_P, _P, _P, _P, _I32, _I32, _I32, _P],
    "ac_moe_top2_bwd": [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P],
    "ac_l2norm_fwd": [_P, _P, _P, _I64, _I32, _P],
    "ac_l2norm_bwd": [_P, _P, _P, _P, _I64, _I32, _P],
    "ac_softmax_fwd": [_P, _P, _I64, _I32, _P],
    "ac_loss_fwd_bwd": [_P, _P, _P, _P, _P, _I32, _I32, _I32, _F, _F, _P],
    "ac_adam_flat": [_P, _P, _P, _P, C.POINTER(AdamSeg), _I32, _I32, _P, _P],
    "ac_adam_flat_dev": [_P, _P, _P, _P, C.POINTER(AdamSeg), _I32, _P, _P, _P],
    "ac_step_advance": [_P, _P],
    "ac_sgd_flat": [_P, _P, _P, _I64, _F, _F, _F, _I32, _P],
    "ac_act_bwd_colsum": [_P, _P, _P, _I64, _P, _I64, _I32, _I32, _I32, _F, _U64, _P, _P],
    "ac_tower_blocks_fwd": [C.POINTER(TowerDesc), _I32, _I32, _F, _I32, _U64, _P, _P],
    "ac_tower_blocks_bwd": [C.POINTER(TowerDesc), _I32, _I32, _F, _I32, _U64, _P, _P],
    "ac_sumsq": [_P, _I64, _P, _P],
    "ac_clip_coef": [_P, _F, _P, _P],
    "ac_ceil_copy": [_P, _P, _I64, _P],
    "ac_ceil_mfma": [_P, _P, _I32, _I32, _I32, _I32, _P],
    "ac_gemm_batched": [C.POINTER(GemmDesc), _I32, _I64, _I64, _I64, _P],
    "ac_gemm_grouped": [C.POINTER(GemmDesc), _I32, C.POINTER(C.c_void_p), _P],
    "ac_collate_photometry": [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P],
    "ac_add_segments": [_P, _P, _P, _P, _P, _I32, _I32, _P],
    "ac_spectail_supported": [_I64, _I32, _I32],
    "ac_spectail_fwd": [_P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P],
    "ac_spectail_bwd_dx": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _I64, _I32, _I32, _P],
    "ac_spectail_bwd_dw": [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _P],
    "ac_fft_rows_fwd": [C.POINTER(FftRowsDesc), _P],
    "ac_fft_rows_inv": [C.POINTER(FftRowsDesc), _P],
    "ac_fft_taps_fwd": [_P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P],
    "ac_fft_taps_inv": [_P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P],
}
ABI_VERSION = 5
_RESTYPES = {"ac_strerror": C.c_char_p}

_lib = None
_libs: dict = {}
_variant = "bf16"


class HipLibraryMissing(RuntimeError):
    pass


def _load_path(path: str):
    if not os.path.exists(path):
        raise HipLibraryMissing(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  applecider_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    if lib.ac_abi_version() != ABI_VERSION:
        raise HipLibraryMissing(f"ABI version mismatch: {lib.ac_abi_version()} != {ABI_VERSION} (rebuild: "
                                "make -C applecider_amd/csrc)")
    return lib


def select(variant: str):
    """Which build of the library `load()` hands out: "bf16" (default; training and inference) or
    "f16" (IEEE fp16 operands, inference only).  hipops.set_math switches it."""
    global _variant
    if variant not in ("bf16", "f16"):
        raise ValueError(variant)
    _variant = variant


def load():
    """The in-tree shared library of the selected operand format (loaded once each).  Raises
    HipLibraryMissing loudly."""
    lib = _libs.get(_variant)
    if lib is None:
        lib = _libs[_variant] = _load_path(LIB_PATH_F16 if _variant == "f16" else LIB_PATH)
    global _lib
    _lib = lib
    return lib


def loaded():
    """Every library variant loaded so far."""
    return list(_libs.values())


def strerror(code: int) -> str:
    return load().ac_strerror(code).decode()


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: code {rc} ({strerror(rc)})")
