"""ctypes binding of libprisma_bands.so (include/prisma_bands.h).

The library is the product; there is no Python/torch/numpy fallback.  If the shared object is
missing or was built without its kernels this module raises at import of the symbols, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRISMA_BANDS_LIB") or os.path.join(_HERE, "libprisma_bands.so")      # the override is for A/B timing of two builds


class pb_tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 6), ("data", C.c_void_p)]


class pb_depth_cfg(C.Structure):
    _fields_ = [("embed_dim", C.c_int32), ("depth", C.c_int32), ("heads", C.c_int32), ("features", C.c_int32),
                ("out_channels", C.c_int32 * 4), ("pos_grid", C.c_int32), ("max_batch", C.c_int32), ("metric", C.c_int32),
                ("precision", C.c_int32)]


class pb_flow_cfg(C.Structure):
    _fields_ = [("precision", C.c_int32)]


PREC_F16, PREC_SPLIT = 0, 1
ABI_VERSION = 3          # include/prisma_bands.h PB_ABI_VERSION this binding was written against


def default_precision() -> int:
    """pb_precision the engine classes use when none is given: PRISMA_PRECISION=0 selects the single-pass fp16 mode
    (faster; max-norm error up to 1.6e-3 against the fp32 reference), anything else / unset the split-fp16 mode that
    meets the 1e-3 bound the parity tests assert."""
    return PREC_F16 if os.environ.get("PRISMA_PRECISION", "1") == "0" else PREC_SPLIT


class pb_mask_cfg(C.Structure):
    _fields_ = [("blocks", C.c_int32 * 4), ("scale_long", C.c_int32), ("scale_short", C.c_int32), ("num_classes", C.c_int32),
                ("feat_channels", C.c_int32), ("stacked_convs", C.c_int32), ("num_grids", C.c_int32 * 5),
                ("strides", C.c_int32 * 5), ("mask_feat_channels", C.c_int32), ("mask_out_channels", C.c_int32),
                ("nms_pre", C.c_int32), ("max_per_img", C.c_int32), ("score_thr", C.c_float), ("mask_thr", C.c_float),
                ("filter_thr", C.c_float), ("sigma", C.c_float), ("max_batch", C.c_int32), ("precision", C.c_int32)]


class pb_kernel_stat(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_double), ("flops", C.c_double), ("exec_flops", C.c_double), ("bytes", C.c_double),
                ("launches", C.c_int32)]


# every symbol include/prisma_bands.h declares: (restype, argtypes)
_P = C.c_void_p
_F = C.POINTER(C.c_float)
_U8 = C.POINTER(C.c_uint8)
SYMBOLS = {
    "pb_last_error": (C.c_char_p, []),
    "pb_version": (C.c_int, []),
    "pb_abi_version": (C.c_int, []),
    "pb_struct_size": (C.c_int, [C.c_int]),
    "pb_device_count": (C.c_int, []),
    "pb_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_char_p, C.POINTER(pb_tensor), C.c_int, _P, C.c_size_t]),
    "pb_destroy": (None, [_P]),
    "pb_depth_infer_batch": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int]),
    "pb_depth_infer_batch_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int]),
    "pb_sync": (C.c_int, [_P]),
    "pb_depth_submit_batch": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int]),
    "pb_flow_submit_sequence": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _P, _P, _P]),
    "pb_wait": (C.c_int, [_P]),
    "pb_comm_unique_id": (C.c_int, [_P]),
    "pb_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "pb_gather_scalars": (C.c_int, [_P, _P, C.c_int, _P]),
    "pb_depth_encode_still": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _F, _F]),
    "pb_depth_net_size": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pb_depth_get_stage": (C.c_int64, [_P, C.c_char_p, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "pb_mask_infer_batch": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P, C.c_int, _P]),
    "pb_mask_infer_batch_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P, C.c_int, _P]),
    "pb_mask_get_instances": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "pb_mask_net_size": (C.c_int, [C.POINTER(pb_mask_cfg), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pb_mask_get_stage": (C.c_int64, [_P, C.c_char_p, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "pb_flow_set_inference_size": (C.c_int, [_P, C.c_int, C.c_int]),
    "pb_mask_set_sdf": (C.c_int, [_P, _P, _P, C.c_int]),
    "pb_mask_sdf_green": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int]),
    "pb_mask_sdf_green_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int]),
    "pb_flow_out_size": (C.c_int, [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pb_flow_infer_sequence": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _P, _P, _P]),
    "pb_flow_infer_sequence_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _P, _P, _P]),
    "pb_flow_infer_sequence_masks": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float,
                                               _P, _P, _P, _P]),
    "pb_flow_infer_sequence_masks_dev": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float,
                                                   C.c_float, _P, _P, _P, _P]),
    "pb_flow_fwdbwd_mask": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P]),
    "pb_flow_get_stage": (C.c_int64, [_P, C.c_char_p, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "pb_dev_alloc": (C.c_int, [_P, C.POINTER(_P), C.c_size_t]),
    "pb_dev_free": (C.c_int, [_P, _P]),
    "pb_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "pb_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "pb_set_profiling": (C.c_int, [_P, C.c_int]),
    "pb_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "pb_get_kernel_stats": (C.c_int, [_P, C.POINTER(pb_kernel_stat), C.c_int]),
    "pb_op_gemm": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pb_op_gemm_bench": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "pb_op_corr_volume": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P]),
    "pb_op_attention_bench": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "pb_op_layernorm": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int]),
    "pb_op_attention": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int]),
    "pb_op_attention128": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int]),
    "pb_op_attention128_split": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pb_op_conv2d": (C.c_int, [_P, _P, _P, _P, _P] + [C.c_int] * 10),
    "pb_op_bilinear": (C.c_int, [_P, _P, _P] + [C.c_int] * 7),
    "pb_op_preprocess": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int]),
    "pb_op_encode_depth": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
}

_lib = None


class PrismaBandsError(RuntimeError):
    pass


def load():
    """Load the shared library and bind every declared symbol (raises if any is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PrismaBandsError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C prisma_amd/csrc).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.pb_abi_version() != ABI_VERSION:
        raise PrismaBandsError(f"{LIB_PATH}: ABI version {lib.pb_abi_version()}, this binding expects {ABI_VERSION} - rebuild the library "
                               "(make -C prisma_amd/csrc) or update prisma_amd/_lib.py")
    for which, st in enumerate((pb_tensor, pb_depth_cfg, pb_flow_cfg, pb_mask_cfg, pb_kernel_stat)):
        if lib.pb_struct_size(which) != C.sizeof(st):
            raise PrismaBandsError(f"{LIB_PATH}: sizeof({st.__name__}) is {lib.pb_struct_size(which)} in the library, {C.sizeof(st)} in the binding")
    _lib = lib
    return lib


def check(rc: int):
    if rc < 0:
        raise PrismaBandsError(f"libprisma_bands: {load().pb_last_error().decode()} (status {rc})")
    return rc
