"""ctypes binding of libmtgv.so (include/mtgv.h).

There is no fallback: if the HIP library is missing or a call fails, this raises.
torch is imported first on purpose - its bundled HIP runtime (soname
libamdhip64.so.7) must already be in the process so that libmtgv.so binds to the
same runtime and torch's streams / device pointers are valid inside it.
"""

from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MTGV_LIB_PATH", os.path.join(_HERE, "libmtgv.so"))  # override: experiments with alternative builds

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
c_fp = C.POINTER(C.c_float)


class EncoderCfg(C.Structure):
    _fields_ = [
        ("kind", c_i32),
        ("image_h", c_i32),
        ("image_w", c_i32),
        ("in_chans", c_i32),
        ("z_size", c_i32),
        ("depths", c_i32 * 4),
        ("dims", c_i32 * 4),
        ("head_type", c_i32),
        ("scale_io", c_i32),
        ("max_batch", c_i32),
    ]


class DetectorCfg(C.Structure):
    _fields_ = [
        ("nc", c_i32),
        ("imgsz", c_i32),
        ("max_batch", c_i32),
        ("conf", c_f32),
        ("iou", c_f32),
        ("max_det", c_i32),
        ("arch", c_i32),
    ]


# name -> (restype, argtypes); every symbol include/mtgv.h declares
SIGNATURES = {
    "mtgv_last_error": (C.c_char_p, []),
    "mtgv_version": (C.c_int, []),
    "mtgv_device_count": (C.c_int, []),
    "mtgv_set_gemm_precision": (C.c_int, [c_i32]),
    "mtgv_get_gemm_precision": (C.c_int, [C.POINTER(c_i32)]),
    "mtgv_bank_topk_packed": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64, c_vp, c_vp]),
    "mtgv_topk_merge_gathered": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp]),
    "mtgv_letterbox_u8": (C.c_int, [c_vp, c_i32, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "mtgv_profile_gemm": (C.c_int, [c_i32]),
    "mtgv_profile_gemm_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_i64)]),
    "mtgv_profile_gemm_bytes": (C.c_int, [C.POINTER(C.c_double)]),
    "mtgv_profile_gemm_dump": (C.c_int, [C.c_char_p]),
    "mtgv_encoder_create": (C.c_int, [C.POINTER(EncoderCfg), C.POINTER(c_vp)]),
    "mtgv_encoder_destroy": (None, [c_vp]),
    "mtgv_encoder_set_param": (C.c_int, [c_vp, C.c_char_p, c_vp, c_i64]),
    "mtgv_encoder_missing_params": (C.c_int, [c_vp]),
    "mtgv_encoder_forward": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "mtgv_encoder_set_graph": (C.c_int, [c_vp, c_i32, c_i32]),
    "mtgv_encoder_set_capture": (C.c_int, [c_vp, c_i32]),
    "mtgv_encoder_stage_output": (C.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "mtgv_encoder_flops": (C.c_int, [c_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mtgv_bank_create": (C.c_int, [c_i32, c_i64, C.POINTER(c_vp)]),
    "mtgv_bank_destroy": (None, [c_vp]),
    "mtgv_bank_size": (c_i64, [c_vp]),
    "mtgv_bank_append": (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    "mtgv_bank_set_row": (C.c_int, [c_vp, c_i64, c_vp, c_vp]),
    "mtgv_bank_clear": (C.c_int, [c_vp]),
    "mtgv_bank_get_rows": (C.c_int, [c_vp, c_i64, c_i64, c_vp]),
    "mtgv_bank_topk": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64, c_f32, c_vp, c_vp, c_vp]),
    "mtgv_bank_prepass_fallbacks": (C.c_int, [c_vp, C.POINTER(c_i64)]),
    "mtgv_topk_merge": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp]),
    "mtgv_detector_create": (C.c_int, [C.POINTER(DetectorCfg), C.POINTER(c_vp)]),
    "mtgv_detector_destroy": (None, [c_vp]),
    "mtgv_detector_set_param": (C.c_int, [c_vp, C.c_char_p, c_vp, c_i64]),
    "mtgv_detector_missing_params": (C.c_int, [c_vp]),
    "mtgv_detector_finalize": (C.c_int, [c_vp]),
    "mtgv_detector_forward": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "mtgv_detector_raw": (C.c_int, [c_vp, c_i32, c_vp, c_vp, c_vp]),
    "mtgv_detector_flops": (C.c_int, [c_vp, C.POINTER(C.c_double)]),
    "mtgv_detector_set_fork": (C.c_int, [c_vp, c_i32]),
    "mtgv_mask_binarize": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mtgv_nms": (
        C.c_int,
        [c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_size_t, c_vp],
    ),
    "mtgv_nms_workspace_bytes": (C.c_size_t, [c_i32, c_i32]),
    "mtgv_select_cards": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "mtgv_warp_workspace_bytes": (C.c_size_t, [c_i32]),
    "mtgv_warp_quads": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_i32, c_i32, C.c_double, c_vp, c_vp, C.c_size_t, c_vp]),
    "mtgv_mask_quads": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mtgv_mask_quads_logits": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mtgv_make_cropped": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "mtgv_op_linear": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "mtgv_op_linear_ex_part_floats": (c_i64, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "mtgv_op_last_grn_layout": (C.c_int, [c_vp, c_vp]),
    "mtgv_op_linear_ex": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "mtgv_op_conv2d": (C.c_int, [c_vp, c_vp, c_vp, c_vp] + [c_i32] * 10 + [c_vp]),
    "mtgv_op_layernorm": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_vp]),
    "mtgv_op_dwconv7": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "mtgv_op_block": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32] + [c_vp] * 10 + [c_vp, c_vp]),
    "mtgv_op_block_workspace_floats": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "mtgv_op_l2norm": (C.c_int, [c_vp, c_vp, c_i64, c_i32, c_vp]),
}

_lib = None


def lib():
    """Load (once) and return the C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python mtg-vision_amd/build.py` "
                "(mtgv has no CPU or eager fallback for the recognition path)"
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int):
    """Map the C status convention back to the reference's exception types."""
    if rc == 0:
        return
    msg = (lib().mtgv_last_error() or b"").decode("utf-8", "replace")
    if rc == 1:
        raise AssertionError(msg)
    if rc == 2:
        raise KeyError(msg)
    raise RuntimeError(msg)


def ptr(t) -> c_vp:
    """Device/host pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return c_vp(0)
    assert t.is_contiguous(), "mtgv: tensors crossing the C ABI must be contiguous"
    return c_vp(t.data_ptr())


def stream() -> c_vp:
    return c_vp(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("mtgv: no HIP device visible - the recognition path runs only on the GPU (no CPU fallback)")


PRECISIONS = {"f32": 0, "f16x3": 1}


def set_gemm_precision(name: str) -> None:
    """'f32' (f32-input MFMA) or 'f16x3' (fp16 hi+lo split, three fp16 MFMAs per product) for every GEMM launch."""
    if name not in PRECISIONS:
        raise AssertionError(f"precision {name!r}: expected one of {sorted(PRECISIONS)}")
    check(lib().mtgv_set_gemm_precision(PRECISIONS[name]))


def get_gemm_precision() -> str:
    v = c_i32(0)
    check(lib().mtgv_get_gemm_precision(C.byref(v)))
    return {n: k for k, n in PRECISIONS.items()}[v.value]
