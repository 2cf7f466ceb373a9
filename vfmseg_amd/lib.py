"""ctypes binding of libvfmseg_hip.so and its fp16 twin libvfmseg_hip_f16.so (include/vfmseg_hip.h: same sources, same ABI, built with
-DVFM_HALF_F16).  The product path has NO fallback: if the library is missing or a tensor is not on the GPU, the ops raise.

Which library `load()` returns follows the precision mode (precision.set_compute_dtype -> set_half): bf16 / f32 / bf16x3 use the
bf16 library, "fp16" (the autocast dtype of the reference's `--amp`) the twin.  dtype code BF16 = "the 16-bit type of the active
library"; handing a torch.float16 tensor to the bf16 library (or the reverse) raises in dt_of()."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvfmseg_hip.so")
LIB_PATH_F16 = os.path.join(_HERE, "csrc", "libvfmseg_hip_f16.so")

F32, BF16, U8, I64, SPLIT3 = 0, 1, 2, 3, 4
ABI_VERSION = 3   # include/vfmseg_hip.h as of round 4 (vfm_gemm_desc.c_plane): an older in-tree .so would misread the descriptors
EP_NONE, EP_GELU, EP_RELU, EP_MUL_GELU_GRAD, EP_MUL, EP_QGELU, EP_MUL_QGELU_GRAD, EP_GELU_DGELU = 0, 1, 2, 3, 4, 5, 6, 7
ACT_NONE, ACT_GELU, ACT_RELU, ACT_QGELU = 0, 1, 2, 3

vp, ci, cl, cf, u64 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint64


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp),
        ("in_dt", ci), ("c_dt", ci),
        ("M", cl), ("N", cl), ("K", cl),
        ("sa_m", cl), ("sa_k", cl), ("sb_n", cl), ("sb_k", cl), ("ldc", cl),
        ("alpha", cf),
        ("bias", vp), ("bias_mod", cl),
        ("colscale", vp),
        ("residual", vp), ("r_dt", ci), ("ldr", cl),
        ("ep_mode", ci), ("aux", vp), ("aux_dt", ci), ("ld_aux", cl),
        ("C2", vp), ("c2_dt", ci), ("ldc2", cl),
        ("batch", cl), ("stride_a", cl), ("stride_b", cl), ("stride_c", cl),
        ("c_plane", cl),
        ("kb_rows", cl),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", vp), ("k", vp), ("v", vp), ("o", vp),
        ("dt", ci), ("ldq", cl), ("ldk", cl), ("ldv", cl), ("ldo", cl),
        ("B", ci), ("H", ci), ("d", ci),
        ("nq_main", ci), ("nq_extra", ci), ("nk_main", ci), ("nk_extra", ci),
        ("scale", cf),
        ("lse", vp),
        ("dout", vp), ("ld_do", cl),
        ("dq", vp), ("dk", vp), ("dv", vp), ("ld_dq", cl), ("ld_dk", cl), ("ld_dv", cl),
        ("delta", vp),
    ]


class _LnFwd(C.Structure):
    _fields_ = [("x", vp), ("ld_x", cl), ("w", vp), ("b", vp), ("eps", cf), ("y", vp), ("y_dt", ci), ("ld_y", cl), ("stats", vp),
                ("rows", cl), ("C", cl)]


class _LnDrop(C.Structure):
    _fields_ = [("x", vp), ("ld_x", cl), ("w", vp), ("b", vp), ("eps", cf), ("y", vp), ("ld_y", cl), ("stats", vp), ("y_drop", vp),
                ("ld_yd", cl), ("mask", vp), ("ld_mask", cl), ("p", cf), ("offset", u64), ("rows", cl), ("C", cl)]


class _LnBwd(C.Structure):
    _fields_ = [("dy", vp), ("dy_dt", ci), ("ld_dy", cl), ("x", vp), ("ld_x", cl), ("w", vp), ("stats", vp), ("dx", vp), ("ld_dx", cl),
                ("accumulate_dx", ci), ("t_out", vp), ("ld_t", cl), ("t_scale", vp), ("rows", cl), ("C", cl)]


class _CastOp(C.Structure):
    _fields_ = [("src", vp), ("src_dt", ci), ("ld_src", cl), ("dst", vp), ("dst_dt", ci), ("ld_dst", cl), ("rows", cl), ("cols", cl),
                ("colscale", vp)]


class _CopyOp(C.Structure):
    _fields_ = [("src", vp), ("src_dt", ci), ("dst", vp), ("dst_dt", ci), ("n", cl * 4), ("ss", cl * 4), ("ds", cl * 4), ("accumulate", ci)]


class _PlanU(C.Union):
    _fields_ = [("gemm", GemmDesc), ("attn", AttnDesc), ("ln_fwd", _LnFwd), ("ln_drop", _LnDrop), ("ln_bwd", _LnBwd), ("cast", _CastOp),
                ("copy", _CopyOp)]


class PlanOp(C.Structure):
    """vfm_plan_op (include/vfmseg_hip.h, "launch plans")."""
    _fields_ = [("kind", ci), ("prof_kind", ci), ("flops", C.c_double), ("u", _PlanU)]


OP_GEMM, OP_LN_FWD, OP_LN_DROPOUT_FWD, OP_LN_BWD_SCALED, OP_ATTN_FWD, OP_ATTN_BWD, OP_CAST, OP_STRIDED_COPY = range(8)
PROF_NONE, PROF_GEMM, PROF_GEMM_TN, PROF_ATTN_FWD, PROF_ATTN_BWD = range(5)
PROF_NAMES = {PROF_GEMM: "gemm", PROF_GEMM_TN: "gemm_tn", PROF_ATTN_FWD: "attn_fwd", PROF_ATTN_BWD: "attn_bwd"}


class SlideWin(C.Structure):
    _fields_ = [("crop", vp), ("nchw", ci), ("h", ci), ("w", ci), ("y0", ci), ("x0", ci), ("hc", ci), ("wc", ci)]


# name -> argtypes (return type is always int); mirrors include/vfmseg_hip.h
SIGNATURES = {
    "vfm_cast": [vp, ci, cl, vp, ci, cl, cl, cl, vp, vp],
    "vfm_split3": [vp, cl, cl, vp, cl, cl, cl, ci, vp],
    "vfm_transpose": [vp, ci, cl, vp, ci, cl, cl, cl, cl, vp],
    "vfm_strided_copy": [vp, ci, vp, ci] + [cl] * 12 + [ci, vp],
    "vfm_strided_copy_batch": [vp, ci, cl, vp],
    "vfm_axpby": [vp, cf, vp, cf, cl, vp],
    "vfm_scale_by_device_scalar": [vp, vp, cl, vp],
    "vfm_colsum": [vp, ci, cl, cl, cl, vp, ci, vp, vp],
    "vfm_preprocess_u8": [vp, ci, ci, vp, ci, ci, C.POINTER(C.c_float), C.POINTER(C.c_float), ci, cf, vp],
    "vfm_lora_pack": [vp, ci, cl, ci, vp],
    "vfm_slab_reduce": [vp, ci, cl, cl, cl, cf, vp, cl, cl, ci, vp],
    "vfm_dropout_mask": [vp, ci, cl, cf, u64, u64, vp],
    "vfm_mul_mask": [vp, ci, cl, vp, ci, cl, cl, vp, ci, cl, cl, cl, vp],
    "vfm_geglu_fwd": [vp, ci, cl, vp, ci, cl, cl, cl, vp],
    "vfm_geglu_bwd": [vp, ci, cl, vp, ci, cl, vp, ci, cl, cl, cl, vp],
    "vfm_swiglu_fwd": [vp, ci, cl, vp, ci, cl, cl, cl, vp],
    "vfm_swiglu_bwd": [vp, ci, cl, vp, ci, cl, vp, ci, cl, cl, cl, vp],
    "vfm_rope": [vp, ci, cl, cl, ci, ci, ci, vp, vp, ci, vp],
    "vfm_act_grad_mul": [vp, ci, cl, vp, ci, cl, vp, ci, cl, cl, cl, ci, vp],
    "vfm_mask_token_fwd": [vp, vp, vp, vp, cl, cl, vp],
    "vfm_mask_token_bwd": [vp, vp, vp, vp, vp, cl, cl, vp],
    "vfm_layernorm_fwd": [vp, cl, vp, vp, cf, vp, ci, cl, vp, cl, cl, vp],
    "vfm_layernorm_fwd_split3": [vp, cl, vp, vp, cf, vp, cl, vp, cl, cl, vp, cl, cl, vp],
    "vfm_layernorm_dropout_fwd": [vp, cl, vp, vp, cf, vp, cl, vp, vp, cl, vp, cl, cf, u64, u64, cl, cl, vp],
    "vfm_layernorm_bwd_scaled": [vp, ci, cl, vp, cl, vp, vp, vp, cl, ci, vp, cl, vp, cl, cl, vp],
    "vfm_layernorm_bwd": [vp, ci, cl, vp, cl, vp, vp, vp, cl, ci, vp, vp, vp, cl, cl, vp],
    "vfm_groupnorm_fwd": [vp, vp, vp, cf, ci, ci, vp, ci, vp, vp, cl, cl, cl, vp],
    "vfm_groupnorm_bwd": [vp, ci, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, cl, cl, cl, vp],
    "vfm_bn_moments": [vp, cl, cl, vp, vp, vp],
    "vfm_bn_finalize": [vp, cf, vp, vp, vp, cf, cl, vp],
    "vfm_bn_apply": [vp, vp, vp, vp, cf, ci, vp, ci, cl, cl, vp],
    "vfm_bn_bwd_reduce": [vp, ci, vp, vp, vp, vp, cf, ci, vp, vp, cl, cl, vp],
    "vfm_bn_bwd_apply": [vp, ci, vp, vp, vp, vp, cf, ci, vp, cf, vp, cl, cl, vp],
    "vfm_gemm": [C.POINTER(GemmDesc), vp],
    "vfm_tune": [C.c_char_p, ci],
    "vfm_attn_fwd": [C.POINTER(AttnDesc), vp],
    "vfm_attn_fwd_x3": [C.POINTER(AttnDesc), cl, vp],
    "vfm_attn_fwd_x3_split": [C.POINTER(AttnDesc), cl, vp, cl, cl, vp],
    "vfm_attn_bwd": [C.POINTER(AttnDesc), vp],
    "vfm_attn_bwd_x3": [C.POINTER(AttnDesc), vp, vp, vp, cl, cl, vp, cl, cl, vp],
    "vfm_sam_relpos_table": [vp, ci, ci, ci, vp, vp],
    "vfm_sam_attn_prep": [vp, ci, cl, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, cf, vp],
    "vfm_softmax_rows": [vp, cl, vp, ci, cl, cl, ci, ci, vp],
    "vfm_sam_attn_merge": [vp, ci, vp, cl, ci, ci, ci, ci, ci, ci, vp],
    "vfm_sam_attn_bwd_prep": [vp, cl, vp, cl, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, cf, vp],
    "vfm_softmax_rows_batched": [vp, cl, vp, ci, cl, cl, ci, ci, ci, ci, vp],
    "vfm_softmax_rows_bwd": [vp, vp, cl, vp, ci, cl, cl, ci, ci, ci, ci, vp],
    "vfm_sam_attn_bwd_merge": [vp, vp, vp, ci, vp, vp, vp, cl, ci, ci, ci, ci, ci, ci, ci, ci, cf, vp],
    "vfm_sam_attn_flash_fwd": [vp, cl, vp, vp, vp, vp, cl, ci, ci, ci, ci, ci, cf, vp],
    "vfm_sam_attn_flash_fwd_train": [vp, cl, vp, vp, vp, vp, cl, ci, ci, ci, ci, ci, cf, vp, vp, vp],
    "vfm_sam_attn_flash_bwd": [vp, cl, vp, vp, vp, vp, vp, cl, vp, vp, vp, vp, cl, ci, ci, ci, ci, ci, cf, vp],
    "vfm_patchify": [vp, cl, cl, cl, ci, ci, ci, ci, ci, vp, ci, cl, ci, vp],
    "vfm_assemble_tokens": [vp, vp, vp, vp, ci, ci, ci, vp],
    "vfm_resize_bilinear": [vp, ci, ci, ci, ci, ci, ci, cl, vp, ci, ci, cl, ci, ci, ci, ci, ci, ci, vp],
    "vfm_resize_bicubic": [vp, ci, ci, ci, vp, ci, ci, cf, cf, vp],
    "vfm_label_resize": [vp, ci, ci, ci, vp, ci, ci, ci, ci, ci, ci, vp],
    "vfm_unblock": [vp, vp, ci, ci, ci, ci, ci, ci, vp],
    "vfm_upsample_ce": [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp],
    "vfm_reduce_sum": [vp, cl, cf, vp, vp],
    "vfm_ce_finish": [vp, cl, cf, vp, cf, vp, vp, vp],
    "vfm_conf_gate": [vp, ci, ci, ci, ci, ci, ci, ci, ci, cf, vp, vp],
    "vfm_slide_accumulate": [vp, ci, ci, ci, ci, ci, vp, vp, ci, ci, ci, ci, ci, ci, vp],
    "vfm_slide_finalize": [vp, vp, vp, ci, ci, ci, ci, vp],
    "vfm_slide_gather": [vp, ci, ci, ci, vp, ci, ci, vp],
    "vfm_conf_gate_windows": [vp, ci, ci, ci, ci, vp, ci, cf, vp, vp],
    "vfm_confusion_hist": [vp, vp, ci, cl, ci, ci, vp, vp],
    "vfm_adamw": [vp, vp, vp, vp, cl, vp, vp, vp, ci, cf, cf, cf, cf, ci, cf, ci, ci, vp],
    "vfm_adamw_guarded": [vp, vp, vp, vp, cl, vp, vp, vp, ci, cf, cf, cf, cf, ci, cf, ci, ci, vp, vp, vp],
    "vfm_amp_update": [vp, vp, cf, cf, ci, ci, vp],
    "vfm_run_plan": [C.POINTER(PlanOp), ci, u64, u64, vp],
    "vfm_prof_config": [ci],
    "vfm_prof_read": [C.POINTER(C.c_double), ci],
}

_lib = None          # the ACTIVE library (what load() returns)
_libs = {}           # half kind (0 bf16, 1 fp16) -> loaded library
_half = 0            # active half kind
HALF_DTYPES = (torch.bfloat16, torch.float16)


class HipLibraryMissing(RuntimeError):
    pass


def _open(kind):
    path = LIB_PATH_F16 if kind else LIB_PATH
    if not os.path.exists(path):
        raise HipLibraryMissing(
            f"{path} not found: build it with `python -m vfmseg_amd.csrc.build` (hipcc --offload-arch=gfx950)")
    lib = C.CDLL(path)
    lib.vfm_last_error.restype = C.c_char_p
    lib.vfm_abi_version.restype = ci
    lib.vfm_half_kind.restype = ci
    if lib.vfm_abi_version() < ABI_VERSION:
        raise HipLibraryMissing(f"{path} has ABI version {lib.vfm_abi_version()}, this binding needs {ABI_VERSION}: rebuild (python -m vfmseg_amd.csrc.build)")
    if lib.vfm_half_kind() != kind:
        raise HipLibraryMissing(f"{path} reports half kind {lib.vfm_half_kind()}, expected {kind}: stale build")
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = ci
    return lib


def load():
    """The active shared library (loaded once). Raises HipLibraryMissing - there is no CPU fallback in the product path."""
    global _lib
    if _lib is not None:
        return _lib
    if _half not in _libs:
        _libs[_half] = _open(_half)
    _lib = _libs[_half]
    return _lib


def set_half(kind):
    """Select the library whose 16-bit type is bf16 (0 / torch.bfloat16) or IEEE fp16 (1 / torch.float16).  Loaded lazily."""
    global _half, _lib
    kind = {torch.bfloat16: 0, torch.float16: 1, "bf16": 0, "fp16": 1, "f16": 1}.get(kind, kind)
    assert kind in (0, 1)
    if kind != _half:
        _half, _lib = kind, None


def half_dtype():
    """torch dtype of the active library's 16-bit type."""
    return torch.float16 if _half else torch.bfloat16


class HipError(RuntimeError):
    pass


def check(rc, name):
    if rc != 0:
        raise HipError(f"{name} failed ({rc}): {load().vfm_last_error().decode()}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_dev_index = None


def stream():
    """Raw hipStream_t of torch's current stream on the current device (fast path: no Stream object)."""
    global _dev_index
    if _raw_stream is not None:
        if _dev_index is None:
            _dev_index = torch.cuda.current_device()
        return _raw_stream(_dev_index)
    return torch.cuda.current_stream().cuda_stream


def set_device_index(i):
    """Call after torch.cuda.set_device() when a process changes its device (one process per GPU: once)."""
    global _dev_index
    _dev_index = i


def dt_of(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16 or t.dtype == torch.float16:
        if (t.dtype == torch.float16) != bool(_half):
            raise TypeError(f"{t.dtype} tensor handed to the {'fp16' if _half else 'bf16'} library (precision mode and tensors disagree)")
        return BF16
    if t.dtype == torch.uint8 or t.dtype == torch.bool:
        return U8
    if t.dtype == torch.int64:
        return I64
    raise TypeError(f"unsupported dtype {t.dtype}")


def torch_dt(code):
    return torch.float32 if code == F32 else half_dtype()


def ptr(t):
    """Device pointer of a tensor (None -> NULL). Refuses CPU tensors: the HIP path must be the one that runs."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError("vfmseg_amd ops need GPU tensors (no CPU fallback in the product path)")
    return t.data_ptr()
