"""fennec_amd -- fennec's per-pixel hot path on AMD Instinct MI355X (gfx950).

A thin ctypes binding of ``libfennec_hip.so`` (C ABI: ``include/fennec_hip.h``):
hand-written HIP kernels behind the reference's own function names
(``SSIM``, ``SSIMFast``, ``MSSSIM``, ``GaussianBlur``, ``Sharpen``,
``AdaptiveSharpen``, ``ApplyOrientation``, ``lanczosResize``, ``smartResize``,
``boxDownsample``; shamspias/fennec ssim.go / resize.go / effects.go / exif.go).

Images are ``image.NRGBA`` byte buffers:

* numpy ``uint8`` arrays of shape ``(h, w, 4)`` with contiguous rows -> host space
  (the library stages them through the GPU and returns numpy arrays);
* torch ``uint8`` CUDA/HIP tensors of shape ``(h, w, 4)`` -> device space (results
  are new device tensors; nothing crosses PCIe).

There is NO CPU implementation in this package: without the built library or
without a GPU every call raises.  Build with ``python -c "import __graft_entry__
as g; g.build()"`` or ``make -C fennec_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

from . import synth  # noqa: F401  (deterministic test/bench images)

__all__ = [
    "Context", "default_context", "load_library", "FennecError",
    "SSIM", "SSIMFast", "MSSSIM", "GaussianBlur", "Sharpen", "AdaptiveSharpen",
    "ApplyOrientation", "lanczosResize", "smartResize", "boxDownsample",
    "gaussianKernel", "blurKernel", "precomputeWeights", "Summarize",
]

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FENNEC_HIP_LIB") or os.path.join(_HERE, "libfennec_hip.so")   # override: A/B builds

FNX_OK, FNX_NOOP, FNX_EMPTY = 0, 1, 2
FNX_HOST, FNX_DEVICE, FNX_DEVICE_SRC = 0, 1, 2
FNX_BLUR_FAST, FNX_BLUR_EXACT, FNX_BLUR_KEEP_BOX_SUMS = 0, 1, 2
PROF_MAIN, PROF_SSIM, PROF_RESIZE, PROF_FX, PROF_JPEG = 1, 2, 4, 8, 16

_u8p = C.c_void_p
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


class Analysis(C.Structure):
    """fnx_analysis (include/fennec_hip.h): what the device computes for Analyze."""
    _fields_ = [("histogram", C.c_uint64 * 256), ("bright_sum", C.c_double), ("variance_sum", C.c_double),
                ("sample_count", C.c_int64), ("edge_count", C.c_int64), ("edge_total", C.c_int64),
                ("unique_colors", C.c_int32), ("has_alpha", C.c_int32), ("is_grayscale", C.c_int32),
                ("pad", C.c_int32)]


class ImageStats(C.Structure):
    """fennec_ImageStats = ImageStats (analyze.go:9-22)."""
    _fields_ = [("Width", C.c_int32), ("Height", C.c_int32), ("HasAlpha", C.c_int32), ("IsGrayscale", C.c_int32),
                ("UniqueColors", C.c_int32), ("RecommendedFormat", C.c_int32), ("RecommendedQuality", C.c_int32),
                ("pad", C.c_int32), ("Entropy", C.c_double), ("EdgeDensity", C.c_double),
                ("MeanBrightness", C.c_double), ("Contrast", C.c_double), ("EstimatedCompression", C.c_double)]


class NativeBatchResult(C.Structure):
    """fennec_BatchResult (include/fennec_hip.h)."""
    _fields_ = [("index", C.c_int32), ("failed", C.c_int32), ("has_result", C.c_int32), ("quality", C.c_int32), ("steps", C.c_int32),
                ("status", C.c_int32), ("original_size", C.c_int64), ("compressed_size", C.c_int64), ("ssim", C.c_double),
                ("device", C.c_int32), ("reserved", C.c_int32)]


FNX_ERR_INVALID = -1
FNX_ERR_UNSUPPORTED = -5


class FileOptions(C.Structure):
    """fennec_FileOptions (include/fennec_hip.h): the Options fields CompressFile's JPEG path reads."""
    _fields_ = [("orient", C.c_int32), ("max_w", C.c_int32), ("max_h", C.c_int32), ("auto_format", C.c_int32), ("target_ssim", C.c_double)]


class FennecError(RuntimeError):
    pass


class FennecUnsupported(FennecError):
    """fnx_jpeg_decode / fnx_jpeg_recompress: a file the device decoder does not take (FNX_ERR_UNSUPPORTED) -- decode it
    on the host."""


_lib = None
_lib_lock = threading.Lock()


def _sig(L, name, restype, argtypes):
    f = getattr(L, name)
    f.restype = restype
    f.argtypes = argtypes


def load_library() -> C.CDLL:
    """dlopen libfennec_hip.so and declare every entry point of include/fennec_hip.h."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise FennecError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(make -C fennec_amd/csrc, or __graft_entry__.build()); there is no CPU fallback")
        # PyTorch wheels bundle their own HIP runtime and link it by file name, so it is not
        # deduplicated against /opt/rocm's copy by soname: if our library pulled the system
        # runtime in first, torch would later load a SECOND runtime that sees no GPUs.  Let
        # torch (when present) load its runtime first; ours then binds to the same one.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        ctx = C.c_void_p
        i = C.c_int
        d = C.c_double
        img = [_u8p, i]                                   # pointer, stride
        _sig(L, "fnx_version", C.c_char_p, [])
        _sig(L, "fnx_device_count", i, [])
        _sig(L, "fnx_set_devices", i, [C.POINTER(C.c_int), i])
        _sig(L, "fnx_last_error", C.c_char_p, [])
        _sig(L, "fnx_ctx_create", i, [i, C.POINTER(ctx)])
        _sig(L, "fnx_ctx_destroy", None, [ctx])
        _sig(L, "fnx_ctx_device", i, [ctx])
        _sig(L, "fnx_ctx_stream", C.c_void_p, [ctx])
        _sig(L, "fnx_ctx_sync", i, [ctx])
        _sig(L, "fnx_ctx_use_stream", i, [ctx, C.c_void_p])
        _sig(L, "fnx_ctx_use_own_stream", i, [ctx])
        _sig(L, "fnx_ctx_profile", i, [ctx, i])
        _sig(L, "fnx_ctx_kernel_ms", i, [ctx, C.POINTER(C.c_float)])
        _sig(L, "fnx_ctx_last_kernel", C.c_char_p, [ctx, i])
        _sig(L, "fnx_ctx_set_form", i, [ctx, C.c_char_p, C.c_char_p])
        _sig(L, "fnx_ctx_set_ssim_mode", i, [ctx, i])
        _sig(L, "fnx_malloc", i, [ctx, C.c_size_t, C.POINTER(C.c_void_p)])
        _sig(L, "fnx_free", i, [ctx, C.c_void_p])
        _sig(L, "fnx_upload", i, [ctx, C.c_void_p, i, C.c_void_p, i, i, i])
        _sig(L, "fnx_download", i, [ctx, C.c_void_p, i, C.c_void_p, i, i, i])
        _sig(L, "fnx_gaussian_blur", i, [ctx, i] + img + [i, i, _f64p, i, i] + img)
        _sig(L, "fnx_blur_fixed_point", i, [_f64p, i, C.POINTER(C.c_longlong), C.POINTER(d)])
        _sig(L, "fnx_blur3x3", i, [ctx, i] + img + [i, i] + img)
        _sig(L, "fnx_sharpen", i, [ctx, i] + img + [i, i, d] + img)
        _sig(L, "fnx_adaptive_sharpen", i, [ctx, i] + img + [i, i, d] + img)
        _sig(L, "fnx_resize_h", i, [ctx, i] + img + [i, i, _i32p, _i32p, _f64p] + img + [i])
        _sig(L, "fnx_resize_v", i, [ctx, i] + img + [i, i, _i32p, _i32p, _f64p] + img + [i])
        _sig(L, "fnx_lanczos_resize", i, [ctx, i] + img + [i, i, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p] + img + [i, i])
        _sig(L, "fnx_box_downsample", i, [ctx, i] + img + [i, i] + img + [i, i])
        _sig(L, "fnx_ssim_fast", i, [ctx, i] + img + img + [i, i, _f64p, _f64p])
        _sig(L, "fnx_ssim", i, [ctx, i] + img + img + [i, i, _f64p, _f64p])
        _sig(L, "fnx_pixel_ssim", i, [ctx, i, _u8p, C.c_size_t, _u8p, C.c_size_t, i, i, _f64p])
        _sig(L, "fnx_msssim", i, [ctx, i] + img + img + [i, i, _f64p, _f64p, _f64p])
        _sig(L, "fnx_ssim_fast_prepare", i, [ctx, i] + img + [i, i, C.POINTER(C.c_void_p)])
        _sig(L, "fnx_ssim_fast_against", i, [ctx, C.c_void_p, i] + img + [_f64p, _f64p])
        _sig(L, "fnx_prepared_free", None, [ctx, C.c_void_p])
        _sig(L, "fnx_orient", i, [ctx, i] + img + [i, i, i] + img)
        _sig(L, "fnx_gaussian_blur_batch", i, [ctx, i, C.POINTER(C.c_void_p), i, i, i, _f64p, i, i, C.POINTER(C.c_void_p), i])
        _sig(L, "fnx_ssim_fast_batch", i, [ctx, i, C.POINTER(C.c_void_p), i, C.POINTER(C.c_void_p), i, i, i, _f64p, _f64p])
        _sig(L, "fnx_ssim_fast_batch_enqueue", i, [ctx, i, C.POINTER(C.c_void_p), i, C.POINTER(C.c_void_p), i, i, i, _f64p])
        _sig(L, "fnx_results_fetch", i, [ctx, i, _f64p])
        _sig(L, "fnx_ssim_enqueue", i, [ctx] + img + img + [i, i, _f64p])
        _sig(L, "fnx_msssim_enqueue", i, [ctx] + img + img + [i, i, _f64p])
        _sig(L, "fennec_MSSSIM_enqueue", i, [ctx] + img + [i, i] + img + [i, i])
        _sig(L, "fnx_jpeg_encode", i, [ctx, i] + img + [i, i, i, _u8p, C.c_size_t, C.POINTER(C.c_size_t)])
        _sig(L, "fennec_CompressBatchNRGBA", i, [i, i, i, i, C.POINTER(C.c_void_p), C.POINTER(i), C.POINTER(i), C.POINTER(i), _i64p, d,
                                                  C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(NativeBatchResult), C.POINTER(i),
                                                  C.c_void_p, C.c_void_p])
        _sig(L, "fennec_CompressBatchJPEG", i, [i, i, i, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), d, C.POINTER(C.c_void_p),
                                                 C.POINTER(C.c_size_t), C.POINTER(NativeBatchResult), C.POINTER(i), C.c_void_p, C.c_void_p])
        _sig(L, "fennec_CompressBatchNRGBADevices", i, [C.POINTER(i), i, i, i, i, C.POINTER(C.c_void_p), C.POINTER(i), C.POINTER(i), C.POINTER(i),
                                                         _i64p, d, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(NativeBatchResult),
                                                         C.POINTER(i), C.c_void_p, C.c_void_p])
        _sig(L, "fennec_CompressBatchJPEGDevices", i, [C.POINTER(i), i, i, i, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), d,
                                                        C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(NativeBatchResult), C.POINTER(i),
                                                        C.c_void_p, C.c_void_p])
        _sig(L, "fennec_CompressFileJPEG", i, [ctx, _u8p, C.c_size_t, C.POINTER(FileOptions), _u8p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(i),
                                                _f64p, C.POINTER(i), C.POINTER(i)])
        _sig(L, "fennec_CompressBatchJPEGOpts", i, [i, i, i, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(FileOptions),
                                                     C.POINTER(C.POINTER(FileOptions)), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                                     C.POINTER(NativeBatchResult), C.POINTER(i), C.POINTER(i), C.c_void_p, C.c_void_p])
        _sig(L, "fennec_pool_release", None, [])
        _sig(L, "fennec_SummarizeResults", d, [i, C.POINTER(NativeBatchResult), _i64p])
        _sig(L, "fnx_jpeg_size_search", i, [ctx, i] + img + [i, i, C.c_longlong, i, _f64p, _u8p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(i),
                                             _f64p, C.POINTER(i)])
        _sig(L, "fnx_jpeg_compress", i, [ctx, i] + img + [i, i, d, _f64p, _u8p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(i), _f64p,
                                          C.POINTER(i)])
        _sig(L, "fnx_jpeg_roundtrip", i, [ctx, i] + img + [i, i, i] + img)
        _sig(L, "fnx_jpeg_decode", i, [ctx, _u8p, C.c_size_t, i, C.c_void_p, i, C.POINTER(i), C.POINTER(i)])
        _sig(L, "fnx_jpeg_progressive_coefficients", i, [_u8p, C.c_size_t, C.POINTER(C.c_int16), C.c_size_t, C.POINTER(C.c_size_t),
                                                        C.POINTER(i), C.POINTER(i), C.POINTER(i)])
        _sig(L, "fnx_jpeg_recompress", i, [ctx, _u8p, C.c_size_t, d, _f64p, _u8p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(i), _f64p,
                                            C.POINTER(i), C.POINTER(i), C.POINTER(i)])
        _sig(L, "fnx_jpeg_quality_search", i, [ctx, i] + img + [i, i, d, _f64p, C.POINTER(i), _f64p, C.POINTER(i)])
        _sig(L, "fnx_gaussian_blur_ssim_fast", i, [ctx, i] + img + [i, i, _f64p, i, i] + img + [_f64p, _f64p])
        _sig(L, "fnx_ssim_batch_enqueue", i, [ctx, i, C.POINTER(C.c_void_p), i, C.POINTER(C.c_void_p), i, i, i, _f64p])
        _sig(L, "fnx_sharpen_batch", i, [ctx, i, C.POINTER(C.c_void_p), i, i, i, C.c_double, C.POINTER(C.c_void_p), i])
        _sig(L, "fnx_adaptive_sharpen_batch", i, [ctx, i, C.POINTER(C.c_void_p), i, i, i, C.c_double, C.POINTER(C.c_void_p), i])
        _sig(L, "fnx_msssim_batch_enqueue", i, [ctx, i, C.POINTER(C.c_void_p), i, C.POINTER(C.c_void_p), i, i, i, _f64p])
        _sig(L, "fennec_MSSSIM_batch_enqueue", i, [ctx, i, C.POINTER(C.c_void_p), i, i, i, C.POINTER(C.c_void_p), i, i, i])
        _sig(L, "fnx_lanczos_resize_batch", i, [ctx, i, C.POINTER(C.c_void_p), i, i, i, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p,
                                                C.POINTER(C.c_void_p), i, i, i])
        _sig(L, "fennec_lanczosResizeBatch", i, [ctx, i, C.POINTER(C.c_void_p), i, i, i, C.POINTER(C.c_void_p), i, i, i])
        _sig(L, "fnx_gaussian_blur_ssim_fast_batch", i,
             [ctx, i, C.POINTER(C.c_void_p), i, i, i, _f64p, i, i, C.POINTER(C.c_void_p), i, _f64p, _f64p])
        _sig(L, "fnx_gaussian_blur_ssim_fast_batch_enqueue", i,
             [ctx, i, C.POINTER(C.c_void_p), i, i, i, _f64p, i, i, C.POINTER(C.c_void_p), i, _f64p])
        _sig(L, "fnx_analyze", i, [ctx, i] + img + [i, i, C.POINTER(Analysis)])
        _sig(L, "fnx_analyze_batch", i, [ctx, i, C.POINTER(C.c_void_p), i, i, i, C.POINTER(Analysis)])
        _sig(L, "fnx_scan_flags", i, [ctx, i, _u8p, C.c_size_t, C.POINTER(i), C.POINTER(i)])
        _sig(L, "fennec_Analyze", i, [ctx, i] + img + [i, i, C.POINTER(ImageStats)])
        _sig(L, "fennec_statsFromAnalysis", None, [C.POINTER(Analysis), i, i, C.POINTER(ImageStats)])
        _sig(L, "fennec_isOpaque", i, [ctx, i] + img + [i, i, C.POINTER(i)])
        _sig(L, "fennec_isGrayscale", i, [ctx, i] + img + [i, i, C.POINTER(i)])
        _sig(L, "fnx_ycbcr_to_nrgba", i, [ctx, i, _u8p, i, _u8p, _u8p, i, i, i, i] + img)
        _sig(L, "fnx_ssim_fast_against_ycbcr", i, [ctx, C.c_void_p, i, _u8p, i, _u8p, _u8p, i, i, _f64p, _f64p])
        _sig(L, "fnx_apply_palette", i, [ctx, i] + img + [i, i, _u8p, i, _u8p, i, _u8p, i])
        _sig(L, "fennec_gaussianKernel", None, [i, d, _f64p])
        _sig(L, "fennec_blurKernel", i, [d, _f64p])
        _sig(L, "fennec_lanczosKernel", d, [d])
        _sig(L, "fennec_precomputeWeights", i, [i, i, _i32p, _i32p, _f64p])
        _sig(L, "fennec_smartResizeDims", i, [i, i, i, i, C.POINTER(i), C.POINTER(i)])
        _sig(L, "fennec_ssimFastDims", i, [i, i, C.POINTER(i), C.POINTER(i)])
        _sig(L, "fennec_SSIM", i, [ctx, i] + img + [i, i] + img + [i, i, _f64p])
        _sig(L, "fennec_SSIMFast", i, [ctx, i] + img + img + [i, i, _f64p])
        _sig(L, "fennec_MSSSIM", i, [ctx, i] + img + [i, i] + img + [i, i, _f64p])
        _sig(L, "fennec_GaussianBlur", i, [ctx, i] + img + [i, i, d] + img)
        _sig(L, "fennec_Sharpen", i, [ctx, i] + img + [i, i, d] + img)
        _sig(L, "fennec_AdaptiveSharpen", i, [ctx, i] + img + [i, i, d] + img)
        _sig(L, "fennec_ApplyOrientation", i, [ctx, i] + img + [i, i, i] + img)
        _sig(L, "fennec_lanczosResize", i, [ctx, i] + img + [i, i] + img + [i, i])
        _sig(L, "fennec_boxDownsample", i, [ctx, i] + img + [i, i] + img + [i, i])
        _sig(L, "fennec_Summarize", d, [i, _i32p, _i32p, _i64p, _i64p, _f64p, _i64p])
        _lib = L
        return L


# names include/fennec_hip.h declares (checked by tests/test_abi.py against the header itself)
def exported_symbols():
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "fennec_hip.h")
    text = open(hdr).read()
    return sorted(set(re.findall(r"\b((?:fnx|fennec)_\w+)\s*\(", text)))


# ---------------------------------------------------------------------------------------
def _is_torch(x) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


class _Img:
    """(space, pointer, stride, w, h) view of a numpy array or a torch device tensor."""

    __slots__ = ("space", "ptr", "stride", "w", "h", "obj")

    def __init__(self, x):
        self.obj = x
        if _is_torch(x):
            import torch
            if not x.is_cuda or x.dtype != torch.uint8 or x.dim() != 3 or x.shape[2] != 4:
                raise FennecError("device images must be uint8 CUDA/HIP tensors of shape (h, w, 4)")
            h, w = int(x.shape[0]), int(x.shape[1])
            if h > 0 and w > 0 and (x.stride(2) != 1 or x.stride(1) != 4):
                raise FennecError("image rows must be contiguous")
            self.space, self.ptr = FNX_DEVICE, x.data_ptr()
            self.stride = int(x.stride(0)) if h > 1 else w * 4
        else:
            a = x
            if not isinstance(a, np.ndarray) or a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 4:
                raise FennecError("host images must be numpy uint8 arrays of shape (h, w, 4)")
            h, w = a.shape[:2]
            if h > 0 and w > 0 and (a.strides[2] != 1 or a.strides[1] != 4):
                raise FennecError("image rows must be contiguous")
            self.space, self.ptr = FNX_HOST, a.ctypes.data
            self.stride = int(a.strides[0]) if h > 1 else w * 4
        self.w, self.h = int(w), int(h)

    def like(self, w: int, h: int):
        """A fresh tight w x h image in the same space.  Every kernel writes every dst pixel
        (zeros where image.NewNRGBA's zero pixel survives in the reference), so no memset --
        a torch.zeros fill would also run on torch's stream and race with the ctx stream."""
        if self.space == FNX_DEVICE:
            import torch
            return torch.empty((h, w, 4), dtype=torch.uint8, device=self.obj.device)
        return np.empty((h, w, 4), dtype=np.uint8)


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_f64p)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


class Context:
    """One GPU, one stream, its scratch (fnx_ctx).  Not re-entrant: one per worker thread."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.fnx_ctx_create(int(device), C.byref(h))
        if rc < 0:
            raise FennecError(f"fnx_ctx_create({device}): {self._err()}")
        self._h = h
        self.device = int(device)
        self._lent = -1            # handle of the torch stream the ctx currently launches on (-1: its own stream)

    # -- plumbing ---------------------------------------------------------------------
    def _err(self) -> str:
        return (self._lib.fnx_last_error() or b"").decode()

    def _chk(self, rc: int, what: str) -> int:
        if rc == FNX_ERR_UNSUPPORTED:
            raise FennecUnsupported(f"{what} ({rc}): {self._err()}")
        if rc < 0:
            raise FennecError(f"{what} failed ({rc}): {self._err()}")
        return rc

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fnx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._chk(self._lib.fnx_ctx_sync(self._h), "fnx_ctx_sync")

    def profile(self, enable=True):
        """Bracket kernel launches with HIP events (fnx_ctx_profile).  True / 1: the blur / analyze pass
        kernels; or a mask of PROF_MAIN | PROF_SSIM | PROF_RESIZE | PROF_FX."""
        self._chk(self._lib.fnx_ctx_profile(self._h, int(enable)), "fnx_ctx_profile")

    def kernel_ms(self) -> float:
        """Duration, in ms, of the oldest bracketed kernel launch not read yet (waits for it): one call per
        launch, in launch order."""
        ms = C.c_float(0.0)
        self._chk(self._lib.fnx_ctx_kernel_ms(self._h, C.byref(ms)), "fnx_ctx_kernel_ms")
        return float(ms.value)

    def last_kernel(self, prof_class: int = 1) -> str:
        """The kernel this ctx's last call of a class (PROF_MAIN = GaussianBlur, PROF_SSIM, PROF_RESIZE) launched: the
        route the library's dispatch really took (fnx_ctx_last_kernel); "" before the first such call."""
        s = self._lib.fnx_ctx_last_kernel(self._h, int(prof_class))
        if s is None:
            raise FennecError("fnx_ctx_last_kernel: bad class")
        return s.decode()

    def set_ssim_mode(self, fast: bool) -> None:
        """fnx_ctx_set_ssim_mode: fp32 moments (<= 1e-6) instead of fp64 (<= 1e-9) for full-resolution SSIM planes"""
        self._chk(self._lib.fnx_ctx_set_ssim_mode(self._h, 1 if fast else 0), "fnx_ctx_set_ssim_mode")

    def set_form(self, name: str, value=None) -> None:
        """fnx_ctx_set_form: which of several kernels that compute the same bytes this ctx takes (tests, A/B timing);
        value None: the product's own choice again."""
        v = None if value is None else str(value).encode()
        self._chk(self._lib.fnx_ctx_set_form(self._h, name.encode(), v), "fnx_ctx_set_form")

    def forms(self, **kw):
        """`with ctx.forms(fx_stream=0, fx_pairs=1): ...` -- set_form for the block, defaults afterwards"""
        ctx = self

        class _Forms:
            def __enter__(self_inner):
                for k, v in kw.items():
                    ctx.set_form(k, v)

            def __exit__(self_inner, *exc):
                for k in kw:
                    ctx.set_form(k, None)
                return False
        return _Forms()

    @property
    def stream(self) -> int:
        return int(self._lib.fnx_ctx_stream(self._h) or 0)

    # -- ordering against torch's streams ---------------------------------------------------
    # Device-space calls only ENQUEUE.  torch knows nothing about a private HIP stream: inputs may still be being
    # written on torch's current stream, outputs would be read by torch before the kernels have run, and the caching
    # allocator could recycle a dropped tensor while kernels still use it.  So every method that takes device tensors
    # runs inside `_ordered(...)`, which makes the ctx LAUNCH ON torch's current stream (fnx_ctx_use_stream): the work
    # is then ordered like any torch op on that stream and the allocator's stream-ordered reuse is safe -- with no
    # cross-stream event per call (the first version waited both ways on every call: ~25 us of idle GPU between two
    # back-to-back calls, measured on the config-4 timeline).  The switch happens once per (ctx, stream) change.
    # Concurrency between worker contexts therefore needs a torch stream per worker (`with torch.cuda.stream(s)`),
    # exactly as for torch's own ops.  Host (numpy) calls are synchronous and skip all of this.
    class _Ordered:
        __slots__ = ("ctx", "tensors")

        def __init__(self, ctx, tensors):
            self.ctx = ctx
            self.tensors = [t for t in tensors if _is_torch(t) and t.is_cuda]

        def __enter__(self):
            if self.tensors:
                import torch
                cur = torch.cuda.current_stream(self.tensors[0].device).cuda_stream
                if self.ctx._lent != cur:
                    self.ctx._chk(self.ctx._lib.fnx_ctx_use_stream(self.ctx._h, C.c_void_p(cur)), "fnx_ctx_use_stream")
                    self.ctx._lent = cur
            return self

        def add(self, *tensors):
            self.tensors.extend(t for t in tensors if _is_torch(t) and t.is_cuda)

        def __exit__(self, *exc):
            return False

    def _ordered(self, *tensors):
        return Context._Ordered(self, tensors)

    def use_own_stream(self):
        """Back to the ctx's private stream (plans and C-style callers that order work themselves)."""
        self._chk(self._lib.fnx_ctx_use_own_stream(self._h), "fnx_ctx_use_own_stream")
        self._lent = -1

    def _pair(self, a, b):
        ia, ib = _Img(a), _Img(b)
        if ia.space != ib.space:
            raise FennecError("both images must live in the same space (numpy or device tensors)")
        return ia, ib

    # -- table generators (Go side of the seam) -----------------------------------------
    def gaussianKernel(self, size: int = 8, sigma: float = 1.5) -> np.ndarray:
        k = np.empty(size * size, dtype=np.float64)
        self._lib.fennec_gaussianKernel(size, sigma, k.ctypes.data_as(_f64p))
        return k

    def blurKernel(self, sigma: float):
        r = self._lib.fennec_blurKernel(float(sigma), None)
        k = np.empty(2 * r + 1, dtype=np.float64)
        self._lib.fennec_blurKernel(float(sigma), k.ctypes.data_as(_f64p))
        return r, k

    def precomputeWeights(self, dst_size: int, src_size: int):
        off = np.zeros(dst_size + 1, dtype=np.int32)
        n = self._lib.fennec_precomputeWeights(dst_size, src_size, off.ctypes.data_as(_i32p), None, None)
        idx = np.zeros(max(n, 1), dtype=np.int32)
        wt = np.zeros(max(n, 1), dtype=np.float64)
        self._lib.fennec_precomputeWeights(dst_size, src_size, off.ctypes.data_as(_i32p),
                                           idx.ctypes.data_as(_i32p), wt.ctypes.data_as(_f64p))
        return off, idx[:max(n, 0)], wt[:max(n, 0)]

    def smartResizeDims(self, w, h, max_w, max_h):
        dw, dh = C.c_int(), C.c_int()
        r = self._lib.fennec_smartResizeDims(w, h, max_w, max_h, C.byref(dw), C.byref(dh))
        return bool(r), dw.value, dh.value

    def ssimFastDims(self, w, h):
        nw, nh = C.c_int(), C.c_int()
        r = self._lib.fennec_ssimFastDims(w, h, C.byref(nw), C.byref(nh))
        return bool(r), nw.value, nh.value

    # -- ssim.go ------------------------------------------------------------------------
    def SSIM(self, img1, img2) -> float:
        """ssim.go:24 -- full-resolution SSIM; a differently sized img2 is Lanczos-resized."""
        a, b = self._pair(img1, img2)
        out = C.c_double()
        with self._ordered(img1, img2):
            self._chk(self._lib.fennec_SSIM(self._h, a.space, a.ptr, a.stride, a.w, a.h, b.ptr, b.stride,
                                            b.w, b.h, C.byref(out)), "SSIM")
        return out.value

    def pixelSSIM(self, pix_a, pix_b, w: int, h: int) -> float:
        """ssim.go:169 over the two FLAT Pix slices as Go holds them (1-D uint8 numpy arrays or device tensors): for a
        SubImage that is everything up to the end of the parent's buffer, which SSIM / SSIMFast cannot know."""
        def flat(x):
            if _is_torch(x):
                return FNX_DEVICE, x.data_ptr(), int(x.numel())
            x = np.ascontiguousarray(x, dtype=np.uint8).reshape(-1)
            keep.append(x)
            return FNX_HOST, x.ctypes.data, int(x.size)
        keep = []
        (sa, pa, na), (sb, pb, nb) = flat(pix_a), flat(pix_b)
        if sa != sb:
            raise FennecError("both Pix slices must live in the same space")
        out = C.c_double()
        with self._ordered(pix_a, pix_b):
            self._chk(self._lib.fnx_pixel_ssim(self._h, sa, C.cast(pa, _u8p), na, C.cast(pb, _u8p), nb, int(w), int(h),
                                               C.byref(out)), "pixelSSIM")
        return out.value

    def ssim_enqueue(self, img1, img2, window=None):
        """fnx_ssim_enqueue: SSIM of a device pair of equal dims, result fetched later with fetch_result()
        (FIFO, at most 4 unfetched).  The caller keeps img1 / img2 alive until then."""
        a, b = self._pair(img1, img2)
        if a.space != FNX_DEVICE or (a.w, a.h) != (b.w, b.h):
            raise FennecError("ssim_enqueue takes two device tensors of equal dims")
        k, pk = _f64(self.gaussianKernel() if window is None else window)
        with self._ordered(img1, img2):
            self._chk(self._lib.fnx_ssim_enqueue(self._h, a.ptr, a.stride, b.ptr, b.stride, a.w, a.h, pk), "fnx_ssim_enqueue")

    def msssim_enqueue(self, img1, img2):
        """fennec_MSSSIM_enqueue: MSSSIM of a device pair (img2 resized to img1's dims first when they differ,
        ssim.go:320-322) through the result FIFO; fetch_result() returns the value.  The caller keeps both alive."""
        a, b = self._pair(img1, img2)
        if a.space != FNX_DEVICE:
            raise FennecError("msssim_enqueue takes two device tensors")
        with self._ordered(img1, img2):
            self._chk(self._lib.fennec_MSSSIM_enqueue(self._h, a.ptr, a.stride, a.w, a.h, b.ptr, b.stride, b.w, b.h),
                      "fennec_MSSSIM_enqueue")

    def msssim_batch_enqueue(self, imgs1, imgs2):
        """fennec_MSSSIM_batch_enqueue: n device pairs (every imgs1[i] of one size, every imgs2[i] of one size) as ONE FIFO
        entry; the imgs2 are resized to imgs1's dims by one batched lanczosResize when the dims differ.  fetch_results(n)."""
        va, vb = [_Img(t) for t in imgs1], [_Img(t) for t in imgs2]
        n = len(va)
        if n == 0 or n != len(vb) or any(v.space != FNX_DEVICE for v in va + vb):
            raise FennecError("msssim_batch_enqueue takes two equally long, non-empty lists of device tensors")
        a0, b0 = va[0], vb[0]
        if any((v.w, v.h, v.stride) != (a0.w, a0.h, a0.stride) for v in va) or any((v.w, v.h, v.stride) != (b0.w, b0.h, b0.stride) for v in vb):
            raise FennecError("every image of a side must share width, height and stride")
        pa = (C.c_void_p * n)(*[v.ptr for v in va])
        pb = (C.c_void_p * n)(*[v.ptr for v in vb])
        with self._ordered(*imgs1, *imgs2):
            self._chk(self._lib.fennec_MSSSIM_batch_enqueue(self._h, n, pa, a0.stride, a0.w, a0.h, pb, b0.stride, b0.w, b0.h),
                      "fennec_MSSSIM_batch_enqueue")

    def ssim_batch_enqueue(self, imgs1, imgs2, window=None):
        """fnx_ssim_batch_enqueue: SSIM of n same-sized device pairs in one launch; fetch_results(n)."""
        va, vb = [_Img(t) for t in imgs1], [_Img(t) for t in imgs2]
        n = len(va)
        if n == 0 or n != len(vb) or any(v.space != FNX_DEVICE for v in va + vb):
            raise FennecError("ssim_batch_enqueue takes two equally long, non-empty lists of device tensors")
        a0, b0 = va[0], vb[0]
        if any((v.w, v.h, v.stride) != (a0.w, a0.h, a0.stride) for v in va) or any((v.w, v.h, v.stride) != (a0.w, a0.h, b0.stride) for v in vb):
            raise FennecError("every image of a batch must share width and height, and every image of a side its stride")
        pa = (C.c_void_p * n)(*[v.ptr for v in va])
        pb = (C.c_void_p * n)(*[v.ptr for v in vb])
        k, pk = _f64(self.gaussianKernel() if window is None else window)
        with self._ordered(*imgs1, *imgs2):
            self._chk(self._lib.fnx_ssim_batch_enqueue(self._h, n, pa, a0.stride, pb, b0.stride, a0.w, a0.h, pk), "fnx_ssim_batch_enqueue")

    def sharpen_batch(self, imgs, strength: float, adaptive: bool = False, outs=None):
        """Sharpen / AdaptiveSharpen (effects.go:10 / :49) of n same-sized device images in one launch (fnx_*_sharpen_batch);
        strength <= 0 or images under 3 x 3 come back as they are, as from the single calls.  Enqueued."""
        views = [_Img(t) for t in imgs]
        if not views or any(v.space != FNX_DEVICE for v in views):
            raise FennecError("sharpen_batch takes a non-empty list of device tensors")
        v0 = views[0]
        if any((v.w, v.h, v.stride) != (v0.w, v0.h, v0.stride) for v in views):
            raise FennecError("every image of a batch must share width, height and stride")
        if strength <= 0 or v0.w < 3 or v0.h < 3:
            return list(imgs)
        amount = 1.0 + min(strength, 1.0) * (2.0 if adaptive else 1.5)
        if outs is None:
            outs = [v0.like(v0.w, v0.h) for _ in views]
        ov = [_Img(t) for t in outs]
        n = len(views)
        srcs = (C.c_void_p * n)(*[v.ptr for v in views])
        dsts = (C.c_void_p * n)(*[v.ptr for v in ov])
        fn = self._lib.fnx_adaptive_sharpen_batch if adaptive else self._lib.fnx_sharpen_batch
        with self._ordered(*imgs, *outs):
            self._chk(fn(self._h, n, srcs, v0.stride, v0.w, v0.h, C.c_double(amount), dsts, ov[0].stride), "sharpen_batch")
        return outs

    def fetch_results(self, n: int) -> np.ndarray:
        """The oldest FIFO entry's n values (fnx_results_fetch)."""
        out = np.empty(n, dtype=np.float64)
        self._chk(self._lib.fnx_results_fetch(self._h, n, out.ctypes.data_as(_f64p)), "fnx_results_fetch")
        return out

    def fetch_result(self) -> float:
        """The oldest enqueued scalar result (fnx_results_fetch)."""
        out = C.c_double()
        self._chk(self._lib.fnx_results_fetch(self._h, 1, C.byref(out)), "fnx_results_fetch")
        return out.value

    def SSIMFast(self, img1, img2) -> float:
        """ssim.go:48 -- SSIM on <=512 px box-downsampled copies."""
        a, b = self._pair(img1, img2)
        out = C.c_double()
        with self._ordered(img1, img2):
            self._chk(self._lib.fennec_SSIMFast(self._h, a.space, a.ptr, a.stride, b.ptr, b.stride, a.w, a.h,
                                                C.byref(out)), "SSIMFast")
        return out.value

    def MSSSIM(self, img1, img2) -> float:
        """ssim.go:313 -- 5-level multi-scale SSIM.  A strided / cropped view is treated as a Go SubImage: the pyramid starts from
        the first 4*w*h flat bytes (convert.go:16), not from the view's rows; pass a contiguous copy for row-wise semantics."""
        a, b = self._pair(img1, img2)
        out = C.c_double()
        with self._ordered(img1, img2):
            self._chk(self._lib.fennec_MSSSIM(self._h, a.space, a.ptr, a.stride, a.w, a.h, b.ptr, b.stride,
                                              b.w, b.h, C.byref(out)), "MSSSIM")
        return out.value

    def msssim_levels(self, img1, img2, window=None):
        """fnx_msssim with per-level SSIMFast values (equal dims)."""
        a, b = self._pair(img1, img2)
        k, pk = _f64(self.gaussianKernel() if window is None else window)
        out = C.c_double()
        lv = np.empty(5, dtype=np.float64)
        with self._ordered(img1, img2):
            self._chk(self._lib.fnx_msssim(self._h, a.space, a.ptr, a.stride, b.ptr, b.stride, a.w, a.h, pk,
                                           C.byref(out), lv.ctypes.data_as(_f64p)), "fnx_msssim")
        return out.value, lv

    def boxDownsample(self, img, dstW: int, dstH: int, to_host: bool = False):
        """ssim.go:244.  to_host=True with a device image (FNX_DEVICE_SRC): the result is a numpy array and the
        call returns when it is filled -- the scale searches of targetsize.go:240-313 keep ONE source
        resident and fetch a small downsample per iteration."""
        s = _Img(img)
        if s.w <= 0 or s.h <= 0 or dstW <= 0 or dstH <= 0:
            return self._out_for(s, 0, 0, to_host)[1]
        space, dst = self._out_for(s, dstW, dstH, to_host)
        d = _Img(dst)
        with self._ordered(img, dst):
            self._chk(self._lib.fennec_boxDownsample(self._h, space, s.ptr, s.stride, s.w, s.h, d.ptr,
                                                     d.stride, dstW, dstH), "boxDownsample")
        return dst

    @staticmethod
    def _out_for(s: "_Img", w: int, h: int, to_host: bool):
        """(space, fresh destination) for an image -> image op on `s`."""
        if to_host and s.space == FNX_DEVICE:
            return FNX_DEVICE_SRC, np.empty((h, w, 4), dtype=np.uint8)
        return s.space, s.like(w, h)

    def ssim_fast_prepare(self, img):
        s = _Img(img)
        p = C.c_void_p()
        with self._ordered(img):
            self._chk(self._lib.fnx_ssim_fast_prepare(self._h, s.space, s.ptr, s.stride, s.w, s.h, C.byref(p)),
                      "fnx_ssim_fast_prepare")
        return _Prepared(self, p, s.w, s.h)

    # -- the JPEG quantisation round trip (first slice) -----------------------
    def jpeg_roundtrip(self, img, quality: int):
        """toNRGBARef(jpeg.Decode(jpeg.Encode(img, quality))) as far as the pixels go (fnx_jpeg_roundtrip): baseline
        4:2:0 with Go's image/jpeg arithmetic, no entropy coding."""
        s = _Img(img)
        dst = s.like(s.w, s.h)
        d = _Img(dst)
        with self._ordered(img, dst):
            self._chk(self._lib.fnx_jpeg_roundtrip(self._h, s.space, s.ptr, s.stride, s.w, s.h, int(quality), d.ptr, d.stride),
                      "fnx_jpeg_roundtrip")
        return dst

    def jpeg_encode(self, img, quality: int) -> bytes:
        """jpeg.Encode(img, &jpeg.Options{Quality: quality}) on the device (fnx_jpeg_encode) -> the file's bytes."""
        s = _Img(img)
        cap = 4096 + (s.w * s.h * 3) // 2
        n = C.c_size_t(0)
        with self._ordered(img):
            for _ in range(2):
                buf = np.empty(cap, dtype=np.uint8)
                rc = self._lib.fnx_jpeg_encode(self._h, s.space, s.ptr, s.stride, s.w, s.h, int(quality), buf.ctypes.data_as(_u8p),
                                               cap, C.byref(n))
                if rc == FNX_OK:
                    return buf[:n.value].tobytes()
                if n.value <= cap:
                    self._chk(rc, "fnx_jpeg_encode")
                cap = n.value
        self._chk(rc, "fnx_jpeg_encode")

    def jpeg_encoded_size(self, img, quality: int) -> int:
        """len(jpeg.Encode(img, quality)) without fetching the bytes (fnx_jpeg_encode's size query)."""
        s = _Img(img)
        n = C.c_size_t(0)
        with self._ordered(img):
            self._chk(self._lib.fnx_jpeg_encode(self._h, s.space, s.ptr, s.stride, s.w, s.h, int(quality), None, 0, C.byref(n)),
                      "fnx_jpeg_encode")
        return int(n.value)

    def jpeg_size_search(self, img, target_bytes: int, skip_ssim: bool = False, window=None):
        """jpegQualitySearchOpt (targetsize.go:125-176) on the device -> (bytes, quality, ssim, steps), or None when no
        quality fits target_bytes."""
        s = _Img(img)
        k, pk = _f64(self.gaussianKernel() if window is None else window)
        cap = max(4096, int(target_bytes) + 16)
        buf = np.empty(cap, dtype=np.uint8)
        n, q, st, v = C.c_size_t(0), C.c_int(), C.c_int(), C.c_double()
        with self._ordered(img):
            rc = self._chk(self._lib.fnx_jpeg_size_search(self._h, s.space, s.ptr, s.stride, s.w, s.h, int(target_bytes), int(bool(skip_ssim)), pk,
                                                          buf.ctypes.data_as(_u8p), cap, C.byref(n), C.byref(q), C.byref(v), C.byref(st)),
                           "fnx_jpeg_size_search")
        if rc != FNX_OK:
            return None
        return buf[:n.value].tobytes(), q.value, v.value, st.value

    def jpeg_compress(self, img, target_ssim: float, window=None):
        """compressJPEGOptimal on the device (fnx_jpeg_compress): search + the winning file -> (bytes, quality, ssim, steps)."""
        s = _Img(img)
        k, pk = _f64(self.gaussianKernel() if window is None else window)
        cap = 4096 + (s.w * s.h * 3) // 2
        n, q, st, v = C.c_size_t(0), C.c_int(), C.c_int(), C.c_double()
        with self._ordered(img):
            for _ in range(2):
                buf = np.empty(cap, dtype=np.uint8)
                rc = self._lib.fnx_jpeg_compress(self._h, s.space, s.ptr, s.stride, s.w, s.h, float(target_ssim), pk,
                                                 buf.ctypes.data_as(_u8p), cap, C.byref(n), C.byref(q), C.byref(v), C.byref(st))
                if rc == FNX_OK:
                    return buf[:n.value].tobytes(), q.value, v.value, st.value
                if n.value <= cap:
                    break
                cap = n.value
        self._chk(rc, "fnx_jpeg_compress")

    @staticmethod
    def jpeg_progressive_coefficients(data: bytes):
        """Host only (fnx_jpeg_progressive_coefficients): a progressive file's quantised coefficients over all its scans, as the
        device's IDCT launch gets them -> (coef [blocks, 64] int16 in natural order, blocks MCU by MCU, (w, h), ratio)."""
        L = load_library()
        buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
        nb, w, h, ratio = C.c_size_t(0), C.c_int(), C.c_int(), C.c_int()

        def chk(rc):
            if rc == FNX_ERR_UNSUPPORTED:
                raise FennecUnsupported(L.fnx_last_error().decode())
            if rc < 0:
                raise FennecError(f"fnx_jpeg_progressive_coefficients failed ({rc}): {L.fnx_last_error().decode()}")
        chk(L.fnx_jpeg_progressive_coefficients(buf.ctypes.data_as(_u8p), len(data), None, 0, C.byref(nb), C.byref(w), C.byref(h), C.byref(ratio)))
        coef = np.empty((nb.value, 64), dtype=np.int16)
        chk(L.fnx_jpeg_progressive_coefficients(buf.ctypes.data_as(_u8p), len(data), coef.ctypes.data_as(C.POINTER(C.c_int16)), nb.value,
                                                C.byref(nb), C.byref(w), C.byref(h), C.byref(ratio)))
        return coef, (w.value, h.value), ratio.value

    @staticmethod
    def jpeg_parse(data: bytes):
        """jpeg_decode_config without a ctx (the segment parser is host code): (w, h), FennecUnsupported, or FennecError."""
        L = load_library()
        buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
        w, h = C.c_int(), C.c_int()
        rc = L.fnx_jpeg_decode(None, buf.ctypes.data_as(_u8p), len(data), FNX_HOST, None, 0, C.byref(w), C.byref(h))
        if rc == FNX_ERR_UNSUPPORTED:
            raise FennecUnsupported(L.fnx_last_error().decode())
        if rc < 0:
            raise FennecError(f"fnx_jpeg_decode failed ({rc}): {L.fnx_last_error().decode()}")
        return w.value, h.value

    def jpeg_decode_config(self, data: bytes):
        """jpeg.DecodeConfig as far as the device decoder goes: (w, h); FennecUnsupported for files it does not take."""
        buf = np.frombuffer(data, dtype=np.uint8)
        w, h = C.c_int(), C.c_int()
        self._chk(self._lib.fnx_jpeg_decode(self._h, buf.ctypes.data_as(_u8p), len(data), FNX_HOST, None, 0, C.byref(w), C.byref(h)),
                  "fnx_jpeg_decode")
        return w.value, h.value

    def jpeg_decode(self, data: bytes, device: bool = False):
        """toNRGBARef(jpeg.Decode(data)) on the device (fnx_jpeg_decode): the file's bytes go up, an (h, w, 4) uint8 array
        comes back -- or, with device=True, a torch tensor that stays on the ctx's device."""
        w, h = self.jpeg_decode_config(data)
        buf = np.frombuffer(data, dtype=np.uint8)
        if device:
            import torch
            dst = torch.empty((h, w, 4), dtype=torch.uint8, device=f"cuda:{self.device}")
        else:
            dst = np.empty((h, w, 4), dtype=np.uint8)
        d = _Img(dst)
        with self._ordered(dst):
            self._chk(self._lib.fnx_jpeg_decode(self._h, buf.ctypes.data_as(_u8p), len(data), d.space, d.ptr, d.stride, C.byref(C.c_int()),
                                                C.byref(C.c_int())), "fnx_jpeg_decode")
        return dst

    def jpeg_recompress(self, data: bytes, target_ssim: float, window=None):
        """CompressBatch's item body for a JPEG source on the device (fnx_jpeg_recompress): decode, quality search, the
        winner's file -> (bytes, quality, ssim, steps, (w, h))."""
        src = np.frombuffer(data, dtype=np.uint8)
        k, pk = _f64(self.gaussianKernel() if window is None else window)
        cap = len(data) + 4096       # a search ends below the quality a camera wrote; a larger file asks again with its size
        n, q, st, v, w, h = C.c_size_t(0), C.c_int(), C.c_int(), C.c_double(), C.c_int(), C.c_int()
        for _ in range(2):
            buf = np.empty(cap, dtype=np.uint8)
            rc = self._lib.fnx_jpeg_recompress(self._h, src.ctypes.data_as(_u8p), len(data), float(target_ssim), pk, buf.ctypes.data_as(_u8p),
                                               cap, C.byref(n), C.byref(q), C.byref(v), C.byref(st), C.byref(w), C.byref(h))
            if rc == FNX_OK:
                return buf[:n.value].tobytes(), q.value, v.value, st.value, (w.value, h.value)
            if n.value <= cap:
                break
            cap = n.value
        self._chk(rc, "fnx_jpeg_recompress")

    def compress_file_jpeg(self, data: bytes, target_ssim: float, orient: int = 1, max_w: int = 0, max_h: int = 0, auto_format: bool = False):
        """CompressFile for a JPEG source in standard mode, every pixel stage on the device (fennec_CompressFileJPEG): decode,
        ApplyOrientation(orient), smartResize(max_w, max_h), analyzeFormat (auto_format), compressJPEGOptimal ->
        (bytes, quality, ssim, steps, original (w, h), final (w, h)); bytes is None when analyzeFormat chose PNG."""
        src = np.frombuffer(data, dtype=np.uint8)
        o = FileOptions(int(orient), int(max_w), int(max_h), 1 if auto_format else 0, float(target_ssim))
        cap = max(4096, len(data) + 4096)
        n, q, st, v = C.c_size_t(0), C.c_int(), C.c_int(), C.c_double()
        dims = (C.c_int * 4)()
        for _ in range(2):
            buf = np.empty(cap, dtype=np.uint8)
            rc = self._lib.fennec_CompressFileJPEG(self._h, src.ctypes.data_as(_u8p), len(data), C.byref(o), buf.ctypes.data_as(_u8p), cap,
                                                   C.byref(n), C.byref(q), C.byref(v), C.byref(st), dims)
            if rc == FNX_OK:
                return buf[:n.value].tobytes(), q.value, v.value, st.value, (dims[0], dims[1]), (dims[2], dims[3])
            if rc == FNX_NOOP:
                return None, 0, 1.0, 0, (dims[0], dims[1]), (dims[2], dims[3])
            if n.value <= cap:
                break
            cap = n.value
        self._chk(rc, "fennec_CompressFileJPEG")

    def jpeg_quality_search(self, img, target_ssim: float, window=None):
        """compressJPEGOptimal's binary search with every candidate round-tripped and scored on the device
        (fnx_jpeg_quality_search) -> (quality, ssim, steps, found)."""
        s = _Img(img)
        k, pk = _f64(self.gaussianKernel() if window is None else window)
        q, st, v = C.c_int(), C.c_int(), C.c_double()
        with self._ordered(img):
            rc = self._chk(self._lib.fnx_jpeg_quality_search(self._h, s.space, s.ptr, s.stride, s.w, s.h, float(target_ssim), pk,
                                                             C.byref(q), C.byref(v), C.byref(st)), "fnx_jpeg_quality_search")
        return q.value, v.value, st.value, rc == FNX_OK

    # -- effects.go ---------------------------------------------------------------------
    def GaussianBlur(self, img, sigma: float, exact: bool | None = False, kernel=None, keep_box_sums: bool = False):
        """effects.go:146.  sigma <= 0 returns `img` itself (same pointer).  exact=False: fast fp32 kernel
        (<= 1 LSB on <= 0.1 % of samples); exact=True: bit-exact fp64 kernels; exact=None: what the
        reference-named mirror fennec_GaussianBlur picks (exact for host images, fast for device tensors).
        keep_box_sums (device tensors, FNX_BLUR_KEEP_BOX_SUMS): the next call on this ctx is SSIMFast(img, result) and nothing
        writes either image in between -- that call then reads neither of them again."""
        if sigma <= 0:
            return img
        s = _Img(img)
        dst = s.like(s.w, s.h)
        d = _Img(dst)
        with self._ordered(img, dst):
            if kernel is None and exact is None:
                rc = self._lib.fennec_GaussianBlur(self._h, s.space, s.ptr, s.stride, s.w, s.h, float(sigma),
                                                   d.ptr, d.stride)
            else:
                if kernel is None:
                    radius, kernel = self.blurKernel(sigma)
                else:
                    radius = (len(kernel) - 1) // 2
                k, pk = _f64(kernel)
                rc = self._lib.fnx_gaussian_blur(self._h, s.space, s.ptr, s.stride, s.w, s.h, pk, radius,
                                                 (FNX_BLUR_EXACT if exact else FNX_BLUR_FAST) | (FNX_BLUR_KEEP_BOX_SUMS if keep_box_sums else 0),
                                                 d.ptr, d.stride)
            self._chk(rc, "GaussianBlur")
        return dst

    def GaussianBlurSSIMFast(self, img, sigma: float, exact: bool = True, window=None, out=None):
        """GaussianBlur(img, sigma) and SSIMFast(img, blurred) in ONE call (fnx_gaussian_blur_ssim_fast): a host image crosses
        PCIe once each way.  -> (blurred, ssim); the bytes and the score of the two separate calls.  out: the image to write
        (same kind and size as img) instead of a new one."""
        if sigma <= 0:
            raise FennecError("GaussianBlurSSIMFast: sigma <= 0 (GaussianBlur would return the image itself)")
        s = _Img(img)
        dst = s.like(s.w, s.h) if out is None else out
        d = _Img(dst)
        if (d.w, d.h, d.space) != (s.w, s.h, s.space):
            raise FennecError("GaussianBlurSSIMFast: out must be an image of img's size in img's memory space")
        radius, kernel = self.blurKernel(sigma)
        k, pk = _f64(kernel)
        wk, pw = _f64(self.gaussianKernel() if window is None else window)
        out = C.c_double()
        with self._ordered(img, dst):
            self._chk(self._lib.fnx_gaussian_blur_ssim_fast(self._h, s.space, s.ptr, s.stride, s.w, s.h, pk, radius,
                                                            FNX_BLUR_EXACT if exact else FNX_BLUR_FAST, d.ptr, d.stride, pw, C.byref(out)),
                      "GaussianBlurSSIMFast")
        return dst, float(out.value)

    def blur3x3(self, img):
        """effects.go:116 gaussianBlur3x3.  A strided / cropped view is a Go SubImage here (as for Sharpen / AdaptiveSharpen):
        borders and alpha come from the first 4*w*h flat bytes (effects.go:120), as in the reference; pass a contiguous copy
        for row-wise semantics."""
        s = _Img(img)
        dst = s.like(s.w, s.h)
        d = _Img(dst)
        with self._ordered(img, dst):
            self._chk(self._lib.fnx_blur3x3(self._h, s.space, s.ptr, s.stride, s.w, s.h, d.ptr, d.stride), "blur3x3")
        return dst

    def _sharpen(self, fn, name, img, strength):
        s = _Img(img)
        dst = s.like(s.w, s.h)
        d = _Img(dst)
        with self._ordered(img, dst):
            rc = self._chk(fn(self._h, s.space, s.ptr, s.stride, s.w, s.h, float(strength), d.ptr, d.stride), name)
        return img if rc == FNX_NOOP else dst

    def sharpen_amount(self, img, amount: float, adaptive: bool = False):
        """fnx_sharpen / fnx_adaptive_sharpen with the unsharp `amount` itself (1 + 1.5 s / 1 + 2 s in the
        reference, effects.go:24,63): the kernel-level entry points, for callers and tests."""
        s = _Img(img)
        dst = s.like(s.w, s.h)
        d = _Img(dst)
        fn = self._lib.fnx_adaptive_sharpen if adaptive else self._lib.fnx_sharpen
        with self._ordered(img, dst):
            self._chk(fn(self._h, s.space, s.ptr, s.stride, s.w, s.h, float(amount), d.ptr, d.stride), "sharpen_amount")
        return dst

    def Sharpen(self, img, strength: float):
        """effects.go:10.  strength <= 0 or an image under 3x3 returns `img` itself.  Strided views: see blur3x3."""
        return self._sharpen(self._lib.fennec_Sharpen, "Sharpen", img, strength)

    def AdaptiveSharpen(self, img, strength: float):
        """effects.go:49.  Strided views: see blur3x3."""
        return self._sharpen(self._lib.fennec_AdaptiveSharpen, "AdaptiveSharpen", img, strength)

    # -- resize.go ----------------------------------------------------------------------
    def lanczosResize(self, img, dstW: int, dstH: int, to_host: bool = False):
        """resize.go:37.  to_host: see boxDownsample."""
        s = _Img(img)
        if s.w <= 0 or s.h <= 0 or dstW <= 0 or dstH <= 0:
            return self._out_for(s, 0, 0, to_host)[1]
        space, dst = self._out_for(s, dstW, dstH, to_host)
        d = _Img(dst)
        with self._ordered(img, dst):
            self._chk(self._lib.fennec_lanczosResize(self._h, space, s.ptr, s.stride, s.w, s.h, d.ptr,
                                                     d.stride, dstW, dstH), "lanczosResize")
        return dst

    def lanczosResizeBatch(self, imgs, dstW: int, dstH: int, outs=None):
        """n same-sized device images through ONE set of launches (fennec_lanczosResizeBatch): the bytes of n lanczosResize
        calls; enqueued (sync() or a later blocking call waits)."""
        views = [_Img(t) for t in imgs]
        if not views or any(v.space != FNX_DEVICE for v in views):
            raise FennecError("lanczosResizeBatch takes a non-empty list of device tensors")
        v0 = views[0]
        if any((v.w, v.h, v.stride) != (v0.w, v0.h, v0.stride) for v in views):
            raise FennecError("every image of a batch must share width, height and stride")
        if v0.w <= 0 or v0.h <= 0 or dstW <= 0 or dstH <= 0:
            return [v0.like(0, 0) for _ in views]
        if outs is None:
            outs = [v0.like(dstW, dstH) for _ in views]
        ov = [_Img(t) for t in outs]
        if len(ov) != len(views) or any((o.w, o.h, o.stride, o.space) != (dstW, dstH, ov[0].stride, FNX_DEVICE) for o in ov):
            raise FennecError("outs must hold one device image of dstW x dstH per input, all of one stride")
        n = len(views)
        srcs = (C.c_void_p * n)(*[v.ptr for v in views])
        dsts = (C.c_void_p * n)(*[v.ptr for v in ov])
        with self._ordered(*imgs, *outs):
            self._chk(self._lib.fennec_lanczosResizeBatch(self._h, n, srcs, v0.stride, v0.w, v0.h, dsts, ov[0].stride, dstW, dstH),
                      "lanczosResizeBatch")
        return outs

    def smartResize(self, img, maxW: int, maxH: int):
        """resize.go:12.  Returns `img` itself when it already fits."""
        s = _Img(img)
        r, dw, dh = self.smartResizeDims(s.w, s.h, maxW, maxH)
        return self.lanczosResize(img, dw, dh) if r else img

    def resize_pass(self, img, dst_size: int, vertical: bool, table=None):
        """resizeH / resizeV (resize.go:77,121) with an explicit CSR tap table."""
        s = _Img(img)
        src_size = s.h if vertical else s.w
        off, idx, wt = table if table is not None else self.precomputeWeights(dst_size, src_size)
        off, po = _i32(off); idx, pi = _i32(idx); wt, pw = _f64(wt)
        dst = s.like(s.w, dst_size) if vertical else s.like(dst_size, s.h)
        d = _Img(dst)
        fn = self._lib.fnx_resize_v if vertical else self._lib.fnx_resize_h
        with self._ordered(img, dst):
            self._chk(fn(self._h, s.space, s.ptr, s.stride, s.w, s.h, po, pi, pw, d.ptr, d.stride, dst_size),
                      "resize_pass")
        return dst

    def lanczos_resize_tables(self, img, dst_w: int, dst_h: int, table_h, table_v):
        """fnx_lanczos_resize: lanczosResize (resize.go:37-53) with the CALLER's two CSR tap tables (what a Go caller
        passes: its own precomputeWeights output) -- both passes, one launch where the tables allow it."""
        s = _Img(img)
        oh, ih, wh = table_h
        ov, iv, wv = table_v
        oh, poh = _i32(oh); ih, pih = _i32(ih); wh, pwh = _f64(wh)
        ov, pov = _i32(ov); iv, piv = _i32(iv); wv, pwv = _f64(wv)
        dst = s.like(dst_w, dst_h)
        d = _Img(dst)
        with self._ordered(img, dst):
            self._chk(self._lib.fnx_lanczos_resize(self._h, s.space, s.ptr, s.stride, s.w, s.h, poh, pih, pwh, pov, piv, pwv,
                                                   d.ptr, d.stride, dst_w, dst_h), "fnx_lanczos_resize")
        return dst

    # -- exif.go ------------------------------------------------------------------------
    def ApplyOrientation(self, img, orient: int):
        """exif.go:178.  Orientation 0, 1 and unknown values return `img` itself."""
        orient = int(orient)
        if orient < 2 or orient > 8:
            return img
        s = _Img(img)
        ow, oh = (s.h, s.w) if orient >= 5 else (s.w, s.h)
        dst = s.like(ow, oh)
        d = _Img(dst)
        with self._ordered(img, dst):
            self._chk(self._lib.fennec_ApplyOrientation(self._h, s.space, s.ptr, s.stride, s.w, s.h, orient,
                                                        d.ptr, d.stride), "ApplyOrientation")
        return dst

    # -- Analyze (analyze.go) ---------------------------------------------------------------
    @staticmethod
    def _stats_dict(st: ImageStats) -> dict:
        return {name: getattr(st, name) for name, _ in ImageStats._fields_ if name != "pad"}

    @staticmethod
    def _analysis_dict(a: Analysis) -> dict:
        d = {name: getattr(a, name) for name, _ in Analysis._fields_ if name not in ("pad", "histogram")}
        d["histogram"] = np.array(a.histogram[:], dtype=np.uint64)
        return d

    def Analyze(self, img) -> dict:
        """Analyze (analyze.go:26-124) -> dict with the ImageStats field names."""
        v = _Img(img)
        st = ImageStats()
        with self._ordered(img):
            self._chk(self._lib.fennec_Analyze(self._h, v.space, v.ptr, v.stride, v.w, v.h, C.byref(st)), "Analyze")
        return self._stats_dict(st)

    def analyze_raw(self, img) -> dict:
        """fnx_analyze: the device-side accumulators (histogram, sums, counts)."""
        v = _Img(img)
        a = Analysis()
        with self._ordered(img):
            self._chk(self._lib.fnx_analyze(self._h, v.space, v.ptr, v.stride, v.w, v.h, C.byref(a)), "fnx_analyze")
        return self._analysis_dict(a)

    def plan_analyze_batch(self, imgs):
        """Pre-marshalled fnx_analyze_batch over n same-sized device images: run() -> list of ImageStats dicts."""
        views = self._batch_views(imgs)
        n, w, h, st = len(views), views[0].w, views[0].h, views[0].stride
        srcs = (C.c_void_p * n)(*[v.ptr for v in views])
        res = (Analysis * n)()
        ctx, lib = self, self._lib

        class _Plan:
            def __init__(p):
                p._keep = (imgs, srcs, res)
                p.raw = res

            def run(p):
                ctx._chk(lib.fnx_analyze_batch(ctx._h, n, srcs, st, w, h, res), "AnalyzeBatch")
                return res

            def stats(p):
                out = []
                for k in range(n):
                    s = ImageStats()
                    lib.fennec_statsFromAnalysis(C.byref(res[k]), w, h, C.byref(s))
                    out.append(ctx._stats_dict(s))
                return out
        return _Plan()

    def AnalyzeBatch(self, imgs):
        plan = self.plan_analyze_batch(imgs)
        plan.run()
        return plan.stats()

    def isOpaque(self, img) -> bool:
        v = _Img(img)
        o = C.c_int(0)
        with self._ordered(img):
            self._chk(self._lib.fennec_isOpaque(self._h, v.space, v.ptr, v.stride, v.w, v.h, C.byref(o)), "isOpaque")
        return bool(o.value)

    def isGrayscale(self, img) -> bool:
        v = _Img(img)
        o = C.c_int(0)
        with self._ordered(img):
            self._chk(self._lib.fennec_isGrayscale(self._h, v.space, v.ptr, v.stride, v.w, v.h, C.byref(o)), "isGrayscale")
        return bool(o.value)

    @staticmethod
    def _planes(y, cb, cr):
        """(space, y ptr/stride, cb ptr, cr ptr, cstride, w, h) of numpy or torch uint8 planes."""
        def view(p):
            if p is None:
                return None, 0, 0
            if _is_torch(p):
                return p.data_ptr(), int(p.stride(0)) if p.shape[0] > 1 else int(p.shape[1]), FNX_DEVICE
            return p.ctypes.data, int(p.strides[0]) if p.shape[0] > 1 else int(p.shape[1]), FNX_HOST
        yp, ys, space = view(y)
        cbp, cs, _ = view(cb)
        crp, cs2, _ = view(cr)
        if cb is not None and cs != cs2:
            raise FennecError("Cb and Cr must share a stride (image.YCbCr.CStride)")
        return space, yp, ys, cbp, crp, cs, int(y.shape[1]), int(y.shape[0])

    def ycbcrToNRGBA(self, y, cb, cr, ratio: int):
        """toNRGBARef of an image.YCbCr (convert.go:22-64); cb = cr = None: image.Gray.  Planes: 2-D uint8."""
        space, yp, ys, cbp, crp, cs, w, h = self._planes(y, cb, cr)
        if space == FNX_DEVICE:
            import torch
            dst = torch.empty((h, w, 4), dtype=torch.uint8, device=y.device)
        else:
            dst = np.empty((h, w, 4), dtype=np.uint8)
        d = _Img(dst)
        with self._ordered(y, cb, cr, dst):
            self._chk(self._lib.fnx_ycbcr_to_nrgba(self._h, space, yp, ys, cbp, crp, cs, int(ratio), w, h, d.ptr, d.stride),
                      "ycbcrToNRGBA")
        return dst

    def applyPalette(self, img, palette, want_quantized: bool = True):
        """applyPalette (targetsize.go:488-527) [+ palettedToNRGBA]: -> (indices (h, w) uint8,
        quantized (h, w, 4) or None).  palette: (n, 4) uint8, opaque."""
        v = _Img(img)
        pal = np.ascontiguousarray(palette, dtype=np.uint8).reshape(-1, 4)
        if v.space == FNX_DEVICE:
            import torch
            idx = torch.empty((v.h, v.w), dtype=torch.uint8, device=img.device)
            iptr, istride = idx.data_ptr(), v.w
        else:
            idx = np.empty((v.h, v.w), dtype=np.uint8)
            iptr, istride = idx.ctypes.data, v.w
        q = v.like(v.w, v.h) if want_quantized else None
        qv = _Img(q) if want_quantized else None
        with self._ordered(img, idx, q):
            self._chk(self._lib.fnx_apply_palette(self._h, v.space, v.ptr, v.stride, v.w, v.h, pal.ctypes.data, len(pal),
                                                  iptr, istride, qv.ptr if qv else None, qv.stride if qv else 0),
                      "applyPalette")
        return idx, q

    # -- batched forms (device tensors) ---------------------------------------------------
    # Contract of the plan_* objects: creating a plan synchronises torch's current stream once (the inputs
    # exist from then on); run() / enqueue() only touch the stream the ctx launches on (its own unless a per-image
    # device call lent it torch's; use_own_stream() goes back), fetch() / run() of the scoring plans
    # block until the results -- and therefore every kernel queued before them -- are complete.  A caller
    # that rewrites the input tensors with torch between two runs must order that itself (ctx.sync() /
    # torch.cuda.synchronize()).  The convenience wrappers (GaussianBlurBatch, ...) order both ways.
    @staticmethod
    def _batch_views(imgs, what="images"):
        """_Img views of n device tensors that share (w, h, stride, device); raises FennecError otherwise."""
        views = [_Img(t) for t in imgs]
        if not views:
            raise FennecError(f"{what}: empty batch")
        if any(v.space != FNX_DEVICE for v in views):
            raise FennecError("batched ops take device tensors")
        v0 = views[0]
        for k, v in enumerate(views):
            if (v.w, v.h, v.stride) != (v0.w, v0.h, v0.stride) or v.obj.device != v0.obj.device:
                raise FennecError(f"{what}[{k}]: every image of a batch must share width, height, stride and device "
                                  f"(got {v.w}x{v.h} stride {v.stride} on {v.obj.device}, "
                                  f"expected {v0.w}x{v0.h} stride {v0.stride} on {v0.obj.device})")
        import torch
        torch.cuda.current_stream(v0.obj.device).synchronize()
        return views

    def GaussianBlurBatch(self, imgs, sigma: float, outs=None, exact: bool = False):
        """n same-sized device images, one launch per stage (enqueued; call sync() to wait)."""
        if sigma <= 0:
            return list(imgs)
        with self._ordered(*imgs):
            plan = self.plan_blur_batch(imgs, sigma, outs=outs, exact=exact)
            plan.run()
        return plan.outs

    def plan_blur_batch(self, imgs, sigma: float, outs=None, exact: bool = False, keep_box_sums: bool = False):
        """Pre-marshal a batched blur (pointer tables, kernel) so that run() is one C call --
        lets a caller bracket the launch tightly with events.  keep_box_sums (FNX_BLUR_KEEP_BOX_SUMS): the caller's next call on
        this ctx is the SSIMFast batch over (imgs[i], outs[i]) and nothing writes the images in between -- that call then reads
        neither full-size image again."""
        views = self._batch_views(imgs)
        w, h, st = views[0].w, views[0].h, views[0].stride
        if outs is None:
            outs = [views[0].like(w, h) for _ in views]
        oviews = self._batch_views(outs, "outs")
        n = len(views)
        if len(oviews) != n or (oviews[0].w, oviews[0].h) != (w, h):
            raise FennecError("outs must hold one image of the inputs' size per input")
        srcs = (C.c_void_p * n)(*[v.ptr for v in views])
        dsts = (C.c_void_p * n)(*[v.ptr for v in oviews])
        radius, kernel = self.blurKernel(sigma)
        k, pk = _f64(kernel)
        flags = (FNX_BLUR_EXACT if exact else FNX_BLUR_FAST) | (FNX_BLUR_KEEP_BOX_SUMS if keep_box_sums else 0)
        ctx, lib, ost = self, self._lib, oviews[0].stride

        class _Plan:
            def __init__(p):
                p.outs = outs
                p._keep = (imgs, outs, srcs, dsts, k)

            def run(p):
                ctx._chk(lib.fnx_gaussian_blur_batch(ctx._h, n, srcs, st, w, h, pk, radius, flags, dsts, ost),
                         "GaussianBlurBatch")
        return _Plan()

    def plan_ssim_fast_batch(self, imgs_a, imgs_b, window=None):
        """Pre-marshalled SSIMFastBatch: run() -> numpy array of n SSIM values (synchronises)."""
        va = self._batch_views(imgs_a, "imgs_a")
        vb = self._batch_views(imgs_b, "imgs_b")
        n = len(va)
        if n != len(vb) or (va[0].w, va[0].h) != (vb[0].w, vb[0].h):
            raise FennecError("batched ops take two equally long lists of equally sized device tensors")
        as_ = (C.c_void_p * n)(*[v.ptr for v in va])
        bs_ = (C.c_void_p * n)(*[v.ptr for v in vb])
        k, pk = _f64(self.gaussianKernel() if window is None else window)
        out = np.empty(n, dtype=np.float64)
        po = out.ctypes.data_as(_f64p)
        ctx, lib = self, self._lib
        sa, sb, w, h = va[0].stride, vb[0].stride, va[0].w, va[0].h

        class _Plan:
            def __init__(p):
                p._keep = (imgs_a, imgs_b, as_, bs_, k, out)

            def run(p):
                ctx._chk(lib.fnx_ssim_fast_batch(ctx._h, n, as_, sa, bs_, sb, w, h, pk, po), "SSIMFastBatch")
                return out

            def enqueue(p):
                """Queue the kernels only; results stay on the device until fetch()."""
                ctx._chk(lib.fnx_ssim_fast_batch_enqueue(ctx._h, n, as_, sa, bs_, sb, w, h, pk), "SSIMFastBatch")

            def fetch(p):
                ctx._chk(lib.fnx_results_fetch(ctx._h, n, po), "fnx_results_fetch")
                return out
        return _Plan()

    def SSIMFastBatch(self, imgs_a, imgs_b, window=None) -> np.ndarray:
        return self.plan_ssim_fast_batch(imgs_a, imgs_b, window).run().copy()

    def plan_blur_ssim_fast_batch(self, imgs, sigma: float, outs=None, exact: bool = False, window=None,
                                  kernel=None):
        """Pre-marshalled `outs[i] = GaussianBlur(imgs[i], sigma); ssim[i] = SSIMFast(imgs[i], outs[i])`
        in one pass over the pixels (fnx_gaussian_blur_ssim_fast_batch).  run() -> numpy array of n
        SSIM values (synchronises); enqueue()/fetch() split it.  exact=True: the blurred images are
        bit-exact (guarded kernel) and so are the planes the score is computed from.  `kernel` replaces
        blurKernel(sigma) with a caller-supplied odd-length 1-D kernel."""
        if sigma <= 0:
            raise FennecError("sigma <= 0 returns the source itself (effects.go:147): nothing to plan")
        views = self._batch_views(imgs)
        w, h, st = views[0].w, views[0].h, views[0].stride
        if outs is None:
            outs = [views[0].like(w, h) for _ in views]
        oviews = self._batch_views(outs, "outs")
        n = len(views)
        if len(oviews) != n or (oviews[0].w, oviews[0].h) != (w, h):
            raise FennecError("outs must hold one image of the inputs' size per input")
        srcs = (C.c_void_p * n)(*[v.ptr for v in views])
        dsts = (C.c_void_p * n)(*[v.ptr for v in oviews])
        if kernel is None:
            radius, kernel = self.blurKernel(sigma)
        else:
            kernel = np.asarray(kernel, dtype=np.float64)
            if kernel.ndim != 1 or len(kernel) % 2 == 0:
                raise FennecError("kernel must be a 1-D array of odd length")
            radius = len(kernel) // 2
        k, pk = _f64(kernel)
        kw, pw = _f64(self.gaussianKernel() if window is None else window)
        flags = FNX_BLUR_EXACT if exact else FNX_BLUR_FAST
        out = np.empty(n, dtype=np.float64)
        po = out.ctypes.data_as(_f64p)
        ctx, lib, ost = self, self._lib, oviews[0].stride

        class _Plan:
            def __init__(p):
                p.outs = outs
                p._keep = (imgs, outs, srcs, dsts, k, kw, out)

            def run(p):
                ctx._chk(lib.fnx_gaussian_blur_ssim_fast_batch(ctx._h, n, srcs, st, w, h, pk, radius, flags,
                                                               dsts, ost, pw, po), "GaussianBlur+SSIMFast batch")
                return out

            def enqueue(p):
                ctx._chk(lib.fnx_gaussian_blur_ssim_fast_batch_enqueue(ctx._h, n, srcs, st, w, h, pk, radius,
                                                                       flags, dsts, ost, pw),
                         "GaussianBlur+SSIMFast batch")

            def fetch(p):
                ctx._chk(lib.fnx_results_fetch(ctx._h, n, po), "fnx_results_fetch")
                return out
        return _Plan()

    def GaussianBlurSSIMFastBatch(self, imgs, sigma: float, outs=None, exact: bool = False, window=None,
                                  kernel=None):
        """-> (blurred images, numpy array of SSIMFast(imgs[i], blurred[i]))."""
        with self._ordered(*imgs):
            plan = self.plan_blur_ssim_fast_batch(imgs, sigma, outs=outs, exact=exact, window=window, kernel=kernel)
            vals = plan.run().copy()
        return plan.outs, vals


class _Prepared:
    """Reference side of an SSIM-guided quality search (compress.go:45-74)."""

    def __init__(self, ctx: Context, handle, w, h):
        self._ctx, self._p, self.w, self.h = ctx, handle, w, h

    def against(self, img, window=None) -> float:
        s = _Img(img)
        if (s.w, s.h) != (self.w, self.h):
            raise FennecError("candidate dims differ from the prepared reference")
        k, pk = _f64(self._ctx.gaussianKernel() if window is None else window)
        out = C.c_double()
        with self._ctx._ordered(img):
            self._ctx._chk(self._ctx._lib.fnx_ssim_fast_against(self._ctx._h, self._p, s.space, s.ptr, s.stride,
                                                                pk, C.byref(out)), "fnx_ssim_fast_against")
        return out.value

    def against_ycbcr(self, y, cb, cr, ratio: int, window=None) -> float:
        """SSIMFast(reference, toNRGBARef(decoded planes)) -- fnx_ssim_fast_against_ycbcr."""
        space, yp, ys, cbp, crp, cs, w, h = Context._planes(y, cb, cr)
        if (w, h) != (self.w, self.h):
            raise FennecError("candidate dims differ from the prepared reference")
        k, pk = _f64(self._ctx.gaussianKernel() if window is None else window)
        out = C.c_double(0.0)
        with self._ctx._ordered(y, cb, cr):
            self._ctx._chk(self._ctx._lib.fnx_ssim_fast_against_ycbcr(self._ctx._h, self._p, space, yp, ys, cbp, crp, cs,
                                                                      int(ratio), pk, C.byref(out)), "against_ycbcr")
        return float(out.value)

    def close(self):
        if self._p:
            self._ctx._lib.fnx_prepared_free(self._ctx._h, self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def Summarize(failed, has_result, original_size, compressed_size, ssim) -> dict:
    """batch.go:140-158 over parallel arrays (pure host arithmetic, index order)."""
    L = load_library()
    n = len(failed)
    f, pf = _i32(np.asarray(failed)); hr, ph = _i32(np.asarray(has_result))
    o = np.ascontiguousarray(original_size, dtype=np.int64)
    c = np.ascontiguousarray(compressed_size, dtype=np.int64)
    s, ps = _f64(np.asarray(ssim))
    out = np.zeros(4, dtype=np.int64)
    avg = L.fennec_Summarize(n, pf, ph, o.ctypes.data_as(_i64p), c.ctypes.data_as(_i64p), ps,
                             out.ctypes.data_as(_i64p))
    return dict(Total=int(out[0]), Succeeded=int(out[1]), Failed=int(out[2]), TotalSaved=int(out[3]),
                AvgSSIM=float(avg))


# ---------------------------------------------------------------------------------------
# module-level functions with the reference's names, on a per-thread default context
_tls = threading.local()


def default_context(device: int | None = None) -> Context:
    dev = int(os.environ.get("FENNEC_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0"))) if device is None else device
    ctxs = getattr(_tls, "ctxs", None)
    if ctxs is None:
        ctxs = _tls.ctxs = {}
    if dev not in ctxs:
        ctxs[dev] = Context(dev)
    return ctxs[dev]


def _dev_of(x):
    if _is_torch(x) and x.is_cuda:
        return x.device.index or 0
    return None


def SSIM(img1, img2): return default_context(_dev_of(img1)).SSIM(img1, img2)
def SSIMFast(img1, img2): return default_context(_dev_of(img1)).SSIMFast(img1, img2)
def MSSSIM(img1, img2): return default_context(_dev_of(img1)).MSSSIM(img1, img2)
def GaussianBlur(img, sigma): return default_context(_dev_of(img)).GaussianBlur(img, sigma, exact=None)
def Sharpen(img, strength): return default_context(_dev_of(img)).Sharpen(img, strength)
def AdaptiveSharpen(img, strength): return default_context(_dev_of(img)).AdaptiveSharpen(img, strength)
def ApplyOrientation(img, orient): return default_context(_dev_of(img)).ApplyOrientation(img, orient)
def lanczosResize(img, dstW, dstH): return default_context(_dev_of(img)).lanczosResize(img, dstW, dstH)
def smartResize(img, maxW, maxH): return default_context(_dev_of(img)).smartResize(img, maxW, maxH)
def boxDownsample(img, dstW, dstH): return default_context(_dev_of(img)).boxDownsample(img, dstW, dstH)
def Analyze(img): return default_context(_dev_of(img)).Analyze(img)


def gaussianKernel(size=8, sigma=1.5):
    k = np.empty(size * size, dtype=np.float64)
    load_library().fennec_gaussianKernel(size, sigma, k.ctypes.data_as(_f64p))
    return k


def blur_fixed_point(kernel):
    """fnx_blur_fixed_point: (wq, err255) of a blur table as the matrix-pipe kernel quantises it, or None when the table is not
    that kernel's.  Host arithmetic only."""
    k = np.ascontiguousarray(kernel, dtype=np.float64)
    radius = (len(k) - 1) // 2
    wq = (C.c_longlong * len(k))()
    err = C.c_double()
    rc = load_library().fnx_blur_fixed_point(k.ctypes.data_as(_f64p), radius, wq, C.byref(err))
    if rc == FNX_NOOP:
        return None
    if rc != 0:
        raise RuntimeError("fnx_blur_fixed_point failed")
    return np.array(list(wq), dtype=np.int64), float(err.value)


def blurKernel(sigma):
    L = load_library()
    r = L.fennec_blurKernel(float(sigma), None)
    k = np.empty(2 * r + 1, dtype=np.float64)
    L.fennec_blurKernel(float(sigma), k.ctypes.data_as(_f64p))
    return r, k


def precomputeWeights(dst_size, src_size):
    L = load_library()
    off = np.zeros(dst_size + 1, dtype=np.int32)
    n = L.fennec_precomputeWeights(dst_size, src_size, off.ctypes.data_as(_i32p), None, None)
    idx = np.zeros(max(n, 1), dtype=np.int32)
    wt = np.zeros(max(n, 1), dtype=np.float64)
    L.fennec_precomputeWeights(dst_size, src_size, off.ctypes.data_as(_i32p), idx.ctypes.data_as(_i32p),
                               wt.ctypes.data_as(_f64p))
    return off, idx[:max(n, 0)], wt[:max(n, 0)]
