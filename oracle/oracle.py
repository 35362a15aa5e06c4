"""ctypes binding of oracle/libfennec_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this module, and only as the checker / timed CPU baseline.  The product package
(fennec_amd) never imports it.  See fennec_oracle.c for the parity statement
("parity unpinned": no Go toolchain, no golden vectors in the reference).

Images are numpy uint8 arrays of shape (h, w, 4), C-contiguous (tight stride),
or any array whose last two axes are contiguous (stride = arr.strides[0]).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfennec_oracle.so")

_u8p = C.POINTER(C.c_uint8)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int)
_i64p = C.POINTER(C.c_int64)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (strict fp64 flags, oracle/Makefile)."""
    src = os.path.join(_HERE, "fennec_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libfennec_oracle.so"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_clampF.restype = C.c_uint8
        L.orc_clampF.argtypes = [C.c_double]
        L.orc_to_luminance.restype = None
        L.orc_to_luminance.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _f64p]
        L.orc_gaussian_kernel.restype = None
        L.orc_gaussian_kernel.argtypes = [C.c_int, C.c_double, _f64p]
        L.orc_windowed_ssim.restype = C.c_double
        L.orc_windowed_ssim.argtypes = [_f64p, _f64p, C.c_int, C.c_int, _f64p, C.c_int]
        L.orc_pixel_ssim.restype = C.c_double
        L.orc_pixel_ssim.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_size_t]
        L.orc_box_downsample.restype = None
        L.orc_box_downsample.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int]
        L.orc_ssim_fast_dims.restype = C.c_int
        L.orc_ssim_fast_dims.argtypes = [C.c_int, C.c_int, _i32p, _i32p]
        L.orc_ssim_fast.restype = C.c_double
        L.orc_ssim_fast.argtypes = [_u8p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, _f64p, C.c_int]
        L.orc_lanczos_kernel.restype = C.c_double
        L.orc_lanczos_kernel.argtypes = [C.c_double]
        L.orc_precompute_weights.restype = C.c_int
        L.orc_precompute_weights.argtypes = [C.c_int, C.c_int, _i32p, _i32p, _f64p]
        L.orc_resize_h.restype = None
        L.orc_resize_h.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, _i32p, _i32p, _f64p, C.c_int]
        L.orc_resize_v.restype = None
        L.orc_resize_v.argtypes = [_u8p, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p, C.c_int]
        L.orc_lanczos_resize.restype = C.c_int
        L.orc_lanczos_resize.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_smart_resize_dims.restype = C.c_int
        L.orc_smart_resize_dims.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _i32p, _i32p]
        L.orc_ssim.restype = C.c_double
        L.orc_ssim.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, _f64p, C.c_int]
        L.orc_msssim.restype = C.c_double
        L.orc_msssim.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int, _f64p, C.c_int, _f64p]
        L.orc_blur_kernel.restype = C.c_int
        L.orc_blur_kernel.argtypes = [C.c_double, _f64p]
        L.orc_gaussian_blur_k.restype = None
        L.orc_gaussian_blur_k.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _f64p, C.c_int, _u8p, C.c_int, C.c_int]
        L.orc_gaussian_blur.restype = None
        L.orc_gaussian_blur.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_double, _u8p, C.c_int, C.c_int]
        L.orc_blur3x3.restype = None
        L.orc_blur3x3.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int]
        L.orc_sharpen.restype = C.c_int
        L.orc_sharpen.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_double, _u8p, C.c_int, C.c_int]
        L.orc_adaptive_sharpen.restype = C.c_int
        L.orc_adaptive_sharpen.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_double, _u8p, C.c_int, C.c_int]
        L.orc_apply_orientation.restype = C.c_int
        L.orc_apply_orientation.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p]
        L.orc_summarize.restype = C.c_double
        L.orc_summarize.argtypes = [C.c_int, _i32p, _i32p, _i64p, _i64p, _f64p, _i64p]
        L.orc_analyze.restype = C.c_int
        L.orc_analyze.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.POINTER(ImageStats)]
        L.orc_compute_entropy.restype = C.c_double
        L.orc_compute_entropy.argtypes = [_f64p, C.c_int, C.c_double]
        L.orc_is_opaque.restype = C.c_int
        L.orc_is_opaque.argtypes = [_u8p, C.c_size_t]
        L.orc_is_grayscale.restype = C.c_int
        L.orc_is_grayscale.argtypes = [_u8p, C.c_size_t]
        L.orc_analyze_format.restype = C.c_int
        L.orc_analyze_format.argtypes = [_u8p, C.c_int, C.c_int, C.c_int]
        L.orc_apply_palette.restype = None
        L.orc_apply_palette.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, _u8p, C.c_int, _u8p, C.c_int]
        L.orc_ycbcr_to_nrgba.restype = None
        L.orc_ycbcr_to_nrgba.argtypes = [_u8p, C.c_int, _u8p, _u8p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]
        _lib = L
    return _lib


class ImageStats(C.Structure):
    """orc_image_stats: ImageStats (analyze.go:9-22) plus the raw accumulators."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("has_alpha", C.c_int32),
                ("is_grayscale", C.c_int32), ("unique_colors", C.c_int32),
                ("recommended_format", C.c_int32), ("recommended_quality", C.c_int32), ("pad", C.c_int32),
                ("entropy", C.c_double), ("edge_density", C.c_double), ("mean_brightness", C.c_double),
                ("contrast", C.c_double), ("estimated_compression", C.c_double),
                ("histogram", C.c_double * 256), ("bright_sum", C.c_double), ("variance_sum", C.c_double),
                ("sample_count", C.c_int64), ("edge_count", C.c_int64), ("edge_total", C.c_int64)]


# ---------------------------------------------------------------- helpers
def _img(a: np.ndarray):
    """(ptr, stride, w, h) of an (h, w, 4) uint8 image with contiguous rows."""
    assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 4, a.shape
    h, w = a.shape[:2]
    if h > 0 and w > 0:
        assert a.strides[2] == 1 and a.strides[1] == 4, "rows must be contiguous"
    stride = a.strides[0] if h > 1 else w * 4
    return a.ctypes.data_as(_u8p), int(stride), int(w), int(h)


def _f64(a: np.ndarray):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_f64p)


def _i32(a: np.ndarray):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def new_image(w: int, h: int) -> np.ndarray:
    """image.NewNRGBA(image.Rect(0,0,w,h)): zeroed, tight."""
    return np.zeros((max(h, 0), max(w, 0), 4), dtype=np.uint8)


# ---------------------------------------------------------------- API
def clampF(x: float) -> int:
    return int(lib().orc_clampF(float(x)))


def gaussian_kernel(size: int = 8, sigma: float = 1.5) -> np.ndarray:
    k = np.empty(size * size, dtype=np.float64)
    lib().orc_gaussian_kernel(size, sigma, k.ctypes.data_as(_f64p))
    return k


def to_luminance(img: np.ndarray) -> np.ndarray:
    p, s, w, h = _img(img)
    lum = np.empty((h, w), dtype=np.float64)
    lib().orc_to_luminance(p, s, w, h, lum.ctypes.data_as(_f64p))
    return lum


def windowed_ssim(lumA, lumB, kernel=None, procs: int = 1) -> float:
    lumA, pa = _f64(lumA)
    lumB, pb = _f64(lumB)
    h, w = lumA.shape
    k, pk = _f64(gaussian_kernel() if kernel is None else kernel)
    return float(lib().orc_windowed_ssim(pa, pb, w, h, pk, procs))


def pixel_ssim(a: np.ndarray, b: np.ndarray) -> float:
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    pa, _, w, h = _img(a)
    pb, _, _, _ = _img(b)
    return float(lib().orc_pixel_ssim(pa, pb, w, h, a.size))


def pixel_ssim_flat(pix_a: np.ndarray, pix_b: np.ndarray, w: int, h: int) -> float:
    """pixelSSIM over two flat Pix slices as Go holds them (ssim.go:178: `i < len(a.Pix)`); w, h only give n."""
    a = np.ascontiguousarray(pix_a, dtype=np.uint8).reshape(-1)
    b = np.ascontiguousarray(pix_b, dtype=np.uint8).reshape(-1)
    assert b.size >= a.size, "the reference panics (index out of range)"
    return float(lib().orc_pixel_ssim(a.ctypes.data_as(_u8p), b.ctypes.data_as(_u8p), int(w), int(h), a.size))


def box_downsample(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    p, s, w, h = _img(img)
    if w <= 0 or h <= 0 or dw <= 0 or dh <= 0:
        return new_image(0, 0)
    dst = new_image(dw, dh)
    lib().orc_box_downsample(p, s, w, h, dst.ctypes.data_as(_u8p), dw * 4, dw, dh)
    return dst


def ssim_fast_dims(w: int, h: int):
    nw, nh = C.c_int(), C.c_int()
    ds = lib().orc_ssim_fast_dims(w, h, C.byref(nw), C.byref(nh))
    return bool(ds), nw.value, nh.value


def ssim_fast(a: np.ndarray, b: np.ndarray, kernel=None, procs: int = 1) -> float:
    pa, sa, w, h = _img(a)
    pb, sb, _, _ = _img(b)
    k, pk = _f64(gaussian_kernel() if kernel is None else kernel)
    return float(lib().orc_ssim_fast(pa, sa, pb, sb, w, h, pk, procs))


def ssim(a: np.ndarray, b: np.ndarray, kernel=None, procs: int = 1) -> float:
    pa, sa, aw, ah = _img(a)
    pb, sb, bw, bh = _img(b)
    k, pk = _f64(gaussian_kernel() if kernel is None else kernel)
    return float(lib().orc_ssim(pa, sa, aw, ah, pb, sb, bw, bh, pk, procs))


def msssim(a: np.ndarray, b: np.ndarray, kernel=None, procs: int = 1, per_level: bool = False):
    pa, sa, aw, ah = _img(a)
    pb, sb, bw, bh = _img(b)
    k, pk = _f64(gaussian_kernel() if kernel is None else kernel)
    lv = np.empty(5, dtype=np.float64)
    r = float(lib().orc_msssim(pa, sa, aw, ah, pb, sb, bw, bh, pk, procs, lv.ctypes.data_as(_f64p)))
    return (r, lv) if per_level else r


def lanczos_kernel(x: float) -> float:
    return float(lib().orc_lanczos_kernel(float(x)))


def precompute_weights(dst_size: int, src_size: int):
    """CSR tap table (offset[dst+1], index[n], weight[n]) of resize.go:164-197."""
    off = np.zeros(dst_size + 1, dtype=np.int32)
    n = lib().orc_precompute_weights(dst_size, src_size, off.ctypes.data_as(_i32p), None, None)
    idx = np.zeros(max(n, 1), dtype=np.int32)
    wt = np.zeros(max(n, 1), dtype=np.float64)
    lib().orc_precompute_weights(dst_size, src_size, off.ctypes.data_as(_i32p),
                                 idx.ctypes.data_as(_i32p), wt.ctypes.data_as(_f64p))
    return off, idx[:n], wt[:n]


def resize_h(img: np.ndarray, dw: int, table=None, procs: int = 1) -> np.ndarray:
    p, s, w, h = _img(img)
    off, idx, wt = precompute_weights(dw, w) if table is None else table
    off, po = _i32(off); idx, pi = _i32(idx); wt, pw = _f64(wt)
    dst = new_image(dw, h)
    lib().orc_resize_h(p, s, h, dst.ctypes.data_as(_u8p), dw * 4, dw, po, pi, pw, procs)
    return dst


def resize_v(img: np.ndarray, dh: int, table=None, procs: int = 1) -> np.ndarray:
    p, s, w, h = _img(img)
    off, idx, wt = precompute_weights(dh, h) if table is None else table
    off, po = _i32(off); idx, pi = _i32(idx); wt, pw = _f64(wt)
    dst = new_image(w, dh)
    lib().orc_resize_v(p, s, dst.ctypes.data_as(_u8p), w * 4, w, dh, po, pi, pw, procs)
    return dst


def lanczos_resize(img: np.ndarray, dw: int, dh: int, procs: int = 1) -> np.ndarray:
    p, s, w, h = _img(img)
    if w <= 0 or h <= 0 or dw <= 0 or dh <= 0:
        return new_image(0, 0)
    dst = new_image(dw, dh)
    lib().orc_lanczos_resize(p, s, w, h, dst.ctypes.data_as(_u8p), dw * 4, dw, dh, procs)
    return dst


def smart_resize_dims(w: int, h: int, max_w: int, max_h: int):
    dw, dh = C.c_int(), C.c_int()
    r = lib().orc_smart_resize_dims(w, h, max_w, max_h, C.byref(dw), C.byref(dh))
    return bool(r), dw.value, dh.value


def smart_resize(img: np.ndarray, max_w: int, max_h: int, procs: int = 1) -> np.ndarray:
    h, w = img.shape[:2]
    r, dw, dh = smart_resize_dims(w, h, max_w, max_h)
    return lanczos_resize(img, dw, dh, procs) if r else img


def blur_kernel(sigma: float):
    r = lib().orc_blur_kernel(float(sigma), None)
    k = np.empty(2 * r + 1, dtype=np.float64)
    lib().orc_blur_kernel(float(sigma), k.ctypes.data_as(_f64p))
    return r, k


def gaussian_blur(img: np.ndarray, sigma: float, kernel=None, procs: int = 1) -> np.ndarray:
    if sigma <= 0:
        return img  # same pointer (effects.go:147-149)
    p, s, w, h = _img(img)
    dst = new_image(w, h)
    if kernel is None:
        radius, kernel = blur_kernel(sigma)
    else:
        radius = (len(kernel) - 1) // 2
    k, pk = _f64(kernel)
    lib().orc_gaussian_blur_k(p, s, w, h, pk, radius, dst.ctypes.data_as(_u8p), w * 4, procs)
    return dst


def blur3x3(img: np.ndarray, procs: int = 1) -> np.ndarray:
    p, s, w, h = _img(img)
    dst = new_image(w, h)
    lib().orc_blur3x3(p, s, w, h, dst.ctypes.data_as(_u8p), w * 4, procs)
    return dst


def sharpen(img: np.ndarray, strength: float, procs: int = 1) -> np.ndarray:
    p, s, w, h = _img(img)
    dst = new_image(w, h)
    r = lib().orc_sharpen(p, s, w, h, float(strength), dst.ctypes.data_as(_u8p), w * 4, procs)
    return dst if r else img


def adaptive_sharpen(img: np.ndarray, strength: float, procs: int = 1) -> np.ndarray:
    p, s, w, h = _img(img)
    dst = new_image(w, h)
    r = lib().orc_adaptive_sharpen(p, s, w, h, float(strength), dst.ctypes.data_as(_u8p), w * 4, procs)
    return dst if r else img


def apply_orientation(img: np.ndarray, orient: int) -> np.ndarray:
    p, s, w, h = _img(img)
    ow, oh = (h, w) if orient in (5, 6, 7, 8) else (w, h)
    dst = new_image(ow, oh)
    r = lib().orc_apply_orientation(p, s, w, h, int(orient), dst.ctypes.data_as(_u8p))
    return dst if r else img


def summarize(failed, has_result, original_size, compressed_size, ssim_vals):
    n = len(failed)
    f, pf = _i32(np.asarray(failed)); hr, ph = _i32(np.asarray(has_result))
    o = np.ascontiguousarray(original_size, dtype=np.int64)
    c = np.ascontiguousarray(compressed_size, dtype=np.int64)
    s, ps = _f64(np.asarray(ssim_vals))
    out = np.zeros(4, dtype=np.int64)
    avg = lib().orc_summarize(n, pf, ph, o.ctypes.data_as(_i64p), c.ctypes.data_as(_i64p), ps,
                              out.ctypes.data_as(_i64p))
    return dict(Total=int(out[0]), Succeeded=int(out[1]), Failed=int(out[2]),
                TotalSaved=int(out[3]), AvgSSIM=float(avg))


def analyze(img: np.ndarray) -> dict:
    """Analyze (analyze.go:26-124) -> dict of ImageStats fields + raw accumulators."""
    p, s, w, h = _img(img)
    st = ImageStats()
    if lib().orc_analyze(p, s, w, h, C.byref(st)) != 0:
        raise MemoryError("orc_analyze")
    d = {name: getattr(st, name) for name, _ in ImageStats._fields_ if name not in ("pad", "histogram")}
    d["histogram"] = np.array(st.histogram[:], dtype=np.float64)
    return d


def _flat(img: np.ndarray):
    """the flat Pix slice: (h-1)*stride + 4*w bytes for a strided view, all of it when tight"""
    p, s, w, h = _img(img)
    n = (h - 1) * s + 4 * w if h > 0 and w > 0 else 0
    return p, n


def is_opaque(img: np.ndarray) -> bool:
    p, n = _flat(img)
    return bool(lib().orc_is_opaque(p, n))


def is_grayscale(img: np.ndarray) -> bool:
    p, n = _flat(img)
    return bool(lib().orc_is_grayscale(p, n))


def analyze_format(img: np.ndarray) -> int:
    """analyzeFormat (convert.go:105-146): 1 = JPEG, 2 = PNG."""
    p, s, w, h = _img(img)
    return int(lib().orc_analyze_format(p, s, w, h))


def apply_palette(img: np.ndarray, palette: np.ndarray):
    """applyPalette + palettedToNRGBA (targetsize.go:488-546) -> (indices (h, w), quantized (h, w, 4))."""
    p, s, w, h = _img(img)
    pal = np.ascontiguousarray(palette, dtype=np.uint8).reshape(-1, 4)
    assert np.all(pal[:, 3] == 255)
    idx = np.zeros((h, w), dtype=np.uint8)
    q = new_image(w, h)
    lib().orc_apply_palette(p, s, w, h, pal.ctypes.data_as(_u8p), len(pal), idx.ctypes.data_as(_u8p), w,
                            q.ctypes.data_as(_u8p), 4 * w)
    return idx, q


def ycbcr_to_nrgba(y: np.ndarray, cb, cr, ratio: int) -> np.ndarray:
    """toNRGBARef of an image.YCbCr / image.Gray (convert.go:22-64 + Go's color.YCbCr.RGBA)."""
    h, w = y.shape
    y = np.ascontiguousarray(y)
    dst = new_image(w, h)
    if cb is None:
        lib().orc_ycbcr_to_nrgba(y.ctypes.data_as(_u8p), w, None, None, 0, 0, w, h, dst.ctypes.data_as(_u8p), 4 * w)
    else:
        cb, cr = np.ascontiguousarray(cb), np.ascontiguousarray(cr)
        lib().orc_ycbcr_to_nrgba(y.ctypes.data_as(_u8p), w, cb.ctypes.data_as(_u8p), cr.ctypes.data_as(_u8p),
                                 cb.shape[1], int(ratio), w, h, dst.ctypes.data_as(_u8p), 4 * w)
    return dst


# ---------------------------------------------------------------- JPEG quantisation round trip (Go's image/jpeg arithmetic)
def jpeg_quant_tables(quality: int):
    """(luminance, chrominance) 8x8 tables in natural order as writer.go scales them for `quality`."""
    lum = np.empty(64, dtype=np.uint8)
    chr_ = np.empty(64, dtype=np.uint8)
    L = lib()
    L.orc_jpeg_quant_tables.restype = None
    L.orc_jpeg_quant_tables.argtypes = [C.c_int, _u8p, _u8p]
    L.orc_jpeg_quant_tables(int(quality), lum.ctypes.data_as(_u8p), chr_.ctypes.data_as(_u8p))
    return lum.reshape(8, 8), chr_.reshape(8, 8)


def jpeg_fdct(block: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(block, dtype=np.int32).reshape(64).copy()
    L = lib()
    L.orc_jpeg_fdct.restype = None
    L.orc_jpeg_fdct.argtypes = [C.POINTER(C.c_int32)]
    L.orc_jpeg_fdct(b.ctypes.data_as(C.POINTER(C.c_int32)))
    return b.reshape(8, 8)


def jpeg_idct(block: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(block, dtype=np.int32).reshape(64).copy()
    L = lib()
    L.orc_jpeg_idct.restype = None
    L.orc_jpeg_idct.argtypes = [C.POINTER(C.c_int32)]
    L.orc_jpeg_idct(b.ctypes.data_as(C.POINTER(C.c_int32)))
    return b.reshape(8, 8)


def rgb_to_ycbcr(r: int, g: int, b: int):
    L = lib()
    L.orc_rgb_to_ycbcr.restype = None
    L.orc_rgb_to_ycbcr.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, _u8p, _u8p, _u8p]
    o = np.zeros(3, dtype=np.uint8)
    L.orc_rgb_to_ycbcr(int(r), int(g), int(b), o[0:].ctypes.data_as(_u8p), o[1:].ctypes.data_as(_u8p), o[2:].ctypes.data_as(_u8p))
    return int(o[0]), int(o[1]), int(o[2])


def jpeg_roundtrip_planes(img: np.ndarray, quality: int):
    """(Y, Cb, Cr) planes of jpeg.Decode(jpeg.Encode(img, quality)): MCU-padded, 4:2:0."""
    p, s, w, h = _img(img)
    mx, my = (w + 15) // 16, (h + 15) // 16
    y = np.empty((16 * my, 16 * mx), dtype=np.uint8)
    cb = np.empty((8 * my, 8 * mx), dtype=np.uint8)
    cr = np.empty((8 * my, 8 * mx), dtype=np.uint8)
    L = lib()
    L.orc_jpeg_roundtrip_planes.restype = None
    L.orc_jpeg_roundtrip_planes.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _u8p, _u8p]
    L.orc_jpeg_roundtrip_planes(p, s, w, h, int(quality), y.ctypes.data_as(_u8p), cb.ctypes.data_as(_u8p), cr.ctypes.data_as(_u8p))
    return y, cb, cr


def jpeg_roundtrip(img: np.ndarray, quality: int) -> np.ndarray:
    """toNRGBARef(jpeg.Decode(jpeg.Encode(img, quality))) for an opaque image (compress.go:50-58)."""
    p, s, w, h = _img(img)
    dst = new_image(w, h)
    L = lib()
    L.orc_jpeg_roundtrip.restype = C.c_int
    L.orc_jpeg_roundtrip.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]
    L.orc_jpeg_roundtrip(p, s, w, h, int(quality), dst.ctypes.data_as(_u8p), w * 4)
    return dst


# ---------------------------------------------------------------- baseline JPEG files (jpeg.Encode's layout) and a decoder
def jpeg_huffman_luts() -> np.ndarray:
    """The four standard tables (luminance DC, luminance AC, chrominance DC, chrominance AC) as length << 24 | code."""
    luts = np.zeros((4, 256), dtype=np.uint32)
    L = lib()
    L.orc_jpeg_huffman_luts.restype = None
    L.orc_jpeg_huffman_luts.argtypes = [C.POINTER(C.c_uint32)]
    L.orc_jpeg_huffman_luts(luts.ctypes.data_as(C.POINTER(C.c_uint32)))
    return luts


def jpeg_encode(img: np.ndarray, quality: int, with_coefficients: bool = False):
    """jpeg.Encode(img, &jpeg.Options{Quality: quality}) as restated (io.go:157-169) -> the file's bytes
    [, the quantised coefficients, (blocks, 64) int16 in zig-zag order, blocks in scan order]."""
    p, s, w, h = _img(img)
    mx, my = (w + 15) // 16, (h + 15) // 16
    cap = 1024 + mx * my * 6 * 64 * 4
    out = np.empty(cap, dtype=np.uint8)
    coef = np.zeros((mx * my * 6, 64), dtype=np.int16) if with_coefficients else None
    L = lib()
    L.orc_jpeg_encode.restype = C.c_long
    L.orc_jpeg_encode.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.c_long, C.POINTER(C.c_int16)]
    n = L.orc_jpeg_encode(p, s, w, h, int(quality), out.ctypes.data_as(_u8p), cap,
                          coef.ctypes.data_as(C.POINTER(C.c_int16)) if coef is not None else None)
    if n <= 0:
        raise RuntimeError(f"orc_jpeg_encode: {n}")
    data = out[:n].tobytes()
    return (data, coef) if with_coefficients else data


def jpeg_decode_planes(data: bytes, with_coefficients: bool = False):
    """A baseline one- or three-component file (4:4:4, 4:2:2, 4:2:0, 4:4:0, 4:1:1, 4:1:0) -> (w, h, ratio, Y, Cb, Cr) MCU-padded planes as
    reader.go would hold them [, coefficients]; ratio -1 and Cb = Cr = None: one component (image.Gray)."""
    buf = np.frombuffer(data, dtype=np.uint8)
    L = lib()
    L.orc_jpeg_decode_planes.restype = C.c_int
    L.orc_jpeg_decode_planes.argtypes = [_u8p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _u8p, _u8p, _u8p,
                                         C.POINTER(C.c_int16)]
    w, h, ratio = C.c_int(), C.c_int(), C.c_int()
    bp = buf.ctypes.data_as(_u8p)
    rc = L.orc_jpeg_decode_planes(bp, len(data), C.byref(w), C.byref(h), C.byref(ratio), None, None, None, None)
    if rc != 1:
        raise RuntimeError(f"orc_jpeg_decode_planes: {rc}")
    hy, vy = {-1: (1, 1), 0: (1, 1), 1: (2, 1), 2: (2, 2), 3: (1, 2), 4: (4, 1), 5: (4, 2)}[ratio.value]
    mx, my = (w.value + 8 * hy - 1) // (8 * hy), (h.value + 8 * vy - 1) // (8 * vy)
    y = np.empty((8 * vy * my, 8 * hy * mx), dtype=np.uint8)
    cb = np.empty((8 * my, 8 * mx), dtype=np.uint8)
    cr = np.empty((8 * my, 8 * mx), dtype=np.uint8)
    nblk = mx * my * (1 if ratio.value < 0 else hy * vy + 2)
    coef = np.zeros((nblk, 64), dtype=np.int16) if with_coefficients else None
    rc = L.orc_jpeg_decode_planes(bp, len(data), C.byref(w), C.byref(h), C.byref(ratio), y.ctypes.data_as(_u8p), cb.ctypes.data_as(_u8p),
                                  cr.ctypes.data_as(_u8p), coef.ctypes.data_as(C.POINTER(C.c_int16)) if coef is not None else None)
    if rc != 1:
        raise RuntimeError(f"orc_jpeg_decode_planes: {rc}")
    out = (w.value, h.value, ratio.value, y, cb, cr) if ratio.value >= 0 else (w.value, h.value, -1, y, None, None)
    return out + (coef,) if with_coefficients else out


def jpeg_decode_cmyk(data: bytes) -> np.ndarray:
    """toNRGBARef(jpeg.Decode(data)) of a four-component file (Adobe CMYK or YCbCrK, every component 1 x 1): orc_jpeg_decode_cmyk."""
    buf = np.frombuffer(data, dtype=np.uint8)
    L = lib()
    L.orc_jpeg_decode_cmyk.restype = C.c_int
    L.orc_jpeg_decode_cmyk.argtypes = [_u8p, C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int), _u8p, C.c_int]
    w, h = C.c_int(), C.c_int()
    rc = L.orc_jpeg_decode_cmyk(buf.ctypes.data_as(_u8p), len(data), C.byref(w), C.byref(h), None, 0)
    if rc != 1:
        raise RuntimeError(f"orc_jpeg_decode_cmyk: {rc}")
    out = np.empty((h.value, w.value, 4), dtype=np.uint8)
    rc = L.orc_jpeg_decode_cmyk(buf.ctypes.data_as(_u8p), len(data), C.byref(w), C.byref(h), out.ctypes.data_as(_u8p), w.value * 4)
    if rc != 1:
        raise RuntimeError(f"orc_jpeg_decode_cmyk: {rc}")
    return out


def _jpeg_components(data: bytes) -> int:
    pos = 2
    while pos + 4 <= len(data) and data[pos] == 0xFF:
        m, n = data[pos + 1], (data[pos + 2] << 8) | data[pos + 3]
        if m in (0xC0, 0xC1, 0xC2):
            return data[pos + 9] if pos + 9 < len(data) else 0
        if m == 0xDA:
            break
        pos += 2 + n
    return 0


def jpeg_decode(data: bytes) -> np.ndarray:
    """toNRGBARef(jpeg.Decode(data)) for such a file."""
    if _jpeg_components(data) == 4:
        return jpeg_decode_cmyk(data)
    w, h, ratio, y, cb, cr = jpeg_decode_planes(data)
    full = ycbcr_to_nrgba(y, cb, cr, ratio)
    return np.ascontiguousarray(full[:h, :w])
