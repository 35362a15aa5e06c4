"""Independent numpy restatement of fennec's per-pixel hot path (TEST ONLY).

Written from the Go source separately from oracle/fennec_oracle.c, vectorised
over pixels but keeping the reference's per-pixel operation ORDER (tap loops stay
explicit, running sums use add.accumulate, numpy never fuses mul+add), so it
must agree with the C oracle bit for bit.  With no Go toolchain and no golden
vectors in the reference, two independently written restatements agreeing is the
strongest pin available for the oracle ("parity unpinned" -- see DESIGN.md).

Images: uint8 (h, w, 4).
"""
from __future__ import annotations

import math

import numpy as np

C1 = 6.5025   # ssim.go:11-17 (exact Go constants)
C2 = 58.5225


def round_half_away(x):
    """math.Round: nearest, ties away from zero (exact: x - trunc(x) is exact)."""
    x = np.asarray(x, dtype=np.float64)
    t = np.trunc(x)
    return t + np.sign(x) * (np.abs(x - t) >= 0.5)


def clampF(x):
    """convert.go:149-158"""
    return np.clip(round_half_away(x), 0, 255).astype(np.uint8)


def seq_sum(v) -> float:
    """Left-to-right running sum (Go's `s += v[i]` loop)."""
    v = np.asarray(v, dtype=np.float64).ravel()
    if v.size == 0:
        return 0.0
    return float(np.add.accumulate(v)[-1])


def flat_pix_image(img):
    """What `copy(dst.Pix, img.Pix)` leaves in a fresh tight w x h image (effects.go:68,120; convert.go:16):
    Go's copy() moves min(len(dst), len(src)) FLAT bytes, and len(img.Pix) >= (h-1)*Stride + 4w >= 4wh for
    every valid NRGBA, so it is the first 4wh bytes of the Pix slice reshaped -- the image's own rows only
    when Stride == 4w.  `img` is an (h, w, 4) view whose first byte is Pix[0] and whose row stride is Stride."""
    h, w = img.shape[:2]
    if h == 0 or w == 0:
        return img.copy()
    assert img.strides[1:] == (4, 1) and (h == 1 or img.strides[0] >= 4 * w), "not an image.NRGBA layout"
    return np.lib.stride_tricks.as_strided(img, shape=(h, w, 4), strides=(4 * w, 4, 1)).copy()


# ------------------------------------------------------------------ ssim.go
def to_luminance(img):
    """ssim.go:207-220"""
    r = img[..., 0].astype(np.float64)
    g = img[..., 1].astype(np.float64)
    b = img[..., 2].astype(np.float64)
    return (0.299 * r + 0.587 * g) + 0.114 * b


def gaussian_kernel(size=8, sigma=1.5):
    """ssim.go:223-241"""
    half = size // 2
    vals = []
    s = 0.0
    for y in range(-half, half):
        for x in range(-half, half):
            v = math.exp(-float(x * x + y * y) / (2 * sigma * sigma))
            vals.append(v)
            s += v
    return np.array([v / s for v in vals], dtype=np.float64)


def ssim_map(lumA, lumB, kernel):
    """Per-window SSIM values of windowedSSIM (ssim.go:110-146), shape (h-8, w-8)."""
    h, w = lumA.shape
    oh, ow = h - 8, w - 8
    if oh <= 0 or ow <= 0:
        return np.zeros((max(oh, 0), max(ow, 0)))
    muA = np.zeros((oh, ow)); muB = np.zeros((oh, ow))
    ki = 0
    for wy in range(8):          # window rows y-4 .. y+3  ->  offset wy from y-4
        for wx in range(8):
            va = lumA[wy:wy + oh, wx:wx + ow]
            vb = lumB[wy:wy + oh, wx:wx + ow]
            muA = muA + va * kernel[ki]
            muB = muB + vb * kernel[ki]
            ki += 1
    sAA = np.zeros((oh, ow)); sBB = np.zeros((oh, ow)); sAB = np.zeros((oh, ow))
    ki = 0
    for wy in range(8):
        for wx in range(8):
            da = lumA[wy:wy + oh, wx:wx + ow] - muA
            db = lumB[wy:wy + oh, wx:wx + ow] - muB
            sAA = sAA + (da * da) * kernel[ki]
            sBB = sBB + (db * db) * kernel[ki]
            sAB = sAB + (da * db) * kernel[ki]
            ki += 1
    num = ((2 * muA) * muB + C1) * (2 * sAB + C2)
    den = ((muA * muA + muB * muB) + C1) * ((sAA + sBB) + C2)
    return num / den


def windowed_ssim(lumA, lumB, kernel=None):
    """ssim.go:73-166 with GOMAXPROCS=1 (single running sum, x then y)."""
    if kernel is None:
        kernel = gaussian_kernel()
    m = ssim_map(lumA, lumB, kernel)
    if m.size == 0:
        return 1.0
    return seq_sum(m) / float(m.size)


def pixel_ssim(a, b):
    """ssim.go:169-204 (tight images)."""
    h, w = a.shape[:2]
    n = float(w * h)
    if n == 0:
        return 1.0
    la = to_luminance(a).ravel(); lb = to_luminance(b).ravel()
    muA = seq_sum(la) / n; muB = seq_sum(lb) / n
    da = la - muA; db = lb - muB
    sAA = seq_sum(da * da) / n; sBB = seq_sum(db * db) / n; sAB = seq_sum(da * db) / n
    num = (2 * muA * muB + C1) * (2 * sAB + C2)
    den = (muA * muA + muB * muB + C1) * (sAA + sBB + C2)
    return num / den


def _box_edges(src, dst):
    """ssim.go:254-278 index rules for one axis."""
    ratio = float(src) / float(dst)
    e0 = np.empty(dst, dtype=np.int64); e1 = np.empty(dst, dtype=np.int64)
    for d in range(dst):
        s0 = int(float(d) * ratio)
        s1 = int(float(d + 1) * ratio)
        if s1 > src:
            s1 = src
        if s0 >= s1:
            s0 = s1 - 1
        if s0 < 0:
            s0 = 0
        e0[d], e1[d] = s0, s1
    return e0, e1


def box_downsample(img, dw, dh):
    """ssim.go:244-309"""
    sh, sw = img.shape[:2]
    if sw <= 0 or sh <= 0 or dw <= 0 or dh <= 0:
        return np.zeros((0, 0, 4), dtype=np.uint8)
    x0, x1 = _box_edges(sw, dw)
    y0, y1 = _box_edges(sh, dh)
    # integer summed-area table: sums are exact in fp64, order is irrelevant
    sat = np.zeros((sh + 1, sw + 1, 4), dtype=np.int64)
    sat[1:, 1:] = np.cumsum(np.cumsum(img.astype(np.int64), axis=0), axis=1)
    S = (sat[y1][:, x1] - sat[y0][:, x1] - sat[y1][:, x0] + sat[y0][:, x0]).astype(np.float64)
    cnt = ((y1 - y0)[:, None] * (x1 - x0)[None, :]).astype(np.float64)
    out = np.zeros((dh, dw, 4), dtype=np.uint8)
    ok = cnt > 0
    inv = np.where(ok, 1.0 / np.where(ok, cnt, 1.0), 0.0)
    val = clampF(S * inv[..., None])
    out[ok] = val[ok]
    return out


def ssim_fast_dims(w, h):
    """ssim.go:52-56"""
    if w > 512 or h > 512:
        scale = 512.0 / max(float(w), float(h))
        nw = int(max(8.0, float(round_half_away(float(w) * scale))))
        nh = int(max(8.0, float(round_half_away(float(h) * scale))))
        return True, nw, nh
    return False, w, h


def ssim_fast(a, b, kernel=None):
    """ssim.go:48-70"""
    h, w = a.shape[:2]
    ds, nw, nh = ssim_fast_dims(w, h)
    if ds:
        a = box_downsample(a, nw, nh); b = box_downsample(b, nw, nh)
        w, h = nw, nh
    if w < 8 or h < 8:
        return pixel_ssim(a, b)
    return windowed_ssim(to_luminance(a), to_luminance(b), kernel)


# ------------------------------------------------------------------ resize.go
def lanczos_kernel(x):
    """resize.go:57-69"""
    if x == 0:
        return 1.0
    if x < 0:
        x = -x
    if x >= 3.0:
        return 0.0
    xpi = x * math.pi
    return (3.0 * math.sin(xpi) * math.sin(xpi / 3.0)) / (xpi * xpi)


def precompute_weights(dst_size, src_size):
    """resize.go:164-197 (ratio/support as resizeH/V derive them); list of (idx[], w[])."""
    ratio = float(src_size) / float(dst_size)
    support = 3.0 * ratio if ratio > 1 else 3.0
    fscale = max(ratio, 1.0)
    out = []
    for d in range(dst_size):
        center = (float(d) + 0.5) * ratio - 0.5
        left = int(math.ceil(center - support))
        right = int(math.floor(center + support))
        left = max(left, 0)
        if right >= src_size:
            right = src_size - 1
        idx, wt, wsum = [], [], 0.0
        for s in range(left, right + 1):
            w = lanczos_kernel((float(s) - center) / fscale)
            if w != 0:
                wsum += w
                idx.append(s); wt.append(w)
        if wsum != 0:
            wt = [v / wsum for v in wt]
        out.append((idx, wt))
    return out


def _resize_axis(src, dst_size, axis):
    """resizeH (axis=1, resize.go:77-118) / resizeV (axis=0, resize.go:121-161)."""
    src_size = src.shape[axis]
    table = precompute_weights(dst_size, src_size)
    f = src.astype(np.float64)
    if axis == 1:
        out = np.zeros((src.shape[0], dst_size, 4), dtype=np.uint8)
    else:
        out = np.zeros((dst_size, src.shape[1], 4), dtype=np.uint8)
    for d, (idx, wt) in enumerate(table):
        n = src.shape[0] if axis == 1 else src.shape[1]
        r = np.zeros(n); g = np.zeros(n); b = np.zeros(n); a = np.zeros(n)
        for s, w in zip(idx, wt):
            px = f[:, s, :] if axis == 1 else f[s, :, :]
            aw = px[:, 3] * w
            r = r + px[:, 0] * aw
            g = g + px[:, 1] * aw
            b = b + px[:, 2] * aw
            a = a + aw
        ok = a > 0.5
        inv = 1.0 / np.where(ok, a, 1.0)
        px = np.stack([clampF(r * inv), clampF(g * inv), clampF(b * inv), clampF(a)], axis=-1)
        px[~ok] = 0
        if axis == 1:
            out[:, d, :] = px
        else:
            out[d, :, :] = px
    return out


def lanczos_resize(img, dw, dh):
    """resize.go:37-53"""
    sh, sw = img.shape[:2]
    if sw <= 0 or sh <= 0 or dw <= 0 or dh <= 0:
        return np.zeros((0, 0, 4), dtype=np.uint8)
    if sw == dw and sh == dh:
        return flat_pix_image(img)                     # resize.go:45-49: copy(dst.Pix, img.Pix)
    return _resize_axis(_resize_axis(img, dw, 1), dh, 0)


def smart_resize_dims(sw, sh, max_w, max_h):
    """resize.go:12-32"""
    if max_w <= 0:
        max_w = sw
    if max_h <= 0:
        max_h = sh
    if sw <= max_w and sh <= max_h:
        return False, sw, sh
    ratio = min(float(max_w) / float(sw), float(max_h) / float(sh))
    return True, int(max(1.0, float(round_half_away(sw * ratio)))), int(max(1.0, float(round_half_away(sh * ratio))))


def ssim(a, b, kernel=None):
    """ssim.go:24-43"""
    h, w = a.shape[:2]
    if b.shape[:2] != (h, w):
        b = lanczos_resize(b, w, h)
    if w < 8 or h < 8:
        return pixel_ssim(a, b)
    return windowed_ssim(to_luminance(a), to_luminance(b), kernel)


def msssim(a, b, kernel=None):
    """ssim.go:313-365"""
    h, w = a.shape[:2]
    if b.shape[:2] != (h, w):
        b = lanczos_resize(b, w, h)
    weights = [0.0448, 0.2856, 0.3001, 0.2363, 0.1333]
    levels = len(weights)
    tw, th = w, h
    for i in range(levels - 1):
        if min(tw, th) < 8:
            weights = weights[:i + 1]
            s = 0.0
            for wt in weights:
                s += wt
            weights = [wt / s for wt in weights]
            break
        tw //= 2
        th //= 2
    ac, bc = flat_pix_image(a), flat_pix_image(b)      # toNRGBA: copy(dst.Pix, nrgba.Pix), ssim.go:345-346
    result = 0.0
    for i, wt in enumerate(weights):
        s = ssim_fast(ac, bc, kernel)
        result += wt * math.log(max(s, 1e-10))
        if i < len(weights) - 1:
            nw, nh = ac.shape[1] // 2, ac.shape[0] // 2
            if nw < 8 or nh < 8:
                break
            ac = box_downsample(ac, nw, nh); bc = box_downsample(bc, nw, nh)
    return math.exp(result)


# ------------------------------------------------------------------ effects.go
def blur_kernel(sigma):
    """effects.go:153-165"""
    radius = int(math.ceil(sigma * 3))
    k = []
    s = 0.0
    for i in range(2 * radius + 1):
        x = float(i - radius)
        v = math.exp(-(x * x) / (2 * sigma * sigma))
        k.append(v); s += v
    return radius, np.array([v / s for v in k], dtype=np.float64)


def gaussian_blur(img, sigma):
    """effects.go:146-220"""
    if sigma <= 0:
        return img
    h, w = img.shape[:2]
    radius, k = blur_kernel(sigma)
    f = img[..., :3].astype(np.float64)
    acc = np.zeros((h, w, 3))
    xs = np.arange(w)
    for i in range(2 * radius + 1):
        sx = np.clip(xs + i - radius, 0, w - 1)
        acc = acc + f[:, sx, :] * k[i]
    tmp = clampF(acc).astype(np.float64)      # uint8 intermediate (effects.go:186-188)
    acc = np.zeros((h, w, 3))
    ys = np.arange(h)
    for i in range(2 * radius + 1):
        sy = np.clip(ys + i - radius, 0, h - 1)
        acc = acc + tmp[sy, :, :] * k[i]
    out = np.empty_like(img)
    out[..., :3] = clampF(acc)
    out[..., 3] = img[..., 3]
    return out


def blur3x3(img):
    """effects.go:116-141: the border and every alpha byte are whatever copy(dst.Pix, img.Pix) put there"""
    out = flat_pix_image(img)
    h, w = img.shape[:2]
    if h < 3 or w < 3:
        return out
    f = img[..., :3].astype(np.float64)
    s = np.zeros((h - 2, w - 2, 3))
    for dy, dx, wt in [(-1, -1, 1), (-1, 0, 2), (-1, 1, 1), (0, -1, 2), (0, 0, 4), (0, 1, 2),
                       (1, -1, 1), (1, 0, 2), (1, 1, 1)]:
        s = s + f[1 + dy:h - 1 + dy, 1 + dx:w - 1 + dx, :] * float(wt)
    out[1:h - 1, 1:w - 1, :3] = clampF(s / 16.0)
    return out


def sharpen(img, strength):
    """effects.go:10-45"""
    if strength <= 0:
        return img
    strength = min(strength, 1)
    h, w = img.shape[:2]
    if w < 3 or h < 3:
        return img
    blur = blur3x3(img)[..., :3].astype(np.float64)
    orig = img[..., :3].astype(np.float64)
    amount = 1.0 + strength * 1.5
    out = np.empty_like(img)
    out[..., :3] = clampF(orig + amount * (orig - blur))
    out[..., 3] = img[..., 3]
    return out


def edge_strength(img):
    """localEdgeStrength (effects.go:93-112) on the interior, shape (h-2, w-2)."""
    L = to_luminance(img)
    h, w = L.shape
    def at(dx, dy):
        return L[1 + dy:h - 1 + dy, 1 + dx:w - 1 + dx]
    gx = ((((-at(-1, -1) + at(1, -1)) - 2 * at(-1, 0)) + 2 * at(1, 0)) - at(-1, 1)) + at(1, 1)
    gy = ((((-at(-1, -1) - 2 * at(0, -1)) - at(1, -1)) + at(-1, 1)) + 2 * at(0, 1)) + at(1, 1)
    mag = np.sqrt(gx * gx + gy * gy)
    return np.minimum(mag / 400.0, 1.0)


def adaptive_sharpen(img, strength):
    """effects.go:49-90"""
    if strength <= 0:
        return img
    strength = min(strength, 1)
    h, w = img.shape[:2]
    if w < 3 or h < 3:
        return img
    blur = blur3x3(img)[1:h - 1, 1:w - 1, :3].astype(np.float64)
    orig = img[1:h - 1, 1:w - 1, :3].astype(np.float64)
    amount = 1.0 + strength * 2.0
    local = amount * edge_strength(img)
    out = flat_pix_image(img)                          # effects.go:68
    out[1:h - 1, 1:w - 1, :3] = clampF(orig + local[..., None] * (orig - blur))
    out[1:h - 1, 1:w - 1, 3] = img[1:h - 1, 1:w - 1, 3]      # effects.go:85
    return out


# ------------------------------------------------------------------ orientation
def apply_orientation(img, orient):
    """exif.go:178-203 over convert.go:186-256"""
    rot90 = lambda m: np.ascontiguousarray(np.rot90(m, k=-1))   # clockwise
    rot270 = lambda m: np.ascontiguousarray(np.rot90(m, k=1))
    fliph = lambda m: np.ascontiguousarray(m[:, ::-1])
    if orient == 2:
        return fliph(img)
    if orient == 3:
        return np.ascontiguousarray(img[::-1, ::-1])
    if orient == 4:
        return np.ascontiguousarray(img[::-1])
    if orient == 5:
        return fliph(rot270(img))
    if orient == 6:
        return rot90(img)
    if orient == 7:
        return fliph(rot90(img))
    if orient == 8:
        return rot270(img)
    return img


# ------------------------------------------------------------------ analyze.go / convert.go scans
def analyze(img):
    """Analyze (analyze.go:26-124) + computeEdgeDensity (analyze.go:142-184): dict of the
    ImageStats fields and raw accumulators (entropy / recommendations left to the caller)."""
    h, w = img.shape[:2]
    out = dict(width=w, height=h)
    if w == 0 or h == 0:
        return out
    px = img.reshape(-1, 4)
    r, g, b, a = (px[:, k] for k in range(4))
    lum = (0.299 * r.astype(np.float64) + 0.587 * g.astype(np.float64)) + 0.114 * b.astype(np.float64)
    out["bright_sum"] = seq_sum(lum)                       # brightSum += lum, raster order
    out["histogram"] = np.bincount((lum + 0.5).astype(np.int64), minlength=256).astype(np.float64)
    out["has_alpha"] = int(np.any(a < 255))
    out["is_grayscale"] = int(np.all((r == g) & (g == b)))
    step = 1
    if w * h > 50000:
        step = w * h // 50000
    keys = (px[::step].astype(np.uint32) << np.array([24, 16, 8, 0], dtype=np.uint32)).sum(axis=1, dtype=np.uint32)
    # the set stops growing at 1024 entries; before that it holds every distinct sampled key
    out["unique_colors"] = min(1024, len(np.unique(keys)))
    n = float(w * h)
    mean = out["bright_sum"] / n
    out["mean_brightness"] = mean
    step_y = int(max(1, math.ceil(h / 100)))
    step_x = int(max(1, math.ceil(w / 100)))
    grid = img[::step_y, ::step_x].astype(np.float64)
    glum = (0.299 * grid[..., 0] + 0.587 * grid[..., 1]) + 0.114 * grid[..., 2]
    d = glum - mean
    out["variance_sum"] = seq_sum(d * d)
    out["sample_count"] = int(d.size)
    out["contrast"] = math.sqrt(out["variance_sum"] / d.size)
    out["edge_count"] = out["edge_total"] = 0
    out["edge_density"] = 0.0
    if w >= 3 and h >= 3:
        sx = int(max(1, w / 200))
        sy = int(max(1, h / 200))
        L = to_luminance(img)
        ys = np.arange(1, h - 1, sy)[:, None]
        xs = np.arange(1, w - 1, sx)[None, :]
        gx = ((((L[ys - 1, xs + 1] - L[ys - 1, xs - 1]) + 2 * L[ys, xs + 1]) - 2 * L[ys, xs - 1]) +
              L[ys + 1, xs + 1]) - L[ys + 1, xs - 1]
        gy = ((((L[ys + 1, xs - 1] - L[ys - 1, xs - 1]) + 2 * L[ys + 1, xs]) - 2 * L[ys - 1, xs]) +
              L[ys + 1, xs + 1]) - L[ys - 1, xs + 1]
        mag = np.sqrt(gx * gx + gy * gy)
        out["edge_count"] = int(np.count_nonzero(mag > 30.0))
        out["edge_total"] = int(mag.size)
        out["edge_density"] = out["edge_count"] / out["edge_total"]
    return out


def is_opaque(img):
    """convert.go:66-74 on a tight image"""
    return bool(np.all(img[..., 3] == 255))


def is_grayscale(img):
    """convert.go:76-84 on a tight image"""
    return bool(np.all((img[..., 0] == img[..., 1]) & (img[..., 1] == img[..., 2])))


def analyze_format(img):
    """analyzeFormat (convert.go:105-146): 1 JPEG, 2 PNG"""
    h, w = img.shape[:2]
    step = 1
    if w * h > 10000:
        step = max(1, w * h // 10000)
    px = img.reshape(-1, 4)[::step]
    seen = set()
    has_alpha = False
    for p in px:                       # order matters: the scan stops at 512 distinct colours
        if len(seen) >= 512:
            break
        if p[3] < 255:
            has_alpha = True
        seen.add(bytes(p))
    if has_alpha or len(seen) < 256:
        return 2
    return 1


def apply_palette(img, palette):
    """applyPalette + palettedToNRGBA (targetsize.go:488-546); opaque palette"""
    pal = np.asarray(palette, dtype=np.int64).reshape(-1, 4)
    rgb = img[..., :3].astype(np.int64)
    dist = ((rgb[:, :, None, :] - pal[None, None, :, :3]) ** 2).sum(axis=3)
    idx = np.argmin(dist, axis=2).astype(np.uint8)          # first minimum, like the strict `<`
    q = np.asarray(palette, dtype=np.uint8).reshape(-1, 4)[idx]
    return idx, q


def ycbcr_to_nrgba(y, cb, cr, ratio):
    """convertToNRGBA (convert.go:34-64) of an image.YCbCr: COffset + color.YCbCr.RGBA() of Go's
    standard library (published algorithm), opaque branch uint8(c >> 8)."""
    h, w = y.shape
    out = np.empty((h, w, 4), dtype=np.uint8)
    out[..., 3] = 255
    if cb is None:
        out[..., 0] = out[..., 1] = out[..., 2] = (y.astype(np.uint32) * 0x101) >> 8
        return out
    xs = np.arange(w) >> [0, 1, 1, 0, 2, 2][ratio]
    ys = np.arange(h) >> [0, 0, 1, 1, 0, 1][ratio]
    cb1 = cb[ys[:, None], xs[None, :]].astype(np.int64) - 128
    cr1 = cr[ys[:, None], xs[None, :]].astype(np.int64) - 128
    yy1 = y.astype(np.int64) * 0x10101
    for k, v in enumerate((yy1 + 91881 * cr1, yy1 - 22554 * cb1 - 46802 * cr1, yy1 + 116130 * cb1)):
        c16 = np.where(v < 0, 0, np.where(v >= (1 << 24), 0xffff, v >> 8))
        out[..., k] = c16 >> 8
    return out
