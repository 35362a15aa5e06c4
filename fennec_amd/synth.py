"""Deterministic synthetic NRGBA images (numpy, host side).

These restate the generators of the reference's own tests so that parity tests
and the benchmark run on the inputs the reference is tested with
(fennec_test.go:20-76, testdata_generate_test.go:70-82; SURVEY.md Appendix B and
section 8(d)).  Images are uint8 arrays of shape (h, w, 4) in R,G,B,A order --
the byte layout of Go's image.NRGBA.Pix with Stride = 4*w.
"""
from __future__ import annotations

import numpy as np


def _grid(w: int, h: int):
    y, x = np.mgrid[0:h, 0:w]
    return x.astype(np.int64), y.astype(np.int64)


def make_test_image(w: int, h: int) -> np.ndarray:
    """makeTestImage (fennec_test.go:20-32): integer-division gradients, A=255."""
    x, y = _grid(w, h)
    img = np.empty((h, w, 4), dtype=np.uint8)
    img[..., 0] = (x * 255 // max(w, 1)).astype(np.uint8)
    img[..., 1] = (y * 255 // max(h, 1)).astype(np.uint8)
    img[..., 2] = ((x + y) % 256).astype(np.uint8)
    img[..., 3] = 0xFF
    return img


def make_test_image_with_alpha(w: int, h: int) -> np.ndarray:
    """makeTestImageWithAlpha (fennec_test.go:34-43): A = x*255/w."""
    img = make_test_image(w, h)
    x, _ = _grid(w, h)
    img[..., 3] = (x * 255 // max(w, 1)).astype(np.uint8)
    return img


def make_solid_image(w: int, h: int, rgba) -> np.ndarray:
    """makeSolidImage (fennec_test.go:45-54)."""
    img = np.empty((h, w, 4), dtype=np.uint8)
    img[...] = np.asarray(rgba, dtype=np.uint8)
    return img


def make_striped_image(w: int, h: int, stripe_width: int) -> np.ndarray:
    """makeStripedImage (fennec_test.go:58-76): (200,50,100) / (50,200,100) stripes."""
    x, _ = _grid(w, h)
    even = ((x // stripe_width) % 2) == 0
    img = np.empty((h, w, 4), dtype=np.uint8)
    img[..., 0] = np.where(even, 200, 50)
    img[..., 1] = np.where(even, 50, 200)
    img[..., 2] = 100
    img[..., 3] = 255
    return img


def large_photo(w: int, h: int, k: int = 0) -> np.ndarray:
    """The reference's `large_photo` pattern (testdata_generate_test.go:70-82)
    salted per image k: R=(xy+3x+17k)%256, G=(xy+7y+31k)%256,
    B=(x+11y+5k)%256, A=255.  k=0 is the reference's fixture verbatim."""
    x, y = _grid(w, h)
    img = np.empty((h, w, 4), dtype=np.uint8)
    img[..., 0] = ((x * y + 3 * x + 17 * k) % 256).astype(np.uint8)
    img[..., 1] = ((x * y + 7 * y + 31 * k) % 256).astype(np.uint8)
    img[..., 2] = ((x + 11 * y + 5 * k) % 256).astype(np.uint8)
    img[..., 3] = 255
    return img


def large_photo_batch(w: int, h: int, ks) -> list:
    """[large_photo(w, h, k) for k in ks], computed from one base image: the salt only adds a constant to each
    channel mod 256, which is uint8 wrap-around addition (a 4K image in ~40 ms instead of ~1 s)."""
    base = large_photo(w, h, 0)
    out = []
    for k in ks:
        img = base.copy()
        img[..., 0] += np.uint8((17 * k) % 256)
        img[..., 1] += np.uint8((31 * k) % 256)
        img[..., 2] += np.uint8((5 * k) % 256)
        out.append(img)
    return out


def noise_image(w: int, h: int, seed: int, alpha: bool = False) -> np.ndarray:
    """Seeded uniform noise (not in the reference; used to hit rounding ties)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    if not alpha:
        img[..., 3] = 255
    return img


def ycbcr_planes(w: int, h: int, ratio: int, seed: int):
    """Seeded image.YCbCr planes as image.NewYCbCr lays them out (Rect.Min = 0,0).
    ratio: image.YCbCrSubsampleRatio (0 4:4:4, 1 4:2:2, 2 4:2:0, 3 4:4:0, 4 4:1:1, 5 4:1:0).
    Full-range noise, so the saturating branches of color.YCbCr.RGBA() are exercised."""
    rng = np.random.default_rng(seed)
    cw = [w, (w + 1) // 2, (w + 1) // 2, w, (w + 3) // 4, (w + 3) // 4][ratio]
    ch = [h, h, (h + 1) // 2, (h + 1) // 2, h, (h + 1) // 2][ratio]
    y = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    cb = rng.integers(0, 256, size=(ch, cw), dtype=np.uint8)
    cr = rng.integers(0, 256, size=(ch, cw), dtype=np.uint8)
    return y, cb, cr


def rgb_to_ycbcr_planes(img: np.ndarray, ratio: int = 2):
    """A plausible decoder output for an NRGBA image (JFIF forward transform, chroma box-averaged):
    only used to get photograph-like planes for tests -- not a restatement of any encoder."""
    r, g, b = (img[..., k].astype(np.float64) for k in range(3))
    y = np.clip(np.rint(0.299 * r + 0.587 * g + 0.114 * b), 0, 255).astype(np.uint8)
    cbf = 128 - 0.168736 * r - 0.331264 * g + 0.5 * b
    crf = 128 + 0.5 * r - 0.418688 * g - 0.081312 * b
    h, w = y.shape
    fx = [1, 2, 2, 1, 4, 4][ratio]
    fy = [1, 1, 2, 2, 1, 2][ratio]
    cw, ch = (w + fx - 1) // fx, (h + fy - 1) // fy

    def sub(p):
        pad = np.pad(p, ((0, ch * fy - h), (0, cw * fx - w)), mode="edge")
        return np.clip(np.rint(pad.reshape(ch, fy, cw, fx).mean(axis=(1, 3))), 0, 255).astype(np.uint8)
    return y, sub(cbf), sub(crf)
