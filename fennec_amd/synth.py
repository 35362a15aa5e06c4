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


def noise_image(w: int, h: int, seed: int, alpha: bool = False) -> np.ndarray:
    """Seeded uniform noise (not in the reference; used to hit rounding ties)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    if not alpha:
        img[..., 3] = 255
    return img
