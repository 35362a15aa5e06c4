"""Independent anchors for the oracle (CPU): implementations of the same named algorithms that this repo did not write.

None of this pins the oracle to Go -- Pillow and scipy round and pad differently in places -- but a restatement that
had drifted from "Lanczos-3 with the usual centre convention", "Gaussian of radius ceil(3 sigma), clamp to edge" or
"3x3 binomial unsharp mask" would show here as more than a rounding's worth of difference."""
import numpy as np
import pytest

from fennec_amd import synth
from oracle import oracle as orc


def _smooth(w, h, seed):
    return orc.gaussian_blur(synth.noise_image(w, h, seed), 3.0)


@pytest.mark.parametrize("dst", [(160, 120), (213, 77), (640, 480), (500, 333)])
def test_lanczos_resize_close_to_pillow(dst):
    from PIL import Image
    img = _smooth(320, 240, 4)
    mine = orc.lanczos_resize(img, *dst)
    pil = np.asarray(Image.fromarray(img[..., :3], "RGB").resize(dst, Image.LANCZOS))
    d = np.abs(mine[..., :3].astype(int) - pil.astype(int))
    # Pillow accumulates in fixed point (8.22) and resizes the axes in the cheaper order: a level here and there
    assert d.max() <= 2 and d.mean() < 0.25, (d.max(), d.mean())


@pytest.mark.parametrize("sigma", [0.8, 2.0, 3.3])
def test_gaussian_blur_close_to_scipy(sigma):
    from scipy.ndimage import gaussian_filter1d
    img = synth.large_photo(200, 150, 3)
    mine = orc.gaussian_blur(img, sigma)
    radius = int(np.ceil(3 * sigma))                                  # effects.go:150
    f = img[..., :3].astype(np.float64)
    hpass = np.clip(np.floor(gaussian_filter1d(f, sigma, axis=1, mode="nearest", radius=radius) + 0.5), 0, 255)   # uint8 intermediate
    both = np.clip(np.floor(gaussian_filter1d(hpass, sigma, axis=0, mode="nearest", radius=radius) + 0.5), 0, 255)
    d = np.abs(mine[..., :3].astype(int) - both.astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 0.01, (d.max(), (d != 0).mean())       # last-ulp ties only
    assert np.array_equal(mine[..., 3], img[..., 3])


def test_blur3x3_and_sharpen_against_a_convolution():
    from scipy.ndimage import convolve
    img = synth.noise_image(97, 61, 8)
    k = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], dtype=np.int64)
    blur = np.stack([(convolve(img[..., c].astype(np.int64), k, mode="nearest") + 8) >> 4 for c in range(3)], axis=-1)
    s = 0.5
    amount = 1.0 + 1.5 * s                                            # effects.go:24
    o = img[..., :3].astype(np.float64)
    want = np.clip(np.floor(o + amount * (o - blur) + 0.5), 0, 255).astype(np.uint8)
    mine = orc.sharpen(img, s)
    # interior only: the reference copies the border (effects.go:120), scipy extends it
    assert np.array_equal(mine[1:-1, 1:-1, :3], want[1:-1, 1:-1])
    assert np.array_equal(mine[0], img[0]) and np.array_equal(mine[:, 0], img[:, 0])


def test_box_downsample_integer_ratio_is_the_block_mean():
    img = synth.noise_image(256, 192, 2, alpha=True)
    mine = orc.box_downsample(img, 64, 48)
    blocks = img.reshape(48, 4, 64, 4, 4).astype(np.float64).mean(axis=(1, 3))
    assert np.array_equal(mine, np.clip(np.floor(blocks + 0.5), 0, 255).astype(np.uint8))


def test_ssim_of_shifted_means_follows_the_closed_form():
    """Two flat images: every window has sigma = 0, so SSIM is the luminance term alone, (2ab + C1) / (a^2 + b^2 + C1)."""
    a = synth.make_solid_image(64, 48, (100, 100, 100, 255))
    b = synth.make_solid_image(64, 48, (140, 140, 140, 255))
    la, lb = 100.0, 140.0                                             # 0.299 + 0.587 + 0.114 = 1
    want = (2 * la * lb + 6.5025) / (la * la + lb * lb + 6.5025)
    assert abs(orc.ssim(a, b) - want) < 1e-12
    assert abs(orc.ssim_fast(a, b) - want) < 1e-12


def test_orientations_are_the_references_permutations():
    """Orientations 2, 3, 4, 6, 8 are the EXIF table's (what Pillow's exif_transpose applies).  For 5 and 7 the reference
    composes rot270 + flipH and rot90 + flipH (exif.go:188-197), which is the ANTI-diagonal flip for 5 and the diagonal one
    for 7 -- the opposite of the EXIF table's transpose / transverse.  The path reproduces the reference, quirk included."""
    from PIL import Image
    img = synth.noise_image(37, 23, 6, alpha=True)
    pil = Image.fromarray(img, "RGBA")
    ops = {2: Image.FLIP_LEFT_RIGHT, 3: Image.ROTATE_180, 4: Image.FLIP_TOP_BOTTOM, 6: Image.ROTATE_270, 8: Image.ROTATE_90}
    for o, op in ops.items():                                        # Pillow's ROTATE_270 is 90 degrees clockwise
        assert np.array_equal(orc.apply_orientation(img, o), np.asarray(pil.transpose(op))), o
    assert np.array_equal(orc.apply_orientation(img, 5), np.asarray(pil.transpose(Image.TRANSVERSE)))
    assert np.array_equal(orc.apply_orientation(img, 7), np.asarray(pil.transpose(Image.TRANSPOSE)))


def test_msssim_and_ssim_identity_and_symmetry():
    a = synth.large_photo(200, 150, 1)
    b = orc.gaussian_blur(a, 1.5)
    assert orc.ssim(a, a) == 1.0 and abs(orc.msssim(a, a) - 1.0) < 1e-12
    assert abs(orc.ssim(a, b) - orc.ssim(b, a)) < 1e-15               # the formula is symmetric in its arguments
    assert 0.0 < orc.msssim(a, b) < 1.0


def test_adaptive_sharpen_against_scipy_sobel():
    """effects.go:49-112 from its description: Sobel magnitude of BT.601 luminance / 400, capped at 1, scales an unsharp
    mask of the 3x3 binomial blur with amount 1 + 2 s; border pixels are copies."""
    from scipy.ndimage import convolve, sobel
    img = orc.gaussian_blur(synth.large_photo(120, 90, 5), 1.0)
    s = 0.6
    lum = 0.299 * img[..., 0].astype(np.float64) + 0.587 * img[..., 1] + 0.114 * img[..., 2]
    e = np.minimum(1.0, np.hypot(sobel(lum, axis=1, mode="nearest"), sobel(lum, axis=0, mode="nearest")) / 400.0)
    k = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], dtype=np.int64)
    blur = np.stack([(convolve(img[..., c].astype(np.int64), k, mode="nearest") + 8) >> 4 for c in range(3)], axis=-1)
    o = img[..., :3].astype(np.float64)
    want = np.clip(np.floor(o + (1.0 + 2.0 * s) * e[..., None] * (o - blur) + 0.5), 0, 255).astype(np.uint8)
    mine = orc.adaptive_sharpen(img, s)
    d = np.abs(mine[1:-1, 1:-1, :3].astype(int) - want[1:-1, 1:-1].astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 0.002, (d.max(), (d != 0).mean())   # the order of the fp64 operations differs
    assert np.array_equal(mine[0], img[0]) and np.array_equal(mine[..., 3], img[..., 3])
