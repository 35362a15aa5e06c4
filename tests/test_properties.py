"""Size-independent properties of the oracle (CPU, hypothesis): what the domain guarantees whatever the
input, stated against the Go source's structure.  They pin the restatement from a different side than
the invariants the reference's own tests assert (tests/test_oracle.py)."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from fennec_amd import synth

dims = st.tuples(st.integers(1, 40), st.integers(1, 40))
seeds = st.integers(0, 2 ** 31 - 1)
FAST = dict(max_examples=25, deadline=None)


def _img(w, h, seed):
    return synth.noise_image(w, h, seed, alpha=True)


# composition table of the EXIF orientations as exif.go:178-203 builds them from rot90/rot180/rot270/flipH/flipV:
# applying o then its inverse restores the image (5 and 7 are involutions, 6 and 8 are each other's inverse)
INVERSE = {2: 2, 3: 3, 4: 4, 5: 5, 6: 8, 7: 7, 8: 6}


@settings(**FAST)
@given(dims, seeds, st.sampled_from(sorted(INVERSE)))
def test_orientation_inverse(orc, wh, seed, o):
    img = _img(wh[0], wh[1], seed)
    assert np.array_equal(orc.apply_orientation(orc.apply_orientation(img, o), INVERSE[o]), img)


@settings(**FAST)
@given(dims, seeds)
def test_orientation_is_a_permutation(orc, wh, seed):
    img = _img(wh[0], wh[1], seed)
    want = np.sort(img.reshape(-1, 4).view(np.uint32).ravel())
    for o in range(2, 9):
        got = orc.apply_orientation(img, o)
        assert np.array_equal(np.sort(np.ascontiguousarray(got).reshape(-1, 4).view(np.uint32).ravel()), want)


@settings(**FAST)
@given(dims, st.tuples(st.integers(0, 255), st.integers(0, 255), st.integers(0, 255), st.integers(0, 255)),
       st.sampled_from([0.4, 1.0, 2.0, 3.1]))
def test_constant_images_are_fixed_points(orc, wh, colour, sigma):
    """kernels are normalised (effects.go:162-165, ssim.go:236-240): a constant image blurs, box-averages and
    3x3-blurs to itself; sharpening it changes nothing (orig - blur == 0)"""
    w, h = wh
    img = synth.make_solid_image(w, h, colour)
    assert np.array_equal(orc.gaussian_blur(img, sigma), img)
    assert np.array_equal(orc.blur3x3(img), img)
    assert np.array_equal(orc.box_downsample(img, max(1, w // 2), max(1, h // 2)),
                          synth.make_solid_image(max(1, w // 2), max(1, h // 2), colour))
    if w >= 3 and h >= 3:
        assert np.array_equal(orc.sharpen(img, 0.7), img)
        assert np.array_equal(orc.adaptive_sharpen(img, 0.7), img)


@settings(**FAST)
@given(st.tuples(st.integers(8, 40), st.integers(8, 40)), seeds, seeds)
def test_ssim_is_symmetric_and_one_on_identity(orc, wh, s1, s2):
    """the SSIM formula (ssim.go:142-145) is symmetric in its arguments term by term, and equals exactly 1
    when they coincide (2*mu*mu == mu*mu + mu*mu in floating point)"""
    a, b = _img(wh[0], wh[1], s1), _img(wh[0], wh[1], s2)
    assert orc.ssim(a, b) == orc.ssim(b, a)
    assert orc.ssim(a, a) == 1.0 and orc.ssim_fast(a, a) == 1.0
    assert -1.0 <= orc.ssim(a, b) <= 1.0


@settings(**FAST)
@given(st.integers(1, 300), st.integers(1, 300))
def test_lanczos_taps_are_normalised_and_in_range(orc, dst, src):
    """precomputeWeights (resize.go:164-197): every output's weights sum to 1, indices lie inside the source"""
    off, idx, wt = orc.precompute_weights(dst, src)
    assert len(off) == dst + 1 and off[0] == 0 and np.all(np.diff(off) >= 1)
    assert idx.min() >= 0 and idx.max() <= src - 1
    sums = np.add.reduceat(wt, off[:-1])
    assert np.allclose(sums, 1.0, rtol=0, atol=1e-12)


@settings(**FAST)
@given(dims, seeds)
def test_resize_to_same_size_is_a_copy(orc, wh, seed):      # resize.go:45-49
    img = _img(wh[0], wh[1], seed)
    assert np.array_equal(orc.lanczos_resize(img, wh[0], wh[1]), img)


@settings(**FAST)
@given(dims, seeds)
def test_opaque_resize_stays_opaque_and_box_is_bounded(orc, wh, seed):
    """premultiplied accumulation (resize.go:95-113): an opaque source gives an opaque result; box averages
    lie between the channel's min and max"""
    w, h = wh
    img = _img(w, h, seed)
    img[..., 3] = 255
    out = orc.lanczos_resize(img, w + 3, max(1, h - 1))
    assert np.all(out[..., 3] == 255)
    small = orc.box_downsample(img, max(1, w // 3), max(1, h // 3))
    for c in range(3):
        assert img[..., c].min() <= small[..., c].min() and small[..., c].max() <= img[..., c].max()


@settings(**FAST)
@given(st.floats(-1e6, 1e6, allow_nan=False))
def test_clampF_is_round_half_away_then_clamp(orc, x):      # convert.go:149-158
    import math
    want = int(min(255, max(0, math.floor(abs(x) + 0.5) * (1 if x >= 0 else -1))))
    assert orc.clampF(x) == want
