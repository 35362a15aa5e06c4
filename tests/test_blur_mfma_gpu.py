"""csrc/blur_mfma.hip (GaussianBlur, radius <= 6, on the i8 matrix pipe) against the oracle: shapes on both sides of every
tile boundary it has (64-px strips, 16-row sets, segment ends), every radius it takes and the ones next to them, pitched
sources and destinations, batches that change its segment length, alpha planes, images built from rounding ties, and the
one-pass SSIMFast form against the two-call route.  Exact mode `==`; fast mode <= 1 LSB on <= 0.01 % of samples (the
integer sums are exact: only the 2^-24 weights differ from the reference's).  FNX_BLUR_MFMA=0 runs the same file through
blur.hip's direct kernels."""
import numpy as np
import pytest

import fennec_amd
from fennec_amd import synth

pytestmark = pytest.mark.gpu

FAST_FRAC = 1e-4


@pytest.fixture(scope="module")
def ctx():
    return fennec_amd.Context(0)


def _close(got, want):
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= 1, f"max diff {diff.max()}"
    n_off = int((diff[..., :3] != 0).sum())
    # one intermediate sample that rounds the other way (2.8e-6 of them do) can move the 3-5 output rows it weighs most:
    # small images get that much room, large ones are held to the rate
    assert n_off <= max(5, FAST_FRAC * diff[..., :3].size), f"{n_off} mismatching samples of {diff[..., :3].size}"
    assert np.array_equal(got[..., 3], want[..., 3])


SHAPES = [(64, 32), (65, 33), (70, 47), (127, 48), (128, 49), (129, 100), (200, 271), (257, 272), (320, 273), (63 + 64 * 3, 545),
          (1000, 31 + 16 * 7), (2049, 64), (64, 2000)]


@pytest.mark.parametrize("w,h", SHAPES)
def test_shapes(ctx, orc, w, h):
    img = synth.noise_image(w, h, 3 * w + h, alpha=True)
    want = orc.gaussian_blur(img, 2.0, procs=8)
    assert np.array_equal(ctx.GaussianBlur(img, 2.0, exact=True), want)
    _close(ctx.GaussianBlur(img, 2.0), want)


@pytest.mark.parametrize("sigma", [0.6, 0.7, 0.9, 1.0, 1.3, 1.34, 1.6, 1.67, 1.9, 2.0, 2.01, 2.3])
def test_radii(ctx, orc, sigma):
    """radius 2 .. 7: tables the kernel takes (centre weight under 0.49, radius <= 6) and the ones just outside"""
    for img in (synth.noise_image(333, 211, 5, alpha=True), synth.large_photo(640, 480, 2)):
        want = orc.gaussian_blur(img, sigma, procs=4)
        assert np.array_equal(ctx.GaussianBlur(img, sigma, exact=True), want)
        _close(ctx.GaussianBlur(img, sigma), want)


def test_pitched_device_views(ctx, orc):
    import torch
    big = torch.from_numpy(synth.noise_image(1500, 900, 77, alpha=True)).cuda()
    torch.cuda.synchronize()
    for (y0, x0, hh, ww) in ((0, 0, 900, 1500), (7, 13, 600, 1000), (100, 1, 333, 1499), (0, 1436, 900, 64)):
        view = big[y0:y0 + hh, x0:x0 + ww]
        host = np.ascontiguousarray(view.cpu().numpy())
        want = orc.gaussian_blur(host, 2.0, procs=8)
        assert np.array_equal(ctx.GaussianBlur(view, 2.0, exact=True).cpu().numpy(), want)
        _close(ctx.GaussianBlur(view, 2.0).cpu().numpy(), want)


@pytest.mark.parametrize("n", [1, 2, 7])
def test_batches_change_the_segment_length(ctx, orc, n):
    """few images: many short segments (the grid must fill the chip); many: long ones"""
    import torch
    imgs = [synth.noise_image(1024, 1100, 11 + k, alpha=True) for k in range(n)]
    d = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    for exact in (True, False):
        outs = ctx.GaussianBlurBatch(d, 2.0, exact=exact)
        for i, o in zip(imgs, outs):
            want = orc.gaussian_blur(i, 2.0, procs=8)
            if exact:
                assert np.array_equal(o.cpu().numpy(), want)
            else:
                _close(o.cpu().numpy(), want)


def _binomial(radius):
    k = np.array([1.0])
    for _ in range(2 * radius):
        k = np.convolve(k, [0.5, 0.5])
    return k


@pytest.mark.parametrize("radius", [2, 3, 4, 5, 6])
def test_every_sample_a_tie(ctx, orc, radius):
    """dyadic weights on stripe images: every H sample, or every V sample, sits exactly on x.5 -- every sample of every
    set is flagged and recomputed; and random bytes mixed with a tie block"""
    k = _binomial(radius)
    assert fennec_amd.blur_fixed_point(k) is not None
    w, h = 448, 200
    cols = np.zeros((h, w, 4), np.uint8); cols[:, 1::2, :3] = 1; cols[..., 3] = 255
    rows = np.zeros((h, w, 4), np.uint8); rows[1::2, :, :3] = 1; rows[..., 3] = 7
    mixed = synth.noise_image(w, h, radius, alpha=True); mixed[40:160, 100:300, :3] &= 1
    for img in (cols, rows, mixed):
        want = orc.gaussian_blur(img, 1.0, kernel=k, procs=4)
        assert np.array_equal(ctx.GaussianBlur(img, 1.0, exact=True, kernel=k), want)


def test_alpha_rides_through(ctx, orc):
    img = synth.noise_image(512, 300, 1, alpha=True)
    img[..., 3] = np.arange(512 * 300, dtype=np.uint32).reshape(300, 512) % 256
    for exact in (True, False):
        got = ctx.GaussianBlur(img, 2.0, exact=exact)
        assert np.array_equal(got[..., 3], img[..., 3])
    flat = np.full((200, 320, 4), 200, np.uint8)
    assert np.array_equal(ctx.GaussianBlur(flat, 2.0), flat)     # a constant image is a fixed point: sum wq == 2^24 exactly


@pytest.mark.parametrize("w,h", [(3840, 2160), (2560, 1440), (4096, 3072), (7680, 4320), (3000, 272 * 3 + 5), (5000, 300)])
def test_one_pass_equals_two_calls(ctx, orc, w, h):
    """the SCORE form: blurred images and SSIMFast scores bit-identical to GaussianBlurBatch + SSIMFastBatch, both modes,
    and the score is the oracle's SSIMFast of the returned pair"""
    import torch
    imgs = [synth.noise_image(w, h, w + 7 * h, alpha=True), synth.large_photo(w, h, 4)]
    d = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    for exact in (False, True):
        outs, ss = ctx.GaussianBlurSSIMFastBatch(d, 2.0, exact=exact)
        ref = ctx.GaussianBlurBatch(d, 2.0, exact=exact)
        ref_ss = ctx.SSIMFastBatch(d, ref)
        for k in range(len(imgs)):
            assert torch.equal(outs[k], ref[k])
            assert ss[k] == ref_ss[k]
        if w * h <= 3840 * 2160:
            assert abs(ss[0] - orc.ssim_fast(imgs[0], outs[0].cpu().numpy(), procs=16)) <= 1e-9
            if exact:
                assert np.array_equal(outs[0].cpu().numpy(), orc.gaussian_blur(imgs[0], 2.0, procs=16))


@pytest.mark.parametrize("w,h", [(3840, 2160), (2560, 1440), (3000, 272 * 3 + 5)])
def test_two_calls_with_kept_box_sums(ctx, orc, w, h):
    """FNX_BLUR_KEEP_BOX_SUMS: fnx_gaussian_blur_batch runs the SCORE form and the fnx_ssim_fast_batch that
    follows over the same pairs reads no image -- the same bytes and the same scores as the one-pass entry and as the plain two
    calls; any other call in between, other pairs, or a shape the one-pass kernel does not take drops the sums."""
    import torch
    imgs = [synth.noise_image(w, h, w + 7 * h, alpha=True), synth.large_photo(w, h, 4), synth.large_photo(w, h, 9)]
    d = [torch.from_numpy(i).cuda() for i in imgs]
    outs = [torch.empty_like(t) for t in d]
    other = [torch.empty_like(t) for t in d]
    torch.cuda.synchronize()
    for exact in (False, True):
        for sigma in (2.0, 3.0):
            keep = ctx.plan_blur_batch(d, sigma, outs=outs, exact=exact, keep_box_sums=True)
            score = ctx.plan_ssim_fast_batch(d, outs)
            keep.run()
            scored = "SCORE" in ctx.last_kernel(1)                                   # (some shapes have no one-pass form at some radii)
            assert scored or (w, h, sigma) != (3840, 2160, 2.0)
            got = score.run().copy()
            assert ctx.last_kernel(2).startswith("kept box planes") == scored        # the one-pass tail, not box_tiled + windowed_ssim
            ctx.sync()
            kept_bytes = [o.clone() for o in outs]
            ref_outs, ref_ss = ctx.GaussianBlurSSIMFastBatch(d, sigma, exact=exact)
            plain_outs = ctx.GaussianBlurBatch(d, sigma, exact=exact)
            plain_ss = ctx.SSIMFastBatch(d, plain_outs)
            assert not ctx.last_kernel(2).startswith("kept box planes")
            for k in range(len(imgs)):
                assert torch.equal(kept_bytes[k], ref_outs[k]) and torch.equal(kept_bytes[k], plain_outs[k])
                assert got[k] == ref_ss[k] == plain_ss[k]
    assert abs(got[1] - orc.ssim_fast(imgs[1], kept_bytes[1].cpu().numpy(), procs=16)) <= 1e-9
    # a call in between (here: the images change under a sync): the sums are dropped, the scores are those of the images as they are
    plain_ss = ctx.SSIMFastBatch(d, ctx.GaussianBlurBatch(d, 2.0))
    keep = ctx.plan_blur_batch(d, 2.0, outs=outs, keep_box_sums=True)
    score = ctx.plan_ssim_fast_batch(d, outs)
    keep.run()
    ctx.sync()
    outs[0].copy_(d[0])
    torch.cuda.synchronize()
    got = score.run().copy()
    assert not ctx.last_kernel(2).startswith("kept box planes") and got[0] == 1.0 and got[1] == plain_ss[1]
    # other pairs than the blur's: dropped
    ctx.plan_blur_batch(d, 2.0, outs=other, keep_box_sums=False).run()
    keep.run()
    got = ctx.plan_ssim_fast_batch(d, other).run().copy()
    assert not ctx.last_kernel(2).startswith("kept box planes") and got[1] == plain_ss[1]
    got = score.run().copy()                                                         # (and the kept set is gone for good)
    assert not ctx.last_kernel(2).startswith("kept box planes") and got[1] == plain_ss[1]
    # two kept blurs in a row, the second one consumed; then the enqueue / fetch form
    keep.run(); keep.run()
    scored = "SCORE" in ctx.last_kernel(1)
    got = score.run().copy()
    assert ctx.last_kernel(2).startswith("kept box planes") == scored and got[1] == plain_ss[1] and got[2] == plain_ss[2]
    keep.run(); score.enqueue()
    assert np.array_equal(score.fetch(), got)
    # a subset of the batch is another batch
    keep.run()
    sub = ctx.plan_ssim_fast_batch(d[:2], outs[:2]).run().copy()
    assert not ctx.last_kernel(2).startswith("kept box planes") and sub[1] == plain_ss[1]


def test_single_calls_with_kept_box_sums(ctx, orc):
    """the per-image entry points (what a cgo caller holding device-resident images calls): fnx_gaussian_blur with the flag, then
    fnx_ssim_fast -- through the binding, whose stream lending between the two does not count as a call in between"""
    import torch
    img = synth.large_photo(3840, 2160, 6)
    d = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()
    for exact in (False, True):
        ref = ctx.GaussianBlur(d, 2.0, exact=exact)
        ref_s = ctx.SSIMFast(d, ref)
        assert not ctx.last_kernel(2).startswith("kept box planes")
        out = ctx.GaussianBlur(d, 2.0, exact=exact, keep_box_sums=True)
        assert "SCORE" in ctx.last_kernel(1)
        s = ctx.SSIMFast(d, out)
        assert ctx.last_kernel(2).startswith("kept box planes")
        assert torch.equal(out, ref) and abs(s - ref_s) <= 1e-12
    assert abs(s - orc.ssim_fast(img, out.cpu().numpy(), procs=16)) <= 1e-9
    # scored against something else: dropped; a host image: the flag means nothing
    out = ctx.GaussianBlur(d, 2.0, keep_box_sums=True)
    assert ctx.SSIMFast(out, d) == ctx.SSIMFast(out, d) and not ctx.last_kernel(2).startswith("kept box planes")
    small = synth.large_photo(640, 480, 1)
    assert np.array_equal(ctx.GaussianBlur(small, 2.0, exact=True, keep_box_sums=True), ctx.GaussianBlur(small, 2.0, exact=True))


def test_kept_box_sums_on_shapes_without_a_one_pass_form(ctx):
    import torch
    for (w, h, sigma) in ((640, 480, 2.0), (3840, 2160, 9.0), (1700, 1000, 2.0)):   # no downsample; radius 27; a box ratio the one-pass form leaves
        imgs = [synth.large_photo(w, h, 2), synth.noise_image(w, h, 5, alpha=True)]
        d = [torch.from_numpy(i).cuda() for i in imgs]
        outs = [torch.empty_like(t) for t in d]
        torch.cuda.synchronize()
        ctx.plan_blur_batch(d, sigma, outs=outs, keep_box_sums=True).run()
        assert "SCORE" not in ctx.last_kernel(1)
        got = ctx.plan_ssim_fast_batch(d, outs).run().copy()
        ctx.sync()
        want_outs = ctx.GaussianBlurBatch(d, sigma)
        assert all(torch.equal(a, b) for a, b in zip(outs, want_outs)) and np.array_equal(got, ctx.SSIMFastBatch(d, want_outs))


@pytest.mark.parametrize("sigma", [2.4, 3.0, 4.6, 4.7, 5.0, 6.0, 7.3, 7.4, 8.0, 8.1])
def test_wide_radii(ctx, orc, sigma):
    """radius 8 .. 24 on blur_mfma_wide_kernel (two, three, four 64-byte K chunks per H set; 64-row V window), 25: blur.hip"""
    for img in (synth.noise_image(517, 301, int(sigma * 10), alpha=True), synth.large_photo(1030, 620, 3)):
        want = orc.gaussian_blur(img, sigma, procs=8)
        assert np.array_equal(ctx.GaussianBlur(img, sigma, exact=True), want)
        _close(ctx.GaussianBlur(img, sigma), want)


@pytest.mark.parametrize("radius", [7, 10, 14, 15, 19, 22, 23, 24])
def test_wide_every_sample_a_tie(ctx, orc, radius):
    k = _binomial(radius)
    w, h = 320, 176
    cols = np.zeros((h, w, 4), np.uint8); cols[:, 1::2, :3] = 1; cols[..., 3] = 255
    rows = np.zeros((h, w, 4), np.uint8); rows[1::2, :, :3] = 1; rows[..., 3] = 9
    mixed = synth.noise_image(w, h, radius, alpha=True); mixed[40:140, 100:260, :3] &= 1
    for img in (cols, rows, mixed):
        assert np.array_equal(ctx.GaussianBlur(img, 1.0, exact=True, kernel=k), orc.gaussian_blur(img, 1.0, kernel=k, procs=4))


def test_wide_4k_sigma6(ctx, orc):
    img = synth.noise_image(3840, 2160, 66, alpha=True)
    want = orc.gaussian_blur(img, 6.0, procs=32)
    assert np.array_equal(ctx.GaussianBlur(img, 6.0, exact=True), want)
    _close(ctx.GaussianBlur(img, 6.0), want)


@pytest.mark.parametrize("w,h", [(1860, 1002), (1849, 855), (640, 1880), (1887, 1760), (2040, 600)])
def test_one_pass_where_the_box_geometry_does_not_fit(ctx, orc, w, h):
    """long side 1843..2047: SSIMFast downsamples by 3.6..4 (3-px boxes), outside the matrix kernel's indicator matrix; the
    one-pass entry point must still return the two-call route's bytes (found by tools/fuzz_blur.py)"""
    import torch
    imgs = [synth.noise_image(w, h, w + h, alpha=True), synth.large_photo(w, h, 1)]
    d = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    for sigma in (1.0, 2.0):
        for exact in (False, True):
            outs, ss = ctx.GaussianBlurSSIMFastBatch(d, sigma, exact=exact)
            ref = ctx.GaussianBlurBatch(d, sigma, exact=exact)
            ref_ss = ctx.SSIMFastBatch(d, ref)
            for k in range(len(imgs)):
                assert torch.equal(outs[k], ref[k]) and ss[k] == ref_ss[k]


# ---- r5: radii 25 .. 62 (two / three chained 64-row V instructions, five to eight H chunks) ----
@pytest.mark.parametrize("sigma", [8.1, 8.4, 10.0, 12.6, 12.7, 15.0, 15.4, 16.0, 18.0, 18.1, 19.0, 20.0, 20.6, 20.7, 22.0])
def test_very_wide_radii(ctx, orc, sigma):
    """radius 25 .. 62 on blur_mfma_wide_kernel<5 .. 8>: on both sides of every frame boundary (38 / 39, 46 / 47, 54 / 55, 62 / 63),
    beyond 62 the generic passes; images shorter and narrower than the window, segment ends, an alpha plane"""
    radius = int(np.ceil(3 * sigma))
    for img in (synth.noise_image(517, 301, int(sigma * 10), alpha=True), synth.large_photo(1030, 620, 3), synth.noise_image(70, 41, 5, alpha=True)):
        want = orc.gaussian_blur(img, sigma, procs=8)
        assert np.array_equal(ctx.GaussianBlur(img, sigma, exact=True), want), (sigma, img.shape)
        if radius <= 62:
            assert "blur_mfma_wide_kernel" in ctx.last_kernel(1), ctx.last_kernel(1)
        _close(ctx.GaussianBlur(img, sigma), want)


@pytest.mark.parametrize("radius", [25, 31, 38, 39, 46, 47, 54, 55, 60, 62])
def test_very_wide_every_sample_a_tie(ctx, orc, radius):
    k = _binomial(radius)
    w, h = 320, 240
    cols = np.zeros((h, w, 4), np.uint8); cols[:, 1::2, :3] = 1; cols[..., 3] = 255
    rows = np.zeros((h, w, 4), np.uint8); rows[1::2, :, :3] = 1; rows[..., 3] = 9
    mixed = synth.noise_image(w, h, radius, alpha=True); mixed[40:200, 100:260, :3] &= 1
    for img in (cols, rows, mixed):
        assert np.array_equal(ctx.GaussianBlur(img, 1.0, exact=True, kernel=k), orc.gaussian_blur(img, 1.0, kernel=k, procs=4))


def test_very_wide_4k_and_views(ctx, orc):
    import torch
    img = synth.noise_image(3840, 2160, 67, alpha=True)
    for sigma in (10.0, 20.0):
        want = orc.gaussian_blur(img, sigma, procs=32)
        assert np.array_equal(ctx.GaussianBlur(img, sigma, exact=True), want)
        _close(ctx.GaussianBlur(img, sigma), want)
    big = torch.from_numpy(synth.noise_image(1500, 900, 21, alpha=True)).cuda()
    view = big[7:7 + 700, 13:13 + 1201]                           # pitched, 4-byte aligned only
    host = np.ascontiguousarray(view.cpu().numpy())
    outs = ctx.GaussianBlurBatch([view, view], 13.0, exact=True)
    want = orc.gaussian_blur(host, 13.0, procs=16)
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), want)


def test_generic_passes_past_53_taps_run_in_fp64(ctx, orc):
    """what the matrix kernels leave at large radii (radius 63, images narrower than 64 px): an fp32 accumulator put 0.4 % of the
    samples of few-colour images one LSB off at 127 taps (tools/fuzz_blur.py seed 71); the fast mode takes the fp64 passes there"""
    few = synth.noise_image(829, 682, 11, alpha=True); few[..., :3] &= 0xF0
    want = orc.gaussian_blur(few, 21.0, procs=16)
    got = ctx.GaussianBlur(few, 21.0)
    assert "double" in ctx.last_kernel(1), ctx.last_kernel(1)
    assert np.array_equal(got, want)
    narrow = synth.noise_image(61, 129, 12, alpha=True); narrow[..., :3] &= 0xF0
    assert np.array_equal(ctx.GaussianBlur(narrow, 15.4), orc.gaussian_blur(narrow, 15.4, procs=4))
