"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs.

Bars (SURVEY.md Appendix A "stated tolerances"):
  * orient, boxDownsample, blur3x3, Sharpen, AdaptiveSharpen, lanczosResize, GaussianBlur
    in EXACT mode: bit-exact.
  * GaussianBlur FAST mode (fp32 FMA): max |delta| <= 1 LSB on <= 0.1 % of samples (<= 3 samples on images
    too small for a rate).
  * SSIM / SSIMFast / MSSSIM (fp64 moments): |delta| <= 1e-9.
"""
import numpy as np
import pytest

import fennec_amd
from fennec_amd import synth

pytestmark = pytest.mark.gpu

SSIM_TOL = 1e-9
BLUR_FAST_MAX_LSB = 1
BLUR_FAST_MAX_FRAC = 1e-3


@pytest.fixture(scope="module")
def _module_ctx():
    return fennec_amd.Context(0)


FORM_NAMES = ("fx_stream", "fx_pairs", "fx_ref", "resize_mfma", "resize_fp64", "resize_fused", "msssim_levelwise",
              "msssim_nofuse0", "msssim_fold", "msssim_boxfly", "palette_grid")


@pytest.fixture()
def ctx(_module_ctx):
    """the module's ctx; whatever kernel forms a test selected (fnx_ctx_set_form) are back at their defaults after it"""
    yield _module_ctx
    for name in FORM_NAMES:
        _module_ctx.set_form(name, None)


IMAGES = {
    "grad_64x48": lambda: synth.make_test_image(64, 48),
    "alpha_37x29": lambda: synth.make_test_image_with_alpha(37, 29),
    "photo_200x150": lambda: synth.large_photo(200, 150, 3),
    "noise_131x77": lambda: synth.noise_image(131, 77, 7, alpha=True),
    "stripes_100x100": lambda: synth.make_striped_image(100, 100, 10),
    "photo_640x480": lambda: synth.large_photo(640, 480, 0),
    "noise_3x5": lambda: synth.noise_image(3, 5, 1, alpha=True),
    "noise_1x1": lambda: synth.noise_image(1, 1, 2, alpha=True),
    "noise_300x2": lambda: synth.noise_image(300, 2, 4, alpha=True),
}


def assert_blur_close(got, want):
    diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= BLUR_FAST_MAX_LSB, f"max diff {diff.max()}"
    # a rate cannot be judged on a few hundred samples: images under 3000 samples may hold up to 3 off-by-one
    # samples (the measured rate is 5e-6: 26 of 5.4 M samples over 6000 images of 1..120 x 1..9 px)
    n_off = int((diff[..., :3] != 0).sum())
    assert n_off <= max(3, BLUR_FAST_MAX_FRAC * diff[..., :3].size), f"{n_off} mismatching samples of {diff[..., :3].size}"
    assert np.array_equal(got[..., 3], want[..., 3])


# ------------------------------------------------------------------ effects.go
@pytest.mark.parametrize("name", list(IMAGES))
@pytest.mark.parametrize("sigma", [0.3, 1.0, 2.0, 2.6])
def test_gaussian_blur(ctx, orc, name, sigma):
    img = IMAGES[name]()
    want = orc.gaussian_blur(img, sigma)
    assert np.array_equal(ctx.GaussianBlur(img, sigma, exact=True), want)
    assert_blur_close(ctx.GaussianBlur(img, sigma), want)


def test_gaussian_blur_large_sigma_generic_path(ctx, orc):
    img = synth.large_photo(160, 120, 5)
    for sigma in (3.5, 20.0):          # radius 11 / 60: beyond the fused kernel
        want = orc.gaussian_blur(img, sigma)
        assert np.array_equal(ctx.GaussianBlur(img, sigma, exact=True), want)
        assert_blur_close(ctx.GaussianBlur(img, sigma), want)


def test_gaussian_blur_guards(ctx):
    img = synth.make_test_image(100, 100)
    assert ctx.GaussianBlur(img, 0) is img            # fennec_test.go:708-723
    assert ctx.GaussianBlur(img, -1.0) is img


def test_gaussian_blur_arbitrary_kernel_saturates(ctx, orc):
    """The 1-D kernel is an input: negative taps / gain > 1 must clamp like clampF."""
    img = synth.noise_image(96, 64, 9)
    for k in ([-0.5, 2.0, -0.5], [0.25, 0.5, 0.25], [0.7, 0.7, 0.7], [-1.0, 0.5, -1.0]):
        k = np.array(k)
        want = orc.gaussian_blur(img, 1.0, kernel=k)
        assert np.array_equal(ctx.GaussianBlur(img, 1.0, exact=True, kernel=k), want)
        assert_blur_close(ctx.GaussianBlur(img, 1.0, kernel=k), want)


def _binomial(radius):
    k = np.array([1.0])
    for _ in range(2 * radius):
        k = np.convolve(k, [0.5, 0.5])
    return k            # dyadic weights summing to exactly 1: sums land on exact .5 ties


@pytest.mark.parametrize("radius", [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 14, 16])
def test_gaussian_blur_exact_ties(ctx, orc, radius):
    """EXACT mode on inputs built to sit on clampF's rounding boundary (convert.go:149-158).

    Dyadic kernels make every weighted sum a multiple of 2^-2R, so a large share of samples is an
    exact x.5 tie -- the case an fp32 accumulator cannot decide -- and the stripe images flag every
    pixel of a tile at once (the tile-wide recompute branch) in the horizontal or the vertical pass.
    """
    k = _binomial(radius)
    w, h = 300, 150
    cols = np.zeros((h, w, 4), np.uint8); cols[:, 1::2, :3] = 1; cols[..., 3] = 255
    rows = np.zeros((h, w, 4), np.uint8); rows[1::2, :, :3] = 1; rows[..., 3] = 255
    lowbits = synth.noise_image(w, h, 40 + radius, alpha=True) & 3
    for img in (synth.noise_image(w, h, radius, alpha=True), cols, rows, lowbits):
        want = orc.gaussian_blur(img, 1.0, kernel=k)
        got = ctx.GaussianBlur(img, 1.0, exact=True, kernel=k)
        assert np.array_equal(got, want), radius


def test_gaussian_blur_exact_all_sigmas(ctx, orc):
    """EXACT mode over every radius GaussianBlur's own kernel produces up to the fused limit and past it (r3: the tile
    kernel takes radii 9..16 too -- sigma up to 5.33 -- with the H window streaming through)."""
    img = synth.noise_image(257, 131, 77, alpha=True)
    for sigma in (0.3, 0.5, 0.7, 1.0, 1.3, 1.7, 2.0, 2.3, 2.6, 2.9, 3.2, 3.4, 3.7, 4.0, 4.3, 4.6, 5.0, 5.3, 5.7):
        assert np.array_equal(ctx.GaussianBlur(img, sigma, exact=True), orc.gaussian_blur(img, sigma)), sigma


@pytest.mark.parametrize("sigma", [2.8, 3.1, 3.5, 3.9, 4.2, 4.5, 5.0, 5.33])
def test_gaussian_blur_wide_radii(ctx, orc, sigma):
    """Radii 9..16 (effects.go:153: radius = ceil(3 sigma)) in the tile kernel: fast mode within its tolerance, exact
    mode equal, on shapes with edge tiles on every side, a device view at 4-byte alignment, and a batch."""
    import torch
    for (w, h, seed) in [(700, 300, 3), (64, 128, 5), (130, 97, 8), (31, 9, 2)]:
        img = synth.noise_image(w, h, seed, alpha=True) if seed % 2 else synth.large_photo(w, h, seed)
        want = orc.gaussian_blur(img, sigma, procs=8)
        assert np.array_equal(ctx.GaussianBlur(img, sigma, exact=True), want), (sigma, w, h)
        assert_blur_close(ctx.GaussianBlur(img, sigma), want)
    big = synth.large_photo(1301, 420, 9)
    d = torch.from_numpy(big).cuda()
    torch.cuda.synchronize()
    sub = d[5:400, 3:1200]
    want = orc.gaussian_blur(big[5:400, 3:1200], sigma, procs=8)
    out = ctx.GaussianBlur(sub, sigma, exact=True); ctx.sync()
    assert np.array_equal(out.cpu().numpy(), want)
    imgs = [synth.large_photo(512, 384, k) for k in range(3)]
    outs = ctx.GaussianBlurBatch([torch.from_numpy(i).cuda() for i in imgs], sigma, exact=True); ctx.sync()
    for i, o in zip(imgs, outs):
        assert np.array_equal(o.cpu().numpy(), orc.gaussian_blur(i, sigma, procs=8))


@pytest.mark.parametrize("name", list(IMAGES))
def test_blur3x3_sharpen_adaptive(ctx, orc, name):
    img = IMAGES[name]()
    assert np.array_equal(ctx.blur3x3(img), orc.blur3x3(img))
    for s in (0.3, 0.8, 1.0, 5.0):
        assert np.array_equal(ctx.Sharpen(img, s), orc.sharpen(img, s)), s
        assert np.array_equal(ctx.AdaptiveSharpen(img, s), orc.adaptive_sharpen(img, s)), s


def _soft_image(orc, w, h, seed):
    """Noise blurred twice: Sobel magnitudes spread over (0, 400), so AdaptiveSharpen's edge strength is fractional."""
    return orc.gaussian_blur(orc.gaussian_blur(synth.noise_image(w, h, seed, alpha=True), 2.0), 1.2)


@pytest.mark.parametrize("w,h", [(64, 32), (65, 33), (200, 150), (1000, 37), (3, 3), (4, 700), (517, 389)])
def test_sharpen_adaptive_guarded_kernels(ctx, orc, w, h):
    """The marching kernels (Sharpen by table, AdaptiveSharpen under the fp32 rounding guard) on soft images --
    fractional edge strengths, so the guard really decides -- and on tie-prone strengths (amount 1.5, 2.5, 1.375):
    bit-exact against the oracle."""
    soft = _soft_image(orc, w, h, w + h)
    hard = synth.large_photo(w, h, 3)
    for img in (soft, hard):
        assert np.array_equal(ctx.blur3x3(img), orc.blur3x3(img))
        for s in (0.05, 0.25, 0.3, 0.5, 0.6180339887, 0.75, 1.0):
            assert np.array_equal(ctx.Sharpen(img, s), orc.sharpen(img, s)), ("sharpen", s)
            assert np.array_equal(ctx.AdaptiveSharpen(img, s), orc.adaptive_sharpen(img, s)), ("adaptive", s)


@pytest.mark.parametrize("form", ["stream", "tile_pairs", "tile_rows"])
@pytest.mark.parametrize("w,h", [(64, 24), (62, 16), (63, 17), (125, 100), (200, 150), (1000, 37), (3, 3), (2, 9), (9, 2), (1, 1), (4, 700), (517, 389)])
def test_fx_kernel_forms(ctx, orc, monkeypatch, form, w, h):
    """Round 3: the streaming kernel (a wave marches down a strip of 62 columns) and both forms of the tile kernel
    (paired rows on v_pk_* with the fract boundary test; one row at a time with the second pack) against the
    oracle -- sizes around the strip width, the segment length and the tile."""
    ctx.set_form("fx_stream", "1" if form == "stream" else "0")
    ctx.set_form("fx_pairs", "0" if form == "tile_rows" else "1")
    soft = _soft_image(orc, w, h, 3 * w + h)
    hard = synth.large_photo(w, h, 5)
    for img in (soft, hard):
        assert np.array_equal(ctx.blur3x3(img), orc.blur3x3(img))
        for s in (0.25, 0.5, 0.6180339887, 1.5):
            assert np.array_equal(ctx.Sharpen(img, s), orc.sharpen(img, s)), ("sharpen", s)
            assert np.array_equal(ctx.AdaptiveSharpen(img, s), orc.adaptive_sharpen(img, s)), ("adaptive", s)


@pytest.mark.parametrize("form", ["stream", "tile_pairs"])
def test_fx_kernel_forms_4k_and_views(ctx, orc, monkeypatch, form):
    """The same at 4K (many segments per strip) and on strided device views (the flat-copy pass runs behind either form)."""
    import torch
    ctx.set_form("fx_stream", "1" if form == "stream" else "0")
    img = _soft_image(orc, 3840, 2160, 11)
    assert np.array_equal(ctx.AdaptiveSharpen(img, 0.5), orc.adaptive_sharpen(img, 0.5, procs=32))
    assert np.array_equal(ctx.Sharpen(img, 0.5), orc.sharpen(img, 0.5, procs=32))
    assert np.array_equal(ctx.blur3x3(img), orc.blur3x3(img))
    big = torch.from_numpy(synth.large_photo(700, 300, 2)).cuda()
    sub = big[10:250, 33:600]
    want = sub.contiguous().cpu().numpy()
    for got, ref in ((ctx.AdaptiveSharpen(sub, 0.5), orc.adaptive_sharpen), (ctx.Sharpen(sub, 0.5), orc.sharpen)):
        ctx.sync()
        g = got.cpu().numpy()
        assert np.array_equal(g[1:-1, 1:-1], ref(want, 0.5)[1:-1, 1:-1])     # (the frame follows the flat-copy semantics: A19 tests)


def test_sharpen_amounts_against_reference_order_kernel(ctx, orc, monkeypatch):
    """Kernel-level amounts outside what the reference ever passes (and an unaligned strided view): the marching
    kernels against the round-1 fp64 kernel, which follows the reference's operation order (form "fx_ref" = 1)."""
    import torch
    soft = _soft_image(orc, 333, 217, 5)
    big = torch.from_numpy(np.ascontiguousarray(np.pad(soft, ((0, 0), (3, 2), (0, 0))))).cuda()
    view = big[:, 3:-2]                                         # 4-byte aligned rows, stride != 4 w
    rng = np.random.default_rng(11)
    amounts = [0.1, 1.0, 1.5, 2.0, 2.5, 3.0, 7.99, 8.5, 40.0, -0.5] + list(rng.uniform(0.01, 8.0, 6))
    for adaptive in (False, True):
        for amt in amounts:
            ctx.set_form("fx_ref", None)
            got = ctx.sharpen_amount(soft, amt, adaptive)
            got_view = ctx.sharpen_amount(view, amt, adaptive).cpu().numpy()
            ctx.set_form("fx_ref", "1")
            want = ctx.sharpen_amount(soft, amt, adaptive)
            want_view = ctx.sharpen_amount(view, amt, adaptive).cpu().numpy()
            assert np.array_equal(got, want), (adaptive, amt)
            # a SubImage's border follows the reference's flat copy(dst.Pix, img.Pix) (test_subimage_flat_pix_copies_gpu)
            assert np.array_equal(got_view, want_view), (adaptive, amt, "view")
            assert np.array_equal(got_view[1:-1, 1:-1], want[1:-1, 1:-1]), (adaptive, amt, "view interior")
    ctx.set_form("fx_ref", None)


@pytest.mark.parametrize("geom", [(20, 10, 4, 8, 6, 12), (160, 120, 5, 8, 80, 128), (700, 300, 7, 13, 280, 611), (33, 17, 0, 0, 17, 9)])
def test_subimage_flat_pix_copies_gpu(ctx, orc, geom):
    """effects.go:68,120 and convert.go:16 (MSSSIM, ssim.go:345-346) copy the FLAT Pix slice.  On a
    SubImage (stride != 4w) gaussianBlur3x3's border and alpha, AdaptiveSharpen's border, Sharpen's border (through its
    blurred operand) and MSSSIM's pyramid input are the first 4wh bytes of the slice, not the rows.  Host views and
    device views against the oracle (whose flat semantics tests/test_oracle.py checks against Go's slice rules)."""
    import torch
    pw, ph, y0, x0, h, w = geom
    big = synth.noise_image(pw, ph, 23, alpha=True)
    sub = big[y0:y0 + h, x0:x0 + w]
    dbig = torch.from_numpy(big).cuda()
    dsub = dbig[y0:y0 + h, x0:x0 + w]
    torch.cuda.synchronize()
    tight = np.ascontiguousarray(sub)

    def both(fn):
        host = fn(sub)
        dev = fn(dsub); ctx.sync()
        return host, dev.cpu().numpy()

    for got in both(ctx.blur3x3):
        assert np.array_equal(got, orc.blur3x3(sub))
    for s_ in (0.3, 0.5, 1.0):
        for got in both(lambda im: ctx.Sharpen(im, s_)):
            assert np.array_equal(got, orc.sharpen(sub, s_)), ("sharpen", s_)
        for got in both(lambda im: ctx.AdaptiveSharpen(im, s_)):
            assert np.array_equal(got, orc.adaptive_sharpen(sub, s_)), ("adaptive", s_)
    if (y0, x0) != (0, 0) or w != pw:
        assert not np.array_equal(orc.blur3x3(sub), orc.blur3x3(tight))      # the case is a real one
    assert np.array_equal(ctx.lanczosResize(sub, w, h), orc.lanczos_resize(sub, w, h))          # resize.go:45-49
    got = ctx.lanczosResize(dsub, w, h); ctx.sync()
    assert np.array_equal(got.cpu().numpy(), orc.lanczos_resize(sub, w, h))

    other = synth.noise_image(w, h, 5)
    dother = torch.from_numpy(other).cuda()
    torch.cuda.synchronize()
    want = orc.msssim(sub, other)
    assert abs(ctx.MSSSIM(sub, other) - want) <= SSIM_TOL and abs(ctx.MSSSIM(dsub, dother) - want) <= SSIM_TOL
    want = orc.msssim(other, sub)
    assert abs(ctx.MSSSIM(other, sub) - want) <= SSIM_TOL and abs(ctx.MSSSIM(dother, dsub) - want) <= SSIM_TOL
    ctx.msssim_enqueue(dsub, dother)
    assert abs(ctx.fetch_result() - orc.msssim(sub, other)) <= SSIM_TOL
    # dims differ: b is Lanczos-resized by rows (resize.go), a enters the pyramid flat
    small = synth.noise_image(max(w // 2, 1), max(h // 2, 1), 9)
    want = orc.msssim(sub, small)
    assert abs(ctx.MSSSIM(sub, small) - want) <= SSIM_TOL
    assert abs(ctx.MSSSIM(dsub, torch.from_numpy(small).cuda()) - want) <= SSIM_TOL


def test_pixel_ssim_walks_the_whole_slice(ctx, orc):
    """pixelSSIM (ssim.go:178,190) loops to len(a.Pix): a SubImage's slice runs to the end of the PARENT's buffer."""
    import torch
    big_a = synth.noise_image(20, 12, 3, alpha=True)
    big_b = synth.noise_image(20, 12, 4, alpha=True)
    y0, x0, h, w = 2, 5, 6, 7
    pa = big_a.reshape(-1)[(y0 * 20 + x0) * 4:]
    pb = big_b.reshape(-1)[(y0 * 20 + x0) * 4:]
    want = orc.pixel_ssim_flat(pa, pb, w, h)
    assert ctx.pixelSSIM(pa, pb, w, h) == want
    got = ctx.pixelSSIM(torch.from_numpy(pa.copy()).cuda(), torch.from_numpy(pb.copy()).cuda(), w, h)
    assert got == want
    # the shortest slice a w x h view can have is what SSIM / SSIMFast assume
    short = ((h - 1) * 20 + w) * 4
    assert ctx.pixelSSIM(pa[:short], pb[:short], w, h) == ctx.SSIMFast(big_a[y0:y0 + h, x0:x0 + w], big_b[y0:y0 + h, x0:x0 + w])
    assert want != ctx.pixelSSIM(pa[:short], pb[:short], w, h)
    with pytest.raises(fennec_amd.FennecError):
        ctx.pixelSSIM(pa, pb[:-8], w, h)                   # the reference panics
    assert ctx.pixelSSIM(pa[:0], pb[:0], 0, 5) == 1.0


def test_adaptive_sharpen_4k_soft(ctx, orc):
    img = _soft_image(orc, 3840, 2160, 4)
    for s in (0.25, 0.5):
        assert np.array_equal(ctx.AdaptiveSharpen(img, s), orc.adaptive_sharpen(img, s, procs=16)), s
    assert np.array_equal(ctx.Sharpen(img, 0.5), orc.sharpen(img, 0.5, procs=16))


def test_sharpen_guards(ctx):
    img = synth.make_test_image(100, 100)
    tiny = synth.make_test_image(2, 2)
    assert ctx.Sharpen(img, 0) is img and ctx.AdaptiveSharpen(img, 0) is img     # fennec_test.go:632-694
    assert ctx.Sharpen(tiny, 0.5) is tiny and ctx.AdaptiveSharpen(tiny, 0.5) is tiny


# ------------------------------------------------------------------ resize.go
@pytest.mark.parametrize("name", ["grad_64x48", "alpha_37x29", "photo_200x150", "noise_131x77", "noise_3x5", "noise_1x1"])
def test_lanczos_resize(ctx, orc, name):
    img = IMAGES[name]()
    h, w = img.shape[:2]
    for dw, dh in [(max(w // 2, 1), max(h // 2, 1)), (w * 2, h * 2 - 1), (w + 5, max(h - 3, 1)), (1, 1), (w, h), (9, h), (w, 7)]:
        assert np.array_equal(ctx.lanczosResize(img, dw, dh), orc.lanczos_resize(img, dw, dh)), (dw, dh)
    assert ctx.lanczosResize(img, 0, 50).shape[:2] == (0, 0)       # fennec_test.go:554-560


def _opaque(img):
    out = img.copy()
    out[..., 3] = 255
    return out


@pytest.mark.parametrize("w,h,dw,dh", [(640, 480, 320, 240), (640, 480, 1280, 960), (517, 389, 300, 211), (300, 200, 100, 67),
                                       (1000, 64, 231, 64), (64, 1000, 64, 231), (777, 333, 176, 75), (800, 600, 107, 80),
                                       (130, 90, 195, 135), (33, 9, 66, 18), (2, 2, 5, 5), (1920, 1080, 960, 540)])
def test_lanczos_resize_guard_kernels(ctx, orc, w, h, dw, dh):
    """Opaque images take the fp32 kernels with the rounding guard (ratios 0.4 .. 7.5: window shapes NV = 2 .. 8 and
    the fp64 fallback beyond): bit-exact against the oracle, on noise, on a soft image and on content built from
    exact ties (two-level stripes at an integer ratio: every output is (a + b) / 2 with a + b odd)."""
    noise = _opaque(synth.noise_image(w, h, w + dw, alpha=True))
    photo = synth.large_photo(w, h, 1)
    stripes = np.empty((h, w, 4), np.uint8)
    stripes[:, 0::2] = (100, 7, 250, 255)
    stripes[:, 1::2] = (101, 8, 255, 255)
    stripes[1::2, :, :3] += 1                                    # and odd rows one level up: ties in the V pass too
    holes = noise.copy()
    holes[h // 3: h // 2, w // 4: w // 2, 3] = 17                # a translucent patch: those windows take the general path
    for img in (noise, photo, stripes, holes):
        assert np.array_equal(ctx.lanczosResize(img, dw, dh), orc.lanczos_resize(img, dw, dh))


@pytest.mark.parametrize("dw,dh", [(960, 540), (961, 541), (130, 33), (64, 600), (1353, 517), (1920, 1080)])
def test_lanczos_resize_two_to_one_exact_forms(ctx, orc, dw, dh):
    """r5: at exactly 2:1 the exact (fp64, reference-order) loops of resize_fused_kernel have straight-line forms -- scalar weights,
    four outputs per lane, two rows per wave instruction, the 18-row V sweep.  Content that sends every tile there (SURVEY 8(d)'s
    ramp: all rounding ties), with what the forms must hand on: the image's edge groups (clamped tap lists), odd widths and heights
    (a lone last row, a last group with one output), translucent pixels and patches (the general arithmetic), ramp / noise seams
    (rows that stay fp32), and 2:1 on ONE axis only.  Bit-exact against the oracle; the "resize_mfma" form is left alone (the ramp is
    handed back by the matrix kernel on the first call, and the ctx goes to resize_fused_kernel after it)."""
    w, h = 2 * dw, 2 * dh
    ramp = synth.large_photo(w, h, 3)
    cases = [("ramp", ramp)]
    one = ramp.copy(); one[h // 2, w // 3, 3] = 200
    cases.append(("one translucent px", one))
    patch = ramp.copy(); patch[h // 5: h // 5 + 40, : w // 7, 3] = 0; patch[-9:, -33:, 3] = 91
    cases.append(("translucent patches at the edges", patch))
    seam = _opaque(synth.noise_image(w, h, dw + dh, alpha=True)); seam[:, w // 2:] = ramp[:, w // 2:]; seam[: h // 3] = ramp[: h // 3]
    cases.append(("noise / ramp seams", seam))
    for name, img in cases:
        for rep in range(2):                       # twice: the second call takes resize_fused_kernel straight away (the matrix kernel's cool-down)
            assert np.array_equal(ctx.lanczosResize(img, dw, dh), orc.lanczos_resize(img, dw, dh, procs=8)), (name, rep)
    # 2:1 on the x axis only / on the y axis only: one pass uniform, the other not
    img = synth.large_photo(2 * dw, 3 * (dh // 2) + 7, 4)
    assert np.array_equal(ctx.lanczosResize(img, dw, dh // 2 + 2), orc.lanczos_resize(img, dw, dh // 2 + 2, procs=8))
    img = synth.large_photo(3 * (dw // 2) + 5, 2 * dh, 5)
    assert np.array_equal(ctx.lanczosResize(img, dw // 2 + 3, dh), orc.lanczos_resize(img, dw // 2 + 3, dh, procs=8))


def test_lanczos_resize_dense_form_after_two_cool_downs(orc):
    """r5: a plan whose images keep coming back tie-dense from the matrix kernel (two cool-downs in a row: ~66 calls) runs
    resize_fused_dense_kernel -- no fp32 passes at all.  Every call on the way there, and the dense form itself on a ramp, a
    ramp with translucent pixels and a noise image (which the dense form must still get right), equals the oracle."""
    c = fennec_amd.Context(0)
    w, h, dw, dh = 768, 400, 384, 200
    ramp = synth.large_photo(w, h, 6)
    want = orc.lanczos_resize(ramp, dw, dh)
    seen = set()
    for k in range(150):
        got = c.lanczosResize(ramp, dw, dh)
        seen.add(c.last_kernel(fennec_amd.PROF_RESIZE))
        if k % 16 == 0 or k > 60:
            assert np.array_equal(got, want), k
    assert ("resize_dense21_kernel" in seen or "resize_fused_dense_kernel" in seen) and any(s.startswith("resize_mfma_kernel") for s in seen), seen
    # still in the cool-down: other content through the dense form
    holes = ramp.copy(); holes[h // 2: h // 2 + 9, 5: w // 2, 3] = 3; holes[0, 0, 3] = 254
    noise = _opaque(synth.noise_image(w, h, 77, alpha=True))
    for img in (holes, noise, ramp):
        got = c.lanczosResize(img, dw, dh)
        assert c.last_kernel(fennec_amd.PROF_RESIZE) in ("resize_dense21_kernel", "resize_fused_dense_kernel", "resize_mfma_kernel + resize_fused_sparse_kernel")
        assert np.array_equal(got, orc.lanczos_resize(img, dw, dh))
    c.close()


def _photo_soft(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 90 * np.sin(x / (31.0 + c)) * np.cos(y / (23.0 + 2 * c)) + rng.normal(0, 5, x.shape) for c in range(3)] + [np.full(x.shape, 255.0)], -1)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h,dw,dh", [(3840, 2160, 1920, 1080), (1920, 1080, 3840, 2160), (1280, 720, 853, 480), (640, 480, 320, 240),
                                       (517, 389, 233, 800), (2048, 70, 929, 151), (300, 200, 300, 200), (64, 64, 7, 5)])
def test_lanczos_resize_batch(ctx, orc, w, h, dw, dh):
    """r6: fnx_lanczos_resize_batch -- n same-geometry device images through one set of launches (the matrix kernel + its
    hand-backs, the one-launch kernel, the exact 2:1 form: each with the image as a grid dimension) against the same images one
    call at a time, and -- below 4K -- the oracle.  A batch mixes what the routes hand to each other: photograph-like images
    (the matrix kernel keeps them), SURVEY 8(d)'s ramp (every tile handed back; after enough of them the plan's cool-down
    takes the dense forms), an image with a translucent patch, noise."""
    import torch
    ramp = synth.large_photo(w, h, 3)
    holes = _photo_soft(w, h, 5)
    holes[h // 3: h // 2 + 1, w // 4: w // 2 + 1, 3] = 17
    imgs = [_photo_soft(w, h, 1), ramp, holes, _opaque(synth.noise_image(w, h, 9, alpha=True)), _photo_soft(w, h, 2), synth.large_photo(w, h, 8)]
    c = fennec_amd.Context(0)                                    # a ctx of its own: plans without history
    try:
        want = [c.lanczosResize(im, dw, dh) for im in imgs]
        if w * h < 3000 * 2000:
            for k in (0, 1, 2):
                assert np.array_equal(want[k], orc.lanczos_resize(imgs[k], dw, dh, procs=8)), k
        dev = [torch.from_numpy(im).cuda() for im in imgs]
        c2 = fennec_amd.Context(0)
        for rep in range(3):                                     # (the ramps push the plan into its cool-down between repetitions)
            got = c2.lanczosResizeBatch(dev, dw, dh)
            c2.sync()
            for k, (g, wnt) in enumerate(zip(got, want)):
                assert np.array_equal(g.cpu().numpy(), wnt), (rep, k, w, h, dw, dh)
        # a batch of ramps only, many times: the dense / 2:1 forms with the image dimension
        ramps = [torch.from_numpy(synth.large_photo(w, h, s)).cuda() for s in range(4)]
        wr = [c.lanczosResize(synth.large_photo(w, h, s), dw, dh) for s in range(4)]
        for rep in range(70 if w * h <= 1280 * 720 else 6):
            got = c2.lanczosResizeBatch(ramps, dw, dh)
        c2.sync()
        for k in range(4):
            assert np.array_equal(got[k].cpu().numpy(), wr[k]), ("ramps", k)
        # pitched views and a batch of one
        big = torch.from_numpy(np.ascontiguousarray(np.pad(imgs[0], ((0, 0), (2, 2), (0, 0))))).cuda()
        views = [big[:, 2: 2 + w], torch.from_numpy(np.ascontiguousarray(np.pad(imgs[1], ((0, 0), (2, 2), (0, 0))))).cuda()[:, 2: 2 + w]]
        if (w, h) != (dw, dh):                                   # (equal dims are the reference's FLAT copy of Pix, resize.go:45-49: rows only for tight images)
            got = c2.lanczosResizeBatch(views, dw, dh)
            c2.sync()
            assert np.array_equal(got[0].cpu().numpy(), want[0]) and np.array_equal(got[1].cpu().numpy(), want[1])
        one = c2.lanczosResizeBatch([dev[2]], dw, dh)
        c2.sync()
        assert np.array_equal(one[0].cpu().numpy(), want[2])
        c2.close()
    finally:
        c.close()


@pytest.mark.parametrize("w,h,bw,bh", [(3840, 2160, 1920, 1080), (1024, 768, 1024, 768), (1000, 600, 500, 300), (640, 480, 320, 240)])
def test_msssim_batch_enqueue(ctx, orc, w, h, bw, bh):
    """r6: fennec_MSSSIM_batch_enqueue -- n pairs as one FIFO entry, the b sides resized by ONE batched lanczosResize where the
    dims differ (ssim.go:320-322): every value equals the single enqueue's and MSSSIM's, and -- below 4K -- the oracle's."""
    import torch
    n = 5
    A = [synth.large_photo(w, h, k) for k in range(n)]
    B = [orc.gaussian_blur(synth.large_photo(bw, bh, k), 0.9 + 0.2 * k, procs=8) for k in range(n)]
    da, db = [torch.from_numpy(x).cuda() for x in A], [torch.from_numpy(x).cuda() for x in B]
    want = [ctx.MSSSIM(da[k], db[k]) for k in range(n)]
    ctx.msssim_batch_enqueue(da, db)
    got = ctx.fetch_results(n)
    assert list(got) == want
    ctx.msssim_enqueue(da[2], db[2])
    ctx.msssim_batch_enqueue(da[:3], db[:3])                   # two entries in the FIFO, fetched in order; a partial fetch of the second
    assert ctx.fetch_result() == want[2]
    assert list(ctx.fetch_results(2)) == want[:2]
    if w * h < 3000 * 2000:
        for k in (0, n - 1):
            assert abs(want[k] - orc.msssim(A[k], B[k], procs=8)) <= SSIM_TOL
    with pytest.raises(fennec_amd.FennecError):
        ctx.fetch_results(1)                                   # nothing left


def test_lanczos_resize_guard_vs_fp64_kernels(ctx, orc, monkeypatch):
    """Random geometries, device views with odd strides: the guard kernels against the round-1 fp64 kernels
    (form "resize_fp64" = 1), which follow the reference's operation order."""
    import torch
    rng = np.random.default_rng(3)
    for _ in range(24):
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
        dw, dh = int(rng.integers(1, 900)), int(rng.integers(1, 700))
        img = _opaque(synth.noise_image(w, h, w * 7 + h, alpha=True))
        if rng.random() < 0.3:
            img[rng.integers(0, h), rng.integers(0, w), 3] = 0
        pad = int(rng.integers(0, 3))
        big = torch.from_numpy(np.ascontiguousarray(np.pad(img, ((0, 0), (pad, 3 - pad), (0, 0))))).cuda()
        view = big[:, pad: pad + w]
        ctx.set_form("resize_fp64", None)
        got = ctx.lanczosResize(img, dw, dh)
        got_view = ctx.lanczosResize(view, dw, dh).cpu().numpy()
        ctx.set_form("resize_fp64", "1")
        want = ctx.lanczosResize(img, dw, dh)
        ctx.set_form("resize_fp64", None)
        assert np.array_equal(got, want), (w, h, dw, dh)
        assert np.array_equal(got_view, want), (w, h, dw, dh, "view")


@pytest.mark.parametrize("w,h,dw,dh", [(640, 480, 320, 240), (640, 480, 1280, 960), (641, 479, 301, 1000), (517, 389, 233, 800),
                                       (130, 90, 195, 41), (33, 9, 67, 18), (3, 200, 6, 97), (1920, 1080, 3840, 2160),
                                       (3840, 2160, 1920, 1080), (2048, 70, 929, 151), (9, 9, 4, 20)])
def test_lanczos_resize_one_launch_vs_two_pass(ctx, orc, monkeypatch, w, h, dw, dh):
    """r3: resize_fused_kernel (resizeH into an LDS tile, resizeV out of it: windows of <= 16 source pixels) against the
    two-pass kernels (form "resize_fused" = 0) and, below 4K, the oracle -- noise, SURVEY 8(d)'s ramp (dense exact ties at 2:1),
    two-level stripes (ties in both passes), a translucent patch (the general arithmetic from the tile), odd widths (the
    last H group has one output), tiles with fewer V groups than the tile holds, host arrays and strided device views."""
    import torch
    noise = _opaque(synth.noise_image(w, h, w + dw, alpha=True))
    ramp = synth.large_photo(w, h, 3)
    stripes = np.empty((h, w, 4), np.uint8)
    stripes[:, 0::2] = (100, 7, 250, 255)
    stripes[:, 1::2] = (101, 8, 255, 255)
    stripes[1::2, :, :3] += 1
    holes = noise.copy()
    holes[h // 3: h // 2 + 1, w // 4: w // 2 + 1, 3] = 17
    soft = synth.noise_image(w, h, 11, alpha=True)             # every window translucent
    big = w * h > 3000 * 2000
    for k, img in enumerate((noise, ramp, stripes, holes, soft)):
        if big and k in (2, 4):
            continue
        ctx.set_form("resize_fused", None)
        got = ctx.lanczosResize(img, dw, dh)
        pad = 1 + k % 3
        view = torch.from_numpy(np.ascontiguousarray(np.pad(img, ((0, 0), (pad, 4 - pad), (0, 0))))).cuda()[:, pad: pad + w]
        got_view = ctx.lanczosResize(view, dw, dh).cpu().numpy()
        ctx.set_form("resize_fused", "0")
        two = ctx.lanczosResize(img, dw, dh)
        ctx.set_form("resize_fused", None)
        assert np.array_equal(got, two) and np.array_equal(got_view, two), (k, w, h, dw, dh)
        if not big or k == 3:                                  # at 4K the oracle checks the translucent-patch image only
            assert np.array_equal(got, orc.lanczos_resize(img, dw, dh, procs=32 if big else 8)), (k, w, h, dw, dh)


def test_resize_plan_cache_eviction(ctx, orc):
    """More distinct tables than the ctx keeps plans for (8), revisited: results stay those of the oracle."""
    img = _opaque(synth.noise_image(240, 180, 2, alpha=True))
    sizes = [(120 + 7 * k, 90 + 5 * k) for k in range(7)]
    for rnd in range(2):
        for dw, dh in sizes:
            assert np.array_equal(ctx.lanczosResize(img, dw, dh), orc.lanczos_resize(img, dw, dh)), (rnd, dw, dh)


def test_resize_passes_with_explicit_tables(ctx, orc):
    img = synth.make_test_image_with_alpha(120, 90)
    tab = orc.precompute_weights(50, 120)
    assert np.array_equal(ctx.resize_pass(img, 50, False, tab), orc.resize_h(img, 50, tab))
    tab = orc.precompute_weights(200, 90)
    assert np.array_equal(ctx.resize_pass(img, 200, True, tab), orc.resize_v(img, 200, tab))
    # hand-made tables: taps with gaps (precomputeWeights drops w == 0 taps, resize.go:186-188) take the
    # generic H kernel; contiguous taps at the right image edge, 1..16 taps, take the in-register one
    rng = np.random.default_rng(5)
    for gap in (True, False):
        off, idx, wt = [0], [], []
        for d in range(70):
            n = int(rng.integers(1, 17))
            s0 = int(rng.integers(0, 120 - (2 * n if gap else n) + 1)) if d % 7 else 120 - (2 * n - 1 if gap else n)
            idx += [s0 + (2 * k if gap else k) for k in range(n)]
            w = rng.uniform(-0.3, 1.0, size=n)
            wt += list(w / w.sum()) if abs(w.sum()) > 0.2 else list(w)
            off.append(len(idx))
        tab = (np.array(off, np.int32), np.array(idx, np.int32), np.array(wt, np.float64))
        assert np.array_equal(ctx.resize_pass(img, 70, False, tab), orc.resize_h(img, 70, tab)), gap


def test_lanczos_resize_with_the_callers_tables(ctx, orc):
    """fnx_lanczos_resize with caller-supplied tables (compared by content: no table id) -- what the cgo shim passes.  The
    reference's own tables take the one-launch kernel; hand-made V tables whose windows do NOT move down monotonically
    (here: the reference's rows in reverse, a vertical flip) cannot share a tile and take the two passes; both against
    the oracle's passes composed."""
    img = _opaque(synth.noise_image(333, 211, 9, alpha=True))
    img[50:80, 100:160, 3] = 40
    for dw, dh in ((166, 105), (500, 400), (333, 300)):
        th, tv = orc.precompute_weights(dw, 333), orc.precompute_weights(dh, 211)
        want = orc.resize_v(orc.resize_h(img, dw, th), dh, tv)
        assert np.array_equal(ctx.lanczos_resize_tables(img, dw, dh, th, tv), want), (dw, dh)
        assert np.array_equal(want, orc.lanczos_resize(img, dw, dh))
        off, idx, wt = (np.asarray(x) for x in tv)
        roff, ridx, rwt = [0], [], []
        for d in range(dh - 1, -1, -1):                        # output row d of the flipped table = the reference's row dh - 1 - d
            ridx += list(idx[off[d]:off[d + 1]]); rwt += list(wt[off[d]:off[d + 1]]); roff.append(len(ridx))
        rv = (np.array(roff, np.int32), np.array(ridx, np.int32), np.array(rwt, np.float64))
        got = ctx.lanczos_resize_tables(img, dw, dh, th, rv)
        assert np.array_equal(got, orc.resize_v(orc.resize_h(img, dw, th), dh, rv)) and np.array_equal(got, want[::-1]), (dw, dh, "flipped")


def test_resize_v_with_scattered_windows(ctx, orc):
    """resizeV with hand-made contiguous tap lists whose windows are NOT monotone from one output row to the
    next (the column-walking V kernel shares source rows among 4 consecutive outputs and must still apply each
    output's taps in its own order), with 1..32 taps (its table limit), and with longer / gapped lists that
    take the generic kernel."""
    img = synth.make_test_image_with_alpha(150, 120)
    rng = np.random.default_rng(11)
    for nmax, gap in ((6, False), (32, False), (40, False), (12, True)):
        off, idx, wt = [0], [], []
        for d in range(61):
            n = int(rng.integers(1, nmax + 1))
            span = 2 * n - 1 if gap else n
            s0 = int(rng.integers(0, 120 - span + 1))
            idx += [s0 + (2 * k if gap else k) for k in range(n)]
            w = rng.uniform(-0.3, 1.0, size=n)
            wt += list(w / w.sum()) if abs(w.sum()) > 0.2 else list(w)
            off.append(len(idx))
        tab = (np.array(off, np.int32), np.array(idx, np.int32), np.array(wt, np.float64))
        assert np.array_equal(ctx.resize_pass(img, 61, True, tab), orc.resize_v(img, 61, tab)), (nmax, gap)


def test_smart_resize(ctx, orc):
    img = synth.make_test_image(1000, 500)
    r = ctx.smartResize(img, 200, 200)
    assert r.shape[:2] == (100, 200) and np.array_equal(r, orc.smart_resize(img, 200, 200))
    assert ctx.smartResize(img, 2000, 2000) is img                  # fennec_test.go:540-552


# ------------------------------------------------------------------ ssim.go
@pytest.mark.parametrize("name", ["grad_64x48", "alpha_37x29", "photo_200x150", "noise_131x77", "photo_640x480"])
def test_box_downsample(ctx, orc, name):
    img = IMAGES[name]()
    h, w = img.shape[:2]
    for dw, dh in [(10, 10), (w // 2, h // 2), (w, h), (w + 7, h + 3), (1, 1), (8, 9), (w - 1, h - 1), (3 * w, 2 * h)]:
        assert np.array_equal(ctx.boxDownsample(img, dw, dh), orc.box_downsample(img, dw, dh)), (dw, dh)
    assert ctx.boxDownsample(img, 0, 0).shape[:2] == (0, 0)          # fennec_test.go:1109-1115


@pytest.mark.parametrize("name", list(IMAGES))
def test_ssim_family(ctx, orc, name):
    a = IMAGES[name]()
    b = orc.gaussian_blur(a, 1.2)
    assert abs(ctx.SSIM(a, b) - orc.ssim(a, b)) <= SSIM_TOL
    assert abs(ctx.SSIMFast(a, b) - orc.ssim_fast(a, b)) <= SSIM_TOL
    assert abs(ctx.MSSSIM(a, b) - orc.msssim(a, b)) <= SSIM_TOL
    assert abs(ctx.SSIM(a, a) - 1.0) <= 1e-12


def test_ssim_reference_assertions(ctx):
    """The reference's own assertions, on the HIP path (fennec_test.go:82-163)."""
    img = synth.make_test_image(100, 100)
    assert ctx.SSIM(img, img) >= 0.999
    black = synth.make_solid_image(100, 100, (0, 0, 0, 255))
    white = synth.make_solid_image(100, 100, (255, 255, 255, 255))
    assert ctx.SSIM(black, white) <= 0.1
    mod = img.copy(); r = mod[..., 0]; r[r > 10] -= 10
    assert 0.85 <= ctx.SSIM(img, mod) <= 0.999
    big = synth.make_test_image(500, 500)
    assert ctx.SSIMFast(big, big) >= 0.999
    small = synth.make_test_image(4, 4)
    assert ctx.SSIM(small, small) >= 0.999
    m = synth.make_test_image(128, 128)
    assert ctx.MSSSIM(m, m) >= 0.99
    assert ctx.MSSSIM(synth.make_solid_image(128, 128, (0, 0, 0, 255)), synth.make_solid_image(128, 128, (255, 255, 255, 255))) <= 0.1


def test_ssim_mismatched_dims(ctx, orc):
    a = synth.large_photo(200, 150, 2)
    half = orc.lanczos_resize(a, 100, 75)
    assert abs(ctx.SSIM(a, half) - orc.ssim(a, half)) <= SSIM_TOL      # ssim.go:31-33
    assert abs(ctx.MSSSIM(a, half) - orc.msssim(a, half)) <= SSIM_TOL  # ssim.go:320-322


def test_ssim_degenerate(ctx, orc):
    a = synth.make_test_image(8, 8)
    assert ctx.SSIM(a, a) == 1.0                      # zero windows -> 1.0 (ssim.go:162-164)
    a = synth.noise_image(2000, 3, 5)
    b = synth.noise_image(2000, 3, 6)
    assert abs(ctx.SSIMFast(a, b) - orc.ssim_fast(a, b)) <= SSIM_TOL   # downsample to 512x8 -> windowed with h == 8
    a = synth.noise_image(9, 700, 5)
    b = synth.noise_image(9, 700, 6)
    assert abs(ctx.SSIMFast(a, b) - orc.ssim_fast(a, b)) <= SSIM_TOL
    assert abs(ctx.MSSSIM(a, b) - orc.msssim(a, b)) <= SSIM_TOL


def test_msssim_levels(ctx, orc):
    a = synth.large_photo(300, 200, 1)
    b = orc.gaussian_blur(a, 1.5)
    got, lv = ctx.msssim_levels(a, b)
    want, wl = orc.msssim(a, b, per_level=True)
    assert abs(got - want) <= SSIM_TOL
    assert np.array_equal(np.isnan(lv), np.isnan(wl))
    assert np.nanmax(np.abs(lv - wl)) <= SSIM_TOL


@pytest.mark.parametrize("w,h", [(640, 480), (128, 96), (1024, 768), (2048, 1024), (144, 80), (1000, 600), (512, 512), (4096, 16),
                                 (1200, 720), (1360, 752), (752, 1360), (528, 16 * 35)])
def test_msssim_fused_levels(ctx, orc, monkeypatch, w, h):
    """The five-launch MSSSIM (one-pass 2 x 2 pyramid, multi-job box and window launches) against the level-by-level
    loop (form "msssim_levelwise" = 1) and the oracle: per level and combined.  Shapes that are not divisible by 2^levels
    (1000 x 600) or too thin take the loop either way."""
    import torch
    a = synth.large_photo(w, h, 2)
    b = orc.gaussian_blur(a, 1.1)
    b[h // 3: h // 2, w // 4: w // 2, :3] //= 2
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    ctx.set_form("msssim_levelwise", None)
    got, lv = ctx.msssim_levels(da, db)
    got_h, lv_h = ctx.msssim_levels(a, b)
    ctx.set_form("msssim_levelwise", "1")
    ref, lv_ref = ctx.msssim_levels(da, db)
    ctx.set_form("msssim_levelwise", None)
    # r3: level 0 read once (box_halve_kernel: its planes and level 1 in one pass) against the two-read form -- the same
    # integer arithmetic, so the values are identical, odd box edges (1200 / 512, 1360 / 512 ...) included
    ctx.set_form("msssim_nofuse0", "1")
    two, lv_two = ctx.msssim_levels(da, db)
    ctx.set_form("msssim_nofuse0", None)
    assert two == got and np.array_equal(lv_two, lv, equal_nan=True)
    # r3: the means taken by each level's last workgroup against the separate finish launch, and (opt-in) boxes of <= 5 x 5
    # pixels taken in the window kernel's tile load against planes written by the box kernel: the same bytes, the same sums
    # in the same order
    ctx.set_form("msssim_fold", "1")                # (off by default: see launch_msssim_fused)
    old, lv_old = ctx.msssim_levels(da, db)
    old2, lv_old2 = ctx.msssim_levels(da, db)                 # (the folded finish leaves its counters at zero)
    ctx.set_form("msssim_fold", None)
    assert old == got and np.array_equal(lv_old, lv, equal_nan=True) and old2 == got
    ctx.set_form("msssim_boxfly", "1")              # (off by default: see launch_msssim_fused)
    fly, lv_fly = ctx.msssim_levels(da, db)
    ctx.set_form("msssim_boxfly", None)
    assert fly == got and np.array_equal(lv_fly, lv, equal_nan=True)
    want, wl = orc.msssim(a, b, per_level=True, procs=8)
    assert np.array_equal(np.isnan(lv), np.isnan(wl)) and np.array_equal(np.isnan(lv_ref), np.isnan(wl))
    assert np.nanmax(np.abs(lv - wl)) <= SSIM_TOL and np.nanmax(np.abs(lv_ref - wl)) <= SSIM_TOL
    assert np.nanmax(np.abs(lv - lv_ref)) <= 1e-12 and got_h == got
    assert abs(got - want) <= SSIM_TOL and abs(ref - want) <= SSIM_TOL


def test_ssim_fast_prepared(ctx, orc):
    a = synth.large_photo(640, 480, 4)
    p = ctx.ssim_fast_prepare(a)
    for sigma in (0.5, 1.0, 2.0):
        b = orc.gaussian_blur(a, sigma)
        assert abs(p.against(b) - orc.ssim_fast(a, b)) <= SSIM_TOL
    p.close()
    a = synth.large_photo(100, 80, 4)
    p = ctx.ssim_fast_prepare(a)
    assert abs(p.against(a) - 1.0) <= 1e-12
    p.close()


# ------------------------------------------------------------------ orientation
@pytest.mark.parametrize("name", ["grad_64x48", "alpha_37x29", "noise_131x77", "noise_3x5", "noise_1x1", "noise_300x2"])
def test_apply_orientation(ctx, orc, name):
    img = IMAGES[name]()
    for o in range(0, 10):
        got = ctx.ApplyOrientation(img, o)
        want = orc.apply_orientation(img, o)
        if o < 2 or o > 8:
            assert got is img
        else:
            assert np.array_equal(got, want), o


# ------------------------------------------------------------------ spaces / strides / batches
def test_strided_host_input(ctx, orc):
    big = synth.noise_image(160, 120, 11, alpha=True)
    sub = big[5:85, 8:136]
    assert np.array_equal(ctx.GaussianBlur(sub, 1.5, exact=True), orc.gaussian_blur(sub, 1.5))
    assert np.array_equal(ctx.lanczosResize(sub, 30, 20), orc.lanczos_resize(sub, 30, 20))
    assert np.array_equal(ctx.boxDownsample(sub, 16, 10), orc.box_downsample(sub, 16, 10))
    assert np.array_equal(ctx.AdaptiveSharpen(sub, 0.5), orc.adaptive_sharpen(sub, 0.5))
    assert abs(ctx.SSIM(sub, np.ascontiguousarray(sub)) - 1.0) <= 1e-12


def test_device_space_equals_host_space(ctx, orc):
    import torch
    img = synth.large_photo(644, 483, 7)          # width not a multiple of 4 pixels x 16 bytes
    d = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()
    fast = ctx.GaussianBlur(d, 2.0); ctx.sync()
    assert np.array_equal(fast.cpu().numpy(), ctx.GaussianBlur(img, 2.0))
    sh = ctx.AdaptiveSharpen(d, 0.5); ctx.sync()
    assert np.array_equal(sh.cpu().numpy(), orc.adaptive_sharpen(img, 0.5))
    rs = ctx.lanczosResize(d, 322, 241); ctx.sync()
    assert np.array_equal(rs.cpu().numpy(), orc.lanczos_resize(img, 322, 241))
    assert abs(ctx.SSIMFast(d, fast) - orc.ssim_fast(img, fast.cpu().numpy())) <= SSIM_TOL
    assert abs(ctx.MSSSIM(d, rs) - orc.msssim(img, rs.cpu().numpy())) <= SSIM_TOL
    # strided device view (rows contiguous, stride != 4*w)
    sub = d[3:403, 4:604]
    bl = ctx.GaussianBlur(sub, 1.0, exact=True); ctx.sync()
    assert np.array_equal(bl.cpu().numpy(), orc.gaussian_blur(img[3:403, 4:604], 1.0))


def test_batched_forms_equal_single(ctx, orc):
    import torch
    imgs = [synth.large_photo(1280, 720, k) for k in range(5)]
    d = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    outs = ctx.GaussianBlurBatch(d, 2.0)
    ss = ctx.SSIMFastBatch(d, outs)
    for k in range(5):
        single = ctx.GaussianBlur(imgs[k], 2.0)
        assert np.array_equal(outs[k].cpu().numpy(), single)
        assert ss[k] == ctx.SSIMFast(imgs[k], single)
        assert abs(ss[k] - orc.ssim_fast(imgs[k], single)) <= SSIM_TOL
    outs = ctx.GaussianBlurBatch(d, 2.0, exact=True); ctx.sync()
    assert np.array_equal(outs[2].cpu().numpy(), orc.gaussian_blur(imgs[2], 2.0, procs=8))


def _one_pass_case(ctx, orc, imgs, sigma, exact=False, check_oracle=(0,)):
    """fnx_gaussian_blur_ssim_fast_batch == the two ops run separately (bit for bit), and the
    score is SSIMFast of (source, the blurred image it returned) per the oracle."""
    import torch
    d = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    outs, ss = ctx.GaussianBlurSSIMFastBatch(d, sigma, exact=exact)
    ref = ctx.GaussianBlurBatch(d, sigma, exact=exact)
    ref_ss = ctx.SSIMFastBatch(d, ref) if max(imgs[0].shape[:2]) > 512 else \
        np.array([ctx.SSIMFast(a, b) for a, b in zip(d, ref)])
    for k in range(len(imgs)):
        assert torch.equal(outs[k], ref[k])
        assert ss[k] == ref_ss[k]
    for k in check_oracle:
        assert abs(ss[k] - orc.ssim_fast(imgs[k], outs[k].cpu().numpy())) <= SSIM_TOL


def test_blur_ssimfast_one_pass_4k(ctx, orc):
    imgs = [synth.large_photo(3840, 2160, 3), synth.noise_image(3840, 2160, 9, alpha=True),
            synth.make_test_image_with_alpha(3840, 2160)]
    _one_pass_case(ctx, orc, imgs, 2.0, check_oracle=(0, 1, 2))
    _one_pass_case(ctx, orc, imgs, 2.0, exact=True, check_oracle=(0, 1, 2))


@pytest.mark.parametrize("w,h", [(3001, 2005), (4099, 2817), (2900, 700), (640, 3333), (7680, 4320),
                                 (2817, 2816), (5000, 64), (6100, 3000), (2560, 1440), (1920, 1080), (8190, 4607), (8200, 120)])
def test_blur_ssimfast_one_pass_shapes(ctx, orc, w, h):
    imgs = [synth.noise_image(w, h, w ^ h, alpha=True), synth.large_photo(w, h, 1)]
    _one_pass_case(ctx, orc, imgs, 2.0, check_oracle=(0,))


def test_blur_ssimfast_one_pass_exact_is_the_reference(ctx, orc):
    """exact=True: the blurred images ARE the oracle's GaussianBlur, bit for bit, in the one-pass kernel too,
    and the score is the oracle's SSIMFast of that pair."""
    import torch
    imgs = [synth.noise_image(3840, 2160, 31, alpha=True), synth.large_photo(3840, 2160, 8)]
    d = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    outs, ss = ctx.GaussianBlurSSIMFastBatch(d, 2.0, exact=True)
    for k, img in enumerate(imgs):
        want = orc.gaussian_blur(img, 2.0, procs=16)
        assert np.array_equal(outs[k].cpu().numpy(), want)
        assert abs(ss[k] - orc.ssim_fast(img, want, procs=16)) <= SSIM_TOL
    for w, h in ((3001, 2005), (2560, 1440), (8190, 4607), (1920, 1080)):
        _one_pass_case(ctx, orc, [synth.noise_image(w, h, w + h, alpha=True), synth.large_photo(w, h, 2)], 2.0,
                       exact=True, check_oracle=())
    one = [synth.noise_image(3000, 2000, 5, alpha=True)]
    for sigma in (0.3, 0.6, 1.0, 1.3, 1.6, 2.3, 2.6, 4.0):      # radius 1..8, then 12 (two-op route)
        _one_pass_case(ctx, orc, one, sigma, exact=True, check_oracle=())


@pytest.mark.parametrize("radius", (1, 3, 6, 8))
def test_blur_ssimfast_one_pass_exact_ties(ctx, orc, radius):
    """The guarded one-pass kernel where every pixel of a tile sits on a rounding boundary (dyadic kernel,
    stripe images): the tile-wide recompute must also rebuild the blurred-side box sums."""
    import torch
    k = _binomial(radius)
    w, h = 2048, 1200
    cols = np.zeros((h, w, 4), np.uint8); cols[:, 1::2, :3] = 1; cols[..., 3] = 255
    rows = np.zeros((h, w, 4), np.uint8); rows[1::2, :, :3] = 3; rows[..., 3] = 255
    mixed = synth.noise_image(w, h, radius, alpha=True); mixed[300:900, 500:1500, :3] &= 1
    imgs = [cols, rows, mixed]
    d = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    outs, ss = ctx.GaussianBlurSSIMFastBatch(d, 1.0, exact=True, kernel=k)
    for i, img in enumerate(imgs):
        want = orc.gaussian_blur(img, 1.0, kernel=k, procs=8)
        assert np.array_equal(outs[i].cpu().numpy(), want), (radius, i)
        assert abs(ss[i] - orc.ssim_fast(img, want, procs=8)) <= SSIM_TOL, (radius, i)
    # a kernel outside the guarded kernel's bound (gain > 1) takes the two-op route and still matches
    k2 = np.array([0.7, 0.7, 0.7])
    outs, ss = ctx.GaussianBlurSSIMFastBatch(d[2:], 1.0, exact=True, kernel=k2)
    want = orc.gaussian_blur(mixed, 1.0, kernel=k2, procs=8)
    assert np.array_equal(outs[0].cpu().numpy(), want)
    assert abs(ss[0] - orc.ssim_fast(mixed, want, procs=8)) <= SSIM_TOL


def test_blur_ssimfast_one_pass_strided_views(ctx, orc):
    """sources that are sub-rectangles of larger device images (stride != 4*w), 4-byte aligned only"""
    import torch
    big = torch.from_numpy(synth.noise_image(4000, 2300, 21, alpha=True)).cuda()
    torch.cuda.synchronize()
    views = [big[10:10 + 2160, 16:16 + 3840], big[33:33 + 2160, 101:101 + 3840]]
    outs, ss = ctx.GaussianBlurSSIMFastBatch(views, 2.0)
    for v, o, s_ in zip(views, outs, ss):
        host = np.ascontiguousarray(v.cpu().numpy())
        assert_blur_close(o.cpu().numpy(), orc.gaussian_blur(host, 2.0, procs=16))
        assert abs(s_ - orc.ssim_fast(host, o.cpu().numpy(), procs=16)) <= SSIM_TOL


def test_blur_ssimfast_one_pass_radii_and_fallbacks(ctx, orc):
    imgs = [synth.noise_image(3000, 2000, 5, alpha=True)]
    for sigma in (0.3, 0.6, 1.0, 1.3, 1.6, 2.3, 2.6, 4.0):      # radius 1..8, then 12 (generic kernels)
        _one_pass_case(ctx, orc, imgs, sigma, check_oracle=())
    # radius 31..33 made the tile-height arithmetic divide by zero before the radius check (round-1 advisor finding)
    for sigma in (10.1, 10.5, 11.0):
        _one_pass_case(ctx, orc, [synth.noise_image(1400, 900, 6, alpha=True)], sigma, check_oracle=())
    # boxes too small for the one-pass tables / no downsample at all
    _one_pass_case(ctx, orc, [synth.large_photo(1280, 720, 2), synth.large_photo(1280, 720, 4)], 2.0)
    _one_pass_case(ctx, orc, [synth.large_photo(500, 300, 2)], 2.0)


@pytest.mark.parametrize("sigma,radius", [(2.3, 7), (2.6, 8), (3.0, 9), (4.0, 12), (4.6, 14), (4.7, 15), (6.0, 18), (7.3, 22), (7.4, 23), (8.0, 24)])
def test_blur_ssimfast_one_pass_wide_radii(ctx, orc, sigma, radius):
    """r5: radii 7 .. 24 take the one-pass form too (blur_mfma_wide_kernel<2 / 3 / 4, ., SCORE>): the images are the two-call route's
    bit for bit (fast and exact), the exact ones are the oracle's, the scores are SSIMFast of what came back; shapes with
    edge strips, short last segments and a translucent image."""
    import torch
    imgs = [synth.large_photo(3840, 2160, 11), synth.noise_image(3840, 2160, 12, alpha=True)]
    for exact in (False, True):
        _one_pass_case(ctx, orc, imgs, sigma, exact=exact, check_oracle=(0, 1))
    d = [torch.from_numpy(imgs[1]).cuda()]
    outs, ss = ctx.GaussianBlurSSIMFastBatch(d, sigma, exact=True)
    assert "blur_mfma_wide_kernel<SCORE" in ctx.last_kernel(1), ctx.last_kernel(1)      # the route the library took
    want = orc.gaussian_blur(imgs[1], sigma, procs=16)
    assert np.array_equal(outs[0].cpu().numpy(), want)
    assert abs(ss[0] - orc.ssim_fast(imgs[1], want, procs=16)) <= SSIM_TOL
    for w, h in ((3001, 2005), (2560, 1440), (4099, 2817), (7680, 4320), (2900, 700)):
        _one_pass_case(ctx, orc, [synth.noise_image(w, h, w + radius, alpha=True), synth.large_photo(w, h, 2)], sigma,
                       exact=(w & 1) == 1, check_oracle=(0,))


def test_blur_ssimfast_one_pass_wide_exact_ties(ctx, orc):
    """the guarded wide kernel's SCORE form where whole tiles sit on rounding boundaries (dyadic kernel, stripes): the box sums
    must be those of the patched image"""
    import torch
    k = _binomial(10)
    w, h = 2048, 1200
    cols = np.zeros((h, w, 4), np.uint8); cols[:, 1::2, :3] = 1; cols[..., 3] = 255
    mixed = synth.noise_image(w, h, 77, alpha=True); mixed[300:900, 500:1500, :3] &= 1
    imgs = [cols, mixed]
    d = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    outs, ss = ctx.GaussianBlurSSIMFastBatch(d, 1.0, exact=True, kernel=k)
    for i, img in enumerate(imgs):
        want = orc.gaussian_blur(img, 1.0, kernel=k, procs=8)
        assert np.array_equal(outs[i].cpu().numpy(), want), i
        assert abs(ss[i] - orc.ssim_fast(img, want, procs=8)) <= SSIM_TOL, i


def test_reference_named_blur_is_exact_for_host_images(orc):
    """fennec_GaussianBlur (the drop-in mirror): bit-exact for host images, fast kernel for device tensors."""
    import torch
    img = synth.noise_image(333, 217, 8, alpha=True)
    want = orc.gaussian_blur(img, 2.0)
    assert np.array_equal(fennec_amd.GaussianBlur(img, 2.0), want)
    dev = fennec_amd.GaussianBlur(torch.from_numpy(img).cuda(), 2.0)
    fennec_amd.default_context(0).sync()
    assert_blur_close(dev.cpu().numpy(), want)


def test_determinism(ctx):
    a = synth.large_photo(1920, 1080, 1)
    b = ctx.GaussianBlur(a, 2.0)
    vals = {ctx.SSIMFast(a, b) for _ in range(5)} | {ctx.SSIM(a, b) for _ in range(3)}
    assert len(vals) == 2          # fixed reduction trees: bit-identical from run to run


# ------------------------------------------------------------------ BASELINE.json full sizes
def test_config2_4k_blur_ssimfast(ctx, orc):
    """4K: GaussianBlur sigma=2 + SSIMFast vs original, against the oracle (threaded)."""
    img = synth.large_photo(3840, 2160, 0)
    want = orc.gaussian_blur(img, 2.0, procs=16)
    assert np.array_equal(ctx.GaussianBlur(img, 2.0, exact=True), want)
    fast = ctx.GaussianBlur(img, 2.0)
    assert_blur_close(fast, want)
    assert abs(ctx.SSIMFast(img, want) - orc.ssim_fast(img, want)) <= SSIM_TOL
    assert np.array_equal(ctx.boxDownsample(img, 512, 288), orc.box_downsample(img, 512, 288))
    # size-independent properties
    solid = synth.make_solid_image(3840, 2160, (13, 200, 77, 128))
    assert np.array_equal(ctx.GaussianBlur(solid, 2.0), solid)     # blur of a constant is the constant
    assert ctx.SSIMFast(img, img) == 1.0


def test_config3_4k_lanczos_msssim(ctx, orc):
    img = synth.large_photo(3840, 2160, 1)
    small = ctx.lanczosResize(img, 1920, 1080)
    assert np.array_equal(small, orc.lanczos_resize(img, 1920, 1080, procs=16))
    got = ctx.MSSSIM(img, small)
    want = orc.msssim(img, small, procs=16)
    assert abs(got - want) <= SSIM_TOL


def test_ssim_enqueue_fifo(ctx, orc):
    """fnx_ssim_enqueue: three pairs queued, fetched oldest first, each equal to the blocking SSIM."""
    import torch
    imgs = [synth.large_photo(640, 480, k) for k in range(3)]
    d = [torch.from_numpy(i).cuda() for i in imgs]
    sharp = [ctx.AdaptiveSharpen(t, 0.5) for t in d]
    for a, b in zip(d, sharp):
        ctx.ssim_enqueue(a, b)
    got = [ctx.fetch_result() for _ in d]
    for k, (a, b) in enumerate(zip(d, sharp)):
        assert got[k] == ctx.SSIM(a, b)
        assert abs(got[k] - orc.ssim(imgs[k], b.cpu().numpy())) <= SSIM_TOL
    tiny = torch.from_numpy(synth.noise_image(5, 3, 1, alpha=True)).cuda()
    ctx.ssim_enqueue(tiny, tiny)
    assert ctx.fetch_result() == ctx.SSIM(tiny, tiny)


def test_config4_8k_adaptive_sharpen_ssim(ctx, orc):
    img = synth.large_photo(7680, 4320, 2)
    sharp = ctx.AdaptiveSharpen(img, 0.5)
    assert np.array_equal(sharp, orc.adaptive_sharpen(img, 0.5, procs=16))
    # orientation round trips at full size (pure permutations)
    assert np.array_equal(ctx.ApplyOrientation(ctx.ApplyOrientation(img, 6), 8), img)
    assert np.array_equal(ctx.ApplyOrientation(ctx.ApplyOrientation(img, 5), 5), img)
    # full-resolution SSIM against the oracle on a 2K crop, by property at 8K
    crop, scrop = np.ascontiguousarray(img[:1080, :2048]), np.ascontiguousarray(sharp[:1080, :2048])
    assert abs(ctx.SSIM(crop, scrop) - orc.ssim(crop, scrop, procs=16)) <= SSIM_TOL
    s = ctx.SSIM(img, sharp)
    assert 0.0 < s < 1.0 and ctx.SSIM(img, img) == 1.0
    # r4: the full 8K pair against the oracle (33 M windows, ~30 s of host time on 64 threads): config 4 at its own size
    want = orc.ssim(img, sharp, procs=64)
    assert abs(s - want) <= SSIM_TOL
    # r6: the fp32-moment mode (FNX_SSIM_FAST, windowed_ssim_march2f_kernel) at config 4's own size: SURVEY Appendix A's
    # tolerance for fp32-moment paths is 1e-6; identical images still give exactly 1
    ctx.set_ssim_mode(True)
    try:
        fast = ctx.SSIM(img, sharp)
        assert ctx.last_kernel(fennec_amd.PROF_SSIM) == "windowed_ssim_march2f_kernel"
        assert abs(fast - want) <= SSIM_FAST_TOL, (fast, want)
        assert ctx.SSIM(img, img) == 1.0
    finally:
        ctx.set_ssim_mode(False)
    assert ctx.SSIM(img, sharp) == s                       # (and the default is the fp64 kernel again)


def test_gaussian_blur_ssim_fast_one_call(ctx, orc):
    """r6: fnx_gaussian_blur_ssim_fast -- GaussianBlur + SSIMFast(src, blurred) of one image in one call (a host image
    crosses PCIe once each way): the bytes and the score of the two separate calls, host arrays (tight and pitched), device
    tensors, sizes on both sides of the one-pass kernel's range and pixelSSIM's."""
    import torch
    for (w, h, sigma) in [(3840, 2160, 2.0), (2048, 1536, 1.2), (640, 480, 2.0), (1920, 1080, 3.0), (5, 5, 1.0), (7, 300, 0.8), (517, 389, 2.0)]:
        img = synth.large_photo(w, h, w % 7)
        for exact in (True, False):
            want_b = ctx.GaussianBlur(img, sigma, exact=exact)
            want_s = ctx.SSIMFast(img, want_b)
            got_b, got_s = ctx.GaussianBlurSSIMFast(img, sigma, exact=exact)
            assert np.array_equal(got_b, want_b) and got_s == want_s, (w, h, sigma, exact)
        if w * h <= 2048 * 1536:
            ob = orc.gaussian_blur(img, sigma, procs=8)
            got_b, got_s = ctx.GaussianBlurSSIMFast(img, sigma)
            assert np.array_equal(got_b, ob) and abs(got_s - orc.ssim_fast(img, ob)) <= SSIM_TOL
        want_b = ctx.GaussianBlur(img, sigma, exact=True)
        if w >= 8 and h >= 8:        # (below that SSIMFast is pixelSSIM, which walks the FLAT Pix slices: a pitched source against a tight result is the reference's panic)
            pitched = np.ascontiguousarray(np.pad(img, ((0, 0), (1, 2), (0, 0))))[:, 1: 1 + w]      # a host view with row padding
            got_b, got_s = ctx.GaussianBlurSSIMFast(pitched, sigma)
            assert np.array_equal(got_b, want_b) and got_s == ctx.SSIMFast(img, want_b)
            reuse = np.empty_like(img)
            rb, rs = ctx.GaussianBlurSSIMFast(img, sigma, out=reuse)
            assert rb is reuse and np.array_equal(reuse, want_b) and rs == got_s
        d = torch.from_numpy(img).cuda()
        db, ds = ctx.GaussianBlurSSIMFast(d, sigma)
        ctx.sync()
        assert np.array_equal(db.cpu().numpy(), want_b) and ds == ctx.SSIMFast(img, want_b)
    # an enqueued result waiting in the FIFO does not get in the way (the call has a result slot of its own), nor the call in its
    img = synth.large_photo(3840, 2160, 1)
    d = torch.from_numpy(img).cuda()
    ctx.ssim_enqueue(d, d)
    db, ds = ctx.GaussianBlurSSIMFast(d, 2.0)
    assert ds == ctx.SSIMFast(img, ctx.GaussianBlur(img, 2.0, exact=True))
    assert ctx.fetch_result() == 1.0


SSIM_FAST_TOL = 1e-6


def _photo_like(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 97.0) * np.cos(y / 61.0) + rng.normal(0, 4, x.shape) for _ in range(3)] + [np.full(x.shape, 255.0)], -1)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("case", ["ramp_sharpen", "ramp_blur", "noise_blur", "unrelated", "photo_blur", "bright_blur", "dark_blur",
                                  "flat_noisy", "half_bright_half_dark", "translucent", "strided_views"])
def test_ssim_fast_moments(ctx, orc, case):
    """FNX_SSIM_FAST against the oracle on planes of 4.4 M windows (the two-column marching kernel's range): the fp32
    cancellation-free form [1 - md^2 / (muA^2 + muB^2 + C1)] [1 - Var(a - b) / (sigma_aa + sigma_bb + C2)] about a per-wave
    centre.  Content chosen against it: a bright flat image against a noisy copy (E[x^2] >> the variances), unrelated images
    (the Var(a - b) term at full weight), levels that change inside a wave's segment (the centre fits neither half), alpha
    (ignored by toLuminance), pitched device views.  |delta| <= 1e-6 (measured <= 6e-8); the exact mode is checked beside it."""
    import torch
    w, h = 2600, 1700
    rng = np.random.default_rng(11)
    ramp = synth.large_photo(w, h, 1)
    if case == "ramp_sharpen":
        a, b = ramp, orc.adaptive_sharpen(ramp, 0.5, procs=16)
    elif case == "ramp_blur":
        a, b = ramp, orc.gaussian_blur(ramp, 2.0, procs=16)
    elif case == "noise_blur":
        a = synth.noise_image(w, h, 3)
        b = orc.gaussian_blur(a, 1.2, procs=16)
    elif case == "unrelated":
        a, b = ramp, synth.noise_image(w, h, 3)
    elif case == "photo_blur":
        a = _photo_like(w, h, 2)
        b = orc.gaussian_blur(a, 2.0, procs=16)
    elif case == "bright_blur":
        a = np.clip(_photo_like(w, h, 2).astype(int) // 8 + 224, 0, 255).astype(np.uint8)
        a[..., 3] = 255
        b = orc.gaussian_blur(a, 1.0, procs=16)
    elif case == "dark_blur":
        a = (_photo_like(w, h, 2) // 16).astype(np.uint8)
        a[..., 3] = 255
        b = orc.gaussian_blur(a, 1.0, procs=16)
    elif case == "flat_noisy":
        a = np.full((h, w, 4), 250, np.uint8)
        a[..., 3] = 255
        b = a.copy()
        b[..., :3] = np.clip(250 + rng.integers(-6, 6, (h, w, 3)), 0, 255)
    elif case == "half_bright_half_dark":
        a = np.full((h, w, 4), 255, np.uint8)
        yy = np.arange(h)[:, None]
        a[..., :3] = np.where((yy // 37) % 2 == 0, 252, 3)[..., None]        # bands shorter than a wave's segment
        a[:, w // 2:, :3] = 255 - a[:, w // 2:, :3]
        b = a.copy()
        b[..., :3] = np.clip(a[..., :3].astype(int) + rng.integers(-5, 6, (h, w, 3)), 0, 255)
    elif case == "translucent":
        a = synth.noise_image(w, h, 5, alpha=True)
        b = orc.gaussian_blur(a, 0.8, procs=16)
        b[..., 3] = rng.integers(0, 256, (h, w), dtype=np.uint8)
    else:
        a = _photo_like(w, h, 4)
        b = orc.gaussian_blur(a, 1.5, procs=16)
    want = orc.ssim(a, b, procs=32)
    if case == "strided_views":
        pa = torch.from_numpy(np.ascontiguousarray(np.pad(a, ((0, 0), (2, 1), (0, 0))))).cuda()[:, 2: 2 + w]
        pb = torch.from_numpy(np.ascontiguousarray(np.pad(b, ((0, 0), (1, 3), (0, 0))))).cuda()[:, 1: 1 + w]
        xa, xb = pa, pb
    else:
        xa, xb = a, b
    exact = ctx.SSIM(xa, xb)
    assert abs(exact - want) <= SSIM_TOL
    ctx.set_ssim_mode(True)
    try:
        fast = ctx.SSIM(xa, xb)
        assert ctx.last_kernel(fennec_amd.PROF_SSIM) == "windowed_ssim_march2f_kernel"
        assert abs(fast - want) <= SSIM_FAST_TOL, (case, fast, want, fast - want)
        # through the enqueue form (config 4's flow) the same number
        da, db = (xa, xb) if case == "strided_views" else (torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
        ctx.ssim_enqueue(da, db)
        assert ctx.fetch_result() == fast
        # planes below the two-column kernel's range keep the fp64 kernels whatever the mode
        small_a, small_b = np.ascontiguousarray(a[:600, :800]), np.ascontiguousarray(b[:600, :800])
        assert abs(ctx.SSIM(small_a, small_b) - orc.ssim(small_a, small_b, procs=8)) <= SSIM_TOL
    finally:
        ctx.set_ssim_mode(False)


@pytest.mark.parametrize("w,h", [(40, 150000), (200000, 40), (2057, 2003)])
def test_ssim_fast_moments_odd_geometry(ctx, orc, w, h):
    """FNX_SSIM_FAST on planes that are one strip wide (every lane past column 32 idle), one segment tall, and of odd size
    (the last strip's idle columns, the last segment's partial group of rows): <= 1e-6 against the oracle, the fp64 kernel <= 1e-9."""
    a = synth.noise_image(w, h, 17)
    a[..., :3] = (a[..., :3] // 3) + 80
    b = a.copy()
    rng = np.random.default_rng(4)
    b[..., :3] = np.clip(b[..., :3].astype(int) + rng.integers(-9, 10, (h, w, 3)), 0, 255)
    want = orc.ssim(a, b, procs=32)
    assert abs(ctx.SSIM(a, b) - want) <= SSIM_TOL
    ctx.set_ssim_mode(True)
    try:
        fast = ctx.SSIM(a, b)
        assert ctx.last_kernel(fennec_amd.PROF_SSIM) == "windowed_ssim_march2f_kernel"
        assert abs(fast - want) <= SSIM_FAST_TOL, (w, h, fast - want)
    finally:
        ctx.set_ssim_mode(False)


@pytest.mark.parametrize("w,h", [(1920, 1080), (640, 480), (2600, 1700), (100, 37), (5, 5)])
def test_ssim_and_sharpen_batches(ctx, orc, w, h):
    """r6: fnx_ssim_batch_enqueue and fnx_(adaptive_)sharpen_batch -- n same-geometry device images in ONE launch (the image is a
    grid dimension of the window / streaming kernel; separately allocated images by pointer array): every value and every
    byte equals the single call's, and -- for the smaller sizes -- the oracle's.  Pitched views take the per-image path."""
    import torch
    n = 5
    A = [synth.large_photo(w, h, k) for k in range(n)]
    da = [torch.from_numpy(x).cuda() for x in A]
    for adaptive in (True, False):
        if w < 3 or h < 3:
            continue
        outs = ctx.sharpen_batch(da, 0.5, adaptive=adaptive)
        ctx.sync()
        for k in range(n):
            want = ctx.AdaptiveSharpen(A[k], 0.5) if adaptive else ctx.Sharpen(A[k], 0.5)
            assert np.array_equal(outs[k].cpu().numpy(), want), (adaptive, k)
            if w * h <= 700 * 500 and k == 0:
                assert np.array_equal(want, orc.adaptive_sharpen(A[k], 0.5) if adaptive else orc.sharpen(A[k], 0.5))
    B = [ctx.GaussianBlur(x, 0.8 + 0.3 * k) for k, x in enumerate(A)]
    db = [torch.from_numpy(x).cuda() for x in B]
    want = [ctx.SSIM(A[k], B[k]) for k in range(n)]
    ctx.ssim_batch_enqueue(da, db)
    got = ctx.fetch_results(n)
    # (a batch cuts the planes into other segments than a single call does: the window values are the same, the order their
    # sum is taken in is not -- 1e-13, against the 1e-9 bar)
    assert np.max(np.abs(got - np.array(want))) <= 1e-12
    if w * h <= 700 * 500:
        assert abs(want[1] - orc.ssim(A[1], B[1], procs=8)) <= SSIM_TOL
    if w >= 8 and h >= 8:
        # pitched device views: the SSIM batch reads rows by stride; the sharpen batch goes image by image (the flat-copy pass)
        pa = [torch.from_numpy(np.ascontiguousarray(np.pad(x, ((0, 0), (1, 2), (0, 0))))).cuda()[:, 1: 1 + w] for x in A[:2]]
        ctx.ssim_batch_enqueue(pa, db[:2])
        assert np.max(np.abs(ctx.fetch_results(2) - np.array(want[:2]))) <= 1e-12
        vo = ctx.sharpen_batch(pa, 0.5, adaptive=True)
        ctx.sync()
        for k in range(2):
            assert np.array_equal(vo[k].cpu().numpy(), ctx.AdaptiveSharpen(pa[k], 0.5).cpu().numpy())
    if w * h >= 2600 * 1700:                                  # the fp32-moment mode through the batch
        ctx.set_ssim_mode(True)
        try:
            ctx.ssim_batch_enqueue(da[:2], db[:2])
            gf = ctx.fetch_results(2)
            assert ctx.last_kernel(fennec_amd.PROF_SSIM) == "windowed_ssim_march2f_kernel"
            assert all(abs(g - x) <= SSIM_FAST_TOL for g, x in zip(gf, want[:2]))
        finally:
            ctx.set_ssim_mode(False)


def test_ssim_mode_argument(ctx):
    with pytest.raises(fennec_amd.FennecError):
        ctx._chk(ctx._lib.fnx_ctx_set_ssim_mode(ctx._h, 7), "fnx_ctx_set_ssim_mode")


# ------------------------------------------------------------------ fused blur kernel: every radius, odd shapes
@pytest.mark.parametrize("name", ["photo_640x480", "noise_131x77", "alpha_37x29", "noise_3x5", "noise_300x2"])
def test_blur_fast_all_radii(ctx, orc, name):
    img = IMAGES[name]()
    for sigma in (0.3, 0.6, 1.0, 1.3, 1.6, 2.0, 2.3, 2.6):     # radius 1..8
        assert_blur_close(ctx.GaussianBlur(img, sigma), orc.gaussian_blur(img, sigma))


def test_blur_fast_shapes(ctx, orc):
    for (w, h) in [(1030, 300), (513, 64), (2049, 33), (1281, 721)]:
        img = synth.noise_image(w, h, w + h, alpha=True)
        assert_blur_close(ctx.GaussianBlur(img, 2.0), orc.gaussian_blur(img, 2.0, procs=8))


# ------------------------------------------------------------------ concurrency (CompressBatch workers)
def test_concurrent_contexts_threads(orc):
    """batch.go:84-123: workers call the hot path concurrently; one fnx_ctx per worker thread."""
    import threading
    imgs = [synth.large_photo(800, 600, k) for k in range(8)]
    want = [(orc.gaussian_blur(i, 2.0, procs=4), None) for i in imgs]
    want = [(b, orc.ssim_fast(i, b)) for i, (b, _) in zip(imgs, want)]
    errs = []

    def worker(wid):
        try:
            c = fennec_amd.Context(0)
            for rep in range(6):
                for k in range(wid, 8, 4):
                    b = c.GaussianBlur(imgs[k], 2.0, exact=True)
                    assert np.array_equal(b, want[k][0])
                    assert abs(c.SSIMFast(imgs[k], b) - want[k][1]) <= SSIM_TOL
                    assert np.array_equal(c.ApplyOrientation(imgs[k], 6), orc.apply_orientation(imgs[k], 6))
            c.close()
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


def test_jpeg_quality_search_on_gpu(ctx, orc):
    """compress.go:21-87 harness: GPU SSIMFast (prepared reference) drives the same binary search
    as the oracle's SSIMFast; same quality, same SSIM to 1e-9."""
    from fennec_amd import batch
    src = synth.make_test_image(640, 480)
    prep = ctx.ssim_fast_prepare(src)
    q1, s1, d1, n1 = batch.compress_jpeg_optimal(lambda dec: prep.against(dec), src, 0.94)
    q2, s2, d2, n2 = batch.compress_jpeg_optimal(lambda dec: orc.ssim_fast(src, dec), src, 0.94)
    prep.close()
    assert (q1, n1) == (q2, n2) and d1 == d2 and abs(s1 - s2) <= SSIM_TOL
    assert 30 <= q1 <= 100 and s1 >= 0.94


# ------------------------------------------------------------------ randomized shapes
def test_fuzz_random_shapes(ctx, orc):
    """Seeded random geometry sweep: every op against the oracle on odd sizes, thin images,
    sizes around the tile edges (64, 52, 128 ...) and random parameters."""
    rng = np.random.default_rng(20260928)
    special = [1, 2, 3, 7, 8, 9, 31, 51, 52, 53, 63, 64, 65, 103, 104, 105, 127, 128, 129, 191, 257]
    for it in range(36):
        w = int(rng.choice(special)) if rng.random() < 0.6 else int(rng.integers(1, 300))
        h = int(rng.choice(special)) if rng.random() < 0.6 else int(rng.integers(1, 300))
        img = synth.noise_image(w, h, 1000 + it, alpha=True)
        sigma = float(rng.choice([0.3, 0.7, 1.0, 1.4, 2.0, 2.5, 3.1]))
        want = orc.gaussian_blur(img, sigma)
        assert np.array_equal(ctx.GaussianBlur(img, sigma, exact=True), want), (w, h, sigma)
        assert_blur_close(ctx.GaussianBlur(img, sigma), want)
        s = float(rng.uniform(0.05, 1.0))
        assert np.array_equal(ctx.Sharpen(img, s), orc.sharpen(img, s)), (w, h, s)
        assert np.array_equal(ctx.AdaptiveSharpen(img, s), orc.adaptive_sharpen(img, s)), (w, h, s)
        dw, dh = int(rng.integers(1, 2 * w + 2)), int(rng.integers(1, 2 * h + 2))
        assert np.array_equal(ctx.lanczosResize(img, dw, dh), orc.lanczos_resize(img, dw, dh)), (w, h, dw, dh)
        assert np.array_equal(ctx.boxDownsample(img, dw, dh), orc.box_downsample(img, dw, dh)), (w, h, dw, dh)
        o = int(rng.integers(2, 9))
        assert np.array_equal(ctx.ApplyOrientation(img, o), orc.apply_orientation(img, o)), (w, h, o)
        other = synth.noise_image(w, h, 5000 + it, alpha=True)
        assert abs(ctx.SSIM(img, other) - orc.ssim(img, other)) <= SSIM_TOL, (w, h)
        assert abs(ctx.SSIMFast(img, want) - orc.ssim_fast(img, want)) <= SSIM_TOL, (w, h)
        assert abs(ctx.MSSSIM(img, want) - orc.msssim(img, want)) <= SSIM_TOL, (w, h)


def test_fuzz_large_downsample_shapes(ctx, orc):
    """SSIMFast's >512 path (tiled box kernel) on awkward sizes and strides."""
    rng = np.random.default_rng(7)
    for (w, h) in [(513, 9), (9, 513), (1000, 1000), (1023, 517), (2047, 64), (777, 1555), (4001, 13)]:
        a = synth.noise_image(w, h, w * 7 + h, alpha=True)
        b = synth.noise_image(w, h, w * 11 + h, alpha=True)
        assert abs(ctx.SSIMFast(a, b) - orc.ssim_fast(a, b)) <= SSIM_TOL, (w, h)
        _, nw, nh = orc.ssim_fast_dims(w, h)
        assert np.array_equal(ctx.boxDownsample(a, nw, nh), orc.box_downsample(a, nw, nh)), (w, h)
        assert np.array_equal(ctx.boxDownsample(a, max(w // 300, 1), max(h // 300, 1)),
                              orc.box_downsample(a, max(w // 300, 1), max(h // 300, 1))), (w, h)   # boxes > 257 rows


def test_async_batch_enqueue_fetch_two_contexts(orc):
    """fnx_ssim_fast_batch_enqueue / fnx_results_fetch: two contexts driven by one host thread,
    work queued on a stream AFTER the batch must not delay or corrupt the fetched results."""
    import torch
    c0, c1 = fennec_amd.Context(0), fennec_amd.Context(0)
    imgs = [synth.large_photo(1024, 768, k) for k in range(6)]
    d = [torch.from_numpy(i).cuda() for i in imgs]
    o = [torch.empty_like(t) for t in d]
    torch.cuda.synchronize()
    pb0 = c0.plan_blur_batch(d[:3], 2.0, outs=o[:3]); pb1 = c1.plan_blur_batch(d[3:], 2.0, outs=o[3:])
    ps0 = c0.plan_ssim_fast_batch(d[:3], o[:3]); ps1 = c1.plan_ssim_fast_batch(d[3:], o[3:])
    pb0.run(); pb1.run()
    for rep in range(3):
        ps0.enqueue(); ps1.enqueue()
        pb0.run(); pb1.run()                      # queued behind the scoring, overwrites the same outputs with the same values
        got = np.concatenate([ps0.fetch().copy(), ps1.fetch().copy()])
        c0.sync(); c1.sync()
        for k in range(6):
            want_blur = c0.GaussianBlur(imgs[k], 2.0)
            assert np.array_equal(o[k].cpu().numpy(), want_blur)
            assert abs(got[k] - orc.ssim_fast(imgs[k], want_blur)) <= SSIM_TOL
    with pytest.raises(fennec_amd.FennecError):
        fennec_amd.Context(0).plan_ssim_fast_batch(d[:1], o[:1]).fetch()     # nothing enqueued on that ctx


def test_device_views_at_4_byte_alignment(ctx, orc):
    """Device sub-views whose rows start only 4-byte aligned (odd x offset, odd width): the blur's
    16-byte window loads and the box kernel's vector path must both cope."""
    import torch
    big = synth.noise_image(1301, 700, 99, alpha=True)
    d = torch.from_numpy(big).cuda()
    torch.cuda.synchronize()
    for (y0, y1, x0, x1) in [(0, 700, 1, 1300), (3, 650, 5, 1234), (10, 80, 7, 72)]:
        sub_h = np.ascontiguousarray(big[y0:y1, x0:x1])
        sub_d = d[y0:y1, x0:x1]
        out = ctx.GaussianBlur(sub_d, 2.0); ctx.sync()
        assert_blur_close(out.cpu().numpy(), orc.gaussian_blur(sub_h, 2.0, procs=8))
        out = ctx.AdaptiveSharpen(sub_d, 0.7); ctx.sync()
        # a SubImage: the border is the reference's FLAT copy (effects.go:68), so the checker gets the view, not a packed copy
        assert np.array_equal(out.cpu().numpy(), orc.adaptive_sharpen(big[y0:y1, x0:x1], 0.7, procs=8))
        assert abs(ctx.SSIMFast(sub_d, sub_d.flip(1).contiguous()) - orc.ssim_fast(sub_h, np.ascontiguousarray(sub_h[:, ::-1]))) <= SSIM_TOL
        sm = ctx.boxDownsample(sub_d, max((x1 - x0) // 3, 1), max((y1 - y0) // 3, 1)); ctx.sync()
        assert np.array_equal(sm.cpu().numpy(), orc.box_downsample(sub_h, max((x1 - x0) // 3, 1), max((y1 - y0) // 3, 1)))
        rs = ctx.lanczosResize(sub_d, 200, 100); ctx.sync()
        assert np.array_equal(rs.cpu().numpy(), orc.lanczos_resize(sub_h, 200, 100, procs=8))


# ------------------------------------------------------------------ Analyze (analyze.go), SURVEY 8(f).3
def _serial_sum_tol(n):
    """brightSum is ONE serial fp64 chain over all n pixels in the reference (analyze.go:63): its
    own rounding error is bounded by (n-1)*2^-53 relative (1.1e-12 observed at 4K, against the
    exact sum); the device's fixed-tree sum is closer to exact, so the bar is the chain's bound."""
    return max(1e-12, n * 2.0 ** -53)



def _check_analysis(raw, st, want):
    """raw = fnx_analyze's accumulators, st = ImageStats dict, want = oracle."""
    assert np.array_equal(raw["histogram"].astype(np.float64), want["histogram"])      # exact
    for k in ("has_alpha", "is_grayscale", "unique_colors", "sample_count", "edge_count", "edge_total"):
        assert raw[k] == want[k], k
    rel = _serial_sum_tol(want["width"] * want["height"])
    assert abs(raw["bright_sum"] - want["bright_sum"]) <= rel * abs(want["bright_sum"])
    # (lum - mean)^2 terms inherit the mean's last-bit difference: absolute floor for flat images
    assert abs(raw["variance_sum"] - want["variance_sum"]) <= 1e-9 * abs(want["variance_sum"]) + 1e-6
    assert (st["Width"], st["Height"]) == (want["width"], want["height"])
    assert (st["HasAlpha"], st["IsGrayscale"], st["UniqueColors"]) == (want["has_alpha"], want["is_grayscale"], want["unique_colors"])
    assert abs(st["Entropy"] - want["entropy"]) <= 1e-12
    assert st["EdgeDensity"] == want["edge_density"]
    assert abs(st["MeanBrightness"] - want["mean_brightness"]) <= rel * max(1.0, want["mean_brightness"])
    assert abs(st["Contrast"] - want["contrast"]) <= 1e-9
    assert (st["RecommendedFormat"], st["RecommendedQuality"]) == (want["recommended_format"], want["recommended_quality"])
    assert st["EstimatedCompression"] == want["estimated_compression"]


ANALYZE_IMAGES = dict(IMAGES)
ANALYZE_IMAGES.update({
    "grey_100x100": lambda: synth.make_solid_image(100, 100, (128, 128, 128, 255)),       # fennec_test.go:579
    "grad_200x200": lambda: synth.make_test_image(200, 200),                               # fennec_test.go:566
    "alpha_100x100": lambda: synth.make_test_image_with_alpha(100, 100),                   # fennec_test.go:591
    "greyramp_333x77": lambda: np.repeat((np.arange(333 * 77, dtype=np.uint32) % 256).astype(np.uint8).reshape(77, 333, 1), 4, axis=2) | np.array([0, 0, 0, 255], dtype=np.uint8),
    "photo_1000x1000": lambda: synth.large_photo(1000, 1000, 2),                           # BenchmarkAnalyze's size
    "noise_1921x1081": lambda: synth.noise_image(1921, 1081, 3, alpha=True),
    # many colours, but only far into the sampled sequence (second colour launch must run)
    "late_noise_1000x800": lambda: np.concatenate([synth.make_solid_image(1000, 500, (5, 6, 7, 255)),
                                                   synth.noise_image(1000, 300, 11)], axis=0),
    # fewer than 1024 distinct colours in total, spread over the whole image
    "palette_900x700": lambda: np.stack([(np.arange(900 * 700, dtype=np.uint32) * 7919 % 700 % 256).astype(np.uint8).reshape(700, 900),
                                         (np.arange(900 * 700, dtype=np.uint32) * 7919 % 700 // 256).astype(np.uint8).reshape(700, 900),
                                         np.zeros((700, 900), np.uint8), np.full((700, 900), 255, np.uint8)], axis=2),
    "fewcolors_640x480": lambda: (synth.large_photo(640, 480, 1) & 0xC0) | np.array([0, 0, 0, 255], dtype=np.uint8),
})


@pytest.mark.parametrize("name", list(ANALYZE_IMAGES))
def test_analyze(ctx, orc, name):
    img = np.ascontiguousarray(ANALYZE_IMAGES[name]())
    _check_analysis(ctx.analyze_raw(img), ctx.Analyze(img), orc.analyze(img))
    assert ctx.isOpaque(img) == orc.is_opaque(img)
    assert ctx.isGrayscale(img) == orc.is_grayscale(img)


def test_analyze_4k_device_batch_and_views(ctx, orc):
    import torch
    imgs = [synth.large_photo(3840, 2160, 0), synth.noise_image(3840, 2160, 1, alpha=True),
            synth.make_test_image(3840, 2160)]
    d = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    plan = ctx.plan_analyze_batch(d)
    raw = plan.run()
    stats = plan.stats()
    for k, img in enumerate(imgs):
        want = orc.analyze(img)
        _check_analysis(ctx._analysis_dict(raw[k]), stats[k], want)
        single = ctx.analyze_raw(d[k])                       # device, single
        assert np.array_equal(single["histogram"], np.array(raw[k].histogram[:], dtype=np.uint64))
        assert abs(single["bright_sum"] - raw[k].bright_sum) <= 1e-13 * raw[k].bright_sum   # partial layout depends on n
    # strided views: device (16-byte aligned or not) and host
    sub = d[1][5:1005, 8:1508]
    _check_analysis(ctx.analyze_raw(sub), ctx.Analyze(sub), orc.analyze(np.ascontiguousarray(imgs[1][5:1005, 8:1508])))
    sub = d[1][5:1005, 3:1500]
    _check_analysis(ctx.analyze_raw(sub), ctx.Analyze(sub), orc.analyze(np.ascontiguousarray(imgs[1][5:1005, 3:1500])))
    hv = imgs[0][7:507, 9:1209]
    _check_analysis(ctx.analyze_raw(hv), ctx.Analyze(hv), orc.analyze(np.ascontiguousarray(hv)))
    # determinism: fixed reduction trees
    assert len({ctx.analyze_raw(d[0])["bright_sum"] for _ in range(4)}) == 1
    # empty image (fennec_test.go:602-608)
    st = ctx.Analyze(np.zeros((0, 0, 4), dtype=np.uint8))
    assert (st["Width"], st["Height"], st["Entropy"]) == (0, 0, 0.0)


def test_analyze_colour_tables_alternate_cleanly(orc):
    """the single-launch Analyze keeps two colour tables per ctx: the launch that uses one clears the other.  Batches of
    different sizes and single calls interleaved must each see an empty table (few-colour images count exactly)."""
    import torch
    c = fennec_amd.Context(0)
    few = (synth.large_photo(640, 480, 1) & 0xC0) | np.array([0, 0, 0, 255], dtype=np.uint8)           # < 1024 colours
    pal = np.ascontiguousarray(ANALYZE_IMAGES["palette_900x700"]())
    noisy = [synth.noise_image(640, 480, 20 + k) for k in range(5)]
    want_few, want_pal = orc.analyze(few), orc.analyze(pal)
    assert want_few["unique_colors"] < 1024 and want_pal["unique_colors"] < 1024
    d_noisy = [torch.from_numpy(i).cuda() for i in noisy]
    d_few = [torch.from_numpy(few).cuda() for _ in range(2)]
    torch.cuda.synchronize()
    seq = [("b", d_noisy), ("s", few), ("s", pal), ("b", d_noisy[:2]), ("b", d_few), ("s", few), ("b", d_noisy), ("b", d_few), ("s", pal)]
    for kind, x in seq * 2:
        if kind == "s":
            _check_analysis(c.analyze_raw(x), c.Analyze(x), want_few if x is few else want_pal)
        else:
            plan = c.plan_analyze_batch(x)
            raw = plan.run()
            for k in range(len(x)):
                img = x[k].cpu().numpy()
                want = orc.analyze(img)
                assert c._analysis_dict(raw[k])["unique_colors"] == want["unique_colors"]
                assert np.array_equal(np.array(raw[k].histogram[:], dtype=np.float64), want["histogram"])
    c.close()


def test_flat_scans_see_row_padding(ctx, orc):
    """isOpaque / isGrayscale walk the flat Pix slice (convert.go:66-84): padding counts."""
    base = synth.make_solid_image(64, 32, (9, 9, 9, 255))
    base[:, 63] = (1, 2, 3, 0)                  # last column: transparent and coloured
    view = base[:, :63]                          # SubImage-like: stride 256, width 63
    assert orc.is_opaque(view) is False and orc.is_grayscale(view) is False
    assert ctx.isOpaque(view) is False and ctx.isGrayscale(view) is False
    one_row = base[:1, :63]                      # a single row ends before the padding
    assert ctx.isOpaque(one_row) is True and ctx.isGrayscale(one_row) is True
    assert orc.is_opaque(one_row) is True
    big = synth.large_photo(3840, 2160, 4)
    assert ctx.isOpaque(big) is True and ctx.isGrayscale(big) is False
    big[2159, 3839, 3] = 254
    assert ctx.isOpaque(big) is False


@pytest.mark.parametrize("w,h", [(1, 1), (3, 5), (17, 1), (64, 64), (257, 129), (1021, 767), (3840, 2160)])
def test_flat_scans_single_odd_pixel_anywhere(ctx, orc, w, h):
    """the single-launch scan (its last workgroup reports, workgroups stop early once both flags are known): one pixel that
    is not opaque / not grey at the first, the last and random positions, host images and device views at any alignment"""
    import torch
    rng = np.random.default_rng(w * 7919 + h)
    grey = np.full((h, w, 4), 255, dtype=np.uint8)
    grey[..., :3] = rng.integers(0, 256, size=(h, w, 1), dtype=np.uint8)
    assert ctx.isOpaque(grey) is True and ctx.isGrayscale(grey) is True
    spots = {(0, 0), (h - 1, w - 1)} | {(int(rng.integers(0, h)), int(rng.integers(0, w))) for _ in range(6)}
    for (y, x) in sorted(spots):
        a = grey.copy(); a[y, x, 3] = 254
        assert ctx.isOpaque(a) is False and ctx.isGrayscale(a) is True, (y, x)
        c = grey.copy(); c[y, x, int(rng.integers(0, 3))] ^= 1
        assert ctx.isOpaque(c) is True and ctx.isGrayscale(c) is False, (y, x)
        both = a.copy(); both[h - 1 - y, w - 1 - x, 1] ^= 0x80
        assert ctx.isOpaque(both) is False and ctx.isGrayscale(both) is False
        assert orc.is_opaque(a) is False and orc.is_grayscale(c) is False
    # device-resident, flat and unaligned (a view starting one pixel in: 4-byte alignment only)
    flat = torch.from_numpy(np.concatenate([grey.reshape(-1, 4), grey.reshape(-1, 4)[:1]])).cuda()
    v = flat[1:1 + w * h].reshape(h, w, 4)
    assert ctx.isOpaque(v) is True and ctx.isGrayscale(v) is True
    flat[w * h, 3] = 7                               # the view's last pixel
    torch.cuda.synchronize()
    assert ctx.isOpaque(v) is False and ctx.isGrayscale(v) is True
    for _ in range(3):                               # the counters go back to zero after every launch
        assert ctx.isOpaque(grey) is True and ctx.isGrayscale(grey) is True


# ------------------------------------------------------------------ applyPalette (targetsize.go:488-546), SURVEY 8(f).4
def _palette(n, seed):
    rng = np.random.default_rng(seed)
    pal = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    pal[:, 3] = 255
    if n > 3:
        pal[n // 2] = pal[1]                # duplicates: the lower index wins (strict <)
    return pal


@pytest.mark.parametrize("n", [1, 2, 7, 16, 64, 255, 256])
@pytest.mark.parametrize("name", ["noise_131x77", "photo_640x480", "noise_3x5", "alpha_37x29"])
def test_apply_palette(ctx, orc, name, n):
    img = IMAGES[name]()
    pal = _palette(n, n)
    wi, wq = orc.apply_palette(img, pal)
    gi, gq = ctx.applyPalette(img, pal)
    assert np.array_equal(gi, wi) and np.array_equal(gq, wq)


def test_apply_palette_device_and_extremes(ctx, orc):
    import torch
    img = synth.noise_image(1921, 1081, 5, alpha=True)
    pal = _palette(256, 9)
    wi, wq = orc.apply_palette(img, pal)
    d = torch.from_numpy(img).cuda()
    gi, gq = ctx.applyPalette(d, pal); ctx.sync()
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gq.cpu().numpy(), wq)
    sub = d[3:900, 5:1500]                                   # strided device view, odd width
    gi, gq = ctx.applyPalette(sub, pal); ctx.sync()
    wi, wq = orc.apply_palette(np.ascontiguousarray(img[3:900, 5:1500]), pal)
    assert np.array_equal(gi.cpu().numpy(), wi) and np.array_equal(gq.cpu().numpy(), wq)
    # extreme entries: black/white palette, all-equal distances
    bw = np.array([[0, 0, 0, 255], [255, 255, 255, 255], [0, 0, 0, 255]], dtype=np.uint8)
    gi, _ = ctx.applyPalette(img, bw)
    assert np.array_equal(gi, orc.apply_palette(img, bw)[0]) and gi.max() <= 1
    with pytest.raises(fennec_amd.FennecError):
        ctx.applyPalette(img, np.array([[1, 2, 3, 200]], dtype=np.uint8))      # translucent palette entry


def _clustered_palette(n, seed):
    """entries crowded around a few centres (what medianCut makes of a photograph): long candidate lists, many near ties"""
    rng = np.random.default_rng(seed)
    centres = rng.integers(0, 256, size=(max(1, n // 24), 3))
    pal = np.clip(centres[rng.integers(0, len(centres), size=n)] + rng.integers(-6, 7, size=(n, 3)), 0, 255).astype(np.uint8)
    pal = np.concatenate([pal, np.full((n, 1), 255, np.uint8)], axis=1)
    if n > 5:
        pal[n - 1] = pal[2]                 # a duplicate far behind its twin
    return pal


@pytest.mark.parametrize("mode", ["1", "0"])
@pytest.mark.parametrize("kind,n", [("random", 256), ("random", 17), ("clustered", 256), ("clustered", 100), ("corner", 256), ("one", 1)])
def test_apply_palette_grid_form(ctx, orc, kind, n, mode):
    """the grid of candidate lists (palette.hip, r5) against the walk over the whole palette: random, crowded (cells with more
    than 31 candidates take the marked path) and degenerate palettes; photo-like, noise and grey-ramp pixels; odd sizes"""
    if kind == "random":
        pal = _palette(n, 100 + n)
    elif kind == "clustered":
        pal = _clustered_palette(n, n)
    elif kind == "corner":
        rng = np.random.default_rng(3)
        pal = rng.integers(0, 12, size=(n, 4), dtype=np.uint8)
        pal[:, 3] = 255
    else:
        pal = np.array([[9, 200, 77, 255]], dtype=np.uint8)
    imgs = [synth.make_test_image(701, 397), synth.noise_image(515, 333, 7, alpha=True)]
    ramp = np.zeros((64, 1024, 4), np.uint8)
    ramp[..., :3] = (np.arange(1024) // 4)[None, :, None]
    ramp[..., 3] = 255
    imgs.append(ramp)
    dark = (synth.noise_image(300, 200, 9) // 20).astype(np.uint8)
    dark[..., 3] = 255
    imgs.append(dark)
    ctx.set_form("palette_grid", mode)           # (a per-ctx selection since r6: no child process per mode)
    for k, img in enumerate(imgs):
        wi, wq = orc.apply_palette(img, pal)
        gi, gq = ctx.applyPalette(img, pal)
        assert np.array_equal(gi, wi) and np.array_equal(gq, wq), (kind, n, mode, k)


# ------------------------------------------------------------------ decoded JPEG planes -> NRGBA, SURVEY 8(f).1
@pytest.mark.parametrize("ratio", [0, 1, 2, 3, 4, 5])
def test_ycbcr_to_nrgba(ctx, orc, ratio):
    import torch
    for (w, h) in [(640, 480), (37, 29), (1, 1), (5, 2), (1283, 719)]:
        y, cb, cr = synth.ycbcr_planes(w, h, ratio, 7 * ratio + w)
        want = orc.ycbcr_to_nrgba(y, cb, cr, ratio)
        assert np.array_equal(ctx.ycbcrToNRGBA(y, cb, cr, ratio), want)
        got = ctx.ycbcrToNRGBA(torch.from_numpy(y).cuda(), torch.from_numpy(cb).cuda(), torch.from_numpy(cr).cuda(), ratio)
        ctx.sync()
        assert np.array_equal(got.cpu().numpy(), want)
    y = synth.ycbcr_planes(301, 200, 0, 5)[0]
    assert np.array_equal(ctx.ycbcrToNRGBA(y, None, None, 0), orc.ycbcr_to_nrgba(y, None, None, 0))   # image.Gray


def test_ssim_fast_against_ycbcr_4k(ctx, orc):
    """compress.go:45-74 with the decoder's planes: SSIMFast(src, toNRGBARef(decoded YCbCr))."""
    src = synth.large_photo(3840, 2160, 6)
    prep = ctx.ssim_fast_prepare(src)
    for ratio in (2, 0, 1):
        y, cb, cr = synth.rgb_to_ycbcr_planes(src, ratio)
        dec = orc.ycbcr_to_nrgba(y, cb, cr, ratio)
        want = orc.ssim_fast(src, dec, procs=16)
        assert abs(prep.against_ycbcr(y, cb, cr, ratio) - want) <= SSIM_TOL
        assert prep.against_ycbcr(y, cb, cr, ratio) == prep.against(dec)      # same kernels after the conversion
        assert 0.5 < want < 1.0
    prep.close()


@pytest.mark.parametrize("ratio", [0, 1, 2, 3, 4, 5])
def test_ssim_fast_against_ycbcr_device_planes(ctx, orc, ratio):
    """Round 3: with the planes on the device the candidate's <= 256 px plane is summed straight from them
    (box_tiled_ycc_kernel: no NRGBA image).  Bit-identical to convert-then-downsample -- sizes whose last chunk
    sticks out of the row, every subsample ratio, unaligned plane views (which take the two-kernel path)."""
    import torch
    for (w, h) in [(1283, 719), (3840, 2160), (1030, 517), (600, 258)]:
        y, cb, cr = synth.ycbcr_planes(w, h, ratio, 11 * ratio + w)
        dec = orc.ycbcr_to_nrgba(y, cb, cr, ratio)
        src = synth.large_photo(w, h, 3)
        prep = ctx.ssim_fast_prepare(src)
        want = prep.against(dec)
        dy, dcb, dcr = (torch.from_numpy(p).cuda() for p in (y, cb, cr))
        assert prep.against_ycbcr(dy, dcb, dcr, ratio) == want
        assert prep.against_ycbcr(y, cb, cr, ratio) == want                      # host planes: staged, converted
        assert abs(want - orc.ssim_fast(src, dec, procs=16)) <= SSIM_TOL
        prep.close()


def test_ctx_profile_hook(ctx):
    """fnx_ctx_profile / fnx_ctx_kernel_ms: the library's own HIP events around its dominant kernel."""
    import torch
    c = fennec_amd.Context(0)
    with pytest.raises(fennec_amd.FennecError):
        c.kernel_ms()                                         # nothing recorded yet
    d = [torch.from_numpy(synth.large_photo(3840, 2160, k)).cuda() for k in range(2)]
    torch.cuda.synchronize()
    c.profile(True)
    c.GaussianBlurSSIMFastBatch(d, 2.0)
    ms_score = c.kernel_ms()
    c.GaussianBlurBatch(d, 2.0); c.sync()
    ms_plain = c.kernel_ms()
    c.plan_analyze_batch(d).run()
    ms_an = c.kernel_ms()
    assert 0.005 < ms_plain < ms_score < 5.0 and 0.001 < ms_an < 5.0
    c.profile(False)
    # fnx_ctx_last_kernel: the route the dispatch took, not what an environment switch suggests
    assert c.last_kernel(fennec_amd.PROF_MAIN) in ("analyze_one_kernel", "analyze_pass_kernel")   # (Analyze shares the MAIN class)
    c.GaussianBlurBatch(d, 2.0); c.sync()
    assert c.last_kernel(fennec_amd.PROF_MAIN) in ("blur_mfma_kernel", "blur_direct_kernel")
    c.GaussianBlurSSIMFastBatch(d, 2.0)
    assert "SCORE" in c.last_kernel(fennec_amd.PROF_MAIN)
    c.GaussianBlur(d[0], 21.5); c.sync()                      # radius 65: beyond the matrix kernels (r5: they take up to 62)
    assert c.last_kernel(fennec_amd.PROF_MAIN).startswith("blur_pass_kernel")
    assert c.last_kernel(fennec_amd.PROF_RESIZE) == ""
    c.lanczosResize(d[0], 1920, 1080); c.sync()
    assert c.last_kernel(fennec_amd.PROF_RESIZE).startswith("resize_")
    with pytest.raises(fennec_amd.FennecError):
        c.last_kernel(3)
    c.close()


def test_full_size_properties(ctx):
    """size-independent properties at BASELINE.json's full sizes, no oracle involved"""
    import torch
    a = synth.large_photo(3840, 2160, 5)
    b = ctx.GaussianBlur(a, 2.0)
    # symmetry of the SSIM family (the separable kernel fuses a^2+b^2 one way round: last-bit differences)
    assert abs(ctx.SSIM(a, b) - ctx.SSIM(b, a)) <= 1e-12
    assert abs(ctx.SSIMFast(a, b) - ctx.SSIMFast(b, a)) <= 1e-12
    assert abs(ctx.MSSSIM(a, b) - ctx.MSSSIM(b, a)) <= 1e-12
    # orientation round trips and permutation property at 8K
    big = synth.large_photo(7680, 4320, 1)
    for o, inv in ((2, 2), (3, 3), (4, 4), (5, 5), (6, 8), (7, 7), (8, 6)):
        assert np.array_equal(ctx.ApplyOrientation(ctx.ApplyOrientation(big, o), inv), big)
    # constant images are fixed points of every filter; resizing to the same size is a copy
    solid = synth.make_solid_image(3840, 2160, (201, 17, 99, 255))
    for out in (ctx.GaussianBlur(solid, 2.0), ctx.GaussianBlur(solid, 2.0, exact=True), ctx.blur3x3(solid),
                ctx.Sharpen(solid, 0.8), ctx.AdaptiveSharpen(solid, 0.8), ctx.lanczosResize(solid, 3840, 2160)):
        assert np.array_equal(out, solid)
    assert np.array_equal(ctx.boxDownsample(solid, 512, 288), synth.make_solid_image(512, 288, (201, 17, 99, 255)))
    assert np.array_equal(ctx.lanczosResize(solid, 1920, 1080), synth.make_solid_image(1920, 1080, (201, 17, 99, 255)))
    # one-pass == its parts on a device batch (checksum of checksums)
    d = [torch.from_numpy(synth.large_photo(3840, 2160, k)).cuda() for k in range(4)]
    torch.cuda.synchronize()
    outs, ss = ctx.GaussianBlurSSIMFastBatch(d, 2.0)
    assert all(torch.equal(o, r) for o, r in zip(outs, ctx.GaussianBlurBatch(d, 2.0)))
    assert np.array_equal(ss, ctx.SSIMFastBatch(d, outs))


def test_device_source_host_destination(orc):
    """FNX_DEVICE_SRC: one resident source, image results straight into host memory -- scaleSearch's loop
    (targetsize.go:286-313: boxDownsample(src, w, h) at 12 bisection scales, each encoded on the host)."""
    import ctypes as C
    import torch
    lib = fennec_amd.load_library()
    c = fennec_amd.Context(0)
    img = synth.large_photo(1900, 1068, 6)
    d = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()
    lo, hi = 0.05, 1.0
    for i in range(12):
        mid = (lo + hi) / 2
        nw, nh = int(1900 * mid), int(1068 * mid)
        got = c.boxDownsample(d, nw, nh, to_host=True)
        assert isinstance(got, np.ndarray) and np.array_equal(got, orc.box_downsample(img, nw, nh)), (i, nw, nh)
        lo, hi = (mid, hi) if i % 3 else (lo, mid)
    assert np.array_equal(c.lanczosResize(d, 713, 401, to_host=True), orc.lanczos_resize(img, 713, 401))
    assert np.array_equal(c.lanczosResize(d, 1900, 1068, to_host=True), img)                  # same-size copy branch
    assert c.boxDownsample(d, 0, 5, to_host=True).shape == (0, 0, 4)
    # the other image -> image entry points, through the raw ABI
    SRC = fennec_amd.FNX_DEVICE_SRC
    small = synth.noise_image(301, 203, 8, alpha=True)
    ds = torch.from_numpy(small).cuda()
    torch.cuda.synchronize()
    h, w = small.shape[:2]
    out = np.empty_like(small)
    args = (ds.data_ptr(), 4 * w, w, h)
    dst = (out.ctypes.data, 4 * w)
    assert lib.fnx_blur3x3(c._h, SRC, *args, *dst) == 0 and np.array_equal(out, orc.blur3x3(small))
    assert lib.fennec_Sharpen(c._h, SRC, *args, 0.8, *dst) == 0 and np.array_equal(out, orc.sharpen(small, 0.8))
    assert lib.fennec_AdaptiveSharpen(c._h, SRC, *args, 0.8, *dst) == 0 and np.array_equal(out, orc.adaptive_sharpen(small, 0.8))
    r, k = c.blurKernel(1.3)
    assert lib.fnx_gaussian_blur(c._h, SRC, *args, k.ctypes.data_as(C.POINTER(C.c_double)), r, 1, *dst) == 0
    assert np.array_equal(out, orc.gaussian_blur(small, 1.3))
    rot = np.empty((w, h, 4), np.uint8)
    assert lib.fnx_orient(c._h, SRC, *args, 6, rot.ctypes.data, 4 * h) == 0 and np.array_equal(rot, orc.apply_orientation(small, 6))
    # scalar-result ops have no host destination: the space is rejected, not misread
    val = C.c_double()
    win = c.gaussianKernel()
    assert lib.fnx_ssim_fast(c._h, SRC, *args[:2], *args[:2], w, h, win.ctypes.data_as(C.POINTER(C.c_double)), C.byref(val)) < 0
    c.close()


def test_enqueue_ahead_fifo(ctx, orc):
    """Several batches enqueued on ONE ctx before any is fetched: fnx_results_fetch hands them back oldest
    first, each with its own scores; a fifth waiting batch and a blocking call over a non-empty FIFO are
    refused; the profile hook reads its launches in the same order."""
    import torch
    c = fennec_amd.Context(0)
    sets = [[synth.large_photo(2048, 1200, 10 * j + k) for k in range(3)] for j in range(4)]
    dev = [[torch.from_numpy(i).cuda() for i in st] for st in sets]
    torch.cuda.synchronize()
    plans = [c.plan_blur_ssim_fast_batch(d, 2.0) for d in dev]
    want = [c.GaussianBlurSSIMFastBatch(d, 2.0)[1] for d in dev]          # one at a time
    c.profile(True)
    for p in plans:
        p.enqueue()
    with pytest.raises(fennec_amd.FennecError):
        plans[0].enqueue()                                                # four batches are waiting
    with pytest.raises(fennec_amd.FennecError):
        c.GaussianBlurSSIMFastBatch(dev[0], 2.0)                          # blocking form over a non-empty FIFO
    for j, p in enumerate(plans):
        assert np.array_equal(p.fetch(), want[j]), j
        assert c.kernel_ms() > 0
    with pytest.raises(fennec_amd.FennecError):
        plans[0].fetch()                                                  # nothing left
    with pytest.raises(fennec_amd.FennecError):
        c.kernel_ms()
    # two-call flavour, interleaved with fetches
    outs = [c.GaussianBlurBatch(d, 2.0) for d in dev[:2]]
    ps = [c.plan_ssim_fast_batch(d, o) for d, o in zip(dev[:2], outs)]
    ps[0].enqueue(); ps[1].enqueue()
    a0 = ps[0].fetch().copy()
    ps[0].enqueue()
    a1 = ps[1].fetch().copy()
    a2 = ps[0].fetch().copy()
    assert np.array_equal(a0, a2) and np.array_equal(a0, want[0]) and np.array_equal(a1, want[1])
    assert abs(a1[2] - orc.ssim_fast(sets[1][2], outs[1][2].cpu().numpy(), procs=8)) <= SSIM_TOL
    c.close()


def test_argument_errors_and_context_lifecycle(orc):
    """bad arguments come back as FennecError with the library's message (never a crash, never a silent
    fallback); contexts can be created and destroyed in a loop without leaking device memory"""
    import ctypes as C
    import torch
    lib = fennec_amd.load_library()
    c = fennec_amd.Context(0)
    img = synth.large_photo(64, 48, 1)
    k = np.array([0.25, 0.5, 0.25])
    out = np.empty_like(img)
    # stride smaller than a row, null source, negative radius, translucent palette, mismatched batch
    assert lib.fnx_gaussian_blur(c._h, 0, img.ctypes.data, 4 * 64 - 4, 64, 48, k.ctypes.data_as(C.POINTER(C.c_double)), 1, 0,
                                 out.ctypes.data, 4 * 64) < 0
    assert b"stride" in lib.fnx_last_error() or b"invalid" in lib.fnx_last_error()
    assert lib.fnx_gaussian_blur(c._h, 0, None, 4 * 64, 64, 48, k.ctypes.data_as(C.POINTER(C.c_double)), 1, 0,
                                 out.ctypes.data, 4 * 64) < 0
    assert lib.fnx_gaussian_blur(c._h, 7, img.ctypes.data, 4 * 64, 64, 48, k.ctypes.data_as(C.POINTER(C.c_double)), 1, 0,
                                 out.ctypes.data, 4 * 64) < 0          # unknown space
    with pytest.raises(fennec_amd.FennecError):
        c.plan_ssim_fast_batch([torch.from_numpy(img).cuda()], [])
    with pytest.raises(fennec_amd.FennecError):
        c.GaussianBlurBatch([img], 2.0)                               # host arrays in a device batch
    # a failed call leaves the ctx usable
    assert np.array_equal(c.GaussianBlur(img, 2.0, exact=True), orc.gaussian_blur(img, 2.0))
    c.close()
    free0 = torch.cuda.mem_get_info()[0]
    big = synth.large_photo(1920, 1080, 2)
    for _ in range(12):
        cc = fennec_amd.Context(0)
        cc.GaussianBlur(big, 2.0)
        cc.SSIMFast(big, big)
        cc.Analyze(big)
        cc.close()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < 64 << 20            # nothing accumulates across contexts


@pytest.mark.gpu
def test_msssim_enqueue_fifo(ctx, orc):
    """fennec_MSSSIM_enqueue / fnx_msssim_enqueue: the same values as the blocking MSSSIM, through the result FIFO,
    several images ahead, img2 resized to img1's dims when they differ (ssim.go:320-322)."""
    import torch
    pairs = []
    for k, (w, h) in enumerate([(640, 480), (333, 217), (1024, 512), (40, 24), (640, 480)]):
        a = synth.large_photo(w, h, k)
        b = orc.gaussian_blur(a, 1.0 + 0.3 * k) if k != 4 else orc.lanczos_resize(a, w // 2, h // 2)
        pairs.append((torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()))
    want = [ctx.MSSSIM(a, b) for a, b in pairs]
    got = []
    for i, (a, b) in enumerate(pairs):
        ctx.msssim_enqueue(a, b)
        if i >= 3:
            got.append(ctx.fetch_result())
    while len(got) < len(pairs):
        got.append(ctx.fetch_result())
    assert got == want
    assert abs(want[0] - orc.msssim(pairs[0][0].cpu().numpy(), pairs[0][1].cpu().numpy())) <= 1e-12
    s0 = ctx.SSIM(*pairs[0])
    ctx.ssim_enqueue(*pairs[0])                      # mixed kinds keep their order
    ctx.msssim_enqueue(*pairs[1])
    assert ctx.fetch_result() == s0
    assert ctx.fetch_result() == want[1]


@pytest.mark.gpu
def test_lanczos_resize_mixed_content_sequence(ctx, orc):
    """One cached plan, content that takes different routes through the guard kernels from call to call: tie-dense
    (the reference's ramps at an integer ratio: every row goes to the exact loops), noise (the fp32 form decides nearly
    everything), translucent ramps (general arithmetic per output).  Every result is the oracle's."""
    import torch
    w, h, dw, dh = 1280, 720, 640, 360
    ramp = [synth.large_photo(w, h, k) for k in range(3)]
    noise = synth.noise_image(w, h, 11)
    trans = synth.large_photo(w, h, 5)
    trans[100:300, 200:900, 3] = 77
    seq = [ramp[0], ramp[1], ramp[2], noise, noise, ramp[0], trans, trans, ramp[1], noise, ramp[2]]
    want = {}
    for k, img in enumerate(seq):
        key = id(img)
        if key not in want:
            want[key] = orc.lanczos_resize(img, dw, dh, procs=8)
        got = ctx.lanczosResize(torch.from_numpy(img).cuda(), dw, dh).cpu().numpy()
        assert np.array_equal(got, want[key]), k


@pytest.mark.gpu
def test_ssim_two_column_march_against_oracle(ctx, orc):
    """windowed_ssim_march2_kernel (images of >= 4 M windows: two pixel columns per lane, strips of 121 window columns)
    against the oracle directly: widths on both sides of strip multiples, an odd width (the last lane's pair re-reads
    the last two columns), a 4-byte-aligned strided device view, a full 4K pair, and the same pair through the FIFO."""
    import torch
    rng_sizes = [(2309, 1801), (2428, 1713), (2429, 1712), (8200, 520), (3840, 2160)]
    for (w, h) in rng_sizes:
        assert (w - 8) * (h - 8) >= 4_000_000
        a = synth.large_photo(w, h, w % 7)
        b = ctx.AdaptiveSharpen(a, 0.5)
        want = orc.ssim(a, b, procs=64)
        assert abs(ctx.SSIM(a, b) - want) <= SSIM_TOL, (w, h)
        da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        assert ctx.SSIM(da, db) == ctx.SSIM(a, b)
        ctx.ssim_enqueue(da, db)
        assert ctx.fetch_result() == ctx.SSIM(a, b)
    # a view whose rows start 4 bytes off a 16-byte boundary, with padding behind every row
    w, h = 2400, 1740
    a = synth.noise_image(w + 4, h, 3)
    b = orc.gaussian_blur(a, 0.8, procs=16)
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    va, vb = da[:, 1:w + 1], db[:, 1:w + 1]
    want = orc.ssim(np.ascontiguousarray(a[:, 1:w + 1]), np.ascontiguousarray(b[:, 1:w + 1]), procs=64)
    assert abs(ctx.SSIM(va, vb) - want) <= SSIM_TOL


@pytest.mark.gpu
def test_batch_entry_points_refuse_bad_arguments(orc):
    """round 6's batch entry points through the raw C ABI: negative counts, null arrays, a null image among the images, null
    tap tables and a full result FIFO come back as errors with a message (never a crash); n = 0 is a no-op; the ctx stays usable"""
    import ctypes as C
    import torch
    lib = fennec_amd.load_library()
    c = fennec_amd.Context(0)
    W, H = 96, 64
    a = [torch.from_numpy(synth.large_photo(W, H, k)).cuda() for k in range(2)]
    b = [torch.from_numpy(synth.noise_image(W, H, 3 + k)).cuda() for k in range(2)]
    o = [torch.empty_like(x) for x in a]
    small = [torch.empty((H // 2, W // 2, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts])  # noqa: E731
    none = C.POINTER(C.c_void_p)()
    win = np.ascontiguousarray(orc.gaussian_kernel(8, 1.5), dtype=np.float64)
    f64 = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    i32 = lambda x: x.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
    pa, pb, po, ps = arr(a), arr(b), arr(o), arr(small)
    bad = []

    def refused(rc, what):
        if rc >= 0 or not lib.fnx_last_error():
            bad.append(what)

    # sharpen / AdaptiveSharpen batches
    refused(lib.fnx_sharpen_batch(c._h, -1, pa, 4 * W, W, H, 0.5, po, 4 * W), "sharpen n = -1")
    refused(lib.fnx_sharpen_batch(c._h, 2, none, 4 * W, W, H, 0.5, po, 4 * W), "sharpen null srcs")
    refused(lib.fnx_adaptive_sharpen_batch(c._h, 2, arr([a[0], None]), 4 * W, W, H, 0.5, po, 4 * W), "adaptive null image")
    refused(lib.fnx_sharpen_batch(c._h, 2, pa, 4 * W, W, H, 0.5, pa, 4 * W), "sharpen dst aliases src")
    refused(lib.fnx_sharpen_batch(c._h, 2, pa, 4 * W - 4, W, H, 0.5, po, 4 * W), "sharpen stride below a row")
    assert lib.fnx_sharpen_batch(c._h, 0, none, 4 * W, W, H, 0.5, none, 4 * W) == 0
    many = (C.c_void_p * 65536)(*([a[0].data_ptr()] * 65536))           # one more than FNX_BATCH_MAX: refused before anything is read
    refused(lib.fnx_sharpen_batch(c._h, 65536, many, 4 * W, W, H, 0.5, many, 4 * W), "sharpen n = 65536")
    refused(lib.fnx_ssim_batch_enqueue(c._h, 65536, many, 4 * W, many, 4 * W, W, H, f64(win)), "ssim n = 65536")
    # SSIM / MSSSIM batches
    refused(lib.fnx_ssim_batch_enqueue(c._h, -2, pa, 4 * W, pb, 4 * W, W, H, f64(win)), "ssim n = -2")
    refused(lib.fnx_ssim_batch_enqueue(c._h, 2, pa, 4 * W, none, 4 * W, W, H, f64(win)), "ssim null bs")
    refused(lib.fnx_ssim_batch_enqueue(c._h, 2, pa, 4 * W, pb, 4 * W, W, H, None), "ssim null window")
    refused(lib.fnx_msssim_batch_enqueue(c._h, 2, arr([None, a[1]]), 4 * W, pb, 4 * W, W, H, f64(win)), "msssim null image")
    refused(lib.fnx_msssim_batch_enqueue(c._h, 2, pa, 4 * W, pb, 4 * W, 0, H, f64(win)), "msssim w = 0")
    # lanczosResize batch: tables
    th, tv = orc.precompute_weights(W // 2, W), orc.precompute_weights(H // 2, H)
    offh, idxh, wh = (np.ascontiguousarray(th[0], np.int32), np.ascontiguousarray(th[1], np.int32), np.ascontiguousarray(th[2], np.float64))
    offv, idxv, wv = (np.ascontiguousarray(tv[0], np.int32), np.ascontiguousarray(tv[1], np.int32), np.ascontiguousarray(tv[2], np.float64))
    ok = lib.fnx_lanczos_resize_batch(c._h, 2, pa, 4 * W, W, H, i32(offh), i32(idxh), f64(wh), i32(offv), i32(idxv), f64(wv), ps, 2 * W, W // 2, H // 2)
    assert ok == 0, lib.fnx_last_error()
    c.sync()
    for k in range(2):
        assert np.array_equal(small[k].cpu().numpy(), orc.lanczos_resize(a[k].cpu().numpy(), W // 2, H // 2))
    refused(lib.fnx_lanczos_resize_batch(c._h, 2, pa, 4 * W, W, H, None, i32(idxh), f64(wh), i32(offv), i32(idxv), f64(wv), ps, 2 * W, W // 2, H // 2),
            "resize null offH")
    refused(lib.fnx_lanczos_resize_batch(c._h, 2, pa, 4 * W, W, H, i32(offh), i32(idxh), f64(wh), i32(offv), i32(idxv), None, ps, 2 * W, W // 2, H // 2),
            "resize null wV")
    refused(lib.fnx_lanczos_resize_batch(c._h, 2, pa, 4 * W, W, H, i32(offh), i32(idxh), f64(wh), i32(offv), i32(idxv), f64(wv), none, 2 * W, W // 2, H // 2),
            "resize null dsts")
    refused(lib.fnx_lanczos_resize_batch(c._h, -1, pa, 4 * W, W, H, i32(offh), i32(idxh), f64(wh), i32(offv), i32(idxv), f64(wv), ps, 2 * W, W // 2, H // 2),
            "resize n = -1")
    refused(lib.fnx_lanczos_resize(c._h, 1, a[0].data_ptr(), 4 * W, W, H, None, None, None, None, None, None, small[0].data_ptr(), 2 * W, W // 2, H // 2),
            "single resize null tables")
    assert not bad, bad
    # a full FIFO: the fifth unfetched batch is refused, the four before it are intact
    want = [orc.ssim(x.cpu().numpy(), y.cpu().numpy()) for x, y in zip(a, b)]
    for _ in range(4):
        assert lib.fnx_ssim_batch_enqueue(c._h, 2, pa, 4 * W, pb, 4 * W, W, H, f64(win)) == 0, lib.fnx_last_error()
    assert lib.fnx_ssim_batch_enqueue(c._h, 2, pa, 4 * W, pb, 4 * W, W, H, f64(win)) < 0
    assert b"fnx_results_fetch" in lib.fnx_last_error()
    for _ in range(4):
        got = c.fetch_results(2)
        assert np.allclose(got, want, rtol=0, atol=1e-12)
    # and the ctx still works
    assert np.array_equal(c.Sharpen(a[0], 0.5).cpu().numpy(), orc.sharpen(a[0].cpu().numpy(), 0.5))
    c.close()
