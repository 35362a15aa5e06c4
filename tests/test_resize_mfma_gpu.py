"""csrc/resize_mfma.hip (lanczosResize on the i8 matrix pipe: planar channels, fixed-point weights under a rounding guard,
fp64 fix-ups, regions it cannot take handed back to resize_fused_sparse_kernel) against the oracle, bit for bit.
The "resize_mfma" form = 2 (fnx_ctx_set_form) sends every table the kernel covers through it (the default keeps it to downscales); shapes on both sides
of its boundaries (64-column strips, 16-row groups and slots, ratios at the edge of the 64-px / 64-row windows, strips that
would stick out of the row), content it must hand back (translucent patches, tie-dense ramps and stripes, all of it or a
corner), pitched device views, and the cool-down after a call that was mostly handed back."""
import numpy as np
import pytest

import fennec_amd
from fennec_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def mctx():
    c = fennec_amd.Context(0)
    c.set_form("resize_mfma", 2)                     # read when a resize plan is built: a ctx of its own, plans of its own
    return c


def _opaque(img):
    out = img.copy()
    out[..., 3] = 255
    return out


def _photo(w, h, seed):
    """noise with structure: smooth gradients + edges + grain, opaque"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = 127 + 90 * np.sin(x / 37.0 + seed) * np.cos(y / 23.0) + 30 * ((x // 50 + y // 40) % 2)
    img = np.empty((h, w, 4), np.uint8)
    for c in range(3):
        img[..., c] = np.clip(base + 15 * c + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)
    img[..., 3] = 255
    return img


SHAPES = [(640, 480, 320, 240), (640, 480, 1280, 960), (641, 479, 301, 1000), (517, 389, 259, 195), (1000, 700, 460, 322),
          (1000, 700, 470, 330), (130, 90, 65, 45), (130, 90, 195, 135), (256, 64, 128, 32), (257, 65, 129, 33), (300, 1000, 150, 500),
          (2049, 70, 1025, 35), (1920, 1080, 3840, 2160), (3840, 2160, 1920, 1080), (800, 600, 800, 600), (640, 480, 400, 200),
          (64, 64, 32, 32), (79, 200, 40, 97), (1023, 33, 512, 17)]


@pytest.mark.parametrize("w,h,dw,dh", SHAPES)
def test_against_the_oracle(mctx, orc, w, h, dw, dh):
    big = w * h > 3000 * 2000
    for k, img in enumerate((_opaque(synth.noise_image(w, h, w + dw, alpha=True)), _photo(w, h, dh))):
        if big and k == 0:
            continue
        got = mctx.lanczosResize(img, dw, dh)
        assert np.array_equal(got, orc.lanczos_resize(img, dw, dh, procs=32 if big else 8)), (k, w, h, dw, dh)


@pytest.mark.parametrize("w,h,dw,dh", [(640, 480, 320, 240), (1000, 700, 470, 330), (517, 389, 1034, 778), (1280, 720, 640, 360)])
def test_what_it_hands_back(mctx, orc, w, h, dw, dh):
    """translucent patches (part of the image goes back), every window translucent, a ramp and two-level stripes (exact ties
    everywhere: all of it goes back), a ramp in one corner of a photograph"""
    noise = _opaque(synth.noise_image(w, h, 7, alpha=True))
    holes = noise.copy()
    holes[h // 3: h // 2 + 1, w // 4: w // 2 + 1, 3] = 17
    one_px = noise.copy()
    one_px[h - 1, w - 1, 3] = 254
    soft = synth.noise_image(w, h, 11, alpha=True)
    ramp = synth.large_photo(w, h, 3)
    stripes = np.empty((h, w, 4), np.uint8)
    stripes[:, 0::2] = (100, 7, 250, 255)
    stripes[:, 1::2] = (101, 8, 255, 255)
    stripes[1::2, :, :3] += 1
    corner = _photo(w, h, 5)
    corner[: h // 2, : w // 2] = ramp[: h // 2, : w // 2]
    for k, img in enumerate((holes, one_px, soft, ramp, stripes, corner)):
        assert np.array_equal(mctx.lanczosResize(img, dw, dh), orc.lanczos_resize(img, dw, dh, procs=8)), (k, w, h, dw, dh)


def test_pitched_device_views(mctx, orc):
    import torch
    big = torch.from_numpy(_opaque(synth.noise_image(1500, 900, 77, alpha=True))).cuda()
    torch.cuda.synchronize()
    for (y0, x0, hh, ww, dw, dh) in ((0, 0, 900, 1500, 750, 450), (7, 13, 600, 1000, 520, 312), (100, 1, 333, 1499, 700, 160), (0, 1372, 900, 128, 64, 450)):
        view = big[y0:y0 + hh, x0:x0 + ww]
        host = np.ascontiguousarray(view.cpu().numpy())
        assert np.array_equal(mctx.lanczosResize(view, dw, dh).cpu().numpy(), orc.lanczos_resize(host, dw, dh, procs=8)), (y0, x0, hh, ww)


def test_default_policy_and_cool_down(orc):
    """the default ctx: a downscale of a ramp is handed back whole, the next calls skip the matrix kernel (a heuristic about
    cost only) -- and a photograph in between, and an upscale (not this kernel's by default), all equal the oracle"""
    ctx = fennec_amd.Context(0)
    w, h, dw, dh = 1280, 960, 640, 480
    ramp, photo = synth.large_photo(w, h, 1), _photo(w, h, 2)
    want_r, want_p = orc.lanczos_resize(ramp, dw, dh, procs=8), orc.lanczos_resize(photo, dw, dh, procs=8)
    for k in range(6):
        assert np.array_equal(ctx.lanczosResize(ramp, dw, dh), want_r), k
        ctx.sync()
        if k % 2:
            assert np.array_equal(ctx.lanczosResize(photo, dw, dh), want_p), k
    assert np.array_equal(ctx.lanczosResize(photo, 2 * w, 2 * h), orc.lanczos_resize(photo, 2 * w, 2 * h, procs=8))


def test_switched_off(orc):
    ctx = fennec_amd.Context(0)
    ctx.set_form("resize_mfma", 0)
    img = _photo(640, 480, 9)
    assert np.array_equal(ctx.lanczosResize(img, 320, 240), orc.lanczos_resize(img, 320, 240, procs=8))
