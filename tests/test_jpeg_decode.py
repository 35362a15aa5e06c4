"""SURVEY 8(f)2, third slice: image.Decode of a JPEG source on the device (jpeg_dec.hip, fnx_jpeg_decode,
fnx_jpeg_recompress).

CPU (not gpu): the checker itself -- the oracle's decoder reads what libjpeg-turbo writes (standard and optimised
Huffman tables, 4:2:0 and 4:4:4) and lands within the IDCT's tolerance of libjpeg's own decode; it reads its own
encoder's files back to exactly the round trip's pixels.

GPU (-m gpu): the device decoder against the oracle's, bit for bit: the oracle's files and libjpeg's, every MCU
geometry, flat images (two symbols per block: many blocks per span) and noise at quality 100 (long codes, spans that
hold less than a block), files large enough for several workgroups and several synchronisation rounds; what it must
refuse (FNX_ERR_UNSUPPORTED) and what it must call corrupt; fnx_jpeg_recompress = decode + fnx_jpeg_compress.
"""
import io
import os

import numpy as np
import pytest

from fennec_amd import synth
from oracle import oracle as orc


def _pil(img, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(img[..., :3]), "RGB").save(buf, "JPEG", **kw)
    return buf.getvalue()


def _pil_decode(data):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def _pil_grey(img, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(img[..., 1]), "L").save(buf, "JPEG", **kw)
    return buf.getvalue()


def _as_440(data):
    """A 4:2:2 file relabelled as 4:4:0: the luminance factors 2x1 -> 1x2 and width <-> height in the frame header.  The
    scan stays a valid bit string (two Y blocks, Cb, Cr per MCU, the same number of MCUs); only the geometry changes."""
    i = data.index(b"\xff\xc0")
    b = bytearray(data)
    assert b[i + 9] == 3 and b[i + 11] == 0x21
    b[i + 5:i + 9] = b[i + 7:i + 9] + b[i + 5:i + 7]
    b[i + 11] = 0x12
    return bytes(b)


def _with_luma_factors(data, hv):
    """the frame header's luminance sampling factors overwritten (for the parser: the scan no longer fits them)"""
    i = data.index(b"\xff\xc0")
    b = bytearray(data)
    b[i + 11] = hv
    return bytes(b)


def _noise(w, h, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    a[..., 3] = 255
    return a


def _photo(w, h, seed=0):
    """smooth structure + texture: what a camera file's statistics look like"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = 128 + 60 * np.sin(x / 37.0) * np.cos(y / 23.0) + 40 * np.sin((x + y) / 91.0)
    img = np.stack([base + rng.normal(0, s, (h, w)) for s in (6, 9, 12)], axis=-1)
    out = np.empty((h, w, 4), dtype=np.uint8)
    out[..., :3] = np.clip(img, 0, 255).astype(np.uint8)
    out[..., 3] = 255
    return out


# ------------------------------------------------------------------------------------------------ CPU: the checker
def test_oracle_decoder_reads_422_440_and_grey_files():
    src = _photo(203, 117, 3)
    d422 = _pil(src, quality=85, subsampling=1)
    got = orc.jpeg_decode(d422)
    d = np.abs(got[..., :3].astype(int) - _pil_decode(d422).astype(int))
    assert got.shape == (117, 203, 4) and d.mean() < 1.2 and np.percentile(d, 99) <= 10          # (libjpeg smooths chroma on the way up)
    w, h, ratio, y, cb, cr = orc.jpeg_decode_planes(d422)
    assert ratio == 1 and y.shape == (120, 208) and cb.shape == (120, 104)
    w, h, ratio, y, cb, cr = orc.jpeg_decode_planes(_as_440(d422))
    assert (w, h, ratio) == (117, 203, 3) and y.shape == (208, 120) and cb.shape == (104, 120)
    grey = _pil_grey(src, quality=85)
    got = orc.jpeg_decode(grey)
    from PIL import Image
    ref = np.asarray(Image.open(io.BytesIO(grey)).convert("L"))
    assert (got[..., 0] == got[..., 1]).all() and (got[..., 1] == got[..., 2]).all() and (got[..., 3] == 255).all()
    assert np.abs(got[..., 0].astype(int) - ref.astype(int)).max() <= 2            # islow IDCT vs idct.go's
    assert orc.jpeg_decode_planes(grey)[2] == -1


def test_oracle_decoder_reads_411_and_410_files():
    """luminance 4 x 1 and 4 x 2 (image.YCbCrSubsampleRatio411 / 410; reader.go takes h in {1, 2, 4}, v in {1, 2}): files from
    tests/jpeg_mini.py, checked against libjpeg, which replicates such chroma as Go does (no smoothing: within the IDCTs'
    and the colour conversions' rounding)"""
    import jpeg_mini
    src = _photo(203, 117, 3)
    for (hy, vy, ratio) in ((4, 1, 4), (4, 2, 5)):
        for rst in (0, 5):
            data = jpeg_mini.encode(src, hy, vy, 85, rst)
            got = orc.jpeg_decode(data)
            d = np.abs(got[..., :3].astype(int) - _pil_decode(data).astype(int))
            assert got.shape == (117, 203, 4) and d.mean() < 0.3 and d.max() <= 4, (hy, vy, rst, d.mean(), d.max())
            w, h, r, y, cb, cr = orc.jpeg_decode_planes(data)
            assert (w, h, r) == (203, 117, ratio) and y.shape == (8 * vy * ((117 + 8 * vy - 1) // (8 * vy)), 224) and cb.shape == (y.shape[0] // vy, 56)
    # and the factors 1 and 2 from the same writer agree with what libjpeg wrote itself being decoded by the same code path
    for (hy, vy) in ((1, 1), (2, 1), (2, 2), (1, 2)):
        data = jpeg_mini.encode(src, hy, vy, 90)
        d = np.abs(orc.jpeg_decode(data)[..., :3].astype(int) - _pil_decode(data).astype(int))
        assert d.mean() < 1.3 and np.percentile(d, 99) <= 10


@pytest.mark.parametrize("kw", [dict(quality=85, subsampling=2), dict(quality=60, subsampling=0), dict(quality=92, subsampling=2, optimize=True)])
def test_oracle_decoder_reads_libjpeg_files(kw):
    src = _photo(203, 117, 3)
    data = _pil(src, **kw)
    got = orc.jpeg_decode(data)
    ref = _pil_decode(data)
    assert got.shape == (117, 203, 4) and (got[..., 3] == 255).all()
    # libjpeg: islow IDCT, fancy chroma upsampling for 4:2:0, its own colour constants -- close, not identical
    d = np.abs(got[..., :3].astype(int) - ref.astype(int))
    assert d.mean() < (1.6 if kw["subsampling"] == 2 else 0.6), d.mean()
    assert np.percentile(d, 99) <= (12 if kw["subsampling"] == 2 else 3)


def test_oracle_decoder_restart_intervals_give_the_same_pixels():
    src = _photo(203, 117, 3)
    for sub in (0, 1, 2):
        want = orc.jpeg_decode(_pil(src, quality=85, subsampling=sub))
        for kw in (dict(restart_marker_blocks=1), dict(restart_marker_blocks=5), dict(restart_marker_rows=1)):
            data = _pil(src, quality=85, subsampling=sub, **kw)
            assert b"\xff\xdd" in data and np.array_equal(orc.jpeg_decode(data), want), (sub, kw)


def test_oracle_decoder_inverts_its_encoder_exactly():
    for (w, h, q) in [(16, 16, 50), (33, 19, 90), (1, 1, 75), (120, 64, 100)]:
        src = _photo(w, h, w)
        assert np.array_equal(orc.jpeg_decode(orc.jpeg_encode(src, q)), orc.jpeg_roundtrip(src, q)), (w, h, q)


def _mutations(data, rng, n, lo, hi):
    """n copies of `data` with 1-4 random bytes in [lo, hi) replaced"""
    out = []
    for _ in range(n):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 5))):
            b[int(rng.integers(lo, min(hi, len(b))))] = int(rng.integers(0, 256))
        out.append(bytes(b))
    return out


def test_segment_parser_answers_or_refuses_never_crashes():
    """fnx_jpeg_decode(dst = NULL) is host code and needs no device: what it says about good files, files it must refuse,
    every truncation of a header and a few thousand random mutations of one (it reads untrusted bytes)."""
    import fennec_amd
    from PIL import Image
    parse = fennec_amd.Context.jpeg_parse
    src = _photo(96, 64, 1)
    good = [_pil(src, quality=80, subsampling=2), _pil(src, quality=80, subsampling=0, optimize=True), orc.jpeg_encode(src, 70)]
    for g in good:
        assert parse(g) == (96, 64)
    assert parse(_pil(src, quality=80, subsampling=1)) == (96, 64) and parse(_pil_grey(src, quality=80)) == (96, 64)
    assert parse(_as_440(_pil(src, quality=80, subsampling=1))) == (64, 96)
    assert parse(_pil(src, quality=80, restart_marker_blocks=2)) == (96, 64)
    assert parse(_pil(src, quality=80, progressive=True)) == (96, 64)          # r5: progressive frames are taken (jpeg_prog.cpp)
    for hv in (0x41, 0x42):                                   # 4:1:1, 4:1:0: taken since r4 (the geometry is the header's)
        assert parse(_with_luma_factors(good[0], hv)) == (96, 64)
    for hv in (0x14, 0x31, 0x44, 0x24):                       # four down (image/jpeg refuses it too), three across
        with pytest.raises(fennec_amd.FennecUnsupported):
            parse(_with_luma_factors(good[0], hv))
    buf = io.BytesIO()
    Image.fromarray(src, "RGBA").convert("CMYK").save(buf, "JPEG")
    cmyk = buf.getvalue()
    assert parse(cmyk) == (96, 64)                              # r5: four components with an Adobe segment are taken (the host's scans)
    i = cmyk.index(b"\xff\xee")
    with pytest.raises(fennec_amd.FennecUnsupported):          # ... without one image/jpeg refuses the file, and so does this parser
        parse(cmyk[:i] + cmyk[i + 2 + ((cmyk[i + 2] << 8) | cmyk[i + 3]):])
    i = good[0].index(b"\xff\xc0")
    with pytest.raises(fennec_amd.FennecUnsupported):          # 12-bit samples
        parse(good[0][:i + 4] + b"\x0c" + good[0][i + 5:])
    for junk in (b"", b"\xff", b"\xff\xd8", b"\xff\xd8\xff", b"GIF89a" + good[0], b"\xff\xd8\xff\xd9", b"\xff\xd8" + b"\xff" * 64):
        with pytest.raises(fennec_amd.FennecError):
            parse(junk)
    rng = np.random.default_rng(7)
    answered = refused = 0
    for g in good:
        head = g.index(b"\xff\xda") + 14
        cases = [g[:k] for k in range(0, head + 4)] + _mutations(g, rng, 1500, 2, head)
        for c in cases:
            try:
                w, h = parse(c)
                assert 0 < w <= 65535 and 0 < h <= 65535
                answered += 1
            except fennec_amd.FennecError:
                refused += 1
    assert answered > 100 and refused > 100


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def ctx():
    import fennec_amd
    return fennec_amd.Context(0)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(16, 16), (17, 9), (1, 1), (8, 8), (640, 480), (333, 217), (1000, 37), (8, 300), (1920, 1080)])
def test_gpu_decode_of_the_oracles_files(ctx, w, h):
    src = _photo(w, h, w + h)
    for q in (10, 75, 100):
        data = orc.jpeg_encode(src, q)
        assert ctx.jpeg_decode_config(data) == (w, h)
        got = ctx.jpeg_decode(data)
        assert np.array_equal(got, orc.jpeg_decode(data)), (w, h, q)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(quality=85, subsampling=2), dict(quality=60, subsampling=0), dict(quality=92, subsampling=2, optimize=True),
                                dict(quality=30, subsampling=0, optimize=True), dict(quality=100, subsampling=0), dict(quality=1, subsampling=2)])
def test_gpu_decode_of_libjpeg_files(ctx, kw):
    for (w, h) in [(203, 117), (64, 64), (1280, 720), (15, 33)]:
        data = _pil(_photo(w, h, 5), **kw)
        assert np.array_equal(ctx.jpeg_decode(data), orc.jpeg_decode(data)), (w, h, kw)


@pytest.mark.gpu
def test_gpu_decode_422_440_and_grey(ctx):
    for (w, h) in [(203, 117), (16, 8), (17, 9), (1, 1), (1280, 720), (3840, 2160)]:
        src = _photo(w, h, w)
        for q in (35, 90):
            d422 = _pil(src, quality=q, subsampling=1, optimize=q < 50)
            for data in (d422, _as_440(d422), _pil_grey(src, quality=q, optimize=q < 50)):
                assert np.array_equal(ctx.jpeg_decode(data), orc.jpeg_decode(data)), (w, h, q)
    # the whole item for such sources: decode + compress
    data = _pil(_photo(640, 480, 2), quality=93, subsampling=1)
    out, q, s, steps, dims = ctx.jpeg_recompress(data, 0.94)
    assert (out, q, s, steps) == ctx.jpeg_compress(orc.jpeg_decode(data), 0.94) and dims == (640, 480)


@pytest.mark.gpu
def test_gpu_decode_411_and_410(ctx):
    """ten blocks per MCU at 4:1:0: the slot tables, the block placement and the colour conversion at a 4:1 chroma step,
    with and without restart intervals, odd sizes, and through the recompress entry"""
    import jpeg_mini
    for (w, h, seed) in ((203, 117, 3), (64, 16, 1), (33, 9, 2), (640, 480, 4)):
        src = _photo(w, h, seed)
        for (hy, vy) in ((4, 1), (4, 2)):
            for rst in (0, 1, 7):
                data = jpeg_mini.encode(src, hy, vy, 88, rst)
                want = orc.jpeg_decode(data)
                assert np.array_equal(ctx.jpeg_decode(data), want), (w, h, hy, vy, rst)
                assert ctx.jpeg_decode_config(data) == (w, h)
    src = _photo(320, 200, 6)
    for (hy, vy) in ((4, 1), (4, 2)):
        data = jpeg_mini.encode(src, hy, vy, 91)
        out, q, sc, steps, dims = ctx.jpeg_recompress(data, 0.94)
        assert (out, q, sc, steps) == ctx.jpeg_compress(orc.jpeg_decode(data), 0.94) and dims == (320, 200)


@pytest.mark.gpu
def test_gpu_decode_with_restart_intervals(ctx):
    """DRI: every interval starts byte-aligned in a known state with the DC predictions at 0 -- intervals of one MCU, of a
    few, of an MCU row, of more MCUs than the image has; every subsampling and grey; the same pixels as the file
    without restart markers (the checker says so on the CPU, here the device must agree with the checker)."""
    for (w, h) in [(203, 117), (16, 16), (1280, 720), (3840, 2160)]:
        src = _photo(w, h, w)
        plain = {}
        for kw in (dict(restart_marker_blocks=1), dict(restart_marker_blocks=7), dict(restart_marker_rows=1), dict(restart_marker_blocks=60000)):     # (<= 65535: what a DRI segment can say)
            if w > 2000 and kw.get("restart_marker_blocks") == 1:
                continue
            for sub in (0, 1, 2, "L"):
                data = _pil_grey(src, quality=88, **kw) if sub == "L" else _pil(src, quality=88, subsampling=sub, **kw)
                got = ctx.jpeg_decode(data)
                assert np.array_equal(got, orc.jpeg_decode(data)), (w, h, kw, sub)
                if sub not in plain:
                    plain[sub] = ctx.jpeg_decode(_pil_grey(src, quality=88) if sub == "L" else _pil(src, quality=88, subsampling=sub))
                assert np.array_equal(got, plain[sub]), (w, h, kw, sub)
    data = _pil(_photo(640, 480, 2), quality=93, subsampling=2, restart_marker_rows=1)
    out, q, s, steps, dims = ctx.jpeg_recompress(data, 0.94)
    assert (out, q, s, steps) == ctx.jpeg_compress(orc.jpeg_decode(data), 0.94)
    # tests/golden/damaged_restart_interval.jpg (the fuzzer's find, seed 91): the last byte of an interval damaged so that
    # its tail reads as one block more.  A decoder that counts MCUs (the checker, Go, libjpeg) never looks at those bits;
    # the device decoder goes by position.  Round 2 made that an error (block accounting per interval); since round 3 the
    # write pass counts blocks from the interval's start (see the phantom-block fixture below) and the file decodes to the
    # checker's pixels
    import os
    import fennec_amd
    from fennec_amd import batch
    bad = open(os.path.join(os.path.dirname(__file__), "golden", "damaged_restart_interval.jpg"), "rb").read()
    assert orc.jpeg_decode(bad).shape == (312, 280, 4)
    assert np.array_equal(ctx.jpeg_decode(bad), orc.jpeg_decode(bad))
    r = batch.jpeg_item_work_device_all([bad], 0.94)(0, ctx)
    assert not r.host_decoded and r.Err is None and r.data[:2] == b"\xff\xd8"
    # an interval that is SHORT of blocks (bytes cut out in front of a marker): the accounting's error, and the harness's
    # host codec takes the file
    good = _pil(_photo(280, 312, 5), quality=88, subsampling=2, restart_marker_blocks=7)
    marks = [i for i in range(len(good) - 1) if good[i] == 0xff and 0xd0 <= good[i + 1] <= 0xd7]
    cut = good[:marks[3] - 24] + good[marks[3]:]
    with pytest.raises(fennec_amd.FennecError):
        ctx.jpeg_decode(cut)
    r = batch.jpeg_item_work_device_all([cut], 0.94)(0, ctx)
    assert (r.host_decoded and r.Err is None and r.data[:2] == b"\xff\xd8") or r.Err is not None      # (Pillow may refuse it too)
    # tests/golden/damaged_interval_ends_early.jpg (the fuzzer's find, seed 41, r3): a damaged byte makes an interval's 11
    # blocks end early and the leftover bits read as the START of a twelfth -- incomplete, so the block accounting above
    # holds, but its coefficients used to stay in the next interval's first block.  Once an interval's blocks are complete
    # the write pass now goes to the boundary, as a decoder that counts MCUs does: the same pixels as the checker.
    bad = open(os.path.join(os.path.dirname(__file__), "golden", "damaged_interval_ends_early.jpg"), "rb").read()
    assert np.array_equal(ctx.jpeg_decode(bad), orc.jpeg_decode(bad))
    # tests/golden/damaged_interval_phantom_block.jpg (seed 77, r3): the same with the leftover bits reading as one more
    # COMPLETE block (DC category 0 + EOB is 4 bits) -- the position-based sync passes count it, and every block number
    # behind it used to be one too high: each later interval lost its last block and was written one block late, with no
    # error.  The write pass now counts blocks from the interval's start (boundary records of the sync passes).
    bad = open(os.path.join(os.path.dirname(__file__), "golden", "damaged_interval_phantom_block.jpg"), "rb").read()
    assert np.array_equal(ctx.jpeg_decode(bad), orc.jpeg_decode(bad))


@pytest.mark.gpu
def test_gpu_decode_flat_and_noise_extremes(ctx):
    flat = np.full((1024, 2048, 4), 255, dtype=np.uint8)
    flat[..., 1] = 77
    for img, q in [(flat, 50), (_noise(1536, 1024, 1), 100), (_noise(512, 512, 2), 1)]:
        # (libjpeg's optimising pass cannot take noise at quality 100 through Pillow's buffer: standard tables there)
        for data in (orc.jpeg_encode(img, q), _pil(img, quality=q, subsampling=2), _pil(img, quality=q, subsampling=0, optimize=q < 100)):
            assert np.array_equal(ctx.jpeg_decode(data), orc.jpeg_decode(data))


@pytest.mark.gpu
def test_gpu_decode_4k_many_workgroups_and_device_output(ctx):
    import torch
    src = synth.make_test_image(3840, 2160)
    for data in (_pil(src, quality=90, subsampling=2), orc.jpeg_encode(_photo(3840, 2160, 9), 95)):
        want = orc.jpeg_decode(data)
        assert np.array_equal(ctx.jpeg_decode(data), want)
        t = ctx.jpeg_decode(data, device=True)
        assert t.is_cuda and torch.equal(t.cpu(), torch.from_numpy(want))
    # back-to-back files of different geometry on one ctx: nothing of the previous decode leaks into the next
    a, b = _pil(_photo(640, 360, 1), quality=70, subsampling=0), orc.jpeg_encode(_photo(97, 61, 2), 88)
    for data in (a, b, a, b):
        assert np.array_equal(ctx.jpeg_decode(data), orc.jpeg_decode(data))


@pytest.mark.gpu
def test_gpu_decode_refuses_what_it_does_not_handle(ctx):
    import fennec_amd
    from PIL import Image
    src = _photo(160, 120, 4)
    # progressive files are decoded since r5 -- bar a restart interval over one-component scans of a component with several
    # blocks per MCU (image/jpeg counts frame MCUs there, T.81 the scan's own: the host codec's call)
    with pytest.raises(fennec_amd.FennecUnsupported):
        ctx.jpeg_decode(_pil(src, quality=80, progressive=True, subsampling=2, restart_marker_blocks=3))
    # restart markers: one missing, two swapped
    rs = _pil(src, quality=80, subsampling=2, restart_marker_blocks=3)
    i0, i1 = rs.index(b"\xff\xd0"), rs.index(b"\xff\xd1")
    for bad in (rs[:i0] + rs[i0 + 2:], rs[:i0] + b"\xff\xd1" + rs[i0 + 2:i1] + b"\xff\xd0" + rs[i1 + 2:]):
        with pytest.raises(fennec_amd.FennecError) as e:
            ctx.jpeg_decode(bad)
        assert not isinstance(e.value, fennec_amd.FennecUnsupported)
    for hv in (0x31, 0x14, 0x44):                      # three across; four down (image/jpeg refuses that too)
        with pytest.raises(fennec_amd.FennecUnsupported):
            ctx.jpeg_decode(_with_luma_factors(_pil(src, quality=80, subsampling=2), hv))
    buf = io.BytesIO()
    Image.fromarray(src, "RGBA").convert("CMYK").save(buf, "JPEG", quality=80)
    cmyk = buf.getvalue()
    i = cmyk.index(b"\xff\xee")
    with pytest.raises(fennec_amd.FennecUnsupported):          # four components without an Adobe segment (r5: with one they decode)
        ctx.jpeg_decode_config(cmyk[:i] + cmyk[i + 2 + ((cmyk[i + 2] << 8) | cmyk[i + 3]):])
    good = _pil(src, quality=80, subsampling=2)
    for bad in (good[: len(good) // 2], good[:-2], b"\x89PNG\r\n\x1a\n" + good, good[:300]):
        with pytest.raises(fennec_amd.FennecError) as e:
            ctx.jpeg_decode(bad)
        assert not isinstance(e.value, fennec_amd.FennecUnsupported)
    # a frame header that claims 65 535 x 65 535 over the same small scan: refused before anything is sized by it
    i = good.index(b"\xff\xc0")
    huge = good[:i + 5] + b"\xff\xff\xff\xff" + good[i + 9:]
    assert ctx.jpeg_decode_config(huge) == (65535, 65535)
    with pytest.raises(fennec_amd.FennecError, match="too short"):
        ctx.jpeg_recompress(huge, 0.94)
    # a scan cut short but closed with an EOI: the blocks run out
    cut = good[: len(good) * 3 // 4] + b"\xff\xd9"
    with pytest.raises(fennec_amd.FennecError):
        ctx.jpeg_decode(cut)
    # and the ctx still works
    assert np.array_equal(ctx.jpeg_decode(good), orc.jpeg_decode(good))


@pytest.mark.gpu
def test_gpu_recompress_is_decode_plus_compress(ctx):
    for (w, h, kw) in [(640, 480, dict(quality=95, subsampling=2)), (333, 217, dict(quality=90, subsampling=0)), (1920, 1080, dict(quality=92, subsampling=2))]:
        data = _pil(_photo(w, h, 7), **kw)
        out, q, s, steps, dims = ctx.jpeg_recompress(data, 0.94)
        assert dims == (w, h)
        out2, q2, s2, steps2 = ctx.jpeg_compress(orc.jpeg_decode(data), 0.94)
        assert (q, steps) == (q2, steps2) and s == s2 and out == out2
        assert out[:2] == b"\xff\xd8" and out[-2:] == b"\xff\xd9"
    # r3: the item never makes the decoded NRGBA image (the encoder's colour conversion and the reference plane read the
    # decoded planes): every subsampling, odd sizes whose last chunk / MCU sticks out, grey (which still takes the image)
    for (w, h) in [(1283, 719), (203, 117), (600, 258)]:
        src = _photo(w, h, w + 1)
        d422 = _pil(src, quality=91, subsampling=1)
        for data in (d422, _as_440(d422), _pil(src, quality=91, subsampling=2), _pil(src, quality=91, subsampling=0), _pil_grey(src, quality=91)):
            out, q, s, steps, dims = ctx.jpeg_recompress(data, 0.94)
            dec = orc.jpeg_decode(data)                                  # (the relabelled 4:4:0 file is h x w)
            assert (out, q, s, steps) == ctx.jpeg_compress(dec, 0.94) and dims == (dec.shape[1], dec.shape[0])


@pytest.mark.gpu
def test_gpu_native_pool_over_jpeg_files(ctx):
    """fennec_CompressBatchJPEG: the C++ pool with fnx_jpeg_recompress per item, against the per-item calls; a file the device
    does not take (a progressive 4:2:0 file with restart intervals: tests/test_jpeg_progressive.py) comes back FNX_ERR_UNSUPPORTED
    and the python caller's host decode takes it from there."""
    from fennec_amd import batch
    files = [_pil(_photo(640 + 16 * k, 480 - 8 * k, k), quality=95 - k, subsampling=2 if k % 2 else 0) for k in range(9)]
    files.append(orc.jpeg_encode(_photo(333, 217, 3), 97))
    files.insert(4, _pil(_photo(320, 200, 11), quality=90, progressive=True, subsampling=2, restart_marker_blocks=4))
    res, outs, summ = batch.compress_batch_jpeg_native(files, 0.94, workers=4)
    assert [r.Index for r in res] == list(range(len(files))) and all(r.Err is None for r in res)
    assert [r.host_decoded for r in res] == [i == 4 for i in range(len(files))]
    for i, (r, f) in enumerate(zip(res, outs)):
        src = batch.pillow_decode(files[i]) if i == 4 else orc.jpeg_decode(files[i])
        data, q, s, n = ctx.jpeg_compress(src, 0.94)
        assert (r.Quality, r.SSIM, r.steps, r.CompressedSize, r.OriginalSize) == (q, s, n, len(data), len(files[i])) and f == data
    want = batch.summarize_local(res)
    assert (summ.Total, summ.Succeeded, summ.Failed, summ.TotalSaved, summ.AvgSSIM) == (want.Total, want.Succeeded, want.Failed, want.TotalSaved, want.AvgSSIM)
    assert batch.compress_batch_jpeg_native([]) == ([], [], batch.BatchSummary())


@pytest.mark.gpu
def test_gpu_decode_of_damaged_scans_terminates_and_agrees_with_the_checker(ctx):
    """Random bytes of the scan replaced: the device either decodes -- then to the checker's pixels whenever the checker
    decodes too -- or reports a corrupt file; it never hangs, and the ctx keeps working."""
    import fennec_amd
    rng = np.random.default_rng(11)
    src = _photo(200, 136, 2)
    both = dev_only = chk_only = neither = 0
    for g in (_pil(src, quality=85, subsampling=2), orc.jpeg_encode(src, 60), _pil(src, quality=95, subsampling=0, optimize=True)):
        scan = g.index(b"\xff\xda") + 14
        for c in _mutations(g, rng, 120, scan, len(g) - 2):
            try:
                want = orc.jpeg_decode(c)
            except Exception:
                want = None
            try:
                got = ctx.jpeg_decode(c)
            except fennec_amd.FennecError:
                got = None
            if got is not None and want is not None:
                assert np.array_equal(got, want)
                both += 1
            elif got is not None:
                dev_only += 1
            elif want is not None:
                chk_only += 1
            else:
                neither += 1
        assert np.array_equal(ctx.jpeg_decode(g), orc.jpeg_decode(g))
    print(f"damaged scans: both decode {both}, device only {dev_only}, checker only {chk_only}, neither {neither}")
    assert both > 50


def _analyze_format_is_png(img):
    """analyzeFormat (convert.go:105-146): True when it answers PNG"""
    h, w = img.shape[:2]
    total = w * h
    step = total // 10000 if total > 10000 else 1
    flat = img.reshape(-1, 4)[::step]
    seen, alpha = set(), False
    for px in flat:
        if len(seen) >= 512:
            break
        alpha = alpha or px[3] < 255
        seen.add(px.tobytes())
    return alpha or len(seen) < 256


@pytest.mark.gpu
def test_gpu_compress_file_jpeg_is_the_reference_pipeline(ctx):
    """fennec_CompressFileJPEG = CompressFile's JPEG path (fennec.go:30-205): decode, ApplyOrientation, smartResize,
    analyzeFormat, compressJPEGOptimal -- against the same stages composed from the oracle."""
    src = _photo(640, 400, 5)
    data = _pil(src, quality=93, subsampling=2)
    dec = orc.jpeg_decode(data)
    for orient, mw, mh in [(1, 0, 0), (6, 0, 0), (3, 320, 0), (8, 0, 150), (5, 200, 200), (1, 4000, 4000), (2, 0, 0)]:
        want = orc.apply_orientation(dec, orient) if orient > 1 else dec
        odims = (want.shape[1], want.shape[0])
        if mw > 0 or mh > 0:
            want = orc.smart_resize(want, mw, mh)
        assert not _analyze_format_is_png(want)
        out, q, s, steps, od, fd = ctx.compress_file_jpeg(data, 0.94, orient=orient, max_w=mw, max_h=mh, auto_format=True)
        assert od == odims and fd == (want.shape[1], want.shape[0]), (orient, mw, mh)
        assert (out, q, s, steps) == ctx.jpeg_compress(want, 0.94), (orient, mw, mh)
    # Format Auto on a source of few colours: PNG is the reference's answer, the call says so and leaves the item to the caller
    flat = np.zeros((240, 320, 4), dtype=np.uint8)
    flat[..., 3] = 255
    flat[:, 160:, 0] = 200
    fdata = _pil(flat, quality=90, subsampling=0)
    assert _analyze_format_is_png(orc.jpeg_decode(fdata))
    out, q, s, steps, od, fd = ctx.compress_file_jpeg(fdata, 0.94, auto_format=True)
    assert out is None and od == fd == (320, 240)
    out, q, s, steps, od, fd = ctx.compress_file_jpeg(fdata, 0.94, auto_format=False)          # Format: JPEG
    assert (out, q, s, steps) == ctx.jpeg_compress(orc.jpeg_decode(fdata), 0.94)


@pytest.mark.gpu
def test_gpu_native_pool_with_file_options(ctx):
    """fennec_CompressBatchJPEGOpts: BatchOptions.DefaultOpts with per-item overrides (batch.go:101-105), against the single call."""
    import ctypes as C
    import fennec_amd as fa
    L = fa.load_library()
    files = [_pil(_photo(400 + 16 * k, 300, k), quality=92, subsampling=2) for k in range(6)]
    n = len(files)
    default = fa.FileOptions(1, 256, 0, 1, 0.94)
    special = fa.FileOptions(6, 0, 0, 0, 0.97)
    per = (C.POINTER(fa.FileOptions) * n)()
    per[2] = C.pointer(special)
    arrs = [np.frombuffer(f, dtype=np.uint8) for f in files]
    srcs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs]); sizes = (C.c_size_t * n)(*[len(f) for f in files])
    bufs = [np.empty(3 * len(f) + 65536, dtype=np.uint8) for f in files]     # (a higher target can pick a higher quality than the source's)
    outs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); caps = (C.c_size_t * n)(*[b.size for b in bufs])
    res = (fa.NativeBatchResult * n)()
    dims = (C.c_int * (4 * n))()
    assert L.fennec_CompressBatchJPEGOpts(0, 3, n, srcs, sizes, C.byref(default), per, outs, caps, res, dims, None, None, None) == fa.FNX_OK
    for i in range(n):
        o = special if i == 2 else default
        out, q, s, steps, od, fd = ctx.compress_file_jpeg(files[i], o.target_ssim, orient=o.orient, max_w=o.max_w, max_h=o.max_h,
                                                          auto_format=bool(o.auto_format))
        r = res[i]
        assert not r.failed and (r.quality, r.ssim, r.steps, r.compressed_size, r.original_size) == (q, s, steps, len(out), len(files[i])), i
        assert bufs[i][:r.compressed_size].tobytes() == out and tuple(dims[4 * i:4 * i + 4]) == od + fd
    assert tuple(dims[8:12]) == (300, 432, 300, 432) and tuple(dims[0:4]) == (400, 300, 256, 192)


@pytest.mark.gpu
def test_gpu_pool_over_a_device_list(ctx):
    """fennec_CompressBatchJPEGDevices / NRGBADevices (batch.go:63-126 with g GPUs): workers g x k, worker i on
    devices[i mod g], ONE index queue.  One GPU here, so the list names device 0 twice -- two 'devices', four workers: every item is
    done once, by index, with the per-item results of the single-device pool, and each result says which list entry served it."""
    from fennec_amd import batch
    files = [_pil(_photo(640 + 16 * k, 480 - 8 * k, k), quality=95 - k, subsampling=2 if k % 2 else 0) for k in range(12)]
    one, outs1, summ1 = batch.compress_batch_jpeg_native(files, 0.94, workers=4)
    two, outs2, summ2 = batch.compress_batch_jpeg_native(files, 0.94, workers=4, devices=[0, 0])
    assert [r.Index for r in two] == list(range(len(files))) and all(r.Err is None for r in two)
    assert outs1 == outs2
    assert [(r.Quality, r.SSIM, r.steps, r.CompressedSize, r.OriginalSize) for r in one] == [(r.Quality, r.SSIM, r.steps, r.CompressedSize, r.OriginalSize) for r in two]
    assert (summ1.Total, summ1.Succeeded, summ1.TotalSaved, summ1.AvgSSIM) == (summ2.Total, summ2.Succeeded, summ2.TotalSaved, summ2.AvgSSIM)
    assert all(r.device == 0 for r in two)                  # both list entries are logical device 0
    imgs = [synth.large_photo(320 + 16 * k, 240, k) for k in range(6)]
    r1, f1, _ = batch.compress_batch_native(imgs, 0.94, workers=2)
    r2, f2, _ = batch.compress_batch_native(imgs, 0.94, workers=4, devices=[0, 0])
    assert f1 == f2 and [r.Quality for r in r1] == [r.Quality for r in r2]
    with pytest.raises(ValueError):
        batch.compress_batch_native(imgs, 0.94, devices=[])
    import fennec_amd as fa
    with pytest.raises(fa.FennecError):                     # a device the library does not have
        batch.compress_batch_native(imgs, 0.94, workers=2, devices=[fa.load_library().fnx_device_count()])


@pytest.mark.gpu
def test_gpu_pool_retries_an_item_whose_file_outgrows_its_buffer(ctx):
    """compress_batch_native sized every output at 4096 + 1.5 B/px and reported larger files as failed.  Noise at a
    target no quality reaches falls back to q = 100 (compress.go:76-84) and writes > 1.5 B/px: the item must come back whole."""
    from fennec_amd import batch
    noisy = synth.noise_image(256, 192, 5)
    calm = synth.make_test_image(256, 192)
    res, files, summ = batch.compress_batch_native([noisy, calm], 0.9999, workers=2)
    assert all(r.Err is None for r in res) and summ.Failed == 0
    data, q, s, n = ctx.jpeg_compress(noisy, 0.9999)
    assert len(data) > 4096 + 256 * 192 * 3 // 2               # the case is real: the first buffer was too small
    assert files[0] == data and (res[0].Quality, res[0].CompressedSize) == (q, len(data))


@pytest.mark.gpu
def test_gpu_pool_keeps_the_batch_when_one_file_is_beyond_repair(ctx):
    """a file neither the device decoder nor the host codec can read is THAT item's Err (batch.go:108-113);
    the other items' results stand."""
    from fennec_amd import batch
    good = [_pil(_photo(320, 240, k), quality=92, subsampling=2) for k in range(3)]
    files = [good[0], b"\xff\xd8 this is not a JPEG file", good[1], good[2][:200]]
    res, outs, summ = batch.compress_batch_jpeg_native(files, 0.94, workers=2)
    assert [r.Index for r in res] == [0, 1, 2, 3]
    assert res[0].Err is None and res[2].Err is None and outs[0] and outs[2]
    assert res[1].Err is not None and res[3].Err is not None and outs[1] == b"" and outs[3] == b""
    assert (summ.Total, summ.Succeeded, summ.Failed) == (4, 2, 2)


@pytest.mark.gpu
def test_gpu_device_selection_and_force_off():
    """fnx_set_devices / FENNEC_HIP_DEVICES / FENNEC_HIP_DISABLE: an empty list is 'no device' -- the status the
    cgo shim falls back on -- and (NULL, -1) restores the default.  In a child process: the choice is process-wide."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import fennec_amd as fa
L = fa.load_library()
n = L.fnx_device_count()
assert n >= 1
assert L.fnx_set_devices(None, 0) == 0 and L.fnx_device_count() == 0
h = C.c_void_p()
assert L.fnx_ctx_create(0, C.byref(h)) == -2 and b"no HIP device" in L.fnx_last_error()      # FNX_ERR_NO_DEVICE
try:
    fa.Context(0)
    raise SystemExit("a context on a switched-off library")
except fa.FennecError:
    pass
assert L.fnx_set_devices((C.c_int * 2)(0, 0), 2) == 0 and L.fnx_device_count() == 2
c = fa.Context(1)                                            # logical device 1 = HIP device 0
assert L.fnx_ctx_device(c._h) == 1
import numpy as np
img = np.zeros((16, 16, 4), np.uint8); img[..., 3] = 255
assert c.SSIMFast(img, img) == 1.0
assert L.fnx_set_devices((C.c_int * 1)(n + 3), 1) == -1      # no such HIP device
assert L.fnx_set_devices(None, -1) == 0 and L.fnx_device_count() == n
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    env = dict(os.environ, FENNEC_HIP_DISABLE="1")
    r = subprocess.run([sys.executable, "-c", "import os,sys; sys.path.insert(0, os.getcwd()); import fennec_amd as fa; print(fa.load_library().fnx_device_count())"],
                       cwd=root, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("0"), r.stdout + r.stderr
    env = dict(os.environ, FENNEC_HIP_DEVICES="0,0,0", FNX_ROCTX="1")      # roctx ranges on: calls must simply keep working
    r = subprocess.run([sys.executable, "-c", "import os,sys; sys.path.insert(0, os.getcwd()); import numpy as np, fennec_amd as fa; "
                        "print(fa.load_library().fnx_device_count()); c = fa.Context(2); a = np.full((32, 32, 4), 9, np.uint8); print(c.SSIMFast(a, a))"],
                       cwd=root, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.split() == ["3", "1.0"], r.stdout + r.stderr
