"""SURVEY 8(f)2, first slice: the JPEG quantisation round trip.

CPU (not gpu): the oracle's restatement of Go's image/jpeg arithmetic against what CAN be checked here -- libjpeg-turbo
through Pillow.  Quantisation tables must be identical for every quality (Go copied the IJG's scaling); the integer
FDCT / IDCT must sit within 1 of the real DCT; a decoded round trip must be close to libjpeg's (not identical: libjpeg
uses other chroma constants, alternates its downsampling bias and smooths chroma on the way up) and give nearly the
same SSIMFast.  None of that pins the restatement against Go itself: unpinned twice over.

GPU (-m gpu): jpeg.hip against the oracle bit for bit -- planes' worth of blocks, odd sizes, translucent pixels, every
quality regime -- and fnx_jpeg_quality_search against the same search driven by the oracle.
"""
import io

import numpy as np
import pytest

from fennec_amd import batch, synth
from oracle import oracle as orc


def _pil_tables(q):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(synth.make_test_image(32, 32)[..., :3], "RGB").save(buf, "JPEG", quality=q, subsampling=2)
    t = Image.open(io.BytesIO(buf.getvalue())).quantization
    return np.array(t[0]).reshape(8, 8), np.array(t[1]).reshape(8, 8)


def test_quant_tables_equal_libjpeg_for_every_quality():
    zig = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])
    for q in list(range(1, 101, 7)) + [49, 50, 51, 92, 99, 100]:
        lum, chr_ = orc.jpeg_quant_tables(q)
        pl, pc = _pil_tables(q)
        # Pillow >= 8.3 reports tables in natural order; older ones in zig-zag order: accept either, identically for both
        nat = np.array_equal(lum, pl) and np.array_equal(chr_, pc)
        zz = np.array_equal(lum.reshape(64)[zig].reshape(8, 8), pl) and np.array_equal(chr_.reshape(64)[zig].reshape(8, 8), pc)
        assert nat or zz, q
    assert orc.jpeg_quant_tables(0)[0].tolist() == orc.jpeg_quant_tables(1)[0].tolist()        # clipped to [1, 100]
    assert orc.jpeg_quant_tables(1000)[0].tolist() == orc.jpeg_quant_tables(100)[0].tolist()
    assert (orc.jpeg_quant_tables(100)[0] == 1).all()


def test_integer_dct_against_the_real_one():
    from scipy.fft import dctn, idctn
    rng = np.random.default_rng(2)
    for k in range(200):
        x = rng.integers(0, 256, (8, 8)) if k % 3 else np.full((8, 8), rng.integers(0, 256))
        f = orc.jpeg_fdct(x)
        assert np.abs(f - 8 * dctn((x - 128).astype(float), norm="ortho")).max() <= 1.5     # jfdctint: scaled by 8; two fixed-point passes
        # coefficients a decoder meets: a pixel block's DCT quantised and dequantised with some step (far larger ones wrap
        # the 32-bit intermediates -- in Go as here -- and no JPEG of 8-bit samples holds them)
        step = int(rng.integers(1, 64))
        c = (np.rint(dctn((x - 128).astype(float), norm="ortho") / step) * step).astype(np.int64)
        assert np.abs(orc.jpeg_idct(c) - idctn(c.astype(float), norm="ortho")).max() <= 1.0 + 1e-9
    assert orc.rgb_to_ycbcr(0, 0, 0) == (0, 128, 128) and orc.rgb_to_ycbcr(255, 255, 255) == (255, 128, 128)
    assert orc.rgb_to_ycbcr(255, 0, 0) == (76, 85, 255) and orc.rgb_to_ycbcr(0, 0, 255) == (29, 255, 107)


@pytest.mark.parametrize("q", [30, 60, 92])
def test_roundtrip_close_to_libjpeg(q):
    smooth = orc.gaussian_blur(synth.noise_image(320, 240, 3), 3.0)
    grad = synth.make_test_image(333, 217)
    for img in (smooth, grad):
        rt = orc.jpeg_roundtrip(img, q)
        pil = batch.pillow_decode(batch.pillow_encode(img, q))
        assert rt.shape == img.shape and (rt[..., 3] == 255).all()
        d = np.abs(rt[..., :3].astype(int) - pil[..., :3].astype(int))
        assert d.mean() < 1.6, d.mean()                                      # about one grey level on average
        assert abs(orc.ssim_fast(img, rt) - orc.ssim_fast(img, pil)) < 2e-3
    assert np.array_equal(orc.jpeg_roundtrip(smooth, 92), orc.jpeg_roundtrip(smooth, 92))


def test_roundtrip_edge_replication_and_premultiplied_alpha():
    # 17 x 9: one full MCU column + 1 px, rows short of one MCU -- the padded planes replicate the last row / column
    img = synth.noise_image(17, 9, 5)
    y, cb, cr = orc.jpeg_roundtrip_planes(img, 100)
    assert y.shape == (16, 32) and cb.shape == (8, 16)
    # at quality 100 (all steps 1) the round trip is the DCT pair alone: luma within 2 of the source's
    yy = np.array([[orc.rgb_to_ycbcr(*img[j, i, :3])[0] for i in range(17)] for j in range(9)])
    assert np.abs(y[:9, :17].astype(int) - yy).max() <= 2
    # a translucent pixel is premultiplied before the colour conversion (color.NRGBA.RGBA())
    a = np.zeros((16, 16, 4), np.uint8); a[...] = (200, 100, 50, 255)
    b = a.copy(); b[..., 3] = 128
    ya = orc.jpeg_roundtrip_planes(a, 100)[0]; yb = orc.jpeg_roundtrip_planes(b, 100)[0]
    want = orc.rgb_to_ycbcr(*(((np.array([200, 100, 50]) * 0x101 * 128 // 0xff) >> 8).tolist()))[0]
    assert abs(int(yb[4, 4]) - want) <= 1 and int(ya[4, 4]) > int(yb[4, 4])


def _segments(data):
    """[(marker, payload)] up to and including SOS, and the offset of the entropy-coded data."""
    out, pos = [], 2
    while pos < len(data):
        assert data[pos] == 0xFF
        m, ln = data[pos + 1], (data[pos + 2] << 8) | data[pos + 3]
        out.append((m, data[pos + 4:pos + 2 + ln]))
        pos += 2 + ln
        if m == 0xDA:
            break
    return out, pos


def test_encoder_files_decode_with_libjpeg_and_with_the_oracle():
    """orc.jpeg_encode: jpeg.Encode's file as restated (baseline, 4:2:0, typical Huffman tables, writer.go's segment
    order).  libjpeg-turbo must decode it, to pixels close to the round trip's (other IDCT, fancy upsampling); the
    oracle's own decoder must return the round trip exactly -- planes and coefficients."""
    from PIL import Image
    imgs = [orc.gaussian_blur(synth.noise_image(333, 217, 3), 2.0), synth.make_test_image(64, 48), synth.noise_image(17, 9, 5),
            synth.large_photo(640, 480, 2), synth.noise_image(16, 16, 1, alpha=True)]
    for img in imgs:
        for q in (1, 30, 75, 92, 100):
            data, coef = orc.jpeg_encode(img, q, with_coefficients=True)
            assert data[:2] == b"\xff\xd8" and data[-2:] == b"\xff\xd9"
            pil = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
            rt = orc.jpeg_roundtrip(img, q)
            assert pil.shape[:2] == img.shape[:2]
            if img.shape[0] * img.shape[1] >= 3000 and q >= 30:
                # smooth content: within a couple of levels (IDCT and chroma upsampling differ by design); high-frequency
                # chroma (the modular pattern, noise): libjpeg's "fancy" upsampling alone is worth several levels
                d = np.abs(pil.astype(int) - rt[..., :3]).mean()
                assert d < (2.5 if img is imgs[0] else 12.0), d
            w, h, ratio, y, cb, cr, coef2 = orc.jpeg_decode_planes(data, with_coefficients=True)
            yy, cbb, crr = orc.jpeg_roundtrip_planes(img, q)
            assert (w, h, ratio) == (img.shape[1], img.shape[0], 2)
            assert np.array_equal(y, yy) and np.array_equal(cb, cbb) and np.array_equal(cr, crr) and np.array_equal(coef, coef2)
            assert np.array_equal(orc.jpeg_decode(data), rt)
            # no 0xff in the scan without a stuffed 0x00 (or the EOI)
            segs, pos = _segments(data)
            scan = data[pos:-2]
            assert all(scan[i + 1] == 0 for i in range(len(scan) - 1) if scan[i] == 0xFF) and (not scan or scan[-1] != 0xFF)
            assert [m for m, _ in segs] == [0xDB, 0xC0, 0xC4, 0xDA]          # writer.go: DQT, SOF0, DHT, SOS -- no APP0


def test_huffman_tables_are_libjpegs_and_the_decoder_reads_libjpeg_files():
    img = orc.gaussian_blur(synth.noise_image(320, 240, 7), 1.5)
    pdata = batch.pillow_encode(img, 75)                              # not optimised: the typical tables of Annex K.3.3
    psegs, _ = _segments(pdata)
    msegs, _ = _segments(orc.jpeg_encode(img, 75))
    dht = lambda segs: b"".join(p for m, p in segs if m == 0xC4)
    assert dht(psegs) == dht(msegs) and len(dht(msegs)) == 4 * 17 + 2 * 12 + 2 * 162
    dqt = lambda segs: b"".join(p for m, p in segs if m == 0xDB)
    assert dqt(psegs) == dqt(msegs)                                   # same tables, same (zig-zag) order in the file
    mine, theirs = orc.jpeg_decode(pdata), batch.pillow_decode(pdata)
    d = np.abs(mine[..., :3].astype(int) - theirs[..., :3].astype(int))
    assert d.mean() < 2.5, d.mean()
    # canonical code lengths of the LUT = the spec's counts
    luts = orc.jpeg_huffman_luts()
    assert sorted((luts[1] >> 24)[luts[1] != 0].tolist()).count(16) == 125 and int((luts[0] != 0).sum()) == 12


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def ctx():
    import fennec_amd
    return fennec_amd.Context(0)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(16, 16), (17, 9), (1, 1), (640, 480), (333, 217), (1000, 37), (8, 300), (1920, 1080)])
def test_gpu_roundtrip_bit_exact(ctx, w, h):
    import torch
    imgs = [synth.large_photo(w, h, 2), synth.noise_image(w, h, w + h, alpha=True)]
    if w * h >= 64:
        imgs.append(orc.gaussian_blur(synth.noise_image(w, h, 4), 2.5))
    for img in imgs:
        for q in (1, 17, 30, 50, 75, 92, 100):
            want = orc.jpeg_roundtrip(img, q)
            assert np.array_equal(ctx.jpeg_roundtrip(img, q), want), (w, h, q)
        d = torch.from_numpy(img).cuda()
        assert np.array_equal(ctx.jpeg_roundtrip(d, 60).cpu().numpy(), orc.jpeg_roundtrip(img, 60))


@pytest.mark.gpu
def test_gpu_roundtrip_4k_and_strided_view(ctx):
    import torch
    img = synth.large_photo(3840, 2160, 1)
    for q in (30, 85):
        assert np.array_equal(ctx.jpeg_roundtrip(img, q), orc.jpeg_roundtrip(img, q)), q
    small = synth.noise_image(150, 70, 1)
    big = torch.from_numpy(np.ascontiguousarray(np.pad(small, ((0, 0), (2, 3), (0, 0))))).cuda()
    assert np.array_equal(ctx.jpeg_roundtrip(big[:, 2:-3], 55).cpu().numpy(), orc.jpeg_roundtrip(small, 55))


def _oracle_search(img, target):
    """compress.go:21-74 with the oracle's round trip and SSIMFast."""
    if target >= 1.0:
        target = 0.999
    lo, hi, best_q, best_s, found, n = batch.search_lower_bound(target), 100, 100, 1.0, False, 0
    while lo <= hi:
        mid = (lo + hi) // 2
        s = orc.ssim_fast(img, orc.jpeg_roundtrip(img, mid), procs=8)
        n += 1
        if s >= target:
            best_q, best_s, found, hi = mid, s, True, mid - 1
        else:
            lo = mid + 1
    return best_q, best_s, n, found


@pytest.mark.gpu
def test_gpu_quality_search_matches_oracle_search(ctx):
    import torch
    cases = [(synth.large_photo(1920, 1080, 3), 0.94), (orc.gaussian_blur(synth.noise_image(800, 600, 2), 2.0), 0.97),
             (synth.make_test_image(640, 480), 0.90), (synth.noise_image(300, 200, 9), 0.999), (synth.make_test_image(400, 300), 1.0),
             (synth.large_photo(700, 500, 1), 0.5)]
    for img, target in cases:
        q, s, n, found = ctx.jpeg_quality_search(img, target)
        wq, ws, wn, wfound = _oracle_search(img, target)
        assert (q, n, found) == (wq, wn, wfound), (target, q, wq)
        assert abs(s - ws) <= 1e-9
        dq, ds, dn, dfound = ctx.jpeg_quality_search(torch.from_numpy(img).cuda(), target)
        assert (dq, dn, dfound) == (q, n, found) and ds == s


@pytest.mark.gpu
def test_gpu_compress_batch_with_device_search(ctx):
    """CompressBatch with the search on the device: every item's (quality, steps, SSIM) equals the oracle-driven search over
    the oracle's round trip, the output is the host codec's encode at that quality."""
    import fennec_amd
    jpegs = [batch.pillow_encode(synth.large_photo(1920, 1080, k), 92) for k in range(4)]
    states = {}
    res = batch.compress_batch(len(jpegs), batch.jpeg_item_work_device_search(jpegs), lambda wid: states.setdefault(wid, fennec_amd.Context(0)),
                               workers=2)
    for r in res:
        assert r.Err is None
        src = batch.pillow_decode(jpegs[r.Index])
        wq, ws, wn, _ = _oracle_search(src, batch.TARGET_SSIM["Balanced"])
        assert (r.Quality, r.steps) == (wq, wn) and abs(r.SSIM - ws) <= 1e-9
        assert r.CompressedSize == len(batch.pillow_encode(src, wq))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(16, 16), (17, 9), (1, 1), (640, 480), (333, 217), (1000, 37), (1920, 1080)])
def test_gpu_encode_is_the_oracles_file(ctx, w, h):
    """fnx_jpeg_encode: byte for byte the file the oracle's encoder writes; libjpeg-turbo decodes it; the oracle's decoder
    returns fnx_jpeg_roundtrip's pixels from it."""
    import torch
    from PIL import Image
    imgs = [synth.large_photo(w, h, 2), synth.noise_image(w, h, w + h, alpha=True)]
    if w * h >= 64:
        imgs.append(orc.gaussian_blur(synth.noise_image(w, h, 4), 2.5))
    for img in imgs:
        for q in (1, 30, 75, 92, 100):
            want = orc.jpeg_encode(img, q)
            got = ctx.jpeg_encode(img, q)
            assert got == want, (w, h, q, len(got), len(want))
        got = ctx.jpeg_encode(torch.from_numpy(img).cuda(), 60)
        assert got == orc.jpeg_encode(img, 60)
        assert Image.open(io.BytesIO(got)).size == (w, h)
        assert np.array_equal(orc.jpeg_decode(got), ctx.jpeg_roundtrip(img, 60))


@pytest.mark.gpu
def test_gpu_encode_4k_and_all_ff(ctx):
    img = synth.large_photo(3840, 2160, 1)
    for q in (30, 92):
        assert ctx.jpeg_encode(img, q) == orc.jpeg_encode(img, q), q
    # noise at quality 100: long codes, many 0xff bytes to stuff
    noise = synth.noise_image(512, 384, 9)
    data = ctx.jpeg_encode(noise, 100)
    assert data == orc.jpeg_encode(noise, 100) and data.count(b"\xff\x00") > 50
    for q in (5, 50, 100):                                        # the size query copies nothing
        assert ctx.jpeg_encoded_size(noise, q) == len(orc.jpeg_encode(noise, q))


@pytest.mark.gpu
def test_gpu_compress_is_search_plus_encode(ctx):
    """fnx_jpeg_compress = compressJPEGOptimal in one call: the search's (quality, SSIM, steps), then the file at that
    quality -- the oracle's file; when nothing reaches the target the file is the one at 100 (compress.go:82-86)."""
    import torch
    cases = [(synth.large_photo(1280, 720, 3), 0.94), (orc.gaussian_blur(synth.noise_image(800, 600, 2), 2.0), 0.97),
             (synth.noise_image(300, 200, 9), 0.999), (synth.make_test_image(400, 300), 1.0)]
    for img, target in cases:
        data, q, s, n = ctx.jpeg_compress(img, target)
        wq, ws, wn, found = ctx.jpeg_quality_search(img, target)
        assert (q, s, n) == (wq, ws, wn)
        assert data == orc.jpeg_encode(img, q)
        assert found or q == 100
        d2, q2, s2, n2 = ctx.jpeg_compress(torch.from_numpy(img).cuda(), target)
        assert (d2, q2, s2, n2) == (data, q, s, n)


@pytest.mark.gpu
def test_gpu_compress_batch_with_device_codec(ctx):
    import fennec_amd
    jpegs = [batch.pillow_encode(synth.large_photo(1920, 1080, k), 92) for k in range(4)]
    states = {}
    res = batch.compress_batch(len(jpegs), batch.jpeg_item_work_device_codec(jpegs), lambda wid: states.setdefault(wid, fennec_amd.Context(0)),
                               workers=2)
    for r in res:
        assert r.Err is None
        src = batch.pillow_decode(jpegs[r.Index])
        wq, ws, wn, _ = _oracle_search(src, batch.TARGET_SSIM["Balanced"])
        assert (r.Quality, r.steps) == (wq, wn) and abs(r.SSIM - ws) <= 1e-9
        assert r.data == orc.jpeg_encode(src, wq) and r.CompressedSize == len(r.data)
        assert np.array_equal(orc.jpeg_decode(r.data), orc.jpeg_roundtrip(src, wq))


def _oracle_size_search(img, target_bytes):
    """targetsize.go:125-176 with the oracle's encoder."""
    h, w = img.shape[:2]
    bpp = float(target_bytes * 8) / float(w * h)
    lo, hi = 1, 100
    if bpp < 0.5:
        hi = 40
    elif bpp < 1.0:
        lo, hi = 10, 70
    elif bpp < 2.0:
        lo, hi = 30, 90
    elif bpp > 4.0:
        lo = 60
    best, best_q, n = None, 0, 0
    while lo <= hi:
        mid = (lo + hi) // 2
        data = orc.jpeg_encode(img, mid)
        n += 1
        if len(data) <= target_bytes:
            best, best_q, lo = data, mid, mid + 1
        else:
            hi = mid - 1
    return best, best_q, n


@pytest.mark.gpu
def test_gpu_size_search_matches_oracle_search(ctx):
    """fnx_jpeg_size_search = jpegQualitySearchOpt: the highest quality whose file fits, its file, its SSIMFast."""
    import torch
    imgs = [synth.large_photo(1280, 720, 3), orc.gaussian_blur(synth.noise_image(800, 600, 2), 2.0), synth.noise_image(300, 200, 9)]
    for img in imgs:
        full = len(orc.jpeg_encode(img, 100))
        for target in (full * 2, full // 2, full // 5, full // 20, 700, 10):
            want, wq, wn = _oracle_size_search(img, target)
            got = ctx.jpeg_size_search(img, target)
            if want is None:
                assert got is None, target
                continue
            data, q, s, n = got
            assert (q, n) == (wq, wn) and data == want and len(data) <= target, (target, q, wq)
            assert abs(s - orc.ssim_fast(img, orc.jpeg_roundtrip(img, q), procs=8)) <= 1e-9
            d2 = ctx.jpeg_size_search(torch.from_numpy(img).cuda(), target, skip_ssim=True)
            assert d2[0] == data and d2[1] == q and d2[2] == 0.0


@pytest.mark.gpu
def test_gpu_native_compress_batch_pool(ctx):
    """fennec_CompressBatchNRGBA: the C++ pool (batch.go:58-128) with the device codec per item, against the per-item
    calls and the oracle's files; Summarize (batch.go:140-158) against the python harness's; host and device items."""
    import torch
    imgs = [synth.large_photo(640 + 16 * k, 480 - 8 * k, k) for k in range(9)] + [orc.gaussian_blur(synth.noise_image(333, 217, 3), 2.0)]
    for items in (imgs, [torch.from_numpy(i).cuda() for i in imgs]):
        res, files, summ = batch.compress_batch_native(items, 0.94, workers=4, original_sizes=[4 * i.shape[0] * i.shape[1] for i in imgs])
        assert [r.Index for r in res] == list(range(len(imgs))) and all(r.Err is None for r in res)
        for r, f, img in zip(res, files, imgs):
            data, q, s, n = ctx.jpeg_compress(img, 0.94)
            assert (r.Quality, r.SSIM, r.steps, r.CompressedSize) == (q, s, n, len(data)) and f == data == orc.jpeg_encode(img, q)
        want = batch.summarize_local(res)
        assert (summ.Total, summ.Succeeded, summ.Failed, summ.TotalSaved) == (want.Total, want.Succeeded, want.Failed, want.TotalSaved)
        assert summ.AvgSSIM == want.AvgSSIM
    assert batch.compress_batch_native([]) == ([], [], batch.BatchSummary())


@pytest.mark.gpu
def test_gpu_native_pool_callbacks_cancel_and_release():
    """fennec_CompressBatchNRGBA's OnItem callback (serialised, 1..n), its cancel flag (ctx.Done(): every item reports
    failed, nothing is encoded) and fennec_pool_release (idle worker contexts destroyed; the next batch makes new ones)."""
    import ctypes as C
    import fennec_amd as fa
    L = fa.load_library()
    imgs = [synth.large_photo(320, 240, k) for k in range(7)]
    n = len(imgs)
    views = [fa._Img(i) for i in imgs]
    srcs = (C.c_void_p * n)(*[v.ptr for v in views])
    strides = (C.c_int * n)(*[v.stride for v in views]); ws = (C.c_int * n)(*[v.w for v in views]); hs = (C.c_int * n)(*[v.h for v in views])
    bufs = [np.empty(200000, dtype=np.uint8) for _ in imgs]
    outs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs]); caps = (C.c_size_t * n)(*[b.size for b in bufs])
    res = (fa.NativeBatchResult * n)()
    seen = []
    CB = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_void_p)
    cb = CB(lambda c, t, u: seen.append((c, t)))
    keep = L.fennec_CompressBatchNRGBA.argtypes
    L.fennec_CompressBatchNRGBA.argtypes = keep[:-2] + [CB, C.c_void_p]
    try:
        _pool_callbacks_body(L, fa, C, n, srcs, strides, ws, hs, outs, caps, res, cb, seen, imgs, bufs)
    finally:
        L.fennec_CompressBatchNRGBA.argtypes = keep            # the wrapper (batch.compress_batch_native) passes None for the callback


def _pool_callbacks_body(L, fa, C, n, srcs, strides, ws, hs, outs, caps, res, cb, seen, imgs, bufs):
    cancel = C.c_int(0)
    assert L.fennec_CompressBatchNRGBA(0, 3, n, fa.FNX_HOST, srcs, strides, ws, hs, None, 0.94, outs, caps, res, C.byref(cancel), cb, None) == fa.FNX_OK
    assert sorted(c for c, _ in seen) == list(range(1, n + 1)) and all(t == n for _, t in seen)
    assert all(not r.failed and r.has_result and r.original_size == 4 * 320 * 240 for r in res)
    assert bufs[0][:res[0].compressed_size].tobytes() == orc.jpeg_encode(imgs[0], res[0].quality)
    cancel.value = 1
    seen.clear()
    assert L.fennec_CompressBatchNRGBA(0, 3, n, fa.FNX_HOST, srcs, strides, ws, hs, None, 0.94, outs, caps, res, C.byref(cancel), cb, None) == fa.FNX_OK
    assert all(r.failed and not r.has_result for r in res) and not seen
    out4 = (C.c_int64 * 4)()
    assert L.fennec_SummarizeResults(n, res, out4) == 0.0 and list(out4) == [n, 0, n, 0]
    L.fennec_pool_release()
    cancel.value = 0
    assert L.fennec_CompressBatchNRGBA(0, 2, n, fa.FNX_HOST, srcs, strides, ws, hs, None, 0.94, outs, caps, res, C.byref(cancel), cb, None) == fa.FNX_OK
    assert all(not r.failed for r in res)
    L.fennec_pool_release()
