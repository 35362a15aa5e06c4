#!/usr/bin/env python3
"""Randomised HIP-vs-oracle sweep (run on a GPU box): python tools/fuzz_gpu.py [seconds] [seed].
Not part of the test suite (time-boxed, random); every failure prints the reproducing seed/case."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = fennec_amd.Context(0)
SSIM_TOL = 1e-9
fails, runs = [], {}


def rand_image(w, h):
    kind = rng.integers(0, 4)
    if kind == 0:
        return synth.noise_image(w, h, int(rng.integers(1 << 30)), alpha=bool(rng.integers(2)))
    if kind == 1:
        return synth.large_photo(w, h, int(rng.integers(100)))
    if kind == 2:
        return synth.make_test_image_with_alpha(w, h)
    img = synth.noise_image(w, h, int(rng.integers(1 << 30)), alpha=True)
    img[..., :3] &= 0xF0                       # few colours, ties in the filters
    return img


def blur_close(got, want):
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    n_off = int((d[..., :3] != 0).sum())        # <= 0.1 % of samples; tiny images: <= 6 samples (a rate needs a population,
    # and a tiny image of few grey levels is all rounding ties: seed 31, it 9 -- 4 of 1 380 on 92 x 5)
    return d.max() <= 1 and n_off <= max(6, 1e-3 * d[..., :3].size) and np.array_equal(got[..., 3], want[..., 3])


def case(name, ok, desc):
    runs[name] = runs.get(name, 0) + 1
    if not ok:
        fails.append((name, desc))
        print("FAIL", name, desc, flush=True)


t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
    if rng.integers(8) == 0:
        w, h = int(rng.integers(500, 2300)), int(rng.integers(300, 1300))
    img = rand_image(w, h)
    desc = f"seed={seed} it={it} {w}x{h}"
    sigma = float(rng.choice([0.3, 0.7, 1.0, 1.5, 2.0, 2.6, 3.4, 3.9, 4.4, 5.0, 5.33]))      # r3: radii 9..16 in the tile kernel
    want = orc.gaussian_blur(img, sigma, procs=8)
    case("blur_exact", np.array_equal(ctx.GaussianBlur(img, sigma, exact=True), want), desc + f" sigma={sigma}")
    fast = ctx.GaussianBlur(img, sigma)
    dd = np.abs(fast.astype(np.int16) - want.astype(np.int16))
    case("blur_fast", blur_close(fast, want), desc + f" sigma={sigma} off={int((dd[..., :3] != 0).sum())} of {dd[..., :3].size} max={int(dd.max())} "
                                                     f"alpha_equal={np.array_equal(fast[..., 3], want[..., 3])}")
    if w >= 6 and h >= 6 and it % 2 == 0:     # r3: SubImages (stride != 4w): the reference's flat copy(dst.Pix, img.Pix), host and device views
        import torch
        yv, xv = int(rng.integers(0, h // 3 + 1)), int(rng.integers(0, w // 3 + 1))
        hv, wv = int(rng.integers(3, h - yv + 1)), int(rng.integers(3, w - xv + 1))
        sub = img[yv:yv + hv, xv:xv + wv]
        dsub = torch.from_numpy(img).cuda()[yv:yv + hv, xv:xv + wv]
        sv = float(rng.uniform(0.05, 1.0))
        vd = desc + f" view {wv}x{hv}+{xv}+{yv} s={sv}"
        case("view_blur3x3", np.array_equal(ctx.blur3x3(sub), orc.blur3x3(sub)) and np.array_equal(ctx.blur3x3(dsub).cpu().numpy(), orc.blur3x3(sub)), vd)
        case("view_sharpen", np.array_equal(ctx.Sharpen(sub, sv), orc.sharpen(sub, sv)) and np.array_equal(ctx.Sharpen(dsub, sv).cpu().numpy(), orc.sharpen(sub, sv)), vd)
        case("view_adaptive", np.array_equal(ctx.AdaptiveSharpen(sub, sv), orc.adaptive_sharpen(sub, sv)) and np.array_equal(ctx.AdaptiveSharpen(dsub, sv).cpu().numpy(), orc.adaptive_sharpen(sub, sv)), vd)
        oth = rand_image(wv, hv)
        case("view_msssim", abs(ctx.MSSSIM(sub, oth) - orc.msssim(sub, oth)) <= SSIM_TOL and abs(ctx.MSSSIM(oth, sub) - orc.msssim(oth, sub)) <= SSIM_TOL, vd)
        # (below 8 px pixelSSIM walks len(a.Pix), ssim.go:178: a SubImage's slice outruns a tight b's and the reference panics)
        case("view_rowwise", np.array_equal(ctx.GaussianBlur(sub, 1.5, exact=True), orc.gaussian_blur(sub, 1.5)) and
             (min(wv, hv) < 8 or abs(ctx.SSIMFast(sub, np.ascontiguousarray(oth)) - orc.ssim_fast(sub, oth)) <= SSIM_TOL), vd)
    st = float(rng.uniform(0.05, 1.3))
    case("sharpen", np.array_equal(ctx.Sharpen(img, st), orc.sharpen(img, st, procs=4)), desc + f" s={st}")
    case("adaptive", np.array_equal(ctx.AdaptiveSharpen(img, st), orc.adaptive_sharpen(img, st, procs=4)), desc + f" s={st}")
    dw, dh = int(rng.integers(1, 900)), int(rng.integers(1, 700))
    case("resize", np.array_equal(ctx.lanczosResize(img, dw, dh), orc.lanczos_resize(img, dw, dh, procs=8)), desc + f" -> {dw}x{dh}")
    # r3: ratios the one-launch kernel takes (windows of <= 16 pixels: down to ~2.3:1 and every upscale), both axes apart
    fw, fh = max(2, int(w * rng.uniform(0.44, 2.6))), max(1, int(h * rng.uniform(0.44, 2.6)))
    case("resize_fused", np.array_equal(ctx.lanczosResize(img, fw, fh), orc.lanczos_resize(img, fw, fh, procs=8)), desc + f" -> {fw}x{fh}")
    bw, bh = int(rng.integers(1, max(2, w + 3))), int(rng.integers(1, max(2, h + 3)))
    case("box", np.array_equal(ctx.boxDownsample(img, bw, bh), orc.box_downsample(img, bw, bh)), desc + f" -> {bw}x{bh}")
    o = int(rng.integers(2, 9))
    case("orient", np.array_equal(ctx.ApplyOrientation(img, o), orc.apply_orientation(img, o)), desc + f" o={o}")
    other = ctx.GaussianBlur(img, 1.0) if rng.integers(2) else rand_image(w, h)
    case("ssim", abs(ctx.SSIM(img, other) - orc.ssim(img, other, procs=8)) <= SSIM_TOL, desc)
    case("ssim_fast", abs(ctx.SSIMFast(img, other) - orc.ssim_fast(img, other, procs=8)) <= SSIM_TOL, desc)
    case("msssim", abs(ctx.MSSSIM(img, other) - orc.msssim(img, other, procs=8)) <= SSIM_TOL, desc)
    # r6: the one-call GaussianBlur + SSIMFast (host image: one crossing each way), the batched resize and MSSSIM entry points
    # (the image as a grid dimension), and -- on planes large enough for the two-column marching kernel -- FNX_SSIM_FAST
    gb, gs = ctx.GaussianBlurSSIMFast(img, sigma, exact=True)
    case("blur_ssim_one_call", np.array_equal(gb, want) and (min(w, h) < 1 or abs(gs - orc.ssim_fast(img, want, procs=8)) <= SSIM_TOL), desc + f" sigma={sigma}")
    if it % 3 == 0:
        import torch
        nb = int(rng.integers(2, 6))
        batch = [img] + [rand_image(w, h) for _ in range(nb - 1)]
        dev = [torch.from_numpy(b).cuda() for b in batch]
        got = ctx.lanczosResizeBatch(dev, fw, fh)
        ctx.sync()
        case("resize_batch", all(np.array_equal(g.cpu().numpy(), orc.lanczos_resize(b, fw, fh, procs=8)) for g, b in zip(got, batch)), desc + f" x{nb} -> {fw}x{fh}")
        if w >= 3 and h >= 3:
            ad = bool(rng.integers(2))
            sh = ctx.sharpen_batch(dev, st, adaptive=ad)
            ctx.sync()
            case("sharpen_batch", all(np.array_equal(g.cpu().numpy(), (orc.adaptive_sharpen if ad else orc.sharpen)(b, st, procs=4)) for g, b in zip(sh, batch)),
                 desc + f" x{nb} s={st} adaptive={ad}")
            ctx.ssim_batch_enqueue(dev, sh)
            gsb = ctx.fetch_results(nb)
            case("ssim_batch", all(abs(g - orc.ssim(b, s_.cpu().numpy(), procs=8)) <= SSIM_TOL for g, b, s_ in zip(gsb, batch, sh)), desc + f" x{nb}")
        if w >= 16 and h >= 16:
            smalls = ctx.lanczosResizeBatch(dev, max(8, w // 2), max(8, h // 2))
            ctx.msssim_batch_enqueue(dev, smalls)
            gm = ctx.fetch_results(nb)
            wm = [orc.msssim(b, s_.cpu().numpy(), procs=8) for b, s_ in zip(batch, smalls)]
            case("msssim_batch", all(abs(g - w_) <= SSIM_TOL for g, w_ in zip(gm, wm)), desc + f" x{nb}")
    if it % 16 == 0:
        bw2, bh2 = int(rng.integers(2100, 2700)), int(rng.integers(2000, 2400))
        big, big2 = rand_image(bw2, bh2), None
        big2 = ctx.GaussianBlur(big, float(rng.choice([0.8, 1.5, 2.5]))) if rng.integers(2) else rand_image(bw2, bh2)
        wbig = orc.ssim(big, big2, procs=16)
        ctx.set_ssim_mode(True)
        try:
            fbig = ctx.SSIM(big, big2)
        finally:
            ctx.set_ssim_mode(False)
        case("ssim_fast_moments", abs(fbig - wbig) <= 1e-6 and abs(ctx.SSIM(big, big2) - wbig) <= SSIM_TOL, desc + f" big {bw2}x{bh2} delta {fbig - wbig:+.2e}")
    a, wa = ctx.analyze_raw(img), orc.analyze(img)
    ok = (np.array_equal(a["histogram"].astype(np.float64), wa["histogram"]) and
          all(a[k] == wa[k] for k in ("has_alpha", "is_grayscale", "unique_colors", "sample_count", "edge_count", "edge_total")) and
          abs(a["bright_sum"] - wa["bright_sum"]) <= max(1e-12, w * h * 2.0 ** -53) * max(1.0, abs(wa["bright_sum"])) and
          abs(a["variance_sum"] - wa["variance_sum"]) <= 1e-9 * abs(wa["variance_sum"]) + 1e-6)
    case("analyze", ok, desc)
    case("flat_scans", ctx.isOpaque(img) == orc.is_opaque(img) and ctx.isGrayscale(img) == orc.is_grayscale(img), desc)   # r5: the single-launch scan
    n = int(rng.integers(1, 257))
    pal = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
    pal[:, 3] = 255
    gi, gq = ctx.applyPalette(img, pal)
    wi, wq = orc.apply_palette(img, pal)
    case("palette", np.array_equal(gi, wi) and np.array_equal(gq, wq), desc + f" n={n}")
    ratio = int(rng.integers(0, 6))
    y, cb, cr = synth.ycbcr_planes(w, h, ratio, int(rng.integers(1 << 30)))
    case("ycbcr", np.array_equal(ctx.ycbcrToNRGBA(y, cb, cr, ratio), orc.ycbcr_to_nrgba(y, cb, cr, ratio)), desc + f" ratio={ratio}")
    # the JPEG path: quantisation round trip, the file, the search (all integer: exact)
    q = int(rng.integers(1, 101))
    case("jpeg_roundtrip", np.array_equal(ctx.jpeg_roundtrip(img, q), orc.jpeg_roundtrip(img, q)), desc + f" q={q}")
    data = ctx.jpeg_encode(img, q)
    okj = data == orc.jpeg_encode(img, q)
    if okj and it % 4 == 0:
        okj = np.array_equal(orc.jpeg_decode(data), orc.jpeg_roundtrip(img, q)) and ctx.jpeg_encoded_size(img, q) == len(data)
    case("jpeg_encode", okj, desc + f" q={q}")
    # the decoder: the device's own file, libjpeg's files (every subsampling it takes, grey, optimised tables), damaged scans
    case("jpeg_decode_own", np.array_equal(ctx.jpeg_decode(data), orc.jpeg_decode(data)), desc + f" q={q}")
    import io
    from PIL import Image
    rgb = Image.fromarray(np.ascontiguousarray(img[..., :3]), "RGB")
    sub = int(rng.integers(0, 4))
    buf = io.BytesIO()
    rs = {} if rng.integers(2) else ({"restart_marker_rows": int(rng.integers(1, 4))} if rng.integers(2) else
                                     {"restart_marker_blocks": int(rng.integers(1, 40))})
    try:
        if sub == 3:
            rgb.convert("L").save(buf, "JPEG", quality=q, optimize=bool(rng.integers(2)) and q < 96, **rs)
        else:
            rgb.save(buf, "JPEG", quality=q, subsampling=sub, optimize=bool(rng.integers(2)) and q < 96, **rs)
        pdata = buf.getvalue()
    except OSError:                          # Pillow's encoder buffer (noise at high quality)
        pdata = None
    if pdata is not None:
        case("jpeg_decode_libjpeg", np.array_equal(ctx.jpeg_decode(pdata), orc.jpeg_decode(pdata)), desc + f" q={q} sub={sub} {rs}")
        scan = pdata.index(b"\xff\xda") + (14 if sub != 3 else 10)
        if len(pdata) - 2 > scan:
            bad = bytearray(pdata)
            for _ in range(int(rng.integers(1, 4))):
                bad[int(rng.integers(scan, len(bad) - 2))] = int(rng.integers(0, 256))
            bad = bytes(bad)
            try:
                want = orc.jpeg_decode(bad)
            except Exception:
                want = None
            try:
                got = ctx.jpeg_decode(bad)
            except fennec_amd.FennecError:
                got = None
            # both decode: the same pixels.  One refuses what the other takes: reported, not failed (a damaged scan's last
            # symbol may straddle the end of the string, which the two treat differently) -- counted under its own name
            if got is not None and want is not None:
                same = np.array_equal(got, want)
                if not same:                 # keep the file: the iteration is not reproducible without everything before it
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    open(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{seed}_{it}.jpg"), "wb").write(bad)
                    open(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{seed}_{it}_orig.jpg"), "wb").write(pdata)
                    np.save(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{seed}_{it}_got.npy"), got)
                case("jpeg_decode_damaged_both", same, desc + f" q={q} sub={sub} {rs}")
            elif (got is None) != (want is None):
                runs["jpeg_decode_damaged_one_sided"] = runs.get("jpeg_decode_damaged_one_sided", 0) + 1
            else:
                runs["jpeg_decode_damaged_neither"] = runs.get("jpeg_decode_damaged_neither", 0) + 1
    if it % 5 == 0:
        target = float(rng.choice([0.5, 0.9, 0.94, 0.97, 0.99, 1.0]))
        data, bq, bs, bn = ctx.jpeg_compress(img, target)
        sq, ss_, sn, found = ctx.jpeg_quality_search(img, target)
        case("jpeg_compress", (bq, bs, bn) == (sq, ss_, sn) and data == orc.jpeg_encode(img, bq) and (found or bq == 100), desc + f" target={target}")
    if it % 3 == 0:                         # the result FIFO: SSIM on the second stream, MSSSIM with the implicit resize
        import torch
        da, db = torch.from_numpy(img).cuda(), torch.from_numpy(other).cuda()
        small = ctx.lanczosResize(da, max(1, w // 2), max(1, h // 2))
        ctx.ssim_enqueue(da, db)
        ctx.msssim_enqueue(da, small)
        v1, v2 = ctx.fetch_result(), ctx.fetch_result()
        case("enqueue", v1 == ctx.SSIM(da, db) and v2 == ctx.MSSSIM(da, small) and
             abs(v2 - orc.msssim(img, small.cpu().numpy(), procs=8)) <= SSIM_TOL, desc)
    if it % 10 == 0:                        # the two-column marching SSIM (>= 4 M windows per image), odd dims, resize at scale
        w3, h3 = int(rng.integers(2100, 4200)), int(rng.integers(1930, 2400))
        big = rand_image(w3, h3)
        oth = ctx.AdaptiveSharpen(big, 0.5)
        case("ssim_big", abs(ctx.SSIM(big, oth) - orc.ssim(big, oth, procs=32)) <= SSIM_TOL, f"seed={seed} it={it} {w3}x{h3}")
        dw3, dh3 = int(rng.integers(600, 2200)), int(rng.integers(400, 1300))
        case("resize_big", np.array_equal(ctx.lanczosResize(big, dw3, dh3), orc.lanczos_resize(big, dw3, dh3, procs=32)),
             f"seed={seed} it={it} {w3}x{h3} -> {dw3}x{dh3}")
        case("adaptive_big", np.array_equal(oth, orc.adaptive_sharpen(big, 0.5, procs=32)), f"seed={seed} it={it} {w3}x{h3}")
    if it % 6 == 0:                         # one-pass kernel: sizes inside its gate, odd dims, all radii
        import torch
        w2, h2 = int(rng.integers(1850, 4300)), int(rng.integers(1100, 2600))
        imgs = [rand_image(w2, h2) for _ in range(2)]
        d = [torch.from_numpy(i).cuda() for i in imgs]
        torch.cuda.synchronize()
        outs, ss = ctx.GaussianBlurSSIMFastBatch(d, sigma)
        ref = ctx.GaussianBlurBatch(d, sigma)
        rs = ctx.SSIMFastBatch(d, ref)
        ok = all(torch.equal(x, y_) for x, y_ in zip(outs, ref)) and np.array_equal(ss, rs)
        ok = ok and abs(ss[0] - orc.ssim_fast(imgs[0], outs[0].cpu().numpy(), procs=16)) <= SSIM_TOL
        ok = ok and blur_close(outs[1].cpu().numpy(), orc.gaussian_blur(imgs[1], sigma, procs=16))
        case("one_pass", ok, f"seed={seed} it={it} {w2}x{h2} sigma={sigma}")
        outs, ss = ctx.GaussianBlurSSIMFastBatch(d, sigma, exact=True)      # guarded kernel: bit-exact images
        want = orc.gaussian_blur(imgs[0], sigma, procs=16)
        ref = ctx.GaussianBlurBatch(d, sigma, exact=True)
        rs = ctx.SSIMFastBatch(d, ref)
        checks = (np.array_equal(outs[0].cpu().numpy(), want),
                  abs(ss[0] - orc.ssim_fast(imgs[0], want, procs=16)) <= SSIM_TOL,
                  torch.equal(outs[1], ref[1]), ss[1] == rs[1])
        ok = all(checks)
        case("one_pass_exact", ok, f"seed={seed} it={it} {w2}x{h2} sigma={sigma} checks={checks} ss={ss} rs={rs}")
print("runs:", runs)
print("FAILURES:", len(fails))
for f in fails[:20]:
    print("  ", f)
sys.exit(1 if fails else 0)
