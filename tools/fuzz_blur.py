#!/usr/bin/env python3
"""Randomised sweep of GaussianBlur (the matrix-pipe kernels of csrc/blur_mfma.hip and everything they fall back to) against
the oracle, on a GPU box: python tools/fuzz_blur.py [seconds] [seed].  Shapes on both sides of the 64-px / 16-row / segment
boundaries, sigma 0.3 .. 21 (radius 1 .. 63), binomial (all-ties) tables of radius 1 .. 63, pitched device views, batches,
and the one-pass SSIMFast form against the two-call route.  Every failure prints its reproducing seed / case."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fennec_amd  # noqa: E402
from fennec_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
import torch  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = fennec_amd.Context(0)
orc.build()
fails, runs = [], {}


def rand_image(w, h):
    kind = rng.integers(0, 5)
    if kind == 0:
        return synth.noise_image(w, h, int(rng.integers(1 << 30)), alpha=bool(rng.integers(2)))
    if kind == 1:
        return synth.large_photo(w, h, int(rng.integers(100)))
    if kind == 2:
        return synth.make_test_image_with_alpha(w, h)
    img = synth.noise_image(w, h, int(rng.integers(1 << 30)), alpha=True)
    if kind == 3:
        img[..., :3] &= 0xF0                   # few colours
    else:
        img[..., :3] &= 1                      # 0 / 1 content: with a binomial table every sample is a tie
    return img


def close(got, want):
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    n_off = int((d[..., :3] != 0).sum())
    return d.max() <= 1 and n_off <= max(6, 1e-3 * d[..., :3].size) and np.array_equal(got[..., 3], want[..., 3]), n_off


def binomial(radius):
    k = np.array([1.0])
    for _ in range(2 * radius):
        k = np.convolve(k, [0.5, 0.5])
    return k


def case(name, ok, desc):
    runs[name] = runs.get(name, 0) + 1
    if not ok:
        fails.append((name, desc))
        print("FAIL", name, desc, flush=True)


EDGES = [63, 64, 65, 127, 128, 129, 191, 192, 256, 271, 272, 273, 287, 288, 543, 544, 545]
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    pick = lambda hi: int(rng.choice(EDGES)) if rng.integers(3) == 0 else int(rng.integers(1, hi))
    w, h = pick(900), pick(700)
    if rng.integers(10) == 0:
        w, h = int(rng.integers(900, 3000)), int(rng.integers(500, 1800))
    img = rand_image(w, h)
    desc = f"seed={seed} it={it} {w}x{h}"
    kernel = None
    if rng.integers(4) == 0:
        kernel = binomial(int(rng.integers(1, 25)) if rng.integers(3) else int(rng.integers(25, 64)))      # r5: 25 .. 62 on the matrix pipe too
        sigma, kd = 1.0, f" binomial R={(len(kernel) - 1) // 2}"
    else:
        sigma = float(rng.choice([0.3, 0.5, 0.66, 0.8, 1.0, 1.34, 1.67, 2.0, 2.01, 2.67, 3.0, 4.66, 4.7, 6.0, 7.33, 7.4, 8.0, 8.1, 8.6,
                                  10.0, 12.66, 12.7, 15.33, 15.4, 18.0, 18.1, 20.0, 20.66, 20.7, 21.0]))
        kd = f" sigma={sigma}"
    want = orc.gaussian_blur(img, sigma, kernel=kernel, procs=16)
    case("blur_exact", np.array_equal(ctx.GaussianBlur(img, sigma, exact=True, kernel=kernel), want), desc + kd)
    ok, n_off = close(ctx.GaussianBlur(img, sigma, kernel=kernel), want)
    case("blur_fast", ok or (kernel is not None), desc + kd + f" off={n_off}")     # tie-built tables: the exact mode is the contract
    if w >= 8 and h >= 8 and it % 2 == 0:          # pitched device views, source and destination
        yv, xv = int(rng.integers(0, h // 3 + 1)), int(rng.integers(0, w // 3 + 1))
        hv, wv = int(rng.integers(4, h - yv + 1)), int(rng.integers(4, w - xv + 1))
        dsub = torch.from_numpy(img).cuda()[yv:yv + hv, xv:xv + wv]
        sub = np.ascontiguousarray(img[yv:yv + hv, xv:xv + wv])
        wv_ = orc.gaussian_blur(sub, sigma, kernel=kernel, procs=8)
        case("view_exact", np.array_equal(ctx.GaussianBlur(dsub, sigma, exact=True, kernel=kernel).cpu().numpy(), wv_), desc + kd + f" view {wv}x{hv}+{xv}+{yv}")
    if it % 5 == 0 and kernel is None:             # batches: the segment length depends on the image count
        n = int(rng.integers(1, 6))
        imgs = [rand_image(w, h) for _ in range(n)]
        d = [torch.from_numpy(i).cuda() for i in imgs]
        torch.cuda.synchronize()
        outs = ctx.GaussianBlurBatch(d, sigma, exact=True)
        case("batch_exact", all(np.array_equal(o.cpu().numpy(), orc.gaussian_blur(i, sigma, procs=8)) for i, o in zip(imgs, outs)), desc + kd + f" n={n}")
    if it % 7 == 0:                                 # one pass == two calls, on shapes that downsample
        bw, bh = int(rng.integers(600, 4200)), int(rng.integers(400, 2400))
        sg = float(rng.choice([1.0, 1.5, 2.0, 2.6, 3.4, 4.6, 4.7, 6.0, 7.3, 7.4, 8.0, 9.0]))      # r5: radii 7 .. 24 run one pass too (27: two calls)
        imgs = [synth.noise_image(bw, bh, int(rng.integers(1 << 30)), alpha=True), synth.large_photo(bw, bh, 5)]
        d = [torch.from_numpy(i).cuda() for i in imgs]
        torch.cuda.synchronize()
        for exact in (False, True):
            outs, ss = ctx.GaussianBlurSSIMFastBatch(d, sg, exact=exact)
            ref = ctx.GaussianBlurBatch(d, sg, exact=exact)
            ref_ss = ctx.SSIMFastBatch(d, ref) if max(bw, bh) > 512 else np.array([ctx.SSIMFast(a, b) for a, b in zip(d, ref)])
            case("one_pass", all(torch.equal(o, r) for o, r in zip(outs, ref)) and all(a == b for a, b in zip(ss, ref_ss)),
                 desc + f" one-pass {bw}x{bh} sigma={sg} exact={exact}")
            # the two calls with FNX_BLUR_KEEP_BOX_SUMS on the first: same bytes, same scores (kept sums or not)
            kouts = [torch.empty_like(t) for t in d]
            ctx.plan_blur_batch(d, sg, outs=kouts, exact=exact, keep_box_sums=True).run()
            kss = ctx.plan_ssim_fast_batch(d, kouts).run().copy() if max(bw, bh) > 512 else ref_ss
            ctx.sync()
            case("two_call_keep", all(torch.equal(o, r) for o, r in zip(kouts, ref)) and all(a == b for a, b in zip(kss, ref_ss)),
                 desc + f" keep {bw}x{bh} sigma={sg} exact={exact} route={ctx.last_kernel(2)}")
        case("one_pass_oracle", abs(ss[0] - orc.ssim_fast(imgs[0], outs[0].cpu().numpy(), procs=16)) <= 1e-9 and
             np.array_equal(outs[0].cpu().numpy(), orc.gaussian_blur(imgs[0], sg, procs=16)), desc + f" one-pass {bw}x{bh} sigma={sg}")

print(f"seed {seed}: {it} iterations in {budget:.0f} s, cases {runs}, failures: {len(fails)}")
for f in fails[:20]:
    print("  ", f)
sys.exit(1 if fails else 0)
