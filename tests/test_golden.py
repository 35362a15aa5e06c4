"""Golden fixtures (tests/golden/, produced by make_golden.py from the oracle).

CPU: the oracle still reproduces them (regression pin).  GPU (-m gpu): the HIP path,
through the C ABI, reproduces them -- bit-exact for the integer ops, 1e-9 for the SSIM family,
<= 1 LSB / <= 0.1 % for fast-mode GaussianBlur (compared with the stored whole images where
available, else exact mode against the hash).
"""
import hashlib
import importlib.util
import json
import os
import re

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)

GOLDEN = json.load(open(os.path.join(HERE, "golden", "golden.json")))
SMALL = np.load(os.path.join(HERE, "golden", "golden_small.npz"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", list(mg.INPUTS))
def test_oracle_reproduces_golden(orc, name):
    img = mg.make_input(name)
    exp = GOLDEN["cases"][name]
    assert sha(img) == exp["input_sha256"]
    for cid, kind, val in mg.cases(name, img):
        if kind == "image":
            assert list(val.shape) == exp[cid]["shape"] and sha(val) == exp[cid]["sha256"], (name, cid)
        elif kind == "stats":
            assert val == exp[cid]["stats"], (name, cid)
        else:
            assert float(val).hex() == exp[cid]["hex"], (name, cid)
    for key in SMALL.files:
        n, cid = key.split("|")
        if n == name:
            got = dict((c, v) for c, k, v in mg.cases(name, img) if k == "image")[cid]
            assert np.array_equal(got, SMALL[key])


def _hip_case(ctx, img, blurred, cid):
    """Run one golden case id on the HIP path."""
    m = re.match(r"(\w+)(?:\[(.*)\])?", cid)
    op, arg = m.group(1), m.group(2)
    if op == "gaussian_blur":
        return ctx.GaussianBlur(img, float(arg), exact=True)
    if op == "blur3x3":
        return ctx.blur3x3(img)
    if op == "sharpen":
        return ctx.Sharpen(img, float(arg))
    if op == "adaptive_sharpen":
        return ctx.AdaptiveSharpen(img, float(arg))
    if op == "lanczos_resize":
        w, h = map(int, arg.split(","))
        return ctx.lanczosResize(img, w, h)
    if op == "box_downsample":
        w, h = map(int, arg.split(","))
        return ctx.boxDownsample(img, w, h)
    if op == "apply_orientation":
        return ctx.ApplyOrientation(img, int(arg))
    if op == "ssim":
        return ctx.SSIM(img, blurred)
    if op == "ssim_fast":
        return ctx.SSIMFast(img, blurred)
    if op == "msssim":
        return ctx.MSSSIM(img, blurred)
    if op == "apply_palette":
        idx, quant = ctx.applyPalette(img, mg.golden_palette(int(arg)))
        return idx if cid.endswith(".indices") else quant
    if op == "ycbcr_to_nrgba":
        from fennec_amd import synth
        h, w = img.shape[:2]
        y, cb, cr = synth.ycbcr_planes(w, h, int(arg), 1000 + w)
        return ctx.ycbcrToNRGBA(y, cb, cr, int(arg))
    raise KeyError(cid)


def _check_stats(ctx, img, want):
    """fnx_analyze against the stored Analyze accumulators: integers exact, the two order-dependent
    sums within the serial chain's own error bound (see tests/test_gpu_parity.py)."""
    a = ctx.analyze_raw(img)
    h, w = img.shape[:2]
    assert sha(a["histogram"].astype(np.uint64)) == want["histogram_sha256"]
    for k in ("has_alpha", "is_grayscale", "unique_colors", "sample_count", "edge_count", "edge_total"):
        assert a[k] == want[k], k
    bs, vs = float.fromhex(want["bright_sum"]), float.fromhex(want["variance_sum"])
    assert abs(a["bright_sum"] - bs) <= max(1e-12, w * h * 2.0 ** -53) * abs(bs)
    assert abs(a["variance_sum"] - vs) <= 1e-9 * abs(vs) + 1e-6
    st = ctx.Analyze(img)
    assert abs(st["Entropy"] - float.fromhex(want["entropy"])) <= 1e-12
    assert st["EdgeDensity"] == float.fromhex(want["edge_density"])
    assert (st["RecommendedFormat"], st["RecommendedQuality"]) == (want["recommended_format"], want["recommended_quality"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mg.INPUTS))
def test_hip_reproduces_golden(name):
    import fennec_amd
    ctx = fennec_amd.Context(0)
    img = mg.make_input(name)
    exp = GOLDEN["cases"][name]
    blurred = ctx.GaussianBlur(img, 1.2, exact=True)      # bit-exact stand-in for the oracle's blur
    for cid, e in exp.items():
        if cid == "input_sha256":
            continue
        if "stats" in e:
            _check_stats(ctx, img, e["stats"])
            continue
        got = _hip_case(ctx, img, blurred, cid)
        if "sha256" in e:
            assert list(got.shape) == e["shape"] and sha(got) == e["sha256"], (name, cid)
        else:
            assert abs(got - float.fromhex(e["hex"])) <= 1e-9, (name, cid, got, e["repr"])
    for key in SMALL.files:
        n, cid = key.split("|")
        if n == name and cid.startswith("gaussian_blur"):
            fast = ctx.GaussianBlur(img, float(cid[cid.index("[") + 1:-1]))
            diff = np.abs(fast.astype(np.int16) - SMALL[key].astype(np.int16))
            assert diff.max() <= 1 and (diff != 0).mean() <= 1e-3
