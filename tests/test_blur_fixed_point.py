"""The arithmetic of the matrix-pipe blur (csrc/blur_mfma.hip) on the CPU: its 24-bit fixed-point weights, the bound its
rounding guard rests on, and -- replayed in exact integers -- the claim that a sample the guard does not flag rounds to the
byte the reference's clampF(fp64 chain) gives (effects.go:169-191, convert.go:149-158).  No GPU: fnx_blur_fixed_point is
host code; the oracle is the checker."""
from fractions import Fraction

import numpy as np
import pytest

import fennec_amd


def _kernel(sigma):
    r, k = fennec_amd.blurKernel(sigma)
    return r, np.asarray(k, dtype=np.float64)


@pytest.mark.parametrize("sigma", [0.9, 1.0, 1.3, 1.5, 1.67, 1.9, 2.0, 2.4, 4.0, 8.0, 12.0, 18.0, 20.6])
def test_quantised_weights(sigma):
    r, k = _kernel(sigma)
    got = fennec_amd.blur_fixed_point(k)
    assert got is not None, "a GaussianBlur table of radius <= 62 is the matrix kernels' (r5: 7 .. 62 too)"
    wq, err255 = got
    assert len(wq) == 2 * r + 1 and int(wq.sum()) == 1 << 24
    assert (wq >= 0).all() and int(wq.max()) <= 8355711           # three signed base-256 digits
    assert np.array_equal(wq, wq[::-1])                           # the table's symmetry survives (the centre takes the remainder)
    diffs = [Fraction(int(q)) - Fraction(float(w)) * (1 << 24) for q, w in zip(wq, k)]
    exact = 255 * max(sum(d for d in diffs if d > 0), -sum(d for d in diffs if d < 0))
    assert abs(Fraction(err255) - exact) < Fraction(1, 1000)      # the bound is what the header says it is: bytes are >= 0,
    assert err255 < (2 * r + 1) * 255 / 2 + 16 * 255               # so the positive and the negative differences cannot both act


def test_tables_outside_the_kernel():
    assert fennec_amd.blur_fixed_point([0.25, 0.5, 0.25]) is None          # a weight the three digits do not reach
    assert fennec_amd.blur_fixed_point([0.3, 0.3, 0.3]) is None            # sum != 1
    assert fennec_amd.blur_fixed_point([-0.1, 0.3, 0.6, 0.3, -0.1]) is None
    r, k = _kernel(21.0)                                                    # radius 63: past the widest frame
    assert r == 63 and fennec_amd.blur_fixed_point(k) is None
    k6 = _kernel(2.0)[1]
    assert fennec_amd.blur_fixed_point(k6 * 0.5) is None


def _clampf(x: float) -> int:
    t = np.trunc(x)
    if abs(x - t) >= 0.5:
        t += np.copysign(1.0, x)
    return int(min(max(t, 0.0), 255.0))


@pytest.mark.parametrize("sigma", [1.0, 1.5, 2.0, 3.0, 8.0, 14.0, 20.0])
def test_unflagged_samples_round_like_the_reference(sigma):
    """For windows of bytes p: S = sum wq[k] p[k] (exact integer).  If (S + 2^23) mod 2^24 is at least G = ceil(err255) + 2
    away from both ends, (S + 2^23) >> 24 must be clampF of the reference's left-to-right fp64 chain.  Random windows, and
    windows built to sit next to the rounding boundary."""
    r, k = _kernel(sigma)
    wq, err255 = fennec_amd.blur_fixed_point(k)
    G = int(np.ceil(err255)) + 2
    rng = np.random.default_rng(int(sigma * 100))
    n = 2 * r + 1
    win = rng.integers(0, 256, size=(200000, n), dtype=np.int64)
    # near-tie windows: nudge a random window until its fixed-point fraction is within 4 G of a boundary
    near = []
    for row in rng.integers(0, 256, size=(4000, n), dtype=np.int64):
        row = row.copy()
        for _ in range(64):
            frac = (int((wq * row).sum()) + (1 << 23)) & 0xFFFFFF
            if frac < 4 * G or frac > (1 << 24) - 4 * G:
                break
            j = int(rng.integers(0, n))
            row[j] = min(255, max(0, row[j] + int(rng.integers(-3, 4))))
        near.append(row)
    win = np.concatenate([win, np.array(near)])
    S = (win * wq[None, :]).sum(axis=1)
    frac = (S + (1 << 23)) & 0xFFFFFF
    byte = (S + (1 << 23)) >> 24
    unflagged = (frac >= G) & (frac < (1 << 24) - G)
    assert unflagged.mean() > 0.97 and (~unflagged).sum() > 0, "both branches must be exercised"
    acc = np.zeros(len(win))
    for t in range(n):                                           # r += float64(pix) * wt, taps ascending (effects.go:181)
        acc = acc + win[:, t].astype(np.float64) * k[t]
    ref = np.array([_clampf(v) for v in acc])
    bad = np.nonzero(unflagged & (byte != ref))[0]
    assert len(bad) == 0, (win[bad[:3]], byte[bad[:3]], ref[bad[:3]])
    # and the guard is not vacuous: among flagged samples the provisional byte does differ sometimes or at least sits at a boundary
    fl = np.nonzero(~unflagged)[0]
    assert (np.minimum(frac[fl], (1 << 24) - frac[fl]) < G).all()
