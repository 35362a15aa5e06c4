"""CPU tests of the C-ABI library: it builds, loads, exports every symbol that
include/fennec_hip.h declares, and fails loudly without a GPU (no CPU fallback).
Also the host-side table generators (fennec_* layer) against the oracle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import fennec_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(fennec_amd.LIB_PATH):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "fennec_amd", "csrc")])
    return fennec_amd.load_library()


def test_exports_every_declared_symbol(lib):
    names = fennec_amd.exported_symbols()
    assert len(names) >= 40
    raw = ctypes.CDLL(fennec_amd.LIB_PATH)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing
    assert b"gfx950" in lib.fnx_version()


def test_only_abi_symbols_are_exported():
    out = subprocess.check_output(["nm", "-D", "--defined-only", fennec_amd.LIB_PATH]).decode()
    syms = [l.split()[-1] for l in out.splitlines() if " T " in l]
    stray = [s for s in syms if not (s.startswith("fnx_") or s.startswith("fennec_") or s in ("_init", "_fini"))]
    assert not stray, stray


def test_no_device_fails_loudly(lib):
    if lib.fnx_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(fennec_amd.FennecError, match="no HIP device"):
        fennec_amd.Context(0)
    with pytest.raises(fennec_amd.FennecError):
        fennec_amd.GaussianBlur(fennec_amd.synth.make_test_image(16, 16), 2.0)


def test_product_does_not_import_oracle():
    """The product package must not reach into oracle/ (test infrastructure)."""
    pkg = os.path.join(ROOT, "fennec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.replace("no CPU", ""), os.path.join(dirpath, f)


RELEASE_ENV = {"FENNEC_HIP_DISABLE", "FENNEC_HIP_DEVICES", "FNX_ROCTX", "FNX_POOL_TRACE", "FNX_JPEG_TRACE"}


def test_release_build_reads_five_environment_names():
    """a release build reads the five names include/fennec_hip.h lists and no other.  Every other
    switch goes through dev_env() (getenv only under -DFNX_DEVELOP) or is a per-ctx kernel-form selection (fnx_ctx_set_form)."""
    import re
    csrc = os.path.join(ROOT, "fennec_amd", "csrc")
    seen = set()
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".cpp", ".hip", ".hpp")):
            continue
        text = open(os.path.join(csrc, f)).read()
        text = re.sub(r"#ifdef FNX_DEVELOP.*?#e(?:lse|ndif)", "", text, flags=re.S)      # development-only blocks
        for m in re.finditer(r"(?<![_a-z])(?:std::)?getenv\(\s*(\"[A-Z0-9_]+\"|[a-z_]+)\s*\)", text):
            seen.add(m.group(1).strip('"'))
    assert seen == RELEASE_ENV, seen ^ RELEASE_ENV
    header = open(os.path.join(ROOT, "include", "fennec_hip.h")).read()
    for name in RELEASE_ENV:
        assert name in header, name
    # and the shipped library holds no other FNX_ / FENNEC_ environment name as a string
    blob = open(fennec_amd.LIB_PATH, "rb").read()
    names = set(m.decode() for m in re.findall(rb"(?<![A-Z0-9_])(?:FNX|FENNEC)_[A-Z0-9_]{3,}(?![A-Za-z0-9_])", blob))
    allowed = RELEASE_ENV | {n for n in names if n.startswith(("FNX_ERR", "FNX_BLUR_", "FNX_HOST", "FNX_DEVICE", "FNX_PROF", "FNX_JPEG_HOST_MAX", "FNX_BATCH_MAX"))}
    assert names <= allowed, sorted(names - allowed)


def test_set_form_needs_a_ctx_and_a_known_name(lib):
    assert lib.fnx_ctx_set_form(None, b"fx_ref", b"1") == fennec_amd.FNX_ERR_INVALID


def test_table_generators_match_oracle(lib, orc):
    assert np.array_equal(fennec_amd.gaussianKernel(), orc.gaussian_kernel())
    for s in (0.3, 1.0, 2.0, 2.5, 20.0):
        r, k = fennec_amd.blurKernel(s)
        ro, ko = orc.blur_kernel(s)
        assert r == ro and np.array_equal(k, ko)
    for d, s in [(1920, 3840), (3840, 1920), (512, 3840), (50, 100), (7, 100), (100, 7), (1, 1)]:
        off, idx, wt = fennec_amd.precomputeWeights(d, s)
        o2, i2, w2 = orc.precompute_weights(d, s)
        assert np.array_equal(off, o2) and np.array_equal(idx, i2) and np.array_equal(wt, w2)
    for x in (0.0, 0.5, -1.25, 2.999, 3.0, 7.0):
        assert lib.fennec_lanczosKernel(x) == orc.lanczos_kernel(x)


def test_dims_helpers_match_oracle(lib, orc):
    nw, nh = ctypes.c_int(), ctypes.c_int()
    for w, h in [(3840, 2160), (7680, 4320), (640, 480), (512, 512), (513, 3), (100, 100), (2000, 3)]:
        r = lib.fennec_ssimFastDims(w, h, ctypes.byref(nw), ctypes.byref(nh))
        assert (bool(r), nw.value, nh.value) == orc.ssim_fast_dims(w, h)
    for args in [(1000, 500, 200, 200), (1000, 500, 2000, 2000), (1000, 500, 0, 100), (3, 1000, 2, 2)]:
        r = lib.fennec_smartResizeDims(*args, ctypes.byref(nw), ctypes.byref(nh))
        assert (bool(r), nw.value, nh.value) == orc.smart_resize_dims(*args)


def test_host_helpers_survive_null_pointers(lib):
    """the pure host helpers of the fennec_* layer (no ctx, no GPU) with NULL where a pointer is expected: an error or a
    no-op, never a dereference"""
    C = ctypes
    assert lib.fennec_smartResizeDims(4000, 3000, 1920, 1080, None, None) < 0 and lib.fnx_last_error()
    assert lib.fennec_ssimFastDims(3840, 2160, None, None) < 0
    lib.fennec_gaussianKernel(8, 1.5, None)
    n = lib.fennec_precomputeWeights(50, 100, None, None, None)
    off = np.zeros(51, np.int32)
    idx = np.zeros(n, np.int32)
    assert lib.fennec_precomputeWeights(50, 100, off.ctypes.data_as(C.POINTER(C.c_int32)), idx.ctypes.data_as(C.POINTER(C.c_int32)), None) == n
    assert off[50] == n
    out4 = (C.c_int64 * 4)(7, 7, 7, 7)
    assert fennec_amd.load_library().fennec_Summarize(3, None, None, None, None, None, out4) == 0.0 and list(out4) == [0, 0, 0, 0]
    assert lib.fennec_SummarizeResults(3, None, out4) == 0.0
    assert lib.fennec_SummarizeResults(0, None, None) == 0.0


def test_summarize_matches_oracle(orc):
    rng = np.random.default_rng(3)
    n = 257
    failed = (rng.random(n) < 0.1).astype(np.int32)
    has = (rng.random(n) < 0.95).astype(np.int32)
    o = rng.integers(1000, 10**7, n); c = rng.integers(100, 10**6, n); s = rng.random(n)
    assert fennec_amd.Summarize(failed, has, o, c, s) == orc.summarize(failed, has, o, c, s)


def test_stats_epilogue_matches_oracle_on_cpu(orc):
    """fennec_statsFromAnalysis is host arithmetic (computeEntropy, sqrt, recommend*, analyze.go:87-230):
    fed with the oracle's own accumulators it must reproduce the oracle's ImageStats -- no GPU involved."""
    import ctypes as C
    import numpy as np
    import fennec_amd
    from fennec_amd import synth
    lib = fennec_amd.load_library()
    for img in (synth.make_test_image(200, 200), synth.make_test_image_with_alpha(100, 100),
                synth.make_solid_image(100, 100, (128, 128, 128, 255)), synth.large_photo(320, 240, 2)):
        want = orc.analyze(img)
        a = fennec_amd.Analysis()
        for i in range(256):
            a.histogram[i] = int(want["histogram"][i])
        for k in ("bright_sum", "variance_sum", "sample_count", "edge_count", "edge_total", "unique_colors",
                  "has_alpha", "is_grayscale"):
            setattr(a, k, want[k])
        st = fennec_amd.ImageStats()
        h, w = img.shape[:2]
        lib.fennec_statsFromAnalysis(C.byref(a), w, h, C.byref(st))
        assert (st.Width, st.Height, st.HasAlpha, st.IsGrayscale, st.UniqueColors) == \
            (w, h, want["has_alpha"], want["is_grayscale"], want["unique_colors"])
        assert st.Entropy == want["entropy"] and st.EdgeDensity == want["edge_density"]
        assert st.MeanBrightness == want["mean_brightness"] and st.Contrast == want["contrast"]
        assert (st.RecommendedFormat, st.RecommendedQuality, st.EstimatedCompression) == \
            (want["recommended_format"], want["recommended_quality"], want["estimated_compression"])


def test_header_is_plain_c_and_links_from_c(lib, tmp_path):
    """include/fennec_hip.h is what cgo sees: it must compile as C (no C++ in the signatures) and a C
    program must link against libfennec_hip.so and call it.  Host-only entry points here (no GPU)."""
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "fennec_hip.h"
int main(void) {
    double k[64], b[40];
    int32_t w = 0, h = 0;
    if (!strstr(fnx_version(), "gfx950")) return 1;
    fennec_gaussianKernel(8, 1.5, k);                       /* ssim.go:223 */
    double s = 0; for (int i = 0; i < 64; i++) s += k[i];
    if (s < 0.999999999 || s > 1.000000001) return 2;
    if (fennec_blurKernel(2.0, b) != 6) return 3;           /* effects.go:151: radius = ceil(3 sigma) */
    if (fennec_ssimFastDims(3840, 2160, &w, &h) != 1 || w != 512 || h != 288) return 4;   /* ssim.go:52-56 */
    fnx_ctx *c = 0;
    int st = fnx_ctx_create(0, &c);                         /* no GPU here: must fail, not fall back */
    if (st == FNX_OK) fnx_ctx_destroy(c);
    printf("%d %d\n", fnx_device_count(), st);
    return 0;
}
''')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(fennec_amd.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lfennec_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)]).decode().split()
    ndev, st = int(out[0]), int(out[1])
    assert (ndev > 0) == (st == 0)
