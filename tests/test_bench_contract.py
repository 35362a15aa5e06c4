"""The bench.py output contract, checked on the JSON lines committed under profiles/ (they are the
bench's own stdout on an MI355X; running the bench needs a GPU)."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_plain.json")))


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_bench_line_contract(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma", "valu") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # the headline (one-pass) lines always carry the CPU baseline; side lines collected with --no-cpu-baseline (the same
    # baseline would be re-timed for nothing) may omit it
    if d["n_gpus"] == 1 and ("cpu_baseline" in d or os.path.basename(path).endswith("_onepass_bench_plain.json")):
        c = d["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert c["kind"] in ("reference", "port")
    # value is units processed / elapsed: MP/s lines must agree with ms_per_step
    if d["unit"] == "MP/s" and "images_per_step_per_gpu" in d["config"]:
        mp = d["config"]["width"] * d["config"]["height"] / 1e6 * d["config"]["images_per_step_per_gpu"] * d["n_gpus"]
        assert abs(d["value"] - mp / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]


@pytest.mark.parametrize("name", ["r01_onepass_bench_plain.json", "r02_onepass_bench_plain.json"])
def test_headline_line_is_config2(name):
    d = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
    assert d["metric"].startswith("megapixels/sec: 4K SSIMFast+GaussianBlur")
    assert d["config"]["workload"].startswith("config2") and d["roofline"]["traffic"] is not None
