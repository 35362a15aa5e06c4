#!/usr/bin/env python3
"""Turn gpurun_out/profile_<tag>/ (tools/collect_profiles.sh) into the files committed under
profiles/: <tag>_kernel_stats.csv (rocprofv3 --stats), <tag>_traffic.json (PMC bytes per launch of
the batched kernels, with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md "HBM") and
<tag>_sq_counters.txt."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"profile_{tag}")
# FNX_PROFILES_DST: somewhere else (tools/collect_round.sh summarises ON the GPU box into gpurun_out/profiles_new/ and deletes
# the raw counter files: a round's raw passes are > 64 MiB, more than gpurun merges back)
dst = os.environ.get("FNX_PROFILES_DST") or os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))


def per_launch(dirname, counter):
    """{kernel: (mean counter value over the LARGEST-grid dispatches, n)}"""
    out = {}
    for f in glob.glob(os.path.join(src, dirname, "**", "*counter_collection.csv"), recursive=True):
        rows = collections.defaultdict(lambda: collections.defaultdict(float))
        grid = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            rows[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
            grid[(k, r["Dispatch_Id"])] = int(r["Grid_Size"])
        for k, d in rows.items():
            gmax = max(grid[(k, i)] for i in d)
            vals = [v for i, v in d.items() if grid[(k, i)] == gmax]
            out[k] = (sum(vals) / len(vals), len(vals))
    return out


fetch = per_launch("pmc_fetch", "FETCH_SIZE")
write = per_launch("pmc_write", "WRITE_SIZE")
traffic = {"_note": "rocprofv3 --pmc, separate passes; KB units; FETCH_SIZE doubled (gfx950 counts 128-B "
                    "requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE uncorrected; "
                    "per launch (images per launch: see images_per_launch)", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    fkb = fetch.get(k, (0, 0))[0]
    wkb = write.get(k, (0, 0))[0]
    traffic["kernels"][k] = {"FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb,
                             "hbm_bytes_per_launch": (2 * fkb + wkb) * 1024.0,
                             "launches_averaged": fetch.get(k, (0, 0))[1]}
traffic["images_per_launch"] = int(sys.argv[2]) if len(sys.argv) > 2 else 32   # bench.py --batch of the profiled run
json.dump(traffic, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)

with open(os.path.join(dst, f"{tag}_sq_counters.txt"), "w") as fo:
    for c in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
              "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"):
        for k, (v, n) in sorted(per_launch("pmc_sq", c).items()):
            if "fnx::" in k or "fnx_" in k:
                fo.write(f"{c:24s} {v:16.0f}  (mean of {n} launches)  {k[:90]}\n")
    # r4: the matrix pipe's share (its own pass: eight SQ slots per pass)
    for c in ("SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE",
              "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_BUSY_CYCLES"):
        for k, (v, n) in sorted(per_launch("pmc_sq2", c).items()):
            if "fnx::" in k or "fnx_" in k:
                fo.write(f"{c:24s} {v:16.0f}  (mean of {n} launches)  {k[:90]}\n")
# shader clock during the profiled launches: GRBM_GUI_ACTIVE is summed over the 8 XCD rows of a dispatch; the
# launch durations come from the same pass's kernel trace
try:
    gui = per_launch("pmc_write", "GRBM_GUI_ACTIVE")
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "pmc_write", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    with open(os.path.join(dst, f"{tag}_sq_counters.txt"), "a") as fo:
        for k, (v, n) in sorted(gui.items()):
            if ("fnx::" in k or "fnx_" in k) and dur.get(k):
                big = sorted(dur[k])[len(dur[k]) // 2:]            # the batched launches (largest half)
                ms = sum(big) / len(big)
                fo.write(f"{'GRBM_GUI_ACTIVE_PER_XCD_PER_MS':24s} {v / 8.0 / ms:16.0f}  (mean of {n} launches)  {k[:90]}\n")
except Exception as e:                                               # the counter is optional
    print("no GRBM clock:", e)
for name in ("bench_plain.json", "bench_under_stats.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
print(open(os.path.join(dst, f"{tag}_traffic.json")).read()[:1500])
