#!/usr/bin/env python3
"""What the worker pools do, on small images so that it runs under ThreadSanitizer in seconds: 6 threads, one fnx ctx
each, every kind of call the batch paths make (tap-table cache shared between threads, thread-local errors, result
FIFOs, plan caches, the second stream), plus deliberate argument errors on every thread."""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fennec_amd  # noqa: E402
from fennec_amd import batch, synth  # noqa: E402

N_THREADS, ROUNDS = 6, 6
errs = []


def worker(k):
    try:
        import torch
        torch.cuda.set_device(0)
        c = fennec_amd.Context(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            for r in range(ROUNDS):
                w, h = 320 + 16 * ((k + r) % 5), 200 + 8 * ((k * 3 + r) % 7)
                img = synth.large_photo(w, h, k + r)
                d = torch.from_numpy(img).cuda()
                small = c.lanczosResize(d, w // 2, h // 2)             # shared tap-table cache, per-ctx plans
                c.msssim_enqueue(d, small)
                sharp = c.AdaptiveSharpen(d, 0.5)
                c.ssim_enqueue(d, sharp)                               # second stream
                v1, v2 = c.fetch_result(), c.fetch_result()
                b = c.GaussianBlur(img, 2.0)                           # host-space call
                v3 = c.SSIMFast(img, b)
                q, s_, steps, found = c.jpeg_quality_search(img, 0.94)
                data = batch.pillow_encode(img, 90)
                out, q2, s2, st2, dims = c.jpeg_recompress(data, 0.94)   # the decoder's host side (parser, unstuffing) + its launches
                assert dims == (w, h) and out[:2] == b"\xff\xd8"
                assert 0 < v1 <= 1 and 0 < v2 <= 1 and 0 < v3 <= 1 and 1 <= q <= 100
                # r5: results that arrive through watched pinned memory (Analyze's two launches, the flat scans' per-workgroup words)
                st_d, st_h = c.Analyze(d), c.Analyze(img)
                assert st_d["UniqueColors"] == st_h["UniqueColors"] and st_d["Width"] == w
                assert c.isOpaque(d) is True and c.isGrayscale(img) is False
                assert c.last_kernel(fennec_amd.PROF_RESIZE).startswith("resize_")
                # r6: the one-call blur + score on a host image, the batched resize / MSSSIM entry points (pointer tables, one FIFO
                # entry for the batch), the per-ctx mode and form selections
                hb, hs = c.GaussianBlurSSIMFast(img, 2.0)
                assert np.array_equal(hb, c.GaussianBlur(img, 2.0, exact=True)) and 0 < hs <= 1
                ds = [d, torch.from_numpy(synth.large_photo(w, h, k + r + 1)).cuda(), sharp]
                smalls = c.lanczosResizeBatch(ds, w // 2, h // 2)
                c.msssim_batch_enqueue(ds, smalls)
                vb = c.fetch_results(3)
                assert all(0 < x <= 1 for x in vb) and abs(vb[0] - v1) <= 1e-12
                c.set_ssim_mode(bool(r & 1))
                c.set_form("resize_fused", "0" if r % 3 == 0 else None)
                assert np.array_equal(c.lanczosResize(d, w // 2, h // 2).cpu().numpy(), small.cpu().numpy())
                c.set_form("resize_fused", None)
                try:
                    c.lanczosResize(np.zeros((4, 4, 3), np.uint8), 2, 2)   # thread-local error text
                except Exception:
                    pass
        c.close()
    except Exception as e:                                             # noqa: BLE001
        errs.append((k, repr(e)))


ts = [threading.Thread(target=worker, args=(k,)) for k in range(N_THREADS)]
[t.start() for t in ts]
[t.join() for t in ts]
# CompressBatch's pool itself
jpegs = [batch.pillow_encode(synth.large_photo(640, 480, k), 92) for k in range(12)]
states = {}
res = batch.compress_batch(len(jpegs), batch.jpeg_item_work(jpegs), lambda wid: states.setdefault(wid, fennec_amd.Context(0)), workers=4)
assert all(r.Err is None for r in res), [r.Err for r in res]
# ... and its C++ twin: std::thread workers inside the library (fennec_CompressBatchNRGBA)
items = [synth.large_photo(320 + 16 * k, 240, k) for k in range(10)]
nres, nfiles, nsumm = batch.compress_batch_native(items, workers=4)
assert all(r.Err is None for r in nres) and nsumm.Succeeded == len(items)
# ... over files (fennec_CompressBatchJPEG: decoder + search + encoder per item on the workers' contexts)
jres, jfiles, jsumm = batch.compress_batch_jpeg_native(jpegs, workers=4)
assert all(r.Err is None for r in jres) and jsumm.Succeeded == len(jpegs)
# ... progressive files among them (r5: every worker runs the host-side scan decoder, jpeg_prog.cpp, for its own items)


def _progressive(img, q):
    import io
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(img[..., :3]), "RGB").save(b, "JPEG", quality=q, subsampling=2, progressive=True)
    return b.getvalue()


mixed = [(_progressive(synth.large_photo(640, 480, k), 92) if k % 2 else jpegs[k]) for k in range(12)]
pres, pfiles, psumm = batch.compress_batch_jpeg_native(mixed, workers=4)
assert all(r.Err is None for r in pres) and psumm.Succeeded == len(mixed) and not any(r.host_decoded for r in pres)
print("errors:", errs)
print("done" if not errs else "FAILED")
sys.exit(1 if errs else 0)
