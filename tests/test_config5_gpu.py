"""BASELINE config 5 assembled on the GPU (-m gpu): CompressBatch over synthetic 4K JPEGs with the SSIM-guided quality
search (batch.go:58-158, compress.go:21-87) -- worker pool, one fnx ctx per worker, every candidate scored by the
HIP SSIMFast against a prepared reference -- checked item by item against the SAME search driven by the CPU oracle's
SSIMFast, and across two ranks (gloo, both on GPU 0) against the single-rank summary.

The codec is Pillow/libjpeg-turbo on both sides (Go's image/jpeg is not available: the chosen quality is "parity
unpinned" with respect to the reference's codec; what is pinned here is that the GPU scorer reproduces the oracle's
decisions bit for bit on identical candidates)."""
import os
import socket

import numpy as np
import pytest

import fennec_amd
from fennec_amd import batch, synth

pytestmark = pytest.mark.gpu

W, H, N_ITEMS = 3840, 2160, 8


def _jpegs(n=N_ITEMS, w=W, h=H):
    return [batch.pillow_encode(synth.large_photo(w, h, k), 92) for k in range(n)]     # "4K JPEGs, q = 92 up front"


def _run_gpu(jpegs, workers, rank=0, world=1, queue_mode="static", device_all=False):
    states = {}

    def make_state(wid):
        if wid not in states:
            states[wid] = fennec_amd.Context(0)
        return states[wid]
    work = batch.jpeg_item_work_device_all(jpegs) if device_all else batch.jpeg_item_work(jpegs)     # device_all: no host codec (3.13)
    res = batch.compress_batch(len(jpegs), work, make_state, workers=workers, rank=rank, world=world, queue_mode=queue_mode)
    for c in states.values():
        c.close()
    return res


def test_config5_compress_batch_matches_oracle_search(orc):
    jpegs = _jpegs()
    seen = []
    res = _run_gpu(jpegs, workers=4)
    assert [r.Index for r in res] == list(range(N_ITEMS)) and all(r.Err is None for r in res)
    # the same search with the oracle as the scorer (serial: the oracle threads internally)
    want_work = batch.jpeg_item_work(jpegs, ssim_fast=lambda a, b: orc.ssim_fast(a, b, procs=16))
    for r in res:
        w = want_work(r.Index, None)
        assert (r.Quality, r.steps, r.CompressedSize, r.OriginalSize) == (w.Quality, w.steps, w.CompressedSize, w.OriginalSize), r.Index
        assert abs(r.SSIM - w.SSIM) <= 1e-9, (r.Index, r.SSIM, w.SSIM)
        assert 30 <= r.Quality <= 100 and (r.SSIM >= 0.94 or r.Quality == 100)
        seen.append(w)
    got, want = batch.summarize_local(res), batch.summarize_local(seen)
    assert (got.Total, got.Succeeded, got.Failed, got.TotalSaved) == (want.Total, want.Succeeded, want.Failed, want.TotalSaved)
    assert abs(got.AvgSSIM - want.AvgSSIM) <= 1e-9
    # the same items with NO host codec (decoder + search + encoder on the device): every file is what the device encoder
    # writes for the oracle's decode of the source at the quality the device search finds -- checked against the CPU side
    res_dev = _run_gpu(jpegs, workers=4, device_all=True)
    assert all(r.Err is None and not r.host_decoded for r in res_dev)
    for r in res_dev[:3]:
        src = orc.jpeg_decode(jpegs[r.Index])
        assert r.data == orc.jpeg_encode(src, r.Quality) and r.OriginalSize == len(jpegs[r.Index])
        assert abs(r.SSIM - orc.ssim_fast(src, orc.jpeg_roundtrip(src, r.Quality), procs=16)) <= 1e-9 and r.SSIM >= 0.94
    s = orc.summarize([False] * N_ITEMS, [True] * N_ITEMS, [r.OriginalSize for r in res], [r.CompressedSize for r in res],
                      [r.SSIM for r in res])
    assert (got.Total, got.Succeeded, got.TotalSaved) == (s["Total"], s["Succeeded"], s["TotalSaved"]) and got.AvgSSIM == s["AvgSSIM"]


def _rank_main(rank, world, port, q, queue_mode="static", device_all=False):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)                       # both ranks on GPU 0 (RCCL cannot share a device: gloo)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    jpegs = _jpegs()
    res = _run_gpu(jpegs, workers=2, rank=rank, world=world, queue_mode=queue_mode, device_all=device_all)
    s = batch.summarize_distributed(res)
    q.put((rank, [(r.Index, r.Quality, r.steps, r.CompressedSize, r.SSIM) for r in res],
           (s.Total, s.Succeeded, s.Failed, s.TotalSaved, s.AvgSSIM)))
    dist.destroy_process_group()


@pytest.mark.parametrize("queue_mode,device_all", [("static", False), ("dynamic", False), ("dynamic", True)])
def test_config5_two_ranks_on_one_gpu_equal_single_rank(queue_mode, device_all):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    single = _run_gpu(_jpegs(), workers=4, device_all=device_all)
    want = batch.summarize_local(single)
    mpx = mp.get_context("spawn")
    q = mpx.Queue()
    procs = [mpx.Process(target=_rank_main, args=(r, 2, port, q, queue_mode, device_all)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    items = {}
    for rank, res, summ in got:
        if queue_mode == "static":
            assert [i for i, *_ in res] == list(range(rank, N_ITEMS, 2))       # item i -> rank i mod W, order kept
        else:
            assert [i for i, *_ in res] == sorted(i for i, *_ in res)          # ONE queue: whichever rank asked first
        for i, qual, steps, size, ssim in res:
            items[i] = (qual, steps, size, ssim)
        assert summ[:4] == (want.Total, want.Succeeded, want.Failed, want.TotalSaved)
        assert abs(summ[4] - want.AvgSSIM) <= 4e-16 * N_ITEMS                     # the all-reduce re-associates the ssim sum
    assert sorted(items) == list(range(N_ITEMS)) and sum(len(res) for _, res, _ in got) == N_ITEMS   # each item once
    for r in single:
        assert items[r.Index] == (r.Quality, r.steps, r.CompressedSize, r.SSIM), r.Index
