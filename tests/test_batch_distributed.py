"""CompressBatch semantics (batch.go / compress.go) -- host logic on CPU, incl. the N>1 path
with world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest

from fennec_amd import batch


def test_search_lower_bound_and_presets():       # compress.go:35-43, types.go:74-91
    assert [batch.search_lower_bound(t) for t in (0.999, 0.99, 0.97, 0.94, 0.90, 0.85)] == [75, 75, 50, 30, 15, 1]
    assert batch.TARGET_SSIM["Balanced"] == 0.94 and batch.TARGET_SSIM["Lossless"] == 1.0


def test_compress_jpeg_optimal_binary_search():
    """compress.go:45-74 with a synthetic monotone quality->SSIM codec."""
    curve = lambda q: 0.80 + 0.002 * q              # SSIM as a function of quality
    enc = lambda img, q: bytes([q])
    dec = lambda data: data[0]
    for target, want in [(0.94, 70), (0.97, 85), (0.90, 50), (0.85, 25), (1.0, 100)]:
        q, s, data, steps = batch.compress_jpeg_optimal(lambda d: curve(d), None, target, enc, dec)
        lo = batch.search_lower_bound(min(target, 0.999))
        brute = next((k for k in range(lo, 101) if curve(k) >= min(target, 0.999)), 100)
        assert q == brute and data == bytes([q]) and steps <= 7, (target, q, brute)
        if brute == want:
            assert q == want
    # nothing reaches the target: bestQuality stays 100, fallback encode (compress.go:82-86)
    q, s, data, _ = batch.compress_jpeg_optimal(lambda d: 0.5, None, 0.94, enc, dec)
    assert q == 100 and s == 1.0 and data == bytes([100])


def test_pillow_codec_roundtrip():
    from fennec_amd import synth
    img = synth.make_test_image(64, 48)
    dec = batch.pillow_decode(batch.pillow_encode(img, 90))
    assert dec.shape == img.shape and dec.dtype == np.uint8 and (dec[..., 3] == 255).all()
    assert np.abs(dec[..., :3].astype(int) - img[..., :3].astype(int)).mean() < 6


def _fake_work(idx, state):
    if idx % 7 == 3:
        raise RuntimeError("decode failed")
    return batch.BatchResult(Index=idx, OriginalSize=1000 + 13 * idx, CompressedSize=400 + 5 * idx,
                             SSIM=0.9 + (idx % 10) * 0.007, Quality=50 + idx % 40)


def test_compress_batch_order_errors_progress(orc):
    seen = []
    res = batch.compress_batch(50, _fake_work, lambda w: None, workers=4, on_item=lambda c, t: seen.append((c, t)))
    assert [r.Index for r in res] == list(range(50))            # results by index (batch.go:71)
    assert sorted(c for c, _ in seen) == list(range(1, 51)) and all(t == 50 for _, t in seen)
    assert all((r.Err is not None) == (r.Index % 7 == 3) for r in res)
    s = batch.summarize_local(res)
    want = orc.summarize([r.Err is not None for r in res], [r.has_result for r in res],
                         [r.OriginalSize for r in res], [r.CompressedSize for r in res], [r.SSIM for r in res])
    assert (s.Total, s.Succeeded, s.Failed, s.TotalSaved) == (want["Total"], want["Succeeded"], want["Failed"], want["TotalSaved"])
    assert s.AvgSSIM == want["AvgSSIM"]
    assert batch.compress_batch(0, _fake_work, lambda w: None) == []     # batch.go:59-61


def _rank_main(rank, world, port, n_items, out_q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = batch.compress_batch(n_items, _fake_work, lambda w: None, workers=2, rank=rank, world=world)
    s = batch.summarize_distributed(res)
    out_q.put((rank, [r.Index for r in res], (s.Total, s.Succeeded, s.Failed, s.TotalSaved, s.AvgSSIM)))
    dist.destroy_process_group()


def test_world_size_2_gloo(orc):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_items, world = 37, 2
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    shards = {r: idx for r, idx, _ in got}
    assert sorted(shards[0] + shards[1]) == list(range(n_items))         # every item exactly once
    assert shards[0] == list(range(0, n_items, 2)) and shards[1] == list(range(1, n_items, 2))
    allres = [(_fake_work(i, None) if i % 7 != 3 else batch.BatchResult(Index=i, Err="x", has_result=False)) for i in range(n_items)]
    want = orc.summarize([r.Err is not None for r in allres], [r.has_result for r in allres],
                         [r.OriginalSize for r in allres], [r.CompressedSize for r in allres], [r.SSIM for r in allres])
    for _, _, (t, ok, bad, saved, avg) in got:                             # identical on every rank
        assert (t, ok, bad, saved) == (want["Total"], want["Succeeded"], want["Failed"], want["TotalSaved"])
        assert abs(avg - want["AvgSSIM"]) <= 1e-15 * 4


def _uneven_work(idx, state):
    """Items cost 25 ms on rank 0 (a slow device) and 1 ms on rank 1."""
    import time
    time.sleep(0.025 if state == 0 else 0.001)
    return _fake_work(idx, None)


def _rank_main_dynamic(rank, world, port, n_items, out_q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    if rank == 1:                                 # rank 1 runs a batch of its own first: the ranks' call COUNTS now differ, which the
        batch.compress_batch(5, _fake_work, lambda w: None, workers=1)       # caller-supplied batch ids below must not mind
    for chunk in (1, 3):                          # two batches in one job: the queue's key is per batch
        seen = []
        res = batch.compress_batch(n_items, _uneven_work, lambda w: rank, workers=2, rank=rank, world=world,
                                   queue_mode="dynamic", chunk=chunk, batch_id=f"job-chunk{chunk}",
                                   on_item=lambda c, t: seen.append((c, t)))
        s = batch.summarize_distributed(res)
        # OnItem reports the JOB's progress (batch.go:113-119 counts the one pool's completions): job-wide counts against the job's total
        assert len(seen) == len(res) and all(t == n_items and 1 <= c <= n_items for c, t in seen)
        assert len({c for c, _ in seen}) == len(seen)
        out.append(([r.Index for r in res], (s.Total, s.Succeeded, s.Failed, s.TotalSaved, s.AvgSSIM), [c for c, _ in seen]))
        dist.barrier()
    from torch.distributed import distributed_c10d as c10d
    store = dist.PrefixStore("fennec_batch_queue", c10d._get_default_store())
    assert not store.check(["next_job-chunk1"]) and not store.check(["next_job-chunk3_completed"])    # the last rank out removed the keys
    out_q.put((rank, out))
    dist.destroy_process_group()


def test_world_size_2_dynamic_queue(orc):
    """ONE queue for the whole job (batch.go:72-126's channel across ranks).  With uneven item cost the
    fast rank takes most of the items; every item is done exactly once; Summarize is the same on every rank and equal
    to the single-process summary."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_items, world = 61, 2
    procs = [ctx.Process(target=_rank_main_dynamic, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    allres = [(_fake_work(i, None) if i % 7 != 3 else batch.BatchResult(Index=i, Err="x", has_result=False)) for i in range(n_items)]
    want = orc.summarize([r.Err is not None for r in allres], [r.has_result for r in allres],
                         [r.OriginalSize for r in allres], [r.CompressedSize for r in allres], [r.SSIM for r in allres])
    for b in range(2):
        i0, s0, c0 = got[0][b]
        i1, s1, c1 = got[1][b]
        assert sorted(c0 + c1) == list(range(1, n_items + 1))             # OnItem: the job's count, each value once across the ranks
        assert sorted(i0 + i1) == list(range(n_items))                    # every item exactly once
        assert i0 == sorted(i0) and i1 == sorted(i1)
        assert len(i1) > 2 * len(i0), (len(i0), len(i1))                  # the fast rank took most of them
        for t, ok, bad, saved, avg in (s0, s1):
            assert (t, ok, bad, saved) == (want["Total"], want["Succeeded"], want["Failed"], want["TotalSaved"])
            assert abs(avg - want["AvgSSIM"]) <= 1e-15 * 4


def test_dynamic_queue_single_rank_is_static():
    res = batch.compress_batch(20, _fake_work, lambda w: None, workers=3, queue_mode="dynamic")
    assert [r.Index for r in res] == list(range(20))
    with pytest.raises(ValueError):
        batch.compress_batch(3, _fake_work, lambda w: None, queue_mode="ring")


def _rank_main_small(rank, world, port, n_items, chunk, out_q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = batch.compress_batch(n_items, _uneven_work, lambda w: rank, workers=2, rank=rank, world=world,
                               queue_mode="dynamic", chunk=chunk, batch_id=f"small-{n_items}-{chunk}")
    s = batch.summarize_distributed(res)
    out_q.put((rank, ([(r.Index, r.Err is not None) for r in res], (s.Total, s.Succeeded, s.Failed, s.TotalSaved, s.AvgSSIM))))
    dist.barrier()
    dist.destroy_process_group()


def _rank_main_evict(rank, world, port, out_q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 0:
        batch._RankQueue._USES_CAP = 2            # rank 0 forgets ids after two newer ones; rank 1 remembers everything
    out = []
    for name in ("A", "B", "C", "D", "A", "A", "B"):      # "A" comes back after rank 0 has evicted it
        res = batch.compress_batch(9, _fake_work, lambda w: None, workers=2, rank=rank, world=world,
                                   queue_mode="dynamic", chunk=2, batch_id=name)
        out.append([r.Index for r in res])
        dist.barrier()
    out_q.put((rank, out))
    dist.destroy_process_group()


def test_batch_id_use_counts_survive_a_rank_forgetting_them():
    """the per-process window of batch-id use counts (newest 1024) restarted an evicted id at #1 on the rank that had
    forgotten it while the others built #k -- two jobs' keys for one job.  The evicted count is parked in the store."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main_evict, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for j in range(7):
        assert sorted(got[0][j] + got[1][j]) == list(range(9)), (j, got[0][j], got[1][j])       # ONE queue per job: every item exactly once


@pytest.mark.parametrize("n_items,chunk", [(7, 4), (7, 1), (1, 4), (3, 8)])
def test_world_size_2_dynamic_queue_uneven_job_with_a_failing_item(orc, n_items, chunk):
    """An uneven job: 7 items over 2 ranks in chunks of 4 (the second chunk is partial), item 3 fails (batch.go:100-106: the
    error stays with its item, the batch goes on); 1 item for 2 ranks (one rank takes nothing and still joins Summarize);
    a chunk larger than the job.  Every item exactly once, the failure counted once, the same summary on both ranks."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main_small, args=(r, 2, port, n_items, chunk, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    items = got[0][0] + got[1][0]
    assert sorted(i for i, _ in items) == list(range(n_items))
    assert [i for i, bad in items if bad] == ([3] if n_items > 3 else [])
    allres = [(_fake_work(i, None) if i % 7 != 3 else batch.BatchResult(Index=i, Err="x", has_result=False)) for i in range(n_items)]
    want = orc.summarize([r.Err is not None for r in allres], [r.has_result for r in allres],
                         [r.OriginalSize for r in allres], [r.CompressedSize for r in allres], [r.SSIM for r in allres])
    for r in range(2):
        t, ok, bad, saved, avg = got[r][1]
        assert (t, ok, bad, saved) == (want["Total"], want["Succeeded"], want["Failed"], want["TotalSaved"])
        assert abs(avg - want["AvgSSIM"]) <= 1e-15 * 4


def test_bind_to_device_numa_is_a_no_op_without_a_gpu():
    """fennec_amd.batch.bind_to_device_numa: no device (this container) or no topology file -> None, the affinity mask untouched"""
    import os
    from fennec_amd import batch as fb
    before = os.sched_getaffinity(0)
    assert fb.bind_to_device_numa(0) is None
    assert os.sched_getaffinity(0) == before
