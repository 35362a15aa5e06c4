"""CompressBatch semantics around the GPU hot path (batch.go:58-158, compress.go:21-87).

What the reference does with goroutines in one process, done here with one process per GPU:

* items are independent (batch.go:88-122), so rank r of W takes items r, r+W, r+2W, ... --
  no data-path collective; inside a rank a small thread pool (one fennec_amd.Context each)
  plays the reference's worker pool so host codec work overlaps device work;
* results are written by index (batch.go:71,108), so order is preserved;
* the only reduction is Summarize (batch.go:140-158): per-rank partial
  {Succeeded, Failed, TotalSaved, ssimSum} -> one all-reduce (RCCL over xGMI on GPUs, gloo in
  the CPU tests).  Integer fields are reduced as int64 (exact); ssimSum as float64, whose
  cross-rank summation order differs from the reference's index order by <= 1 ulp per add.

`compress_jpeg_optimal` restates the SSIM-guided quality binary search (compress.go:21-87)
with the JPEG codec as a pluggable pair of callables (Pillow/libjpeg-turbo by default; Go's
image/jpeg is not available here, so the chosen quality is "parity unpinned") and SSIMFast
evaluated on the GPU against a prepared reference (the search compares every candidate with
the same source image).
"""
from __future__ import annotations

import io
import queue
import collections
import threading
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

# Quality presets -> target SSIM (types.go:74-91)
TARGET_SSIM = {"Lossless": 1.0, "Ultra": 0.99, "High": 0.97, "Balanced": 0.94, "Aggressive": 0.90,
               "Maximum": 0.85}


def search_lower_bound(target_ssim: float) -> int:
    """compress.go:35-43: where the binary search starts."""
    if target_ssim >= 0.99:
        return 75
    if target_ssim >= 0.97:
        return 50
    if target_ssim >= 0.94:
        return 30
    if target_ssim >= 0.90:
        return 15
    return 1


def pillow_encode(rgba: np.ndarray, quality: int) -> bytes:
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(rgba[..., :3]), "RGB").save(buf, "JPEG", quality=int(quality), subsampling=2)
    return buf.getvalue()


def pillow_decode(data: bytes) -> np.ndarray:
    from PIL import Image
    rgb = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    out = np.empty(rgb.shape[:2] + (4,), dtype=np.uint8)
    out[..., :3] = rgb
    out[..., 3] = 255
    return out


def compress_jpeg_optimal(ssim_against: Callable[[np.ndarray], float], src: np.ndarray, target_ssim: float,
                          encode: Callable[[np.ndarray, int], bytes] = pillow_encode,
                          decode: Callable[[bytes], np.ndarray] = pillow_decode):
    """compress.go:21-87.  `ssim_against(decoded)` is SSIMFast(src, decoded) (compress.go:62).
    Returns (quality, ssim, data, steps)."""
    if target_ssim >= 1.0:
        target_ssim = 0.999                       # compress.go:24-26
    lo, hi = search_lower_bound(target_ssim), 100
    best_q, best_ssim, best_data = hi, 1.0, None
    steps = 0
    while lo <= hi:
        mid = (lo + hi) // 2
        data = encode(src, mid)
        s = ssim_against(decode(data))
        steps += 1
        if s >= target_ssim:
            best_q, best_ssim, best_data = mid, s, data
            hi = mid - 1
        else:
            lo = mid + 1
    if best_data is None:                          # compress.go:82-86 fallback
        best_data = encode(src, best_q)
    return best_q, best_ssim, best_data, steps


@dataclass
class BatchResult:
    """batch.go:20-30 (Result reduced to the fields Summarize reads, types.go:221-297)."""
    Index: int
    Err: Optional[str] = None
    OriginalSize: int = 0
    CompressedSize: int = 0
    SSIM: float = 0.0
    Quality: int = 0
    has_result: bool = True
    steps: int = 0          # search steps taken (diagnostics; not part of the reference's Result)


def bind_to_device_numa(device: int = 0) -> Optional[str]:
    """One process per GPU: keep this process's threads (and, by first touch, its host buffers) on the NUMA node the GPU's PCIe
    root hangs off (/sys/bus/pci/devices/<bdf>/local_cpulist).  CompressBatch moves every file up and every result down through
    pageable host memory; on a two-socket box a worker pool that lands on the far socket pays the inter-socket link on each of
    them.  (Standard placement, not a measured gain: on the box it was A/B'd on -- GPU on node 1 of 2 -- `batch` read 2 444-2 532
    images/s with it and 2 473-2 517 without.)  Returns
    the CPU list it bound to, or None (no such file, one node, FENNEC_NO_NUMA_BIND=1, or an affinity mask already set)."""
    import os
    if os.environ.get("FENNEC_NO_NUMA_BIND") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        import torch
        pr = torch.cuda.get_device_properties(device)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            text = f.read().strip()
        cpus = set()
        for part in text.split(","):
            if not part:
                continue
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        have = os.sched_getaffinity(0)
        if not cpus or not (cpus & have) or len(have) < (os.cpu_count() or 0):     # somebody (taskset, a launcher) chose already
            return None
        if cpus >= have:
            return None                                                              # one node
        os.sched_setaffinity(0, cpus & have)
        return text
    except Exception:
        return None


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Items of rank `rank`: r, r+W, ... (independent units, batch.go:88-122)."""
    return list(range(rank, n_items, world))


class _RankQueue:
    """The closed channel of batch.go:72-81 stretched over the ranks of one job: a counter in torch.distributed's
    store hands out the next `chunk` indices to whichever worker of whichever rank asks first, so a rank with slower
    items (or a slower GPU) simply takes fewer -- no data-path collective, one small TCP round trip per chunk."""
    _seq = 0
    _uses: "collections.OrderedDict[str, int]" = collections.OrderedDict()   # per-process use count of each batch_id, newest last
    _USES_CAP = 1024                                                          # ids remembered (a service minting a fresh id per job)

    def __init__(self, n_items: int, chunk: int = 1, group=None, batch_id: Optional[str] = None, world: Optional[int] = None):
        import torch.distributed as dist
        from torch.distributed import distributed_c10d as c10d

        self.n, self.chunk = n_items, max(1, int(chunk))
        self.world = int(world) if world else dist.get_world_size(group)   # the job's ranks: who is "last to leave" in close()
        # The counter's name must be the same batch on every rank.  A caller-supplied batch_id says so outright (use one
        # whenever ranks may call compress_batch a different number of times -- an exception before a batch, a mix of
        # static and dynamic calls); without it the per-process count of DYNAMIC batches names it, which holds as long as
        # every rank runs the same sequence of dynamic batches.
        if batch_id is None:
            _RankQueue._seq += 1
            batch_id = f"seq{_RankQueue._seq}"
        else:
            # a batch_id names ONE job.  Reused for back-to-back jobs without a barrier, a fast rank could re-enter while a
            # slow one has not left: it would read the exhausted counter, take nothing, and the slow rank's close() would
            # then wipe keys both jobs share.  The per-process use count keeps the jobs apart (every rank calls
            # compress_batch with the same id the same number of times -- the contract batch_id had already).
            # CONTRACT: a rank that calls with an id k times must be matched by k calls on every other rank; a rank that
            # retries a batch on its own (an exception mid-job) must mint a NEW id on all ranks, or it would build 'x#2'
            # against the others' 'x#1' and drain that job alone.
            # The count lives in this process (newest _USES_CAP ids) and, once an id falls out of that window, in the job's
            # store under (id, rank): forgetting it would restart the id at #1 on THIS rank only while ranks that still
            # remember it build '#k' -- the very collision the count exists to prevent.  Ranks evict
            # independently; the store copy makes what they evict irrelevant.
            store0 = dist.PrefixStore("fennec_batch_uses", c10d._get_default_store())
            rank = dist.get_rank(group)
            use = _RankQueue._uses.pop(batch_id, None)
            if use is None:
                k = f"{batch_id}@r{rank}"
                use = int(store0.get(k)) if store0.check([k]) else 0
            use += 1
            _RankQueue._uses[batch_id] = use
            while len(_RankQueue._uses) > _RankQueue._USES_CAP:     # ids are usually unique per job: park the oldest in the store
                old_id, old_use = _RankQueue._uses.popitem(last=False)
                store0.set(f"{old_id}@r{rank}", str(old_use))
            batch_id = f"{batch_id}#{use}"
        self.key = f"next_{batch_id}"
        self.store = dist.PrefixStore("fennec_batch_queue", c10d._get_default_store())
        self.lock = threading.Lock()

    def take(self) -> List[int]:
        with self.lock:
            end = int(self.store.add(self.key, self.chunk))
        return list(range(end - self.chunk, min(end, self.n)))

    def completed(self) -> int:
        """One more item of the JOB is done: the job-wide count (what OnItem reports against the job-wide total)."""
        with self.lock:
            return int(self.store.add(self.key + "_completed", 1))

    def close(self):
        """The last rank to leave removes the batch's keys from the store (no barrier: a counter says who is last)."""
        with self.lock:
            if int(self.store.add(self.key + "_left", 1)) == self.world:
                for k in (self.key, self.key + "_completed", self.key + "_left"):
                    try:
                        self.store.delete_key(k)
                    except Exception:          # a store without delete_key: the keys stay, a few bytes per batch
                        pass


def compress_batch(n_items: int, work: Callable[[int, object], BatchResult], make_worker_state: Callable[[int], object],
                   workers: int = 1, rank: int = 0, world: int = 1,
                   on_item: Optional[Callable[[int, int], None]] = None, queue_mode: str = "static",
                   chunk: int = 1, batch_id: Optional[str] = None) -> List[BatchResult]:
    """CompressBatch (batch.go:58-128) for this rank: a closed queue of indices drained by `workers` threads, results
    stored by index, `on_item(completed, total)` under a lock.

    queue_mode "static": the rank owns items i = rank (mod world) (shard_indices) -- no communication at all.
    queue_mode "dynamic" (world > 1, torch.distributed initialised): ONE queue for the whole job (_RankQueue), which
    is what batch.go's channel is to its goroutines; ranks return the items they happened to take, by index.  `batch_id`
    names the job's counter in the store (see _RankQueue): every rank must call with the same id the same number of times
    (a retry of a failed batch takes a new id on all ranks); `on_item` then reports the JOB's completed count against
    the job's total on every rank (one store round trip per item), in static mode the rank's own."""
    if n_items <= 0:
        return []
    dynamic = queue_mode == "dynamic" and world > 1
    if queue_mode not in ("static", "dynamic"):
        raise ValueError("queue_mode must be 'static' or 'dynamic'")
    mine = list(range(n_items)) if dynamic else shard_indices(n_items, rank, world)
    if not mine:
        return []
    workers = max(1, min(workers, len(mine)))       # batch.go:63-69
    q: "queue.Queue[int]" = queue.Queue()
    rq = _RankQueue(n_items, chunk, batch_id=batch_id, world=world) if dynamic else None
    if not dynamic:
        for i in mine:
            q.put(i)
    results: dict = {}
    lock = threading.Lock()
    done = [0]
    total = n_items if dynamic else len(mine)

    def next_index():
        while True:
            try:
                return q.get_nowait()
            except queue.Empty:
                if rq is None:
                    return None
                got = rq.take()
                if not got:
                    return None
                for i in got[1:]:
                    q.put(i)
                return got[0]

    def run(wid: int):
        state = make_worker_state(wid)
        while True:
            idx = next_index()
            if idx is None:
                return
            try:
                r = work(idx, state)
            except Exception as e:                   # per-item error capture (batch.go:108-113)
                r = BatchResult(Index=idx, Err=f"{type(e).__name__}: {e}", has_result=False)
            results[idx] = r
            if on_item:
                if rq is not None:
                    c = rq.completed()
                else:
                    with lock:
                        done[0] += 1
                        c = done[0]
                on_item(c, total)

    threads = [threading.Thread(target=run, args=(w,)) for w in range(workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if rq is not None:
        rq.close()
    return [results[i] for i in sorted(results)]


def jpeg_item_work_device_codec(jpegs: Sequence[bytes], target_ssim: float = TARGET_SSIM["Balanced"],
                                on_gpu_seconds: Optional[Callable[[float], None]] = None,
                                decode: Callable[[bytes], np.ndarray] = pillow_decode):
    """The per-item body of CompressBatch with the search AND the encoder on the device (fnx_jpeg_compress: SURVEY 8(f)2,
    second slice): the host codec only decodes the source.  The output is the file jpeg.Encode would write at the
    chosen quality, as far as that is restated."""
    import time

    def work(idx: int, state) -> BatchResult:
        data = jpegs[idx]
        src = decode(data)
        t0 = time.perf_counter()
        out, q, s_, steps = state.jpeg_compress(src, target_ssim)
        if on_gpu_seconds is not None:
            on_gpu_seconds(time.perf_counter() - t0)
        r = BatchResult(Index=idx, OriginalSize=len(data), CompressedSize=len(out), SSIM=s_, Quality=q)
        r.steps = steps
        r.data = out
        return r
    return work


def jpeg_item_work_device_all(jpegs: Sequence[bytes], target_ssim: float = TARGET_SSIM["Balanced"],
                              on_gpu_seconds: Optional[Callable[[float], None]] = None,
                              decode: Callable[[bytes], np.ndarray] = pillow_decode, file_options: Optional[dict] = None):
    """The per-item body of CompressBatch with NO host codec (fnx_jpeg_recompress: SURVEY 8(f)2, third slice): the file's
    bytes go up, the decoder, the search and the encoder run on the device, the new file's bytes come down.  A file the
    device decoder does not take (12-bit, arithmetic coding, a progressive 4:2:0 file with restart intervals: FennecUnsupported) is decoded on the host by
    THIS function -- the caller's choice, visible in the result's `host_decoded` -- and continues on the device (without
    file_options' stages: those belong to the device call)."""
    import time
    from . import FennecError

    def work(idx: int, state) -> BatchResult:
        data = jpegs[idx]
        t0 = time.perf_counter()
        host_decoded = False
        try:
            if file_options:       # CompressFile's options (orient, max_w, max_h, auto_format): fennec_CompressFileJPEG
                out, q, s_, steps, _od, _fd = state.compress_file_jpeg(data, target_ssim, **file_options)
                if out is None:    # analyzeFormat chose PNG: not this harness's path
                    r = BatchResult(Index=idx, OriginalSize=len(data), Err="analyzeFormat: PNG", has_result=False)
                    r.host_decoded = False
                    return r
            else:
                out, q, s_, steps, _dims = state.jpeg_recompress(data, target_ssim)
        except FennecError:          # refused (FNX_ERR_UNSUPPORTED) or found damaged (FNX_ERR_INVALID): the host codec's call
            host_decoded = True
            src = decode(data)
            t0 = time.perf_counter()
            out, q, s_, steps = state.jpeg_compress(src, target_ssim)
        if on_gpu_seconds is not None:
            on_gpu_seconds(time.perf_counter() - t0)
        r = BatchResult(Index=idx, OriginalSize=len(data), CompressedSize=len(out), SSIM=s_, Quality=q)
        r.steps = steps
        r.data = out
        r.host_decoded = host_decoded
        return r
    return work


def jpeg_item_work_device_search(jpegs: Sequence[bytes], target_ssim: float = TARGET_SSIM["Balanced"],
                                 on_gpu_seconds: Optional[Callable[[float], None]] = None,
                                 encode: Callable[[np.ndarray, int], bytes] = pillow_encode,
                                 decode: Callable[[bytes], np.ndarray] = pillow_decode):
    """The per-item body of CompressBatch with the quality search ON THE DEVICE (first slice): decode the
    source once, fnx_jpeg_quality_search round-trips and scores every candidate quality on the GPU (Go's image/jpeg
    arithmetic without the entropy coder), then the real codec encodes ONCE at the chosen quality -- 2 host codec
    passes per item instead of ~15 (compress.go:45-74 does an encode and a decode per candidate)."""
    import time

    def work(idx: int, state) -> BatchResult:
        data = jpegs[idx]
        src = decode(data)
        t0 = time.perf_counter()
        q, s_, steps, found = state.jpeg_quality_search(src, target_ssim)
        if on_gpu_seconds is not None:
            on_gpu_seconds(time.perf_counter() - t0)
        out = encode(src, q)                      # compress.go:76-86: bestData, or the fallback encode at bestQuality
        r = BatchResult(Index=idx, OriginalSize=len(data), CompressedSize=len(out), SSIM=s_, Quality=q)
        r.steps = steps
        return r
    return work


def jpeg_item_work(jpegs: Sequence[bytes], target_ssim: float = TARGET_SSIM["Balanced"],
                   ssim_fast: Optional[Callable] = None, on_gpu_seconds: Optional[Callable[[float], None]] = None):
    """The per-item body of CompressBatch for JPEG inputs (batch.go:88-122 -> compressJPEGOptimal,
    compress.go:21-87): decode the source JPEG, prepare it as the SSIMFast reference on the worker's ctx, run the
    quality search with every candidate scored on the GPU, return the BatchResult.

    Returns work(idx, state) for compress_batch; `state` is the worker's fennec_amd.Context.  `ssim_fast(src, dec)`
    replaces the GPU scorer (tests drive the identical search with their CPU checker)."""
    import time

    def work(idx: int, state) -> BatchResult:
        data = jpegs[idx]
        src = pillow_decode(data)
        gpu_s = 0.0
        if ssim_fast is None:
            t0 = time.perf_counter()
            prep = state.ssim_fast_prepare(src)
            gpu_s += time.perf_counter() - t0

            def score(dec):
                nonlocal gpu_s
                t1 = time.perf_counter()
                v = prep.against(dec)
                gpu_s += time.perf_counter() - t1
                return v
        else:
            prep = None

            def score(dec):
                return ssim_fast(src, dec)
        try:
            q, s_, out, steps = compress_jpeg_optimal(score, src, target_ssim)
        finally:
            if prep is not None:
                prep.close()
        if on_gpu_seconds is not None:
            on_gpu_seconds(gpu_s)
        r = BatchResult(Index=idx, OriginalSize=len(data), CompressedSize=len(out), SSIM=s_, Quality=q)
        r.steps = steps
        return r
    return work


@dataclass
class BatchSummary:
    """batch.go:131-138"""
    Total: int = 0
    Succeeded: int = 0
    Failed: int = 0
    TotalSaved: int = 0
    AvgSSIM: float = 0.0
    ssim_sum: float = field(default=0.0, repr=False)


def summarize_local(results: Sequence[BatchResult]) -> BatchSummary:
    """Summarize (batch.go:140-158) over this rank's results, index order."""
    s = BatchSummary(Total=len(results))
    for r in results:
        if r.Err is not None:
            s.Failed += 1
            continue
        s.Succeeded += 1
        if r.has_result:
            s.TotalSaved += r.OriginalSize - r.CompressedSize
            s.ssim_sum += r.SSIM
    if s.Succeeded > 0:
        s.AvgSSIM = s.ssim_sum / float(s.Succeeded)
    return s


def summarize_distributed(results: Sequence[BatchResult], device=None, group=None, force: bool = False) -> BatchSummary:
    """Summarize across all ranks: the path's one collective (two tiny all-reduces).  `force`: issue the all-reduces in a
    one-rank group too (legal, and the only way a 1-GPU box can run the RCCL branch)."""
    import torch
    import torch.distributed as dist

    s = summarize_local(results)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return s
    ints = torch.tensor([s.Total, s.Succeeded, s.Failed, s.TotalSaved], dtype=torch.int64, device=device)
    flt = torch.tensor([s.ssim_sum], dtype=torch.float64, device=device)
    dist.all_reduce(ints, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(flt, op=dist.ReduceOp.SUM, group=group)
    t, ok, bad, saved = (int(v) for v in ints.tolist())
    out = BatchSummary(Total=t, Succeeded=ok, Failed=bad, TotalSaved=saved, ssim_sum=float(flt.item()))
    if ok > 0:
        out.AvgSSIM = out.ssim_sum / float(ok)
    return out


def _device_list(device, devices):
    import ctypes as C
    devs = [int(d) for d in devices] if devices is not None else [int(device)]
    if not devs:
        raise ValueError("empty device list")
    return (C.c_int * len(devs))(*devs), len(devs)


def compress_batch_native(images: Sequence, target_ssim: float = TARGET_SSIM["Balanced"], workers: int = 4, device: int = 0,
                          original_sizes: Optional[Sequence[int]] = None, devices: Optional[Sequence[int]] = None):
    """fennec_CompressBatchNRGBA(Devices): CompressBatch's pool in C++ (std::thread workers, one fnx ctx each, one atomic
    queue of indices) with compressJPEGOptimal on the device as the per-item work.  `images`: decoded NRGBA items, all
    numpy (host space) or all torch CUDA tensors (device space).  `devices`: the node's pool -- worker i on
    devices[i mod g], one queue (batch.go:63-126); host-space items only when the list names more than one device.
    -> (results, files, summary): BatchResult per item by index (with .device), the JPEG bytes per item, BatchSummary.
    An item whose file outgrows its buffer (a q=100 fallback, a noisy photograph: > 1.5 B/px) is run again with a
    buffer of the size the library reported -- the reference never fails an item on output size."""
    import ctypes as C

    import fennec_amd as fa
    L = fa.load_library()
    n = len(images)
    if n == 0:
        return [], [], BatchSummary()
    devs, ndev = _device_list(device, devices)
    views = [fa._Img(im) for im in images]
    space = views[0].space
    if any(v.space != space for v in views):
        raise fa.FennecError("compress_batch_native: all items in one space")
    srcs = (C.c_void_p * n)(*[v.ptr for v in views])
    strides = (C.c_int * n)(*[v.stride for v in views])
    ws = (C.c_int * n)(*[v.w for v in views])
    hs = (C.c_int * n)(*[v.h for v in views])
    osz = (C.c_int64 * n)(*[int(x) for x in original_sizes]) if original_sizes is not None else None
    bufs = [np.empty(4096 + (v.w * v.h * 3) // 2, dtype=np.uint8) for v in views]
    outs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    caps = (C.c_size_t * n)(*[b.size for b in bufs])
    res = (fa.NativeBatchResult * n)()
    if space == fa.FNX_DEVICE:
        import torch
        torch.cuda.synchronize()                       # the workers' contexts launch on their own streams
    rc = L.fennec_CompressBatchNRGBADevices(devs, ndev, int(workers), n, space, srcs, strides, ws, hs, osz, float(target_ssim), outs, caps,
                                            res, None, None, None)
    if rc != fa.FNX_OK:
        raise fa.FennecError(f"fennec_CompressBatchNRGBA: {L.fnx_last_error().decode()}")
    for i in range(n):                                 # the file did not fit: once more with the size the library reported
        if res[i].failed and res[i].status == fa.FNX_ERR_INVALID and int(res[i].compressed_size) > bufs[i].size:
            big = np.empty(int(res[i].compressed_size) + 4096, dtype=np.uint8)
            one = (fa.NativeBatchResult * 1)()
            one_osz = (C.c_int64 * 1)(int(original_sizes[i])) if original_sizes is not None else None
            L.fennec_CompressBatchNRGBADevices(devs, ndev, 1, 1, space, (C.c_void_p * 1)(views[i].ptr), (C.c_int * 1)(views[i].stride),
                                               (C.c_int * 1)(views[i].w), (C.c_int * 1)(views[i].h), one_osz, float(target_ssim),
                                               (C.c_void_p * 1)(big.ctypes.data), (C.c_size_t * 1)(big.size), one, None, None, None)
            one[0].index = i
            res[i] = one[0]
            bufs[i] = big
    results, files = [], []
    for i in range(n):
        r = res[i]
        br = BatchResult(Index=r.index, OriginalSize=int(r.original_size), CompressedSize=int(r.compressed_size), SSIM=float(r.ssim),
                         Quality=int(r.quality), Err=None if not r.failed else f"status {r.status}", has_result=bool(r.has_result))
        br.steps = int(r.steps)
        br.device = int(r.device)
        results.append(br)
        files.append(bufs[i][:int(r.compressed_size)].tobytes() if not r.failed else b"")
    out4 = (C.c_int64 * 4)()
    avg = L.fennec_SummarizeResults(n, res, out4)
    summ = BatchSummary(Total=int(out4[0]), Succeeded=int(out4[1]), Failed=int(out4[2]), TotalSaved=int(out4[3]), AvgSSIM=float(avg))
    return results, files, summ


_out_arena = None      # compress_batch_jpeg_native's output pages (not thread-safe: one batch at a time per process)


def compress_batch_jpeg_native(files: Sequence[bytes], target_ssim: float = TARGET_SSIM["Balanced"], workers: int = 4, device: int = 0,
                               decode: Callable[[bytes], np.ndarray] = pillow_decode, devices: Optional[Sequence[int]] = None):
    """fennec_CompressBatchJPEG: CompressBatch over JPEG FILES (batch.go:88-122) with no host codec -- the C++ pool, per
    item decoder + quality search + encoder on the device (fnx_jpeg_recompress).  Items the device decoder refuses
    (status FNX_ERR_UNSUPPORTED) are decoded on the host HERE and sent through fennec_CompressBatchNRGBA; their results
    carry host_decoded = True.  -> (results, files, summary)."""
    import ctypes as C

    import fennec_amd as fa
    L = fa.load_library()
    n = len(files)
    if n == 0:
        return [], [], BatchSummary()
    arrs = [np.frombuffer(f, dtype=np.uint8) for f in files]
    srcs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    sizes = (C.c_size_t * n)(*[len(f) for f in files])
    # one arena for the outputs, kept between calls: fresh pages cost a fault each when the file comes down.  A search
    # ends at a lower quality than a camera wrote, so the source's size (+ the header) is the usual bound; an item whose
    # file is larger comes back FNX_ERR_INVALID with its size and runs again below.
    capl = [len(f) + 4096 for f in files]
    offs = np.concatenate(([0], np.cumsum(capl)))
    global _out_arena
    if _out_arena is None or _out_arena.size < int(offs[-1]):
        _out_arena = np.empty(int(offs[-1]), dtype=np.uint8)
    bufs = [_out_arena[int(offs[i]):int(offs[i + 1])] for i in range(n)]
    outs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    caps = (C.c_size_t * n)(*capl)
    res = (fa.NativeBatchResult * n)()
    devs, ndev = _device_list(device, devices)
    rc = L.fennec_CompressBatchJPEGDevices(devs, ndev, int(workers), n, srcs, sizes, float(target_ssim), outs, caps, res, None, None, None)
    if rc != fa.FNX_OK:
        raise fa.FennecError(f"fennec_CompressBatchJPEG: {L.fnx_last_error().decode()}")
    for i in range(n):
        if res[i].failed and res[i].status == fa.FNX_ERR_INVALID and int(res[i].compressed_size) > capl[i]:
            big = np.empty(int(res[i].compressed_size), dtype=np.uint8)
            one = (fa.NativeBatchResult * 1)()
            L.fennec_CompressBatchJPEGDevices(devs, ndev, 1, 1, (C.c_void_p * 1)(arrs[i].ctypes.data), (C.c_size_t * 1)(len(files[i])),
                                              float(target_ssim), (C.c_void_p * 1)(big.ctypes.data), (C.c_size_t * 1)(big.size), one, None, None, None)
            one[0].index = i
            res[i] = one[0]
            bufs[i] = big
    results, out_files = [], []
    for i in range(n):
        r = res[i]
        br = BatchResult(Index=r.index, OriginalSize=int(r.original_size), CompressedSize=int(r.compressed_size), SSIM=float(r.ssim),
                         Quality=int(r.quality), Err=None if not r.failed else f"status {r.status}", has_result=bool(r.has_result))
        br.steps = int(r.steps)
        br.host_decoded = False
        br.device = int(r.device)
        results.append(br)
        out_files.append(bufs[i][:int(r.compressed_size)].tobytes() if not r.failed else b"")
    # the caller's side of FNX_ERR_UNSUPPORTED: host decode, then the NRGBA pool
    redo = [i for i in range(n) if res[i].failed and res[i].status in (fa.FNX_ERR_UNSUPPORTED, fa.FNX_ERR_INVALID)]
    if redo:
        decoded, ok = [], []
        for i in redo:                                 # batch.go:108-113: a file nobody can decode is THAT item's error, not the batch's
            try:
                decoded.append(decode(files[i]))
                ok.append(i)
            except Exception as e:
                results[i] = BatchResult(Index=i, OriginalSize=len(files[i]), Err=f"{type(e).__name__}: {e}", has_result=False)
                results[i].host_decoded = True
                out_files[i] = b""
        if ok:
            r2, f2, _ = compress_batch_native(decoded, target_ssim, workers=workers, device=device, devices=devices,
                                              original_sizes=[len(files[i]) for i in ok])
            for k, i in enumerate(ok):
                r2[k].Index = i
                r2[k].host_decoded = True
                results[i], out_files[i] = r2[k], f2[k]
    return results, out_files, summarize_local(results)
