#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

Workload (BASELINE.json configs[1], "config2"): 4K (3840x2160) NRGBA, separable
GaussianBlur sigma=2.0 followed by SSIMFast(original, blurred), inputs resident in HBM.
One STEP = one pass of that hot path over a batch of B distinct synthetic 4K images
(B x 33.2 MB of input, far larger than the 256 MiB Infinity Cache, so every image is read
cold from HBM).  Metric: source megapixels per second, whole job.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run (one rank per GPU): images are independent
(CompressBatch items never interact, batch.go:88-122), so each rank processes its own B
images -- weak scaling, no data-path collective; the only reduction is Summarize's
(count, ssim-sum) all-reduce over RCCL after the timed region is closed.

Two ways through the C ABI, bit-identical results (tests/test_gpu_parity.py::test_blur_ssimfast_one_pass_*):
--pipeline one-pass (default) calls fnx_gaussian_blur_ssim_fast_batch, whose blur kernel also
accumulates SSIMFast's boxDownsample sums, so neither full-size image is read a second time;
--pipeline two-call calls fnx_gaussian_blur_batch then fnx_ssim_fast_batch, as the reference does.
The default line carries both beside `value`: `two_call`, and `two_call_keep` -- the same two calls with
FNX_BLUR_KEEP_BOX_SUMS on the blur, whose kernel then leaves the box sums of both sides for the scoring call.

--depth 1 (default): one step at a time -- the blur kernel has the GPU to itself, which is what `roofline`
describes.  --depth 2: two steps in flight; step s+1 is enqueued -- on a second context (= HIP stream), into a
second set of destination images -- before step s's B scores are fetched, the way a CompressBatch worker pool
keeps the queue full (batch.go:84-123).  Every step still blurs and scores all B images and fetches its
results inside the timed region; two blur kernels then share the GPU (each launch takes twice as long, the
pair finishes 7 % sooner than back to back: their phases decorrelate and a step's small tail kernels run
underneath).  The default line reports that rate too (`pipelined`), measured after the timed region.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP events on the stream the
kernel runs on) and, at N == 1, `cpu_baseline` (the oracle -- a C restatement of the Go
reference with its threading model -- on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W4K, H4K = 3840, 2160
SIGMA = 2.0
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
PREWARM_S = 0.3            # untimed: clocks ramp under load before the W warm-up steps


def _dist_setup(torch, dist, local_rank, world):
    """One process per GPU over RCCL.  Test hook: FENNEC_BENCH_BACKEND=gloo with
    FENNEC_BENCH_SINGLE_DEVICE=1 runs the N > 1 control flow (barriers, MAX / SUM reductions, rank-0
    JSON) with every rank on GPU 0 -- RCCL cannot share a device -- so that the multi-rank path can be
    exercised on a 1-GPU box.  Returns (device index, device for reduction tensors)."""
    backend = os.environ.get("FENNEC_BENCH_BACKEND", "nccl")
    dev = 0 if os.environ.get("FENNEC_BENCH_SINGLE_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev)
    # FENNEC_BENCH_FORCE_DIST=1: bring the process group up at world size 1 too (one rank, RCCL initialised, every
    # reduction below really issued): what a 1-GPU box can prove about the N > 1 launch before an 8-GPU lease is spent
    if world > 1 or os.environ.get("FENNEC_BENCH_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    return dev, ("cuda" if backend == "nccl" else "cpu")


def dist_identity(torch, dist, dev: int, rank: int, world: int) -> dict:
    """Who took part: every rank's (rank, device ordinal, PCI bus id, uuid), all-gathered over the job's process group, so
    that the line of an N-GPU run shows N ranks on N DIFFERENT devices.  Over RCCL the ids must be
    distinct -- two ranks on one device is a launch error this run refuses to report a number for; the gloo test hook
    (FENNEC_BENCH_SINGLE_DEVICE=1) shares GPU 0 on purpose and says so."""
    props = torch.cuda.get_device_properties(dev)
    bus = "%04x:%02x:%02x" % (int(getattr(props, "pci_domain_id", 0)), int(getattr(props, "pci_bus_id", -1) & 0xff if getattr(props, "pci_bus_id", -1) >= 0 else 0),
                               int(getattr(props, "pci_device_id", 0)))
    me = {"rank": rank, "device": dev, "pci": bus, "uuid": str(getattr(props, "uuid", "")), "name": props.name, "pid": os.getpid()}
    seen = [None] * world
    if dist.is_initialized():
        dist.all_gather_object(seen, me)
    else:
        seen = [me]
    backend = dist.get_backend() if dist.is_initialized() else "none"
    shared = os.environ.get("FENNEC_BENCH_SINGLE_DEVICE") == "1"
    ids = [(s["pci"], s["uuid"]) for s in seen]
    distinct = len(set(ids)) == len(ids)
    if backend == "nccl" and not shared and not distinct:
        raise RuntimeError(f"bench.py: {world} ranks but only {len(set(ids))} distinct devices: {seen}")
    return {"backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend, "world": world,
            "rccl_ranks_seen": seen, "devices_distinct": distinct,
            "shared_device_test_hook": shared,
            "collectives": "barrier + MAX(elapsed) + SUM(Summarize partials): 3 small all-reduces per run, none on the data path"}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="4K images per step per GPU")
    ap.add_argument("--contexts", type=int, default=None,
                    help="worker contexts (one fnx ctx + HIP stream each) per GPU.  Default: 1 for config 2 (one stream, so "
                         "that `roofline` times the kernel alone) and config 4 (two 100+ us kernels per image already fill "
                         "the GPU); 4 for config 3, whose ~10 short kernels per image leave most of the GPU idle on one stream")
    ap.add_argument("--pipeline", default="one-pass", choices=["one-pass", "two-call"],
                    help="one-pass (default): fnx_gaussian_blur_ssim_fast_batch, the blur kernel also gathers "
                         "SSIMFast's boxDownsample sums, each image crosses HBM once; two-call: "
                         "fnx_gaussian_blur_batch then fnx_ssim_fast_batch (bit-identical results)")
    ap.add_argument("--depth", type=int, default=1,
                    help="one-pass pipeline: steps in flight.  1 (default): one step at a time, so the blur kernel runs "
                         "alone and `roofline` describes it; 2: step s+1 is enqueued (on a second context = stream, into "
                         "a second set of destinations) before step s's results are fetched (+7 %%; the default line "
                         "reports that rate as `pipelined`)")
    ap.add_argument("--config3-chunk", type=int, default=4,
                    help="config 3: images per batched call (fnx_lanczos_resize_batch + fennec_MSSSIM_batch_enqueue: one set of resize "
                         "launches per chunk); 1 = one call per image as in rounds 2-5")
    ap.add_argument("--ssim-mode", default="fast", choices=["fast", "exact"],
                    help="config 4's full-resolution SSIM: fast = FNX_SSIM_FAST (fp32 moments, |delta| <= 1e-6: SURVEY Appendix A's tolerance for "
                         "fp32-moment paths), exact = the default of the library (fp64 moments, <= 1e-9); the other one is reported beside it")
    ap.add_argument("--blur-mode", default="fast", choices=["fast", "exact"],
                    help="fast (default): 24-bit fixed-point weights on the i8 matrix pipe, exact integer sums: <= 1 LSB on "
                         "<= 0.001 %% of samples (north_star allows a stated tolerance; the bar is 0.1 %%); exact: the same sums "
                         "under a rounding guard + fp64 recomputation of the flagged samples, bit-identical to the reference's")
    ap.add_argument("--prewarm", type=float, default=None,
                    help="seconds of untimed steps before the warm-up steps (GPU clock ramp; setup, not measurement)")
    ap.add_argument("--workers", type=int, default=0, help="config5: host worker threads per GPU (0 = min(8, cores))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-queue-ahead", action="store_true",
                    help="one-pass, depth 1: fetch step s before enqueueing step s+1 (the stream drains for ~15 us per step)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the measurements made after the timed region (pipelined, exact_mode, pcie_inclusive): "
                         "profiler runs use it so that kernel statistics cover the timed configuration only")
    ap.add_argument("--threads", type=int, default=None,
                    help="configs 3/4: host threads driving the worker contexts (default: 2 for config3's four contexts -- python "
                         "threads contend for the interpreter lock; every call only enqueues, so one thread feeds several streams)")
    ap.add_argument("--queue", default="static", choices=["static", "dynamic"],
                    help="config5 at N > 1: 'static' gives rank r the items i = r mod N; 'dynamic' is ONE queue for the job "
                         "(a counter in torch.distributed's store, batch.go:72-126 across ranks)")
    ap.add_argument("--device-codec", action="store_true",
                    help="config5: search AND encode on the device (fnx_jpeg_compress); the host codec only decodes the source")
    ap.add_argument("--device-decode", action="store_true",
                    help="config5: no host codec at all -- the source is decoded on the device too (fnx_jpeg_recompress)")
    ap.add_argument("--device-search", action="store_true",
                    help="config5: the quality search round-trips every candidate on the device (fnx_jpeg_quality_search); the host "
                         "codec decodes the source and encodes the winner only")
    ap.add_argument("--no-batch", action="store_true", help="skip the `batch` object (CompressBatch images/s over --batch-items 4K JPEGs)")
    ap.add_argument("--batch-items", type=int, default=4096, help="items of the `batch` job (BASELINE config 5: 4096)")
    ap.add_argument("--batch-files", type=int, default=512,
                    help="distinct synthetic 4K JPEG files the items cycle through (256 distinct images x 2 qualities; 3.2 GB of host memory)")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config4", "config5", "analyze", "palette", "scale-search"],
                    help="BASELINE.json config to run; config2 (default) is the headline metric")
    args = ap.parse_args()
    if args.prewarm is None:                # configs 3 and 4: the governor needs > 1 s of these lighter loads (see other_configs)
        args.prewarm = 1.5 if args.workload in ("config3", "config4") else PREWARM_S
    if args.contexts is None:
        args.contexts = 4 if args.workload == "config3" else 1
    if args.threads is None:
        args.threads = 2 if args.workload == "config3" else args.contexts
    if args.workload != "config2":
        return other_workloads(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus N > 1 must be launched with torch.distributed.run", file=sys.stderr)
            return 2
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        return 2
    local_rank, red_dev = _dist_setup(torch, dist, local_rank, world)

    import fennec_amd
    from fennec_amd import synth

    ctx = fennec_amd.Context(local_rank)
    B = args.batch
    mp_per_image = W4K * H4K / 1e6
    S = 4 * W4K * H4K                       # bytes of one NRGBA 4K image

    # ---- synthetic inputs, resident in HBM before the timed region ------------------------
    srcs, dsts = [], []
    for img in synth.large_photo_batch(W4K, H4K, range(rank * B, rank * B + B)):    # image k = rank*B + i, all distinct
        srcs.append(torch.from_numpy(img).cuda())
    for i in range(B):
        dsts.append(torch.empty((H4K, W4K, 4), dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()

    # ---- worker contexts (= HIP streams).  Default 1: one stream, kernels back to back, so the
    # per-kernel HIP-event durations are the kernels' own (they feed `roofline`).  With
    # --contexts 2+ the batch is split and the contexts run in complementary phases (the blur is
    # VALU-bound, SSIMFast's box-downsample load-bound: while context 0 blurs its share the others
    # score the share they blurred just before -- CompressBatch's worker pool, batch.go:84-123,
    # on one GPU): +4 % here, +11 % from a C++ host (tools/kbench overlap); per-kernel durations
    # then include co-scheduling and no longer describe the kernel alone.  Either way every step
    # blurs and scores all B images and all K steps' work happens inside the timed region.
    one_pass = args.pipeline == "one-pass"
    nctx = max(1, min(args.contexts, B))
    ctxs = [ctx] + [fennec_amd.Context(local_rank) for _ in range(nctx - 1)]
    halves = [list(range(k, B, nctx)) for k in range(nctx)]
    exact = args.blur_mode == "exact"
    blur_plans = [c.plan_blur_batch([srcs[i] for i in hv], SIGMA, outs=[dsts[i] for i in hv], exact=exact)
                  for c, hv in zip(ctxs, halves)]
    ssim_plans = [c.plan_ssim_fast_batch([srcs[i] for i in hv], [dsts[i] for i in hv]) for c, hv in zip(ctxs, halves)]
    fused_plans = [c.plan_blur_ssim_fast_batch([srcs[i] for i in hv], SIGMA, outs=[dsts[i] for i in hv], exact=exact)
                   for c, hv in zip(ctxs, halves)] if one_pass else None
    kernel_ms = []
    fetch_t = []               # host time at which each timed step's results were in hand (queue-ahead loop)
    if one_pass:
        ctx.profile(True)      # the library brackets its blur_direct_kernel launches with HIP events
    ext = torch.cuda.ExternalStream(ctx.stream, device=torch.device("cuda", local_rank))
    # ---- depth-D software pipeline over whole steps (one-pass, one context per step in flight): every step
    # still blurs and scores all B images and its B results are fetched inside the timed region; what
    # overlaps is step s's tail (box_from_slabs, windowed SSIM, finish: small grids) with step s+1's blur
    depth = max(1, args.depth) if one_pass and nctx == 1 else 1
    pipe_ctx = [ctx] + [fennec_amd.Context(local_rank) for _ in range(depth - 1)]
    pipe_dsts = [dsts] + [[torch.empty_like(d) for d in dsts] for _ in range(depth - 1)]
    torch.cuda.synchronize()
    pipe_plans = ([fused_plans[0]] + [c.plan_blur_ssim_fast_batch(srcs, SIGMA, outs=o, exact=exact)
                                      for c, o in zip(pipe_ctx[1:], pipe_dsts[1:])]) if depth > 1 else None
    pipe_ext = [ext] + [torch.cuda.ExternalStream(c.stream, device=torch.device("cuda", local_rank)) for c in pipe_ctx[1:]]
    for c in pipe_ctx[1:]:
        c.profile(True)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    vals = np.zeros(B)

    def run_pipelined(nsteps, events=None):
        """nsteps full passes, `depth` of them in flight: enqueue step s, then fetch step s - depth + 1."""
        for s in range(nsteps + depth - 1):
            if s < nsteps:
                d = s % depth
                pipe_plans[d].enqueue()                        # fnx_gaussian_blur_ssim_fast_batch_enqueue
            if s >= depth - 1:
                d = (s - depth + 1) % depth
                vals[:] = pipe_plans[d].fetch()                # fnx_results_fetch: that step's B scores
                if events:
                    kernel_ms.append(pipe_ctx[d].kernel_ms())  # its blur kernel's HIP events (complete by now)

    def run_steps(nsteps, events=None):
        """nsteps full passes.  Context 0 blurs then scores; the others score (their blur was queued
        at the end of the previous step, or in the prologue) and then blur for the next step."""
        if depth > 1:
            return run_pipelined(nsteps, events)
        if one_pass and nctx == 1 and not args.no_queue_ahead:
            # one context: step s+1 is ENQUEUED before step s's scores are fetched (the ctx keeps a FIFO of unfetched
            # batches), so the GPU does not drain while the host turns around; the blurs run in order on the ctx's
            # stream, step s's tail (box finish, window sums) on its second stream, i.e. under step s+1's blur
            for s in range(nsteps + 1):
                if s < nsteps:
                    fused_plans[0].enqueue()                   # fnx_gaussian_blur_ssim_fast_batch_enqueue
                if s >= 1:
                    vals[:] = fused_plans[0].fetch()           # fnx_results_fetch: the oldest unfetched batch
                    if events:
                        kernel_ms.append(ctx.kernel_ms())      # that step's blur kernel (oldest unread event pair)
                        fetch_t.append(time.perf_counter())
            return
        if one_pass:
            for s in range(nsteps):
                for k in range(nctx):
                    fused_plans[k].enqueue()                   # fnx_gaussian_blur_ssim_fast_batch_enqueue
                for k in range(nctx):
                    vals[halves[k]] = fused_plans[k].fetch()   # fnx_results_fetch: the step's only host wait
                if events:
                    kernel_ms.append(ctx.kernel_ms())          # the library's HIP events around the blur kernel: complete by now
            return
        for k in range(1, nctx):
            blur_plans[k].run()                                # prologue: belongs to the first step
        for s in range(nsteps):
            if events:
                events[s][0].record(ext)
            blur_plans[0].run()                                # fnx_gaussian_blur_batch: one launch
            if events:
                events[s][1].record(ext)
            for k in range(1, nctx):
                ssim_plans[k].enqueue()                        # fnx_ssim_fast_batch_enqueue
            ssim_plans[0].enqueue()
            if events:
                events[s][2].record(ext)
            if s + 1 < nsteps:
                for k in range(1, nctx):
                    blur_plans[k].run()                        # next step's blur of the other halves
            for k in range(nctx):
                vals[halves[k]] = ssim_plans[k].fetch()        # fnx_results_fetch: the step's only syncs

    def barrier():
        if world > 1 or dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # untimed setup: the GPU needs ~50 ms of load to reach its steady clocks (measured: 275 k MP/s
    # with 3 warm-up steps, 301 k with 50+); run the pipeline for PREWARM_S before the W warm-up steps
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm:
        run_steps(4)
    run_steps(args.warmup)
    if one_pass:
        for c in pipe_ctx:
            c.profile(True)          # forget the warm-up launches: kernel_ms() then reads the timed steps' events
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, ev)
    barrier()
    elapsed = time.perf_counter() - t0

    if world > 1 or dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # Summarize (batch.go:140-158) across ranks: the path's only reduction
        red = torch.tensor([float(len(vals)), float(np.sum(vals))], dtype=torch.float64, device=red_dev)
        dist.all_reduce(red, op=dist.ReduceOp.SUM)
        n_items, ssim_sum = int(red[0].item()), float(red[1].item())
    else:
        n_items, ssim_sum = len(vals), float(np.sum(vals))

    total_mp = mp_per_image * B * world * args.steps
    value = total_mp / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel (blur_direct_kernel), HIP events on the ctx stream ----
    nb0 = len(halves[0])
    path_gbs = 4.0 * S * B * world * args.steps / elapsed / 1e9
    if one_pass:
        # events recorded by the library around the kernel itself (fnx_ctx_profile); the launch
        # covers ALL of config 2's full-size traffic: SURVEY 8(d) counts 4*S per image (blur
        # reads S + writes S, SSIMFast reads 2*S); the one-pass kernel moves 2*S of it (each
        # source pixel read once, each blurred pixel written once) and never re-reads either
        # (the only events in the one-pass timed loop are the library's pair around the blur kernel: every
        # event record is a barrier packet on the stream, and five more per step cost 2 % of the step)
        blur_ms = float(np.mean(kernel_ms))
        rest_ms = ms_per_step - blur_ms
        blur_bytes = 4.0 * S * nb0
        blur_gbs = blur_bytes / (blur_ms * 1e-3) / 1e9
        route = pipe_ctx[0].last_kernel(fennec_amd.PROF_MAIN)       # what the library's dispatch launched, not what a switch suggests
        mfma = route.startswith("blur_mfma_kernel")
        kname = "blur_mfma_kernel" if mfma else "blur_direct_kernel"
        traffic = committed_traffic(kname, nb0, scored=True, exact=exact)
        traffic_file = source_file("traffic")
        roofline = {
            "kernel": (f"{route} (GaussianBlur sigma=2 on the i8 matrix pipe + both boxDownsample sums of SSIMFast, "
                       f"one launch of {nb0} images)") if mfma else
                      (f"{route} (GaussianBlur sigma=2, fp32 FMA{' under a rounding guard + fp64 fix-ups' if exact else ''}, + both boxDownsample "
                       f"sums of SSIMFast, one launch of {nb0} images)"),
            "kernel_route": route,
            # achieved / peak / frac are SURVEY 8(d)'s HBM figures (the contract of this object).  The matrix-pipe kernel moves
            # 2.2 S per image at ~3.5 TB/s; the tiled-copy floor of its access shape (64-px column strips, read S + write S)
            # is ~13 us per 4K image on this part (experiments/mfma/pattern2.hip), the kernel takes ~19.5
            # `achieved` / `peak` / `frac` price SURVEY 8(d)'s algorithmic bytes against the HBM peak (the contract of this object).
            # What the counters say BOUNDS the kernel is instruction issue: VALU + matrix + LDS instructions fill the SIMDs while
            # the measured HBM rate (hbm_frac_measured: counter bytes / launch time / peak) is well under the peak.
            "bound": "valu",
            "bound_detail": ("instruction issue: VALU + i8 MFMA + LDS instructions fill the SIMDs (committed SQ counters) while the measured HBM "
                             "rate is hbm_frac_measured of the peak" if mfma else "VALU issue (fp32 FMA form)") + "; frac is priced against hbm",
            "roofline_priced_against": "hbm",
            "achieved": round(blur_gbs, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(blur_gbs / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_file": traffic_file,
            "hbm_frac_measured": (round(traffic / (blur_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None),
            "traffic_source": "the newest profiles/*_traffic.json holding this kernel instantiation (named in traffic_file): rocprofv3 --pmc FETCH_SIZE x 2 + "
                              "WRITE_SIZE passes of this command (committed; not measured in this run -- the driver's run has no profiler attached)",
            "algorithmic_bytes_per_launch": blur_bytes,
            "algorithmic_bytes_note": "SURVEY 8(d): 4*S per image for GaussianBlur+SSIMFast; the one-pass kernel "
                                      "needs only 2*S of HBM traffic for it (see traffic)",
            "achieved_on_2S": round(blur_gbs / 2, 1),
            "avg_launch_ms": round(blur_ms, 4),
            "launch_ms_min_median_max": [round(float(np.min(kernel_ms)), 4), round(float(np.median(kernel_ms)), 4),
                                         round(float(np.max(kernel_ms)), 4)],
            "valu_issue_frac": committed_valu_issue(kname, blur_ms, scored=True, exact=exact),
            "valu_issue_file": source_file("valu_issue"),
            "valu_issue_source": "SQ_ACTIVE_INST_VALU and the shader clock from the newest profiles/*_sq_counters.txt holding this kernel (named in valu_issue_file; committed PMC pass) over THIS run's launch time",
            "avg_launch_how": "HIP events bound to the dispatch itself (hipExtLaunchKernelGGL start / stop events: the kernel packet's own "
                              "timestamps, no barrier packets on the stream), every launch of the timed region",
            "note": "the previous step's tail (box_from_slabs, windowed SSIM with its finish) runs on the ctx's second stream "
                    "under this launch, beside its waves (three 154-register waves per SIMD leave room for them), so its duration "
                    "includes their share of the GPU",
        }
        if depth > 1:
            roofline["note"] = (f"measured live in the timed region, where the kernel shares the GPU with the previous "
                                f"step's tail kernels (pipeline depth {depth}); `serial` has the kernel running alone")
        rest = {
            "kernels": "box_from_slabs_kernel + windowed_ssim_march_kernel<true> (the last workgroup of an image takes its mean; results land "
                       "in pinned host memory), both under the NEXT step's blur",
            "avg_ms": round(rest_ms, 4),
            "how": "step period minus the blur kernel's duration = the dispatch gap between consecutive blur launches (the tail's own "
                   "kernels, ~45 + 40 us alone, run concurrently: the newest profiles/*_onepass_kernel_stats.csv)",
        }
        if depth > 1:
            rest["note"] = "step period minus the (co-scheduled) blur kernel's duration: not a kernel time at this depth"
    else:
        blur_ms = float(np.mean([ev[s][0].elapsed_time(ev[s][1]) for s in range(args.steps)]))
        ssim_ms = float(np.mean([ev[s][1].elapsed_time(ev[s][2]) for s in range(args.steps)]))
        blur_bytes = 2.0 * S * nb0               # context 0's launch: read each source px once + write each dst px once
        blur_gbs = blur_bytes / (blur_ms * 1e-3) / 1e9
        ssim_bytes = 2.0 * S * nb0               # SSIMFast reads both full-size images once
        ssim_gbs = ssim_bytes / (ssim_ms * 1e-3) / 1e9
        route = ctxs[0].last_kernel(fennec_amd.PROF_MAIN)
        mfma = route.startswith("blur_mfma_kernel")
        kname = "blur_mfma_kernel" if mfma else "blur_direct_kernel"
        roofline = {
            "kernel": f"{route} (GaussianBlur sigma=2" + (" on the i8 matrix pipe" if mfma else "") + f", one launch of {nb0} images"
                      + (", co-scheduled with the other context's SSIMFast kernels)" if nctx > 1 else ")"),
            "kernel_route": route,
            "bound": "hbm",
            "achieved": round(blur_gbs, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(blur_gbs / HBM_PEAK_GBS, 4),
            "traffic": committed_traffic(kname, nb0, exact=exact),
            "traffic_file": source_file("traffic"),
            "algorithmic_bytes_per_launch": blur_bytes,
            "avg_launch_ms": round(blur_ms, 4),
        }
        rest = {
            "kernels": "box_tiled_kernel x2 + windowed_ssim_kernel + ssim_finish_kernel (+ D2H of results)",
            "bound": "hbm",
            "achieved": round(ssim_gbs, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(ssim_gbs / HBM_PEAK_GBS, 4),
            "avg_ms": round(ssim_ms, 4),
        }

    out = {
        "metric": "megapixels/sec: 4K SSIMFast+GaussianBlur",
        "value": round(value, 1),
        "unit": "MP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8 (blur: exact int32 sums of 24-bit fixed-point weights on the i8 matrix pipe" + (", flagged samples in fp64" if exact else "") + "; fp64 SSIM)",
        "data": "synthetic",
        "config": {
            "workload": "config2: 4K (3840x2160) NRGBA GaussianBlur sigma=2.0 + SSIMFast(orig, blurred)",
            "images_per_step_per_gpu": B,
            "width": W4K, "height": H4K, "sigma": SIGMA,
            "blur_mode": "exact (integer sums under a rounding guard + fp64 recomputation of the flagged 0.01 %: bit-identical to the reference)" if exact else
                         "fast (24-bit fixed-point weights, exact integer sums: <=1 LSB on <=0.001% samples)",
            "inputs": "device-resident (HBM), batched C-ABI entry points",
            "pipeline": args.pipeline + (" (fnx_gaussian_blur_ssim_fast_batch)" if one_pass else
                                         " (fnx_gaussian_blur_batch, fnx_ssim_fast_batch)"),
            "contexts_per_gpu": nctx if depth == 1 else depth,
            "pipeline_depth": depth,
            "queue_ahead": bool(one_pass and nctx == 1 and depth == 1 and not args.no_queue_ahead),
            "prewarm": f"{args.prewarm} s of untimed steps before the {args.warmup} warm-up steps (GPU clock ramp)",
            "parallelism": f"independent images sharded over {world} GPU(s)",
        },
        "roofline": roofline,
        "roofline_ssimfast": rest,
        "path_hbm_frac": round(path_gbs / world / HBM_PEAK_GBS, 4),
        "step_ms": ({"min": round(float(np.min(np.diff(fetch_t))) * 1e3, 4), "median": round(float(np.median(np.diff(fetch_t))) * 1e3, 4),
                     "max": round(float(np.max(np.diff(fetch_t))) * 1e3, 4), "how": "intervals between consecutive steps' results arriving"}
                    if len(fetch_t) > 2 else None),
        "summarize": {"items": n_items, "avg_ssim": ssim_sum / max(n_items, 1)},
    }

    # kernel-only rate (HIP-event kernel time of the step) and the PCIe-inclusive
    # rate of the same two ops called with HOST buffers (what the cgo shim's FNX_HOST calls see) -- the
    # latter measured after the timed region on one image, never part of `value`
    step_kernel_ms = blur_ms + (rest_ms if one_pass else ssim_ms)
    if depth == 1:
        out["kernel_only"] = {"value": round(mp_per_image * nb0 / (step_kernel_ms * 1e-3), 1), "unit": "MP/s",
                              "ms_per_step": round(step_kernel_ms, 4)}
        if one_pass and nctx == 1 and not args.no_queue_ahead:
            out["kernel_only"]["note"] = "with the next step queued ahead the stream never drains: GPU time per step = wall time per step"
        if rank == 0 and world == 1 and one_pass and nctx == 1 and not args.no_extras:
            # the same steps two in flight (--depth 2), after the timed region; never `value`
            ctx.profile(False)
            c2 = fennec_amd.Context(local_rank)
            d2 = [torch.empty_like(d) for d in dsts]
            torch.cuda.synchronize()
            pl = [fused_plans[0], c2.plan_blur_ssim_fast_batch(srcs, SIGMA, outs=d2, exact=exact)]

            def two_in_flight(n):
                for s_ in range(n + 1):
                    if s_ < n:
                        pl[s_ & 1].enqueue()
                    if s_ >= 1:
                        pl[(s_ - 1) & 1].fetch()
            t_p = time.perf_counter()
            while time.perf_counter() - t_p < 0.1:
                two_in_flight(4)
            t_p = time.perf_counter()
            two_in_flight(20)
            t_p = (time.perf_counter() - t_p) / 20
            out["pipelined"] = {"value": round(mp_per_image * B / t_p, 1), "unit": "MP/s", "ms_per_step": round(t_p * 1e3, 4),
                                "depth": 2, "note": "--depth 2: step s+1 enqueued on a second context before step s is "
                                                    "fetched; 20 steps after the timed region"}
            ctx.profile(True)
            del pl, d2
            c2.close()
    elif rank == 0 and world == 1 and not args.no_extras:
        # the same step one at a time (depth 1), after the timed region: the blur kernel alone on the GPU
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_s = time.perf_counter()
        while time.perf_counter() - t_s < 0.1:
            pipe_plans[0].enqueue()
            pipe_plans[0].fetch()
        ks, ls = [], []
        t_s = time.perf_counter()
        for _ in range(10):
            e0.record(ext)
            pipe_plans[0].enqueue()
            e1.record(ext)
            pipe_plans[0].fetch()
            ks.append(ctx.kernel_ms())
            e1.synchronize()
            ls.append(e0.elapsed_time(e1))
        t_s = (time.perf_counter() - t_s) / 10
        k_ms, l_ms = float(np.mean(ks)), float(np.mean(ls))
        out["serial"] = {"value": round(mp_per_image * B / t_s, 1), "unit": "MP/s", "ms_per_step": round(t_s * 1e3, 4),
                         "blur_kernel_ms": round(k_ms, 4),
                         "roofline_frac": round(4.0 * S * B / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "note": "--depth 1: one step at a time, 10 steps after the timed region"}
        out["kernel_only"] = {"value": round(mp_per_image * B / (l_ms * 1e-3), 1), "unit": "MP/s", "ms_per_step": round(l_ms, 4),
                              "note": "HIP-event time of one serial step's kernels"}
    if rank == 0 and world == 1 and one_pass and not exact and nctx == 1 and not args.no_extras:
        # the same step with bit-exact blurred images (FNX_BLUR_EXACT), after the timed region; never `value`
        ctx.profile(False)
        xplan = ctx.plan_blur_ssim_fast_batch(srcs, SIGMA, outs=dsts, exact=True)

        def exact_steps(n):                               # the default loop's protocol: step s + 1 queued before step s is fetched
            for s_ in range(n + 1):
                if s_ < n:
                    xplan.enqueue()
                if s_ >= 1:
                    xplan.fetch()
        t_x = time.perf_counter()
        while time.perf_counter() - t_x < 0.1:
            exact_steps(4)
        exact_steps(args.warmup)
        torch.cuda.synchronize()
        t_x = time.perf_counter()
        exact_steps(args.steps)
        torch.cuda.synchronize()
        t_x = (time.perf_counter() - t_x) / args.steps
        out["exact_mode"] = {"value": round(mp_per_image * B / t_x, 1), "unit": "MP/s", "ms_per_step": round(t_x * 1e3, 4),
                             "steps": args.steps, "warmup": args.warmup,
                             "note": "the same K steps with FNX_BLUR_EXACT (what the Go shim's GaussianBlur passes): blurred images "
                                     "bit-identical to the reference's, scores from exactly those images; same protocol as the "
                                     "timed region (warm-up, synchronize on both sides), run right after it; "
                                     "`python bench.py --blur-mode exact` makes it the headline line"}
    if rank == 0 and world == 1 and one_pass and nctx == 1 and depth == 1 and not args.no_extras:
        # (a) the timed loop again over 500 steps: the driver's 20 steps last 14 ms, inside the clock governor's settling time
        ctx.profile(False)
        torch.cuda.synchronize()
        t_l = time.perf_counter()
        run_steps(500)
        torch.cuda.synchronize()
        t_l = (time.perf_counter() - t_l) / 500
        out["long_run"] = {"value": round(mp_per_image * B / t_l, 1), "unit": "MP/s", "ms_per_step": round(t_l * 1e3, 4), "steps": 500,
                           "note": "the timed region's loop over 500 steps, right after it (steady clocks); never `value`"}
        # (b) the reference's own shape: GaussianBlur(img) -> img (effects.go:146), then SSIMFast(a, b) (ssim.go:48), as two batched calls
        def two_call_steps(n):
            for _ in range(n):
                blur_plans[0].run()                            # fnx_gaussian_blur_batch
                ssim_plans[0].enqueue()                        # fnx_ssim_fast_batch_enqueue
                ssim_plans[0].fetch()
        t_c = time.perf_counter()
        while time.perf_counter() - t_c < 0.1:
            two_call_steps(2)
        two_call_steps(args.warmup)
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        two_call_steps(args.steps)
        torch.cuda.synchronize()
        t_c = (time.perf_counter() - t_c) / args.steps
        out["two_call"] = {"value": round(mp_per_image * B / t_c, 1), "unit": "MP/s", "ms_per_step": round(t_c * 1e3, 4),
                           "steps": args.steps, "warmup": args.warmup, "roofline_frac": round(4.0 * S * B / t_c / 1e9 / HBM_PEAK_GBS, 4),
                           "note": "fnx_gaussian_blur_batch then fnx_ssim_fast_batch on the same 32 images: the reference's two calls, "
                                   "4*S of HBM traffic per image; same protocol as the timed region, run after it"}
        # (c) the same two calls with the hand-over flag: the blur call promises that the scoring call over the same pairs comes next
        keep_plan = ctx.plan_blur_batch(srcs, SIGMA, outs=dsts, exact=exact, keep_box_sums=True)
        plain_vals = np.array(ssim_plans[0].run())

        def two_call_keep_steps(n):
            for _ in range(n):
                keep_plan.run()                                # fnx_gaussian_blur_batch(FNX_BLUR_KEEP_BOX_SUMS)
                ssim_plans[0].enqueue()                        # fnx_ssim_fast_batch_enqueue: the kept box sums, no image read
                vals[:] = ssim_plans[0].fetch()
        t_k = time.perf_counter()
        while time.perf_counter() - t_k < 0.1:
            two_call_keep_steps(2)
        two_call_keep_steps(args.warmup)
        torch.cuda.synchronize()
        t_k = time.perf_counter()
        two_call_keep_steps(args.steps)
        torch.cuda.synchronize()
        t_k = (time.perf_counter() - t_k) / args.steps
        out["two_call_keep"] = {"value": round(mp_per_image * B / t_k, 1), "unit": "MP/s", "ms_per_step": round(t_k * 1e3, 4),
                                "steps": args.steps, "warmup": args.warmup, "roofline_frac": round(4.0 * S * B / t_k / 1e9 / HBM_PEAK_GBS, 4),
                                "max_abs_diff_vs_plain_two_call": float(np.max(np.abs(vals - plain_vals))),
                                "note": "the same two calls with FNX_BLUR_KEEP_BOX_SUMS on the blur: its kernel takes SSIMFast's box sums of "
                                        "both sides as it passes, fnx_ssim_fast_batch reads neither image again (2*S of traffic, priced "
                                        "against the path's 4*S like `value`); the caller promises the scoring call comes next"}
        ctx.profile(True)
    host0 = srcs[0].cpu().numpy() if rank == 0 else None
    if rank == 0 and not args.no_extras:
        host = host0
        S_mb = host.nbytes / 1e6
        hb = np.empty_like(host)
        ctx.GaussianBlurSSIMFast(host, SIGMA, out=hb)
        reps = 5
        t_h = time.perf_counter()
        for _ in range(reps):
            _, hs = ctx.GaussianBlurSSIMFast(host, SIGMA, out=hb)       # fnx_gaussian_blur_ssim_fast, exact blur: what the shim's GaussianBlurScored calls
        t_h = (time.perf_counter() - t_h) / reps
        t_a = time.perf_counter()
        for _ in range(3):
            ctx.GaussianBlurSSIMFast(host, SIGMA)                # ... into a fresh result image each call (its first-touch page faults included)
        t_a = (time.perf_counter() - t_a) / 3
        ctx.GaussianBlur(host, SIGMA, exact=None)
        t_2 = time.perf_counter()
        for _ in range(3):
            hb2 = ctx.GaussianBlur(host, SIGMA, exact=None)      # the drop-in mirror: exact mode for host images
            hs2 = ctx.SSIMFast(host, hb2)
        t_2 = (time.perf_counter() - t_2) / 3
        # four worker contexts over host images (CompressBatch's shape, batch.go:84-123: one ctx per worker): the link is the
        # shared resource, so this is what a drop-in batch sees per GPU
        import threading
        wn, per = 4, 6
        wctx = [fennec_amd.Context(local_rank) for _ in range(wn)]
        hosts = [host0.copy() for _ in range(wn)]
        houts = [np.empty_like(host0) for _ in range(wn)]
        for c, hh, ho in zip(wctx, hosts, houts):
            c.GaussianBlurSSIMFast(hh, SIGMA, out=ho)

        def wloop(k):
            torch.cuda.set_device(local_rank)
            for _ in range(per):
                wctx[k].GaussianBlurSSIMFast(hosts[k], SIGMA, out=houts[k])
        ths = [threading.Thread(target=wloop, args=(k,)) for k in range(wn)]
        t_w = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        t_w = (time.perf_counter() - t_w) / (wn * per)
        out["pcie_inclusive"] = {
            "value": round(mp_per_image / t_h, 1), "unit": "MP/s", "ms_per_image": round(t_h * 1e3, 3),
            "bytes_moved_per_image": {"up": int(host.nbytes), "down": int(host.nbytes)},
            "link_GBps": round(2 * host.nbytes / t_h / 1e9, 1),
            "ms_per_image_fresh_result": round(t_a * 1e3, 3),
            "identical_to_two_calls": bool(np.array_equal(hb, hb2) and hs == hs2),
            "note": f"one context, pageable host buffers, fnx_gaussian_blur_ssim_fast (GaussianBlur exact + SSIMFast in one call): the source up, "
                    f"the blurred image down = 2 x {S_mb:.1f} MB; this box moves ~55 GB/s one way and no more both ways at once "
                    "(experiments/pcie/bw.py), so 1.2 ms per image is the link's floor; the result image is reused from call to call "
                    "(`ms_per_image_fresh_result`: a new numpy array per call, whose pages are first touched by the copy)",
            "two_calls": {"value": round(mp_per_image / t_2, 1), "unit": "MP/s", "ms_per_image": round(t_2 * 1e3, 3),
                          "bytes_moved_per_image": {"up": int(3 * host.nbytes), "down": int(host.nbytes)},
                          "note": "the reference-shaped pair fennec_GaussianBlur then fennec_SSIMFast on host images: the source goes up twice, "
                                  "the blurred image down and up again (what r5 reported as pcie_inclusive)"},
            "four_workers": {"value": round(mp_per_image / t_w, 1), "unit": "MP/s", "ms_per_image": round(t_w * 1e3, 3),
                             "note": "the one-call form from four threads, one ctx each (images per second over all of them)"}}
        del wctx
    if world > 1 or dist.is_initialized():
        di = dist_identity(torch, dist, local_rank, rank, world)       # a collective: every rank calls it
        di["queue"] = "config 2 (`value`): images sharded statically, no queue; `batch`: one dynamic queue for the job (store counter, chunks of 4 indices)"
        out["dist"] = di
    if not args.no_batch:
        # BASELINE.json's second metric: CompressBatch images/s, at every N (all ranks take part: ONE queue for the job)
        ctx.profile(False)
        bm = batch_metric(args, rank, world, local_rank, red_dev, ctx)
        if rank == 0:
            out["batch"] = bm
    if rank == 0 and world == 1 and not args.no_extras:
        def heavy_burst(seconds=0.5):
            # configs 3 and 4 are light loads; from a cold box their own 1.5 s of pre-warm leaves the clocks where a minute of
            # them would not (first process on a fresh box: 67 k MP/s for config 3, every later one 79-81 k).  Half a second
            # of the config-2 loop in front of each brings the governor to the state their steady state runs in.
            ctx.profile(False)
            t_b = time.perf_counter()
            while time.perf_counter() - t_b < seconds:
                run_steps(8)
            torch.cuda.synchronize()
        out["other_configs"] = other_configs(args, heavy_burst)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(host0)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


def other_configs(args, heavy_burst=None) -> dict:
    """BASELINE configs 3, 4 and 5 in front of the driver: a bounded pass of each after the default line's timed region (same
    protocol as their own `--workload` lines, smaller batches / fewer steps; about 15 s in all).  Never part of `value`."""
    import copy
    out = {}
    # (config 3's four worker streams jitter: 4 steps measured 59-83 k MP/s run to run, 40 steps 77-82 k)
    plan = [("config3", dict(batch=32, steps=40, warmup=3, contexts=4, threads=2)),
            ("config4", dict(batch=8, steps=16, warmup=2, contexts=1, threads=1)),     # (8 images per step as in `--workload config4`: with 4 the
                                                                                        # three-deep result queue drains at every step boundary, 158 k against 165 k)
            ("config5", dict(batch=16, steps=3, warmup=1, contexts=1, threads=1, device_decode=True))]
    for wl, over in plan:
        a = copy.copy(args)
        # pre-warm: configs 3 and 4 are light loads (many small kernels on a few streams) and the clock governor takes more
        # than a second of THEM to settle -- measured in this position: 0.15 s -> 69.7 k MP/s, 0.6 s -> 67.4 k, 1.5 s -> 79.1 k
        # for config 3, whatever ran before (r4; `--workload config3` alone: 67.7 k cold, 81.0 k right after a heavy run)
        a.workload, a.prewarm, a.no_cpu_baseline, a.no_extras = wl, (0.15 if wl == "config5" else 1.5), True, True
        a.device_codec = a.device_search = False
        a.device_decode = False
        for k, v in over.items():
            setattr(a, k, v)
        t0 = time.perf_counter()
        if heavy_burst is not None and wl in ("config3", "config4"):
            heavy_burst()
        import gc
        gc.collect()                            # the batch before this left ~10^5 objects behind: a full collection inside a 3 ms
        gc.disable()                            # step of config 3 (two host threads feeding four streams) is a 5-9 ms stall
        try:
            line = other_workload_line(a, embedded=True)
            keep = {k: line[k] for k in ("metric", "value", "value_before_prewarm", "unit", "steps", "warmup", "ms_per_step", "dtype", "roofline",
                                        "roofline_step", "step_ms", "gpu_stage", "result_sample", "ssim_mode", "other_ssim_mode") if k in line}
            keep["workload"] = line["config"]["workload"]
            keep["images_per_step"] = line["config"]["images_per_step_per_gpu"]
            keep["wall_s"] = round(time.perf_counter() - t0, 2)
            out[wl] = keep
        except Exception as e:                  # a failure here must not cost the headline line
            out[wl] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            gc.enable()
    out["note"] = ("bounded passes after the timed region, the protocol of `bench.py --workload configN` at smaller batches (configs 3 and 4: "
                   "0.5 s of the config-2 loop, then their own 1.5 s pre-warm, garbage collector paused: clock ramp and host jitter, not work); "
                   "config5 here is ONE host thread over 16 files (a per-item latency figure with its kernels' roofline); BASELINE config 5 "
                   "-- 4096 items, the worker pool -- is the `batch` object of this line")
    return out


def batch_metric(args, rank, world, local_rank, red_dev, ctx0) -> dict:
    """BASELINE.json's batch metric: CompressBatch (batch.go:58-128) over N_ITEMS synthetic 4K JPEGs, SSIM-guided quality
    search (Balanced, compress.go:21-87), sharded over the job's GPUs with ONE dynamic queue (a counter in
    torch.distributed's store: whichever worker of whichever rank asks first takes the next indices), no data-path
    collective; Summarize (batch.go:140-158) through the all-reduce (RCCL at N > 1).  Per item the FILE's bytes go up,
    decoder + search + encoder run on the device (fnx_jpeg_recompress), the new file comes down.  The files: DISTINCT
    synthetic 4K images (SURVEY 8(d)'s large_photo pattern, salted per image) encoded by the device encoder at q = 92 and 85
    before the timed region; item i reads file i mod len(files) (4096 distinct 7 MB files would be 29 GB of host memory
    per rank)."""
    import torch
    import torch.distributed as dist
    import fennec_amd
    from fennec_amd import batch as fbatch
    from fennec_amd import synth

    # one process per GPU, its worker pool on the GPU's own NUMA node for the length of the job (restored below: the CPU baseline
    # wants every core)
    affinity_before = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    numa_bound = fbatch.bind_to_device_numa(local_rank)
    n_items = args.batch_items
    n_files = max(1, min(args.batch_files, n_items))
    # 16 host threads per rank.  Round 2's kernels: 8 -> 12 -> 16 -> 24 threads measured 1 526 -> 1 944 -> 1 966 -> 1 548 images/s on
    # one GPU (the GPU runs 2-4 files' kernels side by side; past that the threads only contend for the interpreter lock); with
    # round 3's shorter GPU side of an item (no decoded / candidate images, the packed sync passes) 12 -> 16 -> 20 threads measure
    # 1 990-2 100 -> 2 410-2 440 -> 2 340 on one box
    workers = args.workers or max(1, min(16, host_cpus()[0] // max(world, 1)))
    target = fbatch.TARGET_SSIM["Balanced"]
    # the same file list on every rank (any rank may take any item)
    files = []
    nimg = (n_files + 1) // 2
    base = torch.from_numpy(synth.large_photo(W4K, H4K, 0)).cuda()
    for k in range(nimg):
        # large_photo(w, h, k) = large_photo(w, h, 0) + a per-channel salt mod 256 (synth.large_photo_batch): uint8 wrap-around on the device
        salt = torch.tensor([(17 * k) % 256, (31 * k) % 256, (5 * k) % 256, 0], dtype=torch.uint8, device=base.device)
        d = base + salt
        for q in (92, 85):
            if len(files) < n_files:
                files.append(ctx0.jpeg_encode(d, q))
        del d
    del base
    states = {}

    def make_state(wid):
        if wid not in states:
            states[wid] = fennec_amd.Context(local_rank)
        return states[wid]

    work_full = fbatch.jpeg_item_work_device_all([files[i % n_files] for i in range(n_items)], target)

    def work(idx, state):
        # the new file has come down (its size is in the result); keeping 4096 x 2 MB of them alive made whichever run came
        # second fault in 8.5 GB of fresh pages -- a 5 % swing between the job and its own N = 1 reference
        r = work_full(idx, state)
        if hasattr(r, "data"):
            r.data = None
        return r

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: worker contexts, scratch growth, clocks
    fbatch.compress_batch(min(n_items, 8 * workers), work, make_state, workers=workers)
    # N = 1 reference inside this job: rank 0 alone over a slice of the items, the other ranks idle
    # the WHOLE job (r4): a pool run carries ~40 ms of start and drain, so 384 / 1024-item references read 8-11 % under the
    # 4096-item job's own rate at N = 1; the same item count on both sides makes efficiency_vs_n1 read 1.00 there
    n_ref = n_items

    def reference_rate():
        t_r = time.perf_counter()
        fbatch.compress_batch(n_ref, work, make_state, workers=workers)
        return n_ref / (time.perf_counter() - t_r)
    barrier()
    ref_before = reference_rate() if rank == 0 else None
    barrier()
    t0 = time.perf_counter()
    res = fbatch.compress_batch(n_items, work, make_state, workers=workers, rank=rank, world=world,
                                queue_mode="dynamic" if world > 1 else "static", chunk=4, batch_id="bench-batch")
    barrier()
    elapsed = time.perf_counter() - t0
    # the reference once more AFTER the job (r4): the run before it still pays for the pool's first full-speed seconds (it read
    # 8-11 % under the job's own rate at N = 1); the reference is the better of the two
    ref_after = reference_rate() if rank == 0 else None
    barrier()
    ref_rate = max(ref_before, ref_after) if rank == 0 else None
    mine = len(res)
    up = sum(r.OriginalSize for r in res)
    down = sum(r.CompressedSize for r in res)
    host_dec = sum(1 for r in res if getattr(r, "host_decoded", False))
    summ = fbatch.summarize_distributed(res, device=red_dev if (world > 1 or dist.is_initialized()) else None, force=dist.is_initialized())
    per_rank = [mine]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.zeros(world, dtype=torch.float64, device=red_dev)
        cnt[rank] = mine
        io = torch.tensor([float(up), float(down), float(host_dec)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(io, op=dist.ReduceOp.SUM)
        per_rank = [int(v) for v in cnt.tolist()]
        up, down, host_dec = (int(v) for v in io.tolist())
        r0 = torch.tensor([ref_rate or 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(r0, op=dist.ReduceOp.MAX)
        ref_rate = float(r0.item())
    for c in states.values():
        c.close()
    if numa_bound and affinity_before is not None:
        os.sched_setaffinity(0, affinity_before)
    total = n_items / elapsed
    return {
        "metric": "images/sec: CompressBatch 4K JPEG, SSIM-guided quality search",
        "is": "BASELINE config 5 (the whole job; `other_configs.config5` is a one-thread, 16-file pass kept for its kernels' figures)",
        "value": round(total, 1), "unit": "images/s", "n_gpus": world, "items": n_items, "seconds": round(elapsed, 3),
        "images_per_s_per_rank": [round(c / elapsed, 1) for c in per_rank], "items_per_rank": per_rank,
        "n1_reference_images_per_s": round(ref_rate, 1),
        "efficiency_vs_n1": round(total / (world * ref_rate), 4),
        "n1_reference_how": f"rank 0 alone over {n_ref} of the same items right before and right after the job (the better of the two), the other ranks idle",
        "n1_reference_before_after": [round(ref_before, 1), round(ref_after, 1)] if rank == 0 else None,
        "queue": "one dynamic queue for the job (store counter, chunks of 4 indices)" if world > 1 else "one rank: its own queue",
        "path": "fnx_jpeg_recompress per item: file bytes up, decoder + quality search + encoder on the device, new file down; no host codec",
        "host_threads_per_rank": workers, "host_decoded_items": host_dec,
        "host_numa_binding": numa_bound or "none (one node, an affinity mask already set, or FENNEC_NO_NUMA_BIND=1)",
        "distinct_files": n_files, "file_bytes_mean": round(sum(len(f) for f in files) / n_files),
        "pcie_bytes": {"up": up, "down": down, "per_item_up": round(up / n_items), "per_item_down": round(down / n_items)},
        "summarize": {"Total": summ.Total, "Succeeded": summ.Succeeded, "Failed": summ.Failed, "TotalSaved": summ.TotalSaved,
                      "AvgSSIM": summ.AvgSSIM, "how": (f"batch.go:140-158; all-reduce over {'RCCL' if red_dev == 'cuda' else 'gloo (test hook)'}" if (world > 1 or dist.is_initialized()) else "batch.go:140-158")},
        "target_ssim": target,
        "note": f"the job is the same {n_items} items at every N (strong scaling: `efficiency_vs_n1` is the figure to read); run after the config-2 timed region, never `value`",
    }


def _pooled_step(fennec_amd, device, ctx0, imgs, one, nctx):
    """step() over `imgs` with `nctx` worker threads, one fnx ctx (= HIP stream + scratch) each, items taken from
    a shared index -- CompressBatch's worker pool (batch.go:84-123) on one GPU.  The per-image ops of configs 3
    and 4 are chains of short kernels with a host round trip at the end (the score): one context leaves the GPU
    idle between them, several keep independent images in flight.  ctypes releases the GIL inside every call."""
    import threading
    nctx = max(1, min(nctx, len(imgs)))
    ctxs = [ctx0] + [fennec_amd.Context(device) for _ in range(nctx - 1)]
    if nctx == 1:
        return lambda: [one(ctx0, a) for a in imgs]
    import torch
    streams = [torch.cuda.Stream(device=device) for _ in ctxs]

    def step():
        torch.cuda.synchronize()
        out = [None] * len(imgs)
        nxt = [0]
        lock = threading.Lock()
        err = []

        def run(c):
            import torch
            torch.cuda.set_device(device)
            # a torch stream per worker: the ctx launches on the caller's current stream (fennec_amd._ordered), so
            # workers sharing torch's default stream would serialise on it
            with torch.cuda.stream(streams[ctxs.index(c)]):
                while True:
                    with lock:
                        i = nxt[0]
                        nxt[0] += 1
                    if i >= len(imgs):
                        return
                    try:
                        out[i] = one(c, imgs[i])
                    except Exception as e:      # surfaced below
                        err.append(e)
                        return
        ts = [threading.Thread(target=run, args=(c,)) for c in ctxs]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if err:
            raise err[0]
        return out
    return step


def _pooled_queue_step(fennec_amd, device, ctx0, n_items, submit, drain, nctx, nthreads=None):
    """step() over n_items with `nctx` worker contexts (one fnx ctx + one HIP stream each) driven by `nthreads` host
    threads: item i belongs to context i mod nctx, context k to thread k mod nthreads.  `submit(ctx, i, out)` enqueues
    item i on ctx (and may fetch an older result of that ctx), `drain(ctx, out)` fetches what is left.  Every call only
    ENQUEUES, so one host thread can keep several streams fed: the workers are CompressBatch's pool (batch.go:84-123)
    with fewer threads than streams because python threads contend for the interpreter lock -- four threads issue these
    calls at 92 us per image between them, two at 34 (tools/host_cost.py).  Threads are persistent, woken per step."""
    import threading
    import torch
    nctx = max(1, min(nctx, n_items))
    nthreads = max(1, min(nthreads or nctx, nctx))
    ctxs = [ctx0] + [fennec_amd.Context(device) for _ in range(nctx - 1)]
    if nctx == 1:
        def step1():
            out = [None] * n_items
            for i in range(n_items):
                submit(ctx0, i, out)
            drain(ctx0, out)
            return out
        return step1
    streams = [torch.cuda.Stream(device=device) for _ in ctxs]

    def run_thread(t, out):
        mine = [k for k in range(nctx) if k % nthreads == t]
        if len(mine) == 1:                                # one stream for the whole step: enter it once
            k = mine[0]
            with torch.cuda.stream(streams[k]):
                for i in range(k, n_items, nctx):
                    submit(ctxs[k], i, out)
                drain(ctxs[k], out)
            return
        for i in range(n_items):                          # round-robin over this thread's streams, in item order
            k = i % nctx
            if k % nthreads == t:
                with torch.cuda.stream(streams[k]):
                    submit(ctxs[k], i, out)
        for k in mine:
            with torch.cuda.stream(streams[k]):
                drain(ctxs[k], out)

    go = [threading.Semaphore(0) for _ in range(nthreads)]
    done = threading.Semaphore(0)
    state = {"out": None, "err": [], "stop": False}

    def worker(t):
        torch.cuda.set_device(device)
        while True:
            go[t].acquire()
            if state["stop"]:
                return
            try:
                run_thread(t, state["out"])
            except Exception as e:
                state["err"].append(e)
            done.release()
    ts = [threading.Thread(target=worker, args=(t,), daemon=True) for t in range(1, nthreads)]
    for t in ts:
        t.start()

    def step():
        state["out"] = [None] * n_items
        for g in go[1:]:
            g.release()
        try:
            run_thread(0, state["out"])                   # the calling thread is worker 0
        except Exception as e:
            state["err"].append(e)
        for _ in ts:
            done.acquire()
        if state["err"]:
            raise state["err"][0]
        return state["out"]

    def close():
        state["stop"] = True
        for g in go[1:]:
            g.release()
        for t in ts:
            t.join()
    step.close = close
    return step


DTYPES = {   # the arithmetic each workload's hot kernels compute in (not a precision claim: every integer output is bit-exact)
    "config3": "u8 (fp32 FMA resize under a rounding guard + fp64 reference-order fix-ups; integer box sums; fp64 SSIM moments)",
    "config4": "u8 (integer Sobel + fp32 AdaptiveSharpen under a rounding guard + fp64 fix-ups: bit-exact; integer milli-luminance, SSIM moments fp32 (--ssim-mode fast, the default of this line) or fp64)",
    "config5": "u8 / int32 (Go image/jpeg's integer DCT, quantisation and Huffman arithmetic; integer box sums; fp64 SSIM moments)",
    "analyze": "u8 -> fp64 luminance sums, integer histogram",
    "palette": "u8 / u32 integer distances",
    "scale-search": "u8 (integer box sums, one fp64 multiply per sample)",
}


def other_workloads(args) -> int:
    """BASELINE.json configs 3, 4, 5 (parity-test cases, not the headline bench line): same
    timing protocol, one JSON line.  Per-image C-ABI calls on device-resident tensors."""
    out = other_workload_line(args)
    if out is not None:
        print(json.dumps(out), flush=True)
    return 0


def other_workload_line(args, embedded: bool = False):
    """The line of one --workload (rank 0: the dict; other ranks: None).  embedded: called from the default run at N = 1
    after its timed region (`other_configs`): no process-group setup, no CPU baseline."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if embedded:
        rank, world, red_dev = 0, 1, "cuda"
        local_rank = torch.cuda.current_device()
        args.no_cpu_baseline = True
    else:
        local_rank, red_dev = _dist_setup(torch, dist, local_rank, world)
    import fennec_amd
    from fennec_amd import batch as fbatch
    from fennec_amd import synth

    ctx = fennec_amd.Context(local_rank)
    wl = args.workload
    if wl == "config3":
        W, H, B = 3840, 2160, min(args.batch, 64)
        imgs = [torch.from_numpy(im).cuda() for im in synth.large_photo_batch(W, H, range(rank * B, rank * B + B))]
        alg = 193.4e6     # 41.47 (down) + 41.47 (implicit up) + 110.5 (MSSSIM) MB

        # the library brackets every resize launch with HIP events: one per lanczosResize where the fused kernel takes it
        # (r3: H into an LDS tile, V out of it), two (resizeH, resizeV) otherwise.  One image through the profiled ctx
        # tells which: down-scale first, then MSSSIM's implicit up-scale.
        ctx.profile(fennec_amd.PROF_RESIZE)
        _small = ctx.lanczosResize(imgs[0], W // 2, H // 2)
        small0 = _small
        ctx.MSSSIM(imgs[0], _small)
        _n = 0
        while True:
            try:
                ctx.kernel_ms()
                _n += 1
            except fennec_amd.FennecError:
                break
        ctx.profile(0)
        RESIZE_KEYS = {4: ["resize_h_down", "resize_v_down", "resize_h_up", "resize_v_up"],
                       2: ["resize_fused_down", "resize_fused_up"]}.get(_n, [f"resize_launch_{k}" for k in range(_n)])
        kms = {k: [] for k in RESIZE_KEYS}

        QD3 = 3                                          # images in flight per context (the ctx's result FIFO holds 4)
        pend3 = {}

        def fetch3(c, out):
            j, _ = pend3[id(c)].pop(0)
            out[j] = c.fetch_result()
            if c is ctx and prof_on[0]:                  # the library's event pairs: H, V of the downscale, H, V of the implicit upscale
                for k in kms:
                    kms[k].append(c.kernel_ms())

        def submit3(c, i, out):
            """lanczosResize (async) + MSSSIM through the result FIFO, results fetched QD3 images behind -- the
            stream never drains between images."""
            pend = pend3.setdefault(id(c), [])
            small = c.lanczosResize(imgs[i], W // 2, H // 2)
            c.msssim_enqueue(imgs[i], small)             # ssim.go:320-322 resizes `small` back to 4K
            pend.append((i, small))
            if len(pend) > QD3:
                fetch3(c, out)

        def drain3(c, out):
            while pend3.get(id(c)):
                fetch3(c, out)

        prof_on = [False]
        CH = max(1, min(getattr(args, "config3_chunk", 8), len(imgs)))
        if CH > 1:
            # r6: a worker's images go through the batched entry points CH at a time -- ONE set of resize launches per chunk and
            # direction (a 4K resize alone is ~700 workgroups: one under-filled round), one FIFO entry per chunk
            nch = (len(imgs) + CH - 1) // CH
            chunks = [imgs[k * CH:(k + 1) * CH] for k in range(nch)]
            small_bufs = [[torch.empty((H // 2, W // 2, 4), dtype=torch.uint8, device="cuda") for _ in ch] for ch in chunks]
            QDC = 2                                          # chunks in flight per context

            def fetch3c(c, out):
                ci, _ = pend3[id(c)].pop(0)
                out[ci] = c.fetch_results(len(chunks[ci]))
                if c is ctx and prof_on[0]:
                    for k in kms:
                        kms[k].append(c.kernel_ms())

            def submit3c(c, ci, out):
                pend = pend3.setdefault(id(c), [])
                smalls = c.lanczosResizeBatch(chunks[ci], W // 2, H // 2, outs=small_bufs[ci])
                c.msssim_batch_enqueue(chunks[ci], smalls)   # ssim.go:320-322: the smalls go back to 4K in one batched resize
                pend.append((ci, smalls))
                if len(pend) > QDC:
                    fetch3c(c, out)

            def drain3c(c, out):
                while pend3.get(id(c)):
                    fetch3c(c, out)

            step_chunks = _pooled_queue_step(fennec_amd, local_rank, ctx, nch, submit3c, drain3c, args.contexts, args.threads)

            def step():
                return [v for part in step_chunks() for v in part]
        else:
            step = _pooled_queue_step(fennec_amd, local_rank, ctx, len(imgs), submit3, drain3, args.contexts, args.threads)
        prof_mask = fennec_amd.PROF_RESIZE
        metric, unit, units_per_step = "megapixels/sec: 4K -> 1920x1080 Lanczos-3 downscale + MS-SSIM", "MP/s", B * W * H / 1e6
        name = "config3: 4K lanczosResize(1920x1080) + MSSSIM(4K, 1080p)"
    elif wl == "config4":
        W, H, B = 7680, 4320, min(args.batch, 8)
        imgs = [torch.from_numpy(im).cuda() for im in synth.large_photo_batch(W, H, range(rank * B, rank * B + B))]
        alg = 530.84e6

        kms = {"windowed_ssim": []}
        prof_on = [False]
        prof_mask = fennec_amd.PROF_SSIM
        QD = 3                                           # images in flight per context (the ctx's result FIFO holds 4)

        pend4 = {}

        def fetch4(c, out):
            j, _ = pend4[id(c)].pop(0)
            out[j] = c.fetch_result()
            if c is ctx and prof_on[0]:
                kms["windowed_ssim"].append(c.kernel_ms())

        ssim_fast = [getattr(args, "ssim_mode", "fast") == "fast"]
        mode_of = {}

        def submit4(c, i, out):
            """AdaptiveSharpen (async) + fnx_ssim_enqueue per image, results fetched QD images behind."""
            pend = pend4.setdefault(id(c), [])
            if mode_of.get(id(c)) != ssim_fast[0]:
                c.set_ssim_mode(ssim_fast[0])
                mode_of[id(c)] = ssim_fast[0]
            sharp = c.AdaptiveSharpen(imgs[i], 0.5)
            c.ssim_enqueue(imgs[i], sharp)
            pend.append((i, sharp))
            if len(pend) > QD:
                fetch4(c, out)

        def drain4(c, out):
            while pend4.get(id(c)):
                fetch4(c, out)

        step = _pooled_queue_step(fennec_amd, local_rank, ctx, len(imgs), submit4, drain4, args.contexts, args.threads)
        metric, unit, units_per_step = "megapixels/sec: 8K AdaptiveSharpen + SSIM", "MP/s", B * W * H / 1e6
        name = "config4: 8K AdaptiveSharpen(0.5) + full-resolution SSIM"
    elif wl == "analyze":     # Analyze (analyze.go:26-124), BenchmarkAnalyze's op at 4K
        W, H, B = 3840, 2160, args.batch
        imgs = [torch.from_numpy(im).cuda() for im in synth.large_photo_batch(W, H, range(rank * B, rank * B + B))]
        torch.cuda.synchronize()
        alg = 4.0 * W * H           # every pixel read once; the sampled passes touch < 1 % more
        plan = ctx.plan_analyze_batch(imgs)
        ctx.profile(True)            # the library brackets analyze_pass_kernel with HIP events
        pass_ms = []

        def step():
            plan.run()                                   # fnx_analyze_batch: results on the host at return
            pass_ms.append(ctx.kernel_ms())
            return [s["Entropy"] for s in plan.stats()]  # + the float epilogue (fennec_statsFromAnalysis)
        metric, unit, units_per_step = "megapixels/sec: 4K Analyze", "MP/s", B * W * H / 1e6
        name = "analyze: Analyze() of 4K images (histogram, brightness, flags, sampled colours / contrast / Sobel)"
    elif wl == "palette":     # applyPalette + palettedToNRGBA (targetsize.go:488-546), 256 colours
        W, H, B = 3840, 2160, min(args.batch, 16)
        imgs = [torch.from_numpy(im).cuda() for im in synth.large_photo_batch(W, H, range(rank * B, rank * B + B))]
        pal = np.random.default_rng(1).integers(0, 256, size=(256, 4), dtype=np.uint8)
        pal[:, 3] = 255
        torch.cuda.synchronize()
        alg = 4.0 * W * H + 1.0 * W * H + 4.0 * W * H      # read src, write indices, write quantized NRGBA

        def step():
            last = None
            for a in imgs:
                last = ctx.applyPalette(a, pal)
            ctx.sync()
            return [float(last[0][0, 0].item())]
        metric, unit, units_per_step = "megapixels/sec: 4K applyPalette (256 colours) + palettedToNRGBA", "MP/s", B * W * H / 1e6
        name = "palette: nearest of 256 palette colours per pixel, indices + quantized NRGBA out"
    elif wl == "scale-search":   # scaleSearch's data movement (targetsize.go:286-313)
        W, H, B = 3840, 2160, min(args.batch, 16)
        imgs = [torch.from_numpy(im).cuda() for im in synth.large_photo_batch(W, H, range(rank * B, rank * B + B))]
        torch.cuda.synchronize()
        # the bisection's 12 scales when the encoder says "fits" below 0.3 (a stand-in for testScaleFits: the
        # codec stays on the host and is not part of this line)
        scales, lo, hi = [], 0.05, 1.0
        for _ in range(12):
            mid = (lo + hi) / 2
            scales.append(mid)
            lo, hi = (mid, hi) if mid <= 0.3 else (lo, mid)
        dims = [(int(W * sc), int(H * sc)) for sc in scales]
        alg = sum(4.0 * W * H + 4.0 * dw * dh for dw, dh in dims) / len(dims)   # per downsample: read src, write dst

        def step():
            last = None
            for a in imgs:
                for dw, dh in dims:
                    last = ctx.boxDownsample(a, dw, dh, to_host=True)    # FNX_DEVICE_SRC: resident source, host result
            return [float(last[0, 0, 0])]
        metric, unit, units_per_step = "megapixels/sec: 4K scaleSearch boxDownsample x12 (source pixels)", "MP/s", B * len(dims) * W * H / 1e6
        name = ("scale-search: 12 boxDownsamples per resident 4K source at scaleSearch's bisection scales, every result "
                "copied to host memory (FNX_DEVICE_SRC)")
    else:   # config5: CompressBatch semantics, host JPEG codec (Pillow) + GPU SSIMFast
        W, H, B = 3840, 2160, min(args.batch, 16)
        dyn = args.queue == "dynamic" and world > 1
        # static: this rank's B items; dynamic: every rank can serve any of the job's world * B items
        ks = range(world * B) if dyn else range(rank * B, rank * B + B)
        srcs = synth.large_photo_batch(W, H, ks)
        jpegs = [fbatch.pillow_encode(s, 92) for s in srcs]          # "4096 synthetic 4K JPEGs", q=92 up front
        NI = len(jpegs)
        workers = args.workers or max(1, min(8, host_cpus()[0] // max(world, 1)))
        alg = 2 * 4 * W * H * 6.0                                     # ~6 search steps x (H2D + read)
        gpu_stage = []          # seconds inside the C ABI per item (prepare + every against), all workers
        if args.device_decode:
            work = fbatch.jpeg_item_work_device_all(jpegs, fbatch.TARGET_SSIM["Balanced"], on_gpu_seconds=gpu_stage.append)
        elif args.device_codec:
            work = fbatch.jpeg_item_work_device_codec(jpegs, fbatch.TARGET_SSIM["Balanced"], on_gpu_seconds=gpu_stage.append)
        elif args.device_search:
            work = fbatch.jpeg_item_work_device_search(jpegs, fbatch.TARGET_SSIM["Balanced"], on_gpu_seconds=gpu_stage.append)
        else:
            work = fbatch.jpeg_item_work(jpegs, fbatch.TARGET_SSIM["Balanced"], on_gpu_seconds=gpu_stage.append)

        states = {}

        def make_state(wid):
            if wid not in states:
                states[wid] = fennec_amd.Context(local_rank)
            return states[wid]

        def step():
            if dyn:
                res = fbatch.compress_batch(NI, work, make_state, workers=workers, rank=rank, world=world, queue_mode="dynamic")
            else:
                res = fbatch.compress_batch(NI, work, make_state, workers=workers)
            return [r.SSIM for r in res] or [float("nan")]
        metric, unit, units_per_step = "images/sec: CompressBatch 4K JPEG, SSIM-guided quality search", "images/s", B
        name = f"config5: {B} 4K JPEGs per step per GPU, Balanced (SSIM>=0.94) binary search, Pillow codec on {workers} host threads"
        if dyn:
            name += "; one dynamic queue over all ranks"
        if args.device_decode:
            name = (f"config5: {B} 4K JPEGs per step per GPU, Balanced (SSIM>=0.94) binary search; decoder, search and encoder on the device "
                    f"(fnx_jpeg_recompress: Go image/jpeg's arithmetic and file layout), no host codec, {workers} host threads")
            if dyn:
                name += "; one dynamic queue over all ranks"
        elif args.device_codec:
            name += ("; search and encoder on the device (fnx_jpeg_compress: Go image/jpeg's arithmetic and file layout), host codec: "
                     "1 decode per image")
        elif args.device_search:
            name += ("; search on the device (Go image/jpeg arithmetic without entropy coding, fnx_jpeg_quality_search), host "
                     "codec: 1 decode + 1 encode per image")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()
    cold = None
    if wl in ("config3", "config4"):
        # the same steps BEFORE this workload's own pre-warm: two steps for scratch growth, then eight timed
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        cold = units_per_step * 8 / (time.perf_counter() - t_c)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm:      # untimed: GPU clock ramp, as in the config-2 path
        step()
    for _ in range(args.warmup):
        vals = step()
    if wl in ("config3", "config4"):
        ctx.profile(prof_mask)      # the library brackets the dominant kernels with HIP events on its stream
        prof_on[0] = True
    barrier()
    step_s = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t_s = time.perf_counter()
        vals = step()
        step_s.append(time.perf_counter() - t_s)
    barrier()
    elapsed = time.perf_counter() - t0
    if wl in ("config3", "config4"):
        prof_on[0] = False
        ctx.profile(0)
    other_mode = None
    if wl == "config4":
        # the same steps in the OTHER arithmetic of full-resolution SSIM (fp64 moments when the line is the fp32 form and vice
        # versa), right after the timed region, same protocol, half the steps; never `value`
        kms_main = list(kms["windowed_ssim"])
        name_main = ctx.last_kernel(fennec_amd.PROF_SSIM)
        ssim_fast[0] = not ssim_fast[0]
        for _ in range(2):
            vals_o = step()
        kms["windowed_ssim"].clear()
        ctx.profile(prof_mask)
        prof_on[0] = True
        n_o = max(4, args.steps // 2)
        barrier()
        t_o = time.perf_counter()
        for _ in range(n_o):
            vals_o = step()
        barrier()
        dt_o = time.perf_counter() - t_o
        prof_on[0] = False
        ctx.profile(0)
        other_mode = {"mode": "fast" if ssim_fast[0] else "exact", "value": round(units_per_step * world * n_o / dt_o, 2), "unit": unit, "steps": n_o,
                      "ms_per_step": round(dt_o / n_o * 1e3, 4), "kernel": ctx.last_kernel(fennec_amd.PROF_SSIM),
                      "avg_launch_ms": round(float(np.mean(kms["windowed_ssim"])), 4) if kms["windowed_ssim"] else None,
                      "result_sample": float(vals_o[0]), "delta_vs_line": float(vals_o[0]) - float(vals[0])}
        ssim_fast[0] = not ssim_fast[0]
        kms["windowed_ssim"][:] = kms_main
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = units_per_step * world * args.steps / elapsed
    gbs = alg * (B * 12 if wl == "scale-search" else B) * args.steps / elapsed / 1e9
    out = {
        "metric": metric, "value": round(value, 2), "unit": unit, "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": DTYPES.get(wl, "u8"), "data": "synthetic",
        "config": {"workload": name, "images_per_step_per_gpu": B, "width": W, "height": H,
                   "inputs": "device-resident, per-image C-ABI calls" if wl != "config5" else ("host JPEG bytes; the file is all that crosses PCIe" if getattr(args, "device_decode", False) else "host JPEG bytes, FNX_HOST staging per search step")},
        "roofline": {"kernel": "whole step (all kernels of the workload)", "bound": "hbm", "achieved": round(gbs, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None},
        "result_sample": float(vals[0]),
        "step_ms": {"min": round(min(step_s) * 1e3, 4), "median": round(float(np.median(step_s)) * 1e3, 4),
                    "max": round(max(step_s) * 1e3, 4)},
    }
    if wl in ("config3", "config4"):
        out["config"]["contexts_per_gpu"] = max(1, min(args.contexts, B))
        if wl == "config3":
            out["config"]["calls"] = (f"batched: fnx_lanczos_resize_batch + fennec_MSSSIM_batch_enqueue, {CH} images per call" if CH > 1
                                      else "one lanczosResize + one MSSSIM_enqueue per image")
        out["config"]["prewarm"] = f"{args.prewarm} s of untimed steps before the {args.warmup} warm-up steps (GPU clock ramp)"
        out["config"]["host_threads_per_gpu"] = max(1, min(args.threads or args.contexts, args.contexts, B))
        out["value_before_prewarm"] = {"value": round(cold * world, 2), "unit": unit,
                                       "note": "8 steps timed right after set-up (2 untimed steps for scratch growth), before this workload's own "
                                               f"{args.prewarm} s pre-warm: what the clock governor's state costs; never `value`"}
        out["roofline_step"] = out["roofline"]
        S_img = 4.0 * W * H
        if wl == "config4":
            ms_flow = float(np.mean(kms["windowed_ssim"]))   # in the workload's flow: AdaptiveSharpen of the next image runs beside it
            sharp0 = ctx.AdaptiveSharpen(imgs[0], 0.5)
            ctx.set_ssim_mode(ssim_fast[0])
            ms = kernel_alone_ms(ctx, lambda: ctx.SSIM(imgs[0], sharp0), fennec_amd.PROF_SSIM)
            abytes = 2.0 * S_img                      # SSIM reads both full-size images once
            g = abytes / (ms * 1e-3) / 1e9
            # the kernel is fp64-VALU bound: 4 moments x 8 taps x 2 passes of fp64 FMA per window
            win = float(W - 8) * float(H - 8)
            fastk = name_main == "windowed_ssim_march2f_kernel"
            out["ssim_mode"] = ("fast: FNX_SSIM_FAST, fp32 moments in the cancellation-free form, |delta| <= 1e-6 (SURVEY Appendix A), measured <= 6e-8"
                                if fastk else "exact: fp64 moments, |delta| <= 1e-9 (the library's default)")
            out["other_ssim_mode"] = other_mode
            out["roofline"] = {"kernel": f"{name_main} (full-resolution SSIM of one 8K pair, two pixel columns per lane, luminance fused"
                                         + (", fp32 moments)" if fastk else ", fp64 moments)"),
                               "bound": "valu", "achieved": round(g, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(g / HBM_PEAK_GBS, 4),
                               "traffic": committed_traffic_named(name_main, "config4"),
                               "traffic_file": source_file("traffic"),
                               "traffic_source": "the newest profiles/*config4*_traffic.json (named in traffic_file): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (committed, not measured in this run)",
                               "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": round(ms, 4), "launches_timed": 9,
                               "avg_launch_ms_is": "the kernel ALONE (one call at a time after the timed region): what `achieved`, `frac`, `traffic` and "
                                                   "`hbm_frac_measured` are all about -- one collection, one context",
                               "avg_launch_ms_in_flow": round(ms_flow, 4), "launches_timed_in_flow": len(kms["windowed_ssim"]),
                               "fma_floor_ms": round(win * 64 / (78.6e12 if fastk else 39.3e12) * 1e3, 4),
                               "note": ("bound by instruction issue: 64 FMA per window -- the floor shown is at 78.6 T fp32 FMA/s, which needs "
                                        "packed instructions AND more than two waves per SIMD; at two, every VALU instruction costs a quad-cycle "
                                        "(experiments/ssimf/pkrate.hip)" if fastk else
                                        "bound in practice by fp64 VALU issue + LDS (64 fp64 FMA per window at 39.3 T FMA/s is the floor shown)")
                                       + "; bytes are SURVEY 8(d)'s 2*S per pair"}
            tr4 = out["roofline"]["traffic"]
            out["roofline"]["hbm_frac_measured"] = round(tr4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tr4 else None
        else:
            means = {k: float(np.mean(v)) for k, v in kms.items() if v}
            fused = "resize_fused_down" in kms
            dom = "resize_fused_down" if fused else next(iter(kms))
            ms_flow = means.get(dom, float("nan"))     # in the workload's flow: the other worker streams' kernels run beside it
            # the same downscale alone, the plan in the state the workload left it in (on the ramp: its tie-dense cool-down)
            if CH > 1:
                ups = [torch.empty((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(CH)]
                ms = kernel_alone_ms(ctx, lambda: ctx.lanczosResizeBatch(chunks[0], 1920, 1080, outs=small_bufs[0]), fennec_amd.PROF_RESIZE) / CH
                ms_up_alone = kernel_alone_ms(ctx, lambda: ctx.lanczosResizeBatch(small_bufs[0], W, H, outs=ups), fennec_amd.PROF_RESIZE) / CH
                del ups
                ms_flow /= CH
                means = {k: v / CH for k, v in means.items()}
            else:
                ms = kernel_alone_ms(ctx, lambda: ctx.lanczosResize(imgs[0], 1920, 1080), fennec_amd.PROF_RESIZE)
                ms_up_alone = kernel_alone_ms(ctx, lambda: ctx.lanczosResize(small0, W, H), fennec_amd.PROF_RESIZE) if small0 is not None else None
            # fused: reads S(4K), writes the 1080p result; two-pass: resizeH reads S(4K), writes the 1920 x 2160 intermediate
            abytes = S_img + S_img / 4 if fused else S_img + S_img / 2
            g = abytes / (ms * 1e-3) / 1e9
            # which form of the one-launch kernel the downscale takes in this workload: on SURVEY 8(d)'s ramp the plan is in its
            # tie-dense cool-down from the 66th call on -- the committed profile of this same command says
            # which kernel it launched
            down_traffic, down_file, down_name = None, None, "resize_fused_kernel<4>"
            for pat, label in (("resize_dense21_kernel", "resize_dense21_kernel (a plan in its second tie-dense cool-down: the exact 2:1 form, "
                                                         "weights as kernel arguments, 74-row tiles)"),
                               ("resize_fused_dense_kernel<4", "resize_fused_dense_kernel<4> (a plan in its second tie-dense cool-down: "
                                                               "the tile without fp32 passes)"),
                               ("resize_fused_kernel<4", "resize_fused_kernel<4>")):
                t = committed_traffic_named(pat, "config3", per_image=True)
                if t is not None:
                    down_traffic, down_file, down_name = t, source_file("traffic"), label
                    break
            out["roofline"] = {"kernel": (down_name +
                                          " (lanczosResize 4K -> 1080p in one launch: resizeH into an LDS tile, resizeV out of it; "
                                          "the largest single kernel of the step)") if fused else
                                         "resize_h_guard_kernel<4> (resizeH of the 4K -> 1080p downscale: the largest single kernel of the step)",
                               "bound": "valu" if fused else "hbm", "achieved": round(g, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(g / HBM_PEAK_GBS, 4),
                               "traffic": down_traffic if fused else committed_traffic_named("resize_h_guard_kernel", "config3"),
                               "traffic_file": down_file if fused else source_file("traffic"),
                               "traffic_source": "the newest profiles/*config3*_traffic.json (named in traffic_file): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (committed, not measured in this run)",
                               "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": round(ms, 4),
                               "launches_timed": 9,
                               "avg_launch_ms_is": "the kernel ALONE (one call at a time after the timed region, the plan in the state the workload left): "
                                                   "what `achieved`, `frac`, `traffic` and `hbm_frac_measured` are all about -- one collection, one context",
                               "avg_launch_ms_in_flow": round(ms_flow, 4), "launches_timed_in_flow": len(kms[dom]),
                               "upscale_alone_ms": round(ms_up_alone, 4) if ms_up_alone else None,
                               "images_per_launch": CH,
                               "per": "image (a launch holds images_per_launch of them: durations and bytes are the launch's divided by that)",
                               "resize_kernels_ms_in_flow": {k: round(v, 4) for k, v in means.items()},
                               "note": "SURVEY 8(d)'s synthetic ramp makes every output of the 2:1 downscale an exact rounding tie, so this "
                                       "kernel runs its fp64 reference-order loops (VALU-bound); the step is 7-9 kernels per image (1-2 launches "
                                       "per resize, level 0 + level 1 in one pass, pyramid, boxes, windows, finish): see roofline_step for the "
                                       "whole step against SURVEY 8(d)'s 193.4 MB"}
            tr3 = out["roofline"]["traffic"]
            out["roofline"]["hbm_frac_measured"] = round(tr3 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tr3 else None
            if rank == 0:
                out["roofline"]["photo_like"] = resize_photo_like(ctx, W, H)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_other(wl, W, H)
    if wl == "config5" and rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_config5(jpegs[:2], fbatch.TARGET_SSIM["Balanced"])
    if wl == "config5":
        per_item = float(np.mean(gpu_stage[-B * args.steps:]))
        out["gpu_stage"] = {"seconds_per_image": round(per_item, 6), "images_per_s_per_context": round(1.0 / per_item, 1),
                            "note": ("time inside the C ABI: the whole item (file bytes up, decode + search + entropy coder on the device, "
                                     "the new file down)") if args.device_decode else
                                    ("time inside the C ABI (one H2D of the decoded source, the whole search and the entropy coder on the "
                                     "device, the file's D2H); the rest of a step is the host decode of the source") if args.device_codec else
                                    ("time inside the C ABI (one H2D of the decoded source + the whole search on the device); the rest "
                                     "of a step is the host codec: one decode and one encode per image") if args.device_search else
                                    ("time inside the C ABI (prepare + every SSIMFast of the search, host buffers: PCIe-inclusive); "
                                     "the rest of a step is the host JPEG codec (Pillow here, Go's image/jpeg in the reference)")}
        if rank == 0 and world == 1 and args.device_decode and not args.no_extras:
            # the same items through the C++ pool (fennec_CompressBatchJPEG: no interpreter between the items)
            pools = {}
            for nw in (4, 8, 16):
                fbatch.compress_batch_jpeg_native(jpegs[:B], fbatch.TARGET_SSIM["Balanced"], workers=nw)
                t_g = time.perf_counter()
                for _ in range(3):
                    fbatch.compress_batch_jpeg_native(jpegs[:B], fbatch.TARGET_SSIM["Balanced"], workers=nw)
                pools[str(nw)] = round(3 * B / (time.perf_counter() - t_g), 1)
            out["native_pool"] = {"images_per_s_by_workers": pools, "unit": "images/s",
                                  "note": "fennec_CompressBatchJPEG over the same files, 3 runs per worker count after the timed region "
                                          "(python-side buffer handling included)"}
        if rank == 0 and world == 1 and args.device_codec and not args.no_extras:
            # the GPU stage alone -- decoded sources resident on the device, the C++ pool of
            # fennec_CompressBatchNRGBA (search + entropy coder per item on the device), files copied to the host
            dres = [torch.from_numpy(s_).cuda() for s_ in srcs[:B]]
            torch.cuda.synchronize()
            fbatch.compress_batch_native(dres, workers=4)
            t_g = time.perf_counter()
            for _ in range(3):
                fbatch.compress_batch_native(dres, workers=4)
            t_g = (time.perf_counter() - t_g) / 3
            out["gpu_stage_only"] = {"value": round(len(dres) / t_g, 1), "unit": "images/s", "workers": 4,
                                     "note": "fennec_CompressBatchNRGBA over device-resident decoded sources, 3 runs after the timed region "
                                             "(python-side buffer handling included; tools/time_batch_native.py times the pool alone)"}
            del dres
    if wl == "palette":
        # the kernel is bound by VALU issue, not by HBM (palette.hip): 3 instructions per (pixel, palette entry) -- a 4-clock
        # v_dot4_u32_u8 and two 2-clock integer ops -- on 1024 SIMDs; achieved / peak stay the algorithmic bytes over HBM
        clk, sec_img = 2.4e9, elapsed / args.steps / B
        floor = (W * H / 64.0) * 256 * 8.0 / (1024 * clk)
        out["roofline"].update({"kernel": "apply_palette_kernel (nearest of 256 entries, first minimum wins; one launch per image)",
                                "bound": "valu", "valu_issue_frac": round(floor / sec_img, 4), "valu_floor_ms_per_image": round(floor * 1e3, 4),
                                "ms_per_image": round(sec_img * 1e3, 4),
                                "note": "8 issue clocks per palette entry per wave of 64 pixels at 2.4 GHz is the floor shown; launch, table "
                                        "upload and sync of the per-image call are inside ms_per_image"})
    if wl == "analyze":
        ms = float(np.mean(pass_ms[-args.steps:]))
        g = alg * B / (ms * 1e-3) / 1e9
        out["config"]["inputs"] = "device-resident, one batched C-ABI call per step (fnx_analyze_batch)"
        out["roofline_step"] = out["roofline"]
        out["roofline"] = {"kernel": f"analyze_pass_kernel (histogram + brightness + flags, one launch of {B} images)",
                           "bound": "hbm", "achieved": round(g, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(g / HBM_PEAK_GBS, 4), "traffic": None,
                           "algorithmic_bytes_per_launch": alg * B, "avg_launch_ms": round(ms, 4)}
    if wl == "analyze" and rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        img = imgs[0].cpu().numpy()
        orc.analyze(img)
        t1 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t1 < 5.0:
            orc.analyze(img)
            reps += 1
        dt = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(reps * W * H / 1e6 / dt, 2), "unit": "MP/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} x Analyze(4K) in {dt:.1f} s, oracle/fennec_oracle.c (serial, as the reference)"}
    if hasattr(step, "close"):
        step.close()                       # the worker threads of _pooled_queue_step
    if world > 1 and not embedded:
        dist.destroy_process_group()
    return out if rank == 0 else None


_SOURCES: dict = {}          # which committed profile file the last committed_*() call read (newest round first)


def source_file(kind: str):
    """profiles/ file the last committed_traffic*() / committed_valu_issue() call took its number from (None: no file had it)"""
    return _SOURCES.get(kind)


def committed_valu_issue(kernel_substr: str, launch_ms: float, scored: bool = False, exact: bool = False):
    """Fraction of the chip's VALU issue slots the kernel fills: SQ_ACTIVE_INST_VALU (quad-cycles, all SIMDs; from the
    committed PMC pass of this same command) x 4 clocks / (1024 SIMDs x launch duration x shader clock).  The clock is
    GRBM_GUI_ACTIVE / duration of the profiled launches when that counter was collected, else 2.1 GHz (the steady
    clock of this kernel).  None when no profile has been committed."""
    import glob
    import re
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.txt")), reverse=True):
        try:
            act = clk = None
            for line in open(p):
                m = re.match(r"(\S+)\s+([0-9.]+)\s+\(mean of \d+ launches\)\s+(.*)", line)
                if not m or kernel_substr not in m.group(3) or _template_flags(m.group(3)) != (scored, exact):
                    continue
                if m.group(1) == "SQ_ACTIVE_INST_VALU":
                    act = float(m.group(2))
                if m.group(1) == "GRBM_GUI_ACTIVE_PER_XCD_PER_MS":
                    clk = float(m.group(2)) * 1e3      # Hz
            if act is not None:
                hz = clk or 2.1e9
                _SOURCES["valu_issue"] = os.path.relpath(p, ROOT)
                return round(act * 4.0 / (1024.0 * launch_ms * 1e-3 * hz), 4)
        except Exception:
            continue
    return None


def committed_traffic_named(kernel_substr: str, tag: str, per_image: bool = False):
    """HBM bytes per launch of a kernel from profiles/*<tag>*_traffic.json (see committed_traffic); None if absent.
    per_image: divided by the file's images_per_launch (config 3's batched launches hold several images)."""
    import glob
    _SOURCES["traffic"] = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*{tag}*_traffic.json")), reverse=True):   # r05 before r04 ...
        try:
            t = json.load(open(p))
            for k, v in t["kernels"].items():
                if kernel_substr in k:
                    _SOURCES["traffic"] = os.path.relpath(p, ROOT)
                    return float(v["hbm_bytes_per_launch"]) / (max(1, int(t.get("images_per_launch", 1))) if per_image else 1)
        except Exception:
            continue
    return None


def kernel_alone_ms(ctx, fn, mask, reps=9, warm_s=0.15):
    """The profiled kernel(s) of ONE call at a time with nothing else on the device (the call's stream drained between calls):
    the duration a single-context rocprofv3 collection sees, and the one its counter traffic belongs with."""
    import fennec_amd
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        fn()
        ctx.sync()
    ctx.profile(mask)
    ts = []
    for _ in range(reps):
        fn()
        ctx.sync()
        t = 0.0
        while True:
            try:
                t += ctx.kernel_ms()
            except fennec_amd.FennecError:
                break
        ts.append(t)
    ctx.profile(0)
    return float(np.mean(ts))


def resize_photo_like(ctx, W, H):
    """config 3's two resizes on content WITHOUT exact rounding ties (blurred noise): the kernel time of one call at a time, from
    the library's own event pairs, after the timed region.  The 2:1 downscale takes resize_mfma_kernel (csrc/resize_mfma.hip) +
    resize_fused_sparse_kernel here; on SURVEY 8(d)'s ramp it is handed back whole and the ctx goes back to resize_fused_kernel."""
    import torch
    import fennec_amd
    from fennec_amd import synth
    img = ctx.GaussianBlur(ctx.GaussianBlur(torch.from_numpy(synth.noise_image(W, H, 5)).cuda(), 2.0), 1.2)
    small = ctx.lanczosResize(img, W // 2, H // 2)
    ctx.sync()
    res = {}
    for name, (src, dw, dh) in (("down", (img, W // 2, H // 2)), ("up", (small, W, H))):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15:
            ctx.lanczosResize(src, dw, dh)
            ctx.sync()
        ctx.profile(fennec_amd.PROF_RESIZE)
        ts = []
        for _ in range(9):
            ctx.lanczosResize(src, dw, dh)
            t = 0.0
            while True:
                try:
                    t += ctx.kernel_ms()
                except fennec_amd.FennecError:
                    break
            ts.append(t)
        ctx.profile(0)
        ms = float(np.mean(ts))
        S = 4.0 * (W * H + (W // 2) * (H // 2))
        res[name] = {"kernel_ms": round(ms, 4), "GB/s": round(S / (ms * 1e-3) / 1e9, 1), "frac": round(S / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    res["note"] = "one call at a time (no other stream running), kernels only; algorithmic bytes = source + destination"
    return res


def cpu_baseline_other(wl: str, W: int, H: int) -> dict:
    """configs 3 / 4 on the host cores with the oracle (a C restatement of the Go reference, its threading model),
    on a bounded sample: one 4K image for config 3; a 1920 x 1080 crop of the 8K image for config 4 (every stage is
    per-pixel work with fixed-size neighbourhoods, so MP/s carries over; the full 8K SSIM alone is ~10 s a pass)."""
    from oracle import oracle
    from fennec_amd import synth
    cores = host_cpus()[0]
    if wl == "config3":
        img = synth.large_photo(W, H, 0)
        n, t0 = 0, time.perf_counter()
        while True:
            small = oracle.lanczos_resize(img, W // 2, H // 2, procs=cores)
            oracle.msssim(img, small, procs=cores)
            n += 1
            dt = time.perf_counter() - t0
            if dt > 12.0 or n >= 16:
                break
        return {"value": round(n * W * H / 1e6 / dt, 2), "unit": "MP/s", **_cores_fields(), "kind": "port",
                "sample": f"{n} x (lanczosResize 4K -> 1080p + MSSSIM(4K, 1080p)) in {dt:.1f} s, oracle/fennec_oracle.c with procs={cores}"}
    cw, ch = 1920, 1080
    img = np.ascontiguousarray(synth.large_photo(W, H, 0)[:ch, :cw])
    n, t0 = 0, time.perf_counter()
    while True:
        sharp = oracle.adaptive_sharpen(img, 0.5, procs=cores)
        oracle.ssim(img, sharp, procs=cores)
        n += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or n >= 64:
            break
    return {"value": round(n * cw * ch / 1e6 / dt, 2), "unit": "MP/s", **_cores_fields(), "kind": "port",
            "sample": f"{n} x (AdaptiveSharpen(0.5) + full-resolution SSIM) of a {cw}x{ch} crop of the 8K image in {dt:.1f} s, "
                      f"oracle/fennec_oracle.c with procs={cores}"}


def cpu_baseline_config5(files, target: float) -> dict:
    """CompressBatch on the host cores with the oracle standing in for Go's image/jpeg + ssim.go (the arithmetic the
    device path is checked against): per item decode, compressJPEGOptimal's binary search with a round trip + SSIMFast per
    candidate, the winner's file -- one item per worker thread as batch.go runs them (the oracle's C calls release the
    interpreter lock), each item single-threaded.  A bounded sample: one item per worker, at most 32 workers."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    workers = max(1, min(32, host_cpus()[0]))

    def item(data):
        src = oracle.jpeg_decode(data)
        lo, hi, best_q = 30, 100, 100            # compress.go:24-74 at the Balanced preset's bounds
        steps = 0
        while lo <= hi and steps < 7:
            mid = (lo + hi) // 2
            s = oracle.ssim_fast(src, oracle.jpeg_roundtrip(src, mid), procs=1)
            steps += 1
            if s >= target:
                best_q, hi = mid, mid - 1
            else:
                lo = mid + 1
        return len(oracle.jpeg_encode(src, best_q))

    work = [files[k % len(files)] for k in range(workers)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as ex:
        sizes = list(ex.map(item, work))
    dt = time.perf_counter() - t0
    return {"value": round(len(sizes) / dt, 3), "unit": "images/s", "cores": workers, "kind": "port",
            "sample": f"{len(sizes)} x (decode + quality search + encode of a 4K JPEG), one per thread, in {dt:.1f} s; oracle/fennec_oracle.c, "
                      f"each item single-threaded as in batch.go"}


def _template_flags(kernel_name: str):
    """(SCORE, GUARD) of a blur_mfma_kernel<SCORE, GUARD> or blur_direct_kernel<R, NTH, IH, SCORE, RA, RB, GUARD> instantiation name."""
    try:
        args = [a.strip() for a in kernel_name[kernel_name.index("<") + 1:kernel_name.index(">")].split(",")]
        if "blur_mfma_kernel" in kernel_name:
            return args[0] == "true", args[1] == "true"
        return args[3] == "true", (len(args) > 6 and args[6] == "true")
    except (ValueError, IndexError):
        return False, False


def committed_traffic(kernel_substr: str, batch: int, scored: bool = False, exact: bool = False):
    """HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs of this same command, FETCH doubled per the gfx950
    correction).  None when no profile of this batch size has been committed."""
    import glob
    _SOURCES["traffic"] = None
    _SOURCES["valu_issue"] = _SOURCES.get("valu_issue")
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):          # r05 before r04 ...
        try:
            t = json.load(open(p))
            for k, v in t["kernels"].items():
                if kernel_substr in k and _template_flags(k) == (scored, exact) and batch == int(t.get("images_per_launch", 32)):
                    _SOURCES["traffic"] = os.path.relpath(p, ROOT)
                    return float(v["hbm_bytes_per_launch"])
        except Exception:
            continue
    return None


def host_cpus():
    """CPUs this process can really use: its affinity mask capped by the cgroup's CPU quota (cpu.max / cfs_quota_us).  The boxes of
    this pool show 256 logical CPUs and a quota of 16: threads beyond the quota only add throttling stalls, and a baseline that
    says "256 cores" there would describe a machine the job never had.  -> (usable, quota or None, visible)"""
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                quota = q / p
        except Exception:
            pass
    usable = visible if not quota else max(1, min(visible, int(quota + 0.999)))
    return usable, quota, visible


def _cores_fields():
    usable, quota, visible = host_cpus()
    return {"cores": usable, "cpus_visible": visible, "cpu_quota": None if quota is None else round(quota, 2)}


def cpu_baseline(img: np.ndarray) -> dict:
    """The oracle (C restatement of the Go reference, same threading model: static row/column
    split over T threads, boxDownsample/toLuminance serial) timed on this box's host cores on a
    bounded sample of the same workload.  A reported baseline, not the optimisation target."""
    from oracle import oracle
    cores = host_cpus()[0]
    n = 0
    t0 = time.perf_counter()
    while True:
        b = oracle.gaussian_blur(img, SIGMA, procs=cores)
        oracle.ssim_fast(img, b, procs=cores)
        n += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or (n >= 16 and dt > 3.0) or n >= 256:
            break
    t1 = time.perf_counter()                     # also at T = 1
    b = oracle.gaussian_blur(img, SIGMA, procs=1)
    oracle.ssim_fast(img, b, procs=1)
    dt1 = time.perf_counter() - t1
    import shutil
    return {
        "value": round(n * W4K * H4K / 1e6 / dt, 2),
        "unit": "MP/s",
        **_cores_fields(),
        "kind": "port",
        "sample": f"{n} x (4K GaussianBlur sigma=2 + SSIMFast) in {dt:.1f} s, oracle/fennec_oracle.c with procs={cores}",
        "value_1_thread": round(W4K * H4K / 1e6 / dt1, 2),
        "go_toolchain": "present (not used)" if shutil.which("go") else "absent: the Go reference itself cannot be timed",
    }


if __name__ == "__main__":
    sys.exit(main())
