/*
 * fennec_hip.h -- C ABI of libfennec_hip.so: fennec's per-pixel hot path on
 * AMD Instinct MI355X (gfx950 / CDNA4), hand-written HIP kernels.
 *
 * The reference (shamspias/fennec, pure Go) has no FFI seam; the seam is the set
 * of Go functions whose BODIES a cgo shim replaces.  Every
 * entry point below names the reference function it replaces (file:line).
 * INTEGRATION.md shows the cgo binding.
 *
 * Two layers, both exported:
 *
 *   fnx_*     kernel level.  Takes weight tables (SSIM 8x8 window, blur 1-D
 *             kernel, Lanczos CSR taps) as INPUTS, so a Go caller passes tables
 *             computed with Go's own math.Exp / math.Sin and gets results
 *             identical to the reference's.  This is what cgo binds.
 *   fennec_*  the reference's exported/unexported function set mirrored in C++
 *             above fnx_* (guards, control flow, table generation with libm) --
 *             the host side as it exists where no Go toolchain is available;
 *             also what the Python test/bench harness binds.
 *
 * Conventions
 *   - Images are 8-bit non-premultiplied R,G,B,A ("NRGBA", Go image.NRGBA.Pix):
 *     pixel (x,y) lives at pix[y*stride + 4*x]; stride is in BYTES; Rect.Min is
 *     ignored exactly as the reference's kernels ignore it.
 *   - `space` says where image pointers live: FNX_HOST (the library stages
 *     through its own pinned/device buffers; call returns when dst is filled)
 *     or FNX_DEVICE (HIP device pointers of the ctx's device; the op is enqueued
 *     on the ctx stream and returns immediately unless it has a host scalar
 *     output).  Tables and scalar outputs are always HOST pointers.
 *   - Inputs are never written; C never retains a caller pointer past return
 *     (cgo rule), except device pointers of in-flight FNX_DEVICE ops until
 *     fnx_ctx_sync().
 *   - A fnx_ctx owns one device, one stream and its scratch.  It is NOT
 *     re-entrant: one ctx per worker thread (HIP's current device is per OS
 *     thread and goroutines migrate, so every call binds the ctx's device).
 *   - Return codes: FNX_OK; FNX_NOOP = the reference returns the SAME image
 *     (dst untouched); FNX_EMPTY = the reference returns a 0x0 image (dst
 *     untouched); negative = error, text via fnx_last_error().  There is NO
 *     CPU fallback inside this library: without a usable GPU every op fails
 *     with FNX_ERR_NO_DEVICE.
 *   - Environment: a release build reads exactly these names, each once --
 *       FENNEC_HIP_DISABLE=1    no device is used (fnx_device_count() == 0)
 *       FENNEC_HIP_DEVICES=0,2  which HIP devices, in which order
 *       FNX_ROCTX=1             a roctx range per exported call (tracing)
 *       FNX_POOL_TRACE=1        per-item wall times of the C++ batch pools on stderr
 *       FNX_JPEG_TRACE=1        the device decoder's pass counts on stderr
 *     None of them changes a result or the kernel a call takes.  Which kernel
 *     FORM a ctx takes where several compute the same bytes is a per-ctx
 *     selection (fnx_ctx_set_form); development switches exist only in
 *     `make DEVELOP=1` builds.
 */
#ifndef FENNEC_HIP_H
#define FENNEC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library itself is built with -fvisibility=hidden */
#endif

#define FNX_OK 0
#define FNX_NOOP 1
#define FNX_EMPTY 2
#define FNX_ERR_INVALID (-1)
#define FNX_ERR_NO_DEVICE (-2)
#define FNX_ERR_HIP (-3)
#define FNX_ERR_OOM (-4)
#define FNX_ERR_UNSUPPORTED (-5) /* fnx_jpeg_decode / fnx_jpeg_recompress: a file the device decoder does not handle; decode it on the host */

/* most images one *_batch / *_batch_enqueue call of round 6 takes (the image is a grid dimension of the launch) */
#define FNX_BATCH_MAX 65535

#define FNX_HOST 0
#define FNX_DEVICE 1
/* image -> image ops only (blur, blur3x3, sharpen, resize, box downsample, orient): the source is device
 * memory (fnx_malloc + fnx_upload, or an earlier device result), the destination is host memory and the
 * call returns when it is filled.  For search loops over ONE resident source -- the scale searches of
 * targetsize.go:240-313 call boxDownsample(src, w, h) 10-12 times and encode each result on the host --
 * so that the source crosses PCIe once. */
#define FNX_DEVICE_SRC 2

/* fnx_gaussian_blur flags */
#define FNX_BLUR_FAST 0  /* fp32 FMA accumulation: <=1 LSB off the reference on <=0.1% of samples */
#define FNX_BLUR_EXACT 1 /* bit-exact: fp32 with a rounding guard, fp64 in the reference's tap order for every
                            sample the guard cannot decide (and for kernels with negative taps or gain > 1) */

/* fnx_gaussian_blur_batch (and fnx_gaussian_blur with FNX_DEVICE: a batch of one, scored by fnx_ssim_fast), OR-ed to either
 * mode: "the next call on this ctx is fnx_ssim_fast_batch(_enqueue) over exactly
 * these (srcs[i], dsts[i]) pairs, and nothing writes the images in between".  The blur then runs as the one-pass kernel of
 * fnx_gaussian_blur_ssim_fast_batch -- same blurred bytes -- which also takes SSIMFast's boxDownsample sums (ssim.go:244-309)
 * of both sides, and that scoring call reads neither full-size image again: the reference's two calls (effects.go:146, then
 * ssim.go:48) at the one-pass traffic of 2 S instead of 4 S.  Scores are those of fnx_gaussian_blur_ssim_fast_batch (integer
 * box sums, then the same code).  Any other call on the ctx (fnx_ctx_use_stream / _use_own_stream / _profile / _kernel_ms do not
 * count), other pointers, strides or dims: the sums are dropped and
 * SSIMFast reads the images as always; shapes the one-pass kernel does not take run the plain blur.  The promise about
 * writes is the caller's: the library cannot see a store to device memory it was only lent.  Pays from two 4K images per call
 * up (B = 2: 5 %, 8: 21 %, 32: 23 % less time for the pair of calls); one image per call: no gain (profiles/r05_time_twocall_keep.txt). */
#define FNX_BLUR_KEEP_BOX_SUMS 2

typedef struct fnx_ctx fnx_ctx;
typedef struct fnx_prepared fnx_prepared;

/* ---- runtime ---------------------------------------------------------- */
const char *fnx_version(void);
/* Number of devices this library uses (0 when there is none / no driver / switched off).  Default: every HIP device.
 * Environment, read once: FENNEC_HIP_DEVICES="0,2,3" picks and orders HIP ordinals (device index i below is then the
 * i-th of them), FENNEC_HIP_DISABLE=1 leaves none -- fnx_ctx_create then returns FNX_ERR_NO_DEVICE, the status on
 * which the cgo shim runs the reference's own Go bodies ("force CPU or pick devices"). */
int fnx_device_count(void);
/* The same choice from code (wins over the environment): the n HIP ordinals to use, in order; n == 0: none (force
 * off); (NULL, -1): back to the environment's / every device.  Contexts that exist keep their device. */
int fnx_set_devices(const int *hip_ordinals, int n);
/* Thread-local text of the last error returned on this thread. */
const char *fnx_last_error(void);
int fnx_ctx_create(int device, fnx_ctx **out);
void fnx_ctx_destroy(fnx_ctx *ctx);
int fnx_ctx_device(const fnx_ctx *ctx);
/* The ctx's hipStream_t (as void*), for callers that time or order work. */
void *fnx_ctx_stream(fnx_ctx *ctx);
/* Launch on a stream of the CALLER's (a framework's current stream, the NULL stream included) instead of the ctx's
 * own: FNX_DEVICE work is then ordered with the caller's other work on that stream by construction -- inputs the
 * caller produced there are complete before the kernels read them, outputs are ready for whatever the caller enqueues
 * next, and a stream-ordered allocator may recycle the tensors safely -- with no cross-stream event per call (each such
 * dependency costs ~25 us of idle GPU between two back-to-back calls).  The switch itself orders everything the ctx has
 * already enqueued before the new stream's work.  The ctx does not own a lent stream: keep it alive until the ctx has
 * been switched away from it (or destroyed after a sync).  fnx_ctx_use_own_stream returns to the ctx's stream. */
int fnx_ctx_use_stream(fnx_ctx *ctx, void *hip_stream);
int fnx_ctx_use_own_stream(fnx_ctx *ctx);
/* Block until everything enqueued on the ctx has finished. */
int fnx_ctx_sync(fnx_ctx *ctx);
/* Diagnostics for roofline reporting: while enabled, the ctx brackets every launch of the selected
 * kernel classes with a pair of HIP events on the stream the kernel is launched on.  `enable` is a bit
 * mask of FNX_PROF_* (0: off; 1 = FNX_PROF_MAIN keeps its round-1 meaning: blur_direct_kernel of the
 * GaussianBlur fast / one-pass path and analyze_pass_kernel).  fnx_ctx_kernel_ms waits for the OLDEST
 * bracketed launch not read yet and returns its duration in milliseconds -- one call per launch, in
 * launch order; the last 32 launches are kept (FNX_ERR_INVALID if none is unread).  Calling
 * fnx_ctx_profile with a non-zero mask also forgets the unread ones. */
#define FNX_PROF_MAIN 1    /* blur_direct_kernel, analyze_pass_kernel */
#define FNX_PROF_SSIM 2    /* windowed_ssim_march_kernel */
#define FNX_PROF_RESIZE 4  /* resize kernels: one launch per lanczosResize (fused H + V) or two (resizeH, resizeV) */
#define FNX_PROF_FX 8      /* fx kernels: gaussianBlur3x3 / Sharpen / AdaptiveSharpen */
#define FNX_PROF_JPEG 16   /* jpeg_block_kernel (fdct, quantise, dequantise, idct of every block) */
int fnx_ctx_profile(fnx_ctx *ctx, int enable);
int fnx_ctx_kernel_ms(fnx_ctx *ctx, float *ms);
/* The kernel the ctx's LAST call of one class (one FNX_PROF_* bit: MAIN = GaussianBlur, SSIM = the windowed kernel,
 * RESIZE = lanczosResize) really launched, as a static string ("" before the first such call, NULL for a bad
 * argument): which of the routes the dispatch took (matrix pipe, direct, generic; fused, two-pass ...).  Reporting
 * only: bench.py names its roofline kernel with it.  No reference counterpart. */
const char *fnx_ctx_last_kernel(fnx_ctx *ctx, int prof_class);
/* Arithmetic of the windowed statistics of FULL-RESOLUTION SSIM (ssim.go:24-43, 110-148: planes of 4 M windows and more --
 * fnx_ssim, fnx_ssim_enqueue, fennec_SSIM on large images; SSIMFast / MSSSIM planes are <= 512 px and always exact):
 *   FNX_SSIM_EXACT (default)  fp64 moments, |result - reference| <= 1e-9 (measured ~1e-12)
 *   FNX_SSIM_FAST             fp32 moments of centred luminances in a cancellation-free form of the same expression,
 *                             |result - reference| <= 1e-6 (SURVEY.md Appendix A's tolerance for fp32-moment paths;
 *                             measured <= 3e-7), identical images still give exactly 1.  About 1.5 x the throughput.
 * The analogue of FNX_BLUR_FAST / FNX_BLUR_EXACT; a property of the ctx because the SSIM entry points keep the reference's
 * argument lists.  The Go shim leaves the default. */
#define FNX_SSIM_EXACT 0
#define FNX_SSIM_FAST 1
int fnx_ctx_set_ssim_mode(fnx_ctx *ctx, int mode);
/* Kernel-form selection of ONE ctx, for tests and A/B timing: every form computes the same bytes as the default, the
 * selection only says which kernel does it, so that a fallback the product takes for rare tables (the fp64
 * reference-order kernels, the two-pass twin of a fused launch ...) can be run on any image and compared.  name / value:
 *   "fx_stream" "0"        effects.hip's tile kernel instead of the streaming kernel
 *   "fx_pairs" "0"         the tile kernel's one-row form
 *   "fx_ref" "1"           the fp64 reference-order effects kernel
 *   "resize_mfma" "0|1|2"  matrix-pipe resize never / downscales (default) / wherever the tables allow (read when a plan is built)
 *   "resize_fp64" "1"      the fp64 reference-order resize kernels
 *   "resize_fused" "0"     the two-pass resize kernels
 *   "msssim_levelwise" "1", "msssim_nofuse0" "1", "msssim_fold" "0", "msssim_boxfly" "1"   MSSSIM's launch structure
 *   "palette_grid" "0|1"   applyPalette: whole-palette walk for every image / the candidate grid for every image
 * value NULL: back to the default.  Unknown name or a value longer than 7 characters: FNX_ERR_INVALID.  No reference counterpart. */
int fnx_ctx_set_form(fnx_ctx *ctx, const char *name, const char *value);

/* Device memory on the ctx's device (for FNX_DEVICE callers). */
int fnx_malloc(fnx_ctx *ctx, size_t bytes, void **dptr);
int fnx_free(fnx_ctx *ctx, void *dptr);
/* 2-D copies host<->device, stream-ordered; both return after completion. */
int fnx_upload(fnx_ctx *ctx, void *dptr, int dstride, const void *host, int hstride, int w, int h);
int fnx_download(fnx_ctx *ctx, void *host, int hstride, const void *dptr, int dstride, int w, int h);

/* ---- effects.go ------------------------------------------------------- */
/* GaussianBlur body (effects.go:167-219): separable, clamp-to-edge, RGB only,
 * uint8 intermediate, alpha from the source.  kernel[2*radius+1] is the
 * normalised 1-D kernel of effects.go:153-165.  The sigma<=0 guard
 * (effects.go:147-149) belongs to the caller. */
int fnx_gaussian_blur(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                      const double *kernel, int radius, int flags, uint8_t *dst, int dstride);
/* A source with sstride != 4*w is a Go SubImage to fnx_blur3x3 / fnx_sharpen / fnx_adaptive_sharpen (and to fnx_msssim):
 * the reference copies borders and alpha -- and MSSSIM its whole pyramid base -- from the FIRST 4*w*h bytes of the flat
 * Pix slice (effects.go:68,120; convert.go:16; ssim.go:345), not row by row, and so do these entry points.  A caller
 * that passes an ordinary pitched or cropped view and wants row-wise semantics must hand over a tight copy.  dst must
 * not alias src (the flat-copy pass rewrites dst after the filter). */
/* The fixed-point form the matrix-pipe kernel (csrc/blur_mfma.hip) gives a GaussianBlur table (effects.go:153-165), for
 * callers and tests that want to reason about it: wq[k] = round(kernel[k] * 2^24) with the centre tap taking what is left
 * of 2^24 (so sum wq == 2^24 exactly), and *err255 = 255 * max(P, N), P / N the sums of the positive / negative differences
 * wq[k] - kernel[k] * 2^24 -- the largest distance, in units of 2^-24, between the integer sum sum wq[k] * p[k] and the real
 * sum sum kernel[k] * p[k] * 2^24 over bytes 0 <= p <= 255.  A sample
 * whose (sum + 2^23) mod 2^24 lies at least ceil(*err255) + 2 away from 0 and from 2^24 rounds to the same byte as the
 * reference's clampF of its fp64 chain; the others are recomputed in fp64 (FNX_BLUR_EXACT).  Host arithmetic only, no
 * device.  FNX_NOOP: the table is not the matrix kernels' (radius outside 1..62, a negative or >= 0.49 weight, sum != 1). */
int fnx_blur_fixed_point(const double *kernel, int radius, long long *wq /* 2*radius+1 */, double *err255);
/* gaussianBlur3x3 (effects.go:116-141). */
int fnx_blur3x3(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                uint8_t *dst, int dstride);
/* Sharpen body (effects.go:24-44), amount = 1+1.5*strength; requires w,h >= 3. */
int fnx_sharpen(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                double amount, uint8_t *dst, int dstride);
/* AdaptiveSharpen body (effects.go:63-89) incl. localEdgeStrength
 * (effects.go:93-112), amount = 1+2*strength; requires w,h >= 3. */
int fnx_adaptive_sharpen(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                         double amount, uint8_t *dst, int dstride);
/* Sharpen / AdaptiveSharpen bodies of n same-geometry DEVICE images (srcs / dsts: host arrays of device pointers, dst != src):
 * the bytes of n single calls, one launch of the streaming kernel for tight images (stride == 4 w; the image is its second
 * grid dimension), image by image otherwise.  Frames below 8K do not fill the machine alone (a 1080p image is 2 000 waves of
 * 12 rows each); enqueued, no wait. */
int fnx_sharpen_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h, double amount,
                      uint8_t *const *dsts, int dstride);
int fnx_adaptive_sharpen_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h, double amount,
                               uint8_t *const *dsts, int dstride);

/* ---- resize.go -------------------------------------------------------- */
/* Lanczos tap table in CSR form: taps of output d are
 * index[offset[d] .. offset[d+1]) with weight[...] (precomputeWeights,
 * resize.go:164-197).  offset has dst+1 entries. */
/* resizeH (resize.go:77-118): src srcW x srcH -> dst dstW x srcH. */
int fnx_resize_h(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW, int srcH,
                 const int32_t *offset, const int32_t *index, const double *weight,
                 uint8_t *dst, int dstride, int dstW);
/* resizeV (resize.go:121-161): src srcW x srcH -> dst srcW x dstH. */
int fnx_resize_v(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW, int srcH,
                 const int32_t *offset, const int32_t *index, const double *weight,
                 uint8_t *dst, int dstride, int dstH);
/* lanczosResize (resize.go:37-53): guards, equal-dims flat copy, H then V
 * through a uint8 intermediate.  Tables may be NULL when dims are equal. */
int fnx_lanczos_resize(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW,
                       int srcH, const int32_t *offH, const int32_t *idxH, const double *wH,
                       const int32_t *offV, const int32_t *idxV, const double *wV,
                       uint8_t *dst, int dstride, int dstW, int dstH);
/* lanczosResize of n same-geometry DEVICE images (srcs / dsts: host arrays of device pointers) with one tap-table pair: the
 * bytes of n fnx_lanczos_resize calls, enqueued as ONE set of launches where the one-launch kernels apply (a 4K resize
 * alone is ~700 workgroups on 768-1024 resident slots: one under-filled round; a batch fills the machine and pays the
 * launch, the plan look-up and the tails once).  Other tables (windows wider than the fused tile, tables outside the
 * rounding guard) run image by image.  CompressBatch workers that resize many same-sized frames (fennec.go:127-129). */
int fnx_lanczos_resize_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int srcW, int srcH,
                             const int32_t *offH, const int32_t *idxH, const double *wH,
                             const int32_t *offV, const int32_t *idxV, const double *wV,
                             uint8_t *const *dsts, int dstride, int dstW, int dstH);

/* ---- ssim.go ---------------------------------------------------------- */
/* boxDownsample (ssim.go:244-309). */
int fnx_box_downsample(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW,
                       int srcH, uint8_t *dst, int dstride, int dstW, int dstH);
/* SSIMFast (ssim.go:48-70); window = gaussianKernel(8,1.5) (ssim.go:223-241),
 * 64 doubles.  Both images are w x h (the reference does not check). */
int fnx_ssim_fast(fnx_ctx *ctx, int space, const uint8_t *a, int astride, const uint8_t *b,
                  int bstride, int w, int h, const double *window, double *out);
/* pixelSSIM (ssim.go:169-204), the branch SSIM / SSIMFast take when w < 8 || h < 8, with the slices' REAL
 * lengths: the reference walks `for i := 0; i < len(a.Pix); i += 4` over both flat Pix slices, and for a
 * SubImage len(Pix) runs to the end of the PARENT's buffer (image.NRGBA.SubImage: Pix = p.Pix[i:]), not just to
 * the sub-image's last pixel.  fnx_ssim / fnx_ssim_fast only know (h-1)*stride + 4w -- the shortest slice a
 * w x h image can have, which is what image.NewNRGBA and every image in this path's own pipeline has; a caller
 * holding a true SubImage under 8 px passes len(a.Pix), len(b.Pix) here.  b_pix_len < a_pix_len is
 * FNX_ERR_INVALID (the reference panics: index out of range).  a_pix_len is rounded up to a multiple of 4 the way
 * the loop's last iteration reads it (and must then fit).  Divides by w*h like the reference; w*h == 0 -> 1.0. */
int fnx_pixel_ssim(fnx_ctx *ctx, int space, const uint8_t *a_pix, size_t a_pix_len, const uint8_t *b_pix,
                   size_t b_pix_len, int w, int h, double *out);
/* SSIM (ssim.go:24-43) for equal dims (the dims-differ resize is composed by
 * the caller, as fennec_SSIM does). */
int fnx_ssim(fnx_ctx *ctx, int space, const uint8_t *a, int astride, const uint8_t *b,
             int bstride, int w, int h, const double *window, double *out);
/* MSSSIM (ssim.go:313-365) for equal dims; per_level (NULL or 5 doubles)
 * receives each level's SSIMFast, NaN where the reference stops early.  Strides are NOT honoured for the pyramid base:
 * toNRGBA copies the first 4*w*h flat bytes (convert.go:16, ssim.go:345) -- see the SubImage note at fnx_blur3x3. */
int fnx_msssim(fnx_ctx *ctx, int space, const uint8_t *a, int astride, const uint8_t *b,
               int bstride, int w, int h, const double *window, double *out, double *per_level);

/* Binary-search form of SSIMFast (compress.go:45-74 calls SSIMFast(src, decoded)
 * with the same src every iteration): downsample + luminance of the reference
 * side once, then compare candidates against it. */
int fnx_ssim_fast_prepare(fnx_ctx *ctx, int space, const uint8_t *a, int astride, int w, int h,
                          fnx_prepared **out);
int fnx_ssim_fast_against(fnx_ctx *ctx, const fnx_prepared *ref, int space, const uint8_t *b,
                          int bstride, const double *window, double *out);
void fnx_prepared_free(fnx_ctx *ctx, fnx_prepared *p);

/* SURVEY 8(f)2, first slice -- the JPEG quantisation round trip on the device.  dst = toNRGBARef(jpeg.Decode(
 * jpeg.Encode(src, Options{Quality: quality}))) as far as the PIXELS go (compress.go:50-58 via io.go:157-169): baseline
 * 4:2:0, Go's colour equations, integer FDCT / IDCT and quantiser scaling, no entropy coding (it is lossless).  The
 * arithmetic restates Go's standard library, which is not part of the reference tree: parity with Go is unpinned twice
 * over; against the oracle's restatement it is bit-exact. */
int fnx_jpeg_roundtrip(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, int quality,
                       uint8_t *dst, int dstride);
/* compressJPEGOptimal's binary search (compress.go:21-74) with every candidate quality round-tripped and scored on the
 * device: *quality = lowest quality whose SSIMFast(src, round trip) >= target_ssim (the reference's lower bounds by
 * target, target >= 1 -> 0.999), *ssim its score, *steps the candidates tried.  FNX_NOOP: none reached the target
 * (*quality = 100, *ssim = 1.0, as the reference's fallback).  The caller encodes ONCE, at *quality, with the real codec. */
/* jpeg.Encode(src, &jpeg.Options{Quality: quality}) (io.go:157-169) on the device: the complete file -- baseline, 4:2:0,
 * the typical Huffman tables, writer.go's segment order -- into host memory `out` (capacity cap); *nbytes = its size.
 * FNX_ERR_INVALID with *nbytes set when cap is too small (call again); out == NULL with cap == 0 asks for the size
 * only (what targetsize.go's searches look at) and copies nothing.  Decoding the file gives exactly
 * fnx_jpeg_roundtrip's pixels (the entropy coder is lossless).  Restated from ITU T.81 and Go's file layout, not from
 * Go's source: byte parity with jpeg.Encode is unpinned; libjpeg-turbo decodes the files. */
int fnx_jpeg_encode(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, int quality, uint8_t *out, size_t cap,
                    size_t *nbytes);
/* jpegQualitySearchOpt (targetsize.go:125-176): the HIGHEST quality whose file fits target_bytes -- the bisection over
 * [lo, hi] the reference picks from the bits per pixel, every candidate's size from the device's entropy coder
 * (nothing is copied for a candidate), then the winner's file into `out` and, unless skip_ssim, its SSIMFast against the
 * source (the reference's computeSSIMNRGBA of the decoded winner).  FNX_NOOP: no quality fits (bestBuf == nil).
 * FNX_ERR_INVALID with *nbytes set when cap is too small. */
int fnx_jpeg_size_search(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, long long target_bytes, int skip_ssim,
                         const double *window /* 64; may be NULL with skip_ssim */, uint8_t *out, size_t cap, size_t *nbytes,
                         int *quality, double *ssim, int *steps /* may be NULL */);
/* compressJPEGOptimal (compress.go:21-87) for one image in one call: the quality search of fnx_jpeg_quality_search, then
 * the file at the quality it found (100 when nothing reached the target) as fnx_jpeg_encode writes it -- one upload of
 * the source, no host codec.  *ssim is the winning candidate's SSIMFast (1.0 when none won, as the reference reports). */
int fnx_jpeg_compress(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, double target_ssim,
                      const double *window /* 64 */, uint8_t *out, size_t cap, size_t *nbytes, int *quality, double *ssim,
                      int *steps /* may be NULL */);
int fnx_jpeg_quality_search(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, double target_ssim,
                            const double *window /* 64 */, int *quality, double *ssim, int *steps);
/* SURVEY 8(f)2, third slice -- image.Decode of a JPEG source (batch.go:88-101 via io.go:60-95) on the device:
 * dst = toNRGBARef(jpeg.Decode(data)), *w x *h.  `data` is HOST memory (the file); dst is in `space`.  dst == NULL:
 * only the dimensions (jpeg.DecodeConfig) -- and whether the device decoder takes the file at all (host work: ctx may be
 * NULL and no device is touched).  Handled: 8 bit, three components (4:4:4, 4:2:2, 4:2:0, 4:4:0, 4:1:1, 4:1:0), one
 * (image.Gray) or four (image.CMYK: Adobe CMYK / YCbCrK with every component 1 x 1, reader.go applyBlack), with or without restart
 * intervals: baseline (SOF0), extended sequential (SOF1) and progressive (SOF2) frames in
 * any number of scans; anything else (12 bit, arithmetic coding, lossless) returns
 * FNX_ERR_UNSUPPORTED and the caller decodes on the host (an explicit answer, not a fallback inside the library).
 * FNX_ERR_INVALID: a scan that ends early or holds a code outside its Huffman table.  Baseline: Huffman decoding is parallel
 * over 1024-bit spans of the scan that synchronise with their neighbours (jpeg_dec.hip); the result does not depend on how
 * many rounds that takes.  Progressive (r5): a refinement scan's bits depend on the coefficients of the scans before it, so
 * the scans are entropy-decoded on the host (jpeg_prog.cpp) and the coefficients go up at 2 bytes each -- as are the sequential
 * files that decoder has no form for (SOF1, components in scans of their own, a table that assigns the all-ones code); dequantisation, IDCT
 * and colour conversion are the device's as for baseline.  Restated from ITU T.81 and Go's documented behaviour (scan.go's
 * refine / reconstructProgressiveImage): bit-exact against the tests' CPU restatement, parity with Go unpinned.
 * The host route sizes 136 bytes of host memory per block from the frame header alone, so it takes at most
 * FNX_JPEG_HOST_MAX_BLOCKS blocks (16K x 8K at 4:2:0); a header that promises more is FNX_ERR_UNSUPPORTED (the host codec's). */
#define FNX_JPEG_HOST_MAX_BLOCKS (1 << 22)
int fnx_jpeg_decode(fnx_ctx *ctx, const uint8_t *data, size_t n, int space, uint8_t *dst, int dstride, int *w, int *h);
/* Host only (no ctx, no device): what fnx_jpeg_decode hands the device for a progressive (SOF2) file -- the quantised
 * coefficients over all its scans (scan.go processSOS / refine as published).  coef: [*blocks][64] int16, blocks MCU by MCU
 * in the order of an interleaved scan (Y blocks of the MCU row by row, then Cb, then Cr), natural (row-major) order inside a
 * block, DC as it is.  *blocks, *w, *h, *ratio (image.YCbCrSubsampleRatio 0..5; -1: one component; -2: four) are set whenever the
 * frame parses; coef == NULL or cap_blocks < *blocks: nothing is decoded (FNX_OK / FNX_ERR_INVALID).  A baseline file:
 * FNX_ERR_UNSUPPORTED (its scan is decoded on the device).  Exists for the CPU-side parity tests and the sanitizer runs. */
int fnx_jpeg_progressive_coefficients(const uint8_t *data, size_t n, int16_t *coef, size_t cap_blocks, size_t *blocks, int *w, int *h,
                                      int *ratio);
/* CompressBatch's per-item body for a JPEG source in one call (batch.go:88-122 -> compress.go:21-87): decode `data`
 * on the device, run compressJPEGOptimal's quality search there and write the winner's file into `out` -- the file
 * bytes go up, the new file's bytes come down, no host codec.  Arguments as fnx_jpeg_compress; *w, *h: the image's
 * dimensions.  FNX_ERR_UNSUPPORTED as fnx_jpeg_decode. */
int fnx_jpeg_recompress(fnx_ctx *ctx, const uint8_t *data, size_t n, double target_ssim, const double *window /* 64 */, uint8_t *out,
                        size_t cap, size_t *nbytes, int *quality, double *ssim, int *steps /* may be NULL */, int *w, int *h);

/* ---- convert.go / exif.go --------------------------------------------- */
/* ApplyOrientation (exif.go:178-203 over convert.go:186-256).  orient 2..8;
 * dst is w x h for 2,3,4 and h x w for 5,6,7,8.  0, 1 and unknown values
 * return FNX_NOOP. */
int fnx_orient(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, int orient,
               uint8_t *dst, int dstride);

/* ---- batched forms (FNX_DEVICE only) ------------------------------------ */
/* n independent images of identical geometry in ONE launch per stage
 * (CompressBatch items never interact, batch.go:88-122).  srcs/dsts/as/bs are
 * HOST arrays of n device pointers.  Image outputs are bit-identical to the per-image calls; SSIM
 * scalars may differ from them in the last bits (the window statistics are tiled differently when
 * there are many windows in flight) -- both stay within the 1e-9 bar and are run-to-run reproducible. */
int fnx_gaussian_blur_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w,
                            int h, const double *kernel, int radius, int flags,
                            uint8_t *const *dsts, int dstride);
int fnx_ssim_fast_batch(fnx_ctx *ctx, int n, const uint8_t *const *as, int astride,
                        const uint8_t *const *bs, int bstride, int w, int h,
                        const double *window, double *out /* n, host */);
/* The same split in two, so that one host thread can keep several contexts (= streams) busy, or keep
 * ONE stream from draining between batches: _enqueue only queues the kernels; the n results land in
 * pinned host memory and wait in the ctx's FIFO (up to 4 batches) until fnx_results_fetch takes the
 * OLDEST one (it waits for that batch only, not for work queued behind it).  Enqueueing batch s+1 before
 * fetching batch s is the intended use; the kernels of a ctx still run in order.  The blocking forms
 * require an empty FIFO. */
int fnx_ssim_fast_batch_enqueue(fnx_ctx *ctx, int n, const uint8_t *const *as, int astride,
                                const uint8_t *const *bs, int bstride, int w, int h,
                                const double *window);
int fnx_results_fetch(fnx_ctx *ctx, int n, double *out /* n, host */);
/* SSIM (ssim.go:24-43, equal dims) of ONE device-resident pair, enqueued like the batches above: the value joins the
 * ctx's FIFO and fnx_results_fetch(ctx, 1, &v) returns it later -- a worker that scores a stream of large images
 * (config 4: AdaptiveSharpen + SSIM per 8K image) keeps the next image's kernels queued while this one's result
 * crosses to the host instead of draining the stream at every call. */
int fnx_ssim_enqueue(fnx_ctx *ctx, const uint8_t *a, int astride, const uint8_t *b, int bstride, int w, int h,
                     const double *window /* 64 */);
/* SSIM of n device-resident pairs of ONE geometry: one launch of the window kernel with the image as a grid dimension,
 * one FIFO entry of n values (fnx_results_fetch(ctx, n, v)).  For frames smaller than 8K, where one pair does not fill the
 * machine (a 1080p pair is 500 waves).  Honours fnx_ctx_set_ssim_mode like fnx_ssim_enqueue. */
int fnx_ssim_batch_enqueue(fnx_ctx *ctx, int n, const uint8_t *const *as, int astride, const uint8_t *const *bs, int bstride,
                           int w, int h, const double *window /* 64 */);
/* MSSSIM (ssim.go:313-365, equal dims) of ONE device-resident pair, enqueued the same way: the per-level values wait in
 * the FIFO and fnx_results_fetch(ctx, 1, &v) combines them (exp of the weighted log sum, on the host as in fnx_msssim). */
int fnx_msssim_enqueue(fnx_ctx *ctx, const uint8_t *a, int astride, const uint8_t *b, int bstride, int w, int h,
                       const double *window /* 64 */);
/* n such pairs of ONE geometry as one FIFO entry: fnx_results_fetch(ctx, n, v) returns the n values in order (any k <= n of
 * them; the entry is consumed either way).  What a worker scoring a batch of frames calls instead of n enqueues: one FIFO
 * slot, one result event, and -- through fennec_MSSSIM_batch_enqueue -- ONE batched resize for the pairs whose dims differ. */
int fnx_msssim_batch_enqueue(fnx_ctx *ctx, int n, const uint8_t *const *as, int astride, const uint8_t *const *bs, int bstride,
                             int w, int h, const double *window /* 64 */);

/* dsts[i] = GaussianBlur(srcs[i]) AND out[i] = SSIMFast(srcs[i], dsts[i]) -- the pair of calls
 * the reference makes whenever it scores a processed image against its source (effects.go:146
 * then ssim.go:48; the headline benchmark's step) -- with ONE pass over the pixels: the blur
 * kernel already holds every source and every blurred pixel, so it also accumulates the integer
 * channel sums of boxDownsample (ssim.go:244-309) for both, and SSIMFast never re-reads either
 * full-size image (HBM traffic 2*S instead of 4*S).  Results are identical to the two separate
 * calls: the box sums are integers, everything after them is the same code.  Shapes the one-pass
 * kernel is not built for or does not win on (radius > 24, no downsample, a box-downsample ratio below
 * 3.6 or boxes above 256 px: long side under ~1850 px or over 8192 px; with FNX_BLUR_EXACT also a kernel
 * with negative taps or gain > 1) run the two ops back to back.  With FNX_BLUR_EXACT the blurred images
 * are bit-exact and the scores are computed from exactly those images.  The _enqueue form pairs with fnx_results_fetch. */
/* The same for ONE image in either space: dst = GaussianBlur(src) and *ssim = SSIMFast(src, dst) -- bytes and score those of
 * fnx_gaussian_blur followed by fnx_ssim_fast.  With FNX_HOST the image crosses PCIe ONCE each way (source up, blurred
 * image down: 2 x 33 MB at 4K); the two separate calls upload the source twice and the blurred image once more
 * (4 x 33 MB), and PCIe is all a host-space call costs (0.6 ms per 33 MB against 40 us of kernels).  What the cgo
 * shim's GaussianBlurScored calls.  Blocking (the two kernels back to back on the ctx's stream; batches of device images
 * take fnx_gaussian_blur_ssim_fast_batch, whose one-pass kernel also halves the HBM traffic). */
int fnx_gaussian_blur_ssim_fast(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                                const double *kernel, int radius, int flags, uint8_t *dst, int dstride,
                                const double *window /* 64 */, double *ssim);
int fnx_gaussian_blur_ssim_fast_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride,
                                      int w, int h, const double *kernel, int radius, int flags,
                                      uint8_t *const *dsts, int dstride, const double *window,
                                      double *out /* n, host */);
int fnx_gaussian_blur_ssim_fast_batch_enqueue(fnx_ctx *ctx, int n, const uint8_t *const *srcs,
                                              int sstride, int w, int h, const double *kernel,
                                              int radius, int flags, uint8_t *const *dsts,
                                              int dstride, const double *window);

/* ---- Analyze (analyze.go:26-124), SURVEY 8(f) item 3 ---------------------- */
/* What the device computes: every statistic that needs the pixels.  The float
 * epilogue (computeEntropy, sqrt, the recommend* rules, analyze.go:87-230) is a
 * few hundred flops on these numbers and stays with the caller -- Go computes it
 * with its own math.Log2, fennec_Analyze below with libm. */
typedef struct fnx_analysis {
    uint64_t histogram[256]; /* luminance histogram, bin int(lum + 0.5), all pixels: exact */
    double bright_sum;       /* sum of luminance over all pixels (summation order differs from
                                the reference's serial loop: <= 1e-12 relative) */
    double variance_sum;     /* sum (lum - mean)^2 over the <=100x100 contrast grid (same remark) */
    int64_t sample_count;    /* points of that grid */
    int64_t edge_count;      /* Sobel magnitude > 30 on the sampled grid: exact */
    int64_t edge_total;
    int32_t unique_colors;   /* len(colorSet): distinct colours among every step-th pixel, capped 1024 */
    int32_t has_alpha;       /* some alpha < 255 */
    int32_t is_grayscale;    /* every pixel r == g == b */
    int32_t pad;
} fnx_analysis;
int fnx_analyze(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                fnx_analysis *out /* host */);
/* n same-sized device images (HOST array of device pointers), one launch per stage. */
int fnx_analyze_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h,
                      fnx_analysis *out /* n, host */);
/* isOpaque / isGrayscale (convert.go:66-84): both walk the FLAT Pix slice (pix_len bytes, row
 * padding included), one pass answers both. */
int fnx_scan_flags(fnx_ctx *ctx, int space, const uint8_t *pix, size_t pix_len, int *is_opaque,
                   int *is_grayscale);

/* ---- toNRGBARef of a decoded JPEG (convert.go:22-64), SURVEY 8(f) item 1 ------------------- */
/* Go's image/jpeg returns *image.YCbCr (or *image.Gray): the reference converts it with a
 * per-pixel At().RGBA() loop on the host (convertToNRGBA, convert.go:34-64) and SSIMFast then
 * uploads 4 bytes per pixel.  These take the decoder's planes as they are (Rect.Min == (0,0)):
 * Y: w x h, stride ystride; Cb, Cr: the subsampled planes of image.NewYCbCr, stride cstride;
 * ratio = image.YCbCrSubsampleRatio (0 4:4:4, 1 4:2:2, 2 4:2:0, 3 4:4:0, 4 4:1:1, 5 4:1:0);
 * cb == cr == NULL: image.Gray.  Arithmetic: image.YCbCr.COffset + color.YCbCr.RGBA() of the Go
 * standard library (go.mod:3 pins go 1.25.5), then convert.go:48-53's uint8(c >> 8) -- integer,
 * bit-exact against the restatement in oracle/ (which states its provenance). */
int fnx_ycbcr_to_nrgba(fnx_ctx *ctx, int space, const uint8_t *y, int ystride, const uint8_t *cb,
                       const uint8_t *cr, int cstride, int ratio, int w, int h, uint8_t *dst,
                       int dstride);
/* SSIMFast(prepared reference, toNRGBARef(decoded planes)) for the quality binary search
 * (compress.go:45-74): 1.5 bytes per pixel cross PCIe at 4:2:0 instead of 4. */
int fnx_ssim_fast_against_ycbcr(fnx_ctx *ctx, const fnx_prepared *ref, int space, const uint8_t *y,
                                int ystride, const uint8_t *cb, const uint8_t *cr, int cstride,
                                int ratio, const double *window, double *out);

/* ---- applyPalette + palettedToNRGBA (targetsize.go:488-546), SURVEY 8(f) item 4 ---------- */
/* Nearest palette entry per pixel by squared RGB distance, first minimum wins (targetsize.go:
 * 505-517); `indices` = image.Paletted.Pix (w x h bytes, stride istride), `quantized` =
 * palettedToNRGBA of it (NRGBA, alpha 255).  Either output may be NULL.  palette: ncolors x 4
 * HOST bytes r,g,b,a with a == 255 (what medianCut emits; medianCut itself stays with the
 * caller).  Integer arithmetic: bit-exact. */
int fnx_apply_palette(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                      const uint8_t *palette, int ncolors, uint8_t *indices, int istride,
                      uint8_t *quantized, int qstride);

/* ======================================================================= */
/* fennec_* : the reference's function set (names and argument meaning as in
 * the Go source), mirrored above fnx_*.                                     */
/* ======================================================================= */

/* ImageStats (analyze.go:9-22); Format: 1 JPEG, 2 PNG (types.go:35-42); Quality: 0 Balanced,
 * 3 High, 4 Aggressive (types.go:59-72). */
typedef struct fennec_ImageStats {
    int32_t Width, Height;
    int32_t HasAlpha, IsGrayscale, UniqueColors;
    int32_t RecommendedFormat, RecommendedQuality, pad;
    double Entropy, EdgeDensity, MeanBrightness, Contrast, EstimatedCompression;
} fennec_ImageStats;
/* Analyze (analyze.go:26-124). */
int fennec_Analyze(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                   fennec_ImageStats *out);
/* The float epilogue alone: computeEntropy, contrast, edge density, recommendFormat,
 * recommendQuality, estimateCompression (analyze.go:87-230) from the device's numbers. */
void fennec_statsFromAnalysis(const fnx_analysis *a, int w, int h, fennec_ImageStats *out);
/* isOpaque / isGrayscale (convert.go:66-84) of an image (flat Pix: (h-1)*stride + 4*w bytes). */
int fennec_isOpaque(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, int *out);
int fennec_isGrayscale(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, int *out);

/* gaussianKernel(size, sigma) (ssim.go:223-241): size*size doubles. */
void fennec_gaussianKernel(int size, double sigma, double *kernel);
/* GaussianBlur's radius/kernel (effects.go:153-165); kernel may be NULL. */
int fennec_blurKernel(double sigma, double *kernel);
/* lanczosKernel (resize.go:57-69). */
double fennec_lanczosKernel(double x);
/* precomputeWeights (resize.go:164-197) with ratio/support derived as
 * resizeH/resizeV do; returns the tap count; index/weight may be NULL to size. */
int fennec_precomputeWeights(int dstSize, int srcSize, int32_t *offset, int32_t *index,
                             double *weight);
/* smartResize's dims (resize.go:12-32): returns 0 if the image already fits (negative: a NULL out pointer). */
int fennec_smartResizeDims(int srcW, int srcH, int maxW, int maxH, int *dstW, int *dstH);
/* SSIMFast's downsample dims (ssim.go:52-56): returns 1 if it downsamples (negative: a NULL out pointer). */
int fennec_ssimFastDims(int w, int h, int *newW, int *newH);

int fennec_SSIM(fnx_ctx *ctx, int space, const uint8_t *a, int astride, int aw, int ah,
                const uint8_t *b, int bstride, int bw, int bh, double *out);     /* ssim.go:24 */
int fennec_SSIMFast(fnx_ctx *ctx, int space, const uint8_t *a, int astride, const uint8_t *b,
                    int bstride, int w, int h, double *out);                      /* ssim.go:48 */
int fennec_MSSSIM(fnx_ctx *ctx, int space, const uint8_t *a, int astride, int aw, int ah,
                  const uint8_t *b, int bstride, int bw, int bh, double *out);   /* ssim.go:313 */
/* MSSSIM of a device-resident pair through the ctx's result FIFO (fnx_msssim_enqueue), img2 resized to img1's
 * dims first when they differ (ssim.go:320-322); fnx_results_fetch(ctx, 1, &v) returns the value. */
int fennec_MSSSIM_enqueue(fnx_ctx *ctx, const uint8_t *a, int astride, int aw, int ah, const uint8_t *b, int bstride,
                          int bw, int bh);
/* ... of n device pairs (every a of aw x ah, every b of bw x bh): the b's are resized to the a's dims by ONE batched
 * lanczosResize (fnx_lanczos_resize_batch) when the dims differ, ssim.go:320-322; results as fnx_msssim_batch_enqueue's. */
int fennec_MSSSIM_batch_enqueue(fnx_ctx *ctx, int n, const uint8_t *const *as, int astride, int aw, int ah,
                                const uint8_t *const *bs, int bstride, int bw, int bh);
int fennec_GaussianBlur(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                        double sigma, uint8_t *dst, int dstride);                 /* effects.go:146;
                        FNX_HOST: FNX_BLUR_EXACT (bit-exact; the call is PCIe-bound anyway),
                        FNX_DEVICE: FNX_BLUR_FAST */
int fennec_Sharpen(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                   double strength, uint8_t *dst, int dstride);                   /* effects.go:10 */
int fennec_AdaptiveSharpen(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h,
                           double strength, uint8_t *dst, int dstride);           /* effects.go:49 */
int fennec_ApplyOrientation(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w,
                            int h, int orient, uint8_t *dst, int dstride);        /* exif.go:178 */
int fennec_lanczosResize(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW,
                         int srcH, uint8_t *dst, int dstride, int dstW, int dstH); /* resize.go:37 */
int fennec_lanczosResizeBatch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int srcW, int srcH,
                              uint8_t *const *dsts, int dstride, int dstW, int dstH);   /* resize.go:37 x n (device images) */
int fennec_boxDownsample(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW,
                         int srcH, uint8_t *dst, int dstride, int dstW, int dstH); /* ssim.go:244 */

/* CompressBatch (batch.go:58-128) over decoded NRGBA items with compressJPEGOptimal (compress.go:21-87) as the per-item
 * work, all of it on the device (fnx_jpeg_compress): `workers` threads, ONE closed queue of indices (an atomic counter:
 * the channel of batch.go:72-81), one fnx_ctx per worker on `device`, results stored by index, `on_item(completed,
 * total, user)` serialised as OnItem is, `*cancel != 0` checked before each item (ctx.Done(), batch.go:90-99).  Item i's
 * file goes to outs[i] (capacity caps[i]); a result with failed != 0 carries the FNX_* status in `status`.
 * Returns FNX_OK when the pool ran (per-item failures are in the results), an error when no worker could start. */
typedef struct fennec_BatchResult {
    int32_t index, failed, has_result, quality, steps, status;
    int64_t original_size, compressed_size;
    double ssim;
    int32_t device;      /* index (in this library's device list) of the device whose worker took the item; -1: nobody did */
    int32_t reserved;
} fennec_BatchResult;
typedef void (*fennec_on_item)(int completed, int total, void *user);
int fennec_CompressBatchNRGBA(int device, int workers, int n, int space, const uint8_t *const *srcs, const int *strides,
                              const int *widths, const int *heights, const int64_t *original_sizes, double target_ssim,
                              uint8_t *const *outs, const size_t *caps, fennec_BatchResult *results,
                              const volatile int *cancel /* may be NULL */, fennec_on_item on_item /* may be NULL */, void *user);
/* The same pool over the NODE's devices (batch.go:63-126 with g GPUs): `workers` threads (0: NumCPU, capped at
 * n), worker i bound to a context on devices[i mod ndev], ONE queue of indices for all of them, results by index with the
 * device that served each item.  A device may be listed more than once (more workers' contexts on it).  Device-resident
 * items (space FNX_DEVICE) live on one device, so a list of several distinct devices takes host-space items only. */
int fennec_CompressBatchNRGBADevices(const int *devices, int ndev, int workers, int n, int space, const uint8_t *const *srcs,
                                     const int *strides, const int *widths, const int *heights, const int64_t *original_sizes,
                                     double target_ssim, uint8_t *const *outs, const size_t *caps, fennec_BatchResult *results,
                                     const volatile int *cancel /* may be NULL */, fennec_on_item on_item /* may be NULL */, void *user);
/* CompressFile for a JPEG source in standard mode (fennec.go:30-76 -> compressImageInternal :107-141 -> handleStandardMode
 * :162-205) from the file's bytes with every pixel stage on the device: image.Decode + toNRGBA (fnx_jpeg_decode),
 * ApplyOrientation when opts->orient is 2..8 (Options.AutoOrient; the caller reads the tag as exif.go does),
 * smartResize when max_w or max_h > 0 (Options.MaxWidth / MaxHeight), analyzeFormat when auto_format (Format: Auto),
 * compressJPEGOptimal at target_ssim.  dims = {OriginalDimensions (after orientation), FinalDimensions}.
 * FNX_NOOP: analyzeFormat chose PNG (fewer than 256 sampled colours) -- the caller's compressPNG takes the item, dims are
 * set.  FNX_ERR_UNSUPPORTED as fnx_jpeg_decode.  Target-size mode stays the caller's (fnx_jpeg_size_search is its JPEG leg). */
typedef struct fennec_FileOptions {
    int32_t orient;      /* EXIF orientation 1..8; <= 1: none */
    int32_t max_w, max_h;
    int32_t auto_format; /* != 0: Format Auto */
    double target_ssim;  /* Options.Quality.targetSSIM() or Options.TargetSSIM */
} fennec_FileOptions;
int fennec_CompressFileJPEG(fnx_ctx *ctx, const uint8_t *data, size_t n, const fennec_FileOptions *opts, uint8_t *out, size_t cap,
                            size_t *nbytes, int *quality, double *ssim, int *steps /* may be NULL */, int dims[4]);
/* The same pool over JPEG FILES in host memory (what CompressBatch reads for a .jpg item, batch.go:88-101): per item
 * fnx_jpeg_recompress -- decoder, search and encoder on the device, no host codec.  A file the device decoder does not
 * take comes back with failed != 0 and status == FNX_ERR_UNSUPPORTED: the caller decodes it on the host and sends it
 * through fennec_CompressBatchNRGBA.  original_size = sizes[i]. */
int fennec_CompressBatchJPEG(int device, int workers, int n, const uint8_t *const *files, const size_t *sizes, double target_ssim,
                             uint8_t *const *outs, const size_t *caps, fennec_BatchResult *results,
                             const volatile int *cancel /* may be NULL */, fennec_on_item on_item /* may be NULL */, void *user);
/* ... over a device list, as fennec_CompressBatchNRGBADevices: CompressBatch of .jpg items on all GPUs of the node from one
 * process (file bytes are host memory, so any device may take any item). */
int fennec_CompressBatchJPEGDevices(const int *devices, int ndev, int workers, int n, const uint8_t *const *files, const size_t *sizes,
                                    double target_ssim, uint8_t *const *outs, const size_t *caps, fennec_BatchResult *results,
                                    const volatile int *cancel /* may be NULL */, fennec_on_item on_item /* may be NULL */, void *user);
/* ... with CompressFile's options: per item fennec_CompressFileJPEG under item_opts[i] when that is not NULL, else under
 * *default_opts (batch.go:101-105: item.Opts over BatchOptions.DefaultOpts).  dims (may be NULL): 4 ints per item, as
 * fennec_CompressFileJPEG's.  An item analyzeFormat sends to PNG comes back failed with status FNX_NOOP. */
int fennec_CompressBatchJPEGOpts(int device, int workers, int n, const uint8_t *const *files, const size_t *sizes,
                                 const fennec_FileOptions *default_opts, const fennec_FileOptions *const *item_opts /* may be NULL */,
                                 uint8_t *const *outs, const size_t *caps, fennec_BatchResult *results, int *dims /* may be NULL */,
                                 const volatile int *cancel /* may be NULL */, fennec_on_item on_item /* may be NULL */, void *user);
/* The pool's idle worker contexts (kept between batches, per device) are destroyed. */
void fennec_pool_release(void);
/* Summarize (batch.go:140-158) of such results: out4 = {Total, Succeeded, Failed, TotalSaved}; returns AvgSSIM. */
double fennec_SummarizeResults(int n, const fennec_BatchResult *results, int64_t out4[4]);

/* Summarize (batch.go:140-158) over parallel arrays; out4 = {Total, Succeeded,
 * Failed, TotalSaved}; returns AvgSSIM.  Pure host arithmetic in index order. */
double fennec_Summarize(int n, const int32_t *failed, const int32_t *has_result,
                        const int64_t *original_size, const int64_t *compressed_size,
                        const double *ssim, int64_t out4[4]);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FENNEC_HIP_H */
