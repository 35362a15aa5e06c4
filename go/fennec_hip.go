//go:build cgo && fennec_hip

// Package fennec -- cgo shim that swaps the BODIES of the per-pixel hot-path functions
// (ssim.go, resize.go, effects.go, exif.go) for calls into libfennec_hip.so (MI355X / gfx950).
//
// Drop this file into the reference tree next to the files it shadows and build with
//
//	CGO_CFLAGS="-I$FENNEC_HIP/include" CGO_LDFLAGS="-L$FENNEC_HIP/fennec_amd -lfennec_hip" \
//	    go build -tags fennec_hip ./...
//
// after renaming each shadowed function in the reference to its `...Go` twin (one-line edits,
// listed in INTEGRATION.md) and giving the originals a `//go:build !fennec_hip` twin file.
// Everything above these functions -- CompressFile/CompressBytes/CompressBatch, the JPEG
// quality search, target-size mode, the CLI -- is untouched.
//
// NOTE: there is no Go toolchain in the image this repository is developed in, so this file
// has never been compiled.  It is the binding a maintainer would add; the C ABI it calls is
// exercised by the Python/ctypes harness and the C++ tools in this repository.
//
// Contract kept from the reference (SURVEY.md 8(b)):
//   - none of these functions returns an error, so on ANY non-zero status the shim runs the
//     pure-Go implementation (GPU absent, out of memory, ...);
//   - pointer-identity guards (sigma<=0, strength<=0, dims<3, orientation<=1, already-fits)
//     stay in Go, before the boundary;
//   - weight tables are computed in Go with Go's math.Exp/math.Sin and passed in, so kernels
//     see bit-for-bit the reference's tables;
//   - outputs are Go-allocated (image.NewNRGBA) and filled by C; C retains no Go pointer.
package fennec

/*
#include <stdint.h>
#include "fennec_hip.h"
*/
import "C"

import (
	"image"
	"image/color"
	"log"
	"math"
	"os"
	"runtime"
	"sync"
	"unsafe"
)

// ---- context pool --------------------------------------------------------------------------
// A fnx_ctx is not re-entrant and HIP's current device is per OS thread, while goroutines
// migrate between threads: every call borrows a context (which binds its device itself).
// Idle contexts are kept PER DEVICE and handed out round-robin over the devices on every get(), reuse included:
// a single LIFO free list pinned sequential callers to whichever device's context was returned last (device 0).
type hipPool struct {
	mu   sync.Mutex
	free [][]*C.fnx_ctx // free[dev]: idle contexts of that device
	dev  map[*C.fnx_ctx]int
	next int
	ndev int
}

var pool = func() *hipPool {
	n := int(C.fnx_device_count())
	return &hipPool{ndev: n, free: make([][]*C.fnx_ctx, n), dev: map[*C.fnx_ctx]int{}}
}()

func (p *hipPool) get() *C.fnx_ctx {
	if p.ndev == 0 {
		return nil
	}
	p.mu.Lock()
	dev := p.next % p.ndev // CompressBatch workers spread round-robin over the node's GPUs, on creation AND reuse
	p.next++
	if n := len(p.free[dev]); n > 0 {
		c := p.free[dev][n-1]
		p.free[dev] = p.free[dev][:n-1]
		p.mu.Unlock()
		return c
	}
	p.mu.Unlock()
	var c *C.fnx_ctx
	if C.fnx_ctx_create(C.int(dev), &c) != C.FNX_OK {
		return nil
	}
	p.mu.Lock()
	p.dev[c] = dev
	p.mu.Unlock()
	return c
}

func (p *hipPool) put(c *C.fnx_ctx) {
	p.mu.Lock()
	d := p.dev[c]
	p.free[d] = append(p.free[d], c)
	p.mu.Unlock()
}

// poolGetIf borrows a context only when the call will really use it.
func poolGetIf(cond bool) *C.fnx_ctx {
	if !cond {
		return nil
	}
	return pool.get()
}

// ---- CPU fallbacks: counted, never silent ------------------------------------------------------
// None of the shadowed functions can return an error (SURVEY 8(b)), so a non-zero status -- no GPU, out of memory, a
// refused argument -- runs the reference's own Go body.  That must not go unnoticed in production: every such call is
// counted per function, HIPFallbacks() exposes the counts (a metrics endpoint, a test's assertion that a GPU build
// really ran on the GPU), and FENNEC_HIP_LOG=1 logs the first fallback of each function with the library's error.
var (
	fallbackMu     sync.Mutex
	fallbackCounts = map[string]int64{}
	fallbackLog    = os.Getenv("FENNEC_HIP_LOG") == "1"
)

func fellBack(fn string) {
	fallbackMu.Lock()
	fallbackCounts[fn]++
	first := fallbackCounts[fn] == 1
	fallbackMu.Unlock()
	if first && fallbackLog {
		log.Printf("fennec_hip: %s fell back to the Go path: %s", fn, C.GoString(C.fnx_last_error()))
	}
}

// HIPFallbacks returns how many calls of each shadowed function ran the pure-Go body instead of the HIP library.
func HIPFallbacks() map[string]int64 {
	fallbackMu.Lock()
	defer fallbackMu.Unlock()
	out := make(map[string]int64, len(fallbackCounts))
	for k, v := range fallbackCounts {
		out[k] = v
	}
	return out
}

func pix(img *image.NRGBA) *C.uint8_t {
	if len(img.Pix) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&img.Pix[0]))
}

var ssimWindow = gaussianKernel(8, 1.5) // ssim.go:77, Go's own math.Exp

// ---- ssim.go -------------------------------------------------------------------------------

// SSIMFast replaces ssim.go:48.
func SSIMFast(img1, img2 *image.NRGBA) float64 {
	if c := pool.get(); c != nil {
		defer pool.put(c)
		var out C.double
		w, h := img1.Bounds().Dx(), img1.Bounds().Dy()
		var st C.int
		if w <= 512 && h <= 512 && (w < 8 || h < 8) {
			st = pixelSSIMHIP(c, img1, img2, w, h, &out) // ssim.go:62-64 with len(Pix) as Go sees it
		} else {
			st = C.fnx_ssim_fast(c, C.FNX_HOST, pix(img1), C.int(img1.Stride), pix(img2), C.int(img2.Stride),
				C.int(w), C.int(h), (*C.double)(unsafe.Pointer(&ssimWindow[0])), &out)
		}
		runtime.KeepAlive(img1)
		runtime.KeepAlive(img2)
		if st == C.FNX_OK {
			return float64(out)
		}
	}
	fellBack("SSIMFast")
	return ssimFastGo(img1, img2)
}

// pixelSSIM (ssim.go:169-204) loops `for i := 0; i < len(a.Pix); i += 4`: for a SubImage that runs to the end of the
// parent's buffer, which (pointer, stride, w, h) cannot say -- so this branch hands over the slices' real lengths.
// A b.Pix shorter than a.Pix panics in the reference (index out of range); the library refuses it and the Go
// fallback then panics exactly as the reference does.
func pixelSSIMHIP(c *C.fnx_ctx, a, b *image.NRGBA, w, h int, out *C.double) C.int {
	return C.fnx_pixel_ssim(c, C.FNX_HOST, pix(a), C.size_t(len(a.Pix)), pix(b), C.size_t(len(b.Pix)), C.int(w), C.int(h), out)
}

// SSIM replaces ssim.go:24.
func SSIM(img1, img2 image.Image) float64 {
	a, b := toNRGBARef(img1), toNRGBARef(img2)
	w, h := a.Bounds().Dx(), a.Bounds().Dy()
	if w != b.Bounds().Dx() || h != b.Bounds().Dy() {
		b = lanczosResize(b, w, h) // ssim.go:31-33 (itself on the GPU)
	}
	if c := pool.get(); c != nil {
		defer pool.put(c)
		var out C.double
		var st C.int
		if w < 8 || h < 8 {
			st = pixelSSIMHIP(c, a, b, w, h, &out) // ssim.go:35-37
		} else {
			st = C.fnx_ssim(c, C.FNX_HOST, pix(a), C.int(a.Stride), pix(b), C.int(b.Stride), C.int(w), C.int(h),
				(*C.double)(unsafe.Pointer(&ssimWindow[0])), &out)
		}
		runtime.KeepAlive(a)
		runtime.KeepAlive(b)
		if st == C.FNX_OK {
			return float64(out)
		}
	}
	fellBack("SSIM")
	return ssimGo(a, b)
}

// MSSSIM replaces ssim.go:313.
func MSSSIM(img1, img2 image.Image) float64 {
	a, b := toNRGBARef(img1), toNRGBARef(img2)
	w, h := a.Bounds().Dx(), a.Bounds().Dy()
	if w != b.Bounds().Dx() || h != b.Bounds().Dy() {
		b = lanczosResize(b, w, h) // ssim.go:320-322
	}
	if c := pool.get(); c != nil {
		defer pool.put(c)
		var out C.double
		st := C.fnx_msssim(c, C.FNX_HOST, pix(a), C.int(a.Stride), pix(b), C.int(b.Stride), C.int(w), C.int(h),
			(*C.double)(unsafe.Pointer(&ssimWindow[0])), &out, nil)
		runtime.KeepAlive(a)
		runtime.KeepAlive(b)
		if st == C.FNX_OK {
			return float64(out)
		}
	}
	fellBack("MSSSIM")
	return msssimGo(a, b)
}

// boxDownsample replaces ssim.go:244.
func boxDownsample(img *image.NRGBA, dstW, dstH int) *image.NRGBA {
	srcW, srcH := img.Bounds().Dx(), img.Bounds().Dy()
	if srcW <= 0 || srcH <= 0 || dstW <= 0 || dstH <= 0 {
		return image.NewNRGBA(image.Rect(0, 0, 0, 0))
	}
	if c := pool.get(); c != nil {
		defer pool.put(c)
		dst := image.NewNRGBA(image.Rect(0, 0, dstW, dstH))
		st := C.fnx_box_downsample(c, C.FNX_HOST, pix(img), C.int(img.Stride), C.int(srcW), C.int(srcH),
			pix(dst), C.int(dst.Stride), C.int(dstW), C.int(dstH))
		runtime.KeepAlive(img)
		if st == C.FNX_OK {
			return dst
		}
	}
	fellBack("boxDownsample")
	return boxDownsampleGo(img, dstW, dstH)
}

// ---- resize.go -----------------------------------------------------------------------------

// csr flattens precomputeWeights' [][]weightEntry (resize.go:164-197) for the C ABI.
func csr(weights [][]weightEntry) (off, idx []C.int32_t, wt []C.double) {
	off = make([]C.int32_t, len(weights)+1)
	n := 0
	for d, e := range weights {
		off[d] = C.int32_t(n)
		n += len(e)
	}
	off[len(weights)] = C.int32_t(n)
	idx = make([]C.int32_t, n+1)
	wt = make([]C.double, n+1)
	k := 0
	for _, e := range weights {
		for _, t := range e {
			idx[k], wt[k] = C.int32_t(t.index), C.double(t.weight)
			k++
		}
	}
	return
}

func lanczosTable(dst, src int) ([]C.int32_t, []C.int32_t, []C.double) {
	ratio := float64(src) / float64(dst) // resize.go:81-87
	support := lanczosA
	if ratio > 1 {
		support = lanczosA * ratio
	}
	return csr(precomputeWeights(dst, src, ratio, support))
}

// lanczosResize replaces resize.go:37 (smartResize, resize.go:12, calls it unchanged).
func lanczosResize(img *image.NRGBA, dstW, dstH int) *image.NRGBA {
	srcW, srcH := img.Bounds().Dx(), img.Bounds().Dy()
	if srcW <= 0 || srcH <= 0 || dstW <= 0 || dstH <= 0 {
		return image.NewNRGBA(image.Rect(0, 0, 0, 0))
	}
	if srcW == dstW && srcH == dstH {
		return lanczosResizeGo(img, dstW, dstH) // flat copy, no reason to cross the bus
	}
	if c := pool.get(); c != nil {
		defer pool.put(c)
		offH, idxH, wH := lanczosTable(dstW, srcW)
		offV, idxV, wV := lanczosTable(dstH, srcH)
		dst := image.NewNRGBA(image.Rect(0, 0, dstW, dstH))
		st := C.fnx_lanczos_resize(c, C.FNX_HOST, pix(img), C.int(img.Stride), C.int(srcW), C.int(srcH),
			&offH[0], &idxH[0], &wH[0], &offV[0], &idxV[0], &wV[0],
			pix(dst), C.int(dst.Stride), C.int(dstW), C.int(dstH))
		runtime.KeepAlive(img)
		if st == C.FNX_OK {
			return dst
		}
	}
	fellBack("lanczosResize")
	return lanczosResizeGo(img, dstW, dstH)
}

// ---- effects.go ----------------------------------------------------------------------------

// GaussianBlur replaces effects.go:146.
func GaussianBlur(img *image.NRGBA, sigma float64) *image.NRGBA {
	if sigma <= 0 {
		return img // same pointer (effects.go:147-149)
	}
	w, h := img.Bounds().Dx(), img.Bounds().Dy()
	radius := int(math.Ceil(sigma * 3))
	kernel := make([]float64, radius*2+1) // effects.go:155-165, Go's math.Exp
	var sum float64
	for i := range kernel {
		x := float64(i - radius)
		kernel[i] = math.Exp(-(x * x) / (2 * sigma * sigma))
		sum += kernel[i]
	}
	for i := range kernel {
		kernel[i] /= sum
	}
	if c := poolGetIf(w > 0 && h > 0); c != nil { // (no borrow for empty images: a borrowed ctx must always be returned)
		defer pool.put(c)
		dst := image.NewNRGBA(image.Rect(0, 0, w, h))
		// FNX_BLUR_EXACT reproduces the reference bit for bit.  A host-space call is PCIe-bound
		// (1.2 ms per 4K image, of which the exact kernel is ~0.03 ms and the fast one ~0.03 ms), so
		// the drop-in takes the exact mode; FNX_BLUR_FAST (<= 1 LSB on <= 0.1 % of samples) is for
		// device-resident pipelines that ask for it.
		st := C.fnx_gaussian_blur(c, C.FNX_HOST, pix(img), C.int(img.Stride), C.int(w), C.int(h),
			(*C.double)(unsafe.Pointer(&kernel[0])), C.int(radius), C.FNX_BLUR_EXACT, pix(dst), C.int(dst.Stride))
		runtime.KeepAlive(img)
		if st == C.FNX_OK {
			return dst
		}
	}
	fellBack("GaussianBlur")
	return gaussianBlurGo(img, sigma)
}

func sharpenHIP(img *image.NRGBA, amount float64, adaptive bool) *image.NRGBA {
	c := pool.get()
	if c == nil {
		return nil
	}
	defer pool.put(c)
	w, h := img.Bounds().Dx(), img.Bounds().Dy()
	dst := image.NewNRGBA(image.Rect(0, 0, w, h))
	var st C.int
	if adaptive {
		st = C.fnx_adaptive_sharpen(c, C.FNX_HOST, pix(img), C.int(img.Stride), C.int(w), C.int(h), C.double(amount), pix(dst), C.int(dst.Stride))
	} else {
		st = C.fnx_sharpen(c, C.FNX_HOST, pix(img), C.int(img.Stride), C.int(w), C.int(h), C.double(amount), pix(dst), C.int(dst.Stride))
	}
	runtime.KeepAlive(img)
	if st != C.FNX_OK {
		return nil
	}
	return dst
}

// Sharpen replaces effects.go:10.
func Sharpen(img *image.NRGBA, strength float64) *image.NRGBA {
	if strength <= 0 {
		return img
	}
	if strength > 1 {
		strength = 1
	}
	if img.Bounds().Dx() < 3 || img.Bounds().Dy() < 3 {
		return img
	}
	if dst := sharpenHIP(img, 1.0+strength*1.5, false); dst != nil {
		return dst
	}
	fellBack("Sharpen")
	return sharpenGo(img, strength)
}

// AdaptiveSharpen replaces effects.go:49.
func AdaptiveSharpen(img *image.NRGBA, strength float64) *image.NRGBA {
	if strength <= 0 {
		return img
	}
	if strength > 1 {
		strength = 1
	}
	if img.Bounds().Dx() < 3 || img.Bounds().Dy() < 3 {
		return img
	}
	if dst := sharpenHIP(img, 1.0+strength*2.0, true); dst != nil {
		return dst
	}
	fellBack("AdaptiveSharpen")
	return adaptiveSharpenGo(img, strength)
}

// ---- exif.go -------------------------------------------------------------------------------

// ApplyOrientation replaces exif.go:178.
func ApplyOrientation(img *image.NRGBA, orient Orientation) *image.NRGBA {
	if orient < 2 || orient > 8 {
		return img // exif.go:180-181,200-201
	}
	w, h := img.Bounds().Dx(), img.Bounds().Dy()
	if c := poolGetIf(w > 0 && h > 0); c != nil {
		defer pool.put(c)
		ow, oh := w, h
		if orient >= 5 {
			ow, oh = h, w
		}
		dst := image.NewNRGBA(image.Rect(0, 0, ow, oh))
		st := C.fnx_orient(c, C.FNX_HOST, pix(img), C.int(img.Stride), C.int(w), C.int(h), C.int(orient), pix(dst), C.int(dst.Stride))
		runtime.KeepAlive(img)
		if st == C.FNX_OK {
			return dst
		}
	}
	fellBack("ApplyOrientation")
	return applyOrientationGo(img, orient)
}

// ---- analyze.go ----------------------------------------------------------------------------

// Analyze replaces analyze.go:26: the device gathers every statistic that needs the pixels
// (fnx_analysis); the float epilogue is the reference's own code on those numbers, so entropy
// and the recommend* rules are computed with Go's math package exactly as before.
func Analyze(img image.Image) ImageStats {
	src := toNRGBARef(img)
	w, h := src.Bounds().Dx(), src.Bounds().Dy()
	if w == 0 || h == 0 {
		return ImageStats{Width: w, Height: h} // analyze.go:37-39
	}
	if c := pool.get(); c != nil {
		defer pool.put(c)
		var a C.fnx_analysis
		st := C.fnx_analyze(c, C.FNX_HOST, pix(src), C.int(src.Stride), C.int(w), C.int(h), &a)
		runtime.KeepAlive(src)
		if st == C.FNX_OK {
			n := float64(w * h)
			stats := ImageStats{Width: w, Height: h}
			stats.HasAlpha = a.has_alpha != 0
			stats.IsGrayscale = a.is_grayscale != 0
			stats.UniqueColors = int(a.unique_colors)
			stats.MeanBrightness = float64(a.bright_sum) / n // analyze.go:90
			if a.sample_count > 0 {
				stats.Contrast = math.Sqrt(float64(a.variance_sum) / float64(a.sample_count)) // :111-113
			}
			var histogram [256]float64
			for i := range histogram {
				histogram[i] = float64(a.histogram[i])
			}
			stats.Entropy = computeEntropy(histogram[:], n) // :116
			if a.edge_total > 0 {
				stats.EdgeDensity = float64(a.edge_count) / float64(a.edge_total) // :180-183
			}
			stats.RecommendedFormat = recommendFormat(stats)
			stats.RecommendedQuality = recommendQuality(stats)
			stats.EstimatedCompression = estimateCompression(stats)
			return stats
		}
	}
	fellBack("Analyze")
	return analyzeGo(img)
}

// GaussianBlurScored is an addition, not a replacement: GaussianBlur (effects.go:146) and
// SSIMFast(img, blurred) (ssim.go:48) as ONE crossing of the boundary.  The two reference-shaped calls move
// 4 x 33 MB of a 4K image over PCIe (the source twice up, the blurred image down and up again) for 40 us of
// kernels; this one moves 2 x 33 MB (source up, blurred image down) -- 1.3 ms instead of 2.5 -- and on the
// device the blur kernel gathers SSIMFast's boxDownsample sums while it has the pixels.  Bytes and score are
// those of GaussianBlur followed by SSIMFast.  Callers that score what they blur (a quality loop over sigma,
// a preview pipeline) call this; the exported reference functions stay what they were.
func GaussianBlurScored(img *image.NRGBA, sigma float64) (*image.NRGBA, float64) {
	if sigma <= 0 {
		return img, SSIMFast(img, img) // GaussianBlur returns the same pointer (effects.go:147-149)
	}
	w, h := img.Bounds().Dx(), img.Bounds().Dy()
	radius := int(math.Ceil(sigma * 3))
	kernel := make([]float64, radius*2+1) // effects.go:155-165, Go's math.Exp
	var sum float64
	for i := range kernel {
		x := float64(i - radius)
		kernel[i] = math.Exp(-(x * x) / (2 * sigma * sigma))
		sum += kernel[i]
	}
	for i := range kernel {
		kernel[i] /= sum
	}
	if c := poolGetIf(w > 0 && h > 0); c != nil {
		defer pool.put(c)
		dst := image.NewNRGBA(image.Rect(0, 0, w, h))
		var out C.double
		st := C.fnx_gaussian_blur_ssim_fast(c, C.FNX_HOST, pix(img), C.int(img.Stride), C.int(w), C.int(h),
			(*C.double)(unsafe.Pointer(&kernel[0])), C.int(radius), C.FNX_BLUR_EXACT, pix(dst), C.int(dst.Stride),
			(*C.double)(unsafe.Pointer(&ssimWindow[0])), &out)
		runtime.KeepAlive(img)
		if st == C.FNX_OK {
			return dst, float64(out)
		}
	}
	fellBack("GaussianBlurScored")
	b := gaussianBlurGo(img, sigma)
	return b, ssimFastGo(img, b)
}

// ---- compress.go: the quality binary search (compress.go:45-74) ----------------------------

// ssimAgainst scores decoded candidates against ONE prepared source: the source's downsampled
// side is computed once per search instead of once per step, and a candidate that the decoder
// returned as *image.YCbCr / *image.Gray crosses PCIe as planes (1.5 bytes per pixel at 4:2:0)
// and is converted on the device (the arithmetic of color.YCbCr.RGBA + convert.go:48-53).
// compressJPEGOptimal's loop body changes from
//
//	decoded, _ := jpeg.Decode(...); ssim := SSIMFast(src, toNRGBARef(decoded))
//
// to `ssim := ref.against(decoded)`, with `ref := prepareSSIM(src); defer ref.close()` before it.
type ssimRef struct {
	c   *C.fnx_ctx
	p   *C.fnx_prepared
	src *image.NRGBA
}

func prepareSSIM(src *image.NRGBA) *ssimRef {
	r := &ssimRef{src: src}
	if c := pool.get(); c != nil {
		var p *C.fnx_prepared
		if C.fnx_ssim_fast_prepare(c, C.FNX_HOST, pix(src), C.int(src.Stride),
			C.int(src.Bounds().Dx()), C.int(src.Bounds().Dy()), &p) == C.FNX_OK {
			r.c, r.p = c, p
		} else {
			pool.put(c)
		}
	}
	return r
}

func (r *ssimRef) close() {
	if r.p != nil {
		C.fnx_prepared_free(r.c, r.p)
		pool.put(r.c)
		r.p = nil
	}
}

func (r *ssimRef) against(decoded image.Image) float64 {
	if r.p != nil {
		var out C.double
		st := C.int(-1)
		switch d := decoded.(type) {
		case *image.YCbCr:
			if d.Rect.Min == (image.Point{}) && d.Rect.Dx() == r.src.Bounds().Dx() && d.Rect.Dy() == r.src.Bounds().Dy() {
				st = C.fnx_ssim_fast_against_ycbcr(r.c, r.p, C.FNX_HOST,
					(*C.uint8_t)(unsafe.Pointer(&d.Y[0])), C.int(d.YStride),
					(*C.uint8_t)(unsafe.Pointer(&d.Cb[0])), (*C.uint8_t)(unsafe.Pointer(&d.Cr[0])), C.int(d.CStride),
					C.int(d.SubsampleRatio), (*C.double)(unsafe.Pointer(&ssimWindow[0])), &out)
			}
		case *image.Gray:
			if d.Rect.Min == (image.Point{}) && d.Rect.Dx() == r.src.Bounds().Dx() && d.Rect.Dy() == r.src.Bounds().Dy() {
				st = C.fnx_ssim_fast_against_ycbcr(r.c, r.p, C.FNX_HOST,
					(*C.uint8_t)(unsafe.Pointer(&d.Pix[0])), C.int(d.Stride), nil, nil, 0, 0,
					(*C.double)(unsafe.Pointer(&ssimWindow[0])), &out)
			}
		case *image.NRGBA:
			st = C.fnx_ssim_fast_against(r.c, r.p, C.FNX_HOST, pix(d), C.int(d.Stride),
				(*C.double)(unsafe.Pointer(&ssimWindow[0])), &out)
		}
		runtime.KeepAlive(decoded)
		if st == C.FNX_OK {
			return float64(out)
		}
	}
	fellBack("ssimRef.against")
	return ssimFastGo(r.src, toNRGBARef(decoded)) // anything else: the reference's own path
}

// ---- targetsize.go: scale searches over one source (targetsize.go:240-313) ------------------

// residentSrc keeps ONE source image in device memory for the scale searches: findBestScaleBinary,
// findBestScaleFixed and scaleSearch call boxDownsample(src, w, h) 10-12 times on the same `src` and
// encode every result on the host.  Their loop bodies change from `boxDownsample(src, newW, newH)` to
// `rs.boxDownsample(newW, newH)`, with `rs := uploadSrc(src); defer rs.close()` before the loop: the
// 33 MB source crosses PCIe once, each iteration is the box kernel plus the D2H copy of the small result
// (FNX_DEVICE_SRC).
type residentSrc struct {
	c      *C.fnx_ctx
	d      unsafe.Pointer
	src    *image.NRGBA
	w, h   int
	stride int
}

func uploadSrc(src *image.NRGBA) *residentSrc {
	r := &residentSrc{src: src, w: src.Bounds().Dx(), h: src.Bounds().Dy()}
	if r.w <= 0 || r.h <= 0 {
		return r
	}
	if c := pool.get(); c != nil {
		r.stride = 4 * r.w
		if C.fnx_malloc(c, C.size_t(r.stride*r.h), &r.d) == C.FNX_OK &&
			C.fnx_upload(c, r.d, C.int(r.stride), unsafe.Pointer(pix(src)), C.int(src.Stride), C.int(r.w), C.int(r.h)) == C.FNX_OK {
			r.c = c
		} else {
			if r.d != nil {
				C.fnx_free(c, r.d)
				r.d = nil
			}
			pool.put(c)
		}
	}
	return r
}

func (r *residentSrc) close() {
	if r.c != nil {
		C.fnx_free(r.c, r.d)
		pool.put(r.c)
		r.c, r.d = nil, nil
	}
}

func (r *residentSrc) boxDownsample(dstW, dstH int) *image.NRGBA {
	if r.c != nil && dstW > 0 && dstH > 0 {
		dst := image.NewNRGBA(image.Rect(0, 0, dstW, dstH))
		if C.fnx_box_downsample(r.c, C.FNX_DEVICE_SRC, (*C.uint8_t)(r.d), C.int(r.stride), C.int(r.w), C.int(r.h),
			pix(dst), C.int(dst.Stride), C.int(dstW), C.int(dstH)) == C.FNX_OK {
			return dst
		}
	}
	return boxDownsample(r.src, dstW, dstH) // per-call path (and through it the reference's own body)
}

// ---- targetsize.go: applyPalette (targetsize.go:488) ----------------------------------------

// applyPalette replaces targetsize.go:488 for opaque palettes (what medianCut builds).
func applyPalette(src *image.NRGBA, palette color.Palette) *image.Paletted {
	w, h := src.Bounds().Dx(), src.Bounds().Dy()
	pal := make([]C.uint8_t, 0, 4*len(palette))
	ok := len(palette) >= 1 && len(palette) <= 256 && w > 0 && h > 0
	for _, c := range palette {
		n, isN := c.(color.NRGBA)
		if !isN || n.A != 255 {
			ok = false
			break
		}
		pal = append(pal, C.uint8_t(n.R), C.uint8_t(n.G), C.uint8_t(n.B), 255)
	}
	if !ok {
		return applyPaletteGo(src, palette)
	}
	if c := pool.get(); c != nil {
		defer pool.put(c)
		indexed := image.NewPaletted(src.Bounds(), palette)
		st := C.fnx_apply_palette(c, C.FNX_HOST, pix(src), C.int(src.Stride), C.int(w), C.int(h),
			&pal[0], C.int(len(palette)), (*C.uint8_t)(unsafe.Pointer(&indexed.Pix[0])), C.int(indexed.Stride), nil, 0)
		runtime.KeepAlive(src)
		if st == C.FNX_OK {
			return indexed
		}
	}
	fellBack("applyPalette")
	return applyPaletteGo(src, palette)
}

// ---- compress.go: the whole quality search on the device (opt-in; DESIGN.md 3.11) -----------

// jpegQualitySearchHIP runs compressJPEGOptimal's binary search (compress.go:24-74) with every candidate quality
// round-tripped through image/jpeg's LOSSY arithmetic on the device (colour conversion, 4:2:0, FDCT, quantise,
// dequantise, IDCT -- no entropy coding) and scored with SSIMFast there.  ok == false means the device was not
// used and the caller runs the loop as before; found == false is the reference's "no quality reached the target"
// (compress.go:82-86: encode at 100).  The caller encodes ONCE, at the returned quality:
//
//	if q, ssim, found, ok := jpegQualitySearchHIP(src, targetSSIM); ok { bestQuality, bestSSIM = q, ssim; ... }
//
// Off by default (useDeviceSearch): the arithmetic is restated from the algorithms image/jpeg implements and must be
// pinned against the standard library first -- TestHIPRoundTripMatchesStdlib in INTEGRATION.md 4.2.
var useDeviceSearch = false

func jpegQualitySearchHIP(src *image.NRGBA, targetSSIM float64) (quality int, ssim float64, found, ok bool) {
	w, h := src.Bounds().Dx(), src.Bounds().Dy()
	if !useDeviceSearch || w <= 0 || h <= 0 {
		return 0, 0, false, false
	}
	c := pool.get()
	if c == nil {
		return 0, 0, false, false
	}
	defer pool.put(c)
	var q, steps C.int
	var s C.double
	st := C.fnx_jpeg_quality_search(c, C.FNX_HOST, pix(src), C.int(src.Stride), C.int(w), C.int(h), C.double(targetSSIM),
		(*C.double)(unsafe.Pointer(&ssimWindow[0])), &q, &s, &steps)
	runtime.KeepAlive(src)
	if st != C.FNX_OK && st != C.FNX_NOOP {
		return 0, 0, false, false
	}
	return int(q), float64(s), st == C.FNX_OK, true
}

// jpegRoundTripHIP is toNRGBARef(jpeg.Decode(jpeg.Encode(src, quality))) as far as the pixels go; nil when the device
// was not used.  It exists for the pinning test.
func jpegRoundTripHIP(src *image.NRGBA, quality int) *image.NRGBA {
	w, h := src.Bounds().Dx(), src.Bounds().Dy()
	c := poolGetIf(w > 0 && h > 0)
	if c == nil {
		return nil
	}
	defer pool.put(c)
	dst := image.NewNRGBA(image.Rect(0, 0, w, h))
	st := C.fnx_jpeg_roundtrip(c, C.FNX_HOST, pix(src), C.int(src.Stride), C.int(w), C.int(h), C.int(quality),
		pix(dst), C.int(dst.Stride))
	runtime.KeepAlive(src)
	if st != C.FNX_OK {
		return nil
	}
	return dst
}

// jpegEncodeHIP is jpeg.Encode(w, src, &jpeg.Options{Quality: quality})'s file from the device (fnx_jpeg_encode:
// baseline, 4:2:0, the typical Huffman tables, writer.go's segment order); nil when the device was not used.
func jpegEncodeHIP(src *image.NRGBA, quality int) []byte {
	w, h := src.Bounds().Dx(), src.Bounds().Dy()
	c := poolGetIf(useDeviceSearch && w > 0 && h > 0)
	if c == nil {
		return nil
	}
	defer pool.put(c)
	buf := make([]byte, 4096+w*h*3/2)
	for try := 0; try < 2; try++ {
		var n C.size_t
		st := C.fnx_jpeg_encode(c, C.FNX_HOST, pix(src), C.int(src.Stride), C.int(w), C.int(h), C.int(quality),
			(*C.uint8_t)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)), &n)
		runtime.KeepAlive(src)
		if st == C.FNX_OK {
			return buf[:int(n)]
		}
		if int(n) <= len(buf) {
			return nil
		}
		buf = make([]byte, int(n))
	}
	return nil
}

// jpegCompressHIP is compressJPEGOptimal (compress.go:21-87) in one call: the search and the winning file on the
// device.  ok == false: the device was not used and the caller runs the loop as before.
func jpegCompressHIP(src *image.NRGBA, targetSSIM float64) (data []byte, quality int, ssim float64, ok bool) {
	w, h := src.Bounds().Dx(), src.Bounds().Dy()
	c := poolGetIf(useDeviceSearch && w > 0 && h > 0)
	if c == nil {
		return nil, 0, 0, false
	}
	defer pool.put(c)
	buf := make([]byte, 4096+w*h*3/2)
	for try := 0; try < 2; try++ {
		var n C.size_t
		var q, steps C.int
		var s C.double
		st := C.fnx_jpeg_compress(c, C.FNX_HOST, pix(src), C.int(src.Stride), C.int(w), C.int(h), C.double(targetSSIM),
			(*C.double)(unsafe.Pointer(&ssimWindow[0])), (*C.uint8_t)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)), &n, &q, &s, &steps)
		runtime.KeepAlive(src)
		if st == C.FNX_OK {
			return buf[:int(n)], int(q), float64(s), true
		}
		if int(n) <= len(buf) {
			return nil, 0, 0, false
		}
		buf = make([]byte, int(n))
	}
	return nil, 0, 0, false
}

// jpegDecodeHIP is image.Decode + toNRGBARef for a JPEG file (io.go:60-95) on the device.  ok == false: the device was
// not used, does not take this file (12-bit samples, arithmetic coding: FNX_ERR_UNSUPPORTED) or finds it damaged
// (FNX_ERR_INVALID) -- in every such case the caller runs image.Decode as before, and what the reference says about a
// damaged file (an error, or an image: its decoder forgives some damage the device's strict block accounting does not)
// stays the reference's own answer.
func jpegDecodeHIP(data []byte) (img *image.NRGBA, ok bool) {
	c := poolGetIf(useDeviceSearch && len(data) > 4)
	if c == nil {
		return nil, false
	}
	defer pool.put(c)
	var w, h C.int
	if C.fnx_jpeg_decode(c, (*C.uint8_t)(unsafe.Pointer(&data[0])), C.size_t(len(data)), C.FNX_HOST, nil, 0, &w, &h) != C.FNX_OK {
		return nil, false
	}
	img = image.NewNRGBA(image.Rect(0, 0, int(w), int(h)))
	st := C.fnx_jpeg_decode(c, (*C.uint8_t)(unsafe.Pointer(&data[0])), C.size_t(len(data)), C.FNX_HOST, pix(img), C.int(img.Stride), &w, &h)
	runtime.KeepAlive(data)
	if st != C.FNX_OK {
		return nil, false
	}
	return img, true
}

// jpegRecompressHIP is CompressBatch's item body for a JPEG source (batch.go:88-122 -> compress.go:21-87) in one call:
// the file goes up, decoder + quality search + encoder run on the device, the new file comes down.  ok == false: the
// device was not used or refuses the file; the caller decodes on the host and calls jpegCompressHIP (or its own loop).
func jpegRecompressHIP(data []byte, targetSSIM float64) (out []byte, quality int, ssim float64, w, h int, ok bool) {
	c := poolGetIf(useDeviceSearch && len(data) > 4)
	if c == nil {
		return nil, 0, 0, 0, 0, false
	}
	defer pool.put(c)
	buf := make([]byte, len(data)+4096)
	for try := 0; try < 2; try++ {
		var n C.size_t
		var q, steps, cw, ch C.int
		var s C.double
		st := C.fnx_jpeg_recompress(c, (*C.uint8_t)(unsafe.Pointer(&data[0])), C.size_t(len(data)), C.double(targetSSIM),
			(*C.double)(unsafe.Pointer(&ssimWindow[0])), (*C.uint8_t)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)), &n, &q, &s, &steps, &cw, &ch)
		runtime.KeepAlive(data)
		if st == C.FNX_OK {
			return buf[:int(n)], int(q), float64(s), int(cw), int(ch), true
		}
		if st == C.FNX_ERR_UNSUPPORTED || int(n) <= len(buf) {
			return nil, 0, 0, 0, 0, false
		}
		buf = make([]byte, int(n))
	}
	return nil, 0, 0, 0, 0, false
}

// compressFileJPEGHIP is CompressFile's standard-mode JPEG path for a .jpg source (fennec.go:30-76, :107-141, :162-205) in
// one call: decode, ApplyOrientation (opts.AutoOrient, orient as openWithOrientation read it), smartResize (MaxWidth /
// MaxHeight), analyzeFormat (Format == Auto), compressJPEGOptimal -- every pixel stage on the device.
//   ok == false            the device was not used or refuses the file: the reference's own path
//   ok && data == nil      analyzeFormat chose PNG: the caller's compressPNG (orig / final dims are set)
func compressFileJPEGHIP(file []byte, orient Orientation, opts Options, targetSSIM float64) (data []byte, quality int, ssim float64,
	orig, final image.Point, ok bool) {
	c := poolGetIf(useDeviceSearch && len(file) > 4 && opts.TargetSize == 0 && (opts.Format == JPEG || opts.Format == Auto))
	if c == nil {
		return nil, 0, 0, image.Point{}, image.Point{}, false
	}
	defer pool.put(c)
	var fo C.fennec_FileOptions
	if opts.AutoOrient {
		fo.orient = C.int32_t(orient)
	} else {
		fo.orient = 1
	}
	fo.max_w, fo.max_h = C.int32_t(opts.MaxWidth), C.int32_t(opts.MaxHeight)
	if opts.Format == Auto {
		fo.auto_format = 1
	}
	fo.target_ssim = C.double(targetSSIM)
	buf := make([]byte, len(file)+4096)
	for try := 0; try < 2; try++ {
		var n C.size_t
		var q, steps C.int
		var s C.double
		var dims [4]C.int
		st := C.fennec_CompressFileJPEG(c, (*C.uint8_t)(unsafe.Pointer(&file[0])), C.size_t(len(file)), &fo,
			(*C.uint8_t)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)), &n, &q, &s, &steps, &dims[0])
		runtime.KeepAlive(file)
		orig, final = image.Pt(int(dims[0]), int(dims[1])), image.Pt(int(dims[2]), int(dims[3]))
		switch {
		case st == C.FNX_OK:
			return buf[:int(n)], int(q), float64(s), orig, final, true
		case st == C.FNX_NOOP:
			return nil, 0, 1.0, orig, final, true
		case st == C.FNX_ERR_INVALID && int(n) > len(buf):
			buf = make([]byte, int(n))
		default:
			return nil, 0, 0, image.Point{}, image.Point{}, false
		}
	}
	return nil, 0, 0, image.Point{}, image.Point{}, false
}

// jpegQualitySearchOptHIP is jpegQualitySearchOpt (targetsize.go:125-176) on the device: every candidate's size from the
// device's entropy coder, the winner's file and (unless skipSSIM) its SSIMFast.  ok == false: the device was not used;
// data == nil with ok: no quality fits (the reference returns nil, nil).
func jpegQualitySearchOptHIP(src *image.NRGBA, targetBytes int, skipSSIM bool) (data []byte, quality int, ssim float64, ok bool) {
	w, h := src.Bounds().Dx(), src.Bounds().Dy()
	c := poolGetIf(useDeviceSearch && w > 0 && h > 0)
	if c == nil {
		return nil, 0, 0, false
	}
	defer pool.put(c)
	buf := make([]byte, targetBytes+16)
	var n C.size_t
	var q, steps C.int
	var s C.double
	skip := C.int(0)
	if skipSSIM {
		skip = 1
	}
	st := C.fnx_jpeg_size_search(c, C.FNX_HOST, pix(src), C.int(src.Stride), C.int(w), C.int(h), C.longlong(targetBytes), skip,
		(*C.double)(unsafe.Pointer(&ssimWindow[0])), (*C.uint8_t)(unsafe.Pointer(&buf[0])), C.size_t(len(buf)), &n, &q, &s, &steps)
	runtime.KeepAlive(src)
	switch st {
	case C.FNX_OK:
		return buf[:int(n)], int(q), float64(s), true
	case C.FNX_NOOP:
		return nil, 0, 0, true
	}
	return nil, 0, 0, false
}
