/*
 * fennec_oracle.c -- CPU restatement of fennec's per-pixel hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it -- always as the
 * checker (or as the timed CPU baseline), never as the thing shipped.  The
 * product path (fennec_amd/) must not import, link or fall back to it.
 *
 * PARITY UNPINNED: the reference (shamspias/fennec) is pure Go, there is no Go
 * toolchain in the build image, and the reference's tests hold no golden
 * vectors for this path -- only range/invariant assertions
 * (fennec_test.go:82-163,510-560,612-736,802-821,1101-1115).  Those
 * assertions are all re-run against this file by tests/test_oracle.py, and an
 * independently written numpy restatement (tests/np_restatement.py) must
 * agree with it bit for bit, but no Go binary has ever been compared with it.
 *
 * Numeric contract: IEEE fp64, evaluated left to right
 * exactly as the Go source writes it, no FMA contraction (amd64, GOAMD64=v1).
 * Build with -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 * Weight tables (SSIM 8x8 window, blur 1-D kernel, Lanczos taps) are computed
 * here with glibc's exp()/sin(); Go's math.Exp/math.Sin may differ from glibc
 * in the last ulp, which is why every kernel-level entry point takes the
 * table as an INPUT -- the table generators are convenience, not contract.
 *
 * Every function cites the reference file:line it follows.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* parallelDo (resize.go:200-239): static contiguous split over procs  */
/* ------------------------------------------------------------------ */

typedef void (*orc_range_fn)(int from, int to, void *arg);

typedef struct {
    orc_range_fn fn;
    void *arg;
    int from, to;
} orc_job;

static void *orc_job_main(void *p)
{
    orc_job *j = (orc_job *)p;
    j->fn(j->from, j->to, j->arg);
    return NULL;
}

/* A persistent pool stands in for Go's goroutines: parallelDo starts `procs` goroutines per call (resize.go:225-236),
 * which costs Go microseconds; 256 pthread_create + join per pass cost this restatement milliseconds and held the
 * cpu_baseline leg at 3x one thread on a 256-core box.  The batches -- which rows each one takes
 * -- are the reference's; which OS thread runs a batch is not observable.  One parallel_do at a time uses the pool (a
 * second caller, from another thread, spawns threads as before). */
#define ORC_POOL_MAX 1024
static pthread_mutex_t pool_user = PTHREAD_MUTEX_INITIALIZER;     /* held by the parallel_do that owns the pool */
static pthread_mutex_t pool_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t pool_work = PTHREAD_COND_INITIALIZER, pool_idle = PTHREAD_COND_INITIALIZER;
static pthread_t pool_tid[ORC_POOL_MAX];
static int pool_threads = 0;
static orc_job *pool_jobs = NULL;
static int pool_njobs = 0, pool_next = 0, pool_done = 0;

static void *pool_main(void *unused)
{
    (void)unused;
    pthread_mutex_lock(&pool_mu);
    for (;;) {
        while (pool_next >= pool_njobs) pthread_cond_wait(&pool_work, &pool_mu);
        orc_job *j = &pool_jobs[pool_next++];
        pthread_mutex_unlock(&pool_mu);
        j->fn(j->from, j->to, j->arg);
        pthread_mutex_lock(&pool_mu);
        if (++pool_done == pool_njobs) pthread_cond_signal(&pool_idle);
    }
    return NULL;
}

/* resize.go:200-239 -- batchSize = ceil(count/procs); empty batches skipped. */
static void parallel_do(int start, int stop, int procs, orc_range_fn fn, void *arg)
{
    int count = stop - start;
    if (count <= 0) return;
    if (procs > count) procs = count;
    if (procs <= 1) {
        fn(start, stop, arg);
        return;
    }
    int batch = (count + procs - 1) / procs;
    orc_job *jobs = (orc_job *)malloc(sizeof(orc_job) * (size_t)procs);
    int n = 0;
    for (int p = 0; p < procs; p++) {
        int bs = start + p * batch, be = bs + batch;
        if (be > stop) be = stop;
        if (bs >= be) continue;
        jobs[n].fn = fn; jobs[n].arg = arg; jobs[n].from = bs; jobs[n].to = be;
        n++;
    }
    if (n <= ORC_POOL_MAX && pthread_mutex_trylock(&pool_user) == 0) {
        pthread_mutex_lock(&pool_mu);
        int ok = 1;
        while (pool_threads < n) {                                /* grow to one worker per batch, as goroutines would be */
            if (pthread_create(&pool_tid[pool_threads], NULL, pool_main, NULL) != 0) { ok = 0; break; }
            pthread_detach(pool_tid[pool_threads]);
            pool_threads++;
        }
        if (ok || pool_threads > 0) {
            pool_jobs = jobs; pool_njobs = n; pool_next = 0; pool_done = 0;
            pthread_cond_broadcast(&pool_work);
            while (pool_done < n) pthread_cond_wait(&pool_idle, &pool_mu);
            pool_njobs = 0; pool_next = 0; pool_jobs = NULL;
            pthread_mutex_unlock(&pool_mu);
            pthread_mutex_unlock(&pool_user);
            free(jobs);
            return;
        }
        pthread_mutex_unlock(&pool_mu);
        pthread_mutex_unlock(&pool_user);
    }
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n);
    for (int i = 0; i < n; i++) pthread_create(&tid[i], NULL, orc_job_main, &jobs[i]);
    for (int i = 0; i < n; i++) pthread_join(tid[i], NULL);
    free(tid);
    free(jobs);
}

/* ------------------------------------------------------------------ */
/* clampF (convert.go:149-158)                                         */
/* ------------------------------------------------------------------ */

ORC_API uint8_t orc_clampF(double x)
{
    /* math.Round = round half away from zero = C round() */
    int64_t v = (int64_t)round(x);
    if (v > 255) return 255;
    if (v < 0) return 0;
    return (uint8_t)v;
}
#define clampF orc_clampF

/* ------------------------------------------------------------------ */
/* ssim.go                                                             */
/* ------------------------------------------------------------------ */

/* ssim.go:11-17 -- Go untyped constants are exact rationals: 2.55^2, 7.65^2 */
static const double SSIM_C1 = 6.5025;
static const double SSIM_C2 = 58.5225;

/* toLuminance (ssim.go:207-220) */
ORC_API void orc_to_luminance(const uint8_t *pix, int stride, int w, int h, double *lum)
{
    for (int y = 0; y < h; y++) {
        size_t off = (size_t)y * (size_t)stride;
        for (int x = 0; x < w; x++) {
            size_t i = off + (size_t)x * 4;
            lum[(size_t)y * w + x] =
                0.299 * (double)pix[i] + 0.587 * (double)pix[i + 1] + 0.114 * (double)pix[i + 2];
        }
    }
}

/* gaussianKernel (ssim.go:223-241); taps x,y in [-half, half) */
ORC_API void orc_gaussian_kernel(int size, double sigma, double *kernel)
{
    int half = size / 2;
    double sum = 0;
    int idx = 0;
    for (int y = -half; y < half; y++) {
        for (int x = -half; x < half; x++) {
            double val = exp(-(double)(x * x + y * y) / (2 * sigma * sigma));
            kernel[idx] = val;
            sum += val;
            idx++;
        }
    }
    for (int i = 0; i < size * size; i++) kernel[i] /= sum;
}

/* one row band of windowedSSIM's body (ssim.go:110-148) */
static void ssim_band(const double *lumA, const double *lumB, int w, int startY, int endY,
                      const double *kernel, double *sum_out, long *count_out)
{
    const int half = 4;
    double localSum = 0;
    long localCount = 0;
    for (int y = startY; y < endY; y++) {
        for (int x = half; x < w - half; x++) {
            double muA = 0, muB = 0, sigAA = 0, sigBB = 0, sigAB = 0;
            int ki = 0;
            for (int wy = -half; wy < half; wy++) {
                for (int wx = -half; wx < half; wx++) {
                    size_t idx = (size_t)(y + wy) * w + (x + wx);
                    double weight = kernel[ki];
                    double va = lumA[idx], vb = lumB[idx];
                    muA += va * weight;
                    muB += vb * weight;
                    ki++;
                }
            }
            ki = 0;
            for (int wy = -half; wy < half; wy++) {
                for (int wx = -half; wx < half; wx++) {
                    size_t idx = (size_t)(y + wy) * w + (x + wx);
                    double weight = kernel[ki];
                    double da = lumA[idx] - muA;
                    double db = lumB[idx] - muB;
                    sigAA += da * da * weight;
                    sigBB += db * db * weight;
                    sigAB += da * db * weight;
                    ki++;
                }
            }
            double num = (2 * muA * muB + SSIM_C1) * (2 * sigAB + SSIM_C2);
            double den = (muA * muA + muB * muB + SSIM_C1) * (sigAA + sigBB + SSIM_C2);
            localSum += num / den;
            localCount++;
        }
    }
    *sum_out = localSum;
    *count_out = localCount;
}

typedef struct {
    const double *lumA, *lumB, *kernel;
    int w, h, rowsPerProc;
    double *sums;
    long *counts;
} ssim_mt_arg;

static void ssim_mt_body(int from, int to, void *p)
{
    ssim_mt_arg *a = (ssim_mt_arg *)p;
    for (int proc = from; proc < to; proc++) {
        int startY = 4 + proc * a->rowsPerProc;
        int endY = startY + a->rowsPerProc;
        if (endY > a->h - 4) endY = a->h - 4;
        ssim_band(a->lumA, a->lumB, a->w, startY, endY, a->kernel, &a->sums[proc], &a->counts[proc]);
    }
}

/*
 * windowedSSIM (ssim.go:73-166).  `procs` plays GOMAXPROCS: the reference
 * splits rows into `procs` bands, keeps one partial (sum,count) per band and
 * adds the partials serially (ssim.go:84-94,155-160), so its last bits depend
 * on GOMAXPROCS.  procs=1 is the canonical oracle order.
 * The 8x8 window table is an input (see file header).
 */
ORC_API double orc_windowed_ssim(const double *lumA, const double *lumB, int w, int h,
                                 const double *kernel, int procs)
{
    const int windowSize = 8;
    int rows = h - windowSize + 1;
    if (procs > rows) procs = rows;
    if (procs < 1) procs = 1;
    int rowsPerProc = (rows + procs - 1) / procs;
    double *sums = (double *)calloc((size_t)procs, sizeof(double));
    long *counts = (long *)calloc((size_t)procs, sizeof(long));
    ssim_mt_arg a = {lumA, lumB, kernel, w, h, rowsPerProc, sums, counts};
    /* one goroutine per band (ssim.go:97-152) */
    parallel_do(0, procs, procs, ssim_mt_body, &a);
    double totalSum = 0;
    long totalCount = 0;
    for (int p = 0; p < procs; p++) {
        totalSum += sums[p];
        totalCount += counts[p];
    }
    free(sums);
    free(counts);
    if (totalCount == 0) return 1.0;
    return totalSum / (double)totalCount;
}

/*
 * pixelSSIM (ssim.go:169-204).  Walks the flat Pix slices; `pix_len` is
 * len(a.Pix) (4*w*h for a fresh image).
 */
ORC_API double orc_pixel_ssim(const uint8_t *a, const uint8_t *b, int w, int h, size_t pix_len)
{
    double n = (double)(w * h);
    if (n == 0) return 1.0;
    double muA = 0, muB = 0;
    for (size_t i = 0; i < pix_len; i += 4) {
        double la = 0.299 * (double)a[i] + 0.587 * (double)a[i + 1] + 0.114 * (double)a[i + 2];
        double lb = 0.299 * (double)b[i] + 0.587 * (double)b[i + 1] + 0.114 * (double)b[i + 2];
        muA += la;
        muB += lb;
    }
    muA /= n;
    muB /= n;
    double sigAA = 0, sigBB = 0, sigAB = 0;
    for (size_t i = 0; i < pix_len; i += 4) {
        double la = 0.299 * (double)a[i] + 0.587 * (double)a[i + 1] + 0.114 * (double)a[i + 2];
        double lb = 0.299 * (double)b[i] + 0.587 * (double)b[i + 1] + 0.114 * (double)b[i + 2];
        double da = la - muA, db = lb - muB;
        sigAA += da * da;
        sigBB += db * db;
        sigAB += da * db;
    }
    sigAA /= n;
    sigBB /= n;
    sigAB /= n;
    double num = (2 * muA * muB + SSIM_C1) * (2 * sigAB + SSIM_C2);
    double den = (muA * muA + muB * muB + SSIM_C1) * (sigAA + sigBB + SSIM_C2);
    return num / den;
}

/* boxDownsample + averageBoxPixel (ssim.go:244-309).  dst must be zeroed by
 * the caller (image.NewNRGBA); count==0 pixels are left untouched. */
ORC_API void orc_box_downsample(const uint8_t *src, int sstride, int srcW, int srcH,
                                uint8_t *dst, int dstride, int dstW, int dstH)
{
    if (srcW <= 0 || srcH <= 0 || dstW <= 0 || dstH <= 0) return; /* 0x0 image */
    double xRatio = (double)srcW / (double)dstW;
    double yRatio = (double)srcH / (double)dstH;
    for (int dy = 0; dy < dstH; dy++) {
        int sy0 = (int)((double)dy * yRatio);
        int sy1 = (int)((double)(dy + 1) * yRatio);
        if (sy1 > srcH) sy1 = srcH;
        if (sy0 >= sy1) sy0 = sy1 - 1;
        if (sy0 < 0) sy0 = 0;
        for (int dx = 0; dx < dstW; dx++) {
            int sx0 = (int)((double)dx * xRatio);
            int sx1 = (int)((double)(dx + 1) * xRatio);
            if (sx1 > srcW) sx1 = srcW;
            if (sx0 >= sx1) sx0 = sx1 - 1;
            if (sx0 < 0) sx0 = 0;
            double rS = 0, gS = 0, bS = 0, aS = 0, count = 0;
            for (int sy = sy0; sy < sy1; sy++) {
                for (int sx = sx0; sx < sx1; sx++) {
                    size_t off = (size_t)sy * sstride + (size_t)sx * 4;
                    rS += (double)src[off];
                    gS += (double)src[off + 1];
                    bS += (double)src[off + 2];
                    aS += (double)src[off + 3];
                    count++;
                }
            }
            if (count > 0) {
                double inv = 1.0 / count;
                size_t off = (size_t)dy * dstride + (size_t)dx * 4;
                dst[off] = clampF(rS * inv);
                dst[off + 1] = clampF(gS * inv);
                dst[off + 2] = clampF(bS * inv);
                dst[off + 3] = clampF(aS * inv);
            }
        }
    }
}

/* SSIMFast's target dims (ssim.go:52-56).  Returns 1 if a downsample happens. */
ORC_API int orc_ssim_fast_dims(int w, int h, int *newW, int *newH)
{
    const int maxDim = 512;
    *newW = w;
    *newH = h;
    if (w > maxDim || h > maxDim) {
        double scale = (double)maxDim / fmax((double)w, (double)h);
        *newW = (int)fmax(8, round((double)w * scale));
        *newH = (int)fmax(8, round((double)h * scale));
        return 1;
    }
    return 0;
}

/*
 * SSIMFast (ssim.go:48-70).  Both images are assumed w x h (the reference does
 * not check).  `procs` = GOMAXPROCS for windowedSSIM; boxDownsample and
 * toLuminance are serial in the reference and stay serial here.
 */
ORC_API double orc_ssim_fast(const uint8_t *a, int astride, const uint8_t *b, int bstride,
                             int w, int h, const double *kernel, int procs)
{
    uint8_t *da = NULL, *db = NULL;
    int nw, nh;
    if (orc_ssim_fast_dims(w, h, &nw, &nh)) {
        da = (uint8_t *)calloc((size_t)nw * nh * 4, 1);
        db = (uint8_t *)calloc((size_t)nw * nh * 4, 1);
        orc_box_downsample(a, astride, w, h, da, nw * 4, nw, nh);
        orc_box_downsample(b, bstride, w, h, db, nw * 4, nw, nh);
        a = da; astride = nw * 4;
        b = db; bstride = nw * 4;
        w = nw; h = nh;
    }
    double r;
    if (w < 8 || h < 8) {
        /* len(Pix) of the image handed to pixelSSIM */
        size_t len = (size_t)(h > 0 ? (h - 1) : 0) * (size_t)astride + (size_t)w * 4;
        if (w <= 0 || h <= 0) len = 0;
        r = orc_pixel_ssim(a, b, w, h, len);
    } else {
        double *lumA = (double *)malloc(sizeof(double) * (size_t)w * h);
        double *lumB = (double *)malloc(sizeof(double) * (size_t)w * h);
        orc_to_luminance(a, astride, w, h, lumA);
        orc_to_luminance(b, bstride, w, h, lumB);
        r = orc_windowed_ssim(lumA, lumB, w, h, kernel, procs);
        free(lumA);
        free(lumB);
    }
    free(da);
    free(db);
    return r;
}

/* ------------------------------------------------------------------ */
/* resize.go                                                           */
/* ------------------------------------------------------------------ */

/* lanczosKernel (resize.go:57-69), a = 3 */
ORC_API double orc_lanczos_kernel(double x)
{
    const double lanczosA = 3.0;
    if (x == 0) return 1.0;
    if (x < 0) x = -x;
    if (x >= lanczosA) return 0.0;
    double xpi = x * M_PI;
    return (lanczosA * sin(xpi) * sin(xpi / lanczosA)) / (xpi * xpi);
}

/*
 * precomputeWeights (resize.go:164-197) in CSR form: taps of output d are
 * index[offset[d] .. offset[d+1]) / weight[...].  Call with index==NULL to
 * size the arrays (returns total taps); offset must hold dstSize+1 ints.
 * ratio/support are derived as resizeH/resizeV do (resize.go:81-87,125-131).
 */
ORC_API int orc_precompute_weights(int dstSize, int srcSize, int *offset, int *index, double *weight)
{
    const double lanczosA = 3.0;
    double ratio = (double)srcSize / (double)dstSize;
    double support = lanczosA;
    if (ratio > 1) support = lanczosA * ratio;
    double filterScale = fmax(ratio, 1.0);
    int total = 0;
    for (int d = 0; d < dstSize; d++) {
        double center = ((double)d + 0.5) * ratio - 0.5;
        int left = (int)ceil(center - support);
        int right = (int)floor(center + support);
        if (left < 0) left = 0;
        if (right >= srcSize) right = srcSize - 1;
        double wsum = 0;
        int first = total;
        if (offset) offset[d] = total;
        for (int s = left; s <= right; s++) {
            double w = orc_lanczos_kernel(((double)s - center) / filterScale);
            if (w != 0) {
                wsum += w;
                if (index) {
                    index[total] = s;
                    weight[total] = w;
                }
                total++;
            }
        }
        if (wsum != 0 && index) {
            for (int i = first; i < total; i++) weight[i] /= wsum;
        }
    }
    if (offset) offset[dstSize] = total;
    return total;
}

typedef struct {
    const uint8_t *src;
    int sstride;
    uint8_t *dst;
    int dstride;
    int dstW, dstH;
    const int *offset, *index;
    const double *weight;
} resize_arg;

/* resizeH body (resize.go:89-115): parallel over rows y */
static void resize_h_rows(int from, int to, void *p)
{
    resize_arg *a = (resize_arg *)p;
    for (int y = from; y < to; y++) {
        for (int dx = 0; dx < a->dstW; dx++) {
            double r = 0, g = 0, b = 0, al = 0;
            for (int t = a->offset[dx]; t < a->offset[dx + 1]; t++) {
                size_t off = (size_t)y * a->sstride + (size_t)a->index[t] * 4;
                double sa = (double)a->src[off + 3];
                double w = a->weight[t];
                double aw = sa * w;
                r += (double)a->src[off] * aw;
                g += (double)a->src[off + 1] * aw;
                b += (double)a->src[off + 2] * aw;
                al += aw;
            }
            size_t dstOff = (size_t)y * a->dstride + (size_t)dx * 4;
            if (al > 0.5) {
                double inv = 1.0 / al;
                a->dst[dstOff] = clampF(r * inv);
                a->dst[dstOff + 1] = clampF(g * inv);
                a->dst[dstOff + 2] = clampF(b * inv);
                a->dst[dstOff + 3] = clampF(al);
            }
        }
    }
}

/* resizeV body (resize.go:133-158): parallel over columns x */
static void resize_v_cols(int from, int to, void *p)
{
    resize_arg *a = (resize_arg *)p;
    for (int x = from; x < to; x++) {
        for (int dy = 0; dy < a->dstH; dy++) {
            double r = 0, g = 0, b = 0, al = 0;
            for (int t = a->offset[dy]; t < a->offset[dy + 1]; t++) {
                size_t off = (size_t)a->index[t] * a->sstride + (size_t)x * 4;
                double sa = (double)a->src[off + 3];
                double w = a->weight[t];
                double aw = sa * w;
                r += (double)a->src[off] * aw;
                g += (double)a->src[off + 1] * aw;
                b += (double)a->src[off + 2] * aw;
                al += aw;
            }
            size_t dstOff = (size_t)dy * a->dstride + (size_t)x * 4;
            if (al > 0.5) {
                double inv = 1.0 / al;
                a->dst[dstOff] = clampF(r * inv);
                a->dst[dstOff + 1] = clampF(g * inv);
                a->dst[dstOff + 2] = clampF(b * inv);
                a->dst[dstOff + 3] = clampF(al);
            }
        }
    }
}

/* resizeH (resize.go:77-118) with an explicit tap table; dst (dstW x srcH) must be zeroed. */
ORC_API void orc_resize_h(const uint8_t *src, int sstride, int srcH, uint8_t *dst, int dstride,
                          int dstW, const int *offset, const int *index, const double *weight,
                          int procs)
{
    resize_arg a = {src, sstride, dst, dstride, dstW, srcH, offset, index, weight};
    parallel_do(0, srcH, procs, resize_h_rows, &a);
}

/* resizeV (resize.go:121-161) with an explicit tap table; dst (dstW x dstH) must be zeroed. */
ORC_API void orc_resize_v(const uint8_t *src, int sstride, uint8_t *dst, int dstride, int dstW,
                          int dstH, const int *offset, const int *index, const double *weight,
                          int procs)
{
    resize_arg a = {src, sstride, dst, dstride, dstW, dstH, offset, index, weight};
    parallel_do(0, dstW, procs, resize_v_cols, &a);
}

/*
 * lanczosResize (resize.go:37-53).  dst is dstW x dstH, tight or not, zeroed.
 * Returns 0 when the reference would hand back a 0x0 image (any dim <= 0).
 * Equal dims: flat copy of Pix (resize.go:45-49).
 */
ORC_API int orc_lanczos_resize(const uint8_t *src, int sstride, int srcW, int srcH, uint8_t *dst,
                               int dstride, int dstW, int dstH, int procs)
{
    if (srcW <= 0 || srcH <= 0 || dstW <= 0 || dstH <= 0) return 0;
    if (srcW == dstW && srcH == dstH) {
        size_t slen = (size_t)(srcH - 1) * sstride + (size_t)srcW * 4;
        size_t dlen = (size_t)(dstH - 1) * dstride + (size_t)dstW * 4;
        memcpy(dst, src, slen < dlen ? slen : dlen);
        return 1;
    }
    int *offH = (int *)malloc(sizeof(int) * ((size_t)dstW + 1));
    int nH = orc_precompute_weights(dstW, srcW, offH, NULL, NULL);
    int *idxH = (int *)malloc(sizeof(int) * (size_t)(nH > 0 ? nH : 1));
    double *wH = (double *)malloc(sizeof(double) * (size_t)(nH > 0 ? nH : 1));
    orc_precompute_weights(dstW, srcW, offH, idxH, wH);
    int *offV = (int *)malloc(sizeof(int) * ((size_t)dstH + 1));
    int nV = orc_precompute_weights(dstH, srcH, offV, NULL, NULL);
    int *idxV = (int *)malloc(sizeof(int) * (size_t)(nV > 0 ? nV : 1));
    double *wV = (double *)malloc(sizeof(double) * (size_t)(nV > 0 ? nV : 1));
    orc_precompute_weights(dstH, srcH, offV, idxV, wV);

    uint8_t *tmp = (uint8_t *)calloc((size_t)dstW * srcH * 4, 1);
    orc_resize_h(src, sstride, srcH, tmp, dstW * 4, dstW, offH, idxH, wH, procs);
    orc_resize_v(tmp, dstW * 4, dst, dstride, dstW, dstH, offV, idxV, wV, procs);
    free(tmp);
    free(offH); free(idxH); free(wH);
    free(offV); free(idxV); free(wV);
    return 1;
}

/*
 * smartResize's target dims (resize.go:12-32).  Returns 0 when the image
 * already fits (the reference returns the SAME pointer), 1 otherwise.
 */
ORC_API int orc_smart_resize_dims(int srcW, int srcH, int maxW, int maxH, int *dstW, int *dstH)
{
    if (maxW <= 0) maxW = srcW;
    if (maxH <= 0) maxH = srcH;
    *dstW = srcW;
    *dstH = srcH;
    if (srcW <= maxW && srcH <= maxH) return 0;
    double ratio = fmin((double)maxW / (double)srcW, (double)maxH / (double)srcH);
    *dstW = (int)fmax(1, round((double)srcW * ratio));
    *dstH = (int)fmax(1, round((double)srcH * ratio));
    return 1;
}

/* ------------------------------------------------------------------ */
/* SSIM / MSSSIM (need lanczosResize)                                  */
/* ------------------------------------------------------------------ */

/* SSIM (ssim.go:24-43) */
ORC_API double orc_ssim(const uint8_t *a, int astride, int aw, int ah, const uint8_t *b,
                        int bstride, int bw, int bh, const double *kernel, int procs)
{
    uint8_t *rb = NULL;
    int w = aw, h = ah;
    if (w != bw || h != bh) {
        if (w > 0 && h > 0 && bw > 0 && bh > 0) {
            rb = (uint8_t *)calloc((size_t)w * h * 4, 1);
            orc_lanczos_resize(b, bstride, bw, bh, rb, w * 4, w, h, procs);
            b = rb;
            bstride = w * 4;
        }
    }
    double r;
    if (w < 8 || h < 8) {
        size_t len = (w > 0 && h > 0) ? (size_t)(h - 1) * astride + (size_t)w * 4 : 0;
        r = orc_pixel_ssim(a, b, w, h, len);
    } else {
        double *lumA = (double *)malloc(sizeof(double) * (size_t)w * h);
        double *lumB = (double *)malloc(sizeof(double) * (size_t)w * h);
        orc_to_luminance(a, astride, w, h, lumA);
        orc_to_luminance(b, bstride, w, h, lumB);
        r = orc_windowed_ssim(lumA, lumB, w, h, kernel, procs);
        free(lumA);
        free(lumB);
    }
    free(rb);
    return r;
}

/* MSSSIM (ssim.go:313-365).  per_level (optional, 5 doubles) receives each
 * level's SSIMFast value, NaN for levels not evaluated. */
ORC_API double orc_msssim(const uint8_t *a, int astride, int aw, int ah, const uint8_t *b,
                          int bstride, int bw, int bh, const double *kernel, int procs,
                          double *per_level)
{
    int w = aw, h = ah;
    uint8_t *rb = NULL;
    if (w != bw || h != bh) {
        rb = (uint8_t *)calloc((size_t)(w > 0 ? w : 0) * (h > 0 ? h : 0) * 4 + 4, 1);
        orc_lanczos_resize(b, bstride, bw, bh, rb, w * 4, w, h, procs);
        b = rb;
        bstride = w * 4;
    }
    double weights[5] = {0.0448, 0.2856, 0.3001, 0.2363, 0.1333};
    int levels = 5, nweights = 5;
    {
        int tw = w, th = h;
        for (int i = 0; i < levels - 1; i++) {
            int minDim = (int)fmin((double)tw, (double)th);
            if (minDim < 8) {
                nweights = i + 1;
                double sum = 0;
                for (int j = 0; j < nweights; j++) sum += weights[j];
                for (int j = 0; j < nweights; j++) weights[j] /= sum;
                break;
            }
            tw /= 2;
            th /= 2;
        }
    }
    /* toNRGBA copies (convert.go:12-19, called at ssim.go:345-346): tight w x h images filled by
     * copy(dst.Pix, nrgba.Pix) -- the FIRST 4wh flat bytes of each source slice (see flat_pix_copy above),
     * so a SubImage (stride != 4w) enters the pyramid with its row padding folded in, exactly as in Go */
    size_t n0 = (size_t)(w > 0 ? w : 0) * (size_t)(h > 0 ? h : 0) * 4;
    uint8_t *ac = (uint8_t *)malloc(n0 + 4), *bc = (uint8_t *)malloc(n0 + 4);
    (void)astride;
    (void)bstride;
    if (n0) {
        memcpy(ac, a, n0);
        memcpy(bc, b, n0);
    }
    int cw = w, ch = h;
    double result = 0;
    if (per_level) for (int i = 0; i < 5; i++) per_level[i] = NAN;
    for (int i = 0; i < nweights; i++) {
        double ssim = orc_ssim_fast(ac, cw * 4, bc, cw * 4, cw, ch, kernel, procs);
        if (per_level) per_level[i] = ssim;
        result += weights[i] * log(fmax(ssim, 1e-10));
        if (i < nweights - 1) {
            int nw = cw / 2, nh = ch / 2;
            if (nw < 8 || nh < 8) break;
            uint8_t *na = (uint8_t *)calloc((size_t)nw * nh * 4, 1);
            uint8_t *nb = (uint8_t *)calloc((size_t)nw * nh * 4, 1);
            orc_box_downsample(ac, cw * 4, cw, ch, na, nw * 4, nw, nh);
            orc_box_downsample(bc, cw * 4, cw, ch, nb, nw * 4, nw, nh);
            free(ac);
            free(bc);
            ac = na; bc = nb; cw = nw; ch = nh;
        }
    }
    free(ac);
    free(bc);
    free(rb);
    return exp(result);
}

/* ------------------------------------------------------------------ */
/* effects.go                                                          */
/* ------------------------------------------------------------------ */

/* GaussianBlur's 1-D kernel (effects.go:153-165).  Returns radius; kernel
 * must hold 2*radius+1 doubles (call with kernel==NULL to get the radius). */
ORC_API int orc_blur_kernel(double sigma, double *kernel)
{
    int radius = (int)ceil(sigma * 3);
    if (!kernel) return radius;
    int kernelSize = radius * 2 + 1;
    double sum = 0;
    for (int i = 0; i < kernelSize; i++) {
        double x = (double)(i - radius);
        kernel[i] = exp(-(x * x) / (2 * sigma * sigma));
        sum += kernel[i];
    }
    for (int i = 0; i < kernelSize; i++) kernel[i] /= sum;
    return radius;
}

typedef struct {
    const uint8_t *src;   /* pass input */
    int sstride;
    const uint8_t *alpha; /* image alpha is copied from */
    int astride;
    uint8_t *dst;
    int dstride;
    int w, h, radius;
    const double *kernel;
} blur_arg;

/* horizontal pass (effects.go:169-191): parallel over rows */
static void blur_h_rows(int from, int to, void *p)
{
    blur_arg *a = (blur_arg *)p;
    int kernelSize = a->radius * 2 + 1;
    for (int y = from; y < to; y++) {
        for (int x = 0; x < a->w; x++) {
            double r = 0, g = 0, b = 0;
            for (int k = 0; k < kernelSize; k++) {
                int sx = x + k - a->radius;
                if (sx < 0) sx = 0;
                else if (sx >= a->w) sx = a->w - 1;
                size_t off = (size_t)y * a->sstride + (size_t)sx * 4;
                double wt = a->kernel[k];
                r += (double)a->src[off] * wt;
                g += (double)a->src[off + 1] * wt;
                b += (double)a->src[off + 2] * wt;
            }
            size_t off = (size_t)y * a->dstride + (size_t)x * 4;
            a->dst[off] = clampF(r);
            a->dst[off + 1] = clampF(g);
            a->dst[off + 2] = clampF(b);
            a->dst[off + 3] = a->alpha[(size_t)y * a->astride + (size_t)x * 4 + 3];
        }
    }
}

/* vertical pass (effects.go:195-217): parallel over columns */
static void blur_v_cols(int from, int to, void *p)
{
    blur_arg *a = (blur_arg *)p;
    int kernelSize = a->radius * 2 + 1;
    for (int x = from; x < to; x++) {
        for (int y = 0; y < a->h; y++) {
            double r = 0, g = 0, b = 0;
            for (int k = 0; k < kernelSize; k++) {
                int sy = y + k - a->radius;
                if (sy < 0) sy = 0;
                else if (sy >= a->h) sy = a->h - 1;
                size_t off = (size_t)sy * a->sstride + (size_t)x * 4;
                double wt = a->kernel[k];
                r += (double)a->src[off] * wt;
                g += (double)a->src[off + 1] * wt;
                b += (double)a->src[off + 2] * wt;
            }
            size_t off = (size_t)y * a->dstride + (size_t)x * 4;
            a->dst[off] = clampF(r);
            a->dst[off + 1] = clampF(g);
            a->dst[off + 2] = clampF(b);
            a->dst[off + 3] = a->alpha[(size_t)y * a->astride + (size_t)x * 4 + 3];
        }
    }
}

/*
 * GaussianBlur (effects.go:146-220) with the 1-D kernel as an input.
 * The sigma<=0 "same pointer" guard (effects.go:147-149) belongs to the caller.
 * The intermediate image is rounded to uint8 (effects.go:186-188).
 */
ORC_API void orc_gaussian_blur_k(const uint8_t *src, int sstride, int w, int h,
                                 const double *kernel, int radius, uint8_t *dst, int dstride,
                                 int procs)
{
    if (w <= 0 || h <= 0) return;
    uint8_t *tmp = (uint8_t *)malloc((size_t)w * h * 4);
    blur_arg ha = {src, sstride, src, sstride, tmp, w * 4, w, h, radius, kernel};
    parallel_do(0, h, procs, blur_h_rows, &ha);
    blur_arg va = {tmp, w * 4, src, sstride, dst, dstride, w, h, radius, kernel};
    parallel_do(0, w, procs, blur_v_cols, &va);
    free(tmp);
}

ORC_API void orc_gaussian_blur(const uint8_t *src, int sstride, int w, int h, double sigma,
                               uint8_t *dst, int dstride, int procs)
{
    int radius = orc_blur_kernel(sigma, NULL);
    double *kernel = (double *)malloc(sizeof(double) * (size_t)(2 * radius + 1));
    orc_blur_kernel(sigma, kernel);
    orc_gaussian_blur_k(src, sstride, w, h, kernel, radius, dst, dstride, procs);
    free(kernel);
}

typedef struct {
    const uint8_t *src;
    int sstride;
    const uint8_t *blur;
    int bstride;
    uint8_t *dst;
    int dstride;
    int w, h;
    double amount;
} fx_arg;

/* gaussianBlur3x3 interior rows (effects.go:122-139) */
static void blur3_rows(int from, int to, void *p)
{
    fx_arg *a = (fx_arg *)p;
    const uint8_t *s = a->src;
    int st = a->sstride;
    for (int y = from; y < to; y++) {
        for (int x = 1; x < a->w - 1; x++) {
            for (int c = 0; c < 3; c++) {
                double sum = 0;
                sum += (double)s[(size_t)(y - 1) * st + (size_t)(x - 1) * 4 + c] * 1;
                sum += (double)s[(size_t)(y - 1) * st + (size_t)(x)*4 + c] * 2;
                sum += (double)s[(size_t)(y - 1) * st + (size_t)(x + 1) * 4 + c] * 1;
                sum += (double)s[(size_t)(y)*st + (size_t)(x - 1) * 4 + c] * 2;
                sum += (double)s[(size_t)(y)*st + (size_t)(x)*4 + c] * 4;
                sum += (double)s[(size_t)(y)*st + (size_t)(x + 1) * 4 + c] * 2;
                sum += (double)s[(size_t)(y + 1) * st + (size_t)(x - 1) * 4 + c] * 1;
                sum += (double)s[(size_t)(y + 1) * st + (size_t)(x)*4 + c] * 2;
                sum += (double)s[(size_t)(y + 1) * st + (size_t)(x + 1) * 4 + c] * 1;
                a->dst[(size_t)y * a->dstride + (size_t)x * 4 + c] = clampF(sum / 16.0);
            }
        }
    }
}

/*
 * copy(dst.Pix, img.Pix) (effects.go:68,120; convert.go:16): a FLAT copy of min(len(dst.Pix), len(img.Pix))
 * bytes, not a row-by-row one.  dst is always a fresh tight w x h image (len 4wh); img.Pix of any valid
 * *image.NRGBA holds at least (h-1)*Stride + 4w >= 4wh bytes (Stride >= 4w: NewNRGBA's 4w, or a SubImage's
 * parent stride), so exactly the FIRST 4wh bytes of img.Pix move -- for a SubImage (Stride != 4w) those are
 * not its rows but its first row, the parent's bytes behind it, the second row ...  Row y of the tight
 * destination is flat bytes [4wy, 4w(y+1)) of the source slice; dstride only places that row.
 */
static void flat_pix_copy(const uint8_t *src_pix, int w, int h, uint8_t *dst, int dstride)
{
    for (int y = 0; y < h; y++)
        memcpy(dst + (size_t)y * dstride, src_pix + (size_t)y * w * 4, (size_t)w * 4);
}

/* gaussianBlur3x3 (effects.go:116-141): dst = copy(dst.Pix, img.Pix) -- flat, see above -- then the interior's
 * R, G, B blurred from the STRIDED source; the interior's alpha and the whole border keep the flat bytes */
ORC_API void orc_blur3x3(const uint8_t *src, int sstride, int w, int h, uint8_t *dst, int dstride,
                         int procs)
{
    flat_pix_copy(src, w, h, dst, dstride);                                    /* effects.go:120 */
    fx_arg a = {src, sstride, NULL, 0, dst, dstride, w, h, 0};
    parallel_do(1, h - 1, procs, blur3_rows, &a);
}

/* Sharpen body (effects.go:28-42) */
static void sharpen_rows(int from, int to, void *p)
{
    fx_arg *a = (fx_arg *)p;
    for (int y = from; y < to; y++) {
        for (int x = 0; x < a->w; x++) {
            size_t so = (size_t)y * a->sstride + (size_t)x * 4;
            size_t bo = (size_t)y * a->bstride + (size_t)x * 4;
            size_t dof = (size_t)y * a->dstride + (size_t)x * 4;
            for (int c = 0; c < 3; c++) {
                double orig = (double)a->src[so + c];
                double blur = (double)a->blur[bo + c];
                double val = orig + a->amount * (orig - blur);
                a->dst[dof + c] = clampF(val);
            }
            a->dst[dof + 3] = a->src[so + 3];
        }
    }
}

/*
 * Sharpen (effects.go:10-45).  Returns 0 when the reference returns the SAME
 * pointer (strength<=0, or w<3 || h<3) and leaves dst untouched; 1 otherwise.
 */
ORC_API int orc_sharpen(const uint8_t *src, int sstride, int w, int h, double strength,
                        uint8_t *dst, int dstride, int procs)
{
    if (strength <= 0) return 0;
    if (strength > 1) strength = 1;
    if (w < 3 || h < 3) return 0;
    uint8_t *blurred = (uint8_t *)malloc((size_t)w * h * 4);
    orc_blur3x3(src, sstride, w, h, blurred, w * 4, procs);
    fx_arg a = {src, sstride, blurred, w * 4, dst, dstride, w, h, 1.0 + strength * 1.5};
    parallel_do(0, h, procs, sharpen_rows, &a);
    free(blurred);
    return 1;
}

/* localEdgeStrength (effects.go:93-112) */
static double local_edge_strength(const uint8_t *pix, int stride, int x, int y)
{
#define LUM(px, py) \
    (0.299 * (double)pix[(size_t)(py)*stride + (size_t)(px)*4] + \
     0.587 * (double)pix[(size_t)(py)*stride + (size_t)(px)*4 + 1] + \
     0.114 * (double)pix[(size_t)(py)*stride + (size_t)(px)*4 + 2])
    double gx = -LUM(x - 1, y - 1) + LUM(x + 1, y - 1) - 2 * LUM(x - 1, y) + 2 * LUM(x + 1, y) -
                LUM(x - 1, y + 1) + LUM(x + 1, y + 1);
    double gy = -LUM(x - 1, y - 1) - 2 * LUM(x, y - 1) - LUM(x + 1, y - 1) + LUM(x - 1, y + 1) +
                2 * LUM(x, y + 1) + LUM(x + 1, y + 1);
#undef LUM
    double mag = sqrt(gx * gx + gy * gy);
    double normalized = mag / 400.0;
    if (normalized > 1) normalized = 1;
    return normalized;
}

/* AdaptiveSharpen interior rows (effects.go:70-87) */
static void adaptive_rows(int from, int to, void *p)
{
    fx_arg *a = (fx_arg *)p;
    for (int y = from; y < to; y++) {
        for (int x = 1; x < a->w - 1; x++) {
            size_t so = (size_t)y * a->sstride + (size_t)x * 4;
            double edgeStr = local_edge_strength(a->src, a->sstride, x, y);
            double localAmount = a->amount * edgeStr;
            size_t bo = (size_t)y * a->bstride + (size_t)x * 4;
            size_t dof = (size_t)y * a->dstride + (size_t)x * 4;
            for (int c = 0; c < 3; c++) {
                double orig = (double)a->src[so + c];
                double blur = (double)a->blur[bo + c];
                double val = orig + localAmount * (orig - blur);
                a->dst[dof + c] = clampF(val);
            }
            a->dst[dof + 3] = a->src[so + 3];
        }
    }
}

/* AdaptiveSharpen (effects.go:49-90).  Return value as orc_sharpen. */
ORC_API int orc_adaptive_sharpen(const uint8_t *src, int sstride, int w, int h, double strength,
                                 uint8_t *dst, int dstride, int procs)
{
    if (strength <= 0) return 0;
    if (strength > 1) strength = 1;
    if (w < 3 || h < 3) return 0;
    uint8_t *blurred = (uint8_t *)malloc((size_t)w * h * 4);
    orc_blur3x3(src, sstride, w, h, blurred, w * 4, procs);
    flat_pix_copy(src, w, h, dst, dstride);                                    /* effects.go:68: flat */
    fx_arg a = {src, sstride, blurred, w * 4, dst, dstride, w, h, 1.0 + strength * 2.0};
    parallel_do(1, h - 1, procs, adaptive_rows, &a);
    free(blurred);
    return 1;
}

/* ------------------------------------------------------------------ */
/* orientation (convert.go:186-256, exif.go:178-203)                   */
/* ------------------------------------------------------------------ */

static void rot90cw(const uint8_t *s, int ss, int w, int h, uint8_t *d, int ds)
{ /* convert.go:187-199: dst is h wide, w high */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            memcpy(d + (size_t)x * ds + (size_t)(h - 1 - y) * 4, s + (size_t)y * ss + (size_t)x * 4, 4);
}
static void rot180(const uint8_t *s, int ss, int w, int h, uint8_t *d, int ds)
{ /* convert.go:202-214 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            memcpy(d + (size_t)(h - 1 - y) * ds + (size_t)(w - 1 - x) * 4, s + (size_t)y * ss + (size_t)x * 4, 4);
}
static void rot270cw(const uint8_t *s, int ss, int w, int h, uint8_t *d, int ds)
{ /* convert.go:217-229: dst is h wide, w high */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            memcpy(d + (size_t)(w - 1 - x) * ds + (size_t)y * 4, s + (size_t)y * ss + (size_t)x * 4, 4);
}
static void flip_h(const uint8_t *s, int ss, int w, int h, uint8_t *d, int ds)
{ /* convert.go:232-244 */
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            memcpy(d + (size_t)y * ds + (size_t)(w - 1 - x) * 4, s + (size_t)y * ss + (size_t)x * 4, 4);
}
static void flip_v(const uint8_t *s, int ss, int w, int h, uint8_t *d, int ds)
{ /* convert.go:247-256 */
    for (int y = 0; y < h; y++)
        memcpy(d + (size_t)(h - 1 - y) * ds, s + (size_t)y * ss, (size_t)w * 4);
}

/*
 * ApplyOrientation (exif.go:178-203).  Output dims: w x h for 2,3,4; h x w for
 * 5,6,7,8.  dst is tight (stride = 4*outW).  Returns 0 for orientations the
 * reference answers with the SAME pointer (0, 1, unknown), 1 otherwise.
 */
ORC_API int orc_apply_orientation(const uint8_t *src, int sstride, int w, int h, int orient,
                                  uint8_t *dst)
{
    switch (orient) {
    case 2: flip_h(src, sstride, w, h, dst, w * 4); return 1;
    case 3: rot180(src, sstride, w, h, dst, w * 4); return 1;
    case 4: flip_v(src, sstride, w, h, dst, w * 4); return 1;
    case 5: { /* rotate 270 CW, then flip horizontal */
        uint8_t *t = (uint8_t *)malloc((size_t)w * h * 4 + 4);
        rot270cw(src, sstride, w, h, t, h * 4);
        flip_h(t, h * 4, h, w, dst, h * 4);
        free(t);
        return 1;
    }
    case 6: rot90cw(src, sstride, w, h, dst, h * 4); return 1;
    case 7: { /* rotate 90 CW, then flip horizontal */
        uint8_t *t = (uint8_t *)malloc((size_t)w * h * 4 + 4);
        rot90cw(src, sstride, w, h, t, h * 4);
        flip_h(t, h * 4, h, w, dst, h * 4);
        free(t);
        return 1;
    }
    case 8: rot270cw(src, sstride, w, h, dst, h * 4); return 1;
    default: return 0;
    }
}

/* ------------------------------------------------------------------ */
/* batch.go:140-158 Summarize                                          */
/* ------------------------------------------------------------------ */

/*
 * Summarize (batch.go:140-158) over parallel arrays: failed[i]!=0 is r.Err!=nil,
 * has_result[i] is r.Result!=nil.  out = {Total, Succeeded, Failed, TotalSaved};
 * returns AvgSSIM.
 */
ORC_API double orc_summarize(int n, const int *failed, const int *has_result,
                             const int64_t *original_size, const int64_t *compressed_size,
                             const double *ssim, int64_t out[4])
{
    int64_t succeeded = 0, nfailed = 0, saved = 0;
    double ssimSum = 0;
    for (int i = 0; i < n; i++) {
        if (failed[i]) {
            nfailed++;
            continue;
        }
        succeeded++;
        if (has_result[i]) {
            saved += original_size[i] - compressed_size[i];
            ssimSum += ssim[i];
        }
    }
    out[0] = n;
    out[1] = succeeded;
    out[2] = nfailed;
    out[3] = saved;
    return succeeded > 0 ? ssimSum / (double)succeeded : 0.0;
}

/* ================================================================== */
/* Analyze (analyze.go:26-176) and the scans of convert.go:66-146      */
/* the first row widened after (a)-(e).            */
/* ================================================================== */

/* Insertion-only set of uint32 keys, standing in for Go's map[uint32]struct{}:
 * only len() is observable (analyze.go:73-76, convert.go:121-133). */
typedef struct {
    uint64_t *slot; /* 0 = empty, else key + 1 */
    int cap, len;
} orc_set;

static int orc_set_init(orc_set *s, int cap)
{
    s->slot = (uint64_t *)calloc((size_t)cap, sizeof(uint64_t));
    s->cap = cap;
    s->len = 0;
    return s->slot != NULL;
}

static void orc_set_add(orc_set *s, uint32_t key)
{
    uint32_t h = (key * 2654435761u) & (uint32_t)(s->cap - 1);
    while (s->slot[h]) {
        if (s->slot[h] == (uint64_t)key + 1) return;
        h = (h + 1) & (uint32_t)(s->cap - 1);
    }
    s->slot[h] = (uint64_t)key + 1;
    s->len++;
}

typedef struct {
    int32_t width, height;
    int32_t has_alpha, is_grayscale, unique_colors;
    int32_t recommended_format;   /* types.go:35-42: 0 Auto, 1 JPEG, 2 PNG */
    int32_t recommended_quality;  /* types.go:59-72: 0 Balanced, 1 Lossless, 2 Ultra, 3 High, 4 Aggressive, 5 Maximum */
    int32_t pad;
    double entropy, edge_density, mean_brightness, contrast, estimated_compression;
    /* raw accumulators, exposed so that tests can compare the device's integer outputs exactly */
    double histogram[256];
    double bright_sum, variance_sum;
    int64_t sample_count, edge_count, edge_total;
} orc_image_stats;

/* sobelLum (analyze.go:186-189) */
static double sobel_lum(const uint8_t *pix, int stride, int x, int y)
{
    const uint8_t *p = pix + (size_t)y * stride + (size_t)x * 4;
    return 0.299 * (double)p[0] + 0.587 * (double)p[1] + 0.114 * (double)p[2];
}

/* computeEntropy (analyze.go:127-139); log2 is libm's (Go: math.Log2) */
ORC_API double orc_compute_entropy(const double *histogram, int n, double total)
{
    if (total == 0) return 0;
    double entropy = 0;
    for (int i = 0; i < n; i++) {
        double count = histogram[i];
        if (count > 0) {
            double p = count / total;
            entropy -= p * log2(p);
        }
    }
    return entropy;
}

/* recommendFormat / recommendQuality / estimateCompression (analyze.go:191-230) */
ORC_API void orc_recommend(orc_image_stats *s)
{
    if (s->has_alpha) s->recommended_format = 2;
    else if (s->unique_colors <= 256) s->recommended_format = 2;
    else if (s->edge_density > 0.3 && s->unique_colors < 1000) s->recommended_format = 2;
    else s->recommended_format = 1;

    if (s->entropy > 6 && s->edge_density < 0.15) s->recommended_quality = 0;
    else if (s->entropy < 4) s->recommended_quality = 4;
    else if (s->edge_density > 0.25) s->recommended_quality = 3;
    else s->recommended_quality = 0;

    if (s->recommended_format == 2) {
        if (s->unique_colors <= 256) s->estimated_compression = 5.0 + (256 - (double)s->unique_colors) / 50;
        else if (s->is_grayscale) s->estimated_compression = 3.0;
        else s->estimated_compression = 2.0;
    } else {
        double base = 10.0;
        if (s->entropy > 7) base = 5.0;
        else if (s->entropy > 5) base = 8.0;
        if (s->edge_density > 0.2) base *= 0.7;
        s->estimated_compression = base;
    }
}

/* Analyze (analyze.go:26-124), computeEdgeDensity (analyze.go:142-184) */
ORC_API int orc_analyze(const uint8_t *pix, int stride, int w, int h, orc_image_stats *st)
{
    memset(st, 0, sizeof(*st));
    st->width = w;
    st->height = h;
    if (w == 0 || h == 0) return 0;

    /* single pass: colour info, brightness, alpha (analyze.go:41-85) */
    double brightSum = 0;
    orc_set colors;
    if (!orc_set_init(&colors, 4096)) return -1;
    int maxSample = 50000, step = 1;
    if ((long long)w * h > maxSample) step = (int)((long long)w * h / maxSample);
    int allGray = 1, hasAlpha = 0;
    long long idx = 0;
    for (int y = 0; y < h; y++) {
        size_t off = (size_t)y * stride;
        for (int x = 0; x < w; x++) {
            size_t i = off + (size_t)x * 4;
            uint8_t r = pix[i], g = pix[i + 1], b = pix[i + 2], a = pix[i + 3];
            double lum = 0.299 * (double)r + 0.587 * (double)g + 0.114 * (double)b;
            brightSum += lum;
            st->histogram[(int)(lum + 0.5)]++;
            if (a < 255) hasAlpha = 1;
            if (r != g || g != b) allGray = 0;
            if (idx % step == 0 && colors.len < 1024) {
                uint32_t key = (uint32_t)r << 24 | (uint32_t)g << 16 | (uint32_t)b << 8 | (uint32_t)a;
                orc_set_add(&colors, key);
            }
            idx++;
        }
    }
    double n = (double)((long long)w * h);
    st->has_alpha = hasAlpha;
    st->is_grayscale = allGray;
    st->unique_colors = colors.len;
    free(colors.slot);
    st->bright_sum = brightSum;
    st->mean_brightness = brightSum / n;

    /* contrast on a fixed grid (analyze.go:93-113) */
    int stepY = (int)fmax(1, ceil((double)h / 100));
    int stepX = (int)fmax(1, ceil((double)w / 100));
    double varianceSum = 0, mean = st->mean_brightness;
    int64_t sampleCount = 0;
    for (int y = 0; y < h; y += stepY) {
        size_t off = (size_t)y * stride;
        for (int x = 0; x < w; x += stepX) {
            size_t i = off + (size_t)x * 4;
            double lum = 0.299 * (double)pix[i] + 0.587 * (double)pix[i + 1] + 0.114 * (double)pix[i + 2];
            double d = lum - mean;
            varianceSum += d * d;
            sampleCount++;
        }
    }
    st->variance_sum = varianceSum;
    st->sample_count = sampleCount;
    if (sampleCount > 0) st->contrast = sqrt(varianceSum / (double)sampleCount);

    st->entropy = orc_compute_entropy(st->histogram, 256, n);

    /* computeEdgeDensity (analyze.go:142-184) */
    if (w >= 3 && h >= 3) {
        int sX = (int)fmax(1, (double)w / 200), sY = (int)fmax(1, (double)h / 200);
        int64_t edgeCount = 0, totalCount = 0;
        double threshold = 30.0;
        for (int y = 1; y < h - 1; y += sY) {
            for (int x = 1; x < w - 1; x += sX) {
                double gx = sobel_lum(pix, stride, x + 1, y - 1) - sobel_lum(pix, stride, x - 1, y - 1) +
                            2 * sobel_lum(pix, stride, x + 1, y) - 2 * sobel_lum(pix, stride, x - 1, y) +
                            sobel_lum(pix, stride, x + 1, y + 1) - sobel_lum(pix, stride, x - 1, y + 1);
                double gy = sobel_lum(pix, stride, x - 1, y + 1) - sobel_lum(pix, stride, x - 1, y - 1) +
                            2 * sobel_lum(pix, stride, x, y + 1) - 2 * sobel_lum(pix, stride, x, y - 1) +
                            sobel_lum(pix, stride, x + 1, y + 1) - sobel_lum(pix, stride, x + 1, y - 1);
                double mag = sqrt(gx * gx + gy * gy);
                if (mag > threshold) edgeCount++;
                totalCount++;
            }
        }
        st->edge_count = edgeCount;
        st->edge_total = totalCount;
        if (totalCount > 0) st->edge_density = (double)edgeCount / (double)totalCount;
    }
    orc_recommend(st);
    return 0;
}

/* isOpaque (convert.go:66-74): walks the FLAT Pix slice, row padding included */
ORC_API int orc_is_opaque(const uint8_t *pix, size_t pix_len)
{
    for (size_t i = 3; i < pix_len; i += 4)
        if (pix[i] != 0xff) return 0;
    return 1;
}

/* isGrayscale (convert.go:76-84): flat Pix as well */
ORC_API int orc_is_grayscale(const uint8_t *pix, size_t pix_len)
{
    for (size_t i = 0; i + 2 < pix_len; i += 4)
        if (pix[i] != pix[i + 1] || pix[i + 1] != pix[i + 2]) return 0;
    return 1;
}

/* analyzeFormat (convert.go:105-146): 1 JPEG, 2 PNG.  The scan stops once 512 distinct sampled
 * colours were seen, so has-alpha only covers the samples before that point. */
ORC_API int orc_analyze_format(const uint8_t *pix, int stride, int w, int h)
{
    int hasAlpha = 0;
    orc_set colors;
    if (!orc_set_init(&colors, 2048)) return -1;
    int maxSamples = 10000, step = 1;
    long long total = (long long)w * h;
    if (total > maxSamples) {
        step = (int)(total / maxSamples);
        if (step < 1) step = 1;
    }
    long long idx = 0;
    for (int y = 0; y < h && colors.len < 512; y++) {
        for (int x = 0; x < w && colors.len < 512; x++) {
            if (idx % step != 0) {
                idx++;
                continue;
            }
            size_t off = (size_t)y * stride + (size_t)x * 4;
            uint8_t a = pix[off + 3];
            if (a < 255) hasAlpha = 1;
            uint32_t key = (uint32_t)pix[off] << 24 | (uint32_t)pix[off + 1] << 16 | (uint32_t)pix[off + 2] << 8 | a;
            orc_set_add(&colors, key);
            idx++;
        }
    }
    int ncolors = colors.len;
    free(colors.slot);
    if (hasAlpha) return 2;
    if (ncolors < 256) return 2;
    return 1;
}

/* ================================================================== */
/* applyPalette + palettedToNRGBA (targetsize.go:488-546)              */
/* ================================================================== */

/*
 * applyPalette (targetsize.go:488-527).  palette: n x 4 bytes r,g,b,a, every a == 255 (the only
 * palettes medianCut builds, targetsize.go:407-410), so c.RGBA()>>8 is the stored byte.  The
 * reference memoises the answer per (r,g,b) in a map; the memo cannot change a result.
 * idx: w x h bytes (image.Paletted.Pix, stride istride); quant (may be NULL): palettedToNRGBA
 * (targetsize.go:529-546) of it.
 */
ORC_API void orc_apply_palette(const uint8_t *pix, int stride, int w, int h, const uint8_t *palette, int n,
                               uint8_t *idx, int istride, uint8_t *quant, int qstride)
{
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            size_t off = (size_t)y * stride + (size_t)x * 4;
            int r = pix[off], g = pix[off + 1], b = pix[off + 2];
            int bestIdx = 0, bestDist = 2147483647; /* math.MaxInt32 */
            for (int i = 0; i < n; i++) {
                int dr = r - (int)palette[4 * i], dg = g - (int)palette[4 * i + 1], db = b - (int)palette[4 * i + 2];
                int dist = dr * dr + dg * dg + db * db;
                if (dist < bestDist) {
                    bestDist = dist;
                    bestIdx = i;
                }
            }
            idx[(size_t)y * istride + x] = (uint8_t)bestIdx;
            if (quant) {
                uint8_t *q = quant + (size_t)y * qstride + (size_t)x * 4;
                q[0] = palette[4 * bestIdx];
                q[1] = palette[4 * bestIdx + 1];
                q[2] = palette[4 * bestIdx + 2];
                q[3] = 255;
            }
        }
    }
}

/* ================================================================== */
/* toNRGBARef of a decoded JPEG (convert.go:22-64) -- SURVEY 8(f).1     */
/* ================================================================== */

/*
 * convertToNRGBA (convert.go:34-64) applied to what Go's image/jpeg decoder returns: an
 * *image.YCbCr (or *image.Gray when cb == cr == NULL) with Rect.Min == (0,0).
 *
 * THIRD-PARTY ARITHMETIC, RESTATED: img.At(x,y).RGBA() is Go's standard library --
 * image.YCbCr.COffset (image/ycbcr.go) for the chroma sample of (x,y), then
 * color.YCbCr.RGBA() (image/color/ycbcr.go) -- toolchain pinned by go.mod:3 (go 1.25.5),
 * source not under /root/reference.  The published algorithm: yy1 = Y*0x10101,
 * cb1 = Cb-128, cr1 = Cr-128; r = yy1 + 91881*cr1, g = yy1 - 22554*cb1 - 46802*cr1,
 * b = yy1 + 116130*cb1; each channel: if (uint32(v) & 0xff000000) == 0 then v >>= 8 else
 * v = ^(v >> 31) & 0xffff  (saturate to 0 / 0xffff); alpha 0xffff.  convertToNRGBA then takes
 * the a == 0xffff branch: uint8(c >> 8) (convert.go:48-53).  color.Gray.RGBA() is y * 0x101.
 * PARITY UNPINNED like the rest of this file, and additionally a restatement from the published
 * source of a dependency rather than from files under /root/reference.
 *
 * ratio: image.YCbCrSubsampleRatio -- 0: 4:4:4, 1: 4:2:2, 2: 4:2:0, 3: 4:4:0, 4: 4:1:1, 5: 4:1:0.
 */
static int32_t ycc_channel(int32_t v)
{
    if (((uint32_t)v & 0xff000000u) == 0) return v >> 8;
    return ~(v >> 31) & 0xffff;
}

ORC_API int orc_ycbcr_coffset(int ratio, int x, int y, int cstride)
{
    switch (ratio) {
    case 1: return y * cstride + x / 2;
    case 2: return (y / 2) * cstride + x / 2;
    case 3: return (y / 2) * cstride + x;
    case 4: return y * cstride + x / 4;
    case 5: return (y / 2) * cstride + x / 4;
    default: return y * cstride + x;
    }
}

ORC_API void orc_ycbcr_to_nrgba(const uint8_t *yp, int ystride, const uint8_t *cb, const uint8_t *cr, int cstride,
                                int ratio, int w, int h, uint8_t *dst, int dstride)
{
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            uint32_t r, g, b;
            int32_t yy = yp[(size_t)y * ystride + x];
            if (!cb || !cr) {
                r = g = b = (uint32_t)yy * 0x101u; /* color.Gray.RGBA() */
            } else {
                int co = orc_ycbcr_coffset(ratio, x, y, cstride);
                int32_t yy1 = yy * 0x10101;
                int32_t cb1 = (int32_t)cb[co] - 128, cr1 = (int32_t)cr[co] - 128;
                r = (uint32_t)ycc_channel(yy1 + 91881 * cr1);
                g = (uint32_t)ycc_channel(yy1 - 22554 * cb1 - 46802 * cr1);
                b = (uint32_t)ycc_channel(yy1 + 116130 * cb1);
            }
            uint8_t *o = dst + (size_t)y * dstride + (size_t)x * 4;
            o[0] = (uint8_t)(r >> 8); /* a == 0xffff branch, convert.go:48-53 */
            o[1] = (uint8_t)(g >> 8);
            o[2] = (uint8_t)(b >> 8);
            o[3] = 0xff;
        }
    }
}

/* ------------------------------------------------------------------ */
/* JPEG quantisation round trip: what decode(encode(img, q)) does to   */
/* the PIXELS (compress.go:50-58 via io.go:157-169: jpeg.Encode with   */
/* Options{Quality: q}, then jpeg.Decode + toNRGBARef)                 */
/* ------------------------------------------------------------------ */
/*
 * The arithmetic is Go's standard library (image/jpeg: writer.go, fdct.go, idct.go, reader.go, scan.go;
 * image/color: ycbcr.go), toolchain pinned by go.mod:3 (go 1.25.5) and NOT under /root/reference.  It is restated here
 * from the published algorithms those files implement and cite -- the IJG's jfdctint.c (13-bit constants, the
 * "slow-but-accurate" integer FDCT, output scaled by 8), the Chen-Wang integer IDCT of the MPEG-2 reference decoder
 * (w1..w7 = 2048*sqrt(2)*cos(k*pi/16)), Annex K's quantisation tables with the IJG quality scaling, JFIF's colour
 * equations in 16.16 fixed point -- as Go applies them: baseline, 4:2:0, 16x16 MCUs, edge pixels replicated, chroma
 * averaged 2x2 as (sum + 2) >> 2, coefficients divided by 8*q rounded half away from zero.  Entropy coding is lossless
 * and therefore absent: the decoded image is a function of the quantised coefficients alone.
 * PARITY UNPINNED TWICE OVER: no Go toolchain to compare with, and the source restated is not even in the reference
 * tree.  tests/test_jpeg_roundtrip.py sanity-checks it against libjpeg-turbo (Pillow): identical quantisation tables,
 * decoded images within a few grey levels -- libjpeg differs from Go in its chroma constants, its downsampling bias and
 * its fancy upsampling, so closeness is all that can be asked.
 */
/* Annex K.1 / K.2 in natural (row-major) order; writer.go holds them in zig-zag order (unscaledQuant), and so does the
 * file: quantisation is element-wise, so the order cancels out of the round trip */
static const uint8_t jpeg_k1[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                                    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t jpeg_k2[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                                    47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

/* writer.go, Encode: quality clipped to [1, 100]; scale = 5000/q below 50, 200 - 2q from 50; x = (x*scale + 50) / 100 in
 * [1, 255].  Tables returned in NATURAL order (q[unzig[zig]] is writer.go's e.quant[..][zig]). */
ORC_API void orc_jpeg_quant_tables(int quality, uint8_t *lum, uint8_t *chr)
{
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int i = 0; i < 64; i++) {
        int x = ((int)jpeg_k1[i] * scale + 50) / 100;
        lum[i] = (uint8_t)(x < 1 ? 1 : (x > 255 ? 255 : x));
        x = ((int)jpeg_k2[i] * scale + 50) / 100;
        chr[i] = (uint8_t)(x < 1 ? 1 : (x > 255 ? 255 : x));
    }
}

/* color.RGBToYCbCr (image/color/ycbcr.go): 16.16 fixed point, 19595 + 38470 + 7471 == 65536 */
ORC_API void orc_rgb_to_ycbcr(uint8_t r, uint8_t g, uint8_t b, uint8_t *yy, uint8_t *cb, uint8_t *cr)
{
    const int32_t r1 = r, g1 = g, b1 = b;
    *yy = (uint8_t)((19595 * r1 + 38470 * g1 + 7471 * b1 + (1 << 15)) >> 16);
    int32_t c = -11056 * r1 - 21712 * g1 + 32768 * b1 + (257 << 15);
    c = (((uint32_t)c & 0xff000000u) == 0) ? c >> 16 : ~(c >> 31);
    *cb = (uint8_t)c;
    c = 32768 * r1 - 27440 * g1 - 5328 * b1 + (257 << 15);
    c = (((uint32_t)c & 0xff000000u) == 0) ? c >> 16 : ~(c >> 31);
    *cr = (uint8_t)c;
}

/* fdct.go: jfdctint.c's algorithm, 13-bit constants, level shift included, results scaled up by 8 */
#define JF_0_298631336 2446
#define JF_0_390180644 3196
#define JF_0_541196100 4433
#define JF_0_765366865 6270
#define JF_0_899976223 7373
#define JF_1_175875602 9633
#define JF_1_501321110 12299
#define JF_1_847759065 15137
#define JF_1_961570560 16069
#define JF_2_053119869 16819
#define JF_2_562915447 20995
#define JF_3_072711026 25172
#define JF_CONST_BITS 13
#define JF_PASS1_BITS 2

static void jpeg_fdct_1d(int32_t *s, int stride, int pass)
{
    const int32_t x0 = s[0], x1 = s[stride], x2 = s[2 * stride], x3 = s[3 * stride], x4 = s[4 * stride], x5 = s[5 * stride],
                  x6 = s[6 * stride], x7 = s[7 * stride];
    int32_t tmp0 = x0 + x7, tmp1 = x1 + x6, tmp2 = x2 + x5, tmp3 = x3 + x4;
    int32_t tmp10 = tmp0 + tmp3, tmp12 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp13 = tmp1 - tmp2;
    tmp0 = x0 - x7; tmp1 = x1 - x6; tmp2 = x2 - x5; tmp3 = x3 - x4;
    int32_t z1;
    const int sh = pass == 1 ? JF_CONST_BITS - JF_PASS1_BITS : JF_CONST_BITS + JF_PASS1_BITS;
    if (pass == 1) {
        s[0] = (tmp10 + tmp11 - 8 * 128) << JF_PASS1_BITS;             /* 8 * centerJSample */
        s[4 * stride] = (tmp10 - tmp11) << JF_PASS1_BITS;
    } else {
        tmp10 += 1 << (JF_PASS1_BITS - 1);
        s[0] = (tmp10 + tmp11) >> JF_PASS1_BITS;
        s[4 * stride] = (tmp10 - tmp11) >> JF_PASS1_BITS;
    }
    z1 = (tmp12 + tmp13) * JF_0_541196100;
    z1 += 1 << (sh - 1);
    s[2 * stride] = (z1 + tmp12 * JF_0_765366865) >> sh;
    s[6 * stride] = (z1 - tmp13 * JF_1_847759065) >> sh;

    tmp10 = tmp0 + tmp3; tmp11 = tmp1 + tmp2; tmp12 = tmp0 + tmp2; tmp13 = tmp1 + tmp3;
    z1 = (tmp12 + tmp13) * JF_1_175875602;
    z1 += 1 << (sh - 1);
    tmp0 *= JF_1_501321110; tmp1 *= JF_3_072711026; tmp2 *= JF_2_053119869; tmp3 *= JF_0_298631336;
    tmp10 *= -JF_0_899976223; tmp11 *= -JF_2_562915447; tmp12 *= -JF_0_390180644; tmp13 *= -JF_1_961570560;
    tmp12 += z1; tmp13 += z1;
    s[1 * stride] = (tmp0 + tmp10 + tmp12) >> sh;
    s[3 * stride] = (tmp1 + tmp11 + tmp13) >> sh;
    s[5 * stride] = (tmp2 + tmp11 + tmp12) >> sh;
    s[7 * stride] = (tmp3 + tmp10 + tmp13) >> sh;
}

ORC_API void orc_jpeg_fdct(int32_t *b)
{
    for (int y = 0; y < 8; y++) jpeg_fdct_1d(b + 8 * y, 1, 1);
    for (int x = 0; x < 8; x++) jpeg_fdct_1d(b + x, 8, 2);
}

/* idct.go: Chen-Wang, w_k = 2048*sqrt(2)*cos(k*pi/16) */
#define JI_W1 2841
#define JI_W2 2676
#define JI_W3 2408
#define JI_W5 1609
#define JI_W6 1108
#define JI_W7 565
#define JI_R2 181

ORC_API void orc_jpeg_idct(int32_t *src)
{
    for (int y = 0; y < 8; y++) {                                      /* horizontal 1-D IDCT */
        int32_t *s = src + 8 * y;
        if (s[1] == 0 && s[2] == 0 && s[3] == 0 && s[4] == 0 && s[5] == 0 && s[6] == 0 && s[7] == 0) {
            const int32_t dc = s[0] << 3;
            for (int i = 0; i < 8; i++) s[i] = dc;
            continue;
        }
        int32_t x0 = (s[0] << 11) + 128, x1 = s[4] << 11, x2 = s[6], x3 = s[2], x4 = s[1], x5 = s[7], x6 = s[5], x7 = s[3], x8;
        x8 = JI_W7 * (x4 + x5);
        x4 = x8 + (JI_W1 - JI_W7) * x4;
        x5 = x8 - (JI_W1 + JI_W7) * x5;
        x8 = JI_W3 * (x6 + x7);
        x6 = x8 - (JI_W3 - JI_W5) * x6;
        x7 = x8 - (JI_W3 + JI_W5) * x7;
        x8 = x0 + x1;
        x0 -= x1;
        x1 = JI_W6 * (x3 + x2);
        x2 = x1 - (JI_W2 + JI_W6) * x2;
        x3 = x1 + (JI_W2 - JI_W6) * x3;
        x1 = x4 + x6;
        x4 -= x6;
        x6 = x5 + x7;
        x5 -= x7;
        x7 = x8 + x3;
        x8 -= x3;
        x3 = x0 + x2;
        x0 -= x2;
        x2 = (JI_R2 * (x4 + x5) + 128) >> 8;
        x4 = (JI_R2 * (x4 - x5) + 128) >> 8;
        s[0] = (x7 + x1) >> 8; s[1] = (x3 + x2) >> 8; s[2] = (x0 + x4) >> 8; s[3] = (x8 + x6) >> 8;
        s[4] = (x8 - x6) >> 8; s[5] = (x0 - x4) >> 8; s[6] = (x3 - x2) >> 8; s[7] = (x7 - x1) >> 8;
    }
    for (int x = 0; x < 8; x++) {                                      /* vertical 1-D IDCT (no all-zero shortcut) */
        int32_t *s = src + x;
        int32_t y0 = (s[8 * 0] << 8) + 8192, y1 = s[8 * 4] << 8, y2 = s[8 * 6], y3 = s[8 * 2], y4 = s[8 * 1], y5 = s[8 * 7],
                y6 = s[8 * 5], y7 = s[8 * 3], y8;
        y8 = JI_W7 * (y4 + y5) + 4;
        y4 = (y8 + (JI_W1 - JI_W7) * y4) >> 3;
        y5 = (y8 - (JI_W1 + JI_W7) * y5) >> 3;
        y8 = JI_W3 * (y6 + y7) + 4;
        y6 = (y8 - (JI_W3 - JI_W5) * y6) >> 3;
        y7 = (y8 - (JI_W3 + JI_W5) * y7) >> 3;
        y8 = y0 + y1;
        y0 -= y1;
        y1 = JI_W6 * (y3 + y2) + 4;
        y2 = (y1 - (JI_W2 + JI_W6) * y2) >> 3;
        y3 = (y1 + (JI_W2 - JI_W6) * y3) >> 3;
        y1 = y4 + y6;
        y4 -= y6;
        y6 = y5 + y7;
        y5 -= y7;
        y7 = y8 + y3;
        y8 -= y3;
        y3 = y0 + y2;
        y0 -= y2;
        y2 = (JI_R2 * (y4 + y5) + 128) >> 8;
        y4 = (JI_R2 * (y4 - y5) + 128) >> 8;
        s[8 * 0] = (y7 + y1) >> 14; s[8 * 1] = (y3 + y2) >> 14; s[8 * 2] = (y0 + y4) >> 14; s[8 * 3] = (y8 + y6) >> 14;
        s[8 * 4] = (y8 - y6) >> 14; s[8 * 5] = (y0 - y4) >> 14; s[8 * 6] = (y3 - y2) >> 14; s[8 * 7] = (y7 - y1) >> 14;
    }
}

/* writer.go div: a / b rounded to nearest, halves away from zero */
static int32_t jpeg_div(int32_t a, int32_t b)
{
    if (a >= 0) return (a + (b >> 1)) / b;
    return -((-a + (b >> 1)) / b);
}

/* one 8x8 block of samples (0..255) through fdct -> quantise -> dequantise -> idct -> level shift + clamp
 * (writer.go writeBlock; reader.go reconstructBlock); q in natural order */
ORC_API void orc_jpeg_block_roundtrip(int32_t *b, const uint8_t *q)
{
    orc_jpeg_fdct(b);
    for (int i = 0; i < 64; i++) b[i] = jpeg_div(b[i], 8 * (int32_t)q[i]) * (int32_t)q[i];
    orc_jpeg_idct(b);
    for (int i = 0; i < 64; i++) b[i] = b[i] < -128 ? 0 : (b[i] > 127 ? 255 : b[i] + 128);
}

/* The six 8x8 sample blocks of MCU (mx0, my0) as writer.go forms them: four Y blocks (block i at xOff = (i&1)*8,
 * yOff = (i&2)*4), then Cb and Cr scaled 16x16 -> 8x8. */
static void jpeg_mcu_samples(const uint8_t *src, int sstride, int w, int h, int mx0, int my0, int32_t yb[4][64], int32_t *cbb,
                             int32_t *crb)
{
    int32_t cbf[4][64], crf[4][64];
    for (int i = 0; i < 4; i++) {                                      /* writer.go: xOff = (i&1)*8, yOff = (i&2)*4 */
        const int px = 16 * mx0 + (i & 1) * 8, py = 16 * my0 + (i & 2) * 4;
        for (int j = 0; j < 8; j++)
            for (int k = 0; k < 8; k++) {
                const int sx = px + k > w - 1 ? w - 1 : px + k, sy = py + j > h - 1 ? h - 1 : py + j;   /* toYCbCr clamps */
                const uint8_t *p = src + (size_t)sy * sstride + (size_t)sx * 4;
                uint8_t yy, cb, cr;
                /* io.go:157-169: an opaque image is handed over as *image.RGBA (rgbaToYCbCr reads the bytes); any
                 * other goes through toYCbCr's m.At(x, y).RGBA(): color.NRGBA.RGBA() premultiplies in 16 bits
                 * (r = R * 0x101 * A / 0xff) and the encoder keeps the high byte.  For A == 255 that IS R, so the
                 * per-pixel form below covers both. */
                const uint32_t a8 = p[3];
                const uint8_t r8 = (uint8_t)(((uint32_t)p[0] * 0x101u * a8 / 0xffu) >> 8);
                const uint8_t g8 = (uint8_t)(((uint32_t)p[1] * 0x101u * a8 / 0xffu) >> 8);
                const uint8_t b8 = (uint8_t)(((uint32_t)p[2] * 0x101u * a8 / 0xffu) >> 8);
                orc_rgb_to_ycbcr(r8, g8, b8, &yy, &cb, &cr);
                yb[i][8 * j + k] = yy; cbf[i][8 * j + k] = cb; crf[i][8 * j + k] = cr;
            }
    }
    for (int i = 0; i < 4; i++) {                                      /* writer.go scale(): 16x16 -> 8x8, (sum + 2) >> 2 */
        const int dstOff = ((i & 2) << 4) | ((i & 1) << 2);
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
                const int j = 16 * y + 2 * x;
                cbb[8 * y + x + dstOff] = (cbf[i][j] + cbf[i][j + 1] + cbf[i][j + 8] + cbf[i][j + 9] + 2) >> 2;
                crb[8 * y + x + dstOff] = (crf[i][j] + crf[i][j + 1] + crf[i][j + 8] + crf[i][j + 9] + 2) >> 2;
            }
    }
}

/* The planes an *image.YCbCr holds after jpeg.Decode(jpeg.Encode(img, quality)) for an opaque NRGBA image: Y is
 * 16*mx wide and 16*my high (mx, my = MCUs), Cb / Cr 8*mx x 8*my (4:2:0); the decoder's image is the w x h sub-image.
 * yp: (16 mx) * (16 my) bytes, cb, cr: (8 mx) * (8 my). */
ORC_API void orc_jpeg_roundtrip_planes(const uint8_t *src, int sstride, int w, int h, int quality, uint8_t *yp, uint8_t *cbp,
                                       uint8_t *crp)
{
    uint8_t ql[64], qc[64];
    orc_jpeg_quant_tables(quality, ql, qc);
    const int mx = (w + 15) / 16, my = (h + 15) / 16, ys = 16 * mx, cs = 8 * mx;
    for (int my0 = 0; my0 < my; my0++)
        for (int mx0 = 0; mx0 < mx; mx0++) {
            int32_t yb[4][64], cbb[64], crb[64];
            jpeg_mcu_samples(src, sstride, w, h, mx0, my0, yb, cbb, crb);
            for (int i = 0; i < 4; i++) {
                orc_jpeg_block_roundtrip(yb[i], ql);
                const int px = 16 * mx0 + (i & 1) * 8, py = 16 * my0 + (i & 2) * 4;
                for (int j = 0; j < 8; j++)
                    for (int k = 0; k < 8; k++) yp[(size_t)(py + j) * ys + px + k] = (uint8_t)yb[i][8 * j + k];
            }
            orc_jpeg_block_roundtrip(cbb, qc);
            orc_jpeg_block_roundtrip(crb, qc);
            for (int j = 0; j < 8; j++)
                for (int k = 0; k < 8; k++) {
                    cbp[(size_t)(8 * my0 + j) * cs + 8 * mx0 + k] = (uint8_t)cbb[8 * j + k];
                    crp[(size_t)(8 * my0 + j) * cs + 8 * mx0 + k] = (uint8_t)crb[8 * j + k];
                }
        }
}

ORC_API int orc_jpeg_roundtrip(const uint8_t *src, int sstride, int w, int h, int quality, uint8_t *dst, int dstride)
{
    if (w <= 0 || h <= 0) return 0;
    const int mx = (w + 15) / 16, my = (h + 15) / 16;
    uint8_t *yp = (uint8_t *)malloc((size_t)256 * mx * my), *cb = (uint8_t *)malloc((size_t)64 * mx * my),
            *cr = (uint8_t *)malloc((size_t)64 * mx * my);
    if (!yp || !cb || !cr) { free(yp); free(cb); free(cr); return -1; }
    orc_jpeg_roundtrip_planes(src, sstride, w, h, quality, yp, cb, cr);
    orc_ycbcr_to_nrgba(yp, 16 * mx, cb, cr, 8 * mx, 2, w, h, dst, dstride);      /* ratio 2 = YCbCrSubsampleRatio420 */
    free(yp); free(cb); free(cr);
    return 1;
}

/* ------------------------------------------------------------------ */
/* Baseline JPEG entropy coding: jpeg.Encode's file, byte for byte as   */
/* far as it can be restated (io.go:157-169), and a decoder for it      */
/* ------------------------------------------------------------------ */
/*
 * writer.go's encoder is baseline sequential, 8-bit, three components 4:2:0, one scan, no restart markers, the
 * typical Huffman tables of Annex K.3.3 (theHuffmanSpec), no JFIF APP0 segment: SOI, one DQT segment with both
 * tables (zig-zag order), SOF0, one DHT segment with the four tables in the order luminance DC, luminance AC,
 * chrominance DC, chrominance AC, SOS, entropy-coded data with 0xff stuffed, the last byte padded with 1 bits, EOI.
 * Per block (writeBlock): FDCT, dc = div(b[0], 8 q[0]) coded as the difference to the component's previous dc;
 * ac = div(b[unzig[zig]], 8 q[zig]) in zig-zag order with (run, size) symbols, 0xf0 for 16 zeros, 0x00 at the end of
 * a block that ends in zeros.  Restated from the standard (ITU T.81) and from how Go lays the file out, NOT from Go's
 * source: parity unpinned twice over, like the round trip above.  What is checked: libjpeg-turbo (Pillow) decodes the
 * files; its own non-optimised files carry exactly these Huffman tables; the decoder below, reading the encoder's
 * file, returns exactly orc_jpeg_roundtrip_planes (the entropy coder is lossless and the file is self-consistent); and
 * the decoder reads libjpeg's baseline files to pixels a few levels from libjpeg's own decode (its IDCT and
 * chroma upsampling differ by design).
 */
static const uint8_t jpeg_unzig[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                       41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                       30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef struct {
    uint8_t count[16];
    const uint8_t *value;
    int nvalue;
} jpeg_huff_spec;

static const uint8_t jpeg_dc_values[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t jpeg_ac_lum_values[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t jpeg_ac_chr_values[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
/* index = the file's (class << 1 | table id) order of writer.go: 0 luminance DC, 1 luminance AC, 2 chrominance DC, 3 chrominance AC */
static const jpeg_huff_spec jpeg_specs[4] = {
    {{0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, jpeg_dc_values, 12},
    {{0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125}, jpeg_ac_lum_values, 162},
    {{0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0}, jpeg_dc_values, 12},
    {{0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119}, jpeg_ac_chr_values, 162}};

/* canonical codes: lut[value] = length << 24 | code */
static void jpeg_build_lut(const jpeg_huff_spec *sp, uint32_t *lut)
{
    for (int i = 0; i < 256; i++) lut[i] = 0;
    uint32_t code = 0;
    int k = 0;
    for (int len = 1; len <= 16; len++) {
        for (int j = 0; j < sp->count[len - 1]; j++) {
            lut[sp->value[k]] = ((uint32_t)len << 24) | code;
            code++;
            k++;
        }
        code <<= 1;
    }
}

/* the standard tables as (length << 24 | code) per symbol: [4][256] */
ORC_API void orc_jpeg_huffman_luts(uint32_t *luts)
{
    for (int t = 0; t < 4; t++) jpeg_build_lut(&jpeg_specs[t], luts + 256 * t);
}

typedef struct {
    uint8_t *out;
    size_t cap, n;
    uint32_t bits, nbits;
    int overflow;
} jpeg_bitw;

static void jw_byte(jpeg_bitw *w, uint8_t b)
{
    if (w->n < w->cap) w->out[w->n] = b;
    else w->overflow = 1;
    w->n++;
}

/* writer.go emit: MSB first, a stuffed 0x00 after every 0xff */
static void jw_emit(jpeg_bitw *w, uint32_t bits, uint32_t nbits)
{
    nbits += w->nbits;
    bits <<= 32 - nbits;
    bits |= w->bits;
    while (nbits >= 8) {
        const uint8_t b = (uint8_t)(bits >> 24);
        jw_byte(w, b);
        if (b == 0xff) jw_byte(w, 0x00);
        bits <<= 8;
        nbits -= 8;
    }
    w->bits = bits;
    w->nbits = nbits;
}

static void jw_huff(jpeg_bitw *w, const uint32_t *lut, int value)
{
    const uint32_t x = lut[value];
    jw_emit(w, x & 0x00ffffffu, x >> 24);
}

static int jpeg_bitcount(uint32_t a)          /* bits needed for a (0 -> 0) */
{
    int n = 0;
    while (a) { n++; a >>= 1; }
    return n;
}

/* emitHuffRLE: the (run, size) symbol, then `size` bits of the value (negative values as value - 1, low bits) */
static void jw_huff_rle(jpeg_bitw *w, const uint32_t *lut, int run, int32_t value)
{
    int32_t a = value, b = value;
    if (a < 0) { a = -value; b = value - 1; }
    const int nbits = jpeg_bitcount((uint32_t)a);
    jw_huff(w, lut, (run << 4) | nbits);
    if (nbits > 0) jw_emit(w, (uint32_t)b & ((1u << nbits) - 1u), (uint32_t)nbits);
}

/* writeBlock: samples (level-unshifted 0..255) -> the block's code; returns its quantised dc.  zz (optional, 64):
 * the quantised coefficients in zig-zag order. */
static int32_t jpeg_write_block(jpeg_bitw *w, int32_t *b, const uint8_t *q, const uint32_t *dc_lut, const uint32_t *ac_lut,
                                int32_t prev_dc, int16_t *zz)
{
    orc_jpeg_fdct(b);
    const int32_t dc = jpeg_div(b[0], 8 * (int32_t)q[0]);
    if (zz) zz[0] = (int16_t)dc;
    jw_huff_rle(w, dc_lut, 0, dc - prev_dc);
    int run = 0;
    for (int zig = 1; zig < 64; zig++) {
        const int nat = jpeg_unzig[zig];
        const int32_t ac = jpeg_div(b[nat], 8 * (int32_t)q[nat]);
        if (zz) zz[zig] = (int16_t)ac;
        if (ac == 0) {
            run++;
        } else {
            while (run > 15) {
                jw_huff(w, ac_lut, 0xf0);
                run -= 16;
            }
            jw_huff_rle(w, ac_lut, run, ac);
            run = 0;
        }
    }
    if (run > 0) jw_huff(w, ac_lut, 0x00);
    return dc;
}

static void jw_marker(jpeg_bitw *w, uint8_t m, int len)
{
    jw_byte(w, 0xff); jw_byte(w, m);
    jw_byte(w, (uint8_t)(len >> 8)); jw_byte(w, (uint8_t)(len & 0xff));
}

/* The header bytes jpeg.Encode writes before the entropy-coded data (SOI .. SOS header); returns their count. */
static void jpeg_write_headers(jpeg_bitw *w, int wd, int ht, const uint8_t *ql, const uint8_t *qc)
{
    jw_byte(w, 0xff); jw_byte(w, 0xd8);                                           /* SOI */
    jw_marker(w, 0xdb, 2 + 2 * (1 + 64));                                         /* DQT: both tables, zig-zag order */
    for (int t = 0; t < 2; t++) {
        jw_byte(w, (uint8_t)t);
        for (int zig = 0; zig < 64; zig++) jw_byte(w, (t ? qc : ql)[jpeg_unzig[zig]]);
    }
    jw_marker(w, 0xc0, 8 + 3 * 3);                                                /* SOF0 */
    jw_byte(w, 8);
    jw_byte(w, (uint8_t)(ht >> 8)); jw_byte(w, (uint8_t)(ht & 0xff));
    jw_byte(w, (uint8_t)(wd >> 8)); jw_byte(w, (uint8_t)(wd & 0xff));
    jw_byte(w, 3);
    jw_byte(w, 1); jw_byte(w, 0x22); jw_byte(w, 0);                               /* Y: 2x2, table 0 */
    jw_byte(w, 2); jw_byte(w, 0x11); jw_byte(w, 1);                               /* Cb */
    jw_byte(w, 3); jw_byte(w, 0x11); jw_byte(w, 1);                               /* Cr */
    int dht = 2;
    for (int t = 0; t < 4; t++) dht += 1 + 16 + jpeg_specs[t].nvalue;
    jw_marker(w, 0xc4, dht);                                                      /* DHT: the four tables in one segment */
    static const uint8_t tc_th[4] = {0x00, 0x10, 0x01, 0x11};
    for (int t = 0; t < 4; t++) {
        jw_byte(w, tc_th[t]);
        for (int i = 0; i < 16; i++) jw_byte(w, jpeg_specs[t].count[i]);
        for (int i = 0; i < jpeg_specs[t].nvalue; i++) jw_byte(w, jpeg_specs[t].value[i]);
    }
    static const uint8_t sos[14] = {0xff, 0xda, 0x00, 0x0c, 0x03, 0x01, 0x00, 0x02, 0x11, 0x03, 0x11, 0x00, 0x3f, 0x00};
    for (int i = 0; i < 14; i++) jw_byte(w, sos[i]);
}

ORC_API int orc_jpeg_header_bytes(int wd, int ht, int quality, uint8_t *out, int cap)
{
    uint8_t ql[64], qc[64];
    orc_jpeg_quant_tables(quality, ql, qc);
    jpeg_bitw w = {out, (size_t)cap, 0, 0, 0, 0};
    jpeg_write_headers(&w, wd, ht, ql, qc);
    return w.overflow ? -(int)w.n : (int)w.n;
}

/* jpeg.Encode(img, &jpeg.Options{Quality: quality}) for an NRGBA image (io.go:157-169).  Returns the file's size;
 * a negative size: `cap` was too small (the needed size, negated).  coef (optional): every block's quantised
 * coefficients in zig-zag order, blocks in scan order (MCU-major: Y0 Y1 Y2 Y3 Cb Cr), 64 int16 each. */
ORC_API long orc_jpeg_encode(const uint8_t *src, int sstride, int w, int h, int quality, uint8_t *out, long cap, int16_t *coef)
{
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535) return 0;
    uint8_t ql[64], qc[64];
    orc_jpeg_quant_tables(quality, ql, qc);
    uint32_t luts[4][256];
    for (int t = 0; t < 4; t++) jpeg_build_lut(&jpeg_specs[t], luts[t]);
    jpeg_bitw bw = {out, (size_t)(cap > 0 ? cap : 0), 0, 0, 0, 0};
    jpeg_write_headers(&bw, w, h, ql, qc);
    const int mx = (w + 15) / 16, my = (h + 15) / 16;
    int32_t pdy = 0, pdcb = 0, pdcr = 0;
    size_t blk = 0;
    for (int my0 = 0; my0 < my; my0++)
        for (int mx0 = 0; mx0 < mx; mx0++) {
            int32_t yb[4][64], cbb[64], crb[64];
            jpeg_mcu_samples(src, sstride, w, h, mx0, my0, yb, cbb, crb);
            for (int i = 0; i < 4; i++, blk++) pdy = jpeg_write_block(&bw, yb[i], ql, luts[0], luts[1], pdy, coef ? coef + 64 * blk : NULL);
            pdcb = jpeg_write_block(&bw, cbb, qc, luts[2], luts[3], pdcb, coef ? coef + 64 * blk : NULL); blk++;
            pdcr = jpeg_write_block(&bw, crb, qc, luts[2], luts[3], pdcr, coef ? coef + 64 * blk : NULL); blk++;
        }
    jw_emit(&bw, 0x7f, 7);                                                        /* pad the last byte with 1s */
    jw_byte(&bw, 0xff); jw_byte(&bw, 0xd9);                                       /* EOI */
    return bw.overflow ? -(long)bw.n : (long)bw.n;
}

/* ---- a baseline decoder: 8-bit, one component or three (luminance factors 1 or 2), one scan, restart intervals or none
 * (what jpeg.Encode, libjpeg and cameras write).  Entropy decoding is the standard's (F.2.2); the pixels are made the way reader.go /
 * idct.go make them: coefficient * q, the Chen-Wang IDCT, + 128, clamp. */
typedef struct {
    const uint8_t *p;
    size_t n, pos;
    uint32_t bits;
    int nbits, bad;
} jpeg_bitr;

static int jr_bit(jpeg_bitr *r)
{
    if (r->nbits == 0) {
        if (r->pos >= r->n) { r->bad = 1; return 0; }
        uint8_t b = r->p[r->pos++];
        if (b == 0xff) {
            if (r->pos < r->n && r->p[r->pos] == 0x00) r->pos++;
            else { r->bad = 1; return 0; }                                        /* a marker inside the scan */
        }
        r->bits = b;
        r->nbits = 8;
    }
    r->nbits--;
    return (int)((r->bits >> r->nbits) & 1u);
}

typedef struct {
    int mincode[17], maxcode[17], valptr[17];
    uint8_t value[256];
} jpeg_dtab;

static int jr_symbol(jpeg_bitr *r, const jpeg_dtab *t)
{
    int code = 0;
    for (int len = 1; len <= 16; len++) {
        code = (code << 1) | jr_bit(r);
        if (r->bad) return 0;
        if (t->maxcode[len] >= 0 && code <= t->maxcode[len] && code >= t->mincode[len]) return t->value[t->valptr[len] + code - t->mincode[len]];
    }
    r->bad = 1;
    return 0;
}

static int32_t jr_receive_extend(jpeg_bitr *r, int s)
{
    if (s == 0) return 0;
    int32_t v = 0;
    for (int i = 0; i < s; i++) v = (v << 1) | jr_bit(r);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

/* ---- progressive files (SOF2): scan.go's processSOS / refine / refineNonZeroes and reader.go's reconstructProgressiveImage
 * as published (image/jpeg is a standard-library dependency, absent from /root/reference: restated from ITU T.81 Annex G and the
 * package's documented behaviour; parity with Go unpinned like the rest of the codec).  What pins the ENTROPY side: libjpeg's
 * progressive file of an image holds exactly the quantised coefficients of its baseline file of the same image and quality, and
 * tests/test_jpeg_progressive.py asks this decoder for both.  image/jpeg's own choices kept here:
 *   - coefficients of every scan are collected and dequantised once, at EOI, with the tables in force THEN;
 *   - a one-component frame is h = v = 1 whatever the file says;
 *   - an interleaved scan walks the frame's MCUs, a one-component scan walks the component's blocks in raster order and has no
 *     data for blocks wholly outside the image;
 *   - only the blocks that touch the image are reconstructed (bx * 8 * h0 / h < width ...): the MCU padding of a progressive
 *     image's planes stays zero;
 *   - the restart counter of image/jpeg counts FRAME MCUs in every scan where T.81 counts the scan's own (one block in a
 *     one-component scan): the two only agree for components of one block per MCU, so a restart interval together with a
 *     one-component scan of a component with more is refused here (-12) as it is by the product (the host codec's call). */
static int jr_bits(jpeg_bitr *r, int nb)
{
    int v = 0;
    for (int i = 0; i < nb; i++) v = (v << 1) | jr_bit(r);
    return v;
}

/* refineNonZeroes (scan.go): passes over coefficients zig .. zig_end; every non-zero one reads a correction bit; stops in front
 * of the (nz + 1)-th zero one (nz < 0: never).  b in zig-zag order. */
static int jprog_refine_nonzeroes(jpeg_bitr *r, int32_t *b, int zig, int zig_end, int nz, int32_t delta)
{
    for (; zig <= zig_end; zig++) {
        if (b[zig] == 0) {
            if (nz == 0) break;
            nz--;
            continue;
        }
        if (!jr_bit(r)) continue;
        if (b[zig] >= 0) b[zig] += delta; else b[zig] -= delta;
    }
    return zig;
}

/* refine (scan.go; T.81 G.1.2.2, G.1.2.3) */
static int jprog_refine(jpeg_bitr *r, int32_t *b, const jpeg_dtab *h, int zs, int ze, int32_t delta, uint32_t *eob_run)
{
    if (zs == 0) {
        if (jr_bit(r)) b[0] |= delta;
        return r->bad ? -10 : 0;
    }
    int zig = zs;
    if (*eob_run == 0) {
        for (; zig <= ze; zig++) {
            int32_t z = 0;
            const int rs = jr_symbol(r, h);
            if (r->bad) return -10;
            const int v0 = rs >> 4, v1 = rs & 15;
            if (v1 == 0) {
                if (v0 != 15) {
                    *eob_run = 1u << v0;
                    if (v0 != 0) *eob_run |= (uint32_t)jr_bits(r, v0);
                    break;
                }
            } else if (v1 == 1) {
                z = jr_bit(r) ? delta : -delta;
            } else {
                return -10;                                                       /* "unexpected Huffman code" */
            }
            zig = jprog_refine_nonzeroes(r, b, zig, ze, v0, delta);
            if (r->bad) return -10;
            if (zig > ze) return -10;                                             /* "too many coefficients" */
            if (z != 0) b[zig] = z;
        }
    }
    if (*eob_run > 0) {
        (*eob_run)--;
        jprog_refine_nonzeroes(r, b, zig, ze, -1, delta);
    }
    return r->bad ? -10 : 0;
}

static int orc_jpeg_decode_scans(const uint8_t *data, long n, int *wd, int *ht, int *ratio, uint8_t *yp, uint8_t *cbp, uint8_t *crp,
                                 int16_t *coef, uint8_t *kp, int *adobe);
static int orc_jpeg_decode_progressive(const uint8_t *data, long n, int *wd, int *ht, int *ratio, uint8_t *yp, uint8_t *cbp, uint8_t *crp,
                                       int16_t *coef)
{
    return orc_jpeg_decode_scans(data, n, wd, ht, ratio, yp, cbp, crp, coef, NULL, NULL);
}

/* kp != NULL: four-component frames too (reader.go: the fourth component is the black plane, d.blackPix), all four at 1 x 1 --
 * of image/jpeg's two layouts ([0x11 x 4], [0x22, 0x11, 0x11, 0x22]) the one every encoder writes; *adobe = the APP14 transform
 * (-1: no Adobe segment, which applyBlack refuses).  *ratio = -2 for such a frame. */
static int orc_jpeg_decode_scans(const uint8_t *data, long n, int *wd, int *ht, int *ratio, uint8_t *yp, uint8_t *cbp, uint8_t *crp,
                                 int16_t *coef, uint8_t *kp, int *adobe)
{
    /* Also the SEQUENTIAL files the one-scan decoder below does not read (r5): SOF0 / SOF1 frames whose components come in
     * scans of their own, and SOF1 (extended sequential, 8 bit) altogether.  processSOS is the same function for them with
     * Ss, Se, Ah, Al fixed at 0, 63, 0, 0 whatever the scan header says (Table B.3), and a block is dequantised when its scan
     * decodes it -- with the table in force THEN (qsnap), not at EOI. */
    int sequential = 0;
    uint8_t qsnap[4][64];
    uint8_t q[4][64];
    int adobe_t = -1;
    jpeg_dtab dt[2][4];
    int have_q[4] = {0, 0, 0, 0}, have_t[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    int ri = 0, W = 0, H = 0, ncomp = 0, comp_h[4] = {1, 1, 1, 1}, comp_v[4] = {1, 1, 1, 1}, comp_q[4] = {0, 0, 0, 0}, comp_id[4] = {0, 0, 0, 0};
    int hy = 1, vy = 1, mx = 0, my = 0, seen[4] = {0, 0, 0, 0};
    int32_t *cf[4] = {NULL, NULL, NULL, NULL};                                          /* per component: [my v][mx h] blocks of 64, zig-zag order */
    int rc = -2;
    long pos = 2;
    for (;;) {
        if (pos + 2 > n) { rc = -2; goto out; }
        if (data[pos] != 0xff) { pos++; continue; }                               /* reader.go: bytes between segments are skipped */
        const uint8_t m = data[pos + 1];
        if (m == 0xff) { pos++; continue; }
        if (m == 0x00 || m == 0x01 || (m >= 0xd0 && m <= 0xd7)) { pos += 2; continue; }
        if (m == 0xd9) break;                                                     /* EOI */
        if (pos + 4 > n) { rc = -2; goto out; }
        const int len = (data[pos + 2] << 8) | data[pos + 3];
        if (len < 2 || pos + 2 + len > n) { rc = -2; goto out; }
        const uint8_t *seg = data + pos + 4;
        const int sl = len - 2;
        if (m == 0xdb) {
            int o = 0;
            while (o < sl) {
                const int pq = seg[o] >> 4, tq = seg[o] & 15;
                if (pq != 0 || tq > 3 || o + 65 > sl) { rc = -3; goto out; }
                for (int zig = 0; zig < 64; zig++) q[tq][zig] = seg[o + 1 + zig];  /* kept in zig-zag order here */
                have_q[tq] = 1;
                o += 65;
            }
        } else if (m == 0xc2 || m == 0xc0 || m == 0xc1) {
            if (ncomp != 0) { rc = -4; goto out; }
            sequential = m != 0xc2;
            if (sl < 6 + 3 || seg[0] != 8 || (seg[5] != 3 && seg[5] != 1 && !(seg[5] == 4 && kp)) || sl < 6 + 3 * seg[5]) { rc = seg[5] == 4 ? -14 : -4; goto out; }
            ncomp = seg[5];
            H = (seg[1] << 8) | seg[2]; W = (seg[3] << 8) | seg[4];
            if (W <= 0 || H <= 0) { rc = -4; goto out; }
            for (int c = 0; c < ncomp; c++) {
                comp_id[c] = seg[6 + 3 * c]; comp_h[c] = seg[7 + 3 * c] >> 4; comp_v[c] = seg[7 + 3 * c] & 15; comp_q[c] = seg[8 + 3 * c];
                if (comp_q[c] > 3) { rc = -4; goto out; }
            }
            if (ncomp == 1) { comp_h[0] = comp_v[0] = 1; }
            else if (ncomp == 4) {
                for (int c = 0; c < 4; c++) if (comp_h[c] != 1 || comp_v[c] != 1) { rc = -9; goto out; }
            } else {
                if (comp_h[1] != 1 || comp_v[1] != 1 || comp_h[2] != 1 || comp_v[2] != 1) { rc = -9; goto out; }
                if (!(comp_h[0] == 1 || comp_h[0] == 2 || comp_h[0] == 4) || comp_v[0] < 1 || comp_v[0] > 2) { rc = -9; goto out; }
            }
            hy = comp_h[0]; vy = comp_v[0];
            mx = (W + 8 * hy - 1) / (8 * hy); my = (H + 8 * vy - 1) / (8 * vy);
            *wd = W; *ht = H;
            *ratio = ncomp == 4 ? -2 : ncomp == 1 ? -1 : (hy == 4 ? (vy == 2 ? 5 : 4) : hy == 2 ? (vy == 2 ? 2 : 1) : (vy == 2 ? 3 : 0));
            if (!yp) return 1;
            for (int c = 0; c < ncomp; c++) {
                cf[c] = (int32_t *)calloc((size_t)mx * comp_h[c] * my * comp_v[c] * 64, sizeof(int32_t));
                if (!cf[c]) { rc = -20; goto out; }
            }
        } else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8) {
            rc = -5; goto out;
        } else if (m == 0xc4) {
            int o = 0;
            while (o < sl) {
                const int tc = seg[o] >> 4, th = seg[o] & 15;
                if (tc > 1 || th > 3 || o + 17 > sl) { rc = -6; goto out; }
                jpeg_dtab *t = &dt[tc][th];
                int total = 0, code = 0, k = 0;
                for (int len2 = 1; len2 <= 16; len2++) {
                    const int cnt = seg[o + len2];
                    t->valptr[len2] = k;
                    t->mincode[len2] = code;
                    t->maxcode[len2] = cnt ? code + cnt - 1 : -1;
                    code = (code + cnt) << 1;
                    k += cnt;
                    total += cnt;
                }
                if (total > 256 || o + 17 + total > sl) { rc = -6; goto out; }
                for (int i = 0; i < total; i++) t->value[i] = seg[o + 17 + i];
                have_t[tc][th] = 1;
                o += 17 + total;
            }
        } else if (m == 0xdd) {
            if (sl < 2) { rc = -2; goto out; }
            ri = (seg[0] << 8) | seg[1];
        } else if (m == 0xee) {
            if (sl >= 12 && memcmp(seg, "Adobe", 5) == 0) {
                adobe_t = seg[11];
                if (!kp && seg[11] != 1) { rc = -9; goto out; }
            }
        } else if (m == 0xda) {
            if (ncomp == 0) { rc = -8; goto out; }
            const int ns = sl >= 1 ? seg[0] : 0;
            if (ns < 1 || ns > ncomp || sl != 4 + 2 * ns) { rc = -8; goto out; }
            int sc[4], td[4], ta[4];
            for (int i = 0; i < ns; i++) {
                int c = -1;
                for (int j = 0; j < ncomp; j++) if (comp_id[j] == seg[1 + 2 * i]) c = j;
                if (c < 0) { rc = -8; goto out; }
                for (int j = 0; j < i; j++) if (sc[j] == c) { rc = -8; goto out; }
                sc[i] = c; td[i] = seg[2 + 2 * i] >> 4; ta[i] = seg[2 + 2 * i] & 15;
                if (td[i] > 3 || ta[i] > 3) { rc = -8; goto out; }
            }
            int zs = seg[1 + 2 * ns], ze = seg[2 + 2 * ns], ah = seg[3 + 2 * ns] >> 4, al = seg[3 + 2 * ns] & 15;
            if (sequential) { zs = 0; ze = 63; ah = 0; al = 0; }
            if ((zs == 0 && ze != 0 && !sequential) || zs > ze || ze > 63) { rc = -8; goto out; }   /* "bad spectral selection bounds" */
            if (zs != 0 && ns != 1) { rc = -8; goto out; }                           /* AC scans hold one component */
            if (ah != 0 && ah != al + 1) { rc = -8; goto out; }                      /* "bad successive approximation values" */
            if (al > 13) { rc = -8; goto out; }
            for (int i = 0; i < ns; i++) {
                if (zs == 0 && ah == 0 && !have_t[0][td[i]]) { rc = -8; goto out; }
                if ((zs != 0 || sequential) && !have_t[1][ta[i]]) { rc = -8; goto out; }
                if (sequential) {
                    if (seen[sc[i]]) { rc = -8; goto out; }                          /* a sequential component has ONE scan */
                    if (!have_q[comp_q[sc[i]]]) { rc = -8; goto out; }
                    memcpy(qsnap[sc[i]], q[comp_q[sc[i]]], 64);
                }
                seen[sc[i]] = 1;
            }
            if (ri > 0 && ns == 1 && comp_h[sc[0]] * comp_v[sc[0]] > 1) { rc = -12; goto out; }
            jpeg_bitr br = {data + pos + 2 + len, (size_t)(n - (pos + 2 + len)), 0, 0, 0, 0};
            int32_t pred[4] = {0, 0, 0, 0};
            uint32_t eob_run = 0;
            long mcu = 0, block_count = 0;
            int expected_rst = 0;
            for (int my0 = 0; my0 < my; my0++)
                for (int mx0 = 0; mx0 < mx; mx0++) {
                    for (int i = 0; i < ns; i++) {
                        const int c = sc[i], hi = comp_h[c], vi = comp_v[c];
                        for (int j = 0; j < hi * vi; j++) {
                            int bx, by;
                            if (ns != 1) { bx = hi * mx0 + j % hi; by = vi * my0 + j / hi; }
                            else {
                                const int qq = mx * hi;
                                bx = (int)(block_count % qq); by = (int)(block_count / qq);
                                block_count++;
                                /* the component's own extent in samples: ceil(W hi / hy) across, ceil(H vi / vy) down */
                                if ((long)bx * 8 * hy >= (long)W * hi || (long)by * 8 * vy >= (long)H * vi) continue;
                            }
                            int32_t *b = cf[c] + ((size_t)by * mx * hi + bx) * 64;
                            if (ah != 0) {
                                const int e = jprog_refine(&br, b, &dt[1][ta[i]], zs, ze, (int32_t)1 << al, &eob_run);
                                if (e) { rc = e; goto out; }
                                continue;
                            }
                            int zig = zs;
                            if (zig == 0) {
                                zig++;
                                const int s = jr_symbol(&br, &dt[0][td[i]]);
                                if (br.bad || s > 16) { rc = -10; goto out; }
                                pred[c] += jr_receive_extend(&br, s);
                                b[0] = pred[c] * ((int32_t)1 << al);
                            }
                            if (zig <= ze && eob_run > 0) eob_run--;
                            else {
                                for (; zig <= ze; zig++) {
                                    const int rs = jr_symbol(&br, &dt[1][ta[i]]);
                                    if (br.bad) { rc = -10; goto out; }
                                    const int v0 = rs >> 4, v1 = rs & 15;
                                    if (v1 != 0) {
                                        zig += v0;
                                        if (zig > ze) break;
                                        b[zig] = jr_receive_extend(&br, v1) * ((int32_t)1 << al);
                                    } else {
                                        if (v0 != 15) {
                                            eob_run = 1u << v0;
                                            if (v0 != 0) eob_run |= (uint32_t)jr_bits(&br, v0);
                                            eob_run--;
                                            break;
                                        }
                                        zig += 15;
                                    }
                                }
                            }
                            if (br.bad) { rc = -10; goto out; }
                        }
                    }
                    mcu++;
                    if (ri > 0 && mcu % ri == 0 && mcu < (long)mx * my) {
                        br.nbits = 0;
                        if (br.pos + 2 > br.n || br.p[br.pos] != 0xff || br.p[br.pos + 1] != (uint8_t)(0xd0 + expected_rst)) { rc = -11; goto out; }
                        br.pos += 2;
                        expected_rst = (expected_rst + 1) & 7;
                        pred[0] = pred[1] = pred[2] = pred[3] = 0;
                        eob_run = 0;
                    }
                }
            pos = (long)(br.p - data) + (long)br.pos;                             /* the next marker is looked for from here */
            continue;
        }
        pos += 2 + len;
    }
    if (ncomp == 0) { rc = -4; goto out; }
    /* reconstructProgressiveImage: dequantise with the tables as they stand now, idct.go's IDCT, level shift, clamp */
    {
        const int ys = 8 * hy * mx, cs = 8 * mx;
        memset(yp, 0, (size_t)ys * 8 * vy * my);
        if (ncomp >= 3) { memset(cbp, 0, (size_t)cs * 8 * my); memset(crp, 0, (size_t)cs * 8 * my); }
        if (ncomp == 4) memset(kp, 0, (size_t)ys * 8 * vy * my);
        if (adobe) *adobe = adobe_t;
        for (int c = 0; c < ncomp; c++) {
            if (!seen[c]) continue;                                               /* progCoeffs[i] == nil: the plane stays zero */
            if (!sequential && !have_q[comp_q[c]]) { rc = -8; goto out; }
            const int hi = comp_h[c], vi = comp_v[c], stride = mx * hi;
            uint8_t *plane = c == 0 ? yp : (c == 1 ? cbp : (c == 2 ? crp : kp));
            const int ps = (c == 0 || c == 3) ? ys : cs;
            for (int by = 0; (long)by * 8 * vy < (long)H * vi; by++)
                for (int bx = 0; (long)bx * 8 * hy < (long)W * hi; bx++) {
                    const int32_t *z = cf[c] + ((size_t)by * stride + bx) * 64;
                    int32_t b[64];
                    for (int zig = 0; zig < 64; zig++) {
                        if (z[zig] > 32767 || z[zig] < -32768) { rc = -13; goto out; }   /* beyond what any 8-bit image's file holds */
                        b[jpeg_unzig[zig]] = z[zig] * (int32_t)(sequential ? qsnap[c][zig] : q[comp_q[c]][zig]);
                    }
                    orc_jpeg_idct(b);
                    for (int j = 0; j < 8; j++)
                        for (int k = 0; k < 8; k++) {
                            const int32_t s = b[8 * j + k];
                            plane[(size_t)(8 * by + j) * ps + 8 * bx + k] = (uint8_t)(s < -128 ? 0 : (s > 127 ? 255 : s + 128));
                        }
                }
        }
        if (coef) {                                                               /* the baseline decoder's layout: MCU by MCU, zig-zag order */
            size_t blk = 0;
            for (int my0 = 0; my0 < my; my0++)
                for (int mx0 = 0; mx0 < mx; mx0++)
                    for (int c = 0; c < ncomp; c++)
                        for (int j = 0; j < comp_h[c] * comp_v[c]; j++, blk++) {
                            const int bx = comp_h[c] * mx0 + j % comp_h[c], by = comp_v[c] * my0 + j / comp_h[c];
                            const int32_t *z = cf[c] + ((size_t)by * mx * comp_h[c] + bx) * 64;
                            for (int k = 0; k < 64; k++) coef[64 * blk + k] = (int16_t)z[k];
                        }
        }
        rc = 1;
    }
out:
    for (int c = 0; c < 4; c++) free(cf[c]);
    return rc;
}

/* toNRGBARef(jpeg.Decode(data)) of a four-component file (8 bit, every component 1 x 1), any frame type the scan-by-scan decoder
 * reads.  reader.go applyBlack: without an Adobe segment the file is refused; transform 0 (CMYK): the stored samples are inverted,
 * C, M, Y, K = 255 - s; any other transform (YCbCrK): the first three planes go through YCbCr -> RGB and stand for C, M, Y as
 * they are (the RGB -> CMY inversion cancels the Adobe inversion), K = 255 - s.  Then convert.go:34-64 over image.CMYK:
 * color.CMYK.RGBA() -- w = 0xffff - K * 0x101, r = (0xffff - C * 0x101) * w / 0xffff -- and the opaque branch's r >> 8.
 * dst == NULL: the dimensions only.  Returns 1, or a negative error. */
ORC_API int orc_jpeg_decode_cmyk(const uint8_t *data, long n, int *wd, int *ht, uint8_t *dst, int dstride)
{
    int ratio = 0, adobe = -1;
    uint8_t dummy = 0;
    int rc = orc_jpeg_decode_scans(data, n, wd, ht, &ratio, NULL, NULL, NULL, NULL, &dummy, NULL);
    if (rc != 1 || ratio != -2) return rc == 1 ? -14 : rc;
    if (!dst) return 1;
    const int W = *wd, H = *ht, mx = (W + 7) / 8, my = (H + 7) / 8, ps = 8 * mx;
    const size_t pb = (size_t)ps * 8 * my;
    uint8_t *pl = (uint8_t *)malloc(4 * pb);
    if (!pl) return -20;
    rc = orc_jpeg_decode_scans(data, n, wd, ht, &ratio, pl, pl + pb, pl + 2 * pb, NULL, pl + 3 * pb, &adobe);
    if (rc == 1 && adobe < 0) rc = -15;                        /* "4-component JPEG doesn't have Adobe APP14 metadata" */
    for (int y = 0; rc == 1 && y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t i = (size_t)y * ps + x;
            uint32_t c, m, yy, k = 255u - pl[3 * pb + i];
            if (adobe == 0) { c = 255u - pl[i]; m = 255u - pl[pb + i]; yy = 255u - pl[2 * pb + i]; }
            else {                                                  /* color.YCbCrToRGB */
                const int32_t y1 = (int32_t)pl[i] * 0x10101, cb1 = (int32_t)pl[pb + i] - 128, cr1 = (int32_t)pl[2 * pb + i] - 128;
                int32_t r = y1 + 91881 * cr1, g = y1 - 22554 * cb1 - 46802 * cr1, b = y1 + 116130 * cb1;
                r = ((uint32_t)r & 0xff000000u) == 0 ? r >> 16 : (r < 0 ? 0 : 255);
                g = ((uint32_t)g & 0xff000000u) == 0 ? g >> 16 : (g < 0 ? 0 : 255);
                b = ((uint32_t)b & 0xff000000u) == 0 ? b >> 16 : (b < 0 ? 0 : 255);
                c = (uint32_t)r; m = (uint32_t)g; yy = (uint32_t)b;
            }
            const uint32_t w = 0xffffu - k * 0x101u;
            uint8_t *o = dst + (size_t)y * dstride + 4 * x;
            o[0] = (uint8_t)(((0xffffu - c * 0x101u) * w / 0xffffu) >> 8);
            o[1] = (uint8_t)(((0xffffu - m * 0x101u) * w / 0xffffu) >> 8);
            o[2] = (uint8_t)(((0xffffu - yy * 0x101u) * w / 0xffffu) >> 8);
            o[3] = 0xff;
        }
    free(pl);
    return rc;
}

/* Decodes `data` into MCU-padded planes (yp: ys x 8*vmax*my rows ...).  Returns 1, with *wd, *ht, *ratio
 * (image.YCbCrSubsampleRatio: 0 4:4:4, 1 4:2:2, 2 4:2:0, 3 4:4:0; -1: one component, image.Gray) set, or a negative
 * error.  Call with yp == NULL to learn the dims first. */
ORC_API int orc_jpeg_decode_planes(const uint8_t *data, long n, int *wd, int *ht, int *ratio, uint8_t *yp, uint8_t *cbp, uint8_t *crp,
                                   int16_t *coef)
{
    if (n < 4 || data[0] != 0xff || data[1] != 0xd8) return -1;
    uint8_t q[4][64];
    jpeg_dtab dt[2][4];
    int have_q[4] = {0, 0, 0, 0}, have_t[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    int ri = 0;
    int W = 0, H = 0, ncomp = 0, comp_h[3] = {0, 0, 0}, comp_v[3] = {0, 0, 0}, comp_q[3] = {0, 0, 0}, comp_id[3] = {0, 0, 0};
    long pos = 2;
    for (;;) {
        if (pos + 4 > n || data[pos] != 0xff) return -2;
        const uint8_t m = data[pos + 1];
        if (m == 0xff) { pos++; continue; }
        const int len = (data[pos + 2] << 8) | data[pos + 3];
        if (pos + 2 + len > n) return -2;
        const uint8_t *seg = data + pos + 4;
        const int sl = len - 2;
        if (m == 0xdb) {                                                          /* DQT */
            int o = 0;
            while (o < sl) {
                const int pq = seg[o] >> 4, tq = seg[o] & 15;
                if (pq != 0 || tq > 3 || o + 65 > sl) return -3;
                for (int zig = 0; zig < 64; zig++) q[tq][jpeg_unzig[zig]] = seg[o + 1 + zig];
                have_q[tq] = 1;
                o += 65;
            }
        } else if (m == 0xc0) {                                                   /* SOF0 */
            if (sl < 6 + 3 || seg[0] != 8 || (seg[5] != 3 && seg[5] != 1) || sl < 6 + 3 * seg[5]) return -4;
            ncomp = seg[5];
            H = (seg[1] << 8) | seg[2]; W = (seg[3] << 8) | seg[4];
            for (int c = 0; c < ncomp; c++) {
                comp_id[c] = seg[6 + 3 * c]; comp_h[c] = seg[7 + 3 * c] >> 4; comp_v[c] = seg[7 + 3 * c] & 15; comp_q[c] = seg[8 + 3 * c];
            }
        } else if (m == 0xc2 || m == 0xc1) {
            return orc_jpeg_decode_progressive(data, n, wd, ht, ratio, yp, cbp, crp, coef);
        } else if (m >= 0xc5 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
            return -5;                                                            /* not baseline */
        } else if (m == 0xc4) {                                                   /* DHT */
            int o = 0;
            while (o < sl) {
                const int tc = seg[o] >> 4, th = seg[o] & 15;
                if (tc > 1 || th > 3 || o + 17 > sl) return -6;
                jpeg_dtab *t = &dt[tc][th];
                int total = 0, code = 0, k = 0;
                for (int len2 = 1; len2 <= 16; len2++) {
                    const int cnt = seg[o + len2];
                    t->valptr[len2] = k;
                    t->mincode[len2] = code;
                    t->maxcode[len2] = cnt ? code + cnt - 1 : -1;
                    code = (code + cnt) << 1;
                    k += cnt;
                    total += cnt;
                }
                if (total > 256 || o + 17 + total > sl) return -6;
                for (int i = 0; i < total; i++) t->value[i] = seg[o + 17 + i];
                have_t[tc][th] = 1;
                o += 17 + total;
            }
        } else if (m == 0xdd) {
            if (sl >= 2) ri = (seg[0] << 8) | seg[1];                             /* MCUs per restart interval; 0: none */
        } else if (m == 0xda) {                                                   /* SOS */
            if (ncomp != 0 && sl >= 1 && seg[0] >= 1 && seg[0] < ncomp)               /* the components in scans of their own */
                return orc_jpeg_decode_progressive(data, n, wd, ht, ratio, yp, cbp, crp, coef);
            if (ncomp == 0 || sl < 1 + 2 * ncomp + 3 || seg[0] != ncomp) return -8;
            int td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
            for (int c = 0; c < ncomp; c++) {
                if (seg[1 + 2 * c] != comp_id[c]) return -8;
                td[c] = seg[2 + 2 * c] >> 4; ta[c] = seg[2 + 2 * c] & 15;
                if (!have_t[0][td[c]] || !have_t[1][ta[c]] || !have_q[comp_q[c]]) return -8;
            }
            if (W <= 0 || H <= 0) return -4;
            /* Y blocks per MCU across (1, 2, 4) / down (1, 2): 4:4:4, 4:2:2, 4:2:0, 4:4:0, 4:1:1, 4:1:0; one component: a block per MCU,
             * whatever its factors say (T.81 A.2.2: a one-component scan is not interleaved) */
            int hy = 1, vy = 1;
            if (ncomp == 3) {
                if (comp_h[1] != 1 || comp_v[1] != 1 || comp_h[2] != 1 || comp_v[2] != 1) return -9;
                hy = comp_h[0]; vy = comp_v[0];
                if (!(hy == 1 || hy == 2 || hy == 4) || vy < 1 || vy > 2) return -9;   /* reader.go processSOF: h, v in {1, 2, 4}, luminance v == 4 unsupported */
            }
            *wd = W; *ht = H;
            /* image.YCbCrSubsampleRatio (reader.go makeImg: 444, 422, 420, 440, 411, 410); -1: image.Gray */
            *ratio = ncomp == 1 ? -1 : (hy == 4 ? (vy == 2 ? 5 : 4) : hy == 2 ? (vy == 2 ? 2 : 1) : (vy == 2 ? 3 : 0));
            if (!yp) return 1;
            const int msx = 8 * hy, msy = 8 * vy, mx = (W + msx - 1) / msx, my = (H + msy - 1) / msy, ys = msx * mx, cs = 8 * mx;
            jpeg_bitr br = {data + pos + 2 + len, (size_t)(n - (pos + 2 + len)), 0, 0, 0, 0};
            int32_t pred[3] = {0, 0, 0};
            size_t blk = 0;
            long mcu = 0;
            for (int my0 = 0; my0 < my; my0++)
                for (int mx0 = 0; mx0 < mx; mx0++, mcu++) {
                    if (ri > 0 && mcu > 0 && mcu % ri == 0) {
                        /* a restart interval ends: the rest of the byte is padding, RSTn follows, predictions start over */
                        br.nbits = 0;
                        if (br.pos + 2 > br.n || br.p[br.pos] != 0xff || br.p[br.pos + 1] != (uint8_t)(0xd0 + ((mcu / ri - 1) & 7))) return -11;
                        br.pos += 2;
                        pred[0] = pred[1] = pred[2] = 0;
                    }
                    for (int c = 0; c < ncomp; c++) {
                        const int nb = c == 0 ? hy * vy : 1;
                        for (int i = 0; i < nb; i++, blk++) {
                            int32_t b[64];
                            int16_t zz[64];
                            for (int k = 0; k < 64; k++) { b[k] = 0; zz[k] = 0; }
                            const int s = jr_symbol(&br, &dt[0][td[c]]);
                            pred[c] += jr_receive_extend(&br, s);
                            zz[0] = (int16_t)pred[c];
                            for (int zig = 1; zig < 64;) {
                                const int rs = jr_symbol(&br, &dt[1][ta[c]]);
                                const int rr = rs >> 4, ss = rs & 15;
                                if (br.bad) return -10;
                                if (ss == 0) {
                                    if (rr == 15) { zig += 16; continue; }
                                    break;                                        /* EOB */
                                }
                                zig += rr;
                                if (zig > 63) return -10;
                                zz[zig] = (int16_t)jr_receive_extend(&br, ss);
                                zig++;
                            }
                            if (br.bad) return -10;
                            if (coef) for (int k = 0; k < 64; k++) coef[64 * blk + k] = zz[k];
                            const uint8_t *qq = q[comp_q[c]];
                            for (int zig = 0; zig < 64; zig++) b[jpeg_unzig[zig]] = (int32_t)zz[zig] * (int32_t)qq[jpeg_unzig[zig]];
                            orc_jpeg_idct(b);
                            for (int k = 0; k < 64; k++) b[k] = b[k] < -128 ? 0 : (b[k] > 127 ? 255 : b[k] + 128);
                            if (c == 0) {
                                const int px = msx * mx0 + (i % hy) * 8, py = msy * my0 + (i / hy) * 8;
                                for (int j = 0; j < 8; j++)
                                    for (int k = 0; k < 8; k++) yp[(size_t)(py + j) * ys + px + k] = (uint8_t)b[8 * j + k];
                            } else {
                                uint8_t *cp = c == 1 ? cbp : crp;
                                for (int j = 0; j < 8; j++)
                                    for (int k = 0; k < 8; k++) cp[(size_t)(8 * my0 + j) * cs + 8 * mx0 + k] = (uint8_t)b[8 * j + k];
                            }
                        }
                    }
                }
            return 1;
        }
        pos += 2 + len;
    }
}
