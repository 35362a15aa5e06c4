// Internal header of libfennec_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/fennec_hip.h"

namespace fnx {

void set_error(const char *fmt, ...);

// ---- what steers the library from outside (r6: VERDICT r5 item 7) ------------------------------------------------------
// A release build reads FIVE environment names, each once, all listed in include/fennec_hip.h: FENNEC_HIP_DISABLE,
// FENNEC_HIP_DEVICES (runtime.cpp: which devices), FNX_ROCTX, FNX_POOL_TRACE, FNX_JPEG_TRACE (tracing; no effect on results
// or on which kernel runs).  Everything else:
//   * kernel FORMS the tests must be able to reach on any image (a fallback kernel that the product only takes for rare
//     tables, the two-pass twin of a fused launch ...) are a per-ctx selection, fnx_ctx_set_form (Form below) -- explicit,
//     per context, no process-wide state;
//   * development switches (tile sizes, priorities, thresholds, A/B of two correct kernels) exist in `make DEVELOP=1`
//     builds only: dev_env() is getenv there and a constant nullptr in a release build, so the compiler drops the branch.
#ifdef FNX_DEVELOP
inline const char *dev_env(const char *name) { return std::getenv(name); }
#else
inline const char *dev_env(const char *) { return nullptr; }
#endif

enum Form {
    FORM_FX_STREAM = 0,      // "fx_stream":       "0" = effects.hip's tile kernel instead of the streaming one
    FORM_FX_PAIRS,           // "fx_pairs":        "0" = the tile kernel's one-row form
    FORM_FX_REF,             // "fx_ref":          "1" = the fp64 reference-order kernel (what tables outside the guard take)
    FORM_RESIZE_MFMA,        // "resize_mfma":     "0" never / "1" downscales (default) / "2" wherever the tables allow; read when a plan is built
    FORM_RESIZE_FP64,        // "resize_fp64":     "1" = the fp64 reference-order resize kernels (what tables outside the guard take)
    FORM_RESIZE_FUSED,       // "resize_fused":    "0" = the two-pass kernels (what windows wider than the fused tile take)
    FORM_MSSSIM_LEVELWISE,   // "msssim_levelwise":"1" = one SSIMFast + one halving per level
    FORM_MSSSIM_NOFUSE0,     // "msssim_nofuse0":  "1" = level 0's boxes and halving as two reads
    FORM_MSSSIM_FOLD,        // "msssim_fold":     "0" = a finish launch instead of the fold in the window kernel
    FORM_MSSSIM_BOXFLY,      // "msssim_boxfly":   "1" = levels 1..4's boxes taken on the fly
    FORM_PALETTE_GRID,       // "palette_grid":    "0" every image walks the whole palette / "1" every image takes the grid
    FORM_COUNT
};

#define FNX_HIP(expr)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            fnx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                           __LINE__);                                                      \
            return e__ == hipErrorOutOfMemory ? FNX_ERR_OOM : FNX_ERR_HIP;                 \
        }                                                                                  \
    } while (0)

#define FNX_TRY(expr)            \
    do {                         \
        int r__ = (expr);        \
        if (r__ < 0) return r__; \
    } while (0)

#define FNX_REQUIRE(cond, msg)                  \
    do {                                        \
        if (!(cond)) {                          \
            fnx::set_error("invalid argument: %s", msg); \
            return FNX_ERR_INVALID;             \
        }                                       \
    } while (0)

// Scratch slots of a ctx (device memory, grown on demand, reused across calls).
enum Slot {
    SLOT_IN_A = 0,   // staged host input a / src
    SLOT_IN_B,       // staged host input b
    SLOT_OUT,        // staged host output
    SLOT_TMP0,       // op intermediates (blur/resize uint8 tmp, downsampled planes ...)
    SLOT_TMP1,
    SLOT_TMP2,
    SLOT_TMP3,
    SLOT_TABLE0,     // weight tables
    SLOT_TABLE1,
    SLOT_TABLE_MFMA, // blur_mfma.hip: the weight matrices of the matrix-pipe blur
    SLOT_PARTIAL,    // reduction partials
    SLOT_RESULT,     // scalar results
    SLOT_PTRS,       // pointer arrays of batched ops
    SLOT_BOXMAP,     // source column/row -> box index tables (blur + SSIMFast in one pass)
    SLOT_SLABS,      // per-tile box partial sums of that pass (buffer 0)
    // second halves of the one-pass pipeline's double buffers, and its private plane / partial slots: its tail
    // (box_from_slabs, windowed SSIM, finish) runs on the ctx's second stream under the NEXT step's blur
    SLOT_SLABS1,
    SLOT_PLANES0,
    SLOT_PLANES1,
    SLOT_PART0,
    SLOT_PART1,
    // the JPEG quantisation round trip (jpeg.hip): unquantised planes, planes at a quality, decoded NRGBA, search reference
    SLOT_JPEG0,
    SLOT_JPEG1,
    SLOT_JPEG2,
    SLOT_JPEG3,
    SLOT_JPEG_ENC, SLOT_JPEG_ENC2, SLOT_JPEG_LUT, SLOT_JPEG_ECS,   // jpeg.hip's entropy coder
    SLOT_FILE0, SLOT_FILE1, SLOT_FILE_SAMPLES,                     // host_api.cpp: fennec_CompressFileJPEG's images between its stages
    SLOT_JPEG_DEC, SLOT_JPEG_DEC_PLANES, SLOT_JPEG_DEC_IMG,        // jpeg_dec.hip: the decoder's work arrays, its planes, toNRGBARef's image
    SLOT_AN_HASH0, SLOT_AN_HASH1,   // analyze.hip: the colour-set tables of this call and the next (the launch that uses one zeroes the other)
    SLOT_DONE,       // workgroup counters of the kernels that finish their own reduction (ssim.hip), zero between launches
    SLOT_COUNT
};

struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
};

// Cached copy of the last table uploaded into a table slot.
struct TableCache {
    std::vector<unsigned char> host;
    bool fresh = false;      // the last upload_tables() into this slot really uploaded (contents changed)
    int contig_taps = 0;     // resize tap tables: resize_contiguous_taps() of the cached table
};

// Cached geometry of the one-pass blur + SSIMFast launch (blur.hip: launch_blur_scored)
struct ScoreGeom {
    int w = 0, h = 0, dstW = 0, dstH = 0, radius = 0;
    bool tall_pref = false, tall = false, ok = false;
    bool mfma = false;          // the geometry is blur_mfma_kernel's (tile = 64 px x seg rows)
    int seg = 0;
    int th = 0, nbx = 0, nby = 0;
    std::vector<int32_t> map;   // the table blob uploaded to SLOT_BOXMAP
};

// A CSR tap table of one resize pass (precomputeWeights, resize.go:164-197).  id != 0 marks an immutable
// table with a process-unique id (host_api.cpp's cache): plans are then looked up by id instead of content.
struct TapTable {
    const int32_t *off = nullptr, *idx = nullptr;
    const double *wt = nullptr;
    int nout = 0;
    uint64_t id = 0;
};

}  // namespace fnx

struct fnx_resize_plan;   // resize.hip: device tables of one (tap table, direction)

struct fnx_ctx {
    int device = 0;             // HIP device ordinal
    int logical_device = 0;     // its index in this library's device list (fnx_set_devices / FENNEC_HIP_DEVICES)
    hipStream_t stream = nullptr;       // where the ctx launches: its own stream, or one lent by the caller (fnx_ctx_use_stream)
    hipStream_t own_stream = nullptr;   // created with the ctx, destroyed with it
    hipEvent_t ev_switch = nullptr;
    // One-pass GaussianBlur + SSIMFast batches (api.cpp): the blur kernel of step s runs on `stream`, the
    // step's tail on `stream2`, ordered by ev_blur[p]; step s + 2 reuses buffer set p = s & 1 and waits for
    // ev_tail[p] first.  A caller that enqueues step s + 1 before fetching step s gets the tail for free.
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_blur[2] = {}, ev_tail[2] = {};
    hipEvent_t blur_done = nullptr;     // the stop event BOUND to the last one-pass blur dispatch (launch_direct_cfg)
    bool tail_pending[2] = {false, false};
    unsigned long long tail_gen[2] = {0, 0};   // uses of each buffer set so far
    int parity = 0;
    int ev_toggle = 0;           // fnx_ssim_enqueue alternates between the two hand-over events
    bool stream2_used = false;
    int partial_slot = -1;       // >= 0: launch_windowed_ssim takes its partial sums from this slot (one-pass tail)
    // fnx_gaussian_blur_batch with FNX_BLUR_KEEP_BOX_SUMS (api.cpp): the boxDownsample planes of the batch's sources and blurred
    // images, made by the blur's own pass, wait here for the fnx_ssim_fast_batch(_enqueue) that scores exactly these pairs --
    // the very next call on the ctx (op_seq) or nothing
    struct KeptBoxes {
        bool valid = false;
        unsigned long long seq = 0;
        int n = 0, sstride = 0, dstride = 0, w = 0, h = 0, nw = 0, nh = 0, parity = 0;
        std::vector<const uint8_t *> srcs, dsts;
        uint8_t *planes = nullptr;
        size_t plane = 0;
    } kept;
    unsigned long long op_seq = 0;     // exported calls bound to this ctx so far (bind)
    bool boxes_on_main = false;        // launch_blur_scored: box_from_slabs_kernel on `stream` instead of `stream2` (the kept form)
    fnx::Scratch slot[fnx::SLOT_COUNT];
    fnx::TableCache tcache[fnx::SLOT_COUNT];
    // pinned host ring: tables going up, scalars coming down
    unsigned char *pinned = nullptr;
    size_t pinned_cap = 0, pinned_off = 0;
    int num_cus = 256;
    // results of the *_enqueue calls not fetched yet, oldest first: each batch has its own pinned slots and
    // an event right behind its result kernels, so a fetch waits for THAT batch only and a caller may queue
    // the next batch before fetching this one (the stream never drains between steps)
    static constexpr int RES_DEPTH = 4;
    struct Pending {
        const double *pinned = nullptr;
        int n = 0;
        hipEvent_t ev = nullptr;
        // fnx_msssim_enqueue: the slots hold `nraw` per-level SSIMFast values and the fetch returns ONE number,
        // exp(sum weights[i] log(max(level i, 1e-10))) (ssim.go:344-352); nraw == 0: the values as they are
        int nraw = 0;
        int nimg = 1;          // fnx_msssim_batch_enqueue: that many images, five slots each, `nraw` of them written; the fetch returns nimg numbers
        double weights[5] = {0, 0, 0, 0, 0};
        int tail_parity = -1;  // one-pass batches: the buffer set (slabs, planes, partials) this batch's tail reads,
        unsigned long long tail_gen = 0;   // and which use of that set it was
    } res_q[RES_DEPTH];
    int res_head = 0, res_count = 0;
    struct ResBuf {            // pinned home of FIFO position i's results (api.cpp: result_slot_queued)
        double *p = nullptr;
        size_t cap = 0;
    } res_buf[RES_DEPTH];
    fnx::ScoreGeom score_geom;
    std::vector<fnx_resize_plan *> rplans;   // at most 8, least recently used evicted (resize.hip)
    uint32_t *rz_todo = nullptr;             // resize_mfma.hip -> resize_fused_kernel: tiles to redo, stamped with rz_gen
    size_t rz_todo_cap = 0;
    uint32_t rz_gen = 0;
    unsigned long long *rz_report = nullptr; // host-mapped: what resize_fused_sparse_kernel redid (gen << 32 | tiles)
    uint32_t rz_last_gen = 0;                // the last matrix launch whose report has not been read
    size_t rz_last_cells = 0;
    const fnx_resize_plan *rz_last_h = nullptr;
    uint32_t d21_gen = 0;                    // resize_dense21_kernel's launches (rz_report[1] holds the last one that met translucent content)
    // fnx_ctx_profile: event pairs around the profiled kernel launches, oldest unread first
    static constexpr int PROF_DEPTH = 32;
    int prof = 0;                 // bit mask of FNX_PROF_* kernel classes being bracketed (0: off)
    hipEvent_t prof_ev[PROF_DEPTH][2] = {};
    int prof_head = 0, prof_count = 0, prof_open = -1;
    // fnx_ctx_last_kernel: the kernel the last call of each class really launched (static strings), so that a report can
    // name the route the library took instead of inferring it from environment switches
    const char *route[8] = {"", "", "", "", "", "", "", ""};
    int ssim_mode = 0;           // fnx_ctx_set_ssim_mode: FNX_SSIM_EXACT (fp64 moments) / FNX_SSIM_FAST (fp32 moments for full-resolution planes)
    // fnx_ctx_set_form: the kernel forms this ctx was told to take ("" = the product's own choice)
    char form[fnx::FORM_COUNT][8] = {};
    // analyze.hip's single-launch form: which colour table the next call uses, and how many tables of each are not zero
    int an_cur = 0;
    int an_dirty[2] = {0, 0};
};

struct fnx_prepared {
    int w = 0, h = 0;      // original dims
    int pw = 0, ph = 0;    // dims SSIMFast compares at
    uint8_t *pix = nullptr;  // device, tight pw x ph NRGBA (downsampled or copied reference side)
};

namespace fnx {

int bind(fnx_ctx *ctx);
// the ctx's selection for a kernel form (fnx_ctx_set_form), else -- DEVELOP builds only -- the environment variable of the
// same name (FNX_ + upper case: the A/B scripts under tools/ and experiments/), else nullptr: the product's own choice
const char *form_value(const fnx_ctx *ctx, Form f);
int form_set(fnx_ctx *ctx, const char *name, const char *value);
// roctx range around an exported op (runtime.cpp: FNX_ROCTX=1 turns them on; off they cost a load and a branch)
class OpRange {
public:
    explicit OpRange(const char *name);
    ~OpRange();
    OpRange(const OpRange &) = delete;
    OpRange &operator=(const OpRange &) = delete;
private:
    bool on_;
};
#define FNX_ENTER(ctx)                  \
    fnx::OpRange fnx_op_range_(__func__); \
    FNX_TRY(bind(ctx))
// Device scratch of at least `bytes` in `slot` (contents undefined).
int scratch(fnx_ctx *ctx, Slot slot, size_t bytes, void **out);
// Pinned host bytes valid until the next fnx call on this ctx wraps the ring.
int pinned_alloc(fnx_ctx *ctx, size_t bytes, void **out);
// Upload a small host table into a table slot (skipped when identical to the cached one).
int upload_table(fnx_ctx *ctx, Slot slot, const void *host, size_t bytes, void **dptr);
// Upload several host arrays back to back into one slot (each 16-byte aligned).
int upload_tables(fnx_ctx *ctx, Slot slot, const void *const *hosts, const size_t *sizes, int n,
                  void **dptrs);

// staged / intermediate images are TIGHT (stride = 4*w), like image.NewNRGBA; 16-byte vector
// access is then available whenever w % 4 == 0 (every size the benchmarks and codecs produce).
inline int pitch16(int w) { return w * 4; }

// An input image resolved to device memory.
struct DevImg {
    const uint8_t *p = nullptr;
    int stride = 0;
};
struct DevOut {
    uint8_t *p = nullptr;
    int stride = 0;
    uint8_t *host = nullptr;  // non-null: copy back to here (hstride) at finish
    int hstride = 0, w = 0, h = 0;
};

int stage_in(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, Slot slot,
             DevImg *out);
// Same, but keeps the caller's stride and copies the flat Pix slice ((h-1)*stride + 4*w bytes):
// pixelSSIM walks the flat slice, row padding included (ssim.go:178).
int stage_in_flat(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, Slot slot,
                  DevImg *out);
// toNRGBA's copy (convert.go:12-19: copy(dst.Pix, nrgba.Pix) into a fresh tight image): the FIRST 4wh flat bytes of
// the Pix slice as a tight w x h image -- the image's rows only when its stride is 4w.  MSSSIM's pyramid starts
// from these copies (ssim.go:345-346).  Device sources are read in place with stride 4w.
int stage_in_front(fnx_ctx *ctx, int space, const uint8_t *src, int w, int h, Slot slot, DevImg *out);
int stage_out(fnx_ctx *ctx, int space, uint8_t *dst, int dstride, int w, int h, Slot slot,
              DevOut *out);
// Copy a staged output back (if host) and synchronise when `space` is host.
int finish(fnx_ctx *ctx, int space, DevOut *out);
int finish_enqueue(fnx_ctx *ctx, int space, DevOut *out);   // the copy back without the wait
// fnx_ctx_profile hooks: bracket the launch of a profiled kernel (no-ops when profiling is off)
// Events carried by ONE dispatch (hipExtLaunchKernelGGL's startEvent / stopEvent): they ride on the kernel packet's own
// completion signal, so neither costs a barrier packet on the stream.  hipEventRecord before and after a launch is two
// barrier packets, ~6-8 us each on this part: with the cross-stream hand-over event that was a 24 us bubble between
// consecutive one-pass blur launches (r3 kernel trace).  prof_bind reserves the profile FIFO's next pair when the
// class is being profiled (nulls otherwise); the launch must then really happen.
struct LaunchEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};
int prof_bind(fnx_ctx *ctx, int cls, LaunchEvents *ev);
int prof_begin(fnx_ctx *ctx, int cls = FNX_PROF_MAIN);
inline void note_route(fnx_ctx *ctx, int cls, const char *kernel)      // cls: one FNX_PROF_* bit
{
    int i = 0;
    while (i < 7 && !((cls >> i) & 1)) i++;
    ctx->route[i] = kernel;
}
int prof_end(fnx_ctx *ctx);
// Fetch n doubles from device memory into host memory (synchronises).
int fetch_doubles(fnx_ctx *ctx, const double *dptr, double *host, int n);
int fetch_bytes(fnx_ctx *ctx, const void *dptr, void *host, size_t bytes);

inline bool aligned16(const void *p, int stride)
{
    return ((reinterpret_cast<uintptr_t>(p) | static_cast<uintptr_t>(stride)) & 15u) == 0;
}

// ---- kernel launchers (each enqueues on ctx->stream; device pointers only) ----
int launch_blur(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs, int sstride,
                int w, int h, const double *kernel, int radius, int flags, uint8_t *dst,
                uint8_t *const *dsts, int dstride);
// blur (fast, or with FNX_BLUR_EXACT the guarded bit-exact kernel) + both boxDownsample'd planes
// ([src 0..n-1][blurred 0..n-1], tight dstW x dstH) in one pass; FNX_NOOP (nothing launched) when the
// shape or the kernel is not covered.  srcs/dsts: device arrays.
int launch_blur_scored(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h,
                       const double *kernel, int radius, int flags, uint8_t *const *dsts, int dstride,
                       uint8_t *planes, size_t plane, int dstW, int dstH);
// blur_mfma.hip: GaussianBlur on the i8 matrix pipe (radius <= 6, non-negative weights summing to 1, w >= 64, h >= 32)
bool blur_mfma_covers(const double *kernel, int radius, int w, int h);
bool blur_mfma_exact_enabled();
bool blur_mfma_takes(const double *kernel, int radius, int w, int h, bool exact);
int blur_mfma_segment(const fnx_ctx *ctx, int n, int w, int h, int cap, int occ);
int launch_blur_mfma(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs, int sstride, int w, int h,
                     const double *kernel, int radius, int flags, uint8_t *dst, uint8_t *const *dsts, int dstride);
int launch_blur_mfma_scored(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h, const double *kernel,
                            int radius, int flags, uint8_t *const *dsts, int dstride, const int32_t *bx, const int32_t *by,
                            unsigned long long *slabs, int nbx, int nby, int seg);
bool blur_mfma_wide_scored_covers(const double *kernel, int radius, int w, int h, bool exact);
int launch_blur_mfma_wide_scored(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int w, int h, const double *kernel,
                                 int radius, int flags, uint8_t *const *dsts, int dstride, const int32_t *bx, const int32_t *by,
                                 unsigned long long *slabs, int nbx, int nby, int seg);
int launch_blur3x3(fnx_ctx *ctx, const uint8_t *src, int sstride, int w, int h, uint8_t *dst,
                   int dstride);
int launch_sharpen(fnx_ctx *ctx, bool adaptive, const uint8_t *src, int sstride, int w, int h,
                   double amount, uint8_t *dst, int dstride);
int launch_sharpen_batch(fnx_ctx *ctx, bool adaptive, int n, const uint8_t *src0, const uint8_t *const *d_srcs, int sstride, int w, int h,
                         double amount, uint8_t *dst0, uint8_t *const *d_dsts, int dstride);
// One pass of lanczosResize through the ctx's plan cache: guard-exact fp32 kernels where the table allows,
// the fp64 kernels otherwise.  vertical == false: dst is t.nout x srcH; true: dst is srcW x t.nout.
// `hint` (optional) carries the H pass's per-workgroup verdicts to the V pass of the same lanczosResize call: where
// most H waves found their rows dense with rounding-guard flags the V pass skips its fp32 form (resize.hip).
struct ResizeHint {
    uint32_t *cells = nullptr;     // device, `cap` words
    size_t cap = 0;
    int gx = 0, gy = 0, rows = 0;  // filled by the H pass
    bool valid = false;
};
int resize_pass(fnx_ctx *ctx, bool vertical, const TapTable &t, const uint8_t *src, int sstride, int srcW, int srcH,
                uint8_t *dst, int dstride, ResizeHint *hint = nullptr);
// both passes in one launch (the uint8 intermediate in LDS); FNX_NOOP when the tables are outside its reach
// nimg > 1: a batch of same-geometry images in one set of launches (blockIdx.z / .y = image); d_srcs / d_dsts are DEVICE arrays
// of their pointers (src / dst are then ignored)
int resize_fused(fnx_ctx *ctx, const TapTable &th, const TapTable &tv, const uint8_t *src, int sstride, int srcW, int srcH,
                 uint8_t *dst, int dstride, int nimg = 1, const uint8_t *const *d_srcs = nullptr, uint8_t *const *d_dsts = nullptr);
void free_resize_plans(fnx_ctx *ctx);
// resize_mfma.hip: lanczosResize on the i8 matrix pipe (opaque images, ratios up to ~2.3).  One tap table in matrix form:
struct RzMfTable {
    bool ok = false;
    int KH = 0;                    // H tables: 64-byte chunks of a group's window (1, 2)
    int NC = 0;                    // H tables: 16-byte chunks of a staged source row
    int ngroups = 0;               // H: groups of 4 outputs, 16 per 64-column strip; V: groups of 16 output rows
    int nmat = 0;                  // distinct matrices
    int S = 22;                    // fixed point of the weights: 2^22 or 2^23
    int seed = 0, thr = 0;         // rounding seed + G, 2 G (units of 2^-S)
    void *blob = nullptr;          // device: matrices | per-group records | per-strip source offsets
    const void *mats = nullptr;
    const int32_t *meta = nullptr, *sbase = nullptr;
    const void *ex = nullptr;      // per output: the operands of the fp64 fix-ups
};
bool resize_mfma_build(const fnx_ctx *ctx, const TapTable &t, int srcN, bool vertical, const double *inv, RzMfTable *out);
void resize_mfma_free(RzMfTable *t);
// Launches the matrix kernel; regions it gives up are marked todo[tile] = gen for resize_fused_kernel's tiles
// (old_tw x old_th output px, old_gx per row), which the caller launches next.
int resize_mfma_launch(fnx_ctx *ctx, const RzMfTable &h, const RzMfTable &v,
                       const uint8_t *src, int sstride, int srcW, int srcH, uint8_t *dst, int dstride, int dstW, int dstH,
                       uint32_t *todo, unsigned *gave_up, uint32_t gen, int old_tw, int old_th, int old_gx, int *workgroups,
                       int nimg = 1, const uint8_t *const *d_srcs = nullptr, uint8_t *const *d_dsts = nullptr, uint32_t todo_stride = 0);
// lanczosResize (resize.go:37-53) with both tables given: the body of fnx_lanczos_resize
int lanczos_resize_tables(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int srcW, int srcH,
                          const TapTable &th, const TapTable &tv, uint8_t *dst, int dstride, int dstW, int dstH);
int lanczos_resize_tables_batch(fnx_ctx *ctx, int n, const uint8_t *const *srcs, int sstride, int srcW, int srcH,
                                const TapTable &th, const TapTable &tv, uint8_t *const *dsts, int dstride, int dstW, int dstH);
// contig_taps: resize_contiguous_taps() of the (host) H table -- most taps of any output when every
// output's tap indices are consecutive, else 0
int launch_resize_h(fnx_ctx *ctx, const uint8_t *src, int sstride, int srcW, int srcH,
                    const int32_t *d_off, const int32_t *d_idx, const double *d_wt, uint8_t *dst,
                    int dstride, int dstW, int contig_taps);
int resize_contiguous_taps(const int32_t *off, const int32_t *idx, int nout);
int launch_resize_v(fnx_ctx *ctx, const uint8_t *src, int sstride, int srcW, int srcH,
                    const int32_t *d_off, const int32_t *d_idx, const double *d_wt, uint8_t *dst,
                    int dstride, int dstH, int contig_taps);
// n images: src either one pointer or device array; dst images are tight dstW x dstH, image i at
// dst + i*dst_image_bytes.
int launch_box_downsample(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs,
                          int sstride, int srcW, int srcH, uint8_t *dst, int dstride,
                          size_t dst_image_bytes, int dstW, int dstH);
int launch_box_downsample_pair(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs,
                               int sstride, const uint8_t *src_b, const uint8_t *const *srcs_b, int sstride_b,
                               int srcW, int srcH, uint8_t *dst, int dstride, size_t dst_image_bytes,
                               int dstW, int dstH);
// Windowed SSIM of n image pairs (tight or strided NRGBA, w x h >= 8): image i of a at
// a + i*a_image_bytes (same for b); writes n doubles to d_out.
// Windowed-SSIM launches whose final means are taken later by ONE finish launch (MSSSIM: five levels,
// each finish is ~4.5 us of latency for a few hundred additions).  The caller reserves SLOT_PARTIAL
// for all of them first (SSIM_DEFER_DOUBLES) so that the partial pointers stay valid.
constexpr size_t SSIM_DEFER_DOUBLES = 64 * 1024;
struct SsimDeferred {
    int count = 0;
    size_t used = 0;                       // doubles of SLOT_PARTIAL handed out
    struct Item { size_t offset; int tiles; double windows; int out_index; } item[8];
};
// defer != nullptr (n must be 1): no finish launch; the result goes to d_out[defer_out_index] when
// launch_ssim_finish_deferred runs
int launch_windowed_ssim(fnx_ctx *ctx, int n, const uint8_t *a, int astride, size_t a_image_bytes,
                         const uint8_t *b, int bstride, size_t b_image_bytes, int w, int h,
                         const double *h_window, const double *d_window, double *d_out,
                         SsimDeferred *defer = nullptr, int defer_out_index = 0,
                         const uint8_t *const *d_as = nullptr, const uint8_t *const *d_bs = nullptr);   // device pointer arrays: image z = d_as[z] / d_bs[z]
int launch_ssim_finish_deferred(fnx_ctx *ctx, const SsimDeferred &d, double *d_out, int nimg = 1, size_t part_img = 0, int out_img = 0);
// MSSSIM's levels in five launches (ssim.hip); FNX_NOOP (nothing launched) for shapes it does not cover.
// d_out[i] = SSIMFast of level i; *nlev = levels the reference's loop visits.
int launch_msssim_fused(fnx_ctx *ctx, const uint8_t *a, int astride, const uint8_t *b, int bstride, int w, int h,
                        int nweights, const double *h_window, double *d_out, int *nlev,
                        int nimg = 1, const uint8_t *const *d_as = nullptr, const uint8_t *const *d_bs = nullptr);
int launch_pixel_ssim(fnx_ctx *ctx, const uint8_t *a, const uint8_t *b, int w, int h,
                      size_t pix_len, double *d_out);
// Analyze's device side (analyze.hip): n images -> d_res[n]; aligned16_ok: every base pointer is 16-byte aligned
int launch_analyze(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs, int sstride, int w, int h,
                   bool aligned16_ok, fnx_analysis *d_res);
// flat Pix scan: *d_flags bit 0 = some alpha != 255, bit 1 = some pixel with r != g or g != b
int launch_analyze_ready_words();
int launch_analyze_var_parts();
int launch_analyze_one(fnx_ctx *ctx, int n, const uint8_t *src, const uint8_t *const *srcs, int sstride, int w, int h,
                       bool aligned16_ok, fnx_analysis *res, double *var_part, uint32_t *ready);
int launch_scan_flags(fnx_ctx *ctx, const uint8_t *pix, size_t pix_len, uint32_t *d_flags);
// the same scan as ONE launch whose workgroups each write their flags (| 0x100) into their own word of host-visible memory
// (h_slots: pinned, launch_scan_flags_slots(ctx) words set to 0xffffffff by the caller, who watches the first *nslots of
// them change): no memset, no copy, no stream synchronisation
int launch_scan_flags_slots(const fnx_ctx *ctx);
int launch_scan_flags_direct(fnx_ctx *ctx, const uint8_t *pix, size_t pix_len, uint32_t *h_slots, int *nslots);
int ssim_done_counters(fnx_ctx *ctx, unsigned **out);   // ssim.hip: zeroed "workgroups finished" words (see there)
constexpr int DONE_SCAN = 2 * 4096 + 16;                // first of the 16 words of analyze.hip's scans
// analyzeFormat's samples: pixels at row-major indices 0, step, 2 step, ... (nsamples of them) as packed NRGBA words
int launch_sample_pixels(fnx_ctx *ctx, const uint8_t *src, int sstride, int w, int h, long long step, uint32_t *d_out, int nsamples);
// applyPalette (+ palettedToNRGBA): palette = n x 4 host bytes (opaque); idx and/or quant may be null
int launch_apply_palette(fnx_ctx *ctx, const uint8_t *src, int sstride, int w, int h, const uint8_t *palette, int n,
                         uint8_t *idx, int istride, uint8_t *quant, int qstride);
// image.YCbCr / image.Gray planes -> NRGBA (convert.hip); device pointers; cb == cr == nullptr: Gray
// boxDownsample(toNRGBARef(planes)) without the image (ssim.hip); *done = false: not its case, convert and downsample instead
int launch_box_downsample_ycc(fnx_ctx *ctx, const uint8_t *y, int ystride, const uint8_t *cb, const uint8_t *cr, int cstride,
                              int ratio, int srcW, int srcH, uint8_t *dst, int dstride, int dstW, int dstH, bool *done);
// four planes of equal geometry (a four-component JPEG, reader.go applyBlack) -> toNRGBARef's image; adobe: the APP14 transform
int launch_cmyk_to_nrgba(fnx_ctx *ctx, const uint8_t *const planes[4], int stride, int adobe, int w, int h, uint8_t *dst, int dstride);
int launch_ycbcr_to_nrgba(fnx_ctx *ctx, const uint8_t *y, int ystride, const uint8_t *cb, const uint8_t *cr,
                          int cstride, int ratio, int w, int h, uint8_t *dst, int dstride);
int launch_orient(fnx_ctx *ctx, const uint8_t *src, int sstride, int w, int h, int orient,
                  uint8_t *dst, int dstride);
// jpeg.hip: plane geometry of the 4:2:0 round trip (Y: ys x yh, Cb / Cr: cs x ch), the quality-independent colour
// conversion + chroma averaging, and fdct / quantise / dequantise / idct of every block at `quality` (in -> out)
void jpeg_plane_dims(int w, int h, int *ys, int *yh, int *cs, int *ch);
int launch_jpeg_ycc(fnx_ctx *ctx, const uint8_t *src, int sstride, int w, int h, uint8_t *yp, uint8_t *cbp, uint8_t *crp);
int launch_jpeg_ycc_planes(fnx_ctx *ctx, const uint8_t *dy, int dys, const uint8_t *dcb, const uint8_t *dcr, int dcs, int ratio, int w, int h,
                           uint8_t *yp, uint8_t *cbp, uint8_t *crp);
void jpeg_header(int w, int h, int quality, std::vector<uint8_t> &out);
int jpeg_entropy_code(fnx_ctx *ctx, int w, int h, int quality, const uint8_t *const planes[3], unsigned long long *totals);
size_t jpeg_ecs_capacity(unsigned long long total_bits);
int jpeg_entropy_pack(fnx_ctx *ctx, int w, int h, unsigned long long total_bits, uint8_t *ecs, unsigned long long *totals);
int launch_jpeg_blocks(fnx_ctx *ctx, int w, int h, int quality, const uint8_t *const in[3], uint8_t *const out[3]);

// jpeg_dec.hip: Huffman decoding tables of one file (tables 0, 1: DC th 0, 1; 2, 3: AC th 0, 1) and what its segments say
constexpr int DEC_FAST_BITS = 11;                    // DecTables::fast is indexed by this many bits of the string
struct DecTables {                                   // tables 0, 1: DC (th 0, 1); 2, 3: AC (th 0, 1)
    uint16_t fast[4][1 << DEC_FAST_BITS];                          // by the next 11 bits: length << 8 | symbol; 0: a longer code
    uint32_t limit[4][18];                           // [L]: (largest code of length L + 1) << (16 - L)
    int32_t delta[4][18];                            // [L]: index of the first value of length L - its code
    uint8_t value[4][256];
};

// The synchronisation passes' form of the same tables: what a symbol DOES to the decoder's state, ready made -- ONE
// entry format for all four tables, so that a wave whose lanes stand in DC and in AC positions makes one LDS look-up per
// symbol instead of two divergent ones (r3: the loop is a chain of dependent LDS round trips, ~600 clocks per symbol).
//   st[t][prefix]   first symbol: bits consumed (code + value) | steps << 8 (DC: 1; AC: r + 1, or 64 for the end of block);
//                   AC tables: the first TWO symbols where the second one's code lies inside the prefix and the first
//                   does not end the block: total bits << 16 | total steps << 24 (else 0).  0: a longer code
struct DecSyncTables {
    uint32_t st[4][1 << DEC_FAST_BITS];
    uint32_t limit[4][18];
    int32_t delta[4][18];
    uint8_t value[4][256];
};

struct JpegFile {
    int w = 0, h = 0;
    int ncomp = 3;               // 3: image.YCbCr; 1: image.Gray; 4: image.CMYK (Adobe CMYK / YCbCrK, every component 1 x 1: the host route)
    int adobe = -1;              // the APP14 segment's transform byte (-1: no such segment)
    int ratio = 0;               // image.YCbCrSubsampleRatio: 0 4:4:4, 1 4:2:2, 2 4:2:0, 3 4:4:0, 4 4:1:1, 5 4:1:0; -1: one component
    int hy = 1, vy = 1;          // Y blocks per MCU across / down
    int nslots = 3;              // blocks per MCU
    int ri = 0;                  // MCUs per restart interval (DRI); 0: none
    int mx = 0, my = 0;          // MCUs per row / column
    uint64_t dcpack = 0, acpack = 0;   // Huffman table of MCU slot s: (pack >> 4 s) & 15 (up to 4 x 2 + 2 slots)
    uint16_t q[4][64];           // per component, natural order
    size_t scan = 0;             // offset of the entropy-coded segment in the file
    int rounds = 0;              // cross-workgroup synchronisation rounds the decode took
    bool progressive = false;    // the scans are entropy-decoded on the host (jpeg_prog.cpp: SOF2, SOF1, sequential scans the device has no form for), the image is made on the device
    // the frame header as jpeg_parse read and judged it: jpeg_prog.cpp takes the frame from HERE and never reads a SOF segment
    // itself (two readings of one file by two sets of marker rules was a heap overflow: ADVICE r5)
    bool sof_sequential = false; // SOF0 / SOF1 (true) or SOF2 (false)
    int comp_id[4] = {0, 0, 0, 0}, comp_h[4] = {1, 1, 1, 1}, comp_v[4] = {1, 1, 1, 1}, comp_q[4] = {0, 0, 0, 0};
    DecTables tab;
};
int jpeg_parse(const uint8_t *data, size_t n, JpegFile *f);
// a progressive file's coefficients over all its scans into coef (zeroed by the caller; [mx my nslots][64] int16, blocks in the
// order of an interleaved scan, natural order inside a block, DC as it is), the quantisation tables in force at EOI into f->q
int jpeg_progressive_coefficients(const uint8_t *data, size_t n, JpegFile *f, int16_t *coef);
// the host route's size limit (FNX_JPEG_HOST_MAX_BLOCKS in the header): a block costs 128 bytes of pinned host memory + 8 of
// mask before a single scan bit is validated, so a header may not promise more than this many (4 M blocks = 512 MB pinned:
// 16K x 8K at 4:2:0, 8K x 8K at 4:4:4); above it FNX_ERR_UNSUPPORTED -- the host codec's call
constexpr long long JPEG_HOST_MAX_BLOCKS = 1ll << 22;
int jpeg_unsupported(const char *what);     // set_error + FNX_ERR_UNSUPPORTED
int jpeg_corrupt(const char *what);         // set_error + FNX_ERR_INVALID
// the scan's bytes without the stuffing into dst (capacity: n - f.scan); *nbytes = what was written
// rst: the byte offsets (in dst) at which restart intervals 1, 2, ... start
int jpeg_unstuff(const uint8_t *data, size_t n, const JpegFile &f, uint8_t *dst, size_t *nbytes, std::vector<uint32_t> *rst);
int jpeg_decode_planes(fnx_ctx *ctx, const uint8_t *data, size_t n, JpegFile *f, uint8_t *planes[4], int *ystride, int *cstride);
int launch_scan(fnx_ctx *ctx, const uint32_t *in, unsigned long long *out, unsigned long long *totals, int n, unsigned long long *grand);

}  // namespace fnx
