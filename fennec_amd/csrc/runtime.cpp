// Context, device memory, staging: the runtime under every fnx_* entry point.
#include "common.hpp"

#include <dlfcn.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace fnx {

// ---- which devices the library uses ("one env var / option to force CPU or pick devices") ----------
// Logical device i of this library = HIP device devmap[i].  Default: every HIP device, in order.  FENNEC_HIP_DEVICES
// ("0,2,3") picks and orders them, FENNEC_HIP_DISABLE=1 leaves none -- fnx_device_count() is then 0 and fnx_ctx_create
// returns FNX_ERR_NO_DEVICE, which is what sends the cgo shim (and any caller that honours the status) to its own CPU
// path; the library itself still has none.  fnx_set_devices() does the same from code and wins over the environment.
namespace {
std::mutex g_dev_mu;
bool g_dev_set = false;
std::vector<int> g_devmap;

int hip_device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

void devmap_init_locked()
{
    if (g_dev_set) return;
    g_dev_set = true;
    g_devmap.clear();
    const char *off = std::getenv("FENNEC_HIP_DISABLE");
    if (off && off[0] && off[0] != '0') return;
    const int n = hip_device_count();
    const char *pick = std::getenv("FENNEC_HIP_DEVICES");
    if (pick && pick[0]) {
        const char *p = pick;
        while (*p) {
            char *end = nullptr;
            const long d = std::strtol(p, &end, 10);
            if (end == p) break;
            if (d >= 0 && d < n) g_devmap.push_back(static_cast<int>(d));
            p = *end == ',' ? end + 1 : end;
            if (*end && *end != ',') break;
        }
        return;
    }
    for (int i = 0; i < n; i++) g_devmap.push_back(i);
}

// roctx ranges per exported op, opt-in: FNX_ROCTX=1 resolves roctxRangePushA / roctxRangePop from
// the ROCm tracing library at first use (no link-time dependency); off, a range is one relaxed load and a branch.
typedef int (*roctx_push_fn)(const char *);
typedef int (*roctx_pop_fn)();
std::atomic<int> g_roctx_state{0};        // 0: not looked at, 1: on, 2: off
roctx_push_fn g_roctx_push = nullptr;
roctx_pop_fn g_roctx_pop = nullptr;

bool roctx_on()
{
    int st = g_roctx_state.load(std::memory_order_acquire);
    if (st == 0) {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        st = g_roctx_state.load(std::memory_order_relaxed);
        if (st == 0) {
            st = 2;
            const char *e = std::getenv("FNX_ROCTX");
            if (e && e[0] == '1') {
                for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
                    void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
                    if (!h) continue;
                    g_roctx_push = reinterpret_cast<roctx_push_fn>(dlsym(h, "roctxRangePushA"));
                    g_roctx_pop = reinterpret_cast<roctx_pop_fn>(dlsym(h, "roctxRangePop"));
                    if (g_roctx_push && g_roctx_pop) { st = 1; break; }
                }
            }
            g_roctx_state.store(st, std::memory_order_release);
        }
    }
    return st == 1;
}
}  // namespace

OpRange::OpRange(const char *name) : on_(roctx_on())
{
    if (on_) g_roctx_push(name);
}
OpRange::~OpRange()
{
    if (on_) g_roctx_pop();
}

static const char *const FORM_NAMES[FORM_COUNT] = {"fx_stream", "fx_pairs", "fx_ref", "resize_mfma", "resize_fp64", "resize_fused",
                                                   "msssim_levelwise", "msssim_nofuse0", "msssim_fold", "msssim_boxfly", "palette_grid"};

const char *form_value(const fnx_ctx *ctx, Form f)
{
    if (ctx && ctx->form[f][0]) return ctx->form[f];
#ifdef FNX_DEVELOP
    char name[40] = "FNX_";
    size_t k = 4;
    for (const char *p = FORM_NAMES[f]; *p && k + 1 < sizeof(name); p++) name[k++] = static_cast<char>(*p >= 'a' && *p <= 'z' ? *p - 32 : *p);
    name[k] = 0;
    return std::getenv(name);
#else
    return nullptr;
#endif
}

int form_set(fnx_ctx *ctx, const char *name, const char *value)
{
    for (int f = 0; f < FORM_COUNT; f++) {
        if (std::strcmp(name, FORM_NAMES[f]) != 0) continue;
        if (value && std::strlen(value) >= sizeof(ctx->form[f])) { set_error("fnx_ctx_set_form: value too long"); return FNX_ERR_INVALID; }
        std::memset(ctx->form[f], 0, sizeof(ctx->form[f]));
        if (value) std::strcpy(ctx->form[f], value);
        return FNX_OK;
    }
    set_error("fnx_ctx_set_form: no kernel form called '%s'", name);
    return FNX_ERR_INVALID;
}

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int bind(fnx_ctx *ctx)
{
    if (!ctx) {
        set_error("null ctx");
        return FNX_ERR_INVALID;
    }
    FNX_HIP(hipSetDevice(ctx->device));
    ctx->op_seq++;
    return FNX_OK;
}

// for the calls that only look at the ctx or move it to another stream: they do not come between a blur that kept its box sums
// and the scoring call (api.cpp: KeptBoxes) -- the binding's stream lending sits exactly there
static int bind_quiet(fnx_ctx *ctx)
{
    const int rc = bind(ctx);
    if (rc == FNX_OK) ctx->op_seq--;
    return rc;
}

int scratch(fnx_ctx *ctx, Slot slot, size_t bytes, void **out)
{
    Scratch &s = ctx->slot[slot];
    if (bytes > s.cap) {
        // everything enqueued may still use the old buffer
        FNX_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->stream2_used) FNX_HIP(hipStreamSynchronize(ctx->stream2));
        if (s.p) FNX_HIP(hipFree(s.p));
        s.p = nullptr;
        s.cap = 0;
        size_t cap = bytes + bytes / 4 + 4096;
        cap = (cap + 255) & ~size_t(255);
        FNX_HIP(hipMalloc(&s.p, cap));
        s.cap = cap;
        ctx->tcache[slot].host.clear();
    }
    *out = s.p;
    return FNX_OK;
}

int pinned_alloc(fnx_ctx *ctx, size_t bytes, void **out)
{
    bytes = (bytes + 63) & ~size_t(63);
    if (bytes > ctx->pinned_cap) {
        FNX_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->pinned) FNX_HIP(hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        size_t cap = bytes * 2 > (size_t(4) << 20) ? bytes * 2 : (size_t(4) << 20);
        FNX_HIP(hipHostMalloc(reinterpret_cast<void **>(&ctx->pinned), cap, hipHostMallocDefault));
        ctx->pinned_cap = cap;
        ctx->pinned_off = 0;
    }
    if (ctx->pinned_off + bytes > ctx->pinned_cap) {
        // wrap: earlier slices may still be the source/target of queued copies
        FNX_HIP(hipStreamSynchronize(ctx->stream));
        ctx->pinned_off = 0;
    }
    *out = ctx->pinned + ctx->pinned_off;
    ctx->pinned_off += bytes;
    return FNX_OK;
}

int upload_tables(fnx_ctx *ctx, Slot slot, const void *const *hosts, const size_t *sizes, int n,
                  void **dptrs)
{
    size_t total = 0;
    for (int i = 0; i < n; i++) total += (sizes[i] + 15) & ~size_t(15);
    void *d = nullptr;
    FNX_TRY(scratch(ctx, slot, total ? total : 16, &d));
    // identical to what the slot already holds?  (bench loops / binary searches re-send the same table)
    TableCache &tc = ctx->tcache[slot];
    bool same = tc.host.size() == total;
    if (same) {
        size_t off = 0;
        for (int i = 0; i < n && same; i++) {
            same = std::memcmp(tc.host.data() + off, hosts[i], sizes[i]) == 0;
            off += (sizes[i] + 15) & ~size_t(15);
        }
    }
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        dptrs[i] = static_cast<unsigned char *>(d) + off;
        off += (sizes[i] + 15) & ~size_t(15);
    }
    tc.fresh = !same;
    if (same) return FNX_OK;
    // a one-pass tail still running on the second stream may be reading this slot's previous contents
    if (ctx->stream2_used) FNX_HIP(hipStreamSynchronize(ctx->stream2));
    void *pin = nullptr;
    FNX_TRY(pinned_alloc(ctx, total ? total : 16, &pin));
    tc.host.assign(total, 0);
    off = 0;
    for (int i = 0; i < n; i++) {
        std::memcpy(tc.host.data() + off, hosts[i], sizes[i]);
        off += (sizes[i] + 15) & ~size_t(15);
    }
    std::memcpy(pin, tc.host.data(), total);
    FNX_HIP(hipMemcpyAsync(d, pin, total, hipMemcpyHostToDevice, ctx->stream));
    return FNX_OK;
}

int upload_table(fnx_ctx *ctx, Slot slot, const void *host, size_t bytes, void **dptr)
{
    return upload_tables(ctx, slot, &host, &bytes, 1, dptr);
}

int prof_begin(fnx_ctx *ctx, int cls)
{
    if (!(ctx->prof & cls)) return FNX_OK;
    if (ctx->prof_count == fnx_ctx::PROF_DEPTH) {          // nobody is reading: forget the oldest launch
        ctx->prof_head = (ctx->prof_head + 1) % fnx_ctx::PROF_DEPTH;
        ctx->prof_count--;
    }
    ctx->prof_open = (ctx->prof_head + ctx->prof_count) % fnx_ctx::PROF_DEPTH;
    FNX_HIP(hipEventRecord(ctx->prof_ev[ctx->prof_open][0], ctx->stream));
    return FNX_OK;
}

int prof_bind(fnx_ctx *ctx, int cls, LaunchEvents *ev)
{
    ev->start = ev->stop = nullptr;
    if (!(ctx->prof & cls)) return FNX_OK;
    if (ctx->prof_count == fnx_ctx::PROF_DEPTH) {          // nobody is reading: forget the oldest launch
        ctx->prof_head = (ctx->prof_head + 1) % fnx_ctx::PROF_DEPTH;
        ctx->prof_count--;
    }
    const int slot = (ctx->prof_head + ctx->prof_count) % fnx_ctx::PROF_DEPTH;
    ev->start = ctx->prof_ev[slot][0];
    ev->stop = ctx->prof_ev[slot][1];
    ctx->prof_count++;
    return FNX_OK;
}

int prof_end(fnx_ctx *ctx)
{
    if (!ctx->prof || ctx->prof_open < 0) return FNX_OK;
    FNX_HIP(hipEventRecord(ctx->prof_ev[ctx->prof_open][1], ctx->stream));
    ctx->prof_open = -1;
    ctx->prof_count++;
    return FNX_OK;
}

int stage_in(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, Slot slot,
             DevImg *out)
{
    if (space != FNX_HOST) {   // FNX_DEVICE, FNX_DEVICE_SRC: inputs are device memory
        out->p = src;
        out->stride = sstride;
        return FNX_OK;
    }
    int pitch = pitch16(w);
    void *d = nullptr;
    FNX_TRY(scratch(ctx, slot, size_t(pitch) * h + 16, &d));
    // a tight image (image.NewNRGBA: stride = 4 w) goes up as ONE linear copy: the runtime's 2-D path for pageable memory
    // stages row by row
    if (sstride == pitch) FNX_HIP(hipMemcpyAsync(d, src, size_t(pitch) * h, hipMemcpyHostToDevice, ctx->stream));
    else FNX_HIP(hipMemcpy2DAsync(d, pitch, src, sstride, size_t(w) * 4, h, hipMemcpyHostToDevice, ctx->stream));
    out->p = static_cast<const uint8_t *>(d);
    out->stride = pitch;
    return FNX_OK;
}

int stage_in_flat(fnx_ctx *ctx, int space, const uint8_t *src, int sstride, int w, int h, Slot slot,
                  DevImg *out)
{
    if (space != FNX_HOST) {   // FNX_DEVICE, FNX_DEVICE_SRC: inputs are device memory
        out->p = src;
        out->stride = sstride;
        return FNX_OK;
    }
    const size_t len = static_cast<size_t>(h - 1) * sstride + static_cast<size_t>(w) * 4;
    void *d = nullptr;
    FNX_TRY(scratch(ctx, slot, len + 16, &d));
    FNX_HIP(hipMemcpyAsync(d, src, len, hipMemcpyHostToDevice, ctx->stream));
    out->p = static_cast<const uint8_t *>(d);
    out->stride = sstride;
    return FNX_OK;
}

int stage_in_front(fnx_ctx *ctx, int space, const uint8_t *src, int w, int h, Slot slot, DevImg *out)
{
    out->stride = w * 4;
    if (space != FNX_HOST) {
        out->p = src;
        return FNX_OK;
    }
    const size_t len = static_cast<size_t>(w) * 4 * h;
    void *d = nullptr;
    FNX_TRY(scratch(ctx, slot, len + 16, &d));
    FNX_HIP(hipMemcpyAsync(d, src, len, hipMemcpyHostToDevice, ctx->stream));
    out->p = static_cast<const uint8_t *>(d);
    return FNX_OK;
}

int stage_out(fnx_ctx *ctx, int space, uint8_t *dst, int dstride, int w, int h, Slot slot,
              DevOut *out)
{
    out->w = w;
    out->h = h;
    if (space == FNX_DEVICE) {
        out->p = dst;
        out->stride = dstride;
        out->host = nullptr;
        return FNX_OK;
    }
    int pitch = pitch16(w);
    void *d = nullptr;
    FNX_TRY(scratch(ctx, slot, size_t(pitch) * h + 16, &d));
    out->p = static_cast<uint8_t *>(d);
    out->stride = pitch;
    out->host = dst;
    out->hstride = dstride;
    return FNX_OK;
}

int finish(fnx_ctx *ctx, int space, DevOut *out)
{
    if (space == FNX_DEVICE) return FNX_OK;
    FNX_TRY(finish_enqueue(ctx, space, out));
    FNX_HIP(hipStreamSynchronize(ctx->stream));
    return FNX_OK;
}

// the copy back of a staged output, enqueued on the ctx's stream without waiting for it
int finish_enqueue(fnx_ctx *ctx, int space, DevOut *out)
{
    if (space == FNX_DEVICE) return FNX_OK;
    if (out && out->host && out->w > 0 && out->h > 0) {
        if (out->hstride == out->stride)
            FNX_HIP(hipMemcpyAsync(out->host, out->p, size_t(out->stride) * out->h, hipMemcpyDeviceToHost, ctx->stream));
        else
            FNX_HIP(hipMemcpy2DAsync(out->host, out->hstride, out->p, out->stride, size_t(out->w) * 4,
                                     out->h, hipMemcpyDeviceToHost, ctx->stream));
    }
    return FNX_OK;
}

int fetch_bytes(fnx_ctx *ctx, const void *dptr, void *host, size_t bytes)
{
    void *pin = nullptr;
    FNX_TRY(pinned_alloc(ctx, bytes, &pin));
    FNX_HIP(hipMemcpyAsync(pin, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
    FNX_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(host, pin, bytes);
    return FNX_OK;
}

int fetch_doubles(fnx_ctx *ctx, const double *dptr, double *host, int n)
{
    return fetch_bytes(ctx, dptr, host, sizeof(double) * size_t(n));
}

}  // namespace fnx

using namespace fnx;

extern "C" {

const char *fnx_version(void) { return "fennec-hip 0.1 (gfx950)"; }

int fnx_device_count(void)
{
    std::lock_guard<std::mutex> lk(g_dev_mu);
    devmap_init_locked();
    return static_cast<int>(g_devmap.size());
}

extern "C" void fennec_pool_release(void);   // host_api.cpp: idle worker contexts, keyed by LOGICAL device index

int fnx_set_devices(const int *devices, int n)
{
    // pooled worker contexts were created under the old mapping: they would keep running on GPUs the caller just
    // excluded (and report logical indices of a list that no longer exists)
    fennec_pool_release();
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (n < 0 || (n > 0 && !devices)) {
        if (n < 0 && !devices) {               // (NULL, -1): back to the environment / every device
            g_dev_set = false;
            devmap_init_locked();
            return FNX_OK;
        }
        set_error("invalid argument: fnx_set_devices");
        return FNX_ERR_INVALID;
    }
    const int have = hip_device_count();
    for (int i = 0; i < n; i++)
        if (devices[i] < 0 || devices[i] >= have) {
            set_error("invalid argument: fnx_set_devices names HIP device %d of %d", devices[i], have);
            return FNX_ERR_INVALID;
        }
    g_dev_set = true;
    g_devmap.assign(devices, devices + n);
    return FNX_OK;
}

const char *fnx_last_error(void) { return g_err; }

int fnx_ctx_create(int device, fnx_ctx **out)
{
    FNX_REQUIRE(out != nullptr, "out is null");
    *out = nullptr;
    int n = fnx_device_count();
    if (n <= 0) {
        set_error("no HIP device available (libfennec_hip has no CPU path)");
        return FNX_ERR_NO_DEVICE;
    }
    FNX_REQUIRE(device >= 0 && device < n, "device index out of range");
    int phys;
    {
        std::lock_guard<std::mutex> lk(g_dev_mu);
        FNX_REQUIRE(device < static_cast<int>(g_devmap.size()), "device index out of range");
        phys = g_devmap[static_cast<size_t>(device)];
    }
    FNX_HIP(hipSetDevice(phys));
    fnx_ctx *c = new fnx_ctx();
    c->device = phys;
    c->logical_device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete c;
        return FNX_ERR_HIP;
    }
    c->own_stream = c->stream;
    {
        // the tail stream.  FNX_TAIL_PRIO=low|high moves it to the end of the device's priority range (experiments)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const char *pe = dev_env("FNX_TAIL_PRIO");
        int prio = (least + greatest) / 2;
        if (pe && pe[0] == 'l') prio = least;
        if (pe && pe[0] == 'h') prio = greatest;
        e = pe ? hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio)
               : hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
        for (int i = 0; i < 2 && e == hipSuccess; i++) {
            e = hipEventCreateWithFlags(&c->ev_blur[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_tail[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            set_error("hipStreamCreate (tail stream) failed: %s", hipGetErrorString(e));
            fnx_ctx_destroy(c);
            return FNX_ERR_HIP;
        }
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, phys) == hipSuccess) c->num_cus = prop.multiProcessorCount;
    *out = c;
    return FNX_OK;
}

void fnx_ctx_destroy(fnx_ctx *ctx)
{
    if (!ctx) return;
    if (hipSetDevice(ctx->device) == hipSuccess) {
        // only streams the ctx OWNS: a stream lent through fnx_ctx_use_stream may already be gone (its owner orders and drains it;
        // a lent stream must outlive the work enqueued on it, not the ctx)
        if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
        if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
        for (auto &s : ctx->slot)
            if (s.p) (void)hipFree(s.p);
        if (ctx->pinned) (void)hipHostFree(ctx->pinned);
        for (auto &q : ctx->res_q)
            if (q.ev) (void)hipEventDestroy(q.ev);
        for (auto &rb : ctx->res_buf)
            if (rb.p) (void)hipHostFree(rb.p);
        free_resize_plans(ctx);
        for (auto &pair : ctx->prof_ev)
            for (auto &e : pair)
                if (e) (void)hipEventDestroy(e);
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
        if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
        if (ctx->ev_switch) (void)hipEventDestroy(ctx->ev_switch);
        for (int i = 0; i < 2; i++) {
            if (ctx->ev_blur[i]) (void)hipEventDestroy(ctx->ev_blur[i]);
            if (ctx->ev_tail[i]) (void)hipEventDestroy(ctx->ev_tail[i]);
        }
    }
    delete ctx;
}

int fnx_ctx_device(const fnx_ctx *ctx) { return ctx ? ctx->logical_device : -1; }

int fnx_ctx_profile(fnx_ctx *ctx, int enable)
{
    FNX_TRY(bind_quiet(ctx));
    if (enable)
        for (auto &pair : ctx->prof_ev)
            for (auto &e : pair)
                if (!e) FNX_HIP(hipEventCreate(&e));
    ctx->prof = enable;
    ctx->prof_head = ctx->prof_count = 0;      // (re-)enabling forgets unread launches
    ctx->prof_open = -1;
    return FNX_OK;
}

int fnx_ctx_set_ssim_mode(fnx_ctx *ctx, int mode)
{
    if (!ctx || (mode != FNX_SSIM_EXACT && mode != FNX_SSIM_FAST)) { fnx::set_error("fnx_ctx_set_ssim_mode: bad argument"); return FNX_ERR_INVALID; }
    ctx->ssim_mode = mode;
    return FNX_OK;
}

int fnx_ctx_set_form(fnx_ctx *ctx, const char *name, const char *value)
{
    if (!ctx || !name) { fnx::set_error("fnx_ctx_set_form: null argument"); return FNX_ERR_INVALID; }
    return fnx::form_set(ctx, name, value);
}

const char *fnx_ctx_last_kernel(fnx_ctx *ctx, int prof_class)
{
    if (!ctx || prof_class <= 0 || prof_class > 128 || (prof_class & (prof_class - 1))) return nullptr;
    int i = 0;
    while (!((prof_class >> i) & 1)) i++;
    return ctx->route[i];
}

int fnx_ctx_kernel_ms(fnx_ctx *ctx, float *ms)
{
    FNX_TRY(bind_quiet(ctx));
    FNX_REQUIRE(ms != nullptr && ctx->prof_count > 0, "no unread profiled kernel launch on this ctx");
    hipEvent_t *pair = ctx->prof_ev[ctx->prof_head];
    FNX_HIP(hipEventSynchronize(pair[1]));
    FNX_HIP(hipEventElapsedTime(ms, pair[0], pair[1]));
    ctx->prof_head = (ctx->prof_head + 1) % fnx_ctx::PROF_DEPTH;
    ctx->prof_count--;
    return FNX_OK;
}

void *fnx_ctx_stream(fnx_ctx *ctx) { return ctx ? static_cast<void *>(ctx->stream) : nullptr; }

// Everything already enqueued by the ctx (on the stream it leaves and on its tail stream) is ordered before
// whatever it launches on the new stream: one event hand-over at the switch, nothing per call afterwards.
static int switch_stream(fnx_ctx *ctx, hipStream_t ns)
{
    if (ns == ctx->stream) return FNX_OK;
    if (!ctx->ev_switch) FNX_HIP(hipEventCreateWithFlags(&ctx->ev_switch, hipEventDisableTiming));
    FNX_HIP(hipEventRecord(ctx->ev_switch, ctx->stream));
    FNX_HIP(hipStreamWaitEvent(ns, ctx->ev_switch, 0));
    if (ctx->stream2_used) {
        FNX_HIP(hipEventRecord(ctx->ev_switch, ctx->stream2));
        FNX_HIP(hipStreamWaitEvent(ns, ctx->ev_switch, 0));
    }
    ctx->stream = ns;
    return FNX_OK;
}

int fnx_ctx_use_stream(fnx_ctx *ctx, void *stream)
{
    FNX_TRY(bind_quiet(ctx));
    return switch_stream(ctx, static_cast<hipStream_t>(stream));
}

int fnx_ctx_use_own_stream(fnx_ctx *ctx)
{
    FNX_TRY(bind_quiet(ctx));
    return switch_stream(ctx, ctx->own_stream);
}

int fnx_ctx_sync(fnx_ctx *ctx)
{
    FNX_TRY(bind(ctx));
    FNX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->stream2_used) FNX_HIP(hipStreamSynchronize(ctx->stream2));
    return FNX_OK;
}

int fnx_malloc(fnx_ctx *ctx, size_t bytes, void **dptr)
{
    FNX_TRY(bind(ctx));
    FNX_REQUIRE(dptr != nullptr, "dptr is null");
    FNX_HIP(hipMalloc(dptr, bytes ? bytes : 16));
    return FNX_OK;
}

int fnx_free(fnx_ctx *ctx, void *dptr)
{
    FNX_TRY(bind(ctx));
    if (dptr) {
        FNX_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->stream2_used) FNX_HIP(hipStreamSynchronize(ctx->stream2));
        FNX_HIP(hipFree(dptr));
    }
    return FNX_OK;
}

int fnx_upload(fnx_ctx *ctx, void *dptr, int dstride, const void *host, int hstride, int w, int h)
{
    FNX_TRY(bind(ctx));
    if (w <= 0 || h <= 0) return FNX_OK;
    FNX_REQUIRE(dptr && host && dstride >= 4 * w && hstride >= 4 * w, "upload: null pointer or a stride below 4 * w");
    FNX_HIP(hipMemcpy2DAsync(dptr, dstride, host, hstride, size_t(w) * 4, h, hipMemcpyHostToDevice,
                             ctx->stream));
    FNX_HIP(hipStreamSynchronize(ctx->stream));
    return FNX_OK;
}

int fnx_download(fnx_ctx *ctx, void *host, int hstride, const void *dptr, int dstride, int w, int h)
{
    FNX_TRY(bind(ctx));
    if (w <= 0 || h <= 0) return FNX_OK;
    FNX_REQUIRE(dptr && host && dstride >= 4 * w && hstride >= 4 * w, "download: null pointer or a stride below 4 * w");
    FNX_HIP(hipMemcpy2DAsync(host, hstride, dptr, dstride, size_t(w) * 4, h, hipMemcpyDeviceToHost,
                             ctx->stream));
    FNX_HIP(hipStreamSynchronize(ctx->stream));
    return FNX_OK;
}

}  // extern "C"
