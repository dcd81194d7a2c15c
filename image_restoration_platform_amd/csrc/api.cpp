// api.cpp -- the C ABI of libire.so (include/ire.h): argument checks, exception -> status
// translation, thread-local error text, and the async batcher behind ire_submit/ire_poll.
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <thread>

#include <pthread.h>
#include <sched.h>

#include <cstdlib>

#include "affinity.hpp"
#include "batcher.hpp"
#include "encode.hpp"
#include "engine.hpp"
#include "fusion.hpp"
#include "strips.hpp"

namespace ire {

static thread_local std::string g_last_error;
void set_last_error(int code, const std::string& msg) { (void)code; g_last_error = msg; }

// ---- async batcher (batcher.hpp): the state machine is a template over its backend; this file provides the HIP one -------

// One engine call = one critical section on the host (mutex) AND on the GPU (Engine::enter/leave): the caller's stream first
// waits for the previous call's completion event, so calls from several threads / streams never interleave kernels on
// the shared workspaces.  leave() also runs when f throws: whatever was enqueued before the failure stays fenced.
template <typename F>
static void on_stream(Engine& E, hipStream_t s, F&& f) {
    std::lock_guard<std::mutex> lk(E.mutex());
    E.enter(s);
    try { f(); } catch (...) { E.leave(s); throw; }
    E.leave(s);
}

struct DeviceGuard {       // staging is allocated on first use of a shape, possibly from a caller's thread: leave that thread's current device as it was
    int prev = -1;
    explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); (void)hipSetDevice(dev); }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct HipSlot {
    uint8_t *d_in = nullptr, *d_out = nullptr, *d_jp = nullptr, *d_txt = nullptr;      // d_txt: IRE_FLAG_RESULT_PNG_BASE64, the results' text
    hipEvent_t ev_in = nullptr, ev_c0 = nullptr, ev_c1 = nullptr, ev_cw = nullptr, ev_out = nullptr;
    // ev_c1 and ev_cw mark the same point (the batch's compute is done): the launcher QUERIES ev_c1 under the batcher's lock while
    // the completer SLEEPS on ev_cw -- hipEventSynchronize holds the event's own lock for as long as it waits, so a query of the
    // same event from another thread blocks until the batch is done (measured: with one event for both, every submitter stalled
    // behind the launcher for a whole batch and a 16-deep closed loop fell to batches of one, 464 img/s)
};

// CPU set of this engine's service threads (affinity.hpp): IRE_CPU_AFFINITY = "off" | a cpulist overrides the sysfs plan
static CpuPlan plan_for_device(int device) {
    CpuPlan p;
    const char* env = std::getenv("IRE_CPU_AFFINITY");
    if (env && (!std::strcmp(env, "off") || !std::strcmp(env, "0"))) return p;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), device) == hipSuccess) p = affinity_plan("/sys", bdf);
    if (env && env[0]) { const auto r = parse_cpulist(env); if (!r.empty()) p.ranges = r; return p; }
    // A container's cpuset (a job that was given 16 of the host's CPUs) outranks the plan: only the planned CPUs this process may
    // run on count, and a share of fewer than four of them (launcher + completer + the host's own waiter threads would queue behind
    // each other) is no plan at all -- the threads then stay wherever the scheduler is allowed to put them.
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
        std::vector<std::pair<int, int>> keep;
        int n = 0;
        for (int c : p.cpus())
            if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) {
                if (!keep.empty() && keep.back().second == c - 1) keep.back().second = c; else keep.emplace_back(c, c);
                ++n;
            }
        if (n < 4) keep.clear();
        p.ranges = keep;
    }
    return p;
}

static void bind_this_thread(const CpuPlan& p) {
    const auto cpus = p.cpus();
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);      // (a cpuset that excludes them all: EINVAL, the thread stays where it was)
}

struct HipBatchBackend {
    Engine& E;
    int device;
    CpuPlan plan;
    hipStream_t cs = nullptr, os = nullptr;      // copy-in | copy-out
    HipBatchBackend(Engine& e, int dev) : E(e), device(dev), plan(plan_for_device(dev)) {}
    ~HipBatchBackend() {
        DeviceGuard g(device);
        if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
        if (os) { (void)hipStreamSynchronize(os); (void)hipStreamDestroy(os); }
    }
    int max_batch() const { return E.max_batch(); }
    bool text() const { return (E.flags() & IRE_FLAG_RESULT_PNG_BASE64) != 0; }
    size_t out_bytes(int h, int w) const { return text() ? png_base64_chars(h, w) : (size_t)h * w * 3; }
    void start() {
        DeviceGuard g(device);
        if (!cs) IRE_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        if (!os) IRE_HIP(hipStreamCreateWithFlags(&os, hipStreamNonBlocking));
    }
    void thread_enter(const char*) { (void)hipSetDevice(device); bind_this_thread(plan); }
    // Everything new goes into locals first and is committed only when all of it exists: a failure half way (out of pinned
    // memory at the fifth slot) frees the locals and leaves the slot exactly as it was -- usable at its old capacity, or empty.
    void reserve(SlotBufs& b, size_t bytes, int mb) {
        if (b.fixed && bytes <= b.cap) return;
        DeviceGuard g(device);
        HipSlot nh;
        uint8_t *pj = nullptr, *pi = nullptr, *po = nullptr, *di = nullptr, *dout = nullptr, *dtxt = nullptr;
        double *ps = nullptr, *psi = nullptr;
        const bool want_fixed = !b.fixed, want_img = bytes > b.cap;
        try {
            if (want_fixed) {
                IRE_HIP(hipEventCreateWithFlags(&nh.ev_in, hipEventDisableTiming));
                IRE_HIP(hipEventCreate(&nh.ev_c0));
                IRE_HIP(hipEventCreate(&nh.ev_c1));
                IRE_HIP(hipEventCreateWithFlags(&nh.ev_cw, hipEventDisableTiming));
                IRE_HIP(hipEventCreateWithFlags(&nh.ev_out, hipEventDisableTiming));
                IRE_HIP(hipHostMalloc((void**)&pj, (size_t)mb));
                IRE_HIP(hipHostMalloc((void**)&ps, sizeof(double) * 7 * (size_t)mb));
                IRE_HIP(hipHostMalloc((void**)&psi, sizeof(double) * 7 * (size_t)mb));
                IRE_HIP(hipMalloc((void**)&nh.d_jp, (size_t)mb));
            }
            if (want_img) {
                IRE_HIP(hipHostMalloc((void**)&pi, bytes));
                IRE_HIP(hipHostMalloc((void**)&po, bytes));
                IRE_HIP(hipMalloc((void**)&di, bytes));
                IRE_HIP(hipMalloc((void**)&dout, bytes));
                if (text()) IRE_HIP(hipMalloc((void**)&dtxt, bytes + 256 * (size_t)mb));
            }
        } catch (...) {
            if (nh.ev_in) (void)hipEventDestroy(nh.ev_in);
            if (nh.ev_c0) (void)hipEventDestroy(nh.ev_c0);
            if (nh.ev_c1) (void)hipEventDestroy(nh.ev_c1);
            if (nh.ev_cw) (void)hipEventDestroy(nh.ev_cw);
            if (nh.ev_out) (void)hipEventDestroy(nh.ev_out);
            if (pj) (void)hipHostFree(pj);
            if (ps) (void)hipHostFree(ps);
            if (psi) (void)hipHostFree(psi);
            if (nh.d_jp) (void)hipFree(nh.d_jp);
            if (pi) (void)hipHostFree(pi);
            if (po) (void)hipHostFree(po);
            if (di) (void)hipFree(di);
            if (dout) (void)hipFree(dout);
            if (dtxt) (void)hipFree(dtxt);
            throw;
        }
        HipSlot* hs = static_cast<HipSlot*>(b.impl);
        if (!hs) { hs = new HipSlot(); b.impl = hs; }
        if (want_fixed) {
            hs->ev_in = nh.ev_in; hs->ev_c0 = nh.ev_c0; hs->ev_c1 = nh.ev_c1; hs->ev_cw = nh.ev_cw; hs->ev_out = nh.ev_out; hs->d_jp = nh.d_jp;
            b.pin_jp = pj; b.pin_sc = ps; b.pin_sc_in = psi; b.fixed = true;
        }
        if (want_img) {
            if (b.pin_in) { (void)hipHostFree(b.pin_in); (void)hipHostFree(b.pin_out); (void)hipFree(hs->d_in); (void)hipFree(hs->d_out); if (hs->d_txt) (void)hipFree(hs->d_txt); }
            b.pin_in = pi; b.pin_out = po; hs->d_in = di; hs->d_out = dout; hs->d_txt = dtxt; b.cap = bytes;
        }
    }
    void release(SlotBufs& b) noexcept {
        DeviceGuard g(device);
        HipSlot* hs = static_cast<HipSlot*>(b.impl);
        if (b.pin_in) { (void)hipHostFree(b.pin_in); (void)hipHostFree(b.pin_out); }
        if (b.fixed) { (void)hipHostFree(b.pin_jp); (void)hipHostFree(b.pin_sc); (void)hipHostFree(b.pin_sc_in); }
        if (hs) {
            if (hs->d_in) { (void)hipFree(hs->d_in); (void)hipFree(hs->d_out); if (hs->d_txt) (void)hipFree(hs->d_txt); }
            if (hs->ev_in) { (void)hipEventDestroy(hs->ev_in); (void)hipEventDestroy(hs->ev_c0); (void)hipEventDestroy(hs->ev_c1); (void)hipEventDestroy(hs->ev_cw); (void)hipEventDestroy(hs->ev_out); (void)hipFree(hs->d_jp); }
            delete hs;
        }
        b = SlotBufs{};
    }
    void h2d(SlotBufs& b, size_t off, size_t bytes) {
        HipSlot& hs = *static_cast<HipSlot*>(b.impl);
        IRE_HIP(hipMemcpyAsync(hs.d_in + off, b.pin_in + off, bytes, hipMemcpyHostToDevice, cs));
    }
    void launch(SlotBufs& b, int n, int h, int w, const uint8_t* has_sc) {
        HipSlot& hs = *static_cast<HipSlot*>(b.impl);
        const size_t ib = (size_t)h * w * 3;
        IRE_HIP(hipMemcpyAsync(hs.d_jp, b.pin_jp, (size_t)n, hipMemcpyHostToDevice, cs));
        IRE_HIP(hipEventRecord(hs.ev_in, cs));
        hipStream_t ms = E.main_stream();
        on_stream(E, ms, [&] {
            IRE_HIP(hipStreamWaitEvent(ms, hs.ev_in, 0));
            IRE_HIP(hipEventRecord(hs.ev_c0, ms));
            E.restore_device_mixed(hs.d_in, n, h, w, b.pin_sc_in, has_sc, hs.d_jp, hs.d_out, ms);   // classifies the jobs that brought no scores
            IRE_HIP(hipMemcpyAsync(b.pin_sc, E.scores_device(), sizeof(double) * 7 * n, hipMemcpyDeviceToHost, ms));
            if (text()) E.encode_png_base64_device(hs.d_out, n, h, w, hs.d_txt, (out_bytes(h, w) + 255) / 256 * 256, ms);     // the results leave the device as text
            IRE_HIP(hipEventRecord(hs.ev_c1, ms));
            IRE_HIP(hipEventRecord(hs.ev_cw, ms));
        });
        IRE_HIP(hipStreamWaitEvent(os, hs.ev_c1, 0));
        if (text()) {
            const size_t ob = out_bytes(h, w), st = (ob + 255) / 256 * 256;
            for (int i = 0; i < n; ++i) IRE_HIP(hipMemcpyAsync(b.pin_out + ob * i, hs.d_txt + st * i, ob, hipMemcpyDeviceToHost, os));
        } else IRE_HIP(hipMemcpyAsync(b.pin_out, hs.d_out, ib * n, hipMemcpyDeviceToHost, os));
        IRE_HIP(hipEventRecord(hs.ev_out, os));
    }
    bool computing(SlotBufs& b) noexcept { return hipEventQuery(static_cast<HipSlot*>(b.impl)->ev_c1) == hipErrorNotReady; }
    void wait_compute(SlotBufs& b) noexcept { (void)hipEventSynchronize(static_cast<HipSlot*>(b.impl)->ev_cw); }
    void wait_done(SlotBufs& b, ire_timings& t) {
        HipSlot& hs = *static_cast<HipSlot*>(b.impl);
        const hipError_t rc = hipEventSynchronize(hs.ev_out);
        if (rc != hipSuccess) throw Error{IRE_ERR_INTERNAL, std::string("internal: ") + hipGetErrorString(rc)};
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, hs.ev_c0, hs.ev_c1) == hipSuccess) { t.restore_ms = ms; t.total_ms = ms; }
    }
    void drain() noexcept {
        DeviceGuard g(device);
        if (cs) (void)hipStreamSynchronize(cs);
        (void)hipStreamSynchronize(E.main_stream());
        if (os) (void)hipStreamSynchronize(os);
    }
};

}  // namespace ire

struct ire_job { std::shared_ptr<ire::Job> j; };
// A strip session belongs to an engine: ire_shutdown and ire_load_weights must not pull the engine (or its layer program)
// from under an open session.  The engine keeps a registry of its open sessions; shutdown closes their inner objects (the
// caller's handle stays valid as an empty shell: every later call on it returns IRE_ERR_INVALID_INPUT "invalid strip
// session handle" and ire_strips_close only frees the shell), load_weights refuses while sessions are open.
struct ire_strips { std::unique_ptr<ire::StripSession> s; ire_engine* owner = nullptr; };
static std::mutex g_strips_mu;       // guards every ire_engine::sessions list and every ire_strips::s / owner

struct ire_engine {
    std::unique_ptr<ire::Engine> eng;
    std::unique_ptr<ire::HipBatchBackend> backend;                  // (declared before the batcher: destroyed after it)
    std::unique_ptr<ire::Batcher<ire::HipBatchBackend>> batcher;
    int device = 0;
    std::vector<ire_strips*> sessions;     // open strip sessions (g_strips_mu)
    ~ire_engine();
};

ire_engine::~ire_engine() {
    batcher.reset();          // launches what is gathered, completes what is in flight, frees the staging
    backend.reset();
    std::lock_guard<std::mutex> lk(g_strips_mu);
    for (ire_strips* s : sessions) {       // invalidate: the StripSession dies with its engine, the caller's shell survives
        if (eng) { std::lock_guard<std::mutex> lk2(eng->mutex()); s->s.reset(); } else s->s.reset();
        s->owner = nullptr;
    }
    sessions.clear();
}

namespace ire {

template <typename F>
static int guarded(F&& f) {
    try {
        f();
        return IRE_OK;
    } catch (const Error& e) {
        set_last_error(e.code, e.msg);
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error(IRE_ERR_UNAVAILABLE, "service unavailable: out of host memory");
        return IRE_ERR_UNAVAILABLE;
    } catch (const std::exception& e) {
        set_last_error(IRE_ERR_INTERNAL, std::string("internal: ") + e.what());
        return IRE_ERR_INTERNAL;
    }
}

static Engine& eng(ire_engine* e) {
    if (!e || !e->eng) fail(IRE_ERR_INVALID_INPUT, "invalid engine handle");
    return *e->eng;
}

static int family_id(const char* f) {
    if (!f) fail(IRE_ERR_INVALID_INPUT, "invalid family");
    const char* names[] = {"classifier", "conv3x3", "conv1x1", "stem", "head", "gn_finalize", "fusion"};
    for (int i = 0; i < 7; ++i) if (!std::strcmp(f, names[i])) return i;
    if (!std::strcmp(f, "all")) return -1;
    fail(IRE_ERR_INVALID_INPUT, "invalid family name");
}

}  // namespace ire

using namespace ire;

extern "C" {

int ire_abi_version(void) { return IRE_ABI_VERSION; }

const char* ire_last_error(void) { return g_last_error.c_str(); }

int ire_init(const ire_config* cfg, ire_engine** out) {
    return guarded([&] {
        if (!cfg || !out) fail(IRE_ERR_INVALID_INPUT, "invalid arguments to ire_init");
        if (cfg->struct_size < sizeof(ire_config)) fail(IRE_ERR_INVALID_INPUT, "invalid ire_config.struct_size");
        if (cfg->flags & ~(uint32_t)IRE_FLAG_RESULT_PNG_BASE64) fail(IRE_ERR_INVALID_INPUT, "invalid ire_config.flags (unknown bits set)");
        *out = nullptr;
        std::unique_ptr<ire_engine> E(new ire_engine());
        E->eng.reset(new Engine(*cfg));
        E->device = cfg->device_index;
        E->backend.reset(new HipBatchBackend(*E->eng, E->device));
        E->batcher.reset(new Batcher<HipBatchBackend>(*E->backend));
        *out = E.release();
    });
}

void ire_shutdown(ire_engine* e) { delete e; }

int ire_load_weights(ire_engine* e, const void* blob, size_t bytes) {
    return guarded([&] {
        Engine& E = eng(e);
        if (!blob) fail(IRE_ERR_INVALID_INPUT, "invalid weight blob");
        {
            std::lock_guard<std::mutex> lk(g_strips_mu);     // a session walks the layer program op by op: it must not change under it
            if (!e->sessions.empty()) fail(IRE_ERR_INVALID_INPUT, "invalid call: ire_load_weights while strip sessions are open (close them first)");
        }
        std::lock_guard<std::mutex> lk(E.mutex());
        E.load_weights(blob, bytes);
    });
}

int ire_max_batch_for(ire_engine* e, int h, int w) {
    if (!e || !e->eng) return 0;
    std::lock_guard<std::mutex> lk(e->eng->mutex());
    return e->eng->capacity_for(h, w);
}

int ire_classify(ire_engine* e, const uint8_t* rgb, int n, int h, int w, int row_stride, const uint8_t* is_jpeg,
                 double* scores_out, int32_t* label_out) {
    return guarded([&] {
        Engine& E = eng(e);
        on_stream(E, E.main_stream(), [&] { E.classify_host(rgb, n, h, w, row_stride, is_jpeg, scores_out, label_out); });
    });
}

int ire_restore(ire_engine* e, const uint8_t* rgb, int n, int h, int w, const double* scores, const uint8_t* is_jpeg,
                uint8_t* out_rgb, ire_timings* t) {
    return guarded([&] {
        Engine& E = eng(e);
        on_stream(E, E.main_stream(), [&] { E.restore_host(rgb, n, h, w, scores, is_jpeg, out_rgb, t); });
    });
}

int ire_fuse(ire_engine* e, const uint8_t* rgb_views, int k, int h, int w, double noise_score, uint8_t* out_rgb,
             int32_t* shifts_out, ire_timings* t) {
    return guarded([&] {
        Engine& E = eng(e);
        on_stream(E, E.main_stream(), [&] { fuse_host(E, rgb_views, k, h, w, noise_score, out_rgb, shifts_out, t); });
    });
}

int ire_classify_device(ire_engine* e, const uint8_t* d_rgb, int n, int h, int w, const uint8_t* d_is_jpeg,
                        double* d_scores, int32_t* d_label, void* stream) {
    return guarded([&] {
        Engine& E = eng(e);
        on_stream(E, (hipStream_t)stream, [&] { E.classify_device(d_rgb, n, h, w, d_is_jpeg, d_scores, d_label, (hipStream_t)stream); });
    });
}

int ire_restore_device(ire_engine* e, const uint8_t* d_rgb, int n, int h, int w, const double* d_scores,
                       const uint8_t* d_is_jpeg, uint8_t* d_out_rgb, void* stream) {
    return guarded([&] {
        Engine& E = eng(e);
        on_stream(E, (hipStream_t)stream, [&] { E.restore_device(d_rgb, n, h, w, d_scores, d_is_jpeg, d_out_rgb, (hipStream_t)stream); });
    });
}

int ire_fuse_device(ire_engine* e, const uint8_t* d_rgb_views, int k, int h, int w, double noise_score,
                    uint8_t* d_out_rgb, int32_t* d_shifts, void* stream) {
    return guarded([&] {
        Engine& E = eng(e);
        on_stream(E, (hipStream_t)stream, [&] { fuse_device(E, d_rgb_views, k, h, w, noise_score, d_out_rgb, d_shifts, (hipStream_t)stream); });
    });
}

int ire_fuse_batch_device(ire_engine* e, const uint8_t* d_rgb_views, int nsets, int k, int h, int w, const double* noise_scores,
                          uint8_t* d_out_rgb, int32_t* d_shifts, void* stream) {
    return guarded([&] {
        Engine& E = eng(e);
        on_stream(E, (hipStream_t)stream, [&] { fuse_batch_device(E, d_rgb_views, nsets, k, h, w, noise_scores, d_out_rgb, d_shifts, (hipStream_t)stream); });
    });
}

int ire_preprocess_plan(int width, int height, int orientation, int max_dim, int* out_w, int* out_h, int* resized) {
    return guarded([&] {
        if (!out_w || !out_h) fail(IRE_ERR_INVALID_INPUT, "invalid arguments to ire_preprocess_plan");
        preprocess_plan(width, height, orientation, max_dim, out_w, out_h, resized);
    });
}

int ire_preprocess(ire_engine* e, const uint8_t* rgb, int h, int w, int orientation, int max_dim, uint8_t* out_rgb, int out_h,
                   int out_w) {
    return guarded([&] {
        Engine& E = eng(e);
        on_stream(E, E.main_stream(), [&] { E.preprocess_host(rgb, h, w, orientation, max_dim, out_rgb, out_h, out_w); });
    });
}

int ire_preprocess_device(ire_engine* e, const uint8_t* d_rgb, int h, int w, int orientation, int max_dim, uint8_t* d_out_rgb,
                          int out_h, int out_w, void* stream) {
    return guarded([&] {
        Engine& E = eng(e);
        on_stream(E, (hipStream_t)stream, [&] { E.preprocess_device(d_rgb, h, w, orientation, max_dim, d_out_rgb, out_h, out_w, (hipStream_t)stream); });
    });
}

size_t ire_png_base64_bytes(int h, int w) { return (h >= 1 && w >= 8 && w % 8 == 0 && h <= 16384 && w <= 16384) ? png_base64_chars(h, w) : 0; }

int ire_encode_png_base64_device(ire_engine* e, const uint8_t* d_rgb, int n, int h, int w, uint8_t* d_chars, size_t stride_bytes, void* stream) {
    return guarded([&] {
        Engine& E = eng(e);
        on_stream(E, (hipStream_t)stream, [&] { E.encode_png_base64_device(d_rgb, n, h, w, d_chars, stride_bytes, (hipStream_t)stream); });
    });
}

int ire_encode_png_base64(ire_engine* e, const uint8_t* rgb, int n, int h, int w, uint8_t* chars, size_t stride_bytes) {
    return guarded([&] {
        Engine& E = eng(e);
        on_stream(E, E.main_stream(), [&] { E.encode_png_base64_host(rgb, n, h, w, chars, stride_bytes); });
    });
}

int ire_restore_tiled_device(ire_engine* e, const uint8_t* d_rgb, int h, int w, int nstrips, const double* d_scores,
                             const uint8_t* d_is_jpeg, uint8_t* d_out_rgb, void* stream) {
    return guarded([&] {
        Engine& E = eng(e);
        on_stream(E, (hipStream_t)stream, [&] { E.restore_tiled_device(d_rgb, h, w, nstrips, d_scores, d_is_jpeg, d_out_rgb, (hipStream_t)stream); });
    });
}

size_t ire_strips_stats_bytes(int h, int w) { return (h > 0 && w > 0) ? StripSession::stats_floats(h, w) * 4 : 0; }

int ire_strips_open(ire_engine* e, int h, int w, int nstrips_total, int first_strip, int nlocal, void* d_stats_all, ire_strips** out) {
    return guarded([&] {
        Engine& E = eng(e);
        if (!out) fail(IRE_ERR_INVALID_INPUT, "invalid arguments to ire_strips_open");
        *out = nullptr;
        std::unique_ptr<ire_strips> S(new ire_strips());
        {
            std::lock_guard<std::mutex> lk(E.mutex());
            S->s.reset(new StripSession(E, h, w, nstrips_total, first_strip, nlocal, (float*)d_stats_all));
        }
        std::lock_guard<std::mutex> lk(g_strips_mu);
        S->owner = e;
        e->sessions.push_back(S.get());
        *out = S.release();
    });
}

void ire_strips_close(ire_strips* s) {
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(g_strips_mu);
        if (s->owner) {       // the engine is alive: unregister, then free the session under the engine's lock
            auto& v = s->owner->sessions;
            for (size_t i = 0; i < v.size(); ++i) if (v[i] == s) { v.erase(v.begin() + i); break; }
            if (s->s) { std::lock_guard<std::mutex> lk2(s->s->engine().mutex()); s->s.reset(); }
            s->owner = nullptr;
        }
    }
    delete s;
}

static StripSession& strips_of(ire_strips* s) {
    if (!s || !s->s) fail(IRE_ERR_INVALID_INPUT, "invalid strip session handle");
    return *s->s;
}

int ire_strips_num_ops(ire_strips* s) { return (s && s->s) ? s->s->num_ops() : 0; }

int ire_strips_set_input(ire_strips* s, const uint8_t* d_rows_with_halo, const double* d_scores, void* stream) {
    return guarded([&] {
        StripSession& S = strips_of(s);
        on_stream(S.engine(), (hipStream_t)stream, [&] { S.set_input(d_rows_with_halo, d_scores, (hipStream_t)stream); });
    });
}

int ire_strips_run_op(ire_strips* s, int k, void* stream, ire_strip_xchg* info) {
    return guarded([&] {
        StripSession& S = strips_of(s);
        on_stream(S.engine(), (hipStream_t)stream, [&] { S.run_op(k, (hipStream_t)stream, info); });
    });
}

int ire_strips_pack_halo(ire_strips* s, int k, uint8_t* d_send_up, uint8_t* d_send_down, void* stream) {
    return guarded([&] {
        StripSession& S = strips_of(s);
        on_stream(S.engine(), (hipStream_t)stream, [&] { S.pack_halo(k, d_send_up, d_send_down, (hipStream_t)stream); });
    });
}

int ire_strips_unpack_halo(ire_strips* s, int k, const uint8_t* d_recv_up, const uint8_t* d_recv_down, void* stream) {
    return guarded([&] {
        StripSession& S = strips_of(s);
        on_stream(S.engine(), (hipStream_t)stream, [&] { S.unpack_halo(k, d_recv_up, d_recv_down, (hipStream_t)stream); });
    });
}

int ire_strips_get_output(ire_strips* s, uint8_t* d_out_rows, void* stream) {
    return guarded([&] {
        StripSession& S = strips_of(s);
        if (!d_out_rows) fail(IRE_ERR_INVALID_INPUT, "invalid output pointer");
        on_stream(S.engine(), (hipStream_t)stream, [&] { S.get_output(d_out_rows, (hipStream_t)stream); });
    });
}

int ire_get_stats(ire_engine* e, ire_engine_stats* out) {
    return guarded([&] {
        Engine& E = eng(e);
        if (!out || out->struct_size < sizeof(ire_engine_stats)) fail(IRE_ERR_INVALID_INPUT, "invalid ire_engine_stats.struct_size");
        {
            std::lock_guard<std::mutex> lk(E.mutex());
            E.get_stats(out);
        }
        out->queue_depth = e->batcher->queue_depth();
    });
}

int ire_submit(ire_engine* e, const uint8_t* rgb, int h, int w, int is_jpeg, const double* scores, ire_job** job_out) {
    return guarded([&] {
        eng(e);
        if (!rgb || !job_out) fail(IRE_ERR_INVALID_INPUT, "invalid arguments to ire_submit");
        if (h <= 0 || w <= 0 || h % 8 || w % 8 || h < 16 || w < 16 || h > 8192 || w > 8192)
            fail(IRE_ERR_INVALID_INPUT, "invalid image size for restore: height and width must be multiples of 8, >= 16");
        *job_out = nullptr;
        std::unique_ptr<ire_job> hnd(new ire_job{});
        hnd->j = e->batcher->submit(rgb, h, w, is_jpeg, scores);
        *job_out = hnd.release();
    });
}

int ire_poll(ire_engine* e, ire_job* job, int timeout_ms, uint8_t* out_rgb, double* scores_out, ire_timings* t) {
    return guarded([&] {
        eng(e);
        if (!job || !job->j) fail(IRE_ERR_INVALID_INPUT, "invalid job handle");
        std::string err;
        const int st = e->batcher->poll(job->j, timeout_ms, out_rgb, scores_out, t, &err);
        if (st == IRE_ERR_TIMEOUT) fail(IRE_ERR_TIMEOUT, "timeout: job still pending");       // the handle stays valid: poll again or ire_job_release
        delete job;
        if (st != IRE_OK) fail(st, err);
    });
}

int ire_job_release(ire_engine* e, ire_job* job) {
    return guarded([&] {
        if (!job) return;
        // e == NULL: the engine is gone (ire_shutdown with jobs outstanding): only the handle is left to free
        if (e && e->batcher && job->j) e->batcher->release(job->j);
        delete job;
    });
}

int ire_affinity_plan(const char* sysfs_root, const char* pci_bdf, char* cpulist_out, size_t cap, int32_t* numa_node_out,
                      int32_t* slot_out, int32_t* nslots_out) {
    return guarded([&] {
        if (!pci_bdf || !cpulist_out || cap == 0) fail(IRE_ERR_INVALID_INPUT, "invalid arguments to ire_affinity_plan");
        const CpuPlan p = affinity_plan(sysfs_root && sysfs_root[0] ? sysfs_root : "/sys", pci_bdf);
        const std::string l = p.cpulist();
        if (l.size() + 1 > cap) fail(IRE_ERR_INVALID_INPUT, "invalid buffer size for ire_affinity_plan");
        std::memcpy(cpulist_out, l.c_str(), l.size() + 1);
        if (numa_node_out) *numa_node_out = p.numa_node;
        if (slot_out) *slot_out = p.slot;
        if (nslots_out) *nslots_out = p.nslots;
    });
}

int ire_engine_affinity(ire_engine* e, char* cpulist_out, size_t cap, int32_t* numa_node_out) {
    return guarded([&] {
        eng(e);
        if (!cpulist_out || cap == 0) fail(IRE_ERR_INVALID_INPUT, "invalid arguments to ire_engine_affinity");
        const std::string l = e->backend->plan.cpulist();
        if (l.size() + 1 > cap) fail(IRE_ERR_INVALID_INPUT, "invalid buffer size for ire_engine_affinity");
        std::memcpy(cpulist_out, l.c_str(), l.size() + 1);
        if (numa_node_out) *numa_node_out = e->backend->plan.numa_node;
    });
}

int ire_debug_classifier_sums(ire_engine* e, int n, uint64_t* sums_out) {
    return guarded([&] {
        Engine& E = eng(e);
        if (!sums_out) fail(IRE_ERR_INVALID_INPUT, "invalid output pointer");
        std::lock_guard<std::mutex> lk(E.mutex());
        E.debug_sums(n, sums_out);
    });
}

int ire_debug_capture(ire_engine* e, int on) {
    return guarded([&] {
        Engine& E = eng(e);
        std::lock_guard<std::mutex> lk(E.mutex());
        E.debug_capture(on != 0);
    });
}

int ire_debug_activation(ire_engine* e, const char* name, float* out, size_t* count) {
    return guarded([&] {
        Engine& E = eng(e);
        if (!name) fail(IRE_ERR_INVALID_INPUT, "invalid activation name");
        std::lock_guard<std::mutex> lk(E.mutex());
        if (!E.debug_activation(name, out, count)) fail(IRE_ERR_INVALID_INPUT, std::string("invalid activation name: ") + name);
    });
}

int ire_profile_enable(ire_engine* e, int on) {
    return guarded([&] {
        Engine& E = eng(e);
        std::lock_guard<std::mutex> lk(E.mutex());
        E.profile_enable(on);
    });
}

int ire_profile_query(ire_engine* e, const char* family, double* ms_out, int64_t* launches_out, double* flops_out,
                      double* bytes_out) {
    return guarded([&] {
        Engine& E = eng(e);
        const int fam = family_id(family);
        std::lock_guard<std::mutex> lk(E.mutex());
        E.profile_query(fam, ms_out, launches_out, flops_out, bytes_out);
    });
}

int ire_profile_report(ire_engine* e, char* buf, size_t cap, size_t* needed_out) {
    return guarded([&] {
        Engine& E = eng(e);
        std::string r;
        {
            std::lock_guard<std::mutex> lk(E.mutex());
            r = E.profile_report();
        }
        if (needed_out) *needed_out = r.size() + 1;
        if (buf && cap) {
            if (cap < r.size() + 1) fail(IRE_ERR_INVALID_INPUT, "invalid buffer size for ire_profile_report (query *needed_out first)");
            std::memcpy(buf, r.c_str(), r.size() + 1);
        }
    });
}

int ire_profile_reset(ire_engine* e) {
    return guarded([&] {
        Engine& E = eng(e);
        std::lock_guard<std::mutex> lk(E.mutex());
        E.profile_reset();
    });
}

}  // extern "C"
