// api.cpp -- the C ABI of libire.so (include/ire.h): argument checks, exception -> status
// translation, thread-local error text, and the async batcher behind ire_submit/ire_poll.
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <thread>

#include "engine.hpp"
#include "fusion.hpp"
#include "strips.hpp"

namespace ire {

static thread_local std::string g_last_error;
void set_last_error(int code, const std::string& msg) { (void)code; g_last_error = msg; }

// ---- async batcher: restoreBatch's in-flight promises (restorator.js:181-236) coalesced into
// engine batches of up to max_batch equal-shape images.
struct Job {
    int h, w, is_jpeg;
    bool has_scores = false;
    std::vector<uint8_t> in, out;
    double scores[7];
    ire_timings t{};
    int status = -1;  // -1 pending, else ire_status
    std::string err;
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
    // batcher
    std::mutex qmu;
    std::condition_variable qcv, dcv;
    std::deque<std::shared_ptr<ire::Job>> queue;
    std::thread worker;
    bool stop = false;
    int device = 0;
    std::vector<ire_strips*> sessions;     // open strip sessions (g_strips_mu)
    ~ire_engine() {
        {
            std::lock_guard<std::mutex> lk(qmu);
            stop = true;
        }
        qcv.notify_all();
        if (worker.joinable()) worker.join();
        std::lock_guard<std::mutex> lk(g_strips_mu);
        for (ire_strips* s : sessions) {       // invalidate: the StripSession dies with its engine, the caller's shell survives
            if (eng) { std::lock_guard<std::mutex> lk2(eng->mutex()); s->s.reset(); } else s->s.reset();
            s->owner = nullptr;
        }
        sessions.clear();
    }
};

namespace ire {

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

// Two-slot pipeline (host feeding is what limits scaling): pinned staging buffers, H2D on a copy-in stream,
// classify + restore on the engine's main stream, D2H on a copy-out stream; while the GPU works on batch k the thread
// gathers batch k+1 into the other slot and hands batch k-1's pixels back to its jobs.
struct BatchSlot {
    uint8_t *pin_in = nullptr, *pin_out = nullptr, *d_in = nullptr, *d_out = nullptr, *pin_jp = nullptr, *d_jp = nullptr;
    double* pin_sc = nullptr;
    double* pin_sc_in = nullptr;          // scores the jobs brought along (ire_submit(..., scores, ...))
    uint8_t has_sc[64] = {};
    size_t cap = 0;                       // bytes of each image buffer
    hipEvent_t ev_in = nullptr, ev_c0 = nullptr, ev_c1 = nullptr, ev_out = nullptr;
    std::vector<std::shared_ptr<Job>> jobs;
    int h = 0, w = 0;
    int status = IRE_OK;
    std::string err;
    bool busy = false;
};

static void slot_reserve(BatchSlot& S, size_t bytes, int max_batch) {
    if (!S.ev_in) {
        IRE_HIP(hipEventCreateWithFlags(&S.ev_in, hipEventDisableTiming));
        IRE_HIP(hipEventCreate(&S.ev_c0));
        IRE_HIP(hipEventCreate(&S.ev_c1));
        IRE_HIP(hipEventCreateWithFlags(&S.ev_out, hipEventDisableTiming));
        IRE_HIP(hipHostMalloc((void**)&S.pin_jp, (size_t)max_batch));
        IRE_HIP(hipHostMalloc((void**)&S.pin_sc, sizeof(double) * 7 * (size_t)max_batch));
        IRE_HIP(hipHostMalloc((void**)&S.pin_sc_in, sizeof(double) * 7 * (size_t)max_batch));
        IRE_HIP(hipMalloc((void**)&S.d_jp, (size_t)max_batch));
    }
    if (bytes <= S.cap) return;
    if (S.pin_in) { (void)hipHostFree(S.pin_in); (void)hipHostFree(S.pin_out); (void)hipFree(S.d_in); (void)hipFree(S.d_out); }
    S.cap = 0;
    IRE_HIP(hipHostMalloc((void**)&S.pin_in, bytes));
    IRE_HIP(hipHostMalloc((void**)&S.pin_out, bytes));
    IRE_HIP(hipMalloc((void**)&S.d_in, bytes));
    IRE_HIP(hipMalloc((void**)&S.d_out, bytes));
    S.cap = bytes;
}

static void slot_free(BatchSlot& S) {
    if (S.pin_in) { (void)hipHostFree(S.pin_in); (void)hipHostFree(S.pin_out); (void)hipFree(S.d_in); (void)hipFree(S.d_out); }
    if (S.ev_in) {
        (void)hipEventDestroy(S.ev_in); (void)hipEventDestroy(S.ev_c0); (void)hipEventDestroy(S.ev_c1); (void)hipEventDestroy(S.ev_out);
        (void)hipHostFree(S.pin_jp); (void)hipHostFree(S.pin_sc); (void)hipHostFree(S.pin_sc_in); (void)hipFree(S.d_jp);
    }
    S = BatchSlot{};
}

// wait for the slot's batch and hand results (or the error) to its jobs
static void slot_finish(ire_engine* E, BatchSlot& S) {
    if (!S.busy) return;
    ire_timings t{};
    if (S.status == IRE_OK) {
        const hipError_t rc = hipEventSynchronize(S.ev_out);
        if (rc != hipSuccess) { S.status = IRE_ERR_INTERNAL; S.err = std::string("internal: ") + hipGetErrorString(rc); }
        else {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, S.ev_c0, S.ev_c1) == hipSuccess) { t.restore_ms = ms; t.total_ms = ms; }
        }
    }
    const size_t ib = (size_t)S.h * S.w * 3;
    const int n = (int)S.jobs.size();
    for (int i = 0; i < n; ++i) {          // the pixel copies happen outside the queue lock
        Job& j = *S.jobs[i];
        if (S.status == IRE_OK) j.out.assign(S.pin_out + ib * i, S.pin_out + ib * (i + 1));
    }
    {
        std::lock_guard<std::mutex> lk(E->qmu);
        for (int i = 0; i < n; ++i) {
            Job& j = *S.jobs[i];
            if (S.status == IRE_OK) { std::memcpy(j.scores, S.pin_sc + 7 * i, sizeof(double) * 7); j.t = t; }
            j.err = S.err;
            j.in.clear(); j.in.shrink_to_fit();
            j.status = S.status;
        }
    }
    E->dcv.notify_all();
    S.jobs.clear();
    S.busy = false;
}

static void batcher_loop(ire_engine* E) {
    (void)hipSetDevice(E->device);
    BatchSlot slots[2];
    hipStream_t cs = nullptr, os = nullptr;
    (void)hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&os, hipStreamNonBlocking);
    int p = 0;
    for (;;) {
        BatchSlot& S = slots[p];
        BatchSlot& prev = slots[p ^ 1];
        {
            std::unique_lock<std::mutex> lk(E->qmu);
            // a batch is in flight and nothing new is queued: its jobs are waiting -- finish it before blocking on the queue
            if (prev.busy && E->queue.empty() && !E->stop) { lk.unlock(); slot_finish(E, prev); lk.lock(); }
            E->qcv.wait(lk, [&] { return E->stop || !E->queue.empty(); });
            if (E->stop && E->queue.empty()) break;
            // coalescing: more submissions of the same shape usually follow at once.  Idle GPU: a 200 us window.  GPU busy with the
            // previous batch: its jobs come back (and are re-submitted by a closed-loop caller: 3 per restoreBatch, 5 per worker) only
            // when it ends, so a batch launched now would be whatever trickled in -- keep gathering until this batch is as large as
            // the one in flight, or that one has finished computing (then nothing is gained by waiting)
            if ((int)E->queue.size() < E->eng->max_batch()) {
                if (!prev.busy) {
                    E->qcv.wait_for(lk, std::chrono::microseconds(200),
                                    [&] { return E->stop || (int)E->queue.size() >= E->eng->max_batch(); });
                } else {
                    const int want = std::min(E->eng->max_batch(), (int)prev.jobs.size());
                    while (!E->stop && (int)E->queue.size() < want && prev.status == IRE_OK && hipEventQuery(prev.ev_c1) == hipErrorNotReady)
                        E->qcv.wait_for(lk, std::chrono::microseconds(50));
                }
            }
            S.h = E->queue.front()->h; S.w = E->queue.front()->w;
            for (auto it = E->queue.begin(); it != E->queue.end() && (int)S.jobs.size() < E->eng->max_batch();) {
                if ((*it)->h == S.h && (*it)->w == S.w) { S.jobs.push_back(*it); it = E->queue.erase(it); }
                else ++it;
            }
        }
        const int n = (int)S.jobs.size();
        const size_t ib = (size_t)S.h * S.w * 3;
        S.status = IRE_OK; S.err.clear(); S.busy = true;
        try {
            slot_reserve(S, ib * (size_t)E->eng->max_batch(), E->eng->max_batch());
            for (int i = 0; i < n; ++i) {
                std::memcpy(S.pin_in + ib * i, S.jobs[i]->in.data(), ib); S.pin_jp[i] = (uint8_t)S.jobs[i]->is_jpeg;
                S.has_sc[i] = S.jobs[i]->has_scores ? 1 : 0;
                if (S.has_sc[i]) std::memcpy(S.pin_sc_in + 7 * i, S.jobs[i]->scores, sizeof(double) * 7);
            }
            IRE_HIP(hipMemcpyAsync(S.d_in, S.pin_in, ib * n, hipMemcpyHostToDevice, cs));
            IRE_HIP(hipMemcpyAsync(S.d_jp, S.pin_jp, (size_t)n, hipMemcpyHostToDevice, cs));
            IRE_HIP(hipEventRecord(S.ev_in, cs));
            {
                hipStream_t ms = E->eng->main_stream();
                on_stream(*E->eng, ms, [&] {
                    IRE_HIP(hipStreamWaitEvent(ms, S.ev_in, 0));
                    IRE_HIP(hipEventRecord(S.ev_c0, ms));
                    E->eng->restore_device_mixed(S.d_in, n, S.h, S.w, S.pin_sc_in, S.has_sc, S.d_jp, S.d_out, ms);   // classifies the jobs that brought no scores
                    IRE_HIP(hipMemcpyAsync(S.pin_sc, E->eng->scores_device(), sizeof(double) * 7 * n, hipMemcpyDeviceToHost, ms));
                    IRE_HIP(hipEventRecord(S.ev_c1, ms));
                });
            }
            IRE_HIP(hipStreamWaitEvent(os, S.ev_c1, 0));
            IRE_HIP(hipMemcpyAsync(S.pin_out, S.d_out, ib * n, hipMemcpyDeviceToHost, os));
            IRE_HIP(hipEventRecord(S.ev_out, os));
        } catch (const Error& e) { S.status = e.code; S.err = e.msg; }
        catch (const std::exception& e) { S.status = IRE_ERR_INTERNAL; S.err = std::string("internal: ") + e.what(); }
        slot_finish(E, prev);       // overlaps with the batch just launched
        if (S.status != IRE_OK) slot_finish(E, S);
        p ^= 1;
    }
    slot_finish(E, slots[0]);
    slot_finish(E, slots[1]);
    (void)hipStreamSynchronize(cs);
    (void)hipStreamSynchronize(os);
    slot_free(slots[0]);
    slot_free(slots[1]);
    (void)hipStreamDestroy(cs);
    (void)hipStreamDestroy(os);
}

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
        if (cfg->flags != 0) fail(IRE_ERR_INVALID_INPUT, "invalid ire_config.flags (reserved, must be 0)");
        *out = nullptr;
        std::unique_ptr<ire_engine> E(new ire_engine());
        E->eng.reset(new Engine(*cfg));
        E->device = cfg->device_index;
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
        std::lock_guard<std::mutex> lk(e->qmu);
        out->queue_depth = (int32_t)e->queue.size();
    });
}

int ire_submit(ire_engine* e, const uint8_t* rgb, int h, int w, int is_jpeg, const double* scores, ire_job** job_out) {
    return guarded([&] {
        eng(e);
        if (!rgb || !job_out) fail(IRE_ERR_INVALID_INPUT, "invalid arguments to ire_submit");
        if (h <= 0 || w <= 0 || h % 8 || w % 8 || h < 16 || w < 16 || h > 8192 || w > 8192)
            fail(IRE_ERR_INVALID_INPUT, "invalid image size for restore: height and width must be multiples of 8, >= 16");
        auto j = std::make_shared<Job>();
        j->h = h; j->w = w; j->is_jpeg = is_jpeg ? 1 : 0;
        if (scores) { std::memcpy(j->scores, scores, sizeof(double) * 7); j->has_scores = true; }
        j->in.assign(rgb, rgb + (size_t)h * w * 3);
        {
            std::lock_guard<std::mutex> lk(e->qmu);
            if (!e->worker.joinable()) e->worker = std::thread(batcher_loop, e);
            e->queue.push_back(j);
        }
        e->qcv.notify_all();
        *job_out = new ire_job{j};
    });
}

int ire_poll(ire_engine* e, ire_job* job, int timeout_ms, uint8_t* out_rgb, double* scores_out, ire_timings* t) {
    return guarded([&] {
        eng(e);
        if (!job || !job->j) fail(IRE_ERR_INVALID_INPUT, "invalid job handle");
        std::shared_ptr<Job> j = job->j;
        {
            std::unique_lock<std::mutex> lk(e->qmu);
            auto done = [&] { return j->status >= 0; };
            if (timeout_ms < 0) e->dcv.wait(lk, done);
            else if (!e->dcv.wait_for(lk, std::chrono::milliseconds(timeout_ms), done))
                fail(IRE_ERR_TIMEOUT, "timeout: job still pending");
        }
        const int st = j->status;
        const std::string err = j->err;
        if (st == IRE_OK) {
            if (out_rgb) std::memcpy(out_rgb, j->out.data(), j->out.size());
            if (scores_out) std::memcpy(scores_out, j->scores, sizeof(double) * 7);
            if (t) *t = j->t;
        }
        delete job;
        if (st != IRE_OK) fail(st, err);
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
