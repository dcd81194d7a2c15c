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
//
// Data path, ONE host copy per direction: ire_submit copies the caller's pixels straight into the pinned staging slot of
// the batch being gathered (in the caller's thread: concurrent callers copy concurrently), ire_poll copies from the batch's
// pinned output to the caller (again in the caller's thread).  Round 2 had three copies per direction on one thread.
//
// A slot is FREE -> OPEN (gathering: submits reserve an index and stage into it; the launcher issues each staged image's
// H2D copy at once, under the previous batch's compute) -> CLOSED (launching) -> INFLIGHT (kernels + D2H enqueued) -> DONE
// (results in pin_out, waiting for its jobs' ire_poll) -> FREE.  Two threads: the launcher decides when a gathering batch
// goes (below), the completer waits for the oldest in-flight batch's D2H and completes its jobs.
//
// When a batch goes: at once when it is full (the stream runs it behind the previous batch: no bubble); otherwise it keeps
// gathering while the GPU still computes the previous batch (launching then would only split what a closed-loop caller --
// 3 jobs per restoreBatch, 5 per worker -- is about to resubmit), and when the GPU is idle after a short bounded linger:
// kLingerQuietUs after the last arrival, at most kLingerMaxUs after the first.
struct BatchSlot;
struct Job {
    int h, w, is_jpeg;
    bool has_scores = false;
    bool staged = false;              // input bytes are in the slot's pinned buffer (or in `in` on the overflow path)
    BatchSlot* slot = nullptr;        // where the input was staged and the output will be; null: overflow (no free slot at submit) / evicted
    int idx = -1;
    std::vector<uint8_t> in, out;     // overflow input / evicted output only
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

namespace ire {

constexpr int kSlots = 8;             // gathering | computing | up to six waiting for their polls (a caller that submits a burst before its first poll)
constexpr int kSlotsEager = 3;        // staging allocated at the first job of a shape; the other slots get theirs when first needed
constexpr int kLingerQuietUs = 250, kLingerMaxUs = 1500;

struct BatchSlot {
    enum State { FREE, OPEN, CLOSED, INFLIGHT, DONE } state = FREE;
    uint8_t *pin_in = nullptr, *pin_out = nullptr, *d_in = nullptr, *d_out = nullptr, *pin_jp = nullptr, *d_jp = nullptr;
    double* pin_sc = nullptr;
    double* pin_sc_in = nullptr;          // scores the jobs brought along (ire_submit(..., scores, ...))
    uint8_t has_sc[64] = {};
    size_t cap = 0;                       // bytes of each image buffer
    hipEvent_t ev_in = nullptr, ev_c0 = nullptr, ev_c1 = nullptr, ev_out = nullptr;
    std::vector<std::shared_ptr<Job>> jobs;   // index order = position in the batch
    int h = 0, w = 0;
    int h2d_issued = 0;                   // images whose H2D copy is already on the copy-in stream
    int unread = 0, reading = 0;          // DONE: jobs that have not fetched their output yet / polls copying right now
    std::chrono::steady_clock::time_point first_arrival, last_arrival;
    int status = IRE_OK;
    std::string err;
};

}  // namespace ire

struct ire_engine {
    std::unique_ptr<ire::Engine> eng;
    // batcher
    std::mutex qmu;
    std::condition_variable qcv, dcv, ccv;     // launcher wake-ups | job completion | completer wake-ups
    ire::BatchSlot slots[ire::kSlots];
    std::deque<int> open_order;                // OPEN slots, oldest first
    std::deque<int> inflight;                  // INFLIGHT slots, launch order
    std::deque<std::shared_ptr<ire::Job>> overflow;   // submitted while no slot was free: staged by the launcher later
    int last_launched = -1;
    std::thread worker, completer;
    hipStream_t cs = nullptr, os = nullptr;
    bool stop = false, launcher_done = false;
    int device = 0;
    std::vector<ire_strips*> sessions;     // open strip sessions (g_strips_mu)
    ~ire_engine();
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
    S.cap = 0; S.pin_in = S.pin_out = S.d_in = S.d_out = nullptr;
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

// (qmu held) an OPEN slot of this shape with room, else a FREE one opened for it, else -1.  May allocate staging (first use
// of a shape: once).  A DONE slot nobody is reading is evicted when nothing else is left: its unfetched outputs move to
// their jobs' own vectors (the extra copy only a caller that lets four batches pile up unpolled ever pays).
static int slot_for(ire_engine* E, int h, int w) {
    const int mb = E->eng->max_batch();
    for (auto it = E->open_order.rbegin(); it != E->open_order.rend(); ++it) {
        BatchSlot& S = E->slots[*it];
        if (S.h == h && S.w == w && (int)S.jobs.size() < mb) return *it;
    }
    int pick = -1;
    const size_t need = (size_t)h * w * 3 * (size_t)mb;
    for (int i = 0; i < kSlots && pick < 0; ++i) if (E->slots[i].state == BatchSlot::FREE && E->slots[i].cap >= need) pick = i;     // one whose staging exists
    for (int i = 0; i < kSlots && pick < 0; ++i) if (E->slots[i].state == BatchSlot::FREE) pick = i;
    if (pick < 0) {
        for (int i = 0; i < kSlots && pick < 0; ++i) {
            BatchSlot& S = E->slots[i];
            if (S.state != BatchSlot::DONE || S.reading) continue;
            const size_t ib = (size_t)S.h * S.w * 3;
            for (auto& j : S.jobs)
                if (j->slot == &S) { j->out.assign(S.pin_out + ib * j->idx, S.pin_out + ib * (j->idx + 1)); j->slot = nullptr; }
            S.jobs.clear(); S.unread = 0; S.state = BatchSlot::FREE;
            pick = i;
        }
    }
    if (pick < 0) return -1;
    BatchSlot& S = E->slots[pick];
    {
        // staging is allocated on first use of a shape, possibly from a caller's thread: leave that thread's current device as it was
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(E->device);
        try { slot_reserve(S, (size_t)h * w * 3 * (size_t)mb, mb); } catch (...) { if (prev >= 0) (void)hipSetDevice(prev); throw; }
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    S.state = BatchSlot::OPEN; S.h = h; S.w = w; S.jobs.clear(); S.h2d_issued = 0; S.unread = S.reading = 0;
    S.status = IRE_OK; S.err.clear();
    S.first_arrival = S.last_arrival = std::chrono::steady_clock::now();
    E->open_order.push_back(pick);
    return pick;
}

// (qmu held) reserve the next index of slot si for job j
static void slot_add(ire_engine* E, int si, const std::shared_ptr<Job>& j) {
    BatchSlot& S = E->slots[si];
    j->slot = &S; j->idx = (int)S.jobs.size();
    S.pin_jp[j->idx] = (uint8_t)j->is_jpeg;
    S.has_sc[j->idx] = j->has_scores ? 1 : 0;
    if (j->has_scores) std::memcpy(S.pin_sc_in + 7 * j->idx, j->scores, sizeof(double) * 7);
    S.jobs.push_back(j);
    S.last_arrival = std::chrono::steady_clock::now();
    if (j->idx == 0) S.first_arrival = S.last_arrival;
}

// (launcher, qmu held) H2D of every image staged so far, in index order, on the copy-in stream: rides under the previous
// batch's compute.  Returns false on a HIP error (recorded in the slot).
static void slot_push_h2d(ire_engine* E, BatchSlot& S) {
    const size_t ib = (size_t)S.h * S.w * 3;
    int upto = S.h2d_issued;
    while (upto < (int)S.jobs.size() && S.jobs[upto]->staged) ++upto;
    if (upto == S.h2d_issued || S.status != IRE_OK) return;
    const hipError_t rc = hipMemcpyAsync(S.d_in + ib * S.h2d_issued, S.pin_in + ib * S.h2d_issued, ib * (size_t)(upto - S.h2d_issued), hipMemcpyHostToDevice, E->cs);
    if (rc != hipSuccess) { S.status = IRE_ERR_INTERNAL; S.err = std::string("internal: ") + hipGetErrorString(rc); }
    S.h2d_issued = upto;
}

static void complete_jobs(ire_engine* E, BatchSlot& S, const ire_timings& t) {     // qmu held
    const int n = (int)S.jobs.size();
    for (int i = 0; i < n; ++i) {
        Job& j = *S.jobs[i];
        if (S.status == IRE_OK) { std::memcpy(j.scores, S.pin_sc + 7 * i, sizeof(double) * 7); j.t = t; }
        else j.slot = nullptr;
        j.err = S.err;
        j.status = S.status;
    }
    if (S.status == IRE_OK) { S.state = BatchSlot::DONE; S.unread = n; }
    else { S.jobs.clear(); S.state = BatchSlot::FREE; }
}

static void launcher_loop(ire_engine* E) {
    (void)hipSetDevice(E->device);
    using clk = std::chrono::steady_clock;
    std::unique_lock<std::mutex> lk(E->qmu);
    for (;;) {
        // jobs that found no free slot at submit time: stage them now (this thread copies), oldest first
        while (!E->overflow.empty()) {
            std::shared_ptr<Job> j = E->overflow.front();
            int si = -1;
            try { si = slot_for(E, j->h, j->w); }
            catch (const Error& e) { j->status = e.code; j->err = e.msg; E->overflow.pop_front(); E->dcv.notify_all(); continue; }
            if (si < 0) break;
            E->overflow.pop_front();
            slot_add(E, si, j);
            BatchSlot& S = E->slots[si];
            std::memcpy(S.pin_in + (size_t)j->h * j->w * 3 * j->idx, j->in.data(), j->in.size());
            j->in.clear(); j->in.shrink_to_fit();
            j->staged = true;
        }
        if (E->open_order.empty()) {
            if (E->stop && E->overflow.empty()) break;
            // (overflow jobs wait for a slot: a poll or the completer frees one and notifies)
            if (E->overflow.empty()) E->qcv.wait(lk); else E->qcv.wait_for(lk, std::chrono::microseconds(200));
            continue;
        }
        const int si = E->open_order.front();
        BatchSlot& S = E->slots[si];
        slot_push_h2d(E, S);
        const bool full = (int)S.jobs.size() >= E->eng->max_batch();
        if (!full && !E->stop && S.status == IRE_OK && E->open_order.size() == 1) {
            bool gpu_busy = false;
            if (E->last_launched >= 0) {
                BatchSlot& P = E->slots[E->last_launched];
                gpu_busy = P.state == BatchSlot::INFLIGHT && hipEventQuery(P.ev_c1) == hipErrorNotReady;
            }
            if (gpu_busy) { E->qcv.wait_for(lk, std::chrono::microseconds(100)); continue; }
            const auto now = clk::now();
            const auto quiet = std::chrono::duration_cast<std::chrono::microseconds>(now - S.last_arrival).count();
            const auto age = std::chrono::duration_cast<std::chrono::microseconds>(now - S.first_arrival).count();
            if (quiet < kLingerQuietUs && age < kLingerMaxUs) { E->qcv.wait_for(lk, std::chrono::microseconds(50)); continue; }
        }
        // launch: no more reservations, wait for the copies still running in submitting threads
        S.state = BatchSlot::CLOSED;
        E->open_order.pop_front();
        E->qcv.wait(lk, [&] { for (auto& j : S.jobs) if (!j->staged) return false; return true; });
        slot_push_h2d(E, S);
        const int n = (int)S.jobs.size();
        const size_t ib = (size_t)S.h * S.w * 3;
        lk.unlock();
        try {
            if (S.status != IRE_OK) throw Error{S.status, S.err};
            IRE_HIP(hipMemcpyAsync(S.d_jp, S.pin_jp, (size_t)n, hipMemcpyHostToDevice, E->cs));
            IRE_HIP(hipEventRecord(S.ev_in, E->cs));
            hipStream_t ms = E->eng->main_stream();
            on_stream(*E->eng, ms, [&] {
                IRE_HIP(hipStreamWaitEvent(ms, S.ev_in, 0));
                IRE_HIP(hipEventRecord(S.ev_c0, ms));
                E->eng->restore_device_mixed(S.d_in, n, S.h, S.w, S.pin_sc_in, S.has_sc, S.d_jp, S.d_out, ms);   // classifies the jobs that brought no scores
                IRE_HIP(hipMemcpyAsync(S.pin_sc, E->eng->scores_device(), sizeof(double) * 7 * n, hipMemcpyDeviceToHost, ms));
                IRE_HIP(hipEventRecord(S.ev_c1, ms));
            });
            IRE_HIP(hipStreamWaitEvent(E->os, S.ev_c1, 0));
            IRE_HIP(hipMemcpyAsync(S.pin_out, S.d_out, ib * n, hipMemcpyDeviceToHost, E->os));
            IRE_HIP(hipEventRecord(S.ev_out, E->os));
        } catch (const Error& e) { S.status = e.code; S.err = e.msg; }
        catch (const std::exception& e) { S.status = IRE_ERR_INTERNAL; S.err = std::string("internal: ") + e.what(); }
        lk.lock();
        if (S.status == IRE_OK) {
            S.state = BatchSlot::INFLIGHT;
            E->inflight.push_back(si);
            E->last_launched = si;
            E->ccv.notify_all();
        } else {
            // whatever was enqueued before the failure may still touch the slot's buffers: drain before handing the error out
            lk.unlock(); (void)hipStreamSynchronize(E->cs); (void)hipStreamSynchronize(E->eng->main_stream()); (void)hipStreamSynchronize(E->os); lk.lock();
            complete_jobs(E, S, ire_timings{});
            E->dcv.notify_all();
        }
    }
    E->launcher_done = true;
    E->ccv.notify_all();
}

static void completer_loop(ire_engine* E) {
    (void)hipSetDevice(E->device);
    std::unique_lock<std::mutex> lk(E->qmu);
    for (;;) {
        E->ccv.wait(lk, [&] { return !E->inflight.empty() || E->launcher_done; });
        if (E->inflight.empty()) break;
        BatchSlot& S = E->slots[E->inflight.front()];
        lk.unlock();
        ire_timings t{};
        const hipError_t rc = hipEventSynchronize(S.ev_out);
        int st = IRE_OK; std::string err;
        if (rc != hipSuccess) { st = IRE_ERR_INTERNAL; err = std::string("internal: ") + hipGetErrorString(rc); }
        else {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, S.ev_c0, S.ev_c1) == hipSuccess) { t.restore_ms = ms; t.total_ms = ms; }
        }
        lk.lock();
        if (st != IRE_OK) { S.status = st; S.err = err; }
        E->inflight.pop_front();
        complete_jobs(E, S, t);
        E->dcv.notify_all();
        E->qcv.notify_all();
    }
}

}  // namespace ire

ire_engine::~ire_engine() {
    {
        std::lock_guard<std::mutex> lk(qmu);
        stop = true;
    }
    qcv.notify_all();
    if (worker.joinable()) worker.join();        // launches what is still gathered, then exits
    ccv.notify_all();
    if (completer.joinable()) completer.join();  // completes every batch in flight
    if (eng) {
        (void)hipSetDevice(device);
        if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
        if (os) { (void)hipStreamSynchronize(os); (void)hipStreamDestroy(os); }
        for (auto& S : slots) ire::slot_free(S);
    }
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
        int depth = (int)e->overflow.size();
        for (int si : e->open_order) depth += (int)e->slots[si].jobs.size();
        out->queue_depth = depth;
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
        const size_t ib = (size_t)h * w * 3;
        uint8_t* dst = nullptr;
        {
            std::lock_guard<std::mutex> lk(e->qmu);
            if (!e->worker.joinable()) {
                int prev = -1;
                (void)hipGetDevice(&prev);
                (void)hipSetDevice(e->device);
                if (!e->cs) IRE_HIP(hipStreamCreateWithFlags(&e->cs, hipStreamNonBlocking));
                if (!e->os) IRE_HIP(hipStreamCreateWithFlags(&e->os, hipStreamNonBlocking));
                // staging for the three slots of a steady stream now, at this first shape (pinning 3 x 2 x max_batch images takes tens
                // of ms): the first job pays it once, instead of later jobs paying it one slot at a time in the middle of a stream
                try { for (int i = 0; i < kSlotsEager; ++i) slot_reserve(e->slots[i], ib * (size_t)e->eng->max_batch(), e->eng->max_batch()); }
                catch (...) { if (prev >= 0) (void)hipSetDevice(prev); throw; }
                if (prev >= 0) (void)hipSetDevice(prev);
                e->worker = std::thread(launcher_loop, e);
                e->completer = std::thread(completer_loop, e);
            }
            const int si = e->overflow.empty() ? slot_for(e, h, w) : -1;     // (jobs already overflowing keep their order)
            if (si >= 0) { slot_add(e, si, j); dst = e->slots[si].pin_in + ib * j->idx; }
        }
        if (dst) {
            std::memcpy(dst, rgb, ib);                 // the ONE host copy of the input: caller's buffer -> pinned slot, in the caller's thread
            std::lock_guard<std::mutex> lk(e->qmu);
            j->staged = true;
        } else {
            j->in.assign(rgb, rgb + ib);               // every slot is busy: keep the pixels until the launcher finds one
            std::lock_guard<std::mutex> lk(e->qmu);
            e->overflow.push_back(j);
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
        BatchSlot* S = nullptr;
        {
            std::unique_lock<std::mutex> lk(e->qmu);
            auto done = [&] { return j->status >= 0; };
            if (timeout_ms < 0) e->dcv.wait(lk, done);
            else if (!e->dcv.wait_for(lk, std::chrono::milliseconds(timeout_ms), done))
                fail(IRE_ERR_TIMEOUT, "timeout: job still pending");
            S = j->slot;
            if (S) S->reading += 1;        // the slot cannot be recycled (or evicted) while this thread copies from it
        }
        const int st = j->status;
        const std::string err = j->err;
        if (st == IRE_OK) {
            const size_t ib = (size_t)j->h * j->w * 3;
            // the ONE host copy of the output: pinned slot -> caller's buffer, in the caller's thread
            if (out_rgb) std::memcpy(out_rgb, S ? S->pin_out + ib * j->idx : j->out.data(), ib);
            if (scores_out) std::memcpy(scores_out, j->scores, sizeof(double) * 7);
            if (t) *t = j->t;
        }
        if (S) {
            std::lock_guard<std::mutex> lk(e->qmu);
            S->reading -= 1; S->unread -= 1;
            j->slot = nullptr;
            if (S->unread == 0 && S->state == BatchSlot::DONE) { S->jobs.clear(); S->state = BatchSlot::FREE; e->qcv.notify_all(); }
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
