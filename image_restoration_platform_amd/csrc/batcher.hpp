// batcher.hpp -- the async batcher behind ire_submit / ire_poll / ire_job_release: restoreBatch's in-flight promises
// (server-node/src/services/restorator.js:181-236) and the worker's five concurrent jobs (design.md:851) coalesced into engine
// batches of up to max_batch equal-shape images.
//
// The state machine is a template over its BACKEND -- everything that touches the device (pinned / device staging, copies, the
// restore call, completion events) -- so that the same code is built twice: api.cpp instantiates it over HIP (HipBatchBackend),
// tests/native/batcher_stress.cpp over a host-only stub and runs it under ThreadSanitizer and AddressSanitizer on the CPU box
// (sanitizers on the CPU build).  The stub is test infrastructure: libire.so never contains it.
//
// Data path, ONE host copy per direction: submit copies the caller's pixels straight into the pinned staging slot of the batch
// being gathered (in the caller's thread: concurrent callers copy concurrently), poll copies from the batch's pinned output to
// the caller (again in the caller's thread).
//
// A slot is FREE -> OPEN (gathering: submits reserve an index and stage into it; the launcher issues each staged image's H2D
// copy at once, under the previous batch's compute) -> CLOSED (launching) -> INFLIGHT (kernels + D2H enqueued) -> DONE (results
// in pin_out, waiting for its jobs' polls) -> FREE.  Two service threads: the launcher decides when a gathering batch goes, the
// completer waits for the oldest in-flight batch (first its compute, which wakes the launcher, then its D2H) and completes its
// jobs.  Every wait is on a condition variable with a deadline that means something (the linger bounds) -- no polling loops.
//
// When a batch goes: at once when it is full (the stream runs it behind the previous batch: no bubble); otherwise it keeps
// gathering while the GPU still computes the previous batch (launching then would only split what a closed-loop caller -- 3 jobs
// per restoreBatch, 5 per worker -- is about to resubmit), and when the GPU is idle after a short bounded linger: kLingerQuietUs
// after the last arrival, at most kLingerMaxUs after the first.
//
// A job handle is used by ONE thread at a time (poll it, or release it).  A timed-out poll leaves the job pending; the caller
// polls again or RELEASES it (the reference retries 3x on exactly this path: utils/retry.js:12-47): an abandoned job's slot
// position is still computed with its batch, but nobody is counted as waiting for it.
#pragma once
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "errors.hpp"

namespace ire {

constexpr int kSlots = 8;             // gathering | computing | up to six waiting for their polls (a caller that submits a burst before its first poll)
constexpr int kSlotsEager = 3;        // staging allocated at the first job of a shape; the other slots get theirs when first needed
constexpr int kLingerQuietUs = 250, kLingerMaxUs = 1500;
constexpr int kMaxBatch = 64;

struct BatchSlot;
struct Job {
    int h = 0, w = 0, is_jpeg = 0;
    bool has_scores = false;
    bool staged = false;              // input bytes are in the slot's pinned buffer (or in `in` on the overflow path)
    bool abandoned = false;           // released while pending: its result is dropped when the batch completes
    BatchSlot* slot = nullptr;        // where the input was staged and the output will be; null: overflow (no free slot at submit) / evicted / fetched
    int idx = -1;
    std::vector<uint8_t> in, out;     // overflow input / evicted output only
    double scores[7] = {};
    ire_timings t{};
    int status = -1;                  // -1 pending, else ire_status
    std::string err;
};

// Host-visible staging of a slot.  The backend allocates and frees it (reserve / release) and keeps its own device side in `impl`.
struct SlotBufs {
    uint8_t *pin_in = nullptr, *pin_out = nullptr, *pin_jp = nullptr;
    double *pin_sc = nullptr, *pin_sc_in = nullptr;
    size_t cap = 0;                   // bytes of pin_in / pin_out
    bool fixed = false;               // pin_jp / pin_sc / pin_sc_in (and the backend's events) exist
    void* impl = nullptr;
};

struct BatchSlot {
    enum State { FREE, OPEN, CLOSED, INFLIGHT, DONE } state = FREE;
    SlotBufs b;
    uint8_t has_sc[kMaxBatch] = {};
    std::vector<std::shared_ptr<Job>> jobs;   // index order = position in the batch
    int h = 0, w = 0;
    int h2d_issued = 0;                   // images whose H2D copy is already on the copy-in stream
    int unread = 0, reading = 0;          // DONE: jobs that have not fetched their output yet / polls copying right now
    std::chrono::steady_clock::time_point first_arrival, last_arrival;
    int status = IRE_OK;
    std::string err;
};

// Backend concept (HipBatchBackend in api.cpp; StubBackend in tests/native/batcher_stress.cpp):
//   int  max_batch() const;
//   size_t out_bytes(int h, int w) const;                      bytes of one job's result (h * w * 3 pixels, or the text of an encoded image)
//   void start();                                              first submit, batcher lock held: streams
//   void thread_enter(const char* role);                       at the top of a service thread: device, CPU affinity
//   void reserve(SlotBufs&, size_t bytes, int max_batch);      grow to `bytes` per direction; STRONG guarantee: on a throw the SlotBufs is as before
//   void release(SlotBufs&) noexcept;
//   void h2d(SlotBufs&, size_t off, size_t bytes);             async copy pin_in -> device, copy-in stream; throws Error
//   void launch(SlotBufs&, int n, int h, int w, const uint8_t* has_sc);   flags + restore + scores + D2H, all enqueued; throws Error
//   bool computing(SlotBufs&) noexcept;                        the launched batch's compute has not finished (non-blocking)
//   void wait_compute(SlotBufs&) noexcept;                     blocks until it has
//   void wait_done(SlotBufs&, ire_timings&);                   blocks until pin_out / pin_sc are complete; throws Error
//   void drain() noexcept;                                     after a failed launch: nothing enqueued may still touch a slot
template <class Backend>
class Batcher {
  public:
    using clk = std::chrono::steady_clock;
    explicit Batcher(Backend& be) : be_(be) { const char* t = std::getenv("IRE_BATCH_TRACE"); trace_ = t && t[0] == '1'; }
    Batcher(const Batcher&) = delete;
    Batcher& operator=(const Batcher&) = delete;

    ~Batcher() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        qcv_.notify_all();
        if (launcher_.joinable()) launcher_.join();      // launches what is still gathered (and staged), then exits
        ccv_.notify_all();
        if (completer_.joinable()) completer_.join();    // completes every batch in flight
        be_.drain();
        for (auto& S : slots_) { be_.release(S.b); S = BatchSlot{}; }
    }

    // One timed wait for every caller.  The product waits on the steady clock (pthread_cond_clockwait).  The ThreadSanitizer
    // build of tests/native defines IRE_BATCHER_SYSCLOCK_WAITS: gcc 11's libtsan does not intercept pthread_cond_clockwait, so it
    // misses the unlock / relock inside the wait and reports every other thread's critical section as a race; the same deadline
    // on the system clock goes through pthread_cond_timedwait, which it does intercept.  Returns false at the deadline.
    template <class Lock>
    static bool wait_deadline(std::condition_variable& cv, Lock& lk, clk::time_point tp) {
#ifdef IRE_BATCHER_SYSCLOCK_WAITS
        const auto left = tp - clk::now();
        if (left <= clk::duration::zero()) return false;
        return cv.wait_until(lk, std::chrono::system_clock::now() + left) == std::cv_status::no_timeout;
#else
        return cv.wait_until(lk, tp) == std::cv_status::no_timeout;
#endif
    }

    // Queue one image.  Throws Error (invalid size is the caller's check); the returned job is pending.
    std::shared_ptr<Job> submit(const uint8_t* rgb, int h, int w, int is_jpeg, const double* scores) {
        auto j = std::make_shared<Job>();
        j->h = h; j->w = w; j->is_jpeg = is_jpeg ? 1 : 0;
        if (scores) { std::memcpy(j->scores, scores, sizeof(double) * 7); j->has_scores = true; }
        const size_t ib = (size_t)h * w * 3;
        uint8_t* dst = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (stop_) fail(IRE_ERR_UNAVAILABLE, "service unavailable: the engine is shutting down");
            if (!started_) {
                be_.start();
                // staging for the three slots of a steady stream now, at this first shape (pinning 3 x 2 x max_batch images takes tens
                // of ms): the first job pays it once, instead of later jobs paying it one slot at a time in the middle of a stream
                for (int i = 0; i < kSlotsEager; ++i) be_.reserve(slots_[i].b, std::max(ib, be_.out_bytes(h, w)) * (size_t)be_.max_batch(), be_.max_batch());
                launcher_ = std::thread([this] { launcher_loop(); });
                completer_ = std::thread([this] { completer_loop(); });
                started_ = true;
            }
            const int si = overflow_.empty() ? slot_for(h, w) : -1;     // (jobs already overflowing keep their order)
            if (si >= 0) { slot_add(si, j); dst = slots_[si].b.pin_in + ib * j->idx; }
        }
        if (dst) {
            std::memcpy(dst, rgb, ib);                 // the ONE host copy of the input: caller's buffer -> pinned slot, in the caller's thread
            std::lock_guard<std::mutex> lk(mu_);
            j->staged = true;
        } else {
            j->in.assign(rgb, rgb + ib);               // every slot is busy: keep the pixels until the launcher finds one
            std::lock_guard<std::mutex> lk(mu_);
            j->staged = true;
            overflow_.push_back(j);
            cnt_.overflowed += 1;
        }
        qcv_.notify_all();
        return j;
    }

    // Wait up to timeout_ms (< 0: forever).  Returns IRE_OK (outputs filled, job finished), IRE_ERR_TIMEOUT (job still pending:
    // poll again or release) or the job's failure status (*err_out = its message; job finished).
    int poll(const std::shared_ptr<Job>& j, int timeout_ms, uint8_t* out_rgb, double* scores_out, ire_timings* t, std::string* err_out) {
        BatchSlot* S = nullptr;
        {
            std::unique_lock<std::mutex> lk(mu_);
            auto done = [&] { return j->status >= 0; };
            if (timeout_ms < 0) dcv_.wait(lk, done);
            else {
                const auto until = clk::now() + std::chrono::milliseconds(timeout_ms);
                while (!done()) if (!wait_deadline(dcv_, lk, until) && !done()) return IRE_ERR_TIMEOUT;
            }
            S = j->slot;
            if (S) S->reading += 1;        // the slot cannot be recycled (or evicted) while this thread copies from it
        }
        const int st = j->status;
        if (st == IRE_OK) {
            const size_t ob = be_.out_bytes(j->h, j->w);
            // the ONE host copy of the output: pinned slot -> caller's buffer, in the caller's thread
            if (out_rgb) std::memcpy(out_rgb, S ? S->b.pin_out + ob * j->idx : j->out.data(), ob);
            if (scores_out) std::memcpy(scores_out, j->scores, sizeof(double) * 7);
            if (t) *t = j->t;
        } else if (err_out) *err_out = j->err;
        if (S) {
            std::lock_guard<std::mutex> lk(mu_);
            S->reading -= 1;
            drop_reader(*S, *j);
        }
        return st;
    }

    // Give up a job without fetching it (a rejected promise, a timed-out poll the caller will not repeat): frees what only this
    // handle kept alive -- its place among the slot's unread results, its overflow pixels.
    void release(const std::shared_ptr<Job>& j) {
        std::lock_guard<std::mutex> lk(mu_);
        if (j->status >= 0) {                     // complete: nobody will read it
            if (j->slot) drop_reader(*j->slot, *j);
            j->out.clear(); j->out.shrink_to_fit();
            return;
        }
        for (auto it = overflow_.begin(); it != overflow_.end(); ++it)
            if (it->get() == j.get()) { overflow_.erase(it); j->status = IRE_ERR_INTERNAL; j->in.clear(); j->in.shrink_to_fit(); return; }
        cnt_.abandoned += 1;
        j->abandoned = true;                      // gathered or in flight: the batch runs as staged, its completion skips this job
    }

    // how often the rare paths ran (tests assert that they did)
    struct Counters { long batches = 0, overflowed = 0, evicted = 0, abandoned = 0, failed_batches = 0; };
    Counters counters() { std::lock_guard<std::mutex> lk(mu_); return cnt_; }

    int queue_depth() {
        std::lock_guard<std::mutex> lk(mu_);
        int depth = (int)overflow_.size();
        for (int si : open_order_) depth += (int)slots_[si].jobs.size();
        return depth;
    }

  private:
    // (mu_ held) a job's result has been fetched or given up: the slot is free once the last one has
    void drop_reader(BatchSlot& S, Job& j) {
        j.slot = nullptr;
        S.unread -= 1;
        if (S.unread == 0 && S.state == BatchSlot::DONE) { S.jobs.clear(); S.state = BatchSlot::FREE; }
        qcv_.notify_all();                        // a FREE (or now evictable) slot: overflow jobs may be waiting for it
    }

    // (mu_ held) an OPEN slot of this shape with room, else a FREE one opened for it, else -1.  May allocate staging (first use
    // of a shape: once).  A DONE slot nobody is reading is evicted when nothing else is left: its unfetched outputs move to
    // their jobs' own vectors (the extra copy only a caller that lets a slot's worth of batches pile up unpolled ever pays).
    int slot_for(int h, int w) {
        const int mb = be_.max_batch();
        for (auto it = open_order_.rbegin(); it != open_order_.rend(); ++it) {
            BatchSlot& S = slots_[*it];
            if (S.h == h && S.w == w && (int)S.jobs.size() < mb) return *it;
        }
        int pick = -1;
        const size_t need = std::max((size_t)h * w * 3, be_.out_bytes(h, w)) * (size_t)mb;
        for (int i = 0; i < kSlots && pick < 0; ++i) if (slots_[i].state == BatchSlot::FREE && slots_[i].b.cap >= need) pick = i;     // one whose staging exists
        for (int i = 0; i < kSlots && pick < 0; ++i) if (slots_[i].state == BatchSlot::FREE) pick = i;
        for (int i = 0; i < kSlots && pick < 0; ++i) {
            BatchSlot& S = slots_[i];
            if (S.state != BatchSlot::DONE || S.reading) continue;
            const size_t ib = be_.out_bytes(S.h, S.w);
            // job by job, each move complete before the slot forgets the job: a bad_alloc half way leaves a consistent DONE slot
            for (auto& j : S.jobs)
                if (j->slot == &S) {
                    j->out.assign(S.b.pin_out + ib * j->idx, S.b.pin_out + ib * (j->idx + 1));
                    j->slot = nullptr;
                    S.unread -= 1;
                    cnt_.evicted += 1;
                }
            S.jobs.clear(); S.unread = 0; S.state = BatchSlot::FREE;
            pick = i;
        }
        if (pick < 0) return -1;
        BatchSlot& S = slots_[pick];
        be_.reserve(S.b, need, mb);               // strong guarantee: a throw leaves the slot FREE with what it had
        S.state = BatchSlot::OPEN; S.h = h; S.w = w; S.jobs.clear(); S.h2d_issued = 0; S.unread = S.reading = 0;
        S.status = IRE_OK; S.err.clear();
        S.first_arrival = S.last_arrival = clk::now();
        open_order_.push_back(pick);
        return pick;
    }

    // (mu_ held) reserve the next index of slot si for job j
    void slot_add(int si, const std::shared_ptr<Job>& j) {
        BatchSlot& S = slots_[si];
        j->slot = &S; j->idx = (int)S.jobs.size();
        S.b.pin_jp[j->idx] = (uint8_t)j->is_jpeg;
        S.has_sc[j->idx] = j->has_scores ? 1 : 0;
        if (j->has_scores) std::memcpy(S.b.pin_sc_in + 7 * j->idx, j->scores, sizeof(double) * 7);
        S.jobs.push_back(j);
        S.last_arrival = clk::now();
        if (j->idx == 0) S.first_arrival = S.last_arrival;
    }

    // (launcher, mu_ held) H2D of every image staged so far, in index order, on the copy-in stream: rides under the previous
    // batch's compute.  A failure is recorded in the slot (the launch then fails the batch).
    void slot_push_h2d(BatchSlot& S) {
        const size_t ib = (size_t)S.h * S.w * 3;
        int upto = S.h2d_issued;
        while (upto < (int)S.jobs.size() && S.jobs[upto]->staged) ++upto;
        if (upto == S.h2d_issued || S.status != IRE_OK) return;
        try { be_.h2d(S.b, ib * S.h2d_issued, ib * (size_t)(upto - S.h2d_issued)); }
        catch (const Error& e) { S.status = e.code; S.err = e.msg; }
        S.h2d_issued = upto;
    }

    void complete_jobs(BatchSlot& S, const ire_timings& t) {     // mu_ held
        const int n = (int)S.jobs.size();
        int readers = 0;
        for (int i = 0; i < n; ++i) {
            Job& j = *S.jobs[i];
            if (S.status == IRE_OK && !j.abandoned) { std::memcpy(j.scores, S.b.pin_sc + 7 * i, sizeof(double) * 7); j.t = t; ++readers; }
            else j.slot = nullptr;
            j.err = S.err;
            j.status = j.abandoned && S.status == IRE_OK ? IRE_ERR_INTERNAL : S.status;
        }
        if (S.status == IRE_OK && readers > 0) { S.state = BatchSlot::DONE; S.unread = readers; }
        else { S.jobs.clear(); S.state = BatchSlot::FREE; }
    }

    static void fail_job(Job& j, int code, const std::string& msg) { j.status = code; j.err = msg; j.in.clear(); j.in.shrink_to_fit(); }

    void launcher_loop() {
        be_.thread_enter("launcher");
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            // jobs that found no free slot at submit time: stage them now (this thread copies), oldest first
            while (!overflow_.empty()) {
                std::shared_ptr<Job> j = overflow_.front();
                int si = -1;
                try { si = slot_for(j->h, j->w); }
                catch (const Error& e) { fail_job(*j, e.code, e.msg); overflow_.pop_front(); dcv_.notify_all(); continue; }
                catch (const std::exception& e) {      // bad_alloc while evicting a DONE slot: this job fails, the service thread lives
                    fail_job(*j, IRE_ERR_UNAVAILABLE, std::string("service unavailable: ") + e.what()); overflow_.pop_front(); dcv_.notify_all(); continue;
                }
                if (si < 0) break;
                overflow_.pop_front();
                slot_add(si, j);
                BatchSlot& S = slots_[si];
                std::memcpy(S.b.pin_in + (size_t)j->h * j->w * 3 * j->idx, j->in.data(), j->in.size());
                j->in.clear(); j->in.shrink_to_fit();
            }
            if (open_order_.empty()) {
                if (stop_ && overflow_.empty()) break;
                // nothing gathered: a submit wakes this thread; overflow jobs wait for a slot, and everything that frees one or makes
                // one evictable (poll, release, the completer) notifies qcv_
                qcv_.wait(lk);
                continue;
            }
            const int si = open_order_.front();
            BatchSlot& S = slots_[si];
            slot_push_h2d(S);
            const bool full = (int)S.jobs.size() >= be_.max_batch();
            if (!full && !stop_ && S.status == IRE_OK && open_order_.size() == 1) {
                bool gpu_busy = false;
                if (last_launched_ >= 0) {
                    BatchSlot& P = slots_[last_launched_];
                    gpu_busy = P.state == BatchSlot::INFLIGHT && be_.computing(P.b);
                }
                // the completer notifies when that batch's compute ends (the bound only covers a lost wake-up)
                if (gpu_busy) { (void)wait_deadline(qcv_, lk, clk::now() + std::chrono::milliseconds(2)); continue; }
                const auto now = clk::now();
                const auto go = std::min(S.last_arrival + std::chrono::microseconds(kLingerQuietUs), S.first_arrival + std::chrono::microseconds(kLingerMaxUs));
                if (now < go) { (void)wait_deadline(qcv_, lk, go); continue; }     // an arrival re-evaluates; otherwise one wake-up at the deadline
            }
            if (trace_) {
                const auto now = clk::now();
                std::fprintf(stderr, "[batch] slot %d n %d full %d quiet_us %lld age_us %lld since_prev_launch_us %lld open %zu inflight %zu\n", si, (int)S.jobs.size(), (int)full,
                             (long long)std::chrono::duration_cast<std::chrono::microseconds>(now - S.last_arrival).count(),
                             (long long)std::chrono::duration_cast<std::chrono::microseconds>(now - S.first_arrival).count(),
                             (long long)std::chrono::duration_cast<std::chrono::microseconds>(now - last_launch_time_).count(), open_order_.size(), inflight_.size());
                last_launch_time_ = now;
            }
            // launch: no more reservations, wait for the copies still running in submitting threads
            S.state = BatchSlot::CLOSED;
            open_order_.pop_front();
            qcv_.wait(lk, [&] { for (auto& j : S.jobs) if (!j->staged) return false; return true; });
            slot_push_h2d(S);
            const int n = (int)S.jobs.size();
            bool wanted = false;
            for (auto& j : S.jobs) wanted = wanted || !j->abandoned;
            if (!wanted && S.status == IRE_OK) {       // every job of the batch was given up while it gathered: nothing to compute
                complete_jobs(S, ire_timings{});
                continue;
            }
            lk.unlock();
            try {
                if (S.status != IRE_OK) throw Error{S.status, S.err};
                be_.launch(S.b, n, S.h, S.w, S.has_sc);
            } catch (const Error& e) { S.status = e.code; S.err = e.msg; }
            catch (const std::exception& e) { S.status = IRE_ERR_INTERNAL; S.err = std::string("internal: ") + e.what(); }
            if (S.status != IRE_OK) be_.drain();   // whatever was enqueued before the failure may still touch the slot's buffers
            lk.lock();
            cnt_.batches += 1;
            if (S.status != IRE_OK) cnt_.failed_batches += 1;
            if (S.status == IRE_OK) {
                S.state = BatchSlot::INFLIGHT;
                inflight_.push_back(si);
                last_launched_ = si;
                ccv_.notify_all();
            } else {
                complete_jobs(S, ire_timings{});
                dcv_.notify_all();
            }
        }
        launcher_done_ = true;
        ccv_.notify_all();
    }

    void completer_loop() {
        be_.thread_enter("completer");
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            ccv_.wait(lk, [&] { return !inflight_.empty() || launcher_done_; });
            if (inflight_.empty()) break;
            BatchSlot& S = slots_[inflight_.front()];
            lk.unlock();
            be_.wait_compute(S.b);
            { std::lock_guard<std::mutex> g(mu_); }   // (the launcher is either before its check of computing() or already waiting: no lost wake-up)
            qcv_.notify_all();                     // the GPU is free: a batch that kept gathering behind this one may go
            ire_timings t{};
            int st = IRE_OK; std::string err;
            try { be_.wait_done(S.b, t); }
            catch (const Error& e) { st = e.code; err = e.msg; }
            lk.lock();
            if (st != IRE_OK) { S.status = st; S.err = err; }
            inflight_.pop_front();
            complete_jobs(S, t);
            dcv_.notify_all();
            qcv_.notify_all();
        }
    }

    Backend& be_;
    std::mutex mu_;
    std::condition_variable qcv_, dcv_, ccv_;      // launcher wake-ups | job completion | completer wake-ups
    BatchSlot slots_[kSlots];
    std::deque<int> open_order_;                   // OPEN slots, oldest first
    std::deque<int> inflight_;                     // INFLIGHT slots, launch order
    std::deque<std::shared_ptr<Job>> overflow_;    // submitted while no slot was free: staged by the launcher later
    int last_launched_ = -1;
    Counters cnt_;
    bool trace_ = false;                           // IRE_BATCH_TRACE=1: one stderr line per launched batch (why it went, how long it gathered)
    clk::time_point last_launch_time_ = clk::now();
    std::thread launcher_, completer_;
    bool stop_ = false, launcher_done_ = false, started_ = false;
};

}  // namespace ire
