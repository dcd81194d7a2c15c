// batcher_stress.cpp -- the engine's batcher state machine (image_restoration_platform_amd/csrc/batcher.hpp) over a host-only STUB
// backend, built with ThreadSanitizer and with AddressSanitizer + UBSan and driven from several threads by
// tests/test_batcher_native.py on the CPU box (sanitizers on the CPU build).  Test infrastructure: the stub
// "device" is a thread that runs launches in order, sleeps, and writes out = f(in); libire.so never contains it -- the product
// instantiates the same template over HIP (csrc/api.cpp: HipBatchBackend) and fails loudly without a GPU.
//
// What is driven: concurrent submitters of two shapes staging outside the lock, more jobs than kSlots * max_batch before the
// first poll (overflow queue, launcher-side staging, eviction of unread DONE slots), polls out of order with 0 / 1 ms timeouts
// that are retried, jobs abandoned with release() while pending / in flight / done / overflowing, a closed loop, failing
// allocations (the strong guarantee of reserve) and failing launches, destruction with jobs pending.  Every delivered result
// is compared with f(input); the process exits non-zero on any mismatch and the sanitizers report the rest.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>

#include "../../image_restoration_platform_amd/csrc/batcher.hpp"

using namespace ire;

static std::atomic<int> g_fail{0};
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); g_fail++; } } while (0)

static uint8_t f_px(uint8_t v, size_t k) { return (uint8_t)((v ^ 0x5a) + (uint8_t)(k * 7)); }

struct StubSlot {
    std::vector<uint8_t> d_in, d_out;
    std::mutex mu;
    std::condition_variable cv;
    bool compute_done = true, out_done = true;
};

struct StubBackend {
    int mb;
    std::atomic<int> reserves{0}, launches{0};
    int fail_reserve_every = 0, fail_launch_every = 0;
    // the "device": one thread, launches run in order
    std::mutex qmu;
    std::condition_variable qcv;
    std::deque<std::function<void()>> q;
    bool stop = false;
    std::thread dev;
    explicit StubBackend(int max_batch) : mb(max_batch) {
        dev = std::thread([this] {
            std::unique_lock<std::mutex> lk(qmu);
            for (;;) {
                qcv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                auto fn = std::move(q.front()); q.pop_front();
                lk.unlock(); fn(); lk.lock();
            }
        });
    }
    ~StubBackend() { { std::lock_guard<std::mutex> lk(qmu); stop = true; } qcv.notify_all(); dev.join(); }
    int max_batch() const { return mb; }
    size_t out_bytes(int h, int w) const { return (size_t)h * w * 3; }
    void start() {}
    void thread_enter(const char*) {}
    void reserve(SlotBufs& b, size_t bytes, int max_batch) {
        if (b.fixed && bytes <= b.cap) return;
        const int n = ++reserves;
        const bool boom = fail_reserve_every && n % fail_reserve_every == 0;
        // locals first, commit when everything exists (the contract under test: a throw leaves b as it was)
        std::unique_ptr<uint8_t[]> pj, pi, po;
        std::unique_ptr<double[]> ps, psi;
        const bool want_fixed = !b.fixed, want_img = bytes > b.cap;
        if (want_fixed) { pj.reset(new uint8_t[max_batch]()); ps.reset(new double[7 * max_batch]()); psi.reset(new double[7 * max_batch]()); }
        if (want_img) { pi.reset(new uint8_t[bytes]()); if (boom) throw Error{IRE_ERR_UNAVAILABLE, "service unavailable: injected allocation failure"}; po.reset(new uint8_t[bytes]()); }
        else if (boom) throw Error{IRE_ERR_UNAVAILABLE, "service unavailable: injected allocation failure"};
        StubSlot* ss = static_cast<StubSlot*>(b.impl);
        if (!ss) { ss = new StubSlot(); b.impl = ss; }
        if (want_fixed) { b.pin_jp = pj.release(); b.pin_sc = ps.release(); b.pin_sc_in = psi.release(); b.fixed = true; }
        if (want_img) {
            delete[] b.pin_in; delete[] b.pin_out;
            b.pin_in = pi.release(); b.pin_out = po.release(); b.cap = bytes;
            ss->d_in.assign(bytes, 0); ss->d_out.assign(bytes, 0);
        }
    }
    void release(SlotBufs& b) noexcept {
        delete[] b.pin_in; delete[] b.pin_out; delete[] b.pin_jp; delete[] b.pin_sc; delete[] b.pin_sc_in;
        delete static_cast<StubSlot*>(b.impl);
        b = SlotBufs{};
    }
    void h2d(SlotBufs& b, size_t off, size_t bytes) {
        StubSlot& ss = *static_cast<StubSlot*>(b.impl);
        std::memcpy(ss.d_in.data() + off, b.pin_in + off, bytes);
    }
    void launch(SlotBufs& b, int n, int h, int w, const uint8_t* has_sc) {
        const int k = ++launches;
        if (fail_launch_every && k % fail_launch_every == 0) throw Error{IRE_ERR_INTERNAL, "internal: injected launch failure"};
        StubSlot* ss = static_cast<StubSlot*>(b.impl);
        { std::lock_guard<std::mutex> lk(ss->mu); ss->compute_done = false; ss->out_done = false; }
        std::vector<uint8_t> hs(has_sc, has_sc + n);
        SlotBufs* bp = &b;
        std::lock_guard<std::mutex> lk(qmu);
        q.push_back([=] {
            const size_t ib = (size_t)h * w * 3;
            std::this_thread::sleep_for(std::chrono::microseconds(150));
            for (int i = 0; i < n; ++i) {
                for (size_t p = 0; p < ib; ++p) ss->d_out[ib * i + p] = f_px(ss->d_in[ib * i + p], p);
                for (int s = 0; s < 7; ++s) bp->pin_sc[7 * i + s] = hs[i] ? bp->pin_sc_in[7 * i + s] : (double)ss->d_in[ib * i] + s + (bp->pin_jp[i] ? 0.5 : 0.0);
            }
            { std::lock_guard<std::mutex> l2(ss->mu); ss->compute_done = true; }
            ss->cv.notify_all();
            std::this_thread::sleep_for(std::chrono::microseconds(40));
            std::memcpy(bp->pin_out, ss->d_out.data(), ib * n);
            { std::lock_guard<std::mutex> l2(ss->mu); ss->out_done = true; }
            ss->cv.notify_all();
        });
        qcv.notify_all();
    }
    bool computing(SlotBufs& b) noexcept { StubSlot& ss = *static_cast<StubSlot*>(b.impl); std::lock_guard<std::mutex> lk(ss.mu); return !ss.compute_done; }
    void wait_compute(SlotBufs& b) noexcept { StubSlot& ss = *static_cast<StubSlot*>(b.impl); std::unique_lock<std::mutex> lk(ss.mu); ss.cv.wait(lk, [&] { return ss.compute_done; }); }
    void wait_done(SlotBufs& b, ire_timings& t) { StubSlot& ss = *static_cast<StubSlot*>(b.impl); std::unique_lock<std::mutex> lk(ss.mu); ss.cv.wait(lk, [&] { return ss.out_done; }); t.restore_ms = 0.15; t.total_ms = 0.15; }
    void drain() noexcept {
        std::mutex m; std::condition_variable c; bool done = false;
        { std::lock_guard<std::mutex> lk(qmu); q.push_back([&] { std::lock_guard<std::mutex> l(m); done = true; c.notify_all(); }); }
        qcv.notify_all();
        std::unique_lock<std::mutex> lk(m);
        c.wait(lk, [&] { return done; });
    }
};

struct Img { int h, w; std::vector<uint8_t> px; bool with_scores; double sc[7]; int jpeg; };
static Img make_img(std::mt19937& rng, int shape) {
    Img im;
    im.h = shape ? 24 : 16; im.w = shape ? 32 : 16;
    im.px.resize((size_t)im.h * im.w * 3);
    for (auto& v : im.px) v = (uint8_t)rng();
    im.with_scores = rng() & 1; im.jpeg = rng() & 1;
    for (int s = 0; s < 7; ++s) im.sc[s] = (double)(rng() % 1000) / 1000.0;
    return im;
}
static void verify(const Img& im, const std::vector<uint8_t>& out, const double* sc) {
    bool ok = true;
    for (size_t p = 0; p < im.px.size() && ok; ++p) ok = out[p] == f_px(im.px[p], p);
    CHECK(ok);
    for (int s = 0; s < 7; ++s) CHECK(sc[s] == (im.with_scores ? im.sc[s] : (double)im.px[0] + s + (im.jpeg ? 0.5 : 0.0)));
}

using B = Batcher<StubBackend>;
struct Pending { Img im; std::shared_ptr<Job> j; };

// threads x jobs, every job submitted before the first poll; polls in random order with short timeouts; thread `abandoner` gives up every other job
static void burst(B& bt, int threads, int jobs, int abandoner, unsigned seed, int expect_fail_ok) {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            std::mt19937 rng(seed + 977 * t);
            std::vector<Pending> pend;
            for (int k = 0; k < jobs; ++k) {
                Pending p{make_img(rng, (k + t) & 1), nullptr};
                try { p.j = bt.submit(p.im.px.data(), p.im.h, p.im.w, p.im.jpeg, p.im.with_scores ? p.im.sc : nullptr); }
                catch (const Error& e) { CHECK(expect_fail_ok && e.code == IRE_ERR_UNAVAILABLE); continue; }
                pend.push_back(std::move(p));
            }
            std::shuffle(pend.begin(), pend.end(), rng);
            size_t released = 0;
            while (!pend.empty()) {
                for (size_t i = 0; i < pend.size();) {
                    Pending& p = pend[i];
                    if (t == abandoner && (released++ & 1)) { bt.release(p.j); pend.erase(pend.begin() + i); continue; }
                    std::vector<uint8_t> out(p.im.px.size());
                    double sc[7]; ire_timings tm{}; std::string err;
                    const int st = bt.poll(p.j, (int)(rng() % 2), out.data(), sc, &tm, &err);
                    if (st == IRE_ERR_TIMEOUT) {
                        if (t == abandoner && (rng() % 4) == 0) { bt.release(p.j); pend.erase(pend.begin() + i); continue; }   // a timed-out job given up
                        ++i; continue;
                    }
                    if (st == IRE_OK) verify(p.im, out, sc);
                    else CHECK(expect_fail_ok && !err.empty());
                    pend.erase(pend.begin() + i);
                }
            }
        });
    for (auto& t : th) t.join();
}

// a job released while its batch still gathers stays counted until that batch is launched or dropped (at most the linger bound)
static bool drained(B& bt) {
    for (int i = 0; i < 200 && bt.queue_depth() != 0; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    return bt.queue_depth() == 0;
}

static void closed_loop(B& bt, int threads, int inflight, int total) {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            std::mt19937 rng(4242 + t);
            std::deque<Pending> q;
            int sent = 0, got = 0;
            while (got < total) {
                while (sent < total && (int)q.size() < inflight) {
                    Pending p{make_img(rng, t & 1), nullptr};
                    p.j = bt.submit(p.im.px.data(), p.im.h, p.im.w, p.im.jpeg, p.im.with_scores ? p.im.sc : nullptr);
                    q.push_back(std::move(p)); ++sent;
                }
                Pending p = std::move(q.front()); q.pop_front();
                std::vector<uint8_t> out(p.im.px.size());
                double sc[7]; std::string err;
                const int st = bt.poll(p.j, -1, out.data(), sc, nullptr, &err);
                CHECK(st == IRE_OK);
                if (st == IRE_OK) verify(p.im, out, sc);
                ++got;
            }
        });
    for (auto& t : th) t.join();
}

int main() {
    {   // 1. bursts past every slot (max_batch 2: 8 slots hold 16 jobs; 4 x 16 = 64 submitted before the first poll), two shapes, an abandoner
        StubBackend be(2);
        B bt(be);
        for (unsigned round = 0; round < 6; ++round) {
            burst(bt, 4, 16, (int)(round % 4), 1 + 10 * round, 0);
            CHECK(drained(bt));
        }
        closed_loop(bt, 3, 5, 60);
        CHECK(drained(bt));
        const auto c = bt.counters();
        std::printf("burst: batches %ld overflowed %ld evicted %ld abandoned %ld\n", c.batches, c.overflowed, c.evicted, c.abandoned);
        CHECK(c.overflowed > 0 && c.evicted > 0 && c.abandoned > 0 && c.failed_batches == 0);
    }
    {   // 2. allocations and launches that fail: errors reach the jobs, nothing leaks, the service threads live on
        StubBackend be(4);
        be.fail_reserve_every = 5; be.fail_launch_every = 4;
        B* bt = nullptr;
        for (int tries = 0; tries < 8 && !bt; ++tries) {       // the eager reservation of the first submit may be the injected failure
            bt = new B(be);
            std::mt19937 rng(7);
            Img im = make_img(rng, 0);
            try { auto j = bt->submit(im.px.data(), im.h, im.w, 1, nullptr); std::vector<uint8_t> o(im.px.size()); double sc[7]; std::string e; (void)bt->poll(j, -1, o.data(), sc, nullptr, &e); }
            catch (const Error&) { delete bt; bt = nullptr; }
        }
        CHECK(bt != nullptr);
        if (bt) {
            burst(*bt, 4, 12, 1, 3, 1); burst(*bt, 4, 12, 2, 4, 1);
            CHECK(drained(*bt));
            const auto c = bt->counters();
            std::printf("failures: batches %ld failed %ld\n", c.batches, c.failed_batches);
            CHECK(c.failed_batches > 0);
            delete bt;
        }
    }
    {   // 3. destruction with jobs pending, gathered, in flight, done-unread and overflowing; the handles are dropped afterwards
        StubBackend be(2);
        std::vector<Pending> left;
        {
            B bt(be);
            std::mt19937 rng(99);
            for (int k = 0; k < 40; ++k) {
                Pending p{make_img(rng, k & 1), nullptr};
                p.j = bt.submit(p.im.px.data(), p.im.h, p.im.w, p.im.jpeg, nullptr);
                left.push_back(std::move(p));
            }
            for (int k = 0; k < 3; ++k) {
                std::vector<uint8_t> out(left[k].im.px.size()); double sc[7]; std::string err;
                const int st = bt.poll(left[k].j, -1, out.data(), sc, nullptr, &err);
                CHECK(st == IRE_OK);
            }
            bt.release(left[5].j);
        }
        left.clear();
    }
    if (g_fail.load()) { std::fprintf(stderr, "batcher_stress: %d check(s) failed\n", g_fail.load()); return 1; }
    std::puts("batcher_stress ok");
    return 0;
}
