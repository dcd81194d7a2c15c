// ire_napi.cc -- thin N-API shim over the C ABI of libire.so (include/ire.h) for the server-node worker.
//
// BASELINE.json's north_star names node-ffi-napi; that module is not installable offline, so
// this shim is the guaranteed path (N-API v8 headers ship with node: /usr/include/node/node_api.h) and
// node/ire_ffi.mjs keeps the equivalent ffi-napi declaration.  Every engine call resolves a Promise, because
// the reference's seams are asynchronous (classifier.analyze / geminiClient.restoreImage are awaited:
// restorator.js:59-94) and must tolerate >= 8 calls in flight.  The short synchronous calls (classify, fuse,
// preprocess, multi-image restore) run as napi async work on the libuv pool.  A single-image restore --
// the reference's unit of work -- is queued with the engine's batcher on the JS thread (ire_submit: one copy,
// Buffer -> pinned staging) and waited for by this addon's OWN waiter threads (ire_poll with the job's
// timeout, writing straight into the result Buffer), completed through a thread-safe function: the libuv
// pool (4 threads shared with fs / dns / zlib / crypto) is never parked for the length of a GPU batch, and
// a wedged engine rejects with ENGINE_TIMEOUT so the worker's retry -> DLQ path is reached -- the timeout counts from SUBMIT (a deadline
// stamped on the JS thread, not from the moment one of the four waiter threads picks the job up), and a job that timed out is given back to
// the engine (ire_job_release) so that its handle and its place in the batch's staging slot are freed before the retry resubmits the image.
// The waiter threads run on the CPUs the engine bound its own service threads to (ire_engine_affinity: the GPU's NUMA node).  libire.so is dlopen'ed at run time so the addon builds and loads on GPU-less hosts;
// without a device ire_init fails and the rejection text contains "service unavailable".
//
// Build: g++ -O2 -shared -fPIC -I/usr/include/node ire_napi.cc -o ire_napi.node -ldl
#include <dlfcn.h>
#include <node_api.h>
#include <pthread.h>
#include <sched.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ire.h"
#include "../csrc/affinity.hpp"      // parse_cpulist (host-only header)

namespace {

struct Api {
    void* so = nullptr;
    decltype(&ire_init) init = nullptr;
    decltype(&ire_shutdown) shutdown = nullptr;
    decltype(&ire_last_error) last_error = nullptr;
    decltype(&ire_classify) classify = nullptr;
    decltype(&ire_restore) restore = nullptr;
    decltype(&ire_fuse) fuse = nullptr;
    decltype(&ire_abi_version) abi_version = nullptr;
    decltype(&ire_submit) submit = nullptr;
    decltype(&ire_poll) poll = nullptr;
    decltype(&ire_job_release) job_release = nullptr;
    decltype(&ire_engine_affinity) engine_affinity = nullptr;
    decltype(&ire_png_base64_bytes) png_base64_bytes = nullptr;
    decltype(&ire_preprocess_plan) preprocess_plan = nullptr;
    decltype(&ire_preprocess) preprocess = nullptr;
    decltype(&ire_get_stats) get_stats = nullptr;
    decltype(&ire_max_batch_for) max_batch_for = nullptr;
} g;

bool load_api(const char* path, std::string* err) {
    if (g.so) return true;
    g.so = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!g.so) { *err = std::string("service unavailable: cannot load ") + path + ": " + dlerror(); return false; }
#define SYM(field, name) g.field = (decltype(g.field))dlsym(g.so, name); if (!g.field) { *err = "service unavailable: missing symbol " name; return false; }
    SYM(abi_version, "ire_abi_version")
    if (g.abi_version() != IRE_ABI_VERSION) { *err = "service unavailable: libire.so ABI version mismatch"; return false; }     // before the symbols a stale library lacks
    SYM(init, "ire_init") SYM(shutdown, "ire_shutdown") SYM(last_error, "ire_last_error") SYM(classify, "ire_classify")
    SYM(restore, "ire_restore") SYM(fuse, "ire_fuse")
    SYM(submit, "ire_submit") SYM(poll, "ire_poll") SYM(job_release, "ire_job_release") SYM(engine_affinity, "ire_engine_affinity") SYM(png_base64_bytes, "ire_png_base64_bytes") SYM(preprocess_plan, "ire_preprocess_plan") SYM(preprocess, "ire_preprocess")
    SYM(get_stats, "ire_get_stats") SYM(max_batch_for, "ire_max_batch_for")
#undef SYM
    return true;
}

void throw_err(napi_env env, const std::string& m) { napi_throw_error(env, nullptr, m.c_str()); }

int64_t get_i64(napi_env env, napi_value v) { int64_t x = 0; napi_get_value_int64(env, v, &x); return x; }
double get_f64(napi_env env, napi_value v) { double x = 0; napi_get_value_double(env, v, &x); return x; }

// ---- async job -------------------------------------------------------------------------------------
struct Job {
    enum Kind { CLASSIFY, RESTORE, FUSE, PREPROCESS } kind;
    ire_job* queued = nullptr;     // RESTORE with n == 1: already in the engine's batcher (ire_submit ran on the JS thread)
    int orientation = 1, max_dim = 2048, out_w = 0, out_h = 0, resized = 0;
    bool has_scores = false;
    ire_engine* eng;
    std::vector<uint8_t> in, jpeg, out;
    int n, h, w;
    double noise;
    std::vector<double> scores;
    std::vector<int32_t> labels;
    ire_timings t{};
    int status = 0;
    std::string err;
    napi_deferred deferred;
    napi_async_work work = nullptr;
    // single-image restore through the batcher (waiter threads): the result Buffer is created up front on the JS thread and
    // ire_poll writes into it; the engine handle is referenced so that it cannot be finalized under a pending job
    int timeout_ms = 120000;
    std::chrono::steady_clock::time_point deadline;      // submit time + timeout_ms
    uint8_t* out_ptr = nullptr;
    napi_ref out_ref = nullptr, eng_ref = nullptr;
};

// ---- waiter threads + thread-safe completion (single-image restores) --------------------------------
struct Waiters {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<struct Job*> q;
    napi_threadsafe_function tsfn = nullptr;
    int pending = 0;              // JS thread only: jobs handed to the waiters and not completed yet (keeps the tsfn referenced)
    bool started = false;
    std::string cpulist;          // the engine's service-thread CPUs (ire_engine_affinity) at the time the waiters were started
};
Waiters* W = nullptr;             // leaked on purpose: detached threads may outlive static destruction
constexpr int kWaiterThreads = 4;

void execute(napi_env, void* data) {
    Job* j = (Job*)data;
    switch (j->kind) {
        case Job::CLASSIFY:
            j->scores.resize(7 * (size_t)j->n); j->labels.resize(j->n);
            j->status = g.classify(j->eng, j->in.data(), j->n, j->h, j->w, 3 * j->w, j->jpeg.data(), j->scores.data(), j->labels.data());
            break;
        case Job::RESTORE:
            j->out.resize((size_t)j->n * j->h * j->w * 3);      // (n > 1 only: single images go through the batcher + waiter threads)
            j->status = g.restore(j->eng, j->in.data(), j->n, j->h, j->w, j->has_scores ? j->scores.data() : nullptr, j->jpeg.data(), j->out.data(), &j->t);
            break;
        case Job::PREPROCESS:
            j->out.resize((size_t)j->out_h * j->out_w * 3);
            j->status = g.preprocess(j->eng, j->in.data(), j->h, j->w, j->orientation, j->max_dim, j->out.data(), j->out_h, j->out_w);
            break;
        case Job::FUSE:
            j->out.resize((size_t)j->h * j->w * 3); j->labels.resize(2 * j->n);
            j->status = g.fuse(j->eng, j->in.data(), j->n, j->h, j->w, j->noise, j->out.data(), j->labels.data(), &j->t);
            break;
    }
    if (j->status != 0) j->err = g.last_error();   // thread-local: read on the worker thread
}

void finish(napi_env env, Job* j) {
    if (j->status != 0) {
        napi_value msg, err, code;
        napi_create_string_utf8(env, j->err.c_str(), NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, nullptr, msg, &err);
        const char* codes[] = {"", "ENGINE_INVALID_INPUT", "ENGINE_TIMEOUT", "ENGINE_UNAVAILABLE", "ENGINE_INTERNAL"};
        napi_create_string_utf8(env, codes[j->status >= 1 && j->status <= 4 ? j->status : 4], NAPI_AUTO_LENGTH, &code);
        napi_set_named_property(env, err, "code", code);     // restorator.js:159 propagates error.code
        napi_reject_deferred(env, j->deferred, err);
    } else {
        napi_value res;
        napi_create_object(env, &res);
        if (j->kind == Job::CLASSIFY) {
            napi_value ab, ta; void* p;
            napi_create_arraybuffer(env, j->scores.size() * 8, &p, &ab); std::memcpy(p, j->scores.data(), j->scores.size() * 8);
            napi_create_typedarray(env, napi_float64_array, j->scores.size(), ab, 0, &ta);
            napi_set_named_property(env, res, "scores", ta);
            napi_create_arraybuffer(env, j->labels.size() * 4, &p, &ab); std::memcpy(p, j->labels.data(), j->labels.size() * 4);
            napi_create_typedarray(env, napi_int32_array, j->labels.size(), ab, 0, &ta);
            napi_set_named_property(env, res, "labels", ta);
        } else {
            napi_value buf; void* p;
            if (j->out_ref) napi_get_reference_value(env, j->out_ref, &buf);       // ire_poll wrote into it: no further copy
            else napi_create_buffer_copy(env, j->out.size(), j->out.data(), &p, &buf);
            napi_set_named_property(env, res, "pixels", buf);
            napi_value ms; napi_create_double(env, j->t.restore_ms, &ms); napi_set_named_property(env, res, "restore_ms", ms);
            napi_create_double(env, j->t.classify_ms, &ms); napi_set_named_property(env, res, "classify_ms", ms);
            if (j->kind == Job::PREPROCESS) {
                napi_value v;
                napi_create_int32(env, j->out_w, &v); napi_set_named_property(env, res, "width", v);
                napi_create_int32(env, j->out_h, &v); napi_set_named_property(env, res, "height", v);
                napi_get_boolean(env, j->resized != 0, &v); napi_set_named_property(env, res, "resized", v);
            }
            if (j->kind == Job::RESTORE && j->scores.size() >= 7) {      // the batcher's classification comes back with the pixels
                napi_value ab2, ta2; void* p2;
                napi_create_arraybuffer(env, 7 * 8, &p2, &ab2); std::memcpy(p2, j->scores.data(), 7 * 8);
                napi_create_typedarray(env, napi_float64_array, 7, ab2, 0, &ta2);
                napi_set_named_property(env, res, "scores", ta2);
            }
            if (j->kind == Job::FUSE) {
                napi_value ab, ta;
                napi_create_arraybuffer(env, j->labels.size() * 4, &p, &ab); std::memcpy(p, j->labels.data(), j->labels.size() * 4);
                napi_create_typedarray(env, napi_int32_array, j->labels.size(), ab, 0, &ta);
                napi_set_named_property(env, res, "shifts", ta);
            }
        }
        napi_resolve_deferred(env, j->deferred, res);
    }
    if (j->out_ref) napi_delete_reference(env, j->out_ref);
    if (j->eng_ref) napi_delete_reference(env, j->eng_ref);
    if (j->work) napi_delete_async_work(env, j->work);
    delete j;
}
void complete(napi_env env, napi_status, void* data) { finish(env, (Job*)data); }

// thread-safe function body (JS thread): one waited-for job is done
void complete_ts(napi_env env, napi_value, void*, void* data) {
    if (!env) return;                 // the environment is going away: nothing to resolve
    finish(env, (Job*)data);
    if (--W->pending == 0) napi_unref_threadsafe_function(env, W->tsfn);      // idle: do not keep the event loop alive
}
void waiter_main() {
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        int n = 0;
        for (auto& r : ire::parse_cpulist(W->cpulist)) for (int c = r.first; c <= r.second; ++c) if (c >= 0 && c < CPU_SETSIZE) { CPU_SET(c, &set); ++n; }
        if (n) (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    }
    for (;;) {
        Job* j;
        {
            std::unique_lock<std::mutex> lk(W->mu);
            W->cv.wait(lk, [] { return !W->q.empty(); });
            j = W->q.front(); W->q.pop_front();
        }
        j->scores.resize(7);
        // what is left of the job's timeout (stamped at submit: a job that queued behind three wedged polls does not get a fresh allowance)
        const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(j->deadline - std::chrono::steady_clock::now()).count();
        j->status = g.poll(j->eng, j->queued, left > 0 ? (int)left : 0, j->out_ptr, j->scores.data(), &j->t);
        if (j->status != 0) j->err = g.last_error();          // thread-local: read on this thread
        // IRE_ERR_TIMEOUT leaves the job pending by contract: nobody will poll it again (the promise rejects, the retry submits anew),
        // so give it back -- its handle, its Job and its unread place in the batch's slot are freed now, not at the eighth slot's eviction
        if (j->status == IRE_ERR_TIMEOUT) (void)g.job_release(j->eng, j->queued);
        j->queued = nullptr;
        napi_call_threadsafe_function(W->tsfn, j, napi_tsfn_blocking);
    }
}
bool waiters_start(napi_env env, ire_engine* eng) {
    if (!W) W = new Waiters();
    if (W->started) return true;
    napi_value name;
    napi_create_string_utf8(env, "ire-restore-done", NAPI_AUTO_LENGTH, &name);
    if (napi_create_threadsafe_function(env, nullptr, nullptr, name, 0, 1, nullptr, nullptr, nullptr, complete_ts, &W->tsfn) != napi_ok) return false;
    napi_unref_threadsafe_function(env, W->tsfn);
    {
        char buf[4096] = {0};
        int32_t node = -1;
        if (eng && g.engine_affinity && g.engine_affinity(eng, buf, sizeof(buf), &node) == 0) W->cpulist = buf;
    }
    for (int i = 0; i < kWaiterThreads; ++i) std::thread(waiter_main).detach();
    W->started = true;
    return true;
}

// submit(kind, engineHandle(external), pixels Buffer, n, h, w, jpegFlags Buffer|null, noise) -> Promise
napi_value submit(napi_env env, napi_callback_info info, Job::Kind kind) {
    size_t argc = 9; napi_value argv[9];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    if (argc < 5) { throw_err(env, "invalid arguments"); return nullptr; }
    void* eng = nullptr;
    if (napi_get_value_external(env, argv[0], &eng) != napi_ok || !eng) { throw_err(env, "invalid engine handle"); return nullptr; }
    void* data; size_t len;
    bool is_buf = false;   // napi_get_buffer_info aborts the process on non-buffers in Node 12: test first
    if (napi_is_buffer(env, argv[1], &is_buf) != napi_ok || !is_buf || napi_get_buffer_info(env, argv[1], &data, &len) != napi_ok) {
        throw_err(env, "invalid input: pixels must be a Buffer");
        return nullptr;
    }
    Job* j = new Job();
    j->kind = kind; j->eng = (ire_engine*)eng;
    j->n = (int)get_i64(env, argv[2]); j->h = (int)get_i64(env, argv[3]); j->w = (int)get_i64(env, argv[4]);
    const size_t need = (size_t)(j->n > 0 ? j->n : 0) * (j->h > 0 ? j->h : 0) * (j->w > 0 ? j->w : 0) * 3;
    napi_value promise;
    napi_create_promise(env, &j->deferred, &promise);
    if (need == 0 || len < need) {
        napi_value msg, err;
        napi_create_string_utf8(env, "invalid input: pixel buffer smaller than n*h*w*3", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, nullptr, msg, &err);
        napi_reject_deferred(env, j->deferred, err);
        delete j;
        return promise;
    }
    const bool batched = kind == Job::RESTORE && j->n == 1;    // through ire_submit: the engine copies the Buffer itself (one copy)
    if (!batched) j->in.assign((uint8_t*)data, (uint8_t*)data + need);       // the caller's Buffer is not retained
    j->jpeg.assign(j->n, 1);
    if (argc > 5) {
        void* jd; size_t jl;
        bool jb = false;
        if (napi_is_buffer(env, argv[5], &jb) == napi_ok && jb && napi_get_buffer_info(env, argv[5], &jd, &jl) == napi_ok &&
            jl >= (size_t)j->n)
            std::memcpy(j->jpeg.data(), jd, j->n);
    }
    j->noise = (kind == Job::FUSE && argc > 6) ? get_f64(env, argv[6]) : -1.0;
    if (kind == Job::RESTORE && argc > 6) {      // scores of a previous analyze() of this image: the engine will not classify it again
        bool is_ta = false;
        if (napi_is_typedarray(env, argv[6], &is_ta) == napi_ok && is_ta) {
            napi_typedarray_type tt; size_t tl; void* td; napi_value tab; size_t toff;
            if (napi_get_typedarray_info(env, argv[6], &tt, &tl, &td, &tab, &toff) == napi_ok && tt == napi_float64_array && tl >= 7 * (size_t)j->n) {
                j->scores.assign((double*)td, (double*)td + 7 * (size_t)j->n);
                j->has_scores = true;
            }
        }
    }
    if (batched) {
        // single-image jobs (the reference's unit of work: restorator.js:198-211) go through the engine's batcher: queue NOW,
        // on the JS thread (ire_submit copies Buffer -> pinned staging: the only copy of the input), so that every in-flight
        // promise is in the batcher before anything waits; the waiter threads then ire_poll with the job's timeout
        if (argc > 7) { const int64_t tmo = get_i64(env, argv[7]); if (tmo > 0) j->timeout_ms = (int)(tmo > 0x7fffffff ? 0x7fffffff : tmo); }
        j->deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(j->timeout_ms);
        int rc = waiters_start(env, j->eng) ? 0 : 4;
        if (rc == 0) rc = g.submit(j->eng, (const uint8_t*)data, j->h, j->w, j->jpeg[0], j->has_scores ? j->scores.data() : nullptr, &j->queued);
        if (rc != 0) {
            napi_value msg, err, code;
            napi_create_string_utf8(env, W && W->started ? g.last_error() : "internal: cannot create the completion function", NAPI_AUTO_LENGTH, &msg);
            napi_create_error(env, nullptr, msg, &err);
            const char* codes[] = {"", "ENGINE_INVALID_INPUT", "ENGINE_TIMEOUT", "ENGINE_UNAVAILABLE", "ENGINE_INTERNAL"};
            napi_create_string_utf8(env, codes[rc >= 1 && rc <= 4 ? rc : 4], NAPI_AUTO_LENGTH, &code);
            napi_set_named_property(env, err, "code", code);
            napi_reject_deferred(env, j->deferred, err);
            delete j;
            return promise;
        }
        // (an engine created with IRE_FLAG_RESULT_PNG_BASE64 delivers text: the caller passes its size, pngBase64Bytes(h, w))
        size_t need_out = need;
        if (argc > 8) { const int64_t ob = get_i64(env, argv[8]); if (ob > 0) need_out = (size_t)ob; }
        napi_value outbuf; void* op = nullptr;
        napi_create_buffer(env, need_out, &op, &outbuf);           // the result: ire_poll writes into it from a waiter thread
        j->out_ptr = (uint8_t*)op;
        napi_create_reference(env, outbuf, 1, &j->out_ref);
        napi_create_reference(env, argv[0], 1, &j->eng_ref);     // the engine outlives its pending jobs
        if (W->pending++ == 0) napi_ref_threadsafe_function(env, W->tsfn);
        {
            std::lock_guard<std::mutex> lk(W->mu);
            W->q.push_back(j);
        }
        W->cv.notify_one();
        return promise;
    }
    napi_value name;
    napi_create_string_utf8(env, "ire", NAPI_AUTO_LENGTH, &name);
    napi_create_async_work(env, nullptr, name, execute, complete, j, &j->work);
    napi_queue_async_work(env, j->work);
    return promise;
}
napi_value classify_async(napi_env e, napi_callback_info i) { return submit(e, i, Job::CLASSIFY); }
napi_value restore_async(napi_env e, napi_callback_info i) { return submit(e, i, Job::RESTORE); }
napi_value fuse_async(napi_env e, napi_callback_info i) { return submit(e, i, Job::FUSE); }

// preprocessPlan(width, height, orientation, maxDim) -> {width, height, resized}   (imagePreprocess.js:12-22,46-55; host arithmetic)
napi_value preprocess_plan_sync(napi_env env, napi_callback_info info) {
    size_t argc = 4; napi_value argv[4];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    if (argc < 2 || !g.preprocess_plan) { throw_err(env, "invalid arguments (or engine library not loaded: call init first)"); return nullptr; }
    int ow = 0, oh = 0, rs = 0;
    const int rc = g.preprocess_plan((int)get_i64(env, argv[0]), (int)get_i64(env, argv[1]), argc > 2 ? (int)get_i64(env, argv[2]) : 1,
                                     argc > 3 ? (int)get_i64(env, argv[3]) : 2048, &ow, &oh, &rs);
    if (rc != 0) { throw_err(env, g.last_error()); return nullptr; }
    napi_value res, v;
    napi_create_object(env, &res);
    napi_create_int32(env, ow, &v); napi_set_named_property(env, res, "width", v);
    napi_create_int32(env, oh, &v); napi_set_named_property(env, res, "height", v);
    napi_get_boolean(env, rs != 0, &v); napi_set_named_property(env, res, "resized", v);
    return res;
}

// preprocessAsync(engine, pixels, h, w, orientation, maxDim) -> Promise<{pixels, width, height, resized}>: EXIF orient + fit-inside
// Lanczos-3 on the GPU (imagePreprocess.js:43-55); the JPEG q85 encode stays with the host codec
napi_value preprocess_async(napi_env env, napi_callback_info info) {
    size_t argc = 6; napi_value argv[6];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    if (argc < 4) { throw_err(env, "invalid arguments"); return nullptr; }
    void* eng = nullptr;
    if (napi_get_value_external(env, argv[0], &eng) != napi_ok || !eng) { throw_err(env, "invalid engine handle"); return nullptr; }
    void* data; size_t len; bool is_buf = false;
    if (napi_is_buffer(env, argv[1], &is_buf) != napi_ok || !is_buf || napi_get_buffer_info(env, argv[1], &data, &len) != napi_ok) {
        throw_err(env, "invalid input: pixels must be a Buffer");
        return nullptr;
    }
    Job* j = new Job();
    j->kind = Job::PREPROCESS; j->eng = (ire_engine*)eng; j->n = 1;
    j->h = (int)get_i64(env, argv[2]); j->w = (int)get_i64(env, argv[3]);
    j->orientation = argc > 4 ? (int)get_i64(env, argv[4]) : 1;
    j->max_dim = argc > 5 ? (int)get_i64(env, argv[5]) : 2048;
    napi_value promise;
    napi_create_promise(env, &j->deferred, &promise);
    const size_t need = (size_t)(j->h > 0 ? j->h : 0) * (j->w > 0 ? j->w : 0) * 3;
    int rc = (need == 0 || len < need) ? 1 : g.preprocess_plan(j->w, j->h, j->orientation, j->max_dim, &j->out_w, &j->out_h, &j->resized);
    if (rc != 0) {
        napi_value msg, err;
        napi_create_string_utf8(env, need == 0 || len < need ? "invalid input: pixel buffer smaller than h*w*3" : g.last_error(), NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, nullptr, msg, &err);
        napi_reject_deferred(env, j->deferred, err);
        delete j;
        return promise;
    }
    j->in.assign((uint8_t*)data, (uint8_t*)data + need);
    napi_value name;
    napi_create_string_utf8(env, "ire", NAPI_AUTO_LENGTH, &name);
    napi_create_async_work(env, nullptr, name, execute, complete, j, &j->work);
    napi_queue_async_work(env, j->work);
    return promise;
}

// stats(engine) -> {queueDepth, batches, images, lastBatch, maxBatch, imagesPerSec}   (f4: health entry + images/sec gauge)
napi_value stats_sync(napi_env env, napi_callback_info info) {
    size_t argc = 1; napi_value argv[1];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    void* eng = nullptr;
    if (argc < 1 || napi_get_value_external(env, argv[0], &eng) != napi_ok || !eng) { throw_err(env, "invalid engine handle"); return nullptr; }
    ire_engine_stats st; std::memset(&st, 0, sizeof(st)); st.struct_size = sizeof(st);
    if (g.get_stats((ire_engine*)eng, &st) != 0) { throw_err(env, g.last_error()); return nullptr; }
    napi_value res, v;
    napi_create_object(env, &res);
    napi_create_int32(env, st.queue_depth, &v); napi_set_named_property(env, res, "queueDepth", v);
    napi_create_int64(env, st.batches, &v); napi_set_named_property(env, res, "batches", v);
    napi_create_int64(env, st.images, &v); napi_set_named_property(env, res, "images", v);
    napi_create_int32(env, st.last_batch, &v); napi_set_named_property(env, res, "lastBatch", v);
    napi_create_int32(env, st.max_batch, &v); napi_set_named_property(env, res, "maxBatch", v);
    napi_create_double(env, st.images_per_sec, &v); napi_set_named_property(env, res, "imagesPerSec", v);
    return res;
}

// maxBatchFor(engine, h, w) -> images of that shape one call can take right now (0: unsupported / out of memory)
napi_value max_batch_for_sync(napi_env env, napi_callback_info info) {
    size_t argc = 3; napi_value argv[3];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    void* eng = nullptr;
    if (argc < 3 || napi_get_value_external(env, argv[0], &eng) != napi_ok || !eng) { throw_err(env, "invalid engine handle"); return nullptr; }
    napi_value v;
    napi_create_int32(env, g.max_batch_for((ire_engine*)eng, (int)get_i64(env, argv[1]), (int)get_i64(env, argv[2])), &v);
    return v;
}

void finalize_engine(napi_env, void* data, void*) { if (data && g.shutdown) g.shutdown((ire_engine*)data); }

// pngBase64Bytes(h, w) -> characters of the device-encoded result of an h x w image (0: unsupported size)
napi_value png_base64_bytes_sync(napi_env env, napi_callback_info info) {
    size_t argc = 2; napi_value argv[2];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    if (argc < 2 || !g.png_base64_bytes) { throw_err(env, "invalid arguments (or engine library not loaded: call init first)"); return nullptr; }
    napi_value v;
    napi_create_int64(env, (int64_t)g.png_base64_bytes((int)get_i64(env, argv[0]), (int)get_i64(env, argv[1])), &v);
    return v;
}

// init(libPath, weightsPath|null, deviceIndex, maxBatch, numStreams, flags) -> engine handle (throws with the engine's message)
napi_value init_engine(napi_env env, napi_callback_info info) {
    size_t argc = 6; napi_value argv[6];
    napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr);
    char lib[1024] = {0}, wts[1024] = {0}; size_t l = 0;
    if (argc < 1 || napi_get_value_string_utf8(env, argv[0], lib, sizeof(lib), &l) != napi_ok) { throw_err(env, "invalid arguments"); return nullptr; }
    bool have_w = argc > 1 && napi_get_value_string_utf8(env, argv[1], wts, sizeof(wts), &l) == napi_ok && l > 0;
    std::string err;
    if (!load_api(lib, &err)) { throw_err(env, err); return nullptr; }
    ire_config cfg; std::memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.device_index = argc > 2 ? (int)get_i64(env, argv[2]) : 0;
    cfg.max_batch = argc > 3 ? (int)get_i64(env, argv[3]) : 8;
    cfg.num_streams = argc > 4 ? (int)get_i64(env, argv[4]) : 0;
    cfg.weights_path = have_w ? wts : nullptr;
    cfg.flags = argc > 5 ? (uint32_t)get_i64(env, argv[5]) : 0;          // IRE_FLAG_RESULT_PNG_BASE64 = 1
    ire_engine* e = nullptr;
    const int rc = g.init(&cfg, &e);
    if (rc != 0) { throw_err(env, g.last_error()); return nullptr; }
    napi_value ext;
    napi_create_external(env, e, finalize_engine, nullptr, &ext);
    return ext;
}

napi_value module_init(napi_env env, napi_value exports) {
    napi_property_descriptor d[] = {
        {"init", nullptr, init_engine, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"classifyAsync", nullptr, classify_async, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"restoreAsync", nullptr, restore_async, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"fuseAsync", nullptr, fuse_async, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"preprocessPlan", nullptr, preprocess_plan_sync, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"preprocessAsync", nullptr, preprocess_async, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"stats", nullptr, stats_sync, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"maxBatchFor", nullptr, max_batch_for_sync, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"pngBase64Bytes", nullptr, png_base64_bytes_sync, nullptr, nullptr, nullptr, napi_default, nullptr},
    };
    napi_define_properties(env, exports, sizeof(d) / sizeof(d[0]), d);
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, module_init)
