// Host <-> device staging of the *_host entry points (the ones the reference's NumPy loop calls).
//
// Round 1 allocated ~10 device buffers per call and copied pageable memory synchronously: 13 ms for a 1 M-point call
// whose kernels take 0.1 ms.  Here:
//   * device buffers persist (per context for the fused step, per device for the mesh-free return map, grow-only);
//   * pinned host memory comes from a size-keyed cache (fep_host_alloc / fep_host_free): the Python layer allocates
//     every OUTPUT array (s, ds, ind_p, K data, F ...) there, so results are DMA-ed straight into the NumPy array the
//     caller receives — no staging copy on the way back;
//   * pageable INPUTS (the reference's `B @ U`, its `Ep_old`, the material arrays) go through a ring of pinned slots:
//     a few persistent worker threads copy chunk i+1 into its slot while the DMA engine moves chunk i
//     (hipMemcpyAsync); pageable outputs take the same ring the other way;
//   * everything is ordered on one stream per device; one hipStreamSynchronize at the end of the call.
// The three classes use six HIP calls (hipHostMalloc / hipHostFree / hipMalloc / hipFree / hipMemcpyAsync / stream and
// event handling).  tests/host_san.cpp defines FEP_STAGING_HOST_STUB and supplies host stand-ins for them (plain memory,
// synchronous copies, a switch that makes the next copy fail), so that PinnedCache, CopyPool and Engine run under
// ThreadSanitizer / AddressSanitizer on the CPU.
#pragma once
#ifndef FEP_STAGING_HOST_STUB
#include "fep_common.h"
#endif

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace fep_stage {

// ---------------------------------------------------------------------------------------
// pinned host memory, cached by size
// ---------------------------------------------------------------------------------------
struct PinnedCache {
    std::mutex m;
    struct Block { size_t bytes; bool idle; };
    std::map<const char*, Block> live;                  // every block handed out or cached: base -> (bytes, in the idle list?)
    std::multimap<size_t, void*> idle;                  // returned blocks, reused for requests of the same size
    size_t idle_bytes = 0;
    static constexpr size_t kIdleCap = (size_t)4 << 30;

    // frees every idle block (called when the runtime refuses to pin more memory; also what a caller can do to give the
    // pinned pages back: fep_host_trim)
    int trim() {
        std::vector<void*> drop;
        {
            std::lock_guard<std::mutex> g(m);
            for (auto& kv : idle) { drop.push_back(kv.second); live.erase((const char*)kv.second); }
            idle.clear();
            idle_bytes = 0;
        }
        int r = FEP_OK;
        for (void* p : drop)
            if (hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); r = FEP_EHIP; }
        return r;
    }
    int alloc(void** out, size_t bytes) {
        if (bytes == 0) bytes = 1;
        {
            std::lock_guard<std::mutex> g(m);
            auto it = idle.find(bytes);
            if (it != idle.end()) {
                *out = it->second;
                live[(const char*)it->second].idle = false;
                idle_bytes -= bytes;
                idle.erase(it);
                return FEP_OK;
            }
        }
        void* p = nullptr;
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) {                           // pinned memory is a limited resource: give the idle blocks back, retry once
            (void)hipGetLastError();
            (void)trim();
            e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        }
        if (e != hipSuccess) { fep_g_last_hip = (int)e; (void)hipGetLastError(); return e == hipErrorOutOfMemory ? FEP_ENOMEM : FEP_EHIP; }
        std::lock_guard<std::mutex> g(m);
        live[(const char*)p] = Block{bytes, false};
        *out = p;
        return FEP_OK;
    }
    int release(void* p) {
        if (!p) return FEP_OK;
        std::unique_lock<std::mutex> g(m);
        auto it = live.find((const char*)p);
        if (it == live.end() || it->second.idle) return FEP_EINVAL;          // not ours, or released twice
        const size_t bytes = it->second.bytes;
        if (idle_bytes + bytes <= kIdleCap) { it->second.idle = true; idle.emplace(bytes, p); idle_bytes += bytes; return FEP_OK; }
        live.erase(it);
        g.unlock();
        HIP_TRY(hipHostFree(p));
        return FEP_OK;
    }
    // is [p, p+bytes) inside a block of this cache (i.e. page-locked memory we may DMA from / to directly)?
    bool covers(const void* p, size_t bytes) {
        std::lock_guard<std::mutex> g(m);
        auto it = live.upper_bound((const char*)p);
        if (it == live.begin()) return false;
        --it;
        return (const char*)p + bytes <= it->first + it->second.bytes;
    }
};
inline PinnedCache& pinned() { static PinnedCache c; return c; }

// ---------------------------------------------------------------------------------------
// persistent copy workers: memcpy split over threads (a single core moves ~10 GB/s, the link ~55)
// ---------------------------------------------------------------------------------------
class CopyPool {
    std::vector<std::thread> th;
    std::mutex m, submit;
    std::condition_variable cv_job, cv_done;
    void (*job)(void*, int, int) = nullptr;             // job(ctx, worker, n_workers)
    void* job_ctx = nullptr;
    unsigned long long gen = 0;
    int pending = 0;
    bool stop = false;
    int n = 1;

    void run(int w) {
        unsigned long long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> g(m);
            cv_job.wait(g, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            void (*f)(void*, int, int) = job;
            void* c = job_ctx;
            g.unlock();
            f(c, w, n);
            g.lock();
            if (--pending == 0) cv_done.notify_all();
        }
    }
    struct Cpy { char* dst; const char* src; size_t bytes; };
    static void cpy_piece(void* c, int w, int n) {
        const Cpy& k = *(const Cpy*)c;
        const size_t lo = (k.bytes * (size_t)w / (size_t)n) & ~(size_t)63;
        const size_t hi = w + 1 == n ? k.bytes : (k.bytes * (size_t)(w + 1) / (size_t)n) & ~(size_t)63;
        if (hi > lo) std::memcpy(k.dst + lo, k.src + lo, hi - lo);
    }
    struct Ilv { double* dst; const double* x; const double* y; size_t n; };
    static void ilv_piece(void* c, int w, int n) {
        const Ilv& k = *(const Ilv*)c;
        const size_t lo = k.n * (size_t)w / (size_t)n, hi = k.n * (size_t)(w + 1) / (size_t)n;
        for (size_t i = lo; i < hi; ++i) { k.dst[2 * i] = k.x[i]; k.dst[2 * i + 1] = k.y[i]; }
    }

public:
    CopyPool() {
        int want = 8;
        if (const char* e = std::getenv("FEP_COPY_THREADS")) want = std::max(1, std::min(std::atoi(e), 64));
        want = std::min<int>(want, (int)std::max(1u, std::thread::hardware_concurrency()));
        n = 1;
        for (int w = 1; w < want; ++w) {
            try { th.emplace_back([this, w] { run(w); }); ++n; }
            catch (...) { break; }                       // thread limits: fewer workers, never an exception
        }
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> g(m); stop = true; }
        cv_job.notify_all();
        for (auto& t : th) t.join();
    }
    // f(ctx, w, n) for w = 0..n-1, worker 0 on the calling thread; returns when all are done
    // One job at a time: the pool is process-wide, and two host threads driving *_host calls on two GPUs (ctypes
    // releases the GIL) would otherwise overwrite each other's job before the workers woke up.
    void parallel(void (*f)(void*, int, int), void* ctx) {
        if (n == 1) { f(ctx, 0, 1); return; }
        std::lock_guard<std::mutex> one(submit);
        {
            std::lock_guard<std::mutex> g(m);
            job = f; job_ctx = ctx; pending = n - 1; ++gen;
        }
        cv_job.notify_all();
        f(ctx, 0, n);
        std::unique_lock<std::mutex> g(m);
        cv_done.wait(g, [&] { return pending == 0; });
    }
    void copy(void* d, const void* s, size_t b) {
        if (b < ((size_t)1 << 20)) { std::memcpy(d, s, b); return; }
        Cpy k{(char*)d, (const char*)s, b};
        parallel(cpy_piece, &k);
    }
    // dst[2i] = x[i], dst[2i+1] = y[i]: the reference's (2, n_n) displacement -> DOF order (DP:1043 flattens it 'F')
    void interleave2(double* d, const double* x, const double* y, size_t cnt) {
        Ilv k{d, x, y, cnt};
        if (cnt < ((size_t)1 << 16)) { ilv_piece(&k, 0, 1); return; }
        parallel(ilv_piece, &k);
    }
};
inline CopyPool& pool() { static CopyPool p; return p; }

// ---------------------------------------------------------------------------------------
// per-device engine: one stream, a ring of pinned slots, persistent device buffers
// ---------------------------------------------------------------------------------------
struct Engine {
    static constexpr int kSlots = 4;
    static constexpr size_t kSlotBytes = (size_t)8 << 20;
    int device = -1;
    std::mutex call;                                     // the *_host entry points of one device run one at a time
    hipStream_t stream = nullptr;
    void* slot[kSlots] = {};
    hipEvent_t ev[kSlots] = {};
    struct Pending { void* dst = nullptr; size_t bytes = 0; } pend[kSlots];
    int next = 0;
    std::vector<std::pair<void*, size_t>> dev;           // persistent device buffers by index (grow-only)

    int init(int d) {
        device = d;
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (int i = 0; i < kSlots; ++i) {
            FEP_TRY(pinned().alloc(&slot[i], kSlotBytes));
            HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
        return FEP_OK;
    }
    int buffer(int idx, size_t bytes, void** out) {
        if ((int)dev.size() <= idx) dev.resize(idx + 1, {nullptr, 0});
        if (bytes == 0) bytes = 1;
        if (dev[idx].second < bytes) {
            if (dev[idx].first) { HIP_TRY(hipStreamSynchronize(stream)); HIP_TRY(hipFree(dev[idx].first)); dev[idx] = {nullptr, 0}; }
            void* p = nullptr;
            HIP_TRY(hipMalloc(&p, bytes));
            dev[idx] = {p, bytes};
        }
        *out = dev[idx].first;
        return FEP_OK;
    }
    // the slot's DMA has finished; a device -> host chunk parked in it goes to its pageable destination
    int settle(int i) {
        HIP_TRY(hipEventSynchronize(ev[i]));
        if (pend[i].dst) { pool().copy(pend[i].dst, slot[i], pend[i].bytes); pend[i] = Pending(); }
        return FEP_OK;
    }
    int h2d(void* dst_d, const void* src_h, size_t bytes) {
        if (bytes == 0) return FEP_OK;
        if (pinned().covers(src_h, bytes)) { HIP_TRY(hipMemcpyAsync(dst_d, src_h, bytes, hipMemcpyHostToDevice, stream)); return FEP_OK; }
        for (size_t off = 0; off < bytes; off += kSlotBytes) {
            const size_t b = std::min(kSlotBytes, bytes - off);
            const int i = next; next = (next + 1) % kSlots;
            FEP_TRY(settle(i));
            pool().copy(slot[i], (const char*)src_h + off, b);
            HIP_TRY(hipMemcpyAsync((char*)dst_d + off, slot[i], b, hipMemcpyHostToDevice, stream));
            HIP_TRY(hipEventRecord(ev[i], stream));
        }
        return FEP_OK;
    }
    // (2, n) planar host array -> n interleaved pairs on the device; the interleave happens in the staging copy
    int h2d_interleave2(void* dst_d, const double* src_h, size_t n) {
        const size_t per = kSlotBytes / 16;
        for (size_t off = 0; off < n; off += per) {
            const size_t cnt = std::min(per, n - off);
            const int i = next; next = (next + 1) % kSlots;
            FEP_TRY(settle(i));
            pool().interleave2((double*)slot[i], src_h + off, src_h + n + off, cnt);
            HIP_TRY(hipMemcpyAsync((char*)dst_d + off * 16, slot[i], cnt * 16, hipMemcpyHostToDevice, stream));
            HIP_TRY(hipEventRecord(ev[i], stream));
        }
        return FEP_OK;
    }
    int d2h(void* dst_h, const void* src_d, size_t bytes) {
        if (bytes == 0) return FEP_OK;
        if (pinned().covers(dst_h, bytes)) { HIP_TRY(hipMemcpyAsync(dst_h, src_d, bytes, hipMemcpyDeviceToHost, stream)); return FEP_OK; }
        for (size_t off = 0; off < bytes; off += kSlotBytes) {
            const size_t b = std::min(kSlotBytes, bytes - off);
            const int i = next; next = (next + 1) % kSlots;
            FEP_TRY(settle(i));
            HIP_TRY(hipMemcpyAsync(slot[i], (const char*)src_d + off, b, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipEventRecord(ev[i], stream));
            pend[i].dst = (char*)dst_h + off; pend[i].bytes = b;
        }
        return FEP_OK;
    }
    int finish() {
        for (int k = 0; k < kSlots; ++k) { const int i = (next + k) % kSlots; if (pend[i].dst) FEP_TRY(settle(i)); }
        HIP_TRY(hipStreamSynchronize(stream));
        return FEP_OK;
    }
    // After a failed call: wait for whatever is still in flight and FORGET the parked device -> host chunks — their
    // destinations belong to the failed call's output arrays, which the caller is about to free (a later settle() must
    // never copy into them).  The ring restarts at slot 0.
    void drain() {
        (void)hipStreamSynchronize(stream);
        (void)hipGetLastError();
        for (int i = 0; i < kSlots; ++i) pend[i] = Pending();
        next = 0;
    }
};

// Scope of one *_host entry point: holds the engine's call lock; unless finish() succeeded (disarm) the engine is drained
// on the way out, whatever the exit path (error return or exception).
struct EngineCall {
    Engine* E;
    std::unique_lock<std::mutex> lock;
    bool armed = true;
    explicit EngineCall(Engine* e) : E(e), lock(e->call) {}
    ~EngineCall() { if (armed) E->drain(); }
    int finish() { const int r = E->finish(); if (r == FEP_OK) armed = false; return r; }
};

// engine of a device (created on first use, lives as long as the library)
inline int engine(int device, Engine** out) {
    static std::mutex m;
    static std::map<int, Engine*>& all = *new std::map<int, Engine*>();      // never destroyed: engines outlive static destructors
    std::lock_guard<std::mutex> g(m);
    auto it = all.find(device);
    if (it == all.end()) {
        Engine* e = new (std::nothrow) Engine();
        if (!e) return FEP_ENOMEM;
        const int r = e->init(device);
        if (r != FEP_OK) return r;                       // (a half-built engine is leaked: the device is unusable anyway)
        it = all.emplace(device, e).first;
    }
    *out = it->second;
    return FEP_OK;
}

}  // namespace fep_stage
