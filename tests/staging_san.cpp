// Host-only driver of fem-elastoplasticity_amd/csrc/fep_staging.h for the sanitizer builds (tests/test_host_sanitizers.py:
// g++ -fsanitize=address,undefined and -fsanitize=thread; CPU only).  The header's six kinds of HIP calls are replaced by
// the host stand-ins below (FEP_STAGING_HOST_STUB): "device" memory is heap memory, copies are synchronous memcpy, and
// `g_fail_copy_after` / `g_fail_host_malloc` make the n-th copy / the next pinned allocation fail.
//   1. PinnedCache: reuse by size, double release refused, trim, allocation failure -> idle blocks given back -> retry
//   2. CopyPool: copy / interleave2 from two threads at once (the pool is process-wide; ADVICE r2: jobs overwrote each other)
//   3. Engine: pageable round trips through the ring; a copy that fails after a device -> host chunk was parked must leave
//      no pending destination behind (the failed call's output array is freed right after: ASan sees any later write)
//   4. two engines driven by two threads
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/fep.h"

#define FEP_STAGING_HOST_STUB 1
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
typedef struct StubStream* hipStream_t;
typedef struct StubEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static thread_local int fep_g_last_hip = 0;
static std::atomic<long> g_copies{0}, g_fail_copy_after{-1};
static std::atomic<int> g_fail_host_malloc{0};
static hipError_t hipGetLastError() { return hipSuccess; }
static hipError_t hipHostMalloc(void** p, size_t b, unsigned) {
    if (g_fail_host_malloc.load() > 0) { g_fail_host_malloc.fetch_sub(1); *p = nullptr; return hipErrorOutOfMemory; }
    *p = std::malloc(b);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static hipError_t hipMalloc(void** p, size_t b) { *p = std::malloc(b); return *p ? hipSuccess : hipErrorOutOfMemory; }
static hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)std::malloc(1); return hipSuccess; }
static hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)std::malloc(1); return hipSuccess; }
static hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static hipError_t hipMemcpyAsync(void* d, const void* s, size_t b, hipMemcpyKind, hipStream_t) {
    const long k = g_copies.fetch_add(1);
    const long f = g_fail_copy_after.load();
    if (f >= 0 && k >= f) return hipErrorUnknown;
    std::memcpy(d, s, b);
    return hipSuccess;
}
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { fep_g_last_hip = (int)_e; return _e == hipErrorOutOfMemory ? FEP_ENOMEM : FEP_EHIP; } } while (0)
#define FEP_TRY(expr) do { int _r = (expr); if (_r != FEP_OK) return _r; } while (0)

#include "../fem-elastoplasticity_amd/csrc/fep_staging.h"

using namespace fep_stage;

#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

static int test_pinned() {
    PinnedCache pc;
    void *a = nullptr, *b = nullptr, *c = nullptr;
    CHECK(pc.alloc(&a, 4096) == FEP_OK && pc.alloc(&b, 8192) == FEP_OK);
    CHECK(pc.covers((char*)a + 100, 3000) && !pc.covers((char*)a + 100, 5000));
    CHECK(pc.release(a) == FEP_OK);
    CHECK(pc.release(a) == FEP_EINVAL);                  // released twice: refused, not queued twice
    int dummy;
    CHECK(pc.release(&dummy) == FEP_EINVAL);             // not a block of the cache
    CHECK(pc.alloc(&c, 4096) == FEP_OK && c == a);       // same size: the idle block again
    CHECK(pc.release(c) == FEP_OK && pc.release(b) == FEP_OK && pc.idle_bytes == 4096 + 8192);
    g_fail_host_malloc = 1;                              // the runtime refuses once: idle blocks are given back, then it works
    void* d = nullptr;
    CHECK(pc.alloc(&d, 1 << 20) == FEP_OK && pc.idle_bytes == 0 && pc.idle.empty() && pc.live.size() == 1);
    g_fail_host_malloc = 2;                              // refuses twice: FEP_ENOMEM, nothing leaked
    void* e = nullptr;
    CHECK(pc.alloc(&e, 1 << 20) == FEP_ENOMEM && e == nullptr);
    CHECK(pc.release(d) == FEP_OK && pc.trim() == FEP_OK && pc.live.empty());
    return 0;
}

static int pool_worker(int seed, int* rc) {
    std::vector<double> src((size_t)3 << 18), dst(src.size()), il(2 * ((size_t)1 << 17));
    for (int rep = 0; rep < 12; ++rep) {
        for (size_t i = 0; i < src.size(); ++i) src[i] = (double)(seed * 1000003 + rep * 7919 + (long)i);
        pool().copy(dst.data(), src.data(), src.size() * sizeof(double));              // 6 MiB: split over the pool's threads
        if (std::memcmp(dst.data(), src.data(), src.size() * sizeof(double)) != 0) { *rc = 1; return 1; }
        const size_t n = (size_t)1 << 17;
        pool().interleave2(il.data(), src.data(), src.data() + n, n);
        for (size_t i = 0; i < n; ++i) if (il[2 * i] != src[i] || il[2 * i + 1] != src[n + i]) { *rc = 1; return 1; }
    }
    return 0;
}

static int test_pool_two_callers() {
    int rc1 = 0, rc2 = 0;
    std::thread t1(pool_worker, 1, &rc1), t2(pool_worker, 2, &rc2);
    t1.join(); t2.join();
    CHECK(rc1 == 0 && rc2 == 0);
    return 0;
}

static int engine_round_trip(int device, int* rc) {
    Engine* E = nullptr;
    if (engine(device, &E) != FEP_OK) { *rc = 1; return 1; }
    const size_t n = ((size_t)20 << 20) / sizeof(double) + 12345;       // 2.5 ring slots: exercises the wrap-around
    std::vector<double> in(n), out(n, -1.0);
    for (size_t i = 0; i < n; ++i) in[i] = (double)i + device;
    for (int rep = 0; rep < 3; ++rep) {
        EngineCall call(E);
        void* dbuf = nullptr;
        if (E->buffer(0, n * sizeof(double), &dbuf) != FEP_OK) { *rc = 1; return 1; }
        if (E->h2d(dbuf, in.data(), n * sizeof(double)) != FEP_OK) { *rc = 1; return 1; }
        if (E->d2h(out.data(), dbuf, n * sizeof(double)) != FEP_OK) { *rc = 1; return 1; }
        if (call.finish() != FEP_OK) { *rc = 1; return 1; }
        if (std::memcmp(in.data(), out.data(), n * sizeof(double)) != 0) { *rc = 2; return 1; }
        std::fill(out.begin(), out.end(), -1.0);
    }
    return 0;
}

// what a *_host entry point does, with the copy that follows the first device -> host chunk made to fail
static int failing_call(Engine* E, double* out, size_t n, void* dbuf) {
    EngineCall call(E);
    FEP_TRY(E->d2h(out, dbuf, n * sizeof(double)));      // parks chunks in the ring: pend[i].dst points into `out`
    return call.finish();
}

static int test_engine_error_path() {
    Engine* E = nullptr;
    CHECK(engine(7, &E) == FEP_OK);
    const size_t n = ((size_t)20 << 20) / sizeof(double);
    void* dbuf = nullptr;
    CHECK(E->buffer(0, n * sizeof(double), &dbuf) == FEP_OK);
    std::memset(dbuf, 0, n * sizeof(double));
    {
        std::vector<double>* out = new std::vector<double>(n);
        g_fail_copy_after = g_copies.load() + 1;         // first chunk lands in a slot, the second copy fails
        CHECK(failing_call(E, out->data(), n, dbuf) == FEP_EHIP);
        g_fail_copy_after = -1;
        delete out;                                       // the caller frees its output array on the error
    }
    for (int i = 0; i < Engine::kSlots; ++i) CHECK(E->pend[i].dst == nullptr);
    CHECK(E->next == 0);
    int rc = 0;                                           // the next call on the same engine: ASan reports any write into the freed array
    std::vector<double> in(n, 3.0), out2(n, 0.0);
    {
        EngineCall call(E);
        CHECK(E->h2d(dbuf, in.data(), n * sizeof(double)) == FEP_OK);
        CHECK(E->d2h(out2.data(), dbuf, n * sizeof(double)) == FEP_OK);
        CHECK(call.finish() == FEP_OK);
    }
    for (size_t i = 0; i < n; ++i) if (out2[i] != 3.0) rc = 1;
    CHECK(rc == 0);
    return 0;
}

static int test_two_engines() {
    int rc1 = 0, rc2 = 0;
    std::thread t1(engine_round_trip, 0, &rc1), t2(engine_round_trip, 1, &rc2);
    t1.join(); t2.join();
    CHECK(rc1 == 0 && rc2 == 0);
    return 0;
}

int main() {
    if (test_pinned()) return 1;
    if (test_pool_two_callers()) return 1;
    if (test_engine_error_path()) return 1;
    if (test_two_engines()) return 1;
    std::printf("staging ok\n");
    return 0;
}
