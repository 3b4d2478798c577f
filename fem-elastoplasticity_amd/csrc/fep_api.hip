// C-ABI implementation of include/fep.h (libfep_hip.so).  Host side: context, symbolic
// phase (node graph -> CSR pattern + gather lists), launches.  Device side: fep_kernels.hip.h.
#include "fep_common.h"
#include "fep_host.h"
#include "fep_kernels.hip.h"
#include "fep_staging.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

using namespace fep;

// workgroup shape of the 15-node element's patch form: 16 elements on 256 threads.  The same 16 elements on 512 threads with
// two lanes per (element, node) pair in phase 2 (ElemCfg: JS = 2; 116 instead of 240 VGPRs, twice the waves per CU) measured
// 3-4 % slower in one session (0.931 against 0.899-0.900 ms, profiles/r04_ablation.md): kept in the ablation build
#ifndef FEP_P4_TPB
#define FEP_P4_TPB 256
#define FEP_P4_JS 1
#endif

thread_local int fep_g_last_hip = 0;
#define g_last_hip fep_g_last_hip

// The switches of the PRODUCT library (read once): everything else goes through fep_tune() and exists in -DFEP_ABLATION
// builds only (fep_common.h).
//   FEP_ROUTE=coo | patch   the element route in its COO form (every K_e block through HBM, csr_reduce_kernel) / its patch
//                           form for EVERY element type, P1 included: the independent cross-checks of the parity tests
//                           (unset: P1 takes its node route, the other types the patch form)
//   FEP_VALIDATE_PLAN=1     replay the gather plan against the symbolic phase at context creation (tests)
//   FEP_VERBOSE=1           plan statistics on stderr
static const char* env_once(const char* name) { const char* v = std::getenv(name); return (v && *v) ? v : nullptr; }
static bool route_is(const char* v) { static const char* r = env_once("FEP_ROUTE"); return r && std::strcmp(r, v) == 0; }
static bool validate_plans() { static const bool on = env_once("FEP_VALIDATE_PLAN") != nullptr; return on; }
static bool verbose_on() { static const bool on = env_once("FEP_VERBOSE") != nullptr; return on; }

struct fep_ctx {
    int device = 0;
    int elem_type = 0;
    int n_p = 0, n_q = 0;
    int64_t n_e = 0, n_n = 0, n_int = 0, n_dof = 0, nnz = 0, n_blk = 0, n_contrib = 0;
    bool have_materials = false;
    uint2* pkc = nullptr;                               // COO route: packed block descriptors of csr_reduce_pk_kernel (NULL: fields too wide)
    int csr_gathers = 4;                                // gathers in flight per lane of csr_reduce_kernel (FEP_CSR_GATHERS=2|4|6|8;
                                                        // measured P2 / Q2 / P4 reduce kernel: 2: 0.406 / 0.376 / 0.654 ms, 4: 0.392 / 0.347 / 0.651,
                                                        // 6: 0.382 / 0.371 / 0.650, 8 (5 waves per SIMD): 0.432 / 0.425 / 0.733)
    bool kc_aos = false;                                // K_e half-blocks: all blocks of an element adjacent (AoS) or block-major (SoA)
    // patch route of the element kernel (default; FEP_ROUTE=coo keeps the K_e round trip): fep_host.h, PatchPlan
    bool patch = false;
    int patch_eb = 0, patch_dbg = 0, patch_tpb = 256, patch_js = 1;   // workgroup shape of the patch form (ElemCfg)
    int lds_pad = 0;                                    // FEP_ELEM_LDS_PAD: extra LDS bytes per workgroup of element_kernel (occupancy experiments)
    int64_t n_open = 0, n_fopen = 0;
    int32_t *pt_desc = nullptr, *pt_plist = nullptr, *pt_pel = nullptr, *pt_pnodes = nullptr;
    uint2 *pt_items = nullptr, *pt_fitems = nullptr;
    uint16_t *pt_codes = nullptr, *pt_fcodes = nullptr;
    uint4 *pt_fix = nullptr, *pt_ffix = nullptr;
    uint2* pt_fixT = nullptr;
    int64_t n_patch = 0;
#ifdef FEP_ABLATION
    unsigned long long* phase_clk = nullptr;            // FEP_PHASE_CLK=1: phase stamps of the last element_kernel launch (reported at destroy)
#endif
    double *Pc = nullptr, *Pf = nullptr;                // partial blocks / forces of the open items (scratch, rewritten by every step)
    bool elem_geo = true;                               // element_kernel: geometry from coordinates instead of the dphi arrays
    MatU matu{};                                        // homogeneous-material fast path (arrays not read)
    // device, static
    int32_t* elem = nullptr;
    double *coords = nullptr, *dh1 = nullptr, *dh2 = nullptr, *wf = nullptr;
    double *dphi1 = nullptr, *dphi2 = nullptr, *weight = nullptr, *det = nullptr;
    double *shear = nullptr, *bulk = nullptr, *eta = nullptr, *c = nullptr;
    int32_t *segptr = nullptr, *perm = nullptr, *iptr = nullptr, *ilist = nullptr;
    uint32_t* meta = nullptr;
    // P1 fast path (node-centric gather assembly): 64-byte geometry records, re-encoded gather list
    bool p1_node = false, p1_lds = false;
    double *geo = nullptr, *xy = nullptr;               // xy: interleaved (x,y) per node
    P1Tab p1tab{};
    int32_t *perm2 = nullptr, *ncol = nullptr;
    int32_t *wg_eptr = nullptr, *wg_elist = nullptr, *wg_rng = nullptr;   // wg_rng: <= 8 (start, cum) runs per tile
    bool p1_rng = false, p1_pk = false;
    uint2* pk = nullptr;                                // packed block descriptors, lanes sorted by segment length
    int4* tdesc = nullptr;                              // P1 tiles: (1 + kSegMax) int4 per tile (fep_host.h, P1Plan)
    int p1_segs = 1;                                    // segments per tile of the plan in use
    int32_t* tstart = nullptr;                          // COO route: first block of every tile of csr_reduce_kernel
    int n_wg_p1 = 0;   // LDS-staged variant: per-workgroup element lists
    // one-kernel step (p1_fused_kernel): per-tile node lists / runs, per staged element its tile-local node indices + owner bit
    bool p1_fused = false, p1_fused_rng = false;
    bool asm_from_nodes = false;                        // FEP_P1_ASM=nodes: assembly kernel with geometry from LDS-staged nodes (measured slower)
    bool p1_dma = false;                                // FEP_P1_DMA=1: node data and gather codes of the one-kernel step by LDS-DMA
    int fused_mode = 1;                                 // FEP_P1_FUSED=off|kf|all: 0 = never, 1 = only when no point output is wanted (default), 2 = always
    int lds_NL = 0;                                     // staged nodes per tile (max)
    int32_t *wg_nlist = nullptr, *wg_nrng = nullptr;
    uint32_t* el_nodes = nullptr;
    unsigned long long* slot_counts = nullptr;          // 256 slots x 16 words, zero between steps
    bool has_orphans = false;                           // nodes that belong to no element (their F entries are zeroed per step)
    uint16_t* perm_l = nullptr;
    int tile = 256;                                     // node-pair blocks per workgroup of the assembly kernel
    bool gn = false;                                    // node route for P2 / Q1 / Q2 (point_kernel + node_lds_kernel)
    int gn_tile = 256;
    size_t gn_lds = 0;
    int lds_L = 0, lds_C = 0;                           // staged elements / gather codes per workgroup (max)
    // device, scratch rewritten by every step
    double *Kc = nullptr, *fe = nullptr;
    double *s_int = nullptr, *ds_int = nullptr;     // used when the caller does not ask for s / ds
    uint2* blk_counts = nullptr;
    int n_count_blocks = 0;
    // persistent device buffers of the *_host entry points (u, ep, e, s, ds, ind_p, k, f, counts, q)
    void* hbuf[12] = {};
    size_t hbuf_bytes[12] = {};
    // host copies of the pattern
    std::vector<int32_t> indptr, indices;
    // in-situ profiling: 4 events per step (before element, after element, after csr, after force)
    bool profiling = false;
    std::vector<hipEvent_t> events;
};

__attribute__((visibility("hidden"))) int fep_set_device(int dev) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { g_last_hip = (int)e; (void)hipGetLastError(); return FEP_ENODEV; }
    if (dev < 0 || dev >= n) return FEP_ENODEV;
    e = hipSetDevice(dev);
    if (e != hipSuccess) { g_last_hip = (int)e; (void)hipGetLastError(); return FEP_ENODEV; }
    return FEP_OK;
}

template <class T> static int dmalloc(T** p, int64_t count) {
    *p = nullptr;
    if (count <= 0) count = 1;
    HIP_TRY(hipMalloc((void**)p, (size_t)count * sizeof(T)));
    return FEP_OK;
}
template <class T> static int upload(T** p, const T* src, int64_t count) {
    FEP_TRY(dmalloc(p, count));
    if (count > 0) HIP_TRY(hipMemcpy(*p, src, (size_t)count * sizeof(T), hipMemcpyHostToDevice));
    return FEP_OK;
}

static inline unsigned grid_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// ---------------------------------------------------------------------------------------
extern "C" int fep_version(void) { return 1; }
extern "C" int fep_build_is_ablation(void) {
#ifdef FEP_ABLATION
    return 1;
#else
    return 0;
#endif
}

extern "C" const char* fep_strerror(int code) {
    switch (code) {
        case FEP_OK: return "ok";
        case FEP_EINVAL: return "invalid argument";
        case FEP_ENODEV: return "no usable HIP device";
        case FEP_ENOMEM: return "out of memory";
        case FEP_EHIP: return "HIP runtime error";
        case FEP_ERANGE: return "index out of range";
        case FEP_ESTATE: return "call order violated";
        default: return "unknown error";
    }
}
extern "C" int fep_last_hip_error(void) { return g_last_hip; }

extern "C" int fep_device_count(int* n_out) {
    if (!n_out) return FEP_EINVAL;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { g_last_hip = (int)e; (void)hipGetLastError(); *n_out = 0; return FEP_ENODEV; }
    *n_out = n;
    return FEP_OK;
}

extern "C" int fep_element_shape(int elem_type, int* n_p, int* n_q) {
    int p, q;
    switch (elem_type) {
        case FEP_P1: p = 3; q = 1; break;
        case FEP_P2: p = 6; q = 7; break;
        case FEP_Q1: p = 4; q = 4; break;
        case FEP_Q2: p = 8; q = 9; break;
        case FEP_P4: p = 15; q = 12; break;
        default: return FEP_EINVAL;
    }
    if (n_p) *n_p = p;
    if (n_q) *n_q = q;
    return FEP_OK;
}

extern "C" int fep_malloc(int device_id, void** ptr_d, int64_t bytes) {
    if (!ptr_d || bytes < 0) return FEP_EINVAL;
    FEP_TRY(fep_set_device(device_id));
    HIP_TRY(hipMalloc(ptr_d, (size_t)(bytes > 0 ? bytes : 1)));
    return FEP_OK;
}
extern "C" int fep_free(int device_id, void* ptr_d) {
    if (!ptr_d) return FEP_OK;
    FEP_TRY(fep_set_device(device_id));
    HIP_TRY(hipFree(ptr_d));
    return FEP_OK;
}
extern "C" int fep_memcpy_h2d(int device_id, void* dst_d, const void* src_h, int64_t bytes) {
    if (bytes < 0 || (bytes > 0 && (!dst_d || !src_h))) return FEP_EINVAL;
    FEP_TRY(fep_set_device(device_id));
    if (bytes) HIP_TRY(hipMemcpy(dst_d, src_h, (size_t)bytes, hipMemcpyHostToDevice));
    return FEP_OK;
}
extern "C" int fep_memcpy_d2h(int device_id, void* dst_h, const void* src_d, int64_t bytes) {
    if (bytes < 0 || (bytes > 0 && (!dst_h || !src_d))) return FEP_EINVAL;
    FEP_TRY(fep_set_device(device_id));
    if (bytes) HIP_TRY(hipMemcpy(dst_h, src_d, (size_t)bytes, hipMemcpyDeviceToHost));
    return FEP_OK;
}
extern "C" int fep_sync(int device_id, void* stream) {
    FEP_TRY(fep_set_device(device_id));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return FEP_OK;
}

// ---------------------------------------------------------------------------------------
// a2 mesh-free
// ---------------------------------------------------------------------------------------
static E0 make_e0(const double* e0_h) {
    E0 z;
    for (int i = 0; i < 4; ++i) z.v[i] = e0_h ? e0_h[i] : 0.0;
    return z;
}

// Per-workgroup branch counters of the mesh-free return map (summed by counts_reduce_kernel): one grow-only buffer per
// (device, stream), so that calls on different streams never share it.  The kernel used to add its two counters to the
// caller's counts with global atomics — one contended pair per workgroup, the pattern that cost the P1 kernels 70 us per
// launch in round 1.  Growing the buffer is an allocation: refused (FEP_ESTATE) while the stream is being captured.
#include <map>
#include <mutex>
static int scratch_alloc_allowed(hipStream_t st);
static int rm_scratch(int device, hipStream_t st, size_t n_blocks, uint2** out) {
    static std::mutex m;
    static auto& bufs = *new std::map<std::pair<int, void*>, std::pair<uint2*, size_t>>();
    std::lock_guard<std::mutex> g(m);
    auto& b = bufs[{device, (void*)st}];
    if (b.second < n_blocks) {
        FEP_TRY(scratch_alloc_allowed(st));
        static auto& retired = *new std::vector<uint2*>();
        if (b.first) { retired.push_back(b.first); b = {nullptr, 0}; }
        const size_t cap = n_blocks + n_blocks / 4 + 64;
        HIP_TRY(hipMalloc((void**)&b.first, cap * sizeof(uint2)));
        b.second = cap;
    }
    *out = b.first;
    return FEP_OK;
}

extern "C" int fep_return_map_dev(int device_id, void* stream, int64_t n_int,
                                  const double* e_d, int64_t e_pt_stride, int64_t e_comp_stride,
                                  const double* e0_h, double* ep_prev_d,
                                  const double* shear_d, const double* bulk_d, const double* eta_d, const double* c_d,
                                  int accept, double* s_d, double* ds_d, uint8_t* ind_p_d, int64_t* counts_d) {
    if (n_int < 0) return FEP_EINVAL;
    if (n_int > 0 && (!e_d || !shear_d || !bulk_d || !eta_d || !c_d)) return FEP_EINVAL;
    FEP_TRY(fep_set_device(device_id));
    hipStream_t st = (hipStream_t)stream;
    if (n_int == 0) { if (counts_d) HIP_TRY(hipMemsetAsync(counts_d, 0, 2 * sizeof(int64_t), st)); return FEP_OK; }
    const unsigned n_blocks = grid_for(n_int, kBlock);
    uint2* blk = nullptr;
    if (counts_d) FEP_TRY(rm_scratch(device_id, st, n_blocks, &blk));
    hipLaunchKernelGGL(return_map_kernel, dim3(n_blocks), dim3(kBlock), 0, st,
                       n_int, e_d, e_pt_stride, e_comp_stride, make_e0(e0_h), ep_prev_d,
                       shear_d, bulk_d, eta_d, c_d, accept, s_d, ds_d, ind_p_d, blk);
    HIP_TRY(hipGetLastError());
    if (counts_d) {
        hipLaunchKernelGGL(counts_reduce_kernel, dim3(1), dim3(1024), 0, st, (int)n_blocks, blk, (unsigned long long*)counts_d);
        HIP_TRY(hipGetLastError());
    }
    return FEP_OK;
}

// persistent device buffer `idx` of a context's host entry points (sizes are fixed by the mesh: allocated once)
static int ctx_buf(fep_ctx* c, int idx, int64_t bytes, void** out) {
    if (bytes <= 0) bytes = 1;
    if (c->hbuf_bytes[idx] < (size_t)bytes) {
        if (c->hbuf[idx]) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->hbuf[idx])); c->hbuf[idx] = nullptr; c->hbuf_bytes[idx] = 0; }
        HIP_TRY(hipMalloc(&c->hbuf[idx], (size_t)bytes));
        c->hbuf_bytes[idx] = (size_t)bytes;
    }
    *out = c->hbuf[idx];
    return FEP_OK;
}

extern "C" int fep_host_alloc(void** ptr_h, int64_t bytes) {
    if (!ptr_h || bytes < 0) return FEP_EINVAL;
    *ptr_h = nullptr;
    try { return fep_stage::pinned().alloc(ptr_h, (size_t)bytes); }
    catch (...) { return FEP_ENOMEM; }
}
extern "C" int fep_host_free(void* ptr_h) {
    try { return fep_stage::pinned().release(ptr_h); }
    catch (...) { return FEP_ENOMEM; }
}
extern "C" int fep_host_trim(void) {
    try { return fep_stage::pinned().trim(); }
    catch (...) { return FEP_ENOMEM; }
}

static int return_map_host_impl(int device_id, int64_t n_int,
                                const double* e_h, int64_t e_pt_stride, int64_t e_comp_stride,
                                const double* e0_h, double* ep_prev_h,
                                const double* shear_h, const double* bulk_h, const double* eta_h, const double* c_h,
                                int accept, double* s_h, double* ds_h, uint8_t* ind_p_h, int64_t* counts_h) {
    if (n_int < 0) return FEP_EINVAL;
    if (n_int > 0 && (!e_h || !shear_h || !bulk_h || !eta_h || !c_h)) return FEP_EINVAL;
    if (counts_h) { counts_h[0] = 0; counts_h[1] = 0; }
    if (n_int == 0) return FEP_OK;
    FEP_TRY(fep_set_device(device_id));
    // the strain is copied as the dense block that contains the strided view
    const int64_t span = (n_int - 1) * e_pt_stride + 2 * e_comp_stride + 1;
    if (e_pt_stride <= 0 || e_comp_stride <= 0 || span < 3 * n_int) return FEP_EINVAL;
    const int64_t nb = n_int * (int64_t)sizeof(double);
    fep_stage::Engine* E = nullptr;
    FEP_TRY(fep_stage::engine(device_id, &E));
    fep_stage::EngineCall call(E);                      // drains the engine on every exit that is not a completed finish()
    void *e = nullptr, *ep = nullptr, *sh, *bu, *et, *cc, *s = nullptr, *ds = nullptr, *ip = nullptr, *cnt;
    FEP_TRY(E->buffer(0, span * (int64_t)sizeof(double), &e));
    if (ep_prev_h) FEP_TRY(E->buffer(1, 4 * nb, &ep));
    FEP_TRY(E->buffer(2, nb, &sh)); FEP_TRY(E->buffer(3, nb, &bu)); FEP_TRY(E->buffer(4, nb, &et)); FEP_TRY(E->buffer(5, nb, &cc));
    if (s_h) FEP_TRY(E->buffer(6, 4 * nb, &s));
    if (ds_h) FEP_TRY(E->buffer(7, 9 * nb, &ds));
    if (ind_p_h) FEP_TRY(E->buffer(8, n_int, &ip));
    FEP_TRY(E->buffer(9, 2 * sizeof(int64_t), &cnt));
    static const bool timing = fep_tune("FEP_TIME_HOST") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    FEP_TRY(E->h2d(e, e_h, (size_t)span * sizeof(double)));
    if (ep_prev_h) FEP_TRY(E->h2d(ep, ep_prev_h, (size_t)(4 * nb)));
    FEP_TRY(E->h2d(sh, shear_h, (size_t)nb)); FEP_TRY(E->h2d(bu, bulk_h, (size_t)nb));
    FEP_TRY(E->h2d(et, eta_h, (size_t)nb)); FEP_TRY(E->h2d(cc, c_h, (size_t)nb));
    if (timing) {
        (void)hipStreamSynchronize(E->stream);
        std::fprintf(stderr, "[fep] return_map_host: inputs on the device after %.3f ms\n",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    FEP_TRY(fep_return_map_dev(device_id, E->stream, n_int, (const double*)e, e_pt_stride, e_comp_stride, e0_h, (double*)ep,
                               (const double*)sh, (const double*)bu, (const double*)et, (const double*)cc, accept, (double*)s,
                               (double*)ds, (uint8_t*)ip, (int64_t*)cnt));
    if (s_h) FEP_TRY(E->d2h(s_h, s, (size_t)(4 * nb)));
    if (ds_h) FEP_TRY(E->d2h(ds_h, ds, (size_t)(9 * nb)));
    if (ind_p_h) FEP_TRY(E->d2h(ind_p_h, ip, (size_t)n_int));
    if (counts_h) FEP_TRY(E->d2h(counts_h, cnt, 2 * sizeof(int64_t)));
    if (accept && ep_prev_h) FEP_TRY(E->d2h(ep_prev_h, ep, (size_t)(4 * nb)));
    const int rf = call.finish();
    if (timing)
        std::fprintf(stderr, "[fep] return_map_host: done after %.3f ms\n",
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return rf;
}

#define FEP_GUARD(call) \
    try { return call; } catch (const std::bad_alloc&) { return FEP_ENOMEM; } catch (...) { return FEP_EINVAL; }

extern "C" int fep_return_map_host(int device_id, int64_t n_int,
                                   const double* e_h, int64_t e_pt_stride, int64_t e_comp_stride,
                                   const double* e0_h, double* ep_prev_h,
                                   const double* shear_h, const double* bulk_h, const double* eta_h, const double* c_h,
                                   int accept, double* s_h, double* ds_h, uint8_t* ind_p_h, int64_t* counts_h) {
    FEP_GUARD(return_map_host_impl(device_id, n_int, e_h, e_pt_stride, e_comp_stride, e0_h, ep_prev_h, shear_h, bulk_h, eta_h,
                                   c_h, accept, s_h, ds_h, ind_p_h, counts_h))
}

using fep_host::Symbolic;
using fep_host::build_symbolic;

static_assert(fep_host::kSegMax == fep::kSegMax, "tile descriptor layout shared by host and kernels");

template <int NP, int NQ>
static int launch_geometry(fep_ctx* c) {
    hipLaunchKernelGGL((geometry_kernel<NP, NQ>), dim3(grid_for(c->n_int, kBlock)), dim3(kBlock), 0, nullptr,
                       c->n_e, c->n_n, c->elem, c->coords, c->dh1, c->dh2, c->wf, c->dphi1, c->dphi2, c->weight, c->det,
                       c->geo);
    HIP_TRY(hipGetLastError());
    return FEP_OK;
}

#define DISPATCH_ELEM(type, CALL)                    \
    switch (type) {                                  \
        case FEP_P1: { CALL(3, 1); } break;          \
        case FEP_P2: { CALL(6, 7); } break;          \
        case FEP_Q1: { CALL(4, 4); } break;          \
        case FEP_Q2: { CALL(8, 9); } break;          \
        case FEP_P4: { CALL(15, 12); } break;        \
        default: return FEP_EINVAL;                  \
    }

extern "C" int fep_ctx_destroy(fep_ctx* c) {
    if (!c) return FEP_OK;
    if (fep_set_device(c->device) == FEP_OK) {
#ifdef FEP_ABLATION
        if (c->phase_clk && c->n_patch > 0) {            // mean shader clocks per phase of the LAST element_kernel launch
            std::vector<unsigned long long> h((size_t)(12 * c->n_patch));
            if (hipDeviceSynchronize() == hipSuccess &&
                hipMemcpy(h.data(), c->phase_clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                double acc[6] = {0, 0, 0, 0, 0, 0}, cyc = 0.0, real = 0.0, st_ids = 0.0, st_data = 0.0, st_bar = 0.0;
                unsigned long long lo = ~0ull, hi = 0;
                for (int64_t p = 0; p < c->n_patch; ++p) {
                    const unsigned long long* s = h.data() + 12 * p;
                    for (int k = 0; k < 6; ++k) acc[k] += (double)(s[k + 1] - s[k]);
                    lo = std::min(lo, s[0]); hi = std::max(hi, s[6]);
                    cyc += (double)(s[6] - s[0]); real += (double)(s[9] - s[8]);
                    st_ids += (double)(s[10] - s[0]); st_data += (double)(s[11] - s[10]); st_bar += (double)(s[1] - s[11]);
                }
                std::fprintf(stderr, "[fep] staging of thread 0's wave: ids %.0f, node data %.0f, LDS writes + barrier (the other waves) %.0f\n",
                             st_ids / c->n_patch, st_data / c->n_patch, st_bar / c->n_patch);
                std::fprintf(stderr, "[fep] phase clocks, mean per workgroup of %lld (stage, phase1, phase2, wait, image, phase3): "
                             "%.0f %.0f %.0f %.0f %.0f %.0f  sum %.0f; first start -> last end %llu; shader clock %.0f MHz (s_memtime / s_memrealtime x 100 MHz over the workgroups)\n", (long long)c->n_patch,
                             acc[0] / c->n_patch, acc[1] / c->n_patch, acc[2] / c->n_patch, acc[3] / c->n_patch, acc[4] / c->n_patch,
                             acc[5] / c->n_patch, (acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5]) / c->n_patch, hi - lo,
                             real > 0 ? cyc / real * 100.0 : 0.0);
            }
            (void)hipFree(c->phase_clk);
        }
#endif
        void* ptrs[] = {c->elem, c->coords, c->dh1, c->dh2, c->wf, c->dphi1, c->dphi2, c->weight, c->det, c->shear, c->bulk,
                        c->eta, c->c, c->segptr, c->perm, c->iptr, c->ilist, c->meta, c->Kc, c->fe, c->geo, c->perm2,
                        c->ncol, c->s_int, c->ds_int, c->blk_counts, c->wg_eptr, c->wg_elist, c->wg_rng, c->perm_l, c->xy, c->pk, c->tdesc, c->tstart,
                        c->wg_nlist, c->wg_nrng, c->el_nodes, c->slot_counts, c->pkc, c->pt_desc, c->pt_plist, c->pt_items,
                        c->pt_fitems, c->pt_codes, c->pt_fcodes, c->pt_fix, c->pt_ffix, c->Pc, c->Pf, c->pt_pel, c->pt_pnodes, c->pt_fixT};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        for (hipEvent_t ev : c->events) (void)hipEventDestroy(ev);
        for (void* p : c->hbuf)
            if (p) (void)hipFree(p);
    }
    delete c;
    return FEP_OK;
}

static int ctx_create_impl(fep_ctx*& c, fep_ctx** ctx_out, int device_id, int elem_type, int64_t n_e, int64_t n_n,
                           const int32_t* elements_h, const double* coords_h,
                           const double* dhatp1_h, const double* dhatp2_h, const double* wf_h);

// Nothing may leave an extern "C" entry point by exception: the host symbolic phase allocates multi-GB std::vectors
// (std::bad_alloc at BASELINE configs[4] sizes on a small host) -> FEP_ENOMEM instead of std::terminate.
extern "C" int fep_ctx_create(fep_ctx** ctx_out, int device_id, int elem_type, int64_t n_e, int64_t n_n,
                              const int32_t* elements_h, const double* coords_h,
                              const double* dhatp1_h, const double* dhatp2_h, const double* wf_h) {
    if (!ctx_out) return FEP_EINVAL;
    *ctx_out = nullptr;
    fep_ctx* c = nullptr;
    int r;
    try {
        r = ctx_create_impl(c, ctx_out, device_id, elem_type, n_e, n_n, elements_h, coords_h, dhatp1_h, dhatp2_h, wf_h);
    } catch (const std::bad_alloc&) {
        r = FEP_ENOMEM;
    } catch (...) {
        r = FEP_EINVAL;
    }
    if (r != FEP_OK) { if (c) fep_ctx_destroy(c); *ctx_out = nullptr; }
    return r;
}

// The element kernel's per-type choices (measured; the notes are at their use in ctx_create_impl)
static constexpr bool elem_geo_default(int elem_type, bool patch_form) {
    return elem_type == FEP_Q1 || elem_type == FEP_Q2 || ((elem_type == FEP_P2 || elem_type == FEP_P4) && patch_form);
}
static void patch_shape_default(int elem_type, bool patch_form, int& tpb, int& js) {
    tpb = 256; js = 1;
    if (patch_form && elem_type == FEP_P2) tpb = 512;
    if (patch_form && elem_type == FEP_P4) { tpb = FEP_P4_TPB; js = FEP_P4_JS; }
}

static int ctx_create_impl(fep_ctx*& c, fep_ctx** ctx_out, int device_id, int elem_type, int64_t n_e, int64_t n_n,
                           const int32_t* elements_h, const double* coords_h,
                           const double* dhatp1_h, const double* dhatp2_h, const double* wf_h) {
    int n_p = 0, n_q = 0;
    FEP_TRY(fep_element_shape(elem_type, &n_p, &n_q));
    if (n_e <= 0 || n_n <= 0 || !elements_h || !coords_h || !dhatp1_h || !dhatp2_h || !wf_h) return FEP_EINVAL;
    if (n_e * (int64_t)n_q >= INT32_MAX / 16) return FEP_ERANGE;
    Symbolic S;
    FEP_TRY(build_symbolic(n_p, n_e, n_n, elements_h, S));
    FEP_TRY(fep_set_device(device_id));
    c = new (std::nothrow) fep_ctx();
    if (!c) return FEP_ENOMEM;
    c->device = device_id; c->elem_type = elem_type; c->n_p = n_p; c->n_q = n_q;
    c->n_e = n_e; c->n_n = n_n; c->n_int = n_e * n_q; c->n_dof = 2 * n_n;
    c->n_blk = (int64_t)S.ncol.size(); c->nnz = 4 * c->n_blk; c->n_contrib = (int64_t)S.perm.size();
    for (int64_t n = 0; n < n_n && !c->has_orphans; ++n) c->has_orphans = S.iptr[n + 1] == S.iptr[n];
    // host CSR pattern on DOFs
    c->indptr.resize(c->n_dof + 1);
    c->indices.resize(c->nnz);
    for (int64_t n = 0; n < n_n; ++n) {
        const int64_t d = S.nptr[n + 1] - S.nptr[n];
        for (int i = 0; i < 2; ++i) {
            const int64_t start = 4 * (int64_t)S.nptr[n] + i * 2 * d;
            c->indptr[2 * n + i] = (int32_t)start;
            for (int64_t s = 0; s < d; ++s) {
                c->indices[start + 2 * s] = 2 * S.ncol[S.nptr[n] + s];
                c->indices[start + 2 * s + 1] = 2 * S.ncol[S.nptr[n] + s] + 1;
            }
        }
    }
    c->indptr[c->n_dof] = (int32_t)c->nnz;
    int r = FEP_OK;
#define CK(x) do { if (r == FEP_OK) r = (x); } while (0)
    CK(upload(&c->elem, elements_h, (int64_t)n_p * n_e));
    CK(upload(&c->coords, coords_h, 2 * n_n));
    CK(upload(&c->dh1, dhatp1_h, (int64_t)n_p * n_q));
    CK(upload(&c->dh2, dhatp2_h, (int64_t)n_p * n_q));
    CK(upload(&c->wf, wf_h, (int64_t)n_q));
    CK(dmalloc(&c->dphi1, (int64_t)n_p * c->n_int));
    CK(dmalloc(&c->dphi2, (int64_t)n_p * c->n_int));
    CK(dmalloc(&c->weight, c->n_int));
    CK(dmalloc(&c->det, c->n_int));
    CK(dmalloc(&c->shear, c->n_int)); CK(dmalloc(&c->bulk, c->n_int)); CK(dmalloc(&c->eta, c->n_int)); CK(dmalloc(&c->c, c->n_int));
    // COO route: tiles of whole nodes with at most kBlock node-pair blocks (csr_reduce_kernel's work units)
    std::vector<int32_t> tstart_all;
    CK(fep_host::row_tiles(S, n_n, kBlock, tstart_all));
    // P1 runs the node-centric fast path unless FEP_ROUTE asks for the element route (coo | patch)
    c->p1_node = elem_type == FEP_P1 && !route_is("coo") && !route_is("patch");
    if (c->p1_node) {
        // gather plan of the node route (host, fep_host.h): tiles, staged element / node lists, codes, descriptors
        fep_host::P1Options opt;
#ifdef FEP_ABLATION
        // FEP_P1_TILE=128: tiles of 128 blocks on 128 threads (16 instead of 8 resident per CU; VERDICT r3 item 5's structural
        // attempt: the K,F-only step measured 0.0706-0.0715 against 0.0629-0.0631 ms, profiles/r04_ablation.md)
        if (const char* tl = fep_tune("FEP_P1_TILE")) c->tile = std::atoi(tl) == 128 ? 128 : 256;
#endif
        opt.tile = c->tile;
#ifdef FEP_ABLATION
        {   // FEP_P1_PATH=node_direct | node_list | node_unpacked | node2k: plans without one of the default's ingredients
            const char* pth = fep_tune("FEP_P1_PATH");
            auto is = [&](const char* v) { return pth && std::strcmp(pth, v) == 0; };
            opt.allow_lds = !is("node_direct");
            opt.allow_rng = !is("node_list");
            opt.allow_pk = !is("node_unpacked");
            opt.allow_fused = !is("node2k");
            if (const char* sg = fep_tune("FEP_P1_SEGS")) opt.max_segs = std::max(1, std::min(std::atoi(sg), fep_host::kSegMax));
        }
#endif
        fep_host::P1Plan P;
        CK(fep_host::build_p1_plan(S, n_e, n_n, elements_h, opt, P));
        if (r == FEP_OK && validate_plans()) {
            const int bad = fep_host::validate_p1_plan(P, S, n_e, n_n, elements_h);
            if (bad) { std::fprintf(stderr, "[fep] P1 plan fails check %d\n", bad); r = FEP_EINVAL; }
        }
#ifndef FEP_ABLATION
        // A mesh whose tiles do not fit the LDS-staged form or whose descriptor fields do not pack (node degree >= 2^12, ...)
        // takes the element route like every other element type: the product keeps ONE node route.
        if (r == FEP_OK && !(P.lds && P.pk)) {
            c->p1_node = false;
            if (verbose_on()) std::fprintf(stderr, "[fep] P1 plan without the LDS-staged packed form (lds %d pk %d): element route\n", (int)P.lds, (int)P.pk);
        }
#endif
        if (r == FEP_OK && c->p1_node) {
            c->n_wg_p1 = (int)P.n_wg; c->p1_segs = P.n_segs;
            c->p1_lds = P.lds; c->lds_L = P.L; c->lds_C = P.C; c->p1_rng = P.rng; c->p1_pk = P.pk;
            c->p1_fused = P.fused; c->p1_fused_rng = P.fused_rng; c->lds_NL = P.NL;
#ifdef FEP_ABLATION
            if (const char* am = fep_tune("FEP_P1_ASM")) c->asm_from_nodes = std::strcmp(am, "nodes") == 0;
            if (const char* dm = fep_tune("FEP_P1_DMA")) c->p1_dma = std::strcmp(dm, "0") != 0;
            if (const char* fm = fep_tune("FEP_P1_FUSED"))
                c->fused_mode = std::strcmp(fm, "off") == 0 ? 0 : std::strcmp(fm, "kf") == 0 ? 1 : 2;
#endif
            CK(upload(&c->perm2, P.perm2.data(), (int64_t)P.perm2.size()));
            CK(upload(&c->ncol, S.ncol.data(), (int64_t)S.ncol.size()));
            CK(upload(&c->tdesc, (const int4*)P.tdesc.data(), (int64_t)P.tdesc.size() / 4));
            if (P.lds) {
                CK(upload(&c->wg_elist, P.elist_pad.data(), (int64_t)P.elist_pad.size()));
                CK(upload(&c->perm_l, P.codes_pad.data(), (int64_t)P.codes_pad.size()));
                if (P.rng) CK(upload(&c->wg_rng, P.rng_tab.data(), (int64_t)P.rng_tab.size()));
                if (P.pk) CK(upload(&c->pk, (const uint2*)P.pkv.data(), (int64_t)P.pkv.size()));
            }
            if (P.fused) {
                CK(upload(&c->el_nodes, P.elnodes.data(), (int64_t)P.elnodes.size()));
                if (P.fused_rng) { CK(upload(&c->wg_nrng, P.nrng_tab.data(), (int64_t)P.nrng_tab.size())); }
                else { CK(upload(&c->wg_nlist, P.nlist_pad.data(), (int64_t)P.nlist_pad.size())); }
                CK(dmalloc(&c->slot_counts, 256 * 16));
                if (r == FEP_OK && hipMemset(c->slot_counts, 0, 256 * 16 * sizeof(unsigned long long)) != hipSuccess) r = FEP_EHIP;
            }
            if (verbose_on())
                std::fprintf(stderr, "[fep] P1 plan: %lld tiles of <= %d blocks in <= %d segment(s), staged elements %lld "
                             "(%.2f per element, <= %d per tile), staged nodes <= %d, codes <= %d; lds %d rng %d pk %d fused %d/%d\n",
                             (long long)P.n_wg, P.tile, P.n_segs, (long long)P.staged_total, (double)P.staged_total / (double)n_e,
                             P.L, P.NL, P.C, (int)P.lds, (int)P.rng, (int)P.pk, (int)P.fused, (int)P.fused_rng);
        }
        if (c->p1_node) {
            CK(dmalloc(&c->geo, 6 * n_e));
            for (int a = 0; a < 3; ++a) { c->p1tab.h1[a] = dhatp1_h[a]; c->p1tab.h2[a] = dhatp2_h[a]; }
            c->p1tab.wf = wf_h[0];
            c->n_count_blocks = (int)grid_for(n_e, kBlock);
        }
    }
#ifdef FEP_ABLATION
    else {
        // FEP_GEN_PATH=node: node route for P2 / Q1 / Q2 (point_kernel + node_lds_kernel; measured on MI355X, 1 M elements:
        // Q1 0.276 vs 0.286 ms, P2 1.80 vs 1.30 ms, Q2 0.41 vs 0.34 ms per step — its LDS gather costs n_q times the P1 one)
        const char* gp = fep_tune("FEP_GEN_PATH");
        const bool want_gn = (elem_type == FEP_P2 || elem_type == FEP_Q1 || elem_type == FEP_Q2) &&
                             gp && std::strcmp(gp, "node") == 0 && r == FEP_OK;
        if (want_gn) {
            fep_host::GnPlan G;
            fep_host::build_gn_plan(S, n_p, n_q, n_e, G);
            if (G.ok) {
                c->gn = true; c->gn_tile = G.tile; c->lds_L = G.L; c->lds_C = G.C; c->gn_lds = G.lds;
                CK(upload(&c->wg_elist, G.elist_pad.data(), (int64_t)G.elist_pad.size()));
                CK(upload(&c->perm_l, G.codes_pad.data(), (int64_t)G.codes_pad.size()));
                CK(upload(&c->ncol, S.ncol.data(), (int64_t)S.ncol.size()));
                c->n_count_blocks = (int)grid_for(c->n_int, kBlock);
            }
        }
    }
#endif
    // element route: the patch form (no K_e round trip through HBM; default) or the COO form (FEP_ROUTE=coo).
    // Same session, 0.25-4 M elements (profiles/r03_ablation.md, r03_elem_bench.log): P2 -17 % (K,F-only -19 %, BASELINE
    // configs[4] -14 %), Q2 -14 %, Q1 -13 %, P4 -21 % against the COO form
    const bool coo_form = route_is("coo");
    const bool want_patch = !c->p1_node && !c->gn && !coo_form;
    {   // interleaved (x, y) per node: the kernels recompute dphi / weight from the coordinates
        std::vector<double> xy(2 * (size_t)n_n);
        for (int64_t n = 0; n < n_n; ++n) { xy[2 * n] = coords_h[n]; xy[2 * n + 1] = coords_h[n_n + n]; }
        CK(upload(&c->xy, xy.data(), (int64_t)xy.size()));
        // geometry of the element kernel from the coordinates (GEO) or from the streamed dphi arrays, fixed per type and form
        // (elem_geo_default): measured on MI355X (1 M elements, element kernel alone): recomputed geometry wins for Q1 (0.204 vs
        // 0.218 ms) and Q2 (0.344 vs 0.391 ms with 24 instead of 28 elements per workgroup, so that three workgroups still fit a
        // CU) and loses for P2 in the COO form (0.50 vs 0.47 ms at 28 / 32 elements per workgroup); round 3, patch form:
        // recomputed geometry wins for P2 as well (same session, 1 M elements: 0.754 vs 0.783 ms on a fast box, 0.888 vs 0.936
        // on a slow one); round 4, final patch kernel, one session, two passes each: P4 0.769-0.771 against 0.842-0.869 ms (its 30
        // gradient loads per point were a third of its workgroup's life), Q2 0.604 / 0.709, P2 0.632-0.638 / 0.700-0.710, Q1
        // 0.163-0.164 / 0.167-0.174: recomputed for every type in the patch form
        c->elem_geo = elem_geo_default(elem_type, want_patch);
#ifdef FEP_ABLATION
        if (const char* ge = fep_tune("FEP_ELEM_GEO")) c->elem_geo = std::strcmp(ge, "0") != 0;
        if (const char* lp = fep_tune("FEP_ELEM_LDS_PAD")) c->lds_pad = std::max(0, std::atoi(lp));
#endif
    }
    // node -> incident (element, local node) lists: force gather of the COO route and fep_transform_*
    CK(upload(&c->iptr, S.iptr.data(), (int64_t)S.iptr.size()));
    CK(upload(&c->ilist, S.ilist.data(), (int64_t)S.ilist.size()));
    int elem_eb = 1;
    // Workgroup shape of the patch form (threads; lanes per (element, node) pair in phase 2: ElemCfg's JS), fixed per type:
    //   P2  512 / 1: 56 elements on 512 threads (two workgroups of eight waves per CU instead of four of four: the same waves,
    //       2.0 instead of 3.2 partials per element with four runs, fix-up kernel -35 %, element kernel +3-12 %): three boxes,
    //       same session each: 1 M elements 0.653 / 0.681 / 0.724 against 0.700 / 0.713 / 0.716 ms, BASELINE configs[4] 2.61 /
    //       2.89 / 2.89 against 2.86 / 3.09 / 3.00, K,F-only 0.613 against 0.653
    //   Q2  256 / 1 (40 elements on 512 threads, 3.5 instead of 5.8 partials per element: 0.76-0.80 against 0.62 ms; 24 elements
    //       on 384 threads with JS = 2: 0.82-0.83 against 0.62, profiles/r04_ablation.md)
    patch_shape_default(elem_type, want_patch, c->patch_tpb, c->patch_js);
    bool shape_forced = false;
#ifdef FEP_ABLATION
    if (const char* tpb_env = want_patch ? fep_tune("FEP_PATCH_TPB") : nullptr) {
        c->patch_tpb = std::atoi(tpb_env);
        const char* js_env = fep_tune("FEP_PATCH_JS");
        c->patch_js = js_env ? std::atoi(js_env) : 1;
        shape_forced = true;
    }
#endif
    const auto elements_per_workgroup = [&]() -> int {
        const bool g = c->elem_geo;
        const int tpb = c->patch_tpb, js = c->patch_js;
#define EBOF(NP, NQ, TPB, JS) if (tpb == TPB && js == JS) return g ? ElemCfg<NP, NQ, true, TPB, JS>::EB : ElemCfg<NP, NQ, false, TPB, JS>::EB
        switch (elem_type) {
            case FEP_P1: EBOF(3, 1, 256, 1); break;
            case FEP_P2: EBOF(6, 7, 256, 1); EBOF(6, 7, 512, 1); break;
            case FEP_Q1: EBOF(4, 4, 256, 1); break;
            case FEP_Q2: EBOF(8, 9, 256, 1);
#ifdef FEP_ABLATION
                EBOF(8, 9, 384, 2); EBOF(8, 9, 512, 1);
#endif
                break;
            case FEP_P4: EBOF(15, 12, 256, 1);
#ifdef FEP_ABLATION
                EBOF(15, 12, 512, 2);
#endif
                break;
        }
#undef EBOF
        return 0;                                        // no such instantiation
    };
    elem_eb = elements_per_workgroup();
    if (elem_eb == 0) return FEP_EINVAL;
    if (!c->p1_node && !c->gn && r == FEP_OK) {
        if (want_patch) {
            fep_host::PatchPlan P;
            fep_host::PatchOptions popt;
#ifdef FEP_ABLATION
            if (const char* po = fep_tune("FEP_PATCH_ORDER"))
                popt.order = std::strcmp(po, "consecutive") == 0 ? 0 : std::strcmp(po, "hilbert") == 0 ? 1 : 2;
#endif
            // Q1 (64 elements per patch): 4 runs of 16 measured 6 % faster than 2 of 32; P2 with 56 elements: 4 runs of 14
            // (1.96 partials per element) 1-3 % faster than 2 of 28 (2.86); with 28 elements 2 runs of 14 (4 of 7: +7 %)
            popt.runs = (elem_type == FEP_Q1 || c->patch_tpb == 512) ? 4 : 2;
            // (runs starting on 128-byte boundaries of the point arrays — 16 / gcd(16, n_q) elements — measured no faster and
            // cost a third more partials: the aligned starts leave short leftovers)
#ifdef FEP_ABLATION
            if (const char* pr = fep_tune("FEP_PATCH_RUNS")) popt.runs = std::max(1, std::atoi(pr));
            if (const char* pa = fep_tune("FEP_PATCH_ALIGN")) popt.align = std::max(1, std::atoi(pa));
#endif
            CK(fep_host::build_patch_plan(S, n_p, n_e, n_n, elements_h, coords_h, elem_eb, popt, P));
            if (r == FEP_OK && !P.ok && c->patch_tpb == 512 && c->patch_js == 1 && !shape_forced) {     // the big patches cannot be built: the small ones
                c->patch_tpb = 256;                                            // (a forced size is never changed silently)
                elem_eb = elements_per_workgroup();
                if (!fep_tune("FEP_PATCH_RUNS")) popt.runs = 2;
                CK(fep_host::build_patch_plan(S, n_p, n_e, n_n, elements_h, coords_h, elem_eb, popt, P));
            }
            if (r == FEP_OK && P.ok && validate_plans()) {
                const int bad = fep_host::validate_patch_plan(P, S, n_p, n_e, n_n, elements_h);
                if (bad) { std::fprintf(stderr, "[fep] patch plan fails check %d\n", bad); r = FEP_EINVAL; }
            }
            if (r == FEP_OK && P.ok) {
                c->patch = true; c->patch_eb = P.eb; c->n_open = P.n_open; c->n_fopen = P.n_fopen; c->n_patch = P.n_patch;
                CK(upload(&c->pt_desc, P.pdesc.data(), (int64_t)P.pdesc.size()));
                CK(upload(&c->pt_pel, P.pel.data(), (int64_t)P.pel.size()));
                CK(upload(&c->pt_pnodes, P.pnodes.data(), (int64_t)P.pnodes.size()));
                P.items.push_back(fep_host::U2{0u, 0u});                         // (one pad entry each: element_kernel reads its tables
                P.fitems.push_back(fep_host::U2{0u, 0u});                        //  at clamped indices, also for a patch without items)
                CK(upload(&c->pt_items, (const uint2*)P.items.data(), (int64_t)P.items.size()));
                CK(upload(&c->pt_codes, P.codes.data(), (int64_t)P.codes.size()));
                CK(upload(&c->pt_fitems, (const uint2*)P.fitems.data(), (int64_t)P.fitems.size()));
                CK(upload(&c->pt_fcodes, P.fcodes.data(), (int64_t)P.fcodes.size()));
                CK(upload(&c->pt_fix, (const uint4*)P.fix.data(), (int64_t)P.fix.size()));
                CK(upload(&c->pt_fixT, (const uint2*)P.fixT.data(), (int64_t)P.fixT.size()));
                CK(upload(&c->pt_ffix, (const uint4*)P.ffix.data(), (int64_t)P.ffix.size()));
                CK(upload(&c->pt_plist, P.plist.data(), (int64_t)P.plist.size()));
#ifdef FEP_ABLATION
                if (fep_tune("FEP_PHASE_CLK")) {
                    CK(dmalloc(&c->phase_clk, 12 * P.n_patch));
                    if (r == FEP_OK) (void)hipMemset(c->phase_clk, 0, (size_t)(12 * P.n_patch) * sizeof(unsigned long long));
                }
#endif
                CK(dmalloc(&c->Pc, 4 * P.n_part));
                CK(dmalloc(&c->Pf, 2 * P.n_fpart));
                if (verbose_on())
                    std::fprintf(stderr, "[fep] patch plan: %lld patches of <= %d elements, %zu items (<= %d per patch), %lld open blocks of %lld, "
                                 "%lld partials (%.2f per element), %lld open nodes\n", (long long)P.n_patch, P.eb, P.items.size(), P.max_items,
                                 (long long)P.n_open, (long long)c->n_blk, (long long)P.n_part, (double)P.n_part / (double)n_e,
                                 (long long)P.n_fopen);
            }
        }
        c->n_count_blocks = c->patch ? (int)c->n_patch : (int)grid_for(n_e, elem_eb);
    }
    if (!c->patch && shape_forced && c->patch_tpb != 256) r = r == FEP_OK ? FEP_ESTATE : r;        // (a forced plan that could not be built)
    if (!c->patch) { c->patch_tpb = 256; c->patch_js = 1; }
    if (!c->patch) {
        CK(upload(&c->segptr, S.segptr.data(), (int64_t)S.segptr.size()));
        CK(upload(&c->meta, S.meta.data(), (int64_t)S.meta.size()));
    }
    if (!c->p1_node && !c->gn && !c->patch) {
        // measured at ~1 M points per type (tools/elem_bench.py): the element-major layout pays for the 15-node element only
        // (P4: step 1.47 -> 1.35 ms; P2 0.96 -> 1.06, Q2 0.82 -> 0.94, Q1 0.22 -> 0.26: their stores lose coalescing)
        c->kc_aos = elem_type == FEP_P4;
#ifdef FEP_ABLATION
        if (const char* kl = fep_tune("FEP_KC_LAYOUT")) c->kc_aos = std::strcmp(kl, "aos") == 0;
        if (const char* cg = fep_tune("FEP_CSR_GATHERS")) c->csr_gathers = std::atoi(cg);
#endif
        if ((int64_t)sym_block_count(n_p) * n_e >= (int64_t)1 << 30) r = FEP_ERANGE;     // 2 * position fits int32
        {   // element_kernel stores half of the symmetric K_e: re-address the contributions (block, transposed)
            std::vector<int32_t> perm_sym(S.perm.size());
            for (size_t i = 0; i < S.perm.size(); ++i) {
                const int64_t v = S.perm[i];
                const int ab = (int)(v / n_e);
                const int64_t e = v - (int64_t)ab * n_e;
                int idx; bool tr;
                sym_block_index(n_p, ab / n_p, ab % n_p, idx, tr, c->kc_aos);
                const int64_t pos = c->kc_aos ? e * sym_block_count(n_p) + idx : (int64_t)idx * n_e + e;
                perm_sym[i] = (int32_t)(2 * pos + (tr ? 1 : 0));
            }
            perm_sym.resize(perm_sym.size() + 4, 0);                    // csr_reduce_pk_kernel reads the addresses four at a time
            CK(upload(&c->perm, perm_sym.data(), (int64_t)perm_sym.size()));
            // one 8-byte descriptor per block, if its fields fit (count < 256, degree and slot < 4096)
            std::vector<uint2> pkc((size_t)c->n_blk);
            bool fits = fep_tune("FEP_CSR_UNPACKED") == nullptr;
            for (int64_t b = 0; b < c->n_blk && fits; ++b) {
                const uint32_t len = (uint32_t)(S.segptr[b + 1] - S.segptr[b]), deg = S.meta[b] >> 16, slot = S.meta[b] & 0x7fffu;
                fits = len < 256 && deg < 4096 && slot < 4096;
                pkc[b] = make_uint2((uint32_t)S.segptr[b], len | (deg << 8) | (slot << 20));
            }
            if (fits) CK(upload(&c->pkc, pkc.data(), (int64_t)pkc.size()));
        }
        CK(upload(&c->tstart, tstart_all.data(), (int64_t)tstart_all.size()));
        c->n_wg_p1 = (int)tstart_all.size() - 1;
        CK(dmalloc(&c->Kc, 4 * (int64_t)sym_block_count(n_p) * n_e));
        CK(dmalloc(&c->fe, 2 * (int64_t)n_p * n_e));
    }
    CK(dmalloc(&c->blk_counts, c->n_count_blocks));
#undef CK
    if (r == FEP_OK) {
#define CALL(NP, NQ) r = launch_geometry<NP, NQ>(c)
        switch (elem_type) {
            case FEP_P1: CALL(3, 1); break;
            case FEP_P2: CALL(6, 7); break;
            case FEP_Q1: CALL(4, 4); break;
            case FEP_Q2: CALL(8, 9); break;
            case FEP_P4: CALL(15, 12); break;
        }
#undef CALL
    }
    if (r == FEP_OK && hipDeviceSynchronize() != hipSuccess) { g_last_hip = (int)hipGetLastError(); r = FEP_EHIP; }
    if (r != FEP_OK) return r;                          // the caller destroys the partial context
    *ctx_out = c;
    return FEP_OK;
}

extern "C" int fep_ctx_sizes(const fep_ctx* c, int64_t sizes[8]) {
    if (!c || !sizes) return FEP_EINVAL;
    sizes[0] = c->n_e; sizes[1] = c->n_n; sizes[2] = c->n_p; sizes[3] = c->n_q;
    sizes[4] = c->n_int; sizes[5] = c->n_dof; sizes[6] = c->nnz; sizes[7] = c->n_blk;
    return FEP_OK;
}

// The kernels of a step, named as rocprofv3 prints them (bench.py labels its roofline with these and checks them against the
// kernel names of the recorded counter passes)
extern "C" int fep_ctx_kernel_names(const fep_ctx* c, int which, char* buf, int64_t cap) {
    if (!c || !buf || cap <= 0 || (which != 0 && which != 1)) return FEP_EINVAL;
    auto b = [](bool v) { return v ? "true" : "false"; };
    char tmp[256];
    if (c->p1_node && c->p1_lds) {
        if (which == 1 && c->p1_fused && c->fused_mode >= 1)
            std::snprintf(tmp, sizeof tmp, "p1_fused_kernel<false, %d, %s, 1, 1, false, %s>", c->tile, b(c->p1_fused_rng), b(c->p1_dma));
        else
            std::snprintf(tmp, sizeof tmp, "p1_point_kernel + p1_node_lds_kernel<%d, %s, 1, %s>", c->tile, b(c->p1_rng), b(c->p1_pk));
    } else if (c->p1_node) {
        std::snprintf(tmp, sizeof tmp, "p1_point_kernel + p1_node_kernel");
    } else if (c->gn) {
        std::snprintf(tmp, sizeof tmp, "point_kernel<%d, %d> + node_lds_kernel<%d, %d, %d>", c->n_p, c->n_q, c->n_p, c->n_q, c->gn_tile);
    } else if (c->patch) {
        std::snprintf(tmp, sizeof tmp, "element_kernel<%d, %d, true, %s, true, %d, %d> + fixup_kernel", c->n_p, c->n_q, b(c->elem_geo),
                      c->patch_tpb, c->patch_js);
    } else {
        std::snprintf(tmp, sizeof tmp, "element_kernel<%d, %d, true, %s, false, 256, 1> + %s", c->n_p, c->n_q, b(c->elem_geo),
                      c->pkc ? "csr_reduce_pk_kernel" : "csr_reduce_kernel<4>");
    }
    std::snprintf(buf, (size_t)cap, "%s", tmp);
    return FEP_OK;
}

extern "C" int fep_ctx_geometry_host(fep_ctx* c, double* dphi1_h, double* dphi2_h, double* weight_h, double* det_h) {
    if (!c) return FEP_EINVAL;
    FEP_TRY(fep_set_device(c->device));
    const size_t nb = (size_t)c->n_int * sizeof(double);
    if (dphi1_h) HIP_TRY(hipMemcpy(dphi1_h, c->dphi1, nb * c->n_p, hipMemcpyDeviceToHost));
    if (dphi2_h) HIP_TRY(hipMemcpy(dphi2_h, c->dphi2, nb * c->n_p, hipMemcpyDeviceToHost));
    if (weight_h) HIP_TRY(hipMemcpy(weight_h, c->weight, nb, hipMemcpyDeviceToHost));
    if (det_h) HIP_TRY(hipMemcpy(det_h, c->det, nb, hipMemcpyDeviceToHost));
    return FEP_OK;
}

extern "C" int fep_ctx_pattern_host(const fep_ctx* c, int32_t* indptr_h, int32_t* indices_h) {
    if (!c) return FEP_EINVAL;
    if (indptr_h) std::memcpy(indptr_h, c->indptr.data(), c->indptr.size() * sizeof(int32_t));
    if (indices_h) std::memcpy(indices_h, c->indices.data(), c->indices.size() * sizeof(int32_t));
    return FEP_OK;
}

extern "C" int fep_ctx_set_materials_host(fep_ctx* c, const double* shear_h, const double* bulk_h,
                                          const double* eta_h, const double* c_h) {
    if (!c || !shear_h || !bulk_h || !eta_h || !c_h) return FEP_EINVAL;
    FEP_TRY(fep_set_device(c->device));
    const size_t nb = (size_t)c->n_int * sizeof(double);
    HIP_TRY(hipMemcpy(c->shear, shear_h, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->bulk, bulk_h, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->eta, eta_h, nb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->c, c_h, nb, hipMemcpyHostToDevice));
    c->have_materials = true;
    // the two-kernel node routes pass ds / s from the point kernel to the assembly kernel through HBM: when the caller
    // wants K / F without them the scratch must exist before the first fep_step_dev (which may run inside a stream
    // capture, where allocating is illegal).  P1 contexts with the one-kernel step only need it for accepting calls that
    // also ask for K / F without ds / s (allocated on first use, outside any capture, by those rare callers).
    if ((c->p1_node && !c->p1_fused) || c->gn) {
        if (!c->ds_int) FEP_TRY(dmalloc(&c->ds_int, 9 * c->n_int));
        if (!c->s_int) FEP_TRY(dmalloc(&c->s_int, 4 * c->n_int));
    }
    // every parameter constant over the mesh (the reference's demos): the kernels skip the four arrays
    bool uni = fep_tune("FEP_NO_UNIFORM") == nullptr;
    for (int64_t k = 1; uni && k < c->n_int; ++k)
        uni = shear_h[k] == shear_h[0] && bulk_h[k] == bulk_h[0] && eta_h[k] == eta_h[0] && c_h[k] == c_h[0];
    c->matu = MatU{shear_h[0], bulk_h[0], eta_h[0], c_h[0], uni ? 1 : 0};
    return FEP_OK;
}

extern "C" int fep_ctx_device_ptr(const fep_ctx* c, int which, void** ptr_d) {
    if (!c || !ptr_d) return FEP_EINVAL;
    switch (which) {
        case 0: *ptr_d = c->shear; break;
        case 1: *ptr_d = c->bulk; break;
        case 2: *ptr_d = c->eta; break;
        case 3: *ptr_d = c->c; break;
        case 4: *ptr_d = c->weight; break;
        case 5: *ptr_d = c->dphi1; break;
        case 6: *ptr_d = c->dphi2; break;
        default: return FEP_EINVAL;
    }
    return FEP_OK;
}

// ---------------------------------------------------------------------------------------
// hot path
// ---------------------------------------------------------------------------------------
static int prof_mark(fep_ctx* c, hipStream_t st) {
    if (!c->profiling) return FEP_OK;
    hipEvent_t ev;
    HIP_TRY(hipEventCreate(&ev));
    c->events.push_back(ev);
    HIP_TRY(hipEventRecord(ev, st));
    return FEP_OK;
}

template <int NP, int NQ, bool FROM_U>
static int launch_element(fep_ctx* c, hipStream_t st, const double* u, E0 e0, double* ep, int accept,
                          double* eout, double* s, double* ds, uint8_t* indp, uint2* blk_counts,
                          double* Kc, double* fe) {
    // patch route: Kc / fe carry the caller's CSR values / force (phase 3 writes them, fixup_kernel the open rest)
    const PatchArgs pa{c->pt_desc, c->pt_pel, c->pt_pnodes, c->pt_items, c->pt_codes, c->pt_fitems, c->pt_fcodes, c->Pc, c->Pf,
                       c->patch ? Kc : nullptr, c->patch ? fe : nullptr
#ifdef FEP_ABLATION
                       , c->phase_clk
#endif
    };
    (void)pa;
#define ELEM_LAUNCH_T(GEO, PATCH, TPB, JS)                                                                               \
    do {                                                                                                                 \
        if (c->patch_eb != 0 && c->patch_eb != ElemCfg<NP, NQ, GEO, TPB, JS>::EB) return FEP_ESTATE;                    \
        hipLaunchKernelGGL((element_kernel<NP, NQ, FROM_U, GEO, PATCH, TPB, JS>),                                       \
                           dim3(PATCH ? (unsigned)c->n_patch : grid_for(c->n_e, ElemCfg<NP, NQ, GEO, TPB, JS>::EB)),       \
                           dim3(TPB), (size_t)c->lds_pad, st, c->n_e,                                                   \
                           c->elem, c->dphi1, c->dphi2, c->weight, c->xy, c->dh1, c->dh2, c->wf, u, e0, ep, c->shear,   \
                           c->bulk, c->eta, c->c, c->matu, accept, eout, s, ds, indp, blk_counts,                       \
                           PATCH ? nullptr : Kc, PATCH ? nullptr : fe, c->kc_aos ? 1 : 0, pa);                          \
    } while (0)
#ifdef FEP_ABLATION
#define ELEM_GEO_BOTH(PATCH, TPB, JS) do { if (c->elem_geo) ELEM_LAUNCH_T(true, PATCH, TPB, JS); else ELEM_LAUNCH_T(false, PATCH, TPB, JS); } while (0)
#else       // the product instantiates ONE geometry form per element type and form of the route
#define ELEM_GEO_BOTH(PATCH, TPB, JS)                                                                                    \
    do {                                                                                                                 \
        constexpr bool G = elem_geo_default(NP == 3 ? FEP_P1 : NP == 6 ? FEP_P2 : NP == 4 ? FEP_Q1 : NP == 8 ? FEP_Q2 : FEP_P4, PATCH); \
        if (c->elem_geo != G) return FEP_ESTATE;                                                                         \
        ELEM_LAUNCH_T(G, PATCH, TPB, JS);                                                                                \
    } while (0)
#endif
#define ELEM_PATCH(TPB, JS) ELEM_GEO_BOTH(true, TPB, JS)
    const int shape = c->patch_tpb * 8 + c->patch_js;
    if (!c->patch) ELEM_GEO_BOTH(false, kBlock, 1);
    else if (shape == 256 * 8 + 1) ELEM_PATCH(256, 1);
    else if (NP == 6 && shape == 512 * 8 + 1) { if constexpr (NP == 6) ELEM_PATCH(512, 1); }
#ifdef FEP_ABLATION
    else if (NP == 15 && shape == 512 * 8 + 2) { if constexpr (NP == 15) ELEM_PATCH(512, 2); }
    else if (NP == 8 && shape == 512 * 8 + 1) { if constexpr (NP == 8) ELEM_PATCH(512, 1); }
    else if (NP == 8 && shape == 384 * 8 + 2) { if constexpr (NP == 8) ELEM_PATCH(384, 2); }
#endif
    else return FEP_ESTATE;
#undef ELEM_PATCH
#undef ELEM_GEO_BOTH
#undef ELEM_LAUNCH_T
    HIP_TRY(hipGetLastError());
    return FEP_OK;
}

static int launch_counts(fep_ctx* c, hipStream_t st, unsigned long long* counts_d) {
    if (!counts_d) return FEP_OK;
    hipLaunchKernelGGL(counts_reduce_kernel, dim3(1), dim3(1024), 0, st, c->n_count_blocks, c->blk_counts, counts_d);
    HIP_TRY(hipGetLastError());
    return FEP_OK;
}

// COO route, numeric phase: K_e blocks -> CSR values, f_e pairs -> nodal force (the force gather rides in the reduce
// kernel's launch: the workgroups past the last tile take 256 nodes each)
static int launch_reduce(fep_ctx* c, hipStream_t st, double* k_data, double* f_out,
                         unsigned long long* counts_d = nullptr, bool* counts_done = nullptr) {
    if (counts_done) *counts_done = false;
    FEP_TRY(prof_mark(c, st));
    bool force_done = false;
    if (k_data) {
        ForceArgs fa{c->n_n, c->iptr, c->ilist, c->fe, f_out};
        const unsigned grid = (unsigned)c->n_wg_p1 + (counts_d ? 1u : 0u) + (f_out ? grid_for(c->n_n, kBlock) : 0u);
#define CSR_REDUCE(G)                                                                                                    \
    hipLaunchKernelGGL(csr_reduce_kernel<G>, dim3(grid), dim3(kBlock), 0, st,                                            \
                       c->n_wg_p1, c->tstart, c->segptr, c->perm, c->meta, c->Kc, k_data, c->n_count_blocks,             \
                       c->blk_counts, counts_d, fa)
        if (c->pkc)
            hipLaunchKernelGGL(csr_reduce_pk_kernel, dim3(grid), dim3(kBlock), 0, st, c->n_wg_p1,
                               c->tstart, c->pkc, c->perm, c->Kc, k_data, c->n_count_blocks, c->blk_counts, counts_d, fa);
        else
#ifdef FEP_ABLATION
        switch (c->csr_gathers) {
            case 2: CSR_REDUCE(2); break;
            case 6: CSR_REDUCE(6); break;
            case 8: CSR_REDUCE(8); break;
            default: CSR_REDUCE(4); break;
        }
#else
        CSR_REDUCE(4);                                   // (descriptor fields too wide for the packed form: four gathers in flight)
#endif
#undef CSR_REDUCE
        HIP_TRY(hipGetLastError());
        if (counts_done) *counts_done = counts_d != nullptr;
        force_done = f_out != nullptr;
    }
    FEP_TRY(prof_mark(c, st));
    if (f_out && !force_done) {
        hipLaunchKernelGGL(force_reduce_kernel, dim3(grid_for(c->n_n, kBlock)), dim3(kBlock), 0, st,
                           c->n_n, c->iptr, c->ilist, c->fe, f_out);
        HIP_TRY(hipGetLastError());
    }
    FEP_TRY(prof_mark(c, st));
    return FEP_OK;
}

// patch route, second kernel: open blocks / open nodes from the patches' partials (+ the branch counters on the side)
static int launch_fixup(fep_ctx* c, hipStream_t st, double* k_data, double* f_out, unsigned long long* counts_d, bool* counts_done) {
    if (counts_done) *counts_done = false;
    FEP_TRY(prof_mark(c, st));
    const unsigned nb_k = k_data ? grid_for(c->n_open, kBlock) : 0u;
    const unsigned nb_f = f_out ? grid_for(c->n_fopen, kBlock) : 0u;
    const unsigned grid = nb_k + nb_f + (counts_d ? 1u : 0u);
    if (grid > 0) {
        hipLaunchKernelGGL(fixup_kernel, dim3(grid), dim3(kBlock), 0, st, (int)nb_k, c->n_open, c->pt_fix, c->pt_fixT, c->n_fopen, c->pt_ffix,
                           c->pt_plist, c->Pc, c->Pf, k_data, f_out, c->n_count_blocks, c->blk_counts, counts_d);
        HIP_TRY(hipGetLastError());
        if (counts_done) *counts_done = counts_d != nullptr;
    }
    FEP_TRY(prof_mark(c, st));
    FEP_TRY(prof_mark(c, st));
    return FEP_OK;
}

// P1 node route, assembly kernel (reads ds / s, writes CSR values and nodal force)
// `counts_d` != NULL: the assembly kernel also sums the point kernel's per-workgroup branch counters
// (*counts_done tells the caller whether that happened)
static int launch_p1_node(fep_ctx* c, hipStream_t st, const double* ds, const double* s, double* k_data, double* f_out,
                          unsigned long long* counts_d, bool* counts_done) {
    FEP_TRY(prof_mark(c, st));
    if (counts_done) *counts_done = false;
    if ((k_data && ds) || (f_out && s)) {
#ifdef FEP_ABLATION
        if (c->p1_fused && c->asm_from_nodes) {
            // opt-in (FEP_P1_ASM=nodes): assembly-only form of the one-kernel step, geometry from the tile's LDS-staged nodes
            // instead of the 48-byte record per staged element (same values bit for bit; 25 % fewer bytes but one barrier
            // and the geometry arithmetic more: 72.0 against 66.4 us at 1 M elements, profiles/r02_ablation.md)
            const size_t lds_stage = (((size_t)c->lds_L * 15 * sizeof(double) + (size_t)c->lds_C * 2 + 15) & ~(size_t)15) +
                                     (size_t)c->lds_NL * 2 * sizeof(double2);
            const size_t lds = std::max(lds_stage, (size_t)c->tile * 3 * sizeof(double2));
            const int n_wg = c->n_wg_p1;
            const int chunk = (n_wg + 7) / 8;
#define ASM4(RNG, EPT, NPT)                                                                                              \
    do {                                                                                                                 \
        if (lds > 64 * 1024)                                                                                             \
            HIP_TRY(hipFuncSetAttribute((const void*)p1_fused_kernel<false, 256, RNG, EPT, NPT, true>,                   \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
        hipLaunchKernelGGL((p1_fused_kernel<false, 256, RNG, EPT, NPT, true>), dim3(8 * chunk), dim3(256), lds, st,      \
                           c->n_e, c->lds_L, c->lds_C, c->lds_NL, c->perm_l, c->wg_elist, (const int4*)c->wg_rng,        \
                           c->wg_nlist, (const int4*)c->wg_nrng, c->el_nodes, c->pk, c->tdesc, c->xy, c->p1tab,          \
                           (const double*)nullptr, make_e0(nullptr), (const double*)nullptr, c->shear, c->bulk, c->eta,  \
                           c->c, c->matu, (double*)nullptr, (double*)nullptr, (double*)nullptr, (uint8_t*)nullptr,       \
                           k_data, f_out, n_wg, (unsigned long long*)nullptr, k_data ? ds : (const double*)nullptr,      \
                           f_out ? s : (const double*)nullptr, c->n_count_blocks, c->blk_counts, counts_d);              \
    } while (0)
            if (c->lds_L > 256 || c->lds_NL > 256) return FEP_ESTATE;
            if (c->p1_fused_rng) ASM4(true, 1, 1); else ASM4(false, 1, 1);
#undef ASM4
            if (counts_done) *counts_done = counts_d != nullptr;
        } else
#endif
        if (c->p1_lds) {
            // operands + codes while gathering, then the same LDS holds the tile's output (4*TPB values + forces)
            const size_t lds = std::max((size_t)c->lds_L * 15 * sizeof(double) + (size_t)c->lds_C * sizeof(uint16_t),
                                        (size_t)c->tile * 3 * sizeof(double2));
            const int n_wg = c->n_wg_p1;
            const int chunk = (n_wg + 7) / 8;
            if (verbose_on()) {
                static bool once = false;
                if (!once) {
                    once = true;
                    std::fprintf(stderr, "[fep] p1 assembly kernel: %d tiles of <= %d blocks, staged elements <= %d, codes <= %d, "
                                 "LDS %zu bytes per workgroup\n", n_wg, c->tile, c->lds_L, c->lds_C, lds);
                }
            }
#define NODE_LDS3(TPB, RNG, EPT, PK)                                                                                     \
    do {                                                                                                                 \
        if (lds > 64 * 1024)                                                                                             \
            HIP_TRY(hipFuncSetAttribute((const void*)p1_node_lds_kernel<TPB, RNG, EPT, PK>,                              \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
        hipLaunchKernelGGL((p1_node_lds_kernel<TPB, RNG, EPT, PK>), dim3(8 * chunk), dim3(TPB), lds, st,                 \
                           c->n_e, c->lds_L, c->lds_C, c->segptr, c->perm_l, c->meta, c->ncol, c->wg_elist,              \
                           (const int4*)c->wg_rng, c->pk, c->tdesc, c->geo, k_data ? ds : nullptr,                       \
                           f_out ? s : nullptr, k_data, f_out, n_wg, c->n_count_blocks, c->blk_counts, counts_d);        \
    } while (0)
#ifdef FEP_ABLATION
#define NODE_LDS2(TPB, RNG, EPT) do { if (c->p1_pk) NODE_LDS3(TPB, RNG, EPT, true); else NODE_LDS3(TPB, RNG, EPT, false); } while (0)
#else       // (a plan without packed descriptors never reaches the node route in the product: fep_ctx_create takes the element route)
#define NODE_LDS2(TPB, RNG, EPT) do { if (!c->p1_pk) return FEP_ESTATE; NODE_LDS3(TPB, RNG, EPT, true); } while (0)
#endif
#define NODE_LDS(TPB)                                                                                                    \
    do {                                                                                                                 \
        if (c->lds_L > TPB) return FEP_ESTATE;          /* one staged element per lane (fep_host.h) */                  \
        if (c->p1_rng) NODE_LDS2(TPB, true, 1); else NODE_LDS2(TPB, false, 1);                                           \
    } while (0)
#ifdef FEP_ABLATION
            if (c->tile == 128) NODE_LDS(128); else
#endif
            NODE_LDS(256);                 // tiles of 128 / 512 blocks were measured slower (profiles/r01_ablation.md)
#undef NODE_LDS
#undef NODE_LDS2
#undef NODE_LDS3
            if (counts_done) *counts_done = counts_d != nullptr;
        } else {
#ifdef FEP_ABLATION
            hipLaunchKernelGGL(p1_node_kernel, dim3(grid_for(c->n_blk, kBlock)), dim3(kBlock), 0, st,
                               c->n_blk, c->n_e, c->segptr, c->perm2, c->meta, c->ncol, c->geo,
                               k_data ? ds : nullptr, f_out ? s : nullptr, k_data, f_out);
#else
            return FEP_ESTATE;
#endif
        }
        HIP_TRY(hipGetLastError());
    }
    FEP_TRY(prof_mark(c, st));
    FEP_TRY(prof_mark(c, st));
    return FEP_OK;
}

#ifdef FEP_ABLATION
// node route for P2 / Q1 / Q2: assembly kernel
static int launch_gn_node(fep_ctx* c, hipStream_t st, const double* ds, const double* s, double* k_data, double* f_out,
                          unsigned long long* counts_d, bool* counts_done) {
    FEP_TRY(prof_mark(c, st));
    if (counts_done) *counts_done = false;
    if ((k_data && ds) || (f_out && s)) {
        const int n_wg = (int)grid_for(c->n_blk, c->gn_tile);
        const int chunk = (n_wg + 7) / 8;
        const size_t lds = c->gn_lds;
#define GN(NP, NQ, TPB)                                                                                                  \
    do {                                                                                                                 \
        if (lds > 48 * 1024)                                                                                             \
            HIP_TRY(hipFuncSetAttribute((const void*)node_lds_kernel<NP, NQ, TPB>,                                       \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
        hipLaunchKernelGGL((node_lds_kernel<NP, NQ, TPB>), dim3(8 * chunk), dim3(TPB), lds, st, c->n_blk, c->n_e,        \
                           c->lds_L, c->lds_C, c->segptr, c->perm_l, c->meta, c->ncol, c->wg_elist, c->elem, c->xy,      \
                           c->dh1, c->dh2, c->wf, k_data ? ds : nullptr, f_out ? s : nullptr, k_data, f_out, n_wg,       \
                           c->n_count_blocks, c->blk_counts, counts_d);                                                  \
    } while (0)
        const bool t256 = c->gn_tile == 256;
        switch (c->elem_type) {
            case FEP_P2: if (t256) GN(6, 7, 256); else GN(6, 7, 128); break;
            case FEP_Q1: if (t256) GN(4, 4, 256); else GN(4, 4, 128); break;
            case FEP_Q2: if (t256) GN(8, 9, 256); else GN(8, 9, 128); break;
            default: return FEP_EINVAL;
        }
#undef GN
        HIP_TRY(hipGetLastError());
        if (counts_done) *counts_done = counts_d != nullptr;
    }
    FEP_TRY(prof_mark(c, st));
    FEP_TRY(prof_mark(c, st));
    return FEP_OK;
}

template <int NP, int NQ>
static int launch_point(fep_ctx* c, hipStream_t st, const double* u, E0 e0, double* ep, int accept, double* eout,
                        double* s, double* ds, uint8_t* indp, uint2* blk) {
    hipLaunchKernelGGL((point_kernel<NP, NQ>), dim3(grid_for(c->n_int, kBlock)), dim3(kBlock), 0, st, c->n_e, c->elem,
                       c->xy, c->dh1, c->dh2, c->wf, u, e0, ep, c->shear, c->bulk, c->eta, c->c, c->matu, accept, eout, s, ds, indp, blk);
    HIP_TRY(hipGetLastError());
    return FEP_OK;
}

#endif  // FEP_ABLATION

// P1, one kernel per step (non-accepting calls with K and/or F wanted): p1_fused_kernel
static int launch_p1_fused(fep_ctx* c, hipStream_t st, const double* u, E0 e0, const double* ep, double* eout, double* s,
                           double* ds, uint8_t* indp, double* k_data, double* f_out, unsigned long long* counts_d) {
    const bool dma = c->p1_dma;
    (void)dma;
    const size_t Cp = dma ? ((size_t)c->lds_C + 127) & ~(size_t)127 : (size_t)c->lds_C;
    const size_t NLp = dma ? ((size_t)c->lds_NL + 63) & ~(size_t)63 : (size_t)c->lds_NL;
    const size_t lds_stage = (((size_t)c->lds_L * 15 * sizeof(double) + Cp * 2 + 15) & ~(size_t)15) + NLp * 2 * sizeof(double2);
    const size_t lds = std::max(lds_stage, (size_t)c->tile * 3 * sizeof(double2));
    const int n_wg = c->n_wg_p1;
    const int chunk = (n_wg + 7) / 8;
    const bool full = eout || s || ds || indp;
#ifdef FEP_ABLATION
#define FUSED4(FULL, RNG, EPT, NPT)  do { if (dma) FUSED5(FULL, RNG, EPT, NPT, true); else FUSED5(FULL, RNG, EPT, NPT, false); } while (0)
#else
#define FUSED4(FULL, RNG, EPT, NPT)  FUSED5(FULL, RNG, EPT, NPT, false)
#endif
#define FUSED6(FULL, RNG, EPT, NPT, DMA, TPB)                                                                                 \
    do {                                                                                                                 \
        if (lds > 64 * 1024)                                                                                             \
            HIP_TRY(hipFuncSetAttribute((const void*)p1_fused_kernel<FULL, TPB, RNG, EPT, NPT, false, DMA>,              \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
        hipLaunchKernelGGL((p1_fused_kernel<FULL, TPB, RNG, EPT, NPT, false, DMA>), dim3(8 * chunk), dim3(TPB), lds, st, \
                           c->n_e, c->lds_L, c->lds_C, c->lds_NL, c->perm_l, c->wg_elist, (const int4*)c->wg_rng,        \
                           c->wg_nlist, (const int4*)c->wg_nrng, c->el_nodes, c->pk, c->tdesc, c->xy,                   \
                           c->p1tab, u, e0, ep, c->shear, c->bulk, c->eta, c->c, c->matu, eout, s, ds, indp, k_data,      \
                           f_out, n_wg, counts_d ? c->slot_counts : (unsigned long long*)nullptr,                        \
                           (const double*)nullptr, (const double*)nullptr, 0, (const uint2*)nullptr,                     \
                           (unsigned long long*)nullptr);                                                                \
    } while (0)
#ifdef FEP_ABLATION
#define FUSED5(FULL, RNG, EPT, NPT, DMA) do { if (c->tile == 128) FUSED6(FULL, RNG, EPT, NPT, DMA, 128); else FUSED6(FULL, RNG, EPT, NPT, DMA, 256); } while (0)
#else
#define FUSED5(FULL, RNG, EPT, NPT, DMA) FUSED6(FULL, RNG, EPT, NPT, DMA, 256)
#endif
#define FUSED3(FULL, RNG) FUSED4(FULL, RNG, 1, 1)      /* L, NL <= threads: one staged element / node per lane (fep_host.h) */
    if (c->lds_L > c->tile || c->lds_NL > c->tile) return FEP_ESTATE;
#ifdef FEP_ABLATION
    if (full) { if (c->p1_fused_rng) FUSED3(true, true); else FUSED3(true, false); } else
#else
    if (full) return FEP_ESTATE;                         // (the product runs the one-kernel step for K,F-only calls: fused_mode 1)
#endif
    { if (c->p1_fused_rng) FUSED3(false, true); else FUSED3(false, false); }
#undef FUSED3
#undef FUSED4
#undef FUSED5
#undef FUSED6
    HIP_TRY(hipGetLastError());
    if (counts_d) {
        hipLaunchKernelGGL(counts_finalize_kernel, dim3(1), dim3(256), 0, st, c->slot_counts, counts_d);
        HIP_TRY(hipGetLastError());
    }
    return FEP_OK;
}

// The internal ds / s scratch of the two-kernel routes is allocated on first use.  Allocating is illegal while the
// stream is being captured into a graph: such a call is refused (FEP_ESTATE) instead of breaking the capture — run the
// same call once outside the capture first (fep.h, fep_step_dev).
static int scratch_alloc_allowed(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return FEP_OK; }
    return cs == hipStreamCaptureStatusNone ? FEP_OK : FEP_ESTATE;
}

extern "C" int fep_step_dev(fep_ctx* c, void* stream, const double* u_d, const double* e0_h,
                            double* ep_prev_d, int accept, double* e_out_d, double* s_d, double* ds_d,
                            uint8_t* ind_p_d, double* k_data_d, double* f_out_d, int64_t* counts_d) {
    if (!c || !u_d) return FEP_EINVAL;
    if (!fep_aligned16(u_d) || !fep_aligned16(f_out_d) || !fep_aligned16(k_data_d)) return FEP_EINVAL;
    if (!c->have_materials) return FEP_ESTATE;
    FEP_TRY(fep_set_device(c->device));
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* cnt = (unsigned long long*)counts_d;
    uint2* blk = cnt ? c->blk_counts : nullptr;
    const E0 e0 = make_e0(e0_h);
    if ((c->p1_node || c->gn) && c->has_orphans && f_out_d)        // nodes of no element: no block, no lane writes their force
        HIP_TRY(hipMemsetAsync(f_out_d, 0, (size_t)c->n_dof * sizeof(double), st));
    const bool point_outputs = e_out_d || s_d || ds_d || ind_p_d;
    if (c->p1_node && c->p1_fused && !accept && (k_data_d || f_out_d) && c->fused_mode > (point_outputs ? 1 : 0)) {
        FEP_TRY(prof_mark(c, st));
        FEP_TRY(prof_mark(c, st));                      // profile slots: [0] = 0, [1] = the fused kernel (+ counter sum), [2] = 0
        FEP_TRY(launch_p1_fused(c, st, u_d, e0, ep_prev_d, e_out_d, s_d, ds_d, ind_p_d, k_data_d, f_out_d, cnt));
        FEP_TRY(prof_mark(c, st));
        FEP_TRY(prof_mark(c, st));
        return FEP_OK;
    }
    if (c->p1_node) {
        // the assembly kernel consumes ds / s from HBM: use internal buffers when the caller wants neither
        if (k_data_d && !ds_d) {
            if (!c->ds_int) { FEP_TRY(scratch_alloc_allowed(st)); FEP_TRY(dmalloc(&c->ds_int, 9 * c->n_int)); }
            ds_d = c->ds_int;
        }
        if (f_out_d && !s_d) {
            if (!c->s_int) { FEP_TRY(scratch_alloc_allowed(st)); FEP_TRY(dmalloc(&c->s_int, 4 * c->n_int)); }
            s_d = c->s_int;
        }
        FEP_TRY(prof_mark(c, st));
        hipLaunchKernelGGL(p1_point_kernel, dim3(grid_for(c->n_e, kBlock)), dim3(kBlock), 0, st,
                           c->n_e, c->elem, c->xy, c->p1tab, u_d, e0, ep_prev_d, c->shear, c->bulk, c->eta, c->c, c->matu, accept,
                           e_out_d, s_d, ds_d, ind_p_d, blk);
        HIP_TRY(hipGetLastError());
        bool counted = false;
        FEP_TRY(launch_p1_node(c, st, ds_d, s_d, k_data_d, f_out_d, cnt, &counted));
        return counted ? FEP_OK : launch_counts(c, st, cnt);
    }
#ifdef FEP_ABLATION
    if (c->gn) {
        if (k_data_d && !ds_d) {
            if (!c->ds_int) { FEP_TRY(scratch_alloc_allowed(st)); FEP_TRY(dmalloc(&c->ds_int, 9 * c->n_int)); }
            ds_d = c->ds_int;
        }
        if (f_out_d && !s_d) {
            if (!c->s_int) { FEP_TRY(scratch_alloc_allowed(st)); FEP_TRY(dmalloc(&c->s_int, 4 * c->n_int)); }
            s_d = c->s_int;
        }
        FEP_TRY(prof_mark(c, st));
        switch (c->elem_type) {
            case FEP_P2: FEP_TRY((launch_point<6, 7>(c, st, u_d, e0, ep_prev_d, accept, e_out_d, s_d, ds_d, ind_p_d, blk))); break;
            case FEP_Q1: FEP_TRY((launch_point<4, 4>(c, st, u_d, e0, ep_prev_d, accept, e_out_d, s_d, ds_d, ind_p_d, blk))); break;
            case FEP_Q2: FEP_TRY((launch_point<8, 9>(c, st, u_d, e0, ep_prev_d, accept, e_out_d, s_d, ds_d, ind_p_d, blk))); break;
            default: return FEP_EINVAL;
        }
        bool counted = false;
        FEP_TRY(launch_gn_node(c, st, ds_d, s_d, k_data_d, f_out_d, cnt, &counted));
        return counted ? FEP_OK : launch_counts(c, st, cnt);
    }
#endif
    FEP_TRY(prof_mark(c, st));
    if (c->has_orphans && c->patch && f_out_d)                    // nodes of no element: no item writes their force
        HIP_TRY(hipMemsetAsync(f_out_d, 0, (size_t)c->n_dof * sizeof(double), st));
#define CALL(NP, NQ)                                                                                     \
    FEP_TRY((launch_element<NP, NQ, true>(c, st, u_d, e0, ep_prev_d, accept, e_out_d, s_d, ds_d, ind_p_d, blk, \
                                          c->patch ? k_data_d : (k_data_d ? c->Kc : nullptr),            \
                                          c->patch ? f_out_d : (f_out_d ? c->fe : nullptr))))
    bool counted = false;
#ifdef FEP_ABLATION
    // FEP_FIX_SIDE=1 — an UPPER BOUND, not a route: fixup_kernel on a side stream, NOT ordered behind this call's element kernel (it
    // adds whatever partials the call before left; K's open blocks are stale), joined at the end: what a perfectly overlapped
    // fix-up could save
    if (c->patch && fep_tune("FEP_FIX_SIDE")) {
        static hipStream_t side = nullptr;
        static hipEvent_t ev_a = nullptr, ev_b = nullptr;
        if (!side) { HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking)); HIP_TRY(hipEventCreateWithFlags(&ev_a, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&ev_b, hipEventDisableTiming)); }
        HIP_TRY(hipEventRecord(ev_a, st));
        HIP_TRY(hipStreamWaitEvent(side, ev_a, 0));
        FEP_TRY(launch_fixup(c, side, k_data_d, f_out_d, cnt, &counted));
        HIP_TRY(hipEventRecord(ev_b, side));
        DISPATCH_ELEM(c->elem_type, CALL)
        HIP_TRY(hipStreamWaitEvent(st, ev_b, 0));
        return counted ? FEP_OK : launch_counts(c, st, cnt);
    }
#endif
    DISPATCH_ELEM(c->elem_type, CALL)
#undef CALL
    if (c->patch) FEP_TRY(launch_fixup(c, st, k_data_d, f_out_d, cnt, &counted));
    else FEP_TRY(launch_reduce(c, st, k_data_d, f_out_d, cnt, &counted));
    return counted ? FEP_OK : launch_counts(c, st, cnt);
}

extern "C" int fep_assemble_dev(fep_ctx* c, void* stream, const double* ds_d, const double* s_d,
                                double* k_data_d, double* f_out_d) {
    if (!c) return FEP_EINVAL;
    if ((k_data_d && !ds_d) || (f_out_d && !s_d)) return FEP_EINVAL;
    if (!fep_aligned16(f_out_d) || !fep_aligned16(k_data_d)) return FEP_EINVAL;
    FEP_TRY(fep_set_device(c->device));
    hipStream_t st = (hipStream_t)stream;
    if ((c->p1_node || c->gn) && c->has_orphans && f_out_d)
        HIP_TRY(hipMemsetAsync(f_out_d, 0, (size_t)c->n_dof * sizeof(double), st));
    if (c->p1_node) {
        FEP_TRY(prof_mark(c, st));
        return launch_p1_node(c, st, ds_d, s_d, k_data_d, f_out_d, nullptr, nullptr);
    }
#ifdef FEP_ABLATION
    if (c->gn) {
        FEP_TRY(prof_mark(c, st));
        return launch_gn_node(c, st, ds_d, s_d, k_data_d, f_out_d, nullptr, nullptr);
    }
#endif
    const E0 e0 = make_e0(nullptr);
    FEP_TRY(prof_mark(c, st));
    if (c->has_orphans && c->patch && f_out_d)
        HIP_TRY(hipMemsetAsync(f_out_d, 0, (size_t)c->n_dof * sizeof(double), st));
#define CALL(NP, NQ)                                                                                         \
    FEP_TRY((launch_element<NP, NQ, false>(c, st, nullptr, e0, nullptr, 0, nullptr, const_cast<double*>(s_d), \
                                           const_cast<double*>(ds_d), nullptr, nullptr,                      \
                                           c->patch ? k_data_d : (k_data_d ? c->Kc : nullptr),               \
                                           c->patch ? f_out_d : (f_out_d ? c->fe : nullptr))))
    DISPATCH_ELEM(c->elem_type, CALL)
#undef CALL
    if (c->patch) return launch_fixup(c, st, k_data_d, f_out_d, nullptr, nullptr);
    return launch_reduce(c, st, k_data_d, f_out_d);
}

static int step_host_impl(fep_ctx* c, bool u_planar, const double* u_h, const double* e0_h, double* ep_prev_h, int accept,
                          double* e_out_h, double* s_h, double* ds_h, uint8_t* ind_p_h,
                          double* k_data_h, double* f_out_h, int64_t* counts_h) {
    if (!c || !u_h) return FEP_EINVAL;
    FEP_TRY(fep_set_device(c->device));
    const int64_t nb = c->n_int * (int64_t)sizeof(double);
    fep_stage::Engine* E = nullptr;
    FEP_TRY(fep_stage::engine(c->device, &E));
    fep_stage::EngineCall call(E);                      // drains the engine on every exit that is not a completed finish()
    void *u, *ep = nullptr, *eo = nullptr, *s = nullptr, *ds = nullptr, *ip = nullptr, *kd = nullptr, *f = nullptr, *cnt;
    FEP_TRY(ctx_buf(c, 0, c->n_dof * (int64_t)sizeof(double), &u));
    if (ep_prev_h) FEP_TRY(ctx_buf(c, 1, 4 * nb, &ep));
    if (e_out_h) FEP_TRY(ctx_buf(c, 2, 3 * nb, &eo));
    if (s_h) FEP_TRY(ctx_buf(c, 3, 4 * nb, &s));
    if (ds_h) FEP_TRY(ctx_buf(c, 4, 9 * nb, &ds));
    if (ind_p_h) FEP_TRY(ctx_buf(c, 5, c->n_int, &ip));
    if (k_data_h) FEP_TRY(ctx_buf(c, 6, c->nnz * (int64_t)sizeof(double), &kd));
    if (f_out_h) FEP_TRY(ctx_buf(c, 7, c->n_dof * (int64_t)sizeof(double), &f));
    FEP_TRY(ctx_buf(c, 8, 2 * sizeof(int64_t), &cnt));
    if (u_planar) FEP_TRY(E->h2d_interleave2(u, u_h, (size_t)c->n_n));
    else FEP_TRY(E->h2d(u, u_h, (size_t)c->n_dof * sizeof(double)));
    if (ep_prev_h) FEP_TRY(E->h2d(ep, ep_prev_h, (size_t)(4 * nb)));
    FEP_TRY(fep_step_dev(c, E->stream, (const double*)u, e0_h, (double*)ep, accept, (double*)eo, (double*)s, (double*)ds,
                         (uint8_t*)ip, (double*)kd, (double*)f, (int64_t*)cnt));
    // the largest result first: its transfer hides the others' set-up
    if (k_data_h) FEP_TRY(E->d2h(k_data_h, kd, (size_t)c->nnz * sizeof(double)));
    if (f_out_h) FEP_TRY(E->d2h(f_out_h, f, (size_t)c->n_dof * sizeof(double)));
    if (ds_h) FEP_TRY(E->d2h(ds_h, ds, (size_t)(9 * nb)));
    if (s_h) FEP_TRY(E->d2h(s_h, s, (size_t)(4 * nb)));
    if (e_out_h) FEP_TRY(E->d2h(e_out_h, eo, (size_t)(3 * nb)));
    if (ind_p_h) FEP_TRY(E->d2h(ind_p_h, ip, (size_t)c->n_int));
    if (counts_h) FEP_TRY(E->d2h(counts_h, cnt, 2 * sizeof(int64_t)));
    if (accept && ep_prev_h) FEP_TRY(E->d2h(ep_prev_h, ep, (size_t)(4 * nb)));
    return call.finish();
}

extern "C" int fep_step_host(fep_ctx* c, const double* u_h, const double* e0_h, double* ep_prev_h, int accept,
                             double* e_out_h, double* s_h, double* ds_h, uint8_t* ind_p_h,
                             double* k_data_h, double* f_out_h, int64_t* counts_h) {
    FEP_GUARD(step_host_impl(c, false, u_h, e0_h, ep_prev_h, accept, e_out_h, s_h, ds_h, ind_p_h, k_data_h, f_out_h, counts_h))
}

extern "C" int fep_step_host_planar(fep_ctx* c, const double* u2_h, const double* e0_h, double* ep_prev_h, int accept,
                                    double* e_out_h, double* s_h, double* ds_h, uint8_t* ind_p_h,
                                    double* k_data_h, double* f_out_h, int64_t* counts_h) {
    FEP_GUARD(step_host_impl(c, true, u2_h, e0_h, ep_prev_h, accept, e_out_h, s_h, ds_h, ind_p_h, k_data_h, f_out_h, counts_h))
}

static int assemble_host_impl(fep_ctx* c, const double* ds_h, const double* s_h, double* k_data_h, double* f_out_h) {
    if (!c) return FEP_EINVAL;
    if ((k_data_h && !ds_h) || (f_out_h && !s_h)) return FEP_EINVAL;
    FEP_TRY(fep_set_device(c->device));
    const int64_t nb = c->n_int * (int64_t)sizeof(double);
    fep_stage::Engine* E = nullptr;
    FEP_TRY(fep_stage::engine(c->device, &E));
    fep_stage::EngineCall call(E);                      // drains the engine on every exit that is not a completed finish()
    void *ds = nullptr, *s = nullptr, *kd = nullptr, *f = nullptr;
    if (ds_h) { FEP_TRY(ctx_buf(c, 4, 9 * nb, &ds)); FEP_TRY(E->h2d(ds, ds_h, (size_t)(9 * nb))); }
    if (s_h) { FEP_TRY(ctx_buf(c, 3, 4 * nb, &s)); FEP_TRY(E->h2d(s, s_h, (size_t)(3 * nb))); }
    if (k_data_h) FEP_TRY(ctx_buf(c, 6, c->nnz * (int64_t)sizeof(double), &kd));
    if (f_out_h) FEP_TRY(ctx_buf(c, 7, c->n_dof * (int64_t)sizeof(double), &f));
    FEP_TRY(fep_assemble_dev(c, E->stream, (const double*)ds, (const double*)s, (double*)kd, (double*)f));
    if (k_data_h) FEP_TRY(E->d2h(k_data_h, kd, (size_t)c->nnz * sizeof(double)));
    if (f_out_h) FEP_TRY(E->d2h(f_out_h, f, (size_t)c->n_dof * sizeof(double)));
    return call.finish();
}

extern "C" int fep_assemble_host(fep_ctx* c, const double* ds_h, const double* s_h, double* k_data_h, double* f_out_h) {
    FEP_GUARD(assemble_host_impl(c, ds_h, s_h, k_data_h, f_out_h))
}

extern "C" int fep_gather_f64(int device_id, void* stream, int64_t n, const double* src_d, const int32_t* idx_d, double* dst_d) {
    if (n < 0 || (n > 0 && (!src_d || !idx_d || !dst_d))) return FEP_EINVAL;
    FEP_TRY(fep_set_device(device_id));
    if (n == 0) return FEP_OK;
    hipLaunchKernelGGL(gather_or_zero_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n, src_d, idx_d, dst_d);
    HIP_TRY(hipGetLastError());
    return FEP_OK;
}

extern "C" int fep_scatter_f64(int device_id, void* stream, int64_t n, const double* src_d, const int32_t* src_idx_d,
                               const int32_t* dst_idx_d, double* dst_d) {
    if (n < 0 || (n > 0 && (!src_d || !src_idx_d || !dst_idx_d || !dst_d))) return FEP_EINVAL;
    FEP_TRY(fep_set_device(device_id));
    if (n == 0) return FEP_OK;
    hipLaunchKernelGGL(scatter_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n, src_d, src_idx_d,
                       dst_idx_d, dst_d);
    HIP_TRY(hipGetLastError());
    return FEP_OK;
}

extern "C" int fep_iface_sum_f64(int device_id, void* stream, int64_t n, const int32_t* loc_d, const int32_t* ptr_d,
                                 const int32_t* src_d, const double* recv_d, double* f_d) {
    if (n < 0 || (n > 0 && (!loc_d || !ptr_d || !src_d || !f_d))) return FEP_EINVAL;
    FEP_TRY(fep_set_device(device_id));
    if (n == 0) return FEP_OK;
    hipLaunchKernelGGL(iface_sum_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n, loc_d, ptr_d, src_d,
                       recv_d, f_d);
    HIP_TRY(hipGetLastError());
    return FEP_OK;
}

extern "C" int fep_transform_dev(fep_ctx* c, void* stream, const double* q_int_d, double* q_node_d) {
    if (!c || !q_int_d || !q_node_d) return FEP_EINVAL;
    FEP_TRY(fep_set_device(c->device));
    hipLaunchKernelGGL(nodal_average_kernel, dim3(grid_for(c->n_n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       c->n_n, c->n_e, c->n_q, c->iptr, c->ilist, c->weight, q_int_d, q_node_d);
    HIP_TRY(hipGetLastError());
    return FEP_OK;
}

static int transform_host_impl(fep_ctx* c, const double* q_int_h, double* q_node_h) {
    if (!c || !q_int_h || !q_node_h) return FEP_EINVAL;
    FEP_TRY(fep_set_device(c->device));
    fep_stage::Engine* E = nullptr;
    FEP_TRY(fep_stage::engine(c->device, &E));
    fep_stage::EngineCall call(E);                      // drains the engine on every exit that is not a completed finish()
    void *q, *o;
    FEP_TRY(ctx_buf(c, 9, c->n_int * (int64_t)sizeof(double), &q));
    FEP_TRY(ctx_buf(c, 10, c->n_n * (int64_t)sizeof(double), &o));
    FEP_TRY(E->h2d(q, q_int_h, (size_t)c->n_int * sizeof(double)));
    FEP_TRY(fep_transform_dev(c, E->stream, (const double*)q, (double*)o));
    FEP_TRY(E->d2h(q_node_h, o, (size_t)c->n_n * sizeof(double)));
    return call.finish();
}

extern "C" int fep_transform_host(fep_ctx* c, const double* q_int_h, double* q_node_h) {
    FEP_GUARD(transform_host_impl(c, q_int_h, q_node_h))
}

extern "C" int fep_ctx_profile_begin(fep_ctx* c) {
    if (!c) return FEP_EINVAL;
    for (hipEvent_t ev : c->events) (void)hipEventDestroy(ev);
    c->events.clear();
    c->profiling = true;
    return FEP_OK;
}

extern "C" int fep_ctx_profile_end(fep_ctx* c, void* stream, double ms_out[3], int* n_steps) {
    if (!c || !ms_out) return FEP_EINVAL;
    c->profiling = false;
    FEP_TRY(fep_set_device(c->device));
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    double acc[3] = {0.0, 0.0, 0.0};
    const int n = (int)(c->events.size() / 4);
    for (int s = 0; s < n && e == hipSuccess; ++s)
        for (int k = 0; k < 3 && e == hipSuccess; ++k) {
            float ms = 0.f;
            e = hipEventElapsedTime(&ms, c->events[4 * s + k], c->events[4 * s + k + 1]);
            acc[k] += ms;
        }
    for (hipEvent_t ev : c->events) (void)hipEventDestroy(ev);
    c->events.clear();
    if (e != hipSuccess) { g_last_hip = (int)e; (void)hipGetLastError(); return FEP_EHIP; }
    for (int k = 0; k < 3; ++k) ms_out[k] = n ? acc[k] / n : 0.0;
    if (n_steps) *n_steps = n;
    return FEP_OK;
}
