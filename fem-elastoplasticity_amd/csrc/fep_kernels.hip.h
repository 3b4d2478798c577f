// HIP kernels (gfx950 / CDNA4, wave64) for the Drucker-Prager return map and the element
// tangent / internal-force assembly.  fp64 throughout.  All per-point arrays are SoA
// ("rows x n_int", component-major) so that consecutive lanes read consecutive doubles.
//
// Reference lines restated here (DP = Plasticity2D_DP/pythonFEM.py):
//   geometry            DP:506-546, 585
//   strain              DP:1043
//   return map          DP:646-757 (+ TSX:1052 initial strain)
//   element tangent     DP:1047-1050  (K_e = sum_q w_q B_q^T DS_q B_q)
//   internal force      DP:1058       (f_e = sum_q w_q B_q^T S_q[0:3])
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fep {

constexpr int kBlock = 256;

// FEP_P2_FMA: phase 2 of element_kernel with every product fused into its accumulator (57 instead of 84 vector instructions per
// point and lane; K differs in the last bits from the multiply / multiply-add / add form, -DFEP_P2_FMA=0).  Round 3 measured it
// even (P2 +2 %, Q2 +2.5 %, configs[4] -2.5 %: the kernel then waited for loads it no longer waits for); on round 4's kernel, one
// session, four passes each (profiles/r04_ablation.md section 2d): P4 0.734 against 0.779 ms, Q2 0.538 / 0.582, P2 0.600 / 0.611.  On.
#ifndef FEP_P2_FMA
#define FEP_P2_FMA 1
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding global stores
// (s_waitcnt vmcnt(0)): behind a phase that has just issued its output stores that wait is a full store round trip on
// the workgroup's critical path.  Use where the phases exchange data through LDS alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------
// Per-point Drucker-Prager return map (DP:663-755).
//   e[3]  strain (11,22,gamma12);  z[4] initial strain (TSX e0, zeros for DP);  p[4] previous
//   plastic strain.  Outputs: s[4] stress, d[6] symmetric tangent (00,01,02,11,12,22),
//   p[4] updated in place when `accept`.  Returns 0 elastic, 1 smooth, 2 apex.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int dp_return_map(const double e[3], const double z[4], double p[4],
                                             double G, double K, double eta, double c, bool accept,
                                             double s[4], double d[6]) {
    const double I3 = 1.0 / 3.0;
    const double DD = 1.0 - 1.0 / 3.0;          // diagonal of dev (DP:653)
    const double RS2 = 1.4142135623730951;      // np.sqrt(2)
    const double Et0 = (e[0] + z[0]) - p[0];    // DP:663-668 (C3: E4 aliases E_tr)
    const double Et1 = (e[1] + z[1]) - p[1];
    const double Et2 = (e[2] + z[2]) - p[2];
    const double Et3 = (0.0 + z[3]) - p[3];
    const double tr = Et0 + Et1 + Et3;          // vol @ E_tr
    const double dv0 = DD * Et0 - I3 * Et1 - I3 * Et3;   // dev @ E_tr, DP:673
    const double dv1 = -I3 * Et0 + DD * Et1 - I3 * Et3;
    const double dv2 = 0.5 * Et2;
    const double dv3 = -I3 * Et0 - I3 * Et1 + DD * Et3;
    const double G2 = 2.0 * G;
    const double Ktr = K * tr;                  // p_tr, DP:682
    double s0 = G2 * dv0 + Ktr, s1 = G2 * dv1 + Ktr, s2 = G2 * dv2, s3 = G2 * dv3 + Ktr;   // DP:670
    const double n2 = Et0 * dv0 + Et1 * dv1 + Et2 * dv2 + Et3 * dv3;
    const double nE = sqrt(n2 > 0.0 ? n2 : 0.0);          // DP:676 (C5)
    const double rho = 2.0 * (G * nE);                    // DP:679
    const double da = K * (eta * eta);                    // DP:687
    const double dS = G + da;                             // DP:688
    const double c1 = rho / RS2 + eta * Ktr - c;          // DP:689
    const double c2 = eta * Ktr - da * rho / (G * RS2) - c;   // DP:690
    // elastic tangent 2*Dev*G + Vol*K, DP:703
    double d00 = 2.0 * DD * G + K, d01 = 2.0 * (-I3) * G + K, d02 = 0.0;
    double d11 = d00, d12 = 0.0, d22 = 2.0 * 0.5 * G;
    int branch = 0;
    if (c1 > 0.0) {                                       // DP:693
        if (c2 <= 0.0) {                                  // smooth portion, DP:696
            branch = 1;
            const double lam = c1 / dS;                   // DP:710
            const double N0 = dv0 / nE, N1 = dv1 / nE, N2 = dv2 / nE, N3 = dv3 / nE;   // DP:718
            const double Ke = K * eta;
            const double g = RS2 * G;
            const double M0 = g * N0 + Ke, M1 = g * N1 + Ke, M2 = g * N2, M3 = g * N3 + Ke;  // DP:719
            s0 -= lam * M0; s1 -= lam * M1; s2 -= lam * M2; s3 -= lam * M3;   // DP:720
            const double cf = 2.0 * RS2 * (G * G) * lam / rho;               // DP:727
            d00 = d00 - cf * (DD - N0 * N0) - M0 * M0 / dS;
            d01 = d01 - cf * (-I3 - N0 * N1) - M0 * M1 / dS;
            d02 = d02 - cf * (0.0 - N0 * N2) - M0 * M2 / dS;
            d11 = d11 - cf * (DD - N1 * N1) - M1 * M1 / dS;
            d12 = d12 - cf * (0.0 - N1 * N2) - M1 * M2 / dS;
            d22 = d22 - cf * (0.5 - N2 * N2) - M2 * M2 / dS;
            if (accept) {                                 // DP:752
                const double e3 = eta / 3.0;
                p[0] += lam * (N0 / RS2 + e3);
                p[1] += lam * (N1 / RS2 + e3);
                p[2] += 2.0 * lam * (N2 / RS2);
                p[3] += lam * (N3 / RS2 + e3);
            }
        } else {                                          // apex, DP:699
            branch = 2;
            const double ce = c / eta;                    // DP:721
            s0 = ce; s1 = ce; s2 = 0.0; s3 = ce;
            d00 = d01 = d02 = d11 = d12 = d22 = 0.0;      // DP:728
            if (accept) {                                 // DP:755 (uses E - ep_prev: C3)
                const double sh = c / (3.0 * K * eta);
                p[0] = Et0 - sh; p[1] = Et1 - sh; p[2] = Et2; p[3] = Et3 - sh;
            }
        }
    }
    s[0] = s0; s[1] = s1; s[2] = s2; s[3] = s3;
    d[0] = d00; d[1] = d01; d[2] = d02; d[3] = d11; d[4] = d12; d[5] = d22;
    return branch;
}

// Smooth / apex counters (the numbers the reference logs at DP:730).  Wave ballot -> LDS -> per
// workgroup either a plain store into blk_counts[blockIdx.x] (summed by counts_reduce_kernel; no
// global atomics at all) or, for the mesh-free entry point, one global atomic per workgroup.
// Integer sums => deterministic.  Must be reached by every thread of the block.
__device__ __forceinline__ void count_branches(int branch, unsigned long long* counts, uint2* blk_counts) {
    if (counts == nullptr && blk_counts == nullptr) return;       // uniform
    __shared__ unsigned int sc[2];
    if (threadIdx.x == 0) { sc[0] = 0u; sc[1] = 0u; }
    lds_barrier();                                                // (LDS only: the callers' point outputs are still on their way)
    const unsigned long long ms = __ballot(branch == 1);
    const unsigned long long ma = __ballot(branch == 2);
    if ((threadIdx.x & 63) == 0) {
        if (ms) atomicAdd(&sc[0], (unsigned int)__popcll(ms));
        if (ma) atomicAdd(&sc[1], (unsigned int)__popcll(ma));
    }
    lds_barrier();
    if (threadIdx.x == 0) {
        if (blk_counts) {
            blk_counts[blockIdx.x] = make_uint2(sc[0], sc[1]);
        } else {
            if (sc[0]) atomicAdd(&counts[0], (unsigned long long)sc[0]);
            if (sc[1]) atomicAdd(&counts[1], (unsigned long long)sc[1]);
        }
    }
}

// counts_out[0..1] = sum of blk_counts[0..n_blocks): overwrites (no memset needed).  Called by ONE
// workgroup (any size that is a multiple of 64, <= 1024); all loads of a lane are issued before
// the first use, so the cost is about one memory round trip.
__device__ __forceinline__ void sum_block_counts(int n_blocks, const uint2* __restrict__ blk_counts,
                                                 unsigned long long* __restrict__ counts_out) {
    __shared__ unsigned long long part[2][16];
    unsigned long long a = 0, b = 0;
    const int nt = blockDim.x;
    constexpr int UN = 8;           // loads in flight per lane: the loop is a chain of memory round trips (31 of them with 4 at
                                    // 31 k workgroups and 256 lanes: ~50 us, the floor of the kernel that hosts this sum)
    for (int base = threadIdx.x; base < n_blocks; base += UN * nt) {
        uint2 v[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) v[j] = (base + j * nt < n_blocks) ? blk_counts[base + j * nt] : make_uint2(0u, 0u);
#pragma unroll
        for (int j = 0; j < UN; ++j) { a += v[j].x; b += v[j].y; }
    }
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = a; part[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long ta = 0, tb = 0;
        for (int w = 0; w < (nt >> 6); ++w) { ta += part[0][w]; tb += part[1][w]; }
        counts_out[0] = ta; counts_out[1] = tb;
    }
}

__global__ void __launch_bounds__(1024)
counts_reduce_kernel(int n_blocks, const uint2* __restrict__ blk_counts, unsigned long long* __restrict__ counts_out) {
    sum_block_counts(n_blocks, blk_counts, counts_out);
}

__device__ __forceinline__ void store_point(int64_t k, int64_t n, const double s[4], const double d[6], int branch,
                                            double* __restrict__ S, double* __restrict__ DS,
                                            uint8_t* __restrict__ indp) {
    if (S) { S[k] = s[0]; S[n + k] = s[1]; S[2 * n + k] = s[2]; S[3 * n + k] = s[3]; }
    if (DS) {   // row-major 3x3, m = 3i+j (DP:703)
        DS[k] = d[0];         DS[n + k] = d[1];     DS[2 * n + k] = d[2];
        DS[3 * n + k] = d[1]; DS[4 * n + k] = d[3]; DS[5 * n + k] = d[4];
        DS[6 * n + k] = d[2]; DS[7 * n + k] = d[4]; DS[8 * n + k] = d[5];
    }
    if (indp) indp[k] = branch != 0;
}

// Row m of a point array ("rows x n", C order) at a 32-bit BYTE offset of the point: the row base is uniform (scalar
// registers, scalar arithmetic), the offset one vector register shared by every access of the lane — the
// `global_load/store ... v_off, s[base:base+1]` form.  With 64-bit indices every access carries its own address pair:
// 13 point outputs + 2*NP gradients are 2*(13 + 2*NP) registers and as many 64-bit vector adds.  Callers guarantee
// 8 * n < 2^32 (fep_ctx_create refuses n_int >= 2^27).
// (The row base goes through readfirstlane: it is uniform already, but without the opaque step the compiler re-associates
// base + m*stride + offset into per-lane 64-bit sums again.)
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double ld_row(const double* __restrict__ base, int64_t row_stride, int m, unsigned byte_off) {
    typedef const __attribute__((address_space(1))) char* gptr;        // (global address space: an integer cast alone gives a flat pointer)
    const gptr row = reinterpret_cast<gptr>(uniform_u64(reinterpret_cast<uint64_t>(base + (int64_t)m * row_stride)));
    return *reinterpret_cast<const __attribute__((address_space(1))) double*>(row + byte_off);
}
__device__ __forceinline__ void st_row(double* __restrict__ base, int64_t row_stride, int m, unsigned byte_off, double v) {
    typedef __attribute__((address_space(1))) char* gptr;
    const gptr row = reinterpret_cast<gptr>(uniform_u64(reinterpret_cast<uint64_t>(base + (int64_t)m * row_stride)));
    *reinterpret_cast<__attribute__((address_space(1))) double*>(row + byte_off) = v;
}
__device__ __forceinline__ void store_point_off(unsigned kb, unsigned k32, int64_t n, const double s[4], const double d[6], int branch,
                                                double* __restrict__ S, double* __restrict__ DS, uint8_t* __restrict__ indp) {
    if (S) { st_row(S, n, 0, kb, s[0]); st_row(S, n, 1, kb, s[1]); st_row(S, n, 2, kb, s[2]); st_row(S, n, 3, kb, s[3]); }
    if (DS) {   // row-major 3x3, m = 3i+j (DP:703)
        st_row(DS, n, 0, kb, d[0]); st_row(DS, n, 1, kb, d[1]); st_row(DS, n, 2, kb, d[2]);
        st_row(DS, n, 3, kb, d[1]); st_row(DS, n, 4, kb, d[3]); st_row(DS, n, 5, kb, d[4]);
        st_row(DS, n, 6, kb, d[2]); st_row(DS, n, 7, kb, d[4]); st_row(DS, n, 8, kb, d[5]);
    }
    if (indp) indp[k32] = branch != 0;
}

// ---------------------------------------------------------------------------------------
// Mesh-free pointwise return map: construct_constitutive_problem (DP:604-757 / TSX:990-1157).
// ---------------------------------------------------------------------------------------
struct E0 { double v[4]; };

__global__ void __launch_bounds__(kBlock)
return_map_kernel(int64_t n, const double* __restrict__ e, int64_t eps, int64_t ecs, E0 e0,
                  double* __restrict__ ep, const double* __restrict__ shear, const double* __restrict__ bulk,
                  const double* __restrict__ eta, const double* __restrict__ cc, int accept,
                  double* __restrict__ S, double* __restrict__ DS, uint8_t* __restrict__ indp,
                  uint2* blk_counts) {
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int branch = 0;
    if (k < n) {
        double ev[3] = {e[k * eps], e[k * eps + ecs], e[k * eps + 2 * ecs]};
        double p[4] = {0.0, 0.0, 0.0, 0.0};
        if (ep) { p[0] = ep[k]; p[1] = ep[n + k]; p[2] = ep[2 * n + k]; p[3] = ep[3 * n + k]; }
        double s[4], d[6];
        branch = dp_return_map(ev, e0.v, p, shear[k], bulk[k], eta[k], cc[k], accept != 0, s, d);
        store_point(k, n, s, d, branch, S, DS, indp);
        if (accept && ep && branch) { ep[k] = p[0]; ep[n + k] = p[1]; ep[2 * n + k] = p[2]; ep[3 * n + k] = p[3]; }
    }
    count_branches(branch, nullptr, blk_counts);      // per-workgroup counters, summed by counts_reduce_kernel (no global atomics)
}

// ---------------------------------------------------------------------------------------
// Geometry (setup, once): Jacobian, dphi_1, dphi_2, weight.  DP:530-546, 585.
// Operation order and rounding follow the reference's NumPy expressions exactly
// (no FMA contraction), so dphi / weight are bit-identical to the reference's arrays.
// ---------------------------------------------------------------------------------------
template <int NP, int NQ>
__global__ void __launch_bounds__(kBlock)
geometry_kernel(int64_t n_e, int64_t n_n, const int32_t* __restrict__ elem, const double* __restrict__ coords,
                const double* __restrict__ dh1, const double* __restrict__ dh2, const double* __restrict__ wf,
                double* __restrict__ dphi1, double* __restrict__ dphi2, double* __restrict__ weight,
                double* __restrict__ det_out, double* __restrict__ geo) {
#pragma clang fp contract(off)
    __shared__ double t1[NP * NQ], t2[NP * NQ], tw[NQ];
    for (int i = threadIdx.x; i < NP * NQ; i += kBlock) { t1[i] = dh1[i]; t2[i] = dh2[i]; }
    for (int i = threadIdx.x; i < NQ; i += kBlock) tw[i] = wf[i];
    __syncthreads();
    const int64_t n_int = n_e * NQ;
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (k >= n_int) return;
    const int64_t e = k / NQ;
    const int q = (int)(k - e * NQ);
    double j11 = 0.0, j12 = 0.0, j21 = 0.0, j22 = 0.0;
    for (int a = 0; a < NP; ++a) {                       // builtin sum over rows, DP:530-533
        const int32_t nd = elem[(int64_t)a * n_e + e];
        const double x = coords[nd], y = coords[n_n + nd];
        const double h1 = t1[a * NQ + q], h2 = t2[a * NQ + q];
        j11 = j11 + x * h1; j12 = j12 + y * h1; j21 = j21 + x * h2; j22 = j22 + y * h2;
    }
    const double det = j11 * j22 - j12 * j21;            // DP:536
    const double i11 = j22 / det, i12 = -j12 / det, i21 = -j21 / det, i22 = j11 / det;   // DP:539-542
    for (int a = 0; a < NP; ++a) {
        const double h1 = t1[a * NQ + q], h2 = t2[a * NQ + q];
        dphi1[(int64_t)a * n_int + k] = i11 * h1 + i12 * h2;   // DP:545
        dphi2[(int64_t)a * n_int + k] = i21 * h1 + i22 * h2;   // DP:546
    }
    weight[k] = fabs(det) * tw[q];                       // DP:585
    if (det_out) det_out[k] = det;
    if (NP == 3 && NQ == 1 && geo) {     // 48-byte AoS record of the P1 fast path: d1[0], d1[1], d2[0], d2[1], w, 0;
        double* g = geo + k * 6;         // the third node's gradient is -(first + second) (partition of unity)
        for (int a = 0; a < 2; ++a) {
            const double h1 = t1[a * NQ + q], h2 = t2[a * NQ + q];
            g[a] = i11 * h1 + i12 * h2;
            g[2 + a] = i21 * h1 + i22 * h2;
        }
        g[4] = fabs(det) * tw[q];
        g[5] = 0.0;
    }
}

// ---------------------------------------------------------------------------------------
// Fused element kernel.  One workgroup = EB consecutive elements = EB*NQ consecutive points.
//
//  phase 1 (one lane per integration point): strain from U (a1), return map (a2), results to
//          HBM (coalesced over k) and w*DS, w*S, dphi staged in LDS;
//  phase 2 (one lane per (element, local node a)): K_e = sum_q B^T (w DS) B (a3, a4) is symmetric (DS is), so
//          only half of its 2x2 node-pair blocks are computed and stored: the lane of node a holds the blocks
//          (a, b = (a+j) mod NP), j = 0..NP/2 (for even NP the j = NP/2 block only when a < NP/2) at
//          Kc[((j*NP+a)*n_e + e)*4 + 2i+jj]; the block (b, a) is its transpose (sym_block_index below).
//          f_e = sum_q B^T (w S) (a5) as pairs fe[(a*n_e+e)*2 + i].
//
//  FROM_U = true : inputs U (+ ep, materials);  FROM_U = false : inputs DS, S (assembly only).
// ---------------------------------------------------------------------------------------
// dphi and weight of one integration point from the element's node coordinates: same operations, same order,
// no FMA contraction as geometry_kernel, hence bit-identical to the stored arrays (DP:530-546, 585).
template <int NP>
__device__ __forceinline__ void geometry_at_q(const double* __restrict__ t1, const double* __restrict__ t2, double wfq,
                                              int q, int nq, const double x[NP], const double y[NP],
                                              double d1[NP], double d2[NP], double& w) {
#pragma clang fp contract(off)
    double j11 = 0.0, j12 = 0.0, j21 = 0.0, j22 = 0.0;
#pragma unroll
    for (int a = 0; a < NP; ++a) {                       // DP:530-533
        const double h1 = t1[a * nq + q], h2 = t2[a * nq + q];
        j11 = j11 + x[a] * h1; j12 = j12 + y[a] * h1; j21 = j21 + x[a] * h2; j22 = j22 + y[a] * h2;
    }
    const double det = j11 * j22 - j12 * j21;
    const double i11 = j22 / det, i12 = -j12 / det, i21 = -j21 / det, i22 = j11 / det;
#pragma unroll
    for (int a = 0; a < NP; ++a) {
        const double h1 = t1[a * nq + q], h2 = t2[a * nq + q];
        d1[a] = i11 * h1 + i12 * h2;                     // DP:545-546
        d2[a] = i21 * h1 + i22 * h2;
    }
    w = fabs(det) * wfq;                                 // DP:585
}

// Homogeneous material (the reference's demos: one constant per parameter, DP:972-984): the four per-point
// arrays are not read at all.  `on` is set by fep_ctx_set_materials_host when every array is constant.
struct MatU { double shear, bulk, eta, c; int on; };

// Where element_kernel keeps the block (a, b) of K_e: index of the stored block and whether it is the transpose.
// `row_major`: the stored blocks of an element are numbered a*(n_p/2+1) + j (a lane's blocks adjacent: the AoS layout,
// Kc[(e*NB + idx)*4]) instead of j*n_p + a (the SoA layout, Kc[(idx*n_e + e)*4]).
inline __host__ __device__ void sym_block_index(int n_p, int a, int b, int& idx, bool& transposed, bool row_major = false) {
    const int j = b >= a ? b - a : b - a + n_p;
    const bool direct = 2 * j < n_p || (2 * j == n_p && a < n_p / 2);
    const int nj = n_p / 2 + 1;
    if (direct) { idx = row_major ? a * nj + j : j * n_p + a; transposed = false; }
    else { idx = row_major ? b * nj + (n_p - j) : (n_p - j) * n_p + b; transposed = true; }
}
inline __host__ __device__ int sym_block_count(int n_p) { return (n_p / 2 + 1) * n_p; }   // upper bound on idx + 1

// JS > 1 (patch form): phase 2 runs on JS lanes per (element, local node), each with 1/JS of the node's NJ stored blocks —
// the 15-node element at JS = 2 on 512 threads keeps its 16-element patches (same LDS, same plan) at twice the waves per
// CU and half the accumulators per lane.
template <int NP, int NQ, bool GEO = false, int TPB = kBlock, int JS = 1> struct ElemCfg {
    static constexpr int maxpq = NP > NQ ? NP : NQ;
    static constexpr int T1 = TPB / JS;                 // lanes the (element, node) pairs of phase 2 may take
    static constexpr int EB0 = (T1 / maxpq) >= T1 / 4 ? T1 / 4 : ((T1 / maxpq) >= T1 / 8 ? T1 / 8 : (T1 / maxpq));
    // with the coordinate staging of GEO the big elements take fewer per workgroup, so that the LDS still admits
    // as many resident workgroups as without it (P2: 4 per CU, Q2: 3)
    // (the 15-node element: 16 instead of 17, so that two workgroups with their gather codes fit a CU's LDS)
    // TPB = 512 (patch form; P2's default): twice the elements per workgroup — a patch four runs high with runs as long as
    // before — at the same waves per CU (two workgroups of eight waves)
    static constexpr int S = T1 / kBlock > 0 ? T1 / kBlock : 1;
    static constexpr int EB = NP == 15 ? 16 * S : !GEO ? (NP == 6 && NQ == 7 && S == 2 ? 60 : EB0)
                                                       : (NP == 6 && NQ == 7) ? 28 * S : (NP == 8 && NQ == 9) ? (JS == 1 ? (T1 * 24) / kBlock : 24) : EB0;
    static constexpr int NQS = NQ | 1;                  // odd LDS stride: conflict-free ds_read_b64 over elements
    static constexpr int NPTS = EB * NQS;
    static constexpr int NJ = NP / 2 + 1;               // stored node-pair blocks (a, a+j mod NP) per local node
    static constexpr int NJH = (NJ + JS - 1) / JS;      // ... per lane of phase 2
    // LDS image in doubles.  Phases 1-2: dphi (2*NP rows), w*DS (6), w*S (3) per point, then the per-element node
    // coordinates / displacements and the reference-element tables.  Phase 3 (patch route) re-uses the same memory:
    // stored K_e blocks as two planes of 16-byte pieces (first rows, second rows: NJ*NP*EB double2 each — a lane's store and
    // its neighbour's are then 16 bytes apart, ds_write_b128 without bank conflicts; the interleaved 32-byte image of round 3
    // cost two LDS passes per store), force pairs (NP*EB x 2), the patch's gather codes (uint16).
    static constexpr int kPts = (2 * NP + 9) * NPTS;
    static constexpr int kXY = 2 * NP * EB;             // one double2 per (local node, element)
    static constexpr int kTab = 2 * NP * NQ + NQ + (NQ & 1);
    // skewed image (fep_host.h: patch_image_pos): odd element stride, one more slot per j for even NP
    static constexpr int EBP = EB | 1;
    static constexpr int PER = NP * EBP + ((NP & 1) ? 0 : 1);      // slots per j
    static constexpr int SLOTS = NJ * PER;                          // double2 per plane
    static constexpr int kKl = 4 * SLOTS, kFl = 2 * NP * EBP;
    static constexpr int kCodes = (NP * NP * EB + 3) / 4 + 1, kFcodes = (NP * EB + 3) / 4 + 1;    // doubles holding the uint16 codes
    static constexpr int kPhase3 = kKl + kFl + kCodes + kFcodes;
};

// Patch route (PATCH = true; fep_host.h, PatchPlan): the workgroup's EB elements are a patch.  Phase 2 leaves the stored
// K_e blocks and the force pairs in LDS instead of HBM; phase 3, one lane per item, sums the patch's contributions to one
// node-pair block (force: one node) in the fixed (e, a, b) order and writes either the finished CSR block / nodal force
// (every contribution inside the patch) or a partial for fixup_kernel.  No K_e round trip through HBM.
struct PatchArgs {
    const int32_t* pdesc;                               // 8 ints per patch (scalar loads)
    const int32_t* pel; const int32_t* pnodes;          // the patch's elements (ascending, -1 padded) and their nodes [a][local element]
    const uint2* items; const uint16_t* codes;
    const uint2* fitems; const uint16_t* fcodes;
    double* Pc; double* Pf;                             // partial blocks (4 doubles per slot) / partial forces (2)
    double* data; double* F;                            // CSR values / nodal force (either may be NULL)
#ifdef FEP_ABLATION
    unsigned long long* clk;                            // FEP_PHASE_CLK: 10 stamps per workgroup (thread 0), or NULL: [0..6] s_memtime (shader
                                                        // clock) at the phase boundaries, [8], [9] s_memrealtime (100 MHz) at start / end
#endif
};

// Phase stamps of the ablation build (thread 0 of every workgroup; the barriers keep the waves of a workgroup in step)
#ifdef FEP_ABLATION
#define FEP_STAMP_P(pa, p, i) do { if ((pa).clk && threadIdx.x == 0) { (pa).clk[(size_t)(p) * 12 + (i)] = __builtin_amdgcn_s_memtime(); \
        if ((i) == 0) (pa).clk[(size_t)(p) * 12 + 8] = __builtin_amdgcn_s_memrealtime();                                    \
        if ((i) == 6) (pa).clk[(size_t)(p) * 12 + 9] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define FEP_STAMP_P(pa, p, i) do { } while (0)
#endif

template <int NP, int NQ, bool FROM_U, bool GEO, bool PATCH = false, int TPB = kBlock, int JS = 1>
__global__ void __launch_bounds__(TPB, (NP == 15 && TPB == 512) ? 4 : 1)     // (15 nodes on 512 threads: <= 128 VGPRs, two workgroups per CU)
element_kernel(int64_t n_e, const int32_t* __restrict__ elem,
               const double* __restrict__ dphi1, const double* __restrict__ dphi2, const double* __restrict__ weight,
               // GEO: geometry recomputed from the coordinates (xy interleaved) and the reference-element tables
               const double* __restrict__ xy, const double* __restrict__ dh1, const double* __restrict__ dh2,
               const double* __restrict__ wf,
               // FROM_U inputs
               const double* __restrict__ U, E0 e0, double* __restrict__ ep,
               const double* __restrict__ shear, const double* __restrict__ bulk,
               const double* __restrict__ eta, const double* __restrict__ cc, MatU mu, int accept,
               // outputs of phase 1 (FROM_U) or inputs (!FROM_U)
               double* __restrict__ Eout, double* __restrict__ S, double* __restrict__ DS, uint8_t* __restrict__ indp,
               uint2* blk_counts,
               // outputs of phase 2 (kc_aos: all stored blocks of an element adjacent, see sym_block_index)
               double* __restrict__ Kc, double* __restrict__ fe, int kc_aos, PatchArgs pa) {
    using C = ElemCfg<NP, NQ, GEO, TPB, JS>;
    constexpr int EB = C::EB, NQS = C::NQS, NPTS = C::NPTS;
    constexpr int NJ = C::NJ;                        // stored node-pair blocks (a, a+j mod NP) per (element, local node)
    constexpr int NJH = C::NJH;                      // ... of which a lane of phase 2 computes NJH (all of them unless JS > 1)
    static_assert(PATCH || JS == 1, "the split of a node's blocks over JS lanes exists in the patch form only");
    constexpr int kP12 = C::kPts + (GEO ? C::kXY : 2) + (FROM_U ? C::kXY : 2) + (GEO ? C::kTab : 4);
    constexpr int kLds = (PATCH && C::kPhase3 > kP12) ? C::kPhase3 : kP12;                  // phases 1-2 | phase 3, same memory
    static_assert(EB * NQ <= TPB && EB * NP * JS <= TPB, "one pass per phase");
    __shared__ __attribute__((aligned(16))) double lds[kLds];
    double (*d1s)[NPTS] = reinterpret_cast<double (*)[NPTS]>(lds);
    double (*d2s)[NPTS] = reinterpret_cast<double (*)[NPTS]>(lds + NP * NPTS);
    double (*Ds)[NPTS] = reinterpret_cast<double (*)[NPTS]>(lds + 2 * NP * NPTS);
    double (*Ss)[NPTS] = reinterpret_cast<double (*)[NPTS]>(lds + (2 * NP + 6) * NPTS);
    double2 (*cxy)[EB] = reinterpret_cast<double2 (*)[EB]>(lds + C::kPts);                       // node coordinates per element
    double2 (*cu)[EB] = reinterpret_cast<double2 (*)[EB]>(lds + C::kPts + (GEO ? C::kXY : 2));   // node displacements
    double* t1 = lds + C::kPts + (GEO ? C::kXY : 2) + (FROM_U ? C::kXY : 2);
    double* t2 = t1 + (GEO ? NP * NQ : 1);
    double* tw = t2 + (GEO ? NP * NQ : 1);
    uint32_t* codes32 = reinterpret_cast<uint32_t*>(lds + C::kKl + C::kFl);              // phase 3: behind the blocks and force pairs
    uint32_t* fcodes32 = reinterpret_cast<uint32_t*>(lds + C::kKl + C::kFl + C::kCodes);

    const int t = threadIdx.x;
    const int64_t e0blk = (int64_t)blockIdx.x * EB;     // COO form: the workgroup's elements are e0blk .. e0blk + nel - 1
    const int64_t n_int = n_e * NQ;
    // patch form: the elements are listed in pel (any ids, ascending); pdesc: 8 ints per patch (uniform address: scalar loads)
    const int32_t* pd = PATCH ? pa.pdesc + (int64_t)blockIdx.x * 8 : nullptr;
    const int item_off = PATCH ? pd[0] : 0, n_items = PATCH ? pd[1] : 0, code_off = PATCH ? pd[2] : 0, n_codes = PATCH ? pd[3] : 0;
    const int fitem_off = PATCH ? pd[4] : 0, n_fitems = PATCH ? pd[5] : 0, fcode_off = PATCH ? pd[6] : 0, n_fcodes = PATCH ? pd[7] : 0;
    const int nel = PATCH ? n_fcodes / NP : (int)((n_e - e0blk) < EB ? (n_e - e0blk) : EB);

    FEP_STAMP_P(pa, blockIdx.x, 0);
    __shared__ int32_t pel_s[PATCH ? EB : 1];
    const int el1 = t / NQ, q1 = t - el1 * NQ;         // the lane's point in phase 1
    // Patch form, 6- and 15-node triangles: the point's previous plastic strain is fetched through a chain of its own (element
    // id -> ep) BESIDE the staging chain (node ids -> node data) instead of behind the barrier — one exposed round trip less
    // at the top of phase 1: P2 -2.6...-5 %, P4 -2.5 % in one session.  Not for the quadrilaterals: Q2 +5.7 %, Q1 +3.5 % with it
    // (all waves of the workgroup then sit at the barrier for both chains; behind the barrier each wave waits for its own
    // loads and the waves drift apart, which is what overlaps them) — profiles/r04_ablation.md, section 2b.
    constexpr bool HOIST = PATCH && FROM_U && (NP == 6 || NP == 15);
    double pre_p[4] = {0.0, 0.0, 0.0, 0.0};            // (HOIST) the plastic strain, fetched beside the staging
    if (PATCH) {
        // Every load of the staging is unconditional at a clamped index — a load under a per-lane condition is waited for on
        // the spot — and lanes outside a table's range drop what they fetched.
        const int ts = t < NP * EB ? t : NP * EB - 1;
        const int a = ts / EB, el = ts - a * EB;
        const int elc = el1 < nel ? el1 : nel - 1;
        const int32_t pe = pa.pel[(int64_t)blockIdx.x * EB + (t < EB ? t : EB - 1)];
        int32_t nd = 0, e_pt = 0;
        if (FROM_U || GEO) nd = pa.pnodes[((int64_t)blockIdx.x * NP + a) * EB + el];
        if (HOIST && ep) e_pt = pa.pel[(int64_t)blockIdx.x * EB + elc];
#ifdef FEP_ABLATION
        if (pa.clk) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); FEP_STAMP_P(pa, blockIdx.x, 10); }   // the ids have arrived
#endif
        double2 gxy = make_double2(0.0, 0.0), gu = make_double2(0.0, 0.0);
        if (GEO) gxy = *reinterpret_cast<const double2*>(xy + 2 * (int64_t)nd);
        if (FROM_U) gu = *reinterpret_cast<const double2*>(U + 2 * (int64_t)nd);
        if (HOIST && ep) {
            const unsigned kb = ((unsigned)e_pt * NQ + q1) * 8u;
            pre_p[0] = ld_row(ep, n_int, 0, kb); pre_p[1] = ld_row(ep, n_int, 1, kb); pre_p[2] = ld_row(ep, n_int, 2, kb); pre_p[3] = ld_row(ep, n_int, 3, kb);
        }
#ifdef FEP_ABLATION
        if (pa.clk) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); FEP_STAMP_P(pa, blockIdx.x, 11); }   // the node data (and the plastic strain) too
#endif
        if (t < EB) pel_s[t] = pe;
        if (t < NP * EB) {
            if (GEO) cxy[a][el] = gxy;
            if (FROM_U) cu[a][el] = gu;
        }
    } else if (FROM_U || GEO) {
        if (t < NP * EB) {
            const int a = t / EB, el = t - a * EB;
            const int64_t nd = el < nel ? elem[(int64_t)a * n_e + e0blk + el] : 0;
            if (GEO) cxy[a][el] = *reinterpret_cast<const double2*>(xy + 2 * nd);
            if (FROM_U) cu[a][el] = *reinterpret_cast<const double2*>(U + 2 * nd);
        }
    }
    if (GEO) {
        for (int i = t; i < NP * NQ; i += TPB) { t1[i] = dh1[i]; t2[i] = dh2[i]; }
        for (int i = t; i < NQ; i += TPB) tw[i] = wf[i];
    }
    if (FROM_U || GEO || PATCH) __syncthreads();
    FEP_STAMP_P(pa, blockIdx.x, 1);                                                  // node data staged

    // ---- phase 1 (one pass: EB * NQ <= TPB).  The point's operand loads are issued here, after the barrier (but for the
    // plastic strain of the triangles, HOIST above): hoisting all of them and the patch tables in front of it was measured
    // 7-15 % slower in round 3 — the waves of a workgroup then wait in step.
    int branch = 0;
    if (t < EB * NQ && el1 < nel) {
        const int el = el1, q = q1;
        const int64_t k = PATCH ? (int64_t)pel_s[el] * NQ + q : e0blk * NQ + t;
        const unsigned k32 = (unsigned)k, kb = k32 * 8u;             // byte offset of the point in a row (ld_row / st_row)
        const int li = el * NQS + q;
        double w;
        double ev[3] = {0.0, 0.0, 0.0};
        // Nodes in chunks of CH: gradients -> LDS image + strain sums (DP:1043).  All NP at once is 4*NP registers of
        // loads in flight per lane; the 15-node element takes 5 at a time (its 512-thread form must stay within 128 VGPRs).
        // The strain is written out in fused multiply-adds: which of two products the compiler fuses is its choice, and it
        // chose differently in the 256- and the 512-thread instantiation (point outputs differed in the last bit).
        constexpr int CH = (NP == 15 && TPB == 512) ? 5 : NP;       // (one round trip instead of three where the registers allow)
        if (GEO) {
            double g1[NP], g2[NP];
            double x[NP], y[NP];
#pragma unroll
            for (int a = 0; a < NP; ++a) { const double2 c = cxy[a][el]; x[a] = c.x; y[a] = c.y; }
            geometry_at_q<NP>(t1, t2, tw[q], q, NQ, x, y, g1, g2, w);
#pragma unroll
            for (int a = 0; a < NP; ++a) {
                d1s[a][li] = g1[a]; d2s[a][li] = g2[a];
                if (FROM_U) {
                    const double2 u = cu[a][el];
                    ev[0] = __builtin_fma(g1[a], u.x, ev[0]);
                    ev[1] = __builtin_fma(g2[a], u.y, ev[1]);
                    ev[2] += __builtin_fma(g1[a], u.y, g2[a] * u.x);
                }
            }
        } else {
            w = ld_row(weight, 0, 0, kb);
            // (a real loop for CH < NP: unrolled, the compiler hoists every chunk's loads to the top whatever stands between them)
#pragma unroll 1
            for (int a0 = 0; a0 < NP; a0 += CH) {
                double g1[CH], g2[CH];
#pragma unroll
                for (int a = 0; a < CH; ++a) { g1[a] = ld_row(dphi1, n_int, a0 + a, kb); g2[a] = ld_row(dphi2, n_int, a0 + a, kb); }
#pragma unroll
                for (int a = 0; a < CH; ++a) {
                    d1s[a0 + a][li] = g1[a]; d2s[a0 + a][li] = g2[a];
                    if (FROM_U) {
                        const double2 u = cu[a0 + a][el];
                        ev[0] = __builtin_fma(g1[a], u.x, ev[0]);
                        ev[1] = __builtin_fma(g2[a], u.y, ev[1]);
                        ev[2] += __builtin_fma(g1[a], u.y, g2[a] * u.x);
                    }
                }
            }
        }
        double s[4], d[6];
        if (FROM_U) {
            double p[4] = {0.0, 0.0, 0.0, 0.0};
            if (ep) {
                if (HOIST) { p[0] = pre_p[0]; p[1] = pre_p[1]; p[2] = pre_p[2]; p[3] = pre_p[3]; }
                else { p[0] = ld_row(ep, n_int, 0, kb); p[1] = ld_row(ep, n_int, 1, kb); p[2] = ld_row(ep, n_int, 2, kb); p[3] = ld_row(ep, n_int, 3, kb); }
            }
            const double m_sh = mu.on ? mu.shear : ld_row(shear, 0, 0, kb), m_bu = mu.on ? mu.bulk : ld_row(bulk, 0, 0, kb);
            const double m_eta = mu.on ? mu.eta : ld_row(eta, 0, 0, kb), m_c = mu.on ? mu.c : ld_row(cc, 0, 0, kb);
            branch = dp_return_map(ev, e0.v, p, m_sh, m_bu, m_eta, m_c, accept != 0, s, d);
            store_point_off(kb, k32, n_int, s, d, branch, S, DS, indp);
            if (Eout) { st_row(Eout, n_int, 0, kb, ev[0]); st_row(Eout, n_int, 1, kb, ev[1]); st_row(Eout, n_int, 2, kb, ev[2]); }
            if (accept && ep && branch) {
                st_row(ep, n_int, 0, kb, p[0]); st_row(ep, n_int, 1, kb, p[1]); st_row(ep, n_int, 2, kb, p[2]); st_row(ep, n_int, 3, kb, p[3]);
            }
        } else {
            if (DS) {
                d[0] = ld_row(DS, n_int, 0, kb); d[1] = ld_row(DS, n_int, 1, kb); d[2] = ld_row(DS, n_int, 2, kb);
                d[3] = ld_row(DS, n_int, 4, kb); d[4] = ld_row(DS, n_int, 5, kb); d[5] = ld_row(DS, n_int, 8, kb);
            } else { d[0] = d[1] = d[2] = d[3] = d[4] = d[5] = 0.0; }
            if (S) { s[0] = ld_row(S, n_int, 0, kb); s[1] = ld_row(S, n_int, 1, kb); s[2] = ld_row(S, n_int, 2, kb); } else { s[0] = s[1] = s[2] = 0.0; }
        }
#pragma unroll
        for (int m = 0; m < 6; ++m) Ds[m][li] = w * d[m];                        // vD = w*ds, DP:1047
        Ss[0][li] = w * s[0]; Ss[1][li] = w * s[1]; Ss[2][li] = w * s[2];        // DP:1058
    }
    if (FROM_U) count_branches(branch, nullptr, blk_counts);
    lds_barrier();                                                     // (the point outputs' stores drain behind phase 2)
    FEP_STAMP_P(pa, blockIdx.x, 2);                                                  // phase 1 done

    // ---- phase 2 (one pass: NP * EB * JS <= TPB) ----------------------------------
    // lane (h, a, el): the blocks j = h*NJH .. of the (element, node) pair; h = 0 also sums the force pair
    double kk[NJH][4];
    double f0 = 0.0, f1 = 0.0;
    const int ha = t / EB, el = t - ha * EB;
    const int h = JS == 1 ? 0 : ha / NP, a = JS == 1 ? ha : ha - h * NP;
    const int j0 = h * NJH;
    const bool lane2 = t < NP * EB * JS && el < nel;
    if (lane2) {
#pragma unroll
        for (int j = 0; j < NJH; ++j) { kk[j][0] = 0.0; kk[j][1] = 0.0; kk[j][2] = 0.0; kk[j][3] = 0.0; }
#pragma unroll 1
        for (int q = 0; q < NQ; ++q) {
            const int li = el * NQS + q;
            const double D00 = Ds[0][li], D01 = Ds[1][li], D02 = Ds[2][li];
            const double D11 = Ds[3][li], D12 = Ds[4][li], D22 = Ds[5][li];
            const double a1 = d1s[a][li], a2 = d2s[a][li];
            // rows of B_a^T D:  r0 = (a1,0,a2) D,  r1 = (0,a2,a1) D
            const double r00 = a1 * D00 + a2 * D02, r01 = a1 * D01 + a2 * D12, r02 = a1 * D02 + a2 * D22;
            const double r10 = a2 * D01 + a1 * D02, r11 = a2 * D11 + a1 * D12, r12 = a2 * D12 + a1 * D22;
#if FEP_P2_FMA
            // every product goes into its accumulator with one fused multiply-add (two per entry and point instead of
            // multiply, multiply-add, add)
            if (JS == 1 || h == 0) {
                f0 = __builtin_fma(a1, Ss[0][li], __builtin_fma(a2, Ss[2][li], f0));
                f1 = __builtin_fma(a2, Ss[1][li], __builtin_fma(a1, Ss[2][li], f1));
            }
#pragma unroll
            for (int j = 0; j < NJH; ++j) {
                const int jj = j0 + j;
                const int b = a + jj >= NP ? a + jj - NP : a + jj;
                const double b1 = d1s[b][li], b2 = d2s[b][li];
                kk[j][0] = __builtin_fma(r00, b1, __builtin_fma(r02, b2, kk[j][0]));
                kk[j][1] = __builtin_fma(r01, b2, __builtin_fma(r02, b1, kk[j][1]));
                kk[j][2] = __builtin_fma(r10, b1, __builtin_fma(r12, b2, kk[j][2]));
                kk[j][3] = __builtin_fma(r11, b2, __builtin_fma(r12, b1, kk[j][3]));
            }
#else
            if (JS == 1 || h == 0) {
                f0 += a1 * Ss[0][li] + a2 * Ss[2][li];
                f1 += a2 * Ss[1][li] + a1 * Ss[2][li];
            }
#pragma unroll
            for (int j = 0; j < NJH; ++j) {
                const int jj = j0 + j;                                       // (jj >= NJ on the last lane group of an uneven
                const int b = a + jj >= NP ? a + jj - NP : a + jj;           //  split: computed on a valid node, never stored)
                const double b1 = d1s[b][li], b2 = d2s[b][li];
                kk[j][0] += r00 * b1 + r02 * b2;
                kk[j][1] += r01 * b2 + r02 * b1;
                kk[j][2] += r10 * b1 + r12 * b2;
                kk[j][3] += r11 * b2 + r12 * b1;
            }
#endif
        }
    }
    if (!PATCH) {
        if (!lane2) return;
        const int64_t e = e0blk + el;
        if (Kc) {
#pragma unroll
            for (int j = 0; j < NJH; ++j) {                                  // (JS == 1 here: NJH == NJ)
                if (NP % 2 == 0 && j == NP / 2 && a >= NP / 2) continue;     // held by the lane of node a - NP/2
                double2* dst = reinterpret_cast<double2*>(Kc + (kc_aos ? (e * (NJ * NP) + a * NJ + j) : ((int64_t)(j * NP + a) * n_e + e)) * 4);
                dst[0] = make_double2(kk[j][0], kk[j][1]);
                dst[1] = make_double2(kk[j][2], kk[j][3]);
            }
        }
        if (fe) *reinterpret_cast<double2*>(fe + ((int64_t)a * n_e + e) * 2) = make_double2(f0, f1);
        return;
    }

    // ---- phase 3 (patch route) --------------------------------------------------------
    // stored block (j, a, el): first row at Kr0[pos], second row at Kr1[pos], pos = j*PER + a*EBP + el (two planes, skewed: ElemCfg)
    double2* Kr0 = reinterpret_cast<double2*>(lds);
    double2* Kr1 = Kr0 + C::SLOTS;
    double2* fl2 = reinterpret_cast<double2*>(lds + C::kKl);           // force pair (a, el) at [a * EBP + el]
    const uint16_t* codes_l = reinterpret_cast<const uint16_t*>(codes32);
    const uint16_t* fcodes_l = reinterpret_cast<const uint16_t*>(fcodes32);
    // Phase 3's tables, fetched behind phase 2 — EVERY item descriptor of the patch (at most NP*NP*EB of them), so that phase 3
    // issues no load between its stores: a load there is waited for with everything older than it (vmcnt counts in order),
    // i.e. with the stores of the round before — the 15-node element has four such rounds.  Not in FRONT of phase 2: the
    // vector-memory queue is in order too, and loads issued right behind phase 1's burst of point-output stores stall at
    // issue until those have drained (measured: phase 1 11.7 k -> 20.3 k clocks for P2, step +14 %, profiles/r04_ablation.md)
    constexpr int MAXIT = (NP * NP * EB + TPB - 1) / TPB, FIT = (NP * EB + TPB - 1) / TPB;
    constexpr int IT = 4;                               // items a lane advances together (independent LDS chains)
    constexpr int NG = (MAXIT + IT - 1) / IT;
    const uint2* its = pa.items + item_off;
    uint2 dsc[NG * IT], fdsc[FIT];
    constexpr int CWPT = (NP * NP * EB / 2 + TPB) / TPB, FWPT = (NP * EB / 2 + TPB) / TPB;
    uint32_t cpre[CWPT], fpre[FWPT];
    if (PATCH) {
        // UNCONDITIONAL loads at clamped indices (the tables are padded by one entry; a lane past the end re-reads the last
        // item and never uses it): a load under a per-lane condition is waited for on the spot — fifteen of them for the
        // 15-node element were fifteen round trips in a row (phase-clock: 20 k of the workgroup's 58 k cycles)
#pragma unroll
        for (int u = 0; u < NG * IT; ++u) dsc[u] = make_uint2(0u, 0u);
#pragma unroll
        for (int u = 0; u < FIT; ++u) fdsc[u] = make_uint2(0u, 0u);
#pragma unroll
        for (int r = 0; r < CWPT; ++r) cpre[r] = 0u;
#pragma unroll
        for (int r = 0; r < FWPT; ++r) fpre[r] = 0u;
        if (pa.data) {                                  // (uniform)
            const int last = n_items > 0 ? n_items - 1 : 0, lastc = n_codes > 0 ? (n_codes - 1) / 2 : 0;
            const uint32_t* cg = reinterpret_cast<const uint32_t*>(pa.codes + code_off);     // two codes per 32-bit word (every
#pragma unroll                                                                               // patch's codes start at an even offset)
            for (int u = 0; u < MAXIT; ++u) { const int i = u * TPB + t; dsc[u] = its[i < last ? i : last]; }
#pragma unroll
            for (int r = 0; r < CWPT; ++r) { const int i = r * TPB + t; cpre[r] = cg[i < lastc ? i : lastc]; }
        }
        if (pa.F) {
            const int last = n_fitems > 0 ? n_fitems - 1 : 0, lastc = n_fcodes > 0 ? (n_fcodes - 1) / 2 : 0;
            const uint32_t* fg = reinterpret_cast<const uint32_t*>(pa.fcodes + fcode_off);
#pragma unroll
            for (int u = 0; u < FIT; ++u) { const int i = u * TPB + t; fdsc[u] = pa.fitems[fitem_off + (i < last ? i : last)]; }
#pragma unroll
            for (int r = 0; r < FWPT; ++r) { const int i = r * TPB + t; fpre[r] = fg[i < lastc ? i : lastc]; }
        }
    }
    FEP_STAMP_P(pa, blockIdx.x, 3);                                                  // phase 2 done (thread 0's wave)
    lds_barrier();                                                     // every lane is done reading the phase-2 operands
    FEP_STAMP_P(pa, blockIdx.x, 4);
    if (lane2) {
#pragma unroll
        for (int j = 0; j < NJH; ++j) {
            const int jj = j0 + j;
            if (JS > 1 && jj >= NJ) continue;
            if (NP % 2 == 0 && jj == NP / 2 && a >= NP / 2) continue;
            const int pos = jj * C::PER + a * C::EBP + el;
            Kr0[pos] = make_double2(kk[j][0], kk[j][1]);
            Kr1[pos] = make_double2(kk[j][2], kk[j][3]);
        }
        if (JS == 1 || h == 0) fl2[a * C::EBP + el] = make_double2(f0, f1);
    }
#pragma unroll
    for (int r = 0; r < CWPT; ++r) { const int i = r * TPB + t; if (i < (n_codes + 1) / 2) codes32[i] = cpre[r]; }
#pragma unroll
    for (int r = 0; r < FWPT; ++r) { const int i = r * TPB + t; if (i < (n_fcodes + 1) / 2) fcodes32[i] = fpre[r]; }
    lds_barrier();
    FEP_STAMP_P(pa, blockIdx.x, 5);                                                  // K_e image and codes in LDS
    if (pa.data) {
        double2* data2 = reinterpret_cast<double2*>(pa.data);
        double2* Pc2 = reinterpret_cast<double2*>(pa.Pc);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g * IT * TPB >= n_items) break;                        // (uniform)
            // The lane's IT items advance together, contribution k of each in one round: IT independent chains
            // (code -> the block's two 16-byte rows) in flight instead of one — the loop is a chain of LDS round trips.
            // A lane's item that has run out re-reads its last contribution and drops it: every sum is the same
            // sequence of additions as a per-item loop (0 + c_0 + c_1 + ...).
            int off[IT], cnt[IT], kmax = 0;
            double a00[IT], a01[IT], a10[IT], a11[IT];
#pragma unroll
            for (int u = 0; u < IT; ++u) {
                const bool on = (g * IT + u) * TPB + t < n_items;
                off[u] = (int)(dsc[g * IT + u].x & 8191u);
                cnt[u] = on ? (int)((dsc[g * IT + u].x >> 13) & 63u) + 1 : 0;
                kmax = cnt[u] > kmax ? cnt[u] : kmax;
                a00[u] = 0.0; a01[u] = 0.0; a10[u] = 0.0; a11[u] = 0.0;
            }
            for (int k = 0; k < kmax; ++k) {
                unsigned code[IT];
                double2 r0[IT], r1[IT];
#pragma unroll
                for (int u = 0; u < IT; ++u) { const int kc = k < cnt[u] ? k : (cnt[u] > 0 ? cnt[u] - 1 : 0); code[u] = codes_l[off[u] + kc]; }
#pragma unroll
                for (int u = 0; u < IT; ++u) { r0[u] = Kr0[code[u] >> 1]; r1[u] = Kr1[code[u] >> 1]; }
#pragma unroll
                for (int u = 0; u < IT; ++u) {
                    if (k < cnt[u]) {
                        const bool tr = code[u] & 1u;
                        a00[u] += r0[u].x; a01[u] += tr ? r1[u].x : r0[u].y; a10[u] += tr ? r0[u].y : r1[u].x; a11[u] += r1[u].y;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < IT; ++u) {
                if (cnt[u] == 0) continue;
                const uint32_t x = dsc[g * IT + u].x, y = dsc[g * IT + u].y;
                if (x >> 31) {                                         // partial: slot y of the side buffer
                    Pc2[2 * (int64_t)y] = make_double2(a00[u], a01[u]);
                    Pc2[2 * (int64_t)y + 1] = make_double2(a10[u], a11[u]);
                } else {                                               // finished CSR block: its two rows
                    const int deg = (int)((x >> 19) & 4095u);
                    data2[y] = make_double2(a00[u], a01[u]);
                    data2[(int64_t)y + deg] = make_double2(a10[u], a11[u]);
                }
            }
        }
    }
    if (pa.F) {
        double2* F2 = reinterpret_cast<double2*>(pa.F);
        double2* Pf2 = reinterpret_cast<double2*>(pa.Pf);
#pragma unroll
        for (int u = 0; u < FIT; ++u) {
            if (u * TPB + t >= n_fitems) continue;
            const uint2 d = fdsc[u];
            const int off = (int)(d.x & 8191u), cnt = (int)((d.x >> 13) & 63u) + 1;
            double g0 = 0.0, g1 = 0.0;
            for (int k = 0; k < cnt; ++k) { const double2 v = fl2[fcodes_l[off + k]]; g0 += v.x; g1 += v.y; }
            if (d.x >> 31) Pf2[d.y] = make_double2(g0, g1); else F2[d.y] = make_double2(g0, g1);
        }
    }
    FEP_STAMP_P(pa, blockIdx.x, 6);                                                  // thread 0's wave has issued its last store
}

// Second kernel of the patch route: one lane per UPPER open block (a node pair on a patch boundary, row node <= column
// node) adds the patches' partials in ascending patch order and writes the CSR block and its transposed mirror; the
// workgroups behind those do the same for the open nodes' forces.
// With counts_out the first workgroup sums the element kernel's branch counters on the side.
__global__ void __launch_bounds__(kBlock)
fixup_kernel(int nb_k, int64_t n_open, const uint4* __restrict__ fix, const uint2* __restrict__ fixT,
             int64_t n_fopen, const uint4* __restrict__ ffix,
             const int32_t* __restrict__ plist, const double* __restrict__ Pc, const double* __restrict__ Pf,
             double* __restrict__ data, double* __restrict__ F,
             int n_count_blocks, const uint2* __restrict__ blk_counts, unsigned long long* __restrict__ counts_out) {
    const int g = counts_out != nullptr ? (int)blockIdx.x - 1 : (int)blockIdx.x;
    if (g < 0) { sum_block_counts(n_count_blocks, blk_counts, counts_out); return; }
    if (g < nb_k) {
        const int64_t i = (int64_t)g * kBlock + threadIdx.x;
        if (i >= n_open) return;
        const uint4 f = fix[i];
        const int cnt = (int)(f.y >> 16), deg = (int)(f.y & 0xffffu);
        const double2* P2 = reinterpret_cast<const double2*>(Pc);
        double a00, a01, a10, a11;
        const uint2 ft = fixT[i];                                      // (issued with the partials, not behind them)
        if (cnt <= 2) {
            // both partials go out together: with the second one under `cnt == 2` its load was issued only after the first
            // had been waited for — three dependent levels (descriptor, first partial, second partial) instead of two.  A
            // block with one contributor re-reads its first partial and drops it.
            const int64_t z = (int64_t)f.z, w = cnt == 2 ? (int64_t)f.w : (int64_t)f.z;
            const double2 p0 = P2[2 * z], p1 = P2[2 * z + 1];
            const double2 q0 = P2[2 * w], q1 = P2[2 * w + 1];
            a00 = 0.0 + p0.x; a01 = 0.0 + p0.y; a10 = 0.0 + p1.x; a11 = 0.0 + p1.y;
            if (cnt == 2) { a00 += q0.x; a01 += q0.y; a10 += q1.x; a11 += q1.y; }
        } else {
            a00 = a01 = a10 = a11 = 0.0;
            for (int k = 0; k < cnt; ++k) {
                const int64_t s = plist[(int64_t)f.w + k];
                const double2 p0 = P2[2 * s], p1 = P2[2 * s + 1];
                a00 += p0.x; a01 += p0.y; a10 += p1.x; a11 += p1.y;
            }
        }
        double2* data2 = reinterpret_cast<double2*>(data);
        data2[f.x] = make_double2(a00, a01);
        data2[(int64_t)f.x + deg] = make_double2(a10, a11);
        if (ft.x != 0xffffffffu) {                      // the mirror block (column node, row node): the transpose
            data2[ft.x] = make_double2(a00, a10);
            data2[(int64_t)ft.x + ft.y] = make_double2(a01, a11);
        }
        return;
    }
    const int64_t i = (int64_t)(g - nb_k) * kBlock + threadIdx.x;
    if (i >= n_fopen) return;
    const uint4 f = ffix[i];
    const int cnt = (int)f.y;
    const double2* P2 = reinterpret_cast<const double2*>(Pf);
    double g0 = 0.0, g1 = 0.0;
    if (cnt == 2) {
        const double2 p = P2[f.z], q = P2[f.w];
        g0 = (0.0 + p.x) + q.x; g1 = (0.0 + p.y) + q.y;
    } else {
        for (int k = 0; k < cnt; ++k) { const double2 p = P2[plist[(int64_t)f.w + k]]; g0 += p.x; g1 += p.y; }
    }
    reinterpret_cast<double2*>(F)[f.x] = make_double2(g0, g1);
}

// ---------------------------------------------------------------------------------------
// Numeric COO -> CSR phase: one lane per node-pair block (row node n, neighbour slot s).
// Deterministic: contributions are summed in the fixed order of `perm` (no atomics).
//   segptr[n_blk+1], perm[...] = 2*(sym_block_index(a,b)*n_e + e) + transposed,
//   meta[blk] = (deg(n) << 16) | (diag << 15) | s
//   CSR data: row 2n+i starts at 4*nptr[n] + i*2*deg, entry (slot s, comp j) at + 2s + j.
// ---------------------------------------------------------------------------------------
// Internal force: one lane per node; sums the element pairs fe[(a*n_e+e)*2 + i] in incidence order.
__device__ __forceinline__ void force_rows(int64_t n, int64_t n_n, const int32_t* __restrict__ iptr,
                                           const int32_t* __restrict__ ilist, const double* __restrict__ fe,
                                           double* __restrict__ F) {
    if (n >= n_n) return;
    double f0 = 0.0, f1 = 0.0;
    for (int32_t t = iptr[n]; t < iptr[n + 1]; ++t) {
        const double2 v = *reinterpret_cast<const double2*>(fe + (int64_t)ilist[t] * 2);
        f0 += v.x; f1 += v.y;
    }
    *reinterpret_cast<double2*>(F + 2 * n) = make_double2(f0, f1);
}

__global__ void __launch_bounds__(kBlock)
force_reduce_kernel(int64_t n_n, const int32_t* __restrict__ iptr, const int32_t* __restrict__ ilist,
                    const double* __restrict__ fe, double* __restrict__ F) {
    force_rows((int64_t)blockIdx.x * kBlock + threadIdx.x, n_n, iptr, ilist, fe, F);
}

// What the reduce kernels take to do the force gather in the SAME launch: the workgroups past the last tile take 256
// nodes each (F == nullptr: no such workgroups).  One launch and one tail instead of two.
struct ForceArgs { int64_t n_n; const int32_t* iptr; const int32_t* ilist; const double* fe; double* F; };

// G = gathers in flight per lane: the lane's contributions are fetched G at a time — first their G addresses
// (independent loads: the segment bounds are known), then the G 32-byte blocks, then summed in list order (the
// summation order does not depend on G).  The kernel is a chain of dependent gathers (86 % of its wave cycles sit in
// s_waitcnt with every wave slot taken): rounds per lane, not bytes, set its time.
template <int G>
__global__ void __launch_bounds__(kBlock)
csr_reduce_kernel(int n_tiles, const int32_t* __restrict__ tstart, const int32_t* __restrict__ segptr,
                  const int32_t* __restrict__ perm, const uint32_t* __restrict__ meta,
                  const double* __restrict__ Kc, double* __restrict__ data,
                  int n_count_blocks, const uint2* __restrict__ blk_counts, unsigned long long* __restrict__ counts_out,
                  ForceArgs fa) {
    // one workgroup = one tile of whole nodes, blocks [tstart[g], tstart[g+1]) (<= kBlock): its 4*nb CSR values
    // are one contiguous range, staged in LDS and written as full consecutive lines (a lane's two 16-byte pieces
    // belong to two different rows; direct stores fill every line in two half passes)
    __shared__ double2 out2[2 * kBlock];
    // with counts_out the FIRST workgroup sums the branch counters of the element kernel on the side (first, so that
    // its serial chain runs under the tiles, not after them)
    const int g = counts_out != nullptr ? (int)blockIdx.x - 1 : (int)blockIdx.x;
    if (g < 0) { sum_block_counts(n_count_blocks, blk_counts, counts_out); return; }
    if (g >= n_tiles) { if (fa.F) force_rows((int64_t)(g - n_tiles) * kBlock + threadIdx.x, fa.n_n, fa.iptr, fa.ilist, fa.fe, fa.F); return; }
    const int64_t sb0 = tstart[g];
    const int nb = tstart[g + 1] - (int)sb0;
    const int64_t sb = sb0 + threadIdx.x;
    if ((int)threadIdx.x < nb) {
        const int32_t beg = segptr[sb], end = segptr[sb + 1];
        const uint32_t m = meta[sb];
        double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
        for (int32_t t = beg; t < end; t += G) {
            int32_t pv[G];
#pragma unroll
            for (int k = 0; k < G; ++k) pv[k] = t + k < end ? perm[t + k] : -1;     // (stored block index) * 2 + transposed
            double2 r0[G], r1[G];
#pragma unroll
            for (int k = 0; k < G; ++k) {
                if (pv[k] >= 0) {
                    const double2* src = reinterpret_cast<const double2*>(Kc + (int64_t)(pv[k] >> 1) * 4);
                    r0[k] = src[0]; r1[k] = src[1];
                }
            }
#pragma unroll
            for (int k = 0; k < G; ++k) {
                if (pv[k] >= 0) {
                    const bool tr = pv[k] & 1;
                    a00 += r0[k].x; a01 += tr ? r1[k].x : r0[k].y; a10 += tr ? r0[k].y : r1[k].x; a11 += r1[k].y;
                }
            }
        }
        const int sl = (int)(m & 0x7fffu), deg = (int)(m >> 16);
        const int rel = 2 * (int)threadIdx.x - sl;          // (CSR position - 4*sb0) / 2
        out2[rel] = make_double2(a00, a01);
        out2[rel + deg] = make_double2(a10, a11);
    }
    __syncthreads();
    double2* dst = reinterpret_cast<double2*>(data + 4 * sb0);
    for (int i = threadIdx.x; i < 2 * nb; i += kBlock) dst[i] = out2[i];
}

// The same reduce with fewer vector-memory instructions per lane (its time follows their count): ONE 8-byte descriptor
// per block (x: first contribution, y: count:8 | degree:12 | slot:12) instead of segptr x2 + meta, and the lane's
// contribution addresses four at a time with one 16-byte load (`perm` is padded by 4 entries; the load needs 4-byte
// alignment only).  Same adds in the same order.
struct __attribute__((packed, aligned(4))) PermQuad { int32_t v[4]; };

__global__ void __launch_bounds__(kBlock)
csr_reduce_pk_kernel(int n_tiles, const int32_t* __restrict__ tstart, const uint2* __restrict__ pkc,
                     const int32_t* __restrict__ perm, const double* __restrict__ Kc, double* __restrict__ data,
                     int n_count_blocks, const uint2* __restrict__ blk_counts, unsigned long long* __restrict__ counts_out,
                     ForceArgs fa) {
    __shared__ double2 out2[2 * kBlock];
    const int g = counts_out != nullptr ? (int)blockIdx.x - 1 : (int)blockIdx.x;
    if (g < 0) { sum_block_counts(n_count_blocks, blk_counts, counts_out); return; }
    if (g >= n_tiles) { if (fa.F) force_rows((int64_t)(g - n_tiles) * kBlock + threadIdx.x, fa.n_n, fa.iptr, fa.ilist, fa.fe, fa.F); return; }
    const int64_t sb0 = tstart[g];
    const int nb = tstart[g + 1] - (int)sb0;
    if ((int)threadIdx.x < nb) {
        const uint2 d = pkc[sb0 + threadIdx.x];
        const int32_t* pl = perm + d.x;
        const int len = (int)(d.y & 255u), deg = (int)((d.y >> 8) & 4095u), sl = (int)(d.y >> 20);
        double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
        for (int t = 0; t < len; t += 4) {
            const PermQuad q = *reinterpret_cast<const PermQuad*>(pl + t);
            double2 r0[4], r1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (t + k < len) {
                    const double2* src = reinterpret_cast<const double2*>(Kc + (int64_t)(q.v[k] >> 1) * 4);
                    r0[k] = src[0]; r1[k] = src[1];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (t + k < len) {
                    const bool tr = q.v[k] & 1;
                    a00 += r0[k].x; a01 += tr ? r1[k].x : r0[k].y; a10 += tr ? r0[k].y : r1[k].x; a11 += r1[k].y;
                }
            }
        }
        const int rel = 2 * (int)threadIdx.x - sl;
        out2[rel] = make_double2(a00, a01);
        out2[rel + deg] = make_double2(a10, a11);
    }
    __syncthreads();
    double2* dst = reinterpret_cast<double2*>(data + 4 * sb0);
    for (int i = threadIdx.x; i < 2 * nb; i += kBlock) dst[i] = out2[i];
}



// transform (DP:760-816): value at a node = weighted mean of the values at the integration points of its
// elements, weights = quadrature weight * |det J|.  One thread per node over the incidence lists.
__global__ void __launch_bounds__(kBlock)
nodal_average_kernel(int64_t n_n, int64_t n_e, int n_q, const int32_t* __restrict__ iptr,
                     const int32_t* __restrict__ ilist, const double* __restrict__ weight,
                     const double* __restrict__ q_int, double* __restrict__ q_node) {
    const int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (n >= n_n) return;
    double f1 = 0.0, f2 = 0.0;
    for (int32_t t = iptr[n]; t < iptr[n + 1]; ++t) {
        const int64_t e = (int64_t)ilist[t] % n_e;
        for (int q = 0; q < n_q; ++q) {
            const double w = weight[e * n_q + q];
            f1 += w * q_int[e * n_q + q];
            f2 += w;
        }
    }
    q_node[n] = f1 / f2;
}

// P1 geometry recomputed from the node coordinates (48 bytes gathered through L2 instead of a 64-byte
// record streamed from HBM).  Same operations, same order, no FMA contraction as geometry_kernel, hence
// bit-identical dphi / weight (DP:530-546, 585).  `tab` = the P1 reference-element tables.
struct P1Tab { double h1[3], h2[3], wf; };

__device__ __forceinline__ void p1_geometry(const P1Tab& tab, const double2 c0, const double2 c1, const double2 c2,
                                            double d1[3], double d2[3], double& w) {
#pragma clang fp contract(off)
    const double x[3] = {c0.x, c1.x, c2.x}, y[3] = {c0.y, c1.y, c2.y};
    double j11 = 0.0, j12 = 0.0, j21 = 0.0, j22 = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        j11 = j11 + x[a] * tab.h1[a]; j12 = j12 + y[a] * tab.h1[a];
        j21 = j21 + x[a] * tab.h2[a]; j22 = j22 + y[a] * tab.h2[a];
    }
    const double det = j11 * j22 - j12 * j21;
    const double i11 = j22 / det, i12 = -j12 / det, i21 = -j21 / det, i22 = j11 / det;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        d1[a] = i11 * tab.h1[a] + i12 * tab.h2[a];
        d2[a] = i21 * tab.h1[a] + i22 * tab.h2[a];
    }
    w = fabs(det) * tab.wf;
}

// ---------------------------------------------------------------------------------------
// P1 fast path (3-node triangle, 1 integration point): no element-matrix round trip through HBM.
//
//   p1_point_kernel   one lane per element: geometry from the coordinates, strain from U (a1), return map (a2);
//                     writes s, ds, ind_p.
//   p1_node_kernel    one lane per node-pair block of the CSR pattern: gathers, for every element that
//                     contributes to the block, its tangent (6 values of ds), weight and the two nodes'
//                     dphi, forms the 2x2 block  w * B_a^T DS B_b  on the fly and sums in the fixed
//                     order of `perm2` (a3, a4); the lane that owns the diagonal block of a node also
//                     sums the node's internal force  w * B_a^T s  (a5).  No atomics, no COO buffer.
//
//   geo[e*6 + {0,1}] = dphi_1 of nodes 0, 1, {2,3} = dphi_2, [4] = weight (48-byte record; node 2 = -(0 + 1))
//   perm2[t] = e*16 + a*4 + b ;  meta = (deg << 16) | (diag << 15) | slot
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
p1_point_kernel(int64_t n_e, const int32_t* __restrict__ elem, const double* __restrict__ xy, P1Tab tab,
                const double* __restrict__ U, E0 e0, double* __restrict__ ep,
                const double* __restrict__ shear, const double* __restrict__ bulk,
                const double* __restrict__ eta, const double* __restrict__ cc, MatU mu, int accept,
                double* __restrict__ Eout, double* __restrict__ S, double* __restrict__ DS,
                uint8_t* __restrict__ indp, uint2* blk_counts) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int branch = 0;
    if (e < n_e) {
        const int64_t n0 = elem[e], n1 = elem[n_e + e], n2 = elem[2 * n_e + e];
        const double2 c0 = *reinterpret_cast<const double2*>(xy + 2 * n0);
        const double2 c1 = *reinterpret_cast<const double2*>(xy + 2 * n1);
        const double2 c2 = *reinterpret_cast<const double2*>(xy + 2 * n2);
        const double2 u0 = *reinterpret_cast<const double2*>(U + 2 * n0);
        const double2 u1 = *reinterpret_cast<const double2*>(U + 2 * n1);
        const double2 u2 = *reinterpret_cast<const double2*>(U + 2 * n2);
        double d1[3], d2[3], w;
        p1_geometry(tab, c0, c1, c2, d1, d2, w);
        double ev[3];                                            // DP:1043, local node order
        ev[0] = d1[0] * u0.x + d1[1] * u1.x + d1[2] * u2.x;
        ev[1] = d2[0] * u0.y + d2[1] * u1.y + d2[2] * u2.y;
        ev[2] = (d2[0] * u0.x + d1[0] * u0.y) + (d2[1] * u1.x + d1[1] * u1.y) + (d2[2] * u2.x + d1[2] * u2.y);
        double p[4] = {0.0, 0.0, 0.0, 0.0};
        if (ep) { p[0] = ep[e]; p[1] = ep[n_e + e]; p[2] = ep[2 * n_e + e]; p[3] = ep[3 * n_e + e]; }
        double s[4], d[6];
        const double m_sh = mu.on ? mu.shear : shear[e], m_bu = mu.on ? mu.bulk : bulk[e];
        const double m_eta = mu.on ? mu.eta : eta[e], m_c = mu.on ? mu.c : cc[e];
        branch = dp_return_map(ev, e0.v, p, m_sh, m_bu, m_eta, m_c, accept != 0, s, d);
        store_point(e, n_e, s, d, branch, S, DS, indp);
        if (Eout) { Eout[e] = ev[0]; Eout[n_e + e] = ev[1]; Eout[2 * n_e + e] = ev[2]; }
        if (accept && ep && branch) { ep[e] = p[0]; ep[n_e + e] = p[1]; ep[2 * n_e + e] = p[2]; ep[3 * n_e + e] = p[3]; }
    }
    count_branches(branch, nullptr, blk_counts);
}

__global__ void __launch_bounds__(kBlock)
p1_node_kernel(int64_t n_blk, int64_t n_e, const int32_t* __restrict__ segptr, const int32_t* __restrict__ perm2,
               const uint32_t* __restrict__ meta, const int32_t* __restrict__ ncol,
               const double* __restrict__ geo, const double* __restrict__ DS, const double* __restrict__ S,
               double* __restrict__ data, double* __restrict__ F) {
    const int64_t sb = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (sb >= n_blk) return;
    const int32_t beg = segptr[sb], end = segptr[sb + 1];
    const uint32_t m = meta[sb];
    const bool diag = (m >> 15) & 1u;
    const bool want_f = diag && F != nullptr && S != nullptr;
    double k00 = 0.0, k01 = 0.0, k10 = 0.0, k11 = 0.0, f0 = 0.0, f1 = 0.0;
    for (int32_t t = beg; t < end; ++t) {
        const uint32_t code = (uint32_t)perm2[t];
        const int64_t e = code >> 4;
        const int a = (code >> 2) & 3, b = code & 3;
        const double* g = geo + e * 6;
        const double d1[3] = {g[0], g[1], -(g[0] + g[1])}, d2[3] = {g[2], g[3], -(g[2] + g[3])};
        const double a1 = d1[a], a2 = d2[a], b1 = d1[b], b2 = d2[b], w = g[4];
        if (DS) {
            const double D00 = DS[e], D01 = DS[n_e + e], D02 = DS[2 * n_e + e];
            const double D11 = DS[4 * n_e + e], D12 = DS[5 * n_e + e], D22 = DS[8 * n_e + e];
            // rows of B_a^T D:  r0 = (a1,0,a2) D,  r1 = (0,a2,a1) D                         DP:1047-1050
            const double r00 = a1 * D00 + a2 * D02, r01 = a1 * D01 + a2 * D12, r02 = a1 * D02 + a2 * D22;
            const double r10 = a2 * D01 + a1 * D02, r11 = a2 * D11 + a1 * D12, r12 = a2 * D12 + a1 * D22;
            k00 += w * (r00 * b1 + r02 * b2);
            k01 += w * (r01 * b2 + r02 * b1);
            k10 += w * (r10 * b1 + r12 * b2);
            k11 += w * (r11 * b2 + r12 * b1);
        }
        if (want_f) {                                                                       // DP:1058
            const double s0 = S[e], s1 = S[n_e + e], s2 = S[2 * n_e + e];
            f0 += w * (a1 * s0 + a2 * s2);
            f1 += w * (a2 * s1 + a1 * s2);
        }
    }
    if (data) {
        const int64_t s = m & 0x7fffu, deg = m >> 16;
        const int64_t pos0 = 4 * sb - 2 * s;
        *reinterpret_cast<double2*>(data + pos0) = make_double2(k00, k01);
        *reinterpret_cast<double2*>(data + pos0 + 2 * deg) = make_double2(k10, k11);
    }
    if (want_f) *reinterpret_cast<double2*>(F + 2 * (int64_t)ncol[sb]) = make_double2(f0, f1);
}

// ---------------------------------------------------------------------------------------
// LDS-staged variant of p1_node_kernel (the default).  Uncoalesced 8-byte gathers cost one L1
// (TCP) cycle per lane, which bounds the direct kernel (~100 M lane-loads per launch at 1 M
// elements).  Here each workgroup (256 consecutive node-pair blocks = ~36 consecutive nodes)
// first stages the per-element operands of the elements it touches — w*D (6), w*s (3), dphi (6) —
// into LDS with coalesced loads over its precomputed, sorted element list, then every lane
// gathers from LDS only.
//   wg_elist[wg*L + i]   sorted unique elements of the workgroup, padded to L with its first element
//   perm_l[wg*C + k]     gather codes (local element index << 4) | a << 2 | b   (uint16), padded to C
//   LDS: rec[15][L] doubles, L = max list length, then the workgroup's gather codes (uint16)
// ---------------------------------------------------------------------------------------
// RNG = true: the tile's sorted element list is given as <= 8 runs of consecutive ids, 16 ints per tile
// (start_r, cumulative count_r), fetched with scalar loads; the operand loads then depend on no vector
// load at all, which takes one memory round trip out of every workgroup's critical path.  Chosen by the
// host whenever every tile's list compresses into 8 runs (any mesh numbered with locality).
// PK = true: the lane's block descriptor is ONE 8-byte load (pk.x: beg_local:11 | len:4 | deg:8 | slot:8 | diag:1,
// pk.y: block index in the tile | node index in the tile << 8) instead of segptr x2 + meta + ncol, and the lanes of
// a tile are SORTED by descending segment length: the diagonal blocks (one contribution per incident element, 6 on
// a regular mesh, against 2 for an edge block) fill the first wave instead of setting the trip count of all four.
//
// Tiles (fep_host.h, P1Plan): up to kSegMax segments of consecutive nodes; `tdesc` holds, per tile, 1 + kSegMax int4:
// (pk_base, n_blocks, n_nodes, n_segments) and per segment (first_block, n_blocks, first_node, n_nodes), read with
// scalar loads.  The tile's CSR values and forces leave as one contiguous range per segment.  Without PK every tile
// has exactly one segment (its blocks are first_block + lane).
constexpr int kSegMax = 4;

// LDS-DMA (global_load_lds): the lane's 16 / 4 bytes at `g` go straight to LDS at wave_base + lane * size — no VGPR, no
// ds_write.  `wave_base` must be wave-uniform; the data is visible to ds_read after s_waitcnt vmcnt + a barrier
// (__syncthreads() waits for vmcnt(0)).
__device__ __forceinline__ void glds16(const void* g, void* wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const void* g, void* wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)wave_base, 4, 0, 0);
}

static_assert(kSegMax == 4, "the tile descriptor names its four segments");
// plain ints (no HIP vector types, no arrays): the descriptor stays in scalar registers
struct TileDesc {
    int pk_base, nb, nn, nseg, n_el, n_nd, n_own;       // n_el / n_nd: staged elements / nodes of THIS tile (<= L / NL); the first n_own staged elements are the tile's own
    int fb0, nb0, fn0, nn0, fb1, nb1, fn1, nn1, fb2, nb2, fn2, nn2, fb3, nb3, fn3, nn3;
};

__device__ __forceinline__ TileDesc load_tile_desc(const int4* __restrict__ tdesc, int wg) {
    const int32_t* d = reinterpret_cast<const int32_t*>(tdesc) + (int64_t)wg * 4 * (1 + kSegMax);   // uniform address: scalar loads
    TileDesc t;
    t.pk_base = d[0]; t.nb = d[1]; t.nn = d[2] & 255; t.n_own = d[2] >> 8; t.nseg = d[3] & 15; t.n_el = (d[3] >> 4) & 4095; t.n_nd = (d[3] >> 16) & 4095;
    t.fb0 = d[4]; t.nb0 = d[5]; t.fn0 = d[6]; t.nn0 = d[7];
    t.fb1 = d[8]; t.nb1 = d[9]; t.fn1 = d[10]; t.nn1 = d[11];
    t.fb2 = d[12]; t.nb2 = d[13]; t.fn2 = d[14]; t.nn2 = d[15];
    t.fb3 = d[16]; t.nb3 = d[17]; t.fn3 = d[18]; t.nn3 = d[19];
    return t;
}

// The tile's LDS output image -> HBM: CSR values (2*nb double2 in tile block order) and forces (nn double2 in tile
// node order), each segment to its own contiguous range.
template <int TPB>
__device__ __forceinline__ void store_tile_outputs(const TileDesc t, const double2* __restrict__ out2, const double2* __restrict__ fo2,
                                                   double* __restrict__ data, double* __restrict__ F) {
    if (data) {
        const int n2 = 2 * t.nb;
        const int e0 = 2 * t.nb0, e1 = e0 + 2 * t.nb1, e2 = e1 + 2 * t.nb2;          // ends of the segments' images
        for (int i = threadIdx.x; i < n2; i += TPB) {
            const int64_t fb = i < e0 ? t.fb0 : i < e1 ? t.fb1 : i < e2 ? t.fb2 : t.fb3;
            const int j = i < e0 ? i : i < e1 ? i - e0 : i < e2 ? i - e1 : i - e2;
            reinterpret_cast<double2*>(data + 4 * fb)[j] = out2[i];
        }
    }
    if (F && (int)threadIdx.x < t.nn) {
        const int i = threadIdx.x;
        const int e0 = t.nn0, e1 = e0 + t.nn1, e2 = e1 + t.nn2;
        const int64_t fn = i < e0 ? t.fn0 : i < e1 ? t.fn1 : i < e2 ? t.fn2 : t.fn3;
        const int j = i < e0 ? i : i < e1 ? i - e0 : i < e2 ? i - e1 : i - e2;
        reinterpret_cast<double2*>(F + 2 * fn)[j] = fo2[i];
    }
}

template <int TPB, bool RNG, int EPT, bool PK>
__global__ void __launch_bounds__(TPB)
p1_node_lds_kernel(int64_t n_e, int L, int C, const int32_t* __restrict__ segptr,
                   const uint16_t* __restrict__ perm_l, const uint32_t* __restrict__ meta,
                   const int32_t* __restrict__ ncol,
                   const int32_t* __restrict__ wg_elist, const int4* __restrict__ rng,
                   const uint2* __restrict__ pk, const int4* __restrict__ tdesc,
                   const double* __restrict__ geo,
                   const double* __restrict__ DS, const double* __restrict__ S,
                   double* __restrict__ data, double* __restrict__ F,
                   int n_wg, int n_count_blocks, const uint2* __restrict__ blk_counts,
                   unsigned long long* __restrict__ counts_out) {
    extern __shared__ __attribute__((aligned(16))) double rec[];      // [15][L]
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with a
    // private L2.  Give every XCD one contiguous eighth of the tiles, so that the element rows
    // re-staged by the workgroups of the next node row hit in the SAME L2 instead of the fabric.
    const int chunk = (n_wg + 7) >> 3;
    const int wg = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (counts_out != nullptr && blockIdx.x == 0) sum_block_counts(n_count_blocks, blk_counts, counts_out);
    if (wg >= n_wg) return;
    // Two dependent memory levels only: every table is padded to a fixed per-tile stride (element list:
    // L entries, gather codes: C entries, unused slots repeat a valid entry), so all first-level
    // addresses are functions of (tile, lane) and go out together at kernel entry; the operand loads
    // follow as soon as the list entries land.
    const TileDesc td = load_tile_desc(tdesc, wg);
    const int nb = td.nb;
    const int64_t sb = PK ? (int64_t)td.pk_base + threadIdx.x : (int64_t)td.fb0 + threadIdx.x;
    const bool live = (int)threadIdx.x < nb;
    constexpr int CWPT = 2;                             // 32-bit words of gather codes per lane (2 codes each;
                                                        // host guarantees C <= 2*CWPT*TPB and C even)
    // (1) element ids of the lane's staging slots
    int64_t el[EPT];
    if (RNG) {
        const int4* d = rng + (int64_t)wg * 4;          // uniform address: scalar loads
        const int4 d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
        const int st[8] = {d0.x, d0.z, d1.x, d1.z, d2.x, d2.z, d3.x, d3.z};
        const int cu[8] = {d0.y, d0.w, d1.y, d1.w, d2.y, d2.w, d3.y, d3.w};
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int i = r * TPB + (int)threadIdx.x;
            int e = st[0] + i;
#pragma unroll
            for (int k = 1; k < 8; ++k) e = i >= cu[k - 1] ? st[k] + (i - cu[k - 1]) : e;
            el[r] = i < cu[7] ? e : st[0];
        }
    } else {
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int i = r * TPB + (int)threadIdx.x;
            el[r] = wg_elist[(int64_t)wg * L + (i < L ? i : 0)];
        }
    }
    // (2) operand loads.  Wave-uniform guard: a wave whose 64 slots all lie past the list issues nothing
    // (with L = 150 of 256 slots that is one wave in four: the kernel's time follows its vector-memory
    // instruction count); inside a wave every lane loads (slots past the list repeat a valid element).
    double2 g0[EPT], g1[EPT], g2[EPT];                  // 48-byte geometry record (3 x 16-byte loads; the SoA dphi
    double dv[EPT][6], sv[EPT][3];                      // arrays as 7 x 8-byte loads measured 5-10 us slower)
    const int n_el = td.n_el;                           // slots past the tile's own list stage nothing
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        if (r * TPB + (int)(threadIdx.x & ~63u) >= n_el) continue;
        const int64_t e = el[r];
        const double2* g = reinterpret_cast<const double2*>(geo + e * 6);
        g0[r] = g[0]; g1[r] = g[1]; g2[r] = g[2];
        if (DS) {
            dv[r][0] = DS[e]; dv[r][1] = DS[n_e + e]; dv[r][2] = DS[2 * n_e + e];
            dv[r][3] = DS[4 * n_e + e]; dv[r][4] = DS[5 * n_e + e]; dv[r][5] = DS[8 * n_e + e];
        }
        if (S) { sv[r][0] = S[e]; sv[r][1] = S[n_e + e]; sv[r][2] = S[2 * n_e + e]; }
    }
    // (3) the lane's block descriptors and the tile's gather codes: independent of (1)-(2), consumed last
    int32_t beg, end, ncol_sb = -1;
    uint32_t m;
    uint32_t pk_y = 0;
    if (PK) {
        // lanes of a tile are sorted by descending segment length (host): pk.y = block index inside the tile |
        // index of the lane's node inside the tile << 8
        const uint2 w2 = live ? pk[sb] : make_uint2(0u, 0u);
        const uint32_t w = w2.x;
        pk_y = w2.y;
        beg = (int32_t)(w & 2047u);
        end = beg + (int32_t)((w >> 11) & 15u);
        m = (((w >> 15) & 255u) << 16) | ((w >> 31) << 15) | ((w >> 23) & 255u);     // deg | diag | slot
    } else {
        beg = live ? segptr[sb] : 0;
        end = live ? segptr[sb + 1] : 0;
        m = live ? meta[sb] : 0u;
        ncol_sb = live ? ncol[sb] : -1;
    }
    const int CW = C >> 1;
    const uint32_t* codes_w = reinterpret_cast<const uint32_t*>(perm_l) + (int64_t)wg * CW;
    uint32_t cd[CWPT];
#pragma unroll
    for (int r = 0; r < CWPT; ++r) {
        const int i = r * TPB + (int)threadIdx.x;
        cd[r] = 0u;
        if (r * TPB + (int)(threadIdx.x & ~63u) < CW) cd[r] = codes_w[i < CW ? i : 0];   // wave-uniform: idle waves issue nothing
    }
    // (4) operands -> LDS
    uint16_t* codes = reinterpret_cast<uint16_t*>(rec + 15 * L);
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        const int i = r * TPB + (int)threadIdx.x;
        if (i < n_el) {
            const double w = g2[r].x;
            rec[9 * L + i] = g0[r].x;  rec[10 * L + i] = g0[r].y; rec[11 * L + i] = -(g0[r].x + g0[r].y);   // d1[0..2]
            rec[12 * L + i] = g1[r].x; rec[13 * L + i] = g1[r].y; rec[14 * L + i] = -(g1[r].x + g1[r].y);   // d2[0..2]
            if (DS) {
#pragma unroll
                for (int k = 0; k < 6; ++k) rec[k * L + i] = w * dv[r][k];
            }
            if (S) { rec[6 * L + i] = w * sv[r][0]; rec[7 * L + i] = w * sv[r][1]; rec[8 * L + i] = w * sv[r][2]; }
        }
    }
    // (5) codes -> LDS, first code of the tile
    uint32_t* codes32 = reinterpret_cast<uint32_t*>(rec + 15 * L);
#pragma unroll
    for (int q = 0; q < CWPT; ++q) {
        const int ci = q * TPB + (int)threadIdx.x;
        if (ci < CW) codes32[ci] = cd[q];
    }
    __shared__ int32_t t0_sh;
    const bool is_diag = (m >> 15) & 1u;
    if (!PK && threadIdx.x == 0) t0_sh = beg;
    __syncthreads();
    int32_t t0, fnode_k = -1;
    if (PK) {
        t0 = 0;                                        // pk holds tile-local code offsets
        fnode_k = is_diag ? (int)((pk_y >> 8) & 255u) : -1;       // k-th node of the tile
    } else {
        t0 = t0_sh;
    }
    const bool want_f = live && is_diag && F != nullptr && S != nullptr;
    double k00 = 0.0, k01 = 0.0, k10 = 0.0, k11 = 0.0, f0 = 0.0, f1 = 0.0;
    if (live) {
        for (int32_t t = beg; t < end; ++t) {
            const unsigned code = codes[t - t0];
            const int i = code >> 4, a = (code >> 2) & 3, b = code & 3;
            const double a1 = rec[(9 + a) * L + i], a2 = rec[(12 + a) * L + i];
            const double b1 = rec[(9 + b) * L + i], b2 = rec[(12 + b) * L + i];
            if (DS) {
                const double D00 = rec[i], D01 = rec[L + i], D02 = rec[2 * L + i];
                const double D11 = rec[3 * L + i], D12 = rec[4 * L + i], D22 = rec[5 * L + i];
                const double r00 = a1 * D00 + a2 * D02, r01 = a1 * D01 + a2 * D12, r02 = a1 * D02 + a2 * D22;
                const double r10 = a2 * D01 + a1 * D02, r11 = a2 * D11 + a1 * D12, r12 = a2 * D12 + a1 * D22;
                k00 += r00 * b1 + r02 * b2;
                k01 += r01 * b2 + r02 * b1;
                k10 += r10 * b1 + r12 * b2;
                k11 += r11 * b2 + r12 * b1;
            }
            if (want_f) {
                f0 += a1 * rec[6 * L + i] + a2 * rec[8 * L + i];
                f1 += a2 * rec[7 * L + i] + a1 * rec[8 * L + i];
            }
        }
    }
    // Results leave through LDS: a lane's two 16-byte pieces belong to two different CSR rows, so direct stores
    // write every 128-byte line in two half-filled passes (measured ~14 us per launch); staged, the tile's
    // 4*nb values (and its nodes' forces) go out as full, consecutive lines.
    __syncthreads();                                   // every lane is done reading the staged operands
    double2* out2 = reinterpret_cast<double2*>(rec);
    if (live && data) {
        const int64_t sl = m & 0x7fffu, deg = m >> 16;
        const int blk = PK ? (int)(pk_y & 255u) : (int)threadIdx.x;   // block index inside the tile
        const int rel = 4 * blk - 2 * (int)sl;                   // = CSR position - 4*(first block), always even
        out2[rel >> 1] = make_double2(k00, k01);
        out2[(rel >> 1) + (int)deg] = make_double2(k10, k11);
    }
    double2* fo2 = out2 + 2 * TPB;
    if (PK) { if (want_f) fo2[fnode_k] = make_double2(f0, f1); }
    __syncthreads();
    store_tile_outputs<TPB>(td, out2, fo2, data, (PK && S != nullptr) ? F : nullptr);
    if (!PK && want_f) *reinterpret_cast<double2*>(F + 2 * (int64_t)ncol_sb) = make_double2(f0, f1);
}

// ---------------------------------------------------------------------------------------
// P1, ONE kernel per step (non-accepting calls): the assembly kernel above with the return map moved into its
// staging phase, so that s / ds never make the round trip through HBM (and are not written at all when the caller
// does not ask for them: a Newton iterate reads K and F only, DP:1043-1066).
//
// A tile = 256 node-pair blocks of ~36 consecutive nodes, as in p1_node_lds_kernel.  What is new:
//   * the NODES its ~150 staged elements touch (~110) are staged once per tile: coordinates and displacements go to
//     LDS with one 16-byte load each per node (`nrng`: <= 8 runs of consecutive node ids, scalar loads; or the
//     list `wg_nlist`), instead of 6 dependent 16-byte gathers per staged element;
//   * per staged element one 4-byte word `el_nodes` = its three tile-local node indices (10 bits each) and the
//     OWNER bit: every element is staged by ~2 tiles, exactly one of them (the first in tile order) writes the
//     element's s / ds / ind_p / strain and counts its branch;
//   * the lane of a staged element reads its nodes from LDS, forms the P1 geometry (p1_geometry: bit-identical to
//     geometry_kernel), the strain and the return map (the arithmetic of p1_point_kernel, statement by statement),
//     and leaves w*DS (6), w*s (3), dphi (6) in LDS for the gather phase, which is p1_node_lds_kernel's.
// K and F are bit-identical to the two-kernel route (same operands, same summation order).
// Branch counters: wave ballot -> LDS -> one atomic pair per tile into 256 slots on lines of their own
// (`slot_counts`, summed and re-zeroed by counts_finalize_kernel): integer sums, deterministic.
// Accepting calls (plastic strain updated in place) keep the two-kernel route: a neighbour tile could otherwise
// read an element's updated plastic strain.
// ---------------------------------------------------------------------------------------
// FROM_DS = true: the assembly-only form (fep_assemble_dev, and the second kernel of a full-output / accepting step):
// w*DS and w*s come from the caller's ds / s arrays (`DSin`, `Sin`) instead of the return map, the geometry still from
// the tile's LDS-staged nodes — 48 bytes per staged element less to fetch than p1_node_lds_kernel's geometry record.
// DMA = true: the staged nodes' coordinates / displacements and the tile's gather codes go to LDS by LDS-DMA (the code
// region is padded to 64 words and the node regions to 64 entries: a wave's 64 lanes always land inside their region).
template <bool FULL, int TPB, bool RNG, int EPT, int NPT, bool FROM_DS = false, bool DMA = false>
__global__ void __launch_bounds__(TPB, (!FULL && EPT == 1) ? 8 : 1)     // K/F-only: 64 VGPRs, 8 tiles of 4 waves per CU
p1_fused_kernel(int64_t n_e, int L, int C, int NL,
                const uint16_t* __restrict__ perm_l, const int32_t* __restrict__ wg_elist, const int4* __restrict__ rng,
                const int32_t* __restrict__ wg_nlist, const int4* __restrict__ nrng,
                const uint32_t* __restrict__ el_nodes, const uint2* __restrict__ pk, const int4* __restrict__ tdesc,
                const double* __restrict__ xy, P1Tab tab, const double* __restrict__ U, E0 e0,
                const double* __restrict__ ep,
                const double* __restrict__ shear, const double* __restrict__ bulk,
                const double* __restrict__ eta, const double* __restrict__ cc, MatU mu,
                double* __restrict__ Eout, double* __restrict__ S, double* __restrict__ DS, uint8_t* __restrict__ indp,
                double* __restrict__ data, double* __restrict__ F, int n_wg, unsigned long long* __restrict__ slot_counts,
                const double* __restrict__ DSin, const double* __restrict__ Sin,
                int n_count_blocks, const uint2* __restrict__ blk_counts, unsigned long long* __restrict__ counts_out) {
    static_assert(!(FULL && FROM_DS), "the assembly-only form has no point outputs");
    extern __shared__ __attribute__((aligned(16))) double rec[];      // [15][L] | codes | node coordinates | node displacements
    // assembly-only form after p1_point_kernel: the first workgroup also sums that kernel's per-workgroup branch counters
    if (FROM_DS && counts_out != nullptr && blockIdx.x == 0) sum_block_counts(n_count_blocks, blk_counts, counts_out);
    __shared__ unsigned int sc[2];
    const int chunk = (n_wg + 7) >> 3;                                // XCD-aware tile order (see p1_node_lds_kernel)
    const int wg = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (wg >= n_wg) return;
    const TileDesc td = load_tile_desc(tdesc, wg);
    const int nb = td.nb;
    const int64_t sb = (int64_t)td.pk_base + threadIdx.x;                  // lane of the tile -> its packed descriptor
    const bool live = (int)threadIdx.x < nb;
    constexpr int CWPT = 2;
    if (threadIdx.x == 0) { sc[0] = 0u; sc[1] = 0u; }
    // (1) ids of the lane's element slots and node slots: functions of (tile, lane) through scalar loads
    int64_t el[EPT];
    int64_t nd[NPT];
    if (RNG) {
        const int4* d = rng + (int64_t)wg * 4;
        const int4 d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
        const int st[8] = {d0.x, d0.z, d1.x, d1.z, d2.x, d2.z, d3.x, d3.z};
        const int cu[8] = {d0.y, d0.w, d1.y, d1.w, d2.y, d2.w, d3.y, d3.w};
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int i = r * TPB + (int)threadIdx.x;
            int e = st[0] + i;
#pragma unroll
            for (int k = 1; k < 8; ++k) e = i >= cu[k - 1] ? st[k] + (i - cu[k - 1]) : e;
            el[r] = i < cu[7] ? e : st[0];
        }
        const int4* dn = nrng + (int64_t)wg * 4;
        const int4 n0 = dn[0], n1 = dn[1], n2 = dn[2], n3 = dn[3];
        const int sn[8] = {n0.x, n0.z, n1.x, n1.z, n2.x, n2.z, n3.x, n3.z};
        const int cn[8] = {n0.y, n0.w, n1.y, n1.w, n2.y, n2.w, n3.y, n3.w};
#pragma unroll
        for (int r = 0; r < NPT; ++r) {
            const int i = r * TPB + (int)threadIdx.x;
            int n = sn[0] + i;
#pragma unroll
            for (int k = 1; k < 8; ++k) n = i >= cn[k - 1] ? sn[k] + (i - cn[k - 1]) : n;
            nd[r] = i < cn[7] ? n : sn[0];
        }
    } else {
#pragma unroll
        for (int r = 0; r < EPT; ++r) {
            const int i = r * TPB + (int)threadIdx.x;
            el[r] = wg_elist[(int64_t)wg * L + (i < L ? i : 0)];
        }
#pragma unroll
        for (int r = 0; r < NPT; ++r) {
            const int i = r * TPB + (int)threadIdx.x;
            nd[r] = wg_nlist[(int64_t)wg * NL + (i < NL ? i : 0)];
        }
    }
    // (2) loads, all issued before the first use: node data, the slots' node words / plastic strain / materials
    const int n_el = td.n_el, n_nd = td.n_nd;           // this tile's staged elements / nodes (<= L / NL)
    const int Cp = DMA ? (C + 127) & ~127 : C, NLp = DMA ? (NL + 63) & ~63 : NL;      // padded code / node regions (see above)
    uint16_t* codes = reinterpret_cast<uint16_t*>(rec + 15 * L);
    uint32_t* codes32 = reinterpret_cast<uint32_t*>(rec + 15 * L);
    double2* lxy = reinterpret_cast<double2*>(reinterpret_cast<char*>(rec) + (((size_t)15 * L * 8 + (size_t)Cp * 2 + 15) & ~(size_t)15));
    double2* lu = lxy + NLp;
    double2 nxy[NPT], nu[NPT];
#pragma unroll
    for (int r = 0; r < NPT; ++r) {
        if (r * TPB + (int)(threadIdx.x & ~63u) >= n_nd) continue;   // wave-uniform: idle waves issue nothing
        if (DMA) {
            glds16(xy + 2 * nd[r], lxy + r * TPB + (int)(threadIdx.x & ~63u));
            if (!FROM_DS) glds16(U + 2 * nd[r], lu + r * TPB + (int)(threadIdx.x & ~63u));
        } else {
            nxy[r] = *reinterpret_cast<const double2*>(xy + 2 * nd[r]);
            if (!FROM_DS) nu[r] = *reinterpret_cast<const double2*>(U + 2 * nd[r]);
        }
    }
    uint32_t enw[EPT];
    double pv[EPT][4], mv[EPT][4];
    double dvi[EPT][6], svi[EPT][3];                    // FROM_DS: the caller's tangent / stress of the staged elements
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        if (r * TPB + (int)(threadIdx.x & ~63u) >= n_el) continue;
        const int i = r * TPB + (int)threadIdx.x;
        const int64_t e = el[r];
        enw[r] = el_nodes[(int64_t)wg * L + (i < L ? i : 0)];
        if (FROM_DS) {
            if (DSin) {
                dvi[r][0] = DSin[e]; dvi[r][1] = DSin[n_e + e]; dvi[r][2] = DSin[2 * n_e + e];
                dvi[r][3] = DSin[4 * n_e + e]; dvi[r][4] = DSin[5 * n_e + e]; dvi[r][5] = DSin[8 * n_e + e];
            }
            if (Sin) { svi[r][0] = Sin[e]; svi[r][1] = Sin[n_e + e]; svi[r][2] = Sin[2 * n_e + e]; }
        } else {
            if (ep) { pv[r][0] = ep[e]; pv[r][1] = ep[n_e + e]; pv[r][2] = ep[2 * n_e + e]; pv[r][3] = ep[3 * n_e + e]; }
            else { pv[r][0] = 0.0; pv[r][1] = 0.0; pv[r][2] = 0.0; pv[r][3] = 0.0; }
            if (!mu.on) { mv[r][0] = shear[e]; mv[r][1] = bulk[e]; mv[r][2] = eta[e]; mv[r][3] = cc[e]; }
        }
    }
    // (3) the lane's block descriptor and the tile's gather codes (consumed last)
    const uint2 w2 = live ? pk[sb] : make_uint2(0u, 0u);
    const int CW = C >> 1;
    const uint32_t* codes_w = reinterpret_cast<const uint32_t*>(perm_l) + (int64_t)wg * CW;
    uint32_t cd[CWPT];
#pragma unroll
    for (int r = 0; r < CWPT; ++r) {
        const int i = r * TPB + (int)threadIdx.x;
        cd[r] = 0u;
        if (r * TPB + (int)(threadIdx.x & ~63u) < CW) {
            if (DMA) glds4(codes_w + (i < CW ? i : 0), codes32 + r * TPB + (int)(threadIdx.x & ~63u));
            else cd[r] = codes_w[i < CW ? i : 0];
        }
    }
    // (4) node data and codes -> LDS (already on their way with DMA)
    if (!DMA) {
#pragma unroll
        for (int r = 0; r < NPT; ++r) {
            const int i = r * TPB + (int)threadIdx.x;
            if (i < n_nd) { lxy[i] = nxy[r]; if (!FROM_DS) lu[i] = nu[r]; }
        }
#pragma unroll
        for (int q = 0; q < CWPT; ++q) {
            const int ci = q * TPB + (int)threadIdx.x;
            if (ci < CW) codes32[ci] = cd[q];
        }
    }
    __syncthreads();                                    // (waits for vmcnt(0): the DMA has landed)
    // (5) per staged element: geometry, strain, return map (p1_point_kernel's statements), operands -> LDS
    unsigned int n_sm = 0u, n_ap = 0u;
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        const int i = r * TPB + (int)threadIdx.x;
        int branch = 0;
        bool own = false;
        if (i < n_el) {
            const uint32_t wv = enw[r];
            const int64_t e = el[r];
            own = i < td.n_own;                          // the tile's own elements come first in its staged list
            const int i0 = (int)(wv & 1023u), i1 = (int)((wv >> 10) & 1023u), i2 = (int)((wv >> 20) & 1023u);
            const double2 c0 = lxy[i0], c1 = lxy[i1], c2 = lxy[i2];
            double d1[3], d2[3], w;
            p1_geometry(tab, c0, c1, c2, d1, d2, w);
            double s[4], d[6];
            if (FROM_DS) {
#pragma unroll
                for (int k = 0; k < 6; ++k) d[k] = DSin ? dvi[r][k] : 0.0;
                s[0] = Sin ? svi[r][0] : 0.0; s[1] = Sin ? svi[r][1] : 0.0; s[2] = Sin ? svi[r][2] : 0.0;
            } else {
                const double2 u0 = lu[i0], u1 = lu[i1], u2 = lu[i2];
                double ev[3];                                            // DP:1043, local node order
                ev[0] = d1[0] * u0.x + d1[1] * u1.x + d1[2] * u2.x;
                ev[1] = d2[0] * u0.y + d2[1] * u1.y + d2[2] * u2.y;
                ev[2] = (d2[0] * u0.x + d1[0] * u0.y) + (d2[1] * u1.x + d1[1] * u1.y) + (d2[2] * u2.x + d1[2] * u2.y);
                double p[4] = {pv[r][0], pv[r][1], pv[r][2], pv[r][3]};
                const double m_sh = mu.on ? mu.shear : mv[r][0], m_bu = mu.on ? mu.bulk : mv[r][1];
                const double m_eta = mu.on ? mu.eta : mv[r][2], m_c = mu.on ? mu.c : mv[r][3];
                branch = dp_return_map(ev, e0.v, p, m_sh, m_bu, m_eta, m_c, false, s, d);
                // (wave-uniform first: the waves past the tile's own elements issue none of the 14-17 store instructions)
                if (FULL && r * TPB + (int)(threadIdx.x & ~63u) < td.n_own && own) {
                    store_point(e, n_e, s, d, branch, S, DS, indp);
                    if (Eout) { Eout[e] = ev[0]; Eout[n_e + e] = ev[1]; Eout[2 * n_e + e] = ev[2]; }
                }
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) rec[k * L + i] = w * d[k];
            rec[6 * L + i] = w * s[0]; rec[7 * L + i] = w * s[1]; rec[8 * L + i] = w * s[2];
            rec[9 * L + i] = d1[0];  rec[10 * L + i] = d1[1]; rec[11 * L + i] = -(d1[0] + d1[1]);   // as the 48-byte record
            rec[12 * L + i] = d2[0]; rec[13 * L + i] = d2[1]; rec[14 * L + i] = -(d2[0] + d2[1]);
        }
        if (slot_counts) {                                           // uniform
            n_sm += (unsigned int)__popcll(__ballot(own && branch == 1));
            n_ap += (unsigned int)__popcll(__ballot(own && branch == 2));
        }
    }
    if (slot_counts && (threadIdx.x & 63) == 0) {
        if (n_sm) atomicAdd(&sc[0], n_sm);
        if (n_ap) atomicAdd(&sc[1], n_ap);
    }
    __syncthreads();
    if (slot_counts && threadIdx.x == 0) {
        unsigned long long* slot = slot_counts + (size_t)(wg & 255) * 16;
        if (sc[0]) atomicAdd(&slot[0], (unsigned long long)sc[0]);
        if (sc[1]) atomicAdd(&slot[1], (unsigned long long)sc[1]);
    }
    // (6) gather: p1_node_lds_kernel's, packed descriptors
    const uint32_t wd = w2.x;
    const uint32_t pk_y = w2.y;
    const int32_t beg = (int32_t)(wd & 2047u);
    const int32_t end = beg + (int32_t)((wd >> 11) & 15u);
    const uint32_t deg = (wd >> 15) & 255u, slot_s = (wd >> 23) & 255u;
    const bool is_diag = (wd >> 31) != 0u;
    const int fnode_k = is_diag ? (int)((pk_y >> 8) & 255u) : -1;
    const bool want_f = live && is_diag && F != nullptr;
    double k00 = 0.0, k01 = 0.0, k10 = 0.0, k11 = 0.0, f0 = 0.0, f1 = 0.0;
    if (live) {
        for (int32_t t = beg; t < end; ++t) {
            const unsigned code = codes[t];
            const int i = code >> 4, a = (code >> 2) & 3, b = code & 3;
            const double a1 = rec[(9 + a) * L + i], a2 = rec[(12 + a) * L + i];
            const double b1 = rec[(9 + b) * L + i], b2 = rec[(12 + b) * L + i];
            if (data) {
                const double D00 = rec[i], D01 = rec[L + i], D02 = rec[2 * L + i];
                const double D11 = rec[3 * L + i], D12 = rec[4 * L + i], D22 = rec[5 * L + i];
                const double r00 = a1 * D00 + a2 * D02, r01 = a1 * D01 + a2 * D12, r02 = a1 * D02 + a2 * D22;
                const double r10 = a2 * D01 + a1 * D02, r11 = a2 * D11 + a1 * D12, r12 = a2 * D12 + a1 * D22;
                k00 += r00 * b1 + r02 * b2;
                k01 += r01 * b2 + r02 * b1;
                k10 += r10 * b1 + r12 * b2;
                k11 += r11 * b2 + r12 * b1;
            }
            if (want_f) {
                f0 += a1 * rec[6 * L + i] + a2 * rec[8 * L + i];
                f1 += a2 * rec[7 * L + i] + a1 * rec[8 * L + i];
            }
        }
    }
    __syncthreads();                                   // every lane is done reading the staged operands
    double2* out2 = reinterpret_cast<double2*>(rec);
    if (live && data) {
        const int blk = (int)(pk_y & 255u);
        const int rel = 4 * blk - 2 * (int)slot_s;
        out2[rel >> 1] = make_double2(k00, k01);
        out2[(rel >> 1) + (int)deg] = make_double2(k10, k11);
    }
    double2* fo2 = out2 + 2 * TPB;
    if (want_f) fo2[fnode_k] = make_double2(f0, f1);
    __syncthreads();
    store_tile_outputs<TPB>(td, out2, fo2, data, F);
}

// counts_out[0..1] = sum of the 256 slots (stride 16 words) of p1_fused_kernel; the slots are zeroed for the next step.
__global__ void __launch_bounds__(256)
counts_finalize_kernel(unsigned long long* __restrict__ slot_counts, unsigned long long* __restrict__ counts_out) {
    __shared__ unsigned long long part[2][4];
    unsigned long long* slot = slot_counts + (size_t)threadIdx.x * 16;
    unsigned long long a = slot[0], b = slot[1];
    slot[0] = 0ull; slot[1] = 0ull;
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = a; part[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        counts_out[0] = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        counts_out[1] = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    }
}

// ---------------------------------------------------------------------------------------
// Node route for elements with several integration points (P2, Q1, Q2): the same two-kernel split as
// the P1 fast path, with the geometry recomputed from the node coordinates (gathered through L2)
// instead of streaming dphi (16*n_p bytes per point) from HBM.
//
//   point_kernel<NP,NQ>      one lane per integration point: Jacobian + dphi at its quadrature point
//                            (same operation order as geometry_kernel), strain, return map; writes s, ds, ind_p.
//   node_lds_kernel<NP,NQ>   one workgroup per tile of TPB node-pair blocks: stages, for every (element, q)
//                            of the elements the tile touches, w*D (6), w*s (3) and dphi (2*NP) into LDS
//                            [field][slot], slot = local element * NQ + q, then every lane sums
//                            sum_q w B_a^T DS B_b over its block's contributions in fixed order.
//   gather code = local element << 8 | a << 4 | b  (uint16; <= 256 elements per tile, n_p <= 16)
// ---------------------------------------------------------------------------------------
template <int NP, int NQ>
__global__ void __launch_bounds__(kBlock)
point_kernel(int64_t n_e, const int32_t* __restrict__ elem, const double* __restrict__ xy,
             const double* __restrict__ dh1, const double* __restrict__ dh2, const double* __restrict__ wf,
             const double* __restrict__ U, E0 e0, double* __restrict__ ep,
             const double* __restrict__ shear, const double* __restrict__ bulk,
             const double* __restrict__ eta, const double* __restrict__ cc, MatU mu, int accept,
             double* __restrict__ Eout, double* __restrict__ S, double* __restrict__ DS,
             uint8_t* __restrict__ indp, uint2* blk_counts) {
    __shared__ double t1[NP * NQ], t2[NP * NQ], tw[NQ];
    for (int i = threadIdx.x; i < NP * NQ; i += kBlock) { t1[i] = dh1[i]; t2[i] = dh2[i]; }
    for (int i = threadIdx.x; i < NQ; i += kBlock) tw[i] = wf[i];
    __syncthreads();
    const int64_t n_int = n_e * NQ;
    const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    int branch = 0;
    if (k < n_int) {
        const int64_t e = k / NQ;
        const int q = (int)(k - e * NQ);
        double x[NP], y[NP], ux[NP], uy[NP], d1[NP], d2[NP], w;
#pragma unroll
        for (int a = 0; a < NP; ++a) {
            const int64_t nd = elem[(int64_t)a * n_e + e];
            const double2 c = *reinterpret_cast<const double2*>(xy + 2 * nd);
            const double2 u = *reinterpret_cast<const double2*>(U + 2 * nd);
            x[a] = c.x; y[a] = c.y; ux[a] = u.x; uy[a] = u.y;
        }
        geometry_at_q<NP>(t1, t2, tw[q], q, NQ, x, y, d1, d2, w);
        double ev[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int a = 0; a < NP; ++a) {                   // DP:1043
            ev[0] += d1[a] * ux[a]; ev[1] += d2[a] * uy[a]; ev[2] += d2[a] * ux[a] + d1[a] * uy[a];
        }
        double p[4] = {0.0, 0.0, 0.0, 0.0};
        if (ep) { p[0] = ep[k]; p[1] = ep[n_int + k]; p[2] = ep[2 * n_int + k]; p[3] = ep[3 * n_int + k]; }
        double s[4], d[6];
        const double m_sh = mu.on ? mu.shear : shear[k], m_bu = mu.on ? mu.bulk : bulk[k];
        const double m_eta = mu.on ? mu.eta : eta[k], m_c = mu.on ? mu.c : cc[k];
        branch = dp_return_map(ev, e0.v, p, m_sh, m_bu, m_eta, m_c, accept != 0, s, d);
        store_point(k, n_int, s, d, branch, S, DS, indp);
        if (Eout) { Eout[k] = ev[0]; Eout[n_int + k] = ev[1]; Eout[2 * n_int + k] = ev[2]; }
        if (accept && ep && branch) { ep[k] = p[0]; ep[n_int + k] = p[1]; ep[2 * n_int + k] = p[2]; ep[3 * n_int + k] = p[3]; }
    }
    count_branches(branch, nullptr, blk_counts);
}

template <int NP, int NQ, int TPB>
__global__ void __launch_bounds__(TPB)
node_lds_kernel(int64_t n_blk, int64_t n_e, int L, int C, const int32_t* __restrict__ segptr,
                const uint16_t* __restrict__ perm_l, const uint32_t* __restrict__ meta,
                const int32_t* __restrict__ ncol, const int32_t* __restrict__ wg_elist,
                const int32_t* __restrict__ elem, const double* __restrict__ xy,
                const double* __restrict__ dh1, const double* __restrict__ dh2, const double* __restrict__ wf,
                const double* __restrict__ DS, const double* __restrict__ S,
                double* __restrict__ data, double* __restrict__ F,
                int n_wg, int n_count_blocks, const uint2* __restrict__ blk_counts,
                unsigned long long* __restrict__ counts_out) {
    constexpr int NF = 9 + 2 * NP;                                    // fields per slot
    extern __shared__ __attribute__((aligned(16))) double rec[];      // [NF][SL], tables, codes
    const int SL = L * NQ;
    double* t1 = rec + (int64_t)NF * SL;
    double* t2 = t1 + NP * NQ;
    double* tw = t2 + NP * NQ;
    uint16_t* codes = reinterpret_cast<uint16_t*>(tw + NQ + (NQ & 1));
    __shared__ int32_t t0_sh;
    const int chunk = (n_wg + 7) >> 3;                                // XCD-aware tile order (see p1_node_lds_kernel)
    const int wg = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (counts_out != nullptr && blockIdx.x == 0) sum_block_counts(n_count_blocks, blk_counts, counts_out);
    if (wg >= n_wg) return;
    const int64_t n_int = n_e * NQ;
    const int64_t sb = (int64_t)wg * TPB + threadIdx.x;
    const bool live = sb < n_blk;
    // first-level loads: tables, the lane's block descriptors, codes
    for (int i = threadIdx.x; i < NP * NQ; i += TPB) { t1[i] = dh1[i]; t2[i] = dh2[i]; }
    for (int i = threadIdx.x; i < NQ; i += TPB) tw[i] = wf[i];
    const int32_t beg = live ? segptr[sb] : 0, end = live ? segptr[sb + 1] : 0;
    const uint32_t m = live ? meta[sb] : 0u;
    const int32_t fnode = (live && ((m >> 15) & 1u)) ? ncol[sb] : -1;
    for (int i = threadIdx.x; i < C; i += TPB) codes[i] = perm_l[(int64_t)wg * C + i];
    if (threadIdx.x == 0) t0_sh = beg;
    __syncthreads();                                                  // tables ready
    // staging: one (element, q) slot per lane and pass
    for (int i = threadIdx.x; i < SL; i += TPB) {
        const int el = i / NQ, q = i - el * NQ;
        const int64_t e = wg_elist[(int64_t)wg * L + el];
        const int64_t k = e * NQ + q;
        double x[NP], y[NP], d1[NP], d2[NP], w;
#pragma unroll
        for (int a = 0; a < NP; ++a) {
            const int64_t nd = elem[(int64_t)a * n_e + e];
            const double2 c = *reinterpret_cast<const double2*>(xy + 2 * nd);
            x[a] = c.x; y[a] = c.y;
        }
        double dv[6] = {0, 0, 0, 0, 0, 0}, sv[3] = {0, 0, 0};
        if (DS) {
            dv[0] = DS[k]; dv[1] = DS[n_int + k]; dv[2] = DS[2 * n_int + k];
            dv[3] = DS[4 * n_int + k]; dv[4] = DS[5 * n_int + k]; dv[5] = DS[8 * n_int + k];
        }
        if (S) { sv[0] = S[k]; sv[1] = S[n_int + k]; sv[2] = S[2 * n_int + k]; }
        geometry_at_q<NP>(t1, t2, tw[q], q, NQ, x, y, d1, d2, w);
#pragma unroll
        for (int f = 0; f < 6; ++f) rec[f * SL + i] = w * dv[f];
#pragma unroll
        for (int f = 0; f < 3; ++f) rec[(6 + f) * SL + i] = w * sv[f];
#pragma unroll
        for (int a = 0; a < NP; ++a) { rec[(9 + a) * SL + i] = d1[a]; rec[(9 + NP + a) * SL + i] = d2[a]; }
    }
    __syncthreads();
    if (!live) return;
    const int32_t t0 = t0_sh;
    const bool want_f = fnode >= 0 && F != nullptr && S != nullptr;
    double k00 = 0.0, k01 = 0.0, k10 = 0.0, k11 = 0.0, f0 = 0.0, f1 = 0.0;
    for (int32_t t = beg; t < end; ++t) {
        const unsigned code = codes[t - t0];
        const int a = (code >> 4) & 15, b = code & 15;
        const int s0 = (int)(code >> 8) * NQ;
        for (int q = 0; q < NQ; ++q) {
            const int sl = s0 + q;
            const double a1 = rec[(9 + a) * SL + sl], a2 = rec[(9 + NP + a) * SL + sl];
            if (DS) {
                const double b1 = rec[(9 + b) * SL + sl], b2 = rec[(9 + NP + b) * SL + sl];
                const double D00 = rec[sl], D01 = rec[SL + sl], D02 = rec[2 * SL + sl];
                const double D11 = rec[3 * SL + sl], D12 = rec[4 * SL + sl], D22 = rec[5 * SL + sl];
                const double r00 = a1 * D00 + a2 * D02, r01 = a1 * D01 + a2 * D12, r02 = a1 * D02 + a2 * D22;
                const double r10 = a2 * D01 + a1 * D02, r11 = a2 * D11 + a1 * D12, r12 = a2 * D12 + a1 * D22;
                k00 += r00 * b1 + r02 * b2;
                k01 += r01 * b2 + r02 * b1;
                k10 += r10 * b1 + r12 * b2;
                k11 += r11 * b2 + r12 * b1;
            }
            if (want_f) {
                f0 += a1 * rec[6 * SL + sl] + a2 * rec[8 * SL + sl];
                f1 += a2 * rec[7 * SL + sl] + a1 * rec[8 * SL + sl];
            }
        }
    }
    if (data) {
        const int64_t s = m & 0x7fffu, deg = m >> 16;
        const int64_t pos0 = 4 * sb - 2 * s;
        *reinterpret_cast<double2*>(data + pos0) = make_double2(k00, k01);
        *reinterpret_cast<double2*>(data + pos0 + 2 * deg) = make_double2(k10, k11);
    }
    if (want_f) *reinterpret_cast<double2*>(F + 2 * (int64_t)fnode) = make_double2(f0, f1);
}

// Interface exchange helpers (multi-GPU): pack the rank's interface DOFs into the all-reduce buffer
// (slots owned by other ranks are written as zeros, so no separate memset) and unpack the sums.
__global__ void __launch_bounds__(kBlock)
gather_or_zero_kernel(int64_t n, const double* __restrict__ src, const int32_t* __restrict__ idx, double* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) { const int32_t j = idx[i]; dst[i] = j >= 0 ? src[j] : 0.0; }
}

__global__ void __launch_bounds__(kBlock)
scatter_kernel(int64_t n, const double* __restrict__ src, const int32_t* __restrict__ src_idx,
               const int32_t* __restrict__ dst_idx, double* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[dst_idx[i]] = src[src_idx[i]];
}

// Neighbour-only interface exchange, last step (sharding.py): interface DOF i (local DOF loc[i]) becomes
// 0 + c_0 + c_1 + ... over its holders in ascending rank order, c = this rank's own value (src < 0) or the value a
// neighbour sent (recv[src]).  The same sequence of additions on every holder: the same bits everywhere.
__global__ void __launch_bounds__(kBlock)
iface_sum_kernel(int64_t n, const int32_t* __restrict__ loc, const int32_t* __restrict__ ptr, const int32_t* __restrict__ src,
                 const double* __restrict__ recv, double* __restrict__ f) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int32_t d = loc[i];
    const double own = f[d];
    double acc = 0.0;
    for (int32_t k = ptr[i]; k < ptr[i + 1]; ++k) { const int32_t s = src[k]; acc += s < 0 ? own : recv[s]; }
    f[d] = acc;
}

}  // namespace fep
