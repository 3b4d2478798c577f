// Krylov solver on the tangent system of one Newton iterate (include/fep.h, "linear solve"):
//     K[Q][:,Q] dU[Q] = -F[Q]        (np.linalg.solve at DP:1062-1066 / TSX:1781, dense in the reference)
// Preconditioned conjugate gradients in the single-reduction form of Chronopoulos & Gear: per iteration one
// fused vector kernel (p, s, x, r, u = M^-1 r and the partial sums of (r,u), (r,r)), one block-row SpMV
// (w = K u with the partial sums of (w,u)) and a one-workgroup scalar kernel (alpha, beta, stopping test), all
// stream-ordered; the host only reads 48 bytes every `check_every` iterations.  Sums are taken in a fixed
// order (per-workgroup partials, then one tree), so a solve is reproducible run to run.
//
// K keeps the layout the assembly kernels write (fep_ctx pattern): DOF = 2*node + comp, the rows 2n and 2n+1 of
// node n hold, back to back, deg(n) pairs (k_r0, k_r1) for the sorted neighbour nodes.  Constrained DOFs are
// masked out (rows and columns), the solution is 0 there.  M = the 2x2 node blocks of K[Q][:,Q].
#include "fep_common.h"
#include "fep_host.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <cstdlib>
#include <vector>

namespace {

constexpr int TPB = 256;
constexpr int NODES_PER_WAVE = 8;           // 8 lanes per NODE (both of its DOF rows), 8 nodes per wave and pass
constexpr int SPMV_PASSES = 4;              // passes per wave (independent, for loads in flight)
constexpr int NODES_PER_BLOCK = (TPB / 64) * NODES_PER_WAVE * SPMV_PASSES;

struct Scal {                 // device-resident scalars of one solve
    double alpha, beta, gamma, rr, bb;
    int32_t it;               // iterations applied to x so far
    int32_t state;            // 0 running, 1 converged (frozen), 2 breakdown (K not positive definite on Q / NaN)
};

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sum over the workgroup, result valid in thread 0 (fixed order)
__device__ inline double block_sum(double v, double* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    __syncthreads();
    return t;
}

// M^-1 of node n from the diagonal 2x2 block of K with constrained DOFs replaced by identity rows/columns
__global__ void __launch_bounds__(TPB)
block_jacobi_kernel(int64_t n_n, const int32_t* __restrict__ nptr, const int32_t* __restrict__ ncol,
                    const uint8_t* __restrict__ free_dof, const double* __restrict__ K, double* __restrict__ minv) {
    const int64_t n = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (n >= n_n) return;
    const int b0 = nptr[n], deg = nptr[n + 1] - b0;
    double a = 1.0, b = 0.0, c = 0.0, d = 1.0;
    for (int s = 0; s < deg; ++s)
        if (ncol[b0 + s] == (int32_t)n) {
            const double* base = K + 4 * (int64_t)b0;
            a = base[2 * s]; b = base[2 * s + 1]; c = base[2 * deg + 2 * s]; d = base[2 * deg + 2 * s + 1];
        }
    const bool f0 = free_dof[2 * n] != 0, f1 = free_dof[2 * n + 1] != 0;
    if (!f0) { a = 1.0; b = 0.0; c = 0.0; }
    if (!f1) { d = 1.0; b = 0.0; c = 0.0; }
    const double sym = 0.5 * (b + c);                 // K is symmetric up to the summation order
    double det = a * d - sym * sym;
    if (!(det > 0.0) || !(a > 0.0)) { a = 1.0; d = 1.0; det = 1.0; b = c = 0.0; }
    const double off = (b == 0.0 && c == 0.0) ? 0.0 : -sym / det;
    minv[3 * n] = d / det; minv[3 * n + 1] = off; minv[3 * n + 2] = a / det;
}

// r = Q b, u = M^-1 r, x = p = s = 0; partial sums of (r,u) and (r,r)
__global__ void __launch_bounds__(TPB)
pcg_init_kernel(int64_t n_n, const double2* __restrict__ b, const uint8_t* __restrict__ free_dof,
                const double* __restrict__ minv, double2* __restrict__ x, double2* __restrict__ r,
                double2* __restrict__ u, double2* __restrict__ p, double2* __restrict__ s,
                double* __restrict__ part_g, double* __restrict__ part_r) {
    __shared__ double sh[TPB / 64];
    const int64_t n = (int64_t)blockIdx.x * TPB + threadIdx.x;
    double g = 0.0, rr = 0.0;
    if (n < n_n) {
        double2 rv = b[n];
        if (!free_dof[2 * n]) rv.x = 0.0;
        if (!free_dof[2 * n + 1]) rv.y = 0.0;
        const double m0 = minv[3 * n], m1 = minv[3 * n + 1], m2 = minv[3 * n + 2];
        const double2 uv = make_double2(m0 * rv.x + m1 * rv.y, m1 * rv.x + m2 * rv.y);
        const double2 z = make_double2(0.0, 0.0);
        x[n] = z; p[n] = z; s[n] = z; r[n] = rv; u[n] = uv;
        g = rv.x * uv.x + rv.y * uv.y;
        rr = rv.x * rv.x + rv.y * rv.y;
    }
    g = block_sum(g, sh);
    rr = block_sum(rr, sh);
    if (threadIdx.x == 0) { part_g[blockIdx.x] = g; part_r[blockIdx.x] = rr; }
}

// p = u + beta p, s = w + beta s, x += alpha p, r -= alpha s, u = M^-1 r; partial sums of (r,u), (r,r)
__global__ void __launch_bounds__(TPB)
pcg_update_kernel(int64_t n_n, const Scal* __restrict__ sc, const double2* __restrict__ w,
                  const double* __restrict__ minv, double2* __restrict__ x, double2* __restrict__ r,
                  double2* __restrict__ u, double2* __restrict__ p, double2* __restrict__ s,
                  double* __restrict__ part_g, double* __restrict__ part_r) {
    __shared__ double sh[TPB / 64];
    const double alpha = sc->alpha, beta = sc->beta;
    const int64_t n = (int64_t)blockIdx.x * TPB + threadIdx.x;
    double g = 0.0, rr = 0.0;
    if (n < n_n) {
        const double2 uv = u[n], wv = w[n];
        double2 pv = p[n], sv = s[n], xv = x[n], rv = r[n];
        pv.x = uv.x + beta * pv.x; pv.y = uv.y + beta * pv.y;
        sv.x = wv.x + beta * sv.x; sv.y = wv.y + beta * sv.y;
        xv.x += alpha * pv.x; xv.y += alpha * pv.y;
        rv.x -= alpha * sv.x; rv.y -= alpha * sv.y;
        const double m0 = minv[3 * n], m1 = minv[3 * n + 1], m2 = minv[3 * n + 2];
        const double2 un = make_double2(m0 * rv.x + m1 * rv.y, m1 * rv.x + m2 * rv.y);
        if (alpha != 0.0 || beta != 0.0) { p[n] = pv; s[n] = sv; x[n] = xv; r[n] = rv; u[n] = un; }
        g = rv.x * un.x + rv.y * un.y;
        rr = rv.x * rv.x + rv.y * rv.y;
    }
    g = block_sum(g, sh);
    rr = block_sum(rr, sh);
    if (threadIdx.x == 0) { part_g[blockIdx.x] = g; part_r[blockIdx.x] = rr; }
}

// y = Q K x (x is 0 on constrained DOFs by construction when MASKED); optional partial sums of (y, dotv)
template <bool MASKED>
__global__ void __launch_bounds__(TPB)
spmv_kernel(int64_t n_n, const int32_t* __restrict__ nptr, const int32_t* __restrict__ ncol,
            const uint8_t* __restrict__ free_dof, const double2* __restrict__ K2, const double2* __restrict__ x,
            double* __restrict__ y, const double* __restrict__ dotv, double* __restrict__ part_d) {
    __shared__ double sh[TPB / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 7, grp = lane >> 3;            // 8 lanes per node: lane `sub` takes the node's blocks sub, sub+8, ...
    // both DOF rows of a node share their column list and their x values: one lane forms both partial sums of its
    // blocks (k row 0, k row 1, neighbour id and x fetched once) instead of two lane groups fetching ids and x twice
    const int64_t node0 = (int64_t)blockIdx.x * NODES_PER_BLOCK + (int64_t)wave * (NODES_PER_WAVE * SPMV_PASSES) + grp;
    double dot = 0.0;
    double acc0[SPMV_PASSES], acc1[SPMV_PASSES];
#pragma unroll
    for (int ps = 0; ps < SPMV_PASSES; ++ps) {
        const int64_t n = node0 + ps * NODES_PER_WAVE;
        acc0[ps] = 0.0; acc1[ps] = 0.0;
        if (n < n_n) {
            const int b0 = nptr[n], deg = nptr[n + 1] - b0;
            const double2* row0 = K2 + 2 * (int64_t)b0;
            const double2* row1 = row0 + deg;
            for (int t = sub; t < deg; t += 8) {
                const double2 k0 = row0[t], k1 = row1[t];
                const double2 xv = x[ncol[b0 + t]];
                acc0[ps] += k0.x * xv.x + k0.y * xv.y;
                acc1[ps] += k1.x * xv.x + k1.y * xv.y;
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < SPMV_PASSES; ++ps) {
        double a0 = acc0[ps], a1 = acc1[ps];
        a0 += __shfl_xor(a0, 1, 64); a1 += __shfl_xor(a1, 1, 64);
        a0 += __shfl_xor(a0, 2, 64); a1 += __shfl_xor(a1, 2, 64);
        a0 += __shfl_xor(a0, 4, 64); a1 += __shfl_xor(a1, 4, 64);
        const int64_t n = node0 + ps * NODES_PER_WAVE;
        if (sub == 0 && n < n_n) {
            if (MASKED && !free_dof[2 * n]) a0 = 0.0;
            if (MASKED && !free_dof[2 * n + 1]) a1 = 0.0;
            reinterpret_cast<double2*>(y)[n] = make_double2(a0, a1);
            if (dotv) { const double2 dv = reinterpret_cast<const double2*>(dotv)[n]; dot += a0 * dv.x + a1 * dv.y; }
        }
    }
    if (part_d) {
        dot = block_sum(dot, sh);
        if (threadIdx.x == 0) part_d[blockIdx.x] = dot;
    }
}

__device__ inline double sum_partials(const double* part, int n, double* sh) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += part[i];
    return block_sum(v, sh);
}

// one workgroup: gamma' = (r,u), delta = (w,u), rr = (r,r) from the partials; next alpha, beta; stopping test
__global__ void __launch_bounds__(1024)
pcg_scalar_kernel(Scal* sc, const double* part_g, const double* part_r, int n_vec, const double* part_d, int n_mv,
                  double tol2, int first) {
    __shared__ double sh[16];
    const double g = sum_partials(part_g, n_vec, sh);
    const double rr = sum_partials(part_r, n_vec, sh);
    const double d = sum_partials(part_d, n_mv, sh);
    if (threadIdx.x != 0) return;
    Scal s = *sc;
    if (first) {
        s.bb = rr; s.rr = rr; s.gamma = g; s.it = 0; s.state = 0; s.beta = 0.0;
        if (rr == 0.0) { s.state = 1; s.alpha = 0.0; }
        else if (!(g > 0.0) || !(d > 0.0)) { s.state = 2; s.alpha = 0.0; }
        else s.alpha = g / d;
        *sc = s;
        return;
    }
    if (s.state != 0) return;                               // frozen: alpha = beta = 0 leave x untouched
    s.it += 1;
    s.rr = rr;
    if (rr <= tol2 * s.bb) { s.state = 1; s.alpha = 0.0; s.beta = 0.0; *sc = s; return; }
    const double beta = g / s.gamma;
    const double den = d - beta * g / s.alpha;
    if (!(g > 0.0) || !(den > 0.0) || !(beta == beta)) { s.state = 2; s.alpha = 0.0; s.beta = 0.0; *sc = s; return; }
    s.beta = beta;
    s.alpha = g / den;
    s.gamma = g;
    *sc = s;
}


// ---- multigrid preconditioner (smoothed aggregation; hierarchy built on the host, fep_solver_amg_push_level) ----
// Level 0 works on the block matrix K itself: out = Q (b - K x)  or, SMOOTH, one damped block-Jacobi sweep
// out = x + omega M^-1 Q (b - K x)   (x != out).
// SMOOTH = 2: the second step of a degree-2 Chebyshev smoother,
//   out = ca x + cprev xprev + omega M^-1 Q (b - K x)      (xprev == nullptr: that term is absent).
// KV = float2: the V-cycle's passes read a single-precision copy of K (half the bytes of the pass; the preconditioner need not
// see more than seven digits of the matrix, the outer iteration's own product stays in double precision).
template <int SMOOTH, typename KV>
__global__ void __launch_bounds__(TPB)
block_residual_kernel(int64_t n_n, const int32_t* __restrict__ nptr, const int32_t* __restrict__ ncol,
                      const uint8_t* __restrict__ free_dof, const KV* __restrict__ K2,
                      const double2* __restrict__ x, const double* __restrict__ b, const double* __restrict__ minv,
                      double omega, double* __restrict__ out,
                      double ca = 1.0, double cprev = 0.0, const double2* __restrict__ xprev = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 7, grp = lane >> 3;            // 8 lanes per node, both DOF rows per lane (see spmv_kernel)
    const int64_t node0 = (int64_t)blockIdx.x * NODES_PER_BLOCK + (int64_t)wave * (NODES_PER_WAVE * SPMV_PASSES) + grp;
    double acc0[SPMV_PASSES], acc1[SPMV_PASSES];
#pragma unroll
    for (int ps = 0; ps < SPMV_PASSES; ++ps) {
        const int64_t n = node0 + ps * NODES_PER_WAVE;
        acc0[ps] = 0.0; acc1[ps] = 0.0;
        if (n < n_n) {
            const int b0 = nptr[n], deg = nptr[n + 1] - b0;
            const KV* row0 = K2 + 2 * (int64_t)b0;
            const KV* row1 = row0 + deg;
            for (int t = sub; t < deg; t += 8) {
                const KV k0 = row0[t], k1 = row1[t];
                const double2 xv = x[ncol[b0 + t]];
                acc0[ps] += (double)k0.x * xv.x + (double)k0.y * xv.y;
                acc1[ps] += (double)k1.x * xv.x + (double)k1.y * xv.y;
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < SPMV_PASSES; ++ps) {
        double a0 = acc0[ps], a1 = acc1[ps];
        a0 += __shfl_xor(a0, 1, 64); a1 += __shfl_xor(a1, 1, 64);
        a0 += __shfl_xor(a0, 2, 64); a1 += __shfl_xor(a1, 2, 64);
        a0 += __shfl_xor(a0, 4, 64); a1 += __shfl_xor(a1, 4, 64);
        const int64_t n = node0 + ps * NODES_PER_WAVE;
        if (sub != 0 || n >= n_n) continue;
        const bool f0 = free_dof[2 * n] != 0, f1 = free_dof[2 * n + 1] != 0;
        const double2 bv = reinterpret_cast<const double2*>(b)[n];
        const double r0 = f0 ? bv.x - a0 : 0.0, r1 = f1 ? bv.y - a1 : 0.0;
        if (SMOOTH) {
            const double m0 = minv[3 * n], m1 = minv[3 * n + 1], m2 = minv[3 * n + 2];
            const double d0 = m0 * r0 + m1 * r1, d1 = m1 * r0 + m2 * r1;
            const double2 xo = x[n];
            double o0, o1;
            if (SMOOTH == 2) {
                const double2 xp = xprev ? xprev[n] : make_double2(0.0, 0.0);
                o0 = ca * xo.x + cprev * xp.x + omega * d0;
                o1 = ca * xo.y + cprev * xp.y + omega * d1;
            } else {
                o0 = xo.x + omega * d0;
                o1 = xo.y + omega * d1;
            }
            reinterpret_cast<double2*>(out)[n] = make_double2(f0 ? o0 : 0.0, f1 ? o1 : 0.0);
        } else {
            reinterpret_cast<double2*>(out)[n] = make_double2(r0, r1);
        }
    }
}

// x = omega M^-1 Q b  (first sweep from x = 0)
__global__ void __launch_bounds__(TPB)
block_scale_kernel(int64_t n_n, const uint8_t* __restrict__ free_dof, const double* __restrict__ minv,
                   const double2* __restrict__ b, double omega, double2* __restrict__ x) {
    const int64_t n = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (n >= n_n) return;
    double2 r = b[n];
    if (!free_dof[2 * n]) r.x = 0.0;
    if (!free_dof[2 * n + 1]) r.y = 0.0;
    const double m0 = minv[3 * n], m1 = minv[3 * n + 1], m2 = minv[3 * n + 2];
    x[n] = make_double2(omega * (m0 * r.x + m1 * r.y), omega * (m1 * r.x + m2 * r.y));
}

// single-precision copy of K for the V-cycle (one pass per solve)
__global__ void __launch_bounds__(TPB)
to_float_kernel(int64_t n4, const double4* __restrict__ in, float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n4) return;
    const double4 v = in[i];
    out[i] = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}

__global__ void __launch_bounds__(TPB)
to_float1_kernel(int64_t n, const double* __restrict__ in, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

// Coarse levels and transfers: scalar CSR, 8 lanes per row:  y = c0 * z + c1 * A x   (z == nullptr: y = c1 * A x;
// y may alias z, never x)
// Z2: y = c0 * z + c2 * z2 + c1 * A x  (y may alias z or z2)
// LPR lanes per row: 8 for operator / restriction rows (30-70 entries), 2 for prolongation rows (~7 entries: with 8 lanes a
// wave carried 450 bytes of matrix)
template <bool Z2, int LPR>
__global__ void __launch_bounds__(TPB)
csr_kernel(int64_t n_rows, const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
           const double* __restrict__ vals, const double* __restrict__ x, const double* z, double c0, double c1,
           double* y, const double* z2 = nullptr, double c2 = 0.0) {
    const int sub = threadIdx.x & (LPR - 1);
    const int64_t row = ((int64_t)blockIdx.x * TPB + threadIdx.x) / LPR;
    double a = 0.0;
    if (row < n_rows) {
        const int32_t e = indptr[row + 1];
        for (int32_t t = indptr[row] + sub; t < e; t += LPR) a += vals[t] * x[indices[t]];
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) a += __shfl_xor(a, o, 64);
    if (row < n_rows && sub == 0) {
        if (Z2) y[row] = c0 * z[row] + c2 * z2[row] + c1 * a;
        else y[row] = (z ? c0 * z[row] : 0.0) + c1 * a;
    }
}

// Transfers in node blocks, single-precision values (Level::Blocks).
// Prolongation, one lane per fine node:  y(node) = z(node) + sum_blocks B (BF x 3) xc(coarse node)      (y may alias z)
template <int BF>
__global__ void __launch_bounds__(TPB)
prolong_block_kernel(int64_t n_fine_nodes, const int32_t* __restrict__ ptr, const int32_t* __restrict__ col,
                     const float* __restrict__ val, const double* __restrict__ xc, const double* z, double* y) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n_fine_nodes) return;
    double a[BF];
#pragma unroll
    for (int r = 0; r < BF; ++r) a[r] = 0.0;
    for (int32_t t = ptr[i], e = ptr[i + 1]; t < e; ++t) {
        const double* xj = xc + 3 * (int64_t)col[t];
        const double x0 = xj[0], x1 = xj[1], x2 = xj[2];
        const float* v = val + (int64_t)t * (3 * BF);
#pragma unroll
        for (int r = 0; r < BF; ++r) a[r] += (double)v[3 * r] * x0 + (double)v[3 * r + 1] * x1 + (double)v[3 * r + 2] * x2;
    }
#pragma unroll
    for (int r = 0; r < BF; ++r) y[BF * i + r] = z[BF * i + r] + a[r];
}

// Restriction, 8 lanes per coarse node:  bc(coarse node) = sum_blocks B^T (3 x BF) r(fine node)
template <int BF>
__global__ void __launch_bounds__(TPB)
restrict_block_kernel(int64_t n_coarse_nodes, const int32_t* __restrict__ ptr, const int32_t* __restrict__ col,
                      const float* __restrict__ val, const double* __restrict__ r, double* __restrict__ bc) {
    const int sub = threadIdx.x & 7;
    const int64_t J = ((int64_t)blockIdx.x * TPB + threadIdx.x) >> 3;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (J < n_coarse_nodes) {
        for (int32_t t = ptr[J] + sub, e = ptr[J + 1]; t < e; t += 8) {
            const double* ri = r + BF * (int64_t)col[t];
            const float* v = val + (int64_t)t * (3 * BF);
#pragma unroll
            for (int q = 0; q < BF; ++q) {
                const double rv = ri[q];
                a0 += (double)v[q] * rv; a1 += (double)v[BF + q] * rv; a2 += (double)v[2 * BF + q] * rv;
            }
        }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
    if (J < n_coarse_nodes && sub == 0) { bc[3 * J] = a0; bc[3 * J + 1] = a1; bc[3 * J + 2] = a2; }
}

// standard PCG pieces around the V-cycle
__global__ void __launch_bounds__(TPB)
mg_init_kernel(int64_t n_n, const double2* __restrict__ b, const uint8_t* __restrict__ free_dof, double2* __restrict__ x,
               double2* __restrict__ r, double* __restrict__ part_r) {
    __shared__ double sh[TPB / 64];
    const int64_t n = (int64_t)blockIdx.x * TPB + threadIdx.x;
    double rr = 0.0;
    if (n < n_n) {
        double2 rv = b[n];
        if (!free_dof[2 * n]) rv.x = 0.0;
        if (!free_dof[2 * n + 1]) rv.y = 0.0;
        x[n] = make_double2(0.0, 0.0); r[n] = rv;
        rr = rv.x * rv.x + rv.y * rv.y;
    }
    rr = block_sum(rr, sh);
    if (threadIdx.x == 0) part_r[blockIdx.x] = rr;
}

// partial sums of (a, b); COPY: p = a as well (first iteration: p = z)
template <bool COPY>
__global__ void __launch_bounds__(TPB)
mg_dot_kernel(int64_t n_n, const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ p,
              double* __restrict__ part) {
    __shared__ double sh[TPB / 64];
    const int64_t n = (int64_t)blockIdx.x * TPB + threadIdx.x;
    double d = 0.0;
    if (n < n_n) {
        const double2 av = a[n], bv = b[n];
        d = av.x * bv.x + av.y * bv.y;
        if (COPY) p[n] = av;
    }
    d = block_sum(d, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = d;
}

// x += alpha p, r -= alpha q; partial sums of (r, r)
__global__ void __launch_bounds__(TPB)
mg_update_kernel(int64_t n_n, const Scal* __restrict__ sc, const double2* __restrict__ p, const double2* __restrict__ q,
                 double2* __restrict__ x, double2* __restrict__ r, double* __restrict__ part_r) {
    __shared__ double sh[TPB / 64];
    const double alpha = sc->alpha;
    const int64_t n = (int64_t)blockIdx.x * TPB + threadIdx.x;
    double rr = 0.0;
    if (n < n_n) {
        double2 rv = r[n];
        if (alpha != 0.0) {
            const double2 pv = p[n], qv = q[n];
            double2 xv = x[n];
            xv.x += alpha * pv.x; xv.y += alpha * pv.y;
            rv.x -= alpha * qv.x; rv.y -= alpha * qv.y;
            x[n] = xv; r[n] = rv;
        }
        rr = rv.x * rv.x + rv.y * rv.y;
    }
    rr = block_sum(rr, sh);
    if (threadIdx.x == 0) part_r[blockIdx.x] = rr;
}

// p = z + beta p
__global__ void __launch_bounds__(TPB)
mg_direction_kernel(int64_t n_n, const Scal* __restrict__ sc, const double2* __restrict__ z, double2* __restrict__ p) {
    const double beta = sc->beta;
    const int64_t n = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (n >= n_n || sc->state != 0) return;
    const double2 zv = z[n];
    double2 pv = p[n];
    pv.x = zv.x + beta * pv.x; pv.y = zv.y + beta * pv.y;
    p[n] = pv;
}

// scalar steps of the standard PCG: which = 0 after the first (r, z): gamma, bb; 1 after (p, q): alpha;
// 2 after the update and the new (r, z): stopping test, beta
__global__ void __launch_bounds__(1024)
mg_scalar_kernel(Scal* sc, int which, const double* part_a, int n_a, const double* part_b, int n_b, double tol2) {
    __shared__ double sh[16];
    const double a = sum_partials(part_a, n_a, sh);
    const double b = part_b ? sum_partials(part_b, n_b, sh) : 0.0;
    if (threadIdx.x != 0) return;
    Scal s = *sc;
    if (which == 0) {                       // a = (r, z), b = (r, r)
        s.bb = b; s.rr = b; s.gamma = a; s.it = 0; s.alpha = 0.0; s.beta = 0.0;
        s.state = b == 0.0 ? 1 : (!(a > 0.0) ? 2 : 0);
    } else if (s.state == 0 && which == 1) { // a = (p, q)
        if (!(a > 0.0)) { s.state = 2; s.alpha = 0.0; } else s.alpha = s.gamma / a;
    } else if (s.state == 0) {              // a = (r, z) new, b = (r, r) new
        s.it += 1; s.rr = b;
        if (b <= tol2 * s.bb) { s.state = 1; s.alpha = 0.0; s.beta = 0.0; }
        else if (!(a > 0.0) || !(a == a)) { s.state = 2; s.alpha = 0.0; s.beta = 0.0; }
        else { s.beta = a / s.gamma; s.gamma = a; }
    }
    if (s.state != 0) { s.alpha = 0.0; s.beta = 0.0; }
    *sc = s;
}


// ---- coarse operators re-projected from the current tangent (fep_solver_amg_enable_refresh) ----
// One entry of a sparse product on fixed patterns per thread: out[c] = sum_t coef[t] * V[idx[t]] over the entry's terms
// (fep_host.h: product_plan), in the plan's order.  One factor of either product is a transfer matrix, fixed since the
// set-up: its value rides in the term (one 16-byte load per term and one gather instead of two index loads and two gathers:
// 5.4 -> ... ms per refresh at 1 M DOFs).
struct Term { double coef; int32_t idx; int32_t pad; };
// set-up: the terms from the plan's index pairs; the constant factor's values are on the device already (`first`: it is the
// first factor — R in R * T — else the second — P in A * P)
__global__ void __launch_bounds__(TPB)
compose_terms_kernel(int64_t n, const int32_t* __restrict__ xa, const int32_t* __restrict__ ya, const double* __restrict__ cv,
                     int first, Term* __restrict__ terms) {
    const int64_t t = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (t >= n) return;
    const int32_t x = xa[t], y = ya[t];
    terms[t] = first ? Term{cv[x], y, 0} : Term{cv[y], x, 0};
}
// set-up, the terms built on the device (default; FEP_AMG_PLAN=host keeps fep_host.h's product_plan and the upload of its
// index pairs): one thread per row i of X walks C(i, :) = sum_x X(i, j) Y(j, :) in ascending order of the X entry — the order
// of product_plan, so both builders leave identical term lists — and finds every term's output entry by bisection in row i
// of the (column-sorted) pattern of C.  !FILL: terms per output entry into cnt; FILL: the terms themselves at tptr[c] + cnt[c]++
// (cnt zeroed before either pass; a row owns its output entries, so the counters need no atomics).  X_CONST: the constant
// factor is X (R in R * T) and the term carries X's value and Y's entry index, else Y's value and X's entry index.
// dense_cols > 0: C is dense with that many columns.  *bad != 0: a structural term has no entry in the pattern.
template <bool FILL, bool X_CONST>
__global__ void __launch_bounds__(TPB)
plan_rows_kernel(int64_t n_rows, int64_t n_mid, const int32_t* __restrict__ Xp, const int32_t* __restrict__ Xi,
                 const int32_t* __restrict__ Yp, const int32_t* __restrict__ Yi, const int32_t* __restrict__ Cp,
                 const int32_t* __restrict__ Ci, int64_t dense_cols, int32_t* __restrict__ cnt,
                 const int32_t* __restrict__ tptr, const double* __restrict__ cv, Term* __restrict__ terms, int* bad) {
    const int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n_rows) return;
    const int64_t c0 = dense_cols ? i * dense_cols : (int64_t)Cp[i];
    const int64_t c1 = dense_cols ? c0 + dense_cols : (int64_t)Cp[i + 1];
    for (int32_t x = Xp[i], xe = Xp[i + 1]; x < xe; ++x) {
        const int32_t j = Xi[x];
        if (j < 0 || j >= n_mid) { *bad = 2; return; }
        for (int32_t y = Yp[j], ye = Yp[j + 1]; y < ye; ++y) {
            const int32_t J = Yi[y];
            int64_t c = -1;
            if (dense_cols) {
                if (J >= 0 && J < dense_cols) c = c0 + J;
            } else {
                int64_t lo = c0, hi = c1;                      // first entry of the row with column >= J
                while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (Ci[m] < J) lo = m + 1; else hi = m; }
                if (lo < c1 && Ci[lo] == J) c = lo;
            }
            if (c < 0) { *bad = 1; return; }
            if (!FILL) { ++cnt[c]; continue; }
            const int32_t t = tptr[c] + cnt[c]++;
            terms[t] = X_CONST ? Term{cv[x], y, 0} : Term{cv[y], x, 0};
        }
    }
}
// LPE lanes per output entry: lane l takes the entry's terms l, l + LPE, ... (the lanes of an entry read consecutive 16-byte
// terms: with one lane per entry a wave's loads were 64 lines apart and the kernel ran at 1 TB/s of its 3.4 GB), the partial
// sums are folded in a fixed butterfly — deterministic, the association differs from the plan's sequential order by rounding.
template <int LPE>
__global__ void __launch_bounds__(TPB)
product_kernel(int64_t n_out, const int32_t* __restrict__ tptr, const Term* __restrict__ terms, const double* __restrict__ V,
               double* __restrict__ out) {
    const int sub = threadIdx.x & (LPE - 1);
    const int64_t c = ((int64_t)blockIdx.x * TPB + threadIdx.x) / LPE;
    double a = 0.0;
    if (c < n_out) {
        for (int32_t t = tptr[c] + sub, e = tptr[c + 1]; t < e; t += LPE) {
            const int4 w = *reinterpret_cast<const int4*>(terms + t);
            a += __hiloint2double(w.y, w.x) * V[w.z];
        }
    }
#pragma unroll
    for (int o = 1; o < LPE; o <<= 1) a += __shfl_xor(a, o, 64);
    if (c < n_out && sub == 0) out[c] = a;
}

// D = inverse of the 3x3 diagonal blocks of a coarse operator, as solver.py builds it (_block_diag_inverse: an empty
// row / column — a rotation nobody interpolates from — gets a unit diagonal; 1e-13 of the mean diagonal is added)
__global__ void __launch_bounds__(TPB)
block3_inverse_kernel(int64_t n_nodes, const int32_t* __restrict__ d9, const double* __restrict__ A, double* __restrict__ D) {
    const int64_t n = (int64_t)blockIdx.x * TPB + threadIdx.x;
    if (n >= n_nodes) return;
    double m[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { const int32_t q = d9[9 * n + i]; m[i] = q >= 0 ? A[q] : 0.0; }
#pragma unroll
    for (int b = 0; b < 3; ++b) if (m[4 * b] == 0.0) m[4 * b] = 1.0;
    const double eps = 1e-13 * fabs(m[0] + m[4] + m[8]) / 3.0;
    m[0] += eps; m[4] += eps; m[8] += eps;
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double r = 1.0 / det;
    double* o = D + 9 * n;
    o[0] = c00 * r; o[1] = (m[2] * m[7] - m[1] * m[8]) * r; o[2] = (m[1] * m[5] - m[2] * m[4]) * r;
    o[3] = c01 * r; o[4] = (m[0] * m[8] - m[2] * m[6]) * r; o[5] = (m[2] * m[3] - m[0] * m[5]) * r;
    o[6] = c02 * r; o[7] = (m[1] * m[6] - m[0] * m[7]) * r; o[8] = (m[0] * m[4] - m[1] * m[3]) * r;
}

// Coarse levels under the refresh: three DOFs per node and the three rows of a node stored back to back on the SAME block
// columns (the refresh pads its product pattern to whole 3x3 blocks), so one group of 8 lanes forms all three sums of a
// node from one pass over its block columns (ids and x fetched once) and applies the node's 3x3 block-Jacobi inverse on the
// spot — the smoothing step that took an operator pass and a D pass is one kernel:
//   MODE 0: out = b - A x      MODE 1: out = x + om D (b - A x)      MODE 2: out = ca x + cp xp + om D (b - A x)
// out never aliases x (neighbours read it); it may alias xp.
// AV = float: the operator's single-precision copy (to_float_kernel after every refresh; the smoother need not see more than
// seven digits of it — level 1 at 1 M mesh DOFs: 30 instead of 60 MB per pass, four passes per cycle).
// PASSES nodes per lane group (independent accumulators, loads in flight): 4 on big levels, 1 where four would leave the chip
// with a few hundred workgroups (level 1 at 1 M mesh DOFs: 56 k nodes = 439 workgroups of 128 nodes).
template <int MODE, typename AV, int PASSES>
__global__ void __launch_bounds__(TPB)
node3_kernel(int64_t n_nodes, const int32_t* __restrict__ nbp, const int32_t* __restrict__ nbc, const AV* __restrict__ A,
             const double* __restrict__ D, const double* __restrict__ x, const double* __restrict__ b, double om, double ca,
             double cp, const double* xp, double* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & 7, grp = lane >> 3;
    const int64_t node0 = (int64_t)blockIdx.x * ((TPB / 64) * NODES_PER_WAVE * PASSES) + (int64_t)wave * (NODES_PER_WAVE * PASSES) + grp;
    double acc[PASSES][3];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int64_t n = node0 + ps * NODES_PER_WAVE;
        acc[ps][0] = acc[ps][1] = acc[ps][2] = 0.0;
        if (n < n_nodes) {
            const int b0 = nbp[n], deg = nbp[n + 1] - b0;
            const AV* r0 = A + 9 * (int64_t)b0;
            const AV* r1 = r0 + 3 * deg;
            const AV* r2 = r1 + 3 * deg;
            for (int t = sub; t < deg; t += 8) {
                const double* xv = x + 3 * (int64_t)nbc[b0 + t];
                const double x0 = xv[0], x1 = xv[1], x2 = xv[2];
                acc[ps][0] += (double)r0[3 * t] * x0 + (double)r0[3 * t + 1] * x1 + (double)r0[3 * t + 2] * x2;
                acc[ps][1] += (double)r1[3 * t] * x0 + (double)r1[3 * t + 1] * x1 + (double)r1[3 * t + 2] * x2;
                acc[ps][2] += (double)r2[3 * t] * x0 + (double)r2[3 * t + 1] * x1 + (double)r2[3 * t + 2] * x2;
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        double a0 = acc[ps][0], a1 = acc[ps][1], a2 = acc[ps][2];
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); }
        const int64_t n = node0 + ps * NODES_PER_WAVE;
        if (sub != 0 || n >= n_nodes) continue;
        const double q0 = b[3 * n] - a0, q1 = b[3 * n + 1] - a1, q2 = b[3 * n + 2] - a2;
        if (MODE == 0) { out[3 * n] = q0; out[3 * n + 1] = q1; out[3 * n + 2] = q2; continue; }
        const double* d = D + 9 * n;
        const double d0 = d[0] * q0 + d[1] * q1 + d[2] * q2, d1 = d[3] * q0 + d[4] * q1 + d[5] * q2, d2 = d[6] * q0 + d[7] * q1 + d[8] * q2;
        double o0 = x[3 * n], o1 = x[3 * n + 1], o2 = x[3 * n + 2];
        if (MODE == 2) {
            o0 *= ca; o1 *= ca; o2 *= ca;
            if (xp) { o0 += cp * xp[3 * n]; o1 += cp * xp[3 * n + 1]; o2 += cp * xp[3 * n + 2]; }
        }
        out[3 * n] = o0 + om * d0; out[3 * n + 1] = o1 + om * d1; out[3 * n + 2] = o2 + om * d2;
    }
}

// The bottom of the V-cycle in ONE workgroup: the last smoothed level (<= kTailNodes nodes) and the coarsest solve under it —
// eight launches of a few microseconds each for a few hundred nodes (pre-smoothing x1, x2, residual, restriction, dense solve,
// prolongation, post-smoothing x1, x2): at 1 M mesh DOFs 8 of the iteration's 38 launches and an eighth of its time.  The
// level's vectors live in LDS, the operator (single precision, node3_kernel's layout) and the transfers' node blocks are read
// from global memory (L2-resident: 146 KB) in every phase.  Same arithmetic as the launches it replaces: 8 lanes per node for
// the operator passes, 8 per coarse node for the restriction, one per fine node for the prolongation.
constexpr int kTailNodes = 384;            // three rounds of 128 nodes in a 1024-thread workgroup
constexpr int kTailCoarse = 128;           // DOFs of the coarsest level
struct TailArgs {
    int n_nodes, n_c;                      // nodes of the level (3 DOFs each), DOFs of the coarsest level
    const int32_t *nbp, *nbc; const float* A; const double* D;        // operator in node blocks, 3x3 block-Jacobi inverses
    const int32_t *rptr, *rcol; const float* rval;                    // restriction to the coarsest level (Level::Blocks, BF = 3)
    const int32_t *pptr, *pcol; const float* pval;                    // prolongation from it
    const double* Ainv;                                               // dense inverse, row-major n_c x n_c
    double c1, a2, cp, w2;                                            // Chebyshev coefficients of the level
    const double* b; double* out;                                     // right-hand side in, smoothed iterate out (3 n_nodes)
};

// MODE as node3_kernel: 0: out = b - A x;  1: out = x + om D (b - A x);  2: out = ca x + cp xp + om D (b - A x)
template <int MODE>
__device__ inline void tail_pass(const TailArgs& a, const double* __restrict__ x, const double* __restrict__ b, double om, double ca,
                                 double cp, const double* xp, double* __restrict__ out) {
    const int sub = threadIdx.x & 7;
    for (int n = threadIdx.x >> 3; n < ((a.n_nodes + 127) & ~127); n += 128) {          // (whole waves stay in the loop: shuffles)
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        if (n < a.n_nodes) {
            const int b0 = a.nbp[n], deg = a.nbp[n + 1] - b0;
            const float* r0 = a.A + 9 * (int64_t)b0;
            const float* r1 = r0 + 3 * deg;
            const float* r2 = r1 + 3 * deg;
            for (int t = sub; t < deg; t += 8) {
                const double* xv = x + 3 * a.nbc[b0 + t];
                const double x0 = xv[0], x1 = xv[1], x2 = xv[2];
                s0 += (double)r0[3 * t] * x0 + (double)r0[3 * t + 1] * x1 + (double)r0[3 * t + 2] * x2;
                s1 += (double)r1[3 * t] * x0 + (double)r1[3 * t + 1] * x1 + (double)r1[3 * t + 2] * x2;
                s2 += (double)r2[3 * t] * x0 + (double)r2[3 * t + 1] * x1 + (double)r2[3 * t + 2] * x2;
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (sub != 0 || n >= a.n_nodes) continue;
        const double q0 = b[3 * n] - s0, q1 = b[3 * n + 1] - s1, q2 = b[3 * n + 2] - s2;
        if (MODE == 0) { out[3 * n] = q0; out[3 * n + 1] = q1; out[3 * n + 2] = q2; continue; }
        const double* d = a.D + 9 * n;
        const double d0 = d[0] * q0 + d[1] * q1 + d[2] * q2, d1 = d[3] * q0 + d[4] * q1 + d[5] * q2, d2 = d[6] * q0 + d[7] * q1 + d[8] * q2;
        double o0 = x[3 * n], o1 = x[3 * n + 1], o2 = x[3 * n + 2];
        if (MODE == 2) {
            o0 *= ca; o1 *= ca; o2 *= ca;
            if (xp) { o0 += cp * xp[3 * n]; o1 += cp * xp[3 * n + 1]; o2 += cp * xp[3 * n + 2]; }
        }
        out[3 * n] = o0 + om * d0; out[3 * n + 1] = o1 + om * d1; out[3 * n + 2] = o2 + om * d2;
    }
}

__global__ void __launch_bounds__(1024)
tail_kernel(TailArgs a) {
    __shared__ double B[3 * kTailNodes], X[3 * kTailNodes], T[3 * kTailNodes], R[3 * kTailNodes];
    __shared__ double bc[kTailCoarse], xc[kTailCoarse];
    const int nd = 3 * a.n_nodes;
    for (int i = threadIdx.x; i < nd; i += 1024) B[i] = a.b[i];
    __syncthreads();
    // x1 = c1 D b
    for (int n = threadIdx.x; n < a.n_nodes; n += 1024) {
        const double* d = a.D + 9 * n;
        const double q0 = B[3 * n], q1 = B[3 * n + 1], q2 = B[3 * n + 2];
        X[3 * n] = a.c1 * (d[0] * q0 + d[1] * q1 + d[2] * q2);
        X[3 * n + 1] = a.c1 * (d[3] * q0 + d[4] * q1 + d[5] * q2);
        X[3 * n + 2] = a.c1 * (d[6] * q0 + d[7] * q1 + d[8] * q2);
    }
    __syncthreads();
    tail_pass<2>(a, X, B, a.w2, a.a2, 0.0, nullptr, T);                 // x2 = a2 x1 + w2 D (b - A x1)
    __syncthreads();
    tail_pass<0>(a, T, B, 0.0, 0.0, 0.0, nullptr, R);                   // r = b - A x2
    __syncthreads();
    {   // bc = R r: 8 lanes per coarse node
        const int sub = threadIdx.x & 7, ncn = a.n_c / 3;
        for (int J = threadIdx.x >> 3; J < ((ncn + 7) & ~7); J += 128) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
            if (J < ncn)
                for (int t = a.rptr[J] + sub, e = a.rptr[J + 1]; t < e; t += 8) {
                    const double* ri = R + 3 * a.rcol[t];
                    const float* v = a.rval + 9 * (int64_t)t;
#pragma unroll
                    for (int q = 0; q < 3; ++q) { const double rv = ri[q]; s0 += (double)v[q] * rv; s1 += (double)v[3 + q] * rv; s2 += (double)v[6 + q] * rv; }
                }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
            if (J < ncn && sub == 0) { bc[3 * J] = s0; bc[3 * J + 1] = s1; bc[3 * J + 2] = s2; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.n_c; i += 1024) {                   // xc = A^-1 bc
        const double* row = a.Ainv + (int64_t)i * a.n_c;
        double v = 0.0;
        for (int j = 0; j < a.n_c; ++j) v += row[j] * bc[j];
        xc[i] = v;
    }
    __syncthreads();
    for (int n = threadIdx.x; n < a.n_nodes; n += 1024) {               // x0 = x2 + P xc
        double p0 = 0.0, p1 = 0.0, p2 = 0.0;
        for (int t = a.pptr[n], e = a.pptr[n + 1]; t < e; ++t) {
            const double* xj = xc + 3 * a.pcol[t];
            const float* v = a.pval + 9 * (int64_t)t;
            p0 += (double)v[0] * xj[0] + (double)v[1] * xj[1] + (double)v[2] * xj[2];
            p1 += (double)v[3] * xj[0] + (double)v[4] * xj[1] + (double)v[5] * xj[2];
            p2 += (double)v[6] * xj[0] + (double)v[7] * xj[1] + (double)v[8] * xj[2];
        }
        T[3 * n] += p0; T[3 * n + 1] += p1; T[3 * n + 2] += p2;
    }
    __syncthreads();
    tail_pass<1>(a, T, B, a.c1, 1.0, 0.0, nullptr, X);                  // x1 = x0 + c1 D (b - A x0)
    __syncthreads();
    tail_pass<2>(a, X, B, a.w2, a.a2, a.cp, T, R);                      // x2 = a2 x1 + cp x0 + w2 D (b - A x1)
    __syncthreads();
    for (int i = threadIdx.x; i < nd; i += 1024) a.out[i] = R[i];
}

// Coarsest operator (n <= kDenseMax, symmetric positive definite after the shift solver.py applies: 1e-10 of its largest
// entry on the diagonal): inverted in place by one workgroup, Gauss-Jordan without pivoting, a fixed order.
constexpr int kDenseMax = 256;             // (one workgroup: 1 ms at 128 DOFs, 7 ms at 256, 0.3 s at 900)
__global__ void __launch_bounds__(1024)
dense_inverse_kernel(int n, double* __restrict__ A) {
    __shared__ double col[kDenseMax], row[kDenseMax];
    __shared__ double sh[16];
    double mx = 0.0;
    for (int i = threadIdx.x; i < n * n; i += 1024) mx = fmax(mx, fabs(A[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = 0.0;
    for (int i = 0; i < 16; ++i) mx = fmax(mx, sh[i]);
    for (int i = threadIdx.x; i < n; i += 1024) A[(size_t)i * n + i] += 1e-10 * mx;
    __syncthreads();
    for (int p = 0; p < n; ++p) {
        const double d = 1.0 / A[(size_t)p * n + p];
        __syncthreads();                                           // (everybody has read the pivot)
        for (int i = threadIdx.x; i < n; i += 1024) { col[i] = A[(size_t)i * n + p]; row[i] = A[(size_t)p * n + i] * d; }
        __syncthreads();
        for (int e = threadIdx.x; e < n * n; e += 1024) {
            const int i = e / n, j = e - i * n;
            double v;
            if (i == p) v = j == p ? d : row[j];
            else if (j == p) v = -col[i] * d;
            else v = A[e] - col[i] * row[j];
            A[e] = v;
        }
        __syncthreads();
    }
}

}  // namespace

struct fep_solver {
    int device = 0;
    int64_t n_n = 0, n_dof = 0, n_blk = 0, n_free = 0;
    int32_t *nptr = nullptr, *ncol = nullptr;
    uint8_t* free_dof = nullptr;
    double *minv = nullptr, *r = nullptr, *u = nullptr, *w = nullptr, *p = nullptr, *s = nullptr;
    double *part_g = nullptr, *part_r = nullptr, *part_d = nullptr;
    Scal* scal = nullptr;
    int n_vec_blocks = 0, n_mv_blocks = 0;
    // multigrid preconditioner: level k (k = 0 is the mesh) -> level k+1
    struct Csr { int64_t n_rows = 0, nnz = 0; int32_t *indptr = nullptr, *indices = nullptr; double* vals = nullptr; };
    struct Plan { int64_t n_out = 0, n_terms = 0; int32_t* tptr = nullptr; void* terms = nullptr; };      // terms: (coefficient, value index) x 16 bytes
    struct Level {
        int64_t n_fine = 0, n_coarse = 0;
        Csr P, R, A, D;                       // A, D: operator of level k+1 (A = its inverse when last) and its block-Jacobi inverse
        double omega = 0.0;                   // damping of the smoother on level k
        bool last = false;
        double *x = nullptr, *b = nullptr, *r = nullptr, *t = nullptr;      // vectors of level k+1 (t: Chebyshev smoother)
        std::vector<int32_t> hPp, hPi, hRp, hRi;                            // host patterns of the transfers (refresh plans)
        // refresh: T = A_k P (values only), A_{k+1} = R T, positions of the 3x3 diagonal blocks in A_{k+1}
        Plan ap, rt;
        double* T = nullptr;
        float* A32 = nullptr;                                               // single-precision copy of A's values for node3_kernel
        int32_t* d9 = nullptr;
        int32_t *nbp = nullptr, *nbc = nullptr;                             // node blocks of the padded A (node3_kernel)
        double* xcur = nullptr;                                             // which of x / t / r holds the level's iterate
        // The transfers once more for the V-cycle, in node blocks and single precision (the refresh plans keep the double-precision
        // CSR): a fine node (BF = 2 DOFs on the mesh level, 3 below) against a coarse node (3 DOFs) is one BF x 3 block — one
        // column id per block instead of one per entry, 4-byte values: P_0 at 1 M DOFs 36 instead of 92 MB per pass.
        struct Blocks {
            int bf = 0; int64_t nf = 0, nc = 0, nblk = 0;
            int32_t *pptr = nullptr, *pcol = nullptr; float* pval = nullptr;   // per fine node: (coarse node, BF x 3 values, row-major)
            int32_t *rptr = nullptr, *rcol = nullptr; float* rval = nullptr;   // per coarse node: (fine node, 3 x BF values) = the transposes
        } tb;
    };
    std::vector<int32_t> ip0, ix0;            // scalar pattern of K on the host (refresh plans)
    bool refresh = false;                     // every multigrid solve re-projects the coarse operators from its tangent
    std::vector<Level> levels;
    double *t0 = nullptr, *q = nullptr;       // level-0 residual of the V-cycle, q = K p
    // single-precision copy of the solve's K for the V-cycle's four level-0 passes (the preconditioner need not see more than
    // seven digits of the matrix; CG's own product stays in double precision).  Measured at 1 M DOFs on the LOAD STEPS of BASELINE
    // configs[3], same session, twice each: 5.93 / 5.95 s with it, 6.20 / 6.43 s without, iteration counts 10 873 / 10 910 (the
    // first comparison of the round looked at the whole wall, set-up jitter included, and saw 1 %).  On; FEP_AMG_FP32=0 turns it off.
    float* k32 = nullptr;
    bool fp32 = true;
    bool tail = true;                         // bottom of the V-cycle in one workgroup (tail_kernel; FEP_AMG_TAIL=0: the launches it replaces)
    bool block_transfers = true;              // Level::Blocks for the V-cycle's transfers (FEP_AMG_BLOCK_TRANSFERS=0: the CSR forms)
    // smoother of the V-cycle: degree-2 Chebyshev (default) or two damped block-Jacobi sweeps (FEP_AMG_SMOOTHER=jacobi)
    bool cheb = true;
    double cheb_alpha = 20.0, cheb_safety = 1.2;          // K_elast at 1 M DOFs: alpha 5 / 10 / 20 / 30 -> 77 / 70 / 67 / 66 iterations (Jacobi: 90)
};

static void free_csr(fep_solver::Csr& m) {
    if (m.indptr) (void)hipFree(m.indptr);
    if (m.indices) (void)hipFree(m.indices);
    if (m.vals) (void)hipFree(m.vals);
    m = fep_solver::Csr();
}
static void free_plan(fep_solver::Plan& p) {
    if (p.tptr) (void)hipFree(p.tptr);
    if (p.terms) (void)hipFree(p.terms);
    p = fep_solver::Plan();
}
static void free_blocks(fep_solver::Level::Blocks& b) {
    for (void* v : {(void*)b.pptr, (void*)b.pcol, (void*)b.pval, (void*)b.rptr, (void*)b.rcol, (void*)b.rval}) if (v) (void)hipFree(v);
    b = fep_solver::Level::Blocks();
}
static void free_levels(fep_solver* s) {
    for (auto& l : s->levels) {
        free_csr(l.P); free_csr(l.R); free_csr(l.A); free_csr(l.D);
        for (double* v : {l.x, l.b, l.r, l.t, l.T}) if (v) (void)hipFree(v);
        if (l.A32) (void)hipFree(l.A32);
        free_plan(l.ap); free_plan(l.rt);
        for (int32_t* v : {l.d9, l.nbp, l.nbc}) if (v) (void)hipFree(v);
        free_blocks(l.tb);
    }
    s->levels.clear();
    s->refresh = false;
}

extern "C" int fep_solver_destroy(fep_solver* s) {
    if (!s) return FEP_OK;
    if (fep_set_device(s->device) == FEP_OK) {
        void* ptrs[] = {s->nptr, s->ncol, s->free_dof, s->minv, s->r, s->u, s->w, s->p, s->s,
                        s->part_g, s->part_r, s->part_d, s->scal};
        for (void* q : ptrs)
            if (q) (void)hipFree(q);
        if (s->t0) (void)hipFree(s->t0);
        if (s->q) (void)hipFree(s->q);
        if (s->k32) (void)hipFree(s->k32);
        free_levels(s);
    }
    delete s;
    return FEP_OK;
}

static int solver_create_impl(fep_solver** out, int device_id, int64_t n_n, const int32_t* indptr_h,
                              const int32_t* indices_h, const uint8_t* free_dof_h);

extern "C" int fep_solver_create(fep_solver** out, int device_id, int64_t n_n, const int32_t* indptr_h,
                                 const int32_t* indices_h, const uint8_t* free_dof_h) {
    try {                                               // host vectors of the pattern check: no exception leaves the C ABI
        return solver_create_impl(out, device_id, n_n, indptr_h, indices_h, free_dof_h);
    } catch (const std::bad_alloc&) {
        return FEP_ENOMEM;
    } catch (...) {
        return FEP_EINVAL;
    }
}

static int solver_create_impl(fep_solver** out, int device_id, int64_t n_n, const int32_t* indptr_h,
                              const int32_t* indices_h, const uint8_t* free_dof_h) {
    if (!out) return FEP_EINVAL;
    *out = nullptr;
    if (n_n <= 0 || !indptr_h || !indices_h || !free_dof_h) return FEP_EINVAL;
    const int64_t n_dof = 2 * n_n;
    // node graph from the DOF pattern; the layout contract is checked, not assumed
    std::vector<int32_t> nptr((size_t)n_n + 1, 0);
    if (indptr_h[0] != 0) return FEP_EINVAL;
    for (int64_t n = 0; n < n_n; ++n) {
        const int64_t a = indptr_h[2 * n], b = indptr_h[2 * n + 1], c = indptr_h[2 * n + 2];
        if (b - a != c - b || ((b - a) & 1) || b < a) return FEP_EINVAL;
        if (a != 4 * (int64_t)nptr[n]) return FEP_EINVAL;
        nptr[n + 1] = nptr[n] + (int32_t)((b - a) / 2);
    }
    const int64_t n_blk = nptr[n_n];
    std::vector<int32_t> ncol((size_t)n_blk);
    for (int64_t n = 0; n < n_n; ++n) {
        const int32_t deg = nptr[n + 1] - nptr[n];
        const int32_t* r0 = indices_h + indptr_h[2 * n];
        const int32_t* r1 = indices_h + indptr_h[2 * n + 1];
        for (int32_t t = 0; t < deg; ++t) {
            const int32_t c0 = r0[2 * t];
            if ((c0 & 1) || r0[2 * t + 1] != c0 + 1 || r1[2 * t] != c0 || r1[2 * t + 1] != c0 + 1) return FEP_EINVAL;
            if (c0 < 0 || c0 / 2 >= n_n) return FEP_ERANGE;
            ncol[(size_t)nptr[n] + t] = c0 / 2;
        }
    }
    FEP_TRY(fep_set_device(device_id));
    fep_solver* s = new (std::nothrow) fep_solver();
    if (!s) return FEP_ENOMEM;
    if (const char* sm = fep_tune("FEP_AMG_SMOOTHER")) s->cheb = std::strcmp(sm, "jacobi") != 0;
    if (const char* al = fep_tune("FEP_AMG_CHEB_ALPHA")) { const double v = std::atof(al); if (v > 1.0) s->cheb_alpha = v; }
    if (const char* f32 = fep_tune("FEP_AMG_FP32")) s->fp32 = std::strcmp(f32, "0") != 0;
    if (const char* bt = fep_tune("FEP_AMG_BLOCK_TRANSFERS")) s->block_transfers = std::strcmp(bt, "0") != 0;
    if (const char* tl = fep_tune("FEP_AMG_TAIL")) s->tail = std::strcmp(tl, "0") != 0;
    if (const char* sf = fep_tune("FEP_AMG_CHEB_SAFETY")) { const double v = std::atof(sf); if (v >= 1.0) s->cheb_safety = v; }
    s->device = device_id; s->n_n = n_n; s->n_dof = n_dof; s->n_blk = n_blk;
    s->ip0.assign(indptr_h, indptr_h + n_dof + 1);
    s->ix0.assign(indices_h, indices_h + indptr_h[n_dof]);
    for (int64_t i = 0; i < n_dof; ++i) s->n_free += free_dof_h[i] != 0;
    s->n_vec_blocks = (int)((n_n + TPB - 1) / TPB);
    s->n_mv_blocks = (int)((n_n + NODES_PER_BLOCK - 1) / NODES_PER_BLOCK);
    int rc = FEP_OK;
    auto up = [&](void** dst, const void* src, size_t bytes) {
        if (rc != FEP_OK) return;
        hipError_t e = hipMalloc(dst, bytes ? bytes : 8);
        if (e == hipSuccess && src) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { fep_g_last_hip = (int)e; (void)hipGetLastError(); rc = e == hipErrorOutOfMemory ? FEP_ENOMEM : FEP_EHIP; }
    };
    up((void**)&s->nptr, nptr.data(), nptr.size() * sizeof(int32_t));
    up((void**)&s->ncol, ncol.data(), ncol.size() * sizeof(int32_t));
    up((void**)&s->free_dof, free_dof_h, (size_t)n_dof);
    up((void**)&s->minv, nullptr, (size_t)n_n * 3 * sizeof(double));
    for (double** v : {&s->r, &s->u, &s->w, &s->p, &s->s}) up((void**)v, nullptr, (size_t)n_dof * sizeof(double));
    up((void**)&s->part_g, nullptr, (size_t)s->n_vec_blocks * sizeof(double));
    up((void**)&s->part_r, nullptr, (size_t)s->n_vec_blocks * sizeof(double));
    up((void**)&s->part_d, nullptr, (size_t)s->n_mv_blocks * sizeof(double));
    up((void**)&s->scal, nullptr, sizeof(Scal));
    if (rc != FEP_OK) { fep_solver_destroy(s); return rc; }
    *out = s;
    return FEP_OK;
}

extern "C" int fep_solver_sizes(const fep_solver* s, int64_t sizes[4]) {
    if (!s || !sizes) return FEP_EINVAL;
    sizes[0] = s->n_n; sizes[1] = s->n_dof; sizes[2] = 4 * s->n_blk; sizes[3] = s->n_free;
    return FEP_OK;
}

extern "C" int fep_solver_spmv_dev(fep_solver* s, void* stream, const double* k_data_d, const double* x_d,
                                   double* y_d, int masked) {
    if (!s || !k_data_d || !x_d || !y_d || x_d == y_d) return FEP_EINVAL;
    if (!fep_aligned16(k_data_d) || !fep_aligned16(x_d) || !fep_aligned16(y_d)) return FEP_EINVAL;
    FEP_TRY(fep_set_device(s->device));
    hipStream_t st = (hipStream_t)stream;
    if (masked)
        hipLaunchKernelGGL(spmv_kernel<true>, dim3(s->n_mv_blocks), dim3(TPB), 0, st, s->n_n, s->nptr, s->ncol, s->free_dof,
                           (const double2*)k_data_d, (const double2*)x_d, y_d, (const double*)nullptr, (double*)nullptr);
    else
        hipLaunchKernelGGL(spmv_kernel<false>, dim3(s->n_mv_blocks), dim3(TPB), 0, st, s->n_n, s->nptr, s->ncol, s->free_dof,
                           (const double2*)k_data_d, (const double2*)x_d, y_d, (const double*)nullptr, (double*)nullptr);
    HIP_TRY(hipGetLastError());
    return FEP_OK;
}

namespace {
// How many iterations to enqueue before the host looks at the device-side state again: check_every at most, fewer when the
// residual history says the stopping test is that close.  A solve that has converged is frozen (alpha = beta = 0) but the
// iterations already enqueued still run their passes: with a fixed batch of 10 a two-digit multigrid solve of ~80 iterations
// threw 4.5 of them away on average.  Undershooting costs one more 48-byte read-back, overshooting whole iterations, hence 3/4.
inline int next_batch(int check_every, double rr_prev, int n_prev, double rr, double target) {
    static const bool fixed = fep_tune("FEP_PCG_FIXED_BATCH") != nullptr;      // A/B switch: always check_every
    if (fixed) return check_every;
    if (!(rr_prev > 0.0) || !(rr > 0.0) || n_prev <= 0 || !(rr < rr_prev) || !(target > 0.0)) return check_every;
    if (rr <= target) return 1;
    const double per_it = std::log(rr / rr_prev) / n_prev;             // < 0
    const double need = 0.75 * std::log(target / rr) / per_it;
    if (!(need < (double)check_every)) return check_every;
    const int n = (int)std::ceil(need);
    return n < 1 ? 1 : n;
}
}  // namespace

extern "C" int fep_solver_pcg_dev(fep_solver* s, void* stream, const double* k_data_d, const double* b_d, double* x_d,
                                  double rtol, int max_iter, int check_every, int* iters_out, double* relres_out,
                                  int* state_out) {
    if (!s || !k_data_d || !b_d || !x_d || !(rtol >= 0.0) || max_iter < 0) return FEP_EINVAL;
    if (!fep_aligned16(k_data_d) || !fep_aligned16(b_d) || !fep_aligned16(x_d)) return FEP_EINVAL;
    if (check_every <= 0) check_every = 50;
    FEP_TRY(fep_set_device(s->device));
    hipStream_t st = (hipStream_t)stream;
    const dim3 gv(s->n_vec_blocks), gm(s->n_mv_blocks), tb(TPB);
    const double tol2 = rtol * rtol;
    double2 *x = (double2*)x_d, *r = (double2*)s->r, *u = (double2*)s->u, *p = (double2*)s->p, *sv = (double2*)s->s;
    auto spmv_dot = [&]() {
        hipLaunchKernelGGL(spmv_kernel<true>, gm, tb, 0, st, s->n_n, s->nptr, s->ncol, s->free_dof,
                           (const double2*)k_data_d, (const double2*)s->u, s->w, (const double*)s->u, s->part_d);
    };
    hipLaunchKernelGGL(block_jacobi_kernel, gv, tb, 0, st, s->n_n, s->nptr, s->ncol, s->free_dof, k_data_d, s->minv);
    hipLaunchKernelGGL(pcg_init_kernel, gv, tb, 0, st, s->n_n, (const double2*)b_d, s->free_dof, s->minv, x, r, u, p, sv,
                       s->part_g, s->part_r);
    spmv_dot();
    hipLaunchKernelGGL(pcg_scalar_kernel, dim3(1), dim3(1024), 0, st, s->scal, s->part_g, s->part_r, s->n_vec_blocks,
                       s->part_d, s->n_mv_blocks, tol2, 1);
    HIP_TRY(hipGetLastError());
    Scal h;
    std::memset(&h, 0, sizeof h);
    int launched = 0, n_prev = 0;
    double rr_prev = 0.0;
    for (;;) {
        HIP_TRY(hipMemcpyAsync(&h, s->scal, sizeof h, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h.state != 0 || launched >= max_iter) break;
        const int n = std::min(next_batch(check_every, rr_prev, n_prev, h.rr, tol2 * h.bb), max_iter - launched);
        rr_prev = h.rr; n_prev = n;
        for (int i = 0; i < n; ++i) {
            hipLaunchKernelGGL(pcg_update_kernel, gv, tb, 0, st, s->n_n, s->scal, (const double2*)s->w, s->minv, x, r, u, p,
                               sv, s->part_g, s->part_r);
            spmv_dot();
            hipLaunchKernelGGL(pcg_scalar_kernel, dim3(1), dim3(1024), 0, st, s->scal, s->part_g, s->part_r,
                               s->n_vec_blocks, s->part_d, s->n_mv_blocks, tol2, 0);
        }
        HIP_TRY(hipGetLastError());
        launched += n;
    }
    if (iters_out) *iters_out = h.it;
    if (relres_out) *relres_out = h.bb > 0.0 ? std::sqrt(h.rr / h.bb) : 0.0;
    if (state_out) *state_out = h.state;
    return FEP_OK;
}

// ---------------------------------------------------------------------------------------
// Multigrid-preconditioned conjugate gradients
// ---------------------------------------------------------------------------------------
extern "C" int fep_solver_amg_clear(fep_solver* s) {
    if (!s) return FEP_EINVAL;
    FEP_TRY(fep_set_device(s->device));
    free_levels(s);
    return FEP_OK;
}

static int upload_csr(fep_solver::Csr& m, int64_t n_rows, int64_t n_cols, const int32_t* ip, const int32_t* ix,
                      const double* v) {
    if (!ip || !ix || !v || ip[0] != 0) return FEP_EINVAL;
    const int64_t nnz = ip[n_rows];
    for (int64_t i = 0; i < n_rows; ++i)
        if (ip[i + 1] < ip[i]) return FEP_EINVAL;
    for (int64_t t = 0; t < nnz; ++t)
        if (ix[t] < 0 || ix[t] >= n_cols) return FEP_ERANGE;
    m.n_rows = n_rows; m.nnz = nnz;
    HIP_TRY(hipMalloc((void**)&m.indptr, (size_t)(n_rows + 1) * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void**)&m.indices, (size_t)std::max<int64_t>(nnz, 1) * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void**)&m.vals, (size_t)std::max<int64_t>(nnz, 1) * sizeof(double)));
    HIP_TRY(hipMemcpy(m.indptr, ip, (size_t)(n_rows + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m.indices, ix, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m.vals, v, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice));
    return FEP_OK;
}

static int push_level_impl(fep_solver* s, int64_t n_fine, int64_t n_coarse,
                           const int32_t* p_indptr, const int32_t* p_indices, const double* p_vals,
                           const int32_t* r_indptr, const int32_t* r_indices, const double* r_vals,
                           const int32_t* a_indptr, const int32_t* a_indices, const double* a_vals,
                           const int32_t* d_indptr, const int32_t* d_indices, const double* d_vals,
                           double omega_fine, int last);

extern "C" int fep_solver_amg_push_level(fep_solver* s, int64_t n_fine, int64_t n_coarse,
                                         const int32_t* p_indptr, const int32_t* p_indices, const double* p_vals,
                                         const int32_t* r_indptr, const int32_t* r_indices, const double* r_vals,
                                         const int32_t* a_indptr, const int32_t* a_indices, const double* a_vals,
                                         const int32_t* d_indptr, const int32_t* d_indices, const double* d_vals,
                                         double omega_fine, int last) {
    try {                                               // (host copies of the transfers' patterns: no exception leaves the C ABI)
        return push_level_impl(s, n_fine, n_coarse, p_indptr, p_indices, p_vals, r_indptr, r_indices, r_vals, a_indptr, a_indices,
                               a_vals, d_indptr, d_indices, d_vals, omega_fine, last);
    } catch (const std::bad_alloc&) {
        return FEP_ENOMEM;
    } catch (...) {
        return FEP_EINVAL;
    }
}

// Level::Blocks from the prolongation in CSR (host): the blocks of a fine node = the coarse nodes any of its BF rows reaches
// (ascending), entries a row does not hold are zero; the restriction's blocks are the transposes, fine nodes ascending.
// A level whose sizes are not whole nodes keeps the CSR form (tb.bf stays 0).
static int build_transfer_blocks(fep_solver::Level& l, int bf, const int32_t* Pp, const int32_t* Pi, const double* Pv) {
    if (l.n_fine % bf || l.n_coarse % 3) return FEP_OK;
    const int64_t nf = l.n_fine / bf, nc = l.n_coarse / 3;
    // two passes over the fine nodes on the host's cores: blocks per node, then (after the running sum) their ids and values
    std::vector<int32_t> pptr((size_t)nf + 1, 0), pcol;
    std::vector<float> pval;
    std::atomic<int> bad{0};
    auto node_cols = [&](int64_t i, std::vector<int32_t>& cols) {
        cols.clear();
        for (int32_t t = Pp[bf * i]; t < Pp[bf * i + bf]; ++t) {
            if (Pi[t] < 0 || Pi[t] >= l.n_coarse) { bad = 1; return; }
            cols.push_back(Pi[t] / 3);
        }
        std::sort(cols.begin(), cols.end());
        cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
    };
    fep_host::parallel_chunks(nf, [&](int64_t lo, int64_t hi, int) {
        std::vector<int32_t> cols;
        for (int64_t i = lo; i < hi && !bad; ++i) { node_cols(i, cols); pptr[(size_t)i + 1] = (int32_t)cols.size(); }
    });
    if (bad) return FEP_ERANGE;
    for (int64_t i = 0; i < nf; ++i) {
        const int64_t t = (int64_t)pptr[(size_t)i] + pptr[(size_t)i + 1];
        if (t >= INT32_MAX / 9) return FEP_ERANGE;
        pptr[(size_t)i + 1] = (int32_t)t;
    }
    pcol.resize((size_t)pptr[(size_t)nf]);
    pval.assign(pcol.size() * (size_t)(3 * bf), 0.0f);
    fep_host::parallel_chunks(nf, [&](int64_t lo, int64_t hi, int) {
        std::vector<int32_t> cols;
        for (int64_t i = lo; i < hi; ++i) {
            node_cols(i, cols);
            const size_t b0 = (size_t)pptr[(size_t)i];
            std::copy(cols.begin(), cols.end(), pcol.begin() + (std::ptrdiff_t)b0);
            for (int r = 0; r < bf; ++r)
                for (int32_t t = Pp[bf * i + r]; t < Pp[bf * i + r + 1]; ++t) {
                    const size_t b = b0 + (size_t)(std::lower_bound(cols.begin(), cols.end(), Pi[t] / 3) - cols.begin());
                    pval[b * (size_t)(3 * bf) + (size_t)(3 * r + Pi[t] % 3)] = (float)Pv[t];
                }
        }
    });
    const size_t nblk = pcol.size();
    std::vector<int32_t> rptr((size_t)nc + 1, 0), rcol(nblk), fill;
    std::vector<float> rval(nblk * (size_t)(3 * bf));
    for (size_t b = 0; b < nblk; ++b) ++rptr[(size_t)pcol[b] + 1];
    for (int64_t J = 0; J < nc; ++J) rptr[(size_t)J + 1] += rptr[(size_t)J];
    fill.assign(rptr.begin(), rptr.end() - 1);
    for (int64_t i = 0; i < nf; ++i)
        for (int32_t b = pptr[(size_t)i]; b < pptr[(size_t)i + 1]; ++b) {
            const size_t q = (size_t)fill[(size_t)pcol[(size_t)b]]++;
            rcol[q] = (int32_t)i;
            for (int r = 0; r < bf; ++r)
                for (int c = 0; c < 3; ++c) rval[q * (size_t)(3 * bf) + (size_t)(c * bf + r)] = pval[(size_t)b * (size_t)(3 * bf) + (size_t)(3 * r + c)];
        }
    fep_solver::Level::Blocks& B = l.tb;
    auto up = [&](void** dst, const void* src, size_t bytes) {
        if (hipMalloc(dst, std::max<size_t>(bytes, 16)) != hipSuccess) { (void)hipGetLastError(); return FEP_ENOMEM; }
        if (bytes && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return FEP_EHIP; }
        return FEP_OK;
    };
    int rc = up((void**)&B.pptr, pptr.data(), pptr.size() * 4);
    if (rc == FEP_OK) rc = up((void**)&B.pcol, pcol.data(), pcol.size() * 4);
    if (rc == FEP_OK) rc = up((void**)&B.pval, pval.data(), pval.size() * 4);
    if (rc == FEP_OK) rc = up((void**)&B.rptr, rptr.data(), rptr.size() * 4);
    if (rc == FEP_OK) rc = up((void**)&B.rcol, rcol.data(), rcol.size() * 4);
    if (rc == FEP_OK) rc = up((void**)&B.rval, rval.data(), rval.size() * 4);
    if (rc != FEP_OK) { free_blocks(B); return rc; }
    B.bf = bf; B.nf = nf; B.nc = nc; B.nblk = (int64_t)nblk;
    return FEP_OK;
}

static int push_level_impl(fep_solver* s, int64_t n_fine, int64_t n_coarse,
                           const int32_t* p_indptr, const int32_t* p_indices, const double* p_vals,
                           const int32_t* r_indptr, const int32_t* r_indices, const double* r_vals,
                           const int32_t* a_indptr, const int32_t* a_indices, const double* a_vals,
                           const int32_t* d_indptr, const int32_t* d_indices, const double* d_vals,
                           double omega_fine, int last) {
    if (!s || n_fine <= 0 || n_coarse <= 0 || !(omega_fine > 0.0)) return FEP_EINVAL;
    if (!s->levels.empty() && (s->levels.back().last || s->levels.back().n_coarse != n_fine)) return FEP_ESTATE;
    if (s->levels.empty() && n_fine != s->n_dof) return FEP_EINVAL;
    if (!last && !d_indptr) return FEP_EINVAL;
    FEP_TRY(fep_set_device(s->device));
    s->levels.emplace_back();
    fep_solver::Level& l = s->levels.back();
    l.n_fine = n_fine; l.n_coarse = n_coarse; l.omega = omega_fine; l.last = last != 0;
    int rc = upload_csr(l.P, n_fine, n_coarse, p_indptr, p_indices, p_vals);
    if (rc == FEP_OK) {
        l.hPp.assign(p_indptr, p_indptr + n_fine + 1); l.hPi.assign(p_indices, p_indices + p_indptr[n_fine]);
    }
    if (rc == FEP_OK) rc = upload_csr(l.R, n_coarse, n_fine, r_indptr, r_indices, r_vals);
    if (rc == FEP_OK) {
        l.hRp.assign(r_indptr, r_indptr + n_coarse + 1); l.hRi.assign(r_indices, r_indices + r_indptr[n_coarse]);
    }
    if (rc == FEP_OK) rc = upload_csr(l.A, n_coarse, n_coarse, a_indptr, a_indices, a_vals);
    if (rc == FEP_OK && !last) rc = upload_csr(l.D, n_coarse, n_coarse, d_indptr, d_indices, d_vals);
    if (rc == FEP_OK && s->block_transfers) rc = build_transfer_blocks(l, s->levels.size() == 1 ? 2 : 3, p_indptr, p_indices, p_vals);
    for (double** v : {&l.x, &l.b, &l.r, &l.t})
        if (rc == FEP_OK && hipMalloc((void**)v, (size_t)n_coarse * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            rc = FEP_ENOMEM;
        }
    if (rc == FEP_OK && !s->t0) {
        if (hipMalloc((void**)&s->t0, (size_t)s->n_dof * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&s->q, (size_t)s->n_dof * sizeof(double)) != hipSuccess ||
            (s->fp32 && hipMalloc((void**)&s->k32, (size_t)s->n_blk * 4 * sizeof(float)) != hipSuccess)) { (void)hipGetLastError(); rc = FEP_ENOMEM; }
    }
    if (rc != FEP_OK) {
        fep_solver::Level& b = s->levels.back();
        free_csr(b.P); free_csr(b.R); free_csr(b.A); free_csr(b.D);
        free_blocks(b.tb);
        for (double* v : {b.x, b.b, b.r, b.t}) if (v) (void)hipFree(v);
        s->levels.pop_back();
    }
    return rc;
}

// ---------------------------------------------------------------------------------------
// Coarse operators of the CURRENT tangent.  The hierarchy's transfers stay those of the reference matrix; with the refresh
// enabled every multigrid solve first forms A_1 = R_0 K P_0, A_2 = R_1 A_1 P_1, ... from its own K (numeric products on
// patterns fixed at set-up: fep_host.h product_plan), the 3x3 block-Jacobi inverses of the new operators and the dense
// inverse of the coarsest one.  On the tangents of the strip-footing run this takes a third off the iteration counts
// (tools/deflation_study.py; operators one Newton iterate old are worse than those of the elastic matrix).
// Constrained DOFs need no masking: their rows of P (columns of R) are zero.
// ---------------------------------------------------------------------------------------
namespace {

// RAII for the set-up's temporary device arrays
struct DevTmp {
    void* p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)) == hipSuccess ? FEP_OK : ((void)hipGetLastError(), FEP_ENOMEM); }
    int upload(const std::vector<int32_t>& v) {
        FEP_TRY(alloc(v.size() * sizeof(int32_t)));
        if (!v.empty() && hipMemcpy(p, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return FEP_EHIP; }
        return FEP_OK;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

// The terms of C = X * Y on the device (plan_rows_kernel); every pattern is in device memory.  cv_d: values of the constant
// factor (x_const: X).  Leaves out.tptr / out.terms / out.n_out; *n_terms for the log.
int device_plan(int64_t n_rows, int64_t n_mid, const int32_t* Xp, const int32_t* Xi, const int32_t* Yp, const int32_t* Yi,
                const int32_t* Cp, const int32_t* Ci, int64_t dense_cols, int64_t n_out, const double* cv_d, bool x_const,
                fep_solver::Plan& out, size_t* n_terms) {
    if (n_out >= INT32_MAX) return FEP_ERANGE;
    DevTmp cnt, bad;
    FEP_TRY(cnt.alloc((size_t)n_out * sizeof(int32_t)));
    FEP_TRY(bad.alloc(sizeof(int)));
    HIP_TRY(hipMemset(cnt.p, 0, std::max<size_t>((size_t)n_out * sizeof(int32_t), 16)));
    HIP_TRY(hipMemset(bad.p, 0, sizeof(int)));
    const dim3 grid((unsigned)((n_rows + TPB - 1) / TPB)), tb(TPB);
    if (n_rows > 0)
        hipLaunchKernelGGL((plan_rows_kernel<false, false>), grid, tb, 0, 0, n_rows, n_mid, Xp, Xi, Yp, Yi, Cp, Ci, dense_cols,
                           cnt.as<int32_t>(), (const int32_t*)nullptr, (const double*)nullptr, (Term*)nullptr, bad.as<int>());
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> h((size_t)n_out + 1, 0);
    int hb = 0;
    if (n_out > 0) HIP_TRY(hipMemcpy(h.data() + 1, cnt.p, (size_t)n_out * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&hb, bad.p, sizeof(int), hipMemcpyDeviceToHost));
    if (hb) return hb == 1 ? FEP_EINVAL : FEP_ERANGE;
    int64_t tot = 0;
    for (int64_t c = 1; c <= n_out; ++c) {
        tot += h[(size_t)c];
        if (tot >= INT32_MAX) return FEP_ERANGE;
        h[(size_t)c] = (int32_t)tot;
    }
    out.n_out = n_out;
    out.n_terms = tot;
    if (hipMalloc((void**)&out.tptr, h.size() * sizeof(int32_t)) != hipSuccess) { (void)hipGetLastError(); return FEP_ENOMEM; }
    HIP_TRY(hipMemcpy(out.tptr, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (hipMalloc(&out.terms, std::max<size_t>((size_t)tot, 1) * sizeof(Term)) != hipSuccess) { (void)hipGetLastError(); return FEP_ENOMEM; }
    HIP_TRY(hipMemset(cnt.p, 0, std::max<size_t>((size_t)n_out * sizeof(int32_t), 16)));
    if (n_rows > 0 && tot > 0) {
        if (x_const)
            hipLaunchKernelGGL((plan_rows_kernel<true, true>), grid, tb, 0, 0, n_rows, n_mid, Xp, Xi, Yp, Yi, Cp, Ci, dense_cols,
                               cnt.as<int32_t>(), (const int32_t*)out.tptr, cv_d, (Term*)out.terms, bad.as<int>());
        else
            hipLaunchKernelGGL((plan_rows_kernel<true, false>), grid, tb, 0, 0, n_rows, n_mid, Xp, Xi, Yp, Yi, Cp, Ci, dense_cols,
                               cnt.as<int32_t>(), (const int32_t*)out.tptr, cv_d, (Term*)out.terms, bad.as<int>());
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (n_terms) *n_terms = (size_t)tot;
    return FEP_OK;
}

}  // namespace

static int enable_refresh_impl(fep_solver* s) {
    if (s->levels.empty() || !s->levels.back().last) return FEP_ESTATE;
    if (s->refresh) return FEP_OK;
    if (s->levels.back().n_coarse > kDenseMax) return FEP_ERANGE;          // (nothing touched: the hierarchy stays usable)
    FEP_TRY(fep_set_device(s->device));
    // The host fixes the PATTERNS — A_k P_k, R_k (A_k P_k) padded to whole 3x3 node blocks — level by level; the TERMS of both
    // products (290 M at 1 M DOFs) are listed on the device from those patterns (device_plan).  FEP_AMG_PLAN=host lists them on
    // the host as well (fep_host.h product_plan, the form the sanitizer driver replays) and uploads the index pairs: the same
    // term lists, 0.8 s more of set-up at 1 M DOFs.
    const char* pm = fep_tune("FEP_AMG_PLAN");
    const bool host_plan = pm && std::strcmp(pm, "host") == 0;
    int rc = FEP_OK;
    auto up = [&](int32_t** dst, const std::vector<int32_t>& v) {
        if (rc != FEP_OK) return;
        hipError_t e = hipMalloc((void**)dst, std::max<size_t>(v.size(), 1) * sizeof(int32_t));
        if (e == hipSuccess && !v.empty()) e = hipMemcpy(*dst, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipGetLastError(); rc = e == hipErrorOutOfMemory ? FEP_ENOMEM : FEP_EHIP; }
    };
    // host plans: the constant factor's value goes into the term (composed on the device from the index pairs)
    auto up_plan = [&](fep_solver::Plan& d, const fep_host::ProductPlan& h, const double* cv_d, bool first) {
        d.n_out = (int64_t)h.tptr.size() - 1;
        d.n_terms = (int64_t)h.xa.size();
        up(&d.tptr, h.tptr);
        int32_t *xa = nullptr, *ya = nullptr;
        up(&xa, h.xa); up(&ya, h.ya);
        const int64_t n = (int64_t)h.xa.size();
        if (rc == FEP_OK && hipMalloc(&d.terms, std::max<size_t>((size_t)n, 1) * sizeof(Term)) != hipSuccess) { (void)hipGetLastError(); rc = FEP_ENOMEM; }
        if (rc == FEP_OK && n > 0) {
            hipLaunchKernelGGL(compose_terms_kernel, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, 0, n, xa, ya, cv_d, first ? 1 : 0, (Term*)d.terms);
            if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); rc = FEP_EHIP; }
        }
        if (xa) (void)hipFree(xa);
        if (ya) (void)hipFree(ya);
    };
    auto dalloc = [&](double** dst, size_t n) {
        if (rc == FEP_OK && hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); rc = FEP_ENOMEM; }
    };
    std::vector<size_t> n_ap(s->levels.size(), 0), n_rt(s->levels.size(), 0);
    std::vector<int32_t> Xp = s->ip0, Xi = s->ix0;                       // pattern of the operator of level k, host ...
    DevTmp Xp0_d, Xi0_d;                                                 // ... and device (level 0: uploaded for the set-up only)
    if (!host_plan) { rc = Xp0_d.upload(Xp); if (rc == FEP_OK) rc = Xi0_d.upload(Xi); }
    const int32_t *Xp_d = Xp0_d.as<int32_t>(), *Xi_d = Xi0_d.as<int32_t>();
    for (size_t k = 0; k < s->levels.size() && rc == FEP_OK; ++k) {
        fep_solver::Level& l = s->levels[k];
        std::vector<int32_t> Ap, Ai, d9, nbp, nbc, Tp, Ti;
        fep_host::ProductPlan ap, rt;
        rc = fep_host::product_pattern(l.n_fine, l.n_fine, l.n_coarse, Xp.data(), Xi.data(), l.hPp.data(), l.hPi.data(), Tp, Ti);
        if (rc != FEP_OK) break;
        if (host_plan) rc = fep_host::product_plan(l.n_fine, l.n_fine, Xp.data(), Xi.data(), l.hPp.data(), l.hPi.data(), Tp.data(), Ti.data(), 0, ap);
        if (rc != FEP_OK) break;
        if (l.last) {                                                    // the coarsest operator is dense (it is inverted)
            if (host_plan) rc = fep_host::product_plan(l.n_coarse, l.n_fine, l.hRp.data(), l.hRi.data(), Tp.data(), Ti.data(), nullptr, nullptr, l.n_coarse, rt);
            if (rc != FEP_OK) break;
            Ap.resize((size_t)l.n_coarse + 1); Ai.resize((size_t)(l.n_coarse * l.n_coarse));
            for (int64_t i = 0; i <= l.n_coarse; ++i) Ap[(size_t)i] = (int32_t)(i * l.n_coarse);
            for (int64_t i = 0; i < l.n_coarse * l.n_coarse; ++i) Ai[(size_t)i] = (int32_t)(i % l.n_coarse);
        } else {
            if (l.n_coarse % 3 || l.D.nnz != 3 * l.n_coarse) { rc = FEP_EINVAL; break; }
            // the operator of level k+1 moves onto the pattern of the product (SciPy's may have dropped entries)
            rc = fep_host::product_pattern(l.n_coarse, l.n_fine, l.n_coarse, l.hRp.data(), l.hRi.data(), Tp.data(), Ti.data(), Ap, Ai);
            if (rc != FEP_OK) break;
            // padded to whole 3x3 node blocks, the three rows of a node on the same block columns (node3_kernel)
            {
                const int64_t nn = l.n_coarse / 3;
                std::vector<int32_t> Ap2((size_t)l.n_coarse + 1, 0), Ai2, cols;
                nbp.assign((size_t)nn + 1, 0);
                for (int64_t I = 0; I < nn && rc == FEP_OK; ++I) {
                    cols.clear();
                    for (int32_t t = Ap[(size_t)(3 * I)]; t < Ap[(size_t)(3 * I + 3)]; ++t) cols.push_back(Ai[(size_t)t] / 3);
                    std::sort(cols.begin(), cols.end());
                    cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
                    nbc.insert(nbc.end(), cols.begin(), cols.end());
                    if (nbc.size() * 9 >= (size_t)INT32_MAX) rc = FEP_ERANGE;
                    nbp[(size_t)I + 1] = (int32_t)nbc.size();
                    for (int a = 0; a < 3; ++a) {
                        for (int32_t J : cols) { Ai2.push_back(3 * J); Ai2.push_back(3 * J + 1); Ai2.push_back(3 * J + 2); }
                        Ap2[(size_t)(3 * I + a + 1)] = (int32_t)Ai2.size();
                    }
                }
                if (rc != FEP_OK) break;
                Ap.swap(Ap2); Ai.swap(Ai2);
            }
            if (host_plan) rc = fep_host::product_plan(l.n_coarse, l.n_fine, l.hRp.data(), l.hRi.data(), Tp.data(), Ti.data(), Ap.data(), Ai.data(), 0, rt);
            if (rc != FEP_OK) break;
            d9.assign((size_t)l.n_coarse * 3, -1);
            for (int64_t r = 0; r < l.n_coarse; ++r) {
                const int64_t c0 = r - r % 3;
                for (int32_t t = Ap[(size_t)r]; t < Ap[(size_t)r + 1]; ++t)
                    if (Ai[(size_t)t] >= c0 && Ai[(size_t)t] < c0 + 3) d9[(size_t)(3 * r + (Ai[(size_t)t] - c0))] = t;
            }
        }
        // onto the device
        free_csr(l.A);
        l.A.n_rows = l.n_coarse; l.A.nnz = (int64_t)Ai.size();
        up(&l.A.indptr, Ap); up(&l.A.indices, Ai);
        dalloc(&l.A.vals, Ai.size()); dalloc(&l.T, Ti.size());
        if (rc == FEP_OK && s->fp32 && !l.last && hipMalloc((void**)&l.A32, std::max<size_t>(Ai.size(), 4) * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); rc = FEP_ENOMEM; }
        if (host_plan) {
            up_plan(l.ap, ap, l.P.vals, false); up_plan(l.rt, rt, l.R.vals, true);
            n_ap[k] = ap.xa.size(); n_rt[k] = rt.xa.size();
        } else if (rc == FEP_OK) {
            DevTmp Tp_d, Ti_d;
            rc = Tp_d.upload(Tp);
            if (rc == FEP_OK) rc = Ti_d.upload(Ti);
            if (rc == FEP_OK)                                            // T = A_k P_k: the constant factor is the second one
                rc = device_plan(l.n_fine, l.n_fine, Xp_d, Xi_d, l.P.indptr, l.P.indices, Tp_d.as<int32_t>(), Ti_d.as<int32_t>(), 0,
                                 (int64_t)Ti.size(), l.P.vals, false, l.ap, &n_ap[k]);
            if (rc == FEP_OK)                                            // A_{k+1} = R_k T: the first one
                rc = device_plan(l.n_coarse, l.n_fine, l.R.indptr, l.R.indices, Tp_d.as<int32_t>(), Ti_d.as<int32_t>(), l.A.indptr, l.A.indices,
                                 l.last ? l.n_coarse : 0, (int64_t)Ai.size(), l.R.vals, true, l.rt, &n_rt[k]);
        }
        if (!l.last) { up(&l.d9, d9); up(&l.nbp, nbp); up(&l.nbc, nbc); }
        Xp.swap(Ap); Xi.swap(Ai);
        Xp_d = l.A.indptr; Xi_d = l.A.indices;
    }
    // a failure leaves no half-converted hierarchy behind: it is dropped
    if (rc != FEP_OK) { free_levels(s); return rc; }
    if (std::getenv("FEP_VERBOSE")) {
        std::fprintf(stderr, "[fep] multigrid refresh plans (%s):", host_plan ? "host" : "device");
        for (size_t k = 0; k < s->levels.size(); ++k)
            std::fprintf(stderr, " %lld -> %lld DOFs (A P: %lld entries, %zu terms; R (A P): %lld, %zu)", (long long)s->levels[k].n_fine,
                         (long long)s->levels[k].n_coarse, (long long)s->levels[k].ap.n_out, n_ap[k],
                         (long long)s->levels[k].rt.n_out, n_rt[k]);
        std::fprintf(stderr, "\n");
    }
    s->refresh = true;
    return FEP_OK;
}

extern "C" int fep_solver_amg_enable_refresh(fep_solver* s) {
    if (!s) return FEP_EINVAL;
    try {
        return enable_refresh_impl(s);
    } catch (const std::bad_alloc&) {
        return FEP_ENOMEM;
    } catch (...) {
        return FEP_EINVAL;
    }
}

// numeric phase of one product: lanes per output entry by the plan's mean term count (A P: ~7 terms, R (A P): ~19)
static void product_apply(hipStream_t st, const fep_solver::Plan& p, const double* V, double* out) {
    if (p.n_out <= 0) return;
    const double mean = (double)p.n_terms / (double)p.n_out;
    auto grid = [&](int lpe) { return dim3((unsigned)((p.n_out * lpe + TPB - 1) / TPB)); };
    if (mean >= 12.0) hipLaunchKernelGGL(product_kernel<8>, grid(8), dim3(TPB), 0, st, p.n_out, p.tptr, (const Term*)p.terms, V, out);
    else if (mean >= 3.0) hipLaunchKernelGGL(product_kernel<4>, grid(4), dim3(TPB), 0, st, p.n_out, p.tptr, (const Term*)p.terms, V, out);
    else hipLaunchKernelGGL(product_kernel<1>, grid(1), dim3(TPB), 0, st, p.n_out, p.tptr, (const Term*)p.terms, V, out);
}

extern "C" int fep_solver_amg_refresh_dev(fep_solver* s, void* stream, const double* k_data_d) {
    if (!s || !k_data_d) return FEP_EINVAL;
    if (!s->refresh) return FEP_ESTATE;
    FEP_TRY(fep_set_device(s->device));
    hipStream_t st = (hipStream_t)stream;
    const double* A = k_data_d;
    for (fep_solver::Level& l : s->levels) {
        product_apply(st, l.ap, A, l.T);
        product_apply(st, l.rt, (const double*)l.T, l.A.vals);
        if (l.last)
            hipLaunchKernelGGL(dense_inverse_kernel, dim3(1), dim3(1024), 0, st, (int)l.n_coarse, l.A.vals);
        else {
            hipLaunchKernelGGL(block3_inverse_kernel, dim3((unsigned)((l.n_coarse / 3 + TPB - 1) / TPB)), dim3(TPB), 0, st,
                               l.n_coarse / 3, l.d9, l.A.vals, l.D.vals);
            if (l.A32)
                hipLaunchKernelGGL(to_float1_kernel, dim3((unsigned)((l.A.nnz + TPB - 1) / TPB)), dim3(TPB), 0, st, l.A.nnz,
                                   (const double*)l.A.vals, l.A32);
        }
        A = l.A.vals;
    }
    HIP_TRY(hipGetLastError());
    return FEP_OK;
}

namespace {

inline void csr_apply(hipStream_t st, const fep_solver::Csr& m, const double* x, const double* z, double c0, double c1,
                      double* y) {
    if (m.nnz < 12 * m.n_rows)
        hipLaunchKernelGGL((csr_kernel<false, 2>), dim3((unsigned)((m.n_rows * 2 + TPB - 1) / TPB)), dim3(TPB), 0, st, m.n_rows, m.indptr,
                           m.indices, m.vals, x, z, c0, c1, y, (const double*)nullptr, 0.0);
    else
        hipLaunchKernelGGL((csr_kernel<false, 8>), dim3((unsigned)((m.n_rows * 8 + TPB - 1) / TPB)), dim3(TPB), 0, st, m.n_rows, m.indptr,
                           m.indices, m.vals, x, z, c0, c1, y, (const double*)nullptr, 0.0);
}

inline void csr_apply2(hipStream_t st, const fep_solver::Csr& m, const double* x, const double* z, double c0, const double* z2,
                       double c2, double c1, double* y) {
    const unsigned grid = (unsigned)((m.n_rows * 8 + TPB - 1) / TPB);
    hipLaunchKernelGGL((csr_kernel<true, 8>), dim3(grid), dim3(TPB), 0, st, m.n_rows, m.indptr, m.indices, m.vals, x, z, c0, c1, y, z2, c2);
}

// restriction b_coarse = R r and prolongation x = x + P x_coarse of one transfer: node blocks when the level has them
inline void restrict_apply(hipStream_t st, const fep_solver::Level& l, const double* r, double* bc) {
    const fep_solver::Level::Blocks& B = l.tb;
    if (!B.bf) { csr_apply(st, l.R, r, nullptr, 0.0, 1.0, bc); return; }
    const dim3 g((unsigned)((B.nc * 8 + TPB - 1) / TPB)), tb(TPB);
    if (B.bf == 2) hipLaunchKernelGGL(restrict_block_kernel<2>, g, tb, 0, st, B.nc, B.rptr, B.rcol, B.rval, r, bc);
    else hipLaunchKernelGGL(restrict_block_kernel<3>, g, tb, 0, st, B.nc, B.rptr, B.rcol, B.rval, r, bc);
}
inline void prolong_apply(hipStream_t st, const fep_solver::Level& l, const double* xc, double* x) {
    const fep_solver::Level::Blocks& B = l.tb;
    if (!B.bf) { csr_apply(st, l.P, xc, x, 1.0, 1.0, x); return; }
    const dim3 g((unsigned)((B.nf + TPB - 1) / TPB)), tb(TPB);
    if (B.bf == 2) hipLaunchKernelGGL(prolong_block_kernel<2>, g, tb, 0, st, B.nf, B.pptr, B.pcol, B.pval, xc, x, x);
    else hipLaunchKernelGGL(prolong_block_kernel<3>, g, tb, 0, st, B.nf, B.pptr, B.pcol, B.pval, xc, x, x);
}

// one level-0 pass of the V-cycle (block_residual_kernel) on the solve's K — its single-precision copy when there is one
template <int SMOOTH>
inline void block_pass(fep_solver* s, hipStream_t st, const double* K, const double* x, const double* b, double omega, double* out,
                       double ca, double cprev, const double* xprev) {
    const dim3 gm(s->n_mv_blocks), tb(TPB);
    if (s->fp32 && s->k32)
        hipLaunchKernelGGL((block_residual_kernel<SMOOTH, float2>), gm, tb, 0, st, s->n_n, s->nptr, s->ncol, s->free_dof,
                           (const float2*)s->k32, (const double2*)x, b, s->minv, omega, out, ca, cprev, (const double2*)xprev);
    else
        hipLaunchKernelGGL((block_residual_kernel<SMOOTH, double2>), gm, tb, 0, st, s->n_n, s->nptr, s->ncol, s->free_dof,
                           (const double2*)K, (const double2*)x, b, s->minv, omega, out, ca, cprev, (const double2*)xprev);
}

// Degree-2 Chebyshev smoother for D^-1 A on [lmax / alpha, lmax], lmax = safety * (largest eigenvalue the hierarchy was
// built with: omega = 4 / (3 * 1.05 * rho)).  Two steps, x1 = x0 + c1 D^-1 r0,  x2 = a2 x1 + cp x0 + w2 D^-1 r1:
// the same passes over the operator as two damped-Jacobi sweeps, a better polynomial.
struct Cheb { double c1, a2, cp, w2; };
inline Cheb cheb_coefficients(double omega, double alpha, double safety) {
    const double rho = 4.0 / (3.0 * 1.05 * omega);
    const double lmax = safety * rho, lmin = lmax / alpha;
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
    const double r0 = 1.0 / sigma, r1 = 1.0 / (2.0 * sigma - r0);
    return Cheb{1.0 / theta, 1.0 + r1 * r0, -r1 * r0, 2.0 * r1 / delta};
}

// z = V-cycle(b) on K; returns the buffer that holds z (one of s->u, s->w).  Two damped block-Jacobi sweeps before
// and after the coarse correction on every level, hence a symmetric positive definite operator.
double* vcycle_chebyshev(fep_solver* s, hipStream_t st, const double* K, const double* b0);

double* vcycle(fep_solver* s, hipStream_t st, const double* K, const double* b0) {
    if (s->cheb) return vcycle_chebyshev(s, st, K, b0);
    const dim3 gv(s->n_vec_blocks), tb(TPB);
    std::vector<fep_solver::Level>& L = s->levels;
    const int nl = (int)L.size();
    double *xa = s->u, *xb = s->w;
    const double w0 = L[0].omega;
    // level 0, pre-smoothing from x = 0
    hipLaunchKernelGGL(block_scale_kernel, gv, tb, 0, st, s->n_n, s->free_dof, s->minv, (const double2*)b0, w0, (double2*)xa);
    block_pass<1>(s, st, K, xa, b0, w0, xb, 1.0, 0.0, nullptr);
    block_pass<0>(s, st, K, xb, b0, 0.0, s->t0, 1.0, 0.0, nullptr);
    restrict_apply(st, L[0], s->t0, L[0].b);
    // coarse levels down: level k+1 lives in L[k].{A, D, x, b, r}; its smoother weight is L[k+1].omega
    for (int k = 0; k + 1 < nl; ++k) {
        fep_solver::Level& c = L[k];
        const double w = L[k + 1].omega;
        csr_apply(st, c.D, c.b, nullptr, 0.0, w, c.x);               // x = w D b
        csr_apply(st, c.A, c.x, c.b, 1.0, -1.0, c.r);                // r = b - A x
        csr_apply(st, c.D, c.r, c.x, 1.0, w, c.x);                   // x += w D r
        csr_apply(st, c.A, c.x, c.b, 1.0, -1.0, c.r);
        restrict_apply(st, L[k + 1], c.r, L[k + 1].b);
    }
    // coarsest: x = A^-1 b (A holds the inverse)
    csr_apply(st, L[nl - 1].A, L[nl - 1].b, nullptr, 0.0, 1.0, L[nl - 1].x);
    // up
    for (int k = nl - 2; k >= 0; --k) {
        fep_solver::Level& c = L[k];
        const double w = L[k + 1].omega;
        prolong_apply(st, L[k + 1], L[k + 1].x, c.x);                // x += P x_coarse
        for (int sw = 0; sw < 2; ++sw) {
            csr_apply(st, c.A, c.x, c.b, 1.0, -1.0, c.r);
            csr_apply(st, c.D, c.r, c.x, 1.0, w, c.x);
        }
    }
    prolong_apply(st, L[0], L[0].x, xb);
    block_pass<1>(s, st, K, xb, b0, w0, xa, 1.0, 0.0, nullptr);
    block_pass<1>(s, st, K, xa, b0, w0, xb, 1.0, 0.0, nullptr);
    return xb;
}

// The same V-cycle with the degree-2 Chebyshev smoother on every level: as many passes over every operator, iteration
// counts 20 % lower on plastic tangents.  Pre- and post-smoother are the same polynomial in D^-1 A, so the cycle stays a
// symmetric positive definite operator as long as lmax bounds the spectrum (safety factor 1.2 on the estimate the
// hierarchy was built with; a tangent's own largest eigenvalue measured 3.5 % above its elastic matrix's).
double* vcycle_chebyshev(fep_solver* s, hipStream_t st, const double* K, const double* b0) {
    const dim3 gv(s->n_vec_blocks), tb(TPB);
    std::vector<fep_solver::Level>& L = s->levels;
    const int nl = (int)L.size();
    double *xa = s->u, *xb = s->w;
    const Cheb c0 = cheb_coefficients(L[0].omega, s->cheb_alpha, s->cheb_safety);
    // level 0, pre-smoothing from x = 0: x1 = c1 D^-1 b, x2 = a2 x1 + w2 D^-1 (b - K x1)
    hipLaunchKernelGGL(block_scale_kernel, gv, tb, 0, st, s->n_n, s->free_dof, s->minv, (const double2*)b0, c0.c1, (double2*)xa);
    block_pass<2>(s, st, K, xa, b0, c0.w2, xb, c0.a2, 0.0, nullptr);
    block_pass<0>(s, st, K, xb, b0, 0.0, s->t0, 1.0, 0.0, nullptr);
    restrict_apply(st, L[0], s->t0, L[0].b);
    // with the refresh the coarse operators sit on whole 3x3 node blocks: node3_kernel does an operator pass and the
    // block-Jacobi step behind it in one launch (7 launches per level and cycle instead of 10)
    const bool n3 = s->refresh;
    auto node3 = [&](int mode, const fep_solver::Level& c, const double* x, double om, double ca, double cp, const double* xp,
                     double* out) {
        const int64_t nn = c.n_coarse / 3;
        // four nodes per lane group only where that still leaves a few thousand workgroups
        const bool big = nn >= (int64_t)NODES_PER_BLOCK * 2048;
        const int npb = big ? NODES_PER_BLOCK : NODES_PER_BLOCK / SPMV_PASSES;
        const dim3 g((unsigned)((nn + npb - 1) / npb));
#define FEP_NODE3(MODE_, AV_, A_)                                                                                                     \
        do {                                                                                                                          \
            if (big) hipLaunchKernelGGL((node3_kernel<MODE_, AV_, SPMV_PASSES>), g, tb, 0, st, nn, c.nbp, c.nbc, A_, c.D.vals, x, c.b, om, ca, cp, xp, out); \
            else hipLaunchKernelGGL((node3_kernel<MODE_, AV_, 1>), g, tb, 0, st, nn, c.nbp, c.nbc, A_, c.D.vals, x, c.b, om, ca, cp, xp, out);               \
        } while (0)
        if (c.A32) {
            const float* A = c.A32;
            if (mode == 0) FEP_NODE3(0, float, A); else if (mode == 1) FEP_NODE3(1, float, A); else FEP_NODE3(2, float, A);
            return;
        }
        const double* A = c.A.vals;
        if (mode == 0) FEP_NODE3(0, double, A); else if (mode == 1) FEP_NODE3(1, double, A); else FEP_NODE3(2, double, A);
#undef FEP_NODE3
    };
    // the last smoothed level and the coarsest solve under it in one launch when they are small enough (tail_kernel)
    int kt = -1;
    if (s->tail && n3 && nl >= 2) {
        const fep_solver::Level &c = L[nl - 2], &lst = L[nl - 1];
        if (c.A32 && c.n_coarse / 3 <= kTailNodes && lst.n_coarse <= kTailCoarse && lst.n_coarse % 3 == 0 && lst.tb.bf == 3 &&
            lst.tb.nf == c.n_coarse / 3 && lst.A.nnz == lst.n_coarse * lst.n_coarse)
            kt = nl - 2;
    }
    for (int k = 0; k + 1 < nl; ++k) {
        fep_solver::Level& c = L[k];
        const Cheb ch = cheb_coefficients(L[k + 1].omega, s->cheb_alpha, s->cheb_safety);
        if (k == kt) {
            const fep_solver::Level& lst = L[nl - 1];
            TailArgs ta;
            ta.n_nodes = (int)(c.n_coarse / 3); ta.n_c = (int)lst.n_coarse;
            ta.nbp = c.nbp; ta.nbc = c.nbc; ta.A = c.A32; ta.D = c.D.vals;
            ta.rptr = lst.tb.rptr; ta.rcol = lst.tb.rcol; ta.rval = lst.tb.rval;
            ta.pptr = lst.tb.pptr; ta.pcol = lst.tb.pcol; ta.pval = lst.tb.pval;
            ta.Ainv = lst.A.vals;
            ta.c1 = ch.c1; ta.a2 = ch.a2; ta.cp = ch.cp; ta.w2 = ch.w2;
            ta.b = c.b; ta.out = c.r;
            hipLaunchKernelGGL(tail_kernel, dim3(1), dim3(1024), 0, st, ta);
            c.xcur = c.r;
            break;
        }
        csr_apply(st, c.D, c.b, nullptr, 0.0, ch.c1, c.x);           // x1 = c1 D b
        if (n3) {
            node3(2, c, c.x, ch.w2, ch.a2, 0.0, nullptr, c.t);       // x2 = a2 x1 + w2 D (b - A x1)
            node3(0, c, c.t, 0.0, 0.0, 0.0, nullptr, c.r);           // r = b - A x2
            c.xcur = c.t;
        } else {
            csr_apply(st, c.A, c.x, c.b, 1.0, -1.0, c.r);            // r1 = b - A x1
            csr_apply(st, c.D, c.r, c.x, ch.a2, ch.w2, c.x);         // x2 = a2 x1 + w2 D r1
            csr_apply(st, c.A, c.x, c.b, 1.0, -1.0, c.r);
            c.xcur = c.x;
        }
        restrict_apply(st, L[k + 1], c.r, L[k + 1].b);
    }
    if (kt < 0) {
        csr_apply(st, L[nl - 1].A, L[nl - 1].b, nullptr, 0.0, 1.0, L[nl - 1].x);
        L[nl - 1].xcur = L[nl - 1].x;
    }
    for (int k = kt >= 0 ? kt - 1 : nl - 2; k >= 0; --k) {
        fep_solver::Level& c = L[k];
        const Cheb ch = cheb_coefficients(L[k + 1].omega, s->cheb_alpha, s->cheb_safety);
        prolong_apply(st, L[k + 1], L[k + 1].xcur, c.xcur);         // x0 = x + P x_coarse
        if (n3) {
            double* x1 = c.xcur == c.t ? c.x : c.t;
            node3(1, c, c.xcur, ch.c1, 1.0, 0.0, nullptr, x1);       // x1 = x0 + c1 D (b - A x0)
            node3(2, c, x1, ch.w2, ch.a2, ch.cp, c.xcur, c.r);       // x2 = a2 x1 + cp x0 + w2 D (b - A x1)
            c.xcur = c.r;
        } else {
            csr_apply(st, c.A, c.x, c.b, 1.0, -1.0, c.r);            // r0
            csr_apply(st, c.D, c.r, c.x, 1.0, ch.c1, c.t);           // x1 = x0 + c1 D r0        (x0 stays in c.x)
            csr_apply(st, c.A, c.t, c.b, 1.0, -1.0, c.r);            // r1
            csr_apply2(st, c.D, c.r, c.t, ch.a2, c.x, ch.cp, ch.w2, c.x);   // x2 = a2 x1 + cp x0 + w2 D r1
        }
    }
    prolong_apply(st, L[0], L[0].xcur, xb);                           // x0
    block_pass<1>(s, st, K, xb, b0, c0.c1, xa, 1.0, 0.0, nullptr);   // x1 = x0 + c1 D^-1 r0
    block_pass<2>(s, st, K, xa, b0, c0.w2, s->t0, c0.a2, c0.cp, xb);  // x2 -> t0
    return s->t0;
}

}  // namespace

extern "C" int fep_solver_amg_pcg_dev(fep_solver* s, void* stream, const double* k_data_d, const double* b_d, double* x_d,
                                      double rtol, int max_iter, int check_every, int* iters_out, double* relres_out,
                                      int* state_out) {
    if (!s || !k_data_d || !b_d || !x_d || !(rtol >= 0.0) || max_iter < 0) return FEP_EINVAL;
    if (!fep_aligned16(k_data_d) || !fep_aligned16(b_d) || !fep_aligned16(x_d)) return FEP_EINVAL;
    if (s->levels.empty() || !s->levels.back().last) return FEP_ESTATE;
    if (check_every <= 0) check_every = 10;
    FEP_TRY(fep_set_device(s->device));
    hipStream_t st = (hipStream_t)stream;
    const dim3 gv(s->n_vec_blocks), gm(s->n_mv_blocks), tb(TPB), one(1), big(1024);
    const double tol2 = rtol * rtol;
    double2 *x = (double2*)x_d, *r = (double2*)s->r, *p = (double2*)s->p;
    if (s->refresh) FEP_TRY(fep_solver_amg_refresh_dev(s, stream, k_data_d));
    if (s->fp32 && s->k32)
        hipLaunchKernelGGL(to_float_kernel, dim3((unsigned)((s->n_blk + TPB - 1) / TPB)), tb, 0, st, s->n_blk, (const double4*)k_data_d,
                           (float4*)s->k32);
    hipLaunchKernelGGL(block_jacobi_kernel, gv, tb, 0, st, s->n_n, s->nptr, s->ncol, s->free_dof, k_data_d, s->minv);
    hipLaunchKernelGGL(mg_init_kernel, gv, tb, 0, st, s->n_n, (const double2*)b_d, s->free_dof, x, r, s->part_r);
    double* z = vcycle(s, st, k_data_d, s->r);
    hipLaunchKernelGGL(mg_dot_kernel<true>, gv, tb, 0, st, s->n_n, (const double2*)z, (const double2*)r, p, s->part_g);
    hipLaunchKernelGGL(mg_scalar_kernel, one, big, 0, st, s->scal, 0, s->part_g, s->n_vec_blocks, s->part_r,
                       s->n_vec_blocks, tol2);
    HIP_TRY(hipGetLastError());
    Scal h;
    std::memset(&h, 0, sizeof h);
    int launched = 0, n_prev = 0;
    double rr_prev = 0.0;
    for (;;) {
        HIP_TRY(hipMemcpyAsync(&h, s->scal, sizeof h, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h.state != 0 || launched >= max_iter) break;
        const int n = std::min(next_batch(check_every, rr_prev, n_prev, h.rr, tol2 * h.bb), max_iter - launched);
        rr_prev = h.rr; n_prev = n;
        for (int i = 0; i < n; ++i) {
            hipLaunchKernelGGL(spmv_kernel<true>, gm, tb, 0, st, s->n_n, s->nptr, s->ncol, s->free_dof,
                               (const double2*)k_data_d, (const double2*)p, s->q, (const double*)s->p, s->part_d);
            hipLaunchKernelGGL(mg_scalar_kernel, one, big, 0, st, s->scal, 1, s->part_d, s->n_mv_blocks,
                               (const double*)nullptr, 0, tol2);
            hipLaunchKernelGGL(mg_update_kernel, gv, tb, 0, st, s->n_n, s->scal, (const double2*)p, (const double2*)s->q,
                               x, r, s->part_r);
            z = vcycle(s, st, k_data_d, s->r);
            hipLaunchKernelGGL(mg_dot_kernel<false>, gv, tb, 0, st, s->n_n, (const double2*)z, (const double2*)r,
                               (double2*)nullptr, s->part_g);
            hipLaunchKernelGGL(mg_scalar_kernel, one, big, 0, st, s->scal, 2, s->part_g, s->n_vec_blocks, s->part_r,
                               s->n_vec_blocks, tol2);
            hipLaunchKernelGGL(mg_direction_kernel, gv, tb, 0, st, s->n_n, s->scal, (const double2*)z, p);
        }
        HIP_TRY(hipGetLastError());
        launched += n;
    }
    if (iters_out) *iters_out = h.it;
    if (relres_out) *relres_out = h.bb > 0.0 ? std::sqrt(h.rr / h.bb) : 0.0;
    if (state_out) *state_out = h.state;
    return FEP_OK;
}

// Greedy aggregation of a node graph (CSR, self loops allowed): pass 1 makes an aggregate of every node whose
// neighbours are all free, pass 2 attaches the rest to a neighbouring aggregate (or makes singletons).
extern "C" int fep_aggregate_host(int64_t n, const int32_t* indptr, const int32_t* indices, int32_t* agg_out,
                                  int64_t* n_agg_out) {
    try {
        return fep_host::aggregate(n, indptr, indices, agg_out, n_agg_out);
    } catch (const std::bad_alloc&) {
        return FEP_ENOMEM;
    } catch (...) {
        return FEP_EINVAL;
    }
}

// Host sparse product for the multigrid set-up (fep_host.h spgemm_count / spgemm_fill): no GPU involved.
extern "C" int fep_spgemm_count_host(int64_t n_rows, int64_t n_mid, int64_t n_cols, const int32_t* x_indptr, const int32_t* x_indices,
                                     const int32_t* y_indptr, const int32_t* y_indices, int32_t* c_indptr_out) {
    try {
        return fep_host::spgemm_count(n_rows, n_mid, n_cols, x_indptr, x_indices, y_indptr, y_indices, c_indptr_out);
    } catch (const std::bad_alloc&) {
        return FEP_ENOMEM;
    } catch (...) {
        return FEP_EINVAL;
    }
}

extern "C" int fep_spgemm_fill_host(int64_t n_rows, int64_t n_mid, int64_t n_cols, const int32_t* x_indptr, const int32_t* x_indices,
                                    const double* x_vals, const int32_t* y_indptr, const int32_t* y_indices, const double* y_vals,
                                    const int32_t* c_indptr, int32_t* c_indices_out, double* c_vals_out) {
    try {
        return fep_host::spgemm_fill(n_rows, n_mid, n_cols, x_indptr, x_indices, x_vals, y_indptr, y_indices, y_vals, c_indptr,
                                     c_indices_out, c_vals_out);
    } catch (const std::bad_alloc&) {
        return FEP_ENOMEM;
    } catch (...) {
        return FEP_EINVAL;
    }
}
