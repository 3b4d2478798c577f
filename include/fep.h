/*
 * fep.h — C ABI of libfep_hip.so: the MI355X (gfx950) implementation of the
 * reference's per-integration-point Drucker–Prager return map and per-element
 * tangent-stiffness / internal-force assembly.
 *
 * The reference (MartinBeseda/FEM-ElastoPlasticity) is pure Python and has no FFI;
 * its "operator interface" for this path is a set of module-level functions.
 * Each entry point below names the reference lines it replaces
 * (DP = Plasticity2D_DP/pythonFEM.py, TSX = tsx-tunnel/pythonFEM.py,
 *  EL = Elasticity2D/pythonFEM.py).  INTEGRATION.md shows the ctypes stubs.
 *
 * Conventions
 *   - every function returns 0 (FEP_OK) or a negative FEP_E* code; nothing throws or aborts;
 *   - `double` is IEEE fp64; node/element indices are int32_t, 0-based; sizes are int64_t;
 *   - integration point id  k = e*n_q + q  (q fastest)                       DP:510-511,526-527
 *   - DOF id = 2*node + comp (U, F are the column-major flattening of (2,n_n)) DP:560-565, DP:1043
 *   - strain/stress 3-vectors [11,22,12(engineering)], 4-vectors [11,22,12,33] DP:651
 *   - per-point arrays are "rows x n_int", C-contiguous (component-major, SoA),
 *     exactly the reference's (4,n_int)/(9,n_int) NumPy arrays;
 *   - `ds` holds the 3x3 consistent tangent row-major, m = 3*i + j            DP:703
 *   - pointers named *_h are host memory, *_d are device (HBM) memory of the
 *     context's GPU; the caller owns every pointer for the duration of the call;
 *   - device vectors indexed by DOF or by CSR position (U, F, k_data, the solver's x, b, y) must be 16-byte
 *     aligned (hipMalloc and every tensor library deliver that; checked, FEP_EINVAL otherwise): the kernels move
 *     them as (x, y) pairs;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); *_dev
 *     calls only enqueue work and return, *_host calls are synchronous (they run on a stream of the library's own,
 *     with persistent device buffers and pinned staging: no allocation per call after the first);
 *   - calls on one context are not re-entrant; one context per (host thread, GPU).
 */
#ifndef FEP_H
#define FEP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FEP_OK            0
#define FEP_EINVAL       -1   /* bad argument (NULL pointer, negative size, unknown element type) */
#define FEP_ENODEV       -2   /* no usable HIP device / hipSetDevice failed */
#define FEP_ENOMEM       -3   /* host or device allocation failed */
#define FEP_EHIP         -4   /* a HIP runtime call or kernel launch failed (see fep_last_hip_error) */
#define FEP_ERANGE       -5   /* an index is out of range (element -> node id, 32-bit offset overflow) */
#define FEP_ESTATE       -6   /* call order violated (e.g. materials not set) */

/* LagrangeElementType values of the reference (DP:55-60, TSX:57-63) */
#define FEP_P1 1
#define FEP_P2 2
#define FEP_Q1 3
#define FEP_Q2 4
#define FEP_P4 5

typedef struct fep_ctx fep_ctx;

/* ---- library ------------------------------------------------------------------------ */
int         fep_version(void);                 /* ABI version, currently 1 */
int         fep_build_is_ablation(void);       /* 1: built with -DFEP_ABLATION (measurement switches by environment variable, phase
                                                * clocks; tools/ only), 0: the product library, which reads FEP_ROUTE,
                                                * FEP_VALIDATE_PLAN, FEP_VERBOSE, FEP_HOST_THREADS, FEP_COPY_THREADS and nothing else */
const char* fep_strerror(int code);
int         fep_last_hip_error(void);          /* raw hipError_t of the last FEP_EHIP on this thread */
int         fep_device_count(int* n_out);
/* (n_p, n_q) of an element type; FEP_EINVAL if unknown.  Tables: DP:364-488, TSX:67-274. */
int         fep_element_shape(int elem_type, int* n_p, int* n_q);

/* ---- device memory helpers (for callers that do not bring their own allocator) ------- */
int fep_malloc(int device_id, void** ptr_d, int64_t bytes);
int fep_free(int device_id, void* ptr_d);
int fep_memcpy_h2d(int device_id, void* dst_d, const void* src_h, int64_t bytes);
int fep_memcpy_d2h(int device_id, void* dst_h, const void* src_d, int64_t bytes);
int fep_sync(int device_id, void* stream);
/* Page-locked host memory from a size-keyed cache (a returned block is reused by the next request of the same size;
 * up to 4 GiB stay cached).  The *_host entry points DMA straight from / into such blocks; any other host pointer is
 * staged through a ring of pinned slots filled by copy threads (FEP_COPY_THREADS, default 8).  The Python layer
 * places every output array it hands to the caller (s, ds, ind_p, K data, F: DP:1044-1058) in such blocks. */
int fep_host_alloc(void** ptr_h, int64_t bytes);
int fep_host_free(void* ptr_h);     /* FEP_EINVAL for a pointer that is not a live block of the cache (e.g. freed twice) */
/* Unpins and frees every cached (idle) block.  fep_host_alloc does this by itself, once, before it reports FEP_ENOMEM. */
int fep_host_trim(void);

/* ---- a2: return map, mesh-free (pointwise) --------------------------------------------
 * Replaces construct_constitutive_problem, DP:604-757 (e0_h == NULL) and TSX:990-1157
 * (e0_h = the broadcast (4,1) initial strain).
 *
 *   e          strain, component i of point k at e[k*e_pt_stride + i*e_comp_stride]
 *              ((3,n_int) C-order: (1, n_int); the driver's F-ordered array DP:1043: (3, 1)); never modified
 *   ep_prev    (4,n_int) or NULL (= zeros, DP:667).  If `accept` != 0 it is UPDATED IN PLACE
 *              (the reference returns the mutated `ep_prev` as 'ep', DP:751-755)
 *   shear,bulk,eta,c   (n_int) each
 *   s          out (4,n_int)   ds  out (9,n_int)   ind_p  out (n_int) 0/1 bytes
 *   counts     out [2] = {n_smooth, n_apex} (the numbers the reference logs at DP:730); may be NULL
 */
int fep_return_map_host(int device_id, int64_t n_int,
                        const double* e_h, int64_t e_pt_stride, int64_t e_comp_stride,
                        const double* e0_h, double* ep_prev_h,
                        const double* shear_h, const double* bulk_h, const double* eta_h, const double* c_h,
                        int accept,
                        double* s_h, double* ds_h, uint8_t* ind_p_h, int64_t* counts_h);

/* Same on device-resident arrays.  `counts_d` (2 x int64, device) is zeroed and filled by the call.  With counts_d the
 * call uses a per-(device, stream) scratch of 8 bytes per 256 points that grows on demand (FEP_ESTATE if it would have to
 * grow while the stream is being captured: run the call once outside the capture first); a block handed out is never
 * freed, so a HIP graph captured from this call stays valid for calls of up to the captured n_int. */
int fep_return_map_dev(int device_id, void* stream, int64_t n_int,
                       const double* e_d, int64_t e_pt_stride, int64_t e_comp_stride,
                       const double* e0_h, double* ep_prev_d,
                       const double* shear_d, const double* bulk_d, const double* eta_d, const double* c_d,
                       int accept,
                       double* s_d, double* ds_d, uint8_t* ind_p_d, int64_t* counts_d);

/* ---- a6/a7: mesh context (static operands of the hot path) ----------------------------
 * Replaces the geometry / index part of get_elastic_stiffness_matrix
 * (DP:491-601, TSX:432-542, EL:368-477): Jacobians, dphi_1/dphi_2, weight = |det|*wf,
 * and the symbolic pattern of K = B^T D B.
 *
 *   elements   (n_p, n_e) C-order, 0-based node ids (the reference's `elements`; EL passes 1-based
 *              and shifts in place, EL:389 — shift before calling)
 *   coords     (2, n_n) C-order
 *   dhatp1/2   (n_p, n_q) C-order reference-element derivative tables, wf (n_q) weight factors
 */
int fep_ctx_create(fep_ctx** ctx_out, int device_id, int elem_type,
                   int64_t n_e, int64_t n_n,
                   const int32_t* elements_h, const double* coords_h,
                   const double* dhatp1_h, const double* dhatp2_h, const double* wf_h);
int fep_ctx_destroy(fep_ctx* ctx);

/* sizes[0..7] = n_e, n_n, n_p, n_q, n_int, n_dof (=2*n_n), nnz (CSR entries of K), n_blk (node-pair blocks) */
int fep_ctx_sizes(const fep_ctx* ctx, int64_t sizes[8]);
/* The kernels one step of this context launches, named as rocprofv3 prints them and joined by " + " (which = 0: a step with
 * every output, 1: the K,F-only step of a Newton iterate).  `buf` receives a NUL-terminated string of at most cap - 1
 * characters.  No reference counterpart (measurement support: bench.py's roofline label). */
int fep_ctx_kernel_names(const fep_ctx* ctx, int which, char* buf, int64_t cap);

/* dphi1, dphi2: (n_p, n_int); weight: (n_int); det: (n_int) or NULL.       DP:530-546, 585 */
int fep_ctx_geometry_host(fep_ctx* ctx, double* dphi1_h, double* dphi2_h, double* weight_h, double* det_h);

/* Symbolic CSR pattern of K (rows = DOFs, sorted columns, every structural entry of B^T D B kept,
 * also those that are numerically zero — the reference's SciPy product drops them). */
int fep_ctx_pattern_host(const fep_ctx* ctx, int32_t* indptr_h /* n_dof+1 */, int32_t* indices_h /* nnz */);

/* Per-point material parameters (n_int each): shear, bulk (DP:972-973), eta, c (DP:983-984).  When each of the four
 * arrays is constant over the mesh (the reference's demos) the kernels take the constants as arguments and do not
 * read the arrays; results are bitwise the same either way. */
int fep_ctx_set_materials_host(fep_ctx* ctx, const double* shear_h, const double* bulk_h,
                               const double* eta_h, const double* c_h);
/* Device pointers of the context's static per-point arrays (for the mesh-free entry points):
 * which = 0 shear, 1 bulk, 2 eta, 3 c, 4 weight, 5 dphi1, 6 dphi2. */
int fep_ctx_device_ptr(const fep_ctx* ctx, int which, void** ptr_d);

/* ---- a1..a5 fused: one pass of the hot path ------------------------------------------
 * Replaces, for one Newton iterate (DP:1043-1058 / TSX:1771-1778):
 *     E = B*U                                   (a1)
 *     construct_constitutive_problem(E, ...)     (a2)
 *     vD = w*ds ; D_p ; K_tangent = K_elast + B^T (D_p - D_elast) B   (a3, a4)
 *     F = B^T (w * s[0:3])                       (a5)
 * K_tangent is produced as the `data` array of the context's CSR pattern, computed as
 * B^T D_p B directly (equal to the reference's expression up to fp64 rounding).
 *
 *   U          (n_dof) displacement, DOF order
 *   e0_h       4 host doubles (TSX initial strain zeta*e_init, TSX:1765) or NULL
 *   ep_prev    (4,n_int) or NULL; updated in place when accept != 0
 *   e_out      (3,n_int) C-order strain or NULL (not needed by the path itself)
 *   s, ds, ind_p   as in fep_return_map_*; any of them may be NULL when not wanted
 *   k_data     out (nnz)     f_out  out (n_dof)
 *   counts     out [2] {n_smooth, n_apex} or NULL
 *
 * fep_step_dev neither allocates nor synchronises, so the call can be captured into a hipGraph.  (One exception: a P1
 * context asked for k_data / f_out on an ACCEPTING call without ds / s allocates its ds / s scratch on the first such
 * call; make that call once outside a capture.  P1 non-accepting calls without point outputs run as one kernel and need
 * no scratch; the other routes get theirs in fep_ctx_set_materials_host.)  Results are bitwise reproducible run to run
 * (fixed summation order, no floating-point atomics).
 */
int fep_step_dev(fep_ctx* ctx, void* stream, const double* u_d, const double* e0_h,
                 double* ep_prev_d, int accept,
                 double* e_out_d, double* s_d, double* ds_d, uint8_t* ind_p_d,
                 double* k_data_d, double* f_out_d, int64_t* counts_d);
int fep_step_host(fep_ctx* ctx, const double* u_h, const double* e0_h,
                  double* ep_prev_h, int accept,
                  double* e_out_h, double* s_h, double* ds_h, uint8_t* ind_p_h,
                  double* k_data_h, double* f_out_h, int64_t* counts_h);

/* The same with the displacement as the reference holds it: `u2_h` = the (2, n_n) C-ordered array `U` (DP:1043
 * flattens it column-major on every iterate); the reordering into DOF order happens inside the staging copy. */
int fep_step_host_planar(fep_ctx* ctx, const double* u2_h, const double* e0_h,
                         double* ep_prev_h, int accept,
                         double* e_out_h, double* s_h, double* ds_h, uint8_t* ind_p_h,
                         double* k_data_h, double* f_out_h, int64_t* counts_h);

/* ---- a3..a5 only: assembly from given ds / s ------------------------------------------
 * Replaces DP:1047-1050 + DP:1058 when the caller already holds `ds` (9,n_int) and `s` (>=3 rows
 * used, (4,n_int) layout).  Either output may be NULL.  With ds = the elastic tensor this is the
 * K_elast = B^T D B of DP:595. */
int fep_assemble_dev(fep_ctx* ctx, void* stream, const double* ds_d, const double* s_d,
                     double* k_data_d, double* f_out_d);
int fep_assemble_host(fep_ctx* ctx, const double* ds_h, const double* s_h,
                      double* k_data_h, double* f_out_h);

/* ---- multi-GPU interface exchange helpers (no reference counterpart: the reference is single-process) ----
 * Pack / unpack of the interface DOFs around the RCCL all-reduce of the internal force (sharding.py):
 *   fep_gather_f64   dst[i] = idx[i] >= 0 ? src[idx[i]] : 0      (i < n)
 *   fep_scatter_f64  dst[dst_idx[i]] = src[src_idx[i]]            (i < n; dst_idx must be unique) */
int fep_gather_f64(int device_id, void* stream, int64_t n, const double* src_d, const int32_t* idx_d, double* dst_d);
int fep_scatter_f64(int device_id, void* stream, int64_t n, const double* src_d, const int32_t* src_idx_d,
                    const int32_t* dst_idx_d, double* dst_d);
/* Neighbour-only form of the same exchange (sharding.py, exchange='p2p'): after the send / receive batch, interface DOF i
 * (local DOF loc[i]) becomes 0 + c_0 + c_1 + ... over its holders in ascending rank order; contribution k of DOF i is
 * src[ptr[i] + k] < 0 ? this rank's own f[loc[i]] : recv[src[ptr[i] + k]].  Replaces nothing in the reference (DP:1058
 * is the quantity exchanged). */
int fep_iface_sum_f64(int device_id, void* stream, int64_t n, const int32_t* loc_d, const int32_t* ptr_d,
                      const int32_t* src_d, const double* recv_d, double* f_d);

/* ---- callers of the hot path ----------------------------------------------------------------
 * transform (DP:760-816): integration-point values (n_int) -> nodal values (n_n), mean over the points of the
 * adjacent elements weighted with quadrature weight * |det J| (the footing pressure that steers the load step,
 * DP:1105).  A node that belongs to no element gets 0/0 = NaN, as the reference's F1 / F2 (DP:812) gives. */
int fep_transform_dev(fep_ctx* ctx, void* stream, const double* q_int_d, double* q_node_d);
int fep_transform_host(fep_ctx* ctx, const double* q_int_h, double* q_node_h);

/* Linear solve of a Newton iterate, K[Q][:,Q] dU[Q] = b[Q]  (np.linalg.solve on the dense boolean-masked block at
 * DP:1062-1066 / TSX:1781).  Preconditioned conjugate gradients (2x2 node-block Jacobi) entirely on
 * the device; K is the `data` array fep_step_dev wrote, on the context's CSR pattern.
 *
 *   fep_solver_create   pattern = fep_ctx_pattern_host (checked: rows 2n, 2n+1 share their columns, which come in
 *                       pairs 2m, 2m+1 -> FEP_EINVAL otherwise); free_dof_h (2 n_n) is Q in DOF order, non-zero = free
 *   fep_solver_sizes    {n_n, n_dof, nnz, n_free}
 *   fep_solver_spmv_dev y = K x (masked != 0: y = Q K x, x must then be 0 on the constrained DOFs); x != y
 *   fep_solver_pcg_dev  x = 0 on entry is implied; iterates until |r| <= rtol |b[Q]| (recursive residual), at most
 *                       max_iter iterations; the host looks at the device-side state after at most check_every
 *                       iterations (<= 0: 50; sooner when the residual history says the test is about to be met) and the
 *                       iterate is frozen on the device at the iteration that met the test.
 *                       *state_out: 0 = max_iter reached, 1 = converged, 2 = breakdown (K[Q][:,Q] not positive
 *                       definite or non-finite values); x is 0 on constrained DOFs.  Synchronises `stream`. */
typedef struct fep_solver fep_solver;
int fep_solver_create(fep_solver** solver_out, int device_id, int64_t n_n, const int32_t* indptr_h,
                      const int32_t* indices_h, const uint8_t* free_dof_h);
int fep_solver_destroy(fep_solver* solver);
int fep_solver_sizes(const fep_solver* solver, int64_t sizes[4]);
int fep_solver_spmv_dev(fep_solver* solver, void* stream, const double* k_data_d, const double* x_d, double* y_d,
                        int masked);
int fep_solver_pcg_dev(fep_solver* solver, void* stream, const double* k_data_d, const double* b_d, double* x_d,
                       double rtol, int max_iter, int check_every, int* iters_out, double* relres_out,
                       int* state_out);

/* Multigrid preconditioner for the same solve (smoothed aggregation).  The hierarchy is built on the host (solver.py:
 * aggregates from fep_aggregate_host, rigid-body-mode prolongators, Galerkin products with SciPy) from a reference
 * matrix on the context's pattern — normally K_elast — and pushed level by level; the solver applies it as a V(2,2)
 * cycle in which level 0 is always the CURRENT tangent (k_data_d of the call) and the coarse operators stay those of the
 * reference matrix.  Smoother: degree-2 Chebyshev in D^-1 A on [lmax/20, lmax], lmax = 1.2 x the value the hierarchy's
 * omega encodes (omega = 4 / (3 * 1.05 * rho)).
 * The preconditioner reads single precision — K in the smoother's level-0 passes, the refreshed coarse operators
 * and the transfers, applied in node blocks —;
 * CG's own product, its vectors and the Galerkin products are double precision.
 *
 *   fep_solver_amg_push_level   transfer level k -> k+1 (k = number of levels pushed so far; level 0 = the mesh DOFs):
 *       P (n_fine x n_coarse) and R = P^T (n_coarse x n_fine) in CSR; A = operator of level k+1 (n_coarse^2, CSR) or,
 *       when last != 0, its INVERSE (dense rows in CSR); D = inverse of the block diagonal of A (CSR; NULL when last);
 *       omega_fine = damping of the smoother on level k.  Rows of P belonging to constrained DOFs must be zero.
 *   fep_solver_amg_clear        drops the hierarchy
 *   fep_solver_amg_pcg_dev      as fep_solver_pcg_dev, preconditioned with the V-cycle (FEP_ESTATE without a complete
 *                               hierarchy); check_every <= 0: 10
 *   fep_solver_amg_enable_refresh   after the last level: from now on every fep_solver_amg_pcg_dev first re-projects the
 *                               coarse operators from ITS k_data_d — A_1 = R_0 K P_0, A_2 = R_1 A_1 P_1, ... with the
 *                               transfers as pushed, block-Jacobi inverses and the coarsest inverse recomputed — instead of
 *                               keeping those of the reference matrix (numeric products on patterns fixed here by the host;
 *                               the terms of every output entry are listed on the device:
 *                               40 % fewer iterations on plastic tangents).  FEP_ERANGE: coarsest level > 256 DOFs (its inverse
 *                               is recomputed by one workgroup; checked first, the hierarchy is left as pushed) or a product
 *                               whose pattern / term list exceeds 32-bit counts (found while the plans are built: the
 *                               hierarchy is DROPPED, as after an allocation failure).  A caller that wants to go on with the
 *                               reference operators pushes the levels again after FEP_ERANGE (solver.py: setup_amg does).
 *   fep_solver_amg_refresh_dev  the re-projection alone, stream-ordered (FEP_ESTATE unless enabled)
 *   fep_aggregate_host          greedy aggregation of a node graph in CSR (a neighbour list may repeat ids, in any order):
 *                               agg_out[i] in [0, *n_agg_out) */
int fep_solver_amg_clear(fep_solver* solver);
int fep_solver_amg_enable_refresh(fep_solver* solver);
int fep_solver_amg_refresh_dev(fep_solver* solver, void* stream, const double* k_data_d);
int fep_solver_amg_push_level(fep_solver* solver, int64_t n_fine, int64_t n_coarse,
                              const int32_t* p_indptr, const int32_t* p_indices, const double* p_vals,
                              const int32_t* r_indptr, const int32_t* r_indices, const double* r_vals,
                              const int32_t* a_indptr, const int32_t* a_indices, const double* a_vals,
                              const int32_t* d_indptr, const int32_t* d_indices, const double* d_vals,
                              double omega_fine, int last);
int fep_solver_amg_pcg_dev(fep_solver* solver, void* stream, const double* k_data_d, const double* b_d, double* x_d,
                           double rtol, int max_iter, int check_every, int* iters_out, double* relres_out,
                           int* state_out);
int fep_aggregate_host(int64_t n, const int32_t* indptr, const int32_t* indices, int32_t* agg_out, int64_t* n_agg_out);
/* Host sparse product C = X * Y for the set-up's Galerkin products (P^T A P with SciPy in the first versions; the reference has
 * no multigrid: DP:1062-1066 is a dense solve), rows in parallel.  _count: structural row sizes into c_indptr_out (n_rows + 1);
 * _fill: column ids ascending per row and values on that structure (entries that cancel to zero are kept).  No GPU involved. */
int fep_spgemm_count_host(int64_t n_rows, int64_t n_mid, int64_t n_cols, const int32_t* x_indptr, const int32_t* x_indices,
                          const int32_t* y_indptr, const int32_t* y_indices, int32_t* c_indptr_out);
int fep_spgemm_fill_host(int64_t n_rows, int64_t n_mid, int64_t n_cols, const int32_t* x_indptr, const int32_t* x_indices,
                         const double* x_vals, const int32_t* y_indptr, const int32_t* y_indices, const double* y_vals,
                         const int32_t* c_indptr, int32_t* c_indices_out, double* c_vals_out);

/* ---- in-situ kernel timing (bench.py's roofline figure) --------------------------------
 * Between fep_ctx_profile_begin and fep_ctx_profile_end every fep_step_dev / fep_assemble_dev call
 * brackets each of its kernels with HIP events on the launch stream (the kernels run in their real
 * position inside the step, not replayed back to back).  _end synchronises the stream and returns
 * the average milliseconds per launch.  P1 (node route): ms_out[0] p1_point_kernel (strain + return map),
 * ms_out[1] p1_node_lds_kernel (tangent CSR values + force), ms_out[2] 0; a P1 step that runs as ONE kernel (no point
 * output wanted, not accepting): ms_out[0] = the gap between two event marks (~0.005), ms_out[1] p1_fused_kernel
 * (+ the one-workgroup counter sum).  Other element types / COO route:
 * ms_out[0] element_kernel (strain + return map + K_e, f_e), ms_out[1] csr_reduce_kernel, ms_out[2]
 * force_reduce_kernel.  *n_steps = steps averaged. */
int fep_ctx_profile_begin(fep_ctx* ctx);
int fep_ctx_profile_end(fep_ctx* ctx, void* stream, double ms_out[3], int* n_steps);

#ifdef __cplusplus
}
#endif
#endif /* FEP_H */
