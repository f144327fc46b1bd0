/*
 * diffsol_hip.h — C ABI of libdiffsol_hip.so: the MI355X (gfx950) device backend for diffsol's
 * batched implicit ODE/DAE hot path.
 *
 * This is the drop-in boundary: every entry point is what a Rust `HipContext / HipVec / HipMat / HipLU`
 * backend crate (implementing diffsol's Context / Vector / Matrix / LinearSolver traits) would bind through
 * `extern "C"`; INTEGRATION.md shows the Rust-side stubs.  Each declaration cites the reference interface it
 * replaces (paths relative to the reference checkout, crates/...).
 *
 * Conventions
 *  - All data is f64 on the device (reference: f64-only CUDA kernels, diffsol-la/src/cuda_kernels/all.cu:2-28).
 *  - A "batched vector" holds `n` states for each of `nbatch` independent systems.  DEVICE LAYOUT IS
 *    BATCH-FASTEST (structure of arrays): element i of system b lives at  p[i*nbatch + b].  The reference API
 *    layout is batch-major ([b0 states..., b1 states...], diffsol-la/src/vector/cuda.rs:119-125); conversion
 *    happens only in dsh_vec_upload / dsh_vec_download, so `from_vec` / `clone_as_vec` keep their semantics.
 *    Matrices are column-major per system with the same batch-fastest rule: A_b(i,j) at p[(j*nrows+i)*nbatch + b],
 *    hence column j of a batched matrix is itself a contiguous batched vector (views are pointer offsets).
 *  - An operand may have nbatch 1 (broadcast, diffsol-la/src/context/mod.rs:24-26): it is then a plain
 *    contiguous array of n doubles.  Every operand is passed as (pointer, operand_nbatch) with
 *    operand_nbatch in {1, nbatch}.
 *  - Return value: 0 = DSH_OK, negative = error (message: dsh_last_error(), thread-local like
 *    diffsol-c/src/error_c.rs:12-121).  Shape / nbatch mismatches that panic in the reference return
 *    DSH_E_INVALID.  Reductions are blocking (they synchronise the context's stream), like the reference
 *    (diffsol-la/src/vector/cuda.rs:100-115).
 *  - One host thread per context; all work is issued in order on the context's HIP stream.
 */
#ifndef DIFFSOL_HIP_H
#define DIFFSOL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSH_OK 0
#define DSH_E_INVALID (-1)        /* bad shape / nbatch / argument (reference: panic) */
#define DSH_E_HIP (-2)            /* HIP runtime error */
#define DSH_E_SINGULAR (-3)       /* LU solve met a zero pivot (reference: LaError::LuSolveFailed) */
#define DSH_E_NOT_SETUP (-4)      /* LU used before factorisation (reference: LuNotInitialized) */
#define DSH_E_BATCH_MISMATCH (-5) /* root finding results differ across batches (reference panics, vector/cuda.rs:1166-1171) */
#define DSH_E_UNSUPPORTED (-6)
#define DSH_E_STALE (-7)          /* dsh_reduction_wait: the ticket's result records have been reused by newer launches */

typedef struct dsh_ctx dsh_ctx; /* device + stream + reduction scratch;   replaces CudaContext, diffsol-la/src/context/cuda.rs:41-144 */
typedef struct dsh_lu dsh_lu;   /* batched LU factors + pivots;            replaces CudaLU,      diffsol-la/src/linear_solver/cuda/lu.rs:15-57 */

const char* dsh_last_error(void);
int dsh_version(void);

/* ---- context (Context trait: diffsol-la/src/context/mod.rs:20-68; CudaContext::new context/cuda.rs:48-68) ---- */
/* stream == NULL: the context creates and owns a non-blocking stream; otherwise it borrows the caller's hipStream_t
 * (e.g. torch.cuda.current_stream().cuda_stream). */
int dsh_ctx_create(int device, void* stream, dsh_ctx** out);
void dsh_ctx_destroy(dsh_ctx* ctx);
int dsh_ctx_sync(dsh_ctx* ctx);
/* THREADING CONTRACT: a context and every object created from it (vectors, matrices, LU handles, solvers) may be MOVED between host threads, and clones of one
 * context may be USED from several threads: every entry point that takes a context (or an object made from one) holds the context's lock for the duration of the
 * call and re-binds the calling thread's current HIP device when the thread changed, so concurrent callers are serialised call by call onto the context's one
 * in-order stream (the reference's CudaVec relies on its Arc<CudaStream> the same way, context/cuda.rs:41-44).  Per-context state (reduction records, scratch,
 * the stream-ordered allocation cache) is only touched under that lock.  dsh_last_error is per thread.  dsh_ctx_bind_thread is kept for callers that issue
 * their own HIP calls on dsh_ctx_stream from a new thread. */
int dsh_ctx_bind_thread(dsh_ctx* ctx);
void* dsh_ctx_stream(dsh_ctx* ctx);
int dsh_ctx_device(dsh_ctx* ctx);
/* threads per workgroup for one-lane-per-system kernels (default 64; tuning knob, power of two in [64,1024]) */
int dsh_ctx_set_block(dsh_ctx* ctx, int threads);

/* 1 when the library was built with -DDSH_EXPERIMENTS (make EXPERIMENTS=1): the measured-slower kernel variants and their DSH_REBIN / DSH_REBIN_STEPS /
 * DSH_MEMBER_SCHED / DSH_LANE_BANDED_V1 / DSH_LU_STREAM_THREADS knobs exist; 0 in the shipped library (the knobs are then ignored). */
int dsh_experiments_enabled(void);
/* HIP-event timing of the dominant kernel on the context's own stream: when enabled every launch of the target (dsh_ctx_set_timing_target; default the
 * device-resident integrators and the fused dsh_bdf_newton_iter / dsh_sdirk_newton_iter launch) is bracketed by two events; dsh_ctx_get_timing returns the number
 * of launches and the summed kernel time in milliseconds since timing was (re-)enabled.  Used by bench.py for the live roofline figures. */
int dsh_ctx_set_timing(dsh_ctx* ctx, int enable);
/* Which launches the event brackets go around while timing is enabled (resets the accumulated time): the device-resident integrators and the fused Newton launch
 * (default), every dsh_lu_solve launch, or every dsh_lu_factor call (staging copy + factor kernel for the matrix-core kernel).  bench.py's per-config rooflines. */
#define DSH_TIMING_RESIDENT 0
#define DSH_TIMING_LU_SOLVE 1
#define DSH_TIMING_LU_FACTOR 2
int dsh_ctx_set_timing_target(dsh_ctx* ctx, int target);
/* Order of operations of the linear solves issued on this context.  DSH_SOLVE_EXACT (default): the reference's getrs order — solutions bit-identical to the CPU
 * path.  DSH_SOLVE_REORDERED (opt-in, like dsh_adaptive_options.deterministic_pow = 2): banded solves of small ensembles (K = 1, n <= 1024, <= 16384 systems) run as
 * chunked affine maps (csrc/dsh_lu_band_affine.hpp: 1/8 of the dependent chain, another association of the same sums, reciprocal instead of division) — equal to
 * the exact solve to ~1e-13 relative on the library's models, NOT bit-comparable; everything else keeps the exact kernels. */
#define DSH_SOLVE_EXACT 0
#define DSH_SOLVE_REORDERED 1
int dsh_ctx_set_solve_mode(dsh_ctx* ctx, int mode);
int dsh_ctx_get_solve_mode(const dsh_ctx* ctx);
/* How blocking reductions wait for the device: poll != 0 (default; env DSH_SYNC_MODE=sync flips it) spins on the sequence tags of the
 * per-workgroup result records the kernels write into pinned host memory; poll == 0 uses hipStreamSynchronize. */
int dsh_ctx_set_poll(dsh_ctx* ctx, int poll);
int dsh_ctx_get_timing(dsh_ctx* ctx, int64_t* launches, double* total_ms);
/* elapsed time of an EMPTY event bracket on this stream (mean of 200, measured when timing was enabled): the part of every bracketed
 * measurement that is not the kernel; and the summed duration of the same timed launches measured inside the kernel with the 100 MHz
 * device clock (max workgroup end - min workgroup start), which is what rocprofv3's kernel trace reports */
int dsh_ctx_get_timing_overhead(dsh_ctx* ctx, double* empty_bracket_ms, double* device_clock_total_ms);

/* ---- device memory (cudarc alloc/alloc_zeros/memcpy_*: call sites throughout vector/cuda.rs, matrix/cuda.rs) ---- */
int dsh_malloc(dsh_ctx* ctx, int64_t nbytes, int zero, void** out);
int dsh_free(dsh_ctx* ctx, void* p);
int dsh_memset_zero(dsh_ctx* ctx, void* p, int64_t nbytes);
int dsh_h2d(dsh_ctx* ctx, void* dst, const void* src, int64_t nbytes); /* blocking */
int dsh_d2h(dsh_ctx* ctx, void* dst, const void* src, int64_t nbytes); /* blocking */
int dsh_d2d(dsh_ctx* ctx, void* dst, const void* src, int64_t nbytes); /* stream-ordered */
/* Vector::from_vec / clone_as_vec (vector/cuda.rs:741-760, :888-906): host data is batch-major [b][i]; device is [i][b]. */
int dsh_vec_upload(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* host_batch_major, double* dev);
int dsh_vec_download(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* dev, double* host_batch_major);
/* Vector::get_index / set_index (vector/cuda.rs:762-779): one element of one batch member */
int dsh_vec_get_index(dsh_ctx* ctx, int64_t nbatch, const double* v, int64_t i, int64_t b, double* out);
int dsh_vec_set_index(dsh_ctx* ctx, int64_t nbatch, double* v, int64_t i, int64_t b, double value);
/* Vector::get_batch / get_batch_mut (vector/mod.rs:227-231): member b of a batched vector as a contiguous vector with nbatch = 1, and back
 * (stream-ordered strided copies; with the batch-fastest layout member b is v[i * nbatch + b]). */
int dsh_vec_extract_batch(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* v, int64_t b, double* dst);
/* members of a batched array re-ordered: dst[r * nbatch + b] = src[r * nbatch + idx[b]], rows of 4- or 8-byte elements, idx on the device.  (No reference
   counterpart: diffsol solves a sweep member by member; here a per-member ensemble is sorted by parameters so that wavefronts hold similar members.) */
int dsh_permute_members(dsh_ctx* ctx, int64_t rows, int64_t nbatch, int elem_bytes, const void* src, const int32_t* idx_dev, void* dst);
int dsh_vec_insert_batch(dsh_ctx* ctx, int64_t n, int64_t nbatch, double* v, int64_t b, const double* src);
/* set element i of EVERY batch member to `value` (the reference does nbatch H2D copies for this, vector/cuda.rs:762-774) */
int dsh_vec_set_index_all(dsh_ctx* ctx, int64_t nbatch, double* v, int64_t i, double value);

/* ---- Vector ops.  n = states per system, nbatch = context batch size.  Kernel each one replaces in parentheses. ---- */
/* ret = lhs + rhs / lhs - rhs                       (vec_add.cu:1, vec_sub.cu:1; vector/cuda.rs:439-484) */
int dsh_vec_add(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* lhs, int64_t lhs_nb, const double* rhs, int64_t rhs_nb, double* ret);
int dsh_vec_sub(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* lhs, int64_t lhs_nb, const double* rhs, int64_t rhs_nb, double* ret);
/* lhs += rhs / lhs -= rhs                           (vec_add_assign.cu:1, vec_sub_assign.cu:1; vector/cuda.rs:396-436) */
int dsh_vec_add_assign(dsh_ctx* ctx, int64_t n, int64_t nbatch, double* lhs, const double* rhs, int64_t rhs_nb);
int dsh_vec_sub_assign(dsh_ctx* ctx, int64_t n, int64_t nbatch, double* lhs, const double* rhs, int64_t rhs_nb);
/* lhs *= rhs / lhs /= rhs elementwise               (vec_mul_assign.cu:1, vec_div_assign.cu:1; vector/cuda.rs:1019-1070) */
int dsh_vec_mul_assign(dsh_ctx* ctx, int64_t n, int64_t nbatch, double* lhs, const double* rhs, int64_t rhs_nb);
int dsh_vec_div_assign(dsh_ctx* ctx, int64_t n, int64_t nbatch, double* lhs, const double* rhs, int64_t rhs_nb);
/* v *= s ; res = v*s                                (vec_mul_assign_scalar.cu:1, vec_mul_scalar.cu:1; vector/cuda.rs:267-387) */
int dsh_vec_mul_assign_scalar(dsh_ctx* ctx, int64_t n, int64_t nbatch, double* v, double s);
int dsh_vec_mul_scalar(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* v, double s, double* res);
/* y = alpha*x + beta*y                              (vec_axpy.cu:1; vector/cuda.rs:937-966, :1459-1489) */
int dsh_vec_axpy(dsh_ctx* ctx, int64_t n, int64_t nbatch, double alpha, const double* x, int64_t x_nb, double beta, double* y);
/* out = alpha*x + beta*y0: `out.copy_from(y0); out.axpy(alpha, x, beta)` in one pass (same arithmetic per entry); copy_x_to, when not NULL, also receives x.
 * out must not alias x or copy_x_to (it may be y0).  What the SDIRK stages use for get_f_eval + the stage's column of diff and for predict_stage
 * (op/sdirk.rs:197-203, runge_kutta.rs:610-689). */
int dsh_vec_axpby_to(dsh_ctx* ctx, int64_t n, int64_t nbatch, double alpha, const double* x, double beta, const double* y0, double* out, double* copy_x_to);
/* y_b = alpha[b]*x_b + beta*y_b, alpha is a HOST array of nbatch values (vec_batched_axpy.cu:4; vector/cuda.rs:967-1012) */
int dsh_vec_batched_axpy(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* alpha_host, const double* x, int64_t x_nb, double beta, double* y);
/* dst = src (broadcast if src_nb==1) ; v = value    (vec_copy.cu:1, vec_fill.cu:1; vector/cuda.rs:208-236, :867-886) */
int dsh_vec_copy(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* src, int64_t src_nb, double* dst);
int dsh_vec_fill(dsh_ctx* ctx, int64_t n, int64_t nbatch, double* v, double value);
/* index ops; idx = DEVICE int32 array shared by all batch members (vec_gather.cu:2, vec_scatter.cu:2,
 * vec_copy_from_indices.cu:2, vec_assign_at_indices.cu:2; vector/cuda.rs:1179-1284)
 *   gather:             dst[k]      = src[idx[k]]   k < nidx   (dst has nidx states, src has n_src states)
 *   scatter:            dst[idx[k]] = src[k]        k < nidx   (src has nidx states, dst has n_dst states)
 *   copy_from_indices:  dst[idx[k]] = src[idx[k]]              (both have n states)
 *   assign_at_indices:  dst[idx[k]] = value                                                              */
int dsh_vec_gather(dsh_ctx* ctx, int64_t n_src, int64_t nbatch, const double* src, const int32_t* idx, int64_t nidx, double* dst);
int dsh_vec_scatter(dsh_ctx* ctx, int64_t n_dst, int64_t nbatch, const double* src, const int32_t* idx, int64_t nidx, double* dst);
int dsh_vec_copy_from_indices(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* src, const int32_t* idx, int64_t nidx, double* dst);
int dsh_vec_assign_at_indices(dsh_ctx* ctx, int64_t n, int64_t nbatch, const int32_t* idx, int64_t nidx, double value, double* dst);
/* max_b (sum_i |x_i|^k)^(1/k)                       (vec_norm.cu:11, vec_norm_lk.cu:8, cublasDnrm2; vector/cuda.rs:71-116, :781-799) */
int dsh_vec_norm(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* x, int k, double* out_max);
/* max_b mean_i (x_i / (|y_i|*rtol + atol_i))^2      (vec_squared_norm.cu:13; vector/cuda.rs:1362-1433).
 * NaN lanes PROPAGATE to the result (the reference's host-side `>` max drops them; its CPU path returns NaN).
 * per_batch_dev (optional, device array of nbatch doubles) receives each system's own value — the hook for per-lane control. */
int dsh_vec_squared_norm(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* x, const double* y, int64_t y_nb, const double* atol,
                         int64_t atol_nb, double rtol, double* out_max, double* per_batch_dev);
/* Vector::root_finding (vec_root_finding.cu:11; vector/cuda.rs:1071-1178): per batch member, found = any g1_i == 0,
 * frac = max over sign changes of |g1_i/(g1_i-g0_i)|, idx = its argmax or -1.  Returns batch 0's triple and
 * DSH_E_BATCH_MISMATCH if (found, idx) is not identical for all batch members. */
int dsh_vec_root_finding(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* g0, const double* g1, int* found, double* frac, int* idx);

/* ---- Matrix / DenseMatrix ops (matrix/cuda.rs:848-1468) ---- */
/* mat = diag(v) (n x n) ; v = diag(mat)             (mat_from_diagonal.cu:2, mat_get_diagonal.cu:2; matrix/cuda.rs:129-160, :1333-1365) */
int dsh_mat_from_diagonal(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* v, int64_t v_nb, double* mat);
/* ---- band containers (the storage of a matrix whose structure is declared: SUNDIALS' band matrix in book/src/benchmarks/sundials.md:27-28; the reference's
 * own containers are dense or CSC, matrix/cuda.rs:848-1089).  Layout as dsh_model_jacobian_band_packed.  self = x + beta*y on band containers of the same
 * (kl, ku) is dsh_mat_scale_add_assign over (kl + ku + 1) * n entries; the factorisation is dsh_lu_factor_packed. */
int dsh_mat_band_from_diagonal(dsh_ctx* ctx, int64_t n, int64_t nbatch, int kl, int ku, const double* v, int64_t v_nb, double* band);
/* y = alpha*A*x + beta*y, columns ascending (Matrix::gemv's order on the entries the container holds) */
int dsh_mat_band_gemv(dsh_ctx* ctx, int64_t n, int64_t nbatch, int kl, int ku, double alpha, const double* band, const double* x, int64_t x_nb, double beta,
                      double* y);
int dsh_mat_get_diagonal(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* mat, double* v);
/* column j of mat (nrows x ncols) = v               (mat_set_column.cu:2; matrix/cuda.rs:1389-1421) */
int dsh_mat_set_column(dsh_ctx* ctx, int64_t nrows, int64_t ncols, int64_t nbatch, double* mat, int64_t j, const double* v, int64_t v_nb);
/* self = x + beta*y over nelem = nrows*ncols entries (mat_scale_add_assign.cu:1; matrix/cuda.rs:1424-1458) — the M - cJ assembly */
/* scale_add_and_assign restricted to the band of n x n matrices (entry (i, j) with -ku <= i - j <= kl): for operands whose structure is declared
 * (dsh_model_band) the dense M - cJ assembly touches (kl+ku+1) n entries per system instead of n^2; same arithmetic per entry */
int dsh_mat_scale_add_assign_banded(dsh_ctx* ctx, int64_t n, int64_t nbatch, int kl, int ku, double* self, const double* x, int64_t x_nb, double beta,
                                    const double* y, int64_t y_nb);
int dsh_mat_scale_add_assign(dsh_ctx* ctx, int64_t nelem, int64_t nbatch, double* self, const double* x, int64_t x_nb, double beta,
                             const double* y, int64_t y_nb);
/* self[dst_idx[k]] = data[src_idx[k]]               (mat_set_data_with_indices.cu:2; matrix/cuda.rs:1137-1174) */
int dsh_mat_set_data_with_indices(dsh_ctx* ctx, int64_t nelem_self, int64_t nelem_data, int64_t nbatch, double* self, const int32_t* dst_idx,
                                  const int32_t* src_idx, int64_t nidx, const double* data);
/* column i += alpha * column j                       (vec_axpy_offset.cu:1; matrix/cuda.rs:1048-1088) */
int dsh_mat_column_axpy(dsh_ctx* ctx, int64_t nrows, int64_t nbatch, double* mat, double alpha, int64_t j, int64_t i);
/* y = alpha*A*x + beta*y, A nrows x ncols            (cublasDgemv host loop over batches; matrix/cuda.rs:620-677, :1267-1293) */
int dsh_mat_gemv(dsh_ctx* ctx, int64_t nrows, int64_t ncols, int64_t nbatch, double alpha, const double* a, int64_t a_nb, const double* x,
                 int64_t x_nb, double beta, double* y);
/* y = alpha*A*x + beta*y0: `y.copy_from(y0); gemv(alpha, x, beta, y)` in one pass (y0 NULL: plain gemv on y).  SdirkCallable::set_phi (op/sdirk.rs:174-184). */
int dsh_mat_gemv_from(dsh_ctx* ctx, int64_t nrows, int64_t ncols, int64_t nbatch, double alpha, const double* a, int64_t a_nb, const double* x, int64_t x_nb,
                      double beta, const double* y0, double* y);
/* C = alpha*A*B + beta*C, A m x k, B k x n, C m x n  (cublasDgemmStridedBatched, stride 0 = broadcast; matrix/cuda.rs:757-822, :960) */
int dsh_mat_gemm(dsh_ctx* ctx, int64_t m, int64_t n, int64_t k, int64_t nbatch, double alpha, const double* a, int64_t a_nb, const double* b,
                 int64_t b_nb, double beta, double* c);

/* ---- LinearSolver (diffsol-la/src/linear_solver/mod.rs:19-42; CudaLU linear_solver/cuda/lu.rs:59-191) ---- */
/* set_sparsity: allocate n*n*nbatch factors + n*nbatch int32 pivots (lu.rs:148-190) */
int dsh_lu_create(dsh_ctx* ctx, int64_t n, int64_t nbatch, dsh_lu** out);
void dsh_lu_destroy(dsh_lu* lu);
/* set_linearisation after op.matrix_inplace: partial-pivot LU of all nbatch systems in ONE launch
 * (replaces the serial `for b in 0..nbatch { cusolverDnDgetrf }` loop, lu.rs:80-95).  `a` (n*n*nbatch, device) is not modified.
 * BIT PARITY: every kernel behind this call produces the factors, pivots and solutions of the CPU path bit for bit EXCEPT the default dense kernel for
 * 288 <= n <= 1024 (the FP64 matrix-core kernel of dsh_lu_tiled.hpp): same pivots, factors within ~2e-13 of the largest entry (fused multiply-adds, the
 * matrix cores' summation order, a reciprocal instead of a division per pivot); it also keeps a second n*ldw*nbatch working copy per handle.
 * DSH_LU_EXACT=1 (read per call) selects the bit-exact blocked kernel for that range instead (2.5 - 6x slower, profiles/r04_lu_bench.md). */
int dsh_lu_factor(dsh_lu* lu, const double* a);
/* solve_in_place, nrhs = 1, all systems in ONE launch (replaces the getrs host loop, lu.rs:127-145) */
int dsh_lu_solve(const dsh_lu* lu, double* b);
/* dsh_lu_solve(lu, x) followed by dsh_vec_squared_norm(x, y, atol, rtol) with ONE wait for both results (the zero-pivot count and the norm): the error
 * estimate of an SDIRK step (sdirk.rs:474-495, runge_kutta.rs:783-800).  Same kernels, same bits; DSH_E_SINGULAR like dsh_lu_solve. */
int dsh_lu_solve_squared_norm(const dsh_lu* lu, double* x, const double* y, int64_t y_nb, const double* atol, int64_t atol_nb, double rtol, double* out_norm);
/* nrhs right-hand sides per system with the same factors (the linear algebra of forward sensitivities: Bdf::sensitivity_solve, bdf.rs:934-989, solves one
 * system per parameter with the LU of the state equations): b is an n x nrhs matrix in the library's layout, column r at b + r*n*nbatch.  For n <= 8 one launch
 * loads the factors once for all columns; every column's solution equals dsh_lu_solve's bit for bit. */
int dsh_lu_solve_multi(const dsh_lu* lu, double* b, int64_t nrhs);
/* number of systems whose factorisation met an exactly-zero pivot (cusolver `info`, ignored by the reference lu.rs:83-95); blocking */
int dsh_lu_info(const dsh_lu* lu, int64_t* n_singular);
/* raw device pointers of the factor storage: batch-fastest for n <= 8, system-major for n > 8 (dsh_lu_system_major) */
double* dsh_lu_factors(dsh_lu* lu);
int32_t* dsh_lu_pivots(dsh_lu* lu);
int dsh_lu_system_major(const dsh_lu* lu);
/* Banded matrices in dense containers (PDE-type and compartment models; SURVEY 8(f) row 4 — the reference has no banded solver,
 * book/src/benchmarks/sundials.md:27-28).  With DSH_LU_STRUCTURE_AUTO (default for n >= 16; env DSH_LU_STRUCTURE=dense switches the default)
 * dsh_lu_factor reads the operand once to find its bandwidths (kl, ku) over all systems and factors and solves only the band (LAPACK dgbtrf-style
 * partial pivoting with kl fill-in diagonals): max(kl, ku) <= 4 with one lane per system and the window in registers (dsh_lu_band.hpp); 4 < max(kl, ku) <= 64,
 * n <= 1024 and (2 kl + ku + 1) 2 <= n with one WAVEFRONT per system and the window in LDS (dsh_lu_gband.hpp, round 6: the 2-D PDE models heat2d / foodweb;
 * DSH_LU_GBAND=0 disables).  Every non-trivial operation of the dense elimination is performed in the same order, so solutions are BIT-IDENTICAL to the dense
 * kernels'; only time and traffic change.  dsh_lu_band_width: max(kl, ku) of the current factors, 0 = dense. */
#define DSH_LU_STRUCTURE_AUTO 0
#define DSH_LU_STRUCTURE_DENSE 1
int dsh_lu_set_structure(dsh_lu* lu, int structure);
/* the same with the band DECLARED by the caller (a model that knows the structure of its Jacobian, dsh_model_band): no probe pass.  Entries with
 * |i - j| > max(kl, ku) are not read.  DSH_CHECK_BAND=1 (debug) probes anyway and fails if the declaration is wrong. */
int dsh_lu_factor_banded(dsh_lu* lu, const double* a, int kl, int ku);
/* An LU handle for banded systems only: (3k + 1) n doubles of factor storage per system instead of n^2; takes dsh_lu_factor_packed with max(kl, ku) <= k
 * (1 <= k <= 64, n >= 16; k > 4: n <= 1024); dsh_lu_solve as for any handle; dense operands are refused. */
int dsh_lu_create_banded(dsh_ctx* ctx, int64_t n, int64_t nbatch, int k, dsh_lu** out);
/* Factor a band container: the eliminations of dsh_lu_factor_banded on the same entries (bit-identical factors and solutions).  Works on any handle. */
int dsh_lu_factor_packed(dsh_lu* lu, const double* band, int kl, int ku);
int dsh_lu_band_width(const dsh_lu* lu);
/* packed LU factors as [b][col][row] and pivot rows as [b][k] on the host, whatever the device layout; blocking */
int dsh_lu_download(dsh_lu* lu, double* factors_host, int32_t* pivots_host);

/* ---- Model registry: the OdeEquations plug-in boundary (diffsol/src/ode_equations/mod.rs:245-329; NonLinearOp::call_inplace,
 * NonLinearOpJacobian::{jac_mul_inplace,jacobian_inplace} op/nonlinear_op.rs:10-21,175-221; LinearOp::{gemv_inplace,matrix_inplace}
 * op/linear_op.rs:9-54).  One lane per system; parameters p are a batched vector of nparams entries (batch-fastest on device). ---- */
#define DSH_MODEL_EXPONENTIAL_DECAY 0           /* n=2, p=[k,y0]            test_models/exponential_decay.rs:14-81 */
#define DSH_MODEL_EXPONENTIAL_DECAY_ALGEBRAIC 1 /* n=3, p=[k], init (1,1,0) test_models/exponential_decay_with_algebraic.rs:18-126 */
#define DSH_MODEL_EXPONENTIAL_DECAY_ALGEBRAIC_BATCHED 2 /* same, init (1,1,1) :202-276 */
#define DSH_MODEL_ROBERTSON_ODE 3               /* n=3*size, p=[k1,k2,k3]   test_models/robertson_ode.rs:71-101 */
#define DSH_MODEL_ROBERTSON_DAE 4               /* n=3, M=diag(1,1,0)       test_models/robertson.rs:60-94 */
#define DSH_MODEL_DYDT_Y2 5                     /* n=size                   test_models/dydt_y2.rs:9-19 */
#define DSH_MODEL_GAUSSIAN_DECAY 6              /* n=size, p=[a]*size       test_models/gaussian_decay.rs:12-23 */
#define DSH_MODEL_HEAT1D 7                      /* n=size, p=[D]            test_models/heat1d.rs:16-52, examples/pde-heat/src/main.rs:15-40 */
#define DSH_MODEL_RLC 8                         /* n=4 DAE, p=[R,L,C,V0,omega,ithresh]; size!=0 adds root iR-ithresh  examples/electrical-circuits/src/main.rs:10-41 */
#define DSH_MODEL_EXPONENTIAL_DECAY_ROOT 9      /* exponential decay + root x0-0.6  test_models/exponential_decay.rs:98-100 */
#define DSH_MODEL_SPM 10                        /* single-particle battery model, n=2+2*size (size=0 -> 20 shells), p=[I], roots V-3.105, 4.1-V  book/src/primer/src/spm.ds */
#define DSH_MODEL_HEAT2D 11                     /* 2-D heat equation, size x size grid, n=size^2, DAE (boundary rows algebraic), p=[diffusion scale] (1 = the reference), band size   test_models/heat2d.rs:105-205 */
#define DSH_MODEL_FOODWEB 12                    /* predator-prey food web, size x size grid, n=2 size^2, DAE (predators algebraic), p=[alpha,beta] ((50,1000) = the reference), band 2 size   test_models/foodweb.rs:232-655 */

/* ---- Run-time-compiled models: the device side of OdeBuilder::build_from_diffsl (crates/diffsol/src/ode_equations/diffsl.rs —
 * the reference JIT-compiles DiffSL to host code with Cranelift/LLVM).  `source` is the model as generated by dshs_diffsl_generate
 * (diffsol_hip_solver.h): `struct dsh::JitModel` (DSH_JIT_FORM_STATIC, n <= 8, at most one root function) or the jit_* component functions
 * (DSH_JIT_FORM_DYNAMIC).  It is compiled with hiprtc together with the library's own kernel templates, so the returned model id (>= DSH_MODEL_JIT_BASE)
 * works wherever a registry id does: dsh_model_*, the fused Newton kernels (static form), the device-resident integrators (static form), and the
 * host-side integrators of diffsol_hip_solver.h.  Kernel families are compiled on first use and cached on disk (DSH_JIT_CACHE=<dir> | off; default <package>/_jit_cache, keyed by the
 * translation unit, the options and the library's headers); dsh_model_precompile (family 0 operators, 1 fused Newton,
 * 2 resident BDF, 3 resident SDIRK) pays that cost up front and needs no GPU.  `model_size` arguments are ignored for these ids. */
#define DSH_MODEL_JIT_BASE 1000
#define DSH_JIT_FORM_STATIC 0
#define DSH_JIT_FORM_DYNAMIC 1
#define DSH_JIT_FORM_STATIC_BANDED 2 /* `struct dsh::JitModel` with BAND_K and jac_band (n <= 512 — DiffSL-emitted forms n <= 64 —, identity mass, Jacobian bandwidth <= 4): only the lane-per-member
                                      * device-resident BDF is instantiated for it (state in per-lane memory, banded LU in registers); attach it to the
                                      * run-time-sized form of the same model with dsh_model_set_twin and per-member solve_dense uses it */
int dsh_model_compile(const char* source, int form, int64_t nstates, int64_t nparams, int64_t nroots, int64_t nout, int has_mass, int* model_id);
int dsh_model_release(int model_id);
int dsh_model_precompile(int model_id, int family);
/* code objects this process compiled itself with hiprtc (not loaded from the on-disk cache, DSH_JIT_CACHE): one process per GPU shares the cache, and an
 * exclusive lock per entry makes the first rank that meets a model compile it while the others wait and load (tests/test_dist_cpu.py) */
int64_t dsh_jit_compile_count(void);
/* Manifest of compile requests: with DSH_JIT_RECORD=<file> in the environment every request for a module (served from the cache or compiled) is appended to <file>.
 * dsh_jit_replay compiles the requests i of such a manifest with i % nparts == part into the on-disk cache (no GPU needed, nothing is loaded): the build step replays the
 * committed manifests (diffsol_amd/jit_manifest/) so that a fresh box never compiles at first use.  *requests = distinct records read, *compiled = modules this call compiled. */
int dsh_jit_replay(const char* manifest_path, int part, int nparts, int64_t* requests, int64_t* compiled);
/* A static model (n <= 8) has device-resident integrators up to n = 4; for 5 <= n <= 8 the DiffSL front ends register the same model in the run-time-sized form
 * (DSH_JIT_FORM_DYNAMIC source) and the library compiles it at the first per-member request: dsh_model_member_twin returns its id (-1: none). */
int dsh_model_set_member_twin_source(int model_id, const char* source, int64_t nstates, int64_t nparams, int64_t nroots, int64_t nout);
int dsh_model_member_twin(int model_id);
int dsh_model_set_twin(int model_id, int twin_id);
int dsh_model_twin(int model_id); /* -1: none */
/* the same for any model id: a run-time-compiled model's twin, or — created on first request — the banded lane-per-member form of a built-in
 * run-time-sized model (heat1d, the single-particle model, ...; csrc/dsh_models_lane.hpp).  -1: the model has no such form. */
int dsh_model_lane_twin(int model, int64_t size);
/* structural bandwidths of f_y and of the mass matrix (dshs_diffsl_generate reports them for a DiffSL model): -1 = dense / unknown */
int dsh_model_set_band(int model_id, int jac_kl, int jac_ku, int mass_kl, int mass_ku);
int dsh_model_band(int model, int64_t size, int* jac_kl, int* jac_ku, int* mass_kl, int* mass_ku);
/* out_i of a run-time-compiled model (DiffSl::out, calc_out): out is nout x nbatch, batch-fastest */
int dsh_model_out(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, const double* x, const double* p, double* out);

int dsh_model_info(int model, int64_t size, int64_t* nstates, int64_t* nparams, int* has_mass, int64_t* nroots);
int dsh_model_rhs(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, const double* x, const double* p, double* y);
int dsh_model_jac_mul(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, const double* x, const double* p, const double* v, double* y);
int dsh_model_jacobian(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, const double* x, const double* p, double* jac);
/* the same entries on the band (kl, ku) only, for a container whose entries outside the band are already zero; run-time-sized registry models with a
   declared band (dsh_model_band) covered by (kl, ku).  What a banded matrix type would evaluate (book/src/benchmarks/sundials.md:27-28 notes its absence). */
int dsh_model_has_band_jacobian(int model, int64_t size); /* 1 if dsh_model_jacobian_band serves this model */
int dsh_model_jacobian_band(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, const double* x, const double* p, int kl, int ku, double* jac);
/* ... into a BAND CONTAINER: entry (i, j), -kl <= j - i <= ku, at ((j - i + kl) * n + i) * nbatch + b — (kl + ku + 1) n doubles per member instead of n^2
 * (config 3: 12 KB instead of 2 MB per matrix and member).  Every entry of the container is written (the corners outside the matrix as zeros). */
int dsh_model_jacobian_band_packed(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, const double* x, const double* p, int kl, int ku,
                                   double* band);
int dsh_model_mass_gemv(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, const double* x, const double* p, double beta, double* y);
int dsh_model_mass_matrix(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, const double* p, double* mass);
int dsh_model_init(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, const double* p, double* y);
int dsh_model_root(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, const double* x, const double* p, double* g);
/* Forward sensitivities (OdeEquationsImplicitSens): df/dp at (x, t) and dy0/dp as n x nparams batched matrices (column j =
 * NonLinearOpSens::sens_mul / ConstantOpSens::sens_mul with the unit vector e_j: op/nonlinear_op.rs:51-81, ode_equations/sens_equations.rs:62-70),
 * one launch each.  Built-in models with parameter derivatives: exponential decay (test_models/exponential_decay.rs:33-36, :90-93) and the Robertson
 * ODE (test_models/robertson_ode_with_sens.rs:38-50); dsh_model_has_sens tells. */
/* reset operator of hybrid models (OdeEquations::reset, crates/diffsol/src/ode_equations/mod.rs; DiffSL reset_i, ode_equations/diffsl.rs:872): y = reset(x, t),
   the state after an event; the solver applies it at every root and continues (ode_solver/method.rs:774-797) */
int dsh_model_has_reset(int model, int64_t size);
int dsh_model_reset(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, const double* x, const double* p, double* y);
int dsh_model_has_sens(int model, int64_t size);
int dsh_model_rhs_sens(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, const double* x, const double* p, double* sens);
int dsh_model_init_sens(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, const double* p, double* sens0);

/* ---- Fused fast paths (optional; results are bit-identical to composing the 1:1 ops above) ---- */
/* One Newton iteration of the BDF residual F(y) = M(y + (psi - y0)) - c f(y)  (NoLineSearch::take_optimal_step
 * diffsol-nl/src/line_search.rs:48-69 over BdfCallable::call_inplace diffsol/src/op/bdf.rs:240-256):
 *   y = y_in; delta = F(y); LU-solve(delta); y_out = y - delta; out[0] = max_b ||delta||^2_(error_y)  (Convergence::norm squared, convergence.rs:64-66)
 * and, speculatively, the error-test quantity for the step (Bdf::error_control, ode_solver/bdf.rs:826-835, without the error constant):
 *   out[1] = max_b ||y_out - error_y||^2_(y_old)      (only if y_old != NULL)
 * out[2] = number of systems whose solve met a zero pivot (as a double).  out is a HOST array of 3 doubles.  Blocking.
 * y_in may equal y_out (in place) or error_y (first iteration: `y_delta.copy_from(&y_predict)`, bdf.rs:1326, without a copy launch).
 * The _async form only enqueues the launch and returns a ticket; dsh_reduction_wait(ticket) collects the three results later.  Up to 7
 * further reducing launches may be issued before a ticket is redeemed, which lets the host keep one speculative iteration (writing to a
 * second iterate buffer) in flight while it evaluates the convergence test of the previous one — no idle GPU during the host round trip.
 * Supported for models with a register-resident specialisation (dsh_model_has_fused). */
int dsh_bdf_newton_iter(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, double c, const double* y_in, double* y_out,
                        const double* psi_neg_y0, const double* p, const dsh_lu* lu, const double* error_y, const double* y_old,
                        const double* atol, int64_t atol_nb, double rtol, double* out);
/* nit in 1..4 consecutive iterations in ONE launch: iterate i (0-based) is written to y_out + i*n*nbatch, and dsh_reduction_wait fills
 * out[3*i .. 3*i+3) with that iteration's (norm^2, error^2, singular count).  The host evaluates the convergence test over the nit norms
 * in order and uses the first iterate that converged; later iterates are speculative work that reused the registers of the launch. */
int dsh_bdf_newton_iter_async(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, double c, int nit, const double* y_in,
                              double* y_out, const double* psi_neg_y0, const double* p, const dsh_lu* lu, const double* error_y,
                              const double* y_old, const double* atol, int64_t atol_nb, double rtol, int64_t* ticket);
int dsh_reduction_wait(dsh_ctx* ctx, int64_t ticket, double* out);
/* Same for the SDIRK stage residual F(k) = M k - h f(phi + c k)  (SdirkCallable::call_inplace diffsol/src/op/sdirk.rs:229-244);
 * out[0] = max_b ||delta||^2_(error_y), out[2] = singular count. */
int dsh_sdirk_newton_iter(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, double h, double c, const double* k_in, double* k_out,
                          const double* phi, const double* p, const dsh_lu* lu, const double* error_y, const double* atol, int64_t atol_nb,
                          double rtol, double* out);
int dsh_sdirk_newton_iter_async(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, double h, double c, int nit, const double* k_in,
                                double* k_out, const double* phi, const double* p, const dsh_lu* lu, const double* error_y, const double* atol,
                                int64_t atol_nb, double rtol, int64_t* ticket);
/* Jacobian refresh: [rhs_jac = J(x,t) if recompute_rhs_jac; mass_jac = M(t) if the model has a mass matrix];
 * A = mass_jac + (-c)*rhs_jac; LU-factor A into `lu` — one launch, A never touches HBM
 * (BdfCallable::jacobian_inplace op/bdf.rs:273-300 + CudaLU::set_linearisation lu.rs:59-97). */
/* Sdirk::step's bookkeeping between two Newton solves, single passes (round 5; each equals the sequence of Vector / Matrix trait operations it replaces, bit for bit):
 *   dsh_sdirk_begin_attempt  diff[:,0] = h dy (start_step_attempt, runge_kutta.rs:505-516); phi = y0 + diff[:,0] a10 (set_phi, op/sdirk.rs:174-184); k = diff[:,0] (predictor)
 *   dsh_sdirk_next_stage     stage `stage` converged with increment k: y_stage = c k + phi (get_f_eval :197-203); diff[:,stage] = k; phi = y0 + diff[:,0..=stage] a_next;
 *                            k = pred_a diff[:,stage-1] + pred_b diff[:,stage] (predict_stage_sdirk, runge_kutta.rs:610-629)
 *   dsh_sdirk_finish_error   the last stage converged: y_stage = c k + phi; diff[:,s-1] = k; err = diff[:,0..s) d (runge_kutta.rs:783-800)
 * diff: n x s x nbatch, column-major per member, batch-fastest; every vector n x nbatch; coefficients on the host. */
int dsh_sdirk_begin_attempt(dsh_ctx* ctx, int64_t n, int64_t nbatch, double h, double a10, const double* dy, const double* y0, double* diff0, double* phi, double* k);
int dsh_sdirk_next_stage(dsh_ctx* ctx, int64_t n, int64_t nbatch, int stage, double c, double* k, double* phi, const double* y0, double* y_stage, double* diff,
                         const double* a_next_host, double pred_a, double pred_b);
int dsh_sdirk_finish_error(dsh_ctx* ctx, int64_t n, int64_t nbatch, int nstages, double c, const double* k, const double* phi, double* y_stage, double* diff,
                           const double* d_host, double* err);
int dsh_jac_factor(dsh_ctx* ctx, int model, int64_t size, int64_t nbatch, double t, double c, const double* x, const double* p,
                   int recompute_rhs_jac, double* rhs_jac, double* mass_jac, dsh_lu* lu);
int dsh_model_has_fused(int model, int64_t size);
/* 1 when dsh_sdirk_newton_iter runs the model's SDIRK Newton iteration in its staged form: run-time-sized registry models without a mass matrix (heat1d, spm:
 * no register-resident kernels) — residual / LU solve / update + norm as three launches and one wait instead of the seven launches and two waits of the
 * host-driven vector operations, bit-identical iterates (op/sdirk.rs:229-244, diffsol-nl line_search.rs:43-72) */
int dsh_model_has_staged_newton(int model, int64_t size);
/* BDF step preparation, one launch (Bdf::_update_diff_for_step_size ode_solver/bdf.rs:568-577, _predict_using_diff :667-672,
 * BdfCallable::set_psi_and_y0 op/bdf.rs:182-210):
 *   if ru_host != NULL: diff_tmp[:,0..=order] = diff[:,0..=order] * RU  (RU is (order+1)^2 column-major on the HOST; caller swaps
 *   diff/diff_tmp afterwards exactly like the reference) and the prediction uses the rescaled columns;
 *   y_predict = sum_{i<=order} D[:,i];  psi_neg_y0 = alpha*(sum_{1<=i<=order} gamma_i D[:,i]) - y_predict.
 * gamma_host has order+1 entries. */
int dsh_bdf_prepare_step(dsh_ctx* ctx, int64_t n, int64_t nbatch, int order, const double* diff, double* diff_tmp, const double* ru_host,
                         const double* gamma_host, double alpha, double* y_predict, double* psi_neg_y0);
/* BDF accepted-step update, one launch (Bdf::_update_diff :646-664, state update :1472-1478, predict_error_control :871-900):
 *   d = y_new - y_predict; D[:,k+2] = d - D[:,k+1]; D[:,k+1] = d; D[:,i] += D[:,i+1] for i=k..0; y = y_predict; dy = D[:,1]/h;
 *   out[0] = max_b ||D[:,k]||^2_(y)   (order-1 candidate, 0 if k==1), out[1] = max_b ||D[:,k+2]||^2_(y) (order+1 candidate).
 * out is a HOST array of 2 doubles, filled only if want_norms != 0 (then blocking).
 * If psi_neg_y0_next != NULL the same launch also writes the prediction for the NEXT step at unchanged order / step size
 * (y_predict and psi_neg_y0_next exactly as dsh_bdf_prepare_step would from the updated D, gamma_host / alpha as there); the caller
 * uses it when the controller leaves h and the order alone and otherwise overwrites it with dsh_bdf_prepare_step. */
int dsh_bdf_accept_step(dsh_ctx* ctx, int64_t n, int64_t nbatch, int order, double h, double* diff, double* y_predict, const double* y_new,
                        double* y, double* dy, const double* atol, int64_t atol_nb, double rtol, const double* gamma_host, double alpha,
                        double* psi_neg_y0_next, int want_norms, double* out);
/* enqueue only; redeem the two norms with dsh_reduction_wait(ticket, out3) (out[0], out[1]) if and when they are needed */
int dsh_bdf_accept_step_async(dsh_ctx* ctx, int64_t n, int64_t nbatch, int order, double h, double* diff, double* y_predict, const double* y_new,
                              double* y, double* dy, const double* atol, int64_t atol_nb, double rtol, const double* gamma_host, double alpha,
                              double* psi_neg_y0_next, int64_t* ticket);
/* The accepted-step launch of step k fused with the first `nit` Newton iterations of step k+1 (valid when the controller then keeps order, step size
 * and LU factors — the caller discards the Newton part otherwise): dsh_bdf_accept_step_async + dsh_bdf_newton_iter_async in one launch, the new
 * state / prediction / psi passed in registers.  Two tickets over the same launch: accept_ticket (1 group: order-selection norms) and newton_ticket
 * (nit groups).  Fused static models only. */
int dsh_bdf_accept_newton_async(dsh_ctx* ctx, int model, int64_t size, int64_t nb, int order, double h, double* diff, double* y_predict, const double* y_new,
                                double* y, double* dy, const double* atol, int64_t atol_nb, double rtol, const double* gamma_host, double alpha,
                                double* psi_neg_y0_next, double t_next, double c, int nit, double* y_out, const double* p, const dsh_lu* lu,
                                int64_t* accept_ticket, int64_t* newton_ticket);


/* ---- device-resident per-member adaptive BDF: the whole ensemble solve in ONE launch, one lane per member, each with its own
 * step-size / order history — the semantics of diffsol's CPU path for a parameter sweep (one independent IVP per member), i.e. of
 * Bdf::step (ode_solver/bdf.rs:1277-1589) + NewtonNonlinearSolver (diffsol-nl/src/newton.rs) + solve_dense (method.rs:467-520) per member.
 * Static models with n <= 4 (dsh_model_has_adaptive), mass matrices (consistent initialisation on the device) and root functions included; and run-time-compiled
 * models in the banded lane-per-member form (DSH_JIT_FORM_STATIC_BANDED, dsh_model_lane_twin: built-in models n <= 512, DiffSL models n <= 64; identity mass, Jacobian bandwidth <= 4), whose state lives in
 * per-lane memory and whose LU is banded. */
typedef struct dsh_adaptive_options { /* OdeSolverOptions (problem.rs:132-152) + BdfConfig (config.rs:53-74) */
  int max_nonlinear_solver_iterations, max_error_test_failures, max_nonlinear_solver_failures;
  double nonlinear_solver_tolerance, min_timestep;
  double max_timestep_growth, min_timestep_growth, max_timestep_shrink, min_timestep_shrink;
  int update_jacobian_after_steps, update_rhs_jacobian_after_steps;
  double threshold_to_update_jacobian, threshold_to_update_rhs_jacobian;
  double pi_control_proportional, pi_control_integral;
  /* InitialConditionSolverOptions (problem.rs:15-45): consistent initialisation of DAEs */
  int ic_use_linesearch, ic_max_linesearch_iterations, ic_max_linear_solver_setups, ic_max_newton_iterations;
  double ic_step_reduction_factor, ic_armijo_constant;
  int64_t max_steps; /* per-member guard against a runaway loop (status 99) */
  int deterministic_pow; /* arithmetic mode.  1 (default): pow() of diffsol_detpow.h — the results are bit-identical to the oracle's in the same mode;
                            0: ocml's pow(), everything else exact;
                            2: the opt-in FAST variant of dsh_bdf_solve_adaptive for static models (dsh_adaptive_fast.hip, compiled with -ffp-contract=fast
                               and reciprocal-math division; ocml pow; reciprocal Newton weights): ~1.2x, NOT bit-comparable with the oracle, states within
                               1e-6 relative at tight tolerances; every other kernel treats 2 like 1 */
  int group;         /* control granularity: 1 = every member its own step/order history; 64 = the 64 members of a wavefront in lock-step
                        (the reference's batched semantics with nbatch = 64 per group, max-norms over the wavefront) */
} dsh_adaptive_options;
void dsh_adaptive_default_options(dsh_adaptive_options* opts);
int dsh_model_has_adaptive(int model, int64_t size);
/* p: np x nb (batch-fastest, device); atol: n (atol_nb == 1) or n x nb; t_eval_host: n_eval increasing times, the last one is the stop time;
 * y_out: n_eval x n x nb (device, batch-fastest per save point); stats: 5 x nb int32 (steps, Newton iterations, LU setups, error-test failures,
 * Newton failures) or NULL; status: nb int32 (0 ok, else the OdeSolverError ordinal, 99 = max_steps) or NULL;
 * t_root / root_idx / ncols (nb each, may be NULL): see dsh_sdirk_solve_resident.  totals_host[6]: the five counters summed over members + number of
 * failed members.  Blocking. */
int dsh_bdf_solve_adaptive(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                           double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats,
                           int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host);
/* OdeSolverMethod::solve (crates/diffsol/src/ode_solver/method.rs:227-258 over :881-961) inside ONE launch of the register-resident BDF: every member's state after
 * EVERY accepted step instead of interpolated save points.  Column 0 is (t0, y0) (write_out before the first step), then one column per InternalTimestep, the last
 * one at the stop time t_final (TstopReached) or — models with root functions — at the member's event (state_mut_back(t_root), RootFound); a configured reset
 * operator writes the reset state at the root time and goes on (method.rs:927-941).  group = 1: every member its own columns and count; group = 64: the 64
 * members of a wavefront share them (the reference's batched semantics).  y_out: max_cols x n x nb, t_out: max_cols x nb (device, batch-fastest);
 * ncols[b]: the columns member b PRODUCED — columns beyond max_cols are counted, not stored, so ncols[b] > max_cols says "call again with more room" (the
 * reference grows its matrix instead, method.rs:977-980).  Static models with n <= 4, built-in or run-time-compiled (dsh_model_has_adaptive_steps); other
 * arguments as dsh_bdf_solve_adaptive. */
int dsh_model_has_adaptive_steps(int model, int64_t size);
int dsh_bdf_solve_adaptive_steps(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                                 double h0, const dsh_adaptive_options* opts, double t_final, int64_t max_cols, double* y_out, double* t_out, int32_t* stats,
                                 int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host);
/* HYBRID models (a reset operator: OdeEquations::reset, DiffSL reset_i) in the register-resident form (n <= 4, identity mass): dsh_bdf_solve_adaptive and
 * dsh_sdirk_solve_resident (runge_kutta.rs:396-464, sdirk.rs:368-374) handle every event inside the launch the way the reference's solve_dense does when a reset is configured (method.rs:774-797): the save points up to the root from the step's
 * polynomial, state moved back to the root (bdf.rs:1232-1262), y <- reset(y, t), dy <- f(y, t) (bdf.rs:1017-1020), stop time armed again, restart from the modified
 * state at first order (bdf.rs:1290-1318), on to the last save point.  Per member with group = 1 (every member its own event times); t_root / root_idx report the
 * member's LAST event.  Bit-identical to the oracle's per-member solve_dense with resets. */
int dsh_model_has_adaptive_reset(int model, int64_t size);
/* The same with FORWARD SENSITIVITIES (problem.bdf_sens(), problem.rs:819-832; Bdf::new_augmented bdf.rs:370-432, sensitivity_solve :934-989, the sensitivity
 * terms of error_control :844-858 and predict_error_control :871-932, interpolate_sens :1162-1215, solve_dense_sensitivities sensitivities.rs:114-260):
 * s_j = dy/dp_j of every parameter integrated alongside, in the same launch — per step and parameter one Newton solve with the factors of the state equations
 * and the shared Convergence, the sensitivity difference arrays rescaled and updated with the states'.  ODE models in the register-resident form with parameter
 * derivatives (built-in, or DiffSL / external models compiled at run time), n <= 4, identity mass, no root functions (dsh_model_has_adaptive_sens).  nsens_atol = 0: turn_off_sensitivities_error_control; else sens_rtol / sens_atol_host
 * (length 1 or n, the same for every parameter and member) put the sensitivities into the error test and the order selection.
 * sens_out: n_eval x np x n x nb (device, batch-fastest per save point and parameter).  Results are bit-identical to the oracle's per-member
 * (group = 1) resp. 64-member lock-step (group = 64) solves with sensitivities in the deterministic-pow mode. */
int dsh_model_has_adaptive_sens(int model, int64_t size);
int dsh_bdf_solve_adaptive_sens(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                                double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double sens_rtol,
                                const double* sens_atol_host, int64_t nsens_atol, double* y_out, double* sens_out, int32_t* stats, int32_t* status,
                                int64_t* totals_host);
/* dsh_sdirk_solve_resident with FORWARD SENSITIVITIES (problem.tr_bdf2_sens() / esdirk34_sens(); runge_kutta.rs:196-232 new_augmented, :691-748 the sensitivity half
 * of do_stage_sdirk, :812-822 sensitivities in the error norm, :1237-1330 interpolate_sens; sdirk.rs:251 the linearisation at construction): arguments and scope as
 * dsh_bdf_solve_adaptive_sens, method = 1 (TR-BDF2) or 2 (ESDIRK34). */
int dsh_sdirk_solve_resident_sens(dsh_ctx* ctx, int method, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                                  double t0, double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double sens_rtol,
                                  const double* sens_atol_host, int64_t nsens_atol, double* y_out, double* sens_out, int32_t* stats, int32_t* status,
                                  int64_t* totals_host);
/* Device-resident TR-BDF2 (method 1) / ESDIRK34 (method 2): Sdirk::step (ode_solver/sdirk.rs:409-543) + Rk core (runge_kutta.rs) + consistent DAE initialisation
 * (state.rs:84-162) + RootFinder (nonlinear_solver/root.rs) + solve_dense (method.rs:467-520) per member, one launch per ensemble solve.  Static models with
 * n <= 4, mass matrices and root functions included, and the banded lane-per-member form as above (dsh_model_has_resident).  A member that finds a root stops there: its column after the drained save points holds
 * the state at the root (solve_dense's return), ncols[b] counts its valid columns, later columns are NaN.  t_root / root_idx / ncols (nb each) may be NULL.
 * With group = 64 the members of a wavefront must agree on the crossing (status 20 otherwise, where the reference panics, vector/cuda.rs:1166-1171). */
/* Device-resident BDF for run-time-sized models with n <= 64 (built-in or DiffSL; DiffSL models with a mass matrix — DAEs, made consistent on the device — n <= 48; the fallback for models without a banded lane-per-member form): ONE WAVEFRONT per member, lane = state component, the LU of
 * M - cJ in the wavefront's registers, per-member step sizes / orders / event stops, no host in the loop (dsh_wave_member.hip).
 * Identity-mass models with 64 < n <= 320 (dense Jacobians: the sizes between the wavefront form and the host-driven path) run ONE WORKGROUP per member instead
 * (dsh_team_member_kernel.hpp: thread = state component, the LU of M - cJ in the CU's 160 KB LDS): dsh_model_has_wave_member returns 1 for the wavefront form,
 * 2 for the workgroup form, 0 for neither; dsh_bdf_solve_wave_member takes both.
 * dsh_sdirk_solve_wave_member: the same for TR-BDF2 (method 1) / ESDIRK34 (method 2) — Sdirk::step (sdirk.rs:409-543) over Rk (runge_kutta.rs:466-960),
 * the same models (dsh_model_has_wave_member_sdirk: 1 a wavefront per member, 2 a workgroup per member — 64 < n <= 320, identity mass), DAEs included.
 * Arguments and outputs as dsh_sdirk_solve_resident (opts->group is ignored: control is always per member). */
/* the same for the device-resident TR-BDF2 (method 1) / ESDIRK34 (method 2): every accepted step of every member out (the models of dsh_model_has_resident) */
int dsh_sdirk_solve_resident_steps(dsh_ctx* ctx, int method, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                                   double t0, double h0, const dsh_adaptive_options* opts, double t_final, int64_t max_cols, double* y_out, double* t_out, int32_t* stats,
                                   int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host);
int dsh_model_has_wave_member(int model, int64_t size);
/* OdeSolverMethod::solve (method.rs:227-258) inside the launch of the wavefront- / workgroup-per-member BDF: every accepted step of every member out (arguments as
 * dsh_bdf_solve_adaptive_steps; the models of dsh_model_has_wave_member, n <= 320) */
int dsh_bdf_solve_wave_member_steps(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                                    double h0, const dsh_adaptive_options* opts, double t_final, int64_t max_cols, double* y_out, double* t_out, int32_t* stats,
                                    int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host);
/* the same inside the wavefront- / workgroup-per-member TR-BDF2 (method 1) / ESDIRK34 (method 2) */
int dsh_sdirk_solve_wave_member_steps(dsh_ctx* ctx, int model, int64_t size, int method, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                                      double t0, double h0, const dsh_adaptive_options* opts, double t_final, int64_t max_cols, double* y_out, double* t_out,
                                      int32_t* stats, int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host);
int dsh_bdf_solve_wave_member(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                              double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats,
                              int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host);
/* dsh_bdf_solve_wave_member with FORWARD SENSITIVITIES of every parameter (problem.bdf_sens(), bdf.rs:370-432, :934-989) for dense run-time-compiled ODE models the
 * register-resident and the banded lane forms do not cover: n <= 140, at most 16 parameters, no mass matrix, no root functions (dsh_model_has_wave_member_sens: 1 for
 * n <= 64 — one wavefront per member — 2 for 64 < n <= 320 — one workgroup per member; BDF and the SDIRK methods).  A component per lane; sens_out: n_eval x np x n x nb (device, batch-fastest); sens_atol: one value (nsens_atol = 0: the sensitivities
 * stay out of the error test).  Other arguments as dsh_bdf_solve_adaptive_sens. */
int dsh_model_has_wave_member_sens(int model, int64_t size);
/* 1: the wavefront-per-member kernels (dsh_bdf_solve_wave_member, dsh_sdirk_solve_wave_member) carry this HYBRID model through all its events inside the launch —
 * reset applied at every root, then on to the last save point (solve_dense with a reset operator, method.rs:774-797); t_root / root_idx report a member's LAST event.
 * Run-time-compiled models with reset_i, stop_i and no mass matrix; returns 1 for n <= 64 (a wavefront per member), 2 for 64 < n <= 320 (a
 * workgroup per member); BDF, TR-BDF2, ESDIRK34. */
int dsh_model_has_wave_member_reset(int model, int64_t size);
/* dsh_sdirk_solve_wave_member with forward sensitivities (problem.tr_bdf2_sens() / esdirk34_sens(); runge_kutta.rs:691-748) for the models of
 * dsh_model_has_wave_member_sens; arguments as dsh_sdirk_solve_resident_sens. */
int dsh_sdirk_solve_wave_member_sens(dsh_ctx* ctx, int model, int64_t size, int method, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                                     double t0, double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double sens_rtol,
                                     const double* sens_atol_host, int64_t nsens_atol, double* y_out, double* sens_out, int32_t* stats, int32_t* status, int64_t* totals_host);
int dsh_bdf_solve_wave_member_sens(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                                   double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double sens_rtol, const double* sens_atol_host,
                                   int64_t nsens_atol, double* y_out, double* sens_out, int32_t* stats, int32_t* status, int64_t* totals_host);
int dsh_model_has_wave_member_sdirk(int model, int64_t size);
int dsh_sdirk_solve_wave_member(dsh_ctx* ctx, int model, int64_t size, int method, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                                double t0, double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats,
                                int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host);
int dsh_model_has_resident(int method, int model, int64_t size);
int dsh_sdirk_solve_resident(dsh_ctx* ctx, int method, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                             double t0, double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats,
                             int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host);


/* ------------------------------------------------------------------------------------------------------------------------------------------------
 * Multi-GPU trajectory collection (BASELINE configs[3]: "sharded over 8 MI355X with RCCL gather").  One process per GPU; rank r integrates the
 * members [lo, hi) = dsh_dist_shard_bounds(n_total, r, world) with no collective inside the integration; the solve_dense output ([lead][nb_local] on the device,
 * batch-fastest, lead = save points x states) is gathered along the batch axis with ONE ncclAllGather (RCCL over xGMI) into [lead][n_total] on every rank.
 * The reference has no distributed layer; a Rust caller would hold a dsh_dist next to its HipContext.  librccl is bound at run time (dlopen; an
 * RCCL already in the process, e.g. PyTorch's, is reused; DSH_RCCL_LIB names one explicitly): DSH_E_UNSUPPORTED when there is none.
 *   dsh_dist_unique_id   rank 0 makes the 128-byte id and ships it to the other ranks by whatever the launcher offers (file, environment, MPI, a TCP store)
 *   dsh_dist_init        collective over all ranks (ncclCommInitRank); the communicator runs on its own stream of ctx's device
 *   dsh_gather_batch_axis[_async] / dsh_gather_wait   the gather; the async form returns at once — the transfer waits for what the solver's stream holds at the time
 *                        of the call and overlaps whatever is enqueued there afterwards (the next solve, into another buffer) — `local` and `out` belong to the
 *                        gather until dsh_gather_wait returns; one gather in flight per communicator
 *   dsh_dist_pack_shard / dsh_dist_unpack_gathered    the two layout copies on their own: [lead][nb_local] -> [lead][m] zero-padded (m = ceil(n_total / world));
 *                        [world][lead][m] -> [lead][n_total] */
#define DSH_DIST_ID_BYTES 128
typedef struct dsh_dist dsh_dist;
int dsh_dist_shard_bounds(int64_t n_total, int rank, int world, int64_t* lo, int64_t* hi);
int dsh_dist_unique_id(unsigned char* id128);
int dsh_dist_init(dsh_ctx* ctx, int rank, int world, const unsigned char* id128, dsh_dist** out);
void dsh_dist_destroy(dsh_dist* d);
int dsh_dist_rank(const dsh_dist* d);
int dsh_dist_world(const dsh_dist* d);
int dsh_gather_batch_axis(dsh_dist* d, const double* local, int64_t lead, int64_t n_total, double* out);
int dsh_gather_batch_axis_async(dsh_dist* d, const double* local, int64_t lead, int64_t n_total, double* out);
int dsh_gather_wait(dsh_dist* d);
int dsh_dist_pack_shard(dsh_ctx* ctx, void* stream, const double* local, int64_t lead, int64_t nb_local, int64_t m, double* send);
int dsh_dist_unpack_gathered(dsh_ctx* ctx, void* stream, const double* recv, int64_t lead, int64_t n_total, int world, double* out);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSOL_HIP_H */
