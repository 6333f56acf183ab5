/*
 * ipm_hip.h -- C ABI of libipm_hip.so: the MI355X (gfx950) Newton/KKT hot path of a
 * Mehrotra predictor-corrector interior-point LP solver.
 *
 * Drop-in boundary for payakorn/InteriorPointMethod.  The reference has no FFI;
 * its seams are plain Python calls.  Each entry point below names
 * the reference code it replaces (paths relative to the reference repo root):
 *
 *   solver seam     interior_sparse   main.py:760-815   /  interior  main.py:707-757
 *   direction seam  direction_predicted_sparse(method="normal")  main.py:197,221-229
 *                   direction_corrected_sparse                  main.py:247-269
 *   linear seam     solve_linear                                 main.py:176-182
 *
 * Conventions: plain pointers and sizes only (no torch types).  All arithmetic is
 * IEEE fp64.  Every function returns an int status (IPM_OK == 0, < 0 error) and
 * never throws or aborts.  Buffers passed in are owned by the caller; device
 * scratch lives in a workspace that is either supplied by the caller (a device
 * pointer, e.g. the data_ptr() of a torch.uint8 tensor) or hipMalloc'ed by the
 * library when the caller passes NULL.  A handle is bound to one HIP device and
 * one stream, is not thread-safe; handles on distinct devices are independent.
 */
#ifndef IPM_HIP_H
#define IPM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPM_ABI_VERSION 4

/* return codes */
enum {
    IPM_OK = 0,
    IPM_ERR_INVALID_ARG = -1,
    IPM_ERR_HIP = -2,          /* a HIP runtime call failed; see ipm_last_error() */
    IPM_ERR_NO_DEVICE = -3,
    IPM_ERR_WORKSPACE = -4,    /* caller workspace too small / misaligned */
    IPM_ERR_STATE = -5,        /* call order violated (e.g. solve before set_A) */
    IPM_ERR_INVALID_INPUT = -6 /* non-finite entries in A, b or c (8 Netlib files) */
};

/* solver status written to ipm_stats.status (reference semantics in comments) */
enum {
    IPM_STATUS_RUNNING = 0,    /* stop test still true                     main.py:780 */
    IPM_STATUS_CONVERGED = 1,  /* check_optimality() returned False        main.py:162-173 */
    IPM_STATUS_MAX_ITER = 2,   /* k reached the cap (5000/50000)           main.py:725,780 */
    IPM_STATUS_NAN = 3         /* non-finite iterate / residual            main.py:1141-1148 */
};

/* ipm_options.flags */
enum {
    /* The Cholesky look-ahead hands work between its two streams through device counters polled inside kernels.
     * That is only safe while the handle's streams own their hardware queues: when several handles are driven
     * concurrently on one GPU (more streams than hardware queues), a polling kernel can sit in front of its own
     * producer in a shared queue.  The library protects itself: it counts the live handles per device and uses
     * stream events whenever more than one exists, and a poll that still times out (bounded spin) is not an error --
     * the affected call is rolled back and repeated with events, and the handle keeps events from then on
     * (ipm_get_schedule reports it).  This flag forces events from the start (slower per step, never polls). */
    IPM_FLAG_NO_DEVICE_POLLING = 1,
    /* ipm_solve switches the 1e-14 Tikhonov shift on by itself when the FIRST factorization of a solve had to guard
     * more than 5 % of its pivots (A has that many dependent rows: the QAP family of Netlib, 9-16 %; no other Netlib
     * file exceeds 2.7 %) and restarts the solve from its start state; the guard alone stalls there.
     * Reported in ipm_stats.auto_regularized.  This flag keeps the shift off. */
    IPM_FLAG_NO_AUTO_REGULARIZE = 2,
    /* One stream per handle: the blocked Cholesky runs without its look-ahead on a second stream (10-25 % slower for
     * a solve that has the GPU to itself).  For SEVERAL handles driven concurrently on one GPU this is the faster
     * setting by far: the HIP runtime maps streams onto four hardware queues, a mid-size LP keeps only a few CUs busy,
     * and four single-stream solves overlap almost perfectly (4 x DEGEN3: 1.15x the time of one) where two-stream
     * handles share queues and serialise (2.4x).  The batched mode of the Python host sets it. */
    IPM_FLAG_SINGLE_STREAM = 4,
    /* Sparse handles only: factor A D^2 A^T with the multifrontal SPARSE Cholesky (csrc/sparse_chol.h) instead of the
     * blocked dense one.  What scipy's spsolve does for the reference (SuperLU: fill-reducing order, symbolic
     * analysis, supernodal numeric factor; main.py:180, :226).  ipm_set_A_csc then analyses the pattern of A A^T in the
     * row order it is given -- pass the rows in the order ipm_order_rows returns -- and every factorization, forward and
     * backward substitution is ONE launch that walks the elimination tree.  It pays when the factor stays sparse
     * (STOCFOR3: 2.2e5 entries, tree height 36, against a chain of 131 dense 128-row blocks); ipm_order_rows reports
     * the numbers to decide with.  ipm_set_A_csc fails with IPM_ERR_INVALID_ARG when the structures exceed its caps. */
    IPM_FLAG_SPARSE_FACTOR = 8,
    /* The handle is created for ipm_solve_batch (the LOCKSTEP batch: iteration k of several LPs in the same launches): implies
     * IPM_FLAG_SINGLE_STREAM | IPM_FLAG_NO_DEVICE_POLLING and block-step triangular solves, so that every launch of its iteration has
     * a batched twin.  Such a handle still works with every other entry point (ipm_solve included: same arithmetic, one LP). */
    IPM_FLAG_LOCKSTEP = 16
};

typedef struct ipm_handle ipm_handle;

typedef struct ipm_options {
    double eta;              /* step damping, reference constant 0.91        main.py:607 */
    double pivot_guard_eps;  /* pivot <= eps*max diag(B) -> pivot_guard_big   */
    double pivot_guard_big;  /* replacement pivot, default 1e64 */
    int32_t check_every;     /* iterations enqueued between host status reads (>=1) */
    int32_t flags;           /* IPM_FLAG_*; 0 = defaults */
    int64_t sparse_nnz;      /* > 0: handle keeps A sparse (CSR+CSC on the device, no dense image) with at
                                most this many nonzeros; only ipm_set_A_csc may then supply A */
    double regularize;       /* Tikhonov shift: factor B + regularize*max diag(B)*I (0 = off, the default;
                                1e-12 makes the rank-deficient QAP family converge) */
} ipm_options;

/* per-solve statistics; norms use the reference's scaling (main.py:170-171) */
typedef struct ipm_stats {
    int32_t status;          /* IPM_STATUS_* */
    int32_t iterations;      /* k: completed predictor-corrector steps */
    int32_t pivots_fixed;    /* total Cholesky pivots replaced by the guard */
    int32_t auto_regularized; /* 1: ipm_solve switched the Tikhonov shift on (IPM_FLAG_NO_AUTO_REGULARIZE) */
    double objective;        /* c^T x                                        main.py:815 */
    double rp_norm;          /* ||Ax-b||_2 */
    double rd_norm;          /* ||A^T y + s - c||_2 */
    double gap;              /* x^T s */
    double b_norm, c_norm;   /* ||b||_2, ||c||_2 */
    double mu, mu_aff, sigma;            /* last iteration           main.py:588-601 */
    double alpha_aff_p, alpha_aff_d;     /* predictor ratio tests    main.py:305-322 */
    double alpha_p, alpha_d;             /* damped step lengths      main.py:604-626 */
    double solve_ms;         /* device time of the last ipm_solve/ipm_iterate (HIP events) */
    double objective_last_finite; /* last finite c^T x seen by the stop test: what new_interior_sparse returns
                                     when the iterate goes NaN (main.py:1227-1233) */
} ipm_stats;

/* One record per completed iteration (the reference prints such a line: main.py:808-809, :1186).  objective, norms
 * and gap describe the iterate the iteration STARTED from (they come from its stop test), the step data what it did. */
typedef struct ipm_iter_record {
    int32_t k;               /* 0-based iteration index */
    int32_t pivots_fixed;    /* cumulative guarded pivots after this iteration's factorization */
    double objective, rp_norm, rd_norm, gap, mu, sigma;
    double alpha_aff_p, alpha_aff_d, alpha_p, alpha_d;
} ipm_iter_record;
#define IPM_HISTORY_CAPACITY 1024   /* ring: the most recent records are kept */

/* ---- library ---------------------------------------------------------------------- */
int ipm_abi_version(void);
int ipm_device_count(int* count);
void ipm_default_options(ipm_options* opts);

/* ---- handle ----------------------------------------------------------------------- */
/* Bytes of device workspace a handle for an m x n problem needs. */
int ipm_workspace_bytes(int64_t m, int64_t n, size_t* bytes);
/* Same for a handle created with ipm_options.sparse_nnz = nnz (A kept sparse). */
int ipm_workspace_bytes_csc(int64_t m, int64_t n, int64_t nnz, size_t* bytes);
/* Same from the options the handle will be created with (sparse_nnz, flags): a sparse handle with IPM_FLAG_SPARSE_FACTOR
 * factors with the multifrontal sparse Cholesky (what SuperLU does inside scipy's spsolve, main.py:180 / :226 of the
 * reference) and needs no dense m x m normal matrix in its workspace -- 2.2 GB less at STOCFOR3 (16675 rows).  The dense
 * entry points (ipm_form_normal_matrix, ipm_get_factor, ipm_solve_linear) still work on such a handle: the library
 * allocates the dense buffer itself on their first use.  A workspace sized by the two functions above is always
 * large enough as well. */
int ipm_workspace_bytes_opts(int64_t m, int64_t n, const ipm_options* opts, size_t* bytes);

/* workspace == NULL: the library allocates (and frees in ipm_destroy).
 * stream == NULL: the library creates its own stream on `device`. */
int ipm_create(int device, int64_t m, int64_t n, const ipm_options* opts,
               void* workspace, size_t workspace_bytes, void* stream, ipm_handle** out);
int ipm_destroy(ipm_handle* h);
const char* ipm_last_error(const ipm_handle* h);   /* also valid with h == NULL */

/* ---- problem data (replaces the arguments of interior_sparse, main.py:760) -------- */
/* Row-major m x n fp64 matrix with leading dimension ld (elements). is_device: the
 * pointer is device memory on the handle's device (copied device-to-device). */
int ipm_set_A_dense(ipm_handle* h, const double* A, int64_t ld, int is_device);
/* CSC triplets as scipy.sparse.csc_matrix holds them (sparse_interior.py:215); host memory.
 * Duplicate entries are summed (scipy's constructor semantics).  On a sparse handle A stays sparse
 * (SpMV + sparse formation of A D^2 A^T); on a dense handle it is scattered into the dense image.
 * A sparse handle also derives the 128-row-block ENVELOPE of A A^T from the structure of A in the row order
 * given: blocks below it are structurally zero in B and stay zero in the factor, and the blocked Cholesky
 * and the triangular solves skip them (exact; environment IPM_ENVELOPE=0 disables).  The row order that makes
 * the envelope small is the caller's choice -- the Python host applies reverse Cuthill-McKee to the pattern of
 * A A^T when that pays (SuperLU behind the reference's spsolve reorders too, main.py:180). */
int ipm_set_A_csc(ipm_handle* h, const int32_t* colptr, const int32_t* rowind,
                  const double* val, int64_t nnz);
int ipm_set_bc(ipm_handle* h, const double* b, const double* c);        /* host, length m / n */

/* ---- iterate (x, y, s) ------------------------------------------------------------- */
int ipm_set_state(ipm_handle* h, const double* x, const double* y, const double* s);  /* host */
int ipm_get_state(ipm_handle* h, double* x, double* y, double* s);                    /* host */
/* x = s = 1, y = y0: sparse_interior.py:193-200 (y0=1) / main.py:287-302 (y0=0) */
int ipm_init_state(ipm_handle* h, double y0);

/* ---- direction seam (main.py:197 / :247) ------------------------------------------- */
/* corrector == 0: predictor direction at the current state (forms and factors A D^2 A^T).
 * corrector == 1: corrector direction; requires a preceding predictor call at the same
 * state (reuses its factor and affine direction).  Outputs are host arrays (may be NULL). */
int ipm_newton_direction(ipm_handle* h, int corrector, double* dx, double* dy, double* ds,
                         ipm_stats* stats);

/* ---- solver seam ------------------------------------------------------------------- */
/* Run exactly n_steps predictor-corrector iterations from the current state.  The stop
 * test is evaluated (stats) but not acted on: this is the benchmark entry point.  The iteration count (and the
 * history) restarts at 0 with the first call after ipm_set_state / ipm_init_state and continues over further calls.  The residuals and the stop test
 * are evaluated once more after the last step, so objective / norms / gap in `stats` describe the state
 * ipm_get_state returns (step lengths, mu_aff and sigma are those of the last step taken). */
int ipm_iterate(ipm_handle* h, int32_t n_steps, ipm_stats* stats);
/* Loop of interior_sparse: stop test first, then one iteration, until the test fails or
 * max_iter iterations were taken.  tol_p/tol_d/tol_gap = e1/e2/e3 of main.py:772-774. */
int ipm_solve(ipm_handle* h, double tol_p, double tol_d, double tol_gap, int32_t max_iter,
              ipm_stats* stats);
/* The LOCKSTEP BATCH of the batched-LP mode (the driver loop of script.py:147-173 over independent LPs, on one GPU): ipm_solve for n
 * handles AT ONCE, iteration k of all of them in the same launches (csrc/lockstep.h).  Every handle must have been created with
 * IPM_FLAG_LOCKSTEP on the same device, hold a sparse A of more than 128 rows on the dense-tile factor (not IPM_FLAG_SPARSE_FACTOR)
 * and have A, b, c and a state set.  Same arguments and per-handle semantics as ipm_solve (stop test, iteration cap, automatic
 * Tikhonov shift); stats[i] (may be NULL) describes handle i, with solve_ms the device time of the whole batch.  The arithmetic
 * of a handle is exactly that of ipm_solve on it alone: bit-identical iterates.  Launched on the first handle's stream. */
int ipm_solve_batch(ipm_handle** handles, int32_t n, double tol_p, double tol_d, double tol_gap, int32_t max_iter, ipm_stats* stats);
/* The same batch, incrementally: handles may JOIN between two steps (a host thread finishes an LP's set-up while the batch is already
 * running) and the caller learns which ones finished after every step (and can tear them down while the batch runs on).
 * ipm_batch_add: the handle (requirements as above) starts its solve from its current state, *index = its position in the batch.
 * ipm_batch_step: opt.check_every iterations of every active handle in lockstep; the indices of the handles that finished in this
 * step go to finished[0 .. *n_finished), *n_active = handles still running (0: nothing left to do).  ipm_batch_stats: the statistics
 * of a finished handle.  A batch owns one stream; it is not thread-safe; the handles stay owned by the caller and must outlive their
 * part in the batch (destroy a handle only after ipm_batch_step reported it finished, or after ipm_batch_destroy). */
typedef struct ipm_batch ipm_batch;
int ipm_batch_create(int device, void* stream, ipm_batch** out);   /* stream: a hipStream_t of the caller (kept alive until ipm_batch_destroy), or NULL: the batch creates its own */
int ipm_batch_destroy(ipm_batch* b);
const char* ipm_batch_last_error(const ipm_batch* b);
int ipm_batch_add(ipm_batch* b, ipm_handle* h, double tol_p, double tol_d, double tol_gap, int32_t max_iter, int32_t* index);
int ipm_batch_step(ipm_batch* b, int32_t* finished, int32_t cap, int32_t* n_finished, int32_t* n_active);
int ipm_batch_stats(ipm_batch* b, int32_t index, ipm_stats* stats);
/* Per-iteration records of the last ipm_solve / ipm_iterate, oldest first: min(iterations, IPM_HISTORY_CAPACITY,
 * capacity) records are written to `out` (host) and their number to *count. */
int ipm_get_history(ipm_handle* h, ipm_iter_record* out, int32_t capacity, int32_t* count);
/* How the handle schedules its factorization (tests and diagnostics): out[0] = 128-row blocks, out[1] = group size
 * of the two-level Cholesky (1 = one-level), out[2] = 1 when the 1024-row grouped triangular solves are used,
 * out[3] = 1 while cross-stream hand-offs poll device counters (0: stream events), out[4] / out[5] = bulk trailing
 * updates of the last factorization that signalled a counter / recorded an event, out[6] = 1 when the tile envelope
 * of a sparse handle is exploited, out[7] = live handles on this device, out[8] = poll time-outs recovered so far,
 * out[9] = 1 when the fused single-workgroup small-LP path serves this handle, out[10] = 1 when the last iteration ran the
 * FUSED formation + factorization (one persistent launch beside the pivot chain; dense handles of 20 .. 40 blocks with
 * n <= 3 m that have the device to themselves), out[11] = 1 while a sparse-factor handle runs one launch per level of its panel tree
 * (shared device) instead of one launch per sweep.  (ABI 4: twelve words; ABI 3 had ten.) */
int ipm_get_schedule(ipm_handle* h, int32_t out[12]);

/* Fill-reducing order of the ROWS of an m x n sparse A (CSC, host) for the Cholesky of A D^2 A^T: minimum degree on
 * the pattern of A A^T followed by the elimination-tree postorder.  Pure host code (no device is touched): the
 * reference gets the same service from SuperLU's COLAMD inside spsolve (main.py:180).  perm[new] = old, length m.
 * info (may be NULL, 8 doubles): [0] entries of the strict lower triangle of A A^T, [1] entries of the Cholesky factor in
 * this order (diagonal included), [2] multiply-adds of that factorization (sum over columns of count^2), [3] height of the
 * elimination tree in columns; of the panel tree the device would walk (the analysis ipm_set_A_csc runs): [4] its height in
 * panels, [5] the largest sum of (front rows)^2 along a root-to-leaf path, [6] panels, [7] rows of the widest front
 * ([4..7] zero when that analysis exceeds its caps).  Returns IPM_OK; IPM_ERR_WORKSPACE when the pattern or the ordering work exceeds the
 * built-in caps (A A^T close to dense: keep the dense path), perm is then the identity.
 * info[0] ON INPUT, honoured only together with info[1] = -1.0 (so that an uninitialised array cannot switch it on): the
 * milliseconds per iteration the caller expects from its alternative, the dense-tile path.  The elimination then gives up early (IPM_ERR_WORKSPACE) once a pivot's degree shows that the sparse factor cannot
 * beat that, and works within a budget scaled to it -- a caller that only wants the order when it pays (factor "auto" of the
 * Python host) saves 40 % of the host time the hopeless cases cost. */
int ipm_order_rows(int64_t m, int64_t n, const int32_t* colptr, const int32_t* rowind, int32_t* perm, double info[8]);
/* Structure of the sparse factor of a handle created with IPM_FLAG_SPARSE_FACTOR (IPM_ERR_STATE otherwise):
 * out[0] panels, [1] tasks, [2] panel-tree height, [3] widest front (rows), [4] entries of L stored, [5] entries of the
 * update matrices, [6] product-list terms of the formation, [7] launches that fell back to one workgroup after a
 * hand-off time-out. */
int ipm_get_factor_info(ipm_handle* h, int64_t out[8]);

/* ---- linear-solve seam (main.py:176-182) and kernel-level entry points ------------- */
/* Solve B z = rhs for a dense SYMMETRIC POSITIVE (SEMI)DEFINITE m x m host matrix by the blocked guarded Cholesky
 * (m = the handle's m; only the lower triangle is read).  The reference's solve_linear takes any square matrix
 * (LU); this seam is the normal-equations one, where the matrix is A D^2 A^T -- a general matrix is not accepted.
 * z may alias rhs.  pivots_fixed may be NULL. */
int ipm_solve_linear(ipm_handle* h, const double* B, int64_t ldb, const double* rhs, double* z,
                     int32_t* pivots_fixed);
/* Solve (A diag(d) A^T) z = rhs with the handle's own A: forms the normal matrix on the device (d on the host,
 * length n; NULL = all ones), factors it with the guarded blocked Cholesky (reuse_factor != 0: the factor of the
 * previous ipm_normal_solve / direction call is kept) and back-substitutes.  rhs, z: host, length m; z may alias
 * rhs.  Building block of start-point heuristics (x = A^T (A A^T)^-1 b, ...); the iterate is not touched. */
int ipm_normal_solve(ipm_handle* h, const double* d, const double* rhs, double* z, int reuse_factor,
                     int32_t* pivots_fixed);
/* B = A diag(d) A^T (d on the host, length n); full symmetric m x m written to host B. */
int ipm_form_normal_matrix(ipm_handle* h, const double* d, double* B, int64_t ldb);
/* Cholesky factor of the handle's current normal matrix; lower triangle to host L. */
int ipm_get_factor(ipm_handle* h, double* L, int64_t ldl);
/* Timing of the device phases of the last ipm_iterate call, milliseconds per iteration:
 * out[0]=form A D^2 A^T, out[1]=factor, out[2]=triangular solves, out[3]=everything else.
 * enable = 1: only out[0] (two event records per iteration around the dominant kernel); enable = 2: all four
 * (nine records per iteration, ~1 % slower); 0: off. */
int ipm_set_profiling(ipm_handle* h, int enable);
/* Diagnostic builds only (environment IPM_POTRF_STAMPS=1 at ipm_create): s_memtime stamps of the first
 * diagonal-block factorization, 8 waves x 64 slots.  IPM_ERR_STATE otherwise. */
int ipm_debug_get_stamps(ipm_handle* h, long long* out);
/* Host only (no device is touched; tests): the ordered work list of the fused formation + factorization for `nblk` 128-row
 * blocks, `q` formation chunks per tile and `workers` workgroups (csrc/ff_schedule.h).  Up to `capacity` items of 8 bytes
 * {type, i, c, q, j0, j1, flags, seq} are written to `items`, their number to *count, the number of update items per lower
 * tile (row-major triangle, nblk (nblk + 1) / 2 entries) to tile_items, and the simulated {end of the factorization, end
 * of the formation} in microseconds to sim_us. */
int ipm_debug_ff_schedule(int32_t nblk, int32_t q, int32_t workers, unsigned char* items, int32_t capacity, int32_t* count,
                          int32_t* tile_items, double sim_us[2]);
int ipm_get_phase_ms(ipm_handle* h, double out[4]);
/* Test hook: inv(L_kk) of diagonal block k of the handle's current dense factor, 128 x 128 row-major (zeros above the diagonal), to
 * host `out`.  Rows of the block beyond the LP's row count are padding: the identity (tests/test_gpu_parity.py). */
int ipm_debug_get_block_inverse(ipm_handle* h, int32_t k, double* out);
/* Diagnostic (environment IPM_FF_TRACE_ITEMS=1 at ipm_create; IPM_ERR_STATE otherwise): time line of the LAST fused formation +
 * factorization launch on the device-wide 100 MHz clock.  *count = words of the trace: 4 per work item {drawn, inputs ready,
 * done, worker} followed by 12 per 128-row block {potrf_diag: start, inputs ready, done, -; critical panel: same; critical
 * update: same}; out (capacity words) receives it when large enough; items (8 bytes each, may be NULL) the work list,
 * *nitems its length.  tools/ff_trace.py turns it into per-step stall tables. */
int ipm_debug_ff_trace(ipm_handle* h, long long* out, int64_t capacity, int64_t* count, unsigned char* items, int32_t* nitems);
/* Test hook (host only, no GPU): the schedule merge of the lockstep batch (csrc/lockstep_merge.h).  n launch programs, program i =
 * types_flat[off[i] .. off[i+1]) (kernel types); aligned != 0: progressive alignment (the product's merge), 0: the leader rule it
 * replaced.  out_steps (2 ints per step, capacity cap_steps steps): {type, member count}; out_members (2 ints per launch): {program,
 * position} in step order.  Returns the number of steps, -1 on bad arguments / too small a capacity. */
int ipm_debug_ls_merge(int32_t n, const int32_t* off, const int32_t* types_flat, int32_t max_group, int32_t aligned,
                       int32_t* out_steps, int32_t cap_steps, int32_t* out_members);

#ifdef __cplusplus
}
#endif
#endif /* IPM_HIP_H */
