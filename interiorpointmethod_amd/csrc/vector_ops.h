// vector_ops.h -- the HBM-bound part of one interior-point iteration (gfx950): dense GEMV
// passes over A and the fused elementwise/reduction kernels between them.
//
// Reference code these kernels restate (paths in the reference repo):
//   residuals r_b, r_c, r3          main.py:66-73   (test_create_rhs_predicted)
//   stop test                       main.py:162-173 (check_optimality)
//   predictor rhs / recovery        main.py:223-228 (direction_predicted_sparse "normal")
//   ratio tests                     main.py:305-322 (predicted_stepsize), :604-626 (full_stepsize)
//   mu, mu_aff, sigma               main.py:588-601 (duality_gap)
//   corrector complementarity rhs   main.py:150-152 (create_rhs_corrected)
//   iterate update                  main.py:694-696 (corrected)
//
// All reductions are two-level with a fixed order (per-block partials, then every consumer
// re-sums the <= 64 partials in index order), so a solve is bitwise reproducible; no fp64
// atomics are used.  Scalars (norms, alpha, sigma, mu, stop flag) never leave the device
// inside an iteration.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_nt_f64.h"

namespace ipm {

constexpr int VBLK = 256;        // threads per vector-kernel block
constexpr int MAXPART = 64;      // max blocks (= partials) of a vector kernel

// device-resident scalar state of a solve
struct Scalars {
    double b_norm, c_norm;
    double rb_norm, rc_norm, gap, obj;
    double mu, mu_aff, sigma;
    double alpha_aff_p, alpha_aff_d, alpha_p, alpha_d;
    double maxdiag;
    double e1, e2, e3, eta;
    double obj_last_finite;      // last finite c^T x seen by the stop test (main.py:1227-1233 returns it on NaN)
    int done, status, k, max_iter, fixed, force, fixed_first;   // fixed_first: guarded pivots of the first factorization
    int done_f;                  // `done` as it stood when the iteration's formation began (scaling_kernel): what the formation and
                                 // factorization kernels of the overlapped path test, so that a stop test that flips `done` while
                                 // they are in flight (it runs on the residual stream) never leaves B half factored
};

// per-iteration record (include/ipm_hip.h: ipm_iter_record), written by update_kernel into a ring
struct IterRec {
    int k, fixed;
    double obj, rb, rc, gap, mu, sigma, aap, aad, ap, ad;
};
constexpr int HIST_CAP = 1024;

// partial-sum slots (each MAXPART doubles)
enum { P_RC2 = 0, P_XS, P_CX, P_RB2, P_MINP_AFF, P_MIND_AFF, P_MUAFF, P_MINP, P_MIND, P_NSLOT };

__device__ __forceinline__ double block_sum(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = VBLK / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_min(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = VBLK / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmin(red[tid], red[tid + s]);   // fmin drops NaN like np.min never sees it
        __syncthreads();
    }
    double r = red[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double sum_partials(const double* part, int slot, int nblk) {
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += part[slot * MAXPART + i];
    return s;
}
__device__ __forceinline__ double min_partials(const double* part, int slot, int nblk) {
    double s = 1.0;                                  // min(np.append(ratios, 1)), main.py:309
    for (int i = 0; i < nblk; ++i) s = fmin(s, part[slot * MAXPART + i]);
    return s;
}

// ---------------------------------------------------------------------------------------
// GEMV, A row-major [mp][np] (zero padded).
// ---------------------------------------------------------------------------------------
// out[i] = sa * (A[i,:] . v) + sb * add[i]   -- one wave per row, 16 B per lane per step.
__device__ __forceinline__ void gemv_n_kernel_body(const double* __restrict__ A, int64_t lda, int mp,
                                                     int np, const double* __restrict__ v, double sa,
                                                     double sb, const double* __restrict__ add,
                                                     double* out, const int* done, const unsigned bx_, const unsigned gx_) {
    if (done && *done) return;
    const int lane = threadIdx.x & 63;
    const int row = bx_ * 4 + (threadIdx.x >> 6);
    if (row >= mp) return;
    const double* ar = A + (int64_t)row * lda;
    double acc0 = 0.0, acc1 = 0.0;
    for (int c = lane * 2; c < np; c += 128) {
        f64x2 a2 = *reinterpret_cast<const f64x2*>(ar + c);
        f64x2 v2 = *reinterpret_cast<const f64x2*>(v + c);
        acc0 += a2.x * v2.x;
        acc1 += a2.y * v2.y;
    }
    double s = acc0 + acc1;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[row] = sa * s + (add ? sb * add[row] : 0.0);
}
__global__ __launch_bounds__(256) void gemv_n_kernel(const double* __restrict__ A, int64_t lda, int mp,
                                                     int np, const double* __restrict__ v, double sa,
                                                     double sb, const double* __restrict__ add,
                                                     double* out, const int* done) { gemv_n_kernel_body(A, lda, mp, np, v, sa, sb, add, out, done, blockIdx.x, gridDim.x); }

// part[rc][c] = sum_{r in chunk rc} A[r][c] * u[r]  -- grid (ceil(np/512), RC); 2 columns/thread.
__device__ __forceinline__ void gemv_t_kernel_body(const double* __restrict__ A, int64_t lda, int rows_per_chunk,
                                                     int np, const double* __restrict__ u, double* part,
                                                     const int* done, const unsigned bx_, const unsigned by_) {
    if (done && *done) return;
    const int c = (bx_ * 256 + threadIdx.x) * 2;
    if (c >= np) return;
    const int r0 = by_ * rows_per_chunk;
    double a0 = 0.0, a1 = 0.0;
    const double* ap = A + (int64_t)r0 * lda + c;
#pragma unroll 8
    for (int r = 0; r < rows_per_chunk; ++r) {
        f64x2 a2 = *reinterpret_cast<const f64x2*>(ap + (int64_t)r * lda);
        double ur = u[r0 + r];
        a0 += a2.x * ur;
        a1 += a2.y * ur;
    }
    *reinterpret_cast<f64x2*>(part + (int64_t)by_ * np + c) = (f64x2){a0, a1};
}
__global__ __launch_bounds__(256) void gemv_t_kernel(const double* __restrict__ A, int64_t lda, int rows_per_chunk,
                                                     int np, const double* __restrict__ u, double* part,
                                                     const int* done) { gemv_t_kernel_body(A, lda, rows_per_chunk, np, u, part, done, blockIdx.x, blockIdx.y); }

struct VecArgs {
    int m, n, np, rc_chunks;       // true sizes, padded n, number of gemv_t row chunks
    int nblk;                      // blocks of the vector kernels (<= MAXPART)
    const double* atp;             // gemv_t partials [rc_chunks][np]
    double *x, *y, *s;
    const double *b, *c;
    double *rb, *rc, *d, *v, *q;
    double *dxa, *dya, *dsa, *dx, *dy, *ds;
    double* part;                  // [P_NSLOT][MAXPART]
    Scalars* sc;
    IterRec* hist;                 // [HIST_CAP] ring of per-iteration records
};

// sum over the row chunks of the GEMV-T partials, in chunk order (fixed order: bitwise reproducible).  Eight loads are in
// flight at a time: a plain loop waits for every load before it issues the next one (32 chunks = 32 L2 latencies, 13 us of
// direction_kernel's 13 us at n = 8192).
__device__ __forceinline__ double col_sum(const double* atp, int rc_chunks, int np, int j) {
    double s = 0.0;
    int r = 0;
    for (; r + 8 <= rc_chunks; r += 8) {
        double t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = atp[(int64_t)(r + q) * np + j];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += t[q];
    }
    for (; r < rc_chunks; ++r) s += atp[(int64_t)r * np + j];
    return s;
}

// r_c = A^T y + s - c ; d = x/s ; predictor v = d*(r_c - r3/x) ; partial ||r_c||^2, x.s, c.x, ||r_b||^2
__device__ __forceinline__ void prepare_kernel_body(VecArgs a, const unsigned bx_, const unsigned gx_) {
    __shared__ double red[VBLK];
    const int gid = bx_ * VBLK + threadIdx.x, gsz = gx_ * VBLK;
    double rc2 = 0.0, xs = 0.0, cx = 0.0, rb2 = 0.0;
    for (int j = gid; j < a.n; j += gsz) {
        double xj = a.x[j], sj = a.s[j];
        double rcj = col_sum(a.atp, a.rc_chunks, a.np, j) + sj - a.c[j];
        double dj = xj / sj;
        double r3 = xj * sj;
        a.rc[j] = rcj;
        a.d[j] = dj;
        a.q[j] = r3 / xj;
        a.v[j] = dj * (rcj - r3 / xj);
        rc2 += rcj * rcj;
        xs += r3;
        cx += a.c[j] * xj;
    }
    for (int i = gid; i < a.m; i += gsz) { double r = a.rb[i]; rb2 += r * r; }
    rc2 = block_sum(rc2, red); xs = block_sum(xs, red); cx = block_sum(cx, red); rb2 = block_sum(rb2, red);
    if (threadIdx.x == 0) {
        a.part[P_RC2 * MAXPART + bx_] = rc2;
        a.part[P_XS * MAXPART + bx_] = xs;
        a.part[P_CX * MAXPART + bx_] = cx;
        a.part[P_RB2 * MAXPART + bx_] = rb2;
    }
}
__global__ __launch_bounds__(VBLK) void prepare_kernel(VecArgs a) { prepare_kernel_body(a, blockIdx.x, gridDim.x); }

// d = x / s only (main.py:223): what the formation of A D^2 A^T needs; the full prepare_kernel follows on the residual
// stream while the factorization runs (ipm_api.hip, enqueue_iteration)
__global__ __launch_bounds__(VBLK) void scaling_kernel(VecArgs a) {
    if (blockIdx.x == 0 && threadIdx.x == 0) a.sc->done_f = a.sc->done;      // latch for this iteration's factorization
    if (a.sc->done) return;
    const int gid = blockIdx.x * VBLK + threadIdx.x, gsz = gridDim.x * VBLK;
    for (int j = gid; j < a.n; j += gsz) a.d[j] = a.x[j] / a.s[j];
}

// stop test of check_optimality (main.py:162-173) -- one thread.
__device__ __forceinline__ void stop_test_kernel_body(VecArgs a, const unsigned bx_, const unsigned gx_) {
    if (threadIdx.x != 0 || bx_ != 0) return;
    Scalars* sc = a.sc;
    if (sc->done) return;
    double rb = sqrt(sum_partials(a.part, P_RB2, a.nblk));
    double rc = sqrt(sum_partials(a.part, P_RC2, a.nblk));
    double gap = sum_partials(a.part, P_XS, a.nblk);
    sc->rb_norm = rb; sc->rc_norm = rc; sc->gap = gap;
    const double obj = sum_partials(a.part, P_CX, a.nblk);
    sc->obj = obj;
    if (fabs(obj) < 1.7e308) sc->obj_last_finite = obj;          // false for NaN and Inf
    sc->mu = gap / (double)a.n;
    bool cont = (sc->e1 * (1.0 + sc->b_norm) < rb) || (sc->e2 * (1.0 + sc->c_norm) < rc) || (sc->e3 < gap);
    if (sc->force) return;
    if (!cont) {
        bool finite = (rb == rb) && (rc == rc) && (gap == gap) && (fabs(rb) < 1.7e308) && (fabs(rc) < 1.7e308) &&
                      (fabs(gap) < 1.7e308);
        sc->status = finite ? 1 : 3;
        sc->done = 1;
    } else if (sc->k >= sc->max_iter) {
        sc->status = 2;
        sc->done = 1;
    }
}
__global__ void stop_test_kernel(VecArgs a) { stop_test_kernel_body(a, blockIdx.x, gridDim.x); }

// direction recovery + ratio test.  corr == 0: (dxa, dsa) from dya with q = r3/x;
// corr == 1: (dx, ds) from dy with the corrector q.
__device__ __forceinline__ void direction_kernel_body(VecArgs a, int corr, const unsigned bx_, const unsigned gx_) {
    if (a.sc->done) return;
    __shared__ double red[VBLK];
    const int gid = bx_ * VBLK + threadIdx.x, gsz = gx_ * VBLK;
    double* DX = corr ? a.dx : a.dxa;
    double* DS = corr ? a.ds : a.dsa;
    double mp_ = 1.0, md_ = 1.0;
    for (int j = gid; j < a.n; j += gsz) {
        double xj = a.x[j], sj = a.s[j];
        double w = col_sum(a.atp, a.rc_chunks, a.np, j);
        double dxj = a.d[j] * w + a.v[j];                 // main.py:227
        double dsj = (-sj * dxj) / xj - a.q[j];           // main.py:228
        DX[j] = dxj; DS[j] = dsj;
        if (dxj < 0.0) mp_ = fmin(mp_, -xj / dxj);
        if (dsj < 0.0) md_ = fmin(md_, -sj / dsj);
    }
    mp_ = block_min(mp_, red); md_ = block_min(md_, red);
    if (threadIdx.x == 0) {
        a.part[(corr ? P_MINP : P_MINP_AFF) * MAXPART + bx_] = mp_;
        a.part[(corr ? P_MIND : P_MIND_AFF) * MAXPART + bx_] = md_;
    }
}
__global__ __launch_bounds__(VBLK) void direction_kernel(VecArgs a, int corr) { direction_kernel_body(a, corr, blockIdx.x, gridDim.x); }

// partial sums of (x + a_p dxa).(s + a_d dsa)      main.py:579-584, 598
__device__ __forceinline__ void mu_aff_kernel_body(VecArgs a, const unsigned bx_, const unsigned gx_) {
    if (a.sc->done) return;
    __shared__ double red[VBLK];
    const int gid = bx_ * VBLK + threadIdx.x, gsz = gx_ * VBLK;
    const double ap = min_partials(a.part, P_MINP_AFF, a.nblk);
    const double ad = min_partials(a.part, P_MIND_AFF, a.nblk);
    double acc = 0.0;
    for (int j = gid; j < a.n; j += gsz) acc += (a.x[j] + ap * a.dxa[j]) * (a.s[j] + ad * a.dsa[j]);
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) {
        a.part[P_MUAFF * MAXPART + bx_] = acc;
        if (bx_ == 0) { a.sc->alpha_aff_p = ap; a.sc->alpha_aff_d = ad; }
    }
}
__global__ __launch_bounds__(VBLK) void mu_aff_kernel(VecArgs a) { mu_aff_kernel_body(a, blockIdx.x, gridDim.x); }

// corrector: r3c = x s + dxa dsa - sigma mu ; q = r3c/x ; v = d (r_c - q)     main.py:150-152
__device__ __forceinline__ void corrector_rhs_kernel_body(VecArgs a, const unsigned bx_, const unsigned gx_) {
    if (a.sc->done) return;
    const int gid = bx_ * VBLK + threadIdx.x, gsz = gx_ * VBLK;
    const double mu = a.sc->mu;
    const double mu_aff = sum_partials(a.part, P_MUAFF, a.nblk) / (double)a.n;
    const double r = mu_aff / mu;
    const double sigma = r * r * r;
    const double sm = sigma * mu;
    for (int j = gid; j < a.n; j += gsz) {
        double xj = a.x[j];
        double r3c = xj * a.s[j] + a.dxa[j] * a.dsa[j] - sm;
        double qj = r3c / xj;
        a.q[j] = qj;
        a.v[j] = a.d[j] * (a.rc[j] - qj);
    }
    if (gid == 0) { a.sc->mu_aff = mu_aff; a.sc->sigma = sigma; }
}
__global__ __launch_bounds__(VBLK) void corrector_rhs_kernel(VecArgs a) { corrector_rhs_kernel_body(a, blockIdx.x, gridDim.x); }

// x += a_p dx ; y += a_d dy ; s += a_d ds ; k += 1          main.py:604-626, 694-696
__device__ __forceinline__ void update_kernel_body(VecArgs a, const unsigned bx_, const unsigned gx_) {
    if (a.sc->done) return;
    const int gid = bx_ * VBLK + threadIdx.x, gsz = gx_ * VBLK;
    const double eta = a.sc->eta;
    const double ap = fmin(1.0, eta * min_partials(a.part, P_MINP, a.nblk));
    const double ad = fmin(1.0, eta * min_partials(a.part, P_MIND, a.nblk));
    for (int j = gid; j < a.n; j += gsz) {
        a.x[j] += ap * a.dx[j];
        a.s[j] += ad * a.ds[j];
    }
    for (int i = gid; i < a.m; i += gsz) a.y[i] += ad * a.dy[i];
    if (gid == 0) {
        Scalars* sc = a.sc;
        const int k = sc->k;
        if (k == 0) sc->fixed_first = sc->fixed;
        IterRec r;
        r.k = k; r.fixed = sc->fixed; r.obj = sc->obj; r.rb = sc->rb_norm; r.rc = sc->rc_norm; r.gap = sc->gap;
        r.mu = sc->mu; r.sigma = sc->sigma; r.aap = sc->alpha_aff_p; r.aad = sc->alpha_aff_d; r.ap = ap; r.ad = ad;
        a.hist[k % HIST_CAP] = r;
        sc->alpha_p = ap; sc->alpha_d = ad; sc->k = k + 1;
    }
}
__global__ __launch_bounds__(VBLK) void update_kernel(VecArgs a) { update_kernel_body(a, blockIdx.x, gridDim.x); }

// out[i] = value for i < n (fill)
__global__ void fill_kernel(double* out, int n, double value) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = value;
}

// ||v||_2 of a short vector -> *out (single block)
__global__ __launch_bounds__(VBLK) void norm2_kernel(const double* v, int n, double* out) {
    __shared__ double red[VBLK];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += VBLK) acc += v[i] * v[i];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) *out = sqrt(acc);
}

}  // namespace ipm
