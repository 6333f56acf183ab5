// small_lp.h -- the whole Mehrotra predictor-corrector LOOP of a small LP (m <= 128) in ONE launch of ONE
// workgroup (gfx950).  Replaces, for the AFIRO class of the Netlib set, the loop body of interior_sparse
// (main.py:780-807 of the reference repo) that the multi-kernel path runs as ~60 launch-bound kernels per
// iteration (AFIRO: 106 us per iteration, 93 iterations): here an iteration never leaves the compute unit.
//
//   * A stays sparse in HBM/L2 (CSR + CSC, as uploaded by ipm_set_A_csc); the n-vectors stay in global memory
//     (they are L2 resident: one workgroup, <= 100 KB); the m-vectors, the normal matrix B (16 nt x 16 nt, nt =
//     ceil(m/16)), its factor L and inv(L) live in LDS for the whole solve.
//   * B = A diag(d) A^T is evaluated from a PRODUCT LIST built once on the host: lower entry e = (i, k) is
//     sum_t coef[t] * d[col[t]] over the columns j that rows i and k share (coef = a_ij a_kj) -- a sparse
//     matrix-vector product with d, one thread per entry, no atomics, fixed order.
//   * The factorization is potrf_lds() -- the same guarded LDS Cholesky the blocked path runs on its 128 x 128
//     diagonal blocks -- on nt panels instead of 8; both triangular solves are matvecs with inv(L) in LDS.
//   * Reductions are wave shuffles + a fixed-order sum over the 8 waves: bitwise reproducible.
//
// Same mathematics, constants and stop test as vector_ops.h; the summation orders differ from the
// multi-kernel path, so iterates agree to rounding, not bit for bit (tests compare both against the reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "potrf_f64.h"
#include "sparse_ops.h"
#include "vector_ops.h"

namespace ipm {

constexpr int SMALL_MAX_M = 128;
constexpr int IPM_STATUS_NEEDS_SHIFT = 4;      // internal: left after the first factorization, ipm_solve restarts

struct SmallLP {
    SparseA A;
    int m, n, nt;
    const int* bptr; const unsigned short* bi; const unsigned short* bk; const int* bcol; const double* bcoef; int nb;
    double *x, *y, *s;
    const double *b, *c;
    double *rc, *d, *v, *q, *dxa, *dsa, *dx, *ds;
    Scalars* sc;
    IterRec* hist;
    double eps, big, shift_rel;
    int max_steps;
    int auto_reg;
};

// sum / min of NV values over the 512 threads, fixed order; result in every thread.  `red` = 8 * NV doubles of LDS.
template <int NV, bool MIN>
__device__ __forceinline__ void block_reduce512(double (&v)[NV], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(v[q], off, 64);
            v[q] = MIN ? fmin(v[q], o) : v[q] + o;
        }
    }
    __syncthreads();                                   // red may still be read from the previous reduction
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) red[wave * NV + q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double r = red[q];
#pragma unroll
        for (int w = 1; w < 8; ++w) r = MIN ? fmin(r, red[w * NV + q]) : r + red[w * NV + q];
        v[q] = r;
    }
}

__global__ __launch_bounds__(PD_THREADS) void small_lp_kernel(SmallLP a) {
    __shared__ __attribute__((aligned(16))) double W[NB * WLD];
    __shared__ double dinv_s[NB];
    __shared__ double ys[NB], rbs[NB], t1s[NB], zs[NB], dys[NB];
    __shared__ double red[8 * 4];
    __shared__ double sh[8];                            // broadcast scalars
    __shared__ int go;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m, n = a.n, nt = a.nt, mt = 16 * a.nt;
    const int row4 = tid >> 2, l4 = tid & 3;            // 4 lanes per row of A / of the triangular matvecs
    Scalars* sc = a.sc;

    if (tid < NB) {
        ys[tid] = tid < m ? a.y[tid] : 0.0;
        rbs[tid] = 0.0; t1s[tid] = 0.0; zs[tid] = 0.0; dys[tid] = 0.0;
    }
    __syncthreads();

    // r_b = A x - b into rbs (4 lanes per row), returns this thread's share of ||r_b||^2
    auto residual_rows = [&]() {
        double rb2 = 0.0;
        if (row4 < m) {
            const int pb = a.A.rowptr[row4], pe = a.A.rowptr[row4 + 1];
            double acc = 0.0;
            for (int p = pb + l4; p < pe; p += 4) acc += a.A.rval[p] * a.x[a.A.colind[p]];
            acc += __shfl_xor(acc, 2, 4);
            acc += __shfl_xor(acc, 1, 4);
            const double r = acc - a.b[row4];
            if (l4 == 0) { rbs[row4] = r; rb2 = r * r; }
        }
        return rb2;
    };
    // r_c, d, q, v and the partial sums of ||r_c||^2, x.s, c.x
    auto residual_cols = [&](double& rc2, double& xs, double& cx) {
        for (int j = tid; j < n; j += PD_THREADS) {
            const int pb = a.A.colptr[j], pe = a.A.colptr[j + 1];
            double w = 0.0;
            for (int p = pb; p < pe; ++p) w += a.A.cval[p] * ys[a.A.rowind[p]];
            const double xj = a.x[j], sj = a.s[j], cj = a.c[j];
            const double rcj = w + sj - cj, dj = xj / sj, r3 = xj * sj;
            a.rc[j] = rcj; a.d[j] = dj; a.q[j] = r3 / xj; a.v[j] = dj * (rcj - r3 / xj);
            rc2 += rcj * rcj; xs += r3; cx += cj * xj;
        }
    };
    // stop test of check_optimality (main.py:162-173): thread 0, result in `go`
    auto stop_test = [&](double rb2, double rc2, double gap, double obj) {
        if (tid == 0) {
            const double rb = sqrt(rb2), rcn = sqrt(rc2);
            sc->rb_norm = rb; sc->rc_norm = rcn; sc->gap = gap; sc->obj = obj;
            if (fabs(obj) < 1.7e308) sc->obj_last_finite = obj;
            sc->mu = gap / (double)n;
            sh[0] = gap / (double)n;
            int cont_loop = 1;
            if (!sc->force) {
                const bool cont = (sc->e1 * (1.0 + sc->b_norm) < rb) || (sc->e2 * (1.0 + sc->c_norm) < rcn) || (sc->e3 < gap);
                if (!cont) {
                    const bool finite = (rb == rb) && (rcn == rcn) && (gap == gap) && (fabs(rb) < 1.7e308) &&
                                        (fabs(rcn) < 1.7e308) && (fabs(gap) < 1.7e308);
                    sc->status = finite ? 1 : 3; sc->done = 1; cont_loop = 0;
                } else if (sc->k >= sc->max_iter) {
                    sc->status = 2; sc->done = 1; cont_loop = 0;
                }
            }
            go = cont_loop;
        }
    };
    // t1 = -r_b - A v ; z = inv(L) t1 ; dys = inv(L)^T z       (both matvecs with X = inv(L) from LDS)
    auto solve_normal = [&]() {
        if (row4 < m) {
            const int pb = a.A.rowptr[row4], pe = a.A.rowptr[row4 + 1];
            double acc = 0.0;
            for (int p = pb + l4; p < pe; p += 4) acc += a.A.rval[p] * a.v[a.A.colind[p]];
            acc += __shfl_xor(acc, 2, 4);
            acc += __shfl_xor(acc, 1, 4);
            if (l4 == 0) t1s[row4] = -rbs[row4] - acc;
        }
        __syncthreads();
        if (row4 < mt) {                                   // z_i = sum_{k <= i} X[i][k] t1_k,  X[i][k] at W[k*WLD + i + 1]
            double acc = 0.0;
            for (int k = l4; k <= row4; k += 4) acc += W[k * WLD + row4 + 1] * t1s[k];
            acc += __shfl_xor(acc, 2, 4);
            acc += __shfl_xor(acc, 1, 4);
            if (l4 == 0) zs[row4] = acc;
        }
        __syncthreads();
        if (row4 < mt) {                                   // dy_k = sum_{i >= k} X[i][k] z_i
            double acc = 0.0;
            for (int i = row4 + l4; i < mt; i += 4) acc += W[row4 * WLD + i + 1] * zs[i];
            acc += __shfl_xor(acc, 2, 4);
            acc += __shfl_xor(acc, 1, 4);
            if (l4 == 0) dys[row4] = acc;
        }
        __syncthreads();
    };
    // (dx, ds) from dys with the current q, v; ratio-test minima (main.py:227-228, 305-322)
    auto direction = [&](double* DX, double* DS, double& minp, double& mind) {
        for (int j = tid; j < n; j += PD_THREADS) {
            const int pb = a.A.colptr[j], pe = a.A.colptr[j + 1];
            double w = 0.0;
            for (int p = pb; p < pe; ++p) w += a.A.cval[p] * dys[a.A.rowind[p]];
            const double xj = a.x[j], sj = a.s[j];
            const double dxj = a.d[j] * w + a.v[j];
            const double dsj = (-sj * dxj) / xj - a.q[j];
            DX[j] = dxj; DS[j] = dsj;
            if (dxj < 0.0) minp = fmin(minp, -xj / dxj);
            if (dsj < 0.0) mind = fmin(mind, -sj / dsj);
        }
    };

    int steps = 0;
    for (;;) {
        // ---------------------------------------------------------------- residuals + stop test
        double r4[4] = {0.0, 0.0, 0.0, 0.0};               // ||r_b||^2, ||r_c||^2, x.s, c.x
        r4[0] = residual_rows();
        residual_cols(r4[1], r4[2], r4[3]);
        block_reduce512<4, false>(r4, red);
        stop_test(r4[0], r4[1], r4[2], r4[3]);
        __syncthreads();
        if (!go || steps >= a.max_steps) break;
        const double mu = sh[0];

        // ---------------------------------------------------------------- B = A diag(d) A^T into W, guarded Cholesky
        for (int idx = tid; idx < mt * WLD; idx += PD_THREADS) W[idx] = 0.0;
        __syncthreads();
        double mx[1] = {-1.7976931348623157e308};
        for (int e = tid; e < a.nb; e += PD_THREADS) {
            const int i = a.bi[e], k = a.bk[e];
            double acc = 0.0;
            for (int t = a.bptr[e]; t < a.bptr[e + 1]; ++t) acc += a.bcoef[t] * a.d[a.bcol[t]];
            W[i * WLD + k] = acc;
            if (i != k && (i >> 4) == (k >> 4)) W[k * WLD + i] = acc;       // diagonal tiles are held symmetric
            if (i == k) mx[0] = (acc > mx[0]) ? acc : mx[0];                 // NaN never wins
        }
        if (tid >= m && tid < mt) W[tid * WLD + tid] = 1.0;                  // padding rows of the last tile
        {   // max over the TRUE rows (fmax would drop a NaN as well); -max of negated values = max
            double ng[1] = {-mx[0]};
            block_reduce512<1, true>(ng, red);
            mx[0] = -ng[0];
        }
        if (a.shift_rel != 0.0 && tid < mt) W[tid * WLD + tid] += a.shift_rel * mx[0];
        __syncthreads();
        const int nfix = potrf_lds<false>(W, dinv_s, nt, a.eps * mx[0], a.big, nullptr);
        if (tid == 0) {
            sc->maxdiag = mx[0];
            const int fx = sc->fixed + nfix;
            sc->fixed = fx;
            int leave = 0;
            if (sc->k == 0) {
                sc->fixed_first = fx;
                if (a.auto_reg && (double)fx > 0.05 * (double)m) {          // > 5 % dependent rows: restart with the shift
                    sc->status = IPM_STATUS_NEEDS_SHIFT; sc->done = 1; leave = 1;
                }
            }
            go = !leave;
        }
        __syncthreads();
        if (!go) break;

        // ---------------------------------------------------------------- predictor
        solve_normal();
        double mn[2] = {1.0, 1.0};
        direction(a.dxa, a.dsa, mn[0], mn[1]);
        block_reduce512<2, true>(mn, red);
        const double aap = mn[0], aad = mn[1];
        double ma[1] = {0.0};
        for (int j = tid; j < n; j += PD_THREADS) ma[0] += (a.x[j] + aap * a.dxa[j]) * (a.s[j] + aad * a.dsa[j]);
        block_reduce512<1, false>(ma, red);
        const double mu_aff = ma[0] / (double)n;
        const double rr = mu_aff / mu;
        const double sigma = rr * rr * rr;
        const double sm = sigma * mu;
        // ---------------------------------------------------------------- corrector (main.py:150-152), same factor
        for (int j = tid; j < n; j += PD_THREADS) {
            const double xj = a.x[j];
            const double qj = (xj * a.s[j] + a.dxa[j] * a.dsa[j] - sm) / xj;
            a.q[j] = qj;
            a.v[j] = a.d[j] * (a.rc[j] - qj);
        }
        __syncthreads();
        solve_normal();
        double mc[2] = {1.0, 1.0};
        direction(a.dx, a.ds, mc[0], mc[1]);
        block_reduce512<2, true>(mc, red);
        // ---------------------------------------------------------------- damped step (main.py:604-626, 694-696)
        const double eta = sc->eta;
        const double ap = fmin(1.0, eta * mc[0]), ad = fmin(1.0, eta * mc[1]);
        for (int j = tid; j < n; j += PD_THREADS) {
            a.x[j] += ap * a.dx[j];
            a.s[j] += ad * a.ds[j];
        }
        if (tid < m) ys[tid] += ad * dys[tid];
        if (tid == 0) {
            const int k = sc->k;
            IterRec r;
            r.k = k; r.fixed = sc->fixed; r.obj = sc->obj; r.rb = sc->rb_norm; r.rc = sc->rc_norm; r.gap = sc->gap;
            r.mu = mu; r.sigma = sigma; r.aap = aap; r.aad = aad; r.ap = ap; r.ad = ad;
            a.hist[k % HIST_CAP] = r;
            sc->mu_aff = mu_aff; sc->sigma = sigma; sc->alpha_aff_p = aap; sc->alpha_aff_d = aad;
            sc->alpha_p = ap; sc->alpha_d = ad; sc->k = k + 1;
        }
        ++steps;
        __syncthreads();
    }
    if (tid < m) a.y[tid] = ys[tid];
}

}  // namespace ipm
