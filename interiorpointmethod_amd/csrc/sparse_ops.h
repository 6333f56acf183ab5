// sparse_ops.h -- sparse-A front end of the hot path for the Netlib instances (gfx950).
//
// The reference keeps A as a scipy CSC matrix (sparse_interior.py:215) and forms
// B = A @ D_square @ A.T with two SpGEMMs (main.py:223-224), then A @ v / A.T @ y SpMVs
// (main.py:66-73, 225, 227).  Here A lives on the device in BOTH CSR and CSC form (built
// once at upload); B is still the dense m x m matrix the blocked Cholesky factors, because
// the Cholesky fill of these normal matrices is near dense -- only its
// formation is sparse:  B[i][k] = sum_j a_ij d_j a_kj  costs sum_j nnz(A[:,j])^2 multiply-adds
// instead of m^2 n (STOCFOR3: 4.5e5 instead of 6.5e12).
//
// One workgroup owns one row of B (no atomics, fixed summation order => bitwise
// reproducible): it walks the nonzeros (i,j) of its row of A in CSR order and, for each,
// scatters a_ij d_j * A[:,j] (CSC column) into an LDS accumulator of length mp, then streams
// the finished row to HBM with coalesced 16-byte stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_nt_f64.h"

namespace ipm {

struct SparseA {
    const int* rowptr; const int* colind; const double* rval;    // CSR
    const int* colptr; const int* rowind; const double* cval;    // CSC
    int m, n;
};

constexpr int SP_LDS_MAX_MP = 19456;      // 152 KB accumulator row fits the 160 KB LDS

// B[row][:] = sum_j a_rj d_j A[:,j]^T  for row < m; padding rows get e_row.  grid = mp workgroups.
__device__ __forceinline__ void adat_sparse_kernel_body(SparseA A, const double* __restrict__ d, double* B,
                                                          int64_t ldb, int mp, const int* done, const unsigned bx_, const unsigned gx_) {
    if (done && *done) return;
    extern __shared__ __attribute__((aligned(16))) double acc[];          // mp doubles
    const int row = bx_, tid = threadIdx.x;
    for (int k = tid; k < mp; k += 256) acc[k] = 0.0;
    __syncthreads();
    if (row < A.m) {
        // what a nonzero needs before its column can be walked -- column index -> d_j and the column's extent -- is a chain of
        // dependent loads; it is fetched for up to 256 nonzeros of the row AT ONCE (one per thread) into LDS, then the nonzeros
        // are applied one after the other as before (fixed order: same bits)
        __shared__ double s_coef[256];
        __shared__ int s_qb[256], s_qe[256];
        const int pb = A.rowptr[row], pe = A.rowptr[row + 1];
        for (int p0 = pb; p0 < pe; p0 += 256) {
            const int cnt = min(256, pe - p0);
            if (tid < cnt) {
                const int j = A.colind[p0 + tid];
                s_coef[tid] = A.rval[p0 + tid] * d[j];
                s_qb[tid] = A.colptr[j]; s_qe[tid] = A.colptr[j + 1];
            }
            __syncthreads();
            for (int u = 0; u < cnt; ++u) {             // sequential over the row's nonzeros: fixed order
                const double coef = s_coef[u];
                const int qe = s_qe[u];
                for (int q = s_qb[u] + tid; q < qe; q += 256)    // distinct row indices within one column: no collisions
                    acc[A.rowind[q]] += coef * A.cval[q];
                __syncthreads();
            }
        }
    } else if (tid == 0) {
        acc[row] = 1.0;
    }
    __syncthreads();
    double* out = B + (int64_t)row * ldb;
    for (int k = tid * 2; k < mp; k += 512)
        *reinterpret_cast<f64x2*>(out + k) = (f64x2){acc[k], acc[k + 1]};
}
__global__ __launch_bounds__(256) void adat_sparse_kernel(SparseA A, const double* __restrict__ d, double* B,
                                                          int64_t ldb, int mp, const int* done) { adat_sparse_kernel_body(A, d, B, ldb, mp, done, blockIdx.x, gridDim.x); }

// zero an n-double buffer unless the solve is done (a plain memset would wipe the factor of a converged solve when
// iterations enqueued past convergence run as no-ops)
__device__ __forceinline__ void zero_unless_done_kernel_body(double* p, int64_t n, const int* done, const unsigned bx_, const unsigned gx_) {
    if (done && *done) return;
    const int64_t i = ((int64_t)bx_ * 256 + threadIdx.x) * 2, st = (int64_t)gx_ * 512;
    for (int64_t k = i; k < n; k += st) *reinterpret_cast<f64x2*>(p + k) = (f64x2){0.0, 0.0};
}
__global__ __launch_bounds__(256) void zero_unless_done_kernel(double* p, int64_t n, const int* done) { zero_unless_done_kernel_body(p, n, done, blockIdx.x, gridDim.x); }

// B from the PRODUCT LIST (built once on the host, ipm_set_A_csc): entry e = (bi[e], bk[e]) of A diag(d) A^T is
// sum_t (bai[t] d[bcol[t]]) bak[t], t in [bptr[e], bptr[e+1]) -- one thread per entry, terms in ascending column order,
// the same products in the same order as adat_sparse_kernel forms them (coef = a_ij d_j, then coef a_kj), so B is
// bit-identical to that kernel's.  The list holds the lower triangle plus, for every row, the entries up to the end of
// its 16 x 16 diagonal tile (potrf_diag reads those tiles whole).  B was zeroed by a memset on the same stream; threads
// nb .. nb + (mp - m) - 1 put the unit diagonal on the padding rows.  Replaces adat_sparse_kernel for sparse handles up
// to 1536 padded rows (that kernel gives every row of B a workgroup that walks the row's nonzeros one dependent load
// at a time); beyond that the zero fill of the dense B costs what the list saves.
__device__ __forceinline__ void adat_list_kernel_body(const int* __restrict__ bptr, const int* __restrict__ bi,
                                                        const int* __restrict__ bk, const int* __restrict__ bcol,
                                                        const double* __restrict__ bai, const double* __restrict__ bak, int nb,
                                                        const double* __restrict__ d, double* B, int64_t ldb, int m, int mp,
                                                        const int* done, const unsigned bx_, const unsigned gx_) {
    if (done && *done) return;
    const int e = bx_ * 256 + threadIdx.x;
    if (e < nb) {
        // the terms in ascending order, eight at a time: all loads of a batch are issued before the first is used (the list of an
        // entry is a chain of dependent loads otherwise: column index -> d), the sum itself stays sequential (same bits)
        double acc = 0.0;
        const int t1 = bptr[e + 1];
        int t = bptr[e];
        for (; t + 8 <= t1; t += 8) {
            double a_[8], k_[8], d_[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { a_[u] = bai[t + u]; k_[u] = bak[t + u]; d_[u] = d[bcol[t + u]]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const double coef = a_[u] * d_[u]; acc += coef * k_[u]; }
        }
        for (; t < t1; ++t) {
            const double coef = bai[t] * d[bcol[t]];
            acc += coef * bak[t];
        }
        B[(int64_t)bi[e] * ldb + bk[e]] = acc;
    } else if (e - nb < mp - m) {
        const int r = m + (e - nb);
        B[(int64_t)r * ldb + r] = 1.0;
    }
}
__global__ __launch_bounds__(256) void adat_list_kernel(const int* __restrict__ bptr, const int* __restrict__ bi,
                                                        const int* __restrict__ bk, const int* __restrict__ bcol,
                                                        const double* __restrict__ bai, const double* __restrict__ bak, int nb,
                                                        const double* __restrict__ d, double* B, int64_t ldb, int m, int mp,
                                                        const int* done) { adat_list_kernel_body(bptr, bi, bk, bcol, bai, bak, nb, d, B, ldb, m, mp, done, blockIdx.x, gridDim.x); }

// Same contract with the accumulator row in HBM (mp too large for LDS): the owning workgroup zeroes
// its row of B, then accumulates in place; __syncthreads() orders the read-modify-writes of one CU.
__device__ __forceinline__ void adat_sparse_global_kernel_body(SparseA A, const double* __restrict__ d, double* B,
                                                                 int64_t ldb, int mp, const int* done, const unsigned bx_, const unsigned gx_) {
    if (done && *done) return;
    const int row = bx_, tid = threadIdx.x;
    double* out = B + (int64_t)row * ldb;
    for (int k = tid; k < mp; k += 256) out[k] = (row >= A.m && k == row) ? 1.0 : 0.0;
    __syncthreads();
    if (row >= A.m) return;
    const int pb = A.rowptr[row], pe = A.rowptr[row + 1];
    for (int p = pb; p < pe; ++p) {
        const int j = A.colind[p];
        const double coef = A.rval[p] * d[j];
        const int qb = A.colptr[j], qe = A.colptr[j + 1];
        for (int q = qb + tid; q < qe; q += 256) out[A.rowind[q]] += coef * A.cval[q];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void adat_sparse_global_kernel(SparseA A, const double* __restrict__ d, double* B,
                                                                 int64_t ldb, int mp, const int* done) { adat_sparse_global_kernel_body(A, d, B, ldb, mp, done, blockIdx.x, gridDim.x); }

// out[i] = sa * (A[i,:] . v) + sb * add[i]  (CSR; 16 lanes per row, fixed-order tree reduction);
// rows m..mp-1 (padding) get sb * add[i].
__device__ __forceinline__ void spmv_csr_kernel_body(SparseA A, int mp, const double* __restrict__ v, double sa,
                                                       double sb, const double* __restrict__ add, double* out,
                                                       const int* done, const unsigned bx_, const unsigned gx_) {
    if (done && *done) return;
    const int l16 = threadIdx.x & 15;
    const int row = bx_ * 16 + (threadIdx.x >> 4);
    if (row >= mp) return;
    double s = 0.0;
    if (row < A.m) {
        const int pb = A.rowptr[row], pe = A.rowptr[row + 1];
        for (int p = pb + l16; p < pe; p += 16) s += A.rval[p] * v[A.colind[p]];
    }
    s += __shfl_xor(s, 8, 16);
    s += __shfl_xor(s, 4, 16);
    s += __shfl_xor(s, 2, 16);
    s += __shfl_xor(s, 1, 16);
    if (l16 == 0) out[row] = sa * s + (add ? sb * add[row] : 0.0);
}
__global__ __launch_bounds__(256) void spmv_csr_kernel(SparseA A, int mp, const double* __restrict__ v, double sa,
                                                       double sb, const double* __restrict__ add, double* out,
                                                       const int* done) { spmv_csr_kernel_body(A, mp, v, sa, sb, add, out, done, blockIdx.x, gridDim.x); }

// w[j] = A[:,j] . u  (CSC; 16 lanes per column).  Written to w[0..np): the vector kernels read it as
// the single "row chunk" of the dense GEMV-T partial buffer.
__device__ __forceinline__ void spmv_csc_t_kernel_body(SparseA A, int np, const double* __restrict__ u, double* w,
                                                         const int* done, const unsigned bx_, const unsigned gx_) {
    if (done && *done) return;
    const int l16 = threadIdx.x & 15;
    const int col = bx_ * 16 + (threadIdx.x >> 4);
    if (col >= np) return;
    double s = 0.0;
    if (col < A.n) {
        const int pb = A.colptr[col], pe = A.colptr[col + 1];
        for (int p = pb + l16; p < pe; p += 16) s += A.cval[p] * u[A.rowind[p]];
    }
    s += __shfl_xor(s, 8, 16);
    s += __shfl_xor(s, 4, 16);
    s += __shfl_xor(s, 2, 16);
    s += __shfl_xor(s, 1, 16);
    if (l16 == 0) w[col] = s;
}
__global__ __launch_bounds__(256) void spmv_csc_t_kernel(SparseA A, int np, const double* __restrict__ u, double* w,
                                                         const int* done) { spmv_csc_t_kernel_body(A, np, u, w, done, blockIdx.x, gridDim.x); }

}  // namespace ipm
