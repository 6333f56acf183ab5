// potrf_f64.h -- blocked, pivot-guarded fp64 Cholesky of the normal matrix and the
// triangular solves that go with it (gfx950).
//
// Replaces the LU factorization inside scipy's spsolve that the reference calls through
// solve_linear (main.py:176-182 of the reference repo) on B = A D^2 A^T (main.py:224-226).
// The reference factors twice per iteration (main.py:205, :266); B depends only on (x, s), so
// one factor serves predictor and corrector here.
//
// Right-looking, block size NB = 128:
//   for k: (1) potrf_diag_kernel  -- one workgroup factors the 128 x 128 diagonal block in
//              LDS (16-wide panels: a one-wave register/readlane factorization of the 16 x 16
//              pivot tile, then MFMA panel solve and MFMA trailing update inside LDS) and
//              also produces inv(L_kk), which turns the two steps below and the triangular
//              solves into pure GEMM/GEMV work;
//          (2) panel   L_ik = B_ik inv(L_kk)^T          (gemm_nt_f64_kernel, in place)
//          (3) update  B_ij -= L_ik L_jk^T, i >= j > k   (gemm_nt_f64_kernel, lower tiles)
// Pivot guard (LIPSOL style): a pivot p with !(p > eps*max diag(B)) is replaced
// by `big` (1e64) and counted, which zeroes that component of the solution instead of
// producing NaN on the rank-deficient / numerically semidefinite systems of the Netlib set.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_nt_f64.h"

namespace ipm {

constexpr int NB = 128;      // Cholesky block size
constexpr int WLD = 130;     // LDS leading dimension of the diagonal-block workspace (doubles)

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}

struct PotrfDiag {
    double* Bkk; int64_t ld;     // diagonal block (row-major, lower triangle read/written)
    double* inv;                 // out: inv(L_kk), dense 128 x 128 row-major, zeros above diag
    const double* maxdiag;       // device scalar: max diag of the unfactored B
    double eps, big;
    int* fixed;                  // device counter of guarded pivots (accumulates)
    const int* done;
};

// Workspace layout in LDS (one array, 128 x 130 doubles = 133 KB):
//   L[i][j]   (j <= i)  at W[i*WLD + j]
//   X[i][j]   (j <= i)  at W[j*WLD + i + 1]      X = inv(L), stored transposed one column right
// so the strict upper part of the square holds the inverse without a second array.
__global__ __launch_bounds__(256) void potrf_diag_kernel(PotrfDiag a) {
    if (a.done && *a.done) return;
    __shared__ __attribute__((aligned(16))) double W[NB * WLD];
    __shared__ int s_fixed;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const double thresh = a.eps * (*a.maxdiag);

    if (tid == 0) s_fixed = 0;
    // ---- load the block (rows complete up to the end of their 16-wide diagonal tile)
    for (int idx = tid; idx < NB * (NB / 2); idx += 256) {
        int i = idx / (NB / 2), c2 = (idx % (NB / 2)) * 2;
        if (c2 <= (i | 15)) {
            f64x2 v = *reinterpret_cast<const f64x2*>(a.Bkk + (int64_t)i * a.ld + c2);
            W[i * WLD + c2] = v.x;
            W[i * WLD + c2 + 1] = v.y;
        }
    }
    __syncthreads();

    for (int jb = 0; jb < NB / 16; ++jb) {
        const int c0 = jb * 16;
        // ---- (a) 16 x 16 pivot tile: factor + invert on wave 0, rows on lanes 0..15
        if (wave == 0) {
            const int i = fr;                 // lanes >= 16 mirror lanes 0..15 (results unused)
            double t[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) t[c] = W[(c0 + i) * WLD + c0 + c];
            double dinv[16];
            int nfix = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                double p = readlane_f64(t[j], j);
                if (!(p > thresh)) { p = a.big; ++nfix; }
                double ljj = sqrt(p);
                double inv = 1.0 / ljj;
                dinv[j] = inv;
                t[j] = (i == j) ? ljj : t[j] * inv;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) {
                    double lc = readlane_f64(t[j], c);
                    if (i >= c) t[c] -= t[j] * lc;
                }
            }
            // inverse of the tile: lane c owns column c of X = inv(T)
            double x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                double acc = (r == i) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < r; ++k) acc -= readlane_f64(t[k], r) * x[k];
                x[r] = acc * dinv[r];
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (c <= i) W[(c0 + i) * WLD + c0 + c] = t[c];             // L tile
                    if (c >= i) W[(c0 + i) * WLD + c0 + c + 1] = x[c];         // X[c][i] -> row i
                }
                if (lane == 0 && nfix) s_fixed += nfix;
            }
        }
        __syncthreads();
        // ---- (b) panel below the tile: P <- P * X^T  (X = inv(T)), one 16-row tile per wave turn
        const int nrt = NB / 16 - jb - 1;      // row tiles below
        for (int rt = wave; rt < nrt; rt += 4) {
            const int r0 = (jb + 1 + rt) * 16;
            double pa[4], xb[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                pa[kk] = W[(r0 + fr) * WLD + c0 + kk * 4 + fk];                // P[r][k]
                int k = kk * 4 + fk;                                           // X[fr][k], k <= fr
                xb[kk] = (k <= fr) ? W[(c0 + k) * WLD + c0 + fr + 1] : 0.0;
            }
            f64x4 acc = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[kk], xb[kk], acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) W[(r0 + fk + 4 * q) * WLD + c0 + fr] = acc[q];
        }
        __syncthreads();
        // ---- (c) trailing update inside the block: T(ib,cb) -= L(ib,jb) L(cb,jb)^T, jb<cb<=ib
        {
            const int ntile = nrt * (nrt + 1) / 2;
            for (int tix = wave; tix < ntile; tix += 4) {
                int ib = (int)((sqrtf(8.0f * (float)tix + 1.0f) - 1.0f) * 0.5f);
                while ((ib + 1) * (ib + 2) / 2 <= tix) ++ib;
                while (ib * (ib + 1) / 2 > tix) --ib;
                int cb = tix - ib * (ib + 1) / 2;
                const int r0 = (jb + 1 + ib) * 16, q0 = (jb + 1 + cb) * 16;
                f64x4 acc;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = W[(r0 + fk + 4 * q) * WLD + q0 + fr];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    double av = -W[(r0 + fr) * WLD + c0 + kk * 4 + fk];
                    double bv = W[(q0 + fr) * WLD + c0 + kk * 4 + fk];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) W[(r0 + fk + 4 * q) * WLD + q0 + fr] = acc[q];
            }
        }
        __syncthreads();
    }

    // ---- inverse of the whole block, one 16-row block row at a time:
    //      X(i,j) = -X(i,i) * sum_{k=j}^{i-1} L(i,k) X(k,j),  j < i
    for (int ib = 1; ib < NB / 16; ++ib) {
        for (int jt = wave; jt < ib; jt += 4) {
            f64x4 s = (f64x4){0.0, 0.0, 0.0, 0.0};
            for (int kb = jt; kb < ib; ++kb) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    double av = W[(ib * 16 + fr) * WLD + kb * 16 + kk * 4 + fk];        // L(i,k)[r][k]
                    int kr = kb * 16 + kk * 4 + fk, cc = jt * 16 + fr;                  // X[kr][cc]
                    double bv = (kr >= cc) ? W[cc * WLD + kr + 1] : 0.0;
                    s = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, s, 0, 0, 0);
                }
            }
            // second product: accumulator register q of S is row fk+4q, so pair it with
            // X(i,i)[fr][fk+4q] (any order of k is fine as long as both operands agree)
            f64x4 r = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int k = fk + 4 * q;
                double av = (k <= fr) ? -W[(ib * 16 + k) * WLD + ib * 16 + fr + 1] : 0.0;  // -X(i,i)[fr][k]
                r = __builtin_amdgcn_mfma_f64_16x16x4f64(av, s[q], r, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)                                                  // X(i,j)[fk+4q][fr]
                W[(jt * 16 + fr) * WLD + ib * 16 + fk + 4 * q + 1] = r[q];
        }
        __syncthreads();
    }

    // ---- write back: L (lower) to B, inverse to `inv` (dense, zero above the diagonal)
    for (int idx = tid; idx < NB * NB; idx += 256) {
        int i = idx / NB, j = idx % NB;
        if (j <= i) a.Bkk[(int64_t)i * a.ld + j] = W[i * WLD + j];
    }
    for (int idx = tid; idx < NB * NB; idx += 256) {
        int i = idx / NB, j = idx % NB;
        a.inv[idx] = (j <= i) ? W[j * WLD + i + 1] : 0.0;
    }
    if (tid == 0 && s_fixed) atomicAdd(a.fixed, s_fixed);
}

// max of the diagonal of an n x n matrix (single workgroup; n <= a few 10^4)
__global__ __launch_bounds__(256) void maxdiag_kernel(const double* B, int64_t ld, int n, double* out,
                                                      const int* done) {
    if (done && *done) return;
    __shared__ double red[256];
    double mx = -1.7976931348623157e308;
    for (int i = threadIdx.x; i < n; i += 256) {
        double v = B[(int64_t)i * ld + i];
        mx = (v > mx) ? v : mx;          // NaN never wins: the guard then fires on every NaN pivot
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

// ------------------------------------------------------------------------------------------
// Triangular solves with the factor (L in the lower triangle of B, inv(L_kk) per block).
// One launch per 128-row block step; inside a launch every workgroup first forms the
// solution block z_k = inv(L_kk) r_k (a 128 x 128 GEMV, recomputed per workgroup so no
// inter-workgroup hand-off is needed) and then eliminates it from its own row block.
// Sums are in a fixed order: results are bitwise reproducible.
// ------------------------------------------------------------------------------------------

// y[128] = M[128x128] * v  (M row-major, ldm) -- all 256 threads; v in LDS; result in LDS `out`.
__device__ __forceinline__ void block_gemv_n(const double* __restrict__ M, int64_t ldm,
                                             const double* vs, double* out) {
    const int tid = threadIdx.x;
    const int l16 = tid & 15, rg = tid >> 4;          // 16 lanes per row, 16 rows per pass
    double v[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[2 * q] = vs[(l16 + 16 * q) * 2]; v[2 * q + 1] = vs[(l16 + 16 * q) * 2 + 1]; }
    double part[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const double* row = M + (int64_t)(p * 16 + rg) * ldm;
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f64x2 m2 = *reinterpret_cast<const f64x2*>(row + (l16 + 16 * q) * 2);
            acc += m2.x * v[2 * q] + m2.y * v[2 * q + 1];
        }
        part[p] = acc;
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        double s = part[p];
        s += __shfl_xor(s, 8, 16);
        s += __shfl_xor(s, 4, 16);
        s += __shfl_xor(s, 2, 16);
        s += __shfl_xor(s, 1, 16);
        if (l16 == 0) out[p * 16 + rg] = s;
    }
}

// y[128] = M^T * v  (y[c] = sum_r M[r][c] v[r]) -- all 256 threads; v in LDS; result in LDS `out`;
// `scratch` is 16*128 doubles of LDS.
__device__ __forceinline__ void block_gemv_t(const double* __restrict__ M, int64_t ldm,
                                             const double* vs, double* out, double* scratch) {
    const int tid = threadIdx.x;
    const int l16 = tid & 15, rg = tid >> 4;
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int r = p * 16 + rg;
        const double* row = M + (int64_t)r * ldm;
        const double vr = vs[r];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f64x2 m2 = *reinterpret_cast<const f64x2*>(row + (l16 + 16 * q) * 2);
            acc[2 * q] += m2.x * vr;
            acc[2 * q + 1] += m2.y * vr;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        scratch[rg * NB + (l16 + 16 * q) * 2] = acc[2 * q];
        scratch[rg * NB + (l16 + 16 * q) * 2 + 1] = acc[2 * q + 1];
    }
    __syncthreads();
    if (tid < NB) {
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < 16; ++g) s += scratch[g * NB + tid];
        out[tid] = s;
    }
    __syncthreads();
}

struct TrsvStep {
    const double* L; int64_t ld;     // factor (lower), mp x mp
    const double* inv;               // inv(L_kk) blocks, [nblk][128*128]
    double* r;                       // running right-hand side (consumed)
    double* z;                       // solution
    int k;                           // block step
    const int* done;
};

// forward step k: z_k = inv(L_kk) r_k ; r_i -= L_ik z_k for i > k.  grid = nblk - k.
__global__ __launch_bounds__(256) void trsv_fwd_step_kernel(TrsvStep a) {
    if (a.done && *a.done) return;
    __shared__ double vs[NB], zs[NB], us[NB];
    const int tid = threadIdx.x;
    const int i = a.k + blockIdx.x;
    if (tid < NB) vs[tid] = a.r[(int64_t)a.k * NB + tid];
    __syncthreads();
    block_gemv_n(a.inv + (int64_t)a.k * NB * NB, NB, vs, zs);
    __syncthreads();
    if (blockIdx.x == 0) {
        if (tid < NB) a.z[(int64_t)a.k * NB + tid] = zs[tid];
        return;
    }
    block_gemv_n(a.L + (int64_t)i * NB * a.ld + (int64_t)a.k * NB, a.ld, zs, us);
    __syncthreads();
    if (tid < NB) a.r[(int64_t)i * NB + tid] -= us[tid];
}

// backward step k (descending): w_k = inv(L_kk)^T z_k ; z_j -= L_kj^T w_k for j < k. grid = k+1.
// Block j == k writes w_k to `z` (the solution); blocks j < k update the running rhs `r`.
__global__ __launch_bounds__(256) void trsv_bwd_step_kernel(TrsvStep a) {
    if (a.done && *a.done) return;
    __shared__ double vs[NB], ws[NB], us[NB];
    __shared__ double scratch[16 * NB];
    const int tid = threadIdx.x;
    const int j = blockIdx.x;
    if (tid < NB) vs[tid] = a.r[(int64_t)a.k * NB + tid];
    __syncthreads();
    block_gemv_t(a.inv + (int64_t)a.k * NB * NB, NB, vs, ws, scratch);
    if (j == a.k) {
        if (tid < NB) a.z[(int64_t)a.k * NB + tid] = ws[tid];
        return;
    }
    block_gemv_t(a.L + (int64_t)a.k * NB * a.ld + (int64_t)j * NB, a.ld, ws, us, scratch);
    if (tid < NB) a.r[(int64_t)j * NB + tid] -= us[tid];
}

}  // namespace ipm
