// trsv_grouped.h -- triangular solves with 1024-row groups (gfx950).
//
// The per-128-block substitution (potrf_f64.h: trsv_*_step_kernel) is launch-latency bound:
// 2 x 32 dependent launches per solve at m = 4096, ~8 us each, four sweeps per iteration.  Here
// the factor's diagonal is regrouped into 1024 x 1024 lower-triangular blocks D_g whose explicit
// inverses X_g = inv(D_g) are assembled right after the factorization from the 128-block inverses
// the Cholesky already produced (batched MFMA GEMMs over all groups):
// by recursive doubling (128 -> 256 -> 512 -> 1024), three batched NT GEMMs per level that keep both
// X and XT = X^T current (the K-contiguous "NT" kernel needs one operand of each kind):
//     S   = XT11 * L21^T ;   X21 = -X22 * S^T ;   XT12 = -S * X22^T        ( = inv([[L11,0],[L21,L22]]) )
// A solve is then 4 group steps of dense GEMVs at HBM speed instead of 32 block steps:
//     forward : z_g = X_g r_g ;  r_below -= L[below, g] z_g
//     backward: w_g = XT_g z_g ;  z_left  -= L[g, left]^T w_g
// Same arithmetic as the block substitution up to rounding (explicit inverses of well-scaled
// 1024-blocks; the 128-block inverses they are built from are used by the Cholesky anyway).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_nt_f64.h"

namespace ipm {

constexpr int GS_MAX = 8;             // 128-blocks per group: 8 from 16 blocks on (1024-row groups); smaller handles use the
                                      // largest power of two that divides their block count (8, 4 or 2), so that an LP of 8
                                      // blocks solves with ONE explicit inverse and two GEMVs per substitution instead of
                                      // sixteen block-step launches

// X_g[a*128 + r][a*128 + c] = inv(L_bb)[r][c] and XT_g = its transpose, for every 128-block b = GS g + a.
// grid (4, 4, nblocks), block (32, 8).
__device__ __forceinline__ void group_diag_transpose_kernel_body(const double* __restrict__ invD, double* XT, double* X, int b0, int GS, const int* done,
                                                                  const unsigned bx_, const unsigned by_, const unsigned bz_) {
    if (done && *done) return;
    __shared__ double tile[32][33];
    const int64_t GR = (int64_t)GS * 128;
    const int b = b0 + bz_, g = b / GS, a = b % GS;
    const double* src = invD + (int64_t)b * 128 * 128;
    double* dst = XT + (int64_t)g * GR * GR + (int64_t)(a * 128) * GR + a * 128;
    double* dsx = X + (int64_t)g * GR * GR + (int64_t)(a * 128) * GR + a * 128;
    const int bx = bx_ * 32, by = by_ * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        double v = src[(int64_t)(by + j) * 128 + bx + threadIdx.x];
        tile[j][threadIdx.x] = v;
        dsx[(int64_t)(by + j) * GR + bx + threadIdx.x] = v;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) dst[(int64_t)(bx + j) * GR + by + threadIdx.x] = tile[threadIdx.x][j];
}
__global__ void group_diag_transpose_kernel(const double* __restrict__ invD, double* XT, double* X, int b0, int GS, const int* done) {
    group_diag_transpose_kernel_body(invD, XT, X, b0, GS, done, blockIdx.x, blockIdx.y, blockIdx.z);
}

// z[c] -= sum_rc part[rc*np + c], c < np  (fixed order)
__device__ __forceinline__ void sub_partials_kernel_body(double* z, const double* __restrict__ part, int np, int rc,
                                                           const int* done, const unsigned bx_, const unsigned gx_) {
    if (done && *done) return;
    const int c = bx_ * 256 + threadIdx.x;
    if (c >= np) return;
    double s = 0.0;
    int r = 0;
    for (; r + 8 <= rc; r += 8) {                          // eight loads in flight, summed in chunk order
        double t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = part[(int64_t)(r + q) * np + c];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += t[q];
    }
    for (; r < rc; ++r) s += part[(int64_t)r * np + c];
    z[c] -= s;
}
__global__ __launch_bounds__(256) void sub_partials_kernel(double* z, const double* __restrict__ part, int np, int rc,
                                                           const int* done) { sub_partials_kernel_body(z, part, np, rc, done, blockIdx.x, gridDim.x); }

}  // namespace ipm
