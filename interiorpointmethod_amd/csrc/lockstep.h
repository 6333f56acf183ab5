// lockstep.h -- the LOCKSTEP BATCH of the batched-LP mode (gfx950): iteration k of SEVERAL independent LPs in the SAME launches.
//
// Why.  A Netlib-size LP is a chain of ~100 small dependent launches per interior-point iteration (reference loop:
// main.py:780-807; the driver loop being batched: script.py:147-173).  Solving several LPs at once from several streams does
// not overlap more than FOUR of those chains: HIP multiplexes the streams onto four hardware queues, two streams of one queue
// serialise, and more queues are slower (profiles/r04_netlib_*_rejected.txt) -- the 73-LP suite is bound at (sum of the chains) / 4.
// What raises the concurrency is not more chains but kernels that serve several LPs per launch: blockIdx.y = LP, the arguments of
// every LP's launch come from a device table, blockIdx.x runs up to the largest grid of the group and an LP's surplus blocks
// leave at once.  The launch count of a batch iteration is then that of its LONGEST program, not the sum.
//
// How.  Nothing about an LP's arithmetic changes: the handle's own launch sequence (enqueue_iteration, single-stream path) is
// RECORDED once -- every launch site pushes (kernel type, grid, argument struct) instead of launching -- and the records of all
// LPs are merged, each LP's order preserved, into global steps of one kernel type each (ipm_api.hip: ls_merge).  The device
// code of a step is the body of the kernel the handle would have launched (X_kernel_body, shared with the one-LP kernels), so a
// lockstep solve is BIT-IDENTICAL to the same handle solved alone (tests/test_gpu_lockstep.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chol_update_f64.h"
#include "gemm_nt_f64.h"
#include "potrf_f64.h"
#include "sparse_ops.h"
#include "vector_ops.h"

namespace ipm {

enum LsType {
    LS_SPMV_CSR = 0, LS_SPMV_CSC_T, LS_PREPARE, LS_STOP_TEST, LS_ZERO, LS_ADAT_LIST, LS_ADAT_SPARSE, LS_ADAT_SPARSE_GLOBAL, LS_MAXDIAG,
    LS_POTRF, LS_GEMM_32_128_32, LS_GEMM_64_64_16, LS_GEMM_64_128_16, LS_GEMM_128_128_16, LS_GEMM_32_32_32, LS_CHOL_UPDATE, LS_TRSV_FWD, LS_TRSV_BWD,
    LS_DIRECTION, LS_MU_AFF, LS_CORR_RHS, LS_UPDATE, LS_NTYPES
};

constexpr int LS_ARG_BYTES = 304;
struct LsRec {                                // one LP's share of one global step
    unsigned gridx;                           // blocks of this LP's launch (blockIdx.x beyond it: nothing to do)
    unsigned lds;                             // dynamic LDS bytes of this LP's launch (adat_sparse_kernel only)
    alignas(8) unsigned char args[LS_ARG_BYTES];
};

struct LsVecA { VecArgs a; int corr; };
struct LsSpmv { SparseA A; int mp; const double* v; double sa, sb; const double* add; double* out; const int* done; };
struct LsSpmvT { SparseA A; int np; const double* u; double* w; const int* done; };
struct LsZero { double* p; int64_t n; const int* done; };
struct LsAdatList { const int *bptr, *bi, *bk, *bcol; const double *bai, *bak; int nb; const double* d; double* B; int64_t ldb; int m, mp; const int* done; };
struct LsAdatSp { SparseA A; const double* d; double* B; int64_t ldb; int mp; const int* done; };
struct LsMaxdiag { const double* B; int64_t ld; int n; double* out; const int* done; };
static_assert(sizeof(LsVecA) <= LS_ARG_BYTES && sizeof(GemmNT) <= LS_ARG_BYTES && sizeof(PotrfDiag) <= LS_ARG_BYTES && sizeof(LsAdatList) <= LS_ARG_BYTES, "LsRec::args");

#define LS_ENTER(ARGT)                                                   \
    const LsRec& r_ = recs[blockIdx.y];                                  \
    if (blockIdx.x >= r_.gridx) return;                                  \
    const ARGT& p = *reinterpret_cast<const ARGT*>(r_.args)

__global__ __launch_bounds__(256) void ls_spmv_csr(const LsRec* recs) { LS_ENTER(LsSpmv); spmv_csr_kernel_body(p.A, p.mp, p.v, p.sa, p.sb, p.add, p.out, p.done, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(256) void ls_spmv_csc_t(const LsRec* recs) { LS_ENTER(LsSpmvT); spmv_csc_t_kernel_body(p.A, p.np, p.u, p.w, p.done, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(VBLK) void ls_prepare(const LsRec* recs) { LS_ENTER(LsVecA); prepare_kernel_body(p.a, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(64) void ls_stop_test(const LsRec* recs) { LS_ENTER(LsVecA); stop_test_kernel_body(p.a, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(256) void ls_zero(const LsRec* recs) { LS_ENTER(LsZero); zero_unless_done_kernel_body(p.p, p.n, p.done, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(256) void ls_adat_list(const LsRec* recs) {
    LS_ENTER(LsAdatList);
    adat_list_kernel_body(p.bptr, p.bi, p.bk, p.bcol, p.bai, p.bak, p.nb, p.d, p.B, p.ldb, p.m, p.mp, p.done, blockIdx.x, r_.gridx);
}
__global__ __launch_bounds__(256) void ls_adat_sparse(const LsRec* recs) { LS_ENTER(LsAdatSp); adat_sparse_kernel_body(p.A, p.d, p.B, p.ldb, p.mp, p.done, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(256) void ls_adat_sparse_global(const LsRec* recs) { LS_ENTER(LsAdatSp); adat_sparse_global_kernel_body(p.A, p.d, p.B, p.ldb, p.mp, p.done, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(256) void ls_maxdiag(const LsRec* recs) { LS_ENTER(LsMaxdiag); maxdiag_kernel_body(p.B, p.ld, p.n, p.out, p.done, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(PD_THREADS) void ls_potrf(const LsRec* recs) {
    LS_ENTER(PotrfDiag);
    if (p.done && *p.done) return;                           // (no signal word on this path: a lockstep handle never polls)
    __shared__ __attribute__((aligned(16))) double W[NB * WLD];
    __shared__ double dinv_s[NB];
    potrf_diag_body<false>(p, W, dinv_s);
}
template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 2 < 2 ? 1 : 2) void ls_gemm(const LsRec* recs) {
    LS_ENTER(GemmNT);
    if (p.done && *p.done) return;
    __shared__ __attribute__((aligned(16))) double lds[2 * (BM + BN) * (BK + 2)];
    gemm_nt_body<BM, BN, BK, WM, WN, false>(p, (int)blockIdx.x, 0, 0, lds);
}
__global__ __launch_bounds__(256, 2) void ls_chol_update(const LsRec* recs) { LS_ENTER(GemmNT); chol_update_kernel_body(p, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(256) void ls_trsv_fwd(const LsRec* recs) { LS_ENTER(TrsvStep); trsv_fwd_step_kernel_body(p, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(256) void ls_trsv_bwd(const LsRec* recs) { LS_ENTER(TrsvStep); trsv_bwd_step_kernel_body(p, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(VBLK) void ls_direction(const LsRec* recs) { LS_ENTER(LsVecA); direction_kernel_body(p.a, p.corr, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(VBLK) void ls_mu_aff(const LsRec* recs) { LS_ENTER(LsVecA); mu_aff_kernel_body(p.a, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(VBLK) void ls_corr_rhs(const LsRec* recs) { LS_ENTER(LsVecA); corrector_rhs_kernel_body(p.a, blockIdx.x, r_.gridx); }
__global__ __launch_bounds__(VBLK) void ls_update(const LsRec* recs) { LS_ENTER(LsVecA); update_kernel_body(p.a, blockIdx.x, r_.gridx); }

// launch one global step: `count` LPs, the largest grid `gridx` and dynamic LDS `lds` among them
inline hipError_t ls_launch(int type, const LsRec* d_recs, unsigned count, unsigned gridx, unsigned lds, hipStream_t st) {
    const dim3 g(gridx, count);
    switch (type) {
        case LS_SPMV_CSR: hipLaunchKernelGGL(ls_spmv_csr, g, dim3(256), 0, st, d_recs); break;
        case LS_SPMV_CSC_T: hipLaunchKernelGGL(ls_spmv_csc_t, g, dim3(256), 0, st, d_recs); break;
        case LS_PREPARE: hipLaunchKernelGGL(ls_prepare, g, dim3(VBLK), 0, st, d_recs); break;
        case LS_STOP_TEST: hipLaunchKernelGGL(ls_stop_test, g, dim3(64), 0, st, d_recs); break;
        case LS_ZERO: hipLaunchKernelGGL(ls_zero, g, dim3(256), 0, st, d_recs); break;
        case LS_ADAT_LIST: hipLaunchKernelGGL(ls_adat_list, g, dim3(256), 0, st, d_recs); break;
        case LS_ADAT_SPARSE: hipLaunchKernelGGL(ls_adat_sparse, g, dim3(256), lds, st, d_recs); break;
        case LS_ADAT_SPARSE_GLOBAL: hipLaunchKernelGGL(ls_adat_sparse_global, g, dim3(256), 0, st, d_recs); break;
        case LS_MAXDIAG: hipLaunchKernelGGL(ls_maxdiag, g, dim3(256), 0, st, d_recs); break;
        case LS_POTRF: hipLaunchKernelGGL(ls_potrf, g, dim3(PD_THREADS), 0, st, d_recs); break;
        case LS_GEMM_32_128_32: hipLaunchKernelGGL((ls_gemm<32, 128, 32, 1, 8>), g, dim3(512), 0, st, d_recs); break;
        case LS_GEMM_64_64_16: hipLaunchKernelGGL((ls_gemm<64, 64, 16, 2, 2>), g, dim3(256), 0, st, d_recs); break;
        case LS_GEMM_64_128_16: hipLaunchKernelGGL((ls_gemm<64, 128, 16, 2, 2>), g, dim3(256), 0, st, d_recs); break;
        case LS_GEMM_128_128_16: hipLaunchKernelGGL((ls_gemm<128, 128, 16, 2, 2>), g, dim3(256), 0, st, d_recs); break;
        case LS_GEMM_32_32_32: hipLaunchKernelGGL((ls_gemm<32, 32, 32, 2, 2>), g, dim3(256), 0, st, d_recs); break;
        case LS_CHOL_UPDATE: hipLaunchKernelGGL(ls_chol_update, g, dim3(256), 0, st, d_recs); break;
        case LS_TRSV_FWD: hipLaunchKernelGGL(ls_trsv_fwd, g, dim3(256), 0, st, d_recs); break;
        case LS_TRSV_BWD: hipLaunchKernelGGL(ls_trsv_bwd, g, dim3(256), 0, st, d_recs); break;
        case LS_DIRECTION: hipLaunchKernelGGL(ls_direction, g, dim3(VBLK), 0, st, d_recs); break;
        case LS_MU_AFF: hipLaunchKernelGGL(ls_mu_aff, g, dim3(VBLK), 0, st, d_recs); break;
        case LS_CORR_RHS: hipLaunchKernelGGL(ls_corr_rhs, g, dim3(VBLK), 0, st, d_recs); break;
        case LS_UPDATE: hipLaunchKernelGGL(ls_update, g, dim3(VBLK), 0, st, d_recs); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ipm
