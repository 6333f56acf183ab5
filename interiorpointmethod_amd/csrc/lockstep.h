// lockstep.h -- the LOCKSTEP BATCH of the batched-LP mode (gfx950): iteration k of SEVERAL independent LPs in the SAME launches.
//
// Why.  A Netlib-size LP is a chain of ~100 small dependent launches per interior-point iteration (reference loop:
// main.py:780-807; the driver loop being batched: script.py:147-173).  Solving several LPs at once from several streams does
// not overlap more than FOUR of those chains: HIP multiplexes the streams onto four hardware queues, two streams of one queue
// serialise, and more queues are slower (profiles/r04_netlib_*_rejected.txt) -- the 73-LP suite is bound at (sum of the chains) / 4.
// What raises the concurrency is not more chains but kernels that serve several LPs per launch: the grids of the LPs of a step
// are packed back to back into ONE 1-D grid, the arguments of every LP's launch come from a device table (LsRec), and a block
// finds its LP with one wave-wide load of the start offsets + a ballot (LS_ENTER below) -- no surplus blocks, whatever the size
// mix.  The launch count of a batch iteration is then that of its LONGEST program, not the sum.
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
#include "trsv_grouped.h"
#include "vector_ops.h"

namespace ipm {

enum LsType {
    LS_SPMV_CSR = 0, LS_SPMV_CSC_T, LS_PREPARE, LS_STOP_TEST, LS_ZERO, LS_ADAT_LIST, LS_ADAT_SPARSE, LS_ADAT_SPARSE_GLOBAL, LS_MAXDIAG,
    LS_POTRF, LS_GEMM_32_128_32, LS_GEMM_64_64_16, LS_GEMM_64_128_16, LS_GEMM_128_128_16, LS_GEMM_32_32_32, LS_CHOL_UPDATE, LS_TRSV_FWD, LS_TRSV_BWD,
    LS_DIRECTION, LS_MU_AFF, LS_CORR_RHS, LS_UPDATE, LS_GEMV_N, LS_GEMV_T, LS_SUB_PARTIALS, LS_GROUP_DIAG_T, LS_GEMM_32_32_32_BATCHED, LS_NTYPES
};

constexpr int LS_ARG_BYTES = 304;
constexpr int LS_MAX_GROUP = 64;              // LPs per launch: the block -> LP look-up is one wave-wide load + ballot
struct LsRec {                                // one LP's share of one global step
    unsigned gridx;                           // blocks of this LP's launch
    unsigned lds;                             // dynamic LDS bytes of this LP's launch (adat_sparse_kernel only)
    unsigned start;                           // first block of this LP inside the step's 1-D grid (the grids are packed back to back:
                                              // a 2-D grid of max-gridx x LPs would dispatch tens of thousands of blocks that leave at once)
    unsigned pad_;
    alignas(8) unsigned char args[LS_ARG_BYTES];
};

struct LsVecA { VecArgs a; int corr; };
struct LsSpmv { SparseA A; int mp; const double* v; double sa, sb; const double* add; double* out; const int* done; };
struct LsSpmvT { SparseA A; int np; const double* u; double* w; const int* done; };
struct LsZero { double* p; int64_t n; const int* done; };
struct LsAdatList { const int *bptr, *bi, *bk, *bcol; const double *bai, *bak; int nb; const double* d; double* B; int64_t ldb; int m, mp; const int* done; };
struct LsAdatSp { SparseA A; const double* d; double* B; int64_t ldb; int mp; const int* done; };
struct LsGemvN { const double* A; int64_t lda; int mp, np; const double* v; double sa, sb; const double* add; double* out; const int* done; };
struct LsGemvT { const double* A; int64_t lda; int rows_per_chunk, np; const double* u; double* part; const int* done; unsigned gx; };   // gx: blocks per row chunk
struct LsSubPart { double* z; const double* part; int np, rc; const int* done; };
struct LsGroupDiagT { const double* invD; double* XT; double* X; int b0, GS; const int* done; };
struct LsMaxdiag { const double* B; int64_t ld; int n; double* out; const int* done; };
static_assert(sizeof(LsVecA) <= LS_ARG_BYTES && sizeof(GemmNT) <= LS_ARG_BYTES && sizeof(PotrfDiag) <= LS_ARG_BYTES && sizeof(LsAdatList) <= LS_ARG_BYTES, "LsRec::args");

// block -> (LP, block of that LP's launch): the LPs' first blocks are ascending; every wave looks its own block up
#define LS_ENTER(ARGT)                                                                                                  \
    const unsigned lane_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));                          \
    const unsigned s_ = lane_ < count ? recs[lane_].start : 0xffffffffu;                                                \
    const int lp_ = __popcll(__ballot(s_ <= blockIdx.x)) - 1;                                                           \
    const LsRec& r_ = recs[lp_];                                                                                        \
    const unsigned bx = blockIdx.x - r_.start;                                                                          \
    const ARGT& p = *reinterpret_cast<const ARGT*>(r_.args)

__global__ __launch_bounds__(256) void ls_spmv_csr(const LsRec* recs, const unsigned count) { LS_ENTER(LsSpmv); spmv_csr_kernel_body(p.A, p.mp, p.v, p.sa, p.sb, p.add, p.out, p.done, bx, r_.gridx); }
__global__ __launch_bounds__(256) void ls_spmv_csc_t(const LsRec* recs, const unsigned count) { LS_ENTER(LsSpmvT); spmv_csc_t_kernel_body(p.A, p.np, p.u, p.w, p.done, bx, r_.gridx); }
__global__ __launch_bounds__(VBLK) void ls_prepare(const LsRec* recs, const unsigned count) { LS_ENTER(LsVecA); prepare_kernel_body(p.a, bx, r_.gridx); }
__global__ __launch_bounds__(64) void ls_stop_test(const LsRec* recs, const unsigned count) { LS_ENTER(LsVecA); stop_test_kernel_body(p.a, bx, r_.gridx); }
__global__ __launch_bounds__(256) void ls_zero(const LsRec* recs, const unsigned count) { LS_ENTER(LsZero); zero_unless_done_kernel_body(p.p, p.n, p.done, bx, r_.gridx); }
__global__ __launch_bounds__(256) void ls_adat_list(const LsRec* recs, const unsigned count) {
    LS_ENTER(LsAdatList);
    adat_list_kernel_body(p.bptr, p.bi, p.bk, p.bcol, p.bai, p.bak, p.nb, p.d, p.B, p.ldb, p.m, p.mp, p.done, bx, r_.gridx);
}
__global__ __launch_bounds__(256) void ls_adat_sparse(const LsRec* recs, const unsigned count) { LS_ENTER(LsAdatSp); adat_sparse_kernel_body(p.A, p.d, p.B, p.ldb, p.mp, p.done, bx, r_.gridx); }
__global__ __launch_bounds__(256) void ls_adat_sparse_global(const LsRec* recs, const unsigned count) { LS_ENTER(LsAdatSp); adat_sparse_global_kernel_body(p.A, p.d, p.B, p.ldb, p.mp, p.done, bx, r_.gridx); }
__global__ __launch_bounds__(256) void ls_maxdiag(const LsRec* recs, const unsigned count) { LS_ENTER(LsMaxdiag); maxdiag_kernel_body(p.B, p.ld, p.n, p.out, p.done, bx, r_.gridx); }
__global__ __launch_bounds__(PD_THREADS) void ls_potrf(const LsRec* recs, const unsigned count) {
    LS_ENTER(PotrfDiag);
    if (p.done && *p.done) return;                           // (no signal word on this path: a lockstep handle never polls)
    __shared__ __attribute__((aligned(16))) double W[NB * WLD];
    __shared__ double dinv_s[NB];
    potrf_diag_body<false>(p, W, dinv_s);
}
template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 2 < 2 ? 1 : 2) void ls_gemm(const LsRec* recs, const unsigned count) {
    LS_ENTER(GemmNT);
    if (p.done && *p.done) return;
    __shared__ __attribute__((aligned(16))) double lds[2 * (BM + BN) * (BK + 2)];
    gemm_nt_body<BM, BN, BK, WM, WN, false>(p, (int)bx, 0, 0, lds);
}
__global__ __launch_bounds__(256, 2) void ls_chol_update(const LsRec* recs, const unsigned count) { LS_ENTER(GemmNT); chol_update_kernel_body(p, bx, r_.gridx); }
__global__ __launch_bounds__(256) void ls_trsv_fwd(const LsRec* recs, const unsigned count) { LS_ENTER(TrsvStep); trsv_fwd_step_kernel_body(p, bx, r_.gridx); }
__global__ __launch_bounds__(256) void ls_trsv_bwd(const LsRec* recs, const unsigned count) { LS_ENTER(TrsvStep); trsv_bwd_step_kernel_body(p, bx, r_.gridx); }
__global__ __launch_bounds__(VBLK) void ls_direction(const LsRec* recs, const unsigned count) { LS_ENTER(LsVecA); direction_kernel_body(p.a, p.corr, bx, r_.gridx); }
__global__ __launch_bounds__(VBLK) void ls_mu_aff(const LsRec* recs, const unsigned count) { LS_ENTER(LsVecA); mu_aff_kernel_body(p.a, bx, r_.gridx); }
__global__ __launch_bounds__(VBLK) void ls_corr_rhs(const LsRec* recs, const unsigned count) { LS_ENTER(LsVecA); corrector_rhs_kernel_body(p.a, bx, r_.gridx); }
__global__ __launch_bounds__(VBLK) void ls_update(const LsRec* recs, const unsigned count) { LS_ENTER(LsVecA); update_kernel_body(p.a, bx, r_.gridx); }

__global__ __launch_bounds__(256) void ls_gemv_n(const LsRec* recs, const unsigned count) { LS_ENTER(LsGemvN); gemv_n_kernel_body(p.A, p.lda, p.mp, p.np, p.v, p.sa, p.sb, p.add, p.out, p.done, bx, r_.gridx); }
__global__ __launch_bounds__(256) void ls_gemv_t(const LsRec* recs, const unsigned count) { LS_ENTER(LsGemvT); gemv_t_kernel_body(p.A, p.lda, p.rows_per_chunk, p.np, p.u, p.part, p.done, bx % p.gx, bx / p.gx); }
__global__ __launch_bounds__(256) void ls_sub_partials(const LsRec* recs, const unsigned count) { LS_ENTER(LsSubPart); sub_partials_kernel_body(p.z, p.part, p.np, p.rc, p.done, bx, r_.gridx); }
__global__ __launch_bounds__(256) void ls_group_diag_t(const LsRec* recs, const unsigned count) {      // block (32, 8); grid (4, 4, blocks) packed
    LS_ENTER(LsGroupDiagT);
    group_diag_transpose_kernel_body(p.invD, p.XT, p.X, p.b0, p.GS, p.done, bx & 3u, (bx >> 2) & 3u, bx >> 4);
}
// the batched form of the NT contraction (group inverses: blockIdx.y = pair, blockIdx.z = group), its 3-D grid packed
__global__ __launch_bounds__(256, 2) void ls_gemm_32_batched(const LsRec* recs, const unsigned count) {
    LS_ENTER(GemmNT);
    if (p.done && *p.done) return;
    __shared__ __attribute__((aligned(16))) double lds[2 * (32 + 32) * (32 + 2)];
    const unsigned gx = (unsigned)p.n_direct;
    gemm_nt_body<32, 32, 32, 2, 2, false>(p, (int)(bx % gx), (int)((bx / gx) % (unsigned)p.batch), (int)(bx / (gx * (unsigned)p.batch)), lds);
}

// launch one global step: `count` (<= LS_MAX_GROUP) LPs, `blocks` = the sum of their grids, `lds` = the largest dynamic LDS among them
inline hipError_t ls_launch(int type, const LsRec* d_recs, unsigned count, unsigned blocks, unsigned lds, hipStream_t st) {
    const dim3 g(blocks);
    switch (type) {
        case LS_SPMV_CSR: hipLaunchKernelGGL(ls_spmv_csr, g, dim3(256), 0, st, d_recs, count); break;
        case LS_SPMV_CSC_T: hipLaunchKernelGGL(ls_spmv_csc_t, g, dim3(256), 0, st, d_recs, count); break;
        case LS_PREPARE: hipLaunchKernelGGL(ls_prepare, g, dim3(VBLK), 0, st, d_recs, count); break;
        case LS_STOP_TEST: hipLaunchKernelGGL(ls_stop_test, g, dim3(64), 0, st, d_recs, count); break;
        case LS_ZERO: hipLaunchKernelGGL(ls_zero, g, dim3(256), 0, st, d_recs, count); break;
        case LS_ADAT_LIST: hipLaunchKernelGGL(ls_adat_list, g, dim3(256), 0, st, d_recs, count); break;
        case LS_ADAT_SPARSE: hipLaunchKernelGGL(ls_adat_sparse, g, dim3(256), lds, st, d_recs, count); break;
        case LS_ADAT_SPARSE_GLOBAL: hipLaunchKernelGGL(ls_adat_sparse_global, g, dim3(256), 0, st, d_recs, count); break;
        case LS_MAXDIAG: hipLaunchKernelGGL(ls_maxdiag, g, dim3(256), 0, st, d_recs, count); break;
        case LS_POTRF: hipLaunchKernelGGL(ls_potrf, g, dim3(PD_THREADS), 0, st, d_recs, count); break;
        case LS_GEMM_32_128_32: hipLaunchKernelGGL((ls_gemm<32, 128, 32, 1, 8>), g, dim3(512), 0, st, d_recs, count); break;
        case LS_GEMM_64_64_16: hipLaunchKernelGGL((ls_gemm<64, 64, 16, 2, 2>), g, dim3(256), 0, st, d_recs, count); break;
        case LS_GEMM_64_128_16: hipLaunchKernelGGL((ls_gemm<64, 128, 16, 2, 2>), g, dim3(256), 0, st, d_recs, count); break;
        case LS_GEMM_128_128_16: hipLaunchKernelGGL((ls_gemm<128, 128, 16, 2, 2>), g, dim3(256), 0, st, d_recs, count); break;
        case LS_GEMM_32_32_32: hipLaunchKernelGGL((ls_gemm<32, 32, 32, 2, 2>), g, dim3(256), 0, st, d_recs, count); break;
        case LS_CHOL_UPDATE: hipLaunchKernelGGL(ls_chol_update, g, dim3(256), 0, st, d_recs, count); break;
        case LS_TRSV_FWD: hipLaunchKernelGGL(ls_trsv_fwd, g, dim3(256), 0, st, d_recs, count); break;
        case LS_TRSV_BWD: hipLaunchKernelGGL(ls_trsv_bwd, g, dim3(256), 0, st, d_recs, count); break;
        case LS_DIRECTION: hipLaunchKernelGGL(ls_direction, g, dim3(VBLK), 0, st, d_recs, count); break;
        case LS_MU_AFF: hipLaunchKernelGGL(ls_mu_aff, g, dim3(VBLK), 0, st, d_recs, count); break;
        case LS_CORR_RHS: hipLaunchKernelGGL(ls_corr_rhs, g, dim3(VBLK), 0, st, d_recs, count); break;
        case LS_UPDATE: hipLaunchKernelGGL(ls_update, g, dim3(VBLK), 0, st, d_recs, count); break;
        case LS_GEMV_N: hipLaunchKernelGGL(ls_gemv_n, g, dim3(256), 0, st, d_recs, count); break;
        case LS_GEMV_T: hipLaunchKernelGGL(ls_gemv_t, g, dim3(256), 0, st, d_recs, count); break;
        case LS_SUB_PARTIALS: hipLaunchKernelGGL(ls_sub_partials, g, dim3(256), 0, st, d_recs, count); break;
        case LS_GROUP_DIAG_T: hipLaunchKernelGGL(ls_group_diag_t, g, dim3(32, 8), 0, st, d_recs, count); break;
        case LS_GEMM_32_32_32_BATCHED: hipLaunchKernelGGL(ls_gemm_32_batched, g, dim3(256), 0, st, d_recs, count); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ipm
