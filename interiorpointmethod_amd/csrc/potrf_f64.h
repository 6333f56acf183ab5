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
    double shift_rel;            // Tikhonov shift relative to max diag(B), added to the diagonal at load
    int* fixed;                  // device counter of guarded pivots (accumulates)
    const int* done;
    long long* stamps;           // diagnostic build only (STAMP = true): s_memtime per phase and wave
    // Device-side hand-offs of the fused formation + factorization (form_factor.h), all optional (null / 0):
    // the launch polls *wait_on >= wait_count from one lane before it reads the block (bounded spin, agent acquire) and
    // bumps *signal once, behind an agent-scope release, after L_kk and inv(L_kk) are written.
    const unsigned* wait_on; unsigned wait_count;
    unsigned* signal;
    unsigned* timeout;
    unsigned* dbg; unsigned dbg_tag;   // diagnostic (may be null): see GemmNT::dbg; kind 7
    int nt;                            // 16-wide panels to factor (1 .. 8): the rows from 16 nt on are PADDING rows (unit diagonal, nothing else):
                                       // L and inv(L) are the identity there, exactly what factoring them gives, without the pivots
    long long* trace;                  // diagnostic (may be null): wall_clock64 at {start, inputs ready, done}
};

// sqrt(p) and 1/sqrt(p) from v_rsq_f64 (about 23 good bits) and one Halley step
// (y1 = y0 (1 + r (1/2 + 3/8 r)), r = 1 - p y0^2: cubic, 23 -> 69 bits), i.e. a 7-deep
// dependent chain instead of the ~45 instructions of the IEEE sqrt + divide sequences.  The
// pivot chain (128 sequential pivots per block, 4096 per factorization) is the serial
// bottleneck of the whole Cholesky, so its depth matters more than anything else here.
__device__ __forceinline__ void sqrt_rsqrt(double p, double& root, double& rinv) {
    double y = __builtin_amdgcn_rsq(p);
    double t = p * y;
    double r = __builtin_fma(-t, y, 1.0);
    double c = __builtin_fma(r, 0.375, 0.5);
    double e = r * c;
    y = __builtin_fma(y, e, y);
    rinv = y;
    double g = p * y;
    double d = __builtin_fma(-g, g, p);          // one correction step for the root itself
    root = __builtin_fma(d, 0.5 * y, g);
}

// Workspace layout in LDS (one array, 128 x 130 doubles = 133 KB):
//   L[i][j]   (j <= i)  at W[i*WLD + j]
//   X[i][j]   (j <= i)  at W[j*WLD + i + 1]      X = inv(L), stored transposed one column right
// so the strict upper part of the square holds the inverse without a second array.
//
// The schedule of a factorization (8 waves, two barriers per 16-wide panel) is documented at potrf_lds() below.

// 1/p from v_rcp_f64 (about 23 good bits) and one cubic step y(1 + e + e^2), e = 1 - p y.
__device__ __forceinline__ double fast_rcp(double p) {
    double y = __builtin_amdgcn_rcp(p);
    double e = __builtin_fma(-p, y, 1.0);
    double t = __builtin_fma(e, e, e);
    return __builtin_fma(y, t, y);
}

// value of lane (row-of-16, k) for every lane of the same 16-lane row: v_mov_b32_dpp row_newbcast:k
// (k must be a compile-time constant after unrolling).
__device__ __forceinline__ double row_bcast(double v, int k) {
    union { double d; int i[2]; } u, r;
    u.d = v; r.d = 0.0;
    switch (k) {
#define IPM_RB(K) case K: r.i[0] = __builtin_amdgcn_mov_dpp(u.i[0], 0x150 + K, 0xf, 0xf, true); \
                          r.i[1] = __builtin_amdgcn_mov_dpp(u.i[1], 0x150 + K, 0xf, 0xf, true); break;
        IPM_RB(0) IPM_RB(1) IPM_RB(2) IPM_RB(3) IPM_RB(4) IPM_RB(5) IPM_RB(6) IPM_RB(7)
        IPM_RB(8) IPM_RB(9) IPM_RB(10) IPM_RB(11) IPM_RB(12) IPM_RB(13) IPM_RB(14) IPM_RB(15)
#undef IPM_RB
    }
    return r.d;
}

// Factor the 16 x 16 tile at (c0,c0) in place.  Wave-level, all 64 lanes in a 2-D layout:
// lane (q = lane>>4, c = lane&15) holds T[q+4a][c], a = 0..3, of the FULL symmetric tile.
// Pivot step j needs, per lane, the column entries T[r][j] of its four rows (one DPP row broadcast
// each: they sit in lane (q, j)) and the row entry T[j][c] = T[c][j] of its column (one 64-lane
// shuffle from lane (j&3, c)); only 1/p_j is on the dependent chain because the elimination keeps
// UNSCALED columns (T[r][j] = L[r][j] L[j][j]):  T[r][c] -= T[r][j] T[j][c] / p_j,  r, c > j.
// About 25 instructions per pivot instead of ~65 for the one-lane-per-row / v_readlane form.
// `pre` (optional): the tile already sits in registers in exactly this layout -- the MFMA accumulator of the update that
// produced it (register a of lane l is row (l>>4)+4a, column l&15) -- and the LDS round trip is skipped.
__device__ __forceinline__ int factor_tile(double* W, int c0, int lane, double thresh, double big, double* dinv_s,
                                           const f64x4* pre = nullptr) {
    const int q = lane >> 4, c = lane & 15;
    double t[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) t[a] = pre ? (*pre)[a] : W[(c0 + q + 4 * a) * WLD + c0 + c];
    double myd = 0.0;                                        // 1/sqrt(p_c) of this lane's column
    int nfix = 0;
    // The 64-lane shuffle that fetches row j (T[j][c]) is taken off the pivot chain: row j+1 is shuffled
    // BEFORE pivot j's update touches it and then corrected with one FMA (its own update by pivot j is
    // T[j+1][c] -= T[j+1][j] * T[j][c]/p_j with a wave-uniform T[j+1][j]).
    double rowcur = __shfl(t[0], c, 64);                     // row 0: T[0][c]
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int ja = j >> 2, jq = j & 3;
        double p = readlane_f64(t[ja], 16 * jq + j);         // wave-uniform pivot
        if (!(p > thresh)) { p = big; ++nfix; if (lane == 16 * jq + j) t[ja] = big; }
        double rownext = 0.0, mnext = 0.0;
        if (j + 1 < 16) {
            const int na = (j + 1) >> 2, nq = (j + 1) & 3;
            rownext = __shfl(t[na], 16 * nq + c, 64);        // T[j+1][c] before pivot j's update
            mnext = readlane_f64(t[na], 16 * nq + j);        // T[j+1][j]  (column j is final)
        }
        const double rowcur_m = (c > j) ? rowcur : 0.0;      // columns <= j are final: masked BEFORE 1/p is known,
        const double rp = fast_rcp(p);                       // so no select sits behind the reciprocal chain
        myd = (c == j) ? p : myd;                            // remember this column's pivot; 1/sqrt after the loop
        const double rowc = rowcur * rp;                     // T[j][c] / p_j
        const double rowm = rowcur_m * rp;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (a < ja) continue;                            // rows q+4a <= j for every lane: nothing to do (static)
            double colr = row_bcast(t[a], j);                // T[q+4a][j]
            if (a == ja) colr = (q > jq) ? colr : 0.0;       // the only row group that straddles the pivot
            t[a] = __builtin_fma(-colr, rowm, t[a]);
        }
        rowcur = __builtin_fma(-mnext, rowc, rownext);       // row j+1 after pivot j (valid for c > j)
    }
    {   // all 16 reciprocal square roots at once (lane c holds p_c): one vector chain instead of 16 scalar ones
        double root;
        sqrt_rsqrt(myd, root, myd);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int r = q + 4 * a;
        if (c <= r) W[(c0 + r) * WLD + c0 + c] = t[a] * myd;        // L[r][c] = T[r][c] / sqrt(p_c)
    }
    if (q == 0) dinv_s[c0 + c] = myd;
    return nfix;
}

// X = inv(T) for the factored tile at (c0,c0).  Same 2-D layout: lane (q, i) holds X[q+4a][i]
// (column i) and L[q+4a][i]; column sweep k: x_k of every column comes from lane (k&3, i) by a
// shuffle, L[r][k] from lane (q, k) by a DPP row broadcast.
__device__ __forceinline__ void invert_tile(double* W, int c0, int lane, const double* dinv_s) {
    const int q = lane >> 4, i = lane & 15;
    double x[4], tl[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int r = q + 4 * a;
        x[a] = (r == i) ? 1.0 : 0.0;
        tl[a] = W[(c0 + r) * WLD + c0 + i];
    }
    // same trick as factor_tile: row k+1 of X is shuffled before step k updates it and corrected with one
    // FMA (X[k+1][i] -= L[k+1][k] x_k, L[k+1][k] wave-uniform), so no shuffle sits on the dependent chain
    double xcur = __shfl(x[0], i, 64);                       // X_pre[0][i]
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int ka = k >> 2, kq = k & 3;
        const double xk = xcur * dinv_s[c0 + k];             // X[k][i], final
        if (q == kq) x[ka] = xk;
        double xnext = 0.0, lnext = 0.0;
        if (k + 1 < 16) {
            const int na = (k + 1) >> 2, nq = (k + 1) & 3;
            xnext = __shfl(x[na], 16 * nq + i, 64);          // X_pre[k+1][i] before step k's update
            lnext = readlane_f64(tl[na], 16 * nq + k);       // L[k+1][k]
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (a < ka) continue;                            // rows <= k for every lane (static)
            double lrk = row_bcast(tl[a], k);                // L[q+4a][k]
            if (a == ka) lrk = (q > kq) ? lrk : 0.0;
            x[a] = __builtin_fma(-lrk, xk, x[a]);
        }
        xcur = __builtin_fma(-lnext, xk, xnext);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int r = q + 4 * a;
        if (r >= i) W[(c0 + i) * WLD + c0 + r + 1] = x[a];  // X[r][i] -> row c0+i, col c0+r+1
    }
}

// Four rows of the panel below tile (c0,c0) per wave: lane (q, c) owns element c of row r0+q and keeps
// row c of the factored tile in registers with the entries k >= c ZEROED (trowm), so the column sweep is
// branch- and select-free: x_k = p_k / L[k][k] comes from lane (q,k) by one DPP row broadcast, and
// p_c -= x_k L[c][k] is a plain FMA for every lane (a no-op where k >= c).  p stays unscaled until the end.
// 4 VALU operations per step (scale, two DPP moves, FMA): a sweep is ISSUE bound, which is why running a wave's groups through
// it together instead of one after the other changes nothing (measured in rounds 3 and 4).  Wave-level.
__device__ __forceinline__ void substitute_rows4(double* W, int c0, int r0, int lane, const double* trowm, double dc) {
    const int q = lane >> 4, c = lane & 15;
    double p = W[(r0 + q) * WLD + c0 + c];
#pragma unroll
    for (int k = 0; k < 15; ++k) {                           // column 15 has nothing to its right
        const double xk = row_bcast(p * dc, k);
        p = __builtin_fma(-xk, trowm[k], p);
    }
    W[(r0 + q) * WLD + c0 + c] = p * dc;
}

// T(r0,q0) - L(r0, c0:c0+16) L(q0, c0:c0+16)^T kept in registers (accumulator layout), not written back.
__device__ __forceinline__ f64x4 update_tile_regs(const double* W, int c0, int r0, int q0, int fr, int fk) {
    f64x4 acc0, acc1 = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q) acc0[q] = W[(r0 + fk + 4 * q) * WLD + q0 + fr];
    double av[4], bv[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        av[kk] = -W[(r0 + fr) * WLD + c0 + kk * 4 + fk];
        bv[kk] = W[(q0 + fr) * WLD + c0 + kk * 4 + fk];
    }
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc1, 0, 0, 0);
    return acc0 + acc1;
}

// T(r0,q0) -= L(r0, c0:c0+16) L(q0, c0:c0+16)^T  (16 x 16 tiles, MFMA).  Wave-level.
__device__ __forceinline__ void update_tile(double* W, int c0, int r0, int q0, int fr, int fk) {
    f64x4 acc0, acc1 = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q) acc0[q] = W[(r0 + fk + 4 * q) * WLD + q0 + fr];
    double av[4], bv[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        av[kk] = -W[(r0 + fr) * WLD + c0 + kk * 4 + fk];
        bv[kk] = W[(q0 + fr) * WLD + c0 + kk * 4 + fk];
    }
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc1, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) W[(r0 + fk + 4 * q) * WLD + q0 + fr] = acc0[q] + acc1[q];
}

// X(ib,jt) = -X(ib,ib) * sum_{k=jt}^{ib-1} L(ib,k) X(k,jt)  (tile indices).  Wave-level, in parts: the sum S over a range of k
// (needs block row ib of L and the inverse rows kb of that range) with its two accumulators carried in registers, so a sum can be
// started before the last inverse row above ib exists and continued behind a barrier with the SAME order of additions; then the
// product with the diagonal tile's inverse (needs X(ib,ib)).
struct InvSum { f64x4 s0, s1; };
__device__ __forceinline__ InvSum inv_sum_zero() { InvSum z; z.s0 = (f64x4){0.0, 0.0, 0.0, 0.0}; z.s1 = z.s0; return z; }
__device__ __forceinline__ void inverse_tile_accum(const double* W, int ib, int jt, int kb0, int kb1, int fr, int fk, InvSum& st) {
    for (int kb = kb0; kb < kb1; ++kb) {
        double av[4], bv[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            av[kk] = W[(ib * 16 + fr) * WLD + kb * 16 + kk * 4 + fk];                   // L(ib,kb)[fr][k]
            int kr = kb * 16 + kk * 4 + fk, cc = jt * 16 + fr;                          // X[kr][cc]
            bv[kk] = (kr >= cc) ? W[cc * WLD + kr + 1] : 0.0;
        }
        st.s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], st.s0, 0, 0, 0);
        st.s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], st.s1, 0, 0, 0);
        st.s0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], st.s0, 0, 0, 0);
        st.s1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], st.s1, 0, 0, 0);
    }
}
__device__ __forceinline__ f64x4 inverse_tile_sum(const double* W, int ib, int jt, int fr, int fk) {
    InvSum st = inv_sum_zero();
    inverse_tile_accum(W, ib, jt, jt, ib, fr, fk, st);
    return st.s0 + st.s1;
}
__device__ __forceinline__ void inverse_tile_finish(double* W, int ib, int jt, int fr, int fk, const f64x4 s) {
    // second product: accumulator register q of S is row fk+4q, so it pairs with X(ib,ib)[fr][fk+4q]
    double xa[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int k = fk + 4 * q;
        xa[q] = (k <= fr) ? -W[(ib * 16 + k) * WLD + ib * 16 + fr + 1] : 0.0;
    }
    f64x4 r0 = (f64x4){0.0, 0.0, 0.0, 0.0}, r1 = r0;
    r0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[0], s[0], r0, 0, 0, 0);
    r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[1], s[1], r1, 0, 0, 0);
    r0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[2], s[2], r0, 0, 0, 0);
    r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[3], s[3], r1, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)                                                          // X(ib,jt)[fk+4q][fr]
        W[(jt * 16 + fr) * WLD + ib * 16 + fk + 4 * q + 1] = r0[q] + r1[q];
}
__device__ __forceinline__ void inverse_tile(double* W, int ib, int jt, int fr, int fk) {
    inverse_tile_finish(W, ib, jt, fr, fk, inverse_tile_sum(W, ib, jt, fr, fk));
}

// STAMP = true is a diagnostic instantiation (tools/potrf_stamps.py): every wave records s_memtime
// at each phase boundary into stamps[wave*64 + slot]; the production kernel carries no stamps.
#define IPM_STAMP(slot) do { if (STAMP && lane == 0) stamps[wave * 64 + (slot)] = (long long)clock64(); } while (0)
constexpr int PD_THREADS = 512;      // 8 waves: wave 0 runs the serial pivot chain, waves 1..7 the MFMA work

// Guarded Cholesky of the leading 16 nt x 16 nt block held in the LDS workspace W (layout above), by one workgroup of
// PD_THREADS threads: on return L sits in the lower triangle and X = inv(L) in the shifted upper triangle, dinv_s[i] =
// 1 / L[i][i].  W must hold the lower triangle AND the full symmetric 16 x 16 diagonal tiles on entry; all waves must
// have passed a barrier after the last write to W.  Returns the number of guarded pivots (meaningful on wave 0).
// nt = 8 is the 128 x 128 diagonal block of the blocked factorization (potrf_diag_kernel); the fused small-LP kernel
// (small_lp.h) calls it with nt = ceil(m / 16).
// Hooks of potrf_lds for work that only READS what is already final (the write-back of potrf_diag_body): phase(jb, wave) is called in
// P3(jb) by every wave that is not on the pivot chain, behind its items (the tile rows up to jb of L and up to jb-2 of the inverse
// are final there); last() by every wave behind the products of the last inverse row (tile row nt-2 of the inverse is final there).
struct NoEarlyWork {
    __device__ __forceinline__ void phase(int, int) const {}
    __device__ __forceinline__ void last(int) const {}
};

// tile0_done: the caller has factored tile (0,0) itself (factor_tile with `pre`) and every wave has passed a barrier since.
template <bool STAMP, typename Early = NoEarlyWork>
__device__ __forceinline__ int potrf_lds(double* W, double* dinv_s, int nt, double thresh, double big, long long* stamps,
                                         Early early = Early(), bool tile0_done = false) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    int nfix = 0;
    if (!tile0_done) {
        if (wave == 0) nfix += factor_tile(W, 0, lane, thresh, big, dinv_s);
        IPM_STAMP(2);
        __syncthreads();
    }
    IPM_STAMP(3);

    // Phase schedule per 16-wide panel jb (two barriers per panel):
    //   P2(jb): forward substitution of the panel rows below tile jb, four rows per wave pass (DPP form)
    //   P3(jb): wave 0 : update tile (jb+1,jb+1) and factor it (runs ahead on the serial pivot chain)
    //           wave 7 : invert tile jb, then shares the item list
    //           waves 1..6: item list = rest of the trailing update of panel jb, then block row jb-1 of
    //                       inv(L) (its diagonal tile inverse was produced in P3(jb-1))
    InvSum sumA = inv_sum_zero();                          // last panel: the sum of this wave's tile of the last inverse row
    for (int jb = 0; jb < nt; ++jb) {
        const int c0 = jb * 16;
        const int nrt = nt - jb - 1;                      // 16-row tiles below the pivot tile
        if (nrt > 0) {
            double trow[16];
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                f64x2 v = *reinterpret_cast<const f64x2*>(&W[(c0 + fr) * WLD + c0 + k]);
                trow[k] = (k < fr) ? v.x : 0.0; trow[k + 1] = (k + 1 < fr) ? v.y : 0.0;      // strictly lower part of row fr
            }
            const double dc = dinv_s[c0 + fr];
            for (int g = wave; g < 4 * nrt; g += 8) substitute_rows4(W, c0, c0 + 16 + 4 * g, lane, trow, dc);
        }
        IPM_STAMP(4 + jb * 4);
        __syncthreads();
        IPM_STAMP(5 + jb * 4);
        if (wave == 0 && nrt > 0) {
            const f64x4 nxt = update_tile_regs(W, c0, c0 + 16, c0 + 16, fr, fk);     // stays in registers
            nfix += factor_tile(W, c0 + 16, lane, thresh, big, dinv_s, &nxt);
        } else {
            // items: update tiles 1..ntile-1 of panel jb, then tiles 0..jb-2 of inverse row jb-1.
            // Wave 7 only inverts tile jb (about as long as wave 0's factorization) while a pivot tile is
            // left; the item list is shared by waves 1..6.
            if (nrt > 0) {
                const int ntile = nrt * (nrt + 1) / 2;
                const int nupd = ntile - 1;
                const int ninv = jb >= 1 ? jb - 1 : 0;
                int me = wave - 1;
                if (wave == 7) { invert_tile(W, c0, lane, dinv_s); me = nupd + ninv; }
                for (int it = me; it < nupd + ninv; it += 6) {
                    if (it < nupd) {
                        int tix = it + 1;
                        int ib = (int)((sqrtf(8.0f * (float)tix + 1.0f) - 1.0f) * 0.5f);
                        while ((ib + 1) * (ib + 2) / 2 <= tix) ++ib;
                        while (ib * (ib + 1) / 2 > tix) --ib;
                        int cb = tix - ib * (ib + 1) / 2;
                        update_tile(W, c0, (jb + 1 + ib) * 16, (jb + 1 + cb) * 16, fr, fk);
                    } else {
                        inverse_tile(W, jb - 1, it - nupd, fr, fk);
                    }
                }
                early.phase(jb, wave);
            } else {
                // LAST panel: nothing is left on the pivot chain but the inversion of the last diagonal tile (wave 7), and the last
                // TWO block rows of inv(L) are still to come -- tile (nt-2, jt) costs nt-2-jt products, the sum of tile (nt-1, jt)
                // one more, of which only the last needs row nt-2.  So both rows run here, balanced: with n = nt-2, wave w < n makes
                // tile (nt-2, w) (n-w products) and the sum of tile (nt-1, n-1-w) up to row nt-3 (w+1 products), kept in registers;
                // behind the barrier every sum takes its last term and the product with the tile inverse, one tile per wave.
                // (One tile per wave in column order, first row nt-2 then row nt-1: 13 products on wave 0, 1 on wave 6, and an fp64
                // MFMA is 64 cycles on a SIMD that two waves share: 11.5k + 6k cycles for the two rows, now 7 products per wave.)
                if (wave == 7) invert_tile(W, c0, lane, dinv_s);
                const int n = nt - 2;
                if (wave < n) {
                    inverse_tile(W, nt - 2, wave, fr, fk);
                    inverse_tile_accum(W, nt - 1, n - 1 - wave, n - 1 - wave, nt - 2, fr, fk, sumA);
                }
                early.phase(jb, wave);
            }
        }
        IPM_STAMP(6 + jb * 4);
        __syncthreads();
        IPM_STAMP(7 + jb * 4);
    }
    // last block row of inv(L): the sums (in registers, above) take their last term -- row nt-2 is complete behind the barrier --
    // and the product with the last tile's inverse: X(nt-1, jt) = -X(nt-1, nt-1) S.
    // early.last() (optional): work that only READS what is already final runs here, per wave, behind that.
    {
        const int n = nt - 2;
        if (wave < n) {
            inverse_tile_accum(W, nt - 1, n - 1 - wave, nt - 2, nt - 1, fr, fk, sumA);
            inverse_tile_finish(W, nt - 1, n - 1 - wave, fr, fk, sumA.s0 + sumA.s1);
        } else if (wave == n) {
            inverse_tile(W, nt - 1, nt - 2, fr, fk);          // (one term: the tile inverse of row nt-2 is from the panel before)
        }
    }
    early.last(wave);
    IPM_STAMP(38);
    __syncthreads();
    IPM_STAMP(39);
    return nfix;
}

// One diagonal block: wait for it (optional), factor it in the LDS workspace W, write L_kk and inv(L_kk), signal (optional).
// The body of potrf_diag_kernel and of every step of ff_chain_kernel (form_factor.h).
template <bool STAMP>
__device__ __forceinline__ void potrf_diag_body(const PotrfDiag& a, double* W, double* dinv_s) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.trace && tid == 0) a.trace[0] = (long long)wall_clock64();
    if (a.wait_on) {
        if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(a.wait_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.wait_count) {
                __builtin_amdgcn_s_sleep(2);
                ++spins;
                if (spins > ipm_spin_limit || ((spins & 1023u) == 1u && a.timeout &&
                                           __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    if (spins > ipm_spin_limit && a.dbg && __hip_atomic_fetch_add(a.dbg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                        a.dbg[1] = a.dbg_tag; a.dbg[2] = 7u; a.dbg[3] = a.wait_count;
                        a.dbg[4] = __hip_atomic_load(a.wait_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (a.timeout) __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    const double thresh = a.eps * (*a.maxdiag);
    long long* stamps = a.stamps;
    if (a.trace && tid == 0) a.trace[1] = (long long)wall_clock64();

    IPM_STAMP(0);
    // ---- load the block (rows complete up to the end of their 16-wide diagonal tile): all loads of a thread are issued before the
    //      first LDS write (one memory latency).  Tile (0,0) goes straight into the registers of wave 0 in factor_tile's layout and
    //      is factored from there while the other waves are still storing the rows below it: the first 16 pivots of the chain run
    //      under the load instead of behind it.
    const int nt = a.nt >= 1 && a.nt <= NB / 16 ? a.nt : NB / 16;
    int nfix = 0;
    {
        f64x4 t0 = (f64x4){0.0, 0.0, 0.0, 0.0};
        if (wave == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) t0[q] = a.Bkk[(int64_t)((lane >> 4) + 4 * q) * a.ld + (lane & 15)];
        }
        f64x2 v[14];
#pragma unroll
        for (int u = 0; u < 14; ++u) {
            int idx = tid + (u + 2) * PD_THREADS;                   // rows 16 .. 127
            int i = idx >> 6, c2 = (idx & 63) * 2;
            v[u] = (c2 <= (i | 15)) ? *reinterpret_cast<const f64x2*>(a.Bkk + (int64_t)i * a.ld + c2) : (f64x2){0.0, 0.0};
        }
        if (wave == 0) {
            if (a.shift_rel != 0.0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) if ((lane >> 4) + 4 * q == (lane & 15)) t0[q] += a.shift_rel * (*a.maxdiag);
            }
            nfix = factor_tile(W, 0, lane, thresh, a.big, dinv_s, &t0);
        }
#pragma unroll
        for (int u = 0; u < 14; ++u) {
            int idx = tid + (u + 2) * PD_THREADS;
            int i = idx >> 6, c2 = (idx & 63) * 2;
            if (c2 <= (i | 15)) *reinterpret_cast<f64x2*>(&W[i * WLD + c2]) = v[u];
        }
    }
    IPM_STAMP(2);
    if (nt < NB / 16) {
        // inverse rows of the padding part: X[i][j] = (i == j), stored at W[j][i + 1] (columns beyond the factored block)
        const int r0 = 16 * nt, nr = NB - r0;
        __syncthreads();      // the block load above also stores W[j][j+1..(j|15)] of the padding rows: the fill must come second
        for (int idx = tid; idx < nr * NB; idx += PD_THREADS) {
            const int i = r0 + idx / NB, j = idx % NB;
            if (j <= i) W[j * WLD + i + 1] = (i == j) ? 1.0 : 0.0;
        }
    }
    if (a.shift_rel != 0.0) {
        __syncthreads();
        if (tid >= 16 && tid < 16 * nt) W[tid * WLD + tid] += a.shift_rel * (*a.maxdiag);
    }
    __syncthreads();
    IPM_STAMP(1);

    // ---- write back: L (lower) to B; inverse (lower triangle; above the diagonal only the zeros inside its 16 x 16 diagonal tiles
    //      are written, the rest of the strict upper triangle of `inv` stays zero from the handle's initial memset).
    //      One CU retires a global store INSTRUCTION every ~70 cycles whatever its width or mask (measured: 128 row-wise stores of
    //      L from one wave 20k cycles; round-4 stamps), so the write-back is packed into as few full-wave 16-byte stores as the
    //      triangle allows -- the short rows of L two, four and eight to an instruction (87 instead of ~190), the inverse two
    //      instructions per 16 x 16 tile (72 instead of 144) -- and spread over the phases in which the waves beside the pivot chain
    //      have slack: P3(jb), jb >= 3, takes tile row jb of L (the first one all rows before it too) and tile row jb-2 of the
    //      inverse, dealt round-robin to waves 5..7; the last panel, where waves 0..5 make the last two inverse rows, the rest on
    //      waves 6 and 7; tile row nt-2 of the inverse behind the last row's products, tile row nt-1 at the end.  (Before: all of
    //      it behind the last inverse row, 13.5k cycles after the last panel of 88k in all.)
    struct Hooks {
        const PotrfDiag& a; double* W; int nt, lane; unsigned skip;
        // R = 8, 4, 2, 1 rows of L from row r in one instruction (64/R lanes per row, two columns per lane); the diagonal entry of
        // an even row has no partner inside the triangle and is left to l_diag()
        __device__ __forceinline__ void l_instr(int r, int lg) const {
            if (skip & 1u) return;
            const int lpr = 64 >> lg, i = r + (lane >> (6 - lg)), j = 2 * (lane & (lpr - 1));
            const f64x2 v = *reinterpret_cast<const f64x2*>(&W[i * WLD + j]);
            if (j + 1 <= i) *reinterpret_cast<f64x2*>(a.Bkk + (int64_t)i * a.ld + j) = v;
        }
        // two of them with both LDS reads in flight before the first store (a unit alone is ~300 cycles of latency on its wave)
        __device__ __forceinline__ void l_instr2(int r0, int r1, int lg) const {
            if (skip & 1u) return;
            const int lpr = 64 >> lg, sub = lane >> (6 - lg), j = 2 * (lane & (lpr - 1));
            const int i0 = r0 + sub, i1 = r1 + sub;
            const f64x2 v0 = *reinterpret_cast<const f64x2*>(&W[i0 * WLD + j]);
            const f64x2 v1 = *reinterpret_cast<const f64x2*>(&W[i1 * WLD + j]);
            if (j + 1 <= i0) *reinterpret_cast<f64x2*>(a.Bkk + (int64_t)i0 * a.ld + j) = v0;
            if (j + 1 <= i1) *reinterpret_cast<f64x2*>(a.Bkk + (int64_t)i1 * a.ld + j) = v1;
        }
        __device__ __forceinline__ void l_diag() const {
            if (skip & 1u) return;
            const int i = 2 * lane;
            a.Bkk[(int64_t)i * a.ld + i] = W[i * WLD + i];
        }
        // rows 8 s .. 8 s + 7 of the 16 x 16 tile (ib, jt) of the inverse: X[i][j] sits TRANSPOSED at W[j][i+1]; lane (ri, cj) takes row
        // ri, columns 2 cj and 2 cj + 1: LDS banks 8 cj + 2 ri (two lanes per bank pair, the minimum for 8-byte accesses)
        __device__ __forceinline__ void inv_half(int ib, int jt, int sh) const {
            if (skip & 2u) return;
            const int i = 16 * ib + 8 * sh + (lane >> 3), j = 16 * jt + 2 * (lane & 7);
            double x0 = W[j * WLD + i + 1], x1 = W[(j + 1) * WLD + i + 1];
            if (ib == jt) { x0 = (j <= i) ? x0 : 0.0; x1 = (j + 1 <= i) ? x1 : 0.0; }
            *reinterpret_cast<f64x2*>(a.inv + i * NB + j) = (f64x2){x0, x1};
        }
        __device__ __forceinline__ void inv_half2(int ib, int u0, int u1) const {      // units u = 2 jt + half of tile row ib
            if (skip & 2u) return;
            const int i0 = 16 * ib + 8 * (u0 & 1) + (lane >> 3), j0 = 16 * (u0 >> 1) + 2 * (lane & 7);
            const int i1 = 16 * ib + 8 * (u1 & 1) + (lane >> 3), j1 = 16 * (u1 >> 1) + 2 * (lane & 7);
            double x0 = W[j0 * WLD + i0 + 1], x1 = W[(j0 + 1) * WLD + i0 + 1];
            double y0 = W[j1 * WLD + i1 + 1], y1 = W[(j1 + 1) * WLD + i1 + 1];
            x0 = (j0 <= i0) ? x0 : 0.0; x1 = (j0 + 1 <= i0) ? x1 : 0.0;                 // (only the diagonal tile has j > i)
            y0 = (j1 <= i1) ? y0 : 0.0; y1 = (j1 + 1 <= i1) ? y1 : 0.0;
            *reinterpret_cast<f64x2*>(a.inv + i0 * NB + j0) = (f64x2){x0, x1};
            *reinterpret_cast<f64x2*>(a.inv + i1 * NB + j1) = (f64x2){y0, y1};
        }
        // The instructions of a phase are dealt round-robin to its nw waves, segment by segment; wave `me` steps straight through
        // ITS units (u = first, first + nw, ...): a wave issues an instruction every ~5 cycles, so a loop over all units of a phase
        // with a test per unit costs more than the stores (measured: +4.6k cycles per phase).  `off` = units dealt so far mod nw.
        struct Deal {
            int me, nw, off;
            __device__ __forceinline__ int first() const { const int f = me - off; return f < 0 ? f + nw : f; }
            __device__ __forceinline__ void dealt(int count) {
                off += count;
                if (nw == 3) off -= 3 * ((off * 43) >> 7); else off &= nw - 1;       // nw is 3, 8 or 2; off < 128
            }
        };
        __device__ __forceinline__ void l_segment(int r_lo, int r_hi, int lg, Deal& d) const {      // rows of one packing
            if (r_hi <= r_lo) return;
            const int count = (r_hi - r_lo) >> lg;
            int u = d.first();
            for (; u + d.nw < count; u += 2 * d.nw) l_instr2(r_lo + (u << lg), r_lo + ((u + d.nw) << lg), lg);
            if (u < count) l_instr(r_lo + (u << lg), lg);
            d.dealt(count);
        }
        __device__ __forceinline__ void l_rows(int r_lo, int r_hi, Deal& d) const {                 // r_lo, r_hi: multiples of 16
            if (r_lo >= 64) { l_segment(r_lo, r_hi, 0, d); return; }
            l_segment(r_lo, r_hi < 16 ? r_hi : 16, 3, d);
            l_segment(r_lo > 16 ? r_lo : 16, r_hi < 32 ? r_hi : 32, 2, d);
            l_segment(r_lo > 32 ? r_lo : 32, r_hi < 64 ? r_hi : 64, 1, d);
            l_segment(r_lo > 64 ? r_lo : 64, r_hi, 0, d);
        }
        __device__ __forceinline__ void inv_rows(int ib_lo, int ib_hi, Deal& d) const {
            for (int ib = ib_lo; ib < ib_hi; ++ib) {
                const int count = 2 * (ib + 1);
                int u = d.first();
                for (; u + d.nw < count; u += 2 * d.nw) inv_half2(ib, u, u + d.nw);
                if (u < count) inv_half(ib, u >> 1, u & 1);
                d.dealt(count);
            }
        }
        __device__ __forceinline__ void phase(int jb, int wave) const {
            const int first = nt - 1 < 3 ? nt - 1 : 3;
            if (jb < first) return;
            const bool lastp = jb == nt - 1;
            Deal d;
            d.off = 0;
            if (!lastp) { d.me = wave - 5; d.nw = 3; }                   // waves 5, 6, 7: the ones with the shortest items in every panel
            else if (nt < NB / 16) { d.me = wave; d.nw = 8; }            // partial block: fewer products, more (padding) rows: all waves
            else { d.me = wave - 6; d.nw = 2; }                          // waves 0..5 are busy with the last two inverse rows
            if (d.me < 0) return;
            l_rows(jb == first ? 0 : 16 * jb, 16 * (jb + 1), d);
            if (lastp) {
                l_rows(16 * nt, NB, d);
                if (d.first() == 0) l_diag();
                d.dealt(1);
            }
            inv_rows(jb == first ? 0 : jb - 2, jb - 1, d);
            if (lastp) inv_rows(nt, NB / 16, d);
        }
        __device__ __forceinline__ void last(int wave) const {          // tile row nt-2 of the inverse
            Deal d{wave, 8, 0};
            if (nt >= 2) inv_rows(nt - 2, nt - 1, d);
        }
        __device__ __forceinline__ void end(int wave) const {           // tile row nt-1
            Deal d{wave, 8, 0};
            inv_rows(nt - 1, nt, d);
        }
    };
    const Hooks hooks{a, W, nt, lane, STAMP ? a.dbg_tag : 0u};      // (skip: timing experiments of the diagnostic build)
    nfix += potrf_lds<STAMP>(W, dinv_s, nt, thresh, a.big, stamps, hooks, /*tile0_done=*/true);
    hooks.end(wave);
    IPM_STAMP(40);
    if (lane == 0 && wave == 0 && nfix) atomicAdd(a.fixed, nfix);
    if (a.signal) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(a.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (a.trace && tid == 0) a.trace[2] = (long long)wall_clock64();
}

template <bool STAMP>
__global__ __launch_bounds__(PD_THREADS) void potrf_diag_kernel(PotrfDiag a) {
    if (a.done && *a.done) {
        if (a.signal && threadIdx.x == 0) __hip_atomic_fetch_add(a.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __shared__ __attribute__((aligned(16))) double W[NB * WLD];
    __shared__ double dinv_s[NB];
    potrf_diag_body<STAMP>(a, W, dinv_s);
}

// max of the diagonal of an n x n matrix (single workgroup; n <= a few 10^4)
__device__ __forceinline__ void maxdiag_kernel_body(const double* B, int64_t ld, int n, double* out,
                                                      const int* done, const unsigned bx_, const unsigned gx_) {
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
__global__ __launch_bounds__(256) void maxdiag_kernel(const double* B, int64_t ld, int n, double* out,
                                                      const int* done) { maxdiag_kernel_body(B, ld, n, out, done, blockIdx.x, gridDim.x); }

// ------------------------------------------------------------------------------------------
// Triangular solves with the factor (L in the lower triangle of B, inv(L_kk) per block).
// One launch per 128-row block step; inside a launch every workgroup first forms the
// solution block z_k = inv(L_kk) r_k (a 128 x 128 GEMV, recomputed per workgroup so no
// inter-workgroup hand-off is needed) and then eliminates it from its own row block.
// Sums are in a fixed order: results are bitwise reproducible.
// ------------------------------------------------------------------------------------------

// A 128 x 128 row-major block held in the registers of one 256-thread workgroup: thread
// (rg = tid>>4, l16 = tid&15) keeps, for each of the 8 passes p, the four 16-byte chunks
// l16, l16+16, l16+32, l16+48 of row 16p+rg (so one load instruction of a wave covers four full rows).
struct BlockRegs { f64x2 m[8][4]; };

__device__ __forceinline__ void block_load(BlockRegs& R, const double* __restrict__ M, int64_t ldm) {
    const int l16 = threadIdx.x & 15, rg = threadIdx.x >> 4;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const double* row = M + (int64_t)(p * 16 + rg) * ldm;
#pragma unroll
        for (int q = 0; q < 4; ++q) R.m[p][q] = *reinterpret_cast<const f64x2*>(row + (l16 + 16 * q) * 2);
    }
}

// out[128] = M v   (v, out in LDS; fixed-order 16-lane tree reduction)
__device__ __forceinline__ void block_gemv_n(const BlockRegs& R, const double* vs, double* out) {
    const int l16 = threadIdx.x & 15, rg = threadIdx.x >> 4;
    double v[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[2 * q] = vs[(l16 + 16 * q) * 2]; v[2 * q + 1] = vs[(l16 + 16 * q) * 2 + 1]; }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) s += R.m[p][q].x * v[2 * q] + R.m[p][q].y * v[2 * q + 1];
        s += __shfl_xor(s, 8, 16);
        s += __shfl_xor(s, 4, 16);
        s += __shfl_xor(s, 2, 16);
        s += __shfl_xor(s, 1, 16);
        if (l16 == 0) out[p * 16 + rg] = s;
    }
}

// out[128] = M^T v  (v, out in LDS; `scratch` = 16*128 doubles of LDS; contains two barriers)
__device__ __forceinline__ void block_gemv_t(const BlockRegs& R, const double* vs, double* out, double* scratch) {
    const int tid = threadIdx.x;
    const int l16 = tid & 15, rg = tid >> 4;
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const double vr = vs[p * 16 + rg];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[2 * q] += R.m[p][q].x * vr;
            acc[2 * q + 1] += R.m[p][q].y * vr;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f64x2*>(&scratch[rg * NB + (l16 + 16 * q) * 2]) = (f64x2){acc[2 * q], acc[2 * q + 1]};
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
    int j0;                          // backward step: first column block with a structural nonzero in block row k
    const int* done;
};

// forward step k: z_k = inv(L_kk) r_k ; r_i -= L_ik z_k for i > k.  grid = nblk - k.
// Both 128 x 128 blocks are fetched into registers up front, so the second product does not pay a
// second memory latency after z_k is known.
__device__ __forceinline__ void trsv_fwd_step_kernel_body(TrsvStep a, const unsigned bx_, const unsigned gx_) {
    if (a.done && *a.done) return;
    __shared__ double vs[NB], zs[NB], us[NB];
    const int tid = threadIdx.x;
    const int i = a.k + bx_;
    BlockRegs RI, RL;
    block_load(RI, a.inv + (int64_t)a.k * NB * NB, NB);
    if (bx_ != 0) block_load(RL, a.L + (int64_t)i * NB * a.ld + (int64_t)a.k * NB, a.ld);
    if (tid < NB) vs[tid] = a.r[(int64_t)a.k * NB + tid];
    __syncthreads();
    block_gemv_n(RI, vs, zs);
    __syncthreads();
    if (bx_ == 0) {
        if (tid < NB) a.z[(int64_t)a.k * NB + tid] = zs[tid];
        return;
    }
    block_gemv_n(RL, zs, us);
    __syncthreads();
    if (tid < NB) a.r[(int64_t)i * NB + tid] -= us[tid];
}
__global__ __launch_bounds__(256) void trsv_fwd_step_kernel(TrsvStep a) { trsv_fwd_step_kernel_body(a, blockIdx.x, gridDim.x); }

// backward step k (descending): w_k = inv(L_kk)^T z_k ; z_j -= L_kj^T w_k for j0 <= j < k. grid = k-j0+1.
// Block j == k writes w_k to `z` (the solution); blocks j < k update the running rhs `r`.
__device__ __forceinline__ void trsv_bwd_step_kernel_body(TrsvStep a, const unsigned bx_, const unsigned gx_) {
    if (a.done && *a.done) return;
    __shared__ double vs[NB], ws[NB], us[NB];
    __shared__ __attribute__((aligned(16))) double scratch[16 * NB];
    const int tid = threadIdx.x;
    const int j = a.j0 + bx_;
    BlockRegs RI, RL;
    block_load(RI, a.inv + (int64_t)a.k * NB * NB, NB);
    if (j != a.k) block_load(RL, a.L + (int64_t)a.k * NB * a.ld + (int64_t)j * NB, a.ld);
    if (tid < NB) vs[tid] = a.r[(int64_t)a.k * NB + tid];
    __syncthreads();
    block_gemv_t(RI, vs, ws, scratch);
    if (j == a.k) {
        if (tid < NB) a.z[(int64_t)a.k * NB + tid] = ws[tid];
        return;
    }
    block_gemv_t(RL, ws, us, scratch);
    if (tid < NB) a.r[(int64_t)j * NB + tid] -= us[tid];
}
__global__ __launch_bounds__(256) void trsv_bwd_step_kernel(TrsvStep a) { trsv_bwd_step_kernel_body(a, blockIdx.x, gridDim.x); }

}  // namespace ipm
