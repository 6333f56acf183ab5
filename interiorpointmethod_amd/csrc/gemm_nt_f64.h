// gemm_nt_f64.h -- fp64 MFMA "NT" contraction for gfx950 (CDNA4, wave64).
//
//   C[M x N] = beta*C + alpha * sum_k P[i][k] * w[k] * Q[j][k]
//
// P (M x K) and Q (N x K) are row-major with K contiguous, which is exactly how the
// constraint matrix A is stored, so the three GEMM-shaped pieces of the hot path are
// one kernel:
//   * normal matrix  B = A diag(d) A^T        (P = Q = A, w = d, lower tiles only)
//   * Cholesky panel L_ik = B_ik inv(L_kk)^T  (P = panel, Q = inv(L_kk), in place)
//   * trailing update B_ij -= L_ik L_jk^T     (P = Q = panel, alpha=-1, beta=1, lower)
// It replaces scipy's two SpGEMMs at main.py:224 (reference repo) and the LU inside
// spsolve (main.py:180) for the dense-B formulation.
//
// Tiling: one workgroup = WAVES_M x WAVES_N waves, each wave owns a (BM/WAVES_M) x
// (BN/WAVES_N) sub-tile built from v_mfma_f64_16x16x4_f64 tiles.  Operand tiles are staged
// global -> registers -> LDS (16-byte loads, k contiguous), double buffered with one
// barrier per BK step; the global loads of stage t+1 are in flight while stage t is
// multiplied.  LDS rows are padded by 16 B so the MFMA fragment reads (16 rows x 2
// adjacent k per 32-lane half, ds_read_b64) are bank-conflict free: row stride
// (BK+2)*2 dwords = 36 -> banks 36r+2k mod 64 are all distinct for r<16, k<2.
//
// fp64 MFMA lane maps (gfx950): A operand lane l holds A[i=l&15][k=l>>4]; B operand lane l
// holds B[k=l>>4][j=l&15]; accumulator register q of lane l is D[row=(l>>4)+4q][col=l&15].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ipm {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct GemmNT {
    const double* P; int64_t ldp;
    const double* Q; int64_t ldq;
    const double* w;            // optional diagonal scaling along k (applied to Q); may be null
    double* C; int64_t ldc;
    int M, N, K;                // multiples of BM, BN, BK
    double alpha, beta;
    int lower;                  // 1: only tiles with tile_row >= tile_col (needs BM == BN, M == N)
    int unit_diag_from;         // >= 0: C[r][r] = 1 for r >= unit_diag_from (padding rows); -1 off
    const int* done;            // device flag: kernel is a no-op when *done != 0 (may be null)
};

// bijective XCD-aware remap of the linear workgroup id (blocks b and b+8 share an XCD, so
// give every XCD a contiguous run of tiles: neighbours in the run share operand panels in
// that XCD's private L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int q = nwg / nx, r = nwg % nx;
    int xcd = bid % nx, idx = bid / nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool SCALE>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2)
void gemm_nt_f64_kernel(GemmNT g) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int LDT = BK + 2;                 // padded LDS row (doubles)
    constexpr int CH = BK / 2;                  // 16-byte chunks per tile row
    constexpr int PL = (BM * CH) / NT;          // chunks per thread, P tile
    constexpr int QL = (BN * CH) / NT;          // chunks per thread, Q tile
    static_assert((BM * CH) % NT == 0 && (BN * CH) % NT == 0, "tile/threads mismatch");
    static_assert(WTM % 16 == 0 && WTN % 16 == 0 && BK % 4 == 0, "mfma tiling");

    if (g.done && *g.done) return;

    __shared__ __attribute__((aligned(16))) double lds[2 * (BM + BN) * LDT];
    double* Ps = lds;                           // [2][BM][LDT]
    double* Qs = lds + 2 * BM * LDT;            // [2][BN][LDT]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // ---- tile coordinates
    int ti, tj;
    {
        int bid = xcd_remap(blockIdx.x, gridDim.x);
        if (g.lower) {
            // bid -> (ti, tj), ti >= tj, row-major enumeration of the lower triangle
            int t = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
            while ((t + 1) * (t + 2) / 2 <= bid) ++t;
            while (t * (t + 1) / 2 > bid) --t;
            ti = t; tj = bid - t * (t + 1) / 2;
        } else {
            int ntn = g.N / BN;
            ti = bid / ntn; tj = bid % ntn;
        }
    }
    const int row0 = ti * BM, col0 = tj * BN;
    const double* Pg = g.P + (int64_t)row0 * g.ldp;
    const double* Qg = g.Q + (int64_t)col0 * g.ldq;

    f64x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

    // All global loads of a stage are issued back to back (no use before store_stage), so they
    // stay in flight under the MFMA block of the previous stage.  The k-scaling is applied when
    // the Q tile is written to LDS.  NT % CH == 0, so a thread's chunk column ch is the same for
    // every i: one w load per thread per stage.
    static_assert(NT % CH == 0, "chunk column must be loop invariant");
    f64x2 pr[PL], qr[QL], wr;
    const int ch0 = tid % CH, r0t = tid / CH;
    constexpr int RSTEP = NT / CH;
    auto load_stage = [&](int k0) {
#pragma unroll
        for (int i = 0; i < PL; ++i)
            pr[i] = *reinterpret_cast<const f64x2*>(Pg + (int64_t)(r0t + i * RSTEP) * g.ldp + k0 + ch0 * 2);
#pragma unroll
        for (int i = 0; i < QL; ++i)
            qr[i] = *reinterpret_cast<const f64x2*>(Qg + (int64_t)(r0t + i * RSTEP) * g.ldq + k0 + ch0 * 2);
        if (SCALE) wr = *reinterpret_cast<const f64x2*>(g.w + k0 + ch0 * 2);
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PL; ++i)
            *reinterpret_cast<f64x2*>(Ps + (buf * BM + r0t + i * RSTEP) * LDT + ch0 * 2) = pr[i];
#pragma unroll
        for (int i = 0; i < QL; ++i) {
            f64x2 v = qr[i];
            if (SCALE) { v.x *= wr.x; v.y *= wr.y; }
            *reinterpret_cast<f64x2*>(Qs + (buf * BN + r0t + i * RSTEP) * LDT + ch0 * 2) = v;
        }
    };

    const int nk = g.K / BK;
    load_stage(0);
    store_stage(0);
    __syncthreads();

    const int fr = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_stage((kt + 1) * BK);        // in flight during the MFMAs below
        const double* pa = Ps + (buf * BM + wm * WTM + fr) * LDT + fk;
        const double* qb = Qs + (buf * BN + wn * WTN + fr) * LDT + fk;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = pa[i * 16 * LDT + kk * 4];
#pragma unroll
            for (int j = 0; j < NI; ++j) b[j] = qb[j * 16 * LDT + kk * 4];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_stage(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[row=(l>>4)+4q][col=l&15].  beta is 0 or 1 on this path; for beta != 0 all
    // C loads are issued before the first use so they overlap (a load-use-store chain per
    // element costs an L2 round trip each: 64 serialized round trips per thread).
    double* cbase = g.C + (int64_t)(row0 + wm * WTM + fk) * g.ldc + col0 + wn * WTN + fr;
    if (g.beta != 0.0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            double cold[NI][4];
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) cold[j][q] = cbase[(int64_t)(i * 16 + 4 * q) * g.ldc + j * 16];
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][j][q] = g.alpha * acc[i][j][q] + g.beta * cold[j][q];
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] *= g.alpha;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int r = row0 + wm * WTM + i * 16 + fk + 4 * q;
                int c = col0 + wn * WTN + j * 16 + fr;
                double v = acc[i][j][q];
                if (g.unit_diag_from >= 0 && r == c && r >= g.unit_diag_from) v = 1.0;
                cbase[(int64_t)(i * 16 + 4 * q) * g.ldc + j * 16] = v;
            }
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
inline hipError_t launch_gemm_nt(const GemmNT& g, hipStream_t stream) {
    int ntm = g.M / BM, ntn = g.N / BN;
    int grid = g.lower ? ntm * (ntm + 1) / 2 : ntm * ntn;
    if (grid <= 0) return hipSuccess;
    if (g.w)
        hipLaunchKernelGGL((gemm_nt_f64_kernel<BM, BN, BK, WAVES_M, WAVES_N, true>), dim3(grid),
                           dim3(64 * WAVES_M * WAVES_N), 0, stream, g);
    else
        hipLaunchKernelGGL((gemm_nt_f64_kernel<BM, BN, BK, WAVES_M, WAVES_N, false>), dim3(grid),
                           dim3(64 * WAVES_M * WAVES_N), 0, stream, g);
    return hipGetLastError();
}

}  // namespace ipm
