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

// Bound of every device-side hand-off spin (polls of ~0.2 us: about 0.7 s).  A wait that runs into it sets the handle's time-out
// word; the host rolls the call back and repeats it on a path that does not poll.  TEST KNOB: the environment variable
// IPM_TEST_SPIN_LIMIT (read at ipm_create) lowers it so that the recovery paths can be driven on purpose (tests only).
__device__ unsigned ipm_spin_limit = 1u << 22;

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
    // split-K of the tail tiles (filled by launch_gemm_nt): logical tiles [0, n_direct) are computed
    // whole by one workgroup each; every tile >= n_direct is cut into split_p K-chunks whose partial
    // tiles go to `slab` and are summed in chunk order by splitk_reduce_kernel (deterministic).
    int n_direct, split_p, chunk_stages;
    double* slab;
    int tile_offset;            // logical tile = tile_offset + index (used to skip the first lower tile)
    const int* tile_order;      // optional: logical tile -> (ti << 16 | tj), a 2-D patch order for L2 reuse
    int64_t sP, sQ, sC;         // batch strides (elements) applied with blockIdx.y; 0 for a single problem
    int batch;                  // gridDim.y (>= 1; no split-K when > 1)
    int64_t sP2, sQ2, sC2;      // second-level batch strides applied with blockIdx.z
    int batch2;                 // gridDim.z (>= 1)
    // Cross-stream hand-off without stream events (an event wait costs the pivot chain ~8 us of command-
    // processor time per Cholesky step): a producer launch on another stream bumps *signal once per
    // workgroup after an agent-scope release; a SMALL consumer launch polls *wait_on >= wait_count from one
    // lane (bounded), then acquires.  Only launches of a few workgroups may wait (no CU starvation).
    unsigned* signal;           // may be null
    const unsigned* wait_on;    // may be null
    unsigned wait_count;
    unsigned* timeout;          // set to 1 when the bounded poll gave up (surfaced as an error by the host)
    unsigned* dbg; unsigned dbg_tag;   // diagnostic (may be null): the first poll of a call that ran into its bound records {1, tag, 6, target, seen}
    long long* trace;                  // diagnostic (may be null): wall_clock64 of workgroup 0 at {start, inputs ready, done}
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

// One workgroup's share of the contraction (tile picked from bx; by, bz = batch indices); lds: 2 (BM + BN)(BK + 2) doubles.
// The body of gemm_nt_f64_kernel, and of the persistent critical-step launch of the fused factorization (ff_crit_kernel).
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool SCALE>
__device__ __forceinline__ void gemm_nt_body(const GemmNT& g, const int bx, const int by, const int bz, double* lds) {
    constexpr bool STORE_EARLY = false;   // measured: writing the next stage before the last k-step is slower (54 vs 60 TFLOP/s)
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int LDT = BK + 2;                 // padded LDS row (doubles)
    constexpr int CH = BK / 2;                  // 16-byte chunks per tile row
    constexpr int PL = (BM * CH) / NT;          // chunks per thread, P tile
    constexpr int QL = (BN * CH) / NT;          // chunks per thread, Q tile
    static_assert((BM * CH) % NT == 0 && (BN * CH) % NT == 0, "tile/threads mismatch");
    static_assert(WTM % 16 == 0 && WTN % 16 == 0 && BK % 4 == 0, "mfma tiling");

    double* Ps = lds;                           // [2][BM][LDT]
    double* Qs = lds + 2 * BM * LDT;            // [2][BN][LDT]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // ---- tile coordinates (+ K range when this workgroup owns a split-K chunk of a tail tile)
    int ti, tj;
    int kbeg = 0, kend = g.K / BK;
    double* slab_out = nullptr;
    {
        int bid;
        if (bx < g.n_direct) {
            bid = xcd_remap(bx, g.n_direct) + g.tile_offset;
        } else {
            int r = bx - g.n_direct;
            bid = g.tile_offset + g.n_direct + r / g.split_p;
            kbeg = (r % g.split_p) * g.chunk_stages;
            kend = min(kend, kbeg + g.chunk_stages);
            slab_out = g.slab + (size_t)r * (BM * BN);
        }
        if (g.tile_order) {
            int packed = g.tile_order[bid];
            ti = packed >> 16; tj = packed & 0xffff;
        } else if (g.lower) {
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
    const double* Pg = g.P + (int64_t)by * g.sP + (int64_t)bz * g.sP2 + (int64_t)row0 * g.ldp;
    const double* Qg = g.Q + (int64_t)by * g.sQ + (int64_t)bz * g.sQ2 + (int64_t)col0 * g.ldq;

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
        if (SCALE) wr = *reinterpret_cast<const f64x2*>(g.w + k0 + ch0 * 2);     // first: oldest in the vmcnt queue
#pragma unroll
        for (int i = 0; i < QL; ++i)
            qr[i] = *reinterpret_cast<const f64x2*>(Qg + (int64_t)(r0t + i * RSTEP) * g.ldq + k0 + ch0 * 2);
#pragma unroll
        for (int i = 0; i < PL; ++i)
            pr[i] = *reinterpret_cast<const f64x2*>(Pg + (int64_t)(r0t + i * RSTEP) * g.ldp + k0 + ch0 * 2);
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < QL; ++i) {
            f64x2 v = qr[i];
            if (SCALE) { v.x *= wr.x; v.y *= wr.y; }
            *reinterpret_cast<f64x2*>(Qs + (buf * BN + r0t + i * RSTEP) * LDT + ch0 * 2) = v;
        }
#pragma unroll
        for (int i = 0; i < PL; ++i)
            *reinterpret_cast<f64x2*>(Ps + (buf * BM + r0t + i * RSTEP) * LDT + ch0 * 2) = pr[i];
    };

    const int nk = kend;
    load_stage(kbeg * BK);
    store_stage(kbeg & 1);
    __syncthreads();

    const int fr = lane & 15, fk = lane >> 4;
    for (int kt = kbeg; kt < nk; ++kt) {
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
            // the other LDS buffer is free for the whole stage (its readers passed the last barrier): write the
            // next stage into it before the last k-step so the stage ends with the barrier only
            if (STORE_EARLY && kk == BK / 4 - 1 && kt + 1 < nk) store_stage(buf ^ 1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (!STORE_EARLY && kt + 1 < nk) store_stage(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: D[row=(l>>4)+4q][col=l&15].  beta is 0 or 1 on this path; for beta != 0 all
    // C loads are issued before the first use so they overlap (a load-use-store chain per
    // element costs an L2 round trip each: 64 serialized round trips per thread).
    if (slab_out) {                                     // split-K partial: raw tile, summed later
        double* sb = slab_out + (wm * WTM + fk) * BN + wn * WTN + fr;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) sb[(i * 16 + 4 * q) * BN + j * 16] = acc[i][j][q];
        return;
    }
    double* cbase = g.C + (int64_t)by * g.sC + (int64_t)bz * g.sC2 + (int64_t)(row0 + wm * WTM + fk) * g.ldc + col0 + wn * WTN + fr;
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
    if (g.signal) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(g.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (g.trace && threadIdx.x == 0 && bx == 0) g.trace[2] = (long long)wall_clock64();
}

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, bool SCALE>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N) / 2 < 2 ? 1 : 2)   // (<= two waves per SIMD: the register budget of the tiles)
void gemm_nt_f64_kernel(GemmNT g) {
    if (g.done && *g.done) {
        // a skipped producer still signals: the stop test may flip `done` while a factorization is in flight (it runs
        // on the residual stream), and a consumer that passed its own check must not spin on a counter nobody bumps
        if (g.signal && threadIdx.x == 0) __hip_atomic_fetch_add(g.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (g.trace && threadIdx.x == 0 && blockIdx.x == 0) g.trace[0] = (long long)wall_clock64();
    if (g.wait_on) {
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(g.wait_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.wait_count) {
                __builtin_amdgcn_s_sleep(4);
                ++spins;
                // give up (no hang; the host rolls the call back and repeats it with stream events): after ~1 s of
                // waiting, or at once when an earlier poll of this call already gave up
                if (spins > ipm_spin_limit || ((spins & 1023u) == 1u && g.timeout &&
                                           __hip_atomic_load(g.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    if (spins > ipm_spin_limit && g.dbg && __hip_atomic_fetch_add(g.dbg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                        g.dbg[1] = g.dbg_tag; g.dbg[2] = 6u; g.dbg[3] = g.wait_count;
                        g.dbg[4] = __hip_atomic_load(g.wait_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (g.timeout) __hip_atomic_store(g.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    if (g.trace && threadIdx.x == 0 && blockIdx.x == 0) g.trace[1] = (long long)wall_clock64();

    __shared__ __attribute__((aligned(16))) double lds[2 * (BM + BN) * (BK + 2)];
    gemm_nt_body<BM, BN, BK, WAVES_M, WAVES_N, SCALE>(g, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, lds);
}

// C tile = beta*C + alpha * (sum of the split_p slabs of that tile, in chunk order); one tail tile per
// blockIdx.y, 1024 elements per workgroup.
template <int BM, int BN>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmNT g) {
    if (g.done && *g.done) return;
    const int tile = g.tile_offset + g.n_direct + blockIdx.y;
    int ti, tj;
    if (g.tile_order) {
        int packed = g.tile_order[tile];
        ti = packed >> 16; tj = packed & 0xffff;
    } else if (g.lower) {
        int t = (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
        while ((t + 1) * (t + 2) / 2 <= tile) ++t;
        while (t * (t + 1) / 2 > tile) --t;
        ti = t; tj = tile - t * (t + 1) / 2;
    } else {
        int ntn = g.N / BN;
        ti = tile / ntn; tj = tile % ntn;
    }
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;          // 4 consecutive elements of one tile row
    const int r = e / BN, c = e % BN;
    const double* sl = g.slab + (size_t)blockIdx.y * g.split_p * (BM * BN) + e;
    f64x2 s0 = (f64x2){0.0, 0.0}, s1 = s0;
    for (int p = 0; p < g.split_p; ++p) {
        f64x2 a = *reinterpret_cast<const f64x2*>(sl + (size_t)p * (BM * BN));
        f64x2 b = *reinterpret_cast<const f64x2*>(sl + (size_t)p * (BM * BN) + 2);
        s0 += a; s1 += b;
    }
    double v[4] = {s0.x, s0.y, s1.x, s1.y};
    const int row = ti * BM + r, col = tj * BN + c;
    double* cp = g.C + (int64_t)row * g.ldc + col;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double o = g.alpha * v[q];
        if (g.beta != 0.0) o += g.beta * cp[q];
        if (g.unit_diag_from >= 0 && row == col + q && row >= g.unit_diag_from) o = 1.0;
        cp[q] = o;
    }
}

constexpr int kSlabTiles = 512;      // capacity of the split-K slab workspace, in BM x BN tiles

// Lockstep batch (lockstep.h): while a handle's launch sequence is being RECORDED, the GEMM launchers hand the finished argument
// struct and grid to this hook instead of launching (thread local: a recording belongs to the calling host thread).
struct GemmRecorder { void (*fn)(void* ctx, int bm, int bn, int bk, int wm, int wn, const GemmNT& g, int grid); void* ctx; };
inline thread_local GemmRecorder* g_gemm_recorder = nullptr;

// slots = workgroups resident at once (2 per CU for the 128x128 tile).  slab may be null (no split).
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N>
inline hipError_t launch_gemm_nt(GemmNT g, hipStream_t stream, double* slab = nullptr, int slots = 512,
                                 int skip_first = 0) {
    int ntm = g.M / BM, ntn = g.N / BN;
    int tiles = (g.lower ? ntm * (ntm + 1) / 2 : ntm * ntn) - skip_first;
    g.tile_offset = skip_first;
    if (tiles <= 0) return hipSuccess;
    if (g.signal || g.wait_on) slab = nullptr;               // hand-off launches are never split
    if (g.batch < 1) { g.batch = 1; g.sP = g.sQ = g.sC = 0; }
    if (g.batch2 < 1) { g.batch2 = 1; g.sP2 = g.sQ2 = g.sC2 = 0; }
    if (g.batch > 1 || g.batch2 > 1) slab = nullptr;
    const int nk = g.K / BK;
    g.n_direct = tiles; g.split_p = 1; g.chunk_stages = nk; g.slab = nullptr;
    int grid = tiles;
    if (slab && slots > 0) {
        int tail = tiles % slots;
        if (tail > 0 && nk >= 16) {
            int p = slots / tail;
            if (p > nk / 8) p = nk / 8;                      // keep >= 8 stages per chunk
            if ((long)tail * p > kSlabTiles) p = kSlabTiles / tail;
            if (p >= 2) {
                int per = (nk + p - 1) / p;
                p = (nk + per - 1) / per;
                g.n_direct = tiles - tail; g.split_p = p; g.chunk_stages = per; g.slab = slab;
                grid = g.n_direct + tail * p;
            }
        }
    }
    if (g_gemm_recorder && !g.w && !g.slab) { g_gemm_recorder->fn(g_gemm_recorder->ctx, BM, BN, BK, WAVES_M, WAVES_N, g, grid); return hipSuccess; }
    if (g.w)
        hipLaunchKernelGGL((gemm_nt_f64_kernel<BM, BN, BK, WAVES_M, WAVES_N, true>), dim3(grid, g.batch, g.batch2),
                           dim3(64 * WAVES_M * WAVES_N), 0, stream, g);
    else
        hipLaunchKernelGGL((gemm_nt_f64_kernel<BM, BN, BK, WAVES_M, WAVES_N, false>), dim3(grid, g.batch, g.batch2),
                           dim3(64 * WAVES_M * WAVES_N), 0, stream, g);
    if (g.slab)
        hipLaunchKernelGGL((splitk_reduce_kernel<BM, BN>), dim3(BM * BN / 1024, tiles - g.n_direct), dim3(256), 0,
                           stream, g);
    return hipGetLastError();
}

}  // namespace ipm
