// form_factor.h -- FUSED formation + factorization for dense handles (gfx950): B = A diag(d) A^T (main.py:224 of the
// reference) and everything of its blocked Cholesky (the factorization inside main.py:180 / :226) EXCEPT the pivot chain,
// in ONE persistent launch that runs BESIDE the pivot chain instead of before it.
//
// Why.  Formation (2.15 ms at 4096 x 8192) and factorization (2.14 ms) used to be strictly serial, although the
// factorization is a latency chain of 32 x (potrf_diag + two small GEMMs) that keeps a handful of CUs busy.  Two things are
// needed to overlap them: whole CUs that the chain kernels can always get, and a way to run the bulk
// matrix work of the factorization (panel solves, trailing updates) with low latency while the formation saturates the chip.
//   * CUs: this kernel's workgroups take 136 KB of LDS, so exactly one fits a CU, and it is launched with FEWER workgroups
//     than the chip has CUs (248 of 256): the remaining CUs stay empty for potrf_diag (133 KB of LDS) and the two critical
//     GEMMs of every step, which run there at their solo pace (profiles/r02_reserve_probe.log).  No CU mask involved.
//   * Bulk work: the workgroups are WORKERS that draw items from one ordered list (ff_schedule.h): formation chunks
//     (K = n / Q of one 128 x 128 tile, partial sums to a slab) interleaved with update / panel-solve items of the
//     factorization, ordered so that what the chain needs next is always served first.  Formation commutes with the
//     updates (tile = sum of slabs - sum_j L_ij L_cj^T), so the trailing updates do not wait for the formation.
// Hand-offs between workers, and between workers and the chain kernels (other stream), are device counters under the
// agent-scope release / acquire protocol of gemm_nt_f64.h; every spin is bounded (time-out word -> the host rolls the call
// back and repeats it on the serial path).
//
// GEMM core: one 512-thread workgroup per 128 x 128 tile, 8 waves as 2 (M) x 4 (N), each wave 64 x 32 =
// 4 x 2 tiles of v_mfma_f64_16x16x4_f64; BK = 32 stages, operands global -> registers -> LDS (rows padded to 34 doubles:
// conflict-free ds_read_b64 fragment reads), double buffered, one barrier per stage.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ff_schedule.h"
#include "gemm_nt_f64.h"

namespace ipm {

constexpr int FF_THREADS = 512;
constexpr int FF_BK = 32, FF_LDT = FF_BK + 2;
constexpr int FF_OP = 128 * FF_LDT;                    // doubles of one operand tile of one stage
constexpr int FF_LDS_DOUBLES = 4 * FF_OP;              // P, Q x two buffers: 139,264 B -> one workgroup per CU

struct FFArgs {
    const double* A; int64_t lda;      // [mp][lda] row-major, zero padded
    const double* d;                   // scaling, length >= 32 * nstages
    double* B; int64_t ldb;            // [mp][ldb]: tiles, then L in place
    const double* invD;                // [nblk][128*128] inv(L_kk), written by potrf_diag
    double* slab;                      // [ntile][Q][128*128] formation partials
    const FFItem* items; int nitems;
    unsigned* ticket;                  // [1] next item
    unsigned* fcount;                  // [ntile] formation chunks complete
    unsigned* tprog;                   // [ntile] T items complete (sequence number)
    unsigned* lfinal;                  // [nblk] 4 x leading tiles of row r that are final L
    unsigned* dready;                  // [nblk] diagonal tile k ready for potrf_diag (>= 10)
    const unsigned* potrfdone;         // [nblk] diagonal block k factored, inv(L_kk) written (>= 1)
    unsigned* timeout;
    const int* done;
    int nblk, Q, nstages, fstages;     // nstages = K / 32 of the formation, fstages = stages per chunk
    int m;                             // true rows: padding rows get a unit diagonal
};

__device__ __forceinline__ void ff_wait_ge(const unsigned* p, unsigned v, unsigned* timeout) {
    unsigned spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) {
        __builtin_amdgcn_s_sleep(2);
        ++spins;
        if (spins > (1u << 22) || ((spins & 1023u) == 1u && __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

// acc += P[128 x 32 ns] * (Q[128 x 32 ns] . w)^T  (both operands k-contiguous rows).  SRC_REGS: the P operand of stage s
// comes from `src` (a 128 x 128 tile held in accumulator layout by the waves with wn == s) instead of memory.
template <bool SCALE, bool SRC_REGS>
__device__ __forceinline__ void ff_gemm(const double* __restrict__ Pg, int64_t ldp, const double* __restrict__ Qg, int64_t ldq,
                                        const double* __restrict__ w, int ns, double* lds, f64x4 (&acc)[4][2],
                                        const f64x4 (*src)[2]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fk = lane >> 4;
    double* Ps = lds;                  // [2][128][FF_LDT]
    double* Qs = lds + 2 * FF_OP;      // [2][128][FF_LDT]
    // staging: thread -> 16-byte chunk ch of rows r0 + 32 u (u < 4) of both operand tiles
    const int ch = tid & 15, r0 = tid >> 4;
    const double* pP = SRC_REGS ? nullptr : Pg + (int64_t)r0 * ldp + ch * 2;
    const double* pQ = Qg + (int64_t)r0 * ldq + ch * 2;
    f64x2 pr[4], qr[4], wr;
    auto load_stage = [&](int s) {
        if (SCALE) wr = *reinterpret_cast<const f64x2*>(w + (int64_t)s * FF_BK + ch * 2);
#pragma unroll
        for (int u = 0; u < 4; ++u) qr[u] = *reinterpret_cast<const f64x2*>(pQ + (int64_t)u * 32 * ldq + (int64_t)s * FF_BK);
        if (!SRC_REGS) {
#pragma unroll
            for (int u = 0; u < 4; ++u) pr[u] = *reinterpret_cast<const f64x2*>(pP + (int64_t)u * 32 * ldp + (int64_t)s * FF_BK);
        }
    };
    auto store_stage = [&](int s, int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f64x2 v = qr[u];
            if (SCALE) { v.x *= wr.x; v.y *= wr.y; }
            *reinterpret_cast<f64x2*>(Qs + buf * FF_OP + (r0 + 32 * u) * FF_LDT + ch * 2) = v;
        }
        if (!SRC_REGS) {
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<f64x2*>(Ps + buf * FF_OP + (r0 + 32 * u) * FF_LDT + ch * 2) = pr[u];
        } else if (wn == s) {
            // columns 32 s .. 32 s + 31 of the source tile live in the accumulators of the two waves with wn == s:
            // element (row = wm*64 + i*16 + fk + 4q, k = j*16 + fr)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        Ps[buf * FF_OP + (wm * 64 + i * 16 + fk + 4 * q) * FF_LDT + j * 16 + fr] = src[i][j][q];
        }
    };
    load_stage(0);
    store_stage(0, 0);
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
        const int buf = s & 1;
        if (s + 1 < ns) load_stage(s + 1);                 // in flight during the MFMAs below
        const double* pa = Ps + buf * FF_OP + (wm * 64 + fr) * FF_LDT + fk;
        const double* qb = Qs + buf * FF_OP + (wn * 32 + fr) * FF_LDT + fk;
#pragma unroll
        for (int kk = 0; kk < FF_BK / 4; ++kk) {
            double a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = pa[i * 16 * FF_LDT + kk * 4];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = qb[j * 16 * FF_LDT + kk * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < ns) store_stage(s + 1, buf ^ 1);
        __syncthreads();
    }
}

// every storing wave drains, the workgroup meets, ONE lane releases (agent scope); the caller then bumps its counters
__device__ __forceinline__ void ff_publish_begin() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

__global__ __launch_bounds__(FF_THREADS, 2) void form_factor_kernel(FFArgs g) {
    if (g.done && *g.done) return;
    __shared__ __attribute__((aligned(16))) double lds[FF_LDS_DOUBLES];
    __shared__ unsigned ticket_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fk = lane >> 4;
    // this lane's 32 elements of a 128 x 128 tile (accumulator layout): row = er + i*16 + 4q, col = ec + j*16
    const int er = wm * 64 + fk, ec = wn * 32 + fr;

    for (;;) {
        if (tid == 0) ticket_s = __hip_atomic_fetch_add(g.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned n = ticket_s;
        __syncthreads();                                   // ticket_s is rewritten on the next turn
        if (n >= (unsigned)g.nitems) return;
        const FFItem it = g.items[n];
        const int ti = it.i, tc = it.c;
        const int tile = ff_tile(ti, tc);
        f64x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};

        if (it.type == FF_F) {
            // ---- one K-chunk of the formation of tile (ti, tc): raw partial tile -> slab (tile, q)
            const int s0 = it.f.s0, s1 = min(g.nstages, (int)it.f.s1);
            if (s1 > s0)
                ff_gemm<true, false>(g.A + (int64_t)ti * 128 * g.lda + (int64_t)s0 * FF_BK, g.lda,
                                     g.A + (int64_t)tc * 128 * g.lda + (int64_t)s0 * FF_BK, g.lda, g.d + (int64_t)s0 * FF_BK,
                                     s1 - s0, lds, acc, nullptr);
            double* sb = g.slab + ((size_t)tile * g.Q + it.q) * (128 * 128) + er * 128 + ec;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) sb[(i * 16 + 4 * q) * 128 + j * 16] = acc[i][j][q];
            ff_publish_begin();
            if (tid == 0) __hip_atomic_fetch_add(g.fcount + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }

        // ---- T item: wait for what it needs (one lane, bounded), one acquire
        const int j0 = it.t.j0, j1 = it.t.j1;
        const int flags = it.t.flags, seq = it.t.seq;
        if (tid == 0) {
            if (flags & FF_ADD_BASE) ff_wait_ge(g.fcount + tile, (unsigned)g.Q, g.timeout);
            if (!(flags & FF_INIT)) ff_wait_ge(g.tprog + tile, (unsigned)seq - 1u, g.timeout);
            if (j1 > j0) {
                ff_wait_ge(g.lfinal + ti, 4u * (unsigned)j1, g.timeout);
                if (tc != ti) ff_wait_ge(g.lfinal + tc, 4u * (unsigned)j1, g.timeout);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (j1 > j0)
            ff_gemm<false, false>(g.B + (int64_t)ti * 128 * g.ldb + (int64_t)j0 * 128, g.ldb,
                                  g.B + (int64_t)tc * 128 * g.ldb + (int64_t)j0 * 128, g.ldb, nullptr, (j1 - j0) * (128 / FF_BK), lds, acc, nullptr);
        // new tile = [old tile] + [formation slabs, in chunk order] - acc
        double* bt = g.B + ((int64_t)ti * 128 + er) * g.ldb + (int64_t)tc * 128 + ec;
        f64x4 val[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) val[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
        if (!(flags & FF_INIT)) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) val[i][j][q] = bt[(int64_t)(i * 16 + 4 * q) * g.ldb + j * 16];
        }
        if (flags & FF_ADD_BASE) {
            for (int c = 0; c < g.Q; ++c) {
                const double* sb = g.slab + ((size_t)tile * g.Q + c) * (128 * 128) + er * 128 + ec;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) val[i][j][q] += sb[(i * 16 + 4 * q) * 128 + j * 16];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) val[i][j] -= acc[i][j];
        if (ti == tc && (flags & FF_ADD_BASE)) {
            // padding rows of the normal matrix carry a unit diagonal (their rows of A are zero)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = er + i * 16 + 4 * q, c = ec + j * 16;
                        if (r == c && ti * 128 + r >= g.m) val[i][j][q] = 1.0;
                    }
        }
        if (flags & FF_PANEL) {
            // L(ti,tc) = tile inv(L(tc,tc))^T: the tile goes back through LDS stage by stage as the P operand
            if (tid == 0) {
                ff_wait_ge(g.potrfdone + tc, 1u, g.timeout);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
            ff_gemm<false, true>(nullptr, 0, g.invD + (int64_t)tc * 128 * 128, 128, nullptr, 128 / FF_BK, lds, acc, val);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) val[i][j] = acc[i][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) bt[(int64_t)(i * 16 + 4 * q) * g.ldb + j * 16] = val[i][j][q];
        ff_publish_begin();
        if (tid == 0) {
            __hip_atomic_store(g.tprog + tile, (unsigned)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (flags & FF_PANEL) __hip_atomic_fetch_add(g.lfinal + ti, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (flags & FF_SIG_DIAG0) __hip_atomic_fetch_add(g.dready, 10u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// diag(B) of the true rows straight from A and d -> its maximum (the pivot guard's scale, needed BEFORE the first diagonal
// block is factored, i.e. long before B is complete on the fused path).  One wave per row; rows' maxima through LDS; the
// block maxima go to `part`, the last block to arrive (ticket) reduces them in index order (max is order independent anyway).
__global__ __launch_bounds__(256) void ff_maxdiag_kernel(const double* __restrict__ A, int64_t lda, int m, int n, const double* __restrict__ d,
                                                         double* part, unsigned* ticket, double* out, const int* done) {
    if (done && *done) return;
    __shared__ double red[4];
    __shared__ unsigned last_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double mx = -1.7976931348623157e308;
    for (int row = blockIdx.x * 4 + wave; row < m; row += gridDim.x * 4) {
        const double* a = A + (int64_t)row * lda;
        double s = 0.0;
        for (int k = lane * 2; k < n; k += 128) {
            const f64x2 v = *reinterpret_cast<const f64x2*>(a + k);
            const f64x2 w = *reinterpret_cast<const f64x2*>(d + k);
            s = __builtin_fma(v.x * v.x, w.x, s);
            s = __builtin_fma(v.y * v.y, w.y, s);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        mx = (s > mx) ? s : mx;                      // NaN never wins (as in maxdiag_kernel)
    }
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double b = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        part[blockIdx.x] = b;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last_s = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (last_s == gridDim.x - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            double r = -1.7976931348623157e308;
            for (unsigned i = 0; i < gridDim.x; ++i) r = fmax(r, __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            *out = r;
        }
    }
}

}  // namespace ipm
