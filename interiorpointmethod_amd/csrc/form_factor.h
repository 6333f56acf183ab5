// form_factor.h -- FUSED formation + factorization for dense handles (gfx950): B = A diag(d) A^T (main.py:224 of the
// reference) and ALL of its blocked Cholesky (the factorization inside main.py:180 / :226), the pivot chain included, in ONE
// persistent launch of as many workgroups as the device has CUs (form_factor_roles_kernel).
//
// Why.  Formation (2.15 ms at 4096 x 8192) and factorization (2.14 ms) used to be strictly serial, although the
// factorization is a latency chain of 32 x (potrf_diag + two small GEMMs) that keeps a handful of CUs busy.  Two things are
// needed to overlap them: a pivot chain that can always run, and a way to run the bulk matrix work of the
// factorization (panel solves, trailing updates) with low latency while the formation saturates the chip.
//   * Roles: every workgroup takes 139 KB of LDS (exactly one per CU) and first draws a role number from a counter: 0 = the pivot
//     chain (ff_chain_role: potrf_diag_body per diagonal block in LDS), 1 .. 4 = the critical products of every step (ff_crit_role:
//     32-row strips of L(k+1,k) and of the update of tile (k+1,k+1)), everybody else a worker.  Roles are claimed by workgroups
//     that are RUNNING, so the chain is never the workgroup that did not get a CU, and no CU is kept free for anybody.  (Round 3
//     ran the chain as separate launches on a second stream beside 224 workers and kept one CU per shader engine empty for
//     them: workgroups are dealt to XCDs and engines in a fixed rotation whatever is free -- tools/ff_reserve_probe.hip; this
//     structure is still selectable with IPM_FF_CHAIN_MODE=0 and is what form_factor_kernel + the chain launches of enqueue_form_factor do.)
//   * Bulk work: the WORKERS draw items from one ordered list (ff_schedule.h): formation chunks (a K range of a 256 x 128 PAIR
//     of tiles, partial sums to slabs) interleaved with update / panel-solve items of the factorization, in the start order of a
//     bottom-level list scheduling of the item DAG.  Formation commutes with the updates (tile = sum of slabs - sum_j L_ij
//     L_cj^T), so the trailing updates do not wait for the formation.
// Hand-offs between the roles are device counters under the agent-scope release / acquire protocol of gemm_nt_f64.h; every
// spin is bounded (time-out word -> the host rolls the call back and repeats it on the serial path).
//
// GEMM engines (8 waves, v_mfma_f64_16x16x4_f64): ff_gemm_pair for the formation -- 256 x 128 per workgroup, waves 4 (M) x 2 (N),
// BK = 16 stages on the schedule of adat_syrk_kernel (0.87 of the fp64 MFMA peak standalone, tools/ff_gemm_bench.hip);
// ff_gemm_pipe for the updates -- 128 x 128, waves 2 x 4, BK = 32, software pipelined; ff_gemm for the panel product with P
// taken from the accumulators through LDS.  Operands global -> registers -> LDS (rows padded: conflict-free ds_read_b64
// fragment reads), double buffered, one barrier per stage.  DESIGN.md 4-F has the measurements and the rejected variants.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ff_schedule.h"
#include "gemm_nt_f64.h"
#include "potrf_f64.h"

namespace ipm {

constexpr int FF_THREADS = 512;
constexpr int FF_BK = 32, FF_LDT = FF_BK + 2;
constexpr int FF_OP = 128 * FF_LDT;                    // doubles of one operand tile of one stage
constexpr int FF_LDS_DOUBLES = 4 * FF_OP;              // P, Q x two buffers = 139,264 B: exactly one workgroup per CU

struct FFArgs {
    const double* A; int64_t lda;      // [mp][lda] row-major, zero padded
    const double* d;                   // scaling, length >= 32 * nstages
    double* B; int64_t ldb;            // [mp][ldb]: tiles, then L in place
    const double* invD;                // [nblk][128*128] inv(L_kk), written by potrf_diag
    double* slab;                      // [ntile][Q][128*128] formation partials
    const FFItem* items; int nitems;   // the whole list in simulated order: one ticket counter, blocking waits
    unsigned* ticket;                  // [1] next item
    unsigned* fcount;                  // [ntile] formation chunks complete
    unsigned* tprog;                   // [ntile] T items complete (sequence number)
    unsigned* lfinal;                  // [nblk] 4 x leading tiles of row r that are final L
    unsigned* dready;                  // [nblk] diagonal tile k ready for potrf_diag (>= 10)
    const unsigned* potrfdone;         // [nblk] diagonal block k factored, inv(L_kk) written (>= 1)
    unsigned* timeout;
    unsigned* dbg;                     // [8] diagnostic (first wait that gave up), may be null
    unsigned dbg_words;                // hand-off words to snapshot on that occasion (0: none)
    long long* prof;                   // diagnostic (may be null): [workgroups][16] cycles per phase (s_memtime), see FF_PROF
    long long* trace;                  // diagnostic (may be null, IPM_FF_TRACE_ITEMS=1): [nitems][4] wall_clock64 at {drawn, inputs ready, done} + the worker
    const int* done;
    int nblk, Q, nstages;              // Q = slab capacity per tile; nstages = K / 16 of the formation (BK = 16 stages of the pair engine)
    int m;                             // true rows: padding rows get a unit diagonal
    const int* tile_q;                 // [ntile] formation chunks (slabs in use) of the tile, <= Q
    unsigned long long* maxbits;       // FF_D items: max diag(B) over the true rows as the bit pattern of a non-negative double
    unsigned* dcount;                  // FF_D items complete
};

// dbg (optional, 8 words, zeroed per launch): the FIRST wait of the launch that gave up records {1, item, kind, target, seen}
__device__ __forceinline__ void ff_wait_ge(const unsigned* p, unsigned v, unsigned* timeout, unsigned* dbg = nullptr, unsigned item = 0,
                                           unsigned kind = 0, unsigned nw = 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) {
        __builtin_amdgcn_s_sleep(2);
        ++spins;
        if (spins > ipm_spin_limit || ((spins & 1023u) == 1u && __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            if (spins > ipm_spin_limit && dbg && __hip_atomic_fetch_add(dbg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                dbg[1] = item; dbg[2] = kind; dbg[3] = v; dbg[4] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // snapshot of every hand-off word BEFORE the time-out word releases the other waiters (dbg[5] = words, the
                // copy sits right behind the live words)
                const unsigned* live = dbg - 24;
                unsigned* snap = const_cast<unsigned*>(live) + nw;
                for (unsigned w = 0; w < nw; ++w) snap[w] = __hip_atomic_load(live + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
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

// The same product, software pipelined (the schedule of adat_syrk_f64.h carried over to 8 waves and BK = 32): fragment
// reads run one k-step ahead through two register sets; the next stage's operands -- fetched from memory a whole stage
// earlier -- are written to the other LDS buffer between the MFMAs of the second-to-last k-step; the stage barrier follows,
// and the LAST k-step's eight MFMAs (fragments already in registers) issue right behind it while the first fragments of the
// next stage are read.  A wave's MFMA stream therefore continues across the barrier, and no LDS or memory latency sits in
// front of an MFMA.  Same operand tiles, same summation order as ff_gemm: results are bit-identical to it.
template <bool SCALE>
__device__ __forceinline__ void ff_gemm_pipe(const double* __restrict__ Pg, int64_t ldp, const double* __restrict__ Qg, int64_t ldq,
                                             const double* __restrict__ w, int ns, double* lds, f64x4 (&acc)[4][2]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fk = lane >> 4;
    double* Ps = lds;
    double* Qs = lds + 2 * FF_OP;
    const int ch = tid & 15, r0 = tid >> 4;
    const double* pP = Pg + (int64_t)r0 * ldp + ch * 2;
    const double* pQ = Qg + (int64_t)r0 * ldq + ch * 2;
    f64x2 pr[4], qr[4], wr;
    auto issue_loads = [&](int s) {
        if (SCALE) wr = *reinterpret_cast<const f64x2*>(w + (int64_t)s * FF_BK + ch * 2);
#pragma unroll
        for (int u = 0; u < 4; ++u) qr[u] = *reinterpret_cast<const f64x2*>(pQ + (int64_t)u * 32 * ldq + (int64_t)s * FF_BK);
#pragma unroll
        for (int u = 0; u < 4; ++u) pr[u] = *reinterpret_cast<const f64x2*>(pP + (int64_t)u * 32 * ldp + (int64_t)s * FF_BK);
    };
    const int st_off = r0 * FF_LDT + ch * 2;
    auto store_q = [&](int buf, int u) {
        f64x2 v = qr[u];
        if (SCALE) { v.x *= wr.x; v.y *= wr.y; }
        *reinterpret_cast<f64x2*>(Qs + buf * FF_OP + u * 32 * FF_LDT + st_off) = v;
    };
    auto store_p = [&](int buf, int u) { *reinterpret_cast<f64x2*>(Ps + buf * FF_OP + u * 32 * FF_LDT + st_off) = pr[u]; };
    const int fa_off = (wm * 64 + fr) * FF_LDT + fk, fb_off = (wn * 32 + fr) * FF_LDT + fk;
    double fa[2][4], fb[2][2];
    auto read_frags = [&](int set, int buf, int kk) {
        const double* pa = Ps + buf * FF_OP + fa_off + kk * 4;
        const double* qb = Qs + buf * FF_OP + fb_off + kk * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = pa[i * 16 * FF_LDT];
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[set][j] = qb[j * 16 * FF_LDT];
    };
    auto mfma8 = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };
    // prologue: stage 0 in LDS, loads of stage 1 in flight, first fragments in set 0
    issue_loads(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) { store_q(0, u); store_p(0, u); }
    __syncthreads();
    if (1 < ns) issue_loads(1);
    read_frags(0, 0, 0);
    for (int s = 0; s < ns; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < ns;
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {                      // k-steps 0 .. 5: prefetch kk + 1, multiply kk
            read_frags((kk + 1) & 1, buf, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma8(kk & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        read_frags(1, buf, 7);                                // fragments of the last k-step, before the barrier
        __builtin_amdgcn_sched_barrier(0);
        // k-step 6, with the next stage's operands written to the other LDS buffer between its MFMA rows
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
            if (more) { store_q(buf ^ 1, i); store_p(buf ^ 1, i); }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                      // stage s + 1 visible; every read of stage s is issued
        if (s + 2 < ns) issue_loads(s + 2);
        if (more) read_frags(0, buf ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma8(1);                                             // k-step 7
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                          // (the caller may reuse the LDS at once)
}

// ------------------------------------------------------------------------------------------------------------------------
// FORMATION engine: TWO vertically adjacent tiles at once -- a 256 x 128 block of B = A diag(d) A^T -- by the 8 waves as
// 4 (M) x 2 (N), each wave 64 x 64 = 4 x 4 MFMA tiles: per wave exactly the proven schedule of adat_syrk_f64.h (16 MFMAs per
// k-step against 8 fragment reads, BK = 16 stages, fragment reads one k-step ahead, next stage written to LDS under the
// third k-step, barrier, last k-step behind it), and the column panel (Q operand, with the d scaling) is fetched once for
// both tiles.  acc: [4][4] per wave; rows of the block = wm * 64 + ..., i.e. waves 0-3 hold the upper tile, 4-7 the lower.
constexpr int FF_PBK = 16, FF_PLDT = FF_PBK + 2;
constexpr int FF_POP_P = 256 * FF_PLDT, FF_POP_Q = 128 * FF_PLDT;     // doubles of the P / Q tile of one stage
static_assert(2 * (FF_POP_P + FF_POP_Q) <= FF_LDS_DOUBLES, "pair engine LDS");

// P0 / P1: first row of the upper / lower tile's 128-row panel of A (an invalid half is given any valid panel).
__device__ __forceinline__ void ff_gemm_pair(const double* __restrict__ P0, const double* __restrict__ P1, const double* __restrict__ Qg,
                                             int64_t ld, const double* __restrict__ w, int ns, double* lds, f64x4 (&acc)[4][4]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fk = lane >> 4;
    double* Ps = lds;                            // [2][256][18]
    double* Qs = lds + 2 * FF_POP_P;             // [2][128][18]
    // staging: thread -> 16-byte chunk ch (of 8) of rows r0 + 64 u: P u < 4 (256 rows), Q u < 2 (128 rows)
    const int ch = tid & 7, r0 = tid >> 3;
    const double* pP0 = P0 + (int64_t)r0 * ld + ch * 2;
    const double* pP1 = P1 + (int64_t)r0 * ld + ch * 2;
    const double* pQ = Qg + (int64_t)r0 * ld + ch * 2;
    f64x2 pr[4], qr[2], wr;
    auto issue_loads = [&](int s) {
        wr = *reinterpret_cast<const f64x2*>(w + (int64_t)s * FF_PBK + ch * 2);
#pragma unroll
        for (int u = 0; u < 2; ++u) qr[u] = *reinterpret_cast<const f64x2*>(pQ + (int64_t)u * 64 * ld + (int64_t)s * FF_PBK);
#pragma unroll
        for (int u = 0; u < 2; ++u) pr[u] = *reinterpret_cast<const f64x2*>(pP0 + (int64_t)u * 64 * ld + (int64_t)s * FF_PBK);
#pragma unroll
        for (int u = 0; u < 2; ++u) pr[2 + u] = *reinterpret_cast<const f64x2*>(pP1 + (int64_t)u * 64 * ld + (int64_t)s * FF_PBK);
    };
    const int st_off = r0 * FF_PLDT + ch * 2;
    auto store_q = [&](int buf, int u) {
        f64x2 v = qr[u];
        v.x *= wr.x; v.y *= wr.y;
        *reinterpret_cast<f64x2*>(Qs + buf * FF_POP_Q + u * 64 * FF_PLDT + st_off) = v;
    };
    auto store_p = [&](int buf, int u) { *reinterpret_cast<f64x2*>(Ps + buf * FF_POP_P + u * 64 * FF_PLDT + st_off) = pr[u]; };
    const int fa_off = (wm * 64 + fr) * FF_PLDT + fk, fb_off = (wn * 64 + fr) * FF_PLDT + fk;
    double fa[2][4], fb[2][4];
    auto read_frags = [&](int set, int buf, int kk) {
        const double* pa = Ps + buf * FF_POP_P + fa_off + kk * 4;
        const double* qb = Qs + buf * FF_POP_Q + fb_off + kk * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = pa[i * 16 * FF_PLDT];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[set][j] = qb[j * 16 * FF_PLDT];
    };
    auto mfma16 = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };
    issue_loads(0);
#pragma unroll
    for (int u = 0; u < 2; ++u) store_q(0, u);
#pragma unroll
    for (int u = 0; u < 4; ++u) store_p(0, u);
    __syncthreads();
    if (1 < ns) issue_loads(1);
    read_frags(0, 0, 0);
    for (int s = 0; s < ns; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < ns;
        read_frags(1, buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(0);                                              // k-step 0
        __builtin_amdgcn_sched_barrier(0);
        read_frags(0, buf, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(1);                                              // k-step 1
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1, buf, 3);
        __builtin_amdgcn_sched_barrier(0);
        // k-step 2, with the next stage's operands written to the other LDS buffer between its MFMA rows
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
            if (more) { if (i < 2) store_q(buf ^ 1, i); store_p(buf ^ 1, i); }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (s + 2 < ns) issue_loads(s + 2);
        if (more) read_frags(0, buf ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(1);                                              // k-step 3
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
}

// Hand-off of a tile / slab to workgroups on other CUs and XCDs: plain stores, every storing wave drains its stores, the
// workgroup meets, ONE lane releases at agent scope (L2 write-back) and the caller's lane 0 then bumps the counters;
// consumers poll, take one agent-scope acquire and read with plain loads.  (Measured and dropped in round 3: write-through
// sc1 stores without the release -- 8-byte write-through stores from the MFMA layout cost more than the one fence.)
__device__ __forceinline__ void ff_publish_begin() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The PIVOT CHAIN as a ROLE of the persistent launch (FFModel::chain_mode 1): the first workgroup to arrive factors the diagonal
// blocks in order, each as soon as its tile has been handed over (dready[k]), and publishes L_kk / inv(L_kk) (potrfdone[k]).  No
// launch boundaries on the chain and no CUs kept free for chain launches: ONE launch of as many workgroups as the chip has CUs,
// dealt evenly by the dispatcher whatever its rotation (separate persistent launches are not: a workgroup whose turn falls on a
// shader engine that persistent workgroups fill never starts -- measured: intermittent spin-bound time-outs), and the roles are
// claimed by workgroups that are running.
struct FFChain {
    double* B; int64_t ldb; double* invD;
    const unsigned long long* maxbits; const unsigned* dcount; double* maxdiag_out;
    const unsigned* dready; unsigned* potrfdone;
    unsigned* timeout; unsigned* dbg; long long* trace;      // trace: [nblk][12] (slots 0..2 of each block used)
    double eps, big, shift_rel;
    int* fixed; const int* done;
    int nblk, m;
};

__device__ __forceinline__ void ff_chain_role(const FFChain& c, double* lds) {
    static_assert(NB * WLD + NB + 2 <= FF_LDS_DOUBLES, "chain role LDS");
    double* W = lds;                                         // [NB][WLD]
    double* dinv_s = lds + NB * WLD;                         // [NB]
    double* maxdiag_p = dinv_s + NB;
    const int tid = threadIdx.x;
    if (tid == 0) {
        ff_wait_ge(c.dcount, (unsigned)c.nblk, c.timeout, c.dbg, 9000u, 8, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const double mx = __longlong_as_double((long long)__hip_atomic_load(c.maxbits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        *maxdiag_p = mx;
        *c.maxdiag_out = mx;
    }
    __syncthreads();
    for (int k = 0; k < c.nblk; ++k) {
        PotrfDiag a;
        a.Bkk = c.B + (int64_t)k * NB * (c.ldb + 1); a.ld = c.ldb;
        a.inv = c.invD + (int64_t)k * NB * NB;
        a.maxdiag = maxdiag_p; a.eps = c.eps; a.big = c.big; a.shift_rel = c.shift_rel;
        a.fixed = c.fixed; a.done = nullptr; a.stamps = nullptr;
        a.wait_on = c.dready + k; a.wait_count = 4u; a.signal = c.potrfdone + k; a.timeout = c.timeout; a.dbg = c.dbg; a.dbg_tag = (unsigned)k;
        a.trace = c.trace ? c.trace + 12 * (size_t)k : nullptr;
        const int real = c.m - k * NB;                       // 16-wide panels that hold rows of the LP (potrf_panels of the host)
        a.nt = (c.shift_rel != 0.0 || real >= NB) ? NB / 16 : (real + 15) / 16 < 1 ? 1 : (real + 15) / 16;
        potrf_diag_body<false>(a, W, dinv_s);
        __syncthreads();
    }
}

// The two small products of every chain step as the role of the next FOUR workgroups to arrive (chain_mode 1), each owning a 32-row
// strip: L(k+1,k) = tile inv(L_kk)^T in place (a workgroup owns whole rows: BN = N = 128), then tile (k+1,k+1) -= L(k+1,k)
// L(k+1,k)^T (strip x all rows of L(k+1,k); the part above the diagonal is computed too and never read).  The products are
// gemm_nt_body<32,128,32,1,8> -- the kernels the launch-per-step chain uses -- with the hand-offs of the fused launch around
// them: lfinal[k+1] += 1 per strip (4 = the tile is final L), dready[k+1] += 1 per strip (potrf waits for 4).
struct FFCrit {
    double* B; int64_t ldb; const double* invD;
    const unsigned* tprog; const int* tile_items;           // [ntile] T items per tile: the workers' part of a tile is complete
    const unsigned* potrfdone; unsigned* lfinal; unsigned* dready;
    unsigned* timeout; unsigned* dbg; long long* trace;
    const int* done;
    int nblk;
};
constexpr int FF_CRIT_WGS = 4;

__device__ __forceinline__ void ff_crit_role(const FFCrit& c, const int strip, double* lds) {
    static_assert(2 * (32 + 128) * (32 + 2) <= FF_LDS_DOUBLES, "critical role LDS");
    const int tid = threadIdx.x;
    GemmNT g;
    g.w = nullptr; g.M = 128; g.N = 128; g.K = 128; g.lower = 0; g.unit_diag_from = -1; g.done = nullptr;
    g.n_direct = FF_CRIT_WGS; g.split_p = 1; g.chunk_stages = 128 / 32; g.slab = nullptr; g.tile_offset = 0; g.tile_order = nullptr;
    g.sP = g.sQ = g.sC = 0; g.batch = 1; g.sP2 = g.sQ2 = g.sC2 = 0; g.batch2 = 1;
    g.wait_on = nullptr; g.wait_count = 0; g.timeout = nullptr; g.dbg = nullptr; g.dbg_tag = 0; g.trace = nullptr;
    for (int k = 0; k + 1 < c.nblk; ++k) {
        double* panel = c.B + (int64_t)(k + 1) * 128 * c.ldb + (int64_t)k * 128;
        long long* tr = (c.trace && strip == 0 && tid == 0) ? c.trace + 12 * (size_t)k : nullptr;
        if (tr) tr[4] = (long long)wall_clock64();
        if (tid == 0) {
            const int t = ff_tile(k + 1, k);
            ff_wait_ge(c.tprog + t, (unsigned)c.tile_items[t], c.timeout, c.dbg, 1000u + (unsigned)k, 6, 0);
            ff_wait_ge(c.potrfdone + k, 1u, c.timeout, c.dbg, 1000u + (unsigned)k, 6, 0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (tr) tr[5] = (long long)wall_clock64();
        g.P = panel; g.ldp = c.ldb; g.Q = c.invD + (int64_t)k * 128 * 128; g.ldq = 128;
        g.C = panel; g.ldc = c.ldb; g.alpha = 1.0; g.beta = 0.0; g.signal = c.lfinal + (k + 1);
        gemm_nt_body<32, 128, 32, 1, 8, false>(g, strip, 0, 0, lds);
        if (tr) { tr[6] = (long long)wall_clock64(); tr[8] = tr[6]; }
        if (tid == 0) {
            const int t = ff_tile(k + 1, k + 1);
            ff_wait_ge(c.lfinal + (k + 1), 4u * (unsigned)(k + 1), c.timeout, c.dbg, 2000u + (unsigned)k, 6, 0);
            ff_wait_ge(c.tprog + t, (unsigned)c.tile_items[t], c.timeout, c.dbg, 2000u + (unsigned)k, 6, 0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (tr) tr[9] = (long long)wall_clock64();
        g.P = panel; g.ldp = c.ldb; g.Q = panel; g.ldq = c.ldb;
        g.C = c.B + (int64_t)(k + 1) * 128 * (c.ldb + 1); g.ldc = c.ldb; g.alpha = -1.0; g.beta = 1.0; g.signal = c.dready + (k + 1);
        gemm_nt_body<32, 128, 32, 1, 8, false>(g, strip, 0, 0, lds);
        if (tr) tr[10] = (long long)wall_clock64();
        __syncthreads();
    }
}

// phases of the diagnostic cycle profile (IPM_FF_PROF=1; tools/ff_debug.py): wave 0 stamps s_memtime at phase boundaries
enum { FFP_TICKET = 0, FFP_FGEMM, FFP_FSTORE, FFP_TWAIT, FFP_TGEMM, FFP_TBASE, FFP_PWAIT, FFP_PGEMM, FFP_TSTORE, FFP_NF, FFP_NT, FFP_TOTAL };
#define FF_PROF(slot) do { if (TRACE && g.prof && tid == 0) { const long long t_ = __builtin_amdgcn_s_memtime(); g.prof[(size_t)blockIdx.x * 16 + (slot)] += t_ - tprev; tprev = t_; } } while (0)
// per-item time line on the device-wide 100 MHz clock (tools/ff_trace.py)
#define FF_TRACE(slot) do { if (TRACE && g.trace && tid == 0) g.trace[(size_t)n * 4 + (slot)] = (long long)wall_clock64(); } while (0)

// TRACE: the diagnostic instantiation (IPM_FF_PROF / IPM_FF_TRACE_ITEMS) carries the stamps; the shipped one none of their
// code -- the worker loop sits at the register limit and every extra path costs spills.
// Everything the persistent launch of chain_mode 1 needs: the workers' arguments and those of the two chain roles.
struct FFRoles { FFChain chain; FFCrit crit; unsigned* role; };

template <bool TRACE, bool ROLES>
__device__ __forceinline__ void form_factor_body(const FFArgs& g, const FFRoles* r) {
    __shared__ __attribute__((aligned(16))) double lds[FF_LDS_DOUBLES];
    __shared__ unsigned ticket_s;
    if (ROLES) {
        // roles by ARRIVAL: 0 = the pivot chain, 1 .. 4 = the strips of the critical products, everybody else works
        if (threadIdx.x == 0) ticket_s = __hip_atomic_fetch_add(r->role, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned role = ticket_s;
        __syncthreads();
        if (role == 0u) { ff_chain_role(r->chain, lds); return; }
        if (role <= (unsigned)FF_CRIT_WGS) { ff_crit_role(r->crit, (int)role - 1, lds); return; }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fk = lane >> 4;
    // this lane's 32 elements of a 128 x 128 tile (accumulator layout): row = er + i*16 + 4q, col = ec + j*16
    const int er = wm * 64 + fk, ec = wn * 32 + fr;

    long long tprev = (TRACE && g.prof) ? __builtin_amdgcn_s_memtime() : 0;
    const long long tstart = tprev;
    for (;;) {
        constexpr unsigned SEL_EXIT = 0xffffffffu;
        if (tid == 0) {
            const unsigned n0 = __hip_atomic_fetch_add(g.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket_s = n0 < (unsigned)g.nitems ? n0 : SEL_EXIT;
        }
        __syncthreads();
        const unsigned n = ticket_s;
        __syncthreads();                                   // ticket_s is rewritten on the next turn
        if (n == SEL_EXIT) { if (TRACE && g.prof && tid == 0) g.prof[(size_t)blockIdx.x * 16 + FFP_TOTAL] = __builtin_amdgcn_s_memtime() - tstart; return; }
        const FFItem it = g.items[n];
        FF_PROF(FFP_TICKET);
        FF_TRACE(0);
        if (TRACE && g.trace && tid == 0) g.trace[(size_t)n * 4 + 3] = (long long)blockIdx.x;
        const int ti = it.i, tc = it.c;
        const int tile = ff_tile(ti, tc);
        if (it.type == FF_D) {
            // ---- diag(B) of the true rows of block ti straight from A and d -> running maximum (the pivot guard's scale): one
            //      wave per row, 16 rows per wave; max of non-negative doubles through their bit patterns (order independent)
            double mx = 0.0;
            const int ncols = g.nstages * FF_PBK;
            for (int rr = wave; rr < 128; rr += 8) {
                const int row = ti * 128 + rr;
                if (row >= g.m) break;
                const double* a = g.A + (int64_t)row * g.lda;
                // four independent partial sums, eight 16-byte loads in flight per lane (one dependent FMA chain per row made an
                // item 400 us: 8 MB at 20 GB/s)
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                int kq = lane * 2;
                for (; kq + 384 < ncols; kq += 512) {
                    const f64x2 a0 = *reinterpret_cast<const f64x2*>(a + kq), a1 = *reinterpret_cast<const f64x2*>(a + kq + 128);
                    const f64x2 a2 = *reinterpret_cast<const f64x2*>(a + kq + 256), a3 = *reinterpret_cast<const f64x2*>(a + kq + 384);
                    const f64x2 d0 = *reinterpret_cast<const f64x2*>(g.d + kq), d1 = *reinterpret_cast<const f64x2*>(g.d + kq + 128);
                    const f64x2 d2 = *reinterpret_cast<const f64x2*>(g.d + kq + 256), d3 = *reinterpret_cast<const f64x2*>(g.d + kq + 384);
                    s0 = __builtin_fma(a0.x * a0.x, d0.x, s0); s0 = __builtin_fma(a0.y * a0.y, d0.y, s0);
                    s1 = __builtin_fma(a1.x * a1.x, d1.x, s1); s1 = __builtin_fma(a1.y * a1.y, d1.y, s1);
                    s2 = __builtin_fma(a2.x * a2.x, d2.x, s2); s2 = __builtin_fma(a2.y * a2.y, d2.y, s2);
                    s3 = __builtin_fma(a3.x * a3.x, d3.x, s3); s3 = __builtin_fma(a3.y * a3.y, d3.y, s3);
                }
                for (; kq < ncols; kq += 128) {
                    const f64x2 va = *reinterpret_cast<const f64x2*>(a + kq);
                    const f64x2 vd = *reinterpret_cast<const f64x2*>(g.d + kq);
                    s0 = __builtin_fma(va.x * va.x, vd.x, s0);
                    s0 = __builtin_fma(va.y * va.y, vd.y, s0);
                }
                double sacc = (s0 + s1) + (s2 + s3);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
                mx = (sacc > mx) ? sacc : mx;            // NaN never wins
            }
            if (lane == 0) atomicMax(g.maxbits, (unsigned long long)__double_as_longlong(mx));
            ff_publish_begin();
            if (tid == 0) __hip_atomic_fetch_add(g.dcount, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            FF_TRACE(2);
            continue;
        }
        if (it.type == FF_F) {
            // ---- one K-chunk of the formation of the tile PAIR (ti, tc), (ti + 1, tc): raw partial tiles -> slabs (tile, q).
            //      A half above the diagonal (ti < tc) or below the matrix (ti + 1 == nblk) is computed on a stand-in panel and
            //      dropped.
            const bool up = ti >= tc, lo = ti + 1 < g.nblk;
            const int s0 = it.f.s0, s1 = min(g.nstages, (int)it.f.s1);
            f64x4 pacc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) pacc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
            const int r_up = up ? ti : ti + 1, r_lo = lo ? ti + 1 : ti;
            if (s1 > s0)
                ff_gemm_pair(g.A + (int64_t)r_up * 128 * g.lda + (int64_t)s0 * FF_PBK, g.A + (int64_t)r_lo * 128 * g.lda + (int64_t)s0 * FF_PBK,
                             g.A + (int64_t)tc * 128 * g.lda + (int64_t)s0 * FF_PBK, g.lda, g.d + (int64_t)s0 * FF_PBK, s1 - s0, lds, pacc);
            FF_PROF(FFP_FGEMM);
            {
                const int pm = wave >> 1, pn = wave & 1;              // the pair engine's wave grid: 4 (M) x 2 (N)
                const int half = pm >> 1;                             // 0: upper tile, 1: lower tile
                if (half == 0 ? up : lo) {
                    // slab layout = REGISTER-MAJOR: [wave row 2][wave column pair 2][i 4][j 4][q pair 2][lane 64][2] -- every store (and
                    // every load of the update item that adds the slabs) is one contiguous 1 KB per wave instruction.  The update
                    // engine's wave (wm, wn) holds columns wn * 32 + j * 16: the same lanes as this engine's wave (wm, wn >> 1), j' =
                    // (wn & 1) * 2 + j
                    const int tl = ff_tile(ti + half, tc);
                    double* sb = g.slab + ((size_t)tl * g.Q + it.q) * (128 * 128) + (size_t)((pm & 1) * 2 + pn) * 4096 + lane * 2;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int hq = 0; hq < 2; ++hq)
                                *reinterpret_cast<f64x2*>(sb + ((i * 4 + j) * 2 + hq) * 128) = (f64x2){pacc[i][j][2 * hq], pacc[i][j][2 * hq + 1]};
                }
            }
            ff_publish_begin();
            if (tid == 0) {
                if (up) __hip_atomic_fetch_add(g.fcount + ff_tile(ti, tc), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lo) __hip_atomic_fetch_add(g.fcount + ff_tile(ti + 1, tc), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (TRACE && g.prof) g.prof[(size_t)blockIdx.x * 16 + FFP_NF] += 1;
            }
            FF_PROF(FFP_FSTORE);
            FF_TRACE(2);
            continue;
        }

        // ---- T item: wait for what it needs (one lane, bounded), one acquire
        f64x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
        const int j0 = it.t.j0, j1 = it.t.j1;
        const int flags = it.t.flags, seq = it.t.seq;
        if (tid == 0) {
            if (flags & FF_ADD_BASE) ff_wait_ge(g.fcount + tile, (unsigned)g.tile_q[tile], g.timeout, g.dbg, n, 1, g.dbg_words);
            if (!(flags & FF_INIT)) ff_wait_ge(g.tprog + tile, (unsigned)seq - 1u, g.timeout, g.dbg, n, 2, g.dbg_words);
            if (j1 > j0) {
                ff_wait_ge(g.lfinal + ti, 4u * (unsigned)j1, g.timeout, g.dbg, n, 3, g.dbg_words);
                if (tc != ti) ff_wait_ge(g.lfinal + tc, 4u * (unsigned)j1, g.timeout, g.dbg, n, 4, g.dbg_words);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        FF_PROF(FFP_TWAIT);
        FF_TRACE(1);
        if (j1 > j0)
            ff_gemm_pipe<false>(g.B + (int64_t)ti * 128 * g.ldb + (int64_t)j0 * 128, g.ldb,
                                g.B + (int64_t)tc * 128 * g.ldb + (int64_t)j0 * 128, g.ldb, nullptr, (j1 - j0) * (128 / FF_BK), lds, acc);
        FF_PROF(FFP_TGEMM);
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
            const int qn = g.tile_q[tile];
            // (a second register set that keeps slab c + 1 in flight while slab c is added was tried: the kernel then spills 483
            //  VGPRs -- the worker loop sits at the register limit)
            for (int c = 0; c < qn; ++c) {
                const double* sb = g.slab + ((size_t)tile * g.Q + c) * (128 * 128) + (size_t)(wm * 2 + (wn >> 1)) * 4096 + lane * 2;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int hq = 0; hq < 2; ++hq) {
                            const f64x2 v = *reinterpret_cast<const f64x2*>(sb + ((i * 4 + (wn & 1) * 2 + j) * 2 + hq) * 128);
                            val[i][j][2 * hq] += v.x; val[i][j][2 * hq + 1] += v.y;
                        }
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
        FF_PROF(FFP_TBASE);
        if (flags & FF_PANEL) {
            // L(ti,tc) = tile inv(L(tc,tc))^T: the tile goes back through LDS stage by stage as the P operand
            if (tid == 0) {
                ff_wait_ge(g.potrfdone + tc, 1u, g.timeout, g.dbg, n, 5, g.dbg_words);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            FF_PROF(FFP_PWAIT);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
            ff_gemm<false, true>(nullptr, 0, g.invD + (int64_t)tc * 128 * 128, 128, nullptr, 128 / FF_BK, lds, acc, val);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) val[i][j] = acc[i][j];
            FF_PROF(FFP_PGEMM);
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
            if (flags & FF_SIG_DIAG) __hip_atomic_fetch_add(g.dready + ti, 10u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (TRACE && g.prof) g.prof[(size_t)blockIdx.x * 16 + FFP_NT] += 1;
        }
        FF_PROF(FFP_TSTORE);
        FF_TRACE(2);
    }
}

template <bool TRACE>
__global__ __launch_bounds__(FF_THREADS, 2) void form_factor_kernel(FFArgs g) {
    if (g.done && *g.done) return;
    form_factor_body<TRACE, false>(g, nullptr);
}
// chain_mode 1: one launch of as many workgroups as the device has CUs; the chain and the critical products are roles
template <bool TRACE>
__global__ __launch_bounds__(FF_THREADS, 2) void form_factor_roles_kernel(FFArgs g, FFRoles r) {
    if (g.done && *g.done) return;
    form_factor_body<TRACE, true>(g, &r);
}

// One wave that waits (bounded) until *flag >= value: the gate in front of the kernels of ANOTHER stream that may only run once
// the persistent launch has reached a step (chain_mode 1: stream events cannot mark a point inside a launch).  It becomes
// resident when a CU has room -- at the latest when the first workers leave -- and holds 64 threads while it waits.
__global__ __launch_bounds__(64) void ff_gate_kernel(const unsigned* flag, unsigned value, unsigned* timeout, const int* done) {
    if (done && *done) return;
    if (threadIdx.x == 0) {
        ff_wait_ge(flag, value, timeout);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
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
