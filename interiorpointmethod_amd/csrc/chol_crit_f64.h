// chol_crit_f64.h -- the two GEMMs on the CRITICAL PATH of a blocked Cholesky step, as single-stage, register-resident
// kernels (gfx950).  Between potrf_diag(k) and potrf_diag(k+1) the chain needs exactly two 128 x 128 x 128 products:
//
//     L(k+1,k)   = B(k+1,k) inv(L_kk)^T                    (crit_panel_kernel, in place)
//     B(k+1,k+1) -= L(k+1,k) L(k+1,k)^T                    (crit_syrk_kernel, lower 16 x 16 tiles)
//
// The generic gemm_nt_f64_kernel runs them as four K stages of global -> registers -> LDS -> MFMA with a barrier each:
// at this size every stage is one exposed memory latency (16 MFMAs per wave cannot cover it), ~10 us per kernel, 20 us
// of the ~57 us a step of the pivot chain takes.  Here every operand fragment is loaded straight from global memory
// into registers in MFMA layout (lane (r, k) of a 16 x 16 x 4 step reads its own 8 bytes), ALL loads of a wave are
// issued up front -- one memory latency per kernel -- and 32 MFMAs follow with no LDS and no stage barriers.
// The k order of every accumulation is the generic kernel's (k = 0..127 in MFMA steps of 4), so the results are
// bit-identical to it.  Same cross-stream hand-off protocol (wait_on / signal counters) as gemm_nt_f64_kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_nt_f64.h"

namespace ipm {

struct CritStep {
    double* panel; int64_t ld;       // B(k+1,k): rows of the critical block row, K = 128 contiguous columns; in place -> L(k+1,k)
    const double* inv;               // inv(L_kk), 128 x 128 row-major (lower triangular)
    double* C; int64_t ldc;          // B(k+1,k+1) (crit_syrk_kernel)
    const int* done;
    unsigned* signal;                // bumped once per workgroup after its stores (agent-scope release), may be null
    const unsigned* wait_on;         // polled before the first load (bounded), may be null
    unsigned wait_count;
    unsigned* timeout;
};

__device__ __forceinline__ void crit_wait(const CritStep& g) {
    if (g.wait_on) {
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(g.wait_on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.wait_count) {
                __builtin_amdgcn_s_sleep(4);
                ++spins;
                if (spins > (1u << 22) || ((spins & 1023u) == 1u && g.timeout &&
                                           __hip_atomic_load(g.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    if (g.timeout) __hip_atomic_store(g.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
}
__device__ __forceinline__ void crit_signal(const CritStep& g) {
    if (g.signal) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(g.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// L(k+1,k)[r][c] = sum_k B(k+1,k)[r][k] inv[c][k].  grid = rows / 16 workgroups, 512 threads: the workgroup owns 16 whole
// rows (the product is IN PLACE), wave w owns the 16 columns 16 w .. 16 w + 15.  inv is lower triangular: column tile w
// needs k < 16 (w + 1) only, so waves stop their k loop there (chunks of four MFMA steps, wave-uniform).
__global__ __launch_bounds__(512) void crit_panel_kernel(CritStep g) {
    if (g.done && *g.done) {
        if (g.signal && threadIdx.x == 0) __hip_atomic_fetch_add(g.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    crit_wait(g);
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fk = lane >> 4;
    const double* prow = g.panel + (int64_t)(blockIdx.x * 16 + fr) * g.ld + fk;
    const double* qrow = g.inv + (16 * w + fr) * 128 + fk;
    double a[32], b[32];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c <= w) {                                     // k = 16 c .. 16 c + 15 is inside the triangle for this column tile
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[4 * c + q] = prow[(4 * c + q) * 4]; b[4 * c + q] = qrow[(4 * c + q) * 4]; }
        }
    }
    __builtin_amdgcn_sched_barrier(0);                    // every load is in flight before the first MFMA waits
    f64x4 acc = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c <= w) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 * c + q], b[4 * c + q], acc, 0, 0, 0);
        }
    }
    // in place: every wave of the workgroup has its fragments of these 16 rows in registers before any wave stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double* crow = g.panel + (int64_t)(blockIdx.x * 16 + fk) * g.ld + 16 * w + fr;      // D[row = fk + 4 q][col = fr]
#pragma unroll
    for (int q = 0; q < 4; ++q) crow[(int64_t)(4 * q) * g.ld] = acc[q];
    crit_signal(g);
}

// B(k+1,k+1) -= L L^T on the lower 16 x 16 tiles (diagonal tiles whole: potrf_diag reads them symmetric), L = the 128
// rows at `panel`.  grid = 9 workgroups of 4 waves, one tile per wave (36 tiles).
__global__ __launch_bounds__(256) void crit_syrk_kernel(CritStep g) {
    if (g.done && *g.done) return;
    crit_wait(g);
    const int lane = threadIdx.x & 63;
    const int t = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));     // 0..35
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int tj = t - ti * (ti + 1) / 2;
    const int fr = lane & 15, fk = lane >> 4;
    const double* arow = g.panel + (int64_t)(16 * ti + fr) * g.ld + fk;
    const double* brow = g.panel + (int64_t)(16 * tj + fr) * g.ld + fk;
    double* crow = g.C + (int64_t)(16 * ti + fk) * g.ldc + 16 * tj + fr;
    double a[32], b[32], cold[4];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) { a[kk] = arow[kk * 4]; b[kk] = brow[kk * 4]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) cold[q] = crow[(int64_t)(4 * q) * g.ldc];
    __builtin_amdgcn_sched_barrier(0);                    // every load is in flight before the first MFMA waits
    f64x4 acc = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) crow[(int64_t)(4 * q) * g.ldc] = -1.0 * acc[q] + 1.0 * cold[q];
    crit_signal(g);
}

}  // namespace ipm
