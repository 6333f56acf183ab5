// chol_update_f64.h -- the bulk trailing update of the blocked Cholesky, C = beta C + alpha P Q^T on 128 x 128 tiles
// (B_ij -= L_ik L_jk^T with K = 128, or K = 128 gs for the deferred update of a group of the two-level schedule), fp64 MFMA
// for gfx950.  Part of the factorization behind main.py:180 / :226 of the reference (scipy's spsolve there).
//
// Same tiling, tile enumeration, summation order and epilogue arithmetic as gemm_nt_f64_kernel<128,128,16,2,2,false>
// (results are bit-identical to it, tests/test_gpu_kernels.py); what differs is the stage schedule, which is the one of
// adat_syrk_kernel (adat_syrk_f64.h): fragment reads one k-step ahead through two register sets, the stage barrier
// before the LAST k-step with the next stage written to the other LDS buffer between the MFMA rows of k-step 2, operands
// fetched with buffer loads (one VGPR offset per thread).  The generic kernel needs 256 VGPRs + 17 spilled for this
// shape; this one keeps its K loop free of scratch (two 8-byte spills in the epilogue).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "adat_syrk_f64.h"
#include "gemm_nt_f64.h"

namespace ipm {

__device__ __forceinline__ void chol_update_kernel_body(GemmNT g, const unsigned bx_, const unsigned gx_) {
    constexpr int BM = 128, BK = 16, LDT = BK + 2, RSTEP = 32;
    if (g.done && *g.done) {
        if (g.signal && threadIdx.x == 0) __hip_atomic_fetch_add(g.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __shared__ __attribute__((aligned(16))) double lds[2 * (BM + BM) * LDT];
    double* Ps = lds;                           // [2][128][LDT]
    double* Qs = lds + 2 * BM * LDT;            // [2][128][LDT]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int ti, tj;
    {
        const int bid = xcd_remap(bx_, gx_) + g.tile_offset;
        if (g.tile_order) {
            const int packed = g.tile_order[bid];
            ti = packed >> 16; tj = packed & 0xffff;
        } else if (g.lower) {
            int t = (int)((sqrtf(8.0f * (float)bid + 1.0f) - 1.0f) * 0.5f);
            while ((t + 1) * (t + 2) / 2 <= bid) ++t;
            while (t * (t + 1) / 2 > bid) --t;
            ti = t; tj = bid - t * (t + 1) / 2;
        } else {
            const int ntn = g.N / BM;
            ti = bid / ntn; tj = bid % ntn;
        }
    }
    const int row0 = ti * BM, col0 = tj * BM;
    const int nk = g.K / BK;
    const int ldp = (int)g.ldp, ldq = (int)g.ldq;

    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)(g.P + (int64_t)row0 * g.ldp), 0, (int)((unsigned)BM * (unsigned)ldp * 8u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rQ = __builtin_amdgcn_make_buffer_rsrc((void*)(g.Q + (int64_t)col0 * g.ldq), 0, (int)((unsigned)BM * (unsigned)ldq * 8u), 0x00020000);

    const int ch0 = tid & 7, r0t = tid >> 3;
    const int voffP = (r0t * ldp + ch0 * 2) * 8, voffQ = (r0t * ldq + ch0 * 2) * 8;
    const int rstepP = RSTEP * ldp * 8, rstepQ = RSTEP * ldq * 8;
    f64x2 pr[4], qr[4];
    auto issue_loads = [&](int kt) {
        const int kb = kt * BK * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) qr[i] = buf_load_f64x2(rQ, voffQ, kb + i * rstepQ);
#pragma unroll
        for (int i = 0; i < 4; ++i) pr[i] = buf_load_f64x2(rP, voffP, kb + i * rstepP);
    };
    const int st_off = r0t * LDT + ch0 * 2;
    auto store_q = [&](int buf, int i) { *reinterpret_cast<f64x2*>(Qs + buf * BM * LDT + i * RSTEP * LDT + st_off) = qr[i]; };
    auto store_p = [&](int buf, int i) { *reinterpret_cast<f64x2*>(Ps + buf * BM * LDT + i * RSTEP * LDT + st_off) = pr[i]; };

    const int fr = lane & 15, fk = lane >> 4;
    const int fa_off = (wm * 64 + fr) * LDT + fk, fb_off = (wn * 64 + fr) * LDT + fk;
    double fa[2][4], fb[2][4];
    auto read_frags = [&](int set, int buf, int kk) {
        const double* pa = Ps + buf * BM * LDT + fa_off + kk * 4;
        const double* qb = Qs + buf * BM * LDT + fb_off + kk * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = pa[i * 16 * LDT];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[set][j] = qb[j * 16 * LDT];
    };
    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
    auto mfma16 = [&](int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };

    issue_loads(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_q(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_p(0, i);
    __syncthreads();
    if (1 < nk) issue_loads(1);
    read_frags(0, 0, 0);

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < nk;
        read_frags(1, buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(0, buf, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1, buf, 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
            if (more) { store_q(buf ^ 1, i); store_p(buf ^ 1, i); }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (kt + 2 < nk) issue_loads(kt + 2);
        if (more) read_frags(0, buf ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(1);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue (the arithmetic of the generic kernel): D[row=(l>>4)+4q][col=l&15]
    // C through a buffer resource as well: one VGPR offset per thread, the 64 row / column offsets of a thread are scalars
    const int ldc = (int)g.ldc;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)(g.C + (int64_t)row0 * g.ldc + col0), 0, (int)((unsigned)BM * (unsigned)ldc * 8u), 0x00020000);
    const int voffC = ((wm * 64 + fk) * ldc + wn * 64 + fr) * 8;
    typedef int i32x2_t __attribute__((ext_vector_type(2)));
    if (g.beta != 0.0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double cold[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    cold[j][q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rC, voffC, ((i * 16 + 4 * q) * ldc + j * 16) * 8, 0));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][j][q] = g.alpha * acc[i][j][q] + g.beta * cold[j][q];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] *= g.alpha;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2_t, (double)acc[i][j][q]), rC, voffC, ((i * 16 + 4 * q) * ldc + j * 16) * 8, 0);
    if (g.signal) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(g.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__global__ __launch_bounds__(256, 2) void chol_update_kernel(GemmNT g) { chol_update_kernel_body(g, blockIdx.x, gridDim.x); }

// Drop-in for launch_gemm_nt<128,128,16,2,2>(g, stream, nullptr, 512, skip_first) on the shapes the trailing update uses:
// no scaling, no split-K, no batch, no wait, no unit diagonal; M, N multiples of 128, K of 16.  Anything else -> the
// generic kernel.
// (A 2-D patch order of the tiles, as the formation uses, was measured and changes nothing here: 30.51 vs 30.40 ms per
// factorization at 16384 x 32768, 6.817 vs 6.804 at 8192 x 16384 -- the row-major enumeration stays.)
inline hipError_t launch_chol_update(GemmNT g, hipStream_t stream, int skip_first = 0) {
    const bool plain = !g.w && !g.wait_on && g.unit_diag_from < 0 && g.batch <= 1 && g.batch2 <= 1 && g.M % 128 == 0 && g.N % 128 == 0 &&
                       g.K % 16 == 0 && g.K >= 16 && g.ldp * 128 * 8 < (int64_t)1 << 31 && g.ldq * 128 * 8 < (int64_t)1 << 31 && g.ldc * 128 * 8 < (int64_t)1 << 31;
    if (!plain) return launch_gemm_nt<128, 128, 16, 2, 2>(g, stream, nullptr, 512, skip_first);
    const int ntm = g.M / 128, ntn = g.N / 128;
    const int tiles = (g.lower ? ntm * (ntm + 1) / 2 : ntm * ntn) - skip_first;
    if (tiles <= 0) return hipSuccess;
    g.tile_offset = skip_first;
    g.n_direct = tiles; g.split_p = 1; g.chunk_stages = g.K / 16; g.slab = nullptr; g.batch = 1; g.batch2 = 1;
    if (g_gemm_recorder) { g_gemm_recorder->fn(g_gemm_recorder->ctx, -1, 0, 0, 0, 0, g, tiles); return hipSuccess; }      // (bm = -1: this kernel)
    hipLaunchKernelGGL(chol_update_kernel, dim3(tiles), dim3(256), 0, stream, g);
    return hipGetLastError();
}

}  // namespace ipm
