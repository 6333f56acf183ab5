// adat_syrk_2set.h -- A/B copy (tools/syrk_bench.hip only) of a MEASURED AND REJECTED variant of the formation kernel: two
// staging register sets, inline-asm buffer loads with hand-counted vmcnt, loads 1.75 stages in flight (241 VGPRs).
// Against the shipped one-set kernel (interiorpointmethod_amd/csrc/adat_syrk_f64.h): 2.016 vs 2.011 ms at 4096 x 8192,
// 15.67 vs 15.56 ms at 8192 x 16384, 128.3 vs 124.4 ms at 16384 x 32768 (profiles/r02_syrk_ab_3way.log).
// adat_syrk_f64.h -- the dominant kernel of the dense hot path: B = A diag(d) A^T, lower 128 x 128 tiles,
// fp64 MFMA (v_mfma_f64_16x16x4_f64) for gfx950.  Replaces the two scipy SpGEMMs of main.py:224 (reference repo).
//
// Same tiling, LDS image, summation order and split-K tail as the generic gemm_nt_f64_kernel<128,128,16,2,2,true>
// (results are bit-identical to it); what differs is the schedule of one K stage (BK = 16, four MFMA k-steps):
//
//   * Fragment reads are software pipelined one k-step ahead through two register sets (Fa, Fb): the ds_reads of
//     k-step kk+1 are issued before the 16 MFMAs of k-step kk, so no LDS latency is exposed inside a stage.  The
//     generic kernel reads all fragments of two k-steps, waits, multiplies (two exposed LDS round trips per stage).
//   * The stage barrier sits BEFORE the last k-step instead of after it: the next stage's operands are written to
//     the other LDS buffer during k-step 2, the barrier follows, and the first fragments of the next stage are
//     fetched while the 16 MFMAs of k-step 3 run -- the MFMA stream of a wave continues across the stage boundary.
//   * Operands are fetched with buffer loads (one VGPR offset per thread, row and k offsets in SGPRs) instead of
//     nine 64-bit per-lane pointers: the kernel fits its 256 VGPRs without scratch.
//
// fp64 MFMA lane maps: see gemm_nt_f64.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../interiorpointmethod_amd/csrc/adat_syrk_f64.h"

namespace ipm {

struct AdatSyrk2 {
    const double* A; int ld;        // row-major [mp][ld], zero padded; K = 16 nk columns are used
    const double* w;                // scaling d, length >= 16 nk
    double* C; int ldc;             // B, lower tiles written
    int nk;                         // K / 16
    int unit_diag_from;             // C[r][r] = 1 for r >= unit_diag_from (padding rows)
    const int* done;
    const int* tile_order;          // logical tile -> (ti << 16 | tj)
    int n_direct, split_p, chunk_stages;   // split-K of the tail tiles, as in GemmNT
    double* slab;
};



// Raw buffer resource: 48-bit base, stride 0, num_records in bytes, gfx9 raw-buffer flags.  Out-of-range offsets
// return zeros instead of faulting.
__device__ __forceinline__ i32x4_t make_rsrc(const void* base, unsigned bytes) {
    const uint64_t p = (uint64_t)base;
    i32x4_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)p);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((p >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

// The staging loads are INLINE ASM on purpose: two register sets are in flight at once and the counted waits below
// (vmcnt(15) ... vmcnt(9)) must leave the younger set alone.  With compiler-tracked loads hipcc merges the in-flight
// states of the unrolled stage pair at the loop head and falls back to vmcnt(0) in one of the two copies (cdna guide
// 5.7: loads hidden in inline asm, both queues counted by hand).
__device__ __forceinline__ f64x2 buf_load2_f64x2(i32x4_t r, int voff, int soff) {
    f64x2 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(r), "s"(soff) : "memory");
    return v;
}
#define IPM_WAIT_VMCNT(N) do { asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__global__ __launch_bounds__(256, 2) void adat_syrk_2set_kernel(AdatSyrk2 g) {
    constexpr int BM = 128, BK = 16, LDT = BK + 2, RSTEP = 32;
    if (g.done && *g.done) return;

    __shared__ __attribute__((aligned(16))) double lds[2 * (BM + BM) * LDT];
    double* Ps = lds;                           // [2][128][LDT]
    double* Qs = lds + 2 * BM * LDT;            // [2][128][LDT]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int kbeg = 0, kend = g.nk;
    double* slab_out = nullptr;
    int bid;
    if ((int)blockIdx.x < g.n_direct) {
        bid = xcd_remap(blockIdx.x, g.n_direct);
    } else {
        const int r = (int)blockIdx.x - g.n_direct;
        bid = g.n_direct + r / g.split_p;
        kbeg = (r % g.split_p) * g.chunk_stages;
        kend = min(kend, kbeg + g.chunk_stages);
        slab_out = g.slab + (size_t)r * (BM * BM);
    }
    const int packed = g.tile_order[bid];
    const int ti = packed >> 16, tj = packed & 0xffff;
    const int row0 = ti * BM, col0 = tj * BM;

    // buffer resources (raw SGPR quads): base = first row of the operand panel (wave uniform), 128 rows of ld doubles
    const unsigned panel_bytes = (unsigned)BM * (unsigned)g.ld * 8u;
    const i32x4_t rP = make_rsrc(g.A + (int64_t)row0 * g.ld, panel_bytes);
    const i32x4_t rQ = make_rsrc(g.A + (int64_t)col0 * g.ld, panel_bytes);
    const i32x4_t rW = make_rsrc(g.w, (unsigned)g.nk * BK * 8u);

    // staging: thread -> 16-byte chunk ch0 of rows r0t + 32 i (i < 4) of both panels.  TWO register sets: the loads of
    // stage l+3 are issued right after the barrier of stage l (into the set whose contents were just written to LDS) and
    // are written to LDS during k-step 2 of stage l+2, i.e. they are in flight for ~1.75 stages (~7000 cycles): inside
    // the solver A comes from HBM (~2 us under load), and with one set (0.75 stage in flight) the kernel ran 8 % slower
    // there than back to back on a cache-warm A.  The loads are issued UNCONDITIONALLY (past the last stage they fall
    // outside the buffer resource or into the next row and are never stored): a conditional issue makes the compiler
    // assume the younger set may be absent and wait vmcnt(0) at every LDS write.
    const int ch0 = tid & 7, r0t = tid >> 3;
    const int voff = (r0t * g.ld + ch0 * 2) * 8;
    const int rstep_bytes = RSTEP * g.ld * 8;
    f64x2 pr[2][4], qr[2][4], wr[2];
    auto issue_loads = [&](int set, int kt) {
        const int kb = kt * BK * 8;
        wr[set] = buf_load2_f64x2(rW, ch0 * 16, kb);            // issue order = wait order: w, then (q_i, p_i) pairs
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qr[set][i] = buf_load2_f64x2(rQ, voff, kb + i * rstep_bytes);
            pr[set][i] = buf_load2_f64x2(rP, voff, kb + i * rstep_bytes);
        }
    };
    const int st_off = r0t * LDT + ch0 * 2;                    // LDS element offset of this thread's chunk, row group 0
    auto store_q = [&](int set, int i) {                       // staging set `set` -> LDS buffer `set`
        f64x2 v = qr[set][i];
        v.x *= wr[set].x; v.y *= wr[set].y;
        *reinterpret_cast<f64x2*>(Qs + set * BM * LDT + i * RSTEP * LDT + st_off) = v;
    };
    auto store_p = [&](int set, int i) {
        *reinterpret_cast<f64x2*>(Ps + set * BM * LDT + i * RSTEP * LDT + st_off) = pr[set][i];
    };

    const int fr = lane & 15, fk = lane >> 4;
    const int fa_off = (wm * 64 + fr) * LDT + fk, fb_off = (wn * 64 + fr) * LDT + fk;
    double fa[2][4], fb[2][4];                                  // two fragment register sets
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

    // One K stage.  P (compile time) = parity of the local stage index l = kt - kbeg: stage l lives in LDS buffer P, the
    // operands of stage l+1 wait in staging set P^1 (written to LDS buffer P^1 here), those of stage l+2 are in flight
    // in set P, and set P^1 is refilled with stage l+3 after the barrier.
    const int nst = kend - kbeg;
    auto stage = [&](auto Ptag, int l) {
        constexpr int P = decltype(Ptag)::value;
        const bool more = l + 1 < nst;
        read_frags(1, P, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(0);                                              // k-step 0
        __builtin_amdgcn_sched_barrier(0);
        read_frags(0, P, 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(1);                                              // k-step 1
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1, P, 3);
        __builtin_amdgcn_sched_barrier(0);
        // k-step 2, with the next stage's operands written to the other LDS buffer between its MFMA rows
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
            // set P^1 (older): w, q0, p0, ..., q3, p3, then the 9 loads of set P: row i needs everything up to p_i
            if (i == 0) IPM_WAIT_VMCNT(15); else if (i == 1) IPM_WAIT_VMCNT(13); else if (i == 2) IPM_WAIT_VMCNT(11); else IPM_WAIT_VMCNT(9);
            if (more) { store_q(P ^ 1, i); store_p(P ^ 1, i); }
            __builtin_amdgcn_sched_barrier(0);
        }
        // LDS only: the barrier must not drain the global loads in flight (raw s_barrier)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // writes of stage l+1 visible; reads of stage l issued
        __builtin_amdgcn_sched_barrier(0);
        issue_loads(P ^ 1, kbeg + l + 3);
        if (more) read_frags(0, P ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(1);                                              // k-step 3
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: stage 0 into LDS buffer 0, loads of stage 1 in flight in set 1, first fragments in Fa
    issue_loads(0, kbeg);
    issue_loads(1, kbeg + 1);
    IPM_WAIT_VMCNT(9);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_q(0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_p(0, i);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issue_loads(0, kbeg + 2);
    read_frags(0, 0, 0);
    // (exits only, no conditional call: the back edge must always come from an odd stage, otherwise the compiler's
    // vmcnt bookkeeping merges two different in-flight states at the loop head and falls back to vmcnt(0))
    for (int l = 0;; l += 2) {
        stage(std::integral_constant<int, 0>{}, l);
        if (l + 1 >= nst) break;
        stage(std::integral_constant<int, 1>{}, l + 1);
        if (l + 2 >= nst) break;
    }
    // the run-ahead loads past the last stage are never consumed, so their registers are dead to the compiler: drain
    // them before anything else may be allocated there
    IPM_WAIT_VMCNT(0);

    // ---- epilogue: D[row=(l>>4)+4q][col=l&15]
    if (slab_out) {                                             // split-K partial: raw tile, summed later
        double* sb = slab_out + (wm * 64 + fk) * BM + wn * 64 + fr;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) sb[(i * 16 + 4 * q) * BM + j * 16] = acc[i][j][q];
        return;
    }
    double* cbase = g.C + (int64_t)(row0 + wm * 64 + fk) * g.ldc + col0 + wn * 64 + fr;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = row0 + wm * 64 + i * 16 + fk + 4 * q;
                const int c = col0 + wn * 64 + j * 16 + fr;
                double v = acc[i][j][q];
                if (g.unit_diag_from >= 0 && r == c && r >= g.unit_diag_from) v = 1.0;
                cbase[(int64_t)(i * 16 + 4 * q) * g.ldc + j * 16] = v;
            }
}

// B = A diag(w) A^T, lower 128 x 128 tiles in `tile_order`; tiles beyond a multiple of `slots` resident workgroups are
// split along K into slabs and summed in fixed order by splitk_reduce_kernel (same policy as launch_gemm_nt).
inline hipError_t launch_adat_syrk_2set(const double* A, int64_t ld, const double* w, double* C, int64_t ldc, int M, int K,
                                   int unit_diag_from, const int* done, const int* tile_order, hipStream_t stream,
                                   double* slab, int slots = 512) {
    const int nt = M / 128, tiles = nt * (nt + 1) / 2, nk = K / 16;
    if (tiles <= 0 || nk <= 0) return hipSuccess;
    AdatSyrk2 g;
    g.A = A; g.ld = (int)ld; g.w = w; g.C = C; g.ldc = (int)ldc; g.nk = nk; g.unit_diag_from = unit_diag_from;
    g.done = done; g.tile_order = tile_order;
    g.n_direct = tiles; g.split_p = 1; g.chunk_stages = nk; g.slab = nullptr;
    int grid = tiles;
    if (slab && slots > 0) {
        const int tail = tiles % slots;
        if (tail > 0 && nk >= 16) {
            int p = slots / tail;
            if (p > nk / 8) p = nk / 8;                      // keep >= 8 stages per chunk
            if ((long)tail * p > kSlabTiles) p = kSlabTiles / tail;
            if (p >= 2) {
                const int per = (nk + p - 1) / p;
                p = (nk + per - 1) / per;
                g.n_direct = tiles - tail; g.split_p = p; g.chunk_stages = per; g.slab = slab;
                grid = g.n_direct + tail * p;
            }
        }
    }
    hipLaunchKernelGGL(adat_syrk_2set_kernel, dim3(grid), dim3(256), 0, stream, g);
    if (g.slab) {
        GemmNT r;
        memset(&r, 0, sizeof r);
        r.C = C; r.ldc = ldc; r.alpha = 1.0; r.beta = 0.0; r.lower = 1; r.unit_diag_from = unit_diag_from; r.done = done;
        r.n_direct = g.n_direct; r.split_p = g.split_p; r.slab = slab; r.tile_order = tile_order; r.N = M;
        hipLaunchKernelGGL((splitk_reduce_kernel<128, 128>), dim3(128 * 128 / 1024, tiles - g.n_direct), dim3(256), 0, stream, r);
    }
    return hipGetLastError();
}

}  // namespace ipm
