// sparse_chol.h -- multifrontal SPARSE Cholesky of B = A D^2 A^T and its triangular solves (gfx950); SURVEY.md 8 row f3.
//
// What it replaces: scipy's spsolve on B (SuperLU, main.py:180 / :226) for the Netlib instances whose factor stays sparse
// under a fill-reducing row order (sparse_symbolic.h).  The dense-tile path of ipm_api.hip walks a chain of m / 128 pivot
// blocks (STOCFOR3: 131 steps, 1.8e11 flop inside the tile envelope); the same factor has 2.2e5 entries, 4e6 flop and an
// elimination tree 36 panels high.
//
// Structure (all index arrays built once on the host): the factor is a forest of PANELS (<= 32 consecutive columns with one
// row structure; r rows x w columns, row-major, own columns first).  Panel J's FRONT is the r x r matrix on its rows: the
// entries of B in its columns plus the update matrices of its children.  A task factors the panel (w guarded pivots, in
// LDS), writes L(:, J) and leaves U_J = F22 - L21 L21^T (p x p, p = r - w) for its parent.  Children are added one at a
// time in a fixed order (a panel with many children gets fan-in nodes, w = 0, that sum groups of eight): no atomics on data,
// results are bitwise reproducible.
//
// Scheduling: ONE launch per factorization / forward sweep / backward sweep.  Panels are grouped into TASKS on the host
// (whole subtrees below a work threshold; chains of the remaining top panels); workgroups draw tasks from an atomic
// counter in topological order (ascending top panel; the backward sweep descending), so a task only ever waits for tasks
// that were drawn before it, by workgroups that are running: the launch cannot deadlock whatever the occupancy.  Inside a
// task the panels run back to back in one workgroup; across tasks the hand-off is the release / acquire flag protocol of
// gemm_nt_f64.h (every storing wave drains, barrier, lane 0: agent release + flag; consumer: relaxed poll, agent acquire,
// barrier).  Spins are bounded and set the handle's time-out word (the host then reruns the launch with one workgroup,
// which never waits).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "potrf_f64.h"

namespace ipm {

constexpr int SPC_THREADS = 256;         // workgroup size for handles with fronts beyond 64 rows; one wave (64) below: four times the panels in flight
constexpr int SPC_PANEL = 7680;          // doubles of LDS for one panel (r x w <= SPC_PANEL; sparse_symbolic.h cuts panels to fit)
constexpr int SPC_BATCH = 8;             // independent entries a thread keeps in flight in the front assembly
constexpr int SPC_FRONT = 4096;          // doubles of LDS a whole front may take (r <= 64)
constexpr int SPC_WCAP = 32;             // columns per panel at most

// LDS row stride (doubles) of a panel in sp_chol_kernel.  A whole FRONT (r x r) is read row against row by consecutive lanes in
// its scalar update: an odd stride keeps those 64-bit reads on distinct banks (r = 32, 48, 64 ... would put every lane on the
// same one).  A PANEL (r x w) feeds its update U -= L21 L21^T to the matrix cores: columns padded with zeros to a multiple of
// four (one MFMA k-step) plus two -- stride = 2 mod 4, the 16 rows x 2 k of a half-wave's fragment read then fall on 32
// distinct bank pairs (the rule of gemm_nt_f64.h).
__host__ __device__ __forceinline__ int sp_front_stride(int r) { return r | 1; }
__host__ __device__ __forceinline__ int sp_panel_stride(int w) { return ((w + 3) & ~3) + 2; }
// doubles of LDS the factorization of one panel takes (front: r * r <= lds_front)
__host__ __device__ __forceinline__ long long sp_chol_lds_need(int r, int w, long long lds_front) {
    return (long long)r * r <= lds_front ? (long long)r * sp_front_stride(r) : (long long)r * sp_panel_stride(w);
}

struct SpNode {                          // one panel (64 bytes)
    int c0, w, r, nchild;
    int child0;                          // children: child[child0 .. child0 + nchild)
    int parent;                          // panel, or -1
    int wait_children;                   // 1: some child belongs to another task (factor / forward: wait for them; backward: publish)
    int publish;                         // 1: the parent belongs to another task (factor / forward: publish; backward: wait for it)
    long long rowptr, lptr, uptr, pad2;
};

constexpr int SPC_MAXCH = 12;            // children of a panel at most (sparse_symbolic.h inserts fan-in nodes beyond)
struct SpChild { long long uptr; long long relptr; int pc; int K; int ext; int pad; };   // relptr: rowptr + w of the child (crel, uvec)
struct SpRec {                           // everything a task needs to start on a panel, in TASK order (448 bytes)
    int J, c0, w, r;
    int nchild, parent, wait_children, publish;
    long long rowptr, lptr, uptr, pad;
    SpChild ch[SPC_MAXCH];
};

struct SpFactor {
    int nsn, ntask, m, pad;
    const SpNode* node;
    const SpRec* rec;                    // [nsn] in task order (tasknode)
    const int* rows;                     // panel rows (global row index)
    const int* child;
    const int* crel;                     // aligned with rows: for a >= w, position of rows[a] in the parent's front
    const int* taskptr;                  // [ntask + 1]
    const int* tasknode;                 // panels of a task, ascending
    const int* taskof;                   // [nsn]
    double* L;                           // panel values
    double* U;                           // update matrices (uptr)
    double* uvec;                        // forward sweep: update vector of J at uvec[rowptr[J] + w ..)
    double* dinv;                        // [m] 1 / L_cc, written by the factorization
    unsigned* flag;                      // [3 nsn]: chol, forward, backward (epoch of the last completed launch)
    unsigned* ctr;                       // [6]: {next task, exited workgroups} x 3
    unsigned* timeout;
    const int* done;
};

__device__ __forceinline__ bool sp_wait(const unsigned* flag, unsigned epoch, unsigned* timeout) {
    unsigned spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(2);
        ++spins;
        if (spins > ipm_spin_limit || ((spins & 1023u) == 1u && __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

// consumer side of a hand-off: lanes of wave 0 poll the flags; then that wave's agent acquire, its wait, the barrier
__device__ __forceinline__ void sp_acquire_barrier() {
    if (threadIdx.x < 64) {                  // the polling wave: its fence invalidates this CU's L1 for the whole workgroup
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}
// producer side: every storing wave drains, barrier, lane 0 releases and stores the flag
__device__ __forceinline__ void sp_publish(unsigned* flag, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Hand-off WITHOUT fences (template parameter SC1; opt-in, IPM_SP_SC1=1): every byte another workgroup will read is stored write-through
// (agent-scope relaxed atomic store = global_store ... sc1) and every load of such bytes is an sc1 load (bypasses this CU's L1);
// producer: every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, lane 0 stores the flag sc1; consumer: wave 0
// polls the flag sc1, workgroup barrier, sc1 loads.  (MI355X_MICROARCH.md, inter-workgroup visibility, "valid forms": conditions
// (1)-(3).)  Measured at STOCFOR3: a level of the tree costs ~10 us with the fence pair (release 1.7-6.5 us + acquire 1.7 us
// + poll) against 1.4-3.9 us for the panel itself; without the fences a sweep is 8-12 % faster.  SC1 = false (the default)
// keeps plain accesses and the release / acquire fences.
template <bool SC1> __device__ __forceinline__ double sp_ld(const double* p) {
    if constexpr (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool SC1> __device__ __forceinline__ void sp_st(double* p, double v) {
    if constexpr (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool SC1> __device__ __forceinline__ void sp_consume_barrier() {
    if constexpr (SC1) { asm volatile("" ::: "memory"); __syncthreads(); }
    else sp_acquire_barrier();
}
template <bool SC1> __device__ __forceinline__ void sp_signal(unsigned* flag, unsigned epoch) {
    if constexpr (SC1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        sp_publish(flag, epoch);
    }
}

// draw the next task (workgroup-uniform); the last workgroup to leave resets the two counters for the next launch
__device__ __forceinline__ int sp_next_task(unsigned* ctr, int* s_task) {
    __syncthreads();
    if (threadIdx.x == 0) *s_task = (int)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return *s_task;
}
__device__ __forceinline__ void sp_leave(unsigned* ctr) {
    if (threadIdx.x == 0) {
        const unsigned left = __hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == gridDim.x - 1) {         // every workgroup has made its last (failing) draw
            __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------ formation
// L <- the entries of B = A diag(d) A^T that fall into the panels (everything else in the panels: 0).  One thread per panel
// slot; slot e sums fcoef[t] d[fcol[t]] over its product list (columns ascending: fixed order).  fcoef = a_ij a_kj.
__global__ __launch_bounds__(256) void sp_form_kernel(const int* __restrict__ fptr, const int* __restrict__ fcol,
                                                      const double* __restrict__ fcoef, long long nslot,
                                                      const double* __restrict__ d, double* L, double* maxdiag, const int* done) {
    if (done && *done) return;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e == 0) *maxdiag = 0.0;                             // sp_maxdiag_kernel (next in the stream) takes an atomic max into it
    if (e >= nslot) return;
    const int t0 = fptr[e], t1 = fptr[e + 1];
    double acc = 0.0;
    for (int t = t0; t < t1; ++t) acc += fcoef[t] * d[fcol[t]];
    L[e] = acc;
}

// max diag(B) for the pivot-guard threshold: grid of 256-thread blocks, block maxima merged with an integer atomic max on the bit
// pattern (non-negative doubles order like their bits; max is order independent, so this atomic keeps results reproducible).
// *out was set to 0 by sp_form_kernel; negative or NaN diagonals never win, as in the dense path's maxdiag_kernel.
__global__ __launch_bounds__(256) void sp_maxdiag_kernel(const double* L, const long long* __restrict__ diagpos, int m, double* out,
                                                         const int* done) {
    if (done && *done) return;
    __shared__ double red[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double mx = 0.0;
    if (i < m) { const double v = L[diagpos[i]]; mx = (v > 0.0) ? v : 0.0; }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(red[0]));
}

// ------------------------------------------------------------------------------------------------------------ factorization
// Dynamic LDS: lds doubles.  A front with r^2 <= lds lives in LDS whole (children are added there, U_J leaves with one
// coalesced write); a larger front keeps only its panel (r x w) in LDS and accumulates U_J in memory.  Front assembly is
// CHILD-major: the update matrix of one child at a time is read contiguously and added at crel-mapped positions, a
// workgroup barrier between children -- the work is the children's entries, not (front entries) x (children), and the
// order of additions is fixed.  Fan-in nodes (w = 0) only do this step.  Everything a panel needs to start (its sizes,
// offsets and those of its children) sits in one record in task order (SpRec), read with uniform loads.
template <int NT, bool SC1>
__global__ __launch_bounds__(NT) void sp_chol_kernel(SpFactor f, unsigned epoch, const double* maxdiag, double eps,
                                                              double big, double shift_rel, int* fixed, int lds,
                                                              const SpRec* __restrict__ recs, int lvl_count,
                                                              const double* fwd_rhs, double* fwd_z, int fv_off) {
    // fwd_rhs != nullptr: the forward substitution L z = fwd_rhs rides on the factorization -- both walk the tree leaves first,
    // the panel is in LDS anyway, and the sweep's own pass (one hand-off per level of the tree) is saved.  Same arithmetic in
    // the same order as sp_fwd_kernel (bitwise equal z).  fv_off: doubles of dynamic LDS in front of the r-vector it needs.
    // lvl_count > 0: LEVEL mode -- this launch owns the lvl_count panels recs[0 .. lvl_count) of one level of the tree (all
    // their children were finished by earlier launches): no task queue, no waits, no flags.  lvl_count == 0: the whole tree in
    // one launch (recs = f.rec in task order), tasks drawn from the counter, hand-offs through flags.
    if (f.done && *f.done) return;
    extern __shared__ __attribute__((aligned(16))) double P[];
    __shared__ double rs[SPC_WCAP];
    __shared__ int s_task;
    const int tid = threadIdx.x;
    const double md = *maxdiag;
    const double thresh = eps * md, shift = shift_rel * md;
    unsigned* flag = f.flag;
    const bool level = lvl_count > 0;
    int nfix = 0;
    for (int round = 0;; ++round) {
        int task = 0, tn0, tn1;
        if (level) {
            tn0 = (int)blockIdx.x + round * (int)gridDim.x;
            if (tn0 >= lvl_count) break;
            tn1 = tn0 + 1;
        } else {
            task = sp_next_task(f.ctr, &s_task);
            if (task >= f.ntask) break;
            tn0 = f.taskptr[task]; tn1 = f.taskptr[task + 1];
        }
        for (int tn = tn0; tn < tn1; ++tn) {
            const SpRec& rc = recs[tn];
            const int r = rc.r, w = rc.w, p = r - w, nchild = rc.nchild;
            const bool front = r * r <= lds;
            const int ldp = front ? sp_front_stride(r) : sp_panel_stride(w);
            if (!level && rc.wait_children) {
                if (tid < nchild && rc.ch[tid].ext) (void)sp_wait(flag + rc.ch[tid].K, epoch, f.timeout);   // (a time-out poisons the results; the host reruns the launch)
                sp_consume_barrier<SC1>();
            } else {
                __syncthreads();                            // the previous panel of this task is complete (its U is in memory)
            }
            double* Lp = f.L + rc.lptr;
            double* Up = f.U + rc.uptr;
            const bool kids = nchild > 0;
            if (front) {
                for (int idx = tid; idx < r * r; idx += NT) {
                    const int a = idx / r, b = idx - a * r;
                    P[a * ldp + b] = b < w ? Lp[a * w + b] : 0.0;
                }
            } else {
                // 32 column lanes x NT / 32 row groups; the pad columns w .. ldp - 3 are the zeros of the last MFMA k-step
                const int bl = tid & 31, rg = tid >> 5;
                for (int a = rg; a < r; a += NT / 32) {
                    if (bl < w) P[a * ldp + bl] = Lp[a * w + bl];
                    else if (bl < ldp - 2) P[a * ldp + bl] = 0.0;
                }
                if (kids) for (int idx = tid; idx < p * p; idx += NT) sp_st<SC1>(Up + idx, 0.0);
            }
            __syncthreads();
            if (shift != 0.0) {
                if (tid < w) P[tid * ldp + tid] += shift;
                __syncthreads();
            }
            for (int t = 0; t < nchild; ++t) {
                const int pc = rc.ch[t].pc;
                const double* Uc = f.U + rc.ch[t].uptr;
                const int* rel = f.crel + rc.ch[t].relptr;
                // batches of SPC_BATCH entries per thread: all source loads, then all target loads, then the stores (the
                // targets of one child are distinct, which the compiler cannot know: written out so that the memory
                // latencies of a batch overlap instead of forming one read-modify-write chain per entry)
                for (int e0 = tid; e0 < pc * pc; e0 += SPC_BATCH * NT) {
                    double v[SPC_BATCH], old[SPC_BATCH];
                    long long tg[SPC_BATCH];                 // >= 0: slot of U_J in memory; -1: none; <= -2: LDS slot -(tg + 2)
#pragma unroll
                    for (int k = 0; k < SPC_BATCH; ++k) {
                        const int e = e0 + k * NT;
                        tg[k] = -1; v[k] = 0.0;
                        if (e < pc * pc) {
                            const int i = e / pc, j = e - i * pc;
                            if (j <= i) {
                                v[k] = sp_ld<SC1>(Uc + e);
                                const int a = rel[i], b = rel[j];
                                tg[k] = (front || b < w) ? -(long long)(a * ldp + b) - 2 : (long long)(a - w) * p + (b - w);
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < SPC_BATCH; ++k) old[k] = tg[k] >= 0 ? sp_ld<SC1>(Up + tg[k]) : 0.0;
#pragma unroll
                    for (int k = 0; k < SPC_BATCH; ++k) {
                        if (tg[k] >= 0) sp_st<SC1>(Up + tg[k], old[k] + v[k]);
                        else if (tg[k] <= -2) P[-(tg[k] + 2)] += v[k];
                    }
                }
                __syncthreads();
            }
            // ---- w guarded pivots on the panel, right-looking inside the panel.  ONE barrier per column: the columns stay
            //      unscaled while the loop runs (the update of column b by column c is P[a][c] P[b][c] / pivot_c) and are
            //      scaled by 1 / sqrt(pivot) at the end.  Threads = 32 column lanes x NT / 32 row groups (no integer divisions).
            {
                const int bl = tid & 31, rg = tid >> 5;
                for (int c = 0; c < w; ++c) {
                    double pv = P[c * ldp + c];
                    const bool bad = !(pv > thresh);
                    if (bad) pv = big;
                    double root, rinv;
                    sqrt_rsqrt(pv, root, rinv);                    // v_rsq_f64 + one Halley step (potrf_f64.h): the pivot chain is serial
                    if (tid == 0) { rs[c] = rinv; if (bad) { ++nfix; P[c * ldp + c] = big; } }
                    const double ipv = rinv * rinv;
                    const int b = c + 1 + bl;
                    if (b < w) {
                        const double fb = P[b * ldp + c] * ipv;
                        for (int a = b + rg; a < r; a += NT / 32) P[a * ldp + b] -= P[a * ldp + c] * fb;     // rows a >= b
                    }
                    __syncthreads();
                }
                for (int idx = tid; idx < r * w; idx += NT) {
                    const int a = idx / w, b = idx - a * w;
                    if (a >= b) P[a * ldp + b] *= rs[b];               // diagonal: pivot / sqrt(pivot) = sqrt(pivot)
                }
                __syncthreads();
            }
            if (fwd_rhs) {
                double* fv = P + fv_off;
                double* uv = f.uvec + rc.rowptr;
                for (int a = tid; a < r; a += NT) fv[a] = a < w ? fwd_rhs[rc.c0 + a] : 0.0;
                double cu[SPC_MAXCH];
                int cr[SPC_MAXCH];
#pragma unroll
                for (int t = 0; t < SPC_MAXCH; ++t) {
                    cr[t] = -1; cu[t] = 0.0;
                    if (t < nchild) {
                        const int pc = rc.ch[t].pc;
                        if (tid < pc) { cu[t] = sp_ld<SC1>(f.uvec + rc.ch[t].relptr + tid); cr[t] = f.crel[rc.ch[t].relptr + tid]; }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int t = 0; t < SPC_MAXCH; ++t) {
                    if (t < nchild) {
                        if (cr[t] >= 0) fv[cr[t]] += cu[t];
                        const int pc = rc.ch[t].pc;
                        if (pc > NT) {
                            const double* uc = f.uvec + rc.ch[t].relptr;
                            const int* rel = f.crel + rc.ch[t].relptr;
                            for (int i = NT + tid; i < pc; i += NT) fv[rel[i]] += sp_ld<SC1>(uc + i);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    }
                }
                if (tid < 64 && w > 0) {                      // w x w lower triangular solve in the registers of wave 0 (as sp_fwd_kernel)
                    const int ln = tid < w ? tid : 0;
                    double fa = fv[ln];
                    const double di = rs[ln];
                    for (int c = 0; c < w; ++c) {
                        const double zc = readlane_f64(fa, c) * readlane_f64(di, c);
                        if (tid == c) fa = zc;
                        else if (tid > c && tid < w) fa -= P[tid * ldp + c] * zc;
                    }
                    if (tid < w) fv[tid] = fa;
                }
                __syncthreads();
                for (int a = tid; a < r; a += NT) {
                    if (a < w) { fwd_z[rc.c0 + a] = fv[a]; }
                    else {
                        const double* row = P + a * ldp;
                        double dot = 0.0;
                        for (int c = 0; c < w; ++c) dot += row[c] * fv[c];
                        sp_st<SC1>(uv + a, fv[a] - dot);
                    }
                }
            }
            // ---- U = (children's sum) - L21 L21^T, panel to memory
            if (front) {
                for (int idx = tid; idx < p * p; idx += NT) {
                    const int i = idx / p, j = idx - i * p;
                    if (j > i) continue;
                    const double* ra = P + (w + i) * ldp;
                    const double* rb = P + (w + j) * ldp;
                    double dot = 0.0;
                    for (int c = 0; c < w; ++c) dot += ra[c] * rb[c];
                    sp_st<SC1>(Up + idx, ra[w + j] - dot);
                }
            } else if (w > 0 && p > 0) {
                // matrix cores: one 16 x 16 tile of U per wave pass, v_mfma_f64_16x16x4_f64 over the w columns (zero padded to a
                // multiple of 4); operand fragments straight from the LDS panel.  Lane maps: gemm_nt_f64.h.
                const int lane = tid & 63, wv = tid >> 6, fr = lane & 15, fk = lane >> 4;
                const int nT = (p + 15) >> 4, ksteps = (w + 3) >> 2;
                for (int t = wv; t < nT * (nT + 1) / 2; t += NT / 64) {
                    int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                    while ((I + 1) * (I + 2) / 2 <= t) ++I;
                    while (I * (I + 1) / 2 > t) --I;
                    const int J = t - I * (I + 1) / 2;
                    const int ia = 16 * I + fr, jb = 16 * J + fr;
                    const double* pa = P + (w + (ia < p ? ia : p - 1)) * ldp + fk;
                    const double* pb = P + (w + (jb < p ? jb : p - 1)) * ldp + fk;
                    const double ma = ia < p ? 1.0 : 0.0, mb = jb < p ? 1.0 : 0.0;      // rows beyond the front contribute zeros
                    f64x4 acc0 = (f64x4){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
                    int kk = 0;
                    for (; kk + 1 < ksteps; kk += 2) {
                        const double a0 = pa[4 * kk] * ma, b0 = pb[4 * kk] * mb, a1 = pa[4 * kk + 4] * ma, b1 = pb[4 * kk + 4] * mb;
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc1, 0, 0, 0);
                    }
                    if (kk < ksteps) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * kk] * ma, pb[4 * kk] * mb, acc0, 0, 0, 0);
                    // accumulator register q of lane l: U(16 I + (l >> 4) + 4 q, 16 J + (l & 15))
                    const int j = 16 * J + fr;
                    double old[4];
                    bool on[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = 16 * I + fk + 4 * q;
                        on[q] = i < p && j <= i;
                        old[q] = (on[q] && kids) ? sp_ld<SC1>(Up + (long long)i * p + j) : 0.0;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int i = 16 * I + fk + 4 * q;
                        if (on[q]) sp_st<SC1>(Up + (long long)i * p + j, old[q] - (acc0[q] + acc1[q]));
                    }
                }
            }
            for (int idx = tid; idx < r * w; idx += NT) { const int a = idx / w, b = idx - a * w; Lp[idx] = P[a * ldp + b]; }
            if (tid < w) f.dinv[rc.c0 + tid] = rs[tid];             // 1 / L_cc for the substitutions (a multiply per step instead of a divide)
            if (!level && rc.publish) sp_signal<SC1>(flag + rc.J, epoch);
        }
    }
    if (tid == 0 && nfix) atomicAdd(fixed, nfix);
    if (!level) sp_leave(f.ctr);
}

// ------------------------------------------------------------------------------------------------------------ L z = rhs
// z may alias rhs.  Panel J: f = rhs(J's columns) + children's update vectors (child-major, as the factorization);
// z_J = L_JJ^{-1} f_top; the rows below get f_below - L_21 z_J, handed to the parent.  Dynamic LDS: rmax + 1024 doubles.
template <int NT, bool SC1>
__global__ __launch_bounds__(NT) void sp_fwd_kernel(SpFactor f, unsigned epoch, const double* rhs, double* z, int rmax,
                   const SpRec* __restrict__ recs, int lvl_count) {
    if (f.done && *f.done) return;
    extern __shared__ __attribute__((aligned(16))) double fv[];
    double* D = fv + rmax;
    __shared__ int s_task;
    const int tid = threadIdx.x;
    unsigned* flag = f.flag + f.nsn;
    unsigned* ctr = f.ctr + 2;
    const bool level = lvl_count > 0;                        // (see sp_chol_kernel)
    for (int round = 0;; ++round) {
        int task = 0, tn0, tn1;
        if (level) {
            tn0 = (int)blockIdx.x + round * (int)gridDim.x;
            if (tn0 >= lvl_count) break;
            tn1 = tn0 + 1;
        } else {
            task = sp_next_task(ctr, &s_task);
            if (task >= f.ntask) break;
            tn0 = f.taskptr[task]; tn1 = f.taskptr[task + 1];
        }
        for (int tn = tn0; tn < tn1; ++tn) {
            const SpRec& rc = recs[tn];
            const int r = rc.r, w = rc.w, nchild = rc.nchild;
            if (!level && rc.wait_children) {
                if (tid < nchild && rc.ch[tid].ext) (void)sp_wait(flag + rc.ch[tid].K, epoch, f.timeout);
                sp_consume_barrier<SC1>();
            } else {
                __syncthreads();
            }
            const double* Lp = f.L + rc.lptr;
            double* uv = f.uvec + rc.rowptr;
            for (int a = tid; a < r; a += NT) fv[a] = a < w ? rhs[rc.c0 + a] : 0.0;
            for (int idx = tid; idx < w * w; idx += NT) D[idx] = Lp[idx];
            // children's update vectors: every load is issued before the first add (a child shorter than the workgroup gives each
            // thread at most one entry), the adds then go child by child with LDS-only barriers -- one memory latency for the
            // whole extend-add instead of one per child
            double cu[SPC_MAXCH];
            int cr[SPC_MAXCH];
            bool all_short = true;
#pragma unroll
            for (int t = 0; t < SPC_MAXCH; ++t) {
                cr[t] = -1; cu[t] = 0.0;
                if (t < nchild) {
                    const int pc = rc.ch[t].pc;
                    all_short = all_short && pc <= NT;
                    if (tid < pc) { cu[t] = sp_ld<SC1>(f.uvec + rc.ch[t].relptr + tid); cr[t] = f.crel[rc.ch[t].relptr + tid]; }
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < SPC_MAXCH; ++t) {
                if (t < nchild) {
                    if (cr[t] >= 0) fv[cr[t]] += cu[t];
                    const int pc = rc.ch[t].pc;
                    if (pc > NT) {
                        const double* uc = f.uvec + rc.ch[t].relptr;
                        const int* rel = f.crel + rc.ch[t].relptr;
                        for (int i = NT + tid; i < pc; i += NT) fv[rel[i]] += sp_ld<SC1>(uc + i);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
            }
            (void)all_short;
            if (tid < 64 && w > 0) {                          // w x w lower triangular solve in the registers of wave 0 (w <= 32)
                const int ln = tid < w ? tid : 0;
                double fa = fv[ln];
                const double di = f.dinv[rc.c0 + ln];
                for (int c = 0; c < w; ++c) {
                    const double zc = readlane_f64(fa, c) * readlane_f64(di, c);
                    if (tid == c) fa = zc;
                    else if (tid > c && tid < w) fa -= D[tid * w + c] * zc;
                }
                if (tid < w) fv[tid] = fa;
            }
            __syncthreads();
            for (int a = tid; a < r; a += NT) {
                if (a < w) { z[rc.c0 + a] = fv[a]; }
                else {
                    const double* row = Lp + (long long)a * w;
                    double dot = 0.0;
                    for (int c = 0; c < w; ++c) dot += row[c] * fv[c];
                    sp_st<SC1>(uv + a, fv[a] - dot);
                }
            }
            if (!level && rc.publish) sp_signal<SC1>(flag + rc.J, epoch);
        }
    }
    if (!level) sp_leave(ctr);
}

// ------------------------------------------------------------------------------------------------------------ L^T x = z
// x may alias z.  Panels in DESCENDING order: x_J = L_JJ^{-T} (z_J - L_21^T x(rows below)); the rows below belong to
// ancestors, whose x is final once the parent's flag is up.
template <int NT, bool SC1>
__global__ __launch_bounds__(NT) void sp_bwd_kernel(SpFactor f, unsigned epoch, const double* z, double* x, const SpRec* __restrict__ recs, int lvl_count) {
    if (f.done && *f.done) return;
    constexpr int NSL = NT / 32;
    __shared__ double part[NSL * SPC_WCAP];
    __shared__ double D[SPC_WCAP * SPC_WCAP];
    __shared__ int s_task;
    const int tid = threadIdx.x;
    unsigned* flag = f.flag + 2 * f.nsn;
    unsigned* ctr = f.ctr + 4;
    const bool level = lvl_count > 0;                        // (see sp_chol_kernel; the host launches the levels top down)
    for (int round = 0;; ++round) {
        int tn0, tn1;
        if (level) {
            tn0 = (int)blockIdx.x + round * (int)gridDim.x;
            if (tn0 >= lvl_count) break;
            tn1 = tn0 + 1;
        } else {
            const int draw = sp_next_task(ctr, &s_task);
            if (draw >= f.ntask) break;
            const int task = f.ntask - 1 - draw;
            tn0 = f.taskptr[task]; tn1 = f.taskptr[task + 1];
        }
        for (int tn = tn1 - 1; tn >= tn0; --tn) {
            const SpRec& rc = recs[tn];
            const int r = rc.r, w = rc.w, p = r - w;
            if (!level && rc.publish) {                              // the parent belongs to another task
                if (tid == 0) (void)sp_wait(flag + rc.parent, epoch, f.timeout);
                sp_consume_barrier<SC1>();
            } else {
                __syncthreads();
            }
            if (w > 0) {
                const double* Lp = f.L + rc.lptr;
                const int* rows = f.rows + rc.rowptr;
                // g[c] = sum over the rows below of L[a][c] x[rows[a]]: NSL slices of rows per column, summed in slice order
                {
                    const int c = tid & (SPC_WCAP - 1), sl = tid >> 5;          // NT threads = 32 columns x NSL slices
                    double acc = 0.0;
                    if (c < w) for (int i = sl; i < p; i += NSL) acc += Lp[(long long)(w + i) * w + c] * sp_ld<SC1>(x + rows[w + i]);
                    part[sl * SPC_WCAP + c] = acc;
                }
                for (int idx = tid; idx < w * w; idx += NT) D[idx] = Lp[idx];
                __syncthreads();
                if (tid < 64) {                                   // L_JJ^T x = g in the registers of wave 0 (w <= 32)
                    double ga = 0.0;
                    if (tid < w) {
                        double s = 0.0;
                        for (int sl = 0; sl < NSL; ++sl) s += part[sl * SPC_WCAP + tid];
                        ga = z[rc.c0 + tid] - s;
                    }
                    const double di = f.dinv[rc.c0 + (tid < w ? tid : 0)];
                    for (int c = w - 1; c >= 0; --c) {
                        const double xc = readlane_f64(ga, c) * readlane_f64(di, c);
                        if (tid == c) ga = xc;
                        else if (tid < c) ga -= D[c * w + tid] * xc;
                    }
                    if (tid < w) sp_st<SC1>(x + rc.c0 + tid, ga);
                }
            }
            if (!level && rc.wait_children) sp_signal<SC1>(flag + rc.J, epoch);       // some child belongs to another task: it waits for this x
        }
    }
    if (!level) sp_leave(ctr);
}

// dense image of the factor (ipm_get_factor): out must be zeroed; one workgroup per panel
__global__ __launch_bounds__(256) void sp_expand_kernel(SpFactor f, double* out, long long ld) {
    const SpNode nd = f.node[blockIdx.x];
    const double* Lp = f.L + nd.lptr;
    const int* rows = f.rows + nd.rowptr;
    for (int idx = threadIdx.x; idx < nd.r * nd.w; idx += 256) {
        const int a = idx / nd.w, b = idx - a * nd.w;
        if (a >= b) out[(long long)rows[a] * ld + nd.c0 + b] = Lp[idx];
    }
}

}  // namespace ipm
