// crit_probe.hip -- where do the ~10 us of a critical-path kernel of the Cholesky chain go?  Runs the chain
// [producer on other CUs] -> crit_panel_kernel -> crit_syrk_kernel back to back on one stream with s_memrealtime stamps
// (100 MHz) at kernel entry, after the loads, after the MFMAs and after the stores, plus the host-side event time of the
// pair; variants: generic 4-stage kernels vs the single-stage register kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/crit_probe tools/crit_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../interiorpointmethod_amd/csrc/gemm_nt_f64.h"
#include "../interiorpointmethod_amd/csrc/chol_crit_f64.h"
using namespace ipm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void touch_kernel(double* p, int64_t ld, int rows, double v) {          // "previous kernel": rewrites the panel from many CUs
    const int r = blockIdx.x, c = threadIdx.x;
    if (r < rows) p[(int64_t)r * ld + c] = p[(int64_t)r * ld + c] * 0.999 + v;
}
__global__ void stamp_kernel(long long* out) { if (threadIdx.x == 0) out[blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime(); }

// instrumented copy of crit_syrk_kernel (stamps from wave 0 of every workgroup)
__global__ __launch_bounds__(256) void crit_syrk_stamped(CritStep g, long long* st) {
    long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    const int lane = threadIdx.x & 63;
    const int t = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
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
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
    f64x4 acc = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc, 0, 0, 0);
    double keep = acc[0] + acc[1] + acc[2] + acc[3];
    asm volatile("" :: "v"(keep));
    long long t2 = (long long)__builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int q = 0; q < 4; ++q) crow[(int64_t)(4 * q) * g.ldc] = -1.0 * acc[q] + 1.0 * cold[q];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t3 = (long long)__builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { long long* o = st + blockIdx.x * 4; o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; }
}

int main() {
    const int mp = 4096, NBk = 128;
    double *B, *inv; long long *st, *tk;
    CK(hipMalloc(&B, (size_t)mp * mp * 8)); CK(hipMalloc(&inv, 128 * 128 * 8)); CK(hipMalloc(&st, 64 * 8 * 4)); CK(hipMalloc(&tk, 64));
    std::vector<double> h((size_t)mp * mp);
    for (size_t i = 0; i < h.size(); ++i) h[i] = ((i * 2654435761u) % 1000) * 1e-3 - 0.5;
    CK(hipMemcpy(B, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    std::vector<double> hi(128 * 128, 0.0);
    for (int i = 0; i < 128; ++i) for (int j = 0; j <= i; ++j) hi[i * 128 + j] = (i == j) ? 1.0 : 1e-3 * ((i + j) % 7);
    CK(hipMemcpy(inv, hi.data(), hi.size() * 8, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double* panel = B + (int64_t)1024 * mp + 896;             // some block row / column inside B
    double* C = B + (int64_t)1024 * mp + 1024;
    CritStep cs; memset(&cs, 0, sizeof cs);
    cs.panel = panel; cs.ld = mp; cs.inv = inv; cs.C = C; cs.ldc = mp;
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    // 1. event time of [touch] -> panel -> syrk chains of 20, register kernels vs generic
    for (int variant = 0; variant < 3; ++variant) {
        std::vector<float> t;
        for (int rep = 0; rep < 9; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 20; ++i) {
                if (variant < 2) hipLaunchKernelGGL(touch_kernel, dim3(128), dim3(128), 0, s, panel, (int64_t)mp, 128, 1e-6);
                if (variant == 0) {
                    hipLaunchKernelGGL(crit_panel_kernel, dim3(8), dim3(512), 0, s, cs);
                    hipLaunchKernelGGL(crit_syrk_kernel, dim3(9), dim3(256), 0, s, cs);
                } else if (variant == 1) {
                    GemmNT g; memset(&g, 0, sizeof g); g.batch = 1; g.batch2 = 1; g.unit_diag_from = -1;
                    g.P = panel; g.ldp = mp; g.Q = inv; g.ldq = 128; g.C = panel; g.ldc = mp; g.M = 128; g.N = 128; g.K = 128; g.alpha = 1.0;
                    CK((launch_gemm_nt<32, 128, 32, 1, 8>(g, s)));
                    GemmNT u = g; u.Q = panel; u.ldq = mp; u.C = C; u.alpha = -1.0; u.beta = 1.0; u.lower = 1;
                    CK((launch_gemm_nt<32, 32, 32, 2, 2>(u, s)));
                } else {
                    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, tk);
                    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, tk + 1);
                }
            }
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1000.f / 20.f);
        }
        printf("%-44s: %.2f us per chain step (median of 9 x 20)\n",
               variant == 0 ? "touch + crit_panel + crit_syrk (register)" : variant == 1 ? "touch + generic 4-stage panel + update" : "two empty kernels (boundary floor)", med(t));
    }
    // 2. inside crit_syrk: entry -> loads landed -> MFMAs done -> stores drained; and kernel-to-kernel gaps
    hipLaunchKernelGGL(touch_kernel, dim3(128), dim3(128), 0, s, panel, (int64_t)mp, 128, 1e-6);
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, tk);
    hipLaunchKernelGGL(crit_syrk_stamped, dim3(9), dim3(256), 0, s, cs, st);
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, tk + 1);
    CK(hipStreamSynchronize(s));
    long long hs[36], ht[2];
    CK(hipMemcpy(hs, st, sizeof hs, hipMemcpyDeviceToHost)); CK(hipMemcpy(ht, tk, sizeof ht, hipMemcpyDeviceToHost));
    printf("crit_syrk stamps (us, 0 = stamp kernel before it): ");
    for (int b = 0; b < 9; b += 4) printf("[wg%d entry %.2f loads %.2f mfma %.2f stores %.2f] ", b, (hs[4 * b] - ht[0]) * 0.01, (hs[4 * b + 1] - ht[0]) * 0.01,
                                          (hs[4 * b + 2] - ht[0]) * 0.01, (hs[4 * b + 3] - ht[0]) * 0.01);
    printf("| next kernel's stamp at %.2f\n", (ht[1] - ht[0]) * 0.01);
    return 0;
}
