// ff_gemm_bench.hip -- throughput of the GEMM core of the fused formation + factorization (csrc/form_factor.h) in isolation:
// W workgroups (one per CU, 512 threads, 136 KB of LDS), each multiplying its own 128-row panels of a random A with the d
// scaling for `ns` BK = 32 stages, `reps` times.  Variant 0: the plain double-buffered loop (ff_gemm), variant 1: the
// software-pipelined schedule (ff_gemm_pipe).  Results of the two are compared bit for bit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/ff_gemm_bench tools/ff_gemm_bench.hip && tools/bin/ff_gemm_bench [W ns reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../interiorpointmethod_amd/csrc/form_factor.h"
using namespace ipm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(FF_THREADS, 2) void bench_kernel(const double* A, int64_t lda, const double* d, double* out, int ns, int reps, int nblk) {
    __shared__ __attribute__((aligned(16))) double lds[FF_LDS_DOUBLES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fk = lane >> 4;
    f64x4 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int r = 0; r < reps; ++r) {
        const int ti = (blockIdx.x + 7 * r) % nblk, tc = (blockIdx.x * 3 + r) % nblk;
        const double* P = A + (int64_t)ti * 128 * lda;
        const double* Q = A + (int64_t)tc * 128 * lda;
        if (V == 0) ff_gemm<true, false>(P, lda, Q, lda, d, ns, lds, acc, nullptr);
        else ff_gemm_pipe<true>(P, lda, Q, lda, d, ns, lds, acc);
    }
    double* o = out + (size_t)blockIdx.x * 128 * 128 + (wm * 64 + fk) * 128 + wn * 32 + fr;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 4; ++q) o[(i * 16 + 4 * q) * 128 + j * 16] = acc[i][j][q];
}

// spread: 0 = every workgroup multiplies the SAME K range (the first 16 ns16 columns: 67 MB of A at ns16 = 128, Infinity-Cache resident);
//         1 = workgroup b starts at column chunk (5 b + r) mod 4 (all of A in use at any time, as in the fused launch's pair-major list)
__global__ __launch_bounds__(FF_THREADS, 2) void bench_pair_kernel(const double* A, int64_t lda, const double* d, double* out, int ns16, int reps, int nblk, int spread) {
    __shared__ __attribute__((aligned(16))) double lds[FF_LDS_DOUBLES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fk = lane >> 4;
    f64x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
    for (int r = 0; r < reps; ++r) {
        const int ti = 2 * ((blockIdx.x + 7 * r) % (nblk / 2)), tc = (blockIdx.x * 3 + r) % nblk;
        const int64_t ko = spread ? (int64_t)((5 * blockIdx.x + r) % 4) * 16 * ns16 : 0;
        ff_gemm_pair(A + (int64_t)ti * 128 * lda + ko, A + (int64_t)(ti + 1) * 128 * lda + ko, A + (int64_t)tc * 128 * lda + ko, lda, d + ko, ns16, lds, acc);
    }
    double* o = out + (size_t)blockIdx.x * 256 * 128 + (wm * 64 + fk) * 128 + wn * 64 + fr;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) o[(i * 16 + 4 * q) * 128 + j * 16] = acc[i][j][q];
}

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 248, ns = argc > 2 ? atoi(argv[2]) : 64, reps = argc > 3 ? atoi(argv[3]) : 8;
    const int m = 4096, n = 8192, nblk = m / 128;
    std::vector<double> hA((size_t)m * n), hd(n);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (auto& v : hA) v = 2.0 * rnd() - 1.0;
    for (auto& v : hd) v = 0.5 + rnd();
    double *A, *d, *o0, *o1;
    CK(hipMalloc(&A, hA.size() * 8)); CK(hipMalloc(&d, n * 8)); CK(hipMalloc(&o0, (size_t)W * 128 * 128 * 8)); CK(hipMalloc(&o1, (size_t)W * 128 * 128 * 8));
    CK(hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d, hd.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = (double)W * reps * ns * 128.0 * 128.0 * 32.0 * 2.0;
    for (int round = 0; round < 4; ++round)
        for (int v = 0; v < 2; ++v) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(bench_kernel<0>, dim3(W), dim3(FF_THREADS), 0, 0, A, (int64_t)n, d, o0, ns, reps, nblk);
            else hipLaunchKernelGGL(bench_kernel<1>, dim3(W), dim3(FF_THREADS), 0, 0, A, (int64_t)n, d, o1, ns, reps, nblk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (round) printf("variant %d: %.3f ms  %.1f TFLOP/s (%.3f of 78.6 on all 256 CUs; per-stage %.2f us)\n", v, ms, flops / ms / 1e9, flops / ms / 1e9 / 78.6,
                              ms * 1e3 / (reps * ns));
        }
    {   // the pair engine: 256 x 128 per workgroup, BK = 16 stages (2 ns of them for the same K)
        double* o2; CK(hipMalloc(&o2, (size_t)W * 256 * 128 * 8));
        const double fl2 = (double)W * reps * (2 * ns) * 256.0 * 128.0 * 16.0 * 2.0;
        for (int round = 0; round < 8; ++round) {
            const int spread = round >= 4;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(bench_pair_kernel, dim3(W), dim3(FF_THREADS), 0, 0, A, (int64_t)n, d, o2, 2 * ns, reps, nblk, spread);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (round & 3) printf("pair engine (K ranges %s): %.3f ms  %.1f TFLOP/s (%.3f of 78.6 on all 256 CUs; per BK=16 stage %.2f us)\n", spread ? "spread over A" : "shared", ms, fl2 / ms / 1e9, fl2 / ms / 1e9 / 78.6,
                              ms * 1e3 / (reps * 2 * ns));
        }
    }
    std::vector<double> h0((size_t)W * 128 * 128), h1(h0.size());
    CK(hipMemcpy(h0.data(), o0, h0.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), o1, h1.size() * 8, hipMemcpyDeviceToHost));
    size_t diff = 0; for (size_t i = 0; i < h0.size(); ++i) diff += (h0[i] != h1[i]);
    printf("bitwise differences between the variants: %zu of %zu\n", diff, h0.size());
    return 0;
}
