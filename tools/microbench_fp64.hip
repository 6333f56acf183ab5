// microbench_fp64.hip -- what the fp64 pipes of one MI355X actually deliver (roofline calibration).
//   hipcc --offload-arch=gfx950 -O3 -o microbench_fp64 tools/microbench_fp64.hip && ./microbench_fp64
// Variants: v_mfma_f64_16x16x4_f64 back to back (1/2/4 waves per SIMD), v_mfma_f64_4x4x4_4b_f64,
// v_fma_f64 (VALU), and MFMA waves co-resident with VALU-FMA waves on the same SIMDs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void mfma16_kernel(double* out, int iters, double seed) {
    f64x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f64x4){seed, seed, seed, seed};
    double a = seed + threadIdx.x * 1e-9, b = seed - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma4_kernel(double* out, int iters, double seed) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = seed;
    double a = seed + threadIdx.x * 1e-9, b = seed - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s == 12345.678) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void fma_kernel(double* out, int iters, double seed) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = seed + i;
    double a = 1.0 + threadIdx.x * 1e-12, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s == 12345.678) out[0] = s;
}

// waves 0..3 of a 512-thread block run MFMA, waves 4..7 run VALU FMA (one of each per SIMD)
__global__ __launch_bounds__(512) void mixed_kernel(double* out, int iters, double seed, int fma_per_mfma) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave < 4) {
        f64x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (f64x4){seed, seed, seed, seed};
        double a = seed + threadIdx.x * 1e-9, b = seed - threadIdx.x * 1e-9;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = seed + i;
        double a = 1.0 + threadIdx.x * 1e-12, b = 1e-9;
        for (int it = 0; it < iters * fma_per_mfma / 4; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
        }
        for (int i = 0; i < 16; ++i) s += acc[i];
    }
    if (s == 12345.678) out[0] = s;
}

// same wave interleaves MFMA and independent VALU FMAs
template <int NF>
__global__ __launch_bounds__(256) void interleaved_kernel(double* out, int iters, double seed) {
    f64x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f64x4){seed, seed, seed, seed};
    double f[NF > 0 ? NF : 1];
    for (int i = 0; i < NF; ++i) f[i] = seed + i;
    double a = seed + threadIdx.x * 1e-9, b = seed - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NF / 4; ++j) f[i * (NF / 4) + j] = __builtin_fma(f[i * (NF / 4) + j], a, b);
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < NF; ++i) s += f[i];
    if (s == 12345.678) out[0] = s;
}

template <typename F>
double time_ms(F launch, int reps = 5) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    printf("device %s  CUs %d  clock %d kHz\n", p.name, cus, p.clockRate);
    double* out; CHECK(hipMalloc(&out, 64));
    const int iters = 20000;
    for (int bpc = 1; bpc <= 4; bpc *= 2) {       // blocks of 256 threads per CU = waves per SIMD
        int grid = cus * bpc;
        double ms = time_ms([&] { hipLaunchKernelGGL(mfma16_kernel<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0); });
        double fl = (double)grid * 4 * iters * 4 * 2048.0;
        printf("mfma_f64_16x16x4  4 acc, %d wave/SIMD: %.3f ms  %.2f TFLOP/s  (%.1f cyc/mfma/SIMD @2.4GHz)\n", bpc, ms, fl / ms / 1e9,
               ms * 1e-3 * 2.4e9 / ((double)iters * 4 * bpc));
    }
    {
        int grid = cus;
        double ms = time_ms([&] { hipLaunchKernelGGL(mfma16_kernel<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0); });
        printf("mfma_f64_16x16x4  1 acc (dependent chain), 1 wave/SIMD: %.3f ms  %.2f TFLOP/s (%.1f cyc/mfma)\n", ms,
               (double)grid * 4 * iters * 2048.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / iters);
        ms = time_ms([&] { hipLaunchKernelGGL(mfma16_kernel<16>, dim3(grid), dim3(256), 0, 0, out, iters / 4, 1.0); });
        printf("mfma_f64_16x16x4 16 acc, 1 wave/SIMD: %.3f ms  %.2f TFLOP/s\n", ms, (double)grid * 4 * (iters / 4) * 16 * 2048.0 / ms / 1e9);
    }
    for (int bpc = 1; bpc <= 2; bpc *= 2) {
        int grid = cus * bpc;
        double ms = time_ms([&] { hipLaunchKernelGGL(mfma4_kernel<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0); });
        double fl = (double)grid * 4 * iters * 8 * 512.0;   // 4 blocks of 4x4x4: 4*4*4*4*2 = 512 flop
        printf("mfma_f64_4x4x4(4b) 8 acc, %d wave/SIMD: %.3f ms  %.2f TFLOP/s\n", bpc, ms, fl / ms / 1e9);
    }
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        int grid = cus * bpc;
        double ms = time_ms([&] { hipLaunchKernelGGL(fma_kernel<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0); });
        double fl = (double)grid * 256 * iters * 16 * 2.0;
        printf("v_fma_f64 16 acc, %d wave/SIMD: %.3f ms  %.2f TFLOP/s\n", bpc, ms, fl / ms / 1e9);
    }
    for (int fpm = 0; fpm <= 16; fpm = fpm ? fpm * 2 : 4) {
        int grid = cus;
        double ms = time_ms([&] { hipLaunchKernelGGL(mixed_kernel, dim3(grid), dim3(512), 0, 0, out, iters, 1.0, fpm); });
        double fm = (double)grid * 4 * iters * 4 * 2048.0;
        double ff = (double)grid * 256 * (iters * fpm / 4) * 16 * 2.0;
        printf("mixed (MFMA wave + FMA wave per SIMD), %2d fma/mfma: %.3f ms  mfma %.2f + valu %.2f = %.2f TFLOP/s\n", fpm, ms,
               fm / ms / 1e9, ff / ms / 1e9, (fm + ff) / ms / 1e9);
    }
    {
        int grid = cus;
        double ms0 = time_ms([&] { hipLaunchKernelGGL(interleaved_kernel<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0); });
        double ms4 = time_ms([&] { hipLaunchKernelGGL(interleaved_kernel<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0); });
        double ms8 = time_ms([&] { hipLaunchKernelGGL(interleaved_kernel<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0); });
        double ms16 = time_ms([&] { hipLaunchKernelGGL(interleaved_kernel<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0); });
        double ms32 = time_ms([&] { hipLaunchKernelGGL(interleaved_kernel<32>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0); });
        double fm = (double)grid * 4 * iters * 4 * 2048.0;
        printf("interleaved in one wave (1 wave/SIMD): 0/1/2/4/8 fma per mfma: %.3f %.3f %.3f %.3f %.3f ms ; mfma-only %.2f TF; with 8: mfma %.2f + valu %.2f TF\n",
               ms0, ms4, ms8, ms16, ms32, fm / ms0 / 1e9, fm / ms32 / 1e9, (double)grid * 256 * iters * 32 * 2.0 / ms32 / 1e9);
    }
    CHECK(hipFree(out));
    return 0;
}
