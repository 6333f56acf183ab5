// clock_probe.hip -- what shader clock does a single-workgroup latency chain (the Cholesky pivot chain) run at, and what
// does the chip do to the next full-chip kernel after 3 ms of such light load?  In-kernel clock = d(s_memtime) /
// d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/clock_probe tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double f64x4 __attribute__((ext_vector_type(4)));

// one workgroup: a dependent fp64 FMA chain (like the pivot chain); stamps (memtime, realtime) every `chunk` iterations
__global__ void chain_kernel(long long* stamps, int nstamps, int chunk, double* sink) {
    double x = 1.0 + threadIdx.x * 1e-9;
    for (int s = 0; s < nstamps; ++s) {
        if (threadIdx.x == 0) { stamps[2 * s] = (long long)__builtin_amdgcn_s_memtime(); stamps[2 * s + 1] = (long long)__builtin_amdgcn_s_memrealtime(); }
        for (int i = 0; i < chunk; ++i) x = __builtin_fma(x, 1.0000001, 1e-12);
    }
    if (x == 0.5) *sink = x;
}

// heater: every CU runs a light loop (a few MFMAs then sleep) so the power manager sees an active chip
__global__ void heater_kernel(const int* stop, int duty, double* sink, long long max_cycles) {
    f64x4 acc = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    while (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        for (int i = 0; i < duty; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        __builtin_amdgcn_s_sleep(32);
        if ((long long)__builtin_amdgcn_s_memtime() - t0 > max_cycles) break;          // exit condition every wave reaches
    }
    if (acc[0] == 0.123) *sink = acc[0];
}

// full-chip fp64 MFMA kernel of ~2 ms (stand-in for the formation): returns its duration through events
__global__ __launch_bounds__(256, 2) void mfma_kernel(int iters, double* sink, long long* stamps = nullptr) {
    f64x4 acc[8];
    for (int q = 0; q < 8; ++q) acc[q] = (f64x4){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i) {
        if (stamps && blockIdx.x == 0 && threadIdx.x == 0 && (i & 127) == 0) {
            stamps[2 * (i >> 7)] = (long long)__builtin_amdgcn_s_memtime(); stamps[2 * (i >> 7) + 1] = (long long)__builtin_amdgcn_s_memrealtime(); }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
    double s = 0; for (int q = 0; q < 8; ++q) s += acc[q][0];
    if (s == 0.123) *sink = s;
}

int main() {
    const int NS = 33;
    long long* d_st; double* sink; int* stop;
    CK(hipMalloc(&d_st, NS * 16)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&stop, 4));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<long long> h(NS * 2);
    auto report = [&](const char* what) {
        CK(hipMemcpy(h.data(), d_st, NS * 16, hipMemcpyDeviceToHost));
        printf("%-46s chain clock (GHz) per ~0.1 ms slice:", what);
        for (int s = 0; s + 4 < NS; s += 4) {
            double dt = (double)(h[2 * (s + 4)] - h[2 * s]), dr = (double)(h[2 * (s + 4) + 1] - h[2 * s + 1]);
            printf(" %.2f", dt / dr * 0.1);
        }
        printf("  | total %.2f ms\n", (double)(h[2 * (NS - 1) + 1] - h[1]) / 100e3);
    };
    const int chunk = 28000;          // ~28000 dependent FMAs per stamp (~0.1 ms)
    // (a) after the GPU sat idle
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, s1, d_st, NS, chunk, sink);
        CK(hipDeviceSynchronize());
        report("chain after idle");
    }
    // (b) right after 10 full-chip MFMA kernels
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(mfma_kernel, dim3(512), dim3(256), 0, s1, 4000, sink);
    hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, s1, d_st, NS, chunk, sink);
    CK(hipDeviceSynchronize());
    report("chain right after 10 full-chip MFMA kernels");
    // (c) chain beside a heater on the other CUs, several duty levels
    for (int duty : {0, 1, 4, 16}) {
        CK(hipMemset(stop, 0, 4));
        hipLaunchKernelGGL(heater_kernel, dim3(255), dim3(256), 0, s2, stop, duty, sink, (long long)2.4e9 / 20);   // <= 50 ms
        hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, s1, d_st, NS, chunk, sink);
        CK(hipStreamSynchronize(s1));
        int one = 1; CK(hipMemcpyAsync(stop, &one, 4, hipMemcpyHostToDevice, s1));
        CK(hipDeviceSynchronize());
        char buf[96]; snprintf(buf, sizeof buf, "chain beside heater (255 WGs, %d MFMA + sleep)", duty);
        report(buf);
    }
    // (d) duration of one full-chip MFMA kernel: back to back, after the chain alone, after the chain beside a heater
    auto time_mfma = [&]() { float ms; CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(mfma_kernel, dim3(512), dim3(256), 0, s1, 4000, sink);
                             CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); return ms; };
    for (int i = 0; i < 5; ++i) time_mfma();
    printf("full-chip MFMA kernel back to back: %.3f %.3f %.3f ms\n", time_mfma(), time_mfma(), time_mfma());
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, s1, d_st, NS, chunk, sink);
        printf("full-chip MFMA kernel after the 3 ms chain: %.3f ms\n", time_mfma());
    }
    for (int duty : {1, 4}) for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(stop, 0, 4));
        hipLaunchKernelGGL(heater_kernel, dim3(255), dim3(256), 0, s2, stop, duty, sink, (long long)2.4e9 / 20);
        hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, s1, d_st, NS, chunk, sink);
        CK(hipStreamSynchronize(s1));
        int one = 1; CK(hipMemcpyAsync(stop, &one, 4, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s2));
        printf("full-chip MFMA kernel after the chain beside heater(%d): %.3f ms\n", duty, time_mfma());
    }
    // (e) clock trajectory INSIDE the full-chip kernel (block 0 stamps every 128 iterations = 1024 MFMAs per wave):
    //     back to back vs after the chain
    long long* d_ms; CK(hipMalloc(&d_ms, 64 * 16)); std::vector<long long> hm(128);
    auto traj = [&](const char* what) {
        CK(hipMemset(d_ms, 0, 64 * 16));
        hipLaunchKernelGGL(mfma_kernel, dim3(512), dim3(256), 0, s1, 4000, sink, d_ms);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hm.data(), d_ms, 64 * 16, hipMemcpyDeviceToHost));
        printf("%-40s clock (GHz) / us per 128-iteration slice:", what);
        for (int s = 0; s + 1 < 32; ++s) {
            double dt = (double)(hm[2 * (s + 1)] - hm[2 * s]), dr = (double)(hm[2 * (s + 1) + 1] - hm[2 * s + 1]);
            if (dr > 0) printf(" %.2f/%.0f", dt / dr * 0.1, dr / 100.0);
        }
        printf("\n");
    };
    for (int i = 0; i < 5; ++i) time_mfma();
    traj("inside MFMA kernel, back to back");
    hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, s1, d_st, NS, 6000, sink);
    traj("inside MFMA kernel, after ~3 ms chain");
    CK(hipDeviceSynchronize());
    { struct timespec ts = {0, 50000000}; nanosleep(&ts, nullptr); }
    traj("inside MFMA kernel, after 50 ms idle");
    // two in a row after the chain: does the second one recover?
    hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, s1, d_st, NS, 6000, sink);
    { float a = time_mfma(), b = time_mfma(), c = time_mfma(); printf("after chain: 1st %.3f 2nd %.3f 3rd %.3f ms\n", a, b, c); }
    // heavier heaters during the chain
    for (int duty : {16, 64}) {
        CK(hipMemset(stop, 0, 4));
        hipLaunchKernelGGL(heater_kernel, dim3(255), dim3(256), 0, s2, stop, duty, sink, (long long)2.4e9 / 20);
        hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, s1, d_st, NS, 6000, sink);
        CK(hipStreamSynchronize(s1));
        int one = 1; CK(hipMemcpyAsync(stop, &one, 4, hipMemcpyHostToDevice, s1));
        CK(hipStreamSynchronize(s2));
        printf("full-chip MFMA kernel after the chain beside heater(%d): %.3f ms\n", duty, time_mfma());
    }
    return 0;
}
