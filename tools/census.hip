// census.hip -- where does the dispatcher put workgroups? (HW_REG_HW_ID / XCC_ID per block)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int LDSBYTES>
__global__ void census_kernel(unsigned* out, int iters, double seed, double* sink) {
    __shared__ char pad[LDSBYTES > 0 ? LDSBYTES : 1];
    if (LDSBYTES > 0 && seed == 42.0) pad[threadIdx.x] = 1;
    f64x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f64x4){seed, seed, seed, seed};
    double a = seed + threadIdx.x * 1e-9, b = seed - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s + (LDSBYTES > 0 ? pad[0] : 0);
    if (threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2] = hwid;
        out[blockIdx.x * 2 + 1] = xcc;
    }
}

template <int LDS>
void run(const char* name, int grid, int threads, int iters) {
    unsigned* d; double* sink;
    CHECK(hipMalloc(&d, sizeof(unsigned) * 2 * grid)); CHECK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(census_kernel<LDS>, dim3(grid), dim3(threads), 0, 0, d, iters, 1.0, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(census_kernel<LDS>, dim3(grid), dim3(threads), 0, 0, d, iters, 1.0, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned> h(2 * grid);
    CHECK(hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * grid, hipMemcpyDeviceToHost));
    std::map<unsigned, int> cnt;
    for (int b = 0; b < grid; ++b) {
        unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        cnt[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
    }
    int mx = 0; std::map<int, int> hist;
    for (auto& kv : cnt) { if (kv.second > mx) mx = kv.second; hist[kv.second]++; }
    double fl = (double)grid * (threads / 64) * iters * 4 * 2048.0;
    printf("%-34s grid %4d x %3d thr: %.3f ms %.1f TFLOP/s | distinct CUs %zu, max blocks/CU %d, hist:", name, grid, threads, ms, fl / ms / 1e9, cnt.size(), mx);
    for (auto& kv : hist) printf(" %dx%d", kv.second, kv.first);
    printf("\n");
    if (grid <= 16) { for (int b = 0; b < grid; ++b) printf("   b%d hw=%08x xcc=%u\n", b, h[2*b], h[2*b+1]); }
    CHECK(hipFree(d)); CHECK(hipFree(sink));
}

int main() {
    const int iters = 20000;
    run<0>("256thr lds0", 256, 256, iters);
    run<0>("256thr lds0", 512, 256, iters);
    run<0>("256thr lds0", 528, 256, iters);
    run<0>("256thr lds0", 1024, 256, iters);
    run<0>("512thr lds0", 256, 512, iters);
    run<0>("512thr lds0", 512, 512, iters);
    run<40 * 1024>("256thr lds40K(<=4/CU)", 512, 256, iters);
    run<64 * 1024>("256thr lds64K(<=2/CU)", 256, 256, iters);
    run<64 * 1024>("256thr lds64K(<=2/CU)", 512, 256, iters);
    run<64 * 1024>("256thr lds64K(<=2/CU)", 528, 256, iters);
    run<64 * 1024>("512thr lds64K(<=2/CU)", 256, 512, iters);
    run<64 * 1024>("512thr lds64K(<=2/CU)", 512, 512, iters);
    return 0;
}
