// reserve_probe.hip -- can a latency chain keep whole CUs for itself while a saturating kernel runs, WITHOUT a CU mask?
//   hog:   G workgroups of 256 threads, each holding `hog_lds` bytes of LDS, spinning for ~2 ms   (stream A)
//   chain: 40 dependent launches of ONE workgroup holding 133 KB of LDS (what potrf_diag needs), ~20 us each   (stream B)
// If the dispatcher places one hog workgroup per CU (hog_lds > 80 KB: two do not fit) and G < 256, the remaining CUs stay
// empty and the chain should run at its solo pace; with G >= 256 (or two 64 KB hogs per CU) it should starve.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/reserve_probe tools/reserve_probe.hip && tools/bin/reserve_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void hog(long long cycles, int* sink) {
    extern __shared__ double lds[];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    double a = lds[threadIdx.x];
    while (wall_clock64() - t0 < cycles) { for (int i = 0; i < 64; ++i) a = a * 1.0000001 + 1e-9; }
    if (a == 123.456) *sink = 1;
}
__global__ void link(long long cycles, int* sink) {
    extern __shared__ double lds[];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    double a = lds[threadIdx.x];
    while (wall_clock64() - t0 < cycles) { for (int i = 0; i < 16; ++i) a = a * 1.0000001 + 1e-9; }
    if (a == 123.456) *sink = 1;
}

int main() {
    int* sink; CK(hipMalloc(&sink, 4));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)link, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int rate_khz = 0; CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const long long per_us = rate_khz / 1000;                     // wall_clock64 ticks per microsecond
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto chain_ms = [&](int nlinks) -> float {
        hipEventRecord(e0, sb);
        for (int i = 0; i < nlinks; ++i) hipLaunchKernelGGL(link, dim3(1), dim3(512), 133 * 1024, sb, 20 * per_us, sink);
        hipEventRecord(e1, sb); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); return ms;
    };
    chain_ms(10);
    printf("chain alone: 40 links of 20 us: %.3f ms\n", chain_ms(40));
    struct Cfg { int G, lds_kb; };
    for (Cfg c : {Cfg{512, 64}, Cfg{256, 96}, Cfg{248, 96}, Cfg{240, 96}, Cfg{224, 96}, Cfg{448, 64}, Cfg{480, 64}}) {
        hipLaunchKernelGGL(hog, dim3(c.G), dim3(256), c.lds_kb * 1024, sa, 2000 * per_us, sink);
        const float ms = chain_ms(40);
        CK(hipStreamSynchronize(sa));
        printf("hog %3d workgroups x %2d KB LDS for 2 ms: chain of 40 links took %.3f ms\n", c.G, c.lds_kb, ms);
    }
    return 0;
}
