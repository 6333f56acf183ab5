// Does hipExtAnyOrderLaunch let a kernel start before its predecessor IN THE SAME STREAM has finished on gfx950?
// (hip_ext.h says the flag is not supported on GFX9xx.)  A: one workgroup spinning ~200 us; B: tiny, launched with / without the flag;
// both write wall_clock64 at start and end.   hipcc --offload-arch=gfx950 -O2 -o tools/bin/anyorder_probe tools/anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
__global__ void spin_kernel(long long* t, long long ticks) {
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        t[0] = t0;
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
        t[1] = wall_clock64();
    }
}
__global__ void mark_kernel(long long* t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[2] = wall_clock64(); t[3] = wall_clock64(); }
}
int main() {
    long long* d; hipMalloc(&d, 64); hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int flag = 0; flag < 2; ++flag)
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(d, 0, 64, s);
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, d, 20000LL);     // 100 MHz clock: 200 us
            hipExtLaunchKernelGGL(mark_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, d);
            hipStreamSynchronize(s);
            long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
            printf("flag %d: A %.1f us long; B starts %.1f us after A starts (%s A ends)\n", flag, (h[1] - h[0]) / 100.0, (h[2] - h[0]) / 100.0,
                   h[2] < h[1] ? "BEFORE" : "after");
        }
    return 0;
}
