// ff_reserve_probe.hip -- facts for the fused formation + factorization (csrc/form_factor.h): where do the persistent
// workers land, and where do the workgroups of a small kernel on ANOTHER stream land while they run?
//   hog:   G workgroups x 512 threads, 139,264 B of LDS (one per CU), spinning ~1 ms; records XCC id, CU id, start / end time
//   probe: P workgroups x 256 threads, 35 KB of LDS (the critical tile update), launched on a second stream while the hog
//          runs; records XCC id, CU id and its start time relative to the launch
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ff_reserve_probe tools/ff_reserve_probe.hip && tools/bin/ff_reserve_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Rec { unsigned xcc, hwid; long long t0, t1; };

__global__ __launch_bounds__(512, 2) void hog(Rec* out, long long ticks, int* sink) {
    __shared__ double lds[139264 / 8];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    double a = lds[threadIdx.x];
    while (wall_clock64() - t0 < ticks) { for (int i = 0; i < 64; ++i) a = a * 1.0000001 + 1e-9; }
    if (a == 123.456) *sink = 1;
    if (threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = Rec{xcc & 0xf, hwid, t0, wall_clock64()};
    }
}
template <int LDS>
__global__ void probe(Rec* out, long long ticks, int* sink) {
    __shared__ double lds[LDS / 8];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    double a = lds[threadIdx.x];
    while (wall_clock64() - t0 < ticks) { for (int i = 0; i < 16; ++i) a = a * 1.0000001 + 1e-9; }
    if (a == 123.456) *sink = 1;
    if (threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = Rec{xcc & 0xf, hwid, t0, wall_clock64()};
    }
}

int main() {
    int* sink; CK(hipMalloc(&sink, 4));
    Rec *dh, *dp; CK(hipMalloc(&dh, sizeof(Rec) * 512)); CK(hipMalloc(&dp, sizeof(Rec) * 4096));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    int rate_khz = 0; CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const long long per_us = rate_khz / 1000;
    hipLaunchKernelGGL(hog, dim3(8), dim3(512), 0, sa, dh, 10 * per_us, sink);       // warm up
    hipLaunchKernelGGL(probe<35 * 1024>, dim3(8), dim3(256), 0, sb, dp, 5 * per_us, sink);
    CK(hipDeviceSynchronize());
    for (int G : {240, 248, 256}) {
        for (int mode = 0; mode < 2; ++mode) {
            // mode 0: idle chip at the hog launch; mode 1: a 2048-workgroup filler kernel is still draining when the hog is launched
            CK(hipMemset(dh, 0, sizeof(Rec) * 512)); CK(hipMemset(dp, 0, sizeof(Rec) * 4096));
            if (mode == 1) hipLaunchKernelGGL(probe<35 * 1024>, dim3(2048), dim3(256), 0, sb, dp + 1024, 20 * per_us, sink);
            hipLaunchKernelGGL(hog, dim3(G), dim3(512), 0, sa, dh, 1500 * per_us, sink);
            // three probes in sequence on the other stream: ten 256-thread workgroups (35 KB), four 512-thread (87 KB), one 512-thread (134 KB)
            hipLaunchKernelGGL(probe<35 * 1024>, dim3(10), dim3(256), 0, sb, dp, 5 * per_us, sink);
            hipLaunchKernelGGL(probe<87 * 1024>, dim3(4), dim3(512), 0, sb, dp + 16, 5 * per_us, sink);
            hipLaunchKernelGGL(probe<134 * 1024>, dim3(1), dim3(512), 0, sb, dp + 32, 5 * per_us, sink);
            hipLaunchKernelGGL(probe<35 * 1024>, dim3(64), dim3(256), 0, sb, dp + 64, 5 * per_us, sink);
            CK(hipDeviceSynchronize());
            std::vector<Rec> h(512), p(4096);
            CK(hipMemcpy(h.data(), dh, sizeof(Rec) * 512, hipMemcpyDeviceToHost));
            CK(hipMemcpy(p.data(), dp, sizeof(Rec) * 4096, hipMemcpyDeviceToHost));
            long long tmin = h[0].t0, tmax0 = h[0].t0;
            std::map<unsigned, int> per_xcd;
            std::map<unsigned, int> cus;
            for (int b = 0; b < G; ++b) { tmin = std::min(tmin, h[b].t0); tmax0 = std::max(tmax0, h[b].t0); per_xcd[h[b].xcc]++; cus[(h[b].xcc << 16) | (h[b].hwid & 0xff00)]++; }
            printf("hog %3d (%s): workers per XCD:", G, mode ? "chip draining a filler kernel" : "idle chip");
            for (auto& kv : per_xcd) printf(" %d", kv.second);
            printf(" | distinct CUs %zu | last worker started %.1f us after the first\n", cus.size(), (double)(tmax0 - tmin) / per_us);
            auto show = [&](const char* name, int off, int n) {
                printf("   %-28s start after first worker (us) / XCD:", name);
                for (int b = 0; b < n && b < 16; ++b) printf(" %.0f/%u", (double)(p[off + b].t0 - tmin) / per_us, p[off + b].xcc);
                double worst = 0; for (int b = 0; b < n; ++b) worst = std::max(worst, (double)(p[off + b].t0 - tmin) / per_us);
                printf("  (latest %.0f)\n", worst);
            };
            show("10 x 256 thr, 35 KB", 0, 10); show("4 x 512 thr, 87 KB", 16, 4); show("1 x 512 thr, 134 KB", 32, 1); show("64 x 256 thr, 35 KB", 64, 64);
        }
    }
    return 0;
}
