// xcd_hop_probe.hip -- does a dependent kernel see its predecessor's 128 KB sooner when both run on the SAME XCD?
// The three kernels of the Cholesky pivot chain hand 128 KB to each other per step; a stream restricted to one XCD's CUs
// (hipExtStreamCreateWithCUMask) would keep those lines in that XCD's L2 -- if kernel boundaries leave them there.
// Producer: one workgroup rewrites a 128 x 128 block (row stride 4096 doubles).  Consumer: one workgroup loads it with the
// potrf_diag access pattern and stamps s_memrealtime (100 MHz) at entry and after the loads landed.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/xcd_hop_probe tools/xcd_hop_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double f64x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void producer(double* B, long ld, double v) {
    for (int u = 0; u < 16; ++u) {
        int idx = threadIdx.x + u * 512, i = idx >> 6, c2 = (idx & 63) * 2;
        *reinterpret_cast<f64x2*>(B + (long)i * ld + c2) = (f64x2){v + i, v - c2};
    }
}
__global__ __launch_bounds__(512) void consumer(const double* B, long ld, long long* st, double* sink, int slot) {
    long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    f64x2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        int idx = threadIdx.x + u * 512, i = idx >> 6, c2 = (idx & 63) * 2;
        v[u] = *reinterpret_cast<const f64x2*>(B + (long)i * ld + c2);
    }
    double s = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u].x + v[u].y;
    asm volatile("" :: "v"(s));
    long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { st[3 * slot] = t0; st[3 * slot + 1] = t1; st[3 * slot + 2] = (long long)(xcc & 0xf); }
    if (s == 0.12345) *sink = s;
}

int main() {
    const long ld = 4096;
    double *B, *sink; long long* st;
    CK(hipMalloc(&B, ld * 128 * 8)); CK(hipMalloc(&sink, 8)); CK(hipMalloc(&st, 3 * 64 * 8));
    CK(hipMemset(B, 0, ld * 128 * 8));
    hipStream_t plain, masked;
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    std::vector<uint32_t> mask(8, 0);                       // 256 CUs: bit i <-> XCD i % 8 (round-1 probe): XCD 0 = bits 0, 8, 16, ...
    for (int i = 0; i < 256; i += 8) mask[i / 32] |= 1u << (i % 32);
    CK(hipExtStreamCreateWithCUMask(&masked, 8, mask.data()));
    auto run = [&](hipStream_t s, const char* what) {
        std::vector<double> lat; std::vector<int> xcds;
        for (int rep = 0; rep < 40; ++rep) {
            hipLaunchKernelGGL(producer, dim3(1), dim3(512), 0, s, B, ld, (double)rep);
            hipLaunchKernelGGL(consumer, dim3(1), dim3(512), 0, s, B, ld, st, sink, rep);
        }
        CK(hipStreamSynchronize(s));
        std::vector<long long> h(3 * 40);
        CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        for (int r = 5; r < 40; ++r) { lat.push_back((h[3 * r + 1] - h[3 * r]) * 0.01); xcds.push_back((int)h[3 * r + 2]); }
        std::sort(lat.begin(), lat.end());
        printf("%-34s consumer entry -> 128 KB landed: median %.2f us, min %.2f, max %.2f   (consumer XCC ids seen:", what, lat[lat.size() / 2], lat[0], lat.back());
        std::sort(xcds.begin(), xcds.end()); xcds.erase(std::unique(xcds.begin(), xcds.end()), xcds.end());
        for (int x : xcds) printf(" %d", x);
        printf(")\n");
    };
    run(plain, "plain stream");
    run(masked, "stream masked to one XCD (32 CUs)");
    run(plain, "plain stream");
    run(masked, "stream masked to one XCD (32 CUs)");
    return 0;
}
