// prio_probe.hip -- can a latency chain share the GPU with a saturating kernel?
//   A. what does a CU-masked stream (hipExtStreamCreateWithCUMask) actually exclude?
//   B. does a 133 KB-LDS single-workgroup kernel (potrf_diag-like) starve behind a saturating 2-per-CU
//      kernel, with and without reserved (masked-out) CUs?
//   C. do stream priorities shorten the wait of small launches behind a saturating kernel?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void busy_kernel(unsigned* out, int iters, double seed, double* sink) {
    extern __shared__ char pad[];
    if (seed == 42.0) pad[threadIdx.x] = 1;
    f64x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f64x4){seed, seed, seed, seed};
    double a = seed + threadIdx.x * 1e-9, b = seed - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s + pad[0];
    if (out && threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2] = hwid;
        out[blockIdx.x * 2 + 1] = xcc;
    }
}

static double* g_sink;
static const int LDS_FORM = 73 * 1024, LDS_DIAG = 133 * 1024;

static void census(const char* name, hipStream_t st, int grid) {
    unsigned* d; CHECK(hipMalloc(&d, sizeof(unsigned) * 2 * grid));
    hipLaunchKernelGGL(busy_kernel, dim3(grid), dim3(256), LDS_FORM, st, d, 200, 1.0, g_sink);
    CHECK(hipStreamSynchronize(st));
    std::vector<unsigned> h(2 * grid);
    CHECK(hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * grid, hipMemcpyDeviceToHost));
    std::map<unsigned, int> cnt; std::map<unsigned, int> perx;
    for (int b = 0; b < grid; ++b) {
        unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        if (!cnt.count((xcc << 16) | (se << 8) | (sh << 4) | cu)) perx[xcc]++;
        cnt[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
    }
    printf("[A] %-28s distinct CUs %zu; per XCD:", name, cnt.size());
    for (auto& kv : perx) printf(" %d", kv.second);
    printf("\n");
    CHECK(hipFree(d));
}

// background: nb launches of bg_grid workgroups (LDS_FORM, 256 thr, bg_iters); foreground: nf sequential launches of
// fg_grid workgroups (fg_lds, fg_thr, fg_iters).  Returns foreground total ms and background total ms.
static void contend(const char* name, hipStream_t bg, hipStream_t fg, int nb, int bg_grid, int bg_iters, int nf, int fg_grid,
                    int fg_thr, int fg_lds, int fg_iters) {
    hipEvent_t b0, b1, f0, f1, go;
    CHECK(hipEventCreate(&b0)); CHECK(hipEventCreate(&b1)); CHECK(hipEventCreate(&f0)); CHECK(hipEventCreate(&f1)); CHECK(hipEventCreate(&go));
    CHECK(hipDeviceSynchronize());
    // foreground alone
    CHECK(hipEventRecord(f0, fg));
    for (int i = 0; i < nf; ++i) hipLaunchKernelGGL(busy_kernel, dim3(fg_grid), dim3(fg_thr), fg_lds, fg, (unsigned*)nullptr, fg_iters, 1.0, g_sink);
    CHECK(hipEventRecord(f1, fg)); CHECK(hipEventSynchronize(f1));
    float alone; CHECK(hipEventElapsedTime(&alone, f0, f1));
    // background alone
    CHECK(hipEventRecord(b0, bg));
    for (int i = 0; i < nb; ++i) hipLaunchKernelGGL(busy_kernel, dim3(bg_grid), dim3(256), LDS_FORM, bg, (unsigned*)nullptr, bg_iters, 1.0, g_sink);
    CHECK(hipEventRecord(b1, bg)); CHECK(hipEventSynchronize(b1));
    float bgalone; CHECK(hipEventElapsedTime(&bgalone, b0, b1));
    // together: background first, foreground starts when the first background launch has begun (event after a tiny kernel)
    CHECK(hipEventRecord(b0, bg));
    for (int i = 0; i < nb; ++i) hipLaunchKernelGGL(busy_kernel, dim3(bg_grid), dim3(256), LDS_FORM, bg, (unsigned*)nullptr, bg_iters, 1.0, g_sink);
    CHECK(hipEventRecord(b1, bg));
    CHECK(hipEventRecord(f0, fg));
    for (int i = 0; i < nf; ++i) hipLaunchKernelGGL(busy_kernel, dim3(fg_grid), dim3(fg_thr), fg_lds, fg, (unsigned*)nullptr, fg_iters, 1.0, g_sink);
    CHECK(hipEventRecord(f1, fg));
    CHECK(hipEventSynchronize(f1)); CHECK(hipEventSynchronize(b1));
    float both_f, both_b; CHECK(hipEventElapsedTime(&both_f, f0, f1)); CHECK(hipEventElapsedTime(&both_b, b0, b1));
    printf("%-64s fg alone %.3f ms, with bg %.3f ms | bg alone %.3f ms, with fg %.3f ms\n", name, alone, both_f, bgalone, both_b);
}

int main() {
    CHECK(hipMalloc(&g_sink, 64));
    CHECK(hipFuncSetAttribute((const void*)busy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int lo, hi; CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("stream priority range: least %d greatest %d\n", lo, hi);
    hipStream_t plain, s_hi, s_lo, m224, m32, m224lo, mEven;
    CHECK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, hi));
    CHECK(hipStreamCreateWithPriority(&s_lo, hipStreamNonBlocking, lo));
    uint32_t mask[8];
    for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu; mask[7] = 0;                 // bits 0..223
    hipError_t e = hipExtStreamCreateWithCUMask(&m224, 8, mask);
    printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 0;
    for (int i = 0; i < 8; ++i) mask[i] = 0; mask[0] = 0xffffffffu;                 // bits 0..31
    CHECK(hipExtStreamCreateWithCUMask(&m32, 8, mask));
    for (int i = 0; i < 8; ++i) mask[i] = 0x55555555u;                              // even bits
    CHECK(hipExtStreamCreateWithCUMask(&mEven, 8, mask));
    for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu; mask[7] = 0;
    CHECK(hipExtStreamCreateWithCUMask(&m224lo, 8, mask));

    census("unmasked", plain, 2048);
    census("mask bits 0..223", m224, 2048);
    census("mask bits 0..31", m32, 2048);
    census("mask even bits", mEven, 2048);

    // B: potrf_diag-like foreground (1 wg, 512 thr, 133 KB, ~20 us) x 20 behind 8 x 448-wg background launches (~250 us each)
    contend("[B] diag-like fg (plain) vs bg on plain stream, 8x512 wg", plain, s_hi, 8, 512, 1100, 20, 1, 512, LDS_DIAG, 180);
    contend("[B] diag-like fg (plain) vs bg on 224-CU masked stream, 8x448 wg", m224, plain, 8, 448, 1100, 20, 1, 512, LDS_DIAG, 180);
    contend("[B] diag-like fg vs masked bg, one long launch 3584 wg", m224, plain, 1, 3584, 1100, 20, 1, 512, LDS_DIAG, 180);
    // C: small 73 KB launches (16 wg, ~10 us) x 30 behind a long many-workgroup background (4096 wg x ~55 us)
    contend("[C] small fg on plain vs bg plain (4096 wg x 55 us)", plain, s_hi == plain ? plain : plain, 1, 4096, 250, 30, 16, 256, LDS_FORM, 90);
    hipStream_t plain2; CHECK(hipStreamCreateWithFlags(&plain2, hipStreamNonBlocking));
    contend("[C] small fg on plain2 vs bg plain", plain, plain2, 1, 4096, 250, 30, 16, 256, LDS_FORM, 90);
    contend("[C] small fg HIGH prio vs bg plain", plain, s_hi, 1, 4096, 250, 30, 16, 256, LDS_FORM, 90);
    contend("[C] small fg HIGH prio vs bg LOW prio", s_lo, s_hi, 1, 4096, 250, 30, 16, 256, LDS_FORM, 90);
    contend("[C] small fg plain vs bg masked 224 (4096 wg)", m224, plain2, 1, 4096, 250, 30, 16, 256, LDS_FORM, 90);
    // D: bulk-update-like foreground (400 wg, 73 KB, ~8 us) x 10 vs masked background
    contend("[D] 400-wg fg plain vs bg masked 224 (4096 wg x 55 us)", m224, plain2, 1, 4096, 250, 10, 400, 256, LDS_FORM, 40);
    contend("[D] 400-wg fg HIGH vs bg LOW (4096 wg x 55 us)", s_lo, s_hi, 1, 4096, 250, 10, 400, 256, LDS_FORM, 40);
    contend("[D] 400-wg fg HIGH vs bg LOW (1024 wg x 220 us)", s_lo, s_hi, 1, 1024, 1000, 10, 400, 256, LDS_FORM, 40);
    return 0;
}
