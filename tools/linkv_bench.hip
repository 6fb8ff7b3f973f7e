// Diagnostic: the bf16 key/value-summary kernel alone, on random data, timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ispeech_separation_amd/csrc -Iinclude -Wno-inline-asm -o tools/linkv_bench tools/linkv_bench.hip [-DMSF2_DIAG_LKV_...]
//   tools/linkv_bench [batch]
#include "msf2_common.h"
#include "msf2_gemm.h"
#include "msf2_attn_fused.h"
#include "msf2_linkv_bf16.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace msf2 { thread_local int g_last_hip_error = 0; thread_local bool g_bad_operand = false; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, S = 3999, N = 2048, LD = 2176;
    const long M = (long)B * S;
    unsigned short *lk, *hu; float* part;
    CK(hipMalloc(&lk, M * 128 * 2)); CK(hipMalloc(&hu, M * LD * 2)); CK(hipMalloc(&part, (long)B * 16 * 128 * N * 4));
    std::vector<unsigned short> h(M * 128);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0x3ff);
    CK(hipMemcpy(lk, h.data(), M * 128 * 2, hipMemcpyHostToDevice));
    for (int j = 0; j < LD / 128; ++j) CK(hipMemcpy(hu + (long)j * M * 128, h.data(), M * 128 * 2, hipMemcpyHostToDevice));
    const int nsplit = B >= 64 ? 1 : (64 / B > 15 ? 15 : 64 / B);
    const int chunk = ((S + nsplit - 1) / nsplit + 31) / 32 * 32;
    const int ns = (S + chunk - 1) / chunk;
    msf2::LinKV16Args a = {lk, hu, LD, part, S, N, chunk};
    const dim3 g(N / 256, ns, B);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(msf2::linkv_bf16_kernel, g, dim3(512), 0, 0, a);
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(msf2::linkv_bf16_kernel, g, dim3(512), 0, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = (double)M * (N + 128) * 2;
    printf("batch %d, %d chunks of %d: %.1f us, %.2f TB/s of unique operand bytes, %d workgroups\n", B, ns, chunk, 1e3 * ms, bytes / ms / 1e9, g.x * g.y * g.z);
    return 0;
}
