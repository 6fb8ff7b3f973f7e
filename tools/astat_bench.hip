// Diagnostic: the register-resident-activation bf16 projection kernel alone, on random data, timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ispeech_separation_amd/csrc -Wno-inline-asm -o tools/astat_bench tools/astat_bench.hip [-DMSF2_DIAG_...]
//   tools/astat_bench [batch [N]]
#include "msf2_common.h"
#include "msf2_gemm.h"
#include "msf2_gemm_bf16.h"
#include "msf2_gemm_bf16_astat.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace msf2 { thread_local int g_last_hip_error = 0; thread_local bool g_bad_operand = false; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, N = argc > 2 ? atoi(argv[2]) : 2176, S = 3999, K = 512;
    const long M = (long)B * S;
    unsigned short *A, *W, *O; float *bias, *rs, *zeros;
    CK(hipMalloc(&A, M * K * 2)); CK(hipMalloc(&W, (long)N * K * 2)); CK(hipMalloc(&O, M * N * 2));
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&rs, M * 4)); CK(hipMalloc(&zeros, 4096));
    std::vector<unsigned short> h(M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0x3ff);      // bf16 values around 0.01
    CK(hipMemcpy(A, h.data(), M * K * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (long)N * K * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(zeros, 0, 4096));
    std::vector<float> r(M, 1.f);
    CK(hipMemcpy(rs, r.data(), M * 4, hipMemcpyHostToDevice));
    msf2::ProjArgs a = {};
    a.A16 = A; a.lda = K; a.W = reinterpret_cast<const float*>(W); a.K = K; a.N = N; a.S = S; a.out16 = O; a.ldo = N;
    a.bias = bias; a.rowstat = rs; a.zeros = zeros; a.act = 1;
    const dim3 g((S + 127) / 128, B);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((msf2::proj_astat_bf16_kernel<1, 1>), g, dim3(256), 0, 0, a);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((msf2::proj_astat_bf16_kernel<1, 1>), g, dim3(256), 0, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double flop = 2.0 * M * N * K;
    printf("batch %d, N %d: %.1f us, %.0f TFLOP/s (%.1f %% of 2500), %d workgroups\n", B, N, 1e3 * ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0, g.x * g.y);
    return 0;
}
