// Diagnostic: the register-resident two-product TCN contraction alone against a host evaluation of sampled outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ispeech_separation_amd/csrc -Iinclude -Wno-inline-asm -o tools/tcn_astat_check tools/tcn_astat_check.hip
#include "msf2_common.h"
#include "msf2_gemm.h"
#include "msf2_gemm_bf16.h"
#include "msf2_gemm_bf16_astat.h"
#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace msf2 { thread_local int g_last_hip_error = 0; thread_local bool g_bad_operand = false; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
static unsigned short f2bf(float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4, S = argc > 2 ? atoi(argv[2]) : 3999, C = 512;
    const long M = (long)B * S;
    std::vector<unsigned short> h(M * C), t(M * C), wg((long)C * C), wo((long)C * C);
    std::vector<float> bias(C), c1f((long)B * C), bstat(2 * B);
    srand(1);
    auto rnd = [] { return (rand() % 2001 - 1000) / 1000.f; };
    for (auto& v : h) v = f2bf(rnd());
    for (auto& v : t) v = f2bf(rnd());
    for (auto& v : wg) v = f2bf(rnd() * 0.05f);
    for (auto& v : wo) v = f2bf(rnd() * 0.05f);
    for (auto& v : bias) v = rnd() * 0.1f;
    for (auto& v : c1f) v = rnd() * 0.1f;
    for (int b = 0; b < B; ++b) { bstat[2 * b] = 0.1f * b; bstat[2 * b + 1] = 0.5f + 0.1f * b; }
    unsigned short *dh, *dt, *dwg, *dwo, *dout; float *dbias, *dc1f, *dbstat;
    CK(hipMalloc(&dh, M * C * 2)); CK(hipMalloc(&dt, M * C * 2)); CK(hipMalloc(&dout, M * C * 2));
    CK(hipMalloc(&dwg, (long)C * C * 2)); CK(hipMalloc(&dwo, (long)C * C * 2));
    CK(hipMalloc(&dbias, C * 4)); CK(hipMalloc(&dc1f, (long)B * C * 4)); CK(hipMalloc(&dbstat, 2 * B * 4));
    CK(hipMemcpy(dh, h.data(), M * C * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dt, t.data(), M * C * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwg, wg.data(), (long)C * C * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dwo, wo.data(), (long)C * C * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, bias.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc1f, c1f.data(), (long)B * C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbstat, bstat.data(), 2 * B * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0xff, M * C * 2));
    msf2::ProjArgs a = {};
    a.A16 = dh; a.A2_16 = dt; a.lda = C; a.W = reinterpret_cast<const float*>(dwg); a.W2 = reinterpret_cast<const float*>(dwo); a.K = C; a.N = C; a.S = S;
    a.out16 = dout; a.ldo = C; a.bias = dbias; a.c1f = dc1f; a.bstat = dbstat; a.res16 = dh; a.ldr = C;
    hipLaunchKernelGGL((msf2::proj_astat_bf16_kernel<msf2::A_PLAIN, msf2::ACT_TCN>), dim3((S + 127) / 128, B), dim3(256), 0, 0, a);
    CK(hipDeviceSynchronize());
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((msf2::proj_astat_bf16_kernel<msf2::A_PLAIN, msf2::ACT_TCN>), dim3((S + 127) / 128, B), dim3(256), 0, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%.1f us per launch\n", 100.f * ms);
    }
    std::vector<unsigned short> out(M * C);
    CK(hipMemcpy(out.data(), dout, M * C * 2, hipMemcpyDeviceToHost));
    double worst = 0; long bad = 0, nanc = 0;
    for (int trial = 0; trial < 4000; ++trial) {
        const long m = (trial < 64) ? trial * 61 % M : (long)rand() * 7919 % M;
        const int n = (trial < 64) ? trial * 8 % C : rand() % C;
        const int b = (int)(m / S);
        double g = 0, o = 0;
        for (int k = 0; k < C; ++k) { g += (double)bf2f(h[m * C + k]) * bf2f(wg[(long)n * C + k]); o += (double)bf2f(t[m * C + k]) * bf2f(wo[(long)n * C + k]); }
        const double sg = 1.0 / (1.0 + exp(-(g + bias[n])));
        #ifdef MSF2_DIAG_TCN_NORES
        const double ref = sg *
#else
        const double ref = bf2f(h[m * C + n]) + sg *
#endif
        (bstat[2 * b + 1] * o + c1f[(long)b * C + n]);
        const float got = bf2f(out[m * C + n]);
        if (got != got) { ++nanc; continue; }
        const double err = fabs(got - ref);
        if (err > worst) worst = err;
        if (err > 0.02 * (1 + fabs(ref))) { if (bad < 12) printf("m %ld (b %d, s %ld) n %d: got %f want %f\n", m, b, m % S, n, got, ref); ++bad; }
    }
    printf("B %d S %d: worst abs err %.4g, bad %ld of 4000, nan %ld\n", B, S, worst, bad, nanc);
    return bad || nanc;
}
