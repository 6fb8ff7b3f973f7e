// Calibration: what does a v_mfma_f32_32x32x2_f32 loop sustain on this chip (all CUs
// busy, random operands, 2 blocks of 4 waves per CU), as the ingredients of a real GEMM
// main loop are added one at a time: LDS fragment reads, a barrier per 64 MFMAs,
// LDS writes, global loads.
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE bit 0: LDS fragment reads; bit 1: barrier each iteration; bit 2: LDS writes; bit 3: global loads
template <int MODE>
__global__ __launch_bounds__(256, 2) void mfma_loop(const float* in, float* out, int iters, long stride) {
    __shared__ __attribute__((aligned(16))) float sm[2 * 2 * 128 * 36];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 2 * 128 * 36; i += 256) sm[i] = in[i & 511];
    __syncthreads();
    float4 a0 = make_float4(in[tid], in[tid + 1], in[tid + 2], in[tid + 3]), a1 = a0, b0 = a0, b1 = a0;
    const float* gp = in + (long)blockIdx.x * stride + tid * 4;
    float4 g[8];
    for (int i = 0; i < 8; ++i) g[i] = a0;
    const int rowA = ((tid >> 6) >> 1) * 64 + (lane & 31), rowB = ((tid >> 6) & 1) * 64 + (lane & 31), kh = lane >> 5;
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        if (MODE & 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = *reinterpret_cast<const float4*>(gp + (long)it * 32 + i * 1024 * 16);
            __builtin_amdgcn_sched_barrier(0);
        }
        const float* sa = sm + cur * 2 * 128 * 36;
        const float* sb = sa + 128 * 36;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (MODE & 1) {
                a0 = *reinterpret_cast<const float4*>(&sa[rowA * 36 + c * 8 + kh * 4]);
                a1 = *reinterpret_cast<const float4*>(&sa[(rowA + 32) * 36 + c * 8 + kh * 4]);
                b0 = *reinterpret_cast<const float4*>(&sb[rowB * 36 + c * 8 + kh * 4]);
                b1 = *reinterpret_cast<const float4*>(&sb[(rowB + 32) * 36 + c * 8 + kh * 4]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = j == 0 ? a0.x : j == 1 ? a0.y : j == 2 ? a0.z : a0.w;
                const float x1 = j == 0 ? a1.x : j == 1 ? a1.y : j == 2 ? a1.z : a1.w;
                const float y0 = j == 0 ? b0.x : j == 1 ? b0.y : j == 2 ? b0.z : b0.w;
                const float y1 = j == 0 ? b1.x : j == 1 ? b1.y : j == 2 ? b1.z : b1.w;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[3], 0, 0, 0);
            }
        }
        if (MODE & 4) {
            __builtin_amdgcn_sched_barrier(0);
            float* d = sm + (cur ^ 1) * 2 * 128 * 36;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * (i & 3);
                *reinterpret_cast<float4*>(&d[(i >> 2) * 128 * 36 + (e >> 3) * 36 + (e & 7) * 4]) = g[i];
            }
        }
        if (MODE & 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s + g[0].x;
}

template <int MODE>
void run(const float* in, float* out, long stride, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 512, grid = 512 * 4;
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        mfma_loop<MODE><<<grid, 256>>>(in, out, iters, stride);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double flops = (double)grid * 4 * iters * 64 * 2.0 * 32 * 32 * 2;
    printf("%-40s %.3f ms  %.1f TF/s\n", name, best, flops / best / 1e9);
}

int main() {
    float *in, *out;
    const long stride = 1024 * 16 * 8;          // floats per block region
    const long n = stride * 2048 + 512 * 32 + 65536;
    hipMalloc(&in, n * 4);
    hipMalloc(&out, 4096 * 256 * 4);
    std::vector<float> h(n);
    for (long i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    run<0>(in, out, stride, "bare");
    run<1>(in, out, stride, "+lds reads");
    run<3>(in, out, stride, "+lds reads +barrier");
    run<7>(in, out, stride, "+lds reads +barrier +lds writes");
    run<15>(in, out, stride, "+lds reads +barrier +lds writes +global");
    run<9>(in, out, stride, "+lds reads +global (no barrier/writes)");
    return 0;
}
