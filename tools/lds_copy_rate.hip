// Diagnostic: the rate at which one MI355X moves contraction operands global -> LDS, in the access shape of the bf16
// two-product kernel (per k tile: a piece of 128 activation rows of this workgroup's row tile + a piece of 128 weight
// rows shared by every workgroup), without MFMAs, LDS reads or an epilogue.  Varies the bytes per row piece
// (64 / 128 / 256), the ring depth, workgroups per CU and the source (activations only / weights only / both).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_copy_rate tools/lds_copy_rate.hip && /tmp/lds_copy_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void dma16_so(const void* base, unsigned byte_off, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(byte_off), "s"(base), "s"(lds_base) : "memory", "m0");
}

struct Args {
    const char* h;      // [M][PITCH] bytes, activations of product 1
    const char* t;      // [M][PITCH] bytes, activations of product 2
    const char* w;      // [2][512][PITCH] bytes, weights
    int M;              // rows
    int ncol;           // column tiles a workgroup walks (each re-reads its activation rows)
    int share;          // > 1: `share` consecutive workgroups take the column tiles of ONE row tile (they read the same activation rows)
    int pitch;          // bytes from one row to the next (activations and weights)
    unsigned* sink;
};

constexpr int KBYTES = 1024;  // K = 512 bf16

// ROWB: bytes per row piece; NST: ring slots; a slot = 128 activation rows + 128 weight rows of ROWB bytes.
// SRC: 1 activations only, 2 weights only, 3 both.
// touch: one dword of a line, copied into a 256-byte dummy area of LDS (no register destination, nothing ever waits for it)
__device__ __forceinline__ void touch_so(const void* base, unsigned byte_off, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
                 :: "v"(byte_off), "s"(base), "s"(lds_base) : "memory", "m0");
}

// PF > 0: a fifth wave per workgroup touches the activation lines of the k tile PF tiles ahead of the consumer (its quarter
// of the 128 rows when `share` workgroups read the same row tile), so that the copies find them in the XCD's L2.
template <int ROWB, int NST, int MINB, int SRC, int PF = 0>
__global__ __launch_bounds__((PF > 0 ? 320 : 256), MINB) void copy_rate(Args a) {
    constexpr int TILE = 128 * ROWB, STAGE = 2 * TILE;
    constexpr int RPI = 1024 / ROWB, LPR = ROWB / 16;         // rows per wave instruction, lanes per row
    constexpr int PA = 128 / RPI / 4;                         // copies per wave, operand and k tile
    constexpr int NKT = KBYTES / ROWB;                        // k tiles per product
    const int PITCH = a.pitch;
    extern __shared__ __attribute__((aligned(1024))) char sm[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroups are dealt round-robin over the 8 XCDs (one L2 each): remap so that consecutive logical ids, hence the
    // workgroups that share a row tile, run on ONE XCD
    const unsigned nwg = gridDim.x, L = blockIdx.x, xcd = L & 7, idx = L >> 3, q = nwg >> 3, r = nwg & 7;
    const unsigned lid = a.share > 1 ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx : L;
    const long row0 = (long)(lid / a.share) * 128;
    const int col0 = lid % a.share;
    unsigned off[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (wave * PA + i) * RPI + lane / LPR;
        off[i] = (unsigned)(row * PITCH + (lane % LPR) * 16);
    }
    const int total = a.ncol * 2 * NKT;
    if (PF > 0 && wave == 4) {
        // rows of this helper: 128 / share of them, two 64-byte halves each, spread over the lanes
        const int nshare = a.share > 1 ? a.share : 1;
        const int rows_mine = 128 / nshare, rbase = col0 * rows_mine;
        const unsigned dummy = __builtin_amdgcn_readfirstlane(lds_addr_of(sm + NST * STAGE));
        auto touch_tile = [&](int idx) {
            if (idx >= total) return;
            const int t = idx % (2 * NKT), prod = t / NKT, kt = t % NKT;
            const char* A = (prod ? a.t : a.h) + row0 * PITCH + kt * ROWB;
            for (int e = lane; e < rows_mine * (ROWB / 64); e += 64) {
                const int row = rbase + e / (ROWB / 64), half = e % (ROWB / 64);
                touch_so(A, (unsigned)(row * PITCH + half * 64), dummy);
            }
        };
        for (int i = 0; i < PF + NST - 1; ++i) touch_tile(i);
        for (int c = 0; c < total; ++c) {
            touch_tile(c + PF + NST - 1);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    int issued = 0;
    auto issue = [&]() __attribute__((always_inline)) {
        const int t = issued % (2 * NKT), col = issued / (2 * NKT);
        const int prod = t / NKT, kt = t % NKT;
        const char* A = (prod ? a.t : a.h) + row0 * PITCH + kt * ROWB;
        const char* W = a.w + ((long)prod * 512 + ((col + col0) & 3) * 128) * PITCH + kt * ROWB;
        char* base = sm + (issued % NST) * STAGE;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            if (SRC & 1) dma16_so(A, off[i], __builtin_amdgcn_readfirstlane(lds_addr_of(base + (wave * PA + i) * 1024)));
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            if (SRC & 2) dma16_so(W, off[i], __builtin_amdgcn_readfirstlane(lds_addr_of(base + TILE + (wave * PA + i) * 1024)));
        }
        ++issued;
    };
    for (int i = 0; i < NST - 1; ++i) issue();
    unsigned acc = 0;
    for (int c = 0; c < total; ++c) {
        if (issued < total) issue();
        // tile c has landed once at most the copies of the NST - 1 younger tiles are outstanding
        const int younger = issued - c - 1;
        constexpr int PER = ((SRC & 1) ? PA : 0) + ((SRC & 2) ? PA : 0);
#define WAITN(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((n) * PER > 63 ? 63 : (n) * PER) : "memory")
        switch (younger) {
            case 0: WAITN(0); break;
            case 1: WAITN(1); break;
            case 2: WAITN(2); break;
            case 3: WAITN(3); break;
            case 4: WAITN(4); break;
            case 5: WAITN(5); break;
            case 6: WAITN(6); break;
            default: WAITN(7); break;
        }
        __builtin_amdgcn_s_barrier();
        // touch the landed slot so that nothing is optimised away (one dword per lane)
        acc += *reinterpret_cast<const unsigned*>(sm + (c % NST) * STAGE + threadIdx.x * 4);
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 0x12345678u) a.sink[0] = acc;
}

template <int ROWB, int NST, int MINB, int SRC, int PF = 0>
static void run1(const char* name, Args a, int wgs_per_cu_lds_pad) {
    const int lds = NST * 2 * 128 * ROWB + wgs_per_cu_lds_pad;
    const int threads = PF > 0 ? 320 : 256;
    CK(hipFuncSetAttribute((const void*)copy_rate<ROWB, NST, MINB, SRC, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = a.M / 128 * a.share;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((copy_rate<ROWB, NST, MINB, SRC, PF>), dim3(grid), dim3(threads), lds, 0, a);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy_rate<ROWB, NST, MINB, SRC, PF>), dim3(grid), dim3(threads), lds, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const int nsrc = ((SRC & 1) ? 1 : 0) + ((SRC & 2) ? 1 : 0);
    const double bytes = (double)grid * a.ncol * 2 * KBYTES * 128 * nsrc;
    printf("%-44s lds %6d  %8.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz, 256 CUs)\n", name, lds, 1e3 * ms, bytes / ms / 1e9,
           bytes / (ms * 1e-3) / 256 / 2.4e9);
    fflush(stdout);
}

template <int SRC>
static void run_shared(Args a) {
    a.ncol = 1;
    a.share = 4;
    printf("-- source: %s, one column tile per workgroup, 4 workgroups per row tile, pitch %d\n", SRC == 3 ? "activations + weights" : SRC == 1 ? "activations only" : "weights only", a.pitch);
    run1<64, 4, 2, SRC>("64 B pieces, 4 slots x 16 KB, 2 wg/CU", a, 16384);
    run1<128, 2, 2, SRC>("128 B pieces, 2 slots x 32 KB, 2 wg/CU", a, 16384);
    if (SRC & 1) {
        run1<128, 2, 2, SRC, 1>("  + helper wave touching lines 1 tile ahead", a, 16384);
        run1<128, 2, 2, SRC, 2>("  + helper wave touching lines 2 tiles ahead", a, 16384);
        run1<128, 2, 2, SRC, 4>("  + helper wave touching lines 4 tiles ahead", a, 16384);
        run1<128, 2, 2, SRC, 8>("  + helper wave touching lines 8 tiles ahead", a, 16384);
    }
    run1<128, 4, 1, SRC>("128 B pieces, 4 slots x 32 KB, 1 wg/CU", a, 16384);
    run1<64, 2, 4, SRC>("64 B pieces, 2 slots x 16 KB, 4 wg/CU", a, 0);
}

template <int SRC>
static void run_all(Args a) {
    a.ncol = 2;
    a.share = 1;
    printf("-- source: %s, 2 column tiles per workgroup\n", SRC == 3 ? "activations + weights" : SRC == 1 ? "activations only" : "weights only");
    run1<64, 4, 2, SRC>("64 B pieces, 4 slots x 16 KB, 2 wg/CU", a, 16384);
    run1<128, 2, 2, SRC>("128 B pieces, 2 slots x 32 KB, 2 wg/CU", a, 16384);
    run1<128, 4, 1, SRC>("128 B pieces, 4 slots x 32 KB, 1 wg/CU", a, 16384);
    run1<64, 8, 1, SRC>("64 B pieces, 8 slots x 16 KB, 1 wg/CU", a, 16384);
    run1<256, 2, 1, SRC>("256 B pieces, 2 slots x 64 KB, 1 wg/CU", a, 16384);
    run1<64, 2, 4, SRC>("64 B pieces, 2 slots x 16 KB, 4 wg/CU", a, 0);
    run1<128, 2, 2, SRC>("128 B pieces, 2 slots x 32 KB, 2 wg/CU (no pad)", a, 0);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    const int M = ((B * 3999 + 127) / 128) * 128;
    Args a;
    char *h, *t, *w;
    unsigned* sink;
    const int MAXP = 1280;
    CK(hipMalloc(&h, (size_t)M * MAXP)); CK(hipMalloc(&t, (size_t)M * MAXP)); CK(hipMalloc(&w, 2 * 512 * MAXP));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(h, 1, (size_t)M * MAXP)); CK(hipMemset(t, 2, (size_t)M * MAXP)); CK(hipMemset(w, 3, 2 * 512 * MAXP));
    a.h = h; a.t = t; a.w = w; a.M = M; a.sink = sink; a.pitch = 1024;
    printf("rows %d (batch %d x 3999 frames), %d row tiles\n", M, B, M / 128);
    if (argc > 2) {         // the shared-row pattern at several row pitches
        for (int pitch : {1024}) {
            a.pitch = pitch;
            run_shared<3>(a);
            run_shared<1>(a);
            run_shared<2>(a);
        }
        return 0;
    }
    run_all<3>(a);
    run_all<1>(a);
    run_all<2>(a);
    return 0;
}
