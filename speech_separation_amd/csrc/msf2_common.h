// Shared device helpers and launch plumbing for libmsf2 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/msf2.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace msf2 {

extern thread_local int g_last_hip_error;
extern thread_local bool g_bad_operand;     // a launcher refused an operand (alignment / pitch) and skipped its launch

inline int check_launch() {
    if (g_bad_operand) {
        g_bad_operand = false;
        (void)hipGetLastError();
        return MSF2_ERR_ARG;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return MSF2_ERR_LAUNCH;
    }
    return MSF2_OK;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// Bump allocator over the caller's workspace.
struct Arena {
    char* base;
    size_t cap, off;
    Arena(void* p, size_t n) : base((char*)p), cap(n), off(0) {}
    template <class T> T* take(size_t n) {
        size_t bytes = align_up(n * sizeof(T));
        char* r = base ? base + off : nullptr;
        off += bytes;
        return (T*)r;
    }
    bool ok() const { return off <= cap; }
};

// v_exp_f32 + v_rcp_f32 (1 ulp each) instead of an IEEE division: the quotient sequence costs ~10 instructions per
// element, which is most of an epilogue's arithmetic.  exp(-x) = inf gives rcp(inf) = 0.
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }

// two floats -> packed bf16 pair (low half = a), round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pk_bf16_pair(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}

// Sum over each aligned group of 8 lanes, delivered to all 8, by three DPP adds (quad_perm [1,0,3,2], [2,3,0,1], then
// row_half_mirror pairing the two quads): the same additions as xor-shuffles by 1, 2, 4, without the LDS crossbar
// (__shfl_xor is a ds_bpermute_b32: ~100 cycles of latency per step of the chain).
__device__ __forceinline__ float sum8_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a 256-thread block; result valid in every thread. `red` : >= 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

}  // namespace msf2
