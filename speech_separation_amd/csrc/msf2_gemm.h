// fp32 contraction kernels on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One kernel template, gemm_kernel<P>, parameterised by a "problem" class P
// that knows where the operand tiles live and what to do with the
// accumulators (epilogue).  Block tile 128x128, K step 32, 256 threads =
// 4 waves; each wave owns a 64x64 (2x2 waves) or 32x128 (4x1 waves) sub-tile
// as 32x32 MFMA accumulators; 2 blocks per CU; LDS double buffered, one
// barrier per K step.
//
// k order.  The MFMA takes one fp32 per lane per operand: lane l supplies row
// (l & 31) at k index (l >> 5).  The sum over k does not care about order as
// long as A and B agree, so within each chunk of 8 k values the lower lane
// half takes k = 0..3 and the upper half k = 4..7, one per MFMA: a lane then
// needs 4 CONSECUTIVE k of its row per chunk = one ds_read_b128 per operand
// tile per 4 MFMA steps.
//
// Staging.  Operands that are k-contiguous in global memory (activations
// [M][K], weights [N][K], attention scores) go global -> LDS directly
// (global_load_lds_dwordx4, no VGPRs, no ds_write): one wave instruction moves
// 8 rows x 128 B into 1 KiB of LDS, lane-linear, so the image is unpadded
// [128 rows][8 x 16 B] and the bank-conflict swizzle is applied to the SOURCE
// address: 16-byte chunk c of row r is stored at chunk position
// c ^ ((r >> 1) & 7); a ds_read_b128 lane group (16 rows, same logical chunk)
// then covers 16 distinct 16-byte slots of the 256-byte bank row.  Where the
// reference needs a zero operand (token shift at frame 0) the source address
// points at a zero buffer.  Operands that are row-contiguous in global memory
// (time-major sources of the attention summaries) are staged through registers
// into a k-major image [k][128 + 4] with unconditional clamped loads and a
// 0/1 mask multiply (predicated loads make hipcc wait inside the fetch).
#pragma once
#include "msf2_common.h"

namespace msf2 {

constexpr int BM = 128, BN = 128, BK = 32;
enum { ORI_KC = 0, ORI_RC = 1 };   // operand tile is k-contiguous / row-contiguous in global memory
template <int ORI, int ROWS = BM> struct LdsTile {      // ROWS: operand rows in the tile (128, or 64 for the narrow B tile)
    static constexpr int SIZE = (ORI == ORI_KC) ? ROWS * BK : BK * (BM + 4);   // floats
};
constexpr int RC_PITCH = BM + 4;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// One wave-level global -> LDS copy: lane l moves 16 bytes from its own source address to
// lds_base + 16 * l (lds_base wave-uniform).  Written as inline assembly on purpose: when hipcc
// sees the builtin form it treats every later ds_read as a possible reader of the copy and
// places s_waitcnt vmcnt(0) in front of it, which serialises the prefetch with the MFMAs it is
// meant to overlap.  Completion is tracked by hand instead (s_waitcnt vmcnt + barrier in the
// callers); the "memory" clobber keeps the compiler's own LDS and global accesses in order.
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void dma16(const void* src, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(src), "s"(lds_base) : "memory", "m0");
}

// the same copy with a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset
__device__ __forceinline__ void dma16_so(const void* base, unsigned byte_off, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(byte_off), "s"(base), "s"(lds_base) : "memory", "m0");
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 mul4(float4 v, float m) { v.x *= m; v.y *= m; v.z *= m; v.w *= m; return v; }
__device__ __forceinline__ float f4get(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
__device__ __forceinline__ int kc_swz(int row) { return (row >> 1) & 7; }
// one float -> bf16 bits, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pk_bf16_1(float a) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v = {a, 0.f};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2)) & 0xffffu;
}
// uniform base + 32-bit byte offset: lowers to the saddr + voffset form of global_load / global_store
__device__ __forceinline__ float ldg_off(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void stg_off(float* base, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
__device__ __forceinline__ float4 ldg4_off(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void stg4_off(float* base, unsigned byte_off, float4 v) {
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// 4 consecutive bf16 <-> float4 (8-byte accesses)
__device__ __forceinline__ float4 ldg4_bf16_off(const unsigned short* base, unsigned byte_off) {
    const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(base) + byte_off);
    return make_float4(__builtin_bit_cast(float, h.x << 16), __builtin_bit_cast(float, h.x & 0xffff0000u),
                       __builtin_bit_cast(float, h.y << 16), __builtin_bit_cast(float, h.y & 0xffff0000u));
}
__device__ __forceinline__ unsigned pk_bf16_2(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}
__device__ __forceinline__ void stg4_bf16_off(unsigned short* base, unsigned byte_off, float4 v) {
    uint2 h;
    h.x = pk_bf16_2(v.x, v.y);
    h.y = pk_bf16_2(v.z, v.w);
    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(base) + byte_off) = h;
}

// ---- k-contiguous operand: 128-row tile = 16 KiB = 16 wave-level DMA instructions, 4 per wave (ROWS / 32 per wave)
template <class P, int SEG, bool IS_A, int ROWS = BM>
__device__ __forceinline__ void dma_tile(const P& prob, float* tile, int kt, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int inst = wave * (ROWS / 32) + i;
        const int row = inst * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ kc_swz(row);          // logical chunk stored at position lane & 7
        const float* src;
        if constexpr (IS_A) src = prob.ptrA(SEG, kt, row, chunk * 4);
        else src = prob.ptrB(SEG, kt, row, chunk * 4);
        dma16(src, __builtin_amdgcn_readfirstlane(lds_addr_of(tile + inst * 256)));
    }
}

// ---- row-contiguous operand: float4 along rows, 4 per thread
__device__ __forceinline__ void rc_coords(int tid, int i, int& r, int& k) {
    const int e = tid + 256 * i;
    k = e >> 5;
    r = (e & 31) * 4;
}

// the 4 operand values lane (row, kh) feeds to the 4 MFMA steps of chunk c
template <int ORI>
__device__ __forceinline__ float4 frag_read(const float* s, int row, int c, int kh) {
    if (ORI == ORI_KC) return *reinterpret_cast<const float4*>(&s[row * BK + (((2 * c + kh) ^ kc_swz(row)) << 2)]);
    const int k = c * 8 + kh * 4;
    return make_float4(s[(k + 0) * RC_PITCH + row], s[(k + 1) * RC_PITCH + row], s[(k + 2) * RC_PITCH + row],
                       s[(k + 3) * RC_PITCH + row]);
}

#ifdef MSF2_DIAG_STAMP
__device__ unsigned long long g_stamp[2048 * 4 * 4];
__device__ unsigned long long g_stamp2[2048 * 4 * 2];
#define MSF2_STAMP(x) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define MSF2_STAMP(x)
#endif

// Problems whose row tile may reach past the end of their rows say how many of its rows exist (valid_rows()); a wave then
// skips the fragment reads and MFMAs of its 32-row sub-tiles that hold no row at all (their outputs are never stored) and
// leaves the matrix pipe of its SIMD to the other workgroup of the CU.
template <class P, class = void> struct HasValidRows { static constexpr bool value = false; };
template <class P> struct HasValidRows<P, decltype((void)static_cast<const P*>(nullptr)->valid_rows())> { static constexpr bool value = true; };
template <int V> struct IntC { static constexpr int value = V; };

template <class P, int SEG, int TM, int TN, int BNT = BN>
__device__ __forceinline__ void run_segment(const P& prob, float* smA, float* smB,
                                            f32x16 (&acc)[TM][TN], int tid, int lane, int wave, int wr, int wc) {
    constexpr int SZA = LdsTile<P::A_ORI>::SIZE, SZB = LdsTile<P::B_ORI, BNT>::SIZE;
    static_assert(BNT == BN || P::B_ORI == ORI_KC, "the narrow B tile is for k-contiguous operands");
    constexpr bool A_DMA = P::A_ORI == ORI_KC, B_DMA = P::B_ORI == ORI_KC;
    const int nk = prob.nk(SEG);
    float4 ra[A_DMA ? 1 : 4], rb[B_DMA ? 1 : 4];
    auto stage = [&](int kt, int buf) {       // issue the global reads of tile kt
        if constexpr (A_DMA) dma_tile<P, SEG, true>(prob, smA + buf * SZA, kt, wave, lane);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { int r, k; rc_coords(tid, i, r, k); ra[i] = prob.fetchA(SEG, kt, r, k); }
        }
        if constexpr (B_DMA) dma_tile<P, SEG, false, BNT>(prob, smB + buf * SZB, kt, wave, lane);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { int r, k; rc_coords(tid, i, r, k); rb[i] = prob.fetchB(SEG, kt, r, k); }
        }
    };
    auto land = [&](int buf) {                // make tile data visible in LDS buffer buf
        if constexpr (!A_DMA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { int r, k; rc_coords(tid, i, r, k); *reinterpret_cast<float4*>(&smA[buf * SZA + k * RC_PITCH + r]) = ra[i]; }
        }
        if constexpr (!B_DMA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { int r, k; rc_coords(tid, i, r, k); *reinterpret_cast<float4*>(&smB[buf * SZB + k * RC_PITCH + r]) = rb[i]; }
        }
        if constexpr (A_DMA || B_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    stage(0, 0);
    land(0);
    const int kh = lane >> 5, l31 = lane & 31;
    const int rowA = wr * (TM * 32) + l31, rowB = wc * (TN * 32) + l31;
    // NTM: the 32-row sub-tiles of this wave that hold rows (TM everywhere but at the ragged end)
    auto compute = [&](int cur, auto ntm_c) {
        constexpr int NTM = decltype(ntm_c)::value;
        if constexpr (NTM == 0) return;
        const float* sa = smA + cur * SZA;
        const float* sb = smB + cur * SZB;
        float4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int t = 0; t < NTM; ++t) af[0][t] = frag_read<P::A_ORI>(sa, rowA + t * 32, 0, kh);
#pragma unroll
        for (int t = 0; t < TN; ++t) bf[0][t] = frag_read<P::B_ORI>(sb, rowB + t * 32, 0, kh);
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            if (c + 1 < BK / 8) {     // fragments of chunk c+1 are read while the MFMAs of chunk c issue
#pragma unroll
                for (int t = 0; t < NTM; ++t) af[(c + 1) & 1][t] = frag_read<P::A_ORI>(sa, rowA + t * 32, c + 1, kh);
#pragma unroll
                for (int t = 0; t < TN; ++t) bf[(c + 1) & 1][t] = frag_read<P::B_ORI>(sb, rowB + t * 32, c + 1, kh);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = P::ACC_T
                            ? __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(bf[c & 1][tn], j), f4get(af[c & 1][tm], j), acc[tm][tn], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_32x32x2f32(f4get(af[c & 1][tm], j), f4get(bf[c & 1][tn], j), acc[tm][tn], 0, 0, 0);
        }
    };
    auto main_loop = [&](auto ntm_c) {
    // steady state is branch-free (a conditional around the staging makes hipcc merge
    // register assignments with moves that wait on the loads); the last tile is peeled
#ifdef MSF2_DIAG_STAMP
    unsigned long long t_stage = 0, t_comp = 0, t_land = 0;
#endif
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int cur = kt & 1;
#ifdef MSF2_DIAG_STAMP
        unsigned long long s0_, s1_, s2_, s3_;
#endif
        MSF2_STAMP(s0_)
        stage(kt + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMAs that hide their latency
        MSF2_STAMP(s1_)
        compute(cur, ntm_c);
        __builtin_amdgcn_sched_barrier(0);
        MSF2_STAMP(s2_)
        land(cur ^ 1);
        MSF2_STAMP(s3_)
#ifdef MSF2_DIAG_STAMP
        t_stage += s1_ - s0_; t_comp += s2_ - s1_; t_land += s3_ - s2_;
#endif
    }
#ifdef MSF2_DIAG_STAMP
    if (lane == 0 && SEG == 0) {
        const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (L < 2048) {
            g_stamp[(L * 4 + wave) * 4 + 0] = t_stage; g_stamp[(L * 4 + wave) * 4 + 1] = t_comp;
            g_stamp[(L * 4 + wave) * 4 + 2] = t_land;  g_stamp[(L * 4 + wave) * 4 + 3] = nk - 1;
        }
    }
#endif
    compute((nk - 1) & 1, ntm_c);
    };
    int ntm = TM;
    if constexpr (HasValidRows<P>::value) ntm = min(max((prob.valid_rows() - wr * (TM * 32) + 31) >> 5, 0), TM);      // (wave-uniform)
    if (ntm >= TM || ntm > 2) main_loop(IntC<TM>{});
    else if (ntm == 2) main_loop(IntC<(TM > 2 ? 2 : TM)>{});
    else if (ntm == 1) main_loop(IntC<1>{});
    else main_loop(IntC<0>{});
    __syncthreads();
}

// BNT: output columns per tile.  128 normally; 64 (projections only) when the 128-wide tiling would give fewer
// workgroups than the chip has CUs -- one utterance per call, the callers' own shape: twice the workgroups, each with
// half the accumulators.  Per output element the arithmetic is the same in both forms (same k order).
template <class P, int BNT = BN>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const typename P::Args args) {
    __shared__ __attribute__((aligned(1024))) float smA[2 * LdsTile<P::A_ORI>::SIZE];
    __shared__ __attribute__((aligned(1024))) float smB[2 * LdsTile<P::B_ORI, BNT>::SIZE];
#ifdef MSF2_DIAG_STAMP
    unsigned long long tk0, tk1, tk2;
    MSF2_STAMP(tk0)
#endif
    P prob;
    if constexpr (BNT == BN) prob.init(args);
    else prob.init_tile(args, BM, BNT);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / P::WC, wc = wave % P::WC;
    constexpr int TM = BM / (32 * P::WR), TN = BNT / (32 * P::WC);
    static_assert(TN >= 1 && TN * 32 * P::WC == BNT, "tile columns must split evenly over the waves");
    f32x16 acc[P::NACC][TM][TN];     // NACC accumulator sets; the K segments of a problem with NACC = 1 share one
#pragma unroll
    for (int s = 0; s < P::NACC; ++s)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][tm][tn][r] = 0.f;
    run_segment<P, 0, TM, TN, BNT>(prob, smA, smB, acc[0], tid, lane, wave, wr, wc);
    if constexpr (P::NSEG == 2) run_segment<P, 1, TM, TN, BNT>(prob, smA, smB, acc[P::NACC - 1], tid, lane, wave, wr, wc);
#ifdef MSF2_DIAG_NOEPI
    {   // timing-only build: keep the accumulators alive with one store per thread
        float sum = 0.f;
        for (int s = 0; s < P::NACC; ++s)
            for (int tm = 0; tm < TM; ++tm)
                for (int tn = 0; tn < TN; ++tn)
                    for (int r = 0; r < 16; ++r) sum += acc[s][tm][tn][r];
        if (sum == 123.456f) *reinterpret_cast<float*>(smA) = sum;
    }
#else
#ifdef MSF2_DIAG_STAMP
    MSF2_STAMP(tk1)
#endif
    // main loop ended with a barrier: LDS is free
    if constexpr (P::EPI_RM) {
        typename P::template ColVecs<(P::PAIRED ? TN / 2 : TN)> cv;
        if constexpr (BNT != BN) prob.finish_utt_stats();      // (the small-grid form may reduce the TCN statistics itself)
        prob.template load_colvecs<TN>(prob.n0, wc, lane, cv);
        prob.template fold_colvecs<TN>(cv);
        prob.template epilogue_rm<TM, TN>(acc, wr, wc, lane, smA + wave * 1024, cv);
    }
    else prob.template epilogue<TM, TN>(acc, wr, wc, lane, smA + wave * 1024);
#ifdef MSF2_DIAG_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MSF2_STAMP(tk2)
    if (lane == 0) {
        const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (L < 2048) { g_stamp2[(L * 4 + wave) * 2 + 0] = tk1 - tk0; g_stamp2[(L * 4 + wave) * 2 + 1] = tk2 - tk1; }
    }
#endif
#endif
}

// XCD-aware, bijective remap of the dispatch index: workgroups are dealt round-robin
// over the 8 XCDs (one L2 each); after the remap, tiles with consecutive logical ids
// run consecutively on ONE XCD, so tiles that re-read the same operand rows hit its L2.
__device__ __forceinline__ unsigned xcd_remap() {
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned xcd = L & 7, idx = L >> 3, q = total >> 3, r = total & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// tile-local coordinates of accumulator register `reg` of MFMA tile (tm, tn)
#define MSF2_ACC_ROW(wr, TM, tm, reg, lane) ((wr) * (TM) * 32 + (tm) * 32 + ((reg) & 3) + 8 * ((reg) >> 2) + 4 * ((lane) >> 5))
#define MSF2_ACC_COL(wc, TN, tn, lane) ((wc) * (TN) * 32 + (tn) * 32 + ((lane) & 31))

// ---------------------------------------------------------------------------
// Projections: out[b, s, n] = epilogue( sum_k A'[b, s, k] * W[n, k] )
// grid = (n tiles, ceil(S / 128), B)
// ---------------------------------------------------------------------------
enum { A_PLAIN = 0, A_SHIFT = 1 };
enum {
    EPI_RS_BIAS_ACT = 0,  // act(rowscale[m] * acc + bias[n]); act 0 none, 1 SiLU, 2 sigmoid, 3 ReLU, 4 PReLU(prelu)
    EPI_NORM = 1,         // act(rstd * (acc - mean * c2[n]) + c1[n]) (+ pe[s, n]); stats per row or per utterance; act 0 / 1
    EPI_BIAS_RES = 2,     // res[m, n] + acc + bias[n]
    EPI_RELU_MUL = 3,     // relu(acc) * res[m, n]
    EPI_TANH_SIG = 4,     // paired columns: tanh(accA + bias[n]) * sigmoid(accB + bias2[n])
    EPI_TCN = 5,          // two accumulators: res + sigmoid(acc0 + bias) * (rstd_b * (acc1 - mean_b * c2) + c1)
    EPI_TCN_G = 6,        // EPI_TCN times the elementwise gate gmul[m, n] (its own kernel: the gate costs 16 registers)
};

struct ProjArgs {
    const float* A;  const float* A2;  int lda;
    const float* W;  const float* W2;  int K;  int N;
    int S;
    float* out;  float* out2;  int ldo;
    const float* bias;  const float* bias2;  const float* c1;  const float* c2;
    const float* rowstat;   // [B*S] inverse norms, or [B*S][2] (mean, rstd)
    const float* bstat;     // [B][2] (mean, rstd)
    const float* c1f;       // EPI_TCN, DMA kernel: [B][N] = c1 - rstd_b * mean_b * c2, written by stats_finalize_kernel
    const float* res;  int ldr;
    const float* pe;        // [S][N]
    const float* gmul;      // EPI_TCN_G: elementwise gate multiplied onto the result, [B*S][ldo]
    const unsigned short* gmul16;  // the same gate as bf16, read by the DMA kernel instead of gmul
    const float* zeros;     // A_SHIFT: >= 128 bytes of zeros (>= K bytes where the DMA kernel may run: its zero-row address advances with k)
    int act;
    float prelu;                   // slope of act 4
    // bf16 mode: optional bf16 images of the A operands (same shape / pitch as A, A2) read by the
    // DMA kernel, and an optional bf16 copy of the output written next to the fp32 one
    const unsigned short* A16;  const unsigned short* A2_16;
    unsigned short* out16;
    const unsigned short* res16;   // residual read as bf16 instead of res (same pitch ldr)
    float* sq_part;                // EPI_NORM / EPI_BIAS_RES, row-major epilogue: optional [M][ceil(N / w)] sums of out^2 over the w columns of a wave's slab
    int ncol_loop;                 // DMA kernel: consecutive column tiles one workgroup walks (0 = 1); grid.x shrinks accordingly
    int store_aware;               // DMA kernel walking several tiles: waits behind an epilogue leave its output stores outstanding
    // EPI_TCN / EPI_TCN_G, small-grid tile forms only (init_tile): the utterance statistics as stage A left them, [B][stat_nblk][2]
    // partial (sum, sum of squares); the kernel reduces them itself (bstat and c1f null) in the order stats_finalize_kernel does.
    const float* stat_part;  int stat_nblk;  float stat_eps;  double stat_count;
    unsigned short* out16b;        // register-resident kernel walking two projections of the same rows: bf16 output of the second (W2, c1, c2, rowstat)
};

template <int AMODE, int EPI>
struct ProjNT {
    typedef ProjArgs Args;
    static constexpr bool PAIRED = (EPI == EPI_TANH_SIG);
    static constexpr int EPI_KIND = EPI, A_MODE = AMODE;
    static constexpr int WR = PAIRED ? 4 : 2, WC = PAIRED ? 1 : 2;
    static constexpr bool IS_TCN = (EPI == EPI_TCN || EPI == EPI_TCN_G);
    static constexpr int NSEG = IS_TCN ? 2 : 1, NACC = NSEG;
    static constexpr int A_ORI = ORI_KC, B_ORI = ORI_KC;
    static constexpr bool A_BF16 = false;
    static constexpr bool B_BF16 = true;    // bf16 kernels read W as bf16
    static constexpr bool ACC_T = true;     // accumulators transposed: MFMA(weight fragment, activation fragment)
    static constexpr bool EPI_RM = true;    // the fp32 kernel finishes tiles with epilogue_rm
    Args a;
    int b, s0, n0;
    float mean_u = 0.f, rstd_u = 1.f;     // utterance statistics reduced by the kernel itself (a.stat_part)
    // (mean, rstd) of utterance b from stage A's partial sums: what stats_finalize_kernel computes, every wave for itself.
    // The loads are issued when the tile is set up (ahead of the operand traffic); the reduction runs where the values
    // are first needed.  At most 64 * STAT_NL partial sums per utterance.
    static constexpr int STAT_NL = 16;
    float2 sp[IS_TCN ? STAT_NL : 1];
    __device__ void load_utt_stats() {
        if constexpr (IS_TCN) {
            if (a.stat_part) {
                const float* p = a.stat_part + (long)b * a.stat_nblk * 2;
#pragma unroll
                for (int j = 0; j < STAT_NL; ++j)
                    sp[j] = *reinterpret_cast<const float2*>(p + 2 * min((int)(threadIdx.x & 63) + 64 * j, a.stat_nblk - 1));
            }
        }
    }
    __device__ void finish_utt_stats() {
        if constexpr (IS_TCN) {
            if (a.stat_part) {
                double s = 0.0, q = 0.0;
#pragma unroll
                for (int j = 0; j < STAT_NL; ++j)
                    if ((int)(threadIdx.x & 63) + 64 * j < a.stat_nblk) {
                        s += (double)sp[j].x;
                        q += (double)sp[j].y;
                    }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
                const double mean = s / a.stat_count;
                double var = q / a.stat_count - mean * mean;
                if (var < 0.0) var = 0.0;
                mean_u = (float)mean;
                rstd_u = (float)(1.0 / sqrt(var + (double)a.stat_eps));
            }
        }
    }
    // utterance statistics of the epilogue: finalized by a kernel before this one (bstat) or reduced here
    __device__ void utt_stats(float& mean_b, float& rstd_b) const {
        mean_b = 0.f; rstd_b = 1.f;
        if (a.bstat) { mean_b = a.bstat[2 * b]; rstd_b = a.bstat[2 * b + 1]; }
        else if (IS_TCN && a.stat_part) { mean_b = mean_u; rstd_b = rstd_u; }
    }
    __device__ void init(const Args& args) {
        a = args;
        const unsigned t = xcd_remap();       // column tiles of one row tile share the A rows
        const unsigned nt = t % gridDim.x, rest = t / gridDim.x;
        b = rest / gridDim.y;
        s0 = (rest % gridDim.y) * BM;
        n0 = nt * (a.ncol_loop > 1 ? a.ncol_loop : 1) * NW;
    }
    static constexpr int NW = PAIRED ? BN / 2 : BN;     // output columns one 128-wide tile covers
    // workgroup -> tile for kernels with their own tile shape: tbm rows x tbn columns per tile
    __device__ void init_tile(const Args& args, int tbm, int tbn) {
        a = args;
        const unsigned t = xcd_remap();
        const unsigned nt = t % gridDim.x, rest = t / gridDim.x;
        b = rest / gridDim.y;
        s0 = (rest % gridDim.y) * tbm;
        n0 = nt * (a.ncol_loop > 1 ? a.ncol_loop : 1) * tbn;
        load_utt_stats();
    }
    __device__ int col_tiles_here(int nw) const {      // tiles of nw columns this workgroup walks (DMA kernel)
        const int left = (a.N - n0 + nw - 1) / nw;
        return min(a.ncol_loop > 1 ? a.ncol_loop : 1, left);
    }
    __device__ int nk(int) const { return a.K / BK; }
    __device__ int valid_rows() const { return a.S - s0; }      // rows of this tile that exist
    // rows past the end of the utterance are clamped (their outputs are never stored)
    __device__ const float* ptrA(int seg, int kt, int r, int ko) const {
        const int s = min(s0 + r, a.S - 1);
        const int k = kt * BK + ko;
        long row = (long)b * a.S + s;
        const float* base = (NSEG == 2 && seg == 1) ? a.A2 : a.A;
        if (AMODE == A_SHIFT) {
            // first half of the features comes from the previous frame, zero at frame 0
            const bool sh = k < (a.K >> 1);
            row -= (sh && s > 0) ? 1 : 0;
            const float* p = base + row * a.lda + k;
            return (sh && s == 0) ? a.zeros : p;
        }
        return base + row * a.lda + k;
    }
    __device__ const float* ptrB(int seg, int kt, int c, int ko) const {
        const float* w = a.W;
        int n = n0 + c;
        if (PAIRED && c >= BN / 2) { n -= BN / 2; w = a.W2; }
        if (NSEG == 2 && seg == 1) w = a.W2;
        n = min(n, a.N - 1);
        return w + (long)n * a.K + kt * BK + ko;
    }
    // bf16 image of the A operand (DMA kernel); same clamping and token shift as ptrA
    __device__ const unsigned short* ptrA16(int seg, int kt, int r, int ko) const {
        const int s = min(s0 + r, a.S - 1);
        const int k = kt * BK + ko;
        long row = (long)b * a.S + s;
#ifdef MSF2_DIAG_SAMEROW
        row = r;      // timing-only build: every workgroup reads the same (cache-resident) rows
#endif
        const unsigned short* base = (NSEG == 2 && seg == 1) ? a.A2_16 : a.A16;
        if (AMODE == A_SHIFT) {
            const bool sh = k < (a.K >> 1);
            row -= (sh && s > 0) ? 1 : 0;
            const unsigned short* p = base + row * a.lda + k;
            return (sh && s == 0) ? reinterpret_cast<const unsigned short*>(a.zeros) : p;
        }
        return base + row * a.lda + k;
    }
    // The same two addresses split into a wave-uniform base and a per-lane byte offset (plain A operand, unpaired
    // columns): the base advances with k, the offsets of a wave's copies stay fixed for a whole product.
    __device__ const unsigned short* baseA16(int seg, int kt) const {
        const unsigned short* base = (NSEG == 2 && seg == 1) ? a.A2_16 : a.A16;
        return base + ((long)b * a.S + s0) * a.lda + kt * BK;
    }
    __device__ unsigned offA16(int r, int ko) const {
        return (unsigned)(((min(s0 + r, a.S - 1) - s0) * a.lda + ko) * 2);
    }
    __device__ const unsigned short* baseB16(int seg, int kt) const {
        return reinterpret_cast<const unsigned short*>((NSEG == 2 && seg == 1) ? a.W2 : a.W) + kt * BK;
    }
    __device__ unsigned offB16(int c, int ko) const {
        return (unsigned)((min(n0 + c, a.N - 1) * a.K + ko) * 2);
    }
    // bf16 residual (a.res16, pitch ldr) of row r of this workgroup's tile, 8 columns from column n0 + c
    __device__ const unsigned short* ptrRes16(int r, int c) const {
        const int s = min(s0 + r, a.S - 1);
        return a.res16 + ((long)b * a.S + s) * a.ldr + min(n0 + c, a.N - 8);
    }
    // bf16 kernels: W / W2 point at bf16 weights [N][K]
    __device__ const unsigned short* ptrB16(int seg, int kt, int c, int ko) const {
        const float* w = a.W;
        int n = n0 + c;
        if (PAIRED && c >= BN / 2) { n -= BN / 2; w = a.W2; }
        if (NSEG == 2 && seg == 1) w = a.W2;
        n = min(n, a.N - 1);
        return reinterpret_cast<const unsigned short*>(w) + (long)n * a.K + kt * BK + ko;
    }
    // Row-major form of the epilogue, used by the fp32 kernel.  Each 32x32 accumulator tile (and its partner for
    // the two-accumulator epilogues) is moved through the wave's 4 KiB LDS tile into row-major registers FIRST
    // (lane l: rows 8 i + (l >> 3), i = 0..3, columns 4 (l & 7) + 0..3) and all arithmetic happens there: a
    // lane's 4 columns are the same for every row and tile of a column tile, so the per-column vectors are 3
    // float4 per lane loaded once, instead of one dependent L2 round trip per accumulator group.  The loads of
    // tile k+1 are issued between the arithmetic of tile k (after which its inputs are dead) and its stores,
    // so waiting for them never drains those stores.
    // Per-column vectors of the row-major epilogue for the column tile starting at n_origin: lane l holds
    // columns 4 (l & 7) + 0..3 of each 32-column MFMA tile.  EPI_TCN folds the utterance statistics in:
    // c1 <- c1 - rstd_b * mean_b * c2, so the epilogue evaluates rstd_b * acc1 + c1.  Every load is issued
    // whether or not the vector exists (a null bias reads the weights and is zeroed afterwards): callers
    // that count outstanding memory operations rely on COLVEC_LOADS.
    template <int TNE> struct ColVecs { float4 bias[TNE], bias2[TNE], c1[TNE], c2[TNE]; };
    static constexpr int COLVEC_KINDS = 1 + ((EPI == EPI_NORM || IS_TCN) ? 2 : 0) + ((EPI == EPI_TANH_SIG) ? 1 : 0);
    // FOLDED: c1 comes from a.c1f (statistics already folded in), c2 is not read.
    template <int TN, bool FOLDED = false>
    __device__ void load_colvecs(int n_origin, int wc, int lane, ColVecs<(PAIRED ? TN / 2 : TN)>& cv) const {
        constexpr int TNE = PAIRED ? TN / 2 : TN;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const int rc = lane & 7;
#pragma unroll
        for (int tn = 0; tn < TNE; ++tn) {
            const int nc = min(n_origin + wc * (TN * 32) + tn * 32 + 4 * rc, a.N - 4);
            cv.bias[tn] = ld4((a.bias ? a.bias : a.W) + nc);
            cv.bias2[tn] = (EPI == EPI_TANH_SIG) ? ld4(a.bias2 + nc) : zero4;
            if constexpr (FOLDED) {
                cv.c1[tn] = ld4(a.c1f + (long)b * a.N + nc);
                cv.c2[tn] = zero4;
            } else {
                cv.c1[tn] = (EPI == EPI_NORM || IS_TCN) ? ld4(a.c1 + nc) : zero4;
                cv.c2[tn] = (EPI == EPI_NORM || IS_TCN) ? ld4(a.c2 + nc) : zero4;
            }
        }
    }
    // second half of load_colvecs: everything that reads the loaded values (the caller may put independent
    // memory traffic between the two)
    template <int TN, bool FOLDED = false>
    __device__ void fold_colvecs(ColVecs<(PAIRED ? TN / 2 : TN)>& cv) const {
        constexpr int TNE = PAIRED ? TN / 2 : TN;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float mean_b, rstd_b;
        utt_stats(mean_b, rstd_b);
#pragma unroll
        for (int tn = 0; tn < TNE; ++tn) {
            if (!a.bias) cv.bias[tn] = zero4;
            if constexpr (IS_TCN && !FOLDED) {
                const float k = rstd_b * mean_b;
                cv.c1[tn] = make_float4(cv.c1[tn].x - k * cv.c2[tn].x, cv.c1[tn].y - k * cv.c2[tn].y,
                                        cv.c1[tn].z - k * cv.c2[tn].z, cv.c1[tn].w - k * cv.c2[tn].w);
            }
        }
    }
    // Row-major form of the epilogue.  RESLDS: the bf16 residual of this wave's rows sits in LDS (rtile: 64 rows
    // x 128 columns, 256 B per row, 16-byte chunk c of row r at chunk position c ^ (r & 7)).
    // after_first: called once, after the arithmetic of an early tile (its accumulator registers are free by then).
    struct NoHook { __device__ void operator()() const {} };
    template <int TM, int TN, bool RESLDS = false, class Hook = NoHook>
    __device__ void epilogue_rm(f32x16 (&acc)[NSEG][TM][TN], int wr, int wc, int lane, float* stage,
                                const ColVecs<(PAIRED ? TN / 2 : TN)>& cv, const char* rtile = nullptr,
                                Hook after_first = Hook()) const {
        float mean_b, rstd_b;
        utt_stats(mean_b, rstd_b);
        constexpr int TNE = PAIRED ? TN / 2 : TN;
        constexpr int NT = TM * TNE;
        constexpr bool ROWSTAT = (EPI == EPI_RS_BIAS_ACT || EPI == EPI_NORM);
        constexpr bool RES = (EPI == EPI_BIAS_RES || EPI == EPI_RELU_MUL || IS_TCN);
        constexpr bool TWO = (IS_TCN || EPI == EPI_TANH_SIG);
        const bool has_rs = a.rowstat != nullptr;
        constexpr bool has_g = (EPI == EPI_TCN_G);
        const unsigned ldr4 = 4u * a.ldr, ldo4 = 4u * a.ldo, m_base = (unsigned)b * a.S;
        const int kh = lane >> 5, ar = lane & 31;      // accumulator layout
        const int rr = lane >> 3, rc = lane & 7;       // row-major layout
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 (&biasv)[TNE] = cv.bias;
        const float4 (&bias2v)[TNE] = cv.bias2;
        const float4 (&c1v)[TNE] = cv.c1;
        const float4 (&c2v)[TNE] = cv.c2;
        float4 rv[4], gv[4];
        uint2 rp[4], gp[4];         // RESLDS (bf16 DMA kernel): residual and gate stay packed bf16 until used
        float r0[4], r1[4];
        auto tile_rows = [&](int idx, unsigned (&mrow)[4], bool (&ok)[4]) {
            const int srow = s0 + wr * (TM * 32) + (idx / TNE) * 32;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = srow + 8 * i + rr;
                ok[i] = s < a.S;
                mrow[i] = m_base + min(s, a.S - 1);
            }
        };
        auto load_tile = [&](int idx) {
            unsigned mrow[4];
            bool ok[4];
            tile_rows(idx, mrow, ok);
            const unsigned nc4 = 4u * min(n0 + wc * (TN * 32) + (idx % TNE) * 32 + 4 * rc, a.N - 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rv[i] = zero4;
                gv[i] = one4;
                r0[i] = 1.f;
                r1[i] = 0.f;
                if constexpr (RESLDS) {
                    const int rrow = (idx / TNE) * 32 + 8 * i + rr;                   // row inside the wave's 64-row tile
                    const int chunk = (wc * (TN * 4) + (idx % TNE) * 4 + (rc >> 1)) ^ (rrow & 7);
                    rp[i] = *reinterpret_cast<const uint2*>(rtile + rrow * 256 + chunk * 16 + (rc & 1) * 8);   // kept packed
                } else {
                    if (RES && !a.res16) rv[i] = ldg4_off(a.res, mrow[i] * ldr4 + nc4);
                    if (RES && a.res16) rv[i] = ldg4_bf16_off(a.res16, (mrow[i] * ldr4 + nc4) >> 1);   // uniform branch
                }
                if (EPI == EPI_NORM && a.pe) rv[i] = ldg4_off(a.pe, (mrow[i] - m_base) * (4u * a.N) + nc4);
                if constexpr (has_g && RESLDS)
                    gp[i] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.gmul16) + ((mrow[i] * ldo4 + nc4) >> 1));
                else if (has_g && a.gmul16) gv[i] = ldg4_bf16_off(a.gmul16, (mrow[i] * ldo4 + nc4) >> 1);   // (uniform branch)
                else if (has_g) gv[i] = ldg4_off(a.gmul, mrow[i] * ldo4 + nc4);
                if (ROWSTAT && has_rs) {
                    if (EPI == EPI_RS_BIAS_ACT) r0[i] = ldg_off(a.rowstat, mrow[i] * 4u);
                    else { r1[i] = ldg_off(a.rowstat, mrow[i] * 8u); r0[i] = ldg_off(a.rowstat, mrow[i] * 8u + 4u); }
                }
            }
        };
        auto transpose = [&](const f32x16& t, float4 (&o)[4]) {
#ifdef MSF2_DIAG_EPI_NOTRANSPOSE
            for (int g = 0; g < 4; ++g) o[g] = make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
            return;
#endif
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(stage + (ar * 8 + ((2 * g + kh) ^ (ar & 7))) * 4) =
                    make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + rr;
                o[i] = *reinterpret_cast<const float4*>(stage + (row * 8 + (rc ^ (row & 7))) * 4);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        };
        constexpr bool SQ = (EPI == EPI_NORM || EPI == EPI_BIAS_RES);
        float ssq[4] = {0.f, 0.f, 0.f, 0.f};
        load_tile(0);
#pragma unroll
        for (int idx = 0; idx < NT; ++idx) {
            const int tm = idx / TNE, tn = idx % TNE;
            float4 x4[4], y4[4];
            transpose(acc[0][tm][tn], x4);
            if constexpr (IS_TCN) transpose(acc[NSEG - 1][tm][tn], y4);
            if constexpr (EPI == EPI_TANH_SIG) transpose(acc[0][tm][tn + TN / 2], y4);
            float4 v[4], w2[4];
            uint2 vp[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float mu = (EPI == EPI_NORM && has_rs) ? r1[i] : mean_b, rs = (EPI == EPI_NORM && has_rs) ? r0[i] : rstd_b;
                float ve[4], we[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = f4get(x4[i], e), y = TWO ? f4get(y4[i], e) : 0.f;
                    const unsigned rpw = (e < 2) ? rp[i].x : rp[i].y, gpw = (e < 2) ? gp[i].x : gp[i].y;
                    const float gve = (has_g && RESLDS) ? __builtin_bit_cast(float, (e & 1) ? (gpw & 0xffff0000u) : (gpw << 16)) : f4get(gv[i], e);
                    const float be = f4get(biasv[tn], e);
                    const float rve = RESLDS ? __builtin_bit_cast(float, (e & 1) ? (rpw & 0xffff0000u) : (rpw << 16)) : f4get(rv[i], e);
                    we[e] = 0.f;
                    if constexpr (EPI == EPI_RS_BIAS_ACT) {
                        ve[e] = r0[i] * x + be;
#ifdef MSF2_DIAG_EPI_NOMATH
                        if (a.act == 77) ve[e] = 0.f;
#else
                        if (a.act == 1) ve[e] = silu_f(ve[e]);
                        else if (a.act == 2) ve[e] = sigmoid_f(ve[e]);
#endif
                        else if (a.act == 3) ve[e] = fmaxf(ve[e], 0.f);
                        else if (a.act == 4) ve[e] = ve[e] >= 0.f ? ve[e] : ve[e] * a.prelu;
                    } else if constexpr (EPI == EPI_NORM) {
                        ve[e] = rs * (x - mu * f4get(c2v[tn], e)) + f4get(c1v[tn], e);
                        if (a.act == 1) ve[e] = silu_f(ve[e]);
                        ve[e] += rve;
                    } else if constexpr (EPI == EPI_BIAS_RES) {
                        ve[e] = rve + x + be;
                    } else if constexpr (EPI == EPI_RELU_MUL) {
                        we[e] = fmaxf(x, 0.f);
                        ve[e] = we[e] * rve;
                    } else if constexpr (EPI == EPI_TANH_SIG) {
                        ve[e] = tanhf(x + be) * sigmoid_f(y + f4get(bias2v[tn], e));
                    } else {  // EPI_TCN
#ifdef MSF2_DIAG_EPI_NOMATH
                        ve[e] = rve + (x + be) * (y + f4get(c1v[tn], e));
#else
                        ve[e] = (rve + sigmoid_f(x + be) * (rstd_b * y + f4get(c1v[tn], e))) * gve;   // c1 holds the folded statistics
#endif
                    }
                }
                if constexpr (RESLDS) {        // bf16 output only (the DMA kernel's two-product form): packed at once
                    vp[i].x = pk_bf16_2(ve[0], ve[1]);
                    vp[i].y = pk_bf16_2(ve[2], ve[3]);
                } else {
                    v[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
                    w2[i] = make_float4(we[0], we[1], we[2], we[3]);
                }
            }
            unsigned mrow[4];
            bool ok[4];
            tile_rows(idx, mrow, ok);
            const int n = n0 + wc * (TN * 32) + tn * 32 + 4 * rc;
            const bool nok = n < a.N;
            const unsigned nc4 = 4u * min(n, a.N - 4);
            if (SQ && a.sq_part) {       // (uniform) partial row sums of squares over this wave's TN * 32 columns
                const float keep = nok ? 1.f : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (tn == 0) ssq[i] = 0.f;
                    ssq[i] += keep * (v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w);
                }
                if (tn == TNE - 1) {
                    const unsigned sw = TN * 32, np = (a.N + sw - 1) / sw, slab = (n0 + wc * sw) / sw;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t = ssq[i];
                        t = sum8_dpp(t);
                        if (rc == 0 && ok[i] && slab < np) a.sq_part[(long)mrow[i] * np + slab] = t;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (idx == 0) after_first();
            if (idx + 1 < NT) load_tile(idx + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#ifdef MSF2_DIAG_EPI_NOSTORE
                if (ok[i] && nok && v[i].x == 123.456f) {
#else
                if (ok[i] && nok) {
#endif
                    const unsigned off = mrow[i] * ldo4 + nc4;
                    if constexpr (RESLDS) {
#ifdef MSF2_DIAG_EPI_NOSTORE
                        if (vp[i].x == 0x12345678u)
#endif
                        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(a.out16) + (off >> 1)) = vp[i];
                        continue;
                    }
                    if (a.out) stg4_off(a.out, off, v[i]);
                    if (EPI == EPI_NORM && a.out2) stg4_off(a.out2, off, v[i]);
                    if (EPI == EPI_RELU_MUL && a.out2) stg4_off(a.out2, off, w2[i]);
                    if (a.out16) stg4_bf16_off(a.out16, off >> 1, v[i]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // Accumulators are held TRANSPOSED (ACC_T: the kernels issue the MFMA with the weight fragment
    // as its first operand): in a 32x32 tile lane l owns token row (l & 31) and, per group
    // g = 0..3, the 4 CONSECUTIVE output columns 8 g + 4 (l >> 5) + 0..3 (register 4 g + e).
    // Each tile is finished in two phases.  Phase 1, in that layout: everything that needs only the
    // accumulators, per-row statistics and per-column vectors (float4 loads shared by a half wave);
    // the result goes to a wave-private 4 KiB LDS tile as one 16-byte write per lane and group
    // (16-byte chunk c of row r stored at chunk c ^ (r & 7): conflict-free both ways).  Phase 2,
    // row-major: lane l takes rows 8 i + (l >> 3), i = 0..3, columns 4 (l & 7) + 0..3, so residual,
    // gate, positional and output accesses are 16-byte (fp32) / 8-byte (bf16) vectors and one wave
    // instruction covers 8 rows x 128 contiguous bytes.  All global loads of a tile are issued up
    // front from clamped addresses; only stores are predicated.  Addresses are uniform base + 32-bit
    // byte offset (the host rejects buffers beyond 4 GiB and pointers that are not 16-byte aligned;
    // N is a multiple of 4).  LDS accesses of one wave execute in program order, so the two phases
    // need no barrier, only the compiler fence.
    // RESLDS: the bf16 residual of this wave's rows was copied into LDS with the operand stream (rtile: 64 rows
    // x 128 columns, 256 B per row, 16-byte chunk c of row r at chunk position c ^ (r & 7)) and is read from there.
    template <int TM, int TN, bool RESLDS = false>
    __device__ void epilogue(f32x16 (&acc)[NSEG][TM][TN], int wr, int wc, int lane, float* stage, const char* rtile = nullptr) const {
        float mean_b = 0.f, rstd_b = 1.f;
        if (a.bstat) { mean_b = a.bstat[2 * b]; rstd_b = a.bstat[2 * b + 1]; }
        constexpr int TNE = PAIRED ? TN / 2 : TN;
        constexpr bool ROWSTAT = (EPI == EPI_RS_BIAS_ACT || EPI == EPI_NORM);
        constexpr bool RES = (EPI == EPI_BIAS_RES || EPI == EPI_RELU_MUL || IS_TCN);
        const bool has_rs = a.rowstat != nullptr;
        constexpr bool has_g = (EPI == EPI_TCN_G);
        const unsigned ldr4 = 4u * a.ldr, ldo4 = 4u * a.ldo, m_base = (unsigned)b * a.S;
        const int kh = lane >> 5, ar = lane & 31;      // accumulator layout
        const int rr = lane >> 3, rc = lane & 7;       // row-major layout
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int srow = s0 + wr * (TM * 32) + tm * 32;
            float r0 = 1.f, r1 = 0.f;
            if (ROWSTAT && has_rs) {
                const unsigned ma = m_base + min(srow + ar, a.S - 1);
                if (EPI == EPI_RS_BIAS_ACT) r0 = ldg_off(a.rowstat, ma * 4u);
                else { r1 = ldg_off(a.rowstat, ma * 8u); r0 = ldg_off(a.rowstat, ma * 8u + 4u); }
            }
            const float mu = (EPI == EPI_NORM && has_rs) ? r1 : mean_b, rs = (EPI == EPI_NORM && has_rs) ? r0 : rstd_b;
            unsigned mrow[4];
            bool ok[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = srow + 8 * i + rr;
                ok[i] = s < a.S;
                mrow[i] = m_base + min(s, a.S - 1);
            }
#pragma unroll
            for (int tn = 0; tn < TNE; ++tn) {
                const int ncol = n0 + wc * (TN * 32) + tn * 32;
                const int n = ncol + 4 * rc;
                const bool nok = n < a.N;
                const unsigned nc4 = 4u * min(n, a.N - 4);
                float4 rv[4], gv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    rv[i] = zero4;
                    gv[i] = one4;
                    if constexpr (RESLDS) {
                        const int rrow = tm * 32 + 8 * i + rr;                   // row inside the wave's 64-row tile
                        const int chunk = (wc * (TN * 4) + tn * 4 + (rc >> 1)) ^ (rrow & 7);
                        const uint2 h = *reinterpret_cast<const uint2*>(rtile + rrow * 256 + chunk * 16 + (rc & 1) * 8);
                        rv[i] = make_float4(__builtin_bit_cast(float, h.x << 16), __builtin_bit_cast(float, h.x & 0xffff0000u),
                                            __builtin_bit_cast(float, h.y << 16), __builtin_bit_cast(float, h.y & 0xffff0000u));
                    } else {
                        if (RES && !a.res16) rv[i] = ldg4_off(a.res, mrow[i] * ldr4 + nc4);
                        if (RES && a.res16) rv[i] = ldg4_bf16_off(a.res16, (mrow[i] * ldr4 + nc4) >> 1);   // uniform branch
                    }
                    if (EPI == EPI_NORM && a.pe) rv[i] = ldg4_off(a.pe, (mrow[i] - m_base) * (4u * a.N) + nc4);
                    if (has_g) gv[i] = ldg4_off(a.gmul, mrow[i] * ldo4 + nc4);
                }
                // phase 1
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nc = min(ncol + 8 * g + 4 * kh, a.N - 4);
#ifdef MSF2_DIAG_NOCONST
                    const float4 bias = make_float4(a.prelu, a.prelu, a.prelu, a.prelu + nc), bias2 = bias, c1 = bias, c2 = bias;
#else
                    const float4 bias = a.bias ? ld4(a.bias + nc) : zero4;
                    const float4 bias2 = (EPI == EPI_TANH_SIG) ? ld4(a.bias2 + nc) : zero4;
                    const float4 c1 = (EPI == EPI_NORM || IS_TCN) ? ld4(a.c1 + nc) : zero4;
                    const float4 c2 = (EPI == EPI_NORM || IS_TCN) ? ld4(a.c2 + nc) : zero4;
#endif
                    float p[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int reg = 4 * g + e;
                        const float x = acc[0][tm][tn][reg];
                        const float be = f4get(bias, e);
                        if constexpr (EPI == EPI_RS_BIAS_ACT) {
                            p[e] = r0 * x + be;
                            if (a.act == 1) p[e] = silu_f(p[e]);
                            else if (a.act == 2) p[e] = sigmoid_f(p[e]);
                            else if (a.act == 3) p[e] = fmaxf(p[e], 0.f);
                            else if (a.act == 4) p[e] = p[e] >= 0.f ? p[e] : p[e] * a.prelu;
                        } else if constexpr (EPI == EPI_NORM) {
                            p[e] = rs * (x - mu * f4get(c2, e)) + f4get(c1, e);
                            if (a.act == 1) p[e] = silu_f(p[e]);
                        } else if constexpr (EPI == EPI_BIAS_RES) {
                            p[e] = x + be;
                        } else if constexpr (EPI == EPI_RELU_MUL) {
                            p[e] = fmaxf(x, 0.f);
                        } else if constexpr (EPI == EPI_TANH_SIG) {
                            const float y = acc[0][tm][tn + TN / 2][reg];
                            p[e] = tanhf(x + be) * sigmoid_f(y + f4get(bias2, e));
                        } else {  // EPI_TCN
                            const float y = acc[NSEG - 1][tm][tn][reg];
                            p[e] = sigmoid_f(x + be) * (rstd_b * (y - mean_b * f4get(c2, e)) + f4get(c1, e));
                        }
                    }
                    *reinterpret_cast<float4*>(stage + (ar * 8 + ((2 * g + kh) ^ (ar & 7))) * 4) = make_float4(p[0], p[1], p[2], p[3]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // phase 2
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 8 * i + rr;
                    const float4 p4 = *reinterpret_cast<const float4*>(stage + (row * 8 + (rc ^ (row & 7))) * 4);
                    float4 v;
                    if constexpr (EPI == EPI_RS_BIAS_ACT || EPI == EPI_TANH_SIG) {
                        v = p4;
                    } else if constexpr (EPI == EPI_NORM || EPI == EPI_BIAS_RES) {
                        v = make_float4(p4.x + rv[i].x, p4.y + rv[i].y, p4.z + rv[i].z, p4.w + rv[i].w);
                    } else if constexpr (EPI == EPI_RELU_MUL) {
                        v = make_float4(p4.x * rv[i].x, p4.y * rv[i].y, p4.z * rv[i].z, p4.w * rv[i].w);
                    } else {  // EPI_TCN
                        v = make_float4((rv[i].x + p4.x) * gv[i].x, (rv[i].y + p4.y) * gv[i].y,
                                        (rv[i].z + p4.z) * gv[i].z, (rv[i].w + p4.w) * gv[i].w);
                    }
                    if (ok[i] && nok) {
                        const unsigned off = mrow[i] * ldo4 + nc4;
                        if (a.out) stg4_off(a.out, off, v);
                        if (EPI == EPI_NORM && a.out2) stg4_off(a.out2, off, v);
                        if (EPI == EPI_RELU_MUL && a.out2) stg4_off(a.out2, off, p4);
                        if (a.out16) stg4_bf16_off(a.out16, off >> 1, v);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);   // one tile at a time: accumulator reads are not hoisted across tiles
            }
        }
    }
};

// ---------------------------------------------------------------------------
// Quadratic attention scores of one 256-frame group:
//   P[z, i, j] = relu(q_i . k_j)^2,  q pre-scaled by 1/group.  grid = (2, 2, B*G)
// Rows past the end of the utterance are computed from clamped (finite) rows:
// such P rows are never read back, and such P columns only ever multiply value
// rows that AttnOut zeroes.
// ---------------------------------------------------------------------------
struct SimArgs {
    const float* q;  const float* k;   // [B*S][128]
    float* P;                          // [B*G][256][256]
    int S, G;
    unsigned short* P16;               // when set, the scores are written as bf16 here instead of fp32 to P
};
template <bool IN16>       // IN16: a.q / a.k point at bf16 values (same indexing), read as they are by the bf16 kernel
struct AttnSimT {
    typedef SimArgs Args;
    static constexpr int WR = 2, WC = 2, NSEG = 1, NACC = 1, A_ORI = ORI_KC, B_ORI = ORI_KC;
    static constexpr bool A_BF16 = IN16, B_BF16 = IN16;
    static constexpr bool ACC_T = false, EPI_RM = false;
    Args a;
    int b, g, m0, n0;
    __device__ void init(const Args& args) {
        a = args;
        b = blockIdx.z / a.G;  g = blockIdx.z % a.G;
        m0 = blockIdx.y * BM;  n0 = blockIdx.x * BN;
    }
    __device__ int nk(int) const { return MSF2_QK_DIM / BK; }
    __device__ const float* row(const float* p, int i, int k) const {
        const int s = min(g * MSF2_GROUP + i, a.S - 1);
        return p + ((long)b * a.S + s) * MSF2_QK_DIM + k;
    }
    __device__ const float* ptrA(int, int kt, int r, int ko) const { return row(a.q, m0 + r, kt * BK + ko); }
    __device__ const float* ptrB(int, int kt, int c, int ko) const { return row(a.k, n0 + c, kt * BK + ko); }
    __device__ const unsigned short* row16(const float* p, int i, int k) const {
        const int s = min(g * MSF2_GROUP + i, a.S - 1);
        return reinterpret_cast<const unsigned short*>(p) + ((long)b * a.S + s) * MSF2_QK_DIM + k;
    }
    __device__ const unsigned short* ptrA16(int, int kt, int r, int ko) const { return row16(a.q, m0 + r, kt * BK + ko); }
    __device__ const unsigned short* ptrB16(int, int kt, int c, int ko) const { return row16(a.k, n0 + c, kt * BK + ko); }
    template <int TM, int TN>
    __device__ void epilogue(f32x16 (&acc)[1][TM][TN], int wr, int wc, int lane, float*) const {
        float* P = a.P + (long)blockIdx.z * MSF2_GROUP * MSF2_GROUP;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int i = m0 + MSF2_ACC_ROW(wr, TM, tm, reg, lane);
                    const int j = n0 + MSF2_ACC_COL(wc, TN, tn, lane);
                    const float x = fmaxf(acc[0][tm][tn][reg], 0.f);
                    if (a.P16) a.P16[(long)blockIdx.z * MSF2_GROUP * MSF2_GROUP + i * MSF2_GROUP + j] = (unsigned short)pk_bf16_1(x * x);
                    else P[i * MSF2_GROUP + j] = x * x;
                }
    }
};

typedef AttnSimT<false> AttnSim;

// ---------------------------------------------------------------------------
// Linear-attention key/value summary, split over time:
//   part[b, sp, d, e] = sum_{s in chunk sp} lin_k[b, s, d] * hu[b, s, e]
// grid = (N / 128, nsplit, B); reduced and scaled by 1/S in kv_finalize_kernel.
// N must be a multiple of 128.
// ---------------------------------------------------------------------------
struct LinKVArgs {
    const float* lk;                 // [B*S][128]
    const float* hu;  int ldhu;      // [B*S][ldhu], first N columns used
    float* part;                     // [B][nsplit][128][N]
    int S, N, chunk;
};
template <bool HU16>       // HU16: a.hu and a.lk point at bf16 values (same indexing)
struct LinKVT {
    typedef LinKVArgs Args;
    static constexpr int WR = 2, WC = 2, NSEG = 1, NACC = 1, A_ORI = ORI_RC, B_ORI = ORI_RC;
    static constexpr int BF16_MIN_BLOCKS = HU16 ? 4 : 3;
    static constexpr bool A_BF16 = false;
    static constexpr bool B_BF16 = false;
    static constexpr bool ACC_T = false, EPI_RM = false;
    Args a;
    int b, k0, k1, n0;
    __device__ void init(const Args& args) {
        a = args;
        b = blockIdx.z;
        k0 = blockIdx.y * a.chunk;
        k1 = min(k0 + a.chunk, a.S);
        n0 = blockIdx.x * BN;
    }
    __device__ int nk(int) const { return (k1 - k0 + BK - 1) / BK; }
    __device__ float4 fetchA(int, int kt, int r, int ko) const {
        const int s = k0 + kt * BK + ko;
        const float m = s < k1 ? 1.f : 0.f;
        const long off = ((long)b * a.S + min(s, k1 - 1)) * MSF2_QK_DIM + r;
        if constexpr (HU16) return mul4(ldg4_bf16_off(reinterpret_cast<const unsigned short*>(a.lk) + off, 0u), m);   // lin_k is bf16 too
        else return mul4(ld4(a.lk + off), m);
    }
    __device__ float4 fetchB(int, int kt, int c, int ko) const {
        const int s = k0 + kt * BK + ko;
        const float m = s < k1 ? 1.f : 0.f;
        const long off = ((long)b * a.S + min(s, k1 - 1)) * a.ldhu + n0 + c;
        if constexpr (HU16) return mul4(ldg4_bf16_off(reinterpret_cast<const unsigned short*>(a.hu) + off, 0u), m);
        else return mul4(ld4(a.hu + off), m);
    }
    template <int TM, int TN>
    __device__ void epilogue(f32x16 (&acc)[1][TM][TN], int wr, int wc, int lane, float*) const {
        float* out = a.part + ((long)b * gridDim.y + blockIdx.y) * MSF2_QK_DIM * a.N;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int d = MSF2_ACC_ROW(wr, TM, tm, reg, lane);
                    const int e = n0 + MSF2_ACC_COL(wc, TN, tn, lane);
                    out[(long)d * a.N + e] = acc[0][tm][tn][reg];
                }
    }
};

typedef LinKVT<false> LinKV;

// ---------------------------------------------------------------------------
// Attention output and gating for one half-group (128 query frames) and 64
// value columns (v and u halves side by side in the B tile):
//   att = [P | lin_q] x [vu rows of the group ; KV],   K = 256 + 128
//   o[m, j] = (att_u * v[m, j]) * sigmoid(att_v * u[m, j])
// grid = (E / 64, 2, B*G), remapped so the column tiles of one half-group share an XCD
// ---------------------------------------------------------------------------
struct AttnOutArgs {
    const float* P;                  // [B*G][256][256]
    const float* lq;                 // [B*S][128]
    const float* hu;  int ldhu;      // v | u | ...
    const float* kv;                 // [B][128][2E]
    float* o;                        // [B*S][E]; may be null when o16 is given
    int S, G, E;
    unsigned short* o16;             // optional bf16 copy of o
    const unsigned short* P16;       // HU16: bf16 scores [B*G][256][256] and bf16 lin_q [B*S][128], read instead of P / lq
    const unsigned short* lq16;
    float* sq_part;                  // optional [B*S][E/64]: sum of o^2 over the 64 columns of each column tile (ScaleNorm of o)
};
template <bool HU16>       // HU16: a.hu points at bf16 values (same indexing)
struct AttnOutT {
    typedef AttnOutArgs Args;
    // two K segments into ONE accumulator set: 0 = scores x the group's value rows, 1 = lin_q x KV
    static constexpr int WR = 4, WC = 1, NSEG = 2, NACC = 1, A_ORI = ORI_KC, B_ORI = ORI_RC;
    static constexpr bool A_BF16 = HU16;       // the bf16 kernel then reads P16 / lq16 instead of P / lq
    static constexpr int BF16_MIN_BLOCKS = HU16 ? 4 : 3;  // bf16 kernel, all-bf16 operands: 128 registers, 4 workgroups (32 KiB of LDS each) per CU
    static constexpr bool B_BF16 = false;
    static constexpr bool ACC_T = true;      // MFMA(value fragment, score fragment): a lane owns 4 consecutive output columns
    static constexpr bool EPI_RM = false;
    Args a;
    int z, b, g, m0, j0;
    __device__ void init(const Args& args) {
        a = args;
        const unsigned t = xcd_remap();
        const unsigned jt = t % gridDim.x, rest = t / gridDim.x;
        z = rest / gridDim.y;
        b = z / a.G;  g = z % a.G;
        m0 = (rest % gridDim.y) * BM;  j0 = jt * (BN / 2);
    }
    __device__ int nk(int seg) const { return (seg == 0 ? MSF2_GROUP : MSF2_QK_DIM) / BK; }
    // seg is a compile-time constant at every call site: each K loop reads one source only
    __device__ const float* ptrA(int seg, int kt, int r, int ko) const {
        const int i = m0 + r;
        if (seg == 0) return a.P + ((long)z * MSF2_GROUP + i) * MSF2_GROUP + kt * BK + ko;
        const int s = min(g * MSF2_GROUP + i, a.S - 1);
        return a.lq + ((long)b * a.S + s) * MSF2_QK_DIM + kt * BK + ko;
    }
    __device__ const unsigned short* ptrA16(int seg, int kt, int r, int ko) const {
        const int i = m0 + r;
        if (seg == 0) return a.P16 + ((long)z * MSF2_GROUP + i) * MSF2_GROUP + kt * BK + ko;
        const int s = min(g * MSF2_GROUP + i, a.S - 1);
        return a.lq16 + ((long)b * a.S + s) * MSF2_QK_DIM + kt * BK + ko;
    }
    __device__ float4 fetchB(int seg, int kt, int c, int ko) const {
        const int e = (c < BN / 2) ? (j0 + c) : (a.E + j0 + c - BN / 2);
        if (seg == 1) return ld4(a.kv + ((long)b * MSF2_QK_DIM + kt * BK + ko) * (2 * a.E) + e);
        const int s = g * MSF2_GROUP + kt * BK + ko;
        const long oq = ((long)b * a.S + min(s, a.S - 1)) * a.ldhu + e;
        const float m = (s >= a.S) ? 0.f : 1.f;      // value rows past the utterance end are zero
        if constexpr (HU16) return mul4(ldg4_bf16_off(reinterpret_cast<const unsigned short*>(a.hu) + oq, 0u), m);
        else return mul4(ld4(a.hu + oq), m);
    }
    // bf16 kernel, value rows of the group (segment 0) when they are bf16: the packed words themselves, zero past the utterance end
    template <int SEG> static constexpr bool b_rc16() { return HU16 && SEG == 0; }
    __device__ uint2 fetchB16(int seg, int kt, int c, int ko) const {
        const int e = (c < BN / 2) ? (j0 + c) : (a.E + j0 + c - BN / 2);
        const int s = g * MSF2_GROUP + kt * BK + ko;
        const long oq = ((long)b * a.S + min(s, a.S - 1)) * a.ldhu + e;
        uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(a.hu) + oq);
        if (s >= a.S) v = make_uint2(0u, 0u);
        return v;
    }
    // Accumulators are transposed (ACC_T): each 32 x 32 tile goes through the wave's 4 KiB LDS tile into row-major
    // registers (lane l: rows 8 i + (l >> 3), columns 4 (l & 7) + 0..3), so v, u and o move as 8-byte (bf16) or
    // 16-byte (fp32) vectors, 8 rows x 32 columns per wave instruction.  All v / u loads are issued up front
    // from clamped rows; only the stores are predicated.
    template <int TM, int TN>
    __device__ void epilogue(f32x16 (&acc)[1][TM][TN], int wr, int wc, int lane, float* stage) const {
        static_assert(TM == 1 && WC == 1, "one 32-row slab per wave");
        const unsigned ldh4 = 4u * a.ldhu, e4 = 4u * a.E;
        const int kh = lane >> 5, ar = lane & 31;      // accumulator layout
        const int rr = lane >> 3, rc = lane & 7;       // row-major layout
        unsigned mrow[4];
        bool ok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = g * MSF2_GROUP + m0 + wr * 32 + 8 * i + rr;
            ok[i] = s < a.S;
            mrow[i] = (unsigned)b * a.S + min(s, a.S - 1);
        }
        float4 v[TN / 2][4], u[TN / 2][4];
#pragma unroll
        for (int tn = 0; tn < TN / 2; ++tn) {
            const unsigned j4 = 4u * (j0 + tn * 32 + 4 * rc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (HU16) {
                    v[tn][i] = ldg4_bf16_off(reinterpret_cast<const unsigned short*>(a.hu), (mrow[i] * ldh4 + j4) >> 1);
                    u[tn][i] = ldg4_bf16_off(reinterpret_cast<const unsigned short*>(a.hu), (mrow[i] * ldh4 + e4 + j4) >> 1);
                } else {
                    v[tn][i] = ldg4_off(a.hu, mrow[i] * ldh4 + j4);
                    u[tn][i] = ldg4_off(a.hu, mrow[i] * ldh4 + e4 + j4);
                }
            }
        }
        auto transpose = [&](const f32x16& t, float4 (&o)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(stage + (ar * 8 + ((2 * q + kh) ^ (ar & 7))) * 4) =
                    make_float4(t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + rr;
                o[i] = *reinterpret_cast<const float4*>(stage + (row * 8 + (rc ^ (row & 7))) * 4);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        };
        float ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tn = 0; tn < TN / 2; ++tn) {
            float4 av[4], au[4];
            transpose(acc[0][0][tn], av);
            transpose(acc[0][0][tn + TN / 2], au);
            const unsigned j4 = 4u * (j0 + tn * 32 + 4 * rc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 r;
                r.x = (au[i].x * v[tn][i].x) * sigmoid_f(av[i].x * u[tn][i].x);
                r.y = (au[i].y * v[tn][i].y) * sigmoid_f(av[i].y * u[tn][i].y);
                r.z = (au[i].z * v[tn][i].z) * sigmoid_f(av[i].z * u[tn][i].z);
                r.w = (au[i].w * v[tn][i].w) * sigmoid_f(av[i].w * u[tn][i].w);
                ssq[i] += r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
                if (ok[i]) {
                    if (a.o) stg4_off(a.o, mrow[i] * e4 + j4, r);
                    if (a.o16) stg4_bf16_off(a.o16, (mrow[i] * e4 + j4) >> 1, r);
                }
            }
        }
        if (a.sq_part) {       // the 8 lanes that share a row add up their 64 columns
            const unsigned np = a.E / (BN / 2), jt = j0 / (BN / 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float t = ssq[i];
                t = sum8_dpp(t);
                if (rc == 0 && ok[i]) a.sq_part[(long)mrow[i] * np + jt] = t;
            }
        }
    }
};

typedef AttnOutT<false> AttnOut;

}  // namespace msf2
