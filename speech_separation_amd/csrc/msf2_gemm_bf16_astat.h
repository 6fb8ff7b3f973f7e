// bf16 projection with the activation rows held in REGISTERS ("A-stationary"):
//     out16[m, n] = act(rowscale[m] * sum_k A'[m, k] W[n, k] + bias[n]),   K = 512, N a multiple of 128      (ACT 0 / 1 / 2), or
//     out16[m, n] = rstd[m] * (sum_k A[m, k] W[n, k] - mean[m] * c2[n]) + c1[n]   (ACT_NORM: a folded LayerNorm, N <= 2048).
//
// One workgroup (4 waves, ONE per SIMD, so a wave may use the whole 512-entry register file) owns 128 token rows and
// walks ALL column tiles of 128 outputs.  A wave keeps the MFMA fragments of its 64 rows x 512 k in registers for the
// whole walk (256 registers, loaded once straight from global memory), so
//   * only the WEIGHTS stream global -> LDS (a 4-slot ring of 16 KB stages, whole-line row pieces): half the copy bytes
//     per MFMA of a 128 x 128 tile that streams both operands, and every byte of them L2-resident;
//   * only the weight fragments are read from LDS: half the LDS read traffic per MFMA.
// Two accumulator sets alternate between column tiles: while the MFMAs of column tile c run, the bias / activation
// arithmetic of tile c - 1 is issued between them, two accumulator groups per k tile, in the ACCUMULATOR layout (a lane
// owns one token row and 4 consecutive columns per group), packed to bf16, paired with v_permlane32_swap so that a lane
// stores 16 contiguous bytes, and written -- no LDS transpose, and no epilogue phase in which the matrix pipes idle.
//
// Counted waits.  Copies and output stores retire in issue order on one counter.  The stores of an interior row tile
// (every lane active, so no store instruction is branched around) fall at fixed places of the k loop, which makes the
// number of operations younger than the awaited weight tile a compile-time figure; edge tiles and the first two column
// tiles (whose predecessors issued fewer stores) use the strict counts.
//
// Accumulation order and arithmetic are those of gemm_kernel_bf16_dma (k ascending in MFMA steps of 16, the same
// activation functions and bf16 rounding), so results are bit-identical to that kernel's.
#pragma once
#include <type_traits>
#include "msf2_gemm_bf16.h"

namespace msf2 {

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// The MFMA as an assembly statement, so that the operand files are chosen here: the accumulators in VGPRs (the epilogue
// arithmetic reads them with ordinary VALU instructions), the resident activation fragments in ACCUMULATOR registers
// (all 256 of them; an MFMA reads its A / B operands from either file).  c += w x a, or c = w x a for the first step.
// Nothing else ever writes these registers between MFMAs, consecutive MFMAs go to different accumulator tiles, and the
// first VALU read of a finished tile comes a whole k tile after its last MFMA, so no wait states are needed here.
#ifdef MSF2_DIAG_ASTAT_ACC_AGPR
// experiment: accumulators in the accumulator file, the activation fragments of the first 32 rows there too and those of
// the second 32 rows in VGPRs
template <int TM> __device__ __forceinline__ void mfma_astat(f32x16& c, const u32x4_t& w, const u32x4_t& act) {
    if constexpr (TM == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "a"(act));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(act));
}
template <int TM> __device__ __forceinline__ void mfma_astat_first(f32x16& c, const u32x4_t& w, const u32x4_t& act) {
    if constexpr (TM == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(w), "a"(act));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(w), "v"(act));
}
#else
template <int TM> __device__ __forceinline__ void mfma_astat(f32x16& c, const u32x4_t& w, const u32x4_t& act) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "a"(act));
}
template <int TM> __device__ __forceinline__ void mfma_astat_first(f32x16& c, const u32x4_t& w, const u32x4_t& act) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(w), "a"(act));
}
#endif

constexpr int ACT_NORM = 16;      // epilogue kind: per-row (mean, rstd) from a.rowstat, per-column c1 / c2, no activation
// Two projections of the same rows in one walk (the gate and proj_in of the gated TCN block, mossformer2_block.py:556-566):
// column tiles 0 .. N/128 - 1 are sigmoid(A W^T + bias) -> out16, tiles N/128 .. 2 N/128 - 1 the folded LayerNorm of
// A W2^T -> out16b.  The activation rows are loaded once for both.  Arithmetic: the folded-norm form for every tile
// (rstd' (acc - mean c2') + c1' with c2' = 0, c1' = bias, rstd' = 1 on the gate tiles: the same roundings as r0 acc + bias
// with r0 = 1), the sigmoid phases under a wave-uniform branch on the gate tiles only.
constexpr int ACT_GATE_NORM = 32;
// The two-product contraction of a TCN block (TCNBlock.forward, mossformer2_block.py:508-531, with the gate of this code's fused
// form):   out16[m, n] = h[m, n] + sigmoid(sum_k h[m, k] Wg[n, k] + bias[n]) * (rstd_b * sum_k t[m, k] Wo[n, k] + c1f[b, n])
// (A16 = h, A2_16 = t, W = Wg, W2 = Wo, c1f = the GroupNorm of t folded into a per-utterance column vector, bstat = (mean, rstd) of
// the utterance).  4 x 1 waves; a wave holds 32 rows of h AND of t (128 + 128 registers); a column tile is 64 outputs: per MFMA step
// two gate tiles (h fragment) and two product tiles (t fragment), a weight stage = 64 rows of Wg then 64 rows of Wo.  The residual
// h[m, n] is taken from the resident h fragments (accumulator-file reads and two v_permlane32_swap per 16 columns): no memory
// operation.  Same arithmetic per output as the tile kernel's epilogue.
constexpr int ACT_TCN = 64;

// KDIM = 512: 2 x 2 waves, a wave holds 64 rows x 512 k (two row tiles) and finishes 64 of a column tile's 128 outputs.
// KDIM = 1024: 4 x 1 waves, a wave holds 32 rows x 1024 k (one row tile) and finishes all 128 outputs of the column tile
// (four weight fragments per MFMA step instead of two; twice the k tiles per column tile, so the arithmetic of the previous
// tile has two MFMA shadows per phase).
template <int AMODE, int ACT, int KDIM = 512>
__global__ __launch_bounds__(256, 1) void proj_astat_bf16_kernel(const ProjArgs a) {
    static_assert(KDIM == 512 || KDIM == 1024, "a wave holds 256 registers of activation fragments");
    static_assert(KDIM == 512 || AMODE == A_PLAIN, "the token shift is built for K = 512");
    constexpr bool MERGED = (ACT == ACT_GATE_NORM);
    constexpr bool TCN = (ACT == ACT_TCN);
    static_assert(!TCN || (KDIM == 512 && AMODE == A_PLAIN), "two products of K = 512");
    constexpr bool NORM = (ACT == ACT_NORM) || MERGED;
    constexpr int CW = TCN ? 64 : 128;            // output columns per column tile
    constexpr int NST = 4, STAGE = 128 * 128;     // ring: 4 stages of 128 weight rows x 128 B (k tile of 64)
    constexpr int D = 4;                          // copies per wave and stage (16 of 1 KiB per stage)
    constexpr int KT = KDIM / 64;                 // k tiles per column tile
    constexpr int KS = TCN ? 64 : KDIM / 16;      // resident fragments per row tile (TCN: 32 of h, then 32 of t)
    constexpr int TMW = (KDIM == 512 && !TCN) ? 2 : 1;    // row tiles (32 rows) per wave
    constexpr int TNW = 4 / TMW;                  // MFMA column tiles (32 columns) per wave and column tile
    constexpr int WCN = TMW;                      // waves side by side: 2 (K = 512) or 1
    constexpr int NFT = TCN ? 2 : 4;              // finished 32-column tiles per wave and column tile
    constexpr int KPQ = KT / NFT;                 // k tiles per finished tile of the previous column tile
    constexpr int GPK = 4 / KPQ;                  // accumulator groups finished per k tile: 2 or 1
    constexpr bool COUNTED_OK = (KDIM == 512 && !TCN);    // the counted waits assume a pair of stores every second k tile
    constexpr int MAXN = 4096;
    __shared__ __attribute__((aligned(1024))) char sm[NST * STAGE + MAXN * 4];
    float* sbias = reinterpret_cast<float*>(sm + NST * STAGE);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = (WCN == 2) ? wave >> 1 : wave, wc = (WCN == 2) ? wave & 1 : 0, kh = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y, s0 = blockIdx.x * 128;
    const int S = a.S, K = a.K, NT = MERGED ? 2 * (a.N / CW) : a.N / CW, NT1 = a.N / CW;
    const bool interior = s0 + 128 <= S;

    // ---- bias vector -> LDS (read by the epilogue arithmetic as broadcast 16-byte reads) ----
    for (int i = tid; i < a.N / 4; i += 256) {
        if constexpr (TCN) {          // gate bias; the folded GroupNorm vector of this workgroup's utterance
            reinterpret_cast<float4*>(sbias)[i] = a.bias ? ld4(a.bias + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
            reinterpret_cast<float4*>(sbias + MAXN / 2)[i] = ld4(a.c1f + (long)b * a.N + 4 * i);
        } else if constexpr (MERGED) {       // gate columns: c1' = bias, c2' = 0; proj_in columns behind them
            reinterpret_cast<float4*>(sbias)[i] = a.bias ? ld4(a.bias + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
            reinterpret_cast<float4*>(sbias + MAXN / 2)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            reinterpret_cast<float4*>(sbias + a.N)[i] = ld4(a.c1 + 4 * i);
            reinterpret_cast<float4*>(sbias + MAXN / 2 + a.N)[i] = ld4(a.c2 + 4 * i);
        } else if constexpr (NORM) {
            reinterpret_cast<float4*>(sbias)[i] = ld4(a.c1 + 4 * i);
            reinterpret_cast<float4*>(sbias + MAXN / 2)[i] = ld4(a.c2 + 4 * i);
        } else {
            reinterpret_cast<float4*>(sbias)[i] = a.bias ? ld4(a.bias + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    // ---- this lane's two token rows: scale, output address, activation fragments ----
    float r0[TMW], mu[TMW];                 // row scale (rstd of the folded norm), row mean
    char* orow[TMW];
    bool ok[TMW];
    u32x4_t afr[TMW][KS];
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm) {
        mu[tm] = 0.f;
        const int sr = s0 + wr * (32 * TMW) + tm * 32 + l31;
        ok[tm] = sr < S;
        const int s = min(sr, S - 1);                       // rows past the utterance end are clamped, never stored
        const long m = (long)b * S + s;
        if constexpr (NORM) { mu[tm] = a.rowstat[2 * m]; r0[tm] = a.rowstat[2 * m + 1]; }
        else if constexpr (TCN) r0[tm] = a.bstat[2 * b + 1];          // rstd of the utterance (the mean is folded into c1f)
        else r0[tm] = a.rowstat ? a.rowstat[m] : 1.f;
        orow[tm] = reinterpret_cast<char*>(a.out16) + (m * a.ldo + wc * (32 * TNW)) * 2 + kh * 16;
#ifdef MSF2_DIAG_ASTAT_NOFRAG
        const unsigned short* arow = a.A16 + (m & 31) * a.lda;      // timing-only build: every workgroup's fragments from the same (cache-resident) rows
#else
        const unsigned short* arow = a.A16 + m * a.lda;
#endif
        // token shift: the first half of the features comes from the previous frame, zeros at frame 0
        const unsigned short* prow = (AMODE == A_SHIFT) ? (s > 0 ? arow - a.lda : reinterpret_cast<const unsigned short*>(a.zeros)) : arow;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (TCN) {
#ifdef MSF2_DIAG_ASTAT_NOFRAG
                const unsigned short* base = arow;
#else
                const unsigned short* base = (ks < 32) ? arow : a.A2_16 + m * a.lda;
#endif
                afr[tm][ks] = *reinterpret_cast<const u32x4_t*>(base + (2 * (ks & 31) + kh) * 8);
            } else {
                const int k = (2 * ks + kh) * 8;
                const bool shifted = (AMODE == A_SHIFT) && (ks < KS / 2);
                const unsigned short* src = shifted ? (s > 0 ? prow + k : prow + (k & 63)) : arow + k;
                afr[tm][ks] = *reinterpret_cast<const u32x4_t*>(src);
            }
        }
    }
    // TCN: byte offset of this lane's residual values (row of h, 4 kh columns in) from the uniform base, and the load itself as
    // an assembly statement: the compiler counts only the memory operations it knows of, so a load it issued here would be
    // awaited with a count that ignores the weight copies and stalls on the youngest of them; the wait is written by hand below
    unsigned roff = 0;
    constexpr int RD = 4;                        // residual values are requested RD k tiles before their use (RD divides KT: the buffers rotate in step with the k tiles across column tiles)
    static_assert(KT % RD == 0, "buffer rotation");
    u32x2_t rbuf[RD];
#pragma unroll
    for (int i = 0; i < RD; ++i) rbuf[i] = u32x2_t{0u, 0u};
    if constexpr (TCN) {
        const int s = min(s0 + wr * 32 + l31, S - 1);
        roff = (unsigned)((((long)b * S + s) * a.lda + 4 * kh) * 2);
    }
    auto res_load = [&](u32x2_t& dst, int col) __attribute__((always_inline)) {      // columns col .. col + 3 (+ 4 kh) of the lane's row of h
        const unsigned short* base = a.A16 + col;
#ifndef MSF2_DIAG_TCN_NORES
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(roff), "s"(base) : "memory");
#endif
    };
    __syncthreads();                                       // bias in LDS; nothing of the ring is in flight yet

    // ---- the weight stream ----
    // copy i of a wave moves weight rows (4 wave + i) 8 .. + 7 of the column tile, 128 B each; the 16-byte chunk a lane
    // fetches is swizzled on the SOURCE side (chunk c of row r lands at chunk position c ^ ((r >> 1) & 7))
    unsigned off[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const int row = (wave * D + i) * 8 + (lane >> 3);          // row of the stage; TCN: rows 0-63 are Wg's, 64-127 Wo's
        off[i] = (unsigned)((((TCN ? row & 63 : row) * K) + (((lane & 7) ^ bf_swz128(row)) * 8)) * 2);
    }
    const unsigned short* wbase = reinterpret_cast<const unsigned short*>((TCN && wave >= 2) ? a.W2 : a.W);
    const unsigned short* wnext = wbase;                    // source of the next stage: column tile, k tile
    const unsigned ring = __builtin_amdgcn_readfirstlane(lds_addr_of(sm));
    auto copy_piece = [&](int slot, int i) __attribute__((always_inline)) {
        dma16_so(wnext, off[i], ring + slot * STAGE + (wave * D + i) * 1024);
    };
    auto copy_stage = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < D; ++i) copy_piece(slot, i);
    };
    const long tile_step = (long)CW * K - (KT - 1) * 64;    // from the last k tile of a column tile to the first of the next
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) {                    // stages 0 .. 2 = k tiles 0 .. 2 of column tile 0, and piece 0 of stage 3
        copy_stage(t);
        wnext += 64;
    }
    copy_piece(NST - 1, 0);

    const int rowB = wc * (32 * TNW) + l31;
    const char* fbase = sm + rowB * 128;
    auto read_b1 = [&](int slot, int sub, int tn, u32x4_t& dst) __attribute__((always_inline)) {
#ifdef MSF2_DIAG_ASTAT_NOREAD
        if (slot >= 0) return;         // timing-only build: no fragment reads
#endif
        const int row = rowB + tn * 32;
        dst = *reinterpret_cast<const u32x4_t*>(fbase + slot * STAGE + tn * 32 * 128 + (((2 * sub + kh) ^ bf_swz128(row)) << 4));
    };
    auto read_b = [&](int slot, int sub, u32x4_t (&fb)[TNW]) __attribute__((always_inline)) {
#ifdef MSF2_DIAG_ASTAT_NOREAD
        if (slot >= 0) return;         // timing-only build: no fragment reads
#endif
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn) {
            const int row = rowB + tn * 32;
            fb[tn] = *reinterpret_cast<const u32x4_t*>(fbase + slot * STAGE + tn * 32 * 128 + (((2 * sub + kh) ^ bf_swz128(row)) << 4));
        }
    };

    f32x16 acc[2][TMW][TNW];
    bool st_ok[TMW];                    // whether this lane stores its row of a finished tile (false while there is none)
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm) st_ok[tm] = false;
    u32x2_t pk[4];
    float4 bq[2][2];               // bias (c1) of the two accumulator groups finished per k tile: [k tile parity][group]
    float4 bq2[2][2];              // ACT_NORM: c2 of the same groups
    float val[4] = {0.f, 0.f, 0.f, 0.f};

    // bias of groups g0, g0 + 1 of MFMA tile q of the column tile starting at nb (broadcast reads: two addresses per wave)
    auto load_bias = [&](float4 (&dst)[2], int q, int g0, int nb) __attribute__((always_inline)) {
        const float* p = sbias + nb + wc * (32 * TNW) + (q % TNW) * 32 + 4 * kh;
        dst[0] = *reinterpret_cast<const float4*>(p + 8 * g0);
        if (GPK == 2) dst[1] = *reinterpret_cast<const float4*>(p + 8 * (g0 + 1));
    };
    // (both vectors of the folded norm; the plain epilogue has only the bias)
    auto load_vecs = [&](int par, int q, int g0, int nb) __attribute__((always_inline)) {
        load_bias(bq[par], q, g0, nb);
        if constexpr (NORM || TCN) {
            const float* p = sbias + MAXN / 2 + nb + wc * (32 * TNW) + (q % TNW) * 32 + 4 * kh;
            bq2[par][0] = *reinterpret_cast<const float4*>(p + 8 * g0);
            if (GPK == 2) bq2[par][1] = *reinterpret_cast<const float4*>(p + 8 * (g0 + 1));
        }
    };
    // The arithmetic of one accumulator group (4 values of one row) takes the shadows of 8 MFMAs, one phase per MFMA, each
    // phase a handful of mutually INDEPENDENT instructions (a wave alone on its SIMD issues in order: an instruction that
    // needs the result of the one before it stalls the wave, and with it the next MFMA; the phases of a group depend on each
    // other only across MFMAs).  Slot m = 0 .. 15 of k tile kt is phase m & 7 of accumulator group 2 (kt & 1) + (m >> 3) of
    // MFMA tile kt >> 1 of the PREVIOUS column tile (accumulator set PAR ^ 1).  exp(-x) is written as the multiply and
    // v_exp_f32 that __expf(-x) compiles to, so that the two can sit in different phases.
    // The empty statements pin each phase between the MFMA issued before it and the one after it: they are ordered against
    // the MFMA statements, and the arithmetic hangs on their operands (the optimiser otherwise gathers all of a k tile's
    // arithmetic behind its last MFMA, where nothing hides it).
    float ex[4] = {0.f, 0.f, 0.f, 0.f};
    float ov[4] = {0.f, 0.f, 0.f, 0.f};       // TCN: the normalised product of the group
    bool res_in_loop = true;                  // TCN: the residual loads are pipelined with the weight copies (false: requested and awaited in place)
    int res_col = 0, res_next = 0;            // TCN: first column of the tile being finished / of the one after it
    bool gate_prev = true;         // MERGED: whether the tile being finished is a gate tile (wave-uniform)
    // MERGED: from a row of the first output to the same row of the second, less the second's column offset in the walk (wave-uniform)
    const long out_delta = MERGED ? (reinterpret_cast<char*>(a.out16b) - reinterpret_cast<char*>(a.out16)) - (long)a.N * 2 : 0;
    auto epi_slot = [&](auto par_tag, int kt, int m) __attribute__((always_inline)) {
        constexpr int PP = decltype(par_tag)::value ^ 1;
        constexpr bool SIG = (ACT == 1 || ACT == 2 || MERGED);
        // K = 512: two groups per k tile (slots 0-7, 8-15); K = 1024: one (slots 0-7; slots 8-15 stay empty)
        const int q = kt / KPQ, tm = q / TNW, tn = q % TNW;
        const int gi = TCN ? 0 : m >> 3, g = (kt % KPQ) * GPK + gi, ph = m & 7;
        if constexpr (TCN) {
            // one group per k tile over 16 slots: gate tile q, product tile 2 + q of the previous column tile (res_col = its first column)
            if (m == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = acc[PP][0][q][4 * g + e] + f4get(bq[kt & 1][0], e);
                asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]));
            } else if (m == 1) {
                asm volatile("" : "+v"(r0[0]));
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = r0[0] * acc[PP][0][2 + q][4 * g + e] + f4get(bq2[kt & 1][0], e);
                asm volatile("" : "+v"(ov[0]), "+v"(ov[1]), "+v"(ov[2]), "+v"(ov[3]));
            } else if (m == 2) {
                asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]));
#pragma unroll
                for (int e = 0; e < 4; ++e) ex[e] = val[e] * -1.44269504f;
                asm volatile("" : "+v"(ex[0]), "+v"(ex[1]), "+v"(ex[2]), "+v"(ex[3]));
            } else if (m == 3 || m == 4) {
                const int e0 = (m - 3) * 2;
                asm volatile("" : "+v"(ex[e0]), "+v"(ex[e0 + 1]));
                ex[e0] = __builtin_amdgcn_exp2f(ex[e0]);
                ex[e0 + 1] = __builtin_amdgcn_exp2f(ex[e0 + 1]);
                asm volatile("" : "+v"(ex[e0]), "+v"(ex[e0 + 1]));
            } else if (m == 5) {
                asm volatile("" : "+v"(ex[0]), "+v"(ex[1]), "+v"(ex[2]), "+v"(ex[3]));
#pragma unroll
                for (int e = 0; e < 4; ++e) ex[e] = 1.0f + ex[e];
                asm volatile("" : "+v"(ex[0]), "+v"(ex[1]), "+v"(ex[2]), "+v"(ex[3]));
            } else if (m == 6 || m == 7) {
                const int e0 = (m - 6) * 2;
                asm volatile("" : "+v"(ex[e0]), "+v"(ex[e0 + 1]));
                ex[e0] = __builtin_amdgcn_rcpf(ex[e0]);
                ex[e0 + 1] = __builtin_amdgcn_rcpf(ex[e0 + 1]);
                asm volatile("" : "+v"(ex[e0]), "+v"(ex[e0 + 1]));
            } else if (m == 8) {
                // the residual requested in slot 13 RD k tiles ago: since then one weight piece, then four pieces and a residual load
                // per k tile, then two pieces were issued
#ifndef MSF2_DIAG_TCN_NORES
                if (res_in_loop) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(5 * RD - 2) : "memory");
#endif
                u32x2_t& rb = rbuf[kt % RD];
                asm volatile("" : "+v"(rb));
                val[0] = __builtin_bit_cast(float, rb.x << 16); val[1] = __builtin_bit_cast(float, rb.x & 0xffff0000u);
                val[2] = __builtin_bit_cast(float, rb.y << 16); val[3] = __builtin_bit_cast(float, rb.y & 0xffff0000u);
#ifdef MSF2_DIAG_TCN_NORES
                val[0] = val[1] = val[2] = val[3] = 0.f;
#endif
                asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]));
            } else if (m == 9) {
                asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]));
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = val[e] + ex[e] * ov[e];
                asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]));
            } else if (m == 10) {
                pk[g].x = pk_bf16_2(val[0], val[1]);
                pk[g].y = pk_bf16_2(val[2], val[3]);
                asm volatile("" : "+v"(pk[g].x), "+v"(pk[g].y));
            } else if (m == 13 && res_in_loop) {
                // the residual of the group RD k tiles on: of this column tile, or of the first groups of the one after it
                if (kt + RD < KT) res_load(rbuf[kt % RD], res_col + ((kt + RD) / KPQ) * 32 + 8 * ((kt + RD) % KPQ));
                else res_load(rbuf[kt % RD], res_next + ((kt + RD - KT) / KPQ) * 32 + 8 * ((kt + RD - KT) % KPQ));
            }
            return;
        }
        if (gi >= GPK) return;
        if (ph == 0) {
            asm volatile("" : "+v"(r0[tm]));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (NORM) val[e] = acc[PP][tm][tn][4 * g + e] - mu[tm] * f4get(bq2[kt & 1][gi], e);
                else val[e] = r0[tm] * acc[PP][tm][tn][4 * g + e] + f4get(bq[kt & 1][gi], e);
            }
            asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]));
        } else if (ph == 1) {
            if constexpr (NORM) {
                asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]));
                const float rs = (MERGED && gate_prev) ? 1.f : r0[tm];      // (gate tiles: no row scale)
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = rs * val[e] + f4get(bq[kt & 1][gi], e);
                asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]));
            }
            if constexpr (SIG) {
                asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]));
#pragma unroll
                for (int e = 0; e < 4; ++e) ex[e] = val[e] * -1.44269504f;          // -log2(e)
                asm volatile("" : "+v"(ex[0]), "+v"(ex[1]), "+v"(ex[2]), "+v"(ex[3]));
            }
        } else if (ph == 2 || ph == 3) {
            if (SIG && (!MERGED || gate_prev)) {
                const int e0 = (ph - 2) * 2;
                asm volatile("" : "+v"(ex[e0]), "+v"(ex[e0 + 1]));
                ex[e0] = __builtin_amdgcn_exp2f(ex[e0]);
                ex[e0 + 1] = __builtin_amdgcn_exp2f(ex[e0 + 1]);
                asm volatile("" : "+v"(ex[e0]), "+v"(ex[e0 + 1]));
            }
        } else if (ph == 4) {
            if constexpr (SIG) {
                asm volatile("" : "+v"(ex[0]), "+v"(ex[1]), "+v"(ex[2]), "+v"(ex[3]));
#pragma unroll
                for (int e = 0; e < 4; ++e) ex[e] = 1.0f + ex[e];
                asm volatile("" : "+v"(ex[0]), "+v"(ex[1]), "+v"(ex[2]), "+v"(ex[3]));
            }
        } else if (ph == 5 || ph == 6) {
            if (SIG && (!MERGED || gate_prev)) {
                const int e0 = (ph - 5) * 2;
                asm volatile("" : "+v"(ex[e0]), "+v"(ex[e0 + 1]));
                ex[e0] = __builtin_amdgcn_rcpf(ex[e0]);
                ex[e0 + 1] = __builtin_amdgcn_rcpf(ex[e0 + 1]);
                asm volatile("" : "+v"(ex[e0]), "+v"(ex[e0 + 1]));
            }
        } else {
            asm volatile("" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (ACT == 1) val[e] = val[e] * ex[e];
                else if constexpr (ACT == 2) val[e] = ex[e];
                else if constexpr (MERGED) val[e] = gate_prev ? ex[e] : val[e];
            }
            pk[g].x = pk_bf16_2(val[0], val[1]);
            pk[g].y = pk_bf16_2(val[2], val[3]);
            asm volatile("" : "+v"(pk[g].x), "+v"(pk[g].y));
        }
    };
    // the four packed groups of MFMA tile q -> two 16-byte stores per lane (rows of 64 contiguous bytes per tile)
    auto store_tile = [&](int q, int nb) __attribute__((always_inline)) {
        const int tm = q / TNW, tn = q % TNW;
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
            // lanes 0-31 end up with [own group k | upper half's group k], lanes 32-63 with [lower half's group k + 1 | own]
            const auto rx = __builtin_amdgcn_permlane32_swap(pk[k].x, pk[k + 1].x, false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(pk[k].y, pk[k + 1].y, false, false);
            const uint4 o = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            char* ob = orow[tm];
            if constexpr (MERGED) ob += gate_prev ? 0 : out_delta;
            if (st_ok[tm]) *reinterpret_cast<uint4*>(ob + (nb + tn * 32) * 2 + 16 * k) = o;
        }
    };

    // One column tile: 8 k tiles x 4 MFMA steps x 4 MFMAs into accumulator set PAR, while the previous column tile
    // (set PAR ^ 1) is finished in the shadow of these MFMAs, half a value per MFMA, one MFMA tile (two 16-byte stores)
    // per two k tiles.  The first column tile runs the same code on a zeroed set with its stores masked off, and the
    // weight stream wraps around to column tile 0 after the last one (four stages that nobody reads), so every column
    // tile is the same straight-line code and the stream never has to be wound down inside the loop.
    // The four copies of a stage are spread over four MFMA steps (issuing one holds the wave for tens of cycles: back to
    // back they would leave the matrix pipe idle for most of that): piece 0 right after the barrier that releases the
    // slot, pieces 1 - 3 in the first three MFMA steps of the next k tile.
    // Order of the counted operations per k tile kt: [pieces 1 2 3 of stage kt + 3] [wait for stage kt + 1] [piece 0 of
    // stage kt + 4] [2 stores if kt is odd]: two stages and one pair of stores are younger than the awaited stage.
    u32x4_t fb[2][TNW];            // weight fragments of two consecutive MFMA steps (ping-pong: step parity)
    auto column_tile = [&](auto par_tag, int ct, bool counted) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        const int nprev = (ct - 1) * CW;
        if constexpr (TCN) { res_col = ct > 0 ? nprev : 0; res_next = ct * CW; }      // (the first column tile finishes nothing: any valid column)
        // (the source pointer moves on after piece 3 of a stage; the stage in flight when k tile 4 starts is the last k tile
        // of this column tile, so the pointer then steps to the next column tile -- or wraps)
        const bool wrap_prev = ct == NT - 1;
        if constexpr (MERGED) gate_prev = ct - 1 < NT1;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                // One memory instruction per MFMA shadow at most (issuing one holds the wave for 16 - 60 cycles, and the
                // matrix pipe runs dry behind a wave that is held for longer than one MFMA lasts): after MFMA 0 and 1
                // the two fragment reads of the next MFMA step, after MFMA 2 one piece of the weight stream.
                if (sub == 3) {
                    // the fragments of this step are in registers and the next stage has landed; then every wave is done
                    // reading this stage and it can be refilled
                    if (COUNTED_OK && counted) {
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * D + 2) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * D) : "memory");
                    }
#ifndef MSF2_DIAG_ASTAT_NOBARRIER
                    __builtin_amdgcn_s_barrier();
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (TMW == 2) {
                        if (kt == 0 && sub == 0) {
                            if (j < 2) mfma_astat_first<0>(acc[PAR][0][j & 1], fb[0][j & 1], afr[0][0]);
                            else mfma_astat_first<1>(acc[PAR][TMW - 1][j & 1], fb[0][j & 1], afr[TMW - 1][0]);
                        } else {
                            if (j < 2) mfma_astat<0>(acc[PAR][0][j & 1], fb[sub & 1][j & 1], afr[0][kt * 4 + sub]);
                            else mfma_astat<1>(acc[PAR][TMW - 1][j & 1], fb[sub & 1][j & 1], afr[TMW - 1][kt * 4 + sub]);
                        }
                    } else {
                        const int fo = (TCN && j >= 2) ? 32 : 0;      // TCN: tiles 2, 3 are the product of t (fragments 32 .. 63)
                        if (kt == 0 && sub == 0) mfma_astat_first<0>(acc[PAR][0][j % TNW], fb[0][j % TNW], afr[0][fo]);
                        else mfma_astat<0>(acc[PAR][0][j % TNW], fb[sub & 1][j % TNW], afr[0][fo + kt * 4 + sub]);
                    }
                    if (j < TNW) {
                        if (sub < 3) read_b1(kt & 3, sub + 1, j, fb[(sub + 1) & 1][j]);
                        else read_b1((kt + 1) & 3, 0, j, fb[0][j]);
                    }
                    if (j == 2) {
#ifndef MSF2_DIAG_ASTAT_NOCOPY
                        if (sub < 3) copy_piece((kt + 3) & 3, sub + 1);     // (the stage whose piece 0 went out in the k tile before)
                        else copy_piece(kt & 3, 0);                         // stream tile + 4 goes into the slot just released
#endif
                        if (sub == 2) {
                            if (kt == KT - NST) {
                                if (wrap_prev) wnext = wbase;
                                else if (MERGED && ct == NT1 - 1) wnext = reinterpret_cast<const unsigned short*>(a.W2);     // the second projection's weights
                                else wnext += tile_step;
                            }
                            else wnext += 64;
                        }
                    } else if (j == 3 && sub == 3) {
                        // bias of the group(s) the NEXT k tile finishes (the next column tile's first, after the last k tile)
                        if (kt < KT - 1) load_vecs((kt + 1) & 1, (kt + 1) / KPQ, ((kt + 1) % KPQ) * GPK, nprev);
                        else load_vecs(0, 0, 0, nprev + CW);
                    }
#ifndef MSF2_DIAG_ASTAT_NOEPI
                    epi_slot(par_tag, kt, sub * 4 + j);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#if !defined(MSF2_DIAG_ASTAT_NOEPI) && !defined(MSF2_DIAG_ASTAT_NOSTORE)
            if (kt % KPQ == KPQ - 1) store_tile(kt / KPQ, nprev);
#endif
        }
    };

    // stage 0 has landed once at most the copies of the three younger stages are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * D + 1) : "memory");
    __builtin_amdgcn_s_barrier();
    read_b(0, 0, fb[0]);
    load_vecs(0, 0, 0, 0);
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[1][tm][tn][r] = 0.f;
    typedef std::integral_constant<int, 0> P0;
    typedef std::integral_constant<int, 1> P1;
    for (int ct = 0; ct < NT; ct += 2) {
        column_tile(P0(), ct, interior && ct >= 2);
#pragma unroll
        for (int tm = 0; tm < TMW; ++tm) st_ok[tm] = ok[tm];     // from the second column tile on there is a finished tile to store
        if (ct + 1 < NT) column_tile(P1(), ct + 1, interior && ct + 1 >= 2);
    }
    // the last column tile is finished with nothing to hide behind (same slots, no MFMAs between them)
    const int nlast = (NT - 1) * CW;
    if constexpr (MERGED) gate_prev = false;
    if constexpr (TCN) { res_in_loop = false; res_col = nlast; }
    if (NT & 1) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            load_vecs(kt & 1, kt / KPQ, (kt % KPQ) * GPK, nlast);
            if constexpr (TCN) { if (kt >= RD) { res_load(rbuf[kt % RD], nlast + (kt / KPQ) * 32 + 8 * (kt % KPQ)); } asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#pragma unroll
            for (int m = 0; m < 16; ++m) epi_slot(P1(), kt, m);    // (the tag names the set that would FOLLOW: the slot reads set 0)
            if (kt % KPQ == KPQ - 1) store_tile(kt / KPQ, nlast);
        }
    } else {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            load_vecs(kt & 1, kt / KPQ, (kt % KPQ) * GPK, nlast);
            if constexpr (TCN) { if (kt >= RD) { res_load(rbuf[kt % RD], nlast + (kt / KPQ) * 32 + 8 * (kt % KPQ)); } asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#pragma unroll
            for (int m = 0; m < 16; ++m) epi_slot(P0(), kt, m);
            if (kt % KPQ == KPQ - 1) store_tile(kt / KPQ, nlast);
        }
    }
    // the wrapped-around stages are still landing in this workgroup's LDS: they must have, before it is given away
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace msf2
