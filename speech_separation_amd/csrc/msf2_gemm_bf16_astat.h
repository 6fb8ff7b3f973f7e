// bf16 projection with the activation rows held in REGISTERS ("A-stationary"):
//     out16[m, n] = act(rowscale[m] * sum_k A'[m, k] W[n, k] + bias[n]),   K = 512, N a multiple of 128.
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
__device__ __forceinline__ void mfma_astat(f32x16& c, const u32x4_t& w, const u32x4_t& act) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "a"(act));
}
__device__ __forceinline__ void mfma_astat_first(f32x16& c, const u32x4_t& w, const u32x4_t& act) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(c) : "v"(w), "a"(act));
}

template <int AMODE, int ACT>
__global__ __launch_bounds__(256, 1) void proj_astat_bf16_kernel(const ProjArgs a) {
    constexpr int NST = 4, STAGE = 128 * 128;     // ring: 4 stages of 128 weight rows x 128 B (k tile of 64)
    constexpr int D = 4;                          // copies per wave and stage (16 of 1 KiB per stage)
    constexpr int KT = 8, KS = 32;                // k tiles and MFMA steps (k 16) for K = 512
    constexpr int MAXN = 4096;
    __shared__ __attribute__((aligned(1024))) char sm[NST * STAGE + MAXN * 4];
    float* sbias = reinterpret_cast<float*>(sm + NST * STAGE);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, kh = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y, s0 = blockIdx.x * 128;
    const int S = a.S, K = a.K, NT = a.N / 128;
    const bool interior = s0 + 128 <= S;

    // ---- bias vector -> LDS (read by the epilogue arithmetic as broadcast 16-byte reads) ----
    for (int i = tid; i < a.N / 4; i += 256)
        reinterpret_cast<float4*>(sbias)[i] = a.bias ? ld4(a.bias + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- this lane's two token rows: scale, output address, activation fragments ----
    float r0[2];
    char* orow[2];
    bool ok[2];
    u32x4_t afr[2][KS];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int sr = s0 + wr * 64 + tm * 32 + l31;
        ok[tm] = sr < S;
        const int s = min(sr, S - 1);                       // rows past the utterance end are clamped, never stored
        const long m = (long)b * S + s;
        r0[tm] = a.rowstat ? a.rowstat[m] : 1.f;
        orow[tm] = reinterpret_cast<char*>(a.out16) + (m * a.ldo + wc * 64) * 2 + kh * 16;
        const unsigned short* arow = a.A16 + m * a.lda;
        // token shift: the first half of the features comes from the previous frame, zeros at frame 0
        const unsigned short* prow = (AMODE == A_SHIFT) ? (s > 0 ? arow - a.lda : reinterpret_cast<const unsigned short*>(a.zeros)) : arow;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = (2 * ks + kh) * 8;
            const bool shifted = (AMODE == A_SHIFT) && (ks < KS / 2);
            const unsigned short* src = shifted ? (s > 0 ? prow + k : prow + (k & 63)) : arow + k;
            afr[tm][ks] = *reinterpret_cast<const u32x4_t*>(src);
        }
    }
    __syncthreads();                                       // bias in LDS; nothing of the ring is in flight yet

    // ---- the weight stream ----
    // copy i of a wave moves weight rows (4 wave + i) 8 .. + 7 of the column tile, 128 B each; the 16-byte chunk a lane
    // fetches is swizzled on the SOURCE side (chunk c of row r lands at chunk position c ^ ((r >> 1) & 7))
    unsigned off[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const int row = (wave * D + i) * 8 + (lane >> 3);
        off[i] = (unsigned)((row * K + (((lane & 7) ^ bf_swz128(row)) * 8)) * 2);
    }
    const unsigned short* wnext = reinterpret_cast<const unsigned short*>(a.W);     // source of the next stage: column tile, k tile
    const unsigned ring = __builtin_amdgcn_readfirstlane(lds_addr_of(sm));
    auto copy_stage = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < D; ++i)
            dma16_so(wnext, off[i], ring + slot * STAGE + (wave * D + i) * 1024);
    };
    const long tile_step = (long)128 * K - (KT - 1) * 64;   // from the last k tile of a column tile to the first of the next
#pragma unroll
    for (int t = 0; t < NST; ++t) {                        // stages 0 .. 3 = k tiles 0 .. 3 of column tile 0
        copy_stage(t);
        wnext += 64;
    }

    const int rowB = wc * 64 + l31;
    const char* fbase = sm + rowB * 128;
    auto read_b = [&](int slot, int sub, u32x4_t (&fb)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int row = rowB + tn * 32;
            fb[tn] = *reinterpret_cast<const u32x4_t*>(fbase + slot * STAGE + tn * 32 * 128 + (((2 * sub + kh) ^ bf_swz128(row)) << 4));
        }
    };

    f32x16 acc[2][2][2];
    bool st_ok[2] = {false, false};     // whether this lane stores its row of a finished tile (false while there is none)
    u32x2_t pk[4];
    float4 bq[2][2];               // bias of the two accumulator groups finished per k tile: [k tile parity][group]
    float xv = 0.f, ex = 0.f, val[4] = {0.f, 0.f, 0.f, 0.f};

    // bias of groups g0, g0 + 1 of MFMA tile q of the column tile starting at nb (broadcast reads: two addresses per wave)
    auto load_bias = [&](float4 (&dst)[2], int q, int g0, int nb) __attribute__((always_inline)) {
        const float* p = sbias + nb + wc * 64 + (q & 1) * 32 + 4 * kh;
        dst[0] = *reinterpret_cast<const float4*>(p + 8 * g0);
        dst[1] = *reinterpret_cast<const float4*>(p + 8 * (g0 + 1));
    };
    // One half of the arithmetic of one output value, small enough to issue in the shadow of one MFMA (each half holds
    // one quarter-rate instruction).  Slot m = 0 .. 15 of k tile kt finishes element (m >> 1) & 3 of accumulator group
    // 2 (kt & 1) + (m >> 3) of MFMA tile kt >> 1 of the PREVIOUS column tile (accumulator set PAR ^ 1).
    auto epi_slot = [&](auto par_tag, int kt, int m) __attribute__((always_inline)) {
        constexpr int PP = decltype(par_tag)::value ^ 1;
        const int q = kt >> 1, tm = q >> 1, tn = q & 1;
        const int gi = m >> 3, g = (kt & 1) * 2 + gi, e = (m >> 1) & 3;
        // The empty statements pin each half between the MFMA issued before it and the one after it: they are ordered
        // against the MFMA statements, and the arithmetic hangs on their operands (the optimiser otherwise gathers all
        // of a k tile's arithmetic behind its last MFMA, where nothing hides it).
        if ((m & 1) == 0) {
            float r = r0[tm];
            asm volatile("" : "+v"(r));
            xv = r * acc[PP][tm][tn][4 * g + e] + f4get(bq[kt & 1][gi], e);
            if constexpr (ACT == 1 || ACT == 2) ex = __expf(-xv);
            asm volatile("" : "+v"(xv), "+v"(ex));
        } else {
            asm volatile("" : "+v"(ex));
            if constexpr (ACT == 1) val[e] = xv * __builtin_amdgcn_rcpf(1.0f + ex);
            else if constexpr (ACT == 2) val[e] = __builtin_amdgcn_rcpf(1.0f + ex);
            else val[e] = xv;
            asm volatile("" : "+v"(val[e]));
            if (e == 3) {
                pk[g].x = pk_bf16_2(val[0], val[1]);
                pk[g].y = pk_bf16_2(val[2], val[3]);
                asm volatile("" : "+v"(pk[g].x), "+v"(pk[g].y));
            }
        }
    };
    // the four packed groups of MFMA tile q -> two 16-byte stores per lane (rows of 64 contiguous bytes per tile)
    auto store_tile = [&](int q, int nb) __attribute__((always_inline)) {
        const int tm = q >> 1, tn = q & 1;
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
            // lanes 0-31 end up with [own group k | upper half's group k], lanes 32-63 with [lower half's group k + 1 | own]
            const auto rx = __builtin_amdgcn_permlane32_swap(pk[k].x, pk[k + 1].x, false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(pk[k].y, pk[k + 1].y, false, false);
            const uint4 o = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            if (st_ok[tm]) *reinterpret_cast<uint4*>(orow[tm] + (nb + tn * 32) * 2 + 16 * k) = o;
        }
    };

    // One column tile: 8 k tiles x 4 MFMA steps x 4 MFMAs into accumulator set PAR, while the previous column tile
    // (set PAR ^ 1) is finished in the shadow of these MFMAs, half a value per MFMA, one MFMA tile (two 16-byte stores)
    // per two k tiles.  The first column tile runs the same code on a zeroed set with its stores masked off, and the
    // weight stream wraps around to column tile 0 after the last one (four stages that nobody reads), so every column
    // tile is the same straight-line code and the stream never has to be wound down inside the loop.
    // Order of the counted operations per k tile kt: [wait for stage kt + 1] [4 copies] ... [2 stores if kt is odd].
    u32x4_t fb[2][2];              // weight fragments of two consecutive MFMA steps (ping-pong: step parity)
    auto column_tile = [&](auto par_tag, int ct, bool counted) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        const int nprev = (ct - 1) * 128;
        const bool wrap = ct == NT - 1;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                if (sub < 3) {
                    read_b(kt & 3, sub + 1, fb[(sub + 1) & 1]);
                } else {
                    // the fragments of MFMA step 3 are in registers and the next stage has landed; then every wave is
                    // done reading this stage and it can be refilled
                    if (counted) {
                        if (kt & 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * D + 2) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * D + 4) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * D) : "memory");
                    }
                    __builtin_amdgcn_s_barrier();
                    read_b((kt + 1) & 3, 0, fb[0]);
                    copy_stage(kt & 3);                             // stream tile + 4 goes into the slot just released
                    if (kt == KT - NST - 1) wnext = wrap ? reinterpret_cast<const unsigned short*>(a.W) : wnext + tile_step;
                    else wnext += 64;
                    // bias of the two groups the NEXT k tile finishes (the next column tile's first, after the last k tile)
                    if (kt < KT - 1) load_bias(bq[(kt + 1) & 1], (kt + 1) >> 1, ((kt + 1) & 1) * 2, nprev);
                    else load_bias(bq[0], 0, 0, nprev + 128);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (kt == 0 && sub == 0) mfma_astat_first(acc[PAR][j >> 1][j & 1], fb[0][j & 1], afr[j >> 1][0]);
                    else mfma_astat(acc[PAR][j >> 1][j & 1], fb[sub & 1][j & 1], afr[j >> 1][kt * 4 + sub]);
                    epi_slot(par_tag, kt, sub * 4 + j);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (kt & 1) store_tile(kt >> 1, nprev);
        }
    };

    // stage 0 has landed once at most the copies of the three younger stages are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * D) : "memory");
    __builtin_amdgcn_s_barrier();
    read_b(0, 0, fb[0]);
    load_bias(bq[0], 0, 0, 0);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[1][tm][tn][r] = 0.f;
    typedef std::integral_constant<int, 0> P0;
    typedef std::integral_constant<int, 1> P1;
    for (int ct = 0; ct < NT; ct += 2) {
        column_tile(P0(), ct, interior && ct >= 2);
        st_ok[0] = ok[0]; st_ok[1] = ok[1];               // from the second column tile on there is a finished tile to store
        if (ct + 1 < NT) column_tile(P1(), ct + 1, interior && ct + 1 >= 2);
    }
    // the last column tile is finished with nothing to hide behind (same slots, no MFMAs between them)
    const int nlast = (NT - 1) * 128;
    if (NT & 1) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            load_bias(bq[kt & 1], kt >> 1, (kt & 1) * 2, nlast);
#pragma unroll
            for (int m = 0; m < 16; ++m) epi_slot(P1(), kt, m);    // (the tag names the set that would FOLLOW: the slot reads set 0)
            if (kt & 1) store_tile(kt >> 1, nlast);
        }
    } else {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            load_bias(bq[kt & 1], kt >> 1, (kt & 1) * 2, nlast);
#pragma unroll
            for (int m = 0; m < 16; ++m) epi_slot(P0(), kt, m);
            if (kt & 1) store_tile(kt >> 1, nlast);
        }
    }
    // the wrapped-around stages are still landing in this workgroup's LDS: they must have, before it is given away
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace msf2
