// bf16 contraction kernels (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
// Same problem classes, tiling (128x128 block, 4 waves, 32x32 accumulators) and
// epilogues as the fp32 kernels in msf2_gemm.h -- the accumulator layout of the
// 32x32 MFMA does not depend on the operand type -- but the operands are bf16:
//   * activations stay fp32 in HBM (residual stream, statistics and every
//     elementwise kernel are fp32); they are rounded to bf16 (RNE,
//     v_cvt_pk_bf16_f32) on their way into LDS;
//   * projection weights are stored as bf16 by the host-side packing.
// At bf16 MFMA rates these shapes (K = 512) are bandwidth bound, so the tile
// loop is kept simple: register staging, K step 32, double-buffered LDS, one
// barrier per step, 2-3 workgroups per CU for memory-level parallelism.
//
// LDS image: [128 rows][32 k] bf16 = 64 B per row, 16-byte chunk c of row r at
// chunk position c ^ ((r >> 2) & 3): the 16 rows of a ds_read_b128 lane group
// then cover 16 distinct 16-byte slots of the 256-byte bank row.  An MFMA lane
// (row, half h) takes the 8 consecutive k = 16*step + 8*h .. +7: one
// ds_read_b128.  Row-contiguous (time-major) sources are transposed 4x4 in
// registers on the way in, so every operand ends up in the same image.
#pragma once
#include "msf2_gemm.h"

namespace msf2 {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ int bf_swz(int row) { return (row >> 2) & 3; }
constexpr int BF_TILE = BM * BK * 2;   // bytes of one 128 x 32 bf16 operand tile

enum { SRC_KC_F32 = 0, SRC_KC_BF16 = 1, SRC_RC_F32 = 2, SRC_RC_BF16 = 3 };   // RC_BF16: row-contiguous bf16 B operand, kept packed

template <class P, int SEG, bool IS_A, int SRC>
__device__ __forceinline__ void bf_fetch(const P& prob, int kt, int tid, float4 (&r)[4]) {
    if constexpr (SRC == SRC_KC_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            if constexpr (IS_A) r[i] = ld4(prob.ptrA(SEG, kt, e >> 3, (e & 7) * 4));
            else r[i] = ld4(prob.ptrB(SEG, kt, e >> 3, (e & 7) * 4));
        }
    } else if constexpr (SRC == SRC_KC_BF16) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 256 * i;
            uint4 v;
            if constexpr (IS_A) v = *reinterpret_cast<const uint4*>(prob.ptrA16(SEG, kt, e >> 2, (e & 3) * 8));
            else v = *reinterpret_cast<const uint4*>(prob.ptrB16(SEG, kt, e >> 2, (e & 3) * 8));
            r[i] = __builtin_bit_cast(float4, v);
        }
    } else if constexpr (SRC == SRC_RC_BF16) {
        static_assert(!IS_A, "packed row-contiguous source: B operand only");
#pragma unroll
        for (int i = 0; i < 4; ++i) {       // 4 consecutive columns of k row 4 kq + i as two packed words in .x / .y
            const uint2 v = prob.fetchB16(SEG, kt, (tid & 31) * 4, (tid >> 5) * 4 + i);
            r[i].x = __builtin_bit_cast(float, v.x);
            r[i].y = __builtin_bit_cast(float, v.y);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (IS_A) r[i] = prob.fetchA(SEG, kt, (tid & 31) * 4, (tid >> 5) * 4 + i);
            else r[i] = prob.fetchB(SEG, kt, (tid & 31) * 4, (tid >> 5) * 4 + i);
        }
    }
}

template <int SRC>
__device__ __forceinline__ void bf_put(char* tile, int tid, const float4 (&r)[4]) {
    if constexpr (SRC == SRC_KC_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, row = e >> 3, kq = e & 7;
            uint2 v;
            v.x = pk_bf16(r[i].x, r[i].y);
            v.y = pk_bf16(r[i].z, r[i].w);
            *reinterpret_cast<uint2*>(tile + row * 64 + (((kq >> 1) ^ bf_swz(row)) << 4) + (kq & 1) * 8) = v;
        }
    } else if constexpr (SRC == SRC_KC_BF16) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 256 * i, row = e >> 2, ch = e & 3;
            *reinterpret_cast<uint4*>(tile + row * 64 + ((ch ^ bf_swz(row)) << 4)) = __builtin_bit_cast(uint4, r[i]);
        }
    } else if constexpr (SRC == SRC_RC_BF16) {
        // the same 4 x 4 transpose on packed values: byte permutes instead of unpack, scale and repack
        const int kq = tid >> 5, nq = tid & 31;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = 4 * nq + c;
            const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;      // high / low halves of (second, first) operand
            const unsigned w0 = __builtin_bit_cast(unsigned, (c < 2) ? r[0].x : r[0].y), w1 = __builtin_bit_cast(unsigned, (c < 2) ? r[1].x : r[1].y);
            const unsigned w2 = __builtin_bit_cast(unsigned, (c < 2) ? r[2].x : r[2].y), w3 = __builtin_bit_cast(unsigned, (c < 2) ? r[3].x : r[3].y);
            uint2 v;
            v.x = __builtin_amdgcn_perm(w1, w0, sel);
            v.y = __builtin_amdgcn_perm(w3, w2, sel);
            *reinterpret_cast<uint2*>(tile + row * 64 + (((kq >> 1) ^ bf_swz(row)) << 4) + (kq & 1) * 8) = v;
        }
    } else {
        // r[i] = 4 consecutive rows at k = 4*kq + i: transpose so each row gets its 4 consecutive k
        const int kq = tid >> 5, nq = tid & 31;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = 4 * nq + c;
            uint2 v;
            v.x = pk_bf16(f4get(r[0], c), f4get(r[1], c));
            v.y = pk_bf16(f4get(r[2], c), f4get(r[3], c));
            *reinterpret_cast<uint2*>(tile + row * 64 + (((kq >> 1) ^ bf_swz(row)) << 4) + (kq & 1) * 8) = v;
        }
    }
}

__device__ __forceinline__ bf16x8_t bf_frag(const char* tile, int row, int step, int kh) {
    const uint4 v = *reinterpret_cast<const uint4*>(tile + row * 64 + (((2 * step + kh) ^ bf_swz(row)) << 4));
    return __builtin_bit_cast(bf16x8_t, v);
}

// The 128-byte-row image of the global -> LDS kernel's wide-k geometry: [rows][64 k] bf16, 16-byte chunk c of row r at
// chunk position c ^ ((r >> 1) & 7) (rows are half a bank row apart, so 16 consecutive rows cover 16 distinct slots).
__device__ __forceinline__ int bf_swz128(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ bf16x8_t bf_frag128(const char* tile, int row, int step, int kh) {
    const uint4 v = *reinterpret_cast<const uint4*>(tile + row * 128 + (((2 * step + kh) ^ bf_swz128(row)) << 4));
    return __builtin_bit_cast(bf16x8_t, v);
}

// whether K segment SEG of P reads its row-contiguous B operand as packed bf16 (P::b_rc16<SEG>() where a problem states it)
template <class P, int SEG, class = void> struct RcB16 { static constexpr bool V = false; };
template <class P, int SEG> struct RcB16<P, SEG, decltype((void)P::template b_rc16<SEG>())> { static constexpr bool V = P::template b_rc16<SEG>(); };
template <class P, int SEG = 0> struct BfSrc {
    static constexpr int A = P::A_BF16 ? SRC_KC_BF16 : ((P::A_ORI == ORI_KC) ? SRC_KC_F32 : SRC_RC_F32);
    static constexpr int B = P::B_BF16 ? SRC_KC_BF16 : ((P::B_ORI == ORI_KC) ? SRC_KC_F32 : (RcB16<P, SEG>::V ? SRC_RC_BF16 : SRC_RC_F32));
};

template <class P, int SEG, int TM, int TN>
__device__ __forceinline__ void run_segment_bf16(const P& prob, char* smA, char* smB, f32x16 (&acc)[TM][TN],
                                                 int tid, int lane, int wr, int wc) {
    constexpr int SA = BfSrc<P, SEG>::A, SB = BfSrc<P, SEG>::B;
    const int nk = prob.nk(SEG);
    float4 ra[4], rb[4];
    bf_fetch<P, SEG, true, SA>(prob, 0, tid, ra);
    bf_fetch<P, SEG, false, SB>(prob, 0, tid, rb);
    bf_put<SA>(smA, tid, ra);
    bf_put<SB>(smB, tid, rb);
    __syncthreads();
    const int kh = lane >> 5, l31 = lane & 31;
    const int rowA = wr * (TM * 32) + l31, rowB = wc * (TN * 32) + l31;
    auto compute = [&](int cur) {
        const char* sa = smA + cur * BF_TILE;
        const char* sb = smB + cur * BF_TILE;
        bf16x8_t af[2][TM], bf[2][TN];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int t = 0; t < TM; ++t) af[st][t] = bf_frag(sa, rowA + t * 32, st, kh);
#pragma unroll
            for (int t = 0; t < TN; ++t) bf[st][t] = bf_frag(sb, rowB + t * 32, st, kh);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = P::ACC_T ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[st][tn], af[st][tm], acc[tm][tn], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[st][tm], bf[st][tn], acc[tm][tn], 0, 0, 0);
    };
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int cur = kt & 1;
        bf_fetch<P, SEG, true, SA>(prob, kt + 1, tid, ra);
        bf_fetch<P, SEG, false, SB>(prob, kt + 1, tid, rb);
        __builtin_amdgcn_sched_barrier(0);
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        bf_put<SA>(smA + (cur ^ 1) * BF_TILE, tid, ra);
        bf_put<SB>(smB + (cur ^ 1) * BF_TILE, tid, rb);
        __syncthreads();
    }
    compute((nk - 1) & 1);
    __syncthreads();
}

// workgroups per CU the register allocation must allow (P::BF16_MIN_BLOCKS where a problem states one, else 2)
template <class P, class = void> struct Bf16MinBlocks { static constexpr int V = 2; };
template <class P> struct Bf16MinBlocks<P, decltype((void)P::BF16_MIN_BLOCKS)> { static constexpr int V = P::BF16_MIN_BLOCKS; };

template <class P>
__global__ __launch_bounds__(256, Bf16MinBlocks<P>::V) void gemm_kernel_bf16(const typename P::Args args) {
    __shared__ __attribute__((aligned(16))) char smA[2 * BF_TILE];
    __shared__ __attribute__((aligned(16))) char smB[2 * BF_TILE];
    P prob;
    prob.init(args);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / P::WC, wc = wave % P::WC;
    constexpr int TM = BM / (32 * P::WR), TN = BN / (32 * P::WC);
    f32x16 acc[P::NACC][TM][TN];
#pragma unroll
    for (int s = 0; s < P::NACC; ++s)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][tm][tn][r] = 0.f;
    run_segment_bf16<P, 0, TM, TN>(prob, smA, smB, acc[0], tid, lane, wr, wc);
    if constexpr (P::NSEG == 2) run_segment_bf16<P, 1, TM, TN>(prob, smA, smB, acc[P::NACC - 1], tid, lane, wr, wc);
    prob.template epilogue<TM, TN>(acc, wr, wc, lane, reinterpret_cast<float*>(smA) + (tid >> 6) * 1024);
}

// ---------------------------------------------------------------------------
// bf16 x bf16 projections with BOTH operands already bf16 in HBM: tiles go
// global -> LDS directly (global_load_lds_dwordx4) through a 4-stage ring, three
// tiles in flight per workgroup (2 workgroups per CU, 96 KB outstanding per CU),
// counted vmcnt + raw s_barrier so the prefetch stays in flight across barriers.
// Both K segments of a two-accumulator problem run through one continuous ring.
// One wave instruction moves 16 rows x 64 B; the swizzle is on the source address.
// ---------------------------------------------------------------------------
// Workgroup geometry of the DMA kernel.
//   BIG = 0: 128 x 128 tile, 4 waves, 2 workgroups per CU.
//   BIG = 1: 256 x 128 (two products) or 256 x 256 (one product) tile, 8 waves as 4 x 2, ONE workgroup per
//            CU (still 2 waves per SIMD, 256 registers each).  Per MFMA it moves 0.75x / 0.5x the bytes
//            from L2 into LDS: at 17 B/clk per CU the 128 x 128 form sits at the ceiling of what one CU's
//            load path delivers (nothing else is saturated: MFMA busy 21 %, LDS reads 20 %, HBM 35 %).
template <class P, int BIG> struct DmaGeom;
template <class P> struct DmaGeom<P, 0> {
    static constexpr int THREADS = 256, WR = P::WR, WC = P::WC, BMX = BM, BNX = BN;
    static constexpr int TM = BM / (32 * WR), TN = BN / (32 * WC), NWX = P::NW, NST = 4, MINB = 2, ROWB = 64;
};
template <class P> struct DmaGeom<P, 1> {
    static constexpr int THREADS = 512, WR = 4, WC = 2, TM = 2, TN = (P::NSEG == 2) ? 2 : 4;
    static constexpr int BMX = 256, BNX = WC * TN * 32, NWX = BNX, NST = 4, MINB = 1, ROWB = 64;
};

//   BIG = 2: 128 x 256 tile, 4 waves as 2 x 2 (64 x 128 per wave), 3-slot ring (72 KB, the epilogue tiles reuse slot 0 once
//            the stream has ended), TWO workgroups per CU: 0.75x the copy bytes per MFMA of BIG = 0, and unlike
//            BIG = 1 the epilogue of one workgroup (half of the time of a K = 512 projection) runs under the K loop
//            of the other.
template <class P> struct DmaGeom<P, 2> {
    static constexpr int THREADS = 256, WR = 2, WC = 2, TM = 2, TN = 4;
    static constexpr int BMX = 128, BNX = 256, NWX = 256, NST = 3, MINB = 2, ROWB = 64;
};

//   BIG = 3: 128 x 128 tile, 4 waves, 3-slot ring with the epilogue tiles over slot 0 (48 KB), THREE workgroups per CU
//            (at most 168 registers per lane).
template <class P> struct DmaGeom<P, 3> {
    static constexpr int THREADS = 256, WR = 2, WC = 2, TM = 2, TN = 2;
    static constexpr int BMX = 128, BNX = 128, NWX = 128, NST = 3, MINB = 3, ROWB = 64;
};

//   BIG = 4: as BIG = 3 with a 2-slot ring (32 KB): FOUR workgroups per CU, at most 128 registers per lane.
template <class P> struct DmaGeom<P, 4> {
    static constexpr int THREADS = 256, WR = 2, WC = 2, TM = 2, TN = 2;
    static constexpr int BMX = 128, BNX = 128, NWX = 128, NST = 2, MINB = 4, ROWB = 64;
};

//   BIG = 7: 64 x 128 tile, 4 waves as 2 x 2 (32 x 64 per wave), 4-slot ring of 12 KB stages, two workgroups per CU.  For
//            grids the 128-row tiling cannot spread over the chip (one utterance per call: 32 row tiles x 4 column
//            tiles on 256 CUs): twice the workgroups.  One- and two-product problems; the two-product form reads its
//            bf16 residual in the epilogue (no residual tiles in the ring) and keeps both accumulator sets (64 registers).
template <class P> struct DmaGeom<P, 7> {
    static constexpr int THREADS = 256, WR = 2, WC = 2, TM = 1, TN = 2;
    static constexpr int BMX = 64, BNX = 128, NWX = 128, NST = 4, MINB = 2, ROWB = 64;
};

//   BIG = 9: BIG = 7 on whole-line row pieces (k tiles of 64, 24 KB per stage, 4-slot ring): ONE workgroup per CU, for grids of
//            at most one workgroup per CU, where a CU's copy rate is what a tile costs (72 KB in flight per CU).
template <class P> struct DmaGeom<P, 9> {
    static constexpr int THREADS = 256, WR = 2, WC = 2, TM = 1, TN = 2;
    static constexpr int BMX = 64, BNX = 128, NWX = 128, NST = 4, MINB = 1, ROWB = 128;
};

//   BIG = 10: BIG = 9 with a 2-slot ring (48 KB): two workgroups per CU.
template <class P> struct DmaGeom<P, 10> {
    static constexpr int THREADS = 256, WR = 2, WC = 2, TM = 1, TN = 2;
    static constexpr int BMX = 64, BNX = 128, NWX = 128, NST = 2, MINB = 2, ROWB = 128;
};

//   BIG = 8: 128 x 128 tile, 4 waves, k tiles of 64: row pieces of 128 bytes = whole cache lines (a 64-byte piece makes the
//            CU's vector cache fetch the whole line from L2, and the other half again one k tile later: L2-resident
//            operands stream 1.6x faster in whole lines).  32 KB per stage, 2-slot ring, two workgroups per CU; ONE column
//            tile per workgroup: the column tiles of a row tile are separate workgroups that run side by side on one XCD and
//            read the same activation pieces at the same time, so the fabric delivers them once.  The two-product form
//            takes its bf16 residual tile through the ring as the last stage.  K must be a multiple of 64.
template <class P> struct DmaGeom<P, 8> {
    static constexpr int THREADS = 256, WR = 2, WC = 2, TM = 2, TN = 2;
    static constexpr int BMX = 128, BNX = 128, NWX = 128, NST = 2, MINB = 2, ROWB = 128;
};

template <class P, int BIG>
__global__ __launch_bounds__((DmaGeom<P, BIG>::THREADS), (DmaGeom<P, BIG>::MINB)) void gemm_kernel_bf16_dma(const typename P::Args args) {
    typedef DmaGeom<P, BIG> G;
    constexpr bool SMALL = (BIG == 0);                  // the 128 x 128 stream form: several column tiles, both products, residual tiles
    static_assert(SMALL || BIG >= 7 || (!P::PAIRED && P::NSEG == 1), "paired-column and two-product problems keep the 128 x 128 geometry");
    static_assert(!P::PAIRED || SMALL, "paired columns: 128 x 128 geometry only");
    constexpr int NST = G::NST, TM = G::TM, TN = G::TN, NWAVES = G::THREADS / 64;
    constexpr int ROWB = G::ROWB;                       // bytes of a row piece of one k tile: 64 (k tile 32) or 128 (k tile 64)
    constexpr int KT = ROWB / 2, KTU = KT / BK;         // k per tile; in units of the problem's k-tile argument
    constexpr int NSUB = ROWB / 32;                     // MFMA steps (k 16) per k tile
    constexpr int RPI = 1024 / ROWB, LPR = ROWB / 16;   // rows per copy instruction; lanes per row
    constexpr int TILE_A = G::BMX * ROWB, TILE_B = G::BNX * ROWB, STAGE = TILE_A + TILE_B;   // bytes
    constexpr int PA = G::BMX / RPI / NWAVES, PB = G::BNX / RPI / NWAVES;                // copies per wave and k tile: 1 KiB each
    constexpr int DMA_PER_TILE = PA + PB;
    static_assert(PA >= 1 && PB >= 1 && PA * RPI * NWAVES == G::BMX && PB * RPI * NWAVES == G::BNX, "tile rows must split evenly over the waves");
    constexpr bool STAGE_IN_RING = (BIG >= 2 && BIG <= 4);    // the epilogue tiles lie over slot 0 (no copy is in flight by then)
    static_assert(!STAGE_IN_RING || NWAVES * 4096 <= STAGE, "the epilogue tiles must fit one ring slot");
    __shared__ __attribute__((aligned(1024))) char sm[NST * STAGE + (STAGE_IN_RING ? 0 : NWAVES * 4096)];   // ring (+ one 4 KiB epilogue tile per wave)
    P prob;
    if constexpr (!SMALL) prob.init_tile(args, G::BMX, G::NWX);
    else prob.init(args);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / G::WC, wc = wave % G::WC;
    // A workgroup walks `nrep` consecutive column tiles of its row tile; the k tiles of all of
    // them form one continuous stream through the ring, so the first tiles of the next column
    // tile are already in flight while the epilogue of the current one runs.
    //
    // RESLDS (the two-product TCN contraction): the bf16 residual tile of the epilogue (128 rows x 128
    // columns = 2 ring stages of 64 rows x 256 B) rides the same stream as two extra tiles after the k
    // tiles of each column tile and is read from LDS.  Loaded from global memory by the epilogue itself
    // it queues behind the ~96 KB of operand copies the CU has in flight: those 16 loads cost more
    // cycles than the 32 K steps.
    constexpr bool RESLDS = (SMALL || BIG == 8) && (P::NSEG == 2);
    constexpr int RROWS = STAGE / 256;                  // residual rows (256 B each) one ring stage holds
    constexpr int NRES = RESLDS ? 128 / RROWS : 0;
    static_assert(!RESLDS || ((STAGE == 16384 || STAGE == 32768) && DMA_PER_TILE * NWAVES * 1024 == STAGE), "residual tiles reuse the ring stage and its copy count");
    const int nk0 = prob.nk(0) / KTU, nkt = nk0 * P::NSEG, nstream = nkt + NRES;
    // the 256-row geometry is launched with one column tile per workgroup (walking several gains 4 % from the
    // prefetch across epilogues and loses 20 % to the registers the next tile's column vectors take)
    constexpr bool MULTI = SMALL;
    const int nrep = MULTI ? prob.col_tiles_here(G::NWX) : 1;
    const int total = nrep * nstream;
    // ring slot of stream tile i (NST is 4 or 3)
    auto slot = [&](int i) __attribute__((always_inline)) { return (NST & (NST - 1)) == 0 ? (i & (NST - 1)) : (int)((unsigned)i % (unsigned)NST); };
    P pf = prob;                 // the prefetch position runs NST tiles ahead of the consumer
    int pf_t = 0, issued = 0, plain_left = 0;
    // Source addresses of this wave's copies at the prefetch position.  They advance by one k tile (64 B) per
    // issue and are derived from the problem only where the source is not linear in k: at the first k tile of
    // a product, and where the token-shifted half of the features ends.
    // SADDR (plain A operand, unpaired columns): wave-uniform bases in SGPRs + fixed 32-bit lane offsets instead.
    constexpr bool SADDR = (P::A_MODE == A_PLAIN) && !P::PAIRED;
    const unsigned short* pa[PA];
    const unsigned short* pb[PB];
    unsigned oa[PA], ob[PB];
    const unsigned short* ba = nullptr;
    const unsigned short* bb = nullptr;
    auto derive = [&](int seg, int kt) __attribute__((always_inline)) {
        if constexpr (SADDR) {
            ba = pf.baseA16(seg, kt * KTU);
            bb = pf.baseB16(seg, kt * KTU);
        }
        auto swz = [](int row) __attribute__((always_inline)) { return ROWB == 128 ? bf_swz128(row) : bf_swz(row); };
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = (wave * PA + i) * RPI + lane / LPR;
            if constexpr (SADDR) oa[i] = pf.offA16(row, ((lane % LPR) ^ swz(row)) * 8);
            else pa[i] = pf.ptrA16(seg, kt * KTU, row, ((lane % LPR) ^ swz(row)) * 8);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int row = (wave * PB + i) * RPI + lane / LPR;
            if constexpr (SADDR) ob[i] = pf.offB16(row, ((lane % LPR) ^ swz(row)) * 8);
            else pb[i] = pf.ptrB16(seg, kt * KTU, row, ((lane % LPR) ^ swz(row)) * 8);
        }
    };
    // the copies of one k tile into ring slot `base`, then one k tile further
    auto copy_tile = [&](char* base) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const unsigned lds = __builtin_amdgcn_readfirstlane(lds_addr_of(base + (wave * PA + i) * 1024));
#if defined(MSF2_DIAG_NODMA) || defined(MSF2_DIAG_NODMA_A)
            if (base == nullptr)      // timing-only builds: address arithmetic kept, copy never issued
#endif
            {
                if constexpr (SADDR) dma16_so(ba, oa[i], lds);
                else dma16(pa[i], lds);
            }
            if constexpr (!SADDR) pa[i] += KT;
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const unsigned lds = __builtin_amdgcn_readfirstlane(lds_addr_of(base + TILE_A + (wave * PB + i) * 1024));
#if defined(MSF2_DIAG_NODMA) || defined(MSF2_DIAG_NODMA_B)
            if (base == nullptr)
#endif
            {
                if constexpr (SADDR) dma16_so(bb, ob[i], lds);
                else dma16(pb[i], lds);
            }
            if constexpr (!SADDR) pb[i] += KT;
        }
        if constexpr (SADDR) { ba += KT; bb += KT; }
    };
    auto issue = [&]() __attribute__((always_inline)) {
        char* base = sm + slot(issued) * STAGE;
        if (RESLDS && pf_t >= nkt) {
            // residual rows RROWS r .. RROWS (r + 1) - 1: one copy = 4 rows x 256 B, chunk swizzled on the source side
            const int r = pf_t - nkt;
#pragma unroll
            for (int i = 0; i < DMA_PER_TILE; ++i) {
                const int inst = wave * DMA_PER_TILE + i;
                const int row = inst * 4 + (lane >> 4);
                const int chunk = (lane & 15) ^ (row & 7);
                dma16(pf.ptrRes16(RROWS * r + row, chunk * 8), __builtin_amdgcn_readfirstlane(lds_addr_of(base + inst * 1024)));
            }
        } else {
            const int seg = (P::NSEG == 2 && pf_t >= nk0) ? 1 : 0;
            const int kt = pf_t - seg * nk0;
            if (kt == 0 || (P::A_MODE == A_SHIFT && 2 * kt == nk0)) derive(seg, kt);
            copy_tile(base);
        }
        ++issued;
        if (++pf_t == nstream) { pf_t = 0; pf.n0 += G::NWX; }
        // tiles from here on that need none of the above: same product, same source rows, the next 64 bytes
        int q = nkt;
        if (P::NSEG == 2 && pf_t <= nk0) q = nk0;
        if (P::A_MODE == A_SHIFT && 2 * pf_t <= nk0) q = nk0 >> 1;
        plain_left = (pf_t == 0) ? 0 : max(0, min(q - pf_t, total - issued));
    };
    // The common refill: one comparison, then the copies and their address increments.  Everything data dependent
    // (product switch, residual tiles, next column tile, end of the stream) is in issue(), off the straight path.
    auto refill = [&]() __attribute__((always_inline)) {
        if (__builtin_expect(plain_left > 0, 1)) {
            --plain_left;
            copy_tile(sm + slot(issued) * STAGE);
            ++issued;
            if (++pf_t == nstream) { pf_t = 0; pf.n0 += G::NWX; }     // (the last k tile of a column tile without residual tiles)
        } else if (issued < total) {
            issue();
        }
    };
    // Tile `idx` has landed once at most the copies of the tiles issued after it are outstanding.  Loads,
    // stores and copies retire in issue order, so epilogue stores issued in between only make this stricter.
    // Output stores count on the same counter.  A wave whose 64 x 64 (TM x TN tiles) region lies wholly inside the
    // output issues exactly EPI_STORES of them per epilogue (one 8-byte store per row group and MFMA tile, no lane
    // masked off, so no store instruction is branched around); tiles issued BEFORE those stores (stream index below
    // `mark`) may then be awaited with that many more operations outstanding, which lets the stores drain under the
    // K steps that follow instead of in front of them.  Any other wave keeps the strict counts (mark_stores == 0).
    constexpr int EPI_STORES = 4 * TM * (P::PAIRED ? TN / 2 : TN);
    constexpr bool STORE_AWARE = MULTI && !P::PAIRED && EPI_STORES + 3 * DMA_PER_TILE <= 63;
    int mark = 0;
    bool mark_stores = false;
    auto wait_tile = [&](int idx) __attribute__((always_inline)) {
        const int after = issued - idx - 1;
        if (STORE_AWARE && mark_stores && idx < mark) {
            if (after <= 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(EPI_STORES) : "memory");
            else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DMA_PER_TILE + EPI_STORES) : "memory");
            else if (after == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * DMA_PER_TILE + EPI_STORES) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * DMA_PER_TILE + EPI_STORES) : "memory");
            return;
        }
        if (after <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (after == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DMA_PER_TILE) : "memory");
        else if (after == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * DMA_PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * DMA_PER_TILE) : "memory");
    };
    // Per-column vectors of the epilogue are loaded BEFORE the ring starts: a load issued later queues behind
    // every copy the CU has in flight.
    constexpr int TNE = P::PAIRED ? TN / 2 : TN;
    // the two-product form reads c1 with the utterance statistics folded in (c1f); the small-grid geometry folds them itself,
    // from finalized statistics or from the ones it reduced (ProjArgs::stat_part)
    constexpr bool FOLDED = (P::NSEG == 2) && BIG != 7 && BIG != 9 && BIG != 10;
    typedef typename P::template ColVecs<TNE> CV;
    CV cv;
    prob.template load_colvecs<TN, FOLDED>(prob.n0, wc, lane, cv);
#pragma unroll
    for (int t = 0; t < NST; ++t)
        if (issued < total) issue();
    if constexpr (BIG == 7 || BIG == 9 || BIG == 10) prob.finish_utt_stats();
    prob.template fold_colvecs<TN, FOLDED>(cv);
    const int kh = lane >> 5, l31 = lane & 31;
    const int rowA = wr * (TM * 32) + l31, rowB = wc * (TN * 32) + l31;
    int cons = 0;                // stream index of the next tile to consume (all column tiles)
#ifdef MSF2_DIAG_STAMP
    unsigned long long tq[5] = {0, 0, 0, 0, 0}, tepi = 0;
    unsigned long long q0, q1, q2;
#endif
    // One k tile = two MFMA half steps (k 0..15 and 16..31).  The fragments of a half step are read from LDS
    // while the MFMAs of the half step before it run:
    //     read (t, 1) | MFMA (t, 0) | tile t + 1 landed, barrier | read (t + 1, 0) | refill slot t | MFMA (t, 1)
    // At the barrier every wave holds the second half of tile t in registers (lgkmcnt(0)), so the slot of
    // tile t itself is refilled: 3 tiles stay in flight behind the one being read.  The last k tile of a
    // column tile does not read ahead (the stream continues with residual tiles or after the epilogue).
    bf16x8_t fa0[TM] = {}, fb0[TN] = {}, fa1[TM] = {}, fb1[TN] = {};
    auto read_half = [&](int st, bf16x8_t (&fa)[TM], bf16x8_t (&fb)[TN]) {
        const char* sa = sm + slot(cons) * STAGE;
        const char* sb = sa + TILE_A;
#ifdef MSF2_DIAG_NOREADS
        if (sa != nullptr) return;      // timing-only build: no fragment reads at all
#endif
#pragma unroll
        for (int q = 0; q < TM; ++q) fa[q] = ROWB == 128 ? bf_frag128(sa, rowA + q * 32, st, kh) : bf_frag(sa, rowA + q * 32, st, kh);
#pragma unroll
        for (int q = 0; q < TN; ++q) fb[q] = ROWB == 128 ? bf_frag128(sb, rowB + q * 32, st, kh) : bf_frag(sb, rowB + q * 32, st, kh);
    };
    auto mfma_half = [&](const bf16x8_t (&fa)[TM], const bf16x8_t (&fb)[TN], f32x16 (&ac)[TM][TN]) {
#ifdef MSF2_DIAG_NOMFMA
        // timing-only build: the fragments are consumed by one cheap VALU instruction each instead of the MFMAs
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) ac[tm][0][0] += __builtin_bit_cast(float4, fa[tm]).x;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) ac[0][tn][1] += __builtin_bit_cast(float4, fb[tn]).x;
        return;
#endif
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                ac[tm][tn] = P::ACC_T ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[tn], fa[tm], ac[tm][tn], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm], fb[tn], ac[tm][tn], 0, 0, 0);
    };
    auto step = [&](f32x16 (&ac)[TM][TN], bool last) {
#ifdef MSF2_DIAG_HALFREADS
#pragma unroll
        for (int q = 0; q < TM; ++q) fa1[q] = fa0[q];       // timing-only build: half of the fragment reads never happen
#pragma unroll
        for (int q = 0; q < TN; ++q) fb1[q] = fb0[q];
#else
        if constexpr (NSUB == 4) {          // k tile of 64: two more MFMA steps ahead of the tile switch
            read_half(1, fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(fa0, fb0, ac);
            __builtin_amdgcn_sched_barrier(0);
            read_half(2, fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_half(fa1, fb1, ac);
            __builtin_amdgcn_sched_barrier(0);
        }
        read_half(NSUB - 1, fa1, fb1);
#endif
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(fa0, fb0, ac);
        __builtin_amdgcn_sched_barrier(0);
        MSF2_STAMP(q0)
        ++cons;
        // Second-half fragments are in registers (lgkmcnt) and tile `cons` has landed (vmcnt): while the stream is still
        // being refilled exactly NST - 2 younger tiles are behind it (three right after a column tile's residual tiles: then
        // this is one tile stricter than needed); once nothing is left to issue, everything is awaited.
        if (last) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if (STORE_AWARE && mark_stores && cons < mark) {
            // tile `cons` went out before the last epilogue's stores: they may still be outstanding behind it
            if (issued < total) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((NST - 2) * DMA_PER_TILE + EPI_STORES) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(EPI_STORES) : "memory");
        }
        else if (__builtin_expect(issued < total, 1)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((NST - 2) * DMA_PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        MSF2_STAMP(q1)
#ifndef MSF2_DIAG_NOBARRIER
        __builtin_amdgcn_s_barrier();
#endif
        MSF2_STAMP(q2)
        if (!last) read_half(0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        refill();                                       // into the slot of the tile this step consumed
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(fa1, fb1, ac);
        __builtin_amdgcn_sched_barrier(0);
#ifdef MSF2_DIAG_STAMP
        tq[0] += q1 - q0; tq[1] += q2 - q1; tq[4] += 1;
#endif
    };
#ifdef MSF2_DIAG_STAMP
    unsigned long long e0 = 0, e1 = 0, e2 = 0, l0 = 0;
    MSF2_STAMP(e0)
#endif
    float* stage = reinterpret_cast<float*>(sm + (STAGE_IN_RING ? 0 : NST * STAGE)) + wave * 1024;
    for (int rep = 0; rep < nrep; ++rep) {
        f32x16 acc[P::NSEG][TM][TN];
#pragma unroll
        for (int s = 0; s < P::NSEG; ++s)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[s][tm][tn][r] = 0.f;
        MSF2_STAMP(l0)
        wait_tile(cons);
        __builtin_amdgcn_s_barrier();
        read_half(0, fa0, fb0);
        if constexpr (P::NSEG == 2) {
            for (int t = 0; t < nk0; ++t) step(acc[0], false);
            for (int t = nk0; t < nkt - 1; ++t) step(acc[1], false);
            step(acc[1], true);
        } else {
            for (int t = 0; t < nkt - 1; ++t) step(acc[0], false);
            step(acc[0], true);
        }
        MSF2_STAMP(e1)
#ifdef MSF2_DIAG_STAMP
        tq[2] += e1 - l0;
#endif
#ifdef MSF2_DIAG_NOEPI
        {   // timing-only build: keep the accumulators alive with one conditional store
            float sum = 0.f;
            for (int s = 0; s < P::NSEG; ++s)
                for (int tm = 0; tm < TM; ++tm)
                    for (int tn = 0; tn < TN; ++tn)
                        for (int r = 0; r < 16; ++r) sum += acc[s][tm][tn][r];
            if (sum == 123.456f) sm[0] = (char)sum;
        }
        if constexpr (RESLDS) {
            cons += NRES;
            __builtin_amdgcn_s_barrier();
            for (int i = 0; i < NRES; ++i)
                if (issued < total) issue();
        }
#else
        // The vectors of the NEXT column tile are requested after the first tile of this epilogue and first touched
        // when it ends: by then they have long arrived, and nothing waits on them behind the copies in flight.
        CV cvn;
        const bool more = MULTI && rep + 1 < nrep;
        // (called right after the epilogue, before anything else is issued: `mark` separates the tiles that went out
        // before its stores from those that follow them)
        auto set_mark = [&]() __attribute__((always_inline)) {
            if constexpr (STORE_AWARE) {
                // exactly one store per row group and MFMA tile, by every lane: the wave's region is interior and the epilogue
                // writes one output (anything else -- an edge tile, a second output -- keeps the strict waits)
                const int per_group = (prob.a.out ? 1 : 0) + (prob.a.out16 ? 1 : 0) + (prob.a.out2 ? 1 : 0);
                mark = issued;
                mark_stores = prob.a.store_aware && per_group == 1 && prob.s0 + (wr + 1) * (TM * 32) <= prob.a.S && prob.n0 + (wc + 1) * (TN * 32) <= prob.a.N;
            }
        };
        auto next_vecs = [&]() __attribute__((always_inline)) { if (more) prob.template load_colvecs<TN, FOLDED>(prob.n0 + G::NWX, wc, lane, cvn); };
        if constexpr (RESLDS) {
            // the residual tile(s) are stream tiles cons .. cons + NRES - 1, RROWS rows of 256 B each
            wait_tile(cons + NRES - 1);
            __builtin_amdgcn_s_barrier();
            prob.template epilogue_rm<TM, TN, true>(acc, wr, wc, lane, stage, cv, sm + slot(cons + (wr * TM * 32) / RROWS) * STAGE + ((wr * TM * 32) % RROWS) * 256, next_vecs);
            set_mark();
            cons += NRES;
#ifdef MSF2_DIAG_DRAIN_AFTER_EPI
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // timing-only build: every store acknowledged before the stream goes on
#endif
            // every wave is done with the residual stages: refill them
            __builtin_amdgcn_s_barrier();
            for (int i = 0; i < NRES; ++i)
                if (issued < total) issue();
        } else {
            prob.template epilogue_rm<TM, TN, false>(acc, wr, wc, lane, stage, cv, nullptr, next_vecs);
            set_mark();
        }
        if (more) {
            prob.template fold_colvecs<TN, FOLDED>(cvn);
            cv = cvn;
        }
        MSF2_STAMP(e2)
#ifdef MSF2_DIAG_STAMP
        tepi += e2 - e1;
#endif
#endif
        prob.n0 += G::NWX;
    }
#ifdef MSF2_DIAG_STAMP
    MSF2_STAMP(e2)
    if (lane == 0 && P::NSEG == 2) {     // per wave: vmcnt wait, barrier wait, whole main loops, -, steps, epilogue, total
        const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (L < 2048 && wave < 4) {
            unsigned long long* o = g_stamp + (L * 4 + wave) * 4;
            o[0] = tq[0]; o[1] = tq[1]; o[2] = tq[2]; o[3] = tq[3];
            g_stamp2[(L * 4 + wave) * 2 + 0] = tq[4] | (tepi << 20);
            g_stamp2[(L * 4 + wave) * 2 + 1] = e2 - e0;
        }
    }
#endif
}

}  // namespace msf2
