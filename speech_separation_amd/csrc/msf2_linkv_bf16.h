// msf2_linkv_bf16.h -- linear-attention key/value summary of the bf16 mode.
//
//   part[b, sp, d, e] = sum over the frames s of time chunk sp of  lin_k[b, s, d] * vu[b, s, e]        d < 128, e < N
//
// (reference: FLASH_ShareA_FFConvM.cal_attention, models/mossformer2/mossformer2_block.py:318-322 --
//  lin_kv = einsum('b g n d, b g n e -> b d e', lin_k, v) / n, and the same with u; vu = [v | u] side by side; the sum over
//  chunks and the division by the unpadded length are kv_finalize_kernel's.)
//
// Both operands have the contraction index (time) as their ROW index, the layout the MFMA wants transposed.  The tiles
// therefore go global -> LDS as they lie in memory (32 frames x 128 columns, 256-byte rows, global_load_lds_dwordx4 with
// the bank swizzle applied on the source side) and BOTH fragments come out of LDS through the transposing read
// (ds_read_b64_tr_b16): no register staging, no unpack / transpose / repack arithmetic.  3-slot ring of 24 KB stages,
// counted s_waitcnt vmcnt + raw s_barrier, two workgroups per CU.  The kernel is paced by the L2 -> LDS copies, so a
// workgroup covers 256 value columns per lin_k tile (1.5 bytes copied per 128 x 128 x 32 block of products instead of 2).
// grid (N / 256, nsplit, B), block 512 (8 waves, 2 x 4, each 64 d x 64 e); N a multiple of 256.
#pragma once
#include "msf2_attn_fused.h"

namespace msf2 {

struct LinKV16Args {
    const unsigned short* lk;      // [B*S][128] bf16
    const unsigned short* hu;      // [B*S][ldhu] bf16; the first N columns are v | u
    int ldhu;
    float* part;                   // [B][nsplit][128][N]
    int S, N, chunk;               // chunk: frames per split, a multiple of 32
};

constexpr int LKV_KT = 32;                       // frames per stage
constexpr int LKV_TILE = LKV_KT * 256;           // one operand tile: 32 rows x 128 bf16
constexpr int LKV_STAGE = 3 * LKV_TILE;          // lin_k tile, then two value tiles (columns 0..127 and 128..255 of the workgroup)
constexpr int LKV_NST = 3;

__global__ __launch_bounds__(512, 2) void linkv_bf16_kernel(const LinKV16Args a) {
    __shared__ __attribute__((aligned(1024))) char sm[LKV_NST * LKV_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // the column tiles of one (utterance, chunk) run on ONE XCD: its L2 then fetches the lin_k rows once and sees the value
    // rows as whole contiguous rows
    const unsigned t = xcd_remap();
    const int ct = t % gridDim.x, sp = (t / gridDim.x) % gridDim.y, b = t / (gridDim.x * gridDim.y);
    const int k0 = sp * a.chunk, k1 = min(k0 + a.chunk, a.S), n0 = ct * 256;
    const int nk = (k1 - k0 + LKV_KT - 1) / LKV_KT;
    if (nk <= 0) return;      // (uniform; cannot happen with the host's chunking)

    // A stage is 24 copies of 1 KB (4 rows x 256 B): copies 0..7 the lin_k tile, 8..15 and 16..23 the two value tiles.  Wave w
    // issues copies 3 w .. 3 w + 2.  Lane l of copy j lands at row 4 (j & 7) + (l >> 4), chunk position l & 15, and fetches
    // the chunk that af_off() places there.  Rows past the chunk's end are read from its last row (finite values) and zeroed
    // in LDS on the lin_k side before the last tile is used.
    const unsigned short* lkb = a.lk + (long)b * a.S * MSF2_QK_DIM;
    const unsigned short* hub = a.hu + (long)b * a.S * a.ldhu + n0;
    int prow[3];
    unsigned pch[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = 3 * wave + i;
        const int row = 4 * (j & 7) + (lane >> 4);
        prow[i] = row;
        pch[i] = (unsigned)((lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3))) * 16u + (j >= 16 ? 256u : 0u);
    }
    auto copy_tile = [&](int kt) __attribute__((always_inline)) {
        char* base = sm + (kt % LKV_NST) * LKV_STAGE;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = 3 * wave + i;                        // (wave-uniform)
            const int s = min(k0 + kt * LKV_KT + prow[i], k1 - 1);
            const unsigned lds = __builtin_amdgcn_readfirstlane(lds_addr_of(base + j * 1024));
#ifdef MSF2_DIAG_LKV_NOCOPY
            if (base == nullptr)
#endif
            if (j < 8) dma16_so(lkb, (unsigned)s * (MSF2_QK_DIM * 2u) + pch[i], lds);
            else dma16_so(hub, (unsigned)s * ((unsigned)a.ldhu * 2u) + pch[i], lds);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    int issued = 0;
    for (; issued < LKV_NST - 1 && issued < nk; ++issued) copy_tile(issued);
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most the copies of the tiles issued after it are outstanding (3 per tile and wave)
        const int after = issued - kt - 1;
        if (after <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_s_barrier();       // every wave's copies of tile kt are in; every wave is done with tile kt - 1
        if (issued < nk) { copy_tile(issued); ++issued; }     // into the slot tile kt - 1 occupied
        char* st = sm + (kt % LKV_NST) * LKV_STAGE;
        const int nv = k1 - k0 - kt * LKV_KT;                  // valid frames of this tile
        if (nv < LKV_KT) {                                     // (uniform, last tile only) zero the lin_k rows past the end
            const uint4 z = {0u, 0u, 0u, 0u};
            const int row = tid >> 4;
            if (row >= nv) *reinterpret_cast<uint4*>(st + 256 * row + 16 * (tid & 15)) = z;
            __syncthreads();
        }
#ifdef MSF2_DIAG_LKV_NOMATH
        if (a.S >= 0) continue;      // timing-only build: copies and barriers without the fragment reads and MFMAs
#endif
#pragma unroll
        for (int s16 = 0; s16 < 2; ++s16) {
            bf16x8_t af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) af[t] = af_bfrag(st, wr * 64 + 32 * t, s16, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) bf[t] = af_bfrag(st + LKV_TILE * (1 + (wc >> 1)), (wc & 1) * 64 + 32 * t, s16, lane);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm], bf[tn], acc[tm][tn], 0, 0, 0);
        }
    }
    float* out = a.part + ((long)b * gridDim.y + sp) * MSF2_QK_DIM * a.N;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int d = MSF2_ACC_ROW(wr, 2, tm, reg, lane);
                const int e = n0 + MSF2_ACC_COL(wc, 2, tn, lane);      // wave column wc covers 64 wc .. 64 wc + 63
                out[(long)d * a.N + e] = acc[tm][tn][reg];
            }
}

}  // namespace msf2
