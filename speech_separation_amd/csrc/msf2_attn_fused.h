// Fused FLASH attention core for bf16 mode: scores, both attention products and the gating in ONE kernel, with
// the scores kept on chip (FLASH_ShareA_FFConvM.cal_attention, reference: models/mossformer2/mossformer2_block.py:283-325,
// and the gating of :258).
//
// One workgroup (8 waves) owns 128 query frames -- half of a 256-frame group -- of one utterance:
//   phase 1  P = relu(quad_q quad_k^T)^2 for its 128 x 256 (query, key) pairs: operand fragments straight from global
//            memory (96 KB per workgroup, once), the result rounded to bf16 into LDS (64 KB) in the layout the MFMA
//            A-operand reads;  lin_q of the same 128 frames into LDS as well (32 KB);
//   phase 2  for each of the E / 64 column slabs: [P | lin_q] x [v, u rows of the group ; KV] (K = 256 + 128) into one
//            accumulator set, then o = (att_u * v) * sigmoid(att_v * u) and the partial sums of o^2 for the ScaleNorm
//            that follows.
// The A operand ([P | lin_q]) never moves again: only the value rows stream (register staged -- three steps of two k
// tiles in flight --, stored row-major as they arrive and transposed by the LDS read, ds_read_b64_tr_b16).  Against the two-kernel form (scores to HBM, then one
// workgroup per (half group, column slab) that reloads its score rows) this halves the bytes a CU pulls from L2 per
// MFMA -- the resource these contractions are bound by -- and removes the score tensor's HBM round trip and a launch.
//
// LDS (all 160 KiB of the CU): P 64 KB | lin_q 32 KB | value tiles 2 x 2 x 8 KB | epilogue tiles 8 x 4 KB.
#pragma once
#include "msf2_gemm_bf16.h"

namespace msf2 {

struct AttnFusedArgs {
    const unsigned short* qq;   // quad_q [B*S][128] bf16, pre-scaled by 1 / group size
    const unsigned short* qk;   // quad_k [B*S][128] bf16
    const unsigned short* lq;   // lin_q  [B*S][128] bf16
    const unsigned short* hu;   // hidden tensor v | u | ... as bf16, row pitch ldhu
    int ldhu;
    const float* kv;            // [B][128][2E] linear-attention summary (already divided by S)
    unsigned short* o16;        // [B*S][E] bf16
    float* sq_part;             // optional [B*S][E / 32]: sums of o^2 over 32-column slabs
    int S, G, E;
};

constexpr int AF_P = 0, AF_LQ = 65536, AF_BT = 98304, AF_STG = 131072, AF_LDS = 163840;
// Value-tile image: 256-byte rows (one k row, 128 columns), 16-byte chunk ch of row r at chunk position
// ch ^ (((r & 3) << 2) | ((r >> 2) & 3)): the 8-byte row stores and the transposing reads (4 rows x 16 columns per
// 16-lane group) are then both free of bank conflicts.
__device__ __forceinline__ int af_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
typedef short af_s16x4 __attribute__((ext_vector_type(4)));
// B fragment of v_mfma_f32_32x32x16_bf16 for columns nb .. nb + 31 at MFMA step st: lane (column l & 31, half l >> 5)
// needs k = 16 st + 8 (l >> 5) + 0..7 of its column = two transposed reads of 4 k rows each.  In a 16-lane group,
// lane 4 q + p supplies the address of row q, columns 4 p .. 4 p + 3 of the group's 4 x 16 block and receives column
// (lane & 15) of the 4 rows.
__device__ __forceinline__ bf16x8_t af_bfrag(const char* tile, int nb, int st, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int c0 = (nb + 16 * (g & 1)) >> 3;
    const int r0 = 16 * st + 8 * (g >> 1);
    typedef __attribute__((address_space(3))) af_s16x4* lp;
    const af_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(tile + af_off(r0 + q, c0 + (p >> 1)) + 8 * (p & 1)));
    const af_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(tile + af_off(r0 + 4 + q, c0 + (p >> 1)) + 8 * (p & 1)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}
constexpr int AF_STEPS = 6;     // steps of two k tiles per column slab: 4 over the group's value rows, 2 over KV

__global__ __launch_bounds__(512, 1) void attn_fused_bf16_kernel(const AttnFusedArgs a) {
    __shared__ __attribute__((aligned(1024))) char sm[AF_LDS];
    const unsigned t = xcd_remap();                  // both halves of a group on one XCD, one after the other
    const int half = t & 1, z = t >> 1;
    const int b = z / a.G, g = z % a.G;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;         // 4 x 2 waves: 32 query frames x (32 v + 32 u) columns each
    const int kh = lane >> 5, l31 = lane & 31;
    const long rowbase = (long)b * a.S;
    const int q0 = g * MSF2_GROUP + half * 128;      // first query frame of this workgroup
    const int k0 = g * MSF2_GROUP;                   // first key / value frame of the group

    // ---- phase 1: scores into LDS ------------------------------------------------------------------------------
    {
        const int qrow = min(q0 + 32 * wr + l31, a.S - 1);
        const unsigned short* qp = a.qq + (rowbase + qrow) * MSF2_QK_DIM + 8 * kh;
        bf16x8_t qf[8];
#pragma unroll
        for (int st = 0; st < 8; ++st) qf[st] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(qp + 16 * st));
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f32x16 acc[2];
            bf16x8_t kf[2][8];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int krow = min(k0 + 128 * wc + 64 * p + 32 * tn + l31, a.S - 1);
                const unsigned short* kp = a.qk + (rowbase + krow) * MSF2_QK_DIM + 8 * kh;
#pragma unroll
                for (int st = 0; st < 8; ++st) kf[tn][st] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(kp + 16 * st));
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
            }
#pragma unroll
            for (int st = 0; st < 8; ++st)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[tn][st], qf[st], acc[tn], 0, 0, 0);
            // lane: query frame l31 of the wave's 32, keys 8 q + 4 kh + 0..3 of the 32-key tile = k tile (128 wc + 64 p + 32 tn) / 32
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int kt = 4 * wc + 2 * p + tn, row = 32 * wr + l31;
                char* dst = sm + AF_P + kt * BF_TILE + row * 64 + kh * 8;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float r = fmaxf(acc[tn][4 * q + e], 0.f); x[e] = r * r; }
                    uint2 v;
                    v.x = pk_bf16_2(x[0], x[1]);
                    v.y = pk_bf16_2(x[2], x[3]);
                    *reinterpret_cast<uint2*>(dst + ((q ^ bf_swz(row)) << 4)) = v;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {              // lin_q rows of the 128 query frames: 4 k tiles of [128][32]
            const int e = tid + 512 * i, row = e >> 4, ch = e & 15;
            const int s = min(q0 + row, a.S - 1);
            const uint4 v = *reinterpret_cast<const uint4*>(a.lq + (rowbase + s) * MSF2_QK_DIM + ch * 8);
            *reinterpret_cast<uint4*>(sm + AF_LQ + (ch >> 2) * BF_TILE + row * 64 + (((ch & 3) ^ bf_swz(row)) << 4)) = v;
        }
    }

    // ---- phase 2: column slabs ------------------------------------------------------------------------------------
    const int nslab = a.E >> 6;
    const int hf = __builtin_amdgcn_readfirstlane(tid >> 8);       // which of the step's two k tiles this half of the workgroup stages
    const int c8 = tid & 15, kr = (tid >> 4) & 15;                 // 8 columns (16 bytes) x k rows kr and kr + 16 per thread
    const int ecol = (c8 < 8) ? 8 * c8 : a.E + 8 * c8 - 64;        // v columns | u columns of the slab, relative to 64 * slab
    // step tt of slab sl: the two 8-column row pieces this thread stages.  tt is a literal at every call, so the source
    // (value rows for the first 4 steps, summary rows for the last 2) is fixed at compile time and the loads sit in
    // straight-line code: the compiler can then count them (s_waitcnt vmcnt(N)) instead of draining them.  Nothing here
    // touches the loaded values (no conversion, no masking): that waits until put(), two steps later.
    auto fetch = [&](int sl, int tt, float4 (&R)[4]) __attribute__((always_inline)) {
#ifdef MSF2_DIAG_AF_NOFETCH
        R[0] = R[1] = R[2] = R[3] = make_float4((float)sl, (float)tt, 0.f, 0.f);      // timing-only build: no value loads
        return;
#endif
        const int kk = 2 * tt + hf;
        const int e = 64 * sl + ecol;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (tt < 4) {                   // value rows of the group: 8 bf16 = 16 bytes
                const int s = k0 + 32 * kk + kr + 16 * i;
                R[i] = __builtin_bit_cast(float4, *reinterpret_cast<const uint4*>(a.hu + (rowbase + min(s, a.S - 1)) * a.ldhu + e));
            } else {                        // rows of the key/value summary (fp32): 8 floats
                const float* src = a.kv + ((long)b * MSF2_QK_DIM + 32 * (kk - 8) + kr + 16 * i) * (2 * a.E) + e;
                R[2 * i] = *reinterpret_cast<const float4*>(src);
                R[2 * i + 1] = *reinterpret_cast<const float4*>(src + 4);
            }
        }
    };
    // step tt's pieces, rows past the utterance end zeroed / summary rows rounded to bf16, stored as they were loaded:
    // the tile image is [32 k rows][128 columns] bf16 (256-byte rows, what a coalesced load delivers), and the MFMA
    // operand -- 8 consecutive k of one column per lane -- is gathered from it by the transposing LDS read.
    auto put = [&](const float4 (&R)[4], int tt, int buf) __attribute__((always_inline)) {
#ifdef MSF2_DIAG_AF_NOPUT
        if (R[0].x == 123.456f) sm[0] = 1;      // timing-only build: the values are awaited but never written to LDS
        return;
#endif
        char* tile = sm + AF_BT + buf * (2 * BF_TILE) + hf * BF_TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint4 w;
            if (tt < 4) {
                const int s = k0 + 32 * (2 * tt + hf) + kr + 16 * i;
                w = (s < a.S) ? __builtin_bit_cast(uint4, R[i]) : make_uint4(0u, 0u, 0u, 0u);
            } else
                w = make_uint4(pk_bf16_2(R[2 * i].x, R[2 * i].y), pk_bf16_2(R[2 * i].z, R[2 * i].w),
                               pk_bf16_2(R[2 * i + 1].x, R[2 * i + 1].y), pk_bf16_2(R[2 * i + 1].z, R[2 * i + 1].w));
            *reinterpret_cast<uint4*>(tile + af_off(kr + 16 * i, c8)) = w;
        }
    };
    f32x16 acc[2];       // att_v | att_u of the wave's 32 x 32 output block
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;
    };
    auto compute = [&](int tt, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int kk = 2 * tt + h2;
            const char* A = (kk < 8) ? sm + AF_P + kk * BF_TILE : sm + AF_LQ + (kk - 8) * BF_TILE;
            const char* Bt = sm + AF_BT + buf * (2 * BF_TILE) + h2 * BF_TILE;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const bf16x8_t af = bf_frag(A, 32 * wr + l31, st, kh);
                const bf16x8_t b0 = af_bfrag(Bt, 32 * wc, st, lane);
                const bf16x8_t b1 = af_bfrag(Bt, 64 + 32 * wc, st, lane);
#ifdef MSF2_DIAG_AF_NOMFMA
                acc[0][0] += __builtin_bit_cast(float4, af).x + __builtin_bit_cast(float4, b0).x;   // timing-only build
                acc[1][0] += __builtin_bit_cast(float4, b1).x;
                continue;
#endif
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, af, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, af, acc[1], 0, 0, 0);
            }
        }
    };
    // epilogue of one slab: accumulators through the wave's LDS tile into row-major registers (lane: rows 8 i + (l >> 3),
    // columns 4 (l & 7) + 0..3), gating with the v, u values of the same positions, bf16 store, partial sums of squares
    const int rr = lane >> 3, rc = lane & 7;
    float* stage = reinterpret_cast<float*>(sm + AF_STG) + wave * 1024;
    unsigned mrow[4];
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = q0 + 32 * wr + 8 * i + rr;
        ok[i] = s < a.S;
        mrow[i] = (unsigned)rowbase + min(s, a.S - 1);
    }
    uint2 vv[4], uu[4];
    auto load_vu = [&](int sl) __attribute__((always_inline)) {
        const unsigned j = 64 * sl + 32 * wc + 4 * rc;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vv[i] = *reinterpret_cast<const uint2*>(a.hu + (long)mrow[i] * a.ldhu + j);
            uu[i] = *reinterpret_cast<const uint2*>(a.hu + (long)mrow[i] * a.ldhu + a.E + j);
        }
    };
    auto transpose = [&](const f32x16& tl, float4 (&o)[4]) __attribute__((always_inline)) {
        const int ar = lane & 31;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(stage + (ar * 8 + ((2 * q + kh) ^ (ar & 7))) * 4) = make_float4(tl[4 * q], tl[4 * q + 1], tl[4 * q + 2], tl[4 * q + 3]);
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
    auto unpack = [](uint2 h) __attribute__((always_inline)) {
        return make_float4(__builtin_bit_cast(float, h.x << 16), __builtin_bit_cast(float, h.x & 0xffff0000u),
                           __builtin_bit_cast(float, h.y << 16), __builtin_bit_cast(float, h.y & 0xffff0000u));
    };
    // The epilogue of a slab is split: the two transposes right after its last step (the accumulators are then free for
    // the next slab), the gating arithmetic and the stores of its 4 row groups between the MFMAs of the next slab's
    // first two steps (two row groups each; one per step over four steps measured the same and spilled), where they overlap with the matrix pipe instead of stalling it.
    float4 av[4], au[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = au[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto epi_transpose = [&]() __attribute__((always_inline)) {
#ifdef MSF2_DIAG_AF_NOEPI
        if (acc[0][0] + acc[1][0] == 123.456f) sm[1] = 1;    // timing-only build
        return;
#endif
        transpose(acc[0], av);
        transpose(acc[1], au);
    };
    auto epi_rows = [&](int sl, int i0, bool valid) __attribute__((always_inline)) {
#ifdef MSF2_DIAG_AF_NOEPI
        return;
#endif
        const unsigned j = 64 * sl + 32 * wc + 4 * rc;
        const unsigned np = a.E >> 5;
#pragma unroll
        for (int i = i0; i < i0 + 2; ++i) {
            const float4 v = unpack(vv[i]), u = unpack(uu[i]);
            float4 r;
            r.x = (au[i].x * v.x) * sigmoid_f(av[i].x * u.x);
            r.y = (au[i].y * v.y) * sigmoid_f(av[i].y * u.y);
            r.z = (au[i].z * v.z) * sigmoid_f(av[i].z * u.z);
            r.w = (au[i].w * v.w) * sigmoid_f(av[i].w * u.w);
            float ssq = r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
            ssq = sum8_dpp(ssq);
            if (ok[i] && valid) {
                uint2 h;
                h.x = pk_bf16_2(r.x, r.y);
                h.y = pk_bf16_2(r.z, r.w);
                *reinterpret_cast<uint2*>(a.o16 + (long)mrow[i] * a.E + j) = h;
                if (a.sq_part && rc == 0) a.sq_part[(long)mrow[i] * np + 2 * sl + wc] = ssq;
            }
        }
    };

    // Three steps of values are in flight in registers (R0, R1, R2 in rotation; a fourth gained nothing and spilled);
    // LDS holds the step being contracted and the next one.  Step tt of a slab: request step tt + 3 (of the next slab
    // for tt >= 3), contract step tt from buffer tt & 1, land step tt + 1 in the other buffer.  Six steps per slab = two
    // rotations, so the roles of R0..R2 are the same in every slab and the slab body is straight-line code.  Requests
    // past the last slab repeat the last slab's (never contracted).
    // The v, u values a slab's gating needs are requested a whole slab ahead: memory operations complete in order, so
    // requested just before use they would make the gating wait for every value step issued before them.
    float4 R0[4], R1[4], R2[4];
    load_vu(0);
    fetch(0, 0, R0);
    fetch(0, 1, R1);
    fetch(0, 2, R2);
    zero_acc();
    put(R0, 0, 0);
    __syncthreads();         // scores, lin_q and the first value tiles are in LDS
#define MSF2_AF_STEP(tt, RA, RB, BETWEEN)                                                   \
    {                                                                                       \
        fetch((tt) + 3 < AF_STEPS ? sl : nxt, ((tt) + 3) % AF_STEPS, RA);                   \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        compute((tt), (tt) & 1);                                                            \
        BETWEEN                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        put(RB, ((tt) + 1) % AF_STEPS, ((tt) + 1) & 1);                                      \
        __syncthreads();                                                                    \
    }
    for (int sl = 0; sl < nslab; ++sl) {
        const int nxt = min(sl + 1, nslab - 1), prev = max(sl - 1, 0);
        const bool have_prev = sl > 0;
        MSF2_AF_STEP(0, R0, R1, epi_rows(prev, 0, have_prev);)
        MSF2_AF_STEP(1, R1, R2, epi_rows(prev, 2, have_prev); load_vu(sl);)
        MSF2_AF_STEP(2, R2, R0, )
        MSF2_AF_STEP(3, R0, R1, )
        MSF2_AF_STEP(4, R1, R2, )
        MSF2_AF_STEP(5, R2, R0, )
        epi_transpose();
        zero_acc();
    }
    epi_rows(nslab - 1, 0, true);
    epi_rows(nslab - 1, 2, true);
#undef MSF2_AF_STEP
}

}  // namespace msf2
