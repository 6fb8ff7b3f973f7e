// fp32 contraction kernels on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One kernel template, gemm_kernel<P>, parameterised by a "problem" class P
// that knows where the operand tiles live (fetchA/fetchB) and what to do with
// the accumulators (epilogue).  Block tile 128x128, K step 32, 256 threads =
// 4 waves; each wave owns a 64x64 (2x2 waves) or 32x128 (4x1 waves) sub-tile
// as 32x32 MFMA accumulators.  Operand tiles are staged global -> registers ->
// LDS (double buffered, one barrier per K step) in a k-major image
// [k][row] so that the MFMA operands (one fp32 per lane, lane = row, lane
// half = k parity) are conflict-free ds_read_b32.
#pragma once
#include "msf2_common.h"

namespace msf2 {

constexpr int BM = 128, BN = 128, BK = 32;
enum { ORI_KC = 0, ORI_RC = 1 };   // operand tile is k-contiguous / row-contiguous in global memory
// row pitch of the k-major LDS image: 130 makes the transposing scalar stores
// of a k-contiguous source 2-way (free); 132 keeps float4 row stores 16-B aligned.
template <int ORI> struct LdsPitch { static constexpr int v = (ORI == ORI_KC) ? 130 : 132; };

template <int ORI>
__device__ __forceinline__ void tile_coords(int tid, int i, int& r, int& k) {
    const int e = tid + 256 * i;            // 1024 float4 per 128x32 tile, 4 per thread
    if (ORI == ORI_KC) { r = e >> 3; k = (e & 7) * 4; }     // float4 along k
    else               { k = e >> 5; r = (e & 31) * 4; }    // float4 along rows
}

template <int ORI>
__device__ __forceinline__ void lds_put(float* s, int tid, int i, const float4& v) {
    int r, k;
    tile_coords<ORI>(tid, i, r, k);
    constexpr int LD = LdsPitch<ORI>::v;
    if (ORI == ORI_KC) {
        s[(k + 0) * LD + r] = v.x;
        s[(k + 1) * LD + r] = v.y;
        s[(k + 2) * LD + r] = v.z;
        s[(k + 3) * LD + r] = v.w;
    } else {
        *reinterpret_cast<float4*>(&s[k * LD + r]) = v;
    }
}

template <class P, int SEG, int TM, int TN>
__device__ __forceinline__ void run_segment(const P& prob, float* smA, float* smB,
                                            f32x16 (&acc)[TM][TN], int tid, int lane, int wr, int wc) {
    constexpr int LDA = LdsPitch<P::A_ORI>::v, LDB = LdsPitch<P::B_ORI>::v;
    constexpr int SZA = BK * LDA, SZB = BK * LDB;
    const int nk = prob.nk(SEG);
    float4 ra[4], rb[4];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r, k;
            tile_coords<P::A_ORI>(tid, i, r, k);
            ra[i] = prob.fetchA(SEG, r, kt * BK + k);
            tile_coords<P::B_ORI>(tid, i, r, k);
            rb[i] = prob.fetchB(SEG, r, kt * BK + k);
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lds_put<P::A_ORI>(smA + buf * SZA, tid, i, ra[i]);
            lds_put<P::B_ORI>(smB + buf * SZB, tid, i, rb[i]);
        }
    };
    fetch(0);
    put(0);
    __syncthreads();
    const int kh = lane >> 5, l31 = lane & 31;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) fetch(kt + 1);
        const float* a = smA + cur * SZA + kh * LDA + wr * (TM * 32) + l31;
        const float* b = smB + cur * SZB + kh * LDB + wc * (TN * 32) + l31;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float af[TM], bf[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) af[t] = a[kk * LDA + t * 32];
#pragma unroll
            for (int t = 0; t < TN; ++t) bf[t] = b[kk * LDB + t * 32];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm], bf[tn], acc[tm][tn], 0, 0, 0);
        }
        if (kt + 1 < nk) put(cur ^ 1);
        __syncthreads();
    }
}

template <class P>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const typename P::Args args) {
    constexpr int LDA = LdsPitch<P::A_ORI>::v, LDB = LdsPitch<P::B_ORI>::v;
    __shared__ __attribute__((aligned(16))) float smA[2 * BK * LDA];
    __shared__ __attribute__((aligned(16))) float smB[2 * BK * LDB];
    P prob;
    prob.init(args);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / P::WC, wc = wave % P::WC;
    constexpr int TM = BM / (32 * P::WR), TN = BN / (32 * P::WC);
    f32x16 acc[P::NSEG][TM][TN];
#pragma unroll
    for (int s = 0; s < P::NSEG; ++s)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][tm][tn][r] = 0.f;
    run_segment<P, 0, TM, TN>(prob, smA, smB, acc[0], tid, lane, wr, wc);
    if constexpr (P::NSEG == 2) run_segment<P, 1, TM, TN>(prob, smA, smB, acc[1], tid, lane, wr, wc);
    prob.template epilogue<TM, TN>(acc, wr, wc, lane);
}

// tile-local coordinates of accumulator register `reg` of MFMA tile (tm, tn)
#define MSF2_ACC_ROW(wr, TM, tm, reg, lane) ((wr) * (TM) * 32 + (tm) * 32 + ((reg) & 3) + 8 * ((reg) >> 2) + 4 * ((lane) >> 5))
#define MSF2_ACC_COL(wc, TN, tn, lane) ((wc) * (TN) * 32 + (tn) * 32 + ((lane) & 31))

// ---------------------------------------------------------------------------
// Projections: out[b, s, n] = epilogue( sum_k A'[b, s, k] * W[n, k] )
// grid = (n tiles, ceil(S / 128), B)
// ---------------------------------------------------------------------------
enum { A_PLAIN = 0, A_SHIFT = 1, A_MUL = 2 };
enum {
    EPI_RS_BIAS_ACT = 0,  // act(rowscale[m] * acc + bias[n]); act 0 none, 1 SiLU, 2 sigmoid
    EPI_NORM = 1,         // rstd * (acc - mean * c2[n]) + c1[n] (+ pe[s, n]); stats per row or per utterance
    EPI_BIAS_RES = 2,     // res[m, n] + acc + bias[n]
    EPI_RELU_MUL = 3,     // relu(acc) * res[m, n]
    EPI_TANH_SIG = 4,     // paired columns: tanh(accA + bias[n]) * sigmoid(accB + bias2[n])
    EPI_TCN = 5,          // two accumulators: res + sigmoid(acc0 + bias) * (rstd_b * (acc1 - mean_b * c2) + c1)
};

struct ProjArgs {
    const float* A;  const float* A2;  int lda;
    const float* W;  const float* W2;  int K;  int N;
    int S;
    float* out;  float* out2;  int ldo;
    const float* bias;  const float* bias2;  const float* c1;  const float* c2;
    const float* rowstat;   // [B*S] inverse norms, or [B*S][2] (mean, rstd)
    const float* bstat;     // [B][2] (mean, rstd)
    const float* res;  int ldr;
    const float* pe;        // [S][N]
    int act;
};

template <int AMODE, int EPI>
struct ProjNT {
    typedef ProjArgs Args;
    static constexpr bool PAIRED = (EPI == EPI_TANH_SIG);
    static constexpr int WR = PAIRED ? 4 : 2, WC = PAIRED ? 1 : 2;
    static constexpr int NSEG = (EPI == EPI_TCN) ? 2 : 1;
    static constexpr int A_ORI = ORI_KC, B_ORI = ORI_KC;
    Args a;
    int b, s0, n0;
    __device__ void init(const Args& args) {
        a = args;
        b = blockIdx.z;
        s0 = blockIdx.y * BM;
        n0 = blockIdx.x * (PAIRED ? BN / 2 : BN);
    }
    __device__ int nk(int) const { return a.K / BK; }
    __device__ float4 fetchA(int seg, int r, int k) const {
        const int s = s0 + r;
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s >= a.S) return z;
        long row = (long)b * a.S + s;
        const float* base = (NSEG == 2 && seg == 1) ? a.A2 : a.A;
        if (AMODE == A_SHIFT && k < (a.K >> 1)) {
            if (s == 0) return z;
            row -= 1;
        }
        float4 v = *reinterpret_cast<const float4*>(base + row * a.lda + k);
        if (AMODE == A_MUL) {
            const float4 g = *reinterpret_cast<const float4*>(a.A2 + row * a.lda + k);
            v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
        }
        return v;
    }
    __device__ float4 fetchB(int seg, int c, int k) const {
        const float* w = a.W;
        int n = n0 + c;
        if (PAIRED && c >= BN / 2) { n -= BN / 2; w = a.W2; }
        if (NSEG == 2 && seg == 1) w = a.W2;
        if (n >= a.N) return make_float4(0.f, 0.f, 0.f, 0.f);
        return *reinterpret_cast<const float4*>(w + (long)n * a.K + k);
    }
    template <int TM, int TN>
    __device__ void epilogue(f32x16 (&acc)[NSEG][TM][TN], int wr, int wc, int lane) const {
        float mean_b = 0.f, rstd_b = 1.f;
        if (a.bstat) { mean_b = a.bstat[2 * b]; rstd_b = a.bstat[2 * b + 1]; }
        constexpr int TNE = PAIRED ? TN / 2 : TN;
#pragma unroll
        for (int tn = 0; tn < TNE; ++tn) {
            const int cl = MSF2_ACC_COL(wc, TN, tn, lane);
            const int n = n0 + cl;
            if (n >= a.N) continue;
            const float bias = a.bias ? a.bias[n] : 0.f;
            const float bias2 = (EPI == EPI_TANH_SIG) ? a.bias2[n] : 0.f;
            const float c1 = (EPI == EPI_NORM || EPI == EPI_TCN) ? a.c1[n] : 0.f;
            const float c2 = (EPI == EPI_NORM || EPI == EPI_TCN) ? a.c2[n] : 0.f;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int s = s0 + MSF2_ACC_ROW(wr, TM, tm, reg, lane);
                    if (s >= a.S) continue;
                    const long m = (long)b * a.S + s;
                    const float x = acc[0][tm][tn][reg];
                    float v;
                    if constexpr (EPI == EPI_RS_BIAS_ACT) {
                        const float rs = a.rowstat ? a.rowstat[m] : 1.f;
                        v = rs * x + bias;
                        if (a.act == 1) v = silu_f(v);
                        else if (a.act == 2) v = sigmoid_f(v);
                    } else if constexpr (EPI == EPI_NORM) {
                        float mu = mean_b, rs = rstd_b;
                        if (a.rowstat) { mu = a.rowstat[2 * m]; rs = a.rowstat[2 * m + 1]; }
                        v = rs * (x - mu * c2) + c1;
                        if (a.pe) v += a.pe[(long)s * a.N + n];
                    } else if constexpr (EPI == EPI_BIAS_RES) {
                        v = a.res[m * a.ldr + n] + x + bias;
                    } else if constexpr (EPI == EPI_RELU_MUL) {
                        v = fmaxf(x, 0.f);
                        if (a.out2) a.out2[m * a.ldo + n] = v;
                        v *= a.res[m * a.ldr + n];
                    } else if constexpr (EPI == EPI_TANH_SIG) {
                        const float y = acc[0][tm][tn + TN / 2][reg];
                        v = tanhf(x + bias) * sigmoid_f(y + bias2);
                    } else {  // EPI_TCN
                        const float y = acc[NSEG - 1][tm][tn][reg];
                        v = a.res[m * a.ldr + n] + sigmoid_f(x + bias) * (rstd_b * (y - mean_b * c2) + c1);
                    }
                    a.out[m * a.ldo + n] = v;
                    if (EPI == EPI_NORM && a.out2) a.out2[m * a.ldo + n] = v;
                }
            }
        }
    }
};

// ---------------------------------------------------------------------------
// Quadratic attention scores of one 256-frame group:
//   P[z, i, j] = relu(q_i . k_j)^2,  q pre-scaled by 1/group.  grid = (2, 2, B*G)
// ---------------------------------------------------------------------------
struct SimArgs {
    const float* q;  const float* k;   // [B*S][128]
    float* P;                          // [B*G][256][256]
    int S, G;
};
struct AttnSim {
    typedef SimArgs Args;
    static constexpr int WR = 2, WC = 2, NSEG = 1, A_ORI = ORI_KC, B_ORI = ORI_KC;
    Args a;
    int b, g, m0, n0;
    __device__ void init(const Args& args) {
        a = args;
        b = blockIdx.z / a.G;  g = blockIdx.z % a.G;
        m0 = blockIdx.y * BM;  n0 = blockIdx.x * BN;
    }
    __device__ int nk(int) const { return MSF2_QK_DIM / BK; }
    __device__ float4 row(const float* p, int i, int k) const {
        const int s = g * MSF2_GROUP + i;
        if (s >= a.S) return make_float4(0.f, 0.f, 0.f, 0.f);
        return *reinterpret_cast<const float4*>(p + ((long)b * a.S + s) * MSF2_QK_DIM + k);
    }
    __device__ float4 fetchA(int, int r, int k) const { return row(a.q, m0 + r, k); }
    __device__ float4 fetchB(int, int c, int k) const { return row(a.k, n0 + c, k); }
    template <int TM, int TN>
    __device__ void epilogue(f32x16 (&acc)[1][TM][TN], int wr, int wc, int lane) const {
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
                    P[i * MSF2_GROUP + j] = x * x;
                }
    }
};

// ---------------------------------------------------------------------------
// Linear-attention key/value summary, split over time:
//   part[b, sp, d, e] = sum_{s in chunk sp} lin_k[b, s, d] * hu[b, s, e]
// grid = (N / 128, nsplit, B); reduced and scaled by 1/S in kv_finalize_kernel.
// ---------------------------------------------------------------------------
struct LinKVArgs {
    const float* lk;                 // [B*S][128]
    const float* hu;  int ldhu;      // [B*S][ldhu], first N columns used
    float* part;                     // [B][nsplit][128][N]
    int S, N, chunk;
};
struct LinKV {
    typedef LinKVArgs Args;
    static constexpr int WR = 2, WC = 2, NSEG = 1, A_ORI = ORI_RC, B_ORI = ORI_RC;
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
    __device__ float4 fetchA(int, int r, int k) const {
        const int s = k0 + k;
        if (s >= k1) return make_float4(0.f, 0.f, 0.f, 0.f);
        return *reinterpret_cast<const float4*>(a.lk + ((long)b * a.S + s) * MSF2_QK_DIM + r);
    }
    __device__ float4 fetchB(int, int c, int k) const {
        const int s = k0 + k;
        if (s >= k1 || n0 + c >= a.N) return make_float4(0.f, 0.f, 0.f, 0.f);
        return *reinterpret_cast<const float4*>(a.hu + ((long)b * a.S + s) * a.ldhu + n0 + c);
    }
    template <int TM, int TN>
    __device__ void epilogue(f32x16 (&acc)[1][TM][TN], int wr, int wc, int lane) const {
        float* out = a.part + ((long)b * gridDim.y + blockIdx.y) * MSF2_QK_DIM * a.N;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int d = MSF2_ACC_ROW(wr, TM, tm, reg, lane);
                    const int e = n0 + MSF2_ACC_COL(wc, TN, tn, lane);
                    if (e < a.N) out[(long)d * a.N + e] = acc[0][tm][tn][reg];
                }
    }
};

// ---------------------------------------------------------------------------
// Attention output and gating for one half-group (128 query frames) and 64
// value columns (v and u halves side by side in the B tile):
//   att = [P | lin_q] x [vu rows of the group ; KV],   K = 256 + 128
//   o[m, j] = (att_u * v[m, j]) * sigmoid(att_v * u[m, j])
// grid = (E / 64, 2, B*G)
// ---------------------------------------------------------------------------
struct AttnOutArgs {
    const float* P;                  // [B*G][256][256]
    const float* lq;                 // [B*S][128]
    const float* hu;  int ldhu;      // v | u | ...
    const float* kv;                 // [B][128][2E]
    float* o;                        // [B*S][E]
    int S, G, E;
};
struct AttnOut {
    typedef AttnOutArgs Args;
    static constexpr int WR = 4, WC = 1, NSEG = 1, A_ORI = ORI_KC, B_ORI = ORI_RC;
    Args a;
    int b, g, m0, j0;
    __device__ void init(const Args& args) {
        a = args;
        b = blockIdx.z / a.G;  g = blockIdx.z % a.G;
        m0 = blockIdx.y * BM;  j0 = blockIdx.x * (BN / 2);
    }
    __device__ int nk(int) const { return (MSF2_GROUP + MSF2_QK_DIM) / BK; }
    __device__ float4 fetchA(int, int r, int k) const {
        const int i = m0 + r;
        if (k < MSF2_GROUP)
            return *reinterpret_cast<const float4*>(a.P + ((long)blockIdx.z * MSF2_GROUP + i) * MSF2_GROUP + k);
        const int s = g * MSF2_GROUP + i;
        if (s >= a.S) return make_float4(0.f, 0.f, 0.f, 0.f);
        return *reinterpret_cast<const float4*>(a.lq + ((long)b * a.S + s) * MSF2_QK_DIM + (k - MSF2_GROUP));
    }
    __device__ float4 fetchB(int, int c, int k) const {
        const int e = (c < BN / 2) ? (j0 + c) : (a.E + j0 + c - BN / 2);
        if (k < MSF2_GROUP) {
            const int s = g * MSF2_GROUP + k;
            if (s >= a.S) return make_float4(0.f, 0.f, 0.f, 0.f);
            return *reinterpret_cast<const float4*>(a.hu + ((long)b * a.S + s) * a.ldhu + e);
        }
        return *reinterpret_cast<const float4*>(a.kv + ((long)b * MSF2_QK_DIM + (k - MSF2_GROUP)) * (2 * a.E) + e);
    }
    template <int TM, int TN>
    __device__ void epilogue(f32x16 (&acc)[1][TM][TN], int wr, int wc, int lane) const {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN / 2; ++tn)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int s = g * MSF2_GROUP + m0 + MSF2_ACC_ROW(wr, TM, tm, reg, lane);
                    if (s >= a.S) continue;
                    const int j = j0 + MSF2_ACC_COL(wc, TN, tn, lane);
                    const long m = (long)b * a.S + s;
                    const float av = acc[0][tm][tn][reg], au = acc[0][tm][tn + TN / 2][reg];
                    const float v = a.hu[m * a.ldhu + j], u = a.hu[m * a.ldhu + a.E + j];
                    a.o[m * a.E + j] = (au * v) * sigmoid_f(av * u);
                }
    }
};

}  // namespace msf2
