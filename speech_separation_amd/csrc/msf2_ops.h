// Bandwidth-bound kernels of the separation forward pass: every one streams
// token-major [B, S, C] rows with the channel index on the lanes (float4 per
// lane, 128-bit coalesced), reduces with wavefront shuffles, and folds the
// statistics the next normalisation needs into the producer.
#pragma once
#include "msf2_common.h"

namespace msf2 {

// ---- zero a small buffer (n16 16-byte units).  Used instead of hipMemsetAsync for the zero operand rows of the
// token shift: the launch sequence must stay correct when it is captured into a HIP graph and replayed, and a
// captured memset node was seen to fill with a stale pattern on later replays (ROCm 7.2).
__global__ void zero_fill_kernel(float4* __restrict__ p, int n16) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}


// Mapping used by the per-frame kernels: a 256-thread block covers
// nf = 256 / (C/4) frames at a time, thread = (frame offset fo, channel quad cq).
// Requires C in {64, 128, 256, 512, 1024}.
struct QuadMap {
    int cq, fo, nf;
    __device__ QuadMap(int C) {
        const int q = C >> 2;
        cq = threadIdx.x % q;
        fo = threadIdx.x / q;
        nf = 256 / q;
    }
};

__device__ __forceinline__ void write_partial(float sum, float sq, float* part, float* red) {
    // part -> two floats of this block
    const float s = block_sum_256(sum, red);
    const float q = block_sum_256(sq, red + 4);
    if (threadIdx.x == 0) { part[0] = s; part[1] = q; }
}

// ---- encoder: relu(conv1d(k=16, stride 8)), token-major, + partial sums -----
// grid (ceil(S/TS), B), block 256.  w: [16][C].  part: [B][gridDim.x][2]
__global__ __launch_bounds__(256) void encoder_kernel(const float* __restrict__ mix, const float* __restrict__ w,
                                                      float* __restrict__ enc, float* __restrict__ part,
                                                      int T, int S, int C, int TS) {
    __shared__ float red[8];
    const QuadMap qm(C);
    const int b = blockIdx.y, s0 = blockIdx.x * TS, s1 = min(s0 + TS, S);
    float4 wq[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) wq[k] = *reinterpret_cast<const float4*>(w + k * C + 4 * qm.cq);
    float sum = 0.f, sq = 0.f;
    for (int s = s0 + qm.fo; s < s1; s += qm.nf) {
        const float* x = mix + (long)b * T + 8 * s;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float xv = x[k];
            acc.x += wq[k].x * xv; acc.y += wq[k].y * xv; acc.z += wq[k].z * xv; acc.w += wq[k].w * xv;
        }
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
        *reinterpret_cast<float4*>(enc + ((long)b * S + s) * C + 4 * qm.cq) = acc;
        sum += acc.x + acc.y + acc.z + acc.w;
        sq += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
    }
    write_partial(sum, sq, part + ((long)b * gridDim.x + blockIdx.x) * 2, red);
}

// ---- (mean, rstd) of one utterance from block partials ----------------------
// grid B, block 64.  stats: [B][2]
// Optionally also folds the statistics into a per-column vector of the contraction that consumes them:
// c1f[b][n] = c1[n] - rstd_b * mean_b * c2[n]  (C columns; skipped when c1f is null).
__global__ __launch_bounds__(64) void stats_finalize_kernel(const float* __restrict__ part, int nblk, double count,
                                                            float eps, float* __restrict__ stats,
                                                            const float* __restrict__ c1, const float* __restrict__ c2,
                                                            float* __restrict__ c1f, int C) {
    const int b = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) {
        s += (double)part[((long)b * nblk + i) * 2];
        q += (double)part[((long)b * nblk + i) * 2 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    const double mean = s / count;      // every lane holds the full sums
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mean_f = (float)mean, rstd_f = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        stats[2 * b] = mean_f;
        stats[2 * b + 1] = rstd_f;
    }
    if (c1f) {
        const float k = rstd_f * mean_f;
        for (int i = threadIdx.x; i < C; i += 64) c1f[(long)b * C + i] = c1[i] - k * c2[i];
    }
}

// ---- ScaledSinuEmbedding table: pe[s][j] = sin(s f_j) scale | cos(s f_j) scale
__global__ void posenc_kernel(const float* __restrict__ inv_freq, float scale, float* __restrict__ pe, int S, int C) {
    const int half = C >> 1;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)S * half) return;
    const int s = (int)(i / half), j = (int)(i % half);
    const float ang = (float)s * inv_freq[j];
    pe[(long)s * C + j] = sinf(ang) * scale;
    pe[(long)s * C + half + j] = cosf(ang) * scale;
}

// ---- ScaleNorm inverse norm from partial sums of squares written by the producer of the rows ----------
// part: [M][np]; out[m] = 1 / max(sqrt(sum) / sqrt(C), 1e-5).  shift: the first np / 2 partials (first half of
// the features) are taken from the previous frame of the same utterance, none at frame 0 (token shift).
__global__ void rowsq_finalize_kernel(const float* __restrict__ part, int np, float* __restrict__ out, long M, int S, int C, int shift) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float ss = 0.f;
    const int h = shift ? np / 2 : 0;
    if ((np & 7) == 0) {           // 16-byte loads, summed in the same order as the scalar loop
        const float4* own = reinterpret_cast<const float4*>(part + m * np);
        for (int p = h / 4; p < np / 4; ++p) { const float4 v = own[p]; ss += v.x; ss += v.y; ss += v.z; ss += v.w; }
        if (shift && m % S != 0) {
            const float4* prev = reinterpret_cast<const float4*>(part + (m - 1) * np);
            for (int p = 0; p < h / 4; ++p) { const float4 v = prev[p]; ss += v.x; ss += v.y; ss += v.z; ss += v.w; }
        }
    } else {
        for (int p = h; p < np; ++p) ss += part[m * np + p];
        if (shift && m % S != 0)
            for (int p = 0; p < h; ++p) ss += part[(m - 1) * np + p];
    }
    out[m] = 1.0f / fmaxf(sqrtf(ss) * rsqrtf((float)C), 1e-5f);
}

// ---- per-row statistics, one wave per row -----------------------------------
// MODE 0: ScaleNorm inverse norm 1 / max(|x| / sqrt(C), 1e-5) of the
//         (optionally token-shifted) row -> out[m]
// MODE 1: LayerNorm (mean, rstd) -> out[2m], out[2m+1]
// grid ceil(M/4), block 256.  C <= 1024.
// IN16: x points at bf16 values (same indexing).
template <int MODE, bool IN16 = false>
__global__ __launch_bounds__(256) void rowstat_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out,
                                                      long M, int S, int C, int shift, float eps) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int nq = C >> 2;
    float4 v[4];
    float sum = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = lane + 64 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq) {
            long row = m;
            bool zero = false;
            if (MODE == 0 && shift && 4 * q < (C >> 1)) {
                if (m % S == 0) zero = true;
                row = m - 1;
            }
            if (!zero) {
                if constexpr (IN16) {
                    const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(x) + row * ldx + 4 * q);
                    v[i] = make_float4(__builtin_bit_cast(float, h.x << 16), __builtin_bit_cast(float, h.x & 0xffff0000u),
                                       __builtin_bit_cast(float, h.y << 16), __builtin_bit_cast(float, h.y & 0xffff0000u));
                } else {
                    v[i] = *reinterpret_cast<const float4*>(x + row * ldx + 4 * q);
                }
            }
        }
        sum += v[i].x + v[i].y + v[i].z + v[i].w;
        ss += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
    if (MODE == 0) {
        ss = wave_sum(ss);
        if (lane == 0) out[m] = 1.0f / fmaxf(sqrtf(ss) * rsqrtf((float)C), 1e-5f);
    } else {
        const float mean = wave_sum(sum) / (float)C;
        float d2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (lane + 64 * i < nq) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                d2 += a * a + b * b + c * c + d * d;
            }
        }
        d2 = wave_sum(d2);
        if (lane == 0) {
            out[2 * m] = mean;
            out[2 * m + 1] = rsqrtf(d2 / (float)C + eps);
        }
    }
}

// Sums of 16 per-lane values over the 64 lanes of a wave with 17 exchanges instead of 96: at each of the
// first four levels a lane keeps one half of its values and hands the other half to its partner.  On
// return a[0] of lane l is the wave total of value number rev4(l & 15) (bit-reversed low nibble).
__device__ __forceinline__ int rev4(int v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3); }
__device__ __forceinline__ void rows16_sum(float (&a)[16], int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = 16 >> j, m = 1 << j;
        // bitwise selects (v_bfi_b32): written as `up ? a[k + n / 2] : a[k]` hipcc may turn the pair of selects into one
        // dynamically indexed read of a[], which it then lowers to a 16-way compare / select chain per element
        const unsigned msk = (lane & m) ? 0xffffffffu : 0u;
#pragma unroll
        for (int k = 0; k < n / 2; ++k) {
            const unsigned lo = __builtin_bit_cast(unsigned, a[k]), hi = __builtin_bit_cast(unsigned, a[k + n / 2]);
            const float keep = __builtin_bit_cast(float, (hi & msk) | (lo & ~msk));
            const float send = __builtin_bit_cast(float, (lo & msk) | (hi & ~msk));
            a[k] = keep + __shfl_xor(send, m, 64);
        }
    }
    a[0] += __shfl_xor(a[0], 16, 64);
    a[0] += __shfl_xor(a[0], 32, 64);
}

// ---- ConvModule: y = x + dwconv17(x) along time (zero padded) ---------------
// grid (ceil(C/4 / 128), ceil(S/TS), B), block 128, TS = 16.  dw: [17][C]; y16: optional bf16 copy of y
// LN (needs C <= 512, i.e. one block per row tile): also writes the LayerNorm statistics (mean, rstd) of every
// result row to ln_out[(b S + s) 2 ..], two-pass like rowstat_kernel<1>, from the values still in registers.
constexpr int DW_TS = 16;
// IN16: x points at bf16 values (same indexing, pitch ldx in elements).
template <bool LN, bool IN16 = false>
__global__ __launch_bounds__(128) void dwconv17_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dw,
                                                       float* __restrict__ y, int ldy, int S, int C, int accumulate,
                                                       unsigned short* __restrict__ y16, float* __restrict__ ln_out, float ln_eps) {
    __shared__ float red[2][2][DW_TS];
    const int nq = C >> 2;
    const int cq_raw = blockIdx.x * 128 + threadIdx.x;
    if (!LN && cq_raw >= nq) return;
    const bool act = cq_raw < nq;            // LN: idle lanes stay for the block reductions, reading a valid column
    const int cq = act ? cq_raw : nq - 1;
    const int b = blockIdx.z, s0 = blockIdx.y * DW_TS;
    float4 w[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) w[k] = *reinterpret_cast<const float4*>(dw + k * C + 4 * cq);
    float4 acc[DW_TS];
#pragma unroll
    for (int o = 0; o < DW_TS; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xb = x + (long)b * S * ldx + 4 * cq;
    const unsigned short* xb16 = reinterpret_cast<const unsigned short*>(x) + (long)b * S * ldx + 4 * cq;
    float* yb = y + (long)b * S * ldy + 4 * cq;
    // bf16 input: every load of the tile is issued before the first multiply -- the 32 input rows (8 bytes per lane) and, when
    // accumulating, the 16 previous result rows straight into the accumulators -- so that a wave has its whole tile in flight
    // at once (64 + 64 registers) instead of waiting on the rows in the order the arithmetic consumes them.
    uint2 raw[IN16 ? DW_TS + 16 : 1];
    if constexpr (IN16) {
#pragma unroll
        for (int r = -8; r < DW_TS + 8; ++r)
            raw[r + 8] = *reinterpret_cast<const uint2*>(xb16 + (long)min(max(s0 + r, 0), S - 1) * ldx);
        if (accumulate) {
#pragma unroll
            for (int o = 0; o < DW_TS; ++o) acc[o] = *reinterpret_cast<const float4*>(yb + (long)min(s0 + o, S - 1) * ldy);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = -8; r < DW_TS + 8; ++r) {
        const int s = s0 + r;
        // unconditional load from the clamped row, zeroed by a 0/1 factor: a predicated load makes
        // hipcc branch around it and wait for it on the spot, which serialises the 32 row loads
        const float inb = (s >= 0 && s < S) ? 1.f : 0.f;
        float4 xv;
        if constexpr (IN16) {
            const uint2 hq = raw[r + 8];
            xv = make_float4(__builtin_bit_cast(float, hq.x << 16), __builtin_bit_cast(float, hq.x & 0xffff0000u),
                             __builtin_bit_cast(float, hq.y << 16), __builtin_bit_cast(float, hq.y & 0xffff0000u));
        } else {
            xv = *reinterpret_cast<const float4*>(xb + (long)min(max(s, 0), S - 1) * ldx);
        }
        xv.x *= inb; xv.y *= inb; xv.z *= inb; xv.w *= inb;
#pragma unroll
        for (int o = 0; o < DW_TS; ++o) {
            const int k = r - o + 8;
            if (k >= 0 && k < 17) {
                acc[o].x += w[k].x * xv.x; acc[o].y += w[k].y * xv.y;
                acc[o].z += w[k].z * xv.z; acc[o].w += w[k].w * xv.w;
            }
            if (r == o) { acc[o].x += xv.x; acc[o].y += xv.y; acc[o].z += xv.z; acc[o].w += xv.w; }
        }
    }
#pragma unroll
    for (int o = 0; o < DW_TS; ++o) {
        const int s = s0 + o;
        if (!IN16 && accumulate) {      // (uniform) previous value of the row, from the clamped row like the inputs
            const float4 p = *reinterpret_cast<const float4*>(yb + (long)min(s, S - 1) * ldy);
            acc[o].x += p.x; acc[o].y += p.y; acc[o].z += p.z; acc[o].w += p.w;
        }
        if (s < S && act) {
            const float4 v = acc[o];
            *reinterpret_cast<float4*>(yb + (long)s * ldy) = v;
            if (y16) {   // bf16 image of the result (pitch ldy) for the contractions that read it next
                uint2 h;
                h.x = pk_bf16_pair(v.x, v.y);
                h.y = pk_bf16_pair(v.z, v.w);
                *reinterpret_cast<uint2*>(y16 + ((long)b * S + s) * ldy + 4 * cq) = h;
            }
        }
    }
    if constexpr (LN) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const float on = act ? 1.f : 0.f, inv_c = 1.f / (float)C;
        float t[DW_TS];
#pragma unroll
        for (int o = 0; o < DW_TS; ++o) t[o] = (acc[o].x + acc[o].y + acc[o].z + acc[o].w) * on;
        rows16_sum(t, lane);
        if (lane < 16) red[0][wave][rev4(lane)] = t[0];
        __syncthreads();
#pragma unroll
        for (int o = 0; o < DW_TS; ++o) {
            const float mean = (red[0][0][o] + red[0][1][o]) * inv_c;
            const float a0 = acc[o].x - mean, a1 = acc[o].y - mean, a2 = acc[o].z - mean, a3 = acc[o].w - mean;
            t[o] = (a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3) * on;
        }
        rows16_sum(t, lane);
        if (lane < 16) red[1][wave][rev4(lane)] = t[0];
        __syncthreads();
        if (threadIdx.x < DW_TS && s0 + (int)threadIdx.x < S) {
            const int o = threadIdx.x;
            float* dst = ln_out + ((long)b * S + s0 + o) * 2;
            dst[0] = (red[0][0][o] + red[0][1][o]) * inv_c;
            dst[1] = rsqrtf((red[1][0][o] + red[1][1][o]) * inv_c + ln_eps);
        }
    }
}

// bf16 in, bf16 out (bf16 mode keeps the [M, 4C + 128] hidden tensor in bf16 on both sides of the
// ConvModule); arithmetic in fp32.  Same grid and tiling as dwconv17_kernel, 8-byte accesses.
__device__ __forceinline__ float4 bf16x4_ld(const unsigned short* p) {
    const uint2 h = *reinterpret_cast<const uint2*>(p);
    return make_float4(__builtin_bit_cast(float, h.x << 16), __builtin_bit_cast(float, h.x & 0xffff0000u),
                       __builtin_bit_cast(float, h.y << 16), __builtin_bit_cast(float, h.y & 0xffff0000u));
}
constexpr int DW16_TS = 32;      // frames per block of the bf16 kernel: 48 row reads per 32 results
__global__ __launch_bounds__(128, 2) void dwconv17_bf16_kernel(const unsigned short* __restrict__ x, int ldx,
                                                            const float* __restrict__ dw, unsigned short* __restrict__ y,
                                                            int ldy, int S, int C) {
    const int cq = blockIdx.x * 128 + threadIdx.x;
    if (cq >= (C >> 2)) return;
    const int b = blockIdx.z, s0 = blockIdx.y * DW16_TS;
    float4 w[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) w[k] = *reinterpret_cast<const float4*>(dw + k * C + 4 * cq);
    w[8].x += 1.f; w[8].y += 1.f; w[8].z += 1.f; w[8].w += 1.f;      // y = x + conv(x): the centre tap carries the identity
    float4 acc[DW16_TS];
#pragma unroll
    for (int o = 0; o < DW16_TS; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned short* xb = x + (long)b * S * ldx + 4 * cq;
#pragma unroll
    for (int r = -8; r < DW16_TS + 8; ++r) {
        const int s = s0 + r;
        const float inb = (s >= 0 && s < S) ? 1.f : 0.f;      // see dwconv17_kernel
        float4 xv = bf16x4_ld(xb + (long)min(max(s, 0), S - 1) * ldx);
        xv.x *= inb; xv.y *= inb; xv.z *= inb; xv.w *= inb;
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            const int o = r - k + 8;             // the result row this input row feeds through tap k
            if (o >= 0 && o < DW16_TS) {
                acc[o].x += w[k].x * xv.x; acc[o].y += w[k].y * xv.y;
                acc[o].z += w[k].z * xv.z; acc[o].w += w[k].w * xv.w;
            }
        }
    }
    unsigned short* yb = y + (long)b * S * ldy + 4 * cq;
#pragma unroll
    for (int o = 0; o < DW16_TS; ++o) {
        const int s = s0 + o;
        if (s < S) {
            uint2 h;
            h.x = pk_bf16_pair(acc[o].x, acc[o].y);
            h.y = pk_bf16_pair(acc[o].z, acc[o].w);
            *reinterpret_cast<uint2*>(yb + (long)s * ldy) = h;
        }
    }
}

// ---- OffsetScale + rotary -> quad_q/256, lin_q, quad_k, lin_k ----------------
// qk: [M] rows, row stride ld (already offset to the qk columns).  q4: [4][M][128].
// grid ceil(M/4), block 256 (64 lanes = 64 feature pairs of one row)
// IN16 (bf16 mode): qk points at bf16 values (same indexing) and the outputs are written as bf16
// (slot h of q4 then holds [M][128] bf16 in its first half).
template <bool IN16>
__global__ __launch_bounds__(256) void qk_prep_kernel(const void* __restrict__ qk_, int ld,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ freqs, float* __restrict__ q4,
                                                      long M, int S) {
    const int p = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float2 x;
    if constexpr (IN16) {
        const unsigned h = *reinterpret_cast<const unsigned*>(static_cast<const unsigned short*>(qk_) + m * ld + 2 * p);
        x = make_float2(__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u));
    } else {
        x = *reinterpret_cast<const float2*>(static_cast<const float*>(qk_) + m * ld + 2 * p);
    }
    float c = 1.f, sn = 0.f;
    if (p < 16) {
        const float ang = (float)(m % S) * freqs[p];
        c = cosf(ang);
        sn = sinf(ang);
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float2 g = *reinterpret_cast<const float2*>(gamma + h * MSF2_QK_DIM + 2 * p);
        const float2 bb = *reinterpret_cast<const float2*>(beta + h * MSF2_QK_DIM + 2 * p);
        const float y0 = x.x * g.x + bb.x, y1 = x.y * g.y + bb.y;
        float2 o;
        o.x = y0 * c - y1 * sn;
        o.y = y1 * c + y0 * sn;
        if (h == 0) { o.x *= (1.0f / MSF2_GROUP); o.y *= (1.0f / MSF2_GROUP); }
        if (IN16)      // bf16 mode: the four outputs are only ever bf16 contraction operands; each slot holds [M][128] bf16
            *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(q4 + (long)h * M * MSF2_QK_DIM) + m * MSF2_QK_DIM + 2 * p) = pk_bf16_pair(o.x, o.y);
        else
            *reinterpret_cast<float2*>(q4 + ((long)h * M + m) * MSF2_QK_DIM + 2 * p) = o;
    }
}

// bf16 form with 8-byte accesses: a lane owns 4 features (2 rotary pairs), 32 lanes per row, 8 rows per block of 256.
// qk: bf16 [M] rows of pitch ld; q4 slot h holds [M][128] bf16 in the first half of its fp32-sized slot.
// Sine / cosine from the hardware instructions (argument error <= 3e-4 rad at frame 4000, an eighth of the bf16
// rounding step of the outputs) instead of the library's range reduction.
__global__ __launch_bounds__(256) void qk_prep_bf16_kernel(const unsigned short* __restrict__ qk, int ld,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ freqs, float* __restrict__ q4,
                                                           long M, int S) {
    // a thread handles its 4 columns of QP_ROWS rows, 8 rows apart: all row loads first, the per-head gains once
    constexpr int QP_ROWS = 4;
    const int p = threadIdx.x & 31;
    const long mb = (long)blockIdx.x * (8 * QP_ROWS) + (threadIdx.x >> 5);
    float4 xs[QP_ROWS];
#pragma unroll
    for (int r = 0; r < QP_ROWS; ++r) {
        const long m = mb + 8 * r;
        xs[r] = bf16x4_ld(qk + (m < M ? m : M - 1) * ld + 4 * p);
    }
    float4 gs[4], bs[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        gs[h] = *reinterpret_cast<const float4*>(gamma + h * MSF2_QK_DIM + 4 * p);
        bs[h] = *reinterpret_cast<const float4*>(beta + h * MSF2_QK_DIM + 4 * p);
    }
    const float f0 = (p < 8) ? freqs[2 * p] : 0.f, f1 = (p < 8) ? freqs[2 * p + 1] : 0.f;
#pragma unroll
    for (int r = 0; r < QP_ROWS; ++r) {
        const long m = mb + 8 * r;
        if (m >= M) break;
        const float4 x = xs[r];
        float c0 = 1.f, s0 = 0.f, c1 = 1.f, s1 = 0.f;
        if (p < 8) {
            const float pos = (float)(m % S);
            const float a0 = pos * f0, a1 = pos * f1;
            c0 = __cosf(a0); s0 = __sinf(a0);
            c1 = __cosf(a1); s1 = __sinf(a1);
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const float4 g = gs[h], bb = bs[h];
            const float y0 = x.x * g.x + bb.x, y1 = x.y * g.y + bb.y, y2 = x.z * g.z + bb.z, y3 = x.w * g.w + bb.w;
            float o0 = y0 * c0 - y1 * s0, o1 = y1 * c0 + y0 * s0;
            float o2 = y2 * c1 - y3 * s1, o3 = y3 * c1 + y2 * s1;
            if (h == 0) { o0 *= (1.0f / MSF2_GROUP); o1 *= (1.0f / MSF2_GROUP); o2 *= (1.0f / MSF2_GROUP); o3 *= (1.0f / MSF2_GROUP); }
            uint2 v;
            v.x = pk_bf16_pair(o0, o1);
            v.y = pk_bf16_pair(o2, o3);
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(q4 + (long)h * M * MSF2_QK_DIM) + m * MSF2_QK_DIM + 4 * p) = v;
        }
    }
}

// ---- kv[b][d][e] = sum_sp part[b][sp][d][e] / S ------------------------------
__global__ void kv_finalize_kernel(const float* __restrict__ part, float* __restrict__ kv, int nsplit, long per_b, int S, long total) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= total) return;
    const long b = i / per_b, r = i % per_b;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sp = 0; sp < nsplit; ++sp) {
        const float4 v = *reinterpret_cast<const float4*>(part + (b * nsplit + sp) * per_b + r);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float n = (float)S;
    acc.x /= n; acc.y /= n; acc.z /= n; acc.w /= n;
    *reinterpret_cast<float4*>(kv + i) = acc;
}

// ---- TCN stage A: t = silu(dwconv3_dilated(h) + bias), + partial sums --------
// fp32 in / fp32 out: 4 channels (16 bytes) per lane.  A thread owns NIT frames that are `dil` apart
// (base + j dil), so the taps of one frame are the centre rows of its neighbours and the thread reads NIT + 2
// rows instead of 3 NIT; thread row T = blockIdx.x * (256 / lanes per frame) + (threadIdx.x / lanes per frame)
// maps to base = (T / dil) * NIT * dil + T % dil, which covers every frame exactly once.  All rows are loaded up
// front from clamped rows (taps outside the utterance are zeroed by a 0/1 factor, never by a predicated load,
// which hipcc would wait on in place).  grid (ceil(ceil(S / (NIT dil)) dil / rows per block), B), block 256.
template <int NIT>
__global__ __launch_bounds__(256) void tcn_stage_a_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ t,
                                                          float* __restrict__ part, int S, int C, int dil) {
    __shared__ float red[8];
    const int lpf = C >> 2, cq = threadIdx.x % lpf, fo = threadIdx.x / lpf, nf = 256 / lpf;
    const int b = blockIdx.y;
    const int T = blockIdx.x * nf + fo;
    const int base = (T / dil) * (NIT * dil) + (T % dil);
    const int c = 4 * cq;
    const float4 w0 = *reinterpret_cast<const float4*>(w + c);
    const float4 w1 = *reinterpret_cast<const float4*>(w + C + c);
    const float4 w2 = *reinterpret_cast<const float4*>(w + 2 * C + c);
    const float4 bs = *reinterpret_cast<const float4*>(bias + c);
    const float* hb = h + (long)b * S * C + c;
    float4 x[NIT + 2];       // rows base + (j - 1) dil
#pragma unroll
    for (int j = 0; j < NIT + 2; ++j)
        x[j] = *reinterpret_cast<const float4*>(hb + (long)min(max(base + (j - 1) * dil, 0), S - 1) * C);
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int s = base + j * dil;
        const float fm = (s - dil >= 0) ? 1.f : 0.f, fp = (s + dil < S) ? 1.f : 0.f;
        float4 v;
        v.x = silu_f(w0.x * (x[j].x * fm) + w1.x * x[j + 1].x + w2.x * (x[j + 2].x * fp) + bs.x);
        v.y = silu_f(w0.y * (x[j].y * fm) + w1.y * x[j + 1].y + w2.y * (x[j + 2].y * fp) + bs.y);
        v.z = silu_f(w0.z * (x[j].z * fm) + w1.z * x[j + 1].z + w2.z * (x[j + 2].z * fp) + bs.z);
        v.w = silu_f(w0.w * (x[j].w * fm) + w1.w * x[j + 1].w + w2.w * (x[j + 2].w * fp) + bs.w);
        if (s < S) {
            *reinterpret_cast<float4*>(t + ((long)b * S + s) * C + c) = v;
            sum += v.x + v.y + v.z + v.w;
            sq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    }
    write_partial(sum, sq, part + ((long)b * gridDim.x + blockIdx.x) * 2, red);
}

// bf16 in / bf16 out form of stage A for the bf16 mode: 8 channels (16 bytes) per lane.
// A thread owns NIT frames that are `dil` apart: base + j dil, j = 0 .. NIT-1, so the taps of one frame are the
// centre rows of its neighbours and the thread reads NIT + 2 rows instead of 3 NIT (each row of h once, plus two
// per thread).  Thread row T = blockIdx.x * (256 / lanes per frame) + (threadIdx.x / lanes per frame) maps to
// segment q = T / dil, residue r = T % dil, base = q * NIT * dil + r: every frame belongs to exactly one (q, j, r).
// All rows are loaded up front from clamped rows; taps outside the utterance are zeroed by a 0/1 factor, never by
// a predicated load.  grid (ceil(ceil(S / (NIT dil)) dil / rows per block), B), block 256.  Statistics from the fp32 values.
template <int NIT>
__global__ __launch_bounds__(256) void tcn_stage_a_bf16_kernel(const unsigned short* __restrict__ h16, const float* __restrict__ w,
                                                               const float* __restrict__ bias, unsigned short* __restrict__ t16,
                                                               float* __restrict__ part, int S, int C, int dil) {
    __shared__ float red[8];
    const int lpf = C >> 3, cq = threadIdx.x % lpf, fo = threadIdx.x / lpf, nf = 256 / lpf;
    const int b = blockIdx.y;
    const int T = blockIdx.x * nf + fo;
    const int base = (T / dil) * (NIT * dil) + (T % dil);
    const int c = 8 * cq;
    // arithmetic on channel pairs (2 q, 2 q + 1) -- the two halves of one packed word -- so that it maps onto the packed
    // fp32 instructions: per pair 3 fused multiply-adds, the SiLU (2 exponentials, 2 reciprocals, 3 packed operations),
    // 2 packed operations for the statistics and one conversion
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 wk[3][4], bs[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float4 lo = *reinterpret_cast<const float4*>(w + k * C + c), hi = *reinterpret_cast<const float4*>(w + k * C + c + 4);
        wk[k][0] = f2{lo.x, lo.y}; wk[k][1] = f2{lo.z, lo.w}; wk[k][2] = f2{hi.x, hi.y}; wk[k][3] = f2{hi.z, hi.w};
    }
    {
        const float4 lo = *reinterpret_cast<const float4*>(bias + c), hi = *reinterpret_cast<const float4*>(bias + c + 4);
        bs[0] = f2{lo.x, lo.y}; bs[1] = f2{lo.z, lo.w}; bs[2] = f2{hi.x, hi.y}; bs[3] = f2{hi.z, hi.w};
    }
    const unsigned short* hb = h16 + (long)b * S * C + c;
    uint4 x[NIT + 2];        // rows base + (j - 1) dil, j = 0 .. NIT + 1
#pragma unroll
    for (int j = 0; j < NIT + 2; ++j) {
        const int s = min(max(base + (j - 1) * dil, 0), S - 1);
        x[j] = *reinterpret_cast<const uint4*>(hb + (long)s * C);
    }
    // a row outside the utterance is a zero tap: cleared as packed words (4 operations per row instead of a 0/1 factor on
    // each of its 16 uses); such a row is the centre only of frames that are not stored
#pragma unroll
    for (int j = 0; j < NIT + 2; ++j) {
        const int r = base + (j - 1) * dil;
        const unsigned keep = (r >= 0 && r < S) ? 0xffffffffu : 0u;
        x[j].x &= keep; x[j].y &= keep; x[j].z &= keep; x[j].w &= keep;
    }
    f2 sum2 = {0.f, 0.f}, sq2 = {0.f, 0.f};
    // each row is unpacked once and serves as the right, centre and left tap of three consecutive frames
    auto unpack = [](const uint4& r, f2 (&f)[4]) __attribute__((always_inline)) {
        const unsigned wds[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = f2{__builtin_bit_cast(float, wds[q] << 16), __builtin_bit_cast(float, wds[q] & 0xffff0000u)};
    };
    f2 m_[4], z_[4], p_[4];
    unpack(x[0], m_);
    unpack(x[1], z_);
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
        const int s = base + j * dil;
        unpack(x[j + 2], p_);
        f2 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f2 y = __builtin_elementwise_fma(wk[0][q], m_[q], __builtin_elementwise_fma(wk[1][q], z_[q], __builtin_elementwise_fma(wk[2][q], p_[q], bs[q])));
            const f2 t2 = y * -1.44269504088896341f;
            const f2 d = f2{__builtin_amdgcn_exp2f(t2.x), __builtin_amdgcn_exp2f(t2.y)} + 1.0f;
            v[q] = y * f2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        }
        if (s < S) {
            uint4 pk;
            pk.x = pk_bf16_pair(v[0].x, v[0].y); pk.y = pk_bf16_pair(v[1].x, v[1].y);
            pk.z = pk_bf16_pair(v[2].x, v[2].y); pk.w = pk_bf16_pair(v[3].x, v[3].y);
            *reinterpret_cast<uint4*>(t16 + ((long)b * S + s) * C + c) = pk;
#pragma unroll
            for (int q = 0; q < 4; ++q) { sum2 += v[q]; sq2 = __builtin_elementwise_fma(v[q], v[q], sq2); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { m_[q] = z_[q]; z_[q] = p_[q]; }
    }
    const float sum = sum2.x + sum2.y, sq = sq2.x + sq2.y;
    write_partial(sum, sq, part + ((long)b * gridDim.x + blockIdx.x) * 2, red);
}

// ---- final LayerNorm (eps, affine) in place + partial sums of the result -----
// grid (ceil(S/32), B), block 256: each wave normalises 8 rows.
__global__ __launch_bounds__(256) void final_ln_kernel(float* __restrict__ x, const float* __restrict__ gw,
                                                       const float* __restrict__ gb, float* __restrict__ part,
                                                       int S, int C, float eps) {
    __shared__ float red[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    const int nq = C >> 2;
    float sum = 0.f, sq = 0.f;
    for (int r = 0; r < 8; ++r) {
        const int s = blockIdx.x * 32 + wave * 8 + r;
        if (s >= S) break;
        float* row = x + ((long)b * S + s) * C;
        float4 v[4];
        float rs = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = lane + 64 * i;
            v[i] = (q < nq) ? *reinterpret_cast<const float4*>(row + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            rs += v[i].x + v[i].y + v[i].z + v[i].w;
        }
        const float mean = wave_sum(rs) / (float)C;
        float d2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lane + 64 * i < nq) {
                const float a = v[i].x - mean, bq = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                d2 += a * a + bq * bq + c * c + d * d;
            }
        const float rstd = rsqrtf(wave_sum(d2) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = lane + 64 * i;
            if (q < nq) {
                const float4 g = *reinterpret_cast<const float4*>(gw + 4 * q);
                const float4 o = *reinterpret_cast<const float4*>(gb + 4 * q);
                float4 y;
                y.x = (v[i].x - mean) * rstd * g.x + o.x;
                y.y = (v[i].y - mean) * rstd * g.y + o.y;
                y.z = (v[i].z - mean) * rstd * g.z + o.z;
                y.w = (v[i].w - mean) * rstd * g.w + o.w;
                *reinterpret_cast<float4*>(row + 4 * q) = y;
                sum += y.x + y.y + y.z + y.w;
                sq += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
            }
        }
    }
    write_partial(sum, sq, part + ((long)b * gridDim.x + blockIdx.x) * 2, red);
}

// ---- out = prelu(GroupNorm(y) + skip); optional tap of the pre-PReLU value ---
// flat over M*C/4 float4s; out16: optional bf16 image of out (operand of the next contraction in bf16 mode)
__global__ void final_gn_kernel(const float* __restrict__ y, const float* __restrict__ skip,
                                const float* __restrict__ stats, const float* __restrict__ gw,
                                const float* __restrict__ gb, float prelu_a, float* __restrict__ out,
                                float* __restrict__ tap, long M, int S, int C, unsigned short* __restrict__ out16) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = C >> 2;
    if (i >= M * nq) return;
    const long m = i / nq;
    const int c = (int)(i % nq) * 4;
    const int b = (int)(m / S);
    const float mean = stats[2 * b], rstd = stats[2 * b + 1];
    const float4 v = *reinterpret_cast<const float4*>(y + m * C + c);
    const float4 k = *reinterpret_cast<const float4*>(skip + m * C + c);
    const float4 g = *reinterpret_cast<const float4*>(gw + c);
    const float4 o = *reinterpret_cast<const float4*>(gb + c);
    float4 r;
    r.x = (v.x - mean) * rstd * g.x + o.x + k.x;
    r.y = (v.y - mean) * rstd * g.y + o.y + k.y;
    r.z = (v.z - mean) * rstd * g.z + o.z + k.z;
    r.w = (v.w - mean) * rstd * g.w + o.w + k.w;
    if (tap) *reinterpret_cast<float4*>(tap + m * C + c) = r;
    r.x = r.x >= 0.f ? r.x : prelu_a * r.x;
    r.y = r.y >= 0.f ? r.y : prelu_a * r.y;
    r.z = r.z >= 0.f ? r.z : prelu_a * r.z;
    r.w = r.w >= 0.f ? r.w : prelu_a * r.w;
    *reinterpret_cast<float4*>(out + m * C + c) = r;
    if (out16) {
        uint2 h;
        h.x = pk_bf16_pair(r.x, r.y);
        h.y = pk_bf16_pair(r.z, r.w);
        *reinterpret_cast<uint2*>(out16 + m * C + c) = h;
    }
}

// ---- decoder stage 1: fv[m][k] = sum_c sep[m][c] * W[c][k], one wave per frame
// dynamic LDS: C * 20 floats (rows padded to 20 so float4 reads are conflict-free)
__global__ __launch_bounds__(256) void decoder_frames_kernel(const float* __restrict__ sep, const float* __restrict__ w,
                                                             float* __restrict__ fv, long M, int C) {
    extern __shared__ __attribute__((aligned(16))) float sw[];
    for (int i = threadIdx.x; i < C * 16; i += 256) sw[(i >> 4) * 20 + (i & 15)] = w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long m = (long)blockIdx.x * 4 + wave; m < M; m += (long)gridDim.x * 4) {
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float xv = sep[m * C + c];
            const float4* wr = reinterpret_cast<const float4*>(sw + c * 20);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 wv = wr[j];
                acc[4 * j + 0] += xv * wv.x; acc[4 * j + 1] += xv * wv.y;
                acc[4 * j + 2] += xv * wv.z; acc[4 * j + 3] += xv * wv.w;
            }
        }
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float v = wave_sum(acc[k]);
            if (lane == k) mine = v;
        }
        if (lane < 16) fv[m * 16 + lane] = mine;
    }
}

// The same product for C a multiple of 256 with the weights in registers: a lane owns channels 4 l .. 4 l + 3 of each
// 256-channel slab (16-byte loads, a whole row of 512 channels in two wave loads) and keeps W[c][0..15] of its channels in
// NV * 64 registers, so a frame costs NV loads and NV * 32 packed multiply-adds per lane and no LDS traffic; the 16 sums
// over the 64 lanes take 17 exchanges (rows16_sum) instead of 96.  Four frames per iteration keep 4 NV loads in flight per lane.
template <int NV>      // NV = C / 256
__global__ __launch_bounds__(256) void decoder_frames_reg_kernel(const float* __restrict__ sep, const float* __restrict__ w,
                                                                 float* __restrict__ fv, long M) {
    constexpr int C = 256 * NV;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 wr[NV][4][4];       // [slab][channel of the lane][4 outputs each]
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int j = 0; j < 4; ++j) wr[v][ch][j] = *reinterpret_cast<const float4*>(w + (long)(256 * v + 4 * lane + ch) * 16 + 4 * j);
    const long nw = (long)gridDim.x * 4;
    constexpr int NF = 4;      // frames per iteration
    for (long m0 = NF * ((long)blockIdx.x * 4 + wave); m0 < M; m0 += NF * nw) {
        float4 xs[NF][NV];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const long m = (m0 + f < M) ? m0 + f : M - 1;       // (frames past the end are computed from the last one and not stored)
#pragma unroll
            for (int v = 0; v < NV; ++v) xs[f][v] = *reinterpret_cast<const float4*>(sep + m * C + 256 * v + 4 * lane);
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            float acc[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const float cx[4] = {xs[f][v].x, xs[f][v].y, xs[f][v].z, xs[f][v].w};
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 wv = wr[v][ch][j];
                        acc[4 * j + 0] = __builtin_fmaf(cx[ch], wv.x, acc[4 * j + 0]); acc[4 * j + 1] = __builtin_fmaf(cx[ch], wv.y, acc[4 * j + 1]);
                        acc[4 * j + 2] = __builtin_fmaf(cx[ch], wv.z, acc[4 * j + 2]); acc[4 * j + 3] = __builtin_fmaf(cx[ch], wv.w, acc[4 * j + 3]);
                    }
            }
            rows16_sum(acc, lane);
            if (lane < 16 && m0 + f < M) fv[(m0 + f) * 16 + rev4(lane)] = acc[0];
        }
    }
}

// ---- decoder stage 2: overlap-add of frame vectors, zero pad / trim to T -----
__global__ void decoder_ola_kernel(const float* __restrict__ fv, float* __restrict__ est, int B, int S, int T) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * T) return;
    const int b = (int)(i / T), t = (int)(i % T);
    const int f1 = t >> 3, k = t & 7;
    float v = 0.f;
    if (f1 < S) v += fv[((long)b * S + f1) * 16 + k];
    if (f1 >= 1 && f1 - 1 < S) v += fv[((long)b * S + f1 - 1) * 16 + k + 8];
    est[i] = v;
}

// ---- permutation-invariant SI-SNR of two separated sources -----------------
// est, src: [2][B][T].  One pass over the four waveforms: every block accumulates, in double precision, the
// 12 raw moments (sum and sum of squares of each waveform, the 4 cross products) of its chunk of one
// utterance; pit_finalize turns the moments into the four zero-mean SI-SNR values and picks the better of the
// two speaker assignments.  grid (nblk, B), block 256; part: [B][nblk][12] doubles.
constexpr int PIT_CHUNK = 8192;       // samples per block
__device__ __forceinline__ double block_sum_256_f64(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void pit_moments_kernel(const float* __restrict__ est, const float* __restrict__ src,
                                                          double* __restrict__ part, int B, int T) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    const long t0 = (long)blockIdx.x * PIT_CHUNK, t1 = min(t0 + PIT_CHUNK, (long)T);
    const float* e0 = est + (long)b * T;
    const float* e1 = est + ((long)B + b) * T;
    const float* s0 = src + (long)b * T;
    const float* s1 = src + ((long)B + b) * T;
    double m[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) m[i] = 0.0;
    for (long t = t0 + threadIdx.x; t < t1; t += 256) {
        const double a0 = e0[t], a1 = e1[t], c0 = s0[t], c1 = s1[t];
        m[0] += a0; m[1] += a1; m[2] += c0; m[3] += c1;
        m[4] += a0 * a0; m[5] += a1 * a1; m[6] += c0 * c0; m[7] += c1 * c1;
        m[8] += a0 * c0; m[9] += a0 * c1; m[10] += a1 * c0; m[11] += a1 * c1;
    }
    double* out = part + ((long)b * gridDim.x + blockIdx.x) * 12;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const double v = block_sum_256_f64(m[i], red);
        if (threadIdx.x == 0) out[i] = v;
    }
}
// grid B, block 64.  si_snr: [B][2 (source)][2 (estimate)] in dB; best: [B] mean SI-SNR of the better assignment;
// perm: [B] 0 = (est0->src0, est1->src1), 1 = swapped.
__global__ __launch_bounds__(64) void pit_finalize_kernel(const double* __restrict__ part, int nblk, int T,
                                                          float* __restrict__ si_snr, float* __restrict__ best,
                                                          int* __restrict__ perm) {
    const int b = blockIdx.x;
    double m[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        double v = 0.0;
        for (int k = threadIdx.x; k < nblk; k += 64) v += part[((long)b * nblk + k) * 12 + i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        m[i] = v;
    }
    if (threadIdx.x != 0) return;
    const double n = (double)T, EPS = 1e-8;
    double val[2][2];
    for (int si = 0; si < 2; ++si)
        for (int ei = 0; ei < 2; ++ei) {
            const double se = m[6 + si] - m[2 + si] * m[2 + si] / n;                 // |s~|^2
            const double ee = m[4 + ei] - m[ei] * m[ei] / n;                         // |e~|^2
            const double dot = m[8 + 2 * ei + si] - m[ei] * m[2 + si] / n;           // <e~, s~>
            const double k = dot / (se + EPS);
            const double proj = k * k * se;
            const double noise = fmax(ee - 2.0 * k * dot + proj, 0.0);
            val[si][ei] = 10.0 * log10(proj / (noise + EPS) + EPS);
            if (si_snr) si_snr[(b * 2 + si) * 2 + ei] = (float)val[si][ei];
        }
    const double keep = 0.5 * (val[0][0] + val[1][1]), swap = 0.5 * (val[0][1] + val[1][0]);
    if (best) best[b] = (float)fmax(keep, swap);
    if (perm) perm[b] = swap > keep ? 1 : 0;
}

// ===========================================================================
// FSMN-family block (Gated_FSMN_Block_Dilated): the kernels beside its projections
// ===========================================================================
// ---- LayerNorm over the channels of each row, affine (CLayerNorm): one wave per row, C <= 1024, C % 4 == 0
// y16: optional bf16 image of y (same pitch) for the contraction that reads it next in bf16 mode
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                      const float* __restrict__ bta, float* __restrict__ y, long M, int C, float eps,
                                                      unsigned short* __restrict__ y16) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int nq = C >> 2;
    float4 v[4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = lane + 64 * i;
        v[i] = (q < nq) ? *reinterpret_cast<const float4*>(x + m * C + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        sum += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = wave_sum(sum) / (float)C;
    float d2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < nq) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            d2 += a * a + b * b + c * c + d * d;
        }
    const float rstd = rsqrtf(wave_sum(d2) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = lane + 64 * i;
        if (q < nq) {
            const float4 gg = *reinterpret_cast<const float4*>(g + 4 * q), bb = *reinterpret_cast<const float4*>(bta + 4 * q);
            float4 o;
            o.x = (v[i].x - mean) * rstd * gg.x + bb.x; o.y = (v[i].y - mean) * rstd * gg.y + bb.y;
            o.z = (v[i].z - mean) * rstd * gg.z + bb.z; o.w = (v[i].w - mean) * rstd * gg.w + bb.w;
            *reinterpret_cast<float4*>(y + m * C + 4 * q) = o;
            if (y16) {
                uint2 h;
                h.x = pk_bf16_pair(o.x, o.y);
                h.y = pk_bf16_pair(o.z, o.w);
                *reinterpret_cast<uint2*>(y16 + m * C + 4 * q) = h;
            }
        }
    }
}

// ---- 39-tap convolution along time of the dilated dense net, 'same' zero padding, + per-(utterance, channel)
// partial sums for the InstanceNorm that follows.
//   NIN = 1: depthwise, t[s, c] = sum_k w[k][c] in0[s + (k - 19) dil, c]
//   NIN = 2: 2 input channels per output channel out of the concatenation [in0 | in1] (C channels each):
//            t[s, g] = sum_k sum_i w[k][i][g] cat[s + (k - 19) dil, 2 g + i]
// Sliding window in registers: a thread owns one float2 of input channels (NIN = 1: 2 output channels, NIN = 2: the one
// output channel they feed) and FS_TS output frames that are `dil` apart (s0 + j dil), so the FS_TS + 38 input rows it
// reads are all it needs: 3.4 row reads per result instead of 39.  The window stays in registers (108); every input row
// is loaded once from a clamped address (rows outside the utterance are zeroed by a 0/1 factor, never by a predicated
// load); the taps are read as they are used.
// grid (ceil(threads over channels / 64), ceil(S / (FS_TS dil)) dil, B), block 64; blockIdx.y = tile * dil + residue.
// part: [B][gridDim.y][C][2]
constexpr int FS_TS = 16, FS_K = 39;
template <int NIN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 4))) void fsmn_conv39_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                             const float* __restrict__ w, float* __restrict__ t,
                                                             float* __restrict__ part, int S, int C, int dil) {
    constexpr int CO = (NIN == 1) ? 2 : 1;                 // output channels per thread
    const int c = CO * (blockIdx.x * 64 + threadIdx.x);    // first of them
    if (c >= C) return;
    const int b = blockIdx.z, s0 = (blockIdx.y / dil) * (FS_TS * dil) + (blockIdx.y % dil);
    // input channels of this thread: NIN = 1: c, c+1 of in0; NIN = 2: 2c, 2c+1 of the concatenation
    const float* src = in0;
    int cin = c;
    if (NIN == 2) {
        cin = 2 * c;
        if (cin >= C) { src = in1; cin -= C; }
    }
    const float* base = src + (long)b * S * C + cin;
    // the whole window of this thread: rows s0 + (j - 19) dil, j = 0 .. FS_TS + 37, all requested up front
    constexpr int NR = FS_TS + FS_K - 1;
    float2 x[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int s = s0 + (j - FS_K / 2) * dil;
        x[j] = *reinterpret_cast<const float2*>(base + (long)min(max(s, 0), S - 1) * C);
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int s = s0 + (j - FS_K / 2) * dil;
        const float inb = (s >= 0 && s < S) ? 1.f : 0.f;
        x[j].x *= inb; x[j].y *= inb;
    }
    float2 acc[FS_TS];          // NIN = 2 uses .x only
#pragma unroll
    for (int o = 0; o < FS_TS; ++o) acc[o] = make_float2(0.f, 0.f);
    // tap by tap (the taps of a thread's channels are the same for every frame: cache-resident loads)
#pragma unroll
    for (int k = 0; k < FS_K; ++k) {
        float2 wv;              // NIN = 1: w[k][c .. c+1]; NIN = 2: (w[k][0][c], w[k][1][c])
        if (NIN == 1) wv = *reinterpret_cast<const float2*>(w + (long)k * C + c);
        else wv = make_float2(w[((long)k * 2 + 0) * C + c], w[((long)k * 2 + 1) * C + c]);
#pragma unroll
        for (int o = 0; o < FS_TS; ++o) {
            if (NIN == 1) { acc[o].x += wv.x * x[o + k].x; acc[o].y += wv.y * x[o + k].y; }
            else acc[o].x += wv.x * x[o + k].x + wv.y * x[o + k].y;
        }
    }
    float2 sum = make_float2(0.f, 0.f), sq = sum;
#pragma unroll
    for (int o = 0; o < FS_TS; ++o) {
        const int s = s0 + o * dil;
        if (s < S) {
            if (NIN == 1) *reinterpret_cast<float2*>(t + ((long)b * S + s) * C + c) = acc[o];
            else t[((long)b * S + s) * C + c] = acc[o].x;
            sum.x += acc[o].x; sum.y += acc[o].y;
            sq.x += acc[o].x * acc[o].x; sq.y += acc[o].y * acc[o].y;
        }
    }
    float* pp = part + (((long)b * gridDim.y + blockIdx.y) * C + c) * 2;
    pp[0] = sum.x; pp[1] = sq.x;
    if (NIN == 1) { pp[2] = sum.y; pp[3] = sq.y; }
}
// (mean, rstd) per (utterance, channel) from the tile partials.  grid (ceil(C / 64), B), block 256: lane = channel,
// the four waves take every fourth tile each and are summed in wave order.  stats: [B][C][2]
__global__ __launch_bounds__(256) void in_finalize_kernel(const float* __restrict__ part, int ntile, int S, int C, float eps,
                                                          float* __restrict__ stats) {
    __shared__ double red[2][4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = min(blockIdx.x * 64 + lane, C - 1), b = blockIdx.y;
    double s = 0.0, sq = 0.0;
    for (int i = q; i < ntile; i += 4) {
        const float2 pp = *reinterpret_cast<const float2*>(part + (((long)b * ntile + i) * C + c) * 2);
        s += (double)pp.x;
        sq += (double)pp.y;
    }
    red[0][q][lane] = s;
    red[1][q][lane] = sq;
    __syncthreads();
    if (q == 0 && blockIdx.x * 64 + lane < C) {
        const double st = red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane];
        const double qt = red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane];
        const double mean = st / S, var = fmax(qt / S - mean * mean, 0.0);
        stats[((long)b * C + c) * 2] = (float)mean;
        stats[((long)b * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
__device__ __forceinline__ float in_prelu1(float v, float mean, float rstd, float g, float bta, float slope) {
    const float y = (v - mean) * rstd * g + bta;
    return y >= 0.f ? y : y * slope;
}
// out = PReLU_c(InstanceNorm(t)) elementwise.  One thread per 4 channels of one frame.
__global__ void in_prelu_kernel(const float* __restrict__ t, const float* __restrict__ stats, const float* __restrict__ g,
                                const float* __restrict__ bta, const float* __restrict__ slope, float* __restrict__ out,
                                long M, int S, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = C >> 2;
    if (i >= M * nq) return;
    const long m = i / nq;
    const int c = 4 * (int)(i % nq), b = (int)(m / S);
    const float4 v = *reinterpret_cast<const float4*>(t + m * C + c);
    const float* st = stats + ((long)b * C + c) * 2;
    float4 o;
    o.x = in_prelu1(v.x, st[0], st[1], g[c], bta[c], slope[c]);
    o.y = in_prelu1(v.y, st[2], st[3], g[c + 1], bta[c + 1], slope[c + 1]);
    o.z = in_prelu1(v.z, st[4], st[5], g[c + 2], bta[c + 2], slope[c + 2]);
    o.w = in_prelu1(v.w, st[6], st[7], g[c + 3], bta[c + 3], slope[c + 3]);
    *reinterpret_cast<float4*>(out + m * C + c) = o;
}
// seq = v * (u + PReLU_c(InstanceNorm(t2))) + n1, with u | v the two halves of uv [M][2C]
__global__ void fsmn_gate_kernel(const float* __restrict__ uv, const float* __restrict__ t2, const float* __restrict__ stats,
                                 const float* __restrict__ g, const float* __restrict__ bta, const float* __restrict__ slope,
                                 const float* __restrict__ n1, float* __restrict__ seq, long M, int S, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = C >> 2;
    if (i >= M * nq) return;
    const long m = i / nq;
    const int c = 4 * (int)(i % nq), b = (int)(m / S);
    const float4 tv = *reinterpret_cast<const float4*>(t2 + m * C + c);
    const float4 u = *reinterpret_cast<const float4*>(uv + m * 2 * C + c), v = *reinterpret_cast<const float4*>(uv + m * 2 * C + C + c);
    const float4 r = *reinterpret_cast<const float4*>(n1 + m * C + c);
    const float* st = stats + ((long)b * C + c) * 2;
    float4 o;
    o.x = v.x * (u.x + in_prelu1(tv.x, st[0], st[1], g[c], bta[c], slope[c])) + r.x;
    o.y = v.y * (u.y + in_prelu1(tv.y, st[2], st[3], g[c + 1], bta[c + 1], slope[c + 1])) + r.y;
    o.z = v.z * (u.z + in_prelu1(tv.z, st[4], st[5], g[c + 2], bta[c + 2], slope[c + 2])) + r.z;
    o.w = v.w * (u.w + in_prelu1(tv.w, st[6], st[7], g[c + 3], bta[c + 3], slope[c + 3])) + r.w;
    *reinterpret_cast<float4*>(seq + m * C + c) = o;
}

}  // namespace msf2
