// libmsf2: C-ABI entry points and the host-side launch sequence of the
// MossFormer2_SS forward pass.  All work is enqueued on the caller's stream;
// nothing here allocates, synchronises or keeps mutable global state -- except the
// opt-in kernel timer below (off by default, a measurement aid owned by one thread at a time).
#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include "msf2_common.h"
#include "msf2_gemm.h"
#include "msf2_gemm_bf16.h"
#include "msf2_gemm_bf16_astat.h"
#include "msf2_attn_fused.h"
#include "msf2_linkv_bf16.h"
#include "msf2_ops.h"

namespace msf2 {

thread_local int g_last_hip_error = 0;
thread_local bool g_bad_operand = false;

// Between msf2_timing_begin() and msf2_timing_end() every launch of the TCN pointwise
// contraction (the kernel that dominates the forward pass) is bracketed by a pair of HIP events
// recorded on the stream it is launched on, so its duration is measured where it runs, on the
// data it runs on.  Events cost a few microseconds each: never leave it on in a timed region.
// The timer belongs to the thread that called msf2_timing_begin(): launches made by other threads (a threaded
// server shares one library) are not recorded, and the event list is only touched under the mutex.
struct KernelTimer {
    std::mutex mu;
    std::atomic<bool> on{false};
    std::atomic<int> which{0};          // MSF2_TIMED_*: the kernel whose launches are bracketed
    std::thread::id owner;
    std::vector<hipEvent_t> ev;
};
static KernelTimer g_timer;
struct TimedLaunch {
    hipStream_t st;
    bool on;
    TimedLaunch(hipStream_t s, bool wanted) : st(s), on(wanted && g_timer.on.load(std::memory_order_relaxed)) { mark(); }
    ~TimedLaunch() { mark(); }
    void mark() {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_timer.mu);
        if (!g_timer.on.load() || g_timer.owner != std::this_thread::get_id() || g_timer.ev.size() >= 8192) { on = false; return; }
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
        hipEventRecord(e, st);
        g_timer.ev.push_back(e);
    }
};

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Compute units of the calling thread's current device, asked once per device.  Every grid-size threshold below (which
// tile form fills the chip) is stated in multiples of it; 256 on MI355X, which is also the fallback if the query fails.
static int device_cus() {
    constexpr int MAXDEV = 64;
    static std::atomic<int> cached[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
    return n;
}
static inline bool dim_ok(int C) { return C == 64 || C == 128 || C == 256 || C == 512; }
// the kernels address activation buffers with 32-bit byte offsets: rows * row pitch * 4 must stay below 4 GiB
static inline bool fits32(long rows, long ld) { return rows * ld * 4 < (1L << 32); }

constexpr int ENC_TS = 32;    // frames per block in the encoder / TCN stage A kernels
constexpr int TCN_TS = 32;

// Diagnostic kernel-choice switches (none changes results).  The defaults below are what the product runs with; a
// test can override them for the calling thread through msf2_debug_kernel_choice().  Nothing is read from the environment.
struct Knobs {
    int ncol_loop = 2;          // "ncol_loop": column tiles per workgroup of the 128 x 128 stream form
    int big_maxn = 4096;        // "big_maxn": widest projection that may leave that form
    bool no_big = false;        // "no_big": 128 x 128 stream form only
    int geom2 = -1;             // "geom2": 0 / 1 forces the 128 x 256 geometry off / on
    int attn_fused_min = -1;    // "attn_fused_min": fewest workgroups (2 per group and utterance) for which the fused form is taken (-1: one per CU)
    bool attn_fused = true;     // "attn_fused": bf16 mode, attention scores kept on chip (one kernel) instead of the two-kernel form
    bool own_stats = true;      // "own_stats": the small-grid forms of the two-product TCN contraction reduce stage A's partial sums themselves
    int many_min = -1;          // "many_min": fewest 128 x 128 tiles for which the 3- / 4-workgroup geometries are taken (-1: two per CU)
    int row128 = 1;             // "row128": 1 = the two-product contraction on whole-line row pieces, one column tile per workgroup; 2 = every
                                //           projection that reaches the 128 x 128 stream form; 0 = none
    int narrow_below = -1;      // "narrow_below": the small-grid forms are taken below this many 128 x 128 tiles (-1: two per CU fp32, one per CU + 1 bf16)
    bool astat = true;          // "astat": bf16 K = 512 projections with bias / activation epilogue on the register-resident-activation kernel
    bool store_aware = true;    // "store_aware": a workgroup that walks several tiles lets an epilogue's stores drain under the next K steps
    bool astat_tcn = true;      // "astat_tcn": the two-product TCN contraction on the register-resident-activation kernel
    bool astat_merge = true;    // "astat_merge": gate and proj_in of the gated TCN block in one walk of the register-resident-activation kernel
    bool astat_k1024 = true;    // "astat_k1024": the K = 1024 projection (to_out) on the register-resident-activation kernel
    bool linkv_dma = true;      // "linkv_dma": bf16 key/value summary on the global -> LDS kernel with transposing reads (0: register-staged tile kernel)
    int stage_a_split_below = -1;  // "stage_a_split_below": TCN stage A runs its quarter-frames form below this many blocks (-1: default rule)
    bool narrow = true;         // "narrow": 64-column tiles for fp32 projections whose 128-wide tiling cannot fill the chip
    char g3[16] = "a";          // "geom3" / "geom4": projections that take the 3- / 4-workgroup geometries,
    char g4[16] = "0";          //   'a' + EPI for the token-shifted operand, '0' + EPI otherwise
};
static thread_local Knobs g_knobs;

struct Ctx {
    hipStream_t st;
    int B, S, T;
    long M;
    int prec = MSF2_PREC_FP32;
    Knobs k = g_knobs;
    int cus = device_cus();     // compute units of the device the launches go to
};

// Whether a projection to N columns runs in a small-grid tile form (64-column fp32 tiles / 64-row bf16 tiles): the 128 x 128
// tiling would give fewer workgroups than the chip has CUs (one utterance per call, the callers' own shape).
static bool small_grid(const Ctx& c, int N) {
    if (!c.k.narrow || (N % 128) != 0) return false;
    const long t = (long)cdiv(N, 128) * cdiv(c.S, 128) * c.B;
    if (c.k.narrow_below >= 0) return t < c.k.narrow_below;
    if (c.prec == MSF2_PREC_BF16) return t < c.cus + 1;
    // fp32: a CU works through its share of the tiles at about one 128 x 128 tile per unit of time whether one or two are
    // resident, so the wide form costs ceil(t / CUs) units and the half-size form ceil(2 t / CUs) / 2 at ~6 % less per-tile
    // efficiency; at or below one tile per CU the half-size form also wins, by having two workgroups per CU to overlap.
    // Measured at batch 1: 4 s, 8 s (t <= CUs), 10 s and 12 s (316 / 376 tiles for N = 512: 2 units against 1.5) faster in
    // the small-grid form by 3..11 % per forward; 15 s (472 tiles: 2 against 2) slower by 3 %; N = 2176 at 4 s (544 tiles:
    // 3 against 2.5) faster.
    if (t <= c.cus) return true;
    if (t >= 16L * c.cus) return false;
    return 0.53 * (double)cdiv(2 * t, (long)c.cus) < (double)cdiv(t, (long)c.cus);
}

// launch the fp32 or the bf16 build of a contraction
template <class P>
static void launch_gemm(const Ctx& c, dim3 grid, const typename P::Args& a) {
    if (c.prec == MSF2_PREC_BF16) hipLaunchKernelGGL(gemm_kernel_bf16<P>, grid, dim3(256), 0, c.st, a);
    else hipLaunchKernelGGL(gemm_kernel<P>, grid, dim3(256), 0, c.st, a);
}

// ---- projection launcher ----------------------------------------------------
// Returns the width (columns) of the slabs over which the launched kernel writes partial row sums of squares to
// a.sq_part ([M][ceil(N / width)]), or 0 when a.sq_part is null or the kernel chosen does not write them.
template <int AMODE, int EPI>
static int launch_proj(const Ctx& c, const ProjArgs& a) {
    typedef ProjNT<AMODE, EPI> P;
    dim3 grid(cdiv(a.N, P::PAIRED ? BN / 2 : BN), cdiv(c.S, BM), c.B);
    // bf16 mode with a bf16 image of the A operand(s): both operands stream global -> LDS
    // the epilogue moves 4 consecutive columns per lane as one vector
    auto al = [](const void* p, unsigned mask) { return (reinterpret_cast<uintptr_t>(p) & mask) == 0; };
    auto al16 = [&](const void* p) { return al(p, 15); };
    if ((a.N & 3) || (a.ldo & 3) || (a.ldr & 3) || !al(a.out, 15) || !al(a.out2, 15) || !al(a.res, 15) || !al(a.pe, 15) ||
        !al(a.gmul, 15) || !al(a.bias, 15) || !al(a.bias2, 15) || !al(a.c1, 15) || !al(a.c2, 15) || !al(a.out16, 7) ||
        !al(a.res16, 7)) {
        g_bad_operand = true;
        return 0;
    }
    constexpr int timed_id = P::NSEG == 2 ? MSF2_TIMED_TCN_POINTWISE : (AMODE == A_SHIFT ? MSF2_TIMED_HIDDEN_PROJ : -1);
    const TimedLaunch timed_(c.st, timed_id >= 0 && timed_id == g_timer.which.load(std::memory_order_relaxed));
    // (the two-product form of the DMA kernel streams the bf16 residual with the operands, reads the column vector
    // with the statistics folded in and writes bf16 only: it needs res16, c1f and out16 without out)
    if (c.prec == MSF2_PREC_BF16 && a.A16 && (AMODE != A_SHIFT || (a.K % (2 * BK) == 0 && a.K <= 2048)) && (P::NSEG == 1 || (a.A2_16 && a.res16 && (a.c1f || a.stat_part) && a.out16 && !a.out && (EPI != EPI_TCN_G || a.gmul16)))) {
        // One workgroup walks 2 column tiles of its row tile: its ring then prefetches the second tile
        // under the epilogue of the first, while the partner workgroup on the same XCD (other two column
        // tiles) reads the same A rows at the same time, so they are fetched from HBM once.  Walking all
        // 4 tiles re-reads A from the fabric 4 times (FETCH_SIZE 2.4 GB instead of 0.9 GB per launch).
        int loop = c.k.ncol_loop;
        ProjArgs al = a;
        al.store_aware = c.k.store_aware ? 1 : 0;
        if constexpr (EPI == EPI_NORM && AMODE == A_PLAIN) {
            // the same kernel with the folded-LayerNorm epilogue (per-row statistics, no positional term, bf16 output only):
            // proj_in of the gated TCN block
            const long wgs = (long)cdiv(c.S, 128) * c.B;
            if (c.k.astat && a.K == 512 && a.N % 128 == 0 && a.N <= 2048 && a.out16 && !a.out && !a.out2 && !a.sq_part && a.rowstat &&
                !a.bstat && !a.pe && a.act == 0 && a.c1 && a.c2 && (a.lda & 7) == 0 && (a.ldo & 7) == 0 && al16(a.A16) && al16(a.out16) &&
                al16(a.W) && wgs >= c.cus) {
                hipLaunchKernelGGL((proj_astat_bf16_kernel<A_PLAIN, ACT_NORM>), dim3(cdiv(c.S, 128), c.B), dim3(256), 0, c.st, al);
                return 0;
            }
        }
        if constexpr (EPI == EPI_RS_BIAS_ACT) {
            // Activation rows resident in registers, weights streamed: one 4-wave workgroup per 128 rows walks every column
            // tile.  Needs a workgroup per CU at least (fewer: the tile forms below spread over the chip better), a bf16-only
            // output and K = 512 (a wave holds 64 rows x 512 k as 256 registers).
            const long wgs = (long)cdiv(c.S, 128) * c.B;
            if (c.k.astat && a.K == 512 && a.N % 128 == 0 && a.N <= 4096 && a.out16 && !a.out && !a.out2 && (a.act == 1 || a.act == 2) &&
                (a.lda & 7) == 0 && (a.ldo & 7) == 0 && al16(a.A16) && al16(a.out16) && al16(a.W) && wgs >= c.cus) {
                const dim3 g(cdiv(c.S, 128), c.B);
                if (a.act == 1) hipLaunchKernelGGL((proj_astat_bf16_kernel<AMODE, 1>), g, dim3(256), 0, c.st, al);
                else hipLaunchKernelGGL((proj_astat_bf16_kernel<AMODE, 2>), g, dim3(256), 0, c.st, al);
                return 0;
            }
            // K = 1024 (to_out of a FLASH layer): a wave holds 32 rows x 1024 k
            if constexpr (AMODE == A_PLAIN) {
                if (c.k.astat && c.k.astat_k1024 && a.K == 1024 && a.N % 128 == 0 && a.N <= 4096 && a.out16 && !a.out && !a.out2 && a.act == 1 &&
                    (a.lda & 7) == 0 && (a.ldo & 7) == 0 && al16(a.A16) && al16(a.out16) && al16(a.W) && wgs >= c.cus) {
                    hipLaunchKernelGGL((proj_astat_bf16_kernel<A_PLAIN, 1, 1024>), dim3(cdiv(c.S, 128), c.B), dim3(256), 0, c.st, al);
                    return 0;
                }
            }
        }
        if constexpr (!P::PAIRED && AMODE == A_PLAIN) {
            // fewer 128-row tiles than CUs (one utterance per call): 64-row tiles, twice the workgroups
            if (small_grid(c, a.N)) {
                typedef DmaGeom<P, 7> G7;
                dim3 g(cdiv(a.N, G7::BNX), cdiv(c.S, G7::BMX), c.B);
                al.ncol_loop = 1;
                // at most one workgroup per CU: whole-line row pieces and a deeper ring (one workgroup fills a CU's LDS)
                if (c.k.row128 && a.K % 128 == 0 && (long)g.x * g.y * g.z <= c.cus) {
                    hipLaunchKernelGGL((gemm_kernel_bf16_dma<P, 9>), g, dim3(G7::THREADS), 0, c.st, al);
                    return a.sq_part ? G7::TN * 32 : 0;
                }
                if (c.k.row128 && a.K % 128 == 0) {           // (two per CU: the 2-slot ring; 2.5 % at 2 x 4 s)
                    hipLaunchKernelGGL((gemm_kernel_bf16_dma<P, 10>), g, dim3(G7::THREADS), 0, c.st, al);
                    return a.sq_part ? G7::TN * 32 : 0;
                }
                hipLaunchKernelGGL((gemm_kernel_bf16_dma<P, 7>), g, dim3(G7::THREADS), 0, c.st, al);
                return a.sq_part ? G7::TN * 32 : 0;
            }
        }
        if constexpr (!P::PAIRED && P::NSEG == 1) {
            // 256-row tiles, one 8-wave workgroup per CU, when that still gives every CU several tiles
            typedef DmaGeom<P, 1> G;
            dim3 gb(cdiv(a.N, G::BNX), cdiv(c.S, G::BMX), c.B);
            // measured at 64 x 4 s: 8 % faster for the single-product N = 512 projections, 5 % for the N = 2176 one
            // (its ninth 256-column tile half empty); the two-product TCN contraction keeps the 128 x 128 form
            const bool want = a.N <= c.k.big_maxn && !c.k.no_big;
            // 128 x 256 tiles, two workgroups per CU, where the epilogue weighs most (residual read + two writes; the
            // widest output): measured 6 % / 2 % faster than the 256 x 256 form there, equal elsewhere
            const bool geom2 = c.k.geom2 >= 0 ? c.k.geom2 != 0 : (EPI == EPI_BIAS_RES || a.N > 1024);
            // 128 x 128 tiles with a short ring, three or four workgroups per CU: more waves to cover the copy latency than
            // the large tiles have.  Measured at 64 x 4 s: the N = 2176 projection 7 % faster with three per CU (four spill
            // there), the plain N = 512 projections 12 % faster with four; the others equal or slower.
            // the "geom3" / "geom4" choices name the kernels that take them ('a' + EPI for token-shifted A, '0' + EPI otherwise).
            const char* g3 = c.k.g3;
            const char* g4 = c.k.g4;
            const char key = (AMODE ? 'a' : '0') + EPI;
            const bool many = (long)cdiv(a.N, 128) * cdiv(c.S, 128) * c.B >= (c.k.many_min >= 0 ? c.k.many_min : 2 * c.cus);
            if (want && many && strchr(g4, key)) {
                typedef DmaGeom<P, 4> G4;
                dim3 g(cdiv(a.N, G4::BNX), cdiv(c.S, G4::BMX), c.B);
                al.ncol_loop = 1;
                hipLaunchKernelGGL((gemm_kernel_bf16_dma<P, 4>), g, dim3(G4::THREADS), 0, c.st, al);
                return a.sq_part ? G4::TN * 32 : 0;
            }
            if (want && many && strchr(g3, key)) {
                typedef DmaGeom<P, 3> G3;
                dim3 g(cdiv(a.N, G3::BNX), cdiv(c.S, G3::BMX), c.B);
                al.ncol_loop = 1;
                hipLaunchKernelGGL((gemm_kernel_bf16_dma<P, 3>), g, dim3(G3::THREADS), 0, c.st, al);
                return a.sq_part ? G3::TN * 32 : 0;
            }
            if (want && geom2 && (long)cdiv(a.N, 256) * cdiv(c.S, 128) * c.B >= 4 * c.cus) {
                typedef DmaGeom<P, 2> G2;
                dim3 g2(cdiv(a.N, G2::BNX), cdiv(c.S, G2::BMX), c.B);
                al.ncol_loop = 1;
                hipLaunchKernelGGL((gemm_kernel_bf16_dma<P, 2>), g2, dim3(G2::THREADS), 0, c.st, al);
                return a.sq_part ? G2::TN * 32 : 0;
            }
            if (want && (long)gb.x * gb.y * gb.z >= 4 * c.cus) {
                al.ncol_loop = 1;                 // 256 columns per tile already
                hipLaunchKernelGGL((gemm_kernel_bf16_dma<P, 1>), gb, dim3(G::THREADS), 0, c.st, al);
                return a.sq_part ? G::TN * 32 : 0;
            }
        }
        if (a.stat_part) { g_bad_operand = true; return 0; }      // only the small-grid forms reduce the statistics themselves
        if constexpr (EPI == EPI_TCN) {
            // the two-product contraction with h and t resident in registers (32 rows of each per wave), weights streamed
            const long wgs = (long)cdiv(c.S, 128) * c.B;
            if (c.k.astat && c.k.astat_tcn && a.K == 512 && a.N == 512 && a.res16 == a.A16 && a.ldr == a.lda && a.c1f && a.bstat && a.out16 &&
                (a.lda & 7) == 0 && (a.ldo & 7) == 0 && al16(a.A16) && al16(a.A2_16) && al16(a.out16) && al16(a.W) && al16(a.W2) &&
                al16(a.c1f) && wgs >= c.cus) {
                hipLaunchKernelGGL((proj_astat_bf16_kernel<A_PLAIN, ACT_TCN>), dim3(cdiv(c.S, 128), c.B), dim3(256), 0, c.st, al);
                return 0;
            }
        }
        if constexpr (!P::PAIRED) {
            // Whole-line row pieces, one column tile per workgroup: measured at 64 / 16 / 4 x 4 s, the two-product contraction
            // 6-9 % faster than the stream form (fabric reads 1.33 -> 0.84 GB per launch: the column tiles of a row tile
            // share their activation pieces in L2); slower below two workgroups per CU (its ring holds one tile ahead).
            // Single-product projections measure faster in the geometries chosen above.
            if ((c.k.row128 == 2 || (c.k.row128 == 1 && P::NSEG == 2)) && a.K % 128 == 0 && a.N % 128 == 0 &&
                (long)cdiv(a.N, 128) * cdiv(c.S, 128) * c.B >= 2 * c.cus) {
                typedef DmaGeom<P, 8> G8;
                dim3 g(cdiv(a.N, G8::BNX), cdiv(c.S, G8::BMX), c.B);
                al.ncol_loop = 1;
                hipLaunchKernelGGL((gemm_kernel_bf16_dma<P, 8>), g, dim3(G8::THREADS), 0, c.st, al);
                return a.sq_part ? G8::TN * 32 : 0;
            }
        }
        if (loop > (int)grid.x) loop = grid.x;
        while (loop > 1 && (long)cdiv(grid.x, loop) * grid.y * grid.z < 4 * c.cus) loop = (loop + 1) / 2;
        al.ncol_loop = loop;
        grid.x = cdiv(grid.x, loop);
        hipLaunchKernelGGL((gemm_kernel_bf16_dma<P, 0>), grid, dim3(256), 0, c.st, al);
        return a.sq_part ? DmaGeom<P, 0>::TN * 32 : 0;
    }
    if (EPI == EPI_TCN_G && !a.gmul) { g_bad_operand = true; return 0; }     // only the DMA kernel reads the bf16 gate
    if constexpr (!P::PAIRED) {
        // fp32, fewer 128-wide tiles than CUs (one utterance per call): 64-wide tiles, twice the workgroups
        if (c.prec != MSF2_PREC_BF16 && small_grid(c, a.N)) {
            grid.x = cdiv(a.N, 64);
            hipLaunchKernelGGL((gemm_kernel<P, 64>), grid, dim3(256), 0, c.st, a);
            return a.sq_part ? 64 / P::WC : 0;
        }
    }
    if (a.stat_part) { g_bad_operand = true; return 0; }
    launch_gemm<P>(c, grid, a);
    // the fp32 kernel finishes tiles with the row-major epilogue (64-column wave slabs); the register-staged bf16 one does not
    return (a.sq_part && c.prec != MSF2_PREC_BF16) ? BN / P::WC : 0;
}

static ProjArgs proj_args(const float* A, int lda, const float* W, int K, int N, int S, float* out, int ldo) {
    ProjArgs a = {};
    a.A = A; a.lda = lda; a.W = W; a.K = K; a.N = N; a.S = S; a.out = out; a.ldo = ldo;
    return a;
}

static void launch_stats_finalize(const Ctx& c, const float* part, int nblk, long count_per_b, float eps, float* stats,
                                  const float* c1 = nullptr, const float* c2 = nullptr, float* c1f = nullptr, int C = 0) {
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(c.B), dim3(64), 0, c.st, part, nblk, (double)count_per_b, eps, stats, c1, c2, c1f, C);
}

template <int MODE>
static void launch_rowstat(const Ctx& c, const float* x, int ldx, float* out, int C, int shift, float eps) {
    hipLaunchKernelGGL(rowstat_kernel<MODE>, dim3(cdiv(c.M, 4)), dim3(256), 0, c.st, x, ldx, out, c.M, c.S, C, shift, eps);
}

// ln_out: optional [M][2] LayerNorm statistics (mean, rstd; eps 1e-5) of the result rows, written by the same kernel
// when a row fits one block (C <= 512).  Returns whether they were written.
// x16: the input is read as bf16 from there instead of fp32 from x (only together with ln_out and C <= 512).
static bool launch_dwconv17(const Ctx& c, const float* x, int ldx, const float* dw, float* y, int ldy, int C, int accumulate,
                            unsigned short* y16 = nullptr, float* ln_out = nullptr, const unsigned short* x16 = nullptr) {
    dim3 grid(cdiv(C / 4, 128), cdiv(c.S, DW_TS), c.B);
    if (x16) {
        if (!ln_out || grid.x != 1) { g_bad_operand = true; return false; }
        hipLaunchKernelGGL((dwconv17_kernel<true, true>), grid, dim3(128), 0, c.st, reinterpret_cast<const float*>(x16), ldx, dw, y, ldy,
                           c.S, C, accumulate, y16, ln_out, 1e-5f);
        return true;
    }
    if (ln_out && grid.x == 1) {
        hipLaunchKernelGGL(dwconv17_kernel<true>, grid, dim3(128), 0, c.st, x, ldx, dw, y, ldy, c.S, C, accumulate, y16, ln_out, 1e-5f);
        return true;
    }
    hipLaunchKernelGGL(dwconv17_kernel<false>, grid, dim3(128), 0, c.st, x, ldx, dw, y, ldy, c.S, C, accumulate, y16, (float*)nullptr, 0.f);
    return false;
}

static void launch_dwconv17_bf16(const Ctx& c, const unsigned short* x, int ldx, const float* dw, unsigned short* y, int ldy, int C) {
    dim3 grid(cdiv(C / 4, 128), cdiv(c.S, DW16_TS), c.B);
    hipLaunchKernelGGL(dwconv17_bf16_kernel, grid, dim3(128), 0, c.st, x, ldx, dw, y, ldy, c.S, C);
}

// ---- FLASH attention core ---------------------------------------------------
struct AttnWs {
    float *q4, *P, *part, *kv, *sqp;     // sqp: [M][E/64] partial sums of o^2
    int nsplit, chunk, G;
};
static int attn_nsplit(int B, int S) {
    int n = cdiv(1024, 16L * B);
    const int maxn = cdiv(S, 256);
    if (n > maxn) n = maxn;
    if (n < 1) n = 1;
    return n;
}
static AttnWs attn_ws(Arena& ar, int B, int S, int E) {
    AttnWs w;
    w.G = cdiv(S, MSF2_GROUP);
    w.nsplit = attn_nsplit(B, S);
    w.chunk = cdiv(cdiv(S, w.nsplit), BK) * BK;
    w.nsplit = cdiv(S, w.chunk);
    w.q4 = ar.take<float>(4L * B * S * MSF2_QK_DIM);
    w.P = ar.take<float>((long)B * w.G * MSF2_GROUP * MSF2_GROUP);
    w.part = ar.take<float>((long)B * w.nsplit * MSF2_QK_DIM * 2 * E);
    w.kv = ar.take<float>((long)B * MSF2_QK_DIM * 2 * E);
    w.sqp = ar.take<float>((long)B * S * (E / 32));     // partial sums of o^2: 64-column slabs, 32-column ones from the fused kernel
    return w;
}
// o_rnorm: optional [M] ScaleNorm inverse norms of the output rows (from partial sums the output kernel writes)
static void run_attention(const Ctx& c, const AttnWs& w, const float* hu, int ldhu, const float* gamma,
                          const float* beta, const float* freqs, float* o, int E, unsigned short* o16 = nullptr,
                          const unsigned short* hu16 = nullptr, float* o_rnorm = nullptr) {
    // hu16 (bf16 mode): the hidden tensor [v | u | qk] is read as bf16 from there instead of fp32 from hu
    const long M = c.M;
    if (hu16)
        hipLaunchKernelGGL(qk_prep_bf16_kernel, dim3(cdiv(M, 32)), dim3(256), 0, c.st, hu16 + 2 * E, ldhu, gamma, beta, freqs, w.q4, M, c.S);
    else
        hipLaunchKernelGGL(qk_prep_kernel<false>, dim3(cdiv(M, 4)), dim3(256), 0, c.st, (const void*)(hu + 2 * E), ldhu, gamma, beta, freqs, w.q4, M, c.S);
    const float* hsrc = hu16 ? reinterpret_cast<const float*>(hu16) : hu;
    const float* quad_q = w.q4;
    const float* lin_q = w.q4 + 1 * M * MSF2_QK_DIM;
    const float* quad_k = w.q4 + 2 * M * MSF2_QK_DIM;
    const float* lin_k = w.q4 + 3 * M * MSF2_QK_DIM;
    {
        LinKVArgs a = {lin_k, hsrc, ldhu, w.part, c.S, 2 * E, w.chunk};
        if (hu16 && c.k.linkv_dma && (2 * E) % 256 == 0) {     // bf16 mode: tiles global -> LDS as they lie, both fragments through the transposing LDS read
            LinKV16Args a16 = {reinterpret_cast<const unsigned short*>(lin_k), hu16, ldhu, w.part, c.S, 2 * E, w.chunk};
            hipLaunchKernelGGL(linkv_bf16_kernel, dim3(2 * E / 256, w.nsplit, c.B), dim3(512), 0, c.st, a16);
        } else if (hu16) launch_gemm<LinKVT<true>>(c, dim3(cdiv(2 * E, BN), w.nsplit, c.B), a);
        else launch_gemm<LinKV>(c, dim3(cdiv(2 * E, BN), w.nsplit, c.B), a);
        const long per_b = (long)MSF2_QK_DIM * 2 * E, total = per_b * c.B;
        hipLaunchKernelGGL(kv_finalize_kernel, dim3(cdiv(total / 4, 256)), dim3(256), 0, c.st, w.part, w.kv, w.nsplit, per_b, c.S, total);
    }
    if (hu16 && c.k.attn_fused && (E % 64) == 0 && 2L * c.B * w.G >= (c.k.attn_fused_min >= 0 ? c.k.attn_fused_min : c.cus)) {
        // bf16 mode: scores, both products and the gating in one kernel; the scores never leave the chip
        // (one workgroup per half group and CU: below about one workgroup per CU the two-kernel form, whose
        // grid is E / 64 times larger, fills the chip better -- measured equal at 256 workgroups, 6 % slower at 128)
        AttnFusedArgs a = {reinterpret_cast<const unsigned short*>(quad_q), reinterpret_cast<const unsigned short*>(quad_k),
                           reinterpret_cast<const unsigned short*>(lin_q), hu16, ldhu, w.kv, o16, o_rnorm ? w.sqp : nullptr, c.S, w.G, E};
        if (!o16) { g_bad_operand = true; return; }
        hipLaunchKernelGGL(attn_fused_bf16_kernel, dim3(2, c.B * w.G), dim3(512), 0, c.st, a);
        if (o_rnorm)
            hipLaunchKernelGGL(rowsq_finalize_kernel, dim3(cdiv(M, 256)), dim3(256), 0, c.st, w.sqp, E / 32, o_rnorm, M, c.S, E, 0);
        return;
    }
    {
        // bf16 mode: scores and lin_q exist as bf16 only (the values the contraction would round them to anyway)
        SimArgs a = {quad_q, quad_k, w.P, c.S, w.G, hu16 ? reinterpret_cast<unsigned short*>(w.P) : nullptr};
        if (hu16) launch_gemm<AttnSimT<true>>(c, dim3(2, 2, c.B * w.G), a);
        else launch_gemm<AttnSim>(c, dim3(2, 2, c.B * w.G), a);
    }
    {
        AttnOutArgs a = {w.P, lin_q, hsrc, ldhu, w.kv, o, c.S, w.G, E, o16,
                         reinterpret_cast<const unsigned short*>(w.P), reinterpret_cast<const unsigned short*>(lin_q),
                         o_rnorm ? w.sqp : nullptr};
        if (hu16) launch_gemm<AttnOutT<true>>(c, dim3(E / (BN / 2), 2, c.B * w.G), a);
        else launch_gemm<AttnOut>(c, dim3(E / (BN / 2), 2, c.B * w.G), a);
        if (o_rnorm)
            hipLaunchKernelGGL(rowsq_finalize_kernel, dim3(cdiv(M, 256)), dim3(256), 0, c.st, w.sqp, E / (BN / 2), o_rnorm, M, c.S, E, 0);
    }
}

// ---- gated TCN block ----------------------------------------------------------
struct TcnWs {
    float *rowstat, *g, *h0, *h1, *t, *part, *bstat, *c1f;
    unsigned short *g16, *h16a, *h16b, *t16;     // bf16 mode: the gate and the operand images of h and t
    int nblk;
};
static TcnWs tcn_ws(Arena& ar, int B, int S, int C, int prec) {
    TcnWs w = {};
    const long M = (long)B * S;
    w.nblk = cdiv(S, TCN_TS);
    w.rowstat = ar.take<float>(2 * M);
    if (prec != MSF2_PREC_BF16) {
        w.g = ar.take<float>(M * C);
        w.h0 = ar.take<float>(M * C);
        w.h1 = ar.take<float>(M * C);
    }
    if (prec == MSF2_PREC_BF16) {
        w.c1f = ar.take<float>((long)B * C);
        w.g16 = ar.take<unsigned short>(M * C);
        w.h16a = ar.take<unsigned short>(M * C);
        w.h16b = ar.take<unsigned short>(M * C);
        w.t16 = ar.take<unsigned short>(M * C);
    } else {
        w.t = ar.take<float>(M * C);
    }
    w.part = ar.take<float>(2L * B * (4 * w.nblk + (1 << MSF2_TCN_DEPTH) + 4));     // (stage A's block count grows with the dilation; 4x the blocks in its small-grid form)
    w.bstat = ar.take<float>(2L * B);
    return w;
}

// Frames per thread of a TCN stage A launch.  `full` frames (32 per block) normally; a quarter of that when the launch
// would give less than one block per CU (one short utterance per call): four times the blocks, each thread
// re-reading two neighbour rows per 2..4 frames instead of per 8..16 -- from a tensor that then fits the caches.
static int stage_a_frames_per_thread(const Ctx& c, int full, int rows_per_block) {
    const long blocks = (long)cdiv(c.S, full * rows_per_block) * c.B;
    const long below = c.k.stage_a_split_below >= 0 ? c.k.stage_a_split_below : c.cus;   // measured at batch 1: faster up to 6 s (188 blocks), even at 8 s (250), slower from 10 s (313)
    return (c.k.narrow && blocks < below && full >= 4) ? full / 4 : full;
}

// returns the number of partial-sum entries (blocks per utterance) the launch wrote
static int launch_stage_a_bf16(const Ctx& c, const TcnWs& w, const unsigned short* h16, const msf2_tcn_weights& tw, int C, int dil) {
    // frames per thread NIT = 32 / (256 / (C / 8)); a block holds 256 / (C / 8) thread rows of NIT frames, dil apart
    const int rows_per_block = 256 / (C / 8);
    const int nit = stage_a_frames_per_thread(c, 32 / rows_per_block, rows_per_block);
    const int nblk = cdiv(cdiv(c.S, nit * dil) * dil, rows_per_block);
    const dim3 grid(nblk, c.B), blk(256);
    switch (nit) {
        case 8:  hipLaunchKernelGGL(tcn_stage_a_bf16_kernel<8>, grid, blk, 0, c.st, h16, tw.dw_w, tw.dw_b, w.t16, w.part, c.S, C, dil); break;
        case 4:  hipLaunchKernelGGL(tcn_stage_a_bf16_kernel<4>, grid, blk, 0, c.st, h16, tw.dw_w, tw.dw_b, w.t16, w.part, c.S, C, dil); break;
        case 2:  hipLaunchKernelGGL(tcn_stage_a_bf16_kernel<2>, grid, blk, 0, c.st, h16, tw.dw_w, tw.dw_b, w.t16, w.part, c.S, C, dil); break;
        default: hipLaunchKernelGGL(tcn_stage_a_bf16_kernel<1>, grid, blk, 0, c.st, h16, tw.dw_w, tw.dw_b, w.t16, w.part, c.S, C, dil); break;
    }
    return nblk;
}

// bf16 mode: inside the block h, t and g*h live in bf16 only (contraction operands, depthwise input
// and the 8-step residual chain of h); statistics, gates and the outer residual stream x stay fp32.
// x16: optional bf16 image of x (read by the two projections of x, rewritten with the block's result).
// sq_part: optional [M][..] partial row sums of squares of the block's result (see launch_proj); returns their slab width
static int run_gated_tcn_bf16(const Ctx& c, const TcnWs& w, float* x, unsigned short* x16, const msf2_layer_weights& lw, int C, bool have_ln,
                              float* sq_part) {
    if (!have_ln) launch_rowstat<1>(c, x, C, w.rowstat, C, 0, 1e-5f);
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (c.k.astat && c.k.astat_merge && x16 && C == 512 && (long)cdiv(c.S, 128) * c.B >= c.cus && al16(x16) && al16(w.g16) && al16(w.h16a) &&
        al16(lw.w_gg) && al16(lw.w_in)) {
        // the gate and proj_in read the same rows: one walk of the register-resident-activation kernel over both weight matrices
        ProjArgs a = proj_args(x, C, lw.w_gg, C, C, c.S, nullptr, C);
        a.W2 = lw.w_in; a.bias = lw.b_gg; a.rowstat = w.rowstat; a.c1 = lw.c1_in; a.c2 = lw.c2_in; a.A16 = x16;
        a.out16 = w.g16; a.out16b = w.h16a;
        hipLaunchKernelGGL((proj_astat_bf16_kernel<A_PLAIN, ACT_GATE_NORM>), dim3(cdiv(c.S, 128), c.B), dim3(256), 0, c.st, a);
    } else {
        {
            ProjArgs a = proj_args(x, C, lw.w_gg, C, C, c.S, nullptr, C);      // the gate exists as bf16 only
            a.bias = lw.b_gg; a.act = 2; a.A16 = x16; a.out16 = w.g16;
            launch_proj<A_PLAIN, EPI_RS_BIAS_ACT>(c, a);
        }
        {
            ProjArgs a = proj_args(x, C, lw.w_in, C, C, c.S, nullptr, C);
            a.rowstat = w.rowstat; a.c1 = lw.c1_in; a.c2 = lw.c2_in; a.out16 = w.h16a; a.A16 = x16;
            launch_proj<A_PLAIN, EPI_NORM>(c, a);
        }
    }
    unsigned short* h16 = w.h16a;
    unsigned short* hn16 = w.h16b;
    const bool own_stats = c.k.own_stats && small_grid(c, C);
    for (int j = 0; j < MSF2_TCN_DEPTH; ++j) {
        const msf2_tcn_weights& tw = lw.tcn[j];
        const int nb = launch_stage_a_bf16(c, w, h16, tw, C, 1 << j);
        ProjArgs a = proj_args(nullptr, C, tw.w_gate, C, C, c.S, nullptr, C);
        a.A16 = h16; a.A2_16 = w.t16; a.W2 = tw.w_pw; a.bias = tw.b_gate; a.c1 = tw.c1; a.c2 = tw.c2;
        a.res16 = h16; a.ldr = C; a.out16 = hn16;
        if (own_stats && nb <= 64 * ProjNT<A_PLAIN, EPI_TCN>::STAT_NL) {        // one launch fewer per block where launches are what the block costs
            a.stat_part = w.part; a.stat_nblk = nb; a.stat_eps = 1e-5f; a.stat_count = (double)c.S * C;
        } else {
            launch_stats_finalize(c, w.part, nb, (long)c.S * C, 1e-5f, w.bstat, tw.c1, tw.c2, w.c1f, C);
            a.bstat = w.bstat; a.c1f = w.c1f;
        }
        if (j == MSF2_TCN_DEPTH - 1) {               // the last block hands g * h to proj_out
            a.gmul16 = w.g16;
            launch_proj<A_PLAIN, EPI_TCN_G>(c, a);
        } else
            launch_proj<A_PLAIN, EPI_TCN>(c, a);
        unsigned short* t16 = h16; h16 = hn16; hn16 = t16;
    }
    {
        ProjArgs a = proj_args(nullptr, C, lw.w_po, C, C, c.S, x, C);
        a.A16 = h16; a.bias = lw.b_po; a.res = x; a.ldr = C; a.out16 = x16; a.sq_part = sq_part;
        return launch_proj<A_PLAIN, EPI_BIAS_RES>(c, a);
    }
}

// have_ln: w.rowstat already holds the LayerNorm statistics of x
static int run_gated_tcn(const Ctx& c, const TcnWs& w, float* x, unsigned short* x16, const msf2_layer_weights& lw, int C, bool have_ln = false,
                         float* sq_part = nullptr) {
    if (c.prec == MSF2_PREC_BF16) return run_gated_tcn_bf16(c, w, x, x16, lw, C, have_ln, sq_part);
    // g = sigmoid(x Wgg^T + b)
    {
        ProjArgs a = proj_args(x, C, lw.w_gg, C, C, c.S, w.g, C);
        a.bias = lw.b_gg; a.act = 2;
        launch_proj<A_PLAIN, EPI_RS_BIAS_ACT>(c, a);
    }
    // h = LayerNorm(x) Win^T + b   (normalisation folded into the epilogue)
    if (!have_ln) launch_rowstat<1>(c, x, C, w.rowstat, C, 0, 1e-5f);
    {
        ProjArgs a = proj_args(x, C, lw.w_in, C, C, c.S, w.h0, C);
        a.rowstat = w.rowstat; a.c1 = lw.c1_in; a.c2 = lw.c2_in;
        launch_proj<A_PLAIN, EPI_NORM>(c, a);
    }
    float* h = w.h0;
    float* hn = w.h1;
    const bool own_stats = c.k.own_stats && small_grid(c, C);
    for (int j = 0; j < MSF2_TCN_DEPTH; ++j) {
        const msf2_tcn_weights& tw = lw.tcn[j];
        int nb;
        {
            // frames per thread NIT = 32 / (256 / (C / 4)); a block holds 256 / (C / 4) thread rows of NIT frames, dil apart
            const int dil = 1 << j, rows_per_block = 256 / (C / 4);
            const int nit = stage_a_frames_per_thread(c, 32 / rows_per_block, rows_per_block);
            nb = cdiv(cdiv(c.S, nit * dil) * dil, rows_per_block);
            const dim3 grid(nb, c.B), blk(256);
            switch (nit) {
                case 16: hipLaunchKernelGGL(tcn_stage_a_kernel<16>, grid, blk, 0, c.st, h, tw.dw_w, tw.dw_b, w.t, w.part, c.S, C, dil); break;
                case 8:  hipLaunchKernelGGL(tcn_stage_a_kernel<8>, grid, blk, 0, c.st, h, tw.dw_w, tw.dw_b, w.t, w.part, c.S, C, dil); break;
                case 4:  hipLaunchKernelGGL(tcn_stage_a_kernel<4>, grid, blk, 0, c.st, h, tw.dw_w, tw.dw_b, w.t, w.part, c.S, C, dil); break;
                case 2:  hipLaunchKernelGGL(tcn_stage_a_kernel<2>, grid, blk, 0, c.st, h, tw.dw_w, tw.dw_b, w.t, w.part, c.S, C, dil); break;
                default: hipLaunchKernelGGL(tcn_stage_a_kernel<1>, grid, blk, 0, c.st, h, tw.dw_w, tw.dw_b, w.t, w.part, c.S, C, dil); break;
            }
        }
        ProjArgs a = proj_args(h, C, tw.w_gate, C, C, c.S, hn, C);
        a.A2 = w.t; a.W2 = tw.w_pw; a.bias = tw.b_gate; a.c1 = tw.c1; a.c2 = tw.c2;
        a.res = h; a.ldr = C;
        if (own_stats && nb <= 64 * ProjNT<A_PLAIN, EPI_TCN>::STAT_NL) {
            a.stat_part = w.part; a.stat_nblk = nb; a.stat_eps = 1e-5f; a.stat_count = (double)c.S * C;
        } else {
            launch_stats_finalize(c, w.part, nb, (long)c.S * C, 1e-5f, w.bstat);
            a.bstat = w.bstat;
        }
        if (j == MSF2_TCN_DEPTH - 1) {               // the last block hands g * h to proj_out
            a.gmul = w.g;
            launch_proj<A_PLAIN, EPI_TCN_G>(c, a);
        } else
            launch_proj<A_PLAIN, EPI_TCN>(c, a);
        float* tmp = h; h = hn; hn = tmp;
    }
    // x += (g * h) Wpo^T + b
    {
        ProjArgs a = proj_args(h, C, lw.w_po, C, C, c.S, x, C);
        a.bias = lw.b_po; a.res = x; a.ldr = C; a.sq_part = sq_part;
        return launch_proj<A_PLAIN, EPI_BIAS_RES>(c, a);
    }
}

// ---- FSMN-family block (Gated_FSMN_Block_Dilated) ------------------------------------
struct FsmnWs {
    float *c1, *n1, *uv_pre, *uv, *f1, *p1, *t1, *o1, *t2, *seq, *n2, *rowstat, *part, *stats;
    unsigned short *n1_16, *uv16, *f1_16, *n2_16;      // bf16 mode: operand images of the tensors the block's contractions read
    int ntile;
};
static FsmnWs fsmn_ws(Arena& ar, int B, int S, int inner, int prec = MSF2_PREC_FP32) {
    FsmnWs w = {};
    const long M = (long)B * S;
    w.ntile = cdiv(S, FS_TS);
    w.c1 = ar.take<float>(M * inner);   w.n1 = ar.take<float>(M * inner);
    w.uv_pre = ar.take<float>(M * 2 * inner);  w.uv = ar.take<float>(M * 2 * inner);
    w.f1 = ar.take<float>(M * inner);   w.p1 = ar.take<float>(M * inner);
    w.t1 = ar.take<float>(M * inner);   w.o1 = ar.take<float>(M * inner);
    w.t2 = ar.take<float>(M * inner);   w.seq = ar.take<float>(M * inner);
    w.n2 = ar.take<float>(M * inner);
    w.rowstat = ar.take<float>(2 * M);
    w.part = ar.take<float>(2L * B * (w.ntile + 2) * inner);      // (the dilated convolution's tile count rounds up per residue class)
    w.stats = ar.take<float>(2L * B * inner);
    if (prec == MSF2_PREC_BF16) {
        w.n1_16 = ar.take<unsigned short>(M * inner);   w.uv16 = ar.take<unsigned short>(M * 2 * inner);
        w.f1_16 = ar.take<unsigned short>(M * inner);   w.n2_16 = ar.take<unsigned short>(M * inner);
    }
    return w;
}
static inline bool fsmn_dims_ok(int dim, int inner) {
    return dim >= BK && (dim % BK) == 0 && dim <= 1024 && inner >= BK && (inner % BK) == 0 && inner <= 512;
}
// y = block(x); y may alias x (the last projection reads its residual and writes its result at the same element).
// The contractions follow c.prec (bf16 mode: projection matrices bf16, activations rounded on their way into LDS);
// everything else is fp32.  y16: optional bf16 image of y; sq_part: optional partial row sums of squares of y
// (see launch_proj); returns their slab width.
// x16 (bf16 mode, with the workspace laid out for it): bf16 image of x; the block's contractions then read bf16 images of
// both operands (written by the kernels that produce the tensors) and stream them global -> LDS.
static int run_gated_fsmn(const Ctx& c, const FsmnWs& w, const float* x, const msf2_fsmn_weights* fw, float* y, int dim, int inner,
                          unsigned short* y16 = nullptr, float* sq_part = nullptr, const unsigned short* x16 = nullptr) {
    const bool img = c.prec == MSF2_PREC_BF16 && x16 && w.n1_16;
    const int B = c.B, S = c.S;
    const long M = c.M;
    const int C = inner;
    {   // c1 = PReLU(x W1^T + b1)
        ProjArgs a = proj_args(x, dim, fw->w1, dim, C, S, w.c1, C);
        a.bias = fw->b1; a.act = 4; a.prelu = fw->prelu1;
        if (img) a.A16 = x16;
        launch_proj<A_PLAIN, EPI_RS_BIAS_ACT>(c, a);
    }
    hipLaunchKernelGGL(ln_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, c.st, w.c1, fw->ln1_g, fw->ln1_b, w.n1, M, C, 1e-5f,
                       img ? w.n1_16 : nullptr);
    launch_rowstat<1>(c, w.n1, C, w.rowstat, C, 0, 1e-5f);
    {   // [u | v] = silu(LayerNorm(n1) W^T + b), both FFConvM projections in one contraction
        ProjArgs a = proj_args(w.n1, C, fw->w_uv, C, 2 * C, S, w.uv_pre, 2 * C);
        a.rowstat = w.rowstat; a.c1 = fw->c1_uv; a.c2 = fw->c2_uv; a.act = 1;
        if (img) a.A16 = w.n1_16;
        launch_proj<A_PLAIN, EPI_NORM>(c, a);
    }
    launch_dwconv17(c, w.uv_pre, 2 * C, fw->dw_uv, w.uv, 2 * C, 2 * C, 0, img ? w.uv16 : nullptr);
    {   // f1 = relu(u Wl^T + b)
        ProjArgs a = proj_args(w.uv, 2 * C, fw->w_lin, C, C, S, w.f1, C);
        a.bias = fw->b_lin; a.act = 3;
        if (img) { a.A16 = w.uv16; a.out16 = w.f1_16; }
        launch_proj<A_PLAIN, EPI_RS_BIAS_ACT>(c, a);
    }
    {   // p1 = f1 Wp^T
        ProjArgs a = proj_args(w.f1, C, fw->w_proj, C, C, S, w.p1, C);
        if (img) a.A16 = w.f1_16;
        launch_proj<A_PLAIN, EPI_RS_BIAS_ACT>(c, a);
    }
    // frame tiles of the two convolutions (dilation 1 and 2): blockIdx.y = tile * dil + residue
    const int nt1 = cdiv(S, FS_TS), nt2 = cdiv(S, FS_TS * 2) * 2;
    const dim3 cgrid1(cdiv(C / 2, 64), nt1, B), cgrid2(cdiv(C, 64), nt2, B), fgrid(cdiv(C, 64), B);
    const int egrid = cdiv(M * (C / 4), 256);
    hipLaunchKernelGGL(fsmn_conv39_kernel<1>, cgrid1, dim3(64), 0, c.st, w.p1, (const float*)nullptr, fw->w_c1, w.t1, w.part, S, C, 1);
    hipLaunchKernelGGL(in_finalize_kernel, fgrid, dim3(256), 0, c.st, w.part, nt1, S, C, 1e-5f, w.stats);
    hipLaunchKernelGGL(in_prelu_kernel, dim3(egrid), dim3(256), 0, c.st, w.t1, w.stats, fw->in1_g, fw->in1_b, fw->pr1, w.o1, M, S, C);
    hipLaunchKernelGGL(fsmn_conv39_kernel<2>, cgrid2, dim3(64), 0, c.st, w.o1, w.p1, fw->w_c2, w.t2, w.part, S, C, 2);
    hipLaunchKernelGGL(in_finalize_kernel, fgrid, dim3(256), 0, c.st, w.part, nt2, S, C, 1e-5f, w.stats);
    hipLaunchKernelGGL(fsmn_gate_kernel, dim3(egrid), dim3(256), 0, c.st, w.uv, w.t2, w.stats, fw->in2_g, fw->in2_b, fw->pr2, w.n1,
                       w.seq, M, S, C);
    hipLaunchKernelGGL(ln_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, c.st, w.seq, fw->ln2_g, fw->ln2_b, w.n2, M, C, 1e-5f,
                       img ? w.n2_16 : nullptr);
    {   // y = x + n2 W2^T + b2
        ProjArgs a = proj_args(w.n2, C, fw->w2, C, dim, S, y, dim);
        if (img) a.A16 = w.n2_16;
        a.bias = fw->b2; a.res = x; a.ldr = dim; a.out16 = y16; a.sq_part = sq_part;
        return launch_proj<A_PLAIN, EPI_BIAS_RES>(c, a);
    }
}

// ---- decoder ----------------------------------------------------------------
static void run_decoder(const Ctx& c, const float* sep, const float* dec_w, float* fv, float* est, int C) {
    int grid = cdiv(c.M, 4);
    if (grid > 2048) grid = 2048;
    if (C == 512) hipLaunchKernelGGL(decoder_frames_reg_kernel<2>, dim3(grid), dim3(256), 0, c.st, sep, dec_w, fv, c.M);
    else if (C == 256) hipLaunchKernelGGL(decoder_frames_reg_kernel<1>, dim3(grid), dim3(256), 0, c.st, sep, dec_w, fv, c.M);
    else hipLaunchKernelGGL(decoder_frames_kernel, dim3(grid), dim3(256), (size_t)C * 20 * sizeof(float), c.st, sep, dec_w, fv, c.M, C);
    const long n = (long)c.B * c.T;
    hipLaunchKernelGGL(decoder_ola_kernel, dim3(cdiv(n, 256)), dim3(256), 0, c.st, fv, est, c.B, c.S, c.T);
}

// ---- whole forward ----------------------------------------------------------
struct FwdWs {
    float *enc, *skip, *x, *hu_pre, *hu, *o, *po, *rowstat, *part, *bstat, *pe, *fv, *zeros, *sqx;
    unsigned short *x16, *o16;           // bf16 mode: operand images of the residual stream and the attention output
    unsigned short *hu_pre16, *hu16;     // bf16 mode: the hidden tensor before / after its ConvModule lives in bf16 only
    AttnWs at;
    TcnWs tc;            // MSF2_REC_TCN
    FsmnWs fs;           // MSF2_REC_FSMN
    int nblk_enc, nblk_ln;
};
static FwdWs fwd_ws(Arena& ar, const msf2_config& cfg, int B, int S) {
    FwdWs w;
    const int C = cfg.dim, E = 2 * C, NH = 4 * C + MSF2_QK_DIM;
    const long M = (long)B * S;
    w.nblk_enc = cdiv(S, ENC_TS);
    w.nblk_ln = cdiv(S, 32);
    w.enc = ar.take<float>(M * C);
    w.skip = ar.take<float>(M * C);
    w.x = ar.take<float>(M * C);
    w.hu_pre = ar.take<float>(M * NH);   // reused by the mask head: z [M, 2C] and masked features [2][M, C]
    w.hu = ar.take<float>(M * NH);       // reused by the mask head: gated features [2][M, C]
    w.o = ar.take<float>(M * E);
    w.po = ar.take<float>(M * C);
    w.rowstat = ar.take<float>(2 * M);
    const int nb = w.nblk_enc > w.nblk_ln ? w.nblk_enc : w.nblk_ln;
    w.part = ar.take<float>(2L * B * nb);
    w.bstat = ar.take<float>(2L * B);
    w.pe = ar.take<float>((long)S * C);
    w.fv = ar.take<float>(M * 16);
    w.zeros = ar.take<float>(1024);      // >= K bytes: the DMA kernel's copy address of a zero row advances with k
    w.sqx = ar.take<float>((long)B * S * cdiv(C, 32));   // partial row sums of squares of x, slabs of >= 32 columns
    w.x16 = w.o16 = w.hu_pre16 = w.hu16 = nullptr;
    if (cfg.precision == MSF2_PREC_BF16) {
        w.hu_pre16 = ar.take<unsigned short>(M * NH);
        w.hu16 = ar.take<unsigned short>(M * NH);
        w.x16 = ar.take<unsigned short>(M * C);
        w.o16 = ar.take<unsigned short>(M * E);
    }
    w.at = attn_ws(ar, B, S, E);
    w.tc = TcnWs();
    w.fs = FsmnWs();
    if (cfg.recurrent == MSF2_REC_FSMN) w.fs = fsmn_ws(ar, B, S, cfg.fsmn_inner, cfg.precision);
    else w.tc = tcn_ws(ar, B, S, C, cfg.precision);
    return w;
}

static int check_cfg(const msf2_config* cfg) {
    if (!cfg) return MSF2_ERR_ARG;
    if (!dim_ok(cfg->dim) || cfg->num_layers < 0 || cfg->enc_kernel != 16 || cfg->num_spks != 2) return MSF2_ERR_ARG;
    if (cfg->precision != MSF2_PREC_FP32 && cfg->precision != MSF2_PREC_BF16) return MSF2_ERR_ARG;
    if (cfg->recurrent != MSF2_REC_TCN && cfg->recurrent != MSF2_REC_FSMN) return MSF2_ERR_ARG;
    if (cfg->recurrent == MSF2_REC_FSMN && !fsmn_dims_ok(cfg->dim, cfg->fsmn_inner)) return MSF2_ERR_ARG;
    return MSF2_OK;
}

static void copy_tap(const Ctx& c, float* dst, const float* src, long n) {
    if (dst) hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, c.st);
}

}  // namespace msf2

using namespace msf2;

extern "C" {

const char* msf2_strerror(int code) {
    switch (code) {
        case MSF2_OK: return "ok";
        case MSF2_ERR_ARG: return "invalid argument (null pointer, unsupported shape or configuration)";
        case MSF2_ERR_WORKSPACE: return "workspace too small";
        case MSF2_ERR_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown error";
    }
}
int msf2_last_hip_error(void) { return g_last_hip_error; }
int msf2_version(void) { return 5; }
int msf2_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(msf2_config);
        case 1: return (int)sizeof(msf2_tcn_weights);
        case 2: return (int)sizeof(msf2_layer_weights);
        case 3: return (int)sizeof(msf2_weights);
        case 4: return (int)sizeof(msf2_taps);
        case 5: return (int)sizeof(msf2_fsmn_weights);
        default: return -1;
    }
}
int msf2_timing_begin_kernel(int which) {
    if (which != MSF2_TIMED_TCN_POINTWISE && which != MSF2_TIMED_HIDDEN_PROJ) return MSF2_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_timer.mu);
    for (hipEvent_t e : g_timer.ev) hipEventDestroy(e);
    g_timer.ev.clear();
    g_timer.owner = std::this_thread::get_id();
    g_timer.which.store(which);
    g_timer.on.store(true);
    return MSF2_OK;
}
int msf2_timing_begin(void) { return msf2_timing_begin_kernel(MSF2_TIMED_TCN_POINTWISE); }
int msf2_timing_end(double* total_ms, int* launches) {
    std::vector<hipEvent_t> ev;
    {
        std::lock_guard<std::mutex> lk(g_timer.mu);
        g_timer.on.store(false);
        ev.swap(g_timer.ev);
    }
    double sum = 0.0;
    int n = 0, rc = MSF2_OK;
    for (size_t i = 0; i + 1 < ev.size(); i += 2) {
        float ms = 0.f;
        if (hipEventSynchronize(ev[i + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms, ev[i], ev[i + 1]) != hipSuccess) { rc = MSF2_ERR_LAUNCH; break; }
        sum += ms;
        ++n;
    }
    for (hipEvent_t e : ev) hipEventDestroy(e);
    if (total_ms) *total_ms = sum;
    if (launches) *launches = n;
    return rc;
}
int msf2_debug_kernel_choice(const char* name, const char* value) {
    if (!name) { g_knobs = Knobs(); return MSF2_OK; }
    if (!value) return MSF2_ERR_ARG;
    Knobs& k = g_knobs;
    if (!strcmp(name, "ncol_loop")) k.ncol_loop = atoi(value) > 0 ? atoi(value) : 1;
    else if (!strcmp(name, "big_maxn")) k.big_maxn = atoi(value);
    else if (!strcmp(name, "no_big")) k.no_big = atoi(value) != 0;
    else if (!strcmp(name, "geom2")) k.geom2 = atoi(value);
    else if (!strcmp(name, "narrow")) k.narrow = atoi(value) != 0;
    else if (!strcmp(name, "row128")) k.row128 = atoi(value);
    else if (!strcmp(name, "many_min")) k.many_min = atoi(value);
    else if (!strcmp(name, "narrow_below")) k.narrow_below = atoi(value);
    else if (!strcmp(name, "astat_tcn")) k.astat_tcn = atoi(value) != 0;
    else if (!strcmp(name, "astat_merge")) k.astat_merge = atoi(value) != 0;
    else if (!strcmp(name, "astat_k1024")) k.astat_k1024 = atoi(value) != 0;
    else if (!strcmp(name, "linkv_dma")) k.linkv_dma = atoi(value) != 0;
    else if (!strcmp(name, "stage_a_split_below")) k.stage_a_split_below = atoi(value);
    else if (!strcmp(name, "own_stats")) k.own_stats = atoi(value) != 0;
    else if (!strcmp(name, "store_aware")) k.store_aware = atoi(value) != 0;
    else if (!strcmp(name, "astat")) k.astat = atoi(value) != 0;
    else if (!strcmp(name, "attn_fused")) k.attn_fused = atoi(value) != 0;
    else if (!strcmp(name, "attn_fused_min")) k.attn_fused_min = atoi(value);
    else if (!strcmp(name, "geom3")) { strncpy(k.g3, value, sizeof(k.g3) - 1); k.g3[sizeof(k.g3) - 1] = 0; }
    else if (!strcmp(name, "geom4")) { strncpy(k.g4, value, sizeof(k.g4) - 1); k.g4[sizeof(k.g4) - 1] = 0; }
    else return MSF2_ERR_ARG;
    return MSF2_OK;
}
int msf2_num_frames(int T) { return T < 16 ? 0 : (T - 16) / 8 + 1; }

size_t msf2_workspace_bytes(const msf2_config* cfg, int B, int T) {
    if (check_cfg(cfg) != MSF2_OK || B < 1 || T < 16) return 0;
    if (!fits32((long)B * msf2_num_frames(T), 4 * cfg->dim + MSF2_QK_DIM)) return 0;
    Arena ar(nullptr, 0);
    fwd_ws(ar, *cfg, B, msf2_num_frames(T));
    return ar.off;
}

int msf2_forward(const msf2_config* cfg, const msf2_weights* wt, const float* mix, float* out, int B, int T,
                 void* workspace, size_t workspace_bytes, const msf2_taps* taps, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != MSF2_OK) return rc;
    if (!wt || !mix || !out || !workspace || B < 1 || T < 16 || (cfg->num_layers > 0 && !wt->layers)) return MSF2_ERR_ARG;
    if (cfg->recurrent == MSF2_REC_FSMN)          // checked before the first launch: nothing is enqueued for a bad weight set
        for (int l = 0; l < cfg->num_layers; ++l)
            if (!wt->layers[l].fsmn) return MSF2_ERR_ARG;
    const int C = cfg->dim, E = 2 * C, NH = 4 * C + MSF2_QK_DIM, S = msf2_num_frames(T);
    if (!fits32((long)B * S, NH)) return MSF2_ERR_ARG;
    Arena ar(workspace, workspace_bytes);
    FwdWs w = fwd_ws(ar, *cfg, B, S);
    if (!ar.ok()) return MSF2_ERR_WORKSPACE;
    Ctx c = {(hipStream_t)stream, B, S, T, (long)B * S, cfg->precision};
    const long MC = c.M * C;
    msf2_taps tp = {};
    if (taps) tp = *taps;

    hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(256), 0, c.st, reinterpret_cast<float4*>(w.zeros), 256);  // zero operand rows for the token shift at frame 0
    // encoder + statistics for the mask net's GroupNorm(1, C, eps 1e-8)
    hipLaunchKernelGGL(encoder_kernel, dim3(w.nblk_enc, B), dim3(256), 0, c.st, mix, wt->enc_w, w.enc, w.part, T, S, C, ENC_TS);
    launch_stats_finalize(c, w.part, w.nblk_enc, (long)S * C, 1e-8f, w.bstat);
    hipLaunchKernelGGL(posenc_kernel, dim3(cdiv((long)S * (C / 2), 256)), dim3(256), 0, c.st, wt->pe_inv_freq, wt->pe_scale, w.pe, S, C);
    // Every producer of x also writes partial row sums of squares (slabs of sq_w columns) for the ScaleNorm of the
    // token-shifted x that the next FLASH block starts with; sq_w == 0: the producer could not, a pass over x does it.
    int sq_w = 0;
    {   // x = skip = GroupNorm(enc) Wce^T + pos_enc
        ProjArgs a = proj_args(w.enc, C, wt->w_ce, C, C, S, w.x, C);
        a.out2 = w.skip; a.bstat = w.bstat; a.c1 = wt->c1_ce; a.c2 = wt->c2_ce; a.pe = w.pe; a.out16 = w.x16; a.sq_part = w.sqx;
        sq_w = launch_proj<A_PLAIN, EPI_NORM>(c, a);
    }
    copy_tap(c, tp.enc, w.enc, MC);
    copy_tap(c, tp.x0, w.x, MC);

    for (int l = 0; l < cfg->num_layers; ++l) {
        const msf2_layer_weights& lw = wt->layers[l];
        // ---- FLASH_ShareA_FFConvM ----
        if (sq_w > 0 && (C / 2) % sq_w == 0)
            hipLaunchKernelGGL(rowsq_finalize_kernel, dim3(cdiv(c.M, 256)), dim3(256), 0, c.st, w.sqx, cdiv(C, sq_w), w.rowstat, c.M, S, C, 1);
        else
            launch_rowstat<0>(c, w.x, C, w.rowstat, C, 1, 0.f);
        {   // [v | u | qk] = silu(ScaleNorm(shift(x)) W^T + b)
            ProjArgs a = proj_args(w.x, C, lw.w_hq, C, NH, S, w.hu_pre, NH);
            a.bias = lw.b_hq; a.rowstat = w.rowstat; a.act = 1; a.zeros = w.zeros; a.A16 = w.x16;
            if (w.hu_pre16) { a.out = nullptr; a.out16 = w.hu_pre16; }
            launch_proj<A_SHIFT, EPI_RS_BIAS_ACT>(c, a);
        }
        if (w.hu16) launch_dwconv17_bf16(c, w.hu_pre16, NH, lw.dw_hq, w.hu16, NH, NH);
        else launch_dwconv17(c, w.hu_pre, NH, lw.dw_hq, w.hu, NH, NH, 0);
        // bf16 mode: the attention output exists only as the bf16 image its projection reads
        // (the ScaleNorm inverse norms of o come out of the attention's own output kernel)
        run_attention(c, w.at, w.hu, NH, lw.os_gamma, lw.os_beta, lw.rot_freqs, w.o16 ? nullptr : w.o, E, w.o16, w.hu16, w.rowstat);
        {   // to_out projection
            // bf16 mode (C <= 512): po exists as bf16 only, in the first half of its buffer
            unsigned short* po16 = (w.o16 && C <= 512) ? reinterpret_cast<unsigned short*>(w.po) : nullptr;
            ProjArgs a = proj_args(w.o, E, lw.w_out, E, C, S, po16 ? nullptr : w.po, C);
            a.bias = lw.b_out; a.rowstat = w.rowstat; a.act = 1; a.A16 = w.o16; a.out16 = po16;
            launch_proj<A_PLAIN, EPI_RS_BIAS_ACT>(c, a);
        }
        // x += po + dwconv17(po), and the LayerNorm statistics of x that the gated TCN block starts with
        // (rowstat: the FSMN block has no use for them, but the bf16 form of this kernel always writes them)
        const bool fsmn = cfg->recurrent == MSF2_REC_FSMN;
        const bool have_ln = launch_dwconv17(c, w.po, C, lw.dw_out, w.x, C, C, 1, w.x16, fsmn ? (w.x16 ? w.rowstat : nullptr) : w.tc.rowstat,
                                             (w.o16 && C <= 512) ? reinterpret_cast<const unsigned short*>(w.po) : nullptr);
        if (l == 0) copy_tap(c, tp.flash0, w.x, MC);
        // ---- the recurrent block: GatedTCNBlock, or Gated_FSMN_Block_Dilated in place on x ----
        if (fsmn) {
            sq_w = run_gated_fsmn(c, w.fs, w.x, lw.fsmn, w.x, C, cfg->fsmn_inner, w.x16, w.sqx, w.x16);
        } else
            sq_w = run_gated_tcn(c, w.tc, w.x, w.x16, lw, C, have_ln, w.sqx);
        if (l == 0) copy_tap(c, tp.tcn0, w.x, MC);
    }

    // final LayerNorm(eps 1e-6) -> GroupNorm(1, C, eps 1e-8) + skip -> PReLU
    hipLaunchKernelGGL(final_ln_kernel, dim3(w.nblk_ln, B), dim3(256), 0, c.st, w.x, wt->fin_ln_w, wt->fin_ln_b, w.part, S, C, 1e-6f);
    launch_stats_finalize(c, w.part, w.nblk_ln, (long)S * C, 1e-8f, w.bstat);
    hipLaunchKernelGGL(final_gn_kernel, dim3(cdiv(MC / 4, 256)), dim3(256), 0, c.st, w.x, w.skip, w.bstat, wt->fin_gn_w,
                       wt->fin_gn_b, wt->prelu_a, w.po, tp.mdl, c.M, S, C, w.x16);

    // mask head
    float* z = w.hu_pre;                 // [M, 2C]
    float* zz = w.hu;                    // [2][M, C]
    float* sep = w.hu_pre + c.M * 2 * C; // [2][M, C]
    // bf16 mode: each stage also writes the bf16 image the next contraction reads (into buffers the layers are done with)
    unsigned short* y16 = w.x16;         // [M, C]
    unsigned short* z16 = w.hu_pre16;    // [M, 2C]
    unsigned short* zz16 = w.hu16;       // [2][M, C]
    {
        ProjArgs a = proj_args(w.po, C, wt->w_co, C, 2 * C, S, z, 2 * C);
        a.bias = wt->b_co; a.act = 0; a.A16 = y16; a.out16 = z16;
        launch_proj<A_PLAIN, EPI_RS_BIAS_ACT>(c, a);
    }
    for (int sp = 0; sp < 2; ++sp) {
        {
            ProjArgs a = proj_args(z + sp * C, 2 * C, wt->w_o, C, C, S, zz + sp * MC, C);
            a.W2 = wt->w_og; a.bias = wt->b_o; a.bias2 = wt->b_og;
            if (z16) { a.A16 = z16 + sp * C; a.out16 = zz16 + sp * MC; }
            launch_proj<A_PLAIN, EPI_TANH_SIG>(c, a);
        }
        {
            ProjArgs a = proj_args(zz + sp * MC, C, wt->w_cd, C, C, S, sep + sp * MC, C);
            if (zz16) a.A16 = zz16 + sp * MC;
            a.res = w.enc; a.ldr = C; a.out2 = sp == 0 ? tp.mask0 : tp.mask1;
            launch_proj<A_PLAIN, EPI_RELU_MUL>(c, a);
        }
        run_decoder(c, sep + sp * MC, wt->dec_w, w.fv, out + (long)sp * B * T, C);
    }
    return check_launch();
}

// ---------------------------------------------------------------------------
// single operators
// ---------------------------------------------------------------------------
size_t msf2_stats_ws_bytes(int B, int S) { return align_up(2L * B * cdiv(S, ENC_TS) * sizeof(float)); }

int msf2_encoder(const float* mix, const float* enc_w, float* enc, float* stats, float eps, int B, int T, int C,
                 void* ws, size_t ws_bytes, void* stream) {
    if (!mix || !enc_w || !enc || !stats || !ws || B < 1 || T < 16 || !dim_ok(C)) return MSF2_ERR_ARG;
    const int S = msf2_num_frames(T);
    if (ws_bytes < msf2_stats_ws_bytes(B, S)) return MSF2_ERR_WORKSPACE;
    Ctx c = {(hipStream_t)stream, B, S, T, (long)B * S};
    const int nblk = cdiv(S, ENC_TS);
    hipLaunchKernelGGL(encoder_kernel, dim3(nblk, B), dim3(256), 0, c.st, mix, enc_w, enc, (float*)ws, T, S, C, ENC_TS);
    launch_stats_finalize(c, (float*)ws, nblk, (long)S * C, eps, stats);
    return check_launch();
}

int msf2_dwconv17(const float* x, int ldx, const float* dw, float* y, int ldy, int B, int S, int C, int accumulate,
                  void* stream) {
    if (!x || !dw || !y || B < 1 || S < 1 || C < 4 || (C & 3) || ldx < C || ldy < C || (ldx & 3) || (ldy & 3)) return MSF2_ERR_ARG;
    Ctx c = {(hipStream_t)stream, B, S, 0, (long)B * S};
    launch_dwconv17(c, x, ldx, dw, y, ldy, C, accumulate);
    return check_launch();
}

size_t msf2_linear_ws_bytes(int B, int S) { return align_up((size_t)B * S * sizeof(float)) + 256; }

int msf2_linear(const float* x, int ldx, const float* W, const float* b, float* y, int ldy, int B, int S, int K, int N,
                int shift, int scale_norm, int act, int precision, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !W || !y || B < 1 || S < 1 || K < BK || (K % BK) || K > 1024 || N < 1 || ldx < K || ldy < N || (ldx & 3)) return MSF2_ERR_ARG;
    if ((precision != MSF2_PREC_FP32 && precision != MSF2_PREC_BF16) || act < 0 || act > 2) return MSF2_ERR_ARG;
    if (!fits32((long)B * S, ldx > ldy ? ldx : ldy)) return MSF2_ERR_ARG;
    Ctx c = {(hipStream_t)stream, B, S, 0, (long)B * S, precision};
    ProjArgs a = proj_args(x, ldx, W, K, N, S, y, ldy);
    a.bias = b; a.act = act;
    if (scale_norm || shift) {
        if (!ws || ws_bytes < msf2_linear_ws_bytes(B, S)) return MSF2_ERR_WORKSPACE;
    }
    if (scale_norm) {
        launch_rowstat<0>(c, x, ldx, (float*)ws, K, shift, 0.f);
        a.rowstat = (float*)ws;
    }
    if (shift) {
        a.zeros = (float*)((char*)ws + align_up((size_t)B * S * sizeof(float)));
        hipLaunchKernelGGL(zero_fill_kernel, dim3(1), dim3(64), 0, c.st, reinterpret_cast<float4*>(const_cast<float*>(a.zeros)), 16);
    }
    if (shift) launch_proj<A_SHIFT, EPI_RS_BIAS_ACT>(c, a);
    else launch_proj<A_PLAIN, EPI_RS_BIAS_ACT>(c, a);
    return check_launch();
}

size_t msf2_flash_attention_ws_bytes(int B, int S, int E) {
    Arena ar(nullptr, 0);
    attn_ws(ar, B, S, E);
    return ar.off;
}

int msf2_flash_attention(const float* hu, int ldhu, const float* os_gamma, const float* os_beta, const float* rot_freqs,
                         float* o, int B, int S, int E, int precision, void* ws, size_t ws_bytes, void* stream) {
    if (!hu || !os_gamma || !os_beta || !rot_freqs || !o || !ws || B < 1 || S < 1 || E < 128 || (E % 64) ||
        ldhu < 2 * E + MSF2_QK_DIM || (ldhu & 3))
        return MSF2_ERR_ARG;
    if ((precision != MSF2_PREC_FP32 && precision != MSF2_PREC_BF16) || !fits32((long)B * S, ldhu)) return MSF2_ERR_ARG;
    Arena ar(ws, ws_bytes);
    AttnWs w = attn_ws(ar, B, S, E);
    if (!ar.ok()) return MSF2_ERR_WORKSPACE;
    Ctx c = {(hipStream_t)stream, B, S, 0, (long)B * S, precision};
    run_attention(c, w, hu, ldhu, os_gamma, os_beta, rot_freqs, o, E);
    return check_launch();
}

size_t msf2_gated_tcn_ws_bytes(int B, int S, int C) {
    Arena a32(nullptr, 0), a16(nullptr, 0);
    tcn_ws(a32, B, S, C, MSF2_PREC_FP32);
    tcn_ws(a16, B, S, C, MSF2_PREC_BF16);
    return a32.off > a16.off ? a32.off : a16.off;   // the larger of the two layouts
}

int msf2_gated_tcn(float* x, const msf2_layer_weights* lw, int B, int S, int C, int precision, void* ws, size_t ws_bytes,
                   void* stream) {
    if (!x || !lw || !ws || B < 1 || S < 1 || !dim_ok(C)) return MSF2_ERR_ARG;
    if ((precision != MSF2_PREC_FP32 && precision != MSF2_PREC_BF16) || !fits32((long)B * S, C)) return MSF2_ERR_ARG;
    Arena ar(ws, ws_bytes);
    TcnWs w = tcn_ws(ar, B, S, C, precision);
    if (!ar.ok()) return MSF2_ERR_WORKSPACE;
    Ctx c = {(hipStream_t)stream, B, S, 0, (long)B * S, precision};
    run_gated_tcn(c, w, x, nullptr, *lw, C);
    return check_launch();
}

int msf2_tcn_pointwise(const float* h, const float* t, const float* bstat, const msf2_tcn_weights* tw, float* h_out,
                       int B, int S, int C, int precision, void* stream) {
    if (!h || !t || !bstat || !tw || !h_out || h == h_out || B < 1 || S < 1 || !dim_ok(C)) return MSF2_ERR_ARG;
    if ((precision != MSF2_PREC_FP32 && precision != MSF2_PREC_BF16) || !fits32((long)B * S, C)) return MSF2_ERR_ARG;
    Ctx c = {(hipStream_t)stream, B, S, 0, (long)B * S, precision};
    ProjArgs a = proj_args(h, C, tw->w_gate, C, C, S, h_out, C);
    a.A2 = t; a.W2 = tw->w_pw; a.bias = tw->b_gate; a.c1 = tw->c1; a.c2 = tw->c2;
    a.bstat = bstat; a.res = h; a.ldr = C;
    launch_proj<A_PLAIN, EPI_TCN>(c, a);
    return check_launch();
}

#ifdef MSF2_DIAG_STAMP
// timing-only build: copy out the per-wave cycle sums of the last contraction launch
int msf2_diag_read(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(msf2::g_stamp), n * sizeof(unsigned long long));
}
int msf2_diag_read2(unsigned long long* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(msf2::g_stamp2), n * sizeof(unsigned long long));
}
#endif

// ---- FSMN-family block -------------------------------------------------------
size_t msf2_gated_fsmn_ws_bytes(int B, int S, int dim, int inner) {
    if (B < 1 || S < 1 || dim < BK || inner < BK) return 0;
    Arena ar(nullptr, 0);
    fsmn_ws(ar, B, S, inner);
    return ar.off;
}
int msf2_gated_fsmn(const float* x, const msf2_fsmn_weights* fw, float* y, int B, int S, int dim, int inner, void* ws,
                    size_t ws_bytes, void* stream) {
    if (!x || !fw || !y || !ws || B < 1 || S < 1) return MSF2_ERR_ARG;
    if (!fsmn_dims_ok(dim, inner)) return MSF2_ERR_ARG;
    if (!fits32((long)B * S, 2 * inner > dim ? 2 * inner : dim)) return MSF2_ERR_ARG;
    if (ws_bytes < msf2_gated_fsmn_ws_bytes(B, S, dim, inner)) return MSF2_ERR_WORKSPACE;
    Arena ar(ws, ws_bytes);
    const FsmnWs w = fsmn_ws(ar, B, S, inner);
    Ctx c = {(hipStream_t)stream, B, S, 0, (long)B * S};
    run_gated_fsmn(c, w, x, fw, y, dim, inner);
    return check_launch();
}

size_t msf2_pit_si_snr_ws_bytes(int B, int T) {
    if (B < 1 || T < 1) return 0;
    return align_up((size_t)B * cdiv(T, PIT_CHUNK) * 12 * sizeof(double));
}

int msf2_pit_si_snr(const float* est, const float* src, float* si_snr, float* best, int* perm, int B, int T, void* ws,
                    size_t ws_bytes, void* stream) {
    if (!est || !src || (!si_snr && !best && !perm) || B < 1 || T < 2 || !ws) return MSF2_ERR_ARG;
    if (ws_bytes < msf2_pit_si_snr_ws_bytes(B, T)) return MSF2_ERR_WORKSPACE;
    const int nblk = cdiv(T, PIT_CHUNK);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pit_moments_kernel, dim3(nblk, B), dim3(256), 0, st, est, src, (double*)ws, B, T);
    hipLaunchKernelGGL(pit_finalize_kernel, dim3(B), dim3(64), 0, st, (const double*)ws, nblk, T, si_snr, best, perm);
    return check_launch();
}

// ---- the same statistic for callers that cannot allocate ----------------------
// The partial moments go to a fixed array in device memory instead of a caller-owned workspace.  Every GPU holds
// its own copy of the array at its own address; the address is looked up once per device (hipGetSymbolAddress)
// and only that pointer is ever handed to kernels -- the array's name is a host-side handle, not a device address.
static __device__ __align__(256) double scratch[(size_t)MSF2_PIT_NOALLOC_MAX_BLOCKS * 12];
constexpr int MAX_DEVICES = 64;
static std::mutex g_scratch_mu;
static double* g_scratch_ptr[MAX_DEVICES] = {};

// The array of the device that owns `ptr` (the kernels run where their operands live).  The calling thread's current
// device must be that device: the launches below go to `stream`, which HIP requires to belong to the current device, and
// hipGetSymbolAddress answers for the current device.  A mismatch is refused rather than written to another GPU's array.
static int scratch_for_operands(const void* ptr, double** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return MSF2_ERR_ARG;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return MSF2_ERR_ARG;                       // not a pointer HIP knows (host memory that was never registered)
    }
    if (attr.type != hipMemoryTypeDevice || attr.device != dev) return MSF2_ERR_ARG;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        double* p = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            g_last_hip_error = (int)e;
            return MSF2_ERR_LAUNCH;
        }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return MSF2_OK;
}

int msf2_pit_si_snr_noalloc(const float* est, const float* src, float* si_snr, float* best, int* perm, int B, int T,
                            void* stream) {
    if (!est || !src || (!si_snr && !best && !perm) || B < 1 || T < 2) return MSF2_ERR_ARG;
    const int nblk = cdiv(T, PIT_CHUNK);
    if ((long)B * nblk > MSF2_PIT_NOALLOC_MAX_BLOCKS) return MSF2_ERR_WORKSPACE;     // beyond the fixed array
    double* part = nullptr;
    int rc = scratch_for_operands(est, &part);
    if (rc != MSF2_OK) return rc;
    double* part_src = nullptr;
    rc = scratch_for_operands(src, &part_src);     // both operands on the device whose array is used
    if (rc != MSF2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    // every slot the finalize kernel reads ([B][nblk][12]) is written by the moments kernel of the same call, so
    // nothing of an earlier call's contents survives into this one and nothing needs clearing
    hipLaunchKernelGGL(pit_moments_kernel, dim3(nblk, B), dim3(256), 0, st, est, src, part, B, T);
    hipLaunchKernelGGL(pit_finalize_kernel, dim3(B), dim3(64), 0, st, (const double*)part, nblk, T, si_snr, best, perm);
    return check_launch();
}

size_t msf2_decoder_ws_bytes(int B, int S) { return align_up((size_t)B * S * 16 * sizeof(float)); }

int msf2_decoder(const float* sep, const float* dec_w, float* est, int B, int S, int C, int T, void* ws, size_t ws_bytes,
                 void* stream) {
    if (!sep || !dec_w || !est || !ws || B < 1 || S < 1 || T < 1 || !dim_ok(C)) return MSF2_ERR_ARG;
    if (ws_bytes < msf2_decoder_ws_bytes(B, S)) return MSF2_ERR_WORKSPACE;
    Ctx c = {(hipStream_t)stream, B, S, T, (long)B * S};
    run_decoder(c, sep, dec_w, (float*)ws, est, C);
    return check_launch();
}

}  // extern "C"
