/*
 * msf2.h -- C ABI of libmsf2.so: the MossFormer2_SS separation forward pass as
 * hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference is pure Python on torch.nn and has no native interface, so
 * each entry point cites the reference FUNCTION it replaces
 * (paths relative to the reference tree).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated otherwise;
 *   - activations are token-major: [B, S, C] contiguous (row m = b*S + s);
 *   - the caller owns all memory, including the workspace; the library keeps
 *     no mutable global state, so calls are re-entrant across host threads;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) and the
 *     call returns without synchronising;
 *   - return value 0 = success, negative = error (msf2_strerror()).
 *     No exception crosses the boundary.
 */
#ifndef MSF2_H
#define MSF2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSF2_OK 0
#define MSF2_ERR_ARG (-1)        /* null pointer, bad shape or unsupported size   */
#define MSF2_ERR_WORKSPACE (-2)  /* workspace too small                           */
#define MSF2_ERR_LAUNCH (-3)     /* a HIP launch failed (see msf2_last_hip_error) */

#define MSF2_TCN_DEPTH 8
#define MSF2_QK_DIM 128
#define MSF2_GROUP 256
#define MSF2_DW_TAPS 17

/* Arithmetic used by the contraction kernels. */
#define MSF2_PREC_FP32 0   /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32)                 */
#define MSF2_PREC_BF16 1   /* bf16 MFMA operands, fp32 accumulate/statistics/residual  */

/* The recurrent block interleaved with the FLASH layers (the list the reference calls `fsmn`,
 * mossformer2_block.py:635-643, :652-656). */
#define MSF2_REC_TCN 0     /* GatedTCNBlock (mossformer2_block.py:534-566): what the reference revision instantiates   */
#define MSF2_REC_FSMN 1    /* Gated_FSMN_Block_Dilated (mossformer2_block.py:430-463): the FSMN-family block of the
                              upstream architecture, present in the reference but not instantiated by it              */

/* Model geometry (reference: models/mossformer2/mossformer2.py:742-753). */
typedef struct msf2_config {
    int32_t dim;         /* encoder_embedding_dim == mossformer_sequence_dim; multiple of 64 */
    int32_t num_layers;  /* num_mossformer_layer                                              */
    int32_t enc_kernel;  /* encoder_kernel_size (must be 16; stride 8)                        */
    int32_t num_spks;    /* must be 2                                                         */
    int32_t precision;   /* MSF2_PREC_*                                                       */
    int32_t recurrent;   /* MSF2_REC_*: the block that follows each FLASH layer               */
    int32_t fsmn_inner;  /* MSF2_REC_FSMN: inner_channels of the block (256 in the reference) */
} msf2_config;

/*
 * Packed weights.  Device resident, written once by the host; fp32, except that
 * with precision == MSF2_PREC_BF16 the projection matrices (the [N][K] arrays:
 * w_hq, w_out, w_gg, w_in, w_po, tcn[].w_gate, tcn[].w_pw, w_ce, w_co, w_o,
 * w_og, w_cd) hold bf16 values.  Layout as written by the host
 * side (speech_separation_amd/packing.py) from a reference state_dict:
 * norm gains are folded into the following projection, depthwise filters are
 * stored tap-major [taps][C].  "c1"/"c2" are the two vectors that move a
 * folded normalisation into a GEMM epilogue:
 *     norm(x) W^T + b  ==  rstd * (x (W diag(gain))^T - mean * c2) + c1,
 *     c2[n] = sum_k W[n,k] gain[k],   c1[n] = sum_k W[n,k] offset[k] + b[n].
 */
typedef struct msf2_tcn_weights {   /* reference: mossformer2_block.py:508-531 (TCNBlock) */
    const float* dw_w;     /* [3][C]  depthwise taps, tap-major          */
    const float* dw_b;     /* [C]                                        */
    const float* w_gate;   /* [C][C]  gate.0.weight                      */
    const float* b_gate;   /* [C]                                        */
    const float* w_pw;     /* [C][C]  net.3.weight * diag(net.2.weight)  */
    const float* c1;       /* [C]                                        */
    const float* c2;       /* [C]                                        */
} msf2_tcn_weights;

/* Gated_FSMN_Block_Dilated (reference: mossformer2_block.py:430-463), packed by
 * speech_separation_amd.packing.pack_fsmn_block; see msf2_gated_fsmn below. */
typedef struct msf2_fsmn_weights {
    const float* w1;      /* [inner][dim]  conv1.0.weight                      */
    const float* b1;      /* [inner]                                            */
    float prelu1;         /* conv1.1.weight (one slope)                         */
    const float* ln1_g;   /* [inner] norm1                                      */
    const float* ln1_b;
    const float* w_uv;    /* [2 inner][inner] to_u | to_v Linear, LayerNorm gain folded in */
    const float* c1_uv;   /* [2 inner] W beta + bias                            */
    const float* c2_uv;   /* [2 inner] row sums of the folded weights           */
    const float* dw_uv;   /* [17][2 inner] ConvModule filters, tap major        */
    const float* w_lin;   /* [inner][inner] fsmn.linear                         */
    const float* b_lin;
    const float* w_proj;  /* [inner][inner] fsmn.project (no bias)              */
    const float* w_c1;    /* [39][inner]    fsmn.conv.conv1, tap major          */
    const float* in1_g;   /* [inner] fsmn.conv.norm1 / prelu1                   */
    const float* in1_b;
    const float* pr1;
    const float* w_c2;    /* [39][2][inner] fsmn.conv.conv2, tap major, input-of-group major */
    const float* in2_g;
    const float* in2_b;
    const float* pr2;
    const float* ln2_g;   /* [inner] norm2                                      */
    const float* ln2_b;
    const float* w2;      /* [dim][inner]  conv2.weight                         */
    const float* b2;      /* [dim]                                              */
} msf2_fsmn_weights;

typedef struct msf2_layer_weights {
    /* FLASH_ShareA_FFConvM (reference: mossformer2_block.py:171-325) */
    const float* w_hq;     /* [4*dim + 128][dim]  to_hidden | to_qk rows, ScaleNorm gain folded */
    const float* b_hq;     /* [4*dim + 128]                                                     */
    const float* dw_hq;    /* [17][4*dim + 128]                                                 */
    const float* os_gamma; /* [4][128]  OffsetScale                                             */
    const float* os_beta;  /* [4][128]                                                          */
    const float* rot_freqs;/* [16]                                                              */
    const float* w_out;    /* [dim][2*dim]  to_out, gain folded                                 */
    const float* b_out;    /* [dim]                                                             */
    const float* dw_out;   /* [17][dim]                                                         */
    /* GatedTCNBlock (reference: mossformer2_block.py:534-566) */
    const float* w_gg;     /* [dim][dim] global_gate.0                                          */
    const float* b_gg;     /* [dim]                                                             */
    const float* w_in;     /* [dim][dim] proj_in * diag(norm.weight)                            */
    const float* c1_in;    /* [dim]                                                             */
    const float* c2_in;    /* [dim]                                                             */
    const float* w_po;     /* [dim][dim] proj_out                                               */
    const float* b_po;     /* [dim]                                                             */
    msf2_tcn_weights tcn[MSF2_TCN_DEPTH];
    /* MSF2_REC_FSMN: HOST pointer to the block's weights (the GatedTCNBlock fields above are then unused);
     * with MSF2_PREC_BF16 its projection matrices (w1, w_uv, w_lin, w_proj, w2) hold bf16 values */
    const msf2_fsmn_weights* fsmn;
} msf2_layer_weights;

typedef struct msf2_weights {
    const float* enc_w;    /* [16][dim] encoder taps, tap-major (enc.conv1d.weight)             */
    const float* w_ce;     /* [dim][dim] conv1d_encoder * diag(mask_net.norm.weight)            */
    const float* c1_ce;    /* [dim]                                                             */
    const float* c2_ce;    /* [dim]                                                             */
    const float* pe_inv_freq; /* [dim/2]                                                        */
    float pe_scale;
    float prelu_a;
    const float* fin_ln_w; /* [dim] mask_net.mdl.intra_mdl.norm                                 */
    const float* fin_ln_b;
    const float* fin_gn_w; /* [dim] mask_net.mdl.intra_norm                                     */
    const float* fin_gn_b;
    const float* w_co;     /* [2*dim][dim] conv1d_out                                           */
    const float* b_co;     /* [2*dim]                                                           */
    const float* w_o;      /* [dim][dim] output.0                                               */
    const float* b_o;
    const float* w_og;     /* [dim][dim] output_gate.0                                          */
    const float* b_og;
    const float* w_cd;     /* [dim][dim] conv1_decoder                                          */
    const float* dec_w;    /* [dim][16] dec.weight                                              */
    const msf2_layer_weights* layers;  /* HOST array of num_layers entries                      */
} msf2_weights;

/* Optional device buffers that receive named intermediates (any may be NULL).
 * Used by the parity tests to localise a mismatch. All [B, S, dim]. */
typedef struct msf2_taps {
    float* enc;     /* encoder output                                  */
    float* x0;      /* mask-net front end output (input of layer 0)    */
    float* flash0;  /* after FLASH layer 0                             */
    float* tcn0;    /* after gated-TCN block 0                         */
    float* mdl;     /* after final LayerNorm + GroupNorm + skip        */
    float* mask0;   /* speaker-0 mask                                  */
    float* mask1;   /* speaker-1 mask                                  */
} msf2_taps;

const char* msf2_strerror(int code);
/* hipError_t of the most recent failed launch on the calling thread (0 if none). */
int msf2_last_hip_error(void);
int msf2_version(void);
/* sizeof() of the ABI structs as this library was compiled, for bindings to
 * verify their mirror: 0 msf2_config, 1 msf2_tcn_weights, 2 msf2_layer_weights,
 * 3 msf2_weights, 4 msf2_taps, 5 msf2_fsmn_weights; -1 for any other index. */
int msf2_sizeof(int which);

/* Measurement aid with no counterpart in the reference (the reference times whole
 * utterances only, utils/decode.py callers in inference.py:80-95).  Between
 * msf2_timing_begin and msf2_timing_end every launch of the dominant kernel -- the
 * pointwise contraction of a TCN block -- is bracketed by HIP events on the stream it
 * is launched on; msf2_timing_end waits for them and returns their summed duration
 * and count.  The timer belongs to the thread that called msf2_timing_begin; launches made by other
 * threads meanwhile are not recorded.  Keep it off inside a timed region. */
int msf2_timing_begin(void);
int msf2_timing_end(double* total_ms, int* launches);
/* The same timer on a chosen kernel: MSF2_TIMED_TCN_POINTWISE (what msf2_timing_begin selects), or
 * MSF2_TIMED_HIDDEN_PROJ = the token-shifted to_hidden | to_qk projection of a FLASH layer
 * (FFConvM.forward, mossformer2_block.py:79-100 after the shift of :245-248), one launch per layer: the kernel that
 * dominates the forward when the recurrent block is the FSMN-family one.  Any other value returns -1. */
#define MSF2_TIMED_TCN_POINTWISE 0
#define MSF2_TIMED_HIDDEN_PROJ 1
int msf2_timing_begin_kernel(int which);

/* Test aid with no counterpart in the reference: overrides, for the CALLING THREAD only, which of the equivalent
 * kernel forms a launch takes (the parity tests compare them: bit-identical, except the bf16 64-row tiles, which agree
 * with the 128-row ones to bf16 rounding -- so bf16 output is not batch-invariant to the last bit: the form follows the
 * number of tiles of the call).  name: "ncol_loop", "big_maxn", "no_big", "geom2", "narrow", "narrow_below", "many_min",
 * "row128", "own_stats", "attn_fused", "attn_fused_min", "astat" (0: the bf16 K = 512 projections with a bias / activation
 * epilogue stay on the tile kernels instead of the register-resident-activation kernel), "astat_k1024" (0: the K = 1024 one too), "astat_merge" (0: gate and proj_in of the
 * gated TCN block as two walks of that kernel instead of one), "astat_tcn" (0: the two-product TCN contraction on the tile kernel), "store_aware", "linkv_dma" (0: the bf16
 * key/value summary on the register-staged tile kernel), "stage_a_split_below" (integers as text) or "geom3", "geom4" (letters naming
 * projections).  name == NULL restores the defaults; an unknown name returns -1.  The library reads no environment
 * variable. */
int msf2_debug_kernel_choice(const char* name, const char* value);

/* Frames produced by the encoder for T samples: (T - 16) / 8 + 1. */
int msf2_num_frames(int T);

/* Bytes of workspace msf2_forward needs for a [B, T] input. */
size_t msf2_workspace_bytes(const msf2_config* cfg, int B, int T);

/*
 * Whole forward pass, replaces MossFormer.forward
 * (reference: models/mossformer2/mossformer2.py:712-736 and everything it
 * calls: Encoder :186-210, MossFormer_MaskNet :588-646, Computation_Block
 * :476-509, MossFormerM :331-352, MossformerBlock_GFSMN
 * mossformer2_block.py:652-656, Decoder mossformer2.py:238-260).
 *   mix : [B, T]            out : [num_spks, B, T]
 */
int msf2_forward(const msf2_config* cfg, const msf2_weights* w,
                 const float* mix, float* out, int B, int T,
                 void* workspace, size_t workspace_bytes,
                 const msf2_taps* taps, void* stream);

/* ---- single operators (each is one stage of the forward; exported so the
 *      parity tests can check every kernel against the oracle) ------------- */

/* Encoder.forward (mossformer2.py:186-210): relu(conv1d(k=16, stride 8)) written
 * token-major, plus per-utterance (mean, rstd) of the output with `eps`
 * (the statistics GroupNorm(1, C) needs, mossformer2.py:609).
 *   mix [B,T] -> enc [B,S,C]; stats [B][2]; ws >= msf2_stats_ws_bytes(B,S). */
int msf2_encoder(const float* mix, const float* enc_w, float* enc, float* stats,
                 float eps, int B, int T, int C, void* ws, size_t ws_bytes, void* stream);
size_t msf2_stats_ws_bytes(int B, int S);

/* ConvModule.forward (conv_module.py:298-299): y = x + dwconv17(x) along time.
 *   x [B,S,C] (row stride ldx) -> y (row stride ldy); dw [17][C].
 *   accumulate != 0: y += x + dwconv17(x)   (the residual add of
 *   FLASH_ShareA_FFConvM.forward, mossformer2_block.py:259). */
int msf2_dwconv17(const float* x, int ldx, const float* dw, float* y, int ldy,
                  int B, int S, int C, int accumulate, void* stream);

/* Linear with fused pre-normalisation statistics and activation
 * (FFConvM's ScaleNorm -> Linear -> SiLU, mossformer2_block.py:79-100, with the
 * token shift of :245-248 when shift != 0):
 *   y = act(rowscale(x') W^T + b), x' = token-shifted x; W has the gain folded.
 *   act: 0 none, 1 SiLU, 2 sigmoid.  scale_norm != 0 applies ScaleNorm. */
int msf2_linear(const float* x, int ldx, const float* W, const float* b, float* y, int ldy,
                int B, int S, int K, int N, int shift, int scale_norm, int act,
                int precision, void* ws, size_t ws_bytes, void* stream);
size_t msf2_linear_ws_bytes(int B, int S);

/* FLASH_ShareA_FFConvM.cal_attention + gating (mossformer2_block.py:255-258,
 * :262-325; OffsetScale :68-77; rotary: rotary-embedding-torch 0.8.3
 * rotate_queries_or_keys):
 *   hu [B,S,2*E+128] = v | u | qk (row stride ldhu) ->
 *   o [B,S,E] = (att_u * v) * sigmoid(att_v * u). */
int msf2_flash_attention(const float* hu, int ldhu, const float* os_gamma, const float* os_beta,
                         const float* rot_freqs, float* o, int B, int S, int E,
                         int precision, void* ws, size_t ws_bytes, void* stream);
size_t msf2_flash_attention_ws_bytes(int B, int S, int E);

/* GatedTCNBlock.forward (mossformer2_block.py:556-566, TCNBlock :530-531), in place on x. */
int msf2_gated_tcn(float* x, const msf2_layer_weights* lw, int B, int S, int C,
                   int precision, void* ws, size_t ws_bytes, void* stream);
size_t msf2_gated_tcn_ws_bytes(int B, int S, int C);

/* The pointwise half of one TCNBlock (mossformer2_block.py:513-531), ONE kernel launch:
 *   h_out = h + sigmoid(h Wg^T + bg) * (GroupNorm(t) Wp^T + bp),
 * t = silu(dwconv3(h)) [B,S,C], bstat [B][2] = (mean, rstd) of t per utterance.
 * This is the dominant kernel of the forward (8 per layer). h_out must not alias h. */
int msf2_tcn_pointwise(const float* h, const float* t, const float* bstat, const msf2_tcn_weights* tw,
                       float* h_out, int B, int S, int C, int precision, void* stream);

/* mask multiply + Decoder.forward + length fix-up (mossformer2.py:715-731,
 * :238-260): sep [B,S,C] (already enc*mask) -> est [B,T]. */
int msf2_decoder(const float* sep, const float* dec_w, float* est, int B, int S, int C, int T,
                 void* ws, size_t ws_bytes, void* stream);
size_t msf2_decoder_ws_bytes(int B, int S);

/*
 * FSMN-family recurrent block, replaces Gated_FSMN_Block_Dilated.forward
 * (reference: models/mossformer2/mossformer2_block.py:434-463 with Gated_FSMN_dilated :361-393, FFConvM with
 * LayerNorm :79-100, UniDeepFsmn_dilated / DilatedDenseNet models/mossformer2/fsmn.py:62-114, CLayerNorm
 * models/mossformer2/layer_norm.py:14-35).  The block is not reachable from MossFormer2_SS in the
 * reference revision this library follows (its layers use GatedTCNBlock), so it is a stand-alone
 * operator: fp32, x and y [B, S, dim], inner = the block's inner_channels (256 in the reference).
 * Weights as packed by speech_separation_amd.packing.pack_fsmn_block (struct msf2_fsmn_weights above).
 * The same launch sequence runs inside msf2_forward when msf2_config.recurrent == MSF2_REC_FSMN.
 */
int msf2_gated_fsmn(const float* x, const msf2_fsmn_weights* w, float* y, int B, int S, int dim, int inner,
                    void* workspace, size_t workspace_bytes, void* stream);
size_t msf2_gated_fsmn_ws_bytes(int B, int S, int dim, int inner);

/*
 * Permutation-invariant SI-SNR of two separated sources against their references, per utterance
 * (reference: losses/loss.py:653-723 cal_si_snr -- zero-mean both, project, 10 log10 with EPS = 1e-8 in
 * the same three places -- and the two-source case of PitWrapper, losses/loss.py:24-178, as used by
 * get_si_snr_with_pitwrapper :632-650).  The statistic stays on the device: a sweep gathers B floats per
 * batch instead of the waveforms.
 *   est, src : [2][B][T] fp32
 *   si_snr   : [B][2 (source)][2 (estimate)] dB, may be null
 *   best     : [B] mean SI-SNR (dB) of the better speaker assignment, may be null
 *   perm     : [B] 0 = identity assignment, 1 = swapped, may be null
 */
int msf2_pit_si_snr(const float* est, const float* src, float* si_snr, float* best, int* perm, int B, int T,
                    void* workspace, size_t workspace_bytes, void* stream);
size_t msf2_pit_si_snr_ws_bytes(int B, int T);

/*
 * msf2_pit_si_snr for callers that cannot allocate.  No counterpart in the reference: it exists for code
 * under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same inputs, outputs
 * and results as msf2_pit_si_snr, but the partial moments go to a fixed array in device memory that the
 * library holds on every GPU it runs on until the process ends (MSF2_PIT_NOALLOC_MAX_BLOCKS * 96 bytes =
 * 768 KiB), so the call takes no workspace and performs no allocation of any kind.
 *   limit : B * ceil(T / 8192) <= MSF2_PIT_NOALLOC_MAX_BLOCKS, else MSF2_ERR_WORKSPACE (nothing is launched);
 *   the array is shared by all calls on one GPU: calls that overlap on the same GPU (two streams, two host
 *   threads) race on it -- serialise them, or use msf2_pit_si_snr with a workspace per call;
 *   the first call on a device looks up the array's device address (no allocation); make that call outside
 *   a graph capture;
 *   est and src must be device memory of the calling thread's CURRENT device (the one `stream` belongs to): the
 *   array used is that device's, and operands living elsewhere are refused with MSF2_ERR_ARG.
 */
#define MSF2_PIT_NOALLOC_MAX_BLOCKS 8192
int msf2_pit_si_snr_noalloc(const float* est, const float* src, float* si_snr, float* best, int* perm, int B, int T,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSF2_H */
