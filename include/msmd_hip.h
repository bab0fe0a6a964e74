/*
 * msmd_hip.h -- C ABI of libmsmd_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * audio -> motion-coefficient hot path of ubisoft/ubisoft-laforge-msmd.
 *
 * The reference has no FFI/plugin layer: the path sits behind Python nn.Module calls
 * (SURVEY.md section 8b).  Each entry point below replaces the implicit torch/cuDNN/cuBLAS/HF
 * device ops issued by the cited reference lines.  Conventions:
 *   - plain pointers + sizes, no torch types; every pointer is DEVICE memory owned by the caller;
 *   - no hidden allocation, no host synchronisation (graph-capturable); workspaces are passed in;
 *   - re-entrant per stream; `stream` is a hipStream_t passed as void*;
 *   - returns a hipError_t-compatible int (0 = success); argument errors return hipErrorInvalidValue (1);
 *   - activations are channels-last, row-major (rows, cols); `dtype` selects fp32 (parity mode)
 *     or bf16 storage with fp32 accumulation (speed mode).  Biases / norm affine params / statistics
 *     are always fp32.
 */
#ifndef MSMD_HIP_H
#define MSMD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSMD_F32 0
#define MSMD_BF16 1
#define MSMD_F16 2 /* IEEE half storage, fp32 accumulate: inference kernels (GEMM, attention, norm, audio, diffusion) */
/* "split pair" storage of fp32-grade values for the parity-grade speed mode (inference kernels): x ~= hi + lo * 2^-11
 * with hi = RN_f16(x), lo = RN_f16((x - hi) * 2^11), |err| <= 2^-22 |x| + 2^-35 (typically 2^-24 |x|) for |x| < 65504.  A logical row of C values
 * (C % 32 == 0) is stored as 2 C fp16 numbers in 32-element blocks [hi x 32 | lo x 32]; all sizes / leading
 * dimensions / strides in this header stay in LOGICAL elements.  Contractions run as three f16 MFMAs per k-step
 * (hi.hi, hi.lo, lo.hi; fp32 accumulate), i.e. at 1/3 of the f16 MFMA rate instead of the 1/16 of the exact-fp32
 * MFMA.  The reference computes in fp32 everywhere (training_script.py:548-551: no autocast); this is the mode that
 * meets its 1e-4 tolerance at speed. */
#define MSMD_F16X2 3

#define MSMD_ACT_NONE 0
#define MSMD_ACT_GELU 1 /* exact erf GELU */
#define MSMD_ACT_ELU 2  /* alpha = 1 */

typedef void* msmd_stream_t; /* hipStream_t */

/* Library / device probe: returns the ABI version; safe to call without a GPU. */
int msmd_abi_version(void);
/* The library keeps NO process-global state: every entry point is re-entrant per stream.  What used to be developer
 * knobs travels per call -- the GEMM kernel variant and epilogue flags in `act` (msmd_gemm below), the contraction
 * split count of msmd_gemm_tn in its `accumulate` argument.  The experimental kernel families of DESIGN.md section 5 /
 * 5b and their A/B switch exist only in the developer build (`make -C csrc EXP=1` -> libmsmd_hip_exp.so). */
#define MSMD_GEMM_VARIANT(v) ((v) << 8)   /* bits 8-15 of `act`: 0 = shape heuristic, 9 / 12 / 13 / 14 / 15 / 17 / 80 (bf16, fp16), 1 / 5 / 14 / 80 (f16x2) */
#define MSMD_GEMM_WRITE_THROUGH (1 << 16) /* output stores carry `sc1`: the bytes leave the XCD's L2 as they are stored */
#define MSMD_GEMM_PAIRED_STORES (1 << 17) /* 16-bit outputs: lane pairs swap a fragment row, one 16-byte store each */
#define MSMD_GEMM_STAGGER (1 << 18)       /* multi-round launches: the second workgroup of every CU starts half a tile period late */
#define MSMD_GEMM_ONE_TILE_PER_WORKGROUP (1 << 19) /* opt out of the persistent form of multi-round launches (A/B; same bits) */
#define MSMD_GEMM_NO_256_TILE (1 << 20)   /* opt out of the 256 x 256 8-phase kernel (variant 80) where the heuristic would pick it (A/B) */
#define MSMD_GEMM_W_BELOW_32 (1 << 21)    /* MSMD_F16X2 operands: the caller states |W| < 32 everywhere (model weights).  The 256 x 256
                                             kernel (variant 80) may then scale W's hi plane by 2^11 in registers and keep ONE sum in
                                             units of 2^-11 instead of folding the cross terms once per K tile (11-17 % faster).  Without
                                             the bit nothing is assumed about W.  A W element of 32 or more under this bit overflows fp16. */

/* Measurement aid: one wavefront that spins for `us` microseconds of the 100 MHz constant clock (s_memrealtime) and
 * optionally stores the ticks it actually spun.  bench.py times it at two lengths to calibrate the overhead of a HIP
 * event pair around ONE launch, so that its per-launch roofline figures agree with rocprofv3 --kernel-trace. */
int msmd_spin_us(float us, long* ticks_out, msmd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dense contraction on MFMA:  C = act(A . W^T + bias) + residual
 *   A: (M, K) activations, row m at  A + (m / rows_per_batch) * a_batch_stride + (m % rows_per_batch) * lda
 *      (elements).  With rows_per_batch = T_out, lda = stride*C_in, a_batch_stride = T_in*C_in and
 *      K = k*C_in this IS a strided Conv1d over a channels-last signal with no im2col buffer.
 *   W: (N, K) row-major, leading dimension ldw (torch Linear layout; conv weights repacked (N, k*C_in)).
 *   bias: fp32 (N) or NULL.  residual: (M, N) ld ldr or NULL.  C: (M, N) ld ldc.
 *   in_dtype: dtype of A and W.  out_dtype: dtype of C and residual.
 *   in_dtype MSMD_F16X2 (split storage, the parity-grade speed mode): three f16 MFMAs per k-step, fp32-grade result;
 *   out_dtype MSMD_F32 or MSMD_F16X2; K, lda, ldw, a_batch_stride and the A / W strides must be multiples of 32, and
 *   for split output also N % 4 == 0 and ldc / ldr / strideC / strideR multiples of 32.
 *   batch > 1 launches independent problems with the given element strides (grouped conv).
 *   act: MSMD_ACT_* in bits 0-7; bits 8-15 may carry a kernel-variant hint (MSMD_GEMM_VARIANT; 0 = the library's own
 *   shape heuristic; every variant computes bit-identical results -- up to one fused multiply-add's rounding
 *   where an activation meets a residual, see tests/test_kernels_gpu.py); bits 16-23 epilogue flags (MSMD_GEMM_WRITE_THROUGH,
 *   MSMD_GEMM_PAIRED_STORES: how the output is stored, never what is stored).
 *   Requirements: K % (16 / sizeof(in)) == 0, lda/ldw/a_batch_stride/strideA/strideW multiples of the same.
 * Replaces: nn.Linear / nn.Conv1d / nn.MultiheadAttention projections at reference model.py:115,856-906,
 *   style_encoder.py:135-175, utils/wav2vec2.py:79,95,111 (HF conv stack, projection, pos-conv, encoder FFN).
 */
int msmd_gemm(const void* A, const void* W, const float* bias, const void* residual, void* C,
              int M, int N, int K, int in_dtype, int out_dtype,
              long lda, int rows_per_batch, long a_batch_stride, long ldw, long ldc, long ldr, int act,
              int batch, long strideA, long strideW, long strideC, long strideBias, long strideR,
              msmd_stream_t stream);

/* The shape rule of msmd_gemm / msmd_gemm_ln for their 256 x 256-tile kernel (variant 80: 16-bit operands and output of one
 * problem, N % 256 == 0, K % 64 == 0, K >= 128, inference epilogues): 1 when a call of this shape that the kernel can take
 * is routed to it, 0 otherwise.  Pure function of (M, N, K); lets a caller that times launches (bench.py's roofline leg) file
 * each one under the kernel that ran it.  No counterpart in the reference. */
int msmd_gemm_256_tile_rule(int M, int N, int K);
/* The same for MSMD_F16X2 operands (variant 80 of the split GEMM: one problem, N % 256 == 0, K % 32 == 0, K >= 64, fp32 or split
 * output); w_below_32 = the call carries MSMD_GEMM_W_BELOW_32 (the fold-free form of the kernel wins on more shapes). */
int msmd_gemm_256_tile_rule_f16x2(int M, int N, int K, int w_below_32);

/* msmd_gemm with the training epilogue:  C = dropout_p(act(A . W^T + bias)) + residual, and optionally
 * z_out = A . W^T + bias (the pre-activation the backward needs; layout and dtype of C).  The keep mask is
 * Philox4x32-10(rng_state[seed, step], site, (m * N + n) / 4) -- exactly msmd_dropout's mask on a contiguous (M, N)
 * tensor, so msmd_dropout / msmd_act_bwd_dropout regenerate it in the backward.  p_drop > 0 requires N % 4 == 0,
 * ldc == N, batch == 1.  Replaces nn.Linear -> activation -> nn.Dropout (-> residual add) chains of the HF encoder
 * layers, nn.TransformerDecoderLayer / EncoderLayer and the style encoder in train() mode. */
/* LayerNorm folded into the GEMMs around it (post-LN / pre-LN transformer blocks without LayerNorm launches and without
 * materialising the normalised rows; 16-bit operands and output of one dtype, K % 64 == 0, N % 64 == 0, no batch).
 *   C = act(LN_A(A) . W^T + bias) + LN_R(residual)       bias required; two forms, anything else returns 1:
 *     operand form:  a_stats + w_colsum, no residual, no stats_out;
 *     residual form: residual required, r_stats (+ r_gamma, r_beta) and stats_out each optional.
 *   Row statistics travel as per-row partial (sum, sum of squares) over column slabs, fp32, slab-major: (cols / slab, rows, 2).  The slab
 *   is what one wave of the producing kernel holds of a row: slab_out = 64 selects the 128 x 128 tile (N % 128 == 0),
 *   slab_out = 32 the 64 x 64 tile (for grids that would not fill the chip otherwise); slab_in is the slab the producer
 *   of a_stats / r_stats used.
 *   a_stats (K / slab_in, M, 2) + w_colsum (N): A holds UN-normalised rows u; W must carry the LayerNorm weight folded in
 *     (W'[n][k] = gamma[k] W[n][k]), w_colsum[n] = sum_k W'[n][k] (of the ROUNDED W'), bias[n] = b[n] + sum_k beta[k] W[n][k]:
 *     the epilogue applies  rstd (acc - mu w_colsum[n]) + bias[n].
 *   r_stats (N / slab_in, M, 2) + r_gamma / r_beta (N): the residual operand holds un-normalised rows; LN is applied on the fly.
 *   stats_out (N / slab_out, M, 2): statistics of the stored (rounded) rows of C.
 * Replaces the LayerNorm modules between the Linear layers of HF Wav2Vec2EncoderLayer / HubertEncoderLayer(StableLayerNorm)
 *  and nn.TransformerDecoderLayer (reference model.py:874-878): the sequence Linear -> +residual -> LayerNorm ->
 * Linear becomes two launches. */
int msmd_gemm_ln(const void* A, const void* W, const float* bias, const void* residual, void* C, int M, int N, int K,
                 int in_dtype, int out_dtype, long lda, long ldw, long ldc, long ldr, int act, const float* a_stats,
                 const float* w_colsum, const float* r_stats, const float* r_gamma, const float* r_beta,
                 float* stats_out, int slab_in, int slab_out, float eps, msmd_stream_t stream);

int msmd_gemm_ex(const void* A, const void* W, const float* bias, const void* residual, void* C, int M, int N, int K,
                 int in_dtype, int out_dtype, long lda, int rows_per_batch, long a_batch_stride, long ldw, long ldc,
                 long ldr, int act, int batch, long strideA, long strideW, long strideC, long strideBias, long strideR,
                 void* z_out, float p_drop, const unsigned long* rng_state, unsigned int site, msmd_stream_t stream);

/* Two-level batched GEMM C[zo][zi] = A[zo][zi] . W[zo][zi]^T (no bias / residual / activation): operand z =
 * zo * batch_inner + zi starts at base + zo * stride_o + zi * stride_i.  Used by the explicit (materialised-P)
 * training attention, where zo = batch and zi = head index into packed (B, T, H*64) tensors. */
/* dZ = keep_mask / (1 - p_drop) * act'(Z) * (A . W^T)  (16-bit; C and Z contiguous (M, N), N % 4 == 0): the data gradient of a
 * Linear whose input was dropout(act(Z)), with the backward of that activation + dropout applied in the GEMM epilogue (mask
 * regenerated from (rng_state, site) as by msmd_gemm_ex / msmd_act_bwd_dropout; p_drop = 0: no mask).  One launch for
 * autograd's  linear2.backward -> dropout.backward -> GELU.backward  of nn.TransformerEncoder/DecoderLayer and the HF
 * feed-forward blocks (reference model.py:874-878, utils/wav2vec2.py). */
int msmd_gemm_actbwd(const void* A, const void* W, const void* Z, void* C, int M, int N, int K, int in_dtype, int out_dtype,
                     long lda, long ldw, int act, float p_drop, const unsigned long* rng_state, unsigned int site,
                     msmd_stream_t stream);

int msmd_gemm_batched2(const void* A, const void* W, void* C, int M, int N, int K, int in_dtype, int out_dtype,
                       long lda, long ldw, long ldc, int batch_outer, long strideA_o, long strideW_o, long strideC_o,
                       int batch_inner, long strideA_i, long strideW_i, long strideC_i, msmd_stream_t stream);

/* Weight-gradient GEMM: C (N, K) fp32 = A^T . B with A (M, N) bf16 and B (M, K) bf16, both row-major with the
 * contraction index as the slow axis (A = dZ, B = X of a Linear's backward: no transposed copies).  N, K, lda, ldb
 * multiples of 8; pointers 16-byte aligned.  colsum (N) fp32 or NULL: fused bias gradient sum_m A[m][n]
 * (batch must be 1).  B may be a windowed view (row r at (r / b_rows_per_window) * b_window_stride +
 * (r % b_rows_per_window) * ldb; 0 = plain) so a Conv1d weight gradient needs no unfolded copy.
 * C / colsum are fully overwritten, or added to when bit 0 of `accumulate` is set (gradient accumulation in place);
 * bits 8-15 of `accumulate` force the contraction split count (0 = chosen from the shape).  ws / ws_bytes: optional device workspace for
 * split-contraction partial products (msmd_gemm_tn_workspace() bytes fill the chip; NULL = unsplit).  Replaces what autograd computes for nn.Linear in the
 * reference's loss.backward() (training_script.py:196). */
int msmd_gemm_tn(const void* A, const void* B, float* C, float* colsum, int M, int N, int K, long lda, long ldb,
                 long ldc, int batch, long strideA, long strideB, long strideC, int b_rows_per_window,
                 long b_window_stride, int accumulate, void* ws, long ws_bytes, msmd_stream_t stream);
long msmd_gemm_tn_workspace(int M, int N, int K, int batch);

/* ------------------------------------------------------------------------------------------------
 * y = post_act(LayerNorm(act(x + residual)) * gamma + beta) + post_add      (row-wise over `cols`)
 *   `act` carries the pre-activation in its low byte and post_act in bits 8-15 (MSMD_ACT_* codes; the LayerNorm ->
 *   GELU of HF's layer-norm conv stack).  residual, post_add (fp32, cols) may be NULL.  eps as torch (1e-5).
 *   Biased variance.
 * Replaces: nn.LayerNorm at reference style_encoder.py:142,150,170; HF feature_projection.layer_norm,
 *   encoder.layer_norm, layers.N.{layer_norm,final_layer_norm}; decoder norm1-3 (model.py:874-878).
 */
int msmd_layernorm(const void* x, const void* residual, const float* gamma, const float* beta,
                   const float* post_add, void* y, int rows, int cols, float eps, int act,
                   int in_dtype, int out_dtype, msmd_stream_t stream);
/* y = LN_{gamma, beta}( LN_{pre_gamma, pre_beta}(x) + residual ), 16-bit rows: two consecutive post-LN LayerNorms with a branch
 * added in between (nn.TransformerDecoderLayer norm1 -> + cross-attention branch -> norm2, reference model.py:874-878) in one
 * launch; the inner result is rounded to the storage type as the separate launch would have stored it. */
int msmd_layernorm_pre(const void* x, const float* pre_gamma, const float* pre_beta, const void* residual,
                       const float* gamma, const float* beta, void* y, int rows, int cols, float eps, int dtype,
                       msmd_stream_t stream);

/* msmd_layernorm on fp32 input with the row ALSO written in MSMD_F16X2 split storage (y_split; cols % 32 == 0):
 * in the parity-grade speed mode a LayerNorm output is the next contraction's A operand (split) and the residual of
 * the block after it (fp32, y; may be NULL when no fp32 copy is needed).  Same reference lines as msmd_layernorm. */
int msmd_layernorm_f16x2(const float* x, const float* residual, const float* gamma, const float* beta,
                         const float* post_add, float* y, void* y_split, int rows, int cols, float eps, int act,
                         msmd_stream_t stream);

/* fp32 (rows, cols) with leading dimension ldx -> MSMD_F16X2 split rows of cols_out >= cols logical columns
 * (cols_out % 32 == 0; the padding columns are zero), and back.  Streaming conversions for tensors that enter the
 * parity-grade speed mode from fp32 producers (weights at pack time, small glue tensors); the hot producers
 * (msmd_layernorm_f16x2, msmd_gemm with out_dtype MSMD_F16X2, msmd_conv0_gn_gelu) write split rows themselves. */
int msmd_split_f16x2(const float* x, void* y, long rows, int cols, long ldx, int cols_out, msmd_stream_t stream);
int msmd_unsplit_f16x2(const void* x, float* y, long rows, int cols, int cols_in, long ldy, msmd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused softmax attention for short sequences (Tk <= 512), head_dim 64:
 *   O[b, t, h*64:(h+1)*64] = softmax(scale * Q_h K_h^T  (masked -> -inf)) V_h
 *   Q/K/V/O rows are addressed as base + b*bstride + t*tstride + h*64 (elements).
 *   mask: (Tq, Tk) bytes, nonzero = masked out, or NULL.
 * Replaces: HF Wav2Vec2Attention (called from utils/wav2vec2.py:111), nn.MultiheadAttention inside
 *   nn.TransformerDecoderLayer (model.py:874-878,956) and nn.TransformerEncoderLayer (style_encoder.py:158).
 */
/* msmd_attention that additionally READS up to four byte ranges (16-byte aligned; e.g. the weight matrices of the GEMMs that
 * follow in the layer) and drops the data: the ranges are in the memory-side cache when those kernels start.  On MI355X a
 * GEMM whose operands come from HBM instead of the 256 MB Infinity Cache runs 20 % slower (6400 x 768 x 3072: 51 vs 43 us),
 * and one forward step streams ~1.5 GB through that cache between two uses of a weight.  Host arrays, read at launch. */
int msmd_attention_prefetch(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq, int Tk,
                            long q_bstride, long q_tstride, long k_bstride, long k_tstride, long v_bstride,
                            long v_tstride, long o_bstride, long o_tstride, float scale, const uint8_t* mask, int dtype,
                            const void* const* prefetch_ptrs, const long* prefetch_bytes, int n_prefetch,
                            msmd_stream_t stream);
int msmd_attention(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq, int Tk,
                   long q_bstride, long q_tstride, long k_bstride, long k_tstride, long v_bstride,
                   long v_tstride, long o_bstride, long o_tstride, float scale, const uint8_t* mask,
                   int dtype, msmd_stream_t stream);

/* msmd_attention on MSMD_F16X2 split operands (the parity-grade speed mode): Q / K / V as written by msmd_gemm with
 * out_dtype MSMD_F16X2, both products as three f16 MFMAs per k-step, softmax in fp32 (expf); O in fp32 or in split
 * storage (out_dtype) for the out-projection GEMM.  Strides in logical elements, multiples of 32. */
int msmd_attention_f16x2(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq, int Tk,
                         long q_bstride, long q_tstride, long k_bstride, long k_tstride, long v_bstride,
                         long v_tstride, long o_bstride, long o_tstride, float scale, const uint8_t* mask,
                         int out_dtype, msmd_stream_t stream);


/* Person-token cross-attention query, projection + Tq = 1 attention fused (one wave per sequence and head):
 *   out[n, h*64:(h+1)*64] = softmax(scale * (x[n] Wq_h^T + bq_h) K_h[n]^T) V_h[n]     (no mask, head_dim 64, Tk <= 512)
 * x row n at x + n*x_seq_stride (d elements); Wq (d, d) row-major; K/V rows at base + n*kv_bstride + t*kv_tstride + h*64;
 * out (N, d) contiguous; all of `dtype` (fp32 accumulation throughout).
 * Replaces, for row 0 of the sequence (the person / diffusion-step token) under the diagonal alignment mask: the
 * q-projection and softmax of nn.MultiheadAttention in the reference's TransformerDecoderLayer (model.py:874-878,956). */
int msmd_person_query_attention(const void* x, long x_seq_stride, const void* Wq, const float* bq, const void* K,
                                const void* V, long kv_bstride, long kv_tstride, void* out, int N, int H, int Tk, int d,
                                float scale, int dtype, msmd_stream_t stream);
/* msmd_person_query_attention with the LayerNorm in front of the query projection folded in (the decoder's norm1 in the
 * sampler's diagonal path): x row 0 is un-normalised, Wq / bq are the gamma / beta folded operands (as for msmd_gemm_ln's
 * operand form), wq_colsum[r] = sum_k Wq'[r][k]; mean / rstd of the row are computed in the kernel. */
int msmd_person_query_attention_ln(const void* x, long x_seq_stride, const void* Wq, const float* bq,
                                   const float* wq_colsum, float ln_eps, const void* K, const void* V, long kv_bstride,
                                   long kv_tstride, void* out, int N, int H, int Tk, int d, float scale, int dtype,
                                   msmd_stream_t stream);

/* Training-mode attention forward: as msmd_attention with attention-probability dropout p_drop (HF
 * attention_dropout, nn.MultiheadAttention(dropout=0.1) inside the decoder / encoder layers).  The keep mask is
 * Philox4x32-10(seed = rng_state[0], step = rng_state[1], site, counter = ((b H + h) Tq + q) * 128 + 4 * (key / 32) + (key / 4) % 4):
 * one block = the 16-bit draws of the 8 keys {16 f + 4 fq + e, f in a fragment pair} (even fragment: low halves of the 4 words, odd:
 * high halves); keep <=> draw >= floor(65536 p).  msmd_attention_bwd regenerates the same mask.  rng_state is DEVICE memory so a
 * captured hipGraph draws fresh masks on every replay once the host side advances the step.  Tk <= 512. */
int msmd_attention_dropout(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq, int Tk,
                           long q_bstride, long q_tstride, long k_bstride, long k_tstride, long v_bstride,
                           long v_tstride, long o_bstride, long o_tstride, float scale, const uint8_t* mask,
                           float p_drop, const unsigned long* rng_state, unsigned int site, int dtype,
                           msmd_stream_t stream);
/* msmd_attention_dropout with msmd_attention_prefetch's byte ranges. */
int msmd_attention_dropout_prefetch(const void* Q, const void* K, const void* V, void* O, int B, int H, int Tq, int Tk,
                                    long q_bstride, long q_tstride, long k_bstride, long k_tstride, long v_bstride,
                                    long v_tstride, long o_bstride, long o_tstride, float scale, const uint8_t* mask,
                                    float p_drop, const unsigned long* rng_state, unsigned int site, int dtype,
                                    const void* const* prefetch_ptrs, const long* prefetch_bytes, int n_prefetch,
                                    msmd_stream_t stream);

/* y = x * keep / (1 - p) (+ residual): nn.Dropout in training mode (HF hidden / activation / feat_proj dropout,
 * decoder-layer dropout1-3, PositionalEncoding dropout, style-encoder dropouts; utils/model_common.py:101,
 * style_encoder.py:139-167).  Same Philox stream as above indexed by element / 4; calling it on dy with the same
 * (rng_state, site) is the backward. */
int msmd_dropout(const void* x, const void* residual, void* y, long n, float p, const unsigned long* rng_state,
                 unsigned int site, int dtype, msmd_stream_t stream);

/* Fused attention backward (bf16, head_dim 64, Tk <= 256): recomputes P per 64-row query tile and writes
 * dQ / dK / dV; P, dP and transposed operands never touch HBM.  Same addressing convention as msmd_attention
 * (base + b*bstride + t*tstride + h*64; strides multiples of 8 elements, bases 16-byte aligned); dQ / dK / dV may
 * be slices of one packed gradient buffer.  Replaces autograd of the attention modules listed above under the
 * reference's loss.backward() (training_script.py:196).  p_drop / rng_state / site: attention-probability dropout
 * exactly as the forward msmd_attention_dropout drew it (0 / NULL: none). */
int msmd_attention_bwd(const void* Q, const void* K, const void* V, const void* dO, void* dQ, void* dK, void* dV,
                       int B, int H, int Tq, int Tk, long q_bstride, long q_tstride, long k_bstride,
                       long k_tstride, long v_bstride, long v_tstride, long do_bstride, long do_tstride,
                       long dq_bstride, long dq_tstride, long dk_bstride, long dk_tstride, long dv_bstride,
                       long dv_tstride, float scale, const uint8_t* mask, float p_drop,
                       const unsigned long* rng_state, unsigned int site, msmd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Audio front end.  pad plan = (reflect_len applied twice per side, replicate_len 0/1), computed on
 * the host exactly as reference utils/model_common.py:110-123.
 */
/* out (B, L + 4*reflect_len + 2*replicate_len) fp32 = pad_audio(audio (B, L)). */
int msmd_pad_audio(const float* audio, float* out, int B, int L, int reflect_len, int replicate_len,
                   msmd_stream_t stream);

/* conv0 (1->C, k=10, s=5, no bias) statistics for GroupNorm(C groups): stats (B, C, 2) = {mean, rstd}
 * over the T0 = (Lp - 10)/5 + 1 output frames, reading the UNPADDED audio through the pad map.
 * w0: (C, 10) fp32.  Replaces HF Wav2Vec2GroupNormConvLayer (called at utils/wav2vec2.py:79). */
#define MSMD_CONV0_SPLITS 16 /* ws: (B, MSMD_CONV0_SPLITS, 66) fp32 partial signal moments (sum, S[10], R[55]); at least
                              * B*16*66 floats.  The conv response is linear in the signal, so per-channel mean / variance
                              * follow from these 66 numbers per clip. */
int msmd_conv0_stats(const float* audio, const float* w0, float* stats, float* ws, int B, int L, int reflect_len,
                     int replicate_len, int C, float eps, msmd_stream_t stream);

/* out (B, T0, C) = GELU(GroupNorm(conv0(pad(audio)))) with the statistics above. */
int msmd_conv0_gn_gelu(const float* audio, const float* w0, const float* stats, const float* gamma,
                       const float* beta, void* out, int B, int L, int reflect_len, int replicate_len,
                       int C, int out_dtype, msmd_stream_t stream);

/* conv0 of the feat_extract_norm="layer" stack (hubert-large-style configs; HF HubertLayerNormConvLayer, reached from
 * utils/hubert.py:22 `self.feature_extractor(input_values)`): out[b][t][:] = GELU(LayerNorm_c(bias + conv1d(k=10,
 * s=5)(pad_audio(x)))) channels-last, C = 512, pad_audio fused into the loads as above. */
int msmd_conv0_ln_gelu(const float* audio, const float* w0, const float* bias, const float* gamma, const float* beta,
                       void* out, int B, int L, int reflect_len, int replicate_len, int C, float eps, int out_dtype,
                       msmd_stream_t stream);

/* Linear resample along time of a channels-last tensor with F.interpolate(mode='linear',
 * align_corners=False) semantics: y (B, T_out, C) from the first T_crop frames of x (B, T_in, C).
 * Replaces utils/wav2vec2.py:57-63,82-84 and model.py:260. */
int msmd_interp_linear(const void* x, void* y, int B, int T_in, int T_crop, int T_out, int C, int dtype,
                       msmd_stream_t stream);

/* Regroup (B, T, G*Cg) channels-last into the zero-padded group-major layout (B, G, T + 2*pad, Cg_out >= Cg; the
 * extra channels are zero) the positional grouped conv reads as G windowed GEMMs.  out_dtype = dtype, or MSMD_F16X2
 * from fp32 input (Cg_out % 32 == 0: split storage keeps 32-element blocks whole, so 48 channels are padded to 64). */
int msmd_group_pad(const void* x, void* y, int B, int T, int G, int Cg, int Cg_out, int pad, int dtype, int out_dtype,
                   msmd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Denoiser glue (reference model.py:931-951, 961-996, 231-236, 404-432).
 */
/* feats (N, 1+Lp+L, Kpad): row 0 is left to the caller (person token); rows 1.. = [prev_motion ; motion]
 * (dm cols) ++ indicator col (0 for prev rows) ++ zero pad up to Kpad.  motion: (N, L, dm), prev: (N, Lp, dm),
 * indicator (N, L) fp32 or NULL.  x_t = c0[n]*motion + c1[n]*eps when eps != NULL (q-sample, model.py:231-236). */
int msmd_denoiser_pack_input(const float* motion, const float* eps, const float* c0, const float* c1,
                             const float* prev_motion, const float* indicator, void* feats, int N, int L,
                             int Lp, int dm, int Kpad, int motion_batch, int out_dtype, msmd_stream_t stream);

/* x (N, T, d) += pe (T, d) (learned PE, model.py:949); row 0 is REPLACED by tok0[n] + row0_add + pe[0]
 * (tok0 (N, d) = person projection (+ step embedding); row0_add (d) optional shared step embedding). */
int msmd_add_pe_token(void* x, const float* pe, const void* tok0, const void* row0_add, int N, int T, int d,
                      int dtype, msmd_stream_t stream);

/* out (N, L, dm) fp32 = dyn[:, :, :dm] + sum_b alpha_b * static_b (face dims) / sum_b static_b (last 3 dims)
 * dec: (N, L, dm+nb) decoder head output (row stride ld_dec); stat: (Ns, nb, dm) static bases, Ns in {N, N/entries}.
 * (model.py:961-996.)  use_head_alpha bit 0: weight the last 3 (head pose) dims too; bit 1: alpha = sigmoid(alpha)
 * (regularize_alpha = 'sigmoid', model.py:973-974). */
int msmd_heads_static_mix(const void* dec, long ld_dec, const void* stat, float* out, int N, int L, int dm,
                          int nb, int stat_batch, int use_head_alpha, int dtype, msmd_stream_t stream);

/* One CFG + DDPM ancestral update (model.py:396-432), in place on x (B, L, dm) fp32.
 * res: (n_entries*B, Lp+L, dm) fp32 denoiser outputs; scales[n_entries-1]; coefficients on the host
 * (c0, c1, sigma) as the reference computes them; z (B, L, dm) or NULL for the last step.
 * mode: 0 = incremental, 1 = independent (in-place accumulation order of the reference). target: 0 sample, 1 noise. */
int msmd_cfg_ddpm_step(float* x, const float* res, const float* z, const float* scales, int n_entries,
                       int B, int L, int Lp, int dm, int mode, int target, float c0, float c1, float sigma,
                       msmd_stream_t stream);

/* hipGraph-capturable sampler step (no per-step host scalars): the step index t lives on the device.
 * msmd_sampler_step_select: emb_row (d) = emb_all[t], coefs (3) = coef_table[t] = (c0, c1, sigma_t), then t -= 1.
 * msmd_cfg_ddpm_step_dev: msmd_cfg_ddpm_step with (c0, c1, sigma) read from `coefs` on the device
 * (sigma_1 must be stored as 0: the reference uses z = 0 at t = 1, model.py:378-381). */
int msmd_sampler_step_select(const void* emb_all, const float* coef_table, int* t_dev, void* emb_row,
                             float* coefs, int d, int dtype, msmd_stream_t stream);
int msmd_cfg_ddpm_step_dev(float* x, const float* res, const float* z, const float* scales, const float* coefs,
                           int n_entries, int B, int L, int Lp, int dm, int mode, int target, msmd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * FLAME: blendshapes + pose correctives + joint regression + Rodrigues + kinematic chain + skinning
 * (reference utils/lbs.py:141-223, utils/flame.py:180-217) in two launches.
 *
 * One-time host-side packing (frame-invariant model constants, done by the caller at load time):
 *   JS   (NB+1, J*3): JS[0] = J_regressor . v_template, JS[1+l] = J_regressor . shapedirs[:, :, l]
 *                     (joint regression is linear in the shape, utils/lbs.py:185-189);
 *   dirs (3, Kp, Vp): dirs[c][k][v] = shapedirs[v][c][k] for k < NB, posedirs[k-NB][3v+c] for the next
 *                     (J-1)*9 rows, 0 beyond (Kp = 192, Vp = V rounded up to 64);
 *   v_template (3, Vp) and lbs_weights (J, Vp) as coordinate / joint planes.
 *
 * msmd_lbs_prepare: per frame, coef (B, Kp) = [betas | pose_feature = (R[1:] - I) | 0], the relative rigid
 *   transforms A (B, J, 12) (3x4 row-major) and optionally the posed joints (B, J, 3).
 *   pose: (B, J*3) axis-angle, or (B, J*9) rotation matrices when pose_is_matrix != 0.
 *   coef_hl (B, 2, Kp) bf16 (msmd_lbs_skin_bf16x3's operand) and at_tiles (msmd_lbs_skin_v2's skin_tiles records,
 *   J = 5 and Kp = 192 only) are optional outputs (NULL = skip).
 * msmd_lbs_skin: verts (B, V, 3) = sum_j w[v][j] A[b][j] . [v_template + coef . dirs ; 1].
 */
int msmd_lbs_prepare(const float* betas, const float* pose, const float* JS, const int* parents,
                     float* coef, void* coef_hl, float* A, float* joints, void* at_tiles, int B, int NB, int J, int Kp,
                     int pose_is_matrix, msmd_stream_t stream);
/* FLAME.forward's inputs (utils/flame.py:180-212) straight into msmd_lbs_skin_v2's tile records: coefficient row =
 * [shape (B, NS) | expr (B, NE)], pose6 (B, 6) = [global | jaw] axis-angle with the identity neck, eye (B, 6) or NULL
 * (identity), global rotation dropped when ignore_global_rot != 0 -- no concatenated betas / full_pose tensors.
 * coef (B, 192), A (B, 5, 12), joints (B, 5, 3): optional outputs (NULL = skip).  J = 5, Kp = 192.
 * One subject, many frames: with shape_varies (1 int) and v_template_folded (3, Vp) given (plus dirs (3, 192, Vp) and
 * v_template (3, Vp) to build it from), the call also writes  v_template_folded = v_template + sum_{k < 96} shape[0][k]
 * dirs[k]  and sets *shape_varies to whether any frame's first 96 shape coefficients differ from frame 0's -- on the
 * device, nothing reads back.  msmd_lbs_skin_v2 handed both skips those K groups when *shape_varies == 0. */
int msmd_flame_prepare(const float* shape, const float* expr, const float* pose6, const float* eye, const float* JS,
                       const int* parents, float* coef, float* A, float* joints, void* skin_tiles, int B, int NS, int NE,
                       int ignore_global_rot, int* shape_varies, float* v_template_folded, const float* dirs,
                       const float* v_template, int Vp, msmd_stream_t stream);
int msmd_lbs_skin(const float* coef, const float* A, const float* v_template, const float* dirs,
                  const float* lbs_weights, float* verts, int B, int J, int V, int Vp, int Kp,
                  msmd_stream_t stream);
/* Same contraction as msmd_lbs_skin in split-bf16 form (fp32 accumulation): coef_hl (B, 2, Kp) bf16 = hi/lo parts
 * written by msmd_lbs_prepare (may be NULL there when unused); dirs_hl (2, 3, Kp/8, Vp, 8) bf16 = hi/lo parts of
 * dirs regrouped in 8-element K octets.  coef.dirs ~= hi.hi + hi.lo + lo.hi: relative error 2^-16 per product
 * (measured max-abs-err on FLAME vertices vs the fp32 kernel: see tests), 5x fewer MFMA cycles. */
int msmd_lbs_skin_bf16x3(const void* coef_hl, const float* A, const float* v_template, const void* dirs_hl,
                         const float* lbs_weights, float* verts, int B, int J, int V, int Vp, int Kp,
                         msmd_stream_t stream);

/* msmd_lbs_skin_bf16x3 with the per-(vertex, frame) joint blend T = sum_j w_j A_j on the matrix pipe as well: for each
 * of the 12 components of the 3x4 transforms ONE v_mfma_f32_16x16x32_f16 contracts the five joints (both sides split
 * into fp16 hi + lo: 15 of the 32 K slots), its accumulator landing where the blendshape product puts p(vertex,
 * frame).
 * skin_tiles: ceil(B / 16) records of 18 432 bytes, one per 16 frames, written by msmd_lbs_prepare (its at_tiles
 *   argument) or msmd_lbs_pack and read by 18 contiguous one-KiB LDS-DMA pieces:
 *     [0, 12 288)       coefficients as bf16 hi then lo, [chunk = (hi|lo) * 24 + k / 8][frame % 16][k % 8]
 *     [12 288, 18 432)  blend rows, fp16, [component m of the 3x4][slot / 8][frame % 16][slot % 8] with the 16 K slots
 *                       [Ah_0..4 | Al_0..4 | Ah_0..4 | 0];
 *   frames beyond B - 1 of the last record repeat frame B - 1.
 * 128 vertices per workgroup (8 waves), 16-frame tiles through a 4-deep LDS-DMA ring, one workgroup barrier per two
 * tiles; Vp = padded vertex count of the constant planes (any value >= V).  HBM-bound target: 60 936 algorithmic bytes
 * per frame.  shape_varies / v_template_folded: NULL, or msmd_flame_prepare's outputs (see there).
 * Reference: utils/lbs.py:141-223. */
int msmd_lbs_skin_v2(const void* skin_tiles, const float* v_template, const void* dirs_hl, const float* lbs_weights,
                     float* verts, int B, int J, int V, int Vp, int Kp, const int* shape_varies,
                     const float* v_template_folded, msmd_stream_t stream);

/* msmd_lbs_skin_v2 with fp16 vertices (opt-in; BASELINE configs[4] names the fp16 LBS pass): verts16 (B, V_ld, 3) fp16, V_ld
 * even and >= V (rows of V * 3 halves would put every other frame on a 2-byte boundary; slot V of a row receives a copy of
 * vertex V - 1).  Half the store stream that bounds the fp32 kernel: 30 144 + 660 algorithmic bytes per frame at V = 5023.
 *   single_plane == 0: skin_tiles / dirs = msmd_lbs_skin_v2's operands (18 KB records, dirs_hl): the fp32 kernel's arithmetic
 *     with one fp16 rounding at the store (equal to its output rounded once);
 *   single_plane != 0: skin_tiles = the 12 KB records of msmd_lbs_tiles_f16, dirs = ONE fp16 plane (3, Kp / 8, Vp, 8) of the
 *     blendshape directions: one MFMA per K group and coordinate instead of three; |error| <= 2^-11 |v| + 2^-10 sum_k
 *     |coef_k| |dirs_k| (the operands' own fp16 rounding, on the un-skinned blendshape offset) + the fp32 kernel's 5e-6.
 * Reference: utils/lbs.py:210-221 (fp32 there). */
int msmd_lbs_skin_v2_f16(const void* skin_tiles, const float* v_template, const void* dirs, const float* lbs_weights,
                         void* verts16, int B, int J, int V, int V_ld, int Vp, int Kp, const int* shape_varies,
                         const float* v_template_folded, int single_plane, msmd_stream_t stream);
/* msmd_lbs_skin_v2's tile records (18 432 bytes per 16 frames) -> the single-plane form's (12 288 bytes: coefficients hi + lo
 * as one fp16 number [k / 8][frame % 16][k % 8], then the blend rows). */
int msmd_lbs_tiles_f16(const void* skin_tiles, void* tiles16, int B, msmd_stream_t stream);

/* Training through FLAME (the reference's use_vertex_space branch: training_script.py:167-176 -> utils/common.py:486-513
 * -> utils/lbs.py:141-223, differentiated by autograd there).
 * msmd_lbs_skin_v2_train: msmd_lbs_skin_v2 that also stores the un-skinned vertices v_posed = template + coef . dirs.
 * msmd_lbs_pack: (coef (B, Kp), A (B, 5, 12)) fp32 -> skin_tiles (and coef_hl (B, 2, Kp) bf16 when not NULL), for
 *   per-frame kinematics computed elsewhere (the differentiable pass builds them with autograd on (B, 5, 3, 3)-sized
 *   tensors).  Kp = 192.
 * msmd_lbs_skin_bwd: given grad_verts (B, V, 3) and v_posed: dp_planes (B, 3, Vp) = (sum_j w_j R_j)^T g (the A operand
 *   of dcoef = dp . dirs^T, an msmd_gemm) and dA (B, 5, 12) = sum_v w_j(v) g(v) [v_posed(v) ; 1]^T. */
int msmd_lbs_skin_v2_train(const void* skin_tiles, const float* v_template, const void* dirs_hl,
                           const float* lbs_weights, float* verts, float* v_posed, int B, int J, int V, int Vp, int Kp,
                           msmd_stream_t stream);
int msmd_lbs_pack(const float* coef, const float* A, void* coef_hl, void* skin_tiles, int B, int Kp, msmd_stream_t stream);
int msmd_lbs_skin_bwd(const float* grad_verts, const float* v_posed, const float* A, const float* lbs_weights,
                      float* dp_planes, float* dA, int B, int J, int V, int Vp, msmd_stream_t stream);

/* Landmarks by barycentric interpolation (utils/lbs.py:102-138): out (B, L, 3).
 * faces (F,3) int32; lmk_faces_idx (B or 1, L) int32 with batch stride idx_bstride (0 = shared);
 * bary (B or 1, L, 3) with batch stride bary_bstride. */
int msmd_landmarks(const float* verts, const int* faces, const int* lmk_faces_idx, long idx_bstride,
                   const float* bary, long bary_bstride, float* out, int B, int V, int L, msmd_stream_t stream);

/* Dynamic-contour LUT row (utils/flame.py:126-172): row (B) int32 from the neck-chain yaw; full_pose is (B, J*3)
 * axis-angle (pose2rot=True) or (B, J*9) rotation matrices (pose_is_matrix, pose2rot=False). */
int msmd_dynamic_lmk_row(const float* full_pose, const int* neck_chain, int n_chain, int* row, int B, int J,
                         int pose_is_matrix, msmd_stream_t stream);

/* batch_rodrigues (utils/lbs.py:270-301): R (N, 3, 3) from rot_vecs (N, 3). */
int msmd_batch_rodrigues(const float* rot_vecs, float* R, int N, msmd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Rotation conversions (reference utils/rotation_conversions.py:38-569), elementwise over n items.
 * op codes below; `conv` packs an Euler convention as 3 axis indices (X=0,Y=1,Z=2): a0 | a1<<2 | a2<<4.
 */
#define MSMD_ROT_QUAT_TO_MAT 0
#define MSMD_ROT_MAT_TO_QUAT 1
#define MSMD_ROT_AA_TO_QUAT 2
#define MSMD_ROT_QUAT_TO_AA 3
#define MSMD_ROT_AA_TO_MAT 4
#define MSMD_ROT_MAT_TO_AA 5
#define MSMD_ROT_6D_TO_MAT 6
#define MSMD_ROT_MAT_TO_6D 7
#define MSMD_ROT_AA_TO_6D 8
#define MSMD_ROT_EULER_TO_MAT 9
#define MSMD_ROT_MAT_TO_EULER 10
#define MSMD_ROT_QUAT_STANDARDIZE 11
#define MSMD_ROT_QUAT_INVERT 12
#define MSMD_ROT_QUAT_RAW_MUL 13 /* in2 = second quaternion */
#define MSMD_ROT_QUAT_MUL 14
#define MSMD_ROT_QUAT_APPLY 15 /* in2 = points (n,3) */
int msmd_rotation_convert(int op, const float* in, const float* in2, float* out, long n, int conv,
                          msmd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Losses / training-step pieces (reference utils/common.py:198-620, 443-454, 769-832; training_script.py:548-551).
 */
/* out[0] = scale * mean over valid (n, t) of mean_{c in [c_lo, c_hi)} crit(D^order gt, D^order pred)
 *   gt, pred: (N, T, C) fp32; D^order = order-th temporal difference (0 = value, 1 = velocity, 2 = smoothness);
 *   frame t of the differenced sequence is valid when mask[t + order] holds, mask[tm] = tm < prefix ||
 *   tm - prefix < end_idx[n] (end_idx NULL: all valid); a NEGATIVE prefix masks the first |prefix| frames OUT instead
 *   (--no_constrain_prev, utils/common.py:382-385).  criterion 0 = squared error, 1 = absolute error.
 *   mode 1: compare D^order pred against 0 (the smoothness term).  acc_ws: 2 doubles of scratch.
 *   out is NaN when no frame is valid (the reference returns None there). */
int msmd_masked_seq_loss(const float* gt, const float* pred, const int* end_idx, float* out, double* acc_ws,
                         int N, int T, int C, int c_lo, int c_hi, int order, int prefix, int criterion, int mode,
                         float scale, msmd_stream_t stream);
/* d msmd_masked_seq_loss / d pred, ADDED into grad_pred (N, T, C) (channels outside [c_lo, c_hi) untouched): acc_ws is
 * the forward call's workspace (its valid-row count), upstream the incoming 0-dim gradient on the device.  What the
 * reference gets from autograd through utils/common.py:198-620 (training through the vertex-space loss). */
int msmd_masked_seq_loss_bwd(const float* gt, const float* pred, const int* end_idx, const double* acc_ws,
                             const float* upstream, float* grad_pred, int N, int T, int C, int c_lo, int c_hi, int order,
                             int prefix, int criterion, int mode, float scale, msmd_stream_t stream);
/* out[0] = -0.5 * sum(1 + logvar - mu^2 - exp(logvar)) over n elements. */
int msmd_kl_loss(const float* mu, const float* logvar, float* out, double* acc_ws, long n, msmd_stream_t stream);
/* In place: x (N, L, inner)[n, end_idx[n]*unit :, :] = 0 (or the last kept row when replicate != 0). */
int msmd_truncate_rows(float* x, const int* end_idx, int N, int L, int inner, int unit, int replicate,
                       msmd_stream_t stream);
/* One Adam step (torch.optim.Adam defaults) on flat fp32 arenas of n elements; grad is multiplied by grad_scale
 * first (1/world_size after a sum all-reduce).  step is the 1-based step count for bias correction. */
int msmd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                   float beta1, float beta2, float eps, int step, float grad_scale, msmd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Gradient exchange over RCCL / xGMI: the one exchange step on the path (SURVEY.md 8b "msmd_allreduce_bucket (RCCL)", 8e;
 * the reference steps one process, training_script.py:190-199 -- its data-parallel form sums the gradient of every
 * parameter over the ranks before optimizer.step()).  One process per GPU, one communicator per process; librccl is
 * resolved with dlopen at first use (csrc/comm.hip), so nothing here costs a process that never exchanges.
 *   msmd_comm_unique_id: rank 0 fills 128 bytes that every rank passes to msmd_comm_init (sent over any side channel);
 *   msmd_comm_init: collective over the job; the calling thread's current HIP device is the rank's GPU;
 *   msmd_allreduce_bucket: buf[0 .. n) <- SUM over ranks, in place, enqueued on `stream` (the reducer's side stream, behind
 *     an event of the compute stream: overlapped with the rest of backward); dtype MSMD_F32 (a bucket of the flat gradient
 *     arena), MSMD_BF16 / MSMD_F16 (its 16-bit staging copy: half the bytes over xGMI).
 *   msmd_comm_version: ncclGetVersion of the librccl in use (a copy the process already mapped -- torch's -- is adopted
 *     before anything is loaded), or -(1000 + n).
 * Return codes: 0, an ncclResult_t, or 1000 + n when librccl could not be loaded (1) / lacks a symbol (2) / reports a
 * major version other than the 2.x whose rccl.h slice csrc/comm.hip restates (3). */
int msmd_comm_version(void);
int msmd_comm_unique_id(void* id_out_128_bytes);
int msmd_comm_init(void** comm_out, int world, int rank, const void* id_128_bytes);
int msmd_comm_destroy(void* comm);
int msmd_allreduce_bucket(void* comm, void* buf, long n, int dtype, msmd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Backward-pass building blocks (training; reference training_script.py:195).  dgrad / wgrad are msmd_gemm calls on
 * transposed operands: dX = dZ . W (W^T as the (K, N) operand), dW = dZ^T . X (both operands transposed).
 */
/* y[z][c][r] = x[z][r][c] for batch * batch_inner matrices; matrix z = zo * batch_inner + zi starts at
 * base + zo * stride + zi * stride_i (elements). */
int msmd_transpose(const void* x, void* y, int rows, int cols, long ldx, long ldy, int batch, long stride_x,
                   long stride_y, int batch_inner, long stride_x_i, long stride_y_i, int dtype, msmd_stream_t stream);
/* out[c] (+)= sum_r x[r][c] in fp32 (bias gradient, weight-norm reductions).  With a workspace of
 * msmd_colsum_workspace(rows, cols) bytes the sum is deterministic (per-row-block partial sums added in block order by a
 * second launch); ws = NULL: the row blocks meet through float atomics. */
long msmd_colsum_workspace(long rows, int cols);
int msmd_colsum(const void* x, float* out, long rows, int cols, long ld, int accumulate, int dtype, void* ws, long ws_bytes,
                msmd_stream_t stream);
/* y = act(z);  dz = dy * act'(z)  (exact erf GELU / ELU derivatives). */
int msmd_act_fwd(const void* z, void* y, long n, int act, int dtype, msmd_stream_t stream);
int msmd_act_bwd(const void* dy, const void* z, void* dz, long n, int act, int dtype, msmd_stream_t stream);
/* dz = dropout_mask(dy) * act'(z): backward of y = dropout_p(act(z)) in one pass (mask as msmd_dropout / msmd_gemm_ex). */
int msmd_act_bwd_dropout(const void* dy, const void* z, void* dz, long n, int act, float p,
                         const unsigned long* rng_state, unsigned int site, int dtype, msmd_stream_t stream);
/* LayerNorm backward for y = LN(x)*gamma + beta (x = the LN input, residual already added):
 * dx (rows, cols); dgamma / dbeta (cols) fp32 are ACCUMULATED into (zero them for a fresh gradient).
 * ws: optional msmd_layernorm_bwd_workspace() bytes for per-workgroup partial sums (NULL: fp32 atomics). */
int msmd_layernorm_bwd(const void* dy, const void* x, const float* gamma, void* dx, float* dgamma, float* dbeta,
                       int rows, int cols, float eps, int dtype, void* ws, long ws_bytes, msmd_stream_t stream);
long msmd_layernorm_bwd_workspace(int rows, int cols);
/* msmd_layernorm_bwd that also writes dx_drop = dropout_mask(dx) / (1 - p) (mask of msmd_dropout / msmd_gemm_ex at
 * (rng_state, site): index = element / 4): the gradient the Linear in front of a post-LN block's LayerNorm wants
 * (x = residual + dropout_p(Linear(..)); reference: nn.TransformerDecoderLayer / HF Wav2Vec2EncoderLayer under
 * loss.backward(), training_script.py:196).  cols % 4 == 0, 16-byte aligned tensors. */
int msmd_layernorm_bwd_dropout(const void* dy, const void* x, const float* gamma, void* dx, void* dx_drop, float* dgamma,
                               float* dbeta, int rows, int cols, float eps, float p_drop, const unsigned long* rng_state,
                               unsigned int site, int dtype, void* ws, long ws_bytes, msmd_stream_t stream);
/* In place row softmax of scale * s over the first `cols` entries of rows with stride ld (the ld - cols padding
 * columns are zeroed), optional (Tq, cols) byte mask (row r uses mask row r % Tq);
 * backward (in place on dP): dS = scale * P o (dP - rowsum(dP o P)). */
int msmd_softmax_rows(void* s, const uint8_t* mask, long rows, int cols, int ld, int Tq, float scale, int dtype,
                      msmd_stream_t stream);
int msmd_softmax_bwd_rows(const void* P, void* dP, long rows, int cols, int ld, float scale, int dtype,
                          msmd_stream_t stream);

/* Transposed unfold of the zero-padded group-major signal xp (B, G, Tp, Cg) for the grouped positional-conv
 * weight gradient: out (G, Kk*Cg, ld_out)[g][kk*Cg + ci][b*T + t] = xp[b][g][t + kk][ci] (columns >= B*T zero). */
int msmd_unfold_t(const void* xp, void* out, int B, int T, int Tp, int G, int Cg, int Kk, long ld_out, int dtype,
                  msmd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Small utilities.
 */
int msmd_cast(const void* x, void* y, long n, int in_dtype, int out_dtype, msmd_stream_t stream);
/* y (rows, cols_out) = x (rows, cols_in) zero-padded / truncated per row (dtype convert allowed). */
int msmd_pad_cols(const void* x, void* y, long rows, int cols_in, int cols_out, int in_dtype, int out_dtype,
                  msmd_stream_t stream);
/* y[r, :] = mean over T of x[b, t, :]  (B, T, C) -> (B, C) fp32 (style_encoder.py:189). */
int msmd_mean_time(const void* x, float* y, int B, int T, int C, int dtype, msmd_stream_t stream);

/* Dynamic thresholding of the denoiser output, in place (reference model.py:396-402 / 578-584):
 * s_n = clamp(torch.quantile(|res[n, -L:, :]|, ratio), dt_min, dt_max) ('linear' interpolation between order
 * statistics, ATen's lerp), res[n] <- clamp(res[n], -s_n, s_n) over all T_all frames.  res: (N, T_all, C) fp32,
 * L * C <= 20480.  No sort: bisection on the float bit pattern with block-wide counts. */
int msmd_dynamic_threshold(float* res, int N, int T_all, int L, int C, float ratio, float dt_min, float dt_max,
                           msmd_stream_t stream);

/* Mixed-precision weight refresh, ONE launch per optimizer step: for each of n_weights (N, K) fp32 matrices inside
 * the flat parameter arena `base`, write its bf16 cast (N, K) into cast_arena and its bf16 transpose (K, N) into
 * transposed_arena.  meta (n_weights, 6) int64 device array: [src offset, N, K, cast offset, transposed offset, first
 * tile index]; tiles are 32 x 32, total_tiles = sum of ceil(N/32) ceil(K/32).  Replaces the per-weight autocast copies
 * an AMP training step of the reference would make (training_script.py:163-201 runs fp32; this build trains in bf16). */
int msmd_cast_transpose_multi(const float* base, const long* meta, int n_weights, long total_tiles, void* cast_arena,
                              void* transposed_arena, msmd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Training-batch assembly from an HBM-resident corpus: both consecutive L-frame windows of B samples in one launch
 * (reference DatasetPickle.__getitem__ + collate_fn, datasets.py:251-368, 424-503).
 *   audio_flat / coef_flat: all clips back to back (coef rows of C floats); desc (B, 8) int64 per sample:
 *   [audio offset, audio length, coef row offset, coef rows, start frame of window 0, zero frames padded in front,
 *    zero audio samples padded in front, clip index]; clip_stats (n_clips, 2) = each clip's audio mean / std.
 *   out_audio (2, B, n_audio): (x - mean) / (std + 1e-5) inside the clip, 0 in padding, window w covering samples
 *   [int(start_w * unit), int(end_w * unit)) then padded / trimmed to n_audio; out_motion (2, B, L, C):
 *   (row - coef_mean) / (coef_std + 1e-9) with zero rows where the reference zero-pads (statistics may be NULL). */
int msmd_batch_windows(const float* audio_flat, const float* coef_flat, const long* desc, const float* clip_stats,
                       const float* coef_mean, const float* coef_std, float* out_audio, float* out_motion, int B, int L,
                       int C, int n_audio, double audio_unit, msmd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MSMD_HIP_H */
