/*
 * libs2s_mi355x — C ABI of the MI355X-native STT+TTS inference core (gfx950 only).
 *
 * This is the drop-in boundary for the hot path of huggingface/speech-to-speech
 * (SURVEY.md §8b).  The reference has no FFI of its own: its handlers call
 * third-party Python model objects.  Each group below names the reference call
 * site (path:line under the reference's src/speech_to_speech/) whose arithmetic
 * it replaces.  The Python host side (speech-to-speech_amd/) binds these with
 * ctypes and passes `tensor.data_ptr()` of PyTorch-ROCm tensors as plain
 * device pointers; no torch type appears in any signature.
 *
 * Conventions
 *   - every function returns int32_t: 0 = S2S_OK, <0 = error; the message for
 *     the calling thread's last error is s2s_last_error().  Nothing throws.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     the call returns without synchronising unless stated otherwise.
 *   - pointers named *_dev are device pointers, *_host are host pointers.
 *   - the caller owns every buffer it passes; handles own their own workspace.
 */
#ifndef S2S_MI355X_H
#define S2S_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S2S_OK 0
#define S2S_ERR_ARG (-1)
#define S2S_ERR_HIP (-2)
#define S2S_ERR_STATE (-3)
#define S2S_ERR_UNSUPPORTED (-4)

#define S2S_ABI_VERSION 5 /* 2: continuous-batching entry points, wide (<= 512 slot) frame loops, per-row codec positions; 3: Whisper token-loop slots; 4: s2s_logmel_parakeet_cached; 5: s2s_tts_set_trailing, s2s_logmel_parakeet_cached_rows */

/* element types of device buffers */
#define S2S_F32 0
#define S2S_BF16 1

/* GEMM epilogue activation codes */
#define S2S_ACT_NONE 0
#define S2S_ACT_GELU 1 /* exact erf GELU (Whisper, modeling_whisper.py:618-619) */
#define S2S_ACT_SILU 2 /* SiLU (FastConformer FFN; Qwen3 SwiGLU gate) */
#define S2S_ACT_RELU 3
#define S2S_ACT_SNAKE 4 /* SnakeBeta with per-channel tables (codec decoder; internal GEMM epilogues only) */

const char* s2s_last_error(void);
int32_t s2s_abi_version(void);
/* number of visible HIP devices (0 when there is no GPU); never fails */
int32_t s2s_device_count(void);

/* The autoregressive loops (TTS codec frames, Whisper decode steps) keep every per-step quantity on the
 * device and replay one captured hipGraph per step.  on = 0 forces plain stream launches (the same kernels,
 * the same results); graphs are also bypassed while kernel timing is enabled.  Default on; the environment
 * variable S2S_NO_GRAPHS=1 starts with it off. */
int32_t s2s_graphs_enable(int32_t on);

/* ------------------------------------------------------------------------- *
 * Log-mel front-ends.
 *   Whisper : replaces WhisperSTTHandler.prepare_model_inputs
 *             (STT/whisper_stt_handler.py:83-87 -> transformers
 *             WhisperFeatureExtractor._np_extract_fbank_features,
 *             feature_extraction_whisper.py:105-131).
 *   Parakeet: replaces the feature extraction inside nano_parakeet's
 *             model.transcribe (STT/parakeet_tdt_handler.py:621-623; algorithm
 *             restated from transformers ParakeetFeatureExtractor,
 *             feature_extraction_parakeet.py:101-127,252-278).
 * ------------------------------------------------------------------------- */
#define S2S_FRONTEND_WHISPER 0
#define S2S_FRONTEND_PARAKEET 1

/* mel_filters_host: float32 [n_bins][n_mels] row-major (n_bins = 201 Whisper,
 * 257 Parakeet).  max_batch bounds B of later calls. */
int32_t s2s_frontend_create(int32_t kind, int32_t n_mels, const float* mel_filters_host,
                            int32_t max_batch, void** out_frontend);
int32_t s2s_frontend_destroy(void* frontend);

/* pcm_dev: float32 [B][pcm_stride] mono 16 kHz in [-1,1]; n_samples_dev: int32[B]
 * valid samples per row (<= pcm_stride, <= 480000; longer audio is truncated to
 * 30 s exactly as the reference's extractor does).
 * out_dev: [B][n_mels][3000], element type out_dtype (S2S_F32 or S2S_BF16). */
int32_t s2s_logmel_whisper(void* frontend, void* stream, const float* pcm_dev,
                           int64_t pcm_stride, const int32_t* n_samples_dev, int32_t B,
                           void* out_dev, int32_t out_dtype);
/* Same features in the encoder's own layout: out_dev [B][3002][n_mels] time-major with one zero
 * row before and after each utterance (the conv stem reads it as overlapping GEMM rows). */
int32_t s2s_logmel_whisper_tm(void* frontend, void* stream, const float* pcm_dev,
                              int64_t pcm_stride, const int32_t* n_samples_dev, int32_t B,
                              void* out_dev, int32_t out_dtype);

/* Whisper features of a shorter window with optional zero-mean / unit-variance waveform normalisation: the feature half of the
 * reference's Smart-Turn end-of-turn check (VAD/smart_turn.py:130-143 = WhisperFeatureExtractor(chunk_length=8)(...,
 * do_normalize=True, max_length=128000); the ONNX classifier itself stays on the host).  window_samples: a multiple of 160 in
 * [400, 480000]; rows longer than the window are truncated to their FIRST window_samples samples (the extractor's rule; Smart-Turn
 * keeps the LAST 8 s before calling it, smart_turn.py:123-128 - the host adapter does that).  normalize != 0: every row is
 * replaced by (x - mean) / sqrt(var + 1e-7) over its valid samples, padding stays 0.  out_dev: [B][n_mels][window_samples / 160]. */
int32_t s2s_logmel_whisper_window(void* frontend, void* stream, const float* pcm_dev, int64_t pcm_stride,
                                  const int32_t* n_samples_dev, int32_t B, int32_t window_samples, int32_t normalize,
                                  void* out_dev, int32_t out_dtype);

/* out_dev: [B][t_max][n_mels] (time-major, zero beyond each row's valid
 * frames), n_frames_dev: int32[B] valid frames (= n_samples / 160).
 * t_max must be >= max(n_samples)/160 + 1. */
int32_t s2s_logmel_parakeet(void* frontend, void* stream, const float* pcm_dev,
                            int64_t pcm_stride, const int32_t* n_samples_dev, int32_t B,
                            int32_t t_max, void* out_dev, int32_t out_dtype,
                            int32_t* n_frames_dev);

/* The same features for a GROWING window (live partials: the reference re-transcribes the turn's window every 0.5 s,
 * STT/smart_progressive_streaming.py:87-157, STT/parakeet_tdt_handler.py:259-330).  cache_dev: float32 [B][t_max][n_mels], owned by the
 * caller, holding the UN-normalised log-mel frames of the rows' earlier, shorter prefixes; first_frame_dev int32[B]: frames below it are
 * taken from the cache, frames from it on are computed and written to the cache (0 = compute everything).  A frame depends on the 400
 * samples around it only, so with first_frame = max(0, (previous n_samples - 200) / 160) the result is bit-identical to
 * s2s_logmel_parakeet on the whole window (tests/test_parakeet_progressive_gpu.py); the per-feature normalisation always runs over the whole window.
 * With B = 1 the cache layout does not depend on t_max: one buffer serves a stream as its window grows. */
int32_t s2s_logmel_parakeet_cached(void* frontend, void* stream, const float* pcm_dev, int64_t pcm_stride,
                                   const int32_t* n_samples_dev, const int32_t* first_frame_dev, int32_t B, int32_t t_max,
                                   float* cache_dev, void* out_dev, int32_t out_dtype, int32_t* n_frames_dev);

/* The live partials of one TICK in one call (ABI v5): B growing windows, each with a sample buffer and a log-mel cache of its own.
 * pcm_rows_dev / cache_rows_dev: device arrays of B device pointers (row b: float32 samples so far; float32 [>= t_max][n_mels] cache).
 * new_pcm_dev (may be NULL: nothing to append): the new samples of all rows as one flat float32 block, row b's are
 * new_pcm[new_off[b] .. new_off[b + 1]) (new_off_dev int32[B + 1]) and are appended at sample old_n_dev[b] of its buffer before the
 * features are computed; n_samples_dev[b] = the row's sample count afterwards; first_frame_dev as above.  out_dev [B][t_max][n_mels]:
 * row b is bit-identical to s2s_logmel_parakeet_cached (and therefore to s2s_logmel_parakeet) on that window alone. */
int32_t s2s_logmel_parakeet_cached_rows(void* frontend, void* stream, int32_t B, int32_t t_max, float* const* pcm_rows_dev,
                                        float* const* cache_rows_dev, const int32_t* n_samples_dev, const int32_t* first_frame_dev,
                                        const float* new_pcm_dev, const int32_t* new_off_dev, const int32_t* old_n_dev, void* out_dev,
                                        int32_t out_dtype, int32_t* n_frames_dev);

/* ------------------------------------------------------------------------- *
 * Primitive operators (exported so the parity tests can pin each kernel
 * against the oracle; the model graphs below are built from the same kernels).
 * ------------------------------------------------------------------------- */

/* C[M,N] = act(A[M,K] @ W[N,K]^T + bias[N]) (+ residual[M,N])
 * A, W share `ab_dtype`; bias is f32 (may be NULL); residual is f32 (may be NULL,
 * may alias C when c_dtype is F32); C has `c_dtype`.  lda/ldc in elements.
 * F32 inputs use the exact-f32 MFMA (v_mfma_f32_16x16x4_f32), BF16 inputs the
 * bf16 MFMA with f32 accumulation. */
int32_t s2s_op_gemm(void* stream, const void* A, int64_t lda, const void* W, const float* bias,
                    const float* residual, void* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                    int32_t ab_dtype, int32_t c_dtype, int32_t act);

/* s2s_op_gemm with the decode-path switches exposed, so the parity tests reach every variant the engines launch:
 * flags bit 0: W = [gate rows | up rows] (N each), C = silu(gate) * up (fused SwiGLU, M <= 64);
 *       bit 1: allow split-K (only taken for an in-place f32 residual, act NONE);
 *       bit 2: A is the raw f32 residual stream and RMSNorm (its weight pre-folded into W's columns) runs inside the
 *              kernel with rms_eps (bf16, M <= 64). */
int32_t s2s_op_gemm_ex(void* stream, const void* A, int64_t lda, const void* W, const float* bias,
                       const float* residual, void* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                       int32_t ab_dtype, int32_t c_dtype, int32_t act, int32_t flags, float rms_eps);

/* One decode-step projection of the norm-free decoder layer (csrc/gemm_dec.hip; bf16, K >= 256, K % 32 == 0):
 *   acc = A[M,K] @ W[N,K]^T;  out = rstd[m] * acc + bias  with rstd[m] = rsqrt(sum_p ss_in[p*ss_stride + m] * inv_dim + eps)
 *   when ss_in != NULL (RMSNorm of the raw rows A, its weight pre-folded into W's columns), else out = acc + bias.
 *   W is FRAGMENT-MAJOR (s2s_op_to_frag): one contiguous 1 KiB block per (16-row tile, 32-wide k-step) in MFMA operand
 *   order, so that every wave load is one coalesced request; glu != 0: [gate tiles | up tiles], N rows each (N % 16 == 0).
 *   layout bit 0: A is fragment-major too (else row-major [M][lda]); bit 1: mode-0 C is written fragment-major (row width
 *   ldc); bit 2: mode-2 XB is written fragment-major (row width ldxb); bit 3: the weights are re-read by many launches in a
 *   row (code predictor, codec row slabs): plain cached loads instead of the default non-temporal streaming loads.
 *   M <= 128, or a multiple of 128 (row slabs of 128 walk grid.y; ss_stride >= M).
 *   mode 0: C bf16 = act(out)   (glu != 0: C = silu(out_gate) * out_up)
 *   mode 1: C f32  = act(out)
 *   mode 2: X f32 [M][ldx] += out in place; XB (optional) = bf16 copy of the new rows; ss_out (optional)
 *           [s2s_op_gemm_dec_parts(M,N,K,nf,layout)][ss_stride] = per-column-slab partial sums of squares of the new rows
 *           (the next projection's ss_in).  Deterministic: no atomics on data, fixed summation order.
 *   Two kernels share this entry point.  Slabs of 33 .. 128 rows with fragment-major A run the column-block / K-slice kernel
 *   (csrc/gemm_dec2.hip: a 4-wave workgroup owns 64 or 128 columns, the activations go through LDS once per workgroup, K is
 *   split across workgroups and the slices are added in slice order by the one that arrives last); everything else runs
 *   the 16-column kernel (csrc/gemm_dec.hip).  nf: 16-column fragments per workgroup of the latter (0 = chosen from the
 *   shape; > 0 also keeps the call on it).
 * Replaces the rmsnorm -> linear pairs of the Qwen3 decoder layer (transformers qwen3/modeling_qwen3.py; sibling code
 * predictor qwen3_omni_moe/modeling_qwen3_omni_moe.py:2250-2620) on the decode path of TTS/qwen3_tts_handler.py:910-1003. */
int32_t s2s_op_gemm_dec(void* stream, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int32_t M, int32_t N,
                        int32_t K, const float* ss_in, int32_t n_parts_in, int32_t ss_stride, float inv_dim, float eps, int32_t mode,
                        void* C, int64_t ldc, int32_t act, float* X, int64_t ldx, void* XB, int64_t ldxb, float* ss_out, int32_t glu,
                        int32_t nf, int32_t layout);
int32_t s2s_op_gemm_dec_parts(int32_t M, int32_t N, int32_t K, int32_t nf, int32_t layout);
/* The same projection behind a LAYERNORM instead of an RMSNorm - the pre-LN blocks of the Whisper decoder (transformers
 * whisper/modeling_whisper.py WhisperDecoderLayer: self_attn_layer_norm -> q/k/v, encoder_attn_layer_norm -> q, final_layer_norm -> fc1):
 *   out = rstd * (A W'^T - mean * csum) + bias,  W' = W * gamma (columns), csum[n] = sum_k bf16(W'[n][k]),  bias = b + W beta.
 * The row statistics come from PAIR partials (part 2 i = sum of squares, 2 i + 1 = plain sum of column slab i) written by the producing
 * mode-2 call when layout bit 16 is set (s2s_op_gemm_dec_parts counts both) or by s2s_op_rows_prep_pairs.  mode 0 / 1 only. */
int32_t s2s_op_gemm_dec_ln(void* stream, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int32_t M, int32_t N,
                           int32_t K, const float* ss_in, int32_t n_parts_in, int32_t ss_stride, float inv_dim, float eps, int32_t mode,
                           void* C, int64_t ldc, int32_t act, int32_t layout, const float* csum_dev);
int32_t s2s_op_rows_prep_pairs(void* stream, const float* x, int64_t x_rs, void* xb, int64_t xb_rs, float* ss, int32_t rows, int32_t cols,
                               int32_t frag, int64_t pair_stride);
/* xb = bf16(x) (frag != 0: fragment-major [rows][cols]) and ss[r] = sum_c x[r][c]^2 (one partial per row): entry of a norm-free decode pass */
int32_t s2s_op_rows_prep(void* stream, const float* x, int64_t x_rs, void* xb, int64_t xb_rs, float* ss, int32_t rows, int32_t cols,
                         int32_t frag);
/* row-major bf16 [rows][K] (row stride ld) -> fragment-major copy (roundup(rows,16) * K elements, zero padding rows);
 * gamma (optional f32 [K]) is folded into the columns (out = W * gamma rounded once to bf16). */
int32_t s2s_op_to_frag(void* stream, const void* W, int64_t ld, const float* gamma, void* out, int32_t rows, int32_t K);


/* y[r,:] = (x[r,:]-mean)/sqrt(var+eps)*gamma+beta ; x f32 [rows][cols], y y_dtype */
int32_t s2s_op_layernorm(void* stream, const float* x, const float* gamma, const float* beta,
                         void* y, int32_t rows, int32_t cols, float eps, int32_t y_dtype);
/* y[r,:] = x[r,:]*rsqrt(mean(x^2)+eps)*gamma */
int32_t s2s_op_rmsnorm(void* stream, const float* x, const float* gamma, void* y, int32_t rows,
                       int32_t cols, float eps, int32_t y_dtype);

/* 1-D convolution over a zero-padded time-major buffer as ONE GEMM (no im2col):
 *   x: [B][T_in + 2*pad][Cin] with `pad` zero rows on both sides of every item,
 *   w: [Cout][ksize][Cin] (tap-major), y: [B][y_rows][Cout], frame t written at row t + y_row0.
 * Used for the Whisper stem (modeling_whisper.py:618-619) and the codec-decoder convolutions. */
int32_t s2s_op_conv1d_tm(void* stream, const void* x, const void* w, const float* bias, void* y,
                         int32_t B, int32_t T_in, int32_t Cin, int32_t Cout, int32_t ksize,
                         int32_t stride, int32_t pad, int32_t y_rows, int32_t y_row0,
                         int32_t ab_dtype, int32_t c_dtype, int32_t act);

/* Causal dilated 1-D convolution over a history-padded time-major buffer (the codec decoder's residual-unit convolution,
 * sibling Qwen3OmniMoeCausalConvNet, modeling_qwen3_omni_moe.py:3180-3213): x [B][lpad + T][Cin] where the lpad >=
 * (ksize-1)*dilation rows in front of every item are its history (zeros at utterance start); w [Cout][ksize][Cin];
 * y [B][T][Cout], y[b][t] = bias + sum_j w[:, j, :] . x[b][lpad + t - (ksize-1-j)*dilation]. */
int32_t s2s_op_conv1d_causal_tm(void* stream, const void* x, const void* w, const float* bias, void* y,
                                int32_t B, int32_t T, int32_t Cin, int32_t Cout, int32_t ksize, int32_t dilation,
                                int32_t lpad, int32_t ab_dtype, int32_t c_dtype, int32_t act);

/* Scaled-dot-product attention over token-major buffers (f32 softmax).
 *   q: [B][Tq][q_rs] (head h at columns h*D), k / v: [B][Tk][kv_rs] (Hkv heads, grouped-query
 *   when H > Hkv), o: [B][Tq][H*D].  kv_len_dev: optional int32[B] valid keys.  bias: optional
 *   f32 [B][H][Tq][2*Tk-1] relative-position term, read at column (Tk-1) - t + j (the
 *   Transformer-XL rel-shift of modeling_parakeet.py:356-362).  causal: key j allowed iff
 *   j <= t + (Tk - Tq); window > 0 additionally requires j > t + (Tk - Tq) - window.
 *   use_mfma = 1 selects the bf16 flash kernel (needs H == Hkv, D in {64,128}, no causal mask
 *   and vt_scratch_dev of B*H*D*roundup(Tk,64) elements); 0 selects the row kernel (any dtype). */
int32_t s2s_op_attention(void* stream, const void* q, int64_t q_rs, const void* k, const void* v,
                         int64_t kv_rs, void* o, const int32_t* kv_len_dev, const float* bias,
                         int32_t B, int32_t H, int32_t Hkv, int32_t Tq, int32_t Tk, int32_t D,
                         float scale, int32_t causal, int32_t window, int32_t dtype,
                         int32_t use_mfma, void* vt_scratch_dev);

/* ------------------------------------------------------------------------- *
 * Whisper (encoder + greedy decoder) for one batch of sessions.
 *   Replaces model.get_encoder()(feats), model.detect_language(encoder_outputs) and
 *   model.generate(feats, max_new_tokens, num_beams=1, return_timestamps=False, task, language)
 *   of WhisperSTTHandler (STT/whisper_stt_handler.py:166-197,243).  Arithmetic restated from
 *   transformers/models/whisper/modeling_whisper.py and generation_whisper.py:1455-1608,
 *   1610-1675, 1774-1812.
 * ------------------------------------------------------------------------- */
typedef struct s2s_whisper_config {
    int32_t n_mels;     /* 80 (tiny.en) or 128 (large-v3) */
    int32_t d_model;
    int32_t n_heads;
    int32_t enc_layers;
    int32_t dec_layers;
    int32_t ffn;
    int32_t vocab;
    int32_t max_target; /* 448 */
    int32_t dtype;      /* S2S_BF16 (serving) or S2S_F32 (exact-f32 parity policy) */
} s2s_whisper_config;

int32_t s2s_whisper_create(const s2s_whisper_config* cfg, int32_t max_batch, void** out_model);
int32_t s2s_whisper_destroy(void* model);
/* Register one weight tensor (device pointer, kept by reference - the caller keeps it alive).
 * Names and layouts are listed in speech-to-speech_amd/whisper.py (WEIGHT LAYOUT). */
int32_t s2s_whisper_set_tensor(void* model, const char* name, const void* dev_ptr, int64_t numel);
/* Verifies that every tensor is present with the right size and allocates the workspace. */
int32_t s2s_whisper_finalize(void* model);
int64_t s2s_whisper_workspace_bytes(void* model);
/* feats_dev: [B][3002][n_mels] time-major log-mel in the model dtype, rows 0 and 3001 zero
 * (s2s_logmel_whisper layout 1).  Fills the encoder output and every layer's cross K|V. */
int32_t s2s_whisper_encode(void* model, void* stream, const void* feats_dev, int32_t B);
/* Copies the encoder output of the last s2s_whisper_encode into dst_dev: [B][1500][d_model], model dtype. */
int32_t s2s_whisper_read_encoder_output(void* model, void* stream, void* dst_dev, int32_t B);
/* Greedy decode after s2s_whisper_encode.
 *   prompt_dev: int32 [B][P] forced decoder prefix (<sot><lang><task><notimestamps>); an entry of -1
 *     is filled with the argmax over tokens whose prompt_fill_mask_dev byte is 0 (language
 *     detection: mask = 1 for every non-language token).
 *   suppress_dev / begin_suppress_dev: optional uint8 [vocab], 1 = suppressed (always / only for
 *     the first generated token).  eos: end-of-text id.  min_steps: decode at least this many
 *     steps before polling for "all sessions finished" (0 = poll from the start).
 *   out_ids_dev: int32 [B][max_new_tokens + P]: generated ids first (eos excluded), then P slots
 *     holding any filled-in prompt token (0 elsewhere); out_len_dev: int32 [B].
 *   last_logits_dev: optional f32 [B][vocab], logits of the final executed step (tests). */
int32_t s2s_whisper_decode_greedy(void* model, void* stream, int32_t B, const int32_t* prompt_dev,
                                  int32_t P, const uint8_t* suppress_dev, const uint8_t* begin_suppress_dev,
                                  const uint8_t* prompt_fill_mask_dev, int32_t eos, int32_t max_new_tokens,
                                  int32_t min_steps, int32_t* out_ids_dev, int32_t* out_len_dev,
                                  float* last_logits_dev);

/* Continuous batching of the Whisper token loop (ABI 3).  The reference decodes every request with its own model.generate() call
 * (STT/whisper_stt_handler.py:243); a pool that batches requests per generate() call makes a transcript wait for the whole batch in
 * front of it.  Here the rows of ONE running greedy loop are SLOTS: a session's encoder pass writes its cross K|V into a free slot's
 * rows (s2s_whisper_encode_at, on any stream: the encoder's scratch and the loop's state are disjoint), the session is admitted into
 * the loop at the next step boundary and leaves at ITS eos / token budget.  Results per session are those of
 * s2s_whisper_decode_greedy on that session (same kernels, same per-row arithmetic).  An engine runs either this loop or
 * s2s_whisper_decode_greedy at a time (they share the decoder's self-attention cache and step inputs).
 *   slots_config: the loop's masks (uint8 [vocab] each, optional) and eos - once, while no slot is live.
 * The library keeps a per-slot state (idle -> reserved by encode_at -> live after slots_admit -> idle after slots_release) and CHECKS the
 * scheduler against it: every call below fails with S2S_ERR_ARG instead of overwriting a session's rows or desynchronising the count.
 *   encode_at:    s2s_whisper_encode for B sessions whose cross K|V go to slots slot0 .. slot0 + B - 1; the slots must be idle and
 *                 are RESERVED from here on (s2s_whisper_encode / _decode_greedy refuse while any slot is reserved or live).
 *   slots_admit:  slots slot0 .. slot0 + n - 1 (reserved or idle, never live) start decoding: prompt_dev int32 [n][P] (P <= 8, -1 = fill
 *                 with the masked argmax), max_new_dev int32 [n] per-session budgets, clamped on the device to 1 .. max_new_limit
 *                 (P + max_new_limit is checked against max_target on the host).
 *   slots_step:   n_steps greedy steps over rows 0 .. rows - 1 (one captured graph per row count); idle and finished rows stand still.
 *   slots_collect: pack_dev int32 [rows][stride], stride = s2s_whisper_slots_pack_stride() = 2 + 8 + max_target:
 *                 {finished (0 | 1 eos | 2 budget), n_ids, filled prompt tokens [8] (0 = as given), ids [max_target]}; the filled tokens
 *                 and ids are written for finished rows only.
 *   slots_release: reserved / live slots become idle again (their cache rows are rewritten from position 0 by the next admission);
 *                 releasing an idle slot is an error, except for the whole range (slot0 = 0, n = max_batch: drop everything). */
int32_t s2s_whisper_slots_config(void* model, void* stream, const uint8_t* suppress_dev, const uint8_t* begin_suppress_dev,
                                 const uint8_t* prompt_fill_mask_dev, int32_t eos);
int32_t s2s_whisper_encode_at(void* model, void* stream, const void* feats_dev, int32_t B, int32_t slot0);
int32_t s2s_whisper_slots_admit(void* model, void* stream, int32_t slot0, int32_t n, const int32_t* prompt_dev, int32_t P,
                                const int32_t* max_new_dev, int32_t max_new_limit);
int32_t s2s_whisper_slots_step(void* model, void* stream, int32_t rows, int32_t n_steps);
int32_t s2s_whisper_slots_collect(void* model, void* stream, int32_t rows, int32_t* pack_dev);
int32_t s2s_whisper_slots_release(void* model, void* stream, int32_t slot0, int32_t n);
int32_t s2s_whisper_slots_pack_stride(void* model);

/* ------------------------------------------------------------------------- *
 * Parakeet-TDT (FastConformer encoder + TDT greedy decode) for one batch of sessions.
 *   Replaces nano_parakeet's model.transcribe(audio[, timestamps=True]) as called by
 *   ParakeetTDTSTTHandler._process_nano_parakeet (STT/parakeet_tdt_handler.py:621-623) and
 *   SmartProgressiveStreamingHandler._decode_window (STT/smart_progressive_streaming.py:82-85).
 *   nano-parakeet (>=0.2.0) is not available offline; the arithmetic is restated from
 *   transformers/models/parakeet/modeling_parakeet.py and generation_parakeet.py:113-122,271-294.
 * ------------------------------------------------------------------------- */
typedef struct s2s_parakeet_config {
    int32_t n_mels;          /* 128 */
    int32_t d_model;         /* 1024 */
    int32_t n_heads;         /* 8 */
    int32_t layers;          /* 24 */
    int32_t ffn;             /* 4096 */
    int32_t sub_channels;    /* 256 */
    int32_t conv_kernel;     /* 9 */
    int32_t vocab;           /* 8193 incl. blank (= vocab-1) */
    int32_t dec_hidden;      /* 640 */
    int32_t dec_layers;      /* 2 */
    int32_t n_durations;     /* 5: {0,1,2,3,4} */
    int32_t max_feat_frames; /* longest utterance in 10 ms feature frames (3001 = 30 s) */
    int32_t dtype;           /* S2S_BF16 or S2S_F32 */
} s2s_parakeet_config;

int32_t s2s_parakeet_create(const s2s_parakeet_config* cfg, int32_t max_batch, void** out_model);
int32_t s2s_parakeet_destroy(void* model);
/* encoder frames for max_feat_frames: the relative-position tables "enc.L.R" have 2*this-1 rows */
int32_t s2s_parakeet_max_encoder_frames(void* model);
/* names / layouts: speech-to-speech_amd/parakeet.py (WEIGHT LAYOUT) */
int32_t s2s_parakeet_set_tensor(void* model, const char* name, const void* dev_ptr, int64_t numel);
int32_t s2s_parakeet_finalize(void* model);
int64_t s2s_parakeet_workspace_bytes(void* model);
/* feats_dev: [B][T][n_mels] model dtype, zero beyond each row's n_frames_dev[b] valid frames
 * (the output of s2s_logmel_parakeet with t_max = T). */
int32_t s2s_parakeet_encode(void* model, void* stream, const void* feats_dev, const int32_t* n_frames_dev,
                            int32_t B, int32_t T);
/* encoder output of the last encode as f32 [B][T'][d_model] (+ optional valid lengths int32 [B]) */
int32_t s2s_parakeet_read_encoder_output(void* model, void* stream, float* dst_dev, int32_t* enc_len_dev);
/* TDT greedy decode of the last encoded batch.  max_steps <= 0 selects the reference bound
 * (10 * T' - 1).  Outputs are int32 [B][max_steps] (steps incl. blank emissions: token id, predicted
 * duration, encoder frame at which the step was taken) and n_steps_dev int32 [B]. */
int32_t s2s_parakeet_tdt_greedy(void* model, void* stream, int32_t max_steps, int32_t* tokens_dev,
                                int32_t* durations_dev, int32_t* frames_dev, int32_t* n_steps_dev);

/* ------------------------------------------------------------------------- *
 * Qwen3-TTS: talker + code predictor (codec frames) and codec decoder (waveform), one batch of sessions.
 *   Replaces FasterQwen3TTS.generate_custom_voice_streaming / generate_voice_clone_streaming /
 *   generate_voice_design_streaming as consumed by Qwen3TTSHandler._stream
 *   (TTS/qwen3_tts_handler.py:695-749, 910-1003): prompt -> 12.5 Hz frames of n_groups codes -> 24 kHz PCM,
 *   streamed every `chunk_size` frames.  faster-qwen3-tts / qwentts.cpp are not available offline; the
 *   arithmetic is restated from transformers' qwen3_omni_moe talker / code predictor / Code2Wav and qwen3.
 * ------------------------------------------------------------------------- */
typedef struct s2s_tts_config {
    int32_t t_hidden, t_layers, t_heads, t_kv_heads, t_head_dim, t_ffn;   /* talker (dense Qwen3 decoder) */
    int32_t codec_vocab, codec_eos;
    int32_t p_hidden, p_layers, p_heads, p_kv_heads, p_head_dim, p_ffn;   /* code predictor */
    int32_t n_groups, codebook;
    int32_t c_hidden, c_layers, c_heads, c_ffn, c_window, c_decoder_dim;  /* codec decoder */
    int32_t n_dec_blocks;    /* decoder blocks in use (<= 4) */
    int32_t up_rate[4];      /* decoder-block upsampling (8,5,4,3) */
    int32_t up_ratio[2];     /* pre-decoder upsamplers (2,2) */
    int32_t total_upsample;  /* 1920 samples per frame */
    int32_t max_prompt;      /* longest prefill (tokens) */
    int32_t max_frames;      /* codec-frame budget per utterance */
    int32_t max_pos;         /* talker KV-cache positions >= max_prompt + max_frames */
    int32_t max_codec_frames;/* longest codec_decode call (context + new frames) */
    float eps, c_eps;        /* RMSNorm eps of talker/predictor and of the codec transformer */
    int32_t dtype;
} s2s_tts_config;

int32_t s2s_tts_create(const s2s_tts_config* cfg, int32_t max_batch, void** out_model);
int32_t s2s_tts_destroy(void* model);
/* names / layouts: speech-to-speech_amd/tts.py (WEIGHT LAYOUT) */
int32_t s2s_tts_set_tensor(void* model, const char* name, const void* dev_ptr, int64_t numel);
int32_t s2s_tts_finalize(void* model);
int64_t s2s_tts_workspace_bytes(void* model);
/* Prefill the talker: prompt_embeds_dev f32 [B][Tp][t_hidden] (right-padded), prompt_len_dev int32 [B]. */
int32_t s2s_tts_prefill(void* model, void* stream, const float* prompt_embeds_dev, const int32_t* prompt_len_dev,
                        int32_t B, int32_t Tp);
/* Generate n_frames more frames (greedy).  codes_dev int32 [B][n_frames][n_groups] receives the frames of THIS call
 * (-1 after a session's eos), n_frames_dev int32 [B] the valid frames so far; eos is suppressed until a session has
 * min_total_frames frames; *all_done_host (optional, forces a stream sync) = 1 when every session has ended. */
int32_t s2s_tts_generate(void* model, void* stream, int32_t n_frames, int32_t min_total_frames, int32_t* codes_dev,
                         int32_t* n_frames_dev, int32_t* all_done_host);
/* Continuous batching of the frame loop.  The reference streams every session on its own (TTS/qwen3_tts_handler.py:695-749:
 * `_stream()` per request, independent generators); here the rows ("slots", 0 .. max_batch - 1) of ONE frame loop are shared, and
 * sessions join and leave at chunk boundaries while the others keep generating - every slot has its own frame index, eos gate,
 * frame budget and sampler key on the device.
 *   s2s_tts_admit       prefill n prompts (layout of s2s_tts_prefill) into the slots [slot0, slot0 + n); the other slots are not
 *                       touched.  min_frames_dev / budget_dev: optional int32[n] per session, else min_frames / max_new_frames for all
 *                       (a slot retires itself at eos or after its budget).  utterance_id0 + i keys session i's sampler stream.
 *   s2s_tts_step        n_frames passes of the frame body over the slots [0, rows) (captured graph per row count); finished / idle
 *                       slots stand still.
 *   s2s_tts_read_codes  frames [frame0, frame0 + n_frames) of the slots [slot0, slot0 + n) -> codes_dev int32 [n][n_frames][n_groups]
 *                       (-1 after a session's end); n_frames_dev (optional) int32[n] valid frames so far; valid_samples_dev (optional)
 *                       int32[n] = how many of the chunk's chunk_samples 24 kHz codec samples belong to valid frames (the resampler
 *                       treats the rest as silence, so a session's last samples do not depend on what its batch-mates still generate).
 *   s2s_tts_release     the slots [slot0, slot0 + n) become idle.
 * s2s_tts_prefill = release every slot + admit into slots 0 .. B-1; s2s_tts_generate = step + read of the frames of that call. */
int32_t s2s_tts_admit(void* model, void* stream, const float* prompt_embeds_dev, const int32_t* prompt_len_dev, int32_t n, int32_t Tp,
                      int32_t slot0, const int32_t* min_frames_dev, const int32_t* budget_dev, int32_t min_frames, int32_t max_new_frames,
                      int32_t utterance_id0);
/* Streaming-text mode (ABI v5; reference `non_streaming_mode = False`, arguments_classes/qwen3_tts_arguments.py:103-108: the text is NOT fully
 * pre-filled): for the sessions just admitted into the slots [slot0, slot0 + n) the prompt handed to s2s_tts_admit / s2s_tts_prefill ends with the
 * FIRST text token; the remaining text rows and the text-side eos row are given here and consumed one per generated frame - the input of the
 * step that produces frame f + 1 is the sum of frame f's code embeddings + trailing row f, and `talker.pad_embed` once they are used up
 * (the talker of the architectural sibling, modeling_qwen3_omni_moe.py:3744-3801 prompt split, :3169-3174 per-step rule).
 * rows_dev f32 [n][Tr][t_hidden] (right-padded), n_rows_dev int32[n], 1 <= Tr <= max_prompt.  Admission resets a slot to the pre-filled-text mode
 * (no trailing rows), so call this after the admission and before the next s2s_tts_step; s2s_tts_migrate carries the rows along. */
int32_t s2s_tts_set_trailing(void* model, void* stream, int32_t slot0, int32_t n, const float* rows_dev, const int32_t* n_rows_dev, int32_t Tr);
int32_t s2s_tts_step(void* model, void* stream, int32_t rows, int32_t n_frames);
int32_t s2s_tts_read_codes(void* model, void* stream, int32_t slot0, int32_t n, int32_t frame0, int32_t n_frames, int32_t* codes_dev,
                           int32_t* n_frames_dev, int32_t chunk_samples, int32_t* valid_samples_dev);
int32_t s2s_tts_release(void* model, void* stream, int32_t slot0, int32_t n);
/* Move n live sessions from the slots [src_slot0, ..) of src_model into the slots [dst_slot0, ..) of dst_model - two engines of the same
 * model (replicas sharing the weights, same max_prompt / max_frames): the talker cache rows of the first n_pos positions, the next input
 * row, frame index / eos gate / budget / sampler key, the codes emitted so far, the codec decoder's history rows and the first n_codec_pos
 * positions of its pre-transformer cache.  Asynchronous device copies on `stream`, which the caller has ordered behind the source engine's
 * last launch; the source slots may be re-admitted once the copies have completed.  Lets a session's latency-critical FIRST chunk run in a
 * small eager frame loop and its continuation in the wide, paced one (the reference streams every session independently,
 * TTS/qwen3_tts_handler.py:695-749; its output does not depend on which loop produced which chunk).
 * x_rows_dev (optional, f32 [n][t_hidden]): the sessions' next input rows as snapshotted by s2s_tts_read_x right behind their last frame -
 * needed when the source loop has stepped since (a parked = released slot keeps everything except that row).  The destination slots are live. */
int32_t s2s_tts_migrate(void* dst_model, void* src_model, void* stream, int32_t src_slot0, int32_t dst_slot0, int32_t n, int32_t n_pos,
                        int32_t n_codec_pos, const float* x_rows_dev);
int32_t s2s_tts_read_x(void* model, void* stream, int32_t slot0, int32_t n, float* x_out_dev);
/* s2s_tts_read_codes for rows of DIFFERENT ages in one call: frame0_rows_dev int32[n] (device) = first frame to read per slot; chunk_samples
 * = n_frames * total_upsample (continuation chunks only - a stream's first chunk goes through s2s_tts_read_codes). */
int32_t s2s_tts_read_codes_rows(void* model, void* stream, int32_t slot0, int32_t n, const int32_t* frame0_rows_dev, int32_t n_frames,
                                int32_t* codes_dev, int32_t* n_frames_dev, int32_t chunk_samples, int32_t* valid_samples_dev);
/* Sampling of the frame loop (the reference's backend samples; greedy is the parity policy and the default).  which: 0 = talker
 * (codebook 0; the repetition penalty covers the utterance's earlier codebook-0 picks), 1 = code predictor (residual groups).
 * Processor order = transformers generate(): repetition penalty -> temperature -> top-k -> top-p -> seeded multinomial
 * (counter-based uniform of (seed, utterance id, frame, group): reproducible, independent of batch composition and of the slot).  temperature <= 0
 * restores greedy.  Re-captures the frame graphs on the next generate call. */
int32_t s2s_tts_set_sampling(void* model, int32_t which, float temperature, int32_t top_k, float top_p, float repetition_penalty,
                             uint64_t seed);
/* The frame loop's sampler on plain rows (tests): out_dev int32 [B] = one token per row of logits_dev f32 [B][V], V <= 4096; row b
 * draws the counter-based uniform of (seed, b); hist_dev optional int32 [B][n_hist] (repetition penalty). */
int32_t s2s_op_sample_tokens(void* stream, const float* logits_dev, int32_t B, int32_t V, float temperature, int32_t top_k, float top_p,
                             float repetition_penalty, uint64_t seed, const int32_t* hist_dev, int32_t n_hist, int32_t* out_dev);
/* Teacher-forced continuation (continuing given codec frames in context; the parity tests' teacher-forced logits): like
 * s2s_tts_generate, but every code of the n_frames new frames whose entry in forced_codes_dev int32 [B][n_frames][n_groups]
 * is >= 0 is emitted (and fed back to the code predictor / talker) instead of the greedy pick, and the raw logits of every
 * pick are copied to logits_dev f32 [B][n_frames][n_groups][vlog] (optional; group 0 rows hold codec_vocab values, groups
 * 1.. hold codebook values; vlog >= both).  Plain stream launches (no graph replay). */
int32_t s2s_tts_generate_forced(void* model, void* stream, int32_t n_frames, int32_t min_total_frames, const int32_t* forced_codes_dev,
                                float* logits_dev, int32_t vlog, int32_t* codes_dev, int32_t* n_frames_dev);
/* Codec decoder: codes_dev int32 [B][F][n_groups] (the first ctx frames are left context) -> wav_dev f32
 * [B][n_samples], n_samples = s2s_codec_samples(F) - ctx * total_upsample, values clamped to [-1, 1]. */
int32_t s2s_codec_decode(void* model, void* stream, const int32_t* codes_dev, int32_t B, int32_t F, int32_t ctx,
                         float* wav_dev, int32_t n_samples);
int64_t s2s_codec_samples(void* model, int32_t F);
/* Exact causal streaming: after s2s_codec_stream_reset, successive calls continue ONE decode of the whole utterance -
 * every convolution's history rows and the transformer KV cache are carried over, nothing is recomputed, and the
 * concatenated outputs equal s2s_codec_decode of all frames at once.  n_samples = s2s_codec_samples(F) for the first
 * call, F * total_upsample afterwards.  All sessions of the batch advance together. */
int32_t s2s_codec_stream_reset(void* model);
int32_t s2s_codec_stream_decode(void* model, void* stream, const int32_t* codes_dev, int32_t B, int32_t F, float* wav_dev,
                                int32_t n_samples);
/* The same continuation for a COHORT of a continuously batched engine: rows [slot0, slot0 + n) of the engine's codec state (history
 * rows, pre-transformer cache), frames_done frames already decoded for them (0: their stream starts with this call).  codes_dev
 * int32 [n][F][n_groups]; n_samples = s2s_codec_samples(F) for the first call of a stream, F * total_upsample afterwards.  Cohorts on
 * disjoint row ranges stream independently of each other (sessions admitted at different chunk boundaries). */
int32_t s2s_codec_decode_rows(void* model, void* stream, const int32_t* codes_dev, int32_t slot0, int32_t n, int32_t F, int32_t frames_done,
                              float* wav_dev, int32_t n_samples);
/* Continuation of rows of DIFFERENT ages in ONE pass (all continuation cohorts of a round; the launch count of a round no longer grows
 * with the number of cohorts): frames_done_rows_dev int32[n] (device) = frames already decoded per row (> 0 for every live stream: the
 * first chunk of a stream has its own geometry and goes through s2s_codec_decode_rows with frames_done = 0), max_frames_done = the
 * largest entry.  Rows of [slot0, slot0 + n) that carry no live stream are computed and ignored; their state is rewritten by the
 * first chunk of the next stream in that slot.  wav_dev f32 [n][n_samples], n_samples = F * total_upsample. */
int32_t s2s_codec_decode_rows_at(void* model, void* stream, const int32_t* codes_dev, int32_t slot0, int32_t n, int32_t F,
                                 const int32_t* frames_done_rows_dev, int32_t max_frames_done, float* wav_dev, int32_t n_samples);
/* Post-processing of Qwen3TTSHandler._stream (TTS/qwen3_tts_handler.py:612-613,674-680): per-chunk
 * scipy.signal.resample_poly(x, up=2, down=3) then clip(x*32768).astype(int16).
 *   pcm24_dev f32 [B][in_stride], n_in_dev optional int32 [B] (else in_stride samples), fir_dev f32 [taps] the
 *   host-designed polyphase prototype (already multiplied by `up`), delay = (taps-1)/2.
 *   out16_dev int16 / outf_dev f32 (either may be NULL) [B][out_stride]; n_out_max outputs computed per row
 *   (rows shorter than that are zero-filled past ceil(2n/3)). */
int32_t s2s_resample_24k16k_int16(void* stream, const float* pcm24_dev, int64_t in_stride, const int32_t* n_in_dev,
                                  const float* fir_dev, int32_t taps, int32_t delay, int16_t* out16_dev, float* outf_dev,
                                  int64_t out_stride, int32_t n_out_max, int32_t B);

/* API-edge resampling: batched scipy.signal.resample_poly(x, up, down) per row followed by
 * clip(y * 32768).astype(int16) - the reference's resample() of api/openai_realtime/utils.py:5-12, which handlers/audio.py runs per
 * chunk per session on the asyncio thread (:122-149 client rate -> 16 kHz in, :293-330 16 kHz -> client rate out).
 *   in_dev: int16 [B][in_stride] (in_is_int16 != 0; samples are int16 / 32768 exactly as the reference converts them) or float32;
 *   n_in_dev: optional int32 [B] valid samples per row; fir_dev: the host-designed prototype (scipy's firwin(2*10*max(up,down)+1,
 *   1/max(up,down), kaiser 5.0) * up), odd length; n_out per row = ceil(n_in * up / down), rows are zero-filled up to n_out_max. */
int32_t s2s_resample_poly_int16(void* stream, const void* in_dev, int32_t in_is_int16, int64_t in_stride, const int32_t* n_in_dev, int32_t up,
                                int32_t down, const float* fir_dev, int32_t taps, int16_t* out16_dev, float* outf_dev, int64_t out_stride,
                                int32_t n_out_max, int32_t B);

#ifdef __cplusplus
}
#endif
#endif /* S2S_MI355X_H */
