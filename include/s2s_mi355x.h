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

#define S2S_ABI_VERSION 1

/* element types of device buffers */
#define S2S_F32 0
#define S2S_BF16 1

/* GEMM epilogue activation codes */
#define S2S_ACT_NONE 0
#define S2S_ACT_GELU 1 /* exact erf GELU (Whisper, modeling_whisper.py:618-619) */
#define S2S_ACT_SILU 2 /* SiLU (FastConformer FFN; Qwen3 SwiGLU gate) */
#define S2S_ACT_RELU 3

const char* s2s_last_error(void);
int32_t s2s_abi_version(void);
/* number of visible HIP devices (0 when there is no GPU); never fails */
int32_t s2s_device_count(void);

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

/* out_dev: [B][t_max][n_mels] (time-major, zero beyond each row's valid
 * frames), n_frames_dev: int32[B] valid frames (= n_samples / 160).
 * t_max must be >= max(n_samples)/160 + 1. */
int32_t s2s_logmel_parakeet(void* frontend, void* stream, const float* pcm_dev,
                            int64_t pcm_stride, const int32_t* n_samples_dev, int32_t B,
                            int32_t t_max, void* out_dev, int32_t out_dtype,
                            int32_t* n_frames_dev);

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

/* y[r,:] = (x[r,:]-mean)/sqrt(var+eps)*gamma+beta ; x f32 [rows][cols], y y_dtype */
int32_t s2s_op_layernorm(void* stream, const float* x, const float* gamma, const float* beta,
                         void* y, int32_t rows, int32_t cols, float eps, int32_t y_dtype);
/* y[r,:] = x[r,:]*rsqrt(mean(x^2)+eps)*gamma */
int32_t s2s_op_rmsnorm(void* stream, const float* x, const float* gamma, void* y, int32_t rows,
                       int32_t cols, float eps, int32_t y_dtype);

#ifdef __cplusplus
}
#endif
#endif /* S2S_MI355X_H */
