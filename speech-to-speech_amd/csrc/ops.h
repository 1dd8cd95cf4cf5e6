// Internal (non-ABI) entry points shared between the translation units of libs2s_mi355x.
#pragma once
#include "common.h"

namespace s2s {

struct GemmArgs {
    const void* A; long lda; long a_bs;          // row r of A at A + (r/G)*a_bs + (r%G)*lda
    const void* W; const float* bias;            // W [N][K]
    const float* residual; long ldr; long r_bs;  // optional f32 residual, same row mapping (ldr 0 -> ldc)
    void* C; long ldc; long c_bs;
    int M, N, K, G;                              // G rows per group (<=0: one group of M rows)
    int ab_dtype, c_dtype, act;
    bool allow_splitk;
    long ldw;                                    // row stride of W (0 -> K)
    int nz1, nz2;                                // batched: grid.y = nz1*nz2 independent GEMMs (0 -> 1)
    long a_z1, a_z2, w_z1, w_z2, c_z1, c_z2;     // per-batch element offsets (C strides also apply to residual)
    float alpha; bool alpha_set;                 // C = act(acc+bias)*alpha + residual (alpha_set false -> 1)
    int tap_len; long tap_stride;                // dilated conv taps: A index k -> (k/tap_len)*tap_stride + k%tap_len
    bool glu;                                    // M <= 64: W = [gate rows | up rows] (N each), C = silu(gate) * up (width N)
    // Decode-path RMSNorm fusion (bf16 weights, M <= 64): A is the raw f32 residual stream [M][lda]; the kernel rounds it
    // to bf16 on the fly, accumulates every row's sum of squares from the same loads and scales the output tile by
    // rsqrt(mean(x^2) + rms_eps).  The norm weight must already be folded into W's columns (W[n][k] * gamma[k]).
    bool a_f32_rms; float rms_eps;
    // Epilogue extensions of the tile kernels (codec decoder; C in the model dtype, alpha 1):
    //   act == S2S_ACT_SNAKE: v += sin^2(v * snake_a[c]) * snake_ib[c], c = n % snake_mod (snake_mod 0: c = n)
    //   res_c: residual in C's dtype, ld and batch stride (may BE C: in-place x += projection)
    //   C2: second output in C's dtype / ld / batch stride = snake2(the value as stored in C) - the next layer's activation
    const float* snake_a; const float* snake_ib; int snake_mod;
    const void* res_c;
    void* C2; const float* snake2_a; const float* snake2_ib;
};
int gemm_ex(hipStream_t st, const GemmArgs& g);
void conv7_set(int on);                  // tap-reuse convolution kernel: 1 / 0 = forced on / off, -1 = default (environment S2S_CONV7)
int gemm(hipStream_t st, const void* A, long lda, const void* W, const float* bias,
         const float* residual, void* C, long ldc, int M, int N, int K, int ab_dtype, int c_dtype,
         int act, bool allow_splitk);
// Decode-step projection (gemm_dec.hip): bf16, M <= 128, K % 32 == 0.  See the file header for the norm-free layer scheme.
struct DecGemm {
    const __hip_bfloat16* A; long lda;      // [M][lda]; raw (un-normalised) bf16 rows when ss_in is set
    const __hip_bfloat16* W; long ldw;      // FRAGMENT-MAJOR weights (to_frag): [roundup(N,16)][K]; glu: [gate tiles | up tiles]; ldw unused
    bool a_frag;                            // A is fragment-major (else row-major [M][lda])
    bool w_cached;                          // weights are re-read every step of a loop (code predictor): plain loads that stay in L2 / the
                                            // Infinity Cache; default (false): non-temporal streaming loads
    bool c_frag, xb_frag;                   // mode 0 C / mode 2 XB written fragment-major (row width ldc / ldxb, multiple of 32)
    const float* bias;                      // [N] or nullptr
    int M, N, K;
    // RMS prologue: out = rsqrt(sum_p ss_in[p][m] * inv_dim + eps) * acc  (gamma folded into W's columns)
    const float* ss_in; int n_parts_in; int ss_stride; float inv_dim; float eps;
    int ss_rs;                              // GEMM row m reads ss_in[p * ss_stride + m * ss_rs] (0 -> 1)
    // LayerNorm prologue (Whisper's pre-LN decoder blocks): ss_in holds PAIRS of partials - even parts sums of squares, odd parts plain
    // sums - and out = rstd * (acc - mean * csum[n]) + bias with gamma folded into W's columns, csum[n] = sum_k W'[n][k] (the folded,
    // bf16-rounded weights) and beta folded into the bias (b' = b + W beta).  16-column kernel only.
    bool ln; const float* csum;
    bool ss_pairs;                          // mode 2: write pairs (part 2 i = sum of squares, 2 i + 1 = sum) for a LayerNorm consumer
    int mode;                               // 0: C bf16, 1: C f32, 2: X += out (f32, in place) + XB (bf16 copy) + ss_out
    void* C; long ldc; int act;
    float* X; long ldx; __hip_bfloat16* XB; long ldxb;
    float* ss_out;                          // [dec_parts(N, nf)][ss_stride] partial sums of squares of the new X rows
    bool x_overwrite;                       // mode 2: X = out (no read of the old rows): projections that START a residual stream
    bool glu;                               // C = silu(gate) * up (mode 0)
    int nf;                                 // 16-column fragments per workgroup of the 16-column kernel (0: chosen from the shape; > 0 also
                                            // keeps the call on that kernel)
    // K-slice workspace of the column-block kernel (gemm_dec2.hip): 16 KiB of arrival tickets (zero between launches) followed by the
    // f32 partial tiles.  One per stream of launches (an engine owns its own); null: the process-wide default of the public op.
    void* ws; size_t ws_bytes;
};
int gemm_dec(hipStream_t st, const DecGemm& a);
int dec_nf(const DecGemm& a);
int dec_parts(int N, int NF);               // partial sums the 16-column kernel writes per row for an N-column residual projection
int dec_out_parts(const DecGemm& a);        // partial sums per row gemm_dec(a) (mode 2) writes, whichever kernel it picks
// column-block / K-slice kernel for 33 .. 128-row slabs (gemm_dec2.hip): 1 launched, 0 not eligible, < 0 error
int gemm_dec2(hipStream_t st, const DecGemm& a);
int dec2_parts(const DecGemm& a);           // column blocks (= partial sums per row) when gemm_dec2 takes the shape, else 0
int dec2_ensure_default_ws();
void* dec2_default_ws(size_t* bytes);       // the process-wide default K-slice workspace (null before dec2_ensure_default_ws)
// row-block x column-block kernel for 129 .. 1024 rows (gemm_decw.hip): 1 launched, 0 not eligible, < 0 error
int gemm_decw(hipStream_t st, const DecGemm& a);
int decw_parts(const DecGemm& a);           // column blocks (= partial sums per row) when gemm_decw takes the shape, else 0
constexpr size_t DEC_WS_BYTES = (size_t)48 << 20;   // K-slice workspace an engine allocates per stream of decode launches
int rows_prep(hipStream_t st, const float* x, long x_rs, void* xb, long xb_rs, float* ss, int rows, int cols, bool frag, long pair_stride = 0);   // pair_stride > 0: also ss[pair_stride + row] = plain sum
int to_frag(hipStream_t st, const void* W, long ld, const float* gamma, void* out, int rows, int K);

int layernorm(hipStream_t st, const float* x, const float* gamma, const float* beta, void* y, int rows,
              int cols, float eps, int y_dtype, bool y_frag = false);   // y_frag: bf16 rows written fragment-major (gemm_dec.hip)
// rows at x + r*x_rs; optional second copy of the normalised rows in f32 (y_f32, contiguous)
int rmsnorm_ex(hipStream_t st, const float* x, long x_rs, const float* gamma, void* y, float* y_f32, int rows, int cols, float eps,
               int y_dtype);
int rmsnorm(hipStream_t st, const float* x, const float* gamma, void* y, int rows, int cols, float eps,
            int y_dtype);

struct AttnRowsParams {
    const void* Q; long q_bs, q_rs;   // element (b,t,h,d) at Q + b*q_bs + t*q_rs + h*D + d
    const void* K; const void* V; long kv_bs, kv_rs;   // (b,j,hk,d) at K + b*kv_bs + j*kv_rs + hk*D + d
    void* O; long o_bs, o_rs;
    const float* bias; long bias_bs, bias_hs, bias_rs;   // optional additive bias[b][h][t][(bias_shift - t) + j]
    int bias_shift;
    const int* kv_len;   // optional int32[B]; else kv_len_const
    int kv_len_const;
    int B, H, Hkv, Tq, D;
    int causal;          // 1: key j allowed iff j <= t + causal_off
    int causal_off;
    const int* causal_off_arr;   // optional int32[B]: per-session offset (tokens already in the cache) instead of causal_off
    int window;          // >0: additionally j > t + causal_off - window
    float scale;
    bool o_frag;         // Tq == 1 decode kernel only: write O as fragment-major rows [B][H*D] (gemm_dec.hip operand layout)
};
struct AttnMfmaParams {
    const __hip_bfloat16* Q; long q_bs, q_rs;
    const __hip_bfloat16* K; long k_bs, k_rs;
    const __hip_bfloat16* Vt;          // [B][H][D][Tp] (pre-transposed by transpose_v: rounds 2 - 5, S2S_ATTN_VTR=0), or null:
    const __hip_bfloat16* V; long v_bs, v_rs;   // V rows as they come out of the qkv projection, (b, key, h, d) at V + b*v_bs + key*v_rs + h*D + d -
                                       // the kernel transposes in its LDS read (ds_read_b64_tr_b16), no separate pass (round 6)
    __hip_bfloat16* O; long o_bs, o_rs;
    const float* bias; long bias_bs, bias_hs, bias_rs; int bias_shift;
    const int* kv_len; int Tk, Tp, Tq, H, B;
    float scale_log2e;                 // softmax scale * log2(e)
};
// One decode step (a single new token per session) of a self-attention layer, from the fused projection row to the output:
// q/k head RMSNorm + RoPE (both optional), K/V append at pos[b], attention over cache[max(0, pos-window+1) .. pos].
struct AttnStepParams {
    const void* qkv; long qkv_rs;      // row b: [q heads | k heads | v heads] x D, model dtype
    void* K; void* V; long kv_bs, kv_rs;   // cache (b, j, hk, d) at K + b*kv_bs + j*kv_rs + hk*D + d; row pos[b] is written
    void* O; long o_bs;                // (b, h, d) at O + b*o_bs + h*D + d
    const float* qw; const float* kw; float eps;       // per-head RMSNorm weights [D] (nullptr: none)
    const float* cosT; const float* sinT;              // rotary tables [max_pos][D/2] (nullptr: none)
    const int* pos;                    // int32[B]: tokens already in the cache
    int B, H, Hkv, D, kv_cap, window;  // kv_cap: cache capacity (LDS sizing); window > 0: sliding window
    float scale;
    bool o_frag;                       // write O as fragment-major rows [B][H*D] (operand layout of gemm_dec.hip)
};
int attn_step(hipStream_t st, const AttnStepParams& p, int dtype);
int attn_rows(hipStream_t st, const AttnRowsParams& p, int dtype);
int attn_mfma(hipStream_t st, const AttnMfmaParams& p, int D);
bool attn_mfma_direct_v();             // true: callers hand V (not V^T) to attn_mfma and skip transpose_v
int transpose_v(hipStream_t st, const void* V, long v_bs, long v_rs, void* Vt, int B, int H, int D, int Tk,
                int Tp, int dtype);
// misc.hip
int embed(hipStream_t st, const int* ids, const int* pos, const void* tok_emb, const void* pos_emb, float* x, int B,
          int d, float emb_scale, int dtype);
int kv_append(hipStream_t st, const void* src, long src_rs, int k_off, int v_off, void* kc, void* vc, long c_bs,
              const int* pos, int B, int n, int dtype);
int argmax_suppress(hipStream_t st, const float* logits, long ld, int B, int V, const unsigned char* mask,
                    const unsigned char* begin_mask, const int* step, int begin_step, int* out, float* out_val);
int cast_f32(hipStream_t st, const float* in, void* out, long n, int dtype);

// teacher forcing / logit capture of the TTS frame loop (tts.hip pick kernels); all null / zero on the serving path
struct PickDebug {
    const int* forced;   // [B][max_frames][G] codes to emit instead of the argmax (entries < 0: keep the argmax)
    float* log;          // [B][n_log][G][Vlog] raw logits of frames f0 .. f0 + n_log - 1
    int f0, n_log, Vlog;
};

// Sampling of the TTS frame loop (tts.hip pick kernels).  temperature <= 0 selects the greedy argmax (the parity policy).
// Order of the processors = transformers' generate(): repetition penalty -> temperature -> top-k -> top-p -> multinomial.
struct SampleParams {
    float temperature;      // > 0: sample
    int top_k;              // 0: off
    float top_p;            // >= 1: off
    float rep_penalty;      // 1: off; applied to the codes already emitted for this group in the current utterance
    unsigned long long seed;
};

// conformer.hip
struct TdtState {
    int* frame;      // [B] encoder frame pointer
    int* finished;   // [B]
    int* tok;        // [B] token chosen at the last step (input of the prediction network)
    int* adv;        // [B] 1 = prediction network must advance for this row (non-blank emission)
    int* n_steps;    // [B] steps recorded
    int* n_active;   // [1]
};
int sub_conv1(hipStream_t st, const void* x, const float* w, const float* b, void* out, const int* len_in, int B, int Tin,
              int F, int C, int dtype);
int dwconv2d(hipStream_t st, const void* in, const float* w, const float* b, void* out, const int* len0, int halvings, int B,
             int Tin, int Fin, int C, int dtype);
int mask_rows(hipStream_t st, void* x, const int* len0, int halvings, int B, int Trows, long row_elems, int dtype);
int sub_lengths(hipStream_t st, const int* len0, int halvings, int* out, int B);
int glu_dwconv(hipStream_t st, const void* y, const float* w, const float* b, void* out, const int* len, int B, int Tn, int d,
               int ksize, int dtype);
int add_rowvec(hipStream_t st, const void* in, long ld_in, const float* vec, void* out, long rows, int cols, int dtype);
int tdt_joint_in(hipStream_t st, const float* encp, const float* dec, const int* frame, void* A, int B, int Tn, int Hd, int dtype, const int* dyn = nullptr);
int tdt_select(hipStream_t st, const float* logits, int ld, int V, int ND, int blank, const int* enc_len, const TdtState& s,
               int* out_tok, int* out_dur, int* out_frame, int max_steps, int B, const int* dyn = nullptr);
int lstm_in(hipStream_t st, const void* emb, const int* tok, const float* xin, const float* h, void* X, int B, int Hd, int dtype);
int lstm_cell(hipStream_t st, const float* gates, const float* c, float* h_new, float* c_new, int B, int Hd);
int tdt_commit(hipStream_t st, float* const* dst, const float* const* src, int n_arrays, const int* adv, int force, int B, int Hd);
// fused step kernels (conformer.hip, round 6): select + layer-0 LSTM input; LSTM cell + the next GEMM's input; commit + the next step's joint input
int tdt_select_in(hipStream_t st, const float* logits, int ld, int V, int ND, int blank, const int* enc_len, const TdtState& s, int* out_tok, int* out_dur,
                  int* out_frame, int max_steps, int B, const int* dyn, const void* emb, const float* h0, void* X, int Hd, int dtype);
int lstm_cell_out(hipStream_t st, const float* gates, const float* c, float* h_new, float* c_new, int B, int Hd, void* Xn, const float* h_next, int dtype);
int tdt_commit_joint(hipStream_t st, float* const* dst, const float* const* src, int n_arrays, const int* adv, int force, int B, int Hd, const float* encp,
                     const int* frame, void* A, int Tn, const int* dyn, int* n_active, int dtype);

}  // namespace s2s
