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
};
int gemm_ex(hipStream_t st, const GemmArgs& g);
int gemm(hipStream_t st, const void* A, long lda, const void* W, const float* bias,
         const float* residual, void* C, long ldc, int M, int N, int K, int ab_dtype, int c_dtype,
         int act, bool allow_splitk);
int layernorm(hipStream_t st, const float* x, const float* gamma, const float* beta, void* y, int rows,
              int cols, float eps, int y_dtype);
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
    int window;          // >0: additionally j > t + causal_off - window
    float scale;
};
struct AttnMfmaParams {
    const __hip_bfloat16* Q; long q_bs, q_rs;
    const __hip_bfloat16* K; long k_bs, k_rs;
    const __hip_bfloat16* Vt;          // [B][H][D][Tp]
    __hip_bfloat16* O; long o_bs, o_rs;
    const float* bias; long bias_bs, bias_hs, bias_rs; int bias_shift;
    const int* kv_len; int Tk, Tp, Tq, H, B;
    float scale_log2e;                 // softmax scale * log2(e)
};
int attn_rows(hipStream_t st, const AttnRowsParams& p, int dtype);
int attn_mfma(hipStream_t st, const AttnMfmaParams& p, int D);
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

}  // namespace s2s
