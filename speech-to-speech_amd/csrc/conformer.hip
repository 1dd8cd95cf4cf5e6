// FastConformer-specific kernels (Parakeet) and the TDT greedy-decode step kernels.
//
// Layouts are channels-last so every access is coalesced over channels and the pointwise
// convolutions / the flatten+Linear are plain GEMMs over the same buffers:
//   subsampling activations  [B][T][F][C]      conv module  [B][T][d]
#include "ops.h"

#include <algorithm>

namespace s2s {

__device__ __forceinline__ int sub_len(int n) { return n > 0 ? (n - 1) / 2 + 1 : 0; }   // (n + 2 - 3)/2 + 1

// ---- subsampling conv1: Conv2d(1 -> C, k3, s2, p1) + time mask + ReLU ----------------------
// x [B][T][F] (T rows beyond the valid length are already zero), out [B][T1][F1][C].
// grid (T1, B), block 256.  len_in: valid input frames per item.
template <typename T>
__global__ __launch_bounds__(256) void sub_conv1_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ out,
                                                        const int* __restrict__ len_in, int Tin, int F, int T1, int F1, int C) {
    extern __shared__ float rows[];   // [3][F + 2]
    const int t1 = blockIdx.x, b = blockIdx.y;
    const int FW = F + 2;
    for (int i = threadIdx.x; i < 3 * FW; i += 256) {
        const int r = i / FW, f = i - r * FW - 1;
        const int t = 2 * t1 + r - 1;
        float v = 0.f;
        if (t >= 0 && t < Tin && f >= 0 && f < F) v = Elem<T>::ld(x + ((long)b * Tin + t) * F + f);
        rows[i] = v;
    }
    __syncthreads();
    const bool valid = t1 < sub_len(len_in[b]);
    for (int c = threadIdx.x; c < C; c += 256) {
        float wk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = w[c * 9 + k];
        const float bv = bias[c];
        T* o = out + (((long)b * T1 + t1) * F1) * C + c;
        for (int f1 = 0; f1 < F1; ++f1) {
            float acc = bv;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc = fmaf(wk[i * 3 + j], rows[i * FW + 2 * f1 + j], acc);
            Elem<T>::st(o + (long)f1 * C, valid ? fmaxf(acc, 0.f) : 0.f);
        }
    }
}

// ---- depthwise Conv2d(C, k3, s2, p1, groups=C) + time mask (no activation) ----------------
// in [B][Tin][Fin][C] -> out [B][Tout][Fout][C].  Input rows >= len_in are treated as zero (the mask of
// the producing layer), output rows >= len_out are written as zero.  grid (Tout, B), block 256.
template <typename T>
__global__ __launch_bounds__(256) void dwconv2d_kernel(const T* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, T* __restrict__ out,
                                                       const int* __restrict__ len0, int halvings, int Tin, int Fin,
                                                       int Tout, int Fout, int C) {
    const int to = blockIdx.x, b = blockIdx.y;
    int lin = len0[b];
    for (int i = 0; i < halvings; ++i) lin = sub_len(lin);
    const int lout = sub_len(lin);
    for (int c = threadIdx.x; c < C; c += 256) {
        float wk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) wk[k] = w[c * 9 + k];
        const float bv = bias[c];
        for (int fo = 0; fo < Fout; ++fo) {
            float acc = bv;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int t = 2 * to + i - 1;
                if (t < 0 || t >= Tin || t >= lin) continue;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int f = 2 * fo + j - 1;
                    if (f < 0 || f >= Fin) continue;
                    acc = fmaf(wk[i * 3 + j], Elem<T>::ld(in + (((long)b * Tin + t) * Fin + f) * C + c), acc);
                }
            }
            Elem<T>::st(out + (((long)b * Tout + to) * Fout + fo) * C + c, to < lout ? acc : 0.f);
        }
    }
}

// zero rows t >= len of a [B][T][row_elems] buffer (applied after the pointwise convs)
template <typename T>
__global__ void mask_rows_kernel(T* __restrict__ x, const int* __restrict__ len0, int halvings, int Trows, long row_elems) {
    const int t = blockIdx.x, b = blockIdx.y;
    int l = len0[b];
    for (int i = 0; i < halvings; ++i) l = sub_len(l);
    if (t < l) return;
    T* r = x + ((long)b * Trows + t) * row_elems;
    for (long i = threadIdx.x; i < row_elems; i += blockDim.x) r[i] = Elem<T>::zero();
}

__global__ void sub_lengths_kernel(const int* __restrict__ len0, int halvings, int* __restrict__ out, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int l = len0[b];
    for (int i = 0; i < halvings; ++i) l = sub_len(l);
    out[b] = l;
}

// ---- conv module core: GLU -> key-padding mask -> depthwise k (odd) -> folded BatchNorm -> SiLU --------
// y [B][T][2d] -> out [B][T][d].  grid (ceil(T/32), d/64, B), block 256.
template <typename T, int KS>
__global__ __launch_bounds__(256) void glu_dwconv_kernel(const T* __restrict__ y, const float* __restrict__ w,
                                                         const float* __restrict__ bias, T* __restrict__ out,
                                                         const int* __restrict__ len, int Tn, int d) {
    constexpr int TT = 32, HALO = KS / 2;
    __shared__ float g[TT + 2 * HALO][65];
    const int t0 = blockIdx.x * TT, c0 = blockIdx.y * 64, b = blockIdx.z;
    const int L = len ? min(len[b], Tn) : Tn;
    const int c = threadIdx.x & 63;
    for (int r = threadIdx.x >> 6; r < TT + 2 * HALO; r += 4) {
        const int t = t0 + r - HALO;
        float v = 0.f;
        if (t >= 0 && t < L) {
            const T* row = y + ((long)b * Tn + t) * 2 * d;
            const float a = Elem<T>::ld(row + c0 + c), gate = Elem<T>::ld(row + d + c0 + c);
            v = a / (1.0f + expf(-gate));
        }
        g[r][c] = v;
    }
    __syncthreads();
    float wk[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) wk[k] = w[(c0 + c) * KS + k];
    const float bv = bias[c0 + c];
    for (int r = threadIdx.x >> 6; r < TT; r += 4) {
        const int t = t0 + r;
        if (t >= Tn) break;
        float acc = bv;
#pragma unroll
        for (int k = 0; k < KS; ++k) acc = fmaf(wk[k], g[r + k][c], acc);
        Elem<T>::st(out + ((long)b * Tn + t) * d + c0 + c, acc / (1.0f + expf(-acc)));
    }
}

// out[r][c] = in[r*ld_in + c] + vec[c]
template <typename T>
__global__ void add_rowvec_kernel(const T* __restrict__ in, long ld_in, const float* __restrict__ vec, T* __restrict__ out,
                                  long rows, int cols) {
    const long total = rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        Elem<T>::st(out + i, Elem<T>::ld(in + r * ld_in + c) + vec[c]);
    }
}

// ---------------------------------------------------------------------------------------------
// TDT greedy decode step kernels (state lives on the device; one launch sequence per step)
// ---------------------------------------------------------------------------------------------


// A[b][:] = relu(encp[b][min(frame, T-1)][:] + dec[b][:])
template <typename T>
__global__ void tdt_joint_in_kernel(const float* __restrict__ encp, const float* __restrict__ dec, const int* __restrict__ frame,
                                    T* __restrict__ A, int Tn, int Hd, const int* __restrict__ dyn) {
    const int b = blockIdx.x;
    if (dyn) Tn = dyn[0];        // captured step (parakeet.cpp tdt_step_graph): the utterance's frame count lives in device memory
    const int f = min(frame[b], Tn - 1);
    for (int i = threadIdx.x; i < Hd; i += blockDim.x)
        Elem<T>::st(A + (long)b * Hd + i, fmaxf(encp[((long)b * Tn + f) * Hd + i] + dec[(long)b * Hd + i], 0.f));
}

// token = argmax(logits[:V]), duration = argmax(logits[V:V+ND]); blank & dur 0 -> 1; frame += dur
__global__ __launch_bounds__(256) void tdt_select_kernel(const float* __restrict__ logits, int ld, int V, int ND, int blank,
                                                         const int* __restrict__ enc_len, TdtState s, int* __restrict__ out_tok,
                                                         int* __restrict__ out_dur, int* __restrict__ out_frame, int max_steps,
                                                         const int* __restrict__ dyn) {
    if (dyn) max_steps = dyn[1];
    __shared__ float sv[4];
    __shared__ int si[4];
    const int b = blockIdx.x;
    const float* row = logits + (long)b * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float x = row[v];
        if (x > best || (x == best && v < bi)) { best = x; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int w = 1; w < 4; ++w)
        if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    int dur = 0;
    float db = row[V];
    for (int k = 1; k < ND; ++k)
        if (row[V + k] > db) { db = row[V + k]; dur = k; }
    const bool is_blank = bi == blank;
    if (is_blank && dur == 0) dur = 1;
    const int fin = s.finished[b];
    if (!fin) {
        const int n = s.n_steps[b];
        if (n < max_steps) {
            out_tok[(long)b * max_steps + n] = bi;
            out_dur[(long)b * max_steps + n] = dur;
            out_frame[(long)b * max_steps + n] = s.frame[b];
            s.n_steps[b] = n + 1;
        }
    }
    const int fr = s.frame[b] + dur;
    s.frame[b] = fr;
    const int nf = fin | (fr >= enc_len[b]);
    s.finished[b] = nf;
    s.tok[b] = bi;
    s.adv[b] = is_blank ? 0 : 1;
    if (!nf) atomicAdd(s.n_active, 1);
}

// X[b] = [ emb[tok[b]] | h_prev[b] ]   (input of one LSTM layer; layer > 0 takes x from the layer below)
template <typename T>
__global__ void lstm_in_kernel(const T* __restrict__ emb, const int* __restrict__ tok, const float* __restrict__ xin,
                               const float* __restrict__ h, T* __restrict__ X, int Hd) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < Hd; i += blockDim.x) {
        const float xv = emb ? Elem<T>::ld(emb + (long)tok[b] * Hd + i) : xin[(long)b * Hd + i];
        Elem<T>::st(X + (long)b * 2 * Hd + i, xv);
        Elem<T>::st(X + (long)b * 2 * Hd + Hd + i, h[(long)b * Hd + i]);
    }
}

// gates [B][4Hd] (i, f, g, o) -> new (h, c) written to h_new/c_new (staging), committed only where adv[b] (or force)
__global__ void lstm_cell_kernel(const float* __restrict__ gates, const float* __restrict__ c, float* __restrict__ h_new,
                                 float* __restrict__ c_new, int Hd) {
    const int b = blockIdx.x;
    const float* g = gates + (long)b * 4 * Hd;
    for (int i = threadIdx.x; i < Hd; i += blockDim.x) {
        const float ig = 1.0f / (1.0f + expf(-g[i]));
        const float fg = 1.0f / (1.0f + expf(-g[Hd + i]));
        const float gg = tanhf(g[2 * Hd + i]);
        const float og = 1.0f / (1.0f + expf(-g[3 * Hd + i]));
        const float cn = fg * c[(long)b * Hd + i] + ig * gg;
        c_new[(long)b * Hd + i] = cn;
        h_new[(long)b * Hd + i] = og * tanhf(cn);
    }
}

// commit staged state where the row advanced: dst[b] = adv[b] || force ? src[b] : dst[b]   (n arrays of Hd)
__global__ void tdt_commit_kernel(float* const* __restrict__ dst, const float* const* __restrict__ src, int n_arrays,
                                  const int* __restrict__ adv, int force, int Hd) {
    const int b = blockIdx.x;
    if (!force && !adv[b]) return;
    for (int a = 0; a < n_arrays; ++a)
        for (int i = threadIdx.x; i < Hd; i += blockDim.x) dst[a][(long)b * Hd + i] = src[a][(long)b * Hd + i];
}

// ---- fused step kernels (round 6): 8 launches per TDT step instead of 14 - the same arithmetic, element for element -------------------
// The step is a chain of latency-bound launches (C2: ~560 steps x 14 launches = 39 of the 41 ms of a 30 s utterance's STT), so the launches
// that only move rows between two GEMMs ride along with their neighbours:
//   select  + lstm_in of layer 0          (the pick IS the embedding index; h0 is the committed state)
//   cell l  + lstm_in of layer l + 1      (its x is the h this cell just produced) / + the model-dtype cast in front of dec.dproj (last layer)
//   commit  + joint_in of the NEXT step   (frame[b] was advanced by this step's select; dec is the value this thread just committed or kept)
//             + the reset of the active-row counter (was a memset node per step)
template <typename T>
__global__ __launch_bounds__(256) void tdt_select_in_kernel(const float* __restrict__ logits, int ld, int V, int ND, int blank,
                                                            const int* __restrict__ enc_len, TdtState s, int* __restrict__ out_tok,
                                                            int* __restrict__ out_dur, int* __restrict__ out_frame, int max_steps,
                                                            const int* __restrict__ dyn, const T* __restrict__ emb, const float* __restrict__ h0,
                                                            T* __restrict__ X, int Hd) {
    if (dyn) max_steps = dyn[1];
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int s_tok;
    const int b = blockIdx.x;
    const float* row = logits + (long)b * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float x = row[v];
        if (x > best || (x == best && v < bi)) { best = x; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        int dur = 0;
        float db = row[V];
        for (int k = 1; k < ND; ++k)
            if (row[V + k] > db) { db = row[V + k]; dur = k; }
        const bool is_blank = bi == blank;
        if (is_blank && dur == 0) dur = 1;
        const int fin = s.finished[b];
        if (!fin) {
            const int n = s.n_steps[b];
            if (n < max_steps) {
                out_tok[(long)b * max_steps + n] = bi;
                out_dur[(long)b * max_steps + n] = dur;
                out_frame[(long)b * max_steps + n] = s.frame[b];
                s.n_steps[b] = n + 1;
            }
        }
        const int fr = s.frame[b] + dur;
        s.frame[b] = fr;
        const int nf = fin | (fr >= enc_len[b]);
        s.finished[b] = nf;
        s.tok[b] = bi;
        s.adv[b] = is_blank ? 0 : 1;
        if (!nf) atomicAdd(s.n_active, 1);
        s_tok = bi;
    }
    __syncthreads();
    const int tok = s_tok;
    for (int i = threadIdx.x; i < Hd; i += 256) {      // lstm_in_kernel of layer 0
        X[(long)b * 2 * Hd + i] = emb[(long)tok * Hd + i];
        Elem<T>::st(X + (long)b * 2 * Hd + Hd + i, h0[(long)b * Hd + i]);
    }
}

// lstm_cell_kernel + what the next GEMM reads: Xn[b] = [h_new[b] | h_next[b]] (h_next: the next layer's committed state) or, for the last
// layer (h_next == nullptr), Xn[b] = h_new[b] in the model dtype (the input of dec.dproj)
template <typename T>
__global__ void lstm_cell_out_kernel(const float* __restrict__ gates, const float* __restrict__ c, float* __restrict__ h_new,
                                     float* __restrict__ c_new, int Hd, T* __restrict__ Xn, const float* __restrict__ h_next) {
    const int b = blockIdx.x;
    const float* g = gates + (long)b * 4 * Hd;
    const long xs = h_next ? 2 * (long)Hd : (long)Hd;
    for (int i = threadIdx.x; i < Hd; i += blockDim.x) {
        const float ig = 1.0f / (1.0f + expf(-g[i]));
        const float fg = 1.0f / (1.0f + expf(-g[Hd + i]));
        const float gg = tanhf(g[2 * Hd + i]);
        const float og = 1.0f / (1.0f + expf(-g[3 * Hd + i]));
        const float cn = fg * c[(long)b * Hd + i] + ig * gg;
        const float hn = og * tanhf(cn);
        c_new[(long)b * Hd + i] = cn;
        h_new[(long)b * Hd + i] = hn;
        Elem<T>::st(Xn + (long)b * xs + i, hn);
        if (h_next) Elem<T>::st(Xn + (long)b * xs + Hd + i, h_next[(long)b * Hd + i]);
    }
}

// tdt_commit_kernel + tdt_joint_in_kernel of the next step + reset of the active-row counter; the LAST array of dst / src is dec / dec_new
template <typename T>
__global__ void tdt_commit_joint_kernel(float* const* __restrict__ dst, const float* const* __restrict__ src, int n_arrays,
                                        const int* __restrict__ adv, int force, int Hd, const float* __restrict__ encp,
                                        const int* __restrict__ frame, T* __restrict__ A, int Tn, const int* __restrict__ dyn, int* n_active) {
    const int b = blockIdx.x;
    if (dyn) Tn = dyn[0];
    const bool take = force || adv[b];
    const int f = min(frame[b], Tn - 1);
    if (n_active && b == 0 && threadIdx.x == 0) *n_active = 0;      // (null in the captured step: the host reads the counter AFTER the whole replay)
    for (int i = threadIdx.x; i < Hd; i += blockDim.x) {
        if (take)
            for (int a = 0; a + 1 < n_arrays; ++a) dst[a][(long)b * Hd + i] = src[a][(long)b * Hd + i];
        float d = dst[n_arrays - 1][(long)b * Hd + i];
        if (take) {
            d = src[n_arrays - 1][(long)b * Hd + i];
            dst[n_arrays - 1][(long)b * Hd + i] = d;
        }
        Elem<T>::st(A + (long)b * Hd + i, fmaxf(encp[((long)b * Tn + f) * Hd + i] + d, 0.f));
    }
}

// ---- host launchers -------------------------------------------------------------------------
#define S2S_DT(TT_, ...)                                        \
    do {                                                        \
        if (dtype == S2S_BF16) { using TT_ = __hip_bfloat16; __VA_ARGS__; } \
        else { using TT_ = float; __VA_ARGS__; }                \
    } while (0)

int sub_conv1(hipStream_t st, const void* x, const float* w, const float* b, void* out, const int* len_in, int B, int Tin,
              int F, int C, int dtype) {
    const int T1 = (Tin - 1) / 2 + 1, F1 = (F - 1) / 2 + 1;
    S2S_DT(TT, (sub_conv1_kernel<TT><<<dim3(T1, B), 256, sizeof(float) * 3 * (F + 2), st>>>((const TT*)x, w, b, (TT*)out, len_in, Tin, F, T1, F1, C)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int dwconv2d(hipStream_t st, const void* in, const float* w, const float* b, void* out, const int* len0, int halvings, int B,
             int Tin, int Fin, int C, int dtype) {
    const int Tout = (Tin - 1) / 2 + 1, Fout = (Fin - 1) / 2 + 1;
    S2S_DT(TT, (dwconv2d_kernel<TT><<<dim3(Tout, B), 256, 0, st>>>((const TT*)in, w, b, (TT*)out, len0, halvings, Tin, Fin, Tout, Fout, C)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int mask_rows(hipStream_t st, void* x, const int* len0, int halvings, int B, int Trows, long row_elems, int dtype) {
    S2S_DT(TT, (mask_rows_kernel<TT><<<dim3(Trows, B), 256, 0, st>>>((TT*)x, len0, halvings, Trows, row_elems)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int sub_lengths(hipStream_t st, const int* len0, int halvings, int* out, int B) {
    sub_lengths_kernel<<<cdiv(B, 64), 64, 0, st>>>(len0, halvings, out, B);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int glu_dwconv(hipStream_t st, const void* y, const float* w, const float* b, void* out, const int* len, int B, int Tn, int d,
               int ksize, int dtype) {
    S2S_REQUIRE(d % 64 == 0, "conv module: d_model %d must be a multiple of 64", d);
    S2S_REQUIRE(ksize == 9, "conv module: kernel size %d unsupported (9)", ksize);
    dim3 grid(cdiv(Tn, 32), d / 64, B);
    S2S_DT(TT, (glu_dwconv_kernel<TT, 9><<<grid, 256, 0, st>>>((const TT*)y, w, b, (TT*)out, len, Tn, d)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int add_rowvec(hipStream_t st, const void* in, long ld_in, const float* vec, void* out, long rows, int cols, int dtype) {
    const int blocks = (int)std::min<long>(cdiv(rows * cols, 256), 4096);
    S2S_DT(TT, (add_rowvec_kernel<TT><<<blocks, 256, 0, st>>>((const TT*)in, ld_in, vec, (TT*)out, rows, cols)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int tdt_joint_in(hipStream_t st, const float* encp, const float* dec, const int* frame, void* A, int B, int Tn, int Hd, int dtype, const int* dyn) {
    S2S_DT(TT, (tdt_joint_in_kernel<TT><<<B, 256, 0, st>>>(encp, dec, frame, (TT*)A, Tn, Hd, dyn)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int tdt_select(hipStream_t st, const float* logits, int ld, int V, int ND, int blank, const int* enc_len, const TdtState& s,
               int* out_tok, int* out_dur, int* out_frame, int max_steps, int B, const int* dyn) {
    tdt_select_kernel<<<B, 256, 0, st>>>(logits, ld, V, ND, blank, enc_len, s, out_tok, out_dur, out_frame, max_steps, dyn);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int lstm_in(hipStream_t st, const void* emb, const int* tok, const float* xin, const float* h, void* X, int B, int Hd, int dtype) {
    S2S_DT(TT, (lstm_in_kernel<TT><<<B, 256, 0, st>>>((const TT*)emb, tok, xin, h, (TT*)X, Hd)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int lstm_cell(hipStream_t st, const float* gates, const float* c, float* h_new, float* c_new, int B, int Hd) {
    lstm_cell_kernel<<<B, 256, 0, st>>>(gates, c, h_new, c_new, Hd);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int tdt_select_in(hipStream_t st, const float* logits, int ld, int V, int ND, int blank, const int* enc_len, const TdtState& s, int* out_tok, int* out_dur,
                  int* out_frame, int max_steps, int B, const int* dyn, const void* emb, const float* h0, void* X, int Hd, int dtype) {
    S2S_DT(TT, (tdt_select_in_kernel<TT><<<B, 256, 0, st>>>(logits, ld, V, ND, blank, enc_len, s, out_tok, out_dur, out_frame, max_steps, dyn, (const TT*)emb, h0, (TT*)X, Hd)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int lstm_cell_out(hipStream_t st, const float* gates, const float* c, float* h_new, float* c_new, int B, int Hd, void* Xn, const float* h_next, int dtype) {
    S2S_DT(TT, (lstm_cell_out_kernel<TT><<<B, 256, 0, st>>>(gates, c, h_new, c_new, Hd, (TT*)Xn, h_next)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int tdt_commit_joint(hipStream_t st, float* const* dst, const float* const* src, int n_arrays, const int* adv, int force, int B, int Hd, const float* encp,
                     const int* frame, void* A, int Tn, const int* dyn, int* n_active, int dtype) {
    S2S_DT(TT, (tdt_commit_joint_kernel<TT><<<B, 256, 0, st>>>(dst, src, n_arrays, adv, force, Hd, encp, frame, (TT*)A, Tn, dyn, n_active)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int tdt_commit(hipStream_t st, float* const* dst, const float* const* src, int n_arrays, const int* adv, int force, int B, int Hd) {
    tdt_commit_kernel<<<B, 256, 0, st>>>(dst, src, n_arrays, adv, force, Hd);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

}  // namespace s2s
