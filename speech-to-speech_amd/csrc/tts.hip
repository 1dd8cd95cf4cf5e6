// Kernels for the Qwen3-TTS path: decoder-stack helpers (RoPE + per-head q/k RMSNorm, KV append, SwiGLU),
// code-embedding gathers for the talker / code-predictor loop, and the codec-decoder elementwise pieces
// (SnakeBeta, causal depthwise conv + LayerNorm of the ConvNeXt block, clamp) plus the 24 kHz -> 16 kHz
// polyphase resampler with int16 conversion of the reference's _stream() post-processing.
#include "ops.h"

#include <algorithm>

namespace s2s {

#define S2S_DT(TT_, ...)                                                    \
    do {                                                                    \
        if (dtype == S2S_BF16) { using TT_ = __hip_bfloat16; __VA_ARGS__; } \
        else { using TT_ = float; __VA_ARGS__; }                            \
    } while (0)

// In-place on fused projection rows [rows][(H + 2*Hk) * D]: per head of q and k: optional RMSNorm over D
// (weight qw / kw), then rotary embedding at position pos0[b] + t (row = b*T + t).  One wave per (row, head).
// cos/sin: f32 [max_pos][D/2].  grid = (ceil((H+Hk)/4), rows), block 256.
template <typename T>
__global__ __launch_bounds__(256) void rope_qknorm_kernel(T* __restrict__ qkv, int T_len, int H, int Hk, int D,
                                                          const float* __restrict__ qw, const float* __restrict__ kw, float eps,
                                                          const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                          const int* __restrict__ pos0, T* __restrict__ kc, T* __restrict__ vc, long c_bs) {
    const int head = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (head >= H + 2 * Hk) return;
    const int row = blockIdx.y, lane = threadIdx.x & 63;
    const int b = row / T_len, t = row - b * T_len;
    const int pos = (pos0 ? pos0[b] : 0) + t;
    T* x = qkv + (long)row * (H + 2 * Hk) * D + (long)head * D;
    if (head >= H + Hk) {     // V head: append to the cache (kc != nullptr) - no rotation
        if (vc) {
            T* dst = vc + (long)b * c_bs + ((long)pos * Hk + (head - H - Hk)) * D;
            for (int i = lane; i < D; i += 64) dst[i] = x[i];
        }
        return;
    }
    const float* w = head < H ? qw : kw;
    const int half = D >> 1;
    float a = 0.f, c = 0.f;
    if (lane < half) {
        a = Elem<T>::ld(x + lane);
        c = Elem<T>::ld(x + lane + half);
    }
    if (w) {
        const float ss = wave_sum(a * a + c * c);
        const float r = rsqrtf(ss / (float)D + eps);
        if (lane < half) {
            a = a * r * w[lane];
            c = c * r * w[lane + half];
        }
    }
    if (lane < half) {
        const float cs = cosT[(long)pos * half + lane], sn = sinT[(long)pos * half + lane];
        const float r0 = a * cs - c * sn, r1 = c * cs + a * sn;
        Elem<T>::st(x + lane, r0);
        Elem<T>::st(x + lane + half, r1);
        if (kc && head >= H) {   // rotated K head -> cache row of this position
            T* dst = kc + (long)b * c_bs + ((long)pos * Hk + (head - H)) * D;
            Elem<T>::st(dst + lane, r0);
            Elem<T>::st(dst + lane + half, r1);
        }
    }
}

// kc / vc (optional): the per-layer KV cache [B][max_pos][Hk*D]; when given, the rotated K and the V of every row are
// appended at their position in the same launch.
int rope_qknorm(hipStream_t st, void* qkv, int rows, int T_len, int H, int Hk, int D, const float* qw, const float* kw, float eps,
                const float* cosT, const float* sinT, const int* pos0, void* kc, void* vc, long c_bs, int dtype) {
    S2S_REQUIRE(D <= 128 && D % 2 == 0, "rope: head_dim %d unsupported", D);
    dim3 grid(cdiv(H + 2 * Hk, 4), rows);
    S2S_DT(TT, (rope_qknorm_kernel<TT><<<grid, 256, 0, st>>>((TT*)qkv, T_len, H, Hk, D, qw, kw, eps, cosT, sinT, pos0, (TT*)kc, (TT*)vc, c_bs)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// cache[b][pos0[b] + t][:] = src row (b*T + t) columns [k_off, k_off + n) / [v_off, v_off + n)
template <typename T>
__global__ void kv_append_rows_kernel(const T* __restrict__ src, long src_rs, int k_off, int v_off, T* __restrict__ kc,
                                      T* __restrict__ vc, long c_bs, const int* __restrict__ pos0, int T_len, int n) {
    const int row = blockIdx.x;
    const int b = row / T_len, t = row - b * T_len;
    const long p = (pos0 ? pos0[b] : 0) + t;
    const T* s = src + (long)row * src_rs;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        kc[(long)b * c_bs + p * n + i] = s[k_off + i];
        vc[(long)b * c_bs + p * n + i] = s[v_off + i];
    }
}

int kv_append_rows(hipStream_t st, const void* src, long src_rs, int k_off, int v_off, void* kc, void* vc, long c_bs,
                   const int* pos0, int B, int T_len, int n, int dtype) {
    S2S_DT(TT, (kv_append_rows_kernel<TT><<<B * T_len, 256, 0, st>>>((const TT*)src, src_rs, k_off, v_off, (TT*)kc, (TT*)vc, c_bs, pos0, T_len, n)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// out[r][c] = silu(gu[r][c]) * gu[r][ffn + c]
template <typename T>
__global__ void swiglu_kernel(const T* __restrict__ gu, T* __restrict__ out, long rows, int ffn) {
    const long total = rows * ffn;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / ffn;
        const int c = (int)(i - r * ffn);
        const float g = Elem<T>::ld(gu + r * 2 * ffn + c), u = Elem<T>::ld(gu + r * 2 * ffn + ffn + c);
        Elem<T>::st(out + i, g / (1.0f + expf(-g)) * u);
    }
}

int swiglu(hipStream_t st, const void* gu, void* out, long rows, int ffn, int dtype) {
    const int blocks = (int)std::min<long>(cdiv(rows * ffn, 256), 4096);
    S2S_DT(TT, (swiglu_kernel<TT><<<blocks, 256, 0, st>>>((const TT*)gu, (TT*)out, rows, ffn)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

__global__ void add_int_kernel(int* p, int v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += v;
}
int add_int(hipStream_t st, int* p, int v, int n) {
    add_int_kernel<<<cdiv(n, 64), 64, 0, st>>>(p, v, n);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// dst[b][:] (f32, += when accumulate) = table[ids[b*id_stride] + id_offset][:]  (+ addvec)
template <typename T>
__global__ void gather_embed_kernel(const T* __restrict__ table, const int* __restrict__ ids, int id_stride, int id_offset,
                                    float* __restrict__ dst, long dst_rs, int dim, int accumulate, const T* __restrict__ addvec) {
    const int b = blockIdx.x;
    const long id = (long)ids[(long)b * id_stride] + id_offset;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) {
        float v = id >= 0 ? Elem<T>::ld(table + id * dim + i) : 0.f;
        if (addvec) v += Elem<T>::ld(addvec + i);
        if (accumulate) v += dst[(long)b * dst_rs + i];
        dst[(long)b * dst_rs + i] = v;
    }
}

int gather_embed(hipStream_t st, const void* table, const int* ids, int id_stride, int id_offset, float* dst, long dst_rs, int B,
                 int dim, int accumulate, const void* addvec, int dtype) {
    if (B == 0) return S2S_OK;
    S2S_DT(TT, (gather_embed_kernel<TT><<<B, 256, 0, st>>>((const TT*)table, ids, id_stride, id_offset, dst, dst_rs, dim, accumulate, (const TT*)addvec)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// out[n][k] = W[n][k] * gamma[k]  (RMSNorm weight folded into the consuming projection's input columns)
template <typename T>
__global__ void fold_cols_kernel(const T* __restrict__ W, const float* __restrict__ gamma, T* __restrict__ out, long rows, int K) {
    const long total = rows * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        Elem<T>::st(out + i, Elem<T>::ld(W + i) * gamma[i % K]);
}

int fold_norm_weight(hipStream_t st, const void* W, const float* gamma, void* out, long rows, int K, int dtype) {
    const int blocks = (int)std::min<long>(cdiv(rows * K, 256), 8192);
    S2S_DT(TT, (fold_cols_kernel<TT><<<blocks, 256, 0, st>>>((const TT*)W, gamma, (TT*)out, rows, K)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// element (row m, column k) of a fragment-major [rows][K] activation matrix, ksteps = K / 32 (gemm_dec.hip operand layout)
__device__ __forceinline__ long frag_index_d(int m, int k, int ksteps) {
    return ((((long)(m >> 4) * ksteps + (k >> 5)) * 4 + ((k >> 3) & 3)) * 16 + (m & 15)) * 8 + (k & 7);
}

// ---- fused greedy picks of the frame loop ----------------------------------------------------------------
// Teacher forcing / logit capture (s2s_tts_generate_forced): the emitted code is replaced by forced[b][frame][g] when that is
// >= 0, and the raw logits row is copied to log[b][frame - f0][g][0 .. V).  Both pointers are null on the serving path.
__device__ __forceinline__ void pick_debug(const PickDebug& d, const float* __restrict__ row, int V, int b, int frame, int g, int max_frames, int G,
                                           int& code) {
    if (d.log && frame >= d.f0 && frame < d.f0 + d.n_log) {
        float* dst = d.log + (((long)b * d.n_log + (frame - d.f0)) * G + g) * d.Vlog;
        for (int i = threadIdx.x; i < V; i += blockDim.x) dst[i] = row[i];
    }
    if (d.forced) {
        const int f = d.forced[((long)b * max_frames + frame) * G + g];
        if (f >= 0) code = f;
    }
}

// ---- seeded sampling (one 256-thread block per session row) ---------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {   // splitmix64 finaliser: counter-based, stateless
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Sample one token from logits row `row` (V <= 4096) with transformers' processor order: repetition penalty (over hist[0 .. n_hist),
// stride hist_stride; negative entries skipped) -> temperature -> top-k (ties at the k-th value are kept, like HF) -> top-p (smallest
// set of most likely tokens whose mass exceeds top_p, at least one) -> multinomial with the counter-based uniform u(seed, key).
// mask / mask2: optional suppress masks (-inf).  Dynamic LDS: float[V] + float[8].  Every thread returns the same token.
__device__ int block_sample_256(const float* __restrict__ row, int V, const unsigned char* __restrict__ mask, const unsigned char* __restrict__ mask2,
                                const int* __restrict__ hist, int n_hist, int hist_stride, const SampleParams sp, unsigned long long key) {
    extern __shared__ __attribute__((aligned(16))) float smp[];
    float* sl = smp;            // [V] processed logits, then probabilities
    float* red = smp + V;       // [8]
    const int tid = threadIdx.x;
    for (int v = tid; v < V; v += 256) {
        float x = row[v];
        if ((mask && mask[v]) || (mask2 && mask2[v])) x = -INFINITY;
        sl[v] = x;
    }
    __syncthreads();
    if (sp.rep_penalty != 1.0f && hist) {   // each distinct previously emitted token is penalised once (first occurrence does it)
        for (int f = tid; f < n_hist; f += 256) {
            const int t = hist[(long)f * hist_stride];
            if (t < 0 || t >= V) continue;
            bool first = true;
            for (int q = 0; q < f; ++q)
                if (hist[(long)q * hist_stride] == t) { first = false; break; }
            if (first) {
                const float x = sl[t];
                sl[t] = x > 0.f ? x / sp.rep_penalty : x * sp.rep_penalty;
            }
        }
        __syncthreads();
    }
    const float inv_t = 1.0f / sp.temperature;
    float mx = -INFINITY;
    for (int v = tid; v < V; v += 256) {
        const float x = sl[v] * inv_t;
        sl[v] = x;
        mx = fmaxf(mx, x);
    }
    mx = wave_max(mx);
    __syncthreads();
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    // top-k: largest threshold (in the order-preserving integer image of float) that still keeps >= k values
    if (sp.top_k > 0 && sp.top_k < V) {
        unsigned lo = 0u, hi = 0xffffffffu;          // invariant: count(key >= lo) >= k
        for (int it = 0; it < 32; ++it) {
            const unsigned mid = lo + (unsigned)(((unsigned long long)hi - lo + 1) >> 1);
            float c = 0.f;
            for (int v = tid; v < V; v += 256) c += f32_ordered(sl[v]) >= mid ? 1.f : 0.f;
            c = block_sum_256(c, red);
            if (c >= (float)sp.top_k) lo = mid; else hi = mid - 1;
            if (lo == hi) break;
        }
        __syncthreads();
        for (int v = tid; v < V; v += 256)
            if (f32_ordered(sl[v]) < lo) sl[v] = -INFINITY;
        __syncthreads();
    }
    // probabilities
    float z = 0.f;
    for (int v = tid; v < V; v += 256) {
        const float p = sl[v] == -INFINITY ? 0.f : __expf(sl[v] - mx);
        sl[v] = p;
        z += p;
    }
    z = block_sum_256(z, red);
    // top-p: keep p_i iff the mass of tokens at least as likely as i ... i.e. drop the low tail whose cumulative mass <= 1 - top_p
    float thr = 0.f;      // probabilities >= thr are kept
    if (sp.top_p < 1.0f) {
        const float tail = (1.0f - sp.top_p) * z;
        unsigned lo = 0u, hi = __float_as_uint(1.0f);        // probabilities are in [0, 1]: positive floats order like their bits
        // largest threshold whose strictly-lower tail mass is <= tail (the most likely token always stays)
        for (int it = 0; it < 31; ++it) {
            const unsigned mid = lo + ((hi - lo + 1) >> 1);
            const float t = __uint_as_float(mid);
            float m = 0.f;
            for (int v = tid; v < V; v += 256) m += sl[v] < t ? sl[v] : 0.f;
            m = block_sum_256(m, red);
            if (m <= tail) lo = mid; else hi = mid - 1;
            if (lo == hi) break;
        }
        thr = __uint_as_float(lo);
        float pm = 0.f;
        for (int v = tid; v < V; v += 256) pm = fmaxf(pm, sl[v]);
        pm = wave_max(pm);
        __syncthreads();
        if ((tid & 63) == 0) red[4 + (tid >> 6)] = pm;
        __syncthreads();
        thr = fminf(thr, fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
        __syncthreads();
        float z2 = 0.f;
        for (int v = tid; v < V; v += 256) {
            if (sl[v] < thr) sl[v] = 0.f;
            z2 += sl[v];
        }
        z = block_sum_256(z2, red);
    }
    // multinomial in index order: thread t owns the contiguous range [t*per, (t+1)*per)
    const unsigned long long r = mix64(sp.seed ^ mix64(key));
    const float u = (float)((r >> 40) + 0.5) * (1.0f / 16777216.0f) * z;     // in (0, z)
    const int per = (V + 255) / 256;
    float loc = 0.f;
    for (int j = 0; j < per; ++j) {
        const int v = tid * per + j;
        if (v < V) loc += sl[v];
    }
    // exclusive scan of the 256 partial sums (wave scan + wave offsets)
    float inc = loc;
    for (int o = 1; o < 64; o <<= 1) {
        const float n = __shfl_up(inc, o, 64);
        if ((tid & 63) >= o) inc += n;
    }
    __syncthreads();
    if ((tid & 63) == 63) red[tid >> 6] = inc;
    __syncthreads();
    float base = 0.f;
    for (int w = 0; w < (tid >> 6); ++w) base += red[w];
    const float excl = base + inc - loc;
    __shared__ int chosen;
    if (tid == 0) chosen = 0x7fffffff;
    __syncthreads();
    if (loc > 0.f && u >= excl && u < excl + loc) {
        float c = excl;
        int pick = -1;
        for (int j = 0; j < per; ++j) {
            const int v = tid * per + j;
            if (v < V && sl[v] > 0.f) {
                pick = v;
                c += sl[v];
                if (u < c) break;
            }
        }
        // float rounding can make two threads' [excl, excl + loc) intervals overlap by an ulp: the lowest index wins, always
        if (pick >= 0) atomicMin(&chosen, pick);
    }
    __syncthreads();
    if (chosen == 0x7fffffff) {        // u fell on a rounding seam: take the most likely kept token
        __syncthreads();
        int best = block_argmax_256(sl, V, nullptr, nullptr);
        return best;
    }
    return chosen;
}

// Talker pick (one block per session): code 0 = argmax over the codec vocabulary (specials suppressed; eos too while
// frame < min_frames), eos / frame bookkeeping, then the inputs that depend on it:
//   xnext[b] = talker_emb[code0] + pad_embed          (next talker input, residual-group embeddings are added later)
//              ... + trail[b][frame] instead of pad_embed while frame < n_trail[b]: streaming-text mode (the reference's
//              non_streaming_mode = False; sibling modeling_qwen3_omni_moe.py:3169-3174: the text that was not pre-filled arrives one row per frame)
//   pin[b][0] = hn[b], pin[b][1] = talker_emb[code0]  (code predictor prefill rows; f32 and model-dtype copies)
template <typename T>
__global__ __launch_bounds__(256) void talker_pick_kernel(const float* __restrict__ logits, int V, const unsigned char* __restrict__ sup_mask,
                                                          const unsigned char* __restrict__ eos_mask, int eos,
                                                          const int* __restrict__ frame_b, const int* __restrict__ min_b, const int* __restrict__ utt_b,
                                                          int* __restrict__ code0, int* __restrict__ finished, int* __restrict__ n_frames,
                                                          int* __restrict__ codes_out, int max_frames, int G,
                                                          const T* __restrict__ emb, const T* __restrict__ pad, const float* __restrict__ hn,
                                                          float* __restrict__ xnext, float* __restrict__ pin, T* __restrict__ pin_t, int E,
                                                          int pin_frag, PickDebug dbg, SampleParams sp, const T* __restrict__ trail,
                                                          const int* __restrict__ n_trail, int max_trail) {
    const int b = blockIdx.x;
    const int frame = frame_b[b];       // this SLOT's frame index: sessions admitted at different times share the launch
    const bool gate = frame < min_b[b];
    int code;
    if (sp.temperature > 0.f)      // seeded sampling, keyed on (seed, utterance id, frame): independent of the slot the session landed in;
                                   // the repetition penalty looks at this session's earlier code-0 picks
        code = block_sample_256(logits + (long)b * V, V, sup_mask, gate ? eos_mask : nullptr, codes_out + (long)b * max_frames * G, frame, G, sp,
                                ((unsigned long long)(unsigned)utt_b[b] << 40) ^ ((unsigned long long)frame << 8));
    else
        code = block_argmax_256(logits + (long)b * V, V, sup_mask, gate ? eos_mask : nullptr);
    pick_debug(dbg, logits + (long)b * V, V, b, frame, 0, max_frames, G, code);
    if (threadIdx.x == 0) {
        int f = finished[b];
        if (!f && code == eos) f = 1;
        finished[b] = f;
        code0[b] = code;
        // a finished / idle slot writes nothing: its frame index stands on its last frame (codes_out is -1 from the admission on)
        if (!f) {
            codes_out[((long)b * max_frames + frame) * G] = code;
            n_frames[b] = frame + 1;
        }
    }
    const T* row = emb + (long)code * E;
    const T* text = (trail && frame < n_trail[b]) ? trail + ((long)b * max_trail + frame) * E : pad;      // the text-side row of the next input
    for (int i = threadIdx.x; i < E; i += 256) {
        const float e = Elem<T>::ld(row + i);
        const float h = hn[(long)b * E + i];
        xnext[(long)b * E + i] = e + Elem<T>::ld(text + i);
        pin[((long)b * 2) * E + i] = h;
        pin[((long)b * 2 + 1) * E + i] = e;
        Elem<T>::st(pin_t + (pin_frag ? frag_index_d(b * 2, i, E >> 5) : ((long)b * 2) * E + i), h);
        Elem<T>::st(pin_t + (pin_frag ? frag_index_d(b * 2 + 1, i, E >> 5) : ((long)b * 2 + 1) * E + i), e);
    }
}

// Code-predictor pick for residual group g (one block per session): code = argmax, stored (or -1 once finished), and its
// embedding row both accumulated into the next talker input and written as the predictor's next input row.
template <typename T>
__global__ __launch_bounds__(256) void pred_pick_kernel(const float* __restrict__ logits, int V, long logits_rs, const int* __restrict__ frame_b,
                                                        const int* __restrict__ utt_b, const int* __restrict__ finished, int* __restrict__ codes_out, int max_frames,
                                                        int G, int g, const T* __restrict__ emb, float* __restrict__ xnext,
                                                        float* __restrict__ pin, T* __restrict__ pin_t, int E, int pin_frag, PickDebug dbg,
                                                        SampleParams sp) {
    const int b = blockIdx.x;
    const int frame = frame_b[b];
    int code;
    if (sp.temperature > 0.f)
        code = block_sample_256(logits + (long)b * logits_rs, V, nullptr, nullptr, nullptr, 0, 0, sp,
                                ((unsigned long long)(unsigned)utt_b[b] << 40) ^ ((unsigned long long)frame << 8) ^ (unsigned long long)g);
    else
        code = block_argmax_256(logits + (long)b * logits_rs, V, nullptr, nullptr);
    pick_debug(dbg, logits + (long)b * logits_rs, V, b, frame, g, max_frames, G, code);
    if (threadIdx.x == 0 && !finished[b]) codes_out[((long)b * max_frames + frame) * G + g] = code;
    const T* row = emb + (long)code * E;
    for (int i = threadIdx.x; i < E; i += 256) {
        const float e = Elem<T>::ld(row + i);
        xnext[(long)b * E + i] += e;
        pin[(long)b * E + i] = e;
        Elem<T>::st(pin_t + (pin_frag ? frag_index_d(b, i, E >> 5) : (long)b * E + i), e);
    }
}

// End of a frame, per slot: a live slot advances its cache position and frame index and retires itself when its frame budget is
// used up; a finished / idle slot stands still (its rows keep being computed - the launch geometry is fixed - but its position never
// leaves the cache and its frame index never leaves codes_out).
__global__ void tts_frame_end_kernel(int* pos_t, int* frame_b, int* finished, const int* budget_b, int max_pos, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B || finished[b]) return;
    pos_t[b] = min(pos_t[b] + 1, max_pos - 1);
    const int f = frame_b[b] + 1;
    if (f >= budget_b[b]) finished[b] = 1;      // budget reached: frame_b stays on the last written frame
    else frame_b[b] = f;
}

// Slot bookkeeping of the continuous-batching frame loop (admission / release of slots [slot0, slot0 + n)).
__global__ void tts_slots_set_kernel(int* pos_t, int* frame_b, int* min_b, int* budget_b, int* utt_b, int* finished, int* n_frames, int* code0,
                                     const int* prompt_len, const int* min_frames, const int* budget, int min_all, int budget_all, int utt0,
                                     int max_frames, int slot0, int n, int live) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = slot0 + i;
    if (live) {
        pos_t[s] = prompt_len[i];
        frame_b[s] = 0;
        min_b[s] = min_frames ? min_frames[i] : min_all;
        budget_b[s] = min(budget ? budget[i] : budget_all, max_frames);
        utt_b[s] = utt0 + i;
        finished[s] = 0;
        n_frames[s] = 0;
        code0[s] = 0;
    } else {
        finished[s] = 1;
    }
}

// codes of frames [frame0, frame0 + F) of slots [slot0, slot0 + n): out [n][F][G]
// frame0_rows (optional, int32[n]): per-row first frame instead of frame0 (rows of different ages in one call)
__global__ void tts_read_codes_kernel(const int* __restrict__ codes_out, int max_frames, int G, int slot0, int frame0, const int* __restrict__ frame0_rows, int F,
                                      int* __restrict__ out, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int g = (int)(i % G);
    const long r = i / G;
    const int f = (int)(r % F), b = (int)(r / F);
    const int fr = (frame0_rows ? frame0_rows[b] : frame0) + f;
    out[i] = fr < max_frames ? codes_out[((long)(slot0 + b) * max_frames + fr) * G + g] : -1;
}

int talker_pick(hipStream_t st, const float* logits, int V, const unsigned char* sup_mask, const unsigned char* eos_mask, int eos,
                const int* frame_b, const int* min_b, const int* utt_b, int* code0, int* finished, int* n_frames, int* codes_out, int max_frames, int G,
                const void* emb, const void* pad, const float* hn, float* xnext, float* pin, void* pin_t, int E, bool pin_frag, int B, int dtype,
                const PickDebug& dbg, const SampleParams& sp, const void* trail, const int* n_trail, int max_trail) {
    S2S_REQUIRE(sp.temperature <= 0.f || V <= 4096, "talker_pick: sampling supports vocabularies up to 4096 entries");
    const size_t lds = sp.temperature > 0.f ? sizeof(float) * (V + 8) : 0;
    S2S_DT(TT, (talker_pick_kernel<TT><<<B, 256, lds, st>>>(logits, V, sup_mask, eos_mask, eos, frame_b, min_b, utt_b, code0, finished, n_frames,
                                                           codes_out, max_frames, G, (const TT*)emb, (const TT*)pad, hn, xnext, pin, (TT*)pin_t, E,
                                                           pin_frag ? 1 : 0, dbg, sp, (const TT*)trail, n_trail, max_trail)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// trailing text rows of n admitted sessions: rows f32 [n][Tr][E] -> trail[slot0 + i][0 .. Tr) in the model dtype, n_trail[slot0 + i] = n_rows[i]
template <typename T>
__global__ void trail_set_kernel(const float* __restrict__ rows, const int* __restrict__ n_rows, int Tr, int E, T* __restrict__ trail, int* __restrict__ n_trail,
                                 int max_trail, int slot0) {
    const int i = blockIdx.y, t = blockIdx.x;
    const int nr = min(n_rows[i], min(Tr, max_trail));
    if (t == 0 && threadIdx.x == 0) n_trail[slot0 + i] = nr;
    if (t >= nr) return;
    for (int k = threadIdx.x; k < E; k += blockDim.x) Elem<T>::st(trail + ((long)(slot0 + i) * max_trail + t) * E + k, rows[((long)i * Tr + t) * E + k]);
}
int trail_set(hipStream_t st, const float* rows, const int* n_rows, int n, int Tr, int E, void* trail, int* n_trail, int max_trail, int slot0, int dtype) {
    S2S_DT(TT, (trail_set_kernel<TT><<<dim3(Tr, n), 256, 0, st>>>(rows, n_rows, Tr, E, (TT*)trail, n_trail, max_trail, slot0)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int pred_pick(hipStream_t st, const float* logits, int V, long logits_rs, const int* frame_b, const int* utt_b, const int* finished, int* codes_out, int max_frames, int G,
              int g, const void* emb, float* xnext, float* pin, void* pin_t, int E, bool pin_frag, int B, int dtype, const PickDebug& dbg,
              const SampleParams& sp) {
    S2S_REQUIRE(sp.temperature <= 0.f || V <= 4096, "pred_pick: sampling supports vocabularies up to 4096 entries");
    const size_t lds = sp.temperature > 0.f ? sizeof(float) * (V + 8) : 0;
    S2S_DT(TT, (pred_pick_kernel<TT><<<B, 256, lds, st>>>(logits, V, logits_rs, frame_b, utt_b, finished, codes_out, max_frames, G, g, (const TT*)emb, xnext,
                                                         pin, (TT*)pin_t, E, pin_frag ? 1 : 0, dbg, sp)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int tts_frame_end(hipStream_t st, int* pos_t, int* frame_b, int* finished, const int* budget_b, int max_pos, int B) {
    tts_frame_end_kernel<<<cdiv(B, 64), 64, 0, st>>>(pos_t, frame_b, finished, budget_b, max_pos, B);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int tts_slots_set(hipStream_t st, int* pos_t, int* frame_b, int* min_b, int* budget_b, int* utt_b, int* finished, int* n_frames, int* code0,
                  const int* prompt_len, const int* min_frames, const int* budget, int min_all, int budget_all, int utt0, int max_frames, int slot0,
                  int n, bool live) {
    if (n <= 0) return S2S_OK;
    tts_slots_set_kernel<<<cdiv(n, 64), 64, 0, st>>>(pos_t, frame_b, min_b, budget_b, utt_b, finished, n_frames, code0, prompt_len, min_frames, budget,
                                                     min_all, budget_all, utt0, max_frames, slot0, n, live ? 1 : 0);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// 24 kHz samples of this chunk that belong to VALID frames of each slot: chunk_samples when all F frames are valid, else what the
// valid frames contribute (a first chunk is shorter than F * up by the codec's left trim)
__global__ void tts_valid_samples_kernel(const int* __restrict__ n_frames, int slot0, int n, int frame0, const int* __restrict__ frame0_rows, int F, int up,
                                         int chunk_samples, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int valid = min(max(n_frames[slot0 + i] - (frame0_rows ? frame0_rows[i] : frame0), 0), F);
    out[i] = valid >= F ? chunk_samples : max(0, valid * up - (F * up - chunk_samples));
}

int tts_valid_samples(hipStream_t st, const int* n_frames, int slot0, int n, int frame0, const int* frame0_rows, int F, int up, int chunk_samples, int* out) {
    tts_valid_samples_kernel<<<cdiv(n, 64), 64, 0, st>>>(n_frames, slot0, n, frame0, frame0_rows, F, up, chunk_samples, out);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int tts_read_codes(hipStream_t st, const int* codes_out, int max_frames, int G, int slot0, int n, int frame0, const int* frame0_rows, int F, int* out) {
    const long total = (long)n * F * G;
    if (total <= 0) return S2S_OK;
    tts_read_codes_kernel<<<(unsigned)cdiv(total, 256), 256, 0, st>>>(codes_out, max_frames, G, slot0, frame0, frame0_rows, F, out, total);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// h[b][t][:] = mean_q table[q*cb + codes[b][t][q]][:]   (codes frame-major [B][F][Q]; negative codes count as 0)
template <typename T>
__global__ void code_embed_mean_kernel(const T* __restrict__ table, const int* __restrict__ codes, float* __restrict__ h, int Q,
                                       int cb, int dim) {
    const long row = blockIdx.x;
    const int* c = codes + row * Q;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) {
        float s = 0.f;
        for (int q = 0; q < Q; ++q) s += Elem<T>::ld(table + ((long)q * cb + max(c[q], 0)) * dim + i);
        h[row * dim + i] = s / (float)Q;
    }
}

int code_embed_mean(hipStream_t st, const void* table, const int* codes, float* h, long rows, int Q, int cb, int dim, int dtype) {
    S2S_DT(TT, (code_embed_mean_kernel<TT><<<(unsigned)rows, 256, 0, st>>>((const TT*)table, codes, h, Q, cb, dim)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// ---- codec decoder elementwise -------------------------------------------------------------------------
// Activation buffers are time-major with `lpad` zero rows in front of every item: [B][lpad + T][C].
// ConvNeXt head: causal depthwise conv k7 over time (channels-last) followed by LayerNorm over channels.
// x [B][lpad+T][C] -> y [B][T][C] (no padding, GEMM input).  One wave per output row; C <= 2048.
template <typename T>
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ g, const float* __restrict__ be, T* __restrict__ y,
                                                         int Tn, int C, int lpad, float eps) {
    const int lane = threadIdx.x & 63, b = blockIdx.y;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= Tn) return;
    const T* xb = x + ((long)b * (lpad + Tn) + lpad + t) * C;
    constexpr int PER = 32;
    float v[PER];
    float s = 0.f;
    const int n = (C + 63) / 64;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        v[u] = 0.f;
        if (u < n) {
            const int c = lane + u * 64;
            if (c < C) {
                float acc = bias[c];
#pragma unroll
                for (int k = 0; k < 7; ++k) acc = fmaf(w[c * 7 + k], Elem<T>::ld(xb + (long)(k - 6) * C + c), acc);
                v[u] = acc;
                s += acc;
            }
        }
    }
    s = wave_sum(s);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (u < n && lane + u * 64 < C) {
            const float d = v[u] - mean;
            q += d * d;
        }
    q = wave_sum(q);
    const float rstd = rsqrtf(q / (float)C + eps);
    T* yb = y + ((long)b * Tn + t) * C;
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (u < n) {
            const int c = lane + u * 64;
            if (c < C) Elem<T>::st(yb + c, (v[u] - mean) * rstd * g[c] + be[c]);
        }
}

int dwconv7_ln(hipStream_t st, const void* x, const float* w, const float* bias, const float* g, const float* be, void* y, int B,
               int Tn, int C, int lpad, float eps, int dtype) {
    S2S_REQUIRE(C <= 2048 && lpad >= 6, "convnext: C=%d lpad=%d unsupported", C, lpad);
    dim3 grid(cdiv(Tn, 4), B);
    S2S_DT(TT, (dwconv7_ln_kernel<TT><<<grid, 256, 0, st>>>((const TT*)x, w, bias, g, be, (TT*)y, Tn, C, lpad, eps)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// wav[b][i] = clamp(x[b][lpad + skip + i][0], -1, 1) for i < n
template <typename T>
__global__ void wav_out_kernel(const T* __restrict__ x, float* __restrict__ wav, int Tn, int lpad, int skip, int n) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        wav[(long)b * n + i] = fminf(fmaxf(Elem<T>::ld(x + (long)b * (lpad + Tn) + lpad + skip + i), -1.f), 1.f);
}

int wav_out(hipStream_t st, const void* x, float* wav, int B, int Tn, int lpad, int skip, int n, int dtype) {
    if (n <= 0) return S2S_OK;
    dim3 grid(std::min(cdiv(n, 256), 1024), B);
    S2S_DT(TT, (wav_out_kernel<TT><<<grid, 256, 0, st>>>((const TT*)x, wav, Tn, lpad, skip, n)));
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// ---- 24 kHz -> 16 kHz: scipy.signal.resample_poly(x, up=2, down=3) per chunk, then clip(x*32768) -> int16 ---
// y[m] = sum_k h[k] * xup[3m + off - k], xup[2i] = x[i]; h (taps, host-designed) includes the gain `up`.
__global__ void resample_2_3_int16_kernel(const float* __restrict__ x, long x_bs, const int* __restrict__ n_in, const float* __restrict__ h,
                                          int taps, int off, short* __restrict__ y16, float* __restrict__ yf, long y_bs, int n_out_max) {
    const int b = blockIdx.y;
    const int n = n_in ? n_in[b] : (int)x_bs;
    const int n_out = (2 * n + 2) / 3;   // ceil(2n/3)
    const float* xb = x + (long)b * x_bs;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < n_out_max; m += gridDim.x * blockDim.x) {
        float acc = 0.f;
        if (m < n_out) {
            const int c = 3 * m + off;
            // taps k = c - 2i with 0 <= k < taps; accumulate oldest sample first with separately rounded multiply
            // and add - the order and rounding of scipy's upfirdn (float32), so the int16 truncation matches it
            int i_lo = c - taps + 1;
            i_lo = i_lo <= 0 ? 0 : (i_lo + 1) >> 1;
            const int i_hi = min(c >> 1, n - 1);
            for (int i = i_lo; i <= i_hi; ++i) acc = __fadd_rn(acc, __fmul_rn(xb[i], h[c - 2 * i]));
        }
        if (yf) yf[(long)b * y_bs + m] = acc;
        if (y16) {
            const float s = fminf(fmaxf(acc * 32768.0f, -32768.0f), 32767.0f);
            y16[(long)b * y_bs + m] = (short)s;   // truncation toward zero == numpy astype(int16)
        }
    }
}

int resample_2_3_int16(hipStream_t st, const float* x, long x_bs, const int* n_in, const float* h, int taps, int off, short* y16,
                       float* yf, long y_bs, int n_out_max, int B) {
    if (B == 0 || n_out_max <= 0) return S2S_OK;
    dim3 grid(std::min(cdiv(n_out_max, 256), 1024), B);
    resample_2_3_int16_kernel<<<grid, 256, 0, st>>>(x, x_bs, n_in, h, taps, off, y16, yf, y_bs, n_out_max);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// ---- general polyphase resampler for the API edge (client rate <-> 16 kHz pipeline rate) -----------------------------------------
// scipy.signal.resample_poly(x, up, down) per row, then clip(y * 32768).astype(int16): the reference's api/openai_realtime/utils.py:5-12.
// y[m] = sum_i x[i] * h[down*m + off - up*i]; h (taps, host-designed like scipy: firwin(2*10*max(up,down)+1, 1/max(up,down), kaiser 5) * up),
// off = (taps - 1) / 2.  IN = short: samples are int16 / 32768 (exact).  Same tap order / separately rounded multiply-add as upfirdn (float32).
template <typename IN>
__global__ void resample_poly_kernel(const IN* __restrict__ x, long x_bs, const int* __restrict__ n_in, int up, int down, const float* __restrict__ h,
                                     int taps, int off, short* __restrict__ y16, float* __restrict__ yf, long y_bs, int n_out_max) {
    const int b = blockIdx.y;
    const int n = n_in ? n_in[b] : (int)x_bs;
    const int n_out = (int)(((long)n * up + down - 1) / down);
    const IN* xb = x + (long)b * x_bs;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < n_out_max; m += gridDim.x * blockDim.x) {
        float acc = 0.f;
        if (m < n_out) {
            const long c = (long)down * m + off;
            long lo = c - taps + 1;
            int i_lo = lo <= 0 ? 0 : (int)((lo + up - 1) / up);
            const int i_hi = (int)min((long)(c / up), (long)n - 1);
            for (int i = i_lo; i <= i_hi; ++i) {
                float xv;
                if constexpr (sizeof(IN) == 2) xv = (float)xb[i] * (1.0f / 32768.0f);
                else xv = xb[i];
                acc = __fadd_rn(acc, __fmul_rn(xv, h[c - (long)up * i]));
            }
        }
        if (yf) yf[(long)b * y_bs + m] = acc;
        if (y16) {
            const float s = fminf(fmaxf(acc * 32768.0f, -32768.0f), 32767.0f);
            y16[(long)b * y_bs + m] = (short)s;   // truncation toward zero == numpy astype(int16)
        }
    }
}

}  // namespace s2s

namespace s2s {
__global__ __launch_bounds__(256) void sample_rows_kernel(const float* __restrict__ logits, int V, const int* __restrict__ hist, int n_hist, SampleParams sp,
                                                          int* __restrict__ out) {
    const int b = blockIdx.x;
    const int t = block_sample_256(logits + (long)b * V, V, nullptr, nullptr, hist ? hist + (long)b * n_hist : nullptr, n_hist, 1, sp, (unsigned long long)b);
    if (threadIdx.x == 0) out[b] = t;
}
}  // namespace s2s

/* Public op (tests): one sampled token per row of logits f32 [B][V] (V <= 4096) with the frame loop's sampler; row b uses the
 * counter-based uniform of (seed, b).  hist_dev: optional int32 [B][n_hist] tokens the repetition penalty applies to. */
extern "C" int32_t s2s_op_sample_tokens(void* stream, const float* logits_dev, int32_t B, int32_t V, float temperature, int32_t top_k, float top_p,
                                        float repetition_penalty, uint64_t seed, const int32_t* hist_dev, int32_t n_hist, int32_t* out_dev) {
    S2S_REQUIRE(logits_dev && out_dev && B > 0 && V > 0 && V <= 4096 && temperature > 0.f && top_p > 0.f && repetition_penalty > 0.f, "op_sample_tokens: bad arguments");
    s2s::SampleParams sp{temperature, top_k, top_p, repetition_penalty, (unsigned long long)seed};
    s2s::sample_rows_kernel<<<B, 256, sizeof(float) * (V + 8), (hipStream_t)stream>>>(logits_dev, V, hist_dev, n_hist, sp, out_dev);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

/* Batched scipy.signal.resample_poly(x, up, down) + clip(y*32768).astype(int16) for the API edge (reference
 * api/openai_realtime/utils.py:5-12, used by handlers/audio.py:122-149 on input and :293-330 on output).  in_is_int16 != 0: in_dev is
 * int16 [B][in_stride] (treated as int16 / 32768, exactly like the reference), else float32.  n_out per row = ceil(n_in * up / down). */
extern "C" int32_t s2s_resample_poly_int16(void* stream, const void* in_dev, int32_t in_is_int16, int64_t in_stride, const int32_t* n_in_dev, int32_t up,
                                           int32_t down, const float* fir_dev, int32_t taps, int16_t* out16_dev, float* outf_dev, int64_t out_stride,
                                           int32_t n_out_max, int32_t B) {
    S2S_REQUIRE(in_dev && fir_dev && (out16_dev || outf_dev) && taps > 0 && (taps & 1) && up > 0 && down > 0, "resample_poly: bad arguments");
    if (B == 0 || n_out_max <= 0) return S2S_OK;
    dim3 grid(std::min(s2s::cdiv(n_out_max, 256), 1024), B);
    hipStream_t st = (hipStream_t)stream;
    if (in_is_int16)
        s2s::resample_poly_kernel<short><<<grid, 256, 0, st>>>((const short*)in_dev, (long)in_stride, n_in_dev, up, down, fir_dev, taps, (taps - 1) / 2,
                                                              (short*)out16_dev, outf_dev, (long)out_stride, n_out_max);
    else
        s2s::resample_poly_kernel<float><<<grid, 256, 0, st>>>((const float*)in_dev, (long)in_stride, n_in_dev, up, down, fir_dev, taps, (taps - 1) / 2,
                                                              (short*)out16_dev, outf_dev, (long)out_stride, n_out_max);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

extern "C" int32_t s2s_resample_24k16k_int16(void* stream, const float* pcm24_dev, int64_t in_stride, const int32_t* n_in_dev,
                                             const float* fir_dev, int32_t taps, int32_t delay, int16_t* out16_dev, float* outf_dev,
                                             int64_t out_stride, int32_t n_out_max, int32_t B) {
    S2S_REQUIRE(pcm24_dev && fir_dev && (out16_dev || outf_dev) && taps > 0, "resample: null argument");
    return s2s::resample_2_3_int16((hipStream_t)stream, pcm24_dev, (long)in_stride, n_in_dev, fir_dev, taps, delay, (short*)out16_dev,
                                   outf_dev, (long)out_stride, n_out_max, B);
}
