// Small data-movement / selection kernels used by the decode loops.
#include "ops.h"

#include <algorithm>

namespace s2s {

// x[b][:] = tok_emb[ids[b]][:] * emb_scale + pos_emb[pos[b]][:]   (f32 residual stream)
template <typename T>
__global__ void embed_kernel(const int* __restrict__ ids, const int* __restrict__ pos, const T* __restrict__ tok_emb,
                             const T* __restrict__ pos_emb, float* __restrict__ x, int d, float emb_scale) {
    const int b = blockIdx.x;
    const long id = ids[b];
    const long ps = pos ? pos[b] : 0;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        float v = Elem<T>::ld(tok_emb + id * d + i) * emb_scale;
        if (pos_emb) v += Elem<T>::ld(pos_emb + ps * d + i);
        x[(long)b * d + i] = v;
    }
}

int embed(hipStream_t st, const int* ids, const int* pos, const void* tok_emb, const void* pos_emb, float* x, int B,
          int d, float emb_scale, int dtype) {
    if (B == 0) return S2S_OK;
    if (dtype == S2S_BF16)
        embed_kernel<__hip_bfloat16><<<B, 256, 0, st>>>(ids, pos, (const __hip_bfloat16*)tok_emb, (const __hip_bfloat16*)pos_emb, x, d, emb_scale);
    else
        embed_kernel<float><<<B, 256, 0, st>>>(ids, pos, (const float*)tok_emb, (const float*)pos_emb, x, d, emb_scale);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// Copy the K and V column blocks of a fused projection row into the per-session caches at pos[b].
//   src row b: [.. k_off: k (n) .. v_off: v (n) ..];  cache (b, p) at cache + b*c_bs + p*n
template <typename T>
__global__ void kv_append_kernel(const T* __restrict__ src, long src_rs, int k_off, int v_off, T* __restrict__ kc,
                                 T* __restrict__ vc, long c_bs, const int* __restrict__ pos, int n) {
    const int b = blockIdx.x;
    const long p = pos[b];
    const T* s = src + (long)b * src_rs;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        kc[(long)b * c_bs + p * n + i] = s[k_off + i];
        vc[(long)b * c_bs + p * n + i] = s[v_off + i];
    }
}

int kv_append(hipStream_t st, const void* src, long src_rs, int k_off, int v_off, void* kc, void* vc, long c_bs,
              const int* pos, int B, int n, int dtype) {
    if (B == 0) return S2S_OK;
    if (dtype == S2S_BF16)
        kv_append_kernel<__hip_bfloat16><<<B, 256, 0, st>>>((const __hip_bfloat16*)src, src_rs, k_off, v_off, (__hip_bfloat16*)kc, (__hip_bfloat16*)vc, c_bs, pos, n);
    else
        kv_append_kernel<float><<<B, 256, 0, st>>>((const float*)src, src_rs, k_off, v_off, (float*)kc, (float*)vc, c_bs, pos, n);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// Greedy selection with suppression: out[b] = argmax_v (logits[b][v] unless suppressed).
//   mask: optional uint8[V], 1 = suppressed always;  begin_mask: optional uint8[V] applied when
//   step[b] == begin_step;  ties resolve to the lowest index (torch.argmax semantics).
__global__ __launch_bounds__(256) void argmax_suppress_kernel(const float* __restrict__ logits, long ld, int V,
                                                              const unsigned char* __restrict__ mask,
                                                              const unsigned char* __restrict__ begin_mask,
                                                              const int* __restrict__ step, int begin_step,
                                                              int* __restrict__ out, float* __restrict__ out_val) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int b = blockIdx.x;
    const float* row = logits + (long)b * ld;
    const bool at_begin = begin_mask && step && step[b] == begin_step;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += 256) {
        float x = row[v];
        if ((mask && mask[v]) || (at_begin && begin_mask[v])) x = -INFINITY;
        if (x > best || (x == best && v < bi)) {
            best = x;
            bi = v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sv[wave] = best;
        si[wave] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
                best = sv[w];
                bi = si[w];
            }
        out[b] = bi;
        if (out_val) out_val[b] = best;
    }
}

int argmax_suppress(hipStream_t st, const float* logits, long ld, int B, int V, const unsigned char* mask,
                    const unsigned char* begin_mask, const int* step, int begin_step, int* out, float* out_val) {
    if (B == 0) return S2S_OK;
    argmax_suppress_kernel<<<B, 256, 0, st>>>(logits, ld, V, mask, begin_mask, step, begin_step, out, out_val);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// f32 -> T cast of a contiguous buffer
template <typename T>
__global__ void cast_kernel(const float* __restrict__ in, T* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        Elem<T>::st(out + i, in[i]);
}

int cast_f32(hipStream_t st, const float* in, void* out, long n, int dtype) {
    if (n == 0) return S2S_OK;
    const int blocks = (int)std::min<long>(cdiv(n, 256), 4096);
    if (dtype == S2S_BF16) cast_kernel<__hip_bfloat16><<<blocks, 256, 0, st>>>(in, (__hip_bfloat16*)out, n);
    else cast_kernel<float><<<blocks, 256, 0, st>>>(in, (float*)out, n);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

}  // namespace s2s
