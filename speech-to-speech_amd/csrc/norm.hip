// LayerNorm / RMSNorm over the f32 residual stream, one wave per row, output in the
// GEMM input type.  Rows are read once (float4 per lane, coalesced), statistics are
// wave reductions (no LDS), two-pass variance on the register-resident row.
#include "common.h"

namespace s2s {

constexpr int NORM_MAX_V4 = 8;  // up to 64 lanes * 8 * 4 = 2048 columns

template <typename OutT, bool RMS>
__global__ __launch_bounds__(256) void norm_rows_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        OutT* __restrict__ y, int rows, int cols,
                                                        float eps, long x_rs, float* __restrict__ y_f32, int y_frag) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * x_rs;
    const int nv4 = cols >> 2;
    float4 v[NORM_MAX_V4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NORM_MAX_V4; ++i) {
        const int c = lane + i * 64;
        if (c < nv4) {
            v[i] = reinterpret_cast<const float4*>(xr)[c];
            if (RMS) s += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
            else s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s / (float)cols + eps);
    } else {
        mean = s / (float)cols;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NORM_MAX_V4; ++i) {
            const int c = lane + i * 64;
            if (c < nv4) {
                const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / (float)cols + eps);
    }
    OutT* yr = y + (long)row * cols;
#pragma unroll
    for (int i = 0; i < NORM_MAX_V4; ++i) {
        const int c = lane + i * 64;
        if (c < nv4) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[c];
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!RMS && beta) b = reinterpret_cast<const float4*>(beta)[c];
            float o0 = (v[i].x - mean) * rstd * g.x + b.x;
            float o1 = (v[i].y - mean) * rstd * g.y + b.y;
            float o2 = (v[i].z - mean) * rstd * g.z + b.z;
            float o3 = (v[i].w - mean) * rstd * g.w + b.w;
            if constexpr (sizeof(OutT) == 4) {
                reinterpret_cast<float4*>(yr)[c] = make_float4(o0, o1, o2, o3);
            } else {
                ushort4 p;
                p.x = f32_to_bf16_bits(o0);
                p.y = f32_to_bf16_bits(o1);
                p.z = f32_to_bf16_bits(o2);
                p.w = f32_to_bf16_bits(o3);
                if (y_frag) {   // fragment-major rows (gemm_dec.hip operand layout): 4 consecutive columns stay inside one 8-element chunk
                    const int k = c * 4;
                    *reinterpret_cast<ushort4*>(y + ((((long)(row >> 4) * (cols >> 5) + (k >> 5)) * 4 + ((k >> 3) & 3)) * 16 + (row & 15)) * 8 + (k & 7)) = p;
                } else {
                    reinterpret_cast<ushort4*>(yr)[c] = p;
                }
            }
            if (y_f32) reinterpret_cast<float4*>(y_f32 + (long)row * cols)[c] = make_float4(o0, o1, o2, o3);
        }
    }
}

int layernorm(hipStream_t st, const float* x, const float* gamma, const float* beta, void* y, int rows,
              int cols, float eps, int y_dtype, bool y_frag) {
    S2S_REQUIRE(cols % 4 == 0 && cols <= NORM_MAX_V4 * 256, "layernorm: cols=%d unsupported", cols);
    S2S_REQUIRE(!y_frag || (y_dtype == S2S_BF16 && cols % 32 == 0), "layernorm: fragment-major output needs bf16 and cols %% 32 == 0");
    if (rows == 0) return S2S_OK;
    if (y_dtype == S2S_F32)
        norm_rows_kernel<float, false><<<cdiv(rows, 4), 256, 0, st>>>(x, gamma, beta, (float*)y, rows, cols, eps, cols, nullptr, 0);
    else
        norm_rows_kernel<__hip_bfloat16, false><<<cdiv(rows, 4), 256, 0, st>>>(x, gamma, beta, (__hip_bfloat16*)y, rows, cols, eps, cols, nullptr,
                                                                               y_frag ? 1 : 0);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int rmsnorm_ex(hipStream_t st, const float* x, long x_rs, const float* gamma, void* y, float* y_f32, int rows, int cols, float eps,
               int y_dtype) {
    S2S_REQUIRE(cols % 4 == 0 && cols <= NORM_MAX_V4 * 256 && x_rs % 4 == 0, "rmsnorm: cols=%d unsupported", cols);
    if (rows == 0) return S2S_OK;
    if (y_dtype == S2S_F32)
        norm_rows_kernel<float, true><<<cdiv(rows, 4), 256, 0, st>>>(x, gamma, nullptr, (float*)y, rows, cols, eps, x_rs, y_f32, 0);
    else
        norm_rows_kernel<__hip_bfloat16, true><<<cdiv(rows, 4), 256, 0, st>>>(x, gamma, nullptr, (__hip_bfloat16*)y, rows, cols, eps, x_rs, y_f32, 0);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int rmsnorm(hipStream_t st, const float* x, const float* gamma, void* y, int rows, int cols, float eps,
            int y_dtype) {
    return rmsnorm_ex(st, x, cols, gamma, y, nullptr, rows, cols, eps, y_dtype);
}

}  // namespace s2s

extern "C" int32_t s2s_op_layernorm(void* stream, const float* x, const float* gamma, const float* beta,
                                    void* y, int32_t rows, int32_t cols, float eps, int32_t y_dtype) {
    S2S_REQUIRE(y_dtype == S2S_F32 || y_dtype == S2S_BF16, "bad y_dtype");
    return s2s::layernorm((hipStream_t)stream, x, gamma, beta, y, rows, cols, eps, y_dtype, false);
}
extern "C" int32_t s2s_op_rmsnorm(void* stream, const float* x, const float* gamma, void* y, int32_t rows,
                                  int32_t cols, float eps, int32_t y_dtype) {
    S2S_REQUIRE(y_dtype == S2S_F32 || y_dtype == S2S_BF16, "bad y_dtype");
    return s2s::rmsnorm((hipStream_t)stream, x, gamma, y, rows, cols, eps, y_dtype);
}
