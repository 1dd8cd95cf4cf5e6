// Log-mel STFT front-ends (Whisper + Parakeet) for gfx950.
//
// One fused kernel per 64-frame tile of one utterance:
//   PCM segment -> LDS (reflect / zero padding and pre-emphasis applied on the way in)
//   windowed frames x real-DFT basis on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32),
//   re^2+im^2 kept in registers, power tile -> LDS,
//   power x mel filter bank on the same MFMA, log, tile transposed through LDS and
//   written with full-line stores.
// The DFT is the direct O(N^2) form on purpose: 400 taps x (2 x 201|257) bins is a
// 64 x 400 x 416|544 GEMM per tile, which the f32 matrix pipe retires faster than an
// LDS-shuffled mixed-radix FFT would on 64-wide waves, and it keeps f32 accuracy at
// the k-ordered fmaf-chain level (the oracle is float64).
//
// Algorithm restated from (reference call sites in include/s2s_mi355x.h):
//   transformers/models/whisper/feature_extraction_whisper.py:105-131
//   transformers/models/parakeet/feature_extraction_parakeet.py:101-127,252-278
#include "common.h"

#include <math.h>

#include <algorithm>
#include <vector>

namespace s2s {

constexpr int FT = 64;         // frames per workgroup
constexpr int TAPS = 400;      // window length (both front-ends)
constexpr int HOP = 160;
constexpr int KSTEPS = TAPS / 4;
constexpr int RAW_N = (FT - 1) * HOP + TAPS;          // 10480 samples per tile
constexpr int RAW_LDS = RAW_N + RAW_N / HOP + 16;      // padded: idx + idx/160
constexpr int WHISPER_SAMPLES = 480000;
constexpr int WHISPER_FRAMES = 3000;

struct Frontend {
    int kind;
    int n_mels;
    int n_bins;
    int n_bin_tiles;   // ceil(n_bins/16)
    int n_mel_tiles;   // ceil(n_mels/16)
    int mel_ksteps;    // ceil(n_bins/4)
    int max_batch;
    int max_frames;
    float* basis;      // [KSTEPS][2*n_bin_tiles][64]   fragment-major cos|sin
    float* melw;       // [mel_ksteps][n_mel_tiles][64] fragment-major
    float* window;     // [TAPS]
    float* scratch;    // f32 log-mel before the global (whisper) / per-feature (parakeet) pass
    unsigned int* wmax;  // [max_batch] ordered-uint running max (whisper)
    float* stats;        // [max_batch][2][n_mels] mean, 1/(std+eps) (parakeet)
};

__device__ __forceinline__ int raw_idx(int s) { return s + s / HOP; }

// grid = (ceil(n_frames/FT), B), block = 256
template <int KIND>
__global__ __launch_bounds__(256) void logmel_tile_kernel(
    const float* __restrict__ pcm, long pcm_stride, const int* __restrict__ n_samples,
    const float* __restrict__ basis, const float* __restrict__ melw,
    const float* __restrict__ window, int n_bin_tiles, int n_mel_tiles, int mel_ksteps,
    int n_mels, int n_frames, float* __restrict__ out, unsigned int* __restrict__ wmax, int time_major,
    int window_samples, const float* __restrict__ norm, const int* __restrict__ first_frame = nullptr,
    const float* const* __restrict__ pcm_rows = nullptr, float* const* __restrict__ out_rows = nullptr) {
    // pcm_rows / out_rows (s2s_logmel_parakeet_cached_rows): item b reads its samples from pcm_rows[b] and writes its frames to out_rows[b]
    // ([frames][n_mels]) - the rows of a batch live in buffers of their own (one growing window each)
    // first_frame (s2s_logmel_parakeet_cached): frames below first_frame[b] are already in `out` from an earlier, shorter prefix of the same
    // signal - a frame only depends on the 400 samples around it, so a tile that ends below it has nothing to do
    if (first_frame && (int)blockIdx.x * FT + FT <= first_frame[blockIdx.y]) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* raw = smem;                       // RAW_LDS floats; later reused as the out stage
    float* win = smem + RAW_LDS;             // TAPS
    float* pw = win + TAPS;                  // [FT][pstride]
    const int pstride = n_bin_tiles * 16 + 1;

    const int b = blockIdx.y;
    const int t0 = blockIdx.x * FT;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l15 = lane & 15;
    const int lg = lane >> 4;

    const float* x = pcm_rows ? pcm_rows[b] : pcm + (long)b * pcm_stride;
    int n = n_samples[b];
    if (KIND == S2S_FRONTEND_WHISPER) n = min(n, window_samples);
    // zero-mean / unit-variance waveform normalisation (the Smart-Turn feature path): {mean, 1 / sqrt(var + 1e-7)} per item
    const float nmean = norm ? norm[2 * b] : 0.f, nrstd = norm ? norm[2 * b + 1] : 1.f;

    // ---- stage the PCM segment ------------------------------------------------
    const int s0 = t0 * HOP - TAPS / 2;
    for (int i = tid; i < RAW_N; i += 256) {
        int s = s0 + i;
        float v = 0.f;
        if (KIND == S2S_FRONTEND_WHISPER) {
            // zero-pad to the window (30 s; 8 s for Smart-Turn), then centre/reflect padding of the padded signal
            if (s < 0) s = -s;
            else if (s >= window_samples) s = 2 * (window_samples - 1) - s;
            if (s < n) v = norm ? __fmul_rn(__fsub_rn(x[s], nmean), nrstd) : x[s];
        } else {
            // pre-emphasis on the unpadded signal, zero ("constant") centre padding
            if (s >= 0 && s < n) {
                v = x[s];
                if (s >= 1) v = __fsub_rn(v, __fmul_rn(0.97f, x[s - 1]));
            }
        }
        raw[raw_idx(i)] = v;
    }
    for (int i = tid; i < TAPS; i += 256) win[i] = window[i];
    __syncthreads();

    // ---- windowed frames x DFT basis -> power ---------------------------------
    for (int p = wave; p < n_bin_tiles; p += 4) {
        f32x4_t are[4], aim[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            are[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            aim[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        const float* bre = basis + (long)p * 64 + lane;
        const float* bim = basis + (long)(n_bin_tiles + p) * 64 + lane;
        const long kstride = (long)2 * n_bin_tiles * 64;
        float cre[4], cim[4], nre[4], nim[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            cre[u] = bre[u * kstride];
            cim[u] = bim[u * kstride];
        }
        for (int ks = 0; ks < KSTEPS; ks += 4) {
            const bool more = ks + 4 < KSTEPS;
            if (more) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    nre[u] = bre[(long)(ks + 4 + u) * kstride];
                    nim[u] = bim[(long)(ks + 4 + u) * kstride];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int tap = (ks + u) * 4 + lg;
                const float w = win[tap];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int fr = m * 16 + l15;
                    const float a = raw[raw_idx(fr * HOP + tap)] * w;
                    are[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, cre[u], are[m], 0, 0, 0);
                    aim[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, cim[u], aim[m], 0, 0, 0);
                }
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    cre[u] = nre[u];
                    cim[u] = nim[u];
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int fr = m * 16 + lg * 4 + r;
                pw[fr * pstride + p * 16 + l15] = are[m][r] * are[m][r] + aim[m][r] * aim[m][r];
            }
    }
    __syncthreads();

    // ---- power x mel filters -> log ---------------------------------------------
    float* stage = raw;  // [FT][n_mels+1]
    const int sstride = n_mels + 1;
    float lmax = -INFINITY;
    for (int nt = wave; nt < n_mel_tiles; nt += 4) {
        f32x4_t acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const float* bw = melw + (long)nt * 64 + lane;
        const long kstride = (long)n_mel_tiles * 64;
        for (int ks = 0; ks < mel_ksteps; ++ks) {
            const float bv = bw[ks * kstride];
            const int k = ks * 4 + lg;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float a = pw[(m * 16 + l15) * pstride + k];
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[m], 0, 0, 0);
            }
        }
        const int mel = nt * 16 + l15;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int fr = m * 16 + lg * 4 + r;
                float v;
                if (KIND == S2S_FRONTEND_WHISPER) {
                    v = log10f(fmaxf(acc[m][r], 1e-10f));
                    if (mel < n_mels && t0 + fr < n_frames) lmax = fmaxf(lmax, v);
                } else {
                    v = logf(acc[m][r] + 5.9604644775390625e-08f);  // 2^-24
                }
                if (mel < n_mels) stage[fr * sstride + mel] = v;
            }
    }
    __syncthreads();

    if (KIND == S2S_FRONTEND_WHISPER) {
        lmax = wave_max(lmax);
        if (lane == 0) atomicMax(&wmax[b], f32_ordered(lmax));
    }
    if (KIND == S2S_FRONTEND_WHISPER && !time_major) {
        // out[b][mel][t]: each wave writes 64 consecutive frames of one mel row
        float* ob = out + (long)b * n_mels * n_frames;
        for (int mel = wave; mel < n_mels; mel += 4) {
            const int t = t0 + lane;
            if (t < n_frames) ob[(long)mel * n_frames + t] = stage[lane * sstride + mel];
        }
    } else {
        // out[b][t][mel]: each frame row is n_mels contiguous floats
        float* ob = out_rows ? out_rows[b] : out + (long)b * n_frames * n_mels;
        for (int i = tid; i < FT * n_mels; i += 256) {
            const int fr = i / n_mels, mel = i - fr * n_mels;
            if (t0 + fr < n_frames) ob[(long)(t0 + fr) * n_mels + mel] = stage[fr * sstride + mel];
        }
    }
}

// Whisper: x = (max(x, gmax-8)+4)/4, cast to the output type.  grid-stride.
template <typename OutT>
__global__ void whisper_finalize_kernel(const float* __restrict__ in, OutT* __restrict__ out,
                                        const unsigned int* __restrict__ wmax, long per_item,
                                        long total) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per_item);
        const float floor_v = f32_unordered(wmax[b]) - 8.0f;
        const float v = (fmaxf(in[i], floor_v) + 4.0f) * 0.25f;
        Elem<OutT>::st(out + i, v);
    }
}

// Whisper, encoder layout: out[b][r][mel], r = frame + 1, rows 0 and n_frames+1 are zero padding.
template <typename OutT>
__global__ void whisper_finalize_tm_kernel(const float* __restrict__ in, OutT* __restrict__ out,
                                           const unsigned int* __restrict__ wmax, int n_frames, int n_mels,
                                           long total) {
    const long per_item = (long)(n_frames + 2) * n_mels;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per_item);
        const long r = i - (long)b * per_item;
        const int row = (int)(r / n_mels);
        float v = 0.f;
        if (row >= 1 && row <= n_frames) {
            const float floor_v = f32_unordered(wmax[b]) - 8.0f;
            v = (fmaxf(in[(long)b * n_frames * n_mels + r - n_mels], floor_v) + 4.0f) * 0.25f;
        }
        Elem<OutT>::st(out + i, v);
    }
}

// Parakeet per-feature statistics over the valid frames: two-pass mean / unbiased std.
// grid = (n_mels/32, B), block = (32, 8)
__global__ void parakeet_stats_kernel(const float* __restrict__ x, const int* __restrict__ n_samples,
                                      int t_max, int n_mels, float* __restrict__ stats, const float* const* __restrict__ x_rows = nullptr) {
    __shared__ float red[8][33];
    const int b = blockIdx.y;
    const int mel = blockIdx.x * 32 + threadIdx.x;
    const int ty = threadIdx.y;
    const int T = min(n_samples[b] / HOP, t_max);
    const float* xb = x_rows ? x_rows[b] : x + (long)b * t_max * n_mels;
    float s = 0.f;
    if (mel < n_mels)
        for (int t = ty; t < T; t += 8) s += xb[(long)t * n_mels + mel];
    red[ty][threadIdx.x] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) tot += red[j][threadIdx.x];
    const float mean = T > 0 ? tot / (float)T : 0.f;
    __syncthreads();
    float q = 0.f;
    if (mel < n_mels)
        for (int t = ty; t < T; t += 8) {
            const float d = xb[(long)t * n_mels + mel] - mean;
            q += d * d;
        }
    red[ty][threadIdx.x] = q;
    __syncthreads();
    if (ty == 0 && mel < n_mels) {
        float qt = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) qt += red[j][threadIdx.x];
        const float var = T > 1 ? qt / (float)(T - 1) : 0.f;
        stats[((long)b * 2 + 0) * n_mels + mel] = mean;
        stats[((long)b * 2 + 1) * n_mels + mel] = 1.0f / (sqrtf(var) + 1e-5f);
    }
}

template <typename OutT>
__global__ void parakeet_normalize_kernel(const float* __restrict__ x, const int* __restrict__ n_samples,
                                          const float* __restrict__ stats, int t_max, int n_mels,
                                          OutT* __restrict__ out, int* __restrict__ n_frames_out,
                                          long total, const float* const* __restrict__ x_rows = nullptr) {
    const long per_item = (long)t_max * n_mels;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per_item);
        const long r = i - (long)b * per_item;
        const int t = (int)(r / n_mels);
        const int mel = (int)(r - (long)t * n_mels);
        const int T = min(n_samples[b] / HOP, t_max);
        float v = 0.f;
        if (t < T) {
            const float mean = stats[((long)b * 2 + 0) * n_mels + mel];
            const float inv = stats[((long)b * 2 + 1) * n_mels + mel];
            v = ((x_rows ? x_rows[b][r] : x[i]) - mean) * inv;
        }
        Elem<OutT>::st(out + i, v);
        if (r == 0 && n_frames_out) n_frames_out[b] = T;
    }
}

// Waveform statistics of the first min(n_samples, window) samples of every item: {mean, 1 / sqrt(var + 1e-7)} (population variance),
// what WhisperFeatureExtractor(do_normalize=True) applies before the STFT (feature_extraction_whisper.py zero_mean_unit_var_norm;
// reference VAD/smart_turn.py:135-143).  One workgroup per item, float64 accumulation, two passes.
__global__ __launch_bounds__(256) void waveform_stats_kernel(const float* __restrict__ pcm, long pcm_stride, const int* __restrict__ n_samples,
                                                             int window_samples, float* __restrict__ norm) {
    __shared__ double red[256];
    __shared__ double mean_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x = pcm + (long)b * pcm_stride;
    const int n = min(n_samples[b], window_samples);
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += (double)x[i];
    red[tid] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) mean_s = n > 0 ? red[0] / (double)n : 0.0;
    __syncthreads();
    const double mean = mean_s;
    double q = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double d = (double)x[i] - mean;
        q += d * d;
    }
    __syncthreads();
    red[tid] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) {
        const double var = n > 0 ? red[0] / (double)n : 0.0;
        norm[2 * b] = (float)mean;
        norm[2 * b + 1] = (float)(1.0 / sqrt(var + 1e-7));
    }
}

static size_t tile_lds_bytes(const Frontend* f) {
    const size_t pstride = (size_t)f->n_bin_tiles * 16 + 1;
    size_t stage = (size_t)FT * (f->n_mels + 1);
    size_t rawn = RAW_LDS > stage ? RAW_LDS : stage;
    (void)rawn;
    return sizeof(float) * ((size_t)RAW_LDS + TAPS + FT * pstride);
}

}  // namespace s2s

using namespace s2s;

extern "C" int32_t s2s_frontend_create(int32_t kind, int32_t n_mels, const float* mel_filters_host,
                                       int32_t max_batch, void** out_frontend) {
    std::lock_guard<std::mutex> no_capture(s2s::capture_mutex());   // synchronous uploads below: never during a stream capture
    S2S_REQUIRE(kind == S2S_FRONTEND_WHISPER || kind == S2S_FRONTEND_PARAKEET, "bad frontend kind %d", kind);
    S2S_REQUIRE(n_mels > 0 && n_mels <= 128, "n_mels %d out of range (1..128)", n_mels);
    S2S_REQUIRE(mel_filters_host && out_frontend && max_batch > 0, "null argument");
    Frontend* f = new Frontend();
    f->kind = kind;
    f->n_mels = n_mels;
    const int nfft = kind == S2S_FRONTEND_WHISPER ? 400 : 512;
    f->n_bins = nfft / 2 + 1;
    f->n_bin_tiles = cdiv(f->n_bins, 16);
    f->n_mel_tiles = cdiv(n_mels, 16);
    f->mel_ksteps = cdiv(f->n_bins, 4);
    f->max_batch = max_batch;
    f->max_frames = kind == S2S_FRONTEND_WHISPER ? WHISPER_FRAMES : 6001;  // Parakeet: up to 60 s

    // DFT basis, fragment-major for the 16x16x4 B operand: B[k = lane>>4][j = lane&15]
    std::vector<float> basis((size_t)KSTEPS * 2 * f->n_bin_tiles * 64, 0.f);
    for (int ks = 0; ks < KSTEPS; ++ks)
        for (int tile = 0; tile < 2 * f->n_bin_tiles; ++tile)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = ks * 4 + (lane >> 4);
                const bool is_sin = tile >= f->n_bin_tiles;
                const int k = (is_sin ? tile - f->n_bin_tiles : tile) * 16 + (lane & 15);
                float v = 0.f;
                if (k < f->n_bins) {
                    const long ph = ((long)k * m) % nfft;
                    const double ang = 2.0 * M_PI * (double)ph / (double)nfft;
                    v = (float)(is_sin ? sin(ang) : cos(ang));
                }
                basis[((size_t)ks * 2 * f->n_bin_tiles + tile) * 64 + lane] = v;
            }
    std::vector<float> melw((size_t)f->mel_ksteps * f->n_mel_tiles * 64, 0.f);
    for (int ks = 0; ks < f->mel_ksteps; ++ks)
        for (int nt = 0; nt < f->n_mel_tiles; ++nt)
            for (int lane = 0; lane < 64; ++lane) {
                const int bin = ks * 4 + (lane >> 4);
                const int mel = nt * 16 + (lane & 15);
                float v = 0.f;
                if (bin < f->n_bins && mel < n_mels) v = mel_filters_host[(size_t)bin * n_mels + mel];
                melw[((size_t)ks * f->n_mel_tiles + nt) * 64 + lane] = v;
            }
    std::vector<float> win(TAPS);
    for (int i = 0; i < TAPS; ++i) {
        // Whisper: periodic Hann(400); Parakeet: symmetric Hann(400) centred in n_fft=512
        const double den = kind == S2S_FRONTEND_WHISPER ? 400.0 : 399.0;
        win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)i / den));
    }
    S2S_HIP_CHECK(hipMalloc(&f->basis, basis.size() * sizeof(float)));
    S2S_HIP_CHECK(hipMalloc(&f->melw, melw.size() * sizeof(float)));
    S2S_HIP_CHECK(hipMalloc(&f->window, TAPS * sizeof(float)));
    S2S_HIP_CHECK(hipMalloc(&f->scratch, (size_t)max_batch * f->max_frames * n_mels * sizeof(float)));
    S2S_HIP_CHECK(hipMalloc(&f->wmax, (size_t)max_batch * sizeof(unsigned int)));
    S2S_HIP_CHECK(hipMalloc(&f->stats, (size_t)max_batch * 2 * n_mels * sizeof(float)));
    S2S_HIP_CHECK(hipMemcpy(f->basis, basis.data(), basis.size() * sizeof(float), hipMemcpyHostToDevice));
    S2S_HIP_CHECK(hipMemcpy(f->melw, melw.data(), melw.size() * sizeof(float), hipMemcpyHostToDevice));
    S2S_HIP_CHECK(hipMemcpy(f->window, win.data(), TAPS * sizeof(float), hipMemcpyHostToDevice));
    const size_t lds = tile_lds_bytes(f);
    if (kind == S2S_FRONTEND_WHISPER)
        S2S_HIP_CHECK(hipFuncSetAttribute((const void*)logmel_tile_kernel<S2S_FRONTEND_WHISPER>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    else
        S2S_HIP_CHECK(hipFuncSetAttribute((const void*)logmel_tile_kernel<S2S_FRONTEND_PARAKEET>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    *out_frontend = f;
    return S2S_OK;
}

extern "C" int32_t s2s_frontend_destroy(void* frontend) {
    if (!frontend) return S2S_OK;
    Frontend* f = (Frontend*)frontend;
    hipFree(f->basis);
    hipFree(f->melw);
    hipFree(f->window);
    hipFree(f->scratch);
    hipFree(f->wmax);
    hipFree(f->stats);
    delete f;
    return S2S_OK;
}

extern "C" int32_t s2s_logmel_whisper(void* frontend, void* stream, const float* pcm_dev,
                                      int64_t pcm_stride, const int32_t* n_samples_dev, int32_t B,
                                      void* out_dev, int32_t out_dtype) {
    Frontend* f = (Frontend*)frontend;
    S2S_REQUIRE(f && f->kind == S2S_FRONTEND_WHISPER, "not a Whisper front-end");
    S2S_REQUIRE(B >= 0 && B <= f->max_batch, "batch %d exceeds max_batch %d", B, f->max_batch);
    S2S_REQUIRE(out_dtype == S2S_F32 || out_dtype == S2S_BF16, "bad out_dtype %d", out_dtype);
    if (B == 0) return S2S_OK;
    S2S_REQUIRE(pcm_dev && n_samples_dev && out_dev, "null device pointer");
    hipStream_t st = (hipStream_t)stream;
    S2S_HIP_CHECK(hipMemsetAsync(f->wmax, 0, (size_t)B * sizeof(unsigned int), st));
    dim3 grid(cdiv(WHISPER_FRAMES, FT), B);
    logmel_tile_kernel<S2S_FRONTEND_WHISPER><<<grid, 256, tile_lds_bytes(f), st>>>(
        pcm_dev, (long)pcm_stride, n_samples_dev, f->basis, f->melw, f->window, f->n_bin_tiles,
        f->n_mel_tiles, f->mel_ksteps, f->n_mels, WHISPER_FRAMES, f->scratch, f->wmax, 0, WHISPER_SAMPLES, nullptr);
    S2S_LAUNCH_CHECK();
    const long per_item = (long)f->n_mels * WHISPER_FRAMES;
    const long total = per_item * B;
    const int blocks = (int)std::min<long>(cdiv(total, 256), 2048);
    if (out_dtype == S2S_F32)
        whisper_finalize_kernel<float><<<blocks, 256, 0, st>>>(f->scratch, (float*)out_dev, f->wmax, per_item, total);
    else
        whisper_finalize_kernel<__hip_bfloat16><<<blocks, 256, 0, st>>>(f->scratch, (__hip_bfloat16*)out_dev, f->wmax, per_item, total);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

/* Whisper log-mel of a SHORTER window with optional waveform normalisation - the Smart-Turn end-of-turn feature path
 * (reference VAD/smart_turn.py:130-143: WhisperFeatureExtractor(chunk_length=8), do_normalize=True, padding to 8 s).
 * out [B][n_mels][window_samples / 160]; window_samples a multiple of 160, <= 480000. */
extern "C" int32_t s2s_logmel_whisper_window(void* frontend, void* stream, const float* pcm_dev, int64_t pcm_stride,
                                             const int32_t* n_samples_dev, int32_t B, int32_t window_samples, int32_t normalize,
                                             void* out_dev, int32_t out_dtype) {
    Frontend* f = (Frontend*)frontend;
    S2S_REQUIRE(f && f->kind == S2S_FRONTEND_WHISPER, "not a Whisper front-end");
    S2S_REQUIRE(B >= 0 && B <= f->max_batch, "batch %d exceeds max_batch %d", B, f->max_batch);
    S2S_REQUIRE(window_samples >= TAPS && window_samples <= WHISPER_SAMPLES && window_samples % HOP == 0,
                "window of %d samples unsupported (a multiple of 160 in [400, 480000])", window_samples);
    S2S_REQUIRE(out_dtype == S2S_F32 || out_dtype == S2S_BF16, "bad out_dtype %d", out_dtype);
    if (B == 0) return S2S_OK;
    S2S_REQUIRE(pcm_dev && n_samples_dev && out_dev, "null device pointer");
    hipStream_t st = (hipStream_t)stream;
    const int n_frames = window_samples / HOP;
    S2S_HIP_CHECK(hipMemsetAsync(f->wmax, 0, (size_t)B * sizeof(unsigned int), st));
    float* norm = nullptr;
    if (normalize) {
        norm = f->stats;          // [max_batch][2 * n_mels] floats: the first 2 * B hold {mean, rstd}
        waveform_stats_kernel<<<B, 256, 0, st>>>(pcm_dev, (long)pcm_stride, n_samples_dev, window_samples, norm);
        S2S_LAUNCH_CHECK();
    }
    dim3 grid(cdiv(n_frames, FT), B);
    logmel_tile_kernel<S2S_FRONTEND_WHISPER><<<grid, 256, tile_lds_bytes(f), st>>>(
        pcm_dev, (long)pcm_stride, n_samples_dev, f->basis, f->melw, f->window, f->n_bin_tiles,
        f->n_mel_tiles, f->mel_ksteps, f->n_mels, n_frames, f->scratch, f->wmax, 0, window_samples, norm);
    S2S_LAUNCH_CHECK();
    const long per_item = (long)f->n_mels * n_frames;
    const long total = per_item * B;
    const int blocks = (int)std::min<long>(cdiv(total, 256), 2048);
    if (out_dtype == S2S_F32)
        whisper_finalize_kernel<float><<<blocks, 256, 0, st>>>(f->scratch, (float*)out_dev, f->wmax, per_item, total);
    else
        whisper_finalize_kernel<__hip_bfloat16><<<blocks, 256, 0, st>>>(f->scratch, (__hip_bfloat16*)out_dev, f->wmax, per_item, total);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

extern "C" int32_t s2s_logmel_whisper_tm(void* frontend, void* stream, const float* pcm_dev,
                                         int64_t pcm_stride, const int32_t* n_samples_dev, int32_t B,
                                         void* out_dev, int32_t out_dtype) {
    Frontend* f = (Frontend*)frontend;
    S2S_REQUIRE(f && f->kind == S2S_FRONTEND_WHISPER, "not a Whisper front-end");
    S2S_REQUIRE(B >= 0 && B <= f->max_batch, "batch %d exceeds max_batch %d", B, f->max_batch);
    S2S_REQUIRE(out_dtype == S2S_F32 || out_dtype == S2S_BF16, "bad out_dtype %d", out_dtype);
    if (B == 0) return S2S_OK;
    S2S_REQUIRE(pcm_dev && n_samples_dev && out_dev, "null device pointer");
    hipStream_t st = (hipStream_t)stream;
    S2S_HIP_CHECK(hipMemsetAsync(f->wmax, 0, (size_t)B * sizeof(unsigned int), st));
    dim3 grid(cdiv(WHISPER_FRAMES, FT), B);
    logmel_tile_kernel<S2S_FRONTEND_WHISPER><<<grid, 256, tile_lds_bytes(f), st>>>(
        pcm_dev, (long)pcm_stride, n_samples_dev, f->basis, f->melw, f->window, f->n_bin_tiles,
        f->n_mel_tiles, f->mel_ksteps, f->n_mels, WHISPER_FRAMES, f->scratch, f->wmax, 1, WHISPER_SAMPLES, nullptr);
    S2S_LAUNCH_CHECK();
    const long total = (long)B * (WHISPER_FRAMES + 2) * f->n_mels;
    const int blocks = (int)std::min<long>(cdiv(total, 256), 2048);
    if (out_dtype == S2S_F32)
        whisper_finalize_tm_kernel<float><<<blocks, 256, 0, st>>>(f->scratch, (float*)out_dev, f->wmax, WHISPER_FRAMES, f->n_mels, total);
    else
        whisper_finalize_tm_kernel<__hip_bfloat16><<<blocks, 256, 0, st>>>(f->scratch, (__hip_bfloat16*)out_dev, f->wmax, WHISPER_FRAMES, f->n_mels, total);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

extern "C" int32_t s2s_logmel_parakeet(void* frontend, void* stream, const float* pcm_dev,
                                       int64_t pcm_stride, const int32_t* n_samples_dev, int32_t B,
                                       int32_t t_max, void* out_dev, int32_t out_dtype,
                                       int32_t* n_frames_dev) {
    Frontend* f = (Frontend*)frontend;
    S2S_REQUIRE(f && f->kind == S2S_FRONTEND_PARAKEET, "not a Parakeet front-end");
    S2S_REQUIRE(B >= 0 && B <= f->max_batch, "batch %d exceeds max_batch %d", B, f->max_batch);
    S2S_REQUIRE(t_max > 0 && t_max <= f->max_frames, "t_max %d out of range (1..%d)", t_max, f->max_frames);
    S2S_REQUIRE(out_dtype == S2S_F32 || out_dtype == S2S_BF16, "bad out_dtype %d", out_dtype);
    if (B == 0) return S2S_OK;
    S2S_REQUIRE(pcm_dev && n_samples_dev && out_dev, "null device pointer");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(cdiv(t_max, FT), B);
    logmel_tile_kernel<S2S_FRONTEND_PARAKEET><<<grid, 256, tile_lds_bytes(f), st>>>(
        pcm_dev, (long)pcm_stride, n_samples_dev, f->basis, f->melw, f->window, f->n_bin_tiles,
        f->n_mel_tiles, f->mel_ksteps, f->n_mels, t_max, f->scratch, nullptr, 1, 0, nullptr);
    S2S_LAUNCH_CHECK();
    parakeet_stats_kernel<<<dim3(cdiv(f->n_mels, 32), B), dim3(32, 8), 0, st>>>(
        f->scratch, n_samples_dev, t_max, f->n_mels, f->stats);
    S2S_LAUNCH_CHECK();
    const long total = (long)B * t_max * f->n_mels;
    const int blocks = (int)std::min<long>(cdiv(total, 256), 2048);
    if (out_dtype == S2S_F32)
        parakeet_normalize_kernel<float><<<blocks, 256, 0, st>>>(
            f->scratch, n_samples_dev, f->stats, t_max, f->n_mels, (float*)out_dev, n_frames_dev, total);
    else
        parakeet_normalize_kernel<__hip_bfloat16><<<blocks, 256, 0, st>>>(
            f->scratch, n_samples_dev, f->stats, t_max, f->n_mels, (__hip_bfloat16*)out_dev, n_frames_dev, total);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// new samples of a batch of growing windows, uploaded as ONE flat block: row b gets new_pcm[new_off[b] .. new_off[b + 1]) appended at old_n[b]
__global__ void pcm_append_rows_kernel(float* const* __restrict__ pcm_rows, const float* __restrict__ new_pcm, const int* __restrict__ new_off,
                                       const int* __restrict__ old_n) {
    const int b = blockIdx.y;
    const int o0 = new_off[b], cnt = new_off[b + 1] - o0;
    float* dst = pcm_rows[b] + old_n[b];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) dst[i] = new_pcm[o0 + i];
}

/* The live partials of a TICK in one call (round 6): B growing windows, each with a pcm row and a log-mel cache of its own (device pointer
 * tables), their new samples in one flat block.  Same kernels and per-window results as B calls of s2s_logmel_parakeet_cached - the per-window
 * driver work (three launches, three small uploads) was 0.24 ms x the batch, 43 % of a 48-window pass (profiles/r06_bench_partial_pass.txt). */
extern "C" int32_t s2s_logmel_parakeet_cached_rows(void* frontend, void* stream, int32_t B, int32_t t_max, float* const* pcm_rows_dev,
                                                   float* const* cache_rows_dev, const int32_t* n_samples_dev, const int32_t* first_frame_dev,
                                                   const float* new_pcm_dev, const int32_t* new_off_dev, const int32_t* old_n_dev, void* out_dev,
                                                   int32_t out_dtype, int32_t* n_frames_dev) {
    Frontend* f = (Frontend*)frontend;
    S2S_REQUIRE(f && f->kind == S2S_FRONTEND_PARAKEET, "not a Parakeet front-end");
    S2S_REQUIRE(B >= 0 && B <= f->max_batch, "batch %d exceeds max_batch %d", B, f->max_batch);
    S2S_REQUIRE(t_max > 0 && t_max <= f->max_frames, "t_max %d out of range (1..%d)", t_max, f->max_frames);
    S2S_REQUIRE(out_dtype == S2S_F32 || out_dtype == S2S_BF16, "bad out_dtype %d", out_dtype);
    if (B == 0) return S2S_OK;
    S2S_REQUIRE(pcm_rows_dev && cache_rows_dev && n_samples_dev && first_frame_dev && out_dev, "null device pointer");
    S2S_REQUIRE(!new_pcm_dev || (new_off_dev && old_n_dev), "new samples without their offsets");
    hipStream_t st = (hipStream_t)stream;
    if (new_pcm_dev) {
        pcm_append_rows_kernel<<<dim3(8, B), 256, 0, st>>>(pcm_rows_dev, new_pcm_dev, new_off_dev, old_n_dev);
        S2S_LAUNCH_CHECK();
    }
    dim3 grid(cdiv(t_max, FT), B);
    logmel_tile_kernel<S2S_FRONTEND_PARAKEET><<<grid, 256, tile_lds_bytes(f), st>>>(
        nullptr, 0, n_samples_dev, f->basis, f->melw, f->window, f->n_bin_tiles, f->n_mel_tiles, f->mel_ksteps, f->n_mels, t_max, nullptr, nullptr, 1, 0,
        nullptr, first_frame_dev, (const float* const*)pcm_rows_dev, cache_rows_dev);
    S2S_LAUNCH_CHECK();
    parakeet_stats_kernel<<<dim3(cdiv(f->n_mels, 32), B), dim3(32, 8), 0, st>>>(nullptr, n_samples_dev, t_max, f->n_mels, f->stats, (const float* const*)cache_rows_dev);
    S2S_LAUNCH_CHECK();
    const long total = (long)B * t_max * f->n_mels;
    const int blocks = (int)std::min<long>(cdiv(total, 256), 2048);
    if (out_dtype == S2S_F32)
        parakeet_normalize_kernel<float><<<blocks, 256, 0, st>>>(nullptr, n_samples_dev, f->stats, t_max, f->n_mels, (float*)out_dev, n_frames_dev, total,
                                                                 (const float* const*)cache_rows_dev);
    else
        parakeet_normalize_kernel<__hip_bfloat16><<<blocks, 256, 0, st>>>(nullptr, n_samples_dev, f->stats, t_max, f->n_mels, (__hip_bfloat16*)out_dev,
                                                                        n_frames_dev, total, (const float* const*)cache_rows_dev);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

/* Live partials (reference STT/smart_progressive_streaming.py:87-157 re-transcribes a GROWING window): the un-normalised log-mel frames of the
 * prefix are kept in a caller-owned cache and only the frames the new samples can change are computed. */
extern "C" int32_t s2s_logmel_parakeet_cached(void* frontend, void* stream, const float* pcm_dev, int64_t pcm_stride,
                                              const int32_t* n_samples_dev, const int32_t* first_frame_dev, int32_t B, int32_t t_max,
                                              float* cache_dev, void* out_dev, int32_t out_dtype, int32_t* n_frames_dev) {
    Frontend* f = (Frontend*)frontend;
    S2S_REQUIRE(f && f->kind == S2S_FRONTEND_PARAKEET, "not a Parakeet front-end");
    S2S_REQUIRE(B >= 0 && B <= f->max_batch, "batch %d exceeds max_batch %d", B, f->max_batch);
    S2S_REQUIRE(t_max > 0 && t_max <= f->max_frames, "t_max %d out of range (1..%d)", t_max, f->max_frames);
    S2S_REQUIRE(out_dtype == S2S_F32 || out_dtype == S2S_BF16, "bad out_dtype %d", out_dtype);
    if (B == 0) return S2S_OK;
    S2S_REQUIRE(pcm_dev && n_samples_dev && first_frame_dev && cache_dev && out_dev, "null device pointer");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(cdiv(t_max, FT), B);
    logmel_tile_kernel<S2S_FRONTEND_PARAKEET><<<grid, 256, tile_lds_bytes(f), st>>>(
        pcm_dev, (long)pcm_stride, n_samples_dev, f->basis, f->melw, f->window, f->n_bin_tiles,
        f->n_mel_tiles, f->mel_ksteps, f->n_mels, t_max, cache_dev, nullptr, 1, 0, nullptr, first_frame_dev);
    S2S_LAUNCH_CHECK();
    parakeet_stats_kernel<<<dim3(cdiv(f->n_mels, 32), B), dim3(32, 8), 0, st>>>(cache_dev, n_samples_dev, t_max, f->n_mels, f->stats);
    S2S_LAUNCH_CHECK();
    const long total = (long)B * t_max * f->n_mels;
    const int blocks = (int)std::min<long>(cdiv(total, 256), 2048);
    if (out_dtype == S2S_F32)
        parakeet_normalize_kernel<float><<<blocks, 256, 0, st>>>(cache_dev, n_samples_dev, f->stats, t_max, f->n_mels, (float*)out_dev, n_frames_dev, total);
    else
        parakeet_normalize_kernel<__hip_bfloat16><<<blocks, 256, 0, st>>>(cache_dev, n_samples_dev, f->stats, t_max, f->n_mels, (__hip_bfloat16*)out_dev,
                                                                        n_frames_dev, total);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}
