// Dense projection GEMMs for gfx950:  C[M,N] = act(A[M,K] . W[N,K]^T + bias) (+ residual)
//
// W is kept in the checkpoint's own [out_features][in_features] layout, so both MFMA
// operands are read along K ("B^T input"): no transposes anywhere on the path.
//
//  * gemm_tile_kernel   : 128x128 output tile per 256-thread workgroup (2x2 waves, each
//                         a 4x4 grid of 16x16 MFMA fragments), K-slabs land in LDS by
//                         direct global->LDS loads (global_load_lds_dwordx4, no VGPR round
//                         trip) into two stages: slab t+1 is in flight while slab t is
//                         multiplied, one barrier per slab.  128-byte LDS rows with a
//                         16-byte-chunk XOR swizzle (applied on the source address) keep the
//                         ds_read_b128 fragment reads conflict-free.  Workgroup ids are
//                         remapped so each XCD's L2 sees a contiguous band of tiles sharing W.
//  * gemm_skinny_kernel : M <= 64 (decode steps: one row per live session).  Weights are
//                         streamed exactly once straight from HBM into MFMA B-fragments
//                         (no LDS staging), activations come from L2; one 8-wave workgroup
//                         per 16-column slab, the waves split K and combine through LDS
//                         (deterministic), full epilogue on the combined tile.
//
// Element types: bf16 inputs -> v_mfma_f32_16x16x32_bf16, f32 inputs ->
// v_mfma_f32_16x16x4_f32 (exact f32, used by the parity policy).  Accumulation is f32.
#include <type_traits>
#include "ops.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <string>

namespace s2s {

template <typename T> struct MfmaTraits;
template <> struct MfmaTraits<__hip_bfloat16> {
    static constexpr int KPF = 32;   // K per MFMA
    static constexpr int EPC = 8;    // elements per 16-byte chunk
};
template <> struct MfmaTraits<float> {
    static constexpr int KPF = 4;
    static constexpr int EPC = 4;
};

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7): one v_rcp_f32, one v_exp_f32 and five fmas instead of the ~35
// instructions of erff.  Used where the result is rounded to bf16 anyway (resolution 4e-3): the GELU epilogue of the encoder's
// fc1 GEMM runs 64 elements per thread with nothing else resident on the CU, so its cost is fully exposed.
__device__ __forceinline__ float gelu_erf_bf16(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float e = pl * t * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);      // 1 - erf(z)
    const float erf_abs = 1.0f - e;
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// GELU for a bf16 destination without transcendentals (quarter rate on the vector unit) and two elements per instruction:
// gelu(x) = max(x, -4) * (0.5 + xc * Q(xc^2)), xc = clamp(x, -4, 4), Q = degree-8 polynomial fitted to 0.5 erf(x / sqrt 2) / x on [0, 4]
// (tools/fit_gelu_poly.py).  Error against the exact erf form, float32 evaluation: <= 5.4e-5 inside [-4, 4]; for x < -4 the factor stays
// at its x = -4 value (Phi(-4) = 3.2e-5, it does not decay) so the multiplicand is clamped at -4 as well: the result is the constant
// -7.9e-5 where gelu is < 1.3e-4 and -> 0, i.e. |error| <= 7.9e-5 for EVERY x < -4 (with x unclamped it grew as 2e-5 |x|: ADVICE r5);
// for x > 4 the result is x (1 - 2e-5), below 2^-15 relative.  All of it is below the bf16 rounding of the result (2^-9 relative) wherever the
// activation is not negligible; tests/test_ops_gpu.py::test_gelu_poly_matches_erf_form_over_a_wide_range pins it on [-1e4, 1e4].
// 11 packed instructions + 4 clamps per PAIR against ~16 scalar ones (2 of them v_rcp / v_exp) per element for gelu_erf_bf16: the fc1
// epilogue of the Whisper encoder ran 13 us per 256x256 tile against 4.5 us for the same tile without activation (tools/tile5_phases.py).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_poly2(f32x2_t x) {
    f32x2_t xc, xm;
    xc[0] = __builtin_amdgcn_fmed3f(x[0], -4.0f, 4.0f);
    xc[1] = __builtin_amdgcn_fmed3f(x[1], -4.0f, 4.0f);
    xm[0] = fmaxf(x[0], -4.0f);
    xm[1] = fmaxf(x[1], -4.0f);
    const f32x2_t t = xc * xc;
    f32x2_t q = t * 8.524421446498565e-11f + -7.295211545965685e-09f;
    q = q * t + 2.791634869936388e-07f;
    q = q * t + -6.397717697836924e-06f;
    q = q * t + 9.969574603019282e-05f;
    q = q * t + -0.0011373028391972184f;
    q = q * t + 0.009885039180517197f;
    q = q * t + -0.06641801446676254f;
    q = q * t + 0.3989247679710388f;
    const f32x2_t phi = xc * q + 0.5f;
    return xm * phi;
}

template <typename CT, int ACT>
__device__ __forceinline__ float apply_act_c(float v) {
    if constexpr (ACT == S2S_ACT_GELU) return sizeof(CT) == 2 ? gelu_erf_bf16(v) : gelu_erf(v);
    else if constexpr (ACT == S2S_ACT_SILU) return v / (1.0f + expf(-v));
    else if constexpr (ACT == S2S_ACT_RELU) return fmaxf(v, 0.f);
    else return v;
}
// the activation switch taken ONCE around an epilogue's staging loops instead of once per element (f is called with the activation as a
// compile-time constant; S2S_ACT_SNAKE is applied later, on the staged rows, and counts as none here)
template <typename F>
__device__ __forceinline__ void with_act(int act, F&& f) {
    switch (act) {
        case S2S_ACT_GELU: f(std::integral_constant<int, S2S_ACT_GELU>{}); break;
        case S2S_ACT_SILU: f(std::integral_constant<int, S2S_ACT_SILU>{}); break;
        case S2S_ACT_RELU: f(std::integral_constant<int, S2S_ACT_RELU>{}); break;
        default: f(std::integral_constant<int, S2S_ACT_NONE>{}); break;
    }
}
// LDS staging inside ONE wave: its LDS queue is in order, so a compiler barrier is all the write -> read (and read -> rewrite) hand-over
// needs; the wavefront-scope fences used here until round 5 also made every pass wait for its global stores (s_waitcnt vmcnt(0)).
#define S2S_WAVE_LDS_HANDOVER()                 \
    do {                                        \
        asm volatile("" ::: "memory");          \
        __builtin_amdgcn_wave_barrier();        \
        asm volatile("" ::: "memory");          \
    } while (0)

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case S2S_ACT_GELU: return gelu_erf(v);
        case S2S_ACT_SILU: return v / (1.0f + expf(-v));
        case S2S_ACT_RELU: return fmaxf(v, 0.f);
        default: return v;
    }
}

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;             // bytes of K per LDS row (64 bf16 / 32 f32)
constexpr int TILE_BYTES = BM * ROWB; // 16 KiB per operand

// XCD-aware remap: consecutive logical tiles (sharing a W band) land on one XCD.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Tile order inside an XCD's contiguous range of logical tile ids: groups of GROUP_N tile columns, and inside a group the
// column index runs fastest.  The ~32 tiles an XCD has in flight then form a (4 row-tiles x 8 column-tiles) block that walks
// K together, so every A slab is fetched into that L2 once and shared by 8 tiles and every W slab by 4 - instead of 32
// different A bands against one W band (M-fastest order), which re-fetched A from the fabric once per tile column
// (PMC: 6.9 x the algorithmic bytes on the encoder GEMMs).
constexpr int GROUP_N = 8;
__device__ __forceinline__ void tile_coords(int wg, int tiles_m, int tiles_n, int& tm, int& tn) {
    const int per_group = GROUP_N * tiles_m;
    const int group = wg / per_group, in_group = wg - group * per_group;
    const int first_n = group * GROUP_N;
    const int gsz = min(tiles_n - first_n, GROUP_N);
    tm = in_group / gsz;
    tn = first_n + (in_group - tm * gsz);
}

// Row r of A / C / residual lives at base + (r / G) * batch_stride + (r % G) * ld, so that
// per-utterance zero-padded time-major buffers can be addressed as overlapping im2col rows
// (1-D convolutions become GEMMs without materialising im2col) and positional tables can be
// broadcast over the batch (r_bs = 0).
struct RowMap {
    long a_bs, c_bs, r_bs, ldr;
    int G;
    long ldw;                       // row stride of W (elements)
    int nz2;                        // blockIdx.y = z1 * nz2 + z2: batched GEMMs (e.g. one per (session, head))
    long a_z1, a_z2, w_z1, w_z2, c_z1, c_z2;
    float alpha;                    // C = act(acc + bias) * alpha + residual
    int tap_len;                    // > 0: K index k of A lives at row + (k / tap_len) * tap_stride + k % tap_len
    long tap_stride;                //      (dilated 1-D convolution taps over a time-major buffer)
    // epilogue extensions (GemmArgs): SnakeBeta on the primary output, model-dtype residual, second snake'd output
    const float* sn_a; const float* sn_ib; int sn_mod;
    const void* res_c;
    void* C2; const float* sn2_a; const float* sn2_ib;
};

// SnakeBeta y = x + sin^2(a x) / b.  The bf16 pipeline uses the hardware sine (abs error ~1e-6 on the |a x| < 256 range
// the codec produces, far below bf16 resolution); f32 keeps sinf so the f32 path matches the oracle to rounding.
template <typename CT>
__device__ __forceinline__ float snake_f(float v, float a, float ib) {
    float s;
    if constexpr (sizeof(CT) == 2) s = __sinf(v * a);
    else s = sinf(v * a);
    return v + s * s * ib;
}
template <typename CT>
__device__ __forceinline__ float round_to(float v) {
    if constexpr (sizeof(CT) == 2) return bf16_bits_to_f32(f32_to_bf16_bits(v));
    else return v;
}

__device__ __attribute__((aligned(16))) unsigned char g_zero_page[64];   // source of out-of-range chunks

template <typename T, typename CT>
__global__ __launch_bounds__(256) void gemm_tile_kernel(
    const T* __restrict__ A, long lda, const T* __restrict__ W, const float* __restrict__ bias,
    const float* residual, CT* C, long ldc, int M, int N, int K, int act, int tiles_m, RowMap rm) {
    constexpr int EPC = MfmaTraits<T>::EPC;
    constexpr int BK = ROWB / (int)sizeof(T);
    // two stages x (A tile | W tile); filled by direct global->LDS loads (global_load_lds_dwordx4: the wave's 64 lanes
    // land as one linear 1 KiB piece), so the XOR swizzle is applied to the per-lane SOURCE address: LDS chunk
    // (row, p) holds logical chunk p ^ (row & 7) of that row, and the fragment reads use the same involution.
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2 * TILE_BYTES];

    {
        const int z1 = blockIdx.y / rm.nz2, z2 = blockIdx.y % rm.nz2;
        A += z1 * rm.a_z1 + z2 * rm.a_z2;
        W += z1 * rm.w_z1 + z2 * rm.w_z2;
        C += z1 * rm.c_z1 + z2 * rm.c_z2;
        if (residual) residual += z1 * rm.c_z1 + z2 * rm.c_z2;
    }
    const int nwg = gridDim.x;
    const int wg = xcd_remap(blockIdx.x, nwg);
    int tm, tn;
    tile_coords(wg, tiles_m, nwg / tiles_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // staging: 1024 16-byte chunks per operand tile, 4 per thread; chunk c -> (row c>>3, physical slot c&7)
    const T* zero = reinterpret_cast<const T*>(g_zero_page);
    const T* arow[4];
    const T* wrow[4];
    int lch[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * 256;
        const int row = c >> 3;
        lch[i] = (c & 7) ^ (row & 7);
        const int m = m0 + row, n = n0 + row;
        arow[i] = nullptr;
        if (m < M) arow[i] = rm.G < M ? A + (long)(m / rm.G) * rm.a_bs + (long)(m % rm.G) * lda : A + (long)m * lda;
        wrow[i] = n < N ? W + (long)n * rm.ldw : nullptr;
    }
    const int wave_base = __builtin_amdgcn_readfirstlane(wave * 64 * 16);
    auto issue = [&](int k0, int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + lch[i] * EPC;
            const T* pa = zero;
            const T* pw = zero;
            if (k < K) {
                if (arow[i]) {
                    long ao = k;
                    if (rm.tap_len > 0) {
                        const int tap = k / rm.tap_len;
                        ao = (long)tap * rm.tap_stride + (k - tap * rm.tap_len);
                    }
                    pa = arow[i] + ao;
                }
                if (wrow[i]) pw = wrow[i] + k;
            }
            unsigned char* dst = &lds[stage][0] + i * 4096 + wave_base;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa,
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pw,
                                             (__attribute__((address_space(3))) void*)(dst + TILE_BYTES), 16, 0, 0);
        }
    };

    const int nk = (K + BK - 1) / BK;
    issue(0, 0);
    for (int t = 0; t < nk; ++t) {
        __syncthreads();     // tile t has landed (vmcnt drained) and every wave is done reading the other stage
        if (t + 1 < nk) issue((t + 1) * BK, (t + 1) & 1);
        const unsigned char* sA = &lds[t & 1][0];
        const unsigned char* sW = sA + TILE_BYTES;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t fa[4], fw[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ra = wm + i * 16 + l15;
                    fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + ra * ROWB + (((kk * 4 + lg) ^ (ra & 7)) << 4));
                    const int rw = wn + i * 16 + l15;
                    fw[i] = *reinterpret_cast<const bf16x8_t*>(sW + rw * ROWB + (((kk * 4 + lg) ^ (rw & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fw[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                float fa[4], fw[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ra = wm + i * 16 + l15;
                    fa[i] = *reinterpret_cast<const float*>(sA + ra * ROWB + ((ks ^ (ra & 7)) << 4) + lg * 4);
                    const int rw = wn + i * 16 + l15;
                    fw[i] = *reinterpret_cast<const float*>(sW + rw * ROWB + ((ks ^ (rw & 7)) << 4) + lg * 4);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fw[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // Epilogue through LDS, like gemm_tile3_kernel: the C fragments (2- or 4-byte elements strided over rows per lane, which
    // the texture-address unit serialises into 64 cycles per wave instruction) are staged as f32 rows per wave and read
    // back row-wise, so every lane moves 16 contiguous bytes of one output row.  The 64 KiB ring only holds 32 of the wave's
    // 64 rows at pitch 68, so the tile goes out in two halves.  Residual values are fetched first, all of them: the
    // residual may alias C (in-place adds), and a load can never be hoisted above an earlier store.
    constexpr int VEC = 16 / (int)sizeof(CT);   // output elements per 16-byte store: 8 (bf16) / 4 (f32)
    constexpr int LPR = 64 / VEC;               // lanes per output row
    constexpr int RPI = 64 / LPR;               // rows per wave instruction
    constexpr int NIT = 64 / RPI;
    constexpr int PITCH = 68;
    const bool grouped = rm.G < M;
    const int cc = (lane % LPR) * VEC;
    const int n = n0 + wn + cc;
    const bool full = n + VEC <= N;
    auto row_off = [&](int m, long& coff, long& roff) {
        if (grouped) {
            const int mb = m / rm.G, mr = m - mb * rm.G;
            coff = (long)mb * rm.c_bs + (long)mr * ldc;
            roff = (long)mb * rm.r_bs + (long)mr * rm.ldr;
        } else {
            coff = (long)m * ldc;
            roff = (long)m * rm.ldr;
        }
    };
    float rres[NIT][VEC];
    const bool has_res = residual != nullptr || rm.res_c != nullptr;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) rres[it][e] = 0.f;
        const int m = m0 + wm + it * RPI + lane / LPR;
        if (!has_res || m >= M || n >= N) continue;
        long coff, roff;
        row_off(m, coff, roff);
        if (residual) {
            const float* rp = residual + roff + n;
            if (full && ((uintptr_t)rp & 15) == 0) {
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) {
                    const float4 r4 = *reinterpret_cast<const float4*>(rp + q * 4);
                    rres[it][q * 4 + 0] = r4.x; rres[it][q * 4 + 1] = r4.y; rres[it][q * 4 + 2] = r4.z; rres[it][q * 4 + 3] = r4.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (n + e < N) rres[it][e] = rp[e];
            }
        }
        if (rm.res_c) {
            const CT* rp = (const CT*)rm.res_c + coff + n;
            if (full && ((uintptr_t)rp & 15) == 0) {
                if constexpr (sizeof(CT) == 4) {
                    const float4 r4 = *reinterpret_cast<const float4*>(rp);
                    rres[it][0] += r4.x; rres[it][1] += r4.y; rres[it][2] += r4.z; rres[it][3] += r4.w;
                } else {
                    const u32x4_t pk = *reinterpret_cast<const u32x4_t*>(rp);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        rres[it][2 * q] += bf16_bits_to_f32((unsigned short)(pk[q] & 0xffffu));
                        rres[it][2 * q + 1] += bf16_bits_to_f32((unsigned short)(pk[q] >> 16));
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (n + e < N) rres[it][e] += Elem<CT>::ld(rp + e);
            }
        }
    }
    // per-column constants of the lane's VEC columns: snake tables (channel wraps at sn_mod)
    const int chmod = rm.sn_mod > 0 ? rm.sn_mod : (1 << 30);
    float s1a[VEC], s1b[VEC], s2a[VEC], s2b[VEC];
    {
        int ch = n < N ? n % chmod : 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const bool in = n + e < N;
            s1a[e] = (in && act == S2S_ACT_SNAKE) ? rm.sn_a[ch] : 0.f;
            s1b[e] = (in && act == S2S_ACT_SNAKE) ? rm.sn_ib[ch] : 0.f;
            s2a[e] = (in && rm.C2) ? rm.sn2_a[ch] : 0.f;
            s2b[e] = (in && rm.C2) ? rm.sn2_ib[ch] : 0.f;
            if (++ch == chmod) ch = 0;
        }
    }
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nc = n0 + wn + j * 16 + l15;
        bv[j] = (bias && nc < N) ? bias[nc] : 0.f;
    }
    __syncthreads();            // every wave is done reading the ring
    float* stg = reinterpret_cast<float*>(&lds[0][0]) + wave * (32 * PITCH);
    auto store_vec = [&](CT* dp, const float* w) {
        if (full && ((uintptr_t)dp & 15) == 0) {
            if constexpr (sizeof(CT) == 4) {
                *reinterpret_cast<float4*>(dp) = make_float4(w[0], w[1], w[2], w[3]);
            } else {
                u32x4_t pk;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    pk[q] = pack_bf16x2_hw(w[2 * q], w[2 * q + 1]);
                *reinterpret_cast<u32x4_t*>(dp) = pk;
            }
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (n + e < N) Elem<CT>::st(dp + e, w[e]);
        }
    };
    auto half = [&](auto hc) {       // (compile-time half index: a runtime one under the activation dispatch sends the accumulators to scratch)
        constexpr int h = decltype(hc)::value;
        with_act(act, [&](auto actc) {
            constexpr int ACTC = decltype(actc)::value;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stg[(ii * 16 + lg * 4 + r) * PITCH + j * 16 + l15] = apply_act_c<CT, ACTC>(acc[h * 2 + ii][j][r] + bv[j]) * rm.alpha;
        });
        S2S_WAVE_LDS_HANDOVER();
#pragma unroll
        for (int i2 = 0; i2 < NIT / 2; ++i2) {
            const int it = h * (NIT / 2) + i2;
            const int hrow = i2 * RPI + lane / LPR;      // row inside the staged half
            const int m = m0 + wm + h * 32 + hrow;
            if (m >= M || n >= N) continue;
            long coff, roff;
            row_off(m, coff, roff);
            float v[VEC];
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
                const float4 t4 = *reinterpret_cast<const float4*>(stg + hrow * PITCH + cc + q * 4);
                v[q * 4 + 0] = t4.x; v[q * 4 + 1] = t4.y; v[q * 4 + 2] = t4.z; v[q * 4 + 3] = t4.w;
            }
            if (act == S2S_ACT_SNAKE) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = snake_f<CT>(v[e], s1a[e], s1b[e]);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] += rres[it][e];
            store_vec(C + coff + n, v);
            if (rm.C2) {
                float w[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) w[e] = snake_f<CT>(round_to<CT>(v[e]), s2a[e], s2b[e]);
                store_vec((CT*)rm.C2 + coff + n, w);
            }
        }
        S2S_WAVE_LDS_HANDOVER();      // the next half overwrites the staging rows
    };
    half(std::integral_constant<int, 0>{});
    half(std::integral_constant<int, 1>{});
}

// ---- large-M tile GEMM: 256x128 tile, 8 waves, 3-stage LDS ring ----------------------------
// The encoder-sized projections (M = sessions x frames) run here.  Same fragment layout and swizzle as gemm_tile_kernel,
// but (a) the per-thread source pointers are advanced incrementally (no address arithmetic or branches in the K loop;
// requires K % BK == 0 and no tap addressing), (b) the ring holds three K-slabs, so two slabs of direct global->LDS
// loads are in flight while one is multiplied - the wait before a slab is a partial vmcnt wait (the 6 loads of the
// younger slab may still be outstanding), and (c) the workgroup tile is 256 x 128 (4 x 2 waves of 64 x 64), which
// halves the W re-reads per flop.  LDS: 3 x (32 KiB A + 16 KiB W) = 144 KiB -> one 8-wave workgroup per CU.
constexpr int T3_M = 256, T3_N = 128, T3_A_BYTES = T3_M * ROWB, T3_W_BYTES = T3_N * ROWB, T3_STAGE = T3_A_BYTES + T3_W_BYTES, T3_NST = 3;

// s_waitcnt vmcnt(n) only (expcnt / lgkmcnt left unconstrained); gfx9 encoding: vmcnt = {simm16[15:14], simm16[3:0]}
#define S2S_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14))

template <typename T, typename CT, bool GEN, int WM = 4>
__global__ __launch_bounds__(128 * WM) void gemm_tile3_kernel(
    const T* __restrict__ A, long lda, const T* __restrict__ W, const float* __restrict__ bias,
    const float* residual, CT* C, long ldc, int M, int N, int K, int act, int tiles_m, RowMap rm) {
    constexpr int EPC = MfmaTraits<T>::EPC;
    constexpr int BK = ROWB / (int)sizeof(T);
    // WM = 4: the 256 x 128 tile of 8 waves, 3-stage ring.  WM = 2 (round 6, mid-sized M: a few hundred .. a few thousand rows, where a launch has
    // at most one tile per CU and its time is the tile's own latency): 128 x 128, 4 waves, FOUR stages - three slabs in flight.
    constexpr int TM = 64 * WM, NT = 128 * WM, NWL = 1024 / NT, LPS = 4 + NWL;      // rows, threads, W loads and all loads per thread and slab
    constexpr int A_BYTES = TM * ROWB, STAGE = A_BYTES + T3_W_BYTES, NST = WM == 4 ? T3_NST : 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    {
        const int z1 = blockIdx.y / rm.nz2, z2 = blockIdx.y % rm.nz2;
        A += z1 * rm.a_z1 + z2 * rm.a_z2;
        W += z1 * rm.w_z1 + z2 * rm.w_z2;
        C += z1 * rm.c_z1 + z2 * rm.c_z2;
        if (residual) residual += z1 * rm.c_z1 + z2 * rm.c_z2;
    }
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(wg, tiles_m, (int)gridDim.x / tiles_m, tm, tn);
    const int m0 = tm * TM, n0 = tn * T3_N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // staging: A tile = 2048 16-byte chunks (4 per thread), W tile = 1024 (2 per thread); chunk c -> (row c >> 3, slot c & 7),
    // LDS slot p of a row holds logical chunk p ^ (row & 7).  Out-of-range rows stream the zero page with step 0.
    // GEN = false: K % BK == 0 and plain rows - the K loop only bumps pointers.  GEN = true additionally tracks every
    // chunk's absolute k (zero page past K) and its offset inside the current convolution tap (tap_len >= BK: at most one
    // tap boundary per slab; crossing it adds tap_stride - tap_len), still without divisions in the loop.
    const T* zero = reinterpret_cast<const T*>(g_zero_page);
    const T* pa[4];
    const T* pw[NWL];
    int sa[4], sw[NWL];
    int ka[GEN ? 4 : 1], kin[GEN ? 4 : 1], kw[GEN ? NWL : 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = tid + i * NT, row = c >> 3, m = m0 + row;
        const int lch = (c & 7) ^ (row & 7);
        pa[i] = zero;
        sa[i] = 0;
        if constexpr (GEN) {
            ka[i] = lch * EPC;
            kin[i] = 0;
        }
        if (m < M) {
            const T* rowp = rm.G < M ? A + (long)(m / rm.G) * rm.a_bs + (long)(m % rm.G) * lda : A + (long)m * lda;
            long ao = lch * EPC;
            if constexpr (GEN) {
                if (rm.tap_len > 0) {
                    const int tap = (lch * EPC) / rm.tap_len;
                    kin[i] = lch * EPC - tap * rm.tap_len;
                    ao = (long)tap * rm.tap_stride + kin[i];
                }
            }
            pa[i] = rowp + ao;
            sa[i] = BK;
        }
    }
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
        const int c = tid + i * NT, row = c >> 3, n = n0 + row;
        const int lch = (c & 7) ^ (row & 7);
        pw[i] = zero;
        sw[i] = 0;
        if constexpr (GEN) kw[i] = lch * EPC;
        if (n < N) {
            pw[i] = W + (long)n * rm.ldw + lch * EPC;
            sw[i] = BK;
        }
    }
    const int wave_base = __builtin_amdgcn_readfirstlane(wave * 1024);
    auto issue = [&](int stage) {
        unsigned char* base = lds3 + stage * STAGE + wave_base;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const T* src = pa[i];
            if constexpr (GEN) {
                if (ka[i] >= K) src = zero;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + i * (NT * 16)), 16, 0, 0);
            pa[i] += sa[i];
            if constexpr (GEN) {
                ka[i] += BK;
                if (rm.tap_len > 0 && sa[i] != 0) {
                    kin[i] += BK;
                    if (kin[i] >= rm.tap_len) {
                        kin[i] -= rm.tap_len;
                        pa[i] += rm.tap_stride - rm.tap_len;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const T* src = pw[i];
            if constexpr (GEN) {
                if (kw[i] >= K) src = zero;
                kw[i] += BK;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + A_BYTES + i * (NT * 16)), 16, 0, 0);
            pw[i] += sw[i];
        }
    };

    const int nk = (K + BK - 1) / BK;
    issue(0);
    if (nk > 1) issue(1);
    if (NST == 4 && nk > 2) issue(2);
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
        // slab t has landed; the loads of the NST - 2 younger slabs may still be in flight
        if (NST == 4 && t + 2 < nk) S2S_WAIT_VMCNT(2 * LPS);
        else if (t + 1 < nk) S2S_WAIT_VMCNT(LPS);
        else S2S_WAIT_VMCNT(0);
        __builtin_amdgcn_s_barrier();        // ... for every wave; and every wave is done reading slab t - 1's stage
        if (t + NST - 1 < nk) issue(stage >= 1 ? stage - 1 : NST - 1);   // (t + NST - 1) % NST == (t - 1) % NST
        const unsigned char* sA = lds3 + stage * STAGE;
        const unsigned char* sW = sA + A_BYTES;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t fa[4], fw[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ra = wm + i * 16 + l15;
                    fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + ra * ROWB + (((kk * 4 + lg) ^ (ra & 7)) << 4));
                    const int rw = wn + i * 16 + l15;
                    fw[i] = *reinterpret_cast<const bf16x8_t*>(sW + rw * ROWB + (((kk * 4 + lg) ^ (rw & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fw[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                float fa[4], fw[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ra = wm + i * 16 + l15;
                    fa[i] = *reinterpret_cast<const float*>(sA + ra * ROWB + ((ks ^ (ra & 7)) << 4) + lg * 4);
                    const int rw = wn + i * 16 + l15;
                    fw[i] = *reinterpret_cast<const float*>(sW + rw * ROWB + ((ks ^ (rw & 7)) << 4) + lg * 4);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fw[j], acc[i][j], 0, 0, 0);
            }
        }
        stage = stage + 1 == NST ? 0 : stage + 1;
    }

    // Epilogue through LDS: the C fragments (column-strided per lane) are staged as a 64 x 64 f32 tile per wave
    // (row pitch 68 floats: conflict-free fragment writes) and read back row-wise, so every lane stores 16 contiguous
    // bytes of one output row (8 rows x 128 B per wave instruction for bf16) instead of 2-byte scattered stores.
    constexpr int VEC = 16 / (int)sizeof(CT);   // output elements per 16-byte store: 8 (bf16) / 4 (f32)
    constexpr int LPR = 64 / VEC;               // lanes per output row
    constexpr int RPI = 64 / LPR;               // rows per wave instruction
    constexpr int NIT = 64 / RPI;
    const bool grouped = rm.G < M;
    const int cc = (lane % LPR) * VEC;
    const int n = n0 + wn + cc;
    const bool full = n + VEC <= N;
    // Residual values first, all of them, before the staging barrier: the residual may alias C (in-place adds), so a load
    // can never be moved above an earlier store - inside the store loop they were NIT serial load -> store round trips per
    // lane with nothing else resident on the CU to hide them.  Issued here their latency overlaps the LDS staging.
    float rres[NIT][VEC];
    const bool has_res = residual != nullptr || rm.res_c != nullptr;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) rres[it][e] = 0.f;
        const int m = m0 + wm + it * RPI + lane / LPR;
        if (!has_res || m >= M || n >= N) continue;
        long coff, roff;
        if (grouped) {
            const int mb = m / rm.G, mr = m - mb * rm.G;
            coff = (long)mb * rm.c_bs + (long)mr * ldc;
            roff = (long)mb * rm.r_bs + (long)mr * rm.ldr;
        } else {
            coff = (long)m * ldc;
            roff = (long)m * rm.ldr;
        }
        if (residual) {
            const float* rp = residual + roff + n;
            if (full && ((uintptr_t)rp & 15) == 0) {
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) {
                    const float4 r4 = *reinterpret_cast<const float4*>(rp + q * 4);
                    rres[it][q * 4 + 0] = r4.x; rres[it][q * 4 + 1] = r4.y; rres[it][q * 4 + 2] = r4.z; rres[it][q * 4 + 3] = r4.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (n + e < N) rres[it][e] = rp[e];
            }
        }
        if (rm.res_c) {
            const CT* rp = (const CT*)rm.res_c + coff + n;
            if (full && ((uintptr_t)rp & 15) == 0) {
                if constexpr (sizeof(CT) == 4) {
                    const float4 r4 = *reinterpret_cast<const float4*>(rp);
                    rres[it][0] += r4.x; rres[it][1] += r4.y; rres[it][2] += r4.z; rres[it][3] += r4.w;
                } else {
                    const u32x4_t pk = *reinterpret_cast<const u32x4_t*>(rp);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        rres[it][2 * q] += bf16_bits_to_f32((unsigned short)(pk[q] & 0xffffu));
                        rres[it][2 * q + 1] += bf16_bits_to_f32((unsigned short)(pk[q] >> 16));
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (n + e < N) rres[it][e] += Elem<CT>::ld(rp + e);
            }
        }
    }
    __builtin_amdgcn_s_barrier();            // every wave is done reading the ring
    constexpr int PITCH = 68;
    float* stg = reinterpret_cast<float*>(lds3) + wave * (64 * PITCH);
    {
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nc = n0 + wn + j * 16 + l15;
            bv[j] = (bias && nc < N) ? bias[nc] : 0.f;
        }
        with_act(act, [&](auto actc) {
            constexpr int ACTC = decltype(actc)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stg[(i * 16 + lg * 4 + r) * PITCH + j * 16 + l15] = apply_act_c<CT, ACTC>(acc[i][j][r] + bv[j]) * rm.alpha;
        });
    }
    S2S_WAVE_LDS_HANDOVER();
    const int chmod = rm.sn_mod > 0 ? rm.sn_mod : (1 << 30);      // snake channel of column n (wraps at sn_mod)
    const int ch0 = n < N ? n % chmod : 0;
    float s1a[VEC], s1b[VEC], s2a[VEC], s2b[VEC];      // the lane's columns are fixed over the row loop
    {
        int ch = ch0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const bool in = n + e < N;
            s1a[e] = (in && act == S2S_ACT_SNAKE) ? rm.sn_a[ch] : 0.f;
            s1b[e] = (in && act == S2S_ACT_SNAKE) ? rm.sn_ib[ch] : 0.f;
            s2a[e] = (in && rm.C2) ? rm.sn2_a[ch] : 0.f;
            s2b[e] = (in && rm.C2) ? rm.sn2_ib[ch] : 0.f;
            if (++ch == chmod) ch = 0;
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = it * RPI + lane / LPR;
        const int m = m0 + wm + row;
        if (m >= M || n >= N) continue;
        long coff;
        if (grouped) {
            const int mb = m / rm.G, mr = m - mb * rm.G;
            coff = (long)mb * rm.c_bs + (long)mr * ldc;
        } else {
            coff = (long)m * ldc;
        }
        float v[VEC];
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            const float4 t4 = *reinterpret_cast<const float4*>(stg + row * PITCH + cc + q * 4);
            v[q * 4 + 0] = t4.x; v[q * 4 + 1] = t4.y; v[q * 4 + 2] = t4.z; v[q * 4 + 3] = t4.w;
        }
        CT* cp = C + coff + n;
        if (act == S2S_ACT_SNAKE) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = snake_f<CT>(v[e], s1a[e], s1b[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] += rres[it][e];
        auto store_vec = [&](CT* dp, const float* w) {
            if (full && ((uintptr_t)dp & 15) == 0) {
                if constexpr (sizeof(CT) == 4) {
                    *reinterpret_cast<float4*>(dp) = make_float4(w[0], w[1], w[2], w[3]);
                } else {
                    u32x4_t pk;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        pk[q] = pack_bf16x2_hw(w[2 * q], w[2 * q + 1]);
                    *reinterpret_cast<u32x4_t*>(dp) = pk;
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (n + e < N) Elem<CT>::st(dp + e, w[e]);
            }
        };
        store_vec(cp, v);
        if (rm.C2) {
            float w[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) w[e] = snake_f<CT>(round_to<CT>(v[e]), s2a[e], s2b[e]);
            store_vec((CT*)rm.C2 + coff + n, w);
        }
    }
}

// ---- k = 7 dilated causal convolution with TAP REUSE (round 6) -------------------------------------------------------------------------
// The codec decoder's residual-unit convolutions (C -> C, kernel 7, dilation 1 / 3 / 9; tts.cpp) ran as a GEMM over K = 7 C whose A rows are
// re-staged global -> LDS once per TAP: 7 x the fill bytes of the rows a tile actually needs, on a kernel whose main loop is bound by exactly that
// fill path (profiles/r06_bench_conv.txt: 12.7 - 21.1 GB of LDS fill per launch at 512 live rows against 0.4 - 2 GB of HBM traffic; 0.18 - 0.35 of
// the MFMA peak).  Here the K loop runs [channel chunk][tap]: a TALL tile - the 256 output rows plus the 6 x dilation history rows in front of
// them, BKC channels wide - is staged ONCE per channel chunk (double buffered) and the seven taps read their A fragments from it at row offsets
// tap x dilation; only the weight slabs (128 columns x BKC channels of one tap) still stream through a 3-stage ring.  Fill bytes per 256 x 128
// tile and chunk: (256 + 6 d) rows + 7 x 128 weight rows instead of 7 x (256 + 128).  Same 256 x 128 tile, 8 waves of 64 x 64, LDS-DMA with the
// XOR swizzle in the source address, counted waits, one barrier per (chunk, tap); epilogue = gemm_tile3's (bias, SnakeBeta, 16-byte row stores).
// Preconditions (launch_gemm checks them, anything else keeps the tap-by-tap GEMM): bf16, 7 taps, lda == tap_len == C, C % BKC == 0,
// tap_stride = d x C with 6 d <= 64, every item's rows a multiple of 256 (a tile never straddles two sessions), no residual / second output.
// The accumulation order over K differs from the GEMM's ([chunk][tap] instead of [tap][channel]): same products, f32 sums in another order.
template <int BKC> struct Cv7 {
    static constexpr int ROWB_ = BKC * 2;                 // bytes per LDS row
    static constexpr int RPP = 1024 / ROWB_;              // rows per 1 KiB DMA piece (one wave instruction)
    static constexpr int CPR = ROWB_ / 16;                // 16-byte chunks per row
    static constexpr int NA = BKC == 64 ? 5 : 3;          // tall-tile pieces per wave (8 waves): 320 rows (BKC 64) / 384 rows (BKC 32) >= 256 + 54
    static constexpr int NW = (128 / RPP) / 8;            // weight-slab pieces per wave: 2 / 1
    static constexpr int TALL_ROWS = NA * 8 * RPP;
    static constexpr int TALLB = TALL_ROWS * ROWB_;
    static constexpr int WSTB = 128 * ROWB_;
    static constexpr int KK = BKC / 32;
    __device__ static __forceinline__ int swz(int row) { return BKC == 64 ? (row & 7) : ((row >> 2) & 3); }
};
constexpr int CV7_LDS = 8 * 64 * 68 * 4;                  // the epilogue's staging (8 waves x 64 x 68 floats) is the largest user: 136 KiB

template <typename CT, int BKC>
__global__ __launch_bounds__(512) void conv7_tile_kernel(const __hip_bfloat16* __restrict__ A, const __hip_bfloat16* __restrict__ W,
                                                          const float* __restrict__ bias, CT* C, long ldc, int M, int N, int Cin, int dil, int act,
                                                          int tiles_m, RowMap rm) {
    using V = Cv7<BKC>;
    constexpr int ROWB_ = V::ROWB_, RPP = V::RPP, CPR = V::CPR, NA = V::NA, NW = V::NW, KK = V::KK;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsc[];
    unsigned char* const tall0 = ldsc;
    unsigned char* const wring = ldsc + 2 * V::TALLB;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(wg, tiles_m, (int)gridDim.x / tiles_m, tm, tn);
    const int m0 = tm * 256, n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int ncc = Cin / BKC;
    const int tall_valid = 256 + 6 * dil;                  // rows of the tall tile that exist (and are read); the rest re-read the last one
    // the tile's rows belong to ONE item (G % 256 == 0): input row j of the tall tile = A + item * a_bs + (r0 + j) * C
    const __hip_bfloat16* Abase = A + (long)(m0 / rm.G) * rm.a_bs + (long)(m0 % rm.G) * Cin;

    const __hip_bfloat16* pa[NA];
    const __hip_bfloat16* pw[NW];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = (wave * NA + i) * RPP + lane / CPR, slot = lane % CPR;
        pa[i] = Abase + (long)min(row, tall_valid - 1) * Cin + ((slot ^ V::swz(row)) << 3);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int row = (wave * NW + i) * RPP + lane / CPR, slot = lane % CPR;
        pw[i] = W + (long)min(n0 + row, N - 1) * rm.ldw + ((slot ^ V::swz(row)) << 3);
    }
    auto issue_a = [&](int cc) {
        unsigned char* base = tall0 + (cc & 1) * V::TALLB + wave * (NA * 1024);
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa[i] + cc * BKC),
                                             (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
    };
    // weight slabs in loop order: slab s = (chunk s / 7, tap s % 7) -> ring stage s % 3; slabs past the end re-read the last one (never consumed),
    // so that every wait of the loop is the same instruction
    int w_cc = 0, w_tap = 0, w_stage = 0;
    auto issue_w = [&]() {
        unsigned char* base = wring + w_stage * V::WSTB + wave * (NW * 1024);
        const long off = (long)w_tap * Cin + w_cc * BKC;
#pragma unroll
        for (int i = 0; i < NW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pw[i] + off),
                                             (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
        w_stage = w_stage == 2 ? 0 : w_stage + 1;
        if (w_cc + 1 < ncc || w_tap < 6) {
            if (++w_tap == 7) { w_tap = 0; ++w_cc; }
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    issue_a(0);
    issue_w();
    issue_w();
    int stage = 0;
    for (int cc = 0; cc < ncc; ++cc) {
        const bool more = cc + 1 < ncc;
        const unsigned char* sA = tall0 + (cc & 1) * V::TALLB;
#pragma unroll
        for (int tap = 0; tap < 7; ++tap) {
            // slab (cc, tap) has landed when at most the loads issued AFTER it are outstanding: the next slab (NW), plus - behind tap 0 of a chunk
            // that has a successor - the next chunk's tall tile (NA), which was issued between this slab's request and the next one's
            if ((tap == 1 || tap == 2) && more) S2S_WAIT_VMCNT(NW + NA);
            else S2S_WAIT_VMCNT(NW);
            __builtin_amdgcn_s_barrier();          // ... for every wave; and every wave is done with slab s - 1 (its ring stage, and at tap 0 the other tall buffer)
            issue_w();                             // slab s + 2 into the stage slab s - 1 used
            if (tap == 0 && more) issue_a(cc + 1);
            const unsigned char* sW = wring + stage * V::WSTB;
            const int shift = tap * dil;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                bf16x8_t fa[4], fw[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ra = wm + i * 16 + l15 + shift;
                    fa[i] = *reinterpret_cast<const bf16x8_t*>(sA + ra * ROWB_ + (((kk * 4 + lg) ^ V::swz(ra)) << 4));
                    const int rw = wn + i * 16 + l15;
                    fw[i] = *reinterpret_cast<const bf16x8_t*>(sW + rw * ROWB_ + (((kk * 4 + lg) ^ V::swz(rw)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fw[j], acc[i][j], 0, 0, 0);
            }
            stage = stage == 2 ? 0 : stage + 1;
        }
    }
    S2S_WAIT_VMCNT(0);                           // the redundant tail loads still target the ring
    __builtin_amdgcn_s_barrier();                // every wave is done reading; the staging below reuses the LDS

    // ---- epilogue: gemm_tile3's, without residuals / second output (bias, activation or SnakeBeta, 16-byte row-contiguous stores) ----
    constexpr int VEC = 16 / (int)sizeof(CT), LPR = 64 / VEC, RPI = 64 / LPR, NIT = 64 / RPI, PITCH = 68;
    const int cc_ = (lane % LPR) * VEC;
    const int n = n0 + wn + cc_;
    const bool full = n + VEC <= N;
    float* stg = reinterpret_cast<float*>(ldsc) + wave * (64 * PITCH);
    {
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nc = n0 + wn + j * 16 + l15;
            bv[j] = (bias && nc < N) ? bias[nc] : 0.f;
        }
        with_act(act, [&](auto actc) {
            constexpr int ACTC = decltype(actc)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stg[(i * 16 + lg * 4 + r) * PITCH + j * 16 + l15] = apply_act_c<CT, ACTC>(acc[i][j][r] + bv[j]) * rm.alpha;
        });
    }
    S2S_WAVE_LDS_HANDOVER();
    const int chmod = rm.sn_mod > 0 ? rm.sn_mod : (1 << 30);
    float s1a[VEC], s1b[VEC];
    {
        int ch = n < N ? n % chmod : 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const bool in = n + e < N;
            s1a[e] = (in && act == S2S_ACT_SNAKE) ? rm.sn_a[ch] : 0.f;
            s1b[e] = (in && act == S2S_ACT_SNAKE) ? rm.sn_ib[ch] : 0.f;
            if (++ch == chmod) ch = 0;
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = it * RPI + lane / LPR;
        const int m = m0 + wm + row;
        if (m >= M || n >= N) continue;
        const int mb = m / rm.G, mr = m - mb * rm.G;
        float v[VEC];
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            const float4 t4 = *reinterpret_cast<const float4*>(stg + row * PITCH + cc_ + q * 4);
            v[q * 4 + 0] = t4.x; v[q * 4 + 1] = t4.y; v[q * 4 + 2] = t4.z; v[q * 4 + 3] = t4.w;
        }
        if (act == S2S_ACT_SNAKE) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = snake_f<CT>(v[e], s1a[e], s1b[e]);
        }
        CT* cp = C + (long)mb * rm.c_bs + (long)mr * ldc + n;
        if (full && ((uintptr_t)cp & 15) == 0) {
            if constexpr (sizeof(CT) == 4) {
                *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                u32x4_t pk;
#pragma unroll
                for (int q = 0; q < 4; ++q) pk[q] = pack_bf16x2_hw(v[2 * q], v[2 * q + 1]);
                *reinterpret_cast<u32x4_t*>(cp) = pk;
            }
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (n + e < N) Elem<CT>::st(cp + e, v[e]);
        }
    }
}

// ---- 256x256 tile GEMM: 8 waves of 128x64 (or 4 of 128x128), 4-stage ring of 32-wide K slabs (bf16) ----------------------------
// Validated in round 3 (tools/proto/tile4_check.py, profiles/r03_tile4_check.txt; tests/test_ops_gpu.py runs the encoder shapes through
// it): the 8-wave variant is the default for the encoder-sized projections.
// Why: gemm_tile3's 64x64 wave tiles need 32 flop per LDS byte - exactly what the LDS delivers (128 B/clk/CU against
// 4069 flop/clk/CU) - so its main loop can not pass ~50 % of the MFMA peak.  Here every wave owns a 128x128 output block
// (8x8 fragments, 256 accumulator registers, one wave per SIMD): 16 fragment reads feed 64 MFMAs per 32-wide K slab, i.e.
// 64 flop per LDS byte.  The ring holds FOUR 32 KiB slabs (256 A rows + 256 W rows of 64 bytes), three of them in flight
// behind the one being multiplied (1.3 us of look-ahead at full MFMA rate); loads past the end of K stream the zero page so
// that the `s_waitcnt vmcnt(16)` before every slab is the same instruction for the whole loop.  LDS rows are 64 bytes = 4
// chunks of 16; chunk p of row r holds logical chunk p ^ ((r >> 2) & 3), which makes the 16 rows x 16 bytes of one
// fragment-read group hit 64 distinct banks (rows r, r+4, r+8, r+12 share a 64-byte bank window and take different chunks).
constexpr int T4_M = 256, T4_N = 256, T4_BK = 32, T4_ROWB = 64, T4_A_BYTES = T4_M * T4_ROWB, T4_W_BYTES = T4_N * T4_ROWB,
              T4_STAGE = T4_A_BYTES + T4_W_BYTES, T4_NST = 4;

// NWN = 4: 8 waves (2 x 4), wave tile 128 x 64 (44 flop per LDS byte, 128 accumulator registers, TWO waves per SIMD that hide
//          each other's fragment reads).  (The 4-wave NWN = 2 variant - 128 x 128 wave tiles, one wave per SIMD - measured slower than
//          gemm_tile3 in round 3 and is no longer instantiated.)
// (Round 4's two software-pipelined variants of this main loop - S2S_TILE4_PIPE - measured within +-3 % of the plain loop on an MI355X,
// profiles/r05_tile4_check.txt, and were removed when gemm_tile5 took over the encoder shapes.)
template <typename CT, int NWN>
__global__ __launch_bounds__(128 * NWN) void gemm_tile4_kernel(
    const __hip_bfloat16* __restrict__ A, long lda, const __hip_bfloat16* __restrict__ W, const float* __restrict__ bias,
    const float* residual, CT* C, long ldc, int M, int N, int K, int act, int tiles_m, RowMap rm) {
    using T = __hip_bfloat16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds4[];
    {
        const int z1 = blockIdx.y / rm.nz2, z2 = blockIdx.y % rm.nz2;
        A += z1 * rm.a_z1 + z2 * rm.a_z2;
        W += z1 * rm.w_z1 + z2 * rm.w_z2;
        C += z1 * rm.c_z1 + z2 * rm.c_z2;
        if (residual) residual += z1 * rm.c_z1 + z2 * rm.c_z2;
    }
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn;
    tile_coords(wg, tiles_m, (int)gridDim.x / tiles_m, tm, tn);
    const int m0 = tm * T4_M, n0 = tn * T4_N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    constexpr int NTH = 128 * NWN;              // threads
    constexpr int WNF = 16 / NWN;               // 16-column fragments per wave: 8 / 4
    constexpr int WCOLS = WNF * 16;             // wave tile width: 128 / 64
    constexpr int CPT = 1024 / NTH;             // 16-byte chunks per thread and operand tile: 4 / 2
    const int wm = (wave / NWN) * 128, wn = (wave % NWN) * WCOLS;

    f32x4_t acc[8][WNF];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < WNF; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // staging: each operand tile = 256 rows x 4 chunks of 16 bytes = 1024 chunks, CPT per thread; chunk c -> (row c >> 2, slot c & 3),
    // LDS byte c * 16; the SOURCE chunk of slot p is p ^ ((row >> 2) & 3).  Rows past M / N stream the zero page with step 0.
    const T* zero = reinterpret_cast<const T*>(g_zero_page);
    const T* pa[CPT];
    const T* pw[CPT];
    int sa[CPT], sw[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = tid + i * NTH, row = c >> 2;
        const int lch = (c & 3) ^ ((row >> 2) & 3);
        const int m = m0 + row, n = n0 + row;
        pa[i] = zero; sa[i] = 0;
        pw[i] = zero; sw[i] = 0;
        if (m < M) {
            const T* rowp = rm.G < M ? A + (long)(m / rm.G) * rm.a_bs + (long)(m % rm.G) * lda : A + (long)m * lda;
            pa[i] = rowp + lch * 8;
            sa[i] = T4_BK;
        }
        if (n < N) {
            pw[i] = W + (long)n * rm.ldw + lch * 8;
            sw[i] = T4_BK;
        }
    }
    const int wave_base = __builtin_amdgcn_readfirstlane(wave * 1024);
    auto issue = [&](int stage, bool live) {
        unsigned char* base = lds4 + stage * T4_STAGE + wave_base;
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(live ? pa[i] : zero),
                                             (__attribute__((address_space(3))) void*)(base + i * (NTH * 16)), 16, 0, 0);
            pa[i] += sa[i];
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(live ? pw[i] : zero),
                                             (__attribute__((address_space(3))) void*)(base + T4_A_BYTES + i * (NTH * 16)), 16, 0, 0);
            pw[i] += sw[i];
        }
    };

    const int nk = K / T4_BK;
    issue(0, true);
    issue(1, 1 < nk);
    issue(2, 2 < nk);
    // fragment byte offsets inside a stage (constant over the loop): row * 64 + ((lg ^ ((row >> 2) & 3)) << 4)
    int offa[8], offw[WNF];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ra = wm + i * 16 + l15;
        offa[i] = ra * T4_ROWB + ((lg ^ ((ra >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int j = 0; j < WNF; ++j) {
        const int rw = wn + j * 16 + l15;
        offw[j] = T4_A_BYTES + rw * T4_ROWB + ((lg ^ ((rw >> 2) & 3)) << 4);
    }
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
        S2S_WAIT_VMCNT(4 * CPT);             // slab t has landed; the 2 x 2 CPT loads of slabs t + 1 and t + 2 may still be in flight
        __builtin_amdgcn_s_barrier();        // ... for every wave; and every wave is done reading slab t - 1's stage
        issue(stage >= 1 ? stage - 1 : T4_NST - 1, t + 3 < nk);      // (t + 3) % 4 == (t - 1) % 4
        const unsigned char* sS = lds4 + stage * T4_STAGE;
        {
            bf16x8_t fa[8], fw[WNF];
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sS + offa[i]);
#pragma unroll
            for (int j = 0; j < WNF; ++j) fw[j] = *reinterpret_cast<const bf16x8_t*>(sS + offw[j]);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < WNF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fw[j], acc[i][j], 0, 0, 0);
        }
        stage = stage + 1 == T4_NST ? 0 : stage + 1;
    }
    S2S_WAIT_VMCNT(0);                       // the zero-page loads past the end of K still target the ring
    __builtin_amdgcn_s_barrier();

    // Epilogue: 8 passes of 16 rows per wave, staged through LDS (pitch 132 floats: conflict-free fragment writes) and read
    // back row-wise so that every lane moves 16 contiguous bytes of one output row.  The residual values of a pass are
    // fetched before its first store (the residual may alias C).
    constexpr int VEC = 16 / (int)sizeof(CT);   // 8 (bf16) / 4 (f32)
    constexpr int PITCH = WCOLS + 4;            // 132 / 68 floats: conflict-free fragment writes
    constexpr int LPR = WCOLS / VEC;            // lanes per output row of the wave tile
    constexpr int RPI = 64 / LPR;               // rows per wave instruction
    constexpr int NIT = 16 / RPI;               // instructions per 16-row pass
    const bool grouped = rm.G < M;
    const int cc = (lane % LPR) * VEC;
    const int n = n0 + wn + cc;
    const bool full = n + VEC <= N;
    float* stg = reinterpret_cast<float*>(lds4) + wave * (16 * PITCH);
    float bv[WNF];
#pragma unroll
    for (int j = 0; j < WNF; ++j) {
        const int nc = n0 + wn + j * 16 + l15;
        bv[j] = (bias && nc < N) ? bias[nc] : 0.f;
    }
    auto row_off = [&](int m, long& coff, long& roff) {
        if (grouped) {
            const int mb = m / rm.G, mr = m - mb * rm.G;
            coff = (long)mb * rm.c_bs + (long)mr * ldc;
            roff = (long)mb * rm.r_bs + (long)mr * rm.ldr;
        } else {
            coff = (long)m * ldc;
            roff = (long)m * rm.ldr;
        }
    };
    // (the pass index must be a compile-time constant: a loop the compiler declines to unroll would index the accumulators
    // dynamically and copy all 256 of them to scratch)
    auto pass = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        float rres[NIT][VEC];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) rres[it][e] = 0.f;
            const int m = m0 + wm + i * 16 + it * RPI + lane / LPR;
            if (residual == nullptr || m >= M || n >= N) continue;
            long coff, roff;
            row_off(m, coff, roff);
            const float* rp = residual + roff + n;
            if (full && ((uintptr_t)rp & 15) == 0) {
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) {
                    const float4 r4 = *reinterpret_cast<const float4*>(rp + q * 4);
                    rres[it][q * 4 + 0] = r4.x; rres[it][q * 4 + 1] = r4.y; rres[it][q * 4 + 2] = r4.z; rres[it][q * 4 + 3] = r4.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (n + e < N) rres[it][e] = rp[e];
            }
        }
        with_act(act, [&](auto actc) {
            constexpr int ACTC = decltype(actc)::value;
#pragma unroll
            for (int j = 0; j < WNF; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    stg[(lg * 4 + r) * PITCH + j * 16 + l15] = apply_act_c<CT, ACTC>(acc[i][j][r] + bv[j]) * rm.alpha;
        });
        S2S_WAVE_LDS_HANDOVER();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int prow = it * RPI + lane / LPR;      // row inside the staged pass
            const int m = m0 + wm + i * 16 + prow;
            if (m >= M || n >= N) continue;
            long coff, roff;
            row_off(m, coff, roff);
            float v[VEC];
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
                const float4 t4 = *reinterpret_cast<const float4*>(stg + prow * PITCH + cc + q * 4);
                v[q * 4 + 0] = t4.x + rres[it][q * 4 + 0]; v[q * 4 + 1] = t4.y + rres[it][q * 4 + 1];
                v[q * 4 + 2] = t4.z + rres[it][q * 4 + 2]; v[q * 4 + 3] = t4.w + rres[it][q * 4 + 3];
            }
            CT* dp = C + coff + n;
            if (full && ((uintptr_t)dp & 15) == 0) {
                if constexpr (sizeof(CT) == 4) {
                    *reinterpret_cast<float4*>(dp) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    u32x4_t pk;
#pragma unroll
                    for (int q = 0; q < 4; ++q) pk[q] = pack_bf16x2_hw(v[2 * q], v[2 * q + 1]);
                    *reinterpret_cast<u32x4_t*>(dp) = pk;
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (n + e < N) Elem<CT>::st(dp + e, v[e]);
            }
        }
        S2S_WAVE_LDS_HANDOVER();      // the next pass overwrites the staging rows
    };
    pass(std::integral_constant<int, 0>{}); pass(std::integral_constant<int, 1>{}); pass(std::integral_constant<int, 2>{});
    pass(std::integral_constant<int, 3>{}); pass(std::integral_constant<int, 4>{}); pass(std::integral_constant<int, 5>{});
    pass(std::integral_constant<int, 6>{}); pass(std::integral_constant<int, 7>{});
}

// ---- 256x256 tile, 64-wide K tiles, two staggered wave groups ("8-phase" schedule of cdna_hip_programming.md section 5) --------------
// Round 5.  gemm_tile4's loop is lock-step: every wave reads its fragments, then every wave multiplies, one barrier per 32-wide slab, so
// the matrix pipe idles through every LDS latency (0.20 - 0.25 of the MFMA peak in situ for three rounds).  Here the 8 waves form two
// groups (waves 0-3 / 4-7: one wave of each group on every SIMD) that run the SAME instruction stream one barrier apart: while one group
// multiplies a 64x32 quadrant of its 128x64 wave tile (16 MFMAs under s_setprio 1) the other group issues its LDS fragment reads and the
// direct global->LDS loads of a later K tile, then the roles swap at the next barrier.  A K tile (64 wide) is FOUR phases:
//     phase 0: read B0 (4 x ds_read_b128) + A0 (8)   multiply acc[a0][b0]      stage half B1 of tile t + 1
//     phase 1: read B1 (4)                           multiply acc[a0][b1]      stage half A1 of tile t + 1
//     phase 2: read A1 (8)                           multiply acc[a1][b1]      stage half B0 of tile t + 2
//     phase 3: -                                     multiply acc[a1][b0]      stage half A0 of tile t + 2
// LDS = 2 buffers x 4 half-tiles x 16 KiB (128 rows x 128 bytes).  The half-tiles are laid out so that EVERY wave is done with a half
// after one phase: A half `ah` holds, for both wave rows, the 64 tile rows of quadrant row ah (LDS row r <-> tile row
// (r >> 6) * 128 + ah * 64 + (r & 63)), W half `bh` the 32 columns of quadrant column bh of all four wave columns (LDS row r <-> tile
// column (r >> 5) * 64 + bh * 32 + (r & 31)); the permutation lives in the per-thread SOURCE addresses, the LDS image of a
// global_load_lds is lane-linear as always.  One half-tile (2 loads per thread) is staged per phase, five to six phases ahead of its
// first read:
//   * WAR: a half read in phase g is restaged in phase g + 2 at the earliest - the other group runs one barrier behind, so its reads
//     of phase g are only known complete (its lgkmcnt(0)) once THIS group has passed two more barriers;
//   * RAW: `s_waitcnt vmcnt(8)` after the phase's own issue leaves the four youngest half-tiles in flight, i.e. everything issued
//     up to four phases ago has landed for this wave before the phase's first barrier; a read in the NEXT phase sits behind that
//     barrier for both groups (the lagging group passes it one barrier later still).  Needed: issue phase + 5 <= read phase (table).
// Loads past the end of K re-read the last K tile (never consumed) so that the counted wait is the same instruction all along; rows
// past M / N are clamped to the last valid row (their products are never stored).
// 128-byte LDS rows: 16-byte chunk p of row r holds logical chunk p ^ ((r >> 1) & 7) - the 16 lanes of every ds_read_b128 lane group
// (MI355X_MICROARCH.md, LDS table) then cover the 16 distinct 16-byte slots of a 256-byte bank window.
// VAR bit 0: no stagger (lock-step A/B arm); bit 1: no s_setprio.
constexpr int T5_M = 256, T5_N = 256, T5_BK = 64, T5_HALF = 128 * 128, T5_BUF = 4 * T5_HALF, T5_LDS = 2 * T5_BUF + 2048;   // (+ 2 KiB: tail of the epilogue's staging rows)

// the epilogue of the 256x256 kernels: wave tile 128 x 64 at (wm, wn), 8 passes of 16 rows staged through LDS (pitch 68 floats:
// conflict-free fragment writes) and read back row-wise so that every lane moves 16 contiguous bytes of one output row.  The residual
// values of a pass are fetched before its first store (the residual may alias C).
template <typename CT>
__device__ __forceinline__ void tile256_epilogue(f32x4_t (&acc)[8][4], unsigned char* lds, const float* __restrict__ bias, const float* residual,
                                                 CT* C, long ldc, int M, int N, int act, const RowMap& rm, int m0, int n0, int wm, int wn,
                                                 int lane, int wave) {
    constexpr int WNF = 4, WCOLS = 64;
    constexpr int VEC = 16 / (int)sizeof(CT);   // 8 (bf16) / 4 (f32)
    constexpr int PITCH = WCOLS + 4;
    constexpr int LPR = WCOLS / VEC;            // lanes per output row of the wave tile
    constexpr int RPI = 64 / LPR;               // rows per wave instruction
    constexpr int NIT = 16 / RPI;               // instructions per 16-row pass
    const int l15 = lane & 15, lg = lane >> 4;
    const bool grouped = rm.G < M;
    const int cc = (lane % LPR) * VEC;
    const int n = n0 + wn + cc;
    const bool full = n + VEC <= N;
    float* stg = reinterpret_cast<float*>(lds) + wave * (16 * PITCH);
    float bv[WNF];
#pragma unroll
    for (int j = 0; j < WNF; ++j) {
        const int nc = n0 + wn + j * 16 + l15;
        bv[j] = (bias && nc < N) ? bias[nc] : 0.f;
    }
    auto row_off = [&](int m, long& coff, long& roff) {
        if (grouped) {
            const int mb = m / rm.G, mr = m - mb * rm.G;
            coff = (long)mb * rm.c_bs + (long)mr * ldc;
            roff = (long)mb * rm.r_bs + (long)mr * rm.ldr;
        } else {
            coff = (long)m * ldc;
            roff = (long)m * rm.ldr;
        }
    };
    auto pass = [&](auto ic) {                  // (compile-time pass index: a runtime one would send the accumulators to scratch)
        constexpr int i = decltype(ic)::value;
        float rres[NIT][VEC];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) rres[it][e] = 0.f;
            const int m = m0 + wm + i * 16 + it * RPI + lane / LPR;
            if (residual == nullptr || m >= M || n >= N) continue;
            long coff, roff;
            row_off(m, coff, roff);
            const float* rp = residual + roff + n;
            if (full && ((uintptr_t)rp & 15) == 0) {
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) {
                    const float4 r4 = *reinterpret_cast<const float4*>(rp + q * 4);
                    rres[it][q * 4 + 0] = r4.x; rres[it][q * 4 + 1] = r4.y; rres[it][q * 4 + 2] = r4.z; rres[it][q * 4 + 3] = r4.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (n + e < N) rres[it][e] = rp[e];
            }
        }
        with_act(act, [&](auto actc) {
            constexpr int ACTC = decltype(actc)::value;
#pragma unroll
            for (int j = 0; j < WNF; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    stg[(lg * 4 + r) * PITCH + j * 16 + l15] = apply_act_c<CT, ACTC>(acc[i][j][r] + bv[j]) * rm.alpha;
        });
        S2S_WAVE_LDS_HANDOVER();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int prow = it * RPI + lane / LPR;      // row inside the staged pass
            const int m = m0 + wm + i * 16 + prow;
            if (m >= M || n >= N) continue;
            long coff, roff;
            row_off(m, coff, roff);
            float v[VEC];
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
                const float4 t4 = *reinterpret_cast<const float4*>(stg + prow * PITCH + cc + q * 4);
                v[q * 4 + 0] = t4.x + rres[it][q * 4 + 0]; v[q * 4 + 1] = t4.y + rres[it][q * 4 + 1];
                v[q * 4 + 2] = t4.z + rres[it][q * 4 + 2]; v[q * 4 + 3] = t4.w + rres[it][q * 4 + 3];
            }
            CT* dp = C + coff + n;
            if (full && ((uintptr_t)dp & 15) == 0) {
                if constexpr (sizeof(CT) == 4) {
                    *reinterpret_cast<float4*>(dp) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    u32x4_t pk;
#pragma unroll
                    for (int q = 0; q < 4; ++q) pk[q] = pack_bf16x2_hw(v[2 * q], v[2 * q + 1]);
                    *reinterpret_cast<u32x4_t*>(dp) = pk;
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (n + e < N) Elem<CT>::st(dp + e, v[e]);
            }
        }
        S2S_WAVE_LDS_HANDOVER();      // the next pass overwrites the staging rows
    };
    pass(std::integral_constant<int, 0>{}); pass(std::integral_constant<int, 1>{}); pass(std::integral_constant<int, 2>{});
    pass(std::integral_constant<int, 3>{}); pass(std::integral_constant<int, 4>{}); pass(std::integral_constant<int, 5>{});
    pass(std::integral_constant<int, 6>{}); pass(std::integral_constant<int, 7>{});
}

// The same epilogue for the common case - an interior tile (no ragged edge), rows ldc apart (no row groups), 16-byte aligned rows - as
// straight-line code.  tools/tile5_phases.py (round 5) showed the general epilogue above costing 17 us (bf16) / 34 us (f32 + residual)
// per tile against a 33 us main loop at K = 1280: the activation switch was taken per ELEMENT (128 per lane, ~80 instructions each),
// every store sat behind an s_waitcnt vmcnt(0) of the wavefront fences, and the residual of a pass was fetched only when the pass
// began (one HBM round trip per pass).  Here the activation is a template argument, the LDS staging relies on the in-order LDS queue of
// a wave (no fence, no wait on the stores), and the residual rows of pass p + 2 are in flight while pass p is stored.
// the four bias values of a lane (columns j * 16 + lane % 16 of the wave tile), requested before anything else of the epilogue (ordinary
// loads: the compiler waits for them - and only them - at their first use)
__device__ __forceinline__ void tile256_bias_fetch(const float* __restrict__ bias, int N, int col0, int lane, float (&bv)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = 0.f;
    if (bias) {
        unsigned long long b = (unsigned long long)bias;
        asm volatile("" : "+s"(b));
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const __attribute__((address_space(1))) float*>(b + (unsigned)min(col0 + j * 16 + (lane & 15), N - 1) * 4u);
    }
}
template <typename CT, int ACT, bool RES, typename NextF>
__device__ __forceinline__ void tile256_epilogue_fast(f32x4_t (&acc)[8][4], unsigned char* lds, const float* __restrict__ bias, int N, NextF&& next_prologue,
                                                      const float* residual, CT* C, long ldc, long ldr, float alpha, int m0, int n0, int wm, int wn,
                                                      int lane, int wave) {
    constexpr int VEC = 16 / (int)sizeof(CT);   // 8 (bf16) / 4 (f32) columns per lane
    constexpr int PITCH = 68;
    constexpr int LPR = 64 / VEC;               // lanes per output row of the wave tile
    constexpr int RPI = 64 / LPR;               // rows per wave instruction
    constexpr int NIT = 16 / RPI;               // instructions per 16-row pass
    constexpr int RV = VEC / 4;                 // float4 per lane and row
    constexpr int NL = NIT * RV;                // residual loads per pass and lane (4 for both output types)
    const int l15 = lane & 15, lg = lane >> 4;
    const int cc = (lane % LPR) * VEC, rl = lane / LPR;
    float* stg = reinterpret_cast<float*>(lds) + wave * (16 * PITCH);
    float* swr = stg + lg * 4 * PITCH + l15;
    const float* srd = stg + rl * PITCH + cc;
    float bv[4];
    tile256_bias_fetch(bias, N, n0 + wn, lane, bv);
    // Global addresses = wave-uniform row base (a scalar register pair, made opaque so that it is not folded into 64 per-row vector
    // addresses next to the 128 accumulators) + one 32-bit lane offset.  The LOADS are ordinary loads: the compiler counts their waits -
    // vmcnt(loads issued since), which is what is wanted.  The STORES are inline assembly, invisible to that counting: with loads and
    // stores of one counter in flight it would fall back to s_waitcnt vmcnt(0) before every use.  Loads return in order, stores do not
    // return in order with them, and a count that only relies on the younger LOADS is safe whatever the stores did.
    const unsigned c_lane = (unsigned)((rl * ldc + cc) * (long)sizeof(CT)), r_lane = (unsigned)((rl * ldr + cc) * 4);
    const unsigned long long c_wave = (unsigned long long)C + (unsigned long long)(((long)(m0 + wm) * ldc + (n0 + wn)) * (long)sizeof(CT));
    const unsigned long long r_wave = (unsigned long long)residual + (unsigned long long)(((long)(m0 + wm) * ldr + (n0 + wn)) * 4);
    const unsigned long long c_step = (unsigned long long)(RPI * ldc * (long)sizeof(CT)), r_step = (unsigned long long)(RPI * ldr * 4);
    f32x4_t rr[3][NIT][RV];
    auto fetch = [&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if constexpr (RES && p < 8) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                unsigned long long row = r_wave + (unsigned long long)(p * NIT + it) * r_step;
                unsigned lane_off = r_lane;
                asm volatile("" : "+s"(row), "+v"(lane_off));      // (scalar base + 32-bit lane offset survive instruction selection as such)
#pragma unroll
                for (int q = 0; q < RV; ++q) rr[p % 3][it][q] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4_t*>(row + lane_off + q * 16);    // (global, not flat)
            }
        }
    };
    auto pass = [&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p > 0) fetch(std::integral_constant<int, p + 2>{});
        if constexpr (ACT == S2S_ACT_GELU && sizeof(CT) == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    f32x2_t v2 = {acc[p][j][r] + bv[j], acc[p][j][r + 1] + bv[j]};
                    v2 = gelu_poly2(v2) * alpha;
                    swr[r * PITCH + j * 16] = v2[0];
                    swr[(r + 1) * PITCH + j * 16] = v2[1];
                }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) swr[r * PITCH + j * 16] = apply_act_c<CT, ACT>(acc[p][j][r] + bv[j]) * alpha;
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();          // (the LDS queue of a wave is in order: the reads below see the writes above)
        f32x4_t t[NIT][RV];
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int q = 0; q < RV; ++q) t[it][q] = *reinterpret_cast<const f32x4_t*>(srd + it * RPI * PITCH + q * 4);
        if constexpr (RES) {                      // (requested two passes ago)
#pragma unroll
            for (int it = 0; it < NIT; ++it)
#pragma unroll
                for (int q = 0; q < RV; ++q) t[it][q] += rr[p % 3][it][q];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const unsigned long long row = c_wave + (unsigned long long)(p * NIT + it) * c_step;
            if constexpr (sizeof(CT) == 4) {
                asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(c_lane), "v"(t[it][0]), "s"(row) : "memory");   // (s_nop: the data registers may be rewritten next)
            } else {
                u32x4_t pk;
                pk[0] = pack_bf16x2_hw(t[it][0][0], t[it][0][1]); pk[1] = pack_bf16x2_hw(t[it][0][2], t[it][0][3]);
                pk[2] = pack_bf16x2_hw(t[it][1][0], t[it][1][1]); pk[3] = pack_bf16x2_hw(t[it][1][2], t[it][1][3]);
                asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(c_lane), "v"(pk), "s"(row) : "memory");
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();          // the next pass overwrites the staging rows (same queue: after the reads above)
        __builtin_amdgcn_sched_barrier(0);
    };
    fetch(std::integral_constant<int, 0>{});
    fetch(std::integral_constant<int, 1>{});
    fetch(std::integral_constant<int, 2>{});
    next_prologue();
    pass(std::integral_constant<int, 0>{}); pass(std::integral_constant<int, 1>{}); pass(std::integral_constant<int, 2>{});
    pass(std::integral_constant<int, 3>{}); pass(std::integral_constant<int, 4>{}); pass(std::integral_constant<int, 5>{});
    pass(std::integral_constant<int, 6>{}); pass(std::integral_constant<int, 7>{});
}
// entry: the straight-line variants for the epilogues the encoders use, the general one for everything else
template <typename CT, typename NextF>
__device__ __forceinline__ void tile256_store(f32x4_t (&acc)[8][4], unsigned char* lds, const float* __restrict__ bias, NextF&& next_prologue,
                                              const float* residual, CT* C, long ldc, int M, int N, int act, const RowMap& rm, int m0, int n0,
                                              int wm, int wn, int lane, int wave) {
    const bool fast = rm.G >= M && m0 + T5_M <= M && n0 + T5_N <= N && (((uintptr_t)C | (uintptr_t)residual | (uintptr_t)(ldc * sizeof(CT)) | (uintptr_t)(rm.ldr * 4)) & 15) == 0;
    if (fast && act == S2S_ACT_NONE && residual)
        tile256_epilogue_fast<CT, S2S_ACT_NONE, true>(acc, lds, bias, N, next_prologue, residual, C, ldc, rm.ldr, rm.alpha, m0, n0, wm, wn, lane, wave);
    else if (fast && act == S2S_ACT_NONE && !residual)
        tile256_epilogue_fast<CT, S2S_ACT_NONE, false>(acc, lds, bias, N, next_prologue, residual, C, ldc, rm.ldr, rm.alpha, m0, n0, wm, wn, lane, wave);
    else if (fast && act == S2S_ACT_GELU && !residual)
        tile256_epilogue_fast<CT, S2S_ACT_GELU, false>(acc, lds, bias, N, next_prologue, residual, C, ldc, rm.ldr, rm.alpha, m0, n0, wm, wn, lane, wave);
    else if (fast && act == S2S_ACT_SILU && !residual)
        tile256_epilogue_fast<CT, S2S_ACT_SILU, false>(acc, lds, bias, N, next_prologue, residual, C, ldc, rm.ldr, rm.alpha, m0, n0, wm, wn, lane, wave);
    else {
        next_prologue();
        tile256_epilogue<CT>(acc, lds, bias, residual, C, ldc, M, N, act, rm, m0, n0, wm, wn, lane, wave);
        // conditional loads of the general path may still be in flight as far as the compiler's counting can tell; left alone it protects
        // the fragment registers of the NEXT tile's main loop with an s_waitcnt vmcnt(0) in every K tile (persistent arms)
        S2S_WAIT_VMCNT(0);
    }
}

// Persistent launch (PERSIST): gridDim.x = min(tiles, compute units) workgroups, each walks the logical tile ids blockIdx.x,
// blockIdx.x + gridDim.x, ...  gridDim.x is a multiple of 8, so a workgroup's tiles keep the XCD the one-tile-per-workgroup launch
// gives them (xcd_remap / tile_coords unchanged): what goes away is the workgroup launch between two tiles of a compute unit.
// dbg (development, s2s_op_gemm_tile_debug): 6 x u64 per tile - {start, prologue landed, main loop done, epilogue issued, HW_ID, XCC_ID},
// times on the 100 MHz counter - for tools/tile5_phases.py.
template <typename CT, int VAR>
__global__ __launch_bounds__(512) void gemm_tile5_kernel(
    const __hip_bfloat16* __restrict__ A, long lda, const __hip_bfloat16* __restrict__ W, const float* __restrict__ bias,
    const float* residual, CT* C, long ldc, int M, int N, int K, int act, int tiles_m, int tiles_total, RowMap rm, unsigned long long* dbg, int spread_ticks) {
    using T = __hip_bfloat16;
    constexpr bool STAGGER = (VAR & 1) == 0, PRIO = (VAR & 2) == 0, PERSIST = (VAR & 4) != 0, EARLY = (VAR & 8) != 0, SPREAD = (VAR & 16) != 0;
    static_assert(!EARLY || PERSIST, "the early prologue belongs to the persistent walk");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds5[];
    {
        const int z1 = blockIdx.y / rm.nz2, z2 = blockIdx.y % rm.nz2;
        A += z1 * rm.a_z1 + z2 * rm.a_z2;
        W += z1 * rm.w_z1 + z2 * rm.w_z2;
        C += z1 * rm.c_z1 + z2 * rm.c_z2;
        if (residual) residual += z1 * rm.c_z1 + z2 * rm.c_z2;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;    // wave tile: rows wr * 128 .., columns wc * 64 ..; group = wr
    // staging: a half-tile = 128 LDS rows x 8 chunks of 16 bytes = 1024 chunks, 2 per thread; chunk c -> (row c >> 3, slot c & 7),
    // LDS byte c * 16 inside the half; the SOURCE chunk of slot p is p ^ ((row >> 1) & 7) of the mapped tile row / column.
    const T* pa[2][2];
    const T* pw[2][2];
    auto setup = [&](int v, int& m0_, int& n0_) {
        const int wg = xcd_remap(v, tiles_total);
        int tm, tn;
        tile_coords(wg, tiles_m, tiles_total / tiles_m, tm, tn);
        m0_ = tm * T5_M;
        n0_ = tn * T5_N;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = i * 512 + tid, r = c >> 3;
                const int lch = (c & 7) ^ ((r >> 1) & 7);
                const int m = min(m0_ + (r >> 6) * 128 + hh * 64 + (r & 63), M - 1);
                const T* rowp = rm.G < M ? A + (long)(m / rm.G) * rm.a_bs + (long)(m % rm.G) * lda : A + (long)m * lda;
                pa[hh][i] = rowp + lch * 8;
                const int n = min(n0_ + (r >> 5) * 64 + hh * 32 + (r & 31), N - 1);
                pw[hh][i] = W + (long)n * rm.ldw + lch * 8;
            }
    };
    const int wave_base = wave * 1024;
    // half index inside a buffer: 0 = B0, 1 = A0, 2 = B1, 3 = A1 (the order they are consumed in)
    auto stage_half = [&](int buf, int h, const T* (&p)[2], int adv) {
        unsigned char* base = lds5 + buf * T5_BUF + h * T5_HALF + wave_base;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p[i],
                                             (__attribute__((address_space(3))) void*)(base + i * 8192), 16, 0, 0);
            p[i] += adv;
        }
    };
    // fragment byte offsets inside a half-tile: row * 128 + (((kh * 4 + lg) ^ (row >> 1 & 7)) << 4), row & 15 = l15 for every fragment
    const int sw = (lg ^ (l15 >> 1)) << 4;
    const int offa0 = T5_HALF + (wr * 64 + l15) * 128 + sw, offa1 = offa0 ^ 64;      // A0 of buffer 0, k-half 0 / 1
    const int offb0 = (wc * 32 + l15) * 128 + sw, offb1 = offb0 ^ 64;                // B0 of buffer 0
    bf16x8_t fa[4][2], fb0[2][2], fb1[2][2];
    auto read_a = [&](int buf, int ah) {
        const unsigned char* s = lds5 + buf * T5_BUF + ah * (2 * T5_HALF);
#pragma unroll
        for (int ff = 0; ff < 4; ++ff) {
            fa[ff][0] = *reinterpret_cast<const bf16x8_t*>(s + offa0 + ff * 2048);
            fa[ff][1] = *reinterpret_cast<const bf16x8_t*>(s + offa1 + ff * 2048);
        }
    };
    auto read_b = [&](int buf, int bh, bf16x8_t (&fb)[2][2]) {
        const unsigned char* s = lds5 + buf * T5_BUF + bh * (2 * T5_HALF);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            fb[c][0] = *reinterpret_cast<const bf16x8_t*>(s + offb0 + c * 2048);
            fb[c][1] = *reinterpret_cast<const bf16x8_t*>(s + offb1 + c * 2048);
        }
    };
    // one quadrant: 64 rows (ah) x 32 columns (bh) x K = 64: 16 MFMAs, two of them per accumulator, eight apart
#define S2S_T5_QUAD(AH, BH, FB)                                                                                           \
    do {                                                                                                                  \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                                  \
        _Pragma("unroll") for (int ff = 0; ff < 4; ++ff)                                                                  \
        _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                                     \
            acc[(AH) * 4 + ff][(BH) * 2 + c] =                                                                            \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ff][kh], FB[c][kh], acc[(AH) * 4 + ff][(BH) * 2 + c], 0, 0, 0); \
    } while (0)
    // end of a phase's load segment: everything staged up to four phases ago has landed for this wave; meet the other group; own
    // fragment reads complete.  End of the multiply segment: meet again.
#define S2S_T5_LOADS_DONE()                                  \
    do {                                                     \
        S2S_WAIT_VMCNT(8);                                   \
        __builtin_amdgcn_sched_barrier(0);                   \
        __builtin_amdgcn_s_barrier();                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);                   \
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);   \
    } while (0)
#define S2S_T5_MFMAS_DONE()                                  \
    do {                                                     \
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);   \
        __builtin_amdgcn_sched_barrier(0);                   \
        __builtin_amdgcn_s_barrier();                        \
        __builtin_amdgcn_sched_barrier(0);                   \
    } while (0)

    const int nk = K / T5_BK;
    // prologue: tile 0 complete, B0 / A0 of tile 1 (units 0 .. 5 of the staging sequence; phase g of the loop issues unit g + 6)
    auto prologue = [&]() {
        stage_half(0, 0, pw[0], 1 < nk ? T5_BK : 0);
        stage_half(0, 1, pa[0], 1 < nk ? T5_BK : 0);
        stage_half(0, 2, pw[1], 1 < nk ? T5_BK : 0);
        stage_half(0, 3, pa[1], 1 < nk ? T5_BK : 0);
        stage_half(1, 0, pw[0], 2 < nk ? T5_BK : 0);
        stage_half(1, 1, pa[0], 2 < nk ? T5_BK : 0);
    };
    if constexpr (SPREAD) {
        // All compute units reach their epilogues together and the stores of 256 tiles then share ~2 TB/s of write bandwidth
        // (tools/tile5_phases.py: 17 us per bf16 tile, 34 us with the f32 residual, against a 33 us main loop at K = 1280).  The
        // persistent workgroups start spread over one tile period - position inside the XCD / 32 - and keep that phase, so only a
        // fraction of the units store at any time.  (blockIdx.x >> 3 is small for the workgroups that walk one tile more.)
        const long wait = (long)((blockIdx.x >> 3) & 31) * spread_ticks / 32;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < wait) __builtin_amdgcn_s_sleep(16);
    }
    int m0 = 0, n0 = 0;
    bool primed = false;                         // EARLY: the prologue of this tile was issued before the previous tile's epilogue
  for (int vt = blockIdx.x; vt < tiles_total; vt += gridDim.x) {
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (dbg) ts0 = __builtin_amdgcn_s_memrealtime();
    if (!primed) {
        setup(vt, m0, n0);
        prologue();
    }
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    S2S_WAIT_VMCNT(8);                           // B0 / A0 of tile 0 have landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (dbg) ts1 = __builtin_amdgcn_s_memrealtime();
    if constexpr (STAGGER) {
        if (wr == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0 from here on
    }
    __builtin_amdgcn_sched_barrier(0);
    auto ktile = [&](auto bufc, int t) {
        constexpr int b = decltype(bufc)::value;
        const int adv1 = t + 2 < nk ? T5_BK : 0, adv2 = t + 3 < nk ? T5_BK : 0;
        // phase 0
        read_b(b, 0, fb0);
        read_a(b, 0);
        stage_half(b ^ 1, 2, pw[1], adv1);
        S2S_T5_LOADS_DONE();
        S2S_T5_QUAD(0, 0, fb0);
        S2S_T5_MFMAS_DONE();
        // phase 1
        read_b(b, 1, fb1);
        stage_half(b ^ 1, 3, pa[1], adv1);
        S2S_T5_LOADS_DONE();
        S2S_T5_QUAD(0, 1, fb1);
        S2S_T5_MFMAS_DONE();
        // phase 2
        read_a(b, 1);
        stage_half(b, 0, pw[0], adv2);
        S2S_T5_LOADS_DONE();
        S2S_T5_QUAD(1, 1, fb1);
        S2S_T5_MFMAS_DONE();
        // phase 3
        stage_half(b, 1, pa[0], adv2);
        S2S_T5_LOADS_DONE();
        S2S_T5_QUAD(1, 0, fb0);
        S2S_T5_MFMAS_DONE();
    };
    int t = 0;
    for (; t + 1 < nk; t += 2) {
        ktile(std::integral_constant<int, 0>{}, t);
        ktile(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < nk) ktile(std::integral_constant<int, 0>{}, t);
#undef S2S_T5_QUAD
#undef S2S_T5_LOADS_DONE
#undef S2S_T5_MFMAS_DONE
    if constexpr (STAGGER) {
        if (wr == 0) __builtin_amdgcn_s_barrier();   // group 0 waits for group 1's last phase
    }
    S2S_WAIT_VMCNT(0);                           // the loads past the end of K still target the ring
    __builtin_amdgcn_s_barrier();
    if (dbg) ts2 = __builtin_amdgcn_s_memrealtime();
    int m1 = 0, n1 = 0;
    // EARLY: the ring is quiescent, so the next tile's prologue flies under this tile's epilogue; the epilogue issues it AFTER its own
    // first loads (bias, residual rows), which then never wait behind it.  It is issued unconditionally - the last tile of a workgroup
    // re-requests its own first K tiles - so that every wait count of the epilogue is a constant (a conditional prologue makes the
    // compiler fall back to s_waitcnt vmcnt(0)).
    auto next_prologue = [&]() {
        if constexpr (EARLY) {
            const int nx = vt + (int)gridDim.x;
            primed = nx < tiles_total;
            setup(primed ? nx : vt, m1, n1);
            prologue();
        }
    };
    // the epilogue's staging rows: halves 2 / 3 of buffer 1 (+ 2 KiB past the ring) - the six half-tiles of a prologue stay clear of them
    tile256_store<CT>(acc, lds5 + T5_BUF + 2 * T5_HALF, bias, next_prologue, residual, C, ldc, M, N, act, rm, m0, n0, wr * 128, wc * 64, lane, wave);
    if (dbg && tid == 0) {
        unsigned long long* d = dbg + ((long)blockIdx.y * tiles_total + vt) * 6;
        d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = __builtin_amdgcn_s_memrealtime();
        d[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID: wave / simd / pipe / cu / sh / se
        d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // HW_REG_XCC_ID
    }
    if constexpr (EARLY) { m0 = m1; n0 = n1; }
    if constexpr (PERSIST) __builtin_amdgcn_s_barrier();   // phase 0 of the next tile restages the epilogue's staging rows
  }
    if constexpr (EARLY) S2S_WAIT_VMCNT(0);      // (the last tile's prologue targets this workgroup's LDS)
}

// ---- skinny (decode) GEMM --------------------------------------------------
// M <= 64 rows (one per live session): pure weight streaming.  One workgroup of 8 waves owns a 16-column slab of
// W over the WHOLE of K; the 8 waves split K, so every weight element is read exactly once, straight from HBM into
// an MFMA B-fragment (16 B per lane, all loads of a wave's slice in flight together), activations come from L2 as
// A-fragments.  The 8 partial accumulators are combined through LDS (deterministic - no atomics), then the full
// epilogue (bias, activation, alpha, residual, cast) runs on the combined tile.  grid = ceil(N/16) workgroups.
// NF = 16-column fragments per workgroup: every activation fragment a wave fetches is reused for NF weight
// fragments (the fragment-shaped activation loads, not the weight stream, are what saturates the load path at M = 64).
// GLU = true: W holds [gate rows | up rows] (N each); the workgroup accumulates both for its channels and the
// epilogue writes silu(gate) * up (SwiGLU fused, output width N).
// NORM = true (bf16 only): A points at the raw f32 residual rows; the A fragments are converted to bf16 in registers, the
// rows' sums of squares come out of the same loads (per-wave partials -> LDS), and the combined tile is scaled by
// rsqrt(mean(x^2) + eps) in the epilogue (RMSNorm with its weight pre-folded into W): no separate norm launch.
template <typename T, typename CT, int MT, int NF, bool GLU, bool NORM>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(
    const T* __restrict__ A, long lda, const T* __restrict__ W, const float* __restrict__ bias,
    const float* residual, CT* C, long ldc, int M, int N, int K, int act, RowMap rm, int ksplit, float rms_eps) {
    constexpr int EPC = MfmaTraits<T>::EPC;          // k per 16-byte load: 8 (bf16) / 4 (f32)
    constexpr int KB = EPC * 4;                      // k consumed per load step by the 4 lane groups: 32 / 16
    constexpr int NS = NF * (GLU ? 2 : 1);           // weight streams per wave
    constexpr int U = 8 / NS;                        // load steps in flight per stream
    constexpr int RS = GLU ? 2 * MT : MT;
    __shared__ float red[8][RS][64][4];
    __shared__ float red_ss[NORM ? 8 : 1][64];
    __shared__ float rstd_s[64];
    {
        const int z1 = blockIdx.y / rm.nz2, z2 = blockIdx.y % rm.nz2;
        A += z1 * rm.a_z1 + z2 * rm.a_z2;
        W += z1 * rm.w_z1 + z2 * rm.w_z2;
        C += z1 * rm.c_z1 + z2 * rm.c_z2;
        if (residual) residual += z1 * rm.c_z1 + z2 * rm.c_z2;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    // blockIdx.x = n-slab * ksplit + k-part (ksplit is 1: every output element is produced by exactly one workgroup)
    const int kpart = blockIdx.x % ksplit;
    const int n0 = (blockIdx.x / ksplit) * 16 * NF;
    const int kblk = ((K + ksplit - 1) / ksplit + KB - 1) / KB * KB;
    const int kb0 = kpart * kblk, kb1 = min(K, kb0 + kblk);
    const int kslice = ((kblk + 7) / 8 + KB - 1) / KB * KB;
    const int kbeg = kb0 + wave * kslice, kend = min(kb1, kbeg + kslice);

    f32x4_t acc[MT][NS];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[i][s] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const T* wrow[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int n = n0 + (s % NF) * 16 + l15;
        wrow[s] = n < N ? W + ((long)n + (s >= NF ? N : 0)) * rm.ldw : nullptr;
    }
    const T* arow[MT];
    float ss[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = i * 16 + l15;
        arow[i] = nullptr;
        ss[i] = 0.f;
        if (m < M) {
            if constexpr (NORM) arow[i] = reinterpret_cast<const T*>(reinterpret_cast<const float*>(A) + (long)m * lda);
            else arow[i] = rm.G < M ? A + (long)(m / rm.G) * rm.a_bs + (long)(m % rm.G) * lda : A + (long)m * lda;
        }
    }
    for (int k0 = kbeg; k0 < kend; k0 += U * KB) {
        u32x4_t fw[U][NS];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * KB + lg * EPC;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                fw[u][s] = (u32x4_t){0u, 0u, 0u, 0u};
                if (wrow[s] && k < kend) fw[u][s] = *reinterpret_cast<const u32x4_t*>(wrow[s] + k);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * KB + lg * EPC;
            long ao = k;
            if (rm.tap_len > 0) {
                const int tap = k / rm.tap_len;
                ao = (long)tap * rm.tap_stride + (k - tap * rm.tap_len);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                u32x4_t va = {0u, 0u, 0u, 0u};
                if constexpr (NORM) {
                    if (arow[i] && k < kend) {
                        const float* ap = reinterpret_cast<const float*>(arow[i]) + ao;
                        const f32x4_t x0 = *reinterpret_cast<const f32x4_t*>(ap), x1 = *reinterpret_cast<const f32x4_t*>(ap + 4);
                        ss[i] += (x0[0] * x0[0] + x0[1] * x0[1]) + (x0[2] * x0[2] + x0[3] * x0[3]) +
                                 (x1[0] * x1[0] + x1[1] * x1[1]) + (x1[2] * x1[2] + x1[3] * x1[3]);
                        typedef __attribute__((ext_vector_type(8))) float f32x8_t;
                        const f32x8_t xf = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                        va = __builtin_bit_cast(u32x4_t, __builtin_convertvector(xf, bf16x8_t));
                    }
                } else {
                    if (arow[i] && k < kend) va = *reinterpret_cast<const u32x4_t*>(arow[i] + ao);
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if constexpr (sizeof(T) == 2) {
                        acc[i][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, va), __builtin_bit_cast(bf16x8_t, fw[u][s]), acc[i][s], 0, 0, 0);
                    } else {
                        // lane (lg, .) holds k = kb + lg*4 + e for both operands; MFMA step e pairs element e of each
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4)
                            acc[i][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(va[e4]), __uint_as_float(fw[u][s][e4]), acc[i][s], 0, 0, 0);
                    }
                }
            }
        }
    }
    const bool grouped = rm.G < M;
    if constexpr (NORM) {
        // row sums of squares: lanes with the same l15 hold different k chunks of row i*16 + l15 -> combine the 4 lane
        // groups, then the 8 waves' K slices through LDS
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float v = ss[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lg == 0) red_ss[wave][i * 16 + l15] = v;
        }
        __syncthreads();
        if (tid < MT * 16) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red_ss[w][tid];
            rstd_s[tid] = rsqrtf(v / (float)K + rms_eps);
        }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (f > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red[wave][i][lane][r] = acc[i][f][r];
                if constexpr (GLU) red[wave][MT + i][lane][r] = acc[i][NF + f][r];
            }
        __syncthreads();
        // combined tile element e -> (fragment i, lane l, reg r): row = i*16 + (l>>4)*4 + r, col = l & 15
        for (int e = tid; e < MT * 256; e += 512) {
            const int i = e >> 8, l = (e >> 2) & 63, r = e & 3;
            const int m = i * 16 + (l >> 4) * 4 + r, nn = n0 + f * 16 + (l & 15);
            if (m >= M || nn >= N) continue;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[w][i][l][r];
            if constexpr (NORM) v *= rstd_s[m];
            if constexpr (GLU) {
                float u2 = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) u2 += red[w][MT + i][l][r];
                if constexpr (NORM) u2 *= rstd_s[m];
                v = v / (1.0f + expf(-v)) * u2;
            }
            long coff, roff;
            if (grouped) {
                const int mb = m / rm.G, mr = m - mb * rm.G;
                coff = (long)mb * rm.c_bs + (long)mr * ldc;
                roff = (long)mb * rm.r_bs + (long)mr * rm.ldr;
            } else {
                coff = (long)m * ldc;
                roff = (long)m * rm.ldr;
            }
            CT* cp = C + coff + nn;
            v = apply_act(v + (bias ? bias[nn] : 0.f), act) * rm.alpha;
            if (residual) v += residual[roff + nn];
            Elem<CT>::st(cp, v);
        }
    }
}

// the 256x256 kernel serves the encoder-sized bf16 projections (S2S_TILE4=0 keeps them on gemm_tile3; measured round 3,
// profiles/r03_tile4_check.txt: parity OK on every shape, 8-wave variant +2 .. +19 % over gemm_tile3 on the Whisper encoder GEMMs,
// the 4-wave variant slower than gemm_tile3 - deleted in round 4)
static int tile4_mode() {
    static const int mode = [] {
        const char* e = getenv("S2S_TILE4");
        return e ? atoi(e) : 1;
    }();
    return mode;
}
template <typename T>
static bool tile4_ok(const GemmArgs& g, int M, int N, int K, int nz, bool ext) {
    const bool on = tile4_mode() > 0;
    return on && sizeof(T) == 2 && !ext && g.tap_len == 0 && K % T4_BK == 0 && K >= 4 * T4_BK && M >= 2048 && N >= 512 &&
           (long)cdiv(M, T4_M) * cdiv(N, T4_N) * nz >= 192;
}

// S2S_TILE5 (default 1): the staggered 8-phase 256x256 kernel (round 5) takes the shapes tile4 served (K a multiple of 64); = 0: off
// (gemm_tile4 / gemm_tile3 as in round 4), = 2: every shape it can run (parity tests: small / ragged shapes too).  S2S_TILE5_VAR picks
// the arm: bit 0 lock-step, bit 1 no s_setprio, bit 2 persistent workgroups (default), bit 3 next tile's prologue under the epilogue,
// bit 4 workgroups spread over one tile period (profiles/r05_tile5.txt: 4 and 12 are equal within noise, 28 is slower).
static int tile5_mode() {
    static const int mode = [] {
        const char* e = getenv("S2S_TILE5");
        return e ? atoi(e) : 1;
    }();
    return mode;
}
static int tile5_var() {
    static const int v = [] {
        const char* e = getenv("S2S_TILE5_VAR");
        return e ? atoi(e) & 31 : 4;
    }();
    return v;
}
// persistent arm (S2S_TILE5_VAR=4): workgroups per launch = compute units (a multiple of 8: one XCD per workgroup for all its tiles)
static int tile5_persist_wgs() {
    static const int n = [] {
        if (const char* e = getenv("S2S_TILE5_WGS")) return max(8, atoi(e) / 8 * 8);
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return max(8, cus / 8 * 8);
    }();
    return n;
}
static int tile5_spread_pct() {
    static const int v = [] {
        const char* e = getenv("S2S_TILE5_SPREAD");
        return e ? atoi(e) : 100;
    }();
    return v;
}
static unsigned long long*& tile5_dbg() {
    static unsigned long long* p = nullptr;
    return p;
}
void tile5_set_debug(unsigned long long* dbg) { tile5_dbg() = dbg; }

template <typename T>
static bool tile5_ok(const GemmArgs& g, int M, int N, int K, int nz, bool ext) {
    if (tile5_mode() <= 0 || sizeof(T) != 2 || ext || g.tap_len != 0 || K % T5_BK != 0) return false;
    if (tile5_mode() >= 2) return M > 128;      // parity runs: every shape the kernel can take (the decode kernels keep M <= 128)
    return K >= 2 * T5_BK && M >= 2048 && N >= 512 && (long)cdiv(M, T5_M) * cdiv(N, T5_N) * nz >= 192;
}

// the tap-reuse convolution kernel takes the codec's residual-unit convolutions (conv7_tile_kernel above); S2S_CONV7=0 / s2s_op_conv7_tune(0): off
static int g_conv7 = -1;
void conv7_set(int on) { g_conv7 = on < 0 ? -1 : (on != 0); }
static bool conv7_enabled() {
    static const bool env = [] {
        const char* e = getenv("S2S_CONV7");
        return !e || atoi(e) != 0;
    }();
    return g_conv7 < 0 ? env : g_conv7 != 0;
}
template <typename T>
static bool conv7_ok(const GemmArgs& g, const RowMap& rm, int M, int N, int K, int nz) {
    if (sizeof(T) != 2 || !conv7_enabled() || g.tap_len <= 0 || nz != 1) return false;
    const int Cin = g.tap_len;
    if (K != 7 * Cin || g.lda != Cin || Cin % 32 != 0 || g.tap_stride % Cin != 0) return false;
    const long dil = g.tap_stride / Cin;
    if (dil < 1 || 6 * dil > 64) return false;
    if (rm.G <= 0 || rm.G % 256 != 0 || M % 256 != 0 || rm.ldw < K) return false;
    if (g.residual || g.res_c || g.C2 || (g.alpha_set && g.alpha != 1.0f)) return false;
    if (g.act != S2S_ACT_NONE && g.act != S2S_ACT_SNAKE && g.act != S2S_ACT_GELU && g.act != S2S_ACT_SILU && g.act != S2S_ACT_RELU) return false;
    return (long)(M / 256) * cdiv(N, 128) >= 128;      // enough tiles to fill the chip; small launches keep the GEMM path
}

static bool gemm_trace() {
    static const bool on = getenv("S2S_GEMM_TRACE") != nullptr;
    return on;
}

static bool tile3_disabled() {
    static const bool off = getenv("S2S_NO_TILE3") != nullptr;   // A/B switch for tools/bench_ops.py
    return off;
}

// S2S_TILE3_MID=0: mid-sized launches on gemm_tile_kernel as in rounds 1 - 5 (A/B switch)
static bool tile3_mid_enabled() {
    static const bool on = [] {
        const char* e = getenv("S2S_TILE3_MID");
        return !e || atoi(e) != 0;
    }();
    return on;
}

template <typename T, typename CT>
static int launch_gemm(hipStream_t st, const GemmArgs& g) {
    RowMap rm;
    rm.G = g.G > 0 ? g.G : (g.M > 0 ? g.M : 1);
    rm.a_bs = g.a_bs;
    rm.c_bs = g.c_bs;
    rm.r_bs = g.r_bs;
    rm.ldr = g.ldr > 0 ? g.ldr : g.ldc;
    rm.ldw = g.ldw > 0 ? g.ldw : g.K;
    const int nz1 = g.nz1 > 0 ? g.nz1 : 1;
    rm.nz2 = g.nz2 > 0 ? g.nz2 : 1;
    const int nz = nz1 * rm.nz2;
    rm.a_z1 = g.a_z1; rm.a_z2 = g.a_z2; rm.w_z1 = g.w_z1; rm.w_z2 = g.w_z2; rm.c_z1 = g.c_z1; rm.c_z2 = g.c_z2;
    rm.alpha = g.alpha_set ? g.alpha : 1.0f;
    rm.tap_len = g.tap_len;
    rm.tap_stride = g.tap_stride;
    rm.sn_a = g.snake_a; rm.sn_ib = g.snake_ib; rm.sn_mod = g.snake_mod;
    rm.res_c = g.res_c;
    rm.C2 = g.C2; rm.sn2_a = g.snake2_a; rm.sn2_ib = g.snake2_ib;
    const bool ext = g.act == S2S_ACT_SNAKE || g.res_c || g.C2;      // tile-kernel epilogues only
    const int M = g.M, N = g.N, K = g.K;
    if (gemm_trace()) {      // development (S2S_GEMM_TRACE=1): every distinct shape once, with the path it takes
        static std::mutex mu;
        static std::set<std::string> seen;
        char key[256];
        const char* path = M <= 128 && !(g.glu && M > 64) && !ext ? "skinny" : conv7_ok<T>(g, rm, M, N, K, nz) ? "conv7" : tile5_ok<T>(g, M, N, K, nz, ext) ? "tile5" : tile4_ok<T>(g, M, N, K, nz, ext) ? "tile4" : "tile3/tile";
        snprintf(key, sizeof key, "gemm %s M=%d N=%d K=%d nz=%d G=%d tap=%d act=%d res=%d ext=%d%d%d in=%zu out=%zu", path, M, N, K, nz, rm.G, g.tap_len, g.act,
                 g.residual != nullptr, g.act == S2S_ACT_SNAKE, g.res_c != nullptr, g.C2 != nullptr, sizeof(T), sizeof(CT));
        std::lock_guard<std::mutex> lk(mu);
        if (seen.insert(key).second) fprintf(stderr, "[s2s] %s\n", key);
    }
    if (M <= 128 && !(g.glu && M > 64) && !ext) {
        const int MT = cdiv(M, 16);
        int NF = 1;
        if (MT <= 4) {
            if (g.glu) NF = N >= 4096 ? 2 : 1;
            else NF = N >= 8192 ? 4 : (N >= 2048 ? 2 : 1);
        }
        const int slabs = cdiv(N, 16 * NF);
        const int ksplit = 1;   // split-K (f32 atomics on the residual) was removed: not run-to-run deterministic
        dim3 grid(slabs * ksplit, nz);
        ProfScope prof(st, S2S_PROF_GEMM_SKINNY, 2.0 * M * N * K * nz * (g.glu ? 2 : 1),
                       ((double)N * K * sizeof(T) * (g.glu ? 2 : 1) + (double)M * K * sizeof(T) + (double)M * N * sizeof(CT)) * nz);
#define S2S_SK1(MTV, NFV, GLUV, NORMV)                                                                                        \
    launch_k(gemm_skinny_kernel<T, CT, MTV, NFV, GLUV, NORMV>, grid, dim3(512), 0, st, (const T*)g.A, g.lda, (const T*)g.W, g.bias, g.residual, \
             (CT*)g.C, g.ldc, M, N, K, g.act, rm, ksplit, g.rms_eps)
#define S2S_SK(MTV, NFV, GLUV)                                           \
    do {                                                                 \
        if constexpr (sizeof(T) == 2) {                                  \
            if (g.a_f32_rms) { S2S_SK1(MTV, NFV, GLUV, true); break; }   \
        }                                                                \
        S2S_SK1(MTV, NFV, GLUV, false);                                  \
    } while (0)
#define S2S_SK_MT(NFV, GLUV)                  \
    switch (MT) {                             \
        case 1: S2S_SK(1, NFV, GLUV); break;  \
        case 2: S2S_SK(2, NFV, GLUV); break;  \
        case 3: S2S_SK(3, NFV, GLUV); break;  \
        default: S2S_SK(4, NFV, GLUV); break; \
    }
        if (MT > 4) {
            switch (MT) {
                case 5: S2S_SK(5, 1, false); break;
                case 6: S2S_SK(6, 1, false); break;
                case 7: S2S_SK(7, 1, false); break;
                default: S2S_SK(8, 1, false); break;
            }
        } else if (g.glu) {
            if (NF == 2) { S2S_SK_MT(2, true) } else { S2S_SK_MT(1, true) }
        } else if (NF == 4) {
            S2S_SK_MT(4, false)
        } else if (NF == 2) {
            S2S_SK_MT(2, false)
        } else {
            S2S_SK_MT(1, false)
        }
#undef S2S_SK_MT
#undef S2S_SK
#undef S2S_SK1
    } else if (conv7_ok<T>(g, rm, M, N, K, nz)) {
        if constexpr (sizeof(T) == 2) {
            const int Cin = g.tap_len, dil = (int)(g.tap_stride / g.tap_len);
            const int tiles_m = M / 256, tiles_n = cdiv(N, 128);
            static bool attr7[64] = {};
            int dev = 0;
            S2S_HIP_CHECK(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64 || !attr7[dev]) {
                S2S_HIP_CHECK(hipFuncSetAttribute((const void*)conv7_tile_kernel<CT, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, CV7_LDS));
                S2S_HIP_CHECK(hipFuncSetAttribute((const void*)conv7_tile_kernel<CT, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, CV7_LDS));
                if (dev >= 0 && dev < 64) attr7[dev] = true;
            }
            ProfScope prof(st, S2S_PROF_GEMM_TILE, 2.0 * M * N * K, (double)N * K * sizeof(T) + (double)M * Cin * sizeof(T) + (double)M * N * sizeof(CT));
            if (Cin % 64 == 0)
                launch_k(conv7_tile_kernel<CT, 64>, dim3(tiles_m * tiles_n), dim3(512), CV7_LDS, st, (const __hip_bfloat16*)g.A, (const __hip_bfloat16*)g.W, g.bias,
                         (CT*)g.C, g.ldc, M, N, Cin, dil, g.act, tiles_m, rm);
            else
                launch_k(conv7_tile_kernel<CT, 32>, dim3(tiles_m * tiles_n), dim3(512), CV7_LDS, st, (const __hip_bfloat16*)g.A, (const __hip_bfloat16*)g.W, g.bias,
                         (CT*)g.C, g.ldc, M, N, Cin, dil, g.act, tiles_m, rm);
        }
    } else if (tile5_ok<T>(g, M, N, K, nz, ext)) {
        if constexpr (sizeof(T) == 2) {
            const int tiles_m = cdiv(M, T5_M), tiles_n = cdiv(N, T5_N);
            static bool attr5[64] = {};
            int dev = 0;
            S2S_HIP_CHECK(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64 || !attr5[dev]) {
                S2S_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tile5_kernel<CT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, T5_LDS));
                S2S_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tile5_kernel<CT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, T5_LDS));
                S2S_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tile5_kernel<CT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, T5_LDS));
                S2S_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tile5_kernel<CT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, T5_LDS));
                S2S_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tile5_kernel<CT, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, T5_LDS));
                S2S_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tile5_kernel<CT, 28>, hipFuncAttributeMaxDynamicSharedMemorySize, T5_LDS));
                if (dev >= 0 && dev < 64) attr5[dev] = true;
            }
            ProfScope prof(st, S2S_PROF_GEMM_TILE, 2.0 * M * N * K * nz,
                           ((double)N * K * sizeof(T) + (double)M * K * sizeof(T) + (double)M * N * sizeof(CT)) * nz);
            const int var5 = tile5_var();
            const int tiles = tiles_m * tiles_n;
            unsigned long long* dbg = tile5_dbg();
            // one tile period on the 100 MHz counter: ~1.65 us per 64-wide K tile + the epilogue (S2S_TILE5_SPREAD: percent of it)
            const int spread = (int)((long)(K / T5_BK * 165 + 600) * tile5_spread_pct() / 100);
#define S2S_T5_LAUNCH(V, GRID)                                                                                            \
    launch_k(gemm_tile5_kernel<CT, V>, dim3(GRID, nz), dim3(512), T5_LDS, st, (const T*)g.A, g.lda, (const T*)g.W, g.bias, \
             g.residual, (CT*)g.C, g.ldc, M, N, K, g.act, tiles_m, tiles, rm, dbg, spread)
            if (var5 == 1) S2S_T5_LAUNCH(1, tiles);
            else if (var5 == 2) S2S_T5_LAUNCH(2, tiles);
            else if (var5 == 4) S2S_T5_LAUNCH(4, min(tiles, tile5_persist_wgs()));
            else if (var5 == 12) S2S_T5_LAUNCH(12, min(tiles, tile5_persist_wgs()));
            else if (var5 == 28) S2S_T5_LAUNCH(28, min(tiles, tile5_persist_wgs()));
            else S2S_T5_LAUNCH(0, tiles);
#undef S2S_T5_LAUNCH
        }
    } else if (tile4_ok<T>(g, M, N, K, nz, ext)) {
        if constexpr (sizeof(T) == 2) {
            const int tiles_m = cdiv(M, T4_M), tiles_n = cdiv(N, T4_N);
            constexpr int lds_bytes = T4_NST * T4_STAGE;
            static bool attr4[64] = {};
            int dev = 0;
            S2S_HIP_CHECK(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64 || !attr4[dev]) {
                S2S_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tile4_kernel<CT, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
                if (dev >= 0 && dev < 64) attr4[dev] = true;
            }
            ProfScope prof(st, S2S_PROF_GEMM_TILE, 2.0 * M * N * K * nz,
                           ((double)N * K * sizeof(T) + (double)M * K * sizeof(T) + (double)M * N * sizeof(CT)) * nz);
            launch_k(gemm_tile4_kernel<CT, 4>, dim3(tiles_m * tiles_n, nz), dim3(512), lds_bytes, st,
                     (const T*)g.A, g.lda, (const T*)g.W, g.bias, g.residual, (CT*)g.C, g.ldc, M, N, K, g.act, tiles_m, rm);
        }
    } else if (constexpr int BK3 = ROWB / (int)sizeof(T);
               (g.tap_len == 0 || g.tap_len >= BK3) && K >= 3 * BK3 && (long)cdiv(M, T3_M) * cdiv(N, T3_N) * nz >= 192 && !tile3_disabled()) {
        const int tiles_m = cdiv(M, T3_M), tiles_n = cdiv(N, T3_N);
        constexpr int lds_bytes = T3_NST * T3_STAGE;
        const bool gen = g.tap_len > 0 || K % BK3 != 0;
        // the dynamic-LDS limit is a per-DEVICE function attribute: one flag per device of this process (a pool per GPU may
        // live in one process); per (T, CT) instantiation; benign race (idempotent)
        static bool attr_set[64] = {};
        int dev = 0;
        S2S_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            S2S_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tile3_kernel<T, CT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
            S2S_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tile3_kernel<T, CT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
        ProfScope prof(st, S2S_PROF_GEMM_TILE, 2.0 * M * N * K * nz,
                       ((double)N * K * sizeof(T) + (double)M * K * sizeof(T) + (double)M * N * sizeof(CT)) * nz);
        if (gen)
            launch_k(gemm_tile3_kernel<T, CT, true>, dim3(tiles_m * tiles_n, nz), dim3(512), lds_bytes, st,
                     (const T*)g.A, g.lda, (const T*)g.W, g.bias, g.residual, (CT*)g.C, g.ldc, M, N, K, g.act, tiles_m, rm);
        else
            launch_k(gemm_tile3_kernel<T, CT, false>, dim3(tiles_m * tiles_n, nz), dim3(512), lds_bytes, st,
                     (const T*)g.A, g.lda, (const T*)g.W, g.bias, g.residual, (CT*)g.C, g.ldc, M, N, K, g.act, tiles_m, rm);
    } else if (constexpr int BK3 = ROWB / (int)sizeof(T); (g.tap_len == 0 || g.tap_len >= BK3) && K >= 3 * BK3 && tile3_mid_enabled() && !tile3_disabled()) {
        // mid-sized launches (fewer than 192 tiles of 256 x 128: at most one 128 x 128 tile per CU, the launch takes as long as ONE tile): the
        // 4-stage 128 x 128 instantiation keeps three K slabs in flight where gemm_tile_kernel drains its only one before every k-step
        // (FastConformer projections at live-partial batch sizes: 19 -> 9 us at K = 1024, 56 -> 26 us at K = 4096, profiles/r06_bench_mid_gemm.txt)
        const int tiles_m = cdiv(M, 128), tiles_n = cdiv(N, T3_N);
        constexpr int lds_bytes = 4 * (128 * ROWB + T3_W_BYTES);
        const bool gen = g.tap_len > 0 || K % BK3 != 0;
        static bool attr_mid[64] = {};
        int dev = 0;
        S2S_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr_mid[dev]) {
            S2S_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tile3_kernel<T, CT, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
            S2S_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_tile3_kernel<T, CT, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
            if (dev >= 0 && dev < 64) attr_mid[dev] = true;
        }
        ProfScope prof(st, S2S_PROF_GEMM_TILE, 2.0 * M * N * K * nz,
                       ((double)N * K * sizeof(T) + (double)M * K * sizeof(T) + (double)M * N * sizeof(CT)) * nz);
        if (gen)
            launch_k(gemm_tile3_kernel<T, CT, true, 2>, dim3(tiles_m * tiles_n, nz), dim3(256), lds_bytes, st,
                     (const T*)g.A, g.lda, (const T*)g.W, g.bias, g.residual, (CT*)g.C, g.ldc, M, N, K, g.act, tiles_m, rm);
        else
            launch_k(gemm_tile3_kernel<T, CT, false, 2>, dim3(tiles_m * tiles_n, nz), dim3(256), lds_bytes, st,
                     (const T*)g.A, g.lda, (const T*)g.W, g.bias, g.residual, (CT*)g.C, g.ldc, M, N, K, g.act, tiles_m, rm);
    } else {
        const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
        ProfScope prof(st, S2S_PROF_GEMM_TILE, 2.0 * M * N * K * nz,
                       ((double)N * K * sizeof(T) + (double)M * K * sizeof(T) + (double)M * N * sizeof(CT)) * nz);
        launch_k(gemm_tile_kernel<T, CT>, dim3(tiles_m * tiles_n, nz), dim3(256), 0, st,
                 (const T*)g.A, g.lda, (const T*)g.W, g.bias, g.residual, (CT*)g.C, g.ldc, M, N, K, g.act, tiles_m, rm);
    }
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int gemm_ex(hipStream_t st, const GemmArgs& g) {
    S2S_REQUIRE(g.M >= 0 && g.N > 0 && g.K > 0, "gemm: bad shape M=%d N=%d K=%d", g.M, g.N, g.K);
    if (g.M == 0) return S2S_OK;
    S2S_REQUIRE(g.A && g.W && g.C, "gemm: null operand");
    S2S_REQUIRE(!g.glu || g.M <= 64, "gemm: fused SwiGLU epilogue is a decode-path (M <= 64) feature");
    const int epc = g.ab_dtype == S2S_BF16 ? 8 : 4;
    S2S_REQUIRE(g.K % epc == 0 && g.lda % epc == 0 && g.a_bs % epc == 0,
                "gemm: K=%d / lda=%ld / a_bs=%ld must be multiples of %d", g.K, g.lda, g.a_bs, epc);
    S2S_REQUIRE(((uintptr_t)g.A & 15) == 0 && ((uintptr_t)g.W & 15) == 0, "gemm: operands must be 16-byte aligned");
    S2S_REQUIRE(g.tap_len % epc == 0 && g.tap_stride % epc == 0, "gemm: tap_len / tap_stride must be multiples of %d", epc);
    S2S_REQUIRE(g.ldw % epc == 0 && g.a_z1 % epc == 0 && g.a_z2 % epc == 0 && g.w_z1 % epc == 0 && g.w_z2 % epc == 0,
                "gemm: ldw / batch strides must be multiples of %d elements", epc);
    if (g.act == S2S_ACT_SNAKE || g.res_c || g.C2) {
        S2S_REQUIRE(g.ab_dtype == g.c_dtype && !g.glu && !g.a_f32_rms && !g.alpha_set, "gemm: the snake / model-dtype residual / second-output epilogues need C in the operand dtype");
        S2S_REQUIRE(g.act != S2S_ACT_SNAKE || (g.snake_a && g.snake_ib), "gemm: S2S_ACT_SNAKE without tables");
        S2S_REQUIRE(!g.C2 || (g.snake2_a && g.snake2_ib), "gemm: second output without snake tables");
        S2S_REQUIRE(g.C2 != g.A && g.C != g.A, "gemm: outputs must not alias A");
    }
    if (g.a_f32_rms)
        S2S_REQUIRE(g.ab_dtype == S2S_BF16 && g.M <= 64 && g.G <= 0 && g.tap_len == 0 && g.nz1 <= 1 && g.nz2 <= 1 && g.rms_eps > 0.f &&
                        g.residual == nullptr,
                    "gemm: the RMSNorm-fused path is a bf16 decode (M <= 64) feature over plain f32 rows");
    if (g.ab_dtype == S2S_BF16 && g.c_dtype == S2S_BF16) return launch_gemm<__hip_bfloat16, __hip_bfloat16>(st, g);
    if (g.ab_dtype == S2S_BF16 && g.c_dtype == S2S_F32) return launch_gemm<__hip_bfloat16, float>(st, g);
    if (g.ab_dtype == S2S_F32 && g.c_dtype == S2S_F32) return launch_gemm<float, float>(st, g);
    return fail(S2S_ERR_UNSUPPORTED, "gemm: unsupported dtype combination ab=%d c=%d", g.ab_dtype, g.c_dtype);
}

int gemm(hipStream_t st, const void* A, long lda, const void* W, const float* bias,
         const float* residual, void* C, long ldc, int M, int N, int K, int ab_dtype, int c_dtype,
         int act, bool allow_splitk) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.bias = bias; g.residual = residual; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.G = 0; g.ab_dtype = ab_dtype; g.c_dtype = c_dtype; g.act = act;
    g.allow_splitk = allow_splitk;
    return gemm_ex(st, g);
}

}  // namespace s2s

extern "C" int32_t s2s_op_gemm(void* stream, const void* A, int64_t lda, const void* W, const float* bias,
                               const float* residual, void* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                               int32_t ab_dtype, int32_t c_dtype, int32_t act) {
    return s2s::gemm((hipStream_t)stream, A, (long)lda, W, bias, residual, C, (long)ldc, M, N, K, ab_dtype,
                     c_dtype, act, /*allow_splitk=*/false);
}

/* s2s_op_gemm with the decode-path switches exposed (parity tests of the fused-SwiGLU / split-K / in-kernel RMSNorm variants):
 * flags bit 0: W = [gate rows | up rows], C = silu(gate) * up (M <= 64); bit 1: allow split-K (in-place f32 residual only);
 * bit 2: A is the raw f32 residual stream, RMSNorm (weight folded into W) runs inside the kernel with rms_eps. */
extern "C" int32_t s2s_op_gemm_ex(void* stream, const void* A, int64_t lda, const void* W, const float* bias,
                                  const float* residual, void* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                                  int32_t ab_dtype, int32_t c_dtype, int32_t act, int32_t flags, float rms_eps) {
    s2s::GemmArgs g{};
    g.A = A; g.lda = (long)lda; g.W = W; g.bias = bias; g.residual = residual; g.C = C; g.ldc = (long)ldc;
    g.M = M; g.N = N; g.K = K; g.G = 0; g.ab_dtype = ab_dtype; g.c_dtype = c_dtype; g.act = act;
    g.glu = (flags & 1) != 0; g.allow_splitk = (flags & 2) != 0; g.a_f32_rms = (flags & 4) != 0; g.rms_eps = rms_eps;
    return s2s::gemm_ex((hipStream_t)stream, g);
}

/* 1-D convolution over a zero-padded time-major buffer, as one GEMM (no im2col):
 * x: [B][T_in + 2*pad][Cin] with `pad` zero rows on both sides of every item;
 * w: [Cout][ksize][Cin]; y: [B][y_rows][Cout], output frame t written at row t + y_row0. */
extern "C" int32_t s2s_op_conv1d_tm(void* stream, const void* x, const void* w, const float* bias, void* y,
                                    int32_t B, int32_t T_in, int32_t Cin, int32_t Cout, int32_t ksize,
                                    int32_t stride, int32_t pad, int32_t y_rows, int32_t y_row0,
                                    int32_t ab_dtype, int32_t c_dtype, int32_t act) {
    S2S_REQUIRE(B > 0 && T_in > 0 && ksize > 0 && stride > 0 && pad >= 0, "conv1d: bad arguments");
    const int T_out = (T_in + 2 * pad - ksize) / stride + 1;
    S2S_REQUIRE(y_row0 >= 0 && y_row0 + T_out <= y_rows, "conv1d: output rows out of range");
    s2s::GemmArgs g{};
    g.A = x; g.lda = (long)stride * Cin; g.a_bs = (long)(T_in + 2 * pad) * Cin;
    g.W = w; g.bias = bias;
    const size_t esz = c_dtype == S2S_F32 ? 4 : 2;
    g.C = (char*)y + (size_t)y_row0 * Cout * esz; g.ldc = Cout; g.c_bs = (long)y_rows * Cout;
    g.M = B * T_out; g.N = Cout; g.K = ksize * Cin; g.G = T_out;
    g.ab_dtype = ab_dtype; g.c_dtype = c_dtype; g.act = act;
    return s2s::gemm_ex((hipStream_t)stream, g);
}

/* Causal dilated 1-D convolution over a history-padded time-major buffer, as one GEMM with tap addressing (the codec
 * decoder's residual-unit convolution):  x: [B][lpad + T][Cin], the lpad >= (ksize-1)*dilation rows before every item hold
 * its history (zeros at utterance start);  w: [Cout][ksize][Cin];  y: [B][T][Cout],
 * y[b][t] = bias + sum_j w[:, j, :] . x[b][lpad + t - (ksize-1-j)*dilation]. */
extern "C" int32_t s2s_op_conv1d_causal_tm(void* stream, const void* x, const void* w, const float* bias, void* y,
                                           int32_t B, int32_t T, int32_t Cin, int32_t Cout, int32_t ksize,
                                           int32_t dilation, int32_t lpad, int32_t ab_dtype, int32_t c_dtype, int32_t act) {
    S2S_REQUIRE(B > 0 && T > 0 && ksize > 0 && dilation > 0 && lpad >= (ksize - 1) * dilation, "conv1d_causal: bad arguments");
    s2s::GemmArgs g{};
    const size_t e = ab_dtype == S2S_F32 ? 4 : 2;
    g.A = (const char*)x + (size_t)(lpad - (ksize - 1) * dilation) * Cin * e; g.lda = Cin; g.a_bs = (long)(lpad + T) * Cin;
    g.W = w; g.bias = bias;
    g.C = y; g.ldc = Cout; g.c_bs = (long)T * Cout;
    g.M = B * T; g.N = Cout; g.K = ksize * Cin; g.G = T;
    g.ab_dtype = ab_dtype; g.c_dtype = c_dtype; g.act = act;
    if (ksize > 1) { g.tap_len = Cin; g.tap_stride = (long)dilation * Cin; }
    return s2s::gemm_ex((hipStream_t)stream, g);
}

/* Development hook (include/s2s_mi355x_dev.h): phase timestamps of every following gemm_tile5_kernel launch, 6 x u64 per tile, or null. */
extern "C" int32_t s2s_op_gemm_tile_debug(void* dbg_dev) {
    s2s::tile5_set_debug(reinterpret_cast<unsigned long long*>(dbg_dev));
    return S2S_OK;
}

/* development: the k = 7 convolutions of the codec on the tap-reuse kernel (1, default) or on the tap-by-tap GEMM of rounds 1 - 5 (0); -1 = default */
extern "C" int32_t s2s_op_conv7_tune(int32_t on) {
    s2s::conv7_set(on);
    return S2S_OK;
}
