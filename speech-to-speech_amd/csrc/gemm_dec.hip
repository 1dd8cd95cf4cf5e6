// Decode-step projections for gfx950 (bf16, M <= 128 rows = one row per live session).
//
// One decoder layer of a decode step is FIVE launches on this path (qkv | attention | out-proj | gate/up | down) instead
// of seven: the two RMSNorm launches are gone.  The kernel that writes the residual stream (out-proj / down-proj
// epilogue) also emits (a) a bf16 copy of the new residual rows and (b) per-column-slab partial sums of squares; the
// consuming projection multiplies the RAW bf16 rows with weights that have the norm's gamma folded into their columns
// and scales its accumulators by rsqrt(mean(x^2) + eps) in the epilogue (the partials are summed in a fixed order:
// deterministic, no atomics).
//
// Operand layout ("fragment-major"): a fragment-shaped load from a row-major matrix makes every 16-lane quad touch 16
// different rows - 16 tag look-ups per quad instead of 2 - and the texture-address unit, not HBM, bounds the kernel
// (measured in round 2: 12.7 us for the 16.8 MB qkv projection, = 64 cycles x 5 loads x 8 steps x 8 waves per CU).
// Decode weights are therefore stored pre-swizzled, one contiguous 1 KiB block per (16-row tile, 32-wide k-step) in
// exactly the MFMA operand order [k-chunk lg][row l15][8 elements]; a wave's whole K-slice of a tile is one contiguous
// run (8 KiB at K = 2048) and every wave load is a single fully coalesced 1 KiB request with a wave-uniform (SGPR)
// base address.  Activations that only ever feed these projections (the bf16 residual copy, the SwiGLU output, the
// decode-attention output) are written in the same layout by their producers (a_frag / c_frag); row-major activations
// are still accepted (template AF = false).
//
// Work split: one 8-wave workgroup owns 16*NF output columns over the whole of K; the waves split K, every weight
// element is read exactly once straight from HBM into MFMA B fragments (non-temporal, 16 B per lane), the activation
// fragments come from L2.  Loads are software-pipelined in two register stages of UC k-steps (32 k each), so a stage's
// worth of weight and activation loads is always in flight behind the MFMAs.  The 8 partial tiles are combined
// through LDS in wave order (deterministic) and the whole epilogue (bias, activation, SwiGLU, RMS scale, residual add,
// bf16 copy, sum of squares) runs on the combined tile with one row's 16 columns in 16 adjacent lanes (64-byte f32 /
// 32-byte bf16 row segments per store).
//
// Replaces, per decode step, the rmsnorm -> gemm pairs of the Qwen3 decoder layer as restated in oracle/qwen3_tts.py
// (transformers qwen3/modeling_qwen3.py decoder layer; qwen3_omni_moe code predictor :2250-2620).
#include "ops.h"

namespace s2s {

namespace {

constexpr int DEC_NW = 8;            // waves per workgroup

// element (row m, column k) of a fragment-major [rows][K] matrix with ksteps = K / 32
__host__ __device__ __forceinline__ long frag_index(int m, int k, int ksteps) {
    return ((((long)(m >> 4) * ksteps + (k >> 5)) * 4 + ((k >> 3) & 3)) * 16 + (m & 15)) * 8 + (k & 7);
}

// a pointer the compiler must keep in SGPRs (wave-uniform by construction): loads become  saddr + 32-bit lane offset
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ u32x4_t ld_nt16(const void* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
}

// MODE 0: C bf16 = act(rstd * acc + bias); GLU: C = silu(rstd*gate) * (rstd*up)
// MODE 1: C f32  = act(rstd * acc + bias)
// MODE 2: X f32 (in place) += rstd * acc + bias; XB = bf16(X); ss_out[slab][m] = sum over the slab's columns of X^2
template <int MT, int NF, bool GLU, int MODE, int UC, bool AF, bool NT>
__global__ __launch_bounds__(DEC_NW * 64) void gemm_dec_kernel(DecGemm a) {
    constexpr int NS = NF * (GLU ? 2 : 1);       // weight streams per wave
    constexpr int GRMAX = GLU ? 2 : 4;
    constexpr int GR = (MT < GRMAX ? MT : GRMAX);   // row fragments combined per LDS pass (32 KiB of LDS)
    constexpr int PASSES = (MT + GR - 1) / GR;
    __shared__ float red[DEC_NW][GR * (GLU ? 2 : 1)][4][64];
    __shared__ float ssp[DEC_NW][MT * 16];
    __shared__ float rstd_s[MT * 16];
    __shared__ float mean_s[MT * 16];

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: k-slice bounds live in SGPRs
    if (gridDim.y > 1) {
        // M > 128: blockIdx.y walks slabs of 128 rows (M % 128 == 0); every slab re-streams the weights (L2 / MALL hits after
        // the first) and is an independent M = 128 problem on shifted row pointers.
        const long r0 = (long)blockIdx.y * 128;
        a.A += r0 * (AF ? (long)a.K : a.lda);
        if constexpr (MODE == 0) a.C = reinterpret_cast<__hip_bfloat16*>(a.C) + r0 * a.ldc;
        else if constexpr (MODE == 1) a.C = reinterpret_cast<float*>(a.C) + r0 * a.ldc;
        else {
            a.X += r0 * a.ldx;
            if (a.XB) a.XB += r0 * a.ldxb;
            if (a.ss_out) a.ss_out += r0;
        }
        if (a.ss_in) a.ss_in += r0 * (a.ss_rs > 0 ? a.ss_rs : 1);
        a.M = 128;
    }
    const int n0 = blockIdx.x * 16 * NF;
    const int steps = a.K >> 5;                                  // k-steps of 32
    const int spw = (steps + DEC_NW - 1) / DEC_NW;
    const int s0 = wave * spw, s1 = min(steps, s0 + spw);

    // weights: tile (n / 16), step s -> 512-element block at ((tile * steps + s) * 512); lane l reads elements l*8 .. l*8+7
    // of it.  Tiles past N re-read the last valid tile (finite garbage in columns that are never stored).
    const int ntiles = (a.N + 15) >> 4;
    const __hip_bfloat16* wtile[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int t = min((n0 >> 4) + (s % NF), ntiles - 1) + (s >= NF ? ntiles : 0);
        wtile[s] = a.W + (long)t * steps * 512;
    }
    // activations: fragment-major (same block structure over 16-row tiles) or row-major [M][lda]
    const int mtiles = (a.M + 15) >> 4;
    const __hip_bfloat16* arow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        if constexpr (AF) {
            arow[i] = a.A + (long)min(i, mtiles - 1) * steps * 512;
        } else {
            const int m = min(i * 16 + l15, a.M - 1);
            arow[i] = a.A + (long)m * a.lda + lg * 8;
        }
    }
    f32x4_t acc[MT][NS];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[i][s] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    u32x4_t fw[2][UC][NS], fa[2][UC][MT];
    int kbase[2], sfrom[2];
    // A stage is UC consecutive k-steps read with ONE address per stream plus immediate offsets.  A stage that would run
    // past the end of K is shifted back inside the row (kb < sb); the steps it then holds twice, and the steps past the
    // wave's slice, are zeroed with an AND mask on the narrower operand at USE time.  Every load is unconditional and
    // nothing consumes a loaded value before its MFMA, so the compiler's vmcnt accounting stays exact and a whole stage
    // remains in flight behind the running one.
    const int kmax = max(steps - UC, 0);
    auto load = [&](int st, int sb) {
        const int kb = min(sb, kmax);
        kbase[st] = kb;
        sfrom[st] = sb;
        const unsigned loff = lane * 16;                   // byte offset of this lane inside a 1 KiB fragment block
        const char* wb[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) wb[q] = uniform_ptr(wtile[q] + (long)kb * 512);
#pragma unroll
        for (int u = 0; u < UC; ++u)
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                if constexpr (NT) fw[st][u][q] = ld_nt16(wb[q] + u * 1024 + loff);      // streamed once: do not displace cached lines
                else fw[st][u][q] = *reinterpret_cast<const u32x4_t*>(wb[q] + u * 1024 + loff);   // re-read every step: keep in L2 / MALL
            }
        if constexpr (AF) {
            const char* ab[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) ab[i] = uniform_ptr(arow[i] + (long)kb * 512);
#pragma unroll
            for (int u = 0; u < UC; ++u)
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[st][u][i] = *reinterpret_cast<const u32x4_t*>(ab[i] + u * 1024 + loff);
        } else {
#pragma unroll
            for (int u = 0; u < UC; ++u)
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[st][u][i] = *reinterpret_cast<const u32x4_t*>(arow[i] + (long)(kb + u) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);      // the stage is issued HERE: the scheduler must not sink loads behind MFMAs
    };
    auto compute = [&](int st) {
#pragma unroll
        for (int u = 0; u < UC; ++u) {
            const int ks = kbase[st] + u;
            const unsigned keep = (ks >= sfrom[st] && ks < s1) ? 0xffffffffu : 0u;     // wave-uniform
            if constexpr (NS <= MT) {
#pragma unroll
                for (int q = 0; q < NS; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) fw[st][u][q][e] &= keep;
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[st][u][i][e] &= keep;
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[st][u][i]),
                                                                        __builtin_bit_cast(bf16x8_t, fw[st][u][q]), acc[i][q], 0, 0, 0);
        }
    };

    // RMS prologue: this wave's share of the input rows' partial sums of squares (parts p = wave, wave + 8, ...).  The
    // loads are issued BEFORE the weight stream and reduced right after stage 0 is in flight: loads return in order, so
    // waiting for these (oldest, L2-resident) values never waits for the HBM stream behind them.
    // prefetched parts per wave (the rest: tail loop); one weight stream leaves room for 20 x 8 waves = the 160 pair-partials of a
    // 1280-wide LayerNorm input
    constexpr int RH = (MT * 16 + 63) / 64, RP = MT > 4 ? 8 : (NS > 1 ? 16 : 20);
    float sv[RH][RP];
    const int ssrs = a.ss_rs > 0 ? a.ss_rs : 1;
    if (a.ss_in) {
#pragma unroll
        for (int h = 0; h < RH; ++h) {
            const int m = min(h * 64 + lane, a.M - 1);
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const int p = wave + j * DEC_NW;
                sv[h][j] = a.ss_in[(long)min(p, a.n_parts_in - 1) * a.ss_stride + m * ssrs];     // parts past the end: dropped below
            }
        }
    }
    auto rms_parts = [&]() {
        if (a.ss_in) {
#pragma unroll
            for (int h = 0; h < RH; ++h) {
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < RP; ++j) v += wave + j * DEC_NW < a.n_parts_in ? sv[h][j] : 0.f;
                for (int p = wave + RP * DEC_NW; p < a.n_parts_in; p += DEC_NW)      // more than 8 * RP parts: rare tail
                    v += a.ss_in[(long)p * a.ss_stride + min(h * 64 + lane, a.M - 1) * ssrs];
                if (h * 64 + lane < MT * 16) ssp[wave][h * 64 + lane] = v;
            }
        }
    };
    // Three straight-line schedules (separate code, so each one's wait counts and register assignment are exact).  When the
    // registers allow it, BOTH stages are issued before the row-scale partials are reduced (their wait would otherwise delay
    // the second stage's issue by an L2 round trip).
    constexpr bool EARLY = 2 * UC * (NS + MT) * 4 + MT * NS * 4 + RH * RP <= 216;
    if (s1 - s0 <= UC) {                 // short slice (e.g. K = 1024): a single stage holds it, nothing is re-read
        load(0, s0);
        rms_parts();
        compute(0);
    } else if (s1 - s0 <= 2 * UC) {      // the whole slice is in flight at once (K = 2048)
        load(0, s0);
        if constexpr (!EARLY) rms_parts();
        load(1, s0 + UC);
        if constexpr (EARLY) rms_parts();
        compute(0);
        compute(1);
    } else {                             // long slice: two stages in flight, refilled behind the MFMAs
        load(0, s0);
        if constexpr (!EARLY) rms_parts();
        load(1, s0 + UC);
        if constexpr (EARLY) rms_parts();
        int sb = s0;
        do {
            compute(0);
            load(0, sb + 2 * UC);
            compute(1);
            load(1, sb + 3 * UC);
            sb += 2 * UC;
        } while (sb + 2 * UC < s1);
        compute(0);
        compute(1);
    }

    // ---- combine the 8 K-slices, then the epilogue on the combined tile ------------------------------------------
    bool need_rstd = a.ss_in != nullptr;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            if (f > 0 || ps > 0) __syncthreads();
#pragma unroll
            for (int g = 0; g < GR; ++g) {
                const int i = ps * GR + g;
                if (i < MT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        red[wave][g][r][lane] = acc[i][f][r];
                        if constexpr (GLU) red[wave][GR + g][r][lane] = acc[i][NF + f][r];
                    }
                }
            }
            __syncthreads();
            if (need_rstd) {                        // first pass only: finish the row scales
                if (tid < MT * 16) {
                    if (a.ln) {      // pairs: wave w reduced the parts w, w + 8, ... - all even (sums of squares) or all odd (sums)
                        const float sq = (ssp[0][tid] + ssp[2][tid]) + (ssp[4][tid] + ssp[6][tid]);
                        const float sm = (ssp[1][tid] + ssp[3][tid]) + (ssp[5][tid] + ssp[7][tid]);
                        const float mean = sm * a.inv_dim;
                        mean_s[tid] = mean;
                        rstd_s[tid] = rsqrtf(fmaxf(sq * a.inv_dim - mean * mean, 0.f) + a.eps);
                    } else {
                        float v = 0.f;
#pragma unroll
                        for (int w = 0; w < DEC_NW; ++w) v += ssp[w][tid];
                        rstd_s[tid] = rsqrtf(v * a.inv_dim + a.eps);
                    }
                }
                need_rstd = false;
                __syncthreads();
            }
            // element e -> local row e >> 4 (0 .. GR*16), column e & 15: one row's 16 columns sit in 16 adjacent lanes
#pragma unroll
            for (int e0 = 0; e0 < GR * 256; e0 += DEC_NW * 64) {
                const int e = e0 + tid;
                const int ml = e >> 4, col = e & 15;
                const int g = ml >> 4, r = ml & 3, l = ((ml & 15) >> 2) * 16 + col;
                const int m = ps * GR * 16 + ml, nn = n0 + f * 16 + col;
                const bool ok = (GR * 256 >= DEC_NW * 64 || e < GR * 256) && m < a.M && nn < a.N;
                float v = 0.f, u2 = 0.f;
                if (e < GR * 256) {
#pragma unroll
                    for (int w = 0; w < DEC_NW; ++w) v += red[w][g][r][l];
                    if constexpr (GLU) {
#pragma unroll
                        for (int w = 0; w < DEC_NW; ++w) u2 += red[w][GR + g][r][l];
                    }
                }
                const float rs = (a.ss_in && m < MT * 16) ? rstd_s[m] : 1.0f;
                if (a.ln && ok) v -= mean_s[m] * a.csum[nn];      // (x - mean) W'^T = x W'^T - mean * rowsum(W')
                v *= rs;
                if constexpr (GLU) {
                    u2 *= rs;
                    v = v / (1.0f + __expf(-v)) * u2;
                } else {
                    if (a.bias && ok) v += a.bias[nn];
                    if (a.act == S2S_ACT_GELU) v = gelu_erf(v);
                    else if (a.act == S2S_ACT_SILU) v = v / (1.0f + __expf(-v));
                    else if (a.act == S2S_ACT_RELU) v = fmaxf(v, 0.f);
                }
                if constexpr (MODE == 0) {
                    if (ok) Elem<__hip_bfloat16>::st(reinterpret_cast<__hip_bfloat16*>(a.C) + (a.c_frag ? frag_index(m, nn, (int)(a.ldc >> 5)) : (long)m * a.ldc + nn), v);
                } else if constexpr (MODE == 1) {
                    if (ok) reinterpret_cast<float*>(a.C)[(long)m * a.ldc + nn] = v;
                } else {
                    float xn = 0.f;
                    if (ok) {
                        float* xp = a.X + (long)m * a.ldx + nn;
                        xn = a.x_overwrite ? v : *xp + v;
                        *xp = xn;
                        if (a.XB) Elem<__hip_bfloat16>::st(a.XB + (a.xb_frag ? frag_index(m, nn, (int)(a.ldxb >> 5)) : (long)m * a.ldxb + nn), xn);
                    }
                    if (a.ss_out) {
                        float q = xn * xn;
                        q += __shfl_xor(q, 1, 64);
                        q += __shfl_xor(q, 2, 64);
                        q += __shfl_xor(q, 4, 64);
                        q += __shfl_xor(q, 8, 64);
                        const long part = blockIdx.x * NF + f;
                        if (a.ss_pairs) {
                            float sx = xn;
                            sx += __shfl_xor(sx, 1, 64);
                            sx += __shfl_xor(sx, 2, 64);
                            sx += __shfl_xor(sx, 4, 64);
                            sx += __shfl_xor(sx, 8, 64);
                            if (col == 0 && m < a.M && e < GR * 256) {
                                a.ss_out[2 * part * a.ss_stride + m] = q;
                                a.ss_out[(2 * part + 1) * a.ss_stride + m] = sx;
                            }
                        } else if (col == 0 && m < a.M && e < GR * 256) a.ss_out[part * a.ss_stride + m] = q;
                    }
                }
            }
        }
    }
}

static bool dec_low_regs() {
    static const bool on = getenv("S2S_DEC_LOWREG") != nullptr;
    return on;
}

// loaded k-steps per wave for a slice of T steps with stages of UC: everything past T is re-read and masked
static inline int dec_loaded_steps(int T, int UC) { return T <= UC ? UC : (T + 2 * UC - 1) / (2 * UC) * (2 * UC); }

template <int MT, int NF, bool GLU, int MODE, bool AF, bool NT>
int launch_dec3(hipStream_t st, const DecGemm& a) {
    constexpr int NS = NF * (GLU ? 2 : 1);
    // register budget: two stages of UC * (NS + MT) 16-byte fragments + MT * NS accumulators
    constexpr int FR = NS + MT;      // 16-byte fragments per k-step
    constexpr int UC = FR <= 2 ? 8 : FR == 3 ? 6 : FR <= 6 ? 4 : FR <= 9 ? 2 : 1;
    constexpr int UC_ALT = FR == 2 ? 6 : (FR == 5 ? 3 : UC);      // second stage depth for slices UC does not divide (K = 3072)
    const int slabs = cdiv(a.N, 16 * NF);
    const int rslabs = a.M > 128 ? a.M / 128 : 1;
    const int T = cdiv(a.K >> 5, DEC_NW);
    // Shallow-stage variant (half the fragments in flight per wave, <= 128 VGPRs): two workgroups fit on a CU, so decode
    // kernels of DIFFERENT streams (TTS lanes, the STT worker) co-reside instead of splitting the CUs between them.
    constexpr int UC_LOW = GLU ? 1 : (UC >= 2 ? UC / 2 : 1);
    if constexpr (MT >= 4 && UC_LOW != UC) {
        if (dec_low_regs()) {
            launch_k(gemm_dec_kernel<MT, NF, GLU, MODE, UC_LOW, AF, NT>, dim3(slabs, rslabs), dim3(DEC_NW * 64), 0, st, a);
            S2S_LAUNCH_CHECK();
            return S2S_OK;
        }
    }
    if (UC_ALT != UC && dec_loaded_steps(T, UC_ALT) < dec_loaded_steps(T, UC))
        launch_k(gemm_dec_kernel<MT, NF, GLU, MODE, UC_ALT, AF, NT>, dim3(slabs, rslabs), dim3(DEC_NW * 64), 0, st, a);
    else
        launch_k(gemm_dec_kernel<MT, NF, GLU, MODE, UC, AF, NT>, dim3(slabs, rslabs), dim3(DEC_NW * 64), 0, st, a);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}
template <int MT, int NF, bool GLU, int MODE, bool AF>
int launch_dec2(hipStream_t st, const DecGemm& a) {
    return a.w_cached ? launch_dec3<MT, NF, GLU, MODE, AF, false>(st, a) : launch_dec3<MT, NF, GLU, MODE, AF, true>(st, a);
}
template <int MT, int NF, bool GLU, int MODE>
int launch_dec(hipStream_t st, const DecGemm& a) {
    return a.a_frag ? launch_dec2<MT, NF, GLU, MODE, true>(st, a) : launch_dec2<MT, NF, GLU, MODE, false>(st, a);
}

template <int MT>
int launch_dec_mt(hipStream_t st, const DecGemm& a, int NF) {
    constexpr bool NF2 = MT <= 4, NF2_GLU = MT <= 2;      // register budget (see launch_dec2)
    if (a.glu) {
        if constexpr (NF2_GLU) { if (NF == 2) return launch_dec<MT, 2, true, 0>(st, a); }
        return launch_dec<MT, 1, true, 0>(st, a);
    }
    if constexpr (NF2) {
        if (NF == 2) {
            if (a.mode == 0) return launch_dec<MT, 2, false, 0>(st, a);
            if (a.mode == 1) return launch_dec<MT, 2, false, 1>(st, a);
            return launch_dec<MT, 2, false, 2>(st, a);
        }
    }
    if (a.mode == 0) return launch_dec<MT, 1, false, 0>(st, a);
    if (a.mode == 1) return launch_dec<MT, 1, false, 1>(st, a);
    return launch_dec<MT, 1, false, 2>(st, a);
}

}  // namespace

int dec_parts(int N, int NF) { return cdiv(N, 16 * NF) * NF; }

int dec_out_parts(const DecGemm& a) {
    const int pw = decw_parts(a);
    const int p2 = pw > 0 ? pw : dec2_parts(a);
    return (p2 > 0 ? p2 : dec_parts(a.N, dec_nf(a))) * (a.ss_pairs ? 2 : 1);
}

int dec_nf(const DecGemm& a) {
    if (a.nf > 0) return a.nf;
    // enough workgroups to put one on every CU first; above that, two column fragments per workgroup halve the
    // activation re-reads from L2.  M > 64 keeps NF = 1 (register budget).
    if (a.M > 64 || (a.glu && a.M > 32)) return 1;
    return a.N >= 8192 ? 2 : 1;
}

int gemm_dec(hipStream_t st, const DecGemm& a) {
    S2S_REQUIRE(a.M >= 1 && (a.M <= 128 || a.M % 128 == 0 || decw_parts(a) > 0) && a.N > 0 && a.K > 0,
                "gemm_dec: bad shape M=%d N=%d K=%d (M <= 128, a multiple of 128, or fragment-major rows for the wide kernel)", a.M, a.N, a.K);
    S2S_REQUIRE(a.K >= 256 && a.K % 32 == 0 && (a.a_frag || a.lda % 8 == 0), "gemm_dec: K=%d must be >= 256 and a multiple of 32, lda a multiple of 8", a.K);
    S2S_REQUIRE((!a.c_frag || (a.mode == 0 && a.ldc % 32 == 0)) && (!a.xb_frag || a.ldxb % 32 == 0), "gemm_dec: fragment-major outputs need a row width that is a multiple of 32");
    S2S_REQUIRE(a.A && a.W && ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm_dec: operands must be 16-byte aligned");
    S2S_REQUIRE(!a.glu || (a.mode == 0 && !a.bias && a.act == S2S_ACT_NONE && a.N % 16 == 0), "gemm_dec: SwiGLU output is bf16 without bias / activation, N a multiple of 16");
    S2S_REQUIRE(a.mode == 2 ? (a.X != nullptr) : (a.C != nullptr), "gemm_dec: null output");
    S2S_REQUIRE(!a.ss_in || (a.n_parts_in > 0 && a.ss_stride >= a.M && a.inv_dim > 0.f), "gemm_dec: bad RMS prologue arguments");
    S2S_REQUIRE(!a.ln || (a.ss_in && a.csum && a.n_parts_in % 2 == 0 && !a.glu), "gemm_dec: the LayerNorm prologue needs pair partials, column sums and no SwiGLU");
    const int NF = dec_nf(a);
    ProfScope prof(st, S2S_PROF_GEMM_SKINNY, 2.0 * a.M * a.N * a.K * (a.glu ? 2 : 1),
                   (double)a.N * a.K * 2 * (a.glu ? 2 : 1) + (double)a.M * a.K * 2 + (double)a.M * a.N * (a.mode == 0 ? 2 : 4));
    // wide frame loops (129 .. 1024 rows, fragment-major activations): the row-block x column-block kernel (gemm_decw.hip)
    if (a.M > 64) {
        DecGemm w = a;
        if (!w.ws) w.ws = dec2_default_ws(&w.ws_bytes);
        if (const int rw = gemm_decw(st, w)) return rw < 0 ? rw : S2S_OK;
    }
    // 33 .. 128-row slabs with fragment-major activations: the column-block / K-slice kernel (gemm_dec2.hip)
    if (const int r2 = gemm_dec2(st, a)) return r2 < 0 ? r2 : S2S_OK;
    const int MT = cdiv(a.M, 16);
    if (MT <= 1) return launch_dec_mt<1>(st, a, NF);
    if (MT <= 2) return launch_dec_mt<2>(st, a, NF);
    if (MT <= 4) return launch_dec_mt<4>(st, a, NF);
    return launch_dec_mt<8>(st, a, 1);
}

// rows of the f32 residual stream -> bf16 copy (row-major or fragment-major) + one partial sum of squares per row (entry of
// a norm-free decode pass)
__global__ __launch_bounds__(256) void rows_prep_kernel(const float* __restrict__ x, long x_rs, __hip_bfloat16* __restrict__ xb, long xb_rs,
                                                        float* __restrict__ ss, int rows, int cols, int frag, long pair_stride) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * x_rs;
    float s = 0.f, t = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        t += (v.x + v.y) + (v.z + v.w);
        ushort4 p;
        p.x = f32_to_bf16_bits(v.x); p.y = f32_to_bf16_bits(v.y); p.z = f32_to_bf16_bits(v.z); p.w = f32_to_bf16_bits(v.w);
        *reinterpret_cast<ushort4*>(xb + (frag ? frag_index(row, c, cols >> 5) : (long)row * xb_rs + c)) = p;
    }
    s = wave_sum(s);
    if (lane == 0) ss[row] = s;
    if (pair_stride > 0) {
        t = wave_sum(t);
        if (lane == 0) ss[pair_stride + row] = t;
    }
}

int rows_prep(hipStream_t st, const float* x, long x_rs, void* xb, long xb_rs, float* ss, int rows, int cols, bool frag, long pair_stride) {
    S2S_REQUIRE(cols % 4 == 0 && x_rs % 4 == 0 && xb_rs % 4 == 0 && (!frag || cols % 32 == 0), "rows_prep: cols=%d unsupported", cols);
    if (rows == 0) return S2S_OK;
    rows_prep_kernel<<<cdiv(rows, 4), 256, 0, st>>>(x, x_rs, (__hip_bfloat16*)xb, xb_rs, ss, rows, cols, frag ? 1 : 0, pair_stride);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

// row-major bf16 matrix [rows][K] (row stride ld) -> fragment-major copy, rows padded with zeros to a multiple of 16;
// gamma (optional, [K]): fold a norm weight into the columns (out = W * gamma, rounded once to bf16)
__global__ void to_frag_kernel(const __hip_bfloat16* __restrict__ W, long ld, const float* __restrict__ gamma, __hip_bfloat16* __restrict__ out, int rows,
                               int rows_pad, int K) {
    const long chunk = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one 8-element chunk of the output
    const long total = (long)rows_pad * K / 8;
    if (chunk >= total) return;
    const int ksteps = K >> 5;
    const int l15 = (int)(chunk & 15), lg = (int)((chunk >> 4) & 3);
    const long blk = chunk >> 6;
    const int ks = (int)(blk % ksteps), tile = (int)(blk / ksteps);
    const int r = tile * 16 + l15, k0 = ks * 32 + lg * 8;
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float f = 0.f;
        if (r < rows) {
            f = Elem<__hip_bfloat16>::ld(W + (long)r * ld + k0 + e);
            if (gamma) f *= gamma[k0 + e];
        }
        v[e] = f32_to_bf16_bits(f);
    }
    u32x4_t pk;
#pragma unroll
    for (int q = 0; q < 4; ++q) pk[q] = (unsigned)v[2 * q] | ((unsigned)v[2 * q + 1] << 16);
    *reinterpret_cast<u32x4_t*>(out + chunk * 8) = pk;
}

int to_frag(hipStream_t st, const void* W, long ld, const float* gamma, void* out, int rows, int K) {
    S2S_REQUIRE(K % 32 == 0 && rows > 0, "to_frag: K=%d must be a multiple of 32", K);
    const int rows_pad = (rows + 15) / 16 * 16;
    const long total = (long)rows_pad * K / 8;
    to_frag_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>((const __hip_bfloat16*)W, ld, gamma, (__hip_bfloat16*)out, rows, rows_pad, K);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

}  // namespace s2s

/* Public op (parity tests / microbenchmarks): one decode-step projection.  See include/s2s_mi355x.h. */
static int32_t op_gemm_dec_impl(void* stream, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int32_t M, int32_t N,
                                int32_t K, const float* ss_in, int32_t n_parts_in, int32_t ss_stride, float inv_dim, float eps, int32_t mode,
                                void* C, int64_t ldc, int32_t act, float* X, int64_t ldx, void* XB, int64_t ldxb, float* ss_out, int32_t glu,
                                int32_t nf, int32_t layout, const float* csum) {
    S2S_REQUIRE(mode >= 0 && mode <= 2 && nf >= 0 && nf <= 2, "op_gemm_dec: bad mode / nf");
    s2s::DecGemm a{};
    a.ln = csum != nullptr; a.csum = csum; a.ss_pairs = (layout & 16) != 0;
    a.A = (const __hip_bfloat16*)A; a.lda = (long)lda; a.W = (const __hip_bfloat16*)W; a.ldw = (long)ldw; a.bias = bias;
    a.M = M; a.N = N; a.K = K;
    a.ss_in = ss_in; a.n_parts_in = n_parts_in; a.ss_stride = ss_stride; a.inv_dim = inv_dim; a.eps = eps;
    a.mode = mode; a.C = C; a.ldc = (long)ldc; a.act = act;
    a.X = X; a.ldx = (long)ldx; a.XB = (__hip_bfloat16*)XB; a.ldxb = (long)ldxb; a.ss_out = ss_out;
    a.glu = glu != 0; a.nf = nf;
    a.a_frag = (layout & 1) != 0; a.c_frag = (layout & 2) != 0; a.xb_frag = (layout & 4) != 0; a.w_cached = (layout & 8) != 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess || cs == hipStreamCaptureStatusNone) {
        const int rc = s2s::dec2_ensure_default_ws();     // (never allocates inside a capture: s2s_graph_bench_begin does it first)
        if (rc) return rc;
    }
    return s2s::gemm_dec((hipStream_t)stream, a);
}

extern "C" int32_t s2s_op_gemm_dec(void* stream, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int32_t M, int32_t N,
                                   int32_t K, const float* ss_in, int32_t n_parts_in, int32_t ss_stride, float inv_dim, float eps, int32_t mode,
                                   void* C, int64_t ldc, int32_t act, float* X, int64_t ldx, void* XB, int64_t ldxb, float* ss_out, int32_t glu,
                                   int32_t nf, int32_t layout) {
    return op_gemm_dec_impl(stream, A, lda, W, ldw, bias, M, N, K, ss_in, n_parts_in, ss_stride, inv_dim, eps, mode, C, ldc, act, X, ldx, XB, ldxb, ss_out, glu,
                            nf, layout, nullptr);
}

/* The same projection behind a LAYERNORM (Whisper's pre-LN decoder blocks): ss_in holds pair partials (layout bit 16 on the producing
 * mode-2 call / s2s_op_rows_prep_pairs), W has gamma folded into its columns, csum_dev f32 [N] = column sums of those folded bf16
 * weights, bias = b + W beta.  out = rstd * (A W'^T - mean * csum) + bias. */
extern "C" int32_t s2s_op_gemm_dec_ln(void* stream, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int32_t M, int32_t N,
                                      int32_t K, const float* ss_in, int32_t n_parts_in, int32_t ss_stride, float inv_dim, float eps, int32_t mode,
                                      void* C, int64_t ldc, int32_t act, int32_t layout, const float* csum_dev) {
    S2S_REQUIRE(csum_dev && ss_in && mode != 2, "op_gemm_dec_ln: needs csum, pair partials and an output mode 0 / 1");
    // (up to 128 rows the call is pinned to the 16-column kernel's one-fragment variant; wider calls take the library's own choice - the
    // row-block kernel of gemm_decw.hip when the rows are fragment-major)
    return op_gemm_dec_impl(stream, A, lda, W, ldw, bias, M, N, K, ss_in, n_parts_in, ss_stride, inv_dim, eps, mode, C, ldc, act, nullptr, 0, nullptr, 0, nullptr, 0,
                            M > 128 ? 0 : 1, layout, csum_dev);
}

/* number of partial sums of squares per row s2s_op_gemm_dec (mode 2) writes for this shape (nf 0 -> the library's choice); layout as
 * in s2s_op_gemm_dec (bit 1, fragment-major A, and bit 16, pair partials, matter) */
extern "C" int32_t s2s_op_gemm_dec_parts(int32_t M, int32_t N, int32_t K, int32_t nf, int32_t layout) {
    s2s::DecGemm a{};
    a.M = M; a.N = N; a.K = K; a.nf = nf; a.mode = 2; a.a_frag = (layout & 1) != 0; a.ss_pairs = (layout & 16) != 0;
    return s2s::dec_out_parts(a);
}

extern "C" int32_t s2s_op_rows_prep(void* stream, const float* x, int64_t x_rs, void* xb, int64_t xb_rs, float* ss, int32_t rows, int32_t cols,
                                    int32_t frag) {
    return s2s::rows_prep((hipStream_t)stream, x, (long)x_rs, xb, (long)xb_rs, ss, rows, cols, frag != 0);
}

/* ... and ss[pair_stride + r] = sum_c x[r][c]: the pair partials (parts 0 and 1) a LayerNorm consumer reads */
extern "C" int32_t s2s_op_rows_prep_pairs(void* stream, const float* x, int64_t x_rs, void* xb, int64_t xb_rs, float* ss, int32_t rows, int32_t cols,
                                          int32_t frag, int64_t pair_stride) {
    S2S_REQUIRE(pair_stride >= rows, "op_rows_prep_pairs: pair_stride %ld < rows %d", (long)pair_stride, rows);
    return s2s::rows_prep((hipStream_t)stream, x, (long)x_rs, xb, (long)xb_rs, ss, rows, cols, frag != 0, (long)pair_stride);
}

/* row-major bf16 [rows][K] -> the fragment-major layout s2s_op_gemm_dec reads (rows padded to a multiple of 16 with zeros;
 * out holds roundup(rows, 16) * K elements); gamma (optional f32 [K]) is folded into the columns. */
extern "C" int32_t s2s_op_to_frag(void* stream, const void* W, int64_t ld, const float* gamma, void* out, int32_t rows, int32_t K) {
    return s2s::to_frag((hipStream_t)stream, W, (long)ld, gamma, out, rows, K);
}
