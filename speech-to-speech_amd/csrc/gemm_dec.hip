// Decode-step projections for gfx950 (bf16, M <= 128 rows = one row per live session).
//
// One decoder layer of a decode step is FIVE launches on this path (qkv | attention | out-proj | gate/up | down) instead
// of seven: the two RMSNorm launches are gone.  The kernel that writes the residual stream (out-proj / down-proj
// epilogue) also emits (a) a bf16 copy of the new residual rows and (b) per-column-slab partial sums of squares; the
// consuming projection multiplies the RAW bf16 rows with weights that have the norm's gamma folded into their columns
// and scales its accumulators by rsqrt(mean(x^2) + eps) in the epilogue (the partials are summed in a fixed order:
// deterministic, no atomics).
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

__device__ __forceinline__ u32x4_t ld_nt16(const void* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
}

// MODE 0: C bf16 = act(rstd * acc + bias); GLU: C = silu(rstd*gate) * (rstd*up)
// MODE 1: C f32  = act(rstd * acc + bias)
// MODE 2: X f32 (in place) += rstd * acc + bias; XB = bf16(X); ss_out[slab][m] = sum over the slab's columns of X^2
template <int MT, int NF, bool GLU, int MODE, int UC>
__global__ __launch_bounds__(DEC_NW * 64) void gemm_dec_kernel(DecGemm a) {
    constexpr int NS = NF * (GLU ? 2 : 1);       // weight streams per wave
    constexpr int GRMAX = GLU ? 2 : 4;
    constexpr int GR = (MT < GRMAX ? MT : GRMAX);   // row fragments combined per LDS pass (32 KiB of LDS)
    constexpr int PASSES = (MT + GR - 1) / GR;
    __shared__ float red[DEC_NW][GR * (GLU ? 2 : 1)][4][64];
    __shared__ float ssp[DEC_NW][MT * 16];
    __shared__ float rstd_s[MT * 16];

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: k-slice bounds live in SGPRs
    const int n0 = blockIdx.x * 16 * NF;
    const int steps = a.K >> 5;                                  // k-steps of 32
    const int spw = (steps + DEC_NW - 1) / DEC_NW;
    const int s0 = wave * spw, s1 = min(steps, s0 + spw);

    const __hip_bfloat16* wrow[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        // columns past N / rows past M read the last valid row: finite garbage that is never stored, no divergent loads
        const int n = min(n0 + (s % NF) * 16 + l15, a.N - 1);
        wrow[s] = a.W + ((long)n + (s >= NF ? a.N : 0)) * a.ldw + lg * 8;
    }
    const __hip_bfloat16* arow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = min(i * 16 + l15, a.M - 1);
        arow[i] = a.A + (long)m * a.lda + lg * 8;
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
        const __hip_bfloat16* pw[NS];
        const __hip_bfloat16* pa[MT];
#pragma unroll
        for (int q = 0; q < NS; ++q) pw[q] = wrow[q] + (long)kb * 32;
#pragma unroll
        for (int i = 0; i < MT; ++i) pa[i] = arow[i] + (long)kb * 32;
#pragma unroll
        for (int u = 0; u < UC; ++u)
#pragma unroll
            for (int q = 0; q < NS; ++q) fw[st][u][q] = ld_nt16(pw[q] + u * 32);
#pragma unroll
        for (int u = 0; u < UC; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[st][u][i] = *reinterpret_cast<const u32x4_t*>(pa[i] + u * 32);
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
    constexpr int RH = (MT * 16 + 63) / 64, RP = MT > 4 ? 8 : 16;     // prefetched parts per wave (the rest: tail loop)
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
    // Three straight-line schedules (separate code, so each one's wait counts and register assignment are exact):
    if (s1 - s0 <= UC) {                 // short slice (e.g. K = 1024): a single stage holds it, nothing is re-read
        load(0, s0);
        rms_parts();
        compute(0);
    } else if (s1 - s0 <= 2 * UC) {      // the whole slice is in flight at once (K = 2048)
        load(0, s0);
        load(1, s0 + UC);
        rms_parts();
        compute(0);
        compute(1);
    } else {                             // long slice: two stages in flight, refilled behind the MFMAs
        load(0, s0);
        load(1, s0 + UC);
        rms_parts();
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
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < DEC_NW; ++w) v += ssp[w][tid];
                    rstd_s[tid] = rsqrtf(v * a.inv_dim + a.eps);
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
                    if (ok) Elem<__hip_bfloat16>::st(reinterpret_cast<__hip_bfloat16*>(a.C) + (long)m * a.ldc + nn, v);
                } else if constexpr (MODE == 1) {
                    if (ok) reinterpret_cast<float*>(a.C)[(long)m * a.ldc + nn] = v;
                } else {
                    float xn = 0.f;
                    if (ok) {
                        float* xp = a.X + (long)m * a.ldx + nn;
                        xn = a.x_overwrite ? v : *xp + v;
                        *xp = xn;
                        if (a.XB) Elem<__hip_bfloat16>::st(a.XB + (long)m * a.ldxb + nn, xn);
                    }
                    if (a.ss_out) {
                        float q = xn * xn;
                        q += __shfl_xor(q, 1, 64);
                        q += __shfl_xor(q, 2, 64);
                        q += __shfl_xor(q, 4, 64);
                        q += __shfl_xor(q, 8, 64);
                        if (col == 0 && m < a.M && e < GR * 256) a.ss_out[(long)(blockIdx.x * NF + f) * a.ss_stride + m] = q;
                    }
                }
            }
        }
    }
}

template <int MT, int NF, bool GLU, int MODE>
int launch_dec(hipStream_t st, const DecGemm& a) {
    constexpr int NS = NF * (GLU ? 2 : 1);
    // register budget: two stages of UC * (NS + MT) 16-byte fragments + MT * NS accumulators
    constexpr int FR = NS + MT;      // 16-byte fragments per k-step
    constexpr int UC = FR <= 2 ? 8 : FR == 3 ? 6 : FR <= 5 ? 4 : FR == 6 ? 3 : FR <= 9 ? 2 : 1;
    const int slabs = cdiv(a.N, 16 * NF);
    gemm_dec_kernel<MT, NF, GLU, MODE, UC><<<slabs, DEC_NW * 64, 0, st>>>(a);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

template <int MT>
int launch_dec_mt(hipStream_t st, const DecGemm& a, int NF) {
    constexpr bool NF2 = MT <= 4, NF2_GLU = MT <= 2;      // register budget (see launch_dec)
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

int dec_nf(const DecGemm& a) {
    if (a.nf > 0) return a.nf;
    // enough workgroups to put one on every CU first; above that, two column fragments per workgroup halve the
    // activation re-reads from L2.  M > 64 keeps NF = 1 (register budget).
    if (a.M > 64 || (a.glu && a.M > 32)) return 1;
    return a.N >= 8192 ? 2 : 1;
}

int gemm_dec(hipStream_t st, const DecGemm& a) {
    S2S_REQUIRE(a.M >= 1 && a.M <= 128 && a.N > 0 && a.K > 0, "gemm_dec: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    S2S_REQUIRE(a.K >= 256 && a.K % 32 == 0 && a.lda % 8 == 0 && a.ldw % 8 == 0, "gemm_dec: K=%d must be >= 256 and a multiple of 32, lda/ldw multiples of 8", a.K);
    S2S_REQUIRE(a.A && a.W && ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm_dec: operands must be 16-byte aligned");
    S2S_REQUIRE(!a.glu || (a.mode == 0 && !a.bias && a.act == S2S_ACT_NONE), "gemm_dec: SwiGLU output is bf16 without bias / activation");
    S2S_REQUIRE(a.mode == 2 ? (a.X != nullptr) : (a.C != nullptr), "gemm_dec: null output");
    S2S_REQUIRE(!a.ss_in || (a.n_parts_in > 0 && a.ss_stride >= a.M && a.inv_dim > 0.f), "gemm_dec: bad RMS prologue arguments");
    const int NF = dec_nf(a);
    ProfScope prof(st, S2S_PROF_GEMM_SKINNY, 2.0 * a.M * a.N * a.K * (a.glu ? 2 : 1),
                   (double)a.N * a.K * 2 * (a.glu ? 2 : 1) + (double)a.M * a.K * 2 + (double)a.M * a.N * (a.mode == 0 ? 2 : 4));
    const int MT = cdiv(a.M, 16);
    if (MT <= 1) return launch_dec_mt<1>(st, a, NF);
    if (MT <= 2) return launch_dec_mt<2>(st, a, NF);
    if (MT <= 4) return launch_dec_mt<4>(st, a, NF);
    return launch_dec_mt<8>(st, a, 1);
}

// rows of the f32 residual stream -> bf16 copy + one partial sum of squares per row (entry of a fused decode pass)
__global__ __launch_bounds__(256) void rows_prep_kernel(const float* __restrict__ x, long x_rs, __hip_bfloat16* __restrict__ xb, long xb_rs,
                                                        float* __restrict__ ss, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * x_rs;
    float s = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        ushort4 p;
        p.x = f32_to_bf16_bits(v.x); p.y = f32_to_bf16_bits(v.y); p.z = f32_to_bf16_bits(v.z); p.w = f32_to_bf16_bits(v.w);
        *reinterpret_cast<ushort4*>(xb + (long)row * xb_rs + c) = p;
    }
    s = wave_sum(s);
    if (lane == 0) ss[row] = s;
}

int rows_prep(hipStream_t st, const float* x, long x_rs, void* xb, long xb_rs, float* ss, int rows, int cols) {
    S2S_REQUIRE(cols % 4 == 0 && x_rs % 4 == 0 && xb_rs % 4 == 0, "rows_prep: cols=%d unsupported", cols);
    if (rows == 0) return S2S_OK;
    rows_prep_kernel<<<cdiv(rows, 4), 256, 0, st>>>(x, x_rs, (__hip_bfloat16*)xb, xb_rs, ss, rows, cols);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

}  // namespace s2s

/* Public op (parity tests / microbenchmarks): one decode-step projection.  See include/s2s_mi355x.h. */
extern "C" int32_t s2s_op_gemm_dec(void* stream, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, int32_t M, int32_t N,
                                   int32_t K, const float* ss_in, int32_t n_parts_in, int32_t ss_stride, float inv_dim, float eps, int32_t mode,
                                   void* C, int64_t ldc, int32_t act, float* X, int64_t ldx, void* XB, int64_t ldxb, float* ss_out, int32_t glu,
                                   int32_t nf) {
    S2S_REQUIRE(mode >= 0 && mode <= 2 && nf >= 0 && nf <= 2, "op_gemm_dec: bad mode / nf");
    s2s::DecGemm a{};
    a.A = (const __hip_bfloat16*)A; a.lda = (long)lda; a.W = (const __hip_bfloat16*)W; a.ldw = (long)ldw; a.bias = bias;
    a.M = M; a.N = N; a.K = K;
    a.ss_in = ss_in; a.n_parts_in = n_parts_in; a.ss_stride = ss_stride; a.inv_dim = inv_dim; a.eps = eps;
    a.mode = mode; a.C = C; a.ldc = (long)ldc; a.act = act;
    a.X = X; a.ldx = (long)ldx; a.XB = (__hip_bfloat16*)XB; a.ldxb = (long)ldxb; a.ss_out = ss_out;
    a.glu = glu != 0; a.nf = nf;
    return s2s::gemm_dec((hipStream_t)stream, a);
}

/* number of partial sums of squares per row s2s_op_gemm_dec (mode 2) writes for this shape (0 -> default nf) */
extern "C" int32_t s2s_op_gemm_dec_parts(int32_t M, int32_t N, int32_t nf) {
    s2s::DecGemm a{};
    a.M = M; a.N = N; a.nf = nf; a.mode = 2;
    return s2s::dec_parts(N, s2s::dec_nf(a));
}

extern "C" int32_t s2s_op_rows_prep(void* stream, const float* x, int64_t x_rs, void* xb, int64_t xb_rs, float* ss, int32_t rows, int32_t cols) {
    return s2s::rows_prep((hipStream_t)stream, x, (long)x_rs, xb, (long)xb_rs, ss, rows, cols);
}
