// Attention kernels for gfx950.
//
//  * attn_rows_kernel  : one wave per query row, f32 math on any storage type.  Serves the
//                        decode steps (one query per session against the self / cross KV
//                        cache, grouped-query, causal and sliding-window masks) and is the
//                        exact-f32 parity path for every attention in the models.
//                        Scores live in LDS; pass 1 = q.K^T with each lane owning keys,
//                        pass 2 = P.V with each lane owning output channels (coalesced V rows).
//  * attn_mfma_kernel  : bf16 flash attention for the encoders (Whisper T=1500 d_h=64,
//                        FastConformer d_h=128 with the Transformer-XL relative-position
//                        term as an additive bias).  128 queries per workgroup (4 waves x 2
//                        query fragments), K and V^T tiles of 64 keys staged through
//                        swizzled LDS, S^T = K.Q^T ("swapped") so each lane owns one query
//                        column: row max / sum need two cross-lane steps only and the
//                        probabilities feed the P.V MFMA straight from registers.
//  * transpose_v_kernel: V [tokens][H*D] -> V^T [B][H][D][Tp] (zero padded) for the above.
#include "ops.h"

namespace s2s {

// ------------------------------------------------------------------------------------
// generic row-wise attention
// ------------------------------------------------------------------------------------


template <typename T, int D>
__global__ __launch_bounds__(256) void attn_rows_kernel(AttnRowsParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    const int h = blockIdx.y, b = blockIdx.z;
    const int kvmax = p.kv_len_const;
    float* qs = smem + wave * (D + kvmax + 8);   // q (f32) then scores
    float* sc = qs + D;
    if (t >= p.Tq) return;   // whole wave exits together; no block-level barriers below
    const int hk = h / (p.H / p.Hkv);
    int kv = p.kv_len ? min(p.kv_len[b], kvmax) : kvmax;
    int jlo = 0, jhi = kv;   // allowed keys [jlo, jhi)
    const int coff = p.causal_off_arr ? p.causal_off_arr[b] : p.causal_off;
    if (p.causal) jhi = min(kv, t + coff + 1);
    if (p.window > 0) jlo = max(0, t + coff - p.window + 1);

    const T* q = (const T*)p.Q + (long)b * p.q_bs + (long)t * p.q_rs + (long)h * D;
    for (int d = lane; d < D; d += 64) qs[d] = Elem<T>::ld(q + d) * p.scale;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the LDS writes above are visible to the wave

    const T* Kb = (const T*)p.K + (long)b * p.kv_bs + (long)hk * D;
    const T* Vb = (const T*)p.V + (long)b * p.kv_bs + (long)hk * D;
    const float* bias = p.bias ? p.bias + (long)b * p.bias_bs + (long)h * p.bias_hs + (long)t * p.bias_rs + (p.bias_shift - t)
                               : nullptr;
    float mx = -INFINITY;
    for (int j = jlo + lane; j < jhi; j += 64) {
        const T* kr = Kb + (long)j * p.kv_rs;
        float s = 0.f;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int c = 0; c < D / 8; ++c) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(kr + c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s = fmaf(__uint_as_float(v[e] << 16), qs[c * 8 + 2 * e], s);
                    s = fmaf(__uint_as_float(v[e] & 0xffff0000u), qs[c * 8 + 2 * e + 1], s);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < D / 4; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(kr + c * 4);
                s = fmaf(v.x, qs[c * 4], s);
                s = fmaf(v.y, qs[c * 4 + 1], s);
                s = fmaf(v.z, qs[c * 4 + 2], s);
                s = fmaf(v.w, qs[c * 4 + 3], s);
            }
        }
        if (bias) s += bias[j];
        sc[j - jlo] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = jlo + lane; j < jhi; j += 64) {
        const float e = expf(sc[j - jlo] - mx);
        sc[j - jlo] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;

    T* o = (T*)p.O + (long)b * p.o_bs + (long)t * p.o_rs + (long)h * D;
    constexpr int PER = (D + 63) / 64;
    float acc[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) acc[u] = 0.f;
    const int n = jhi - jlo;
    int j = 0;
    for (; j + 4 <= n; j += 4) {
        float pv[4];
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) pv[u4] = sc[j + u4];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int d = lane + u * 64;
            if (d < D) {
                float v0 = Elem<T>::ld(Vb + (long)(jlo + j) * p.kv_rs + d);
                float v1 = Elem<T>::ld(Vb + (long)(jlo + j + 1) * p.kv_rs + d);
                float v2 = Elem<T>::ld(Vb + (long)(jlo + j + 2) * p.kv_rs + d);
                float v3 = Elem<T>::ld(Vb + (long)(jlo + j + 3) * p.kv_rs + d);
                acc[u] = fmaf(pv[0], v0, acc[u]);
                acc[u] = fmaf(pv[1], v1, acc[u]);
                acc[u] = fmaf(pv[2], v2, acc[u]);
                acc[u] = fmaf(pv[3], v3, acc[u]);
            }
        }
    }
    for (; j < n; ++j) {
        const float pj = sc[j];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int d = lane + u * 64;
            if (d < D) acc[u] = fmaf(pj, Elem<T>::ld(Vb + (long)(jlo + j) * p.kv_rs + d), acc[u]);
        }
    }
    // fragment-major output rows (operand layout of the decode projections, gemm_dec.hip): row b * Tq + t, column h * D + d
    const long orow = (long)b * p.Tq + t;
    const int oksteps = (p.H * D) >> 5;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int d = lane + u * 64;
        if (d < D && p.o_frag) {
            const int k = h * D + d;
            Elem<T>::st((T*)p.O + ((((orow >> 4) * oksteps + (k >> 5)) * 4 + ((k >> 3) & 3)) * 16 + (orow & 15)) * 8 + (k & 7), acc[u] * inv);
        } else if (d < D) Elem<T>::st(o + d, acc[u] * inv);
    }
}

// ------------------------------------------------------------------------------------
// single-query (decode) attention: one workgroup per (session, head); CH = 16-byte chunks per K/V row lanes cooperate
// on one key so every load instruction covers whole 128/256-byte rows (64/CH keys per wave instruction); the 4 waves
// split the keys.  Pass 1 scores -> LDS, block softmax, pass 2 P.V with per-lane chunk accumulators.
// ------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnRowsParams p) {
    constexpr int EPL = 16 / (int)sizeof(T);     // elements per lane chunk
    constexpr int CH = D / EPL;                  // lanes per key row
    constexpr int KPI = 64 / CH;                 // keys per wave instruction
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sc = smem;                             // [kvmax]
    float* red = smem + p.kv_len_const + 8;       // [8] + [4][D]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int hk = h / (p.H / p.Hkv);
    int kv = p.kv_len ? min(p.kv_len[b], p.kv_len_const) : p.kv_len_const;
    const int coff = p.causal_off_arr ? p.causal_off_arr[b] : p.causal_off;
    int jlo = 0, jhi = kv;
    if (p.causal) jhi = min(kv, coff + 1);
    if (p.window > 0) jlo = max(0, coff - p.window + 1);
    const int n = max(jhi - jlo, 0);
    const int c = lane % CH, slot = lane / CH;

    float q[EPL];
    {
        const T* qp = (const T*)p.Q + (long)b * p.q_bs + (long)h * D + c * EPL;
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[e] = Elem<T>::ld(qp + e) * p.scale;
    }
    const T* Kb = (const T*)p.K + (long)b * p.kv_bs + (long)hk * D + c * EPL;
    const T* Vb = (const T*)p.V + (long)b * p.kv_bs + (long)hk * D + c * EPL;
    // keys of this wave: contiguous quarter, rounded to KPI
    const int per = ((n + 3) / 4 + KPI - 1) / KPI * KPI;
    const int w0 = jlo + wave * per, w1 = min(jhi, w0 + per);

    float mx = -INFINITY;
    constexpr int UN = 4;     // key groups in flight per wave
    for (int j0 = w0; j0 < w1; j0 += UN * KPI) {
        u32x4_t kvv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * KPI + slot;
            kvv[u] = (u32x4_t){0u, 0u, 0u, 0u};
            if (j < w1) kvv[u] = *reinterpret_cast<const u32x4_t*>(Kb + (long)j * p.kv_rs);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * KPI + slot;
            const u32x4_t v = kvv[u];
            float s = 0.f;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s = fmaf(__uint_as_float(v[e] << 16), q[2 * e], s);
                    s = fmaf(__uint_as_float(v[e] & 0xffff0000u), q[2 * e + 1], s);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) s = fmaf(__uint_as_float(v[e]), q[e], s);
            }
#pragma unroll
            for (int o = 1; o < CH; o <<= 1) s += __shfl_xor(s, o, 64);
            if (j < w1) {
                if (c == 0) sc[j - jlo] = s;
                mx = fmaxf(mx, s);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < n; j += 256) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;

    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    for (int j0 = w0; j0 < w1; j0 += UN * KPI) {
        u32x4_t kvv[UN];
        float pj[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * KPI + slot;
            kvv[u] = (u32x4_t){0u, 0u, 0u, 0u};
            pj[u] = 0.f;
            if (j < w1) {
                kvv[u] = *reinterpret_cast<const u32x4_t*>(Vb + (long)j * p.kv_rs);
                pj[u] = sc[j - jlo];
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const u32x4_t v = kvv[u];
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] = fmaf(pj[u], __uint_as_float(v[e] << 16), acc[2 * e]);
                    acc[2 * e + 1] = fmaf(pj[u], __uint_as_float(v[e] & 0xffff0000u), acc[2 * e + 1]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(pj[u], __uint_as_float(v[e]), acc[e]);
            }
        }
    }
    // combine the KPI key slots of the wave (lanes with the same chunk c), then the 4 waves
#pragma unroll
    for (int e = 0; e < EPL; ++e)
#pragma unroll
        for (int o = CH; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
    float* part = red + 8;     // [4][D]
    if (slot == 0) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) part[wave * D + c * EPL + e] = acc[e];
    }
    __syncthreads();
    T* o = (T*)p.O + (long)b * p.o_bs + (long)h * D;
    const int ksteps = (p.H * D) >> 5;
    for (int d = tid; d < D; d += 256) {
        const float v = ((part[d] + part[D + d]) + (part[2 * D + d] + part[3 * D + d])) * inv;
        if (p.o_frag) {   // fragment-major rows [B][H*D] for the projection that follows (gemm_dec.hip operand layout)
            const int k = h * D + d;
            Elem<T>::st((T*)p.O + ((((long)(b >> 4) * ksteps + (k >> 5)) * 4 + ((k >> 3) & 3)) * 16 + (b & 15)) * 8 + (k & 7), v);
        } else {
            Elem<T>::st(o + d, v);
        }
    }
}

// ------------------------------------------------------------------------------------
// Decode STEP attention (one new token per session) with everything in front of it fused: per-head RMSNorm of q and k
// (optional), rotary embedding at the session's position (optional), the append of the new K / V row to the cache, and
// the attention itself over cache[jlo .. pos) + the new row (kept in LDS, never re-read from memory).  One workgroup per
// (head, session); the q heads of a GQA group each rebuild the (tiny) new K row, the group's first head stores it.
// Same key split / chunk-lane layout as attn_decode_kernel.  Replaces rope_qknorm + kv_append + attn_decode launches.
// ------------------------------------------------------------------------------------
// GQ > 1 (round 6): ONE workgroup per (KV head, session) serves the GQ query heads of the group - every K / V row is loaded once and
// multiplied with all of them.  With one workgroup per QUERY head the two heads of a Qwen3-TTS group (16 Q / 8 KV) sat on different XCDs
// (consecutive block ids go round-robin over the 8 L2s) and each streamed the group's cache rows from HBM: 2 x the traffic of the kernel that
// bounds the talker step at 512 rows (0.35 GB per layer and step).  Per-head arithmetic - the key split over waves, the chunk-lane layout,
// every reduction order - is exactly the GQ = 1 kernel's, so the outputs are bit-identical to it.  GQ = 1: one workgroup per query head, any
// group size (runtime), as before.
template <typename T, int D, int GQ>
__global__ __launch_bounds__(256) void attn_step_kernel(AttnStepParams p) {
    constexpr int EPL = 16 / (int)sizeof(T);
    constexpr int CH = D / EPL;
    constexpr int KPI = 64 / CH;
    constexpr int HALF = D / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int sc_ld = p.kv_cap + 8;
    float* sc = smem;                               // [GQ][kv_cap + 8]
    float* red = smem + GQ * sc_ld;                 // per head: [8] + [4][D]
    constexpr int RED_LD = 8 + 4 * D;
    float* sq = red + GQ * RED_LD;                  // [GQ][D] scaled, rotated queries
    float* sk = sq + GQ * D;                        // [D] new key (values as stored in the cache)
    float* sv = sk + D;                             // [D] new value
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int group = p.H / p.Hkv;
    const int hk = GQ > 1 ? (int)blockIdx.x : (int)blockIdx.x / group;
    const int h0 = GQ > 1 ? hk * GQ : (int)blockIdx.x;            // first (GQ = 1: the only) query head of this workgroup
    const bool owns_kv = GQ > 1 || h0 % group == 0;               // this workgroup appends the new K / V row to the cache
    const int pos = p.pos[b];
    const T* row = (const T*)p.qkv + (long)b * p.qkv_rs;
    T* Kc = (T*)p.K + (long)b * p.kv_bs + (long)hk * D;
    T* Vc = (T*)p.V + (long)b * p.kv_bs + (long)hk * D;

    // ---- new row: tasks 0 .. GQ-1 -> the queries, GQ -> k (same arithmetic as rope_qknorm_kernel), GQ + 1 -> v; one task per wave and round ----
    for (int task = wave; task < GQ + 2; task += 4) {
        if (task <= GQ) {
            const bool is_q = task < GQ;
            const T* x = row + (is_q ? (long)(h0 + task) * D : (long)(p.H + hk) * D);
            const float* w = is_q ? p.qw : p.kw;
            float a = 0.f, c = 0.f;
            if (lane < HALF) {
                a = Elem<T>::ld(x + lane);
                c = Elem<T>::ld(x + lane + HALF);
            }
            if (w) {
                const float ss = wave_sum(a * a + c * c);
                const float r = rsqrtf(ss / (float)D + p.eps);
                if (lane < HALF) {
                    a = a * r * w[lane];
                    c = c * r * w[lane + HALF];
                }
            }
            if (lane < HALF) {
                float r0 = a, r1 = c;
                if (p.cosT) {
                    const float cs = p.cosT[(long)pos * HALF + lane], sn = p.sinT[(long)pos * HALF + lane];
                    r0 = a * cs - c * sn;
                    r1 = c * cs + a * sn;
                }
                // round through the storage type exactly like the unfused path (q was written back, k goes to the cache)
                T t0, t1;
                Elem<T>::st(&t0, r0);
                Elem<T>::st(&t1, r1);
                const float f0 = Elem<T>::ld(&t0), f1 = Elem<T>::ld(&t1);
                if (is_q) {
                    sq[task * D + lane] = f0 * p.scale;
                    sq[task * D + lane + HALF] = f1 * p.scale;
                } else {
                    sk[lane] = f0;
                    sk[lane + HALF] = f1;
                    if (owns_kv) {
                        Kc[(long)pos * p.kv_rs + lane] = t0;
                        Kc[(long)pos * p.kv_rs + lane + HALF] = t1;
                    }
                }
            }
        } else {
            const T* x = row + (long)(p.H + p.Hkv + hk) * D;
            for (int d = lane; d < D; d += 64) {
                const T t = x[d];
                sv[d] = Elem<T>::ld(&t);
                if (owns_kv) Vc[(long)pos * p.kv_rs + d] = t;
            }
        }
    }
    __syncthreads();

    const int jlo = p.window > 0 ? max(0, pos - p.window + 1) : 0;
    const int nc = pos - jlo;                       // cached keys [jlo, pos); the new key is entry nc
    const int c = lane % CH, slot = lane / CH;
    float q[GQ][EPL];
#pragma unroll
    for (int g = 0; g < GQ; ++g)
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[g][e] = sq[g * D + c * EPL + e];
    const T* Kb = Kc + c * EPL;
    const T* Vb = Vc + c * EPL;
    const int per = ((nc + 3) / 4 + KPI - 1) / KPI * KPI;
    const int w0 = jlo + wave * per, w1 = min(pos, w0 + per);

    float mx[GQ];
#pragma unroll
    for (int g = 0; g < GQ; ++g) mx[g] = -INFINITY;
    constexpr int UN = 4;
    for (int j0 = w0; j0 < w1; j0 += UN * KPI) {
        u32x4_t kvv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * KPI + slot;
            kvv[u] = (u32x4_t){0u, 0u, 0u, 0u};
            if (j < w1) kvv[u] = *reinterpret_cast<const u32x4_t*>(Kb + (long)j * p.kv_rs);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * KPI + slot;
            const u32x4_t v = kvv[u];
#pragma unroll
            for (int g = 0; g < GQ; ++g) {
                float s = 0.f;
                if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s = fmaf(__uint_as_float(v[e] << 16), q[g][2 * e], s);
                        s = fmaf(__uint_as_float(v[e] & 0xffff0000u), q[g][2 * e + 1], s);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) s = fmaf(__uint_as_float(v[e]), q[g][e], s);
                }
#pragma unroll
                for (int o = 1; o < CH; o <<= 1) s += __shfl_xor(s, o, 64);
                if (j < w1) {
                    if (c == 0) sc[g * sc_ld + j - jlo] = s;
                    mx[g] = fmaxf(mx[g], s);
                }
            }
        }
    }
    // the new key: every wave forms the same dot products from LDS (D <= 128: two elements per lane)
    float sn[GQ];
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
        float t = 0.f;
        for (int d = lane; d < D; d += 64) t = fmaf(sq[g * D + d], sk[d], t);
        sn[g] = wave_sum(t);
        mx[g] = fmaxf(wave_max(mx[g]), sn[g]);
        if (lane == 0) red[g * RED_LD + wave] = mx[g];
    }
    __syncthreads();
    float inv[GQ], pn[GQ];
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
        float* rg = red + g * RED_LD;
        mx[g] = fmaxf(fmaxf(rg[0], rg[1]), fmaxf(rg[2], rg[3]));
        float sum = 0.f;
        for (int j = tid; j < nc; j += 256) {
            const float e = expf(sc[g * sc_ld + j] - mx[g]);
            sc[g * sc_ld + j] = e;
            sum += e;
        }
        pn[g] = expf(sn[g] - mx[g]);
        sum = wave_sum(sum);
        if (lane == 0) rg[4 + wave] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
        const float* rg = red + g * RED_LD;
        const float sum = ((rg[4] + rg[5]) + (rg[6] + rg[7])) + pn[g];
        inv[g] = 1.0f / sum;
    }

    float acc[GQ][EPL];
#pragma unroll
    for (int g = 0; g < GQ; ++g)
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[g][e] = 0.f;
    for (int j0 = w0; j0 < w1; j0 += UN * KPI) {
        u32x4_t kvv[UN];
        float pj[GQ][UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * KPI + slot;
            kvv[u] = (u32x4_t){0u, 0u, 0u, 0u};
#pragma unroll
            for (int g = 0; g < GQ; ++g) pj[g][u] = 0.f;
            if (j < w1) {
                kvv[u] = *reinterpret_cast<const u32x4_t*>(Vb + (long)j * p.kv_rs);
#pragma unroll
                for (int g = 0; g < GQ; ++g) pj[g][u] = sc[g * sc_ld + j - jlo];
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const u32x4_t v = kvv[u];
#pragma unroll
            for (int g = 0; g < GQ; ++g) {
                if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[g][2 * e] = fmaf(pj[g][u], __uint_as_float(v[e] << 16), acc[g][2 * e]);
                        acc[g][2 * e + 1] = fmaf(pj[g][u], __uint_as_float(v[e] & 0xffff0000u), acc[g][2 * e + 1]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[g][e] = fmaf(pj[g][u], __uint_as_float(v[e]), acc[g][e]);
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
#pragma unroll
            for (int o = CH; o < 64; o <<= 1) acc[g][e] += __shfl_xor(acc[g][e], o, 64);
        float* part = red + g * RED_LD + 8;     // [4][D]
        if (slot == 0) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) part[wave * D + c * EPL + e] = acc[g][e];
        }
    }
    __syncthreads();
    const int ksteps = (p.H * D) >> 5;
    for (int i = tid; i < GQ * D; i += 256) {
        const int g = i / D, d = i - g * D, h = h0 + g;
        const float* part = red + g * RED_LD + 8;
        const float v = (((part[d] + part[D + d]) + (part[2 * D + d] + part[3 * D + d])) + pn[g] * sv[d]) * inv[g];
        if (p.o_frag) {   // fragment-major rows for the out-projection (gemm_dec.hip operand layout): row b, column h*D + d
            const int k = h * D + d;
            Elem<T>::st((T*)p.O + ((((long)(b >> 4) * ksteps + (k >> 5)) * 4 + ((k >> 3) & 3)) * 16 + (b & 15)) * 8 + (k & 7), v);
        } else {
            Elem<T>::st((T*)p.O + (long)b * p.o_bs + (long)h * D + d, v);
        }
    }
}

// ---- decode-step attention over a SHORT cache (round 6): one WAVE per (KV head, session) ----------------------------------------------
// The code predictor runs 15 x 5 attention steps per frame over a cache of at most 17 positions.  attn_step_kernel spends a 256-thread
// workgroup, four block barriers and two LDS round trips on each (head group, session): 26.7 us per launch at 512 rows for 19 MB of cache rows
// (0.7 TB/s) - 1.3 ms of the 9 ms frame at 256 rows.  Here one wave does the whole item - row preparation (head norms, RoPE, cache append) of
// the group's queries, the new key and value; scores of all cached keys (the chunk-lane layout of the big kernel: CH lanes per key row, KPI keys
// per wave instruction) kept in registers; softmax; PV - with wave shuffles only, no block barrier, and 4 items per workgroup, so all items of a
// launch are resident at once.  Same arithmetic per element as attn_step_kernel except for the summation ORDER of the softmax denominator and of
// PV (one wave owns all keys instead of four waves owning a quarter each): used for the bf16 serving engines when kv_cap <= 32; the float32
// parity engines keep attn_step_kernel (S2S_ATTN_STEP_SMALL=0 / s2s_op_attn_step_tune bit 1 switch it off).
template <typename T, int D, int GQ>
__global__ __launch_bounds__(256) void attn_step_small_kernel(AttnStepParams p, int units) {
    constexpr int EPL = 16 / (int)sizeof(T);
    constexpr int CH = D / EPL;
    constexpr int KPI = 64 / CH;
    constexpr int HALF = D / 2;
    constexpr int MAXIT = 32 / KPI;                 // register-resident scores: caches of up to 32 positions
    __shared__ float smw[4][(GQ + 2) * D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int item = blockIdx.x * 4 + wave;
    if (item >= units * p.B) return;                // (no block-level synchronisation below)
    const int u = item % units, b = item / units;
    const int group = p.H / p.Hkv;
    const int hk = GQ > 1 ? u : u / group;
    const int h0 = GQ > 1 ? hk * GQ : u;
    const bool owns_kv = GQ > 1 || h0 % group == 0;
    const int pos = p.pos[b];
    float* sq = smw[wave];                          // [GQ][D] scaled, rotated queries
    float* sk = sq + GQ * D;                        // [D] new key
    float* sv = sk + D;                             // [D] new value
    const T* row = (const T*)p.qkv + (long)b * p.qkv_rs;
    T* Kc = (T*)p.K + (long)b * p.kv_bs + (long)hk * D;
    T* Vc = (T*)p.V + (long)b * p.kv_bs + (long)hk * D;
#pragma unroll
    for (int task = 0; task <= GQ; ++task) {        // queries, then the new key: the arithmetic of attn_step_kernel / rope_qknorm_kernel
        const bool is_q = task < GQ;
        const T* x = row + (is_q ? (long)(h0 + task) * D : (long)(p.H + hk) * D);
        const float* w = is_q ? p.qw : p.kw;
        for (int i0 = 0; i0 < HALF; i0 += 64) {     // (D = 128: one pass; D = 256 would take two - the sum below is then per pass: D <= 128 only)
            const int i = i0 + lane;
            float a = 0.f, c = 0.f;
            if (i < HALF) {
                a = Elem<T>::ld(x + i);
                c = Elem<T>::ld(x + i + HALF);
            }
            if (w) {
                const float ss = wave_sum(a * a + c * c);
                const float r = rsqrtf(ss / (float)D + p.eps);
                if (i < HALF) {
                    a = a * r * w[i];
                    c = c * r * w[i + HALF];
                }
            }
            if (i < HALF) {
                float r0 = a, r1 = c;
                if (p.cosT) {
                    const float cs = p.cosT[(long)pos * HALF + i], sn = p.sinT[(long)pos * HALF + i];
                    r0 = a * cs - c * sn;
                    r1 = c * cs + a * sn;
                }
                T t0, t1;
                Elem<T>::st(&t0, r0);
                Elem<T>::st(&t1, r1);
                const float f0 = Elem<T>::ld(&t0), f1 = Elem<T>::ld(&t1);
                if (is_q) {
                    sq[task * D + i] = f0 * p.scale;
                    sq[task * D + i + HALF] = f1 * p.scale;
                } else {
                    sk[i] = f0;
                    sk[i + HALF] = f1;
                    if (owns_kv) {
                        Kc[(long)pos * p.kv_rs + i] = t0;
                        Kc[(long)pos * p.kv_rs + i + HALF] = t1;
                    }
                }
            }
        }
    }
    {
        const T* x = row + (long)(p.H + p.Hkv + hk) * D;
        for (int d = lane; d < D; d += 64) {
            const T t = x[d];
            sv[d] = Elem<T>::ld(&t);
            if (owns_kv) Vc[(long)pos * p.kv_rs + d] = t;
        }
    }
    __builtin_amdgcn_wave_barrier();                // (LDS traffic of one wave is in order; this only pins the compiler's order)

    const int jlo = p.window > 0 ? max(0, pos - p.window + 1) : 0;
    const int nc = pos - jlo;                       // cached keys [jlo, pos), at most 31; the new key is handled apart
    const int c = lane % CH, slot = lane / CH;
    float q[GQ][EPL], kn[EPL], vn[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        kn[e] = sk[c * EPL + e];
        vn[e] = sv[c * EPL + e];
#pragma unroll
        for (int g = 0; g < GQ; ++g) q[g][e] = sq[g * D + c * EPL + e];
    }
    const T* Kb = Kc + c * EPL;
    const T* Vb = Vc + c * EPL;
    u32x4_t kk[MAXIT], vv[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {            // every cache row this wave needs, K and V, requested up front
        const int j = jlo + it * KPI + slot;
        kk[it] = (u32x4_t){0u, 0u, 0u, 0u};
        vv[it] = (u32x4_t){0u, 0u, 0u, 0u};
        if (it * KPI < nc && j < pos) {
            kk[it] = *reinterpret_cast<const u32x4_t*>(Kb + (long)j * p.kv_rs);
            vv[it] = *reinterpret_cast<const u32x4_t*>(Vb + (long)j * p.kv_rs);
        }
    }
    float sc[GQ][MAXIT], mx[GQ], sn[GQ];
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) t = fmaf(q[g][e], kn[e], t);
#pragma unroll
        for (int o = 1; o < CH; o <<= 1) t += __shfl_xor(t, o, 64);
        sn[g] = t;
        mx[g] = t;
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int j = jlo + it * KPI + slot;
        const bool valid = it * KPI < nc && j < pos;
        const u32x4_t v = kk[it];
#pragma unroll
        for (int g = 0; g < GQ; ++g) {
            float s_ = 0.f;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s_ = fmaf(__uint_as_float(v[e] << 16), q[g][2 * e], s_);
                    s_ = fmaf(__uint_as_float(v[e] & 0xffff0000u), q[g][2 * e + 1], s_);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) s_ = fmaf(__uint_as_float(v[e]), q[g][e], s_);
            }
#pragma unroll
            for (int o = 1; o < CH; o <<= 1) s_ += __shfl_xor(s_, o, 64);
            sc[g][it] = valid ? s_ : -INFINITY;
            mx[g] = fmaxf(mx[g], sc[g][it]);
        }
    }
    float inv[GQ], pn[GQ];
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
#pragma unroll
        for (int o = CH; o < 64; o <<= 1) mx[g] = fmaxf(mx[g], __shfl_xor(mx[g], o, 64));
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            sc[g][it] = expf(sc[g][it] - mx[g]);    // (-inf -> 0 for the slots past the cache)
            sum += sc[g][it];
        }
#pragma unroll
        for (int o = CH; o < 64; o <<= 1) sum += __shfl_xor(sum, o, 64);
        pn[g] = expf(sn[g] - mx[g]);
        inv[g] = 1.0f / (sum + pn[g]);
    }
    const int ksteps = (p.H * D) >> 5;
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
        float acc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const u32x4_t v = vv[it];
            const float pj = sc[g][it];
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] = fmaf(pj, __uint_as_float(v[e] << 16), acc[2 * e]);
                    acc[2 * e + 1] = fmaf(pj, __uint_as_float(v[e] & 0xffff0000u), acc[2 * e + 1]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(pj, __uint_as_float(v[e]), acc[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
#pragma unroll
            for (int o = CH; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
            acc[e] = (acc[e] + pn[g] * vn[e]) * inv[g];
        }
        if (slot == 0) {
            const int h = h0 + g, k = h * D + c * EPL;
            T* dst = p.o_frag ? (T*)p.O + ((((long)(b >> 4) * ksteps + (k >> 5)) * 4 + ((k >> 3) & 3)) * 16 + (b & 15)) * 8 + (k & 7)
                              : (T*)p.O + (long)b * p.o_bs + k;
#pragma unroll
            for (int e = 0; e < EPL; ++e) Elem<T>::st(dst + e, acc[e]);
        }
    }
}

static int g_attn_step_share = -1;               // s2s_op_attn_step_tune: -1 = environment / default, 0 / 1 = forced

template <typename T>
static int launch_attn_step(hipStream_t st, const AttnStepParams& p) {
    static const bool share_env = [] {           // S2S_ATTN_STEP_GQA=0: one workgroup per query head also for grouped heads (rounds 2 - 5; A/B)
        const char* e = getenv("S2S_ATTN_STEP_GQA");
        return !e || atoi(e) != 0;
    }();
    const bool share = g_attn_step_share < 0 ? share_env : (g_attn_step_share & 1) != 0;
    const int group = p.H / p.Hkv;
    const int gq = share && (group == 2 || group == 4) ? group : 1;
    static const bool small_env = [] {           // S2S_ATTN_STEP_SMALL=0: short caches on attn_step_kernel too (rounds 2 - 5)
        const char* e = getenv("S2S_ATTN_STEP_SMALL");
        return !e || atoi(e) != 0;
    }();
    // one wave per item for caches of at most 32 positions (the code predictor): bf16 engines by default; tune bit 1 forces it for any dtype,
    // tune value 0 / 1 without bit 1 switches it off (tests compare the kernels)
    const bool small_ok = p.kv_cap <= 32 && (p.D == 64 || p.D == 128) && (sizeof(T) == 2 || sizeof(T) == 4);
    const bool small = small_ok && (g_attn_step_share < 0 ? (small_env && sizeof(T) == 2) : (g_attn_step_share & 2) != 0);
    if (small) {
        const int units = gq > 1 ? p.Hkv : p.H;
        dim3 sgrid(cdiv(units * p.B, 4));
        ProfScope prof(st, S2S_PROF_ATTN_ROWS, 4.0 * p.B * p.H * (double)p.kv_cap * p.D,
                       (double)sizeof(T) * p.B * ((double)p.Hkv * p.kv_cap * p.D * 2 + (double)p.H * p.D * 2));
#define S2S_STEP_SMALL(DV)                                                                             \
    do {                                                                                               \
        if (gq == 2) launch_k(attn_step_small_kernel<T, DV, 2>, sgrid, dim3(256), 0, st, p, units);    \
        else if (gq == 4) launch_k(attn_step_small_kernel<T, DV, 4>, sgrid, dim3(256), 0, st, p, units); \
        else launch_k(attn_step_small_kernel<T, DV, 1>, sgrid, dim3(256), 0, st, p, units);            \
    } while (0)
        if (p.D == 64) S2S_STEP_SMALL(64);
        else S2S_STEP_SMALL(128);
#undef S2S_STEP_SMALL
        S2S_LAUNCH_CHECK();
        return S2S_OK;
    }
    dim3 grid(gq > 1 ? p.Hkv : p.H, p.B);
    ProfScope prof(st, S2S_PROF_ATTN_ROWS, 4.0 * p.B * p.H * (double)p.kv_cap * p.D,
                   (double)sizeof(T) * p.B * ((double)p.Hkv * p.kv_cap * p.D * 2 + (double)p.H * p.D * 2));
    const size_t lds = sizeof(float) * (size_t)(gq * (p.kv_cap + 8 + 8 + 4 * p.D + p.D) + 2 * p.D);
    S2S_REQUIRE(lds <= 64 * 1024, "attention step: cache capacity %d too long", p.kv_cap);
#define S2S_STEP(DV)                                                                              \
    do {                                                                                          \
        if (gq == 2) launch_k(attn_step_kernel<T, DV, 2>, grid, dim3(256), lds, st, p);           \
        else if (gq == 4) launch_k(attn_step_kernel<T, DV, 4>, grid, dim3(256), lds, st, p);      \
        else launch_k(attn_step_kernel<T, DV, 1>, grid, dim3(256), lds, st, p);                   \
    } while (0)
    switch (p.D) {
        case 32: S2S_STEP(32); break;
        case 64: S2S_STEP(64); break;
        case 128: S2S_STEP(128); break;
        default: return fail(S2S_ERR_UNSUPPORTED, "attention: head_dim %d unsupported (32/64/128)", p.D);
    }
#undef S2S_STEP
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int attn_step(hipStream_t st, const AttnStepParams& p, int dtype) {
    S2S_REQUIRE(p.B > 0 && p.H > 0 && p.Hkv > 0 && p.H % p.Hkv == 0 && p.kv_cap > 0 && p.pos, "attention step: bad arguments");
    return dtype == S2S_BF16 ? launch_attn_step<__hip_bfloat16>(st, p) : launch_attn_step<float>(st, p);
}

template <typename T>
static int launch_attn_decode(hipStream_t st, const AttnRowsParams& p) {
    dim3 grid(p.H, p.B);
    ProfScope prof(st, S2S_PROF_ATTN_ROWS, 4.0 * p.B * p.H * (double)p.kv_len_const * p.D,
                   (double)sizeof(T) * p.B * ((double)p.Hkv * p.kv_len_const * p.D * 2 + (double)p.H * p.D * 2));
    const size_t lds = sizeof(float) * (size_t)(p.kv_len_const + 8 + 8 + 4 * p.D);
#define S2S_DEC(DV)                                                                                       \
    do {                                                                                                  \
        if (lds > 64 * 1024)                                                                              \
            S2S_HIP_CHECK(hipFuncSetAttribute((const void*)attn_decode_kernel<T, DV>,                     \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));     \
        launch_k(attn_decode_kernel<T, DV>, grid, dim3(256), lds, st, p);                                 \
    } while (0)
    switch (p.D) {
        case 32: S2S_DEC(32); break;
        case 64: S2S_DEC(64); break;
        case 128: S2S_DEC(128); break;
        default: return fail(S2S_ERR_UNSUPPORTED, "attention: head_dim %d unsupported (32/64/128)", p.D);
    }
#undef S2S_DEC
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

template <typename T>
static int launch_attn_rows(hipStream_t st, const AttnRowsParams& p) {
    dim3 grid(cdiv(p.Tq, 4), p.H, p.B);
    ProfScope prof(st, S2S_PROF_ATTN_ROWS, 4.0 * p.B * p.H * (double)p.Tq * p.kv_len_const * p.D,
                   (double)sizeof(T) * p.B * ((double)p.Hkv * p.kv_len_const * p.D * 2 + (double)p.H * p.Tq * p.D * 2));
    const size_t lds = sizeof(float) * 4 * (size_t)(p.D + p.kv_len_const + 8);
    S2S_REQUIRE(lds <= 160 * 1024, "attention: kv length %d too long for the row kernel", p.kv_len_const);
#define S2S_ROWS(DV)                                                                                     \
    do {                                                                                                 \
        if (lds > 64 * 1024)                                                                             \
            S2S_HIP_CHECK(hipFuncSetAttribute((const void*)attn_rows_kernel<T, DV>,                      \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
        launch_k(attn_rows_kernel<T, DV>, grid, dim3(256), lds, st, p);                                  \
    } while (0)
    switch (p.D) {
        case 32: S2S_ROWS(32); break;
        case 64: S2S_ROWS(64); break;
        case 128: S2S_ROWS(128); break;
        default: return fail(S2S_ERR_UNSUPPORTED, "attention: head_dim %d unsupported (32/64/128)", p.D);
    }
#undef S2S_ROWS
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int attn_rows(hipStream_t st, const AttnRowsParams& p, int dtype) {
    S2S_REQUIRE(p.B > 0 && p.H > 0 && p.Hkv > 0 && p.H % p.Hkv == 0 && p.Tq > 0, "attention: bad shape");
    if (p.kv_len_const <= 0) return fail(S2S_ERR_ARG, "attention: kv_len_const must be > 0");
    if (p.Tq == 1 && !p.bias && p.kv_len_const >= 64)
        return dtype == S2S_BF16 ? launch_attn_decode<__hip_bfloat16>(st, p) : launch_attn_decode<float>(st, p);
    S2S_REQUIRE(!p.o_frag || ((p.H * p.D) % 32 == 0 && p.o_bs == (long)p.Tq * p.H * p.D && p.o_rs == (long)p.H * p.D),
                "attention: fragment-major output needs contiguous [B * Tq][H * D] rows, H * D a multiple of 32");
    return dtype == S2S_BF16 ? launch_attn_rows<__hip_bfloat16>(st, p) : launch_attn_rows<float>(st, p);
}

// ------------------------------------------------------------------------------------
// V transpose: V rows (b, t) at V + b*v_bs + t*v_rs + h*D  ->  Vt[b][h][d][Tp]
// grid = (Tp/64, H, B), block 256
// ------------------------------------------------------------------------------------
template <typename T, int D>
__global__ __launch_bounds__(256) void transpose_v_kernel(const T* __restrict__ V, long v_bs, long v_rs,
                                                          T* __restrict__ Vt, int Tk, int Tp, int H) {
    __shared__ T tile[64][D + 2];
    const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    for (int i = threadIdx.x; i < 64 * D; i += 256) {
        const int r = i / D, d = i - r * D;
        const int t = t0 + r;
        tile[r][d] = t < Tk ? V[(long)b * v_bs + (long)t * v_rs + (long)h * D + d] : Elem<T>::zero();
    }
    __syncthreads();
    T* out = Vt + ((long)(b * H + h) * D) * Tp + t0;
    for (int i = threadIdx.x; i < 64 * D; i += 256) {
        const int d = i >> 6, r = i & 63;
        out[(long)d * Tp + r] = tile[r][d];
    }
}

// ------------------------------------------------------------------------------------
// bf16 MFMA flash attention (non-causal; key-length mask; optional additive bias)
// ------------------------------------------------------------------------------------


// DMA (round 5): the K and V^T tiles go global -> LDS directly (global_load_lds, the XOR swizzle in the per-lane SOURCE address) into a DOUBLE
// buffer: tile t + 1 is requested right after the barrier that opens tile t, has the whole tile's MFMA + softmax time to land, and the loop needs
// ONE barrier per tile instead of two (register staging: load -> compute -> barrier -> ds_write -> barrier).  K rows past the valid length are
// clamped to the last valid row (their scores are masked below), V^T is zero-padded by the transpose; same arithmetic in both variants.
//
// VTR (round 6, DMA only): V is read where the qkv projection left it - rows [key][d] - and the tile sits in LDS row-major; the A operand of the
// PV product (V^T[d = 16 dt + l15][8 keys]) comes out of two ds_read_b64_tr_b16 per (dt, 32-key half): the 16 lanes of a group hand in the
// four 8-byte pieces of four key rows (lane i: row i >> 2, columns 4 (i & 3) .. + 3 of the 16-wide d block) and each receives ITS column of that
// 4 x 16 block.  Swizzle (conflict-free for the two 32-lane halves the instruction is served in: 8 key rows x 32 bytes = all 64 banks): the
// 32-byte group index of a row is XORed with (row >> 1) & 3 (D = 64: rows of equal parity share banks) / row & 7 (D = 128); with LDS-DMA the
// XOR sits in the per-lane SOURCE chunk, inside the row's own 128-B / 256-B line.  Rows past the valid length re-read the last valid row (finite
// values; their probabilities are exactly 0).  Removes transpose_v_kernel (6.4 ms per 64-session turn-batch) and its [B][H][D][Tp] scratch.
template <int D, bool DMA, bool VTR>
__global__ __launch_bounds__(256, D == 64 ? 3 : 2) void attn_mfma_kernel(AttnMfmaParams p) {
    static_assert(!VTR || DMA, "the direct-V variant is built on the LDS-DMA loop");
    constexpr int KT = 64;                       // keys per tile
    constexpr int KROWB = D * 2;                 // bytes per K row in LDS
    constexpr int KCH = KROWB / 16;              // 16-byte chunks per K row (8 or 16)
    constexpr int K_CHUNKS = KT * KCH;           // per tile
    constexpr int V_CHUNKS = D * 8;              // V^T tile: D rows x 128 bytes
    constexpr int KPT = K_CHUNKS / 256, VPT = V_CHUNKS / 256;
    constexpr int KK = D / 32;                   // k-steps of the QK^T MFMA
    constexpr int DT = D / 16;                   // output d fragments
    constexpr int TILE_LDS = KT * KROWB + D * 128;
    __shared__ __attribute__((aligned(16))) unsigned char lds[(DMA ? 2 : 1) * TILE_LDS];
    unsigned char* sK = lds;
    unsigned char* sV = lds + KT * KROWB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs, so with (q-block, head, session) in grid
    // order the q-blocks of one (session, head) - which all stream the SAME K / V^T rows - would land on 8 different L2s and
    // fetch them 8 times.  Re-map so that every q-block of a pair runs on one XCD (pair -> XCD pair % 8).
    int qblk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    {
        const int nq = gridDim.x, npairs = gridDim.y * gridDim.z;
        if ((npairs & 7) == 0) {
            const int L = blockIdx.x + nq * (blockIdx.y + gridDim.y * blockIdx.z);      // dispatch order (x fastest)
            const int xcd = L & 7, idx = L >> 3;
            const int pair = (idx / nq) * 8 + xcd;
            qblk = idx % nq;
            h = pair % (int)gridDim.y;
            b = pair / (int)gridDim.y;
        }
    }
    const int q0 = qblk * 128 + wave * 32;
    const int kv = p.kv_len ? min(p.kv_len[b], p.Tk) : p.Tk;

    // Q fragments (B operand of S^T = K.Q^T): Q[q = l15 + 16*qf][d = kk*32 + lg*8 .. +7]
    bf16x8_t qf[2][KK];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int q = q0 + f * 16 + l15;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (q < p.Tq)
                v = *reinterpret_cast<const u32x4_t*>(p.Q + (long)b * p.q_bs + (long)q * p.q_rs + (long)h * D + kk * 32 + lg * 8);
            qf[f][kk] = __builtin_bit_cast(bf16x8_t, v);
        }
    }
    f32x4_t oacc[2][DT];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oacc[f][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float mrun[2] = {-INFINITY, -INFINITY}, lrun[2] = {0.f, 0.f};

    const __hip_bfloat16* Kb = p.K + (long)b * p.k_bs + (long)h * D;
    const __hip_bfloat16* Vb = VTR ? p.V + (long)b * p.v_bs + (long)h * D : p.Vt + ((long)(b * p.H + h) * D) * p.Tp;
    const float* bias = p.bias ? p.bias + (long)b * p.bias_bs + (long)h * p.bias_hs : nullptr;

    u32x4_t gk[KPT], gv[VPT];
    auto load_tile = [&](int j0) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int c = tid + i * 256;
            const int row = c / KCH, ch = c % KCH;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (j0 + row < kv) v = *reinterpret_cast<const u32x4_t*>(Kb + (long)(j0 + row) * p.k_rs + ch * 8);
            gk[i] = v;
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c = tid + i * 256;
            const int row = c >> 3, ch = c & 7;
            gv[i] = *reinterpret_cast<const u32x4_t*>(Vb + (long)row * p.Tp + j0 + ch * 8);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int c = tid + i * 256;
            const int row = c / KCH, ch = c % KCH;
            *reinterpret_cast<u32x4_t*>(sK + row * KROWB + ((ch ^ (row & (KCH - 1))) << 4)) = gk[i];
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c = tid + i * 256;
            const int row = c >> 3, ch = c & 7;
            *reinterpret_cast<u32x4_t*>(sV + row * 128 + ((ch ^ (row & 7)) << 4)) = gv[i];
        }
    };

    // DMA variant: one wave instruction moves 64 lanes x 16 bytes = 1 KiB of LDS (K: 1024 / KROWB rows; V^T: 8 rows of 128 bytes), lane-linear in
    // LDS, so the lane that fills slot (row, pos) fetches the source chunk pos ^ (row & mask) - the same image store_tile writes
    constexpr int K_INSTR = KT * KROWB / 1024, V_INSTR = D * 128 / 1024;        // wave instructions per tile (8 / 16 and 8 / 16)
    constexpr int KI_PW = K_INSTR / 4, VI_PW = V_INSTR / 4;                      // ... per wave
    auto dma_tile = [&](int j0, int buf) {
        unsigned char* base = lds + buf * TILE_LDS;
#pragma unroll
        for (int i = 0; i < KI_PW; ++i) {
            const int blk = wave * KI_PW + i;                                    // 1 KiB block of the K tile
            const int row = blk * (1024 / KROWB) + lane / KCH, pos = lane % KCH;
            const int ch = pos ^ (row & (KCH - 1));
            const __hip_bfloat16* src = Kb + (long)min(j0 + row, kv - 1) * p.k_rs + ch * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(base + blk * 1024), 16, 0, 0);
        }
        if constexpr (VTR) {
#pragma unroll
            for (int i = 0; i < KI_PW; ++i) {
                const int blk = wave * KI_PW + i;                                // 1 KiB block of the V tile: 1024 / KROWB key rows
                const int row = blk * (1024 / KROWB) + lane / KCH, pos = lane % KCH;
                const int swz = D == 64 ? ((row >> 1) & 3) : (row & 7);
                const int ch = (((pos >> 1) ^ swz) << 1) | (pos & 1);            // the 32-byte group is swizzled, the 16-byte half inside it stays
                const __hip_bfloat16* src = Vb + (long)min(j0 + row, kv - 1) * p.v_rs + ch * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(base + KT * KROWB + blk * 1024), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < VI_PW; ++i) {
                const int blk = wave * VI_PW + i;                                // 8 rows (d) of the V^T tile
                const int row = blk * 8 + (lane >> 3), pos = lane & 7;
                const int ch = pos ^ (row & 7);
                const __hip_bfloat16* src = Vb + (long)row * p.Tp + j0 + ch * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(base + KT * KROWB + blk * 1024), 16, 0, 0);
            }
        }
    };
    // direct-V reads: this lane's piece of the 4 x 16 blocks - key row lg * 4 + (l15 >> 2) of a 16-key group, 8 bytes at column 4 (l15 & 3) of
    // the d block; the swizzle depends on the row modulo 8 only, so it is the same for every 16-key group
    [[maybe_unused]] int vtr_off[DT];
    if constexpr (VTR) {
        const int r = lg * 4 + (l15 >> 2);
        const int swz = D == 64 ? ((r >> 1) & 3) : (r & 7);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vtr_off[dt] = r * KROWB + ((dt ^ swz) << 5) + (l15 & 3) * 8;
    }
    const int ntiles = (kv + KT - 1) / KT;
    if constexpr (DMA) {
        if (ntiles > 0) dma_tile(0, 0);
    } else {
        if (ntiles > 0) {
            load_tile(0);
            store_tile();
        }
        __syncthreads();
    }
    for (int t = 0; t < ntiles; ++t) {
        const int j0 = t * KT;
        if constexpr (DMA) {
            __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));      // vmcnt(0): this wave's part of tile t has landed ...
            __builtin_amdgcn_s_barrier();                               // ... everyone's has; and every wave is done with tile t - 1's buffer
            if (t + 1 < ntiles) dma_tile(j0 + KT, (t + 1) & 1);
            sK = lds + (t & 1) * TILE_LDS;
            sV = sK + KT * KROWB;
        } else {
            if (t + 1 < ntiles) load_tile(j0 + KT);
        }
        // ---- S^T[key][q] for 4 key fragments x 2 query fragments
        f32x4_t s[4][2];
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) {
            s[kf][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            s[kf][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            const int row = kf * 16 + l15;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const bf16x8_t ka = *reinterpret_cast<const bf16x8_t*>(
                    sK + row * KROWB + (((kk * 4 + lg) ^ (row & (KCH - 1))) << 4));
                s[kf][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[0][kk], s[kf][0], 0, 0, 0);
                s[kf][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qf[1][kk], s[kf][1], 0, 0, 0);
            }
        }
        // ---- online softmax per query column (lane-local column l15, rows = keys lg*4+r)
        bf16x8_t pf[2][2];   // [qf][kt32]
        const bool fast = bias == nullptr && j0 + KT <= kv;      // whole tile valid, no additive bias (the Whisper encoder's case)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            if (fast) {
                // The softmax is the VALU bottleneck of a d_h = 64 flash attention (32 scores per lane per tile against 32
                // MFMAs per wave), so the common case is stripped to max / fma / v_exp_f32 / add per score and the
                // hardware pair conversion to bf16: the scale is applied inside the exponent's fma (max over the raw scores,
                // scale > 0), no masking, no bias.
                float tmax = fmaxf(fmaxf(s[0][f][0], s[0][f][1]), fmaxf(s[0][f][2], s[0][f][3]));
#pragma unroll
                for (int kf = 1; kf < 4; ++kf) tmax = fmaxf(tmax, fmaxf(fmaxf(s[kf][f][0], s[kf][f][1]), fmaxf(s[kf][f][2], s[kf][f][3])));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float mnew = fmaxf(mrun[f], tmax * p.scale_log2e);       // finite: the tile has 64 valid keys
                const float alpha = __builtin_amdgcn_exp2f(mrun[f] - mnew);   // mrun = -inf -> 0
                float psum = 0.f;
                float pv[4][4];
#pragma unroll
                for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pv[kf][r] = __builtin_amdgcn_exp2f(fmaf(s[kf][f][r], p.scale_log2e, -mnew));
                        psum += pv[kf][r];
                    }
                psum += __shfl_xor(psum, 16, 64);
                psum += __shfl_xor(psum, 32, 64);
                lrun[f] = lrun[f] * alpha + psum;
                mrun[f] = mnew;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) oacc[f][dt] *= alpha;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    u32x4_t w;
                    w[0] = pack_bf16x2_hw(pv[2 * k2][0], pv[2 * k2][1]);
                    w[1] = pack_bf16x2_hw(pv[2 * k2][2], pv[2 * k2][3]);
                    w[2] = pack_bf16x2_hw(pv[2 * k2 + 1][0], pv[2 * k2 + 1][1]);
                    w[3] = pack_bf16x2_hw(pv[2 * k2 + 1][2], pv[2 * k2 + 1][3]);
                    pf[f][k2] = __builtin_bit_cast(bf16x8_t, w);
                }
                continue;
            }
            const int q = q0 + f * 16 + l15;
            float tmax = -INFINITY;
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = j0 + kf * 16 + lg * 4 + r;
                    float v = s[kf][f][r] * p.scale_log2e;
                    if (bias && q < p.Tq && j < kv)
                        v += bias[(long)q * p.bias_rs + (p.bias_shift - q) + j] * 1.4426950408889634f;
                    v = j < kv ? v : -INFINITY;
                    s[kf][f][r] = v;
                    tmax = fmaxf(tmax, v);
                }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mnew = fmaxf(mrun[f], tmax);
            const float msafe = mnew == -INFINITY ? 0.f : mnew;
            const float alpha = exp2f(mrun[f] - msafe);   // mrun = -inf -> 0
            float psum = 0.f;
            float pv[4][4];
#pragma unroll
            for (int kf = 0; kf < 4; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = exp2f(s[kf][f][r] - msafe);
                    pv[kf][r] = e;
                    psum += e;
                }
            psum += __shfl_xor(psum, 16, 64);
            psum += __shfl_xor(psum, 32, 64);
            lrun[f] = lrun[f] * alpha + psum;
            mrun[f] = mnew;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) oacc[f][dt] *= alpha;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                unsigned short e[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = f32_to_bf16_bits(pv[2 * k2][r]);
                    e[4 + r] = f32_to_bf16_bits(pv[2 * k2 + 1][r]);
                }
                u32x4_t w;
                w[0] = e[0] | ((unsigned)e[1] << 16);
                w[1] = e[2] | ((unsigned)e[3] << 16);
                w[2] = e[4] | ((unsigned)e[5] << 16);
                w[3] = e[6] | ((unsigned)e[7] << 16);
                pf[f][k2] = __builtin_bit_cast(bf16x8_t, w);
            }
        }
        // ---- O^T[d][q] += V^T[d][key] . P^T[key][q]
        if constexpr (VTR) {
            typedef short v4s_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    // MFMA k index lg*8+e  <->  key k2*32 + (e<4 ? lg*4+e : 16+lg*4+e-4): one transposing read per 16-key group
                    const unsigned char* a0 = sV + vtr_off[dt] + k2 * 32 * KROWB;
                    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a0));
                    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a0 + 16 * KROWB));
                    const bf16x8_t va = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    oacc[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pf[0][k2], oacc[0][dt], 0, 0, 0);
                    oacc[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pf[1][k2], oacc[1][dt], 0, 0, 0);
                }
            }
        } else
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int row = dt * 16 + l15;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                // MFMA k index lg*8+e  <->  key k2*32 + (e<4 ? lg*4+e : 16+lg*4+e-4)
                const int c0 = (k2 * 64 + lg * 8) >> 4, o0 = (lg * 8) & 15;          // bytes k2*64 + lg*8
                const int c1 = (k2 * 64 + 32 + lg * 8) >> 4;
                const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(sV + row * 128 + ((c0 ^ (row & 7)) << 4) + o0);
                const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(sV + row * 128 + ((c1 ^ (row & 7)) << 4) + o0);
                u32x4_t w;
                w[0] = lo[0]; w[1] = lo[1]; w[2] = hi[0]; w[3] = hi[1];
                const bf16x8_t va = __builtin_bit_cast(bf16x8_t, w);
                oacc[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pf[0][k2], oacc[0][dt], 0, 0, 0);
                oacc[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pf[1][k2], oacc[1][dt], 0, 0, 0);
            }
        }
        if constexpr (!DMA) {
            __syncthreads();
            if (t + 1 < ntiles) {
                store_tile();
                __syncthreads();
            }
        }
    }
    // ---- write O[q][h*D + d]: lane holds d = dt*16 + lg*4 + r (4 consecutive) for q = l15
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int q = q0 + f * 16 + l15;
        if (q >= p.Tq) continue;
        const float inv = lrun[f] > 0.f ? 1.0f / lrun[f] : 0.f;
        __hip_bfloat16* o = p.O + (long)b * p.o_bs + (long)q * p.o_rs + (long)h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            u32x2_t w;
            w[0] = f32_to_bf16_bits(oacc[f][dt][0] * inv) | ((unsigned)f32_to_bf16_bits(oacc[f][dt][1] * inv) << 16);
            w[1] = f32_to_bf16_bits(oacc[f][dt][2] * inv) | ((unsigned)f32_to_bf16_bits(oacc[f][dt][3] * inv) << 16);
            *reinterpret_cast<u32x2_t*>(o + dt * 16 + lg * 4) = w;
        }
    }
}

bool attn_mfma_direct_v() {                      // S2S_ATTN_VTR=0: transpose_v + the V^T kernel of rounds 2 - 5 (A/B); needs the DMA loop
    static const bool on = [] {
        const char* e = getenv("S2S_ATTN_VTR");
        const char* d = getenv("S2S_ATTN_DMA");
        return (!e || atoi(e) != 0) && (!d || atoi(d) != 0);
    }();
    return on;
}

int attn_mfma(hipStream_t st, const AttnMfmaParams& p, int D) {
    const bool direct = p.Vt == nullptr;
    S2S_REQUIRE(direct ? (p.V != nullptr && p.v_rs % 8 == 0 && p.v_bs % 8 == 0 && ((uintptr_t)p.V & 15) == 0)
                       : (p.Tp % 64 == 0 && p.Tp >= p.Tk),
                "attention: needs V rows (16-byte aligned, strides multiples of 8) or V^T with Tp=%d a multiple of 64 and >= Tk=%d", p.Tp, p.Tk);
    dim3 grid(cdiv(p.Tq, 128), p.H, p.B);
    ProfScope prof(st, S2S_PROF_ATTN_MFMA, 4.0 * p.B * p.H * (double)p.Tq * p.Tk * D,
                   2.0 * p.B * p.H * ((double)p.Tq * D * 2 + (double)p.Tk * D * 2));
    static const bool dma = [] {                 // S2S_ATTN_DMA=0: the register-staged loop of rounds 2 - 4 (A/B)
        const char* e = getenv("S2S_ATTN_DMA");
        return !e || atoi(e) != 0;
    }();
    if (direct && D == 64) launch_k(attn_mfma_kernel<64, true, true>, grid, dim3(256), 0, st, p);
    else if (direct && D == 128) launch_k(attn_mfma_kernel<128, true, true>, grid, dim3(256), 0, st, p);
    else if (direct) return fail(S2S_ERR_UNSUPPORTED, "attention: MFMA path supports head_dim 64/128, got %d", D);
    else if (D == 64 && dma) launch_k(attn_mfma_kernel<64, true, false>, grid, dim3(256), 0, st, p);
    else if (D == 128 && dma) launch_k(attn_mfma_kernel<128, true, false>, grid, dim3(256), 0, st, p);
    else if (D == 64) launch_k(attn_mfma_kernel<64, false, false>, grid, dim3(256), 0, st, p);
    else if (D == 128) launch_k(attn_mfma_kernel<128, false, false>, grid, dim3(256), 0, st, p);
    else return fail(S2S_ERR_UNSUPPORTED, "attention: MFMA path supports head_dim 64/128, got %d", D);
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

int transpose_v(hipStream_t st, const void* V, long v_bs, long v_rs, void* Vt, int B, int H, int D, int Tk,
                int Tp, int dtype) {
    S2S_REQUIRE(Tp % 64 == 0 && Tp >= Tk, "transpose_v: bad Tp");
    dim3 grid(Tp / 64, H, B);
#define S2S_TV(TT, DV) transpose_v_kernel<TT, DV><<<grid, 256, 0, st>>>((const TT*)V, v_bs, v_rs, (TT*)Vt, Tk, Tp, H)
    if (dtype == S2S_BF16 && D == 64) S2S_TV(__hip_bfloat16, 64);
    else if (dtype == S2S_BF16 && D == 128) S2S_TV(__hip_bfloat16, 128);
    else return fail(S2S_ERR_UNSUPPORTED, "transpose_v: unsupported dtype/head_dim");
#undef S2S_TV
    S2S_LAUNCH_CHECK();
    return S2S_OK;
}

}  // namespace s2s

using namespace s2s;

extern "C" int32_t s2s_op_attn_step_tune(int32_t share) {
    g_attn_step_share = share < 0 ? -1 : (share & 3);      // bit 0: grouped heads share a workgroup; bit 1: one wave per item for caches <= 32 positions
    return S2S_OK;
}

extern "C" int32_t s2s_op_attn_step(void* stream, const void* qkv, int64_t qkv_rs, void* K, void* V, int64_t kv_bs, int64_t kv_rs, void* O, int64_t o_bs,
                                    const float* qw, const float* kw, float eps, const float* cos_t, const float* sin_t, const int32_t* pos_dev,
                                    int32_t B, int32_t H, int32_t Hkv, int32_t D, int32_t kv_cap, int32_t window, float scale, int32_t o_frag, int32_t dtype) {
    S2S_REQUIRE(qkv && K && V && O && pos_dev, "attn_step: null argument");
    S2S_REQUIRE(dtype == S2S_BF16 || dtype == S2S_F32, "attn_step: bad dtype");
    AttnStepParams a{};
    a.qkv = qkv; a.qkv_rs = qkv_rs; a.K = K; a.V = V; a.kv_bs = kv_bs; a.kv_rs = kv_rs; a.O = O; a.o_bs = o_bs;
    a.qw = qw; a.kw = kw; a.eps = eps; a.cosT = cos_t; a.sinT = sin_t; a.pos = pos_dev;
    a.B = B; a.H = H; a.Hkv = Hkv; a.D = D; a.kv_cap = kv_cap; a.window = window; a.scale = scale; a.o_frag = o_frag != 0;
    return attn_step((hipStream_t)stream, a, dtype);
}

// Public op: self/cross attention over token-major buffers.
//   q: [B][Tq][q_rs] with head h at columns h*D.. ; k, v: [B][Tk][kv_rs] likewise (Hkv heads);
//   o: [B][Tq][H*D].  use_mfma selects the bf16 flash kernel (needs vt_scratch [B][Hkv==H][D][Tp]).
extern "C" int32_t s2s_op_attention(void* stream, const void* q, int64_t q_rs, const void* k, const void* v,
                                    int64_t kv_rs, void* o, const int32_t* kv_len_dev, const float* bias,
                                    int32_t B, int32_t H, int32_t Hkv, int32_t Tq, int32_t Tk, int32_t D,
                                    float scale, int32_t causal, int32_t window, int32_t dtype,
                                    int32_t use_mfma, void* vt_scratch) {
    hipStream_t st = (hipStream_t)stream;
    if (use_mfma) {
        S2S_REQUIRE(dtype == S2S_BF16 && H == Hkv && !causal && window <= 0 && vt_scratch, "attention: MFMA path preconditions");
        const int Tp = cdiv(Tk, 64) * 64;
        const bool direct = attn_mfma_direct_v() && kv_rs % 8 == 0 && ((uintptr_t)v & 15) == 0;
        if (!direct) {
            int rc = transpose_v(st, v, (long)Tk * kv_rs, kv_rs, vt_scratch, B, H, D, Tk, Tp, dtype);
            if (rc) return rc;
        }
        AttnMfmaParams p{};
        p.Q = (const __hip_bfloat16*)q; p.q_bs = (long)Tq * q_rs; p.q_rs = q_rs;
        p.K = (const __hip_bfloat16*)k; p.k_bs = (long)Tk * kv_rs; p.k_rs = kv_rs;
        p.Vt = direct ? nullptr : (const __hip_bfloat16*)vt_scratch;
        p.V = (const __hip_bfloat16*)v; p.v_bs = (long)Tk * kv_rs; p.v_rs = kv_rs;
        p.O = (__hip_bfloat16*)o; p.o_bs = (long)Tq * H * D; p.o_rs = (long)H * D;
        p.bias = bias; p.bias_bs = (long)H * Tq * (2 * Tk - 1); p.bias_hs = (long)Tq * (2 * Tk - 1);
        p.bias_rs = 2 * Tk - 1; p.bias_shift = Tk - 1;
        p.kv_len = kv_len_dev; p.Tk = Tk; p.Tp = Tp; p.Tq = Tq; p.H = H; p.B = B;
        p.scale_log2e = scale * 1.4426950408889634f;
        return attn_mfma(st, p, D);
    }
    AttnRowsParams p{};
    p.Q = q; p.q_bs = (long)Tq * q_rs; p.q_rs = q_rs;
    p.K = k; p.V = v; p.kv_bs = (long)Tk * kv_rs; p.kv_rs = kv_rs;
    p.O = o; p.o_bs = (long)Tq * H * D; p.o_rs = (long)H * D;
    p.bias = bias; p.bias_bs = (long)H * Tq * (2 * Tk - 1); p.bias_hs = (long)Tq * (2 * Tk - 1);
    p.bias_rs = 2 * Tk - 1; p.bias_shift = Tk - 1;
    p.kv_len = kv_len_dev; p.kv_len_const = Tk;
    p.B = B; p.H = H; p.Hkv = Hkv; p.Tq = Tq; p.D = D;
    p.causal = causal; p.causal_off = Tk - Tq; p.window = window; p.scale = scale;
    return attn_rows(st, p, dtype);
}
