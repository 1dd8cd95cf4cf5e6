// Decode-step projections, 33 .. 128 rows per slab (bf16, one row per live session): the column-block / K-slice kernel.
//
// Why a second kernel (round 3).  gemm_dec_kernel (gemm_dec.hip) gives one 8-wave workgroup 16 output columns over the whole of K,
// so EVERY workgroup reads the whole activation matrix: at M = 64 the talker's qkv projection moves 16.8 MB of weights and
// 64 MB of activations from L2 into the CUs, and the measured cost per launch follows the per-CU load-path bytes (about
// 70 GB/s per CU marginal, profiles/r02_microbench_decode_gemm.txt: 160 KB per CU at K = 1024 -> 4.0 us, 320 KB at K = 2048 ->
// 6.3 us whether 128 or 256 workgroups run), not HBM.  /opt/skills/guides/cdna_hip_programming.md:444-452 names the cure for
// this regime: a workgroup owns a BLOCK of columns so that an activation byte fetched by the CU feeds BN / 16 MFMAs
// instead of one, the activations go through LDS once per workgroup, K is split ACROSS workgroups so that the launch still
// fills the chip, and the K-slices are combined in a fixed order by the slice that arrives last.
//
// Work item = (column block cb of BN = 64 * NT columns, K-slice ks); grid.x = column blocks x SPLITK, grid.y = 128-row slabs.
//   * 4 waves; wave w owns the 16 * NT columns [(cb * 4 + w) * NT * 16, ...) of the block (GLU: the same gate and up tiles).
//     Its weights stream HBM -> registers exactly once (fragment-major 1 KiB blocks, non-temporal unless w_cached),
//     double-buffered in chunks of KC k-steps; no other wave needs them, so they never touch LDS.
//   * the activation chunk (MT row tiles x KC k-steps, fragment-major 1 KiB blocks) is fetched ONCE per workgroup - each wave
//     loads a quarter of the blocks and writes them to an LDS image that IS the MFMA operand order, two chunks deep - and every
//     wave reads its operands with ds_read_b128 at lane * 16 (conflict free, linear).
//   * the MFMA runs with the WEIGHTS as the row operand: a lane then holds 4 consecutive output columns of ONE activation
//     row (D[n = 4 * (lane >> 4) + r][m = lane & 15]), so the epilogue needs no LDS transpose - f32 residual rows are 16-byte
//     accesses, bf16 rows 8-byte ones, a fragment-major output block is written whole - and per-row quantities (the
//     reciprocal RMS, the sums of squares) are one value per lane and row tile.
//   * every load is unconditional (chunks past the slice read a harmless block that stays in L1), so the compiler's vmcnt
//     accounting is exact and one chunk of weights and activations is always in flight behind the MFMAs.
//   * SPLITK > 1: a slice stores its f32 partial tile write-through (sc1, 16 B per lane, the accumulator layout), drains,
//     and draws a ticket; the LAST arriver reads all SPLITK slabs (sc1 loads) and adds them in slice order 0 .. SPLITK-1 -
//     deterministic, no atomics on data, independent of placement and arrival order (cdna_hip_programming.md:454-480) -
//     then runs the epilogue and resets the ticket for the next launch.
// Epilogues = gemm_dec.hip's: MODE 0 bf16 (row- / fragment-major, optional SwiGLU), MODE 1 f32, MODE 2 X += out with the
// bf16 copy and ONE partial sum of squares per column block (n_parts = N / BN instead of N / 16).
#include "ops.h"

#include <cmath>
#include <type_traits>

namespace s2s {

namespace {

constexpr int D2_NW = 4;

__device__ __forceinline__ const char* uptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

__host__ __device__ __forceinline__ long frag_idx(int m, int k, int ksteps) {
    return ((((long)(m >> 4) * ksteps + (k >> 5)) * 4 + ((k >> 3) & 3)) * 16 + (m & 15)) * 8 + (k & 7);
}

struct Dec2Plan {
    int ncb;            // column blocks
    int splitk;         // K-slices per column block
    int nch;            // chunks of KC k-steps per slice
    float* slabs;       // [row slab][cb][slice][wave][MT * NS][256] f32 partial tiles
    unsigned* cnt;      // [row slab][cb] arrival tickets (zero between launches)
    unsigned long long* dbg;   // development: 8 phase timestamps (100 MHz) per workgroup, or null
};

template <int MT, int NT, bool GLU, int KC, bool NTL, int NCH>      // NCH: chunks per slice known at compile time (1, 2: every load up front, straight-line), 0: any
__global__ __launch_bounds__(D2_NW * 64) void gemm_dec2_kernel(DecGemm a, Dec2Plan p) {
    constexpr int NS = NT * (GLU ? 2 : 1);            // weight streams per wave
    constexpr int APW = MT * KC / D2_NW;              // activation blocks a wave stages per chunk
    constexpr int ABYTES = MT * KC * 1024;            // one activation chunk in LDS
    static_assert((MT * KC) % D2_NW == 0, "activation blocks must divide among the waves");
    // ONE LDS object (a second one makes hipcc drain the load queue before every ds_read, cdna_hip_programming.md:496)
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * ABYTES + (1 + D2_NW) * MT * 16 * 4 + 16];
    float* rstd_s = reinterpret_cast<float*>(lds + 2 * ABYTES);               // [MT * 16]
    float* ssw = rstd_s + MT * 16;                                             // [4][MT * 16]
    unsigned* flag_s = reinterpret_cast<unsigned*>(ssw + D2_NW * MT * 16);

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rs = blockIdx.y;
    // development timestamps: kept in scalar registers and written out at the END (a store in flight would change what the wait
    // counts of the phases being measured mean)
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto stamp = [&](int i) {
        if (p.dbg) {
            __builtin_amdgcn_sched_barrier(0);
            ts[i] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto stamps_out = [&]() {
        if (p.dbg && tid == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p.dbg[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + i] = ts[i];
        }
    };
    stamp(0);
    if (gridDim.y > 1) {             // M > 128: independent 128-row problems on shifted row pointers
        const long r0 = (long)rs * 128;
        a.A += r0 * (long)a.K;
        if (a.mode == 0) a.C = reinterpret_cast<__hip_bfloat16*>(a.C) + r0 * a.ldc;
        else if (a.mode == 1) a.C = reinterpret_cast<float*>(a.C) + r0 * a.ldc;
        else {
            a.X += r0 * a.ldx;
            if (a.XB) a.XB += r0 * a.ldxb;
            if (a.ss_out) a.ss_out += r0;
        }
        if (a.ss_in) a.ss_in += r0 * (a.ss_rs > 0 ? a.ss_rs : 1);
        a.M = 128;
    }
    const int cb = blockIdx.x % p.ncb, ks = blockIdx.x / p.ncb;
    const int steps = a.K >> 5;
    const int nch = NCH > 0 ? NCH : p.nch;
    const int s0 = ks * nch * KC;                       // first k-step of this slice
    const int ntiles = (a.N + 15) >> 4;
    const int mtiles = (a.M + 15) >> 4;
    const int t0 = (cb * D2_NW + wave) * NT;            // this wave's first column tile

    // reciprocal RMS of the input rows (gamma is folded into W): 4 threads per row, <= 64 partial sums per row.  The loads are issued
    // BEHIND the first weight / activation loads (the HBM latency of the weights is the long pole of the launch - phase timestamps,
    // profiles/r03_decode_gemm_phases.txt - so they go first; everything is waited for together before the first MFMA anyway).
    // Branch-free on purpose - without ss_in the same loads read row 0 of A and the result is ignored: a branch around loads makes
    // hipcc's wait counts conservative (vmcnt(0)) for everything issued after it.
    constexpr int RH = (MT * 16 * 4 + D2_NW * 64 - 1) / (D2_NW * 64);
    constexpr int RP = 16;
    float sv[RH][RP];
    auto sv_load = [&]() {
        const bool has = a.ss_in != nullptr;
        const float* base = has ? a.ss_in : reinterpret_cast<const float*>(a.A);
        const long ssrs = has ? (a.ss_rs > 0 ? a.ss_rs : 1) : 0, sst = has ? a.ss_stride : 0;
        const int np1 = has ? a.n_parts_in - 1 : 0;
#pragma unroll
        for (int h = 0; h < RH; ++h) {
            const int e = h * D2_NW * 64 + tid;
            const float* src = base + (long)min(e >> 2, a.M - 1) * ssrs;
#pragma unroll
            for (int j = 0; j < RP; ++j) sv[h][j] = src[(long)min((e & 3) + 4 * j, np1) * sst];
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto rstd_reduce = [&]() {
#pragma unroll
        for (int h = 0; h < RH; ++h) {
            const int e = h * D2_NW * 64 + tid;
            const int row = e >> 2, sub = e & 3;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const unsigned keep = sub + 4 * j < a.n_parts_in ? 0xffffffffu : 0u;      // (a mask, not a select: a select lets hipcc sink each load under a branch)
                v += __uint_as_float(__float_as_uint(sv[h][j]) & keep);
            }
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            if (sub == 0 && row < MT * 16) rstd_s[row] = a.ss_in ? rsqrtf(v * a.inv_dim + a.eps) : 1.0f;
        }
    };

    const char* wt[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const int t = min(t0 + (q % NT), ntiles - 1) + (q >= NT ? ntiles : 0);
        wt[q] = reinterpret_cast<const char*>(a.W + ((long)t * steps + s0) * 512);
    }
    // this wave's share of a chunk's activation blocks: block j -> (k-step u = blk / MT, row tile i = blk % MT)
    const char* at[APW];
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        const int blk = wave * APW + j;
        const int u = blk / MT, i = blk % MT;
        at[j] = reinterpret_cast<const char*>(a.A + ((long)min(i, mtiles - 1) * steps + s0 + u) * 512);
    }
    const unsigned loff = lane * 16;

    f32x4_t acc[MT][NS];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < NS; ++q) acc[i][q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    u32x4_t fw[2][KC][NS], ra[2][APW];

    // every load below is unconditional and straight-line inside its region (prologue, steady-state trip, tail), so the compiler's
    // vmcnt accounting is exact: a wait for chunk c never drains the loads of chunks c + 1 / c + 2 issued after it
    auto load_w = [&](int c, int buf) {
        const long off = (long)c * KC * 1024;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const char* b = uptr(wt[q] + off);
#pragma unroll
            for (int u = 0; u < KC; ++u) {
                if constexpr (NTL) fw[buf][u][q] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(b + u * 1024 + loff));
                else fw[buf][u][q] = *reinterpret_cast<const u32x4_t*>(b + u * 1024 + loff);
            }
        }
        __builtin_amdgcn_sched_barrier(0);       // issued HERE: the scheduler must not sink loads behind the MFMAs that follow
    };
    auto load_a = [&](int c, int rb) {
        const long off = (long)c * KC * 1024;
#pragma unroll
        for (int j = 0; j < APW; ++j) ra[rb][j] = *reinterpret_cast<const u32x4_t*>(uptr(at[j] + off) + loff);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto store_a = [&](int rb, int buf) {
#pragma unroll
        for (int j = 0; j < APW; ++j) *reinterpret_cast<u32x4_t*>(lds + buf * ABYTES + (wave * APW + j) * 1024 + loff) = ra[rb][j];
    };
    // MFMAs of the chunks [b0, b0 + nb) as ONE sequence of k-steps (LDS image / weight registers b hold chunk b).  One wave per SIMD:
    // nothing but the wave's own instruction stream hides LDS latency, and hipcc reads each fragment right before its MFMA - so the
    // fragments of k-step u + PD are read while the MFMAs of k-step u issue, with the order pinned by scheduling barriers.
    auto compute_seq = [&](auto b0c, auto nbc) {
        constexpr int b0 = decltype(b0c)::value, nb = decltype(nbc)::value;
        constexpr int PD = MT >= 8 ? 1 : 2;          // k-steps of fragments in flight ahead of the MFMAs
        constexpr int NU = nb * KC;
        u32x4_t fa[PD + 1][MT];
        auto rd = [&](int u, int i) {
            const int b = b0 + u / KC, uu = u % KC;
            return *reinterpret_cast<const u32x4_t*>(lds + b * ABYTES + (uu * MT + i) * 1024 + loff);
        };
#pragma unroll
        for (int u = 0; u < PD && u < NU; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[u][i] = rd(u, i);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + PD < NU) {
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[(u + PD) % (PD + 1)][i] = rd(u + PD, i);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw[b0 + u / KC][u % KC][q]),
                                                                        __builtin_bit_cast(bf16x8_t, fa[u % (PD + 1)][i]), acc[i][q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    auto compute = [&](int buf) {
        if (buf == 0) compute_seq(I0{}, I1{});
        else compute_seq(I1{}, I1{});
    };

    if constexpr (NCH == 1) {
        load_w(0, 0);
        load_a(0, 0);
        sv_load();
        rstd_reduce();
        stamp(7);
        store_a(0, 0);
        __syncthreads();
        stamp(1);
        compute(0);
    } else if constexpr (NCH == 2) {
        // the whole slice in flight at once: activations (L2) first, then the weight stream; ONE barrier
        load_w(0, 0);
        load_w(1, 1);
        load_a(0, 0);
        load_a(1, 1);
        sv_load();
        rstd_reduce();
        stamp(7);
        store_a(0, 0);
        store_a(1, 1);
        __syncthreads();
        stamp(1);
        compute_seq(I0{}, I2{});
    } else if (nch == 1) {
        load_w(0, 0);
        load_a(0, 0);
        sv_load();
        rstd_reduce();
        stamp(7);
        store_a(0, 0);
        __syncthreads();
        stamp(1);
        compute(0);
    } else {
        // two chunks in flight from the start; in the steady state the activations run two chunks ahead of the MFMAs (registers
        // -> LDS one chunk ahead) and the weights one to two chunks ahead
        load_w(0, 0);
        load_a(0, 0);
        load_w(1, 1);
        load_a(1, 1);
        sv_load();
        rstd_reduce();
        stamp(7);
        store_a(0, 0);
        __syncthreads();
        stamp(1);
        int c = 0;
        for (; c + 3 < nch; c += 2) {
            load_a(c + 2, 0);
            compute(0);
            load_w(c + 2, 0);
            store_a(1, 1);
            __syncthreads();
            load_a(c + 3, 1);
            compute(1);
            load_w(c + 3, 1);
            store_a(0, 0);
            __syncthreads();
        }
        if (nch - c == 2) {
            compute(0);
            store_a(1, 1);
            __syncthreads();
            compute(1);
        } else {
            load_a(c + 2, 0);
            compute(0);
            load_w(c + 2, 0);
            store_a(1, 1);
            __syncthreads();
            compute(1);
            store_a(0, 0);
            __syncthreads();
            compute(0);
        }
    }

    stamp(2);

    // ---- K-slices: publish, ticket, last arriver adds the slabs in slice order --------------------------------------
    if (p.splitk > 1) {
        const int tile = rs * p.ncb + cb;
        const long slab_f = (long)MT * NS * 256;                                // floats per wave per slice
        const long tile_f = slab_f * D2_NW * p.splitk;
        float* base = p.slabs + (long)tile * tile_f;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(tile_f * 4), 0x00020000);
        const unsigned my = (unsigned)(((long)ks * D2_NW + wave) * slab_f * 4) + lane * 16;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < NS; ++q)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][q]), rsrc, my + (unsigned)((i * NS + q) * 1024), 0, 16);   // sc1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains its write-through stores
        __syncthreads();
        stamp(3);
        if (tid == 0) *flag_s = __hip_atomic_fetch_add(p.cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        stamp(4);
        if (*flag_s != (unsigned)(p.splitk - 1)) {
            stamps_out();
            return;
        }
        if (tid == 0) __hip_atomic_store(p.cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ticket for the next launch
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < NS; ++q) acc[i][q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // slabs in slice order, SG slices' loads in flight at a time (slices past the end re-read the last one and are masked off)
        constexpr int SG = MT * NS >= 16 ? 1 : (MT * NS >= 8 ? 2 : 4);
        for (int sb = 0; sb < p.splitk; sb += SG) {
            u32x4_t part[SG][MT][NS];
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                const unsigned so = (unsigned)(((long)min(sb + g, p.splitk - 1) * D2_NW + wave) * slab_f * 4) + lane * 16;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int q = 0; q < NS; ++q) part[g][i][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, so + (unsigned)((i * NS + q) * 1024), 0, 16);   // sc1
            }
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                const unsigned keep = sb + g < p.splitk ? 0xffffffffu : 0u;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int q = 0; q < NS; ++q) {
                        u32x4_t v = part[g][i][q];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] &= keep;
                        acc[i][q] += __builtin_bit_cast(f32x4_t, v);
                    }
            }
        }
        stamp(5);
    }

    // ---- epilogue on the accumulator layout: lane = (row m = 16 i + l15, columns n .. n + 3 with n = tile * 16 + 4 lg) ----
    float ssq[MT], ssm[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) ssq[i] = ssm[i] = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = i * 16 + l15;
        const bool mok = m < a.M;
        const float rsd = rstd_s[m];
#pragma unroll
        for (int q = 0; q < NT; ++q) {
            const int tile = t0 + q;
            const int n = tile * 16 + lg * 4;
            const bool full = mok && tile < ntiles && n + 3 < a.N;
            const bool some = mok && tile < ntiles && n < a.N;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][q][r] * rsd;
            if constexpr (GLU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float up = acc[i][NT + q][r] * rsd;
                    v[r] = v[r] / (1.0f + __expf(-v[r])) * up;
                }
            } else {
                if (a.bias && some) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += a.bias[min(n + r, a.N - 1)];
                }
                if (a.act == S2S_ACT_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                } else if (a.act == S2S_ACT_SILU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + __expf(-v[r]));
                } else if (a.act == S2S_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
            }
            if (a.mode == 0) {
                __hip_bfloat16* C = reinterpret_cast<__hip_bfloat16*>(a.C);
                if (full) {
                    u32x2_t pk;
                    pk[0] = pack_bf16x2_hw(v[0], v[1]);
                    pk[1] = pack_bf16x2_hw(v[2], v[3]);
                    *reinterpret_cast<u32x2_t*>(C + (a.c_frag ? frag_idx(m, n, (int)(a.ldc >> 5)) : (long)m * a.ldc + n)) = pk;
                } else if (some) {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < a.N) Elem<__hip_bfloat16>::st(C + (a.c_frag ? frag_idx(m, n + r, (int)(a.ldc >> 5)) : (long)m * a.ldc + n + r), v[r]);
                }
            } else if (a.mode == 1) {
                float* C = reinterpret_cast<float*>(a.C) + (long)m * a.ldc + n;
                if (full && ((a.ldc & 3) == 0)) *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
                else if (full && ((a.ldc & 1) == 0)) {       // e.g. 51 866 logits columns: rows stay 8-byte aligned
                    *reinterpret_cast<float2*>(C) = make_float2(v[0], v[1]);
                    *reinterpret_cast<float2*>(C + 2) = make_float2(v[2], v[3]);
                } else if (some) {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < a.N) C[r] = v[r];
                }
            } else {
                float xn[4] = {0.f, 0.f, 0.f, 0.f};
                if (full && ((a.ldx & 3) == 0)) {
                    float* xp = a.X + (long)m * a.ldx + n;
                    if (a.x_overwrite) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) xn[r] = v[r];
                    } else {
                        const float4 xo = *reinterpret_cast<const float4*>(xp);
                        xn[0] = xo.x + v[0]; xn[1] = xo.y + v[1]; xn[2] = xo.z + v[2]; xn[3] = xo.w + v[3];
                    }
                    *reinterpret_cast<float4*>(xp) = make_float4(xn[0], xn[1], xn[2], xn[3]);
                    if (a.XB) {
                        u32x2_t pk;
                        pk[0] = pack_bf16x2_hw(xn[0], xn[1]);
                        pk[1] = pack_bf16x2_hw(xn[2], xn[3]);
                        *reinterpret_cast<u32x2_t*>(a.XB + (a.xb_frag ? frag_idx(m, n, (int)(a.ldxb >> 5)) : (long)m * a.ldxb + n)) = pk;
                    }
                } else if (some) {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < a.N) {
                            float* xp = a.X + (long)m * a.ldx + n + r;
                            xn[r] = a.x_overwrite ? v[r] : *xp + v[r];
                            *xp = xn[r];
                            if (a.XB) Elem<__hip_bfloat16>::st(a.XB + (a.xb_frag ? frag_idx(m, n + r, (int)(a.ldxb >> 5)) : (long)m * a.ldxb + n + r), xn[r]);
                        }
                }
                ssq[i] += (xn[0] * xn[0] + xn[1] * xn[1]) + (xn[2] * xn[2] + xn[3] * xn[3]);
                ssm[i] += (xn[0] + xn[1]) + (xn[2] + xn[3]);
            }
        }
    }
    if (a.mode == 2 && a.ss_out) {
        // one partial per column block: lanes of a row (4 lane groups) -> waves of the block, fixed order
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float q = ssq[i];
            q += __shfl_xor(q, 16, 64);
            q += __shfl_xor(q, 32, 64);
            if (lg == 0) ssw[wave * MT * 16 + i * 16 + l15] = q;
        }
        __syncthreads();
        if (tid < MT * 16 && tid < a.M) {
            float q = 0.f;
#pragma unroll
            for (int w = 0; w < D2_NW; ++w) q += ssw[w * MT * 16 + tid];
            a.ss_out[(long)(a.ss_pairs ? 2 * cb : cb) * a.ss_stride + tid] = q;
        }
        if (a.ss_pairs) {        // plain sums for a LayerNorm consumer: the same fixed-order reduction through the same LDS rows
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float q = ssm[i];
                q += __shfl_xor(q, 16, 64);
                q += __shfl_xor(q, 32, 64);
                if (lg == 0) ssw[wave * MT * 16 + i * 16 + l15] = q;
            }
            __syncthreads();
            if (tid < MT * 16 && tid < a.M) {
                float q = 0.f;
#pragma unroll
                for (int w = 0; w < D2_NW; ++w) q += ssw[w * MT * 16 + tid];
                a.ss_out[(long)(2 * cb + 1) * a.ss_stride + tid] = q;
            }
        }
    }
    if (p.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(6);
        stamps_out();
    }
}

struct Tune {
    int enable = 1;       // 0: always the 16-column kernel
    int splitk = 0;       // 0: chosen from the shape
    int nt = 0;           // 0: chosen from the shape
    int min_m = 33;
    unsigned long long* dbg = nullptr;
};
Tune& tune() {
    static Tune t = [] {
        Tune x;
        if (const char* e = getenv("S2S_DEC2")) x.enable = atoi(e);
        if (const char* e = getenv("S2S_DEC2_SPLITK")) x.splitk = atoi(e);
        if (const char* e = getenv("S2S_DEC2_NT")) x.nt = atoi(e);
        return x;
    }();
    return t;
}

constexpr size_t D2_CNT_BYTES = 16384;       // 4096 tickets

// default workspace of the public op / of callers that pass none (one per process: single-stream use only)
struct DefaultWs { void* p = nullptr; size_t bytes = 0; };
DefaultWs& default_ws() {
    static DefaultWs w;
    return w;
}

template <int MT, int NT, bool GLU, int KC, int NCH>
void launch3(hipStream_t st, const DecGemm& a, const Dec2Plan& p, int rslabs) {
    const dim3 grid(p.ncb * p.splitk, rslabs), block(D2_NW * 64);
    if (a.w_cached) launch_k(gemm_dec2_kernel<MT, NT, GLU, KC, false, NCH>, grid, block, 0, st, a, p);
    else launch_k(gemm_dec2_kernel<MT, NT, GLU, KC, true, NCH>, grid, block, 0, st, a, p);
}
template <int MT, int NT, bool GLU, int KC>
void launch2(hipStream_t st, const DecGemm& a, const Dec2Plan& p, int rslabs) {
    if (p.nch == 1) launch3<MT, NT, GLU, KC, 1>(st, a, p, rslabs);
    else if (p.nch == 2) launch3<MT, NT, GLU, KC, 2>(st, a, p, rslabs);
    else launch3<MT, NT, GLU, KC, 0>(st, a, p, rslabs);
}

}  // namespace

int dec2_ensure_default_ws() {
    DefaultWs& w = default_ws();
    if (w.p) return S2S_OK;
    const size_t bytes = (size_t)64 << 20;
    S2S_HIP_CHECK(hipMalloc(&w.p, bytes));
    S2S_HIP_CHECK(hipMemset(w.p, 0, bytes));
    w.bytes = bytes;
    return S2S_OK;
}

void* dec2_default_ws(size_t* bytes) {
    if (bytes) *bytes = default_ws().bytes;
    return default_ws().p;
}

void decw_set_debug(unsigned long long* dbg);      // gemm_decw.hip
void dec2_set_debug(unsigned long long* dbg) { tune().dbg = dbg; }

void dec2_set_tune(int enable, int splitk, int nt) {
    Tune& t = tune();
    t.enable = enable;
    t.splitk = splitk;
    t.nt = nt;
}

// The plan for this shape, or ncb = 0 when the 16-column kernel should run it.
struct Dec2Choice { int MT, NT, KC, ncb, splitk, nch, rslabs; };
static Dec2Choice dec2_choose(const DecGemm& a) {
    Dec2Choice c{};
    const Tune& t = tune();
    if (!t.enable || a.nf > 0 || a.M < t.min_m || !a.a_frag || (a.M > 128 && a.M % 128 != 0) || a.ln) return c;     // (LayerNorm prologue: 16-column kernel)
    if (a.glu && a.N % 16 != 0) return c;
    if (a.ss_in && a.n_parts_in > 64) return c;        // (the RMS prologue prefetches at most 64 partial sums per row)
    if (a.K % 32 != 0 || a.K < 256) return c;
    // Measured (profiles/r03_microbench_decode_gemm.txt): the 16-column kernel wins while ONE round of its N / 16 workgroups covers the
    // chip and K is short (its cost is 2.2 us + 2.2 us per 1024 of K per round); this kernel wins on long K (>= 4096; >= 3072 above 64
    // rows) and when the 16-column kernel would need more than one round (N / 16 > 256 workgroups).  enable = 2 forces it everywhere.
    if (t.enable == 1 && t.splitk == 0 && t.nt == 0) {
        const int wgs16 = (a.N + 15) / 16;
        const bool long_k = a.K >= (a.M > 64 ? 3072 : 4096);
        if (!long_k && wgs16 <= 256) return c;
    }
    const int Mslab = a.M > 128 ? 128 : a.M;
    c.rslabs = a.M > 128 ? a.M / 128 : 1;
    c.MT = Mslab <= 64 ? 4 : 8;
    const int ntiles = (a.N + 15) / 16;
    int nt = t.nt > 0 ? t.nt : 1;
    if (a.glu) nt = 1;
    // chunk length: 8 k-steps for 64-row slabs, 4 for 128-row slabs.  (With two weight streams per wave - SwiGLU, 128-column blocks -
    // the 8-step chunk loop needs 272 VGPRs and spills 32; 4-step chunks removed the spill and measured 12 % SLOWER on the talker's
    // gate/up projection - twice the barriers - so the spill stays.)
    c.KC = c.MT == 4 ? 8 : 4;
    const int steps = a.K / 32;
    const int KS = c.MT == 4 ? 8 : 4;              // granularity of the K-slice choice (k-steps): the sweep's units
    if (steps % KS != 0) return c;
    const int chunks = steps / KS;
    if (t.nt == 0 && !a.glu && c.MT == 8 && ntiles >= 2 * D2_NW * 48) nt = 2;       // 128-column blocks when there are enough of them
    c.NT = nt;
    c.ncb = (ntiles + D2_NW * nt - 1) / (D2_NW * nt);
    // K-slices: the slice count that brings the launch closest to one workgroup per CU (256), measured over the sweep of
    // profiles/r03_microbench_decode_gemm.txt: gate/up 96 column blocks -> 2 slices (192 workgroups; 4 slices = 384 is 20 % slower),
    // down 32 blocks -> 8 slices, Whisper fc1 80 blocks -> 5 slices (400), the 811-block logits -> 1.  Slices of one or two chunks
    // additionally run the straight-line schedule (every load of the slice in flight at once).
    int sk = 1;
    if (t.splitk > 0) sk = t.splitk;
    else {
        const int items = c.ncb * c.rslabs;
        double best_d = 1e30;
        for (int f = 1; f <= 16 && f <= chunks; ++f) {
            if (chunks % f != 0) continue;
            const double r = (double)items * f / 256.0;
            const double dd = r >= 1.0 ? r : 1.0 / r;
            if (dd < best_d - 1e-9 || (fabs(dd - best_d) <= 1e-9 && f > sk)) {
                best_d = dd;
                sk = f;
            }
        }
    }
    if (sk < 1 || chunks % sk != 0) sk = 1;
    c.splitk = sk;
    c.nch = (steps / sk) / c.KC;
    return c;
}

int dec2_parts(const DecGemm& a) {
    const Dec2Choice c = dec2_choose(a);
    return c.ncb;
}

// 1: launched, 0: not eligible (the caller runs the 16-column kernel), < 0: error
int gemm_dec2(hipStream_t st, const DecGemm& a) {
    Dec2Choice c = dec2_choose(a);
    if (c.ncb == 0) return 0;
    const int NS = c.NT * (a.glu ? 2 : 1);
    Dec2Plan p{};
    p.ncb = c.ncb;
    p.splitk = c.splitk;
    p.nch = c.nch;
    p.dbg = tune().dbg;
    if (c.splitk > 1) {
        void* ws = a.ws;
        size_t wsb = a.ws_bytes;
        if (!ws) {
            ws = default_ws().p;
            wsb = default_ws().bytes;
        }
        const size_t need = D2_CNT_BYTES + (size_t)c.rslabs * c.ncb * c.splitk * D2_NW * c.MT * NS * 1024;
        if (!ws || wsb < need || (size_t)c.rslabs * c.ncb > D2_CNT_BYTES / 4) {      // no room for the slabs: one slice per tile
            p.splitk = 1;
            p.nch = c.nch * c.splitk;
        } else {
            p.cnt = reinterpret_cast<unsigned*>(ws);
            p.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + D2_CNT_BYTES);
        }
    }
    if (c.MT == 4) {
        if (a.glu) launch2<4, 1, true, 8>(st, a, p, c.rslabs);
        else if (c.NT == 2) launch2<4, 2, false, 8>(st, a, p, c.rslabs);
        else launch2<4, 1, false, 8>(st, a, p, c.rslabs);
    } else {
        if (a.glu) launch2<8, 1, true, 4>(st, a, p, c.rslabs);
        else if (c.NT == 2) launch2<8, 2, false, 4>(st, a, p, c.rslabs);
        else launch2<8, 1, false, 4>(st, a, p, c.rslabs);
    }
    S2S_LAUNCH_CHECK();
    return 1;
}

}  // namespace s2s

/* Tuning hook of the decode projections (microbenchmarks / parity sweeps): enable = 0 forces the 16-column kernel,
 * splitk / nt = 0 let the library choose.  Process-wide; not for use while engines are serving. */
/* Development hook: 8 phase timestamps (100 MHz counter) per workgroup of every following column-block launch are written to
 * dbg (device memory, >= 8 * workgroups u64), null switches it off. */
extern "C" int32_t s2s_op_gemm_dec_debug(void* dbg) {
    s2s::dec2_set_debug(reinterpret_cast<unsigned long long*>(dbg));
    s2s::decw_set_debug(reinterpret_cast<unsigned long long*>(dbg));
    return S2S_OK;
}

extern "C" int32_t s2s_op_gemm_dec_tune(int32_t enable, int32_t splitk, int32_t nt) {
    s2s::dec2_set_tune(enable, splitk, nt);
    return S2S_OK;
}
