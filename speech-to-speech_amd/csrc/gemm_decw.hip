// Decode-step projections, 129 .. 1024 rows (bf16, one row per live session of a WIDE frame loop): the row-block x column-block kernel.
//
// Why a third kernel (round 4).  A decode launch costs 6 us at 1 row and 10 us at 64 rows (profiles/r03_microbench_decode_gemm.txt): up
// to 64 rows the projections are a chain of latencies, so the only way to make a session's frame cheaper is to put MORE sessions into
// one launch.  Above 128 rows gemm_dec / gemm_dec2 ran independent 128-row slabs over grid.y - every slab re-streamed the weights and
// (16-column kernel) every workgroup re-read its whole activation slab.  At 256 - 512 rows a projection is an ordinary GEMM
// (qkv 4096 x 2048 at 512 rows: 8.6 GFLOP, 3.4 us of MFMA at peak against 2.7 us of weight streaming), so it gets a GEMM's blocking:
//
//   * work item = (row block rb of 128 rows, column block cb of 128 weight rows [SwiGLU: 64 gate + 64 up], K-slice ks);
//     4 waves as 2 x 2, wave tile 64 x 64 = 4 x 4 MFMA tiles (mfma_f32_16x16x32_bf16, the WEIGHTS as the row operand: a lane then
//     holds 4 consecutive output columns of one activation row - the epilogue of gemm_dec2.hip carries over unchanged).
//   * BOTH operands are fragment-major (one contiguous 1 KiB block per 16-row tile and 32-wide k-step, ops.h), so one
//     global_load_lds_dwordx4 wave instruction moves one block into LDS in exactly the order the MFMA consumes it: the LDS image
//     needs no swizzle (ds_read_b128 at lane * 16 is linear and conflict free), no VGPR staging and no address arithmetic.
//     A k-step = 8 activation blocks + 8 weight blocks = 16 KiB per stage; four stages form a ring, three of them in flight behind
//     the one being multiplied (counted s_waitcnt vmcnt(8) + raw s_barrier, cdna_hip_programming.md:398-422,440); loads past the
//     slice re-read its last k-step so that the wait is the same instruction for the whole loop.  64 KiB of LDS and < 128 VGPRs:
//     two workgroups per CU cover each other's barriers.
//   * every weight byte leaves HBM ONCE: the row blocks that share a (column block, K-slice) are dispatched next to each other on the
//     SAME XCD (block id = group * 8 * nrb + rb * 8 + xcd), so the second .. nrb-th readers hit that XCD's L2.
//   * short grids are filled by K-slices, combined in slice order by the last arriver (gemm_dec2.hip's ticket scheme: sc1 slabs,
//     drained, one relaxed ticket - deterministic, no atomics on data, placement independent).
// Epilogues = gemm_dec2.hip's (MODE 0 bf16 row- / fragment-major + SwiGLU, MODE 1 f32, MODE 2 X += out with the bf16 copy and ONE
// partial sum of squares per column block), plus the LayerNorm prologue of the 16-column kernel (pair partials, column sums).
#include "ops.h"

#include <cmath>
#include <cstdlib>

namespace s2s {

namespace {

constexpr int DW_STAGE = 16 * 1024;       // 8 activation blocks + 8 weight blocks
constexpr int DW_NST = 4;
constexpr int DW_RING = DW_NST * DW_STAGE;

// s_waitcnt vmcnt(n) only (expcnt / lgkmcnt left unconstrained); gfx9 encoding: vmcnt = {simm16[15:14], simm16[3:0]}
#define DW_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14))

__host__ __device__ __forceinline__ long dw_frag_idx(int m, int k, int ksteps) {
    return ((((long)(m >> 4) * ksteps + (k >> 5)) * 4 + ((k >> 3) & 3)) * 16 + (m & 15)) * 8 + (k & 7);
}

struct DecwPlan {
    int ncb;            // column blocks
    int nrb;            // row blocks
    int splitk;         // K-slices per tile
    int nsteps;         // k-steps per slice
    float* slabs;       // [tile][slice][wave][16][256] f32 partial tiles
    unsigned* cnt;      // [tile] arrival tickets (zero between launches)
};

template <bool GLU>
__global__ __launch_bounds__(256, 2) void gemm_decw_kernel(DecGemm a, DecwPlan p) {
    constexpr int NTO = GLU ? 2 : 4;          // output column tiles per wave
    // ONE LDS object (a second one makes hipcc drain the load queue before every ds_read, cdna_hip_programming.md:496)
    __shared__ __attribute__((aligned(16))) unsigned char lds[DW_RING + 4 * 128 * 4 + 16];
    float* rstd_s = reinterpret_cast<float*>(lds + DW_RING);       // [128]
    float* mean_s = rstd_s + 128;                                   // [128]
    float* ssw = mean_s + 128;                                      // [2][128]
    unsigned* flag_s = reinterpret_cast<unsigned*>(ssw + 256);

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // work item: the row blocks of one (column block, K-slice) pair sit at stride 8 in the grid - same XCD, dispatched together
    const int pairs = p.ncb * p.splitk;
    const int grp = blockIdx.x / (8 * p.nrb), rem = blockIdx.x % (8 * p.nrb);
    const int rb = rem >> 3, q = grp * 8 + (rem & 7);
    if (q >= pairs) return;
    const int cb = q % p.ncb, ks = q / p.ncb;
    const int steps = a.K >> 5;
    const int ns = p.nsteps, s0 = ks * ns;
    const int ntiles = (a.N + 15) >> 4, mtiles = (a.M + 15) >> 4;
    const int m0 = rb * 128;

    // ---- loader: wave w moves blocks 4 w .. 4 w + 3 of every stage (blocks 0 .. 7: activation row tiles, 8 .. 15: weight tiles) ----
    const char* src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = wave * 4 + j;
        long blk;
        if (b < 8) {
            blk = (long)min(rb * 8 + b, mtiles - 1) * steps;                 // rows past M: the last tile again (never stored)
            src[j] = reinterpret_cast<const char*>(a.A + (blk + s0) * 512) + lane * 16;
        } else {
            const int bb = b - 8;
            int t;
            if constexpr (GLU) t = min(cb * 4 + (bb >> 2) * 2 + (bb & 1), ntiles - 1) + ((bb & 2) ? ntiles : 0);      // [wc][gate 0, gate 1, up 0, up 1]
            else t = min(cb * 8 + bb, ntiles - 1);
            blk = (long)t * steps;
            src[j] = reinterpret_cast<const char*>(a.W + (blk + s0) * 512) + lane * 16;
        }
    }
    unsigned char* const my_dst = lds + wave * 4096;
    auto issue = [&](int stage, int s) {
        const long off = (long)min(s, ns - 1) * 1024;
        unsigned char* d = my_dst + stage * DW_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + off),
                                             (__attribute__((address_space(3))) void*)(d + j * 1024), 16, 0, 0);
    };

    issue(0, 0);
    issue(1, 1);
    issue(2, 2);

    // ---- row statistics of the input rows (RMS: gamma folded into W; LayerNorm: pair partials - even parts sums of squares, odd
    // parts plain sums): 2 threads per row, thread `sub` sums the parts sub, sub + 2, ...  The loads go out right BEHIND the first ring
    // stages (hipcc waits vmcnt(0) for an ordinary load next to LDS-DMA traffic, cdna_hip_programming.md:499 - that wait is the HBM
    // latency of the first stages, which the main loop needs anyway).
    const bool has_ss = a.ss_in != nullptr;
    const int srow = tid >> 1, sub = tid & 1;
    float sacc = 0.f;
    {
        const float* base = has_ss ? a.ss_in : reinterpret_cast<const float*>(a.A);
        const long ssrs = has_ss ? (a.ss_rs > 0 ? a.ss_rs : 1) : 0, sst = has_ss ? a.ss_stride : 0;
        const int np = has_ss ? a.n_parts_in : 1;
        const float* rowp = base + (long)min(m0 + srow, a.M - 1) * ssrs;
        for (int p0 = 0; p0 < np; p0 += 16) {
            float sv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) sv[j] = rowp[(long)min(p0 + sub + 2 * j, np - 1) * sst];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned keep = p0 + sub + 2 * j < np ? 0xffffffffu : 0u;      // (a mask, not a select: a select lets hipcc sink each load under a branch)
                sacc += __uint_as_float(__float_as_uint(sv[j]) & keep);
            }
        }
    }

    {
        const float other = __shfl_xor(sacc, 1, 64);
        if (sub == 0) {
            float rs = 1.0f, mean = 0.f;
            if (has_ss) {
                if (a.ln) {
                    mean = other * a.inv_dim;
                    rs = rsqrtf(fmaxf(sacc * a.inv_dim - mean * mean, 0.f) + a.eps);
                } else {
                    rs = rsqrtf((sacc + other) * a.inv_dim + a.eps);
                }
            }
            rstd_s[srow] = rs;
            mean_s[srow] = mean;
        }
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const unsigned char* const rdA = lds + (wr * 4) * 1024 + lane * 16;
    const unsigned char* const rdW = lds + 8192 + (wc * 4) * 1024 + lane * 16;
    for (int t = 0; t < ns; ++t) {
        DW_WAIT_VMCNT(8);                    // this wave's blocks of k-step t have landed; k-steps t + 1, t + 2 may still be in flight
        __builtin_amdgcn_s_barrier();        // ... for every wave; and every wave is done reading k-step t - 1's stage
        issue((t + 3) & 3, t + 3);
        const int so = (t & 3) * DW_STAGE;
        bf16x8_t fa[4], fw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(rdA + so + i * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) fw[j] = *reinterpret_cast<const bf16x8_t*>(rdW + so + j * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j], fa[i], acc[i][j], 0, 0, 0);
    }
    DW_WAIT_VMCNT(0);                        // the tail's redundant ring loads: nothing may land in LDS after this point
    __syncthreads();                         // (also orders rstd_s / mean_s for the epilogue)

    // ---- K-slices: publish, ticket, last arriver adds the slabs in slice order (gemm_dec2.hip) --------------------------
    if (p.splitk > 1) {
        const int tile = rb * p.ncb + cb;
        const long slab_f = 16 * 256;                                           // floats per wave per slice
        const long tile_f = slab_f * 4 * p.splitk;
        float* base = p.slabs + (long)tile * tile_f;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(tile_f * 4), 0x00020000);
        const unsigned my = (unsigned)(((long)ks * 4 + wave) * slab_f * 4) + lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][j]), rsrc, my + (unsigned)((i * 4 + j) * 1024), 0, 16);   // sc1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains its write-through stores
        __syncthreads();
        if (tid == 0) *flag_s = __hip_atomic_fetch_add(p.cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag_s != (unsigned)(p.splitk - 1)) return;
        if (tid == 0) __hip_atomic_store(p.cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ticket for the next launch
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int sb = 0; sb < p.splitk; ++sb) {                  // slice order 0 .. splitk - 1, whoever arrived last
            u32x4_t part[4][4];
            const unsigned so = (unsigned)(((long)sb * 4 + wave) * slab_f * 4) + lane * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) part[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, so + (unsigned)((i * 4 + j) * 1024), 0, 16);   // sc1
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += __builtin_bit_cast(f32x4_t, part[i][j]);
        }
    }

    // ---- epilogue on the accumulator layout: lane = (row m = m0 + 16 (4 wr + i) + l15, columns n .. n + 3, n = 16 tile + 4 lg) ----
    float ssq[4], ssm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ssq[i] = ssm[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ml = (wr * 4 + i) * 16 + l15;      // row inside the block
        const int m = m0 + ml;
        const bool mok = m < a.M;
        const float rsd = rstd_s[ml], mean = mean_s[ml];
#pragma unroll
        for (int jo = 0; jo < NTO; ++jo) {
            const int tile = GLU ? cb * 4 + wc * 2 + jo : cb * 8 + wc * 4 + jo;
            const int n = tile * 16 + lg * 4;
            const bool full = mok && tile < ntiles && n + 3 < a.N;
            const bool some = mok && tile < ntiles && n < a.N;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][jo][r];
            if (a.ln && some) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] -= mean * a.csum[min(n + r, a.N - 1)];      // (x - mean) W'^T = x W'^T - mean * rowsum(W')
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= rsd;
            if constexpr (GLU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float up = acc[i][2 + jo][r] * rsd;
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
                    *reinterpret_cast<u32x2_t*>(C + (a.c_frag ? dw_frag_idx(m, n, (int)(a.ldc >> 5)) : (long)m * a.ldc + n)) = pk;
                } else if (some) {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < a.N) Elem<__hip_bfloat16>::st(C + (a.c_frag ? dw_frag_idx(m, n + r, (int)(a.ldc >> 5)) : (long)m * a.ldc + n + r), v[r]);
                }
            } else if (a.mode == 1) {
                float* C = reinterpret_cast<float*>(a.C) + (long)m * a.ldc + n;
                if (full && ((a.ldc & 3) == 0)) *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
                else if (full && ((a.ldc & 1) == 0)) {
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
                        *reinterpret_cast<u32x2_t*>(a.XB + (a.xb_frag ? dw_frag_idx(m, n, (int)(a.ldxb >> 5)) : (long)m * a.ldxb + n)) = pk;
                    }
                } else if (some) {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < a.N) {
                            float* xp = a.X + (long)m * a.ldx + n + r;
                            xn[r] = a.x_overwrite ? v[r] : *xp + v[r];
                            *xp = xn[r];
                            if (a.XB) Elem<__hip_bfloat16>::st(a.XB + (a.xb_frag ? dw_frag_idx(m, n + r, (int)(a.ldxb >> 5)) : (long)m * a.ldxb + n + r), xn[r]);
                        }
                }
                ssq[i] += (xn[0] * xn[0] + xn[1] * xn[1]) + (xn[2] * xn[2] + xn[3] * xn[3]);
                ssm[i] += (xn[0] + xn[1]) + (xn[2] + xn[3]);
            }
        }
    }
    if (a.mode == 2 && a.ss_out) {
        // one partial per column block: the 4 lane groups of a row, then the block's two column waves, in a fixed order
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1 && !a.ss_pairs) break;
            if (pass == 1) __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float qv = pass == 0 ? ssq[i] : ssm[i];
                qv += __shfl_xor(qv, 16, 64);
                qv += __shfl_xor(qv, 32, 64);
                if (lg == 0) ssw[wc * 128 + (wr * 4 + i) * 16 + l15] = qv;
            }
            __syncthreads();
            if (tid < 128 && m0 + tid < a.M) {
                const float qv = ssw[tid] + ssw[128 + tid];
                const long part = a.ss_pairs ? 2 * cb + pass : cb;
                a.ss_out[part * a.ss_stride + m0 + tid] = qv;
            }
        }
    }
}

struct WideTune {
    int enable = 1;       // 0: never (row slabs of the 16-column / column-block kernels)
    int splitk = 0;       // 0: chosen from the shape
    int min_m = 129;      // smallest M this kernel takes
};
WideTune& wtune() {
    static WideTune t = [] {
        WideTune x;
        if (const char* e = getenv("S2S_DECW")) x.enable = atoi(e);
        if (const char* e = getenv("S2S_DECW_SPLITK")) x.splitk = atoi(e);
        if (const char* e = getenv("S2S_DECW_MIN_M")) x.min_m = atoi(e);
        return x;
    }();
    return t;
}

constexpr size_t DW_CNT_BYTES = 16384;       // 4096 tickets (shared with gemm_dec2.hip: launches of one stream never overlap)

struct WideChoice { int ncb, nrb, splitk, nsteps; };

WideChoice decw_choose(const DecGemm& a) {
    WideChoice c{};
    const WideTune& t = wtune();
    if (!t.enable || a.nf > 0 || a.M < t.min_m || !a.a_frag) return c;
    if (a.K % 32 != 0 || a.K < 256) return c;
    if (a.glu && a.N % 16 != 0) return c;
    if (a.ss_in && a.n_parts_in > 256) return c;
    const int ntiles = (a.N + 15) / 16, steps = a.K / 32;
    c.ncb = a.glu ? (ntiles + 3) / 4 : (ntiles + 7) / 8;
    c.nrb = (a.M + 127) / 128;
    // K-slices: a grid far below one workgroup per CU is filled by up to 4 slices of >= 8 k-steps (a slab is 64 KiB per workgroup and
    // slice: the last arriver reads splitk of them, so more slices than that cost more than they fill)
    const int tiles = c.ncb * c.nrb;
    int sk = 1;
    if (t.splitk > 0) sk = t.splitk;
    else if (tiles <= 96) {
        sk = 256 / tiles;
        if (sk > 4) sk = 4;
    }
    while (sk > 1 && (steps % sk != 0 || steps / sk < 8)) --sk;
    if (sk < 1) sk = 1;
    c.splitk = sk;
    c.nsteps = steps / sk;
    return c;
}

}  // namespace

void decw_set_tune(int enable, int splitk, int min_m) {
    WideTune& t = wtune();
    t.enable = enable;
    t.splitk = splitk;
    if (min_m > 0) t.min_m = min_m;
}

int decw_parts(const DecGemm& a) {
    return decw_choose(a).ncb;
}

// 1: launched, 0: not eligible (the caller runs the row-slab path), < 0: error
int gemm_decw(hipStream_t st, const DecGemm& a) {
    WideChoice c = decw_choose(a);
    if (c.ncb == 0) return 0;
    DecwPlan p{};
    p.ncb = c.ncb;
    p.nrb = c.nrb;
    p.splitk = c.splitk;
    p.nsteps = c.nsteps;
    if (c.splitk > 1) {
        const size_t tiles = (size_t)c.ncb * c.nrb;
        const size_t need = DW_CNT_BYTES + tiles * c.splitk * 4 * 16 * 1024;
        if (!a.ws || a.ws_bytes < need || tiles > DW_CNT_BYTES / 4) {      // no room for the slabs: one slice per tile
            p.splitk = 1;
            p.nsteps = c.nsteps * c.splitk;
        } else {
            p.cnt = reinterpret_cast<unsigned*>(a.ws);
            p.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(a.ws) + DW_CNT_BYTES);
        }
    }
    const int pairs = p.ncb * p.splitk;
    const dim3 grid((unsigned)((pairs + 7) / 8 * 8 * p.nrb)), block(256);
    if (a.glu) launch_k(gemm_decw_kernel<true>, grid, block, 0, st, a, p);
    else launch_k(gemm_decw_kernel<false>, grid, block, 0, st, a, p);
    S2S_LAUNCH_CHECK();
    return 1;
}

}  // namespace s2s

/* Tuning hook of the wide decode projections (microbenchmarks / parity sweeps): enable = 0 forces the row-slab path, splitk = 0 lets the
 * library choose, min_m > 0 moves the smallest row count the kernel takes.  Process-wide; not for use while engines are serving. */
extern "C" int32_t s2s_op_gemm_decw_tune(int32_t enable, int32_t splitk, int32_t min_m) {
    s2s::decw_set_tune(enable, splitk, min_m);
    return S2S_OK;
}
