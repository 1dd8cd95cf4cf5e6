// Decode-step projections, 129 .. 1024 rows (bf16, one row per live session of a WIDE frame loop): the row-block x column-block kernel.
//
// Why a third kernel (round 4).  A decode launch costs 6 us at 1 row and 10 us at 64 rows (profiles/r03_microbench_decode_gemm.txt): up
// to 64 rows the projections are a chain of latencies, so the only way to make a session's frame cheaper is to put MORE sessions into
// one launch.  Above 128 rows gemm_dec / gemm_dec2 ran independent 128-row slabs over grid.y - every slab re-streamed the weights and
// (16-column kernel) every workgroup re-read its whole activation slab.  At 256 - 512 rows a projection is an ordinary GEMM
// (qkv 4096 x 2048 at 512 rows: 8.6 GFLOP, 3.4 us of MFMA at peak against 2.7 us of weight streaming), so it gets a GEMM's blocking:
//
//   * work item = (row block rb of BM = 32 TMW rows, column block cb of BN = 32 TNW weight rows [SwiGLU: 16 TNW gate + 16 TNW up],
//     K-slice ks); 4 waves as 2 x 2, wave tile 16 TMW x 16 TNW (mfma_f32_16x16x32_bf16, the WEIGHTS as the row operand: a lane then
//     holds 4 consecutive output columns of one activation row).  TMW, TNW in {4, 2}: 128 x 128 .. 64 x 64 tiles.
//   * BOTH operands are fragment-major (one contiguous 1 KiB block per 16-row tile and 32-wide k-step, ops.h), so one
//     global_load_lds_dwordx4 wave instruction moves one block into LDS in exactly the order the MFMA consumes it: the LDS image
//     needs no swizzle (ds_read_b128 at lane * 16 is linear and conflict free), no VGPR staging and no address arithmetic.
//     A k-step = 2 TMW activation blocks + 2 TNW weight blocks per stage; four stages form a ring, two to three of them in flight
//     behind the one being multiplied (counted s_waitcnt vmcnt + raw s_barrier, cdna_hip_programming.md:398-422,440); the fragments
//     of k-step t + 1 are read while the MFMAs of k-step t run; loads past the slice re-read its last k-step so that the wait is the
//     same instruction for the whole loop.
//   * every weight byte leaves HBM ONCE: the row blocks that share a (column block, K-slice) are dispatched next to each other on the
//     SAME XCD (block id = group * 8 * nrb + rb * 8 + xcd), so the second .. nrb-th readers hit that XCD's L2 (PMC: 17 MB of
//     fabric reads for the 16.8 MB qkv projection at 512 rows, 76 % L2 hits, profiles/r04_decw_pmc.txt).
//   * WHAT BOUNDS IT (phase timestamps, profiles/r04_decode_gemm_phases.txt): a CU's global -> LDS fill path delivers about 52 GB/s
//     whatever the ring depth, L2 policy or placement, so the main loop costs (stage bytes) / 52 GB/s per k-step and workgroup -
//     316 ns for a 16 KiB stage against 107 ns of MFMA work.  The plan therefore minimises the bytes the busiest CU has to pull:
//     the tile shape and the K-slice count are chosen per shape so that the launch covers all 256 CUs (dec_choose below).
//   * short grids are filled by K-slices, combined in slice order by the last arriver (gemm_dec2.hip's ticket scheme: sc1 slabs,
//     drained, one relaxed ticket - deterministic, no atomics on data, placement independent).
// Epilogue (on the accumulator layout, no LDS round trip): MODE 0 bf16 row- / fragment-major + SwiGLU, MODE 1 f32, MODE 2 X += out
// with the bf16 copy and ONE partial sum of squares per column block; the LayerNorm prologue of the 16-column kernel (pair partials,
// column sums).  bf16 outputs leave as 16-byte stores: the two lanes that hold the halves of an 8-column chunk of two adjacent tiles
// swap one half each (lane ^ 16), so a wave instruction writes one whole fragment block / 64 contiguous bytes per row instead of
// 8-byte pieces; the residual rows of MODE 2 are ALL fetched before the first store (in place: a load can not pass an earlier store
// that may alias it - interleaved they were 16 serial load -> store round trips per lane, 9 - 14 us per launch).
#include "ops.h"

#include <cmath>
#include <cstdlib>

namespace s2s {

namespace {

constexpr int DW_NST = 4;

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
    float* slabs;       // [tile][slice][wave][TMW * TNW][256] f32 partial tiles
    unsigned* cnt;      // [tile] arrival tickets (zero between launches)
    unsigned long long* dbg;   // development: 8 phase timestamps (100 MHz) per workgroup, or null
};

template <bool GLU, int TMW, int TNW>
__global__ __launch_bounds__(256, 2) void gemm_decw_kernel(DecGemm a, DecwPlan p) {
    constexpr int NST = DW_NST;
    constexpr int BM = 32 * TMW;                 // rows per block
    constexpr int NA = 2 * TMW, NW = 2 * TNW;    // activation / weight blocks per stage
    constexpr int STAGE = (NA + NW) * 1024;
    constexpr int LPW = (NA + NW) / 4;           // blocks a wave moves per stage
    constexpr int RING = NST * STAGE;
    constexpr int NTO = GLU ? TNW / 2 : TNW;     // output column tiles per wave
    constexpr int TPR = 256 / BM;                // threads per row of the row statistics
    static_assert(!GLU || TNW % 2 == 0, "SwiGLU: gate and up tiles in pairs");
    static_assert((NA + NW) % 4 == 0 && (TPR == 2 || TPR == 4), "stage blocks divide among the four waves");
    // ONE LDS object (a second one makes hipcc drain the load queue before every ds_read, cdna_hip_programming.md:496)
    __shared__ __attribute__((aligned(16))) unsigned char lds[RING + 4 * BM * 4 + 16];
    float* rstd_s = reinterpret_cast<float*>(lds + RING);          // [BM]
    float* mean_s = rstd_s + BM;                                    // [BM]
    float* ssw = mean_s + BM;                                       // [2][BM]
    unsigned* flag_s = reinterpret_cast<unsigned*>(ssw + 2 * BM);

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // development timestamps (scalar registers, written out at the end - gemm_dec2.hip)
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
            for (int i = 0; i < 8; ++i) p.dbg[(long)blockIdx.x * 8 + i] = ts[i];
        }
    };
    stamp(0);

    // work item: the row blocks of one (column block, K-slice) pair sit at stride 8 in the grid - same XCD, dispatched together
    const int pairs = p.ncb * p.splitk;
    const int grp = blockIdx.x / (8 * p.nrb), rem = blockIdx.x % (8 * p.nrb);
    const int rb = rem >> 3, q = grp * 8 + (rem & 7);
    if (q >= pairs) return;
    const int cb = q % p.ncb, ks = q / p.ncb;
    const int steps = a.K >> 5;
    const int ns = p.nsteps, s0 = ks * ns;
    const int ntiles = (a.N + 15) >> 4, mtiles = (a.M + 15) >> 4;
    const int m0 = rb * BM;

    // ---- loader: wave w moves blocks LPW w .. LPW w + LPW - 1 of every stage (blocks 0 .. NA - 1: activation row tiles, then the
    // weight tiles; SwiGLU: per column wave [gate tiles | up tiles]) ----
    const char* src[LPW];
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
        const int b = wave * LPW + j;
        if (b < NA) {
            const long blk = (long)min(rb * NA + b, mtiles - 1) * steps;         // rows past M: the last tile again (never stored)
            src[j] = reinterpret_cast<const char*>(a.A + (blk + s0) * 512) + lane * 16;
        } else {
            const int bb = b - NA;
            int t;
            if constexpr (GLU) {
                const int w_ = bb / TNW, e = bb % TNW;                           // column wave, entry: [0, NTO) gate, [NTO, TNW) up
                t = min(cb * (2 * NTO) + w_ * NTO + (e % NTO), ntiles - 1) + (e >= NTO ? ntiles : 0);
            } else {
                t = min(cb * NW + bb, ntiles - 1);
            }
            src[j] = reinterpret_cast<const char*>(a.W + ((long)t * steps + s0) * 512) + lane * 16;
        }
    }
    unsigned char* const my_dst = lds + wave * (LPW * 1024);
    auto issue = [&](int stage, int s) {
        const long off = (long)min(s, ns - 1) * 1024;
        unsigned char* d = my_dst + stage * STAGE;
#pragma unroll
        for (int j = 0; j < LPW; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + off),
                                             (__attribute__((address_space(3))) void*)(d + j * 1024), 16, 0, 0);
    };

#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(s, s);

    // ---- row statistics of the input rows (RMS: gamma folded into W; LayerNorm: pair partials - even parts sums of squares, odd
    // parts plain sums): TPR threads per row, thread `sub` sums the parts sub, sub + TPR, ...  The loads go out right BEHIND the first
    // ring stages (hipcc waits vmcnt(0) for an ordinary load next to LDS-DMA traffic, cdna_hip_programming.md:499 - that wait is the
    // HBM latency of the first stages, which the main loop needs anyway).
    const bool has_ss = a.ss_in != nullptr;
    const int srow = tid / TPR, sub = tid % TPR;
    float sacc = 0.f;
    {
        const float* base = has_ss ? a.ss_in : reinterpret_cast<const float*>(a.A);
        const long ssrs = has_ss ? (a.ss_rs > 0 ? a.ss_rs : 1) : 0, sst = has_ss ? a.ss_stride : 0;
        const int np = has_ss ? a.n_parts_in : 1;
        const float* rowp = base + (long)min(m0 + srow, a.M - 1) * ssrs;
        for (int p0 = 0; p0 < np; p0 += 8 * TPR) {
            float sv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) sv[j] = rowp[(long)min(p0 + sub + TPR * j, np - 1) * sst];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned keep = p0 + sub + TPR * j < np ? 0xffffffffu : 0u;      // (a mask, not a select: a select lets hipcc sink each load under a branch)
                sacc += __uint_as_float(__float_as_uint(sv[j]) & keep);
            }
        }
    }
    {
        if constexpr (TPR == 4) sacc += __shfl_xor(sacc, 2, 64);      // parts of the same parity
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

    f32x4_t acc[TMW][TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // Main loop, software-pipelined inside the wave: the fragments of k-step t + 1 are read from LDS while the MFMAs of k-step t run.
    // Ring protocol per k-step t: counted wait (this wave's blocks of k-step t + 1 have landed) -> barrier (every wave's have; and
    // every wave has finished the MFMAs of k-step t - 1, i.e. its reads of that stage) -> refill the stage of k-step t - 1 with
    // k-step t + NST - 1 -> read k-step t + 1 -> multiply k-step t.
    const unsigned char* const rdA = lds + (wr * TMW) * 1024 + lane * 16;
    const unsigned char* const rdW = lds + NA * 1024 + (wc * TNW) * 1024 + lane * 16;
    bf16x8_t fa[2][TMW], fw[2][TNW];
    auto rd = [&](int buf, int slot) {
        const int so = slot * STAGE;
#pragma unroll
        for (int i = 0; i < TMW; ++i) fa[buf][i] = *reinterpret_cast<const bf16x8_t*>(rdA + so + i * 1024);
#pragma unroll
        for (int j = 0; j < TNW; ++j) fw[buf][j] = *reinterpret_cast<const bf16x8_t*>(rdW + so + j * 1024);
    };
    auto mm = [&](int buf) {
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[buf][j], fa[buf][i], acc[i][j], 0, 0, 0);
    };
    auto step = [&](int t, int cur, int nxt) {
        DW_WAIT_VMCNT((NST - 3) * LPW);
        __builtin_amdgcn_s_barrier();
        issue((t + NST - 1) & (NST - 1), t + NST - 1);
        rd(nxt, (t + 1) & (NST - 1));
        mm(cur);
    };
    stamp(1);
    DW_WAIT_VMCNT((NST - 2) * LPW);          // k-step 0 has landed
    __builtin_amdgcn_s_barrier();
    stamp(2);
    rd(0, 0);
    {
        int t = 0;
        for (; t + 1 < ns; t += 2) {
            step(t, 0, 1);
            step(t + 1, 1, 0);
        }
        if (t < ns) step(t, 0, 1);
    }
    stamp(3);
    DW_WAIT_VMCNT(0);                        // the tail's redundant ring loads: nothing may land in LDS after this point
    __syncthreads();                         // (also orders rstd_s / mean_s for the epilogue)
    stamp(4);

    // ---- K-slices: publish, ticket, last arriver adds the slabs in slice order (gemm_dec2.hip) --------------------------
    if (p.splitk > 1) {
        const int tile = rb * p.ncb + cb;
        constexpr long slab_f = TMW * TNW * 256;                                // floats per wave per slice
        const long tile_f = slab_f * 4 * p.splitk;
        float* base = p.slabs + (long)tile * tile_f;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(tile_f * 4), 0x00020000);
        const unsigned my = (unsigned)(((long)ks * 4 + wave) * slab_f * 4) + lane * 16;
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][j]), rsrc, my + (unsigned)((i * TNW + j) * 1024), 0, 16);   // sc1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave drains its write-through stores
        __syncthreads();
        if (tid == 0) *flag_s = __hip_atomic_fetch_add(p.cnt + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag_s != (unsigned)(p.splitk - 1)) {
            stamps_out();
            return;
        }
        if (tid == 0) __hip_atomic_store(p.cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ticket for the next launch
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int sb = 0; sb < p.splitk; ++sb) {                  // slice order 0 .. splitk - 1, whoever arrived last
            u32x4_t part[TMW][TNW];
            const unsigned so = (unsigned)(((long)sb * 4 + wave) * slab_f * 4) + lane * 16;
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j) part[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, so + (unsigned)((i * TNW + j) * 1024), 0, 16);   // sc1
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < TNW; ++j) acc[i][j] += __builtin_bit_cast(f32x4_t, part[i][j]);
        }
    }
    stamp(5);

    // ---- epilogue on the accumulator layout: lane = (row m = m0 + 16 (TMW wr + i) + l15, columns n .. n + 3, n = 16 tile + 4 lg) ----
    const int tile0 = cb * (2 * NTO) + wc * NTO;         // this wave's first output column tile
    const bool lo_half = (lg & 1) == 0;                  // pair exchange: even lane groups keep the even tile of a pair, odd ones the odd tile
    // the whole block inside M x N with 16-byte aligned rows: the fast paths (everything the engines launch); anything else element-wise
    const bool interior = m0 + BM <= a.M && (tile0 + NTO) * 16 <= a.N;
    const bool x_fast = interior && a.mode == 2 && (a.ldx & 3) == 0;
    // residual values: fetched for TWO row tiles at a time, all of them before the pair's first store (in place: the stores may alias
    // later loads, so every load -> store pair in program order is a serial round trip; 2 round trips instead of 16, 32 registers)
    float4 xo[2][NTO];
    auto xo_load = [&](int i0) {
        if (x_fast && !a.x_overwrite) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int jo = 0; jo < NTO; ++jo)
                    xo[h][jo] = *reinterpret_cast<const float4*>(a.X + (long)(m0 + (wr * TMW + i0 + h) * 16 + l15) * a.ldx + (tile0 + jo) * 16 + lg * 4);
        }
    };
    float ssq[TMW], ssm[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i) ssq[i] = ssm[i] = 0.f;
    // one 8-column chunk of bf16 per lane from the values of a tile pair: keep the own tile's 4 columns, swap the other tile's
    auto pair_pack = [&](const float* v_even, const float* v_odd, u32x4_t& out) {
        const unsigned e0 = pack_bf16x2_hw(v_even[0], v_even[1]), e1 = pack_bf16x2_hw(v_even[2], v_even[3]);
        const unsigned o0 = pack_bf16x2_hw(v_odd[0], v_odd[1]), o1 = pack_bf16x2_hw(v_odd[2], v_odd[3]);
        const unsigned s0_ = lo_half ? o0 : e0, s1_ = lo_half ? o1 : e1;          // what the partner keeps
        const unsigned r0 = (unsigned)__shfl_xor((int)s0_, 16, 64), r1 = (unsigned)__shfl_xor((int)s1_, 16, 64);
        if (lo_half) { out[0] = e0; out[1] = e1; out[2] = r0; out[3] = r1; }       // columns 8 (lg >> 1) .. + 7 of the even tile
        else { out[0] = r0; out[1] = r1; out[2] = o0; out[3] = o1; }               // ... of the odd tile
    };
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        if ((i & 1) == 0) xo_load(i);
        const int ml = (wr * TMW + i) * 16 + l15;      // row inside the block
        const int m = m0 + ml;
        const bool mok = m < a.M;
        const float rsd = rstd_s[ml], mean = mean_s[ml];
        float v[NTO][4];
#pragma unroll
        for (int jo = 0; jo < NTO; ++jo) {
            const int tile = tile0 + jo;
            const int n = tile * 16 + lg * 4;
            const bool some = mok && tile < ntiles && n < a.N;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[jo][r] = acc[i][jo][r];
            if (a.ln && some) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[jo][r] -= mean * a.csum[min(n + r, a.N - 1)];      // (x - mean) W'^T = x W'^T - mean * rowsum(W')
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[jo][r] *= rsd;
            if constexpr (GLU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float up = acc[i][NTO + jo][r] * rsd;
                    v[jo][r] = v[jo][r] / (1.0f + __expf(-v[jo][r])) * up;
                }
            } else {
                if (a.bias && some) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[jo][r] += a.bias[min(n + r, a.N - 1)];
                }
                if (a.act == S2S_ACT_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[jo][r] = gelu_erf(v[jo][r]);
                } else if (a.act == S2S_ACT_SILU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[jo][r] = v[jo][r] / (1.0f + __expf(-v[jo][r]));
                } else if (a.act == S2S_ACT_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[jo][r] = fmaxf(v[jo][r], 0.f);
                }
            }
        }
        if (a.mode == 0) {
            __hip_bfloat16* C = reinterpret_cast<__hip_bfloat16*>(a.C);
            if (interior && NTO % 2 == 0 && (a.c_frag || (a.ldc & 7) == 0)) {
#pragma unroll
                for (int jp = 0; jp + 1 < NTO; jp += 2) {
                    u32x4_t pk;
                    pair_pack(v[jp], v[jp + 1], pk);
                    const int n8 = (tile0 + jp + (lo_half ? 0 : 1)) * 16 + (lg >> 1) * 8;
                    *reinterpret_cast<u32x4_t*>(C + (a.c_frag ? dw_frag_idx(m, n8, (int)(a.ldc >> 5)) : (long)m * a.ldc + n8)) = pk;
                }
            } else {
#pragma unroll
                for (int jo = 0; jo < NTO; ++jo) {
                    const int tile = tile0 + jo, n = tile * 16 + lg * 4;
                    if (!(mok && tile < ntiles)) continue;
                    if (n + 3 < a.N) {
                        u32x2_t pk;
                        pk[0] = pack_bf16x2_hw(v[jo][0], v[jo][1]);
                        pk[1] = pack_bf16x2_hw(v[jo][2], v[jo][3]);
                        *reinterpret_cast<u32x2_t*>(C + (a.c_frag ? dw_frag_idx(m, n, (int)(a.ldc >> 5)) : (long)m * a.ldc + n)) = pk;
                    } else {
                        for (int r = 0; r < 4; ++r)
                            if (n + r < a.N) Elem<__hip_bfloat16>::st(C + (a.c_frag ? dw_frag_idx(m, n + r, (int)(a.ldc >> 5)) : (long)m * a.ldc + n + r), v[jo][r]);
                    }
                }
            }
        } else if (a.mode == 1) {
#pragma unroll
            for (int jo = 0; jo < NTO; ++jo) {
                const int tile = tile0 + jo, n = tile * 16 + lg * 4;
                if (!(mok && tile < ntiles && n < a.N)) continue;
                float* C = reinterpret_cast<float*>(a.C) + (long)m * a.ldc + n;
                const bool full = n + 3 < a.N;
                if (full && ((a.ldc & 3) == 0)) *reinterpret_cast<float4*>(C) = make_float4(v[jo][0], v[jo][1], v[jo][2], v[jo][3]);
                else if (full && ((a.ldc & 1) == 0)) {
                    *reinterpret_cast<float2*>(C) = make_float2(v[jo][0], v[jo][1]);
                    *reinterpret_cast<float2*>(C + 2) = make_float2(v[jo][2], v[jo][3]);
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < a.N) C[r] = v[jo][r];
                }
            }
        } else {
            float xn[NTO][4];
#pragma unroll
            for (int jo = 0; jo < NTO; ++jo)
#pragma unroll
                for (int r = 0; r < 4; ++r) xn[jo][r] = 0.f;
            if (x_fast) {
#pragma unroll
                for (int jo = 0; jo < NTO; ++jo) {
                    if (a.x_overwrite) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) xn[jo][r] = v[jo][r];
                    } else {
                        xn[jo][0] = xo[i & 1][jo].x + v[jo][0]; xn[jo][1] = xo[i & 1][jo].y + v[jo][1];
                        xn[jo][2] = xo[i & 1][jo].z + v[jo][2]; xn[jo][3] = xo[i & 1][jo].w + v[jo][3];
                    }
                    *reinterpret_cast<float4*>(a.X + (long)m * a.ldx + (tile0 + jo) * 16 + lg * 4) = make_float4(xn[jo][0], xn[jo][1], xn[jo][2], xn[jo][3]);
                }
                if (a.XB) {
                    if (NTO % 2 == 0 && (a.xb_frag || (a.ldxb & 7) == 0)) {
#pragma unroll
                        for (int jp = 0; jp + 1 < NTO; jp += 2) {
                            u32x4_t pk;
                            pair_pack(xn[jp], xn[jp + 1], pk);
                            const int n8 = (tile0 + jp + (lo_half ? 0 : 1)) * 16 + (lg >> 1) * 8;
                            *reinterpret_cast<u32x4_t*>(a.XB + (a.xb_frag ? dw_frag_idx(m, n8, (int)(a.ldxb >> 5)) : (long)m * a.ldxb + n8)) = pk;
                        }
                    } else {
#pragma unroll
                        for (int jo = 0; jo < NTO; ++jo) {
                            const int n = (tile0 + jo) * 16 + lg * 4;
                            u32x2_t pk;
                            pk[0] = pack_bf16x2_hw(xn[jo][0], xn[jo][1]);
                            pk[1] = pack_bf16x2_hw(xn[jo][2], xn[jo][3]);
                            *reinterpret_cast<u32x2_t*>(a.XB + (a.xb_frag ? dw_frag_idx(m, n, (int)(a.ldxb >> 5)) : (long)m * a.ldxb + n)) = pk;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int jo = 0; jo < NTO; ++jo) {
                    const int tile = tile0 + jo, n = tile * 16 + lg * 4;
                    if (!(mok && tile < ntiles)) continue;
                    for (int r = 0; r < 4; ++r)
                        if (n + r < a.N) {
                            float* xp = a.X + (long)m * a.ldx + n + r;
                            xn[jo][r] = a.x_overwrite ? v[jo][r] : *xp + v[jo][r];
                            *xp = xn[jo][r];
                            if (a.XB) Elem<__hip_bfloat16>::st(a.XB + (a.xb_frag ? dw_frag_idx(m, n + r, (int)(a.ldxb >> 5)) : (long)m * a.ldxb + n + r), xn[jo][r]);
                        }
                }
            }
#pragma unroll
            for (int jo = 0; jo < NTO; ++jo) {
                ssq[i] += (xn[jo][0] * xn[jo][0] + xn[jo][1] * xn[jo][1]) + (xn[jo][2] * xn[jo][2] + xn[jo][3] * xn[jo][3]);
                ssm[i] += (xn[jo][0] + xn[jo][1]) + (xn[jo][2] + xn[jo][3]);
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
            for (int i = 0; i < TMW; ++i) {
                float qv = pass == 0 ? ssq[i] : ssm[i];
                qv += __shfl_xor(qv, 16, 64);
                qv += __shfl_xor(qv, 32, 64);
                if (lg == 0) ssw[wc * BM + (wr * TMW + i) * 16 + l15] = qv;
            }
            __syncthreads();
            if (tid < BM && m0 + tid < a.M) {
                const float qv = ssw[tid] + ssw[BM + tid];
                const long part = a.ss_pairs ? 2 * cb + pass : cb;
                a.ss_out[part * a.ss_stride + m0 + tid] = qv;
            }
        }
    }
    if (p.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(6);
        stamps_out();
    }
}

struct WideTune {
    int enable = 1;       // 0: never (row slabs of the 16-column / column-block kernels); 44 / 42 / 24 / 22: that tile shape (10 TMW + TNW)
    int splitk = 0;       // 0: chosen from the shape
    int min_m = 65;       // smallest M this kernel takes (the 128-row bucket of the frame loops: 76 vs 94 us per talker layer, r04 microbench)
    unsigned long long* dbg = nullptr;
};
WideTune& wtune() {
    static WideTune t = [] {
        WideTune x;
        if (const char* e = getenv("S2S_DECW")) x.enable = atoi(e);
        if (const char* e = getenv("S2S_DECW_SPLITK")) x.splitk = atoi(e);
        if (const char* e = getenv("S2S_DECW_MIN_M")) x.min_m = atoi(e) < 65 ? 65 : atoi(e);     // gemm_dec() only offers this kernel M > 64: one predicate for the launch and for dec_out_parts()
        return x;
    }();
    return t;
}

constexpr size_t DW_CNT_BYTES = 16384;       // 4096 tickets (shared with gemm_dec2.hip: launches of one stream never overlap)

struct WideChoice { int tmw, tnw, ncb, nrb, splitk, nsteps; };

// The plan for a shape, from the sweep of profiles/r04_microbench_decode_gemm.txt (every tile shape x K-slice count at 128 / 256 / 512
// rows).  What the sweep says: a workgroup's ring holds two to three k-steps in flight, so ONE workgroup pulls about 52 GB/s whatever
// its tile; a CU reaches 120 GB/s once four workgroups share it.  Linear projections therefore take the 64 x 64 tile (33 KiB of LDS,
// < 80 VGPRs: four workgroups per CU) with as many K-slices (<= 4) as bring the launch to about two workgroups per CU; SwiGLU
// projections (twice the weight blocks per output tile) keep the 128 x 128 tile while it yields >= 192 workgroups, else 64 x 128.
WideChoice decw_choose(const DecGemm& a) {
    WideChoice c{};
    const WideTune& t = wtune();
    if (!t.enable || a.nf > 0 || a.M < t.min_m || !a.a_frag) return c;
    if (a.K % 32 != 0 || a.K < 256) return c;
    if (a.glu && a.N % 16 != 0) return c;
    if (a.ss_in && a.n_parts_in > 256) return c;
    const int ntiles = (a.N + 15) / 16, steps = a.K / 32;
    auto plan = [&](int tmw, int tnw, int sk) {
        const int out_tiles = a.glu ? tnw : 2 * tnw;             // output column tiles per block
        return WideChoice{tmw, tnw, (ntiles + out_tiles - 1) / out_tiles, (a.M + 32 * tmw - 1) / (32 * tmw), sk, steps / sk};
    };
    int tmw = 2, tnw = 2;
    if (a.glu) {
        tnw = 4;
        tmw = (long)plan(4, 4, 1).ncb * plan(4, 4, 1).nrb >= 192 ? 4 : 2;
    }
    if (t.enable >= 10) {         // a forced tile shape (sweeps), when this kind of projection has it
        const int ftm = t.enable / 10, ftn = t.enable % 10;
        if ((ftm == 4 || ftm == 2) && (ftn == 4 || ftn == 2) && (!a.glu || ftn % 2 == 0)) {
            tmw = ftm;
            tnw = ftn;
        }
    }
    int sk = 1;
    if (t.splitk > 0) sk = t.splitk;
    else if (!a.glu) {
        const long tiles = (long)plan(tmw, tnw, 1).ncb * plan(tmw, tnw, 1).nrb;
        sk = (int)(512 / (tiles > 0 ? tiles : 1));
        if (sk > 4) sk = 4;
        if (sk < 1) sk = 1;
    }
    while (sk > 1 && (steps % sk != 0 || steps / sk < 4)) --sk;
    c = plan(tmw, tnw, sk);
    if ((long)c.ncb * c.nrb > (long)(DW_CNT_BYTES / 4)) c.splitk = 1, c.nsteps = steps;
    return c;
}

}  // namespace

void decw_set_tune(int enable, int splitk, int min_m) {
    WideTune& t = wtune();
    t.enable = enable;
    t.splitk = splitk;
    if (min_m > 0) t.min_m = min_m < 65 ? 65 : min_m;       // (same clamp as the environment switch)
}

void decw_set_debug(unsigned long long* dbg) { wtune().dbg = dbg; }

int decw_parts(const DecGemm& a) {
    return decw_choose(a).ncb;
}

template <bool GLU, int TMW, int TNW>
static void decw_launch(hipStream_t st, const DecGemm& a, const DecwPlan& p) {
    const int pairs = p.ncb * p.splitk;
    const dim3 grid((unsigned)((pairs + 7) / 8 * 8 * p.nrb)), block(256);
    launch_k(gemm_decw_kernel<GLU, TMW, TNW>, grid, block, 0, st, a, p);
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
    p.dbg = wtune().dbg;
    if (c.splitk > 1) {
        const size_t tiles = (size_t)c.ncb * c.nrb;
        const size_t need = DW_CNT_BYTES + tiles * c.splitk * 4 * c.tmw * c.tnw * 1024;
        if (!a.ws || a.ws_bytes < need || tiles > DW_CNT_BYTES / 4) {      // no room for the slabs: one slice per tile
            p.splitk = 1;
            p.nsteps = c.nsteps * c.splitk;
        } else {
            p.cnt = reinterpret_cast<unsigned*>(a.ws);
            p.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(a.ws) + DW_CNT_BYTES);
        }
    }
    const int shape = c.tmw * 10 + c.tnw;
    if (a.glu) {
        if (shape == 44) decw_launch<true, 4, 4>(st, a, p);
        else if (shape == 24) decw_launch<true, 2, 4>(st, a, p);
        else if (shape == 42) decw_launch<true, 4, 2>(st, a, p);
        else decw_launch<true, 2, 2>(st, a, p);
    } else {
        if (shape == 44) decw_launch<false, 4, 4>(st, a, p);
        else if (shape == 42) decw_launch<false, 4, 2>(st, a, p);
        else if (shape == 24) decw_launch<false, 2, 4>(st, a, p);
        else decw_launch<false, 2, 2>(st, a, p);
    }
    S2S_LAUNCH_CHECK();
    return 1;
}

}  // namespace s2s

/* Tuning hook of the wide decode projections (microbenchmarks / parity sweeps): enable = 0 forces the row-slab path, 44 / 42 / 24 / 22
 * force a tile shape (10 x row tiles + column tiles per wave; SwiGLU: 44 / 24), splitk = 0 lets the library choose, min_m > 0 moves the
 * smallest row count the kernel takes.  Process-wide; not for use while engines are serving. */
extern "C" int32_t s2s_op_gemm_decw_tune(int32_t enable, int32_t splitk, int32_t min_m) {
    s2s::decw_set_tune(enable, splitk, min_m);
    return S2S_OK;
}
