/* libs2s_mi355x - DEVELOPMENT and MEASUREMENT hooks (not part of the drop-in boundary).
 *
 * The public boundary is include/s2s_mi355x.h: what a reference-side binding needs to run the hot path.  This header declares what only
 * this repository's own tooling calls - bench.py's kernel-family timers, the microbenchmark graph harness, the tuning switches of the
 * decode projections and their phase timestamps (tools/, tests/test_gemm_dec_gpu.py).  Same library, same error convention
 * (int32 status + s2s_last_error()); no stability promise: these may change without an S2S_ABI_VERSION bump. */
#ifndef S2S_MI355X_DEV_H
#define S2S_MI355X_DEV_H

#include "s2s_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel timing for bench.py's roofline line: while a category bit is enabled every launch of that
 * kernel family is bracketed by HIP events on its own stream; s2s_prof_collect waits for them and
 * returns (and resets) the summed device time, the algorithmic FLOPs / bytes the launches were
 * asked to do, and their count. */
#define S2S_PROF_GEMM_TILE 0   /* bf16/f32 MFMA tile GEMM (encoder projections, conv stems) */
#define S2S_PROF_GEMM_SKINNY 1 /* weight-streaming decode GEMM (M <= 64) */
#define S2S_PROF_ATTN_MFMA 2   /* flash attention */
#define S2S_PROF_ATTN_ROWS 3   /* decode / parity attention */
#define S2S_PROF_LOGMEL 4
#define S2S_PROF_NCAT 8
int32_t s2s_prof_enable(uint32_t category_mask);
int32_t s2s_prof_collect(int32_t category, double* total_ms, double* flops, double* bytes, int64_t* launches);
/* Device time of an empty start/stop event pair (median, measured on first collect); s2s_prof_collect subtracts it from
 * every launch so that the family averages match profiler kernel durations.  0 before the first collect. */
double s2s_prof_event_overhead_us(void);

/* Tuning hook of the decode projections (microbenchmarks, parity sweeps): enable = 0 keeps every call on the 16-column kernel;
 * splitk / nt > 0 force the K-slices per column block / the 16-column tiles per wave of the column-block kernel (0 = chosen
 * from the shape).  Process-wide: not for use while engines are serving. */
int32_t s2s_op_gemm_dec_tune(int32_t enable, int32_t splitk, int32_t nt);
/* The same for the row-block kernel of wide frame loops (65 .. 1024 rows, csrc/gemm_decw.hip): enable = 0 sends such calls down the
 * <= 128-row kernels (row slabs above 128), 44 / 42 / 24 / 22 force a tile shape (10 x row tiles + column tiles per wave), splitk > 0 forces
 * the K-slices per tile, min_m > 0 moves the smallest row count the kernel takes (default 65). */
int32_t s2s_op_gemm_decw_tune(int32_t enable, int32_t splitk, int32_t min_m);
/* development: per-workgroup phase timestamps of the column-block / row-block decode kernels (8 x u64 per workgroup, 100 MHz), null = off */
int32_t s2s_op_gemm_dec_debug(void* dbg_dev);
/* development: per-tile phase timestamps of the 256x256 staggered tile GEMM (csrc/gemm.hip gemm_tile5_kernel): 6 x u64 per tile of every
 * following launch - {start, prologue landed, main loop done, epilogue issued} on the 100 MHz counter, HW_ID, XCC_ID - null = off
 * (tools/tile5_phases.py) */
int32_t s2s_op_gemm_tile_debug(void* dbg_dev);

/* The decode-step attention of the Qwen3 stacks and the Whisper decoder's self-attention as an operator (csrc/attention.hip attn_step_kernel):
 * per-head RMSNorm of q / k (qw / kw, null = none), rotary embedding at pos[b] (cos_t / sin_t [max_pos][D/2], null = none), append of the new
 * K / V row at cache row pos[b], attention over cache[jlo, pos[b]) + the new row.  qkv row b = [H q heads | Hkv k heads | Hkv v heads] x D;
 * K / V element (b, j, hk, d) at base + b*kv_bs + j*kv_rs + hk*D + d; O row b at O + b*o_bs (o_frag: fragment-major [roundup(B, 16)][H*D], gemm_dec layout).
 * s2s_op_attn_step_tune(bits): bit 1 = caches of <= 32 positions on one WAVE per item (attn_step_small_kernel; default for bf16), bit 0 / value 1 = grouped heads (H / Hkv = 2 or 4) share one workgroup per KV head (round 6, default), 0 = one workgroup
 * per query head (rounds 2 - 5), -1 = default / environment S2S_ATTN_STEP_GQA.  tests/test_attention_gpu.py checks the two bit for bit. */
int32_t s2s_op_attn_step(void* stream, const void* qkv, int64_t qkv_rs, void* K, void* V, int64_t kv_bs, int64_t kv_rs, void* O, int64_t o_bs,
                         const float* qw, const float* kw, float eps, const float* cos_t, const float* sin_t, const int32_t* pos_dev, int32_t B,
                         int32_t H, int32_t Hkv, int32_t D, int32_t kv_cap, int32_t window, float scale, int32_t o_frag, int32_t dtype);
int32_t s2s_op_attn_step_tune(int32_t share);
/* The codec's k = 7 dilated convolutions (s2s_op_conv1d_causal_tm shapes with 256-row-aligned items): 1 = tap-reuse kernel (round 6: the tall
 * input tile is staged once per channel chunk and the seven taps read it at row offsets), 0 = the tap-by-tap GEMM of rounds 1 - 5, -1 = default /
 * environment S2S_CONV7.  tests/test_attention_gpu.py compares the two. */
int32_t s2s_op_conv7_tune(int32_t on);

/* Microbenchmark aid: capture whatever is enqueued on *out_stream between begin and end into a hipGraph, replay it
 * `warm` + `reps` times and return the average device time of one replay (kernel cost INSIDE a graph, no host launch). */
int32_t s2s_graph_bench_begin(void** out_stream);
int32_t s2s_graph_bench_end(int32_t warm, int32_t reps, double* us_per_replay);

/* S2S_GUARD (environment, read once): development allocator mode for cornering stray device accesses.
 *   1: every workspace sub-buffer of the engines is followed by a 64 KiB gap filled with 0xA5; s2s_guard_check() copies the gaps back,
 *      writes one line per gap a kernel stored into ("<owner> buffer <i>: n byte(s) written behind its end, offsets +a .. +b") into
 *      `report`, re-poisons it and returns the number of corrupted gaps (0 = clean, < 0 = HIP error).
 *   2: additionally every guarded allocation is a HIP virtual-memory mapping that ENDS at an unmapped page: a load or store past the
 *      end of an arena faults at once, and the table printed to stderr at allocation time ("[s2s guard] owner lo .. hi") attributes
 *      the faulting address.
 * s2s_guard_check synchronises the device: call it between launches (tests/conftest.py does, after every GPU test, when S2S_GUARD is set). */
int32_t s2s_guard_mode(void);
int32_t s2s_guard_check(char* report, int32_t cap);

#ifdef __cplusplus
}
#endif

#endif /* S2S_MI355X_DEV_H */
