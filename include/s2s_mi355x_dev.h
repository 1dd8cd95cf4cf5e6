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
