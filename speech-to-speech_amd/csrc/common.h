// Shared host/device helpers for libs2s_mi355x (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/s2s_mi355x.h"
#include "../../include/s2s_mi355x_dev.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

namespace s2s {

// ---- error plumbing: no exceptions cross the C ABI -------------------------
void set_error(const std::string& msg);
int fail(int code, const char* fmt, ...);

#define S2S_HIP_CHECK(expr)                                                         \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess)                                                       \
            return ::s2s::fail(S2S_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__,   \
                               #expr, hipGetErrorString(_e));                       \
    } while (0)

#define S2S_LAUNCH_CHECK()                                                          \
    do {                                                                            \
        hipError_t _e = hipGetLastError();                                          \
        if (_e != hipSuccess)                                                       \
            return ::s2s::fail(S2S_ERR_HIP, "%s:%d launch -> %s", __FILE__,         \
                               __LINE__, hipGetErrorString(_e));                    \
    } while (0)

#define S2S_REQUIRE(cond, ...)                                                      \
    do {                                                                            \
        if (!(cond)) return ::s2s::fail(S2S_ERR_ARG, __VA_ARGS__);                  \
    } while (0)

// Optional per-kernel timing (s2s_prof_enable / _collect).  While a category is enabled, the launch made inside a ProfScope goes
// through hipExtLaunchKernelGGL with a start / stop event pair: those events carry the KERNEL's own begin / end timestamps (what
// rocprofv3 --kernel-trace reports as its duration), not stream markers - so no event-overhead correction is involved and the
// family averages can be recomputed from a profiler CSV.
class ProfScope {
  public:
    ProfScope(hipStream_t st, int cat, double flops, double bytes);
    ~ProfScope();

  private:
    [[maybe_unused]] hipStream_t st_;
    int cat_;
    bool on_;
    void* start_ = nullptr;
    void* stop_ = nullptr;
};

// the innermost active ProfScope's event pair on this thread (false: timing off for the current launch)
bool prof_current_events(hipEvent_t* start, hipEvent_t* stop);

// kernel launch that is timed when it happens inside an active ProfScope
template <typename... KArgs, typename... Args>
inline void launch_k(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof_current_events(&e0, &e1)) hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, e0, e1, 0, static_cast<KArgs>(args)...);
    else hipLaunchKernelGGL(kernel, grid, block, (unsigned)lds, st, static_cast<KArgs>(args)...);
}

// Stream captures are serialised process-wide: two threads capturing at the same time (e.g. the TTS lanes of a session pool warming up
// together) leave the HIP runtime in hipErrorIllegalState for later launches, thread-local capture mode notwithstanding.
std::mutex& capture_mutex();

// decode loops replay captured graphs unless kernel timing is on (events cannot be recorded inside a replay)
bool prof_active();
bool graphs_enabled();

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- S2S_GUARD: development allocator mode (corner the unexplained "Memory access fault") -------------------
//   S2S_GUARD=1  every workspace sub-buffer is followed by a poisoned gap (0xA5); s2s_guard_check() reports the gaps a kernel wrote
//                into - a store past the end of a buffer that stays inside the arena never faults, it corrupts a neighbour.
//   S2S_GUARD=2  additionally every guarded allocation ends at an UNMAPPED page (HIP virtual memory management): a load or store
//                past the end faults at once, at an address the table printed at allocation time attributes to its owner.
// Off (unset / 0): plain hipMalloc, no gaps - the product path is byte-for-byte what it was.
int guard_mode();
size_t guard_gap();                                                           // bytes of poison behind every sub-buffer (0 when off)
int guard_alloc(char** out, size_t bytes, const char* owner);                 // hipMalloc, or (mode 2) a mapping that ends at an unmapped page
void guard_free(void* p);
int guard_add_gap(char* p, size_t len, const char* owner, int index);         // poison + register [p, p + len)

// carve-up of one workspace arena (whisper.cpp / parakeet.cpp / tts.cpp): offsets of 256-byte aligned sub-buffers
struct ArenaPlan {
    size_t off = 0;
    std::vector<std::pair<size_t, size_t>> gaps;      // guard mode: (offset, length) from the exact end of a sub-buffer to the next one
    size_t take(size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        if (const size_t g = guard_gap()) {
            off += g;
            gaps.emplace_back(o + bytes, off - (o + bytes));
        }
        return o;
    }
    // allocate the arena (zero-filled); guard mode poisons and registers the gaps
    int commit(char** arena, const char* owner) {
        int rc = guard_alloc(arena, off, owner);
        if (rc) return rc;
        S2S_HIP_CHECK(hipMemset(*arena, 0, off));
        int idx = 0;
        for (const auto& g : gaps) {
            if ((rc = guard_add_gap(*arena + g.first, g.second, owner, idx++))) return rc;
        }
        return S2S_OK;
    }
};

// ---- device helpers --------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned int)b) << 16);
}
// round-to-nearest-even f32 -> bf16 bits (NaN kept quiet)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    unsigned int u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ float zero() { return 0.f; }
};
template <> struct Elem<__hip_bfloat16> {
    static __device__ __forceinline__ float ld(const __hip_bfloat16* p) {
        return bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(p));
    }
    static __device__ __forceinline__ void st(__hip_bfloat16* p, float v) {
        *reinterpret_cast<unsigned short*>(p) = f32_to_bf16_bits(v);
    }
    static __device__ __forceinline__ __hip_bfloat16 zero() {
        unsigned short z = 0;
        return *reinterpret_cast<__hip_bfloat16*>(&z);
    }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// order-preserving float <-> uint mapping for atomicMax on floats
__device__ __forceinline__ unsigned int f32_ordered(float f) {
    unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_unordered(unsigned int u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// two f32 -> one dword of bf16 (round to nearest even) with the gfx950 pair conversion v_cvt_pk_bf16_f32
typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float hw_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2_hw(float lo, float hi) {
    const hw_f32x2_t f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, hw_bf16x2_t));
}

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// block-wide argmax of one logits row with optional suppress masks; ties -> lowest index (torch.argmax)
// (range form: indices v_lo .. V - 1 of the row; `best_out` receives the winning value)
__device__ __forceinline__ int block_argmax_256_range(const float* __restrict__ row, int v_lo, int V, const unsigned char* __restrict__ mask,
                                                      const unsigned char* __restrict__ mask2, float* best_out) {
    __shared__ float sv[4];
    __shared__ int si[4];
    float best = -INFINITY;
    int bi = 0x7fffffff;
    // 8 independent loads per thread in flight (a vocabulary row is 50 k floats for 256 threads: one load at a time made the Whisper pick 98 us
    // per step, rocprof round 5); compared in index order, so ties still go to the lowest index
    for (int v0 = v_lo + threadIdx.x; v0 < V; v0 += 256 * 8) {
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int v = v0 + k * 256;
            x[k] = v < V ? row[v] : -INFINITY;
        }
        if (mask) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int v = v0 + k * 256;
                if (v < V && mask[v]) x[k] = -INFINITY;
            }
        }
        if (mask2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int v = v0 + k * 256;
                if (v < V && mask2[v]) x[k] = -INFINITY;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int v = v0 + k * 256;
            if (v < V && (x[k] > best || (x[k] == best && v < bi))) {
                best = x[k];
                bi = v;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sv[wave] = best;
        si[wave] = bi;
    }
    __syncthreads();
    best = sv[0];
    bi = si[0];
    for (int w = 1; w < 4; ++w)
        if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
            best = sv[w];
            bi = si[w];
        }
    if (best_out) *best_out = best;
    return bi;
}
__device__ __forceinline__ int block_argmax_256(const float* __restrict__ row, int V, const unsigned char* __restrict__ mask,
                                                const unsigned char* __restrict__ mask2) {
    return block_argmax_256_range(row, 0, V, mask, mask2, nullptr);
}
// The argmax of one row by ARGMAX_SLICES workgroups (blockIdx.y = slice): every workgroup publishes its slice's candidate and takes a ticket;
// the LAST arriver combines the candidates in slice order (lowest index wins ties, as in the one-workgroup form), resets the ticket and gets
// the result - every other workgroup gets -1 and returns.  Candidates and tickets are agent-scope atomics (a row's workgroups run on
// different XCDs; their L2s are not coherent with each other).  A vocabulary row of 50 k floats took one 256-thread workgroup 90 - 98 us.
constexpr int ARGMAX_SLICES = 8;
__device__ __forceinline__ int sliced_argmax_256(const float* __restrict__ row, int V, const unsigned char* __restrict__ mask,
                                                 const unsigned char* __restrict__ mask2, int b, float* cand_v, int* cand_i, unsigned* ticket) {
    __shared__ unsigned last_s;
    const int s = blockIdx.y;
    const int chunk = ((V + ARGMAX_SLICES - 1) / ARGMAX_SLICES + 255) / 256 * 256;
    float best;
    const int bi = block_argmax_256_range(row, s * chunk, min(V, (s + 1) * chunk), mask, mask2, &best);
    if (threadIdx.x == 0) {
        __hip_atomic_store(cand_v + b * ARGMAX_SLICES + s, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(cand_i + b * ARGMAX_SLICES + s, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        last_s = __hip_atomic_fetch_add(ticket + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (last_s != (unsigned)(ARGMAX_SLICES - 1)) return -1;
    int win = 0x7fffffff;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        float wv = -INFINITY;
        for (int k = 0; k < ARGMAX_SLICES; ++k) {
            const float v = __hip_atomic_load(cand_v + b * ARGMAX_SLICES + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int i = __hip_atomic_load(cand_i + b * ARGMAX_SLICES + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v > wv || (v == wv && i < win)) {
                wv = v;
                win = i;
            }
        }
        __hip_atomic_store(ticket + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (for the next step's launch: stream order)
    }
    return win;      // (valid in thread 0 - the thread that advances the row)
}


}  // namespace s2s
