// Internal helpers shared by the libggan translation units (gfx950 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include "../../include/ggan.h"

namespace ggan {

// ---- error plumbing: C ABI never throws -------------------------------------------------------
void set_error(const char* fmt, ...);

#define GGAN_CHECK_ARG(cond, msg)                                  \
    do {                                                           \
        if (!(cond)) {                                             \
            ggan::set_error("%s: %s", __func__, msg);             \
            return -1;                                             \
        }                                                          \
    } while (0)

// ---- per-kernel profiling ---------------------------------------------------------------------
struct ProfScope {
    ProfScope(const char* name, hipStream_t s, double flops, double bytes, long grid);
    ~ProfScope();
    hipStream_t s_;
    int idx_;
};
int check_launch(const char* name);
// Timing experiments only (tools/criticality.sh): kernels whose name contains one of the ';'-separated substrings of
// GGAN_SKIP_KERNELS are not launched at all.  Results are garbage; what the step then gains is that kernel's share of the critical path.
bool launch_skipped(const char* name);
// debugging aid: GGAN_TRACE_LAUNCHES=1 prints one stderr line per launch (name, grid, block, dynamic LDS, flops)
void trace_launch(const char* name, dim3 grid, dim3 block, size_t shmem, double flops);

// LAUNCH(name, flops, bytes, kernel, grid, block, shmem, stream, args...)
// (GGAN_SKIP_KERNELS is compiled in only with -DGGAN_DIAG -- tools/variant_lib.sh builds such a library for tools/criticality*.sh; the
//  product library has no switch that silently drops launches)
#ifdef GGAN_DIAG
#define GGAN_LAUNCH_SKIPPED(name) ggan::launch_skipped(name)
#else
#define GGAN_LAUNCH_SKIPPED(name) false
#endif
#define GGAN_LAUNCH(name, flops, bytes, kernel, grid, block, shmem, stream, ...)          \
    do {                                                                                    \
        if (GGAN_LAUNCH_SKIPPED(name)) break;                                               \
        ggan::trace_launch(name, grid, block, shmem, (double)(flops));                      \
        ggan::ProfScope _ps(name, stream, (double)(flops), (double)(bytes), ggan::grid_wgs(grid) * ggan::grid_wgs(block)); \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                \
    } while (0);                                                                            \
    if (ggan::check_launch(name)) return -2

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// true the first time it is called with `seen` on the current device: hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device,
// a process-wide "once" flag would leave a second device of the same process on the 64 KB default
static inline bool first_on_device(std::atomic<unsigned long long>& seen) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    return !(seen.fetch_or(bit) & bit);
}
static inline long grid_wgs(dim3 g) { return (long)g.x * g.y * g.z; }
static inline long grid_wgs(long g) { return g; }

// Workspace convention (include/ggan.h): the first GGAN_WS_RESERVED bytes of every caller workspace are arrival counters
// for in-kernel split-K combines.  They must be zero before the first call and every kernel leaves them zero.
constexpr size_t kWsReserved = GGAN_WS_RESERVED;
static inline void* ws_scratch(void* ws, size_t& bytes) {
    if (!ws || bytes <= kWsReserved) { bytes = 0; return nullptr; }
    bytes -= kWsReserved;
    return (char*)ws + kWsReserved;
}
static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

// ---- device helpers ---------------------------------------------------------------------------
// A kernel's argument block is read through the scalar cache, which starts every dispatch cold: each group of dependent argument
// reads (the compiler issues them where the values are first needed, often inside branches) then pays a full miss, one after
// the other (measured: ~1 us of a conv kernel's prologue).  Touching one dword of every 64-byte line of the block in ONE batch at
// kernel entry turns those into a single miss time; every later read hits.
template <class T>
__device__ __forceinline__ void warm_kernarg(const T& P) {
    const int* pk = reinterpret_cast<const int*>(&P);
    constexpr int n = (int)(sizeof(T) / 4);
    static_assert(n <= 16 * 16, "argument blocks up to 1 KB");
    auto at = [&](int i) { return pk[i < n ? i : n - 1]; };
    const int v0 = at(0), v1 = at(16), v2 = at(32), v3 = at(48), v4 = at(64), v5 = at(80), v6 = at(96), v7 = at(112);
    asm volatile("" ::"s"(v0), "s"(v1), "s"(v2), "s"(v3), "s"(v4), "s"(v5), "s"(v6), "s"(v7));
    if constexpr (n > 128) {
        const int w0 = at(128), w1 = at(144), w2 = at(160), w3 = at(176), w4 = at(192), w5 = at(208), w6 = at(224), w7 = at(240);
        asm volatile("" ::"s"(w0), "s"(w1), "s"(w2), "s"(w3), "s"(w4), "s"(w5), "s"(w6), "s"(w7));
    }
}

__device__ __forceinline__ float act_apply(float v, int act, float alpha) {
    switch (act) {
        case GGAN_ACT_LRELU: return fmaxf(alpha * v, v);
        case GGAN_ACT_RELU: return fmaxf(v, 0.f);
        case GGAN_ACT_TANH: return tanhf(v);
        case GGAN_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        default: return v;
    }
}

// g * act'(.) given the reference value: forward INPUT or OUTPUT for lrelu/relu (only the sign matters, and
// sign(output) == sign(input) for both), forward OUTPUT for tanh/sigmoid.
__device__ __forceinline__ float act_grad(float g, float r, int act, float alpha) {
    switch (act) {
        case GGAN_ACT_LRELU: return r > 0.f ? g : alpha * g;
        case GGAN_ACT_RELU: return r > 0.f ? g : 0.f;
        case GGAN_ACT_TANH: return g * (1.f - r * r);
        case GGAN_ACT_SIGMOID: return g * r * (1.f - r);
        default: return g;
    }
}

// wave64 all-reduce sum via DPP-free shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x multiple of 64 (<=1024); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 17 floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += smem[i];
    return t;
}

// block_sum into a buffer no earlier reduction of the kernel used: no leading barrier (the reads of an earlier reduction cannot be
// overtaken).  block_sum2_fresh: two values in one pass (one barrier instead of four).
__device__ __forceinline__ float block_sum_fresh(float v, float* smem /* >= 16 floats, not yet used */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += smem[i];
    return t;
}
__device__ __forceinline__ void block_sum2_fresh(float& a, float& b, float* smem /* >= 32 floats, not yet used */) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) { smem[wid] = a; smem[16 + wid] = b; }
    __syncthreads();
    float ta = 0.f, tb = 0.f;
    for (int i = 0; i < nw; ++i) { ta += smem[i]; tb += smem[16 + i]; }
    a = ta; b = tb;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

}  // namespace ggan
