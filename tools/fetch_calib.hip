// FETCH_SIZE calibration (gfx950, rocprofv3 --pmc FETCH_SIZE): kernels that read a KNOWN number of bytes from HBM through each of
// the access paths the conv / GEMM kernels stage their operands with, so that tools/pmc_summary.py can turn the counter into bytes
// per path instead of guessing (MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of a wide coalesced streaming read; "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
//   dword_k       raw buffer load, 4 B per lane, coalesced        (register-staged slab of the older forward kinds, descriptors)
//   b128_k        raw buffer load, 16 B per lane                  (slab units of the <.., true> kinds, wgrad x / gy, GEMM operands)
//   dma_dword_k   buffer_load ... lds, 4 B per lane               (LDS-DMA slab gathers of corr_kernel<0, .., false>)
//   dma_b128_k    buffer_load ... lds, 16 B per lane              (LDS-DMA filter slices, slab units of corr_kernel<0, .., true>)
// Every kernel reads the same `bytes` once (a buffer far larger than L2 + Infinity Cache is walked, so what is counted is fabric
// traffic, not cache hits); the sum of what was read goes to `out` so that nothing is optimised away.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_fetch_calib tools/fetch_calib.hip
//   gpurun -- 'bash tools/fetch_calib.sh'        -> gpurun_out/fetch_calib.json
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

__global__ __launch_bounds__(256) void dword_k(const float* src, size_t n, float* out) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, 0x7FFFFFF0, 0x00020000);
    float acc = 0.f;
    // each workgroup owns a contiguous 1 MiB stripe per trip (<= 2 GiB buffer: 32-bit offsets)
    for (size_t base = (size_t)blockIdx.x * 262144; base < n; base += (size_t)gridDim.x * 262144)
        for (int i = threadIdx.x; i < 262144; i += 256 * 4) {
            float a = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)((base + i) * 4), 0, 0));
            float b = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)((base + i + 256) * 4), 0, 0));
            float c = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)((base + i + 512) * 4), 0, 0));
            float d = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)((base + i + 768) * 4), 0, 0));
            acc += (a + b) + (c + d);
        }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void b128_k(const float* src, size_t n, float* out) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, 0x7FFFFFF0, 0x00020000);
    float acc = 0.f;
    for (size_t base = (size_t)blockIdx.x * 262144; base < n; base += (size_t)gridDim.x * 262144)
        for (int i = threadIdx.x * 4; i < 262144; i += 256 * 4 * 2) {
            u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((base + i) * 4), 0, 0);
            u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((base + i + 1024) * 4), 0, 0);
            acc += __int_as_float(a.x ^ a.y ^ a.z ^ a.w) + __int_as_float(b.x ^ b.y ^ b.z ^ b.w);
        }
    if (acc == 12345.678f) out[0] = acc;
}

template <int BYTES>
__global__ __launch_bounds__(256) void dma_k(const float* src, size_t n, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 4096];
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, 0x7FFFFFF0, 0x00020000);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int FPL = BYTES / 4;                 // floats per lane and instruction
    float acc = 0.f;
    for (size_t base = (size_t)blockIdx.x * 262144; base < n; base += (size_t)gridDim.x * 262144)
        for (int i = 0; i < 262144; i += 256 * FPL * 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {          // 4 wave-instructions in flight, each lands in its own LDS rows
                const unsigned off = (unsigned)((base + i + (j * 256 + wv * 64 + lane) * FPL) * 4);
                if constexpr (BYTES == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(lds + (j * 4 + wv) * 64 * FPL), 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(lds + (j * 4 + wv) * 64 * FPL), 4, off, 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0f70 & 0x3f70);   // vmcnt(0)
            acc += lds[(threadIdx.x * 5) & 4095];
        }
    if (acc == 12345.678f) out[0] = acc;
}

// Partial lines: every second SEG-byte segment of the buffer is read (dword loads, a wave covers 256 useful bytes per instruction), so
// the useful bytes are HALF the buffer while every 128-byte line (SEG <= 64) or every second one (SEG = 128) is touched.  What FETCH_SIZE
// reports for these tells whether a 64-byte row of a 16-wide image (the x slab rows of the 16x16 layers) is counted as 64 or 128 bytes.
template <int SEG>
__global__ __launch_bounds__(256) void seg_k(const float* src, size_t n, float* out) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, 0x7FFFFFF0, 0x00020000);
    constexpr int SF = SEG / 4;                    // floats per segment
    float acc = 0.f;
    for (size_t base = (size_t)blockIdx.x * 262144; base < n; base += (size_t)gridDim.x * 262144)
        for (int i = threadIdx.x; i < 131072; i += 256) {      // useful float i of this stripe -> segment i / SF, every second segment
            const int e = (i / SF) * (2 * SF) + (i % SF);
            acc += __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)((base + e) * 4), 0, 0));
        }
    if (acc == 12345.678f) out[0] = acc;
}

// WRITE_SIZE: the same question for the store paths (round-4 review: the counter read half of a forward kernel's output tensor).
//   st_dword_k    global store, 4 B per lane, coalesced           (pointwise kernels, scalar epilogues)
//   st_b128_k     global store, 16 B per lane                     (the conv / GEMM epilogues: float4 rows through LDS)
//   st_b128_buf_k raw buffer store, 16 B per lane
__global__ __launch_bounds__(256) void st_dword_k(float* dst, size_t n) {
    for (size_t base = (size_t)blockIdx.x * 262144; base < n; base += (size_t)gridDim.x * 262144)
        for (int i = threadIdx.x; i < 262144; i += 256) dst[base + i] = (float)i;
}

__global__ __launch_bounds__(256) void st_b128_k(float* dst, size_t n) {
    for (size_t base = (size_t)blockIdx.x * 262144; base < n; base += (size_t)gridDim.x * 262144)
        for (int i = threadIdx.x * 4; i < 262144; i += 256 * 4)
            *reinterpret_cast<float4*>(dst + base + i) = make_float4((float)i, 1.f, 2.f, 3.f);
}

__global__ __launch_bounds__(256) void st_b128_buf_k(float* dst, size_t n) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)dst, (short)0, 0x7FFFFFF0, 0x00020000);
    for (size_t base = (size_t)blockIdx.x * 262144; base < n; base += (size_t)gridDim.x * 262144)
        for (int i = threadIdx.x * 4; i < 262144; i += 256 * 4) {
            u32x4 v = {(unsigned)i, 1u, 2u, 3u};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)((base + i) * 4), 0, 0);
        }
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)1 << 30;          // 1 GiB: four times L2 + Infinity Cache
    const size_t n = bytes / 4;
    float *d, *o;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
    (void)hipMemset(d, 0, bytes);
    const int which = argc > 1 ? atoi(argv[1]) : -1;
    for (int rep = 0; rep < 3; ++rep) {
        if (which < 0 || which == 0) hipLaunchKernelGGL(dword_k, dim3(1024), dim3(256), 0, 0, d, n, o);
        if (which < 0 || which == 1) hipLaunchKernelGGL(b128_k, dim3(1024), dim3(256), 0, 0, d, n, o);
        if (which < 0 || which == 2) hipLaunchKernelGGL(dma_k<4>, dim3(1024), dim3(256), 0, 0, d, n, o);
        if (which < 0 || which == 3) hipLaunchKernelGGL(dma_k<16>, dim3(1024), dim3(256), 0, 0, d, n, o);
        if (which < 0 || which == 7) hipLaunchKernelGGL(seg_k<32>, dim3(1024), dim3(256), 0, 0, d, n, o);
        if (which < 0 || which == 8) hipLaunchKernelGGL(seg_k<64>, dim3(1024), dim3(256), 0, 0, d, n, o);
        if (which < 0 || which == 9) hipLaunchKernelGGL(seg_k<128>, dim3(1024), dim3(256), 0, 0, d, n, o);
        if (which < 0 || which == 4) hipLaunchKernelGGL(st_dword_k, dim3(1024), dim3(256), 0, 0, d, n);
        if (which < 0 || which == 5) hipLaunchKernelGGL(st_b128_k, dim3(1024), dim3(256), 0, 0, d, n);
        if (which < 0 || which == 6) hipLaunchKernelGGL(st_b128_buf_k, dim3(1024), dim3(256), 0, 0, d, n);
    }
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    printf("bytes_per_launch %zu\n", bytes);
    return 0;
}
