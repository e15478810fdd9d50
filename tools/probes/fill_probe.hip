// L2 -> CU fill-rate probe (gfx950): what one workgroup per CU pulls in when the WHOLE chip stages L2-resident data at once --
// the number the conv kernels' staging analysis rests on (the guide's ~11 B/cycle/CU is an HBM-bound burst, not an L2-hit rate).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/_fill_probe tools/probes/fill_probe.hip && gpurun -- tools/probes/_fill_probe
// modes: 0 LDS-DMA b128 (buffer_load ... lds), 1 buffer_load b128 into registers (+ one ds_write_b128 each), 2 LDS-DMA b32
// sources: P private 64 KB region per workgroup (cycled: misses the 32 KB L1, sits in L2), S one 512 KB region shared by all workgroups
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int U>
__global__ __launch_bounds__(512) void fill_k(const float* src, unsigned region_bytes, unsigned wg_stride_bytes, int rounds,
                                              unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)src + (size_t)blockIdx.x * wg_stride_bytes), (short)0, (int)region_bytes, 0x00020000);
    constexpr unsigned LB = MODE == 2 ? 4u : 16u;             // bytes per lane and instruction
    constexpr unsigned ROUND_BYTES = 8u * U * 64u * LB;       // one round of the workgroup
    u32x4 acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned base = 0;
    for (int r = 0; r < rounds; ++r) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned off = base + ((unsigned)(u * 8 + wv) * 64u + lane) * LB;
            float* dst = lds + ((r & 1) * (ROUND_BYTES / 4)) + (u * 8 + wv) * (64 * LB / 4);
            if (MODE == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
            else if (MODE == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 4, off, 0, 0, 0);
            else v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        }
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) *(u32x4*)(lds + ((r & 1) * (ROUND_BYTES / 4)) + ((u * 8 + wv) * 64 + lane) * 4) = v[u];
        }
        __builtin_amdgcn_s_waitcnt(0x0f70 & 0x3f70);   // vmcnt(0)
        __syncthreads();
        acc.x += __float_as_uint(lds[(r & 1) * (ROUND_BYTES / 4) + tid]);
        base += ROUND_BYTES;
        if (base + ROUND_BYTES > region_bytes) base = 0;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc.x == 0x12345678u) sink[tid] = 1.f;
}

template <int MODE, int U>
void run(const char* what, const float* src, unsigned region, unsigned stride, int grid, unsigned long long* dcyc, float* sink) {
    constexpr unsigned LB = MODE == 2 ? 4u : 16u;
    const unsigned round_bytes = 8u * U * 64u * LB;
    const int rounds = (int)((8u << 20) / round_bytes);          // 8 MB per workgroup
    const size_t shmem = 2 * (size_t)round_bytes;
    hipFuncSetAttribute((const void*)fill_k<MODE, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((fill_k<MODE, U>), dim3(grid), dim3(512), shmem, 0, src, region, stride, rounds, dcyc, sink);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> c(grid);
    hipMemcpy(c.data(), dcyc, grid * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double bytes = (double)rounds * round_bytes;
    printf("%-8s mode %d U %2d round %3u KB grid %3d | median %.1f B/cycle/CU (slowest %.1f) | %.2f TB/s aggregate | %.1f us\n", what, MODE, U,
           round_bytes >> 10, grid, bytes / c[grid / 2], bytes / c[grid - 1], bytes * grid / (ms * 1e-3) / 1e12, ms * 1e3);
}

int main() {
    const int grid = 256;
    float* src; hipMalloc(&src, (size_t)grid * (64 << 10));
    hipMemset(src, 0, (size_t)grid * (64 << 10));
    unsigned long long* dcyc; hipMalloc(&dcyc, grid * 8);
    float* sink; hipMalloc(&sink, 4096);
    const unsigned P = 64u << 10, S = 512u << 10;
    for (int g : {256, 32}) {
        printf("-- grid %d\n", g);
        run<0, 2>("private", src, P, P, g, dcyc, sink);
        run<0, 4>("private", src, P, P, g, dcyc, sink);
        run<0, 8>("private", src, P, P, g, dcyc, sink);
        run<1, 4>("private", src, P, P, g, dcyc, sink);
        run<1, 8>("private", src, P, P, g, dcyc, sink);
        run<2, 8>("private", src, P, P, g, dcyc, sink);
        run<2, 16>("private", src, P, P, g, dcyc, sink);
        run<0, 2>("shared", src, S, 0, g, dcyc, sink);
        run<0, 4>("shared", src, S, 0, g, dcyc, sink);
        run<0, 8>("shared", src, S, 0, g, dcyc, sink);
        run<1, 8>("shared", src, S, 0, g, dcyc, sink);
    }
    return 0;
}
