// cost of one filler instruction between two v_mfma_f32_16x16x4_f32 of ONE wave (one wave per SIMD), by filler kind
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#ifndef BIG
#define BIG 0
#endif
#ifndef ITERS
#define ITERS 64
#endif
#ifndef NTHREADS
#define NTHREADS 256
#endif
template <int KIND, int NF>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    unsigned addr = (unsigned)((threadIdx.x & 15) * 66 + ((threadIdx.x >> 4) & 3)) * 4u + (unsigned)(size_t)(__attribute__((address_space(3))) float*)s;
    typedef float f16v __attribute__((ext_vector_type(16)));
    f16v big; for (int i = 0; i < 16; ++i) big[i] = 0;
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f4){0, 0, 0, 0};
    float a = threadIdx.x, b = 1.f; f2 v = {0, 0}; float w = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (BIG) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(big) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (KIND == 1) asm volatile("s_nop 0");
                if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(15)");
                if (KIND == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "v"(a));
                if (KIND == 4) asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(w) : "v"(addr));
                if (KIND == 5) asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:26" : "=v"(v) : "v"(addr));
                if (KIND == 6) asm volatile("s_mov_b32 s20, 0" ::: "s20");
                if (KIND == 7) asm volatile("s_waitcnt lgkmcnt(4)");
                if (KIND == 8) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a) : "s20");
                if (KIND == 9) asm volatile("v_add_u32 %0, 0x6410, %1" : "=v"(w) : "v"(a));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float r = w + v[0] + v[1] + big[0];
    for (int i = 0; i < 16; ++i) r += acc[i][0];
    out[threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND, int NF> void run(const char* name, float* d, long long* c, int wgs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, NF><<<wgs, NTHREADS>>>(d, c);
    hipEventRecord(e0); k<KIND, NF><<<wgs, NTHREADS>>>(d, c); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)wgs * (NTHREADS / 64) * ITERS * 16 * (BIG ? 4096.0 : 2048.0);
    long long hc[256];
    hipMemcpy(hc, c, 8 * wgs, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < wgs; ++i) m += hc[i]; m /= wgs;
    printf("%-28s x%d, %3d WGs x %d threads, %s: %.1f ticks per MFMA of a wave; %.3f ms, %.1f TFLOP/s, %.2f GHz\n", name, NF, wgs, NTHREADS, BIG ? "32x32x2" : "16x16x4", m / (ITERS * 16.), ms, flop / ms * 1e-9, m / ms * 1e-6);
}
int main() {
    float* d; long long* c; (void)hipMalloc(&d, 4096); (void)hipMalloc(&c, 8 * 256);
    for (int wgs = 256; wgs <= 256; wgs *= 256) {
        run<0, 0>("none", d, c, wgs);
        run<1, 1>("s_nop", d, c, wgs); run<1, 2>("s_nop", d, c, wgs);
        run<2, 1>("s_waitcnt (satisfied, 15)", d, c, wgs); run<2, 2>("s_waitcnt (satisfied, 15)", d, c, wgs);
        run<6, 1>("s_mov", d, c, wgs); run<6, 2>("s_mov", d, c, wgs);
        run<3, 1>("v_mov", d, c, wgs); run<3, 2>("v_mov", d, c, wgs);
        run<4, 1>("ds_read_b32", d, c, wgs); run<4, 2>("ds_read_b32", d, c, wgs);
        run<5, 1>("ds_read2_b32", d, c, wgs); run<5, 2>("ds_read2_b32", d, c, wgs);
        run<7, 1>("s_waitcnt lgkmcnt(4), none outstanding", d, c, wgs);
        run<8, 1>("v_readlane", d, c, wgs); run<8, 2>("v_readlane", d, c, wgs);
        run<9, 1>("v_add imm", d, c, wgs); run<9, 2>("v_add imm", d, c, wgs);
    }
    return 0;
}
