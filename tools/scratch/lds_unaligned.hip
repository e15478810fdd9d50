// does gfx950 serve 4-byte-aligned ds_read_b64 / b96 / b128?  prints mismatches and cycles per read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int shift, long long* cyc) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    const int l15 = threadIdx.x & 15, qq = (threadIdx.x >> 4) & 3;
    unsigned addr = (unsigned)(l15 * 66 + qq + shift) * 4u + (unsigned)(size_t)(__attribute__((address_space(3))) float*)s;
    f2 a; f3 b; f4 c;
    asm volatile("ds_read_b64 %0, %1 offset:4\n s_waitcnt lgkmcnt(0)" : "=v"(a) : "v"(addr), "v"(s) : "memory");
    asm volatile("ds_read_b96 %0, %1 offset:8\n s_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(addr) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:12\n s_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(addr) : "memory");
    float* o = out + (threadIdx.x & 63) * 9;
    o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1]; o[4] = b[2]; o[5] = c[0]; o[6] = c[1]; o[7] = c[2]; o[8] = c[3];
    // throughput: 256 dependent-free reads of each kind
    long long t0 = __builtin_readcyclecounter();
    f3 acc = {0, 0, 0};
#pragma unroll 16
    for (int i = 0; i < 256; ++i) { f3 v; asm volatile("ds_read_b96 %0, %1 offset:8" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(v)); acc += v; }
    f2 acc2 = {0, 0};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1b = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < 256; ++i) { f2 v; asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(v)); acc2 += v; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1c = __builtin_readcyclecounter();
    f2 acc3 = {0, 0};
#pragma unroll 16
    for (int i = 0; i < 256; ++i) { f2 v; asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:26" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(v)); acc3 += v; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float acc1 = 0;
#pragma unroll 16
    for (int i = 0; i < 256; ++i) { float v; asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(v)); acc1 += v; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    long long t2 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cyc[0] = t1b - t0; cyc[1] = t2 - t1; cyc[2] = t1c - t1b; cyc[3] = t1 - t1c; }
    if (acc[0] + acc1 + acc2[0] + acc3[1] == 12345.f) out[0] = 0;
}
int main() {
    float* d; long long* c; hipMalloc(&d, 64 * 9 * 4); hipMalloc(&c, 32);
    for (int nt = 64; nt <= 512; nt *= 2) for (int shift = 0; shift < 2; ++shift) {
        k<<<1, nt>>>(d, shift, c);
        float h[64 * 9]; long long hc[4];
        if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { printf("shift %d: FAULT %s\n", shift, hipGetErrorString(hipGetLastError())); return 1; }
        hipMemcpy(hc, c, 32, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 64; ++t) {
            const int base = (t & 15) * 66 + (t >> 4) + shift;
            const int exp[9] = {base + 1, base + 2, base + 2, base + 3, base + 4, base + 3, base + 4, base + 5, base + 6};
            for (int j = 0; j < 9; ++j) bad += h[t * 9 + j] != (float)exp[j];
        }
        printf("threads %d shift %d: %d mismatches; per read: b96 %.1f  b64 %.1f  read2_b32 %.1f  b32 %.1f ticks\n", nt, shift, bad, hc[0] / 256., hc[2] / 256., hc[3] / 256., hc[1] / 256.);
    }
    return 0;
}
