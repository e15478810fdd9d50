// LDS-DMA semantics probe (gfx950): what buffer_load ... lds does with per-lane sources, out-of-range lanes and LDS bases that are
// only 8- / 4-byte aligned -- the facts the staging paths of csrc/conv_corr.hip rely on.
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/_dma_probe tools/dma_probe.hip && gpurun -- tools/_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, int n_bytes, float* out, int mode) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = -7.f;
    __syncthreads();
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, n_bytes, 0x00020000);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    if (mode == 0) {   // dword: lane l of wave w reads src[(w*64+l)*2] (stride-2 gather), odd waves' upper half OOB
        unsigned off = (unsigned)((wv * 64 + lane) * 2) * 4u;
        if (lane >= 48) off = 0x7FFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wv * 64), 4, off, 0, 0, 0);
    } else if (mode == 2) {   // dwordx4 into an LDS base that is only 8-byte aligned (+2 dwords) / 4-byte aligned (+1 for odd waves)
        unsigned off = (unsigned)((wv * 64 + lane) * 4) * 4u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wv * 260 + 2 + (wv & 1)), 16, off, 0, 0, 0);
    } else {           // dwordx4: lane l reads 16 B at src[(w*64+l)*4 ...], lanes >= 60 OOB
        unsigned off = (unsigned)((wv * 64 + lane) * 4) * 4u;
        if (lane >= 60) off = 0x7FFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wv * 256), 16, off, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70 & 0x3f70);   // vmcnt(0)
    __syncthreads();
    for (int i = tid; i < 4096; i += blockDim.x) out[i] = lds[i];
}
int main() {
    const int n = 8192;
    std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, 4096 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> r(4096);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, n * 4, o, mode);
        hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        if (mode == 0) {
            for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) {
                float want = l >= 48 ? 0.f : (float)((w * 64 + l) * 2);
                if (r[w * 64 + l] != want) { if (bad < 5) printf("mode0 w%d l%d got %g want %g\n", w, l, r[w*64+l], want); ++bad; }
            }
            printf("mode 0 (dword gather, OOB lanes): %d mismatches; untouched lds[300]=%g (want -7)\n", bad, r[300]);
        } else if (mode == 2) {
            for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int c = 0; c < 4; ++c) {
                float want = (float)((w * 64 + l) * 4 + c);
                float got = r[w * 260 + 2 + (w & 1) + l * 4 + c];
                if (got != want) { if (bad < 5) printf("mode2 w%d l%d c%d got %g want %g\n", w, l, c, got, want); ++bad; }
            }
            printf("mode 2 (dwordx4 into 8- / 4-byte aligned LDS bases): %d mismatches\n", bad);
        } else {
            for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int c = 0; c < 4; ++c) {
                float want = l >= 60 ? 0.f : (float)((w * 64 + l) * 4 + c);
                float got = r[w * 256 + l * 4 + c];
                if (got != want) { if (bad < 5) printf("mode1 w%d l%d c%d got %g want %g\n", w, l, c, got, want); ++bad; }
            }
            printf("mode 1 (dwordx4, OOB lanes): %d mismatches; untouched lds[2000]=%g (want -7)\n", bad, r[2000]);
        }
    }
    return 0;
}
