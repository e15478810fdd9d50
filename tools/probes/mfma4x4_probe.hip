// layout probe of v_mfma_f32_4x4x1_16b_f32: which (block, row, col) each lane / register holds.  hipcc --offload-arch=gfx950 -o probe probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int l = threadIdx.x;
    // A value encodes (lane): a = 1 + l ; B value encodes b = 1000 * (1 + l)  -> D = a * b identifies the (A lane, B lane) pair per output
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + l), 1000.f * (1 + l), c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = c[v];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 4; ++v) {
            const float expect = 1000.f * (1 + 4 * (l / 4) + v) * (1 + l);      // D[i = v][j = l % 4] of block l / 4 = A(lane 4 blk + v) * B(lane l)
            if (h[l * 4 + v] != expect) { if (bad < 8) printf("lane %d v%d: got %.0f expected %.0f\n", l, v, h[l * 4 + v], expect); ++bad; }
        }
    printf("4x4x1_16b layout D[vgpr i][lane 4b+j] = A[lane 4b+i] * B[lane 4b+j]: %s (%d mismatches)\n", bad ? "NO" : "confirmed", bad);
    return 0;
}
