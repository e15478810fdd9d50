// Linear + BatchNorm over the batch axis + activation in ONE launch: the head of every Generator of the image scripts,
//   output = Linear('Generator.Input', DIM_LATENT, 4*4*4*DIM, noise);  output = Batchnorm('Generator.BN1', [0], output);  relu
//   (gan_inference_cifar10.py:134-138, gmgan_inference_cifar10.py:176-179, gan_inference_mnist.py:122-126).
//
// The product is tiny (64 x 4096 x 128: 0.07 GFLOP) and sits at the head of the Generator chain, which is the longer of the two
// chains of a nets pass: as ggan_gemm (10 us: 64-row products are a poor fit for the 64x64-tile split-K kernel) + ggan_bn_fwd_train
// (6 us) it cost 16 us plus a kernel boundary on the critical path of every step.  Here a workgroup owns 32 output features for
// ALL rows of the minibatch, so the statistics BatchNorm needs (batch axis = the rows) never leave the workgroup:
//   * the whole input [M <= 128, K <= 256] and the weight slice [K, 32] sit in LDS (34 + 16 KB at 64 x 128);
//   * thread (row group g of 16, column c) accumulates R = M / 16 rows of column c with plain fp32 FMAs in k order (exact fp32 like the
//     MFMA path; 1024 FMAs per thread), x read as 16-byte broadcasts, w as conflict-free dwords;
//   * mean and centred variance over the 16 row groups through LDS in fixed order (deterministic, two passes as ggan_bn_fwd_train),
//     then y = act(scale * (h - mean) * invstd + offset); h (the Linear output, BatchNorm's input: its backward reads it), y,
//     mean and invstd are written -- what ggan_bn_fwd_train would have left for ggan_bn_bwd_act.
#include "common.h"
using namespace ggan;

namespace {

constexpr int LB_COLS = 32, LB_GROUPS = 16, LB_THR = LB_COLS * LB_GROUPS;      // 8 waves per workgroup: two per SIMD hide each other's LDS latency

struct LinBnParams {
    const float* x;        // [M][K]
    const float* w;        // [K][N]
    const float* b;        // [N] or null
    const float* scale;    // [N]
    const float* offset;   // [N]
    float* h;              // [M][N]
    float* y;              // [M][N]
    float* save_mean;      // [N]
    float* save_invstd;    // [N]
    int M, K, N, XS;       // XS: LDS row stride of x (K + 4: rows stay 16-byte aligned)
    float eps, alpha;
    int act;
};

template <int R>
__global__ __launch_bounds__(LB_THR) void linear_bn_rows_k(const LinBnParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                          // [M][XS]
    float* ws = smem + P.M * P.XS;             // [K][32]
    float* red = ws + P.K * LB_COLS;           // [8][32]
    const int tid = threadIdx.x, c = tid & (LB_COLS - 1), g = tid >> 5;
    const int n0 = blockIdx.x * LB_COLS;
    const int K = P.K, K4 = K >> 2;
    // ---- stage x (whole) and the weight slice ------------------------------------------------------------------------
    for (int u = tid; u < P.M * K4; u += LB_THR) {
        const int m = u / K4, k4 = u - m * K4;
        *reinterpret_cast<float4*>(xs + m * P.XS + k4 * 4) = *reinterpret_cast<const float4*>(P.x + (size_t)m * K + k4 * 4);
    }
    for (int u = tid; u < K * (LB_COLS / 4); u += LB_THR) {
        const int k = u >> 3, c4 = u & 7;
        *reinterpret_cast<float4*>(ws + k * LB_COLS + c4 * 4) = *reinterpret_cast<const float4*>(P.w + (size_t)k * P.N + n0 + c4 * 4);
    }
    __syncthreads();
    // ---- h[r][c] = sum_k x[r][k] * w[k][c] (+ b[c]) for my R rows ----------------------------------------------------------
    float acc[R];
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.f;
    const float* xr = xs + (g * R) * P.XS;
    // operands of step k+4 are fetched before the FMAs of step k (one wave per SIMD would otherwise sit out an LDS round trip per step)
    float4 xv[R], xn[R];
    float wv[4], wn[4];
#pragma unroll
    for (int i = 0; i < R; ++i) xv[i] = *reinterpret_cast<const float4*>(xr + i * P.XS);
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[j] = ws[j * LB_COLS + c];
    for (int k = 0; k < K; k += 4) {
        const int kn = k + 4 < K ? k + 4 : k;
#pragma unroll
        for (int i = 0; i < R; ++i) xn[i] = *reinterpret_cast<const float4*>(xr + i * P.XS + kn);
#pragma unroll
        for (int j = 0; j < 4; ++j) wn[j] = ws[(kn + j) * LB_COLS + c];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            acc[i] = fmaf(xv[i].x, wv[0], acc[i]);
            acc[i] = fmaf(xv[i].y, wv[1], acc[i]);
            acc[i] = fmaf(xv[i].z, wv[2], acc[i]);
            acc[i] = fmaf(xv[i].w, wv[3], acc[i]);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) xv[i] = xn[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = wn[j];
    }
    const float bias = P.b ? P.b[n0 + c] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) { acc[i] += bias; s += acc[i]; }
    // ---- batch statistics of column c over all M rows: 8 partials in row-group order ------------------------------------
    red[g * LB_COLS + c] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < LB_GROUPS; ++j) tot += red[j * LB_COLS + c];
    const float inv_cnt = 1.f / (float)P.M;
    const float mean = tot * inv_cnt;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) { const float d = acc[i] - mean; q += d * d; }
    __syncthreads();
    red[g * LB_COLS + c] = q;
    __syncthreads();
    float qt = 0.f;
#pragma unroll
    for (int j = 0; j < LB_GROUPS; ++j) qt += red[j * LB_COLS + c];
    const float invstd = 1.f / sqrtf(qt * inv_cnt + P.eps);
    const float sc = P.scale[n0 + c], of = P.offset[n0 + c];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const size_t idx = (size_t)(g * R + i) * P.N + n0 + c;
        P.h[idx] = acc[i];
        P.y[idx] = act_apply(sc * ((acc[i] - mean) * invstd) + of, P.act, P.alpha);
    }
    if (g == 0) {
        P.save_mean[n0 + c] = mean;
        P.save_invstd[n0 + c] = invstd;
    }
}

}  // namespace

extern "C" int ggan_linear_bn_rows_fwd(const float* x, const float* w, const float* b, const float* scale, const float* offset, float* h,
                                       float* y, float* save_mean, float* save_invstd, int M, int K, int N, float eps, int act,
                                       float alpha, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && w && scale && offset && h && y && save_mean && save_invstd, "null pointer");
    GGAN_CHECK_ARG(M > 0 && K > 0 && N > 0, "bad shape");
    // covered: whole minibatch in one workgroup's LDS, rows in 16 equal groups, 16-byte rows
    if (M > 128 || (M & 15) || (K & 3) || K > 256 || (N & 31)) return 1;
    if (((uintptr_t)x | (uintptr_t)w) & 15) return 1;
    LinBnParams P;
    P.x = x; P.w = w; P.b = b; P.scale = scale; P.offset = offset; P.h = h; P.y = y; P.save_mean = save_mean; P.save_invstd = save_invstd;
    P.M = M; P.K = K; P.N = N; P.XS = K + 4; P.eps = eps; P.alpha = alpha; P.act = act;
    const size_t shmem = ((size_t)M * P.XS + (size_t)K * LB_COLS + LB_GROUPS * LB_COLS) * sizeof(float);
    if (shmem > 160 * 1024) return 1;
    hipStream_t s = (hipStream_t)stream;
    const double fl = 2.0 * M * K * (double)N, bytes = 4.0 * ((double)M * K + (double)K * N + 2.0 * M * N);
    static bool once = false;
    if (!once) {      // (every instantiation: a request above the 64 KB default must not depend on which row count came first)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bn_rows_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bn_rows_k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bn_rows_k<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bn_rows_k<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bn_rows_k<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bn_rows_k<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bn_rows_k<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bn_rows_k<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    const dim3 grid(N / LB_COLS), block(LB_THR);
    switch (M / LB_GROUPS) {
        case 1: GGAN_LAUNCH("linear_bn_rows_k", fl, bytes, linear_bn_rows_k<1>, grid, block, shmem, s, P); break;
        case 2: GGAN_LAUNCH("linear_bn_rows_k", fl, bytes, linear_bn_rows_k<2>, grid, block, shmem, s, P); break;
        case 3: GGAN_LAUNCH("linear_bn_rows_k", fl, bytes, linear_bn_rows_k<3>, grid, block, shmem, s, P); break;
        case 4: GGAN_LAUNCH("linear_bn_rows_k", fl, bytes, linear_bn_rows_k<4>, grid, block, shmem, s, P); break;
        case 5: GGAN_LAUNCH("linear_bn_rows_k", fl, bytes, linear_bn_rows_k<5>, grid, block, shmem, s, P); break;
        case 6: GGAN_LAUNCH("linear_bn_rows_k", fl, bytes, linear_bn_rows_k<6>, grid, block, shmem, s, P); break;
        case 7: GGAN_LAUNCH("linear_bn_rows_k", fl, bytes, linear_bn_rows_k<7>, grid, block, shmem, s, P); break;
        case 8: GGAN_LAUNCH("linear_bn_rows_k", fl, bytes, linear_bn_rows_k<8>, grid, block, shmem, s, P); break;
        default: return 1;
    }
    return 0;     // (GGAN_LAUNCH checks the launch: a refused one returns -2 with the error text)
}
