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

// ---- the backward of the same head, where the input needs no gradient (the generators' input is noise): ggan_bn_bwd_act on [M, N] rows
//      + the weight-gradient product x^T gh (+ its column sums) were two launches at the very END of the Generator's backward chain, in
//      front of the step's pack + Adam launch.  A workgroup again owns 32 features for all rows: BatchNorm's backward needs nothing
//      but its own columns (sum g, sum g xh over the rows), and dW[:, its columns] = x^T gh[:, its columns] needs x (staged once, as the
//      forward does) and the gh columns it has just formed (kept in LDS).  gh itself is never written: nothing else reads it.
struct LinBnBwdParams {
    const float* x;        // [M][K]
    const float* gy;       // [M][N]
    const float* h;        // [M][N] the Linear output kept by the forward
    const float* y;        // [M][N] forward output (activation reference) or null
    const float* scale;    // [N]
    const float* mean;     // [N]
    const float* invstd;   // [N]
    float* dw;             // [K][N]
    float* db;             // [N] or null
    float* dscale;         // [N]
    float* doffset;        // [N]
    int M, K, N, XS, R;
    int act;
    float alpha;
};

template <int KPT>      // k values per thread of the product phase: K = 16 * KPT
__global__ __launch_bounds__(LB_THR) void linear_bn_rows_bwd_k(const LinBnBwdParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                          // [M][XS]
    float* ghs = smem + P.M * P.XS;            // [M][32]
    float* red = ghs + P.M * LB_COLS;          // [16][32]
    const int tid = threadIdx.x, c = tid & (LB_COLS - 1), g = tid >> 5;
    const int n0 = blockIdx.x * LB_COLS, col = n0 + c;
    const int K = P.K, K4 = K >> 2, R = P.R;
    for (int u = tid; u < P.M * K4; u += LB_THR) {
        const int m = u / K4, k4 = u - m * K4;
        *reinterpret_cast<float4*>(xs + m * P.XS + k4 * 4) = *reinterpret_cast<const float4*>(P.x + (size_t)m * K + k4 * 4);
    }
    // ---- BatchNorm backward of column `col` over all rows: my R rows, then the 16 row groups in order (as the forward) ----------------
    const float mean = P.mean[col], invstd = P.invstd[col];
    float g1[8], xh[8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        g1[i] = 0.f; xh[i] = 0.f;
        if (i < R) {
            const size_t idx = (size_t)(g * R + i) * P.N + col;
            const float gv = P.gy[idx];
            g1[i] = P.y ? act_grad(gv, P.y[idx], P.act, P.alpha) : gv;
            xh[i] = (P.h[idx] - mean) * invstd;
            s1 += g1[i];
            s2 += g1[i] * xh[i];
        }
    }
    red[g * LB_COLS + c] = s1;
    __syncthreads();
    float sum_g = 0.f;
#pragma unroll
    for (int j = 0; j < LB_GROUPS; ++j) sum_g += red[j * LB_COLS + c];
    __syncthreads();
    red[g * LB_COLS + c] = s2;
    __syncthreads();
    float sum_gx = 0.f;
#pragma unroll
    for (int j = 0; j < LB_GROUPS; ++j) sum_gx += red[j * LB_COLS + c];
    const float inv_cnt = 1.f / (float)P.M;
    const float kk = P.scale[col] * invstd, mg = sum_g * inv_cnt, mgx = sum_gx * inv_cnt;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < R) ghs[(g * R + i) * LB_COLS + c] = kk * (g1[i] - mg - xh[i] * mgx);
    if (g == 0) {
        P.dscale[col] = sum_gx;
        P.doffset[col] = sum_g;
    }
    __syncthreads();                           // (x staged, gh columns complete)
    // ---- dW[k][col] = sum_m x[m][k] gh[m][col] for my KPT values of k, rows in order ------------------------------------------------
    float acc[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) acc[j] = 0.f;
    const float* xk = xs + g * KPT;
    float bsum = 0.f;
    for (int m = 0; m < P.M; ++m) {
        const float gv = ghs[m * LB_COLS + c];
        bsum += gv;
#pragma unroll
        for (int j4 = 0; j4 < KPT / 4; ++j4) {
            const float4 xv = *reinterpret_cast<const float4*>(xk + m * P.XS + j4 * 4);
            acc[j4 * 4 + 0] = fmaf(xv.x, gv, acc[j4 * 4 + 0]);
            acc[j4 * 4 + 1] = fmaf(xv.y, gv, acc[j4 * 4 + 1]);
            acc[j4 * 4 + 2] = fmaf(xv.z, gv, acc[j4 * 4 + 2]);
            acc[j4 * 4 + 3] = fmaf(xv.w, gv, acc[j4 * 4 + 3]);
        }
    }
#pragma unroll
    for (int j = 0; j < KPT; ++j) P.dw[(size_t)(g * KPT + j) * P.N + col] = acc[j];
    if (g == 0 && P.db) P.db[col] = bsum;
}

}  // namespace

extern "C" int ggan_linear_bn_rows_bwd(const float* x, const float* gy, const float* h, const float* y, const float* scale,
                                       const float* save_mean, const float* save_invstd, float* dw, float* db, float* dscale,
                                       float* doffset, int M, int K, int N, int act, float alpha, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && gy && h && scale && save_mean && save_invstd && dw && dscale && doffset, "null pointer");
    GGAN_CHECK_ARG(M > 0 && K > 0 && N > 0 && (y || act == GGAN_ACT_NONE), "bad shape");
    if (M > 128 || (M & 15) || (K != 64 && K != 128 && K != 256) || (N & 31)) return 1;
    if (((uintptr_t)x) & 15) return 1;
    LinBnBwdParams P;
    P.x = x; P.gy = gy; P.h = h; P.y = act != GGAN_ACT_NONE ? y : nullptr; P.scale = scale; P.mean = save_mean; P.invstd = save_invstd;
    P.dw = dw; P.db = db; P.dscale = dscale; P.doffset = doffset;
    P.M = M; P.K = K; P.N = N; P.XS = K + 4; P.R = M / LB_GROUPS; P.act = act; P.alpha = alpha;
    const size_t shmem = ((size_t)M * P.XS + (size_t)M * LB_COLS + LB_GROUPS * LB_COLS) * sizeof(float);
    if (shmem > 160 * 1024) return 1;
    hipStream_t s = (hipStream_t)stream;
    const double fl = 2.0 * M * K * (double)N, bytes = 4.0 * ((double)M * K + (double)K * N + 3.0 * M * N);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bn_rows_bwd_k<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bn_rows_bwd_k<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(linear_bn_rows_bwd_k<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    const dim3 grid(N / LB_COLS), block(LB_THR);
    if (K == 64) { GGAN_LAUNCH("linear_bn_rows_bwd_k", fl, bytes, linear_bn_rows_bwd_k<4>, grid, block, shmem, s, P); }
    else if (K == 128) { GGAN_LAUNCH("linear_bn_rows_bwd_k", fl, bytes, linear_bn_rows_bwd_k<8>, grid, block, shmem, s, P); }
    else { GGAN_LAUNCH("linear_bn_rows_bwd_k", fl, bytes, linear_bn_rows_bwd_k<16>, grid, block, shmem, s, P); }
    return 0;
}

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
