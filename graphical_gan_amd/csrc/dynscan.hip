// The state-space scripts' transition operator as ONE scan per direction (ssgan_inference_moving_mnist.py:98-141):
//
//   ImplicitOperator(z, eps):  h1 = lrelu([z | eps] W_in + b_in);  h2 = lrelu(h1 W_1 + b_1);  o = h2 W_out + b_out
//                              z' = o + z                     (OP_DYN_MODE 'res')
//                              z' = o + z ZW + b_zw           ('res_w', ssgan_inference_chairs.py)
//   DynamicGenerator:          z_1 .. z_{LEN-1} by LEN-1 sequential applications, one eps per sequence
//
// The reference (and the layer-by-layer composition here) spends ~8 launches per application and direction on [B, 16..256]
// matrices: B = 32 rows, 15 steps, 0.07 GFLOP in ~350 launches per iteration.  The steps are sequential in t but the rows are
// independent, so a workgroup owns ONE sequence and walks its LEN-1 steps with the activations in LDS; the weights (W_1: 256 KB)
// stream from L2 each step.  Everything is exact fp32 FMA in a fixed order (deterministic).
//   forward : thread n of 256 owns hidden unit n: h1[n], then h2[n] = sum_k h1[k] W_1[k, n] (W_1 rows are contiguous in n: every
//             wave-load is a coalesced 256-byte run), then the dl outputs by a block reduction.  h1 / h2 / z_t are kept for backward.
//   backward: walks t down from LEN-2.  g_h2[n] = lrelu'(h2[n]) * sum_j g_o[j] W_out[n, j];  g_h1[k] needs the TRANSPOSED product
//             sum_n g_h2[n] W_1[k, n]: wave w takes rows k = w, w+4, ...: lanes hold 4 consecutive n each (one float4 load covers a
//             whole 1 KB row), dot with the lane's four g_h2 values, wave shuffle-reduce.  The masked g_h1 / g_h2 / g_o rows and the
//             operator inputs are written out; the weight gradients are three products over all (t, b) rows at once (caller: GEMM).
#include "common.h"
using namespace ggan;

namespace {

constexpr int H = 256;          // DIM_OP of both scripts (:55); other widths take the layer-by-layer path

struct ScanParams {
    int B, T, dl, dt;           // sequences, steps (LEN-1), latent width, width of the second operator input
    const float* z0;            // [B, dl]
    const float* eps;           // [B, dt]
    const float* w_in; const float* b_in;       // [dl+dt, H], [H]
    const float* w_1; const float* b_1;         // [H, H], [H]
    const float* w_out; const float* b_out;     // [H, dl], [dl]
    const float* zw; const float* b_zw;         // [dl, dl], [dl] or NULL ('res')
    float alpha;
    float* zs;                  // [B, T+1, dl]   (z_0 copied in)
    float* h1; float* h2;       // [T, B, H] post-activation
    // backward
    const float* g_zs;          // [B, T+1, dl]
    float* G1; float* G2;       // [T, B, H] masked gradients at the two hidden layers
    float* Go;                  // [T, B, dl]
    float* Xin;                 // [T, B, dl+dt] operator inputs [z_t | eps]
    float* d_z0; float* d_eps;  // [B, dl], [B, dt]
};

constexpr int MAXD = 16;        // dl, dt <= 16

__device__ __forceinline__ float lrelu(float v, float a) { return fmaxf(a * v, v); }

__global__ __launch_bounds__(H) void dyn_scan_fwd_k(const ScanParams P) {
    __shared__ float in_s[2 * MAXD];        // [z_t | eps]
    __shared__ float h_s[H];
    __shared__ float red[4][MAXD];
    const int b = blockIdx.x, n = threadIdx.x, lane = n & 63, wave = n >> 6;
    const int dl = P.dl, dt = P.dt, din = dl + dt;
    if (n < dl) { const float v = P.z0[b * dl + n]; in_s[n] = v; P.zs[(size_t)b * (P.T + 1) * dl + n] = v; }
    if (n >= MAXD && n < MAXD + dt) in_s[dl + (n - MAXD)] = P.eps[b * dt + (n - MAXD)];
    const float bi = P.b_in[n], b1 = P.b_1[n];
    float wout[MAXD];
#pragma unroll
    for (int j = 0; j < MAXD; ++j) wout[j] = j < dl ? P.w_out[n * dl + j] : 0.f;
    __syncthreads();
    for (int t = 0; t < P.T; ++t) {
        // h1[n] = lrelu(sum_i in[i] W_in[i, n] + b_in[n])
        float a = bi;
#pragma unroll 4
        for (int i = 0; i < din; ++i) a = fmaf(in_s[i], P.w_in[i * H + n], a);
        a = lrelu(a, P.alpha);
        const size_t row = ((size_t)t * P.B + b) * H;
        P.h1[row + n] = a;
        h_s[n] = a;
        __syncthreads();
        // h2[n] = lrelu(sum_k h1[k] W_1[k, n] + b_1[n]): 8 independent chains
        float s[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] = 0.f;
#pragma unroll 4
        for (int k = 0; k < H; k += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] = fmaf(h_s[k + u], P.w_1[(size_t)(k + u) * H + n], s[u]);
        }
        float c = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])) + b1;
        c = lrelu(c, P.alpha);
        P.h2[row + n] = c;
        // o[j] = sum_n h2[n] W_out[n, j]: per-wave shuffle sums, then 4 partials per output through LDS
#pragma unroll
        for (int j = 0; j < MAXD; ++j) {
            if (j < dl) {
                const float v = wave_sum(c * wout[j]);
                if (lane == 0) red[wave][j] = v;
            }
        }
        __syncthreads();          // (also: every thread is done reading h_s of this step)
        if (n < dl) {
            float o = ((red[0][n] + red[1][n]) + (red[2][n] + red[3][n])) + P.b_out[n];
            float zn;
            if (P.zw) {           // res_w: o + z_t ZW + b_zw
                float r = P.b_zw[n];
                for (int i = 0; i < dl; ++i) r = fmaf(in_s[i], P.zw[i * dl + n], r);
                zn = o + r;
            } else {
                zn = o + in_s[n];
            }
            P.zs[((size_t)b * (P.T + 1) + t + 1) * dl + n] = zn;
            red[0][n] = zn;       // (parked: in_s is still being read by the res_w products of the other threads)
        }
        __syncthreads();
        if (n < dl) in_s[n] = red[0][n];
        __syncthreads();
    }
}

__global__ __launch_bounds__(H) void dyn_scan_bwd_k(const ScanParams P) {
    __shared__ float go_s[MAXD];            // gradient arriving at z_{t+1} (downstream + carried)
    __shared__ float g1_s[H];               // masked g_h1 of this step
    __shared__ float gin_s[2 * MAXD];
    __shared__ float in_s[2 * MAXD];
    const int b = blockIdx.x, n = threadIdx.x, lane = n & 63, wave = n >> 6;
    const int dl = P.dl, dt = P.dt, din = dl + dt, T = P.T;
    float wout[MAXD];
#pragma unroll
    for (int j = 0; j < MAXD; ++j) wout[j] = j < dl ? P.w_out[n * dl + j] : 0.f;
    float deps = 0.f;                       // thread n < dt: accumulated d eps[n]
    if (n < dl) go_s[n] = P.g_zs[((size_t)b * (T + 1) + T) * dl + n];
    if (n >= MAXD && n < MAXD + dt) in_s[dl + (n - MAXD)] = P.eps[b * dt + (n - MAXD)];
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        const size_t row = ((size_t)t * P.B + b) * H;
        if (n < dl) {
            in_s[n] = P.zs[((size_t)b * (T + 1) + t) * dl + n];
            P.Go[((size_t)t * P.B + b) * dl + n] = go_s[n];
        }
        // g_h2[n] = lrelu'(h2[n]) * sum_j g_o[j] W_out[n, j]
        float g2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXD; ++j) if (j < dl) g2 = fmaf(go_s[j], wout[j], g2);
        g2 *= P.h2[row + n] > 0.f ? 1.f : P.alpha;
        P.G2[row + n] = g2;
        // g_h1[k] = lrelu'(h1[k]) * sum_n g_h2[n] W_1[k, n]: wave w takes rows k = w, w+4, ...; lanes hold 4 consecutive n
        float4 gq;
        {
            // lane l needs g_h2[4l .. 4l+3] of the WHOLE block: through LDS
            __syncthreads();      // (g1_s of the previous step fully consumed, in_s / go_s written)
            g1_s[n] = g2;
            __syncthreads();
            gq = *reinterpret_cast<const float4*>(g1_s + 4 * lane);
            __syncthreads();      // (g1_s is reused for g_h1 below)
        }
        // (8 rows per trip: the eight 1 KB row loads are in flight together and the eight shuffle reductions interleave)
        for (int k0 = wave; k0 < H; k0 += 32) {
            float4 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = *reinterpret_cast<const float4*>(P.w_1 + (size_t)(k0 + 4 * u) * H + 4 * lane);
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = fmaf(gq.x, w[u].x, fmaf(gq.y, w[u].y, fmaf(gq.z, w[u].z, gq.w * w[u].w)));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] += __shfl_xor(v[u], o, 64);
            }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < 8; ++u) g1_s[k0 + 4 * u] = v[u];
            }
        }
        __syncthreads();
        const float g1 = g1_s[n] * (P.h1[row + n] > 0.f ? 1.f : P.alpha);
        P.G1[row + n] = g1;
        __syncthreads();
        g1_s[n] = g1;
        __syncthreads();
        // g_in[i] = sum_k g_h1[k] W_in[i, k]: wave w takes rows i = w, w+4, ...
        {
            const float4 gk = *reinterpret_cast<const float4*>(g1_s + 4 * lane);
            for (int i = wave; i < din; i += 4) {
                const float4 w = *reinterpret_cast<const float4*>(P.w_in + (size_t)i * H + 4 * lane);
                float v = fmaf(gk.x, w.x, fmaf(gk.y, w.y, fmaf(gk.z, w.z, gk.w * w.w)));
                v = wave_sum(v);
                if (lane == 0) gin_s[i] = v;
            }
        }
        if (n < din) P.Xin[((size_t)t * P.B + b) * din + n] = in_s[n];
        __syncthreads();
        // carried gradient at z_t: operator input path + residual path + downstream
        float gz = 0.f;
        if (n < dl) {
            gz = gin_s[n];
            if (P.zw) { for (int j = 0; j < dl; ++j) gz = fmaf(go_s[j], P.zw[n * dl + j], gz); }
            else gz += go_s[n];
            gz += P.g_zs[((size_t)b * (T + 1) + t) * dl + n];
        }
        if (n < dt) deps += gin_s[dl + n];
        __syncthreads();
        if (n < dl) go_s[n] = gz;
        __syncthreads();
    }
    if (n < dl && P.d_z0) P.d_z0[b * dl + n] = go_s[n];
    if (n < dt && P.d_eps) P.d_eps[b * dt + n] = deps;
}

}  // namespace

extern "C" {

int ggan_dyn_scan_fwd(int B, int T, int dl, int dt, int Hdim, const float* z0, const float* eps, const float* w_in, const float* b_in,
                      const float* w_1, const float* b_1, const float* w_out, const float* b_out, const float* zw, const float* b_zw,
                      float alpha, float* zs, float* h1, float* h2, ggan_stream_t stream) {
    GGAN_CHECK_ARG(z0 && eps && w_in && b_in && w_1 && b_1 && w_out && b_out && zs && h1 && h2, "null pointer");
    GGAN_CHECK_ARG(B > 0 && T > 0 && dl > 0 && dl <= MAXD && dt > 0 && dt <= MAXD && Hdim == H, "unsupported shape (H must be 256, widths <= 16)");
    GGAN_CHECK_ARG((zw == nullptr) == (b_zw == nullptr), "zw and b_zw go together");
    ScanParams P;
    memset(&P, 0, sizeof(P));
    P.B = B; P.T = T; P.dl = dl; P.dt = dt; P.z0 = z0; P.eps = eps; P.w_in = w_in; P.b_in = b_in; P.w_1 = w_1; P.b_1 = b_1;
    P.w_out = w_out; P.b_out = b_out; P.zw = zw; P.b_zw = b_zw; P.alpha = alpha; P.zs = zs; P.h1 = h1; P.h2 = h2;
    const double fl = 2.0 * B * T * ((double)(dl + dt) * H + (double)H * H + (double)H * dl);
    GGAN_LAUNCH("dyn_scan_fwd_k", fl, 0, dyn_scan_fwd_k, dim3(B), dim3(H), 0, (hipStream_t)stream, P);
    return 0;
}

int ggan_dyn_scan_bwd(int B, int T, int dl, int dt, int Hdim, const float* g_zs, const float* zs, const float* eps, const float* h1,
                      const float* h2, const float* w_in, const float* w_1, const float* w_out, const float* zw, float alpha, float* G1,
                      float* G2, float* Go, float* Xin, float* d_z0, float* d_eps, ggan_stream_t stream) {
    GGAN_CHECK_ARG(g_zs && zs && eps && h1 && h2 && w_in && w_1 && w_out && G1 && G2 && Go && Xin, "null pointer");
    GGAN_CHECK_ARG(B > 0 && T > 0 && dl > 0 && dl <= MAXD && dt > 0 && dt <= MAXD && Hdim == H, "unsupported shape (H must be 256, widths <= 16)");
    GGAN_CHECK_ARG((((uintptr_t)w_1 | (uintptr_t)w_in) & 15) == 0, "w_1 / w_in must be 16-byte aligned");
    ScanParams P;
    memset(&P, 0, sizeof(P));
    P.B = B; P.T = T; P.dl = dl; P.dt = dt; P.g_zs = g_zs; P.zs = const_cast<float*>(zs); P.eps = eps;
    P.h1 = const_cast<float*>(h1); P.h2 = const_cast<float*>(h2); P.w_in = w_in; P.w_1 = w_1; P.w_out = w_out; P.zw = zw;
    P.alpha = alpha; P.G1 = G1; P.G2 = G2; P.Go = Go; P.Xin = Xin; P.d_z0 = d_z0; P.d_eps = d_eps;
    const double fl = 2.0 * B * T * ((double)(dl + dt) * H + (double)H * H + (double)H * dl);
    GGAN_LAUNCH("dyn_scan_bwd_k", fl, 0, dyn_scan_bwd_k, dim3(B), dim3(H), 0, (hipStream_t)stream, P);
    return 0;
}

}  // extern "C"
