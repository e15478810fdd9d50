// Training-mode batch normalisation (batch statistics, biased variance, eps inside the sqrt).
// Replaces tf.nn.fused_batch_norm(NCHW) (tflib/ops/batchnorm.py:29-30) and the
// tf.nn.moments + tf.nn.batch_normalization branch (:74-87) -- SURVEY.md A.4 / K8, K8', K9.
//
// HBM-bound: algorithmic traffic fwd = 2 reads + 1 write of the activation (stat passes hit L2),
// bwd = 2 reads (x, gy) + 1 write.  Two layouts:
//   HW > 1 : NCHW, one workgroup per channel; lanes stream the N contiguous HW-chunks; wave-shuffle
//            + LDS tree for the per-channel sums (two-pass mean / centred variance for accuracy).
//   HW == 1: [N, C] rows (Generator.BN1 over [B,4096]); lanes run along C so every row read is a
//            coalesced 256-B segment; a 64x4 thread tile splits N four ways and combines through LDS.
#include "common.h"
#include "conv.h"
using namespace ggan;

namespace {

// dL/d(bn output): the incoming gradient with the fused activation's derivative applied on load (GyMask, conv.h)
__device__ __forceinline__ float ld_gy(const float* __restrict__ gy, const GyMask& mk, size_t idx) {
    float g = gy[idx];
    if (mk.act) g = act_grad(g, mk.ref[idx], mk.act, mk.alpha);
    return g;
}

constexpr int kThreads = 1024;

// Register-resident variants: when a channel's N*HW values fit 16 per thread of a 1024-thread workgroup the activation is read from HBM/L2 ONCE
// (mean, centred variance and the normalised write all come from registers): 2 passes of traffic instead of 4.
constexpr int kRegE = 16;

__global__ __launch_bounds__(kThreads) void bn_fwd_nchw_reg_k(const float* __restrict__ x, const float* __restrict__ scale,
                                                              const float* __restrict__ offset, float* __restrict__ y,
                                                              float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                              int N, int C, int HW, float eps, int act, float alpha, FastDiv dHW) {
    // (element index -> (image, pixel) by one v_mul_hi: a division by the run-time HW is ~35 instructions, twice per element)
    __shared__ float sm[32], sm2[32];      // (one buffer per reduction: no barrier in front of the second one)
    const int c = blockIdx.x;
    const int total = N * HW;
    const float inv_cnt = 1.f / (float)total;
    float v[kRegE];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kRegE; ++j) {
        const int i = threadIdx.x + j * kThreads;
        float t = 0.f;
        if (i < total) {
            const int n = (int)fdiv((uint32_t)i, dHW), p = i - n * HW;
            t = x[((size_t)n * C + c) * HW + p];
        }
        v[j] = t;
        s += t;
    }
    const float mean = block_sum_fresh(s, sm) * inv_cnt;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < kRegE; ++j) {
        const int i = threadIdx.x + j * kThreads;
        const float d = v[j] - mean;
        if (i < total) q += d * d;
    }
    const float var = block_sum_fresh(q, sm2) * inv_cnt;
    const float invstd = 1.f / sqrtf(var + eps);
    const float g = scale[c], b = offset[c];
#pragma unroll
    for (int j = 0; j < kRegE; ++j) {
        const int i = threadIdx.x + j * kThreads;
        if (i < total) {
            const int n = (int)fdiv((uint32_t)i, dHW), p = i - n * HW;
            y[((size_t)n * C + c) * HW + p] = act_apply(g * ((v[j] - mean) * invstd) + b, act, alpha);
        }
    }
    if (threadIdx.x == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
    }
}

__global__ __launch_bounds__(kThreads) void bn_bwd_nchw_reg_k(const float* __restrict__ x, const float* __restrict__ gy, GyMask mk,
                                                              const float* __restrict__ scale, const float* __restrict__ save_mean,
                                                              const float* __restrict__ save_invstd, float* __restrict__ gx,
                                                              float* __restrict__ gscale, float* __restrict__ goffset, float* __restrict__ gx_sum, int N,
                                                              int C, int HW, FastDiv dHW) {
    __shared__ float sm[32], sm2[32];      // (the two sums in one pass; the third reduction has its own buffer)
    const int c = blockIdx.x;
    const int total = N * HW;
    const float mean = save_mean[c], invstd = save_invstd[c];
    float xh[kRegE], g[kRegE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < kRegE; ++j) {
        const int i = threadIdx.x + j * kThreads;
        float a = 0.f, b = 0.f;
        if (i < total) {
            const int n = (int)fdiv((uint32_t)i, dHW), p = i - n * HW;
            const size_t idx = ((size_t)n * C + c) * HW + p;
            b = ld_gy(gy, mk, idx);
            a = (x[idx] - mean) * invstd;
        }
        xh[j] = a; g[j] = b;
        s1 += b;
        s2 += b * a;
    }
    block_sum2_fresh(s1, s2, sm);
    const float sum_g = s1, sum_gx = s2;
    const float inv_cnt = 1.f / (float)total;
    const float k = scale[c] * invstd, mg = sum_g * inv_cnt, mgx = sum_gx * inv_cnt;
    float s3 = 0.f;
#pragma unroll
    for (int j = 0; j < kRegE; ++j) {
        const int i = threadIdx.x + j * kThreads;
        if (i < total) {
            const int n = (int)fdiv((uint32_t)i, dHW), p = i - n * HW;
            const float o = k * (g[j] - mg - xh[j] * mgx);
            gx[((size_t)n * C + c) * HW + p] = o;
            s3 += o;
        }
    }
    if (gx_sum) {   // channel sum of the result: the bias gradient of the layer below (mathematically 0, kept for parity)
        const float t = block_sum_fresh(s3, sm2);
        if (threadIdx.x == 0) gx_sum[c] = t;
    }
    if (threadIdx.x == 0) {
        gscale[c] = sum_gx;
        goffset[c] = sum_g;
    }
}

// The backward with 16-byte accesses (HW % 4 == 0: 4 consecutive elements of a channel share their image): a thread holds 4 float4 of
// each operand, NT = 256 threads for up to 4096 elements per channel (4 waves: cheaper block sums), 1024 up to 16384.  One quarter
// of the load / store instructions and of the index arithmetic: 11.2 -> 9.1 us per launch (three operands to read; the forward,
// with one, measured no faster this way and keeps the scalar kernel).  The per-thread partial sums cover other elements than in the
// scalar kernel, so results differ from it by summation order only.
__device__ __forceinline__ float4 ld_gy4(const float* __restrict__ gy, const GyMask& mk, size_t idx) {
    float4 g = *reinterpret_cast<const float4*>(gy + idx);
    if (mk.act) {
        const float4 r = *reinterpret_cast<const float4*>(mk.ref + idx);
        g.x = act_grad(g.x, r.x, mk.act, mk.alpha); g.y = act_grad(g.y, r.y, mk.act, mk.alpha);
        g.z = act_grad(g.z, r.z, mk.act, mk.alpha); g.w = act_grad(g.w, r.w, mk.act, mk.alpha);
    }
    return g;
}

template <int NT>
__global__ __launch_bounds__(NT) void bn_bwd_nchw_reg4_k(const float* __restrict__ x, const float* __restrict__ gy, GyMask mk,
                                                         const float* __restrict__ scale, const float* __restrict__ save_mean,
                                                         const float* __restrict__ save_invstd, float* __restrict__ gx,
                                                         float* __restrict__ gscale, float* __restrict__ goffset,
                                                         float* __restrict__ gx_sum, int N, int C, int HW, FastDiv dHW) {
    __shared__ float sm[32], sm2[32];      // (the two sums in one pass; the third reduction has its own buffer)
    const int c = blockIdx.x;
    const int total = N * HW, units = total >> 2;
    const float mean = save_mean[c], invstd = save_invstd[c];
    float4 xh[4], g[4];
    size_t off[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int u = threadIdx.x + j * NT;
        xh[j] = g[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        off[j] = 0;
        if (u < units) {
            const int i = 4 * u, n = (int)fdiv((uint32_t)i, dHW), p = i - n * HW;
            off[j] = ((size_t)n * C + c) * HW + p;
            g[j] = ld_gy4(gy, mk, off[j]);
            const float4 t = *reinterpret_cast<const float4*>(x + off[j]);
            xh[j] = make_float4((t.x - mean) * invstd, (t.y - mean) * invstd, (t.z - mean) * invstd, (t.w - mean) * invstd);
            s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
            s2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
        }
    }
    block_sum2_fresh(s1, s2, sm);
    const float sum_g = s1, sum_gx = s2;
    const float inv_cnt = 1.f / (float)total;
    const float k = scale[c] * invstd, mg = sum_g * inv_cnt, mgx = sum_gx * inv_cnt;
    float s3 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (threadIdx.x + j * NT < units) {
            float4 o;
            o.x = k * (g[j].x - mg - xh[j].x * mgx); o.y = k * (g[j].y - mg - xh[j].y * mgx);
            o.z = k * (g[j].z - mg - xh[j].z * mgx); o.w = k * (g[j].w - mg - xh[j].w * mgx);
            *reinterpret_cast<float4*>(gx + off[j]) = o;
            s3 += (o.x + o.y) + (o.z + o.w);
        }
    }
    if (gx_sum) {
        const float t = block_sum_fresh(s3, sm2);
        if (threadIdx.x == 0) gx_sum[c] = t;
    }
    if (threadIdx.x == 0) {
        gscale[c] = sum_gx;
        goffset[c] = sum_g;
    }
}

__global__ __launch_bounds__(kThreads) void bn_fwd_nchw_k(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ offset, float* __restrict__ y,
                                                          float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                          int N, int C, int HW, float eps, int act, float alpha) {
    __shared__ float sm[32];
    const int c = blockIdx.x;
    const int total = N * HW;
    const float inv_cnt = 1.f / (float)total;
    float s = 0.f;
    for (int i = threadIdx.x; i < total; i += kThreads) {
        int n = i / HW, p = i - n * HW;
        s += x[((size_t)n * C + c) * HW + p];
    }
    const float mean = block_sum(s, sm) * inv_cnt;
    float q = 0.f;
    for (int i = threadIdx.x; i < total; i += kThreads) {
        int n = i / HW, p = i - n * HW;
        float d = x[((size_t)n * C + c) * HW + p] - mean;
        q += d * d;
    }
    const float var = block_sum(q, sm) * inv_cnt;
    const float invstd = 1.f / sqrtf(var + eps);
    const float g = scale[c], b = offset[c];
    for (int i = threadIdx.x; i < total; i += kThreads) {
        int n = i / HW, p = i - n * HW;
        size_t idx = ((size_t)n * C + c) * HW + p;
        float v = g * ((x[idx] - mean) * invstd) + b;
        y[idx] = act_apply(v, act, alpha);
    }
    if (threadIdx.x == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
    }
}

__global__ __launch_bounds__(kThreads) void bn_bwd_nchw_k(const float* __restrict__ x, const float* __restrict__ gy, GyMask mk,
                                                          const float* __restrict__ scale, const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, float* __restrict__ gx,
                                                          float* __restrict__ gscale, float* __restrict__ goffset, float* __restrict__ gx_sum, int N,
                                                          int C, int HW) {
    __shared__ float sm[32];
    const int c = blockIdx.x;
    const int total = N * HW;
    const float mean = save_mean[c], invstd = save_invstd[c];
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < total; i += kThreads) {
        int n = i / HW, p = i - n * HW;
        size_t idx = ((size_t)n * C + c) * HW + p;
        float g = ld_gy(gy, mk, idx);
        s1 += g;
        s2 += g * ((x[idx] - mean) * invstd);
    }
    const float sum_g = block_sum(s1, sm);
    const float sum_gx = block_sum(s2, sm);
    const float inv_cnt = 1.f / (float)total;
    const float k = scale[c] * invstd;
    const float mg = sum_g * inv_cnt, mgx = sum_gx * inv_cnt;
    float s3 = 0.f;
    for (int i = threadIdx.x; i < total; i += kThreads) {
        int n = i / HW, p = i - n * HW;
        size_t idx = ((size_t)n * C + c) * HW + p;
        float xh = (x[idx] - mean) * invstd;
        const float o = k * (ld_gy(gy, mk, idx) - mg - xh * mgx);
        gx[idx] = o;
        s3 += o;
    }
    if (gx_sum) {
        const float t = block_sum(s3, sm);
        if (threadIdx.x == 0) gx_sum[c] = t;
    }
    if (threadIdx.x == 0) {
        gscale[c] = sum_gx;
        goffset[c] = sum_g;
    }
}

// ---- [N, C] layout: block = 64 columns x 4 row-slices -----------------------------------------
constexpr int kCols = 32, kSlices = 16;   // 128-byte row segments; 16 row slices keep each thread at N/16 dependent loads

__device__ __forceinline__ float slice_sum(float v, float (*sm)[kCols]) {
    const int col = threadIdx.x, sl = threadIdx.y;
    __syncthreads();
    sm[sl][col] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < kSlices; ++i) t += sm[i][col];
    return t;
}

__global__ void bn_fwd_rows_k(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
                              float* __restrict__ y, float* __restrict__ save_mean, float* __restrict__ save_invstd, int N,
                              int C, float eps, int act, float alpha) {
    __shared__ float sm[kSlices][kCols];
    const int c = blockIdx.x * kCols + threadIdx.x;
    const bool ok = c < C;
    const float inv_cnt = 1.f / (float)N;
    float s = 0.f;
    if (ok) for (int n = threadIdx.y; n < N; n += kSlices) s += x[(size_t)n * C + c];
    const float mean = slice_sum(s, sm) * inv_cnt;
    float q = 0.f;
    if (ok) for (int n = threadIdx.y; n < N; n += kSlices) {
        float d = x[(size_t)n * C + c] - mean;
        q += d * d;
    }
    const float var = slice_sum(q, sm) * inv_cnt;
    const float invstd = 1.f / sqrtf(var + eps);
    if (!ok) return;
    const float g = scale[c], b = offset[c];
    for (int n = threadIdx.y; n < N; n += kSlices) {
        size_t idx = (size_t)n * C + c;
        y[idx] = act_apply(g * ((x[idx] - mean) * invstd) + b, act, alpha);
    }
    if (threadIdx.y == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
    }
}

__global__ void bn_bwd_rows_k(const float* __restrict__ x, const float* __restrict__ gy, GyMask mk, const float* __restrict__ scale,
                              const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                              float* __restrict__ gx, float* __restrict__ gscale, float* __restrict__ goffset, int N, int C) {
    __shared__ float sm[kSlices][kCols];
    const int c = blockIdx.x * kCols + threadIdx.x;
    const bool ok = c < C;
    const float mean = ok ? save_mean[c] : 0.f, invstd = ok ? save_invstd[c] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if (ok) for (int n = threadIdx.y; n < N; n += kSlices) {
        size_t idx = (size_t)n * C + c;
        float g = ld_gy(gy, mk, idx);
        s1 += g;
        s2 += g * ((x[idx] - mean) * invstd);
    }
    const float sum_g = slice_sum(s1, sm);
    const float sum_gx = slice_sum(s2, sm);
    if (!ok) return;
    const float inv_cnt = 1.f / (float)N;
    const float k = scale[c] * invstd, mg = sum_g * inv_cnt, mgx = sum_gx * inv_cnt;
    for (int n = threadIdx.y; n < N; n += kSlices) {
        size_t idx = (size_t)n * C + c;
        float xh = (x[idx] - mean) * invstd;
        gx[idx] = k * (ld_gy(gy, mk, idx) - mg - xh * mgx);
    }
    if (threadIdx.y == 0) {
        gscale[c] = sum_gx;
        goffset[c] = sum_g;
    }
}

// ---- cross-replica ("sync") batch norm, SURVEY.md 8(e): the statistics and the normalisation are separate launches with
//      an all-gather of 2*C floats per replica between them (issued by the host through torch.distributed).  Every
//      replica merges the gathered rows in rank order, so all of them hold bit-identical statistics.  One workgroup per
//      channel for both layouts (element i of channel c sits at ((i / HW) * C + c) * HW + i % HW): this is the parity
//      mode for "1 GPU x B == N GPUs x B/N", not the throughput path. ------------------------------------------------
constexpr int kSyncThreads = 256;

__device__ __forceinline__ size_t chan_idx(int i, int c, int C, int HW) {
    const int n = i / HW, p = i - n * HW;
    return ((size_t)n * C + c) * HW + p;
}

// local (mean, M2 = sum of squared deviations from the local mean) per channel -> out[0][c], out[1][c]
__global__ __launch_bounds__(kSyncThreads) void bn_stats_k(const float* __restrict__ x, float* __restrict__ out, int N, int C, int HW) {
    __shared__ float sm[32];
    const int c = blockIdx.x, total = N * HW;
    float s = 0.f;
    for (int i = threadIdx.x; i < total; i += kSyncThreads) s += x[chan_idx(i, c, C, HW)];
    const float mean = block_sum(s, sm) / (float)total;
    float q = 0.f;
    for (int i = threadIdx.x; i < total; i += kSyncThreads) {
        const float d = x[chan_idx(i, c, C, HW)] - mean;
        q += d * d;
    }
    const float m2 = block_sum(q, sm);
    if (threadIdx.x == 0) {
        out[c] = mean;
        out[C + c] = m2;
    }
}

// stats [W][2][C] (equal element counts on every replica): Chan's pairwise merge collapses to
//   mean = avg(mean_r),  M2 = sum_r M2_r + n_local * sum_r (mean_r - mean)^2
__global__ __launch_bounds__(kSyncThreads) void bn_apply_sync_k(const float* __restrict__ x, const float* __restrict__ stats, int W,
                                                                const float* __restrict__ scale, const float* __restrict__ offset,
                                                                float* __restrict__ y, float* __restrict__ save_mean,
                                                                float* __restrict__ save_invstd, int N, int C, int HW, float eps,
                                                                int act, float alpha) {
    const int c = blockIdx.x, total = N * HW;
    float mean = 0.f;
    for (int r = 0; r < W; ++r) mean += stats[(size_t)r * 2 * C + c];
    mean /= (float)W;
    float m2 = 0.f, dev = 0.f;
    for (int r = 0; r < W; ++r) {
        const float d = stats[(size_t)r * 2 * C + c] - mean;
        m2 += stats[(size_t)r * 2 * C + C + c];
        dev += d * d;
    }
    const float var = (m2 + (float)total * dev) / ((float)total * (float)W);
    const float invstd = 1.f / sqrtf(var + eps);
    const float g = scale[c], b = offset[c];
    for (int i = threadIdx.x; i < total; i += kSyncThreads) {
        const size_t idx = chan_idx(i, c, C, HW);
        y[idx] = act_apply(g * ((x[idx] - mean) * invstd) + b, act, alpha);
    }
    if (threadIdx.x == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
    }
}

// local (sum g, sum g*xhat) per channel -> out[0][c], out[1][c]
__global__ __launch_bounds__(kSyncThreads) void bn_bwd_stats_k(const float* __restrict__ x, const float* __restrict__ gy, GyMask mk,
                                                               const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                               float* __restrict__ out, int N, int C, int HW) {
    __shared__ float sm[32];
    const int c = blockIdx.x, total = N * HW;
    const float mean = save_mean[c], invstd = save_invstd[c];
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < total; i += kSyncThreads) {
        const size_t idx = chan_idx(i, c, C, HW);
        const float g = ld_gy(gy, mk, idx);
        s1 += g;
        s2 += g * ((x[idx] - mean) * invstd);
    }
    const float a = block_sum(s1, sm), b = block_sum(s2, sm);
    if (threadIdx.x == 0) {
        out[c] = a;
        out[C + c] = b;
    }
}

// gx with the means over the GLOBAL batch (W * N * HW elements per channel); the parameter gradients stay this replica's own
// sums -- the gradient exchange averages them like every other parameter gradient.
__global__ __launch_bounds__(kSyncThreads) void bn_bwd_apply_sync_k(const float* __restrict__ x, const float* __restrict__ gy, GyMask mk,
                                                                    const float* __restrict__ scale, const float* __restrict__ save_mean,
                                                                    const float* __restrict__ save_invstd, const float* __restrict__ sums,
                                                                    int W, int rank, float* __restrict__ gx, float* __restrict__ gscale,
                                                                    float* __restrict__ goffset, int N, int C, int HW) {
    const int c = blockIdx.x, total = N * HW;
    float tg = 0.f, tgx = 0.f;
    for (int r = 0; r < W; ++r) {
        tg += sums[(size_t)r * 2 * C + c];
        tgx += sums[(size_t)r * 2 * C + C + c];
    }
    const float inv_cnt = 1.f / ((float)total * (float)W);
    const float mean = save_mean[c], invstd = save_invstd[c];
    const float k = scale[c] * invstd, mg = tg * inv_cnt, mgx = tgx * inv_cnt;
    for (int i = threadIdx.x; i < total; i += kSyncThreads) {
        const size_t idx = chan_idx(i, c, C, HW);
        const float xh = (x[idx] - mean) * invstd;
        gx[idx] = k * (ld_gy(gy, mk, idx) - mg - xh * mgx);
    }
    if (threadIdx.x == 0) {
        goffset[c] = sums[(size_t)rank * 2 * C + c];
        gscale[c] = sums[(size_t)rank * 2 * C + C + c];
    }
}

// ---- second derivative of training-mode BatchNorm (the gradient penalty of MODE vegan-wgan-gp differentiates the latent critic,
//      BatchNorm included, twice).  With xh = (x - mean) * invstd, P(v) = v - mean(v) - xh * mean(v * xh), a = P(gy):
//        first backward        gx  = scale * invstd * a
//        given h = dL/d(gx):   ggy     = scale * invstd * P(h)                                    (P is self-adjoint)
//                              gx2     = -scale * invstd^2 * ( mean(h*a) * xh + mean(gy*xh) * P(h) + mean(h*xh) * a )
//                              gscale2 = invstd * sum(h * a)
//      (the whole dependence on x, through mean and invstd too).  One workgroup per channel, three passes over the channel. ----
__global__ __launch_bounds__(kSyncThreads) void bn_bwd_bwd_k(const float* __restrict__ x, const float* __restrict__ gy, GyMask mk,
                                                             const float* __restrict__ h, const float* __restrict__ scale,
                                                             const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                             float* __restrict__ ggy, float* __restrict__ gx2, float* __restrict__ gscale2,
                                                             int N, int C, int HW) {
    __shared__ float sm[32];
    const int c = blockIdx.x, total = N * HW;
    const float mean = save_mean[c], invstd = save_invstd[c], g = scale[c];
    const float inv_cnt = 1.f / (float)total;
    float s_g = 0.f, s_gx = 0.f, s_h = 0.f, s_hx = 0.f;
    for (int i = threadIdx.x; i < total; i += kSyncThreads) {
        const size_t idx = chan_idx(i, c, C, HW);
        const float xh = (x[idx] - mean) * invstd, gv = ld_gy(gy, mk, idx), hv = h[idx];
        s_g += gv; s_gx += gv * xh; s_h += hv; s_hx += hv * xh;
    }
    const float m_g = block_sum(s_g, sm) * inv_cnt, m_gx = block_sum(s_gx, sm) * inv_cnt;
    const float m_h = block_sum(s_h, sm) * inv_cnt, m_hx = block_sum(s_hx, sm) * inv_cnt;
    float s_ha = 0.f;
    for (int i = threadIdx.x; i < total; i += kSyncThreads) {
        const size_t idx = chan_idx(i, c, C, HW);
        const float xh = (x[idx] - mean) * invstd;
        const float a = ld_gy(gy, mk, idx) - m_g - xh * m_gx;
        s_ha += h[idx] * a;
    }
    const float sum_ha = block_sum(s_ha, sm), m_ha = sum_ha * inv_cnt;
    const float k1 = g * invstd, k2 = -g * invstd * invstd;
    for (int i = threadIdx.x; i < total; i += kSyncThreads) {
        const size_t idx = chan_idx(i, c, C, HW);
        const float xh = (x[idx] - mean) * invstd;
        const float a = ld_gy(gy, mk, idx) - m_g - xh * m_gx;
        const float ph = h[idx] - m_h - xh * m_hx;
        float o = k1 * ph;
        if (mk.act) o = act_grad(o, mk.ref[idx], mk.act, mk.alpha);      // d(gy * act'(y)) / d(gy); act' is piecewise constant
        ggy[idx] = o;
        gx2[idx] = k2 * (m_ha * xh + m_gx * ph + m_hx * a);
    }
    if (threadIdx.x == 0) gscale2[c] = invstd * sum_ha;
}

}  // namespace

extern "C" {

int ggan_bn_fwd_train(const float* x, const float* scale, const float* offset, float* y, float* save_mean,
                      float* save_invstd, int N, int C, int HW, float eps, int act, float alpha, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && scale && offset && y && save_mean && save_invstd, "null pointer");
    GGAN_CHECK_ARG(N > 0 && C > 0 && HW > 0, "bad shape");
    hipStream_t s = (hipStream_t)stream;
    const double bytes = 12.0 * N * C * HW;
    if (HW > 1 && N * HW <= kRegE * kThreads) {
        GGAN_LAUNCH("bn_fwd_nchw", 0, 8.0 * N * C * HW, bn_fwd_nchw_reg_k, dim3(C), dim3(kThreads), 0, s, x, scale, offset, y, save_mean, save_invstd, N, C, HW, eps, act, alpha,
                    make_fastdiv((uint32_t)HW));
    } else if (HW > 1) {
        GGAN_LAUNCH("bn_fwd_nchw", 0, bytes, bn_fwd_nchw_k, dim3(C), dim3(kThreads), 0, s, x, scale, offset, y, save_mean, save_invstd, N, C, HW, eps, act, alpha);
    } else {
        GGAN_LAUNCH("bn_fwd_rows", 0, bytes, bn_fwd_rows_k, dim3(cdiv(C, kCols)), dim3(kCols, kSlices), 0, s, x, scale, offset, y, save_mean, save_invstd, N, C, eps, act, alpha);
    }
    return 0;
}

int ggan_bn_bwd_act(const float* x, const float* gy, const float* y, int y_act, float y_alpha, const float* scale,
                    const float* save_mean, const float* save_invstd, float* gx, float* gscale, float* goffset,
                    float* gx_chansum, int N, int C, int HW, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && gy && scale && save_mean && save_invstd && gx && gscale && goffset, "null pointer");
    GGAN_CHECK_ARG(y || y_act == GGAN_ACT_NONE, "activation mask needs the forward output");
    GGAN_CHECK_ARG(N > 0 && C > 0 && HW > 0, "bad shape");
    hipStream_t s = (hipStream_t)stream;
    const GyMask mk{y_act != GGAN_ACT_NONE ? y : nullptr, y_act, y_alpha};
    const double bytes = 20.0 * N * C * HW;
    const bool v4 = (HW & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)gx) | ((uintptr_t)mk.ref)) & 15) == 0 && !getenv("GGAN_BN_SCALAR");
    if (HW > 1 && v4 && N * HW <= 16 * 256) {
        GGAN_LAUNCH("bn_bwd_nchw", 0, 12.0 * N * C * HW, bn_bwd_nchw_reg4_k<256>, dim3(C), dim3(256), 0, s, x, gy, mk, scale, save_mean, save_invstd, gx, gscale, goffset, gx_chansum, N, C, HW,
                    make_fastdiv((uint32_t)HW));
    } else if (HW > 1 && v4 && N * HW <= 16 * 1024) {
        GGAN_LAUNCH("bn_bwd_nchw", 0, 12.0 * N * C * HW, bn_bwd_nchw_reg4_k<1024>, dim3(C), dim3(1024), 0, s, x, gy, mk, scale, save_mean, save_invstd, gx, gscale, goffset, gx_chansum, N, C, HW,
                    make_fastdiv((uint32_t)HW));
    } else if (HW > 1 && N * HW <= kRegE * kThreads) {
        GGAN_LAUNCH("bn_bwd_nchw", 0, 12.0 * N * C * HW, bn_bwd_nchw_reg_k, dim3(C), dim3(kThreads), 0, s, x, gy, mk, scale, save_mean, save_invstd, gx, gscale, goffset, gx_chansum, N, C, HW,
                    make_fastdiv((uint32_t)HW));
    } else if (HW > 1) {
        GGAN_LAUNCH("bn_bwd_nchw", 0, bytes, bn_bwd_nchw_k, dim3(C), dim3(kThreads), 0, s, x, gy, mk, scale, save_mean, save_invstd, gx, gscale, goffset, gx_chansum, N, C, HW);
    } else {
        GGAN_CHECK_ARG(!gx_chansum, "gx_chansum is only produced for NCHW inputs (HW > 1)");
        GGAN_LAUNCH("bn_bwd_rows", 0, bytes, bn_bwd_rows_k, dim3(cdiv(C, kCols)), dim3(kCols, kSlices), 0, s, x, gy, mk, scale, save_mean, save_invstd, gx, gscale, goffset, N, C);
    }
    return 0;
}

int ggan_bn_bwd(const float* x, const float* gy, const float* scale, const float* save_mean, const float* save_invstd,
                float* gx, float* gscale, float* goffset, int N, int C, int HW, ggan_stream_t stream) {
    return ggan_bn_bwd_act(x, gy, nullptr, GGAN_ACT_NONE, 0.f, scale, save_mean, save_invstd, gx, gscale, goffset, nullptr, N, C,
                           HW, stream);
}

int ggan_bn_sync_stats(const float* x, float* stats, int N, int C, int HW, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && stats, "null pointer");
    GGAN_CHECK_ARG(N > 0 && C > 0 && HW > 0, "bad shape");
    GGAN_LAUNCH("bn_sync_stats", 0, 8.0 * N * C * HW, bn_stats_k, dim3(C), dim3(kSyncThreads), 0, (hipStream_t)stream, x, stats, N, C, HW);
    return 0;
}

int ggan_bn_sync_apply(const float* x, const float* stats, int world, const float* scale, const float* offset, float* y,
                       float* save_mean, float* save_invstd, int N, int C, int HW, float eps, int act, float alpha,
                       ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && stats && scale && offset && y && save_mean && save_invstd, "null pointer");
    GGAN_CHECK_ARG(N > 0 && C > 0 && HW > 0 && world > 0, "bad shape");
    GGAN_LAUNCH("bn_sync_apply", 0, 8.0 * N * C * HW, bn_apply_sync_k, dim3(C), dim3(kSyncThreads), 0, (hipStream_t)stream, x, stats,
                world, scale, offset, y, save_mean, save_invstd, N, C, HW, eps, act, alpha);
    return 0;
}

int ggan_bn_sync_bwd_stats(const float* x, const float* gy, const float* y, int y_act, float y_alpha, const float* save_mean,
                           const float* save_invstd, float* sums, int N, int C, int HW, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && gy && save_mean && save_invstd && sums, "null pointer");
    GGAN_CHECK_ARG(y || y_act == GGAN_ACT_NONE, "activation mask needs the forward output");
    GGAN_CHECK_ARG(N > 0 && C > 0 && HW > 0, "bad shape");
    const GyMask mk{y_act != GGAN_ACT_NONE ? y : nullptr, y_act, y_alpha};
    GGAN_LAUNCH("bn_sync_bwd_stats", 0, 8.0 * N * C * HW, bn_bwd_stats_k, dim3(C), dim3(kSyncThreads), 0, (hipStream_t)stream, x, gy,
                mk, save_mean, save_invstd, sums, N, C, HW);
    return 0;
}

int ggan_bn_sync_bwd_apply(const float* x, const float* gy, const float* y, int y_act, float y_alpha, const float* scale,
                           const float* save_mean, const float* save_invstd, const float* sums, int world, int rank, float* gx,
                           float* gscale, float* goffset, int N, int C, int HW, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && gy && scale && save_mean && save_invstd && sums && gx && gscale && goffset, "null pointer");
    GGAN_CHECK_ARG(y || y_act == GGAN_ACT_NONE, "activation mask needs the forward output");
    GGAN_CHECK_ARG(N > 0 && C > 0 && HW > 0 && world > 0 && rank >= 0 && rank < world, "bad shape");
    const GyMask mk{y_act != GGAN_ACT_NONE ? y : nullptr, y_act, y_alpha};
    GGAN_LAUNCH("bn_sync_bwd_apply", 0, 12.0 * N * C * HW, bn_bwd_apply_sync_k, dim3(C), dim3(kSyncThreads), 0, (hipStream_t)stream, x,
                gy, mk, scale, save_mean, save_invstd, sums, world, rank, gx, gscale, goffset, N, C, HW);
    return 0;
}

int ggan_bn_bwd_bwd(const float* x, const float* gy, const float* y, int y_act, float y_alpha, const float* h, const float* scale,
                    const float* save_mean, const float* save_invstd, float* ggy, float* gx2, float* gscale2, int N, int C, int HW,
                    ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && gy && h && scale && save_mean && save_invstd && ggy && gx2 && gscale2, "null pointer");
    GGAN_CHECK_ARG(y || y_act == GGAN_ACT_NONE, "activation mask needs the forward output");
    GGAN_CHECK_ARG(y_act == GGAN_ACT_NONE || y_act == GGAN_ACT_LRELU || y_act == GGAN_ACT_RELU,
                   "second derivative only through piecewise-linear fused activations");
    GGAN_CHECK_ARG(N > 0 && C > 0 && HW > 0, "bad shape");
    const GyMask mk{y_act != GGAN_ACT_NONE ? y : nullptr, y_act, y_alpha};
    GGAN_LAUNCH("bn_bwd_bwd", 0, 32.0 * N * C * HW, bn_bwd_bwd_k, dim3(C), dim3(kSyncThreads), 0, (hipStream_t)stream, x, gy, mk, h, scale,
                save_mean, save_invstd, ggy, gx2, gscale2, N, C, HW);
    return 0;
}

}  // extern "C"
