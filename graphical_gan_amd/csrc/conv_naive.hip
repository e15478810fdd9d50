// Plain one-thread-per-output HIP convolution kernels for any (k, stride, padding).
// They serve geometries the MFMA kernels do not cover (e.g. stride-1 or VALID convolutions used
// outside the hot path, odd MNIST widths in the filter-gradient) and as an on-device cross-check
// (ggan_conv_geom.plan_flags & GGAN_PLAN_PLAIN).  Same arithmetic definition as SURVEY.md A.1 / A.2.
#include "common.h"
#include "conv.h"
using namespace ggan;

namespace {

__global__ void conv_fwd_naive_k(ggan_conv_geom g, const float* __restrict__ x, const float* __restrict__ w,
                                 const float* __restrict__ bias, float* __restrict__ y, int act, float alpha) {
    const size_t total = (size_t)g.N * g.Co * g.Ho * g.Wo;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int ow = idx % g.Wo;
        size_t t = idx / g.Wo;
        int oh = t % g.Ho; t /= g.Ho;
        int co = t % g.Co;
        int n = t / g.Co;
        float acc = 0.f;
        for (int ci = 0; ci < g.Ci; ++ci) {
            const float* xp = x + ((size_t)n * g.Ci + ci) * g.H * g.W;
            for (int kh = 0; kh < g.k; ++kh) {
                int ih = oh * g.stride + kh - g.pad_t;
                if (ih < 0 || ih >= g.H) continue;
                for (int kw = 0; kw < g.k; ++kw) {
                    int iw = ow * g.stride + kw - g.pad_l;
                    if (iw < 0 || iw >= g.W) continue;
                    acc = fmaf(xp[ih * g.W + iw], w[(((size_t)kh * g.k + kw) * g.Ci + ci) * g.Co + co], acc);
                }
            }
        }
        if (bias) acc += bias[co];
        y[idx] = act_apply(acc, act, alpha);
    }
}

__global__ void conv_dgrad_naive_k(ggan_conv_geom g, const float* __restrict__ gy, GyMask mk, const float* __restrict__ w,
                                   const float* __restrict__ bias, float* __restrict__ gx, int act, float alpha) {
    const size_t total = (size_t)g.N * g.Ci * g.H * g.W;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int iw = idx % g.W;
        size_t t = idx / g.W;
        int ih = t % g.H; t /= g.H;
        int ci = t % g.Ci;
        int n = t / g.Ci;
        float acc = 0.f;
        for (int kh = 0; kh < g.k; ++kh) {
            int th = ih + g.pad_t - kh;
            if (th < 0 || th % g.stride) continue;
            int oh = th / g.stride;
            if (oh >= g.Ho) continue;
            for (int kw = 0; kw < g.k; ++kw) {
                int tw = iw + g.pad_l - kw;
                if (tw < 0 || tw % g.stride) continue;
                int ow = tw / g.stride;
                if (ow >= g.Wo) continue;
                const float* wp = w + (((size_t)kh * g.k + kw) * g.Ci + ci) * g.Co;
                const size_t gb = (size_t)n * g.Co * g.Ho * g.Wo + (size_t)oh * g.Wo + ow;
                for (int co = 0; co < g.Co; ++co) {
                    const size_t gi = gb + (size_t)co * g.Ho * g.Wo;
                    float gv = gy[gi];
                    if (mk.act) gv = act_grad(gv, mk.ref[gi], mk.act, mk.alpha);
                    acc = fmaf(gv, wp[co], acc);
                }
            }
        }
        if (bias) acc += bias[ci];
        gx[idx] = act_apply(acc, act, alpha);
    }
}

__global__ void conv_wgrad_naive_k(ggan_conv_geom g, const float* __restrict__ x, const float* __restrict__ gy, GyMask mk,
                                   float* __restrict__ gw) {
    const size_t total = (size_t)g.k * g.k * g.Ci * g.Co;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int co = idx % g.Co;
        size_t t = idx / g.Co;
        int ci = t % g.Ci; t /= g.Ci;
        int kw = t % g.k;
        int kh = t / g.k;
        float acc = 0.f;
        for (int n = 0; n < g.N; ++n) {
            const float* xp = x + ((size_t)n * g.Ci + ci) * g.H * g.W;
            const size_t gb = ((size_t)n * g.Co + co) * g.Ho * g.Wo;
            for (int oh = 0; oh < g.Ho; ++oh) {
                int ih = oh * g.stride + kh - g.pad_t;
                if (ih < 0 || ih >= g.H) continue;
                for (int ow = 0; ow < g.Wo; ++ow) {
                    int iw = ow * g.stride + kw - g.pad_l;
                    if (iw < 0 || iw >= g.W) continue;
                    float gv = gy[gb + oh * g.Wo + ow];
                    if (mk.act) gv = act_grad(gv, mk.ref[gb + oh * g.Wo + ow], mk.act, mk.alpha);
                    acc = fmaf(xp[ih * g.W + iw], gv, acc);
                }
            }
        }
        gw[idx] = acc;
    }
}

inline int grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    if (b > 65535) b = 65535;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

namespace ggan {

int conv_fwd_naive(const ggan_conv_geom& g, const float* x, const float* w, const float* bias, float* y, int act,
                   float alpha, hipStream_t s) {
    size_t total = (size_t)g.N * g.Co * g.Ho * g.Wo;
    double fl = 2.0 * total * g.Ci * g.k * g.k;
    GGAN_LAUNCH("conv_fwd_naive", fl, 0, conv_fwd_naive_k, dim3(grid_for(total)), dim3(256), 0, s, g, x, w, bias, y, act, alpha);
    return 0;
}

int conv_dgrad_naive(const ggan_conv_geom& g, const float* gy, GyMask m, const float* w, const float* bias, float* gx,
                     int act, float alpha, hipStream_t s) {
    size_t total = (size_t)g.N * g.Ci * g.H * g.W;
    double fl = 2.0 * g.N * g.Co * g.Ho * g.Wo * (double)g.Ci * g.k * g.k;
    GGAN_LAUNCH("conv_dgrad_naive", fl, 0, conv_dgrad_naive_k, dim3(grid_for(total)), dim3(256), 0, s, g, gy, m, w, bias, gx, act, alpha);
    return 0;
}

int conv_wgrad_naive(const ggan_conv_geom& g, const float* x, const float* gy, GyMask m, float* gw, hipStream_t s) {
    size_t total = (size_t)g.k * g.k * g.Ci * g.Co;
    double fl = 2.0 * g.N * g.Co * g.Ho * g.Wo * (double)g.Ci * g.k * g.k;
    GGAN_LAUNCH("conv_wgrad_naive", fl, 0, conv_wgrad_naive_k, dim3(grid_for(total)), dim3(256), 0, s, g, x, gy, m, gw);
    return 0;
}

}  // namespace ggan
