// Conv3D of the state-space scripts' '3dcnn' sequence critic (tflib/ops/conv3d.py:6-51: tf.nn.conv3d, NDHWC, filter
// [fl, fs, fs, in, out], strides (stride_len, stride, stride), SAME padding).  The filter is stored exactly as the [K, Co] operand of a
// GEMM with K = (dl, dh, dw, ci), so the layer is  y = im2col(x) @ W + b  on the MFMA GEMM of gemm.hip (bias / activation in its
// epilogue), the filter gradient is im2col(x)^T @ gy and the data gradient col2im(gy @ W^T).  This file holds the two layout kernels;
// both are pure HBM streaming (the patch matrix of the largest layer is ~0.5-0.8 GB -- 288 GB of HBM make materialising it the cheap
// option) and each is the other's adjoint, so the pair is closed under differentiation.
#include "common.h"
using namespace ggan;

namespace {

struct C3 {
    int N, L, H, W, Ci, Co, Lo, Ho, Wo, kl, k, sl, s, pl, ph, pw;
};

template <int V> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<4> { typedef float4 T; };
template <int V> __device__ __forceinline__ typename VecT<V>::T vzero();
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }
template <> __device__ __forceinline__ float4 vzero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void vacc(float& a, float b) { a += b; }
__device__ __forceinline__ void vacc(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// col[(n,ol,oh,ow)][(dl,dh,dw,ci)] = x[n, ol*sl+dl-pl, oh*s+dh-ph, ow*s+dw-pw, ci]  (0 outside the volume).  One thread per V
// consecutive channels (V = 4: 16-byte loads / stores when Ci % 4 == 0); IT = 32-bit index arithmetic whenever the element count
// allows it (the 64-bit divisions of the general case cost more than the memory traffic).
template <typename IT, int V>
__global__ void im2col3d_k(const C3 g, const float* __restrict__ x, float* __restrict__ col) {
    typedef typename VecT<V>::T VT;
    const IT Cv = (IT)(g.Ci / V), total = (IT)g.N * g.Lo * g.Ho * g.Wo * g.kl * g.k * g.k * Cv;
    for (IT i = (IT)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IT)gridDim.x * blockDim.x) {
        IT t = i;
        const int cv = (int)(t % Cv); t /= Cv;
        const int dw = (int)(t % (IT)g.k); t /= (IT)g.k;
        const int dh = (int)(t % (IT)g.k); t /= (IT)g.k;
        const int dl = (int)(t % (IT)g.kl); t /= (IT)g.kl;
        const int ow = (int)(t % (IT)g.Wo); t /= (IT)g.Wo;
        const int oh = (int)(t % (IT)g.Ho); t /= (IT)g.Ho;
        const int ol = (int)(t % (IT)g.Lo);
        const int n = (int)(t / (IT)g.Lo);
        const int l = ol * g.sl + dl - g.pl, h = oh * g.s + dh - g.ph, w = ow * g.s + dw - g.pw;
        const bool in = (unsigned)l < (unsigned)g.L && (unsigned)h < (unsigned)g.H && (unsigned)w < (unsigned)g.W;
        VT v = vzero<V>();
        if (in) v = reinterpret_cast<const VT*>(x)[((((size_t)n * g.L + l) * g.H + h) * g.W + w) * Cv + cv];
        reinterpret_cast<VT*>(col)[i] = v;
    }
}

// the adjoint as a gather (deterministic): gx[n,l,h,w,ci] = sum over the taps (dl,dh,dw) whose window covers the voxel
template <typename IT, int V>
__global__ void col2im3d_k(const C3 g, const float* __restrict__ col, float* __restrict__ gx) {
    typedef typename VecT<V>::T VT;
    const IT Cv = (IT)(g.Ci / V), total = (IT)g.N * g.L * g.H * g.W * Cv;
    const size_t Kv = (size_t)g.kl * g.k * g.k * Cv;
    for (IT i = (IT)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (IT)gridDim.x * blockDim.x) {
        IT t = i;
        const int cv = (int)(t % Cv); t /= Cv;
        const int w = (int)(t % (IT)g.W); t /= (IT)g.W;
        const int h = (int)(t % (IT)g.H); t /= (IT)g.H;
        const int l = (int)(t % (IT)g.L);
        const int n = (int)(t / (IT)g.L);
        VT acc = vzero<V>();
        for (int dl = 0; dl < g.kl; ++dl) {
            const int a = l + g.pl - dl;
            if (a < 0 || a % g.sl) continue;
            const int ol = a / g.sl;
            if (ol >= g.Lo) continue;
            for (int dh = 0; dh < g.k; ++dh) {
                const int b = h + g.ph - dh;
                if (b < 0 || b % g.s) continue;
                const int oh = b / g.s;
                if (oh >= g.Ho) continue;
                for (int dw = 0; dw < g.k; ++dw) {
                    const int c = w + g.pw - dw;
                    if (c < 0 || c % g.s) continue;
                    const int ow = c / g.s;
                    if (ow >= g.Wo) continue;
                    vacc(acc, reinterpret_cast<const VT*>(col)[((((size_t)n * g.Lo + ol) * g.Ho + oh) * g.Wo + ow) * Kv +
                                                               (((size_t)dl * g.k + dh) * g.k + dw) * Cv + cv]);
                }
            }
        }
        reinterpret_cast<VT*>(gx)[i] = acc;
    }
}

int fill(C3& g, const int* d) {
    // d: N, L, H, W, Ci, Co, kl, k, sl, s  (SAME padding: out = ceil(in / stride), extra padding at the end)
    g.N = d[0]; g.L = d[1]; g.H = d[2]; g.W = d[3]; g.Ci = d[4]; g.Co = d[5]; g.kl = d[6]; g.k = d[7]; g.sl = d[8]; g.s = d[9];
    for (int i = 0; i < 10; ++i)
        if (d[i] <= 0) return -1;
    auto out = [](int n, int s) { return (n + s - 1) / s; };
    auto pad = [](int n, int o, int k, int s) { int t = (o - 1) * s + k - n; return t > 0 ? t / 2 : 0; };
    g.Lo = out(g.L, g.sl); g.Ho = out(g.H, g.s); g.Wo = out(g.W, g.s);
    g.pl = pad(g.L, g.Lo, g.kl, g.sl); g.ph = pad(g.H, g.Ho, g.k, g.s); g.pw = pad(g.W, g.Wo, g.k, g.s);
    return 0;
}

int blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 32768 ? 32768 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int ggan_conv3d_out_shape(const int* dims10, int* lo_ho_wo) {
    C3 g;
    GGAN_CHECK_ARG(dims10 && lo_ho_wo && fill(g, dims10) == 0, "bad geometry");
    lo_ho_wo[0] = g.Lo; lo_ho_wo[1] = g.Ho; lo_ho_wo[2] = g.Wo;
    return 0;
}

int ggan_im2col3d(const int* dims10, const float* x, float* col, ggan_stream_t stream) {
    C3 g;
    GGAN_CHECK_ARG(dims10 && x && col && fill(g, dims10) == 0, "bad argument");
    const size_t n = (size_t)g.N * g.Lo * g.Ho * g.Wo * g.kl * g.k * g.k * g.Ci;
    const bool v4 = (g.Ci & 3) == 0 && (((uintptr_t)x | (uintptr_t)col) & 15) == 0, i32 = n < 0xF0000000ull;
    hipStream_t s = (hipStream_t)stream;
    if (v4 && i32) { GGAN_LAUNCH("im2col3d", 0, 8.0 * n, (im2col3d_k<uint32_t, 4>), dim3(blocks(n / 4)), dim3(256), 0, s, g, x, col); }
    else if (v4) { GGAN_LAUNCH("im2col3d", 0, 8.0 * n, (im2col3d_k<size_t, 4>), dim3(blocks(n / 4)), dim3(256), 0, s, g, x, col); }
    else if (i32) { GGAN_LAUNCH("im2col3d", 0, 8.0 * n, (im2col3d_k<uint32_t, 1>), dim3(blocks(n)), dim3(256), 0, s, g, x, col); }
    else { GGAN_LAUNCH("im2col3d", 0, 8.0 * n, (im2col3d_k<size_t, 1>), dim3(blocks(n)), dim3(256), 0, s, g, x, col); }
    return 0;
}

int ggan_col2im3d(const int* dims10, const float* col, float* gx, ggan_stream_t stream) {
    C3 g;
    GGAN_CHECK_ARG(dims10 && col && gx && fill(g, dims10) == 0, "bad argument");
    const size_t n = (size_t)g.N * g.L * g.H * g.W * g.Ci;
    const double bytes = 4.0 * n * (1 + (g.kl / g.sl + 1) * (g.k / g.s + 1) * (g.k / g.s + 1));
    const bool v4 = (g.Ci & 3) == 0 && (((uintptr_t)gx | (uintptr_t)col) & 15) == 0, i32 = n < 0xF0000000ull;
    hipStream_t s = (hipStream_t)stream;
    if (v4 && i32) { GGAN_LAUNCH("col2im3d", 0, bytes, (col2im3d_k<uint32_t, 4>), dim3(blocks(n / 4)), dim3(256), 0, s, g, col, gx); }
    else if (v4) { GGAN_LAUNCH("col2im3d", 0, bytes, (col2im3d_k<size_t, 4>), dim3(blocks(n / 4)), dim3(256), 0, s, g, col, gx); }
    else if (i32) { GGAN_LAUNCH("col2im3d", 0, bytes, (col2im3d_k<uint32_t, 1>), dim3(blocks(n)), dim3(256), 0, s, g, col, gx); }
    else { GGAN_LAUNCH("col2im3d", 0, bytes, (col2im3d_k<size_t, 1>), dim3(blocks(n)), dim3(256), 0, s, g, col, gx); }
    return 0;
}

}  // extern "C"
