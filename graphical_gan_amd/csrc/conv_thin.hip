// "Thin" ends of the conv stacks on gfx950: the 5x5 / stride-2 layers whose image side has 1..4 channels (Extractor.1 /
// Discriminator.1 and the generators' last Deconv2D: 3 channels for CIFAR / SVHN / CelebA, 1 for MNIST and moving-MNIST).
//
// The general kernels (conv_corr.hip, conv_wgrad.hip) tile channels x channels; with 3 image channels their MFMA tiles are
// 3/32 (data gradient) and 3/16 (filter gradient) full, and those layers cost as much as the 64->128 ones (29 us for
// 0.16 GFLOP).  Here the taps join the thin channel dimension instead:
//
//   data gradient / Deconv2D forward (col2im form)
//       T[p, tap*Ci + c] = sum_k gy[n, k, p] * W[tap, c, k]          one dense GEMM, N = 25*Ci (75 -> 80: 94 % full)
//       gx[n, c, y, x]   = sum_{taps hitting (y, x)} T[p(y, x, tap), tap*Ci + c]      (<= 9 terms, fixed order)
//     One workgroup = one image x a band of 2 gy rows (4 output rows): the 4 gy rows it depends on are staged in LDS
//     ([K][pixels], straight from NCHW), the filter as it lies in memory ([25*Ci][K], k contiguous, rows padded by 4 floats:
//     conflict-free MFMA B-fragment reads), v_mfma_f32_16x16x4_f32, T goes back to LDS over the gy tile, and every output
//     element gathers its terms, adds the bias, applies the activation and is stored once (coalesced rows).  No atomics, no
//     split-K: deterministic.  Halo rows are recomputed (2x the GEMM, which is 3 % of the old kernel's MFMA time).
//
// Replaces tf.nn.conv2d_transpose of Generator.5 (tflib/ops/deconv2d.py:101-114) and Conv2DBackpropInput of
// Discriminator.1 / Extractor.1 (tflib/ops/conv2d.py:106) -- SURVEY.md 8(a) a2/a3.
#include "common.h"
#include "conv.h"
#include <stdlib.h>
using namespace ggan;

namespace {

constexpr int NTHR = 256;

struct ThinDgradParams {
    const float* gy;     // [N][K][Ho][Wo]
    const float* ref;    // activation reference of gy (mask), or null
    const float* w;      // [25][Ci][K]
    const float* bias;   // [Ci] or null
    float* gx;           // [N][Ci][H][W], H = 2*Ho, W = 2*Wo
    int N, K, Ci, Ho, Wo, H, W;
    int MT;              // m-tiles of 16 pixels covering the 4 staged rows
    int PS, KP, TS;      // LDS row strides: gy tile [K][PS], filter [NT*16][KP], T [MT*16][TS]
    int mask_act, act;
    float mask_alpha, alpha;
    int dbg;
};

template <int NT>
__global__ __launch_bounds__(NTHR) void thin_dgrad_kernel(const ThinDgradParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int g0 = blockIdx.x * 2, n = blockIdx.y;
    const int K = P.K, Wo = P.Wo, J = 25 * P.Ci;
    float* Ws = smem;                          // [NT*16][KP]
    float* As = smem + NT * 16 * P.KP;         // [K][PS]; later T [MT*16][TS]

    // ---- stage the filter (rows >= 25*Ci zero) and the 4 gy rows g0-1 .. g0+2 (rows outside the image zero) ----
    if (!(P.dbg & 4)) {
        const int k4 = K >> 2;
        for (int u = tid; u < NT * 16 * k4; u += NTHR) {
            const int j = u / k4, c4 = u - j * k4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < J) v = *reinterpret_cast<const float4*>(P.w + (size_t)j * K + c4 * 4);
            *reinterpret_cast<float4*>(Ws + j * P.KP + c4 * 4) = v;
        }
        const int P4 = 4 * Wo;                 // staged pixels per channel
        const size_t img = (size_t)n * K * P.Ho * Wo;
        if ((Wo & 3) == 0) {
            const int w4 = Wo >> 2;
            for (int u = tid; u < K * 4 * w4; u += NTHR) {
                const int c4 = u % w4, t = u / w4, r = t & 3, k = t >> 2;
                const int oh = g0 - 1 + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((unsigned)oh < (unsigned)P.Ho) {
                    const size_t off = img + ((size_t)k * P.Ho + oh) * Wo + c4 * 4;
                    v = *reinterpret_cast<const float4*>(P.gy + off);
                    if (P.ref) {
                        const float4 rf = *reinterpret_cast<const float4*>(P.ref + off);
                        v.x = act_grad(v.x, rf.x, P.mask_act, P.mask_alpha);
                        v.y = act_grad(v.y, rf.y, P.mask_act, P.mask_alpha);
                        v.z = act_grad(v.z, rf.z, P.mask_act, P.mask_alpha);
                        v.w = act_grad(v.w, rf.w, P.mask_act, P.mask_alpha);
                    }
                }
                *reinterpret_cast<float4*>(As + k * P.PS + r * Wo + c4 * 4) = v;
            }
        } else {
            for (int u = tid; u < K * P4; u += NTHR) {
                const int p = u % P4, k = u / P4, r = p / Wo, col = p - r * Wo;
                const int oh = g0 - 1 + r;
                float v = 0.f;
                if ((unsigned)oh < (unsigned)P.Ho) {
                    const size_t off = img + ((size_t)k * P.Ho + oh) * Wo + col;
                    v = P.gy[off];
                    if (P.ref) v = act_grad(v, P.ref[off], P.mask_act, P.mask_alpha);
                }
                As[k * P.PS + p] = v;
            }
        }
        // pixel slots P4 .. MT*16-1 of every channel (only when 4*Wo is not a multiple of 16)
        const int tail = P.MT * 16 - P4;
        for (int u = tid; u < K * tail; u += NTHR) {
            const int k = u / tail, p = P4 + (u - k * tail);
            As[k * P.PS + p] = 0.f;
        }
    }
    __syncthreads();

    // ---- T = gy_tile^T x W^T: wave w owns m-tiles w, w+4 ----------------------------------------------------
    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool two = wave + 4 < P.MT;
    if (wave < P.MT && !(P.dbg & 1)) {
        const float* ap = As + q * P.PS + wave * 16 + l15;
        const float* bp = Ws + l15 * P.KP + q;
#pragma unroll 4
        for (int ks = 0; ks < (K >> 2); ++ks) {
            const float a0 = ap[ks * 4 * P.PS];
            const float a1 = two ? ap[ks * 4 * P.PS + 64] : 0.f;
            float b[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = bp[t * 16 * P.KP + ks * 4];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[t], acc[0][t], 0, 0, 0);
            if (two) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[t], acc[1][t], 0, 0, 0);
            }
        }
    }
    __syncthreads();                           // every wave is done reading the gy tile: T may overwrite it
    float* Ts = As;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mt = wave + 4 * i;
        if (mt < P.MT) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ts[(mt * 16 + 4 * q + r) * P.TS + t * 16 + l15] = acc[i][t][r];
        }
    }
    __syncthreads();

    // ---- gather: output rows 2*g0 .. 2*g0+3; SAME padding (1, 2): 2*oh + kh - 1 = y ----------------------------
    const int W = P.W, Ci = P.Ci;
    for (int idx = tid; idx < Ci * 4 * W; idx += NTHR) {
        const int x = idx % W, t = idx / W, yy = t & 3, c = t >> 2;
        float s = 0.f;
        for (int kh = (P.dbg & 2) ? 5 : ((yy + 1) & 1); kh < 5; kh += 2) {
            const int r = (yy + 1 - kh) / 2 + 1;               // staged row of oh = g0 + (yy+1-kh)/2
            for (int kw = (x + 1) & 1; kw < 5; kw += 2) {
                const int d = x + 1 - kw;
                if (d < 0 || (d >> 1) >= Wo) continue;
                s += Ts[(r * Wo + (d >> 1)) * P.TS + (kh * 5 + kw) * Ci + c];
            }
        }
        if (P.bias) s += P.bias[c];
        P.gx[(((size_t)n * Ci + c) * P.H + 2 * g0 + yy) * W + x] = act_apply(s, P.act, P.alpha);
    }
}

}  // namespace

namespace ggan {

int conv_dgrad_thin(const ggan_conv_geom& g, const float* gy, GyMask m, const float* w, const float* bias, float* gx, int act,
                    float alpha, hipStream_t s) {
    if (g.k != 5 || g.stride != 2 || g.pad_t != 1 || g.pad_l != 1) return 1;
    if (g.Ci > 4 || (g.Co & 3) || g.Co > 128 || g.H != 2 * g.Ho || g.W != 2 * g.Wo || (g.Ho & 1) || g.Wo > 32) return 1;
    if ((((uintptr_t)gy) & 15) || (((uintptr_t)w) & 15) || (m.act != GGAN_ACT_NONE && (((uintptr_t)m.ref) & 15))) return 1;
    if (getenv("GGAN_NO_THIN")) return 1;
    ThinDgradParams P;
    memset(&P, 0, sizeof(P));
    P.gy = gy; P.ref = m.act != GGAN_ACT_NONE ? m.ref : nullptr; P.w = w; P.bias = bias; P.gx = gx;
    P.N = g.N; P.K = g.Co; P.Ci = g.Ci; P.Ho = g.Ho; P.Wo = g.Wo; P.H = g.H; P.W = g.W;
    P.mask_act = m.act; P.mask_alpha = m.alpha; P.act = act; P.alpha = alpha;
    { const char* d = getenv("GGAN_THIN_DBG"); P.dbg = d ? atoi(d) : 0; }
    const int NT = cdiv(25 * g.Ci, 16);
    P.MT = cdiv(4 * g.Wo, 16);
    if (P.MT > 8) return 1;
    P.PS = P.MT * 16;
    while ((P.PS & 31) != 16) P.PS += 16;      // == 16 (mod 32): the four k-rows of an A fragment hit disjoint bank halves
    P.KP = g.Co + 4;                            // == 4 (mod 32) for K = 32, 64, 96, 128: conflict-free B fragments
    P.TS = NT * 16 + 4;
    const size_t tile = (size_t)g.Co * P.PS, tt = (size_t)P.MT * 16 * P.TS;
    const size_t shmem = ((size_t)NT * 16 * P.KP + (tile > tt ? tile : tt)) * sizeof(float);
    if (shmem > 64 * 1024) return 1;
    const double fl = 2.0 * g.N * g.Co * g.Ho * g.Wo * (double)g.Ci * 25.0;
    const dim3 grid(g.Ho / 2, g.N);
    switch (NT) {
        case 2: GGAN_LAUNCH("thin_dgrad_kernel", fl, 0, thin_dgrad_kernel<2>, grid, dim3(NTHR), shmem, s, P); break;
        case 4: GGAN_LAUNCH("thin_dgrad_kernel", fl, 0, thin_dgrad_kernel<4>, grid, dim3(NTHR), shmem, s, P); break;
        case 5: GGAN_LAUNCH("thin_dgrad_kernel", fl, 0, thin_dgrad_kernel<5>, grid, dim3(NTHR), shmem, s, P); break;
        case 7: GGAN_LAUNCH("thin_dgrad_kernel", fl, 0, thin_dgrad_kernel<7>, grid, dim3(NTHR), shmem, s, P); break;
        default: return 1;
    }
    return 0;
}

}  // namespace ggan
