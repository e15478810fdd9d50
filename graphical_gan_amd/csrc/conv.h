// Internal convolution entry points (host side), shared between conv_naive / conv_mfma / capi.
#pragma once
#include "common.h"

namespace ggan {

int conv_fwd_naive(const ggan_conv_geom& g, const float* x, const float* w, const float* bias, float* y, int act,
                   float alpha, hipStream_t s);
// "gy mask": when m.act != GGAN_ACT_NONE the kernels consume gy[i] * act'(m.ref[i]) instead of gy[i] -- the activation
// backward of a fused conv+bias+act layer applied while the operand is staged (no separate act_bwd pass / tensor).
struct GyMask {
    const float* ref;
    int act;
    float alpha;
};
int conv_dgrad_naive(const ggan_conv_geom& g, const float* gy, GyMask m, const float* w, const float* bias, float* gx,
                     int act, float alpha, hipStream_t s);
int conv_wgrad_naive(const ggan_conv_geom& g, const float* x, const float* gy, GyMask m, float* gw, hipStream_t s);

// MFMA paths: return 1 when the geometry is not covered (caller falls back to the naive kernel),
// 0 on success, <0 on error.
int conv_fwd_mfma(const ggan_conv_geom& g, const float* x, const float* w, const float* bias, float* y, int act,
                  float alpha, void* ws, size_t ws_bytes, hipStream_t s);
int conv_dgrad_mfma(const ggan_conv_geom& g, const float* gy, GyMask m, const float* w, const float* bias, float* gx,
                    int act, float alpha, void* ws, size_t ws_bytes, hipStream_t s);
// all-class data gradient on 64-pixel x 16 / 32-channel tiles (conv_dg16.hip, round 4): same contract; target_wgs = ggan_conv_geom.plan_wgs
int conv_dgrad_dg16(const ggan_conv_geom& g, const float* gy, GyMask m, const float* w, const float* bias, float* gx, int act,
                    float alpha, int target_wgs, void* ws, size_t ws_bytes, hipStream_t s);
// image side with <= 4 channels (conv_thin.hip): same contract as the MFMA paths
int conv_dgrad_thin(const ggan_conv_geom& g, const float* gy, GyMask m, const float* w, const float* bias, float* gx, int act,
                    float alpha, hipStream_t s);
// gbias (may be NULL): sum over n,oh,ow of the (masked) gy, produced from the gy tiles the kernel stages anyway
// parts != NULL: the split-K slabs stay in parts->buf ([n][stride], bias-gradient tail after each slab) for a consumer that
// sums them (ggan_pack_parts); no reduce kernel is launched and gw/gbias/ws are ignored.
struct WgradParts {
    float* buf;
    size_t cap_floats;
    int with_bias;
    int n;            // out: number of slabs
    size_t stride;    // out: floats between slabs
};
int conv_wgrad_mfma(const ggan_conv_geom& g, const float* x, const float* gy, GyMask m, float* gw, float* gbias, void* ws,
                    size_t ws_bytes, hipStream_t s, WgradParts* parts = nullptr);
// cast (optional): the input is an int32 minibatch in a device ring, scaled while it is staged and written to cast->x_out (x is ignored)
struct ThinCastSrc {
    const int32_t* ring;     // [nslots][N*Ci*H*W]
    const int32_t* ctr_a;    // slot = (offset + *ctr_a + *ctr_b) mod nslots (either counter may be null)
    const int32_t* ctr_b;
    const float* noise;      // [N*Ci*H*W] or null
    float* x_out;            // [N][Ci][H][W]
    int nslots, offset;
    float div, mul;
};
// Output mask of a forward launch (conv_corr.hip epilogue): while *g_out_mask is set, conv_fwd_mfma stores act_grad(value, ref[i]) instead of
// the value -- or returns 1 before launching anything when the planned launch cannot (split-K: the epilogue does not see final values).
struct OutMask {
    const float* ref;
    int act;
    float alpha;
    bool applied;
};
extern thread_local OutMask* g_out_mask;
// mask (optional): the thin-channel forward stores act_grad(value, mask->ref[i]) (the masked forward of a first layer)
int conv_fwd_thin(const ggan_conv_geom& g, const float* x, const float* w, const float* bias, float* y, int act, float alpha,
                  hipStream_t s, const ThinCastSrc* cast = nullptr, const OutMask* mask = nullptr);
int conv_wgrad_thin(const ggan_conv_geom& g, const float* x, const float* gy, GyMask m, float* gw, float* gbias, void* ws,
                    size_t ws_bytes, hipStream_t s, WgradParts* parts = nullptr);
size_t conv_workspace_bytes(const ggan_conv_geom& g);
// out[i] = act(sum_s partial[s][i] + bias[(i/HW)%C]) -- deterministic split-K combine (conv + gemm)
int launch_splitk_reduce(const float* partial, int SK, size_t elems, float* out, const float* bias, int C, int HW, int act,
                         float alpha, hipStream_t s, size_t slab_stride = 0, float* tail_out = nullptr, size_t tail = 0);

// exact n / d for n*d < 2^32 via one mulhi
struct FastDiv {
    uint32_t mul, d;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    f.mul = (d <= 1) ? 0u : (uint32_t)((0x100000000ull / d) + 1ull);
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) { return f.d <= 1 ? n : __umulhi(n, f.mul); }

// Division of SMALL numerators (n <= nmax, a few thousand: element / pixel indices inside one tile) by a run-time divisor as one
// full-rate 24-bit multiply and a shift: v_mul_hi_u32 is a quarter-rate instruction (16 cycles per wave), and the per-thread
// staging descriptors of the conv kernels are a few hundred such instructions executed once per launch.
struct FastDiv24 {
    uint32_t mul, shift;
};
inline bool make_fastdiv24(uint32_t d, uint32_t nmax, FastDiv24* out) {
    if (d <= 1) { out->mul = 1; out->shift = 0; return true; }
    for (int s = 31; s >= 1; --s) {
        const uint64_t m = ((1ull << s) + d - 1) / d;
        if (m >= (1ull << 24) || m * nmax >= (1ull << 32)) continue;
        const uint64_t err = m * d - (1ull << s);               // q(n) = floor(n/d) for every n with n * err < 2^s
        if ((uint64_t)nmax * err >= (1ull << s)) return false;
        out->mul = (uint32_t)m; out->shift = (uint32_t)s;
        return true;
    }
    return false;
}
__device__ __forceinline__ uint32_t fdiv24(uint32_t n, FastDiv24 f) { return __umul24(n, f.mul) >> f.shift; }

}  // namespace ggan
