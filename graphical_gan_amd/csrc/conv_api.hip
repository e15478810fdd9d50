// C-ABI entry points of the Conv2D / Deconv2D family: argument checks + dispatch between the MFMA kernels
// (5x5 stride-2 hot path) and the plain kernels (everything else, or GGAN_PLAN_PLAIN in the call's geometry).
#include "common.h"
#include "conv.h"
#include <stdlib.h>
using namespace ggan;

namespace {

int check_geom(const ggan_conv_geom* g) {
    if (!g) { set_error("null geometry"); return -1; }
    if (g->N <= 0 || g->Ci <= 0 || g->Co <= 0 || g->H <= 0 || g->W <= 0 || g->Ho <= 0 || g->Wo <= 0 || g->k <= 0 ||
        g->stride <= 0 || g->pad_t < 0 || g->pad_l < 0) {
        set_error("bad conv geometry N=%d Ci=%d H=%d W=%d Co=%d Ho=%d Wo=%d k=%d s=%d pad=(%d,%d)", g->N, g->Ci, g->H,
                  g->W, g->Co, g->Ho, g->Wo, g->k, g->stride, g->pad_t, g->pad_l);
        return -1;
    }
    // the launch plan travels in the struct (ABI 500): a caller that hands over the shorter pre-plan struct leaves garbage here
    if ((g->plan_flags & ~GGAN_PLAN_PLAIN) || g->plan_wgs < 0 || g->plan_wgs > 65536 || g->plan_wgs_filter < 0 || g->plan_wgs_filter > 65536) {
        set_error("conv geometry: launch plan fields out of range (plan_wgs=%d plan_wgs_filter=%d plan_flags=%d): struct built against "
                  "an older ggan.h? (GGAN_ABI_VERSION %d)", g->plan_wgs, g->plan_wgs_filter, g->plan_flags, GGAN_ABI_VERSION);
        return -1;
    }
    // the last window must start inside the padded input
    if ((g->Ho - 1) * g->stride - g->pad_t >= g->H || (g->Wo - 1) * g->stride - g->pad_l >= g->W) {
        set_error("conv geometry: output grid larger than the input allows");
        return -1;
    }
    return 0;
}

}  // namespace

extern "C" {

size_t ggan_conv2d_workspace(const ggan_conv_geom* g) { return g ? conv_workspace_bytes(*g) : 0; }

int ggan_conv2d_fwd(const ggan_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act,
                    float alpha, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    if (check_geom(g)) return -1;
    GGAN_CHECK_ARG(x && w && y, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (!(g->plan_flags & GGAN_PLAN_PLAIN) && !getenv("GGAN_NAIVE_FWD")) {
        int r = conv_fwd_thin(*g, x, w, bias, y, act, alpha, s);
        if (r <= 0) return r;
        r = conv_fwd_mfma(*g, x, w, bias, y, act, alpha, ws, ws ? ws_bytes : 0, s);
        if (r <= 0) return r;
    }
    return conv_fwd_naive(*g, x, w, bias, y, act, alpha, s);
}

// Conv2D on a minibatch that still sits in the device ring as int32: real_x = mul*(float(v)/div - .5) (+ noise) is formed while the
// first layer stages its input and written to x_out for the other readers (the critic's [fake; real] input, the filter gradient) --
// ggan_cast_scale_ring_i32 + ggan_conv2d_fwd in one launch.  Returns 1 (nothing launched) where the thin-channel forward kernel does
// not cover the geometry: the caller issues the two calls.
int ggan_conv2d_fwd_cast_ring(const ggan_conv_geom* g, const int32_t* ring, int nslots, const int32_t* ctr_a, const int32_t* ctr_b, int offset,
                              const float* noise, float div, float mul, float* x_out, const float* w, const float* bias, float* y, int act,
                              float alpha, ggan_stream_t stream) {
    if (check_geom(g)) return -1;
    GGAN_CHECK_ARG(ring && x_out && w && y && nslots > 0, "bad argument");
    if ((g->plan_flags & GGAN_PLAN_PLAIN) || getenv("GGAN_NAIVE_FWD") || getenv("GGAN_NO_CAST_FUSION")) return 1;
    ThinCastSrc c;
    c.ring = ring; c.ctr_a = ctr_a; c.ctr_b = ctr_b; c.noise = noise; c.x_out = x_out; c.nslots = nslots; c.offset = offset; c.div = div; c.mul = mul;
    return conv_fwd_thin(*g, nullptr, w, bias, y, act, alpha, (hipStream_t)stream, &c);
}

static int bwd_data(const ggan_conv_geom* g, const float* gy, GyMask m, const float* w, const float* bias, float* gx, int act,
                    float alpha, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    if (check_geom(g)) return -1;
    if (!(gy && w && gx)) { set_error("ggan_conv2d_bwd_data: null pointer"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (getenv("GGAN_TRACE_CONV"))
        fprintf(stderr, "[ggan] bwd_data N=%d Ci=%d H=%d W=%d Co=%d Ho=%d Wo=%d pad=(%d,%d) mask=%d act=%d\n", g->N, g->Ci, g->H, g->W, g->Co, g->Ho,
                g->Wo, g->pad_t, g->pad_l, m.act, act);
    if (!(g->plan_flags & GGAN_PLAN_PLAIN) && !getenv("GGAN_NAIVE_DGRAD")) {
        int r = conv_dgrad_thin(*g, gy, m, w, bias, gx, act, alpha, s);
        if (r <= 0) return r;
        r = conv_dgrad_dg16(*g, gy, m, w, bias, gx, act, alpha, g->plan_wgs, ws, ws ? ws_bytes : 0, s);
        if (r <= 0) return r;
        r = conv_dgrad_mfma(*g, gy, m, w, bias, gx, act, alpha, ws, ws ? ws_bytes : 0, s);
        if (r <= 0) return r;
    }
    return conv_dgrad_naive(*g, gy, m, w, bias, gx, act, alpha, s);
}

// out[row][w'] = in[row][w' - shift] (0 outside [0, W)), optionally times the activation derivative of ref at the same place
__global__ void pad_width_k(const float* __restrict__ in, const float* __restrict__ ref, float* __restrict__ out, size_t rows,
                            int W, int Wp, int shift, int act, float alpha) {
    const size_t total = rows * (size_t)Wp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / Wp;
        const int w = (int)(i - r * Wp) - shift;
        float v = 0.f;
        if (w >= 0 && w < W) {
            v = in[r * W + w];
            if (ref) v = act_grad(v, ref[r * W + w], act, alpha);
        }
        out[i] = v;
    }
}

// Filter gradient of a 5x5 / stride-2 conv whose width the MFMA kernel does not take (MNIST: 28, 14, 7 wide; SAME pad_l = 2
// at 7): the same sum over zero-extended copies.  x gets `pad_l - 1` zero columns in front and zeros behind up to W' = 2*Wo',
// gy zeros behind up to Wo' (a multiple of 4): every added product has a zero factor, and SAME padding of W' -> Wo' is
// (1, 2) again, so the widened problem IS the original one.  Two small copy launches instead of the one-thread-per-output
// fallback (which made the MNIST step 10x slower than the CIFAR one).
static int wgrad_widened(const ggan_conv_geom& g, const float* x, const float* gy, GyMask m, float* gw, float* gbias, void* ws,
                         size_t ws_bytes, hipStream_t s) {
    if (g.k != 5 || g.stride != 2 || (g.pad_l != 1 && g.pad_l != 2)) return 1;
    const int shift = g.pad_l - 1;
    int Wop = (g.Wo + 3) & ~3;
    while (2 * Wop < g.W + shift) Wop += 4;
    const int Wp = 2 * Wop;
    if (Wop > 64) return 1;
    ws = ws_scratch(ws, ws_bytes);
    const size_t xr = (size_t)g.N * g.Ci * g.H, gr = (size_t)g.N * g.Co * g.Ho;
    const size_t xb = (xr * Wp * sizeof(float) + 255) & ~(size_t)255, gb = (gr * Wop * sizeof(float) + 255) & ~(size_t)255;
    if (!ws || ws_bytes < xb + gb + (1u << 20)) return 1;
    float* xp = (float*)ws;
    float* gp = (float*)((char*)ws + xb);
    auto blocks = [](size_t n) { size_t b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); };
    GGAN_LAUNCH("pad_width_k", 0, 8.0 * xr * Wp, pad_width_k, dim3(blocks(xr * Wp)), dim3(256), 0, s, x, (const float*)nullptr, xp, xr,
                g.W, Wp, shift, 0, 0.f);
    GGAN_LAUNCH("pad_width_k", 0, 8.0 * gr * Wop, pad_width_k, dim3(blocks(gr * Wop)), dim3(256), 0, s, gy,
                m.act != GGAN_ACT_NONE ? m.ref : (const float*)nullptr, gp, gr, g.Wo, Wop, 0, m.act, m.alpha);
    ggan_conv_geom gp2 = g;
    gp2.W = Wp; gp2.Wo = Wop; gp2.pad_l = 1;
    void* rest = (char*)ws + xb + gb;
    const size_t rest_bytes = ws_bytes - xb - gb;
    const GyMask none{nullptr, GGAN_ACT_NONE, 0.f};
    int r = conv_wgrad_mfma(gp2, xp, gp, none, gw, gbias, rest, rest_bytes, s);
    if (r == 1 && gbias) {     // tile shape without the in-kernel bias sum: channel sums of the widened (already masked) gy instead
        r = conv_wgrad_mfma(gp2, xp, gp, none, gw, nullptr, rest, rest_bytes, s);
        if (r == 0) r = ggan_chansum(gp, gbias, g.N, g.Co, g.Ho * Wop, rest, rest_bytes, (ggan_stream_t)s);
    }
    return r;
}

static int bwd_filter(const ggan_conv_geom* g, const float* x, const float* gy, GyMask m, float* gw, float* gbias, void* ws,
                      size_t ws_bytes, ggan_stream_t stream) {
    if (check_geom(g)) return -1;
    if (!(x && gy && gw)) { set_error("ggan_conv2d_bwd_filter: null pointer"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (!(g->plan_flags & GGAN_PLAN_PLAIN) && !getenv("GGAN_NAIVE_WGRAD")) {
        int r = conv_wgrad_thin(*g, x, gy, m, gw, gbias, ws, ws ? ws_bytes : 0, s);
        if (r <= 0) return r;
        r = conv_wgrad_mfma(*g, x, gy, m, gw, gbias, ws, ws ? ws_bytes : 0, s);
        if (r <= 0) return r;
        r = wgrad_widened(*g, x, gy, m, gw, gbias, ws, ws ? ws_bytes : 0, s);
        if (r <= 0) return r;
    }
    if (gbias) {   // plain path: separate reduction (needs the masked gradient materialised only when a mask is given)
        if (m.act != GGAN_ACT_NONE) return 1;   // masked bias gradient only exists fused: caller uses act_bwd + the plain entry points
        int r = ggan_chansum(gy, gbias, g->N, g->Co, g->Ho * g->Wo, ws, ws_bytes, stream);
        if (r) return r;
    }
    if (getenv("GGAN_TRACE_NAIVE"))
        fprintf(stderr, "[ggan] plain filter-gradient kernel for N=%d Ci=%d H=%d W=%d Co=%d Ho=%d Wo=%d k=%d s=%d pad=(%d,%d)\n", g->N, g->Ci,
                g->H, g->W, g->Co, g->Ho, g->Wo, g->k, g->stride, g->pad_t, g->pad_l);
    return conv_wgrad_naive(*g, x, gy, m, gw, s);
}

int ggan_conv2d_bwd_data(const ggan_conv_geom* g, const float* gy, const float* w, const float* bias, float* gx, int act,
                         float alpha, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    return bwd_data(g, gy, GyMask{nullptr, GGAN_ACT_NONE, 0.f}, w, bias, gx, act, alpha, ws, ws_bytes, stream);
}

int ggan_conv2d_bwd_filter(const ggan_conv_geom* g, const float* x, const float* gy, float* gw, float* gbias, void* ws,
                           size_t ws_bytes, ggan_stream_t stream) {
    return bwd_filter(g, x, gy, GyMask{nullptr, GGAN_ACT_NONE, 0.f}, gw, gbias, ws, ws_bytes, stream);
}

int ggan_conv2d_bwd_data_act(const ggan_conv_geom* g, const float* gy, const float* y, int y_act, float y_alpha, const float* w,
                             float* gx, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    GGAN_CHECK_ARG(y || y_act == GGAN_ACT_NONE, "null activation reference");
    return bwd_data(g, gy, GyMask{y, y_act, y_alpha}, w, nullptr, gx, GGAN_ACT_NONE, 0.f, ws, ws_bytes, stream);
}

int ggan_conv2d_bwd_filter_act(const ggan_conv_geom* g, const float* x, const float* gy, const float* y, int y_act, float y_alpha,
                               float* gw, float* gbias, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    GGAN_CHECK_ARG(y || y_act == GGAN_ACT_NONE, "null activation reference");
    return bwd_filter(g, x, gy, GyMask{y, y_act, y_alpha}, gw, gbias, ws, ws_bytes, stream);
}

int ggan_conv2d_bwd_filter_parts(const ggan_conv_geom* g, const float* x, const float* gy, const float* y, int y_act,
                                 float y_alpha, int with_bias, float* part, size_t part_cap, int* n_parts, size_t* stride,
                                 ggan_stream_t stream) {
    if (check_geom(g)) return -1;
    GGAN_CHECK_ARG(x && gy && part && n_parts && stride, "null pointer");
    GGAN_CHECK_ARG(y || y_act == GGAN_ACT_NONE, "null activation reference");
    if ((g->plan_flags & GGAN_PLAN_PLAIN) || getenv("GGAN_NAIVE_WGRAD")) return 1;
    WgradParts po{part, part_cap, with_bias, 0, 0};
    int r = conv_wgrad_thin(*g, x, gy, GyMask{y, y_act, y_alpha}, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, &po);
    if (r == 1) r = conv_wgrad_mfma(*g, x, gy, GyMask{y, y_act, y_alpha}, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, &po);
    if (r == 0) { *n_parts = po.n; *stride = po.stride; }
    return r;
}

// y = conv(x, w) masked by the derivative of the activation that produced yref (see ggan.h)
int ggan_conv2d_fwd_masked(const ggan_conv_geom* g, const float* x, const float* w, float* y, const float* yref, int ref_act,
                           float ref_alpha, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    if (check_geom(g)) return -1;
    GGAN_CHECK_ARG(x && w && y && yref, "null pointer");
    if ((g->plan_flags & GGAN_PLAN_PLAIN) || getenv("GGAN_NAIVE_FWD") || getenv("GGAN_NO_FWD_MASK")) return 1;
    if ((((uintptr_t)yref) & 15) != 0) return 1;
    // (thin first layers: the thin-channel forward kernel with the mask in its epilogue since round 6 -- K = 25 * Ci instead of 25 x a padded
    //  16-channel chunk; rounds 3-5 ran the padded-channel MFMA launch with the mask.  GGAN_FWD_MASK_THIN=0 selects that one,
    //  GGAN_NO_FWD_MASK_THIN the unfused pair conv_thin.hip + act_bwd)
    if (g->Ci <= 4 && getenv("GGAN_NO_FWD_MASK_THIN")) return 1;
    OutMask M{yref, ref_act, ref_alpha, false};
    if (g->Ci <= 4 && (ref_act == GGAN_ACT_LRELU || ref_act == GGAN_ACT_RELU)) {
        const char* e = getenv("GGAN_FWD_MASK_THIN");
        if (!e || atoi(e) != 0) {
            const int r = conv_fwd_thin(*g, x, w, nullptr, y, GGAN_ACT_NONE, 0.f, (hipStream_t)stream, nullptr, &M);
            if (r <= 0) return r;
        }
    }
    g_out_mask = &M;
    const int rc = conv_fwd_mfma(*g, x, w, nullptr, y, GGAN_ACT_NONE, 0.f, ws, ws ? ws_bytes : 0, (hipStream_t)stream);
    g_out_mask = nullptr;
    if (rc == 0 && !M.applied) { set_error("conv2d_fwd_masked: launch without the mask"); return -3; }
    return rc;
}

// Deconv2D = the adjoint family with the same filter bytes (see ggan.h)
int ggan_deconv2d_fwd(const ggan_conv_geom* g, const float* x_small, const float* w, const float* bias, float* y_big,
                      int act, float alpha, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    return ggan_conv2d_bwd_data(g, x_small, w, bias, y_big, act, alpha, ws, ws_bytes, stream);
}

int ggan_deconv2d_bwd_data(const ggan_conv_geom* g, const float* gy_big, const float* w, float* gx_small, void* ws,
                           size_t ws_bytes, ggan_stream_t stream) {
    return ggan_conv2d_fwd(g, gy_big, w, nullptr, gx_small, GGAN_ACT_NONE, 0.f, ws, ws_bytes, stream);
}

int ggan_deconv2d_bwd_filter(const ggan_conv_geom* g, const float* gy_big, const float* x_small, float* gw, float* gbias,
                             void* ws, size_t ws_bytes, ggan_stream_t stream) {
    if (check_geom(g)) return -1;
    if (gbias) {
        GGAN_CHECK_ARG(gy_big, "null pointer");
        int r = ggan_chansum(gy_big, gbias, g->N, g->Ci, g->H * g->W, ws, ws_bytes, stream);
        if (r) return r;
    }
    return ggan_conv2d_bwd_filter(g, gy_big, x_small, gw, nullptr, ws, ws_bytes, stream);
}

}  // extern "C"
