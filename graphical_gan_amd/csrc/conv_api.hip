// C-ABI entry points of the Conv2D / Deconv2D family: argument checks + dispatch between the MFMA kernels
// (5x5 stride-2 hot path) and the plain kernels (everything else, or ggan_set_naive(1)).
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
    if (!g_force_naive && !getenv("GGAN_NAIVE_FWD")) {
        int r = conv_fwd_mfma(*g, x, w, bias, y, act, alpha, ws, ws ? ws_bytes : 0, s);
        if (r <= 0) return r;
    }
    return conv_fwd_naive(*g, x, w, bias, y, act, alpha, s);
}

static int bwd_data(const ggan_conv_geom* g, const float* gy, GyMask m, const float* w, const float* bias, float* gx, int act,
                    float alpha, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    if (check_geom(g)) return -1;
    if (!(gy && w && gx)) { set_error("ggan_conv2d_bwd_data: null pointer"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (!g_force_naive && !getenv("GGAN_NAIVE_DGRAD")) {
        int r = conv_dgrad_mfma(*g, gy, m, w, bias, gx, act, alpha, ws, ws ? ws_bytes : 0, s);
        if (r <= 0) return r;
    }
    return conv_dgrad_naive(*g, gy, m, w, bias, gx, act, alpha, s);
}

static int bwd_filter(const ggan_conv_geom* g, const float* x, const float* gy, GyMask m, float* gw, float* gbias, void* ws,
                      size_t ws_bytes, ggan_stream_t stream) {
    if (check_geom(g)) return -1;
    if (!(x && gy && gw)) { set_error("ggan_conv2d_bwd_filter: null pointer"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (!g_force_naive && !getenv("GGAN_NAIVE_WGRAD")) {
        int r = conv_wgrad_mfma(*g, x, gy, m, gw, gbias, ws, ws ? ws_bytes : 0, s);
        if (r <= 0) return r;
    }
    if (gbias) {   // plain path: separate reduction (needs the masked gradient materialised only when a mask is given)
        if (m.act != GGAN_ACT_NONE) return 1;   // masked bias gradient only exists fused: caller uses act_bwd + the plain entry points
        int r = ggan_chansum(gy, gbias, g->N, g->Co, g->Ho * g->Wo, ws, ws_bytes, stream);
        if (r) return r;
    }
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
    if (g_force_naive || getenv("GGAN_NAIVE_WGRAD")) return 1;
    WgradParts po{part, part_cap, with_bias, 0, 0};
    int r = conv_wgrad_mfma(*g, x, gy, GyMask{y, y_act, y_alpha}, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, &po);
    if (r == 0) { *n_parts = po.n; *stride = po.stride; }
    return r;
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
