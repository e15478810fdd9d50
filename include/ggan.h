/*
 * ggan.h -- C ABI of libggan.so, the MI355X (gfx950) kernels behind the tflib.ops operator API
 * of zhenxuan00/graphical-gan.
 *
 * Boundary: the reference's Python operator constructors
 *   tflib/ops/conv2d.py:20 Conv2D, tflib/ops/deconv2d.py:20 Deconv2D, tflib/ops/linear.py:24 Linear,
 *   tflib/ops/batchnorm.py:6 Batchnorm, tflib/objs/gan_inference.py:28-119,307-358 objectives
 * hand their arithmetic to TensorFlow kernels (tf.nn.conv2d, tf.nn.conv2d_transpose, tf.matmul,
 * tf.nn.fused_batch_norm, tf.nn.sigmoid_cross_entropy_with_logits, AdamOptimizer, tf.gradients).
 * Each entry point below replaces one of those TF kernels (cited per function).
 *
 * Conventions
 *   - all tensors are dense fp32 in device memory; activations NCHW; the caller owns every buffer;
 *   - Conv2D filters HWIO [k][k][Cin][Cout]; Deconv2D filters [k][k][Cout][Cin] (reference layouts);
 *   - every function is asynchronous on `stream` (a hipStream_t passed as void*), allocates
 *     nothing, keeps no state between calls and is hipGraph-capturable;
 *   - `ws` is caller-provided scratch of at least ggan_*_workspace() bytes (may be NULL when 0).  Its first
 *     GGAN_WS_RESERVED bytes hold arrival counters of in-kernel split-K combines: the caller zeroes them ONCE when it
 *     allocates the workspace, every kernel leaves them zero; calls sharing a workspace must be stream-ordered;
 *   - return value 0 = ok, negative = error; ggan_last_error() gives the message (thread-local).
 */
#ifndef GGAN_H
#define GGAN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ggan_stream_t; /* hipStream_t */
#define GGAN_WS_RESERVED 16384

/* activation codes for fused epilogues and ggan_act_* */
enum { GGAN_ACT_NONE = 0, GGAN_ACT_LRELU = 1, GGAN_ACT_RELU = 2, GGAN_ACT_TANH = 3, GGAN_ACT_SIGMOID = 4 };

/* The ABI's version: changes whenever a struct layout or an entry point's meaning changes (500: ggan_conv_geom carries the launch plan,
 * ggan_prof_rec the grid; 600: a profiler record is one problem SHAPE -- (kernel, grid, flop per launch) --, ggan_noise_fill_steps is
 * gone).  A binding compares it with the header it was written against before the first call. */
#define GGAN_ABI_VERSION 600
int ggan_version(void);
const char* ggan_last_error(void);
/* ---- convolution geometry ------------------------------------------------------------------
 * A strided cross-correlation y[N,Co,Ho,Wo] = conv(x[N,Ci,H,W], w[k,k,Ci,Co]) with explicit
 * top/left padding (TF 'SAME' puts the extra row/col at the bottom/right -- SURVEY.md A.1; the
 * caller computes pad_t/pad_l = pad_total/2, bottom/right padding is implied by Ho/Wo). */
typedef struct {
    int N, Ci, H, W;      /* the LARGE-spatial tensor (conv input / deconv output)  */
    int Co, Ho, Wo;       /* the SMALL-spatial tensor (conv output / deconv input)  */
    int k, stride, pad_t, pad_l;
    /* The launch plan travels with the call (round 4: these were process-wide setters; the library keeps no mutable state besides the
     * guarded profiler table and the caches of plan-time tables, so calls from several threads / streams do not interfere):
     *   plan_wgs         workgroups the forward / data-gradient launch plans for (tile size, split-K); 0 = default (GGAN_TARGET_WGS or
     *                    200, about one per CU: right for a launch that has the chip to itself).  A caller running TWO conv chains side
     *                    by side on two streams asks for ~128, so that each launch leaves CUs to the other chain.
     *   plan_wgs_filter  the same for the filter-gradient launch (0 = GGAN_WGRAD_WGS or 256 split-K workgroups)
     *   plan_flags       GGAN_PLAN_PLAIN: the plain one-thread-per-output kernels (debug cross-check) */
    int plan_wgs, plan_wgs_filter, plan_flags;
} ggan_conv_geom;
#define GGAN_PLAN_PLAIN 1

/* tf.nn.conv2d(NCHW) + tf.nn.bias_add (tflib/ops/conv2d.py:106-120).  bias may be NULL.
 * act/alpha: optional fused pointwise epilogue (GGAN_ACT_NONE for reference behaviour). */
size_t ggan_conv2d_workspace(const ggan_conv_geom* g);
int ggan_conv2d_fwd(const ggan_conv_geom* g, const float* x, const float* w, const float* bias,
                    float* y, int act, float alpha, void* ws, size_t ws_bytes, ggan_stream_t stream);
/* Conv2DBackpropInput: gx[N,Ci,H,W] from gy[N,Co,Ho,Wo] (what tf.gradients derives from conv2d.py:106).
 * bias (len Ci, may be NULL) / act are a fused epilogue used by the Deconv2D forward. */
int ggan_conv2d_bwd_data(const ggan_conv_geom* g, const float* gy, const float* w, const float* bias,
                         float* gx, int act, float alpha, void* ws, size_t ws_bytes, ggan_stream_t stream);
/* Conv2DBackpropFilter: gw[k,k,Ci,Co]; gbias[Co] = sum over N,Ho,Wo of gy (BiasAddGrad), may be NULL. */
int ggan_conv2d_bwd_filter(const ggan_conv_geom* g, const float* x, const float* gy, float* gw,
                           float* gbias, void* ws, size_t ws_bytes, ggan_stream_t stream);
/* Backward of a FUSED conv+bias+activation layer: the two gradients above computed from gy[i]*act'(y[i]) where y is the
 * layer's saved output, with the activation derivative applied while gy is staged (no separate pointwise pass and no
 * intermediate tensor) and the bias gradient produced by the filter-gradient kernel from the tiles it stages anyway. */
int ggan_conv2d_bwd_data_act(const ggan_conv_geom* g, const float* gy, const float* y, int y_act, float y_alpha,
                             const float* w, float* gx, void* ws, size_t ws_bytes, ggan_stream_t stream);
int ggan_conv2d_bwd_filter_act(const ggan_conv_geom* g, const float* x, const float* gy, const float* y, int y_act,
                               float y_alpha, float* gw, float* gbias, void* ws, size_t ws_bytes, ggan_stream_t stream);
/* Filter gradient left as its split-K partial slabs, for a consumer that sums them anyway (ggan_pack_parts): saves the
 * reduce launch.  part[s*stride + i], s < *n_parts, i < 25*Ci*Co is slab s of gw; with_bias != 0 appends the Co
 * bias-gradient partials to every slab (i in [25*Ci*Co, 25*Ci*Co+Co)).  part_cap (floats) bounds the number of slabs.
 * Returns 1 when the geometry is not covered by the MFMA kernel (nothing written; use ggan_conv2d_bwd_filter_act). */
int ggan_conv2d_bwd_filter_parts(const ggan_conv_geom* g, const float* x, const float* gy, const float* y, int y_act,
                                 float y_alpha, int with_bias, float* part, size_t part_cap, int* n_parts,
                                 size_t* stride, ggan_stream_t stream);

/* tf.nn.conv2d_transpose + bias_add (tflib/ops/deconv2d.py:101-114) computed natively in NCHW (the two
 * layout transposes at :91/:116 are mathematically no-ops).  g describes the forward conv whose
 * input-gradient this is: x_small[N,Co,Ho,Wo] -> y_big[N,Ci,H,W]; w is the Deconv2D filter
 * [k,k,out=Ci,in=Co], which is bit-for-bit the HWIO filter of that forward conv. */
int ggan_deconv2d_fwd(const ggan_conv_geom* g, const float* x_small, const float* w, const float* bias,
                      float* y_big, int act, float alpha, void* ws, size_t ws_bytes, ggan_stream_t stream);
int ggan_deconv2d_bwd_data(const ggan_conv_geom* g, const float* gy_big, const float* w, float* gx_small,
                           void* ws, size_t ws_bytes, ggan_stream_t stream);
int ggan_deconv2d_bwd_filter(const ggan_conv_geom* g, const float* gy_big, const float* x_small, float* gw,
                             float* gbias /* [Ci], sum of gy_big */, void* ws, size_t ws_bytes,
                             ggan_stream_t stream);

/* Linear backward with the following activation's derivative folded into the operand load, so g' = g * act'(y) is never
 * written (replaces tf.matmul's MatMul gradients + the script's tf.maximum/LeakyReLU gradient + BiasAddGrad,
 * tflib/ops/linear.py:133-146).  x[M,K] layer input, w[K,N], y[M,N] the layer's activated output, g[M,N] = dL/dy.
 *   ggan_linear_bwd_data_act:   dx[M,K] = g' w^T
 *   ggan_linear_bwd_weight_act: dw[K,N] = x^T g',  db[N] = column sums of g' (db may be NULL) */
int ggan_linear_bwd_data_act(int M, int N, int K, const float* g, const float* y, int y_act, float y_alpha,
                             const float* w, float* dx, void* ws, size_t ws_bytes, ggan_stream_t stream);
int ggan_linear_bwd_weight_act(int M, int N, int K, const float* x, const float* g, const float* y, int y_act,
                               float y_alpha, float* dw, float* db, void* ws, size_t ws_bytes, ggan_stream_t stream);

/* ggan_gemm with "concatenated" operands that are never concatenated (the critic's Linear on [conv features | z features],
 * gan_inference_cifar10.py:246-248 tf.concat + Linear, and its two gradients):
 *   A2 != NULL: op(A) = [A | A2] side by side along the SECOND stored dimension of A (k when ta == 0, m when ta == 1); the
 *               first a_split columns come from A (leading dimension a_split), the rest from A2 (leading dimension whole -
 *               a_split); a_split must be a multiple of 64.
 *   C2 != NULL: output columns [0, c_split) go to C (leading dimension c_split), [c_split, N) to C2 (leading dimension N -
 *               c_split); c_split a multiple of 64; no bias / activation / column sums with a split output.
 *   colsum_b (may be NULL; tb == 0 only): column sums of B, as ggan_gemm_colsum. */
int ggan_gemm_split(int ta, int tb, int M, int N, int K, const float* A, const float* A2, int a_split, const float* B,
                    const float* bias, float* C, float* C2, int c_split, float* colsum_b, int act, float alpha, void* ws,
                    size_t ws_bytes, ggan_stream_t stream);

/* The state-space scripts' transition operator unrolled over a sequence, as ONE scan per direction
 * (/root/reference/ssgan_inference_moving_mnist.py:98-114 ImplicitOperator, :134-141 DynamicGenerator; 'res_w' variant:
 * ssgan_inference_chairs.py): z_{t+1} = Linear_out(lrelu(Linear_1(lrelu(Linear_in([z_t | eps]))))) + z_t   (zw == NULL)
 *                                      ... + z_t zw + b_zw                                                 (zw != NULL)
 * for t = 0 .. T-1 (T = LEN-1), one workgroup per sequence.  H (DIM_OP) must be 256, dl, dt <= 16.
 *   forward : zs[B, T+1, dl] (zs[:, 0] = z0), h1 / h2 [T, B, H] = the two hidden activations (kept for the backward).
 *   backward: from g_zs[B, T+1, dl] = d cost / d zs: the masked hidden-layer gradients G1, G2 [T, B, H], the gradient at every
 *             operator output Go[T, B, dl], the operator inputs Xin[T, B, dl+dt], d_z0[B, dl], d_eps[B, dt] (may be NULL).  The
 *             weight gradients are products over all T*B rows: dW_in = Xin^T G1, dW_1 = h1^T G2, dW_out = h2^T Go,
 *             dZW = Xin[:, :dl]^T Go, biases = column sums (ggan_gemm_colsum). */
int ggan_dyn_scan_fwd(int B, int T, int dl, int dt, int H, const float* z0, const float* eps, const float* w_in, const float* b_in,
                      const float* w_1, const float* b_1, const float* w_out, const float* b_out, const float* zw, const float* b_zw,
                      float alpha, float* zs, float* h1, float* h2, ggan_stream_t stream);
int ggan_dyn_scan_bwd(int B, int T, int dl, int dt, int H, const float* g_zs, const float* zs, const float* eps, const float* h1,
                      const float* h2, const float* w_in, const float* w_1, const float* w_out, const float* zw, float alpha, float* G1,
                      float* G2, float* Go, float* Xin, float* d_z0, float* d_eps, ggan_stream_t stream);

/* The tail of a critic as ONE op: Linear ([a1 | a2] -> H) + LeakyReLU(alpha) + Linear (H -> 1).
 * Replaces, for the joint critic, tf.concat + lib.ops.linear.Linear('Discriminator.zx1') + LeakyReLU +
 * lib.ops.linear.Linear('Discriminator.Output') (/root/reference/gan_inference_cifar10.py:246-254,
 * gmgan_inference_cifar10.py:294-300) and, with a2 == NULL, 'Discriminator.Hyper3' + 'Discriminator.HyperOutput'
 * (gmgan_inference_cifar10.py:288-291) and the state-space critics' last two layers.
 *   forward:  h[M,H] = lrelu([a1|a2] w + b) (kept: the backward's activation reference), logits[M] = h w_out + b_out.
 *             a1 [M,K1], a2 [M,K2] (NULL when K2 == 0; K1 a multiple of 64 otherwise), w [K1+K2,H], b [H], w_out [H], b_out [1].
 *   backward: from g[M] = d cost / d logits: gh[M,H] = g w_out^T * lrelu'(h) (caller-owned scratch), d_wout[H], d_bout[1],
 *             d_w[K1+K2,H], d_b[H], d_a1[M,K1], d_a2[M,K2]; every output pointer may be NULL (not wanted), d_b needs d_w,
 *             d_a2 goes with d_a1.  Launches: one head kernel + one grouped product launch.  g == NULL: gh (and d_wout / d_bout,
 *             which are then ignored here) were produced by ggan_bce_head_bwd -- only the products are launched. */
int ggan_critic_head_fwd(int M, int K1, int K2, int H, const float* a1, const float* a2, const float* w, const float* b,
                         const float* w_out, const float* b_out, float alpha, float* h, float* logits, void* ws, size_t ws_bytes,
                         ggan_stream_t stream);
int ggan_critic_head_bwd(int M, int K1, int K2, int H, const float* g, const float* a1, const float* a2, const float* w, const float* h,
                         const float* w_out, float alpha, float* gh, float* d_a1, float* d_a2, float* d_w, float* d_b, float* d_wout,
                         float* d_bout, void* ws, size_t ws_bytes, ggan_stream_t stream);

/* The same head when its caller knows BEFORE it runs that its logits feed one sigmoid-cross-entropy cost
 * (/root/reference/tflib/objs/gan_inference.py:104-117: cost = sum_k w_k * reduce_mean(sigmoid_cross_entropy_with_logits(rows of term k,
 * label z_k)); terms = 1..4 consecutive row ranges covering the M logits): the gradient of that cost for a unit upstream gradient is
 * row-local, so the forward's tail launch also leaves g[M] = d cost / d logits and gh[M,H] = g w_out^T * lrelu'(h), and the backward's
 * product launch carries, as extra workgroups, what needs all rows: loss[0] (bit-identical to ggan_bce_logits_multi_fwd on the same
 * terms), d_wout[H] = h^T g, d_bout[1] = sum g.  One launch less per step on the chain  tail product -> logits -> cost -> head backward ->
 * products  than ggan_critic_head_fwd + ggan_bce_head_bwd + ggan_critic_head_bwd(g = NULL); same values bit for bit.  H <= 2048.
 * kind 0: sigmoid cross-entropy terms; kind 1: plain means (the Wasserstein costs of gan_inference.py:4-45, sum_k w_k * reduce_mean(rows of term k);
 * labels unused).  ext (backward only, may be NULL): ext[k] != NULL makes term k a value read THERE (term_rows[k] floats) instead of a row range -- the
 * one-element gradient penalty that a wali-gp critic cost adds (gan_inference_cifar10.py:365); such a term only enters loss[0].  The cost's value is
 * bit-identical to ggan_bce_logits_multi_fwd / ggan_mean_multi_fwd_grad on the same terms. */
int ggan_critic_head_fwd_bce(int M, int K1, int K2, int H, const float* a1, const float* a2, const float* w, const float* b,
                             const float* w_out, const float* b_out, float alpha, float* h, float* logits, int kind, int nterms,
                             const int* term_rows, const float* labels, const float* weights, float* g, float* gh, void* ws,
                             size_t ws_bytes, ggan_stream_t stream);
int ggan_critic_head_bwd_tail(int M, int K1, int K2, int H, const float* a1, const float* a2, const float* w, const float* h,
                              const float* w_out, float alpha, const float* gh, float* d_a1, float* d_a2, float* d_w, float* d_b,
                              float* d_wout, float* d_bout, const float* logits, const float* g, int kind, int nterms, const int* term_rows,
                              const float* labels, const float* weights, const float* const* ext, float* loss, void* ws, size_t ws_bytes,
                              ggan_stream_t stream);

/* ---- dense -------------------------------------------------------------------------------
 * C[M,N] = op(A) * op(B) (+ bias[N]) (+act), row-major, ta/tb = 1 reads the operand transposed
 * (A stored [K,M] / B stored [N,K]).  tf.matmul + bias_add of tflib/ops/linear.py:133-146 is
 * (ta=0,tb=0); its gradients are (0,1) for dX = dY*W^T and (1,0) for dW = X^T*dY. */
size_t ggan_gemm_workspace(int M, int N, int K);
int ggan_gemm(int ta, int tb, int M, int N, int K, const float* A, const float* B, const float* bias,
              float* C, int act, float alpha, void* ws, size_t ws_bytes, ggan_stream_t stream);
/* Same, plus colsum_b[n] = sum_k op(B)[k][n] from the B tiles the kernel stages anyway (tb must be 0): dW = X^T dY and
 * db = sum_rows dY of the Linear backward in ONE launch. */
int ggan_gemm_colsum(int ta, int M, int N, int K, const float* A, const float* B, float* C, float* colsum_b,
                     void* ws, size_t ws_bytes, ggan_stream_t stream);
/* out[c] = sum_r x[r,c] over a [rows,cols] matrix (BiasAddGrad of Linear). */
int ggan_colsum(const float* x, float* out, int rows, int cols, ggan_stream_t stream);
/* the same sum for tall matrices (Conv3D bias gradients: 10^5..10^6 rows): row slabs + a fixed-order second stage; ws = scratch */
int ggan_colsum_tall(const float* x, float* out, int rows, int cols, void* ws, size_t ws_bytes, ggan_stream_t stream);
/* out[c] = sum_{n,hw} x[n,c,hw] (BiasAddGrad NCHW).  ws: optional scratch (>= 4 KiB * C) enabling a chip-wide
 * two-stage reduction; with ws == NULL one workgroup per channel is used. */
int ggan_chansum(const float* x, float* out, int N, int C, int HW, void* ws, size_t ws_bytes, ggan_stream_t stream);

/* ---- batch normalisation, training mode, batch statistics, biased variance -------------------
 * tf.nn.fused_batch_norm(NCHW, eps) (tflib/ops/batchnorm.py:29-30) for HW>1 and the
 * tf.nn.moments + tf.nn.batch_normalization branch (:74-87) as the HW=1 case over [N,C].
 * save_mean/save_invstd: [C] each, written by fwd and consumed by bwd. */
int ggan_bn_fwd_train(const float* x, const float* scale, const float* offset, float* y,
                      float* save_mean, float* save_invstd, int N, int C, int HW, float eps,
                      int act, float alpha, ggan_stream_t stream);
int ggan_bn_bwd(const float* x, const float* gy, const float* scale, const float* save_mean,
                const float* save_invstd, float* gx, float* gscale, float* goffset,
                int N, int C, int HW, ggan_stream_t stream);
/* same, with the activation fused into the forward (ggan_bn_fwd_train act != NONE) differentiated on load: gy is
 * dL/d(activated output y); no separate ggan_act_bwd pass.  gx_chansum (may be NULL; HW > 1 only) receives
 * sum_{n,h,w} gx per channel = the BiasAddGrad of the layer that feeds this BatchNorm, from the values in registers. */
int ggan_bn_bwd_act(const float* x, const float* gy, const float* y, int y_act, float y_alpha, const float* scale,
                    const float* save_mean, const float* save_invstd, float* gx, float* gscale, float* goffset,
                    float* gx_chansum, int N, int C, int HW, ggan_stream_t stream);

/* y = conv2d(x, w) * act'(yref), act' the derivative of the activation (ref_act, ref_alpha) that produced yref, taken at yref (as
 * ggan_act_bwd does): ggan_conv2d_fwd followed by ggan_act_bwd in one launch.  This is the backward of ggan_conv2d_bwd_data_act with
 * respect to its gy operand -- what the gradient-penalty pass of MODE wali-gp differentiates through (gan_inference_cifar10.py:353-364:
 * tf.gradients of a cost that contains tf.gradients(disc_hat, [x_hat])).  Returns 1, having launched nothing, where no kernel fuses
 * the mask for this geometry (filters other than 5x5 stride 2, split-K launches): the caller composes the two calls. */
int ggan_conv2d_fwd_masked(const ggan_conv_geom* g, const float* x, const float* w, float* y, const float* yref, int ref_act,
                           float ref_alpha, void* ws, size_t ws_bytes, ggan_stream_t stream);

/* Linear + Batchnorm([0]) + activation in one launch: the head of every Generator of the image scripts
 * (gan_inference_cifar10.py:134-138: Linear 'Generator.Input' -> Batchnorm 'Generator.BN1' over the batch axis -> relu).
 * x [M,K], w [K,N], b [N] (may be NULL); h [M,N] = x @ w + b (BatchNorm's input, kept for its backward), y [M,N] =
 * act(scale * (h - mean) * invstd + offset) with the batch statistics of the M rows, save_mean / save_invstd [N] as
 * ggan_bn_fwd_train leaves them (the backward is ggan_bn_bwd_act followed by the Linear layer's gradients).  Returns 1 when the
 * shape is not covered (M > 128 or not a multiple of 16, K > 256 or not a multiple of 4, N not a multiple of 32): nothing
 * written, use ggan_gemm + ggan_bn_fwd_train. */
int ggan_linear_bn_rows_fwd(const float* x, const float* w, const float* b, const float* scale, const float* offset, float* h,
                            float* y, float* save_mean, float* save_invstd, int M, int K, int N, float eps, int act,
                            float alpha, ggan_stream_t stream);
/* The backward of that head where its input needs no gradient (the generators' input is noise): ggan_bn_bwd_act over the rows + the
 * weight-gradient product x^T gh + its column sums in ONE launch (the last two launches of the Generator's backward chain, in front of
 * the step's pack + Adam launch).  gy [M,N] = d cost / d y, h / y / save_mean / save_invstd as the forward left them (y: the activation
 * reference, NULL for act == NONE); dw [K,N], db [N] (may be NULL), dscale [N], doffset [N].  Returns 1 when the shape is not covered
 * (K in {64, 128, 256}, M <= 128 in 16 equal row groups, N a multiple of 32): the caller composes ggan_bn_bwd_act + ggan_gemm_colsum. */
int ggan_linear_bn_rows_bwd(const float* x, const float* gy, const float* h, const float* y, const float* scale, const float* save_mean,
                            const float* save_invstd, float* dw, float* db, float* dscale, float* doffset, int M, int K, int N, int act,
                            float alpha, ggan_stream_t stream);

/* Second derivative of ggan_bn_bwd_act w.r.t. its inputs, for objectives that differentiate a network containing BatchNorm twice
 * (MODE vegan-wgan-gp: the gradient penalty on the latent critic, gan_inference_cifar10.py:305-317 with BN_FLAG = True).  h =
 * dL/d(gx).  Outputs: ggy = dL/d(gy), gx2 = dL/d(x) (all of it: through the batch statistics too), gscale2 = dL/d(scale).  The
 * fused activation must be piecewise linear (lrelu / relu / none). */
int ggan_bn_bwd_bwd(const float* x, const float* gy, const float* y, int y_act, float y_alpha, const float* h, const float* scale,
                    const float* save_mean, const float* save_invstd, float* ggy, float* gx2, float* gscale2, int N, int C, int HW,
                    ggan_stream_t stream);

/* Cross-replica ("sync") BatchNorm, SURVEY.md 8(e): statistics over the GLOBAL batch of `world` equal-sized replicas.  The
 * reference has no multi-GPU path; this is the mode under which N GPUs x B/N reproduce 1 GPU x B.  The host all-gathers the
 * 2*C floats each *_stats call produces ([2][C]: forward (mean, M2), backward (sum g, sum g*xhat)) into [world][2][C] and hands
 * them to the *_apply call, which merges them in rank order (bit-identical on every replica).  gscale / goffset are this
 * replica's own sums: the gradient exchange averages them with the other parameter gradients.  No second derivative. */
int ggan_bn_sync_stats(const float* x, float* stats, int N, int C, int HW, ggan_stream_t stream);
int ggan_bn_sync_apply(const float* x, const float* stats, int world, const float* scale, const float* offset, float* y,
                       float* save_mean, float* save_invstd, int N, int C, int HW, float eps, int act, float alpha,
                       ggan_stream_t stream);
int ggan_bn_sync_bwd_stats(const float* x, const float* gy, const float* y, int y_act, float y_alpha, const float* save_mean,
                           const float* save_invstd, float* sums, int N, int C, int HW, ggan_stream_t stream);
int ggan_bn_sync_bwd_apply(const float* x, const float* gy, const float* y, int y_act, float y_alpha, const float* scale,
                           const float* save_mean, const float* save_invstd, const float* sums, int world, int rank, float* gx,
                           float* gscale, float* goffset, int N, int C, int HW, ggan_stream_t stream);

/* ---- pointwise -------------------------------------------------------------------------------
 * LeakyReLU = tf.maximum(alpha*x, x) (gmgan_inference_cifar10.py:122-123), tf.nn.relu, tf.tanh,
 * tf.nn.sigmoid.  bwd takes the forward INPUT for lrelu/relu and the forward OUTPUT for
 * tanh/sigmoid in `ref`. */
int ggan_act_fwd(const float* x, float* y, size_t n, int act, float alpha, ggan_stream_t stream);
int ggan_act_bwd(const float* gy, const float* ref, float* gx, size_t n, int act, float alpha,
                 ggan_stream_t stream);
/* ggan_act_bwd that also emits the bias gradient of the layer as partial slabs: gx = gy * act'(ref) over [N, C, HW] and
 * parts[s*C + c] = sum of gx over channel c of the s-th image range, s < *n_parts (<= parts_cap / C); the slabs are summed by
 * ggan_pack_parts (stride C).  One launch for what tf.nn.bias_add's BiasAddGrad and the activation's gradient op are in the
 * backward of Deconv2D / Conv2D + activation (tflib/ops/deconv2d.py:110-116 followed by tf.nn.relu / tanh in the scripts). */
int ggan_act_bwd_chansum(const float* gy, const float* ref, float* gx, float* parts, int parts_cap, int* n_parts, int N, int C, int HW,
                         int act, float alpha, ggan_stream_t stream);
/* y = x + bias broadcast: NCHW bias[C] (HW>1) or [rows,C] (HW=1).  tf.nn.bias_add. */
int ggan_bias_add(const float* x, const float* bias, float* y, int N, int C, int HW, ggan_stream_t stream);
/* real_x = mul*(float(int32)/div - .5) (+ noise) (gmgan_inference_cifar10.py:342; face :242-243). */
int ggan_cast_scale_i32(const int32_t* x, const float* noise /* may be NULL */, float* y, size_t n,
                        float div, float mul, ggan_stream_t stream);
/* The same op reading its minibatch from a device-resident ring of `nslots` pre-staged minibatches of n int32 each: slot
 * (*ctr_a + *ctr_b + offset) mod nslots (either counter may be NULL).  The counters are device integers that other launches of
 * the step advance (the optimizers' step counts), so a replayed HIP graph walks the ring without a host-side copy of the next
 * minibatch into a staging buffer -- what the feed_dict of session.run is to the reference (gan_inference_cifar10.py:392-401),
 * with the data already in HBM. */
int ggan_cast_scale_ring_i32(const int32_t* ring, int nslots, const int32_t* ctr_a, const int32_t* ctr_b, int offset,
                             const float* noise /* may be NULL */, float* y, size_t n, float div, float mul, ggan_stream_t stream);
/* ggan_cast_scale_ring_i32 and the ggan_conv2d_fwd that reads its result, in ONE launch: real_x = 2*((tf.cast(real_x_int, tf.float32)/255.)-.5)
 * in front of lib.ops.conv2d.Conv2D('Extractor.1', 3, DIM, 5, ., stride=2) (/root/reference/gan_inference_cifar10.py:342,155-157).  The
 * scaled image x_out [N,Ci,H,W] is still written (the critic's input and the layer's filter gradient read it).  Returns 1 without
 * launching where the geometry is outside the thin-channel forward kernel (Ci <= 4, Co 32 | 64, k 5, stride 2): issue the two calls. */
int ggan_conv2d_fwd_cast_ring(const ggan_conv_geom* g, const int32_t* ring, int nslots, const int32_t* ctr_a, const int32_t* ctr_b, int offset,
                              const float* noise /* may be NULL */, float div, float mul, float* x_out, const float* w,
                              const float* bias /* may be NULL */, float* y, int act, float alpha, ggan_stream_t stream);
/* out[B,D] = k[B,K] mu[K,D] + noise[B,D] in one pointwise launch: HyperGenerator of the gmgan scripts, tf.add(tf.matmul(tf.cast(hyper_k,
 * tf.float32), com_mu), hyper_noise) (/root/reference/gmgan_inference_cifar10.py:150-153).  D a multiple of 4, buffers 16-byte aligned. */
int ggan_mix_mean(const float* k, const float* mu, const float* noise, float* out, int B, int K, int D, ggan_stream_t stream);
/* out = a*x + b*y (+c) elementwise (interpolates, residuals). */
int ggan_axpby(const float* x, const float* y, float* out, size_t n, float a, float b, float c,
               ggan_stream_t stream);
/* out[r,:] = x[r,:] + alpha[r]*(y[r,:]-x[r,:])  (gan_inference_cifar10.py:358-361). */
int ggan_row_lerp(const float* x, const float* y, const float* alpha, float* out, int rows, int cols,
                  ggan_stream_t stream);

/* ---- losses ----------------------------------------------------------------------------------
 * loss[0] = weight * mean_i( max(x,0) - x*z + log1p(exp(-|x|)) ), z = label (0 or 1)
 * (tf.nn.sigmoid_cross_entropy_with_logits + reduce_mean, tflib/objs/gan_inference.py:85-101).
 * accumulate != 0 adds into loss[0] instead of overwriting. */
int ggan_bce_logits_fwd(const float* x, float label, float weight, float* loss, int n, int accumulate,
                        ggan_stream_t stream);
/* gx[i] = gloss[0]*weight*(sigmoid(x[i]) - z)/n */
int ggan_bce_logits_bwd(const float* x, float label, float weight, const float* gloss, float* gx, int n,
                        ggan_stream_t stream);
/* every term of one cost in a single launch: loss = sum_i weights[i] * mean(bce(xs[i], labels[i])), terms added in index
 * order (same result as one ggan_bce_logits_fwd per term with accumulate); the backward writes gxs[i] for every term. */
#define GGAN_BCE_MAX 16
int ggan_bce_logits_multi_fwd(const float* const* xs, const float* labels, const float* weights, const int* ns,
                              int count, float* loss, ggan_stream_t stream);
/* ggan_bce_logits_multi_fwd_grad AND the head kernel of ggan_critic_head_bwd in one launch, for a cost whose logits are the output
 * of ONE critic head (the terms xs[i] are consecutive row ranges of its logits[M], in order): loss and the unit-seed gradients gxs as
 * above, plus gh[M,H] = g w_out^T * lrelu'(h), d_wout[H] = h^T g, d_bout = sum g (d_wout / d_bout may be NULL) from the head's kept
 * activation h (gan_inference_cifar10.py:246-254 + tflib/objs/gan_inference.py:104-117).  M <= GGAN_HEAD_BCE_MAX_ROWS.  The products of
 * the head's backward follow with ggan_critic_head_bwd(g = NULL: gh is given). */
#define GGAN_HEAD_BCE_MAX_ROWS 2048
int ggan_bce_head_bwd(const float* const* xs, const float* labels, const float* weights, const int* ns, int count, float* loss,
                      float* const* gxs, int M, int H, const float* h, const float* w_out, float alpha, float* gh, float* d_wout,
                      float* d_bout, ggan_stream_t stream);
/* the same for a cost over the logits of up to GGAN_BCE_HEADS heads (the mixture scripts: joint critic + mixture critic,
 * gmgan_inference_cifar10.py:282-301): head a owns the next head_terms[a] terms; all arrays indexed by head (host arrays). */
#define GGAN_BCE_HEADS 2
int ggan_bce_heads_bwd(const float* const* xs, const float* labels, const float* weights, const int* ns, int count, float* loss,
                       float* const* gxs, int nheads, const int* head_terms, const int* Ms, const int* Hs, const float* const* hs,
                       const float* const* w_outs, const float* alphas, float* const* ghs, float* const* d_wouts /* entries may be NULL */,
                       float* const* d_bouts /* entries may be NULL */, ggan_stream_t stream);
int ggan_bce_logits_multi_bwd(const float* const* xs, const float* labels, const float* weights, const int* ns,
                              int count, const float* gloss, float* const* gxs, ggan_stream_t stream);
/* forward AND the gradients for an upstream gradient of exactly 1 in one launch: gxs[i][j] = weights[i]*(sigmoid(x)-z)/n, bit for
 * bit what ggan_bce_logits_multi_bwd writes for gloss[0] == 1.  The cost of a train op is differentiated with a unit seed
 * (tf.gradients(cost, var_list), gan_inference.py:108-117), so the backward launch of the cost disappears from the step. */
int ggan_bce_logits_multi_fwd_grad(const float* const* xs, const float* labels, const float* weights, const int* ns,
                                   int count, float* loss, float* const* gxs, ggan_stream_t stream);
/* loss[0] (+)= weight*mean(x); bwd gx[i] = gloss[0]*weight/n  (wali_gp, gan_inference.py:29-30). */
int ggan_mean_fwd(const float* x, float weight, float* loss, int n, int accumulate, ggan_stream_t stream);
int ggan_mean_bwd(const float* gloss, float weight, float* gx, int n, ggan_stream_t stream);
/* every term of a Wasserstein cost in one launch: loss = sum_i weights[i]*mean(xs[i]) (terms added in index order, as
 * ggan_mean_fwd with accumulate), and -- gxs non-NULL -- the gradients for an upstream gradient of exactly 1,
 * gxs[i][:] = weights[i]/ns[i] (what ggan_mean_bwd writes for gloss[0] == 1).  count <= GGAN_BCE_MAX. */
int ggan_mean_multi_fwd_grad(const float* const* xs, const float* weights, const int* ns, int count, float* loss,
                             float* const* gxs /* may be NULL */, ggan_stream_t stream);

/* Conv3D of the '3dcnn' sequence critic (tflib/ops/conv3d.py:6-51: tf.nn.conv3d, data NDHWC [N,L,H,W,C], filter [fl,fs,fs,in,out],
 * strides (stride_len, stride, stride), SAME padding; ssgan_inference_moving_mnist.py:352-405).  The filter is already the [K, Co]
 * operand of a GEMM (K = fl*fs*fs*in, ordered (dl,dh,dw,ci)), so the layer is ggan_im2col3d followed by ggan_gemm (bias / activation in
 * its epilogue); the filter gradient is ggan_gemm(col^T, gy) and the data gradient ggan_col2im3d(ggan_gemm(gy, W^T)).
 * dims10 = {N, L, H, W, Ci, Co, fl, fs, stride_len, stride} (host array).
 *   col [N*Lo*Ho*Wo, K]: col[(n,ol,oh,ow)][(dl,dh,dw,ci)] = x[n, ol*sl+dl-pl, oh*s+dh-ph, ow*s+dw-pw, ci], 0 in the padding
 *   ggan_col2im3d is its adjoint (a gather: deterministic). */
int ggan_conv3d_out_shape(const int* dims10, int* lo_ho_wo);
int ggan_im2col3d(const int* dims10, const float* x, float* col, ggan_stream_t stream);
int ggan_col2im3d(const int* dims10, const float* col, float* gx, ggan_stream_t stream);
/* The same three products as implicit GEMMs (no patch matrix: the gathered operand is addressed in place, csrc/conv3d.hip):
 *   ggan_conv3d_fwd    y  = act(conv3d(x, w) + bias)                       (tflib/ops/conv3d.py:33-48)
 *   ggan_conv3d_wgrad  gw = d/dw  for an upstream gradient gy [N,Lo,Ho,Wo,Co] (already multiplied by act'(y))
 *   ggan_conv3d_dgrad  gx = d/dx  (one product per residue class of the input voxels, all classes in one launch)
 * ggan_conv3d_igemm_ok(dims10, kind) (kind 0 fwd, 1 filter grad, 2 data grad) returns 1 when the geometry is covered (Co % 4 == 0,
 * 32-bit sizes; data grad: Ci % 4 == 0, Ci >= 16, strides <= 2); the callers keep ggan_im2col3d / ggan_col2im3d + ggan_gemm for the
 * rest.  Operands 16-byte aligned; ws as for ggan_gemm (split of the reduction when the tile grid is small). */
int ggan_conv3d_igemm_ok(const int* dims10, int kind);
int ggan_conv3d_fwd(const int* dims10, const float* x, const float* w, const float* bias /* may be NULL */, float* y, int act,
                    float alpha, void* ws, size_t ws_bytes, ggan_stream_t stream);
int ggan_conv3d_wgrad(const int* dims10, const float* x, const float* gy, float* gw, void* ws, size_t ws_bytes,
                      ggan_stream_t stream);
int ggan_conv3d_dgrad(const int* dims10, const float* gy, const float* w, float* gx, ggan_stream_t stream);

/* Biased MMD^2 with a mixture of RBF kernels between two sets of codes X[m,d], Y[n,d] (MODE vegan-mmd:
 * tflib/objs/mmd.py:20-71 mix_rbf_mmd2(q_z, p_z, sigmas, wts, biased=True)): k(a,b) = sum_s wt_s exp(-||a-b||^2 / (2 sigma_s^2)),
 * out = mean k(X,X) + mean k(Y,Y) - 2 mean k(X,Y).  sigmas / wts are HOST arrays (wts may be NULL = all 1), ns <= 8, m + n <= 512.
 * row_scratch: m + n floats of device scratch.  Backward: gout = dL/d(out) (device scalar), dX / dY may be NULL. */
int ggan_mix_rbf_mmd2_fwd(const float* X, const float* Y, int m, int n, int d, const float* sigmas, const float* wts, int ns,
                          float* out, float* row_scratch, ggan_stream_t stream);
int ggan_mix_rbf_mmd2_bwd(const float* X, const float* Y, int m, int n, int d, const float* sigmas, const float* wts, int ns,
                          const float* gout, float* dX, float* dY, ggan_stream_t stream);

/* Stochastic encoder head, TYPE_Q = 'learn_std' (gan_inference_cifar10.py:173-188): std = exp(log_std), z = mean + eps * std; the backward
 * takes the gradients arriving at z and at std (either may be NULL). */
int ggan_reparam_fwd(const float* mean, const float* log_std, const float* eps, float* z, float* std_out, size_t n, ggan_stream_t stream);
int ggan_reparam_bwd(const float* gz, const float* gstd, const float* eps, const float* std_in, float* gmean, float* glog_std, size_t n,
                     ggan_stream_t stream);

/* tflib/objs/kl_aggregated.py:46-74 (MODE vegan-kl / vegan-ikl / vegan-jsd): Monte-Carlo divergence between the aggregated posterior -- the
 * equal-weight mixture of the minibatch's nx diagonal Gaussians (mu, sd: [nx, d]) -- and the N(0, I) prior (gan_inference_cifar10.py:269-270).
 * kind 0: KL(q||p) on nz samples of q (k_onehot [nz, nx] one-hot component draws, eps_q [nz, d]; z = k @ mu + (k @ sd) * eps, :6-16);
 * kind 1: KL(p||q) on the nz prior samples z_p [nz, d]; kind 2: JSD on both sets, the mixture m built from the nx components and n_coms
 * copies of the prior term (:31-44).  out: device scalar.  Z [ns, d], A [ns, nx], Bv [ns], T [ns] (ns = nz, or 2 nz for kind 2) are
 * caller-owned buffers the forward fills and the backward reads; W [ns, nx] and GZ [nz, d] are backward scratch.  gout: device scalar. */
int ggan_agg_div_fwd(int kind, const float* mu, const float* sd, const float* k_onehot, const float* eps_q, const float* z_p, int nx, int nz,
                     int d, int n_coms, float* out, float* Z, float* A, float* Bv, float* T, ggan_stream_t stream);
int ggan_agg_div_bwd(int kind, const float* mu, const float* sd, const float* k_onehot, const float* eps_q, int nx, int nz, int d, int n_coms,
                     const float* Z, const float* A, const float* Bv, const float* gout, float* W, float* GZ, float* gmu, float* gsd,
                     ggan_stream_t stream);

/* All the noise of one session.run in one launch: up to GGAN_NOISE_MAX device tensors, each filled with kind 0 = a + b*N(0,1),
 * 1 = uniform [a, b), 2 = one-hot rows of width `widths[i]` with a uniformly drawn index (the prior's k ~ Cat(1/K)).  Replaces
 * tf.random_normal / tf.random_uniform / Categorical.sample + one_hot of the scripts (gmgan_inference_cifar10.py:115-120,344-346;
 * gan_inference_cifar10.py:353-357).  state = {seed, draw number, arrival counter} (3 x uint64 in DEVICE memory, arrival counter
 * zero): Philox4x32-10 keyed by the seed; the draw number is advanced on the device, so a captured HIP graph yields new noise on
 * every replay.  Same seed + draw number => same values on every box. */
#define GGAN_NOISE_MAX 16
int ggan_noise_fill(float* const* dsts, const size_t* sizes, const int* kinds, const float* a, const float* b, const int* widths,
                    int count, uint64_t* state, ggan_stream_t stream);
/* Mixture-of-Gaussians latent glue of the gmgan scripts (HyperExtractor, gmgan_inference_cifar10.py:156-173 with MODE_K =
 * 'CONCRETE'): logits[b,j] = -.5*||z_b - mu_j||^2 + log_pi and k[b,:] = softmax((logits[b,:] + gumbel(u[b,:])) / temp), gumbel(u) =
 * -log(-log(u + 1e-20) + 1e-20) (:117-120).  One launch instead of the dozen [B,K] / [B,K,D] pointwise ops of the TF graph.
 * logits may be NULL (not fetched).  Backward: g_logits / g_k are dL/dlogits, dL/dk (either may be NULL); dz / dmu may be NULL.
 * K, B <= 256. */
int ggan_gmm_latent_fwd(const float* z, const float* mu, const float* gumbel_u, float* logits, float* k, int B, int K, int D,
                        float log_pi, float temp, ggan_stream_t stream);
int ggan_gmm_latent_bwd(const float* z, const float* mu, const float* k, const float* g_logits, const float* g_k, float* dz,
                        float* dmu, int B, int K, int D, float temp, ggan_stream_t stream);

/* reconstruction distances of tflib/utils/distance.py:3-17 (`distance(x, y, 'l1'|'l2')` = reduce_mean(|x-y|^p)), used by the
 * alice / local_epce / vegan objectives as rec_penalty: out[0] (+)= weight * mean(|x-y|^p), p = 1 | 2; the backward writes
 * gx and/or gy (either may be NULL). */
int ggan_dist_fwd(const float* x, const float* y, float* out, size_t n, int p, float weight, int accumulate,
                  ggan_stream_t stream);
int ggan_dist_bwd(const float* x, const float* y, const float* gout, float* gx, float* gy, size_t n, int p,
                  float weight, ggan_stream_t stream);
/* slopes[b] = sqrt(sum_j g[b,j]^2); pen[0] = lam*mean_b((slopes-1)^2)  (gan_inference_cifar10.py:363-364).
 * bwd: gg[b,j] = gpen[0]*lam*2*(slopes[b]-1)/B * g[b,j]/slopes[b]. */
int ggan_gp_penalty_fwd(const float* g, float* slopes, float* pen, int B, int D, float lam,
                        ggan_stream_t stream);
int ggan_gp_penalty_bwd(const float* g, const float* slopes, const float* gpen, float* gg, int B, int D,
                        float lam, ggan_stream_t stream);
/* ggan_gp_penalty_fwd together with ggan_gp_penalty_bwd for a unit upstream gradient (gg_unit = d pen / d g) in one launch -- the
 * penalty enters the critic cost with weight 1 (gan_inference_cifar10.py:365), so that IS its upstream gradient.  arrive: one int32,
 * zero before the first call, left zero (the last workgroup forms the penalty: fixed order, deterministic). */
int ggan_gp_penalty_fwd_grad(const float* g, float* slopes, float* pen, float* gg_unit, int32_t* arrive, int B, int D, float lam,
                             ggan_stream_t stream);

/* ---- optimiser --------------------------------------------------------------------------------
 * tf.train.AdamOptimizer step over a flat parameter buffer (tflib/objs/gan_inference.py:108-117;
 * SURVEY.md A.5): t = *step + 1; m,v EMA; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps).
 * `step` lives in device memory so that a captured graph advances it: the kernel reads it and
 * ggan_adam_advance increments it afterwards.  grad_scale multiplies g first (1/world for DP). */
int ggan_adam_step(float* theta, const float* g, float* m, float* v, size_t n, const int32_t* step,
                   float lr, float beta1, float beta2, float eps, float grad_scale, ggan_stream_t stream);
int ggan_adam_advance(int32_t* step, ggan_stream_t stream);

/* tf.train.RMSPropOptimizer step over a flat buffer (decay 0.9, momentum 0, eps 1e-10 are TF's defaults), followed by the
 * weight clipping of the `wali` objective (tflib/objs/gan_inference.py:4-26: tf.clip_by_value(var, -.01, .01) on the critic);
 * pass clip_lo = -INFINITY, clip_hi = INFINITY for no clipping.  ms is initialised to 1.0 by TF. */
int ggan_rmsprop_step(float* theta, const float* g, float* ms, size_t n, float lr, float decay, float eps,
                      float grad_scale, float clip_lo, float clip_hi, ggan_stream_t stream);
/* Adam step whose counter was already advanced: *step holds THIS update's ordinal t (>= 1), e.g. incremented by the
 * ggan_pack_parts launch of the same optimizer step (its `bump` argument) -- no separate ggan_adam_advance launch. */
int ggan_adam_step_counted(float* theta, const float* g, float* m, float* v, size_t n, const int32_t* step, float lr,
                           float beta1, float beta2, float eps, float grad_scale, ggan_stream_t stream);
/* gather up to GGAN_PACK_MAX scattered tensors into one flat buffer (gradient bucket for RCCL). */
#define GGAN_PACK_MAX 64
int ggan_pack(const float* const* srcs, const size_t* sizes, const size_t* offsets, int count,
              float* flat, ggan_stream_t stream);
/* same, where source i is the sum of parts[i] slabs strides[i] floats apart (summed in slab order: deterministic);
 * bump (may be NULL): an int32 incremented once by this launch. */
int ggan_pack_parts(const float* const* srcs, const size_t* sizes, const size_t* offsets, const int* parts,
                    const size_t* strides, int count, float* flat, int32_t* bump, ggan_stream_t stream);
/* ggan_pack_parts with an optional SECOND contribution per tensor (srcs2[i] may be NULL; parts2 / strides2 as parts / strides):
 * flat[off_i ..] = sum of the slabs of srcs[i] (zeros if NULL) + sum of the slabs of srcs2[i].  A parameter that two passes of one
 * step reach (the critic's main pass and its gradient-penalty pass, gan_inference_cifar10.py:351-366) gets its two gradient
 * contributions summed here instead of by an addition launch per parameter. */
int ggan_pack_parts2(const float* const* srcs, const size_t* sizes, const size_t* offsets, const int* parts, const size_t* strides,
                     const float* const* srcs2, const int* parts2, const size_t* strides2, int count, float* flat, int32_t* bump,
                     ggan_stream_t stream);
/* ggan_pack_parts2 followed by ggan_adam_step_counted in ONE launch (single-replica optimizer steps: nothing happens to the packed
 * gradient between the two; tf.train.AdamOptimizer.minimize = compute_gradients + apply_gradients, gan_inference_cifar10.py:376-380).
 * The summed gradient is still written to `flat`; theta / m / v are the optimizer's flat buffers with the same offsets.  Every
 * workgroup uses step[0] + 1 as the update's ordinal and the last one to finish advances step[0]; `arrive` is
 * GGAN_PACK_ARRIVE_INTS int32 of device memory (arrival counters) that are zero before the call and are left zero.  Same
 * arithmetic, in the same order, as the two launches.  arrive == NULL: the launch applies PART of an update (a subset of the
 * tensors) and leaves step[0] alone -- the launch that completes the update, ordered behind it, passes the counters. */
#define GGAN_PACK_ARRIVE_STRIDE 1024
#define GGAN_PACK_ARRIVE_INTS (33 * GGAN_PACK_ARRIVE_STRIDE)
int ggan_pack_adam(const float* const* srcs, const size_t* sizes, const size_t* offsets, const int* parts, const size_t* strides,
                   const float* const* srcs2 /* may be NULL */, const int* parts2, const size_t* strides2, int count, float* flat,
                   float* theta, float* m, float* v, int32_t* step, int32_t* arrive, float lr, float beta1, float beta2, float eps,
                   float grad_scale, ggan_stream_t stream);

/* ---- per-kernel timing (bench.py roofline leg) -----------------------------------------------
 * When enabled every launch is bracketed by hipEvents on its own stream.  ggan_prof_report
 * synchronises, then writes up to `cap` records -- one per (kernel, work-items per launch, algorithmic flop per launch): a kernel launched
 * on two problem shapes is two records even where the two launches have the same grid, so that per-launch figures are never averages
 * over different problems -- and returns their number.  `flops` / `launches` is the record's flop per launch (its key). */
typedef struct { char name[48]; double total_ms; long launches; double flops; double bytes; long grid; } ggan_prof_rec;   /* one record per (kernel, grid = work-items per launch = rocprofv3's Grid_Size, flop per launch) */
int ggan_prof_enable(int on);
int ggan_prof_reset(void);
int ggan_prof_report(ggan_prof_rec* out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* GGAN_H */
