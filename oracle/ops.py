"""numpy restatement of the arithmetic behind the reference's tflib.ops call sites.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: TensorFlow is absent, these
follow the documented TF semantics restated in SURVEY.md Appendix A.

Every function works in the dtype of its inputs (float64 for parity, float32 for the CPU
baseline timing).  Layouts are the reference's: activations NCHW, Conv2D filters HWIO
[kh,kw,in,out] (tflib/ops/conv2d.py:55-88), Deconv2D filters [kh,kw,out,in]
(tflib/ops/deconv2d.py:63-76), Linear weights [in,out] (tflib/ops/linear.py:108-111).
"""
import numpy as np


# --------------------------------------------------------------------------------------
# TF 'SAME' / 'VALID' padding arithmetic (tf.nn.conv2d as called at tflib/ops/conv2d.py:106-112)
# --------------------------------------------------------------------------------------
def conv_geometry(size, k, stride, padding='SAME'):
    """-> (out_size, pad_before, pad_after).  SAME: out=ceil(size/stride), the extra
    padding goes at the bottom/right."""
    if padding == 'SAME':
        out = -(-size // stride)
        total = max((out - 1) * stride + k - size, 0)
        before = total // 2
        return out, before, total - before
    if padding == 'VALID':
        return (size - k) // stride + 1, 0, 0
    raise Exception('Unsupported configuration')


def _pad_hw(x, pt, pb, pl, pr):
    if pt == pb == pl == pr == 0:
        return x
    return np.pad(x, ((0, 0), (0, 0), (pt, pb), (pl, pr)))


def conv2d(x, w, stride=1, padding='SAME'):
    """Cross-correlation, NCHW, HWIO filter (tflib/ops/conv2d.py:106-112).
    y[n,co,oh,ow] = sum_{ci,kh,kw} x[n,ci,oh*s+kh-pt,ow*s+kw-pl] * w[kh,kw,ci,co]."""
    n, ci, h, wd = x.shape
    kh, kw, ci2, co = w.shape
    assert ci == ci2
    ho, pt, pb = conv_geometry(h, kh, stride, padding)
    wo, pl, pr = conv_geometry(wd, kw, stride, padding)
    xp = _pad_hw(x, pt, pb, pl, pr)
    y = np.zeros((n, ho, wo, co), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, :, i:i + stride * ho:stride, j:j + stride * wo:stride]  # [n,ci,ho,wo]
            y += np.tensordot(patch, w[i, j], axes=([1], [0]))                     # [n,ho,wo,co]
    return np.ascontiguousarray(y.transpose(0, 3, 1, 2))


def conv2d_bwd_data(gy, w, in_hw, stride=1, padding='SAME'):
    """Adjoint of conv2d w.r.t. its input (TF Conv2DBackpropInput).  gy [n,co,ho,wo] ->
    gx [n,ci,H,W] with in_hw=(H,W).  This is also tf.nn.conv2d_transpose as called by
    tflib/ops/deconv2d.py:101-107."""
    n, co, ho, wo = gy.shape
    kh, kw, ci, co2 = w.shape
    assert co == co2
    h, wd = in_hw
    ho2, pt, pb = conv_geometry(h, kh, stride, padding)
    wo2, pl, pr = conv_geometry(wd, kw, stride, padding)
    assert (ho2, wo2) == (ho, wo), 'output_shape inconsistent with input/stride/padding'
    gxp = np.zeros((n, ci, h + pt + pb, wd + pl + pr), dtype=gy.dtype)
    g = gy.transpose(0, 2, 3, 1)                                                  # [n,ho,wo,co]
    for i in range(kh):
        for j in range(kw):
            t = np.tensordot(g, w[i, j], axes=([3], [1]))                          # [n,ho,wo,ci]
            gxp[:, :, i:i + stride * ho:stride, j:j + stride * wo:stride] += t.transpose(0, 3, 1, 2)
    return np.ascontiguousarray(gxp[:, :, pt:pt + h, pl:pl + wd])


def conv2d_bwd_filter(x, gy, ksize, stride=1, padding='SAME'):
    """Adjoint of conv2d w.r.t. its filter (TF Conv2DBackpropFilter) -> HWIO [kh,kw,ci,co]."""
    n, ci, h, wd = x.shape
    _, co, ho, wo = gy.shape
    kh = kw = ksize
    ho2, pt, pb = conv_geometry(h, kh, stride, padding)
    wo2, pl, pr = conv_geometry(wd, kw, stride, padding)
    assert (ho2, wo2) == (ho, wo)
    xp = _pad_hw(x, pt, pb, pl, pr)
    gw = np.zeros((kh, kw, ci, co), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, :, i:i + stride * ho:stride, j:j + stride * wo:stride]
            gw[i, j] = np.tensordot(patch, gy, axes=([0, 2, 3], [0, 2, 3]))
    return gw


def deconv_out_hw(h, wd, k, stride=2, padding='SAME'):
    """output_shape of tflib/ops/deconv2d.py:93-99 (SAME: stride*H; VALID: stride*(H-1)+k)."""
    if padding == 'SAME':
        return stride * h, stride * wd
    return stride * (h - 1) + k, stride * (wd - 1) + k


def deconv2d(x, w, stride=2, padding='SAME'):
    """tflib/ops/deconv2d.py:91-116 without the (mathematically no-op) layout transposes.
    w is [kh,kw,out,in]; read as HWIO of the forward conv out->in it is the same array."""
    oh, ow = deconv_out_hw(x.shape[2], x.shape[3], w.shape[0], stride, padding)
    return conv2d_bwd_data(x, w, (oh, ow), stride, padding)


# --------------------------------------------------------------------------------------
# pointwise / losses
# --------------------------------------------------------------------------------------
def leaky_relu(x, alpha=0.2):
    """tf.maximum(alpha*x, x) (gmgan_inference_cifar10.py:122-123)."""
    return np.maximum(alpha * x, x)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def bce_with_logits(x, z):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log(1+exp(-|x|))."""
    return np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))


def batchnorm_train(x, scale, offset, axes, eps=1e-5):
    """Training-mode BN with batch statistics, biased variance (tflib/ops/batchnorm.py:29-30
    fused NCHW branch for axes [0,2,3]; :74-87 non-fused branch for axes [0])."""
    axes = tuple(axes)
    mean = x.mean(axis=axes, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=axes, keepdims=True)
    shp = [1] * x.ndim
    for a in range(x.ndim):
        if a not in axes:
            shp[a] = x.shape[a]
    return scale.reshape(shp) * (x - mean) / np.sqrt(var + eps) + offset.reshape(shp)


def adam_update(theta, g, m, v, t, lr, beta1, beta2, eps=1e-8):
    """One tf.train.AdamOptimizer step (tflib/objs/gan_inference.py:108-117).
    t is the step count AFTER increment (1 for the first update).  epsilon is added to the
    un-corrected sqrt(v) -- not PyTorch's form.  Returns (theta, m, v)."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    theta = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta, m, v


def conv3d_geometry(n, k, stride):
    """SAME: (out, pad_before)"""
    o = -(-n // stride)
    t = max((o - 1) * stride + k - n, 0)
    return o, t // 2


def conv3d(x, w, stride_len, stride):
    """tf.nn.conv3d, NDHWC, SAME (tflib/ops/conv3d.py:33-39).  x [N,L,H,W,Ci], w [fl,fs,fs,Ci,Co]."""
    N, L, H, W, Ci = x.shape
    fl, fs, _, _, Co = w.shape
    (Lo, pl), (Ho, ph), (Wo, pw) = conv3d_geometry(L, fl, stride_len), conv3d_geometry(H, fs, stride), conv3d_geometry(W, fs, stride)
    xp = np.zeros((N, (Lo - 1) * stride_len + fl, (Ho - 1) * stride + fs, (Wo - 1) * stride + fs, Ci), dtype=x.dtype)
    xp[:, pl:pl + L, ph:ph + H, pw:pw + W] = x[:, :xp.shape[1] - pl, :xp.shape[2] - ph, :xp.shape[3] - pw]
    y = np.zeros((N, Lo, Ho, Wo, Co), dtype=x.dtype)
    for dl in range(fl):
        for dh in range(fs):
            for dw in range(fs):
                patch = xp[:, dl:dl + (Lo - 1) * stride_len + 1:stride_len, dh:dh + (Ho - 1) * stride + 1:stride,
                           dw:dw + (Wo - 1) * stride + 1:stride]
                y += patch @ w[dl, dh, dw]
    return y


def conv3d_bwd_data(gy, w, in_shape, stride_len, stride):
    N, L, H, W, Ci = in_shape
    fl, fs, _, _, Co = w.shape
    _, Lo, Ho, Wo, _ = gy.shape
    pl, ph, pw = conv3d_geometry(L, fl, stride_len)[1], conv3d_geometry(H, fs, stride)[1], conv3d_geometry(W, fs, stride)[1]
    gxp = np.zeros((N, (Lo - 1) * stride_len + fl, (Ho - 1) * stride + fs, (Wo - 1) * stride + fs, Ci), dtype=gy.dtype)
    for dl in range(fl):
        for dh in range(fs):
            for dw in range(fs):
                gxp[:, dl:dl + (Lo - 1) * stride_len + 1:stride_len, dh:dh + (Ho - 1) * stride + 1:stride,
                    dw:dw + (Wo - 1) * stride + 1:stride] += gy @ w[dl, dh, dw].T
    gx = np.zeros(in_shape, dtype=gy.dtype)
    src = gxp[:, pl:pl + L, ph:ph + H, pw:pw + W]
    gx[:, :src.shape[1], :src.shape[2], :src.shape[3]] = src
    return gx


def conv3d_bwd_filter(x, gy, fl, fs, stride_len, stride):
    N, L, H, W, Ci = x.shape
    _, Lo, Ho, Wo, Co = gy.shape
    pl, ph, pw = conv3d_geometry(L, fl, stride_len)[1], conv3d_geometry(H, fs, stride)[1], conv3d_geometry(W, fs, stride)[1]
    xp = np.zeros((N, (Lo - 1) * stride_len + fl, (Ho - 1) * stride + fs, (Wo - 1) * stride + fs, Ci), dtype=x.dtype)
    xp[:, pl:pl + L, ph:ph + H, pw:pw + W] = x[:, :xp.shape[1] - pl, :xp.shape[2] - ph, :xp.shape[3] - pw]
    gw = np.zeros((fl, fs, fs, Ci, Co), dtype=x.dtype)
    for dl in range(fl):
        for dh in range(fs):
            for dw in range(fs):
                patch = xp[:, dl:dl + (Lo - 1) * stride_len + 1:stride_len, dh:dh + (Ho - 1) * stride + 1:stride,
                           dw:dw + (Wo - 1) * stride + 1:stride]
                gw[dl, dh, dw] = np.tensordot(patch, gy, axes=([0, 1, 2, 3], [0, 1, 2, 3]))
    return gw

