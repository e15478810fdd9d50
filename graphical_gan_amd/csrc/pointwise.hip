// HBM-bound pointwise / reduction kernels of the training step: activations, bias, input scaling,
// interpolation, BCE / Wasserstein / gradient-penalty losses, Adam, gradient packing.
// All are grid-stride, float4-vectorised where the layout allows, wave-shuffle reductions.
#include "common.h"
#include "conv.h"
using namespace ggan;

namespace {

constexpr int kBlock = 256;
inline int grid_for(size_t n, int per_thread = 4) {
    size_t b = cdivz(n, (size_t)kBlock * per_thread);
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;   // 256 CUs x 8 blocks, grid-stride the rest
    return (int)b;
}

// ---------------------------------------------------------------------------------------------
__global__ void act_fwd_k(const float* __restrict__ x, float* __restrict__ y, size_t n, int act, float alpha) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    size_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (size_t j = i; j < n4; j += stride) {
        float4 v = x4[j];
        v.x = act_apply(v.x, act, alpha); v.y = act_apply(v.y, act, alpha);
        v.z = act_apply(v.z, act, alpha); v.w = act_apply(v.w, act, alpha);
        y4[j] = v;
    }
    for (size_t j = (n4 << 2) + i; j < n; j += stride) y[j] = act_apply(x[j], act, alpha);
}

__global__ void act_bwd_k(const float* __restrict__ gy, const float* __restrict__ ref, float* __restrict__ gx,
                          size_t n, int act, float alpha) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    size_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(gy);
    const float4* r4 = reinterpret_cast<const float4*>(ref);
    float4* o4 = reinterpret_cast<float4*>(gx);
    for (size_t j = i; j < n4; j += stride) {
        float4 g = g4[j], r = r4[j], o;
        o.x = act_grad(g.x, r.x, act, alpha); o.y = act_grad(g.y, r.y, act, alpha);
        o.z = act_grad(g.z, r.z, act, alpha); o.w = act_grad(g.w, r.w, act, alpha);
        o4[j] = o;
    }
    for (size_t j = (n4 << 2) + i; j < n; j += stride) gx[j] = act_grad(gy[j], ref[j], act, alpha);
}

// act_bwd + the bias gradient of the layer in one pass: workgroup (c, s) handles channel c of the images [s*ipb, (s+1)*ipb),
// writes gx = gy * act'(ref) and its partial sum to parts[s*C + c] (slab s of a "parts" gradient: the pack kernel or a split-K
// reduce adds the slabs in order -- deterministic, no second reduction launch).  Replaces act_bwd + chansum_part + chansum_final
// in the backward of a Deconv2D / Conv2D whose bias does not feed a BatchNorm.
__global__ __launch_bounds__(256) void act_bwd_chansum_k(const float* __restrict__ gy, const float* __restrict__ ref, float* __restrict__ gx,
                                                         float* __restrict__ parts, int N, int C, int HW, int ipb, int act, float alpha) {
    __shared__ float red[4];
    const int c = blockIdx.x, sidx = blockIdx.y, tid = threadIdx.x;
    const int n0 = sidx * ipb, n1 = min(n0 + ipb, N);
    float acc = 0.f;
    const int hw4 = HW >> 2;
    for (int n = n0; n < n1; ++n) {
        const size_t base = ((size_t)n * C + c) * HW;
        if ((HW & 3) == 0) {
            const float4* g4 = reinterpret_cast<const float4*>(gy + base);
            const float4* r4 = reinterpret_cast<const float4*>(ref + base);
            float4* o4 = reinterpret_cast<float4*>(gx + base);
            for (int j = tid; j < hw4; j += 256) {
                const float4 g = g4[j], r = r4[j];
                float4 o;
                o.x = act_grad(g.x, r.x, act, alpha); o.y = act_grad(g.y, r.y, act, alpha);
                o.z = act_grad(g.z, r.z, act, alpha); o.w = act_grad(g.w, r.w, act, alpha);
                o4[j] = o;
                acc += (o.x + o.y) + (o.z + o.w);
            }
        } else {
            for (int j = tid; j < HW; j += 256) {
                const float o = act_grad(gy[base + j], ref[base + j], act, alpha);
                gx[base + j] = o;
                acc += o;
            }
        }
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) parts[(size_t)sidx * C + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void bias_add_k(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ y,
                           size_t total, int C, int HW) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = i; j < total; j += stride) {
        int c = (int)((j / (size_t)HW) % (size_t)C);
        y[j] = x[j] + bias[c];
    }
}

__global__ void cast_scale_k(const int32_t* __restrict__ x, const float* __restrict__ noise, float* __restrict__ y,
                             size_t n, float div, float mul) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = i; j < n; j += stride) {
        // reference op order: 2*((float(x)/255.)-.5)  (gmgan_inference_cifar10.py:342)
        float v = mul * (((float)x[j] / div) - 0.5f);
        if (noise) v += noise[j];
        y[j] = v;
    }
}

__global__ void cast_scale_ring_k(const int32_t* __restrict__ ring, int nslots, const int32_t* __restrict__ ctr_a,
                                  const int32_t* __restrict__ ctr_b, int offset, const float* __restrict__ noise,
                                  float* __restrict__ y, size_t n, float div, float mul) {
    // (nothing in this launch writes the counters: every workgroup sees the same slot)
    long long c = (long long)offset + (ctr_a ? *ctr_a : 0) + (ctr_b ? *ctr_b : 0);
    const int slot = (int)(((c % nslots) + nslots) % nslots);
    const int32_t* x = ring + (size_t)slot * n;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = i; j < n; j += stride) {
        float v = mul * (((float)x[j] / div) - 0.5f);
        if (noise) v += noise[j];
        y[j] = v;
    }
}

__global__ void axpby_k(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                        size_t n, float a, float b, float c) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = i; j < n; j += stride) out[j] = a * x[j] + (y ? b * y[j] : 0.f) + c;
}

// out[b][d] = sum_j k[b][j] * mu[j][d] + noise[b][d]: HyperGenerator of the gmgan scripts, tf.add(tf.matmul(tf.cast(hyper_k, tf.float32), com_mu),
// hyper_noise) (gmgan_inference_cifar10.py:150-153), as ONE pointwise launch instead of a 30-deep GEMM launch and an addition launch at the head of
// the Generator chain.  The fmaf chain in j order is what the MFMA GEMM computes; with one-hot rows it is the selected mean exactly.
__global__ void mix_mean_k(const float* __restrict__ k, const float* __restrict__ mu, const float* __restrict__ noise, float* __restrict__ out,
                           int B, int K, int D4) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D4) return;
    const int b = idx / D4, d4 = idx - b * D4;
    const float* kr = k + (size_t)b * K;
    const float4* m = reinterpret_cast<const float4*>(mu) + d4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < K; ++j) {
        const float kv = kr[j];
        const float4 mv = m[(size_t)j * D4];
        acc.x = fmaf(kv, mv.x, acc.x); acc.y = fmaf(kv, mv.y, acc.y); acc.z = fmaf(kv, mv.z, acc.z); acc.w = fmaf(kv, mv.w, acc.w);
    }
    const float4 nv = reinterpret_cast<const float4*>(noise)[idx];
    acc.x += nv.x; acc.y += nv.y; acc.z += nv.z; acc.w += nv.w;
    reinterpret_cast<float4*>(out)[idx] = acc;
}

__global__ void row_lerp_k(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ alpha,
                           float* __restrict__ out, int rows, int cols) {
    size_t total = (size_t)rows * cols;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t j = i; j < total; j += stride) {
        float a = alpha[j / (size_t)cols];
        out[j] = x[j] + a * (y[j] - x[j]);
    }
}
// (32-bit indices and the row by one v_mul_hi: the 64-bit division above is ~100 instructions per element)
__global__ void row_lerp32_k(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ alpha,
                             float* __restrict__ out, unsigned total, int cols, FastDiv dc) {
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
        const float a = alpha[fdiv(j, dc)];
        out[j] = x[j] + a * (y[j] - x[j]);
    }
}

// ---- reductions -------------------------------------------------------------------------------
// column sums of a [rows, cols] matrix: one thread per column chunk, lanes along columns (coalesced)
__global__ void colsum_k(const float* __restrict__ x, float* __restrict__ out, int rows, int cols) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += x[(size_t)r * cols + c];
    out[c] = s;
}

// tall matrices (the patch-matrix layers of Conv3D: 10^5..10^6 rows): block (column tile of 64, slab p) sums a contiguous range of rows
// with 4 row lanes, the lanes are added through LDS in a fixed order, part[c*P + p]; chansum_final_k adds the slabs in order.
__global__ void __launch_bounds__(256) colsum_part_k(const float* __restrict__ x, float* __restrict__ part, int rows, int cols,
                                                     int rows_per_slab, int P) {
    __shared__ float sm[4][64];
    const int lc = threadIdx.x & 63, lr = threadIdx.x >> 6, c = blockIdx.x * 64 + lc, p = blockIdx.y;
    const int r0 = p * rows_per_slab, r1 = min(rows, r0 + rows_per_slab);
    float s = 0.f;
    if (c < cols)
        for (int r = r0 + lr; r < r1; r += 4) s += x[(size_t)r * cols + c];
    sm[lr][lc] = s;
    __syncthreads();
    if (lr == 0 && c < cols) part[(size_t)c * P + p] = (sm[0][lc] + sm[1][lc]) + (sm[2][lc] + sm[3][lc]);
}

// per-channel sum over (n, hw) of NCHW: block (c, p) sums images p, p+P, ... of channel c (each image-channel plane is HW
// contiguous floats: float4 loads when HW % 4 == 0) into part[c*P + p]; launched with P = 1 (one block per channel).
__global__ void chansum_part_k(const float* __restrict__ x, float* __restrict__ part, int N, int C, int HW, int P) {
    __shared__ float sm[32];
    const int c = blockIdx.x, p = blockIdx.y;
    float s = 0.f;
    if ((HW & 3) == 0) {
        const int hw4 = HW >> 2;
        for (int n = p; n < N; n += P) {
            const float4* pl = reinterpret_cast<const float4*>(x + ((size_t)n * C + c) * HW);
            for (int i = threadIdx.x; i < hw4; i += blockDim.x) {
                const float4 v = pl[i];
                s += (v.x + v.y) + (v.z + v.w);
            }
        }
    } else {
        for (int n = p; n < N; n += P) {
            const float* pl = x + ((size_t)n * C + c) * HW;
            for (int i = threadIdx.x; i < HW; i += blockDim.x) s += pl[i];
        }
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) part[(size_t)c * P + p] = s;
}

// one wave per channel: lane l adds partials l, l+64, ... (coalesced), the 64 lane sums are combined by a fixed butterfly --
// deterministic, and the chain is P/64 long instead of P (one thread per channel took 50-60 us for the ~1000 partials of a tall sum)
__global__ void __launch_bounds__(256) chansum_final_k(const float* __restrict__ part, float* __restrict__ out, int C, int P) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    float s = 0.f;
    for (int p = lane; p < P; p += 64) s += part[(size_t)c * P + p];
    s = wave_sum(s);
    if (lane == 0) out[c] = s;
}

// ---- stochastic encoder head (TYPE_Q = 'learn_std', gan_inference_cifar10.py:173-188): std = exp(log_std), z = mean + eps * std ----------
__global__ void reparam_fwd_k(const float* __restrict__ mean, const float* __restrict__ log_std, const float* __restrict__ eps,
                              float* __restrict__ z, float* __restrict__ sd, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float s = expf(log_std[i]);
        sd[i] = s;
        z[i] = fmaf(eps[i], s, mean[i]);
    }
}

__global__ void reparam_bwd_k(const float* __restrict__ gz, const float* __restrict__ gsd, const float* __restrict__ eps,
                              const float* __restrict__ sd, float* __restrict__ gmean, float* __restrict__ glog, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = gz ? gz[i] : 0.f, b = gsd ? gsd[i] : 0.f;
        gmean[i] = a;
        glog[i] = (a * eps[i] + b) * sd[i];
    }
}

// ---- tflib/objs/kl_aggregated.py: KL / inverse KL / JSD between the aggregated posterior (equal-weight mixture of the minibatch's nx
// diagonal Gaussians) and the N(0, I) prior on nz Monte-Carlo samples.  kind 0 kl (samples from q), 1 ikl (samples from p), 2 jsd
// (both: rows [0, nz) from q, [nz, 2nz) from p).  One block per sample; every reduction in a fixed order.
struct AggP {
    const float *mu, *sd, *k, *eps_q, *z_p;
    int kind, nx, nz, d, n_coms;
};
#define GGAN_LOG2PI 1.8378770664093453f

__device__ __forceinline__ float block_max(float v, float* smem) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float t = smem[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, smem[i]);
    return t;
}

__global__ __launch_bounds__(128) void agg_div_fwd_k(const AggP P, float* __restrict__ Z, float* __restrict__ A, float* __restrict__ Bv,
                                                      float* __restrict__ T) {
    extern __shared__ float zs[];          // d
    __shared__ float red[20];
    const int i = blockIdx.x, nx = P.nx, d = P.d;
    const bool qs = P.kind == 0 || (P.kind == 2 && i < P.nz);
    const int ip = P.kind == 2 ? i - P.nz : i;
    float part = 0.f;
    for (int dd = threadIdx.x; dd < d; dd += blockDim.x) {
        float z;
        if (qs) {                           // mixture_gaussian (:6-16): k @ mu + (k @ std) * eps, k one-hot
            float m = 0.f, sg = 0.f;
            for (int j = 0; j < nx; ++j) {
                const float kk = P.k[(size_t)i * nx + j];
                m = fmaf(kk, P.mu[(size_t)j * d + dd], m);
                sg = fmaf(kk, P.sd[(size_t)j * d + dd], sg);
            }
            z = fmaf(sg, P.eps_q[(size_t)i * d + dd], m);
        } else {
            z = P.z_p[(size_t)ip * d + dd];
        }
        zs[dd] = z;
        Z[(size_t)i * d + dd] = z;
        part += z * z + GGAN_LOG2PI;
    }
    const float b = -0.5f * block_sum(part, red);       // log N(z; 0, I)   (also orders the zs writes before the reads below)
    float lmax = -3.0e38f;
    for (int j = threadIdx.x; j < nx; j += blockDim.x) {
        float acc = 0.f;
        for (int dd = 0; dd < d; ++dd) {
            const float sg = P.sd[(size_t)j * d + dd], r = (zs[dd] - P.mu[(size_t)j * d + dd]) / sg;
            acc += r * r + GGAN_LOG2PI + 2.f * logf(sg);
        }
        const float a = -0.5f * acc;
        A[(size_t)i * nx + j] = a;
        lmax = fmaxf(lmax, a);
    }
    // (as the reference: log q is shifted by ITS row maximum (:26-29), the mixture m by the maximum over its own terms (:41-44) -- a
    // common shift would underflow sum exp(a_j) to 0 wherever the prior term dominates)
    const float mq = block_max(lmax, red);
    float se = 0.f;
    for (int j = threadIdx.x; j < nx; j += blockDim.x) se += expf(A[(size_t)i * nx + j] - mq);
    const float Sq = block_sum(se, red);
    if (threadIdx.x == 0) {
        const float lq = mq + logf(Sq) - logf((float)nx);
        float t;
        if (P.kind == 0) t = lq - b;
        else if (P.kind == 1) t = b - lq;
        else {
            const float mm = fmaxf(mq, b);
            const float lm = mm + logf(Sq * expf(mq - mm) + (float)P.n_coms * expf(b - mm)) - logf((float)(nx + P.n_coms));
            t = qs ? 0.5f * (lq - lm) : 0.5f * (b - lm);
        }
        T[i] = t;
        Bv[i] = b;
    }
}

__global__ void agg_div_final_k(const float* __restrict__ T, int ns, int nz, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < ns; ++i) s += T[i];           // sample order: deterministic
        out[0] = s / (float)nz;
    }
}

// stage A (block per sample): W[i][j] = dL/da_ij, GZ[i][:] = dL/dz_i for the samples drawn from q (they depend on mu / std)
__global__ __launch_bounds__(128) void agg_div_bwd_samples_k(const AggP P, const float* __restrict__ Z, const float* __restrict__ A,
                                                              const float* __restrict__ Bv, const float* __restrict__ gout,
                                                              float* __restrict__ W, float* __restrict__ GZ) {
    extern __shared__ float ws[];          // nx
    __shared__ float red[20];
    const int i = blockIdx.x, nx = P.nx, d = P.d;
    const bool qs = P.kind == 0 || (P.kind == 2 && i < P.nz);
    const float b = Bv[i], gs = gout[0] / (float)P.nz;
    float lmax = -3.0e38f;
    for (int j = threadIdx.x; j < nx; j += blockDim.x) lmax = fmaxf(lmax, A[(size_t)i * nx + j]);
    const float mq = block_max(lmax, red);
    float se = 0.f;
    for (int j = threadIdx.x; j < nx; j += blockDim.x) se += expf(A[(size_t)i * nx + j] - mq);
    const float Sq = block_sum(se, red);
    const float mm = P.kind == 2 ? fmaxf(mq, b) : mq;                       // shift of the mixture m (jsd)
    const float pe = (float)P.n_coms * expf(b - mm), M = Sq * expf(mq - mm) + pe;
    for (int j = threadIdx.x; j < nx; j += blockDim.x) {
        const float a = A[(size_t)i * nx + j], rq = expf(a - mq) / Sq;      // responsibility under q
        float w;
        if (P.kind == 0) w = rq;
        else if (P.kind == 1) w = -rq;
        else {
            const float rm = expf(a - mm) / M;                                // ... under m
            w = qs ? 0.5f * (rq - rm) : -0.5f * rm;
        }
        w *= gs;
        ws[j] = w;
        W[(size_t)i * nx + j] = w;
    }
    __syncthreads();
    if (!qs) return;
    const float v = gs * (P.kind == 0 ? -1.f : -0.5f * pe / M);      // dL/db; db/dz = -z
    for (int dd = threadIdx.x; dd < d; dd += blockDim.x) {
        const float z = Z[(size_t)i * d + dd];
        float acc = -v * z;
        for (int j = 0; j < nx; ++j) {
            const float sg = P.sd[(size_t)j * d + dd];
            acc -= ws[j] * (z - P.mu[(size_t)j * d + dd]) / (sg * sg);
        }
        GZ[(size_t)i * d + dd] = acc;
    }
}

// stage B (block per component j): the direct terms of every sample, then the samples drawn from this component (through z)
__global__ __launch_bounds__(128) void agg_div_bwd_comps_k(const AggP P, int ns, const float* __restrict__ Z, const float* __restrict__ W,
                                                            const float* __restrict__ GZ, float* __restrict__ gmu, float* __restrict__ gsd) {
    const int j = blockIdx.x, nx = P.nx, d = P.d;
    const int nq = P.kind == 1 ? 0 : P.nz;
    for (int dd = threadIdx.x; dd < d; dd += blockDim.x) {
        const float m = P.mu[(size_t)j * d + dd], sg = P.sd[(size_t)j * d + dd], i2 = 1.f / (sg * sg);
        float gm = 0.f, gsg = 0.f;
        for (int i = 0; i < ns; ++i) {
            const float w = W[(size_t)i * nx + j], r = Z[(size_t)i * d + dd] - m;
            gm = fmaf(w, r * i2, gm);
            gsg = fmaf(w, r * r * i2 / sg - 1.f / sg, gsg);
        }
        for (int i = 0; i < nq; ++i) {
            const float kk = P.k[(size_t)i * nx + j];
            if (kk != 0.f) {
                const float gz = kk * GZ[(size_t)i * d + dd];
                gm += gz;
                gsg = fmaf(gz, P.eps_q[(size_t)i * d + dd], gsg);
            }
        }
        gmu[(size_t)j * d + dd] = gm;
        gsd[(size_t)j * d + dd] = gsg;
    }
}

// ---- losses (single block: n is a minibatch of logits) ---------------------------------------------
__global__ void bce_fwd_k(const float* __restrict__ x, float z, float weight, float* __restrict__ loss, int n,
                          int accumulate) {
    __shared__ float sm[32];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float v = x[i];
        s += fmaxf(v, 0.f) - v * z + log1pf(expf(-fabsf(v)));
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) {
        float r = weight * (s / (float)n);
        loss[0] = accumulate ? loss[0] + r : r;
    }
}

__global__ void bce_bwd_k(const float* __restrict__ x, float z, float weight, const float* __restrict__ gloss,
                          float* __restrict__ gx, int n) {
    const float g = gloss[0] * weight / (float)n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float v = x[i];
        float sg = 1.f / (1.f + expf(-v));
        gx[i] = g * (sg - z);
    }
}

// all BCE terms of one cost in ONE launch: loss = sum_i w_i * mean(bce(x_i, z_i))
struct BceTable {
    const float* x[GGAN_BCE_MAX];
    float* gx[GGAN_BCE_MAX];
    float z[GGAN_BCE_MAX];
    float w[GGAN_BCE_MAX];
    int n[GGAN_BCE_MAX];
    int count;
};

__global__ void bce_multi_fwd_k(BceTable t, float* __restrict__ loss) {
    __shared__ float sm[32];
    float tot = 0.f;
    for (int k = 0; k < t.count; ++k) {      // terms in order: the same summation order as one launch per term
        const float* x = t.x[k];
        const float z = t.z[k];
        float s = 0.f;
        float* gx = t.gx[k];                 // optional: the gradient for a unit upstream gradient (bce_multi_bwd_k with gloss == 1)
        const float g = 1.f * t.w[k] / (float)t.n[k];
        for (int i = threadIdx.x; i < t.n[k]; i += blockDim.x) {
            float v = x[i];
            s += fmaxf(v, 0.f) - v * z + log1pf(expf(-fabsf(v)));
            if (gx) gx[i] = g * (1.f / (1.f + expf(-v)) - z);
        }
        s = block_sum(s, sm);
        const float r = t.w[k] * (s / (float)t.n[k]);
        tot = k ? tot + r : r;
    }
    if (threadIdx.x == 0) loss[0] = tot;
}

// bce_multi_fwd_k with unit-seed gradients AND the row-local part of the critic head's backward (head_out_bwd_k, gemm.hip) in one
// launch: the logits of a cost are the output of ONE critic head (terms = consecutive row ranges of its logits), the cost's
// gradient for a unit upstream gradient is row-local, so  gh[r,c] = g[r] * w_out[c] * lrelu'(h[r,c]),  d_wout[c] = sum_r g[r] h[r,c]
// and d_bout = sum_r g[r]  need nothing but the logits.  Workgroup 0 is bce_multi_fwd_k unchanged (loss, g -- same arithmetic, same
// order); workgroups 1.. are head_out_bwd_k with g[r] formed on the fly by the same expression.  Saves the head kernel's launch on
// the critical chain of every BCE step (tail GEMM -> logits -> cost -> head backward -> products).
struct BceHead {
    const float* h;
    const float* w_out;
    float* gh;
    float* d_wout;
    float* d_bout;
    float alpha;
    int M, H;
    int k0, k1;             // its terms [k0, k1) of the table: consecutive row ranges of its logits
    int first;              // its first workgroup
};
struct BceHeads { BceHead hd[GGAN_BCE_HEADS]; int count; };

__global__ __launch_bounds__(256) void bce_head_bwd_k(BceTable t, float* __restrict__ loss, const BceHeads hs) {
    __shared__ float sm[32];
    __shared__ float red[16][16];
    __shared__ float gs_[GGAN_HEAD_BCE_MAX_ROWS];
    if (blockIdx.x == 0) {
        float tot = 0.f;
        for (int k = 0; k < t.count; ++k) {
            const float* x = t.x[k];
            const float z = t.z[k];
            float s = 0.f;
            float* gx = t.gx[k];
            const float g = 1.f * t.w[k] / (float)t.n[k];
            for (int i = threadIdx.x; i < t.n[k]; i += blockDim.x) {
                float v = x[i];
                s += fmaxf(v, 0.f) - v * z + log1pf(expf(-fabsf(v)));
                if (gx) gx[i] = g * (1.f / (1.f + expf(-v)) - z);
            }
            s = block_sum(s, sm);
            const float r = t.w[k] * (s / (float)t.n[k]);
            tot = k ? tot + r : r;
        }
        if (threadIdx.x == 0) loss[0] = tot;
        return;
    }
    const int hi = (hs.count > 1 && (int)blockIdx.x >= hs.hd[1].first) ? 1 : 0;
    const BceHead& hd = hs.hd[hi];
    const float* __restrict__ h = hd.h;
    const float* __restrict__ w_out = hd.w_out;
    float* __restrict__ gh = hd.gh;
    float* __restrict__ d_wout = hd.d_wout;
    float* __restrict__ d_bout = hd.d_bout;
    const float alpha = hd.alpha;
    const int M = hd.M, H = hd.H;
    // g of every row (the head's terms are consecutive row ranges of its logits, in order)
    {
        int r0 = 0;
        for (int k = hd.k0; k < hd.k1; ++k) {
            const float g = 1.f * t.w[k] / (float)t.n[k], z = t.z[k];
            for (int i = threadIdx.x; i < t.n[k]; i += blockDim.x) gs_[r0 + i] = g * (1.f / (1.f + expf(-t.x[k][i])) - z);
            r0 += t.n[k];
        }
    }
    __syncthreads();
    const int tid = threadIdx.x, cl = tid & 15, rg = tid >> 4;          // (the loop of head_out_bwd_k, gemm.hip)
    const int c = ((int)blockIdx.x - hd.first) * 16 + cl;
    float acc = 0.f, gsum = 0.f;
    if (c < H) {
        const float w = w_out[c];
#pragma unroll 8
        for (int r = rg; r < M; r += 16) {
            const float gr = gs_[r], hv = h[(size_t)r * H + c];
            gh[(size_t)r * H + c] = gr * w * (hv > 0.f ? 1.f : alpha);
            acc = fmaf(gr, hv, acc);
            gsum += gr;
        }
    }
    red[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < H && d_wout)
        d_wout[c] = (((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]))) +
                    (((red[8][cl] + red[9][cl]) + (red[10][cl] + red[11][cl])) + ((red[12][cl] + red[13][cl]) + (red[14][cl] + red[15][cl])));
    if ((int)blockIdx.x == hd.first && d_bout) {
        __syncthreads();
        if (cl == 0) red[rg][0] = gsum;
        __syncthreads();
        if (tid == 0) d_bout[0] = (((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])) + ((red[4][0] + red[5][0]) + (red[6][0] + red[7][0]))) +
                                (((red[8][0] + red[9][0]) + (red[10][0] + red[11][0])) + ((red[12][0] + red[13][0]) + (red[14][0] + red[15][0])));
    }
}

__global__ void bce_multi_bwd_k(BceTable t, const float* __restrict__ gloss) {
    const int k = blockIdx.y;
    const int n = t.n[k];
    const float g = gloss[0] * t.w[k] / (float)n, z = t.z[k];
    const float* x = t.x[k];
    float* gx = t.gx[k];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float v = x[i];
        float sg = 1.f / (1.f + expf(-v));
        gx[i] = g * (sg - z);
    }
}

// reconstruction distances of tflib/utils/distance.py: mean(|x-y|^p), p = 1 | 2, one workgroup (n <= a few 100 K)
__global__ void dist_fwd_k(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, size_t n, int p,
                           float weight, int accumulate) {
    __shared__ float sm[32];
    float s = 0.f;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float d = x[i] - y[i];
        s += p == 2 ? d * d : fabsf(d);
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) {
        const float r = weight * (s / (float)n);
        out[0] = accumulate ? out[0] + r : r;
    }
}

__global__ void dist_bwd_k(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gout,
                           float* __restrict__ gx, float* __restrict__ gy, size_t n, int p, float weight) {
    const float g = gout[0] * weight / (float)n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = x[i] - y[i];
        const float v = g * (p == 2 ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)));
        if (gx) gx[i] = v;
        if (gy) gy[i] = -v;
    }
}

__global__ void mean_fwd_k(const float* __restrict__ x, float weight, float* __restrict__ loss, int n, int accumulate) {
    __shared__ float sm[32];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) {
        float r = weight * (s / (float)n);
        loss[0] = accumulate ? loss[0] + r : r;
    }
}

__global__ void mean_multi_fwd_k(BceTable t, float* __restrict__ loss) {
    __shared__ float sm[32];
    float tot = 0.f;
    for (int k = 0; k < t.count; ++k) {      // terms in order: the same summation order as one accumulating launch per term
        const float* x = t.x[k];
        float* gx = t.gx[k];                 // optional: the gradient for a unit upstream gradient (mean_bwd_k with gloss == 1)
        const float g = 1.f * t.w[k] / (float)t.n[k];
        float s = 0.f;
        for (int i = threadIdx.x; i < t.n[k]; i += blockDim.x) {
            s += x[i];
            if (gx) gx[i] = g;
        }
        s = block_sum(s, sm);
        const float r = t.w[k] * (s / (float)t.n[k]);
        tot = k ? tot + r : r;
    }
    if (threadIdx.x == 0) loss[0] = tot;
}

__global__ void mean_bwd_k(const float* __restrict__ gloss, float weight, float* __restrict__ gx, int n) {
    const float g = gloss[0] * weight / (float)n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) gx[i] = g;
}

// one block per sample row: slopes[b] = ||g[b,:]||_2
__global__ void gp_slopes_k(const float* __restrict__ g, float* __restrict__ slopes, int D) {
    __shared__ float sm[32];
    const float* row = g + (size_t)blockIdx.x * D;
    float s = 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) s += row[i] * row[i];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) slopes[blockIdx.x] = sqrtf(s);
}

__global__ void gp_pen_k(const float* __restrict__ slopes, float* __restrict__ pen, int B, float lam) {
    __shared__ float sm[32];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        float d = slopes[i] - 1.f;
        s += d * d;
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) pen[0] = lam * (s / (float)B);
}

// gp_slopes_k + gp_pen_k + gp_bwd_k for a unit upstream gradient in ONE launch: a workgroup per row computes its norm, writes the
// row of d(pen)/dg, and the LAST workgroup to arrive (counter in `arrive`, left zero) forms the penalty from all slopes with the same
// fixed-order block sum as gp_pen_k -- deterministic.
__global__ void gp_fwd_grad_k(const float* __restrict__ g, float* __restrict__ slopes, float* __restrict__ pen, float* __restrict__ gg,
                              int32_t* __restrict__ arrive, int B, int D, float lam) {
    __shared__ float sm[32];
    __shared__ int last;
    const int b = blockIdx.x;
    const float* row = g + (size_t)b * D;
    float s = 0.f;
    for (int i = threadIdx.x; i < D; i += blockDim.x) s += row[i] * row[i];
    s = block_sum(s, sm);
    const float sl = sqrtf(s);
    const float coef = lam * 2.f * (sl - 1.f) / ((float)B * sl);
    float* orow = gg + (size_t)b * D;
    for (int i = threadIdx.x; i < D; i += blockDim.x) orow[i] = coef * row[i];
    if (threadIdx.x == 0) {
        __hip_atomic_store(slopes + b, sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(arrive, 1) == B - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    float t = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const float d = __hip_atomic_load(slopes + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1.f;
        t += d * d;
    }
    t = block_sum(t, sm);
    if (threadIdx.x == 0) {
        pen[0] = lam * (t / (float)B);
        arrive[0] = 0;
    }
}

__global__ void gp_bwd_k(const float* __restrict__ g, const float* __restrict__ slopes, const float* __restrict__ gpen,
                         float* __restrict__ gg, int B, int D, float lam) {
    const int b = blockIdx.x;
    const float s = slopes[b];
    const float coef = gpen[0] * lam * 2.f * (s - 1.f) / ((float)B * s);
    const float* row = g + (size_t)b * D;
    float* orow = gg + (size_t)b * D;
    for (int i = threadIdx.x; i < D; i += blockDim.x) orow[i] = coef * row[i];
}

// ---- Adam (TF flavour) ------------------------------------------------------------------------
// PRE: step[0] already holds this update's ordinal (the gradient-pack kernel of the same optimizer step incremented it)
template <bool PRE>
__global__ void adam_k(float* __restrict__ theta, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, size_t n, const int32_t* __restrict__ step, float lr, float b1,
                       float b2, float eps, float gscale) {
    const float t = (float)(step[0] + (PRE ? 0 : 1));
    const float lr_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    size_t n4 = n >> 2;
    float4* th4 = reinterpret_cast<float4*>(theta);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
#define ADAM1(TH, G, M, V)                                     \
    {                                                          \
        float gg = (G) * gscale;                               \
        (M) = b1 * (M) + (1.f - b1) * gg;                      \
        (V) = b2 * (V) + (1.f - b2) * gg * gg;                 \
        (TH) = (TH) - lr_t * (M) / (sqrtf(V) + eps);           \
    }
    for (size_t j = i; j < n4; j += stride) {
        float4 th = th4[j], gg4 = g4[j], mm = m4[j], vv = v4[j];
        ADAM1(th.x, gg4.x, mm.x, vv.x) ADAM1(th.y, gg4.y, mm.y, vv.y)
        ADAM1(th.z, gg4.z, mm.z, vv.z) ADAM1(th.w, gg4.w, mm.w, vv.w)
        th4[j] = th; m4[j] = mm; v4[j] = vv;
    }
    for (size_t j = (n4 << 2) + i; j < n; j += stride) {
        float th = theta[j], mm = m[j], vv = v[j];
        ADAM1(th, g[j], mm, vv)
        theta[j] = th; m[j] = mm; v[j] = vv;
    }
#undef ADAM1
}

__global__ void adam_advance_k(int32_t* step) { step[0] += 1; }

// tf.train.RMSPropOptimizer (momentum 0, centered=False): ms = decay*ms + (1-decay)*g^2; theta -= lr*g/sqrt(ms+eps);
// then (optionally) the weight clipping of the original WGAN, tf.clip_by_value(var, lo, hi)
__global__ void rmsprop_k(float* __restrict__ theta, const float* __restrict__ g, float* __restrict__ ms, size_t n, float lr,
                          float decay, float eps, float gscale, float lo, float hi) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gg = g[i] * gscale;
        const float m = decay * ms[i] + (1.f - decay) * gg * gg;
        ms[i] = m;
        theta[i] = fminf(fmaxf(theta[i] - lr * gg / sqrtf(m + eps), lo), hi);
    }
}

struct PackTable {
    const float* src[GGAN_PACK_MAX];
    size_t size[GGAN_PACK_MAX];
    size_t off[GGAN_PACK_MAX];
    size_t pstride[GGAN_PACK_MAX];   // floats between the partial slabs of source k
    int parts[GGAN_PACK_MAX];        // number of slabs to sum (1 = plain copy)
    const float* src2[GGAN_PACK_MAX];  // optional SECOND gradient contribution of the same tensor (a parameter used by two passes of a
    size_t pstride2[GGAN_PACK_MAX];    // step, e.g. the critic's main pass and its gradient-penalty pass), added after the first
    int parts2[GGAN_PACK_MAX];
    int count;
    int32_t* bump;                   // optional: counter incremented once per launch (the optimizer's step ordinal)
    int first[GGAN_PACK_MAX + 1];    // pack_adam_k: first workgroup of tensor k (1-D grid of kPackChunk-float chunks)
};
constexpr int kPackChunk = 4096;
__host__ __device__ inline int pack_chunk(int slabs) { return slabs > 4 ? 1024 : kPackChunk; }

// blockIdx.y = tensor, blockIdx.x grid-strides inside it (16-byte accesses when every side is aligned); a source made of
// several split-K slabs is summed in slab order on the way
__global__ void pack_k(PackTable t, float* __restrict__ flat) {
    const int k = blockIdx.y;
    const float* s = t.src[k];
    const float* s2 = t.src2[k];
    float* d = flat + t.off[k];
    const size_t n = t.size[k];
    const int np = t.parts[k], np2 = s2 ? t.parts2[k] : 0;
    const size_t ps = t.pstride[k], ps2 = t.pstride2[k];
    const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    if (t.bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) t.bump[0] += 1;
    if (s && ((((uintptr_t)s) | ((uintptr_t)d) | ((uintptr_t)s2)) & 15) == 0 && (np == 1 || (ps & 3) == 0) && (np2 <= 1 || (ps2 & 3) == 0)) {
        const size_t n4 = n >> 2;
        float4* d4 = reinterpret_cast<float4*>(d);
        for (size_t i = i0; i < n4; i += stride) {
            float4 a = reinterpret_cast<const float4*>(s)[i];
#pragma unroll 8
            for (int p = 1; p < np; ++p) {      // (unrolled: the slab loads are independent, only the adds are ordered)
                const float4 b = reinterpret_cast<const float4*>(s + (size_t)p * ps)[i];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
#pragma unroll 8
            for (int p = 0; p < np2; ++p) {
                const float4 b = reinterpret_cast<const float4*>(s2 + (size_t)p * ps2)[i];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            d4[i] = a;
        }
        for (size_t i = (n4 << 2) + i0; i < n; i += stride) {
            float a = s[i];
            for (int p = 1; p < np; ++p) a += s[(size_t)p * ps + i];
            for (int p = 0; p < np2; ++p) a += s2[(size_t)p * ps2 + i];
            d[i] = a;
        }
    } else {
        for (size_t i = i0; i < n; i += stride) {
            float a = 0.f;
            if (s) {
                a = s[i];
                for (int p = 1; p < np; ++p) a += s[(size_t)p * ps + i];
            }
            for (int p = 0; p < np2; ++p) a += s2[(size_t)p * ps2 + i];
            d[i] = a;
        }
    }
}

// pack_k and adam_k<true> in one pass (single-replica steps: nothing happens to the packed gradient between the two): a
// parameter's gradient is summed from its sources exactly as pack_k does, written to the flat gradient buffer (it stays
// inspectable) and applied at once -- the flat gradient is not read back, and one launch of the step's tail is gone.
// The update's ordinal is step[0] + 1 for every workgroup; the LAST workgroup to finish advances step[0] (arrival counter, left
// at zero), so no workgroup can see the advanced value.
__global__ void pack_adam_k(PackTable t, float* __restrict__ flat, float* __restrict__ theta, float* __restrict__ m,
                            float* __restrict__ v, int32_t* __restrict__ arrive, float lr, float b1, float b2, float eps,
                            float gscale) {
    // 1-D grid: workgroups [first[k], first[k+1]) own 4096-float chunks of tensor k (only workgroups with work exist: every one of
    // them pays a same-address atomic at the end, and those are served one at a time)
    int k = 0;
    while (k + 1 < t.count && (int)blockIdx.x >= t.first[k + 1]) ++k;
    const float* s = t.src[k];
    const float* s2 = t.src2[k];
    const size_t off = t.off[k];
    float* d = flat + off;
    float* th = theta + off;
    float* mp = m + off;
    float* vp = v + off;
    const size_t n = t.size[k];
    const int np = t.parts[k], np2 = s2 ? t.parts2[k] : 0;
    const size_t ps = t.pstride[k], ps2 = t.pstride2[k];
    // (a tensor summed from many slabs gets one float4 per thread: its loads are the long chain)
    const int chunk = pack_chunk(t.parts[k] + (s2 ? t.parts2[k] : 0));
    const size_t base = (size_t)((int)blockIdx.x - t.first[k]) * chunk;
    const float tt = (float)(t.bump[0] + 1);
    const float lr_t = lr * sqrtf(1.f - powf(b2, tt)) / (1.f - powf(b1, tt));
#define ADAM1(TH, G, M, V)                                     \
    {                                                          \
        float gg = (G) * gscale;                               \
        (M) = b1 * (M) + (1.f - b1) * gg;                      \
        (V) = b2 * (V) + (1.f - b2) * gg * gg;                 \
        (TH) = (TH) - lr_t * (M) / (sqrtf(V) + eps);           \
    }
    auto one = [&](size_t i) {
        float a = 0.f;
        if (s) {
            a = s[i];
            for (int p = 1; p < np; ++p) a += s[(size_t)p * ps + i];
        }
        for (int p = 0; p < np2; ++p) a += s2[(size_t)p * ps2 + i];
        d[i] = a;
        float x = th[i], mm = mp[i], vv = vp[i];
        ADAM1(x, a, mm, vv)
        th[i] = x; mp[i] = mm; vp[i] = vv;
    };
    if (s && ((((uintptr_t)s) | ((uintptr_t)d) | ((uintptr_t)s2)) & 15) == 0 && (np == 1 || (ps & 3) == 0) && (np2 <= 1 || (ps2 & 3) == 0)) {
        const size_t n4 = n >> 2;
        // every load of the chunk first (the four float4 groups of a thread, 16 independent loads in flight), then the arithmetic
        // and the stores: written as "load, update, store" per group the stores of one group ordered the loads of the next behind
        // them (the buffers may alias as far as the compiler knows) -- four memory round trips per thread instead of one
        constexpr int NQ = kPackChunk / 4 / kBlock;
        float4 X[NQ], Mq[NQ], Vq[NQ], Aq[NQ];
        bool ok[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const size_t i = base / 4 + threadIdx.x + (size_t)q * kBlock;
            ok[q] = i < n4 && q * kBlock * 4 < chunk;
            if (ok[q]) {
                X[q] = reinterpret_cast<const float4*>(th)[i]; Mq[q] = reinterpret_cast<const float4*>(mp)[i];
                Vq[q] = reinterpret_cast<const float4*>(vp)[i]; Aq[q] = reinterpret_cast<const float4*>(s)[i];
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (!ok[q]) continue;
            const size_t i = base / 4 + threadIdx.x + (size_t)q * kBlock;
            float4 x = X[q], mm = Mq[q], vv = Vq[q], a = Aq[q];
#pragma unroll 8
            for (int p = 1; p < np; ++p) {
                const float4 b = reinterpret_cast<const float4*>(s + (size_t)p * ps)[i];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
#pragma unroll 8
            for (int p = 0; p < np2; ++p) {
                const float4 b = reinterpret_cast<const float4*>(s2 + (size_t)p * ps2)[i];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            reinterpret_cast<float4*>(d)[i] = a;
            ADAM1(x.x, a.x, mm.x, vv.x) ADAM1(x.y, a.y, mm.y, vv.y)
            ADAM1(x.z, a.z, mm.z, vv.z) ADAM1(x.w, a.w, mm.w, vv.w)
            reinterpret_cast<float4*>(th)[i] = x; reinterpret_cast<float4*>(mp)[i] = mm; reinterpret_cast<float4*>(vp)[i] = vv;
        }
        // (the 1..3 elements behind the last whole float4: the workgroup that owns the end of the tensor)
        if (base + chunk >= n && (n4 << 2) + threadIdx.x < n) one((n4 << 2) + threadIdx.x);
    } else {
        for (int q = 0; q < kPackChunk / kBlock; ++q) {
            const size_t i = base + threadIdx.x + (size_t)q * kBlock;
            if (i >= n || q * kBlock >= chunk) break;
            one(i);
        }
    }
#undef ADAM1
    // arrival in two levels (same-address atomics are served one at a time, ~13 ns each, and the workgroups of this one-round
    // launch all finish together): 32 group counters 4 KB apart, the last workgroup of each group reports to counter 0
    // (arrive == NULL: this launch applies PART of an update -- the ordinal stays step[0] + 1 for the launch that completes it)
    if (!arrive) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int grp = blockIdx.x & 31, members = ((int)gridDim.x - grp + 31) >> 5, groups = gridDim.x < 32 ? (int)gridDim.x : 32;
        int32_t* c = arrive + (size_t)(1 + grp) * GGAN_PACK_ARRIVE_STRIDE;
        if (atomicAdd(c, 1) == members - 1) {
            c[0] = 0;
            __threadfence();
            if (atomicAdd(arrive, 1) == groups - 1) {
                arrive[0] = 0;
                t.bump[0] += 1;
            }
        }
    }
}


inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }


// ---- mixture-of-Gaussians latent glue of the gmgan scripts (gmgan_inference_cifar10.py:156-173, MODE_K = 'CONCRETE'):
//        logits[b, j] = -.5 * sum_d (z[b, d] - mu[j, d])^2 + log_pi
//        k[b, :]      = softmax((logits[b, :] - log(-log(u[b, :] + 1e-20) + 1e-20)) / temp)          (Gumbel-softmax relaxation)
//      a dozen [B, K] / [B, K, D] pointwise launches in the TF graph, one here: a workgroup per batch row, a wave per
//      component for the distance (lanes over d, shuffle reduce), the softmax over the K values in LDS. ----
constexpr int kGmmMaxK = 256;

__global__ __launch_bounds__(256) void gmm_latent_fwd_k(const float* __restrict__ z, const float* __restrict__ mu,
                                                        const float* __restrict__ u, float* __restrict__ logits, float* __restrict__ k,
                                                        int K, int D, float log_pi, float inv_temp) {
    __shared__ float sv[kGmmMaxK];
    __shared__ float red[32];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* zb = z + (size_t)b * D;
    for (int j = wave; j < K; j += 4) {
        const float* mj = mu + (size_t)j * D;
        float s = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float t = zb[d] - mj[d];
            s = fmaf(t, t, s);
        }
        s = wave_sum(s);
        if (lane == 0) {
            const float lg = -0.5f * s + log_pi;
            if (logits) logits[(size_t)b * K + j] = lg;
            const float g = -logf(-logf(u[(size_t)b * K + j] + 1e-20f) + 1e-20f);
            sv[j] = (lg + g) * inv_temp;
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = threadIdx.x; j < K; j += 256) m = fmaxf(m, sv[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float e = 0.f;
    for (int j = threadIdx.x; j < K; j += 256) {
        const float t = expf(sv[j] - m);
        sv[j] = t;
        e += t;
    }
    const float tot = block_sum(e, red + 8);
    const float inv = 1.f / tot;
    for (int j = threadIdx.x; j < K; j += 256) k[(size_t)b * K + j] = sv[j] * inv;
}

// Backward in ONE launch: blocks [0, B) own a batch row (-> dz), blocks [B, B + K) own a component (-> dmu, a sum over the batch
// in row order: deterministic).  dlog[b, j] = gl[b, j] + k[b, j] * (gk[b, j] - sum_i gk[b, i] * k[b, i]) / temp.
__global__ __launch_bounds__(256) void gmm_latent_bwd_k(const float* __restrict__ z, const float* __restrict__ mu,
                                                        const float* __restrict__ kk, const float* __restrict__ gl,
                                                        const float* __restrict__ gk, float* __restrict__ dz, float* __restrict__ dmu,
                                                        int B, int K, int D, float inv_temp) {
    __shared__ float sv[kGmmMaxK];
    __shared__ float red[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x < B) {
        const int b = blockIdx.x;
        float dot = 0.f;
        if (gk)
            for (int j = threadIdx.x; j < K; j += 256) dot += gk[(size_t)b * K + j] * kk[(size_t)b * K + j];
        dot = block_sum(dot, red);
        for (int j = threadIdx.x; j < K; j += 256) {
            float v = gl ? gl[(size_t)b * K + j] : 0.f;
            if (gk) v += kk[(size_t)b * K + j] * (gk[(size_t)b * K + j] - dot) * inv_temp;
            sv[j] = v;
        }
        __syncthreads();
        if (dz) {
            // dz[b, d] = -sum_j dlog[j] * (z[b, d] - mu[j, d])
            for (int d = threadIdx.x; d < D; d += 256) {
                const float zv = z[(size_t)b * D + d];
                float acc = 0.f;
                for (int j = 0; j < K; ++j) acc = fmaf(sv[j], zv - mu[(size_t)j * D + d], acc);
                dz[(size_t)b * D + d] = -acc;
            }
        }
        return;
    }
    if (!dmu) return;
    const int j = blockIdx.x - B;
    // dlog[b, j] for every b (one wave per row: the softmax-backward dot product is a K-long reduction), then
    // dmu[j, d] = sum_b dlog[b, j] * (z[b, d] - mu[j, d])
    float* col = sv;                       // B <= kGmmMaxK values
    for (int b = wave; b < B; b += 4) {
        float dot = 0.f;
        if (gk)
            for (int i = lane; i < K; i += 64) dot += gk[(size_t)b * K + i] * kk[(size_t)b * K + i];
        dot = wave_sum(dot);
        if (lane == 0) {
            float v = gl ? gl[(size_t)b * K + j] : 0.f;
            if (gk) v += kk[(size_t)b * K + j] * (gk[(size_t)b * K + j] - dot) * inv_temp;
            col[b] = v;
        }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) {
        const float mv = mu[(size_t)j * D + d];
        float acc = 0.f;
        for (int b = 0; b < B; ++b) acc = fmaf(col[b], z[(size_t)b * D + d] - mv, acc);
        dmu[(size_t)j * D + d] = acc;
    }
}


// ---- all the noise of one session.run in ONE launch (the TF graph draws p_z ~ N(0,1), k ~ Cat(1/K) as a one-hot, Gumbel U, the GP
//      alpha and the dequantisation noise with separate random ops: gmgan_inference_cifar10.py:115-120,344-346).  Counter-based
//      generator (Philox4x32-10, key = seed, counter = (element group, tensor, draw number)); the draw number lives in device
//      memory and is advanced by the last workgroup to arrive, so a captured graph produces fresh noise on every replay. ----
struct NoiseTable {
    float* dst[GGAN_NOISE_MAX];
    unsigned n[GGAN_NOISE_MAX];
    int kind[GGAN_NOISE_MAX];      // 0 normal(a, b) = a + b*N(0,1); 1 uniform [a, b); 2 one-hot rows of width K (n = rows * K)
    float a[GGAN_NOISE_MAX], b[GGAN_NOISE_MAX];
    int K[GGAN_NOISE_MAX];
    int slot[GGAN_NOISE_MAX];      // the tensor's ordinal within the step's draw (part of the counter): its index
    int step[GGAN_NOISE_MAX];      // draw number = state + step (always 0: one session.run per launch)
    int count, advance;
};

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0, 1)

__global__ __launch_bounds__(256) void noise_fill_k(const NoiseTable t, unsigned long long* __restrict__ state) {
    const int tb = blockIdx.y, ti = t.slot[tb];
    const unsigned long long seed = state[0], draw0 = state[1], draw = draw0 + (unsigned long long)t.step[tb];
    const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    float* dst = t.dst[tb];
    const unsigned n = t.n[tb];
    const int kind = t.kind[tb];
    for (unsigned g = blockIdx.x * 256 + threadIdx.x; g * 4 < n; g += gridDim.x * 256) {
        unsigned r[4];
        float v[4];
        if (kind == 2) {
            // element e of a one-hot row: the row's component index comes from a draw keyed by the ROW
            const int K = t.K[tb];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned e = g * 4 + q, row = e / (unsigned)K, col = e - row * (unsigned)K;
                philox4x32_10(row, (unsigned)ti | 0x80000000u, (unsigned)draw, (unsigned)(draw >> 32), k0, k1, r);
                const unsigned idx = min((unsigned)(u01(r[0]) * (float)K), (unsigned)K - 1u);
                v[q] = col == idx ? 1.f : 0.f;
            }
        } else {
            philox4x32_10(g, (unsigned)ti, (unsigned)draw, (unsigned)(draw >> 32), k0, k1, r);
            if (kind == 0) {            // Box-Muller, two pairs
                const float r0 = sqrtf(-2.f * logf(u01(r[0]))), r1 = sqrtf(-2.f * logf(u01(r[2])));
                float s0, c0, s1, c1;
                sincosf(6.28318530718f * u01(r[1]), &s0, &c0);
                sincosf(6.28318530718f * u01(r[3]), &s1, &c1);
                v[0] = t.a[tb] + t.b[tb] * r0 * c0; v[1] = t.a[tb] + t.b[tb] * r0 * s0;
                v[2] = t.a[tb] + t.b[tb] * r1 * c1; v[3] = t.a[tb] + t.b[tb] * r1 * s1;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = t.a[tb] + (t.b[tb] - t.a[tb]) * u01(r[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (g * 4 + q < n) dst[g * 4 + q] = v[q];
    }
    // every workgroup has read `draw` before it arrives here; the last one advances it.  (No fence: nothing is PUBLISHED to the
    // other workgroups -- the arrival count only orders "all have read draw" before "draw is overwritten", and every thread's use
    // of `draw` precedes its workgroup's barrier.  A __threadfence() per workgroup cost 16 us on the 512-workgroup launches.)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long total = (unsigned long long)gridDim.x * gridDim.y;
        const unsigned long long prev = atomicAdd(&state[2], 1ull);
        if (prev == total - 1) {
            state[1] = draw0 + (unsigned long long)t.advance;
            state[2] = 0;
        }
    }
}


// ---- biased MMD^2 with a mixture of RBF kernels between two sets of codes (MODE vegan-mmd: tflib/objs/mmd.py:20-71,
//      mix_rbf_mmd2(q_z, p_z), sigmas 2..80): k(a, b) = sum_s wt_s exp(-gamma_s ||a-b||^2), gamma_s = 1/(2 sigma_s^2),
//      mmd2 = mean k(X,X) + mean k(Y,Y) - 2 mean k(X,Y).  The TF graph builds three Gram matrices and 18 exp maps; here the
//      pairwise distances are formed on the fly.  One workgroup per row of Z = [X; Y] computes that row's weighted kernel sum
//      (forward: into a per-row partial, summed in row order by the last stage) or its gradient row (backward). ----
constexpr int kMmdMaxSigmas = 8;
struct MmdParams {
    const float* X;
    const float* Y;
    int m, n, d, ns;
    float gamma[kMmdMaxSigmas], wt[kMmdMaxSigmas];
};

__device__ __forceinline__ const float* mmd_row(const MmdParams& P, int r) { return r < P.m ? P.X + (size_t)r * P.d : P.Y + (size_t)(r - P.m) * P.d; }

// coefficient of k(z_r, z_j) in mmd2 seen from row r: 1/m^2 (both in X), 1/n^2 (both in Y), -1/(mn) (mixed: each unordered mixed
// pair is visited from both sides, which makes the -2/(mn) of the definition)
__device__ __forceinline__ float mmd_coef(const MmdParams& P, int r, int j) {
    const bool rx = r < P.m, jx = j < P.m;
    if (rx && jx) return 1.f / ((float)P.m * (float)P.m);
    if (!rx && !jx) return 1.f / ((float)P.n * (float)P.n);
    return -1.f / ((float)P.m * (float)P.n);
}

__global__ __launch_bounds__(256) void mmd2_rows_k(const MmdParams P, float* __restrict__ partial) {
    __shared__ float sm[32];
    const int r = blockIdx.x, tot = P.m + P.n;
    const float* zr = mmd_row(P, r);
    float acc = 0.f;
    for (int j = threadIdx.x; j < tot; j += 256) {
        const float* zj = mmd_row(P, j);
        float dist = 0.f;
        for (int k = 0; k < P.d; ++k) {
            const float t = zr[k] - zj[k];
            dist = fmaf(t, t, dist);
        }
        float kv = 0.f;
        for (int s2 = 0; s2 < P.ns; ++s2) kv += P.wt[s2] * expf(-P.gamma[s2] * dist);
        acc += mmd_coef(P, r, j) * kv;
    }
    const float t = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[r] = t;
}

__global__ __launch_bounds__(256) void mmd2_final_k(const float* __restrict__ partial, int tot, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s2 = 0.f;
        for (int r = 0; r < tot; ++r) s2 += partial[r];          // row order: deterministic
        out[0] = s2;
    }
}

// gradient row r: sum_j 2 * coef(r, j) * (-2 G_rj) (z_r - z_j) with G_rj = sum_s wt_s gamma_s exp(-gamma_s D_rj)
// (the factor 2: the pair (r, j) enters the double sum as (r, j) and as (j, r))
__global__ __launch_bounds__(256) void mmd2_bwd_k(const MmdParams P, const float* __restrict__ gout, float* __restrict__ dX,
                                                  float* __restrict__ dY) {
    __shared__ float cf[512];
    const int r = blockIdx.x, tot = P.m + P.n;
    const float* zr = mmd_row(P, r);
    const float go = gout[0];
    for (int j = threadIdx.x; j < tot; j += 256) {
        const float* zj = mmd_row(P, j);
        float dist = 0.f;
        for (int k = 0; k < P.d; ++k) {
            const float t = zr[k] - zj[k];
            dist = fmaf(t, t, dist);
        }
        float gv = 0.f;
        for (int s2 = 0; s2 < P.ns; ++s2) gv += P.wt[s2] * P.gamma[s2] * expf(-P.gamma[s2] * dist);
        cf[j] = -4.f * mmd_coef(P, r, j) * gv * go;
    }
    __syncthreads();
    float* dst = r < P.m ? (dX ? dX + (size_t)r * P.d : nullptr) : (dY ? dY + (size_t)(r - P.m) * P.d : nullptr);
    if (!dst) return;
    for (int k = threadIdx.x; k < P.d; k += 256) {
        const float zv = zr[k];
        float acc = 0.f;
        for (int j = 0; j < tot; ++j) acc = fmaf(cf[j], zv - mmd_row(P, j)[k], acc);
        dst[k] = acc;
    }
}

}  // namespace

extern "C" {

int ggan_act_fwd(const float* x, float* y, size_t n, int act, float alpha, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && y, "null pointer");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    GGAN_CHECK_ARG(aligned16(x) && aligned16(y), "buffers must be 16-byte aligned");
    GGAN_LAUNCH("act_fwd", 0, 8.0 * n, act_fwd_k, dim3(grid_for(n, 16)), dim3(kBlock), 0, s, x, y, n, act, alpha);
    return 0;
}

int ggan_act_bwd(const float* gy, const float* ref, float* gx, size_t n, int act, float alpha, ggan_stream_t stream) {
    GGAN_CHECK_ARG(gy && ref && gx, "null pointer");
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    GGAN_CHECK_ARG(aligned16(gy) && aligned16(ref) && aligned16(gx), "buffers must be 16-byte aligned");
    GGAN_LAUNCH("act_bwd", 0, 12.0 * n, act_bwd_k, dim3(grid_for(n, 16)), dim3(kBlock), 0, s, gy, ref, gx, n, act, alpha);
    return 0;
}

int ggan_act_bwd_chansum(const float* gy, const float* ref, float* gx, float* parts, int parts_cap, int* n_parts, int N, int C, int HW,
                         int act, float alpha, ggan_stream_t stream) {
    GGAN_CHECK_ARG(gy && ref && gx && parts && n_parts, "null pointer");
    GGAN_CHECK_ARG(N > 0 && C > 0 && HW > 0 && parts_cap >= C, "bad shape");
    GGAN_CHECK_ARG(aligned16(gy) && aligned16(ref) && aligned16(gx), "buffers must be 16-byte aligned");
    // enough workgroups to cover the chip, at most one slab per image and at most parts_cap / C slabs
    int S = (512 + C - 1) / C;
    if (S > N) S = N;
    if (S > parts_cap / C) S = parts_cap / C;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    const int ipb = (N + S - 1) / S;
    S = (N + ipb - 1) / ipb;
    *n_parts = S;
    GGAN_LAUNCH("act_bwd_chansum", 0, 12.0 * N * C * HW, act_bwd_chansum_k, dim3(C, S), dim3(256), 0, (hipStream_t)stream, gy, ref, gx, parts,
                N, C, HW, ipb, act, alpha);
    return 0;
}

int ggan_bias_add(const float* x, const float* bias, float* y, int N, int C, int HW, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && bias && y && N > 0 && C > 0 && HW > 0, "bad argument");
    size_t total = (size_t)N * C * HW;
    GGAN_LAUNCH("bias_add", 0, 8.0 * total, bias_add_k, dim3(grid_for(total)), dim3(kBlock), 0, (hipStream_t)stream, x, bias, y, total, C, HW);
    return 0;
}

int ggan_cast_scale_i32(const int32_t* x, const float* noise, float* y, size_t n, float div, float mul, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && y, "null pointer");
    if (n == 0) return 0;
    GGAN_LAUNCH("cast_scale_i32", 0, 8.0 * n, cast_scale_k, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, x, noise, y, n, div, mul);
    return 0;
}

int ggan_cast_scale_ring_i32(const int32_t* ring, int nslots, const int32_t* ctr_a, const int32_t* ctr_b, int offset,
                             const float* noise, float* y, size_t n, float div, float mul, ggan_stream_t stream) {
    GGAN_CHECK_ARG(ring && y && nslots > 0, "bad argument");
    if (n == 0) return 0;
    GGAN_LAUNCH("cast_scale_ring_i32", 0, 8.0 * n, cast_scale_ring_k, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, ring,
                nslots, ctr_a, ctr_b, offset, noise, y, n, div, mul);
    return 0;
}

int ggan_axpby(const float* x, const float* y, float* out, size_t n, float a, float b, float c, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && out, "null pointer");
    if (n == 0) return 0;
    GGAN_LAUNCH("axpby", 0, 12.0 * n, axpby_k, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, x, y, out, n, a, b, c);
    return 0;
}

int ggan_mix_mean(const float* k, const float* mu, const float* noise, float* out, int B, int K, int D, ggan_stream_t stream) {
    GGAN_CHECK_ARG(k && mu && noise && out && B > 0 && K > 0 && D > 0, "bad argument");
    GGAN_CHECK_ARG((D & 3) == 0 && !(((uintptr_t)mu | (uintptr_t)noise | (uintptr_t)out) & 15), "D a multiple of 4, 16-byte aligned buffers");
    const int n = B * (D / 4);
    GGAN_LAUNCH("mix_mean", 2.0 * B * K * D, 4.0 * (2.0 * B * D + (double)K * D), mix_mean_k, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, k,
                mu, noise, out, B, K, D / 4);
    return 0;
}

int ggan_row_lerp(const float* x, const float* y, const float* alpha, float* out, int rows, int cols, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && y && alpha && out && rows > 0 && cols > 0, "bad argument");
    size_t n = (size_t)rows * cols;
    if ((double)n * cols < 4.0e9) {
        GGAN_LAUNCH("row_lerp", 0, 12.0 * n, row_lerp32_k, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, x, y, alpha, out, (unsigned)n, cols,
                    make_fastdiv((uint32_t)cols));
        return 0;
    }
    GGAN_LAUNCH("row_lerp", 0, 12.0 * n, row_lerp_k, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, x, y, alpha, out, rows, cols);
    return 0;
}

int ggan_colsum(const float* x, float* out, int rows, int cols, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && out && rows > 0 && cols > 0, "bad argument");
    GGAN_LAUNCH("colsum", 0, 4.0 * rows * cols, colsum_k, dim3(cdiv(cols, 64)), dim3(64), 0, (hipStream_t)stream, x, out, rows, cols);
    return 0;
}

int ggan_colsum_tall(const float* x, float* out, int rows, int cols, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && out && rows > 0 && cols > 0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int tiles = cdiv(cols, 64);
    int P = cdiv(1024, tiles);                       // ~1024 workgroups, at least 64 rows per slab
    if (P > rows / 64) P = rows / 64;
    ws = ws_scratch(ws, ws_bytes);
    if (P < 2 || !ws || (size_t)cols * P * sizeof(float) > ws_bytes) return ggan_colsum(x, out, rows, cols, stream);
    const int rows_per_slab = cdiv(rows, P);
    float* part = (float*)ws;
    GGAN_LAUNCH("colsum_tall", 0, 4.0 * rows * cols, colsum_part_k, dim3(tiles, P), dim3(256), 0, s, x, part, rows, cols, rows_per_slab, P);
    GGAN_LAUNCH("chansum_final", 0, 4.0 * cols * P, chansum_final_k, dim3(cdiv(cols, 4)), dim3(256), 0, s, (const float*)part, out, cols, P);
    return 0;
}

int ggan_chansum(const float* x, float* out, int N, int C, int HW, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && out && N > 0 && C > 0 && HW > 0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    // stage 1 over ~1024 workgroups (C x P, each sums every P-th image plane of one channel), stage 2 adds the P
    // partials per channel in a fixed order: chip-wide bandwidth, deterministic, scratch supplied by the caller
    int P = 1024 / C;
    if (P > N) P = N;
    ws = ws_scratch(ws, ws_bytes);
    if (!ws || (size_t)C * P * sizeof(float) > ws_bytes) P = 1;
    if (P <= 1) {
        const int threads = (size_t)N * HW >= 16384 ? 1024 : 256;
        GGAN_LAUNCH("chansum", 0, 4.0 * N * C * HW, chansum_part_k, dim3(C, 1), dim3(threads), 0, s, x, out, N, C, HW, 1);
        return 0;
    }
    const int threads = HW >= 1024 ? 256 : (HW >= 256 ? 128 : 64);
    float* part = (float*)ws;
    GGAN_LAUNCH("chansum", 0, 4.0 * N * C * HW, chansum_part_k, dim3(C, P), dim3(threads), 0, s, x, part, N, C, HW, P);
    GGAN_LAUNCH("chansum_final", 0, 4.0 * C * P, chansum_final_k, dim3(cdiv(C, 4)), dim3(256), 0, s, (const float*)part, out, C, P);
    return 0;
}

int ggan_bce_logits_fwd(const float* x, float label, float weight, float* loss, int n, int accumulate, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && loss && n > 0, "bad argument");
    GGAN_LAUNCH("bce_logits_fwd", 0, 4.0 * n, bce_fwd_k, dim3(1), dim3(256), 0, (hipStream_t)stream, x, label, weight, loss, n, accumulate);
    return 0;
}

int ggan_bce_logits_bwd(const float* x, float label, float weight, const float* gloss, float* gx, int n, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && gloss && gx && n > 0, "bad argument");
    GGAN_LAUNCH("bce_logits_bwd", 0, 8.0 * n, bce_bwd_k, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, label, weight, gloss, gx, n);
    return 0;
}

static int bce_table(BceTable& t, const float* const* xs, float* const* gxs, const float* labels, const float* weights,
                     const int* ns, int count, int* max_n) {
    *max_n = 0;
    for (int i = 0; i < count; ++i) {
        if (!xs[i] || ns[i] <= 0 || (gxs && !gxs[i])) return -1;
        t.x[i] = xs[i]; t.gx[i] = gxs ? gxs[i] : nullptr; t.z[i] = labels[i]; t.w[i] = weights[i]; t.n[i] = ns[i];
        if (ns[i] > *max_n) *max_n = ns[i];
    }
    t.count = count;
    return 0;
}

int ggan_bce_logits_multi_fwd(const float* const* xs, const float* labels, const float* weights, const int* ns, int count,
                              float* loss, ggan_stream_t stream) {
    GGAN_CHECK_ARG(xs && labels && weights && ns && loss, "null pointer");
    GGAN_CHECK_ARG(count > 0 && count <= GGAN_BCE_MAX, "count out of range");
    BceTable t;
    int mx;
    GGAN_CHECK_ARG(bce_table(t, xs, nullptr, labels, weights, ns, count, &mx) == 0, "bad term");
    GGAN_LAUNCH("bce_logits_fwd", 0, 4.0 * mx * count, bce_multi_fwd_k, dim3(1), dim3(256), 0, (hipStream_t)stream, t, loss);
    return 0;
}

int ggan_bce_logits_multi_fwd_grad(const float* const* xs, const float* labels, const float* weights, const int* ns, int count,
                                   float* loss, float* const* gxs, ggan_stream_t stream) {
    GGAN_CHECK_ARG(xs && labels && weights && ns && loss && gxs, "null pointer");
    GGAN_CHECK_ARG(count > 0 && count <= GGAN_BCE_MAX, "count out of range");
    BceTable t;
    int mx;
    GGAN_CHECK_ARG(bce_table(t, xs, gxs, labels, weights, ns, count, &mx) == 0, "bad term");
    GGAN_LAUNCH("bce_logits_fwd_grad", 0, 8.0 * mx * count, bce_multi_fwd_k, dim3(1), dim3(256), 0, (hipStream_t)stream, t, loss);
    return 0;
}

int ggan_bce_heads_bwd(const float* const* xs, const float* labels, const float* weights, const int* ns, int count, float* loss,
                       float* const* gxs, int nheads, const int* head_terms, const int* Ms, const int* Hs, const float* const* hs,
                       const float* const* w_outs, const float* alphas, float* const* ghs, float* const* d_wouts,
                       float* const* d_bouts, ggan_stream_t stream) {
    GGAN_CHECK_ARG(xs && labels && weights && ns && loss && gxs && head_terms && Ms && Hs && hs && w_outs && alphas && ghs, "null pointer");
    GGAN_CHECK_ARG(count > 0 && count <= GGAN_BCE_MAX && nheads >= 1 && nheads <= GGAN_BCE_HEADS, "count out of range");
    BceTable t;
    int mx;
    GGAN_CHECK_ARG(bce_table(t, xs, gxs, labels, weights, ns, count, &mx) == 0, "bad term");
    BceHeads H;
    memset(&H, 0, sizeof(H));
    H.count = nheads;
    int k = 0, wg = 1;
    for (int a = 0; a < nheads; ++a) {
        BceHead& d = H.hd[a];
        GGAN_CHECK_ARG(hs[a] && w_outs[a] && ghs[a] && Ms[a] > 0 && Ms[a] <= GGAN_HEAD_BCE_MAX_ROWS && Hs[a] > 0 && head_terms[a] > 0 &&
                       k + head_terms[a] <= count, "bad head");
        d.h = hs[a]; d.w_out = w_outs[a]; d.gh = ghs[a]; d.d_wout = d_wouts ? d_wouts[a] : nullptr; d.d_bout = d_bouts ? d_bouts[a] : nullptr;
        d.alpha = alphas[a]; d.M = Ms[a]; d.H = Hs[a]; d.k0 = k; d.k1 = k + head_terms[a]; d.first = wg;
        int rows = 0;
        for (int i = d.k0; i < d.k1; ++i) {
            GGAN_CHECK_ARG(xs[i] == xs[d.k0] + rows, "a head's terms must be consecutive row ranges of one logits vector");
            rows += ns[i];
        }
        GGAN_CHECK_ARG(rows == d.M, "a head's terms must cover its rows");
        k = d.k1;
        wg += cdiv(d.H, 16);
    }
    GGAN_CHECK_ARG(k == count, "every term belongs to a head");
    GGAN_LAUNCH("bce_head_bwd", 0, 0, bce_head_bwd_k, dim3(wg), dim3(256), 0, (hipStream_t)stream, t, loss, H);
    return 0;
}

int ggan_bce_head_bwd(const float* const* xs, const float* labels, const float* weights, const int* ns, int count, float* loss,
                      float* const* gxs, int M, int H, const float* h, const float* w_out, float alpha, float* gh, float* d_wout,
                      float* d_bout, ggan_stream_t stream) {
    return ggan_bce_heads_bwd(xs, labels, weights, ns, count, loss, gxs, 1, &count, &M, &H, &h, &w_out, &alpha, &gh, &d_wout, &d_bout, stream);
}

int ggan_bce_logits_multi_bwd(const float* const* xs, const float* labels, const float* weights, const int* ns, int count,
                              const float* gloss, float* const* gxs, ggan_stream_t stream) {
    GGAN_CHECK_ARG(xs && labels && weights && ns && gloss && gxs, "null pointer");
    GGAN_CHECK_ARG(count > 0 && count <= GGAN_BCE_MAX, "count out of range");
    BceTable t;
    int mx;
    GGAN_CHECK_ARG(bce_table(t, xs, gxs, labels, weights, ns, count, &mx) == 0, "bad term");
    GGAN_LAUNCH("bce_logits_bwd", 0, 8.0 * mx * count, bce_multi_bwd_k, dim3(cdiv(mx, 256), count), dim3(256), 0,
                (hipStream_t)stream, t, gloss);
    return 0;
}

int ggan_dist_fwd(const float* x, const float* y, float* out, size_t n, int p, float weight, int accumulate, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && y && out && n > 0 && (p == 1 || p == 2), "bad argument");
    GGAN_LAUNCH("dist_fwd", 0, 8.0 * n, dist_fwd_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, y, out, n, p, weight, accumulate);
    return 0;
}

int ggan_dist_bwd(const float* x, const float* y, const float* gout, float* gx, float* gy, size_t n, int p, float weight,
                  ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && y && gout && (gx || gy) && n > 0 && (p == 1 || p == 2), "bad argument");
    GGAN_LAUNCH("dist_bwd", 0, 16.0 * n, dist_bwd_k, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, x, y, gout, gx, gy, n, p, weight);
    return 0;
}

int ggan_mean_fwd(const float* x, float weight, float* loss, int n, int accumulate, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && loss && n > 0, "bad argument");
    GGAN_LAUNCH("mean_fwd", 0, 4.0 * n, mean_fwd_k, dim3(1), dim3(256), 0, (hipStream_t)stream, x, weight, loss, n, accumulate);
    return 0;
}

int ggan_mean_multi_fwd_grad(const float* const* xs, const float* weights, const int* ns, int count, float* loss, float* const* gxs,
                             ggan_stream_t stream) {
    GGAN_CHECK_ARG(xs && weights && ns && loss, "null pointer");
    GGAN_CHECK_ARG(count > 0 && count <= GGAN_BCE_MAX, "count out of range");
    BceTable t;
    int mx;
    float zeros[GGAN_BCE_MAX] = {0.f};
    GGAN_CHECK_ARG(bce_table(t, xs, gxs, zeros, weights, ns, count, &mx) == 0, "bad term");
    GGAN_LAUNCH("mean_multi_fwd", 0, 8.0 * mx * count, mean_multi_fwd_k, dim3(1), dim3(256), 0, (hipStream_t)stream, t, loss);
    return 0;
}

int ggan_mean_bwd(const float* gloss, float weight, float* gx, int n, ggan_stream_t stream) {
    GGAN_CHECK_ARG(gloss && gx && n > 0, "bad argument");
    GGAN_LAUNCH("mean_bwd", 0, 4.0 * n, mean_bwd_k, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, gloss, weight, gx, n);
    return 0;
}

int ggan_gp_penalty_fwd(const float* g, float* slopes, float* pen, int B, int D, float lam, ggan_stream_t stream) {
    GGAN_CHECK_ARG(g && slopes && pen && B > 0 && D > 0, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    GGAN_LAUNCH("gp_slopes", 0, 4.0 * B * D, gp_slopes_k, dim3(B), dim3(256), 0, s, g, slopes, D);
    GGAN_LAUNCH("gp_penalty", 0, 4.0 * B, gp_pen_k, dim3(1), dim3(256), 0, s, slopes, pen, B, lam);
    return 0;
}

int ggan_gp_penalty_fwd_grad(const float* g, float* slopes, float* pen, float* gg_unit, int32_t* arrive, int B, int D, float lam,
                             ggan_stream_t stream) {
    GGAN_CHECK_ARG(g && slopes && pen && gg_unit && arrive && B > 0 && D > 0, "bad argument");
    GGAN_LAUNCH("gp_fwd_grad", 0, 8.0 * B * D, gp_fwd_grad_k, dim3(B), dim3(256), 0, (hipStream_t)stream, g, slopes, pen, gg_unit, arrive, B, D,
                lam);
    return 0;
}

int ggan_gp_penalty_bwd(const float* g, const float* slopes, const float* gpen, float* gg, int B, int D, float lam, ggan_stream_t stream) {
    GGAN_CHECK_ARG(g && slopes && gpen && gg && B > 0 && D > 0, "bad argument");
    GGAN_LAUNCH("gp_penalty_bwd", 0, 8.0 * B * D, gp_bwd_k, dim3(B), dim3(256), 0, (hipStream_t)stream, g, slopes, gpen, gg, B, D, lam);
    return 0;
}

int ggan_adam_step(float* theta, const float* g, float* m, float* v, size_t n, const int32_t* step, float lr,
                   float beta1, float beta2, float eps, float grad_scale, ggan_stream_t stream) {
    GGAN_CHECK_ARG(theta && g && m && v && step, "null pointer");
    if (n == 0) return 0;
    GGAN_CHECK_ARG(aligned16(theta) && aligned16(g) && aligned16(m) && aligned16(v), "buffers must be 16-byte aligned");
    GGAN_LAUNCH("adam_step", 0, 28.0 * n, adam_k<false>, dim3(grid_for(n, 8)), dim3(kBlock), 0, (hipStream_t)stream, theta, g, m, v, n, step, lr, beta1, beta2, eps, grad_scale);
    return 0;
}

int ggan_adam_step_counted(float* theta, const float* g, float* m, float* v, size_t n, const int32_t* step, float lr,
                           float beta1, float beta2, float eps, float grad_scale, ggan_stream_t stream) {
    GGAN_CHECK_ARG(theta && g && m && v && step, "null pointer");
    if (n == 0) return 0;
    GGAN_CHECK_ARG(aligned16(theta) && aligned16(g) && aligned16(m) && aligned16(v), "buffers must be 16-byte aligned");
    GGAN_LAUNCH("adam_step", 0, 28.0 * n, adam_k<true>, dim3(grid_for(n, 8)), dim3(kBlock), 0, (hipStream_t)stream, theta, g, m, v, n, step, lr, beta1, beta2, eps, grad_scale);
    return 0;
}

int ggan_rmsprop_step(float* theta, const float* g, float* ms, size_t n, float lr, float decay, float eps, float grad_scale,
                      float clip_lo, float clip_hi, ggan_stream_t stream) {
    GGAN_CHECK_ARG(theta && g && ms, "null pointer");
    GGAN_CHECK_ARG(clip_lo <= clip_hi, "empty clip interval");
    if (n == 0) return 0;
    GGAN_LAUNCH("rmsprop_step", 0, 20.0 * n, rmsprop_k, dim3(grid_for(n, 8)), dim3(kBlock), 0, (hipStream_t)stream, theta, g, ms, n, lr,
                decay, eps, grad_scale, clip_lo, clip_hi);
    return 0;
}

int ggan_adam_advance(int32_t* step, ggan_stream_t stream) {
    GGAN_CHECK_ARG(step, "null pointer");
    GGAN_LAUNCH("adam_advance", 0, 8, adam_advance_k, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
    return 0;
}

int ggan_pack_parts2(const float* const* srcs, const size_t* sizes, const size_t* offsets, const int* parts, const size_t* strides,
                     const float* const* srcs2, const int* parts2, const size_t* strides2, int count, float* flat, int32_t* bump,
                     ggan_stream_t stream) {
    GGAN_CHECK_ARG(srcs && sizes && offsets && flat, "null pointer");
    GGAN_CHECK_ARG(count > 0 && count <= GGAN_PACK_MAX, "count out of range");
    PackTable t;
    size_t mx = 0, tot = 0;
    for (int i = 0; i < count; ++i) {
        t.src[i] = srcs[i]; t.size[i] = sizes[i]; t.off[i] = offsets[i];
        t.parts[i] = parts ? parts[i] : 1;
        t.pstride[i] = strides ? strides[i] : 0;
        t.src2[i] = srcs2 ? srcs2[i] : nullptr;
        t.parts2[i] = (srcs2 && parts2) ? parts2[i] : 1;
        t.pstride2[i] = (srcs2 && strides2) ? strides2[i] : 0;
        GGAN_CHECK_ARG(t.parts[i] >= 1 && t.parts2[i] >= 1, "parts must be >= 1");
        if (sizes[i] > mx) mx = sizes[i];
        tot += sizes[i] * (size_t)(t.parts[i] + (t.src2[i] ? t.parts2[i] : 0));
    }
    t.count = count;
    t.bump = bump;
    int gx = (int)cdivz(mx, (size_t)kBlock * 16);
    if (gx < 1) gx = 1;
    if (gx > 512) gx = 512;
    GGAN_LAUNCH("pack", 0, 4.0 * tot + 4.0 * mx, pack_k, dim3(gx, count), dim3(kBlock), 0, (hipStream_t)stream, t, flat);
    return 0;
}

int ggan_pack_adam(const float* const* srcs, const size_t* sizes, const size_t* offsets, const int* parts, const size_t* strides,
                   const float* const* srcs2, const int* parts2, const size_t* strides2, int count, float* flat, float* theta,
                   float* m, float* v, int32_t* step, int32_t* arrive, float lr, float beta1, float beta2, float eps,
                   float grad_scale, ggan_stream_t stream) {
    GGAN_CHECK_ARG(srcs && sizes && offsets && flat && theta && m && v && step, "null pointer");
    GGAN_CHECK_ARG(count > 0 && count <= GGAN_PACK_MAX, "count out of range");
    PackTable t;
    size_t mx = 0, tot = 0, all = 0;
    for (int i = 0; i < count; ++i) {
        t.src[i] = srcs[i]; t.size[i] = sizes[i]; t.off[i] = offsets[i];
        t.parts[i] = parts ? parts[i] : 1;
        t.pstride[i] = strides ? strides[i] : 0;
        t.src2[i] = srcs2 ? srcs2[i] : nullptr;
        t.parts2[i] = (srcs2 && parts2) ? parts2[i] : 1;
        t.pstride2[i] = (srcs2 && strides2) ? strides2[i] : 0;
        GGAN_CHECK_ARG(t.parts[i] >= 1 && t.parts2[i] >= 1, "parts must be >= 1");
        if (sizes[i] > mx) mx = sizes[i];
        tot += sizes[i] * (size_t)(t.parts[i] + (t.src2[i] ? t.parts2[i] : 0));
        all += sizes[i];
    }
    t.count = count;
    t.bump = step;
    int nb = 0;
    for (int i = 0; i < count; ++i) {
        t.first[i] = nb;
        nb += (int)cdivz(sizes[i] > 0 ? sizes[i] : 1, (size_t)pack_chunk(t.parts[i] + (t.src2[i] ? t.parts2[i] : 0)));
    }
    t.first[count] = nb;
    (void)mx;
    GGAN_LAUNCH("pack_adam", 0, 4.0 * tot + 28.0 * all, pack_adam_k, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream, t, flat, theta,
                m, v, arrive, lr, beta1, beta2, eps, grad_scale);
    return 0;
}

int ggan_pack_parts(const float* const* srcs, const size_t* sizes, const size_t* offsets, const int* parts,
                    const size_t* strides, int count, float* flat, int32_t* bump, ggan_stream_t stream) {
    return ggan_pack_parts2(srcs, sizes, offsets, parts, strides, nullptr, nullptr, nullptr, count, flat, bump, stream);
}

int ggan_pack(const float* const* srcs, const size_t* sizes, const size_t* offsets, int count, float* flat, ggan_stream_t stream) {
    return ggan_pack_parts(srcs, sizes, offsets, nullptr, nullptr, count, flat, nullptr, stream);
}

int ggan_gmm_latent_fwd(const float* z, const float* mu, const float* gumbel_u, float* logits, float* k, int B, int K, int D,
                        float log_pi, float temp, ggan_stream_t stream) {
    GGAN_CHECK_ARG(z && mu && gumbel_u && k, "null pointer");
    GGAN_CHECK_ARG(B > 0 && K > 0 && K <= kGmmMaxK && D > 0 && temp > 0.f, "bad shape");
    GGAN_LAUNCH("gmm_latent_fwd", 3.0 * B * K * D, 0, gmm_latent_fwd_k, dim3(B), dim3(256), 0, (hipStream_t)stream, z, mu, gumbel_u, logits, k,
                K, D, log_pi, 1.f / temp);
    return 0;
}

int ggan_gmm_latent_bwd(const float* z, const float* mu, const float* k, const float* g_logits, const float* g_k, float* dz,
                        float* dmu, int B, int K, int D, float temp, ggan_stream_t stream) {
    GGAN_CHECK_ARG(z && mu && k && (g_logits || g_k) && (dz || dmu), "null pointer");
    GGAN_CHECK_ARG(B > 0 && B <= kGmmMaxK && K > 0 && K <= kGmmMaxK && D > 0 && temp > 0.f, "bad shape");
    GGAN_LAUNCH("gmm_latent_bwd", 4.0 * B * K * D, 0, gmm_latent_bwd_k, dim3(B + (dmu ? K : 0)), dim3(256), 0, (hipStream_t)stream, z, mu, k,
                g_logits, g_k, dz, dmu, B, K, D, 1.f / temp);
    return 0;
}

int ggan_noise_fill(float* const* dsts, const size_t* sizes, const int* kinds, const float* a, const float* b, const int* widths,
                    int count, uint64_t* state, ggan_stream_t stream) {
    GGAN_CHECK_ARG(dsts && sizes && kinds && a && b && widths && state, "null pointer");
    GGAN_CHECK_ARG(count > 0 && count <= GGAN_NOISE_MAX, "count out of range");
    NoiseTable t;
    size_t mx = 0, tot = 0;
    for (int i = 0; i < count; ++i) {
        GGAN_CHECK_ARG(dsts[i] && sizes[i] > 0 && sizes[i] < 0x7FFFFFFFull && kinds[i] >= 0 && kinds[i] <= 2, "bad noise spec");
        GGAN_CHECK_ARG(kinds[i] != 2 || (widths[i] > 0 && sizes[i] % (size_t)widths[i] == 0), "one-hot rows need a width dividing the size");
        t.dst[i] = dsts[i]; t.n[i] = (unsigned)sizes[i]; t.kind[i] = kinds[i]; t.a[i] = a[i]; t.b[i] = b[i]; t.K[i] = widths[i];
        t.slot[i] = i;
        t.step[i] = 0;
        if (sizes[i] > mx) mx = sizes[i];
        tot += sizes[i];
    }
    t.count = count;
    t.advance = 1;
    int gx = (int)cdivz(mx, (size_t)1024);
    if (gx < 1) gx = 1;
    if (gx > 256) gx = 256;
    GGAN_LAUNCH("noise_fill", 0, 4.0 * tot, noise_fill_k, dim3(gx, count), dim3(256), 0, (hipStream_t)stream, t, (unsigned long long*)state);
    return 0;
}

static int mmd_params(MmdParams& P, const float* X, const float* Y, int m, int n, int d, const float* sigmas, const float* wts, int ns) {
    if (!(X && Y && sigmas) || m <= 0 || n <= 0 || d <= 0 || ns <= 0 || ns > kMmdMaxSigmas || m + n > 512) return -1;
    P.X = X; P.Y = Y; P.m = m; P.n = n; P.d = d; P.ns = ns;
    for (int i = 0; i < ns; ++i) {
        P.gamma[i] = 1.f / (2.f * sigmas[i] * sigmas[i]);
        P.wt[i] = wts ? wts[i] : 1.f;
    }
    return 0;
}

int ggan_mix_rbf_mmd2_fwd(const float* X, const float* Y, int m, int n, int d, const float* sigmas, const float* wts, int ns,
                          float* out, float* row_scratch, ggan_stream_t stream) {
    MmdParams P;
    GGAN_CHECK_ARG(mmd_params(P, X, Y, m, n, d, sigmas, wts, ns) == 0 && out && row_scratch, "bad argument");
    GGAN_LAUNCH("mmd2_rows", 3.0 * (m + n) * (m + n) * d, 0, mmd2_rows_k, dim3(m + n), dim3(256), 0, (hipStream_t)stream, P, row_scratch);
    GGAN_LAUNCH("mmd2_final", 0, 0, mmd2_final_k, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)row_scratch, m + n, out);
    return 0;
}

int ggan_mix_rbf_mmd2_bwd(const float* X, const float* Y, int m, int n, int d, const float* sigmas, const float* wts, int ns,
                          const float* gout, float* dX, float* dY, ggan_stream_t stream) {
    MmdParams P;
    GGAN_CHECK_ARG(mmd_params(P, X, Y, m, n, d, sigmas, wts, ns) == 0 && gout && (dX || dY), "bad argument");
    GGAN_LAUNCH("mmd2_bwd", 5.0 * (m + n) * (m + n) * d, 0, mmd2_bwd_k, dim3(m + n), dim3(256), 0, (hipStream_t)stream, P, gout, dX, dY);
    return 0;
}

int ggan_reparam_fwd(const float* mean, const float* log_std, const float* eps, float* z, float* std_out, size_t n, ggan_stream_t stream) {
    GGAN_CHECK_ARG(mean && log_std && eps && z && std_out, "null pointer");
    if (n == 0) return 0;
    GGAN_LAUNCH("reparam_fwd", 0, 20.0 * n, reparam_fwd_k, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, mean, log_std, eps, z, std_out, n);
    return 0;
}

int ggan_reparam_bwd(const float* gz, const float* gstd, const float* eps, const float* std_in, float* gmean, float* glog_std, size_t n,
                     ggan_stream_t stream) {
    GGAN_CHECK_ARG((gz || gstd) && eps && std_in && gmean && glog_std, "null pointer");
    if (n == 0) return 0;
    GGAN_LAUNCH("reparam_bwd", 0, 24.0 * n, reparam_bwd_k, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, gz, gstd, eps, std_in, gmean,
                glog_std, n);
    return 0;
}

static int agg_params(AggP& P, int kind, const float* mu, const float* sd, const float* k_onehot, const float* eps_q, const float* z_p, int nx,
                      int nz, int d, int n_coms) {
    if (kind < 0 || kind > 2 || !mu || !sd || nx <= 0 || nz <= 0 || d <= 0 || n_coms <= 0 || d > 8192 || nx > 8192) return -1;
    if (kind != 1 && (!k_onehot || !eps_q)) return -1;
    if (kind != 0 && !z_p) return -1;
    P.mu = mu; P.sd = sd; P.k = k_onehot; P.eps_q = eps_q; P.z_p = z_p;
    P.kind = kind; P.nx = nx; P.nz = nz; P.d = d; P.n_coms = n_coms;
    return 0;
}

int ggan_agg_div_fwd(int kind, const float* mu, const float* sd, const float* k_onehot, const float* eps_q, const float* z_p, int nx, int nz,
                     int d, int n_coms, float* out, float* Z, float* A, float* Bv, float* T, ggan_stream_t stream) {
    AggP P;
    GGAN_CHECK_ARG(agg_params(P, kind, mu, sd, k_onehot, eps_q, z_p, nx, nz, d, n_coms) == 0 && out && Z && A && Bv && T, "bad argument");
    const int ns = kind == 2 ? 2 * nz : nz;
    GGAN_LAUNCH("agg_div_fwd", 8.0 * ns * nx * d, 0, agg_div_fwd_k, dim3(ns), dim3(128), d * sizeof(float), (hipStream_t)stream, P, Z, A, Bv, T);
    GGAN_LAUNCH("agg_div_final", 0, 0, agg_div_final_k, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)T, ns, nz, out);
    return 0;
}

int ggan_agg_div_bwd(int kind, const float* mu, const float* sd, const float* k_onehot, const float* eps_q, int nx, int nz, int d, int n_coms,
                     const float* Z, const float* A, const float* Bv, const float* gout, float* W, float* GZ, float* gmu, float* gsd,
                     ggan_stream_t stream) {
    AggP P;
    GGAN_CHECK_ARG(agg_params(P, kind, mu, sd, k_onehot, eps_q, Z /* unused */, nx, nz, d, n_coms) == 0 && Z && A && Bv && gout && W && GZ && gmu &&
                       gsd, "bad argument");
    const int ns = kind == 2 ? 2 * nz : nz;
    GGAN_LAUNCH("agg_div_bwd_samples", 6.0 * ns * nx * d, 0, agg_div_bwd_samples_k, dim3(ns), dim3(128), nx * sizeof(float), (hipStream_t)stream, P,
                Z, A, Bv, gout, W, GZ);
    GGAN_LAUNCH("agg_div_bwd_comps", 8.0 * ns * nx * d, 0, agg_div_bwd_comps_k, dim3(nx), dim3(128), 0, (hipStream_t)stream, P, ns, Z,
                (const float*)W, (const float*)GZ, gmu, gsd);
    return 0;
}

}  // extern "C"
