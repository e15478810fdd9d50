// Conv3D of the state-space scripts' '3dcnn' sequence critic (tflib/ops/conv3d.py:6-51: tf.nn.conv3d, NDHWC, filter
// [fl, fs, fs, in, out], strides (stride_len, stride, stride), SAME padding).  The filter is stored exactly as the [K, Co] operand of a
// GEMM with K = (dl, dh, dw, ci), so the layer is  y = im2col(x) @ W + b  on the MFMA GEMM of gemm.hip (bias / activation in its
// epilogue), the filter gradient is im2col(x)^T @ gy and the data gradient col2im(gy @ W^T).  This file holds the two layout kernels;
// both are pure HBM streaming (the patch matrix of the largest layer is ~0.5-0.8 GB -- 288 GB of HBM make materialising it the cheap
// option) and each is the other's adjoint, so the pair is closed under differentiation.
#include "common.h"
#include "conv.h"
#include <string.h>
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
        // only the taps of the voxel's residue class reach it (d = (coordinate + pad) mod stride, then every stride-th), in ascending order
        // as before; walking all kl * k * k taps and skipping left 7 of 8 iterations of a stride-2 layer on `continue` (round 6)
        for (int dl = (l + g.pl) % g.sl; dl < g.kl; dl += g.sl) {
            const int a = l + g.pl - dl;
            if (a < 0) break;
            const int ol = a / g.sl;
            if (ol >= g.Lo) continue;
            for (int dh = (h + g.ph) % g.s; dh < g.k; dh += g.s) {
                const int b = h + g.ph - dh;
                if (b < 0) break;
                const int oh = b / g.s;
                if (oh >= g.Ho) continue;
                for (int dw = (w + g.pw) % g.s; dw < g.k; dw += g.s) {
                    const int c = w + g.pw - dw;
                    if (c < 0) break;
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


// ------------------------------------------------------------------------------------------------------------------------------
// Implicit GEMM: the same three products without the patch matrix.  The gathered operand is addressed in place -- patch-matrix
// element (row, column) = (voxel (n, r_l, r_h, r_w), (t_l, t_h, t_w, c)) is  src[n, r_l*rs_l + o_l + ts*t_l, .., c]  or 0 outside the
// volume -- with raw buffer loads whose out-of-range lanes carry an out-of-bounds offset (no branches around the loads):
//   forward        y[M, Co]   = gather(x) @ W            rows = output voxels, taps ascending (rs = stride, o = -pad, ts = +1)
//   filter grad    gw[K, Co]  = gather(x)^T @ gy         the same elements, read as the transposed operand, split over the rows
//   data grad      gx[class]  = gather(gy) @ W_class^T   per residue class (l % sl, h % s, w % s) of the input voxels: only the
//                  taps d = d0 + stride*t reach a class, and the output voxel is o = r' + q - t (rs = 1, ts = -1): one
//                  product per class with K = (taps of the class) * Co, all classes in one launch (blockIdx.z)
// Tile 32*WM x 32*WN, four waves of 32x32x2 fp32 MFMAs, 32 k per step, LDS double-buffered with a register ring two steps deep
// (the scheme of gemm.hip).  A 16-byte unit is 4 consecutive channels of one tap (C % 4 == 0) or four dword gathers (the
// one-channel first layer: 4 consecutive taps along w).
constexpr unsigned IOOB = 0x7FFFFFF0u;
constexpr int IKS = 32;

struct IgClass {
    int M, K;               // rows and reduction length (columns of the gathered operand) of this class's product
    int RL, RH, RW;         // row = ((n*RL + rl)*RH + rh)*RW + rw
    int TL, TH, TW;         // column = ((tl*TH + th)*TW + tw)*SC + c
    int ol, oh, ow;         // source coordinate = r*rs + o + ts*t
    int cl, ch, cw;         // data grad: the class residues (result voxel = r*stride + c) ...
    int dl0, dh0, dw0;      // ... and the first filter tap of the class (filter tap = d0 + stride*t)
    FastDiv d_RW, d_RH, d_RL, d_TW, d_TH;
};

struct IgParams {
    const float* A;         // gathered tensor [n][SL][SH][SW][SC]
    const float* B;
    const float* bias;
    float* C;
    unsigned a_bytes, b_bytes;
    int N;                  // result columns
    int SL, SH, SW, SC;
    int rsl, rs, ts;        // row strides (length, spatial), tap direction
    int sl, s, fk;          // data grad: strides and spatial filter size (filter tap index ((dl*fk + dh)*fk + dw))
    int OL, OH, OW;         // data grad: the result volume [n][OL][OH][OW][N]
    int SK, kps;            // split of the reduction (forward / filter grad: slabs of `slab` floats, summed by splitk_reduce)
    size_t slab;
    int act;
    float alpha;
    FastDiv d_SC;
    IgClass cls[8];
};

struct IgCol { int tl, th, tw, c; };

__device__ __forceinline__ IgCol ig_col(int k, const IgParams& P, const IgClass& Q) {
    IgCol r;
    const int t = (int)fdiv((uint32_t)k, P.d_SC);
    r.c = k - t * P.SC;
    const int t2 = (int)fdiv((uint32_t)t, Q.d_TW);
    r.tw = t - t2 * Q.TW;
    r.tl = (int)fdiv((uint32_t)t2, Q.d_TH);
    r.th = t2 - r.tl * Q.TH;
    return r;
}

struct IgRow { int n, l0, h0, w0; };      // n: image index (or -1: row beyond M), l0/h0/w0 = r*rs + o

__device__ __forceinline__ IgRow ig_row(int row, const IgParams& P, const IgClass& Q) {
    IgRow r;
    const int t = (int)fdiv((uint32_t)row, Q.d_RW);
    const int rw = row - t * Q.RW;
    const int t2 = (int)fdiv((uint32_t)t, Q.d_RH);
    const int rh = t - t2 * Q.RH;
    const int n = (int)fdiv((uint32_t)t2, Q.d_RL);
    const int rl = t2 - n * Q.RL;
    r.n = row < Q.M ? n : -1;
    r.l0 = rl * P.rsl + Q.ol; r.h0 = rh * P.rs + Q.oh; r.w0 = rw * P.rs + Q.ow;
    return r;
}

__device__ __forceinline__ unsigned ig_off(const IgRow& r, const IgCol& c, bool kok, const IgParams& P) {
    const int l = r.l0 + P.ts * c.tl, h = r.h0 + P.ts * c.th, w = r.w0 + P.ts * c.tw;
    const bool ok = kok && r.n >= 0 && (unsigned)l < (unsigned)P.SL && (unsigned)h < (unsigned)P.SH && (unsigned)w < (unsigned)P.SW;
    return ok ? (unsigned)((((r.n * P.SL + l) * P.SH + h) * P.SW + w) * P.SC + c.c) * 4u : IOOB;
}

typedef unsigned int u32x4c __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ig_ld4(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    const u32x4c t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    return make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
}
__device__ __forceinline__ float ig_ld1(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}

// KIND 0 forward, 1 filter gradient, 2 data gradient.  AV4: the gathered tensor has C % 4 == 0 (16-byte units).
template <int KIND, int WM, int WN, bool AV4>
__global__ __launch_bounds__(256) void conv3d_igemm_k(const IgParams P) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr bool AKC = KIND != 1;                       // gathered operand read k-contiguous (rows fixed per thread)
    constexpr bool BKC = KIND == 2;                       // filter read k-contiguous (data grad: W[tap][ci][co], k = co)
    constexpr int LA = AKC ? BM + 2 : BM + 4, LB = BKC ? BN + 2 : BN + 4;
    constexpr int UA = BM / 32, UB = BN / 32;             // 16-byte units per thread and step
    constexpr int ASZ = IKS * LA, BSZ = IKS * LB;
    static_assert(WM * WN == 4, "four waves");
    static_assert(KIND != 1 || (WM == 2 && WN == 2), "filter gradient: 64x64 tiles");
    constexpr int ABUF = 2 * ASZ > BM * (BN + 4) ? 2 * ASZ : BM * (BN + 4);      // (the result tile is staged in the A buffers)
    __shared__ __attribute__((aligned(16))) float As[ABUF];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BSZ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM, half = lane >> 5, l31 = lane & 31;
    const int ci = blockIdx.z / P.SK, split = blockIdx.z - ci * P.SK;
    const IgClass& Q = P.cls[ci];
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;       // (row tiles on x: up to 2^31 of them)
    const int Mrows = KIND == 1 ? Q.K : Q.M;              // rows of the result tile grid
    const int Kred = KIND == 1 ? Q.M : Q.K;               // reduction length
    if (m0 >= Mrows) return;
    const int kb = split * P.kps, ke = min(kb + P.kps, Kred);
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, (short)0, (int)P.a_bytes, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, (short)0, (int)P.b_bytes, 0x00020000);

    // ---- per-thread descriptors of the part that does not change over the reduction -------------------------------------------
    IgRow arow[AKC ? UA : 1];           // forward / data grad: the UA rows this thread stages
    IgCol acol[AKC ? 1 : 4];            // filter grad: the 4 patch columns (one unit) this thread stages
    bool acol_ok[AKC ? 1 : 4];
    if constexpr (AKC) {
#pragma unroll
        for (int u = 0; u < UA; ++u) arow[u] = ig_row(m0 + ((tid + 256 * u) >> 3), P, Q);
    } else {
        const int mc = m0 + (tid & 15) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acol[j] = ig_col(mc + (AV4 ? 0 : j), P, Q);
            if (AV4) acol[j].c += j;
            acol_ok[j] = mc + j < Q.K;
        }
    }

    float4 ra[2][UA], rb[2][UB];
    auto load_step = [&](int k0, float4 (&xa)[UA], float4 (&xb)[UB]) {
        if constexpr (AKC) {
            const int k = k0 + (tid & 7) * 4;
            IgCol c[AV4 ? 1 : 4];
#pragma unroll
            for (int j = 0; j < (AV4 ? 1 : 4); ++j) c[j] = ig_col(k + j, P, Q);
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                if constexpr (AV4) {
                    xa[u] = ig_ld4(rsA, ig_off(arow[u], c[0], k < ke, P));
                } else {
                    xa[u].x = ig_ld1(rsA, ig_off(arow[u], c[0], k < ke, P));
                    xa[u].y = ig_ld1(rsA, ig_off(arow[u], c[1], k + 1 < ke, P));
                    xa[u].z = ig_ld1(rsA, ig_off(arow[u], c[2], k + 2 < ke, P));
                    xa[u].w = ig_ld1(rsA, ig_off(arow[u], c[3], k + 3 < ke, P));
                }
            }
            if constexpr (BKC) {            // W[(filter tap)][n = ci][k = co]: the unit is 4 consecutive co of one tap and input channel
                const int ft = ((Q.dl0 + P.sl * c[0].tl) * P.fk + Q.dh0 + P.s * c[0].th) * P.fk + Q.dw0 + P.s * c[0].tw;
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int n = n0 + ((tid + 256 * u) >> 3);
                    xb[u] = ig_ld4(rsB, (n < P.N && k < ke) ? (unsigned)((ft * P.N + n) * P.SC + c[0].c) * 4u : IOOB);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int row = k0 + u * 16 + (tid >> 4);
                const IgRow r = ig_row(row, P, Q);
                const bool rok = row < ke;
                if constexpr (AV4) {
                    xa[u] = ig_ld4(rsA, ig_off(r, acol[0], rok && acol_ok[0], P));
                } else {
                    xa[u].x = ig_ld1(rsA, ig_off(r, acol[0], rok && acol_ok[0], P));
                    xa[u].y = ig_ld1(rsA, ig_off(r, acol[1], rok && acol_ok[1], P));
                    xa[u].z = ig_ld1(rsA, ig_off(r, acol[2], rok && acol_ok[2], P));
                    xa[u].w = ig_ld1(rsA, ig_off(r, acol[3], rok && acol_ok[3], P));
                }
            }
        }
        if constexpr (!BKC) {               // B[k][n] row-major with leading dimension N (the filter, or gy): 4 consecutive n
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int unit = tid + 256 * u, kk = unit / (BN / 4), n = n0 + (unit % (BN / 4)) * 4;
                const int k = k0 + kk;
                xb[u] = ig_ld4(rsB, (k < ke && n < P.N) ? (unsigned)(k * P.N + n) * 4u : IOOB);
            }
        }
    };
    auto store_step = [&](int buf, const float4 (&xa)[UA], const float4 (&xb)[UB]) {
        float* Ab = As + buf * ASZ;
        float* Bb = Bs + buf * BSZ;
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            if constexpr (AKC) {
                const int unit = tid + 256 * u;
                float* d = Ab + ((unit & 7) * 4) * LA + (unit >> 3);
                d[0] = xa[u].x; d[LA] = xa[u].y; d[2 * LA] = xa[u].z; d[3 * LA] = xa[u].w;
            } else {
                *reinterpret_cast<float4*>(Ab + (u * 16 + (tid >> 4)) * LA + (tid & 15) * 4) = xa[u];
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int unit = tid + 256 * u;
            if constexpr (BKC) {
                float* d = Bb + ((unit & 7) * 4) * LB + (unit >> 3);
                d[0] = xb[u].x; d[LB] = xb[u].y; d[2 * LB] = xb[u].z; d[3 * LB] = xb[u].w;
            } else {
                *reinterpret_cast<float4*>(Bb + (unit / (BN / 4)) * LB + (unit % (BN / 4)) * 4) = xb[u];
            }
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto mma_step = [&](int buf) {
        const float* Ab = As + buf * ASZ;
        const float* Bb = Bs + buf * BSZ;
        float fa[IKS / 2], fb[IKS / 2];
#pragma unroll
        for (int i = 0; i < IKS / 2; ++i) {
            fa[i] = Ab[(2 * i + half) * LA + wm * 32 + l31];
            fb[i] = Bb[(2 * i + half) * LB + wn * 32 + l31];
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
        for (int i = 0; i < IKS / 2; ++i) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[i], acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
    };
    load_step(kb, ra[0], rb[0]);
    store_step(0, ra[0], rb[0]);
    if (kb + IKS < ke) load_step(kb + IKS, ra[1], rb[1]);
    __syncthreads();
    for (int k0 = kb; k0 < ke; k0 += 2 * IKS) {
        if (k0 + 2 * IKS < ke) load_step(k0 + 2 * IKS, ra[0], rb[0]);
        mma_step(0);
        if (k0 + IKS < ke) store_step(1, ra[1], rb[1]);
        __syncthreads();
        if (k0 + IKS >= ke) break;
        if (k0 + 3 * IKS < ke) load_step(k0 + 3 * IKS, ra[1], rb[1]);
        mma_step(1);
        if (k0 + 2 * IKS < ke) store_step(0, ra[0], rb[0]);
        __syncthreads();
    }
    // ---- result tile through LDS, one float4 row segment per thread and pass ----------------------------------------------------
    constexpr int LC = BN + 4;
    float* Cs = As;
#pragma unroll
    for (int r = 0; r < 16; ++r) Cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LC + wn * 32 + l31] = acc[r];
    __syncthreads();
    const bool direct = P.SK == 1;
#pragma unroll
    for (int pass = 0; pass < BM * BN / 1024; ++pass) {
        const int idx = tid + pass * 256, ml = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
        const int m = m0 + ml, n = n0 + c4;
        if (m >= Mrows || n >= P.N) continue;
        float4 v = *reinterpret_cast<const float4*>(Cs + ml * LC + c4);
        size_t o;
        if constexpr (KIND == 2) {          // row of the class -> voxel of the result volume
            const IgRow r = ig_row(m, P, Q);      // (l0 = rl + ql: undo the offset)
            const int l = (r.l0 - Q.ol) * P.sl + Q.cl, h = (r.h0 - Q.oh) * P.s + Q.ch, w = (r.w0 - Q.ow) * P.s + Q.cw;
            o = ((((size_t)r.n * P.OL + l) * P.OH + h) * P.OW + w) * P.N + n;
        } else {
            o = (direct ? 0 : (size_t)split * P.slab) + (size_t)m * P.N + n;
        }
        if (KIND == 0 && direct) {
            float* vv = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int q = 0; q < 4; ++q) vv[q] = act_apply(vv[q] + (P.bias ? P.bias[n + q] : 0.f), P.act, P.alpha);
        }
        *reinterpret_cast<float4*>(P.C + o) = v;
    }
}

// which products the implicit kernels cover (the rest stays on im2col / col2im + GEMM)
bool ig_ok(const C3& g, int kind) {
    const size_t xb = (size_t)g.N * g.L * g.H * g.W * g.Ci * 4, yb = (size_t)g.N * g.Lo * g.Ho * g.Wo * g.Co * 4;
    const size_t rows = (size_t)g.N * g.Lo * g.Ho * g.Wo, vox = (size_t)g.N * g.L * g.H * g.W;
    const size_t K = (size_t)g.kl * g.k * g.k * g.Ci;
    if (xb >= 0x7FFFFFF0ull || yb >= 0x7FFFFFF0ull || K * g.Co * 4 >= 0x7FFFFFF0ull) return false;
    if (rows >= (1u << 24) || vox >= (1u << 24) || K >= (1u << 16) || (g.Co & 3)) return false;
    if (kind == 2) return (g.Ci & 3) == 0 && g.sl <= 2 && g.s <= 2 && g.Ci >= 16;   // (thin inputs: a 32-wide tile would idle)
    return true;
}

void ig_common(IgParams& P, const C3& g) {
    memset(&P, 0, sizeof(P));
    P.SK = 1;
}

void ig_fwd_class(IgParams& P, const C3& g) {      // rows = output voxels, ascending taps over x
    P.SL = g.L; P.SH = g.H; P.SW = g.W; P.SC = g.Ci;
    P.rsl = g.sl; P.rs = g.s; P.ts = 1;
    P.d_SC = make_fastdiv(g.Ci);
    IgClass& Q = P.cls[0];
    Q.M = g.N * g.Lo * g.Ho * g.Wo; Q.K = g.kl * g.k * g.k * g.Ci;
    Q.RL = g.Lo; Q.RH = g.Ho; Q.RW = g.Wo; Q.TL = g.kl; Q.TH = g.k; Q.TW = g.k;
    Q.ol = -g.pl; Q.oh = -g.ph; Q.ow = -g.pw;
    Q.d_RW = make_fastdiv(Q.RW); Q.d_RH = make_fastdiv(Q.RH); Q.d_RL = make_fastdiv(Q.RL);
    Q.d_TW = make_fastdiv(Q.TW); Q.d_TH = make_fastdiv(Q.TH);
}

template <int KIND>
int ig_launch(const IgParams& P, dim3 grid, bool wide, bool av4, hipStream_t s, double fl) {
    if constexpr (KIND == 1) {
        if (av4) { GGAN_LAUNCH("conv3d_igemm_k<1, 2, 2, true>", fl, 0, (conv3d_igemm_k<1, 2, 2, true>), grid, dim3(256), 0, s, P); }
        else { GGAN_LAUNCH("conv3d_igemm_k<1, 2, 2, false>", fl, 0, (conv3d_igemm_k<1, 2, 2, false>), grid, dim3(256), 0, s, P); }
    } else if (wide) {
        if (av4) { GGAN_LAUNCH((KIND == 0 ? "conv3d_igemm_k<0, 2, 2, true>" : "conv3d_igemm_k<2, 2, 2, true>"), fl, 0, (conv3d_igemm_k<KIND, 2, 2, true>), grid, dim3(256), 0, s, P); }
        else { GGAN_LAUNCH((KIND == 0 ? "conv3d_igemm_k<0, 2, 2, false>" : "conv3d_igemm_k<2, 2, 2, false>"), fl, 0, (conv3d_igemm_k<KIND, 2, 2, false>), grid, dim3(256), 0, s, P); }
    } else {
        if (av4) { GGAN_LAUNCH((KIND == 0 ? "conv3d_igemm_k<0, 4, 1, true>" : "conv3d_igemm_k<2, 4, 1, true>"), fl, 0, (conv3d_igemm_k<KIND, 4, 1, true>), grid, dim3(256), 0, s, P); }
        else { GGAN_LAUNCH((KIND == 0 ? "conv3d_igemm_k<0, 4, 1, false>" : "conv3d_igemm_k<2, 4, 1, false>"), fl, 0, (conv3d_igemm_k<KIND, 4, 1, false>), grid, dim3(256), 0, s, P); }
    }
    return 0;
}

// split of the reduction so that the launch reaches `target` workgroups (slabs in the workspace).  Several workgroups per CU: the
// index arithmetic of the gathers is VALU work that only another wave's MFMAs can hide
int ig_split(int tiles, int kred, size_t out_elems, size_t ws_floats, int target) {
    int sk = 1;
    if (tiles < target / 2) {
        sk = target / (tiles < 1 ? 1 : tiles);
        const int max_sk = kred / (4 * IKS);
        if (sk > max_sk) sk = max_sk;
        if (sk > 512) sk = 512;
        while (sk > 1 && (size_t)sk * out_elems > ws_floats) --sk;
    }
    return sk < 1 ? 1 : sk;
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

int ggan_conv3d_igemm_ok(const int* dims10, int kind) {
    C3 g;
    if (!dims10 || fill(g, dims10) != 0 || kind < 0 || kind > 2) return 0;
    return ig_ok(g, kind) ? 1 : 0;
}

int ggan_conv3d_fwd(const int* dims10, const float* x, const float* w, const float* bias, float* y, int act, float alpha,
                    void* ws, size_t ws_bytes, ggan_stream_t stream) {
    C3 g;
    GGAN_CHECK_ARG(dims10 && x && w && y && fill(g, dims10) == 0, "bad argument");
    GGAN_CHECK_ARG(ig_ok(g, 0), "geometry not covered by the implicit kernels (ggan_conv3d_igemm_ok)");
    GGAN_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)y)) & 15) == 0, "operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    IgParams P;
    ig_common(P, g);
    ig_fwd_class(P, g);
    const IgClass& Q = P.cls[0];
    P.A = x; P.B = w; P.bias = bias; P.C = y; P.N = g.Co; P.act = act; P.alpha = alpha;
    P.a_bytes = (unsigned)((size_t)g.N * g.L * g.H * g.W * g.Ci * 4); P.b_bytes = (unsigned)((size_t)Q.K * g.Co * 4);
    const bool wide = g.Co > 32;
    const int BM = wide ? 64 : 128, BN = wide ? 64 : 32;
    const int gx = cdiv(g.Co, BN), gy = cdiv(Q.M, BM);
    size_t wsb = ws_bytes;
    float* scratch = (float*)ws_scratch(ws, wsb);
    const size_t out_elems = (size_t)Q.M * g.Co;
    P.SK = ig_split(gx * gy, Q.K, out_elems, scratch ? wsb / 4 : 0, 512);
    P.kps = cdiv(cdiv(Q.K, P.SK), IKS) * IKS;
    P.SK = cdiv(Q.K, P.kps);
    P.slab = out_elems;
    if (P.SK > 1) P.C = scratch;
    const double fl = 2.0 * Q.M * (double)Q.K * g.Co;
    int rc = ig_launch<0>(P, dim3(gy, gx, P.SK), wide, (g.Ci & 3) == 0, s, fl);
    if (rc) return rc;
    if (P.SK > 1) return launch_splitk_reduce(scratch, P.SK, out_elems, y, bias, g.Co, 1, act, alpha, s);
    return 0;
}

int ggan_conv3d_wgrad(const int* dims10, const float* x, const float* gy, float* gw, void* ws, size_t ws_bytes,
                      ggan_stream_t stream) {
    C3 g;
    GGAN_CHECK_ARG(dims10 && x && gy && gw && fill(g, dims10) == 0, "bad argument");
    GGAN_CHECK_ARG(ig_ok(g, 1), "geometry not covered by the implicit kernels (ggan_conv3d_igemm_ok)");
    GGAN_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)gy) | ((uintptr_t)gw)) & 15) == 0, "operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    IgParams P;
    ig_common(P, g);
    ig_fwd_class(P, g);
    const IgClass& Q = P.cls[0];
    P.A = x; P.B = gy; P.C = gw; P.N = g.Co;
    P.a_bytes = (unsigned)((size_t)g.N * g.L * g.H * g.W * g.Ci * 4); P.b_bytes = (unsigned)((size_t)Q.M * g.Co * 4);
    const int gx = cdiv(g.Co, 64), gyt = cdiv(Q.K, 64);
    size_t wsb = ws_bytes;
    float* scratch = (float*)ws_scratch(ws, wsb);
    const size_t out_elems = (size_t)Q.K * g.Co;
    P.SK = ig_split(gx * gyt, Q.M, out_elems, scratch ? wsb / 4 : 0, 1024);
    P.kps = cdiv(cdiv(Q.M, P.SK), IKS) * IKS;
    P.SK = cdiv(Q.M, P.kps);
    P.slab = out_elems;
    if (P.SK > 1) P.C = scratch;
    const double fl = 2.0 * Q.M * (double)Q.K * g.Co;
    int rc = ig_launch<1>(P, dim3(gyt, gx, P.SK), true, (g.Ci & 3) == 0, s, fl);
    if (rc) return rc;
    if (P.SK > 1) return launch_splitk_reduce(scratch, P.SK, out_elems, gw, nullptr, g.Co, 1, GGAN_ACT_NONE, 0.f, s);
    return 0;
}

int ggan_conv3d_dgrad(const int* dims10, const float* gy, const float* w, float* gx, ggan_stream_t stream) {
    C3 g;
    GGAN_CHECK_ARG(dims10 && gy && w && gx && fill(g, dims10) == 0, "bad argument");
    GGAN_CHECK_ARG(ig_ok(g, 2), "geometry not covered by the implicit kernels (ggan_conv3d_igemm_ok)");
    GGAN_CHECK_ARG(((((uintptr_t)gy) | ((uintptr_t)w) | ((uintptr_t)gx)) & 15) == 0, "operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    IgParams P;
    ig_common(P, g);
    P.A = gy; P.B = w; P.C = gx; P.N = g.Ci;
    P.SL = g.Lo; P.SH = g.Ho; P.SW = g.Wo; P.SC = g.Co;
    P.rsl = 1; P.rs = 1; P.ts = -1;
    P.sl = g.sl; P.s = g.s; P.fk = g.k;
    P.OL = g.L; P.OH = g.H; P.OW = g.W;
    P.d_SC = make_fastdiv(g.Co);
    P.a_bytes = (unsigned)((size_t)g.N * g.Lo * g.Ho * g.Wo * g.Co * 4);
    P.b_bytes = (unsigned)((size_t)g.kl * g.k * g.k * g.Ci * g.Co * 4);
    // one axis of a class: residue c -> first tap d0, taps nt, offset q, rows nr
    struct Ax { int d0, nt, q, nr; };
    auto axis = [](int c, int n, int k, int st, int pad) {
        Ax a;
        a.d0 = (c + pad) % st;
        a.nt = a.d0 < k ? (k - a.d0 + st - 1) / st : 0;
        a.q = (c + pad - a.d0) / st;
        a.nr = c < n ? (n - c + st - 1) / st : 0;
        return a;
    };
    int nc = 0, max_m = 0;
    double fl = 0;
    for (int cl = 0; cl < g.sl; ++cl)
        for (int ch = 0; ch < g.s; ++ch)
            for (int cw = 0; cw < g.s; ++cw) {
                const Ax al = axis(cl, g.L, g.kl, g.sl, g.pl), ah = axis(ch, g.H, g.k, g.s, g.ph), aw = axis(cw, g.W, g.k, g.s, g.pw);
                IgClass& Q = P.cls[nc];
                Q.M = g.N * al.nr * ah.nr * aw.nr;
                if (Q.M == 0) continue;
                Q.K = al.nt * ah.nt * aw.nt * g.Co;
                Q.RL = al.nr; Q.RH = ah.nr; Q.RW = aw.nr; Q.TL = al.nt; Q.TH = ah.nt; Q.TW = aw.nt;
                Q.ol = al.q; Q.oh = ah.q; Q.ow = aw.q; Q.cl = cl; Q.ch = ch; Q.cw = cw;
                Q.dl0 = al.d0; Q.dh0 = ah.d0; Q.dw0 = aw.d0;
                Q.d_RW = make_fastdiv(Q.RW); Q.d_RH = make_fastdiv(Q.RH); Q.d_RL = make_fastdiv(Q.RL);
                Q.d_TW = make_fastdiv(Q.TW > 0 ? Q.TW : 1); Q.d_TH = make_fastdiv(Q.TH > 0 ? Q.TH : 1);
                if (Q.M > max_m) max_m = Q.M;
                fl += 2.0 * Q.M * (double)Q.K * g.Ci;
                ++nc;
            }
    const bool wide = g.Ci > 32;
    const int BM = wide ? 64 : 128, BN = wide ? 64 : 32;
    P.kps = 1 << 30;
    return ig_launch<2>(P, dim3(cdiv(max_m, BM), cdiv(g.Ci, BN), nc), wide, true, s, fl);
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
