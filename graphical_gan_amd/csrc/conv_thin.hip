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
//   filter gradient (im2col form)
//       gw[tap*Ci + c, co] = sum_{n, p} x[n, c, 2*oh+kh-pt, 2*ow+kw-pl] * gy[n, co, p]      M = 25*Ci (+1), N = Co, K = pixels
//     The row after the last tap is an all-ones operand row, so the bias gradient sum_p gy[n, co, p] falls out of the same
//     MFMAs (it lands exactly where the split-K slab convention keeps the bias tail).  One workgroup walks a range of
//     (image, 4-row band) items: x rows and the gy tile are prefetched into registers under the MFMAs of the previous item,
//     each wave owns whole 16x16 output tiles (no cross-wave reduction), and the slab leaves through LDS as float4 rows.
//     At most 64 slabs (what the pack kernel / split-K reduce sums), deterministic.
//
//   forward / Deconv2D data gradient (im2col form)
//       y[n, co, p] = act(bias[co] + sum_j x[n, c(j), 2*oh+kh(j)-pt, 2*ow+kw(j)-pl] * W[j, co])      K = 25*Ci (75 -> 80)
//     instead of 8 padded channels x 25 taps = 200.  One workgroup = one image x band of output rows; the x rows sit in LDS
//     once and the A fragments are gathered from them (pixel offset + tap offset), the filter is staged as it lies in memory
//     ([j][Co]), every lane ends up with 4 consecutive pixels of one output channel: 16-byte stores straight into NCHW.
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
    FastDiv d_k4, d_w4, d_W;     // staging / gather index splits (a division by a run-time value is ~35 instructions, ~17 per thread)
};

// NWV waves: 4, or 8 -- waves w and w + 4 share their m-tiles and take the channel tiles t = (w >> 2) + 2 j: a second wave per SIMD under the
// LDS round trips of the MFMA loop, and twice the threads for the staging and gather phases.  (The filter image in LDS has one tile of
// zero rows behind the last real one: with an odd tile count the upper waves multiply it and drop the result.)
template <int NT, bool TWO, int NWV>
__global__ __launch_bounds__(64 * NWV) void thin_dgrad_kernel(const ThinDgradParams P) {
    constexpr int FT = 64 * NWV, NH = NWV / 4, NTW = (NT + NH - 1) / NH, NTP = NTW * NH;     // NTP: tiles incl. the padding one
    static_assert(NWV == 4 || NWV == 8, "4 or 8 waves");
    warm_kernarg(P);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: keeps the MFMA blocks out of exec-mask branches)
    const int l15 = lane & 15, q = lane >> 4;
    const int g0 = blockIdx.x * 2, n = blockIdx.y;
    const int K = P.K, Wo = P.Wo, J = 25 * P.Ci;
    const float mslope = P.mask_act == GGAN_ACT_LRELU ? P.mask_alpha : 0.f;
    float* Ws = smem;                          // [NTP*16][KP]
    float* As = smem + NTP * 16 * P.KP;        // [K][PS]; later T [MT*16][TS]

    // ---- stage the filter (rows >= 25*Ci zero) and the 4 gy rows g0-1 .. g0+2 (rows outside the image zero) ----
    if (!(P.dbg & 4)) {
        const int k4 = K >> 2;
        for (int u = tid; u < NTP * 16 * k4; u += FT) {
            const int j = (int)fdiv((uint32_t)u, P.d_k4), c4 = u - j * k4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < J) v = *reinterpret_cast<const float4*>(P.w + (size_t)j * K + c4 * 4);
            *reinterpret_cast<float4*>(Ws + j * P.KP + c4 * 4) = v;
        }
        const int P4 = 4 * Wo;                 // staged pixels per channel
        const size_t img = (size_t)n * K * P.Ho * Wo;
        if ((Wo & 3) == 0) {
            const int w4 = Wo >> 2;
            for (int u = tid; u < K * 4 * w4; u += FT) {
                const int t = (int)fdiv((uint32_t)u, P.d_w4), c4 = u - t * w4, r = t & 3, k = t >> 2;
                const int oh = g0 - 1 + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((unsigned)oh < (unsigned)P.Ho) {
                    const size_t off = img + ((size_t)k * P.Ho + oh) * Wo + c4 * 4;
                    v = *reinterpret_cast<const float4*>(P.gy + off);
                    if (P.ref) {        // lrelu / relu derivative (the launcher admits no other mask): a select
                        const float4 rf = *reinterpret_cast<const float4*>(P.ref + off);
                        v.x = rf.x > 0.f ? v.x : v.x * mslope;
                        v.y = rf.y > 0.f ? v.y : v.y * mslope;
                        v.z = rf.z > 0.f ? v.z : v.z * mslope;
                        v.w = rf.w > 0.f ? v.w : v.w * mslope;
                    }
                }
                *reinterpret_cast<float4*>(As + k * P.PS + r * Wo + c4 * 4) = v;
            }
        } else {
            for (int u = tid; u < K * P4; u += FT) {
                const int p = u % P4, k = u / P4, r = p / Wo, col = p - r * Wo;
                const int oh = g0 - 1 + r;
                float v = 0.f;
                if ((unsigned)oh < (unsigned)P.Ho) {
                    const size_t off = img + ((size_t)k * P.Ho + oh) * Wo + col;
                    v = P.gy[off];
                    if (P.ref) v = P.ref[off] > 0.f ? v : v * mslope;
                }
                As[k * P.PS + p] = v;
            }
        }
        // pixel slots P4 .. MT*16-1 of every channel (only when 4*Wo is not a multiple of 16)
        const int tail = P.MT * 16 - P4;
        for (int u = tid; u < K * tail; u += FT) {
            const int k = u / tail, p = P4 + (u - k * tail);
            As[k * P.PS + p] = 0.f;
        }
    }
    __syncthreads();

    // ---- T = gy_tile^T x W^T: wave w owns m-tiles w, w+4 ----------------------------------------------------
    const int mw = wave & 3, half = wave >> 2;
    f32x4 acc[2][NTW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (mw < P.MT && !(P.dbg & 1)) {
        // (TWO: MT == 8, every wave owns two m-tiles; otherwise MT <= 4 and one)
        const float* ap = As + q * P.PS + mw * 16 + l15;
        const float* bp = Ws + (half * 16 + l15) * P.KP + q;          // (tile t of this wave: half + NH * t)
        const int nks = K >> 2;
        for (int ks = 0; ks < nks; ++ks) {
            const float a0 = ap[ks * 4 * P.PS];
            float a1 = 0.f;
            if (TWO) a1 = ap[ks * 4 * P.PS + 64];
            float b[NTW];
#pragma unroll
            for (int t = 0; t < NTW; ++t) b[t] = bp[t * NH * 16 * P.KP + ks * 4];
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[t], acc[0][t], 0, 0, 0);
            if (TWO) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[t], acc[1][t], 0, 0, 0);
            }
        }
    }
    __syncthreads();                           // every wave is done reading the gy tile: T may overwrite it
    float* Ts = As;
#pragma unroll
    for (int i = 0; i < (TWO ? 2 : 1); ++i) {
        const int mt = mw + 4 * i;
        if (mt < P.MT) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int tile = half + NH * t;
                if (tile < NT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Ts[(mt * 16 + 4 * q + r) * P.TS + tile * 16 + l15] = acc[i][t][r];
                }
            }
        }
    }
    __syncthreads();

    // ---- gather: output rows 2*g0 .. 2*g0+3; SAME padding (1, 2): 2*oh + kh - 1 = y ----------------------------
    const int W = P.W, Ci = P.Ci;
    for (int idx = tid; idx < Ci * 4 * W; idx += FT) {
        const int t = (int)fdiv((uint32_t)idx, P.d_W), x = idx - t * W, yy = t & 3, c = t >> 2;
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

// ---------------------------------------------------------------------------------------------------------------------
struct ThinWgradParams {
    const float* x;      // [N][Ci][H][W]
    const float* gy;     // [N][Co][Ho][Wo]
    const float* ref;    // activation reference of gy (mask), or null
    float* out;          // slabs [SK][slab_stride] (or gw itself when SK == 1)
    float* bias_direct;  // SK == 1 without slabs: the bias gradient goes here (may be null)
    int N, Ci, H, W, Co, Ho, Wo, pad_t, pad_l;
    int GBR, nb, items, ipw;
    int SR, XRS, XCS, GS, PB, J;
    int xunits, gunits;
    int with_bias, mask_act;
    float mask_alpha;
    size_t out_elems, slab_stride;
    FastDiv d_Wo, d_nb, d_W4, d_SR, d_PB4, d_Ci;
    unsigned x_bytes, gy_bytes;
    int gstride;         // floats between the wave groups' staging areas
};

constexpr int XU_MAX = 3, GU_MAX = 2;

// NTN: 16-wide tiles along Co (2 or 4); the 4 waves are NTN tile columns x (4/NTN) groups of tile rows
// Grid = (image ranges, 16-wide tiles of Co): the slab count (<= 64, what the consumer sums) caps the image ranges, so the output
// COLUMNS are spread over workgroups as well -- every workgroup of a column tile writes its own columns of the same slab.
// NG: groups of 4 waves per workgroup, each walking its own items of the image range (the step is bound by the latency of the
// operand loads: more items in flight); within a group the waves own the 16-row tiles of the 25*Ci (+1) operand rows.  The
// groups' partial tiles are added in group order through LDS at the end.
template <int MPW, int NG>
__global__ __launch_bounds__(NTHR * NG) void thin_wgrad_kernel(const ThinWgradParams P) {
    warm_kernarg(P);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int MG = 4;
    const int tid = threadIdx.x & (NTHR - 1), lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x / NTHR);
    const int l15 = lane & 15, q = lane >> 4;
    const int nt = blockIdx.y, mg = wave;
    float* gsm = smem + (size_t)grp * P.gstride;    // this group's staging area
    float* xs = gsm;                           // [Ci][XCS]: rows SR x (2 halo | W | 2 halo)
    float* gs = gsm + P.Ci * P.XCS;            // [16][GS]: this workgroup's 16 output channels
    const int W4 = P.W >> 2, PB4 = P.PB >> 2, HW = P.H * P.W, HoWo = P.Ho * P.Wo;

    // per-lane operand-row descriptors: row j = tap*Ci + c of the im2col matrix -> offset of (c, kh, kw) in the slab
    // rows past the taps read a constant word kept behind the tiles instead: 1.0 (the bias-gradient row) or 0.0 (padding)
    const int cbase = P.Ci * P.XCS + 16 * P.GS;
    int offA[MPW], pmA[MPW];
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        const int j = (mg + i * MG) * 16 + l15;
        int off = cbase + 1, pm = 0;
        if (j < P.J) {
            const int tap = fdiv(j, P.d_Ci), c = j - tap * P.Ci, kh = tap / 5, kw = tap - kh * 5;
            off = c * P.XCS + kh * P.XRS + kw + 2 - P.pad_l;
            pm = -1;
        } else if (j == P.J && P.with_bias) {
            off = cbase;
        }
        offA[i] = off; pmA[i] = pm;
    }
    if (tid == 0) { gsm[cbase] = 1.f; gsm[cbase + 1] = 0.f; }
    // staging descriptors
    int xrel[XU_MAX], xrow[XU_MAX], xlds[XU_MAX];
#pragma unroll
    for (int j = 0; j < XU_MAX; ++j) {
        const int u = tid + j * NTHR;
        int rel = 0, row = -1, l = 0;
        if (u < P.xunits) {
            const int t = fdiv(u, P.d_W4), f4 = u - t * W4;
            const int c = fdiv(t, P.d_SR);
            row = t - c * P.SR;
            rel = c * HW + row * P.W + f4 * 4;
            l = c * P.XCS + row * P.XRS + 2 + f4 * 4;
        }
        xrel[j] = rel; xrow[j] = row; xlds[j] = l;
    }
    int grel[GU_MAX], gpix[GU_MAX], glds[GU_MAX];
#pragma unroll
    for (int j = 0; j < GU_MAX; ++j) {
        const int u = tid + j * NTHR;
        int rel = 0, pix = -1, l = 0;
        if (u < P.gunits) {
            const int co = fdiv(u, P.d_PB4), p4 = u - co * PB4;
            pix = p4 * 4;
            rel = (nt * 16 + co) * HoWo + pix;
            l = co * P.GS + pix;
        }
        grel[j] = rel; gpix[j] = pix; glds[j] = l;
    }
    for (int e = tid; e < P.Ci * P.XCS; e += NTHR) xs[e] = 0.f;      // halo columns stay zero

    f32x4 acc[MPW];
#pragma unroll
    for (int i = 0; i < MPW; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // branch-free prefetch: raw buffer loads, invalid units aimed past the end of the buffer (hardware returns zeros) --
    // conditional loads would be serialised behind their exec-mask branches, one memory latency each
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned OOB = 0x7FFFFFF0u;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, (short)0, (int)P.x_bytes, 0x00020000);
    const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)P.gy, (short)0, (int)P.gy_bytes, 0x00020000);
    const bool masked = P.ref != nullptr;
    const float mslope = P.mask_act == GGAN_ACT_LRELU ? P.mask_alpha : 0.f;
    const auto rr = __builtin_amdgcn_make_buffer_rsrc((void*)(masked ? P.ref : P.gy), (short)0, (int)P.gy_bytes, 0x00020000);
    u32x4 xreg[XU_MAX], greg[GU_MAX], rreg[GU_MAX];
    auto prefetch = [&](int item) {
        const bool live = item < P.items;
        const int n = fdiv(item, P.d_nb), b = item - n * P.nb;
        const int oh0 = b * P.GBR, in_row0 = 2 * oh0 - P.pad_t;
        const int xbase = n * P.Ci * HW + in_row0 * P.W;
#pragma unroll
        for (int j = 0; j < XU_MAX; ++j) {
            const bool ok = live && xrow[j] >= 0 && (unsigned)(in_row0 + xrow[j]) < (unsigned)P.H;
            xreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(xbase + xrel[j]) * 4u : OOB, 0, 0);
        }
        const int gbase = n * P.Co * HoWo + oh0 * P.Wo;
#pragma unroll
        for (int j = 0; j < GU_MAX; ++j) {
            const bool ok = live && gpix[j] >= 0 && oh0 * P.Wo + gpix[j] < HoWo;
            const unsigned vo = ok ? (unsigned)(gbase + grel[j]) * 4u : OOB;
            greg[j] = __builtin_amdgcn_raw_buffer_load_b128(rg, vo, 0, 0);
            if (masked) rreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rr, vo, 0, 0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < XU_MAX; ++j) {
            if (xrow[j] >= 0) {
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u32x2*>(xs + xlds[j]) = (u32x2){xreg[j].x, xreg[j].y};
                *reinterpret_cast<u32x2*>(xs + xlds[j] + 2) = (u32x2){xreg[j].z, xreg[j].w};
            }
        }
#pragma unroll
        for (int j = 0; j < GU_MAX; ++j) {
            if (gpix[j] >= 0) {
                float4 v = make_float4(__uint_as_float(greg[j].x), __uint_as_float(greg[j].y), __uint_as_float(greg[j].z),
                                       __uint_as_float(greg[j].w));
                if (masked) {       // lrelu / relu derivative (the launcher admits no other mask): a select
                    v.x = __uint_as_float(rreg[j].x) > 0.f ? v.x : v.x * mslope;
                    v.y = __uint_as_float(rreg[j].y) > 0.f ? v.y : v.y * mslope;
                    v.z = __uint_as_float(rreg[j].z) > 0.f ? v.z : v.z * mslope;
                    v.w = __uint_as_float(rreg[j].w) > 0.f ? v.w : v.w * mslope;
                }
                *reinterpret_cast<float4*>(gs + glds[j]) = v;
            }
        }
    };

    // items i_begin + grp, + NG, ...: every group runs the same number of rounds (the barriers are workgroup-wide); a round
    // past the end stages zeros (item index >= N * nb -> every unit out of bounds)
    const int split = blockIdx.x;
    const int i_begin = split * P.ipw, i_end = min(i_begin + P.ipw, P.items);
    const int rounds = (i_end - i_begin + NG - 1) / NG;
    auto item_of = [&](int rd) { const int it = i_begin + rd * NG + grp; return it < i_end ? it : P.items; };
    if (rounds > 0) prefetch(item_of(0));
    const float* bp = gs + l15 * P.GS + q;
    for (int rd = 0; rd < rounds; ++rd) {
        __syncthreads();
        commit();
        __syncthreads();
        if (rd + 1 < rounds) prefetch(item_of(rd + 1));
        // operand reads of k-step ks+1 are issued before the MFMAs of k-step ks (one wave per SIMD: nothing else hides the
        // LDS latency)
        float a[2][MPW], b[2];
        auto load_step = [&](int ks, float* av, float& bv) {
            const int p = ks * 4 + q;
            const int r = fdiv(p, P.d_Wo), ow = p - r * P.Wo;
            const int pix = 2 * r * P.XRS + 2 * ow;
            bv = bp[ks * 4];
#pragma unroll
            for (int i = 0; i < MPW; ++i) av[i] = xs[offA[i] + (pix & pmA[i])];
        };
        load_step(0, a[0], b[0]);
        for (int ks = 0; ks < PB4; ks += 2) {          // PB4 is even (launcher); the last prefetch re-reads the final step
            load_step(ks + 1, a[1], b[1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][i], b[0], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_step(min(ks + 2, PB4 - 1), a[0], b[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][i], b[1], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- [J (+1 bias row)][16] per group through LDS; added in group order and written into this tile's columns of the slab
    //      (row J is the bias gradient: it goes to the bias tail behind the slab) ---------------------------------------------
    __syncthreads();
    const int rows = P.J + (P.with_bias ? 1 : 0);
    const int esz = (P.J + 1) * 16;
    float* es = smem + (size_t)grp * esz;
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        const int mt = mg + i * MG;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = mt * 16 + 4 * q + r;
            if (j < rows) es[j * 16 + l15] = acc[i][r];
        }
    }
    __syncthreads();
    float* dst = P.out + (size_t)split * P.slab_stride;
    float* bdst = P.bias_direct ? P.bias_direct : dst + P.out_elems;
    for (int u = threadIdx.x; u < rows * 4; u += NTHR * NG) {
        const int j = u >> 2, c4 = (u & 3) * 4;
        float4 v = *reinterpret_cast<const float4*>(smem + j * 16 + c4);
#pragma unroll
        for (int g2 = 1; g2 < NG; ++g2) {
            const float4 w2 = *reinterpret_cast<const float4*>(smem + (size_t)g2 * esz + j * 16 + c4);
            v.x += w2.x; v.y += w2.y; v.z += w2.z; v.w += w2.w;
        }
        float* d = (j < P.J ? dst + (size_t)j * P.Co : bdst) + nt * 16 + c4;
        *reinterpret_cast<float4*>(d) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct ThinFwdParams {
    const float* x;      // [N][Ci][H][W]
    const float* w;      // [25*Ci][Co]
    const float* bias;   // [Co] or null
    float* y;            // [N][Co][Ho][Wo]
    int N, Ci, H, W, Co, Ho, Wo, pad_t, pad_l;
    int GBR, nb, SR, XRS, XCS, WS, PB, MT, J, KS;
    int xunits, act;
    float alpha;
    FastDiv d_Wo, d_W4, d_SR, d_Ci, d_5, d_c4;
    unsigned x_bytes;
    int dbg;
    // optional output mask (ggan_conv2d_fwd_masked: the backward of a masked data gradient, the gradient-penalty pass's second derivative
    // through the critic's first layer): y = act_grad(conv, mref, mask_act, mask_alpha) -- mref has y's layout
    const float* mref;
    int mask_act;
    float mask_alpha;
    // optional: the input is an int32 minibatch in a device-resident ring, scaled on the way into LDS (the reference's
    // real_x = 2*((tf.cast(real_x_int, tf.float32)/255.)-.5), gan_inference_cifar10.py:342, in front of Extractor.1): x = mul*(float(v)/div - .5)
    // (+ noise), written to x_out as well (every input row by the one band that owns it) -- no cast launch, no float read of the images
    ThinCastSrc cast;
};

// NWV waves: 4 (one per SIMD), or 8 -- waves w and w + 4 share an m-tile and split the channel tiles, so that every SIMD has a second wave
// to issue under the first one's LDS round trips (the loop is 20 steps of 5 fragment reads + 4 MFMAs).
template <int NTN, int MPW, int NWV>
__global__ __launch_bounds__(64 * NWV) void thin_fwd_kernel(const ThinFwdParams P) {
    constexpr int FT = 64 * NWV, NTW = NTN / (NWV / 4), XUF = (XU_MAX * NTHR + FT - 1) / FT;
    static_assert(NWV == 4 || NWV == 8, "4 or 8 waves");
    warm_kernarg(P);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    const int b = blockIdx.x, n = blockIdx.y;
    float* xs = smem;                                    // [Ci][XCS]: rows SR x (2 halo | W | 2 halo), then {1.0, 0.0}
    const int cbase = P.Ci * P.XCS;
    float* wsm = smem + ((cbase + 2 + 3) & ~3);          // [KS*4][WS]
    const int W4 = P.W >> 2, HW = P.H * P.W, HoWo = P.Ho * P.Wo;
    const int oh0 = b * P.GBR, in_row0 = 2 * oh0 - P.pad_t;

    // ---- stage: zero the slab (halo columns), x rows (hardware-zero-filled out of range), the filter -----------------
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    constexpr unsigned OOB = 0x7FFFFFF0u;
    const bool casting = P.cast.ring != nullptr;         // (uniform)
    const void* xsrc = P.x;
    if (casting) {
        // (nothing in this launch writes the counters: every workgroup sees the same slot -- as cast_scale_ring_k)
        const long long cc = (long long)P.cast.offset + (P.cast.ctr_a ? *P.cast.ctr_a : 0) + (P.cast.ctr_b ? *P.cast.ctr_b : 0);
        const int slot = (int)(((cc % P.cast.nslots) + P.cast.nslots) % P.cast.nslots);
        xsrc = P.cast.ring + (size_t)slot * ((size_t)P.N * P.Ci * HW);
    }
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)xsrc, (short)0, (int)P.x_bytes, 0x00020000);
    const auto rn = __builtin_amdgcn_make_buffer_rsrc((void*)(P.cast.noise ? (const void*)P.cast.noise : xsrc), (short)0, (int)P.x_bytes, 0x00020000);
    u32x4 xreg[XUF], nreg[XUF];
    int xlds[XUF], xgo[XUF];                        // xgo: float index in x_out of a unit this band owns (-1: halo row / not casting)
    bool xin[XUF];
    const int xbase = n * P.Ci * HW + in_row0 * P.W;
#pragma unroll
    for (int j = 0; j < XUF; ++j) {
        const int u = tid + j * FT;
        unsigned vo = OOB;
        int l = -1, go = -1;
        if (u < P.xunits) {
            const int t = fdiv(u, P.d_W4), f4 = u - t * W4;
            const int c = fdiv(t, P.d_SR), row = t - c * P.SR;
            l = c * P.XCS + row * P.XRS + 2 + f4 * 4;
            if ((unsigned)(in_row0 + row) < (unsigned)P.H) {
                vo = (unsigned)(xbase + c * HW + row * P.W + f4 * 4) * 4u;
                if (casting && row >= P.pad_t && row < P.pad_t + 2 * P.GBR) go = xbase + c * HW + row * P.W + f4 * 4;
            }
        }
        xlds[j] = l; xgo[j] = go; xin[j] = vo != OOB;
        xreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0);
        if (casting) nreg[j] = __builtin_amdgcn_raw_buffer_load_b128(rn, P.cast.noise ? vo : OOB, 0, 0);
    }
    for (int e = tid; e < cbase; e += FT) xs[e] = 0.f;
    if (tid == 0) { xs[cbase] = 1.f; xs[cbase + 1] = 0.f; }
    int* jtab = reinterpret_cast<int*>(wsm + P.KS * 4 * P.WS);     // [KS*4] operand-row table (behind the filter)
    if (tid < P.KS * 4) {
        const int j = tid;
        const int tap = fdiv(j, P.d_Ci), c = j - tap * P.Ci, kh = fdiv(tap, P.d_5), kw = tap - kh * 5;
        jtab[j] = j < P.J ? c * P.XCS + kh * P.XRS + kw + 2 - P.pad_l : (int)(0x80000000u | (unsigned)(cbase + 1));
    }
    if (!(P.dbg & 2)) {
        const int c4 = P.Co >> 2, rows = P.KS * 4;
        for (int u = tid; u < rows * c4; u += FT) {
            const int j = (int)fdiv((uint32_t)u, P.d_c4), f = u - j * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < P.J) v = *reinterpret_cast<const float4*>(P.w + (size_t)j * P.Co + f * 4);
            *reinterpret_cast<float4*>(wsm + j * P.WS + f * 4) = v;
        }
    }
    __syncthreads();                                     // zeros before the rows land on top of them
    if (casting) {
#pragma unroll
        for (int j = 0; j < XUF; ++j) {
            // the expression of cast_scale_ring_k, element by element (rows outside the image stay 0: SAME padding pads the SCALED image)
            float4 v;
            v.x = P.cast.mul * (((float)(int)xreg[j].x / P.cast.div) - 0.5f); v.y = P.cast.mul * (((float)(int)xreg[j].y / P.cast.div) - 0.5f);
            v.z = P.cast.mul * (((float)(int)xreg[j].z / P.cast.div) - 0.5f); v.w = P.cast.mul * (((float)(int)xreg[j].w / P.cast.div) - 0.5f);
            if (P.cast.noise) {
                v.x += __uint_as_float(nreg[j].x); v.y += __uint_as_float(nreg[j].y);
                v.z += __uint_as_float(nreg[j].z); v.w += __uint_as_float(nreg[j].w);
            }
            if (!xin[j]) v = make_float4(0.f, 0.f, 0.f, 0.f);
            xreg[j] = (u32x4){__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            if (xgo[j] >= 0) *reinterpret_cast<float4*>(P.cast.x_out + xgo[j]) = v;
        }
    }
#pragma unroll
    for (int j = 0; j < XUF; ++j) {
        if (xlds[j] >= 0) {
            *reinterpret_cast<u32x2*>(xs + xlds[j]) = (u32x2){xreg[j].x, xreg[j].y};
            *reinterpret_cast<u32x2*>(xs + xlds[j] + 2) = (u32x2){xreg[j].z, xreg[j].w};
        }
    }
    __syncthreads();

    const int mw = wave & 3, half = wave >> 2;
    // ---- wave w: m-tiles (w & 3), (w & 3) + 4 (16 pixels each) x its NTW channel tiles ------------------------------------------------
    int pix[MPW];
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        const int mt = mw + 4 * i;
        int p = mt * 16 + l15;
        if (mt >= P.MT) p = 0;                            // (results of a tile past the band are never stored)
        const int r = fdiv(p, P.d_Wo), ow = p - r * P.Wo;
        pix[i] = 2 * r * P.XRS + 2 * ow;
    }
    f32x4 acc[MPW][NTW];
#pragma unroll
    for (int i = 0; i < MPW; ++i)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* bp = wsm + q * P.WS + l15;
    float av[2][MPW], bv[2][NTW];
    // slab offset of operand row j = (tap, channel) -- geometry only: the table built during staging (one entry per row, sign bit = a
    // padding row that reads the constant 0.0), read a step ahead; computed per lane and step it was ~15 VALU instructions in front
    // of every A-fragment read of a loop whose single wave per SIMD has nothing to hide them under
    auto row_entry = [&](int ks) { return jtab[min(ks, P.KS - 1) * 4 + q]; };
    auto load_step = [&](int e, int ks, float* a, float* bb) {
        const int joff = e & 0x7fffffff, pm = ~(e >> 31);
#pragma unroll
        for (int i = 0; i < MPW; ++i) a[i] = xs[joff + (pix[i] & pm)];
#pragma unroll
        for (int t = 0; t < NTW; ++t) bb[t] = bp[ks * 4 * P.WS + (half * NTW + t) * 16];
    };
    int e0 = row_entry(0), e1 = row_entry(1);
    load_step(e0, 0, av[0], bv[0]);
    e0 = row_entry(2);
    for (int ks = 0; ks < ((P.dbg & 1) ? 0 : P.KS); ks += 2) {               // KS is even (K padded to a multiple of 8 with zero rows)
        load_step(e1, ks + 1, av[1], bv[1]);
        e1 = row_entry(ks + 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MPW; ++i)
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][i], bv[0][t], acc[i][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_step(e0, min(ks + 2, P.KS - 1), av[0], bv[0]);
        e0 = row_entry(ks + 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MPW; ++i)
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][i], bv[1][t], acc[i][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- lane (q, l15) holds pixels 4q..4q+3 of channel l15 of each tile: bias, activation, one 16-byte store ----------
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        const int mt = mw + 4 * i;
        const int pp = oh0 * P.Wo + mt * 16 + 4 * q;     // pixel index within the channel plane (a multiple of 4)
        if (mt < P.MT && mt * 16 + 4 * q < P.PB && pp < HoWo && !(P.dbg & 4)) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int co = (half * NTW + t) * 16 + l15;
                const float bs = P.bias ? P.bias[co] : 0.f;
                const size_t o = ((size_t)n * P.Co + co) * HoWo + pp;
                float4 v;
                v.x = act_apply(acc[i][t][0] + bs, P.act, P.alpha);
                v.y = act_apply(acc[i][t][1] + bs, P.act, P.alpha);
                v.z = act_apply(acc[i][t][2] + bs, P.act, P.alpha);
                v.w = act_apply(acc[i][t][3] + bs, P.act, P.alpha);
                if (P.mref) {                            // (uniform)
                    const float4 rf = *reinterpret_cast<const float4*>(P.mref + o);
                    v.x = act_grad(v.x, rf.x, P.mask_act, P.mask_alpha); v.y = act_grad(v.y, rf.y, P.mask_act, P.mask_alpha);
                    v.z = act_grad(v.z, rf.z, P.mask_act, P.mask_alpha); v.w = act_grad(v.w, rf.w, P.mask_act, P.mask_alpha);
                }
                *reinterpret_cast<float4*>(P.y + o) = v;
            }
        }
    }
}

}  // namespace

namespace ggan {

int conv_dgrad_thin(const ggan_conv_geom& g, const float* gy, GyMask m, const float* w, const float* bias, float* gx, int act,
                    float alpha, hipStream_t s) {
    if (g.k != 5 || g.stride != 2 || g.pad_t != 1 || g.pad_l != 1) return 1;
    if (m.act != GGAN_ACT_NONE && m.act != GGAN_ACT_LRELU && m.act != GGAN_ACT_RELU) return 1;   // other masks: general kernels
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
    P.d_k4 = make_fastdiv((uint32_t)(g.Co / 4)); P.d_w4 = make_fastdiv((uint32_t)(g.Wo / 4 > 0 ? g.Wo / 4 : 1)); P.d_W = make_fastdiv((uint32_t)g.W);
    const size_t tile = (size_t)g.Co * P.PS, tt = (size_t)P.MT * 16 * P.TS;
    static const int w8 = [] { const char* e = getenv("GGAN_THIN_DGRAD_W8"); return e ? atoi(e) : 1; }();
    const int ntp = w8 ? 2 * cdiv(NT, 2) : NT;      // (eight waves: the filter image padded to an even number of tiles)
    const size_t shmem = ((size_t)ntp * 16 * P.KP + (tile > tt ? tile : tt)) * sizeof(float);
    if (shmem > 64 * 1024) return 1;
    const double fl = 2.0 * g.N * g.Co * g.Ho * g.Wo * (double)g.Ci * 25.0;
    const dim3 grid(g.Ho / 2, g.N);
    if (P.MT > 4 && P.MT != 8) return 1;       // one m-tile per wave, or exactly two
#define THIN_DGRAD(NT_) \
    if (w8 && P.MT == 8) { GGAN_LAUNCH("thin_dgrad_kernel", fl, 0, (thin_dgrad_kernel<NT_, true, 8>), grid, dim3(512), shmem, s, P); } \
    else if (w8) { GGAN_LAUNCH("thin_dgrad_kernel", fl, 0, (thin_dgrad_kernel<NT_, false, 8>), grid, dim3(512), shmem, s, P); } \
    else if (P.MT == 8) { GGAN_LAUNCH("thin_dgrad_kernel", fl, 0, (thin_dgrad_kernel<NT_, true, 4>), grid, dim3(NTHR), shmem, s, P); } \
    else { GGAN_LAUNCH("thin_dgrad_kernel", fl, 0, (thin_dgrad_kernel<NT_, false, 4>), grid, dim3(NTHR), shmem, s, P); }
    switch (NT) {
        case 2: THIN_DGRAD(2); break;
        case 4: THIN_DGRAD(4); break;
        case 5: THIN_DGRAD(5); break;
        case 7: THIN_DGRAD(7); break;
        default: return 1;
    }
#undef THIN_DGRAD
    return 0;
}

int conv_wgrad_thin(const ggan_conv_geom& g, const float* x, const float* gy, GyMask m, float* gw, float* gbias, void* ws,
                    size_t ws_bytes, hipStream_t s, WgradParts* parts) {
    if (g.k != 5 || g.stride != 2 || g.pad_l < 1 || g.pad_l > 2 || g.Ci > 4 || (g.W & 3)) return 1;
    if ((g.Co & 15) || g.Co > 256) return 1;
    if (m.act != GGAN_ACT_NONE && m.act != GGAN_ACT_LRELU && m.act != GGAN_ACT_RELU) return 1;   // other masks: general kernels
    if (((g.Ho * g.Wo) & 3) || getenv("GGAN_NO_THIN")) return 1;
    if ((((uintptr_t)x) & 15) || (((uintptr_t)gy) & 15) || (m.act != GGAN_ACT_NONE && (((uintptr_t)m.ref) & 15))) return 1;
    const bool with_bias = parts ? parts->with_bias != 0 : gbias != nullptr;
    const size_t xb = (size_t)g.N * g.Ci * g.H * g.W * 4, gb = (size_t)g.N * g.Co * g.Ho * g.Wo * 4;
    if (xb >= 0x7FFFFFF0ull || gb >= 0x7FFFFFF0ull) return 1;
    ThinWgradParams P;
    memset(&P, 0, sizeof(P));
    P.x = x; P.gy = gy; P.ref = m.act != GGAN_ACT_NONE ? m.ref : nullptr;
    P.mask_act = m.act; P.mask_alpha = m.alpha;
    P.x_bytes = (unsigned)xb; P.gy_bytes = (unsigned)gb;
    P.N = g.N; P.Ci = g.Ci; P.H = g.H; P.W = g.W; P.Co = g.Co; P.Ho = g.Ho; P.Wo = g.Wo; P.pad_t = g.pad_t; P.pad_l = g.pad_l;
    P.GBR = g.Ho < 4 ? g.Ho : 4;
    while (P.GBR > 1 && (((P.GBR * g.Wo) & 7) || P.GBR * g.Wo > 128)) --P.GBR;
    P.PB = P.GBR * g.Wo;
    if ((P.PB & 7) || P.PB > 128) return 1;      // whole k-steps of 4 pixels, an even number of them
    P.nb = cdiv(g.Ho, P.GBR);
    P.items = g.N * P.nb;
    P.SR = 2 * P.GBR + 3;
    P.XRS = g.W + 4;
    P.XCS = P.SR * P.XRS;
    P.GS = P.PB + 4;
    P.J = 25 * g.Ci;
    P.with_bias = with_bias ? 1 : 0;
    P.xunits = g.Ci * P.SR * (g.W / 4);
    P.gunits = 16 * (P.PB / 4);
    if (P.xunits > XU_MAX * NTHR || P.gunits > GU_MAX * NTHR) return 1;
    P.out_elems = (size_t)P.J * g.Co;
    P.slab_stride = P.out_elems + (with_bias ? (size_t)g.Co : 0);
    size_t cap_slabs;
    float* slabs;
    if (parts) {
        slabs = parts->buf;
        cap_slabs = parts->cap_floats / P.slab_stride;
        if (cap_slabs < 1) { set_error("conv_wgrad_thin: partial-slab buffer too small"); return -1; }
    } else {
        ws = ws_scratch(ws, ws_bytes);
        slabs = (float*)ws;
        cap_slabs = ws ? ws_bytes / (P.slab_stride * sizeof(float)) : 0;
    }
    // (slab cap 64 keeps the consumer of the slabs cheap; with thousands of items -- the 512..1024-frame launches of the
    //  state-space scripts, 134 MB of gy to stream -- 64 x Co/16 workgroups leave half the chip idle: 256 small slabs then)
    //  (round 6: a 32-channel first layer has only Co / 16 = 2 column groups -- 64 slabs were 128 workgroups, half of the chip, for the
    //   launch that ends the face critic's backward pass alone on the chip: 128 slabs there)
    int max_sk = (P.items >= 2048 && P.slab_stride <= 8192) ? 256 : 64;
    if (max_sk == 64 && g.Co / 16 <= 2 && P.slab_stride <= 8192 && !getenv("GGAN_THIN_WGRAD_SK64")) max_sk = 128;
    int sk = P.items < max_sk ? P.items : max_sk;
    if ((size_t)sk > cap_slabs) sk = (int)cap_slabs;
    if (sk < 1) sk = 1;
    P.ipw = cdiv(P.items, sk);
    const int SK = cdiv(P.items, P.ipw);
    if (SK > 1 || parts) {
        P.out = slabs;
        P.bias_direct = nullptr;
    } else {
        P.out = gw;
        P.bias_direct = gbias;
        P.slab_stride = 0;
    }
    P.d_Wo = make_fastdiv(g.Wo); P.d_nb = make_fastdiv(P.nb); P.d_W4 = make_fastdiv(g.W / 4); P.d_SR = make_fastdiv(P.SR);
    P.d_PB4 = make_fastdiv(P.PB / 4); P.d_Ci = make_fastdiv(g.Ci);
    const int NTM = cdiv(P.J + (with_bias ? 1 : 0), 16);
    const int MPW = cdiv(NTM, 4);
    if (MPW > 2) return 1;
    constexpr int NG = 4;
    size_t stage = (size_t)g.Ci * P.XCS + (size_t)16 * P.GS + 4;
    stage = (stage + 3) & ~(size_t)3;
    P.gstride = (int)stage;
    const size_t ep = (size_t)(P.J + 1) * 16;
    const size_t shmem = NG * (stage > ep ? stage : ep) * sizeof(float);
    if (shmem > 160 * 1024) return 1;
    const double fl = 2.0 * g.N * g.Co * g.Ho * g.Wo * (double)g.Ci * 25.0;
    const dim3 grid(SK, g.Co / 16);
#define THIN_WGRAD(MPW_)                                                                                                    \
    do {                                                                                                                    \
        static std::atomic<unsigned long long> attr_set{0};                                                                 \
        if (first_on_device(attr_set)) {                                                                                    \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(thin_wgrad_kernel<MPW_, NG>),                           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                              \
        }                                                                                                                   \
    } while (0);                                                                                                            \
    GGAN_LAUNCH("thin_wgrad_kernel", fl, 0, (thin_wgrad_kernel<MPW_, NG>), grid, dim3(NTHR * NG), shmem, s, P)
    if (MPW == 1) { THIN_WGRAD(1); }
    else { THIN_WGRAD(2); }
#undef THIN_WGRAD
    if (parts) {
        parts->n = SK;
        parts->stride = P.out_elems + (with_bias ? (size_t)g.Co : 0);
        return 0;
    }
    if (SK > 1)
        return launch_splitk_reduce(slabs, SK, P.out_elems, gw, nullptr, 1, 1, GGAN_ACT_NONE, 0.f, s, P.slab_stride, gbias,
                                    gbias ? (size_t)g.Co : 0);
    return 0;
}

int conv_fwd_thin(const ggan_conv_geom& g, const float* x, const float* w, const float* bias, float* y, int act, float alpha,
                  hipStream_t s, const ThinCastSrc* cast, const OutMask* mask) {
    if (g.k != 5 || g.stride != 2 || g.pad_l < 1 || g.pad_l > 2 || g.Ci > 4 || (g.W & 3)) return 1;
    if ((g.Co != 32 && g.Co != 64) || ((g.Ho * g.Wo) & 3) || getenv("GGAN_NO_THIN")) return 1;
    if (cast) {
        x = cast->x_out;       // (alignment / size checks below: the ring slots and the noise have the float tensor's extent)
        if ((((uintptr_t)cast->ring) & 15) || (cast->noise && (((uintptr_t)cast->noise) & 15)) || (((size_t)g.N * g.Ci * g.H * g.W) & 3)) return 1;
    }
    if ((((uintptr_t)x) & 15) || (((uintptr_t)w) & 15) || (((uintptr_t)y) & 15)) return 1;
    const size_t xb = (size_t)g.N * g.Ci * g.H * g.W * 4;
    if (xb >= 0x7FFFFFF0ull) return 1;
    ThinFwdParams P;
    memset(&P, 0, sizeof(P));
    P.x = x; P.w = w; P.bias = bias; P.y = y; P.act = act; P.alpha = alpha; P.x_bytes = (unsigned)xb;
    if (cast) P.cast = *cast;
    if (mask) {
        if ((((uintptr_t)mask->ref) & 15) || !mask->ref) return 1;
        P.mref = mask->ref; P.mask_act = mask->act; P.mask_alpha = mask->alpha;
    }
    { const char* d = getenv("GGAN_THIN_DBG"); P.dbg = d ? atoi(d) : 0; }
    P.N = g.N; P.Ci = g.Ci; P.H = g.H; P.W = g.W; P.Co = g.Co; P.Ho = g.Ho; P.Wo = g.Wo; P.pad_t = g.pad_t; P.pad_l = g.pad_l;
    P.d_c4 = make_fastdiv((uint32_t)(g.Co / 4 > 0 ? g.Co / 4 : 1));
    // band of output rows: whole 16-pixel tiles, at most 8 of them; the smallest band that still gives every wave a tile
    // (more workgroups to spread over the chip), one that divides the image if possible
    P.GBR = 0;
    for (int pass = 0; pass < 2 && !P.GBR; ++pass)
        for (int r = 1; r <= g.Ho && r * g.Wo <= 128; ++r)
            if (((r * g.Wo) & 15) == 0 && r * g.Wo >= 64 && (pass == 1 || g.Ho % r == 0)) { P.GBR = r; break; }
    if (!P.GBR)
        for (int r = 1; r <= g.Ho && r * g.Wo <= 128; ++r)
            if (((r * g.Wo) & 15) == 0) P.GBR = r;
    if (!P.GBR) return 1;
    P.PB = P.GBR * g.Wo;
    P.MT = P.PB / 16;
    P.nb = cdiv(g.Ho, P.GBR);
    P.SR = 2 * P.GBR + 3;
    P.XRS = g.W + 4;
    P.XCS = P.SR * P.XRS;
    P.J = 25 * g.Ci;
    P.KS = 2 * cdiv(P.J, 8);
    P.WS = g.Co + 16;                                        // == 16 (mod 32): the four k-rows of a B fragment on disjoint bank halves
    P.xunits = g.Ci * P.SR * (g.W / 4);
    if (P.xunits > XU_MAX * NTHR) return 1;
    P.d_Wo = make_fastdiv(g.Wo); P.d_W4 = make_fastdiv(g.W / 4); P.d_SR = make_fastdiv(P.SR); P.d_Ci = make_fastdiv(g.Ci);
    P.d_5 = make_fastdiv(5);
    const size_t shmem = ((((size_t)g.Ci * P.XCS + 2 + 3) & ~(size_t)3) + (size_t)P.KS * 4 * P.WS + (size_t)P.KS * 4) * sizeof(float);
    if (shmem > 64 * 1024 || P.KS * 4 > NTHR) return 1;
    const int NTN = g.Co / 16, MPW = cdiv(P.MT, 4);
    const double fl = 2.0 * g.N * g.Co * g.Ho * g.Wo * (double)g.Ci * 25.0;
    const dim3 grid(P.nb, g.N);
    // (eight waves where a wave still owns two channel tiles: 3->64 @32 at 64 / 128 images 10.3 -> 9.2 / 13.1 -> 12.3 us; with 32 output
    //  channels a wave would be left with ONE tile -- two fragment reads per MFMA -- and the face layer measured 12.3 -> 13.9 us)
    static const int w8 = [] { const char* e = getenv("GGAN_THIN_FWD_W8"); return e ? atoi(e) : 1; }();
    if (w8 && NTN == 4 && MPW == 1) { GGAN_LAUNCH("thin_fwd_kernel", fl, 0, (thin_fwd_kernel<4, 1, 8>), grid, dim3(512), shmem, s, P); }
    else if (w8 && NTN == 4 && MPW == 2) { GGAN_LAUNCH("thin_fwd_kernel", fl, 0, (thin_fwd_kernel<4, 2, 8>), grid, dim3(512), shmem, s, P); }
    else if (NTN == 4 && MPW == 1) { GGAN_LAUNCH("thin_fwd_kernel", fl, 0, (thin_fwd_kernel<4, 1, 4>), grid, dim3(NTHR), shmem, s, P); }
    else if (NTN == 4 && MPW == 2) { GGAN_LAUNCH("thin_fwd_kernel", fl, 0, (thin_fwd_kernel<4, 2, 4>), grid, dim3(NTHR), shmem, s, P); }
    else if (NTN == 2 && MPW == 1) { GGAN_LAUNCH("thin_fwd_kernel", fl, 0, (thin_fwd_kernel<2, 1, 4>), grid, dim3(NTHR), shmem, s, P); }
    else if (NTN == 2 && MPW == 2) { GGAN_LAUNCH("thin_fwd_kernel", fl, 0, (thin_fwd_kernel<2, 2, 4>), grid, dim3(NTHR), shmem, s, P); }
    else return 1;
    return 0;
}

}  // namespace ggan
