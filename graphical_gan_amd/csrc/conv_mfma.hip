// fp32-MFMA convolution kernels for gfx950 (CDNA4): the Conv2D / Deconv2D hot path.
//
// Two kernels cover the six conv-like ops of the training step:
//
//  corr_kernel  "strided correlation":  out[n,cn,U,V] = sum_{ck,i,j} in[n,ck,SU*u+DI*i+ou, SU*v+DI*j+ov] * w[i,j,ck,cn]
//     MODE 0  Conv2D forward / Deconv2D data-gradient:   SU=stride(2), DI=+1, 5x5 taps, ck=Ci, cn=Co
//     MODE 1  Conv2D data-gradient / Deconv2D forward:   the stride-2 transposed conv is split into its 4 output
//             parity classes; each class is a dense stride-1 correlation (DI=-1) with a 3x3/3x2/2x3/2x2 sub-filter,
//             so MFMA never multiplies the structural zeros of the zero-insertion formulation (SURVEY.md "hard parts").
//     Implicit GEMM per workgroup: 64 pixels x 64 output channels, K = ck x taps, v_mfma_f32_32x32x2_f32.
//     LDS staging: the raw input patch of the pixel tile (zero halo = TF SAME padding, so the inner loop has no
//     bounds checks and every fragment address is lane_base + uniform offset) and the filter slice [tap][ck][cn].
//     The MFMA A operand is the FILTER (rows = cn) and B the pixels (cols), so the accumulator's lane index runs
//     along pixels and every store is a coalesced run of NCHW floats.  Next chunk's global loads are issued into
//     registers before the MFMA block of the current chunk (async-STAGE split) and written to LDS after it.
//     Small-GEMM problem sizes (M=1024..16384 pixels) are spread over the 256 CUs by split-K over ck with a
//     deterministic partial-slab reduction (no atomics => bitwise reproducible).
//
//  wgrad_kernel  filter gradient:  gw[kh,kw,ci,co] = sum_{n,oh,ow} x[n,ci,S*oh+kh-pt,S*ow+kw-pl] * gy[n,co,oh,ow]
//     GEMM M=ci, N=co per tap, K = pixels.  v_mfma_f32_16x16x4_f32 so that ONE wave holds all 25 taps of a 16x16
//     (ci,co) tile in 100 accumulator registers: each k-step (4 pixels) costs 1 gy-fragment read, 25 x-fragment
//     reads (lane_base + immediate tap offset into the x slab) and 25 MFMAs.  Split-K over image groups.
//
// Exact fp32: MFMA f32 is a k-ordered fmaf chain (cdna_hip_programming.md section 3), so results match an fp32
// CPU implementation up to summation order.
#include "common.h"
#include "conv.h"
#include <stdlib.h>
using namespace ggan;

namespace {

constexpr int TN = 64;   // output channels per workgroup tile
constexpr int TM = 64;   // pixels per workgroup tile
constexpr int CK = 4;    // reduction channels staged per LDS chunk (2 MFMA k-pairs per tap)
constexpr int XE_MAX = 12;

struct CorrClass {
    int Hu, Wv;          // pixel grid of this class
    int ou, ov;          // input row = SU*u + DI*i + ou
    int or0, oc0;        // output row = or0 + ors*u
    int tiles_r, tiles_c;
    long wbase;
};

struct CorrParams {
    const float* in;
    const float* w;
    const float* bias;
    float* out;
    int N, CKtot, Hin, Win;
    int CNtot, Hout, Wout;
    int ors, ocs;
    long w_si, w_sj, w_sk, w_sn;
    int TR, TC, TI;
    int SR, SCp, CS;
    FastDiv d_CS, d_SRSC, d_SCp, d_TRTC, d_TC;
    int img_groups, cps, SK;
    int act;
    float alpha;
    size_t out_elems;
    CorrClass cls[4];
};

template <int TH, int TW, int SU, int DI, bool WK>
__device__ __forceinline__ void corr_body(const CorrParams& P, const CorrClass& c, const int split, float* smem) {
    constexpr int NT = TH * TW;
    constexpr int WUNITS = WK ? NT * TN : NT * CK * (TN / 4);
    constexpr int WE = (WUNITS + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int half = lane >> 5, l31 = lane & 31;

    // ---- which tile ---------------------------------------------------------------------------
    const int tiles_per_img = c.tiles_r * c.tiles_c;
    const int ig = blockIdx.x / tiles_per_img;
    if (ig >= P.img_groups) return;
    const int tt = blockIdx.x - ig * tiles_per_img;
    const int tr = tt / c.tiles_c, tc = tt - tr * c.tiles_c;
    const int n0 = ig * P.TI, u0 = tr * P.TR, v0 = tc * P.TC;
    const int cn0 = blockIdx.y * TN;
    const int ck_begin = split * P.cps;
    const int ck_end = min(ck_begin + P.cps, P.CKtot);
    const int in_row0 = SU * u0 + c.ou - (DI < 0 ? TH - 1 : 0);
    const int in_col0 = SU * v0 + c.ov - (DI < 0 ? TW - 1 : 0);
    const int HWin = P.Hin * P.Win;

    float* xs = smem;                                   // [CK][CS]
    float* ws = smem + ((CK * P.CS + 3) & ~3);          // [NT][CK][TN]

    // ---- per-thread staging descriptors (fixed for all chunks) ----------------------------------
    int xoff[XE_MAX];
    int xck[XE_MAX];
    const int xe_cnt = CK * P.CS;
#pragma unroll
    for (int j = 0; j < XE_MAX; ++j) {
        const int e = tid + j * 256;
        int off = -1, ckl = 0;
        if (e < xe_cnt) {
            ckl = fdiv(e, P.d_CS);
            const int r1 = e - ckl * P.CS;
            const int img = fdiv(r1, P.d_SRSC);
            const int r2 = r1 - img * (P.SR * P.SCp);
            const int r = fdiv(r2, P.d_SCp);
            const int cc = r2 - r * P.SCp;
            const int ih = in_row0 + r, iw = in_col0 + cc, n = n0 + img;
            if (n < P.N && ih >= 0 && ih < P.Hin && iw >= 0 && iw < P.Win)
                off = ((n * P.CKtot + ckl) * P.Hin + ih) * P.Win + iw;
        }
        xoff[j] = off;
        xck[j] = ckl;
    }
    long woff[WE];
    int wlds[WE];
    int wck[WE];
#pragma unroll
    for (int q = 0; q < WE; ++q) {
        const int u = tid + q * 256;
        long off = -1;
        int l = 0, ckl = 0;
        if (u < WUNITS) {
            if (WK) {   // unit = (tap, cn): 4 consecutive ck in global memory
                const int tap = u / TN, cn = u % TN;
                const int i = tap / TW, j = tap % TW;
                if (cn0 + cn < P.CNtot) off = c.wbase + i * P.w_si + j * P.w_sj + (long)(cn0 + cn) * P.w_sn;
                l = tap * CK * TN + cn;
            } else {    // unit = (tap, ck, cn4): 4 consecutive cn in global memory
                const int tap = u / (CK * (TN / 4));
                const int rem = u % (CK * (TN / 4));
                ckl = rem / (TN / 4);
                const int cn4 = rem % (TN / 4);
                const int i = tap / TW, j = tap % TW;
                if (cn0 + cn4 * 4 < P.CNtot)
                    off = c.wbase + i * P.w_si + j * P.w_sj + (long)ckl * P.w_sk + (long)(cn0 + cn4 * 4) * P.w_sn;
                l = (tap * CK + ckl) * TN + cn4 * 4;
            }
        }
        woff[q] = off;
        wlds[q] = l;
        wck[q] = ckl;
    }

    // ---- per-lane MFMA fragment bases -------------------------------------------------------------
    int pixbase;
    bool pix_ok;
    int o_off = 0;   // output offset of this lane's pixel (without the channel term)
    {
        const int p = wm * 32 + l31;
        const int img = fdiv(p, P.d_TRTC);
        const int rem = p - img * (P.TR * P.TC);
        const int ur = fdiv(rem, P.d_TC);
        const int vc = rem - ur * P.TC;
        pix_ok = img < P.TI && (n0 + img) < P.N && (u0 + ur) < c.Hu && (v0 + vc) < c.Wv;
        const int b = pix_ok ? img * (P.SR * P.SCp) + SU * ur * P.SCp + SU * vc : 0;
        pixbase = b + half * P.CS + (DI < 0 ? (TH - 1) * P.SCp + (TW - 1) : 0);
        if (pix_ok)
            o_off = (((n0 + img) * P.CNtot) * P.Hout + (c.or0 + P.ors * (u0 + ur))) * P.Wout + (c.oc0 + P.ocs * (v0 + vc));
    }
    const int wfrag = half * TN + wn * 32 + l31;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    float xreg[XE_MAX];
    float4 wreg[WE];

    auto prefetch = [&](int ck0) {
        const float* inb = P.in + (size_t)ck0 * HWin;
#pragma unroll
        for (int j = 0; j < XE_MAX; ++j) {
            const bool ok = xoff[j] >= 0 && (ck0 + xck[j]) < ck_end;
            const float v = inb[ok ? xoff[j] : 0 - ck0 * HWin];   // always a valid address (in[0])
            xreg[j] = ok ? v : 0.f;
        }
#pragma unroll
        for (int q = 0; q < WE; ++q) {
            bool ok;
            long o;
            if (WK) {
                ok = woff[q] >= 0 && ck0 < ck_end;
                o = ok ? woff[q] + (long)ck0 * P.w_sk : 0;
            } else {
                ok = woff[q] >= 0 && (ck0 + wck[q]) < ck_end;
                o = ok ? woff[q] + (long)ck0 * P.w_sk : 0;
            }
            const float4 v = *reinterpret_cast<const float4*>(P.w + o);
            wreg[q] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < XE_MAX; ++j) {
            const int e = tid + j * 256;
            if (e < xe_cnt) xs[e] = xreg[j];
        }
#pragma unroll
        for (int q = 0; q < WE; ++q) {
            const int u = tid + q * 256;
            if (u < WUNITS) {
                if (WK) {
                    float4 v = wreg[q];
                    ws[wlds[q] + 0 * TN] = v.x;
                    ws[wlds[q] + 1 * TN] = v.y;
                    ws[wlds[q] + 2 * TN] = v.z;
                    ws[wlds[q] + 3 * TN] = v.w;
                } else {
                    *reinterpret_cast<float4*>(ws + wlds[q]) = wreg[q];
                }
            }
        }
    };

    // ---- main loop over reduction-channel chunks -------------------------------------------------
    prefetch(ck_begin);
    for (int ck0 = ck_begin; ck0 < ck_end; ck0 += CK) {
        __syncthreads();          // previous chunk's fragment reads are done
        commit();
        __syncthreads();
        if (ck0 + CK < ck_end) prefetch(ck0 + CK);
#pragma unroll
        for (int i = 0; i < TH; ++i) {
#pragma unroll
            for (int j = 0; j < TW; ++j) {
#pragma unroll
                for (int cp = 0; cp < CK / 2; ++cp) {
                    const float a = ws[((i * TW + j) * CK + cp * 2) * TN + wfrag];
                    const float b = xs[cp * 2 * P.CS + pixbase + DI * (i * P.SCp + j)];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: lanes run along pixels => coalesced NCHW stores -----------------------------------
    if (!pix_ok) return;
    const bool direct = P.SK == 1;
    float* outp = direct ? P.out : P.out + (size_t)split * P.out_elems;   // P.out = partial slab when SK > 1
    const int chw = P.Hout * P.Wout;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cn = cn0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (cn < P.CNtot) {
            float v = acc[r];
            if (direct) {
                if (P.bias) v += P.bias[cn];
                v = act_apply(v, P.act, P.alpha);
            }
            outp[(size_t)o_off + (size_t)cn * chw] = v;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void corr_kernel(const CorrParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cls = blockIdx.z / P.SK, split = blockIdx.z % P.SK;
    if (MODE == 0) {
        corr_body<5, 5, 2, 1, false>(P, P.cls[0], split, smem);
    } else {
        switch (cls) {
            case 0: corr_body<3, 3, 1, -1, true>(P, P.cls[0], split, smem); break;
            case 1: corr_body<3, 2, 1, -1, true>(P, P.cls[1], split, smem); break;
            case 2: corr_body<2, 3, 1, -1, true>(P, P.cls[2], split, smem); break;
            default: corr_body<2, 2, 1, -1, true>(P, P.cls[3], split, smem); break;
        }
    }
}

// out[idx] = act(sum_s partial[s][idx] + bias[c])
__global__ void splitk_reduce_k(const float* __restrict__ partial, int SK, size_t elems, float* __restrict__ out,
                                const float* __restrict__ bias, int C, int HW, int act, float alpha) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < SK; ++k) s += partial[(size_t)k * elems + i];
        if (bias) s += bias[(i / (size_t)HW) % (size_t)C];
        out[i] = act_apply(s, act, alpha);
    }
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// choose the pixel tile: TI images x TR rows x TC cols <= 64 pixels
void pick_tile(int Hu, int Wv, int N, int& TR, int& TC, int& TI) {
    TC = Wv < TM ? Wv : TM;
    TR = TM / TC;
    if (TR < 1) TR = 1;
    if (TR > Hu) TR = Hu;
    TI = TM / (TR * TC);
    if (TI < 1) TI = 1;
    if (TI > N) TI = N;
}

int pick_splitk(int base_wgs, int CKtot, const char* envname) {
    int sk = env_int(envname, 0);
    if (sk <= 0) {
        const int target = env_int("GGAN_TARGET_WGS", 512);
        sk = target / (base_wgs > 0 ? base_wgs : 1);
        const int max_sk = CKtot / (2 * CK);   // at least 2 chunks per split
        if (sk > max_sk) sk = max_sk;
        if (sk > 16) sk = 16;
    }
    if (sk < 1) sk = 1;
    return sk;
}

}  // namespace
namespace ggan {
int launch_splitk_reduce(const float* partial, int SK, size_t elems, float* out, const float* bias, int C, int HW, int act,
                         float alpha, hipStream_t s) {
    size_t b = (elems + 255) / 256;
    if (b > 2048) b = 2048;
    GGAN_LAUNCH("conv_splitk_reduce", 0, 4.0 * elems * (SK + 1), splitk_reduce_k, dim3((int)b), dim3(256), 0, s, partial,
                SK, elems, out, bias, C, HW, act, alpha);
    return 0;
}
}  // namespace ggan
namespace {
inline int launch_reduce(const float* partial, int SK, size_t elems, float* out, const float* bias, int C, int HW, int act,
                         float alpha, hipStream_t s) {
    return ggan::launch_splitk_reduce(partial, SK, elems, out, bias, C, HW, act, alpha, s);
}

// ================================================================================================
// filter gradient
// ================================================================================================
constexpr int WG_CI = 32, WG_CO = 32;   // (ci, co) tile per workgroup: 2x2 waves of 16x16

struct WgradParams {
    const float* x;
    const float* gy;
    float* out;   // gw or partial slabs
    int N, Ci, H, W, Co, Ho, Wo;
    int pad_t, pad_l;
    int TR, TI;          // pixel chunk = TI images x TR rows x Wo cols (<= 64, multiple of 4)
    int SR, SCp, CS;     // x slab per channel (CS odd)
    int PC, PCp;         // pixels per chunk, padded gy row stride (odd)
    FastDiv d_CS, d_SRSC, d_SCp, d_PC, d_TRWo, d_Wo;
    int row_tiles;       // ceil(Ho / TR)
    int chunks_total;    // ceil(N / TI) * row_tiles
    int chunks_per_split;
    int SK;
    size_t out_elems;    // 25*Ci*Co
};

template <int KS, int S>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = KS * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 1, wj = wave >> 1;
    const int l15 = lane & 15, q = lane >> 4;
    const int ci0 = blockIdx.x * WG_CI, co0 = blockIdx.y * WG_CO, split = blockIdx.z;

    float* xs = smem;                      // [WG_CI][CS]
    float* gs = smem + WG_CI * P.CS;       // [WG_CO][PCp]

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int xa = (wi * 16 + l15) * P.CS + S * q;     // x-fragment lane base (pixel q of the 4-pixel k-step)
    const int gb = (wj * 16 + l15) * P.PCp + q;        // gy-fragment lane base
    const int HW = P.H * P.W, HoWo = P.Ho * P.Wo;
    const int xe_cnt = WG_CI * P.CS, ge_cnt = WG_CO * P.PC;

    const int c_begin = split * P.chunks_per_split;
    const int c_end = min(c_begin + P.chunks_per_split, P.chunks_total);
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int ig = ch / P.row_tiles, rt = ch - ig * P.row_tiles;
        const int n0 = ig * P.TI, oh0 = rt * P.TR;
        const int in_row0 = S * oh0 - P.pad_t, in_col0 = -P.pad_l;
        __syncthreads();
        // ---- stage the x slab (zero halo) and the gy tile -------------------------------------------
        for (int e = tid; e < xe_cnt; e += 256) {
            const int cil = fdiv(e, P.d_CS);
            const int r1 = e - cil * P.CS;
            const int img = fdiv(r1, P.d_SRSC);
            const int r2 = r1 - img * (P.SR * P.SCp);
            const int r = fdiv(r2, P.d_SCp);
            const int cc = r2 - r * P.SCp;
            const int ih = in_row0 + r, iw = in_col0 + cc, n = n0 + img, ci = ci0 + cil;
            const bool ok = img < P.TI && n < P.N && ci < P.Ci && ih >= 0 && ih < P.H && iw >= 0 && iw < P.W;
            const float v = P.x[ok ? ((size_t)(n * P.Ci + ci) * HW + ih * P.W + iw) : 0];
            xs[e] = ok ? v : 0.f;
        }
        for (int e = tid; e < ge_cnt; e += 256) {
            const int col = fdiv(e, P.d_PC);
            const int p = e - col * P.PC;
            const int img = fdiv(p, P.d_TRWo);
            const int rem = p - img * (P.TR * P.Wo);
            const int r = fdiv(rem, P.d_Wo);
            const int cc = rem - r * P.Wo;
            const int n = n0 + img, oh = oh0 + r, co = co0 + col;
            const bool ok = n < P.N && oh < P.Ho && co < P.Co;
            const float v = P.gy[ok ? ((size_t)(n * P.Co + co) * HoWo + oh * P.Wo + cc) : 0];
            gs[col * P.PCp + p] = ok ? v : 0.f;
        }
        __syncthreads();
        // ---- MFMA: k-steps of 4 consecutive pixels along ow -----------------------------------------
        for (int p0 = 0; p0 < P.PC; p0 += 4) {
            const int img = fdiv(p0, P.d_TRWo);
            const int rem = p0 - img * (P.TR * P.Wo);
            const int r = fdiv(rem, P.d_Wo);
            const int c0 = rem - r * P.Wo;
            const int pixoff = img * (P.SR * P.SCp) + S * r * P.SCp + S * c0;
            const float b = gs[gb + p0];
            const float* xp = xs + xa + pixoff;
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int kw = 0; kw < KS; ++kw) {
                    const float a = xp[kh * P.SCp + kw];
                    acc[kh * KS + kw] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[kh * KS + kw], 0, 0, 0);
                }
        }
    }
    // ---- store: D col = lane&15 -> co (contiguous), row = 4*(lane>>4)+reg -> ci -------------------------
    float* outp = P.out + (size_t)split * P.out_elems;
    const int co = co0 + wj * 16 + l15;
    if (co < P.Co) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + wi * 16 + q * 4 + r;
                if (ci < P.Ci) outp[((size_t)t * P.Ci + ci) * P.Co + co] = acc[t][r];
            }
    }
}

}  // namespace

namespace ggan {

size_t conv_workspace_bytes(const ggan_conv_geom& g) {
    // upper bound over the three ops: 16 split-K slabs of the largest result
    size_t big = (size_t)g.N * g.Ci * g.H * g.W, small_ = (size_t)g.N * g.Co * g.Ho * g.Wo;
    size_t wsz = (size_t)g.k * g.k * g.Ci * g.Co;
    size_t m = big > small_ ? big : small_;
    size_t a = 16 * m * sizeof(float), b = 64 * wsz * sizeof(float);
    return a > b ? a : b;
}

static bool hot_geometry(const ggan_conv_geom& g) { return g.k == 5 && g.stride == 2; }

int conv_fwd_mfma(const ggan_conv_geom& g, const float* x, const float* w, const float* bias, float* y, int act,
                  float alpha, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!hot_geometry(g) || (g.Co & 3)) return 1;
    CorrParams P;
    memset(&P, 0, sizeof(P));
    P.in = x; P.w = w; P.bias = bias;
    P.N = g.N; P.CKtot = g.Ci; P.Hin = g.H; P.Win = g.W;
    P.CNtot = g.Co; P.Hout = g.Ho; P.Wout = g.Wo;
    P.ors = 1; P.ocs = 1;
    P.w_si = (long)g.k * g.Ci * g.Co; P.w_sj = (long)g.Ci * g.Co; P.w_sk = g.Co; P.w_sn = 1;
    pick_tile(g.Ho, g.Wo, g.N, P.TR, P.TC, P.TI);
    P.SR = 2 * (P.TR - 1) + 5;
    P.SCp = 2 * (P.TC - 1) + 5;
    while (CK * P.TI * P.SR * P.SCp > XE_MAX * 256 && P.TR > 1) {
        P.TR = (P.TR + 1) / 2;
        P.SR = 2 * (P.TR - 1) + 5;
    }
    P.CS = P.TI * P.SR * P.SCp;
    if (CK * P.CS > XE_MAX * 256) return 1;
    P.d_CS = make_fastdiv(P.CS); P.d_SRSC = make_fastdiv(P.SR * P.SCp); P.d_SCp = make_fastdiv(P.SCp);
    P.d_TRTC = make_fastdiv(P.TR * P.TC); P.d_TC = make_fastdiv(P.TC);
    P.img_groups = cdiv(g.N, P.TI);
    CorrClass& c = P.cls[0];
    c.Hu = g.Ho; c.Wv = g.Wo; c.ou = -g.pad_t; c.ov = -g.pad_l; c.or0 = 0; c.oc0 = 0; c.wbase = 0;
    c.tiles_r = cdiv(g.Ho, P.TR); c.tiles_c = cdiv(g.Wo, P.TC);
    const int gx = P.img_groups * c.tiles_r * c.tiles_c, gy_ = cdiv(g.Co, TN);
    P.SK = pick_splitk(gx * gy_, g.Ci, "GGAN_FWD_SK");
    P.cps = cdiv(cdiv(g.Ci, P.SK), CK) * CK;
    P.SK = cdiv(g.Ci, P.cps);
    P.act = act; P.alpha = alpha;
    P.out_elems = (size_t)g.N * g.Co * g.Ho * g.Wo;
    if (P.SK > 1 && (size_t)P.SK * P.out_elems * sizeof(float) > ws_bytes) {
        P.SK = 1; P.cps = cdiv(g.Ci, CK) * CK;
    }
    P.out = P.SK > 1 ? (float*)ws : y;
    const size_t shmem = (((CK * P.CS + 3) & ~3) + 25 * CK * TN) * sizeof(float);
    const double fl = 2.0 * P.out_elems * g.Ci * 25.0;
    GGAN_LAUNCH("conv_fwd_mfma", fl, 0, corr_kernel<0>, dim3(gx, gy_, P.SK), dim3(256), shmem, s, P);
    if (P.SK > 1) return launch_reduce((const float*)ws, P.SK, P.out_elems, y, bias, g.Co, g.Ho * g.Wo, act, alpha, s);
    return 0;
}

int conv_dgrad_mfma(const ggan_conv_geom& g, const float* gy, const float* w, const float* bias, float* gx, int act,
                    float alpha, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!hot_geometry(g) || (g.Co & 3) || g.Ci < 16) return 1;
    const int S = 2;
    CorrParams P;
    memset(&P, 0, sizeof(P));
    P.in = gy; P.w = w; P.bias = bias;
    P.N = g.N; P.CKtot = g.Co; P.Hin = g.Ho; P.Win = g.Wo;
    P.CNtot = g.Ci; P.Hout = g.H; P.Wout = g.W;
    P.ors = S; P.ocs = S;
    P.w_si = (long)S * g.k * g.Ci * g.Co; P.w_sj = (long)S * g.Ci * g.Co; P.w_sk = 1; P.w_sn = g.Co;
    int maxHu = 0, maxWv = 0;
    int offs[2][2], bases[2][2], cnt[2][2];   // [dim][parity]
    for (int d = 0; d < 2; ++d) {
        const int pad = d == 0 ? g.pad_t : g.pad_l, L = d == 0 ? g.H : g.W;
        for (int p = 0; p < S; ++p) {
            const int off = ((p - pad) % S + S) % S;
            offs[d][p] = off;
            bases[d][p] = (off + pad - p) / S;
            cnt[d][p] = off < L ? (L - off + S - 1) / S : 0;
        }
    }
    for (int p = 0; p < S; ++p) {
        if (cnt[0][p] > maxHu) maxHu = cnt[0][p];
        if (cnt[1][p] > maxWv) maxWv = cnt[1][p];
    }
    pick_tile(maxHu, maxWv, g.N, P.TR, P.TC, P.TI);
    P.SR = (P.TR - 1) + 3;
    P.SCp = (P.TC - 1) + 3;
    P.CS = P.TI * P.SR * P.SCp;
    if (CK * P.CS > XE_MAX * 256) return 1;
    P.d_CS = make_fastdiv(P.CS); P.d_SRSC = make_fastdiv(P.SR * P.SCp); P.d_SCp = make_fastdiv(P.SCp);
    P.d_TRTC = make_fastdiv(P.TR * P.TC); P.d_TC = make_fastdiv(P.TC);
    P.img_groups = cdiv(g.N, P.TI);
    int max_tiles = 0;
    for (int ph = 0; ph < S; ++ph)
        for (int pw = 0; pw < S; ++pw) {
            CorrClass& c = P.cls[ph * 2 + pw];
            c.Hu = cnt[0][ph]; c.Wv = cnt[1][pw];
            c.ou = bases[0][ph]; c.ov = bases[1][pw];
            c.or0 = offs[0][ph]; c.oc0 = offs[1][pw];
            c.wbase = ((long)ph * g.k + pw) * g.Ci * g.Co;
            c.tiles_r = cdiv(c.Hu > 0 ? c.Hu : 1, P.TR); c.tiles_c = cdiv(c.Wv > 0 ? c.Wv : 1, P.TC);
            if (c.tiles_r * c.tiles_c > max_tiles) max_tiles = c.tiles_r * c.tiles_c;
        }
    const int gxd = P.img_groups * max_tiles, gyd = cdiv(g.Ci, TN);
    P.SK = pick_splitk(gxd * gyd * 4, g.Co, "GGAN_DGRAD_SK");
    P.cps = cdiv(cdiv(g.Co, P.SK), CK) * CK;
    P.SK = cdiv(g.Co, P.cps);
    P.act = act; P.alpha = alpha;
    P.out_elems = (size_t)g.N * g.Ci * g.H * g.W;
    if (P.SK > 1 && (size_t)P.SK * P.out_elems * sizeof(float) > ws_bytes) {
        P.SK = 1; P.cps = cdiv(g.Co, CK) * CK;
    }
    P.out = P.SK > 1 ? (float*)ws : gx;
    const size_t shmem = (((CK * P.CS + 3) & ~3) + 9 * CK * TN) * sizeof(float);
    const double fl = 2.0 * g.N * g.Co * g.Ho * g.Wo * (double)g.Ci * 25.0;
    GGAN_LAUNCH("conv_dgrad_mfma", fl, 0, corr_kernel<1>, dim3(gxd, gyd, 4 * P.SK), dim3(256), shmem, s, P);
    if (P.SK > 1) return launch_reduce((const float*)ws, P.SK, P.out_elems, gx, bias, g.Ci, g.H * g.W, act, alpha, s);
    return 0;
}

int conv_wgrad_mfma(const ggan_conv_geom& g, const float* x, const float* gy, float* gw, void* ws, size_t ws_bytes,
                    hipStream_t s) {
    if (!hot_geometry(g) || (g.Wo & 3)) return 1;
    WgradParams P;
    memset(&P, 0, sizeof(P));
    P.x = x; P.gy = gy;
    P.N = g.N; P.Ci = g.Ci; P.H = g.H; P.W = g.W; P.Co = g.Co; P.Ho = g.Ho; P.Wo = g.Wo;
    P.pad_t = g.pad_t; P.pad_l = g.pad_l;
    if (g.Wo > 64) return 1;
    P.TR = 64 / g.Wo; if (P.TR > g.Ho) P.TR = g.Ho;
    P.TI = 64 / (P.TR * g.Wo); if (P.TI < 1) P.TI = 1; if (P.TI > g.N) P.TI = g.N;
    P.SR = 2 * (P.TR - 1) + 5;
    P.SCp = 2 * (g.Wo - 1) + 5;
    P.CS = P.TI * P.SR * P.SCp;
    if (!(P.CS & 1)) P.CS += 1;          // odd channel stride => the 16 ci lanes hit distinct banks
    P.PC = P.TI * P.TR * g.Wo;
    P.PCp = P.PC | 1;
    P.d_CS = make_fastdiv(P.CS); P.d_SRSC = make_fastdiv(P.SR * P.SCp); P.d_SCp = make_fastdiv(P.SCp);
    P.d_PC = make_fastdiv(P.PC); P.d_TRWo = make_fastdiv(P.TR * g.Wo); P.d_Wo = make_fastdiv(g.Wo);
    P.row_tiles = cdiv(g.Ho, P.TR);
    P.chunks_total = cdiv(g.N, P.TI) * P.row_tiles;
    const int gx = cdiv(g.Ci, WG_CI), gy_ = cdiv(g.Co, WG_CO);
    int sk = env_int("GGAN_WGRAD_SK", 0);
    if (sk <= 0) {
        sk = env_int("GGAN_TARGET_WGS", 512) / (gx * gy_);
        if (sk > P.chunks_total) sk = P.chunks_total;
        if (sk > 64) sk = 64;
    }
    if (sk < 1) sk = 1;
    P.out_elems = (size_t)25 * g.Ci * g.Co;
    while (sk > 1 && (size_t)sk * P.out_elems * sizeof(float) > ws_bytes) sk /= 2;
    P.chunks_per_split = cdiv(P.chunks_total, sk);
    P.SK = cdiv(P.chunks_total, P.chunks_per_split);
    P.out = P.SK > 1 ? (float*)ws : gw;
    const size_t shmem = ((size_t)WG_CI * P.CS + (size_t)WG_CO * P.PCp) * sizeof(float);
    if (shmem > 160 * 1024) return 1;
    const double fl = 2.0 * g.N * g.Co * g.Ho * g.Wo * (double)g.Ci * 25.0;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<5, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    GGAN_LAUNCH("conv_wgrad_mfma", fl, 0, (wgrad_kernel<5, 2>), dim3(gx, gy_, P.SK), dim3(256), shmem, s, P);
    if (P.SK > 1) return launch_reduce((const float*)ws, P.SK, P.out_elems, gw, nullptr, 1, 1, GGAN_ACT_NONE, 0.f, s);
    return 0;
}

}  // namespace ggan
