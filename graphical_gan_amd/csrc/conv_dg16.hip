// Data gradient of the 5x5 / stride-2 SAME convolutions (Conv2DBackpropInput = Deconv2D FORWARD, tflib/ops/deconv2d.py:101-107,
// and what tf.gradients derives from tflib/ops/conv2d.py:106-112) on 64-pixel x 16-channel tiles -- round 4.
//
// Why another kernel.  The stride-2 transposed conv splits into its four output-parity classes, each a dense stride-1
// correlation with a 3x3 / 3x2 / 2x3 / 2x2 sub-filter (conv_corr.hip).  A filter tap belongs to ONE class, so a workgroup's
// filter slice is reused only by the CLASS pixels of its tile: the 32-pixel x 32-channel tile of corr_kernel<2,1,1,8,1> stages
// 51 KB of filter per 16 reduction channels for 3200 cycles of MFMA issue -- more than a CU pulls in while the whole chip stages
// (11.6 B/cycle/CU measured), and that kernel's chunk took 4810 cycles.  Here a workgroup owns 64 class
// pixels x 16 output channels (x 4 classes): half the filter bytes per FLOP at the same workgroup count.  A 16-channel tile
// needs v_mfma_f32_16x16x4_f32 (the 32x32x2 form cannot go below 32 channels); its k = 4 is four consecutive reduction
// channels, which is exactly what a 16-byte lane of an LDS-DMA instruction fetches from the HWIO filter read in place (the
// reduction channel is the contiguous index there).
//
// Structure (8 waves, one workgroup per CU):
//   * wave = (k quad of the chunk, 16-channel tile, 32-pixel half): 25 taps x 2 pixel fragments = 50 MFMAs per chunk and wave,
//     1 filter fragment + 2 slab fragments (ds_read_b32 each, immediate offsets only: the slab row pitch is a template parameter)
//     per tap; the k quads are combined through LDS at the end (4-way at 16 channels, 2-way at 32).
//   * both operands by LDS-DMA, one chunk ahead into the other of two staging buffers, the instructions dealt out between the MFMA
//     pairs from the second pair on.  (First version: a ring of three buffers fetched TWO chunks ahead, so that the wait at a chunk's
//     end would retire loads a whole chunk old.  Measured slower everywhere -- 24.0 vs 22.8 us on 128->64 @8->16, the first barrier
//     at 5560 instead of 3740 cycles: twice the bytes in flight lengthen every request's queue, and the dealt-out loads land within
//     their chunk anyway.  GGAN_DG16_AHEAD=2 still selects it.)
//   * no per-launch descriptor arithmetic: the slab's per-lane byte offsets (halo lanes = out-of-range offset = zeros) depend on
//     the geometry and the tile position only; they are built ONCE per geometry on the host (plan cache below) and read with one
//     coalesced load per DMA instruction while the first filter blocks -- whose lane offsets are shifts of the lane id -- are
//     already in flight.
//   * MASKED variant (the critic's layers: gy * lrelu'(y) fused into the operand staging): the slab goes through registers with the
//     select on the way -- loaded two chunks ahead, committed at the start of the chunk in between into a third buffer -- the filter
//     as above.
#include "common.h"
#include "conv.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
using namespace ggan;

namespace {

constexpr int NTHR = 512, NWAVE = 8;
constexpr unsigned OOB = 0x7FFFFFF0u;
constexpr int XN_MAX = 3;            // slab DMA wave-instructions per wave and chunk (<= 1536 16-byte units per chunk)
constexpr int WBLK = 258;            // floats between filter blocks: the two 8-channel blocks of a fragment sit 2 banks apart
constexpr int PS = 68;               // epilogue: pixel pitch of a (k quad, class, channel) row
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Dg16Params {
    const float* in;       // gy [N, CK, Hin, Win]
    const float* w;        // HWIO filter [5][5][CN][CK] (CK contiguous)
    const float* bias;     // [CN] or NULL
    float* out;            // gx [N, CN, 2*Hin, 2*Win]
    const float* in_ref;   // MASKED: slab values are in[i] * act'(in_ref[i])
    const unsigned* xtab;  // [tile position][XN_MAX][NTHR] slab byte offsets relative to the tile's first image and the chunk's first channel
    int N, CK, Hin, Win, CN;
    int lTC, lTR;          // log2 of the tile's class-pixel columns / rows (TI = 64 >> (lTC + lTR) images)
    int CS;                // slab floats per reduction channel (rows padded to the pitch, padded for conflict-free fragment reads)
    int xinstr;            // slab wave-instructions per chunk (whole workgroup)
    int tiles_r, tiles_c;
    int nchunks;
    int ahead;             // chunks the DMA runs ahead: 1 (2: experiment)
    int nstage;            // staging buffers: 2, or 3 when something runs two chunks ahead (the MASKED slab's register loads, ahead == 2)
    unsigned in_bytes, w_bytes;
    int xcd_p;                    // XCD-aware tile order (conv_corr.hip, CorrParams::xcd_p): pixel-tile groups among the 8 XCDs; 0 = plain order
    int act;
    float alpha;
    float in_slope;        // MASKED: slope of the negative side (lrelu alpha, relu 0)
    int dbg;
    unsigned long long* stamps;
    int tapoff[28];        // byte offset of tap t (class-major order) inside the filter; unused slots: -1
};

// class-major tap order: class c = (ph, pw) = (c >> 1, c & 1) holds the taps kh = ph + 2 i, kw = pw + 2 j
__host__ __device__ constexpr int tap_cls(int t) { return t < 9 ? 0 : t < 15 ? 1 : t < 21 ? 2 : 3; }
__host__ __device__ constexpr int tap_i(int t) { return t < 9 ? t / 3 : t < 15 ? (t - 9) / 2 : t < 21 ? (t - 15) / 3 : (t - 21) / 2; }
__host__ __device__ constexpr int tap_j(int t) { return t < 9 ? t % 3 : t < 15 ? (t - 9) % 2 : t < 21 ? (t - 15) % 3 : (t - 21) % 2; }
// SAME padding of an even-sized input is (1, 2): output row ih = off_p + 2 a of parity class p (off_0 = 1, off_1 = 0) reads gy row
// a + base_p - i (base_0 = 1, base_1 = 0).  Slab row r <-> gy row u0 - 1 + r, slab column c <-> gy column v0 - 4 + c (16-byte units).
__host__ __device__ constexpr int tap_slab_off(int t, int scp) {
    const int c = tap_cls(t), ph = c >> 1, pw = c & 1;
    return (1 + (1 - ph) - tap_i(t)) * scp + 4 + (1 - pw) - tap_j(t);
}

__device__ __forceinline__ void wait_vm(int n) {      // s_waitcnt vmcnt(n) for a wave-uniform run-time n
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// SCP: slab row pitch in floats (class-pixel columns of the tile + 8).  KQ: k quads per chunk (4: 16-channel tile and 16-channel chunks,
// 2: 32-channel tile and 8-channel chunks).
// (the body lives in a __device__ function: with the LDS-DMA builtin directly inside the __global__ template, hipcc 7.2 compiled the
//  translation unit without a diagnostic but never emitted the host-side launch stub)
template <int SCP, int KQ, bool MASKED>
__device__ __forceinline__ void dg16_body(const Dg16Params& P, float* smem) {
    constexpr int NCHT = 4 / KQ;               // 16-channel tiles of the workgroup
    constexpr int CHT = 16 * NCHT;
    constexpr int CKC = 4 * KQ;                // reduction channels per chunk
    constexpr int TPB = 8 / KQ;                // taps per filter block (a block = 8 output channels x TPB taps x KQ k quads = 64 lanes x 16 B)
    constexpr int NOCT = 2 * NCHT;             // 8-channel groups
    constexpr int NTG = (25 + TPB - 1) / TPB;
    constexpr int NBLK = NTG * NOCT;
    constexpr int WQ = (NBLK + NWAVE - 1) / NWAVE;
    constexpr int WREG = (NBLK * WBLK + 3) & ~3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kqw = wave % KQ, chh = (wave / KQ) % NCHT, pxh = wave / (KQ * NCHT);
    const int wg_lin = blockIdx.y * gridDim.x + blockIdx.x;
    const bool stamping = (P.dbg & 4) && P.stamps && tid == 0;
    auto stamp = [&](int i) { if (stamping) P.stamps[(size_t)wg_lin * 16 + i] = __builtin_readcyclecounter(); };
    stamp(0);

    // ---- which tile ---------------------------------------------------------------------------------------------------------
    const int lTC = P.lTC, lTR = P.lTR, TC = 1 << lTC, TR = 1 << lTR, TI = 64 >> (lTC + lTR);
    const int tpi = P.tiles_r * P.tiles_c;
    int bx = blockIdx.x, by = blockIdx.y;
    if (P.xcd_p > 0) {
        const int xcd = wg_lin & 7, slot = wg_lin >> 3;
        const int npx = gridDim.x / P.xcd_p, nch = gridDim.y / (8 / P.xcd_p);
        bx = (xcd % P.xcd_p) * npx + slot % npx;
        by = (xcd / P.xcd_p) * nch + slot / npx;
    }
    const int ig = bx / tpi, tpos = bx - ig * tpi;
    const int tr = tpos / P.tiles_c, tc = tpos - tr * P.tiles_c;
    const int n0 = ig * TI, u0 = tr * TR, v0 = tc * TC;
    const int cn0 = by * CHT;
    const int HWin = P.Hin * P.Win;
    const int XREG = P.xinstr * 256;                      // slab region of a stage: whole wave-instructions (64 lanes x 4 floats)
    const int STAGE = WREG + XREG;

    const auto rin = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, (short)0, (int)P.in_bytes, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, (short)0, (int)P.w_bytes, 0x00020000);
    const auto rref = __builtin_amdgcn_make_buffer_rsrc((void*)(MASKED ? P.in_ref : P.in), (short)0, (int)P.in_bytes, 0x00020000);

    // ---- slab descriptors: one coalesced table read per DMA instruction of this wave (issued first: the filter blocks below need none) ----
    unsigned xvo[XN_MAX];
    int nx = 0;                                            // slab instructions of this wave
    {
        const unsigned* tab = P.xtab + (size_t)tpos * (XN_MAX * NTHR) + tid;
#pragma unroll
        for (int j = 0; j < XN_MAX; ++j) {
            xvo[j] = OOB;
            if (wave + j * NWAVE < P.xinstr) { xvo[j] = tab[j * NTHR]; nx = j + 1; }
        }
    }
    // ---- filter blocks of this wave: block b = (tap group, 8-channel group); lane = (tap of the group, k quad, channel) ----------------
    const int l_t = lane / (KQ * 8), l_kq = (lane >> 3) % KQ, l_c8 = lane & 7;
    unsigned wvo[WQ];
    int wlds[WQ];
    int nw = 0;
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
        const int b = wave + q * NWAVE;
        wvo[q] = OOB; wlds[q] = 0;
        if (b < NBLK) {
            const int tg = b / NOCT, oct = b - tg * NOCT;
            // (the taps of the group through wave-uniform scalar loads and selects: a per-lane index into the argument block would be a
            //  dependent vector load)
            int toff = P.tapoff[tg * TPB];
#pragma unroll
            for (int i = 1; i < TPB; ++i) { const int ti = P.tapoff[tg * TPB + i]; toff = l_t == i ? ti : toff; }   // (-1: no such tap -> zeros)
            wvo[q] = toff >= 0 ? (unsigned)toff + (unsigned)(((oct * 8 + l_c8) * P.CK + l_kq * 4) * 4) : OOB;
            wlds[q] = (tg * NOCT + oct) * WBLK;
            nw = q + 1;
        }
    }
    const int w_s0 = cn0 * P.CK * 4;                       // scalar part of the filter offset: first output channel of the tile
    auto dma_w = [&](int q, int chunk, int stage) {
        float* dst = smem + stage * STAGE + wlds[q];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)dst, 16, wvo[q], w_s0 + chunk * CKC * 4, 0, 0);
    };
    // chunks 0 and 1 of the filter go out before anything else is known
    const int nch = P.nchunks;
    const int AH = P.ahead, NST = P.nstage;
#pragma unroll
    for (int q = 0; q < WQ; ++q)
        if (q < nw) dma_w(q, 0, 0);
    if (nch > 1 && AH == 2) {
#pragma unroll
        for (int q = 0; q < WQ; ++q)
            if (q < nw) dma_w(q, 1, 1);
    }
    stamp(1);

    // the image part of a slab offset goes into the per-lane offset (so that images beyond N fall out of range), the chunk part is scalar
    const unsigned img_base = (unsigned)n0 * (unsigned)P.CK * (unsigned)HWin * 4u;
#pragma unroll
    for (int j = 0; j < XN_MAX; ++j) {
        const unsigned t = xvo[j];
        xvo[j] = t >= OOB ? OOB : t + img_base;
    }
    auto dma_x = [&](int j, int chunk, int stage) {
        float* dst = smem + stage * STAGE + WREG + (wave + j * NWAVE) * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)dst, 16, xvo[j], chunk * CKC * HWin * 4, 0, 0);
    };
    u32x4 xr[MASKED ? XN_MAX : 1], xf[MASKED ? XN_MAX : 1];
    auto load_x = [&](int j, int chunk) {                 // MASKED: slab unit + its mask reference into registers
        xr[j] = __builtin_amdgcn_raw_buffer_load_b128(rin, xvo[j], chunk * CKC * HWin * 4, 0);
        xf[j] = __builtin_amdgcn_raw_buffer_load_b128(rref, xvo[j], chunk * CKC * HWin * 4, 0);
    };
    auto commit_x = [&](int stage) {
#pragma unroll
        for (int j = 0; j < XN_MAX; ++j)
            if (j < nx) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float a = __uint_as_float(xr[j][c]);
                    v[c] = __uint_as_float(xf[j][c]) > 0.f ? a : a * P.in_slope;
                }
                *reinterpret_cast<f32x4*>(smem + stage * STAGE + WREG + (wave + j * NWAVE) * 256 + lane * 4) = v;
            }
    };

    // ---- per-lane fragment bases ------------------------------------------------------------------------------------------------
    const int l15 = lane & 15, kk = lane >> 4;
    // filter fragment: channel ch = l15 of this wave's 16-channel tile, k = kk of this wave's quad
    const int afrag = (chh * 2 + (l15 >> 3)) * WBLK + (kqw * 8 + (l15 & 7)) * 4 + kk;
    // slab fragments: pixel p = pxh*32 + f*16 + l15 -> (image, row a, column b)
    int bfrag[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const int p = pxh * 32 + f * 16 + l15;
        const int b = p & (TC - 1), a = (p >> lTC) & (TR - 1), img = p >> (lTC + lTR);
        bfrag[f] = WREG + (kqw * 4 + kk) * P.CS + (img * (TR + 2) + a) * SCP + b;
    }

    f32x4 acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[c][f] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- first two chunks of the slab ----------------------------------------------------------------------------------------------
    if constexpr (MASKED) {
#pragma unroll
        for (int j = 0; j < XN_MAX; ++j)
            if (j < nx) load_x(j, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        commit_x(0);
        if (nch > 1) {       // (MASKED keeps the two-ahead timing)
#pragma unroll
            for (int j = 0; j < XN_MAX; ++j)
                if (j < nx) load_x(j, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            commit_x(1);
        }
    } else {
#pragma unroll
        for (int j = 0; j < XN_MAX; ++j)
            if (j < nx) dma_x(j, 0, 0);
        if (nch > 1 && AH == 2) {
#pragma unroll
            for (int j = 0; j < XN_MAX; ++j)
                if (j < nx) dma_x(j, 1, 1);
            wait_vm(nx);                                  // chunk 0 complete (filter 0, filter 1, slab 0 were issued before slab 1)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    stamp(3);

    // ---- main loop: chunk c out of stage c % 3; the loads of chunk c + 2 are dealt out between its MFMA pairs ---------------------------
    // per wave and chunk: items 0 .. nxi-1 = slab (MASKED: register loads, committed at the start of the NEXT chunk), then nw filter blocks
    const int nxi = nx;
    int stage = 0;
    for (int c = 0; c < nch; ++c) {
        const int s1 = stage + 1 == NST ? 0 : stage + 1;  // the next chunk's buffer
        const int s2 = AH == 2 ? (stage >= 1 ? stage - 1 : 2) : s1;         // the buffer of chunk c + AH (two ahead: NST == 3)
        const bool more = c + AH < nch;                    // DMA (filter blocks; the slab when it goes by DMA): chunk c + AH
        const bool more_x = MASKED ? c + 2 < nch : more;   // MASKED: the slab's register loads run two chunks ahead (committed in chunk c + 1)
        if constexpr (MASKED) {
            // the slab of chunk c + 1 sits in registers since chunk c - 1 (behind it in the queue: the filter blocks of chunk c + 1)
            if (c >= 1 && c + 1 < nch) {
                if (AH == 2) wait_vm(nw);                  // (AH == 1: the end of the last chunk waited for everything)
                commit_x(s1);
            }
        }
        const float* sb = smem + stage * STAGE;
        float a[25], b0[25], b1[25];
        auto load = [&](int t) {
            a[t] = sb[afrag + (t / TPB) * NOCT * WBLK + (t % TPB) * KQ * 32];
            b0[t] = sb[bfrag[0] + tap_slab_off(t, SCP)];
            b1[t] = sb[bfrag[1] + tap_slab_off(t, SCP)];
        };
        auto item = [&](int it) {                          // staging instruction number `it` of this wave
            if (it < XN_MAX) {
                if (more_x && it < nxi) {
                    if constexpr (MASKED) load_x(it, c + 2); else dma_x(it, c + AH, s2);
                }
            } else if (it - XN_MAX < WQ) {
                if (more && it - XN_MAX < nw) dma_w(it - XN_MAX, c + AH, s2);
            }
        };
        load(0);
        load(1);
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            if (t + 2 < 25) load(t + 2);
            acc[tap_cls(t)][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b0[t], acc[tap_cls(t)][0], 0, 0, 0);
            acc[tap_cls(t)][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b1[t], acc[tap_cls(t)][1], 0, 0, 0);
            // one staging instruction behind every third pair, from the second pair on (MASKED: two loads per slab item)
            if (t >= 1 && (t - 1) % 3 == 0) item((t - 1) / 3);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // the fragment reads of pair t + 2
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // this pair's MFMAs
            __builtin_amdgcn_sched_group_barrier(0x030, 2, 0);   // the staging instruction dealt to this pair
        }
        // retire the loads of chunk c + 1 (issued during chunk c - 1); those of chunk c + 2 stay in flight
        if (more && AH == 2) wait_vm(MASKED ? 2 * nxi + nw : nxi + nw);     // (two ahead: the loads issued in this chunk stay in flight)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c < 8) stamp(4 + c);
        stage = s1;
    }
    stamp(12);

    // ---- epilogue: the k quads are summed and the four classes interleaved into whole output rows through LDS ---------------------------
    float* red = smem;                                     // [kq][class][channel][PS]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[((kqw * 4 + c) * CHT + chh * 16 + kk * 4 + r) * PS + pxh * 32 + f * 16 + l15] = acc[c][f][r];
    __syncthreads();
    stamp(13);
    const int Hout = 2 * P.Hin, Wout = 2 * P.Win;
    const int lQ = lTC - 1;                                // float4 units per output row of the tile: TC / 2
    for (int u = tid; u < 64 * CHT; u += NTHR) {
        const int cq = u & ((1 << lQ) - 1);
        const int row2 = (u >> lQ) & (2 * TR - 1);
        const int ch = (u >> (lQ + lTR + 1)) & (CHT - 1);
        const int img = u >> (lQ + lTR + 1 + (NCHT == 1 ? 4 : 5));
        const int ph = 1 - (row2 & 1), arow = row2 >> 1;
        const int px = (img << (lTC + lTR)) + (arow << lTC) + 2 * cq;
        float2 e = make_float2(0.f, 0.f), o = make_float2(0.f, 0.f);
        float2 te[KQ], to[KQ];
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            te[k] = *reinterpret_cast<const float2*>(red + ((k * 4 + ph * 2 + 1) * CHT + ch) * PS + px);   // pw = 1: even columns
            to[k] = *reinterpret_cast<const float2*>(red + ((k * 4 + ph * 2 + 0) * CHT + ch) * PS + px);   // pw = 0: odd columns
        }
#pragma unroll
        for (int k = 0; k < KQ; ++k) { e.x += te[k].x; e.y += te[k].y; o.x += to[k].x; o.y += to[k].y; }
        const float bv = P.bias ? P.bias[cn0 + ch] : 0.f;
        float4 v = make_float4(e.x + bv, o.x + bv, e.y + bv, o.y + bv);
        v.x = act_apply(v.x, P.act, P.alpha); v.y = act_apply(v.y, P.act, P.alpha);
        v.z = act_apply(v.z, P.act, P.alpha); v.w = act_apply(v.w, P.act, P.alpha);
        if (n0 + img < P.N && !(P.dbg & 8))
            *reinterpret_cast<float4*>(P.out + (((size_t)(n0 + img) * P.CN + cn0 + ch) * Hout + 2 * u0 + row2) * Wout + 2 * v0 + 4 * cq) = v;
    }
    stamp(14);
}

template <int SCP, int KQ, bool MASKED>
__global__ __launch_bounds__(NTHR) void dg16_kernel(const Dg16Params P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    warm_kernarg(P);
    dg16_body<SCP, KQ, MASKED>(P, smem);
}

// ---- host: plan cache ---------------------------------------------------------------------------------------------------------------
struct PlanKey {
    int CK, Hin, Win, lTC, lTR, CS, ckc;
    bool operator<(const PlanKey& o) const {
        return std::tie(CK, Hin, Win, lTC, lTR, CS, ckc) < std::tie(o.CK, o.Hin, o.Win, o.lTC, o.lTR, o.CS, o.ckc);
    }
};
std::mutex g_plan_mu;
std::map<std::pair<int, PlanKey>, unsigned*> g_plans;      // (device, key) -> device table

int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// slab floats per reduction channel: rows padded to the pitch, then padded until the 16-pixel x 2-channel half-wave fragment reads of
// every tap are bank-conflict free (ds_read_b32: lanes 0-31 and 32-63 are served separately, 32 banks)
int plan_cs(int TI, int TR, int TC, int SCP) {
    const int base = TI * (TR + 2) * SCP;
    const int lTC = ilog2(TC), lTR = ilog2(TR);
    for (int pad = 0; pad < 64; pad += 4) {
        const int CS = base + pad;
        bool ok = true;
        {
            int seen[32] = {0};
            for (int l = 0; l < 32 && ok; ++l) {
                const int p = l & 15, kk = l >> 4;
                const int b = p & (TC - 1), a = (p >> lTC) & (TR - 1), img = p >> (lTC + lTR);
                const int addr = kk * CS + (img * (TR + 2) + a) * SCP + b;
                if (seen[addr & 31]++) ok = false;
            }
        }
        if (ok) return CS;
    }
    return base;
}

const unsigned* get_plan(const PlanKey& k, int SCP, int xinstr, int tiles_r, int tiles_c, hipStream_t s) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plans.find({dev, k});
    if (it != g_plans.end()) return it->second;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    const int TC = 1 << k.lTC, TR = 1 << k.lTR, TI = 64 >> (k.lTC + k.lTR), SR = TR + 2, upr = SCP / 4;
    const int npos = tiles_r * tiles_c;
    std::vector<unsigned> tab((size_t)npos * XN_MAX * NTHR, OOB);
    for (int tr = 0; tr < tiles_r; ++tr)
        for (int tc = 0; tc < tiles_c; ++tc)
            for (int j = 0; j < XN_MAX; ++j)
                for (int tid = 0; tid < NTHR; ++tid) {
                    const int wave = tid >> 6, lane = tid & 63;
                    const int instr = wave + j * NWAVE;
                    if (instr >= xinstr) continue;
                    const int e = instr * 64 + lane;           // 16-byte unit of the chunk's slab
                    const int kch = e / (k.CS / 4), rem = e - kch * (k.CS / 4);
                    if (kch >= k.ckc || rem >= TI * SR * upr) continue;
                    const int img = rem / (SR * upr), r = (rem / upr) % SR, u = rem % upr;
                    const int ih = tr * TR - 1 + r, iw = tc * TC - 4 + 4 * u;
                    if (ih < 0 || ih >= k.Hin || iw < 0 || iw + 3 >= k.Win) continue;
                    tab[((size_t)(tr * tiles_c + tc) * XN_MAX + j) * NTHR + tid] =
                        (unsigned)((((size_t)img * k.CK + kch) * k.Hin + ih) * k.Win + iw) * 4u;
                }
    unsigned* d = nullptr;
    if (hipMalloc((void**)&d, tab.size() * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemcpy(d, tab.data(), tab.size() * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d); return nullptr; }
    g_plans[{dev, k}] = d;
    return d;
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

int launch_dg16(int scp, int kq, const Dg16Params& P, bool masked, dim3 grid, size_t shmem, hipStream_t s, double fl, double ab) {
    static std::atomic<unsigned long long> once{0};
#define DG16_EACH(X) X(12, 4) X(16, 4) X(24, 4) X(12, 2) X(16, 2) X(24, 2)
    if (first_on_device(once)) {
#define DG16_ATTR(SCP, KQ)                                                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dg16_kernel<SCP, KQ, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dg16_kernel<SCP, KQ, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        DG16_EACH(DG16_ATTR)
#undef DG16_ATTR
    }
    // (names as rocprofv3 prints the instantiations)
#define DG16_CASE(SCP, KQ)                                                                                                                        \
    if (scp == SCP && kq == KQ) {                                                                                                                 \
        if (masked) { GGAN_LAUNCH("dg16_kernel<" #SCP ", " #KQ ", true>", fl, ab, (dg16_kernel<SCP, KQ, true>), grid, dim3(NTHR), shmem, s, P); } \
        else { GGAN_LAUNCH("dg16_kernel<" #SCP ", " #KQ ", false>", fl, ab, (dg16_kernel<SCP, KQ, false>), grid, dim3(NTHR), shmem, s, P); }      \
        return 0;                                                                                                                                 \
    }
    DG16_EACH(DG16_CASE)
#undef DG16_CASE
#undef DG16_EACH
    return 1;
}

}  // namespace

namespace ggan {

// Returns 1 when the geometry (or the launch plan) is not covered: the caller goes on to conv_dgrad_mfma.
int conv_dgrad_dg16(const ggan_conv_geom& g, const float* gy, GyMask m, const float* w, const float* bias, float* gx, int act,
                    float alpha, int target_wgs, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!env_int("GGAN_DG16", 1)) return 1;
    if (g.k != 5 || g.stride != 2 || g.pad_t != 1 || g.pad_l != 1) return 1;
    if ((g.H & 1) || (g.W & 1) || g.Ho * 2 != g.H || g.Wo * 2 != g.W) return 1;
    const int CK = g.Co, CN = g.Ci, Hin = g.Ho, Win = g.Wo;
    if ((CK & 15) || (CN & 15) || CK < 16) return 1;
    if (Win < 4 || (Win & (Win - 1)) || (Hin & (Hin - 1))) return 1;
    if ((((uintptr_t)gy) | ((uintptr_t)w) | ((uintptr_t)gx) | ((uintptr_t)m.ref)) & 15) return 1;
    if (m.act != GGAN_ACT_NONE && m.act != GGAN_ACT_LRELU && m.act != GGAN_ACT_RELU) return 1;
    const size_t in_bytes = (size_t)g.N * CK * Hin * Win * 4, w_bytes = (size_t)25 * CN * CK * 4;
    if (in_bytes >= 0x7FFFFFF0ull || w_bytes >= 0x7FFFFFF0ull || (size_t)g.N * CN * g.H * g.W * 4 >= 0x7FFFFFF0ull) return 1;
    const int TC = Win < 16 ? Win : 16;
    int TR = 64 / TC; if (TR > Hin) TR = Hin;
    const int TI = 64 / (TR * TC);
    if (TI * TR * TC != 64) return 1;
    const int SCP = TC + 8;
    const int tiles_r = Hin / TR, tiles_c = Win / TC, igroups = cdiv(g.N, TI);
    const int ptiles = igroups * tiles_r * tiles_c;
    if (target_wgs <= 0) target_wgs = env_int("GGAN_TARGET_WGS", 200);
    // 32-channel tiles while they still give the planned number of workgroups, else 16-channel tiles; a grid far below the plan keeps
    // the older kernels (cross-workgroup split-K)
    int kq = env_int("GGAN_DG16_KQ", 0);
    if (kq != 2 && kq != 4) kq = ((CN & 31) == 0 && ptiles * (CN / 32) >= target_wgs) ? 2 : 4;
    if (kq == 2 && (CN & 31)) return 1;
    const int wgs = ptiles * (CN / (kq == 2 ? 32 : 16));
    if (wgs * 4 < target_wgs * env_int("GGAN_DG16_MINQ", 3) && !env_int("GGAN_DG16_FORCE", 0)) return 1;     // (under 3/4 of the plan)
    const int ckc = 4 * kq;
    if (CK % ckc) return 1;

    Dg16Params P;
    memset(&P, 0, sizeof(P));
    P.in = gy; P.w = w; P.bias = bias; P.out = gx;
    const bool masked = m.act != GGAN_ACT_NONE;
    if (masked) { P.in_ref = m.ref; P.in_slope = m.act == GGAN_ACT_LRELU ? m.alpha : 0.f; }
    P.N = g.N; P.CK = CK; P.Hin = Hin; P.Win = Win; P.CN = CN;
    P.lTC = ilog2(TC); P.lTR = ilog2(TR);
    P.CS = plan_cs(TI, TR, TC, SCP);
    P.xinstr = cdiv(ckc * P.CS / 4, 64);
    if (P.xinstr > XN_MAX * NWAVE) return 1;
    P.tiles_r = tiles_r; P.tiles_c = tiles_c;
    P.nchunks = CK / ckc;
    // chunks the DMA runs ahead.  ONE: measured faster than two on every layer (128->64 @8->16, 64 images: 22.8 vs 24.0 us, first barrier
    // 3740 vs 5560 cycles; 64->32 @16->32: 25.1 vs 28.0 us) -- as for the forward kinds (conv_corr.hip, GGAN_CORR_NSTG), more bytes in
    // flight only lengthen every request's queue; the loads dealt out from the second MFMA pair on land within their chunk.
    P.ahead = env_int("GGAN_DG16_AHEAD", 1) == 2 ? 2 : 1;
    P.nstage = (masked || P.ahead == 2) ? 3 : 2;
    P.in_bytes = (unsigned)in_bytes; P.w_bytes = (unsigned)w_bytes;
    P.act = act; P.alpha = alpha;
    P.dbg = env_int("GGAN_DBG", 0);
    for (int t = 0; t < 28; ++t) {
        P.tapoff[t] = -1;
        if (t < 25) {
            const int c = tap_cls(t), kh = (c >> 1) + 2 * tap_i(t), kw = (c & 1) + 2 * tap_j(t);
            P.tapoff[t] = (kh * 5 + kw) * CN * CK * 4;
        }
    }
    const PlanKey key{CK, Hin, Win, P.lTC, P.lTR, P.CS, ckc};
    P.xtab = get_plan(key, SCP, P.xinstr, tiles_r, tiles_c, s);
    if (!P.xtab) return 1;
    const int nblk = (kq == 4 ? 13 * 2 : 7 * 4);
    const int wreg = (nblk * WBLK + 3) & ~3;
    const size_t stage = (size_t)wreg + (size_t)P.xinstr * 256;
    const size_t red = (size_t)kq * 4 * (kq == 4 ? 16 : 32) * PS;
    const size_t shmem = (P.nstage * stage > red ? P.nstage * stage : red) * sizeof(float);
    if (shmem > 160 * 1024) return 1;
    ws = ws_scratch(ws, ws_bytes);
    if ((P.dbg & 4) && ws && ws_bytes > (64u << 20)) P.stamps = (unsigned long long*)((char*)ws + ws_bytes - (32u << 20));
    const dim3 grid(ptiles, CN / (kq == 2 ? 32 : 16));
    {   // XCD-aware tile order: what the eight L2s fetch together is 8 * input / p + filter * p for p pixel-tile groups
        const int gx = (int)grid.x, gy = (int)grid.y, force = env_int("GGAN_CORR_XCD", -1);
        P.xcd_p = 0;
        if (force != 0 && (gx * gy) % 8 == 0) {
            double best = (gx % 8 == 0) ? (double)in_bytes + 8.0 * w_bytes : 8.0 * ((double)in_bytes + w_bytes);
            for (int p = 1; p <= 8; p *= 2) {
                if (gx % p || gy % (8 / p) || (force > 0 && p != force)) continue;
                const double cost = 8.0 * in_bytes / p + (double)w_bytes * p;
                if (cost < 0.9 * best || force > 0) { best = cost; P.xcd_p = p; }
            }
        }
    }
    const double fl = 2.0 * g.N * g.Co * g.Ho * g.Wo * (double)g.Ci * 25.0;
    const double ab = (double)in_bytes + (double)w_bytes + 4.0 * (double)g.N * CN * g.H * g.W;
    return launch_dg16(SCP, kq, P, masked, grid, shmem, s, fl, ab);
}

}  // namespace ggan
