// fp32-MFMA filter-gradient kernel for gfx950 (CDNA4): Conv2DBackpropFilter of the 5x5 stride-2 layers (and, with the
// roles of the two tensors swapped, of Deconv2D).
//
//   gw[kh,kw,ci,co] = sum_{n,oh,ow} x[n,ci,2*oh+kh-pt,2*ow+kw-pl] * gy[n,co,oh,ow]
//
// GEMM per tap: M = ci, N = co, K = pixels.  v_mfma_f32_16x16x4_f32 so that ONE wave holds all 25 taps of a 16x16 (ci,co)
// tile in 100 accumulator registers; a k-step is 4 consecutive pixels along ow: 1 gy-fragment read + 25 x-fragment reads
// (lane base + immediate tap offset into the zero-haloed x slab) + 25 MFMAs.  The 4 waves of a workgroup take different
// pixel quads of the staged chunk (in-workgroup split-K) and combine through LDS at the end, so only a light global
// split-K over image groups is needed to reach one workgroup per CU (deterministic partial slabs, no atomics).
//
// LDS x slab: [16 ci][TI images][SR rows][even cols | odd cols]; de-interleaving the columns by parity turns the stride-2
// pixel walk into stride-1 LDS addresses, and a channel stride == 2 (mod 32) makes the 16ci x 2px half-wave fragment read
// conflict-free.  Staging = 16-byte raw buffer loads of whole image rows (hardware bounds check supplies the zero rows),
// prefetched into registers under the MFMA block of the previous chunk; halo columns are zeroed once.
#include "common.h"
#include "conv.h"
#include <stdlib.h>
#include <string.h>
using namespace ggan;

namespace {

// Compile-time ablations for timing experiments (tools/variant_lib.sh builds a second libggan.so with -DGGAN_ABL=bits and
// tools/stamps.py prints the per-workgroup phase times): 1 = no staging inside the chunk loop, 4 = MFMAs on register constants
// instead of LDS fragments.  0 in the product build: the branches fold away.
#ifndef GGAN_ABL
#define GGAN_ABL 0
#endif
// conv_wgrad_split.hip includes this file with GGAN_WGRAD_SPLIT_TU set and is compiled with -mllvm -amdgpu-mfma-vgpr-form: the
// role-split kernel's 256-register waves keep all 200 accumulators in architectural VGPRs (with AGPRs in play the allocator splits
// the budget 128 / 128 and spills hundreds of registers), so the epilogue's parked quads are "v" operands there.
#ifndef GGAN_WGRAD_SPLIT_TU
#define GGAN_WGRAD_SPLIT_TU 0
#endif
#ifndef GGAN_SPLIT_SCHED
#define GGAN_SPLIT_SCHED 1
#endif
#if GGAN_WGRAD_SPLIT_TU
#define GGAN_ACC_REG(q) "v"(q)
#else
#define GGAN_ACC_REG(q) "a"(q)
#endif
constexpr int TCI = 16, TCO = 16;
constexpr int NW = 8;                      // waves per workgroup: two per SIMD, pixel quads dealt round-robin
constexpr int NTHR = 64 * NW;
constexpr int XU_MAX = 6;                  // float4 slab units per thread per chunk (3072 per workgroup)
constexpr unsigned OOB = 0x7FFFFFF0u;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct WgradParams {
    const float* x;
    const float* gy;
    float* out;          // gw, or the partial slabs when SK > 1
    int N, Ci, H, W, Co, Ho, Wo;
    int pad_t;
    int TR, TI;          // pixel chunk = TI images x TR rows x Wo cols (<= 64 pixels, multiple of 4)
    int SR, SCp, SCh, CS;
    int PC, PCp;
    FastDiv d_F4, d_SR, d_TI, d_PC4, d_TRWo, d_Wo;
    int row_tiles, chunks_total, chunks_per_split, SK;
    int xunits;
    unsigned x_bytes, gy_bytes;
    size_t out_elems;      // 25*Ci*Co
    size_t slab_stride;    // out_elems (+ Co when the bias gradient rides along)
    const float* gy_ref;   // optional fused activation backward: gy[i] * act'(gy_ref[i])
    int gy_act;
    float gy_alpha;
    float* gbias;          // optional: sum over n,oh,ow of the (masked) gy
    int dbg_nostore;
    unsigned long long* stamps;   // debug: per-workgroup s_memtime stamps (GGAN_DBG & 4)
};

// NIT: pairs of pixel quads per wave and chunk when that is a compile-time number (1: 64-pixel chunks, 2: 128), 0: any chunk size.
// With NIT > 0 the quad loop unrolls into 2*NIT straight-line MFMA blocks and the global loads of the chunk after next are dealt
// out between their MFMAs (as in the correlation kernels: eight waves issuing 7 KB each at once queue up at the CU's 64 B/clk
// vector-memory path while the matrix pipes wait).
template <int NIT>
__global__ __launch_bounds__(NTHR) void wgrad_kernel(const WgradParams P) {
    warm_kernarg(P);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = 5, NT = 25;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, qq = lane >> 4;
    const int ci0 = blockIdx.x * TCI, co0 = blockIdx.y * TCO, split = blockIdx.z;

    const int wg_lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const bool stamping = P.stamps != nullptr && tid == 0;
    auto stamp = [&](int i) { if (stamping) P.stamps[(size_t)wg_lin * 16 + i] = __builtin_readcyclecounter(); };
    stamp(0);
    // two staging buffers, each [TCI][CS] x slab + [TCO][PCp] gy tile: chunk c is multiplied out of buffer c&1 while chunk c+1 is
    // committed to the other one (ONE barrier per chunk; with a single buffer the commit sat between two barriers with every
    // matrix pipe idle: 5700 cycles per chunk where the MFMAs need 3200)
    const int STG = TCI * P.CS + TCO * P.PCp;
    const int HW = P.H * P.W, HoWo = P.Ho * P.Wo, F4 = P.W >> 2;

    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)P.x, (short)0, (int)P.x_bytes, 0x00020000);
    const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)P.gy, (short)0, (int)P.gy_bytes, 0x00020000);
    const bool masked = P.gy_ref != nullptr;
    const float mslope = P.gy_act == GGAN_ACT_LRELU ? P.gy_alpha : 0.f;
    const auto rr = __builtin_amdgcn_make_buffer_rsrc((void*)(masked ? P.gy_ref : P.gy), (short)0, (int)P.gy_bytes, 0x00020000);
    const bool do_bias = P.gbias != nullptr && blockIdx.x == 0;
    float bsum = 0.f;

    // ---- staging descriptors -----------------------------------------------------------------------------
    int xrel[XU_MAX], xinfo[XU_MAX], xlds[XU_MAX];
#pragma unroll
    for (int j = 0; j < XU_MAX; ++j) {
        const int u = tid + j * NTHR;
        int rel = 0, info = -1, l = 0;
        if (u < P.xunits) {
            const int row = fdiv(u, P.d_F4);
            const int f4 = u - row * F4;
            const int t = fdiv(row, P.d_SR);
            const int r = row - t * P.SR;
            const int cil = fdiv(t, P.d_TI);
            const int img = t - cil * P.TI;
            rel = (img * P.Ci + cil) * HW + r * P.W + f4 * 4;
            info = (ci0 + cil < P.Ci) ? (r | (img << 8)) : -1;
            l = cil * P.CS + (img * P.SR + r) * P.SCp + 2 * f4 + 2;     // even-plane index of slab col 4*f4+4
        }
        xrel[j] = rel; xinfo[j] = info; xlds[j] = l;
    }
    // gy tile: one float4 (4 consecutive pixels) per thread
    const int gcol = fdiv(tid, P.d_PC4);
    const int gp4 = tid - gcol * (P.PC >> 2);
    int grel, gimg, gr;
    {
        const int p = gp4 * 4;
        gimg = fdiv(p, P.d_TRWo);
        const int rem = p - gimg * (P.TR * P.Wo);
        gr = fdiv(rem, P.d_Wo);
        const int c = rem - gr * P.Wo;
        grel = (gimg * P.Co + gcol) * HoWo + gr * P.Wo + c;
    }
    const bool gvalid = gcol < TCO && (co0 + gcol) < P.Co;

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int xa = l15 * P.CS + qq;        // x-fragment lane base
    const int gb = l15 * P.PCp + qq;       // gy-fragment lane base

    u32x4 xreg[XU_MAX];
    u32x4 greg, gref;

    // loads of one chunk: XU_MAX slab units, the gy quad, its mask reference -- `live` false: every lane out of range (nothing fetched)
    struct ChunkBase { int n0, oh0, in_row0, xbase; bool live; };
    auto chunk_base = [&](int ch, bool live) {
        ChunkBase b;
        const int ig = ch / P.row_tiles, rt = ch - ig * P.row_tiles;
        b.n0 = ig * P.TI; b.oh0 = rt * P.TR;
        b.in_row0 = 2 * b.oh0 - P.pad_t;
        b.xbase = (b.n0 * P.Ci + ci0) * HW + b.in_row0 * P.W;
        b.live = live;
        return b;
    };
    auto pf_item = [&](const ChunkBase& b, int i) {
        if (i < XU_MAX) {
            const int r = xinfo[i] & 255, img = (xinfo[i] >> 8) & 255;
            const bool ok = b.live && xinfo[i] >= 0 && (unsigned)(b.in_row0 + r) < (unsigned)P.H && (b.n0 + img) < P.N;
            const unsigned vo = ok ? (unsigned)(b.xbase + xrel[i]) * 4u : OOB;
            xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0);
        } else {
            const bool gok = b.live && gvalid && (b.n0 + gimg) < P.N && (b.oh0 + gr) < P.Ho;
            const unsigned gvo = gok ? (unsigned)((b.n0 * P.Co + co0) * HoWo + b.oh0 * P.Wo + grel) * 4u : OOB;
            if (i == XU_MAX) greg = __builtin_amdgcn_raw_buffer_load_b128(rg, gvo, 0, 0);
            else gref = __builtin_amdgcn_raw_buffer_load_b128(rr, masked ? gvo : OOB, 0, 0);   // (unmasked: no fetch, the value is unused)
        }
    };
    auto prefetch = [&](int ch) {
        const ChunkBase b = chunk_base(ch, true);
#pragma unroll
        for (int i = 0; i < XU_MAX + 1; ++i) pf_item(b, i);
        if (masked) pf_item(b, XU_MAX + 1);
    };

    auto commit = [&](int buf) {
        float* xs = smem + buf * STG;
        float* gs = xs + TCI * P.CS;
#pragma unroll
        for (int j = 0; j < XU_MAX; ++j) {
            if (tid + j * NTHR < P.xunits) {
                u32x2 ev = {xreg[j].x, xreg[j].z}, od = {xreg[j].y, xreg[j].w};
                *reinterpret_cast<u32x2*>(xs + xlds[j]) = ev;
                *reinterpret_cast<u32x2*>(xs + xlds[j] + P.SCh) = od;
            }
        }
        if (gcol < TCO) {
            float4 gv = make_float4(__uint_as_float(greg.x), __uint_as_float(greg.y), __uint_as_float(greg.z), __uint_as_float(greg.w));
            if (masked) {       // lrelu / relu derivative (launcher admits no other mask): a select, no per-element branches
                gv.x = __uint_as_float(gref.x) > 0.f ? gv.x : gv.x * mslope;
                gv.y = __uint_as_float(gref.y) > 0.f ? gv.y : gv.y * mslope;
                gv.z = __uint_as_float(gref.z) > 0.f ? gv.z : gv.z * mslope;
                gv.w = __uint_as_float(gref.w) > 0.f ? gv.w : gv.w * mslope;
            }
            *reinterpret_cast<float4*>(gs + gcol * P.PCp + gp4 * 4) = gv;
            // bias gradient: this thread's 4 pixels; the PC/4 threads of a channel are contiguous lanes (power of two <= 16)
            if (do_bias) bsum += (gv.x + gv.y) + (gv.z + gv.w);
        }
    };

    const int c_begin = split * P.chunks_per_split;
    const int c_end = min(c_begin + P.chunks_per_split, P.chunks_total);
    stamp(1);
    if (c_begin < c_end) prefetch(c_begin);
    // zero the slabs once (halo columns and padded channels are never written again) -- under the latency of the first loads;
    // float4 stores (CS is even, the slab base 16-byte aligned: pairs of channels)
    {
        const int n4 = (TCI * P.CS) >> 2;
        float4* z0 = reinterpret_cast<float4*>(smem);
        float4* z1 = reinterpret_cast<float4*>(smem + STG);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = tid; e < n4; e += NTHR) { z0[e] = z; z1[e] = z; }
    }
    __syncthreads();                        // (slabs zeroed)
    if (c_begin < c_end) commit(0);
    if (c_begin + 1 < c_end) prefetch(c_begin + 1);
    __syncthreads();
    stamp(2);
    // the two waves that share a SIMD run the chunk in opposite order (stage-then-multiply / multiply-then-stage): the LDS-write
    // and vector-memory burst of one overlaps the MFMA phase of the other
    const bool stage_first = wave < NW / 2;
    const int nq = P.PC >> 2;
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int buf = (ch - c_begin) & 1;
        const float* xs = smem + buf * STG;
        const float* gs = xs + TCI * P.CS;
        float av[2][NT], bv[2];
        // ---- MFMA: this wave's pixel quads; the 26 fragment reads of the NEXT quad are dealt out between the 25 MFMAs of the
        //      current one (a burst of 26 reads in front of the MFMA block costs the wave ~200 issue cycles per quad with the
        //      matrix pipe waiting on the other wave alone) -----------------------------------------------------------------------
        auto load_quad = [&](int qd, float* a, float& b) {
            const int p0 = qd * 4;
            if (GGAN_ABL & 4) { b = __int_as_float(gb + qd); for (int t = 0; t < NT; ++t) a[t] = __int_as_float(xa + t); return; }
            const int img = fdiv(p0, P.d_TRWo);
            const int rem = p0 - img * (P.TR * P.Wo);
            const int r = fdiv(rem, P.d_Wo);
            const int c0 = rem - r * P.Wo;
            b = gs[gb + p0];
            const float* xp = xs + xa + (img * P.SR + 2 * r) * P.SCp + c0;
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int kw = 0; kw < KS; ++kw)   // slab col = 2*ow+kw+3 -> parity plane (kw+3)&1, index ow+((kw+3)>>1)
                    a[kh * KS + kw] = xp[kh * P.SCp + ((kw + 3) & 1) * P.SCh + ((kw + 3) >> 1)];
        };
        if constexpr (NIT > 0) {
            // every wave commits chunk ch+1 first (its registers were filled during chunk ch-1), then multiplies; the loads of
            // chunk ch+2 ride in the MFMA blocks, IPB per block, one every few MFMAs
            constexpr int NB = 2 * NIT, NITEM = XU_MAX + 2, IPB = (NITEM + NB - 1) / NB;
            if (!(GGAN_ABL & 1) && ch + 1 < c_end) commit(buf ^ 1);
            const ChunkBase nb = chunk_base(ch + 2, ch + 2 < c_end && !(GGAN_ABL & 1));
            auto block = [&](int bi, const float* a, float bq) {
#pragma unroll
                for (int i = bi * IPB; i < (bi + 1) * IPB; ++i)
                    if (i < NITEM) pf_item(nb, i);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bq, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (t % 6 == 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            };
            load_quad(wave, av[0], bv[0]);
#pragma unroll
            for (int itq = 0; itq < NIT; ++itq) {
                const int qd = wave + itq * 2 * NW;
                load_quad(qd + NW, av[1], bv[1]);
                block(2 * itq, av[0], bv[0]);
                load_quad(itq + 1 < NIT ? qd + 2 * NW : qd, av[0], bv[0]);     // (after the last pair: re-read, dropped)
                block(2 * itq + 1, av[1], bv[1]);
            }
        } else {
            auto stage_next = [&]() {
                if (!(GGAN_ABL & 1) && ch + 1 < c_end) {
                    commit(buf ^ 1);
                    if (ch + 2 < c_end) prefetch(ch + 2);
                }
            };
            // the two waves that share a SIMD run the chunk in opposite order (stage-then-multiply / multiply-then-stage)
            if (stage_first) stage_next();
            auto interleave = [&]() {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            };
            if (wave < nq) load_quad(wave, av[0], bv[0]);
            for (int qd = wave; qd < nq; qd += 2 * NW) {
                // (unconditional: reads and MFMAs must sit in one basic block to be interleaved; beyond the last quad the wave
                //  re-reads its current one and the values are dropped)
                load_quad(qd + NW < nq ? qd + NW : qd, av[1], bv[1]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0][t], bv[0], acc[t], 0, 0, 0);
                interleave();
                if (qd + NW < nq) {
                    load_quad(qd + 2 * NW < nq ? qd + 2 * NW : qd, av[0], bv[0]);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1][t], bv[1], acc[t], 0, 0, 0);
                    interleave();
                }
            }
            if (!stage_first) stage_next();
        }
        __syncthreads();
        if (ch - c_begin < 8) stamp(4 + ch - c_begin);
    }

    stamp(12);
    if (do_bias) {
        // threads gcol*(PC/4) .. +PC/4-1 hold the partial sums of channel gcol: xor-shuffle within that lane group
        const int grp = P.PC >> 2;
        for (int o = grp >> 1; o > 0; o >>= 1) bsum += __shfl_xor(bsum, o, 64);
        if (gcol < TCO && gp4 == 0 && co0 + gcol < P.Co) {
            if (P.SK == 1) P.gbias[co0 + gcol] = bsum;
            else P.out[(size_t)split * P.slab_stride + P.out_elems + co0 + gcol] = bsum;
        }
    }

    // ---- combine the NW pixel-split waves.  One halving round with 16-byte accesses (wave w+4 parks its 25 accumulator quads,
    //      wave w adds them: any layout serves, the two waves hold the same elements in the same lanes), then the four
    //      remaining partial tiles go to LDS transposed for the store phase, where ALL waves add the four and store float4 runs
    //      along co (one wave storing 100 dwords per lane is a ~50k-cycle issue-bound tail).  Three barriers; the previous
    //      three halving rounds of dword accesses (seven barriers, two waves busy in the last ones) took 9400 cycles.
    __syncthreads();
    float* red = smem;                     // [NW/2][NT*4][64]
    static_assert(NW == 8, "one halving round, four partial tiles");
    {
        f32x4* slab = reinterpret_cast<f32x4*>(red) + (size_t)(wave & 3) * NT * 64 + lane;
        if (wave >= 4) {
#pragma unroll
            for (int t = 0; t < NT; ++t) slab[t * 64] = acc[t];
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 o = slab[t * 64];
                acc[t][0] += o[0]; acc[t][1] += o[1]; acc[t][2] += o[2]; acc[t][3] += o[3];
            }
            // (slab w is read and rewritten by wave w alone: no barrier between the two)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(wave * NT * 4 + t * 4 + r) * 64 + lane] = acc[t][r];
        }
    }
    __syncthreads();
    stamp(13);
    if (P.dbg_nostore) return;
    // element (t, ci_l, co_l) sits at red[w][(t*4 + (ci_l&3))*64 + (ci_l>>2)*16 + co_l]
    float* outp = P.out + (size_t)split * P.slab_stride;
    const bool vec = (P.Co & 3) == 0;
    for (int u = tid; u < NT * TCI * (TCO / 4); u += NTHR) {
        const int c4 = u & 3, cil = (u >> 2) & 15, t = u >> 6;
        const int idx = (t * 4 + (cil & 3)) * 64 + (cil >> 2) * 16 + c4 * 4;
        const float4 a = *reinterpret_cast<const float4*>(red + idx);
        const float4 b2 = *reinterpret_cast<const float4*>(red + NT * 4 * 64 + idx);
        const float4 c2 = *reinterpret_cast<const float4*>(red + 2 * NT * 4 * 64 + idx);
        const float4 d2 = *reinterpret_cast<const float4*>(red + 3 * NT * 4 * 64 + idx);
        // ((w0 + w4) + (w2 + w6)) + ((w1 + w5) + (w3 + w7)): the order of the halving rounds this replaces
        const float4 v = make_float4((a.x + c2.x) + (b2.x + d2.x), (a.y + c2.y) + (b2.y + d2.y), (a.z + c2.z) + (b2.z + d2.z),
                                     (a.w + c2.w) + (b2.w + d2.w));
        const int ci = ci0 + cil, co = co0 + c4 * 4;
        if (ci >= P.Ci || co >= P.Co) continue;
        float* dst = outp + ((size_t)t * P.Ci + ci) * P.Co + co;
        if (vec && co + 3 < P.Co) {
            *reinterpret_cast<float4*>(dst) = v;
        } else {
            dst[0] = v.x;
            if (co + 1 < P.Co) dst[1] = v.y;
            if (co + 2 < P.Co) dst[2] = v.z;
            if (co + 3 < P.Co) dst[3] = v.w;
        }
    }
    stamp(14);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: the same product on FOUR waves -- one per SIMD, so a wave has the whole 512-register file -- with a 16 ci x 32 co tile
// for all 25 taps per wave (200 accumulator registers).  What that buys over the eight-wave kernel above:
//   * the x slab is staged once for twice the MFMAs (each x element feeds 32 output channels instead of 16), and a k-step is
//     25 x-fragment + 2 gy-fragment LDS reads for 50 MFMAs instead of 26 reads for 25;
//   * a chunk carries twice the MFMA issue time (12 800 cycles per 128 pixels) between the barrier-synchronous chunk ends;
//   * gy is the MFMA's A operand here, so an accumulator quad is 4 CONSECUTIVE output channels of one (tap, ci): the four
//     pixel-split waves combine in the accumulators' own layout with 16-byte LDS accesses (every wave parks the 3/4 it does not
//     own, adds the three foreign copies of the quarter it owns in wave order) and store float4 runs straight from registers --
//     no transposed round trip;
//   * the geometry is a template parameter (square images of width 8 / 16 / 32 / 64, the layers of the scripts): every fragment
//     read is one base register + an immediate offset, and the chunk body has no branch, so the fragment reads of the next quad
//     and the global loads of the chunk after next are dealt out between the MFMAs of ONE scheduling region (one wave per SIMD
//     hides at most ~5 single-issue instructions per 32-cycle MFMA: MI355X_MICROARCH.md);
//   * workgroups are numbered split-fastest: the (ci, co) tiles that read the same image range sit on ONE XCD (id % 8), whose L2
//     then fetches that range of x and gy once for all of them.
// Workgroup = (16 ci) x (32 co) x (a split-K range of pixel chunks); the four waves take the pixel quads w, w + 4, ... of a chunk.
constexpr int W4_NW = 4, W4_NTHR = 64 * W4_NW, W4_TCO = 32;
constexpr int W4_RED_BYTES = 4 * 3 * 13 * 64 * 16;      // epilogue: [owner][foreign slot][unit / 4][lane] accumulator quads

template <int W_> struct W4Geom {
    static constexpr int W = W_, H = W_, Wo = W_ / 2, Ho = W_ / 2;
    static constexpr int PC = W_ == 8 ? 64 : 128;                    // pixels per chunk (two staging buffers must fit 160 KB)
    static constexpr int TR = PC / Wo < Ho ? PC / Wo : Ho;            // output rows per chunk
    static constexpr int TI = PC / (TR * Wo);                         // images per chunk
    static constexpr int SR = 2 * (TR - 1) + 5, SCp = W + 8, SCh = SCp / 2;
    static constexpr int CS0 = TI * SR * SCp;
    static constexpr int CS = CS0 + ((34 - (CS0 & 31)) & 31);         // == 2 (mod 32): conflict-free 16 ci x 2 px fragment reads
    static constexpr int NQ = PC / 16;                                // pixel quads per wave and chunk
    static constexpr int ROWT = Ho / TR;                              // chunks per image group
    static constexpr int F4 = W / 4;
    static constexpr int XUNITS = TCI * TI * SR * F4;                 // float4 slab units per chunk
    static constexpr int XU = (XUNITS + W4_NTHR - 1) / W4_NTHR;
    static constexpr int PC4 = PC / 4, PCp = PC + 4;
    static constexpr int GU = W4_TCO * PC4 / W4_NTHR;                 // gy float4 units per thread and chunk
    static constexpr int GSTEP = W4_NTHR / PC4;                       // output channels between a thread's gy units
    static constexpr int STG = TCI * CS + W4_TCO * PCp;               // one staging buffer: x slab + gy tile (floats)
    // pixel quad qd = wave + 4 q of a chunk -> slab offset of its first pixel = wave part (run time) + q part (compile time)
    static constexpr int q_img(int q) { return (16 * q) / (TR * Wo); }
    static constexpr int q_row(int q) { return ((16 * q) % (TR * Wo)) / Wo; }
    static constexpr int q_col(int q) { return (16 * q) % Wo; }
    static constexpr int q_off(int q) { return (q_img(q) * SR + 2 * q_row(q)) * SCp + q_col(q); }
    static_assert(Wo >= 4 && (Wo & (Wo - 1)) == 0 && TR * Wo * TI == PC && Ho % TR == 0, "power-of-two chunk geometry");
    static_assert(2 * STG * 4 <= 160 * 1024 && TCI * CS * 4 < 65536, "two staging buffers in LDS, slab within the immediate range");
};

template <int W> struct WaveTag { static constexpr int value = W; };

// SPLIT (round 5, experiment GGAN_WGRAD_SPLIT=1): eight waves with the roles split -- waves 0-3 (one per SIMD) only read fragments and
// multiply, waves 4-7 (their SIMD partners) only stage the next chunk -- so that no staging instruction sits in an MFMA stream.
template <int GW, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? 2 * W4_NTHR : W4_NTHR) void wgrad4_kernel(const WgradParams P) {
    using G = W4Geom<GW>;
    warm_kernarg(P);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KS = 5, NT = 25, NA = 2 * NT, NQ = G::NQ;
    constexpr int XU = G::XU, GU = G::GU, NTHR = W4_NTHR, NITEM = XU + 2 * GU;
    constexpr int HW = G::H * G::W, HoWo = G::Ho * G::Wo;
    const int tid_all = threadIdx.x, lane = tid_all & 63, wave_all = tid_all >> 6;
    const bool prod = !SPLIT || wave_all >= 4, cons = !SPLIT || wave_all < 4;
    const int tid = SPLIT ? (tid_all & (NTHR - 1)) : tid_all;        // staging thread id (producers) / thread id among the multiplying waves
    const int wave = wave_all & 3;                                   // pixel-quad owner index of a multiplying wave
    constexpr int NALL = SPLIT ? 2 * NTHR : NTHR;
    const int l15 = lane & 15, qq = lane >> 4;
    // split-fastest numbering (XCD = id % 8): the tiles of one image range share an L2
    const int split = blockIdx.x % P.SK, tile = blockIdx.x / P.SK;
    const int gxt = (P.Ci + TCI - 1) / TCI;
    const int ci0 = (tile % gxt) * TCI, co0 = (tile / gxt) * W4_TCO;

    const bool stamping = P.stamps != nullptr && tid_all == 0;
    auto stamp = [&](int i) { if (stamping) P.stamps[(size_t)blockIdx.x * 16 + i] = __builtin_readcyclecounter(); };
    stamp(0);

    // (the x buffer starts pad_t rows above the tensor: slab row 0 of the first row tile is a padding row, and every offset stays >= 0)
    const int lead = P.pad_t * G::W;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)(P.x - lead), (short)0, (int)(P.x_bytes + 4u * lead), 0x00020000);
    const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)P.gy, (short)0, (int)P.gy_bytes, 0x00020000);
    const bool masked = P.gy_ref != nullptr;
    // the mask is a select  ref > 0 ? g : g * mslope  on every path: unmasked launches fetch no reference (zeros) and multiply by 1
    const float mslope = !masked ? 1.f : (P.gy_act == GGAN_ACT_LRELU ? P.gy_alpha : 0.f);
    const auto rr = __builtin_amdgcn_make_buffer_rsrc((void*)(masked ? P.gy_ref : P.gy), (short)0, (int)P.gy_bytes, 0x00020000);
    const bool do_bias = P.gbias != nullptr && ci0 == 0;
    float bsum[GU];
#pragma unroll
    for (int j = 0; j < GU; ++j) bsum[j] = 0.f;

    // ---- staging descriptors: slab unit u = tid + j * NTHR -> (channel, image, slab row, float4 column); all divisors compile-time.
    //      Units past the slab duplicate its last unit (same bytes to the same place: no predicate anywhere in the chunk body).
    //      A load is ONE instruction in the chunk body: voffset = the unit's byte offset inside a chunk (fixed for the launch; bit 31 set
    //      where nothing is to be fetched -- channel past Ci, padding row: past every buffer, the load returns zeros), soffset = the
    //      chunk's base (scalar).  One wave per SIMD hides nothing: every VALU instruction between two MFMAs costs ~6 cycles of the
    //      matrix pipe (profiles/r05_notes.md), so the chunk body computes no addresses. ----
    constexpr unsigned DEAD = 0x80000000u;
    unsigned xvo[XU];                   // global byte offset of the unit relative to the chunk's x base
    int xl[XU];                         // LDS float index (even plane) in staging buffer 0
    unsigned xrow[XU];                  // ROWT > 1 only: bit 31 / 30 = a padding row in the first / last row tile of an image
#pragma unroll
    for (int j = 0; j < XU; ++j) {
        const int u = min(tid + j * NTHR, G::XUNITS - 1);
        const int row = u / G::F4, f4 = u % G::F4;
        const int t = row / G::SR, r = row % G::SR;
        const int cil = t / G::TI, img = t % G::TI;
        xl[j] = cil * G::CS + (img * G::SR + r) * G::SCp + 2 * f4 + 2;           // even-plane index of slab col 4*f4+4
        // x row of slab row r in row tile rt: 2 * TR * rt - pad_t + r; with one row tile per image the padding rows are fixed
        const bool top = r < P.pad_t, bot = 2 * G::TR * (G::ROWT - 1) - P.pad_t + r >= G::H;
        unsigned vo = (unsigned)((img * P.Ci + cil) * HW + r * G::W + f4 * 4) * 4u;
        if (ci0 + cil >= P.Ci) vo |= DEAD;
        if (G::ROWT == 1 && (top || bot)) vo |= DEAD;
        xvo[j] = vo;
        xrow[j] = G::ROWT > 1 ? ((top ? DEAD : 0u) | (bot ? 0x40000000u : 0u)) : 0u;
    }
    // gy tile: unit j of a thread = one float4 (4 consecutive pixels) of channel gcol0 + j*GSTEP
    const int gcol0 = tid / G::PC4, gp4 = tid % G::PC4;
    const int gimg = (gp4 * 4) / (G::TR * G::Wo), gr = ((gp4 * 4) % (G::TR * G::Wo)) / G::Wo, gc = (gp4 * 4) % G::Wo;
    unsigned gvo[GU];
#pragma unroll
    for (int j = 0; j < GU; ++j) {
        const int col = gcol0 + j * G::GSTEP;
        gvo[j] = (unsigned)((gimg * P.Co + col) * HoWo + gr * G::Wo + gc) * 4u | (co0 + col < P.Co ? 0u : DEAD);
    }
    const int gl = TCI * G::CS + gcol0 * G::PCp + gp4 * 4;      // LDS float index of unit 0 in staging buffer 0 (unit j: + j*GSTEP*PCp)

    // fragment lane bases (floats): x (ci = lane & 15, pixel = lane >> 4) + this wave's first quad; gy (co = lane & 15, pixel) likewise
    const int wrow = ((4 * wave) % (G::TR * G::Wo)) / G::Wo, wcol = (4 * wave) % G::Wo;
    const int xa = l15 * G::CS + qq + 2 * wrow * G::SCp + wcol;
    const int gb = TCI * G::CS + l15 * G::PCp + qq + 4 * wave;

    u32x4 sreg[NITEM];                  // staging registers: x units, gy units, their mask references

    // chunk -> scalar bases (bytes) and the row-tile mask; the chunk after the last one is the last one again (valid addresses, its
    // bytes land in the buffer nobody reads any more)
    struct ChunkBase { unsigned xs, gs, rowmask, gdead; };
    auto chunk_base = [&](int ch, bool live) {
        ChunkBase b;
        b.gdead = live ? 0u : DEAD;          // (gy of the repeated chunk reads as zeros: its rows must not enter the bias sums twice)
        const int ig = ch / G::ROWT, rt = ch % G::ROWT;
        b.xs = (unsigned)((ig * G::TI * P.Ci + ci0) * HW + 2 * G::TR * rt * G::W) * 4u;
        b.gs = (unsigned)((ig * G::TI * P.Co + co0) * HoWo + G::TR * rt * G::Wo) * 4u;
        b.rowmask = (rt == 0 ? DEAD : 0u) | (rt == G::ROWT - 1 ? 0x40000000u : 0u);
        return b;
    };
    const unsigned rdead = masked ? 0u : DEAD;
    auto pf_item = [&](const ChunkBase& b, int i) {
        if (i < XU) {
            unsigned vo = xvo[i];
            if constexpr (G::ROWT > 1) {       // bit 31 or 30 of the flags that apply in this row tile -> bit 31 of the offset
                const unsigned f = xrow[i] & b.rowmask;
                vo |= f | (f << 1);
            }
            sreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, b.xs, 0);
        } else if (i < XU + GU) {
            sreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rg, gvo[i - XU] | b.gdead, b.gs, 0);
        } else {
            sreg[i] = __builtin_amdgcn_raw_buffer_load_b128(rr, gvo[i - XU - GU] | rdead, b.gs, 0);   // (unmasked: no fetch)
        }
    };
    // the LDS side of the same items.  The gy unit is committed with its reference (item XU + GU + j; item XU + j alone commits nothing)
    auto commit_item = [&](int bo, int i) {        // bo: float offset of the staging buffer
        if (i < XU) {
            unsigned* d = reinterpret_cast<unsigned*>(smem) + bo + xl[i];
            d[0] = sreg[i].x; d[1] = sreg[i].z;
            d[G::SCh] = sreg[i].y; d[G::SCh + 1] = sreg[i].w;
        } else if (i >= XU + GU) {
            const int j = i - XU - GU;
            const u32x4 g = sreg[XU + j], rf = sreg[i];
            float4 gv = make_float4(__uint_as_float(g.x), __uint_as_float(g.y), __uint_as_float(g.z), __uint_as_float(g.w));
            if (GGAN_ABL & 2) { *reinterpret_cast<float4*>(smem + bo + gl + j * G::GSTEP * G::PCp) = gv; return; }
            gv.x = __uint_as_float(rf.x) > 0.f ? gv.x : gv.x * mslope;
            gv.y = __uint_as_float(rf.y) > 0.f ? gv.y : gv.y * mslope;
            gv.z = __uint_as_float(rf.z) > 0.f ? gv.z : gv.z * mslope;
            gv.w = __uint_as_float(rf.w) > 0.f ? gv.w : gv.w * mslope;
            *reinterpret_cast<float4*>(smem + bo + gl + j * G::GSTEP * G::PCp) = gv;
            bsum[j] += (gv.x + gv.y) + (gv.z + gv.w);
        }
    };

    const int c_begin = split * P.chunks_per_split;
    const int c_end = min(c_begin + P.chunks_per_split, P.chunks_total);
    stamp(1);
    if (c_begin >= c_end) return;        // (the host plans no empty split)
    if constexpr (SPLIT) {
        // The staging waves' whole program: nothing of it is reachable from the multiplying waves' code and the reverse, so neither
        // role carries the other's registers (200 accumulators here would leave the staging set in scratch).  Both roles pass the
        // same number of workgroup barriers: two in the prologue, one per chunk, one in the epilogue.
        if (wave_all >= 4) {
            // What a staging instruction costs the partner wave's MFMA stream (tools/scratch/mfma_filler.hip, profiles/r05_notes.md): LDS and
            // scalar instructions nothing, a VALU instruction ~10 cycles.  So this loop has next to no VALU: LDS byte addresses of both
            // buffers sit in registers (the loop is unrolled over the buffer parity), an x unit leaves as two ds_write2_b32 straight out
            // of the loaded registers (the compiler's ds_write2_b64 first shuffled them into pairs: 3 v_mov per unit), the activation mask
            // and the bias sums are wave-uniform branches, and an unmasked launch does not issue the reference loads at all.
            const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
            unsigned xaddr[2][XU];
#pragma unroll
            for (int j = 0; j < XU; ++j) { xaddr[0][j] = lds0 + 4u * (unsigned)xl[j]; xaddr[1][j] = xaddr[0][j] + 4u * G::STG; }
            auto fetch = [&](const ChunkBase& b) {
#pragma unroll
                for (int i = 0; i < XU + GU; ++i) pf_item(b, i);
                if (masked) {
#pragma unroll
                    for (int i = XU + GU; i < NITEM; ++i) pf_item(b, i);
                }
            };
            auto commit = [&](auto par) {
                constexpr int B = decltype(par)::value;
#pragma unroll
                for (int i = 0; i < XU; ++i) {
                    const unsigned x0 = sreg[i].x, x1 = sreg[i].y, x2 = sreg[i].z, x3 = sreg[i].w;
                    asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" : : "v"(xaddr[B][i]), "v"(x0), "v"(x2) : "memory");
                    asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(xaddr[B][i]), "v"(x1), "v"(x3), "n"(G::SCh), "n"(G::SCh + 1) : "memory");
                }
#pragma unroll
                for (int j = 0; j < GU; ++j) {
                    const u32x4 g = sreg[XU + j];
                    float4 gv = make_float4(__uint_as_float(g.x), __uint_as_float(g.y), __uint_as_float(g.z), __uint_as_float(g.w));
                    if (masked && !(GGAN_ABL & 2)) {
                        const u32x4 rf = sreg[XU + GU + j];
                        gv.x = __uint_as_float(rf.x) > 0.f ? gv.x : gv.x * mslope;
                        gv.y = __uint_as_float(rf.y) > 0.f ? gv.y : gv.y * mslope;
                        gv.z = __uint_as_float(rf.z) > 0.f ? gv.z : gv.z * mslope;
                        gv.w = __uint_as_float(rf.w) > 0.f ? gv.w : gv.w * mslope;
                    }
                    *reinterpret_cast<float4*>(smem + B * G::STG + gl + j * G::GSTEP * G::PCp) = gv;
                    if (do_bias) bsum[j] += (gv.x + gv.y) + (gv.z + gv.w);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the x stores above are not the compiler's to wait for)
            };
            fetch(chunk_base(c_begin, true));
            {
                constexpr int n4 = (TCI * G::CS) >> 2;
                float4* z0 = reinterpret_cast<float4*>(smem);
                float4* z1 = reinterpret_cast<float4*>(smem + G::STG);
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int e = tid_all; e < n4; e += NALL) { z0[e] = z; z1[e] = z; }
            }
            __syncthreads();
            commit(WaveTag<0>{});
            __syncthreads();
            for (int ch = c_begin; ch < c_end; ch += 2) {
                if (!(GGAN_ABL & 1) && ch + 1 < c_end) {        // (behind the last chunk there is nothing to stage)
                    fetch(chunk_base(ch + 1, true));
                    commit(WaveTag<1>{});
                }
                __syncthreads();
                if (ch + 1 >= c_end) break;
                if (!(GGAN_ABL & 1) && ch + 2 < c_end) {
                    fetch(chunk_base(ch + 2, true));
                    commit(WaveTag<0>{});
                }
                __syncthreads();
            }
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < GU; ++j) {
                    float b = bsum[j];
#pragma unroll
                    for (int o = G::PC4 >> 1; o > 0; o >>= 1) b += __shfl_xor(b, o, 64);
                    const int co = co0 + gcol0 + j * G::GSTEP;
                    if (gp4 == 0 && co < P.Co) {
                        if (P.SK == 1) P.gbias[co] = b;
                        else P.out[(size_t)split * P.slab_stride + P.out_elems + co] = b;
                    }
                }
            }
            __syncthreads();
            return;
        }
    }
    f32x4 acc[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (prod) {
        const ChunkBase b0 = chunk_base(c_begin, true);
#pragma unroll
        for (int i = 0; i < NITEM; ++i) pf_item(b0, i);
    }
    {   // zero the slabs once (halo columns and padded channels are never written again), under the latency of the first loads
        constexpr int n4 = (TCI * G::CS) >> 2;
        float4* z0 = reinterpret_cast<float4*>(smem);
        float4* z1 = reinterpret_cast<float4*>(smem + G::STG);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = tid_all; e < n4; e += NALL) { z0[e] = z; z1[e] = z; }
    }
    __syncthreads();
    if (prod) {
#pragma unroll
        for (int i = 0; i < NITEM; ++i) commit_item(0, i);
    }
    __syncthreads();
    stamp(2);

    // A chunk: NQ blocks of 50 MFMAs out of buffer `buf`.  Chunk ch + 1 travels through registers meanwhile: its loads are dealt out over
    // the first blocks, each item's LDS write follows LAG blocks behind its load (3 200 cycles at LAG = 2: a load that misses every cache
    // is back in ~1 000), so an item occupies registers for a fraction of the chunk and the file holds no second set.
    constexpr int LAG = 2, LB = NQ - LAG;                 // load blocks 0 .. LB-1, commit blocks LAG .. NQ-1
    constexpr int IPB = (NITEM + LB - 1) / LB;
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int buf = (ch - c_begin) & 1;
        const float* xq = smem + buf * G::STG + xa;        // + q_off(q) + tap offset: immediates
        const float* gq = smem + buf * G::STG + gb;        // + 16 q (+ 16 * PCp for the second channel tile)
        const int obo = (buf ^ 1) * G::STG;
        if constexpr (SPLIT) {
            {
                // multiplying waves: ONE set of fragment registers, a fragment refilled for the next quad right behind its two MFMAs
                // (200 accumulators + 27 fragments have to fit the 256 registers of a wave that shares its SIMD)
                float a[NT], b0, b1;
                auto tap = [&](int q, int t) { if (GGAN_ABL & 4) return __int_as_float(xa + t + q); return xq[G::q_off(q) + (t / KS) * G::SCp + (((t % KS) + 3) & 1) * G::SCh + (((t % KS) + 3) >> 1)]; };
                b0 = gq[0]; b1 = gq[16 * G::PCp];
#pragma unroll
                for (int t = 0; t < NT; ++t) a[t] = tap(0, t);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    float nb0 = 0.f, nb1 = 0.f;
                    if (q + 1 < NQ) { nb0 = gq[16 * (q + 1)]; nb1 = gq[16 * (q + 1) + 16 * G::PCp]; }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0, a[t], acc[t], 0, 0, 0);
                        acc[NT + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1, a[t], acc[NT + t], 0, 0, 0);
                        if (q + 1 < NQ) a[t] = tap(q + 1, t);
                        // keep each refill where it is written (the scheduler otherwise sinks the read to just above its use a quad
                        // later and the wave waits out the LDS latency with the matrix pipe idle)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    b0 = nb0; b1 = nb1;
                }
            }
            __syncthreads();
            if (ch - c_begin < 8) stamp(4 + ch - c_begin);
            continue;
        }
        float av[2][NT], bv[2][2];
        auto load_quad = [&](int q, float* a, float* b) {
            if (GGAN_ABL & 4) { b[0] = __int_as_float(gb + q); b[1] = b[0]; for (int t = 0; t < NT; ++t) a[t] = __int_as_float(xa + t); return; }
            b[0] = gq[16 * q];
            b[1] = gq[16 * q + 16 * G::PCp];
            const int o = G::q_off(q);
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int kw = 0; kw < KS; ++kw)   // slab col = 2*ow+kw+3 -> parity plane (kw+3)&1, index ow+((kw+3)>>1)
                    a[kh * KS + kw] = xq[o + kh * G::SCp + ((kw + 3) & 1) * G::SCh + ((kw + 3) >> 1)];
        };
        const ChunkBase nb = chunk_base(min(ch + 1, c_end - 1), ch + 1 < c_end);
        load_quad(0, av[0], bv[0]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float* a = av[q & 1];
            const float* b = bv[q & 1];
            if (q + 1 < NQ) load_quad(q + 1, av[(q + 1) & 1], bv[(q + 1) & 1]);
            if (!(GGAN_ABL & 1)) {
#pragma unroll
                for (int i = (q - LAG) * IPB; i < (q - LAG + 1) * IPB; ++i)
                    if (i >= 0 && i < NITEM) commit_item(obo, i);
#pragma unroll
                for (int i = q * IPB; i < (q + 1) * IPB; ++i)
                    if (i < NITEM && q < LB) pf_item(nb, i);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[0], a[t], acc[t], 0, 0, 0);
                acc[NT + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[1], a[t], acc[NT + t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < NA; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (q + 1 < NQ && t < NT + 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (t % 8 == 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // a global load
                if (t % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // an LDS write
            }
        }
        __syncthreads();
        if (ch - c_begin < 8) stamp(4 + ch - c_begin);
    }

    stamp(12);
    if (do_bias && prod) {
        // lanes gcol*PC4 .. +PC4-1 hold the partial sums of one channel: xor-shuffle within that lane group (PC4 = 32 / 16)
#pragma unroll
        for (int j = 0; j < GU; ++j) {
            float b = bsum[j];
#pragma unroll
            for (int o = G::PC4 >> 1; o > 0; o >>= 1) b += __shfl_xor(b, o, 64);
            const int co = co0 + gcol0 + j * G::GSTEP;
            if (gp4 == 0 && co < P.Co) {
                if (P.SK == 1) P.gbias[co] = b;
                else P.out[(size_t)split * P.slab_stride + P.out_elems + co] = b;
            }
        }
    }

    // ---- combine the four pixel-split waves in the accumulators' own layout (the loop's last barrier retired every staging read).
    //      Unit j (an accumulator quad) is owned by wave j & 3; region [owner][foreign slot] holds 13 units x 64 lanes x 16 bytes.
    //      The parking stores read the accumulators where they live (AGPRs): written as the instruction, because the compiler's own
    //      stores first copied all 200 quads into VGPRs (800 v_accvgpr_read, ~1 700 cycles, and spills once both files were full). ----
    constexpr int REGION = 13 * 64 * 16;           // bytes
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    auto park = [&](auto tag) {
        constexpr int W = decltype(tag)::value;
        const unsigned lane16 = (unsigned)lane * 16u;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o == W) continue;
            const unsigned base = lane16 + (unsigned)(o * 3 * REGION);
#pragma unroll
            for (int j = o; j < NA; j += 4) {
                const f32x4 q = acc[j];         // (an asm operand cannot name a captured variable)
                asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(base), GGAN_ACC_REG(q), "n"((((W - o) & 3) - 1) * REGION + (j >> 2) * 1024) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    float* outp = P.out + (size_t)split * P.slab_stride;
    const bool vec = (P.Co & 3) == 0;
    auto sum_store = [&](auto tag) {
        constexpr int W = decltype(tag)::value;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if ((j & 3) != W) continue;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) {      // wave order: the sum does not depend on which wave owns the unit
                const f32x4 o = s == W ? acc[j] : red[((W * 3 + (((s - W) & 3) - 1)) * 13 + (j >> 2)) * 64 + lane];
                if (s == 0) v = o;
                else { v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
            }
            __builtin_amdgcn_sched_barrier(0);      // (one unit at a time)
            const int t = j % NT, ct = j / NT;
            const int ci = ci0 + l15, co = co0 + ct * 16 + 4 * qq;
            if (ci >= P.Ci || co >= P.Co || P.dbg_nostore) continue;
            float* dst = outp + ((size_t)t * P.Ci + ci) * P.Co + co;
            if (vec) {
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
                dst[0] = v[0];
                if (co + 1 < P.Co) dst[1] = v[1];
                if (co + 2 < P.Co) dst[2] = v[2];
                if (co + 3 < P.Co) dst[3] = v[3];
            }
        }
    };
    if (cons) switch (wave) {
        case 0: park(WaveTag<0>{}); break;
        case 1: park(WaveTag<1>{}); break;
        case 2: park(WaveTag<2>{}); break;
        default: park(WaveTag<3>{}); break;
    }
    __syncthreads();
    stamp(13);
    if (cons) switch (wave) {
        case 0: sum_store(WaveTag<0>{}); break;
        case 1: sum_store(WaveTag<1>{}); break;
        case 2: sum_store(WaveTag<2>{}); break;
        default: sum_store(WaveTag<3>{}); break;
    }
    stamp(14);
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

}  // namespace

#if GGAN_WGRAD_SPLIT_TU
namespace ggan {
int wgrad4_split_launch(int W, unsigned grid, size_t shmem, hipStream_t s, const void* params, double fl, double ab) {
    WgradParams P;
    memcpy(&P, params, sizeof(P));
    static std::atomic<unsigned long long> attr_set{0};
    if (first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad4_kernel<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad4_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad4_kernel<32, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad4_kernel<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const dim3 grid4(grid);
    if (W == 16) { GGAN_LAUNCH("wgrad4_kernel<16, true>", fl, ab, (wgrad4_kernel<16, true>), grid4, dim3(2 * W4_NTHR), shmem, s, P); }
    else if (W == 8) { GGAN_LAUNCH("wgrad4_kernel<8, true>", fl, ab, (wgrad4_kernel<8, true>), grid4, dim3(2 * W4_NTHR), shmem, s, P); }
    else if (W == 32) { GGAN_LAUNCH("wgrad4_kernel<32, true>", fl, ab, (wgrad4_kernel<32, true>), grid4, dim3(2 * W4_NTHR), shmem, s, P); }
    else { GGAN_LAUNCH("wgrad4_kernel<64, true>", fl, ab, (wgrad4_kernel<64, true>), grid4, dim3(2 * W4_NTHR), shmem, s, P); }
    return 0;
}
}  // namespace ggan
#else
namespace ggan {

int wgrad4_split_launch(int W, unsigned grid, size_t shmem, hipStream_t s, const void* params, double fl, double ab);

int conv_wgrad_mfma(const ggan_conv_geom& g, const float* x, const float* gy, GyMask m, float* gw, float* gbias, void* ws,
                    size_t ws_bytes, hipStream_t s, WgradParts* parts) {
    if (g.k != 5 || g.stride != 2 || (g.Wo & 3) || (g.W & 3) || g.pad_l != 1 || g.Wo > 64) return 1;
    if (parts) {   // leave the split-K slabs (bias-gradient tail after each) in the caller's buffer: no reduce launch
        ws = parts->buf;
        ws_bytes = parts->cap_floats * sizeof(float);
        gw = parts->buf;
        gbias = parts->with_bias ? parts->buf + (size_t)25 * g.Ci * g.Co : nullptr;
    } else {
        ws = ws_scratch(ws, ws_bytes);
    }
    const size_t xb = (size_t)g.N * g.Ci * g.H * g.W * 4, gb = (size_t)g.N * g.Co * g.Ho * g.Wo * 4;
    if (xb >= 0x7FFFFFF0ull || gb >= 0x7FFFFFF0ull) return 1;
    if ((((uintptr_t)x) & 15) || (((uintptr_t)gy) & 15)) return 1;
    WgradParams P;
    memset(&P, 0, sizeof(P));
    P.x = x; P.gy = gy;
    if (m.act != GGAN_ACT_NONE && m.act != GGAN_ACT_LRELU && m.act != GGAN_ACT_RELU) return 1;   // other masks: plain path
    if (m.act != GGAN_ACT_NONE) { P.gy_ref = m.ref; P.gy_act = m.act; P.gy_alpha = m.alpha; }
    P.gbias = gbias;
    P.dbg_nostore = env_int("GGAN_DBG", 0) & 8;
    const bool want_stamps = (env_int("GGAN_DBG", 0) & 4) != 0;
    P.x_bytes = (unsigned)xb; P.gy_bytes = (unsigned)gb;
    P.N = g.N; P.Ci = g.Ci; P.H = g.H; P.W = g.W; P.Co = g.Co; P.Ho = g.Ho; P.Wo = g.Wo;
    P.pad_t = g.pad_t;
    // pixel chunk: after every barrier all eight waves fetch their first quad's fragments at once (~800 cycles of LDS pipe with the
    // matrix pipes idle); 128 pixels per chunk instead of 64 halves the number of those bursts per MFMA (when two such staging
    // buffers fit the LDS)
    // tco: output channels per workgroup (16: eight-wave kernel, 32: four-wave kernel), nthr: its threads
    auto plan_chunk = [&](int pcmax, int tco, int nthr) {
        P.TR = pcmax / g.Wo; if (P.TR > g.Ho) P.TR = g.Ho;
        P.TI = pcmax / (P.TR * g.Wo); if (P.TI < 1) P.TI = 1; if (P.TI > g.N) P.TI = g.N;
        for (;;) {
            P.SR = 2 * (P.TR - 1) + 5;
            P.SCp = g.W + 8;
            P.xunits = TCI * P.TI * P.SR * (g.W / 4);
            if (P.xunits <= XU_MAX * NTHR && P.TI < 256 && P.SR < 256) break;
            if (P.TI > 1) P.TI = (P.TI + 1) / 2;
            else if (P.TR > 1) P.TR = (P.TR + 1) / 2;
            else return false;
        }
        P.SCh = P.SCp / 2;
        P.CS = P.TI * P.SR * P.SCp;
        while ((P.CS & 31) != 2) P.CS += 1;     // == 2 (mod 32): conflict-free 16ci x 2px fragment reads; even for b64 stores
        P.PC = P.TI * P.TR * g.Wo;
        P.PCp = P.PC + 4;
        if ((P.PC & 3) || (tco == TCO && tco * (P.PC / 4) > nthr)) return false;   // (eight-wave kernel: one gy unit per thread)
        if (gbias && ((P.PC / 4) & (P.PC / 4 - 1))) return false;      // lane-group shuffle needs a power-of-two group
        return 2 * ((size_t)TCI * P.CS + (size_t)tco * P.PCp) * sizeof(float) <= 160 * 1024;
    };
    // the four-wave kernel is compiled per image width (square images of 8 / 16 / 32 / 64: every layer of the scripts); a minibatch that is
    // not a whole number of its chunks' image groups and every other geometry stay on the eight-wave kernel
    auto geom_is = [&](auto tag) {
        using G = decltype(tag);
        return g.W == G::W && g.H == G::H && g.Ho == G::Ho && g.Wo == G::Wo && g.N % G::TI == 0 && plan_chunk(G::PC, W4_TCO, W4_NTHR) &&
               P.TR == G::TR && P.TI == G::TI && P.SR == G::SR && P.SCp == G::SCp && P.CS == G::CS && P.PC == G::PC;
    };
    int four = 0;
    if (env_int("GGAN_WGRAD_W4", 1) != 0) {
        if (geom_is(W4Geom<16>{})) four = 16;
        else if (geom_is(W4Geom<8>{})) four = 8;
        else if (geom_is(W4Geom<32>{})) four = 32;
        else if (geom_is(W4Geom<64>{})) four = 64;
    }
    if (!four && !(env_int("GGAN_WGRAD_PC", 128) >= 128 && plan_chunk(128, TCO, NTHR)) && !plan_chunk(64, TCO, NTHR)) return 1;
    const int tco = four ? W4_TCO : TCO;
    P.d_F4 = make_fastdiv(g.W / 4); P.d_SR = make_fastdiv(P.SR); P.d_TI = make_fastdiv(P.TI);
    P.d_PC4 = make_fastdiv(P.PC / 4); P.d_TRWo = make_fastdiv(P.TR * g.Wo); P.d_Wo = make_fastdiv(g.Wo);
    P.row_tiles = cdiv(g.Ho, P.TR);
    P.chunks_total = cdiv(g.N, P.TI) * P.row_tiles;
    const int gx = cdiv(g.Ci, TCI), gy_ = cdiv(g.Co, tco);
    int sk = env_int("GGAN_WGRAD_SK", 0);
    if (sk <= 0) {
        // (a caller running two conv chains side by side asks for fewer workgroups per launch: ggan_conv_geom.plan_wgs_filter)
        const int wg_target = g.plan_wgs_filter > 0 ? g.plan_wgs_filter : env_int("GGAN_WGRAD_WGS", 256);
        sk = cdiv(wg_target, gx * gy_);
        if (sk > P.chunks_total / 2) sk = P.chunks_total / 2;
        if (sk > 64) sk = 64;
    }
    if (sk < 1) sk = 1;
    P.out_elems = (size_t)25 * g.Ci * g.Co;
    P.slab_stride = P.out_elems + (gbias ? (size_t)g.Co : 0);
    while (sk > 1 && (size_t)sk * P.slab_stride * sizeof(float) > ws_bytes) sk /= 2;
    P.chunks_per_split = cdiv(P.chunks_total, sk);
    P.SK = cdiv(P.chunks_total, P.chunks_per_split);
    P.out = P.SK > 1 ? (float*)ws : gw;
    size_t stage = 2 * ((size_t)TCI * P.CS + (size_t)tco * P.PCp);      // double-buffered
    size_t red = four ? (size_t)W4_RED_BYTES / sizeof(float) : (size_t)(NW / 2) * 100 * 64;
    const size_t shmem = (stage > red ? stage : red) * sizeof(float);
    if (shmem > 160 * 1024) return 1;
    static std::atomic<unsigned long long> attr_set{0};
    if (first_on_device(attr_set)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad4_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad4_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad4_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad4_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const int nit = env_int("GGAN_WGRAD_DEAL", 1) ? (P.PC == 128 ? 2 : (P.PC == 64 ? 1 : 0)) : 0;
    const double fl = 2.0 * g.N * g.Co * g.Ho * g.Wo * (double)g.Ci * 25.0;
    const double ab = 4.0 * ((double)g.N * g.Ci * g.H * g.W + (double)g.N * g.Co * g.Ho * g.Wo + 25.0 * g.Ci * g.Co);
    if (parts && ws_bytes < P.slab_stride * sizeof(float)) { set_error("conv_wgrad: partial-slab buffer too small"); return -1; }
    if (want_stamps && !parts && ws && ws_bytes > (64u << 20)) P.stamps = (unsigned long long*)((char*)ws + ws_bytes - (32u << 20));
    const dim3 grid4(gx * gy_ * P.SK);       // split-fastest workgroup numbering (decoded in the kernel)
    const bool split_roles = env_int("GGAN_WGRAD_SPLIT", 1) != 0;          // (read per call, like GGAN_WGRAD_W4 / _SK / _WGS / _PC)
    const int split_roles_mask = env_int("GGAN_WGRAD_SPLIT_W", 8 | 16 | 32 | 64);
    if (split_roles && four > 0 && (split_roles_mask & four)) { const int rc = wgrad4_split_launch(four, grid4.x, shmem, s, &P, fl, ab); if (rc) return rc; }
    else if (four == 16) { GGAN_LAUNCH("wgrad4_kernel<16>", fl, ab, wgrad4_kernel<16>, grid4, dim3(W4_NTHR), shmem, s, P); }
    else if (four == 8) { GGAN_LAUNCH("wgrad4_kernel<8>", fl, ab, wgrad4_kernel<8>, grid4, dim3(W4_NTHR), shmem, s, P); }
    else if (four == 32) { GGAN_LAUNCH("wgrad4_kernel<32>", fl, ab, wgrad4_kernel<32>, grid4, dim3(W4_NTHR), shmem, s, P); }
    else if (four == 64) { GGAN_LAUNCH("wgrad4_kernel<64>", fl, ab, wgrad4_kernel<64>, grid4, dim3(W4_NTHR), shmem, s, P); }
    else if (nit == 2) { GGAN_LAUNCH("wgrad_kernel<2>", fl, ab, wgrad_kernel<2>, dim3(gx, gy_, P.SK), dim3(NTHR), shmem, s, P); }
    else if (nit == 1) { GGAN_LAUNCH("wgrad_kernel<1>", fl, ab, wgrad_kernel<1>, dim3(gx, gy_, P.SK), dim3(NTHR), shmem, s, P); }
    else { GGAN_LAUNCH("wgrad_kernel<0>", fl, ab, wgrad_kernel<0>, dim3(gx, gy_, P.SK), dim3(NTHR), shmem, s, P); }
    if (parts) {
        parts->n = P.SK;
        parts->stride = P.slab_stride;
        return 0;
    }
    if (P.SK > 1)
        return launch_splitk_reduce((const float*)ws, P.SK, P.out_elems, gw, nullptr, 1, 1, GGAN_ACT_NONE, 0.f, s, P.slab_stride,
                                    gbias, gbias ? (size_t)g.Co : 0);
    return 0;
}

}  // namespace ggan
#endif  // GGAN_WGRAD_SPLIT_TU
