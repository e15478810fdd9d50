// fp32-MFMA "strided correlation" kernel for gfx950 (CDNA4): Conv2D / Deconv2D forward and data-gradients.
//
//   out[n,cn,U,V] = sum_{ck,i,j} in[n,ck, SU*u+DI*i+.., SU*v+DI*j+..] * w[i,j,ck,cn]
//
//   fwd    (Conv2D forward, Deconv2D data-gradient): SU=2, DI=+1, 5x5 taps, ck=Ci, cn=Co, filter rows contiguous in cn.
//   dgrad  (Conv2D data-gradient, Deconv2D FORWARD): the stride-2 transposed conv is split into its 4 output parity
//          classes; each is a dense stride-1 correlation (DI=-1) with a 3x3 / 3x2 / 2x3 / 2x2 sub-filter, so no MFMA
//          multiplies the structural zeros of a zero-insertion formulation.  One workgroup computes a PAIR of classes
//          {3x3 + 2x2} or {3x2 + 2x3} (13 / 12 taps: balanced) from ONE staged gy slab.
//
// Implicit GEMM with v_mfma_f32_32x32x2_f32 (exact fp32).  Per workgroup (4 waves, one per SIMD):
//   * LDS slab = raw input patch of the pixel tile for CK reduction channels, zero halo (= TF SAME padding) so the
//     MFMA loop has no bounds checks: fragment address = per-lane base + wave-uniform tap offset;
//   * LDS filter slice [tap][ck][cn]; the FILTER is the MFMA A operand (rows -> accumulator registers), pixels are the
//     B operand (cols -> lanes), so every accumulator store is a coalesced run of NCHW floats;
//   * staging = raw buffer loads (hardware bounds check returns 0 for the halo: no branches, no selects) issued into
//     registers BEFORE the MFMA block of the current chunk and committed to LDS after it (async-STAGE split);
//   * wave layout WM x WN x KS: KS > 1 splits the reduction channels of each chunk across waves and combines the
//     accumulators through LDS at the end.  The problems here are small GEMMs (1024..16384 pixels x 32..256 channels);
//     picking (2,2,1) / (2,1,2) / (1,1,4) per layer yields >= 256 workgroups (one per CU) WITHOUT a global split-K pass.
//     A deterministic global split-K (partial slabs + reduce) remains as a fallback for even smaller problems.
#include "common.h"
#include "conv.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <vector>
using namespace ggan;

namespace {

constexpr int XE_MAX = 24;                 // slab elements staged per thread per chunk (x256 threads: 6144 budget)
constexpr unsigned OOB = 0x7FFFFFF0u;      // voffset >= num_records: the buffer load returns 0
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// LDS floats of the forward kinds' staging regions: the data plus the zero fill of the last DMA wave-instruction's unused lanes
// (exact: the 64x32 layout with 16-byte slab units then needs 2 x 40 KB = exactly half of a CU's LDS), at least 4 floats (the
// register-staged path parks out-of-tile elements there)
__host__ __device__ inline int fwd_region(int data_floats, int lanes_used, int floats_per_lane) {
    const int tail = (((lanes_used + 63) & ~63) - lanes_used) * floats_per_lane;
    return (data_floats + (tail > 4 ? tail : 4) + 3) & ~3;
}

struct CorrClass {
    int Hu, Wv;        // pixel grid of this class
    int roff, coff;    // slab row/col of tap (0,0) for pixel (0,0)
    int or0, oc0;      // output row = or0 + ors*u
    int wbase;         // filter offset of tap (0,0) (floats)
    int pad_;
};

struct CorrParams {
    const float* in;
    const float* w;
    const float* bias;
    float* out;
    int N, CKtot, Hin, Win;
    int CNtot, Hout, Wout;
    int ors, ocs;
    int w_si, w_sj, w_sk, w_sn;
    int TR, TC, TI;
    int SR, SCp, CS;
    int row0, col0;
    FastDiv24 d_CS, d_SRSC, d_SCp, d_TRTC, d_TC;   // numerators <= XE_MAX * 256
    int img_groups, tiles_r, tiles_c;
    int cps, SK;
    int act;
    float alpha;
    unsigned in_bytes, w_bytes;
    int dbg;
    int dma;                      // forward kinds: stage the slab and the filter slice by LDS-DMA (buffer_load ... lds)
    int xq;                       // 4: the slab is staged in 16-byte units of image rows (forward DMA path, see plan_and_launch), else 1
    int plan_wgs;                 // ggan_conv_geom.plan_wgs of the call (0: default)
    // XCD-aware tile order (round 5): workgroup id % 8 is the XCD it runs on and every XCD has its own L2.  The eight XCDs are dealt out as
    // xcd_p pixel-tile groups x (8 / xcd_p) channel-tile groups, so that an XCD's L2 fetches 1 / xcd_p of the input tensor and
    // xcd_p / 8 of the filter instead of (with the plain order: pixel tiles fastest) 1/8 of the input and ALL of the filter -- 26 of the
    // 29 MB the 128->256 layer fetched for a 6.4 MB problem.  0: plain order.
    int xcd_p;
    int nstg;                     // forward DMA path: LDS staging buffers (3: ring fetched two chunks ahead, round 4; 2: the round-2 scheme)
    const unsigned* xtab;         // forward DMA path: plan-time slab offsets [tile position][XE][NTHR] (relative to the tile's first image), or NULL
    unsigned long long* stamps;   // debug: per-workgroup s_memtime stamps (GGAN_DBG & 4)
    const float* out_ref;    // optional (forward kind, SK == 1): the stored value is act_grad(v, out_ref[same index]) -- the double backward of
    int out_act;             // a masked data gradient (functional.ConvDgradMasked) without an act_bwd launch behind the conv
    float out_alpha;
    const float* in_ref;     // optional: slab values are in[i] * act'(in_ref[i]) (fused activation backward)
    int in_act;
    float in_alpha;
    size_t out_elems;
    CorrClass cls[4];
};

// Compile-time ablations for timing experiments (tools/variant_lib.sh builds a second libggan.so with -DGGAN_ABL=bits and
// tools/stamps.py prints the per-workgroup phase times): 1 = no staging inside the chunk loop, 4 = MFMAs on register constants
// instead of LDS fragments.  0 in the product build: the branches fold away.
#ifndef GGAN_ABL
#define GGAN_ABL 0
#endif
// One chunk's MFMAs for one class.  hipcc emits "ds_read; s_waitcnt lgkmcnt(0); v_mfma; v_mfma" for the straightforward loop
// (operands fetched right before use into the same four registers), which leaves the matrix pipe idle for one LDS latency
// per MFMA pair (measured: 5300 cycles per chunk where the MFMAs need 3200).  Here the operands of MFMA pair s+2,s+3 are
// read BEFORE MFMA pair s,s+1 is issued, and sched_group_barrier pins that interleave (<= 4 LDS reads in flight).
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

// `hook(step)` runs after MFMA pair number `step` of the chunk (STEP0 = pairs of the classes before this one): the staging
// instructions of the next chunk are dealt out between the pairs, one or two per pair.  Eight waves issuing their whole share
// of a chunk at once back up the CU's vector-memory path, and every wave then sits in the issue queue instead of multiplying
// (measured on the 64->128 data gradient: 5640 cycles per chunk with the burst, 4750 with the filter slice dealt out).
template <int TH, int TW, int DI, int PW, int CK, int RS, int STEP0 = 0, class Hook = NoHook>
__device__ __forceinline__ void mma_taps(const float* __restrict__ xs, const float* __restrict__ ws, int xfrag, int wfrag,
                                         int CS, int SCp, f32x16& acc, Hook&& hook = Hook()) {
    constexpr int NS = TH * TW * PW;
    float a[NS], b[NS];
    auto load = [&](int s) {
        const int tap = s / PW, p = s - tap * PW;
        const int i = tap / TW, j = tap - i * TW;
        a[s] = ws[(tap * CK + p * 2) * RS + wfrag];
        b[s] = xs[p * 2 * CS + xfrag + DI * (i * SCp + j)];
    };
    load(0);
    if (NS > 1) load(1);
#pragma unroll
    for (int s = 0; s < NS; s += 2) {
        if (s + 2 < NS) load(s + 2);
        if (s + 3 < NS) load(s + 3);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        if (s + 1 < NS) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s + 1], b[s + 1], acc, 0, 0, 0);
        hook(STEP0 + s / 2);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // DS reads of the NEXT pair first
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // then this pair's two MFMAs
        __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);   // then the staging instructions dealt to this pair
    }
}

// Same product out of the DMA-staged filter image of the data-gradient kinds (corr_body, WD): per tap and block of 16 output
// channels the image holds [k quad][channel][4 k] floats exactly as an LDS-DMA wave-instruction of 16-byte lanes lays them down
// (258-float block stride: the two channel blocks a wave reads sit 2 banks apart, so the 4-float channel pitch is conflict free).
// PW == 1: one float per MFMA (k = 2*ks + half).  PW == 2: a lane's two k of a tap are adjacent -> one 8-byte read feeds both
// MFMAs (k = 4*ks + 2*half + p; the slab rows are addressed to match).
template <int TH, int TW, int DI, int PW, int TS, int STEP0, class Hook>
__device__ __forceinline__ void mma_taps_wd(const float* __restrict__ xs, const float* __restrict__ ws, int xfrag, int wfrag,
                                            int CS, int SCp, f32x16& acc, Hook&& hook) {
    constexpr int NT = TH * TW;
    if constexpr (PW == 1) {
        float a[NT], b[NT];
        auto load = [&](int s) {
            const int i = s / TW, j = s - i * TW;
            if (GGAN_ABL & 4) { a[s] = __int_as_float(wfrag + s); b[s] = __int_as_float(xfrag + s); return; }
            a[s] = ws[s * TS + wfrag];
            b[s] = xs[xfrag + DI * (i * SCp + j)];
        };
        load(0);
        if (NT > 1) load(1);
#pragma unroll
        for (int s = 0; s < NT; s += 2) {
            if (s + 2 < NT) load(s + 2);
            if (s + 3 < NT) load(s + 3);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
            if (s + 1 < NT) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s + 1], b[s + 1], acc, 0, 0, 0);
            hook(STEP0 + s / 2);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
        }
    } else {
        static_assert(PW == 2, "one 8-byte filter read per tap");
        float2 a[NT];
        float b0[NT], b1[NT];
        auto load = [&](int t) {
            const int i = t / TW, j = t - i * TW;
            a[t] = *reinterpret_cast<const float2*>(ws + t * TS + wfrag);
            b0[t] = xs[xfrag + DI * (i * SCp + j)];
            b1[t] = xs[CS + xfrag + DI * (i * SCp + j)];
        };
        load(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) load(t + 1);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b0[t], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b1[t], acc, 0, 0, 0);
            hook(STEP0 + t);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
        }
    }
}

__device__ __forceinline__ void wait_vm_n(int n) {      // s_waitcnt vmcnt(n) for a wave-uniform run-time n
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
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// Class lists.  KIND 0: forward, one 5x5 class.  KIND 1 / 2: data-gradient class pairs {0,3} / {1,2} (13 / 12 taps).
// KIND 3: all four parity classes in one workgroup (25 taps): each lane then owns the 2x2 output block of its class pixel
// and the epilogue writes FULL output rows as float2 (the pair kinds leave every 128-B line half-written by two different
// workgroups -- measured 29 % of the workgroup's lifetime in the store tail).
template <int KIND> struct ClassList;
template <> struct ClassList<0> { static constexpr int NC = 1; static constexpr int cls(int) { return 0; }
                                  static constexpr int th(int) { return 5; } static constexpr int tw(int) { return 5; } };
template <> struct ClassList<1> { static constexpr int NC = 2; static constexpr int cls(int i) { return i == 0 ? 0 : 3; }
                                  static constexpr int th(int i) { return i == 0 ? 3 : 2; } static constexpr int tw(int i) { return i == 0 ? 3 : 2; } };
template <> struct ClassList<2> { static constexpr int NC = 2; static constexpr int cls(int i) { return i == 0 ? 1 : 2; }
                                  static constexpr int th(int i) { return i == 0 ? 3 : 2; } static constexpr int tw(int i) { return i == 0 ? 2 : 3; } };
template <> struct ClassList<3> { static constexpr int NC = 4; static constexpr int cls(int i) { return i; }
                                  static constexpr int th(int i) { return i < 2 ? 3 : 2; } static constexpr int tw(int i) { return (i & 1) ? 2 : 3; } };

template <int KIND, int SU, int DI, int WM, int WN, int KS, int PW, bool X4 = false>
__device__ __forceinline__ void corr_body(const CorrParams& P, const int split, float* smem, const int bx, const int by) {
    using CL = ClassList<KIND>;
    constexpr int NC = CL::NC;
    constexpr int CK = 2 * KS * PW;
    constexpr int TNW = 32 * WN;
    // The data-gradient kinds read the filter in its own HWIO layout, where the REDUCTION channel (co) is the contiguous
    // one: a staging unit is then 4 consecutive k of one output channel, scattered into 4 LDS rows; rows are padded by 2
    // floats so that scatter (lanes = 8 channels x 8 k-quads) and the fragment reads are both bank-conflict free.
    constexpr bool WK = KIND != 0;
    constexpr int RS = TNW + (WK ? 2 : 0);
    constexpr int NT0 = CL::th(0) * CL::tw(0);
    constexpr int NT1 = NC > 1 ? CL::th(1) * CL::tw(1) : 0;
    constexpr int NT2 = NC > 2 ? CL::th(2) * CL::tw(2) : 0;
    constexpr int NT3 = NC > 3 ? CL::th(3) * CL::tw(3) : 0;
    constexpr int NTT = NT0 + NT1 + NT2 + NT3;
    constexpr int WUNITS = NTT * CK * (TNW / 4);
    constexpr int NTHR = 64 * WM * WN * KS;
    // WD: the data-gradient kinds with 16-channel chunks stage the filter slice by LDS-DMA (see mma_taps_wd); P.dma selects it
    constexpr bool WD = KIND != 0 && CK % 16 == 0 && (PW == 1 || PW == 2);
    constexpr int GH = CK / 16 > 0 ? CK / 16 : 1;      // 16-channel groups of a chunk: a DMA block is (tap, group, 16 output channels)
    constexpr int NCB = TNW / 16, WBLK = 258, NBLK = NTT * GH * NCB, NWAVE = NTHR / 64;
    constexpr int WQ = (NBLK + NWAVE - 1) / NWAVE;
    constexpr int WE = (WUNITS + NTHR - 1) / NTHR;
    constexpr int XE = (KIND == 0 ? XE_MAX : XE_MAX / 2) * 256 / NTHR;   // per-thread slab elements (4096 / 2048 budget)
    static_assert(NTHR == 256 || NTHR == 512, "4 or 8 waves per workgroup");
    static_assert(CK % 4 == 0, "chunk must hold whole float4 groups");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = (wave / WM) % WN, ks = wave / (WM * WN);
    const bool stamping = (P.dbg & 4) && P.stamps && tid == ((P.dbg & 64) ? NTHR / 2 : 0);   // dbg 64: a second-half wave
    const bool fine = (P.dbg & 128) != 0;   // dbg 128: stamps 4.. = phase boundaries inside the first chunks instead of chunk ends
    const int wg_lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    auto stamp = [&](int i) { if (stamping) P.stamps[(size_t)wg_lin * 16 + i] = __builtin_readcyclecounter(); };
    stamp(0);
    const int half = lane >> 5, l31 = lane & 31;

    // ---- which tile ---------------------------------------------------------------------------------
    const int tiles_per_img = P.tiles_r * P.tiles_c;
    const int ig = bx / tiles_per_img;
    const int tt = bx - ig * tiles_per_img;
    const int tr = tt / P.tiles_c, tc = tt - tr * P.tiles_c;
    const int n0 = ig * P.TI, u0 = tr * P.TR, v0 = tc * P.TC;
    const int cn0 = by * TNW;
    const int ck_begin = split * P.cps;
    const int ck_end = min(ck_begin + P.cps, P.CKtot);
    const int in_row0 = SU * u0 + P.row0, in_col0 = SU * v0 + P.col0;
    const int HWin = P.Hin * P.Win;

    // two staging buffers: [CK][CS] slab + [NTT][CK][TNW] filter slice each
    // (+1 / +4 rows: trash slots that absorb the commit of staging elements beyond the tile, so commits are branch-free)
    // (forward kinds: the slab region is padded to whole 64-float wave-instructions and the filter slice gets a 256-float tail,
    //  so the zero fill of an LDS-DMA instruction's unused lanes lands in padding)
    constexpr int xq = X4 ? 4 : 1;       // (compile-time: a run-time choice of the unit size would put branches between the MFMAs)
    // data-gradient kinds with 16-byte units: the slab goes through registers as whole units (one b128 load + one b128 LDS write
    // for what took four dword loads and writes); XN units per thread cover the slab with its rows padded to unit multiples
    constexpr bool XR4 = KIND != 0 && X4;
    constexpr int XN = XR4 ? (XE + 3) / 4 + 1 : XE;
    const int XS_SZ = KIND == 0 ? fwd_region(CK * P.CS, CK * P.CS / xq, xq) : ((CK * P.CS + 1 + 3) & ~3);
    constexpr int WS_USED = NTT * CK * RS;
    const int STAGE = XS_SZ + (KIND == 0 ? fwd_region(WS_USED, WUNITS, 4) : ((WS_USED + 4 * RS + 3) & ~3));

    const auto rin = __builtin_amdgcn_make_buffer_rsrc((void*)P.in, (short)0, (int)P.in_bytes, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)P.w, (short)0, (int)P.w_bytes, 0x00020000);
    constexpr bool CAN_MASK = KIND != 0;              // only the data-gradient kinds consume gy * act'(y)
    const bool masked = CAN_MASK && P.in_ref != nullptr;
    const float mslope = P.in_act == GGAN_ACT_LRELU ? P.in_alpha : 0.f;   // lrelu / relu only (launcher checks)
    const auto rref = __builtin_amdgcn_make_buffer_rsrc((void*)(masked ? P.in_ref : P.in), (short)0, (int)P.in_bytes, 0x00020000);

    const bool dma = KIND == 0 && P.dma != 0;        // forward kinds: slab and filter slice by LDS-DMA
    const bool wdma = WD && P.dma != 0;             // data-gradient kinds: filter slice by LDS-DMA, slab through registers
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    unsigned xreg[XR4 ? 1 : XE], xref[XR4 ? 1 : XE];
    u32x4 xreg4[XR4 ? XN : 1], xref4[XR4 ? XN : 1];
    if constexpr (XR4) {
#pragma unroll
        for (int j = 0; j < XN; ++j) xref4[j] = u32x4{0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};
    } else {
#pragma unroll
        for (int j = 0; j < XE; ++j) xref[j] = 0x3f800000u;      // 1.0f: "positive" reference = identity mask when unmasked
    }

    // ---- per-thread staging descriptors (fixed across chunks) ---------------------------------------------
    // (the forward DMA path issues the first chunk's loads as soon as each descriptor exists: part of their latency runs under
    //  the rest of the descriptor arithmetic; measured neutral-to-worse for the data-gradient kinds, which keep them together)
    unsigned xvo[XN];
    // (xq == 4: the "elements" are 16-byte units -- 4 consecutive floats of an image row, wholly inside or outside the image --
    //  and the divisors in P are those of the unit grid)
    const int CSu = P.CS / xq, SCpu = P.SCp / xq;
    const int xe_cnt = CK * CSu, srsc = P.SR * SCpu;
    // Round 4: the forward DMA path reads its slab offsets from a table built ONCE per geometry on the host (they depend on the tile
    // position and the geometry only): one coalesced load per staging instruction instead of ~24 instructions of index arithmetic
    // each (4300 of the 5400 cycles this kernel spent before its first barrier), and the filter slice -- whose offsets are cheap --
    // goes out while the table is on its way.
    const bool tabled = KIND == 0 && dma && P.xtab != nullptr;
    if (tabled) {
        const unsigned* tab = P.xtab + (size_t)tt * (XN * NTHR) + tid;
#pragma unroll
        for (int j = 0; j < XN; ++j) xvo[j] = (wave_u * 64 + j * NTHR < xe_cnt) ? tab[j * NTHR] : OOB;
    } else {
#pragma unroll
    for (int j = 0; j < XN; ++j) {
        const int e = tid + j * NTHR;
        unsigned off = OOB;
        if (e < xe_cnt) {
            // (every factor below is < 2^24 and every product < 2^32: full-rate 24-bit multiplies)
            const int ckl = fdiv24(e, P.d_CS);
            const int r1 = e - __umul24(ckl, CSu);
            const int img = fdiv24(r1, P.d_SRSC);
            const int r2 = r1 - __umul24(img, srsc);
            const int r = fdiv24(r2, P.d_SCp);
            const int cc = (r2 - __umul24(r, SCpu)) * xq;
            const int ih = in_row0 + r, iw = in_col0 + cc, n = n0 + img;
            if (n < P.N && ih >= 0 && ih < P.Hin && iw >= 0 && iw < P.Win)
                off = (__umul24(__umul24(__umul24(n, P.CKtot) + ckl, P.Hin) + ih, P.Win) + iw) * 4u;
        }
        xvo[j] = off;
        if (dma) {
            const int e0 = wave_u * 64 + j * NTHR;
            if (e0 < xe_cnt) {
                float* dst = smem + e0 * xq;
                const int so = ck_begin * HWin * 4;
                if (xq == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)dst, 16, off, so, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)dst, 4, off, so, 0, 0);
            }
        }
    }
    }
    stamp(2);
    // (filter base of each class of the list, fetched with STATIC kernel-argument offsets: indexing P.cls by a run-time class
    //  number turns into one dependent scalar / vector load + wait per use -- measured 6.5 k cycles of prologue)
    const int cls_wb0 = P.cls[CL::cls(0)].wbase, cls_wb1 = P.cls[CL::cls(NC > 1 ? 1 : 0)].wbase;
    const int cls_wb2 = P.cls[CL::cls(NC > 2 ? 2 : 0)].wbase, cls_wb3 = P.cls[CL::cls(NC > 3 ? 3 : 0)].wbase;
    auto wbase_of = [&](int ci_) {      // (separate scalars, not an array: a run-time index would send it to scratch memory)
        return ci_ == 0 ? cls_wb0 : ci_ == 1 ? cls_wb1 : ci_ == 2 ? cls_wb2 : cls_wb3;
    };
    auto tw_of = [](int ci_) { return ci_ == 0 ? CL::tw(0) : ci_ == 1 ? CL::tw(1) : ci_ == 2 ? CL::tw(2) : CL::tw(3); };
    unsigned wvo[WE];
    // unit -> (tap group, first reduction channel, output channel) and its LDS float index
    //   !WK: unit = (tap, ck, cn4): 4 consecutive cn in memory => every wave-load is a run of whole 128-B filter rows
    //    WK: unit = (tap, cn, ck4): 4 consecutive ck in memory; 8 lanes cover one 128-B run of a channel's filter row
    auto w_tap = [](int u) { return u / (CK * (TNW / 4)); };
    auto w_ck = [](int u) { return WK ? (u % (CK / 4)) * 4 : (u / (TNW / 4)) % CK; };
    auto w_cn = [](int u) { return WK ? (u / (CK / 4)) % TNW : (u % (TNW / 4)) * 4; };
    auto w_lds = [&](int u) { return (w_tap(u) * CK + w_ck(u)) * RS + w_cn(u); };
    if (!(WD && P.dma)) {
#pragma unroll
    for (int q = 0; q < WE; ++q) {
        const int u = tid + q * NTHR;
        unsigned off = OOB;
        if (u < WUNITS) {
            const int cn = w_cn(u);
            const int ckl = w_ck(u);
            const int tapg = w_tap(u);
            int ci_ = 0, tap = tapg;                     // which class of the list, tap index inside it
            if (NC > 1 && tap >= NT0) { tap -= NT0; ci_ = 1; }
            if (NC > 2 && ci_ == 1 && tap >= NT1) { tap -= NT1; ci_ = 2; }
            if (NC > 3 && ci_ == 2 && tap >= NT2) { tap -= NT2; ci_ = 3; }
            const int i = tw_of(ci_) == 3 ? tap / 3 : (tw_of(ci_) == 2 ? tap / 2 : tap / 5), j = tap - i * tw_of(ci_);
            const int wb = wbase_of(ci_);
            if (cn0 + cn < P.CNtot)
                off = (unsigned)(wb + i * P.w_si + j * P.w_sj + ckl * P.w_sk + (cn0 + cn) * P.w_sn) * 4u;
        }
        wvo[q] = off;
        if (dma) {
            const int u0 = wave_u * 64 + q * NTHR;
            const unsigned vo = (ck_begin + w_ck(u) < ck_end) ? off : OOB;
            if (u0 < WUNITS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(smem + XS_SZ + 4 * u0), 16, vo, ck_begin * P.w_sk * 4, 0, 0);
        }
    }
    }

    if (tabled) {
        // (the image part joins the per-lane offset, so that images beyond N fall out of the buffer's range; the chunk part is scalar)
        const unsigned img_base = (unsigned)n0 * (unsigned)P.CKtot * (unsigned)HWin * 4u;
#pragma unroll
        for (int j = 0; j < XN; ++j) {
            const unsigned t = xvo[j];
            xvo[j] = t >= OOB ? OOB : t + img_base;
            const int e0 = wave_u * 64 + j * NTHR;
            if (e0 < xe_cnt) {
                float* dst = smem + e0 * xq;
                const int so = ck_begin * HWin * 4;
                if (xq == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)dst, 16, xvo[j], so, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)dst, 4, xvo[j], so, 0, 0);
            }
        }
    }
    stamp(15);
    // ---- per-lane MFMA fragment bases ----------------------------------------------------------------------
    int xfrag[NC];
    {
        const int p = wm * 32 + l31;
        const int img = fdiv24(p, P.d_TRTC);
        const int rem = p - img * (P.TR * P.TC);
        const int ur = fdiv24(rem, P.d_TC);
        const int vc = rem - ur * P.TC;
        const bool in_tile = img < P.TI && (n0 + img) < P.N;
        const int kbase = (WD && PW == 2 && P.dma) ? (ks * 4 + 2 * half) * P.CS : (ks * PW * 2 + half) * P.CS;
        const int b = in_tile ? img * (P.SR * P.SCp) + SU * ur * P.SCp + SU * vc : 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const CorrClass& cc = P.cls[CL::cls(c)];
            xfrag[c] = kbase + b + cc.roff * P.SCp + cc.coff;
        }
    }
    const int wfrag = (ks * PW * 2 + half) * RS + wn * 32 + l31;
    // (WD image: block of 16 channels, k quad, channel, element)
    const int wd_quad = PW == 2 ? ks : (ks >> 1);                 // this wave's k quad of the chunk
    const int wfrag_wd = ((wd_quad >> 2) * NCB + wn * 2 + (l31 >> 4)) * WBLK + ((wd_quad & 3) * 16 + (l31 & 15)) * 4 +
                         (PW == 2 ? 2 * half : (ks & 1) * 2 + half);

    f32x16 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    u32x4 wreg[WE];

    auto prefetch = [&](int ck0, bool with_w = true) {
        const int soff_x = ck0 * HWin * 4;
        const int soff_w = ck0 * P.w_sk * 4;
        if constexpr (XR4) {
#pragma unroll
            for (int j = 0; j < XN; ++j) xreg4[j] = __builtin_amdgcn_raw_buffer_load_b128(rin, xvo[j], soff_x, 0);
            if (masked) {
#pragma unroll
                for (int j = 0; j < XN; ++j) xref4[j] = __builtin_amdgcn_raw_buffer_load_b128(rref, xvo[j], soff_x, 0);
            }
        } else {
        if (!(P.dbg & 512)) {
#pragma unroll
        for (int j = 0; j < XE; ++j) xreg[j] = __builtin_amdgcn_raw_buffer_load_b32(rin, xvo[j], soff_x, 0);
        }
        if (masked) {
#pragma unroll
            for (int j = 0; j < XE; ++j) xref[j] = __builtin_amdgcn_raw_buffer_load_b32(rref, xvo[j], soff_x, 0);
        }
        }
        if (with_w && !(P.dbg & 256)) {
#pragma unroll
        for (int q = 0; q < WE; ++q) {
            const unsigned vo = (ck0 + w_ck(tid + q * NTHR) < ck_end) ? wvo[q] : OOB;     // reduction-channel tail -> zero filter rows
            wreg[q] = __builtin_amdgcn_raw_buffer_load_b128(rw, vo, soff_w, 0);
        }
        }
    };

    auto commit = [&](int buf, bool with_w = true) {
        float* xsb = smem + buf * STAGE;
        float* wsb = xsb + XS_SZ;
        // straight-line: elements beyond the tile (their loads returned 0) land in a trash slot instead of being predicated
        if constexpr (XR4) {
#pragma unroll
            for (int j = 0; j < XN; ++j) {
                const int e = tid + j * NTHR;
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float a = __uint_as_float(xreg4[j][c]);
                    v[c] = __uint_as_float(xref4[j][c]) > 0.f ? a : a * mslope;
                }
                *reinterpret_cast<f32x4*>(xsb + 4 * min(e, xe_cnt)) = v;      // (unit e = floats [4e, 4e+4): rows are whole units)
            }
        } else {
#pragma unroll
        for (int j = 0; j < XE; ++j) {
            const int e = tid + j * NTHR;
            float v = __uint_as_float(xreg[j]);
            if (CAN_MASK) v = __uint_as_float(xref[j]) > 0.f ? v : v * mslope;
            xsb[min(e, xe_cnt)] = v;
        }
        }
        if (with_w)
#pragma unroll
        for (int q = 0; q < WE; ++q) {
            const int u = tid + q * NTHR;
            const int li = (q < WE - 1 || u < WUNITS) ? w_lds(u) : WS_USED;
            if (WK) {
                unsigned* d = reinterpret_cast<unsigned*>(wsb + li);
                d[0] = wreg[q][0]; d[RS] = wreg[q][1]; d[2 * RS] = wreg[q][2]; d[3 * RS] = wreg[q][3];
            } else {
                *reinterpret_cast<u32x4*>(wsb + li) = wreg[q];
            }
        }
    };

    // LDS-DMA staging (forward kinds): the same per-lane offsets, but the loads land in LDS directly -- lane l of a
    // wave-instruction writes LDS[M0 base + l * size], which is exactly how the slab (element e = tid + j*NTHR) and the filter slice
    // (float4 unit u = tid + q*NTHR, LDS float index 4u) are laid out; halo / tail lanes carry the out-of-bounds offset and write
    // zeros.  No staging registers, no ds_write traffic competing with the fragment reads of the MFMA loop.

    // WD: this wave's filter blocks (tap, 16 output channels) of a chunk -- lane = (k quad, channel) reads 16 bytes = 4 consecutive
    // reduction channels of its output channel; the block part of the address is wave-uniform (SGPR offset)
    int wso[WQ];
    unsigned wd_lane = 0;
    int wd_cnlim = 0;
    if (WD) {
        wd_lane = (unsigned)((lane >> 4) * 4 * P.w_sk + (cn0 + (lane & 15)) * P.w_sn) * 4u;
        wd_cnlim = P.CNtot - cn0 - (lane & 15);
        // (the block offsets are computed ONCE, lane b for block b, and each wave picks its own by v_readlane: evaluated per
        //  wave-instruction in a scalar loop the divisions and selects took 3150 cycles of the prologue -- tools/stamps.py)
        constexpr int BP = (NBLK + 63) / 64;
        static_assert(BP <= 2, "block offsets: at most two lane passes");
        int blk[BP];
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            const int b = lane + 64 * p;
            const int cnb = b % NCB, gh = (b / NCB) % GH;
            int tap = b / (NCB * GH), ci_ = 0;
            if (NC > 1 && tap >= NT0) { tap -= NT0; ci_ = 1; }
            if (NC > 2 && ci_ == 1 && tap >= NT1) { tap -= NT1; ci_ = 2; }
            if (NC > 3 && ci_ == 2 && tap >= NT2) { tap -= NT2; ci_ = 3; }
            const int i = tw_of(ci_) == 3 ? tap / 3 : (tw_of(ci_) == 2 ? tap / 2 : tap / 5), j = tap - i * tw_of(ci_);
            const int wb = wbase_of(ci_);
            blk[p] = b < NBLK ? (wb + i * P.w_si + j * P.w_sj + gh * 16 * P.w_sk + cnb * 16 * P.w_sn) * 4 : -1;
        }
#pragma unroll
        for (int q = 0; q < WQ; ++q) {
            const int b = wave_u + q * NWAVE;
            const int lo = __builtin_amdgcn_readlane(blk[0], b & 63);
            const int hi = BP > 1 ? __builtin_amdgcn_readlane(blk[BP - 1], b & 63) : -1;
            wso[q] = b < 64 ? lo : (b < 64 * BP ? hi : -1);
        }
    }
    auto stage_w1 = [&](int q, int ck0, int buf) {                 // one wave-instruction of the filter slice
        float* wsb = smem + buf * STAGE + XS_SZ;
        const int b = wave_u + q * NWAVE;
        const bool k_ok = ck0 + ((b / NCB) % GH) * 16 + (lane >> 4) * 4 < ck_end;            // reduction-channel tail -> zero filter rows
        const unsigned vo = (k_ok && (b % NCB) * 16 < wd_cnlim) ? wd_lane : OOB;
        const int so = wso[q] + ck0 * P.w_sk * 4;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(wsb + b * WBLK), 16, vo, so, 0, 0);
    };
    auto stage_w = [&](int ck0, int buf) {
#pragma unroll
        for (int q = 0; q < WQ; ++q)
            if (wso[q] >= 0) stage_w1(q, ck0, buf);
    };

    auto dma_x1 = [&](int j, int ck0, int buf) {                   // forward kinds: one wave-instruction of the slab
        float* xsb = smem + buf * STAGE;
        const int e0 = wave_u * 64 + j * NTHR;
        if (e0 < xe_cnt) {
            float* dst = xsb + e0 * xq;
            const int so = ck0 * HWin * 4;
            const unsigned vo = xvo[j];
            if (xq == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)dst, 16, vo, so, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)dst, 4, vo, so, 0, 0);
        }
    };
    auto dma_w1 = [&](int q, int ck0, int buf) {                   // ... of the filter slice
        float* wsb = smem + buf * STAGE + XS_SZ;
        const int u0 = wave_u * 64 + q * NTHR;
        const unsigned vo = (ck0 + w_ck(tid + q * NTHR) < ck_end) ? wvo[q] : OOB;
        if (u0 < WUNITS)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(wsb + 4 * u0), 16, vo, ck0 * P.w_sk * 4, 0, 0);
    };

    // ---- main loop over reduction-channel chunks: chunk c is multiplied out of buffer c&1 while chunk c+1 is committed
    //      to the other buffer and chunk c+2's global loads are in flight; ONE barrier per chunk ---------------------
    stamp(1);
    int buf = 0;
    int it_ = 0;
    if (dma && P.nstg == 3) {
        // Round 4: a ring of THREE staging buffers, fetched TWO chunks ahead.  The wait at the end of chunk c retires the loads of chunk
        // c+1, issued during chunk c-1 -- they landed long ago, so the end of a chunk costs the skew of the waves at the barrier and
        // not a memory round trip (with two buffers the loads retired there had been issued in the same chunk, the last of them two
        // MFMA pairs earlier).  WAR: the buffer chunk c+2 lands in was last read in chunk c-1, and every wave is past that barrier.
        int ndma = 0;                                   // staging instructions of this wave per chunk
#pragma unroll
        for (int j = 0; j < XE; ++j) ndma += (wave_u * 64 + j * NTHR < xe_cnt) ? 1 : 0;
#pragma unroll
        for (int q = 0; q < WE; ++q) ndma += (wave_u * 64 + q * NTHR < WUNITS) ? 1 : 0;
        if (ck_begin + CK < ck_end) {
#pragma unroll
            for (int q = 0; q < WE; ++q) dma_w1(q, ck_begin + CK, 1);
#pragma unroll
            for (int j = 0; j < XE; ++j) dma_x1(j, ck_begin + CK, 1);
            wait_vm_n(ndma);                            // (chunk 0 was issued with the descriptors)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        stamp(3);
        constexpr int NSTEP = (NT0 * PW + 1) / 2, NITEM = XE + WE;
        constexpr int IPS = (NITEM + NSTEP - 3) / (NSTEP - 2);
        for (int ck0 = ck_begin; ck0 < ck_end; ck0 += CK, buf = buf == 2 ? 0 : buf + 1) {
            const int ck2 = ck0 + 2 * CK;
            const bool more2 = ck2 < ck_end;
            const int b2 = buf == 0 ? 2 : buf - 1;      // (buf + 2) % 3
            auto hook = [&](int g) {
#pragma unroll
                for (int it = g * IPS; it < (g + 1) * IPS; ++it) {
                    if (more2 && it < XE) dma_x1(it, ck2, b2);
                    if (more2 && it >= XE && it < NITEM) dma_w1(it - XE, ck2, b2);
                }
            };
            const float* xs = smem + buf * STAGE;
            const float* ws = xs + XS_SZ;
            mma_taps<CL::th(0), CL::tw(0), DI, PW, CK, RS, 0>(xs, ws, xfrag[0], wfrag, P.CS, P.SCp, acc[0], hook);
            if (more2) wait_vm_n(ndma);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (it_ < 8) stamp(4 + it_);
            ++it_;
        }
    } else if (dma) {
        // chunk c+1 is in flight into the other buffer while chunk c is multiplied: the wait + barrier at the end of the chunk
        // retires it for every wave (RAW), and every wave's fragment reads of that buffer were retired one barrier earlier (WAR)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (chunk 0 was issued with the descriptors)
        __syncthreads();
        stamp(3);
        constexpr int NSTEP = (NT0 * PW + 1) / 2, NITEM = XE + WE;
        constexpr int IPS = (NITEM + NSTEP - 3) / (NSTEP - 2);      // items per MFMA pair: all issued two pairs before the chunk ends
        for (int ck0 = ck_begin; ck0 < ck_end; ck0 += CK, buf ^= 1) {
            const int ckn = ck0 + CK;
            const bool more = ckn < ck_end;
            auto hook = [&](int g) {
#pragma unroll
                for (int it = g * IPS; it < (g + 1) * IPS; ++it) {
                    if (more && it < XE) dma_x1(it, ckn, buf ^ 1);
                    if (more && it >= XE && it < NITEM) dma_w1(it - XE, ckn, buf ^ 1);
                }
            };
            const float* xs = smem + buf * STAGE;
            const float* ws = xs + XS_SZ;
            mma_taps<CL::th(0), CL::tw(0), DI, PW, CK, RS, 0>(xs, ws, xfrag[0], wfrag, P.CS, P.SCp, acc[0], hook);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (it_ < 8) stamp(4 + it_);
            ++it_;
        }
    } else if (wdma) {
        // the slab still goes through registers (it may carry the fused activation mask); the filter slice of chunk c+1 is in flight
        // into the other buffer while chunk c is multiplied, retired by the wait + barrier at the end of the chunk
        constexpr int TS = GH * NCB * WBLK;
        constexpr int WD_STEPS = (NT0 * (PW == 2 ? 2 : 1) + 1) / 2 + (NT1 * (PW == 2 ? 2 : 1) + 1) / 2 +
                                 (NT2 * (PW == 2 ? 2 : 1) + 1) / 2 + (NT3 * (PW == 2 ? 2 : 1) + 1) / 2;
        // (a filter block is 1 KB: eight waves x one block per MFMA pair is what the CU's 64 B/clk vector-memory path moves in the
        //  128 cycles of a pair -- more per pair and the waves queue up at the issue port again)
        constexpr int WD_DPS = (WQ + WD_STEPS - 2) / (WD_STEPS - 1);                      // filter blocks per pair
        constexpr int WD_S0 = (WQ + WD_DPS - 1) / WD_DPS < WD_STEPS - 1 ? (WQ + WD_DPS - 1) / WD_DPS : WD_STEPS - 1;   // first slab pair
        constexpr int WD_SPS = (XN + WD_STEPS - WD_S0 - 1) / (WD_STEPS - WD_S0);          // slab loads per pair
        prefetch(ck_begin, false);
        stage_w(ck_begin, 0);
        commit(0, false);
        if (ck_begin + CK < ck_end) prefetch(ck_begin + CK, false);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp(3);
        for (int ck0 = ck_begin; ck0 < ck_end; ck0 += CK, buf ^= 1) {
            const int ckn = ck0 + CK;       // (beyond the last chunk every lane is out of range: zero rows into the idle buffer)
            commit(buf ^ 1, false);
            const float* xs = smem + buf * STAGE;
            const float* ws = xs + XS_SZ;
            // dealt out between the MFMA pairs: first the filter blocks of chunk c+1 (deadline: the end of this chunk), then the slab
            // loads of chunk c+2 (beyond the last chunk every lane is out of range and nothing is fetched).  Vector-memory loads
            // retire in issue order, so the wait at the end of the chunk leaves the slab loads in flight across the barrier.
            const int ck2 = ck0 + 2 * CK;
            const bool more2 = ck2 < ck_end;
            auto hook = [&](int g) {
#pragma unroll
                for (int q = g * WD_DPS; q < (g + 1) * WD_DPS; ++q)
                    if (q < WQ && (q < WQ - 1 || wso[WQ - 1] >= 0)) stage_w1(q, ckn, buf ^ 1);
#pragma unroll
                for (int j = (g - WD_S0) * WD_SPS; j < (g - WD_S0 + 1) * WD_SPS; ++j)
                    if (j >= 0 && j < XN) {
                        if constexpr (XR4) {
                            xreg4[j] = __builtin_amdgcn_raw_buffer_load_b128(rin, more2 ? xvo[j] : OOB, ck2 * HWin * 4, 0);
                            if (masked) xref4[j] = __builtin_amdgcn_raw_buffer_load_b128(rref, more2 ? xvo[j] : OOB, ck2 * HWin * 4, 0);
                        } else {
                            xreg[j] = __builtin_amdgcn_raw_buffer_load_b32(rin, more2 ? xvo[j] : OOB, ck2 * HWin * 4, 0);
                            if (masked) xref[j] = __builtin_amdgcn_raw_buffer_load_b32(rref, more2 ? xvo[j] : OOB, ck2 * HWin * 4, 0);
                        }
                    }
            };
            if constexpr (WD) {
                constexpr int S1 = (CL::th(0) * CL::tw(0) * (PW == 2 ? 2 : 1) + 1) / 2;
                constexpr int S2 = NC > 1 ? (CL::th(1) * CL::tw(1) * (PW == 2 ? 2 : 1) + 1) / 2 : 0;
                constexpr int S3 = NC > 2 ? (CL::th(2) * CL::tw(2) * (PW == 2 ? 2 : 1) + 1) / 2 : 0;
                mma_taps_wd<CL::th(0), CL::tw(0), DI, PW, TS, 0>(xs, ws, xfrag[0], wfrag_wd, P.CS, P.SCp, acc[0], hook);
                if constexpr (NC > 1) mma_taps_wd<CL::th(1), CL::tw(1), DI, PW, TS, S1>(xs, ws + NT0 * TS, xfrag[1], wfrag_wd, P.CS, P.SCp, acc[1], hook);
                if constexpr (NC > 2) mma_taps_wd<CL::th(2), CL::tw(2), DI, PW, TS, S1 + S2>(xs, ws + (NT0 + NT1) * TS, xfrag[2], wfrag_wd, P.CS, P.SCp, acc[2], hook);
                if constexpr (NC > 3) mma_taps_wd<CL::th(3), CL::tw(3), DI, PW, TS, S1 + S2 + S3>(xs, ws + (NT0 + NT1 + NT2) * TS, xfrag[3], wfrag_wd, P.CS, P.SCp, acc[3], hook);
            }
            static_assert(2 * XN < 64, "vmcnt field");
            if (masked) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XN) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XN) : "memory");
            __syncthreads();
            if (it_ < 8) stamp(4 + it_);
            ++it_;
        }
    } else {
    prefetch(ck_begin);
    commit(0);
    stamp(2);
    if (ck_begin + CK < ck_end) prefetch(ck_begin + CK);
    __syncthreads();
    stamp(3);
    // The two waves that share a SIMD run the chunk in OPPOSITE order: waves of the first half stage chunk c+1 (LDS commit +
    // global prefetch of c+2) and then multiply chunk c, waves of the second half multiply first and stage afterwards.  The
    // vector-memory / LDS-write burst of one half therefore overlaps the MFMA phase of the other (with every wave in the
    // same phase the staging burst saturates the CU's single texture-address path while the matrix pipes idle: measured
    // staging 5.5 us + MFMA 13.1 us = 29 us kernel, i.e. zero overlap).
    const bool stage_first = NTHR == 256 || wave < (NTHR / 128);
    auto stage_next = [&](int ck0) {
        if (ck0 + CK < ck_end && !(P.dbg & 1)) {
            if (!(P.dbg & 16)) commit(buf ^ 1);
            if (ck0 + 2 * CK < ck_end && !(P.dbg & 32)) prefetch(ck0 + 2 * CK);
        }
    };
    for (int ck0 = ck_begin; ck0 < ck_end; ck0 += CK, buf ^= 1) {
        if (stage_first) stage_next(ck0);
        if (fine && it_ < 4) stamp(4 + 2 * it_);
        if (!(P.dbg & 2)) {
            const float* xs = smem + ((P.dbg & 1) ? 0 : buf * STAGE);
            const float* ws = xs + XS_SZ;
            mma_taps<CL::th(0), CL::tw(0), DI, PW, CK, RS>(xs, ws, xfrag[0], wfrag, P.CS, P.SCp, acc[0]);
            if constexpr (NC > 1) mma_taps<CL::th(1), CL::tw(1), DI, PW, CK, RS>(xs, ws + NT0 * CK * RS, xfrag[1], wfrag, P.CS, P.SCp, acc[1]);
            if constexpr (NC > 2) mma_taps<CL::th(2), CL::tw(2), DI, PW, CK, RS>(xs, ws + (NT0 + NT1) * CK * RS, xfrag[2], wfrag, P.CS, P.SCp, acc[2]);
            if constexpr (NC > 3) mma_taps<CL::th(3), CL::tw(3), DI, PW, CK, RS>(xs, ws + (NT0 + NT1 + NT2) * CK * RS, xfrag[3], wfrag, P.CS, P.SCp, acc[3]);
        }
        if (fine && it_ < 4 && stage_first) stamp(5 + 2 * it_);
        if (!stage_first) stage_next(ck0);
        if (fine && it_ < 4 && !stage_first) stamp(5 + 2 * it_);
        __syncthreads();
        if (!fine && it_ < 8) stamp(4 + it_);
        ++it_;
    }
    }
    stamp(12);

    // ---- epilogue.  Global stores are issue-bound (~500 cycles per wave-instruction when the whole chip stores at once), so
    //      the tile is not written by the accumulator owners (16-64 dword stores per lane, and only the ks==0 waves after a
    //      k-split combine) but re-distributed through LDS: every wave parks its partial accumulators as
    //      red[ks][class][cn][pixel], then EVERY thread adds the KS partials of 4 consecutive pixels of one channel and
    //      writes them as float4 (forward / all-class data-gradient: two float4 = 8 consecutive output floats).
    constexpr int TMW = 32 * WM;
    float* red = smem;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cnl = wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            red[((ks * NC + c) * TNW + cnl) * TMW + wm * 32 + l31] = acc[c][r];
        }
    __syncthreads();
    stamp(13);
    if (P.dbg & 8) return;      // (timing experiment: skip the output stores)
    const bool direct = P.SK == 1;
    float* outp = direct ? P.out : P.out + (size_t)split * P.out_elems;   // P.out = partial slab when SK > 1
    const int chw = P.Hout * P.Wout;
    constexpr int GROUPS = NC == 4 ? 2 : NC;           // stores per (channel, pixel quad): classes, or row parities
    constexpr int UNITS = GROUPS * TNW * (TMW / 4);
    const bool quad_rows = (P.TC & 3) == 0;            // 4 consecutive tile pixels share (image, row)
    for (int u = tid; u < UNITS; u += NTHR) {
        const int p4 = u % (TMW / 4);
        const int cnl = (u / (TMW / 4)) % TNW;
        // (a group is (TMW / 4) * TNW = a multiple of 64 units: wave-uniform, so the class records below are scalar loads -- indexed by
        //  a per-lane value they were dependent VECTOR loads from the kernel-argument block)
        static_assert(((TMW / 4) * TNW) % 64 == 0, "store groups are whole waves");
        const int gsel = __builtin_amdgcn_readfirstlane(u / ((TMW / 4) * TNW));
        const int cn = cn0 + cnl;
        if (cn >= P.CNtot) continue;
        const int ca = NC == 4 ? 2 * gsel : gsel, cb = NC == 4 ? 2 * gsel + 1 : gsel;
        float va[4] = {0.f, 0.f, 0.f, 0.f}, vb[4] = {0.f, 0.f, 0.f, 0.f};
        // (all KS partials fetched first, then added in k order: the rolled loop paid one LDS round trip per k)
        float4 ta[KS], tb[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            ta[k] = *reinterpret_cast<const float4*>(red + ((k * NC + ca) * TNW + cnl) * TMW + 4 * p4);
            if (NC == 4) tb[k] = *reinterpret_cast<const float4*>(red + ((k * NC + cb) * TNW + cnl) * TMW + 4 * p4);
        }
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            va[0] += ta[k].x; va[1] += ta[k].y; va[2] += ta[k].z; va[3] += ta[k].w;
            if (NC == 4) { vb[0] += tb[k].x; vb[1] += tb[k].y; vb[2] += tb[k].z; vb[3] += tb[k].w; }
        }
        if (direct) {
            const float bv = P.bias ? P.bias[cn] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                va[j] = act_apply(va[j] + bv, P.act, P.alpha);
                if (NC == 4) vb[j] = act_apply(vb[j] + bv, P.act, P.alpha);
            }
        }
        const CorrClass& A = P.cls[CL::cls(ca)];
        const CorrClass& B = P.cls[CL::cls(cb)];
        // first pixel of the quad
        const int p = 4 * p4;
        const int img = fdiv24(p, P.d_TRTC);
        const int rem = p - img * (P.TR * P.TC);
        const int ur = fdiv24(rem, P.d_TC);
        const int vc = rem - ur * P.TC;
        const bool img_ok = img < P.TI && (n0 + img) < P.N;
        const size_t cbase = ((size_t)(n0 + img) * P.CNtot + cn) * chw;
        bool done = false;
        if (quad_rows && img_ok && (P.Wout & 3) == 0) {
            if (NC != 4 && P.ocs == 1 && (A.Wv & 3) == 0 && (u0 + ur) < A.Hu && (v0 + vc + 3) < A.Wv) {
                const int off = (A.or0 + P.ors * (u0 + ur)) * P.Wout + A.oc0 + v0 + vc;
                if ((off & 3) == 0) {
                    if (KIND == 0 && P.out_ref) {
                        const float4 r4 = *reinterpret_cast<const float4*>(P.out_ref + cbase + off);
                        va[0] = act_grad(va[0], r4.x, P.out_act, P.out_alpha); va[1] = act_grad(va[1], r4.y, P.out_act, P.out_alpha);
                        va[2] = act_grad(va[2], r4.z, P.out_act, P.out_alpha); va[3] = act_grad(va[3], r4.w, P.out_act, P.out_alpha);
                    }
                    *reinterpret_cast<float4*>(outp + cbase + off) = make_float4(va[0], va[1], va[2], va[3]);
                    done = true;
                }
            }
            if (NC == 4 && A.Wv == B.Wv && A.Hu == B.Hu && (A.Wv & 3) == 0 && (u0 + ur) < A.Hu && (v0 + vc + 3) < A.Wv) {
                // the two column parities interleave into 8 consecutive floats of output row or0 + 2*(u0+ur)
                const bool a_even = A.oc0 < B.oc0;
                const int off = (A.or0 + P.ors * (u0 + ur)) * P.Wout + (a_even ? A.oc0 : B.oc0) + P.ocs * (v0 + vc);
                if ((off & 3) == 0 && P.ocs == 2) {
                    const float* e = a_even ? va : vb;
                    const float* o = a_even ? vb : va;
                    *reinterpret_cast<float4*>(outp + cbase + off) = make_float4(e[0], o[0], e[1], o[1]);
                    *reinterpret_cast<float4*>(outp + cbase + off + 4) = make_float4(e[2], o[2], e[3], o[3]);
                    done = true;
                }
            }
        }
        if (!done) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pj = p + j;
                const int im = fdiv24(pj, P.d_TRTC);
                const int rm = pj - im * (P.TR * P.TC);
                const int urj = fdiv24(rm, P.d_TC);
                const int vcj = rm - urj * P.TC;
                if (!(im < P.TI && (n0 + im) < P.N)) continue;
                const size_t cb2 = ((size_t)(n0 + im) * P.CNtot + cn) * chw;
                if ((u0 + urj) < A.Hu && (v0 + vcj) < A.Wv) {
                    const size_t oi = cb2 + (A.or0 + P.ors * (u0 + urj)) * P.Wout + A.oc0 + P.ocs * (v0 + vcj);
                    outp[oi] = (KIND == 0 && P.out_ref) ? act_grad(va[j], P.out_ref[oi], P.out_act, P.out_alpha) : va[j];
                }
                if (NC == 4 && (u0 + urj) < B.Hu && (v0 + vcj) < B.Wv)
                    outp[cb2 + (B.or0 + P.ors * (u0 + urj)) * P.Wout + B.oc0 + P.ocs * (v0 + vcj)] = vb[j];
            }
        }
    }
    stamp(14);
}

// MODE 0: fwd.  MODE 1: dgrad class pairs, blockIdx.z = pair * SK + split.  MODE 2: dgrad, all four classes per workgroup.
template <int MODE, int WM, int WN, int KS, int PW, bool X4 = false>
__global__ __launch_bounds__(64 * WM * WN * KS) void corr_kernel(const CorrParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    warm_kernarg(P);
    const int grp = blockIdx.z / P.SK, split = blockIdx.z - grp * P.SK;
    int bx = blockIdx.x, by = blockIdx.y;
    if (P.xcd_p > 0) {
        // (gridDim.x % xcd_p == 0, gridDim.y % (8 / xcd_p) == 0 and gridDim.x * gridDim.y % 8 == 0: checked by the launcher)
        const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, slot = lin >> 3;
        const int pg = xcd % P.xcd_p, cg = xcd / P.xcd_p;
        const int npx = gridDim.x / P.xcd_p, nch = gridDim.y / (8 / P.xcd_p);
        bx = pg * npx + slot % npx;
        by = cg * nch + slot / npx;
    }
    if constexpr (MODE == 0) {
        corr_body<0, 2, 1, WM, WN, KS, PW, X4>(P, split, smem, bx, by);
    } else if constexpr (MODE == 2) {
        corr_body<3, 1, -1, WM, WN, KS, PW, X4>(P, split, smem, bx, by);
    } else if (grp == 0) {
        corr_body<1, 1, -1, WM, WN, KS, PW, X4>(P, split, smem, bx, by);
    } else {
        corr_body<2, 1, -1, WM, WN, KS, PW, X4>(P, split, smem, bx, by);
    }
}

// out[idx] = act(sum_s partial[s*stride + idx] + bias[c]); entries idx >= elems (a "tail" of n2 extra sums, e.g. the bias
// gradient riding along with the filter-gradient slabs) go to out2 unmodified.
__global__ void splitk_reduce_k(const float* __restrict__ partial, int SK, size_t elems, size_t stride, float* __restrict__ out,
                                const float* __restrict__ bias, int C, int HW, int act, float alpha, float* __restrict__ out2,
                                size_t n2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < elems + n2; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < SK; ++k) s += partial[(size_t)k * stride + i];      // (independent loads, ordered adds)
        if (i < elems) {
            if (bias) s += bias[(i / (size_t)HW) % (size_t)C];
            out[i] = act_apply(s, act, alpha);
        } else {
            out2[i - elems] = s;
        }
    }
}

// Many slabs, few outputs (filter gradients of the 3-channel layers: 64 slabs x 4.8 K floats): 64 outputs per block,
// the slabs dealt to 4 thread groups, combined through LDS in group order (deterministic).
__global__ void splitk_reduce_small_k(const float* __restrict__ partial, int SK, size_t elems, size_t stride, float* __restrict__ out,
                                      const float* __restrict__ bias, int C, int HW, int act, float alpha,
                                      float* __restrict__ out2, size_t n2) {
    __shared__ float sm[4][64];
    const int o = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + o;
    float s = 0.f;
    if (i < elems + n2)
#pragma unroll 4
        for (int k = kg; k < SK; k += 4) s += partial[(size_t)k * stride + i];
    sm[kg][o] = s;
    __syncthreads();
    if (kg == 0 && i < elems + n2) {
        float v = (sm[0][o] + sm[1][o]) + (sm[2][o] + sm[3][o]);
        if (i < elems) {
            if (bias) v += bias[(i / (size_t)HW) % (size_t)C];
            out[i] = act_apply(v, act, alpha);
        } else {
            out2[i - elems] = v;
        }
    }
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

struct WaveCfg { int WM, WN, KS, PW; };
// fwd: 25 taps per chunk-channel-pair; dgrad class pairs carry only 12-13 taps, so they stage twice as many channels per
// chunk (PW doubled) to keep ~25+ MFMAs per wave between barriers
const WaveCfg kCfgsFwd[9] = {{2, 2, 1, 2}, {2, 1, 2, 2}, {1, 1, 4, 1}, {1, 1, 4, 2}, {2, 1, 4, 1}, {1, 1, 8, 1}, {2, 2, 2, 1}, {2, 1, 4, 2}, {4, 1, 2, 2}};
const WaveCfg kCfgsDgrad[9] = {{2, 2, 1, 4}, {2, 1, 2, 4}, {1, 1, 4, 2}, {1, 1, 4, 4}, {2, 1, 4, 2}, {1, 1, 8, 1}, {2, 2, 2, 2}, {1, 1, 8, 2}, {1, 1, 8, 2}};

// pixel tile TI x TR x TC <= TM whose CK-channel slab fits the per-thread staging budget
bool pick_tile(CorrParams& P, int Hu, int Wv, int TM, int CK, int su, int ext_r, int ext_c) {
    const int budget = (su == 2 ? XE_MAX : XE_MAX / 2) * 256;     // staged slab elements per chunk (fwd / dgrad)
    P.TC = Wv < TM ? Wv : TM;
    P.TR = TM / P.TC; if (P.TR < 1) P.TR = 1; if (P.TR > Hu) P.TR = Hu;
    P.TI = TM / (P.TR * P.TC); if (P.TI < 1) P.TI = 1; if (P.TI > P.N) P.TI = P.N;
    for (;;) {
        P.SR = su * (P.TR - 1) + ext_r;
        P.SCp = su * (P.TC - 1) + ext_c;
        P.CS = P.TI * P.SR * P.SCp;
        if (CK * P.CS <= budget) return true;
        if (P.TI > 1) P.TI = (P.TI + 1) / 2;
        else if (P.TR > 1) P.TR = (P.TR + 1) / 2;
        else return false;
    }
}

bool finish_tile(CorrParams& P, int Hu, int Wv) {
    const uint32_t nmax = XE_MAX * 256;
    const int xq = P.xq == 4 ? 4 : 1;
    if (!make_fastdiv24(P.CS / xq, nmax, &P.d_CS) || !make_fastdiv24(P.SR * P.SCp / xq, nmax, &P.d_SRSC) || !make_fastdiv24(P.SCp / xq, nmax, &P.d_SCp) ||
        !make_fastdiv24(P.TR * P.TC, nmax, &P.d_TRTC) || !make_fastdiv24(P.TC, nmax, &P.d_TC))
        return false;
    P.img_groups = cdiv(P.N, P.TI);
    P.tiles_r = cdiv(Hu, P.TR); P.tiles_c = cdiv(Wv, P.TC);
    return true;
}

template <typename K>
void allow_big_lds(K kernel) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// wave layouts of the data-gradient kinds that have a 16-byte-unit variant (launch_cfg)
bool dgrad_x4_cfg(int mode, int cfg) { return mode == 1 ? (cfg == 4 || cfg == 5 || cfg == 7) : (cfg == 5 || cfg == 6 || cfg == 7); }

template <int MODE>
int launch_cfg(int cfg, const CorrParams& P, dim3 grid, size_t shmem, hipStream_t s, const char* name, double fl) {
    // algorithmic bytes of the launch: input + filter + output, one pass each (SURVEY.md App. B "bytes")
    const double ab = (double)P.in_bytes + (double)P.w_bytes + 4.0 * (double)P.out_elems;
    static std::atomic<unsigned long long> once{0};
    if (first_on_device(once)) {   // double-buffered staging can exceed the 64 KiB default dynamic-LDS limit
        allow_big_lds(corr_kernel<0, 2, 2, 1, 2>); allow_big_lds(corr_kernel<0, 2, 1, 2, 2>); allow_big_lds(corr_kernel<0, 1, 1, 4, 1>);
        allow_big_lds(corr_kernel<1, 2, 2, 1, 4>); allow_big_lds(corr_kernel<1, 2, 1, 2, 4>); allow_big_lds(corr_kernel<1, 1, 1, 4, 2>);
        allow_big_lds(corr_kernel<0, 1, 1, 4, 2>); allow_big_lds(corr_kernel<1, 1, 1, 4, 4>);
        allow_big_lds(corr_kernel<0, 2, 1, 4, 1>); allow_big_lds(corr_kernel<0, 1, 1, 8, 1>);
        allow_big_lds(corr_kernel<1, 2, 1, 4, 2>); allow_big_lds(corr_kernel<1, 1, 1, 8, 1>);
        allow_big_lds(corr_kernel<0, 2, 2, 2, 1>); allow_big_lds(corr_kernel<1, 2, 2, 2, 2>);
        allow_big_lds(corr_kernel<2, 2, 2, 1, 2>); allow_big_lds(corr_kernel<2, 2, 1, 2, 2>); allow_big_lds(corr_kernel<2, 1, 1, 4, 1>);
        allow_big_lds(corr_kernel<2, 1, 1, 4, 2>); allow_big_lds(corr_kernel<2, 2, 1, 4, 1>); allow_big_lds(corr_kernel<2, 1, 1, 8, 1>);
        allow_big_lds(corr_kernel<2, 2, 2, 2, 1>); allow_big_lds(corr_kernel<0, 2, 1, 4, 2>); allow_big_lds(corr_kernel<2, 2, 1, 4, 2>);
        allow_big_lds(corr_kernel<0, 4, 1, 2, 2>); allow_big_lds(corr_kernel<2, 4, 1, 2, 2>); allow_big_lds(corr_kernel<1, 1, 1, 8, 2>);
        allow_big_lds(corr_kernel<0, 2, 1, 4, 1, true>); allow_big_lds(corr_kernel<0, 1, 1, 8, 1, true>); allow_big_lds(corr_kernel<0, 2, 2, 2, 1, true>);
        allow_big_lds(corr_kernel<0, 2, 1, 4, 2, true>); allow_big_lds(corr_kernel<0, 4, 1, 2, 2, true>);
        allow_big_lds(corr_kernel<1, 2, 1, 4, 2, true>); allow_big_lds(corr_kernel<1, 1, 1, 8, 1, true>); allow_big_lds(corr_kernel<1, 1, 1, 8, 2, true>);
        allow_big_lds(corr_kernel<2, 1, 1, 8, 1, true>); allow_big_lds(corr_kernel<2, 2, 1, 4, 2, true>); allow_big_lds(corr_kernel<2, 2, 2, 2, 1, true>);
    }
    if (shmem > 160 * 1024) { set_error("%s: LDS request %zu too large", name, shmem); return -3; }
    if constexpr (MODE == 0) {
        if (P.xq == 4) {       // slab staged in 16-byte units (the 8-wave layouts only: plan_and_launch)
            switch (cfg) {
                case 4: GGAN_LAUNCH("corr_kernel<0, 2, 1, 4, 1, true>", fl, ab, (corr_kernel<0, 2, 1, 4, 1, true>), grid, dim3(512), shmem, s, P); break;
                case 5: GGAN_LAUNCH("corr_kernel<0, 1, 1, 8, 1, true>", fl, ab, (corr_kernel<0, 1, 1, 8, 1, true>), grid, dim3(512), shmem, s, P); break;
                case 6: GGAN_LAUNCH("corr_kernel<0, 2, 2, 2, 1, true>", fl, ab, (corr_kernel<0, 2, 2, 2, 1, true>), grid, dim3(512), shmem, s, P); break;
                case 7: GGAN_LAUNCH("corr_kernel<0, 2, 1, 4, 2, true>", fl, ab, (corr_kernel<0, 2, 1, 4, 2, true>), grid, dim3(512), shmem, s, P); break;
                case 8: GGAN_LAUNCH("corr_kernel<0, 4, 1, 2, 2, true>", fl, ab, (corr_kernel<0, 4, 1, 2, 2, true>), grid, dim3(512), shmem, s, P); break;
                default: set_error("%s: no 16-byte-unit variant of wave layout %d", name, cfg); return -3;
            }
            return 0;
        }
    }
    if constexpr (MODE == 2) {
        if (P.xq == 4) {
            switch (cfg) {
                case 5: GGAN_LAUNCH("corr_kernel<2, 1, 1, 8, 1, true>", fl, ab, (corr_kernel<2, 1, 1, 8, 1, true>), grid, dim3(512), shmem, s, P); break;
                case 6: GGAN_LAUNCH("corr_kernel<2, 2, 2, 2, 1, true>", fl, ab, (corr_kernel<2, 2, 2, 2, 1, true>), grid, dim3(512), shmem, s, P); break;
                default: GGAN_LAUNCH("corr_kernel<2, 2, 1, 4, 2, true>", fl, ab, (corr_kernel<2, 2, 1, 4, 2, true>), grid, dim3(512), shmem, s, P); break;
            }
            return 0;
        }
    }
    if constexpr (MODE == 1) {
        if (P.xq == 4) {
            switch (cfg) {
                case 4: GGAN_LAUNCH("corr_kernel<1, 2, 1, 4, 2, true>", fl, ab, (corr_kernel<1, 2, 1, 4, 2, true>), grid, dim3(512), shmem, s, P); break;
                case 5: GGAN_LAUNCH("corr_kernel<1, 1, 1, 8, 1, true>", fl, ab, (corr_kernel<1, 1, 1, 8, 1, true>), grid, dim3(512), shmem, s, P); break;
                default: GGAN_LAUNCH("corr_kernel<1, 1, 1, 8, 2, true>", fl, ab, (corr_kernel<1, 1, 1, 8, 2, true>), grid, dim3(512), shmem, s, P); break;
            }
            return 0;
        }
    }
    if constexpr (MODE == 0 || MODE == 2) {
        switch (cfg) {
            case 0: GGAN_LAUNCH((MODE == 0 ? "corr_kernel<0, 2, 2, 1, 2, false>" : "corr_kernel<2, 2, 2, 1, 2, false>"), fl, ab, (corr_kernel<MODE, 2, 2, 1, 2>), grid, dim3(256), shmem, s, P); break;
            case 1: GGAN_LAUNCH((MODE == 0 ? "corr_kernel<0, 2, 1, 2, 2, false>" : "corr_kernel<2, 2, 1, 2, 2, false>"), fl, ab, (corr_kernel<MODE, 2, 1, 2, 2>), grid, dim3(256), shmem, s, P); break;
            case 2: GGAN_LAUNCH((MODE == 0 ? "corr_kernel<0, 1, 1, 4, 1, false>" : "corr_kernel<2, 1, 1, 4, 1, false>"), fl, ab, (corr_kernel<MODE, 1, 1, 4, 1>), grid, dim3(256), shmem, s, P); break;
            case 3: GGAN_LAUNCH((MODE == 0 ? "corr_kernel<0, 1, 1, 4, 2, false>" : "corr_kernel<2, 1, 1, 4, 2, false>"), fl, ab, (corr_kernel<MODE, 1, 1, 4, 2>), grid, dim3(256), shmem, s, P); break;
            case 4: GGAN_LAUNCH((MODE == 0 ? "corr_kernel<0, 2, 1, 4, 1, false>" : "corr_kernel<2, 2, 1, 4, 1, false>"), fl, ab, (corr_kernel<MODE, 2, 1, 4, 1>), grid, dim3(512), shmem, s, P); break;
            case 5: GGAN_LAUNCH((MODE == 0 ? "corr_kernel<0, 1, 1, 8, 1, false>" : "corr_kernel<2, 1, 1, 8, 1, false>"), fl, ab, (corr_kernel<MODE, 1, 1, 8, 1>), grid, dim3(512), shmem, s, P); break;
            case 6: GGAN_LAUNCH((MODE == 0 ? "corr_kernel<0, 2, 2, 2, 1, false>" : "corr_kernel<2, 2, 2, 2, 1, false>"), fl, ab, (corr_kernel<MODE, 2, 2, 2, 1>), grid, dim3(512), shmem, s, P); break;
            case 7: GGAN_LAUNCH((MODE == 0 ? "corr_kernel<0, 2, 1, 4, 2, false>" : "corr_kernel<2, 2, 1, 4, 2, false>"), fl, ab, (corr_kernel<MODE, 2, 1, 4, 2>), grid, dim3(512), shmem, s, P); break;
            default: GGAN_LAUNCH((MODE == 0 ? "corr_kernel<0, 4, 1, 2, 2, false>" : "corr_kernel<2, 4, 1, 2, 2, false>"), fl, ab, (corr_kernel<MODE, 4, 1, 2, 2>), grid, dim3(512), shmem, s, P); break;
        }
    } else {
        switch (cfg) {
            case 0: GGAN_LAUNCH("corr_kernel<1, 2, 2, 1, 4, false>", fl, ab, (corr_kernel<1, 2, 2, 1, 4>), grid, dim3(256), shmem, s, P); break;
            case 1: GGAN_LAUNCH("corr_kernel<1, 2, 1, 2, 4, false>", fl, ab, (corr_kernel<1, 2, 1, 2, 4>), grid, dim3(256), shmem, s, P); break;
            case 2: GGAN_LAUNCH("corr_kernel<1, 1, 1, 4, 2, false>", fl, ab, (corr_kernel<1, 1, 1, 4, 2>), grid, dim3(256), shmem, s, P); break;
            case 3: GGAN_LAUNCH("corr_kernel<1, 1, 1, 4, 4, false>", fl, ab, (corr_kernel<1, 1, 1, 4, 4>), grid, dim3(256), shmem, s, P); break;
            case 4: GGAN_LAUNCH("corr_kernel<1, 2, 1, 4, 2, false>", fl, ab, (corr_kernel<1, 2, 1, 4, 2>), grid, dim3(512), shmem, s, P); break;
            case 5: GGAN_LAUNCH("corr_kernel<1, 1, 1, 8, 1, false>", fl, ab, (corr_kernel<1, 1, 1, 8, 1>), grid, dim3(512), shmem, s, P); break;
            case 6: GGAN_LAUNCH("corr_kernel<1, 2, 2, 2, 2, false>", fl, ab, (corr_kernel<1, 2, 2, 2, 2>), grid, dim3(512), shmem, s, P); break;
            default: GGAN_LAUNCH("corr_kernel<1, 1, 1, 8, 2, false>", fl, ab, (corr_kernel<1, 1, 1, 8, 2>), grid, dim3(512), shmem, s, P); break;
        }
    }
    return 0;
}

// Plan-time slab table of the forward DMA path (corr_body, `tabled`): for every tile position and every staging element e = tid + j * NTHR
// of a chunk the byte offset of its image unit relative to the tile's first image and the chunk's first channel, or the out-of-range
// offset for the halo.  Built once per (device, geometry, tile plan) and kept; NULL while a stream capture is under way and the plan
// is not there yet (the launch then computes its descriptors in the kernel, as before).
std::mutex g_xtab_mu;
std::map<std::vector<int>, unsigned*> g_xtabs;

const unsigned* fwd_slab_table(const CorrParams& P, int CK, int nthr, int su, hipStream_t s) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int xq = P.xq == 4 ? 4 : 1;
    const int XE = XE_MAX * 256 / nthr;
    const std::vector<int> key = {dev, P.CKtot, P.Hin, P.Win, P.TI, P.TR, P.TC, P.SR, P.SCp, P.CS, xq, P.row0, P.col0, CK, nthr, P.tiles_r, P.tiles_c, su};
    std::lock_guard<std::mutex> lk(g_xtab_mu);
    auto it = g_xtabs.find(key);
    if (it != g_xtabs.end()) return it->second;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    const int CSu = P.CS / xq, SCpu = P.SCp / xq, srsc = P.SR * SCpu, xe_cnt = CK * CSu;
    const int npos = P.tiles_r * P.tiles_c;
    std::vector<unsigned> tab((size_t)npos * XE * nthr, OOB);
    for (int tr = 0; tr < P.tiles_r; ++tr)
        for (int tc = 0; tc < P.tiles_c; ++tc) {
            const int in_row0 = su * tr * P.TR + P.row0, in_col0 = su * tc * P.TC + P.col0;
            for (int j = 0; j < XE; ++j)
                for (int tid = 0; tid < nthr; ++tid) {
                    const int e = tid + j * nthr;
                    if (e >= xe_cnt) continue;
                    const int ckl = e / CSu, r1 = e - ckl * CSu;
                    const int img = r1 / srsc, r2 = r1 - img * srsc;
                    const int r = r2 / SCpu, cc = (r2 - r * SCpu) * xq;
                    const int ih = in_row0 + r, iw = in_col0 + cc;
                    if (img >= P.TI || ih < 0 || ih >= P.Hin || iw < 0 || iw >= P.Win) continue;
                    tab[((size_t)(tr * P.tiles_c + tc) * XE + j) * nthr + tid] = (unsigned)((((size_t)img * P.CKtot + ckl) * P.Hin + ih) * P.Win + iw) * 4u;
                }
        }
    unsigned* d = nullptr;
    if (hipMalloc((void**)&d, tab.size() * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemcpy(d, tab.data(), tab.size() * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d); return nullptr; }
    g_xtabs[key] = d;
    return d;
}

// common tail: wave-config choice, split-K fallback, launch
template <int MODE>
int plan_and_launch(CorrParams& P, int Hu, int Wv, int su, int ext_r, int ext_c, int ntaps, int groups, float* dst,
                    const float* bias, int act, float alpha, void* ws, size_t ws_bytes, hipStream_t s, const char* name,
                    double fl, const char* sk_env, const char* cfg_env) {
    const int target = P.plan_wgs > 0 ? P.plan_wgs : env_int("GGAN_TARGET_WGS", 200);
    const WaveCfg* kCfgs = MODE == 1 ? kCfgsDgrad : kCfgsFwd;
    int cfg = env_int(cfg_env, -1);
    if (cfg < 0 || cfg > (MODE == 1 ? 7 : 8)) {
        // 8 waves per workgroup (two per SIMD: one wave's LDS / barrier stalls hide under the other's MFMAs; measured
        // 12-19 % faster than the 4-wave layouts).  Largest tile that still yields ~one workgroup per CU.
        // 128x32, 64x64, 64x32, 32x32 (pixels x channels).  The filter slice is 2/3 of what a 64x32 workgroup stages per chunk and
        // is the same for every pixel tile: the 128-pixel layout stages it once for twice the MFMA work (forward kinds only:
        // measured 43.2 vs 47.9 us on the critic's 64->128 layer at 128 images, slower for the data-gradient kinds)
        // (all-class data gradient: the 64x32 tile with two accumulator pairs per wave = 16-channel chunks, so that its filter slice
        //  goes by LDS-DMA with 8-byte fragment reads: 47.4 vs 53.9 us on the 64->128 layer at 128 images)
        static const int order_fwd[4] = {8, 6, 4, 5}, order_all[4] = {8, 6, 7, 5};
        const int* order = MODE == 2 ? order_all : order_fwd;
        // class pairs: 32x32 tiles with 32-channel chunks (13 / 12 taps leave room for them in LDS).  Measured equal to the
        // 16-channel chunks (33.6 vs 33.9 us on the 128->256 layer: the time per chunk follows the staged bytes, not the barriers),
        // so it stays an option
        const bool deep_pairs = MODE == 1 && getenv("GGAN_DEEP_PAIRS");
        const int first = (MODE == 0 && !getenv("GGAN_NO_WIDE_TILE")) ? 0 : 1;
        cfg = 5;
        for (int oi = first; oi < 4; ++oi) {
            const int c = order[oi];
            const WaveCfg& wc = kCfgs[c];
            const int CK = 2 * wc.KS * wc.PW, TM = 32 * wc.WM, TNW = 32 * wc.WN;
            if (P.CNtot <= 32 && TNW > 32) continue;                               // don't pad tiny channel counts to 64
            CorrParams T = P;
            if (!pick_tile(T, Hu, Wv, TM, CK, su, ext_r, ext_c)) continue;
            if (oi < 3 && T.TI * T.TR * T.TC * 2 <= TM && Hu * Wv * P.N >= TM) continue;   // staging budget forced a half-empty tile
            const int wgs = cdiv(P.N, T.TI) * cdiv(Hu, T.TR) * cdiv(Wv, T.TC) * cdiv(P.CNtot, TNW) * groups;
            // (the 128-pixel layout pays only while the grid is about one workgroup per CU: with several rounds of 64x64 tiles --
            //  the 512-frame launches of the state-space scripts -- those stay faster, 200 vs 190+ us measured)
            if (oi == 0 && wgs >= 2 * target) continue;
            if (wgs >= target || oi == 3) { cfg = c; break; }
        }
        if (deep_pairs && cfg == 5 && P.CKtot >= 64) {
            // 13 / 12 taps per chunk-channel leave room for 32-channel chunks in LDS: twice the MFMAs per barrier
            CorrParams T = P;
            if (pick_tile(T, Hu, Wv, 32, 32, su, ext_r, ext_c) && !(T.TI * T.TR * T.TC * 2 <= 32 && Hu * Wv * P.N >= 32)) cfg = 7;
        }
        if (P.CKtot < 8) cfg = P.CNtot <= 32 ? 1 : 6;                              // 3-channel inputs: smallest chunks
    }
    const WaveCfg& wc = kCfgs[cfg];
    const int CK = 2 * wc.KS * wc.PW, TM = 32 * wc.WM, TNW = 32 * wc.WN;
    if (!pick_tile(P, Hu, Wv, TM, CK, su, ext_r, ext_c)) return 1;
    P.dbg = env_int("GGAN_DBG", 0);
    P.dma = (MODE == 0 ? env_int("GGAN_CORR_DMA", 1) : (CK % 16 == 0 && wc.PW <= 2 && env_int("GGAN_DGRAD_DMA", 1))) && (P.dbg & 3) == 0;
    const int RS = TNW + (MODE != 0 ? 2 : 0);     // filter row stride in LDS (corr_body: padded for the k-contiguous staging)
    P.xq = 1;
    if (MODE == 0 && cfg >= 4 && P.dma && (P.Win & 3) == 0 && (((uintptr_t)P.in) & 15) == 0 && ((su * P.TC) & 3) == 0 && env_int("GGAN_CORR_X4", 1)) {
        // Slab rows in 16-byte units of the image rows: a dword-gather DMA instruction costs the texture path ~64 cycles (a lane
        // per cycle) and the slab took 6 of them per wave and chunk -- as much texture-path time as the chunk has MFMA time.  With
        // the slab columns shifted so that LDS column c holds image column (unit-aligned start) + c, a lane moves 4 floats, and
        // units wholly outside the image (left / right halo) are the out-of-range lanes that arrive as zeros.
        const int shift = ((P.col0 % 4) + 4) % 4;
        const int scp4 = (P.SCp + shift + 3) & ~3;
        const size_t cs4 = (size_t)P.TI * P.SR * scp4;
        const int wregion = fwd_region(ntaps * CK * RS, ntaps * CK * (TNW / 4), 4);
        const size_t stage4 = 2 * ((size_t)fwd_region((int)(CK * cs4), (int)(CK * cs4 / 4), 4) + (size_t)wregion);
        // (not where the wider slab rows push a workgroup over half of the CU's LDS: two forward workgroups of the two graph
        //  branches then no longer share a CU -- measured +0.8 % on the iteration although the kernel alone is 1.6 % faster)
        const size_t stage1 = 2 * ((size_t)fwd_region(CK * P.CS, CK * P.CS, 1) + (size_t)wregion);
        const bool crosses = stage1 * sizeof(float) <= 80 * 1024 && stage4 * sizeof(float) > 80 * 1024;
        if (stage4 * sizeof(float) <= 160 * 1024 && CK * cs4 / 4 <= (size_t)XE_MAX * 256 && (!crosses || env_int("GGAN_CORR_X4", 1) > 1)) {
            P.xq = 4;
            P.col0 -= shift;
            for (int c = 0; c < 4; ++c) P.cls[c].coff += shift;
            P.SCp = scp4;
            P.CS = (int)cs4;
        }
    }
    if (MODE != 0 && dgrad_x4_cfg(MODE, cfg) && (P.Win & 3) == 0 && (((uintptr_t)P.in) & 15) == 0 &&
        (((uintptr_t)P.in_ref) & 15) == 0 && ((su * P.TC) & 3) == 0 && env_int("GGAN_DGRAD_X4", 1)) {
        // The same units for the register-staged slab of the data-gradient kinds: a thread fetches and commits 4 floats of an image
        // row per instruction (the 8-wave layouts staged 3 dwords per thread and chunk, twice that with the activation mask).
        const int shift = ((P.col0 % 4) + 4) % 4;
        const int scp4 = (P.SCp + shift + 3) & ~3;
        const size_t cs4 = (size_t)P.TI * P.SR * scp4;
        const int nthr = 64 * wc.WM * wc.WN * wc.KS;
        const int xn = ((XE_MAX / 2) * 256 / nthr + 3) / 4 + 1;           // corr_body: XN
        const size_t stage4 = 2 * ((size_t)((CK * cs4 + 1 + 3) & ~(size_t)3) + (size_t)((ntaps * CK * RS + 4 * RS + 3) & ~3));
        if (stage4 * sizeof(float) <= 160 * 1024 && CK * cs4 / 4 <= (size_t)xn * nthr) {
            P.xq = 4;
            P.col0 -= shift;
            for (int c = 0; c < 4; ++c) P.cls[c].coff += shift;
            P.SCp = scp4;
            P.CS = (int)cs4;
        }
    }
    if (!finish_tile(P, Hu, Wv)) return 1;
    if ((size_t)P.N * P.CKtot * P.Hin >= (1u << 24)) return 1;      // the staging descriptors use 24-bit multiplies
    const int gx = P.img_groups * P.tiles_r * P.tiles_c, gy = cdiv(P.CNtot, TNW);
    int sk = env_int(sk_env, 0);
    if (sk <= 0) {
        sk = 1;
        const int base = gx * gy * groups;
        if (base < target / 2) {            // still far from one workgroup per CU: global split-K
            sk = target / base;
            const int max_sk = P.CKtot / (2 * CK);
            if (sk > max_sk) sk = max_sk;
            if (sk > 16) sk = 16;
        }
    }
    if (sk < 1) sk = 1;
    P.cps = cdiv(cdiv(P.CKtot, sk), CK) * CK;
    P.SK = cdiv(P.CKtot, P.cps);
    if (P.SK > 1 && (size_t)P.SK * P.out_elems * sizeof(float) > ws_bytes) {
        P.SK = 1;
        P.cps = cdiv(P.CKtot, CK) * CK;
    }
    P.out = P.SK > 1 ? (float*)ws : dst;
    P.bias = bias; P.act = act; P.alpha = alpha;
    if (g_out_mask) {
        // the stored values masked by a reference tensor (conv.h: OutMask): only where the epilogue writes the final values
        if (MODE != 0 || P.SK != 1) return 1;
        P.out_ref = g_out_mask->ref; P.out_act = g_out_mask->act; P.out_alpha = g_out_mask->alpha;
        g_out_mask->applied = true;
    }
    if ((P.dbg & 4) && ws && ws_bytes > (64u << 20)) P.stamps = (unsigned long long*)((char*)ws + ws_bytes - (32u << 20));
    size_t stage = MODE == 0 ? 2 * ((size_t)fwd_region(CK * P.CS, CK * P.CS / P.xq, P.xq) + (size_t)fwd_region(ntaps * CK * RS, ntaps * CK * (TNW / 4), 4))
                             : 2 * ((size_t)((CK * P.CS + 1 + 3) & ~3) + (size_t)((ntaps * CK * RS + 4 * RS + 3) & ~3));
    size_t red = (size_t)wc.KS * (MODE == 2 ? 4 : (MODE == 1 ? 2 : 1)) * TNW * TM;     // epilogue: [ks][class][cn][pixel]
    P.nstg = 2;
    if (MODE == 0 && P.dma && env_int("GGAN_CORR_NSTG", 2) >= 3 && stage / 2 * 3 * sizeof(float) <= 160 * 1024 && P.cps >= 3 * CK) {
        // round 4 experiment (GGAN_CORR_NSTG=3): three staging buffers, fetched two chunks ahead.  Measured SLOWER than the two-buffer
        // scheme on every forward layout (64->128 @16: 24.6 -> 26.1 us, chunk 3990 -> 4250 cycles, first barrier 5440 -> 6600; face 32->64
        // at 128 images 40.3 -> 48.1 us; iteration 1.046 -> 1.057 ms): the loads dealt out from the first MFMA pair on already land within
        // their chunk, and twice the bytes in flight only lengthen every request's queue.  Off by default.
        P.nstg = 3;
        stage = stage / 2 * 3;
    }
    if (MODE == 0 && P.dma && env_int("GGAN_CORR_XTAB", 1)) P.xtab = fwd_slab_table(P, CK, 64 * wc.WM * wc.WN * wc.KS, su, s);
    const size_t shmem = (stage > red ? stage : red) * sizeof(float);
    {   // XCD-aware tile order: what the eight L2s fetch together is 8 * input / p + filter * p for p pixel-tile groups
        P.xcd_p = 0;
        const int force = env_int("GGAN_CORR_XCD", -1);
        if (force != 0 && (gx * gy) % 8 == 0) {
            double best = (gx % 8 == 0) ? (double)P.in_bytes + 8.0 * P.w_bytes : 8.0 * ((double)P.in_bytes + P.w_bytes);    // plain order
            for (int p = 1; p <= 8; p *= 2) {
                if (gx % p || gy % (8 / p) || (force > 0 && p != force)) continue;
                const double cost = 8.0 * P.in_bytes / p + (double)P.w_bytes * p;
                if (cost < 0.9 * best || force > 0) { best = cost; P.xcd_p = p; }
            }
        }
    }
    int rc = launch_cfg<MODE>(cfg, P, dim3(gx, gy, groups * P.SK), shmem, s, name, fl);
    if (rc) return rc;
    if (P.SK > 1)
        return launch_splitk_reduce((const float*)ws, P.SK, P.out_elems, dst, bias, P.CNtot, P.Hout * P.Wout, act, alpha, s);
    return 0;
}

}  // namespace

namespace ggan {

thread_local OutMask* g_out_mask = nullptr;

int launch_splitk_reduce(const float* partial, int SK, size_t elems, float* out, const float* bias, int C, int HW, int act,
                         float alpha, hipStream_t s, size_t slab_stride, float* tail_out, size_t tail) {
    if (slab_stride == 0) slab_stride = elems;
    if (!tail_out) tail = 0;
    if (SK >= 8 && elems + tail <= 65536) {
        GGAN_LAUNCH("splitk_reduce_small_k", 0, 4.0 * (elems + tail) * (SK + 1), splitk_reduce_small_k,
                    dim3((int)((elems + tail + 63) / 64)), dim3(256), 0, s, partial, SK, elems, slab_stride, out, bias, C, HW, act,
                    alpha, tail_out, tail);
        return 0;
    }
    size_t b = (elems + tail + 255) / 256;
    if (b > 2048) b = 2048;
    GGAN_LAUNCH("splitk_reduce_k", 0, 4.0 * (elems + tail) * (SK + 1), splitk_reduce_k, dim3((int)b), dim3(256), 0, s, partial,
                SK, elems, slab_stride, out, bias, C, HW, act, alpha, tail_out, tail);
    return 0;
}

size_t conv_workspace_bytes(const ggan_conv_geom& g) {
    // upper bound over the three ops: 16 split-K slabs of the largest result / 64 filter-gradient slabs
    size_t big = (size_t)g.N * g.Ci * g.H * g.W, small_ = (size_t)g.N * g.Co * g.Ho * g.Wo;
    size_t wsz = (size_t)g.k * g.k * g.Ci * g.Co;
    size_t m = big > small_ ? big : small_;
    size_t a = 16 * m * sizeof(float), b = 64 * wsz * sizeof(float);
    return a > b ? a : b;
}

static bool hot_geometry(const ggan_conv_geom& g) { return g.k == 5 && g.stride == 2; }
static bool fits32(size_t bytes) { return bytes < 0x7FFFFFF0ull; }

int conv_fwd_mfma(const ggan_conv_geom& g, const float* x, const float* w, const float* bias, float* y, int act,
                  float alpha, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!hot_geometry(g) || (g.Co & 3)) return 1;
    ws = ws_scratch(ws, ws_bytes);
    const size_t in_bytes = (size_t)g.N * g.Ci * g.H * g.W * 4, w_bytes = (size_t)25 * g.Ci * g.Co * 4;
    if (!fits32(in_bytes) || !fits32(w_bytes) || !fits32((size_t)g.N * g.Co * g.Ho * g.Wo * 4)) return 1;
    if (((uintptr_t)w & 15) != 0) return 1;
    CorrParams P;
    memset(&P, 0, sizeof(P));
    P.in = x; P.w = w;
    P.plan_wgs = g.plan_wgs;
    P.in_bytes = (unsigned)in_bytes; P.w_bytes = (unsigned)w_bytes;
    P.N = g.N; P.CKtot = g.Ci; P.Hin = g.H; P.Win = g.W;
    P.CNtot = g.Co; P.Hout = g.Ho; P.Wout = g.Wo;
    P.ors = 1; P.ocs = 1;
    P.w_si = g.k * g.Ci * g.Co; P.w_sj = g.Ci * g.Co; P.w_sk = g.Co; P.w_sn = 1;
    P.row0 = -g.pad_t; P.col0 = -g.pad_l;
    CorrClass& c = P.cls[0];
    c.Hu = g.Ho; c.Wv = g.Wo; c.roff = 0; c.coff = 0; c.or0 = 0; c.oc0 = 0; c.wbase = 0;
    P.out_elems = (size_t)g.N * g.Co * g.Ho * g.Wo;
    const double fl = 2.0 * P.out_elems * g.Ci * 25.0;
    return plan_and_launch<0>(P, g.Ho, g.Wo, 2, 5, 5, 25, 1, y, bias, act, alpha, ws, ws_bytes, s, "conv_fwd_mfma", fl,
                              "GGAN_FWD_SK", "GGAN_FWD_CFG");
}

int conv_dgrad_mfma(const ggan_conv_geom& g, const float* gy, GyMask m, const float* w, const float* bias, float* gx,
                    int act, float alpha, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!hot_geometry(g)) return 1;
    ws = ws_scratch(ws, ws_bytes);
    // the filter is read in place (HWIO: the reduction channel co is the contiguous one, staged k-contiguously by corr_body)
    if ((g.Co & 3) || ((uintptr_t)w & 15)) return 1;
    const size_t in_bytes = (size_t)g.N * g.Co * g.Ho * g.Wo * 4, w_bytes = (size_t)25 * g.Ci * g.Co * 4;
    if (!fits32(in_bytes) || !fits32(w_bytes) || !fits32((size_t)g.N * g.Ci * g.H * g.W * 4)) return 1;
    const int S = 2;
    CorrParams P;
    memset(&P, 0, sizeof(P));
    P.in = gy; P.w = w;
    P.plan_wgs = g.plan_wgs;
    if (m.act != GGAN_ACT_NONE && m.act != GGAN_ACT_LRELU && m.act != GGAN_ACT_RELU) return 1;   // other masks: plain kernels
    if (m.act != GGAN_ACT_NONE) { P.in_ref = m.ref; P.in_act = m.act; P.in_alpha = m.alpha; }
    P.in_bytes = (unsigned)in_bytes; P.w_bytes = (unsigned)w_bytes;
    P.N = g.N; P.CKtot = g.Co; P.Hin = g.Ho; P.Win = g.Wo;
    P.CNtot = g.Ci; P.Hout = g.H; P.Wout = g.W;
    P.ors = S; P.ocs = S;
    P.w_si = S * g.k * g.Ci * g.Co; P.w_sj = S * g.Ci * g.Co; P.w_sk = 1; P.w_sn = g.Co;
    // per dimension and parity p (kh = p + S*i): output positions ih = off + S*a, input row oh = a + base - i
    int offs[2][2], bases[2][2], cnt[2][2], th[2][2], lo[2], hi[2];
    for (int d = 0; d < 2; ++d) {
        const int pad = d == 0 ? g.pad_t : g.pad_l, L = d == 0 ? g.H : g.W;
        lo[d] = 1 << 30; hi[d] = -(1 << 30);
        for (int p = 0; p < S; ++p) {
            const int off = ((p - pad) % S + S) % S;
            offs[d][p] = off;
            bases[d][p] = (off + pad - p) / S;
            cnt[d][p] = off < L ? (L - off + S - 1) / S : 0;
            th[d][p] = (g.k - p + S - 1) / S;
            if (bases[d][p] - (th[d][p] - 1) < lo[d]) lo[d] = bases[d][p] - (th[d][p] - 1);
            if (bases[d][p] > hi[d]) hi[d] = bases[d][p];
        }
    }
    const int Hu = cnt[0][0] > cnt[0][1] ? cnt[0][0] : cnt[0][1];
    const int Wv = cnt[1][0] > cnt[1][1] ? cnt[1][0] : cnt[1][1];
    P.row0 = lo[0]; P.col0 = lo[1];
    for (int ph = 0; ph < S; ++ph)
        for (int pw = 0; pw < S; ++pw) {
            CorrClass& c = P.cls[ph * 2 + pw];
            c.Hu = cnt[0][ph]; c.Wv = cnt[1][pw];
            c.roff = bases[0][ph] - lo[0]; c.coff = bases[1][pw] - lo[1];
            c.or0 = offs[0][ph]; c.oc0 = offs[1][pw];
            c.wbase = (ph * g.k + pw) * g.Ci * g.Co;
        }
    P.out_elems = (size_t)g.N * g.Ci * g.H * g.W;
    const double fl = 2.0 * g.N * g.Co * g.Ho * g.Wo * (double)g.Ci * 25.0;
    // all four parity classes in one workgroup (full-row float2 stores) when 32x32 tiles still give ~one workgroup per CU;
    // otherwise balanced class pairs (twice the workgroups)
    const long wgs_all = (long)cdiv(g.N * Hu * Wv, 32) * cdiv(g.Ci, 32);
    int mode = env_int("GGAN_DGRAD_MODE", 0);
    if (mode == 0) mode = wgs_all >= (g.plan_wgs > 0 ? g.plan_wgs : env_int("GGAN_TARGET_WGS", 200)) ? 2 : 1;
    if (mode == 2)
        return plan_and_launch<2>(P, Hu, Wv, 1, hi[0] - lo[0] + 1, hi[1] - lo[1] + 1, 25, 1, gx, bias, act, alpha, ws, ws_bytes,
                                  s, "conv_dgrad_mfma", fl, "GGAN_DGRAD_SK", "GGAN_DGRAD_CFG");
    return plan_and_launch<1>(P, Hu, Wv, 1, hi[0] - lo[0] + 1, hi[1] - lo[1] + 1, 13, 2, gx, bias, act, alpha, ws, ws_bytes,
                              s, "conv_dgrad_mfma", fl, "GGAN_DGRAD_SK", "GGAN_DGRAD_CFG");
}

}  // namespace ggan
