// The mixture critic of the gmgan scripts as ONE launch per direction:
//   HyperDiscriminator(z, k) = Linear('Discriminator.HyperInput', DIM_LATENT + N_COMS, 512) on tf.concat([z, k], 1) -> LeakyReLU
//                              -> Linear('Discriminator.Hyper2', 512, 512) -> LeakyReLU -> Linear('Discriminator.Hyper3', 512, 512)
//                              -> LeakyReLU -> Linear('Discriminator.HyperOutput', 512, 1)
//   (/root/reference/gmgan_inference_cifar10.py:255-271, gmgan_inference_mnist.py / gmgan_inference_face.py: the same net; dropout is
//    the identity: tf.layers.dropout without training=True).
//
// As four products through gemm_kernel the chain is 8 launches forward (three split-K products + their reduce launches + the tail) and
// 6-8 backward for 0.15 GFLOP per direction, and it sits on the critical path of both steps (the cost needs its logits, the Extractor's
// backward pass its data gradient).  Every layer is ROW-LOCAL (no BatchNorm in this critic), so here a workgroup owns 8 rows of the
// minibatch for the whole chain -- no seam between the layers, nothing but the kept activations goes back to memory:
//   * the rows' activations stay in LDS between layers ([8][512] fp32, input / output buffer of a layer);
//   * the weights stream from L2 straight into the MFMA B operand -- each element is used by exactly one MFMA pair per workgroup, so
//     there is nothing to stage -- with fully coalesced 16-byte loads in BOTH directions and a register ring of loads in flight.
//     A workgroup reads every weight once (2.3 MB at ~110 GB/s per CU): that stream, not the arithmetic, bounds the kernel, which
//     is why a workgroup carries only 8 rows (16 workgroups at 128 rows; v_mfma_f32_16x16x4_f32 tiles need 16 rows and measured
//     44 us forward / 73 us backward at 8 workgroups, against 34 / 49 us for the composed launches);
//   * v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer-product blocks per instruction, exact fp32 fmaf chains; layout
//     D[vgpr i][lane 4b+j] += A[lane 4b+i] B[lane 4b+j], tools/probes/mfma4x4_probe.hip): block b <-> 4 output columns, lane <-> column,
//     so a weight row is read as 64 lanes x 16 bytes (four column sets per lane); the rows' input values are the A operand, the same
//     in every block (LDS broadcast reads); the 8 waves are 2 column halves x 4 quarters of the reduction, combined through LDS in
//     fixed order;
//   * backward (g W^T): the reduction runs along a weight row's contiguous dimension, which no lane assignment of the MFMA turns into
//     coalesced loads (lane = 4 block + column: with blocks as reduction slices a quarter-wave reads 4 rows x 64 bytes, measured
//     75 us).  The FORWARD launch therefore also writes the three transposed weight matrices (its otherwise idle placement blocks
//     do it, 64x64 tiles through LDS, beside the chain workgroups), and the backward is the same product kernel on those: the
//     LeakyReLU mask of the layer below is applied in the combine step (reference activations fetched before the main loop), the
//     masked gradients gh2 / gh1 are written for the weight-gradient products (one grouped launch, gemm.hip).
// The workgroups are placed on ONE XCD (grid = 8 x blocks, only blocks b % 8 == 0 work: observed placement b -> XCD b % 8, used for
// speed only) so that the weights are fetched into one L2.  Deterministic: fixed summation order everywhere.
#include "mlp.h"
#include <stdlib.h>
using namespace ggan;

namespace {

constexpr int MC_H = kMlpHidden;
constexpr int MC_ROWS = 8;
constexpr int MC_LD = MC_H + 4;          // LDS row stride of an activation buffer (floats)
constexpr int MC_THR = 512;              // 8 waves
constexpr int MC_D = 4;                  // macro steps (4 k = four 16-byte loads per lane) in flight per wave
constexpr int MC_ACT = MC_ROWS * MC_LD;  // floats per activation buffer
constexpr int MC_RED = 4 * MC_ROWS * MC_H;                                                // the 4 reduction-quarter slabs
constexpr size_t MC_LDS = (size_t)(2 * MC_ACT + MC_RED) * sizeof(float);
constexpr int MC_TT = 64, MC_TLD = MC_TT + 1;                                             // transposer tile (fits the slabs' LDS)
constexpr int MC_TBLOCKS = 38;           // transposer blocks appended when the grid has no idle placement blocks (spread == 1)
constexpr int MC_SENT = 0x40000000;      // byte offset beyond every weight buffer (lanes without a column: loads return 0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ldw(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0);
    float4 v;
    v.x = __uint_as_float(t.x); v.y = __uint_as_float(t.y); v.z = __uint_as_float(t.z); v.w = __uint_as_float(t.w);
    return v;
}

__device__ __forceinline__ float lrelu(float v, float alpha) { return fmaxf(alpha * v, v); }

#define MC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

// macro steps (4 k each) per reduction quarter: K spread over 4 quarters, rounded up to a multiple of the ring depth
__host__ __device__ __forceinline__ int mc_steps(int K) { return ((K + 16 * MC_D - 1) / (16 * MC_D)) * MC_D; }

// ---- product core: red[quarter][8][N] = partial sums of in[8][K] W[K][N] (W row-major, N a multiple of 4, N <= 512) -----------------
// wave = (column half ch, reduction quarter kq); lane's tile t = columns 256 ch + 4 lane + t, rows 0..3 (acc0) and 4..7 (acc1).
// NS = macro steps (4 k) per quarter, FULLY UNROLLED: as a loop, the refilled ring slots are loop-carried values, and the register
// allocator kept every refill in fresh registers and copied them into the carried ones at the end of each iteration -- behind
// s_waitcnt vmcnt(0), so no load stayed in flight across an iteration (29 us forward; with the tail guards as branches the waits
// were vmcnt(0) at every step).  Straight-line, the compiler's wait counts are exact.  One copy of the code per NS (noinline).
// Ends with the barrier after which every thread may read all four slabs.
template <int NS>
__device__ __attribute__((noinline)) void mm_core(const float* __restrict__ in, float* __restrict__ red, const float* __restrict__ W, int K, int N) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, ch = wv & 1, kq = wv >> 1, i = lane & 3;
    const int k0 = kq * 4 * NS;                       // (K padded: the input buffer is zero there, rows k >= K of W lie beyond the buffer: 0)
    const int col = 256 * ch + 4 * lane;
    if (256 * ch < N) {                               // (wave-uniform: a half without columns has nothing to do)
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, (short)0, K * N * 4, 0x00020000);
        f32x4 acc0[4], acc1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc0[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        float4 ring[MC_D][4];
        const int lane_off = col < N ? (k0 * N + col) * 4 : MC_SENT;
        const int rowb = N * 4;
#pragma unroll
        for (int d = 0; d < MC_D; ++d) {
#pragma unroll
            for (int s = 0; s < 4; ++s) ring[d][s] = ldw(rs, lane_off + (4 * d + s) * rowb);
        }
        const float* a0p = in + i * MC_LD + k0;
        const float* a1p = in + (4 + i) * MC_LD + k0;
        // step st: [the rows' values of step st + 1 requested from LDS, slot (st - 1) % MC_D -- emptied by the step before -- refilled with
        // the weights of step st - 1 + MC_D] | sched_barrier | [32 MFMAs of step st].  The barrier keeps the machine scheduler from sinking
        // the loads of several steps into one bunch; the LDS values land while the step's MFMAs issue.
        float4 a0 = *reinterpret_cast<const float4*>(a0p), a1 = *reinterpret_cast<const float4*>(a1p);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int d = st % MC_D;
            float4 n0 = a0, n1 = a1;
            if (st + 1 < NS) {
                n0 = *reinterpret_cast<const float4*>(a0p + 4 * (st + 1));
                n1 = *reinterpret_cast<const float4*>(a1p + 4 * (st + 1));
            }
            if (st >= 1 && st - 1 + MC_D < NS) {
#pragma unroll
                for (int s = 0; s < 4; ++s) ring[(st - 1) % MC_D][s] = ldw(rs, lane_off + (4 * (st - 1 + MC_D) + s) * rowb);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float a0s[4] = {a0.x, a0.y, a0.z, a0.w}, a1s[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float4 w = ring[d][s];
                acc0[0] = MC_MFMA(a0s[s], w.x, acc0[0]); acc1[0] = MC_MFMA(a1s[s], w.x, acc1[0]);
                acc0[1] = MC_MFMA(a0s[s], w.y, acc0[1]); acc1[1] = MC_MFMA(a1s[s], w.y, acc1[1]);
                acc0[2] = MC_MFMA(a0s[s], w.z, acc0[2]); acc1[2] = MC_MFMA(a1s[s], w.z, acc1[2]);
                acc0[3] = MC_MFMA(a0s[s], w.w, acc0[3]); acc1[3] = MC_MFMA(a1s[s], w.w, acc1[3]);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0; a1 = n1;
        }
        // the quarter's partial sums: red[kq][row][col] (slab row stride 512); register r of tile t is row r (acc0) / 4 + r (acc1)
        if (col < N) {
            float* rq = red + (kq * MC_ROWS) * MC_H + col;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                *reinterpret_cast<float4*>(rq + r * MC_H) = float4{acc0[0][r], acc0[1][r], acc0[2][r], acc0[3][r]};
                *reinterpret_cast<float4*>(rq + (4 + r) * MC_H) = float4{acc1[0][r], acc1[1][r], acc1[2][r], acc1[3][r]};
            }
        }
    }
    __syncthreads();
}

// first layer: K = K1 + K2 <= 256 on the MC_D-step grid
__device__ __forceinline__ void mm_core_in(const float* __restrict__ in, float* __restrict__ red, const float* __restrict__ W, int K) {
    switch (mc_steps(K)) {
        case MC_D: mm_core<MC_D>(in, red, W, K, MC_H); break;
        case 2 * MC_D: mm_core<2 * MC_D>(in, red, W, K, MC_H); break;
        case 3 * MC_D: mm_core<3 * MC_D>(in, red, W, K, MC_H); break;
        default: mm_core<4 * MC_D>(in, red, W, K, MC_H); break;
    }
}
constexpr int MC_NSH = MC_H / 16;        // macro steps per quarter of a 512-long reduction
static_assert(MC_NSH % MC_D == 0 && kMlpMaxIn <= 64 * MC_D, "step grid");

// the four quarters of piece u (row u >> 7, columns 4 (u & 127) ..) added in order
__device__ __forceinline__ float4 red_sum(const float* __restrict__ red, int u) {
    const float* p = red + (u >> 7) * MC_H + 4 * (u & 127);
    float4 v = *reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int qq = 1; qq < 4; ++qq) {
        const float4 t = *reinterpret_cast<const float4*>(p + qq * MC_ROWS * MC_H);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    return v;
}

// ---- forward layer: out[8][512] = lrelu(in[8][K] W[K][512] + bias), kept in LDS and written to hout -----------------------------------
__device__ __forceinline__ void fwd_layer(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ red,
                                          const float* __restrict__ W, int K, const float* __restrict__ bias, float* __restrict__ hout, int row0,
                                          int M, float alpha) {
    if (K == MC_H) mm_core<MC_NSH>(in, red, W, MC_H, MC_H);
    else mm_core_in(in, red, W, K);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int u = threadIdx.x + MC_THR * q, r = u >> 7, c4 = u & 127;
        float4 v = red_sum(red, u);
        const float4 bb = *reinterpret_cast<const float4*>(bias + 4 * c4);
        v.x = lrelu(v.x + bb.x, alpha); v.y = lrelu(v.y + bb.y, alpha); v.z = lrelu(v.z + bb.z, alpha); v.w = lrelu(v.w + bb.w, alpha);
        *reinterpret_cast<float4*>(out + r * MC_LD + 4 * c4) = v;
        if (row0 + r < M) *reinterpret_cast<float4*>(hout + (size_t)(row0 + r) * MC_H + 4 * c4) = v;
    }
}

struct MlpFwdParams {
    const float* x1;
    const float* x2;
    const float* w[3];
    const float* b[3];
    const float* w_out;
    const float* b_out;
    float* h[3];
    float* logits;
    float* wt[3];        // transposed weights for the backward launch ([512][KP], [512][512], [512][512]; NULL: not wanted)
    int M, K1, K2, KP, spread, nblk;
    float alpha;
};

// dst[N][ldd] (columns < Kd written) = src[K][N]^T, 64 x 64 tiles through LDS; tiles tb, tb + ntb, ...; rows k >= K of the source read 0
__device__ __forceinline__ void transpose_tiles(const float* __restrict__ src, int K, int N, float* __restrict__ dst, int ldd, int Kd,
                                                float* __restrict__ tile, int& tb, int ntb) {
    const int tk = (K + MC_TT - 1) / MC_TT, tn = N / MC_TT, nt = tk * tn;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, K * N * 4, 0x00020000);
    const int tid = threadIdx.x;
    for (; tb < nt; tb += ntb) {
        const int k0 = (tb / tn) * MC_TT, n0 = (tb % tn) * MC_TT;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int u = tid + MC_THR * q, r = u >> 4, c4 = u & 15;
            const float4 v = ldw(rs, ((k0 + r) * N + n0 + 4 * c4) * 4);
            float* t = tile + r * MC_TLD + 4 * c4;
            t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int u = tid + MC_THR * q, n = u >> 4, k4 = u & 15;
            if (k0 + 4 * k4 < Kd) {
                const float* t = tile + (4 * k4) * MC_TLD + n;
                *reinterpret_cast<float4*>(dst + (size_t)(n0 + n) * ldd + k0 + 4 * k4) = float4{t[0], t[MC_TLD], t[2 * MC_TLD], t[3 * MC_TLD]};
            }
        }
        __syncthreads();
    }
    tb -= nt;
}

__global__ __launch_bounds__(MC_THR) void mlp_chain_fwd_k(const MlpFwdParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    if (bid >= P.nblk * P.spread || (bid % P.spread)) {
        // ---- a placement block / an appended block: the transposed weights for the backward launch ------------------------------------
        if (!P.wt[0]) return;
        int tb, ntb;
        if (P.spread > 1) {
            if (bid >= P.nblk * P.spread) return;
            tb = (bid / P.spread) * (P.spread - 1) + (bid % P.spread) - 1;
            ntb = P.nblk * (P.spread - 1);
        } else {
            tb = bid - P.nblk;
            ntb = MC_TBLOCKS;
        }
        transpose_tiles(P.w[1], MC_H, MC_H, P.wt[1], MC_H, MC_H, smem, tb, ntb);
        transpose_tiles(P.w[2], MC_H, MC_H, P.wt[2], MC_H, MC_H, smem, tb, ntb);
        transpose_tiles(P.w[0], P.K1 + P.K2, MC_H, P.wt[0], P.KP, P.KP, smem, tb, ntb);
        return;
    }
    float* bufA = smem;
    float* bufB = smem + MC_ACT;
    float* red = smem + 2 * MC_ACT;
    const int tid = threadIdx.x;
    const int row0 = (bid / P.spread) * MC_ROWS;
    const int K = P.K1 + P.K2, Kp = 16 * mc_steps(K);
    // ---- the rows' input [z | k], zero-padded to the step grid -------------------------------------------------------------------
    for (int u = tid; u < MC_ROWS * Kp; u += MC_THR) {
        const int r = u / Kp, c = u - r * Kp, row = row0 + r;
        float v = 0.f;
        if (row < P.M) {
            if (c < P.K1) v = P.x1[(size_t)row * P.K1 + c];
            else if (c < K) v = P.x2[(size_t)row * P.K2 + (c - P.K1)];
        }
        bufA[r * MC_LD + c] = v;
    }
    __syncthreads();
    fwd_layer(bufA, bufB, red, P.w[0], K, P.b[0], P.h[0], row0, P.M, P.alpha);
    __syncthreads();            // (out complete; the slabs free again)
    fwd_layer(bufB, bufA, red, P.w[1], MC_H, P.b[1], P.h[1], row0, P.M, P.alpha);
    __syncthreads();
    fwd_layer(bufA, bufB, red, P.w[2], MC_H, P.b[2], P.h[2], row0, P.M, P.alpha);
    __syncthreads();
    // ---- logits[row] = h3[row] . w_out + b_out: one wave per row, 8 columns per lane, fixed-order halving -------------------------------
    {
        const int r = tid >> 6, lane = tid & 63;
        const float* hr = bufB + r * MC_LD + 8 * lane;
        const float* wo = P.w_out + 8 * lane;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float4 hv = *reinterpret_cast<const float4*>(hr + 4 * q);
            const float4 wv = *reinterpret_cast<const float4*>(wo + 4 * q);
            s = fmaf(hv.x, wv.x, fmaf(hv.y, wv.y, fmaf(hv.z, wv.z, fmaf(hv.w, wv.w, s))));
        }
        s = wave_sum(s);
        if (lane == 0 && row0 + r < P.M) P.logits[row0 + r] = s + P.b_out[0];
    }
}

struct MlpBwdParams {
    const float* gh3;
    const float* wt[3];      // transposed weights written by the forward launch
    const float* h1;
    const float* h2;
    float* gh2;
    float* gh1;
    float* dx1;
    float* dx2;
    int M, K1, K2, KP, spread;
    float alpha;
};

// the rows' reference activations of a hidden layer: thread's two 16-byte pieces of the [8][512] tile (piece u = tid + 512 q)
__device__ __forceinline__ void fetch_ref(const float* __restrict__ h, int row0, int M, float4 (&ref)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int u = threadIdx.x + MC_THR * q, r = u >> 7, c4 = u & 127;
        ref[q] = (row0 + r < M) ? *reinterpret_cast<const float4*>(h + (size_t)(row0 + r) * MC_H + 4 * c4) : float4{1.f, 1.f, 1.f, 1.f};
    }
}

// hidden backward layer: out[8][512] = (in[8][512] Wt[512][512]) * lrelu'(ref), kept in LDS and written to gout when wanted
__device__ __forceinline__ void bwd_layer(const float* __restrict__ in, float* __restrict__ out, float* __restrict__ red,
                                          const float* __restrict__ Wt, const float4 (&ref)[2], float alpha, float* __restrict__ gout, int row0,
                                          int M) {
    mm_core<MC_NSH>(in, red, Wt, MC_H, MC_H);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int u = threadIdx.x + MC_THR * q, r = u >> 7, c4 = u & 127;
        float4 v = red_sum(red, u);
        v.x = ref[q].x > 0.f ? v.x : alpha * v.x; v.y = ref[q].y > 0.f ? v.y : alpha * v.y;
        v.z = ref[q].z > 0.f ? v.z : alpha * v.z; v.w = ref[q].w > 0.f ? v.w : alpha * v.w;
        *reinterpret_cast<float4*>(out + r * MC_LD + 4 * c4) = v;
        if (gout && row0 + r < M) *reinterpret_cast<float4*>(gout + (size_t)(row0 + r) * MC_H + 4 * c4) = v;
    }
}

__global__ __launch_bounds__(MC_THR) void mlp_chain_bwd_k(const MlpBwdParams P) {
    if (blockIdx.x % P.spread) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bufA = smem;
    float* bufB = smem + MC_ACT;
    float* red = smem + 2 * MC_ACT;
    const int tid = threadIdx.x;
    const int row0 = (blockIdx.x / P.spread) * MC_ROWS;
    float4 ref[2];
    fetch_ref(P.h2, row0, P.M, ref);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int u = tid + MC_THR * q, r = u >> 7, c4 = u & 127;
        const float4 v = (row0 + r < P.M) ? *reinterpret_cast<const float4*>(P.gh3 + (size_t)(row0 + r) * MC_H + 4 * c4) : float4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<float4*>(bufA + r * MC_LD + 4 * c4) = v;
    }
    __syncthreads();
    bwd_layer(bufA, bufB, red, P.wt[2], ref, P.alpha, P.gh2, row0, P.M);     // gh2 = (gh3 w3^T) * lrelu'(h2)
    fetch_ref(P.h1, row0, P.M, ref);
    __syncthreads();
    bwd_layer(bufB, bufA, red, P.wt[1], ref, P.alpha, P.gh1, row0, P.M);     // gh1 = (gh2 w2^T) * lrelu'(h1)
    if (!P.dx1) return;
    __syncthreads();
    mm_core<MC_NSH>(bufA, red, P.wt[0], MC_H, P.KP);                                 // [dx1 | dx2] = gh1 w1^T
    const int K = P.K1 + P.K2;
    for (int u = tid; u < MC_ROWS * K; u += MC_THR) {
        const int r = u / K, c = u - r * K, row = row0 + r;
        if (row < P.M) {
            const float* p = red + r * MC_H + c;
            const float v = ((p[0] + p[MC_ROWS * MC_H]) + p[2 * MC_ROWS * MC_H]) + p[3 * MC_ROWS * MC_H];
            if (c < P.K1) P.dx1[(size_t)row * P.K1 + c] = v;
            else P.dx2[(size_t)row * P.K2 + (c - P.K1)] = v;
        }
    }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int spread_factor() {
    static const int v = [] { const char* e = getenv("GGAN_MLP_SPREAD"); const int x = e ? atoi(e) : 8; return x >= 1 ? x : 1; }();
    return v;
}

int ensure_lds() {
    static const int rc = [] {
        if (hipFuncSetAttribute((const void*)mlp_chain_fwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MC_LDS) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void*)mlp_chain_bwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MC_LDS) != hipSuccess) return -1;
        return 0;
    }();
    if (rc) set_error("mlp_chain: cannot raise the dynamic LDS limit");
    return rc;
}

static_assert(MC_TT * MC_TLD <= MC_RED + 2 * MC_ACT, "transposer tile fits the launch's LDS");

}  // namespace

namespace ggan {

size_t mlp_chain_wt_floats(int K1, int K2) { return (size_t)MC_H * (((K1 + K2 + 3) / 4) * 4) + 2 * (size_t)MC_H * MC_H; }

int mlp_chain_fwd_launch(int M, int K1, int K2, const float* x1, const float* x2, const float* const w[3], const float* const b[3],
                         const float* w_out, const float* b_out, float alpha, float* const h[3], float* logits, float* wt, hipStream_t s) {
    MlpFwdParams P;
    memset(&P, 0, sizeof(P));
    P.x1 = x1; P.x2 = x2; P.w_out = w_out; P.b_out = b_out; P.logits = logits;
    for (int i = 0; i < 3; ++i) {
        P.w[i] = w[i]; P.b[i] = b[i]; P.h[i] = h[i];
        if (!al16(w[i]) || !al16(b[i]) || !al16(h[i])) { set_error("mlp_chain_fwd: w, b, h must be 16-byte aligned"); return -1; }
    }
    if (!al16(w_out) || !al16(wt)) { set_error("mlp_chain_fwd: w_out, wt must be 16-byte aligned"); return -1; }
    P.M = M; P.K1 = K1; P.K2 = K2; P.alpha = alpha;
    P.KP = ((K1 + K2 + 3) / 4) * 4;
    if (wt) { P.wt[0] = wt; P.wt[1] = wt + (size_t)MC_H * P.KP; P.wt[2] = P.wt[1] + (size_t)MC_H * MC_H; }
    P.spread = spread_factor();
    if (ensure_lds()) return -2;
    P.nblk = cdiv(M, MC_ROWS);
    const int grid = P.nblk * P.spread + ((wt && P.spread == 1) ? MC_TBLOCKS : 0);
    const double fl = 2.0 * M * ((double)(K1 + K2) * MC_H + 2.0 * MC_H * MC_H + MC_H);
    GGAN_LAUNCH("mlp_chain_fwd_k", fl, 0, mlp_chain_fwd_k, dim3(grid), dim3(MC_THR), MC_LDS, s, P);
    return 0;
}

int mlp_chain_bwd_launch(int M, int K1, int K2, const float* gh3, const float* wt, const float* h1, const float* h2, float alpha,
                         float* gh2, float* gh1, float* dx1, float* dx2, hipStream_t s) {
    MlpBwdParams P;
    memset(&P, 0, sizeof(P));
    P.gh3 = gh3; P.h1 = h1; P.h2 = h2; P.gh2 = gh2; P.gh1 = gh1; P.dx1 = dx1; P.dx2 = dx2;
    if (!wt || !al16(wt) || !al16(gh3) || !al16(h1) || !al16(h2) || !al16(gh2) || !al16(gh1)) {
        set_error("mlp_chain_bwd: wt, gh, h must be 16-byte aligned (wt non-null)");
        return -1;
    }
    P.M = M; P.K1 = K1; P.K2 = K2; P.alpha = alpha;
    P.KP = ((K1 + K2 + 3) / 4) * 4;
    P.wt[0] = wt; P.wt[1] = wt + (size_t)MC_H * P.KP; P.wt[2] = P.wt[1] + (size_t)MC_H * MC_H;
    P.spread = spread_factor();
    if (ensure_lds()) return -2;
    const int nblk = cdiv(M, MC_ROWS);
    const double fl = 2.0 * M * (2.0 * MC_H * MC_H + (dx1 ? (double)(K1 + K2) * MC_H : 0.0));
    GGAN_LAUNCH("mlp_chain_bwd_k", fl, 0, mlp_chain_bwd_k, dim3(nblk * P.spread), dim3(MC_THR), MC_LDS, s, P);
    return 0;
}

}  // namespace ggan
