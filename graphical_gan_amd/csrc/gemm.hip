// fp32-MFMA GEMM for the Linear layers (tf.matmul + bias_add, tflib/ops/linear.py:133-146) and
// their gradients: C[M,N] = op(A) op(B) (+bias[N]) (+act), row-major, op = identity or transpose.
//
// 64x64 workgroup tile, BK=16, 2x2 waves of 32x32, v_mfma_f32_32x32x2_f32 (exact fp32).  op(A) is the
// MFMA A operand (rows -> accumulator registers) and op(B) the B operand (cols -> lanes), so stores run
// along the contiguous N dimension.  Both LDS tiles are k-major ([BK][64+pad]): a fragment read is 32
// consecutive floats per half-wave (conflict-free).  The minibatch dimension is only 64-128 wide, so
// long-K products (Extractor.Output K=4096, Discriminator.zx1 K=4608) are spread over the chip with a
// deterministic split-K (partial slabs + launch_splitk_reduce, no atomics).
#include "common.h"
#include "conv.h"
#include <stdlib.h>
using namespace ggan;

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDP = 68;   // LDP: padded tile row (floats)
constexpr int KSTEP = 32;        // k per main-loop step
constexpr int LDPK = 66;         // tile row for k-contiguous operands (scalar scatter: 2 (mod 8) keeps it conflict free)

struct GemmParams {
    const float* A;
    const float* B;
    const float* bias;
    float* C;
    int M, N, K;
    int lda, ldb;      // leading dimensions of the STORED matrices
    int kps;           // k per split (multiple of BK)
    int SK;
    int vecA, vecB;    // float4 loads legal
    int act;
    float alpha;
    size_t out_elems;
    float* colsum;     // optional: column sums of op(B) (SK == 1, tb == 0), from the B tiles staged anyway
    // optional activation-derivative mask on one operand: element e of A (or B) becomes act_grad(e, ref[e]); `ref` has the
    // operand's own layout (the forward output of the layer whose gradient this is)
    const float* a_ref;
    const float* b_ref;
    int ref_act;
    float ref_alpha;
    // "concatenated" operands without the concatenation (the critic's Linear on [conv features | latent features]):
    //   A2 != null: op(A) is [A | A2] side by side along its second stored dimension -- columns (k, or m when TA) >= a_split
    //               come from A2 (leading dimension lda2); a_split is a multiple of the tile step, so a tile has ONE source
    //   C2 != null: output columns >= c_split go to C2 (leading dimension N - c_split), the others to C with leading dimension
    //               c_split (direct stores only: SK == 1)
    const float* A2;
    int a_split, lda2;
    float* C2;
    int c_split;
};

// Load a [64 rows(r) x 16 k] tile of a matrix into LDS as T[k][r].
//   KCONTIG:  element (r,k) at base[r*ld + k]   (k contiguous in memory)
//   !KCONTIG: element (r,k) at base[k*ld + r]   (r contiguous in memory)
template <bool KCONTIG>
__device__ __forceinline__ void load_tile(const float* __restrict__ base, int ld, int R, int r0, int k0, int kend,
                                          int vec, float4& reg) {
    const int tid = threadIdx.x;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (KCONTIG) {
        const int kq = tid >> 6, r = r0 + (tid & 63), k = k0 + kq * 4;
        if (r < R) {
            const float* p = base + (size_t)r * ld + k;
            if (vec && k + 3 < kend) {
                const float4 t = *reinterpret_cast<const float4*>(p);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (k + j < kend) v[j] = p[j];
            }
        }
    } else {
        const int kk = tid >> 4, r = r0 + (tid & 15) * 4, k = k0 + kk;
        if (k < kend) {
            const float* p = base + (size_t)k * ld + r;
            if (vec && r + 3 < R) {
                const float4 t = *reinterpret_cast<const float4*>(p);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (r + j < R) v[j] = p[j];
            }
        }
    }
    reg = make_float4(v[0], v[1], v[2], v[3]);
}

// masked operand (compile-time): g * act'(ref), same addressing for both; the raw pair is kept in registers while the
// MFMA block of the current tile runs and combined when it is committed to LDS
template <bool KCONTIG, bool MASK>
__device__ __forceinline__ void load_tile_m(const float* __restrict__ base, const float* __restrict__ ref, int ld, int R,
                                            int r0, int k0, int kend, int vec, float4& reg, float4& rref) {
    load_tile<KCONTIG>(base, ld, R, r0, k0, kend, vec, reg);
    if (MASK) load_tile<KCONTIG>(ref, ld, R, r0, k0, kend, vec, rref);
}
template <bool MASK>
__device__ __forceinline__ float4 apply_mask(const float4& g, const float4& r, int act, float alpha) {
    if (!MASK) return g;
    return make_float4(act_grad(g.x, r.x, act, alpha), act_grad(g.y, r.y, act, alpha), act_grad(g.z, r.z, act, alpha),
                       act_grad(g.w, r.w, act, alpha));
}

// ---- one main-loop step (KSTEP = 32 k) of an operand: [64 rows x 32 k] -> LDS T[k][r] ------------------------------------
// k-contiguous operands are read as 128-byte runs: 8 lanes x float4 cover 32 consecutive k of one row (the 64 rows of the
// earlier row-per-lane mapping touched 64 different cache lines per wave-load, 16 useful bytes each).
template <bool KCONTIG>
__device__ __forceinline__ void load_step_tile(const float* __restrict__ base, int ld, int R, int r0, int k0, int kend,
                                               int vec, float4 (&reg)[2]) {
    const int tid = threadIdx.x;
    if (KCONTIG) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int unit = tid + 256 * u, row = r0 + (unit >> 3), k = k0 + (unit & 7) * 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (row < R) {
                const float* p = base + (size_t)row * ld + k;
                if (vec && k + 3 < kend) {
                    const float4 t = *reinterpret_cast<const float4*>(p);
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k + j < kend) v[j] = p[j];
                }
            }
            reg[u] = make_float4(v[0], v[1], v[2], v[3]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < 2; ++u) load_tile<false>(base, ld, R, r0, k0 + u * BK, kend, vec, reg[u]);
    }
}

template <bool KCONTIG>
__device__ __forceinline__ void store_step_tile(float* T, const float4 (&reg)[2]) {
    const int tid = threadIdx.x;
    if (KCONTIG) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int unit = tid + 256 * u, row = unit >> 3, kq = unit & 7;
            float* d = T + (kq * 4) * LDPK + row;
            d[0] = reg[u].x; d[LDPK] = reg[u].y; d[2 * LDPK] = reg[u].z; d[3 * LDPK] = reg[u].w;
        }
    } else {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kk = tid >> 4, r = (tid & 15) * 4;
            *reinterpret_cast<float4*>(T + (u * BK + kk) * LDP + r) = reg[u];
        }
    }
}

template <bool KCONTIG>
__device__ __forceinline__ void store_tile(float* T, const float4& reg) {
    const int tid = threadIdx.x;
    if (KCONTIG) {
        const int kq = tid >> 6, r = tid & 63;
        T[(kq * 4 + 0) * LDP + r] = reg.x;
        T[(kq * 4 + 1) * LDP + r] = reg.y;
        T[(kq * 4 + 2) * LDP + r] = reg.z;
        T[(kq * 4 + 3) * LDP + r] = reg.w;
    } else {
        const int kk = tid >> 4, r = (tid & 15) * 4;
        *reinterpret_cast<float4*>(T + kk * LDP + r) = reg;
    }
}

// TA: A stored [K,M] (transposed)  => element (m,k) at A[k*lda + m]  => !KCONTIG
// TB: B stored [N,K] (transposed)  => element (n,k) at B[n*ldb + k]  =>  KCONTIG
// MASK: 0 none, 1 activation-derivative mask on A, 2 on B
template <bool TA, bool TB, int MASK = 0>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams P) {
    // LDS double-buffered so a step costs ONE barrier: the next step's global loads are issued before the MFMA block, parked in
    // registers, and committed to the other buffer after it
    constexpr bool AK = !TA, BKc = TB;                     // operand is k-contiguous in memory
    constexpr int LA = AK ? LDPK : LDP, LB = BKc ? LDPK : LDP;
    constexpr int TSZ = KSTEP * LDP;                       // (sized for the wider row)
    __shared__ __attribute__((aligned(16))) float As[2 * TSZ];
    __shared__ __attribute__((aligned(16))) float Bs[2 * TSZ];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, split = blockIdx.z;
    const int kb = split * P.kps, ke = min(kb + P.kps, P.K);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool do_colsum = P.colsum != nullptr && blockIdx.y == 0 && tid < BN;   // rows beyond K are zero in the tile
    float csum = 0.f;
    float4 ra[2], rb[2], rma[2], rmb[2];
    auto load_step = [&](int k0) {
        if (P.A2 == nullptr) {
            load_step_tile<AK>(P.A, P.lda, P.M, m0, k0, ke, P.vecA, ra);
        } else if (AK) {        // element (m, k) at A[m*lda + k]: the sources split the k range
            const bool second = k0 >= P.a_split;
            load_step_tile<AK>(second ? P.A2 - P.a_split : P.A, second ? P.lda2 : P.lda, P.M, m0, k0, second ? ke : min(ke, P.a_split),
                               P.vecA, ra);
        } else {                // element (m, k) at A[k*lda + m]: the sources split the m range
            const bool second = m0 >= P.a_split;
            load_step_tile<AK>(second ? P.A2 - P.a_split : P.A, second ? P.lda2 : P.lda, second ? P.M : P.a_split, m0, k0, ke,
                               P.vecA, ra);
        }
        if (MASK == 1) load_step_tile<AK>(P.a_ref, P.lda, P.M, m0, k0, ke, P.vecA, rma);
        load_step_tile<BKc>(P.B, P.ldb, P.N, n0, k0, ke, P.vecB, rb);
        if (MASK == 2) load_step_tile<BKc>(P.b_ref, P.ldb, P.N, n0, k0, ke, P.vecB, rmb);
    };
    auto store_step = [&](int buf) {
        if (MASK == 1) {
#pragma unroll
            for (int u = 0; u < 2; ++u) ra[u] = apply_mask<true>(ra[u], rma[u], P.ref_act, P.ref_alpha);
        }
        if (MASK == 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) rb[u] = apply_mask<true>(rb[u], rmb[u], P.ref_act, P.ref_alpha);
        }
        store_step_tile<AK>(As + buf * TSZ, ra);
        store_step_tile<BKc>(Bs + buf * TSZ, rb);
    };
    load_step(kb);
    store_step(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = kb; k0 < ke; k0 += KSTEP, buf ^= 1) {
        const bool more = k0 + KSTEP < ke;
        if (more) load_step(k0 + KSTEP);
        const float* Ab = As + buf * TSZ;
        const float* Bb = Bs + buf * TSZ;
        if (do_colsum) {
#pragma unroll
            for (int kk = 0; kk < KSTEP; ++kk) csum += Bb[kk * LB + tid];
        }
#pragma unroll
        for (int kk = 0; kk < KSTEP; kk += 2) {
            const float a = Ab[(kk + half) * LA + wm * 32 + l31];
            const float b = Bb[(kk + half) * LB + wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (more) store_step(buf ^ 1);
        __syncthreads();
    }
    if (do_colsum && n0 + tid < P.N) P.colsum[n0 + tid] = csum;
    // epilogue: global stores are issue-bound (16 dword stores per lane took longer than the k-loop of the short-K layers), so
    // the tile goes through LDS and leaves as one float4 row segment per thread and pass (4 passes)
    constexpr int LC = BN + 4;
    float* Cs = As;                                   // 64 x 68 floats <= 2 * TSZ (all waves are past the last barrier)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ml = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        Cs[ml * LC + wn * 32 + l31] = acc[r];
    }
    __syncthreads();
    const bool direct = P.SK == 1;
    float* Cp = direct ? P.C : P.C + (size_t)split * P.out_elems;
    int ldc = P.N, ncol0 = 0, nend = P.N;       // output columns [ncol0, nend) live in Cp with leading dimension ldc
    if (P.C2 != nullptr) {
        if (n0 >= P.c_split) { Cp = P.C2; ldc = P.N - P.c_split; ncol0 = P.c_split; }
        else { ldc = P.c_split; nend = P.c_split; }
    }
    const bool vec_out = (ldc & 3) == 0 && ((uintptr_t)Cp & 15) == 0;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int idx = tid + pass * 256, ml = idx >> 4, c4 = (idx & 15) * 4;
        const int m = m0 + ml, n = n0 + c4;
        if (m >= P.M || n >= P.N) continue;
        float4 v = *reinterpret_cast<const float4*>(Cs + ml * LC + c4);
        float* vv = reinterpret_cast<float*>(&v);
        if (direct) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float bv = (P.bias && n + q < P.N) ? P.bias[n + q] : 0.f;
                vv[q] = act_apply(vv[q] + bv, P.act, P.alpha);
            }
        }
        float* dst = Cp + (size_t)m * ldc + (n - ncol0);
        if (vec_out && n + 3 < nend) {
            *reinterpret_cast<float4*>(dst) = v;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (n + q < nend) dst[q] = vv[q];
        }
    }
}

// N == 1 (the critics' Output layers, 512 -> 1): a matrix-vector product -- one wave per row, no tiles, no split-K, no reduce
__global__ __launch_bounds__(256) void gemv_rows_k(const float* __restrict__ A, const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ C, int M, int K, int act, float alpha) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* a = A + (size_t)row * K;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s = fmaf(a[k], w[k], s);
    s = wave_sum(s);
    if (lane == 0) C[row] = act_apply(s + (bias ? bias[0] : 0.f), act, alpha);
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" {

size_t ggan_gemm_workspace(int M, int N, int K) {
    (void)K;
    return (size_t)64 * M * N * sizeof(float);
}

static int gemm_launch(int ta, int tb, int M, int N, int K, const float* A, const float* B, const float* bias, float* C,
                       float* colsum, int act, float alpha, void* ws, size_t ws_bytes, hipStream_t s,
                       const float* a_ref = nullptr, const float* b_ref = nullptr, int ref_act = 0, float ref_alpha = 0.f,
                       const float* A2 = nullptr, int a_split = 0, float* C2 = nullptr, int c_split = 0) {
    if (N == 1 && !ta && !tb && !colsum && !a_ref && !b_ref) {
        GGAN_LAUNCH("gemv_rows_k", 2.0 * M * K, 0, gemv_rows_k, dim3(cdiv(M, 4)), dim3(256), 0, s, A, B, bias, C, M, K, act, alpha);
        return 0;
    }
    GemmParams P;
    memset(&P, 0, sizeof(P));
    P.a_ref = a_ref; P.b_ref = b_ref; P.ref_act = ref_act; P.ref_alpha = ref_alpha;
    P.A = A; P.B = B; P.bias = bias; P.C = C;
    P.M = M; P.N = N; P.K = K;
    P.lda = ta ? M : K;
    P.ldb = tb ? K : N;
    if (A2) {
        const int whole = ta ? M : K;
        if (a_split <= 0 || a_split >= whole || (a_split % 64) || a_ref) { set_error("gemm: bad operand split"); return -1; }
        P.A2 = A2; P.a_split = a_split;
        P.lda = a_split; P.lda2 = whole - a_split;
    }
    if (C2) {
        if (c_split <= 0 || c_split >= N || (c_split % BN) || colsum || bias || act != GGAN_ACT_NONE) { set_error("gemm: bad output split"); return -1; }
        P.C2 = C2; P.c_split = c_split;
    }
    // float4 legality: base aligned, leading dimension multiple of 4 (row starts stay aligned)
    P.vecA = al16(A) && (P.lda % 4 == 0) && (!a_ref || al16(a_ref)) && (!A2 || (al16(A2) && P.lda2 % 4 == 0));
    P.vecB = al16(B) && (P.ldb % 4 == 0) && (!b_ref || al16(b_ref));
    P.act = act; P.alpha = alpha;
    P.out_elems = (size_t)M * N;
    P.colsum = colsum;
    ws = ws_scratch(ws, ws_bytes);
    const int gx = cdiv(N, BN), gy = cdiv(M, BM);
    int sk = 1;
    if (!colsum && ws && !C2) {       // (a fused column sum needs the whole K range in one workgroup; so does a split output)
        const char* e = getenv("GGAN_GEMM_SK");
        if (e) sk = atoi(e);
        else {
            // a workgroup's k-step is latency-bound (~0.6 us measured), so spread K over the idle CUs down to one
            // BK-step per split; measured optimum on the hot path's shapes (tools/bench_gemm.py)
            const int base = gx * gy;
            // (long reductions -- the Conv3D patch-matrix filter gradients, K = 10^4..10^6 rows: four workgroups per CU and up to
            // 256 slabs, so each workgroup's serial chain stays ~10^3 rows)
            const bool tall = K >= 16384;
            sk = (tall ? 1024 : 256) / base;
            const int max_sk = K / BK;
            if (sk > max_sk) sk = max_sk;
            if (sk > (tall ? 256 : 64)) sk = tall ? 256 : 64;
        }
        if (sk < 1) sk = 1;
        while (sk > 1 && (size_t)sk * P.out_elems * sizeof(float) > ws_bytes) sk /= 2;
    }
    P.kps = cdiv(cdiv(K, sk), BK) * BK;
    P.SK = cdiv(K, P.kps);
    if (P.SK > 1) P.C = (float*)ws;
    const double fl = 2.0 * M * N * (double)K;
    const dim3 grid(gx, gy, P.SK), block(256);
    if (a_ref || b_ref) {
        if (a_ref && !b_ref && !ta && tb) { GGAN_LAUNCH("gemm_kernel<false, true, 1>", fl, 0, (gemm_kernel<false, true, 1>), grid, block, 0, s, P); }
        else if (b_ref && !a_ref && ta && !tb) { GGAN_LAUNCH("gemm_kernel<true, false, 2>", fl, 0, (gemm_kernel<true, false, 2>), grid, block, 0, s, P); }
        else { set_error("gemm: unsupported mask/transposition combination"); return -1; }
    } else
    if (!ta && !tb) { GGAN_LAUNCH("gemm_kernel<false, false>", fl, 0, (gemm_kernel<false, false>), grid, block, 0, s, P); }
    else if (!ta && tb) { GGAN_LAUNCH("gemm_kernel<false, true>", fl, 0, (gemm_kernel<false, true>), grid, block, 0, s, P); }
    else if (ta && !tb) { GGAN_LAUNCH("gemm_kernel<true, false>", fl, 0, (gemm_kernel<true, false>), grid, block, 0, s, P); }
    else { GGAN_LAUNCH("gemm_kernel<true, true>", fl, 0, (gemm_kernel<true, true>), grid, block, 0, s, P); }
    if (P.SK > 1) return launch_splitk_reduce((const float*)ws, P.SK, P.out_elems, C, bias, N, 1, act, alpha, s);
    return 0;
}

int ggan_gemm(int ta, int tb, int M, int N, int K, const float* A, const float* B, const float* bias, float* C, int act,
              float alpha, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    GGAN_CHECK_ARG(A && B && C, "null pointer");
    GGAN_CHECK_ARG(M > 0 && N > 0 && K > 0, "bad shape");
    return gemm_launch(ta, tb, M, N, K, A, B, bias, C, nullptr, act, alpha, ws, ws_bytes, (hipStream_t)stream);
}

int ggan_gemm_split(int ta, int tb, int M, int N, int K, const float* A, const float* A2, int a_split, const float* B,
                    const float* bias, float* C, float* C2, int c_split, float* colsum_b, int act, float alpha, void* ws,
                    size_t ws_bytes, ggan_stream_t stream) {
    GGAN_CHECK_ARG(A && B && C, "null pointer");
    GGAN_CHECK_ARG(M > 0 && N > 0 && K > 0, "bad shape");
    GGAN_CHECK_ARG(!colsum_b || !tb, "column sums need B stored [K,N]");
    return gemm_launch(ta, tb, M, N, K, A, B, bias, C, colsum_b, act, alpha, ws, ws_bytes, (hipStream_t)stream, nullptr, nullptr, 0,
                       0.f, A2, a_split, C2, c_split);
}

int ggan_gemm_colsum(int ta, int M, int N, int K, const float* A, const float* B, float* C, float* colsum_b, void* ws,
                     size_t ws_bytes, ggan_stream_t stream) {
    GGAN_CHECK_ARG(A && B && C && colsum_b, "null pointer");
    GGAN_CHECK_ARG(M > 0 && N > 0 && K > 0, "bad shape");
    return gemm_launch(ta, 0, M, N, K, A, B, nullptr, C, colsum_b, GGAN_ACT_NONE, 0.f, ws, ws_bytes, (hipStream_t)stream);
}

int ggan_linear_bwd_data_act(int M, int N, int K, const float* g, const float* y, int y_act, float y_alpha, const float* w,
                             float* dx, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    GGAN_CHECK_ARG(g && w && dx && (y || y_act == GGAN_ACT_NONE), "null pointer");
    GGAN_CHECK_ARG(M > 0 && N > 0 && K > 0, "bad shape");
    // dx[M,K] = (g * act'(y))[M,N] @ w[K,N]^T
    return gemm_launch(0, 1, M, K, N, g, w, nullptr, dx, nullptr, GGAN_ACT_NONE, 0.f, ws, ws_bytes, (hipStream_t)stream,
                       y_act != GGAN_ACT_NONE ? y : nullptr, nullptr, y_act, y_alpha);
}

int ggan_linear_bwd_weight_act(int M, int N, int K, const float* x, const float* g, const float* y, int y_act, float y_alpha,
                               float* dw, float* db, void* ws, size_t ws_bytes, ggan_stream_t stream) {
    GGAN_CHECK_ARG(x && g && dw && (y || y_act == GGAN_ACT_NONE), "null pointer");
    GGAN_CHECK_ARG(M > 0 && N > 0 && K > 0, "bad shape");
    // dw[K,N] = x[M,K]^T @ (g * act'(y))[M,N];  db[N] = column sums of the masked g (from the tiles staged anyway)
    return gemm_launch(1, 0, K, N, M, x, g, nullptr, dw, db, GGAN_ACT_NONE, 0.f, ws, ws_bytes, (hipStream_t)stream, nullptr,
                       y_act != GGAN_ACT_NONE ? y : nullptr, y_act, y_alpha);
}

}  // extern "C"
