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
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams P) {
    __shared__ __attribute__((aligned(16))) float As[BK * LDP];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, split = blockIdx.z;
    const int kb = split * P.kps, ke = min(kb + P.kps, P.K);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool do_colsum = P.colsum != nullptr && blockIdx.y == 0 && tid < BN;   // rows beyond K are zero in the tile
    float csum = 0.f;
    float4 ra, rb;
    load_tile<!TA>(P.A, P.lda, P.M, m0, kb, ke, P.vecA, ra);
    load_tile<TB>(P.B, P.ldb, P.N, n0, kb, ke, P.vecB, rb);
    for (int k0 = kb; k0 < ke; k0 += BK) {
        __syncthreads();
        store_tile<!TA>(As, ra);
        store_tile<TB>(Bs, rb);
        __syncthreads();
        if (k0 + BK < ke) {
            load_tile<!TA>(P.A, P.lda, P.M, m0, k0 + BK, ke, P.vecA, ra);
            load_tile<TB>(P.B, P.ldb, P.N, n0, k0 + BK, ke, P.vecB, rb);
        }
        if (do_colsum) {
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) csum += Bs[kk * LDP + tid];
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = As[(kk + half) * LDP + wm * 32 + l31];
            const float b = Bs[(kk + half) * LDP + wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    if (do_colsum && n0 + tid < P.N) P.colsum[n0 + tid] = csum;
    const int n = n0 + wn * 32 + l31;
    if (n >= P.N) return;
    const bool direct = P.SK == 1;
    float* Cp = direct ? P.C : P.C + (size_t)split * P.out_elems;
    const float bv = (direct && P.bias) ? P.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < P.M) {
            float v = acc[r];
            if (direct) v = act_apply(v + bv, P.act, P.alpha);
            Cp[(size_t)m * P.N + n] = v;
        }
    }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" {

size_t ggan_gemm_workspace(int M, int N, int K) {
    (void)K;
    return (size_t)64 * M * N * sizeof(float);
}

static int gemm_launch(int ta, int tb, int M, int N, int K, const float* A, const float* B, const float* bias, float* C,
                       float* colsum, int act, float alpha, void* ws, size_t ws_bytes, hipStream_t s) {
    GemmParams P;
    memset(&P, 0, sizeof(P));
    P.A = A; P.B = B; P.bias = bias; P.C = C;
    P.M = M; P.N = N; P.K = K;
    P.lda = ta ? M : K;
    P.ldb = tb ? K : N;
    // float4 legality: base aligned, leading dimension multiple of 4 (row starts stay aligned)
    P.vecA = al16(A) && (P.lda % 4 == 0);
    P.vecB = al16(B) && (P.ldb % 4 == 0);
    P.act = act; P.alpha = alpha;
    P.out_elems = (size_t)M * N;
    P.colsum = colsum;
    ws = ws_scratch(ws, ws_bytes);
    const int gx = cdiv(N, BN), gy = cdiv(M, BM);
    int sk = 1;
    if (!colsum && ws) {       // (a fused column sum needs the whole K range in one workgroup)
        const char* e = getenv("GGAN_GEMM_SK");
        if (e) sk = atoi(e);
        else {
            const int base = gx * gy;
            sk = 256 / base;
            const int max_sk = K / (4 * BK);
            if (sk > max_sk) sk = max_sk;
            if (sk > 64) sk = 64;
        }
        if (sk < 1) sk = 1;
        while (sk > 1 && (size_t)sk * P.out_elems * sizeof(float) > ws_bytes) sk /= 2;
    }
    P.kps = cdiv(cdiv(K, sk), BK) * BK;
    P.SK = cdiv(K, P.kps);
    if (P.SK > 1) P.C = (float*)ws;
    const double fl = 2.0 * M * N * (double)K;
    const dim3 grid(gx, gy, P.SK), block(256);
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

int ggan_gemm_colsum(int ta, int M, int N, int K, const float* A, const float* B, float* C, float* colsum_b, void* ws,
                     size_t ws_bytes, ggan_stream_t stream) {
    GGAN_CHECK_ARG(A && B && C && colsum_b, "null pointer");
    GGAN_CHECK_ARG(M > 0 && N > 0 && K > 0, "bad shape");
    return gemm_launch(ta, 0, M, N, K, A, B, nullptr, C, colsum_b, GGAN_ACT_NONE, 0.f, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
